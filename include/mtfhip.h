/*
 * mtfhip.h -- C ABI of libmtfhip.so: the MI355X (gfx950) implementation of MTF's
 * Lucas-Kanade inner loop (sample -> image gradient -> steepest-descent image ->
 * J^T r / J^T J), behind the reference's AppearanceModel / StateSpaceModel boundary.
 *
 * The reference (abhineet123/MTF) has no FFI or plugin ABI: its extension mechanism is
 * C++ subclassing of mtf::AppearanceModel (AM/include/mtf/AM/AppearanceModel.h:63-396,
 * ImageBase.h:51-191) and mtf::StateSpaceModel (SSM/include/mtf/SSM/StateSpaceModel.h:49-408).
 * Each entry point below is what one of those virtuals becomes once its data lives in HBM;
 * the adapter subclasses in mtf_amd/host/ (and the maintainer-side stub in INTEGRATION.md)
 * forward to them one-to-one.  Plain C types only, caller owns every host buffer, handles
 * own all device memory, every call returns 0 on success or a negative mtfhip_status
 * (text via mtfhip_last_error(); the adapters rethrow it as mtf::utils::Exception,
 * Utilities/include/mtf/Utilities/excpUtils.h:8-55).
 *
 * One handle type covers a single target and a batch: a `mtfhip_batch` holds B independent
 * targets (each with its own template, warp and N = resx*resy sample points) that share the
 * current image -- B = 1 is the AM/SSM pair of one tracker, B = 256 is GridTracker's patch
 * set (SM/src/GridTracker.cc:247-261), B = n_trackers is runMTF's concurrent targets.
 *
 * Host-side array layouts are the reference's Eigen column-major typedefs
 * (Macros/include/mtf/Macros/common.h:190-258), target-major when B > 1:
 *   pts 2xN (x,y interleaved) | grad_pts 8xN | pix_grad Nx2 (N Ix then N Iy)
 *   J NxS (S columns of N)    | H SxS column-major | corners 2x4 (TL,TR,BR,BL, x,y interleaved)
 * All arithmetic is IEEE double except the float32 image, exactly as in the reference.
 */
#ifndef MTFHIP_H
#define MTFHIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mtfhip_ctx mtfhip_ctx;
typedef struct mtfhip_batch mtfhip_batch;

typedef enum mtfhip_status {
	MTFHIP_OK = 0,
	MTFHIP_ERR_INVALID_ARG = -1,   /* mtf::utils::InvalidArgument */
	MTFHIP_ERR_NOT_IMPLEMENTED = -2, /* mtf::utils::FunctonNotImplemented */
	MTFHIP_ERR_LOGIC = -3,         /* mtf::utils::LogicError (call order) */
	MTFHIP_ERR_HIP = -4,           /* HIP runtime failure */
	MTFHIP_ERR_NO_DEVICE = -5
} mtfhip_status;

enum { MTFHIP_AM_SSD = 0, MTFHIP_AM_NCC = 1, MTFHIP_AM_MI = 2 };
enum { MTFHIP_SSM_HOMOGRAPHY = 0, MTFHIP_SSM_AFFINE = 1 };
enum { MTFHIP_SM_ESM = 0, MTFHIP_SM_FCLK = 1, MTFHIP_SM_ICLK = 2 };
/* pixel-Jacobian variants of StateSpaceModel.h:170-181 */
enum { MTFHIP_JAC_INIT = 0, MTFHIP_JAC_PIX = 1, MTFHIP_JAC_WARPED = 2, MTFHIP_JAC_APPROX = 3 };
/* device-resident buffers of a batch (per target sizes in doubles) */
enum {
	MTFHIP_BUF_I0 = 0,       /* N      ImageBase::I0 */
	MTFHIP_BUF_IT = 1,       /* N      ImageBase::It */
	MTFHIP_BUF_DI0_DX = 2,   /* N x 2  ImageBase::dI0_dx */
	MTFHIP_BUF_DIT_DX = 3,   /* N x 2  ImageBase::dIt_dx */
	MTFHIP_BUF_DF_DI0 = 4,   /* N      AppearanceModel::df_dI0 */
	MTFHIP_BUF_DF_DIT = 5,   /* N      AppearanceModel::df_dIt */
	MTFHIP_BUF_J0 = 6,       /* N x S  the SM's init_pix_jacobian / dI0_dpssm */
	MTFHIP_BUF_JT = 7,       /* N x S  the SM's curr_pix_jacobian / dIt_dpssm */
	MTFHIP_BUF_JM = 8,       /* N x S  the SM's mean_pix_jacobian (ESM jac/hess type Original) */
	MTFHIP_BUF_INIT_PTS = 9, /* 2 x N  StateSpaceModel::init_pts */
	MTFHIP_BUF_CURR_PTS = 10,/* 2 x N  StateSpaceModel::curr_pts */
	MTFHIP_BUF_GRAD_PTS = 11,/* 8 x N  StateSpaceModel::grad_pts */
	MTFHIP_BUF_INIT_Z = 12,  /* N      third row of ProjectiveBase::init_pts_hm */
	MTFHIP_BUF_CURR_Z = 13,  /* N      third row of ProjectiveBase::curr_pts_hm */
	MTFHIP_BUF_INIT_HXY = 14,/* 2 x N  first two rows of ProjectiveBase::init_pts_hm (homography keeps the
	                                   un-normalised DLT product, Homography.cc:66) */
	MTFHIP_BUF_CURR_HXY = 15,/* 2 x N  first two rows of ProjectiveBase::curr_pts_hm */
	/* second-order path (sec_ord_hess) */
	MTFHIP_BUF_D2I0_DX2 = 16,/* 4 x N  ImageBase::d2I0_dx2 (xx, xy, yx, yy per pixel) */
	MTFHIP_BUF_D2IT_DX2 = 17,/* 4 x N  ImageBase::d2It_dx2 */
	MTFHIP_BUF_HESS_PTS = 18,/* 16 x N StateSpaceModel::hess_pts */
	MTFHIP_BUF_D2I0_DP2 = 19,/* the SM's init_pix_hessian (Eigen S^2 x N): stored as S*S planes of N, plane r + S*c = entry (r,c) */
	MTFHIP_BUF_D2IT_DP2 = 20,/* the SM's curr_pix_hessian, same layout */
	MTFHIP_BUF_D2IM_DP2 = 21,/* the SM's mean_pix_hessian (ESM hess type Original), same layout */
	MTFHIP_BUF_COUNT = 22
};

typedef struct mtfhip_patch_desc {
	int am;                 /* MTFHIP_AM_* */
	int ssm;                /* MTFHIP_SSM_* */
	int resx, resy;         /* ImgParams / SSMParams resx, resy */
	double grad_eps;        /* ImgParams::grad_eps (1e-8, AM/include/mtf/AM/ImageBase.h:7-8) */
	double likelihood_alpha;/* AMParams::likelihood_alpha */
	int mi_n_bins;          /* MIParams::n_bins */
	double mi_pre_seed;     /* MIParams::pre_seed */
	int mi_partition_of_unity;
	double hess_eps;        /* ImgParams::hess_eps (1, AM/include/mtf/AM/ImageBase.h:9); <= 0 selects that default */
	int n_channels;         /* 1 (or 0): SSD / NCC / MI ; 3: MCSSD / MCNCC / MCMI (AM/src/MCSSD.cc, MCNCC.cc, MCMI.cc: the same class
	                           built with n_channels = 3).  Every per-pixel AM array then has n_pix * 3 rows interleaved per pixel,
	                           the image is CV_32FC3 (mtfhip_image_upload_mc).  The per-function entry points, the fused
	                           init_template / set_region / iterate / track calls (first-order Hessians; MCMI on the recompute
	                           passes), the candidate scorer and the particle filter apply; a grid of multi-channel patches takes
	                           the launch-per-pass loop instead of the one-launch grid kernel */
} mtfhip_patch_desc;

/* Search-method configuration; field meanings and enum values are the reference's
 * (SM/include/mtf/SM/ESMParams.h:13-17, FCLKParams.h:8, ICLKParams.h) */
typedef struct mtfhip_sm_desc {
	int sm;            /* MTFHIP_SM_* */
	int jac_type;      /* ESM: 0 Original, 1 DiffOfJacs */
	int hess_type;     /* ESM: 0 InitialSelf 1 CurrentSelf 2 SumOfSelf 3 Original 4 SumOfStd 5 Std
	                      FCLK/ICLK: 0 InitialSelf 1 CurrentSelf 2 Std */
	int chained_warp;
	int materialize;   /* 1: It, dIt_dx and Jt are written to HBM as the interface exposes them;
	                      0: kept in registers only (getters for them then fail with ERR_LOGIC) */
	int max_iters;     /* used by mtfhip_batch_track only */
	double epsilon;
	int leven_marq;
	double lm_delta_init, lm_delta_update;
	int sec_ord_hess;  /* second-order Hessians (ESMParams / FCLKParams / ICLKParams sec_ord_hess; off in every shipped config) */
} mtfhip_sm_desc;

/* ------------------------------------------------------------------ context */
const char *mtfhip_last_error(void);
int mtfhip_device_count(void);
/* `hip_stream` is a hipStream_t (or NULL for a stream owned by the context) */
int mtfhip_ctx_create(int device, void *hip_stream, mtfhip_ctx **out);
void mtfhip_ctx_destroy(mtfhip_ctx *ctx); /* destroy every batch created on the context first (a batch keeps a pointer to it) */
int mtfhip_ctx_synchronize(mtfhip_ctx *ctx);
void *mtfhip_ctx_stream(mtfhip_ctx *ctx);

/* Pre-processing on the device: what PreProcBase::processFrame + GaussianSmoothing::apply do with OpenCV for the default
 * CV_32FC1 output (Utilities/include/mtf/Utilities/preprocUtils.h:20-73, Utilities/src/preprocUtils.cc:108-127):
 * raw frame (uint8 or float32, 1 channel or 3 interleaved BGR) -> float32 -> gray (B*0.114f + G*0.587f + R*0.299f) ->
 * GaussianBlur(ksize x ksize, sigma_x, sigma_y) with BORDER_REFLECT_101 -> the context's current image.  The frame crosses
 * PCIe once, as it was captured (1 or 3 bytes per pixel instead of 4).  ksize 5 (the reference's default, sigma 3) or 0
 * (no smoothing); hist_eq and resize_factor: mtfhip_image_preprocess_ex.  OpenCV itself is absent from this image: the arithmetic
 * follows its float32 filter engine as restated in oracle/preproc_ref.py. */
enum { MTFHIP_DEPTH_U8 = 0, MTFHIP_DEPTH_F32 = 1 };
int mtfhip_image_preprocess(mtfhip_ctx *ctx, const void *host_raw, int rows, int cols, int row_stride_bytes, int channels,
	int depth, int ksize, double sigma_x, double sigma_y);
/* ... with the pre-processor's other two switches (PreProcBase(name, output_type, resize_factor, hist_eq), preprocUtils.h:28,56-58;
 * preprocUtils.cc:120-137): hist_eq -- the gray frame goes through 8 bits and cv::equalizeHist before the smoothing -- and
 * resize_factor != 1 -- cv::resize(INTER_LINEAR) of the smoothed frame to (int)(rows f) x (int)(cols f), which becomes the
 * current image */
int mtfhip_image_preprocess_ex(mtfhip_ctx *ctx, const void *host_raw, int rows, int cols, int row_stride_bytes, int channels,
	int depth, int ksize, double sigma_x, double sigma_y, int hist_eq, double resize_factor);
/* One level of PyramidalTracker's image pyramid (SM/src/PyramidalTracker.cc:88-97): dst's current image =
 * cv::pyrDown(src's image) when use_pyr_down (scale_factor 0.5), else cv::resize(INTER_LINEAR) + GaussianBlur(5x5, 3). */
int mtfhip_image_pyramid_level(mtfhip_ctx *dst, mtfhip_ctx *src, int dst_rows, int dst_cols, int use_pyr_down);
/* read-back of the current image (PreProcBase::getFrame) and its shape */
int mtfhip_image_download(mtfhip_ctx *ctx, float *host_img, int rows, int cols);
int mtfhip_image_shape(mtfhip_ctx *ctx, int *rows, int *cols);

/* ImageBase::setCurrImg (AM/src/ImageBase.cc:38-60).  The reference borrows the caller's
 * cv::Mat buffer, which the caller overwrites in place every frame, so upload must be
 * repeated per frame; `borrow` adopts a float32 image that is already in HBM.
 * A BORROWED image must stay unchanged until the next call of this library that synchronises with the context's stream
 * (mtfhip_ctx_synchronize, any call that returns results to the host): since r05 mtfhip_batch_init_template (its fused form) and
 * mtfhip_grid_reset(reinit) return with their sampling kernel still running.  Uploaded images are safe: the next upload is
 * ordered behind that kernel on the stream. */
int mtfhip_image_upload(mtfhip_ctx *ctx, const float *host_img, int height, int width, int row_stride);
/* CV_32FC3 input of the multi-channel appearance models: `channels` (1 or 3) interleaved floats per pixel, row_stride in floats */
int mtfhip_image_upload_mc(mtfhip_ctx *ctx, const float *host_img, int height, int width, int row_stride, int channels);
int mtfhip_image_borrow(mtfhip_ctx *ctx, const float *dev_img, int height, int width, int row_stride);

/* ------------------------------------------------------------------ batch of targets */
int mtfhip_batch_create(mtfhip_ctx *ctx, const mtfhip_patch_desc *desc, int n_targets, mtfhip_batch **out);
void mtfhip_batch_destroy(mtfhip_batch *b);
int mtfhip_batch_n_targets(const mtfhip_batch *b);
int mtfhip_batch_n_pix(const mtfhip_batch *b);       /* ImageBase::getNPix: sample points per target */
int mtfhip_batch_patch_size(const mtfhip_batch *b);  /* ImageBase::getPatchSize: n_pix * n_channels rows */
int mtfhip_batch_state_size(const mtfhip_batch *b);
/* Arithmetic of the kernels that are FP64-issue bound rather than HBM bound (the lean fused iteration, the one-launch
 * ICLK / grid loop, candidate scoring, the MI device passes):
 *   MTFHIP_MATH_REPLAY  the reference's operation order bit for bit (unfused mul/add, IEEE divisions, the grad_eps = 1e-8
 *                       central difference of utils::getImgGrad, Utilities/src/imgUtils.cc:233-254): per-pixel quantities
 *                       are bit-identical to the CPU path;
 *   MTFHIP_MATH_FAST    the same quantities with FMA contraction, one reciprocal per homography point
 *                       (SSM/src/Homography.cc:803-827 needs eight divisions) and the closed-form derivative of the bilinear
 *                       interpolant on interior non-integer points (integer coordinates / cell edges / borders replay the
 *                       finite difference): within north_star's 1e-5 on H, dp and 1e-9 on candidate scores.
 * Default: FAST (environment MTFHIP_MATH=replay selects REPLAY for new batches).  Everything that materialises
 * interface-visible arrays (It, dIt_dx, Jt ...) is always REPLAY: those kernels are bound by their stores. */
enum { MTFHIP_MATH_REPLAY = 0, MTFHIP_MATH_FAST = 1 };
int mtfhip_batch_set_math_mode(mtfhip_batch *b, int mode);
int mtfhip_batch_get_math_mode(const mtfhip_batch *b);
/* lazy read-back / overwrite of a device buffer (all targets, target-major);
 * the getters of ImageBase.h:83-89 / setters :93-100 and StateSpaceModel.h:82-88 */
int mtfhip_batch_read(mtfhip_batch *b, int buf, double *dst);
int mtfhip_batch_write(mtfhip_batch *b, int buf, const double *src);
/* raw device pointer of a buffer, for zero-copy interop (torch / RCCL) */
void *mtfhip_batch_device_ptr(mtfhip_batch *b, int buf);

/* ---- StateSpaceModel side ---- */
/* setCorners / initialize: Homography.cc:50-71, Affine.cc:64-88 (normalized_init = false) */
int mtfhip_ssm_set_corners(mtfhip_batch *b, const double *corners /* B x 8 */);
/* setState: ProjectiveBase.cc:41-49, Affine.cc:108-114 */
int mtfhip_ssm_set_state(mtfhip_batch *b, const double *states /* B x S */);
/* compositionalUpdate: Homography.cc:73-92, Affine.cc:90-106 */
int mtfhip_ssm_compositional_update(mtfhip_batch *b, const double *state_updates /* B x S */);
/* invertState: Homography.cc:109-114, Affine.cc:145-150 (pure host math, per target) */
int mtfhip_ssm_invert_state(mtfhip_batch *b, const double *states, double *inv_states);
/* updateGradPts / initializeGradPts: Homography.cc:803-827, Affine.cc:293-313 */
int mtfhip_ssm_update_grad_pts(mtfhip_batch *b, double grad_eps);
/* cmpt{Init,,Warped,Approx}PixJacobian: Homography.cc:157-358, Affine.cc:160-242.
 * grad_buf is MTFHIP_BUF_DI0_DX or _DIT_DX, dst_buf is MTFHIP_BUF_J0 / _JT / _JM */
int mtfhip_ssm_cmpt_pix_jacobian(mtfhip_batch *b, int variant, int grad_buf, int dst_buf);
int mtfhip_ssm_get_corners(mtfhip_batch *b, double *corners /* B x 8 */);
/* StateSpaceModel::estimateStateSigma (StateSpaceModel.h:336-338; ProjectiveBase.cc:201-213): state_sigma[k] = pix_sigma / the mean
 * over the sample points of |column k of dw/dp| -- nt::PF's pix_sigma -> sampler sigma (PF.cc:142-149) */
int mtfhip_ssm_estimate_state_sigma(mtfhip_batch *b, double pix_sigma, double *state_sigma /* B x S */);
int mtfhip_ssm_get_init_corners(mtfhip_batch *b, double *corners /* B x 8 */);
int mtfhip_ssm_get_state(mtfhip_batch *b, double *states /* B x S */);
int mtfhip_ssm_get_warp(mtfhip_batch *b, double *warps /* B x 9 row-major */);
/* applyWarpToCorners: ProjectiveBase.cc:137-144, Affine.cc:366-375 (host math) */
int mtfhip_ssm_apply_warp_to_corners(mtfhip_batch *b, const double *in_corners, const double *states,
	double *out_corners);
/* ---- SSM functions that are 3 x 3 algebra on the host: no device and no context needed (corners / points x, y interleaved) ---- */
int mtfhip_ssm_identity_warp(int ssm, double *state /* S */);                                         /* ProjectiveBase.cc:321-323 */
int mtfhip_ssm_compose_warps(int ssm, const double *state_1, const double *state_2, double *composed);  /* ProjectiveBase.cc:324-331: W(state_2) * W(state_1) */
int mtfhip_ssm_estimate_warp_from_corners(int ssm, const double *in_corners /* 8 */, const double *out_corners /* 8 */,
	double *state_update /* S */);                                                                      /* Homography.cc:877-883, Affine.cc:352-357 */
int mtfhip_ssm_apply_warp_to_pts(int ssm, const double *in_pts /* n x 2 */, int n_pts, const double *state, double *out_pts);
                                                                                                       /* ProjectiveBase.cc:142-160, Affine.cc:382-393 */
int mtfhip_ssm_additive_update(mtfhip_batch *b, const double *state_updates /* B x S */);              /* ProjectiveBase.cc:51-55 */

/* ---- ImageBase / AppearanceModel side ----
 * `pts` arguments: NULL means "the SSM's device-resident points of this batch" (the fast path the
 * adapters use when the PtsT reference they receive is the paired SSM's own getPts()/getGradPts());
 * otherwise B x 2N (or B x 8N) host doubles that are uploaded first. */
int mtfhip_am_initialize_pix_vals(mtfhip_batch *b, const double *pts);      /* ImageBase.cc:62-99 */
int mtfhip_am_update_pix_vals(mtfhip_batch *b, const double *pts);          /* ImageBase.cc:268-290 */
/* SSD::updateModel AM/src/SSD.cc:49-75, NCC::updateModel AM/src/NCC.cc:539-566 (called by the search methods when enable_learning is
 * set, NT/ESM.cc:293-295): template <- running (learning_rate outside [0, 1]) or weighted average with the patch at pts (NULL = the
 * current points), then AppearanceModel::reinitialize.  MI: ERR_NOT_IMPLEMENTED, as in the reference. */
int mtfhip_am_update_model(mtfhip_batch *b, const double *pts /* B x N x 2 or NULL */, double learning_rate);
int mtfhip_am_initialize_pix_grad(mtfhip_batch *b, const double *pts);      /* ImageBase.cc:101-132 (PtsT) */
int mtfhip_am_update_pix_grad(mtfhip_batch *b, const double *pts);          /* ImageBase.cc:292-314 */
int mtfhip_am_initialize_pix_grad_warped(mtfhip_batch *b, const double *grad_pts); /* ImageBase.cc:134-172 */
int mtfhip_am_update_pix_grad_warped(mtfhip_batch *b, const double *grad_pts);     /* ImageBase.cc:340-362 */
int mtfhip_am_initialize_similarity(mtfhip_batch *b);  /* SSDBase.cc:29-45, NCC.cc:55-95, MI.cc:207-287 */
int mtfhip_am_initialize_grad(mtfhip_batch *b);        /* SSDBase.cc:47-63, NCC.cc:97-122, MI.cc:299-332 */
int mtfhip_am_initialize_hess(mtfhip_batch *b);        /* SSDBase.h:58-63, MI.cc:443-459 */
int mtfhip_am_update_similarity(mtfhip_batch *b, int prereq_only); /* SSDBase.cc:75-96, NCC.cc:124-161, MI.cc:346-382 */
int mtfhip_am_update_curr_grad(mtfhip_batch *b);       /* SSDBase.cc:115-121, NCC.cc:196-234, MI.cc:426-442 */
int mtfhip_am_update_init_grad(mtfhip_batch *b);       /* SSDBase.h:67-72, NCC.cc:163-194, MI.cc:398-416 */
int mtfhip_am_get_similarity(mtfhip_batch *b, double *f /* B */);
int mtfhip_am_get_likelihood(mtfhip_batch *b, double *l /* B */);  /* SSD.h:41-43, NCC.cc:50-53, MI.cc:384-387 */
/* interfacing functions; J arguments are buffer ids, results go to host (B x S, B x S x S) */
int mtfhip_am_cmpt_init_jacobian(mtfhip_batch *b, int j0_buf, double *g);             /* AppearanceModel.h:146-149 */
int mtfhip_am_cmpt_curr_jacobian(mtfhip_batch *b, int jt_buf, double *g);             /* AppearanceModel.h:150-153 */
int mtfhip_am_cmpt_difference_of_jacobians(mtfhip_batch *b, int j0_buf, int jt_buf, double *g); /* SSDBase.cc:169-191 */
int mtfhip_am_cmpt_init_hessian(mtfhip_batch *b, int j0_buf, double *H);              /* SSDBase.cc:251-267, NCC.cc:282-303 */
int mtfhip_am_cmpt_curr_hessian(mtfhip_batch *b, int jt_buf, double *H);              /* SSDBase.cc:268-285, NCC.cc:304-335 */
int mtfhip_am_cmpt_self_hessian(mtfhip_batch *b, int jt_buf, double *H);              /* SSDBase.h:91-94, NCC.cc:337-389 */
int mtfhip_am_cmpt_sum_of_hessians(mtfhip_batch *b, int j0_buf, int jt_buf, double *H); /* SSDBase.cc:287-311 */
/* JM = (J0 + JT) / 2 on device: the SM-side `mean_pix_jacobian` of NT/ESM.cc:239-242 */
int mtfhip_sm_mean_jacobian(mtfhip_batch *b);

/* ---- second-order Hessians (sec_ord_hess = 1; NT/ESM.cc:315-377,406-432, NT/FCLK.cc:121-143,243-283, NT/ICLK.cc:96-124,223-252) ----
 * pts / hess_pts arguments: NULL = the batch's device-resident curr_pts / hess_pts, else host arrays (2 x N, 16 x N per target). */
int mtfhip_ssm_update_hess_pts(mtfhip_batch *b, double hess_eps);                 /* Homography.cc:829-875, Affine.cc:315-350 (= initializeHessPts) */
int mtfhip_am_initialize_pix_hess(mtfhip_batch *b, const double *pts);            /* ImageBase.cc:208-240 -> utils::getImgHess imgUtils.cc:334-366 */
int mtfhip_am_update_pix_hess(mtfhip_batch *b, const double *pts);                /* ImageBase.cc:316-338 */
int mtfhip_am_initialize_pix_hess_warped(mtfhip_batch *b, const double *pts, const double *hess_pts); /* ImageBase.cc:174-206 -> getWarpedImgHess imgUtils.cc:259-289 */
int mtfhip_am_update_pix_hess_warped(mtfhip_batch *b, const double *pts, const double *hess_pts);     /* ImageBase.cc:364-386 */
/* StateSpaceModel::cmpt{Init,,Warped,Approx}PixHessian (variant = MTFHIP_JAC_*): Homography.cc:360-425, 427-513, 515-618,
 * 696-801; Affine.cc:243-291 (Init, Warped; the other two return MTFHIP_ERR_NOT_IMPLEMENTED as the reference throws).
 * hess_buf = MTFHIP_BUF_D2I0_DX2 / _D2IT_DX2, grad_buf = _DI0_DX / _DIT_DX, dst_buf = _D2I0_DP2 / _D2IT_DP2 / _D2IM_DP2 */
int mtfhip_ssm_cmpt_pix_hessian(mtfhip_batch *b, int variant, int hess_buf, int grad_buf, int dst_buf);
int mtfhip_sm_mean_pix_hessian(mtfhip_batch *b);                                  /* D2IM = (D2I0 + D2IT) / 2, NT/ESM.cc:325 */
int mtfhip_am_cmpt_init_hessian2(mtfhip_batch *b, int j0_buf, int d2_buf, double *H); /* SSDBase.cc:313-343, NCC.cc:391-400, MI.cc:659-673 */
int mtfhip_am_cmpt_curr_hessian2(mtfhip_batch *b, int jt_buf, int d2_buf, double *H); /* SSDBase.cc:345-375, NCC.cc:401-410, MI.cc:680-694 */
int mtfhip_am_cmpt_self_hessian2(mtfhip_batch *b, int jt_buf, int d2_buf, double *H); /* SSDBase.h:95-98, MI.cc:696-733; NCC: not implemented */
int mtfhip_am_cmpt_sum_of_hessians2(mtfhip_batch *b, int j0_buf, int jt_buf, int d2_0_buf, int d2_t_buf, double *H); /* SSDBase.cc:377-415 */

/* ---- fused path: one launch per LK iteration for all targets of the batch ----
 * Appearance models: SSD (k_fused_ssd), NCC (k_fused_ncc: one pass over raw moments, every first-order Hessian type) and
 * MI (four pixel-level launches, five for SumOfStd; every first-order Jacobian / Hessian type).  Anything outside
 * returns MTFHIP_ERR_NOT_IMPLEMENTED and belongs to the per-function entry points above -- which defer and fuse the same
 * way internally when the call sequence is one of the search methods' ("Deferred fusion").
 * sec_ord_hess = 1 is honoured by init_template / iterate / track for SSD (SSDBase.cc:313-415), NCC (NCC.cc:391-410) and MI
 * (MI.cc:659-735, 8 bins): one more pixel pass per iteration accumulates
 * sum_p df_dI[p] d2I_dp2[:, p] -- MI's self types: its self gradient factor as the weight -- without building the S^2 x N matrices,
 * and the device loop solves the then indefinite system with pivoting.  NCC's self Hessian types (no second-order cmptSelfHessian
 * in the reference, AppearanceModel.h:188-191), MI with another bin count and the multi-channel models return MTFHIP_ERR_NOT_IMPLEMENTED for it.
 * init_template = the body of nt::{ESM,FCLK,ICLK}::initialize after ssm->initialize
 * (NT/ESM.cc:110-146, NT/FCLK.cc:102-169, NT/ICLK.cc:71-128): I0, dI0_dx, J0 and the constant
 * Hessian from the current image at the current points.  ICLK with a constant Hessian over SSD / NCC, <= 1024 pixels, at the identity
 * warp: one launch (k_template_init), ASYNCHRONOUS -- the call returns with the kernel enqueued; its small results reach the host
 * mirrors when a later call asks for them (see mtfhip_image_borrow for what that means for a borrowed image). */
int mtfhip_batch_init_template(mtfhip_batch *b, const mtfhip_sm_desc *sm);
/* setRegion of the search method between frames (NT/ESM.cc:148-168, NT/FCLK.cc:360-376, NT/ICLK.cc:131-157): SSM reset to
 * the new corners, template kept; ESM (and FCLK / InitialSelf) refresh init_pix_jacobian on the new grid and the constant
 * Hessian where the Hessian type uses it; ICLK keeps its template Jacobian.  (mtfhip_ssm_set_corners alone = SSM reset only.) */
int mtfhip_batch_set_region(mtfhip_batch *b, const double *corners, const mtfhip_sm_desc *sm);
/* One iteration's device work (A2..A9 of SURVEY.md section 8a) at the current warp:
 * f (B), g (B x S) and H (B x S x S col-major) exactly as the SM holds them before LM damping
 * and the S x S solve, which stay with the caller as in the reference. */
int mtfhip_batch_iterate(mtfhip_batch *b, const mtfhip_sm_desc *sm, double *f, double *g, double *H);
/* The whole update() loop on device (solve, compositional update and the corner-change
 * convergence test included) without host round trips; returns per-target iteration counts
 * and final corners.  SM/src/NT/{ESM,FCLK,ICLK}.cc update(). */
int mtfhip_batch_track(mtfhip_batch *b, const mtfhip_sm_desc *sm, int *n_iters /* B */, double *corners /* B x 8 */);
/* setRegion(region_corners) followed by update() in one call -- the pair GridTracker::update issues per patch tracker
 * (SM/src/GridTracker.cc:345-363) and PyramidalTracker per level (SM/src/PyramidalTracker.cc:70-96).  Same results as
 * mtfhip_batch_set_region + mtfhip_batch_track; one staged upload per frame where the search method keeps its template Jacobian. */
int mtfhip_batch_track_region(mtfhip_batch *b, const mtfhip_sm_desc *sm, const double *region_corners /* B x 8 */,
	int *n_iters /* B */, double *corners /* B x 8 */);
/* GridTracker::update's patch half (SM/src/GridTracker.cc:345-363) in one call: mtfhip_batch_track_region with regions / corners as
 * row-major 2 x 4 arrays per patch (x row, y row) and the patch centroids (utils::getCentroid, miscUtils.h:473-480) */
int mtfhip_grid_update(mtfhip_batch *b, const mtfhip_sm_desc *sm, const double *regions_2x4 /* B x 2 x 4 */, int *n_iters /* B, or NULL */,
	double *corners_2x4 /* B x 2 x 4, or NULL */, double *centroids /* B x 2, or NULL */);
/* GridTrackerParams (SM/include/mtf/SM/GridTracker.h:8-60, SM/src/GridTracker.cc:20-94): the fields that shape the frame.  Class
 * defaults 10 x 10 patches of 10 x 10, reset_at_each_frame 1, dyn_patch_size 0, patch_centroid_inside 1
 * (Config/parameters.h:505-510; shipped Config/modules.cfg:75-80: grid_res 10, grid_patch_size 25). */
typedef struct mtfhip_grid_desc {
	int grid_size_x, grid_size_y;
	int patch_size_x, patch_size_y;
	int reset_at_each_frame;      /* 0: patch trackers run on; 1: re-initialised on the new grid every frame; other: setRegion only (GridTracker.cc:136,273) */
	int dyn_patch_size;           /* 1: a patch is the quadrilateral of its four surrounding grid points */
	int patch_centroid_inside;    /* 1: patch_size rectangles centred on the centroid of the four surrounding grid points */
} mtfhip_grid_desc;
/* GridTrackerParams::updateRes (GridTracker.cc:86-94): the sampling resolution of the grid SSM -- grid_size + 1 when
 * dyn_patch_size || patch_centroid_inside, else grid_size */
int mtfhip_grid_res(const mtfhip_grid_desc *g, int *resx, int *resy);
/* GridTracker::resetTrackers' geometry (GridTracker.cc:345-380) for the grid SSM laid over region_corners (CornersT: x, y per
 * corner, TL TR BR BL): grid_pts = ssm.getPts() after ssm.setCorners(region) -- the resx x resy unit-square grid through the
 * 4-corner DLT homography (ProjectiveBase::getPtsFromCorners ProjectiveBase.cc:20-25; Homography and Affine both, with
 * normalized_init = 0), row-major (y outer) -- and the corners resetTrackers hands patch tracker k = row * grid_size_x + col:
 * the four surrounding points (_linear_idx, :139-146, :357-367), then unless dyn_patch_size the patch_size rectangle
 * (utils::Corners(cv::Rect_<double>) miscUtils.h:42-52) centred on their centroid (patch_centroid_inside) or on grid point k.
 * Host arithmetic only: no device, no batch.  grid_pts (2 x resx*resy, x, y interleaved) may be NULL. */
int mtfhip_grid_layout(const mtfhip_grid_desc *g, const double *region_corners /* 8 */, double *grid_pts /* or NULL */,
	double *patch_corners /* grid_size_x * grid_size_y x 8 */);
/* One frame of GridTracker's patch half for a batch of grid_size_x * grid_size_y patch trackers, CornersT layout throughout:
 * with region_corners the patches are first laid over that region (mtfhip_grid_layout) and reset -- setRegion, in the same launch
 * as the update where the search method allows it -- then every patch tracker runs its update(); centroids are utils::getCentroid
 * into cv::Point2f (miscUtils.h:472-480: rounded to float), the points GridTracker::update hands to ssm.estimateWarpFromPts
 * (GridTracker.cc:256-270).  region_corners NULL: update only. */
int mtfhip_grid_frame(mtfhip_batch *b, const mtfhip_sm_desc *sm, const mtfhip_grid_desc *g, const double *region_corners /* 8 or NULL */,
	int *n_iters /* B or NULL */, double *corners /* B x 8 or NULL */, float *centroids /* B x 2 or NULL */);
/* resetTrackers(reinit) (GridTracker.cc:345-392) for the same batch: the patches laid over region_corners, then every patch tracker
 * initialize()d on the current image (reinit != 0: mtfhip_ssm_set_corners + mtfhip_batch_init_template) or setRegion()ed
 * (mtfhip_batch_set_region).  Nothing is waited for; patch_corners / prev_pts (the centroids of the regions the trackers report
 * afterwards = of the patches) may be NULL. */
int mtfhip_grid_reset(mtfhip_batch *b, const mtfhip_sm_desc *sm, const mtfhip_grid_desc *g, const double *region_corners /* 8 */, int reinit,
	double *patch_corners /* B x 8 or NULL */, float *prev_pts /* B x 2 or NULL */);
/* ---- GridTracker's forward-backward error estimation (SM/src/GridTracker.cc:186-190 enable, :241-243 / :266 prev_img, :294-343
 * backwardEstimation; shipped Config/modules.cfg:81-82: grid_fb_err_thresh 2, grid_fb_reinit 1) ---- */
typedef struct mtfhip_grid_fb_desc {
	double fb_err_thresh;   /* > 0: a patch whose backward track misses its starting point by more than this (squared distance of the
	                           cv::Point2f centroids, :309-313) is left out of the fit */
	int fb_reinit;          /* 1: every patch tracker is re-initialised at its tracked location before it runs backwards (:297-299) */
	int n_model_pts;        /* est_params.n_model_pts (SSM/src/SSMEstimatorParams.cc:63; shipped Config/modules.cfg:39: 4): the surviving
	                           set is filled up to it in tracker order (:321-332) */
} mtfhip_grid_fb_desc;
/* prev_img = curr_img.clone() (GridTracker.cc:241-243, :266): the current image of the context becomes its previous image.  An image
 * the context owns (mtfhip_image_upload / _preprocess) is kept without a copy -- the next frame goes to the other of two device
 * buffers --, a borrowed one is copied device-to-device on the context's stream. */
int mtfhip_image_keep_prev(mtfhip_ctx *ctx);
int mtfhip_image_has_prev(mtfhip_ctx *ctx);
/* setImage(prev_img) / setImage(curr_img) of every tracker on the context (GridTracker.cc:300, :304): current and previous image change
 * places; a second call changes them back */
int mtfhip_image_swap_prev(mtfhip_ctx *ctx);
/* The patch loop of backwardEstimation (:295-306) for the whole batch: location = getRegion(); initialize(location) on the current
 * frame when fb_reinit; setImage(prev_img); update(); fb_prev_pts = getCentroid(getRegion()) (cv::Point2f: rounded to float);
 * setImage(curr_img); setRegion(location).  Needs mtfhip_image_keep_prev.  n_iters (B), fb_corners (B x 8: where the patch trackers
 * arrived on the previous frame) and fb_prev_pts (B x 2) may be NULL. */
int mtfhip_grid_backward(mtfhip_batch *b, const mtfhip_sm_desc *sm, const mtfhip_grid_desc *g, const mtfhip_grid_fb_desc *fb, int *n_iters,
	double *fb_corners, float *fb_prev_pts);
/* The mask half of backwardEstimation (:307-332): fb_err_mask[k] = the squared distance fb_prev_pts[k] - prev_pts[k] (float differences,
 * double squares) is not above fb_err_thresh; the surviving (prev, curr) pairs in tracker order, filled up to n_model_pts with the
 * first rejected ones (their mask set) when fewer survive -- what estimateWarpFromPts is then handed (:334-335).  Host arithmetic
 * only.  prev_masked / curr_masked (n x 2 each) may be NULL. */
int mtfhip_grid_fb_mask(int n, const float *prev_pts, const float *curr_pts, const float *fb_prev_pts, const mtfhip_grid_fb_desc *fb,
	unsigned char *fb_err_mask, float *prev_masked, float *curr_masked, int *n_masked);
/* GridTracker::update's patch loop with the estimation on (:254-266): mtfhip_grid_frame (region_corners as there), then
 * mtfhip_grid_backward and mtfhip_grid_fb_mask against prev_pts (the centroids the last reset / frame left, B x 2).  With
 * g->reset_at_each_frame != 0 the caller's resetTrackers follows (:273-274: mtfhip_grid_reset, or the region of the next frame) and
 * replaces whatever setRegion(tracker_location) would leave, so that last step of the backward pass is left out.
 * The reset-every-frame configuration (g->reset_at_each_frame == 1, no region -- the shipped one, with fb->fb_reinit; also reset_at_each_frame == 0
 * without fb_reinit, setRegion(tracker_location) following as one more call; tolerance mode, ICLK with a
 * constant Hessian over SSD / NCC, <= 1024 pixels; with fb_reinit an affine patch SSM) is ONE launch (k_grid_fb, kernels_grid_fb.hip): a patch's
 * update(), its initialize(tracker_location) when fb_reinit, and its update() on the previous frame run back to back in its workgroup,
 * bit-identical to the launch-by-launch form (MTFHIP_GRID_FB_FUSED=0).  The patch trackers are then left as the FORWARD pass left them (state,
 * corners, template): the caller's mtfhip_grid_reset(reinit) re-initialises them. */
int mtfhip_grid_frame_fb(mtfhip_batch *b, const mtfhip_sm_desc *sm, const mtfhip_grid_desc *g, const mtfhip_grid_fb_desc *fb,
	const double *region_corners /* 8 or NULL */, const float *prev_pts /* B x 2 */, int *n_iters /* B or NULL */, double *corners /* B x 8 or NULL */,
	float *centroids /* B x 2 or NULL */, float *fb_prev_pts /* B x 2 */, unsigned char *fb_err_mask /* B */, float *prev_masked /* B x 2 or NULL */,
	float *curr_masked /* B x 2 or NULL */, int *n_masked);
/* Debug trace of the loop above: with max_passes > 0 every pass also records what it solved, per target
 * [max_passes][96]: H (64, row-major 8 x 8, before Levenberg-Marquardt damping) | g (8) | the state update applied (8) | the
 * corners it produced (8) | f | pass | LM undo | LM damping | 1 when H was recorded (the one-launch grid loop uses the
 * constant template Hessian and records 0).  This is how the parity tests compare the device-side loop with the CPU trackers
 * iteration by iteration; 0 switches it off (the default).  _read copies B x max_passes x 96 doubles. */
int mtfhip_batch_track_trace(mtfhip_batch *b, int max_passes);
int mtfhip_batch_track_trace_read(mtfhip_batch *b, double *dst);
/* how many targets one launch of the loop above covers (all of them, or an Infinity-Cache sized chunk) */
int mtfhip_batch_track_targets_per_launch(mtfhip_batch *b, const mtfhip_sm_desc *sm);
/* how many queues (HIP streams) the loop above keeps busy with independent chunks of the targets: 2 for the launches that write the
 * interface arrays (one chunk's solve + update runs under the other's pixel pass), else 1; MTFHIP_TRACK_STREAMS=1 forces 1 */
int mtfhip_batch_track_queues(mtfhip_batch *b, const mtfhip_sm_desc *sm);

/* ---- candidate scoring (PF / NN batch axis): target 0's template, C warps ----
 * per candidate: setState -> updatePixVals -> updateSimilarity(false) -> getLikelihood
 * (SM/src/PF.cc:247-262), SSD (SSD.h:41-43) and NCC (NCC.cc:50-53, 124-161; the candidate's scalars from its raw
 * moments).  The *_dev form takes and fills device pointers (for RCCL). */
int mtfhip_score_candidates(mtfhip_batch *b, const double *states /* C x S */, int n_candidates,
	double *likelihoods /* C or NULL */, double *similarities /* C or NULL */);
int mtfhip_score_candidates_dev(mtfhip_batch *b, const double *dev_states, int n_candidates,
	double *dev_likelihoods, double *dev_similarities);

/* ---- the particle filter on the device: nt::PF (SM/src/NT/PF.cc) over the batch's single target ----
 * One iteration of update()'s loop is three launches: proposal (the SSM's stochastic sampler and dynamic models:
 * ProjectiveBase.cc:163-317, Homography.cc:899-942, Affine.cc:464-553) + scoring + particle weight; chunked cumulative weights;
 * multinomial resampling + estimate.  Only the estimate (state, corners, best weight) crosses PCIe.  Enum values are the
 * reference's (SM/include/mtf/SM/PFParams.h:10-33).
 * Affine follows Affine.cc: compositional updates with point based sampling (pt_based_sampling 1 / 2: three canonical points
 * disturbed, affine map of the three pairs) or, for AutoRegression1, the geometric perturbation (geomToState of six draws);
 * the combinations the reference itself throws for (additive + point based, compositional RandomWalk + geometric) return
 * MTFHIP_ERR_NOT_IMPLEMENTED with its message, and so does additive + geometric, which needs Affine::stateToGeom -- a 2 x 2
 * JacobiSVD whose sign / ordering conventions decide its branches and cannot be reproduced without Eigen.
 * Several sampler distributions with adaptive weights (mtfhip_pf_set_distributions) and adaptive resampling run on the device too: the
 * cumulative-weight launch also takes the per-distribution weight sums and sum w^2, its last workgroup derives the next iteration's
 * distribution weights and the verdict "this iteration resamples", the selection pass obeys it.  jacobian_as_sigma (PF.cc:58-64,
 * 156-165, 214-227) is host logic over entry points of this header: mtf_amd/sm.py ParticleFilter, mtf_amd/host/PF.cpp.
 * pix_sigma (PF.cc:142-149) likewise: mtfhip_ssm_estimate_state_sigma gives the sigma rows, the host installs them at initialize(). */
typedef struct mtfhip_pf mtfhip_pf;
typedef struct mtfhip_comm mtfhip_comm;
typedef struct mtfhip_pf_desc {
	int n_particles, max_iters;
	double epsilon;
	int dynamic_model;        /* 0 RandomWalk, 1 AutoRegression1 */
	int update_type;          /* 0 Additive, 1 Compositional */
	int likelihood_func;      /* 0 AM, 1 Gaussian, 2 Reciprocal */
	int resampling_type;      /* 0 None, 1 BinaryMultinomial, 2 LinearMultinomial, 3 Residual (PF.cc:538-582: particle_wts are normalised in place) */
	int mean_type;            /* 0 None (highest weight), 1 SSM (mean state), 2 Corners (mean corners, then setCorners) */
	int corner_based_sampling;/* HomographyParams::corner_based_sampling (on by default, parameters.h:262) */
	int reset_to_mean;
	double measurement_sigma; /* Gaussian likelihood (PF.cc:69-70, 352-354) */
	double ar_coeff;          /* a of the AutoRegression1 models (StateSpaceModel.h:311-318: 0.5) */
	double ssm_sigma[8], ssm_mean[8]; /* the sampler's normal distributions (ProjectiveBase::initializeSampler) */
	unsigned long long seed;  /* device generator (Philox4x32-10), used when no draws are handed in */
	int pt_based_sampling;    /* AffineParams::pt_based_sampling (0 geometric, 1, 2: Affine.cc:464-503; default 0, parameters.h:254) */
	/* (appended in r03; the shipped Config/modules.cfg:157-176 uses all three) */
	double adaptive_resampling_thresh; /* PFParams::adaptive_resampling_thresh in (0, 1]: resample only when the effective particle count
	                                      1 / sum (w / sum w)^2 is <= thresh * n (PF.cc:114-118, 381-390); 0: every iteration */
	int update_distr_wts;     /* PFParams::update_distr_wts: the weights of several sampler distributions follow the average particle weight
	                             each produced (PF.cc:345-369); REQUIRED by mtfhip_pf_set_distributions with more than one distribution: without it the
	                             reference zeroes the weights and draws from an all-zero discrete distribution (NT/PF.cc:241-257), which every front end
	                             of this library refuses (MTFHIP_ERR_NOT_IMPLEMENTED; mtf::hip::PF, nt::PF of the harness, the Python wrapper, the oracle) */
	double min_distr_wt;      /* PFParams::min_distr_wt: floor of a distribution's weight */
} mtfhip_pf_desc;
int mtfhip_pf_create(mtfhip_batch *b, const mtfhip_pf_desc *desc, mtfhip_pf **out);
void mtfhip_pf_destroy(mtfhip_pf *pf);   /* before the batch it was created on */
/* nt::PF::initialize after ssm->initialize / am->initializePixVals / am->initializeSimilarity (PF.cc:136-183) */
int mtfhip_pf_initialize(mtfhip_pf *pf);
int mtfhip_pf_set_region(mtfhip_pf *pf, const double *corners /* 8 */);          /* PF.cc:616-620 */
int mtfhip_pf_set_sampler(mtfhip_pf *pf, const double *sigma, const double *mean);
/* n_distr (1 .. 8) sampler distributions, rows of 8 (PFParams::processDistributions, PF.cc:55-56): see mtfhip_pf_desc.update_distr_wts */
int mtfhip_pf_set_distributions(mtfhip_pf *pf, int n_distr, const double *sigma /* n_distr x 8 */, const double *mean /* n_distr x 8 */);
/* the distribution draws of the NEXT iteration supplied by the caller (n uniforms in (0, 1]); NULL: the device generator */
int mtfhip_pf_set_distr_draws(mtfhip_pf *pf, const double *uniforms);
/* the distribution weights the next iteration draws from, the particles' distribution ids (or NULL), whether the last iteration resampled.
 * The ids belong to the particles' CURRENT proposals: with look-ahead proposals (the default with the device generator: the selection
 * pass of iteration t already draws iteration t + 1) those are the PENDING iteration's draws, not the ones the last weights were
 * produced by; MTFHIP_PF_LOOKAHEAD=0, or draws handed in by the caller, keep them those of the last iteration. */
int mtfhip_pf_get_distributions(mtfhip_pf *pf, double *distr_wts /* n_distr */, int *distr_ids /* n or NULL */, int *resampled); /* ProjectiveBase.cc:208-215 */
/* one iteration of update()'s loop (PF.cc:260-447); normals n x nz (nz = 10 with corner based homography sampling, 6 / 8 / 6 for
 * Affine point based 1 / 2 / geometric, else S) and uniforms n: host arrays, or NULL for the device generator; update_norm =
 * squared corner change of the estimate */
int mtfhip_pf_iteration(mtfhip_pf *pf, const double *normals, const double *uniforms, double *update_norm);
/* PF.cc:207-447.  epsilon < 0 (no convergence test) and mean_type != Corners: the max_iters iterations are enqueued back to
 * back and only the last estimate is read back */
int mtfhip_pf_update(mtfhip_pf *pf, int *n_iters);
int mtfhip_pf_get_particles(mtfhip_pf *pf, double *states /* n x S */, double *ars, double *wts /* n */, int *resample_ids /* n */);
int mtfhip_pf_set_particles(mtfhip_pf *pf, const double *states, const double *ars /* or NULL: zeros */);
double mtfhip_pf_max_similarity(const mtfhip_pf *pf);
int mtfhip_pf_set_max_similarity(mtfhip_pf *pf, double max_similarity);   /* PF.cc:443-446: after am->updateModel (enable_learning) */
/* ---- the collective of the sharded candidate axis: RCCL directly (no torch), bound with dlopen at first use ----
 * rank 0 obtains the 128-byte id and hands it to the other ranks by whatever channel the host program has (MPI, a file, a
 * socket, torch.distributed); every rank then creates its communicator.  SM/src/PF.cc:262-277 is the weights vector this
 * replaces once particles are sharded: rank r scores the contiguous block [r m, (r + 1) m), m = ceil(n / world) (the last
 * blocks may be short or empty: mtfhip_pf_shard_bounds), ONE all-gather of m weights per rank -- in place, the blocks already
 * sit at their global positions -- puts every weight on every rank, and resampling runs redundantly on identical data. */
int mtfhip_comm_unique_id(void *id128);
int mtfhip_comm_create(const void *id128 /* NULL allowed for world 1 */, int rank, int world, int device, mtfhip_comm **out);
/* `world` ranks as threads of ONE process on ONE device (out[r] = rank r's communicator, one host thread per rank): the
 * all-gather is a rendezvous of the threads plus device copies.  It exists so that the sharded code path can be executed and
 * compared with the unsharded filter where only one GPU is available; everything but the exchange itself is the RCCL path. */
int mtfhip_comm_create_loopback(int world, int device, mtfhip_comm **out /* [world] */);
void mtfhip_comm_destroy(mtfhip_comm *comm);
int mtfhip_comm_rank(const mtfhip_comm *comm);
int mtfhip_comm_world(const mtfhip_comm *comm);
/* in place when dev_send == dev_recv + rank * count_per_rank (ncclAllGather's in-place form) */
int mtfhip_allgather_scores(mtfhip_comm *comm, const double *dev_send, int count_per_rank, double *dev_recv /* world x count */, void *hip_stream);
int mtfhip_pf_set_comm(mtfhip_pf *pf, mtfhip_comm *comm);   /* shard the filter's scoring over the communicator's ranks */
/* How the sharded filter's weights travel (collective call: every rank, after mtfhip_pf_set_comm).
 * COLLECTIVE, the default: one in-place all-gather per iteration (RCCL), enqueued between the scoring and the scan.
 * PEER: no collective and no extra launch -- the scoring kernel stores each weight into every rank's mailbox (fine-grained device
 * memory mapped into the peers with hipIpcOpenMemHandle at this call; loopback ranks share an address space) and adds to an arrival
 * counter per rank; the scan waits for the counters (bounded: a rank that never arrives becomes MTFHIP_ERR_HIP at the next estimate).
 * Two mailbox vectors alternate, so a rank may run one iteration ahead of the slowest.  Up to 8 ranks (one node).  The result is
 * the same bits as with the collective: the same weights reach the same places.  Measured on one GPU only (loopback ranks); between
 * GPUs it is an opt-in until a multi-GPU node has run it -- DESIGN.md section 6. */
enum { MTFHIP_PF_EXCHANGE_COLLECTIVE = 0, MTFHIP_PF_EXCHANGE_PEER = 1 };
int mtfhip_pf_set_exchange(mtfhip_pf *pf, int mode);
/* The two halves of the PEER set-up for a host program that moves the handles itself (as it moves mtfhip_comm_unique_id's bytes):
 * export -> this rank's mailbox as a 64-byte hipIpc handle; connect <- the handles of all ranks, rank-major (the own one is skipped).
 * connect also compares the seeds the ranks left in their mailboxes (identical proposals need ONE seed).
 * mtfhip_comm_create_detached: a communicator that is rank and world only -- no RCCL behind it, no all-gather; a filter sharded over
 * it exchanges through export / connect + peer stores.  (Two processes on ONE GPU can run it, which RCCL refuses: the cross-process
 * half of the exchange -- IPC mapping, system-scope counters -- is tested that way, tests/test_gpu_trackers.py.) */
int mtfhip_pf_exchange_export(mtfhip_pf *pf, void *handle64);
int mtfhip_pf_exchange_connect(mtfhip_pf *pf, const void *handles /* world x 64 bytes */);
int mtfhip_comm_create_detached(int rank, int world, int device, mtfhip_comm **out);
/* the partition mtfhip_pf_set_comm uses (host arithmetic, no device): rank's block [lo, lo + count), per_rank = ceil(n / world) */
int mtfhip_pf_shard_bounds(int n_particles, int world, int rank, int *lo, int *count, int *per_rank);

/* ---- NN-SM dataset generation (the second batch axis of the path; the search itself stays with FLANN) ----
 * Row c of the C x N feature matrix = updateDistFeat() of the patch sampled under state c:
 * setState / compositionalUpdate -> updatePixVals -> updateDistFeat (SM/src/NT/NN.cc:131-191;
 * SSD feature = It, AM/include/mtf/AM/SSDBase.h:116-125; NCC feature = (It - mean)/||It - mean||, AM/src/NCC.cc:530-537) */
int mtfhip_sample_candidates(mtfhip_batch *b, const double *states /* C x S */, int n_samples, double *features /* C x N */);
int mtfhip_sample_candidates_dev(mtfhip_batch *b, const double *dev_states, int n_samples, double *dev_features);

/* NN::generateDataset itself (SM/src/NT/NN.cc:131-191, compositional update) in one launch: per sample the perturbation p -- row c of
 * perturbations_in, or (NULL) drawn on the device as ProjectiveBase::generatePerturbation does (ProjectiveBase.cc:283-288: component k ~
 * N(mean[k], sigma[k]); Philox4x32-10 + Box-Muller keyed by (seed, c): a pure function of the sample's index) --, the SSM moved by
 * invertState(p) (Homography.cc:109-114, Affine.cc:145-150) through compositionalUpdate (Homography.cc:73-92), updatePixVals, and
 * updateDistFeat into row c: SSD the patch (SSDBase.h:116-125), NCC centred and of unit norm (NCC.cc:530-537), MI the 5 x N matrix
 * floor(It) | four cubic B-spline weights (MI.cc:736-747); multi-channel models: rows of N = n_pix x n_channels entries.  The batch holds
 * ONE target; its SSM is left where it was (the reference's compositionalUpdate(p) back).  additive_update (NNParams): not implemented. */
typedef struct mtfhip_nn_desc {
	int n_samples;
	int additive_update;      /* NNParams::additive_update (SM/src/NT/NNParams.cc): must be 0 */
	double sigma[8], mean[8]; /* state_sigma[0] / state_mean[0] of NN::initialize (NT/NN.cc:56-84); several distributions: one call per distribution
	                             with its rows (row_lo / row_count of the _dev form) */
	unsigned long long seed;
} mtfhip_nn_desc;
int mtfhip_nn_feature_size(mtfhip_batch *b, int *feat_size);   /* am->getDistFeatSize(): N, MI 5 N */
/* host form: perturbations_in (n_samples x S, or NULL: drawn), perturbations_out (n_samples x S, or NULL), features (n_samples x feat_size) */
int mtfhip_nn_dataset(mtfhip_batch *b, const mtfhip_nn_desc *d, const double *perturbations_in, double *perturbations_out, double *features);
/* device form (the dataset stays in HBM; RCCL all-gather of row blocks): rows [row_lo, row_lo + row_count) of the n_samples x feat_size
 * matrix into dev_features[row_count][feat_size]; dev_perturbations_in / _out are indexed by the GLOBAL sample index (n_samples x S) */
int mtfhip_nn_dataset_dev(mtfhip_batch *b, const mtfhip_nn_desc *d, const double *dev_perturbations_in, double *dev_perturbations_out,
	double *dev_features, int row_lo, int row_count);

/* ---- measurement hooks ---- */
/* average duration in milliseconds of the launches of the named kernel family since the last
 * reset, measured with hipEvents on the context's stream (0 if timing is disabled).
 * mtfhip_timing_enable(ctx, n): n = 0 off, 1 every launch, n > 1 every n-th launch of a family (sampling keeps the
 * event overhead, ~5 % of a 100 us step at n = 1, out of the timed region) */
int mtfhip_timing_enable(mtfhip_ctx *ctx, int on);
int mtfhip_timing_reset(mtfhip_ctx *ctx);
int mtfhip_timing_get(mtfhip_ctx *ctx, const char *kernel_family, double *avg_ms, int *n_launches);
/* time (ms) during which at least one launch of the family was executing since the last reset -- the union of the timed
 * launches' intervals -- and their number.  The device-side loop keeps two chunks of targets in flight on two queues
 * (mtfhip_batch_track), so launches of one kernel overlap there: bytes moved / busy time is the bandwidth the kernel's launches
 * reached together, bytes per launch / average duration what one of them saw. */
int mtfhip_timing_get_busy(mtfhip_ctx *ctx, const char *kernel_family, double *busy_ms, int *n_launches);
/* 1 when the single-target launches carry the warp inside the kernel arguments (the library probes once per process that the
 * runtime lays the kernel-argument segment out the way the kernels read it; MTFHIP_INLINE_WARP=0 or a failed probe: 0, and
 * the warp is uploaded in front of every launch instead -- same results, 4-6 us more per iteration of a single target) */
int mtfhip_batch_inline_warp(const mtfhip_batch *b);

#ifdef __cplusplus
}
#endif
#endif
