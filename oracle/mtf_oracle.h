/*
 * mtf_oracle.h -- C API of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain FP64 CPU restatement of the
 * Lucas-Kanade hot path of abhineet123/MTF (file:line citations are in
 * mtf_oracle.cpp next to each function).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (libmtfhip.so and
 * everything under mtf_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors for this path
 * and cannot be built in this image (Eigen, OpenCV and Boost are absent), so
 * this restatement is pinned only by (i) line-by-line review against the cited
 * files, (ii) an independent NumPy float64 re-derivation (oracle/numpy_ref.py,
 * fixtures under tests/golden/) and (iii) the reference's own Diagnostics
 * relations (Hessian equalities at identity, analytic-vs-numeric Jacobians).
 *
 * Layout conventions follow the reference's Eigen typedefs
 * (Macros/include/mtf/Macros/common.h:190-258), all column-major:
 *   pts        2 x N   -> x,y interleaved per point
 *   grad_pts   8 x N   -> 8 doubles interleaved per point
 *   pix_grad   N x 2   -> N Ix then N Iy
 *   J          N x S   -> S contiguous columns of N
 *   H          S x S   -> column-major
 *   corners    2 x 4   -> x,y interleaved, TL,TR,BR,BL
 *   image      H x W float32 row-major, contiguous
 */
#ifndef MTF_ORACLE_H
#define MTF_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { MTFO_AM_SSD = 0, MTFO_AM_NCC = 1, MTFO_AM_MI = 2 };
enum { MTFO_SSM_HOMOGRAPHY = 0, MTFO_SSM_AFFINE = 1 };
enum { MTFO_SM_ESM = 0, MTFO_SM_FCLK = 1, MTFO_SM_ICLK = 2 };

/* ---- L1 pixel utilities (Utilities/imgUtils) ---- */
double mtfo_get_pix_val(const float *img, int h, int w, double x, double y);
void mtfo_get_pix_vals(double *out, const float *img, int h, int w,
	const double *pts, int n, double norm_mult, double norm_add);
void mtfo_get_img_grad(double *grad, const float *img, int h, int w,
	const double *pts, double grad_eps, int n, double pix_mult);
void mtfo_get_warped_img_grad(double *grad, const float *img, int h, int w,
	const double *grad_pts, double grad_eps, int n, double pix_mult);

/* hess: 4 x N col-major (xx, xy, yx, yy per pixel); hess_pts: 16 x N */
void mtfo_get_img_hess(double *hess, const float *img, int h, int w,
	const double *pts, double hess_eps, int n, double pix_mult);
void mtfo_get_warped_img_hess(double *hess, const float *img, int h, int w,
	const double *pts, const double *hess_pts, double hess_eps, int n, double pix_mult);

/* ---- small host math the SMs / SSMs use (Eigen in the reference) ---- */
void mtfo_homography_dlt(const double *in_corners, const double *out_corners,
	double *warp_rowmajor9);
/* x = A^-1 b through a column-pivoted Householder QR; A is n x n column-major */
void mtfo_colpiv_qr_solve(int n, const double *A, const double *b, double *x);
void mtfo_norm_unit_square_pts(double *pts, double *corners, int resx, int resy,
	double min_x, double min_y, double max_x, double max_y);

/* ---- state space model ---- */
typedef struct mtfo_ssm mtfo_ssm;
mtfo_ssm *mtfo_ssm_create(int kind, int resx, int resy);
void mtfo_ssm_destroy(mtfo_ssm *s);
int mtfo_ssm_state_size(const mtfo_ssm *s);
int mtfo_ssm_n_pts(const mtfo_ssm *s);
void mtfo_ssm_set_corners(mtfo_ssm *s, const double *corners);
void mtfo_ssm_set_state(mtfo_ssm *s, const double *state);
void mtfo_ssm_estimate_state_sigma(mtfo_ssm *s, double pix_sigma, double *state_sigma /* S */);
void mtfo_ssm_compositional_update(mtfo_ssm *s, const double *state_update);
void mtfo_ssm_invert_state(mtfo_ssm *s, double *inv_state, const double *state);
void mtfo_ssm_update_grad_pts(mtfo_ssm *s, double grad_eps);
void mtfo_ssm_cmpt_init_pix_jacobian(mtfo_ssm *s, double *J, const double *pix_grad);
void mtfo_ssm_cmpt_pix_jacobian(mtfo_ssm *s, double *J, const double *pix_grad);
void mtfo_ssm_cmpt_warped_pix_jacobian(mtfo_ssm *s, double *J, const double *pix_grad);
void mtfo_ssm_cmpt_approx_pix_jacobian(mtfo_ssm *s, double *J, const double *pix_grad);
/* second order (sec_ord_hess): pix_hess 4 x N, pix_grad N x 2 -> d2I_dp2 S^2 x N (one col-major S x S block per pixel);
 * return -2 where the reference's SSM leaves the virtual unimplemented (Affine: Pix, Approx) */
void mtfo_ssm_update_hess_pts(mtfo_ssm *s, double hess_eps);
int mtfo_ssm_cmpt_init_pix_hessian(mtfo_ssm *s, double *d2, const double *pix_hess, const double *pix_grad);
int mtfo_ssm_cmpt_pix_hessian(mtfo_ssm *s, double *d2, const double *pix_hess, const double *pix_grad);
int mtfo_ssm_cmpt_warped_pix_hessian(mtfo_ssm *s, double *d2, const double *pix_hess, const double *pix_grad);
int mtfo_ssm_cmpt_approx_pix_hessian(mtfo_ssm *s, double *d2, const double *pix_hess, const double *pix_grad);
void mtfo_ssm_apply_warp_to_corners(mtfo_ssm *s, double *out_corners,
	const double *in_corners, const double *state);
void mtfo_ssm_apply_warp_to_pts(mtfo_ssm *s, double *out_pts, const double *in_pts, int n_pts, const double *state);
void mtfo_ssm_compose_warps(mtfo_ssm *s, double *composed, const double *state_1, const double *state_2);
void mtfo_ssm_estimate_warp_from_corners(mtfo_ssm *s, double *state_update, const double *in_corners, const double *out_corners);
void mtfo_ssm_additive_update(mtfo_ssm *s, const double *state_update);
void mtfo_ssm_compositional_random_walk(mtfo_ssm *s, double *perturbed_state,
	const double *base_state, const double *perturbation);
/* what: 0 curr_pts(2N) 1 init_pts(2N) 2 curr_corners(8) 3 init_corners(8)
 *       4 curr_state(S) 5 curr_warp(9,row-major) 6 grad_pts(8N)
 *       7 curr_pts_hm(3N) 8 init_pts_hm(3N) 9 hess_pts(16N) */
void mtfo_ssm_get(const mtfo_ssm *s, int what, double *dst);

/* ---- appearance model ---- */
typedef struct mtfo_am mtfo_am;
mtfo_am *mtfo_am_create(int kind, int resx, int resy, double grad_eps,
	double likelihood_alpha, int mi_n_bins, double mi_pre_seed, int mi_pou);
void mtfo_am_destroy(mtfo_am *a);
int mtfo_am_n_pix(const mtfo_am *a);
void mtfo_am_set_curr_img(mtfo_am *a, const float *img, int h, int w);
/* multi-channel variants (MCSSD / MCNCC / MCMI = the same AM with n_channels 3, AM/src/MCSSD.cc etc.): the image is then
 * H x W x C float32 interleaved (CV_32FC3), every per-pixel AM vector has n_pix * C entries interleaved per pixel, and the
 * SSM's pixel Jacobians / Hessians have one row / block per (pixel, channel) (StateSpaceModel::initialize(corners, n_channels)).
 * Call both before initialize. */
void mtfo_am_set_channels(mtfo_am *a, int n_channels);
void mtfo_ssm_set_channels(mtfo_ssm *s, int n_channels);
int mtfo_am_patch_size(const mtfo_am *a);
void mtfo_am_initialize_pix_vals(mtfo_am *a, const double *pts);
void mtfo_am_update_pix_vals(mtfo_am *a, const double *pts);
int mtfo_am_update_model(mtfo_am *a, const double *pts, double learning_rate); /* -1: FunctonNotImplemented (MI) */
void mtfo_am_initialize_pix_grad_pts(mtfo_am *a, const double *pts);
void mtfo_am_initialize_pix_grad_warped(mtfo_am *a, const double *grad_pts);
void mtfo_am_update_pix_grad_pts(mtfo_am *a, const double *pts);
void mtfo_am_update_pix_grad_warped(mtfo_am *a, const double *grad_pts);
void mtfo_am_set_hess_eps(mtfo_am *a, double hess_eps); /* default 1 (ImageBase.h:9) */
void mtfo_am_initialize_pix_hess_pts(mtfo_am *a, const double *pts);
void mtfo_am_initialize_pix_hess_warped(mtfo_am *a, const double *pts, const double *hess_pts);
void mtfo_am_update_pix_hess_pts(mtfo_am *a, const double *pts);
void mtfo_am_update_pix_hess_warped(mtfo_am *a, const double *pts, const double *hess_pts);
/* second-order Hessians; -2 = FunctonNotImplemented in the reference (NCC self) */
int mtfo_am_cmpt_init_hessian2(mtfo_am *a, double *H, const double *J0, const double *d2I0, int S);
int mtfo_am_cmpt_curr_hessian2(mtfo_am *a, double *H, const double *Jt, const double *d2It, int S);
int mtfo_am_cmpt_self_hessian2(mtfo_am *a, double *H, const double *Jt, const double *d2It, int S);
int mtfo_am_cmpt_sum_of_hessians2(mtfo_am *a, double *H, const double *J0, const double *Jt,
	const double *d2I0, const double *d2It, int S);
void mtfo_am_initialize_similarity(mtfo_am *a);
void mtfo_am_initialize_grad(mtfo_am *a);
void mtfo_am_initialize_hess(mtfo_am *a);
void mtfo_am_update_similarity(mtfo_am *a, int prereq_only);
void mtfo_am_update_curr_grad(mtfo_am *a);
void mtfo_am_update_init_grad(mtfo_am *a);
double mtfo_am_get_similarity(const mtfo_am *a);
double mtfo_am_get_likelihood(const mtfo_am *a);
void mtfo_am_cmpt_init_jacobian(mtfo_am *a, double *g, const double *J0, int S);
void mtfo_am_cmpt_curr_jacobian(mtfo_am *a, double *g, const double *Jt, int S);
void mtfo_am_cmpt_difference_of_jacobians(mtfo_am *a, double *g,
	const double *J0, const double *Jt, int S);
void mtfo_am_cmpt_init_hessian(mtfo_am *a, double *H, const double *J0, int S);
void mtfo_am_cmpt_curr_hessian(mtfo_am *a, double *H, const double *Jt, int S);
void mtfo_am_cmpt_self_hessian(mtfo_am *a, double *H, const double *Jt, int S);
void mtfo_am_cmpt_sum_of_hessians(mtfo_am *a, double *H,
	const double *J0, const double *Jt, int S);
/* what: 0 I0 1 It 2 dI0_dx(2N) 3 dIt_dx(2N) 4 df_dI0 5 df_dIt 6 d2I0_dx2(4N) 7 d2It_dx2(4N) */
void mtfo_am_get(const mtfo_am *a, int what, double *dst);
/* AM::getDistFeatSize / updateDistFeat (SSDBase.h:116-125, NCC.cc:530-537, MI.cc:122, 736-747) and NN::generateDataset (NT/NN.cc:131-191) for
 * given perturbations (n_samples x S): dataset n_samples x feat_size */
int mtfo_am_dist_feat_size(const mtfo_am *a);
void mtfo_am_update_dist_feat(const mtfo_am *a, double *feat);
void mtfo_nn_generate_dataset(mtfo_am *am, mtfo_ssm *ssm, const double *perturbations, int n_samples, double *dataset);

/* ---- search methods (NT ESM / FCLK / ICLK) ---- */
typedef struct mtfo_sm_params {
	int max_iters;
	double epsilon;
	int jac_type;     /* ESM only: 0 Original, 1 DiffOfJacs */
	int hess_type;    /* per-SM enum, see the reference's *Params.h */
	int chained_warp;
	int leven_marq;
	double lm_delta_init;
	double lm_delta_update;
	int sec_ord_hess; /* second-order Hessians (off in every shipped config) */
} mtfo_sm_params;

typedef struct mtfo_tracker mtfo_tracker;
mtfo_tracker *mtfo_tracker_create(int sm_kind, mtfo_am *am, mtfo_ssm *ssm,
	const mtfo_sm_params *params);
void mtfo_tracker_destroy(mtfo_tracker *t);
void mtfo_tracker_initialize(mtfo_tracker *t, const double *corners);
/* returns the number of iterations executed */
int mtfo_tracker_update(mtfo_tracker *t);
void mtfo_tracker_set_region(mtfo_tracker *t, const double *corners);
void mtfo_tracker_get_region(const mtfo_tracker *t, double *corners);
/* per-iteration trace of the last update(): each record is
 * [f, g(S), H(S*S col-major), dp(S), corners(8)] ; returns record length */
/* 0, or -2 once the loop needed a virtual the reference leaves unimplemented */
int mtfo_tracker_status(const mtfo_tracker *t);
int mtfo_tracker_trace_len(const mtfo_tracker *t);
int mtfo_tracker_trace(const mtfo_tracker *t, int iter, double *dst);

/* ---- PF scoring (SM/src/PF.cc:198-278, deterministic part) ---- */
void mtfo_pf_score(mtfo_am *am, mtfo_ssm *ssm, const double *states, int n_particles,
	double *likelihoods, double *similarities);
/* binary multinomial resampling given the uniforms (PF.cc:345-394);
 * writes resampled source index per particle; returns max_wt_id */
int mtfo_pf_binary_multinomial_resample(const double *wts, int n,
	const double *uniforms, int *resample_ids);

/* one iteration of nt::PF::update's loop (SM/src/NT/PF.cc:260-447) with the random draws supplied; enum values are
 * PFParams.h:10-33 */
typedef struct mtfo_pf_params {
	int n_particles;
	int dynamic_model;        /* 0 RandomWalk, 1 AutoRegression1 */
	int update_type;          /* 0 Additive, 1 Compositional */
	int likelihood_func;      /* 0 AM, 1 Gaussian, 2 Reciprocal */
	int resampling_type;      /* 0 None, 1 BinaryMultinomial, 2 LinearMultinomial, 3 Residual */
	int mean_type;            /* 0 None, 1 SSM, 2 Corners */
	int corner_based_sampling;/* HomographyParams::corner_based_sampling */
	double measurement_sigma, ar_coeff;
	double sigma[8], mean[8]; /* the sampler's normal distributions (ProjectiveBase::initializeSampler) */
	int pt_based_sampling;    /* AffineParams::pt_based_sampling (Affine.cc:464-503): 0 geometric, 1, 2 */
} mtfo_pf_params;
/* the options of the shipped configuration on top of mtfo_pf_params: several sampler distributions (n_distr <= 8; sigma[0] / mean[0]
 * replace pp->sigma / pp->mean when n_distr > 1) with adaptive weights, and adaptive resampling */
typedef struct mtfo_pf_mix {
	int n_distr;
	double sigma[8][8], mean[8][8];
	int update_distr_wts;               /* PFParams::update_distr_wts */
	double min_distr_wt;                /* PFParams::min_distr_wt */
	double adaptive_resampling_thresh;  /* PFParams::adaptive_resampling_thresh: resample only when n_eff <= thresh * n */
	double distr_wts[8];                /* in: the weights the distribution ids are drawn from; out: those of the next iteration */
	const double *distr_uniforms;       /* [n] uniforms in (0, 1]: the distribution draw of every particle (n_distr > 1) */
	int *distr_ids_out;                 /* [n] or NULL */
	int resampled;                      /* out: 1 when this iteration resampled */
} mtfo_pf_mix;
int mtfo_pf_iteration_ex(mtfo_am *am, mtfo_ssm *ssm, const mtfo_pf_params *pp, mtfo_pf_mix *mx, double *states, double *ars,
	const double *normals, const double *uniforms, double max_similarity, double *wts_out, int *resample_ids, int *max_wt_id_out);
/* returns 0; -2 unknown resampling type; -4 an unsupported distribution setting; -3 an Affine sampler combination the reference throws for, or the additive
 * geometric one (Affine::stateToGeom: Eigen JacobiSVD conventions, not restated) */
int mtfo_pf_iteration(mtfo_am *am, mtfo_ssm *ssm, const mtfo_pf_params *pp, double *states, double *ars, const double *normals,
	const double *uniforms, double max_similarity, double *wts_out, int *resample_ids, int *max_wt_id_out);

/* ---- GridTracker (SM/src/GridTracker.cc:20-94 parameters, :97-160 constructor, :233-292 initialize / update / setRegion,
 * :294-343 backwardEstimation, :345-392 resetTrackers): the patch layout over a grid SSM, the frame loop over patch trackers and the
 * forward-backward error estimate.  The robust fit of the grid SSM to the patch centroids (estimateWarpFromPts: RANSAC / LMedS, out
 * of scope) is a callback. ---- */
typedef struct mtfo_grid_params {
	int grid_size_x, grid_size_y, patch_size_x, patch_size_y;
	int reset_at_each_frame;      /* 0 never, 1 re-initialise the patch trackers every frame, 2 setRegion only (GridTracker.cc:136) */
	int dyn_patch_size, patch_centroid_inside;
	double fb_err_thresh;         /* > 0: forward-backward error estimation (GridTracker.cc:186-190); shipped Config/modules.cfg:81: 2 */
	int fb_reinit;                /* re-initialise every patch tracker at its tracked location before the backward pass (:298-300); shipped 1 */
	int n_model_pts;              /* est_params.n_model_pts (SSMEstimatorParams.cc:63, shipped Config/modules.cfg:39: 4): the masked set is filled up to it (:322-332) */
} mtfo_grid_params;
/* estimateWarpFromPts(ssm_update, mask, prev_pts, curr_pts, est_params): n float point pairs -> S doubles */
typedef void (*mtfo_grid_estimator)(void *user, int n, const float *prev_pts, const float *curr_pts, double *ssm_update);
typedef struct mtfo_grid mtfo_grid;
void mtfo_grid_res(const mtfo_grid_params *gp, int *resx, int *resy);   /* GridTrackerParams::updateRes */
/* grid_ssm: resolution mtfo_grid_res; trackers: grid_size_x * grid_size_y patch trackers, or none (n_trackers 0: initialize / set_region
 * then only lay the patches out).  NULL on the mismatches the reference's constructor throws for. */
mtfo_grid *mtfo_grid_create(const mtfo_grid_params *gp, mtfo_ssm *grid_ssm, mtfo_tracker **trackers, int n_trackers);
void mtfo_grid_destroy(mtfo_grid *g);
void mtfo_grid_set_estimator(mtfo_grid *g, mtfo_grid_estimator est, void *user);
/* GridTracker::setImage :205-231 (enable_pyr off): every patch tracker's setImage(img) and curr_img = img.  The buffer is borrowed, as a
 * cv::Mat header borrows it; prev_img is the grid's own clone (:241-243, :266). */
void mtfo_grid_set_image(mtfo_grid *g, const float *img, int h, int w);
/* the mask half of GridTracker::backwardEstimation (:307-332) alone: fb_err_mask (n) and the surviving point pairs (up to n x 2 each,
 * filled up to n_model_pts in tracker order); returns their count */
int mtfo_grid_fb_mask(int n, const float *prev_pts, const float *curr_pts, const float *fb_prev_pts, double fb_err_thresh, int n_model_pts,
	unsigned char *fb_err_mask, float *prev_masked, float *curr_masked);
void mtfo_grid_initialize(mtfo_grid *g, const double *corners);
int mtfo_grid_update(mtfo_grid *g);   /* -1 without an estimator */
void mtfo_grid_set_region(mtfo_grid *g, const double *corners);
/* what: 0 region corners (8) 1 patch corners handed to the trackers (n x 8) 2 prev_pts (n x 2, float values) 3 curr_pts 4 ssm_update (S)
 * 5 fb_prev_pts (n x 2) 6 fb_err_mask (n, 0 / 1) 7 the tracker locations the backward pass started from (n x 8) 8 the regions the patch
 * trackers reached on the previous frame (n x 8) 9 the point pairs handed to the estimator in the last update: count, then prev (count x 2), curr (count x 2) */
void mtfo_grid_get(const mtfo_grid *g, int what, double *dst);

#ifdef __cplusplus
}
#endif
#endif
