/*
 * mtfhip_internal.h -- shared between the HIP kernel translation units (kernels_*.hip) and the C-ABI
 * implementation (api_core.hip, api_am.hip, api_fused.hip).  Not installed; the public contract is include/mtfhip.h.
 */
#ifndef MTFHIP_INTERNAL_H
#define MTFHIP_INTERNAL_H

#include <cstdlib>
#include <hip/hip_runtime.h>
#include "../../include/mtfhip.h"

namespace mtfhip {

struct ImgView {
	const float *data;
	int h, w, stride;     /* stride in floats between rows */
	int channels = 1;     /* 1: CV_32FC1 ; 3: CV_32FC3, interleaved (the mc:: sampling path) */
};

/* POD view of a batch, passed by value to kernels.  Every buf[i] is [B][per-target size]. */
struct BatchView {
	int B, N, S, ssm, am; /* N = patch_size = NP * C rows of every per-pixel AM array */
	int NP, C;            /* sample points per target, channels (MCSSD / MCNCC / MCMI: 3) */
	int unit_z;           /* 1: every init_z == 1 (affine SSM or parallelogram corners) */
	double *buf[MTFHIP_BUF_COUNT];
	double *warps;        /* [B][9] row-major curr_warp */
	double *states;       /* [B][8] curr_state */
};

/* accumulator slots produced by the reducing kernels, per target */
enum {
	ACC_H = 0,            /* 36: upper triangle of sum J_a J_b, row-major over (a<=b) with S=8 stride */
	ACC_G = 36,           /* 8 : sum v_i * Jrow_i  (v = residual-type vector of the mode) */
	ACC_RR = 44,          /* 1 : sum r^2 */
	ACC_G2 = 45,          /* 8 : second gemv (ESM generic: df_dI0 * J0) */
	ACC_COUNT = 56        /* padded so the halving butterfly divides evenly 3 times */
};
/* partial / reduced row of the fused NCC iteration: raw moments over the pixels of a target (J = the row the Hessian is
 * built from -- Jt, or the mean of J0 and Jt for ESM's Original Hessian; the g-type sums always use Jt and J0 themselves) */
enum {
	NCC_GRAM = 0,         /* 36: sum J_a J_b, upper triangle as ACC_H */
	NCC_SJ = 36,          /* 8 : sum Jt */
	NCC_ITJ = 44,         /* 8 : sum It Jt */
	NCC_I0J = 52,         /* 8 : sum I0 Jt */
	NCC_ITJ0 = 60,        /* 8 : sum It J0 */
	NCC_IT = 68, NCC_IT2 = 69, NCC_I0IT = 70,
	NCC_ACC_COUNT = 72    /* 72 -> 36 -> 18 -> 9 under the halving butterfly */
};

/* Every kernel launch of the library goes through this macro: the launch status (invalid configuration, missing code object,
 * out-of-resources ...) is read back at once and kept as a sticky error that the next C-ABI call which synchronises reports,
 * instead of surfacing as a timeout of the host-flag wait or at some later, unrelated call. */
void note_launch_error(hipError_t e, const char *file, int line);
#define MTFHIP_LAUNCH(...) do { hipLaunchKernelGGL(__VA_ARGS__); const hipError_t _le = hipGetLastError(); \
	if (_le != hipSuccess) ::mtfhip::note_launch_error(_le, __FILE__, __LINE__); } while (0)
constexpr int kBlock = 256;       /* threads per workgroup: 4 wave64 */
#ifndef MTFHIP_SLOTS
#define MTFHIP_SLOTS 512          /* resident workgroups of the fused kernel: 256 CUs x 2 (2 waves/SIMD, 4-wave groups) */
#endif
#ifndef MTFHIP_MIN_ROWS
#define MTFHIP_MIN_ROWS 4         /* at least this many 256-pixel rows per workgroup (amortises the reduction) */
#endif
constexpr int kMaxS = 8;

/* Work decomposition of the fused kernel: every target is cut into nblk chunks of `rows` 256-pixel rows so that
 * B * nblk is as close as possible to the number of resident workgroups -- one full round of workgroups, no
 * half-empty tail round (a 2.5-round grid costs ~8 % against a 2- or 1-round grid) */
inline void fused_decomposition(int N, int B, int &nblk, int &rows, int slots = MTFHIP_SLOTS) {
	const int total_rows = (N + kBlock - 1) / kBlock;
	int nb_max = (total_rows + MTFHIP_MIN_ROWS - 1) / MTFHIP_MIN_ROWS;
	if (nb_max < 1) nb_max = 1;
	if (B < 1) B = 1;
	/* cost model: rounds of resident workgroups x (rows per workgroup + the reduction epilogue, ~1.5 rows) */
	double best = 1e300;
	int best_nb = 1;
	for (int nb = 1; nb <= nb_max; ++nb) {
		const int r = (total_rows + nb - 1) / nb;
		const int real_nb = (total_rows + r - 1) / r;
		const long blocks = (long)B * real_nb;
		const long rounds = (blocks + slots - 1) / slots;
		const double cost = (double)rounds * (r + 1.5);
		if (cost < best - 1e-9) { best = cost; best_nb = real_nb; }
	}
	rows = (total_rows + best_nb - 1) / best_nb;
	nblk = (total_rows + rows - 1) / rows;
}
inline int fused_blocks_per_target(int N, int B) { int nb, r; fused_decomposition(N, B, nb, r); return nb; }
inline int simple_blocks_per_target(int N) {
	int nb = (N + kBlock * 4 - 1) / (kBlock * 4);
	return nb < 1 ? 1 : nb;
}

/* device-side solve + compositional update + convergence test for mtfhip_batch_track */
struct TrackState {
	double *acc;        /* [B][ACC_COUNT] reduced accumulators of this iteration */
	double *h0;         /* [B][64] constant (init) self Hessian, column-major, already negated sums */
	double *corners;    /* [B][8] current corners */
	double *init_corners_hm; /* [B][12] */
	int *active;        /* [B] 1 while the target still iterates */
	int *n_iters;       /* [B] */
	const double *ncc;    /* [B][8] NCC scalars (mean(I0), |I0 - mean|, ...), NULL for SSD */
	const double *ncc_tm; /* [B][52] NCC template moments: sum J0 | sum I0 J0 | Gram(J0) */
	int h_from_acc;       /* 1: every non-constant Hessian type reads the reduced row, ICLK's included (MI: k_finish_track_mi) */
	/* Levenberg-Marquardt (NT/ESM.cc:186-232, NT/FCLK.cc:205-250, NT/ICLK.cc:181-199) inside the device-side loop, per target:
	 * [0] prev_similarity [1] leven_marq_delta [2] state_reset [3] iter_id [4..11] the last state_update.  NULL: no LM. */
	double *lm;
	const double *f_ext;  /* [B] similarity of this pass when it is not a function of the reduced row (MI: d_mi_f), else NULL */
	/* debug trace of the device-side loop (mtfhip_batch_track_trace): record `pass` of target t at trace + (t cap + pass) kTraceStride:
	 * [0..63] H as the search method holds it before damping, row-major 8 x 8 | [64..71] g | [72..79] the state update applied |
	 * [80..87] corners after it | [88] f [89] pass [90] Levenberg-Marquardt undo [91] damping [92] 1 if H was recorded.  NULL: off. */
	double *trace;
	int trace_cap;
	/* second-order term of the Hessian (sec_ord_hess; k_second_order_ssd's output [B][S * S], entry (r, c) at c S + r), added to the
	 * first-order one times h_extra_scale (ESM SumOfStd halves the whole sum, NT/ESM.cc:339); the system is then solved with pivoting.
	 * NULL: none. */
	const double *h_extra;
	double h_extra_scale;
	int fast_finish;      /* tolerance mode, SSD family, first-order Hessian from the reduced row or the constant one: finish_track_fast_body */
	/* k_iclk_track / k_grid_fb, plain mode, right behind a fused re-initialisation (k_template_init in region mode, mtfhip_grid_reset): every patch
	 * starts from the identity warp and the zero state at its template's corners, which that kernel left in init_corners_hm -- the slab's warps /
	 * states / corners are not read, so the host does not upload them (one ingest launch and its gap less per frame, r06) */
	int fresh_reset;
};
constexpr int kLmStride = 12;
constexpr int kTraceStride = 96;

/* k_track_persist: per-target arrival counter and generation number of the in-kernel barrier between the pixel pass and the solve */
struct PersistState {
	int *arrive;            /* [B] zero between launches */
	unsigned *gen;          /* [B] monotonic: gen_base + pass + 1 once pass `pass` has been solved */
	unsigned gen_base;
	unsigned long long timeout_ticks;   /* of the 100 MHz wall clock */
};

/* results of the one-launch loop delivered straight into the host-coherent mirror of the state slab (warps | states | corners |
 * ... | iteration counts) by the workgroup that produced them; the last workgroup to finish raises the flag the host spins on.
 * host == NULL: nothing is published (k_publish_host does it in a launch of its own). */
/* The square-to-quadrilateral map behind set_corners (see rect_to_quad, mtfhip_api_internal.h), as ONE set of expressions for the host
 * and for the kernel that lays out a patch's grid itself (k_iclk_track in region mode): both are compiled without contraction, so the
 * nine entries are the same bits on either side.  H: row-major 3 x 3, H[8] = 1; false: degenerate corners. */
__host__ __device__ inline bool rect_to_quad_hd(double lo_x, double lo_y, double hi_x, double hi_y, const double *q, double *H) {
	const double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
	const double dx1 = x1 - x2, dx2 = x3 - x2, sx = x0 - x1 + x2 - x3;
	const double dy1 = y1 - y2, dy2 = y3 - y2, sy = y0 - y1 + y2 - y3;
	double a, b, c, d, e, f, g, h;
	if (sx == 0 && sy == 0) {   /* parallelogram: affine */
		a = x1 - x0; b = x3 - x0; c = x0; d = y1 - y0; e = y3 - y0; f = y0; g = 0; h = 0;
		if (a * e - b * d == 0) return false;
	} else {
		const double den = dx1 * dy2 - dy1 * dx2;
		if (den == 0) return false;
		g = (sx * dy2 - dx2 * sy) / den; h = (dx1 * sy - sx * dy1) / den;
		a = x1 - x0 + g * x1; b = x3 - x0 + h * x3; c = x0;
		d = y1 - y0 + g * y1; e = y3 - y0 + h * y3; f = y0;
	}
	/* (u, v) = ((x - lo_x) / wx, (y - lo_y) / wy) */
	const double wx = hi_x - lo_x, wy = hi_y - lo_y;
	const double rows[3][3] = {{a, b, c}, {d, e, f}, {g, h, 1.0}};
	double m[9];
	for (int r = 0; r < 3; ++r) {
		m[3 * r] = rows[r][0] / wx; m[3 * r + 1] = rows[r][1] / wy;
		m[3 * r + 2] = rows[r][2] - rows[r][0] * lo_x / wx - rows[r][1] * lo_y / wy;
	}
	if (m[8] == 0 || !(m[8] - m[8] == 0)) return false;   /* zero, infinite or NaN */
	if (m[8] == 1.0) for (int i = 0; i < 9; ++i) H[i] = m[i];   /* (x / 1.0 == x: a parallelogram's nine divisions are skipped) */
	else for (int i = 0; i < 9; ++i) H[i] = m[i] / m[8];
	H[8] = 1;
	return true;
}

/* rect_to_quad_hd's first refusals alone (collinear / coincident corners), ~12 flops: what a deferred reset checks on the host BEFORE it
 * commits anything, so that degenerate corners are an error without a state change, as on the non-deferred path (the kernel's own
 * report through n_iters = -1 stays as the backstop for the map's remaining refusal, a vanishing m[8]) */
__host__ __device__ inline bool quad_degenerate_hd(const double *q) {
	const double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
	const double dx1 = x1 - x2, dx2 = x3 - x2, sx = x0 - x1 + x2 - x3;
	const double dy1 = y1 - y2, dy2 = y3 - y2, sy = y0 - y1 + y2 - y3;
	if (sx == 0 && sy == 0) return (x1 - x0) * (y3 - y0) - (x3 - x0) * (y1 - y0) == 0;
	return dx1 * dy2 - dy1 * dx2 == 0;
}

/* GridTracker::resetTrackers' geometry for ONE patch (SM/src/GridTracker.cc:345-380; mtfhip_grid_layout, api_core.hip, describes it): the
 * grid SSM's points are the resx x resy grid of the unit square through the region's 4-corner map Wr = rect_to_quad(-0.5 .. 0.5, region),
 * a patch is the cell spanned by four of them (dyn_patch_size), a fixed-size rectangle centred on its own point, or one centred on the
 * cell's centroid (patch_centroid_inside).  One set of expressions for the host layout and for the kernel that lays its own patch out
 * (k_iclk_track, RegionIngest::layout; both compiled without contraction: the same bits). */
struct GridLayoutHD { int grid_size_x, grid_size_y, patch_size_x, patch_size_y, dyn_patch_size, patch_centroid_inside; };
__host__ __device__ inline double lin_spaced_hd(int i, int n, double lo, double hi) { return (n == 1 || i == n - 1) ? hi : lo + i * ((hi - lo) / (n - 1)); }
__host__ __device__ inline void grid_pt_hd(const double *Wr, int resx, int resy, int idx, double *x, double *y) {
	const int r = idx / resx, c = idx % resx;
	const double ny = lin_spaced_hd(r, resy, -0.5, 0.5), nx = lin_spaced_hd(c, resx, -0.5, 0.5);
	const double X = Wr[0] * nx + Wr[1] * ny + Wr[2], Y = Wr[3] * nx + Wr[4] * ny + Wr[5], Z = Wr[6] * nx + Wr[7] * ny + Wr[8];
	*x = X / Z; *y = Y / Z;
}
__host__ __device__ inline void grid_patch_corners_hd(const GridLayoutHD &g, const double *Wr, int k, double *pc) {
	const int extra = (g.dyn_patch_size || g.patch_centroid_inside) ? 1 : 0;
	const int resx = g.grid_size_x + extra, resy = g.grid_size_y + extra;
	const bool surround = extra != 0;
	const int sub_x = g.grid_size_x + 1;   /* _linear_idx(idy, idx) = idy * (grid_size_x + 1) + idx, :139-146 */
	const int row = k / g.grid_size_x, col = k % g.grid_size_x;   /* :354-355 */
	for (int q = 0; q < 8; ++q) pc[q] = 0.0;
	if (surround) {   /* :357-367 TL, TR, BR, BL of the cell */
		const int id[4] = {row * sub_x + col, row * sub_x + col + 1, (row + 1) * sub_x + col + 1, (row + 1) * sub_x + col};
		for (int q = 0; q < 4; ++q) grid_pt_hd(Wr, resx, resy, id[q], pc + 2 * q, pc + 2 * q + 1);
	}
	if (!g.dyn_patch_size) {   /* :369-380 */
		double cx, cy;
		if (g.patch_centroid_inside) {   /* utils::getCentroid miscUtils.h:481-487 */
			cx = (pc[0] + pc[2] + pc[4] + pc[6]) / 4.0;
			cy = (pc[1] + pc[3] + pc[5] + pc[7]) / 4.0;
		} else grid_pt_hd(Wr, resx, resy, k, &cx, &cy);   /* ssm.getPts().col(tracker_id) */
		const double half_x = g.patch_size_x / 2.0, half_y = g.patch_size_y / 2.0;   /* centrod_dist_x / _y :156-157 */
		const double min_x = cx - half_x, min_y = cy - half_y;   /* utils::Corners(cv::Rect_<double>) miscUtils.h:42-52 */
		const double max_x = min_x + g.patch_size_x, max_y = min_y + g.patch_size_y;
		pc[0] = pc[6] = min_x; pc[2] = pc[4] = max_x;
		pc[1] = pc[3] = min_y; pc[5] = pc[7] = max_y;
	}
}

/* k_iclk_track in REGION mode (mtfhip_batch_track_region / mtfhip_grid_update, r04): the workgroup of a patch takes the patch's region
 * corners (and its template's NCC scalars) straight from the pinned staging buffer, derives the square-to-quadrilateral map, lays out
 * its own sample grid -- kept in registers for the loop, written to INIT_PTS / INIT_HXY / INIT_Z for whoever asks later -- and starts
 * from the identity warp: what set_corners' ingest + k_init_grid did in a launch of their own in front of the loop (7.8 us of a 64 us
 * frame, plus the gap).  corners == NULL: off (the slab and the grid are already on the device). */
struct RegionIngest {
	const double *corners;      /* device-visible pinned host memory: [B][8], (x, y) per corner, TL TR BR BL */
	const double *ncc;          /* likewise: [B][8] NCC scalars of the templates (slab layout) */
	double *d_ncc, *d_w0, *d_init_corners_hm;   /* the slab's device copies of what the workgroup derives */
	double lo_x, lo_y, hi_x, hi_y;
	int resx, resy, force_unit_z;
	/* layout != 0 (mtfhip_grid_frame with fixed-size patches, r05): the workgroup computes its patch's corners itself from the GRID's
	 * region (grid_patch_corners_hd over region_map = rect_to_quad(-0.5 .. 0.5, region), formed on the host once) -- `corners` is not
	 * read: no PCIe round trip in front of the grid layout, and the host lays the patches out for its mirrors AFTER the launch */
	int layout;
	GridLayoutHD grid;
	double region_map[9];
};

struct HostPublish {
	char *host;
	size_t dbl_bytes;   /* offset of the int part of the slab */
	int B;
	int *count;
	unsigned long long *flag, seq;
	int fenced;         /* publish_fenced(): the release / acquire form of the hand-over instead of acknowledged write-through stores */
};
/* How a kernel hands results to the host (k_finish_host, k_publish_host, publish_target, the particle filter's estimate).  Default:
 * relaxed system-scope (write-through) stores, s_waitcnt vmcnt(0), a relaxed agent-scope arrival counter, a relaxed system-scope flag
 * store by the last arriver -- what a release does in hardware minus the L2 write-backs (2.5 us each), but a data race under the
 * HIP / HSA memory model: it relies on the stores being write-through atomics, on per-wave vmcnt and on ordered posted writes
 * (r04 advisor).  MTFHIP_PUBLISH_FENCE=1 restores the model-conforming form at run time for every publisher: agent-scope release
 * before the counter, acq_rel on the counter, __threadfence_system() + a system-scope release store of the flag. */
inline int publish_fenced() {
	static const int v = [] { const char *e = std::getenv("MTFHIP_PUBLISH_FENCE"); return (e && e[0] == '1') ? 1 : 0; }();
	return v;
}

struct FusedArgs {
	int mode;          /* accumulation mode: 0 FCLK-type, 1 ESM-type, 2 ICLK-lite (see k_fused_ssd) */
	int chained;
	int materialize;
	int hess_mean;     /* ESM hess_type Original: outer products of (J0+Jt)/2 instead of Jt */
	int j0_init_variant; /* with j0_recompute: J0 is the Init variant (non-chained initialize, or refreshed by set_region), else Warped at identity */
	int j0_recompute;  /* 1: the template's SD rows are rebuilt from dI0_dx (bit-identical to the stored J0, 16 instead of 8 S bytes per pixel) */
	int rows_per_block; /* 256-pixel rows walked by one workgroup (fused_decomposition) */
	double grad_eps;
	double norm_mult, norm_add;
	const int *active; /* optional [B] mask: targets with 0 are skipped (device-side loop) */
	/* single-target batches: the warp and state travel in the kernel arguments (read from the kernarg segment instead of
	 * bv.warps / bv.states; the kernel stores them there afterwards) -- no separate upload in front of every launch */
	int inline_warp;
	double iw[9], is[8];
	/* 1: tolerance-mode arithmetic (mtfhip_device.h, "tolerance-mode arithmetic"); only with materialize == 0 -- the
	 * materialised arrays stay bit-identical to what the per-function kernels write */
	int fast_math;
};

/* one-time probe: does the kernel-argument segment hold (BatchView, ImgView, FusedArgs) where fused_lk_body's inline-warp path reads them? */
bool kernarg_layout_verified(hipStream_t st);
void launch_queue_delay(double microseconds, hipStream_t st);
/* ---- launchers (all asynchronous on `st`) ---- */
void launch_apply_warp(const BatchView &bv, hipStream_t st);
void launch_grad_pts(const BatchView &bv, double eps, hipStream_t st);
void launch_sample(const BatchView &bv, const ImgView &im, const double *pts, double *out,
	double mult, double add, hipStream_t st);
void launch_update_model(const BatchView &bv, const ImgView &im, const double *pts, double *I0, double mult, double add,
	double frame_count, double alpha, int running_avg, hipStream_t st);
void launch_img_grad(const BatchView &bv, const ImgView &im, const double *pts, double *grad,
	double eps, double mult, hipStream_t st);
void launch_warped_img_grad(const BatchView &bv, const ImgView &im, const double *grad_pts, double *grad,
	double eps, double mult, hipStream_t st);
void launch_pix_jacobian(const BatchView &bv, int variant, const double *grad, double *J, hipStream_t st);
void launch_mean_jacobian(const BatchView &bv, hipStream_t st);
/* MI device-side loop: g and H of the fused MI passes handed to the finish in one launch (k_finish_track_mi) */
void launch_finish_track_mi(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts, int sum_std, int gmode,
	const double *mi_H, const double *gpart, int ng, double *rows, hipStream_t st);
/* df_dI0 = It - I0 ; partial sums of r^2 into `partials` ([B][nblk][ACC_COUNT]) */
void launch_ssd_residual(const BatchView &bv, double *partials, int nblk, hipStream_t st);
void launch_negate(const double *src, double *dst, size_t n, hipStream_t st);
/* g-type partials: sum_i (v1[i]*J1[i,:] (+ v2[i]*J2[i,:] into ACC_G2, or + v1[i]*J2[i,:] into ACC_G when sum_mode)) */
void launch_gemv(const BatchView &bv, const double *v1, const double *J1, const double *v2, const double *J2,
	int sum_mode, double *partials, int nblk, hipStream_t st);
/* H-type partials: sum_i J[i,a] J[i,b] */
void launch_gram(const BatchView &bv, const double *J, double *partials, int nblk, hipStream_t st);
/* NCC pieces (see the kernel comments); `sc` is the [B][8] per-target scalar block */
void launch_vec_sum(const BatchView &bv, const double *v, double *partials, int nblk, hipStream_t st);
void launch_ncc_centered(const BatchView &bv, const double *sc, double *partials, int nblk, hipStream_t st);
void launch_ncc_grad(const BatchView &bv, const double *sc, int curr, double *out, double *partials, int nblk, hipStream_t st);
void launch_sub_mean(const BatchView &bv, double *v, const double *sc, hipStream_t st);
void launch_col_sum(const BatchView &bv, const double *J, double *partials, int nblk, hipStream_t st);
void launch_ncc_hess(const BatchView &bv, const double *sc, const double *colmean, const double *J, double *partials,
	int nblk, hipStream_t st);
/* MI pieces; table block offsets (doubles) of the per-target MI state, see kernels_mi.hip */
enum {
	MI_NB = 16,
	MI_HIST_INIT = 0, MI_HIST_CURR = 16, MI_LOG_INIT = 32, MI_LOG_CURR = 48,
	MI_JOINT = 64, MI_JOINT_LOG = 320, MI_T_CURR = 576, MI_T_INIT = 832,
	MI_SELF_JOINT = 1088, MI_T_SELF = 1344, MI_SIZE = 1600
};
void launch_mi_hist(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, double *partials,
	int nblk, int row_len, hipStream_t st);
void launch_mi_hist_finish(const BatchView &bv, int nb, double pre_seed, double norm_mult, int mode, int first_init,
	const double *partials, int nblk, int row_len, double *tb, double *f_out, hipStream_t st);
void launch_mi_factor(const BatchView &bv, int nb, int curr, double *tb, hipStream_t st);
void launch_mi_grad(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, const double *tb,
	int table_off, double *out, hipStream_t st);
void launch_mi_hess(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, const double *tb,
	int table_off, int transpose_q, const double *J, double *partials, int nblk, int row_len, hipStream_t st);
void launch_mi_hess_finish(const BatchView &bv, int nb, const double *partials, int nblk, int row_len, const double *tb,
	int joint_off, int hist_off, int transpose_q, double *out, hipStream_t st);
void launch_mi_tables_iter(const BatchView &bv, int nb, double pre_seed, double norm_mult, int with_self, const double *partials, int nblk,
	int row_len, double *tb, double *f_out, hipStream_t st);
void launch_mi_hist_self(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, double *partials,
	int nblk, int row_len, hipStream_t st);
/* dI0 != NULL: k_mi_grad_gemv rebuilds the template's steepest-descent row from dI0_dx [B][2][N], the grid points [B][N] x,y
 * (and z [B][N] of the homogeneous grid, NULL = 1) instead of reading J0 -- valid under the conditions of FusedArgs::j0_recompute */
struct MiJ0Rebuild { const double *dI0, *pts, *z; int hom, init_variant; };
void launch_mi_grad_gemv(const BatchView &bv, int nb, double norm_mult, const double *It, const double *I0, const double *tb,
	const double *Jt, const double *J0, const MiJ0Rebuild &rb, double *df_dIt, double *df_dI0, double *partials, int nblk, hipStream_t st);
/* ---- the recompute form of the MI iteration (kernels_mi_fused.hip; tolerance-mode arithmetic, 8 bins) ---- */
struct MiFastPlan {
	int nb = 8;           /* the AM's n_bins: 8 -> the NB = 8 kernels, otherwise (<= 10; hk 0 / 1) the NB = 10 ones */
	int hk;               /* Hessian pass: 0 none (constant Hessian), 1 self(Jt), 2 curr, 3 init(J0) */
	int hrow;             /* pixel Jacobian of the Hessian pass: 0 Jt, 1 J0, 2 (J0 + Jt) / 2 */
	int j0_mode;          /* 0 the template's row is not needed, 1 rebuilt from dI0_dx, 2 read from J0 */
	int j0_init_variant;
	int need_dft, need_df0, g_mean;
	int nonchained = 0;   /* the search method's chained_warp = 0: pass 2 takes the rounded steps of updateGradPts' four points (mi_finish) */
	int hist_from_joint = 0;   /* pass 1: the histogram of It as the row sums of the joint histogram (partition of unity: every window of I0 lies
	                              inside the bins and sums to one) instead of a block product of its own -- a third of the pass's matrix work */
	double grad_eps, norm_mult, norm_add, hist_norm;
	const int *active;    /* device flags, NULL = all */
	const double *tb;     /* [B][MI_SIZE] */
	const double *poly = nullptr;   /* [B][mi_poly_size()]: the tables as per-class polynomials (launch_mi_poly_tables), read by pass 2 */
};
void launch_mi_poly_tables(const BatchView &bv, const double *tb, double hist_norm, int with_self, double *poly, hipStream_t st);
/* launch_mi_tables_iter + launch_mi_poly_tables in one launch (hist_norm is the tables' norm_mult) */
void launch_mi_tables_poly(const BatchView &bv, int nb, double pre_seed, double norm_mult, int with_self, const double *partials, int nblk, int row_len,
	double *tb, double *f_out, double *poly, hipStream_t st);
int mi_poly_size(int nb = 8);
void launch_mi_pass_hist(const BatchView &bv, const ImgView &im, const MiFastPlan &pl, double *partials, int nblk, int row_len, hipStream_t st);
void launch_mi_pass_grad_hess(const BatchView &bv, const ImgView &im, const MiFastPlan &pl, double *partials, int nblk, hipStream_t st);
void launch_mi_finish_fast(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts, const MiFastPlan &pl, int gmode, int do_track,
	const double *partials, int nblk, double *out_H, double *out_g, double *rows, hipStream_t st);
void launch_mi_score_candidates(const BatchView &bv, const ImgView &im, const MiFastPlan &pl, const double *dev_states, int lo, int cnt,
	double *partials, int nblk, int row_len, double pre_seed, double alpha, int likelihood_func, double measurement_sigma, double max_similarity,
	double *wts, double *sim, hipStream_t st);
int mi_fast_row_len(int nb = 8);
/* ---- particle filter (kernels_pf.hip): proposal + scoring, cumulative weights, resampling + estimate ---- */
enum { PF_SAMPLER_STATE = 0,       /* one normal per state component (ProjectiveBase::generatePerturbation) */
	PF_SAMPLER_HOM_CORNERS = 1,    /* Homography, corner based: 10 normals (Homography.cc:899-911) */
	PF_SAMPLER_AFF_PTS1 = 2,       /* Affine, pt_based_sampling 1: 6 normals (Affine.cc:475-482) */
	PF_SAMPLER_AFF_PTS2 = 3,       /* Affine, pt_based_sampling 2: 8 normals (Affine.cc:483-491) */
	PF_SAMPLER_AFF_GEOM = 4 };     /* Affine, geometric: 6 normals through geomToState (Affine.cc:495-502) */
struct PfLaunch {
	int n, S;
	int dynamic_model, update_type, sampler, nz, likelihood_func, resampling_type, mean_type;
	double ar_coeff, measurement_sigma, max_similarity;
	double sigma[8], mean[8], init_corners[8], init_corners_hm[12], aux_inv[9], canon[6];
	unsigned long long seed;
	unsigned iter;
	const double *normals, *uniforms;   /* device arrays or NULL (Philox) */
	/* several sampler distributions (PF.cc:240-269, 345-369) and adaptive resampling (PF.cc:381-390); n_distr == 1 and
	 * min_eff_particles == 0: neither */
	int n_distr;
	const double *distr_uniforms;       /* [n] device array or NULL (Philox): the distribution draw of every particle */
	double min_distr_wt, min_eff_particles;
};
struct PfBuffers {
	double *st, *ar;            /* [n][S] the current (resampled) particle set */
	double *prop, *prop_ar;     /* [n][S] this iteration's proposals */
	double *next, *next_ar;     /* [n][S] where the selection pass leaves the proposals of the next iteration (look-ahead) */
	double *wts, *sim;          /* [>= n] weights at global particle indices; similarities or NULL */
	double *cum, *sub16, *chunk_tot, *chunk_incl;
	double *parts, *gparts, *out;   /* per-workgroup rows of the selection pass, their per-group folds, the estimate */
	int *res_order;             /* residual resampling: particle indices by weight, highest first (the last index stays last) */
	/* distributions / adaptive resampling (all NULL when neither is in use): [n_distr][8] sigmas and means, the cumulative distribution
	 * weights the ids are drawn from (rewritten by the scan for the next iteration), the weights themselves, the particles'
	 * distribution ids, per-chunk statistics of the scan [nch][17], the flag "this iteration resamples" */
	double *distr_sigma, *distr_mean, *distr_cum, *distr_wts, *scan_stats;
	int *distr_ids, *resample_flag;
	int *ids, *counters;        /* [0] the scan's arrival counter, [1] the selection pass's top-level counter, [2 ...] one per group of 64
	                               workgroups of the selection pass; zero between launches */
};
/* Peer-store exchange of the sharded filter's weights (the alternative to the all-gather, api_pf.hip: mtfhip_pf_set_exchange): every
 * rank owns a MAILBOX -- two weight vectors (the exchanges alternate between them, so a rank that runs one iteration ahead never
 * overwrites what a slower one still reads) and one arrival counter per source rank.  The scoring kernel stores every weight it
 * produces into every rank's mailbox (peer memory, mapped once at set-up); the last of its workgroups to finish adds 1 to its
 * rank's counter in every mailbox (release, system scope); the first kernel that reads the weights spins until every counter
 * has reached the count the host knows it must reach (acquire, system scope).  No host-enqueued collective, no extra launch. */
constexpr int kPfMaxPeers = 8;
struct PfPeerPush {
	int world, rank;                              /* world == 0: no peer stores */
	double *wts[kPfMaxPeers];                     /* every rank's weight vector of this exchange (own rank: unused, the scorer's wts) */
	unsigned long long *counters[kPfMaxPeers];    /* every rank's arrival counters [world]: this rank bumps entry `rank` */
	unsigned *arrive;                             /* this rank's own: the workgroups of the storing launch count themselves in; zero between launches */
};
struct PfPeerWait {
	int world, rank;                              /* world == 0: nothing to wait for */
	const unsigned long long *counters;           /* this rank's mailbox counters [world] */
	unsigned long long expected[kPfMaxPeers];     /* what each must have reached before the weights are complete */
	int *err;                                     /* host-mapped: set to 1 when the bounded spin gave up */
};
void launch_pf_propose(int ssm, const PfLaunch &p, const PfBuffers &bf, const double *st_in, const double *ar_in, double *st_out, double *ar_out, hipStream_t st);
void launch_score_block(const BatchView &bv, const ImgView &im, const double *states, int lo, int cnt, double alpha, double norm_mult,
	double norm_add, const double *ncc_sc, double *wts, double *sim, int likelihood_func, double measurement_sigma, double max_similarity,
	int fast_math, const PfPeerPush *peer /* or NULL */, const double *hull /* [8] host, or NULL: PfScoreArgs::hull */, const float *pair /* the row-pair copy of the frame, or NULL */, hipStream_t st);
/* weights [lo, lo + cnt) of `wts` -> every other rank's mailbox + the arrival: for scorers that do not store to the peers themselves */
void launch_pf_peer_push(const PfPeerPush &peer, const double *wts, int lo, int cnt, hipStream_t st);
void launch_pf_peer_wait(const PfPeerWait &w, hipStream_t st);   /* the wait in a launch of its own (no scan in this iteration) */
/* weights -> chunk-local cumulative weights + chunk table; wait != NULL: every workgroup first waits for the peers' weights */
void launch_pf_scan(const PfLaunch &p, const PfBuffers &bf, const PfPeerWait *wait, hipStream_t st);
/* local: no scan in front, the launch builds the cumulative weights in LDS itself (n <= pf_local_max(), multinomial resampling, no scan
 * statistics; wait: the peers' weights, as the scan's); pert_in: the perturbations of iteration p.iter + 1 drawn ahead (NULL: the
 * selection pass draws them itself); pert_out: where extra workgroups leave those of p.iter + 2 (NULL: none) -- [n][8] each */
struct PfSelectPlan { int local = 0; const PfPeerWait *wait = nullptr; const double *pert_in = nullptr; double *pert_out = nullptr; int estimate = 1; };
int pf_local_max();
void launch_pf_select(int ssm, const PfLaunch &p, const PfBuffers &bf, int lookahead, double *host_out /* or NULL */,
	unsigned long long *host_flag, unsigned long long seq, const PfSelectPlan &plan, hipStream_t st);
void launch_pf_fill(int n, int S, const double *dev_state, double *states, double *ars, hipStream_t st);
/* flag: NULL, or the scan's "this iteration resamples" (adaptive resampling): with 0 the residual kernels leave everything as it is */
void launch_pf_residual_prep(int n, const double *total, double *wts, double *keys, int *idx, const int *flag, hipStream_t st);
void launch_pf_residual_copies(int n, const double *wts, const int *order, int *copies, hipStream_t st);
void launch_pf_residual_map(int n, const int *order, const int *copies, const int *starts, int *ids, hipStream_t st);
int pf_parts_per_block();
int pf_chunk();
/* sums partials over blocks: out[B][ACC_COUNT] */
void launch_finish(double *partials, int nblk, double *out, int B, hipStream_t st);
void launch_finish_rows(double *partials, int nblk, int row_len, double *out, int B, hipStream_t st);
void launch_finish_host(double *partials, int nblk, int row_len, double *out_host, int *count, unsigned long long *flag_host,
	unsigned long long seq, int B, hipStream_t st);
void launch_publish_host(const void *src, void *dst_host, size_t bytes, int *count, unsigned long long *flag_host,
	unsigned long long seq, hipStream_t st);
/* the fused LK iteration for SSD */
void launch_fused_ssd(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials,
	int nblk, hipStream_t st);
/* second-order path: hess_pts, image Hessians ([N][4]), SSM pixel Hessians ([S*S][N] planes), sum_p w[p] d2[:, p] */
void launch_hess_pts(const BatchView &bv, double eps, hipStream_t st);
void launch_img_hess(const BatchView &bv, const ImgView &im, const double *pts, double *hess, double eps, double mult, hipStream_t st);
void launch_warped_img_hess(const BatchView &bv, const ImgView &im, const double *pts, const double *hp, double *hess, double eps,
	double mult, hipStream_t st);
void launch_pix_hessian(const BatchView &bv, int variant, const double *hess, const double *grad, double *D, hipStream_t st);
void launch_weighted_plane_sum(const BatchView &bv, const double *d2a, const double *d2b, const double *w, double *partials, int nblk,
	double *out, hipStream_t st);
void launch_mean_planes(const double *a, const double *b, double *o, size_t n, hipStream_t st);
/* fused second-order term of the SSD Hessians, pixel-Hessian blocks in registers only; out[t][S*S] */
/* NCC weights for the fused second-order term: the fused NCC pass's partial rows of this iteration ([B][nblk][NCC_ACC_COUNT]) and
 * the template scalars ([B][8]: mean(I0), |I0 - mean|); rows == nullptr selects SSD's residual weights */
struct SecondOrderNcc { const double *rows; int nblk; const double *sc; };
/* MI weights for the fused second-order term (MI.cc:659-695: df_dI0 / df_dIt of the pixel): the gradient-factor tables of this
 * iteration ([B][MI_SIZE], k_mi_tables_iter) and the histogram normaliser; tb == nullptr: not MI.  Eight bins (the recompute passes). */
struct SecondOrderMi { const double *tb; double hist_norm; };
void launch_second_order_ssd(const BatchView &bv, const ImgView &im, int term, int chained, int d0_variant, double grad_eps,
	double hess_eps, double norm_mult, double norm_add, double *partials, int nblk, double *out, hipStream_t st, int own_pts = 0,
	SecondOrderNcc nc = SecondOrderNcc{nullptr, 0, nullptr}, SecondOrderMi mi = SecondOrderMi{nullptr, 0.0});
/* pre-processing / pyramid (float32 images) */
void launch_hist_eq(float *gray, int rows, int cols, unsigned *hist256, float *lut256, hipStream_t st);
void launch_to_gray(const void *raw, int rows, int cols, size_t stride_bytes, int channels, int depth_f32, float *out, hipStream_t st);
void launch_sym5(const float *src, float *tmp, float *dst, int rows, int cols, const float kx[3], const float ky[3], hipStream_t st);
void launch_pyr_down(const float *src, int srows, int scols, float *dst, int drows, int dcols, hipStream_t st);
void launch_resize_linear(const float *src, int srows, int scols, float *dst, int drows, int dcols, hipStream_t st);
/* NN dataset rows: features of C warped patches of target 0 (SSD: It, NCC: centred / normalised It) */
/* nt::NN's dataset generation in one launch (kernels_nn.hip) */
struct NnArgs {
	const double *perts_in;   /* [n_samples][S] device: the perturbations (any sampler of the SSM, made by the caller), or NULL: drawn on the device */
	double *perts_out;        /* [n_samples][S] device or NULL: the perturbations used (NN keeps them: ssm_perturbations, NT/NN.cc:141-147) */
	double sigma[8], mean[8]; /* ProjectiveBase::generatePerturbation: component k ~ N(mean_k, sigma_k) */
	unsigned long long seed;
	double base[9];           /* the SSM's current warp */
	int row_lo;               /* global index of the launch's first sample (row sharding: draws are keyed by the global index) */
	double norm_mult, norm_add;
};
void launch_pair_image(const ImgView &im, float *pair /* [h][w][2] */, hipStream_t st);
void launch_nn_dataset(const BatchView &bv, const ImgView &im, const NnArgs &a, int count, double *feat, double *warps, const double *hull, hipStream_t st);
size_t nn_warps_bytes(int count);
bool nn_two_launch_ok(const BatchView &bv, const ImgView &im, int fast_math);
void launch_sample_candidates(const BatchView &bv, const ImgView &im, const double *dev_states, int C, double norm_mult,
	double norm_add, double *dev_feat, hipStream_t st);
/* whole ICLK loop in one launch, one workgroup per target (N <= 16 * kBlock); false if N is too large */
constexpr int kIclkTrackMaxPix = 8 * kBlock;   /* (the grid points of a thread's pixels stay in registers: k_iclk_track) */
/* k_grid_fb (kernels_grid_fb.hip): a frame's forward pass, the re-initialisation at the tracked location and the backward pass on the
 * previous frame in one launch; per patch the backward pass leaves 8 corners + its iteration count (-1: degenerate tracked corners) */
struct GridFbOut { double *host; double *dev; int reinit; };   /* [B][9] each: device-visible pinned memory | device memory (either may be NULL); reinit = fb_reinit: 0 = the backward pass keeps the forward pass's template and state (GridTracker.cc:297-299 skipped) */
bool launch_grid_fb(const BatchView &bv, const ImgView &im, const ImgView &imp, const mtfhip_sm_desc &sm, const TrackState &ts, const double *h0inv,
	const double *ncc_sc, double norm_mult, double norm_add, double grad_eps, const HostPublish &pub, const GridFbOut &fo, const RegionIngest &rg, hipStream_t st);
bool launch_iclk_track(const BatchView &bv, const ImgView &im, const mtfhip_sm_desc &sm, const TrackState &ts,
	const double *h0inv, const double *ncc_sc, double norm_mult, double norm_add, int fast_math, const HostPublish &pub, const RegionIngest &rg, hipStream_t st);
/* skip_off / skip_len (multiples of 16 bytes): a section that is left as it is on the device */
void launch_ingest_host(const void *src_host, void *dst, size_t bytes, hipStream_t st, size_t skip_off = 0, size_t skip_len = 0);
void launch_fused_mc(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk, hipStream_t st);   /* bv.C > 1 */
/* a small patch's whole nt::ICLK::initialize in one launch (kernels_init.hip): template sample, gradient, steepest-descent rows,
 * moments, constant self Hessian and its inverse; the small results also go to a pinned host record of kInitRec doubles per target
 * (H0 64 | NCC scalars 8 | sum J0 8 | sum I0 J0 8 | Gram(J0) 36 | pad) behind the usual publish hand-over */
constexpr int kInitRec = 128;
constexpr int kTemplateInitMaxPix = 4 * 256;
struct InitPublish { double *host; int *count; unsigned long long *flag, seq; int fenced; };
void launch_template_init(const BatchView &bv, const ImgView &im, double grad_eps, double norm_mult, double norm_add, double *h0, double *h0inv,
	double *ncc, double *ncc_tm, const InitPublish &pub, const RegionIngest &rg, hipStream_t st);   /* rg.corners != NULL: the kernel lays out the grid itself */
/* one launch per pass with the finish folded in behind a last-arriver counter (kernels_step.hip); arrive: [B] ints, zero between launches */
bool track_step_available(const BatchView &bv, const FusedArgs &fa);
void launch_track_step(const BatchView &bv, const ImgView &im, const FusedArgs &fa, const mtfhip_sm_desc &sm, const TrackState &ts,
	double *partials, int nblk, int *arrive, hipStream_t st);
void launch_track_persist(const BatchView &bv, const ImgView &im, const FusedArgs &fa, const mtfhip_sm_desc &sm, const TrackState &ts,
	double *partials, int nblk, const PersistState &ps, int max_passes, hipStream_t st);
bool launch_init_grid_ingest(const BatchView &bv, const double *host_w0_dev, int resx, int resy, double lo_x, double lo_y,
	double hi_x, double hi_y, int force_unit_z, const void *src_host, void *dst, size_t bytes, int write_curr, hipStream_t st);
/* phase control of the two-queue loop (k_finish_track): this queue's and the other queue's time stamps, the fraction of a period to keep */
struct PhaseCtl { unsigned long long *mine; const unsigned long long *other; double frac; };
void launch_finish_track(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts, const double *partials,
	int nblk, hipStream_t st, PhaseCtl pc = PhaseCtl{nullptr, nullptr, 0.0});

} // namespace mtfhip
#endif
