/*
 * mtfhip_api_internal.h -- what the translation units of the C-ABI implementation share: error plumbing, the small
 * host-side math (3x3 warps, 4-corner DLT), the context / batch handles and their helpers, and the declarations of
 * the functions that cross unit boundaries (the deferred-fusion layer lives in api_am.hip, the NCC moment forms in
 * api_fused.hip).  Not installed: include/mtfhip.h is the public contract.
 */
#ifndef MTFHIP_API_INTERNAL_H
#define MTFHIP_API_INTERNAL_H
#include "mtfhip_internal.h"

#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace mtfhip;

namespace mtfhip {
void launch_init_grid(const BatchView &bv, const double *dev_w0, int resx, int resy, double lo_x, double lo_y,
	double hi_x, double hi_y, int force_unit_z, hipStream_t st);
}

/* ------------------------------------------------------------------ errors */
extern thread_local std::string g_last_error;   /* one per thread for the whole library; defined in api_core.hip */
static int fail(int code, const char *fmt, ...) {
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof(buf), fmt, ap);
	va_end(ap);
	g_last_error = buf;
	return code;
}
#define HIP_TRY(expr)                                                                        \
	do {                                                                                     \
		hipError_t _e = (expr);                                                              \
		if (_e != hipSuccess)                                                                \
			return fail(MTFHIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
	} while (0)
#define TRY(expr)                  \
	do {                           \
		int _r = (expr);           \
		if (_r != MTFHIP_OK) return _r; \
	} while (0)

/* ------------------------------------------------------------------ small host math */
struct M3 {
	double m[9];
};
static M3 m3_identity() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
static M3 m3_mul(const M3 &a, const M3 &b) {
	M3 c;
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j)
			c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
	return c;
}
/* Matrix3d::inverse() as Eigen evaluates it for fixed 3x3: cofactors / determinant */
static M3 m3_inverse(const M3 &a) {
	const double *u = a.m;
	M3 c;
	c.m[0] = u[4] * u[8] - u[5] * u[7]; c.m[1] = u[2] * u[7] - u[1] * u[8]; c.m[2] = u[1] * u[5] - u[2] * u[4];
	c.m[3] = u[5] * u[6] - u[3] * u[8]; c.m[4] = u[0] * u[8] - u[2] * u[6]; c.m[5] = u[2] * u[3] - u[0] * u[5];
	c.m[6] = u[3] * u[7] - u[4] * u[6]; c.m[7] = u[1] * u[6] - u[0] * u[7]; c.m[8] = u[0] * u[4] - u[1] * u[3];
	double det = u[0] * c.m[0] + u[1] * c.m[3] + u[2] * c.m[6];
	double inv_det = 1.0 / det;
	for (int i = 0; i < 9; ++i) c.m[i] *= inv_det;
	return c;
}
/* getWarpFromState: Homography.cc:94-107, Affine.cc:116-130 */
static M3 warp_from_state(int ssm, const double *p) {
	M3 W;
	if (ssm == MTFHIP_SSM_HOMOGRAPHY) {
		W.m[0] = 1 + p[0]; W.m[1] = p[1]; W.m[2] = p[2];
		W.m[3] = p[3]; W.m[4] = 1 + p[4]; W.m[5] = p[5];
		W.m[6] = p[6]; W.m[7] = p[7]; W.m[8] = 1;
	} else {
		W.m[0] = 1 + p[2]; W.m[1] = p[3]; W.m[2] = p[0];
		W.m[3] = p[4]; W.m[4] = 1 + p[5]; W.m[5] = p[1];
		W.m[6] = 0; W.m[7] = 0; W.m[8] = 1;
	}
	return W;
}
/* getStateFromWarp: Homography.cc:116-132, Affine.cc:132-143 */
static void state_from_warp(int ssm, double *p, const M3 &W) {
	if (ssm == MTFHIP_SSM_HOMOGRAPHY) {
		p[0] = W.m[0] - 1; p[1] = W.m[1]; p[2] = W.m[2]; p[3] = W.m[3]; p[4] = W.m[4] - 1; p[5] = W.m[5];
		p[6] = W.m[6]; p[7] = W.m[7];
	} else {
		p[0] = W.m[2]; p[1] = W.m[5]; p[2] = W.m[0] - 1; p[3] = W.m[1]; p[4] = W.m[3]; p[5] = W.m[4] - 1;
		p[6] = p[7] = 0;
	}
}
/* The homography that maps the corners of the axis-aligned rectangle [lo_x, hi_x] x [lo_y, hi_y] (TL, TR, BR, BL) onto
 * four given corners -- what the reference obtains from the 4-point DLT (utils::computeHomographyDLT,
 * Utilities/src/warpUtils.cc:171-224, JacobiSVD of the 8 x 9 system).  Four point pairs determine the matrix uniquely up
 * to scale, so the closed form of the square-to-quadrilateral map (Heckbert 1989, eq. 2.12), composed with the
 * rectangle's normalisation and scaled to m[8] = 1, is the same matrix to rounding -- ~40 flops instead of an 8 x 8
 * elimination per target (GridTracker sets 256 of them per frame).  false: degenerate corners. */
static bool rect_to_quad(double lo_x, double lo_y, double hi_x, double hi_y, const double *q, M3 &H) {
	return rect_to_quad_hd(lo_x, lo_y, hi_x, hi_y, q, H.m);   /* (one set of expressions for host and device: mtfhip_internal.h) */
}

/* ------------------------------------------------------------------ handles */
struct Timer {
	std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
	double total_ms = 0;
	int n = 0;
	long launches = 0;
	/* [start, end) of every timed launch in ms since the context's reference event: launches of one family can overlap when the
	 * device-side loop runs on two queues, and the time the family was executing is then the union, not the sum */
	std::vector<std::pair<double, double>> spans;
};

struct mtfhip_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	/* second queue of the device-side loop (track_core: two chunks of independent targets in flight, one's solve + update under the
	 * other's pixel pass); created on first use, ordered against `stream` by the two events */
	hipStream_t extra_streams[3] = {nullptr, nullptr, nullptr};
	hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
	unsigned long long *d_phase = nullptr;   /* [4] wall-clock stamps of the queues' last solve (PhaseCtl) */
	ImgView img{nullptr, 0, 0, 0};
	float *img_owned = nullptr;
	size_t img_capacity = 0;
	/* the previous frame (GridTracker's prev_img = curr_img.clone(), SM/src/GridTracker.cc:241-243, 266): mtfhip_image_keep_prev.  An image
	 * the context owns is kept by exchanging the two owned buffers (the next upload fills the other one: no copy); a borrowed one is copied. */
	ImgView prev{nullptr, 0, 0, 0};
	float *prev_owned = nullptr;
	size_t prev_capacity = 0;
	unsigned char *raw = nullptr; size_t raw_capacity = 0;      /* staging of the raw frame (pre-processing) */
	float *tmp_a = nullptr, *tmp_b = nullptr; size_t tmp_capacity = 0; /* gray / row-pass intermediates */
	/* The current image with every texel next to the one BELOW it -- pair[2 (y W + x)] = I[y][x], pair[2 (y W + x) + 1] = I[y + 1][x] -- so that
	 * the four texels of a bilinear cell are 16 contiguous bytes: ONE gather per sample instead of two for the kernels that sit on the vector
	 * memory path (the candidate scorer; r06 -- the NN rows kernel gained nothing from it).  Built on demand (ensure_pair_image) for images the context owns (an uploaded or derived
	 * image cannot change behind the library's back; a borrowed one can), rebuilt when img_serial has moved. */
	float *pair_owned = nullptr; size_t pair_capacity = 0;
	unsigned long long img_serial = 1, pair_serial = 0, pair_demand_serial = 0;
	size_t pair_demand = 0;   /* candidates scored on the current image so far without the copy (pair_image_if_it_pays) */
	int n_cus = 0;   /* compute units: the persistent loop needs its whole grid resident */
	bool timing = false;
	int timing_stride = 1;   /* events are recorded around every timing_stride-th launch of a family */
	std::map<std::string, Timer> timers;
	hipEvent_t ev_ref = nullptr;   /* recorded by timing_reset: origin of Timer::spans */
	std::vector<hipEvent_t> free_events;
	std::vector<struct mtfhip_batch *> batches;   /* live batches: deferred work is flushed before the image changes */
};

struct TimedScope {
	mtfhip_ctx *ctx;
	hipEvent_t a = nullptr, b = nullptr;
	Timer *tm = nullptr;
	hipStream_t on;
	TimedScope(mtfhip_ctx *c, const char *family, hipStream_t stream = nullptr) : ctx(c), on(stream ? stream : c->stream) {
		if (!ctx->timing) return;
		Timer *cand = &ctx->timers[family];
		if ((cand->launches++ % ctx->timing_stride) != 0) return;
		tm = cand;
		auto get = [&]() {
			hipEvent_t e;
			if (!ctx->free_events.empty()) { e = ctx->free_events.back(); ctx->free_events.pop_back(); }
			else (void)hipEventCreate(&e);
			return e;
		};
		a = get(); b = get();
		(void)hipEventRecord(a, on);
	}
	~TimedScope() {
		if (!tm) return;
		(void)hipEventRecord(b, on);
		tm->pending.emplace_back(a, b);
	}
};

constexpr int kAccRowMax = NCC_ACC_COUNT > ACC_COUNT ? NCC_ACC_COUNT : ACC_COUNT;   /* widest partial / reduced row */

struct TargetHost {
	M3 warp;
	double state[8];
	double corners[8], init_corners[8];
	double init_corners_hm[12];
	double f;
	/* NCC scalars (AM/src/NCC.cc members) */
	double I0_mean, It_mean, a, b, c, gmean;
	double h0[64]; /* constant self Hessian of the template (column-major), set by init_template */
	/* NCC, fused path: moments of the template's pixel Jacobian (constant while J0 is): sum J0, sum I0 J0, Gram(J0) */
	double ncc_sj0[8], ncc_i0j0[8], ncc_gram0[36];
};

struct mtfhip_batch;
static int push_warps(mtfhip_batch *b);
struct mtfhip_batch {
	mtfhip_ctx *ctx;
	mtfhip_patch_desc desc;
	int B, N, S;          /* N = patch_size = NP * C (rows of the per-pixel AM arrays) */
	int NP = 0, C = 1;    /* sample points per target, channels */
	double norm_mult = 1, norm_add = 0;
	double *buf[MTFHIP_BUF_COUNT];
	size_t per_target[MTFHIP_BUF_COUNT];
	double *d_warps = nullptr, *d_states = nullptr;
	double *d_partials = nullptr, *d_acc = nullptr, *d_scratch_pts = nullptr, *d_w0 = nullptr;
	double *d_h0 = nullptr, *d_corners = nullptr, *d_init_corners_hm = nullptr, *d_cand = nullptr;
	double *d_ncc = nullptr, *d_colmean = nullptr; /* [B][8] NCC scalars / column means */
	/* MI: per-target table block, block partial rows, similarity and Hessian outputs */
	double *d_mi_tb = nullptr, *d_mi_part = nullptr, *d_mi_f = nullptr, *d_mi_H = nullptr;
	double *d_mi_poly = nullptr;   /* [B][mi_poly_size()], allocated by the first recompute iteration */
	/* the fused template initialisation (kernels_init.hip, init_template's ICLK fast path): its small results arrive in this pinned
	 * record (kInitRec doubles per target) behind init_flag; init_mirror_seq != 0: the host mirrors (th[].h0, NCC scalars and template
	 * moments) are older than the device copies until pull_init_mirrors() has folded the record in (lazy_flush does) */
	double *h_init_rec = nullptr, *h_init_rec_dev = nullptr;
	unsigned long long *h_init_flag = nullptr, *h_init_flag_dev = nullptr, init_seq = 0, init_mirror_seq = 0;
	bool init_rec_device = false;   /* the pending record was NOT published to pinned memory (grid re-initialisations, r05): pull_init_mirrors copies d_h0 / d_ncc / d_ncc_tm instead */
	double *d_mi_red = nullptr;   /* [B][mi_row_len] block rows summed (the Hessian assembly then reads one row per target) */
	double *d_h0inv = nullptr; /* [B][64] inverse of the constant Hessian (one-launch ICLK) */
	double *d_d2_part = nullptr, *d_d2_out = nullptr, *d_d2_w = nullptr; /* second-order term: block rows, [B][64] sums, MI self weights */
	double hess_eps = 1.0;
	/* MTFHIP_MATH_*: arithmetic of the non-materialising kernels (include/mtfhip.h) */
	int math_mode = (std::getenv("MTFHIP_MATH") && (std::getenv("MTFHIP_MATH")[0] == 'r' || std::getenv("MTFHIP_MATH")[0] == '0')) ? MTFHIP_MATH_REPLAY : MTFHIP_MATH_FAST;
	bool init_pix_hess = false;
	/* J0 and dI0_dx are still exactly what init_template produced (no setter / pixel-Jacobian call touched them since): the
	 * fused kernel may then rebuild J0's rows from dI0_dx instead of reading them (MTFHIP_J0_RECOMPUTE=0 disables) */
	unsigned int frame_count = 0;   /* ImageBase::frame_count: ++ in initializePixVals and updateModel (ImageBase.cc:74, SSD.cc:51) */
	bool j0_is_template = false;
	long corners_epoch = 0, j0_template_corners_epoch = -1;   /* set_corners moves the grid: J0 rows depend on init_pts */
	int j0_variant = MTFHIP_JAC_WARPED;
	std::vector<double> template_corners;   /* [B][8] corners the stored J0 was computed on */
	bool j0_recompute_enabled = !(std::getenv("MTFHIP_J0_RECOMPUTE") && std::getenv("MTFHIP_J0_RECOMPUTE")[0] == '0');
	int d0_variant = MTFHIP_JAC_WARPED; /* how the template's pixel Hessian was formed (fused second-order path) */
	int mi_row_len = 0;
	double mi_hist_norm = 0;
	size_t cand_capacity = 0;
	double *d_cand_mi = nullptr; size_t cand_mi_capacity = 0;   /* MI candidate scoring: histogram rows per candidate */
	int *d_active = nullptr, *d_iters = nullptr;
	/* The small per-target state lives in ONE device allocation (warps | states | corners | init_corners_hm | ncc | w0 |
	 * active | iters) mirrored by two pinned staging buffers, so that set_corners and track each move it with a single
	 * copy (a grid frame used to cost 14 small copies and 4 stream syncs around a 100 us kernel). */
	char *d_slab = nullptr, *h_stage_a = nullptr, *h_stage_b = nullptr;
	hipEvent_t ev_a = nullptr, ev_b = nullptr;
	/* k_track_persist: barrier words, the generation counter the host advances per launch; persist_ok is cleared for good when a
	 * launch could not keep its workgroups resident (the two-launch loop finishes the call and serves the later ones) */
	int *d_persist = nullptr;
	unsigned persist_gen = 0;
	bool persist_ok = true;
	bool stage_a_busy = false;   /* ev_a guards an upload from h_stage_a that may still be in flight */
	/* warp + state of every target after setState / compositionalUpdate: one copy from a pinned double buffer, no sync */
	double *h_wstage[2] = {nullptr, nullptr};
	hipEvent_t ev_w[2] = {nullptr, nullptr};
	int wflip = 0;
	/* CURR_PTS / CURR_HXY / CURR_Z lag behind the warp: only the un-fused kernels read them, so k_apply_warp runs when
	 * one of those is about to be launched (lazy_flush) or the arrays are read, not after every update */
	bool pts_stale = false;
	double *d_it_shadow = nullptr;
	double *d_ncc_tm = nullptr;   /* [B][52] NCC template moments for the device-side finish */
	/* the one-launch forward-backward frame (k_grid_fb): requested by mtfhip_grid_frame_fb around its track call */
	/* set by a fused grid re-initialisation (grid_reinit_fused: identity warps, zero states, corners = the templates' corners on host AND device),
	 * cleared by whatever changes the mirrors' warps / states / corners next (set_corners_core, apply_states) and consumed by the next track_core:
	 * its one-launch kernel then starts from init_corners_hm and the slab is not uploaded (TrackState::fresh_reset) */
	bool fresh_reinit = false;
	bool fb_fused_req = false, fb_fused_reinit = true;
	double *h_fb = nullptr, *h_fb_dev = nullptr, *d_fb = nullptr;   /* [B][9] the backward pass's corners | iteration count: pinned host copy, device copy */
	double *d_nn_warps = nullptr; size_t nn_warps_cap = 0;   /* NN dataset, tolerance mode: the samples' warps between k_nn_warps and k_nn_rows */
	double *d_lm = nullptr;       /* [B][kLmStride] Levenberg-Marquardt state of the device-side loop */
	double *d_trace = nullptr; int trace_cap = 0;   /* [B][trace_cap][kTraceStride] debug trace of the device-side loop, or NULL */
	/* NCC: a fused iteration updated the scalars (It_mean, a, b, f) on the host only; the un-fused kernels read d_ncc */
	bool ncc_host_newer = false;
	/* mtfhip_grid_frame behind a fused re-initialisation (reset-every-frame mode): the loop kernel is ENQUEUED behind k_template_init instead
	 * of the host first waiting for that kernel's record and folding 1 KB per patch into its mirrors -- lazy_flush leaves the record
	 * pending while this is set, and the slab upload of the call leaves the device's NCC scalars (the kernel's own) alone */
	bool hold_init_pull = false;
	size_t slab_bytes = 0, slab_dbl_bytes = 0;
	double *h_acc = nullptr; /* pinned */
	/* Zero-copy read-back of the reduced rows: h_acc is host-coherent pinned memory the reduction kernel writes directly
	 * (h_acc_dev = its device address) followed by a sequence number in h_flag; the host spins on the flag instead of
	 * paying a copy command plus a stream synchronisation per iteration (MTFHIP_ZERO_COPY=0: copy + sync) */
	double *h_acc_dev = nullptr;
	unsigned long long *h_flag = nullptr, *h_flag_dev = nullptr, acc_seq = 0;
	char *h_stage_a_dev = nullptr, *h_stage_b_dev = nullptr;   /* device addresses of the staging buffers when kernels may read them (k_ingest_host), else NULL */
	char *h_pub = nullptr, *h_pub_dev = nullptr;   /* host-coherent mirror of the state slab, written by k_publish_host (track's read-back) */
	int *d_fin_count = nullptr;
	int nblk_max;
	int unit_z = 1;
	/* INIT_PTS is the lattice set_corners (or the grid kernel's region mode) laid out INSIDE init_corners: what lets the candidate scorer
	 * trust the corners as the hull of the sample points.  Cleared when the caller writes a grid of its own (mtfhip_batch_write /
	 * mtfhip_batch_device_ptr of INIT_PTS / INIT_HXY / INIT_Z). */
	bool grid_from_corners = false;
	/* a deferred affine reset whose host half is still to be written (set_corners_core / set_corners_finish_deferred): the caller's
	 * corners, valid for the duration of the C-ABI call that deferred them */
	const double *deferred_corners = nullptr;
	bool deferred_for_track = false;
	/* mtfhip_grid_frame, fixed-size patches (r05): the kernel lays its patches out itself from the grid's region (RegionIngest::layout),
	 * so the HOST layout -- for the mirrors, the staged slab and the template-grid comparison -- is also deferred behind the launch:
	 * set_corners_finish_deferred runs mtfhip_grid_layout into deferred_patches first */
	bool deferred_layout = false, deferred_template_check = false;
	mtfhip_grid_desc deferred_gdesc{};
	double deferred_region[8] = {0, 0, 0, 0, 0, 0, 0, 0}, deferred_region_map[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	std::vector<double> deferred_patches;
	bool have_corners = false, init_pix_vals = false, init_pix_grad = false, init_sim = false, init_grad = false;
	bool it_valid = false, dit_valid = false, jt_valid = false;
	std::vector<TargetHost> th;

	/* Deferred fusion of the per-function entry points (SSD, single channel; DESIGN.md "drop-in path").  The pixel-level
	 * producers of an iteration (updatePixVals, updatePixGrad, cmpt*PixJacobian, updateSimilarity, update*Grad,
	 * mean Jacobian) only RECORD themselves; the first call that needs a number on the host (cmpt*Jacobian) runs the
	 * fused kernel with materialize=1 when the recorded set is one of the ESM / FCLK / ICLK call sequences, and every
	 * other entry point first replays what is pending through the un-fused kernels, in the recorded order -- so the
	 * buffers and results are those of the call-by-call execution either way. */
	struct Lazy {
		bool enabled = false;
		long seq = 0;
		long pv = 0;            /* update_pix_vals(NULL) */
		long gp = 0;            /* ssm_update_grad_pts(grad_eps) */
		long pg = 0; int pg_kind = 0;   /* update_pix_grad(NULL) = 1, update_pix_grad_warped(NULL) = 2 */
		long pj = 0; int pj_variant = -1;   /* ssm_cmpt_pix_jacobian(variant, DIT_DX -> JT) */
		long sim = 0, cg = 0, ig = 0, jm = 0;
		bool any() const { return pv || gp || pg || pj || sim || cg || ig || jm; }
		bool sim_need_f = false;
		/* DF_DI0 (updateSimilarity) / DF_DIT (updateCurrGrad) were skipped by a fused launch: refreshed from IT / I0 on
		 * first use, and in any case before IT is overwritten by something that does not overwrite them as well */
		bool df0_stale = false, dft_stale = false;
		/* ...and when IT has to change first, the old IT is kept instead of being consumed: the launch writes the other
		 * of two IT buffers (pointer swap, no copy) and the stale vector remembers that it refers to the shadow */
		bool df0_sh = false, dft_sh = false, shadow_valid = false;
		struct NccSave { double It_mean, a, b, f; };
		std::vector<NccSave> ncc_shadow;   /* NCC scalars that belong to the shadow IT */
		bool no_cache = false;
		/* Levenberg-Marquardt reads f in the middle of the iteration (NT/ESM.cc:186-204), which replays updatePixVals +
		 * updateSimilarity un-fused; the fused launch may still serve the rest when IT and DF_DI0 are known to belong to
		 * the current warp and image (it recomputes the same IT bits): epoch counts warp / image changes */
		long epoch = 0, it_epoch = -1, df0_it_ver = -1;
		/* NCC: the moment rows of the last fused launch ([B][NCC_ACC_COUNT]); Hessian requests are answered from them while
		 * IT and the Jacobian they were taken from are unchanged.  ncc_tm_ver: J0 version of the template moments. */
		std::vector<double> ncc_M; bool ncc_M_mean = false; long ncc_M_it = -1, ncc_M_jt = -1, ncc_M_jm = -1, ncc_tm_ver = -1;
		/* SSD: sum r J0 of the lean launch that served getSimilarity() -- it IS cmptInitJacobian(J0) for this IT and J0 */
		std::vector<double> sim_g; long sim_g_it = -1, sim_g_j0 = -1;
		/* MI: IT version the self joint histogram / its factor table belong to, and whether the search method has been
		 * asking for the self Hessian (then the fused histogram pass takes the self histogram along) */
		long mi_self_it = -1; bool mi_want_self = false;
		/* Gram matrices that are already on the host: [B][36] upper triangles, valid while version matches */
		long ver[MTFHIP_BUF_COUNT] = {0};
		int gram_buf = -1; long gram_ver = -1; std::vector<double> gram;
		long gram0_ver = -1; std::vector<double> gram0;   /* J0: constant between template changes */
	} lz;

	/* Single-target batches (the literal drop-in shape): set_state / compositional_update only mark the device copy of the warp
	 * stale.  The fused launch that normally follows carries the warp in its kernel arguments (fused_view) and refreshes the
	 * device copy itself; anything else that builds a BatchView uploads it first (view).  MTFHIP_INLINE_WARP=0: always upload. */
	mutable bool warps_dirty = false;
	bool inline_warp_ok = false;
	BatchView view() const {
		if (warps_dirty) { warps_dirty = false; (void)push_warps(const_cast<mtfhip_batch *>(this)); }
		return view_raw();
	}
	BatchView view_raw() const {
		BatchView v;
		v.B = B; v.N = N; v.S = S; v.ssm = desc.ssm; v.am = desc.am; v.unit_z = unit_z; v.NP = NP; v.C = C;
		for (int i = 0; i < MTFHIP_BUF_COUNT; ++i) v.buf[i] = buf[i];
		v.warps = d_warps; v.states = d_states;
		return v;
	}
};

static int ensure_buf(mtfhip_batch *b, int id) {
	if (b->buf[id]) return MTFHIP_OK;
	HIP_TRY(hipMalloc(&b->buf[id], sizeof(double) * b->per_target[id] * b->B));
	HIP_TRY(hipMemsetAsync(b->buf[id], 0, sizeof(double) * b->per_target[id] * b->B, b->ctx->stream));
	return MTFHIP_OK;
}

/* host state of every target -> one staging image of the slab */
static void fill_stage(const mtfhip_batch *b, char *stage, const double *w0 /* [B][9] or NULL */, int active, bool zero_iters) {
	const size_t Bt = (size_t)b->B;
	double *p = reinterpret_cast<double *>(stage);
	double *w = p, *s = p + 9 * Bt, *cr = p + 17 * Bt, *ic = p + 25 * Bt, *nc = p + 37 * Bt, *pw0 = p + 45 * Bt;
	int *act = reinterpret_cast<int *>(stage + b->slab_dbl_bytes), *it = act + Bt;
	for (int t = 0; t < b->B; ++t) {
		const TargetHost &h = b->th[t];
		std::memcpy(w + 9 * t, h.warp.m, sizeof(double) * 9);
		std::memcpy(s + 8 * t, h.state, sizeof(double) * 8);
		std::memcpy(cr + 8 * t, h.corners, sizeof(double) * 8);
		std::memcpy(ic + 12 * t, h.init_corners_hm, sizeof(double) * 12);
		double *q = nc + 8 * t;
		q[0] = h.I0_mean; q[1] = h.c; q[2] = h.It_mean; q[3] = h.b; q[4] = h.f; q[5] = h.gmean; q[6] = q[7] = 0;
		if (w0) std::memcpy(pw0 + 9 * t, w0 + 9 * t, sizeof(double) * 9);
		act[t] = active;
		if (zero_iters) it[t] = 0;
	}
}

static int push_warps(mtfhip_batch *b) {
	const int k = b->wflip; b->wflip ^= 1;
	HIP_TRY(hipEventSynchronize(b->ev_w[k]));   /* the copy that last read this buffer (two updates ago) is long done */
	double *w = b->h_wstage[k], *s = w + 9 * (size_t)b->B;   /* d_states follows d_warps in the slab */
	for (int t = 0; t < b->B; ++t) {
		std::memcpy(&w[9 * t], b->th[t].warp.m, sizeof(double) * 9);
		std::memcpy(&s[8 * t], b->th[t].state, sizeof(double) * 8);
	}
	HIP_TRY(hipMemcpyAsync(b->d_warps, w, sizeof(double) * 17 * (size_t)b->B, hipMemcpyHostToDevice, b->ctx->stream));
	HIP_TRY(hipEventRecord(b->ev_w[k], b->ctx->stream));
	return MTFHIP_OK;
}

/* the BatchView of a fused launch: a stale single-target warp goes into the kernel arguments instead of being uploaded */
static inline BatchView fused_view(mtfhip_batch *b, FusedArgs &fa) {
	fa.inline_warp = 0;
	if (b->warps_dirty && b->B == 1) {
		b->warps_dirty = false;
		fa.inline_warp = 1;
		std::memcpy(fa.iw, b->th[0].warp.m, sizeof(double) * 9);
		std::memcpy(fa.is, b->th[0].state, sizeof(double) * 8);
		return b->view_raw();
	}
	return b->view();
}
/* corners = dehomogenise(curr_warp * init_corners_hm) (Homography.cc:87-90) / affine top rows (Affine.cc:105) */
static void update_corners(mtfhip_batch *b, int t) {
	TargetHost &h = b->th[t];
	for (int q = 0; q < 4; ++q) {
		const double *c = &h.init_corners_hm[3 * q];
		const double *W = h.warp.m;
		double x = W[0] * c[0] + W[1] * c[1] + W[2] * c[2];
		double y = W[3] * c[0] + W[4] * c[1] + W[5] * c[2];
		if (b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY) {
			double d = W[6] * c[0] + W[7] * c[1] + W[8] * c[2];
			x = x / d; y = y / d;
		}
		h.corners[2 * q] = x; h.corners[2 * q + 1] = y;
	}
}

int launch_error_pending();   /* api_core.hip: the sticky status of MTFHIP_LAUNCH, cleared when reported */
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
	__builtin_ia32_pause();
#elif defined(__aarch64__)
	__asm__ __volatile__("yield");
#endif
}
/* wait until the kernel that was given `seq` has stored it behind its host-coherent writes.  The kernel normally reports within a
 * few microseconds of the launch that precedes this call, so the wait starts as a spin; when the stream is busy with longer
 * work (a whole device-side loop: tens to hundreds of microseconds) the spin gives the core away between polls after 200 us,
 * and after 50 ms the runtime's blocking synchronisation takes over (it costs ~100 us of its own, hence not earlier). */
static int wait_host_flag(mtfhip_batch *b, unsigned long long seq) {
	TRY(launch_error_pending());
	const auto t0 = std::chrono::steady_clock::now();
	bool yielding = false;
	for (unsigned spins = 0;; ++spins) {
		if (__atomic_load_n(b->h_flag, __ATOMIC_ACQUIRE) == seq) return MTFHIP_OK;
		if (yielding) std::this_thread::yield(); else cpu_relax();
		if ((spins & 0x3ff) == 0x3ff || yielding) {
			const auto dt = std::chrono::steady_clock::now() - t0;
			if (dt > std::chrono::milliseconds(50)) break;
			if (dt > std::chrono::microseconds(1000)) yielding = true;
		}
	}
	/* the kernel did not report in: let the runtime tell why */
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	if (__atomic_load_n(b->h_flag, __ATOMIC_ACQUIRE) == seq) return MTFHIP_OK;
	return fail(MTFHIP_ERR_HIP, "device results were not delivered to host memory");
}
/* fold the record of a fused template initialisation into the host mirrors (see mtfhip_batch::init_mirror_seq) */
static int pull_init_mirrors(mtfhip_batch *b) {
	if (!b->init_mirror_seq) return MTFHIP_OK;
	const bool ncc_am = b->desc.am == MTFHIP_AM_NCC;
	if (b->init_rec_device) {
		/* the kernel left its results on the device only (what it publishes otherwise is a copy of d_h0 | d_ncc | d_ncc_tm): three small copies and
		 * a synchronisation, paid by the rare caller that reads the mirrors behind a grid re-initialisation instead of by every frame */
		static thread_local std::vector<double> h0v, ncv, tmv;
		const size_t Bt = (size_t)b->B;
		h0v.resize(64 * Bt); ncv.resize(8 * Bt); tmv.resize(52 * Bt);
		HIP_TRY(hipMemcpyAsync(h0v.data(), b->d_h0, sizeof(double) * 64 * Bt, hipMemcpyDeviceToHost, b->ctx->stream));
		if (ncc_am) {
			HIP_TRY(hipMemcpyAsync(ncv.data(), b->d_ncc, sizeof(double) * 8 * Bt, hipMemcpyDeviceToHost, b->ctx->stream));
			HIP_TRY(hipMemcpyAsync(tmv.data(), b->d_ncc_tm, sizeof(double) * 52 * Bt, hipMemcpyDeviceToHost, b->ctx->stream));
		}
		HIP_TRY(hipStreamSynchronize(b->ctx->stream));
		for (int t = 0; t < b->B; ++t) {
			TargetHost &h = b->th[t];
			std::memcpy(h.h0, &h0v[64 * (size_t)t], sizeof(double) * 64);
			if (ncc_am) {
				const double *r = &ncv[8 * (size_t)t], *m = &tmv[52 * (size_t)t];
				h.I0_mean = r[0]; h.c = r[1]; h.It_mean = r[2]; h.b = r[3]; h.f = r[4]; h.gmean = r[5];
				std::memcpy(h.ncc_sj0, m, sizeof(double) * 8);
				std::memcpy(h.ncc_i0j0, m + 8, sizeof(double) * 8);
				std::memcpy(h.ncc_gram0, m + 16, sizeof(double) * 36);
			}
		}
		b->init_mirror_seq = 0; b->init_rec_device = false;
		return MTFHIP_OK;
	}
	const unsigned long long seq = b->init_mirror_seq;
	const auto t0 = std::chrono::steady_clock::now();
	bool ok = false;
	for (unsigned spins = 0;; ++spins) {
		if (__atomic_load_n(b->h_init_flag, __ATOMIC_ACQUIRE) == seq) { ok = true; break; }
		cpu_relax();
		if ((spins & 0x3ff) == 0x3ff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
	}
	if (!ok) {
		HIP_TRY(hipStreamSynchronize(b->ctx->stream));
		if (__atomic_load_n(b->h_init_flag, __ATOMIC_ACQUIRE) != seq) return fail(MTFHIP_ERR_HIP, "the template initialisation's results were not delivered to host memory");
	}
	const bool ncc = b->desc.am == MTFHIP_AM_NCC;
	for (int t = 0; t < b->B; ++t) {
		const double *r = b->h_init_rec + (size_t)t * mtfhip::kInitRec;
		TargetHost &h = b->th[t];
		std::memcpy(h.h0, r, sizeof(double) * 64);
		if (ncc) {
			h.I0_mean = r[64]; h.c = r[65]; h.It_mean = r[66]; h.b = r[67]; h.f = r[68]; h.gmean = r[69];
			std::memcpy(h.ncc_sj0, r + 72, sizeof(double) * 8);
			std::memcpy(h.ncc_i0j0, r + 80, sizeof(double) * 8);
			std::memcpy(h.ncc_gram0, r + 88, sizeof(double) * 36);
		}
	}
	b->init_mirror_seq = 0;
	return MTFHIP_OK;
}
/* fixed-order sum of the per-workgroup rows -> h_acc ([B][row_len]) on the host, and wait for it */
static int read_rows(mtfhip_batch *b, int nblk, int row_len) {
	if (b->h_acc_dev) {
		const unsigned long long seq = ++b->acc_seq;
		launch_finish_host(b->d_partials, nblk, row_len, b->h_acc_dev, b->d_fin_count, b->h_flag_dev, seq, b->B, b->ctx->stream);
		return wait_host_flag(b, seq);
	}
	if (row_len == ACC_COUNT) launch_finish(b->d_partials, nblk, b->d_acc, b->B, b->ctx->stream);
	else launch_finish_rows(b->d_partials, nblk, row_len, b->d_acc, b->B, b->ctx->stream);
	HIP_TRY(hipMemcpyAsync(b->h_acc, b->d_acc, sizeof(double) * row_len * b->B, hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	return MTFHIP_OK;
}
static int read_acc(mtfhip_batch *b, int nblk) { return read_rows(b, nblk, ACC_COUNT); }

static int need_image(mtfhip_batch *b) {
	if (!b->ctx->img.data) return fail(MTFHIP_ERR_LOGIC, "no current image: call mtfhip_image_upload/borrow first");
	if (b->ctx->img.channels != b->C)   /* ImageBase::setCurrImg: "Input image type does not match the required type" */
		return fail(MTFHIP_ERR_INVALID_ARG, "ImageBase::setCurrImg::Input image has %d channel(s), the appearance model expects %d", b->ctx->img.channels, b->C);
	return MTFHIP_OK;
}
/* the fused, one-launch and candidate kernels are single-channel; MCSSD / MCNCC / MCMI go through the per-function entry points */
static int single_channel(const mtfhip_batch *b, const char *fn) {
	if (b->C != 1) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "%s: multi-channel appearance models use the per-function entry points", fn);
	return MTFHIP_OK;
}
static int j_buf_ok(int id) { return id == MTFHIP_BUF_J0 || id == MTFHIP_BUF_JT || id == MTFHIP_BUF_JM; }

/* resolves a `pts` argument: NULL -> the batch's own device buffer, else upload into scratch */
static int resolve_pts(mtfhip_batch *b, const double *host, int own_buf, size_t per_target, const double **out) {
	if (!host) {
		if (!b->buf[own_buf]) return fail(MTFHIP_ERR_LOGIC, "device points not available yet");
		*out = b->buf[own_buf];
		return MTFHIP_OK;
	}
	HIP_TRY(hipMemcpyAsync(b->d_scratch_pts, host, sizeof(double) * per_target * b->B, hipMemcpyHostToDevice, b->ctx->stream));
	*out = b->d_scratch_pts;
	return MTFHIP_OK;
}

extern "C" {

/* ------------------------------------------------------------------ deferred fusion (see mtfhip_batch::Lazy) */
static inline void touch(mtfhip_batch *b, int id) { ++b->lz.ver[id]; }
static inline void touch_all(mtfhip_batch *b) { for (int i = 0; i < MTFHIP_BUF_COUNT; ++i) ++b->lz.ver[i]; }
int lazy_flush(mtfhip_batch *b, bool pts = true);
int ensure_df(mtfhip_batch *b);
static int ensure_one(mtfhip_batch *b, bool curr);
void stale_clear(mtfhip_batch *b, bool df0, bool dft);
int protect_stale(mtfhip_batch *b, bool w0, bool wt);
int lazy_try_similarity(mtfhip_batch *b);
int do_cmpt_pix_jacobian(mtfhip_batch *b, int variant, int grad_buf, int dst_buf);
static int do_mean_jacobian(mtfhip_batch *b);
static int lazy_flush_ctx(mtfhip_ctx *c) {   /* called by everything that replaces the current image */
	for (mtfhip_batch *b : c->batches) {
		/* the record of a fused template initialisation does not depend on the image any more (d_h0 / d_ncc / d_ncc_tm are written): with no
		 * recorded interface calls to replay it stays pending -- a reset-every-frame video loop (setImage, update, reset) would otherwise pay
		 * three device-to-host copies and a synchronisation per frame for mirrors the next re-initialisation supersedes unread */
		const bool hold = b->hold_init_pull;
		if (!b->lz.any()) b->hold_init_pull = true;
		const int rc = lazy_flush(b, false);   /* (pts = false: the current points follow the warp when an un-fused kernel next asks -- refreshing them at every setImage was one k_apply_warp launch per frame of a video loop; recorded calls that read them refresh them inside) */
		b->hold_init_pull = hold;
		if (rc) return rc;
		++b->lz.epoch;
	}
	return MTFHIP_OK;
}
#define FLUSH(b) do { if (b) { int _rc = lazy_flush(b); if (_rc) return _rc; } } while (0)
/* for entry points whose own kernels never read the current points (the AM's reductions over It / I0 / J buffers) */
#define FLUSH_AM(b) do { if (b) { int _rc = lazy_flush(b, false); if (_rc) return _rc; } } while (0)

/* MI gradient pass: rebuild the template rows instead of reading J0, under the conditions of FusedArgs::j0_recompute
 * (J0 is the search method's own template Jacobian on the current grid) */
static inline mtfhip::MiJ0Rebuild mi_j0_rebuild(const mtfhip_batch *b) {
	mtfhip::MiJ0Rebuild rb{nullptr, nullptr, nullptr, 0, 0};
	const bool ok = b->C == 1 && b->j0_is_template && b->j0_recompute_enabled && b->j0_template_corners_epoch == b->corners_epoch &&
		b->buf[MTFHIP_BUF_DI0_DX] && b->buf[MTFHIP_BUF_INIT_PTS] && (b->unit_z || b->buf[MTFHIP_BUF_INIT_Z]);
	if (ok) {
		rb.dI0 = b->buf[MTFHIP_BUF_DI0_DX]; rb.pts = b->buf[MTFHIP_BUF_INIT_PTS];
		rb.z = b->unit_z ? nullptr : b->buf[MTFHIP_BUF_INIT_Z];
		rb.hom = b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY; rb.init_variant = b->j0_variant == MTFHIP_JAC_INIT;
	}
	return rb;
}

/* ---- functions defined in one api_*.hip unit and used in another ---- */
enum { LAZY_CURR_JAC = 0, LAZY_DIFF_JAC = 1, LAZY_INIT_JAC = 2 };
int ensure_pts(mtfhip_batch *b);
const float *ensure_pair_image(mtfhip_ctx *c);   /* NULL: not applicable (borrowed / multi-channel / too large / MTFHIP_PAIR_IMAGE=0) */
const float *pair_image_if_it_pays(mtfhip_ctx *c, int n_candidates);
int set_corners_core(mtfhip_batch *b, const double *corners, bool for_track, bool defer_grid = false, bool layout_later = false);   /* api_core.hip */
void set_corners_finish_deferred(mtfhip_batch *b);
int do_update_grad_pts(mtfhip_batch *b, double grad_eps);
int gemv_to_host(mtfhip_batch *b, const double *v1, int j1, const double *v2, int j2, int sum_mode, double *g, int diff);
int ncc_template_moments(mtfhip_batch *b);
int ncc_lazy_outputs(mtfhip_batch *b, int trig, int j_a, bool hess_mean, double *g);
int ncc_hessian_from_cache(mtfhip_batch *b, int j_buf, int kind, double *H);
int fused_args(const mtfhip_batch *b, const mtfhip_sm_desc *sm, FusedArgs &fa);
int mi_blocks(const mtfhip_batch *b);
int push_ncc(mtfhip_batch *b);
int score_block_dev(mtfhip_batch *b, const double *dev_states, int lo, int cnt, double *wts, double *sim, int likelihood_func,
	double measurement_sigma, double max_similarity, const mtfhip::PfPeerPush *peer = nullptr);   /* api_fused.hip; peer: also store the
	                                                                                       weights to the other ranks' mailboxes */
} /* extern "C" */
#endif
