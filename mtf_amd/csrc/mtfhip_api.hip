/*
 * mtfhip_api.hip -- C-ABI implementation (include/mtfhip.h): handles, device memory, the tiny
 * host-side math the reference keeps on the host (4-corner DLT at initialisation, 3x3 warp algebra),
 * and the sequencing of kernels for every AppearanceModel / StateSpaceModel entry point.
 *
 * No CPU fallback exists: every entry point either runs its HIP kernels or returns an error.
 */
#include "mtfhip_internal.h"

#include <algorithm>
#include <chrono>
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

/* Kernel arguments in device memory instead of host-coherent memory: the fused kernel's first instruction is a
 * scalar load of its 400-byte argument block, and every step is two launches, so the PCIe round trip of that load is
 * 2-3 us of a 65 us step (measured: 68.6 -> 65.6 us/step at 64 targets, 22.0 -> 17.9 us at one).  The HIP runtime
 * reads the variable when it initialises (first HIP call of the process), so this has to run at load time; a value
 * already present in the environment wins. */
__attribute__((constructor)) static void mtfhip_runtime_defaults() { setenv("HIP_FORCE_DEV_KERNARG", "1", 0); }

/* ------------------------------------------------------------------ errors */
static thread_local std::string g_last_error;
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
/* 4-corner homography (utils::computeHomographyDLT, Utilities/src/warpUtils.cc:171-224): the null vector
 * of the 8x9 constraint matrix scaled to h8 = 1 is the solution of the 8x8 system below; solved by
 * Gaussian elimination with partial pivoting. corners are 2x4 interleaved. */
static bool dlt4(const double *in, const double *out, M3 &H) {
	double A[8][9];
	for (int i = 0; i < 4; ++i) {
		double x = in[2 * i], y = in[2 * i + 1], u = out[2 * i], v = out[2 * i + 1];
		double r0[9] = {x, y, 1, 0, 0, 0, -u * x, -u * y, u};
		double r1[9] = {0, 0, 0, x, y, 1, -v * x, -v * y, v};
		std::memcpy(A[2 * i], r0, sizeof(r0));
		std::memcpy(A[2 * i + 1], r1, sizeof(r1));
	}
	for (int k = 0; k < 8; ++k) {
		int piv = k;
		for (int i = k + 1; i < 8; ++i)
			if (std::fabs(A[i][k]) > std::fabs(A[piv][k])) piv = i;
		if (A[piv][k] == 0) return false;
		if (piv != k)
			for (int j = 0; j < 9; ++j) std::swap(A[piv][j], A[k][j]);
		for (int i = k + 1; i < 8; ++i) {
			double f = A[i][k] / A[k][k];
			for (int j = k; j < 9; ++j) A[i][j] -= f * A[k][j];
		}
	}
	double h[8];
	for (int k = 7; k >= 0; --k) {
		double s = A[k][8];
		for (int j = k + 1; j < 8; ++j) s -= A[k][j] * h[j];
		h[k] = s / A[k][k];
	}
	for (int i = 0; i < 8; ++i) H.m[i] = h[i];
	H.m[8] = 1;
	return true;
}

/* ------------------------------------------------------------------ handles */
struct Timer {
	std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
	double total_ms = 0;
	int n = 0;
	long launches = 0;
};

struct mtfhip_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	bool own_stream = false;
	ImgView img{nullptr, 0, 0, 0};
	float *img_owned = nullptr;
	size_t img_capacity = 0;
	unsigned char *raw = nullptr; size_t raw_capacity = 0;      /* staging of the raw frame (pre-processing) */
	float *tmp_a = nullptr, *tmp_b = nullptr; size_t tmp_capacity = 0; /* gray / row-pass intermediates */
	bool timing = false;
	int timing_stride = 1;   /* events are recorded around every timing_stride-th launch of a family */
	std::map<std::string, Timer> timers;
	std::vector<hipEvent_t> free_events;
	std::vector<struct mtfhip_batch *> batches;   /* live batches: deferred work is flushed before the image changes */
};

struct TimedScope {
	mtfhip_ctx *ctx;
	hipEvent_t a = nullptr, b = nullptr;
	Timer *tm = nullptr;
	TimedScope(mtfhip_ctx *c, const char *family) : ctx(c) {
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
		(void)hipEventRecord(a, ctx->stream);
	}
	~TimedScope() {
		if (!tm) return;
		(void)hipEventRecord(b, ctx->stream);
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
	double *d_h0inv = nullptr; /* [B][64] inverse of the constant Hessian (one-launch ICLK) */
	double *d_units = nullptr; /* per-work-unit partial sums of the LDS-staged candidate scorer */
	double *d_d2_part = nullptr, *d_d2_out = nullptr, *d_d2_w = nullptr; /* second-order term: block rows, [B][64] sums, MI self weights */
	double hess_eps = 1.0;
	bool init_pix_hess = false;
	/* J0 and dI0_dx are still exactly what init_template produced (no setter / pixel-Jacobian call touched them since): the
	 * fused kernel may then rebuild J0's rows from dI0_dx instead of reading them (MTFHIP_J0_RECOMPUTE=0 disables) */
	bool j0_is_template = false;
	long corners_epoch = 0, j0_template_corners_epoch = -1;   /* set_corners moves the grid: J0 rows depend on init_pts */
	int j0_variant = MTFHIP_JAC_WARPED;
	std::vector<double> template_corners;   /* [B][8] corners the stored J0 was computed on */
	bool j0_recompute_enabled = !(std::getenv("MTFHIP_J0_RECOMPUTE") && std::getenv("MTFHIP_J0_RECOMPUTE")[0] == '0');
	int d0_variant = MTFHIP_JAC_WARPED; /* how the template's pixel Hessian was formed (fused second-order path) */
	size_t unit_capacity = 0;
	int mi_row_len = 0;
	double mi_hist_norm = 0;
	size_t cand_capacity = 0;
	int *d_active = nullptr, *d_iters = nullptr, *d_done = nullptr;
	/* The small per-target state lives in ONE device allocation (warps | states | corners | init_corners_hm | ncc | w0 |
	 * active | iters) mirrored by two pinned staging buffers, so that set_corners and track each move it with a single
	 * copy (a grid frame used to cost 14 small copies and 4 stream syncs around a 100 us kernel). */
	char *d_slab = nullptr, *h_stage_a = nullptr, *h_stage_b = nullptr;
	hipEvent_t ev_a = nullptr, ev_b = nullptr;
	/* warp + state of every target after setState / compositionalUpdate: one copy from a pinned double buffer, no sync */
	double *h_wstage[2] = {nullptr, nullptr};
	hipEvent_t ev_w[2] = {nullptr, nullptr};
	int wflip = 0;
	/* CURR_PTS / CURR_HXY / CURR_Z lag behind the warp: only the un-fused kernels read them, so k_apply_warp runs when
	 * one of those is about to be launched (lazy_flush) or the arrays are read, not after every update */
	bool pts_stale = false;
	double *d_it_shadow = nullptr;
	double *d_ncc_tm = nullptr;   /* [B][52] NCC template moments for the device-side finish */
	/* NCC: a fused iteration updated the scalars (It_mean, a, b, f) on the host only; the un-fused kernels read d_ncc */
	bool ncc_host_newer = false;
	size_t slab_bytes = 0, slab_dbl_bytes = 0;
	/* last-workgroup-done epilogue instead of the separate k_finish_track launch: measured equal per step (84.3 vs 84.6 us at
	 * B = 64, 19.6 vs 19.2 us for one target -- the finish's dependent scalar chain is the cost, not the launch), so off */
	bool epilogue = std::getenv("MTFHIP_EPILOGUE") && std::getenv("MTFHIP_EPILOGUE")[0] == '1';
	double *h_acc = nullptr; /* pinned */
	/* Zero-copy read-back of the reduced rows: h_acc is host-coherent pinned memory the reduction kernel writes directly
	 * (h_acc_dev = its device address) followed by a sequence number in h_flag; the host spins on the flag instead of
	 * paying a copy command plus a stream synchronisation per iteration (MTFHIP_ZERO_COPY=0: copy + sync) */
	double *h_acc_dev = nullptr;
	unsigned long long *h_flag = nullptr, *h_flag_dev = nullptr, acc_seq = 0;
	int *d_fin_count = nullptr;
	int nblk_max;
	int unit_z = 1;
	/* The LDS-staged candidate scorer (template + image tile in LDS) measured 10 % SLOWER than the plain one
	 * (116 vs 105 us for 10 000 x 2 500 samples): the kernel is bound by FP64 VALU work (two IEEE divisions per
	 * sample), not by the gather path.  It stays selectable for A/B runs. */
	bool score_lds = std::getenv("MTFHIP_SCORE_LDS") != nullptr;
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
		/* Gram matrices that are already on the host: [B][36] upper triangles, valid while version matches */
		long ver[MTFHIP_BUF_COUNT] = {0};
		int gram_buf = -1; long gram_ver = -1; std::vector<double> gram;
		long gram0_ver = -1; std::vector<double> gram0;   /* J0: constant between template changes */
	} lz;

	BatchView view() const {
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

/* fixed-order sum of the per-workgroup rows -> h_acc ([B][row_len]) on the host, and wait for it */
static int read_rows(mtfhip_batch *b, int nblk, int row_len) {
	if (b->h_acc_dev) {
		const unsigned long long seq = ++b->acc_seq;
		launch_finish_host(b->d_partials, nblk, row_len, b->h_acc_dev, b->d_fin_count, b->h_flag_dev, seq, b->B, b->ctx->stream);
		const auto t0 = std::chrono::steady_clock::now();
		for (unsigned spins = 0;; ++spins) {
			if (__atomic_load_n(b->h_flag, __ATOMIC_ACQUIRE) == seq) return MTFHIP_OK;
			__builtin_ia32_pause();
			if ((spins & 0xffff) == 0xffff && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
		}
		/* the kernel did not report in: let the runtime tell why */
		HIP_TRY(hipStreamSynchronize(b->ctx->stream));
		if (__atomic_load_n(b->h_flag, __ATOMIC_ACQUIRE) == seq) return MTFHIP_OK;
		return fail(MTFHIP_ERR_HIP, "reduced rows were not delivered to host memory");
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
static int lazy_flush(mtfhip_batch *b, bool pts = true);
static int ensure_df(mtfhip_batch *b);
static int ensure_one(mtfhip_batch *b, bool curr);
static void stale_clear(mtfhip_batch *b, bool df0, bool dft);
static int protect_stale(mtfhip_batch *b, bool w0, bool wt);
static int lazy_try_similarity(mtfhip_batch *b);
static int do_cmpt_pix_jacobian(mtfhip_batch *b, int variant, int grad_buf, int dst_buf);
static int do_mean_jacobian(mtfhip_batch *b);
static int lazy_flush_ctx(mtfhip_ctx *c) {   /* called by everything that replaces the current image */
	for (mtfhip_batch *b : c->batches) { int rc = lazy_flush(b); if (rc) return rc; ++b->lz.epoch; }
	return MTFHIP_OK;
}
#define FLUSH(b) do { if (b) { int _rc = lazy_flush(b); if (_rc) return _rc; } } while (0)
/* for entry points whose own kernels never read the current points (the AM's reductions over It / I0 / J buffers) */
#define FLUSH_AM(b) do { if (b) { int _rc = lazy_flush(b, false); if (_rc) return _rc; } } while (0)

/* ------------------------------------------------------------------ context */
const char *mtfhip_last_error(void) { return g_last_error.c_str(); }

int mtfhip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int mtfhip_ctx_create(int device, void *hip_stream, mtfhip_ctx **out) {
	if (!out) return fail(MTFHIP_ERR_INVALID_ARG, "ctx_create: out is NULL");
	int n = mtfhip_device_count();
	if (n <= 0) return fail(MTFHIP_ERR_NO_DEVICE, "no HIP device visible");
	if (device < 0 || device >= n) return fail(MTFHIP_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, n);
	HIP_TRY(hipSetDevice(device));
	mtfhip_ctx *c = new mtfhip_ctx();
	c->device = device;
	if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
	else {
		hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
		if (e != hipSuccess) { delete c; return fail(MTFHIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
		c->own_stream = true;
	}
	*out = c;
	return MTFHIP_OK;
}

/* The destroy calls may run from a host-language finaliser after the HIP runtime has begun tearing itself
 * down at process exit (its calls then throw from inside the runtime); nothing may escape a C entry point. */
void mtfhip_ctx_destroy(mtfhip_ctx *c) {
	if (!c) return;
	try {
		(void)hipSetDevice(c->device);
		(void)hipStreamSynchronize(c->stream);
		for (auto &kv : c->timers)
			for (auto &p : kv.second.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
		for (auto e : c->free_events) (void)hipEventDestroy(e);
		if (c->img_owned) (void)hipFree(c->img_owned);
		if (c->raw) (void)hipFree(c->raw);
		if (c->tmp_a) (void)hipFree(c->tmp_a);
		if (c->tmp_b) (void)hipFree(c->tmp_b);
		if (c->own_stream) (void)hipStreamDestroy(c->stream);
	} catch (...) {
	}
	delete c;
}

int mtfhip_ctx_synchronize(mtfhip_ctx *c) {
	if (!c) return fail(MTFHIP_ERR_INVALID_ARG, "ctx is NULL");
	HIP_TRY(hipStreamSynchronize(c->stream));
	return MTFHIP_OK;
}
void *mtfhip_ctx_stream(mtfhip_ctx *c) { return c ? (void *)c->stream : nullptr; }

int mtfhip_image_upload(mtfhip_ctx *c, const float *host_img, int height, int width, int row_stride) {
	return mtfhip_image_upload_mc(c, host_img, height, width, row_stride, 1);
}
/* CV_32FC3: `channels` interleaved floats per pixel, row_stride in floats */
int mtfhip_image_upload_mc(mtfhip_ctx *c, const float *host_img, int height, int width, int row_stride, int channels) {
	if (!c || !host_img) return fail(MTFHIP_ERR_INVALID_ARG, "image_upload: NULL argument");
	TRY(lazy_flush_ctx(c));
	if (channels != 1 && channels != 3) return fail(MTFHIP_ERR_INVALID_ARG, "image_upload: %d channels (1 or 3 expected)", channels);
	if (height <= 0 || width <= 0 || row_stride < width * channels) return fail(MTFHIP_ERR_INVALID_ARG, "image_upload: bad shape %dx%d stride %d", height, width, row_stride);
	if ((double)height * width * channels * 4.0 >= 4294967296.0) return fail(MTFHIP_ERR_INVALID_ARG, "image_upload: %dx%dx%d floats exceed the 4 GiB a 32-bit texel offset can address", height, width, channels);
	HIP_TRY(hipSetDevice(c->device));
	const int logical_width = width;
	width *= channels;   /* floats per row */
	size_t need = (size_t)height * width;
	if (need > c->img_capacity) {
		if (c->img_owned) HIP_TRY(hipFree(c->img_owned));
		c->img_owned = nullptr;
		HIP_TRY(hipMalloc(&c->img_owned, need * sizeof(float)));
		c->img_capacity = need;
	}
	HIP_TRY(hipMemcpy2DAsync(c->img_owned, (size_t)width * sizeof(float), host_img, (size_t)row_stride * sizeof(float),
		(size_t)width * sizeof(float), (size_t)height, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream)); /* the caller may overwrite its buffer right after (TrackerBase.h:22-26) */
	c->img = ImgView{c->img_owned, height, logical_width, width, channels};
	return MTFHIP_OK;
}

int mtfhip_image_borrow(mtfhip_ctx *c, const float *dev_img, int height, int width, int row_stride) {
	if (c) TRY(lazy_flush_ctx(c));
	if (!c || !dev_img) return fail(MTFHIP_ERR_INVALID_ARG, "image_borrow: NULL argument");
	if (height <= 0 || width <= 0 || row_stride < width) return fail(MTFHIP_ERR_INVALID_ARG, "image_borrow: bad shape");
	c->img = ImgView{dev_img, height, width, row_stride};
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ pre-processing / pyramid */
static int ensure_image(mtfhip_ctx *c, int rows, int cols) {
	const size_t need = (size_t)rows * cols;
	if (need > c->img_capacity) {
		if (c->img_owned) HIP_TRY(hipFree(c->img_owned));
		c->img_owned = nullptr;
		HIP_TRY(hipMalloc(&c->img_owned, need * sizeof(float)));
		c->img_capacity = need;
	}
	return MTFHIP_OK;
}
static int ensure_tmp(mtfhip_ctx *c, size_t need) {
	if (need > c->tmp_capacity) {
		if (c->tmp_a) HIP_TRY(hipFree(c->tmp_a));
		if (c->tmp_b) HIP_TRY(hipFree(c->tmp_b));
		c->tmp_a = c->tmp_b = nullptr;
		HIP_TRY(hipMalloc(&c->tmp_a, need * sizeof(float)));
		HIP_TRY(hipMalloc(&c->tmp_b, need * sizeof(float)));
		c->tmp_capacity = need;
	}
	return MTFHIP_OK;
}
/* cv::getGaussianKernel(ksize, sigma, CV_32F) for sigma > 0: exp(-x^2 / (2 sigma^2)) rounded to float, normalised by the
 * double sum of those floats; k[0] is the centre tap, k[1], k[2] the taps one and two samples out */
static void gaussian5(double sigma, float k[3]) {
	const double scale2x = -0.5 / (sigma * sigma);
	float cf[5];
	double sum = 0;
	for (int i = 0; i < 5; ++i) { const double x = i - 2.0; cf[i] = (float)std::exp(scale2x * x * x); sum += cf[i]; }
	sum = 1. / sum;
	for (int i = 0; i < 5; ++i) cf[i] = (float)(cf[i] * sum);
	k[0] = cf[2]; k[1] = cf[3]; k[2] = cf[4];
}

int mtfhip_image_preprocess(mtfhip_ctx *c, const void *host_raw, int rows, int cols, int row_stride_bytes, int channels, int depth,
	int ksize, double sigma_x, double sigma_y) {
	if (c) TRY(lazy_flush_ctx(c));
	if (!c || !host_raw) return fail(MTFHIP_ERR_INVALID_ARG, "image_preprocess: NULL argument");
	if (rows <= 0 || cols <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "image_preprocess: bad shape %dx%d", rows, cols);
	if (channels != 1 && channels != 3) return fail(MTFHIP_ERR_INVALID_ARG, "image_preprocess: %d channels (1 or 3 expected)", channels);
	if (depth != MTFHIP_DEPTH_U8 && depth != MTFHIP_DEPTH_F32) return fail(MTFHIP_ERR_INVALID_ARG, "PreProcBase::processFrame : Invalid input image depth provided: %d", depth);
	const size_t px = (size_t)channels * (depth == MTFHIP_DEPTH_F32 ? 4 : 1);
	if ((size_t)row_stride_bytes < px * cols) return fail(MTFHIP_ERR_INVALID_ARG, "image_preprocess: row stride %d shorter than a row", row_stride_bytes);
	if (ksize != 0 && ksize != 5) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "image_preprocess: Gaussian kernel size %d (5, or 0 for no smoothing)", ksize);
	if (ksize == 5 && sigma_x <= 0) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "image_preprocess: sigma <= 0 selects OpenCV's fixed kernel table, not available");
	HIP_TRY(hipSetDevice(c->device));
	const size_t raw_bytes = px * cols * rows;
	if (raw_bytes > c->raw_capacity) {
		if (c->raw) HIP_TRY(hipFree(c->raw));
		c->raw = nullptr;
		HIP_TRY(hipMalloc(&c->raw, raw_bytes));
		c->raw_capacity = raw_bytes;
	}
	TRY(ensure_image(c, rows, cols));
	TRY(ensure_tmp(c, (size_t)rows * cols));
	HIP_TRY(hipMemcpy2DAsync(c->raw, px * cols, host_raw, (size_t)row_stride_bytes, px * cols, (size_t)rows, hipMemcpyHostToDevice, c->stream));
	{
		TimedScope ts(c, "preprocess");
		if (ksize == 0) launch_to_gray(c->raw, rows, cols, px * cols, channels, depth == MTFHIP_DEPTH_F32, c->img_owned, c->stream);
		else {
			float kx[3], ky[3];
			gaussian5(sigma_x, kx);
			gaussian5(sigma_y > 0 ? sigma_y : sigma_x, ky);   /* sigma2 <= 0 -> sigma2 = sigma1 (createGaussianKernels) */
			launch_to_gray(c->raw, rows, cols, px * cols, channels, depth == MTFHIP_DEPTH_F32, c->tmp_a, c->stream);
			launch_sym5(c->tmp_a, c->tmp_b, c->img_owned, rows, cols, kx, ky, c->stream);
		}
	}
	HIP_TRY(hipStreamSynchronize(c->stream)); /* the caller may reuse its frame buffer */
	c->img = ImgView{c->img_owned, rows, cols, cols};
	return MTFHIP_OK;
}

int mtfhip_image_pyramid_level(mtfhip_ctx *dst, mtfhip_ctx *src, int dst_rows, int dst_cols, int use_pyr_down) {
	if (dst) TRY(lazy_flush_ctx(dst));
	if (!dst || !src) return fail(MTFHIP_ERR_INVALID_ARG, "image_pyramid_level: NULL argument");
	if (!src->img.data) return fail(MTFHIP_ERR_LOGIC, "image_pyramid_level: the source context has no image");
	if (dst == src) return fail(MTFHIP_ERR_INVALID_ARG, "image_pyramid_level: source and destination contexts must differ");
	if (dst->device != src->device) return fail(MTFHIP_ERR_INVALID_ARG, "image_pyramid_level: contexts live on different devices");
	if (dst_rows <= 0 || dst_cols <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "image_pyramid_level: bad destination shape");
	if (src->img.stride != src->img.w) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "image_pyramid_level: padded source rows");
	const int sr = src->img.h, sc = src->img.w;
	if (use_pyr_down && (std::abs(dst_cols * 2 - sc) > 2 || std::abs(dst_rows * 2 - sr) > 2))   /* cv::pyrDown's own assertion */
		return fail(MTFHIP_ERR_INVALID_ARG, "pyrDown: destination %dx%d is not half of %dx%d", dst_rows, dst_cols, sr, sc);
	HIP_TRY(hipSetDevice(dst->device));
	HIP_TRY(hipStreamSynchronize(src->stream));
	TRY(ensure_image(dst, dst_rows, dst_cols));
	{
		TimedScope ts(dst, "pyramid_level");
		if (use_pyr_down) launch_pyr_down(src->img.data, sr, sc, dst->img_owned, dst_rows, dst_cols, dst->stream);
		else {   /* cv::resize + GaussianBlur(5x5, 3), PyramidalTracker.cc:93-94 */
			TRY(ensure_tmp(dst, (size_t)dst_rows * dst_cols));
			float k[3];
			gaussian5(3.0, k);
			launch_resize_linear(src->img.data, sr, sc, dst->tmp_a, dst_rows, dst_cols, dst->stream);
			launch_sym5(dst->tmp_a, dst->tmp_b, dst->img_owned, dst_rows, dst_cols, k, k, dst->stream);
		}
	}
	dst->img = ImgView{dst->img_owned, dst_rows, dst_cols, dst_cols};
	return MTFHIP_OK;
}

int mtfhip_image_download(mtfhip_ctx *c, float *host_img, int rows, int cols) {
	if (!c || !host_img) return fail(MTFHIP_ERR_INVALID_ARG, "image_download: NULL argument");
	if (!c->img.data) return fail(MTFHIP_ERR_LOGIC, "image_download: no current image");
	if (rows != c->img.h || cols != c->img.w) return fail(MTFHIP_ERR_INVALID_ARG, "image_download: the image is %dx%d", c->img.h, c->img.w);
	HIP_TRY(hipMemcpy2DAsync(host_img, (size_t)cols * sizeof(float), c->img.data, (size_t)c->img.stride * sizeof(float),
		(size_t)cols * sizeof(float), (size_t)rows, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return MTFHIP_OK;
}
int mtfhip_image_shape(mtfhip_ctx *c, int *rows, int *cols) {
	if (!c || !rows || !cols) return fail(MTFHIP_ERR_INVALID_ARG, "image_shape: NULL argument");
	*rows = c->img.h; *cols = c->img.w;
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ batch */
int mtfhip_batch_create(mtfhip_ctx *c, const mtfhip_patch_desc *d, int n_targets, mtfhip_batch **out) {
	if (!c || !d || !out) return fail(MTFHIP_ERR_INVALID_ARG, "batch_create: NULL argument");
	/* ImageBase ctor AM/src/ImageBase.cc:33-35, StateSpaceModel ctor StateSpaceModel.h:58-60 */
	if (d->resx <= 0 || d->resy <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "Invalid sampling resolution provided");
	if (n_targets <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "batch_create: n_targets must be positive");
	/* the fused kernel addresses a target's arrays with 32-bit byte offsets (ld_off / st_off): 8 columns of N doubles */
	if ((double)d->resx * d->resy * 3.0 >= (double)(1u << 26)) return fail(MTFHIP_ERR_INVALID_ARG, "batch_create: %dx%d sample points per target exceed the 2^26-row limit", d->resx, d->resy);
	if (d->grad_eps <= 0 || d->hess_eps < 0) return fail(MTFHIP_ERR_INVALID_ARG, "batch_create: grad_eps must be positive (got %g)", d->grad_eps);
	if (d->am < MTFHIP_AM_SSD || d->am > MTFHIP_AM_MI) return fail(MTFHIP_ERR_INVALID_ARG, "unknown appearance model %d", d->am);
	if (d->am == MTFHIP_AM_MI && (d->mi_n_bins < 2 || d->mi_n_bins > MI_NB)) return fail(MTFHIP_ERR_INVALID_ARG, "MI: n_bins %d outside [2, %d]", d->mi_n_bins, (int)MI_NB);
	if (d->am == MTFHIP_AM_MI && d->mi_partition_of_unity && d->mi_n_bins < 4) /* MI.cc:83-87 */
		return fail(MTFHIP_ERR_INVALID_ARG, "MI::Too few bins %d specified to enforce partition of unity constraint", d->mi_n_bins);
	if (d->ssm != MTFHIP_SSM_HOMOGRAPHY && d->ssm != MTFHIP_SSM_AFFINE) return fail(MTFHIP_ERR_INVALID_ARG, "unknown state space model %d", d->ssm);
	if (d->n_channels != 0 && d->n_channels != 1 && d->n_channels != 3) return fail(MTFHIP_ERR_INVALID_ARG, "n_channels %d (1 or 3 expected)", d->n_channels);
	HIP_TRY(hipSetDevice(c->device));
	mtfhip_batch *b = new mtfhip_batch();
	b->ctx = c; b->desc = *d; b->B = n_targets;
	b->C = d->n_channels > 1 ? d->n_channels : 1;
	b->NP = d->resx * d->resy; b->N = b->NP * b->C;   /* ImageBase: patch_size = n_pix * n_channels */
	b->S = d->ssm == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	if (d->am == MTFHIP_AM_MI) {
		/* MI ctor AM/src/MI.cc:80-94 */
		double lo = 0, hi = d->mi_n_bins - 1;
		if (d->mi_partition_of_unity) { lo = 1; hi = d->mi_n_bins - 2; }
		b->norm_mult = (hi - lo) / (255.0 - 0.0 + 1);
		b->norm_add = lo;
	}
	const size_t N = b->N, S = b->S, NP = b->NP;
	size_t per[MTFHIP_BUF_COUNT] = {N, N, 2 * N, 2 * N, N, N, N * S, N * S, N * S, 2 * NP, 2 * NP, 8 * NP, NP, NP, 2 * NP, 2 * NP,
		4 * N, 4 * N, 16 * NP, S * S * N, S * S * N, S * S * N};
	b->hess_eps = d->hess_eps > 0 ? d->hess_eps : 1.0; /* HESS_EPS, AM/include/mtf/AM/ImageBase.h:9 */
	for (int i = 0; i < MTFHIP_BUF_COUNT; ++i) { b->per_target[i] = per[i]; b->buf[i] = nullptr; }
	b->th.resize(n_targets);
	for (auto &h : b->th) { std::memset(&h, 0, sizeof(h)); h.warp = m3_identity(); }
	b->nblk_max = simple_blocks_per_target(b->N);
	int nf = fused_blocks_per_target(b->N, 1);   /* the finest decomposition is the single-target one */
	if (nf > b->nblk_max) b->nblk_max = nf;
	auto cleanup = [&](int code) { mtfhip_batch_destroy(b); return code; };
	const int eager[] = {MTFHIP_BUF_I0, MTFHIP_BUF_IT, MTFHIP_BUF_DI0_DX, MTFHIP_BUF_DIT_DX, MTFHIP_BUF_DF_DI0,
		MTFHIP_BUF_DF_DIT, MTFHIP_BUF_J0, MTFHIP_BUF_JT, MTFHIP_BUF_INIT_PTS, MTFHIP_BUF_CURR_PTS,
		MTFHIP_BUF_INIT_Z, MTFHIP_BUF_CURR_Z, MTFHIP_BUF_INIT_HXY, MTFHIP_BUF_CURR_HXY};
	for (int id : eager) { int r = ensure_buf(b, id); if (r) return cleanup(r); }
	{
		const size_t Bt = (size_t)n_targets, d = sizeof(double);
		b->slab_dbl_bytes = 54 * Bt * d;
		b->slab_bytes = b->slab_dbl_bytes + 2 * sizeof(int) * Bt;
		if (hipMalloc(&b->d_slab, b->slab_bytes) != hipSuccess || hipHostMalloc(&b->h_stage_a, b->slab_bytes) != hipSuccess ||
			hipHostMalloc(&b->h_stage_b, b->slab_bytes) != hipSuccess || hipEventCreateWithFlags(&b->ev_a, hipEventDisableTiming) != hipSuccess ||
			hipEventCreateWithFlags(&b->ev_b, hipEventDisableTiming) != hipSuccess ||
			hipHostMalloc(&b->h_wstage[0], 17 * Bt * d) != hipSuccess || hipHostMalloc(&b->h_wstage[1], 17 * Bt * d) != hipSuccess ||
			hipEventCreateWithFlags(&b->ev_w[0], hipEventDisableTiming) != hipSuccess ||
			hipEventCreateWithFlags(&b->ev_w[1], hipEventDisableTiming) != hipSuccess)
			return cleanup(fail(MTFHIP_ERR_HIP, "allocation of the per-target state slab failed"));
		double *p = reinterpret_cast<double *>(b->d_slab);
		b->d_warps = p; b->d_states = p + 9 * Bt; b->d_corners = p + 17 * Bt; b->d_init_corners_hm = p + 25 * Bt;
		b->d_ncc = p + 37 * Bt; b->d_w0 = p + 45 * Bt;
		b->d_active = reinterpret_cast<int *>(b->d_slab + b->slab_dbl_bytes); b->d_iters = b->d_active + Bt;
		(void)hipMemsetAsync(b->d_slab, 0, b->slab_bytes, c->stream);
	}
#define ALLOC(ptr, bytes) do { if (hipMalloc(&(ptr), (bytes)) != hipSuccess) return cleanup(fail(MTFHIP_ERR_HIP, "hipMalloc(%zu) failed", (size_t)(bytes))); } while (0)
	ALLOC(b->d_partials, sizeof(double) * kAccRowMax * b->nblk_max * n_targets);
	ALLOC(b->d_acc, sizeof(double) * kAccRowMax * n_targets);
	ALLOC(b->d_scratch_pts, sizeof(double) * 18 * NP * n_targets); /* largest upload: pts (2 NP) + hess_pts (16 NP) */
	ALLOC(b->d_h0, sizeof(double) * 64 * n_targets);
	ALLOC(b->d_h0inv, sizeof(double) * 64 * n_targets);
	ALLOC(b->d_colmean, sizeof(double) * 8 * n_targets);
	if (d->am == MTFHIP_AM_MI) {
		const int nb = d->mi_n_bins;
		b->mi_row_len = std::max(nb + nb * nb, 36 + nb * nb * b->S);
		/* hist_norm_mult = 1 / (patch_size + hist_pre_seed * n_bins), hist_pre_seed = n_bins * pre_seed (MI.cc:97,104) */
		b->mi_hist_norm = 1.0 / ((double)b->N + (nb * d->mi_pre_seed) * nb);
		ALLOC(b->d_mi_tb, sizeof(double) * MI_SIZE * n_targets);
		ALLOC(b->d_mi_part, sizeof(double) * (size_t)b->mi_row_len * b->nblk_max * n_targets);
		ALLOC(b->d_mi_f, sizeof(double) * n_targets);
		ALLOC(b->d_mi_H, sizeof(double) * 64 * n_targets);
		(void)hipMemsetAsync(b->d_mi_tb, 0, sizeof(double) * MI_SIZE * n_targets, c->stream);
	}
#undef ALLOC
	if (hipHostMalloc(&b->h_acc, sizeof(double) * kAccRowMax * n_targets, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
		return cleanup(fail(MTFHIP_ERR_HIP, "hipHostMalloc failed"));
	{
		const char *zc = std::getenv("MTFHIP_ZERO_COPY");
		void *dp = nullptr, *fp = nullptr;
		if (!(zc && zc[0] == '0') && hipHostGetDevicePointer(&dp, b->h_acc, 0) == hipSuccess &&
			hipHostMalloc(&b->h_flag, sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
			hipHostGetDevicePointer(&fp, b->h_flag, 0) == hipSuccess && hipMalloc(&b->d_fin_count, sizeof(int)) == hipSuccess) {
			*b->h_flag = 0;
			(void)hipMemsetAsync(b->d_fin_count, 0, sizeof(int), c->stream);
			b->h_acc_dev = static_cast<double *>(dp); b->h_flag_dev = static_cast<unsigned long long *>(fp);
		} else (void)hipGetLastError();
	}
	(void)hipMemsetAsync(b->d_partials, 0, sizeof(double) * kAccRowMax * b->nblk_max * n_targets, c->stream);
	int r = push_warps(b);
	if (r) return cleanup(r);
	{
		const char *lazy_env = std::getenv("MTFHIP_LAZY");
		b->lz.enabled = (d->am == MTFHIP_AM_SSD || d->am == MTFHIP_AM_NCC) && b->C == 1 && !(lazy_env && lazy_env[0] == '0');
	}
	c->batches.push_back(b);
	*out = b;
	return MTFHIP_OK;
}

void mtfhip_batch_destroy(mtfhip_batch *b) {
	if (!b) return;
	try {
		auto &reg = b->ctx->batches;
		reg.erase(std::remove(reg.begin(), reg.end(), b), reg.end());
		(void)hipSetDevice(b->ctx->device);
		(void)hipStreamSynchronize(b->ctx->stream);
		for (int i = 0; i < MTFHIP_BUF_COUNT; ++i)
			if (b->buf[i]) (void)hipFree(b->buf[i]);
		void *ptrs[] = {b->d_slab, b->d_partials, b->d_acc, b->d_scratch_pts, b->d_h0,
			b->d_cand, b->d_colmean, b->d_mi_tb, b->d_mi_part,
			b->d_mi_f, b->d_mi_H, b->d_h0inv, b->d_units, b->d_d2_part, b->d_d2_out, b->d_d2_w, b->d_done, b->d_it_shadow, b->d_ncc_tm};
		for (void *p : ptrs)
			if (p) (void)hipFree(p);
		if (b->h_acc) (void)hipHostFree(b->h_acc);
		if (b->h_flag) (void)hipHostFree(b->h_flag);
		if (b->d_fin_count) (void)hipFree(b->d_fin_count);
		if (b->h_stage_a) (void)hipHostFree(b->h_stage_a);
		if (b->h_stage_b) (void)hipHostFree(b->h_stage_b);
		if (b->ev_a) (void)hipEventDestroy(b->ev_a);
		if (b->ev_b) (void)hipEventDestroy(b->ev_b);
		for (int k = 0; k < 2; ++k) {
			if (b->h_wstage[k]) (void)hipHostFree(b->h_wstage[k]);
			if (b->ev_w[k]) (void)hipEventDestroy(b->ev_w[k]);
		}
	} catch (...) {
	}
	delete b;
}

int mtfhip_batch_n_targets(const mtfhip_batch *b) { return b ? b->B : 0; }
int mtfhip_batch_n_pix(const mtfhip_batch *b) { return b ? b->NP : 0; }          /* ImageBase::getNPix */
int mtfhip_batch_patch_size(const mtfhip_batch *b) { return b ? b->N : 0; }      /* ImageBase::getPatchSize = n_pix * n_channels */
int mtfhip_batch_state_size(const mtfhip_batch *b) { return b ? b->S : 0; }

int mtfhip_batch_read(mtfhip_batch *b, int id, double *dst) {
	if (!b || !dst || id < 0 || id >= MTFHIP_BUF_COUNT) return fail(MTFHIP_ERR_INVALID_ARG, "batch_read: bad argument");
	FLUSH(b);
	if (id == MTFHIP_BUF_DF_DI0 || id == MTFHIP_BUF_DF_DIT) TRY(ensure_df(b));
	if (!b->buf[id]) return fail(MTFHIP_ERR_LOGIC, "batch_read: buffer %d was never produced", id);
	if ((id == MTFHIP_BUF_IT && !b->it_valid) || (id == MTFHIP_BUF_DIT_DX && !b->dit_valid) || (id == MTFHIP_BUF_JT && !b->jt_valid))
		return fail(MTFHIP_ERR_LOGIC, "batch_read: buffer %d is not materialised (last fused iteration ran with materialize=0)", id);
	HIP_TRY(hipMemcpyAsync(dst, b->buf[id], sizeof(double) * b->per_target[id] * b->B, hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	return MTFHIP_OK;
}

int mtfhip_batch_write(mtfhip_batch *b, int id, const double *src) {
	if (!b || !src || id < 0 || id >= MTFHIP_BUF_COUNT) return fail(MTFHIP_ERR_INVALID_ARG, "batch_write: bad argument");
	FLUSH(b);
	TRY(ensure_df(b));
	if (id == MTFHIP_BUF_DF_DI0 || id == MTFHIP_BUF_DF_DIT) stale_clear(b, true, true);
	touch(b, id); ++b->lz.epoch;
	TRY(ensure_buf(b, id));
	HIP_TRY(hipMemcpyAsync(b->buf[id], src, sizeof(double) * b->per_target[id] * b->B, hipMemcpyHostToDevice, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	if (id == MTFHIP_BUF_IT) b->it_valid = true;
	if (id == MTFHIP_BUF_DIT_DX) b->dit_valid = true;
	if (id == MTFHIP_BUF_JT) b->jt_valid = true;
	if (id == MTFHIP_BUF_J0 || id == MTFHIP_BUF_DI0_DX || id == MTFHIP_BUF_INIT_PTS || id == MTFHIP_BUF_INIT_Z) b->j0_is_template = false;
	/* a caller that supplies its own homogeneous grid gets the general (non unit-z) kernels */
	if (id == MTFHIP_BUF_INIT_Z || id == MTFHIP_BUF_INIT_HXY) b->unit_z = 0;
	return MTFHIP_OK;
}

void *mtfhip_batch_device_ptr(mtfhip_batch *b, int id) {
	if (!b || id < 0 || id >= MTFHIP_BUF_COUNT) return nullptr;
	/* a raw pointer lets the caller write behind the library's back: no more deferral or host-side caches for this batch */
	if (lazy_flush(b) != MTFHIP_OK || ensure_df(b) != MTFHIP_OK) return nullptr;
	b->lz.enabled = false; b->lz.no_cache = true;
	if (ensure_buf(b, id) != MTFHIP_OK) return nullptr;
	return b->buf[id];
}

/* ------------------------------------------------------------------ SSM */
int mtfhip_ssm_set_corners(mtfhip_batch *b, const double *corners) {
	FLUSH(b);
	if (b) ++b->lz.epoch;
	if (b) { touch_all(b); b->lz.it_epoch = -1; TRY(ensure_df(b)); }
	if (!b || !corners) return fail(MTFHIP_ERR_INVALID_ARG, "set_corners: NULL argument");
	const bool hom = b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY;
	/* normalised grid extents: ProjectiveBase.cc:14 (unit square) ; Affine.cc:56-57 */
	double lo_x = -0.5, lo_y = -0.5, hi_x = 0.5, hi_y = 0.5;
	if (!hom) { lo_x = 1 - b->desc.resx / 2.0; lo_y = 1 - b->desc.resy / 2.0; hi_x = b->desc.resx / 2.0; hi_y = b->desc.resy / 2.0; }
	const double nc[8] = {lo_x, lo_y, hi_x, lo_y, hi_x, hi_y, lo_x, hi_y};
	std::vector<double> w0(9 * b->B);
	int unit_z = 1;
	for (int t = 0; t < b->B; ++t) {
		M3 W0;
		if (!dlt4(nc, corners + 8 * t, W0)) return fail(MTFHIP_ERR_INVALID_ARG, "set_corners: degenerate corners for target %d", t);
		if (!hom || (std::fabs(W0.m[6]) < 1e-15 && std::fabs(W0.m[7]) < 1e-15)) {
			if (hom) { W0.m[6] = 0; W0.m[7] = 0; }
		} else unit_z = 0;
		std::memcpy(&w0[9 * t], W0.m, sizeof(double) * 9);
		TargetHost &h = b->th[t];
		std::memcpy(h.corners, corners + 8 * t, sizeof(double) * 8);
		std::memcpy(h.init_corners, corners + 8 * t, sizeof(double) * 8);
		for (int q = 0; q < 4; ++q) {
			h.init_corners_hm[3 * q] = corners[8 * t + 2 * q];
			h.init_corners_hm[3 * q + 1] = corners[8 * t + 2 * q + 1];
			h.init_corners_hm[3 * q + 2] = 1;
		}
		h.warp = m3_identity();
		std::memset(h.state, 0, sizeof(h.state));
	}
	b->unit_z = hom ? unit_z : 1;
	/* w0, init_corners_hm, identity warps, zero states and the corners in ONE pinned async copy; the staging buffer is
	 * protected by an event instead of a stream sync */
	HIP_TRY(hipEventSynchronize(b->ev_a));
	fill_stage(b, b->h_stage_a, w0.data(), 0, false);
	HIP_TRY(hipMemcpyAsync(b->d_slab, b->h_stage_a, b->slab_dbl_bytes, hipMemcpyHostToDevice, b->ctx->stream));
	HIP_TRY(hipEventRecord(b->ev_a, b->ctx->stream));
	{
		TimedScope ts(b->ctx, "init_grid");
		launch_init_grid(b->view(), b->d_w0, b->desc.resx, b->desc.resy, lo_x, lo_y, hi_x, hi_y, hom ? 0 : 1, b->ctx->stream);
	}
	b->have_corners = true;
	b->pts_stale = false;   /* k_init_grid writes the current points too */
	++b->corners_epoch;
	return MTFHIP_OK;
}

static int ensure_pts(mtfhip_batch *b) {
	if (!b->pts_stale) return MTFHIP_OK;
	b->pts_stale = false;
	TimedScope ts(b->ctx, "apply_warp");
	launch_apply_warp(b->view(), b->ctx->stream);
	return MTFHIP_OK;
}
static int apply_states(mtfhip_batch *b) {
	TRY(push_warps(b));
	b->pts_stale = true;
	if (!b->lz.enabled) return ensure_pts(b);
	return MTFHIP_OK;
}

int mtfhip_ssm_set_state(mtfhip_batch *b, const double *states) {
	FLUSH(b);
	if (b) ++b->lz.epoch;
	if (!b || !states) return fail(MTFHIP_ERR_INVALID_ARG, "set_state: NULL argument");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "set_state before set_corners");
	for (int t = 0; t < b->B; ++t) {
		TargetHost &h = b->th[t];
		std::memset(h.state, 0, sizeof(h.state));
		std::memcpy(h.state, states + (size_t)t * b->S, sizeof(double) * b->S);
		h.warp = warp_from_state(b->desc.ssm, h.state);
		update_corners(b, t);
	}
	return apply_states(b);
}

int mtfhip_ssm_compositional_update(mtfhip_batch *b, const double *dps) {
	FLUSH(b);
	if (b) ++b->lz.epoch;
	if (!b || !dps) return fail(MTFHIP_ERR_INVALID_ARG, "compositional_update: NULL argument");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "compositional_update before set_corners");
	for (int t = 0; t < b->B; ++t) {
		TargetHost &h = b->th[t];
		double dp[8] = {0};
		std::memcpy(dp, dps + (size_t)t * b->S, sizeof(double) * b->S);
		M3 upd = warp_from_state(b->desc.ssm, dp);
		h.warp = m3_mul(h.warp, upd);
		if (b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY) {
			double s = h.warp.m[8];
			for (int i = 0; i < 9; ++i) h.warp.m[i] /= s;
		}
		state_from_warp(b->desc.ssm, h.state, h.warp);
		update_corners(b, t);
	}
	return apply_states(b);
}

int mtfhip_ssm_invert_state(mtfhip_batch *b, const double *states, double *inv_states) {
	if (!b || !states || !inv_states) return fail(MTFHIP_ERR_INVALID_ARG, "invert_state: NULL argument");
	for (int t = 0; t < b->B; ++t) {
		double p[8] = {0}, q[8];
		std::memcpy(p, states + (size_t)t * b->S, sizeof(double) * b->S);
		M3 Wi = m3_inverse(warp_from_state(b->desc.ssm, p));
		double s = Wi.m[8];
		for (int i = 0; i < 9; ++i) Wi.m[i] /= s;
		state_from_warp(b->desc.ssm, q, Wi);
		std::memcpy(inv_states + (size_t)t * b->S, q, sizeof(double) * b->S);
	}
	return MTFHIP_OK;
}

static int do_update_grad_pts(mtfhip_batch *b, double grad_eps) {
	TRY(ensure_buf(b, MTFHIP_BUF_GRAD_PTS));
	TimedScope ts(b->ctx, "grad_pts");
	launch_grad_pts(b->view(), grad_eps, b->ctx->stream);
	return MTFHIP_OK;
}
int mtfhip_ssm_update_grad_pts(mtfhip_batch *b, double grad_eps) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_grad_pts: NULL batch");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "update_grad_pts before set_corners");
	if (b->lz.enabled && grad_eps == b->desc.grad_eps) {
		if (b->lz.gp || b->lz.pg) FLUSH(b);
		b->lz.gp = ++b->lz.seq;
		return MTFHIP_OK;
	}
	FLUSH(b);
	return do_update_grad_pts(b, grad_eps);
}

static int do_cmpt_pix_jacobian(mtfhip_batch *b, int variant, int grad_buf, int dst_buf) {
	TRY(ensure_buf(b, dst_buf));
	TimedScope ts(b->ctx, "pix_jacobian");
	launch_pix_jacobian(b->view(), variant, b->buf[grad_buf], b->buf[dst_buf], b->ctx->stream);
	touch(b, dst_buf);
	if (dst_buf == MTFHIP_BUF_JT) b->jt_valid = true;
	if (dst_buf == MTFHIP_BUF_J0) b->j0_is_template = false;
	return MTFHIP_OK;
}
int mtfhip_ssm_cmpt_pix_jacobian(mtfhip_batch *b, int variant, int grad_buf, int dst_buf) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_pix_jacobian: NULL batch");
	if (variant < MTFHIP_JAC_INIT || variant > MTFHIP_JAC_APPROX) return fail(MTFHIP_ERR_INVALID_ARG, "unknown Jacobian variant %d", variant);
	if (grad_buf != MTFHIP_BUF_DI0_DX && grad_buf != MTFHIP_BUF_DIT_DX) return fail(MTFHIP_ERR_INVALID_ARG, "grad_buf must be DI0_DX or DIT_DX");
	if (!j_buf_ok(dst_buf)) return fail(MTFHIP_ERR_INVALID_ARG, "dst_buf must be J0, JT or JM");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "cmpt_pix_jacobian before set_corners");
	if (b->lz.enabled && grad_buf == MTFHIP_BUF_DIT_DX && dst_buf == MTFHIP_BUF_JT &&
		(variant == MTFHIP_JAC_WARPED || variant == MTFHIP_JAC_INIT)) {
		if (b->lz.pj || b->lz.jm) FLUSH(b);
		TRY(ensure_buf(b, dst_buf));
		b->lz.pj = ++b->lz.seq; b->lz.pj_variant = variant;
		b->jt_valid = true;
		return MTFHIP_OK;
	}
	FLUSH(b);
	return do_cmpt_pix_jacobian(b, variant, grad_buf, dst_buf);
}

int mtfhip_ssm_get_corners(mtfhip_batch *b, double *corners) {
	if (!b || !corners) return fail(MTFHIP_ERR_INVALID_ARG, "get_corners: NULL argument");
	for (int t = 0; t < b->B; ++t) std::memcpy(corners + 8 * t, b->th[t].corners, sizeof(double) * 8);
	return MTFHIP_OK;
}
int mtfhip_ssm_get_init_corners(mtfhip_batch *b, double *corners) {
	if (!b || !corners) return fail(MTFHIP_ERR_INVALID_ARG, "get_init_corners: NULL argument");
	for (int t = 0; t < b->B; ++t) std::memcpy(corners + 8 * t, b->th[t].init_corners, sizeof(double) * 8);
	return MTFHIP_OK;
}
int mtfhip_ssm_get_state(mtfhip_batch *b, double *states) {
	if (!b || !states) return fail(MTFHIP_ERR_INVALID_ARG, "get_state: NULL argument");
	for (int t = 0; t < b->B; ++t) std::memcpy(states + (size_t)t * b->S, b->th[t].state, sizeof(double) * b->S);
	return MTFHIP_OK;
}
int mtfhip_ssm_get_warp(mtfhip_batch *b, double *warps) {
	if (!b || !warps) return fail(MTFHIP_ERR_INVALID_ARG, "get_warp: NULL argument");
	for (int t = 0; t < b->B; ++t) std::memcpy(warps + 9 * t, b->th[t].warp.m, sizeof(double) * 9);
	return MTFHIP_OK;
}
int mtfhip_ssm_apply_warp_to_corners(mtfhip_batch *b, const double *in_corners, const double *states, double *out_corners) {
	if (!b || !in_corners || !states || !out_corners) return fail(MTFHIP_ERR_INVALID_ARG, "apply_warp_to_corners: NULL argument");
	for (int t = 0; t < b->B; ++t) {
		double p[8] = {0};
		std::memcpy(p, states + (size_t)t * b->S, sizeof(double) * b->S);
		M3 W = warp_from_state(b->desc.ssm, p);
		for (int q = 0; q < 4; ++q) {
			double x = in_corners[8 * t + 2 * q], y = in_corners[8 * t + 2 * q + 1];
			double nx = W.m[0] * x + W.m[1] * y + W.m[2], ny = W.m[3] * x + W.m[4] * y + W.m[5];
			if (b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY) {
				double d = W.m[6] * x + W.m[7] * y + W.m[8];
				nx = nx / d; ny = ny / d;
			}
			out_corners[8 * t + 2 * q] = nx; out_corners[8 * t + 2 * q + 1] = ny;
		}
	}
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ ImageBase */
int mtfhip_am_initialize_pix_vals(mtfhip_batch *b, const double *pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_pix_vals: NULL batch");
	TRY(need_image(b));
	const double *dp;
	TRY(resolve_pts(b, pts, MTFHIP_BUF_CURR_PTS, 2 * (size_t)b->NP, &dp));
	{
		TimedScope ts(b->ctx, "sample");
		launch_sample(b->view(), b->ctx->img, dp, b->buf[MTFHIP_BUF_I0], b->norm_mult, b->norm_add, b->ctx->stream);
	}
	if (!b->init_pix_vals) {
		HIP_TRY(hipMemcpyAsync(b->buf[MTFHIP_BUF_IT], b->buf[MTFHIP_BUF_I0], sizeof(double) * b->N * b->B, hipMemcpyDeviceToDevice, b->ctx->stream));
		b->init_pix_vals = true;
		b->it_valid = true;
	}
	return MTFHIP_OK;
}
static int do_update_pix_vals(mtfhip_batch *b, const double *pts) {
	TRY(need_image(b));
	TRY(protect_stale(b, false, false));   /* IT is about to change: gradients skipped by a fused launch keep the old IT */
	const double *dp;
	TRY(resolve_pts(b, pts, MTFHIP_BUF_CURR_PTS, 2 * (size_t)b->NP, &dp));
	TimedScope ts(b->ctx, "sample");
	launch_sample(b->view(), b->ctx->img, dp, b->buf[MTFHIP_BUF_IT], b->norm_mult, b->norm_add, b->ctx->stream);
	touch(b, MTFHIP_BUF_IT);
	b->lz.it_epoch = pts ? -1 : b->lz.epoch;
	b->it_valid = true;
	return MTFHIP_OK;
}
int mtfhip_am_update_pix_vals(mtfhip_batch *b, const double *pts) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_pix_vals: NULL batch");
	if (b->lz.enabled && !pts && b->init_pix_vals && b->have_corners && b->ctx->img.data && b->ctx->img.channels == 1) {
		if (b->lz.pv || b->lz.sim) FLUSH(b);
		b->lz.pv = ++b->lz.seq;
		b->it_valid = true;
		return MTFHIP_OK;
	}
	FLUSH(b);
	return do_update_pix_vals(b, pts);
}
static int pix_grad_common(mtfhip_batch *b, const double *pts, bool warped, bool init) {
	TRY(need_image(b));
	const double *dp;
	if (warped) TRY(resolve_pts(b, pts, MTFHIP_BUF_GRAD_PTS, 8 * (size_t)b->NP, &dp));
	else TRY(resolve_pts(b, pts, MTFHIP_BUF_CURR_PTS, 2 * (size_t)b->NP, &dp));
	double *dst = b->buf[init ? MTFHIP_BUF_DI0_DX : MTFHIP_BUF_DIT_DX];
	{
		TimedScope ts(b->ctx, warped ? "warped_img_grad" : "img_grad");
		if (warped) launch_warped_img_grad(b->view(), b->ctx->img, dp, dst, b->desc.grad_eps, b->norm_mult, b->ctx->stream);
		else launch_img_grad(b->view(), b->ctx->img, dp, dst, b->desc.grad_eps, b->norm_mult, b->ctx->stream);
	}
	touch(b, init ? MTFHIP_BUF_DI0_DX : MTFHIP_BUF_DIT_DX);
	if (init) b->j0_is_template = false;
	if (init && !b->init_pix_grad) {
		HIP_TRY(hipMemcpyAsync(b->buf[MTFHIP_BUF_DIT_DX], b->buf[MTFHIP_BUF_DI0_DX], sizeof(double) * 2 * b->N * b->B, hipMemcpyDeviceToDevice, b->ctx->stream));
		b->init_pix_grad = true;
		b->dit_valid = true;
	}
	if (!init) b->dit_valid = true;
	return MTFHIP_OK;
}
int mtfhip_am_initialize_pix_grad(mtfhip_batch *b, const double *pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_pix_grad: NULL batch");
	return pix_grad_common(b, pts, false, true);
}
static int lazy_record_pix_grad(mtfhip_batch *b, int kind) {
	if (b->lz.pg || b->lz.pj) FLUSH(b);
	b->lz.pg = ++b->lz.seq; b->lz.pg_kind = kind;
	b->dit_valid = true;
	return MTFHIP_OK;
}
int mtfhip_am_update_pix_grad(mtfhip_batch *b, const double *pts) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_pix_grad: NULL batch");
	if (b->lz.enabled && !pts && b->have_corners && b->ctx->img.data && b->ctx->img.channels == 1)
		return lazy_record_pix_grad(b, 1);
	FLUSH(b);
	return pix_grad_common(b, pts, false, false);
}
int mtfhip_am_initialize_pix_grad_warped(mtfhip_batch *b, const double *gp) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_pix_grad_warped: NULL batch");
	return pix_grad_common(b, gp, true, true);
}
int mtfhip_am_update_pix_grad_warped(mtfhip_batch *b, const double *gp) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_pix_grad_warped: NULL batch");
	/* only after a deferred update_grad_pts: the fused kernel derives the warped gradient points from the current warp */
	if (b->lz.enabled && !gp && b->lz.gp && b->ctx->img.data && b->ctx->img.channels == 1)
		return lazy_record_pix_grad(b, 2);
	FLUSH(b);
	return pix_grad_common(b, gp, true, false);
}


/* ------------------------------------------------------------------ NCC (AM/src/NCC.cc) */
static int push_ncc(mtfhip_batch *b) {
	std::vector<double> s(8 * (size_t)b->B, 0.0);
	for (int t = 0; t < b->B; ++t) {
		const TargetHost &h = b->th[t];
		double *p = &s[8 * t];
		p[0] = h.I0_mean; p[1] = h.c; p[2] = h.It_mean; p[3] = h.b; p[4] = h.f; p[5] = h.gmean;
	}
	HIP_TRY(hipMemcpyAsync(b->d_ncc, s.data(), sizeof(double) * s.size(), hipMemcpyHostToDevice, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	return MTFHIP_OK;
}
static int ncc_mean_of(mtfhip_batch *b, int buf, double TargetHost::*dst) {
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "ncc_stats");
		launch_vec_sum(b->view(), b->buf[buf], b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) b->th[t].*dst = b->h_acc[(size_t)t * ACC_COUNT + ACC_RR] / (double)b->N;
	return MTFHIP_OK;
}
/* NCC::initializeSimilarity NCC.cc:55-95 */
static int ncc_initialize_similarity(mtfhip_batch *b) {
	TRY(ncc_mean_of(b, MTFHIP_BUF_I0, &TargetHost::I0_mean));
	if (!b->init_sim)
		for (auto &h : b->th) h.It_mean = h.I0_mean;
	TRY(push_ncc(b));
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "ncc_stats");
		launch_ncc_centered(b->view(), b->d_ncc, b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) {
		TargetHost &h = b->th[t];
		h.c = std::sqrt(b->h_acc[(size_t)t * ACC_COUNT + ACC_G + 2]);
		if (!b->init_sim) { h.f = 1; h.b = h.c; }
	}
	b->init_sim = true;
	return push_ncc(b);
}
/* NCC::updateSimilarity NCC.cc:124-161 */
static int ncc_update_similarity(mtfhip_batch *b) {
	TRY(ncc_mean_of(b, MTFHIP_BUF_IT, &TargetHost::It_mean));
	TRY(push_ncc(b));
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "ncc_stats");
		launch_ncc_centered(b->view(), b->d_ncc, b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) {
		TargetHost &h = b->th[t];
		h.a = b->h_acc[(size_t)t * ACC_COUNT + ACC_G + 0];
		h.b = std::sqrt(b->h_acc[(size_t)t * ACC_COUNT + ACC_G + 1]);
		double bc = h.b * h.c;
		h.f = h.a / bc;
	}
	return push_ncc(b);
}
/* NCC::updateCurrGrad / updateInitGrad NCC.cc:163-234 */
static int ncc_update_grad(mtfhip_batch *b, int curr) {
	if (b->ncc_host_newer) { TRY(push_ncc(b)); b->ncc_host_newer = false; }
	int nblk = simple_blocks_per_target(b->N);
	double *dst = b->buf[curr ? MTFHIP_BUF_DF_DIT : MTFHIP_BUF_DF_DI0];
	{
		TimedScope ts(b->ctx, "ncc_grad");
		launch_ncc_grad(b->view(), b->d_ncc, curr, dst, b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) b->th[t].gmean = b->h_acc[(size_t)t * ACC_COUNT + ACC_RR] / (double)b->N;
	TRY(push_ncc(b));
	TimedScope ts(b->ctx, "ncc_grad");
	launch_sub_mean(b->view(), dst, b->d_ncc, b->ctx->stream);
	return MTFHIP_OK;
}
/* NCC::cmptInitHessian / cmptCurrHessian / cmptSelfHessian NCC.cc:282-389 (fast_hess = 0);
 * kind 0 init, 1 curr, 2 self.  H is column-major S x S per target. */
static int ncc_hessian_from_cache(mtfhip_batch *b, int j_buf, int kind, double *H);
static int ncc_hessian(mtfhip_batch *b, int j_buf, int kind, double *H) {
	if (ncc_hessian_from_cache(b, j_buf, kind, H)) return MTFHIP_OK;
	if (b->ncc_host_newer) { TRY(push_ncc(b)); b->ncc_host_newer = false; }
	const int S = b->S;
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "ncc_hess");
		launch_col_sum(b->view(), b->buf[j_buf], b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	std::vector<double> cm(8 * (size_t)b->B, 0.0);
	for (int t = 0; t < b->B; ++t)
		for (int s = 0; s < S; ++s) cm[8 * t + s] = b->h_acc[(size_t)t * ACC_COUNT + ACC_G + s] / (double)b->N;
	HIP_TRY(hipMemcpyAsync(b->d_colmean, cm.data(), sizeof(double) * cm.size(), hipMemcpyHostToDevice, b->ctx->stream));
	TRY(push_ncc(b));
	{
		TimedScope ts(b->ctx, "ncc_hess");
		launch_ncc_hess(b->view(), b->d_ncc, b->d_colmean, b->buf[j_buf], b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) {
		const double *acc = b->h_acc + (size_t)t * ACC_COUNT;
		const double f = b->th[t].f;
		double *Ht = H + (size_t)t * S * S;
		int k = 0;
		for (int r = 0; r < 8; ++r)
			for (int c = r; c < 8; ++c) {
				if (r < S && c < S) {
					const double G = acc[ACC_H + k];
					const double ut_r = acc[ACC_G + r], ut_c = acc[ACC_G + c];
					const double u0_r = acc[ACC_G2 + r], u0_c = acc[ACC_G2 + c];
					double v;
					if (kind == 0) v = -f * G - ut_r * u0_c - u0_r * ut_c + 3 * u0_r * u0_c;
					else if (kind == 1) v = -f * G - ut_r * u0_c - u0_r * ut_c + 3 * ut_r * ut_c;
					else v = -G + ut_r * ut_c;
					Ht[c * S + r] = v; Ht[r * S + c] = v;
				}
				++k;
			}
	}
	return MTFHIP_OK;
}


/* ------------------------------------------------------------------ MI (AM/src/MI.cc) */
static int mi_read_f(mtfhip_batch *b) {
	std::vector<double> f(b->B);
	HIP_TRY(hipMemcpyAsync(f.data(), b->d_mi_f, sizeof(double) * b->B, hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	for (int t = 0; t < b->B; ++t) b->th[t].f = f[t];
	return MTFHIP_OK;
}
/* workgroups per target for the MI histogram / Hessian passes: enough to fill the chip (~4 per CU over the batch), few
 * enough that every wave amortises its register-resident bin accumulators over many 64-pixel chunks and that the
 * fixed-order finish has short columns to add */
static int mi_blocks(const mtfhip_batch *b) {
	int nb = 1024 / b->B;
	if (nb < 1) nb = 1;
	return std::min(nb, simple_blocks_per_target(b->N));
}
/* mode 0 initialise (A = B = I0), 1 update (A = It, B = I0), 2 self (A = B = It) */
static int mi_hist_pass(mtfhip_batch *b, int mode, int first_init) {
	const int nb = b->desc.mi_n_bins, nblk = mi_blocks(b);
	const double *A = b->buf[mode == 0 ? MTFHIP_BUF_I0 : MTFHIP_BUF_IT];
	const double *Bv = b->buf[mode == 2 ? MTFHIP_BUF_IT : MTFHIP_BUF_I0];
	TimedScope ts(b->ctx, "mi_hist");
	launch_mi_hist(b->view(), nb, b->mi_hist_norm, A, Bv, b->d_mi_part, nblk, b->mi_row_len, b->ctx->stream);
	launch_mi_hist_finish(b->view(), nb, b->desc.mi_pre_seed, b->mi_hist_norm, mode, first_init, b->d_mi_part, nblk,
		b->mi_row_len, b->d_mi_tb, b->d_mi_f, b->ctx->stream);
	return MTFHIP_OK;
}
/* kind 0 init (MI.cc:461-513), 1 curr (:603-637), 2 self (:515-601, the returned second pass) */
static int mi_hessian(mtfhip_batch *b, int j_buf, int kind, double *H) {
	const int nb = b->desc.mi_n_bins, nblk = mi_blocks(b), S = b->S;
	if (kind == 2) TRY(mi_hist_pass(b, 2, 0));   /* cmptSelfHist MI.cc:639-659 */
	const double *A = b->buf[kind == 0 ? MTFHIP_BUF_I0 : MTFHIP_BUF_IT];
	const double *Bv = b->buf[kind == 1 ? MTFHIP_BUF_I0 : MTFHIP_BUF_IT];
	const int table = kind == 0 ? MI_T_INIT : (kind == 1 ? MI_T_CURR : MI_T_SELF);
	const int joint = kind == 2 ? MI_SELF_JOINT : MI_JOINT;
	const int hist = kind == 0 ? MI_HIST_INIT : MI_HIST_CURR;
	{
		TimedScope ts(b->ctx, "mi_hess");
		launch_mi_hess(b->view(), nb, b->mi_hist_norm, A, Bv, b->d_mi_tb, table, kind == 0, b->buf[j_buf], b->d_mi_part, nblk,
			b->mi_row_len, b->ctx->stream);
		launch_mi_hess_finish(b->view(), nb, b->d_mi_part, nblk, b->mi_row_len, b->d_mi_tb, joint, hist, kind == 0, b->d_mi_H,
			b->ctx->stream);
	}
	std::vector<double> h(64 * (size_t)b->B);
	HIP_TRY(hipMemcpyAsync(h.data(), b->d_mi_H, sizeof(double) * h.size(), hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	for (int t = 0; t < b->B; ++t) std::memcpy(H + (size_t)t * S * S, &h[64 * t], sizeof(double) * S * S);
	return MTFHIP_OK;
}
static int mi_grad(mtfhip_batch *b, int curr) {
	const int nb = b->desc.mi_n_bins;
	TimedScope ts(b->ctx, "mi_grad");
	launch_mi_factor(b->view(), nb, curr, b->d_mi_tb, b->ctx->stream);
	if (curr) launch_mi_grad(b->view(), nb, b->mi_hist_norm, b->buf[MTFHIP_BUF_IT], b->buf[MTFHIP_BUF_I0], b->d_mi_tb, MI_T_CURR,
		b->buf[MTFHIP_BUF_DF_DIT], b->ctx->stream);
	else launch_mi_grad(b->view(), nb, b->mi_hist_norm, b->buf[MTFHIP_BUF_I0], b->buf[MTFHIP_BUF_IT], b->d_mi_tb, MI_T_INIT,
		b->buf[MTFHIP_BUF_DF_DI0], b->ctx->stream);
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ AppearanceModel */
static int am_supported(mtfhip_batch *b, const char *fn) {
	if (b->desc.am == MTFHIP_AM_SSD || b->desc.am == MTFHIP_AM_NCC || b->desc.am == MTFHIP_AM_MI) return MTFHIP_OK;
	return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "%s :: appearance model %d is not available on the device path yet", fn, b->desc.am);
}

int mtfhip_am_initialize_similarity(mtfhip_batch *b) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_similarity: NULL batch");
	TRY(am_supported(b, "initializeSimilarity"));
	if (b->desc.am == MTFHIP_AM_NCC) return ncc_initialize_similarity(b);
	if (b->desc.am == MTFHIP_AM_MI) {
		/* MI::initializeSimilarity MI.cc:207-287 */
		const int first = b->init_sim ? 0 : 1;
		TRY(mi_hist_pass(b, 0, first));
		if (first) TRY(mi_read_f(b));
		b->init_sim = true;
		return MTFHIP_OK;
	}
	if (b->init_sim) return MTFHIP_OK;
	HIP_TRY(hipMemsetAsync(b->buf[MTFHIP_BUF_DF_DI0], 0, sizeof(double) * b->N * b->B, b->ctx->stream));
	for (auto &h : b->th) h.f = 0;
	b->init_sim = true;
	return MTFHIP_OK;
}
int mtfhip_am_initialize_grad(mtfhip_batch *b) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_grad: NULL batch");
	TRY(am_supported(b, "initializeGrad"));
	if (b->desc.am == MTFHIP_AM_MI) {
		/* MI::initializeGrad MI.cc:299-332: df_dI0 from the initial tables, df_dIt = df_dI0 */
		if (b->init_grad) return MTFHIP_OK;
		{
			TimedScope ts(b->ctx, "mi_grad");
			launch_mi_grad(b->view(), b->desc.mi_n_bins, b->mi_hist_norm, b->buf[MTFHIP_BUF_I0], b->buf[MTFHIP_BUF_I0], b->d_mi_tb,
				MI_T_INIT, b->buf[MTFHIP_BUF_DF_DI0], b->ctx->stream);
		}
		HIP_TRY(hipMemcpyAsync(b->buf[MTFHIP_BUF_DF_DIT], b->buf[MTFHIP_BUF_DF_DI0], sizeof(double) * b->N * b->B, hipMemcpyDeviceToDevice, b->ctx->stream));
		b->init_grad = true;
		return MTFHIP_OK;
	}
	if (b->desc.am == MTFHIP_AM_NCC) {
		/* NCC::initializeGrad NCC.cc:97-122: gradient vectors start at zero */
		if (!b->init_grad) {
			HIP_TRY(hipMemsetAsync(b->buf[MTFHIP_BUF_DF_DI0], 0, sizeof(double) * b->N * b->B, b->ctx->stream));
			HIP_TRY(hipMemsetAsync(b->buf[MTFHIP_BUF_DF_DIT], 0, sizeof(double) * b->N * b->B, b->ctx->stream));
			b->init_grad = true;
		}
		return MTFHIP_OK;
	}
	if (b->init_grad) return MTFHIP_OK;
	HIP_TRY(hipMemcpyAsync(b->buf[MTFHIP_BUF_DF_DIT], b->buf[MTFHIP_BUF_DF_DI0], sizeof(double) * b->N * b->B, hipMemcpyDeviceToDevice, b->ctx->stream));
	b->init_grad = true;
	return MTFHIP_OK;
}
int mtfhip_am_initialize_hess(mtfhip_batch *b) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_hess: NULL batch");
	return am_supported(b, "initializeHess");
}
static int do_update_similarity(mtfhip_batch *b, int prereq_only);
int mtfhip_am_update_similarity(mtfhip_batch *b, int prereq_only) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_similarity: NULL batch");
	TRY(am_supported(b, "updateSimilarity"));
	if (!b->init_sim) return fail(MTFHIP_ERR_LOGIC, "updateSimilarity before initializeSimilarity");
	if (b->lz.enabled && b->lz.pv) {   /* only behind a deferred updatePixVals: otherwise nothing to fuse with */
		if (b->lz.sim || b->lz.cg) FLUSH(b);
		if (b->lz.pv) {
			b->lz.sim = ++b->lz.seq; b->lz.sim_need_f = !prereq_only;
			return MTFHIP_OK;
		}
	}
	FLUSH(b);
	return do_update_similarity(b, prereq_only);
}
static int do_update_similarity(mtfhip_batch *b, int prereq_only) {
	if (b->desc.am == MTFHIP_AM_NCC) { int rc = ncc_update_similarity(b); b->lz.df0_it_ver = b->lz.ver[MTFHIP_BUF_IT]; return rc; }
	if (b->desc.am == MTFHIP_AM_MI) {
		/* MI::updateSimilarity MI.cc:346-382 */
		TRY(mi_hist_pass(b, 1, 0));
		if (!prereq_only) TRY(mi_read_f(b));
		return MTFHIP_OK;
	}
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "ssd_residual");
		launch_ssd_residual(b->view(), b->d_partials, nblk, b->ctx->stream);
	}
	stale_clear(b, true, false);
	b->lz.df0_it_ver = b->lz.ver[MTFHIP_BUF_IT];
	if (prereq_only) return MTFHIP_OK;
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) b->th[t].f = -b->h_acc[(size_t)t * ACC_COUNT + ACC_RR] / 2;
	return MTFHIP_OK;
}
static int do_update_curr_grad(mtfhip_batch *b) {
	if (b->desc.am == MTFHIP_AM_NCC) { int rc = ncc_update_grad(b, 1); stale_clear(b, false, true); return rc; }
	if (b->desc.am == MTFHIP_AM_MI) return mi_grad(b, 1);
	if (b->lz.df0_stale) TRY(ensure_one(b, false));
	TimedScope ts(b->ctx, "negate");
	launch_negate(b->buf[MTFHIP_BUF_DF_DI0], b->buf[MTFHIP_BUF_DF_DIT], (size_t)b->N * b->B, b->ctx->stream);
	stale_clear(b, false, true);
	return MTFHIP_OK;
}
int mtfhip_am_update_curr_grad(mtfhip_batch *b) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_curr_grad: NULL batch");
	TRY(am_supported(b, "updateCurrGrad"));
	if (b->lz.enabled) {
		if (b->lz.cg) FLUSH(b);
		b->lz.cg = ++b->lz.seq;
		return MTFHIP_OK;
	}
	FLUSH(b);
	return do_update_curr_grad(b);
}
int mtfhip_am_update_init_grad(mtfhip_batch *b) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_init_grad: NULL batch");
	TRY(am_supported(b, "updateInitGrad"));
	if (b->desc.am == MTFHIP_AM_SSD) return MTFHIP_OK;   /* SSD::updateInitGrad is empty: df_dI0 is updateSimilarity's residual */
	if (b->lz.enabled) {   /* NCC */
		if (b->lz.ig) FLUSH(b);
		b->lz.ig = ++b->lz.seq;
		return MTFHIP_OK;
	}
	FLUSH(b);
	if (b->desc.am == MTFHIP_AM_NCC) { int rc = ncc_update_grad(b, 0); stale_clear(b, true, false); return rc; }
	if (b->desc.am == MTFHIP_AM_MI) return mi_grad(b, 0);
	return MTFHIP_OK;
}
int mtfhip_am_get_similarity(mtfhip_batch *b, double *f) {
	if (!b || !f) return fail(MTFHIP_ERR_INVALID_ARG, "get_similarity: NULL argument");
	TRY(lazy_try_similarity(b));
	FLUSH_AM(b);
	for (int t = 0; t < b->B; ++t) f[t] = b->th[t].f;
	return MTFHIP_OK;
}
int mtfhip_am_get_likelihood(mtfhip_batch *b, double *l) {
	if (!b || !l) return fail(MTFHIP_ERR_INVALID_ARG, "get_likelihood: NULL argument");
	TRY(lazy_try_similarity(b));
	FLUSH_AM(b);
	for (int t = 0; t < b->B; ++t) {
		double f = b->th[t].f;
		if (b->desc.am == MTFHIP_AM_SSD) l[t] = std::exp(-b->desc.likelihood_alpha * std::sqrt(-f / (double)b->N));
		else { double d = (1.0 / f) - 1; l[t] = std::exp(-b->desc.likelihood_alpha * d * d); }
	}
	return MTFHIP_OK;
}

/* ---- deferred fusion: replay, refresh, and the fused execution of a recognised call sequence ---- */
static int pix_grad_common(mtfhip_batch *b, const double *pts, bool warped, bool init);
/* SSD's DF_DI0 = It - I0 and DF_DIT = -DF_DI0 (SSDBase.cc:75-121), NCC's gradient vectors (NCC.cc:163-234), when a fused
 * launch stood in for the calls that write them: derived from the IT (current or shadow) and scalars they belong to */
static void stale_clear(mtfhip_batch *b, bool df0, bool dft) {
	mtfhip_batch::Lazy &L = b->lz;
	if (df0) L.df0_stale = L.df0_sh = false;
	if (dft) L.dft_stale = L.dft_sh = false;
	if (!L.df0_sh && !L.dft_sh) L.shadow_valid = false;
}
static void swap_shadow(mtfhip_batch *b) {
	std::swap(b->buf[MTFHIP_BUF_IT], b->d_it_shadow);
	if (b->desc.am == MTFHIP_AM_NCC)
		for (int t = 0; t < b->B; ++t) {
			TargetHost &h = b->th[t]; mtfhip_batch::Lazy::NccSave &v = b->lz.ncc_shadow[t];
			std::swap(h.It_mean, v.It_mean); std::swap(h.a, v.a); std::swap(h.b, v.b); std::swap(h.f, v.f);
		}
	b->ncc_host_newer = true;   /* d_ncc has to follow whichever set of scalars is current */
}
static int ensure_one(mtfhip_batch *b, bool curr) {
	mtfhip_batch::Lazy &L = b->lz;
	if (curr ? !L.dft_stale : !L.df0_stale) return MTFHIP_OK;
	const bool sh = curr ? L.dft_sh : L.df0_sh;
	if (sh) swap_shadow(b);
	int rc = MTFHIP_OK;
	if (b->desc.am == MTFHIP_AM_NCC) {
		rc = ncc_update_grad(b, curr ? 1 : 0);
	} else {
		/* the residual kernel writes It - I0 into the view's DF_DI0; for df_dIt it is pointed at DF_DIT and negated in place */
		BatchView v = b->view();
		if (curr) v.buf[MTFHIP_BUF_DF_DI0] = b->buf[MTFHIP_BUF_DF_DIT];
		{
			TimedScope ts(b->ctx, "ssd_residual");
			launch_ssd_residual(v, b->d_partials, simple_blocks_per_target(b->N), b->ctx->stream);
		}
		if (curr) {
			TimedScope ts(b->ctx, "negate");
			launch_negate(b->buf[MTFHIP_BUF_DF_DIT], b->buf[MTFHIP_BUF_DF_DIT], (size_t)b->N * b->B, b->ctx->stream);
		}
	}
	if (sh) swap_shadow(b);
	if (rc) return rc;
	stale_clear(b, !curr, curr);
	return MTFHIP_OK;
}
static int ensure_df(mtfhip_batch *b) {
	TRY(ensure_one(b, false));
	return ensure_one(b, true);
}
/* IT is about to be overwritten by a launch that re-produces df_dI0 (w0) / df_dIt (wt) or not: stale vectors that it does
 * not re-produce keep their IT by a buffer swap instead of being derived now */
static int protect_stale(mtfhip_batch *b, bool w0, bool wt) {
	mtfhip_batch::Lazy &L = b->lz;
	const bool cur0 = L.df0_stale && !w0 && !L.df0_sh, curt = L.dft_stale && !wt && !L.dft_sh;
	if (!cur0 && !curt) return MTFHIP_OK;
	if (L.shadow_valid) {   /* an older shadow is still referenced (rare): settle it first */
		if (L.df0_sh) TRY(ensure_one(b, false));
		if (L.dft_sh) TRY(ensure_one(b, true));
	}
	if (!b->d_it_shadow) HIP_TRY(hipMalloc(&b->d_it_shadow, sizeof(double) * b->per_target[MTFHIP_BUF_IT] * b->B));
	L.ncc_shadow.resize(b->B);
	for (int t = 0; t < b->B; ++t) { const TargetHost &h = b->th[t]; L.ncc_shadow[t] = {h.It_mean, h.a, h.b, h.f}; }
	std::swap(b->buf[MTFHIP_BUF_IT], b->d_it_shadow);   /* the launch fills the other buffer; th keeps the current scalars */
	L.shadow_valid = true;
	if (cur0) L.df0_sh = true;
	if (curt) L.dft_sh = true;
	return MTFHIP_OK;
}
/* replays the recorded calls through the un-fused kernels, in the order they were made */
static int lazy_flush(mtfhip_batch *b, bool pts) {
	mtfhip_batch::Lazy &L = b->lz;
	/* whatever follows a full flush may launch a kernel that reads the current points; so may the replayed calls */
	if (pts || L.pv || L.gp || L.pg || L.pj) TRY(ensure_pts(b));
	if (!L.any()) return MTFHIP_OK;
	struct Op { long seq; int kind; };
	Op ops[8]; int n = 0;
	if (L.pv) ops[n++] = {L.pv, 0};
	if (L.gp) ops[n++] = {L.gp, 1};
	if (L.pg) ops[n++] = {L.pg, 2};
	if (L.pj) ops[n++] = {L.pj, 3};
	if (L.sim) ops[n++] = {L.sim, 4};
	if (L.cg) ops[n++] = {L.cg, 5};
	if (L.ig) ops[n++] = {L.ig, 6};
	if (L.jm) ops[n++] = {L.jm, 7};
	std::sort(ops, ops + n, [](const Op &x, const Op &y) { return x.seq < y.seq; });
	const int pg_kind = L.pg_kind, pj_variant = L.pj_variant; const bool need_f = L.sim_need_f;
	L.pv = L.gp = L.pg = L.pj = L.sim = L.cg = L.ig = L.jm = 0;   /* cleared first: the executors below may flush */
	for (int i = 0; i < n; ++i) {
		switch (ops[i].kind) {
		case 0: TRY(do_update_pix_vals(b, nullptr)); break;
		case 1: TRY(do_update_grad_pts(b, b->desc.grad_eps)); break;
		case 2: TRY(pix_grad_common(b, nullptr, pg_kind == 2, false)); break;
		case 3: TRY(do_cmpt_pix_jacobian(b, pj_variant, MTFHIP_BUF_DIT_DX, MTFHIP_BUF_JT)); break;
		case 4: TRY(do_update_similarity(b, need_f ? 0 : 1)); break;
		case 5: TRY(do_update_curr_grad(b)); break;
		case 6:          /* SSD::updateInitGrad is empty (SSDBase.h) */
			if (b->desc.am == MTFHIP_AM_NCC) { TRY(ncc_update_grad(b, 0)); stale_clear(b, true, false); }
			break;
		default: TRY(do_mean_jacobian(b)); break;
		}
	}
	return MTFHIP_OK;
}
static int fused_args(const mtfhip_batch *b, const mtfhip_sm_desc *sm, FusedArgs &fa);
static int ncc_template_moments(mtfhip_batch *b);
static int ncc_lazy_outputs(mtfhip_batch *b, int trig, int j_a, bool hess_mean, double *g);
enum { LAZY_CURR_JAC = 0, LAZY_DIFF_JAC = 1, LAZY_INIT_JAC = 2 };
/* `*done` = 1 when the pending calls plus this Jacobian request were served by ONE fused launch (g filled with the AM's
 * raw Jacobian), 0 when the caller has to flush and take the un-fused route.
 *   FCLK  NT/FCLK.cc:171-358: updatePixVals, updateSimilarity, updateCurrGrad, pixel gradient + pixel Jacobian, cmptCurrJacobian(Jt)
 *   ESM   NT/ESM.cc:170-296: ... updateInitGrad, cmptDifferenceOfJacobians(J0, Jt)   (jac_type Original: cmptCurrJacobian(Jm))
 *   ICLK  NT/ICLK.cc:160-299: updatePixVals, updateSimilarity, updateInitGrad, cmptInitJacobian(J0) */
static int lazy_try_fused(mtfhip_batch *b, int trig, int j_a, int j_b, double *g, int *done) {
	*done = 0;
	mtfhip_batch::Lazy &L = b->lz;
	if (!L.enabled) return MTFHIP_OK;
	/* either updatePixVals + updateSimilarity are part of the pending set, or they already ran for this very warp and image */
	const bool replay = L.pv && L.sim && L.pv < L.sim;
	const bool current = !L.pv && !L.sim && L.it_epoch == L.epoch && L.df0_it_ver == L.ver[MTFHIP_BUF_IT];
	if (!replay && !current) return MTFHIP_OK;
	if (!b->init_pix_vals || !b->init_sim || !b->have_corners || !b->ctx->img.data || b->ctx->img.channels != 1) return MTFHIP_OK;
	mtfhip_sm_desc sm;
	std::memset(&sm, 0, sizeof(sm));
	sm.materialize = 1; sm.max_iters = 1; sm.chained_warp = 1;
	bool pixel_chain = false;
	if (L.pg || L.pj || L.gp) {
		if (!L.pg || !L.pj || L.pg > L.pj) return MTFHIP_OK;
		if (L.pg_kind == 1) { if (L.gp || L.pj_variant != MTFHIP_JAC_WARPED) return MTFHIP_OK; }
		else { if (!L.gp || L.gp > L.pg || L.pj_variant != MTFHIP_JAC_INIT) return MTFHIP_OK; sm.chained_warp = 0; }
		pixel_chain = true;
	}
	if (L.jm && (!pixel_chain || L.jm < L.pj)) return MTFHIP_OK;
	double gscale = 1.0;
	if (trig == LAZY_INIT_JAC) {
		if (pixel_chain || L.jm || j_a != MTFHIP_BUF_J0 || !b->buf[MTFHIP_BUF_J0]) return MTFHIP_OK;
		sm.sm = MTFHIP_SM_ICLK; sm.hess_type = 0;
	} else {
		if (!pixel_chain || !L.cg || L.cg < L.sim) return MTFHIP_OK;   /* (L.sim is 0 when it already ran) */
		if (trig == LAZY_DIFF_JAC) {
			if (j_a != MTFHIP_BUF_J0 || j_b != MTFHIP_BUF_JT || !b->buf[MTFHIP_BUF_J0]) return MTFHIP_OK;
			sm.sm = MTFHIP_SM_ESM; sm.hess_type = L.jm ? 3 : 5;
		} else if (j_a == MTFHIP_BUF_JT) {
			sm.sm = MTFHIP_SM_FCLK; sm.hess_type = 2;
		} else if (j_a == MTFHIP_BUF_JM && L.jm && b->buf[MTFHIP_BUF_J0]) {
			sm.sm = MTFHIP_SM_ESM; sm.hess_type = 3; gscale = 0.5;   /* df_dIt . (J0 + Jt) / 2, the halving is exact */
		} else return MTFHIP_OK;
	}
	const bool ncc = b->desc.am == MTFHIP_AM_NCC;
	if (ncc && trig == LAZY_INIT_JAC && !L.ig) return MTFHIP_OK;   /* NCC's df_dI0 comes from updateInitGrad */
	/* gradients a previous fused launch skipped and this one will not re-produce keep their IT (when IT is current the
	 * launch rewrites the same bits, nothing to protect) */
	if (replay) TRY(protect_stale(b, ncc ? L.ig != 0 : true, L.cg != 0));
	if (current && trig == LAZY_INIT_JAC && !L.no_cache) {
		/* the lean launch behind getSimilarity() already accumulated this Jacobian for the same IT and J0: no launch */
		bool served = false;
		if (!ncc && L.sim_g_it == L.ver[MTFHIP_BUF_IT] && L.sim_g_j0 == L.ver[MTFHIP_BUF_J0] && !L.sim_g.empty()) {
			for (int t = 0; t < b->B; ++t) std::memcpy(g + (size_t)t * b->S, &L.sim_g[(size_t)8 * t], sizeof(double) * b->S);
			served = true;
		} else if (ncc && !L.ncc_M.empty() && L.ncc_M_it == L.ver[MTFHIP_BUF_IT] && L.ncc_tm_ver == L.ver[MTFHIP_BUF_J0]) {
			std::memcpy(b->h_acc, L.ncc_M.data(), sizeof(double) * L.ncc_M.size());
			const long jt = L.ncc_M_jt, jm = L.ncc_M_jm; const bool mean = L.ncc_M_mean;
			TRY(ncc_lazy_outputs(b, trig, j_a, mean, g));
			L.ncc_M_jt = jt; L.ncc_M_jm = jm;   /* the rows are unchanged: what they hold about Jt / Jm stays as it was */
			served = true;
		}
		if (served) {
			if (ncc && L.ig) { L.df0_stale = true; L.df0_sh = false; if (!L.dft_sh) L.shadow_valid = false; }
			L.ig = 0;
			*done = 1;
			return MTFHIP_OK;
		}
	}
	if (ncc && sm.sm != MTFHIP_SM_FCLK && L.ncc_tm_ver != L.ver[MTFHIP_BUF_J0]) {   /* moments of the template's Jacobian */
		TRY(ncc_template_moments(b));
		L.ncc_tm_ver = L.ver[MTFHIP_BUF_J0];
	}
	FusedArgs fa;
	TRY(fused_args(b, &sm, fa));
	const int nblk = fused_blocks_per_target(b->N, b->B);
	{
		TimedScope ts(b->ctx, "fused_lk");
		launch_fused_ssd(b->view(), b->ctx->img, fa, b->d_partials, nblk, b->ctx->stream);
	}
	touch(b, MTFHIP_BUF_IT);
	b->it_valid = true;
	L.it_epoch = L.epoch;
	if (fa.mode != 2) { touch(b, MTFHIP_BUF_DIT_DX); touch(b, MTFHIP_BUF_JT); b->dit_valid = b->jt_valid = true; }
	const bool want_mean = L.jm != 0;
	/* the N-sized gradient vectors the consumed calls would have written: SSD's df_dI0 is updateSimilarity's residual,
	 * NCC's comes from updateInitGrad; df_dIt from updateCurrGrad in both */
	if (ncc ? L.ig != 0 : replay) { L.df0_stale = true; L.df0_sh = false; }
	L.df0_it_ver = L.ver[MTFHIP_BUF_IT];          /* (when IT was current the launch rewrote the same bits) */
	if (L.cg) { L.dft_stale = true; L.dft_sh = false; }
	if (!L.df0_sh && !L.dft_sh) L.shadow_valid = false;
	L.pv = L.gp = L.pg = L.pj = L.sim = L.cg = L.ig = L.jm = 0;
	if (want_mean) TRY(do_mean_jacobian(b));
	if (ncc) {
		TRY(read_rows(b, nblk, NCC_ACC_COUNT));
		TRY(ncc_lazy_outputs(b, trig, j_a, fa.hess_mean != 0, g));
		*done = 1;
		return MTFHIP_OK;
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) {
		const double *acc = b->h_acc + (size_t)t * ACC_COUNT;
		b->th[t].f = -acc[ACC_RR] / 2;
		for (int s = 0; s < b->S; ++s) g[(size_t)t * b->S + s] = gscale * acc[ACC_G + s];
	}
	if (fa.mode != 2 && !L.no_cache) {   /* the Gram matrix the launch accumulated: Jt, or Jm with hess_mean */
		L.gram_buf = fa.hess_mean ? MTFHIP_BUF_JM : MTFHIP_BUF_JT;
		L.gram_ver = L.ver[L.gram_buf];
		L.gram.resize((size_t)36 * b->B);
		for (int t = 0; t < b->B; ++t) std::memcpy(&L.gram[(size_t)36 * t], b->h_acc + (size_t)t * ACC_COUNT + ACC_H, sizeof(double) * 36);
	}
	*done = 1;
	return MTFHIP_OK;
}

static int gemv_to_host(mtfhip_batch *b, const double *v1, int j1, const double *v2, int j2, int sum_mode, double *g, int diff) {
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "gemv");
		launch_gemv(b->view(), v1, b->buf[j1], v2, j2 >= 0 ? b->buf[j2] : nullptr, sum_mode, b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t)
		for (int s = 0; s < b->S; ++s) {
			double v = b->h_acc[(size_t)t * ACC_COUNT + ACC_G + s];
			if (diff) v -= b->h_acc[(size_t)t * ACC_COUNT + ACC_G2 + s];
			g[(size_t)t * b->S + s] = v;
		}
	return MTFHIP_OK;
}
static int j_ready(mtfhip_batch *b, int id, const char *fn) {
	if (!j_buf_ok(id)) return fail(MTFHIP_ERR_INVALID_ARG, "%s: Jacobian buffer id %d is not J0/JT/JM", fn, id);
	if (!b->buf[id]) return fail(MTFHIP_ERR_LOGIC, "%s: Jacobian buffer %d was never produced", fn, id);
	if (id == MTFHIP_BUF_JT && !b->jt_valid) return fail(MTFHIP_ERR_LOGIC, "%s: JT is not materialised", fn);
	return MTFHIP_OK;
}
int mtfhip_am_cmpt_init_jacobian(mtfhip_batch *b, int j0_buf, double *g) {
	if (!b || !g) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_init_jacobian: NULL argument");
	TRY(am_supported(b, "cmptInitJacobian"));
	TRY(j_ready(b, j0_buf, "cmptInitJacobian"));
	{ int done; TRY(lazy_try_fused(b, LAZY_INIT_JAC, j0_buf, -1, g, &done)); if (done) return MTFHIP_OK; }
	FLUSH_AM(b);
	TRY(ensure_df(b));
	return gemv_to_host(b, b->buf[MTFHIP_BUF_DF_DI0], j0_buf, nullptr, -1, 0, g, 0);
}
int mtfhip_am_cmpt_curr_jacobian(mtfhip_batch *b, int jt_buf, double *g) {
	if (!b || !g) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_curr_jacobian: NULL argument");
	TRY(am_supported(b, "cmptCurrJacobian"));
	TRY(j_ready(b, jt_buf, "cmptCurrJacobian"));
	{ int done; TRY(lazy_try_fused(b, LAZY_CURR_JAC, jt_buf, -1, g, &done)); if (done) return MTFHIP_OK; }
	FLUSH_AM(b);
	TRY(ensure_df(b));
	return gemv_to_host(b, b->buf[MTFHIP_BUF_DF_DIT], jt_buf, nullptr, -1, 0, g, 0);
}
int mtfhip_am_cmpt_difference_of_jacobians(mtfhip_batch *b, int j0_buf, int jt_buf, double *g) {
	if (!b || !g) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_difference_of_jacobians: NULL argument");
	TRY(am_supported(b, "cmptDifferenceOfJacobians"));
	TRY(j_ready(b, j0_buf, "cmptDifferenceOfJacobians"));
	TRY(j_ready(b, jt_buf, "cmptDifferenceOfJacobians"));
	{ int done; TRY(lazy_try_fused(b, LAZY_DIFF_JAC, j0_buf, jt_buf, g, &done)); if (done) return MTFHIP_OK; }
	FLUSH_AM(b);
	TRY(ensure_df(b));
	if (b->desc.am != MTFHIP_AM_SSD) /* (df_dIt * dIt_dp) - (df_dI0 * dI0_dp), NCC.cc:268-280, AppearanceModel.h:161-164 */
		return gemv_to_host(b, b->buf[MTFHIP_BUF_DF_DIT], jt_buf, b->buf[MTFHIP_BUF_DF_DI0], j0_buf, 0, g, 1);
	/* SSD: df_dIt * (dI0_dpssm + dIt_dpssm), SSDBase.cc:186 */
	return gemv_to_host(b, b->buf[MTFHIP_BUF_DF_DIT], jt_buf, nullptr, j0_buf, 1, g, 0);
}
/* J^T J of a pixel Jacobian, from the host-side copy when the buffer has not been written since that copy was made
 * (the fused launch of this iteration accumulated it; the template's J0 only changes with the template) */
static int gram_to_host(mtfhip_batch *b, int j_buf, double *H, double scale, bool accumulate) {
	mtfhip_batch::Lazy &L = b->lz;
	const double *src = nullptr;
	if (!L.no_cache) {
		if (j_buf == L.gram_buf && L.gram_ver == L.ver[j_buf] && !L.gram.empty()) src = L.gram.data();
		else if (j_buf == MTFHIP_BUF_J0 && L.gram0_ver == L.ver[j_buf] && !L.gram0.empty()) src = L.gram0.data();
	}
	if (!src) {
		int nblk = simple_blocks_per_target(b->N);
		{
			TimedScope ts(b->ctx, "gram");
			launch_gram(b->view(), b->buf[j_buf], b->d_partials, nblk, b->ctx->stream);
		}
		TRY(read_acc(b, nblk));
		std::vector<double> &dst = j_buf == MTFHIP_BUF_J0 ? L.gram0 : L.gram;
		dst.resize((size_t)36 * b->B);
		for (int t = 0; t < b->B; ++t) std::memcpy(&dst[(size_t)36 * t], b->h_acc + (size_t)t * ACC_COUNT + ACC_H, sizeof(double) * 36);
		if (j_buf == MTFHIP_BUF_J0) L.gram0_ver = L.ver[j_buf];
		else { L.gram_buf = j_buf; L.gram_ver = L.ver[j_buf]; }
		src = dst.data();
	}
	const int S = b->S;
	for (int t = 0; t < b->B; ++t) {
		int k = 0;
		for (int a = 0; a < 8; ++a)
			for (int c = a; c < 8; ++c) {
				if (a < S && c < S) {
					double v = scale * src[(size_t)36 * t + k];
					double *Ht = H + (size_t)t * S * S;
					if (accumulate) { Ht[c * S + a] += v; if (a != c) Ht[a * S + c] += v; }
					else { Ht[c * S + a] = v; Ht[a * S + c] = v; }
				}
				++k;
			}
	}
	return MTFHIP_OK;
}
int mtfhip_am_cmpt_init_hessian(mtfhip_batch *b, int j0_buf, double *H) {
	FLUSH_AM(b);
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_init_hessian: NULL argument");
	TRY(am_supported(b, "cmptInitHessian"));
	TRY(j_ready(b, j0_buf, "cmptInitHessian"));
	if (b->desc.am == MTFHIP_AM_NCC) return ncc_hessian(b, j0_buf, 0, H);
	if (b->desc.am == MTFHIP_AM_MI) return mi_hessian(b, j0_buf, 0, H);
	return gram_to_host(b, j0_buf, H, -1.0, false);
}
int mtfhip_am_cmpt_curr_hessian(mtfhip_batch *b, int jt_buf, double *H) {
	FLUSH_AM(b);
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_curr_hessian: NULL argument");
	TRY(am_supported(b, "cmptCurrHessian"));
	TRY(j_ready(b, jt_buf, "cmptCurrHessian"));
	if (b->desc.am == MTFHIP_AM_NCC) return ncc_hessian(b, jt_buf, 1, H);
	if (b->desc.am == MTFHIP_AM_MI) return mi_hessian(b, jt_buf, 1, H);
	return gram_to_host(b, jt_buf, H, -1.0, false);
}
int mtfhip_am_cmpt_self_hessian(mtfhip_batch *b, int jt_buf, double *H) {
	FLUSH_AM(b);
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_self_hessian: NULL argument");
	TRY(am_supported(b, "cmptSelfHessian"));
	TRY(j_ready(b, jt_buf, "cmptSelfHessian"));
	if (b->desc.am == MTFHIP_AM_NCC) return ncc_hessian(b, jt_buf, 2, H);
	if (b->desc.am == MTFHIP_AM_MI) return mi_hessian(b, jt_buf, 2, H);
	return gram_to_host(b, jt_buf, H, -1.0, false);
}
int mtfhip_am_cmpt_sum_of_hessians(mtfhip_batch *b, int j0_buf, int jt_buf, double *H) {
	FLUSH_AM(b);
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_sum_of_hessians: NULL argument");
	TRY(am_supported(b, "cmptSumOfHessians"));
	TRY(j_ready(b, j0_buf, "cmptSumOfHessians"));
	TRY(j_ready(b, jt_buf, "cmptSumOfHessians"));
	if (b->desc.am != MTFHIP_AM_SSD) {
		/* generic AppearanceModel::cmptSumOfHessians AppearanceModel.h:196-208 */
		std::vector<double> H0((size_t)b->B * b->S * b->S);
		if (b->desc.am == MTFHIP_AM_NCC) { TRY(ncc_hessian(b, j0_buf, 0, H0.data())); TRY(ncc_hessian(b, jt_buf, 1, H)); }
		else { TRY(mi_hessian(b, j0_buf, 0, H0.data())); TRY(mi_hessian(b, jt_buf, 1, H)); }
		for (size_t i = 0; i < H0.size(); ++i) H[i] += H0[i];
		return MTFHIP_OK;
	}
	TRY(gram_to_host(b, j0_buf, H, -1.0, false));
	return gram_to_host(b, jt_buf, H, -1.0, true);
}
static int do_mean_jacobian(mtfhip_batch *b) {
	TRY(ensure_buf(b, MTFHIP_BUF_JM));
	TimedScope ts(b->ctx, "mean_jacobian");
	launch_mean_jacobian(b->view(), b->ctx->stream);
	touch(b, MTFHIP_BUF_JM);
	return MTFHIP_OK;
}
int mtfhip_sm_mean_jacobian(mtfhip_batch *b) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "mean_jacobian: NULL batch");
	TRY(j_ready(b, MTFHIP_BUF_J0, "mean_jacobian"));
	TRY(j_ready(b, MTFHIP_BUF_JT, "mean_jacobian"));
	if (b->lz.enabled && b->lz.pj) {
		if (b->lz.jm) FLUSH(b);
		if (b->lz.pj) { TRY(ensure_buf(b, MTFHIP_BUF_JM)); b->lz.jm = ++b->lz.seq; return MTFHIP_OK; }
	}
	FLUSH(b);
	return do_mean_jacobian(b);
}

/* ------------------------------------------------------------------ second order (sec_ord_hess) */
static int hess_buf_ok(int id) { return id == MTFHIP_BUF_D2I0_DX2 || id == MTFHIP_BUF_D2IT_DX2; }
static int d2_buf_ok(int id) { return id == MTFHIP_BUF_D2I0_DP2 || id == MTFHIP_BUF_D2IT_DP2 || id == MTFHIP_BUF_D2IM_DP2; }

int mtfhip_ssm_update_hess_pts(mtfhip_batch *b, double hess_eps) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_hess_pts: NULL batch");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "update_hess_pts before set_corners");
	TRY(ensure_buf(b, MTFHIP_BUF_HESS_PTS));
	TimedScope ts(b->ctx, "hess_pts");
	launch_hess_pts(b->view(), hess_eps, b->ctx->stream);
	return MTFHIP_OK;
}

/* ImageBase::initializePixHess / updatePixHess, both overloads (AM/src/ImageBase.cc:174-240, 316-338, 364-386) */
static int pix_hess_common(mtfhip_batch *b, const double *pts, const double *hess_pts, bool warped, bool init) {
	TRY(need_image(b));
	TRY(ensure_buf(b, MTFHIP_BUF_D2I0_DX2));
	TRY(ensure_buf(b, MTFHIP_BUF_D2IT_DX2));
	const size_t N = b->N, NP = b->NP;
	const double *dp, *dh = nullptr;
	TRY(resolve_pts(b, pts, MTFHIP_BUF_CURR_PTS, 2 * NP, &dp));
	if (warped) {
		if (!hess_pts) {
			if (!b->buf[MTFHIP_BUF_HESS_PTS]) return fail(MTFHIP_ERR_LOGIC, "pix_hess: device hess_pts not available (call update_hess_pts)");
			dh = b->buf[MTFHIP_BUF_HESS_PTS];
		} else {
			double *stage = b->d_scratch_pts + 2 * NP * b->B;
			HIP_TRY(hipMemcpyAsync(stage, hess_pts, sizeof(double) * 16 * NP * b->B, hipMemcpyHostToDevice, b->ctx->stream));
			dh = stage;
		}
	}
	double *dst = b->buf[init ? MTFHIP_BUF_D2I0_DX2 : MTFHIP_BUF_D2IT_DX2];
	{
		TimedScope ts(b->ctx, warped ? "warped_img_hess" : "img_hess");
		if (warped) launch_warped_img_hess(b->view(), b->ctx->img, dp, dh, dst, b->hess_eps, b->norm_mult, b->ctx->stream);
		else launch_img_hess(b->view(), b->ctx->img, dp, dst, b->hess_eps, b->norm_mult, b->ctx->stream);
	}
	if (init && !b->init_pix_hess) {
		HIP_TRY(hipMemcpyAsync(b->buf[MTFHIP_BUF_D2IT_DX2], b->buf[MTFHIP_BUF_D2I0_DX2], sizeof(double) * 4 * N * b->B, hipMemcpyDeviceToDevice, b->ctx->stream));
		b->init_pix_hess = true;
	}
	return MTFHIP_OK;
}
int mtfhip_am_initialize_pix_hess(mtfhip_batch *b, const double *pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_pix_hess: NULL batch");
	return pix_hess_common(b, pts, nullptr, false, true);
}
int mtfhip_am_update_pix_hess(mtfhip_batch *b, const double *pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_pix_hess: NULL batch");
	return pix_hess_common(b, pts, nullptr, false, false);
}
int mtfhip_am_initialize_pix_hess_warped(mtfhip_batch *b, const double *pts, const double *hess_pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_pix_hess_warped: NULL batch");
	return pix_hess_common(b, pts, hess_pts, true, true);
}
int mtfhip_am_update_pix_hess_warped(mtfhip_batch *b, const double *pts, const double *hess_pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_pix_hess_warped: NULL batch");
	return pix_hess_common(b, pts, hess_pts, true, false);
}

int mtfhip_ssm_cmpt_pix_hessian(mtfhip_batch *b, int variant, int hess_buf, int grad_buf, int dst_buf) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_pix_hessian: NULL batch");
	if (variant < MTFHIP_JAC_INIT || variant > MTFHIP_JAC_APPROX) return fail(MTFHIP_ERR_INVALID_ARG, "unknown pixel Hessian variant %d", variant);
	if (!hess_buf_ok(hess_buf)) return fail(MTFHIP_ERR_INVALID_ARG, "hess_buf must be D2I0_DX2 or D2IT_DX2");
	if (grad_buf != MTFHIP_BUF_DI0_DX && grad_buf != MTFHIP_BUF_DIT_DX) return fail(MTFHIP_ERR_INVALID_ARG, "grad_buf must be DI0_DX or DIT_DX");
	if (!d2_buf_ok(dst_buf)) return fail(MTFHIP_ERR_INVALID_ARG, "dst_buf must be D2I0_DP2, D2IT_DP2 or D2IM_DP2");
	/* Affine implements Init and Warped only (SSM/include/mtf/SSM/Affine.h); the others are ssm_func_not_implemeted
	 * (StateSpaceModel.h:186-197) */
	if (b->desc.ssm == MTFHIP_SSM_AFFINE && (variant == MTFHIP_JAC_PIX || variant == MTFHIP_JAC_APPROX))
		return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "%s :: function not implemented yet", variant == MTFHIP_JAC_PIX ? "cmptPixHessian" : "cmptApproxPixHessian");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "cmpt_pix_hessian before set_corners");
	if (!b->buf[hess_buf]) return fail(MTFHIP_ERR_LOGIC, "cmpt_pix_hessian: image Hessian %d was never computed", hess_buf);
	TRY(ensure_buf(b, dst_buf));
	TimedScope ts(b->ctx, "pix_hessian");
	launch_pix_hessian(b->view(), variant, b->buf[hess_buf], b->buf[grad_buf], b->buf[dst_buf], b->ctx->stream);
	return MTFHIP_OK;
}

int mtfhip_sm_mean_pix_hessian(mtfhip_batch *b) {
	FLUSH_AM(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "mean_pix_hessian: NULL batch");
	if (!b->buf[MTFHIP_BUF_D2I0_DP2] || !b->buf[MTFHIP_BUF_D2IT_DP2]) return fail(MTFHIP_ERR_LOGIC, "mean_pix_hessian: init / curr pixel Hessians not computed");
	TRY(ensure_buf(b, MTFHIP_BUF_D2IM_DP2));
	TimedScope ts(b->ctx, "mean_pix_hessian");
	launch_mean_planes(b->buf[MTFHIP_BUF_D2I0_DP2], b->buf[MTFHIP_BUF_D2IT_DP2], b->buf[MTFHIP_BUF_D2IM_DP2],
		(size_t)b->B * b->N * b->S * b->S, b->ctx->stream);
	return MTFHIP_OK;
}

/* H[t] += sum_p w[p] * (d2a[:, p] (+ d2b[:, p])) */
static int add_second_order(mtfhip_batch *b, int d2a, int d2b, const double *dev_w, double *H) {
	if (!d2_buf_ok(d2a) || (d2b >= 0 && !d2_buf_ok(d2b))) return fail(MTFHIP_ERR_INVALID_ARG, "pixel-Hessian buffer must be D2I0_DP2, D2IT_DP2 or D2IM_DP2");
	if (!b->buf[d2a] || (d2b >= 0 && !b->buf[d2b])) return fail(MTFHIP_ERR_LOGIC, "second-order Hessian: pixel Hessian buffer was never computed");
	const int nblk = simple_blocks_per_target(b->N), S = b->S;
	if (!b->d_d2_part) {
		HIP_TRY(hipMalloc(&b->d_d2_part, sizeof(double) * 64 * (size_t)nblk * b->B));
		HIP_TRY(hipMalloc(&b->d_d2_out, sizeof(double) * 64 * (size_t)b->B));
	}
	{
		TimedScope ts(b->ctx, "pix_hess_weighted_sum");
		launch_weighted_plane_sum(b->view(), b->buf[d2a], d2b >= 0 ? b->buf[d2b] : nullptr, dev_w, b->d_d2_part, nblk, b->d_d2_out, b->ctx->stream);
	}
	std::vector<double> h((size_t)S * S * b->B);
	HIP_TRY(hipMemcpyAsync(h.data(), b->d_d2_out, sizeof(double) * h.size(), hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	for (size_t i = 0; i < h.size(); ++i) H[i] += h[i];
	return MTFHIP_OK;
}
/* SSDBase.cc:313-343 ; NCC.cc:391-400 ; MI.cc:659-673 */
int mtfhip_am_cmpt_init_hessian2(mtfhip_batch *b, int j0_buf, int d2_buf, double *H) {
	FLUSH_AM(b);
	if (b) TRY(ensure_df(b));   /* the second-order terms are weighted by df_dI */
	TRY(mtfhip_am_cmpt_init_hessian(b, j0_buf, H));
	return add_second_order(b, d2_buf, -1, b->buf[MTFHIP_BUF_DF_DI0], H);
}
/* SSDBase.cc:345-375 ; NCC.cc:401-410 ; MI.cc:680-694 */
int mtfhip_am_cmpt_curr_hessian2(mtfhip_batch *b, int jt_buf, int d2_buf, double *H) {
	FLUSH_AM(b);
	if (b) TRY(ensure_df(b));   /* the second-order terms are weighted by df_dI */
	TRY(mtfhip_am_cmpt_curr_hessian(b, jt_buf, H));
	return add_second_order(b, d2_buf, -1, b->buf[MTFHIP_BUF_DF_DIT], H);
}
/* SSD: first order only (SSDBase.h:95-98) ; NCC: am_func_not_implemeted (AppearanceModel.h:188-191) ; MI.cc:696-733 */
int mtfhip_am_cmpt_self_hessian2(mtfhip_batch *b, int jt_buf, int d2_buf, double *H) {
	FLUSH_AM(b);
	if (b) TRY(ensure_df(b));   /* the second-order terms are weighted by df_dI */
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_self_hessian (second order): NULL argument");
	if (b->desc.am == MTFHIP_AM_NCC) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "ncc :: cmptSelfHessian(second order) :: function not implemented yet");
	TRY(mtfhip_am_cmpt_self_hessian(b, jt_buf, H));
	if (b->desc.am == MTFHIP_AM_SSD) return MTFHIP_OK;
	/* MI: weight = sum_r curr_hist_grad(r) * sum_t curr_hist_mat(t) * self_grad_factor(r, t)  (MI.cc:710-723);
	 * the self table was filled by the first-order call above */
	if (!b->d_d2_w) HIP_TRY(hipMalloc(&b->d_d2_w, sizeof(double) * (size_t)b->N * b->B));
	launch_mi_grad(b->view(), b->desc.mi_n_bins, b->mi_hist_norm, b->buf[MTFHIP_BUF_IT], b->buf[MTFHIP_BUF_IT], b->d_mi_tb, MI_T_SELF,
		b->d_d2_w, b->ctx->stream);
	return add_second_order(b, d2_buf, -1, b->d_d2_w, H);
}
/* SSDBase.cc:377-415 (both pixel Hessians weighted by df_dI0) ; NCC / MI: generic AppearanceModel.h:209-219 */
int mtfhip_am_cmpt_sum_of_hessians2(mtfhip_batch *b, int j0_buf, int jt_buf, int d20_buf, int d2t_buf, double *H) {
	FLUSH_AM(b);
	if (b) TRY(ensure_df(b));   /* the second-order terms are weighted by df_dI */
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_sum_of_hessians (second order): NULL argument");
	if (b->desc.am == MTFHIP_AM_SSD) {
		TRY(mtfhip_am_cmpt_sum_of_hessians(b, j0_buf, jt_buf, H));
		return add_second_order(b, d20_buf, d2t_buf, b->buf[MTFHIP_BUF_DF_DI0], H);
	}
	std::vector<double> H0((size_t)b->B * b->S * b->S);
	TRY(mtfhip_am_cmpt_init_hessian2(b, j0_buf, d20_buf, H0.data()));
	TRY(mtfhip_am_cmpt_curr_hessian2(b, jt_buf, d2t_buf, H));
	for (size_t i = 0; i < H0.size(); ++i) H[i] += H0[i];
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ fused path */
static int check_sm(const mtfhip_batch *b, const mtfhip_sm_desc *sm, const char *fn) {
	if (!b || !sm) return fail(MTFHIP_ERR_INVALID_ARG, "%s: NULL argument", fn);
	if (sm->sm < MTFHIP_SM_ESM || sm->sm > MTFHIP_SM_ICLK) return fail(MTFHIP_ERR_INVALID_ARG, "%s: unknown search method %d", fn, sm->sm);
	int max_h = sm->sm == MTFHIP_SM_ESM ? 5 : 2;
	if (sm->hess_type < 0 || sm->hess_type > max_h) return fail(MTFHIP_ERR_INVALID_ARG, "%s: hess_type %d invalid for search method %d", fn, sm->hess_type, sm->sm);
	if (b->desc.am == MTFHIP_AM_NCC) {
		if (sm->sec_ord_hess) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "%s: second-order NCC Hessians go through the per-function entry points", fn);
		return MTFHIP_OK;
	}
	if (b->desc.am != MTFHIP_AM_SSD) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "%s: the fused path supports SSD and NCC; MI uses the per-function entry points", fn);
	return MTFHIP_OK;
}

/* inverse of a definite S x S matrix (column-major) by Gauss-Jordan on the diagonally scaled system */
static bool invert_definite(int S, const double *H, double *Hinv) {
	double A[8][16], sc[8];
	for (int i = 0; i < S; ++i) { double d = std::fabs(H[i * S + i]); sc[i] = d > 0 ? 1.0 / std::sqrt(d) : 1.0; }
	for (int i = 0; i < S; ++i)
		for (int j = 0; j < S; ++j) { A[i][j] = H[j * S + i] * sc[i] * sc[j]; A[i][S + j] = i == j ? 1.0 : 0.0; }
	for (int k = 0; k < S; ++k) {
		int piv = k;
		for (int i = k + 1; i < S; ++i) if (std::fabs(A[i][k]) > std::fabs(A[piv][k])) piv = i;
		if (A[piv][k] == 0) return false;
		if (piv != k) for (int j = 0; j < 2 * S; ++j) std::swap(A[piv][j], A[k][j]);
		const double p = A[k][k];
		for (int j = 0; j < 2 * S; ++j) A[k][j] /= p;
		for (int i = 0; i < S; ++i) {
			if (i == k) continue;
			const double f = A[i][k];
			if (f == 0) continue;
			for (int j = 0; j < 2 * S; ++j) A[i][j] -= f * A[k][j];
		}
	}
	for (int i = 0; i < S; ++i)
		for (int j = 0; j < S; ++j) Hinv[j * S + i] = A[i][S + j] * sc[i] * sc[j];
	return true;
}

/* ---- NCC on the fused path: everything NCC.cc derives from centred vectors, written in raw moments ----
 * With mt = mean(It), m0 = mean(I0), b = |It - mt|, c = |I0 - m0|, f = a / (b c)  (NCC.cc:124-161) and, for a pixel
 * Jacobian X with column sums sX, Gram(X), sum It X = itX, sum I0 X = i0X:
 *   Jc = (X - mean(X)) / b                       G(X)  = -Jc^T Jc            = -(Gram(X) - sX sX^T / N) / b^2
 *   ut(X) = Jc^T (It - mt) / b = (itX - mt sX) / b^2        u0(X) = Jc^T (I0 - m0) / c = (i0X - m0 sX) / (b c)
 *   df_dIt . X = u0 - f ut   (NCC.cc:196-234, 252-266)       df_dI0 . X = (b / c) (ut - f u0)   (NCC.cc:163-194, 236-250)
 *   cmptCurrHessian = f G - ut u0^T - u0 ut^T + 3 ut ut^T   (NCC.cc:304-335)    cmptInitHessian: ... + 3 u0 u0^T (NCC.cc:282-303)
 *   cmptSelfHessian = G + ut ut^T   (NCC.cc:337-389)
 * (the reference also subtracts the mean of the gradient vectors, which is zero up to rounding because the centred
 * vectors sum to zero; it does not survive into the moments).  Moments of the mean Jacobian (J0 + Jt) / 2 are the means of
 * the moments, except its Gram matrix, which the kernel accumulates itself when hess_mean is set. */
struct NccX { const double *gram; double s[8], it[8], i0[8]; };
struct NccScalars { double N, mt, m0, b, b2, c, f; };
static void ncc_vecs(const NccScalars &q, const NccX &X, int S, double *ut, double *u0) {
	for (int s = 0; s < S; ++s) {
		ut[s] = (X.it[s] - q.mt * X.s[s]) / q.b2;
		u0[s] = (X.i0[s] - q.m0 * X.s[s]) / (q.b * q.c);
	}
}
/* kind 0 init, 1 curr, 2 self; H column-major S x S */
static void ncc_hess_from_moments(const NccScalars &q, const NccX &X, int S, int kind, double *H) {
	double ut[8], u0[8];
	ncc_vecs(q, X, S, ut, u0);
	for (int r = 0; r < S; ++r)
		for (int c = 0; c < S; ++c) {
			const int a = r < c ? r : c, d = r < c ? c : r;
			const double G = -(X.gram[a * 8 - (a * (a - 1)) / 2 + (d - a)] - X.s[r] * X.s[c] / q.N) / q.b2;
			double v;
			if (kind == 2) v = G + ut[r] * ut[c];
			else v = q.f * G - ut[r] * u0[c] - u0[r] * ut[c] + 3 * (kind == 1 ? ut[r] * ut[c] : u0[r] * u0[c]);
			H[c * S + r] = v;
		}
}
static NccScalars ncc_scalars(const mtfhip_batch *b, const TargetHost &h, const double *M) {
	NccScalars q;
	q.N = (double)b->N; q.mt = M[NCC_IT] / q.N; q.m0 = h.I0_mean; q.c = h.c;
	const double a = M[NCC_I0IT] - q.N * q.m0 * q.mt;
	q.b2 = M[NCC_IT2] - q.N * q.mt * q.mt; q.b = std::sqrt(q.b2);
	q.f = a / (q.b * q.c);
	return q;
}
static void ncc_x(const mtfhip_batch *b, const TargetHost &h, const double *M, int which /* 0 J0, 1 Jt, 2 Jm */, bool gram_is_mean, NccX &X) {
	const int S = b->S;
	for (int s = 0; s < 8; ++s) X.s[s] = X.it[s] = X.i0[s] = 0;
	for (int s = 0; s < S; ++s) {
		const double s0 = h.ncc_sj0[s], it0 = M[NCC_ITJ0 + s], i00 = h.ncc_i0j0[s];
		const double st = M[NCC_SJ + s], itt = M[NCC_ITJ + s], i0t = M[NCC_I0J + s];
		if (which == 0) { X.s[s] = s0; X.it[s] = it0; X.i0[s] = i00; }
		else if (which == 1) { X.s[s] = st; X.it[s] = itt; X.i0[s] = i0t; }
		else { X.s[s] = (s0 + st) / 2; X.it[s] = (it0 + itt) / 2; X.i0[s] = (i00 + i0t) / 2; }
	}
	X.gram = which == 0 ? h.ncc_gram0 : ((which == 2) == gram_is_mean ? M + NCC_GRAM : nullptr);
}
/* one target's reduced moment row -> the SM's f, g, H (before LM damping); NT/ESM.cc:298-377, NT/FCLK.cc:260-288, NT/ICLK.cc:206-251 */
static int ncc_assemble(const mtfhip_batch *b, const mtfhip_sm_desc *sm, bool hess_mean, const double *M, TargetHost &h,
	double *f, double *g, double *H) {
	const int S = b->S;
	const NccScalars q = ncc_scalars(b, h, M);
	h.It_mean = q.mt; h.b = q.b; h.a = M[NCC_I0IT] - q.N * q.m0 * q.mt; h.f = q.f;
	if (f) *f = q.f;
	NccX X0, Xt, Xm;
	ncc_x(b, h, M, 0, hess_mean, X0); ncc_x(b, h, M, 1, hess_mean, Xt); ncc_x(b, h, M, 2, hess_mean, Xm);
	double ut[8], u0[8];
	auto curr_jac = [&](const NccX &X, double *o) { ncc_vecs(q, X, S, ut, u0); for (int s = 0; s < S; ++s) o[s] = u0[s] - q.f * ut[s]; };
	auto init_jac = [&](const NccX &X, double *o) { ncc_vecs(q, X, S, ut, u0); for (int s = 0; s < S; ++s) o[s] = (q.b / q.c) * (ut[s] - q.f * u0[s]); };
	if (sm->sm == MTFHIP_SM_FCLK) curr_jac(Xt, g);
	else if (sm->sm == MTFHIP_SM_ICLK) init_jac(X0, g);
	else if (sm->jac_type == 0) curr_jac(Xm, g);
	else { double gt[8], g0[8]; curr_jac(Xt, gt); init_jac(X0, g0); for (int s = 0; s < S; ++s) g[s] = 0.5 * (gt[s] - g0[s]); }
	const int ht = sm->hess_type;
	auto need = [&](const NccX &X) { return X.gram ? MTFHIP_OK : fail(MTFHIP_ERR_LOGIC, "fused NCC: the Gram matrix this Hessian needs was not accumulated"); };
	if (ht == 0) { std::memcpy(H, h.h0, sizeof(double) * S * S); return MTFHIP_OK; }
	if (sm->sm == MTFHIP_SM_ICLK) { ncc_hess_from_moments(q, X0, S, 0, H); return MTFHIP_OK; }   /* Std: cmptInitHessian(J0) */
	if (sm->sm == MTFHIP_SM_FCLK || ht == 1 || ht == 5) { TRY(need(Xt)); ncc_hess_from_moments(q, Xt, S, ht == 1 ? 2 : 1, H); return MTFHIP_OK; }
	if (ht == 2) {   /* SumOfSelf */
		TRY(need(Xt)); ncc_hess_from_moments(q, Xt, S, 2, H);
		for (int k = 0; k < S * S; ++k) H[k] = 0.5 * (H[k] + h.h0[k]);
		return MTFHIP_OK;
	}
	if (ht == 3) { TRY(need(Xm)); ncc_hess_from_moments(q, Xm, S, 1, H); return MTFHIP_OK; }   /* Original: cmptCurrHessian(mean) */
	/* SumOfStd: (cmptInitHessian(J0) + cmptCurrHessian(Jt)) / 2 */
	TRY(need(Xt));
	double Hi[64];
	ncc_hess_from_moments(q, X0, S, 0, Hi); ncc_hess_from_moments(q, Xt, S, 1, H);
	for (int k = 0; k < S * S; ++k) H[k] = 0.5 * (H[k] + Hi[k]);
	return MTFHIP_OK;
}
/* sum J0, sum I0 J0 and Gram(J0) of the template (after every change of J0) */
static int gemv_to_host(mtfhip_batch *b, const double *v1, int j1, const double *v2, int j2, int sum_mode, double *g, int diff);
static int ncc_template_moments(mtfhip_batch *b) {
	const int nblk = simple_blocks_per_target(b->N), S = b->S;
	{
		TimedScope ts(b->ctx, "ncc_hess");
		launch_col_sum(b->view(), b->buf[MTFHIP_BUF_J0], b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t)
		for (int s = 0; s < 8; ++s) b->th[t].ncc_sj0[s] = s < S ? b->h_acc[(size_t)t * ACC_COUNT + ACC_G + s] : 0.0;
	std::vector<double> g((size_t)b->B * S);
	TRY(gemv_to_host(b, b->buf[MTFHIP_BUF_I0], MTFHIP_BUF_J0, nullptr, -1, 0, g.data(), 0));
	for (int t = 0; t < b->B; ++t)
		for (int s = 0; s < 8; ++s) b->th[t].ncc_i0j0[s] = s < S ? g[(size_t)t * S + s] : 0.0;
	{
		TimedScope ts(b->ctx, "gram");
		launch_gram(b->view(), b->buf[MTFHIP_BUF_J0], b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) std::memcpy(b->th[t].ncc_gram0, b->h_acc + (size_t)t * ACC_COUNT + ACC_H, sizeof(double) * 36);
	/* device copy for the device-side finish (k_finish_track) */
	if (!b->d_ncc_tm) HIP_TRY(hipMalloc(&b->d_ncc_tm, sizeof(double) * 52 * (size_t)b->B));
	std::vector<double> tm((size_t)52 * b->B);
	for (int t = 0; t < b->B; ++t) {
		std::memcpy(&tm[52 * (size_t)t], b->th[t].ncc_sj0, sizeof(double) * 8);
		std::memcpy(&tm[52 * (size_t)t + 8], b->th[t].ncc_i0j0, sizeof(double) * 8);
		std::memcpy(&tm[52 * (size_t)t + 16], b->th[t].ncc_gram0, sizeof(double) * 36);
	}
	HIP_TRY(hipMemcpyAsync(b->d_ncc_tm, tm.data(), sizeof(double) * tm.size(), hipMemcpyHostToDevice, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	return MTFHIP_OK;
}

/* deferred fusion, NCC: the AM-level Jacobian the trigger asked for, and the moment rows kept for the Hessian calls */
static int ncc_lazy_outputs(mtfhip_batch *b, int trig, int j_a, bool hess_mean, double *g) {
	mtfhip_batch::Lazy &L = b->lz;
	const int S = b->S;
	for (int t = 0; t < b->B; ++t) {
		const double *M = b->h_acc + (size_t)t * NCC_ACC_COUNT;
		TargetHost &h = b->th[t];
		const NccScalars q = ncc_scalars(b, h, M);
		h.It_mean = q.mt; h.b = q.b; h.a = M[NCC_I0IT] - q.N * q.m0 * q.mt; h.f = q.f;
		NccX X;
		double ut[8], u0[8], *o = g + (size_t)t * S;
		if (trig == LAZY_INIT_JAC) {
			ncc_x(b, h, M, 0, hess_mean, X); ncc_vecs(q, X, S, ut, u0);
			for (int s = 0; s < S; ++s) o[s] = (q.b / q.c) * (ut[s] - q.f * u0[s]);
		} else {
			ncc_x(b, h, M, (trig == LAZY_CURR_JAC && j_a == MTFHIP_BUF_JM) ? 2 : 1, hess_mean, X); ncc_vecs(q, X, S, ut, u0);
			for (int s = 0; s < S; ++s) o[s] = u0[s] - q.f * ut[s];
			if (trig == LAZY_DIFF_JAC) {   /* (df_dIt . Jt) - (df_dI0 . J0), NCC.cc:268-280 */
				ncc_x(b, h, M, 0, hess_mean, X); ncc_vecs(q, X, S, ut, u0);
				for (int s = 0; s < S; ++s) o[s] -= (q.b / q.c) * (ut[s] - q.f * u0[s]);
			}
		}
	}
	b->ncc_host_newer = true;
	if (!L.no_cache) {
		L.ncc_M.assign(b->h_acc, b->h_acc + (size_t)NCC_ACC_COUNT * b->B);
		L.ncc_M_mean = hess_mean;
		L.ncc_M_it = L.ver[MTFHIP_BUF_IT]; L.ncc_M_jt = L.ver[MTFHIP_BUF_JT]; L.ncc_M_jm = L.ver[MTFHIP_BUF_JM];
	}
	return MTFHIP_OK;
}
/* 1 when H was produced from the cached moment rows */
static int ncc_hessian_from_cache(mtfhip_batch *b, int j_buf, int kind, double *H) {
	mtfhip_batch::Lazy &L = b->lz;
	if (L.no_cache || L.ncc_M.empty() || L.ncc_M_it != L.ver[MTFHIP_BUF_IT]) return 0;
	int which;
	if (j_buf == MTFHIP_BUF_J0) { if (L.ncc_tm_ver != L.ver[MTFHIP_BUF_J0]) return 0; which = 0; }
	else if (j_buf == MTFHIP_BUF_JT) { if (L.ncc_M_mean || L.ncc_M_jt != L.ver[MTFHIP_BUF_JT]) return 0; which = 1; }
	else { if (!L.ncc_M_mean || L.ncc_M_jm != L.ver[MTFHIP_BUF_JM] || L.ncc_M_jt != L.ver[MTFHIP_BUF_JT] || L.ncc_tm_ver != L.ver[MTFHIP_BUF_J0]) return 0; which = 2; }
	for (int t = 0; t < b->B; ++t) {
		const double *M = &L.ncc_M[(size_t)t * NCC_ACC_COUNT];
		const NccScalars q = ncc_scalars(b, b->th[t], M);
		NccX X;
		ncc_x(b, b->th[t], M, which, L.ncc_M_mean, X);
		if (!X.gram) return 0;
		ncc_hess_from_moments(q, X, b->S, kind, H + (size_t)t * b->S * b->S);
	}
	return 1;
}

/* getSimilarity() right after updatePixVals + updateSimilarity -- Levenberg-Marquardt's test in the middle of every
 * iteration (NT/ESM.cc:186-204, FCLK.cc:205-223, ICLK.cc:181-199): one launch of the lean (ICLK-type) fused kernel
 * writes IT and accumulates what f needs, instead of sample + residual (SSD) or sample + two reduction passes with
 * two host round trips (NCC).  Anything else pending, or nothing pending: not taken, the caller flushes. */
static int lazy_try_similarity(mtfhip_batch *b) {
	mtfhip_batch::Lazy &L = b->lz;
	if (!L.enabled || !L.pv || !L.sim || L.pv > L.sim || L.gp || L.pg || L.pj || L.cg || L.ig || L.jm) return MTFHIP_OK;
	if (!b->init_pix_vals || !b->init_sim || !b->have_corners || !b->ctx->img.data || b->ctx->img.channels != 1) return MTFHIP_OK;
	const bool ncc = b->desc.am == MTFHIP_AM_NCC;
	mtfhip_sm_desc sm;
	std::memset(&sm, 0, sizeof(sm));
	sm.sm = MTFHIP_SM_ICLK; sm.hess_type = 0; sm.materialize = 1; sm.max_iters = 1; sm.chained_warp = 1;
	FusedArgs fa;
	TRY(fused_args(b, &sm, fa));
	TRY(protect_stale(b, !ncc, false));   /* SSD's updateSimilarity re-produces df_dI0 */
	const int nblk = fused_blocks_per_target(b->N, b->B);
	{
		TimedScope ts(b->ctx, "fused_lk");
		launch_fused_ssd(b->view(), b->ctx->img, fa, b->d_partials, nblk, b->ctx->stream);
	}
	touch(b, MTFHIP_BUF_IT);
	b->it_valid = true;
	L.it_epoch = L.epoch;
	L.df0_it_ver = L.ver[MTFHIP_BUF_IT];
	if (!ncc) { L.df0_stale = true; L.df0_sh = false; if (!L.dft_sh) L.shadow_valid = false; }
	L.pv = L.sim = 0;
	if (ncc) {
		TRY(read_rows(b, nblk, NCC_ACC_COUNT));
		for (int t = 0; t < b->B; ++t) {
			const double *M = b->h_acc + (size_t)t * NCC_ACC_COUNT;
			TargetHost &h = b->th[t];
			const NccScalars q = ncc_scalars(b, h, M);
			h.It_mean = q.mt; h.b = q.b; h.a = M[NCC_I0IT] - q.N * q.m0 * q.mt; h.f = q.f;
		}
		b->ncc_host_newer = true;
		if (!L.no_cache) {   /* sum It J0 and the scalars: enough for cmptInitJacobian / cmptInitHessian of this IT */
			L.ncc_M.assign(b->h_acc, b->h_acc + (size_t)NCC_ACC_COUNT * b->B);
			L.ncc_M_mean = false; L.ncc_M_it = L.ver[MTFHIP_BUF_IT]; L.ncc_M_jt = L.ncc_M_jm = -1;
		}
		return MTFHIP_OK;
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) b->th[t].f = -b->h_acc[(size_t)t * ACC_COUNT + ACC_RR] / 2;
	if (!L.no_cache) {
		L.sim_g.resize((size_t)8 * b->B);
		for (int t = 0; t < b->B; ++t) std::memcpy(&L.sim_g[(size_t)8 * t], b->h_acc + (size_t)t * ACC_COUNT + ACC_G, sizeof(double) * 8);
		L.sim_g_it = L.ver[MTFHIP_BUF_IT]; L.sim_g_j0 = L.ver[MTFHIP_BUF_J0];
	}
	return MTFHIP_OK;
}

int mtfhip_batch_init_template(mtfhip_batch *b, const mtfhip_sm_desc *sm) {
	FLUSH(b);
	if (b) { touch_all(b); b->lz.it_epoch = -1; TRY(ensure_df(b)); }
	TRY(check_sm(b, sm, "init_template"));
	TRY(single_channel(b, "init_template"));
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "init_template before set_corners");
	/* am->clearInitStatus() (NT/ESM.cc:113, NT/FCLK.cc:105, NT/ICLK.cc:74) */
	b->init_pix_vals = b->init_pix_grad = b->init_sim = b->init_grad = false;
	TRY(mtfhip_am_initialize_pix_vals(b, nullptr));
	if (sm->chained_warp) {
		TRY(mtfhip_am_initialize_pix_grad(b, nullptr));
		TRY(mtfhip_ssm_cmpt_pix_jacobian(b, MTFHIP_JAC_WARPED, MTFHIP_BUF_DI0_DX, MTFHIP_BUF_J0));
	} else {
		TRY(mtfhip_ssm_update_grad_pts(b, b->desc.grad_eps));
		TRY(mtfhip_am_initialize_pix_grad_warped(b, nullptr));
		TRY(mtfhip_ssm_cmpt_pix_jacobian(b, MTFHIP_JAC_INIT, MTFHIP_BUF_DI0_DX, MTFHIP_BUF_J0));
	}
	if (sm->sec_ord_hess) {   /* initializePixHess, NT/ESM.cc:406-416 ; the template's pixel Hessian is rebuilt per pixel from
	                           * d2I0_dx2 and dI0_dx inside k_second_order_ssd instead of being stored as an S^2 x N matrix */
		b->init_pix_hess = false;
		if (sm->chained_warp) TRY(mtfhip_am_initialize_pix_hess(b, nullptr));
		else { TRY(mtfhip_ssm_update_hess_pts(b, b->hess_eps)); TRY(mtfhip_am_initialize_pix_hess_warped(b, nullptr, nullptr)); }
		b->d0_variant = sm->chained_warp ? MTFHIP_JAC_WARPED : MTFHIP_JAC_INIT;
	}
	TRY(mtfhip_am_initialize_similarity(b));
	TRY(mtfhip_am_initialize_grad(b));
	TRY(mtfhip_am_initialize_hess(b));
	std::vector<double> H0((size_t)b->B * b->S * b->S), h0dev((size_t)b->B * 64, 0.0);
	TRY(mtfhip_am_cmpt_self_hessian(b, MTFHIP_BUF_J0, H0.data()));
	for (int t = 0; t < b->B; ++t) {
		std::memset(b->th[t].h0, 0, sizeof(b->th[t].h0));
		std::memcpy(b->th[t].h0, &H0[(size_t)t * b->S * b->S], sizeof(double) * b->S * b->S);
		std::memcpy(&h0dev[(size_t)t * 64], b->th[t].h0, sizeof(double) * 64);
	}
	HIP_TRY(hipMemcpyAsync(b->d_h0, h0dev.data(), sizeof(double) * h0dev.size(), hipMemcpyHostToDevice, b->ctx->stream));
	std::vector<double> hinv((size_t)b->B * 64, 0.0);
	for (int t = 0; t < b->B; ++t)
		if (!invert_definite(b->S, b->th[t].h0, &hinv[(size_t)t * 64]))
			std::fill(hinv.begin() + (size_t)t * 64, hinv.begin() + (size_t)(t + 1) * 64, 0.0); /* flat template: no update */
	HIP_TRY(hipMemcpyAsync(b->d_h0inv, hinv.data(), sizeof(double) * hinv.size(), hipMemcpyHostToDevice, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	if (b->desc.am == MTFHIP_AM_NCC) TRY(ncc_template_moments(b));
	b->j0_is_template = true;
	b->j0_template_corners_epoch = b->corners_epoch;
	b->j0_variant = sm->chained_warp ? MTFHIP_JAC_WARPED : MTFHIP_JAC_INIT;
	b->template_corners.resize(8 * (size_t)b->B);
	for (int t = 0; t < b->B; ++t) std::memcpy(&b->template_corners[8 * t], b->th[t].init_corners, sizeof(double) * 8);
	return MTFHIP_OK;
}

/* nt::ESM::setRegion NT/ESM.cc:148-168, nt::FCLK::setRegion NT/FCLK.cc:360-376, nt::ICLK::setRegion NT/ICLK.cc:131-157 (update_ssm
 * off): the SSM is reset to the new corners; ESM (and FCLK with the InitialSelf Hessian) recompute init_pix_jacobian with
 * cmptInitPixJacobian on the new grid and, for the Hessian types that use it, the constant self Hessian; ICLK keeps its
 * template Jacobian.  The template (I0, dI0_dx) is kept in every case. */
int mtfhip_batch_set_region(mtfhip_batch *b, const double *corners, const mtfhip_sm_desc *sm) {
	FLUSH(b);
	if (b) { touch_all(b); b->lz.it_epoch = -1; TRY(ensure_df(b)); }
	TRY(check_sm(b, sm, "set_region"));
	TRY(single_channel(b, "set_region"));
	if (!b->init_pix_vals) return fail(MTFHIP_ERR_LOGIC, "set_region before init_template");
	TRY(mtfhip_ssm_set_corners(b, corners));
	const bool refresh = sm->sm == MTFHIP_SM_ESM || (sm->sm == MTFHIP_SM_FCLK && sm->hess_type == 0);
	if (!refresh) {
		/* back on exactly the grid the kept template Jacobian was computed on: its rows can still be rebuilt from dI0_dx */
		if (b->j0_is_template && b->template_corners.size() == 8 * (size_t)b->B &&
			std::memcmp(b->template_corners.data(), corners, sizeof(double) * 8 * b->B) == 0)
			b->j0_template_corners_epoch = b->corners_epoch;
		return MTFHIP_OK;
	}
	TRY(mtfhip_ssm_cmpt_pix_jacobian(b, MTFHIP_JAC_INIT, MTFHIP_BUF_DI0_DX, MTFHIP_BUF_J0));
	const bool need_h0 = sm->hess_type == 0 || (sm->sm == MTFHIP_SM_ESM && sm->hess_type == 2);
	if (need_h0) {
		std::vector<double> H0((size_t)b->B * b->S * b->S), h0dev((size_t)b->B * 64, 0.0), hinv((size_t)b->B * 64, 0.0);
		TRY(mtfhip_am_cmpt_self_hessian(b, MTFHIP_BUF_J0, H0.data()));
		for (int t = 0; t < b->B; ++t) {
			std::memset(b->th[t].h0, 0, sizeof(b->th[t].h0));
			std::memcpy(b->th[t].h0, &H0[(size_t)t * b->S * b->S], sizeof(double) * b->S * b->S);
			std::memcpy(&h0dev[(size_t)t * 64], b->th[t].h0, sizeof(double) * 64);
			if (!invert_definite(b->S, b->th[t].h0, &hinv[(size_t)t * 64]))
				std::fill(hinv.begin() + (size_t)t * 64, hinv.begin() + (size_t)(t + 1) * 64, 0.0);
		}
		HIP_TRY(hipMemcpyAsync(b->d_h0, h0dev.data(), sizeof(double) * h0dev.size(), hipMemcpyHostToDevice, b->ctx->stream));
		HIP_TRY(hipMemcpyAsync(b->d_h0inv, hinv.data(), sizeof(double) * hinv.size(), hipMemcpyHostToDevice, b->ctx->stream));
		HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	}
	if (b->desc.am == MTFHIP_AM_NCC) TRY(ncc_template_moments(b));
	b->j0_is_template = true;
	b->j0_template_corners_epoch = b->corners_epoch;
	b->j0_variant = MTFHIP_JAC_INIT;
	b->template_corners.assign(corners, corners + 8 * (size_t)b->B);
	return MTFHIP_OK;
}

static int fused_args(const mtfhip_batch *b, const mtfhip_sm_desc *sm, FusedArgs &fa) {
	fa.chained = sm->chained_warp ? 1 : 0;
	fa.materialize = sm->materialize ? 1 : 0;
	fa.hess_mean = 0;
	fa.j0_recompute = (b->j0_is_template && b->j0_recompute_enabled && b->j0_template_corners_epoch == b->corners_epoch) ? 1 : 0;
	fa.j0_init_variant = b->j0_variant == MTFHIP_JAC_INIT ? 1 : 0;
	fa.grad_eps = b->desc.grad_eps;
	fa.norm_mult = b->norm_mult; fa.norm_add = b->norm_add;
	fa.active = nullptr;
	fa.done = nullptr;
	{ int nb; fused_decomposition(b->N, b->B, nb, fa.rows_per_block); }
	switch (sm->sm) {
	case MTFHIP_SM_FCLK: fa.mode = 0; break;
	case MTFHIP_SM_ESM: fa.mode = 1; fa.hess_mean = sm->hess_type == 3; break;
	default:
		if (sm->hess_type == 1) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "fused ICLK with hess_type CurrentSelf: use the un-fused entry points");
		fa.mode = 2;
	}
	return MTFHIP_OK;
}

/* The second-order term an SSD search method adds to its Hessian (k_second_order_ssd's `term`), -1 for none:
 * SSD's self Hessians are first order by definition (SSDBase.h:95-98) and InitialSelf never looks at the frame. */
static int second_order_term(const mtfhip_sm_desc *sm) {
	if (!sm->sec_ord_hess) return -1;
	switch (sm->sm) {
	case MTFHIP_SM_FCLK: return sm->hess_type == 2 ? 0 : -1;
	case MTFHIP_SM_ESM: return sm->hess_type == 5 ? 0 : (sm->hess_type == 4 ? 1 : (sm->hess_type == 3 ? 2 : -1));
	default: return sm->hess_type == 2 ? 3 : -1;
	}
}

/* turns one target's reduced accumulators into the SM's g and H (before LM damping):
 * NT/FCLK.cc:260-288 ; NT/ESM.cc:298-377 with SSDBase.cc:169-191,287-311 ; NT/ICLK.cc:206-251 */
static void assemble(const mtfhip_batch *b, const mtfhip_sm_desc *sm, const double *acc, const double *h0,
	double *f, double *g, double *H) {
	const int S = b->S;
	if (f) *f = -acc[ACC_RR] / 2;
	const double gscale = sm->sm == MTFHIP_SM_ESM ? 0.5 : 1.0;
	for (int s = 0; s < S; ++s) g[s] = gscale * acc[ACC_G + s];
	const bool use_h0 = (sm->hess_type == 0) || (sm->sm == MTFHIP_SM_ICLK);
	const bool sum_h0 = (sm->sm == MTFHIP_SM_ESM) && (sm->hess_type == 2 || sm->hess_type == 4);
	int k = 0;
	for (int a = 0; a < 8; ++a)
		for (int c = a; c < 8; ++c) {
			if (a < S && c < S) {
				double v = use_h0 ? h0[c * S + a] : -acc[ACC_H + k];
				if (sum_h0) v = (v + h0[c * S + a]) * 0.5;
				H[c * S + a] = v; H[a * S + c] = v;
			}
			++k;
		}
}

int mtfhip_batch_iterate(mtfhip_batch *b, const mtfhip_sm_desc *sm, double *f, double *g, double *H) {
	FLUSH(b);
	if (b) { touch_all(b); b->lz.it_epoch = -1; TRY(ensure_df(b)); }
	TRY(check_sm(b, sm, "iterate"));
	TRY(single_channel(b, "iterate"));
	if (!g || !H) return fail(MTFHIP_ERR_INVALID_ARG, "iterate: NULL output");
	if (!b->init_pix_vals) return fail(MTFHIP_ERR_LOGIC, "iterate before init_template");
	TRY(need_image(b));
	FusedArgs fa;
	TRY(fused_args(b, sm, fa));
	int nblk = fused_blocks_per_target(b->N, b->B);
	{
		TimedScope ts(b->ctx, "fused_lk");
		launch_fused_ssd(b->view(), b->ctx->img, fa, b->d_partials, nblk, b->ctx->stream);
	}
	b->it_valid = fa.materialize;
	b->dit_valid = fa.materialize && fa.mode != 2;
	b->jt_valid = fa.materialize && fa.mode != 2;
	if (b->desc.am == MTFHIP_AM_NCC) {
		TRY(read_rows(b, nblk, NCC_ACC_COUNT));
		for (int t = 0; t < b->B; ++t) {
			double ft;
			TRY(ncc_assemble(b, sm, fa.hess_mean != 0, b->h_acc + (size_t)t * NCC_ACC_COUNT, b->th[t], &ft, g + (size_t)t * b->S,
				H + (size_t)t * b->S * b->S));
			if (f) f[t] = ft;
		}
		b->ncc_host_newer = true;
		return MTFHIP_OK;
	}
	const int term = second_order_term(sm);
	std::vector<double> so;
	if (term >= 0) {
		if (term != 0 && !b->init_pix_hess) return fail(MTFHIP_ERR_LOGIC, "iterate: init_template was run without sec_ord_hess");
		const int nb2 = simple_blocks_per_target(b->N);
		if (!b->d_d2_part) {
			HIP_TRY(hipMalloc(&b->d_d2_part, sizeof(double) * 64 * (size_t)nb2 * b->B));
			HIP_TRY(hipMalloc(&b->d_d2_out, sizeof(double) * 64 * (size_t)b->B));
		}
		{
			TimedScope ts(b->ctx, "second_order");
			launch_second_order_ssd(b->view(), b->ctx->img, term, fa.chained, b->d0_variant, fa.grad_eps, b->hess_eps, b->norm_mult,
				b->norm_add, b->d_d2_part, nb2, b->d_d2_out, b->ctx->stream);
		}
		so.resize((size_t)b->S * b->S * b->B);
		HIP_TRY(hipMemcpyAsync(so.data(), b->d_d2_out, sizeof(double) * so.size(), hipMemcpyDeviceToHost, b->ctx->stream));
	}
	TRY(read_acc(b, nblk));
	const int S2 = b->S * b->S;
	for (int t = 0; t < b->B; ++t) {
		double ft;
		double *Ht = H + (size_t)t * S2;
		assemble(b, sm, b->h_acc + (size_t)t * ACC_COUNT, b->th[t].h0, &ft, g + (size_t)t * b->S, Ht);
		if (term >= 0) {   /* SumOfStd halves the whole sum (NT/ESM.cc:339) */
			const double sc = term == 1 ? 0.5 : 1.0;
			for (int k = 0; k < S2; ++k) Ht[k] += sc * so[(size_t)t * S2 + k];
		}
		b->th[t].f = ft;
		if (f) f[t] = ft;
	}
	return MTFHIP_OK;
}

/* Targets per launch of the device-side loop.  Chunking pays where an iteration both re-reads a large constant operand
 * set and writes as much again (ESM with materialisation: 88 B/px read, 88 B/px written): +15-17 % at B = 128-256.
 * FCLK reads only 24 B/px (fits anyway) and the lean / ICLK variants barely write, so for them a chunk only multiplies
 * the per-iteration finish launches (measured 7-20 % slower) and they keep one launch for all targets.
 * MTFHIP_TRACK_CHUNK_PX overrides the pixel budget (tests force tiny chunks with it, in every mode). */
static int track_chunk(const mtfhip_batch *b, const mtfhip_sm_desc *sm, const FusedArgs &fa) {
	const char *env_px = std::getenv("MTFHIP_TRACK_CHUNK_PX");
	if (!env_px && !(fa.mode == 1 && fa.materialize)) return b->B;
	const double chunk_px = env_px ? std::atof(env_px) : 2.6e6;
	int chunk = (int)(chunk_px / (double)b->N);
	if (chunk < 1) chunk = 1;
	if (chunk >= b->B || sm->max_iters == 1) return b->B;
	const int n_chunks = (b->B + chunk - 1) / chunk;
	return (b->B + n_chunks - 1) / n_chunks;   /* balanced: 100 targets -> 50 + 50, not 65 + 35 */
}
int mtfhip_batch_track_targets_per_launch(mtfhip_batch *b, const mtfhip_sm_desc *sm) {
	FLUSH(b);
	if (check_sm(b, sm, "track_targets_per_launch") != MTFHIP_OK) return 0;
	const bool one_launch = sm->sm == MTFHIP_SM_ICLK && (sm->hess_type == 0 || (sm->hess_type == 2 && b->desc.am == MTFHIP_AM_SSD)) &&
		b->N <= kIclkTrackMaxPix;
	if (one_launch) return b->B;
	FusedArgs fa;
	if (fused_args(b, sm, fa) != MTFHIP_OK) return 0;
	return track_chunk(b, sm, fa);
}

int mtfhip_batch_track(mtfhip_batch *b, const mtfhip_sm_desc *sm, int *n_iters, double *corners) {
	FLUSH(b);
	if (b) { touch_all(b); b->lz.it_epoch = -1; TRY(ensure_df(b)); }
	TRY(check_sm(b, sm, "track"));
	TRY(single_channel(b, "track"));
	if (sm->leven_marq) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "track: Levenberg-Marquardt is only available through iterate + host solve");
	if (sm->max_iters <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "track: max_iters must be positive");
	if (b->desc.am != MTFHIP_AM_SSD ? sm->sec_ord_hess != 0 : second_order_term(sm) >= 0)
		return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "track: a second-order Hessian is indefinite and needs the pivoted host solve; use iterate");
	if (!b->init_pix_vals) return fail(MTFHIP_ERR_LOGIC, "track before init_template");
	TRY(need_image(b));
	hipStream_t st = b->ctx->stream;
	const bool one_launch = sm->sm == MTFHIP_SM_ICLK && (sm->hess_type == 0 || (sm->hess_type == 2 && b->desc.am == MTFHIP_AM_SSD)) &&
		b->N <= kIclkTrackMaxPix;
	FusedArgs fa;
	if (!one_launch) TRY(fused_args(b, sm, fa));
	else { fa.materialize = 0; fa.mode = 2; fa.active = nullptr; fa.done = nullptr; fa.rows_per_block = 1; fa.j0_recompute = 0; }
	/* active = 1, iters = 0, corners, warps, states, NCC scalars: one pinned async copy of the whole slab
	 * (w0 is copied along; init_grid consumed it long ago) */
	HIP_TRY(hipEventSynchronize(b->ev_b));
	std::memcpy(b->h_stage_b + 45 * sizeof(double) * (size_t)b->B, b->h_stage_a + 45 * sizeof(double) * (size_t)b->B, 9 * sizeof(double) * (size_t)b->B);
	fill_stage(b, b->h_stage_b, nullptr, 1, true);
	HIP_TRY(hipMemcpyAsync(b->d_slab, b->h_stage_b, b->slab_bytes, hipMemcpyHostToDevice, st));
	fa.active = b->d_active;
	const bool ncc = b->desc.am == MTFHIP_AM_NCC;
	const size_t RL = ncc ? NCC_ACC_COUNT : ACC_COUNT;   /* partial / reduced row length */
	if (ncc && !one_launch && !b->d_ncc_tm) return fail(MTFHIP_ERR_LOGIC, "track before init_template");
	TrackState ts{b->d_acc, b->d_h0, b->d_corners, b->d_init_corners_hm, b->d_active, b->d_iters, ncc ? b->d_ncc : nullptr, ncc ? b->d_ncc_tm : nullptr};
	BatchView bv = b->view();
	if (one_launch) {
		TimedScope tsc(b->ctx, "iclk_track");
		launch_iclk_track(bv, b->ctx->img, *sm, ts, b->d_h0inv, b->d_ncc, b->norm_mult, b->norm_add, st);
	} else {
		/* MTFHIP_EPILOGUE=1: one launch per iteration, the workgroup that completes a target's partial rows also runs the
		 * finish; default: the separate k_finish_track launch (same step time, cleaner kernel timing). */
		if (b->epilogue && !ncc) {
			if (!b->d_done) {
				HIP_TRY(hipMalloc(&b->d_done, sizeof(int) * b->B));
				HIP_TRY(hipMemsetAsync(b->d_done, 0, sizeof(int) * b->B, st));
			}
			fa.done = b->d_done; fa.sm = *sm; fa.ts = ts;
		}
		/* Targets are independent, so the loops commute: all iterations of a chunk of targets run before the next chunk
		 * starts.  A chunk is sized so that what an iteration reads once (J0, I0, grid: 88 B/px for ESM) stays resident in
		 * the 256 MB Infinity Cache from one iteration to the next -- B = 64 at 200 x 200; larger batches used to fall back
		 * to plain HBM for both streams (0.62 instead of 0.75 of peak).  See track_chunk(). */
		const int chunk = track_chunk(b, sm, fa);
		for (int t0 = 0; t0 < b->B; t0 += chunk) {
			const int nt = std::min(chunk, b->B - t0);
			BatchView bc = bv;
			bc.B = nt;
			for (int i = 0; i < MTFHIP_BUF_COUNT; ++i)
				if (bc.buf[i]) bc.buf[i] += (size_t)t0 * b->per_target[i];
			bc.warps += 9 * (size_t)t0; bc.states += 8 * (size_t)t0;
			FusedArgs fc = fa;
			fc.active = fa.active + t0;
			TrackState tc{ts.acc + (size_t)t0 * RL, ts.h0 + (size_t)t0 * 64, ts.corners + 8 * (size_t)t0,
				ts.init_corners_hm + 12 * (size_t)t0, ts.active + t0, ts.n_iters + t0, ncc ? ts.ncc + 8 * (size_t)t0 : nullptr,
				ncc ? ts.ncc_tm + 52 * (size_t)t0 : nullptr};
			if (fc.done) { fc.done = fa.done + t0; fc.ts = tc; }
			int nblk_c; { int rows; fused_decomposition(b->N, nt, nblk_c, rows); fc.rows_per_block = rows; }
			double *part = b->d_partials + (size_t)t0 * b->nblk_max * RL;
			for (int it = 0; it < sm->max_iters; ++it) {
				{
					TimedScope tsc(b->ctx, "fused_lk");
					launch_fused_ssd(bc, b->ctx->img, fc, part, nblk_c, st);
				}
				if (!b->epilogue || ncc) launch_finish_track(bc, *sm, tc, part, nblk_c, st);
			}
		}
	}
	/* one download of the slab (warps, states, corners, iteration counts), one sync */
	HIP_TRY(hipMemcpyAsync(b->h_stage_b, b->d_slab, b->slab_bytes, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipEventRecord(b->ev_b, st));
	HIP_TRY(hipStreamSynchronize(st));
	{
		const size_t Bt = (size_t)b->B;
		const double *p = reinterpret_cast<const double *>(b->h_stage_b);
		const double *w = p, *s = p + 9 * Bt, *cr = p + 17 * Bt;
		const int *iters = reinterpret_cast<const int *>(b->h_stage_b + b->slab_dbl_bytes) + Bt;
		for (int t = 0; t < b->B; ++t) {
			std::memcpy(b->th[t].warp.m, w + 9 * t, sizeof(double) * 9);
			std::memcpy(b->th[t].state, s + 8 * t, sizeof(double) * 8);
			std::memcpy(b->th[t].corners, cr + 8 * t, sizeof(double) * 8);
			if (n_iters) n_iters[t] = iters[t];
			if (corners) std::memcpy(corners + 8 * t, cr + 8 * t, sizeof(double) * 8);
		}
	}
	b->it_valid = fa.materialize;
	b->dit_valid = b->jt_valid = fa.materialize && fa.mode != 2;
	/* curr_pts follow the final warp */
	launch_apply_warp(b->view(), st);
	b->pts_stale = false;
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ candidate scoring */
int mtfhip_score_candidates_dev(mtfhip_batch *b, const double *dev_states, int C, double *dev_lik, double *dev_sim) {
	FLUSH(b);
	if (!b || !dev_states) return fail(MTFHIP_ERR_INVALID_ARG, "score_candidates: NULL argument");
	if (C <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "score_candidates: n_candidates must be positive");
	if (b->desc.am != MTFHIP_AM_SSD) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "score_candidates: SSD only");
	TRY(single_channel(b, "score_candidates"));
	if (!b->init_pix_vals) return fail(MTFHIP_ERR_LOGIC, "score_candidates before the template was initialised");
	TRY(need_image(b));
	TimedScope ts(b->ctx, "score_candidates");
	/* image tile = bounding box of the template region + a margin for the candidate cloud (PF sigmas are a few pixels) */
	const TargetHost &h0 = b->th[0];
	double xmin = h0.init_corners[0], xmax = xmin, ymin = h0.init_corners[1], ymax = ymin;
	for (int q = 1; q < 4; ++q) {
		xmin = std::min(xmin, h0.init_corners[2 * q]); xmax = std::max(xmax, h0.init_corners[2 * q]);
		ymin = std::min(ymin, h0.init_corners[2 * q + 1]); ymax = std::max(ymax, h0.init_corners[2 * q + 1]);
	}
	const int margin = 16;
	const size_t lds_left = 160 * 1024 > (size_t)b->N * 32 ? 160 * 1024 - (size_t)b->N * 32 : 0;
	int tx0 = (int)std::floor(xmin) - margin, ty0 = (int)std::floor(ymin) - margin;
	int tw = (int)std::ceil(xmax) + margin + 2 - tx0, th = (int)std::ceil(ymax) + margin + 2 - ty0;
	bool staged = false;
	if (b->score_lds && C >= 64 && (size_t)tw * th * 4 <= lds_left) {
		if ((size_t)C * kScoreUnitsPerCandidate > b->unit_capacity) {
			if (b->d_units) HIP_TRY(hipFree(b->d_units));
			b->d_units = nullptr;
			HIP_TRY(hipMalloc(&b->d_units, sizeof(double) * C * kScoreUnitsPerCandidate));
			b->unit_capacity = (size_t)C * kScoreUnitsPerCandidate;
		}
		staged = launch_score_candidates_lds(b->view(), b->ctx->img, dev_states, C, tx0, ty0, tw, th, b->desc.likelihood_alpha,
			b->d_units, dev_lik, dev_sim, b->ctx->stream);
	}
	if (!staged)
		launch_score_candidates(b->view(), b->ctx->img, dev_states, C, b->desc.likelihood_alpha, dev_lik, dev_sim, b->ctx->stream);
	return MTFHIP_OK;
}

int mtfhip_score_candidates(mtfhip_batch *b, const double *states, int C, double *lik, double *sim) {
	FLUSH(b);
	if (!b || !states) return fail(MTFHIP_ERR_INVALID_ARG, "score_candidates: NULL argument");
	if (C <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "score_candidates: n_candidates must be positive");
	size_t need = (size_t)C * (b->S + 2);
	if (need > b->cand_capacity) {
		if (b->d_cand) HIP_TRY(hipFree(b->d_cand));
		b->d_cand = nullptr;
		HIP_TRY(hipMalloc(&b->d_cand, sizeof(double) * need));
		b->cand_capacity = need;
	}
	double *d_states = b->d_cand, *d_lik = b->d_cand + (size_t)C * b->S, *d_sim = d_lik + C;
	HIP_TRY(hipMemcpyAsync(d_states, states, sizeof(double) * C * b->S, hipMemcpyHostToDevice, b->ctx->stream));
	TRY(mtfhip_score_candidates_dev(b, d_states, C, d_lik, d_sim));
	if (lik) HIP_TRY(hipMemcpyAsync(lik, d_lik, sizeof(double) * C, hipMemcpyDeviceToHost, b->ctx->stream));
	if (sim) HIP_TRY(hipMemcpyAsync(sim, d_sim, sizeof(double) * C, hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ NN dataset generation */
int mtfhip_sample_candidates_dev(mtfhip_batch *b, const double *dev_states, int C, double *dev_features) {
	FLUSH(b);
	if (!b || !dev_states || !dev_features) return fail(MTFHIP_ERR_INVALID_ARG, "sample_candidates: NULL argument");
	if (C <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "sample_candidates: n_samples must be positive");
	if (b->desc.am == MTFHIP_AM_MI) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "sample_candidates: MI distance features (5 x N B-spline rows) are not available");
	TRY(single_channel(b, "sample_candidates"));
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "sample_candidates before set_corners");
	TRY(need_image(b));
	TimedScope ts(b->ctx, "sample_candidates");
	launch_sample_candidates(b->view(), b->ctx->img, dev_states, C, b->norm_mult, b->norm_add, dev_features, b->ctx->stream);
	return MTFHIP_OK;
}
int mtfhip_sample_candidates(mtfhip_batch *b, const double *states, int C, double *features) {
	FLUSH(b);
	if (!b || !states || !features) return fail(MTFHIP_ERR_INVALID_ARG, "sample_candidates: NULL argument");
	if (C <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "sample_candidates: n_samples must be positive");
	double *d_states = nullptr, *d_feat = nullptr;
	HIP_TRY(hipMalloc(&d_states, sizeof(double) * C * b->S));
	if (hipMalloc(&d_feat, sizeof(double) * (size_t)C * b->N) != hipSuccess) { (void)hipFree(d_states); return fail(MTFHIP_ERR_HIP, "hipMalloc of the %d x %d feature matrix failed", C, b->N); }
	int rc = MTFHIP_OK;
	if (hipMemcpyAsync(d_states, states, sizeof(double) * C * b->S, hipMemcpyHostToDevice, b->ctx->stream) != hipSuccess) rc = fail(MTFHIP_ERR_HIP, "state upload failed");
	if (rc == MTFHIP_OK) rc = mtfhip_sample_candidates_dev(b, d_states, C, d_feat);
	if (rc == MTFHIP_OK && hipMemcpyAsync(features, d_feat, sizeof(double) * (size_t)C * b->N, hipMemcpyDeviceToHost, b->ctx->stream) != hipSuccess) rc = fail(MTFHIP_ERR_HIP, "feature read-back failed");
	if (hipStreamSynchronize(b->ctx->stream) != hipSuccess && rc == MTFHIP_OK) rc = fail(MTFHIP_ERR_HIP, "stream synchronisation failed");
	(void)hipFree(d_states); (void)hipFree(d_feat);
	return rc;
}

/* ------------------------------------------------------------------ timing */
int mtfhip_timing_enable(mtfhip_ctx *c, int on) {
	if (!c) return fail(MTFHIP_ERR_INVALID_ARG, "timing_enable: NULL ctx");
	c->timing = on != 0;
	c->timing_stride = on > 1 ? on : 1;
	return MTFHIP_OK;
}
static void drain(mtfhip_ctx *c) {
	(void)hipStreamSynchronize(c->stream);
	for (auto &kv : c->timers) {
		for (auto &p : kv.second.pending) {
			float ms = 0;
			if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) { kv.second.total_ms += ms; kv.second.n += 1; }
			c->free_events.push_back(p.first);
			c->free_events.push_back(p.second);
		}
		kv.second.pending.clear();
	}
}
int mtfhip_timing_reset(mtfhip_ctx *c) {
	if (!c) return fail(MTFHIP_ERR_INVALID_ARG, "timing_reset: NULL ctx");
	drain(c);
	for (auto &kv : c->timers) { kv.second.total_ms = 0; kv.second.n = 0; kv.second.launches = 0; }
	return MTFHIP_OK;
}
int mtfhip_timing_get(mtfhip_ctx *c, const char *family, double *avg_ms, int *n_launches) {
	if (!c || !family) return fail(MTFHIP_ERR_INVALID_ARG, "timing_get: NULL argument");
	drain(c);
	auto it = c->timers.find(family);
	double avg = 0; int n = 0;
	if (it != c->timers.end() && it->second.n > 0) { avg = it->second.total_ms / it->second.n; n = it->second.n; }
	if (avg_ms) *avg_ms = avg;
	if (n_launches) *n_launches = n;
	return MTFHIP_OK;
}

} /* extern "C" */
