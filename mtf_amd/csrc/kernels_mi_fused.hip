/*
 * kernels_mi_fused.hip -- the recompute form of the MI Lucas-Kanade iteration (MTFHIP_MATH_FAST, 8 bins)
 * (one of the translation units of libmtfhip.so; conventions and the shared device helpers: mtfhip_device.h, mtfhip_mi_device.h)
 *
 * The materialising form (api_fused.hip::mi_enqueue with the kernels of kernels_mi.hip) runs the fused LK kernel first
 * -- it writes It, dIt_dx and Jt, 88 B/px -- and three more passes read those arrays back: 324 B/px moved for 132 B/px
 * of algorithmic traffic (profiles/r01_mi_kernel_stats.csv: 1.09 ms per iteration of 64 x 400 x 400 pixels).  Here nothing
 * per-pixel is written at all:
 *   pass 1  k_mi_pass_hist   warp + sample It (lean), B-spline windows of It and I0, histogram of It, joint histogram with
 *                            I0 and (self Hessians) with itself                    reads 28 B/px: texels 4, I0 8, grid point 16
 *           k_mi_tables_iter pre-seeding, logs, similarity, the gradient-factor tables (unchanged, kernels_mi.hip)
 *   pass 2  k_mi_pass_grad_hess   warp + sample It AND its gradient again, the steepest-descent row of the pixel and the
 *                            template's row rebuilt from dI0_dx in registers, both gradient vectors (MI.cc:406-415, 432-441),
 *                            their Jacobian products, and the first-order Hessian sums (MI.cc:461-637)   reads 44 B/px (+16 dI0_dx)
 *           k_mi_finish_fast fixed-order sum of the block rows, the Hessian assembly (MI.cc:497-511, 590-600, 626-636) and the
 *                            body of k_finish_track (solve, compositional update, convergence test) in ONE launch
 * 72 B/px instead of 324.  The per-pixel quantities are those of the reference up to the tolerance-mode sampling arithmetic
 * (mtfhip_device.h); the windows themselves are bit-identical given the same pixel value.
 *
 * Bin mode of both passes runs on v_mfma_f64_4x4x4_4b_f64 as in kernels_mi.hip (operand lane l: row / column = l & 3, block =
 * (l >> 2) & 3, k = l >> 4; result lane l: column = l & 3, block = (l >> 2) & 3, row = l >> 4; tools/mfma_layout_test_4x4.hip);
 * the rank-one sum  sum_p hess_term(p) J_p J_p^T  rides on the same staged rows as a ninth block product instead of 36
 * register accumulators per lane.
 */
#include "mtfhip_mi_device.h"
#include "mtfhip_finish_device.h"

namespace mtfhip {

#ifndef MTFHIP_MI_RS
#define MTFHIP_MI_RS 65   /* r03 A/B (profiles/r03_experiments.md): any stride that is not a multiple of 4 -- 65, 66, 67, 69, 70, 73 -- takes pass 1
                            * from 150 to 119 us; 68 (kMiRowMfma), 72 and 80 put the per-lane-row window stores of lanes 4 / 8 / 12 apart on one bank */
#endif
constexpr int kRS = MTFHIP_MI_RS;   /* slab row stride, doubles (build-time knob of tools/r03_mi_stride_ab.sh) */

/* warp one grid point and sample the current image there (tolerance-mode arithmetic); GRAD: also the gradient with
 * respect to the warped coordinates.  Wave-uniform interior path (closed-form gradient); any lane on a cell edge, an
 * integer coordinate or near the border sends the wave through the reference's five-sample finite difference. */
struct MiSample { double it, gx, gy, wx, wy, inv; };
/* stage 1: the warp and the texel fetch of the bilinear cell -- issued one chunk ahead of its use (a wave has at most one
 * other wave on its SIMD to hide an L2 round trip behind, so the fetch is software-pipelined like the fused LK kernel's) */
struct MiTex { double wx, wy, inv, lxd, lyd; float t00, t01, t10, t11; bool ok; };
template <int SSM, bool GRAD>
__device__ __forceinline__ MiTex mi_issue(const ImgView &im, const Warp9 &W, double hx, double hy, double z, bool uz, double eps) {
	MiTex s;
	s.wx = fma(W.m[0], hx, fma(W.m[1], hy, uz ? W.m[2] : W.m[2] * z));
	s.wy = fma(W.m[3], hx, fma(W.m[4], hy, uz ? W.m[5] : W.m[5] * z));
	s.inv = 1.0;
	if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
		s.inv = rcp_fast(fma(W.m[6], hx, fma(W.m[7], hy, uz ? W.m[8] : W.m[8] * z)));
		s.wx *= s.inv; s.wy *= s.inv;
	}
	const int lx = (int)s.wx, ly = (int)s.wy;
	s.lxd = (double)lx; s.lyd = (double)ly;
	s.ok = (s.wx >= 0) & (s.wy >= 0) & (lx < im.w - 1) & (ly < im.h - 1);
	if constexpr (GRAD) s.ok = s.ok & (s.wx - eps > s.lxd) & (s.wx + eps < s.lxd + 1) & (s.wy - eps > s.lyd) & (s.wy + eps < s.lyd + 1);
	const unsigned off = s.ok ? (unsigned)(ly * im.stride + lx) * 4u : 0u;
	const float *r0 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(im.data) + off);
	const float *r1 = r0 + im.stride;
	s.t00 = r0[0]; s.t01 = r0[1]; s.t10 = r1[0]; s.t11 = r1[1];
	return s;
}
/* stage 2: value (and gradient with respect to the warped coordinates) from the fetched cell */
template <int SSM, bool GRAD>
__device__ __forceinline__ MiSample mi_finish(const ImgView &im, const MiTex &tx, double eps, double norm_mult, double norm_add, bool lane_valid) {
	MiSample s;
	s.wx = tx.wx; s.wy = tx.wy; s.inv = tx.inv; s.gx = s.gy = 0.0;
	if (__builtin_amdgcn_ballot_w64(lane_valid & !tx.ok) == 0) {
		double v, bgx, bgy;
		bilin_fast(tx.t00, tx.t01, tx.t10, tx.t11, tx.wx - tx.lxd, tx.wy - tx.lyd, v, bgx, bgy);
		s.it = fma(norm_mult, v, norm_add);
		if constexpr (GRAD) { s.gx = bgx * norm_mult; s.gy = bgy * norm_mult; }
	} else {
		s.it = norm_mult * pix_val(im, s.wx, s.wy) + norm_add;
		if constexpr (GRAD) {   /* utils::getImgGrad, imgUtils.cc:233-254 */
			const double gm = norm_mult / (2 * eps);
			s.gx = (pix_val(im, s.wx + eps, s.wy) - pix_val(im, s.wx - eps, s.wy)) * gm;
			s.gy = (pix_val(im, s.wx, s.wy + eps) - pix_val(im, s.wx, s.wy - eps)) * gm;
		}
	}
	return s;
}

struct MiPassArgs {
	int nb;                 /* 8 */
	int j0_mode;            /* pass 2: 0 no template row needed, 1 rebuilt from dI0_dx, 2 read from J0 */
	int j0_init_variant;    /* with j0_mode 1: Init variant (gradient as is) instead of Warped at identity (gradient / z) */
	int need_dft, need_df0; /* which Jacobian products the search method uses */
	int g_mean;             /* ESM jac_type Original: df_dIt . (J0 + Jt) / 2 */
	int table_off;          /* pass 2, Hessian: MI_T_SELF / MI_T_CURR / MI_T_INIT */
	int transpose_q;
	double grad_eps, norm_mult, norm_add, hist_norm;
	const int *active;
	const double *tb;       /* [B][MI_SIZE] */
	const double *cand_states;   /* candidate mode of pass 1 (k_mi_pass_hist<.., CAND = true>): [n][S] warps of ONE template */
};

/* ---------------------------------------------------------------------------------------------
 * pass 1: histograms of the freshly sampled It (MI.cc:346-367 update, :639-659 self); block rows [8 | 64 | 64 self]
 * ------------------------------------------------------------------------------------------- */
/* slab rows are indexed with (bin + 1): row 0 and rows 9, 10 take the taps of the un-clamped windows that fall outside the
 * histogram (bspl_window4) and are never read by the bin mode */
constexpr int kWinRows = 11;
/* CAND: the candidate axis (PF / NN, SM/src/PF.cc:247-262 with MI as the appearance model): blockIdx.y is a candidate of target 0 --
 * its warp comes from the candidate's state, the template arrays are target 0's */
template <int SSM, bool SELF, bool CAND = false>
__global__ __launch_bounds__(kBlock) void k_mi_pass_hist(BatchView bv, ImgView im, MiPassArgs pa, double *partials, int nblk, int row_len) {
	__shared__ __attribute__((aligned(16))) double slabs[4 * 2 * kWinRows * kRS];
	constexpr int nb = 8;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int t = blockIdx.y;
	if (!CAND && pa.active && !pa.active[t]) return;
	double *wa = slabs + (size_t)wave * 2 * kWinRows * kRS, *wb = wa + kWinRows * kRS;
	const unsigned N = (unsigned)bv.N;
	const bool uz = bv.unit_z != 0;
	Warp9 W;
	if constexpr (CAND) {   /* getWarpFromState (Homography.cc:94-107, Affine.cc:116-130) of candidate t */
		const double *p = pa.cand_states + (size_t)t * (SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6);
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
			W.m[0] = 1 + p[0]; W.m[1] = p[1]; W.m[2] = p[2]; W.m[3] = p[3]; W.m[4] = 1 + p[4]; W.m[5] = p[5]; W.m[6] = p[6]; W.m[7] = p[7]; W.m[8] = 1;
		} else {
			W.m[0] = 1 + p[2]; W.m[1] = p[3]; W.m[2] = p[0]; W.m[3] = p[4]; W.m[4] = 1 + p[5]; W.m[5] = p[1]; W.m[6] = 0; W.m[7] = 0; W.m[8] = 1;
		}
	} else {
		W = load_warp(bv.warps + 9 * t);
	}
	const size_t tt = CAND ? 0 : (size_t)t;   /* whose template */
	const double *pp = bv.buf[uz ? MTFHIP_BUF_INIT_PTS : MTFHIP_BUF_INIT_HXY] + tt * 2 * N;
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z] + tt * N;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + tt * N;
	for (int k2 = 0; k2 < 2 * kWinRows; ++k2) wa[k2 * kRS + lane] = 0.0;   /* the slabs start clean and every chunk leaves them clean */
	double bj8 = 0.0, bs8 = 0.0, bh8 = 0.0;
	const int li = lane & 3, lb = (lane >> 2) & 3, lk = lane >> 4;
	const double one0 = li == 0 ? 1.0 : 0.0;
	const unsigned stride = (unsigned)nblk * kBlock;
	unsigned base = (blockIdx.x * (kBlock / 64) + wave) * 64;
	/* two-stage pipeline: grid point + template value two chunks ahead, warp + texel fetch one chunk ahead */
	double2 q_nx = make_double2(0.0, 0.0); double z_nx = 1.0, i0_nx = 0.0, i0_cur;
	/* every load of the loop is issued unconditionally (the third homogeneous coordinate is fetched from the template when it is
	 * not needed): with guarded loads the compiler cannot count what is in flight and waits with vmcnt(0), which exposes the
	 * full memory latency of the just-issued prefetch in front of every chunk (65 % of the wave cycles in the first version) */
	const double *zsrc = uz ? I0 : iz;
	auto fetch = [&](unsigned i) {
		q_nx = ld_off<double2>(pp, i * 16u); i0_nx = ld_off<double>(I0, i * 8u);
		z_nx = ld_off<double>(zsrc, i * 8u);
	};
	fetch(min(base + lane, N - 1));
	MiTex tx_cur = mi_issue<SSM, false>(im, W, q_nx.x, q_nx.y, z_nx, uz, pa.grad_eps);
	i0_cur = i0_nx;
	fetch(min(base + lane + stride, N - 1));
	for (; base < N; base += stride) {
		const unsigned i = base + lane;
		const double vm = i < N ? 1.0 : 0.0;   /* lanes behind the end of the patch carry zero weights */
		const double i0 = i0_cur;
		const MiTex tx_nx = mi_issue<SSM, false>(im, W, q_nx.x, q_nx.y, z_nx, uz, pa.grad_eps);
		i0_cur = i0_nx;
		fetch(min(i + 2 * stride, N - 1));
		const MiSample sp = mi_finish<SSM, false>(im, tx_cur, pa.grad_eps, pa.norm_mult, pa.norm_add, i < N);
		tx_cur = tx_nx;
		const BsplWin4 a = bspl_window4<false>(sp.it, nb, pa.hist_norm);
		const BsplWin4 b = bspl_window4<false>(i0, nb, pa.hist_norm);
		double *ra = wa + a.row0 * kRS + lane, *rb = wb + b.row0 * kRS + lane;
#pragma unroll
		for (int k = 0; k < 4; ++k) { ra[k * kRS] = a.w[k] * vm; rb[k * kRS] = b.w[k]; }
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int qq = 0; qq < 16; ++qq) {
			const int p = 4 * qq + lk;
			const double av = wa[(1 + 4 * (lb >> 1) + li) * kRS + p], bvv = wb[(1 + 4 * (lb & 1) + li) * kRS + p];
			bj8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bvv, bj8, 0, 0, 0);
			bh8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, one0, bh8, 0, 0, 0);
			if constexpr (SELF) bs8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, wa[(1 + 4 * (lb & 1) + li) * kRS + p], bs8, 0, 0, 0);
		}
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int k = 0; k < 4; ++k) { ra[k * kRS] = 0.0; rb[k * kRS] = 0.0; }   /* leave the slabs clean: 8 stores instead of 22 */
	}
	__syncthreads();
	double *red = slabs;
	constexpr int rl = nb + nb * nb + (SELF ? nb * nb : 0);
	{
		const int r = 4 * (lb >> 1) + (lane >> 4), c = 4 * (lb & 1) + (lane & 3);
		red[wave * rl + nb + r * nb + c] = bj8;
		if constexpr (SELF) red[wave * rl + nb + nb * nb + r * nb + c] = bs8;
		if ((lb & 1) == 0 && (lane & 3) == 0) red[wave * rl + r] = bh8;
	}
	__syncthreads();
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * row_len;
	for (int k2 = threadIdx.x; k2 < rl; k2 += kBlock) dst[k2] = (red[k2] + red[rl + k2]) + (red[2 * rl + k2] + red[3 * rl + k2]);
}

/* ---------------------------------------------------------------------------------------------
 * pass 2: gradient vectors, Jacobian products and the Hessian sums in one sweep.
 * HK: 0 no Hessian pass (constant Hessian), 1 self (A = B = It, MI.cc:515-601), 2 curr (A = It, B = I0, :603-637),
 *     3 init (A = I0, B = It, transposed joint indexing, :461-513).   HROW: the pixel Jacobian the Hessian is taken of:
 *     0 Jt, 1 J0, 2 (J0 + Jt) / 2.
 * block rows: [16: df_dIt . J | df_dI0 . J0] [64: sum_p hess_term J J^T, x-major] [512: Q[(r, c)][s]]
 * ------------------------------------------------------------------------------------------- */
/* Self Hessian (HK = 1, the class default of ESM / FCLK): both windows of a pixel are the window of It, so its 16 live bin
 * pairs are the 4 x 4 block at (fl - 1, fl - 1) of the 8 x 8 pair table -- the dense form spends 64 block products per four
 * pixels on them.  Sorted form: the 64 pixels of a chunk are counting-sorted by fl inside the wave (eight ballots; classes
 * padded to multiples of four with zero-gradient slots), staged with WINDOW-RELATIVE rows (static row index, the slot is the
 * column: no bank conflicts, nothing to zero but the gradient taps), and every group of four same-class pixels costs two
 * block products into that class's accumulators -- Q_rel[fl][k][m][s] -- plus one for sum hess_term J J^T.  ~20 groups x 3
 * = 60 matrix instructions per chunk instead of 144; the classes are folded into the absolute table once per workgroup. */
#ifndef MTFHIP_MI_RS2
#define MTFHIP_MI_RS2 100
#endif
#ifndef MTFHIP_MI_QR
#define MTFHIP_MI_QR 64
#endif
constexpr int kQR = MTFHIP_MI_QR;     /* row stride (doubles) of a wave's absolute table Q[r][c][s]: 64 = dense */
constexpr int kRS2 = MTFHIP_MI_RS2;   /* slot-major row stride: >= 64 + 8 * 3 padded slots; 100 = 4 mod 32 keeps 4 rows x 4 slots on 16 banks */
struct ClassSort { int slot; unsigned long long ends; int total; };   /* ends: byte c = end slot of class c (<= 88) */
__device__ __forceinline__ ClassSort class_sort8(int key /* 0..7, or negative: not placed */) {
	ClassSort cs;
	cs.slot = 0; cs.ends = 0;
	int off = 0;
#pragma unroll
	for (int c = 0; c < 8; ++c) {
		const unsigned long long m = __builtin_amdgcn_ballot_w64(key == c);
		const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
		cs.slot = key == c ? off + rank : cs.slot;
		off += (__builtin_popcountll(m) + 3) & ~3;
		cs.ends |= (unsigned long long)off << (8 * c);
	}
	cs.total = off;
	return cs;
}
constexpr int kMiFastRow = 16 + 64 + 512;
constexpr int kTRows = 12;   /* gradient-factor tables in LDS, indexed with (bin + 1) in both directions, zero borders */
template <int SSM, int HK, int HROW>
__global__ __launch_bounds__(kBlock, 2) void k_mi_pass_grad_hess(BatchView bv, ImgView im, MiPassArgs pa, double *partials, int nblk) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	constexpr int nb = 8;
	constexpr bool SORTED = HK == 1;
	constexpr int SLAB = SORTED ? 17 * kRS2 + 8 * kQR : (HK ? (2 * kWinRows + 9) * kRS : 0);   /* dense: gd[11] | wd[11] | rw[8] | ht ; sorted: d[4] | w[4] | rw[8] | ht | Q[512] */
	__shared__ __attribute__((aligned(16))) double Tc[kTRows * MI_NB], Ti[kTRows * MI_NB], Th[HK == 1 ? kTRows * MI_NB : 1];
	__shared__ __attribute__((aligned(16))) double slabs[HK ? 4 * SLAB : 4 * 16];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int t = blockIdx.y;
	if (pa.active && !pa.active[t]) return;
	const double *tb = pa.tb + (size_t)t * MI_SIZE;
	for (int k = threadIdx.x; k < kTRows * MI_NB; k += kBlock) {
		const int r = k / MI_NB - 1, c = k % MI_NB - 1;
		const bool in = r >= 0 && r < nb && c >= 0 && c < nb;
		Tc[k] = in ? tb[MI_T_CURR + r * MI_NB + c] : 0.0; Ti[k] = in ? tb[MI_T_INIT + r * MI_NB + c] : 0.0;
		if constexpr (HK == 1) Th[k] = in ? tb[MI_T_SELF + r * MI_NB + c] : 0.0;
	}
	const double *Tq = HK == 1 ? Th : (HK == 2 ? Tc : Ti);
	double *gd = slabs + (size_t)wave * SLAB, *wd = gd + kWinRows * kRS, *rw = wd + kWinRows * kRS, *hts = rw + 8 * kRS;
	double *sd = slabs + (size_t)wave * SLAB, *sw = sd + 4 * kRS2, *srw = sw + 4 * kRS2, *sht = srw + 8 * kRS2;   /* sorted form */
	double *qabs = sht + kRS2;   /* this wave's Q[(r, c)][s] */
	if constexpr (SORTED) { for (int k2 = lane; k2 < SLAB; k2 += 64) sd[k2] = 0.0; }
	else if constexpr (HK != 0) { for (int k2 = 0; k2 < 2 * kWinRows + 9; ++k2) gd[k2 * kRS + lane] = 0.0; }
	__syncthreads();
	const unsigned N = (unsigned)bv.N;
	const bool uz = bv.unit_z != 0;
	const Warp9 W = load_warp(bv.warps + 9 * t);
	const double *pp = bv.buf[uz ? MTFHIP_BUF_INIT_PTS : MTFHIP_BUF_INIT_HXY] + (size_t)t * 2 * N;
	const double *ipts = bv.buf[MTFHIP_BUF_INIT_PTS] + (size_t)t * 2 * N;
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * N;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N;
	const double *dI0 = bv.buf[MTFHIP_BUF_DI0_DX] + (size_t)t * 2 * N;
	const double *J0 = bv.buf[MTFHIP_BUF_J0] + (size_t)t * N * S;
	double acc[16];
#pragma unroll
	for (int k = 0; k < 16; ++k) acc[k] = 0.0;
	double cq[8], chs = 0.0;   /* dense: (Rg, Cg, Sg) blocks; sorted: two running accumulators, flushed per class */
#pragma unroll
	for (int k = 0; k < 8; ++k) cq[k] = 0.0;
	const int li = lane & 3, lb = (lane >> 2) & 3, lk = lane >> 4;
	const unsigned stride = (unsigned)nblk * kBlock;
	unsigned base = (blockIdx.x * (kBlock / 64) + wave) * 64;
	/* operands of the next chunk are requested before the current one is processed */
	double2 q_nx, p_nx; double z_nx = 1.0, i0_nx, g0x_nx = 0.0, g0y_nx = 0.0;
	/* all loads unconditional (see pass 1): operands that are not needed are fetched from arrays that are */
	const double *zsrc = uz ? I0 : iz, *gsrc = pa.j0_mode == 1 ? dI0 : I0;
	const unsigned gy_off = pa.j0_mode == 1 ? N * 8u : 0u;
	auto prefetch = [&](unsigned i) {
		q_nx = ld_off<double2>(pp, i * 16u); i0_nx = ld_off<double>(I0, i * 8u);
		p_nx = ld_off<double2>(ipts, i * 16u);
		z_nx = ld_off<double>(zsrc, i * 8u);
		g0x_nx = ld_off<double>(gsrc, i * 8u); g0y_nx = ld_off<double>(gsrc, i * 8u + gy_off);
	};
	/* two-stage pipeline as in pass 1 */
	prefetch(min(base + lane, N - 1));
	MiTex tx_cur = mi_issue<SSM, true>(im, W, q_nx.x, q_nx.y, z_nx, uz, pa.grad_eps);
	double2 p_cur = p_nx; double z_cur = z_nx, i0_cur = i0_nx, g0x_cur = g0x_nx, g0y_cur = g0y_nx;
	prefetch(min(base + lane + stride, N - 1));
	for (; base < N; base += stride) {
		const unsigned i = base + lane;
		const double vm = i < N ? 1.0 : 0.0;   /* lanes behind the end of the patch contribute zeros */
		const double2 pxy = p_cur; const double z = z_cur, i0 = i0_cur, g0x = g0x_cur, g0y = g0y_cur;
		const MiTex tx_nx = mi_issue<SSM, true>(im, W, q_nx.x, q_nx.y, z_nx, uz, pa.grad_eps);
		p_cur = p_nx; z_cur = z_nx; i0_cur = i0_nx; g0x_cur = g0x_nx; g0y_cur = g0y_nx;
		prefetch(min(i + 2 * stride, N - 1));
		const MiSample sp = mi_finish<SSM, true>(im, tx_cur, pa.grad_eps, pa.norm_mult, pa.norm_add, i < N);
		tx_cur = tx_nx;
		const double x = pxy.x, y = pxy.y;
		/* steepest-descent row of the pixel (Homography.cc:252-289, Affine.cc:213-242) */
		double jt[8], j0[8];
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
			const double dwx_dx = fma(-W.m[6], sp.wx, W.m[0]), dwx_dy = fma(-W.m[7], sp.wx, W.m[1]);
			const double dwy_dx = fma(-W.m[6], sp.wy, W.m[3]), dwy_dy = fma(-W.m[7], sp.wy, W.m[4]);
			hom_row_fast(jt, fma(dwx_dx, sp.gx, dwy_dx * sp.gy) * sp.inv, fma(dwx_dy, sp.gx, dwy_dy * sp.gy) * sp.inv, x, y);
		} else {
			const double Ix = fma(sp.gx, W.m[0], sp.gy * W.m[3]), Iy = fma(sp.gx, W.m[1], sp.gy * W.m[4]);
			jt[0] = Ix; jt[1] = Iy; jt[2] = Ix * x; jt[3] = Ix * y; jt[4] = Iy * x; jt[5] = Iy * y; jt[6] = jt[7] = 0.0;
		}
		/* the template's row: rebuilt from dI0_dx as the fused LK kernel does, or read back */
#pragma unroll
		for (int s = 0; s < 8; ++s) j0[s] = 0.0;
		if (pa.j0_mode == 1) {
			if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
				const double inv0 = (pa.j0_init_variant || uz) ? 1.0 : 1.0 / z;
				hom_row_fast(j0, g0x * inv0, g0y * inv0, x, y);
			} else {
				j0[0] = g0x; j0[1] = g0y; j0[2] = g0x * x; j0[3] = g0x * y; j0[4] = g0y * x; j0[5] = g0y * y;
			}
		} else if (pa.j0_mode == 2) {
			const unsigned ic = min(i, N - 1);
#pragma unroll
			for (int s = 0; s < S; ++s) j0[s] = ld_off<double>(J0 + (size_t)s * N, ic * 8u);
		}
		const BsplWin4 a = bspl_window4<HK == 1 || HK == 2>(sp.it, nb, pa.hist_norm);
		const BsplWin4 c0 = bspl_window4<HK == 3>(i0, nb, pa.hist_norm);
		/* df_dIt = sum gradIt(r) matI0(c) T_curr(r, c), df_dI0 = sum gradI0(r) matIt(c) T_init(r, c) (MI.cc:406-415, 432-441),
		 * factored: the inner sums over the second window first.  Taps outside the histogram meet the tables' zero border. */
		double dft = 0, df0 = 0;
#if defined(MTFHIP_MI_ABL) && MTFHIP_MI_ABL >= 2
		dft = a.d[0] + c0.w[1]; df0 = c0.d[2] + a.w[3];
		if (false) {
#else
		if (pa.need_dft) {
#endif
			const double *T0 = Tc + a.row0 * MI_NB + c0.row0;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const double *Tr = T0 + r * MI_NB;
				const double inner = fma(c0.w[3], Tr[3], fma(c0.w[2], Tr[2], fma(c0.w[1], Tr[1], c0.w[0] * Tr[0])));
				dft = fma(a.d[r], inner, dft);
			}
			dft *= vm;
		}
#if defined(MTFHIP_MI_ABL) && MTFHIP_MI_ABL >= 2
		if (false) {
#else
		if (pa.need_df0) {
#endif
			const double *T0 = Ti + c0.row0 * MI_NB + a.row0;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const double *Tr = T0 + r * MI_NB;
				const double inner = fma(a.w[3], Tr[3], fma(a.w[2], Tr[2], fma(a.w[1], Tr[1], a.w[0] * Tr[0])));
				df0 = fma(c0.d[r], inner, df0);
			}
			df0 *= vm;
		}
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const double jg = pa.g_mean ? 0.5 * (j0[s] + jt[s]) : jt[s];
			acc[s] = fma(dft, jg, acc[s]); acc[8 + s] = fma(df0, j0[s], acc[8 + s]);
		}
#ifdef MTFHIP_MI_ABL   /* ablation builds (tools/mi_ablation.sh): 1 no bin mode, 2 no table sums either */
		if constexpr (SORTED) { acc[0] += a.d[0] + a.h[1] + jt[3] + jt[7]; } else
#endif
		if constexpr (SORTED) {
			double hess_term = 0;
			{
				const double *T0 = Tq + a.row0 * MI_NB + a.row0;
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const double *Tr = T0 + r * MI_NB;
					const double inner = fma(a.w[3], Tr[3], fma(a.w[2], Tr[2], fma(a.w[1], Tr[1], a.w[0] * Tr[0])));
					hess_term = fma(a.h[r], inner, hess_term);
				}
			}
			const bool valid = i < N;
			const ClassSort cs = class_sort8(valid ? a.row0 : -1);
			if (valid) {
#pragma unroll
				for (int k = 0; k < 4; ++k) { sd[k * kRS2 + cs.slot] = a.d[k]; sw[k * kRS2 + cs.slot] = a.w[k]; }
#pragma unroll
				for (int s2 = 0; s2 < 8; ++s2) srw[s2 * kRS2 + cs.slot] = jt[s2];
				sht[cs.slot] = hess_term;
			}
			__builtin_amdgcn_wave_barrier();
			/* groups of four slots in slot order; the operands of the next group are requested before the current one's block
			 * products are issued (padding slots carry a zero gradient tap and a zero hess_term: reading one group past the end
			 * is harmless).  At a class boundary the two running accumulators -- Q_rel[fl][k][m][s half] -- are added into this
			 * wave's absolute table: Q[(fl - 1 + k, fl - 1 + m)][s]. */
			const double *pd = sd + lb * kRS2 + lk, *pw = sw + li * kRS2 + lk, *pr0 = srw + li * kRS2 + lk, *pr1 = pr0 + 4 * kRS2;
			const double *pht = sht + lk;
			const bool hx = (lb >> 1) != 0, hy = (lb & 1) != 0;   /* sum hess_term J J^T: block (lb >> 1, lb & 1) of the 8 x 8, from the J operands already held */
			int c = 0;
			while (((cs.ends >> (8 * c)) & 255) == 0) ++c;   /* first class with pixels (a chunk has at least one) */
			int bound = (int)((cs.ends >> (8 * c)) & 255);
			double acc0 = 0.0, acc1 = 0.0;
			/* two groups per trip, each group's operands requested two groups ahead (one group ahead the LDS latency was exposed
			 * behind three block products in every trip).  An odd number of groups ends with an all-zero padding group: slots
			 * behind the last class keep a zero gradient tap and a zero hess_term, and kRS2 leaves room for the look-ahead. */
			auto flush_if = [&](int end) {
				if (end == bound) {
					const int r = c - 1 + lb, cc = c - 1 + lk;   /* result lane: block = k, row = m, column = s in its half */
					if (r >= 0 && r < nb && cc >= 0 && cc < nb) {
						double *qe = qabs + r * kQR + cc * 8 + li;
						qe[0] += acc0; qe[4] += acc1;
					}
					acc0 = 0.0; acc1 = 0.0;
					do { ++c; } while (c < 8 && (int)((cs.ends >> (8 * c)) & 255) <= end);
					bound = c < 8 ? (int)((cs.ends >> (8 * c)) & 255) : 1 << 30;
				}
			};
			double a_d = pd[0], a_w = pw[0], a_r0 = pr0[0], a_r1 = pr1[0], a_ht = pht[0];
			double b_d = pd[4], b_w = pw[4], b_r0 = pr0[4], b_r1 = pr1[4], b_ht = pht[4];
			for (int g = 0; g < cs.total; g += 8) {
				{
					const double av = a_d * a_w;   /* block = gradient tap k, row = weight tap m */
					acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, a_r0, acc0, 0, 0, 0);
					acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, a_r1, acc1, 0, 0, 0);
					chs = __builtin_amdgcn_mfma_f64_4x4x4f64((hx ? a_r1 : a_r0) * a_ht, hy ? a_r1 : a_r0, chs, 0, 0, 0);
				}
				a_d = pd[g + 8]; a_w = pw[g + 8]; a_r0 = pr0[g + 8]; a_r1 = pr1[g + 8]; a_ht = pht[g + 8];
				flush_if(g + 4);
				{
					const double av = b_d * b_w;
					acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b_r0, acc0, 0, 0, 0);
					acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, b_r1, acc1, 0, 0, 0);
					chs = __builtin_amdgcn_mfma_f64_4x4x4f64((hx ? b_r1 : b_r0) * b_ht, hy ? b_r1 : b_r0, chs, 0, 0, 0);
				}
				b_d = pd[g + 12]; b_w = pw[g + 12]; b_r0 = pr0[g + 12]; b_r1 = pr1[g + 12]; b_ht = pht[g + 12];
				flush_if(g + 8);
			}
			__builtin_amdgcn_wave_barrier();
			if (valid) {   /* padding slots must keep a zero gradient tap and a zero hess_term */
#pragma unroll
				for (int k = 0; k < 4; ++k) sd[k * kRS2 + cs.slot] = 0.0;
				sht[cs.slot] = 0.0;
			}
		} else if constexpr (HK != 0) {
			const BsplWin4 &A = HK == 3 ? c0 : a;
			const BsplWin4 &Bw = HK == 1 ? a : (HK == 2 ? c0 : a);
			/* pixel mode: the scalar hess_term and the dense windows (MI.cc:478-496, 574-583, 620-629) */
			double hess_term = 0;
			{
				const double *T0 = Tq + A.row0 * MI_NB + Bw.row0;
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const double *Tr = T0 + r * MI_NB;
					const double inner = fma(Bw.w[3], Tr[3], fma(Bw.w[2], Tr[2], fma(Bw.w[1], Tr[1], Bw.w[0] * Tr[0])));
					hess_term = fma(A.h[r], inner, hess_term);
				}
			}
			double *rg = gd + A.row0 * kRS + lane, *rwd = wd + Bw.row0 * kRS + lane;
#pragma unroll
			for (int k = 0; k < 4; ++k) { rg[k * kRS] = A.d[k] * vm; rwd[k * kRS] = Bw.w[k]; }
#pragma unroll
			for (int s = 0; s < 8; ++s) rw[s * kRS + lane] = HROW == 0 ? jt[s] : (HROW == 1 ? j0[s] : 0.5 * (j0[s] + jt[s]));
			hts[lane] = hess_term * vm;
			__builtin_amdgcn_wave_barrier();
			/* bin mode: Q[(r, c)][s] += grad(r, p) mat(c, p) J[p][s] (MI.cc:484-486, 576-577, 622-623) as eight 4x4x4 block
			 * products per four pixels, and sum_p hess_term J J^T as a ninth on the same J operands */
#pragma unroll
			for (int qq = 0; qq < 16; ++qq) {
				const int p = 4 * qq + lk;
				const double g0 = gd[(1 + lb) * kRS + p], g1 = gd[(5 + lb) * kRS + p];
				const double w0 = wd[(1 + li) * kRS + p], w1 = wd[(5 + li) * kRS + p];
				const double r0 = rw[li * kRS + p], r1 = rw[(4 + li) * kRS + p];
				const double a00 = g0 * w0, a01 = g0 * w1, a10 = g1 * w0, a11 = g1 * w1;
				cq[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a00, r0, cq[0], 0, 0, 0);
				cq[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a00, r1, cq[1], 0, 0, 0);
				cq[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01, r0, cq[2], 0, 0, 0);
				cq[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01, r1, cq[3], 0, 0, 0);
				cq[4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a10, r0, cq[4], 0, 0, 0);
				cq[5] = __builtin_amdgcn_mfma_f64_4x4x4f64(a10, r1, cq[5], 0, 0, 0);
				cq[6] = __builtin_amdgcn_mfma_f64_4x4x4f64(a11, r0, cq[6], 0, 0, 0);
				cq[7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a11, r1, cq[7], 0, 0, 0);
				/* block (Xg, Yg) = (lb >> 1, lb & 1): rows x = 4 Xg + i weighted by hess_term, columns y = 4 Yg + j */
				const double ha = ((lb >> 1) ? r1 : r0) * hts[p], hb = (lb & 1) ? r1 : r0;
				chs = __builtin_amdgcn_mfma_f64_4x4x4f64(ha, hb, chs, 0, 0, 0);
			}
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int k = 0; k < 4; ++k) { rg[k * kRS] = 0.0; rwd[k * kRS] = 0.0; }
		}
	}
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * kMiFastRow;
	__syncthreads();
	{
		double *red = slabs;   /* >= 4 * 16 doubles in every instantiation */
		block_reduce_store<16>(acc, dst, red);
	}
	if constexpr (HK != 0) {
		__syncthreads();
		/* the four waves' Q and H blocks through the (now free) slabs: [4][512 + 64] */
		double *qred = slabs;
		constexpr int ql = 512 + 64;
		if constexpr (SORTED) {
			/* the waves' absolute tables sit behind their slabs; only the H blocks go through qred */
		} else {
#pragma unroll
		for (int a8 = 0; a8 < 8; ++a8) {
			const int r = 4 * (a8 >> 2) + lb, c = 4 * ((a8 >> 1) & 1) + lk, sx = 4 * (a8 & 1) + li;
			const int row_idx = pa.transpose_q ? c * nb + r : r * nb + c;
			qred[wave * ql + 64 + row_idx * 8 + sx] = cq[a8];
		}
		}
		constexpr int hl = SORTED ? 64 : ql;   /* sorted: only the H blocks go through qred, in front of the first wave's Q table */
		static_assert(!SORTED || 4 * 64 <= 17 * kRS2, "H blocks of the four waves must fit in front of the first Q table");
		qred[wave * hl + (4 * (lb >> 1) + lk) * 8 + 4 * (lb & 1) + li] = chs;
		__syncthreads();
		if constexpr (SORTED) {
			for (int k2 = threadIdx.x; k2 < 64; k2 += kBlock)
				dst[16 + k2] = (qred[k2] + qred[64 + k2]) + (qred[128 + k2] + qred[192 + k2]);
			const double *q0 = slabs + 17 * kRS2;
			for (int k2 = threadIdx.x; k2 < 512; k2 += kBlock) {
				const int q = (k2 >> 6) * kQR + (k2 & 63);
				dst[80 + k2] = (q0[q] + q0[SLAB + q]) + (q0[2 * SLAB + q] + q0[3 * SLAB + q]);
			}
		} else {
			for (int k2 = threadIdx.x; k2 < ql; k2 += kBlock)
				dst[16 + k2] = (qred[k2] + qred[ql + k2]) + (qred[2 * ql + k2] + qred[3 * ql + k2]);
		}
	}
}

/* ---------------------------------------------------------------------------------------------
 * candidate mode: the histogram rows of a candidate -> MI (MI.cc:369-381) -> its likelihood (MI.cc:384-387) -> the particle
 * weight (PF.cc:341-365).  One wave per candidate, lane = bin pair; the wave sum is a fixed butterfly.
 * ------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(64) void k_mi_cand_score(int n, int lo, const double *partials, int nblk, int row_len, const double *tb0, double pre_seed,
	double hist_norm, double alpha, int likelihood_func, double measurement_sigma, double max_similarity, double *wts, double *sim) {
	constexpr int nb = 8;
	const int cnd = blockIdx.x, lane = threadIdx.x;
	if (cnd >= n) return;
	const double *p = partials + (size_t)cnd * nblk * row_len;
	const int r = lane >> 3, c = lane & 7;
	const double js = column_sum(p + nb + lane, nblk, row_len);
	const double hs = lane < nb ? column_sum(p + lane, nblk, row_len) : 0.0;
	const double jv = (js + pre_seed) * hist_norm;
	const double lhc_own = lane < nb ? log((hs + nb * pre_seed) * hist_norm) : 0.0;   /* log curr_hist(lane) */
	const double lhc = __shfl(lhc_own, r);
	double part = jv * (log(jv) - lhc - tb0[MI_LOG_INIT + c]);
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
	if (lane == 0) {
		const double f = part;
		const double dd = (1.0 / f) - 1;
		double w = exp(-alpha * dd * dd);
		if (likelihood_func != 0) {
			const double pi = 3.14159265358979323846;
			const double val = max_similarity - f;
			w = likelihood_func == 1 ? (1.0 / sqrt(2 * pi * measurement_sigma)) * exp(-0.5 * val / measurement_sigma) : 1.0 / (1.0 + val);
		}
		if (wts) wts[lo + cnd] = w;
		if (sim) sim[lo + cnd] = f;
	}
}

/* ---------------------------------------------------------------------------------------------
 * finish: block rows -> g, H of the search method -> (device-side loop) solve + update + convergence test.
 * out_H [B][64] column-major S x S Hessian of the pass, out_g [B][16] the two Jacobian products (for iterate's host solve).
 * gmode: 0 ICLK (df_dI0 . J0), 1 FCLK (df_dIt . Jt), 2 ESM Original (df_dIt . Jm), 3 ESM DiffOfJacs.
 * ------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(kBlock) void k_mi_finish_fast(BatchView bv, mtfhip_sm_desc sm, TrackState ts, int have_hess, int transpose_q,
	int joint_off, int hist_off, int sum_h0_host, int gmode, int do_track, const double *partials, int nblk, const double *tb_all,
	double *out_H, double *out_g, double *rows) {
	__shared__ double gs[16], Hs[64], Q[512], fac[64];
	const int t = blockIdx.x, S = bv.S;
	if (do_track && !ts.active[t]) return;   /* (uniform per workgroup) */
	const double *p = partials + (size_t)t * nblk * kMiFastRow;
	const double *tb = tb_all + (size_t)t * MI_SIZE;
	const int ncol = have_hess ? kMiFastRow : 16;
	if (have_hess && threadIdx.x >= kBlock - 64) {
		/* the 64 bin-pair factors (1 / joint - 1 / hist), one per thread of the last wave, while the others sum the block rows:
		 * evaluated inside the assembly loop below they were 64 dependent global loads and 128 divisions per entry of H */
		const int k = threadIdx.x - (kBlock - 64), rr = k >> 3, cc = k & 7;
		const double jv = tb[joint_off + rr * MI_NB + cc];
		const double hv = tb[hist_off + (transpose_q ? cc : rr)];
		fac[k] = (1.0 / jv) - (1.0 / hv);
	}
	for (int k = threadIdx.x; k < ncol; k += kBlock) {
		const double s = column_sum(p + k, nblk, kMiFastRow);
		if (k < 16) gs[k] = s; else if (k < 80) Hs[k - 16] = s; else Q[k - 80] = s;
	}
	__syncthreads();
	double *row = rows + (size_t)t * ACC_COUNT;
	if (threadIdx.x < 64) {
		const int r2 = threadIdx.x >> 3, c2 = threadIdx.x & 7;
		double h = 0.0;
		if (have_hess && r2 < S && c2 < S) {
			const int a = r2 < c2 ? r2 : c2, b2 = r2 < c2 ? c2 : r2;
			h = Hs[a * 8 + b2];
			for (int rr = 0; rr < 8; ++rr)
				for (int cc = 0; cc < 8; ++cc) {
					/* MI.cc:497-511, 590-600, 626-636: joint_hist_jacobian.row(idx)^T row(idx) * (1 / joint - 1 / hist) */
					const double *q = Q + (rr * 8 + cc) * 8;
					h += q[r2] * q[c2] * fac[rr * 8 + cc];
				}
			out_H[(size_t)t * 64 + c2 * S + r2] = h;
		}
		if (r2 <= c2) row[ACC_H + r2 * 8 - (r2 * (r2 - 1)) / 2 + (c2 - r2)] = -h;
		if (threadIdx.x < 16) out_g[(size_t)t * 16 + threadIdx.x] = gs[threadIdx.x];
		if (threadIdx.x < ACC_COUNT - 36) row[36 + threadIdx.x] = 0.0;
	}
	__syncthreads();
	if (threadIdx.x < S) {
		const double gt = gs[threadIdx.x], g0 = gs[8 + threadIdx.x];
		row[ACC_G + threadIdx.x] = gmode == 0 ? g0 : (gmode == 1 ? gt : (gmode == 2 ? 2.0 * gt : gt - g0));
	}
	if (!do_track) return;
	__syncthreads();   /* the row is read back by the same workgroup */
	finish_track_body(bv, sm, ts, rows, 1, t);
}

/* ===================================================================== */
/* launchers                                                              */
/* ===================================================================== */
static MiPassArgs make_args(const MiFastPlan &pl) {
	MiPassArgs pa;
	pa.nb = 8; pa.j0_mode = pl.j0_mode; pa.j0_init_variant = pl.j0_init_variant; pa.need_dft = pl.need_dft; pa.need_df0 = pl.need_df0;
	pa.g_mean = pl.g_mean; pa.table_off = 0; pa.transpose_q = pl.hk == 3;
	pa.grad_eps = pl.grad_eps; pa.norm_mult = pl.norm_mult; pa.norm_add = pl.norm_add; pa.hist_norm = pl.hist_norm;
	pa.active = pl.active; pa.tb = pl.tb; pa.cand_states = nullptr;
	return pa;
}
void launch_mi_pass_hist(const BatchView &bv, const ImgView &im, const MiFastPlan &pl, double *partials, int nblk, int row_len, hipStream_t st) {
	const MiPassArgs pa = make_args(pl);
	const dim3 g = grid2(nblk, bv.B);
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY, self = pl.hk == 1;
	if (hom && self) MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_HOMOGRAPHY, true>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	else if (hom) MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_HOMOGRAPHY, false>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	else if (self) MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_AFFINE, true>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	else MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_AFFINE, false>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
}
/* MI over the candidate axis: candidates [lo, lo + cnt) of dev_states ([.][S]), weights / similarities at their global indices */
void launch_mi_score_candidates(const BatchView &bv, const ImgView &im, const MiFastPlan &pl, const double *dev_states, int lo, int cnt,
	double *partials, int nblk, int row_len, double pre_seed, double alpha, int likelihood_func, double measurement_sigma, double max_similarity,
	double *wts, double *sim, hipStream_t st) {
	if (cnt <= 0) return;
	MiPassArgs pa = make_args(pl);
	pa.active = nullptr;
	pa.cand_states = dev_states + (size_t)lo * bv.S;
	const dim3 g(nblk, cnt);
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_HOMOGRAPHY, false, true>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	else MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_AFFINE, false, true>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	MTFHIP_LAUNCH(k_mi_cand_score, dim3(cnt), dim3(64), 0, st, cnt, lo, (const double *)partials, nblk, row_len, pl.tb, pre_seed, pl.hist_norm, alpha,
		likelihood_func, measurement_sigma, max_similarity, wts, sim);
}
template <int SSM>
static void launch_pass2(const BatchView &bv, const ImgView &im, const MiPassArgs &pa, int hk, int hrow, double *partials, int nblk, hipStream_t st) {
	const dim3 g = grid2(nblk, bv.B);
#define MTFHIP_MI_P2(HK_, HR_) MTFHIP_LAUNCH((k_mi_pass_grad_hess<SSM, HK_, HR_>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk)
	if (hk == 0) MTFHIP_MI_P2(0, 0);
	else if (hk == 1) MTFHIP_MI_P2(1, 0);
	else if (hk == 2 && hrow == 2) MTFHIP_MI_P2(2, 2);
	else if (hk == 2) MTFHIP_MI_P2(2, 0);
	else MTFHIP_MI_P2(3, 1);
#undef MTFHIP_MI_P2
}
void launch_mi_pass_grad_hess(const BatchView &bv, const ImgView &im, const MiFastPlan &pl, double *partials, int nblk, hipStream_t st) {
	const MiPassArgs pa = make_args(pl);
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) launch_pass2<MTFHIP_SSM_HOMOGRAPHY>(bv, im, pa, pl.hk, pl.hrow, partials, nblk, st);
	else launch_pass2<MTFHIP_SSM_AFFINE>(bv, im, pa, pl.hk, pl.hrow, partials, nblk, st);
}
void launch_mi_finish_fast(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts, const MiFastPlan &pl, int gmode, int do_track,
	const double *partials, int nblk, double *out_H, double *out_g, double *rows, hipStream_t st) {
	const int joint = pl.hk == 1 ? MI_SELF_JOINT : MI_JOINT, hist = pl.hk == 3 ? MI_HIST_INIT : MI_HIST_CURR;
	MTFHIP_LAUNCH(k_mi_finish_fast, dim3(bv.B), dim3(kBlock), 0, st, bv, sm, ts, pl.hk != 0, pl.hk == 3, joint, hist, 0, gmode, do_track,
		partials, nblk, pl.tb, out_H, out_g, rows);
}
int mi_fast_row_len() { return kMiFastRow; }

} // namespace mtfhip
