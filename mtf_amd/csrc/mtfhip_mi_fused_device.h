/*
 * mtfhip_mi_fused_device.h -- device code shared by the translation units of the MI recompute passes: the tolerance-mode sampler
 * (mi_issue / mi_finish), the pass arguments, and pass 2 (k_mi_pass_grad_hess), which is instantiated per (SSM, channels) in its own
 * unit (kernels_mi_pass2_*.hip -> mtfhip_mi_pass2_unit.h) because one unit with every instantiation took more than three minutes
 * to compile.  Overview of the passes: kernels_mi_fused.hip.
 */
#pragma once
#include <type_traits>
#include "mtfhip_mi_device.h"
#include "mtfhip_finish_device.h"

namespace mtfhip {

#ifndef MTFHIP_MI_RS
#define MTFHIP_MI_RS 65   /* r03 A/B (profiles/r03_experiments.md): any stride that is not a multiple of 4 -- 65, 66, 67, 69, 70, 73 -- takes pass 1
                            * from 150 to 119 us; 68 (kMiRowMfma), 72 and 80 put the per-lane-row window stores of lanes 4 / 8 / 12 apart on one bank */
#endif
#ifndef MTFHIP_MI_OP_AHEAD
#define MTFHIP_MI_OP_AHEAD 1
#endif
#ifndef MTFHIP_MI_TEX_AHEAD
#define MTFHIP_MI_TEX_AHEAD 1
#endif
constexpr int kMiOpAhead = MTFHIP_MI_OP_AHEAD, kMiTexAhead = MTFHIP_MI_TEX_AHEAD;   /* pass 1's software pipeline: chunks of operands / texels in flight (kernels_mi_fused.hip) */
constexpr int kRS = MTFHIP_MI_RS;   /* slab row stride, doubles (build-time knob of tools/r03_mi_stride_ab.sh) */

/* warp one grid point and sample the current image there (tolerance-mode arithmetic); GRAD: also the gradient with
 * respect to the warped coordinates.  Wave-uniform interior path (closed-form gradient); any lane on a cell edge, an
 * integer coordinate or near the border sends the wave through the reference's five-sample finite difference. */
struct MiSample { double it, gx, gy, wx, wy, inv; };
/* stage 1: the warp and the texel fetch of the bilinear cell -- issued one chunk ahead of its use (a wave has at most one
 * other wave on its SIMD to hide an L2 round trip behind, so the fetch is software-pipelined like the fused LK kernel's) */
struct MiTex { double wx, wy, inv, lxd, lyd; float t00, t01, t10, t11; bool ok; };
/* MC: the multi-channel models (MCMI = MI constructed with n_channels = 3, AM/src/MCMI.cc): a row is a (pixel, channel) pair, the
 * pixel's grid point is shared by its rows, the texels of channel ch sit at x * C + ch of the interleaved 32FC3 frame
 * (imgUtils.cc:861-1005) */
/* nonch (the non-chained route, updateGradPts + getWarpedImgGrad): the four finite-difference points are eps times a column of the
 * warp away from the centre, not eps: the interior test takes a bound of that distance as its margin */
template <int SSM, bool GRAD, bool MC = false>
__device__ __forceinline__ MiTex mi_issue(const ImgView &im, const Warp9 &W, double hx, double hy, double z, bool uz, double eps,
	unsigned Cc = 1u, unsigned ch = 0u, bool nonch = false) {
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
	if constexpr (GRAD) {
		double mg = eps;
		if (nonch) {   /* |d(wx, wy)| <= eps (|W00| + |W01| ... + (|W20| + |W21|) max(|wx|, |wy|)) / |D|, doubled */
			const double w1 = fmax(fmax(fabs(W.m[0]), fabs(W.m[1])), fmax(fabs(W.m[3]), fabs(W.m[4])));
			const double w2 = fmax(fabs(W.m[6]), fabs(W.m[7]));
			mg = 2 * eps * fabs(s.inv) * fma(w2, fmax(fabs(s.wx), fabs(s.wy)), w1);
		}
		s.ok = s.ok & (s.wx - mg > s.lxd) & (s.wx + mg < s.lxd + 1) & (s.wy - mg > s.lyd) & (s.wy + mg < s.lyd + 1);
	}
	const unsigned off = s.ok ? (MC ? (unsigned)(ly * im.stride + lx * (int)Cc) + ch : (unsigned)(ly * im.stride + lx)) * 4u : 0u;
	const float *r0 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(im.data) + off);
	const float *r1 = r0 + im.stride;
	s.t00 = r0[0]; s.t01 = r0[MC ? Cc : 1u]; s.t10 = r1[0]; s.t11 = r1[MC ? Cc : 1u];
	return s;
}
/* stage 2: value (and gradient with respect to the warped coordinates) from the fetched cell */
/* nonch (GRAD only; W = the warp): the NON-CHAINED route -- gx, gy are then the gradient with respect to the TEMPLATE coordinates as
 * updateGradPts + getWarpedImgGrad produce it (Homography.cc:803-827 / Affine.cc:293-313, imgUtils.cc:177-202), ready for
 * cmptInitPixJacobian (no chain rule at the caller).  Interior path: the cell's slopes times the ROUNDED steps the reference's four
 * offset points take -- numerators and denominator rounded on their own grids, px0 - px1 = ((n0 - n1) - wx (d0 - d1)) / D to second
 * order in eps -- the QSTEP form of the fused LK body (mtfhip_fused_device.h); r04 ran both routes through the chained form and sat
 * 4.9e-6 (H) / 2.3e-5 (dp) from the non-chained oracle.  The homogeneous coordinates are rebuilt from the warped point and 1 / inv:
 * only their binades matter to the rounded steps. */
template <int SSM, bool GRAD, bool MC = false>
__device__ __forceinline__ MiSample mi_finish(const ImgView &im, const MiTex &tx, double eps, double norm_mult, double norm_add, bool lane_valid,
	int ch = 0, bool nonch = false, const Warp9 &W = Warp9{}) {
	auto pv = [&](double x, double y) -> double { if constexpr (MC) return pix_val_mc(im, x, y, ch); else return pix_val(im, x, y); };
	MiSample s;
	s.wx = tx.wx; s.wy = tx.wy; s.inv = tx.inv; s.gx = s.gy = 0.0;
	if (__builtin_amdgcn_ballot_w64(lane_valid & !tx.ok) == 0) {
		double v, bgx, bgy;
		bilin_fast(tx.t00, tx.t01, tx.t10, tx.t11, tx.wx - tx.lxd, tx.wy - tx.lyd, v, bgx, bgy);
		s.it = fma(norm_mult, v, norm_add);
		if constexpr (GRAD) {   /* the cell's slope times the rounded step of the reference's central difference (fd_step, mtfhip_device.h) */
			const double gm = norm_mult / (2 * eps);
			if (nonch) {
				double dpx_x, dpy_x, dpx_y, dpy_y;
				if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
					const double D = rcp_fast(tx.inv), cx = tx.wx * D, cy = tx.wy * D;
					const double dd_x = fd_step_sym(D, W.m[6] * eps), dd_y = fd_step_sym(D, W.m[7] * eps);
					dpx_x = fma(-tx.wx, dd_x, fd_step_sym(cx, W.m[0] * eps)) * tx.inv; dpy_x = fma(-tx.wy, dd_x, fd_step_sym(cy, W.m[3] * eps)) * tx.inv;
					dpx_y = fma(-tx.wx, dd_y, fd_step_sym(cx, W.m[1] * eps)) * tx.inv; dpy_y = fma(-tx.wy, dd_y, fd_step_sym(cy, W.m[4] * eps)) * tx.inv;
				} else {
					dpx_x = fd_step_sym(tx.wx, W.m[0] * eps); dpy_x = fd_step_sym(tx.wy, W.m[3] * eps);
					dpx_y = fd_step_sym(tx.wx, W.m[1] * eps); dpy_y = fd_step_sym(tx.wy, W.m[4] * eps);
				}
				s.gx = fma(bgx, dpx_x, bgy * dpy_x) * gm; s.gy = fma(bgx, dpx_y, bgy * dpy_y) * gm;
			} else {
				s.gx = bgx * (fd_step(tx.wx, eps) * gm); s.gy = bgy * (fd_step(tx.wy, eps) * gm);
			}
		}
	} else {
		s.it = norm_mult * pv(s.wx, s.wy) + norm_add;
		if constexpr (GRAD) {
			const double gm = norm_mult / (2 * eps);
			if (nonch) {   /* the reference's four offset points, Homography.cc:803-827 / Affine.cc:293-313, sampled one by one (imgUtils.cc:177-202) */
				const double ex0 = W.m[0] * eps, ex1 = W.m[3] * eps, ey0 = W.m[1] * eps, ey1 = W.m[4] * eps;
				if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
					const double D = 1.0 / tx.inv, cx = tx.wx * D, cy = tx.wy * D, ex2 = W.m[6] * eps, ey2 = W.m[7] * eps;
					s.gx = (pv((cx + ex0) / (D + ex2), (cy + ex1) / (D + ex2)) - pv((cx - ex0) / (D - ex2), (cy - ex1) / (D - ex2))) * gm;
					s.gy = (pv((cx + ey0) / (D + ey2), (cy + ey1) / (D + ey2)) - pv((cx - ey0) / (D - ey2), (cy - ey1) / (D - ey2))) * gm;
				} else {
					s.gx = (pv(s.wx + ex0, s.wy + ex1) - pv(s.wx - ex0, s.wy - ex1)) * gm;
					s.gy = (pv(s.wx + ey0, s.wy + ey1) - pv(s.wx - ey0, s.wy - ey1)) * gm;
				}
			} else {   /* utils::getImgGrad, imgUtils.cc:233-254 (mc:: :861-905) */
				s.gx = (pv(s.wx + eps, s.wy) - pv(s.wx - eps, s.wy)) * gm;
				s.gy = (pv(s.wx, s.wy + eps) - pv(s.wx, s.wy - eps)) * gm;
			}
		}
	}
	return s;
}

/* slab rows are indexed with (bin + 1): row 0 and rows 9, 10 take the taps of the un-clamped windows that fall outside the
 * histogram (bspl_window4) and are never read by the bin mode */
constexpr int kWinRows = 11;
struct MiPassArgs {
	int nb;                 /* the AM's n_bins: 8 in the NB = 8 instantiations, <= 10 in the NB = 10 ones */
	int j0_mode;            /* pass 2: 0 no template row needed, 1 rebuilt from dI0_dx, 2 read from J0 */
	int j0_init_variant;    /* with j0_mode 1: Init variant (gradient as is) instead of Warped at identity (gradient / z) */
	int need_dft, need_df0; /* which Jacobian products the search method uses */
	int g_mean;             /* ESM jac_type Original: df_dIt . (J0 + Jt) / 2 */
	int table_off;          /* pass 2, Hessian: MI_T_SELF / MI_T_CURR / MI_T_INIT */
	int transpose_q;
	int nonchained;         /* pass 2: the search method's chained_warp = 0 (updateGradPts + getWarpedImgGrad + cmptInitPixJacobian) */
	int hist_from_joint;    /* pass 1: no histogram product; the histogram row = the row sums of the joint histogram (MiFastPlan) */
	double grad_eps, norm_mult, norm_add, hist_norm;
	const int *active;
	const double *tb;       /* [B][MI_SIZE] */
	const double *poly;     /* [B][kMiPolySize]: the gradient-factor tables as per-class polynomials (k_mi_poly_tables) */
	const double *cand_states;   /* candidate mode of pass 1 (k_mi_pass_hist<.., CAND = true>): [n][S] warps of ONE template */
};

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
#define MTFHIP_MI_RS2 104
#endif
#ifndef MTFHIP_MI_QR
#define MTFHIP_MI_QR 64
#endif
constexpr int kQR = MTFHIP_MI_QR;     /* row stride (doubles) of a wave's absolute table Q[r][c][s]: 64 = dense */
constexpr int kRS2 = MTFHIP_MI_RS2;   /* row stride of the sorted staging rows: >= 96 columns (64 + 8 * 3 padded slots, rounded up to whole steps of 16);
                                        * 104 = 8 mod 32: the four rows x eight columns a half-wave reads in one step land on 32 different 8-byte banks */
static_assert(kRS2 >= 96 && kRS2 % 32 == 8, "sorted staging rows: one step of 16 columns per read, rows 8 banks apart");
/* Physical column of sorted slot s (r04).  A step of the group loop reads 16 consecutive COLUMNS = one class-aligned quad of every
 * block; lane (li, lb, lk) of the 4x4x4 block products works on block lb's quad, member lk.  The block index sits in the LOW bits of
 * the column (a half-wave's eight columns are 0..7 of the window, consecutive, not {0, 1, 4, 5, 8, 9, 12, 13}). */
/* column of member m of block b's quad in step j: the window of step j is columns [16 j, 16 j + 16); the block index is skewed by
 * j / 2 so that consecutive quads of a block -- consecutive lanes of the staging stores -- fall on different banks: member -> 4 m,
 * step parity -> 16, (b + j / 2) & 3 -> the low two bits: eight consecutive quads cover the 32 banks once (measured without the skew:
 * the 17 staging stores ran 3-way conflicted, +90 us on the pass) */
__device__ __forceinline__ int window_col(int j, int b, int m) { return 16 * j + ((b + (j >> 1)) & 3) + 4 * m; }
__device__ __forceinline__ int sorted_col(int s, int qb /* quads per block = steps of the chunk */) {
	const int q = s >> 2, m = s & 3;
	const int b = (q >= qb ? 1 : 0) + (q >= 2 * qb ? 1 : 0) + (q >= 3 * qb ? 1 : 0);   /* block b takes the quads [b qb, (b + 1) qb): contiguous in sorted order */
	const int j = q - b * qb;
	return window_col(j, b, m);
}
__device__ __forceinline__ void lds_add_f64(double *p, double v) {
	(void)__builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)p, v);
}
struct ClassSort { int slot; unsigned long long ends; int total; };   /* ends: byte c = end slot of class c (<= 88) */
/* inclusive prefix sum over the 64 lanes (row_shr 1 / 2 / 4 / 8 inside the rows of 16, then row_bcast:15 / :31 across them) */
__device__ __forceinline__ unsigned wave_scan_u32(unsigned x) {
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
	x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
	return x;
}
/* Counting sort of the wave's keys: eight 8-bit counters packed into two words (a class holds at most 64 pixels, the padded total
 * is at most 88: no byte ever carries), one prefix sum per word gives every lane its rank inside its class and lane 63 the class
 * sizes; sizes padded to multiples of four and multiplied by 0x0101...01 are the classes' end slots.  (r03: eight ballots, each a
 * VALU -> SALU -> VALU round trip, took ~25 us of the pass.) */
__device__ __forceinline__ ClassSort class_sort8(int key /* 0..7, or negative: not placed */) {
	ClassSort cs;
	const unsigned sh = ((unsigned)key & 3u) * 8u;
	const unsigned lo = (key >= 0 && key < 4) ? 1u << sh : 0u, hi = key >= 4 ? 1u << sh : 0u;
	const unsigned slo = wave_scan_u32(lo), shi = wave_scan_u32(hi);
	const unsigned long long counts = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)slo, 63) |
		((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)shi, 63) << 32);
	const unsigned long long padded = (counts + 0x0303030303030303ull) & 0xFCFCFCFCFCFCFCFCull;
	cs.ends = padded * 0x0101010101010101ull;
	const unsigned long long starts = cs.ends - padded;
	const unsigned rank = ((key >= 4 ? shi : slo) >> sh) & 255u;                       /* 1-based */
	const unsigned start = (unsigned)(starts >> (8u * ((unsigned)key & 7u))) & 255u;
	cs.slot = key >= 0 ? (int)(start + rank) - 1 : 0;
	cs.total = (int)(cs.ends >> 56);
	return cs;
}
/* the same for up to twelve classes (the NB = 10 instantiations of pass 2: n_bins 10 as shipped, Config/modules.cfg:115): three words of four
 * 8-bit counters; the padded total is at most 64 + 3 x 10 = 94 (ten classes), no byte carries.  ends12: byte c of (lo | hi << 64-bit) */
struct ClassSort12 { int slot; unsigned e0, e1, e2; int total; };
__device__ __forceinline__ ClassSort12 class_sort12(int key /* 0..11, or negative: not placed */, int n_classes) {
	ClassSort12 cs;
	const unsigned sh = ((unsigned)key & 3u) * 8u;
	const unsigned w0 = (key >= 0 && key < 4) ? 1u << sh : 0u, w1 = (key >= 4 && key < 8) ? 1u << sh : 0u, w2 = key >= 8 ? 1u << sh : 0u;
	const unsigned s0 = wave_scan_u32(w0), s1 = wave_scan_u32(w1), s2 = wave_scan_u32(w2);
	const unsigned c0 = (unsigned)__builtin_amdgcn_readlane((int)s0, 63), c1 = (unsigned)__builtin_amdgcn_readlane((int)s1, 63), c2 = (unsigned)__builtin_amdgcn_readlane((int)s2, 63);
	const unsigned p0 = (c0 + 0x03030303u) & 0xFCFCFCFCu, p1 = (c1 + 0x03030303u) & 0xFCFCFCFCu, p2 = (c2 + 0x03030303u) & 0xFCFCFCFCu;
	/* inclusive byte-wise prefix sums inside a word (x 0x01010101), then the words' totals carried over */
	const unsigned i0 = p0 * 0x01010101u, i1 = p1 * 0x01010101u, i2 = p2 * 0x01010101u;
	const unsigned t0 = i0 >> 24, t1 = t0 + (i1 >> 24);
	cs.e0 = i0; cs.e1 = i1 + t0 * 0x01010101u; cs.e2 = i2 + t1 * 0x01010101u;
	const unsigned ew = key < 4 ? cs.e0 : (key < 8 ? cs.e1 : cs.e2), pw = key < 4 ? p0 : (key < 8 ? p1 : p2), sw = key < 4 ? s0 : (key < 8 ? s1 : s2);
	const unsigned start = ((ew - pw) >> sh) & 255u, rank = (sw >> sh) & 255u;   /* (ew - pw: byte-wise, no borrow: every end >= its class's padded size) */
	cs.slot = key >= 0 ? (int)(start + rank) - 1 : 0;
	cs.total = (int)((n_classes <= 4 ? cs.e0 : (n_classes <= 8 ? cs.e1 : cs.e2)) >> 24);   /* (classes behind n_classes are empty: the top byte of the last used word is the end of everything) */
	return cs;
}
/* r04, the self-Hessian bin mode in MOMENT form.  Both windows of a pixel are the window of It, so gradient tap k and weight tap m
 * are fixed polynomials of ONE number, phi = It - fl (bspl_window4: tap k lies in piece k of the cubic B-spline):
 *   w0 = (1 - phi)^3 / 6        w1 = c23 - phi^2 + phi^3 / 2        w2 = (c23 - 1/2) + (phi + phi^2 - phi^3) / 2        w3 = phi^3 / 6
 *   d0 = -(1 - phi)^2 / 2       d1 = -2 phi + 3 phi^2 / 2           d2 = 1/2 + phi - 3 phi^2 / 2                          d3 = phi^2 / 2     (x -hist_norm' sign folded: d = dw/dIt x norm)
 * (c23 = 0.66666666666, the reference's truncated constant, histUtils.h:11), hence d_k w_m = sum_j C[k][m][j] phi^j, j = 0..5, and
 *   Q_rel[fl][k][m][s] = sum_{p in class fl} d_k w_m J_s = sum_j C[k][m][j] M[fl][j][s],      M[fl][j][s] = sum_p phi_p^j J_s(p).
 * The pass accumulates the 6 x 8 moments per class (48 multiply-adds per pixel instead of 128: four block products per 16 pixels instead
 * of eight, one staged number per pixel instead of eight window taps) and the workgroup turns them into Q once, at the end. */
struct MiMomentCoef { double c[4][4][6]; };
__host__ __device__ constexpr MiMomentCoef mi_moment_coef() {
	constexpr double c23 = 0.66666666666;
	const double w[4][4] = {{1.0 / 6, -0.5, 0.5, -1.0 / 6}, {c23, 0.0, -1.0, 0.5}, {c23 - 0.5, 0.5, 0.5, -0.5}, {0.0, 0.0, 0.0, 1.0 / 6}};   /* coefficients of phi^0..3 */
	const double d[4][3] = {{-0.5, 1.0, -0.5}, {0.0, -2.0, 1.5}, {0.5, 1.0, -1.5}, {0.0, 0.0, 0.5}};                                      /* phi^0..2, x hist_norm */
	MiMomentCoef o{};
	for (int k = 0; k < 4; ++k)
		for (int m = 0; m < 4; ++m) {
			for (int j = 0; j < 6; ++j) o.c[k][m][j] = 0.0;
			for (int a = 0; a < 3; ++a)
				for (int b = 0; b < 4; ++b) o.c[k][m][a + b] += d[k][a] * w[m][b];
		}
	return o;
}
/* r05: the table sums of pass 2 as POLYNOMIALS.  With fl = floor(It), phi = It - fl (and fl0, phi0 of the template value) the taps
 * of a pixel's windows are fixed polynomials of phi / phi0 (above), so
 *   df_dIt = sum_r d_r(phi) sum_c w_c(phi0) T_curr[fl - 1 + r][fl0 - 1 + c] = sum_{a <= 2, b <= 3} PT[fl][fl0][a][b] phi^a phi0^b
 *   df_dI0 = sum_r d_r(phi0) sum_c w_c(phi) T_init[fl0 - 1 + r][fl - 1 + c] = sum_{a <= 2, b <= 3} PI[fl0][fl][a][b] phi0^a phi^b
 *   hess_term = sum_r h_r(phi) sum_c w_c(phi) T_self[fl - 1 + r][fl - 1 + c] = sum_{j <= 4} PH[fl][j] phi^j
 * with coefficients that depend on the iteration's tables only: k_mi_poly_tables builds them once per target and iteration (12 per
 * class pair), pass 2 evaluates 11 + 11 + 4 multiply-adds per pixel by Horner's rule instead of two B-spline windows (~90 instructions)
 * and three 4 x 4 table contractions (60 multiply-adds, 48 LDS reads).  Bins outside the histogram meet the zero border of the
 * tables when the coefficients are built, which is what the reference's clamped id range amounts to (MI.cc:114-117). */
constexpr int kMiPolyPair = 12;                      /* [a = 0..2][b = 0..3] */
/* per instantiation (NB = the bin count the tables are laid out for): [NB^2 class pairs][12] of df_dIt | the same of df_dI0 | [NB][8] of hess_term */
__host__ __device__ constexpr int mi_poly_t(int) { return 0; }
__host__ __device__ constexpr int mi_poly_i(int NB) { return NB * NB * kMiPolyPair; }
__host__ __device__ constexpr int mi_poly_h(int NB) { return 2 * NB * NB * kMiPolyPair; }
__host__ __device__ constexpr int mi_poly_size_nb(int NB) { return 2 * NB * NB * kMiPolyPair + NB * 8; }
__host__ __device__ constexpr int mi_fast_row_nb(int NB) { return 16 + 64 + NB * NB * 8; }   /* block rows of pass 2: g (16) | sum hess_term J J^T (64) | Q[(r, c)][s] */
constexpr int kMiPolyT = 0, kMiPolyI = mi_poly_i(8), kMiPolyH = mi_poly_h(8), kMiPolySize = mi_poly_size_nb(8);
struct MiPolyCoef { double w[4][4], d[4][3], h[4][2]; };   /* coefficients of phi^0.. of tap k's weight, derivative (x -1: d = -dw/dv) and second derivative; x hist_norm outside */
__host__ __device__ constexpr MiPolyCoef mi_poly_coef() {
	constexpr double c23 = 0.66666666666;
	return MiPolyCoef{{{1.0 / 6, -0.5, 0.5, -1.0 / 6}, {c23, 0.0, -1.0, 0.5}, {c23 - 0.5, 0.5, 0.5, -0.5}, {0.0, 0.0, 0.0, 1.0 / 6}},
		{{-0.5, 1.0, -0.5}, {0.0, -2.0, 1.5}, {0.5, 1.0, -1.5}, {0.0, 0.0, 0.5}},
		{{1.0, -1.0}, {-2.0, 3.0}, {1.0, -3.0}, {0.0, 1.0}}};
}
constexpr int kMiFastRow = mi_fast_row_nb(8);
constexpr int kTRows = 12;   /* gradient-factor tables in LDS, indexed with (bin + 1) in both directions, zero borders */
/* NONCH: the search method's chained_warp = 0 (mi_finish's non-chained form + cmptInitPixJacobian rows); its own instantiation so that
 * the chained kernels keep their register budget */
/* NB (r06): the bin count the class tables are laid out for.  8 = the reference's default (parameters.h:344), the r03-r05 kernels unchanged
 * (pa.nb == 8).  10 = the shipped configuration (Config/modules.cfg:115-117: mi_n_bins 10, partition of unity) and every other count up to
 * ten: pa.nb classes and bins at run time, ten-class sort, the moment table M[class][power][s] SHARED by the four waves (ds_add_f64
 * at class boundaries only) so that the workgroup stays under half a CU's LDS -- constant-Hessian and self-Hessian forms (HK 0 / 1). */
template <int SSM, int HK, int HROW, bool MC = false, bool NONCH = false, int NB = 8>
__global__ __launch_bounds__(kBlock, 2) void k_mi_pass_grad_hess(BatchView bv, ImgView im, MiPassArgs pa, double *partials, int nblk) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	static_assert(NB == 8 || (NB == 10 && HK <= 1), "pass 2: 8 bins in every form, up to 10 in the polynomial forms (HK 0 / 1)");
	const int nb = NB == 8 ? 8 : pa.nb;   /* (NB == 8: folded as a constant) */
	constexpr bool SORTED = HK == 1;
	constexpr bool SHARED_M = NB != 8;
	constexpr int kSRows = 15;   /* sorted staging rows: valid phi^0..5 | J[8] | hess_term */
	constexpr int kRowF = mi_fast_row_nb(NB), kPolyI = mi_poly_i(NB), kPolyH = mi_poly_h(NB), kPolySz = mi_poly_size_nb(NB);
	constexpr int SLAB = SORTED ? kSRows * kRS2 + (SHARED_M ? 0 : 8 * kQR) : (HK ? (2 * kWinRows + 9) * kRS : 0);   /* dense: gd[11] | wd[11] | rw[8] | ht ; sorted: valid | phi | rw[8] | ht | M[8 classes][8 powers][8] */
	constexpr bool POLY = HK == 0 || HK == 1;   /* the table sums as per-class polynomials (no window is needed: the bin mode of the self Hessian is in moment form) */
	__shared__ __attribute__((aligned(16))) double Tc[POLY ? 2 : kTRows * MI_NB], Ti[POLY ? 2 : kTRows * MI_NB], Th[1];
	__shared__ __attribute__((aligned(16))) double Pl[POLY ? kPolySz : 2];
	__shared__ __attribute__((aligned(16))) double slabs[HK ? 4 * SLAB + (SORTED && SHARED_M ? NB * kQR : 0) : 4 * 16];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int t = blockIdx.y;
	if (pa.active && !pa.active[t]) return;
	const double *tb = pa.tb + (size_t)t * MI_SIZE;
	if constexpr (POLY) {
		const double *pl = pa.poly + (size_t)t * kPolySz;
		for (int k = threadIdx.x; k < kPolySz; k += kBlock) Pl[k] = pl[k];
	} else {
		for (int k = threadIdx.x; k < kTRows * MI_NB; k += kBlock) {
			const int r = k / MI_NB - 1, c = k % MI_NB - 1;
			const bool in = r >= 0 && r < nb && c >= 0 && c < nb;
			Tc[k] = in ? tb[MI_T_CURR + r * MI_NB + c] : 0.0; Ti[k] = in ? tb[MI_T_INIT + r * MI_NB + c] : 0.0;
		}
	}
	const double *Tq = HK == 1 ? Th : (HK == 2 ? Tc : Ti);
	double *gd = slabs + (size_t)wave * SLAB, *wd = gd + kWinRows * kRS, *rw = wd + kWinRows * kRS, *hts = rw + 8 * kRS;
	double *sd = slabs + (size_t)wave * SLAB, *srw = sd + 6 * kRS2, *sht = srw + 8 * kRS2;   /* sorted form: sd = the six power rows valid phi^k (r05: staged once per pixel instead of rebuilt by every lane of every step) */
	double *qabs = SHARED_M ? slabs + 4 * SLAB : sht + kRS2;   /* this wave's moment table M[class][power][s] (NB != 8: the workgroup's) */
	if constexpr (SORTED) {
		for (int k2 = lane; k2 < SLAB; k2 += 64) sd[k2] = 0.0;
		if constexpr (SHARED_M) { for (int k2 = threadIdx.x; k2 < NB * kQR; k2 += kBlock) qabs[k2] = 0.0; }
	}
	else if constexpr (HK != 0) { for (int k2 = 0; k2 < 2 * kWinRows + 9; ++k2) gd[k2 * kRS + lane] = 0.0; }
	__syncthreads();
#ifdef MTFHIP_MI_DESYNC   /* experiment: the two workgroups that share a CU start half a chunk apart (matrix phase of one under the sampling phase of the other) */
	if (((blockIdx.y * gridDim.x + blockIdx.x) >> 8) & 1) __builtin_amdgcn_s_sleep(MTFHIP_MI_DESYNC);
#endif
	const unsigned N = (unsigned)bv.N;            /* rows: (pixel, channel) pairs */
	const unsigned NPt = MC ? (unsigned)bv.NP : N, Cc = MC ? (unsigned)bv.C : 1u;
	auto pix_of = [&](unsigned i) -> unsigned { if constexpr (MC) return Cc == 3u ? i / 3u : i / Cc; else return i; };
	const bool uz = bv.unit_z != 0;
	const Warp9 W = load_warp(bv.warps + 9 * t);
	const double *pp = bv.buf[uz ? MTFHIP_BUF_INIT_PTS : MTFHIP_BUF_INIT_HXY] + (size_t)t * 2 * NPt;
	const double *ipts = bv.buf[MTFHIP_BUF_INIT_PTS] + (size_t)t * 2 * NPt;
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * NPt;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N;
	const double *dI0 = bv.buf[MTFHIP_BUF_DI0_DX] + (size_t)t * 2 * N;
	const double *J0 = bv.buf[MTFHIP_BUF_J0] + (size_t)t * N * S;
	double acc[16];
#pragma unroll
	for (int k = 0; k < 16; ++k) acc[k] = 0.0;
	double cq[8], chs = 0.0;   /* dense: (Rg, Cg, Sg) blocks */
	double chs4[4] = {0.0, 0.0, 0.0, 0.0};   /* sorted: the four 4 x 4 tiles of sum hess_term J J^T, per block = per quad of pixels (summed over the blocks at the end) */
	/* sorted: the moments M[j][s] = sum phi^j J_s of the class this lane's block is in; carried ACROSS chunks (a block leaves its
	 * class only when its next quad is of another one) */
	double q00 = 0, q01 = 0, q10 = 0, q11 = 0;   /* M[class][power 4 T + lk][s = 4 h + li]: q<T><h> */
	int cls = NB;   /* NB: none */
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
	unsigned ch_nx = 0;   /* MC: the channel of the row whose operands q_nx ... hold */
	auto prefetch = [&](unsigned i) {
		const unsigned pi = pix_of(i);
		if constexpr (MC) ch_nx = i - pi * Cc;
		q_nx = ld_off<double2>(pp, pi * 16u); i0_nx = ld_off<double>(I0, i * 8u);
		p_nx = ld_off<double2>(ipts, pi * 16u);
		z_nx = ld_off<double>(zsrc, (uz ? i : pi) * 8u);
		g0x_nx = ld_off<double>(gsrc, i * 8u); g0y_nx = ld_off<double>(gsrc, i * 8u + gy_off);
	};
	/* two-stage pipeline as in pass 1 */
	prefetch(min(base + lane, N - 1));
	constexpr bool nonch = NONCH;
	MiTex tx_cur = mi_issue<SSM, true, MC>(im, W, q_nx.x, q_nx.y, z_nx, uz, pa.grad_eps, Cc, ch_nx, nonch);
	unsigned ch_cur = ch_nx;
	double2 p_cur = p_nx; double z_cur = z_nx, i0_cur = i0_nx, g0x_cur = g0x_nx, g0y_cur = g0y_nx;
	prefetch(min(base + lane + stride, N - 1));
	for (; base < N; base += stride) {
		const unsigned i = base + lane;
		const double vm = i < N ? 1.0 : 0.0;   /* lanes behind the end of the patch contribute zeros */
		const double2 pxy = p_cur; const double z = z_cur, i0 = i0_cur, g0x = g0x_cur, g0y = g0y_cur;
		/* the template's row when it is read back (j0_mode 2): requested FIRST, so that the wait in front of its use leaves the
		 * texel fetch of the next chunk and the operand prefetch behind it in flight (vmcnt counts in order; requested where it
		 * is used, the wait was vmcnt(0) -- and because the two forms share registers, the rebuilt form waited as well) */
		double j0[8];
#pragma unroll
		for (int s = 0; s < 8; ++s) j0[s] = 0.0;
		if (pa.j0_mode == 2) {
			const unsigned ic = min(i, N - 1);
#pragma unroll
			for (int s = 0; s < S; ++s) j0[s] = ld_off<double>(J0 + (size_t)s * N, ic * 8u);
		}
		const MiTex tx_nx = mi_issue<SSM, true, MC>(im, W, q_nx.x, q_nx.y, z_nx, uz, pa.grad_eps, Cc, ch_nx, nonch);
		const unsigned ch_here = ch_cur;
		ch_cur = ch_nx;
		p_cur = p_nx; z_cur = z_nx; i0_cur = i0_nx; g0x_cur = g0x_nx; g0y_cur = g0y_nx;
		prefetch(min(i + 2 * stride, N - 1));
		const MiSample sp = mi_finish<SSM, true, MC>(im, tx_cur, pa.grad_eps, pa.norm_mult, pa.norm_add, i < N, (int)ch_here, nonch, W);
		tx_cur = tx_nx;
		const double x = pxy.x, y = pxy.y;
		/* steepest-descent row of the pixel (Homography.cc:252-289, Affine.cc:213-242) */
		double jt[8];
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
			double Ix, Iy;
			if (nonch) { Ix = sp.gx; Iy = sp.gy; }   /* cmptInitPixJacobian (Homography.cc:157-191): the gradient is already with respect to the template */
			else {
				const double dwx_dx = fma(-W.m[6], sp.wx, W.m[0]), dwx_dy = fma(-W.m[7], sp.wx, W.m[1]);
				const double dwy_dx = fma(-W.m[6], sp.wy, W.m[3]), dwy_dy = fma(-W.m[7], sp.wy, W.m[4]);
				Ix = fma(dwx_dx, sp.gx, dwy_dx * sp.gy) * sp.inv; Iy = fma(dwx_dy, sp.gx, dwy_dy * sp.gy) * sp.inv;
			}
			hom_row_fast(jt, Ix, Iy, x, y);
		} else {
			/* (Affine.cc:160-182 / :213-242) */
			const double Ix = nonch ? sp.gx : fma(sp.gx, W.m[0], sp.gy * W.m[3]), Iy = nonch ? sp.gy : fma(sp.gx, W.m[1], sp.gy * W.m[4]);
			jt[0] = Ix; jt[1] = Iy; jt[2] = Ix * x; jt[3] = Ix * y; jt[4] = Iy * x; jt[5] = Iy * y; jt[6] = jt[7] = 0.0;
		}
		/* the template's row: rebuilt from dI0_dx as the fused LK kernel does, or read back */
		if (pa.j0_mode == 1) {
			if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
				const double inv0 = (pa.j0_init_variant || uz) ? 1.0 : 1.0 / z;
				hom_row_fast(j0, g0x * inv0, g0y * inv0, x, y);
			} else {
				j0[0] = g0x; j0[1] = g0y; j0[2] = g0x * x; j0[3] = g0x * y; j0[4] = g0y * x; j0[5] = g0y * y;
			}
		}
		BsplWin4 a, c0;
		double dft = 0, df0 = 0;
		int fl_it = 0; double phi_it = 0.0, hess_term = 0.0;
		if constexpr (POLY) {
			/* class and fraction of both pixel values; the three table sums by Horner's rule on the class pair's coefficients (see kMiPoly*) */
			fl_it = min(max((int)sp.it, 0), nb - 1); phi_it = sp.it - (double)fl_it;
			const int fl0 = min(max((int)i0, 0), nb - 1);
			constexpr int kPairStride = NB;   /* class pairs are laid out [NB][NB] whatever pa.nb is */
			const double phi0 = i0 - (double)fl0;
#if defined(MTFHIP_MI_ABL) && MTFHIP_MI_ABL == 2
			dft = phi_it + phi0; df0 = phi0;
#else
			if (pa.need_dft) {
				const double *c = Pl + kMiPolyT + (fl_it * kPairStride + fl0) * kMiPolyPair;
				const double r0 = fma(fma(fma(c[3], phi0, c[2]), phi0, c[1]), phi0, c[0]);
				const double r1 = fma(fma(fma(c[7], phi0, c[6]), phi0, c[5]), phi0, c[4]);
				const double r2 = fma(fma(fma(c[11], phi0, c[10]), phi0, c[9]), phi0, c[8]);
				dft = fma(fma(r2, phi_it, r1), phi_it, r0) * vm;
			}
			if (pa.need_df0) {
				const double *c = Pl + kPolyI + (fl0 * kPairStride + fl_it) * kMiPolyPair;
				const double r0 = fma(fma(fma(c[3], phi_it, c[2]), phi_it, c[1]), phi_it, c[0]);
				const double r1 = fma(fma(fma(c[7], phi_it, c[6]), phi_it, c[5]), phi_it, c[4]);
				const double r2 = fma(fma(fma(c[11], phi_it, c[10]), phi_it, c[9]), phi_it, c[8]);
				df0 = fma(fma(r2, phi0, r1), phi0, r0) * vm;
			}
#endif
			if constexpr (SORTED) {
				const double *c = Pl + kPolyH + fl_it * 8;
				hess_term = fma(fma(fma(fma(c[4], phi_it, c[3]), phi_it, c[2]), phi_it, c[1]), phi_it, c[0]);
			}
		} else {
		a = bspl_window4<HK == 1 || HK == 2>(sp.it, nb, pa.hist_norm);
		c0 = bspl_window4<HK == 3>(i0, nb, pa.hist_norm);
		/* df_dIt = sum gradIt(r) matI0(c) T_curr(r, c), df_dI0 = sum gradI0(r) matIt(c) T_init(r, c) (MI.cc:406-415, 432-441),
		 * factored: the inner sums over the second window first.  Taps outside the histogram meet the tables' zero border. */
		if (pa.need_dft) {
			const double *T0 = Tc + a.row0 * MI_NB + c0.row0;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const double *Tr = T0 + r * MI_NB;
				const double inner = fma(c0.w[3], Tr[3], fma(c0.w[2], Tr[2], fma(c0.w[1], Tr[1], c0.w[0] * Tr[0])));
				dft = fma(a.d[r], inner, dft);
			}
			dft *= vm;
		}
		if (pa.need_df0) {
			const double *T0 = Ti + c0.row0 * MI_NB + a.row0;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const double *Tr = T0 + r * MI_NB;
				const double inner = fma(a.w[3], Tr[3], fma(a.w[2], Tr[2], fma(a.w[1], Tr[1], a.w[0] * Tr[0])));
				df0 = fma(c0.d[r], inner, df0);
			}
			df0 *= vm;
		}
		}
#pragma unroll
		for (int s = 0; s < 8; ++s) {
			const double jg = pa.g_mean ? 0.5 * (j0[s] + jt[s]) : jt[s];
			acc[s] = fma(dft, jg, acc[s]); acc[8 + s] = fma(df0, j0[s], acc[8 + s]);
		}
#if defined(MTFHIP_MI_ABL) && MTFHIP_MI_ABL <= 2   /* ablation builds (tools/mi_ablation.sh): 1 no bin mode, 2 no table sums either */
		if constexpr (SORTED) { acc[0] += hess_term + jt[3] + jt[7]; } else
#endif
		if constexpr (SORTED) {
			const bool valid = i < N;
#if defined(MTFHIP_MI_ABL) && MTFHIP_MI_ABL == 4   /* ablation: no sort (one class of 64 slots) */
			ClassSort cs; cs.slot = lane; cs.ends = 0x4040404040404040ull; cs.total = 64;
#else
			using SortT = typename std::conditional<NB == 8, ClassSort, ClassSort12>::type;
			SortT cs;
			if constexpr (NB == 8) cs = class_sort8(valid ? fl_it : -1);
			else cs = class_sort12(valid ? fl_it : -1, nb);
#endif
			const int steps = (cs.total + 15) >> 4;   /* quads per block */
			const int mycol = sorted_col(cs.slot, steps);
			if (valid) {
				const double ph = phi_it;   /* phi = It - fl: every tap of the window is a polynomial of it */
				const double ph2 = ph * ph, ph3 = ph2 * ph, ph4 = ph2 * ph2, ph5 = ph4 * ph;
				sd[mycol] = 1.0; sd[kRS2 + mycol] = ph; sd[2 * kRS2 + mycol] = ph2; sd[3 * kRS2 + mycol] = ph3; sd[4 * kRS2 + mycol] = ph4; sd[5 * kRS2 + mycol] = ph5;
#pragma unroll
				for (int s2 = 0; s2 < 8; ++s2) srw[s2 * kRS2 + mycol] = jt[s2];
				sht[mycol] = hess_term;
			}
			__builtin_amdgcn_wave_barrier();
			/* r04: steps of SIXTEEN pixels, moment form.  The four blocks of a 4x4x4 product are four quads of pixels, so a lane works on
			 * ONE pixel per step -- its block's quad, member lk -- and reads that pixel's validity, phi, J[li], J[4 + li] and hess_term:
			 * 5 operands per lane per 16 pixels (r03's tap-block form: 5 per 4).  Seven block products per step: the moments
			 * M[4 T + i][4 h + j] += (valid phi^(4 T + i)) J[4 h + j] (T: powers 0-3 | 4, 5; h: the halves of J) and the three upper tiles of
			 * sum hess_term J J^T.  A block's accumulators belong to the class of ITS quad: block b takes the quads [b steps, (b + 1) steps)
			 * of the sorted order, crosses each class boundary of its range once, and when its next quad is of another class its sums
			 * go to the wave's moment table through ds_add_f64.  Slots behind a class's last pixel have valid = 0 and hess_term = 0. */
			/* A operands: row li of the low tile = valid phi^li, row li of the high tile = valid phi^(4 + li) for li < 2; rows 2, 3 of the high
			 * tile are padding -- their results (rows 2, 3 of q10 / q11) are never flushed, so those lanes may read any finite row */
			const double *plo = sd + li * kRS2, *phi_ = sd + (4 + (li & 1)) * kRS2, *pja = srw + li * kRS2, *pjb = pja + 4 * kRS2, *pht = sht;   /* + this lane's column of the step */
			/* class of the quad that starts at slot s0 = number of classes that end at or before it (empty classes included: they end
			 * where their predecessor does); byte-wise on the packed end slots, no borrow between bytes: (s0 | 0x80) - end >= 0x80 - 88 > 0 */
			unsigned ends_lo, ends_hi, ends_x = 0xFFFFFFFFu;   /* (ends_x: the third word of the ten-class sort; 0xFF bytes are never "at or before") */
			if constexpr (NB == 8) { ends_lo = (unsigned)cs.ends; ends_hi = (unsigned)(cs.ends >> 32); }
			else { ends_lo = cs.e0; ends_hi = cs.e1; ends_x = cs.e2; }
			auto class_of = [&](int s0) -> int {
				const unsigned S = (unsigned)s0 * 0x01010101u | 0x80808080u;
				int c = __builtin_popcount((S - ends_lo) & 0x80808080u) + __builtin_popcount((S - ends_hi) & 0x80808080u);
				if constexpr (NB != 8) c += __builtin_popcount((S - ends_x) & 0x80808080u);
				return c;
			};
			/* leave class `cls` for `to`: M[cls][4 T + lk][4 h + li] += q<T><h>.  Result layout of the block product: column = li, block = lb,
			 * row = lk. */
			auto leave_class = [&](int to) {
#if !(defined(MTFHIP_MI_ABL) && MTFHIP_MI_ABL == 5)   /* ablation 5: class boundaries without the table update */
				if (cls < NB) {
					double *qe = qabs + cls * 64 + lk * 8 + li;
					lds_add_f64(qe, q00); lds_add_f64(qe + 4, q01);
					if (lk < 2) { lds_add_f64(qe + 32, q10); lds_add_f64(qe + 36, q11); }   /* powers 4, 5 */
				}
#endif
				q00 = q01 = q10 = q11 = 0.0;
				cls = to;
			};
			int o = window_col(0, lb, lk);
			double c_lo = plo[o], c_hi = phi_[o], c_ja = pja[o], c_jb = pjb[o], c_ht = pht[o];
			const int s_first = 4 * lb * steps;   /* first slot of this block's range */
			{
				const int c_first = class_of(s_first);
				/* (a range that starts behind the last class -- class NB -- holds zeros: the block keeps what it has) */
				if (c_first != cls && c_first < NB) leave_class(c_first);
			}
#if defined(MTFHIP_MI_ABL) && MTFHIP_MI_ABL == 3   /* ablation: sort and stores, no block products */
			for (int j = 0; j < 0; ++j) {
#else
#pragma unroll 1
			for (int j = 0; j < steps; ++j) {
#endif
				/* operands of the next step first (one step past the end reads the columns behind the window: discarded) */
				o = window_col(j + 1, lb, lk);
				const double n_lo = plo[o], n_hi = phi_[o], n_ja = pja[o], n_jb = pjb[o], n_ht = pht[o];
				{
					const double a_lo = c_lo, a_hi = c_hi;
					q00 = __builtin_amdgcn_mfma_f64_4x4x4f64(a_lo, c_ja, q00, 0, 0, 0); q01 = __builtin_amdgcn_mfma_f64_4x4x4f64(a_lo, c_jb, q01, 0, 0, 0);
					q10 = __builtin_amdgcn_mfma_f64_4x4x4f64(a_hi, c_ja, q10, 0, 0, 0); q11 = __builtin_amdgcn_mfma_f64_4x4x4f64(a_hi, c_jb, q11, 0, 0, 0);
					const double ha = c_ja * c_ht, hb = c_jb * c_ht;   /* sum hess_term J J^T: tile (X, Y) = rows 4 X + i weighted, columns 4 Y + j */
					chs4[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(ha, c_ja, chs4[0], 0, 0, 0); chs4[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(ha, c_jb, chs4[1], 0, 0, 0);
					chs4[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(hb, c_jb, chs4[3], 0, 0, 0);   /* (tile (1, 0) is the transpose of (0, 1): mirrored at the end) */
				}
				if (j + 1 < steps) {
					const int cls_nx = class_of(s_first + 4 * (j + 1));
					if (cls_nx != cls && cls_nx < NB) leave_class(cls_nx);
				}
				c_lo = n_lo; c_hi = n_hi; c_ja = n_ja; c_jb = n_jb; c_ht = n_ht;
			}
			__builtin_amdgcn_wave_barrier();
			if (valid) {   /* padding slots must keep zero power rows (valid = 0) and a zero hess_term */
#pragma unroll
				for (int k2 = 0; k2 < 6; ++k2) sd[k2 * kRS2 + mycol] = 0.0;
				sht[mycol] = 0.0;
			}
		} else if constexpr (HK != 0) {
			const BsplWin4 &A = HK == 3 ? c0 : a;
			const BsplWin4 &Bw = HK == 1 ? a : (HK == 2 ? c0 : a);
			/* pixel mode: the scalar hess_term and the dense windows (MI.cc:478-496, 574-583, 620-629) */
			hess_term = 0;
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
	if constexpr (SORTED) {   /* what the blocks still hold goes to the wave's table */
		if (cls < NB) {
			double *qe = qabs + cls * 64 + lk * 8 + li;
			lds_add_f64(qe, q00); lds_add_f64(qe + 4, q01);
			if (lk < 2) { lds_add_f64(qe + 32, q10); lds_add_f64(qe + 36, q11); }
		}
	}
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * kRowF;
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
		static_assert(!SORTED || 4 * 64 <= kSRows * kRS2, "H blocks of the four waves must fit in front of the first moment table");
		if constexpr (SORTED) {
			/* (block_reduce_store above used the first 64 doubles as its scratch) */
			for (int k2 = threadIdx.x; k2 < 4 * 64; k2 += kBlock) qred[k2] = 0.0;
			__syncthreads();
			/* tile (X, Y) of every block: row 4 X + lk, column 4 Y + li.  The four blocks of a wave meet through ds_add_f64; the target --
			 * the first 256 doubles of the slabs, nobody's staging rows any more */
#pragma unroll
			for (int xy = 0; xy < 4; ++xy) if (xy != 2) lds_add_f64(qred + wave * hl + (4 * (xy >> 1) + lk) * 8 + 4 * (xy & 1) + li, chs4[xy]);
		} else {
			qred[wave * hl + (4 * (lb >> 1) + lk) * 8 + 4 * (lb & 1) + li] = chs;
		}
		__syncthreads();
		if constexpr (SORTED) {
			for (int k2 = threadIdx.x; k2 < 64; k2 += kBlock) {
				const int r = k2 >> 3, c = k2 & 7, src = (r >= 4 && c < 4) ? c * 8 + r : k2;   /* the lower-left tile from the upper-right one */
				dst[16 + k2] = (qred[src] + qred[64 + src]) + (qred[128 + src] + qred[192 + src]);
			}
			/* the four waves' moment tables -> one (in the first wave's, in place), then Q[(r, c)][s] = hist_norm sum_fl sum_j C[k][m][j] M[fl][j][s]
			 * over the classes whose window holds both bins: k = r - (fl - 1), m = c - (fl - 1) in 0..3 */
			double *m0 = SHARED_M ? slabs + 4 * SLAB : slabs + kSRows * kRS2;
			if constexpr (!SHARED_M) {
				double msum[2];
#pragma unroll
				for (int u = 0; u < 2; ++u) { const int k2 = threadIdx.x + u * kBlock; msum[u] = (m0[k2] + m0[SLAB + k2]) + (m0[2 * SLAB + k2] + m0[3 * SLAB + k2]); }
				__syncthreads();
#pragma unroll
				for (int u = 0; u < 2; ++u) m0[threadIdx.x + u * kBlock] = msum[u];
				__syncthreads();
			}
			constexpr MiMomentCoef CF = mi_moment_coef();
			for (int k2 = threadIdx.x; k2 < NB * NB * 8; k2 += kBlock) {
				const int r = NB == 8 ? k2 >> 6 : k2 / (NB * 8), c = NB == 8 ? (k2 >> 3) & 7 : (k2 >> 3) % NB, sx = k2 & 7;
				double qv = 0.0;
#pragma unroll
				for (int k = 0; k < 4; ++k) {
					const int fl = r + 1 - k, m = c - (fl - 1);
					if (fl >= 0 && fl < nb && m >= 0 && m < 4) {
						const double *mm = m0 + fl * 64 + sx;
						double acc6 = 0.0;
#pragma unroll
						for (int j = 0; j < 6; ++j) {
							double cf = 0.0;   /* C[k][m][j]: k static, m dynamic */
#pragma unroll
							for (int mq = 0; mq < 4; ++mq) cf = m == mq ? CF.c[k][mq][j] : cf;
							acc6 = fma(cf, mm[j * 8], acc6);
						}
						qv += acc6;
					}
				}
				dst[80 + k2] = qv * pa.hist_norm;
			}
		} else {
			for (int k2 = threadIdx.x; k2 < ql; k2 += kBlock)
				dst[16 + k2] = (qred[k2] + qred[ql + k2]) + (qred[2 * ql + k2] + qred[3 * ql + k2]);
		}
	}
}

} // namespace mtfhip
