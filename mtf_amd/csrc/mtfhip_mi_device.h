/*
 * mtfhip_mi_device.h -- device helpers shared by the MI translation units (kernels_mi.hip: the per-function and materialising
 * passes; kernels_mi_fused.hip: the recompute passes of the device-side MI iteration)
 */
#ifndef MTFHIP_MI_DEVICE_H
#define MTFHIP_MI_DEVICE_H
#include "mtfhip_device.h"

namespace mtfhip {

/* ---------------------------------------------------------------------------------------------
 * MI (AM/src/MI.cc): cubic B-spline Parzen histograms.  The reference materialises n_bins x N weight /
 * gradient / Hessian matrices and n_bins^2 x N joint-gradient matrices (MI.cc:297-302, 164 MB per
 * 400x400 target); each pixel only touches a 4-bin window, so here the window is recomputed from the
 * pixel value (I0 / It) wherever it is needed and only the n_bins^2-sized tables live in memory.
 * Per-target table block `tb` (doubles): see the MI_* offsets.  "A" is the image whose B-spline gradient /
 * Hessian enters (rows r of the joint table), "B" the one whose plain weights enter (columns c).
 * ------------------------------------------------------------------------------------------- */
/* utils::bSpl3WithGrad Utilities/include/mtf/Utilities/histUtils.h:206-226 (truncated constant kept, :11) */
__device__ __forceinline__ void bspl3_with_grad(double &val, double &diff, double x) {
	const double k2by3 = 0.66666666666;
	val = 0; diff = 0;
	if ((x > -2) && (x <= -1)) { double t = 2 + x; diff = (t * t) / 2; val = (diff * t) / 3; }
	else if ((x > -1) && (x <= 0)) { double t = x / 2; val = k2by3 - x * x * (1 + t); diff = -x * (t + x + 2); }
	else if ((x > 0) && (x <= 1)) { double t = x / 2; val = k2by3 - x * x * (1 - t); diff = x * (t + x - 2); }
	else if ((x > 1) && (x < 2)) { double t = 2 - x; diff = -(t * t) / 2; val = -(diff * t) / 3; }
}
/* utils::bSpl3Hess histUtils.h:271-283 */
__device__ __forceinline__ double bspl3_hess(double x) {
	if ((x > -2) && (x <= -1)) return 2 + x;
	if ((x > -1) && (x <= 0)) return -(3 * x + 2);
	if ((x > 0) && (x <= 1)) return 3 * x - 2;
	if ((x > 1) && (x < 2)) return 2 - x;
	return 0;
}
/* the <= 4-bin window of a pixel value: ids [lo, hi] = std_bspl_ids.row((int)v) (MI.cc:114-117), weights
 * w[k], derivative d[k] (already * -hist_norm_mult as MI.cc:229,360) and second derivative h[k] */
struct BsplWin { int lo, n; double w[4], d[4], h[4]; };
/* piece F of bSpl3WithGrad / bSpl3Hess, F = 0..3 in the order of the reference's if-chain; F >= 4: outside the support */
template <int F>
__device__ __forceinline__ void bspl3_piece(double &val, double &diff, double &hess, double x) {
	const double k2by3 = 0.66666666666;
	if constexpr (F == 0) { double t = 2 + x; diff = (t * t) / 2; val = (diff * t) / 3; hess = 2 + x; }
	else if constexpr (F == 1) { double t = x / 2; val = k2by3 - x * x * (1 + t); diff = -x * (t + x + 2); hess = -(3 * x + 2); }
	else if constexpr (F == 2) { double t = x / 2; val = k2by3 - x * x * (1 - t); diff = x * (t + x - 2); hess = 3 * x - 2; }
	else if constexpr (F == 3) { double t = 2 - x; diff = -(t * t) / 2; val = -(diff * t) / 3; hess = 2 - x; }
	else { val = 0; diff = 0; hess = 0; }
}
/* The window's first bin is lo = max(0, fl - 1), so tap k sits at x_k = lo - v + k: in piece k of the spline when
 * fl >= 1 (x_0 in (-2, -1]) and in piece k + 1 when the window is clamped at bin 0 (fl == 0, x_0 in (-1, 0]) -- the
 * reference's bSpl3WithGradFast<bspl_id> (histUtils.h:176-204) rests on the same fact.  When every active lane of the
 * wave is in one of those two regular situations the pieces are evaluated straight-line (both candidates, one select)
 * instead of walking the four-range if-chain per tap, which diverges across the wave and costs all four pieces anyway.
 * x_k is accumulated by `diff += 1` exactly like MI.cc:232,359; the additions are exact for fl >= 1, and for fl == 0
 * x_2 can round onto the closed end of piece 2 only for v = 1 - 2^-53 (one double), where the two pieces agree to 1e-12. */
__device__ __forceinline__ BsplWin bspl_window(double v, int nb, double norm_mult, bool want_hess) {
	BsplWin s;
	const int fl = (int)v;
	s.lo = max(0, fl - 1);
	const int hi = min(nb - 1, fl + 2);
	s.n = hi - s.lo + 1;
	double diff = s.lo - v;
	const bool sh = fl < 1;
	const bool regular = sh ? ((diff > -1) & (diff <= 0)) : ((diff > -2) & (diff <= -1));
	if (__builtin_amdgcn_ballot_w64(!regular) == 0) {
		double x[4];
		x[0] = diff; x[1] = x[0] + 1; x[2] = x[1] + 1; x[3] = x[2] + 1;
		double v0, d0, h0, v1, d1, h1;
		bspl3_piece<0>(v0, d0, h0, x[0]); bspl3_piece<1>(v1, d1, h1, x[0]);
		s.w[0] = sh ? v1 : v0; s.d[0] = sh ? d1 : d0; s.h[0] = sh ? h1 : h0;
		bspl3_piece<1>(v0, d0, h0, x[1]); bspl3_piece<2>(v1, d1, h1, x[1]);
		s.w[1] = sh ? v1 : v0; s.d[1] = sh ? d1 : d0; s.h[1] = sh ? h1 : h0;
		bspl3_piece<2>(v0, d0, h0, x[2]); bspl3_piece<3>(v1, d1, h1, x[2]);
		s.w[2] = sh ? v1 : v0; s.d[2] = sh ? d1 : d0; s.h[2] = sh ? h1 : h0;
		bspl3_piece<3>(v0, d0, h0, x[3]);
		s.w[3] = sh ? 0.0 : v0; s.d[3] = sh ? 0.0 : d0; s.h[3] = sh ? 0.0 : h0;
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const bool in = k < s.n;
			s.w[k] = in ? s.w[k] : 0.0;
			s.d[k] = in ? s.d[k] * -norm_mult : 0.0;
			s.h[k] = (in && want_hess) ? norm_mult * s.h[k] : 0.0;
		}
		return s;
	}
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		s.w[k] = 0; s.d[k] = 0; s.h[k] = 0;
		if (k < s.n) {
			bspl3_with_grad(s.w[k], s.d[k], diff);
			s.d[k] *= -norm_mult;
			if (want_hess) s.h[k] = norm_mult * bspl3_hess(diff);
			diff += 1;   /* ++curr_diff, MI.cc:232,359 */
		}
	}
	return s;
}
/* The same window with four piece evaluations instead of seven: tap k of the un-clamped window [fl - 1, fl + 2] lies in
 * piece k; when the window is clamped at bin 0 (fl == 0) the first tap falls on the non-existent bin -1 and the other
 * three move down one slot.  The tap positions are accumulated exactly like MI.cc:232,359 (x += 1 from lo - v), every piece
 * is the reference's expression, so weights / derivatives are bit-identical to bspl_window() for every regular value
 * (anything else, NaN included, takes bspl_window itself). */
__device__ __forceinline__ BsplWin bspl_window_fast(double v, int nb, double norm_mult, bool want_hess) {
	const int fl = (int)v;
	const bool sh = fl < 1;
	const int lo = max(0, fl - 1), hi = min(nb - 1, fl + 2);
	const double base = lo - v;
	const bool regular = sh ? ((base > -1) & (base <= 0)) : ((base > -2) & (base <= -1));
	if (__builtin_amdgcn_ballot_w64(!regular) != 0) return bspl_window(v, nb, norm_mult, want_hess);
	BsplWin s;
	s.lo = lo; s.n = hi - lo + 1;
	const double x1 = sh ? base : base + 1, x2 = x1 + 1, x3 = x2 + 1;
	double v0, d0, h0, v1, d1, h1, v2, d2, h2, v3, d3, h3;
	bspl3_piece<0>(v0, d0, h0, base); bspl3_piece<1>(v1, d1, h1, x1); bspl3_piece<2>(v2, d2, h2, x2); bspl3_piece<3>(v3, d3, h3, x3);
	s.w[0] = sh ? v1 : v0; s.w[1] = sh ? v2 : v1; s.w[2] = sh ? v3 : v2; s.w[3] = sh ? 0.0 : v3;
	s.d[0] = sh ? d1 : d0; s.d[1] = sh ? d2 : d1; s.d[2] = sh ? d3 : d2; s.d[3] = sh ? 0.0 : d3;
	s.h[0] = sh ? h1 : h0; s.h[1] = sh ? h2 : h1; s.h[2] = sh ? h3 : h2; s.h[3] = sh ? 0.0 : h3;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const bool in = k < s.n;
		s.w[k] = in ? s.w[k] : 0.0;
		s.d[k] = in ? s.d[k] * -norm_mult : 0.0;
		s.h[k] = (in && want_hess) ? norm_mult * s.h[k] : 0.0;
	}
	return s;
}
/* Tolerance-mode window without clamps, selects or divisions (kernels_mi_fused.hip): the four taps of the UN-clamped
 * window [fl - 1, fl + 2], tap k in piece k at x_k = (fl - 1 - v) + k.  Taps that fall on the non-existent bins -1, nb,
 * nb + 1 are kept: the callers index their slabs / tables with (bin + 1) and give those positions dummy rows (slabs) or
 * zero entries (tables), which is what the reference's clamping of the id range amounts to (MI.cc:114-117, 222-235).
 * row0 = fl = (first bin) + 1.  d is already * -norm_mult, h * norm_mult (MI.cc:229, 360). */
struct BsplWin4 { int row0; double w[4], d[4], h[4]; };
template <bool HESS>
__device__ __forceinline__ BsplWin4 bspl_window4(double v, int nb, double norm_mult) {
	BsplWin4 s;
	const int fl = min(max((int)v, 0), nb - 1);
	s.row0 = fl;
	const double k2by3 = 0.66666666666, third = 1.0 / 3.0;
	const double x0 = (double)(fl - 1) - v, x1 = x0 + 1, x2 = x1 + 1, x3 = x2 + 1;
	const double nm = -norm_mult;
	{ const double t = 2 + x0, df = (t * t) * 0.5; s.w[0] = (df * t) * third; s.d[0] = df * nm; if (HESS) s.h[0] = norm_mult * t; }
	{ const double t = x1 * 0.5; s.w[1] = fma(-(x1 * x1), 1 + t, k2by3); s.d[1] = (x1 * (t + x1 + 2)) * norm_mult; if (HESS) s.h[1] = nm * fma(3.0, x1, 2.0); }
	{ const double t = x2 * 0.5; s.w[2] = fma(-(x2 * x2), 1 - t, k2by3); s.d[2] = (x2 * (t + x2 - 2)) * nm; if (HESS) s.h[2] = norm_mult * fma(3.0, x2, -2.0); }
	{ const double t = 2 - x3, df = (t * t) * 0.5; s.w[3] = (df * t) * third; s.d[3] = df * norm_mult; if (HESS) s.h[3] = norm_mult * t; }
	if (!HESS) { s.h[0] = s.h[1] = s.h[2] = s.h[3] = 0.0; }
	return s;
}
__device__ __forceinline__ void lds_add(double *p, double v) {
	__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
/* ---------------------------------------------------------------------------------------------
 * Bin-owner accumulation.  MI's histograms and the `joint_hist_jacobian` rows are scatter-adds whose targets are
 * decided by pixel intensities; neighbouring pixels hit the same few bins, so LDS atomics serialise almost
 * completely inside a wave (the first version: 682 us for one Hessian of 8 x 160 000 px).  Here the roles are
 * swapped per 64-pixel chunk: in "pixel mode" lane p evaluates pixel p's B-spline windows and writes them as DENSE
 * n_bins vectors to the wave's LDS slab; in "bin mode" lane q owns the bin pair (r, c) = (q / nb, q % nb) and walks
 * the 64 staged pixels, accumulating in registers.  No atomics, no conflicts (lanes with equal r read one address:
 * a broadcast), deterministic sums.  pairs per lane = ceil(nb^2 / 64): 1 for the reference's 8 bins, 4 for 16.
 * ------------------------------------------------------------------------------------------- */
constexpr int kMiPairs = (MI_NB * MI_NB + 63) / 64;
constexpr int kMiRow = 65;
constexpr int kMiRowMfma = 65;   /* (r03: was 68 -- a multiple of 4 puts the window stores of lanes 4 / 8 / 12 apart on one bank, profiles/r03_mi_stride_ab.txt) */
/* Bin mode on the matrix cores.  Over a 64-pixel chunk the bin-mode sums are small dense products whose K axis is the
 * pixel: joint(r, c) = sum_p wa[r][p] wb[c][p] is (nb x 64)(64 x nb), and the joint_hist_jacobian block
 * Q[(r, c)][s] = sum_p (gd[r][p] wd[c][p]) J[p][s] is (nb^2 x 64)(64 x S).  v_mfma_f64_16x16x4_f64 takes K = 4 pixels
 * per issue; operand layout (checked on gfx950 with tools/mfma_layout_test.hip): lane l supplies A[i = l % 16][k = l / 16]
 * and B[k = l / 16][j = l % 16] and receives D[i = l / 16 + 4 v][j = l % 16] in element v of its 4-double accumulator.
 * The operands are read straight from the staged slabs (bin-major rows: lanes of one k read consecutive rows, the same
 * column -> no bank conflict beyond the 2-way of 64-bit reads).  Dense FP64 products are exact in the same sense as the
 * VALU path (fused multiply-add per k); only the summation order over pixels differs (4-pixel groups). */
typedef double mfma_d4 __attribute__((ext_vector_type(4)));


/* the body of k_mi_tables_iter (kernels_mi.hip) -- also the first half of k_mi_tables_poly (kernels_mi_fused.hip), which goes on to the
 * per-class polynomial tables in the same launch.  red: kBlock doubles of LDS. */
__device__ __forceinline__ void mi_tables_iter_body(int nb, double pre_seed, double norm_mult, int with_self, const double *partials,
	int nblk, int row_len, double *tb_all, double *f_out, double *red) {
	const int t = blockIdx.x;
	double *tb = tb_all + (size_t)t * MI_SIZE;
	const double *p = partials + (size_t)t * nblk * row_len;
	const double hist_seed = nb * pre_seed;
	for (int k = threadIdx.x; k < nb + nb * nb * (with_self ? 2 : 1); k += kBlock) {
		const double s = column_sum(p + k, nblk, row_len);
		if (k < nb) {
			const double hv = (s + hist_seed) * norm_mult;
			tb[MI_HIST_CURR + k] = hv; tb[MI_LOG_CURR + k] = log(hv);
		} else if (k < nb + nb * nb) {
			const int q = k - nb, r = q / nb, c = q % nb;
			const double jv = (s + pre_seed) * norm_mult;
			tb[MI_JOINT + r * MI_NB + c] = jv; tb[MI_JOINT_LOG + r * MI_NB + c] = log(jv);
		} else {
			const int q = k - nb - nb * nb, r = q / nb, c = q % nb;
			tb[MI_SELF_JOINT + r * MI_NB + c] = (s + pre_seed) * norm_mult;
		}
	}
	__syncthreads();
	double part = 0;
	for (int q = threadIdx.x; q < nb * nb; q += kBlock) {
		const int r = q / nb, c = q % nb;
		const double jv = tb[MI_JOINT + r * MI_NB + c], lg = tb[MI_JOINT_LOG + r * MI_NB + c];
		part += jv * (lg - tb[MI_LOG_CURR + r] - tb[MI_LOG_INIT + c]);
		tb[MI_T_CURR + r * MI_NB + c] = 1 + lg - tb[MI_LOG_CURR + r];
		tb[MI_T_INIT + r * MI_NB + c] = 1 + tb[MI_JOINT_LOG + c * MI_NB + r] - tb[MI_LOG_INIT + r];   /* (init, curr) indexing */
		if (with_self) tb[MI_T_SELF + r * MI_NB + c] = 1 + log(tb[MI_SELF_JOINT + r * MI_NB + c]) - tb[MI_LOG_CURR + r];
	}
	red[threadIdx.x] = part;
	__syncthreads();
	if (threadIdx.x == 0) {
		double s = 0;
		for (int i = 0; i < kBlock; ++i) s += red[i];
		f_out[t] = s;
	}
}

} // namespace mtfhip
#endif
