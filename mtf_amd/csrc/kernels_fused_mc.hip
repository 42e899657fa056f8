/*
 * kernels_fused_mc.hip -- the fused Lucas-Kanade iteration of the multi-channel appearance models (MCSSD, MCNCC: SSD / NCC with
 * n_channels = 3, AM/src/MCSSD.cc, AM/src/MCNCC.cc over Utilities/src/imgUtils.cc:861-1005).  One of the translation units of
 * libmtfhip.so; the body is fused_lk_body<..., MC = true> (mtfhip_fused_device.h): one launch per iteration, a thread per
 * (pixel, channel) row, the pixel's grid point shared by its C rows, the partial rows those of the single-channel pass
 * (so k_finish_track, the host assembly and the NCC moment algebra serve both).  Launches that materialise nothing take the
 * tolerance-mode form (k_fused_mc_fast) unless the batch is in replay arithmetic.
 */
#include "mtfhip_fused_device.h"

namespace mtfhip {

template <int AM, int SSM, bool CHAINED, int MODE, bool MAT>
__global__ __launch_bounds__(kBlock, MTFHIP_FUSED_WAVES) void k_fused_mc(BatchView bv, ImgView im, FusedArgs fa, double *partials, int nblk) {
	fused_lk_body<AM, SSM, CHAINED, MODE, MAT, false, false, true>(bv, im, fa, partials, nblk);
}

/* tolerance-mode lean launches (FAST: closed-form gradient of the channel's bilinear cell, reciprocals, FMAs; see fused_lk_body) */
template <int AM, int SSM, int MODE, bool CHAINED>
__global__ __launch_bounds__(kBlock, MTFHIP_FAST_WAVES) void k_fused_mc_fast(BatchView bv, ImgView im, FusedArgs fa, double *partials, int nblk) {
	fused_lk_body<AM, SSM, CHAINED, MODE, false, true, false, true>(bv, im, fa, partials, nblk);
}
template <int AM, int SSM>
static void launch_mc_fast(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk, hipStream_t st) {
	const dim3 g = grid2(nblk, bv.B);
	if (fa.mode == 2) MTFHIP_LAUNCH((k_fused_mc_fast<AM, SSM, 2, true>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
	else if (fa.mode == 0 && fa.chained) MTFHIP_LAUNCH((k_fused_mc_fast<AM, SSM, 0, true>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
	else if (fa.mode == 0) MTFHIP_LAUNCH((k_fused_mc_fast<AM, SSM, 0, false>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
	else if (fa.chained) MTFHIP_LAUNCH((k_fused_mc_fast<AM, SSM, 1, true>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
	else MTFHIP_LAUNCH((k_fused_mc_fast<AM, SSM, 1, false>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
}

template <int AM, int SSM, bool CHAINED>
static void launch_mc_mode(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk, hipStream_t st) {
	const dim3 g = grid2(nblk, bv.B);
#define MTFHIP_MC(MD, MT) MTFHIP_LAUNCH((k_fused_mc<AM, SSM, CHAINED, MD, MT>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk)
	if (fa.materialize) { if (fa.mode == 0) MTFHIP_MC(0, true); else if (fa.mode == 1) MTFHIP_MC(1, true); else MTFHIP_MC(2, true); }
	else { if (fa.mode == 0) MTFHIP_MC(0, false); else if (fa.mode == 1) MTFHIP_MC(1, false); else MTFHIP_MC(2, false); }
#undef MTFHIP_MC
}
template <int AM>
static void launch_mc_am(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk, hipStream_t st) {
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY;
	if (fa.fast_math && !fa.materialize) {
		if (hom) launch_mc_fast<AM, MTFHIP_SSM_HOMOGRAPHY>(bv, im, fa, partials, nblk, st);
		else launch_mc_fast<AM, MTFHIP_SSM_AFFINE>(bv, im, fa, partials, nblk, st);
		return;
	}
	if (hom && fa.chained) launch_mc_mode<AM, MTFHIP_SSM_HOMOGRAPHY, true>(bv, im, fa, partials, nblk, st);
	else if (hom) launch_mc_mode<AM, MTFHIP_SSM_HOMOGRAPHY, false>(bv, im, fa, partials, nblk, st);
	else if (fa.chained) launch_mc_mode<AM, MTFHIP_SSM_AFFINE, true>(bv, im, fa, partials, nblk, st);
	else launch_mc_mode<AM, MTFHIP_SSM_AFFINE, false>(bv, im, fa, partials, nblk, st);
}
void launch_fused_mc(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk, hipStream_t st) {
	if (bv.am == MTFHIP_AM_NCC) launch_mc_am<MTFHIP_AM_NCC>(bv, im, fa, partials, nblk, st);
	else launch_mc_am<MTFHIP_AM_SSD>(bv, im, fa, partials, nblk, st);
}

} // namespace mtfhip
