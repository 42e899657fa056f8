/*
 * mtfhip_mi_pass2_unit.h -- body of the kernels_mi_pass2_*.hip translation units: the five (Hessian kind, Jacobian row) forms of pass 2 of the
 * MI recompute iteration (k_mi_pass_grad_hess, mtfhip_mi_fused_device.h) for ONE state-space model and channel layout, named by
 * MTFHIP_P2_SSM / MTFHIP_P2_MC / MTFHIP_P2_NAME.
 */
#include "mtfhip_mi_fused_device.h"

#ifndef MTFHIP_P2_NB
#define MTFHIP_P2_NB 8   /* the bin count the class tables are laid out for: 8, or 10 (the polynomial forms only: constant and self Hessian) */
#endif

namespace mtfhip {

void MTFHIP_P2_NAME(const BatchView &bv, const ImgView &im, const MiPassArgs &pa, int hk, int hrow, double *partials, int nblk, hipStream_t st) {
	const dim3 g = grid2(nblk, bv.B);
#define MTFHIP_MI_P2(HK_, HR_) do { \
		if (pa.nonchained) MTFHIP_LAUNCH((k_mi_pass_grad_hess<MTFHIP_P2_SSM, HK_, HR_, MTFHIP_P2_MC, true, MTFHIP_P2_NB>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk); \
		else MTFHIP_LAUNCH((k_mi_pass_grad_hess<MTFHIP_P2_SSM, HK_, HR_, MTFHIP_P2_MC, false, MTFHIP_P2_NB>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk); } while (0)
	if (hk == 0) MTFHIP_MI_P2(0, 0);
	else if (hk == 1) MTFHIP_MI_P2(1, 0);
#if MTFHIP_P2_NB == 8
	else if (hk == 2 && hrow == 2) MTFHIP_MI_P2(2, 2);
	else if (hk == 2) MTFHIP_MI_P2(2, 0);
	else MTFHIP_MI_P2(3, 1);
#else
	else ::mtfhip::note_launch_error(hipErrorInvalidValue, __FILE__, __LINE__);   /* (other than 8 bins: the constant and self Hessian forms only -- mi_fast_ok does not send the dense forms here) */
#endif
#undef MTFHIP_MI_P2
}

} // namespace mtfhip
