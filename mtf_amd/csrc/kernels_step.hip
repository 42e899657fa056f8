/*
 * kernels_step.hip -- one launch per pass of a small batch's Lucas-Kanade loop (r05): the pixel pass of fused_lk_body with the solve +
 * update of finish_track_*_body folded in behind a last-arriver counter.  One of the translation units of libmtfhip.so.
 *
 * A single 200 x 200 target (configs 1 / 2 as an MTF tracker runs them) spent an iteration in two dependent launches -- the pixel
 * pass, then a one-workgroup finish -- each with its own launch-to-first-load latency (6.6 + 4.9 us of a 12.3 us iteration, r04).
 * Here every workgroup stores its partial row as write-through stores, waits for their acknowledgement and counts itself in on an
 * agent-scope counter (the hand-over of kernels_persist.hip and of the particle filter's scan); the LAST workgroup to arrive sums the
 * rows in the fixed order of the stand-alone finish, solves, updates the warp and the iteration state -- and the kernel ends.  Nobody
 * waits inside the kernel (the persistent loop's waiting workgroups cost what a launch boundary costs, r03): the next pass is the
 * next launch of the same kernel, already enqueued, ordered behind this one by the stream.
 *
 * MEASURED (r05): no gain -- 200 x 200 full 12.22 -> 12.54 us per iteration, lean 10.92 -> 10.79, 50 x 50 10.55 -> 10.44; the hand-over
 * inside the kernel costs what the launch boundary cost.  The device-side loop therefore keeps its two launches by default and takes
 * this form only with MTFHIP_STEP=1 (api_fused.hip, track_core).
 *
 * Same decomposition, same partial rows, same summation order and the same finish bodies as launch_fused_ssd + launch_finish_track:
 * bit-identical results (tests/test_gpu_trackers.py::test_one_launch_per_pass_equals_two_launch_loop).
 * References: SM/src/NT/ESM.cc:170-292, NT/FCLK.cc:187-342, NT/ICLK.cc:160-298 (one loop pass).
 */
#include "mtfhip_finish_device.h"
#include "mtfhip_fused_device.h"

namespace mtfhip {

template <int AM, int SSM, bool CHAINED, int MODE, bool MAT, bool FAST>
__global__ __launch_bounds__(kBlock) void k_track_step(BatchView bv, ImgView im, FusedArgs fa, mtfhip_sm_desc sm, TrackState ts,
	double *partials, int nblk, int *arrive) {
	__shared__ int s_last;
	const int t = blockIdx.y, tid = threadIdx.x;
	/* (the flag was written by the previous launch's finish: a plain load; fa.active points at the same words, so the body agrees) */
	const int live = ts.active[t];
	fused_lk_body<AM, SSM, CHAINED, MODE, MAT, FAST, false, false, true>(bv, im, fa, partials, nblk);
	if (!live) return;   /* (uniform over the target's workgroups: nobody counts itself in) */
	wait_stores_acked();   /* the row's write-through stores (the first 48 / 72 threads) are performed */
	__syncthreads();
	if (tid == 0) s_last = __hip_atomic_fetch_add(arrive + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
	__syncthreads();
	if (!s_last) return;
	if (tid == 0) st_coh(arrive + t, 0);
	if (ts.fast_finish) finish_track_fast_body<true>(bv, sm, ts, partials, nblk, t);
	else finish_track_body<true>(bv, sm, ts, partials, nblk, t);
}

template <int AM, int SSM, bool MAT, bool FAST>
static void launch_step_mode(const BatchView &bv, const ImgView &im, const FusedArgs &fa, const mtfhip_sm_desc &sm, const TrackState &ts,
	double *partials, int nblk, int *arrive, hipStream_t st) {
	const dim3 g = grid2(nblk, bv.B);
#define MTFHIP_STEP(CH, MD) MTFHIP_LAUNCH((k_track_step<AM, SSM, CH, MD, MAT, FAST>), g, dim3(kBlock), 0, st, bv, im, fa, sm, ts, partials, nblk, arrive)
	if (fa.chained || (FAST && fa.mode == 2)) {   /* (ICLK takes no gradient: the tolerance-mode body is instantiated once for it) */
		if (fa.mode == 0) MTFHIP_STEP(true, 0); else if (fa.mode == 1) MTFHIP_STEP(true, 1); else MTFHIP_STEP(true, 2);
	} else {
		if (fa.mode == 0) MTFHIP_STEP(false, 0); else if (fa.mode == 1) MTFHIP_STEP(false, 1); else MTFHIP_STEP(false, 2);
	}
#undef MTFHIP_STEP
}
template <int AM, bool MAT, bool FAST>
static void launch_step_ssm(const BatchView &bv, const ImgView &im, const FusedArgs &fa, const mtfhip_sm_desc &sm, const TrackState &ts,
	double *partials, int nblk, int *arrive, hipStream_t st) {
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) launch_step_mode<AM, MTFHIP_SSM_HOMOGRAPHY, MAT, FAST>(bv, im, fa, sm, ts, partials, nblk, arrive, st);
	else launch_step_mode<AM, MTFHIP_SSM_AFFINE, MAT, FAST>(bv, im, fa, sm, ts, partials, nblk, arrive, st);
}
/* the two forms the device-side loop launches by default: tolerance mode without materialised arrays (the lean loop), replay
 * arithmetic with them (the full loop); track_step_available() says which (fa) this unit serves */
bool track_step_available(const BatchView &bv, const FusedArgs &fa) {
	if (bv.C != 1) return false;
	return (fa.fast_math && !fa.materialize) || (!fa.fast_math && fa.materialize);
}
void launch_track_step(const BatchView &bv, const ImgView &im, const FusedArgs &fa, const mtfhip_sm_desc &sm, const TrackState &ts,
	double *partials, int nblk, int *arrive, hipStream_t st) {
	const bool ncc = bv.am == MTFHIP_AM_NCC;
	if (fa.fast_math && !fa.materialize) {
		if (ncc) launch_step_ssm<MTFHIP_AM_NCC, false, true>(bv, im, fa, sm, ts, partials, nblk, arrive, st);
		else launch_step_ssm<MTFHIP_AM_SSD, false, true>(bv, im, fa, sm, ts, partials, nblk, arrive, st);
	} else {
		if (ncc) launch_step_ssm<MTFHIP_AM_NCC, true, false>(bv, im, fa, sm, ts, partials, nblk, arrive, st);
		else launch_step_ssm<MTFHIP_AM_SSD, true, false>(bv, im, fa, sm, ts, partials, nblk, arrive, st);
	}
}

} // namespace mtfhip
