/*
 * kernels_persist.hip -- every iteration of a target's Lucas-Kanade loop in ONE launch (single large targets: configs 1 and 2 as a
 * tracker runs them).  One of the translation units of libmtfhip.so.
 *
 * nt::ESM / FCLK / ICLK::update (SM/src/NT/ESM.cc:170-296, NT/FCLK.cc:187-342, NT/ICLK.cc:160-298) is a chain of dependent
 * iterations; with one launch for the pixel pass and one for the solve, a 200 x 200 target spends more time between kernels than in
 * them (10 us of kernel in a 16-17 us iteration).  Here the workgroups of a target stay resident for the whole loop:
 *
 *   every workgroup:  fused_lk_body at the current warp -> its partial row (write-through stores) -> arrive (agent-scope counter)
 *   last arriver:     finish_track_body -- fixed-order sum of the partial rows, g / H of the search method, Levenberg-Marquardt
 *                     accept / undo, pivoted solve, compositional update, corner test -- then publishes the generation number
 *   everyone else:    waits for the generation number, re-reads warp and live flag, goes on
 *
 * The arithmetic, the partial rows and their summation order are those of the two-launch loop (launch_fused_ssd +
 * launch_finish_track): results are bit-identical to it.
 *
 * Co-residency: a waiting workgroup occupies its CU, so the launcher only takes this route when the whole grid fits the device at
 * one workgroup per CU.  The wait is bounded all the same (another process may hold CUs): a workgroup that gives up leaves the
 * target `active` with its iteration count short of the limit, and the host finishes the loop with the two-launch form.
 */
#include "mtfhip_finish_device.h"
#include "mtfhip_fused_device.h"

namespace mtfhip {

template <int AM, int SSM, bool CHAINED, int MODE, bool FAST>
__global__ __launch_bounds__(kBlock, 1) void k_track_persist(BatchView bv, ImgView im, FusedArgs fa,
	mtfhip_sm_desc sm, TrackState ts, double *partials, int nblk, PersistState ps, int max_passes) {
	__shared__ int s_last, s_abort;
	const int t = blockIdx.y, tid = threadIdx.x;
	if (tid == 0) s_abort = 0;
	for (int pass = 0; pass < max_passes; ++pass) {
		/* (uniform: written by the last arriver of the previous pass, ordered by the generation number) */
		if (ld_coh(ts.active + t) == 0) break;
		fused_lk_body<AM, SSM, CHAINED, MODE, false, FAST, true>(bv, im, fa, partials, nblk);
		/* the partial row was stored (write-through) by the first 48 (SSD) / 72 (NCC) threads: once those stores are acknowledged the
		 * workgroup counts itself in.  No cache maintenance anywhere in the hand-over (see st_coh in mtfhip_device.h). */
		wait_stores_acked();
		__syncthreads();
		if (tid == 0) s_last = __hip_atomic_fetch_add(ps.arrive + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1;
		__syncthreads();
		const unsigned want = ps.gen_base + (unsigned)pass + 1u;
		if (s_last) {
			if (tid == 0) st_coh(ps.arrive + t, 0);
			if (ts.fast_finish) finish_track_fast_body<true>(bv, sm, ts, partials, nblk, t);
			else finish_track_body<true>(bv, sm, ts, partials, nblk, t);
			wait_stores_acked();
			__syncthreads();
			if (tid == 0) st_coh(ps.gen + t, want);
		} else if (tid == 0) {
			const unsigned long long t0 = wall_clock64();   /* 100 MHz */
			unsigned spins = 0;
			while ((int)(ld_coh(ps.gen + t) - want) < 0) {
				__builtin_amdgcn_s_sleep(4);
				if ((++spins & 63u) == 0 && wall_clock64() - t0 > ps.timeout_ticks) { s_abort = 1; break; }
			}
		}
		asm volatile("" ::: "memory");
		__syncthreads();
		if (s_abort) break;
	}
}

template <int AM, int SSM, bool FAST>
static void launch_persist_mode(const BatchView &bv, const ImgView &im, const FusedArgs &fa, const mtfhip_sm_desc &sm, const TrackState &ts,
	double *partials, int nblk, const PersistState &ps, int max_passes, hipStream_t st) {
	const dim3 g = grid2(nblk, bv.B);
#define MTFHIP_PERSIST(CH, MD) MTFHIP_LAUNCH((k_track_persist<AM, SSM, CH, MD, FAST>), g, dim3(kBlock), 0, st, bv, im, fa, sm, ts, partials, nblk, ps, max_passes)
	if (fa.chained || (FAST && fa.mode == 2)) {   /* (ICLK takes no gradient: the tolerance-mode body is instantiated once for it) */
		if (fa.mode == 0) MTFHIP_PERSIST(true, 0); else if (fa.mode == 1) MTFHIP_PERSIST(true, 1); else MTFHIP_PERSIST(true, 2);
	} else {
		if (fa.mode == 0) MTFHIP_PERSIST(false, 0); else if (fa.mode == 1) MTFHIP_PERSIST(false, 1); else MTFHIP_PERSIST(false, 2);
	}
#undef MTFHIP_PERSIST
}
template <int AM, bool FAST>
static void launch_persist_ssm(const BatchView &bv, const ImgView &im, const FusedArgs &fa, const mtfhip_sm_desc &sm, const TrackState &ts,
	double *partials, int nblk, const PersistState &ps, int max_passes, hipStream_t st) {
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) launch_persist_mode<AM, MTFHIP_SSM_HOMOGRAPHY, FAST>(bv, im, fa, sm, ts, partials, nblk, ps, max_passes, st);
	else launch_persist_mode<AM, MTFHIP_SSM_AFFINE, FAST>(bv, im, fa, sm, ts, partials, nblk, ps, max_passes, st);
}
/* fa.materialize must be 0 (the interface-visible arrays of an iteration are the two-launch loop's business) */
void launch_track_persist(const BatchView &bv, const ImgView &im, const FusedArgs &fa, const mtfhip_sm_desc &sm, const TrackState &ts,
	double *partials, int nblk, const PersistState &ps, int max_passes, hipStream_t st) {
	const bool ncc = bv.am == MTFHIP_AM_NCC;
	if (fa.fast_math) {
		if (ncc) launch_persist_ssm<MTFHIP_AM_NCC, true>(bv, im, fa, sm, ts, partials, nblk, ps, max_passes, st);
		else launch_persist_ssm<MTFHIP_AM_SSD, true>(bv, im, fa, sm, ts, partials, nblk, ps, max_passes, st);
	} else {
		if (ncc) launch_persist_ssm<MTFHIP_AM_NCC, false>(bv, im, fa, sm, ts, partials, nblk, ps, max_passes, st);
		else launch_persist_ssm<MTFHIP_AM_SSD, false>(bv, im, fa, sm, ts, partials, nblk, ps, max_passes, st);
	}
}

} // namespace mtfhip
