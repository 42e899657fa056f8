/*
 * mtfhip_grid_device.h -- device helpers shared by the grid kernels (k_iclk_track, kernels_batch.hip; k_template_init, kernels_init.hip;
 * k_grid_fb, kernels_grid_fb.hip): the workgroup reductions of the one-launch ICLK loop, the reductions and the in-LDS inversion of the
 * fused template initialisation, the hand-over of a patch's results to the host.
 */
#pragma once
#include "mtfhip_device.h"

namespace mtfhip {

/* ===================================================================== */
/* one-launch inverse-compositional tracker for small patches (GridTracker) */
/* ===================================================================== */
/* sum of K per-thread values over the workgroup, result broadcast to every thread */
template <int K>
__device__ __forceinline__ void block_allsum(double *v, double *lds /* [4][K] */) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < K; ++k)
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1) v[k] += __shfl_xor(v[k], m);
	__syncthreads();   /* previous round's readers are done with lds */
	if (lane == 0) {
#pragma unroll
		for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < K; ++k) v[k] = (lds[k] + lds[K + k]) + (lds[2 * K + k] + lds[3 * K + k]);
}

/* the same with DPP wave sums (wave_sum_dpp): for loops whose critical path is this reduction */
template <int K>
__device__ __forceinline__ void block_allsum_dpp(double *v, double *lds /* [4][K] */) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < K; ++k) v[k] = wave_sum_dpp(v[k]);
	__syncthreads();   /* previous round's readers are done with lds */
	if (lane == 0) {
#pragma unroll
		for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < K; ++k) v[k] = (lds[k] + lds[K + k]) + (lds[2 * K + k] + lds[3 * K + k]);
}

/* Twelve sums at once, by halving: the two cross-row levels of the wave are exchanges of register halves between lanes
 * (v_permlane32_swap / v_permlane16_swap, gfx950) -- after the first every lane carries six of the twelve, after the second three --
 * and only those three go through the four in-row DPP steps.  63 VALU instructions per wave instead of the 12 x 18 of one wave_sum_dpp
 * per value, on a loop whose critical path is this reduction (k_iclk_track: one wave per SIMD, an FP64 instruction every ~7 cycles).
 * Row r of a wave ends with the wave totals of indices (r >= 2 ? 6 : 0) + (r & 1 ? 3 : 0) + {0, 1, 2} in every lane. */
__device__ __forceinline__ double swap_add32(double a, double b) {   /* lanes 0..31: a[l] + a[l + 32]; lanes 32..63: b[l - 32] + b[l] */
	const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
	const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
	return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double swap_add16(double a, double b) {   /* even rows: a[row] + a[row + 1]; odd rows: b[row - 1] + b[row] */
	const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
	const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
	return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double row_sum_dpp(double x) {   /* the sum over the lane's row of 16, in every lane of the row */
#define MTFHIP_ROW_STEP(CTRL) { \
		const int tl = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false), th = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false); \
		x += __hiloint2double(th, tl); }
	MTFHIP_ROW_STEP(0xB1) MTFHIP_ROW_STEP(0x4E) MTFHIP_ROW_STEP(0x141) MTFHIP_ROW_STEP(0x140)   /* quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror */
#undef MTFHIP_ROW_STEP
	return x;
}
/* v[0..12) summed over the workgroup, every thread gets every total.  lds: [4][12], a buffer the caller ALTERNATES between
 * consecutive calls (the readers of one round are then separated from the next writers of the same buffer by the round in between:
 * one barrier per call).  The four waves' totals are combined by lanes 0..11 (one index each) and handed to everybody through the
 * scalar unit: 4 LDS reads + 3 additions + 24 v_readlane instead of 48 broadcast reads + 36 additions per thread. */
__device__ __forceinline__ void block_allsum_h12(double *v, double *lds) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	double h6[6], h3[3];
#pragma unroll
	for (int j = 0; j < 6; ++j) h6[j] = swap_add32(v[j], v[j + 6]);
#pragma unroll
	for (int j = 0; j < 3; ++j) h3[j] = row_sum_dpp(swap_add16(h6[j], h6[j + 3]));
	if ((lane & 15) == 0) {
		const int base = wave * 12 + ((lane & 32) ? 6 : 0) + ((lane & 16) ? 3 : 0);
#pragma unroll
		for (int j = 0; j < 3; ++j) lds[base + j] = h3[j];
	}
	__syncthreads();
	const int k = lane < 12 ? lane : 0;
	const double mine = (lds[k] + lds[12 + k]) + (lds[24 + k] + lds[36 + k]);
#pragma unroll
	for (int q = 0; q < 12; ++q) v[q] = readlane_f64(mine, q);
}

/* the tail of k_iclk_track: target t's final warp / state / corners / iteration count go to the host mirror of the slab, and the
 * last workgroup of the launch releases the host (one kernel launch and its gap less per frame than k_publish_host).
 * Called by wave 0 only, lane q holding entry q: the stores are system-scope (write-through to the pinned page), ONE agent-scope
 * release per workgroup orders them before the counter -- a system-scope fence in every wave of every workgroup walks the L2
 * for dirty lines 1024 times and cost 25 us of a 47 us launch -- and (fenced form) only the last arriver pays the system-scope release
 * before it raises the flag. */
__device__ __forceinline__ void publish_target(const HostPublish &pub, int t, double wq, double sq, double cq, int n_it) {
	const int lane = threadIdx.x;
	double *p = reinterpret_cast<double *>(pub.host);
	const size_t Bt = (size_t)pub.B;
	if (lane < 9) __hip_atomic_store(p + 9 * (size_t)t + lane, wq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	if (lane < 8) {
		__hip_atomic_store(p + 9 * Bt + 8 * (size_t)t + lane, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__hip_atomic_store(p + 17 * Bt + 8 * (size_t)t + lane, cq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (lane == 0) __hip_atomic_store(reinterpret_cast<int *>(pub.host + pub.dbl_bytes) + Bt + t, n_it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	/* Default: the stores above are write-through (system scope): when the wave's vmcnt has drained they are performed, which is all
	 * the counter has to order -- an agent-scope release here and an acq_rel on the counter wrote this XCD's L2 back twice and
	 * invalidated it once per workgroup (the launch has just laid 6 MB of template grids into the L2s): ~3 of the 7 us between the last
	 * iteration and the end of the workgroup, r04 phase trace.  The counter itself is an agent-scope atomic: performed at the memory
	 * side.  pub.fenced (MTFHIP_PUBLISH_FENCE=1, publish_fenced()): the release / acq_rel / system-release form the memory model asks for. */
	if (pub.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   /* (the whole wave's stores: s_waitcnt vmcnt(0) is per wave) */
	else wait_stores_acked();
	if (lane == 0) {
		const int done = pub.fenced ? __hip_atomic_fetch_add(pub.count, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
		                            : __hip_atomic_fetch_add(pub.count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (done == (int)gridDim.x - 1) {
			__hip_atomic_store(pub.count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			/* Every workgroup's results left as system-scope (write-through) stores that were acknowledged before it counted itself in,
			 * and this one has read the counter they all bumped (a RELAXED read-modify-write at the memory side, not an acquire): the
			 * stores are performed, and the flag -- one more posted write of the same device -- cannot pass them on the link.  A
			 * system-scope RELEASE here (r03: __threadfence_system + a release store) writes back the whole L2 twice -- since r04 that
			 * includes the 6 MB of template grids the same launch laid out -- for nothing the host reads: 2.5 us of a 50 us frame. */
			if (pub.fenced) {
				__threadfence_system();
				__hip_atomic_store(pub.flag, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
			} else __hip_atomic_store(pub.flag, pub.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

/* ===================================================================== */
/* fused template initialisation (k_template_init)                        */
/* ===================================================================== */
template <int K>
__device__ __forceinline__ void init_allsum(double *v, double *lds /* [4][K] */) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < K; ++k) v[k] = wave_sum_dpp(v[k]);
	__syncthreads();   /* previous round's readers are done with lds */
	if (lane == 0) {
#pragma unroll
		for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < K; ++k) v[k] = (lds[k] + lds[K + k]) + (lds[2 * K + k] + lds[3 * K + k]);
}

/* H (S x S column-major, packed) -> its inverse, or zeros when a pivot vanishes (a flat template: no update, as invert_definite on the
 * host).  One wave: lane (i, j) = (lane >> 3, lane & 7) owns A[i][j] and A[i][8 + j] of the augmented matrix in LDS; diagonal
 * equilibration and partial pivoting as the host routine.  a: [8][17] doubles. */
__device__ __forceinline__ void invert_definite_wave(int S, const double *Hs /* LDS, packed S x S */, double *a, double *out /* global, packed */) {
	const int lane = threadIdx.x & 63, i = lane >> 3, j = lane & 7;
	constexpr int LD = 17;
	const bool in = i < S && j < S;
	const double di = i < S ? fabs(Hs[i * S + i]) : 1.0, dj = j < S ? fabs(Hs[j * S + j]) : 1.0;
	const double sci = di > 0 ? 1.0 / sqrt(di) : 1.0, scj = dj > 0 ? 1.0 / sqrt(dj) : 1.0;
	a[i * LD + j] = in ? Hs[j * S + i] * sci * scj : (i == j ? 1.0 : 0.0);
	a[i * LD + 8 + j] = i == j ? 1.0 : 0.0;
	__builtin_amdgcn_wave_barrier();
	bool singular = false;
	for (int k = 0; k < S; ++k) {
		/* pivot row: the largest |A[r][k]|, r >= k (every lane walks the <= 8 candidates: same result everywhere) */
		int piv = k; double best = fabs(a[k * LD + k]);
		for (int r = k + 1; r < S; ++r) { const double v = fabs(a[r * LD + k]); if (v > best) { best = v; piv = r; } }
		if (best == 0) { singular = true; break; }
		__builtin_amdgcn_wave_barrier();
		if (piv != k && i == 0) {   /* lanes 0..7 swap both halves of the two rows */
			const double t0 = a[piv * LD + j], t1 = a[piv * LD + 8 + j];
			a[piv * LD + j] = a[k * LD + j]; a[piv * LD + 8 + j] = a[k * LD + 8 + j];
			a[k * LD + j] = t0; a[k * LD + 8 + j] = t1;
		}
		__builtin_amdgcn_wave_barrier();
		const double p = a[k * LD + k];
		__builtin_amdgcn_wave_barrier();
		if (i == 0) { a[k * LD + j] /= p; a[k * LD + 8 + j] /= p; }
		__builtin_amdgcn_wave_barrier();
		const double f = a[i * LD + k], r0 = a[k * LD + j], r1 = a[k * LD + 8 + j];
		__builtin_amdgcn_wave_barrier();
		if (i != k && f != 0) { a[i * LD + j] -= f * r0; a[i * LD + 8 + j] -= f * r1; }
		__builtin_amdgcn_wave_barrier();
	}
	if (in) out[j * S + i] = singular ? 0.0 : a[i * LD + 8 + j] * sci * scj;
}

} // namespace mtfhip
