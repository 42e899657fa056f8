/*
 * mtfhip_device.h -- device-side helpers shared by the kernel translation units of libmtfhip.so (gfx950).
 *
 * Compiled with -ffp-contract=off: the per-pixel arithmetic (bilinear sample, finite-difference
 * gradient, warp, steepest-descent row) is written in the reference's operation order so that,
 * without FMA contraction, it rounds exactly like the CPU/Eigen path; only the N-wide reductions
 * (explicit fma accumulation + wavefront shuffles) sum in a different order.
 *
 * Execution model: 256-thread workgroups (4 wave64), each thread walks n_rows pixels strided by
 * the workgroup size so that every wave touches 64 consecutive pixels of a column-major N x S
 * array per instruction (512-byte coalesced segments).  The S x S Hessian is never a GEMM: 36
 * upper-triangle products + 8 gradient terms + r^2 are kept in registers per thread, reduced across
 * the wave with a halving butterfly (each exchange step halves the number of live accumulators, so
 * 48 accumulators cost 51 exchanges instead of 6 x 48), then across the 4 waves through LDS, and
 * written as one partial row per workgroup; a second tiny kernel sums the rows in a fixed order
 * (deterministic, no atomics).
 */
#ifndef MTFHIP_DEVICE_H
#define MTFHIP_DEVICE_H
#include <type_traits>

#include "mtfhip_internal.h"

/* tuning knobs of the fused kernel ("fused kernel tuning") */
#ifndef MTFHIP_FUSED_WAVES
#define MTFHIP_FUSED_WAVES 2   /* minimum waves per SIMD requested from the register allocator */
#endif
#ifndef MTFHIP_FAST_WAVES
#define MTFHIP_FAST_WAVES 2    /* k_fused_fast: 3 or 4 waves per SIMD spill 90-1070 registers (checked with -Rpass-analysis) */
#endif
#ifndef MTFHIP_NT_STORE
#define MTFHIP_NT_STORE 1      /* 1: the materialised It / dIt_dx / Jt are written with non-temporal stores */
#endif
#if MTFHIP_NT_STORE
#define MAT_STORE(ptr, v) __builtin_nontemporal_store((v), (ptr))
#else
#define MAT_STORE(ptr, v) (*(ptr) = (v))
#endif
#ifndef MTFHIP_NT_LOAD
#define MTFHIP_NT_LOAD 0       /* 1: the read-once operands (grid points, I0, J0 columns) are fetched with non-temporal loads */
#endif

namespace mtfhip {

/* ===================================================================== */
/* device helpers                                                         */
/* ===================================================================== */

/* utils::getPixVal<Linear, Constant> -- Utilities/include/mtf/Utilities/imgUtils.h:91-113
 * (overflow test :51-53, overflow_val = 128).  Same operation order as the reference. */
__device__ __forceinline__ double pix_val(const ImgView &im, double x, double y) {
	const double w = (double)(unsigned int)im.w, h = (double)(unsigned int)im.h;
	if ((x < 0) || (x >= w) || (y < 0) || (y >= h)) return 128.0;
	int lx = (int)x, ly = (int)y;
	double dx = x - lx, dy = y - ly;
	int ux = dx == 0 ? lx : lx + 1;
	int uy = dy == 0 ? ly : ly + 1;
	if (ux >= im.w || uy >= im.h) return 128.0;
	const float *r0 = im.data + (size_t)ly * im.stride;
	const float *r1 = im.data + (size_t)uy * im.stride;
	double t00 = r0[lx], t01 = r0[ux], t10 = r1[lx], t11 = r1[ux];
	return t00 * (1 - dx) * (1 - dy) + t01 * dx * (1 - dy) + t10 * (1 - dx) * dy + t11 * dx * dy;
}

/* The centre sample and its four finite-difference neighbours (step 1e-8) almost always fall in
 * one bilinear cell; the cell's four texels are fetched once and every sample that lands in the
 * same cell is evaluated from registers with the reference's expression, so the result is
 * bit-identical to five independent getPixVal calls while issuing 4 loads instead of 20. */
struct Cell {
	int lx, ly, ux, uy;
	double t00, t01, t10, t11;
	bool valid;
};
__device__ __forceinline__ Cell load_cell(const ImgView &im, double x, double y) {
	Cell c;
	c.valid = false;
	c.lx = c.ly = c.ux = c.uy = -1;
	c.t00 = c.t01 = c.t10 = c.t11 = 0;
	const double w = (double)(unsigned int)im.w, h = (double)(unsigned int)im.h;
	if ((x < 0) || (x >= w) || (y < 0) || (y >= h)) return c;
	int lx = (int)x, ly = (int)y;
	double dx = x - lx, dy = y - ly;
	int ux = dx == 0 ? lx : lx + 1;
	int uy = dy == 0 ? ly : ly + 1;
	if (ux >= im.w || uy >= im.h) return c;
	const float *r0 = im.data + (size_t)ly * im.stride;
	const float *r1 = im.data + (size_t)uy * im.stride;
	c.lx = lx; c.ly = ly; c.ux = ux; c.uy = uy;
	c.t00 = r0[lx]; c.t01 = r0[ux]; c.t10 = r1[lx]; c.t11 = r1[ux];
	c.valid = true;
	return c;
}
__device__ __forceinline__ double pix_val_cell(const ImgView &im, const Cell &c, double x, double y) {
	const double w = (double)(unsigned int)im.w, h = (double)(unsigned int)im.h;
	if ((x < 0) || (x >= w) || (y < 0) || (y >= h)) return 128.0;
	int lx = (int)x, ly = (int)y;
	double dx = x - lx, dy = y - ly;
	int ux = dx == 0 ? lx : lx + 1;
	int uy = dy == 0 ? ly : ly + 1;
	if (ux >= im.w || uy >= im.h) return 128.0;
	if (c.valid && lx == c.lx && ly == c.ly && ux == c.ux && uy == c.uy)
		return c.t00 * (1 - dx) * (1 - dy) + c.t01 * dx * (1 - dy) + c.t10 * (1 - dx) * dy + c.t11 * dx * dy;
	const float *r0 = im.data + (size_t)ly * im.stride;
	const float *r1 = im.data + (size_t)uy * im.stride;
	double t00 = r0[lx], t01 = r0[ux], t10 = r1[lx], t11 = r1[ux];
	return t00 * (1 - dx) * (1 - dy) + t01 * dx * (1 - dy) + t10 * (1 - dx) * dy + t11 * dx * dy;
}

/* ===================================================================== */
/* tolerance-mode arithmetic (mtfhip_batch_set_math_mode(MTFHIP_MATH_FAST)) */
/* ===================================================================== */
/* The replay build above reproduces the reference's rounding (unfused mul/add, IEEE divisions, the 1e-8 finite
 * difference of the interpolant); north_star only asks for 1e-5 on H / dp and 1e-9 on candidate scores.  The helpers
 * below compute the same mathematical quantities with explicit FMAs, one reciprocal per homography point and the
 * closed-form derivative of the bilinear interpolant -- the kernels that are FP64-issue bound (lean LK, ICLK, candidate
 * scoring, the grid loop, the MI passes) use them unless the batch is switched to MTFHIP_MATH_REPLAY.
 * Differences against the replay: ~1e-15 relative on samples, and on gradients the reference's own finite-difference
 * noise (128 * 2^-52 / 2e-8 ~ 1.4e-6 absolute), which the closed form does not have. */

/* 1 / d to ~1 ulp: v_rcp_f64 (~2^-26 relative) + two Newton steps; d is a homography denominator (~1), no scaling needed */
__device__ __forceinline__ double rcp_fast(double d) {
	double r = __builtin_amdgcn_rcp(d);
	double e = fma(-d, r, 1.0);
	r = fma(r, e, r);
	e = fma(-d, r, 1.0);
	return fma(r, e, r);
}
/* 1 / sqrt(x) to ~1 ulp: v_rsq_f64 (~2^-26 relative) + two Newton steps (y <- y + y (1 - x y^2) / 2); x > 0, no scaling */
__device__ __forceinline__ double rsq_fast(double x) {
	double y = __builtin_amdgcn_rsq(x);
	double e = fma(-x * y, y, 1.0);
	y = fma(0.5 * y, e, y);
	e = fma(-x * y, y, 1.0);
	return fma(0.5 * y, e, y);
}
/* The step the reference's central difference actually takes.  imgUtils.cc:233-254 samples at fl(w + eps) and fl(w - eps): on a
 * coordinate of a few hundred pixels eps = 1e-8 is rounded to the coordinate's ulp grid (2^-44 in [256, 512): 175 921.86 ulps become
 * 175 922), a SYSTEMATIC relative error of ~8e-7 .. 1e-5 of every gradient that the division by the nominal 2 eps does not undo.
 * Inside one bilinear cell inc - dec = slope * (fl(w + eps) - fl(w - eps)) exactly (the subtraction of the two neighbours is
 * exact), so the closed-form slope times this step is the reference's value without its per-pixel rounding noise.
 * fd_step_sym: the same for a step that is not the nominal eps (updateGradPts' eps * warp column); both neighbours are rounded on
 * the grid of w's binade, so the up-step is the down-step except across a power of two (one grid unit on one pixel). */
__device__ __forceinline__ double fd_step(double w, double eps) { return (w + eps) - (w - eps); }
__device__ __forceinline__ double fd_step_sym(double w, double e) { const double h = (w + e) - w; return h + h; }
/* bilinear interpolant of one cell and its two partial derivatives at fractional position (dx, dy):
 *   v = t00 + dx a + dy b + dx dy c,  dv/dx = a + dy c,  dv/dy = b + dx c   (a = t01 - t00, b = t10 - t00, c = t11 - t10 - a)
 * The reference's central difference with step 1e-8 (imgUtils.cc:233-254) of this function is exactly dv/dx, dv/dy
 * while both neighbours stay inside the cell (the interpolant is linear along each axis). */
__device__ __forceinline__ void bilin_fast(float t00f, float t01f, float t10f, float t11f, double dx, double dy,
	double &v, double &gx, double &gy) {
	const double t00 = t00f, t10 = t10f;
	const double a = (double)t01f - t00, b = t10 - t00, c = ((double)t11f - t10) - a;
	gx = fma(dy, c, a);
	gy = fma(dx, c, b);
	v = fma(dx, gx, fma(dy, b, t00));
}
__device__ __forceinline__ double bilin_val_fast(float t00f, float t01f, float t10f, float t11f, double dx, double dy) {
	const double t00 = t00f, t10 = t10f;
	const double a = (double)t01f - t00, b = t10 - t00, c = ((double)t11f - t10) - a;
	return fma(dx, fma(dy, c, a), fma(dy, b, t00));
}
/* getPixVal<Linear, Constant> (imgUtils.h:91-113) without control flow and with the factored interpolant: the four
 * texel loads are always issued from clamped addresses and the border value 128 is selected afterwards.  The
 * reference's `dx == 0 ? lx : lx + 1` rule only matters for the overflow test (the weight of the upper texel is zero
 * there), which is reproduced: a sample is inside iff (x, y) is and every upper neighbour that carries weight is. */
__device__ __forceinline__ double pix_val_fast(const ImgView &im, double x, double y) {
	const double w = (double)(unsigned int)im.w, h = (double)(unsigned int)im.h;
	const bool in0 = (x >= 0) & (x < w) & (y >= 0) & (y < h);
	const double xs = in0 ? x : 0.0, ys = in0 ? y : 0.0;
	const int lx = (int)xs, ly = (int)ys;
	const double dx = xs - (double)lx, dy = ys - (double)ly;
	const bool in1 = ((lx + 1 < im.w) | (dx == 0)) & ((ly + 1 < im.h) | (dy == 0));
	const int ux = min(lx + 1, im.w - 1), uy = min(ly + 1, im.h - 1);
	const float *r0 = im.data + (unsigned)(ly * im.stride), *r1 = im.data + (unsigned)(uy * im.stride);
	const double v = bilin_val_fast(r0[lx], r0[ux], r1[lx], r1[ux], dx, dy);
	return (in0 & in1) ? v : 128.0;
}
/* steepest-descent row of a homography pixel (Homography.cc:166-186 etc.), contracted */
__device__ __forceinline__ void hom_row_fast(double *r, double Ix, double Iy, double x, double y) {
	const double Ixx = Ix * x, Iyy = Iy * y, Ixy = Ix * y, Iyx = Iy * x;
	r[0] = Ixx; r[1] = Ixy; r[2] = Ix; r[3] = Iyx; r[4] = Iyy; r[5] = Iy;
	r[6] = -fma(x, Ixx, y * Iyx);
	r[7] = -fma(x, Ixy, y * Iyy);
}

struct Warp9 { double m[9]; };
__device__ __forceinline__ Warp9 load_warp(const double *p) {
	Warp9 W;
#pragma unroll
	for (int i = 0; i < 9; ++i) W.m[i] = p[i];
	return W;
}

/* halving butterfly over the 64 lanes of a wave: on entry every lane holds K partial sums in
 * v[0..K); on exit slot j of lane l holds the wave total of index final_index<K,32>(j, l). */
/* The exchanges stay in the VALU (r04): lane l's partner is l ^ MASK as with __shfl_xor -- the same pairs, so the same bits -- but
 * the value arrives by v_permlane32_swap / v_permlane16_swap (MASK 32 / 16: two instructions exchange a register between the halves
 * of the wave / of every row pair, and a halving step needs no keep / send selects: swap(a, b) leaves "mine and the partner's a" in
 * the low half and "the partner's and my b" in the high half) or by DPP moves inside the row of 16 (row_ror:8, row_shl / shr:4 by
 * bank, the two quad permutations) instead of a ds_bpermute round trip through the LDS crossbar (~120 cycles each, 140 of them in
 * the tail of every fused-LK workgroup). */
template <int MASK>
__device__ __forceinline__ double lane_xor_f64(double x) {
	static_assert(MASK == 8 || MASK == 4 || MASK == 2 || MASK == 1, "in-row partners only");
	int lo = __double2loint(x), hi = __double2hiint(x);
	if constexpr (MASK == 8) {          /* row_ror:8 */
		lo = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xF, 0xF, false);
	} else if constexpr (MASK == 4) {   /* banks 0, 2 read lane + 4 (row_shl:4), banks 1, 3 lane - 4 (row_shr:4) */
		const int l1 = __builtin_amdgcn_update_dpp(0, lo, 0x104, 0xF, 0x5, false), h1 = __builtin_amdgcn_update_dpp(0, hi, 0x104, 0xF, 0x5, false);
		lo = __builtin_amdgcn_update_dpp(l1, lo, 0x114, 0xF, 0xA, false); hi = __builtin_amdgcn_update_dpp(h1, hi, 0x114, 0xF, 0xA, false);
	} else if constexpr (MASK == 2) {   /* quad_perm [2, 3, 0, 1] */
		lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false);
	} else {                            /* quad_perm [1, 0, 3, 2] */
		lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false);
	}
	return __hiloint2double(hi, lo);
}
/* MASK 32 / 16: (a, b) -> lanes without the bit: a + partner's a; lanes with it: partner's b + b */
template <int MASK>
__device__ __forceinline__ double swap_pair_add(double a, double b) {
	static_assert(MASK == 32 || MASK == 16, "cross-row partners only");
	if constexpr (MASK == 32) {
		const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
		const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
		return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
	} else {
		const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
		const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
		return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
	}
}
template <int K, int MASK>
__device__ __forceinline__ void wave_halve(double *v, int lane) {
	if constexpr (MASK == 0) {
		return;
	} else if constexpr (K % 2 == 0) {
		constexpr int H = K / 2;
		if constexpr (MASK >= 16) {
#pragma unroll
			for (int j = 0; j < H; ++j) v[j] = swap_pair_add<MASK>(v[j], v[j + H]);
		} else {
			const bool up = (lane & MASK) != 0;
#pragma unroll
			for (int j = 0; j < H; ++j) {
				double keep = up ? v[j + H] : v[j];
				double send = up ? v[j] : v[j + H];
				v[j] = keep + lane_xor_f64<MASK>(send);
			}
		}
		wave_halve<H, (MASK >> 1)>(v, lane);
	} else {
#pragma unroll
		for (int j = 0; j < K; ++j) {
			if constexpr (MASK >= 16) v[j] = swap_pair_add<MASK>(v[j], v[j]);   /* (both halves: own + partner's) */
			else v[j] += lane_xor_f64<MASK>(v[j]);
		}
		wave_halve<K, (MASK >> 1)>(v, lane);
	}
}
template <int K, int MASK>
__device__ __forceinline__ int final_index(int j, int lane) {
	if constexpr (MASK == 0) return j;
	else if constexpr (K % 2 == 0) return final_index<K / 2, (MASK >> 1)>(j, lane) + ((lane & MASK) ? K / 2 : 0);
	else return final_index<K, (MASK >> 1)>(j, lane);
}
template <int K, int MASK>
__device__ __forceinline__ constexpr int final_count() {
	if constexpr (MASK == 0) return K;
	else if constexpr (K % 2 == 0) return final_count<K / 2, (MASK >> 1)>();
	else return final_count<K, (MASK >> 1)>();
}
/* lanes that differ only in bits handled by a full (non-halving) step hold duplicates */
template <int K, int MASK>
__device__ __forceinline__ constexpr int dup_mask() {
	if constexpr (MASK == 0) return 0;
	else if constexpr (K % 2 == 0) return dup_mask<K / 2, (MASK >> 1)>();
	else return MASK | dup_mask<K, (MASK >> 1)>();
}

/* Wave total of a double with DPP moves instead of ds_bpermute: the four in-row steps (xor 1, xor 2, half-row mirror, row mirror)
 * and the two row broadcasts stay in the VALU (a DPP move is ~8 cycles; a ds_bpermute round trip through the LDS crossbar
 * ~120), which matters where a loop of dependent reductions is the critical path (the one-launch grid kernel: one wave per SIMD,
 * nothing to overlap).  The result is uniform (read from lane 63). */
__device__ __forceinline__ double wave_sum_dpp(double v) {
	auto step = [](double x, int ctrl, int row_mask) {
		const int lo = __double2loint(x), hi = __double2hiint(x);
		int tl, th;
		switch (ctrl) {   /* the control word must be an immediate */
		case 0: tl = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, false); th = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, false); break;     /* quad_perm [1,0,3,2] */
		case 1: tl = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, false); th = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, false); break;     /* quad_perm [2,3,0,1] */
		case 2: tl = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, false); th = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, false); break;   /* row_half_mirror */
		case 3: tl = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, false); th = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, false); break;   /* row_mirror */
		case 4: tl = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false); th = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false); break;   /* row_bcast:15 -> rows 1, 3 */
		default: tl = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, false); th = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, false); break;  /* row_bcast:31 -> rows 2, 3 */
		}
		(void)row_mask;
		return x + __hiloint2double(th, tl);
	};
	v = step(v, 0, 0xF); v = step(v, 1, 0xF); v = step(v, 2, 0xF); v = step(v, 3, 0xF); v = step(v, 4, 0xA); v = step(v, 5, 0xC);
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
	return __hiloint2double(hi, lo);
}

/* Data handed from one workgroup to another INSIDE a launch (k_track_persist): agent-scope relaxed atomics are sc1 accesses --
 * stores write through to the memory side, loads never hit a stale line of this XCD's L2 -- so the hand-over needs no L2
 * write-back / invalidate (`buffer_wbl2` / `buffer_inv`, microseconds each on eight 4 MB L2s): the producer waits for its stores
 * to be acknowledged (vmcnt) and then raises a flag the same way. */
template <typename T> __device__ __forceinline__ void st_coh(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T ld_coh(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
/* lane `src` (a compile-time constant or a wave-uniform value) of a double, through the scalar unit */
__device__ __forceinline__ double readlane_f64(double v, int src) {
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wait_stores_acked() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

/* reduce K per-thread accumulators over the workgroup and write them to dst[0..K) */
template <int K, bool COH = false>
__device__ __forceinline__ void block_reduce_store(double *v, double *dst, double *lds /* [4][K] */) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	wave_halve<K, 32>(v, lane);
	constexpr int CNT = final_count<K, 32>();
	constexpr int DUP = dup_mask<K, 32>();
	if ((lane & DUP) == 0) {
#pragma unroll
		for (int j = 0; j < CNT; ++j) lds[wave * K + final_index<K, 32>(j, lane)] = v[j];
	}
	__syncthreads();
	if (threadIdx.x < K) {
		double s = lds[threadIdx.x];
#pragma unroll
		for (int wv = 1; wv < kBlock / 64; ++wv) s += lds[wv * K + threadIdx.x];
		if constexpr (COH) st_coh(dst + threadIdx.x, s); else dst[threadIdx.x] = s;
	}
}

/* grid_patch_corners_hd (mtfhip_internal.h) with the cell's four grid points on four lanes: a point is four IEEE divisions (two in
 * lin_spaced_hd, two projective), ~1 us as a dependent chain of sixteen on a wave that has its SIMD to itself; lane q & 3 evaluates
 * point q -- the same expressions, so the same bits -- and v_readlane hands the eight coordinates to everybody (every wave does the same) */
__device__ __forceinline__ void grid_patch_corners_lanes(const GridLayoutHD &gl, const double *Wr, int t, double *q8) {
	const int extra = (gl.dyn_patch_size || gl.patch_centroid_inside) ? 1 : 0;
	const int gresx = gl.grid_size_x + extra, gresy = gl.grid_size_y + extra, sub_x = gl.grid_size_x + 1;
	const int prow = t / gl.grid_size_x, pcol = t % gl.grid_size_x, lq = threadIdx.x & 3;
	const int pid = extra ? (prow + (lq >> 1)) * sub_x + pcol + ((lq == 1 || lq == 2) ? 1 : 0) : t;   /* TL, TR, BR, BL of the cell | the patch's own point */
	double gx, gy;
	grid_pt_hd(Wr, gresx, gresy, pid, &gx, &gy);
#pragma unroll
	for (int q = 0; q < 8; ++q) q8[q] = 0.0;
	if (extra) {
#pragma unroll
		for (int q = 0; q < 4; ++q) { q8[2 * q] = readlane_f64(gx, q); q8[2 * q + 1] = readlane_f64(gy, q); }
	}
	if (!gl.dyn_patch_size) {
		double cx = readlane_f64(gx, 0), cy = readlane_f64(gy, 0);
		if (gl.patch_centroid_inside) {
			cx = (q8[0] + q8[2] + q8[4] + q8[6]) / 4.0;
			cy = (q8[1] + q8[3] + q8[5] + q8[7]) / 4.0;
		}
		const double half_x = gl.patch_size_x / 2.0, half_y = gl.patch_size_y / 2.0;
		const double min_x = cx - half_x, min_y = cy - half_y;
		const double max_x = min_x + gl.patch_size_x, max_y = min_y + gl.patch_size_y;
		q8[0] = q8[6] = min_x; q8[2] = q8[4] = max_x;
		q8[1] = q8[3] = min_y; q8[5] = q8[7] = max_y;
	}
}

/* steepest-descent row of one pixel: S values */
template <int SSM>
struct Row { double v[SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6]; };

/* Homography row writer shared by cmptInitPixJacobian / cmptPixJacobian / cmptWarpedPixJacobian /
 * cmptApproxPixJacobian (SSM/src/Homography.cc:166-186, 213-224, 270-289, 330-341) */
__device__ __forceinline__ void hom_row(double *r, double Ix, double Iy, double x, double y, double px, double py) {
	double Ixx = Ix * x, Iyy = Iy * y, Ixy = Ix * y, Iyx = Iy * x;
	r[0] = Ixx; r[1] = Ixy; r[2] = Ix; r[3] = Iyx; r[4] = Iyy; r[5] = Iy;
	r[6] = -px * Ixx - py * Iyx;
	r[7] = -px * Ixy - py * Iyy;
}

/* fixed-order sum of one column of the block rows, eight loads in flight */
__device__ __forceinline__ double column_sum(const double *col, int nblk, int row_len) {
	double s[8];
#pragma unroll
	for (int u = 0; u < 8; ++u) s[u] = 0.0;
	int b = 0;
	for (; b + 7 < nblk; b += 8) {
#pragma unroll
		for (int u = 0; u < 8; ++u) s[u] += col[(size_t)(b + u) * row_len];
	}
	for (; b < nblk; ++b) s[0] += col[(size_t)b * row_len];
	return ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

/* ---- multi-channel (mc::) sampling: image H x W x C interleaved; one thread per (pixel, channel) row.
 * mc::PixVal<Linear, Constant>::get (imgUtils.h:505-551) forms the four bilinear weights first and applies them per
 * channel -- not the single-channel operation order -- and so does this. ---- */
__device__ __forceinline__ double pix_val_mc(const ImgView &im, double x, double y, int ch) {
	const double w = (double)(unsigned int)im.w, h = (double)(unsigned int)im.h;
	if ((x < 0) || (x >= w) || (y < 0) || (y >= h)) return 128.0;
	int lx = (int)x, ly = (int)y;
	double dx = x - lx, dy = y - ly;
	int ux = dx == 0 ? lx : lx + 1;
	int uy = dy == 0 ? ly : ly + 1;
	if (ux >= im.w || uy >= im.h) return 128.0;
	const double ly_lx = (1 - dx) * (1 - dy), ly_ux = dx * (1 - dy), uy_lx = (1 - dx) * dy, uy_ux = dx * dy;
	const float *r0 = im.data + (size_t)ly * im.stride, *r1 = im.data + (size_t)uy * im.stride;
	const int C = im.channels;
	const double t00 = r0[lx * C + ch], t01 = r0[ux * C + ch], t10 = r1[lx * C + ch], t11 = r1[ux * C + ch];
	return t00 * ly_lx + t01 * ly_ux + t10 * uy_lx + t11 * uy_ux;
}

/* launch grid: x = workgroups per target, y = targets */
static inline dim3 grid2(int nblk, int B) { return dim3((unsigned)nblk, (unsigned)B, 1); }

/* Uniform base + 32-bit byte offset: the form the `global_load/store v, v_off, s[base]` encodings take directly.
 * With `ptr[i]` the compiler cannot prove that i * sizeof(T) stays below 2^32 and builds a 64-bit address per
 * access (v_lshl_add_u64 / v_mad_u64), ~50 extra VALU instructions per row in a loop that is VALU-issue bound.
 * mtfhip_batch_create bounds the per-target arrays to < 4 GiB so the offsets cannot wrap. */
template <typename T>
__device__ __forceinline__ T ld_off(const void *base, unsigned byte_off) {
	return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void st_off(void *base, unsigned byte_off, T v) {
	MAT_STORE(reinterpret_cast<T *>(static_cast<char *>(base) + byte_off), v);
}

} // namespace mtfhip
#endif
