/*
 * mtfhip_fused_device.h -- the body of the fused Lucas-Kanade iteration (SSD and NCC), shared by the translation units that
 * instantiate it: kernels_fused.hip (one launch per iteration) and kernels_persist.hip (all iterations of a target in one
 * launch).  No relocatable device code: the body is inlined into each kernel.
 */
#ifndef MTFHIP_FUSED_DEVICE_H
#define MTFHIP_FUSED_DEVICE_H
#include "mtfhip_device.h"

namespace mtfhip {

/* ===================================================================== */
/* the fused Lucas-Kanade iteration, SSD                                  */
/* ===================================================================== */
/*
 * One pass per pixel, everything in registers:
 *   warp the grid point (A11) -> bilinear sample It (A1/A2) -> residual (A7) ->
 *   finite-difference gradient, chained (A3) or of the warped image (A4) ->
 *   steepest-descent row (A5 / A6) -> accumulate J^T r (A8) and J^T J (A9)
 * MODE 0 FCLK: g += -r * Jt            H += Jt (x) Jt
 * MODE 1 ESM : g += -r * (J0 + Jt)     H += Jt (x) Jt   (or Jm (x) Jm when hess_mean)
 * MODE 2 ICLK: g += +r * J0            (no gradient, no H: InitialSelf / Std Hessians are constant)
 * With MAT the interface-visible arrays It, dIt_dx and Jt are also written (88 B/pixel).
 *
 * Memory-level parallelism: the streaming operands of pixel i+256 (grid point, template value, the
 * eight J0 columns) are fetched into registers before pixel i is processed, so every wave keeps two
 * rows of HBM requests in flight.  Sampling takes a wave-uniform fast path when, for all 64 lanes,
 * the centre sample and its four finite-difference neighbours lie in one interior bilinear cell (the
 * normal case): 4 texel loads, straight-line arithmetic, no divergent control flow.  Any lane near
 * the border or on an integer coordinate sends the wave through the general per-sample path.  Both
 * paths evaluate the reference's expressions in the reference's order.
 */
__device__ __forceinline__ double bilin(double t00, double t01, double t10, double t11, double dx, double dy) {
	return t00 * (1 - dx) * (1 - dy) + t01 * dx * (1 - dy) + t10 * (1 - dx) * dy + t11 * dx * dy;
}
/* true when (x, y) is sampled from the interior cell (lx, ly) with both upper neighbours lx+1, ly+1 */
__device__ __forceinline__ bool in_cell(double x, double y, int lx, int ly) {
	/* bitwise on purpose: six compares and five s_and instead of a chain of exec-masked branches */
	return (x >= 0) & (y >= 0) & ((int)x == lx) & ((int)y == ly) & ((x - lx) != 0) & ((y - ly) != 0);
}

/* index of (a, b), a <= b, in the upper-triangle order of the accumulator row (stride 8) */
__host__ __device__ constexpr int tri8(int a, int b) { return a * 8 - (a * (a - 1)) / 2 + (b - a); }
template <int S, int MODE>
struct PixIn {
	double2 p;
	double2 hp;
	double z;
	double i0;
	double j0[MODE == 0 ? 1 : S];
	int ch;   /* MC: the row's channel */
};
/* warped position of a grid point and the four texels of its bilinear cell, fetched one row ahead */
struct Tex {
	double wx, wy, cx, cy, D;
	double inv;        /* FAST: 1 / D */
	float t00, t01, t10, t11;
	int lx, ly;
	double lxd, lyd;   /* (double)lx, (double)ly */
	bool ok;   /* interior cell, non-integer coordinates: the texels above are the sample's own */
};

/*
 * Software pipeline (per thread, rows are 256 pixels apart):
 *   iteration i:  [texel loads of row i+1] -> [streaming loads of row i+2: grid point, I0, J0 columns]
 *                 -> arithmetic of row i -> [stores of row i]
 * vmcnt retires in order, so the texels of a row are requested before the younger streaming loads and
 * are consumed one iteration later, when everything older has long completed; each wave keeps two rows
 * of HBM reads plus one row of writes in flight.
 */

/* AM = MTFHIP_AM_SSD: the residual-weighted sums above.  AM = MTFHIP_AM_NCC: the same pass accumulates the raw moments
 * NCC's similarity, Jacobians and first-order Hessians are functions of (NCC.cc:124-389 restated in ncc_from_moments,
 * api_fused.hip) -- Gram(row) | sum Jt | sum It Jt | sum I0 Jt | sum It J0 | sum It, It^2, I0 It -- so an NCC iteration
 * needs no second pass over the pixels for the means; the partial rows are NCC_ACC_COUNT wide. */
/* FAST (lean launches only, MAT = false): tolerance-mode arithmetic -- one reciprocal per point, FMA-contracted warp / interpolant /
 * rows, and on the wave-uniform interior path the closed-form slope of the bilinear cell times the ROUNDED step the reference's 1e-8
 * central difference takes (fd_step, mtfhip_device.h) instead of four more samples: the reference's gradient without its per-pixel
 * rounding noise but WITH its systematic step quantisation, which is what brings H / g / dp inside 1e-5 of the reference-parameter
 * oracle.  The chained route (getImgGrad at the warped point) and the non-chained route (Homography.cc:803-827 / Affine.cc:293-313 +
 * cmptInitPixJacobian) are the same mathematical row but quantise differently, so both are instantiated (r04; r03 served both with
 * the chained form).  On integer coordinates, cell edges and the border the wave falls back to the replay of the reference's five
 * samples, SURVEY A4's corner case. */
/* PERSIST (kernels_persist.hip): the body runs once per iteration inside one launch; the warp and the state were written by another
 * workgroup a moment ago, so they are read with agent-scope loads and moved back to scalar registers (a plain load of memory the
 * kernel itself modifies would be kept in vector registers: 26 of them). */
/* the general sampler from its scalar arguments.  (Tried as a real call -- __attribute__((noinline)) -- so that the rare path's registers
 * would not count against the row loop's: no spill went away, and a 64 x 50 x 50 lean launch went from 13.1 to 14.6 us: inlined.) */
__device__ __forceinline__ double pix_val_call(const float *data, int w, int h, int stride, double x, double y) {
	ImgView im; im.data = data; im.w = w; im.h = h; im.stride = stride;
	return pix_val(im, x, y);
}
__device__ __forceinline__ double to_uniform(double v) {
	const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uniform_fresh_load(const double *p) { return to_uniform(ld_coh(p)); }
/* MC (kernels_fused_mc.hip): the multi-channel models -- MCSSD / MCNCC = SSD / NCC constructed with n_channels = 3 (AM/src/MCSSD.cc)
 * -- through the same pass.  A row of every per-pixel array is a (pixel, channel) pair, row = pixel * C + channel
 * (mc::getPixVals imgUtils.cc:867-882): the grid point is the pixel's, the texels are the channel's (interleaved image), and
 * the interpolant is evaluated in mc::PixVal's order -- the four weights first, then the weighted texels (imgUtils.h:505-551). */
__device__ __forceinline__ double bilin_mc(double t00, double t01, double t10, double t11, double dx, double dy) {
	const double ly_lx = (1 - dx) * (1 - dy), ly_ux = dx * (1 - dy), uy_lx = (1 - dx) * dy, uy_ux = dx * dy;
	return t00 * ly_lx + t01 * ly_ux + t10 * uy_lx + t11 * uy_ux;
}
/* COHROW: the workgroup's partial row leaves as write-through stores (the persistent loop, and the one-launch-per-pass kernel of
 * kernels_step.hip whose last-arriving workgroup reads every row in the same launch) */
template <int AM, int SSM, bool CHAINED, int MODE, bool MAT, bool FAST = false, bool PERSIST = false, bool MC = false, bool COHROW = PERSIST>
__device__ __forceinline__ void fused_lk_body(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk) {
	constexpr int S = (SSM == MTFHIP_SSM_HOMOGRAPHY) ? 8 : 6;
	constexpr bool NCC = AM == MTFHIP_AM_NCC;
	constexpr int K = NCC ? NCC_ACC_COUNT : 48;
	constexpr int ROW_LEN = NCC ? NCC_ACC_COUNT : ACC_COUNT;
	__shared__ double lds[4 * K];
	/* NCC + ESM carries 71 accumulators (142 VGPRs) and every instantiation of it sat at the 256-register limit with reloads inside the
	 * row loop (r03: 36-140 B of scratch per lane).  The sixteen sums that are touched once per row and feed nothing in the row -- sum
	 * I0 Jt and sum It J0 -- live in LDS instead (one slot per thread and sum, ds_add_f64 without return: the LDS pipe, not a VALU
	 * slot) and come back into the accumulator row just before the workgroup reduction. */
	constexpr bool PARK = NCC && MODE == 1;
	static_assert(NCC_ITJ0 == NCC_I0J + 8, "the parked accumulators are one contiguous run");
	/* PARK_I0J: sum I0 Jt parked as well (what the instantiation needs to stay clear of scratch, from -Rpass-analysis=kernel-resource-usage) */
#ifndef MTFHIP_PARK_FAST_CHAINED
#define MTFHIP_PARK_FAST_CHAINED 0   /* measured r04 (64 x 200 x 200 lean): 0 -> 26.7 us with 14 registers in scratch, 8 -> 28.4, 16 -> 29.9 */
#endif
	constexpr int NPARK = !PARK ? 0 : ((FAST && CHAINED) ? MTFHIP_PARK_FAST_CHAINED : 16);
	constexpr bool PARK_I0J = NPARK == 16, PARK_ITJ0 = NPARK >= 8;
	__shared__ double park[NPARK ? 16 * kBlock : 1];
	auto park_add = [&](int a, double v) {
		(void)__builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)&park[a * kBlock + (int)threadIdx.x], v);
	};
	if constexpr (NPARK > 0) {
#pragma unroll
		for (int a = 0; a < 16; ++a) park[a * kBlock + threadIdx.x] = 0.0;
	}
	const int t = blockIdx.y;
	const unsigned N = (unsigned)bv.N;
	/* The per-target scalars (live flag, warp, state) sit a scalar-load round trip behind the kernel arguments and
	 * the first row's streaming operands do not depend on them: the scalar loads are requested here, but nothing
	 * waits for them (no early exit, no derived constant) until the first row's vector loads have been issued
	 * (setup_target, called from run_rows).  They must stay ahead of the asm memory fences to remain s_loads. */
	/* branch-free: without a flag array the load is pointed at this target's warp (always readable) and ignored */
	const int *live_ptr = fa.active ? fa.active + t : reinterpret_cast<const int *>(bv.warps + 9 * t);
	const int live_word = *live_ptr;
	const int live = PERSIST ? 1 : (fa.active ? live_word : 1);   /* (the persistent loop tests the flag itself, once per iteration) */
	/* inline_warp (one target): the same scalar loads, pointed at the copy inside the kernel-argument segment -- the explicit
	 * arguments are laid out in declaration order at their natural alignment: bv, im, fa */
	static_assert(sizeof(BatchView) % 8 == 0 && sizeof(ImgView) % 8 == 0 && alignof(BatchView) == 8 && alignof(ImgView) <= 8 &&
		alignof(FusedArgs) == 8 && offsetof(FusedArgs, is) == offsetof(FusedArgs, iw) + 9 * sizeof(double),
		"kernarg layout assumed below: bv at 0, im right behind it, fa right behind im, iw[9] | is[8] contiguous");
	const char *kernarg = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
	const double *kw = reinterpret_cast<const double *>(kernarg + sizeof(BatchView) + sizeof(ImgView) + offsetof(FusedArgs, iw));
	const double *wsrc = (!PERSIST && fa.inline_warp) ? kw : bv.warps + 9 * t;
	const double *st = (!PERSIST && fa.inline_warp) ? kw + 9 : bv.states + 8 * t;
	Warp9 W;
	double st2, st3, st4, st5;
	if constexpr (PERSIST) {
#pragma unroll
		for (int q = 0; q < 9; ++q) W.m[q] = uniform_fresh_load(wsrc + q);
		st2 = uniform_fresh_load(st + 2); st3 = uniform_fresh_load(st + 3); st4 = uniform_fresh_load(st + 4); st5 = uniform_fresh_load(st + 5);
	} else {
		W = load_warp(wsrc);
		st2 = st[2]; st3 = st[3]; st4 = st[4]; st5 = st[5];
	}
	static_assert(!(MC && PERSIST), "no multi-channel instantiation of the persistent loop");
	const unsigned NPt = MC ? (unsigned)bv.NP : N, Cc = MC ? (unsigned)bv.C : 1u;   /* points per target, channels */
	const double2 *__restrict__ ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * NPt;
	const double *__restrict__ iz = bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * NPt;
	const double2 *__restrict__ ih = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_HXY]) + (size_t)t * NPt;
	const double *__restrict__ I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N;
	const double *__restrict__ J0 = bv.buf[MTFHIP_BUF_J0] + (size_t)t * N * S;
	const double *__restrict__ dI0 = bv.buf[MTFHIP_BUF_DI0_DX] + (size_t)t * N * 2;
	double *__restrict__ It = bv.buf[MTFHIP_BUF_IT] + (size_t)t * N;
	double *__restrict__ dIt = bv.buf[MTFHIP_BUF_DIT_DX] + (size_t)t * N * 2;
	double *__restrict__ Jt = bv.buf[MTFHIP_BUF_JT] + (size_t)t * N * S;
	const float *__restrict__ img = im.data;
	const int iw = im.w, ih_ = im.h, istride = im.stride;
	const float *__restrict__ img_row1 = img + istride;
	const bool unit_z = bv.unit_z != 0;
	const double eps = fa.grad_eps;
	const double gmult = fa.norm_mult / (2 * eps);
	double ex0, ex1, ex2, ey0, ey1, ey2;
	double aa, ab, ac, ad;   /* affine a,b,c,d (Affine.cc:216-217) */
	auto setup_target = [&]() {
#ifndef MTFHIP_NO_UNIFORM_WARP
		if constexpr (!PERSIST) {
			/* wsrc / st are generic pointers (kernarg segment or global), so the loads above are vector loads of a wave-uniform address
			 * and the warp would sit in 26 VGPRs for the whole pass (r03 ISA): move it to scalar registers here, behind the first row's
			 * vector loads (nothing waits for the warp before this point) */
#pragma unroll
			for (int q = 0; q < 9; ++q) W.m[q] = to_uniform(W.m[q]);
			st2 = to_uniform(st2); st3 = to_uniform(st3); st4 = to_uniform(st4); st5 = to_uniform(st5);
		}
#endif
		ex0 = W.m[0] * eps; ex1 = W.m[3] * eps; ex2 = W.m[6] * eps;
		ey0 = W.m[1] * eps; ey1 = W.m[4] * eps; ey2 = W.m[7] * eps;
		aa = st2 + 1; ab = st3; ac = st4; ad = st5 + 1;
	};

	double acc[K];
#pragma unroll
	for (int k = 0; k < K; ++k) acc[k] = 0.0;

	auto load_in = [&](unsigned i, auto uz, auto jr) {
		PixIn<S, MODE> in;
		constexpr bool JR = decltype(jr)::value;   /* J0 rows rebuilt from dI0_dx (2 loads) instead of read back (S loads) */
		const unsigned pi = MC ? i / Cc : i;   /* the row's pixel */
		if constexpr (MC) in.ch = (int)(i - pi * Cc);
#if MTFHIP_NT_LOAD
		typedef double d2v __attribute__((ext_vector_type(2)));
		{ const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(ip) + pi); in.p = make_double2(v.x, v.y); }
		in.i0 = __builtin_nontemporal_load(&I0[i]);
		if constexpr (MODE != 0) {
#pragma unroll
			for (int s = 0; s < S; ++s) in.j0[s] = __builtin_nontemporal_load(&J0[(unsigned)s * N + i]);
		} else {
			in.j0[0] = 0;
		}
		const unsigned o16 = pi * 16u, oz = pi * 8u;
#else
		const unsigned o8 = i * 8u, o16 = pi * 16u, oz = pi * 8u;
		in.p = ld_off<double2>(ip, o16);
		in.i0 = ld_off<double>(I0, o8);
		if constexpr (MODE != 0 && JR) {
			in.j0[0] = ld_off<double>(dI0, o8); in.j0[1] = ld_off<double>(dI0 + N, o8);
		} else if constexpr (MODE != 0) {
#pragma unroll
			for (int s = 0; s < S; ++s) in.j0[s] = ld_off<double>(J0 + (size_t)s * N, o8);
		} else {
			in.j0[0] = 0;
		}
#endif
		if constexpr (decltype(uz)::value) { in.hp = make_double2(0.0, 0.0); in.z = 1.0; }   /* hp is taken from p at use */
		else { in.hp = ld_off<double2>(ih, o16); in.z = ld_off<double>(iz, oz); }
		return in;
	};
	/* curr_pts_hm = curr_warp * init_pts_hm and its dehomogenisation (Homography.cc:86-90, Affine.cc:104),
	 * then the texel fetch of the bilinear cell */
	auto issue_tex = [&](const PixIn<S, MODE> &in, auto uz) {
		Tex tx;
		constexpr bool UZ = decltype(uz)::value;
		const double z = UZ ? 1.0 : in.z, hx = UZ ? in.p.x : in.hp.x, hy = UZ ? in.p.y : in.hp.y;
		tx.inv = 1.0;
		if constexpr (FAST && SSM == MTFHIP_SSM_HOMOGRAPHY) {
			tx.cx = fma(W.m[0], hx, fma(W.m[1], hy, UZ ? W.m[2] : W.m[2] * z));
			tx.cy = fma(W.m[3], hx, fma(W.m[4], hy, UZ ? W.m[5] : W.m[5] * z));
			tx.D = fma(W.m[6], hx, fma(W.m[7], hy, UZ ? W.m[8] : W.m[8] * z));
			tx.inv = rcp_fast(tx.D);
			tx.wx = tx.cx * tx.inv; tx.wy = tx.cy * tx.inv;
		} else if constexpr (FAST) {
			tx.wx = fma(W.m[0], hx, fma(W.m[1], hy, UZ ? W.m[2] : W.m[2] * z));
			tx.wy = fma(W.m[3], hx, fma(W.m[4], hy, UZ ? W.m[5] : W.m[5] * z));
			tx.cx = tx.wx; tx.cy = tx.wy; tx.D = 1.0;
		} else if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
			tx.cx = W.m[0] * hx + W.m[1] * hy + W.m[2] * z;
			tx.cy = W.m[3] * hx + W.m[4] * hy + W.m[5] * z;
			tx.D = W.m[6] * hx + W.m[7] * hy + W.m[8] * z;
			tx.wx = tx.cx / tx.D; tx.wy = tx.cy / tx.D;
		} else {
			tx.wx = W.m[0] * hx + W.m[1] * hy + W.m[2] * z;
			tx.wy = W.m[3] * hx + W.m[4] * hy + W.m[5] * z;
			tx.cx = tx.wx; tx.cy = tx.wy; tx.D = 1.0;
		}
		tx.lx = (int)tx.wx; tx.ly = (int)tx.wy;
		tx.lxd = (double)tx.lx; tx.lyd = (double)tx.ly;
		/* in_cell(wx, wy, (int)wx, (int)wy) with the trivially true terms dropped, and both upper neighbours inside */
		tx.ok = (tx.wx >= 0) & (tx.wy >= 0) & (tx.wx != tx.lxd) & (tx.wy != tx.lyd) & (tx.lx < iw - 1) & (tx.ly < ih_ - 1);
		const int sx = tx.ok ? tx.lx : 0, sy = tx.ok ? tx.ly : 0;
		const unsigned to = MC ? (unsigned)(sy * istride + sx * (int)Cc + in.ch) * 4u : (unsigned)(sy * istride + sx) * 4u;
#ifdef MTFHIP_EXPERIMENT_NOTEX
		tx.t00 = tx.t01 = tx.t10 = tx.t11 = (float)in.i0; (void)to;
#else
		const float *r0 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(img) + to);
		const float *r1 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(img_row1) + to);
		tx.t00 = r0[0]; tx.t01 = r0[MC ? Cc : 1u]; tx.t10 = r1[0]; tx.t11 = r1[MC ? Cc : 1u];
#endif
		return tx;
	};

	const int n_rows = fa.rows_per_block;
	const unsigned base = blockIdx.x * (unsigned)(kBlock * n_rows) + threadIdx.x;
	/* arithmetic + stores of one row; `cur` holds its streaming operands, `tcur` its position and texels */
	auto row_compute = [&](unsigned i, const PixIn<S, MODE> &cur, const Tex &tcur, auto jr) {
		constexpr bool JR = decltype(jr)::value;
#ifdef MTFHIP_EXPERIMENT_TRIVIAL   /* membench-equivalent body: same loads and stores, no arithmetic to speak of */
		{
			const double v = cur.p.x + cur.p.y + cur.i0 + tcur.wx;
			acc[44] += v;
			if constexpr (MAT) {
				MAT_STORE(&It[i], v); MAT_STORE(&dIt[i], v * 2); MAT_STORE(&dIt[N + i], v * 3);
#pragma unroll
				for (int s = 0; s < S; ++s) MAT_STORE(&Jt[(unsigned)s * N + i], (MODE != 0 ? cur.j0[s] : 0.0) + v);
			}
			return;
		}
#endif
		const double x = cur.p.x, y = cur.p.y;
		const double wx = tcur.wx, wy = tcur.wy, cx = tcur.cx, cy = tcur.cy, D = tcur.D;
		const int lx = tcur.lx, ly = tcur.ly;
		const double lxd = tcur.lxd, lyd = tcur.lyd;
		const unsigned o8 = i * 8u;
		/* the four finite-difference sample points */
		double px0, py0, px1, py1, px2, py2, px3, py3;
		auto fd_points = [&]() {
			if constexpr (CHAINED) {
				/* utils::getImgGrad at the warped point (imgUtils.cc:233-254) */
				px0 = wx + eps; py0 = wy; px1 = wx - eps; py1 = wy;
				px2 = wx; py2 = wy + eps; px3 = wx; py3 = wy - eps;
			} else if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
				/* Homography::updateGradPts SSM/src/Homography.cc:803-827 */
				double a0 = cx + ex0, a1 = cy + ex1, a2 = D + ex2;
				px0 = a0 / a2; py0 = a1 / a2;
				a0 = cx - ex0; a1 = cy - ex1; a2 = D - ex2;
				px1 = a0 / a2; py1 = a1 / a2;
				a0 = cx + ey0; a1 = cy + ey1; a2 = D + ey2;
				px2 = a0 / a2; py2 = a1 / a2;
				a0 = cx - ey0; a1 = cy - ey1; a2 = D - ey2;
				px3 = a0 / a2; py3 = a1 / a2;
			} else {
				/* Affine::updateGradPts SSM/src/Affine.cc:293-313 */
				px0 = wx + ex0; py0 = wy + ex1; px1 = wx - ex0; py1 = wy - ex1;
				px2 = wx + ey0; py2 = wy + ey1; px3 = wx - ey0; py3 = wy - ey1;
			}
		};
		constexpr bool QSTEP = FAST && !CHAINED && MODE != 2;   /* tolerance mode, non-chained: rounded steps instead of the four points */
		if constexpr (MODE != 2 && !QSTEP) fd_points();
		/* QSTEP: px0 - px1, py0 - py1 (x direction) and px2 - px3, py2 - py3 (y direction) of updateGradPts, taken of the ROUNDED
		 * numerators and denominators (Homography.cc:803-827: px = (cx +- ex0) / (D +- ex2), to second order in eps
		 * px0 - px1 = ((n0 - n1) - wx (d0 - d1)) / D; Affine.cc:293-313: px = wx +- ex0) */
		double dpx_x = 0, dpy_x = 0, dpx_y = 0, dpy_y = 0;
		if constexpr (QSTEP && SSM == MTFHIP_SSM_HOMOGRAPHY) {
			const double inv = tcur.inv;
			const double dd_x = fd_step_sym(D, ex2), dd_y = fd_step_sym(D, ey2);
			dpx_x = fma(-wx, dd_x, fd_step_sym(cx, ex0)) * inv; dpy_x = fma(-wy, dd_x, fd_step_sym(cy, ex1)) * inv;
			dpx_y = fma(-wx, dd_y, fd_step_sym(cx, ey0)) * inv; dpy_y = fma(-wy, dd_y, fd_step_sym(cy, ey1)) * inv;
		} else if constexpr (QSTEP) {
			dpx_x = fd_step_sym(wx, ex0); dpy_x = fd_step_sym(wy, ex1); dpx_y = fd_step_sym(wx, ey0); dpy_y = fd_step_sym(wy, ey1);
		}
		bool fast = tcur.ok;
		if constexpr (QSTEP) {
			/* all four points strictly inside the cell, with the whole step as the margin where half of it would do (a few more
			 * waves than necessary take the five-sample path, which is the reference's own arithmetic) */
			const double mx = fmax(fabs(dpx_x), fabs(dpx_y)), my = fmax(fabs(dpy_x), fabs(dpy_y));
			fast = fast & (wx - mx > lxd) & (wx + mx < lxd + 1) & (wy - my > lyd) & (wy + my < lyd + 1);
		} else if constexpr (MODE != 2 && CHAINED) {
			/* axis-aligned neighbours of a centre that is strictly inside the cell (eps > 0, rounding is monotonic):
			 * wx + eps >= wx > lx and wx - eps <= wx < lx + 1 hold already, so in_cell reduces to the other bound */
			fast = fast & (px0 < lxd + 1) & (px1 > lxd) & (py2 < lyd + 1) & (py3 > lyd);
		} else if constexpr (MODE != 2) {
			fast = fast & in_cell(px0, py0, lx, ly) & in_cell(px1, py1, lx, ly) & in_cell(px2, py2, lx, ly) &
				in_cell(px3, py3, lx, ly);
		}
		double it, gx = 0, gy = 0;
#ifdef MTFHIP_EXPERIMENT_NOMATH
		if (true) { it = tcur.t00 + tcur.t01 + tcur.t10 + tcur.t11 + wx; gx = wy; gy = px0 + py3; } else
#endif
		if (FAST && __builtin_amdgcn_ballot_w64(!fast) == 0) {
			/* closed form: value and both partial derivatives of the cell's interpolant from one evaluation */
			double v, bgx, bgy;
			bilin_fast(tcur.t00, tcur.t01, tcur.t10, tcur.t11, wx - lxd, wy - lyd, v, bgx, bgy);
			it = fma(fa.norm_mult, v, fa.norm_add);
			if constexpr (MODE != 2 && CHAINED) {
				/* utils::getImgGrad (imgUtils.cc:233-254) inside one cell: inc - dec = slope * ((wx + eps) - (wx - eps)) exactly,
				 * the ROUNDED step included (fd_step, mtfhip_device.h) */
				gx = bgx * (fd_step(wx, eps) * gmult); gy = bgy * (fd_step(wy, eps) * gmult);
			} else if constexpr (MODE != 2) {
				/* getWarpedImgGrad (imgUtils.cc:177-202) inside one cell: inc - dec = slope_x (px0 - px1) + slope_y (py0 - py1)
				 * (the cross term x0 y0 - x1 y1 = (x0 - x1) ybar + (y0 - y1) xbar is inside the slopes at the centre) */
				gx = fma(bgx, dpx_x, bgy * dpy_x) * gmult; gy = fma(bgx, dpx_y, bgy * dpy_y) * gmult;
			}
		} else if (!FAST && __builtin_amdgcn_ballot_w64(!fast) == 0) {
			const double t00 = tcur.t00, t01 = tcur.t01, t10 = tcur.t10, t11 = tcur.t11;
			auto bl = [&](double dx, double dy) { if constexpr (MC) return bilin_mc(t00, t01, t10, t11, dx, dy); else return bilin(t00, t01, t10, t11, dx, dy); };
			it = fa.norm_mult * bl(wx - lxd, wy - lyd) + fa.norm_add;
			if constexpr (MODE != 2) {
				double inc = bl(px0 - lxd, py0 - lyd);
				double dec = bl(px1 - lxd, py1 - lyd);
				gx = (inc - dec) * gmult;
				inc = bl(px2 - lxd, py2 - lyd);
				dec = bl(px3 - lxd, py3 - lyd);
				gy = (inc - dec) * gmult;
			}
		} else if constexpr (MC) {
			/* border / integer coordinates: the general sampler, sample by sample (mc::getPixVals, mc::getImgGrad imgUtils.cc:867-1005) */
			const int ch = cur.ch;
			it = fa.norm_mult * pix_val_mc(im, wx, wy, ch) + fa.norm_add;
			if constexpr (MODE != 2) {
				if constexpr (QSTEP) fd_points();
				gx = (pix_val_mc(im, px0, py0, ch) - pix_val_mc(im, px1, py1, ch)) * gmult;
				gy = (pix_val_mc(im, px2, py2, ch) - pix_val_mc(im, px3, py3, ch)) * gmult;
			}
		} else if constexpr (FAST) {
			/* border / integer coordinates / cell edges in a tolerance-mode launch: the reference's five samples one after the other
			 * (utils::getPixVal + getImgGrad, imgUtils.h:91-113, imgUtils.cc:233-254).  Rare, so written for few live registers, not
			 * for speed: the cached-cell form below keeps a dozen values alive across the hot loop's accumulators. */
			it = fa.norm_mult * pix_val_call(img, iw, ih_, istride, wx, wy) + fa.norm_add;
			if constexpr (MODE != 2) {
				if constexpr (QSTEP) fd_points();
				gx = (pix_val_call(img, iw, ih_, istride, px0, py0) - pix_val_call(img, iw, ih_, istride, px1, py1)) * gmult;
				gy = (pix_val_call(img, iw, ih_, istride, px2, py2) - pix_val_call(img, iw, ih_, istride, px3, py3)) * gmult;
			}
		} else {
			const Cell c = load_cell(im, wx, wy);
			it = fa.norm_mult * pix_val_cell(im, c, wx, wy) + fa.norm_add;
			if constexpr (MODE != 2) {
				double inc = pix_val_cell(im, c, px0, py0);
				double dec = pix_val_cell(im, c, px1, py1);
				gx = (inc - dec) * gmult;
				inc = pix_val_cell(im, c, px2, py2);
				dec = pix_val_cell(im, c, px3, py3);
				gy = (inc - dec) * gmult;
			}
		}
		const double r = it - cur.i0;
		if constexpr (NCC) {
			acc[NCC_IT] += it; acc[NCC_IT2] = fma(it, it, acc[NCC_IT2]); acc[NCC_I0IT] = fma(cur.i0, it, acc[NCC_I0IT]);
		} else {
			acc[44] = fma(r, r, acc[44]);
		}
		if constexpr (MAT) st_off<double>(It, o8, it);

		if constexpr (FAST) {
			/* Tolerance mode: the same sums, ordered for short live ranges (r04: the NCC instantiations sat at 256 VGPRs with reloads
			 * inside the row loop).  A steepest-descent row is linear in the pixel's (Ix, Iy) -- row = L(x, y) (Ix, Iy) for every SSM
			 * and route -- so the template's row is consumed before the current one is built, and ESM's mean row (hess_mean) is rebuilt
			 * from the mean gradient instead of keeping both rows alive. */
			constexpr bool HOM = SSM == MTFHIP_SSM_HOMOGRAPHY;
			auto sd_row = [&](double *o, double Ix, double Iy) {
				if constexpr (HOM) hom_row_fast(o, Ix, Iy, x, y);   /* Homography.cc:166-186, 282-289 */
				else { o[0] = Ix; o[1] = Iy; o[2] = Ix * x; o[3] = Ix * y; o[4] = Iy * x; o[5] = Iy * y; o[6] = o[7] = 0.0; }   /* Affine.cc:160-182 */
			};
			double Ix0 = 0.0, Iy0 = 0.0;
			double r0[8];
			if constexpr (MODE != 0) {
				if constexpr (JR) {
					/* Warped at the identity = gradient / z, Init = gradient (z folds to 1 in the unit-z instantiation; affine: z = 1) */
					Ix0 = cur.j0[0]; Iy0 = cur.j0[1];
					if constexpr (HOM) { const double inv0 = fa.j0_init_variant ? 1.0 : 1.0 / cur.z; Ix0 *= inv0; Iy0 *= inv0; }
					sd_row(r0, Ix0, Iy0);
				} else {
#pragma unroll
					for (int s = 0; s < S; ++s) r0[s] = cur.j0[s];
				}
#pragma unroll
				for (int s = 0; s < S; ++s) {
					if constexpr (PARK_ITJ0) park_add(8 + s, it * r0[s]);
					else if constexpr (NCC) acc[NCC_ITJ0 + s] = fma(it, r0[s], acc[NCC_ITJ0 + s]);
					else acc[36 + s] = fma(MODE == 1 ? -r : r, r0[s], acc[36 + s]);
				}
			}
			if constexpr (MODE != 2) {
				double Ix, Iy;
				if constexpr (!CHAINED) {
					Ix = gx; Iy = gy;   /* cmptInitPixJacobian of the warped image's gradient */
				} else if constexpr (HOM) {
					/* Homography.cc:252-264 with the point's reciprocal reused */
					const double dwx_dx = fma(-W.m[6], wx, W.m[0]), dwx_dy = fma(-W.m[7], wx, W.m[1]);
					const double dwy_dx = fma(-W.m[6], wy, W.m[3]), dwy_dy = fma(-W.m[7], wy, W.m[4]);
					Ix = fma(dwx_dx, gx, dwy_dx * gy) * tcur.inv;
					Iy = fma(dwx_dy, gx, dwy_dy * gy) * tcur.inv;
				} else {
					Ix = fma(gx, aa, gy * ac); Iy = fma(gx, ab, gy * ad);   /* Affine.cc:213-242, factored */
				}
				double row[8];
				sd_row(row, Ix, Iy);
#pragma unroll
				for (int s = 0; s < S; ++s) {
					if constexpr (NCC) {
						acc[NCC_SJ + s] += row[s];
						acc[NCC_ITJ + s] = fma(it, row[s], acc[NCC_ITJ + s]);
						if constexpr (PARK_I0J) park_add(s, cur.i0 * row[s]);
						else acc[NCC_I0J + s] = fma(cur.i0, row[s], acc[NCC_I0J + s]);
					} else {
						acc[36 + s] = fma(-r, row[s], acc[36 + s]);
					}
				}
				if constexpr (MODE == 1) {
					if (fa.hess_mean) {
						if constexpr (JR) sd_row(row, 0.5 * (Ix0 + Ix), 0.5 * (Iy0 + Iy));
						else {
#pragma unroll
							for (int s = 0; s < S; ++s) row[s] = 0.5 * (r0[s] + row[s]);
						}
					}
				}
				int k = 0;
#pragma unroll
				for (int a = 0; a < 8; ++a)
#pragma unroll
					for (int b = a; b < 8; ++b) {
						if (a < S && b < S) acc[k] = fma(row[a], row[b], acc[k]);
						++k;
					}
			}
			return;
		}

		double row[8];
		if constexpr (MODE != 2) {
			if constexpr (MAT) { st_off<double>(dIt, o8, gx); st_off<double>(dIt + N, o8, gy); }
			if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
				if constexpr (CHAINED) {
					/* Homography::cmptWarpedPixJacobian SSM/src/Homography.cc:231-294 */
					double inv_det = 1.0 / D;
					double dwx_dx = (W.m[0] - W.m[6] * wx), dwx_dy = (W.m[1] - W.m[7] * wx);
					double dwy_dx = (W.m[3] - W.m[6] * wy), dwy_dy = (W.m[4] - W.m[7] * wy);
					double Ix = (dwx_dx * gx + dwy_dx * gy) * inv_det;
					double Iy = (dwx_dy * gx + dwy_dy * gy) * inv_det;
					hom_row(row, Ix, Iy, x, y, x, y);
				} else {
					/* Homography::cmptInitPixJacobian SSM/src/Homography.cc:157-191 */
					hom_row(row, gx, gy, x, y, x, y);
				}
			} else {
				double Ixx = gx * x, Ixy = gx * y, Iyy = gy * y, Iyx = gy * x;
				if constexpr (CHAINED) {
					/* Affine::cmptWarpedPixJacobian SSM/src/Affine.cc:213-242 */
					row[0] = gx * aa + gy * ac; row[1] = gx * ab + gy * ad;
					row[2] = Ixx * aa + Iyx * ac; row[3] = Ixy * aa + Iyy * ac;
					row[4] = Ixx * ab + Iyx * ad; row[5] = Ixy * ab + Iyy * ad;
				} else {
					/* Affine::cmptInitPixJacobian SSM/src/Affine.cc:160-182 */
					row[0] = gx; row[1] = gy; row[2] = Ixx; row[3] = Ixy; row[4] = Iyx; row[5] = Iyy;
				}
				row[6] = row[7] = 0.0;
			}
			if constexpr (MAT) {
#pragma unroll
				for (int s = 0; s < S; ++s) st_off<double>(Jt + (size_t)s * N, o8, row[s]);
			}
		}

		/* the template's steepest-descent row: read back, or rebuilt from dI0_dx with the expressions (and operation
		 * order) k_pix_jacobian used when J0 was produced -- Warped at the identity warp by a chained initialize
		 * (Homography.cc:231-294 with curr_warp = I, curr_pts_hm = init_pts_hm), Init by a non-chained initialize and by
		 * setRegion (NT/ESM.cc:153) -- so the bits are those of the stored matrix, for 16 B/px of traffic instead of 8 S */
		double r0[8];
		if constexpr (MODE != 0) {
			if constexpr (JR) {
				const double g0x = cur.j0[0], g0y = cur.j0[1];
				if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
					double Ix0 = g0x, Iy0 = g0y;
					if (!fa.j0_init_variant) {   /* produced by cmptWarpedPixJacobian at the identity warp (chained initialize) */
						const double inv_det0 = 1.0 / cur.z;
						const double dwx_dx = (1.0 - 0.0 * x), dwx_dy = (0.0 - 0.0 * x), dwy_dx = (0.0 - 0.0 * y), dwy_dy = (1.0 - 0.0 * y);
						Ix0 = (dwx_dx * g0x + dwy_dx * g0y) * inv_det0;
						Iy0 = (dwx_dy * g0x + dwy_dy * g0y) * inv_det0;
					}
					hom_row(r0, Ix0, Iy0, x, y, x, y);
				} else {
					const double Ixx0 = g0x * x, Ixy0 = g0x * y, Iyy0 = g0y * y, Iyx0 = g0y * x;
					if (!fa.j0_init_variant) {   /* Affine.cc:213-242 with a = d = 1, b = c = 0 */
						r0[0] = g0x * 1.0 + g0y * 0.0; r0[1] = g0x * 0.0 + g0y * 1.0;
						r0[2] = Ixx0 * 1.0 + Iyx0 * 0.0; r0[3] = Ixy0 * 1.0 + Iyy0 * 0.0;
						r0[4] = Ixx0 * 0.0 + Iyx0 * 1.0; r0[5] = Ixy0 * 0.0 + Iyy0 * 1.0;
					} else {
						r0[0] = g0x; r0[1] = g0y; r0[2] = Ixx0; r0[3] = Ixy0; r0[4] = Iyx0; r0[5] = Iyy0;
					}
					r0[6] = r0[7] = 0.0;
				}
			} else {
#pragma unroll
				for (int s = 0; s < S; ++s) r0[s] = cur.j0[s];
			}
		}
		if constexpr (NCC) {
			if constexpr (MODE != 2) {
#pragma unroll
				for (int s = 0; s < S; ++s) {
					acc[NCC_SJ + s] += row[s];
					acc[NCC_ITJ + s] = fma(it, row[s], acc[NCC_ITJ + s]);
					if constexpr (PARK_I0J) park_add(s, cur.i0 * row[s]);
					else acc[NCC_I0J + s] = fma(cur.i0, row[s], acc[NCC_I0J + s]);
				}
			}
			if constexpr (MODE != 0) {
#pragma unroll
				for (int s = 0; s < S; ++s) {
					if constexpr (PARK_ITJ0) park_add(8 + s, it * r0[s]);
					else acc[NCC_ITJ0 + s] = fma(it, r0[s], acc[NCC_ITJ0 + s]);
				}
			}
			if constexpr (MODE == 1) {
				if (fa.hess_mean) {
#pragma unroll
					for (int s = 0; s < S; ++s) row[s] = (r0[s] + row[s]) / 2.0;
				}
			}
		} else if constexpr (MODE == 0) {
			const double v = -r;
#pragma unroll
			for (int s = 0; s < S; ++s) acc[36 + s] = fma(v, row[s], acc[36 + s]);
		} else if constexpr (MODE == 1) {
			const double v = -r;
#pragma unroll
			for (int s = 0; s < S; ++s) acc[36 + s] = fma(v, r0[s] + row[s], acc[36 + s]);
			if (fa.hess_mean) {
#pragma unroll
				for (int s = 0; s < S; ++s) row[s] = (r0[s] + row[s]) / 2.0;
			}
		} else {
#pragma unroll
			for (int s = 0; s < S; ++s) acc[36 + s] = fma(r, r0[s], acc[36 + s]);
		}
		if constexpr (MODE != 2) {
#ifndef MTFHIP_EXPERIMENT_NOACC
			int k = 0;
#pragma unroll
			for (int a = 0; a < 8; ++a)
#pragma unroll
				for (int b = a; b < 8; ++b) {
					if (a < S && b < S) acc[k] = fma(row[a], row[b], acc[k]);
					++k;
				}
#endif
		}
	};
	/* Streaming operands of the next row are requested before the current row is processed.  Every load of the
	 * loop over full rows is issued unconditionally (the prefetch index is clamped into the target instead of being
	 * guarded, the unit-z variant is chosen at compile time, the partial last row is peeled off): the number of
	 * memory operations issued after a row's texel fetch is then a compile-time constant and the compiler can wait
	 * for the texels with `s_waitcnt vmcnt(<next-row loads>)` and for the next row with `vmcnt(<stores>)`.
	 * With guarded loads it has to assume the shortest path and emits vmcnt(0), which silently serialises the
	 * prefetch behind the current row (that is what the ISA of the first version did). */
	auto run_rows = [&](auto uz, auto jr) {
		const unsigned blk_first = blockIdx.x * (unsigned)(kBlock * n_rows);
		/* full 256-pixel rows of this workgroup: no lane is masked, so nothing in the loop body is conditional */
		int full = 0;
		if (blk_first < N) {
			const unsigned avail = (N - blk_first) / kBlock;
			full = avail < (unsigned)n_rows ? (int)avail : n_rows;
		}
		if (full > 0) {
			PixIn<S, MODE> cur = load_in(base, uz, jr);
			asm volatile("" ::: "memory");
			setup_target();
			if (!live) return;
#pragma unroll 1
			for (int kk = 0; kk < full; ++kk) {
				const unsigned i = base + (unsigned)kk * kBlock;
				const Tex tcur = issue_tex(cur, uz);
				asm volatile("" ::: "memory");      /* texel fetch first, then the next row's operands: keeps the order */
				const unsigned inext = i + kBlock;
				const PixIn<S, MODE> nxt = load_in(inext < N ? inext : N - 1, uz, jr);
				asm volatile("" ::: "memory");
				row_compute(i, cur, tcur, jr);
				cur = nxt;
			}
		}
		/* the partial last row of a target (only the workgroup that owns the end of the patch gets here) */
		if (full < n_rows) {
			const unsigned i = base + (unsigned)full * kBlock;
			if (full == 0) { setup_target(); if (!live) return; }
			if (i < N) {
				const PixIn<S, MODE> c = load_in(i, uz, jr);
				const Tex tc = issue_tex(c, uz);
				row_compute(i, c, tc, jr);
			}
		}
	};
	if constexpr (MODE == 0) {      /* no template row in the FCLK accumulation */
		if (unit_z) run_rows(std::true_type{}, std::false_type{}); else run_rows(std::false_type{}, std::false_type{});
	} else if (fa.j0_recompute) {
		if (unit_z) run_rows(std::true_type{}, std::true_type{}); else run_rows(std::false_type{}, std::true_type{});
	} else {
		if (unit_z) run_rows(std::true_type{}, std::false_type{}); else run_rows(std::false_type{}, std::false_type{});
	}
	if (!live) return;
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * ROW_LEN;
	if constexpr (NPARK > 0) {   /* (a thread's LDS operations execute in order: its own sums are complete) */
#pragma unroll
		for (int a = PARK_I0J ? 0 : 8; a < (PARK_ITJ0 ? 16 : 8); ++a) acc[NCC_I0J + a] = park[a * kBlock + threadIdx.x];
	}
	block_reduce_store<K, COHROW>(acc, dst, lds);
	if (!PERSIST && fa.inline_warp && blockIdx.x == 0 && threadIdx.x < 17) {   /* keep the device copy current for whoever reads it next */
		const double v = kw[threadIdx.x];   /* iw[9] | is[8] */
		if (threadIdx.x < 9) bv.warps[9 * t + threadIdx.x] = v;
		else bv.states[8 * t + threadIdx.x - 9] = v;
	}
}
} // namespace mtfhip
#endif
