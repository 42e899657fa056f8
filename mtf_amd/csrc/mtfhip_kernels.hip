/*
 * mtfhip_kernels.hip -- hand-written CDNA4 (gfx950) kernels of MTF's Lucas-Kanade inner loop.
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
#include <type_traits>

#include "mtfhip_internal.h"

/* tuning knobs of the fused kernel ("fused kernel tuning") */
#ifndef MTFHIP_FUSED_WAVES
#define MTFHIP_FUSED_WAVES 2   /* minimum waves per SIMD requested from the register allocator */
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

struct Warp9 { double m[9]; };
__device__ __forceinline__ Warp9 load_warp(const double *p) {
	Warp9 W;
#pragma unroll
	for (int i = 0; i < 9; ++i) W.m[i] = p[i];
	return W;
}

/* halving butterfly over the 64 lanes of a wave: on entry every lane holds K partial sums in
 * v[0..K); on exit slot j of lane l holds the wave total of index final_index<K,32>(j, l). */
template <int K, int MASK>
__device__ __forceinline__ void wave_halve(double *v, int lane) {
	if constexpr (MASK == 0) {
		return;
	} else if constexpr (K % 2 == 0) {
		constexpr int H = K / 2;
		const bool up = (lane & MASK) != 0;
#pragma unroll
		for (int j = 0; j < H; ++j) {
			double keep = up ? v[j + H] : v[j];
			double send = up ? v[j] : v[j + H];
			v[j] = keep + __shfl_xor(send, MASK);
		}
		wave_halve<H, (MASK >> 1)>(v, lane);
	} else {
#pragma unroll
		for (int j = 0; j < K; ++j) v[j] += __shfl_xor(v[j], MASK);
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

/* reduce K per-thread accumulators over the workgroup and write them to dst[0..K) */
template <int K, bool COHERENT = false>
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
		/* COHERENT: written through to the device coherence point (sc1), for readers on another XCD in the same launch */
		if constexpr (COHERENT) __hip_atomic_store(&dst[threadIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		else dst[threadIdx.x] = s;
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

/* ===================================================================== */
/* StateSpaceModel kernels                                                */
/* ===================================================================== */

/* Sample grid of a target from its corners: utils::getNormUnitSquarePts (Utilities/src/warpUtils.cc:15-34,
 * LinSpaced = lo + i*step with the last element pinned to hi) pushed through the 4-corner DLT warp
 * (ProjectiveBase::getPtsFromCorners SSM/src/ProjectiveBase.cc:20-25), then the bookkeeping of
 * Homography::setCorners (Homography.cc:61-69: init_pts_hm keeps the un-normalised third row) or
 * Affine::setCorners (Affine.cc:74-87: init_pts_hm is re-homogenised, third row = 1). */
__device__ __forceinline__ double lin_spaced(int i, int n, double lo, double hi) {
	if (n == 1 || i == n - 1) return hi;
	return lo + i * ((hi - lo) / (n - 1));
}
__global__ __launch_bounds__(kBlock) void k_init_grid(BatchView bv, const double *w0_all, int resx, int resy,
	double lo_x, double lo_y, double hi_x, double hi_y, int force_unit_z) {
	const int t = blockIdx.y;
	const Warp9 W = load_warp(w0_all + 9 * t);
	double2 *ip = reinterpret_cast<double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * bv.NP;
	double2 *cp = reinterpret_cast<double2 *>(bv.buf[MTFHIP_BUF_CURR_PTS]) + (size_t)t * bv.NP;
	double *iz = bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * bv.NP;
	double *cz = bv.buf[MTFHIP_BUF_CURR_Z] + (size_t)t * bv.NP;
	double2 *ih = reinterpret_cast<double2 *>(bv.buf[MTFHIP_BUF_INIT_HXY]) + (size_t)t * bv.NP;
	double2 *ch = reinterpret_cast<double2 *>(bv.buf[MTFHIP_BUF_CURR_HXY]) + (size_t)t * bv.NP;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < bv.NP; i += gridDim.x * kBlock) {
		const int col = i % resx, row = i / resx;
		const double nx = lin_spaced(col, resx, lo_x, hi_x), ny = lin_spaced(row, resy, lo_y, hi_y);
		const double X = W.m[0] * nx + W.m[1] * ny + W.m[2] * 1.0;
		const double Y = W.m[3] * nx + W.m[4] * ny + W.m[5] * 1.0;
		const double Z = W.m[6] * nx + W.m[7] * ny + W.m[8] * 1.0;
		const double2 p = make_double2(X / Z, Y / Z);
		const double z = force_unit_z ? 1.0 : Z;
		/* affine re-homogenises (x, y, 1); homography keeps (X, Y, Z) */
		const double2 hxy = force_unit_z ? p : make_double2(X, Y);
		ip[i] = p; cp[i] = p; iz[i] = z; cz[i] = z; ih[i] = hxy; ch[i] = hxy;
	}
}

/* curr_pts_hm = curr_warp * init_pts_hm, dehomogenise (ProjectiveBase::setState
 * SSM/src/ProjectiveBase.cc:41-49, Homography::compositionalUpdate Homography.cc:86-90);
 * affine: curr_pts = curr_warp.topRows<2>() * init_pts_hm (Affine.cc:104,113) */
__global__ __launch_bounds__(kBlock) void k_apply_warp(BatchView bv) {
	const int t = blockIdx.y;
	const Warp9 W = load_warp(bv.warps + 9 * t);
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * bv.NP;
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * bv.NP;
	const double2 *ih = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_HXY]) + (size_t)t * bv.NP;
	double2 *cp = reinterpret_cast<double2 *>(bv.buf[MTFHIP_BUF_CURR_PTS]) + (size_t)t * bv.NP;
	double2 *ch = reinterpret_cast<double2 *>(bv.buf[MTFHIP_BUF_CURR_HXY]) + (size_t)t * bv.NP;
	double *cz = bv.buf[MTFHIP_BUF_CURR_Z] + (size_t)t * bv.NP;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < bv.NP; i += gridDim.x * kBlock) {
		double2 hp = bv.unit_z ? ip[i] : ih[i];
		double z = bv.unit_z ? 1.0 : iz[i];
		double hx = hp.x, hy = hp.y;
		double2 o;
		if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
			double cx = W.m[0] * hx + W.m[1] * hy + W.m[2] * z;
			double cy = W.m[3] * hx + W.m[4] * hy + W.m[5] * z;
			double d = W.m[6] * hx + W.m[7] * hy + W.m[8] * z;
			o.x = cx / d; o.y = cy / d;
			cz[i] = d;
			ch[i] = make_double2(cx, cy);
		} else {
			o.x = W.m[0] * hx + W.m[1] * hy + W.m[2] * z;
			o.y = W.m[3] * hx + W.m[4] * hy + W.m[5] * z;
			cz[i] = 1.0;
			ch[i] = o;
		}
		cp[i] = o;
	}
}

/* Homography::updateGradPts SSM/src/Homography.cc:803-827 ; Affine::updateGradPts Affine.cc:293-313 */
__global__ __launch_bounds__(kBlock) void k_grad_pts(BatchView bv, double eps) {
	const int t = blockIdx.y;
	const Warp9 W = load_warp(bv.warps + 9 * t);
	const double2 *cp = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_CURR_PTS]) + (size_t)t * bv.NP;
	const double *cz = bv.buf[MTFHIP_BUF_CURR_Z] + (size_t)t * bv.NP;
	const double2 *ch = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_CURR_HXY]) + (size_t)t * bv.NP;
	double *gp = bv.buf[MTFHIP_BUF_GRAD_PTS] + (size_t)t * bv.NP * 8;
	const double dx0 = W.m[0] * eps, dx1 = W.m[3] * eps, dx2 = W.m[6] * eps;
	const double dy0 = W.m[1] * eps, dy1 = W.m[4] * eps, dy2 = W.m[7] * eps;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < bv.NP; i += gridDim.x * kBlock) {
		double2 p = cp[i];
		double g[8];
		if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
			double2 h = ch[i];
			double q0 = h.x, q1 = h.y, q2 = cz[i];
			double a0 = q0 + dx0, a1 = q1 + dx1, a2 = q2 + dx2;
			g[0] = a0 / a2; g[1] = a1 / a2;
			a0 = q0 - dx0; a1 = q1 - dx1; a2 = q2 - dx2;
			g[2] = a0 / a2; g[3] = a1 / a2;
			a0 = q0 + dy0; a1 = q1 + dy1; a2 = q2 + dy2;
			g[4] = a0 / a2; g[5] = a1 / a2;
			a0 = q0 - dy0; a1 = q1 - dy1; a2 = q2 - dy2;
			g[6] = a0 / a2; g[7] = a1 / a2;
		} else {
			g[0] = p.x + dx0; g[1] = p.y + dx1;
			g[2] = p.x - dx0; g[3] = p.y - dx1;
			g[4] = p.x + dy0; g[5] = p.y + dy1;
			g[6] = p.x - dy0; g[7] = p.y - dy1;
		}
		double2 *o = reinterpret_cast<double2 *>(gp + (size_t)i * 8);
		o[0] = make_double2(g[0], g[1]); o[1] = make_double2(g[2], g[3]);
		o[2] = make_double2(g[4], g[5]); o[3] = make_double2(g[6], g[7]);
	}
}

/* ===================================================================== */
/* ImageBase kernels                                                      */
/* ===================================================================== */

/* utils::getPixVals Utilities/src/imgUtils.cc:163-173 */
__global__ __launch_bounds__(kBlock) void k_sample(int N, ImgView im, const double *pts_all, double *out_all,
	double mult, double add) {
	const int t = blockIdx.y;
	const double2 *pts = reinterpret_cast<const double2 *>(pts_all) + (size_t)t * N;
	double *out = out_all + (size_t)t * N;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		double2 p = pts[i];
		out[i] = mult * pix_val(im, p.x, p.y) + add;
	}
}

/* utils::getImgGrad Utilities/src/imgUtils.cc:233-254 */
__global__ __launch_bounds__(kBlock) void k_img_grad(int N, ImgView im, const double *pts_all, double *grad_all,
	double eps, double pix_mult) {
	const int t = blockIdx.y;
	const double2 *pts = reinterpret_cast<const double2 *>(pts_all) + (size_t)t * N;
	double *grad = grad_all + (size_t)t * N * 2;
	const double mult = pix_mult / (2 * eps);
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		double2 p = pts[i];
		Cell c = load_cell(im, p.x, p.y);
		double inc = pix_val_cell(im, c, p.x + eps, p.y);
		double dec = pix_val_cell(im, c, p.x - eps, p.y);
		grad[i] = (inc - dec) * mult;
		inc = pix_val_cell(im, c, p.x, p.y + eps);
		dec = pix_val_cell(im, c, p.x, p.y - eps);
		grad[N + i] = (inc - dec) * mult;
	}
}

/* utils::getWarpedImgGrad Utilities/src/imgUtils.cc:177-202 */
__global__ __launch_bounds__(kBlock) void k_warped_img_grad(int N, ImgView im, const double *gp_all, double *grad_all,
	double eps, double pix_mult) {
	const int t = blockIdx.y;
	const double *gp = gp_all + (size_t)t * N * 8;
	double *grad = grad_all + (size_t)t * N * 2;
	const double mult = pix_mult / (2 * eps);
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		const double2 *q = reinterpret_cast<const double2 *>(gp + (size_t)i * 8);
		double2 a = q[0], b = q[1], c2 = q[2], d = q[3];
		Cell c = load_cell(im, a.x, a.y);
		double inc = pix_val_cell(im, c, a.x, a.y);
		double dec = pix_val_cell(im, c, b.x, b.y);
		grad[i] = (inc - dec) * mult;
		inc = pix_val_cell(im, c, c2.x, c2.y);
		dec = pix_val_cell(im, c, d.x, d.y);
		grad[N + i] = (inc - dec) * mult;
	}
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
/* mc::getPixVals imgUtils.cc:867-882 */
__global__ __launch_bounds__(kBlock) void k_sample_mc(int NP, int C, ImgView im, const double *pts_all, double *out_all, double mult, double add) {
	const int t = blockIdx.y, P = NP * C;
	const double2 *pts = reinterpret_cast<const double2 *>(pts_all) + (size_t)t * NP;
	double *out = out_all + (size_t)t * P;
	for (int q = blockIdx.x * kBlock + threadIdx.x; q < P; q += gridDim.x * kBlock) {
		const double2 p = pts[q / C];
		out[q] = mult * pix_val_mc(im, p.x, p.y, q % C) + add;
	}
}
/* mc::getImgGrad imgUtils.cc:977-1005 ; mc::getWarpedImgGrad :914-944 (gp != NULL) */
__global__ __launch_bounds__(kBlock) void k_img_grad_mc(int NP, int C, ImgView im, const double *pts_all, const double *gp_all,
	double *grad_all, double eps, double pix_mult) {
	const int t = blockIdx.y, P = NP * C;
	double *grad = grad_all + (size_t)t * P * 2;
	const double mult = pix_mult / (2 * eps);
	for (int q = blockIdx.x * kBlock + threadIdx.x; q < P; q += gridDim.x * kBlock) {
		const int i = q / C, ch = q % C;
		double ix, dx, iy, dy;
		if (gp_all) {
			const double2 *g = reinterpret_cast<const double2 *>(gp_all + ((size_t)t * NP + i) * 8);
			ix = pix_val_mc(im, g[0].x, g[0].y, ch); dx = pix_val_mc(im, g[1].x, g[1].y, ch);
			iy = pix_val_mc(im, g[2].x, g[2].y, ch); dy = pix_val_mc(im, g[3].x, g[3].y, ch);
		} else {
			const double2 p = (reinterpret_cast<const double2 *>(pts_all) + (size_t)t * NP)[i];
			ix = pix_val_mc(im, p.x + eps, p.y, ch); dx = pix_val_mc(im, p.x - eps, p.y, ch);
			iy = pix_val_mc(im, p.x, p.y + eps, ch); dy = pix_val_mc(im, p.x, p.y - eps, ch);
		}
		grad[q] = (ix - dx) * mult;
		grad[P + q] = (iy - dy) * mult;
	}
}
/* mc::getImgHess imgUtils.cc:1127-1168 ; mc::getWarpedImgHess :1036-1075 (hp != NULL) */
__global__ __launch_bounds__(kBlock) void k_img_hess_mc(int NP, int C, ImgView im, const double *pts_all, const double *hp_all,
	double *hess_all, double eps, double pix_mult) {
	const int t = blockIdx.y, P = NP * C;
	double2 *hess = reinterpret_cast<double2 *>(hess_all + (size_t)t * P * 4);
	const double eps2 = 2 * eps, mult = pix_mult / (eps2 * eps2);
	for (int q = blockIdx.x * kBlock + threadIdx.x; q < P; q += gridDim.x * kBlock) {
		const int i = q / C, ch = q % C;
		const double2 p = (reinterpret_cast<const double2 *>(pts_all) + (size_t)t * NP)[i];
		const double c = pix_val_mc(im, p.x, p.y, ch);
		double hxx, hyy, hxy;
		if (hp_all) {
			const double2 *s = reinterpret_cast<const double2 *>(hp_all + ((size_t)t * NP + i) * 16);
			hxx = (pix_val_mc(im, s[0].x, s[0].y, ch) + pix_val_mc(im, s[1].x, s[1].y, ch) - 2 * c) * mult;
			hyy = (pix_val_mc(im, s[2].x, s[2].y, ch) + pix_val_mc(im, s[3].x, s[3].y, ch) - 2 * c) * mult;
			hxy = ((pix_val_mc(im, s[4].x, s[4].y, ch) + pix_val_mc(im, s[5].x, s[5].y, ch)) -
				(pix_val_mc(im, s[6].x, s[6].y, ch) + pix_val_mc(im, s[7].x, s[7].y, ch))) * mult;
		} else {
			hxx = (pix_val_mc(im, p.x + eps2, p.y, ch) + pix_val_mc(im, p.x - eps2, p.y, ch) - 2 * c) * mult;
			hyy = (pix_val_mc(im, p.x, p.y + eps2, ch) + pix_val_mc(im, p.x, p.y - eps2, ch) - 2 * c) * mult;
			const double inc_x = p.x + eps, dec_x = p.x - eps, inc_y = p.y + eps, dec_y = p.y - eps;
			hxy = ((pix_val_mc(im, inc_x, inc_y, ch) + pix_val_mc(im, dec_x, dec_y, ch)) -
				(pix_val_mc(im, inc_x, dec_y, ch) + pix_val_mc(im, dec_x, inc_y, ch))) * mult;
		}
		hess[2 * q] = make_double2(hxx, hxy);
		hess[2 * q + 1] = make_double2(hxy, hyy);
	}
}

/* SSM pixel Jacobians as stand-alone ops (the fused kernel inlines the same row formulas):
 * Homography.cc:157-191 (init), :193-229 (pix), :231-294 (warped), :296-358 (approx);
 * Affine.cc:160-182 (init = pix), :213-242 (warped), :184-211 (approx) */
__global__ __launch_bounds__(kBlock) void k_pix_jacobian(BatchView bv, int variant, const double *grad_all, double *J_all) {
	const int t = blockIdx.y, N = bv.N, S = bv.S, NP = bv.NP, C = bv.C;
	const Warp9 W = load_warp(bv.warps + 9 * t);
	const double *st = bv.states + 8 * t;
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * NP;
	const double2 *cp = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_CURR_PTS]) + (size_t)t * NP;
	const double *cz = bv.buf[MTFHIP_BUF_CURR_Z] + (size_t)t * NP;
	const double *grad = grad_all + (size_t)t * N * 2;
	double *J = J_all + (size_t)t * N * S;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		const int pt = C == 1 ? i : i / C;   /* row (pixel, channel) -> sample point (Homography.cc:160-189 inner ch loop) */
		double2 p0 = ip[pt];
		double x = p0.x, y = p0.y;
		double gx = grad[i], gy = grad[N + i];
		double r[8];
		if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
			if (variant == MTFHIP_JAC_INIT) {
				hom_row(r, gx, gy, x, y, x, y);
			} else if (variant == MTFHIP_JAC_PIX) {
				double2 c = cp[pt];
				double inv_d = 1.0 / cz[pt];
				hom_row(r, gx * inv_d, gy * inv_d, x, y, c.x, c.y);
			} else if (variant == MTFHIP_JAC_WARPED) {
				double2 c = cp[pt];
				double inv_det = 1.0 / cz[pt];
				double dwx_dx = (W.m[0] - W.m[6] * c.x), dwx_dy = (W.m[1] - W.m[7] * c.x);
				double dwy_dx = (W.m[3] - W.m[6] * c.y), dwy_dy = (W.m[4] - W.m[7] * c.y);
				double Ix = (dwx_dx * gx + dwy_dx * gy) * inv_det;
				double Iy = (dwx_dy * gx + dwy_dy * gy) * inv_det;
				hom_row(r, Ix, Iy, x, y, x, y);
			} else {
				double2 c = cp[pt];
				double a = (W.m[0] - W.m[6] * c.x), b = (W.m[1] - W.m[7] * c.x);
				double cc = (W.m[3] - W.m[6] * c.y), d = (W.m[4] - W.m[7] * c.y);
				double inv_factor = 1.0 / (a * d - b * cc);
				double Ix = (d * gx - cc * gy) * inv_factor;
				double Iy = (a * gy - b * gx) * inv_factor;
				hom_row(r, Ix, Iy, x, y, c.x, c.y);
			}
		} else {
			double a = st[2] + 1, b = st[3], c = st[4], d = st[5] + 1;
			double Ixx = gx * x, Ixy = gx * y, Iyy = gy * y, Iyx = gy * x;
			if (variant == MTFHIP_JAC_INIT || variant == MTFHIP_JAC_PIX) {
				r[0] = gx; r[1] = gy; r[2] = Ixx; r[3] = Ixy; r[4] = Iyx; r[5] = Iyy;
			} else if (variant == MTFHIP_JAC_WARPED) {
				r[0] = gx * a + gy * c; r[1] = gx * b + gy * d;
				r[2] = Ixx * a + Iyx * c; r[3] = Ixy * a + Iyy * c;
				r[4] = Ixx * b + Iyx * d; r[5] = Ixy * b + Iyy * d;
			} else {
				double inv_det = 1.0 / (a * d - b * c);
				r[0] = (gx * d - gy * c) * inv_det; r[1] = (gy * a - gx * b) * inv_det;
				r[2] = (Ixx * d - Iyx * c) * inv_det; r[3] = (Ixy * d - Iyy * c) * inv_det;
				r[4] = (Iyx * a - Ixx * b) * inv_det; r[5] = (Iyy * a - Ixy * b) * inv_det;
			}
		}
		for (int s = 0; s < S; ++s) J[(size_t)s * N + i] = r[s];
	}
}

/* ===================================================================== */
/* second-order path (sec_ord_hess): image Hessians, SSM pixel Hessians    */
/* ===================================================================== */

/* Homography::updateHessPts SSM/src/Homography.cc:829-875 (ProjectiveBase.cc:88-129) ; Affine.cc:315-350.
 * 16 doubles per pixel: (+xx, -xx, +yy, -yy, +xy, -xy, +yx, -yx) offsets of the warped point. */
__global__ __launch_bounds__(kBlock) void k_hess_pts(BatchView bv, double eps) {
	const int t = blockIdx.y;
	const Warp9 W = load_warp(bv.warps + 9 * t);
	const double2 *cp = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_CURR_PTS]) + (size_t)t * bv.NP;
	const double *cz = bv.buf[MTFHIP_BUF_CURR_Z] + (size_t)t * bv.NP;
	const double2 *ch = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_CURR_HXY]) + (size_t)t * bv.NP;
	double *hp = bv.buf[MTFHIP_BUF_HESS_PTS] + (size_t)t * bv.NP * 16;
	const double eps2 = 2 * eps;
	double dv[4][3];
#pragma unroll
	for (int r = 0; r < 3; ++r) {
		dv[0][r] = W.m[3 * r] * eps2;
		dv[1][r] = W.m[3 * r + 1] * eps2;
		dv[2][r] = (W.m[3 * r] + W.m[3 * r + 1]) * eps;
		dv[3][r] = (W.m[3 * r] - W.m[3 * r + 1]) * eps;
	}
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < bv.NP; i += gridDim.x * kBlock) {
		double2 *o = reinterpret_cast<double2 *>(hp + (size_t)i * 16);
		if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
			const double2 h = ch[i];
			const double q2 = cz[i];
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				double a0 = h.x + dv[k][0], a1 = h.y + dv[k][1], a2 = q2 + dv[k][2];
				o[2 * k] = make_double2(a0 / a2, a1 / a2);
				a0 = h.x - dv[k][0]; a1 = h.y - dv[k][1]; a2 = q2 - dv[k][2];
				o[2 * k + 1] = make_double2(a0 / a2, a1 / a2);
			}
		} else {
			const double2 p = cp[i];
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				o[2 * k] = make_double2(p.x + dv[k][0], p.y + dv[k][1]);
				o[2 * k + 1] = make_double2(p.x - dv[k][0], p.y - dv[k][1]);
			}
		}
	}
}

/* utils::getImgHess Utilities/src/imgUtils.cc:334-366 ; hess is [N][4] = (xx, xy, yx, yy) per pixel (PixHessT 4 x N) */
__global__ __launch_bounds__(kBlock) void k_img_hess(int N, ImgView im, const double *pts_all, double *hess_all,
	double eps, double pix_mult) {
	const int t = blockIdx.y;
	const double2 *pts = reinterpret_cast<const double2 *>(pts_all) + (size_t)t * N;
	double2 *hess = reinterpret_cast<double2 *>(hess_all + (size_t)t * N * 4);
	const double eps2 = 2 * eps;
	const double mult = pix_mult / (eps2 * eps2);
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		const double2 p = pts[i];
		const double c = pix_val(im, p.x, p.y);
		const double ix = pix_val(im, p.x + eps2, p.y), dx = pix_val(im, p.x - eps2, p.y);
		const double hxx = (ix + dx - 2 * c) * mult;
		const double iy = pix_val(im, p.x, p.y + eps2), dy = pix_val(im, p.x, p.y - eps2);
		const double hyy = (iy + dy - 2 * c) * mult;
		const double inc_x = p.x + eps, dec_x = p.x - eps, inc_y = p.y + eps, dec_y = p.y - eps;
		const double ixiy = pix_val(im, inc_x, inc_y), dxdy = pix_val(im, dec_x, dec_y);
		const double ixdy = pix_val(im, inc_x, dec_y), iydx = pix_val(im, dec_x, inc_y);
		const double hxy = ((ixiy + dxdy) - (ixdy + iydx)) * mult;
		hess[2 * i] = make_double2(hxx, hxy);
		hess[2 * i + 1] = make_double2(hxy, hyy);
	}
}

/* utils::getWarpedImgHess Utilities/src/imgUtils.cc:259-289 */
__global__ __launch_bounds__(kBlock) void k_warped_img_hess(int N, ImgView im, const double *pts_all, const double *hp_all,
	double *hess_all, double eps, double pix_mult) {
	const int t = blockIdx.y;
	const double2 *pts = reinterpret_cast<const double2 *>(pts_all) + (size_t)t * N;
	const double *hp = hp_all + (size_t)t * N * 16;
	double2 *hess = reinterpret_cast<double2 *>(hess_all + (size_t)t * N * 4);
	const double eps2 = 2 * eps;
	const double mult = pix_mult / (eps2 * eps2);
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		const double2 p = pts[i];
		const double2 *q = reinterpret_cast<const double2 *>(hp + (size_t)i * 16);
		const double c = pix_val(im, p.x, p.y);
		double inc = pix_val(im, q[0].x, q[0].y), dec = pix_val(im, q[1].x, q[1].y);
		const double hxx = (inc + dec - 2 * c) * mult;
		inc = pix_val(im, q[2].x, q[2].y); dec = pix_val(im, q[3].x, q[3].y);
		const double hyy = (inc + dec - 2 * c) * mult;
		inc = pix_val(im, q[4].x, q[4].y); dec = pix_val(im, q[5].x, q[5].y);
		const double inc2 = pix_val(im, q[6].x, q[6].y), dec2 = pix_val(im, q[7].x, q[7].y);
		const double hxy = ((inc + dec) - (inc2 + dec2)) * mult;
		hess[2 * i] = make_double2(hxx, hxy);
		hess[2 * i + 1] = make_double2(hxy, hyy);
	}
}

/* d2 (S x S, column-major, in registers) = dw_dp^T * M * dw_dp for the 2 x S dw_dp with rows r0, r1 */
template <int S>
__device__ __forceinline__ void sandwich(double *d2, const double *r0, const double *r1, double m00, double m01, double m10, double m11) {
	double a0[S], a1[S];
#pragma unroll
	for (int j = 0; j < S; ++j) { a0[j] = m00 * r0[j] + m01 * r1[j]; a1[j] = m10 * r0[j] + m11 * r1[j]; }
#pragma unroll
	for (int j = 0; j < S; ++j)
#pragma unroll
		for (int i = 0; i < S; ++i) d2[j * S + i] = r0[i] * a0[j] + r1[i] * a1[j];
}
/* third-order tail of Homography's Init / Warped / Approx pixel Hessians (Homography.cc:403-421, :591-613, :778-796):
 * the reference mirrors only rows 0..4 of columns 6,7 into rows 6,7 -- entries (6,5) and (7,5) keep the plain
 * sandwich value, so the block is not exactly symmetric.  Kept as is. */
__device__ __forceinline__ void hom_tail(double *d2, double Ix, double Iy, double x, double y, double sgn, double corner) {
	const double Ixx = Ix * x, Ixy = Ix * y, Iyy = Iy * y, Iyx = Iy * x;
	const double Ixxx = Ixx * x, Ixxy = Ixx * y, Ixyy = Ixy * y;
	const double Iyyy = Iyy * y, Iyyx = Iyy * x, Iyxx = Iyx * x;
#define D2(r, c) d2[(c) * 8 + (r)]
	D2(0, 6) += sgn * Ixxx; D2(0, 7) += sgn * Ixxy;
	D2(1, 6) += sgn * Ixxy; D2(1, 7) += sgn * Ixyy;
	D2(2, 6) += sgn * Ixx;  D2(2, 7) += sgn * Ixy;
	D2(3, 6) += sgn * Iyxx; D2(3, 7) += sgn * Iyyx;
	D2(4, 6) += sgn * Iyyx; D2(4, 7) += sgn * Iyyy;
	D2(5, 6) += sgn * Iyx;  D2(5, 7) += sgn * Iyy;
	D2(6, 6) += corner * (Ixxx * x + Iyxx * y);
	D2(6, 7) += corner * (Ixxy * x + Iyyx * y);
	D2(7, 6) += corner * (Ixxy * x + Iyyx * y);
	D2(7, 7) += corner * (Ixyy * x + Iyyy * y);
#pragma unroll
	for (int r = 0; r < 5; ++r) { D2(6, r) = D2(r, 6); D2(7, r) = D2(r, 7); }
#undef D2
}

/* one pixel's S x S block d2I_dp2 in registers.  Homography.cc:360-425 (init), :427-513 (pix), :515-618 (warped),
 * :696-801 (approx) ; Affine.cc:243-263 (init), :264-291 (warped).  m = (xx, xy, yx, yy). */
template <int SSM>
__device__ __forceinline__ void pix_hessian_block(double *d2, int variant, const Warp9 &W, const double *st, double x, double y,
	double cx, double cy, double D, double m0, double m1, double m2, double m3, double gx, double gy) {
	if constexpr (SSM == MTFHIP_SSM_AFFINE) {
		const double r0[6] = {1, 0, x, y, 0, 0}, r1[6] = {0, 1, 0, 0, x, y};
		if (variant == MTFHIP_JAC_INIT) { sandwich<6>(d2, r0, r1, m0, m2, m1, m3); return; }
		const double a2 = st[2] + 1, a3 = st[3], a4 = st[4], a5 = st[5] + 1;
		const double t00 = m0 * a2 + m2 * a4, t01 = m0 * a3 + m2 * a5;
		const double t10 = m1 * a2 + m3 * a4, t11 = m1 * a3 + m3 * a5;
		sandwich<6>(d2, r0, r1, a2 * t00 + a4 * t10, a2 * t01 + a4 * t11, a3 * t00 + a5 * t10, a3 * t01 + a5 * t11);
	} else {
		if (variant == MTFHIP_JAC_INIT) {
			const double r0[8] = {x, y, 1, 0, 0, 0, -x * x, -x * y}, r1[8] = {0, 0, 0, x, y, 1, -y * x, -y * y};
			sandwich<8>(d2, r0, r1, m0, m2, m1, m3);
			hom_tail(d2, gx, gy, x, y, -1.0, 2.0);
		} else if (variant == MTFHIP_JAC_PIX) {
			double r0[8] = {x, y, 1, 0, 0, 0, -cx * x, -cx * y}, r1[8] = {0, 0, 0, x, y, 1, -cy * x, -cy * y};
#pragma unroll
			for (int j = 0; j < 8; ++j) { r0[j] /= D; r1[j] /= D; }
			const double inv_d2 = 1.0 / (D * D);
			sandwich<8>(d2, r0, r1, m0, m2, m1, m3);
			const double Ixx = gx * x, Ixy = gx * y, Iyy = gy * y, Iyx = gy * x;
			const double Ixxx = Ixx * x, Ixxy = Ixx * y, Ixyy = Ixy * y;
			const double Iyyy = Iyy * y, Iyyx = Iyy * x, Iyxx = Iyx * x;
#define D2(r, c) d2[(c) * 8 + (r)]
			D2(0, 6) -= Ixxx * inv_d2; D2(1, 6) -= Ixxy * inv_d2; D2(2, 6) -= Ixx * inv_d2;
			D2(3, 6) -= Iyxx * inv_d2; D2(4, 6) -= Iyyx * inv_d2; D2(5, 6) -= Iyx * inv_d2;
			D2(6, 6) += 2 * (Ixxx * cx + Iyxx * cy) * inv_d2;
			D2(7, 6) += 2 * (Ixxy * cx + Iyyx * cy) * inv_d2;
			D2(0, 7) -= Ixxy * inv_d2; D2(1, 7) -= Ixyy * inv_d2; D2(2, 7) -= Ixy * inv_d2;
			D2(3, 7) -= Iyyx * inv_d2; D2(4, 7) -= Iyyy * inv_d2; D2(5, 7) -= Iyy * inv_d2;
			D2(6, 7) += 2 * (Ixxy * cx + Iyyx * cy) * inv_d2;
			D2(7, 7) += 2 * (Ixyy * cx + Iyyy * cy) * inv_d2;
#pragma unroll
			for (int r = 0; r < 5; ++r) { D2(6, r) = D2(r, 6); D2(7, r) = D2(r, 7); }
#undef D2
		} else if (variant == MTFHIP_JAC_WARPED) {
			const double a00 = W.m[0], a01 = W.m[1], a10 = W.m[3], a11 = W.m[4], a20 = W.m[6], a21 = W.m[7];
			const double D_inv = 1.0 / D;
			const double dwx_dx = (a00 - a20 * cx) * D_inv, dwx_dy = (a01 - a21 * cx) * D_inv;
			const double dwy_dx = (a10 - a20 * cy) * D_inv, dwy_dy = (a11 - a21 * cy) * D_inv;
			const double d2wx_dx2 = -2 * a20 * dwx_dx * D_inv, d2wx_dxdy = -(a21 * dwx_dx + a20 * dwx_dy) * D_inv;
			const double d2wx_dy2 = -2 * a21 * dwx_dy * D_inv;
			const double d2wy_dx2 = -2 * a20 * dwy_dx * D_inv, d2wy_dxdy = -(a21 * dwy_dx + a20 * dwy_dy) * D_inv;
			const double d2wy_dy2 = -2 * a21 * dwy_dy * D_inv;
			const double t00 = m0 * dwx_dx + m2 * dwy_dx, t01 = m0 * dwx_dy + m2 * dwy_dy;
			const double t10 = m1 * dwx_dx + m3 * dwy_dx, t11 = m1 * dwx_dy + m3 * dwy_dy;
			double q00 = dwx_dx * t00 + dwy_dx * t10, q01 = dwx_dx * t01 + dwy_dx * t11;
			double q10 = dwx_dy * t00 + dwy_dy * t10, q11 = dwx_dy * t01 + dwy_dy * t11;
			q00 = q00 + gx * d2wx_dx2 + gy * d2wy_dx2;
			q01 = q01 + gx * d2wx_dxdy + gy * d2wy_dxdy;
			q10 = q10 + gx * d2wx_dxdy + gy * d2wy_dxdy;
			q11 = q11 + gx * d2wx_dy2 + gy * d2wy_dy2;
			const double r0[8] = {x, y, 1, 0, 0, 0, -x * x, -x * y}, r1[8] = {0, 0, 0, x, y, 1, -y * x, -y * y};
			sandwich<8>(d2, r0, r1, q00, q01, q10, q11);
			hom_tail(d2, dwx_dx * gx + dwy_dx * gy, dwx_dy * gx + dwy_dy * gy, x, y, -1.0, 2.0);
		} else {
			const double h00 = W.m[0], h01 = W.m[1], h10 = W.m[3], h11 = W.m[4], h20 = W.m[6], h21 = W.m[7];
			const double inv_det2 = 1.0 / (D * D), inv_det = 1.0 / D;
			const double a = (h00 - h20 * cx) * inv_det, b = (h01 - h21 * cx) * inv_det;
			const double c = (h10 - h20 * cy) * inv_det, d = (h11 - h21 * cy) * inv_det;
			const double inv_factor = 1.0 / (a * d - b * c);
			const double i00 = d * inv_factor, i01 = -b * inv_factor, i10 = -c * inv_factor, i11 = a * inv_factor;
			const double ax = -h20 * (h00 + a * D - h20 * cx) * inv_det2;
			const double bx = -(h20 * h01 + h21 * (a * D - h20 * cx)) * inv_det2;
			const double cxx = -h20 * (h10 + c * D - h20 * cy) * inv_det2;
			const double dx = -(h20 * h11 + h21 * (c * D - h20 * cy)) * inv_det2;
			const double ay = -(h21 * h00 + h20 * (b * D - h21 * cx)) * inv_det2;
			const double by = -h21 * (h01 + b * D - h21 * cx) * inv_det2;
			const double cyy = -(h21 * h10 + h20 * (d * D - h21 * cy)) * inv_det2;
			const double dy = -h21 * (h11 + d * D - h21 * cy) * inv_det2;
			const double Ix = (d * gx - c * gy) * inv_factor;
			const double Iy = (a * gy - b * gx) * inv_factor;
			const double n00 = m0 - (Ix * ax + Iy * ay), n01 = m2 - (Ix * bx + Iy * by);
			const double n10 = m1 - (Ix * cxx + Iy * cyy), n11 = m3 - (Ix * dx + Iy * dy);
			const double t00 = n00 * i00 + n01 * i10, t01 = n00 * i01 + n01 * i11;
			const double t10 = n10 * i00 + n11 * i10, t11 = n10 * i01 + n11 * i11;
			const double r0[8] = {x, y, 1, 0, 0, 0, -x * x, -x * y}, r1[8] = {0, 0, 0, x, y, 1, -y * x, -y * y};
			sandwich<8>(d2, r0, r1, i00 * t00 + i10 * t10, i00 * t01 + i10 * t11, i01 * t00 + i11 * t10, i01 * t01 + i11 * t11);
			hom_tail(d2, Ix, Iy, x, y, 1.0, -1.0);
		}
	}
}

/* stand-alone SSM pixel Hessian: writes d2I_dp2 as S*S planes of N ([S*S][N], plane r + S*c = entry (r, c)) */
template <int SSM>
__global__ __launch_bounds__(kBlock) void k_pix_hessian(BatchView bv, int variant, const double *hess_all, const double *grad_all,
	double *D_all) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	const int t = blockIdx.y, N = bv.N, NP = bv.NP, C = bv.C;
	const Warp9 W = load_warp(bv.warps + 9 * t);
	const double *st = bv.states + 8 * t;
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * NP;
	const double2 *cp = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_CURR_PTS]) + (size_t)t * NP;
	const double *cz = bv.buf[MTFHIP_BUF_CURR_Z] + (size_t)t * NP;
	const double2 *ph = reinterpret_cast<const double2 *>(hess_all + (size_t)t * N * 4);
	const double *grad = grad_all + (size_t)t * N * 2;
	double *Dm = D_all + (size_t)t * N * S * S;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		const int pt = C == 1 ? i : i / C;
		const double2 p0 = ip[pt], c = cp[pt];
		const double2 ma = ph[2 * i], mb = ph[2 * i + 1];
		double d2[S * S];
		pix_hessian_block<SSM>(d2, variant, W, st, p0.x, p0.y, c.x, c.y, cz[pt], ma.x, ma.y, mb.x, mb.y, grad[i], grad[N + i]);
#pragma unroll
		for (int k = 0; k < S * S; ++k) Dm[(size_t)k * N + i] = d2[k];
	}
}

/* sum_p w[p] * d2[k][p] for the S*S planes (the second-order term of SSDBase.cc:334-342, NCC.cc:396-399, MI.cc:670-672);
 * with d2b the planes of two matrices are added first (SSDBase::cmptSumOfHessians, SSDBase.cc:405-413).
 * One partial row of S*S sums per workgroup. */
template <int S2>
__global__ __launch_bounds__(kBlock) void k_weighted_plane_sum(int N, const double *d2a_all, const double *d2b_all, const double *w_all,
	double *partials, int nblk) {
	__shared__ double lds[4 * S2];
	const int t = blockIdx.y;
	const double *da = d2a_all + (size_t)t * N * S2;
	const double *db = d2b_all ? d2b_all + (size_t)t * N * S2 : nullptr;
	const double *w = w_all + (size_t)t * N;
	double acc[S2];
#pragma unroll
	for (int k = 0; k < S2; ++k) acc[k] = 0.0;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) {
		const double wi = w[i];
		if (db) {
#pragma unroll
			for (int k = 0; k < S2; ++k) acc[k] = fma(wi, da[(size_t)k * N + i] + db[(size_t)k * N + i], acc[k]);
		} else {
#pragma unroll
			for (int k = 0; k < S2; ++k) acc[k] = fma(wi, da[(size_t)k * N + i], acc[k]);
		}
	}
	block_reduce_store<S2>(acc, partials + ((size_t)t * nblk + blockIdx.x) * S2, lds);
}
/* fixed-order sum of the per-workgroup rows of k_weighted_plane_sum: out[t][k] */
__global__ __launch_bounds__(64) void k_plane_sum_finish(const double *partials, int nblk, int S2, double *out) {
	const int t = blockIdx.x, k = threadIdx.x;
	if (k >= S2) return;
	const double *p = partials + (size_t)t * nblk * S2 + k;
	double s = 0;
	for (int b = 0; b < nblk; ++b) s += p[(size_t)b * S2];
	out[(size_t)t * S2 + k] = s;
}

/* The second-order term of the SSD Hessians for the fused path: sum_p (wt[p] * Dt[:, p] + w0[p] * D0[:, p]) in one
 * pass, the S x S pixel-Hessian blocks living in registers only (the reference materialises two S^2 x N matrices:
 * 20 MB each at 200 x 200).  Per pixel: re-sample It (residual r), the image Hessian of the current image by the 9-sample
 * stencil of getImgHess / getWarpedImgHess, its FD gradient, the current block Dt (Warped variant when chained, Init
 * otherwise, NT/ESM.cc:418-432), and the template block D0 rebuilt from the stored d2I0_dx2 / dI0_dx (6 doubles per pixel).
 *   term  0: -r * Dt                 cmptCurrHessian (2nd order), SSDBase.cc:345-375   (FCLK / ESM Std)
 *   term  1:  r * (D0 + Dt)          cmptSumOfHessians (2nd order), SSDBase.cc:377-415 (ESM SumOfStd)
 *   term  2: -r * ((D0 + Dt) / 2)    cmptCurrHessian on the mean pixel Hessian, NT/ESM.cc:324-327 (ESM Original)
 *   term  3:  r * D0                 cmptInitHessian (2nd order), SSDBase.cc:313-343   (ICLK Std)
 * d0_variant: how init_pix_hessian was produced (Warped at the identity warp by initialize, Init after setRegion). */
template <int SSM>
__global__ __launch_bounds__(kBlock) void k_second_order_ssd(BatchView bv, ImgView im, int term, int chained, int d0_variant,
	double grad_eps, double hess_eps, double norm_mult, double norm_add, double *partials, int nblk) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	__shared__ double lds[4 * S * S];
	const int t = blockIdx.y, N = bv.N;
	const Warp9 W = load_warp(bv.warps + 9 * t);
	const double *st = bv.states + 8 * t;
	Warp9 Wid;
#pragma unroll
	for (int q = 0; q < 9; ++q) Wid.m[q] = (q == 0 || q == 4 || q == 8) ? 1.0 : 0.0;
	const double st0[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * N;
	const double2 *cp = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_CURR_PTS]) + (size_t)t * N;
	const double2 *ch = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_CURR_HXY]) + (size_t)t * N;
	const double *cz = bv.buf[MTFHIP_BUF_CURR_Z] + (size_t)t * N;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N;
	const double *g0 = bv.buf[MTFHIP_BUF_DI0_DX] + (size_t)t * N * 2;
	const double2 *h0 = term != 0 ? reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_D2I0_DX2] + (size_t)t * N * 4) : nullptr;
	const double heps2 = 2 * hess_eps;
	const double hmult = norm_mult / (heps2 * heps2), gmult = norm_mult / (2 * grad_eps);
	double acc[S * S];
#pragma unroll
	for (int k = 0; k < S * S; ++k) acc[k] = 0.0;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) {
		const double2 p0 = ip[i], c = cp[i];
		const double D = cz[i];
		const double cv = pix_val(im, c.x, c.y);
		const double r = (norm_mult * cv + norm_add) - I0[i];
		double d2[S * S];
		if (term != 3) {
			double hxx, hyy, hxy, gx, gy;
			if (chained) {   /* getImgHess imgUtils.cc:334-366 + getImgGrad :233-254 at the current points */
				const double ix = pix_val(im, c.x + heps2, c.y), dx = pix_val(im, c.x - heps2, c.y);
				hxx = (ix + dx - 2 * cv) * hmult;
				const double iy = pix_val(im, c.x, c.y + heps2), dy = pix_val(im, c.x, c.y - heps2);
				hyy = (iy + dy - 2 * cv) * hmult;
				const double inc_x = c.x + hess_eps, dec_x = c.x - hess_eps, inc_y = c.y + hess_eps, dec_y = c.y - hess_eps;
				hxy = ((pix_val(im, inc_x, inc_y) + pix_val(im, dec_x, dec_y)) - (pix_val(im, inc_x, dec_y) + pix_val(im, dec_x, inc_y))) * hmult;
				gx = (pix_val(im, c.x + grad_eps, c.y) - pix_val(im, c.x - grad_eps, c.y)) * gmult;
				gy = (pix_val(im, c.x, c.y + grad_eps) - pix_val(im, c.x, c.y - grad_eps)) * gmult;
			} else {         /* updateHessPts + getWarpedImgHess imgUtils.cc:259-289 ; updateGradPts + getWarpedImgGrad :177-202 */
				double q0, q1, q2;
				if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double2 h = ch[i]; q0 = h.x; q1 = h.y; q2 = D; }
				else { q0 = c.x; q1 = c.y; q2 = 1.0; }
				auto at = [&](double o0, double o1, double o2, double sgn) -> double {
					if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
						const double a0 = q0 + sgn * o0, a1 = q1 + sgn * o1, a2 = q2 + sgn * o2;
						return pix_val(im, a0 / a2, a1 / a2);
					} else {
						return pix_val(im, q0 + sgn * o0, q1 + sgn * o1);
					}
				};
				const double xx0 = W.m[0] * heps2, xx1 = W.m[3] * heps2, xx2 = W.m[6] * heps2;
				const double yy0 = W.m[1] * heps2, yy1 = W.m[4] * heps2, yy2 = W.m[7] * heps2;
				const double xy0 = (W.m[0] + W.m[1]) * hess_eps, xy1 = (W.m[3] + W.m[4]) * hess_eps, xy2 = (W.m[6] + W.m[7]) * hess_eps;
				const double yx0 = (W.m[0] - W.m[1]) * hess_eps, yx1 = (W.m[3] - W.m[4]) * hess_eps, yx2 = (W.m[6] - W.m[7]) * hess_eps;
				hxx = (at(xx0, xx1, xx2, 1.0) + at(xx0, xx1, xx2, -1.0) - 2 * cv) * hmult;
				hyy = (at(yy0, yy1, yy2, 1.0) + at(yy0, yy1, yy2, -1.0) - 2 * cv) * hmult;
				hxy = ((at(xy0, xy1, xy2, 1.0) + at(xy0, xy1, xy2, -1.0)) - (at(yx0, yx1, yx2, 1.0) + at(yx0, yx1, yx2, -1.0))) * hmult;
				const double gx0 = W.m[0] * grad_eps, gx1 = W.m[3] * grad_eps, gx2 = W.m[6] * grad_eps;
				const double gy0 = W.m[1] * grad_eps, gy1 = W.m[4] * grad_eps, gy2 = W.m[7] * grad_eps;
				gx = (at(gx0, gx1, gx2, 1.0) - at(gx0, gx1, gx2, -1.0)) * gmult;
				gy = (at(gy0, gy1, gy2, 1.0) - at(gy0, gy1, gy2, -1.0)) * gmult;
			}
			pix_hessian_block<SSM>(d2, chained ? MTFHIP_JAC_WARPED : MTFHIP_JAC_INIT, W, st, p0.x, p0.y, c.x, c.y, D, hxx, hxy, hxy, hyy, gx, gy);
		}
		/* one block live at a time (acc + d2 + d0 together would not fit the register file):
		 * r (D0 + Dt) = r Dt + r D0 and -r (D0 + Dt) / 2 = (-r / 2) Dt + (-r / 2) D0, equal to the reference's order to round-off */
		const double wt = term == 0 ? -r : (term == 1 ? r : (term == 2 ? -r / 2.0 : 0.0));
		const double w0 = term == 1 ? r : (term == 2 ? -r / 2.0 : (term == 3 ? r : 0.0));
		if (term != 3) {
#pragma unroll
			for (int k = 0; k < S * S; ++k) acc[k] = fma(wt, d2[k], acc[k]);
		}
		if (term != 0) {
			const double2 ma = h0[2 * i], mb = h0[2 * i + 1];
			pix_hessian_block<SSM>(d2, d0_variant, Wid, st0, p0.x, p0.y, p0.x, p0.y, 1.0, ma.x, ma.y, mb.x, mb.y, g0[i], g0[N + i]);
#pragma unroll
			for (int k = 0; k < S * S; ++k) acc[k] = fma(w0, d2[k], acc[k]);
		}
	}
	block_reduce_store<S * S>(acc, partials + ((size_t)t * nblk + blockIdx.x) * (S * S), lds);
}

/* mean_pix_jacobian = (init_pix_jacobian + curr_pix_jacobian) / 2.0 (SM/src/NT/ESM.cc:239-242) */
__global__ __launch_bounds__(kBlock) void k_mean_jacobian(const double *a, const double *b, double *o, size_t n) {
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock)
		o[i] = (a[i] + b[i]) / 2.0;
}

__global__ __launch_bounds__(kBlock) void k_negate(const double *a, double *o, size_t n) {
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) o[i] = -a[i];
}

/* ===================================================================== */
/* reductions used by the un-fused AppearanceModel entry points           */
/* ===================================================================== */

/* SSDBase::updateSimilarity AM/src/SSDBase.cc:75-96: I_diff (= df_dI0 storage) = It - I0, sum r^2 */
__global__ __launch_bounds__(kBlock) void k_ssd_residual(BatchView bv, double *partials, int nblk) {
	__shared__ double lds[4 * 1];
	const int t = blockIdx.y, N = bv.N;
	const double *It = bv.buf[MTFHIP_BUF_IT] + (size_t)t * N;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N;
	double *r = bv.buf[MTFHIP_BUF_DF_DI0] + (size_t)t * N;
	double acc[1] = {0.0};
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) {
		double d = It[i] - I0[i];
		r[i] = d;
		acc[0] = fma(d, d, acc[0]);
	}
	block_reduce_store<1>(acc, partials + ((size_t)t * nblk + blockIdx.x) * ACC_COUNT + ACC_RR, lds);
}

/* df_dp = df_dI * dI_dp (AppearanceModel.h:146-153, SSDBase.cc:137,163); with sum_mode the two
 * Jacobians are added first: df_dIt * (dI0_dpssm + dIt_dpssm) (SSDBase.cc:186); otherwise a second
 * product v2 * J2 goes to ACC_G2 (AppearanceModel.h:161-164) */
__global__ __launch_bounds__(kBlock) void k_gemv(int N, int S, const double *v1_all, const double *J1_all,
	const double *v2_all, const double *J2_all, int sum_mode, double *partials, int nblk) {
	__shared__ double lds[4 * 16];
	const int t = blockIdx.y;
	const double *v1 = v1_all + (size_t)t * N, *J1 = J1_all + (size_t)t * N * S;
	const double *v2 = v2_all ? v2_all + (size_t)t * N : nullptr;
	const double *J2 = J2_all ? J2_all + (size_t)t * N * S : nullptr;
	double acc[16];
#pragma unroll
	for (int k = 0; k < 16; ++k) acc[k] = 0.0;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) {
		double a = v1[i];
		double b = v2 ? v2[i] : 0.0;
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) {
			if (s < S) {
				double j1 = J1[(size_t)s * N + i];
				if (J2 && sum_mode) {
					acc[s] = fma(a, j1 + J2[(size_t)s * N + i], acc[s]);
				} else {
					acc[s] = fma(a, j1, acc[s]);
					if (J2) acc[8 + s] = fma(b, J2[(size_t)s * N + i], acc[8 + s]);
				}
			}
		}
	}
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * ACC_COUNT;
	/* ACC_G .. ACC_G+8 and ACC_G2 .. ACC_G2+8 are not adjacent to ACC_RR: reduce into scratch then scatter */
	__shared__ double outv[16];
	block_reduce_store<16>(acc, outv, lds);
	__syncthreads();
	if (threadIdx.x < 8) dst[ACC_G + threadIdx.x] = outv[threadIdx.x];
	else if (threadIdx.x < 16) dst[ACC_G2 + threadIdx.x - 8] = outv[threadIdx.x];
}

/* d2f_dp2 = -J^T J pieces (SSDBase.cc:263,280): upper triangle of sum_i J[i,a] J[i,b] */
__global__ __launch_bounds__(kBlock) void k_gram(int N, int S, const double *J_all, double *partials, int nblk) {
	__shared__ double lds[4 * 36];
	const int t = blockIdx.y;
	const double *J = J_all + (size_t)t * N * S;
	double acc[36];
#pragma unroll
	for (int k = 0; k < 36; ++k) acc[k] = 0.0;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) {
		double r[kMaxS];
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) r[s] = s < S ? J[(size_t)s * N + i] : 0.0;
		int k = 0;
#pragma unroll
		for (int a = 0; a < kMaxS; ++a)
#pragma unroll
			for (int b = a; b < kMaxS; ++b) { acc[k] = fma(r[a], r[b], acc[k]); ++k; }
	}
	block_reduce_store<36>(acc, partials + ((size_t)t * nblk + blockIdx.x) * ACC_COUNT + ACC_H, lds);
}

/* ---------------------------------------------------------------------------------------------
 * NCC (AM/src/NCC.cc).  Per-target scalars live in `sc` ([B][8]): 0 I0_mean, 1 c, 2 It_mean, 3 b, 4 f,
 * 5 mean of the un-centred gradient vector being built.  The centred / normalised vectors the reference
 * stores (I0_cntr, It_cntr, I0_cntr_c, It_cntr_b) are recomputed from I0, It and these scalars.
 * ------------------------------------------------------------------------------------------- */
enum { NCC_I0_MEAN = 0, NCC_C = 1, NCC_IT_MEAN = 2, NCC_B = 3, NCC_F = 4, NCC_GMEAN = 5, NCC_SC = 8 };

/* sum of a vector (means: NCC.cc:76,141) -> ACC_RR */
__global__ __launch_bounds__(kBlock) void k_vec_sum(int N, const double *v_all, double *partials, int nblk) {
	__shared__ double lds[4];
	const int t = blockIdx.y;
	const double *v = v_all + (size_t)t * N;
	double acc[1] = {0.0};
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) acc[0] += v[i];
	block_reduce_store<1>(acc, partials + ((size_t)t * nblk + blockIdx.x) * ACC_COUNT + ACC_RR, lds);
}
/* a = sum I0c*Itc, b^2 = sum Itc^2, c^2 = sum I0c^2 (NCC.cc:77-78,142-144) -> ACC_G[0..2] */
__global__ __launch_bounds__(kBlock) void k_ncc_centered(BatchView bv, const double *sc_all, double *partials, int nblk) {
	__shared__ double lds[4 * 4];
	__shared__ double outv[4];
	const int t = blockIdx.y, N = bv.N;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N, *It = bv.buf[MTFHIP_BUF_IT] + (size_t)t * N;
	const double m0 = sc_all[t * NCC_SC + NCC_I0_MEAN], mt = sc_all[t * NCC_SC + NCC_IT_MEAN];
	double acc[4] = {0, 0, 0, 0};
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) {
		const double a0 = I0[i] - m0, at = It[i] - mt;
		acc[0] = fma(a0, at, acc[0]); acc[1] = fma(at, at, acc[1]); acc[2] = fma(a0, a0, acc[2]);
	}
	block_reduce_store<4>(acc, outv, lds);
	__syncthreads();
	if (threadIdx.x < 3) partials[((size_t)t * nblk + blockIdx.x) * ACC_COUNT + ACC_G + threadIdx.x] = outv[threadIdx.x];
}
/* un-centred gradient vectors of NCC::updateCurrGrad / updateInitGrad (NCC.cc:163-234) + their sum:
 * curr: (I0c/c - f*Itc/b)/b    init: (Itc/b - f*I0c/c)/c */
__global__ __launch_bounds__(kBlock) void k_ncc_grad(BatchView bv, const double *sc_all, int curr, double *out_all,
	double *partials, int nblk) {
	__shared__ double lds[4];
	const int t = blockIdx.y, N = bv.N;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N, *It = bv.buf[MTFHIP_BUF_IT] + (size_t)t * N;
	double *out = out_all + (size_t)t * N;
	const double *sc = sc_all + t * NCC_SC;
	const double m0 = sc[NCC_I0_MEAN], c = sc[NCC_C], mt = sc[NCC_IT_MEAN], b = sc[NCC_B], f = sc[NCC_F];
	double acc[1] = {0.0};
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) {
		const double i0c_c = (I0[i] - m0) / c, itc_b = (It[i] - mt) / b;
		const double v = curr ? (i0c_c - f * itc_b) / b : (itc_b - f * i0c_c) / c;
		out[i] = v;
		acc[0] += v;
	}
	block_reduce_store<1>(acc, partials + ((size_t)t * nblk + blockIdx.x) * ACC_COUNT + ACC_RR, lds);
}
/* v -= mean (df_dI = df_dI_ncntr - mean, NCC.cc:191,231) */
__global__ __launch_bounds__(kBlock) void k_sub_mean(int N, double *v_all, const double *sc_all) {
	const int t = blockIdx.y;
	double *v = v_all + (size_t)t * N;
	const double m = sc_all[t * NCC_SC + NCC_GMEAN];
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) v[i] -= m;
}
/* column sums of a pixel Jacobian (dI_dp.colwise().mean(), NCC.cc:290,322,363) -> ACC_G */
__global__ __launch_bounds__(kBlock) void k_col_sum(int N, int S, const double *J_all, double *partials, int nblk) {
	__shared__ double lds[4 * 8];
	__shared__ double outv[8];
	const int t = blockIdx.y;
	const double *J = J_all + (size_t)t * N * S;
	double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) {
#pragma unroll
		for (int s = 0; s < kMaxS; ++s)
			if (s < S) acc[s] += J[(size_t)s * N + i];
	}
	block_reduce_store<8>(acc, outv, lds);
	__syncthreads();
	if (threadIdx.x < 8) partials[((size_t)t * nblk + blockIdx.x) * ACC_COUNT + ACC_G + threadIdx.x] = outv[threadIdx.x];
}
/* NCC Hessian pieces with Jc = (J - colmean)/b (NCC.cc:290-299, 322-331, 363-382):
 * ACC_H <- sum Jc_a Jc_b,  ACC_G <- Jc^T It_cntr_b,  ACC_G2 <- Jc^T I0_cntr_c */
__global__ __launch_bounds__(kBlock) void k_ncc_hess(BatchView bv, const double *sc_all, const double *colmean_all,
	const double *J_all, double *partials, int nblk) {
	__shared__ double lds[4 * 52];
	__shared__ double outv[52];
	const int t = blockIdx.y, N = bv.N, S = bv.S;
	const double *J = J_all + (size_t)t * N * S;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N, *It = bv.buf[MTFHIP_BUF_IT] + (size_t)t * N;
	const double *sc = sc_all + t * NCC_SC, *cm = colmean_all + t * 8;
	const double m0 = sc[NCC_I0_MEAN], c = sc[NCC_C], mt = sc[NCC_IT_MEAN], b = sc[NCC_B];
	double acc[52];
#pragma unroll
	for (int k = 0; k < 52; ++k) acc[k] = 0.0;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) {
		const double i0c_c = (I0[i] - m0) / c, itc_b = (It[i] - mt) / b;
		double r[kMaxS];
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) r[s] = s < S ? (J[(size_t)s * N + i] - cm[s]) / b : 0.0;
		int k = 0;
#pragma unroll
		for (int a = 0; a < kMaxS; ++a)
#pragma unroll
			for (int b2 = a; b2 < kMaxS; ++b2) { acc[k] = fma(r[a], r[b2], acc[k]); ++k; }
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) { acc[36 + s] = fma(r[s], itc_b, acc[36 + s]); acc[44 + s] = fma(r[s], i0c_c, acc[44 + s]); }
	}
	block_reduce_store<52>(acc, outv, lds);
	__syncthreads();
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * ACC_COUNT;
	if (threadIdx.x < 44) dst[threadIdx.x] = outv[threadIdx.x];          /* ACC_H (36) + ACC_G (8) are contiguous */
	else if (threadIdx.x < 52) dst[ACC_G2 + threadIdx.x - 44] = outv[threadIdx.x];
}

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
/* Bin mode on the matrix cores.  Over a 64-pixel chunk the bin-mode sums are small dense products whose K axis is the
 * pixel: joint(r, c) = sum_p wa[r][p] wb[c][p] is (nb x 64)(64 x nb), and the joint_hist_jacobian block
 * Q[(r, c)][s] = sum_p (gd[r][p] wd[c][p]) J[p][s] is (nb^2 x 64)(64 x S).  v_mfma_f64_16x16x4_f64 takes K = 4 pixels
 * per issue; operand layout (checked on gfx950 with tools/mfma_layout_test.hip): lane l supplies A[i = l % 16][k = l / 16]
 * and B[k = l / 16][j = l % 16] and receives D[i = l / 16 + 4 v][j = l % 16] in element v of its 4-double accumulator.
 * The operands are read straight from the staged slabs (bin-major rows: lanes of one k read consecutive rows, the same
 * column -> no bank conflict beyond the 2-way of 64-bit reads).  Dense FP64 products are exact in the same sense as the
 * VALU path (fused multiply-add per k); only the summation order over pixels differs (4-pixel groups). */
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

/* histogram of A and joint histogram A x B (MI.cc:222-235 init, :245-252 init joint, :352-367 update,
 * :641-649 self).  Block partial rows: [nb hist | nb*nb joint] */
template <bool MFMA>
__global__ __launch_bounds__(kBlock) void k_mi_hist(int N, int nb, double norm_mult, const double *A_all,
	const double *B_all, double *partials, int nblk, int row_len) {
	extern __shared__ __attribute__((aligned(16))) double dyn[];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	/* slabs are bin-major with rows of kMiRow = 65 doubles: pixel-mode lanes write consecutive words, bin-mode lanes
	 * (different r, same p) land in different banks */
	double *wa = dyn + (size_t)wave * 2 * nb * kMiRow;  /* [nb][65] dense A weights of this wave's chunk */
	double *wb = wa + nb * kMiRow;                      /* [nb][65] dense B weights */
	const int t = blockIdx.y;
	const double *A = A_all + (size_t)t * N, *Bv = B_all + (size_t)t * N;
	double accj[kMiPairs], acch = 0.0;
	mfma_d4 cj = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
	for (int m = 0; m < kMiPairs; ++m) accj[m] = 0.0;
	int pr[kMiPairs], pc[kMiPairs];
#pragma unroll
	for (int m = 0; m < kMiPairs; ++m) { const int q = lane + 64 * m; pr[m] = q < nb * nb ? q / nb : -1; pc[m] = q < nb * nb ? q % nb : 0; }
	/* a wave walks only a handful of chunks and each needs its pixel values first: the next chunk's are requested
	 * before the current one is processed, otherwise every chunk starts with an exposed HBM round trip */
	int base = (blockIdx.x * (kBlock / 64) + wave) * 64;
	double a_nx = 0.0, b_nx = 0.0;
	if (base + lane < N) { a_nx = A[base + lane]; b_nx = Bv[base + lane]; }
	for (; base < N; base += nblk * kBlock) {
		const int i = base + lane;
		const double a_cur = a_nx, b_cur = b_nx;
		{
			const int in = i + nblk * kBlock;
			if (in < N) { a_nx = A[in]; b_nx = Bv[in]; }
		}
		for (int k2 = 0; k2 < nb; ++k2) { wa[k2 * kMiRow + lane] = 0.0; wb[k2 * kMiRow + lane] = 0.0; }
		if (i < N) {
			const BsplWin a = bspl_window(a_cur, nb, norm_mult, false);
			const BsplWin b = bspl_window(b_cur, nb, norm_mult, false);
			/* static indices only: a runtime-indexed window array would live in scratch memory */
#pragma unroll
			for (int r = 0; r < 4; ++r) if (r < a.n) wa[(a.lo + r) * kMiRow + lane] = a.w[r];
#pragma unroll
			for (int c = 0; c < 4; ++c) if (c < b.n) wb[(b.lo + c) * kMiRow + lane] = b.w[c];
		}
		__builtin_amdgcn_wave_barrier();
		if constexpr (MFMA) {
			/* one 16x16 tile: rows r, columns c; when nb < 16 column nb of B is all ones, so D[r][nb] is the histogram */
			const int idx = lane & 15, kq = lane >> 4, row = idx < nb ? idx : nb - 1;
#pragma unroll 4
			for (int ks = 0; ks < 16; ++ks) {
				const int p = 4 * ks + kq;
				const double av = wa[row * kMiRow + p], bv = wb[row * kMiRow + p];
				cj = __builtin_amdgcn_mfma_f64_16x16x4f64(idx < nb ? av : 0.0, idx < nb ? bv : (idx == nb ? 1.0 : 0.0), cj, 0, 0, 0);
			}
			if (nb == 16) {
#pragma unroll 8
				for (int p = 0; p < 64; ++p) if (lane < nb) acch += wa[lane * kMiRow + p];
			}
		} else {
#pragma unroll 8
			for (int p = 0; p < 64; ++p) {
#pragma unroll
				for (int m = 0; m < kMiPairs; ++m)
					if (pr[m] >= 0) accj[m] = fma(wa[pr[m] * kMiRow + p], wb[pc[m] * kMiRow + p], accj[m]);
				if (lane < nb) acch += wa[lane * kMiRow + p];
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	/* four waves -> one partial row per workgroup */
	__syncthreads();
	double *red = dyn;                                  /* [4][nb + nb*nb], the slabs are free now */
	const int rl = nb + nb * nb;
	if constexpr (MFMA) {
		const int j = lane & 15;
#pragma unroll
		for (int v = 0; v < 4; ++v) {
			const int i = (lane >> 4) + 4 * v;
			if (i < nb && j < nb) red[wave * rl + nb + i * nb + j] = cj[v];
			if (i < nb && j == nb) red[wave * rl + i] = cj[v];
		}
		if (nb == 16 && lane < nb) red[wave * rl + lane] = acch;
	} else {
		if (lane < nb) red[wave * rl + lane] = acch;
#pragma unroll
		for (int m = 0; m < kMiPairs; ++m)
			if (pr[m] >= 0) red[wave * rl + nb + pr[m] * nb + pc[m]] = accj[m];
	}
	__syncthreads();
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * row_len;
	for (int k2 = threadIdx.x; k2 < rl; k2 += kBlock) dst[k2] = (red[k2] + red[rl + k2]) + (red[2 * rl + k2] + red[3 * rl + k2]);
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
/* sums the block rows, applies pre-seeding and normalisation, logs, similarity and the gradient-factor
 * table of the requested flavour (MI.cc:237-262, 369-381, 310-314, 399-403, 427-431, 651-658).
 * mode 0: initialise (A = B = I0), 1: update (A = It, B = I0), 2: self (A = B = It) */
__global__ __launch_bounds__(kBlock) void k_mi_hist_finish(int nb, double pre_seed, double norm_mult, int mode, int first_init,
	const double *partials, int nblk, int row_len, double *tb_all, double *f_out) {
	__shared__ double red[kBlock];
	const int t = blockIdx.x;
	double *tb = tb_all + (size_t)t * MI_SIZE;
	const double *p = partials + (size_t)t * nblk * row_len;
	const double hist_seed = nb * pre_seed;
	for (int k = threadIdx.x; k < nb + nb * nb; k += kBlock) {
		const double s = column_sum(p + k, nblk, row_len);
		if (k < nb) {
			const double hv = (s + hist_seed) * norm_mult;
			if (mode == 0) { tb[MI_HIST_INIT + k] = hv; tb[MI_LOG_INIT + k] = log(hv); if (first_init) { tb[MI_HIST_CURR + k] = hv; tb[MI_LOG_CURR + k] = log(hv); } }
			else if (mode == 1) { tb[MI_HIST_CURR + k] = hv; tb[MI_LOG_CURR + k] = log(hv); }
		} else {
			const int q = k - nb, r = q / nb, c = q % nb;
			const double jv = (s + pre_seed) * norm_mult;
			if (mode == 2) tb[MI_SELF_JOINT + r * MI_NB + c] = jv;
			else if (mode == 1 || first_init) { tb[MI_JOINT + r * MI_NB + c] = jv; tb[MI_JOINT_LOG + r * MI_NB + c] = log(jv); }
		}
	}
	__syncthreads();
	double part = 0;
	for (int q = threadIdx.x; q < nb * nb; q += kBlock) {
		const int r = q / nb, c = q % nb;
		if (mode == 2) {
			const double lg = log(tb[MI_SELF_JOINT + r * MI_NB + c]);
			tb[MI_T_SELF + r * MI_NB + c] = 1 + lg - tb[MI_LOG_CURR + r];
		} else if (mode == 1 || first_init) {
			const double jv = tb[MI_JOINT + r * MI_NB + c], lg = tb[MI_JOINT_LOG + r * MI_NB + c];
			const double lr = mode == 0 ? tb[MI_LOG_INIT + r] : tb[MI_LOG_CURR + r];
			part += jv * (lg - lr - tb[MI_LOG_INIT + c]);
			if (mode == 0) {
				/* MI::initializeGrad MI.cc:310-314: both tables start as 1 + log(joint/init_hist(row)) */
				const double v = 1 + lg - tb[MI_LOG_INIT + r];
				tb[MI_T_INIT + r * MI_NB + c] = v; tb[MI_T_CURR + r * MI_NB + c] = v;
			}
		}
	}
	red[threadIdx.x] = part;
	__syncthreads();
	if (threadIdx.x == 0 && mode != 2) {
		double s = 0;
		for (int i = 0; i < kBlock; ++i) s += red[i];
		f_out[t] = s;
	}
}
/* gradient-factor tables refreshed by updateCurrGrad / updateInitGrad (MI.cc:399-403, 427-431) */
__global__ __launch_bounds__(kBlock) void k_mi_factor(int nb, int curr, double *tb_all) {
	double *tb = tb_all + (size_t)blockIdx.x * MI_SIZE;
	for (int q = threadIdx.x; q < nb * nb; q += kBlock) {
		const int r = q / nb, c = q % nb;
		if (curr) tb[MI_T_CURR + r * MI_NB + c] = 1 + tb[MI_JOINT_LOG + r * MI_NB + c] - tb[MI_LOG_CURR + r];
		else tb[MI_T_INIT + r * MI_NB + c] = 1 + tb[MI_JOINT_LOG + c * MI_NB + r] - tb[MI_LOG_INIT + r]; /* (init, curr) indexing */
	}
}
/* df_dI[p] = sum_r sum_c gradA(r,p) * matB(c,p) * T(r,c)  (MI.cc:318-326, 406-415, 432-441) */
__global__ __launch_bounds__(kBlock) void k_mi_grad(int N, int nb, double norm_mult, const double *A_all,
	const double *B_all, const double *tb_all, int table_off, double *out_all) {
	__shared__ double T[MI_NB * MI_NB];
	const int t = blockIdx.y;
	const double *tb = tb_all + (size_t)t * MI_SIZE + table_off;
	for (int k = threadIdx.x; k < MI_NB * MI_NB; k += kBlock) T[k] = tb[k];
	__syncthreads();
	const double *A = A_all + (size_t)t * N, *Bv = B_all + (size_t)t * N;
	double *out = out_all + (size_t)t * N;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		const BsplWin a = bspl_window(A[i], nb, norm_mult, false);
		const BsplWin b = bspl_window(Bv[i], nb, norm_mult, false);
		double acc = 0;
#pragma unroll
		for (int r = 0; r < 4; ++r)
#pragma unroll
			for (int c = 0; c < 4; ++c)
				if (r < a.n && c < b.n) acc += a.d[r] * b.w[c] * T[(a.lo + r) * MI_NB + b.lo + c];
		out[i] = acc;
	}
}
/* first-order MI Hessians (MI.cc:461-513 init, 565-601 self (the returned pass), 603-637 curr):
 *   Hsum  += hess_term(p) * Jrow Jrow^T,  hess_term = sum_r hessA(r) * sum_c matB(c) T(r,c)
 *   Q[row(r,c)] += gradA(r) matB(c) Jrow          row(r,c) = (r,c), or (c,r) when transpose_q (init flavour)
 * Block partial rows: [36 Hsum | nb*nb*S Q] */
template <bool MFMA>   /* MFMA: nb == 8 (the reference's 8-bin histograms): 64 (r, c) rows = four 16-row tiles */
__global__ __launch_bounds__(kBlock) void k_mi_hess(int N, int S, int nb, double norm_mult, const double *A_all,
	const double *B_all, const double *tb_all, int table_off, int transpose_q, const double *J_all,
	double *partials, int nblk, int row_len) {
	extern __shared__ __attribute__((aligned(16))) double dyn[];
	double *T = dyn;                                    /* MI_NB*MI_NB gradient-factor table */
	double *red = dyn + MI_NB * MI_NB;                  /* 4 * 36 */
	double *slabs = red + 4 * 36;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int slab = kMiRow * (2 * nb + kMaxS);
	double *gd = slabs + (size_t)wave * slab;           /* [nb][65] dense curr_hist_grad-type vector of A */
	double *wd = gd + nb * kMiRow;                      /* [nb][65] dense weights of B */
	double *rw = wd + nb * kMiRow;                      /* [kMaxS][65] J rows */
	const int t = blockIdx.y;
	const double *tb = tb_all + (size_t)t * MI_SIZE + table_off;
	for (int k2 = threadIdx.x; k2 < MI_NB * MI_NB; k2 += kBlock) T[k2] = tb[k2];
	__syncthreads();
	const double *A = A_all + (size_t)t * N, *Bv = B_all + (size_t)t * N;
	const double *J = J_all + (size_t)t * N * S;
	double acc[36];
#pragma unroll
	for (int k2 = 0; k2 < 36; ++k2) acc[k2] = 0.0;
	constexpr int NQ = MFMA ? 1 : kMiPairs;
	double accq[NQ][kMaxS];
	int pr[NQ], pc[NQ];
	mfma_d4 cq[4];
#pragma unroll
	for (int mt = 0; mt < 4; ++mt) cq[mt] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
	for (int m = 0; m < NQ; ++m) {
		const int q = lane + 64 * m;
		pr[m] = q < nb * nb ? q / nb : -1; pc[m] = q < nb * nb ? q % nb : 0;
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) accq[m][s] = 0.0;
	}
	/* operands of the next chunk (two pixel values, S Jacobian entries) are requested before the current one is processed */
	int base = (blockIdx.x * (kBlock / 64) + wave) * 64;
	double a_nx = 0.0, b_nx = 0.0, row_nx[kMaxS];
#pragma unroll
	for (int s = 0; s < kMaxS; ++s) row_nx[s] = 0.0;
	if (base + lane < N) {
		a_nx = A[base + lane]; b_nx = Bv[base + lane];
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) if (s < S) row_nx[s] = J[(size_t)s * N + base + lane];
	}
	for (; base < N; base += nblk * kBlock) {
		const int i = base + lane;
		const double a_cur = a_nx, b_cur = b_nx;
		double row[kMaxS];
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) row[s] = i < N ? row_nx[s] : 0.0;
		{
			const int in = i + nblk * kBlock;
			if (in < N) {
				a_nx = A[in]; b_nx = Bv[in];
#pragma unroll
				for (int s = 0; s < kMaxS; ++s) if (s < S) row_nx[s] = J[(size_t)s * N + in];
			}
		}
		for (int k2 = 0; k2 < nb; ++k2) { gd[k2 * kMiRow + lane] = 0.0; wd[k2 * kMiRow + lane] = 0.0; }
		if (i < N) {
			/* pixel mode: windows, the scalar hess_term and its rank-1 contribution (MI.cc:478-496, 574-583, 620-629) */
			const BsplWin a = bspl_window(a_cur, nb, norm_mult, true);
			const BsplWin b = bspl_window(b_cur, nb, norm_mult, false);
			double hess_term = 0;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				if (r < a.n) {
					double inner = 0;
#pragma unroll
					for (int c = 0; c < 4; ++c) if (c < b.n) inner += b.w[c] * T[(a.lo + r) * MI_NB + b.lo + c];
					hess_term += a.h[r] * inner;
					gd[(a.lo + r) * kMiRow + lane] = a.d[r];
				}
			}
#pragma unroll
			for (int c = 0; c < 4; ++c) if (c < b.n) wd[(b.lo + c) * kMiRow + lane] = b.w[c];
			int k2 = 0;
#pragma unroll
			for (int x = 0; x < kMaxS; ++x) {
				const double hx = hess_term * row[x];
#pragma unroll
				for (int y = x; y < kMaxS; ++y) { acc[k2] = fma(hx, row[y], acc[k2]); ++k2; }
			}
		}
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) rw[s * kMiRow + lane] = row[s];
		__builtin_amdgcn_wave_barrier();
		/* bin mode: joint_hist_jacobian.row(r, c) += grad(r, p) * mat(c, p) * J.row(p)  (MI.cc:484-486, 576-577, 622-623) */
		if constexpr (MFMA) {
			/* tile mt holds rows 16 mt .. 16 mt + 15 = (r, c) with r = 2 mt + i / 8, c = i % 8; columns s (8 of 16 used) */
			const int idx = lane & 15, kq = lane >> 4, cc = idx & 7, rh = idx >> 3;
#pragma unroll 2
			for (int ks = 0; ks < 16; ++ks) {
				const int p = 4 * ks + kq;
				const double jv = rw[cc * kMiRow + p];
				const double bj = idx < kMaxS ? jv : 0.0;
				const double wc = wd[cc * kMiRow + p];
#pragma unroll
				for (int mt = 0; mt < 4; ++mt)
					cq[mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(gd[(2 * mt + rh) * kMiRow + p] * wc, bj, cq[mt], 0, 0, 0);
			}
		} else {
#pragma unroll 4
			for (int p = 0; p < 64; ++p) {
				double jr[kMaxS];
#pragma unroll
				for (int s = 0; s < kMaxS; ++s) jr[s] = rw[s * kMiRow + p];
#pragma unroll
				for (int m = 0; m < NQ; ++m) {
					if (pr[m] >= 0) {
						const double gr = gd[pr[m] * kMiRow + p] * wd[pc[m] * kMiRow + p];
#pragma unroll
						for (int s = 0; s < kMaxS; ++s) accq[m][s] = fma(gr, jr[s], accq[m][s]);
					}
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * row_len;
	block_reduce_store<36>(acc, dst, red);
	__syncthreads();
	/* the four waves' Q blocks through the (now free) slabs: [4][nb*nb*S], indexed as the finish expects */
	double *qred = slabs;
	const int ql = nb * nb * S;
	if constexpr (MFMA) {
		const int sidx = lane & 15;
#pragma unroll
		for (int mt = 0; mt < 4; ++mt)
#pragma unroll
			for (int v = 0; v < 4; ++v) {
				const int rowq = 16 * mt + (lane >> 4) + 4 * v, r = rowq >> 3, c = rowq & 7;
				const int row_idx = transpose_q ? c * nb + r : r * nb + c;
				if (sidx < S) qred[wave * ql + row_idx * S + sidx] = cq[mt][v];
			}
	} else {
#pragma unroll
		for (int m = 0; m < NQ; ++m) {
			if (pr[m] >= 0) {
				const int row_idx = transpose_q ? pc[m] * nb + pr[m] : pr[m] * nb + pc[m];
#pragma unroll
				for (int s = 0; s < kMaxS; ++s) if (s < S) qred[wave * ql + row_idx * S + s] = accq[m][s];
			}
		}
	}
	__syncthreads();
	for (int k2 = threadIdx.x; k2 < ql; k2 += kBlock)
		dst[36 + k2] = (qred[k2] + qred[ql + k2]) + (qred[2 * ql + k2] + qred[3 * ql + k2]);
}
__global__ __launch_bounds__(kBlock) void k_mi_hess_finish(int S, int nb, const double *partials, int nblk, int row_len,
	const double *tb_all, int joint_off, int hist_off, int transpose_q, double *out) {
	extern __shared__ __attribute__((aligned(16))) double dyn[];
	double *Q = dyn;               /* nb*nb*S */
	double *Hs = dyn + nb * nb * S; /* 36 */
	const int t = blockIdx.x;
	const double *p = partials + (size_t)t * nblk * row_len;
	const double *tb = tb_all + (size_t)t * MI_SIZE;
	for (int k = threadIdx.x; k < 36 + nb * nb * S; k += kBlock) {
		const double s = column_sum(p + k, nblk, row_len);
		if (k < 36) Hs[k] = s; else Q[k - 36] = s;
	}
	__syncthreads();
	if (threadIdx.x < 64) {
		const int r2 = threadIdx.x >> 3, c2 = threadIdx.x & 7;
		if (r2 < S && c2 < S) {
			const int a = r2 < c2 ? r2 : c2, b2 = r2 < c2 ? c2 : r2;
			double h = Hs[a * 8 - (a * (a - 1)) / 2 + (b2 - a)];
			for (int rr = 0; rr < nb; ++rr)
				for (int cc = 0; cc < nb; ++cc) {
					/* Q row (rr,cc) is joint_hist_jacobian.row(linear_idx(rr,cc)); its factor uses joint(rr,cc) and the
					 * histogram of the image whose gradient was taken: rows for curr/self, columns for the init flavour */
					const double jv = tb[joint_off + rr * MI_NB + cc];
					const double hv = tb[hist_off + (transpose_q ? cc : rr)];
					const double fac = (1.0 / jv) - (1.0 / hv);
					const double *q = Q + (size_t)(rr * nb + cc) * S;
					h += q[r2] * q[c2] * fac;
				}
			out[(size_t)t * 64 + c2 * S + r2] = h;
		}
	}
}

/* the same for rows of any length (the NCC moment rows) */
__global__ __launch_bounds__(128) void k_finish_rows(const double *partials, int nblk, int row_len, double *out) {
	const int t = blockIdx.x, k = threadIdx.x;
	if (k >= row_len) return;
	out[(size_t)t * row_len + k] = column_sum(partials + (size_t)t * nblk * row_len + k, nblk, row_len);
}
/* fixed-order sum of the per-workgroup rows: out[t][k] = sum_b partials[t][b][k] */
__global__ __launch_bounds__(64) void k_finish(const double *partials, int nblk, double *out) {
	const int t = blockIdx.x, k = threadIdx.x;
	if (k >= ACC_COUNT) return;
	const double *p = partials + (size_t)t * nblk * ACC_COUNT + k;
	double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
	int b = 0;
	for (; b + 3 < nblk; b += 4) {
		s0 += p[(size_t)b * ACC_COUNT]; s1 += p[(size_t)(b + 1) * ACC_COUNT];
		s2 += p[(size_t)(b + 2) * ACC_COUNT]; s3 += p[(size_t)(b + 3) * ACC_COUNT];
	}
	for (; b < nblk; ++b) s0 += p[(size_t)b * ACC_COUNT];
	out[(size_t)t * ACC_COUNT + k] = (s0 + s1) + (s2 + s3);
}

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
};
/* warped position of a grid point and the four texels of its bilinear cell, fetched one row ahead */
struct Tex {
	double wx, wy, cx, cy, D;
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
template <bool COHERENT>
__device__ __forceinline__ void finish_track_body(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts,
	const double *partials, int nblk, int t);

/* AM = MTFHIP_AM_SSD: the residual-weighted sums above.  AM = MTFHIP_AM_NCC: the same pass accumulates the raw moments
 * NCC's similarity, Jacobians and first-order Hessians are functions of (NCC.cc:124-389 restated in ncc_from_moments,
 * mtfhip_api.hip) -- Gram(row) | sum Jt | sum It Jt | sum I0 Jt | sum It J0 | sum It, It^2, I0 It -- so an NCC iteration
 * needs no second pass over the pixels for the means; the partial rows are NCC_ACC_COUNT wide. */
template <int AM, int SSM, bool CHAINED, int MODE, bool MAT>
__device__ __forceinline__ void fused_lk_body(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk) {
	constexpr int S = (SSM == MTFHIP_SSM_HOMOGRAPHY) ? 8 : 6;
	constexpr bool NCC = AM == MTFHIP_AM_NCC;
	constexpr int K = NCC ? NCC_ACC_COUNT : 48;
	constexpr int ROW_LEN = NCC ? NCC_ACC_COUNT : ACC_COUNT;
	__shared__ double lds[4 * K];
	const int t = blockIdx.y;
	const unsigned N = (unsigned)bv.N;
	/* The per-target scalars (live flag, warp, state) sit a scalar-load round trip behind the kernel arguments and
	 * the first row's streaming operands do not depend on them: the scalar loads are requested here, but nothing
	 * waits for them (no early exit, no derived constant) until the first row's vector loads have been issued
	 * (setup_target, called from run_rows).  They must stay ahead of the asm memory fences to remain s_loads. */
	/* branch-free: without a flag array the load is pointed at this target's warp (always readable) and ignored */
	const int *live_ptr = fa.active ? fa.active + t : reinterpret_cast<const int *>(bv.warps + 9 * t);
	const int live_word = *live_ptr;
	const int live = fa.active ? live_word : 1;
	const Warp9 W = load_warp(bv.warps + 9 * t);
	const double *st = bv.states + 8 * t;
	const double st2 = st[2], st3 = st[3], st4 = st[4], st5 = st[5];
	const double2 *__restrict__ ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * N;
	const double *__restrict__ iz = bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * N;
	const double2 *__restrict__ ih = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_HXY]) + (size_t)t * N;
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
#if MTFHIP_NT_LOAD
		typedef double d2v __attribute__((ext_vector_type(2)));
		{ const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(ip) + i); in.p = make_double2(v.x, v.y); }
		in.i0 = __builtin_nontemporal_load(&I0[i]);
		if constexpr (MODE != 0) {
#pragma unroll
			for (int s = 0; s < S; ++s) in.j0[s] = __builtin_nontemporal_load(&J0[(unsigned)s * N + i]);
		} else {
			in.j0[0] = 0;
		}
#else
		const unsigned o8 = i * 8u, o16 = i * 16u;
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
		else { in.hp = ld_off<double2>(ih, o16); in.z = ld_off<double>(iz, o8); }
		return in;
	};
	/* curr_pts_hm = curr_warp * init_pts_hm and its dehomogenisation (Homography.cc:86-90, Affine.cc:104),
	 * then the texel fetch of the bilinear cell */
	auto issue_tex = [&](const PixIn<S, MODE> &in, auto uz) {
		Tex tx;
		constexpr bool UZ = decltype(uz)::value;
		const double z = UZ ? 1.0 : in.z, hx = UZ ? in.p.x : in.hp.x, hy = UZ ? in.p.y : in.hp.y;
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
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
		const unsigned to = (unsigned)(sy * istride + sx) * 4u;
#ifdef MTFHIP_EXPERIMENT_NOTEX
		tx.t00 = tx.t01 = tx.t10 = tx.t11 = (float)in.i0; (void)to;
#else
		const float *r0 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(img) + to);
		const float *r1 = reinterpret_cast<const float *>(reinterpret_cast<const char *>(img_row1) + to);
		tx.t00 = r0[0]; tx.t01 = r0[1]; tx.t10 = r1[0]; tx.t11 = r1[1];
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
		if constexpr (MODE != 2) {
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
		}
		bool fast = tcur.ok;
		if constexpr (MODE != 2 && CHAINED) {
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
		if (__builtin_amdgcn_ballot_w64(!fast) == 0) {
			const double t00 = tcur.t00, t01 = tcur.t01, t10 = tcur.t10, t11 = tcur.t11;
			it = fa.norm_mult * bilin(t00, t01, t10, t11, wx - lxd, wy - lyd) + fa.norm_add;
			if constexpr (MODE != 2) {
				double inc = bilin(t00, t01, t10, t11, px0 - lxd, py0 - lyd);
				double dec = bilin(t00, t01, t10, t11, px1 - lxd, py1 - lyd);
				gx = (inc - dec) * gmult;
				inc = bilin(t00, t01, t10, t11, px2 - lxd, py2 - lyd);
				dec = bilin(t00, t01, t10, t11, px3 - lxd, py3 - lyd);
				gy = (inc - dec) * gmult;
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
					acc[NCC_I0J + s] = fma(cur.i0, row[s], acc[NCC_I0J + s]);
				}
			}
			if constexpr (MODE != 0) {
#pragma unroll
				for (int s = 0; s < S; ++s) acc[NCC_ITJ0 + s] = fma(it, r0[s], acc[NCC_ITJ0 + s]);
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
	if (NCC || !fa.done) { block_reduce_store<K>(acc, dst, lds); return; }
	/* Last-workgroup-done epilogue.  The only data that crosses workgroups inside the launch are the partial rows and
	 * the arrival counter; both are accessed with agent-scope (sc1) atomics, which are performed at the device
	 * coherence point, so no L2 write-back / invalidate is needed (a __threadfence() per workgroup flushes the whole
	 * XCD L2 and doubled the kernel time when tried).  Order: row stores -> vmcnt(0) -> workgroup barrier -> counter. */
	block_reduce_store<K, true>(acc, dst, lds);
	__shared__ int s_last;
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(&fa.done[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblk - 1);
	__syncthreads();
	if (s_last) {
		finish_track_body<true>(bv, fa.sm, fa.ts, partials, nblk, t);
		if (threadIdx.x == 0) __hip_atomic_store(&fa.done[t], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}
template <int SSM, bool CHAINED, int MODE, bool MAT>
__global__ __launch_bounds__(kBlock, MTFHIP_FUSED_WAVES) void k_fused_ssd(BatchView bv, ImgView im, FusedArgs fa, double *partials, int nblk) {
	fused_lk_body<MTFHIP_AM_SSD, SSM, CHAINED, MODE, MAT>(bv, im, fa, partials, nblk);
}
template <int SSM, bool CHAINED, int MODE, bool MAT>
__global__ __launch_bounds__(kBlock, MTFHIP_FUSED_WAVES) void k_fused_ncc(BatchView bv, ImgView im, FusedArgs fa, double *partials, int nblk) {
	fused_lk_body<MTFHIP_AM_NCC, SSM, CHAINED, MODE, MAT>(bv, im, fa, partials, nblk);
}

/* ===================================================================== */
/* pre-processing and pyramid levels (the step upstream of the path)      */
/* ===================================================================== */
/* What the reference gets from OpenCV (Utilities/src/preprocUtils.cc:108-127 for the default CV_32FC1 output):
 *   frame_raw.convertTo(float) -> cvtColor(BGR2GRAY) when the input has 3 channels -> GaussianBlur(5x5, sigma 3)
 * and for PyramidalTracker (SM/src/PyramidalTracker.cc:88-97) cv::pyrDown (scale 0.5) or cv::resize + GaussianBlur.
 * OpenCV is a third-party dependency that is absent here; these kernels follow its published float32 algorithms
 * (operation order of the symmetric separable filter engine, BORDER_REFLECT_101, INTER_LINEAR with pixel-centre
 * alignment), restated in NumPy in oracle/preproc_ref.py.  All arithmetic is float32, no contraction. */
__device__ __forceinline__ int reflect101(int p, int n) {
	if (n == 1) return 0;
	while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
	return p;
}
/* convertTo(CV_32F) + cvtColor(BGR2GRAY): gray = B*0.114f + G*0.587f + R*0.299f (float, left to right) */
__global__ __launch_bounds__(kBlock) void k_to_gray_f32(const unsigned char *raw, int rows, int cols, size_t stride_bytes, int channels,
	int depth_f32, float *out) {
	const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
	if (x >= cols) return;
	const unsigned char *row = raw + (size_t)y * stride_bytes;
	float v;
	if (channels == 1) {
		v = depth_f32 ? reinterpret_cast<const float *>(row)[x] : (float)row[x];
	} else {
		float b, g, r;
		if (depth_f32) { const float *p = reinterpret_cast<const float *>(row) + 3 * x; b = p[0]; g = p[1]; r = p[2]; }
		else { const unsigned char *p = row + 3 * x; b = (float)p[0]; g = (float)p[1]; r = (float)p[2]; }
		v = b * 0.114f + g * 0.587f + r * 0.299f;
	}
	out[(size_t)y * cols + x] = v;
}
/* symmetric 5-tap row pass: S[0]*k0 + (S[-1]+S[1])*k1 + (S[-2]+S[2])*k2 (SymmRowSmallFilter, ksize 5) */
__global__ __launch_bounds__(kBlock) void k_sym5_rows(const float *src, int rows, int cols, float k0, float k1, float k2, float *dst) {
	const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
	if (x >= cols) return;
	const float *S = src + (size_t)y * cols;
	const float s0 = S[x], m1 = S[reflect101(x - 1, cols)], p1 = S[reflect101(x + 1, cols)];
	const float m2 = S[reflect101(x - 2, cols)], p2 = S[reflect101(x + 2, cols)];
	dst[(size_t)y * cols + x] = s0 * k0 + (m1 + p1) * k1 + (m2 + p2) * k2;
}
/* symmetric 5-tap column pass: s = k0*S0 + 0; s += k1*(S+1 + S-1); s += k2*(S+2 + S-2) (SymmColumnFilter) */
__global__ __launch_bounds__(kBlock) void k_sym5_cols(const float *src, int rows, int cols, float k0, float k1, float k2, float *dst) {
	const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
	if (x >= cols) return;
	const float c0 = src[(size_t)y * cols + x];
	const float m1 = src[(size_t)reflect101(y - 1, rows) * cols + x], p1 = src[(size_t)reflect101(y + 1, rows) * cols + x];
	const float m2 = src[(size_t)reflect101(y - 2, rows) * cols + x], p2 = src[(size_t)reflect101(y + 2, rows) * cols + x];
	float s = k0 * c0 + 0.0f;
	s += k1 * (p1 + m1);
	s += k2 * (p2 + m2);
	dst[(size_t)y * cols + x] = s;
}
/* cv::pyrDown, float: rows  r[x] = S[2x]*6 + (S[2x-1]+S[2x+1])*4 + S[2x-2] + S[2x+2] ; columns the same on the five row buffers,
 * times 1/256 */
__global__ __launch_bounds__(kBlock) void k_pyr_down(const float *src, int srows, int scols, int drows, int dcols, float *dst) {
	const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
	if (x >= dcols) return;
	float r[5];
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		const float *S = src + (size_t)reflect101(2 * y - 2 + j, srows) * scols;
		const float c = S[reflect101(2 * x, scols)], m1 = S[reflect101(2 * x - 1, scols)], p1 = S[reflect101(2 * x + 1, scols)];
		const float m2 = S[reflect101(2 * x - 2, scols)], p2 = S[reflect101(2 * x + 2, scols)];
		r[j] = c * 6.0f + (m1 + p1) * 4.0f + m2 + p2;
	}
	dst[(size_t)y * dcols + x] = (r[2] * 6.0f + (r[1] + r[3]) * 4.0f + r[0] + r[4]) * (1.0f / 256.0f);
}
/* cv::resize INTER_LINEAR, float: fx = (float)((dx + 0.5) * scale - 0.5), clamped like resizeGeneric's index tables */
__device__ __forceinline__ void lin_coord(int d, double scale, int n, int &s, float &f) {
	f = (float)(((double)d + 0.5) * scale - 0.5);
	s = (int)floorf(f);
	f -= (float)s;
	if (s < 0) { f = 0.0f; s = 0; }
	if (s >= n - 1) { f = 0.0f; s = n - 1; }
}
__global__ __launch_bounds__(kBlock) void k_resize_linear(const float *src, int srows, int scols, int drows, int dcols, float *dst) {
	const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
	if (x >= dcols) return;
	int sx, sy; float fx, fy;
	lin_coord(x, (double)scols / dcols, scols, sx, fx);
	lin_coord(y, (double)srows / drows, srows, sy, fy);
	const int sx1 = sx + 1 < scols ? sx + 1 : sx, sy1 = sy + 1 < srows ? sy + 1 : sy;
	const float *S0 = src + (size_t)sy * scols, *S1 = src + (size_t)sy1 * scols;
	const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
	const float h0 = sx >= scols - 1 ? S0[sx] * 1.0f : S0[sx] * a0 + S0[sx1] * a1;
	const float h1 = sx >= scols - 1 ? S1[sx] * 1.0f : S1[sx] * a0 + S1[sx1] * a1;
	dst[(size_t)y * dcols + x] = h0 * b0 + h1 * b1;
}

/* ===================================================================== */
/* candidate scoring (PF / NN batch axis)                                 */
/* ===================================================================== */
/* One wave64 per candidate: setState -> updatePixVals -> updateSimilarity -> likelihood
 * (SM/src/PF.cc:247-262, ProjectiveBase.cc:41-49, SSDBase.cc:75-96, SSD.h:41-43). */
__global__ __launch_bounds__(kBlock) void k_score_candidates(BatchView bv, ImgView im, const double *states, int C,
	double alpha, double norm_mult, double norm_add, double *lik, double *sim) {
	const int lane = threadIdx.x & 63;
	const int cand = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
	if (cand >= C) return;
	const int N = bv.N, S = bv.S;
	const double *p = states + (size_t)cand * S;
	double W[9];
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
		W[0] = 1 + p[0]; W[1] = p[1]; W[2] = p[2]; W[3] = p[3]; W[4] = 1 + p[4]; W[5] = p[5];
		W[6] = p[6]; W[7] = p[7]; W[8] = 1;
	} else {
		W[0] = 1 + p[2]; W[1] = p[3]; W[2] = p[0]; W[3] = p[4]; W[4] = 1 + p[5]; W[5] = p[1];
		W[6] = 0; W[7] = 0; W[8] = 1;
	}
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]);
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z];
	const double2 *ih = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_HXY]);
	const double *I0 = bv.buf[MTFHIP_BUF_I0];
	double acc = 0.0;
	for (int i = lane; i < N; i += 64) {
		double2 q = bv.unit_z ? ip[i] : ih[i];
		double z = bv.unit_z ? 1.0 : iz[i];
		double hx = q.x, hy = q.y;
		double wx, wy;
		if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
			double cx = W[0] * hx + W[1] * hy + W[2] * z;
			double cy = W[3] * hx + W[4] * hy + W[5] * z;
			double d = W[6] * hx + W[7] * hy + W[8] * z;
			wx = cx / d; wy = cy / d;
		} else {
			wx = W[0] * hx + W[1] * hy + W[2] * z;
			wy = W[3] * hx + W[4] * hy + W[5] * z;
		}
		double r = (norm_mult * pix_val(im, wx, wy) + norm_add) - I0[i];
		acc = fma(r, r, acc);
	}
#pragma unroll
	for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
	if (lane == 0) {
		double f = -acc / 2;
		if (sim) sim[cand] = f;
		if (lik) lik[cand] = exp(-alpha * sqrt(-f / (double)N));
	}
}

/* LDS-staged candidate scoring.  All candidates of a frame sample the same template through slightly
 * different warps, so a workgroup stages, once, (a) the template's homogeneous grid points and I0 and (b) the
 * image tile that covers the template's bounding box plus a margin, then its 16 waves score CPW candidates
 * each entirely out of LDS: ds_read_b128 / _b64 for the template, two ds_read2_b32 for the four texels.
 * A sample whose bilinear cell is not inside the tile (a far-out candidate) takes the global-memory path;
 * either way the reference's sampling expression is evaluated unchanged (imgUtils.h:91-113). */
constexpr int kScoreBlock = 1024;
constexpr int kScoreSplit = 4;      /* a candidate's pixels are cut into this many work units (load balance) */
template <int SSM>
__global__ __launch_bounds__(kScoreBlock) void k_score_candidates_lds(BatchView bv, ImgView im, const double *states, int C,
	int tx0, int ty0, int tw, int th, double norm_mult, double norm_add, double *unit_sums /* [C][kScoreSplit] */) {
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int N = bv.N;
	double2 *sp = reinterpret_cast<double2 *>(smem);                 /* N grid points (x, y or X, Y) */
	double *sz = reinterpret_cast<double *>(sp + N);                  /* N third homogeneous coordinates */
	double *s0 = sz + N;                                              /* N template values */
	float *tile = reinterpret_cast<float *>(s0 + N);                  /* th x tw texels */
	const double2 *gp = reinterpret_cast<const double2 *>(bv.buf[bv.unit_z ? MTFHIP_BUF_INIT_PTS : MTFHIP_BUF_INIT_HXY]);
	const double *gz = bv.buf[MTFHIP_BUF_INIT_Z], *g0 = bv.buf[MTFHIP_BUF_I0];
	for (int i = threadIdx.x; i < N; i += kScoreBlock) { sp[i] = gp[i]; sz[i] = bv.unit_z ? 1.0 : gz[i]; s0[i] = g0[i]; }
	for (int i = threadIdx.x; i < tw * th; i += kScoreBlock) {
		const int yy = ty0 + i / tw, xx = tx0 + i % tw;
		tile[i] = (yy >= 0 && yy < im.h && xx >= 0 && xx < im.w) ? im.data[(size_t)yy * im.stride + xx] : 0.0f;
	}
	__syncthreads();
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int n_waves = gridDim.x * (kScoreBlock / 64);
	const double w = (double)(unsigned)im.w, h = (double)(unsigned)im.h;
	const int chunk = ((N + kScoreSplit - 1) / kScoreSplit + 63) / 64 * 64;   /* pixels per work unit, multiple of 64 */
	/* work unit u = (candidate, pixel chunk); waves take units round-robin over the whole launch */
	for (int u = blockIdx.x * (kScoreBlock / 64) + wave; u < C * kScoreSplit; u += n_waves) {
		const int cand = u / kScoreSplit, part = u % kScoreSplit;
		const double *p = states + (size_t)cand * bv.S;
		double W[9];
		if (SSM == MTFHIP_SSM_HOMOGRAPHY) {
			W[0] = 1 + p[0]; W[1] = p[1]; W[2] = p[2]; W[3] = p[3]; W[4] = 1 + p[4]; W[5] = p[5]; W[6] = p[6]; W[7] = p[7]; W[8] = 1;
		} else {
			W[0] = 1 + p[2]; W[1] = p[3]; W[2] = p[0]; W[3] = p[4]; W[4] = 1 + p[5]; W[5] = p[1]; W[6] = 0; W[7] = 0; W[8] = 1;
		}
		double acc = 0.0;
		const int i_end = min(N, (part + 1) * chunk);
		for (int i = part * chunk + lane; i < i_end; i += 64) {
			const double2 hp = sp[i];
			const double z = sz[i];
			double x, y;
			if (SSM == MTFHIP_SSM_HOMOGRAPHY) {
				const double cx = W[0] * hp.x + W[1] * hp.y + W[2] * z, cy = W[3] * hp.x + W[4] * hp.y + W[5] * z;
				const double d = W[6] * hp.x + W[7] * hp.y + W[8] * z;
				x = cx / d; y = cy / d;
			} else {
				x = W[0] * hp.x + W[1] * hp.y + W[2] * z; y = W[3] * hp.x + W[4] * hp.y + W[5] * z;
			}
			double v = 128.0;
			if (!((x < 0) || (x >= w) || (y < 0) || (y >= h))) {
				const int lx = (int)x, ly = (int)y;
				const double dx = x - lx, dy = y - ly;
				const int ux = dx == 0 ? lx : lx + 1, uy = dy == 0 ? ly : ly + 1;
				if (ux < im.w && uy < im.h) {
					double t00, t01, t10, t11;
					if (lx >= tx0 && ux < tx0 + tw && ly >= ty0 && uy < ty0 + th) {
						const float *r0 = tile + (ly - ty0) * tw + (lx - tx0), *r1 = tile + (uy - ty0) * tw + (lx - tx0);
						t00 = r0[0]; t01 = r0[ux - lx]; t10 = r1[0]; t11 = r1[ux - lx];
					} else {
						const float *r0 = im.data + (size_t)ly * im.stride, *r1 = im.data + (size_t)uy * im.stride;
						t00 = r0[lx]; t01 = r0[ux]; t10 = r1[lx]; t11 = r1[ux];
					}
					v = t00 * (1 - dx) * (1 - dy) + t01 * dx * (1 - dy) + t10 * (1 - dx) * dy + t11 * dx * dy;
				}
			}
			const double r = (norm_mult * v + norm_add) - s0[i];
			acc = fma(r, r, acc);
		}
#pragma unroll
		for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
		if (lane == 0) unit_sums[u] = acc;
	}
}
/* fixed-order sum of a candidate's work units, then f = -|r|^2/2 and the SSD likelihood (SSD.h:41-43) */
__global__ __launch_bounds__(kBlock) void k_score_finish(const double *unit_sums, int C, int N, double alpha, double *lik, double *sim) {
	const int c = blockIdx.x * kBlock + threadIdx.x;
	if (c >= C) return;
	double s = 0;
#pragma unroll
	for (int q = 0; q < kScoreSplit; ++q) s += unit_sums[(size_t)c * kScoreSplit + q];
	const double f = -s / 2;
	if (sim) sim[c] = f;
	if (lik) lik[c] = exp(-alpha * sqrt(-f / (double)N));
}

/* ===================================================================== */
/* on-device solve + compositional update (batched drivers only)          */
/* ===================================================================== */
/* One wave64 per target, one launch per LK iteration:
 *   (1) fixed-order sum of the per-workgroup partial rows (what k_finish does for the host-driven path),
 *   (2) g and H of the search method from the accumulators (NT/FCLK.cc:260-288, NT/ESM.cc:298-377 with
 *       SSDBase.cc:169-191,287-311, NT/ICLK.cc:206-251),
 *   (3) H dp = -g by Gauss-Jordan elimination spread over the 64 lanes (lane = matrix entry) on the
 *       symmetrically diagonal-scaled system; every SSD Hessian here is a negated Gram matrix, i.e.
 *       definite, so no pivoting is needed (the reference uses Eigen's colPivHouseholderQr, NT/FCLK.cc:298),
 *   (4) the (inverse) compositional update and the corner-change test on lane 0
 *       (Homography.cc:73-92,109-114, Affine.cc:90-106,145-150, NT/FCLK.cc:314-339). */
/* Body of the device-side finish, executed by the first wave of the calling workgroup (all threads of the workgroup
 * must call it: it contains workgroup barriers).  COHERENT: the partial rows were written by OTHER workgroups of the
 * same launch (last-workgroup-done epilogue of k_fused_ssd), possibly on another XCD whose L2 is not coherent with
 * ours, so they are read with agent-scope atomic loads instead of plain ones. */
template <bool COHERENT>
__device__ __forceinline__ void finish_track_body(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts,
	const double *partials, int nblk, int t) {
	__shared__ double acc_s[NCC_ACC_COUNT];   /* >= ACC_COUNT */
	__shared__ double A[8][9];
	__shared__ double dps[8];
	__shared__ double h0s[64], Ws[9], crs[8], ics[12], tms[52], ncs[2];
	const int lane = threadIdx.x;
	const bool wv0 = lane < 64;
	const int S = bv.S;
	/* NCC: the reduced row holds raw moments (NCC_* slots, NCC_ACC_COUNT wide); tms = sum J0 | sum I0 J0 | Gram(J0) of the
	 * template, ncs = mean(I0), |I0 - mean|.  The calling workgroup then has at least 128 threads. */
	const bool ncc = bv.am == MTFHIP_AM_NCC;
	const int RL = ncc ? (int)NCC_ACC_COUNT : (int)ACC_COUNT;
	/* every global operand of this target -- the `active` flag included -- is requested up front, in parallel across
	 * the lanes, and only then is the flag tested: one memory round trip instead of two (flag, then operands); the
	 * rest of the routine runs out of LDS / registers */
	const int act = ts.active[t];
	int n_it_prev = 0;
	double v_h0 = 0, v_w = 0, v_cr = 0, v_ic = 0, v_acc = 0, v_tm = 0, v_nc = 0;
	if (wv0) {
		v_h0 = ts.h0[(size_t)t * 64 + lane];
		if (lane < 9) v_w = bv.warps[9 * t + lane];
		if (lane < 8) v_cr = ts.corners[8 * t + lane];
		if (lane < 12) v_ic = ts.init_corners_hm[12 * t + lane];
		n_it_prev = ts.n_iters[t];
		if (ncc) {
			if (lane < 52) v_tm = ts.ncc_tm[(size_t)t * 52 + lane];
			if (lane < 2) v_nc = ts.ncc[(size_t)t * 8 + lane];
		}
	}
	if (lane < RL) {
		const double *p = partials + (size_t)t * nblk * RL + lane;
		auto ld = [&](size_t off) -> double {
			if constexpr (COHERENT) return __hip_atomic_load(p + off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			else return p[off];
		};
		double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
		int b = 0;
		for (; b + 3 < nblk; b += 4) {
			s0 += ld((size_t)b * RL); s1 += ld((size_t)(b + 1) * RL);
			s2 += ld((size_t)(b + 2) * RL); s3 += ld((size_t)(b + 3) * RL);
		}
		for (; b < nblk; ++b) s0 += ld((size_t)b * RL);
		v_acc = (s0 + s1) + (s2 + s3);
	}
	if (!act) return;
	if (wv0) {
		h0s[lane] = v_h0;
		if (lane < 9) Ws[lane] = v_w;
		if (lane < 8) crs[lane] = v_cr;
		if (lane < 12) ics[lane] = v_ic;
		if (lane < 52) tms[lane] = v_tm;
		if (lane < 2) ncs[lane] = v_nc;
	}
	if (lane < RL) { acc_s[lane] = v_acc; ts.acc[(size_t)t * RL + lane] = v_acc; }
	__syncthreads();
	const int i = (lane >> 3) & 7, j = lane & 7;
	const bool use_h0 = (sm.hess_type == 0) || (sm.sm == MTFHIP_SM_ICLK);
	const bool sum_h0 = (sm.sm == MTFHIP_SM_ESM) && (sm.hess_type == 2 || sm.hess_type == 4);
	const double gscale = (sm.sm == MTFHIP_SM_ESM) ? 0.5 : 1.0;
	/* NCC from its moments (ncc_assemble in mtfhip_api.hip is the host twin; formulas and citations there) */
	const double nN = (double)bv.N;
	const double n_mt = ncc ? acc_s[NCC_IT] / nN : 0.0, n_m0 = ncs[0], n_c = ncc ? ncs[1] : 1.0;
	const double n_b2 = ncc ? acc_s[NCC_IT2] - nN * n_mt * n_mt : 1.0, n_b = ncc ? sqrt(n_b2) : 1.0;
	const double n_f = ncc ? (acc_s[NCC_I0IT] - nN * n_m0 * n_mt) / (n_b * n_c) : 0.0;
	auto mom = [&](int which, int c0, int ct, int s) -> double {   /* which: 0 J0, 1 Jt, 2 their mean */
		const double v0 = c0 >= 0 ? tms[c0 + s] : acc_s[NCC_ITJ0 + s], vt = acc_s[ct + s];
		return which == 0 ? v0 : (which == 1 ? vt : (v0 + vt) / 2);
	};
	auto n_ut = [&](int which, int s) { return (mom(which, -1, NCC_ITJ, s) - n_mt * mom(which, 0, NCC_SJ, s)) / n_b2; };
	auto n_u0 = [&](int which, int s) { return (mom(which, 8, NCC_I0J, s) - n_m0 * mom(which, 0, NCC_SJ, s)) / (n_b * n_c); };
	auto n_hess = [&](int kind, int which, int r, int c, int kk) -> double {   /* kind: 0 init, 1 curr, 2 self */
		const double gram = which == 0 ? tms[16 + kk] : acc_s[NCC_GRAM + kk];
		const double G = -(gram - mom(which, 0, NCC_SJ, r) * mom(which, 0, NCC_SJ, c) / nN) / n_b2;
		const double utr = n_ut(which, r), utc = n_ut(which, c);
		if (kind == 2) return G + utr * utc;
		const double u0r = n_u0(which, r), u0c = n_u0(which, c);
		return n_f * G - utr * u0c - u0r * utc + 3 * (kind == 1 ? utr * utc : u0r * u0c);
	};
	auto h_entry = [&](int r, int c) -> double {
		if (r >= S || c >= S) return r == c ? -1.0 : 0.0;
		const int a = r < c ? r : c, b2 = r < c ? c : r;
		const int kk = a * 8 - (a * (a - 1)) / 2 + (b2 - a);
		if (ncc) {
			const int ht = sm.hess_type;
			const double h0v = h0s[b2 * S + a];
			if (ht == 0) return h0v;
			if (sm.sm == MTFHIP_SM_ICLK) return n_hess(0, 0, r, c, kk);
			if (sm.sm == MTFHIP_SM_FCLK || ht == 1 || ht == 5) return n_hess(ht == 1 ? 2 : 1, 1, r, c, kk);
			if (ht == 2) return 0.5 * (n_hess(2, 1, r, c, kk) + h0v);
			if (ht == 3) return n_hess(1, 2, r, c, kk);
			return 0.5 * (n_hess(0, 0, r, c, kk) + n_hess(1, 1, r, c, kk));
		}
		double v = use_h0 ? h0s[b2 * S + a] : -acc_s[ACC_H + kk];
		if (sum_h0) v = (v + h0s[b2 * S + a]) * 0.5;
		return v;
	};
	auto g_entry = [&](int s) -> double {
		if (!ncc) return gscale * acc_s[ACC_G + s];
		auto cj = [&](int which) { return n_u0(which, s) - n_f * n_ut(which, s); };
		auto ij = [&](int which) { return (n_b / n_c) * (n_ut(which, s) - n_f * n_u0(which, s)); };
		if (sm.sm == MTFHIP_SM_FCLK) return cj(1);
		if (sm.sm == MTFHIP_SM_ICLK) return ij(0);
		if (sm.jac_type == 0) return cj(2);
		return 0.5 * (cj(1) - ij(0));
	};
	const double dii = h_entry(i, i), djj = h_entry(j, j);
	const double si = dii != 0 ? 1.0 / sqrt(fabs(dii)) : 1.0, sj = djj != 0 ? 1.0 / sqrt(fabs(djj)) : 1.0;
	if (wv0) {
		A[i][j] = h_entry(i, j) * si * sj;
		if (j == 0) A[i][8] = (i < S ? g_entry(i) : 0.0) * si;
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		const double piv = A[k][k], aik = A[i][k], akj = A[k][j], bk = A[k][8];
		const double f = (i != k && piv != 0) ? aik / piv : 0.0;
		__syncthreads();
		if (wv0 && i != k) {
			A[i][j] -= f * akj;
			if (j == 0) A[i][8] -= f * bk;
		}
		__syncthreads();
	}
	if (wv0 && j == 0) {
		const double d = A[i][i];
		dps[i] = (i < S && d != 0) ? -(A[i][8] / d) * si : 0.0;
	}
	__syncthreads();
	if (lane != 0) return;

	double dp[8];
#pragma unroll
	for (int s = 0; s < 8; ++s) dp[s] = dps[s];
	double *Wp = bv.warps + 9 * t, *st = bv.states + 8 * t;
	double U[9];
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
		U[0] = 1 + dp[0]; U[1] = dp[1]; U[2] = dp[2]; U[3] = dp[3]; U[4] = 1 + dp[4]; U[5] = dp[5];
		U[6] = dp[6]; U[7] = dp[7]; U[8] = 1;
	} else {
		U[0] = 1 + dp[2]; U[1] = dp[3]; U[2] = dp[0]; U[3] = dp[4]; U[4] = 1 + dp[5]; U[5] = dp[1];
		U[6] = 0; U[7] = 0; U[8] = 1;
	}
	if (sm.sm == MTFHIP_SM_ICLK) {
		/* invertState: inverse through cofactors, normalised by (2,2) */
		double c[9];
		c[0] = U[4] * U[8] - U[5] * U[7]; c[1] = U[2] * U[7] - U[1] * U[8]; c[2] = U[1] * U[5] - U[2] * U[4];
		c[3] = U[5] * U[6] - U[3] * U[8]; c[4] = U[0] * U[8] - U[2] * U[6]; c[5] = U[2] * U[3] - U[0] * U[5];
		c[6] = U[3] * U[7] - U[4] * U[6]; c[7] = U[1] * U[6] - U[0] * U[7]; c[8] = U[0] * U[4] - U[1] * U[3];
		double det = U[0] * c[0] + U[1] * c[3] + U[2] * c[6];
		double inv_det = 1.0 / det;
#pragma unroll
		for (int q = 0; q < 9; ++q) c[q] *= inv_det;
		double n22 = c[8];
#pragma unroll
		for (int q = 0; q < 9; ++q) U[q] = c[q] / n22;
		/* round-trip through the state parameterisation as getStateFromWarp / getWarpFromState do */
		U[0] = 1 + (U[0] - 1); U[4] = 1 + (U[4] - 1); U[8] = 1;
		if (bv.ssm != MTFHIP_SSM_HOMOGRAPHY) { U[6] = 0; U[7] = 0; }
	}
	double Wo[9], Wn[9];
#pragma unroll
	for (int q = 0; q < 9; ++q) Wo[q] = Ws[q];
#pragma unroll
	for (int r = 0; r < 3; ++r)
#pragma unroll
		for (int c2 = 0; c2 < 3; ++c2)
			Wn[3 * r + c2] = Wo[3 * r] * U[c2] + Wo[3 * r + 1] * U[3 + c2] + Wo[3 * r + 2] * U[6 + c2];
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
		double n22 = Wn[8];
#pragma unroll
		for (int q = 0; q < 9; ++q) Wn[q] /= n22;
		st[0] = Wn[0] - 1; st[1] = Wn[1]; st[2] = Wn[2]; st[3] = Wn[3]; st[4] = Wn[4] - 1; st[5] = Wn[5];
		st[6] = Wn[6]; st[7] = Wn[7];
	} else {
		st[0] = Wn[2]; st[1] = Wn[5]; st[2] = Wn[0] - 1; st[3] = Wn[1]; st[4] = Wn[3]; st[5] = Wn[4] - 1;
	}
#pragma unroll
	for (int q = 0; q < 9; ++q) Wp[q] = Wn[q];
	double *cr = ts.corners + 8 * t;
	double change = 0;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		double X = ics[3 * q], Y = ics[3 * q + 1], Z = ics[3 * q + 2];
		double nx = Wn[0] * X + Wn[1] * Y + Wn[2] * Z, ny = Wn[3] * X + Wn[4] * Y + Wn[5] * Z;
		if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
			double d = Wn[6] * X + Wn[7] * Y + Wn[8] * Z;
			nx = nx / d; ny = ny / d;
		}
		double ddx = crs[2 * q] - nx, ddy = crs[2 * q + 1] - ny;
		change += ddx * ddx + ddy * ddy;
		cr[2 * q] = nx; cr[2 * q + 1] = ny;
	}
	const int n_it = n_it_prev + 1;
	ts.n_iters[t] = n_it;
	if (change < sm.epsilon || n_it >= sm.max_iters) ts.active[t] = 0;
}
/* stand-alone finish: one wave per target */
__global__ __launch_bounds__(128) void k_finish_track(BatchView bv, mtfhip_sm_desc sm, TrackState ts,
	const double *partials, int nblk) {
	finish_track_body<false>(bv, sm, ts, partials, nblk, blockIdx.x);
}

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

/* NN-SM dataset generation (SM/src/NT/NN.cc:131-191): per sample state, setState -> updatePixVals ->
 * updateDistFeat into row `c` of the n_samples x N feature matrix.  SSD's feature is the patch itself
 * (AM/include/mtf/AM/SSDBase.h:116-125); NCC's is the centred patch over its norm (AM/src/NCC.cc:530-537),
 * applied by k_ncc_feature_rows afterwards.  One workgroup per sample. */
__global__ __launch_bounds__(kBlock) void k_sample_candidates(BatchView bv, ImgView im, const double *states, int C,
	double norm_mult, double norm_add, double *feat) {
	const int cand = blockIdx.x;
	const int N = bv.N, S = bv.S;
	const double *p = states + (size_t)cand * S;
	double W[9];
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
		W[0] = 1 + p[0]; W[1] = p[1]; W[2] = p[2]; W[3] = p[3]; W[4] = 1 + p[4]; W[5] = p[5]; W[6] = p[6]; W[7] = p[7]; W[8] = 1;
	} else {
		W[0] = 1 + p[2]; W[1] = p[3]; W[2] = p[0]; W[3] = p[4]; W[4] = 1 + p[5]; W[5] = p[1]; W[6] = 0; W[7] = 0; W[8] = 1;
	}
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[bv.unit_z ? MTFHIP_BUF_INIT_PTS : MTFHIP_BUF_INIT_HXY]);
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z];
	double *out = feat + (size_t)cand * N;
	for (int i = threadIdx.x; i < N; i += kBlock) {
		const double2 q = ip[i];
		const double z = bv.unit_z ? 1.0 : iz[i];
		double wx, wy;
		if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) {
			const double cx = W[0] * q.x + W[1] * q.y + W[2] * z, cy = W[3] * q.x + W[4] * q.y + W[5] * z;
			const double d = W[6] * q.x + W[7] * q.y + W[8] * z;
			wx = cx / d; wy = cy / d;
		} else {
			wx = W[0] * q.x + W[1] * q.y + W[2] * z; wy = W[3] * q.x + W[4] * q.y + W[5] * z;
		}
		out[i] = norm_mult * pix_val(im, wx, wy) + norm_add;
	}
}
/* NCC::updateDistFeat NCC.cc:530-537: row <- (row - mean) / ||row - mean|| */
__global__ __launch_bounds__(kBlock) void k_ncc_feature_rows(int N, double *feat) {
	__shared__ double red[4];
	double *row = feat + (size_t)blockIdx.x * N;
	double s[1] = {0.0};
	for (int i = threadIdx.x; i < N; i += kBlock) s[0] += row[i];
	block_allsum<1>(s, red);
	const double mean = s[0] / (double)N;
	double q[1] = {0.0};
	for (int i = threadIdx.x; i < N; i += kBlock) { const double d = row[i] - mean; q[0] = fma(d, d, q[0]); }
	block_allsum<1>(q, red);
	const double sd = sqrt(q[0]);
	for (int i = threadIdx.x; i < N; i += kBlock) row[i] = (row[i] - mean) / sd;
}

/*
 * nt::ICLK::update (SM/src/NT/ICLK.cc:160-299) for one patch per workgroup, all iterations inside the
 * kernel: updatePixVals -> updateSimilarity -> updateInitGrad -> cmptInitJacobian(g, J0) ->
 * dp = -H0^-1 g (hess_type InitialSelf: the Hessian is the constant computed by initialize) ->
 * invertState -> compositionalUpdate -> corner-change test.  AM = SSD (SSDBase.cc:75-96,138) or NCC
 * (NCC.cc:124-194, 236-250).  This is what GridTracker's per-patch loop (SM/src/GridTracker.cc:247-261)
 * becomes: 256 patches = 256 workgroups, one launch per frame, no host round trips.
 * Patch operands (grid points, I0, J0: ~35 KB for 25x25 affine) are re-read from L2 every iteration.
 */
/* getPixVal<Linear, Constant> without control flow: the four texel loads are always issued (from clamped, valid
 * addresses) and the border value is selected afterwards, so several independent samples of one thread can be in
 * flight together.  Same expression and operation order as pix_val() for every in-range sample. */
__device__ __forceinline__ double pix_val_select(const ImgView &im, double x, double y) {
	const double w = (double)(unsigned int)im.w, h = (double)(unsigned int)im.h;
	const bool in0 = !((x < 0) || (x >= w) || (y < 0) || (y >= h));
	const double xs = in0 ? x : 0.0, ys = in0 ? y : 0.0;
	const int lx = (int)xs, ly = (int)ys;
	const double dx = xs - lx, dy = ys - ly;
	const int ux = dx == 0 ? lx : lx + 1, uy = dy == 0 ? ly : ly + 1;
	const bool in1 = !(ux >= im.w || uy >= im.h);
	const int uxc = in1 ? ux : lx, uyc = in1 ? uy : ly;
	const float *r0 = im.data + (size_t)ly * im.stride, *r1 = im.data + (size_t)uyc * im.stride;
	const double t00 = r0[lx], t01 = r0[uxc], t10 = r1[lx], t11 = r1[uxc];
	const double v = t00 * (1 - dx) * (1 - dy) + t01 * dx * (1 - dy) + t10 * (1 - dx) * dy + t11 * dx * dy;
	return (in0 && in1) ? v : 128.0;
}

template <int AM, int PPT>
__global__ __launch_bounds__(kBlock) void k_iclk_track(BatchView bv, ImgView im, mtfhip_sm_desc sm, TrackState ts,
	const double *h0inv_all, const double *ncc_sc_all, double norm_mult, double norm_add) {
	__shared__ double red[4 * 8];
	__shared__ double sW[9], sSt[8], sHinv[64], sIc[12], sCr[8];
	__shared__ int sDone;
	const int t = blockIdx.x, N = bv.N, S = bv.S, tid = threadIdx.x;
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY;
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * N;
	const double2 *ih = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_HXY]) + (size_t)t * N;
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * N;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N;
	const double *J0 = bv.buf[MTFHIP_BUF_J0] + (size_t)t * N * S;
	const double m0 = AM == MTFHIP_AM_NCC ? ncc_sc_all[t * 8 + 0] : 0.0;
	const double cn = AM == MTFHIP_AM_NCC ? ncc_sc_all[t * 8 + 1] : 1.0;
	/* Everything that does not change over the iterations is fetched ONCE: the thread's grid points, template values
	 * and J0 rows into registers, the inverse Hessian and the corner sets into LDS.  An iteration then touches global
	 * memory only for its texels (the loop is a chain of dependent latencies: one workgroup per patch, nothing to
	 * overlap with). */
	constexpr bool HOIST_J = PPT <= 4;   /* 8 J0 values per pixel: beyond 4 pixels per thread they would spill */
	constexpr bool HOIST_P = PPT <= 8;   /* grid point + z: 3 doubles per pixel */
	double2 hpv[HOIST_P ? PPT : 1];
	double zv[HOIST_P ? PPT : 1], i0v[PPT], j0v[HOIST_J ? PPT : 1][8];
#pragma unroll
	for (int k = 0; k < PPT; ++k) {
		const int i = tid + k * kBlock;
		const int ic = i < N ? i : N - 1;
		if constexpr (HOIST_P) {
			hpv[k] = bv.unit_z ? ip[ic] : ih[ic];
			zv[k] = bv.unit_z ? 1.0 : iz[ic];
		}
		i0v[k] = i < N ? I0[ic] : 0.0;
		if constexpr (HOIST_J) {
#pragma unroll
			for (int s = 0; s < 8; ++s) j0v[k][s] = (s < S && i < N) ? J0[(size_t)s * N + ic] : 0.0;
		}
	}
	if (tid < 64) sHinv[tid] = h0inv_all[(size_t)t * 64 + tid];
	if (tid < 12) sIc[tid] = ts.init_corners_hm[12 * t + tid];
	if (tid < 8) sCr[tid] = ts.corners[8 * t + tid];
	if (tid < 9) sW[tid] = bv.warps[9 * t + tid];
	if (tid < 8) sSt[tid] = bv.states[8 * t + tid];
	if (tid == 0) sDone = 0;
	__syncthreads();
	int n_it = 0;
	double f_last = 0;
	for (int it = 0; it < sm.max_iters; ++it) {
		double W[9];
#pragma unroll
		for (int q = 0; q < 9; ++q) W[q] = sW[q];
		/* ---- updatePixVals: It = sample(curr_warp * init_pts) ---- */
		double itv[PPT];
		double s1[1] = {0.0};
#pragma unroll
		for (int k = 0; k < PPT; ++k) {
			const int ick = (tid + k * kBlock < N) ? tid + k * kBlock : N - 1;
			const double2 hp = HOIST_P ? hpv[HOIST_P ? k : 0] : (bv.unit_z ? ip[ick] : ih[ick]);
			const double z = HOIST_P ? zv[HOIST_P ? k : 0] : (bv.unit_z ? 1.0 : iz[ick]);
			double wx, wy;
			if (hom) {
				const double cx = W[0] * hp.x + W[1] * hp.y + W[2] * z, cy = W[3] * hp.x + W[4] * hp.y + W[5] * z;
				const double d = W[6] * hp.x + W[7] * hp.y + W[8] * z;
				wx = cx / d; wy = cy / d;
			} else {
				wx = W[0] * hp.x + W[1] * hp.y + W[2] * z; wy = W[3] * hp.x + W[4] * hp.y + W[5] * z;
			}
			const double v = norm_mult * pix_val_select(im, wx, wy) + norm_add;
			itv[k] = (tid + k * kBlock < N) ? v : 0.0;
		}
#pragma unroll
		for (int k = 0; k < PPT; ++k) if (tid + k * kBlock < N) s1[0] += itv[k];
		double dfv[PPT];
		if constexpr (AM == MTFHIP_AM_NCC) {
			/* ---- NCC::updateSimilarity + updateInitGrad ---- */
			block_allsum<1>(s1, red);
			const double mt = s1[0] / (double)N;
			double s2[2] = {0.0, 0.0};
#pragma unroll
			for (int k = 0; k < PPT; ++k)
				if (tid + k * kBlock < N) {
					const double a0 = i0v[k] - m0, at = itv[k] - mt;
					s2[0] = fma(a0, at, s2[0]); s2[1] = fma(at, at, s2[1]);
				}
			block_allsum<2>(s2, red);
			const double b = sqrt(s2[1]);
			const double f = s2[0] / (b * cn);
			f_last = f;
			double s3[1] = {0.0};
#pragma unroll
			for (int k = 0; k < PPT; ++k) {
				dfv[k] = 0;
				if (tid + k * kBlock < N) {
					const double itc_b = (itv[k] - mt) / b, i0c_c = (i0v[k] - m0) / cn;
					dfv[k] = (itc_b - f * i0c_c) / cn;
					s3[0] += dfv[k];
				}
			}
			block_allsum<1>(s3, red);
			const double gm = s3[0] / (double)N;
#pragma unroll
			for (int k = 0; k < PPT; ++k) dfv[k] -= gm;
		} else {
			/* ---- SSD: df_dI0 = I_diff = It - I0, f = -|r|^2 / 2 ---- */
			double s2[1] = {0.0};
#pragma unroll
			for (int k = 0; k < PPT; ++k) {
				dfv[k] = (tid + k * kBlock < N) ? itv[k] - i0v[k] : 0.0;
				s2[0] = fma(dfv[k], dfv[k], s2[0]);
			}
			block_allsum<1>(s2, red);
			f_last = -s2[0] / 2;
		}
		/* ---- cmptInitJacobian: g = df_dI0 * J0 ---- */
		double g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
		for (int k = 0; k < PPT; ++k) {
			if (tid + k * kBlock < N) {
#pragma unroll
				for (int s = 0; s < 8; ++s)
					if (s < S) g[s] = fma(dfv[k], HOIST_J ? j0v[HOIST_J ? k : 0][s] : J0[(size_t)s * N + tid + k * kBlock], g[s]);
			}
		}
		block_allsum<8>(g, red);
		/* ---- solve, invert, compose, converge (thread 0) ---- */
		if (tid == 0) {
			double dp[8];
			for (int r = 0; r < 8; ++r) {
				double acc = 0;
				if (r < S) for (int c = 0; c < S; ++c) acc += sHinv[c * S + r] * g[c];
				dp[r] = -acc;
			}
			double U[9];
			if (hom) { U[0] = 1 + dp[0]; U[1] = dp[1]; U[2] = dp[2]; U[3] = dp[3]; U[4] = 1 + dp[4]; U[5] = dp[5]; U[6] = dp[6]; U[7] = dp[7]; U[8] = 1; }
			else { U[0] = 1 + dp[2]; U[1] = dp[3]; U[2] = dp[0]; U[3] = dp[4]; U[4] = 1 + dp[5]; U[5] = dp[1]; U[6] = 0; U[7] = 0; U[8] = 1; }
			double c9[9];
			c9[0] = U[4] * U[8] - U[5] * U[7]; c9[1] = U[2] * U[7] - U[1] * U[8]; c9[2] = U[1] * U[5] - U[2] * U[4];
			c9[3] = U[5] * U[6] - U[3] * U[8]; c9[4] = U[0] * U[8] - U[2] * U[6]; c9[5] = U[2] * U[3] - U[0] * U[5];
			c9[6] = U[3] * U[7] - U[4] * U[6]; c9[7] = U[1] * U[6] - U[0] * U[7]; c9[8] = U[0] * U[4] - U[1] * U[3];
			const double inv_det = 1.0 / (U[0] * c9[0] + U[1] * c9[3] + U[2] * c9[6]);
			for (int q = 0; q < 9; ++q) c9[q] *= inv_det;
			const double n22 = c9[8];
			for (int q = 0; q < 9; ++q) U[q] = c9[q] / n22;
			U[0] = 1 + (U[0] - 1); U[4] = 1 + (U[4] - 1); U[8] = 1;
			if (!hom) { U[6] = 0; U[7] = 0; }
			double Wn[9];
			for (int r = 0; r < 3; ++r)
				for (int c = 0; c < 3; ++c) Wn[3 * r + c] = W[3 * r] * U[c] + W[3 * r + 1] * U[3 + c] + W[3 * r + 2] * U[6 + c];
			if (hom) {
				const double w22 = Wn[8];
				for (int q = 0; q < 9; ++q) Wn[q] /= w22;
				sSt[0] = Wn[0] - 1; sSt[1] = Wn[1]; sSt[2] = Wn[2]; sSt[3] = Wn[3]; sSt[4] = Wn[4] - 1; sSt[5] = Wn[5]; sSt[6] = Wn[6]; sSt[7] = Wn[7];
			} else {
				sSt[0] = Wn[2]; sSt[1] = Wn[5]; sSt[2] = Wn[0] - 1; sSt[3] = Wn[1]; sSt[4] = Wn[3]; sSt[5] = Wn[4] - 1; sSt[6] = 0; sSt[7] = 0;
			}
			for (int q = 0; q < 9; ++q) sW[q] = Wn[q];
			double change = 0;
			for (int q = 0; q < 4; ++q) {
				const double X = sIc[3 * q], Y = sIc[3 * q + 1], Z = sIc[3 * q + 2];
				double nx = Wn[0] * X + Wn[1] * Y + Wn[2] * Z, ny = Wn[3] * X + Wn[4] * Y + Wn[5] * Z;
				if (hom) { const double d = Wn[6] * X + Wn[7] * Y + Wn[8] * Z; nx = nx / d; ny = ny / d; }
				const double ddx = sCr[2 * q] - nx, ddy = sCr[2 * q + 1] - ny;
				change += ddx * ddx + ddy * ddy;
				sCr[2 * q] = nx; sCr[2 * q + 1] = ny;
			}
			if (change < sm.epsilon) sDone = 1;
		}
		++n_it;
		__syncthreads();
		if (sDone) break;
	}
	if (tid < 9) bv.warps[9 * t + tid] = sW[tid];
	if (tid < 8) bv.states[8 * t + tid] = sSt[tid];
	if (tid < 8) ts.corners[8 * t + tid] = sCr[tid];
	if (tid == 0) { ts.n_iters[t] = n_it; ts.acc[(size_t)t * ACC_COUNT + ACC_RR] = f_last; }
}

/* ===================================================================== */
/* launchers                                                              */
/* ===================================================================== */
static inline dim3 grid2(int nblk, int B) { return dim3((unsigned)nblk, (unsigned)B, 1); }

void launch_init_grid(const BatchView &bv, const double *dev_w0, int resx, int resy, double lo_x, double lo_y,
	double hi_x, double hi_y, int force_unit_z, hipStream_t st) {
	hipLaunchKernelGGL(k_init_grid, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv, dev_w0, resx, resy,
		lo_x, lo_y, hi_x, hi_y, force_unit_z);
}
void launch_apply_warp(const BatchView &bv, hipStream_t st) {
	hipLaunchKernelGGL(k_apply_warp, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv);
}
void launch_grad_pts(const BatchView &bv, double eps, hipStream_t st) {   /* per sample point */
	hipLaunchKernelGGL(k_grad_pts, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv, eps);
}
void launch_sample(const BatchView &bv, const ImgView &im, const double *pts, double *out, double mult, double add,
	hipStream_t st) {
	if (bv.C > 1) { hipLaunchKernelGGL(k_sample_mc, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.NP, bv.C, im, pts, out, mult, add); return; }
	hipLaunchKernelGGL(k_sample, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, pts, out, mult, add);
}
void launch_img_grad(const BatchView &bv, const ImgView &im, const double *pts, double *grad, double eps, double mult,
	hipStream_t st) {
	if (bv.C > 1) {
		hipLaunchKernelGGL(k_img_grad_mc, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.NP, bv.C, im, pts, (const double *)nullptr, grad, eps, mult);
		return;
	}
	hipLaunchKernelGGL(k_img_grad, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, pts, grad, eps, mult);
}
void launch_warped_img_grad(const BatchView &bv, const ImgView &im, const double *gp, double *grad, double eps,
	double mult, hipStream_t st) {
	if (bv.C > 1) {
		hipLaunchKernelGGL(k_img_grad_mc, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.NP, bv.C, im, (const double *)nullptr, gp, grad, eps, mult);
		return;
	}
	hipLaunchKernelGGL(k_warped_img_grad, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, gp, grad, eps, mult);
}
void launch_pix_jacobian(const BatchView &bv, int variant, const double *grad, double *J, hipStream_t st) {
	hipLaunchKernelGGL(k_pix_jacobian, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv, variant, grad, J);
}
void launch_hess_pts(const BatchView &bv, double eps, hipStream_t st) {
	hipLaunchKernelGGL(k_hess_pts, dim3(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv, eps);
}
void launch_img_hess(const BatchView &bv, const ImgView &im, const double *pts, double *hess, double eps, double mult, hipStream_t st) {
	if (bv.C > 1) {
		hipLaunchKernelGGL(k_img_hess_mc, dim3(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.NP, bv.C, im, pts, (const double *)nullptr, hess, eps, mult);
		return;
	}
	hipLaunchKernelGGL(k_img_hess, dim3(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, pts, hess, eps, mult);
}
void launch_warped_img_hess(const BatchView &bv, const ImgView &im, const double *pts, const double *hp, double *hess, double eps,
	double mult, hipStream_t st) {
	if (bv.C > 1) {
		hipLaunchKernelGGL(k_img_hess_mc, dim3(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.NP, bv.C, im, pts, hp, hess, eps, mult);
		return;
	}
	hipLaunchKernelGGL(k_warped_img_hess, dim3(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, pts, hp, hess, eps, mult);
}
void launch_pix_hessian(const BatchView &bv, int variant, const double *hess, const double *grad, double *D, hipStream_t st) {
	const dim3 grid(simple_blocks_per_target(bv.N), bv.B);
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) hipLaunchKernelGGL(k_pix_hessian<MTFHIP_SSM_HOMOGRAPHY>, grid, dim3(kBlock), 0, st, bv, variant, hess, grad, D);
	else hipLaunchKernelGGL(k_pix_hessian<MTFHIP_SSM_AFFINE>, grid, dim3(kBlock), 0, st, bv, variant, hess, grad, D);
}
void launch_weighted_plane_sum(const BatchView &bv, const double *d2a, const double *d2b, const double *w, double *partials, int nblk,
	double *out, hipStream_t st) {
	const dim3 grid(nblk, bv.B);
	if (bv.S == 8) hipLaunchKernelGGL(k_weighted_plane_sum<64>, grid, dim3(kBlock), 0, st, bv.N, d2a, d2b, w, partials, nblk);
	else hipLaunchKernelGGL(k_weighted_plane_sum<36>, grid, dim3(kBlock), 0, st, bv.N, d2a, d2b, w, partials, nblk);
	hipLaunchKernelGGL(k_plane_sum_finish, dim3(bv.B), dim3(64), 0, st, partials, nblk, bv.S * bv.S, out);
}
void launch_second_order_ssd(const BatchView &bv, const ImgView &im, int term, int chained, int d0_variant, double grad_eps,
	double hess_eps, double norm_mult, double norm_add, double *partials, int nblk, double *out, hipStream_t st) {
	const dim3 grid(nblk, bv.B);
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY)
		hipLaunchKernelGGL(k_second_order_ssd<MTFHIP_SSM_HOMOGRAPHY>, grid, dim3(kBlock), 0, st, bv, im, term, chained, d0_variant, grad_eps,
			hess_eps, norm_mult, norm_add, partials, nblk);
	else
		hipLaunchKernelGGL(k_second_order_ssd<MTFHIP_SSM_AFFINE>, grid, dim3(kBlock), 0, st, bv, im, term, chained, d0_variant, grad_eps,
			hess_eps, norm_mult, norm_add, partials, nblk);
	hipLaunchKernelGGL(k_plane_sum_finish, dim3(bv.B), dim3(64), 0, st, partials, nblk, bv.S * bv.S, out);
}
void launch_mean_planes(const double *a, const double *b, double *o, size_t n, hipStream_t st) {
	hipLaunchKernelGGL(k_mean_jacobian, dim3((unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, 4096)), dim3(kBlock), 0, st, a, b, o, n);
}
void launch_mean_jacobian(const BatchView &bv, hipStream_t st) {
	size_t n = (size_t)bv.B * bv.N * bv.S;
	int nb = (int)((n + kBlock * 4 - 1) / (kBlock * 4));
	hipLaunchKernelGGL(k_mean_jacobian, dim3(nb), dim3(kBlock), 0, st, bv.buf[MTFHIP_BUF_J0], bv.buf[MTFHIP_BUF_JT],
		bv.buf[MTFHIP_BUF_JM], n);
}
void launch_negate(const double *src, double *dst, size_t n, hipStream_t st) {
	int nb = (int)((n + kBlock * 4 - 1) / (kBlock * 4));
	hipLaunchKernelGGL(k_negate, dim3(nb), dim3(kBlock), 0, st, src, dst, n);
}
void launch_ssd_residual(const BatchView &bv, double *partials, int nblk, hipStream_t st) {
	hipLaunchKernelGGL(k_ssd_residual, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv, partials, nblk);
}
void launch_gemv(const BatchView &bv, const double *v1, const double *J1, const double *v2, const double *J2,
	int sum_mode, double *partials, int nblk, hipStream_t st) {
	hipLaunchKernelGGL(k_gemv, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv.N, bv.S, v1, J1, v2, J2, sum_mode, partials, nblk);
}
void launch_gram(const BatchView &bv, const double *J, double *partials, int nblk, hipStream_t st) {
	hipLaunchKernelGGL(k_gram, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv.N, bv.S, J, partials, nblk);
}
void launch_vec_sum(const BatchView &bv, const double *v, double *partials, int nblk, hipStream_t st) {
	hipLaunchKernelGGL(k_vec_sum, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv.N, v, partials, nblk);
}
void launch_ncc_centered(const BatchView &bv, const double *sc, double *partials, int nblk, hipStream_t st) {
	hipLaunchKernelGGL(k_ncc_centered, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv, sc, partials, nblk);
}
void launch_ncc_grad(const BatchView &bv, const double *sc, int curr, double *out, double *partials, int nblk, hipStream_t st) {
	hipLaunchKernelGGL(k_ncc_grad, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv, sc, curr, out, partials, nblk);
}
void launch_sub_mean(const BatchView &bv, double *v, const double *sc, hipStream_t st) {
	hipLaunchKernelGGL(k_sub_mean, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, v, sc);
}
void launch_col_sum(const BatchView &bv, const double *J, double *partials, int nblk, hipStream_t st) {
	hipLaunchKernelGGL(k_col_sum, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv.N, bv.S, J, partials, nblk);
}
void launch_ncc_hess(const BatchView &bv, const double *sc, const double *colmean, const double *J, double *partials,
	int nblk, hipStream_t st) {
	hipLaunchKernelGGL(k_ncc_hess, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv, sc, colmean, J, partials, nblk);
}
void launch_mi_hist(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, double *partials,
	int nblk, int row_len, hipStream_t st) {
	/* per-wave staging slabs [64][2 nb]; the same LDS later holds the four waves' [nb + nb^2] rows */
	const size_t lds = sizeof(double) * std::max<size_t>((size_t)4 * kMiRow * 2 * nb, (size_t)4 * (nb + nb * nb));
	static const bool use_mfma = !(getenv("MTFHIP_MI_MFMA") && atoi(getenv("MTFHIP_MI_MFMA")) == 0);
	if (use_mfma) hipLaunchKernelGGL(k_mi_hist<true>, grid2(nblk, bv.B), dim3(kBlock), lds, st, bv.N, nb, norm_mult, A, Bv, partials, nblk, row_len);
	else hipLaunchKernelGGL(k_mi_hist<false>, grid2(nblk, bv.B), dim3(kBlock), lds, st, bv.N, nb, norm_mult, A, Bv, partials, nblk, row_len);
}
void launch_mi_hist_finish(const BatchView &bv, int nb, double pre_seed, double norm_mult, int mode, int first_init,
	const double *partials, int nblk, int row_len, double *tb, double *f_out, hipStream_t st) {
	hipLaunchKernelGGL(k_mi_hist_finish, dim3(bv.B), dim3(kBlock), 0, st, nb, pre_seed, norm_mult, mode, first_init, partials,
		nblk, row_len, tb, f_out);
}
void launch_mi_factor(const BatchView &bv, int nb, int curr, double *tb, hipStream_t st) {
	hipLaunchKernelGGL(k_mi_factor, dim3(bv.B), dim3(kBlock), 0, st, nb, curr, tb);
}
void launch_mi_grad(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, const double *tb,
	int table_off, double *out, hipStream_t st) {
	hipLaunchKernelGGL(k_mi_grad, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, nb, norm_mult, A, Bv,
		tb, table_off, out);
}
void launch_mi_hess(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, const double *tb,
	int table_off, int transpose_q, const double *J, double *partials, int nblk, int row_len, hipStream_t st) {
	const size_t slabs = std::max<size_t>((size_t)4 * kMiRow * (2 * nb + kMaxS), (size_t)4 * nb * nb * bv.S);
	const size_t lds = sizeof(double) * (MI_NB * MI_NB + 4 * 36 + slabs);
	static bool attr_set = false;
	if (!attr_set) {   /* 16 bins need 82 KB of dynamic LDS; the default cap is 64 KB */
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_mi_hess<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_mi_hess<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		attr_set = true;
	}
	static const bool use_mfma = !(getenv("MTFHIP_MI_MFMA") && atoi(getenv("MTFHIP_MI_MFMA")) == 0);
	if (use_mfma && nb == 8)
		hipLaunchKernelGGL(k_mi_hess<true>, grid2(nblk, bv.B), dim3(kBlock), lds, st, bv.N, bv.S, nb, norm_mult, A, Bv, tb, table_off,
			transpose_q, J, partials, nblk, row_len);
	else
		hipLaunchKernelGGL(k_mi_hess<false>, grid2(nblk, bv.B), dim3(kBlock), lds, st, bv.N, bv.S, nb, norm_mult, A, Bv, tb, table_off,
			transpose_q, J, partials, nblk, row_len);
}
void launch_mi_hess_finish(const BatchView &bv, int nb, const double *partials, int nblk, int row_len, const double *tb,
	int joint_off, int hist_off, int transpose_q, double *out, hipStream_t st) {
	size_t lds = sizeof(double) * ((size_t)nb * nb * bv.S + 36);
	hipLaunchKernelGGL(k_mi_hess_finish, dim3(bv.B), dim3(kBlock), lds, st, bv.S, nb, partials, nblk, row_len, tb, joint_off,
		hist_off, transpose_q, out);
}
void launch_finish_rows(double *partials, int nblk, int row_len, double *out, int B, hipStream_t st) {
	hipLaunchKernelGGL(k_finish_rows, dim3(B), dim3(128), 0, st, partials, nblk, row_len, out);
}
void launch_finish(double *partials, int nblk, double *out, int B, hipStream_t st) {
	hipLaunchKernelGGL(k_finish, dim3(B), dim3(64), 0, st, partials, nblk, out);
}

template <int SSM, bool CHAINED, int MODE>
static void launch_fused_mat(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk,
	hipStream_t st) {
	dim3 g = grid2(nblk, bv.B);
	if (bv.am == MTFHIP_AM_NCC) {
		if (fa.materialize)
			hipLaunchKernelGGL((k_fused_ncc<SSM, CHAINED, MODE, true>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
		else
			hipLaunchKernelGGL((k_fused_ncc<SSM, CHAINED, MODE, false>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
		return;
	}
	if (fa.materialize)
		hipLaunchKernelGGL((k_fused_ssd<SSM, CHAINED, MODE, true>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
	else
		hipLaunchKernelGGL((k_fused_ssd<SSM, CHAINED, MODE, false>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
}
template <int SSM, bool CHAINED>
static void launch_fused_mode(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk,
	hipStream_t st) {
	if (fa.mode == 0) launch_fused_mat<SSM, CHAINED, 0>(bv, im, fa, partials, nblk, st);
	else if (fa.mode == 1) launch_fused_mat<SSM, CHAINED, 1>(bv, im, fa, partials, nblk, st);
	else launch_fused_mat<SSM, CHAINED, 2>(bv, im, fa, partials, nblk, st);
}
void launch_fused_ssd(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk,
	hipStream_t st) {
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY;
	if (hom && fa.chained) launch_fused_mode<MTFHIP_SSM_HOMOGRAPHY, true>(bv, im, fa, partials, nblk, st);
	else if (hom) launch_fused_mode<MTFHIP_SSM_HOMOGRAPHY, false>(bv, im, fa, partials, nblk, st);
	else if (fa.chained) launch_fused_mode<MTFHIP_SSM_AFFINE, true>(bv, im, fa, partials, nblk, st);
	else launch_fused_mode<MTFHIP_SSM_AFFINE, false>(bv, im, fa, partials, nblk, st);
}

/* LDS-staged scorer: returns false (nothing launched) when template + tile do not fit the 160 KB of a CU */
bool launch_score_candidates_lds(const BatchView &bv, const ImgView &im, const double *dev_states, int C, int tx0, int ty0,
	int tw, int th, double likelihood_alpha, double *unit_sums, double *dev_lik, double *dev_sim, hipStream_t st) {
	const size_t lds = (size_t)bv.N * (16 + 8 + 8) + (size_t)tw * th * 4;
	if (lds > 160 * 1024 || tw <= 1 || th <= 1) return false;
	static bool attr_set = false;
	if (!attr_set) {
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_score_candidates_lds<MTFHIP_SSM_HOMOGRAPHY>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_score_candidates_lds<MTFHIP_SSM_AFFINE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		attr_set = true;
	}
	/* one workgroup per CU (the staged template + tile take most of its LDS); work units are dealt round-robin */
	const int units = C * kScoreSplit, waves = kScoreBlock / 64;
	int nb = (units + waves - 1) / waves;
	if (nb > 256) nb = 256;
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY)
		hipLaunchKernelGGL(k_score_candidates_lds<MTFHIP_SSM_HOMOGRAPHY>, dim3(nb), dim3(kScoreBlock), lds, st, bv, im, dev_states, C,
			tx0, ty0, tw, th, 1.0, 0.0, unit_sums);
	else
		hipLaunchKernelGGL(k_score_candidates_lds<MTFHIP_SSM_AFFINE>, dim3(nb), dim3(kScoreBlock), lds, st, bv, im, dev_states, C,
			tx0, ty0, tw, th, 1.0, 0.0, unit_sums);
	hipLaunchKernelGGL(k_score_finish, dim3((C + kBlock - 1) / kBlock), dim3(kBlock), 0, st, unit_sums, C, bv.N, likelihood_alpha, dev_lik, dev_sim);
	return true;
}
void launch_to_gray(const void *raw, int rows, int cols, size_t stride_bytes, int channels, int depth_f32, float *out, hipStream_t st) {
	hipLaunchKernelGGL(k_to_gray_f32, dim3((cols + kBlock - 1) / kBlock, rows), dim3(kBlock), 0, st, (const unsigned char *)raw, rows, cols,
		stride_bytes, channels, depth_f32, out);
}
void launch_sym5(const float *src, float *tmp, float *dst, int rows, int cols, const float kx[3], const float ky[3], hipStream_t st) {
	const dim3 grid((cols + kBlock - 1) / kBlock, rows);
	hipLaunchKernelGGL(k_sym5_rows, grid, dim3(kBlock), 0, st, src, rows, cols, kx[0], kx[1], kx[2], tmp);
	hipLaunchKernelGGL(k_sym5_cols, grid, dim3(kBlock), 0, st, (const float *)tmp, rows, cols, ky[0], ky[1], ky[2], dst);
}
void launch_pyr_down(const float *src, int srows, int scols, float *dst, int drows, int dcols, hipStream_t st) {
	hipLaunchKernelGGL(k_pyr_down, dim3((dcols + kBlock - 1) / kBlock, drows), dim3(kBlock), 0, st, src, srows, scols, drows, dcols, dst);
}
void launch_resize_linear(const float *src, int srows, int scols, float *dst, int drows, int dcols, hipStream_t st) {
	hipLaunchKernelGGL(k_resize_linear, dim3((dcols + kBlock - 1) / kBlock, drows), dim3(kBlock), 0, st, src, srows, scols, drows, dcols, dst);
}
void launch_score_candidates(const BatchView &bv, const ImgView &im, const double *dev_states, int C,
	double likelihood_alpha, double *dev_lik, double *dev_sim, hipStream_t st) {
	int nb = (C + (kBlock / 64) - 1) / (kBlock / 64);
	hipLaunchKernelGGL(k_score_candidates, dim3(nb), dim3(kBlock), 0, st, bv, im, dev_states, C, likelihood_alpha,
		1.0, 0.0, dev_lik, dev_sim);
}

template <int AM>
static bool launch_iclk_track_am(const BatchView &bv, const ImgView &im, const mtfhip_sm_desc &sm, const TrackState &ts,
	const double *h0inv, const double *ncc_sc, double norm_mult, double norm_add, hipStream_t st) {
	const int ppt = (bv.N + kBlock - 1) / kBlock;
#define MTFHIP_ICLK_CASE(P) hipLaunchKernelGGL((k_iclk_track<AM, P>), dim3(bv.B), dim3(kBlock), 0, st, bv, im, sm, ts, h0inv, ncc_sc, norm_mult, norm_add)
	if (ppt <= 1) MTFHIP_ICLK_CASE(1);
	else if (ppt <= 2) MTFHIP_ICLK_CASE(2);
	else if (ppt <= 3) MTFHIP_ICLK_CASE(3);
	else if (ppt <= 4) MTFHIP_ICLK_CASE(4);
	else if (ppt <= 8) MTFHIP_ICLK_CASE(8);
	else if (ppt <= 16) MTFHIP_ICLK_CASE(16);
	else return false;
#undef MTFHIP_ICLK_CASE
	return true;
}
bool launch_iclk_track(const BatchView &bv, const ImgView &im, const mtfhip_sm_desc &sm, const TrackState &ts,
	const double *h0inv, const double *ncc_sc, double norm_mult, double norm_add, hipStream_t st) {
	if (bv.am == MTFHIP_AM_NCC) return launch_iclk_track_am<MTFHIP_AM_NCC>(bv, im, sm, ts, h0inv, ncc_sc, norm_mult, norm_add, st);
	return launch_iclk_track_am<MTFHIP_AM_SSD>(bv, im, sm, ts, h0inv, ncc_sc, norm_mult, norm_add, st);
}
void launch_sample_candidates(const BatchView &bv, const ImgView &im, const double *dev_states, int C, double norm_mult,
	double norm_add, double *dev_feat, hipStream_t st) {
	hipLaunchKernelGGL(k_sample_candidates, dim3(C), dim3(kBlock), 0, st, bv, im, dev_states, C, norm_mult, norm_add, dev_feat);
	if (bv.am == MTFHIP_AM_NCC) hipLaunchKernelGGL(k_ncc_feature_rows, dim3(C), dim3(kBlock), 0, st, bv.N, dev_feat);
}
void launch_finish_track(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts, const double *partials,
	int nblk, hipStream_t st) {
	/* NCC rows are 72 wide: two waves load them, the first one solves */
	hipLaunchKernelGGL(k_finish_track, dim3(bv.B), dim3(bv.am == MTFHIP_AM_NCC ? 128 : 64), 0, st, bv, sm, ts, partials, nblk);
}

} // namespace mtfhip
