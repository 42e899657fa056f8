/*
 * mtfhip_rng_device.h -- the device-side random draws of the batch drivers (the particle filter's proposals, kernels_pf.hip; the NN dataset's
 * perturbations, kernels_nn.hip): Philox4x32-10 keyed by (seed, iteration / stream, item, draw) + Box-Muller.  Counter-based and stateless, so a
 * draw is a pure function of its indices: every rank of a sharded run produces the same numbers for the same item without any exchange.
 */
#pragma once
#include "mtfhip_device.h"

namespace mtfhip {

/* ---- 3 x 3 helpers (row-major) ---- */
__device__ __forceinline__ void m3_mul_dev(const double *a, const double *b, double *c) {
#pragma unroll
	for (int i = 0; i < 3; ++i)
#pragma unroll
		for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
__device__ __forceinline__ void m3_inv_dev(const double *u, double *c) {   /* Matrix3d::inverse(): cofactors / determinant */
	c[0] = u[4] * u[8] - u[5] * u[7]; c[1] = u[2] * u[7] - u[1] * u[8]; c[2] = u[1] * u[5] - u[2] * u[4];
	c[3] = u[5] * u[6] - u[3] * u[8]; c[4] = u[0] * u[8] - u[2] * u[6]; c[5] = u[2] * u[3] - u[0] * u[5];
	c[6] = u[3] * u[7] - u[4] * u[6]; c[7] = u[1] * u[6] - u[0] * u[7]; c[8] = u[0] * u[4] - u[1] * u[3];
	const double inv_det = 1.0 / (u[0] * c[0] + u[1] * c[3] + u[2] * c[6]);
#pragma unroll
	for (int q = 0; q < 9; ++q) c[q] *= inv_det;
}
template <int SSM>
__device__ __forceinline__ void warp_from_state_dev(const double *p, double *W) {   /* Homography.cc:94-107, Affine.cc:116-130 */
	if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
		W[0] = 1 + p[0]; W[1] = p[1]; W[2] = p[2]; W[3] = p[3]; W[4] = 1 + p[4]; W[5] = p[5]; W[6] = p[6]; W[7] = p[7]; W[8] = 1;
	} else {
		W[0] = 1 + p[2]; W[1] = p[3]; W[2] = p[0]; W[3] = p[4]; W[4] = 1 + p[5]; W[5] = p[1]; W[6] = 0; W[7] = 0; W[8] = 1;
	}
}
template <int SSM>
__device__ __forceinline__ void state_from_warp_dev(double *p, const double *W) {   /* Homography.cc:116-132, Affine.cc:132-143 */
	if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
		p[0] = W[0] - 1; p[1] = W[1]; p[2] = W[2]; p[3] = W[3]; p[4] = W[4] - 1; p[5] = W[5]; p[6] = W[6]; p[7] = W[7];
	} else {
		p[0] = W[2]; p[1] = W[5]; p[2] = W[0] - 1; p[3] = W[1]; p[4] = W[3]; p[5] = W[4] - 1; p[6] = p[7] = 0;
	}
}
/* ---- Philox4x32-10 (Salmon et al., SC'11): counter-based, stateless ---- */
struct Philox4 { unsigned c[4]; };
__device__ __forceinline__ Philox4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
	for (int r = 0; r < 10; ++r) {
		const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
		const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
		c0 = n0; c1 = n1; c2 = n2; c3 = n3;
		k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
	}
	return Philox4{{c0, c1, c2, c3}};
}
/* two uniforms in (0, 1] with 53 and 32 + 21 random bits */
__device__ __forceinline__ void philox_uniform2(const Philox4 &r, double &u0, double &u1) {
	const unsigned long long a = ((unsigned long long)r.c[0] << 21) | (r.c[1] >> 11), b = ((unsigned long long)r.c[2] << 21) | (r.c[3] >> 11);
	u0 = ((double)a + 1.0) * (1.0 / 9007199254740992.0);
	u1 = ((double)b + 1.0) * (1.0 / 9007199254740992.0);
}
/* log(u) for u in (0, 1]: u = m 2^e with m in [sqrt(1/2), sqrt(2)), log m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716 --
 * twelve terms of the odd series (the next one is below 1e-18).  ~1 ulp; about a quarter of the instructions of the library
 * routine, whose special cases (negative, zero, infinite, subnormal arguments) cannot occur here. */
__device__ __forceinline__ double pf_log_unit(double u) {
	int e;
	double m = frexp(u, &e);                 /* m in [0.5, 1) */
	if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
	const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
	double p = 1.0 / 23.0;
	p = fma(p, s2, 1.0 / 21.0); p = fma(p, s2, 1.0 / 19.0); p = fma(p, s2, 1.0 / 17.0); p = fma(p, s2, 1.0 / 15.0);
	p = fma(p, s2, 1.0 / 13.0); p = fma(p, s2, 1.0 / 11.0); p = fma(p, s2, 1.0 / 9.0); p = fma(p, s2, 1.0 / 7.0);
	p = fma(p, s2, 1.0 / 5.0); p = fma(p, s2, 1.0 / 3.0);
	const double lm = 2.0 * fma(s * s2, p, s);
	const double ed = (double)e;
	return fma(ed, 0.69314718055994528623, fma(ed, 2.3190468138462995584e-17, lm));   /* ln 2 = hi + lo */
}
/* (sin, cos)(2 pi u) for u in (0, 1]: octant reduction in units of pi / 4 (exact: 8 u, its floor and the remainder are all
 * representable), Taylor polynomials on [-pi/4, pi/4] (degree 17 / 16: truncation below 1e-18), quadrant rotation */
__device__ __forceinline__ void pf_sincos_2pi(double u, double &sn, double &cs) {
	const double t = u * 8.0;
	int o = (int)t;                           /* 0 .. 8 */
	double f = t - (double)o;                 /* [0, 1) */
	if (o & 1) { o += 1; f -= 1.0; }
	const double y = f * 0.78539816339744830962, y2 = y * y;
	double ps = -1.0 / 355687428096000.0;     /* -1 / 17! */
	ps = fma(ps, y2, 1.0 / 1307674368000.0); ps = fma(ps, y2, -1.0 / 6227020800.0); ps = fma(ps, y2, 1.0 / 39916800.0);
	ps = fma(ps, y2, -1.0 / 362880.0); ps = fma(ps, y2, 1.0 / 5040.0); ps = fma(ps, y2, -1.0 / 120.0); ps = fma(ps, y2, 1.0 / 6.0);
	const double sy = fma(-(y * y2), ps, y);
	double pc = 1.0 / 20922789888000.0;       /* 1 / 16! */
	pc = fma(pc, y2, -1.0 / 87178291200.0); pc = fma(pc, y2, 1.0 / 479001600.0); pc = fma(pc, y2, -1.0 / 3628800.0);
	pc = fma(pc, y2, 1.0 / 40320.0); pc = fma(pc, y2, -1.0 / 720.0); pc = fma(pc, y2, 1.0 / 24.0); pc = fma(pc, y2, -0.5);
	const double cy = fma(y2, pc, 1.0);
	const int k = (o >> 1) & 3;
	const double a = (k & 1) ? cy : sy, b = (k & 1) ? sy : cy;
	sn = (k & 2) ? -a : a;
	cs = (k == 1 || k == 2) ? -b : b;
}
/* Box-Muller (the reference draws from boost::normal_distribution over mt11213b, ProjectiveBase.cc:192-197: any exact N(0, 1)
 * sampler is equivalent) */
__device__ __forceinline__ void philox_normal2(unsigned long long seed, unsigned iter, unsigned particle, unsigned draw, double &z0, double &z1) {
	const Philox4 r = philox4x32_10(particle, draw, iter, 0x4E4F524Du /* "NORM" */, (unsigned)seed, (unsigned)(seed >> 32));
	double u0, u1;
	philox_uniform2(r, u0, u1);
	const double rad = sqrt(-2.0 * pf_log_unit(u0));
	double sn, cs;
	pf_sincos_2pi(u1, sn, cs);
	z0 = rad * cs; z1 = rad * sn;
}
__device__ __forceinline__ double philox_uniform(unsigned long long seed, unsigned iter, unsigned particle) {
	const Philox4 r = philox4x32_10(particle, 0u, iter, 0x554E4946u /* "UNIF" */, (unsigned)seed, (unsigned)(seed >> 32));
	double u0, u1;
	philox_uniform2(r, u0, u1);
	return u0;
}

} // namespace mtfhip
