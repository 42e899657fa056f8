/*
 * kernels_pf.hip -- the particle filter's per-iteration work besides scoring, on the device: sample generation (the SSM's
 * stochastic sampler and dynamic models), likelihood mapping, cumulative weights, multinomial resampling, the estimate.
 * (one of the translation units of libmtfhip.so; the scorer itself is k_score_candidates[_fast] in kernels_batch.hip)
 *
 * Reference: nt::PF::update SM/src/NT/PF.cc:207-447, binaryMultinomialResampling :455-502, linearMultinomialResampling
 * :505-536, updateMeanCorners :607-614; Homography::generatePerturbation / compositionalRandomWalk /
 * compositionalAutoRegression1 SSM/src/Homography.cc:899-942; ProjectiveBase::additiveRandomWalk / additiveAutoRegression1 /
 * generatePerturbation / estimateMeanOfSamples SSM/src/ProjectiveBase.cc:236-317.
 *
 * In the reference every particle of every iteration pays a 4-corner DLT through an 8 x 9 JacobiSVD
 * (hom_corner_based_sampling is on by default, parameters.h:262) on one host core; here a particle is one thread and the
 * corner perturbation is the closed-form square-to-quadrilateral map composed with the inverse of the template's.
 * Random draws: the reference seeds boost::mt11213b from random_device (not reproducible), so the draws are an INPUT here --
 * either arrays of standard normals / uniforms handed in by the caller (parity tests, reproducible runs), or a counter-based
 * Philox4x32-10 generator + Box-Muller on the device keyed by (seed, iteration, particle): every rank of a sharded filter
 * regenerates the same particle set without communication.
 */
#include "mtfhip_device.h"

namespace mtfhip {

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
__device__ __forceinline__ void philox_normal2(unsigned long long seed, unsigned iter, unsigned particle, unsigned draw, double &z0, double &z1) {
	const Philox4 r = philox4x32_10(particle, draw, iter, 0x4E4F524Du /* "NORM" */, (unsigned)seed, (unsigned)(seed >> 32));
	double u0, u1;
	philox_uniform2(r, u0, u1);
	const double rad = sqrt(-2.0 * log(u0)), ang = 6.283185307179586476925 * u1;
	z0 = rad * cos(ang); z1 = rad * sin(ang);
}
__device__ __forceinline__ double philox_uniform(unsigned long long seed, unsigned iter, unsigned particle) {
	const Philox4 r = philox4x32_10(particle, 0u, iter, 0x554E4946u /* "UNIF" */, (unsigned)seed, (unsigned)(seed >> 32));
	double u0, u1;
	philox_uniform2(r, u0, u1);
	return u0;
}

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
/* the homography that maps the unit square (0,0) (1,0) (1,1) (0,1) onto four corners TL, TR, BR, BL (Heckbert 1989, eq. 2.12),
 * scaled to m[8] = 1: what the 4-point DLT (utils::computeHomographyDLT, warpUtils.cc:171-224) returns for that input */
__device__ __forceinline__ void square_to_quad_dev(const double *q, double *H) {
	const double x0 = q[0], y0 = q[1], x1 = q[2], y1 = q[3], x2 = q[4], y2 = q[5], x3 = q[6], y3 = q[7];
	const double dx1 = x1 - x2, dx2 = x3 - x2, sx = x0 - x1 + x2 - x3;
	const double dy1 = y1 - y2, dy2 = y3 - y2, sy = y0 - y1 + y2 - y3;
	const double den = dx1 * dy2 - dy1 * dx2;
	const double g = (sx * dy2 - dx2 * sy) / den, h = (dx1 * sy - sx * dy1) / den;
	H[0] = x1 - x0 + g * x1; H[1] = x3 - x0 + h * x3; H[2] = x0;
	H[3] = y1 - y0 + g * y1; H[4] = y3 - y0 + h * y3; H[5] = y0;
	H[6] = g; H[7] = h; H[8] = 1.0;
}

struct PfArgs {
	int n, S;
	int dynamic_model, update_type, corner_based;
	double ar_coeff;
	double sigma[8], mean[8];
	double init_corners[8];
	double sq_inv[9];          /* inverse of square_to_quad(init_corners): template corners -> unit square */
	unsigned long long seed;
	unsigned iter;
	const double *normals;     /* [n][nz] standard normals, or NULL: Philox */
};

/* sample generation: particle_states[k] <- dynamic model(particle_states[k], particle_ar[k], perturbation) (PF.cc:307-335) */
template <int SSM>
__global__ __launch_bounds__(kBlock) void k_pf_propagate(PfArgs a, double *states, double *ars) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	const int k = blockIdx.x * kBlock + threadIdx.x;
	if (k >= a.n) return;
	const int nz = a.corner_based ? 10 : S;
	double z[10];
	if (a.normals) {
		for (int j = 0; j < nz; ++j) z[j] = a.normals[(size_t)k * nz + j];
	} else {
#pragma unroll
		for (int j = 0; j < 10; j += 2) philox_normal2(a.seed, a.iter, (unsigned)k, (unsigned)(j >> 1), z[j], z[j + 1]);
	}
	double pert[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	if (SSM == MTFHIP_SSM_HOMOGRAPHY && a.corner_based) {
		/* Homography::generatePerturbation, corner based (Homography.cc:899-911): one translation for all corners from
		 * distribution 0, one displacement per corner coordinate from distribution 1, then the warp that takes the template
		 * corners to the disturbed ones */
		double dc[8], Hq[9], Hp[9];
		const double tx = a.mean[0] + a.sigma[0] * z[0], ty = a.mean[0] + a.sigma[0] * z[1];
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			dc[2 * c] = a.init_corners[2 * c] + (a.mean[1] + a.sigma[1] * z[2 + 2 * c]) + tx;
			dc[2 * c + 1] = a.init_corners[2 * c + 1] + (a.mean[1] + a.sigma[1] * z[3 + 2 * c]) + ty;
		}
		square_to_quad_dev(dc, Hq);
		m3_mul_dev(Hq, a.sq_inv, Hp);
		const double n22 = Hp[8];
#pragma unroll
		for (int q = 0; q < 9; ++q) Hp[q] /= n22;
		state_from_warp_dev<SSM>(pert, Hp);
	} else {
#pragma unroll
		for (int s = 0; s < S; ++s) pert[s] = a.mean[s] + a.sigma[s] * z[s];   /* ProjectiveBase::generatePerturbation :283-288 */
	}
	double st[8], ar[8], ns[8], nar[8];
	{   /* rows of S contiguous doubles, 16-byte aligned: pairs */
		const double2 *ps = reinterpret_cast<const double2 *>(states + (size_t)k * S), *pa = reinterpret_cast<const double2 *>(ars + (size_t)k * S);
#pragma unroll
		for (int s2 = 0; s2 < 4; ++s2) {
			const double2 v = 2 * s2 < S ? ps[s2] : make_double2(0.0, 0.0), w = 2 * s2 < S ? pa[s2] : make_double2(0.0, 0.0);
			st[2 * s2] = v.x; st[2 * s2 + 1] = v.y; ar[2 * s2] = w.x; ar[2 * s2 + 1] = w.y;
		}
#pragma unroll
		for (int s = 0; s < 8; ++s) nar[s] = ar[s];
	}
	if (a.dynamic_model == 1 && a.update_type == 0) {          /* additiveAutoRegression1 :254-259 */
#pragma unroll
		for (int s = 0; s < 8; ++s) { ns[s] = st[s] + ar[s] + pert[s]; nar[s] = a.ar_coeff * (ns[s] - st[s]); }
	} else if (a.dynamic_model == 1) {                         /* compositionalAutoRegression1 Homography.cc:928-942 */
		double B[9], P[9], A[9], BA[9], W[9], Bi[9], AW[9];
		warp_from_state_dev<SSM>(st, B); warp_from_state_dev<SSM>(pert, P); warp_from_state_dev<SSM>(ar, A);
		m3_mul_dev(B, A, BA); m3_mul_dev(BA, P, W);
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double n22 = W[8]; for (int q = 0; q < 9; ++q) W[q] /= n22; }
		m3_inv_dev(B, Bi); m3_mul_dev(Bi, W, AW);
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double n22 = AW[8]; for (int q = 0; q < 9; ++q) AW[q] /= n22; }
		state_from_warp_dev<SSM>(ns, W); state_from_warp_dev<SSM>(nar, AW);
#pragma unroll
		for (int s = 0; s < 8; ++s) nar[s] *= a.ar_coeff;
	} else if (a.update_type == 0) {                           /* additiveRandomWalk :236-240 */
#pragma unroll
		for (int s = 0; s < 8; ++s) ns[s] = st[s] + pert[s];
	} else {                                                   /* compositionalRandomWalk Homography.cc:916-926 */
		double B[9], P[9], W[9];
		warp_from_state_dev<SSM>(st, B); warp_from_state_dev<SSM>(pert, P);
		m3_mul_dev(B, P, W);
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double n22 = W[8]; for (int q = 0; q < 9; ++q) W[q] /= n22; }
		state_from_warp_dev<SSM>(ns, W);
	}
	{
		double2 *ps = reinterpret_cast<double2 *>(states + (size_t)k * S), *pa = reinterpret_cast<double2 *>(ars + (size_t)k * S);
#pragma unroll
		for (int s2 = 0; s2 < 4; ++s2)
			if (2 * s2 < S) { ps[s2] = make_double2(ns[2 * s2], ns[2 * s2 + 1]); pa[s2] = make_double2(nar[2 * s2], nar[2 * s2 + 1]); }
	}
}

/* ---- weights -> cumulative weights -> resampling -> the estimate, one workgroup ---- */
struct PfResampleArgs {
	int n, S, ssm;
	int likelihood_func, resampling_type, mean_type;
	double measurement_sigma, max_similarity;
	unsigned long long seed;
	unsigned iter;
	const double *uniforms;       /* [n] or NULL: Philox */
	const double *lik, *sim;      /* [n] AM likelihoods and similarities of the scorer */
	double *wts, *cum;            /* [n] out: particle_wts, normalised particle_cum_wts */
	const double *st_in, *ar_in;  /* current set */
	double *st_out, *ar_out;      /* the other set (resampling) */
	int *ids;                     /* [n] resample ids (diagnostics / tests) */
	double init_corners_hm[12];
	double *out;                  /* [32]: estimate state (8) | max_wt | max_wt_id | mean corners (8) | n_eff */
};
constexpr int kPfBlock = 1024;
__device__ __forceinline__ double block_scan_incl(double v, double *lds /* [kPfBlock / 64 + 1] */, double &total) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	double x = v;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const double y = __shfl_up(x, d); if (lane >= d) x += y; }
	if (lane == 63) lds[wave] = x;
	__syncthreads();
	if (threadIdx.x == 0) {
		double run = 0;
		for (int w = 0; w < kPfBlock / 64; ++w) { const double t = lds[w]; lds[w] = run; run += t; }
		lds[kPfBlock / 64] = run;
	}
	__syncthreads();
	const double r = x + lds[wave];
	total = lds[kPfBlock / 64];
	__syncthreads();
	return r;
}
/* Three launches: (1) one workgroup: weights, their inclusive scan, the normalised cumulative weights, the best particle before
 * resampling; (2) n / 256 workgroups: one particle per thread draws, searches, copies its source particle into the other set and
 * contributes to its workgroup's best / sums (a single workgroup doing all n dependent binary searches was 270 us of the 400 us
 * iteration); (3) one workgroup folds the per-workgroup results into the estimate. */
__global__ __launch_bounds__(kPfBlock) void k_pf_weights(PfResampleArgs a) {
	__shared__ double lds[kPfBlock / 64 + 1];
	__shared__ double red_v[kPfBlock / 64]; __shared__ int red_i[kPfBlock / 64];
	const int n = a.n, tid = threadIdx.x;
	const int per = (n + kPfBlock - 1) / kPfBlock;   /* contiguous run of particles per thread: the scan is a scan of run sums */
	const int lo = min(tid * per, n), hi = min(lo + per, n);
	const double pi = 3.14159265358979323846;
	const double mfac = 1.0 / sqrt(2 * pi * a.measurement_sigma);   /* PF.cc:69-70 */
	/* particle_wts (PF.cc:348-365) and their running sum.  Up to 16 particles per thread (n <= 16 384) the run lives in registers:
	 * every load is issued before the first use and nothing is read back -- the loop form below re-reads wts[] after writing
	 * cum[] through pointers the compiler must assume to alias, one dependent memory round trip per particle (15 of the
	 * kernel's 24 us at 10 000 particles).  Same operations in the same order either way. */
	constexpr int kRun = 16;
	auto weight = [&](double lik, double sim) -> double {
		if (a.likelihood_func == 0) return lik;
		const double val = a.max_similarity - sim;
		return a.likelihood_func == 1 ? mfac * exp(-0.5 * val / a.measurement_sigma) : 1.0 / (1.0 + val);
	};
	double run = 0, total;
	double bv = -1.7976931348623157e308; int bi = -1;
	if (per <= kRun) {
		double wv[kRun];
		const double *src = a.likelihood_func == 0 ? a.lik : a.sim;
		/* a thread's run is contiguous, so a wave's loads / stores are strided (80 bytes apart at 10 000 particles): pairs of
		 * doubles halve the number of requests the single CU this kernel runs on has to issue (lo is even whenever per is) */
		const bool pairs = (per & 1) == 0;
#pragma unroll
		for (int j = 0; j < kRun; j += 2) {
			if (pairs && lo + j + 1 < hi) { const double2 v = *reinterpret_cast<const double2 *>(src + lo + j); wv[j] = v.x; wv[j + 1] = v.y; }
			else { wv[j] = lo + j < hi ? src[lo + j] : 0.0; wv[j + 1] = lo + j + 1 < hi ? src[lo + j + 1] : 0.0; }
		}
#pragma unroll
		for (int j = 0; j < kRun; ++j)
			if (lo + j < hi) { wv[j] = a.likelihood_func == 0 ? wv[j] : weight(0.0, wv[j]); run += wv[j]; }
		const double incl = block_scan_incl(run, lds, total);
		double c = incl - run;
		double cv[kRun];
#pragma unroll
		for (int j = 0; j < kRun; ++j) {
			cv[j] = 0.0;
			if (lo + j < hi) {
				c += wv[j];
				cv[j] = c / total;                                    /* particle_cum_wts /= particle_cum_wts[n - 1] */
				if (wv[j] >= bv) { bv = wv[j]; bi = lo + j; }         /* the highest weighted particle, last index on ties (`>=`, PF.cc:378-381) */
			}
		}
#pragma unroll
		for (int j = 0; j < kRun; j += 2) {
			if (pairs && lo + j + 1 < hi) {
				*reinterpret_cast<double2 *>(a.wts + lo + j) = make_double2(wv[j], wv[j + 1]);
				*reinterpret_cast<double2 *>(a.cum + lo + j) = make_double2(cv[j], cv[j + 1]);
			} else {
				if (lo + j < hi) { a.wts[lo + j] = wv[j]; a.cum[lo + j] = cv[j]; }
				if (lo + j + 1 < hi) { a.wts[lo + j + 1] = wv[j + 1]; a.cum[lo + j + 1] = cv[j + 1]; }
			}
		}
	} else {
		for (int k = lo; k < hi; ++k) {
			const double w = weight(a.likelihood_func == 0 ? a.lik[k] : 0.0, a.likelihood_func == 0 ? 0.0 : a.sim[k]);
			a.wts[k] = w;
			run += w;
		}
		const double incl = block_scan_incl(run, lds, total);
		double c = incl - run;
		for (int k = lo; k < hi; ++k) { c += a.wts[k]; a.cum[k] = c / total; }
		for (int k = lo; k < hi; ++k) if (a.wts[k] >= bv) { bv = a.wts[k]; bi = k; }
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		const double ov = __shfl_xor(bv, d); const int oi = __shfl_xor(bi, d);
		if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; }
	}
	if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
	__syncthreads();
	if (tid == 0) {
		for (int w = 1; w < kPfBlock / 64; ++w) if (red_v[w] > bv || (red_v[w] == bv && red_i[w] > bi)) { bv = red_v[w]; bi = red_i[w]; }
		a.out[8] = bv; a.out[9] = (double)bi;
	}
}
/* per-workgroup partial results of the selection pass: [0] best weight [1] its new index [2..9] sum of states [10..17] sum of corners */
constexpr int kPfPart = 18;
__global__ __launch_bounds__(kBlock) void k_pf_select(PfResampleArgs a, double *parts) {
	__shared__ double red_v[kBlock / 64]; __shared__ int red_i[kBlock / 64];
	__shared__ double lds[4 * 16];
	/* every kCoarse-th (or coarser) cumulative weight: the first levels of each particle's binary search run out of LDS instead of
	 * being fourteen dependent global loads (10 of the kernel's 14.6 us) */
	constexpr int kCoarse = 2048;
	__shared__ double coarse[kCoarse];
	const int n = a.n, S = a.S, tid = threadIdx.x, k = blockIdx.x * kBlock + tid;
	const bool resample = a.resampling_type == 1 || a.resampling_type == 2;
	const int cstride = max(32, (n + kCoarse - 1) / kCoarse), ncoarse = (n + cstride - 1) / cstride;
	if (resample) {
		for (int j = tid; j < ncoarse; j += kBlock) coarse[j] = a.cum[min((j + 1) * cstride, n) - 1];   /* last element of block j */
		__syncthreads();
	}
	double bv = -1.7976931348623157e308; int bi = -1;
	double acc[16];
#pragma unroll
	for (int s = 0; s < 16; ++s) acc[s] = 0.0;
	if (k < n) {
		int id = k;
		if (resample) {
			/* multinomial resampling (PF.cc:455-502 binary search; :505-536 linear search: the same smallest index whose
			 * normalised cumulative weight reaches the draw), into the other particle set */
			const double u = a.uniforms ? a.uniforms[k] : philox_uniform(a.seed, a.iter, (unsigned)k);
			/* the smallest index whose normalised cumulative weight reaches the draw: first the block (its last element reaches
			 * it), then inside the block -- the index the one-level search over cum[] returns */
			int l = 0, h = ncoarse - 1;
			int j = (l + h) / 2;
			while (h > l) { if (coarse[j] >= u) h = j; else l = j + 1; j = (l + h) / 2; }
			l = j * cstride; h = min(l + cstride, n) - 1;
			id = (l + h) / 2;
			while (h > l) { if (a.cum[id] >= u) h = id; else l = id + 1; id = (l + h) / 2; }
			if (a.ids) a.ids[k] = id;
		}
		double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		/* a particle's row is S (6 or 8) contiguous doubles: copied as pairs (rows start 16-byte aligned), all loads first */
		const double2 *src_s = reinterpret_cast<const double2 *>(a.st_in + (size_t)id * S), *src_a = reinterpret_cast<const double2 *>(a.ar_in + (size_t)id * S);
		double2 ps[4], pa[4];
#pragma unroll
		for (int s2 = 0; s2 < 4; ++s2) {
			ps[s2] = 2 * s2 < S ? src_s[s2] : make_double2(0.0, 0.0);
			pa[s2] = (resample && 2 * s2 < S) ? src_a[s2] : make_double2(0.0, 0.0);
		}
#pragma unroll
		for (int s2 = 0; s2 < 4; ++s2) { p[2 * s2] = ps[s2].x; p[2 * s2 + 1] = ps[s2].y; }
		if (resample) {
			double2 *dst_s = reinterpret_cast<double2 *>(a.st_out + (size_t)k * S), *dst_a = reinterpret_cast<double2 *>(a.ar_out + (size_t)k * S);
#pragma unroll
			for (int s2 = 0; s2 < 4; ++s2) if (2 * s2 < S) { dst_s[s2] = ps[s2]; dst_a[s2] = pa[s2]; }
		}
		bv = a.wts[id]; bi = k;
		if (a.mean_type == 1) {
#pragma unroll
			for (int s = 0; s < 8; ++s) acc[s] = p[s];
		} else if (a.mean_type == 2) {   /* updateMeanCorners :607-614 */
			double W[9];
			if (a.ssm == MTFHIP_SSM_HOMOGRAPHY) warp_from_state_dev<MTFHIP_SSM_HOMOGRAPHY>(p, W); else warp_from_state_dev<MTFHIP_SSM_AFFINE>(p, W);
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const double X = a.init_corners_hm[3 * q], Y = a.init_corners_hm[3 * q + 1], Z = a.init_corners_hm[3 * q + 2];
				double nx = W[0] * X + W[1] * Y + W[2] * Z, ny = W[3] * X + W[4] * Y + W[5] * Z;
				if (a.ssm == MTFHIP_SSM_HOMOGRAPHY) { const double d = W[6] * X + W[7] * Y + W[8] * Z; nx = nx / d; ny = ny / d; }
				acc[8 + 2 * q] = nx; acc[9 + 2 * q] = ny;
			}
		}
	}
	/* the best of the (resampled) set, last index on ties (PF.cc:487-490) */
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		const double ov = __shfl_xor(bv, d); const int oi = __shfl_xor(bi, d);
		if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; }
	}
	if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
	__syncthreads();
	double *part = parts + (size_t)blockIdx.x * kPfPart;
	if (tid == 0) {
		for (int w = 1; w < kBlock / 64; ++w) if (red_v[w] > bv || (red_v[w] == bv && red_i[w] > bi)) { bv = red_v[w]; bi = red_i[w]; }
		part[0] = bv; part[1] = (double)bi;
	}
	__syncthreads();
	block_reduce_store<16>(acc, part + 2, lds);
}
/* (pub: the 32 doubles of `out` also go to host-coherent memory, followed by the sequence number the host spins on -- the estimate is
 * what nt::PF reads back every iteration for its convergence test) */
struct PfPublish { double *host; unsigned long long *flag, seq; };
__global__ __launch_bounds__(64) void k_pf_estimate(PfResampleArgs a, const double *parts, int nparts, PfPublish pub) {
	const int lane = threadIdx.x, S = a.S, n = a.n;
	const bool resample = a.resampling_type == 1 || a.resampling_type == 2;
	const double *st_final = resample ? a.st_out : a.st_in;
	/* the per-workgroup rows are fetched by all lanes at once into LDS and folded from there in workgroup order (a loop of
	 * dependent global loads was 12 of this kernel's 16 us) */
	constexpr int kStage = 256;   /* rows staged: 65 536 particles; beyond that the rows are read from memory */
	__shared__ double rows[kStage * kPfPart];
	const bool staged = nparts <= kStage;
	if (staged) {
		for (int q = lane; q < nparts * kPfPart; q += 64) rows[q] = parts[q];
		__syncthreads();
	}
	const double *pr = staged ? rows : parts;
	/* best over the workgroups, in workgroup order (= particle order): last index on ties */
	double bv = a.out[8]; int bi = (int)a.out[9];
	if (resample) {
		bv = -1.7976931348623157e308; bi = -1;
		for (int w = 0; w < nparts; ++w) { const double v = pr[(size_t)w * kPfPart]; const int i2 = (int)pr[(size_t)w * kPfPart + 1]; if (v > bv || (v == bv && i2 > bi)) { bv = v; bi = i2; } }
	}
	if (lane < 16) {
		double s = 0;
		for (int w = 0; w < nparts; ++w) s += pr[(size_t)w * kPfPart + 2 + lane];
		if (a.mean_type == 1 && lane < S) a.out[lane] = s / (double)n;             /* estimateMeanOfSamples :311-317 */
		if (a.mean_type == 2 && lane >= 8) a.out[10 + lane - 8] = s / (double)n;   /* mean corners */
	}
	if (a.mean_type != 1 && lane < S) a.out[lane] = st_final[(size_t)bi * S + lane];
	if (lane == 0) { a.out[8] = bv; a.out[9] = (double)bi; }
	if (pub.host) {
		__threadfence();
		__syncthreads();
		if (lane < 32) __hip_atomic_store(pub.host + lane, ld_coh(a.out + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		__threadfence_system();
		if (lane == 0) __hip_atomic_store(pub.flag, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

/* PF::initializeParticles (PF.cc:185-197): every particle at the current state, AR terms zero */
__global__ void k_pf_fill(int n, int S, const double *state, double *states, double *ars) {
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n) return;
	for (int s = 0; s < S; ++s) { states[(size_t)k * S + s] = state[s]; ars[(size_t)k * S + s] = 0.0; }
}

/* ===================================================================== */
/* launchers                                                              */
/* ===================================================================== */
void launch_pf_propagate(int ssm, const PfLaunch &p, double *states, double *ars, hipStream_t st) {
	PfArgs a;
	a.n = p.n; a.S = p.S; a.dynamic_model = p.dynamic_model; a.update_type = p.update_type; a.corner_based = p.corner_based; a.ar_coeff = p.ar_coeff;
	for (int k = 0; k < 8; ++k) { a.sigma[k] = p.sigma[k]; a.mean[k] = p.mean[k]; a.init_corners[k] = p.init_corners[k]; }
	for (int k = 0; k < 9; ++k) a.sq_inv[k] = p.sq_inv[k];
	a.seed = p.seed; a.iter = p.iter; a.normals = p.normals;
	const dim3 g((p.n + kBlock - 1) / kBlock);
	if (ssm == MTFHIP_SSM_HOMOGRAPHY) MTFHIP_LAUNCH(k_pf_propagate<MTFHIP_SSM_HOMOGRAPHY>, g, dim3(kBlock), 0, st, a, states, ars);
	else MTFHIP_LAUNCH(k_pf_propagate<MTFHIP_SSM_AFFINE>, g, dim3(kBlock), 0, st, a, states, ars);
}
void launch_pf_resample(int ssm, const PfLaunch &p, const double *lik, const double *sim, double *wts, double *cum, const double *st_in,
	const double *ar_in, double *st_out, double *ar_out, int *ids, double *out, double *parts, double *host_out, unsigned long long *host_flag,
	unsigned long long seq, hipStream_t st) {
	PfResampleArgs a;
	a.n = p.n; a.S = p.S; a.ssm = ssm; a.likelihood_func = p.likelihood_func; a.resampling_type = p.resampling_type; a.mean_type = p.mean_type;
	a.measurement_sigma = p.measurement_sigma; a.max_similarity = p.max_similarity; a.seed = p.seed; a.iter = p.iter; a.uniforms = p.uniforms;
	a.lik = lik; a.sim = sim; a.wts = wts; a.cum = cum; a.st_in = st_in; a.ar_in = ar_in; a.st_out = st_out; a.ar_out = ar_out; a.ids = ids;
	for (int k = 0; k < 12; ++k) a.init_corners_hm[k] = p.init_corners_hm[k];
	a.out = out;
	const int nparts = (p.n + kBlock - 1) / kBlock;
	MTFHIP_LAUNCH(k_pf_weights, dim3(1), dim3(kPfBlock), 0, st, a);
	MTFHIP_LAUNCH(k_pf_select, dim3(nparts), dim3(kBlock), 0, st, a, parts);
	MTFHIP_LAUNCH(k_pf_estimate, dim3(1), dim3(64), 0, st, a, parts, nparts, PfPublish{host_out, host_flag, seq});
}
void launch_pf_fill(int n, int S, const double *dev_state, double *states, double *ars, hipStream_t st) {
	MTFHIP_LAUNCH(k_pf_fill, dim3((n + 255) / 256), dim3(256), 0, st, n, S, dev_state, states, ars);
}

} // namespace mtfhip
