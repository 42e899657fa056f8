/*
 * kernels_nn.hip -- nt::NN's dataset generation on the device (SM/src/NT/NN.cc:131-191, the second batch axis north_star names):
 * per sample
 *     ssm->generatePerturbation(p)            ProjectiveBase.cc:283-288: independent N(mean_k, sigma_k) per state component
 *     ssm->invertState(inv, p)                Homography.cc:109-114 / Affine.cc:145-150
 *     ssm->compositionalUpdate(inv)           Homography.cc:73-92 / Affine.cc:87-109
 *     am->updatePixVals(ssm->getPts())        ImageBase.cc:268-290
 *     am->updateDistFeat(row)                 SSD: the patch (SSDBase.h:116-125); NCC: centred, unit norm (NCC.cc:530-537);
 *                                             MI: floor(It) | four cubic B-spline weights, 5 x N (MI.cc:736-747)
 *     ssm->compositionalUpdate(p)             back to where it was
 * in ONE launch: one workgroup per sample draws the perturbation (Philox4x32-10 + Box-Muller keyed by (seed, sample): a pure function of the
 * sample's index, so the ranks of a sharded run agree without an exchange), forms curr_warp * inverse(W(p)) in registers, samples the
 * patch there and streams the feature row out with non-temporal stores.  Nothing is read per pixel but the template grid (L2) and the
 * texels: the kernel is bound by the 8 N C bytes it writes (SSD / NCC; MI 40 N C).
 * (The reference's SSM walks W <- W inv(P) then W <- W P per sample, so that its warp is the identity only up to the rounding of that
 * product; here every sample starts from the SSM's warp itself: differences of a few 1e-16 in the sampled coordinates.)
 * One of the translation units of libmtfhip.so.
 */
#include "mtfhip_device.h"
#include "mtfhip_rng_device.h"
#include <type_traits>
#include <map>
#include <tuple>
#include <mutex>

namespace mtfhip {

/* utils::bSpl3 (Utilities/include/mtf/Utilities/histUtils.h:161-175) */
__device__ __forceinline__ double bspl3_ref(double x) {
	if ((x > -2) && (x <= -1)) { const double t = 2 + x; return (t * t * t) / 6; }
	if ((x > -1) && (x <= 0)) return (4 - 3 * x * x * (2 + x)) / 6;
	if ((x > 0) && (x <= 1)) return (4 - 3 * x * x * (2 - x)) / 6;
	if ((x > 1) && (x < 2)) { const double t = 2 - x; return (t * t * t) / 6; }
	return 0;
}

template <int K>
__device__ __forceinline__ void nn_allsum(double *v, double *lds /* [4][K] */) {
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < K; ++k) v[k] = wave_sum_dpp(v[k]);
	__syncthreads();
	if (lane == 0) {
#pragma unroll
		for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
	}
	__syncthreads();
#pragma unroll
	for (int k = 0; k < K; ++k) v[k] = (lds[k] + lds[K + k]) + (lds[2 * K + k] + lds[3 * K + k]);
}

constexpr int kNnKeep = 16;   /* values a thread keeps in registers between NCC's passes: rows up to 16 x 256 = 4096 entries (larger: re-read from the row) */

template <int SSM, int AM, bool MC>
__global__ __launch_bounds__(kBlock) void k_nn_dataset(BatchView bv, ImgView im, NnArgs a, double *feat) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	__shared__ double sP[8], red[4];
	const int local = blockIdx.x;
	const unsigned g = (unsigned)(a.row_lo + local);   /* the sample's global index: what its draws are keyed by */
	const int tid = threadIdx.x;
	/* the perturbation: given, or drawn -- lane q < S / 2 draws the pair (2 q, 2 q + 1) */
	if (tid < 8) {
		double v = 0.0;
		if (tid < S) {
			if (a.perts_in) v = a.perts_in[(size_t)g * S + tid];
			else {
				double z0, z1;
				philox_normal2(a.seed, 0x4E4E4453u /* "NNDS" */, g, (unsigned)(tid >> 1), z0, z1);
				v = a.mean[tid] + a.sigma[tid] * ((tid & 1) ? z1 : z0);
			}
			if (a.perts_out) a.perts_out[(size_t)g * S + tid] = v;
		}
		sP[tid] = v;
	}
	__syncthreads();
	/* every thread: W = curr_warp * inverse(W(p)), normalised as invertState and compositionalUpdate do (the same expressions as the host's
	 * mtfhip_ssm_invert_state / compose) */
	double p[8], P[9], Pi[9], W[9];
#pragma unroll
	for (int q = 0; q < 8; ++q) p[q] = sP[q];
	warp_from_state_dev<SSM>(p, P);
	m3_inv_dev(P, Pi);
	{
		const double n22 = Pi[8];
#pragma unroll
		for (int q = 0; q < 9; ++q) Pi[q] /= n22;
	}
	if constexpr (SSM == MTFHIP_SSM_AFFINE) { Pi[6] = 0; Pi[7] = 0; Pi[8] = 1; }   /* (getStateFromWarp -> getWarpFromState drops the last row's rounding, Affine.cc:132-143) */
	m3_mul_dev(a.base, Pi, W);
	if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
		const double n22 = W[8];
#pragma unroll
		for (int q = 0; q < 9; ++q) W[q] /= n22;
	}
	const int N = bv.N;                       /* rows of the per-pixel arrays: (pixel, channel) pairs */
	const int Cc = MC ? bv.C : 1;
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[bv.unit_z ? MTFHIP_BUF_INIT_PTS : MTFHIP_BUF_INIT_HXY]);
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z];
	const int F = AM == MTFHIP_AM_MI ? 5 * N : N;
	double *out = feat + (size_t)local * F;
	auto sample = [&](int i) -> double {
		const int pi = MC ? i / Cc : i;
		const double2 q = ip[pi];
		const double z = bv.unit_z ? 1.0 : iz[pi];
		double wx, wy;
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
			const double cx = W[0] * q.x + W[1] * q.y + W[2] * z, cy = W[3] * q.x + W[4] * q.y + W[5] * z;
			const double d = W[6] * q.x + W[7] * q.y + W[8] * z;
			wx = cx / d; wy = cy / d;
		} else {
			wx = W[0] * q.x + W[1] * q.y + W[2] * z; wy = W[3] * q.x + W[4] * q.y + W[5] * z;
		}
		const double pv = MC ? pix_val_mc(im, wx, wy, i - pi * Cc) : pix_val(im, wx, wy);
		return a.norm_mult * pv + a.norm_add;
	};
	if constexpr (AM == MTFHIP_AM_SSD) {
		for (int i = tid; i < N; i += kBlock) MAT_STORE(out + i, sample(i));
	} else if constexpr (AM == MTFHIP_AM_NCC) {
		/* NCC::updateDistFeat NCC.cc:530-537: It - mean(It), over its norm */
		double keep[kNnKeep];
		double s1[1] = {0.0};
#pragma unroll
		for (int k = 0; k < kNnKeep; ++k) {
			const int i = tid + k * kBlock;
			keep[k] = i < N ? sample(i) : 0.0;
			s1[0] += keep[k];
		}
		for (int i = tid + kNnKeep * kBlock; i < N; i += kBlock) { const double v = sample(i); out[i] = v; s1[0] += v; }
		nn_allsum<1>(s1, red);
		const double mean = s1[0] / (double)N;
		double s2[1] = {0.0};
#pragma unroll
		for (int k = 0; k < kNnKeep; ++k) { const int i = tid + k * kBlock; const double d = i < N ? keep[k] - mean : 0.0; s2[0] = fma(d, d, s2[0]); }
		for (int i = tid + kNnKeep * kBlock; i < N; i += kBlock) { const double d = out[i] - mean; s2[0] = fma(d, d, s2[0]); }
		nn_allsum<1>(s2, red);
		const double sd = sqrt(s2[0]);
#pragma unroll
		for (int k = 0; k < kNnKeep; ++k) { const int i = tid + k * kBlock; if (i < N) MAT_STORE(out + i, (keep[k] - mean) / sd); }
		for (int i = tid + kNnKeep * kBlock; i < N; i += kBlock) out[i] = (out[i] - mean) / sd;
	} else {
		/* MI::updateDistFeat MI.cc:736-747: row-major 5 x N -- floor(It) | bSpl3(d), bSpl3(d + 1), bSpl3(d + 2), bSpl3(d + 3), d = std_bspl_ids(floor, 0) - It */
		for (int i = tid; i < N; i += kBlock) {
			const double v = sample(i);
			const int fl = (int)v;
			double d = (double)(fl - 1 > 0 ? fl - 1 : 0) - v;
			MAT_STORE(out + i, (double)fl);
			MAT_STORE(out + N + i, bspl3_ref(d)); d += 1;
			MAT_STORE(out + 2 * (size_t)N + i, bspl3_ref(d)); d += 1;
			MAT_STORE(out + 3 * (size_t)N + i, bspl3_ref(d)); d += 1;
			MAT_STORE(out + 4 * (size_t)N + i, bspl3_ref(d));
		}
	}
}


/* ---- tolerance mode (MTFHIP_MATH_FAST, the batch default), single channel: warps first, then rows ----
 * The workgroup-per-sample form above spends its time outside the pixels: every thread of its four waves forms the sample's warp
 * (nineteen IEEE divisions: ~800 FP64 instructions per wave, as many as its ten pixels cost), one pixel is two dependent memory round
 * trips, and NCC's reductions cross the workgroup twice (r06: 146 us per 10 000 x 2500 px = 0.17 of the 200 MB it writes at 8 TB/s).
 * Here, as in the particle filter (k_pf_propose / k_pf_score):
 *   k_nn_warps  one THREAD per sample draws (or takes) the perturbation and forms W = curr_warp * inverse(W(p)) with the expressions of
 *               the workgroup form -- the same bits -- plus the flag "the sample's warped hull is inside the frame" (its samples then
 *               skip the border test, k_pf_score's shortcut): 10 doubles per sample into a scratch array.  A sample's ~800 instructions
 *               are issued once per 64 samples instead of once per wave of the sample.
 *   k_nn_rows   persistent workgroups, one sample at a time, a quarter of the row per wave; W arrives through scalar loads; the pixels are sampled with the candidate scorer's
 *               arithmetic (one reciprocal per homography point, factored interpolant, paired texel loads), four at a time (grid
 *               points requested together, then texel pairs together), a lane owns pairs of neighbouring entries and stores them as 16 bytes (8-byte accesses run
 *               at 0.54-0.70 of the 16-byte rate, MI355X_MICROARCH.md); NCC keeps its entries in registers across the two workgroup sums.
 * Differences against the workgroup form: ~1e-13 of a pixel value (tests: 1e-9). */
constexpr int kNnRowKeep = 12;   /* NCC: entries a lane keeps: rows of up to 4 waves x 64 lanes x 12 = 3072 entries (larger rows take the workgroup form) */
constexpr int kNnWarpStride = 10;   /* W (9, row-major) | inside flag */
struct __attribute__((packed, aligned(4))) NnTexPair { float a, b; };
typedef double nn_d2 __attribute__((ext_vector_type(2)));

template <int SSM>
__global__ __launch_bounds__(kBlock) void k_nn_warps(NnArgs a, int count, double *warps, int img_w, int img_h, int hull_ok, double hx0, double hy0,
	double hx1, double hy1, double hx2, double hy2, double hx3, double hy3) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	const int local = blockIdx.x * kBlock + threadIdx.x;
	if (local >= count) return;
	const unsigned g = (unsigned)(a.row_lo + local);
	double p[8];
#pragma unroll
	for (int q = 0; q < 8; ++q) p[q] = 0.0;
	if (a.perts_in) {
#pragma unroll
		for (int q = 0; q < S; ++q) p[q] = a.perts_in[(size_t)g * S + q];
	} else {
#pragma unroll
		for (int q = 0; q < S; q += 2) {   /* pair (q, q + 1) from one Philox block, as the workgroup form draws them */
			double z0, z1;
			philox_normal2(a.seed, 0x4E4E4453u /* "NNDS" */, g, (unsigned)(q >> 1), z0, z1);
			p[q] = a.mean[q] + a.sigma[q] * z0; p[q + 1] = a.mean[q + 1] + a.sigma[q + 1] * z1;
		}
	}
	if (a.perts_out) {
#pragma unroll
		for (int q = 0; q < S; ++q) a.perts_out[(size_t)g * S + q] = p[q];
	}
	double P[9], Pi[9], W[9];
	warp_from_state_dev<SSM>(p, P);
	m3_inv_dev(P, Pi);
	{
		const double n22 = Pi[8];
#pragma unroll
		for (int q = 0; q < 9; ++q) Pi[q] /= n22;
	}
	if constexpr (SSM == MTFHIP_SSM_AFFINE) { Pi[6] = 0; Pi[7] = 0; Pi[8] = 1; }
	m3_mul_dev(a.base, Pi, W);
	if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
		const double n22 = W[8];
#pragma unroll
		for (int q = 0; q < 9; ++q) W[q] /= n22;
	}
	/* the sample's hull: the template grid's own corners through W (a margin of a thousandth of a pixel: the samples' own rounding is ~1e-13) */
	bool in = hull_ok != 0;
	const double hxs[4] = {hx0, hx1, hx2, hx3}, hys[4] = {hy0, hy1, hy2, hy3};
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		double hx = fma(W[0], hxs[c], fma(W[1], hys[c], W[2])), hy = fma(W[3], hxs[c], fma(W[4], hys[c], W[5]));
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
			const double dd = fma(W[6], hxs[c], fma(W[7], hys[c], W[8]));
			in = in & (dd > 1e-9);
			const double inv = rcp_fast(dd);
			hx *= inv; hy *= inv;
		}
		in = in & (hx > 1e-3) & (hy > 1e-3) & (hx < (double)(img_w - 1) - 1e-3) & (hy < (double)(img_h - 1) - 1e-3);
	}
	double *w = warps + (size_t)local * kNnWarpStride;
#pragma unroll
	for (int q = 0; q < 9; ++q) w[q] = W[q];
	w[9] = in ? 1.0 : 0.0;
}

template <int SSM, int AM>
__global__ __launch_bounds__(kBlock) void k_nn_rows(BatchView bv, ImgView im, const double *warps, int count, double norm_mult, double norm_add, double *feat) {
	__shared__ double red[16];
	extern __shared__ double2 nn_lds[];                          /* the chunk's grid points (x, y), then -- grids that are not unit-z -- their z */
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int N = bv.N;
	const bool uz = bv.unit_z != 0;
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[uz ? MTFHIP_BUF_INIT_PTS : MTFHIP_BUF_INIT_HXY]);
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z];
	const int F = AM == MTFHIP_AM_MI ? 5 * N : N;
	const float *img = im.data;
	const int iw1 = im.w - 1, ih1 = im.h - 1, stride = im.stride;
	const bool wide = (N & 1) == 0;                              /* (uniform) rows start on 16-byte boundaries and pairs do not straddle a row's end */
	constexpr int EM = kNnRowKeep;                               /* entries of a row a lane holds at a time */
	constexpr int kChunk = 4 * 64 * EM;                          /* entries of a row the workgroup holds at a time */
	const int R = (N + 127) / 128;                               /* the row in pair-rounds of 128 entries (lane l: entries 128 r + 2 l, 128 r + 2 l + 1) */
	const int lds_entries = 128 * 4 * (((R < kChunk / 128 ? R : kChunk / 128) + 3) / 4);   /* what the launcher sized the LDS for: the largest chunk of the row */
	/* The workgroups are persistent (the launch fills the device once; a sample is a quarter row per wave, so 10 000 samples do not leave
	 * the last fifth of the resident slots one wave each).  Rows longer than 3072 entries are walked chunk by chunk (SSD / MI: a row's
	 * entries are independent). */
	for (int c0 = 0; c0 < R; c0 += kChunk / 128) {
		const int Rc = min(R - c0, kChunk / 128), R4 = (Rc + 3) / 4, E = 2 * R4;   /* wave w takes pair-rounds c0 + [w R4, (w + 1) R4) of the chunk */
		auto pix = [&](int k) { return 128 * (c0 + wave * R4 + (k >> 1)) + 2 * lane + (k & 1); };
		/* The chunk's grid points go to LDS once per workgroup and are read from there by every sample (r06 PMC: 3.6 vector-memory reads per
		 * wave-pixel with the 16-byte grid load of every pixel of every sample among them; through LDS 573 -> 542 us per 100 000 samples.
		 * Keeping them in registers instead: 169 VGPRs, two workgroups per CU, 639 us.  Forming the warp inside this kernel for launches of at
		 * most one sample per resident workgroup instead of a k_nn_warps launch in front: 20.8 -> 27.2 us per 1000 samples, not kept.) */
		double *lz = reinterpret_cast<double *>(nn_lds + lds_entries);
		if (c0 > 0) __syncthreads();
		for (int j = threadIdx.x; j < 128 * 4 * R4; j += kBlock) {   /* (every entry a wave may touch: past the row's end, copies of its last point) */
			const int i = min(128 * c0 + j, N - 1);
			nn_lds[j] = ip[i];
			if (!uz) lz[j] = iz[i];
		}
		__syncthreads();
		int round = 0;
		for (int local = blockIdx.x; local < count; local += gridDim.x, ++round) {
			const double *wp = warps + (size_t)local * kNnWarpStride;   /* uniform: scalar loads */
			double W[9];
#pragma unroll
			for (int t = 0; t < 9; ++t) W[t] = wp[t];
			const bool all_inside = wp[9] != 0.0;
			double *out = feat + (size_t)local * F;
			auto sample_batch = [&](auto kbase, auto ucount, double *v, auto inside_tag) {
				constexpr bool INSIDE = decltype(inside_tag)::value;
				constexpr int U = decltype(ucount)::value, K0 = decltype(kbase)::value;
				double2 q[U]; double z[U];
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const int j = pix(K0 + u) - 128 * c0;    /* (entries past the row's end hold copies of its last point) */
					q[u] = nn_lds[j];
					z[u] = uz ? 1.0 : lz[j];
				}
				double wx[U], wy[U];
				bool ok = true;
#pragma unroll
				for (int u = 0; u < U; ++u) {
					const double2 qq = q[u];
					if (uz) {   /* (uniform) z is exactly 1.0 */
						wx[u] = fma(W[0], qq.x, fma(W[1], qq.y, W[2])); wy[u] = fma(W[3], qq.x, fma(W[4], qq.y, W[5]));
						if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double inv = rcp_fast(fma(W[6], qq.x, fma(W[7], qq.y, W[8]))); wx[u] *= inv; wy[u] *= inv; }
					} else {
						const double zz = z[u];
						wx[u] = fma(W[0], qq.x, fma(W[1], qq.y, W[2] * zz)); wy[u] = fma(W[3], qq.x, fma(W[4], qq.y, W[5] * zz));
						if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double inv = rcp_fast(fma(W[6], qq.x, fma(W[7], qq.y, W[8] * zz))); wx[u] *= inv; wy[u] *= inv; }
					}
					if constexpr (!INSIDE) ok = ok & (wx[u] >= 0) & (wy[u] >= 0) & ((int)wx[u] < iw1) & ((int)wy[u] < ih1);
				}
				if (INSIDE || __builtin_amdgcn_ballot_w64(!ok) == 0) {
					NnTexPair t0[U], t1[U];
#pragma unroll
					for (int u = 0; u < U; ++u) {
						const unsigned off = (__umul24((unsigned)(int)wy[u], (unsigned)stride) + (unsigned)(int)wx[u]) * 4u;
						t0[u] = ld_off<NnTexPair>(img, off); t1[u] = ld_off<NnTexPair>(img + stride, off);
					}
#pragma unroll
					for (int u = 0; u < U; ++u)
						v[u] = fma(norm_mult, bilin_val_fast(t0[u].a, t0[u].b, t1[u].a, t1[u].b, __builtin_amdgcn_fract(wx[u]), __builtin_amdgcn_fract(wy[u])), norm_add);
				} else {
#pragma unroll
					for (int u = 0; u < U; ++u) v[u] = fma(norm_mult, pix_val_fast(im, wx[u], wy[u]), norm_add);
				}
			};
			/* entries k, k + 1 (a lane's pair) of the row that starts byte_off into the sample's block */
			auto store_pair = [&](unsigned byte_off, int k, double v0, double v1) {
				const int i = pix(k);
				if (wide) { if (i < N) { nn_d2 pr; pr.x = v0; pr.y = v1; st_off<nn_d2>(out, byte_off + (unsigned)i * 8u, pr); } }
				else {
					if (i < N) st_off<double>(out, byte_off + (unsigned)i * 8u, v0);
					if (i + 1 < N) st_off<double>(out, byte_off + (unsigned)i * 8u + 8u, v1);
				}
			};
			/* MI::updateDistFeat MI.cc:736-747: row-major 5 x N -- floor(It) | bSpl3(d), bSpl3(d + 1), bSpl3(d + 2), bSpl3(d + 3), d = std_bspl_ids(floor, 0) - It */
			auto mi_pair = [&](int k, double v0, double v1) {
				const unsigned rowb = (unsigned)N * 8u;
				double r[2][5];
#pragma unroll
				for (int e = 0; e < 2; ++e) {
					const double v = e ? v1 : v0;
					const int fl = (int)v;
					double d = (double)(fl - 1 > 0 ? fl - 1 : 0) - v;
					r[e][0] = (double)fl;
					r[e][1] = bspl3_ref(d); d += 1;
					r[e][2] = bspl3_ref(d); d += 1;
					r[e][3] = bspl3_ref(d); d += 1;
					r[e][4] = bspl3_ref(d);
				}
#pragma unroll
				for (int w = 0; w < 5; ++w) store_pair((unsigned)w * rowb, k, r[0][w], r[1][w]);
			};
			auto body = [&](auto inside_tag) {
				double keep[EM];
				/* EM = 12 entries in three batches of four; a batch the row does not reach is skipped (uniform), a half batch takes two */
#define MTFHIP_NN_BATCH(K0) \
				if (K0 + 4 <= E) sample_batch(std::integral_constant<int, K0>{}, std::integral_constant<int, 4>{}, keep + K0, inside_tag); \
				else if (K0 + 2 <= E) { sample_batch(std::integral_constant<int, K0>{}, std::integral_constant<int, 2>{}, keep + K0, inside_tag); keep[K0 + 2] = 0.0; keep[K0 + 3] = 0.0; } \
				else { keep[K0] = 0.0; keep[K0 + 1] = 0.0; keep[K0 + 2] = 0.0; keep[K0 + 3] = 0.0; }
				if constexpr (AM == MTFHIP_AM_NCC) {
					/* NCC::updateDistFeat NCC.cc:530-537: It - mean(It), over its norm (one chunk: the launcher sends longer rows to the workgroup form) */
					MTFHIP_NN_BATCH(0) MTFHIP_NN_BATCH(4) MTFHIP_NN_BATCH(8)
					double s1 = 0.0;
#pragma unroll
					for (int k = 0; k < EM; ++k) { keep[k] = (k < E && pix(k) < N) ? keep[k] : 0.0; s1 += keep[k]; }
					double *rd = red + 8 * (round & 1);   /* (alternating: a wave may start the next sample while another still reads this one's sums) */
					s1 = wave_sum_dpp(s1);
					if (lane == 0) rd[wave] = s1;
					__syncthreads();
					const double mean = ((rd[0] + rd[1]) + (rd[2] + rd[3])) / (double)N;
					double s2 = 0.0;
#pragma unroll
					for (int k = 0; k < EM; ++k) { const double d = (k < E && pix(k) < N) ? keep[k] - mean : 0.0; keep[k] = d; s2 = fma(d, d, s2); }
					s2 = wave_sum_dpp(s2);
					if (lane == 0) rd[4 + wave] = s2;
					__syncthreads();
					const double inv_sd = rsq_fast((rd[4] + rd[5]) + (rd[6] + rd[7]));
#pragma unroll
					for (int k = 0; k < EM; k += 2) if (k < E) store_pair(0u, k, keep[k] * inv_sd, keep[k + 1] * inv_sd);
				} else {
#define MTFHIP_NN_OUT(K0) \
					if (K0 + 2 <= E) { if constexpr (AM == MTFHIP_AM_SSD) store_pair(0u, K0, keep[K0], keep[K0 + 1]); else mi_pair(K0, keep[K0], keep[K0 + 1]); } \
					if (K0 + 4 <= E) { if constexpr (AM == MTFHIP_AM_SSD) store_pair(0u, K0 + 2, keep[K0 + 2], keep[K0 + 3]); else mi_pair(K0 + 2, keep[K0 + 2], keep[K0 + 3]); }
					MTFHIP_NN_BATCH(0) MTFHIP_NN_OUT(0)
					MTFHIP_NN_BATCH(4) MTFHIP_NN_OUT(4)
					MTFHIP_NN_BATCH(8) MTFHIP_NN_OUT(8)
#undef MTFHIP_NN_OUT
				}
#undef MTFHIP_NN_BATCH
			};
			if (all_inside) body(std::true_type{}); else body(std::false_type{});
		}
	}
}

template <int SSM, bool MC>
static void launch_nn_ssm(const BatchView &bv, const ImgView &im, const NnArgs &a, int count, double *feat, hipStream_t st) {
	const dim3 g((unsigned)count), blk(kBlock);
	if (bv.am == MTFHIP_AM_NCC) MTFHIP_LAUNCH((k_nn_dataset<SSM, MTFHIP_AM_NCC, MC>), g, blk, 0, st, bv, im, a, feat);
	else if (bv.am == MTFHIP_AM_MI) MTFHIP_LAUNCH((k_nn_dataset<SSM, MTFHIP_AM_MI, MC>), g, blk, 0, st, bv, im, a, feat);
	else MTFHIP_LAUNCH((k_nn_dataset<SSM, MTFHIP_AM_SSD, MC>), g, blk, 0, st, bv, im, a, feat);
}
template <int SSM>
static void launch_nn_rows(const BatchView &bv, const ImgView &im, const NnArgs &a, int count, double *feat, const double *hull, double *warps, hipStream_t st) {
	const int ok = hull ? 1 : 0;
	const double h[8] = {hull ? hull[0] : 0, hull ? hull[1] : 0, hull ? hull[2] : 0, hull ? hull[3] : 0, hull ? hull[4] : 0, hull ? hull[5] : 0, hull ? hull[6] : 0, hull ? hull[7] : 0};
	MTFHIP_LAUNCH((k_nn_warps<SSM>), dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a, count, warps, im.w, im.h, ok, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
	/* persistent workgroups: as many as the device holds at once (occupancy x compute units), at most one per sample */
	/* dynamic LDS: the grid points of the chunk the row needs (a 50 x 50 row: 20 pair-rounds = 40 KB, four workgroups per CU; the full 3072-entry
	 * chunk is 48 KB: three), z behind them for grids that are not unit-z.  The resident count is a property of (kernel, LDS bytes): cached per pair. */
	constexpr int kChunkEntries = 4 * 64 * kNnRowKeep;
	const int rounds = (bv.N + 127) / 128, r4 = ((rounds < kChunkEntries / 128 ? rounds : kChunkEntries / 128) + 3) / 4;
	const int lds_entries = 128 * 4 * r4;
	const size_t lds = (size_t)lds_entries * (bv.unit_z ? 16 : 24);
	static std::map<std::tuple<int, int, size_t>, int> resident_cache;
	static std::mutex resident_mutex;   /* (contexts on several host threads: the loopback ranks of the tests) */
	std::lock_guard<std::mutex> resident_lock(resident_mutex);
	const int ai = bv.am == MTFHIP_AM_NCC ? 1 : (bv.am == MTFHIP_AM_MI ? 2 : 0), si = SSM == MTFHIP_SSM_HOMOGRAPHY ? 0 : 1;
	int &resident = resident_cache[std::make_tuple(si, ai, lds)];
	if (!resident) {
		int per_cu = 0, dev = 0;
		hipDeviceProp_t prop;
		hipError_t e = hipGetDevice(&dev);
		if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
		if (e == hipSuccess) {
			if (ai == 1) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nn_rows<SSM, MTFHIP_AM_NCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_nn_rows<SSM, MTFHIP_AM_NCC>, kBlock, lds); }
			else if (ai == 2) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nn_rows<SSM, MTFHIP_AM_MI>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_nn_rows<SSM, MTFHIP_AM_MI>, kBlock, lds); }
			else { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_nn_rows<SSM, MTFHIP_AM_SSD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_nn_rows<SSM, MTFHIP_AM_SSD>, kBlock, lds); }
		}
		resident = (e == hipSuccess && per_cu > 0) ? per_cu * prop.multiProcessorCount : 1024;
	}
	const dim3 g((unsigned)(count < resident ? count : resident)), blk(kBlock);
#define MTFHIP_NN_ROWS(A) MTFHIP_LAUNCH((k_nn_rows<SSM, A>), g, blk, lds, st, bv, im, warps, count, a.norm_mult, a.norm_add, feat)
	if (bv.am == MTFHIP_AM_NCC) MTFHIP_NN_ROWS(MTFHIP_AM_NCC);
	else if (bv.am == MTFHIP_AM_MI) MTFHIP_NN_ROWS(MTFHIP_AM_MI);
	else MTFHIP_NN_ROWS(MTFHIP_AM_SSD);
#undef MTFHIP_NN_ROWS
}
/* the scratch the two-launch form needs: kNnWarpStride doubles per sample of the launch */
size_t nn_warps_bytes(int count) { return sizeof(double) * (size_t)kNnWarpStride * (size_t)(count > 0 ? count : 0); }
bool nn_two_launch_ok(const BatchView &bv, const ImgView &im, int fast_math) {
	return fast_math && bv.C == 1 && !(bv.am == MTFHIP_AM_NCC && bv.N > 4 * 64 * kNnRowKeep) && im.w >= 2 && im.h >= 2;
}
/* rows [a.row_lo, a.row_lo + count) of the dataset into feat[count][F].  warps != NULL (nn_two_launch_ok: the batch's MTFHIP_MATH_FAST, one
 * channel, NCC rows of at most 3072 entries; nn_warps_bytes(count) of scratch): k_nn_warps + k_nn_rows; otherwise the workgroup-per-sample form
 * in the reference's operation order.  hull: the template grid's own corners x0 y0 .. x3 y3 when the grid is a unit-z lattice laid out
 * inside them, else NULL */
void launch_nn_dataset(const BatchView &bv, const ImgView &im, const NnArgs &a, int count, double *feat, double *warps, const double *hull, hipStream_t st) {
	if (count <= 0) return;
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY, mc = bv.C > 1;
	if (warps) {
		if (hom) launch_nn_rows<MTFHIP_SSM_HOMOGRAPHY>(bv, im, a, count, feat, hull, warps, st);
		else launch_nn_rows<MTFHIP_SSM_AFFINE>(bv, im, a, count, feat, hull, warps, st);
		return;
	}
	if (hom && mc) launch_nn_ssm<MTFHIP_SSM_HOMOGRAPHY, true>(bv, im, a, count, feat, st);
	else if (hom) launch_nn_ssm<MTFHIP_SSM_HOMOGRAPHY, false>(bv, im, a, count, feat, st);
	else if (mc) launch_nn_ssm<MTFHIP_SSM_AFFINE, true>(bv, im, a, count, feat, st);
	else launch_nn_ssm<MTFHIP_SSM_AFFINE, false>(bv, im, a, count, feat, st);
}

} // namespace mtfhip
