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

template <int SSM, bool MC>
static void launch_nn_ssm(const BatchView &bv, const ImgView &im, const NnArgs &a, int count, double *feat, hipStream_t st) {
	const dim3 g((unsigned)count), blk(kBlock);
	if (bv.am == MTFHIP_AM_NCC) MTFHIP_LAUNCH((k_nn_dataset<SSM, MTFHIP_AM_NCC, MC>), g, blk, 0, st, bv, im, a, feat);
	else if (bv.am == MTFHIP_AM_MI) MTFHIP_LAUNCH((k_nn_dataset<SSM, MTFHIP_AM_MI, MC>), g, blk, 0, st, bv, im, a, feat);
	else MTFHIP_LAUNCH((k_nn_dataset<SSM, MTFHIP_AM_SSD, MC>), g, blk, 0, st, bv, im, a, feat);
}
/* rows [a.row_lo, a.row_lo + count) of the dataset into feat[count][F] */
void launch_nn_dataset(const BatchView &bv, const ImgView &im, const NnArgs &a, int count, double *feat, hipStream_t st) {
	if (count <= 0) return;
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY, mc = bv.C > 1;
	if (hom && mc) launch_nn_ssm<MTFHIP_SSM_HOMOGRAPHY, true>(bv, im, a, count, feat, st);
	else if (hom) launch_nn_ssm<MTFHIP_SSM_HOMOGRAPHY, false>(bv, im, a, count, feat, st);
	else if (mc) launch_nn_ssm<MTFHIP_SSM_AFFINE, true>(bv, im, a, count, feat, st);
	else launch_nn_ssm<MTFHIP_SSM_AFFINE, false>(bv, im, a, count, feat, st);
}

} // namespace mtfhip
