/*
 * kernels_pf.hip -- one iteration of the particle filter on the device in three launches: scoring + particle weight (k_pf_score),
 * cumulative weights (k_pf_scan), multinomial resampling + estimate + the next iteration's proposals (k_pf_select).
 * (one of the translation units of libmtfhip.so)
 *
 * Reference: nt::PF::update SM/src/NT/PF.cc:207-447, binaryMultinomialResampling :455-502, linearMultinomialResampling
 * :505-536, updateMeanCorners :607-614; Homography::generatePerturbation / compositionalRandomWalk /
 * compositionalAutoRegression1 SSM/src/Homography.cc:899-942; ProjectiveBase::additiveRandomWalk / additiveAutoRegression1 /
 * compositionalRandomWalk / compositionalAutoRegression1 / generatePerturbation / estimateMeanOfSamples
 * SSM/src/ProjectiveBase.cc:236-317; Affine::generatePerturbation / geomToState SSM/src/Affine.cc:380-410,464-503.
 *
 * In the reference every particle of every iteration pays a 4-corner DLT through an 8 x 9 JacobiSVD
 * (hom_corner_based_sampling is on by default, parameters.h:262) on one host core; here the corner perturbation is the
 * closed-form square-to-quadrilateral map composed with the inverse of the template's.
 * Random draws: the reference seeds boost::mt11213b from random_device (not reproducible), so the draws are an INPUT here --
 * either arrays of standard normals / uniforms handed in by the caller (parity tests, reproducible runs), or a counter-based
 * Philox4x32-10 generator + Box-Muller on the device keyed by (seed, iteration, particle).
 *
 * The shape of an iteration (r03).  A particle's proposal is a pure function of (its resampled state, the draws of its index
 * and iteration).  With the device generator the draws of iteration t + 1 are known at iteration t, so k_pf_select -- one thread
 * per particle, all lanes busy -- follows "copy the selected particle" directly with "propose its successor" and leaves the
 * proposal set of the NEXT iteration behind; k_pf_score then finds its four candidates' states ready (scalar loads, as
 * k_score_candidates_fast) and maps the similarity to the particle weight in its epilogue -- no propagation launch and no
 * likelihood / similarity vectors in between.  (When the draws are handed in by the caller, after initialize / setRegion /
 * set_particles / setSampler, and with MeanType::Corners, whose setCorners moves the sampler's reference corners every
 * iteration, the proposals come from a launch of their own, k_pf_propose.)  On R ranks each rank scores the contiguous block
 * [r m, (r + 1) m), m = ceil(n / R), and writes its weights at their global positions, so that ONE in-place all-gather of m
 * doubles per rank leaves the flat weight vector on every rank (SM/src/PF.cc:262-277) -- no second vector, no unscrambling
 * copies; proposals and resampling are replicated (identical draws and weights everywhere: re-evaluating a proposal costs less
 * than moving its 128 bytes over xGMI).  k_pf_scan turns the weights into chunk-local running sums (256 particles per
 * wave) and the last workgroup to arrive scans the chunk totals; k_pf_select draws, finds the source particle with a
 * three-level search (chunk table in LDS, the chunk's sixteen sub-block sums, the sub-block's sixteen particles: two rounds of
 * independent loads, four cache lines), writes the resampled set, and its last workgroup folds the per-workgroup rows into the estimate and hands it to the
 * host.  Every sum is taken in a fixed order that depends on n only: all ranks of a sharded filter resample identically.
 */
#include "mtfhip_device.h"
#include "mtfhip_rng_device.h"

namespace mtfhip {


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

/* ---- the sampler and the dynamic models ---- */
struct PfArgs {
	int n, S;
	int dynamic_model, update_type;
	int sampler;               /* PF_SAMPLER_* (mtfhip_internal.h) */
	int nz;                    /* standard normals per particle */
	double ar_coeff;
	double sigma[8], mean[8];
	double init_corners[8];
	double aux_inv[9];         /* homography, corner based: inverse of square_to_quad(init_corners) (template corners -> unit square);
	                              affine, point based: inverse of the 3 x 3 matrix with rows (x_i, y_i, 1) of the three canonical points */
	double canon[6];           /* affine, point based: the canonical points (bottom right, bottom left, top centre), x,y interleaved */
	unsigned long long seed;
	unsigned iter;
	const double *normals;     /* [n][nz] standard normals, or NULL: Philox */
	/* several sampler distributions (PF.cc:240-269): particle k draws its distribution id from the weights of the previous
	 * iteration (distr_cum: their running sums) and perturbs with that distribution's sigma / mean */
	int n_distr;
	const double *distr_sigma, *distr_mean;   /* [n_distr][8] */
	const double *distr_cum;                  /* [n_distr] */
	const double *distr_uniforms;             /* [n] or NULL: Philox */
	int *distr_ids;                           /* [n] out */
};

/* pair `q` (draws 2 q, 2 q + 1) of particle k's normals */
__device__ __forceinline__ void pf_normal_pair(const PfArgs &a, unsigned iter, unsigned k, int q, double &z0, double &z1) {
	if (a.normals) {
		const double2 v = *reinterpret_cast<const double2 *>(a.normals + (size_t)k * a.nz + 2 * q);   /* nz is even: rows and pairs are 16-byte aligned */
		z0 = v.x; z1 = v.y;
	} else {
		philox_normal2(a.seed, iter, k, (unsigned)q, z0, z1);
	}
}

/* the SSM's generatePerturbation for one particle from its standard normals z[0..nz) */
template <int SSM>
__device__ __forceinline__ void pf_perturbation(const PfArgs &a, const double (&sg)[8], const double (&mn)[8], const double *z, double *pert) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
#pragma unroll
	for (int s = 0; s < 8; ++s) pert[s] = 0.0;
	if (SSM == MTFHIP_SSM_HOMOGRAPHY && a.sampler == PF_SAMPLER_HOM_CORNERS) {
		/* Homography::generatePerturbation, corner based (Homography.cc:899-911): one translation for all corners from
		 * distribution 0, one displacement per corner coordinate from distribution 1, then the warp that takes the template
		 * corners to the disturbed ones */
		double dc[8], Hq[9], Hp[9];
		const double tx = mn[0] + sg[0] * z[0], ty = mn[0] + sg[0] * z[1];
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			dc[2 * c] = a.init_corners[2 * c] + (mn[1] + sg[1] * z[2 + 2 * c]) + tx;
			dc[2 * c + 1] = a.init_corners[2 * c + 1] + (mn[1] + sg[1] * z[3 + 2 * c]) + ty;
		}
		square_to_quad_dev(dc, Hq);
		m3_mul_dev(Hq, a.aux_inv, Hp);
		const double n22 = Hp[8];
#pragma unroll
		for (int q = 0; q < 9; ++q) Hp[q] /= n22;
		state_from_warp_dev<SSM>(pert, Hp);
	} else if (SSM == MTFHIP_SSM_AFFINE && (a.sampler == PF_SAMPLER_AFF_PTS1 || a.sampler == PF_SAMPLER_AFF_PTS2)) {
		/* Affine::generatePerturbation, point based (Affine.cc:466-494): bottom right, bottom left and top centre of the template
		 * are disturbed -- 1: coordinate j by distribution j; 2: every coordinate by distribution 1 plus one translation from
		 * distribution 0 -- and the affine map of the three point pairs (utils::computeAffineDLT of three points,
		 * warpUtils.cc:388-421: an exactly determined 6 x 6 system) is the perturbation */
		double px[3], py[3];
		if (a.sampler == PF_SAMPLER_AFF_PTS1) {
#pragma unroll
			for (int i = 0; i < 3; ++i) {
				px[i] = a.canon[2 * i] + (mn[2 * i] + sg[2 * i] * z[2 * i]);
				py[i] = a.canon[2 * i + 1] + (mn[2 * i + 1] + sg[2 * i + 1] * z[2 * i + 1]);
			}
		} else {
			const double tx = mn[0] + sg[0] * z[6], ty = mn[0] + sg[0] * z[7];
#pragma unroll
			for (int i = 0; i < 3; ++i) {
				px[i] = (a.canon[2 * i] + (mn[1] + sg[1] * z[2 * i])) + tx;
				py[i] = (a.canon[2 * i + 1] + (mn[1] + sg[1] * z[2 * i + 1])) + ty;
			}
		}
		double W[9];
#pragma unroll
		for (int j = 0; j < 3; ++j) {
			W[j] = a.aux_inv[3 * j] * px[0] + a.aux_inv[3 * j + 1] * px[1] + a.aux_inv[3 * j + 2] * px[2];
			W[3 + j] = a.aux_inv[3 * j] * py[0] + a.aux_inv[3 * j + 1] * py[1] + a.aux_inv[3 * j + 2] * py[2];
		}
		W[6] = 0; W[7] = 0; W[8] = 1;
		state_from_warp_dev<SSM>(pert, W);
	} else if (SSM == MTFHIP_SSM_AFFINE && a.sampler == PF_SAMPLER_AFF_GEOM) {
		/* Affine::generatePerturbation, geometric (Affine.cc:495-502) = geomToState of six draws (Affine.cc:393-410):
		 * (tx, ty, scale, theta, aspect, phi) */
		double gm[6];
#pragma unroll
		for (int s = 0; s < 6; ++s) gm[s] = mn[s] + sg[s] * z[s];
		const double sc = gm[2], r = gm[4], theta = gm[3], phi = gm[5];
		const double cos_theta = cos(theta), sin_theta = sin(theta), cos_phi = cos(phi), sin_phi = sin(phi);
		const double ccc = cos_theta * cos_phi * cos_phi, ccs = cos_theta * cos_phi * sin_phi, css = cos_theta * sin_phi * sin_phi;
		const double scc = sin_theta * cos_phi * cos_phi, scs = sin_theta * cos_phi * sin_phi, sss = sin_theta * sin_phi * sin_phi;
		pert[0] = gm[0]; pert[1] = gm[1];
		pert[2] = sc * (ccc + scs + r * (css - scs)) - 1;
		pert[3] = sc * (r * (ccs - scc) - ccs - sss);
		pert[4] = sc * (scc - ccs + r * (ccs + sss));
		pert[5] = sc * (r * (ccc + scs) - scs + css) - 1;
	} else {
#pragma unroll
		for (int s = 0; s < S; ++s) pert[s] = mn[s] + sg[s] * z[s];   /* ProjectiveBase::generatePerturbation :283-288 */
	}
}
/* (st, ar) -> (ns, nar): PF.cc:307-335 */
template <int SSM>
__device__ __forceinline__ void pf_dynamics(const PfArgs &a, const double *pert, const double *st, const double *ar, double *ns, double *nar) {
#pragma unroll
	for (int s = 0; s < 8; ++s) nar[s] = ar[s];
	if (a.dynamic_model == 1 && a.update_type == 0) {          /* additiveAutoRegression1 :254-259 */
#pragma unroll
		for (int s = 0; s < 8; ++s) { ns[s] = st[s] + ar[s] + pert[s]; nar[s] = a.ar_coeff * (ns[s] - st[s]); }
	} else if (a.dynamic_model == 1) {                         /* compositionalAutoRegression1 Homography.cc:928-942, ProjectiveBase.cc:260-276 */
		double B[9], P[9], A[9], BA[9], W[9], Bi[9], AW[9];
		warp_from_state_dev<SSM>(st, B); warp_from_state_dev<SSM>(pert, P); warp_from_state_dev<SSM>(ar, A);
		m3_mul_dev(B, A, BA); m3_mul_dev(BA, P, W);
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double n22 = W[8];
_Pragma("unroll") for (int q = 0; q < 9; ++q) W[q] /= n22; }
		m3_inv_dev(B, Bi); m3_mul_dev(Bi, W, AW);
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double n22 = AW[8];
_Pragma("unroll") for (int q = 0; q < 9; ++q) AW[q] /= n22; }
		state_from_warp_dev<SSM>(ns, W); state_from_warp_dev<SSM>(nar, AW);
#pragma unroll
		for (int s = 0; s < 8; ++s) nar[s] *= a.ar_coeff;
	} else if (a.update_type == 0) {                           /* additiveRandomWalk :236-240 */
#pragma unroll
		for (int s = 0; s < 8; ++s) ns[s] = st[s] + pert[s];
	} else {                                                   /* compositionalRandomWalk Homography.cc:916-926, ProjectiveBase.cc:241-249 */
		double B[9], P[9], W[9];
		warp_from_state_dev<SSM>(st, B); warp_from_state_dev<SSM>(pert, P);
		m3_mul_dev(B, P, W);
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double n22 = W[8];
_Pragma("unroll") for (int q = 0; q < 9; ++q) W[q] /= n22; }
		state_from_warp_dev<SSM>(ns, W);
	}
}
/* a particle's row is S (6 or 8) contiguous doubles, 16-byte aligned: moved as pairs, all loads first */
template <int S>
__device__ __forceinline__ void pf_load_row(const double *base, size_t k, double *v) {
	const double2 *p = reinterpret_cast<const double2 *>(base + k * S);
	double2 t[4];
#pragma unroll
	for (int s2 = 0; s2 < 4; ++s2) t[s2] = 2 * s2 < S ? p[s2] : make_double2(0.0, 0.0);
#pragma unroll
	for (int s2 = 0; s2 < 4; ++s2) { v[2 * s2] = t[s2].x; v[2 * s2 + 1] = t[s2].y; }
}
template <int S>
__device__ __forceinline__ void pf_store_row(double *base, size_t k, const double *v) {
	double2 *p = reinterpret_cast<double2 *>(base + k * S);
#pragma unroll
	for (int s2 = 0; s2 < 4; ++s2) if (2 * s2 < S) p[s2] = make_double2(v[2 * s2], v[2 * s2 + 1]);
}
/* one thread: particle k's perturbation of iteration a.iter -- a function of the counter-based draws and the sampler only, NOT of
 * the particle's state: it can be made before the resampling that decides which state it is applied to (k_pf_select's second role) */
template <int SSM>
__device__ __forceinline__ void pf_draw_perturbation(const PfArgs &a, unsigned iter, unsigned k, double *pert) {
	double z[10];
#pragma unroll
	for (int q = 0; q < 5; ++q) {
		z[2 * q] = z[2 * q + 1] = 0.0;
		if (2 * q < a.nz) pf_normal_pair(a, iter, k, q, z[2 * q], z[2 * q + 1]);
	}
	double sg[8], mn[8];
#pragma unroll
	for (int s = 0; s < 8; ++s) { sg[s] = a.sigma[s]; mn[s] = a.mean[s]; }
	if (a.n_distr > 1) {
		/* the particle's distribution (PF.cc:261-269): the reference asks a boost discrete_distribution with a generator of its own; here
		 * one more counter-based uniform, inverted on the running sums of the distribution weights */
		double u;
		if (a.distr_uniforms) u = a.distr_uniforms[k];
		else {
			const Philox4 r = philox4x32_10(k, 0u, iter, 0x44495354u /* "DIST" */, (unsigned)a.seed, (unsigned)(a.seed >> 32));
			double u1;
			philox_uniform2(r, u, u1);
		}
		const double tgt = u * a.distr_cum[a.n_distr - 1];
		int id = 0;
		while (id < a.n_distr - 1 && a.distr_cum[id] < tgt) ++id;
		if (a.distr_ids) a.distr_ids[k] = id;
#pragma unroll
		for (int s = 0; s < 8; ++s) { sg[s] = a.distr_sigma[8 * id + s]; mn[s] = a.distr_mean[8 * id + s]; }
	}
	pf_perturbation<SSM>(a, sg, mn, z, pert);
}
/* one thread: (st, ar) of particle k -> its proposal */
template <int SSM>
__device__ __forceinline__ void pf_propose(const PfArgs &a, unsigned iter, unsigned k, const double *st, const double *ar, double *ns, double *nar) {
	double pert[8];
	pf_draw_perturbation<SSM>(a, iter, k, pert);
	pf_dynamics<SSM>(a, pert, st, ar, ns, nar);
}
/* the proposals of a whole set in a launch of their own: the first iteration after the particles were (re)initialised, draws
 * handed in by the caller, MeanType::Corners */
template <int SSM>
__global__ __launch_bounds__(kBlock) void k_pf_propose(PfArgs a, const double *st_in, const double *ar_in, double *st_out, double *ar_out) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	const int k = blockIdx.x * kBlock + threadIdx.x;
	if (k >= a.n) return;
	double st[8], ar[8], ns[8], nar[8];
	pf_load_row<S>(st_in, (size_t)k, st); pf_load_row<S>(ar_in, (size_t)k, ar);
	pf_propose<SSM>(a, a.iter, (unsigned)k, st, ar, ns, nar);
	pf_store_row<S>(st_out, (size_t)k, ns); pf_store_row<S>(ar_out, (size_t)k, nar);
}

/* ===================================================================== */
/* launch 1: scoring + weight                                             */
/* ===================================================================== */
/* One workgroup scores K = 4 consecutive particles of this rank's block (the structure of k_score_candidates_fast,
 * kernels_batch.hip: every wave takes a quarter of the pixels and evaluates all four warps on each grid point it loads; the
 * candidate index is uniform per workgroup, so the proposals arrive through scalar loads and the warps live in scalar registers).
 * Epilogue: similarity -> AM likelihood (SSD.h:41-43, NCC.cc:50-53) -> particle weight (PF.cc:341-365). */
struct PfScoreArgs {
	const double *prop;            /* [n][S] the proposals */
	int lo, cnt;                   /* this rank's block of particles */
	double alpha, norm_mult, norm_add;
	const double *ncc_sc;          /* NCC: [0] mean(I0) [1] |I0 - mean| */
	int likelihood_func;
	double measurement_sigma, max_similarity;
	double *wts;                   /* [>= n] particle_wts, written at the particles' global indices */
	double *sim;                   /* [n] similarities or NULL */
	PfPeerPush peer;               /* world > 0: the weights also go to every other rank's mailbox (mtfhip_internal.h) */
	/* hull_ok: the template grid is a unit-z grid laid out inside the quadrilateral hull[] = x0 y0 ... x3 y3 (its own corners, in the
	 * grid's coordinates).  A candidate whose warped hull lies inside the image, denominators positive, has EVERY sample inside (the
	 * map is projective: a point of the hull goes to a convex combination of the warped corners): its samples skip the border test. */
	int hull_ok;
	double hull[8];
	int lean_ok;                   /* 0: experiments (MTFHIP_PF_LEAN=0) */
	/* the row-pair copy of the frame (mtfhip_ctx::pair_owned; NULL: none): pair[2 (y W + x)] = I[y][x] | I[y + 1][x] -- a sample's cell is ONE
	 * 16-byte gather instead of two 8-byte ones (r06: the loop sits on the vector memory path as much as on FP64 issue -- 39.1 -> 36.0 us per
	 * 10 000 candidates, 3041 -> 2684 per million) */
	const float *pair;
	int pair_w;
};
/* A weight for the peers: a relaxed system-scope store -- it goes through to the peer's memory, and the wave's vmcnt tells when it
 * has (what every release fence relies on), so no fence and no L2 write-back per workgroup.  (First version: a system-scope release
 * per scoring workgroup.  Each is a write-back of the whole L2: the scoring launch of a 1 250-particle block went from 40 to 131 us.) */
__device__ __forceinline__ void pf_peer_store(const PfPeerPush &peer, int idx, double w) {
	for (int q = 0; q < peer.world; ++q)
		if (q != peer.rank) __hip_atomic_store(peer.wts[q] + idx, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
/* ... and the arrival: wave 0 of every workgroup (the wave whose lanes stored) waits for its stores to be acknowledged and counts
 * itself in on this rank's own counter (agent scope, like the scan's); the LAST workgroup of the launch then adds one arrival to
 * this rank's entry in every peer's counters, release at system scope -- once per launch. */
__device__ __forceinline__ void pf_peer_arrive(const PfPeerPush &peer, unsigned n_workgroups) {
	if (threadIdx.x >= 64) return;
	wait_stores_acked();
	if (threadIdx.x != 0) return;
	if (__hip_atomic_fetch_add(peer.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != n_workgroups - 1) return;
	__hip_atomic_store(peer.arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   /* zero between launches */
	for (int q = 0; q < peer.world; ++q)
		if (q != peer.rank) (void)__hip_atomic_fetch_add(peer.counters[q] + peer.rank, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
/* every workgroup of the first kernel that reads the gathered weights: thread q waits for rank q's arrivals.  The spin is bounded
 * (~seconds): a peer that never arrives becomes an error the host reports (PfPeerWait::err), not a hung queue. */
constexpr unsigned kPfPeerSpinLimit = 1u << 21;
__device__ __forceinline__ void pf_peer_wait(const PfPeerWait &w) {
	if (w.world == 0) return;
	const int q = threadIdx.x;
	if (q < w.world && q != w.rank) {
		unsigned long long need = 0;
#pragma unroll
		for (int r = 0; r < kPfMaxPeers; ++r) if (r == q) need = w.expected[r];
		unsigned spins = 0;
		while (__hip_atomic_load(w.counters + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
			__builtin_amdgcn_s_sleep(8);
			/* (once a wait has given up every later one does at its first look: a failed filter drains in milliseconds) */
			if ((++spins & 1023u) == 1u && __hip_atomic_load(w.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
			if (spins > kPfPeerSpinLimit) { __hip_atomic_store(w.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
		}
	}
	__syncthreads();
	__atomic_thread_fence(__ATOMIC_ACQUIRE);   /* (system scope: the peers' weights, not a cached older exchange) */
}
struct __attribute__((packed, aligned(4))) PfTexPair { float a, b; };
struct __attribute__((packed, aligned(4))) PfTexQuad { float t00, t10, t01, t11; };   /* a cell of the row-pair image: (x, y) (x, y + 1) (x + 1, y) (x + 1, y + 1) */
/* MC: the multi-channel models (MCSSD / MCNCC = SSD / NCC built with n_channels = 3, AM/src/MCSSD.cc): a row of the per-pixel
 * arrays is a (pixel, channel) pair, row = pixel * C + channel (mc::getPixVals imgUtils.cc:867-882); the grid point is the
 * pixel's, the texels the channel's (interleaved image); replay arithmetic uses mc::PixVal's weights-first order (pix_val_mc). */
/* K candidates per workgroup: 4 amortises the per-pixel operands (grid point, template value) over four warps; small blocks -- a
 * rank's share of a sharded filter, the reference's shipped 500 particles -- take 2 or 1, so that the launch still covers the device
 * (1 250 candidates are 313 workgroups at K = 4: one per CU and a 14 us launch).  A candidate's sums do not depend on K: the same
 * pixels per thread, the same xor butterfly, the same order over the waves. */
#ifdef MTFHIP_PF_TRACE   /* tools/pf_score_trace.sh: wall-clock stamps (100 MHz) of workgroup 0's phases */
__device__ unsigned long long g_pf_trace[16];
#define PF_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_pf_trace[k] = wall_clock64(); } while (0)
#else
#define PF_STAMP(k) do { } while (0)
#endif
template <int SSM, bool NCC, bool FAST, bool MC, int K, bool PAIR = false>
__global__ __launch_bounds__(kBlock) void k_pf_score(BatchView bv, ImgView im, PfScoreArgs s) {
	constexpr int M = NCC ? 3 : 1, S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	__shared__ double red[4 * K * M], tot[K * M];
	PF_STAMP(0);
	const int c0 = s.lo + blockIdx.x * K, cend = s.lo + s.cnt;
	const int tid = threadIdx.x;
	double W[K][9];
#pragma unroll
	for (int k = 0; k < K; ++k) warp_from_state_dev<SSM>(s.prop + (size_t)min(c0 + k, cend - 1) * S, W[k]);   /* uniform address: scalar loads */
	const unsigned N = (unsigned)bv.N;
	const unsigned Cc = MC ? (unsigned)bv.C : 1u;
	const bool uz = bv.unit_z != 0;
	const double *__restrict__ pp = bv.buf[uz ? MTFHIP_BUF_INIT_PTS : MTFHIP_BUF_INIT_HXY];
	const double *__restrict__ iz = bv.buf[MTFHIP_BUF_INIT_Z];
	const double *__restrict__ I0 = bv.buf[MTFHIP_BUF_I0];
	const float *__restrict__ img = im.data;
	const int iw1 = im.w - 1, ih1 = im.h - 1, stride = im.stride;
	double acc[K * M];
#pragma unroll
	for (int k = 0; k < K * M; ++k) acc[k] = 0.0;
	/* the workgroup's four candidates x the hull's four corners, one per lane (lanes 16.. repeat them): all sixteen inside? */
	bool all_inside = false;
	if constexpr (FAST) {
		if (s.hull_ok) {
			const int kk = (tid >> 2) & (K - 1), cc = tid & 3;
			double Wl[9];
			warp_from_state_dev<SSM>(s.prop + (size_t)min(c0 + kk, cend - 1) * S, Wl);
			const double X = cc == 0 ? s.hull[0] : cc == 1 ? s.hull[2] : cc == 2 ? s.hull[4] : s.hull[6];
			const double Y = cc == 0 ? s.hull[1] : cc == 1 ? s.hull[3] : cc == 2 ? s.hull[5] : s.hull[7];
			double hx = fma(Wl[0], X, fma(Wl[1], Y, Wl[2])), hy = fma(Wl[3], X, fma(Wl[4], Y, Wl[5]));
			bool in = true;
			if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
				const double dd = fma(Wl[6], X, fma(Wl[7], Y, Wl[8]));
				in = dd > 1e-9;
				const double inv = rcp_fast(dd);   /* (the margin below is 1e-3 px: no need for an IEEE division in every wave's prologue) */
				hx *= inv; hy *= inv;
			}
			/* (a margin of a thousandth of a pixel: the samples' own rounding is ~1e-13) */
			in = in & (hx > 1e-3) & (hy > 1e-3) & (hx < (double)iw1 - 1e-3) & (hy < (double)ih1 - 1e-3);
			all_inside = __builtin_amdgcn_ballot_w64(!in) == 0;
		}
	}
	PF_STAMP(1);
	/* LEAN (wave-uniform, decided once): a unit-z grid and the identity pixel normalisation -- W2 * 1.0 and fma(1.0, v, 0.0) are the
	 * same bits without the instruction: 4 of the 44 VALU instructions of a homography candidate-sample, on a loop that sits at the
	 * measured FP64 issue ceiling (profiles/r04_fp64_rates.txt) */
	auto pixel_loop = [&](auto inside_tag, auto lean_tag) {
	constexpr bool INSIDE = decltype(inside_tag)::value;
	constexpr bool LEAN = FAST && decltype(lean_tag)::value;
	for (unsigned i = tid; i < N; i += kBlock) {
		const unsigned pi = MC ? i / Cc : i;          /* the row's pixel */
		const int ch = MC ? (int)(i - pi * Cc) : 0;   /* ... and channel */
		const double2 q = ld_off<double2>(pp, pi * 16u);
		const double z = uz ? 1.0 : ld_off<double>(iz, pi * 8u);
		const double i0 = ld_off<double>(I0, i * 8u);
#pragma unroll
		for (int k = 0; k < K; ++k) {
			double it;
			if constexpr (FAST) {
				/* (this loop is bound by FP64 issue, ~35 double-rate instructions per candidate-pixel: the cell fractions come from
				 * v_fract_f64 -- exact, the same bits as x - (double)(int)x for the non-negative coordinates the fast path accepts --
				 * instead of convert-back + subtract.  A reciprocal with one Newton step instead of two saved as much again, but its
				 * last-bit differences flip round(w n) ties of the residual resampler against the oracle: not taken.) */
				/* (z is exactly 1.0 on a unit-z grid and x * 1.0 == x: no select on the flag -- a per-lane select of a wave-uniform
				 * condition was two moves + two v_cndmask per coordinate, 8 of the 44 VALU instructions of a candidate-sample) */
				double wx = fma(W[k][0], q.x, fma(W[k][1], q.y, LEAN ? W[k][2] : W[k][2] * z));
				double wy = fma(W[k][3], q.x, fma(W[k][4], q.y, LEAN ? W[k][5] : W[k][5] * z));
				if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
					const double dd = fma(W[k][6], q.x, fma(W[k][7], q.y, LEAN ? W[k][8] : W[k][8] * z));
					const double inv = rcp_fast(dd);
					wx *= inv; wy *= inv;
				}
				const int lx = (int)wx, ly = (int)wy;
				const bool ok = INSIDE | ((wx >= 0) & (wy >= 0) & (lx < iw1) & (ly < ih1));
				double v;
				if (INSIDE || __builtin_amdgcn_ballot_w64(!ok) == 0) {
					const double fx = __builtin_amdgcn_fract(wx), fy = __builtin_amdgcn_fract(wy);
					if constexpr (MC) {
						const unsigned off = (unsigned)(ly * stride + lx * (int)Cc + ch) * 4u;
						const float t00 = ld_off<float>(img, off), t01 = ld_off<float>(img, off + 4u * Cc);
						const float t10 = ld_off<float>(img + stride, off), t11 = ld_off<float>(img + stride, off + 4u * Cc);
						v = bilin_val_fast(t00, t01, t10, t11, fx, fy);
					} else {
						/* (v_mul_lo_u32 is a quarter-rate instruction, 16 cycles of a ~900-cycle iteration each: the row and the pitch are
						 * below 2^24 -- mtfhip_image_upload / _borrow refuse larger frames -- so the 24-bit multiply-add, full rate, gives the same offset) */
						const unsigned off = (__umul24((unsigned)ly, (unsigned)stride) + (unsigned)lx) * 4u;
						if constexpr (PAIR) {   /* (a launch parameter as a template argument: with both forms behind a run-time flag the loop was 1.7 x slower in either) */
							const PfTexQuad q4 = ld_off<PfTexQuad>(s.pair, (__umul24((unsigned)ly, (unsigned)s.pair_w) + (unsigned)lx) * 8u);
							v = bilin_val_fast(q4.t00, q4.t01, q4.t10, q4.t11, fx, fy);
						} else {
							const PfTexPair t0 = ld_off<PfTexPair>(img, off), t1 = ld_off<PfTexPair>(img + stride, off);
							v = bilin_val_fast(t0.a, t0.b, t1.a, t1.b, fx, fy);
						}
					}
				} else {
					if constexpr (MC) v = pix_val_mc(im, wx, wy, ch); else v = pix_val_fast(im, wx, wy);
				}
				it = LEAN ? v : fma(s.norm_mult, v, s.norm_add);
			} else {   /* the reference's operation order (ProjectiveBase.cc:41-49, imgUtils.h:91-113, 505-551) */
				double wx = W[k][0] * q.x + W[k][1] * q.y + W[k][2] * z, wy = W[k][3] * q.x + W[k][4] * q.y + W[k][5] * z;
				if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
					const double d = W[k][6] * q.x + W[k][7] * q.y + W[k][8] * z;
					wx = wx / d; wy = wy / d;
				}
				it = s.norm_mult * (MC ? pix_val_mc(im, wx, wy, ch) : pix_val(im, wx, wy)) + s.norm_add;
			}
			if constexpr (NCC) {
				acc[3 * k] += it; acc[3 * k + 1] = fma(it, it, acc[3 * k + 1]); acc[3 * k + 2] = fma(i0, it, acc[3 * k + 2]);
			} else {
				const double r = it - i0;
				acc[k] = fma(r, r, acc[k]);
			}
		}
	}
	};
	const bool lean = FAST && uz && s.lean_ok && s.norm_mult == 1.0 && s.norm_add == 0.0;
	if (lean) { if (all_inside) pixel_loop(std::true_type{}, std::true_type{}); else pixel_loop(std::false_type{}, std::true_type{}); }
	else { if (all_inside) pixel_loop(std::true_type{}, std::false_type{}); else pixel_loop(std::false_type{}, std::false_type{}); }
	PF_STAMP(2);
	block_reduce_store<K * M>(acc, tot, red);
	__syncthreads();
	PF_STAMP(3);
	const int k = tid, cand = c0 + k;
	if (k < K && cand < cend) {
		double f, lik;
		if constexpr (NCC) {
			const double n = (double)N, m0 = s.ncc_sc[0], c = s.ncc_sc[1], mt = tot[3 * k] / n;
			f = (tot[3 * k + 2] - n * m0 * mt) / (sqrt(tot[3 * k + 1] - n * mt * mt) * c);
			const double d = (1.0 / f) - 1;
			lik = exp(-s.alpha * d * d);
		} else {
			f = -tot[k] / 2;
			lik = exp(-s.alpha * sqrt(-f / (double)N));
		}
		double w = lik;
		if (s.likelihood_func != 0) {
			const double pi = 3.14159265358979323846;
			const double val = s.max_similarity - f;
			w = s.likelihood_func == 1 ? (1.0 / sqrt(2 * pi * s.measurement_sigma)) * exp(-0.5 * val / s.measurement_sigma)   /* PF.cc:69-70, 352-354 */
			                           : 1.0 / (1.0 + val);
		}
		if (s.wts) s.wts[cand] = w;
		if (s.sim) s.sim[cand] = f;
		if (s.peer.world) pf_peer_store(s.peer, cand, w);
	}
	if (s.peer.world) pf_peer_arrive(s.peer, gridDim.x);
	PF_STAMP(4);
}
/* the same for a scorer that does not store to the peers itself (MI): 64 weights per workgroup, stored by wave 0 */
constexpr int kPfPushPerGroup = 64;
__global__ __launch_bounds__(64) void k_pf_peer_push(PfPeerPush peer, const double *wts, int lo, int cnt) {
	const int i = blockIdx.x * kPfPushPerGroup + threadIdx.x;
	if (i < cnt) pf_peer_store(peer, lo + i, wts[lo + i]);
	pf_peer_arrive(peer, gridDim.x);
}
__global__ __launch_bounds__(kBlock) void k_pf_peer_wait(PfPeerWait w) { pf_peer_wait(w); }

/* ===================================================================== */
/* launch 2: cumulative weights                                           */
/* ===================================================================== */
/* A chunk is 256 consecutive particles = what one wave of the scan covers (four per lane): the wave turns its weights into
 * chunk-local inclusive sums cum[] (entries past n repeat the chunk total, so a search inside a chunk never needs a bound) and
 * hands the chunk total to the last workgroup to arrive, which scans the totals into chunk_incl[] (inclusive).
 * particle_cum_wts[k] of the reference (PF.cc:366-367) is chunk_incl[c - 1] + cum[k]; its normalisation (PF.cc:455-459) becomes a
 * scaled draw in the selection pass.  (256 and not more: the chunk table of up to a million particles then fits the selection
 * pass's LDS, and what is left to search in memory is 32 cache lines.) */
constexpr int kPfChunk = 256;
struct PfScanArgs {
	int n, nch;
	const double *wts;
	double *cum;          /* [nch * 256] */
	double *sub16;        /* [nch * 16] chunk-local inclusive sum at the end of every 16 particles: the middle level of the search */
	double *chunk_tot;    /* [nch] */
	double *chunk_incl;   /* [nch] */
	int *counter;         /* zero between launches */
	/* optional (stats != NULL): sum of squared weights for adaptive resampling (PF.cc:381-390) and, with several sampler
	 * distributions, the weight sum and the particle count of every distribution (PF.cc:345-369) */
	double *stats;        /* [nch][17]: sum w^2 | 8 x sum w | 8 x count */
	const int *distr_ids; /* [n] or NULL */
	int n_distr;
	double min_distr_wt, min_eff;   /* min_eff: adaptive_resampling_thresh x n, 0 = resample every iteration */
	double *distr_cum, *distr_wts;  /* [n_distr] out: the distribution weights of the next iteration and their running sums */
	int *resample_flag;   /* out: 1 when this iteration resamples */
	PfPeerWait wait;      /* sharded filter with the peer-store exchange: the weights are complete when the peers have arrived */
};
/* inclusive prefix sum over the wave with DPP moves (row_shr 1 / 2 / 4 / 8 inside the rows of 16, row_bcast:15 / :31 across them; a
 * lane without a source adds 0.0): ~18 VALU instructions instead of six ds_bpermute round trips through the LDS crossbar (~120
 * cycles each) -- 1.5 us of the 5.5 us scan launch at 10 000 particles */
__device__ __forceinline__ double wave_scan_incl(double x, int /* lane */) {
#define MTFHIP_SCAN_STEP(CTRL, ROWS) { \
		const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROWS, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROWS, 0xF, false); \
		x += __hiloint2double(hi, lo); }
	MTFHIP_SCAN_STEP(0x111, 0xF) MTFHIP_SCAN_STEP(0x112, 0xF) MTFHIP_SCAN_STEP(0x114, 0xF) MTFHIP_SCAN_STEP(0x118, 0xF)
	MTFHIP_SCAN_STEP(0x142, 0xA) MTFHIP_SCAN_STEP(0x143, 0xC)
#undef MTFHIP_SCAN_STEP
	return x;
}
__global__ __launch_bounds__(kBlock) void k_pf_scan(PfScanArgs a) {
	__shared__ double wsum[kBlock / 64];
	__shared__ int is_last;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int chunk = blockIdx.x * (kBlock / 64) + wave;
	const int base = chunk * kPfChunk + 4 * lane;
	const int nch = a.nch;
	pf_peer_wait(a.wait);
	if (chunk < nch) {
		double w[4];
		if (base + 3 < a.n) {
			const double2 v0 = *reinterpret_cast<const double2 *>(a.wts + base), v1 = *reinterpret_cast<const double2 *>(a.wts + base + 2);
			w[0] = v0.x; w[1] = v0.y; w[2] = v1.x; w[3] = v1.y;
		} else {
#pragma unroll
			for (int j = 0; j < 4; ++j) w[j] = base + j < a.n ? a.wts[base + j] : 0.0;
		}
		const double p0 = w[0], p1 = p0 + w[1], p2 = p1 + w[2], p3 = p2 + w[3];
		const double incl = wave_scan_incl(p3, lane);
		const double off = incl - p3;
		*reinterpret_cast<double2 *>(a.cum + base) = make_double2(off + p0, off + p1);
		*reinterpret_cast<double2 *>(a.cum + base + 2) = make_double2(off + p2, off + p3);
		if ((lane & 3) == 3) a.sub16[chunk * 16 + (lane >> 2)] = off + p3;
		if (lane == 63) st_coh(a.chunk_tot + chunk, incl);
		if (a.stats) {
			double sv[17];
#pragma unroll
			for (int q = 0; q < 17; ++q) sv[q] = 0.0;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				sv[0] = fma(w[j], w[j], sv[0]);
				if (a.n_distr > 1 && base + j < a.n) {
					const int id = a.distr_ids[base + j];
#pragma unroll
					for (int q = 0; q < 8; ++q) if (id == q) { sv[1 + q] += w[j]; sv[9 + q] += 1.0; }
				}
			}
#pragma unroll
			for (int q = 0; q < 17; ++q)
#pragma unroll
				for (int d = 32; d >= 1; d >>= 1) sv[q] += __shfl_xor(sv[q], d);   /* a fixed balanced tree */
			if (lane == 0) {
#pragma unroll
				for (int q = 0; q < 17; ++q) st_coh(a.stats + (size_t)chunk * 17 + q, sv[q]);
			}
		}
	}
	wait_stores_acked();
	__syncthreads();
	if (tid == 0) is_last = __hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
	__syncthreads();
	if (!is_last) return;
	/* chunk totals -> inclusive prefix: a thread sums a contiguous run (up to 16 totals -- a million particles -- are fetched with
	 * independent loads first: these are L2-bypassing loads, a microsecond each when they wait for one another), the runs are
	 * scanned over the workgroup */
	const int per = (nch + kBlock - 1) / kBlock, lo = min(tid * per, nch), hi = min(lo + per, nch);
	constexpr int kRun = 16;
	double run = 0.0, tv[kRun];
	if (per <= kRun) {
#pragma unroll
		for (int j = 0; j < kRun; ++j) tv[j] = lo + j < hi ? ld_coh(a.chunk_tot + lo + j) : 0.0;
#pragma unroll
		for (int j = 0; j < kRun; ++j) run += tv[j];   /* (+ 0.0 past the run: exact) */
	} else {
		for (int c = lo; c < hi; ++c) run += ld_coh(a.chunk_tot + c);
	}
	const double ri = wave_scan_incl(run, lane);
	if (lane == 63) wsum[wave] = ri;
	__syncthreads();
	double c0 = ri - run;
#pragma unroll
	for (int v = 0; v < kBlock / 64 - 1; ++v) if (v < wave) c0 += wsum[v];
	if (per <= kRun) {
#pragma unroll
		for (int j = 0; j < kRun; ++j) if (lo + j < hi) { c0 += tv[j]; a.chunk_incl[lo + j] = c0; }
	} else {
		for (int c = lo; c < hi; ++c) { c0 += ld_coh(a.chunk_tot + c); a.chunk_incl[c] = c0; }
	}
	if (a.stats) {
		/* the per-chunk statistics, summed in chunk order (fixed: every rank gets the same bits): thread q < 17 owns column q */
		__shared__ double tot_s[17];
		if (tid < 17) {
			double acc8[8];
#pragma unroll
			for (int u = 0; u < 8; ++u) acc8[u] = 0.0;
			int c = 0;
			for (; c + 7 < nch; c += 8) {
#pragma unroll
				for (int u = 0; u < 8; ++u) acc8[u] += ld_coh(a.stats + (size_t)(c + u) * 17 + tid);
			}
			for (; c < nch; ++c) acc8[0] += ld_coh(a.stats + (size_t)c * 17 + tid);
			tot_s[tid] = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
		}
		__syncthreads();   /* (also: chunk_incl[nch - 1] below was written by this workgroup) */
		if (tid == 0) {
			const double total = a.chunk_incl[nch - 1];
			/* n_eff = 1 / sum (w / sum w)^2 (PF.cc:382-384) */
			const double sq = tot_s[0] / (total * total);
			const double n_eff = (sq == 0 || !(sq == sq)) ? 0.0 : 1.0 / sq;
			if (a.resample_flag) *a.resample_flag = (a.min_eff > 0 && n_eff > a.min_eff) ? 0 : 1;
			if (a.n_distr > 1) {   /* PF.cc:354-369 */
				double wv[8], wt_sum = 0.0;
#pragma unroll
				for (int q = 0; q < 8; ++q) {
					wv[q] = q < a.n_distr ? tot_s[1 + q] : 0.0;
					if (q < a.n_distr && tot_s[9 + q] > 0) { wv[q] /= tot_s[9 + q]; wt_sum += wv[q]; }
				}
				double run = 0.0;
#pragma unroll
				for (int q = 0; q < 8; ++q) {
					if (q < a.n_distr) {
						double v = wv[q] / wt_sum;
						if (v < a.min_distr_wt) v = a.min_distr_wt;
						run += v;
						a.distr_wts[q] = v; a.distr_cum[q] = run;
					}
				}
			}
		}
	}
	if (tid == 0) __hip_atomic_store(a.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* ===================================================================== */
/* launch 3: resampling + estimate (+ the proposals of the next iteration) */
/* ===================================================================== */
/* per-workgroup row of the selection pass: [0] best weight [1] its new index [2..9] sum of states [10..17] sum of corners
 * [18..25] the state of the workgroup's best particle */
constexpr int kPfPart = 26;
constexpr int kPfTable = 4096;   /* chunk table in LDS: up to 1 048 576 particles; beyond, the table is searched in memory */
struct PfPublish { double *host; unsigned long long *flag, seq; int fenced; };
struct PfSelectArgs {
	int resampling_type, mean_type;
	int lookahead;                /* 1: the proposals of the next iteration (draws of a.iter + 1) are produced here */
	const double *uniforms;       /* [n] or NULL: Philox */
	const int *resample_flag;     /* NULL, or the scan's verdict (adaptive resampling): 0 = this iteration keeps its proposals */
	const double *wts, *cum, *sub16, *chunk_incl;
	const double *prop, *prop_ar; /* [n][S] this iteration's proposals */
	double *st_out, *ar_out;      /* [n][S] the (resampled) set the iteration leaves behind */
	double *next, *next_ar;       /* [n][S] lookahead: the proposal set of the next iteration */
	int *ids;                     /* [n] resample ids (diagnostics / tests); residual resampling: an INPUT (k_pf_residual_map) */
	const int *forced_best;       /* residual resampling: max_wt_id = particle_idx[0] (PF.cc:581), else NULL */
	double init_corners_hm[12];
	double *parts;                /* [nblocks][kPfPart] */
	double *gparts;               /* [ceil(nblocks / 64)][kPfPart] group rows */
	int *counter;                 /* [1 + ceil(nblocks / 64)]: top-level counter, then one per group; zero between launches */
	double *out;                  /* [32]: estimate state (8) | max_wt | max_wt_id | mean corners (8) */
	PfPublish pub;
	/* workgroups [0, nsel) select; workgroups [nsel, gridDim.x) (pert_out != NULL) draw the perturbations of iteration pert_out_iter =
	 * a.iter + 2 into pert_out [n][8]: they need nothing of this iteration, run beside the selection on otherwise idle CUs, and the
	 * selection pass after next finds them done (pert_in: those of a.iter + 1, drawn two launches ago) -- Philox + Box-Muller + the
	 * corner-based homography were 5 of this launch's 16 us at 10 000 particles, behind the estimate but in front of the next scorer */
	/* 0: no estimate from this launch (PF.cc:421-437 computes it every iteration; with a negative epsilon and MeanType other than
	 * Corners only the LAST iteration's is ever read -- mtfhip_pf_update's chained form): no best-particle search, no per-workgroup
	 * rows, no last-arriver fold, ~3 us of the launch's dependent chain at 10 000 particles */
	int estimate;
	int nsel;
	const double *pert_in;
	double *pert_out;
	unsigned pert_out_iter;
	PfPeerWait wait;              /* LOCAL: this launch is the first reader of the gathered weights */
};
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "kernels_pf.hip: k_pf_select<.., LOCAL> keeps 16 384 cumulative weights (128 KB) in one workgroup's LDS -- gfx950's 160 KB; this library is built for gfx950 only (Makefile: ARCH)"
#endif
constexpr int kPfLocalMax = 16384;   /* LOCAL: particles whose cumulative weights fit one workgroup's LDS (128 KB of the 160) */
/* a folded row: best weight and its index, the sixteen sums, the state of the best particle (all wave-uniform) */
struct PfRow { double v; int i; double sum[16]; double st[8]; };
/* element `idx` (per-lane) of a small uniform array without indexing registers dynamically */
template <int N>
__device__ __forceinline__ double pf_pick(const double (&a)[N], int idx) {
	double v = a[0];
#pragma unroll
	for (int q = 1; q < N; ++q) v = idx == q ? a[q] : v;
	return v;
}
/* `count` <= 64 rows of kPfPart doubles, one per lane, every load issued before the first use; best: larger weight, then
 * larger index (rows are in particle order: `>=` in index order); sums: xor butterfly = a fixed balanced tree */
__device__ __forceinline__ void pf_fold64(const double *rows, int count, int lane, PfRow &o) {
	double x[kPfPart];
	const double *p = rows + (size_t)min(lane, count - 1) * kPfPart;
#pragma unroll
	for (int q = 0; q < kPfPart; ++q) x[q] = ld_coh(p + q);
	const bool live = lane < count;
	double bv = live ? x[0] : -1.7976931348623157e308;
	int bi = live ? (int)x[1] : -1, bl = lane;
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		const double ov = __shfl_xor(bv, d); const int oi = __shfl_xor(bi, d), ol = __shfl_xor(bl, d);
		if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; bl = ol; }
	}
	o.v = bv; o.i = bi;
#pragma unroll
	for (int q = 0; q < 16; ++q) {
		double sv = live ? x[2 + q] : 0.0;
#pragma unroll
		for (int d = 32; d >= 1; d >>= 1) sv += __shfl_xor(sv, d);
		o.sum[q] = sv;
	}
#pragma unroll
	for (int q = 0; q < 8; ++q) o.st[q] = __shfl(x[18 + q], bl);
}
/* LOCAL (n <= kPfLocalMax, multinomial resampling, no scan statistics): no k_pf_scan launch in front -- every workgroup builds
 * the cumulative weights of the WHOLE set in its own LDS (10 000 weights are 80 KB, read from L2 by 40 workgroups), with the scan's
 * arithmetic (chunk-local sums by wave_scan_incl, the chunk totals' prefix in the order of its last workgroup: the same bits), and
 * the three-level search reads LDS instead of four cache lines.  One launch and two dependent memory round trips less per iteration
 * (k_pf_scan: 5.4 us + the launch gap at 10 000 particles). */
template <int SSM, bool LOCAL>
__global__ __launch_bounds__(kBlock) void k_pf_select(PfArgs a, PfSelectArgs r) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	__shared__ double table[LOCAL ? 64 : kPfTable];
	__shared__ double lcum[LOCAL ? kPfLocalMax : 1];
	__shared__ double ctot[64];
	__shared__ double lds[4 * 16];
	__shared__ double red_v[kBlock / 64]; __shared__ int red_i[kBlock / 64];
	__shared__ double best_state[8];
	__shared__ int wg_best, is_last;
	const int n = a.n, tid = threadIdx.x;
	if ((int)blockIdx.x >= r.nsel) {   /* the perturbations of the iteration after next */
		const int kp = ((int)blockIdx.x - r.nsel) * kBlock + tid;
		if (kp < n) {
			double pert[8];
			pf_draw_perturbation<SSM>(a, r.pert_out_iter, (unsigned)kp, pert);
			pf_store_row<8>(r.pert_out, (size_t)kp, pert);
		}
		return;
	}
	const int k = blockIdx.x * kBlock + tid;
	const bool go = r.resample_flag ? *r.resample_flag != 0 : true;   /* (uniform: a scalar load) */
	const bool resample = go && (r.resampling_type == 1 || r.resampling_type == 2);
	const int nch = (n + kPfChunk - 1) / kPfChunk;
	const bool in_lds = LOCAL || nch <= kPfTable;
	double total = 0.0;
	/* (requested now, used after the search: the row this thread's successor is proposed from) */
	double pin[8];
#pragma unroll
	for (int q = 0; q < 8; ++q) pin[q] = 0.0;
	const bool have_pert = r.lookahead && r.pert_in != nullptr;
	if (have_pert && k < n) pf_load_row<8>(r.pert_in, (size_t)k, pin);
	if constexpr (LOCAL) {
		pf_peer_wait(r.wait);
		if (resample) {
			const int lane = tid & 63, wave = tid >> 6;
			constexpr int kB = 8;   /* chunks per wave in flight: wave w takes chunks w, w + 4, ... */
			for (int cb = wave; cb < nch; cb += 4 * kB) {
				double w[kB][4];
#pragma unroll
				for (int u = 0; u < kB; ++u) {
					const int c = cb + 4 * u, base = c * kPfChunk + 4 * lane;
#pragma unroll
					for (int j = 0; j < 4; ++j) w[u][j] = 0.0;
					if (c < nch) {   /* k_pf_scan's loads */
						if (base + 3 < n) {
							const double2 v0 = *reinterpret_cast<const double2 *>(r.wts + base), v1 = *reinterpret_cast<const double2 *>(r.wts + base + 2);
							w[u][0] = v0.x; w[u][1] = v0.y; w[u][2] = v1.x; w[u][3] = v1.y;
						} else {
#pragma unroll
							for (int j = 0; j < 4; ++j) w[u][j] = base + j < n ? r.wts[base + j] : 0.0;
						}
					}
				}
#pragma unroll
				for (int u = 0; u < kB; ++u) {
					const int c = cb + 4 * u, base = c * kPfChunk + 4 * lane;
					if (c < nch) {   /* ... and its sums */
						const double p0 = w[u][0], p1 = p0 + w[u][1], p2 = p1 + w[u][2], p3 = p2 + w[u][3];
						const double incl = wave_scan_incl(p3, lane);
						const double off = incl - p3;
						*reinterpret_cast<double2 *>(lcum + base) = make_double2(off + p0, off + p1);
						*reinterpret_cast<double2 *>(lcum + base + 2) = make_double2(off + p2, off + p3);
						if (lane == 63) ctot[c] = incl;
					}
				}
			}
			__syncthreads();
			/* the chunk totals' inclusive prefix as k_pf_scan's last workgroup takes it with one total per thread (nch <= 64: wave 0) */
			if (tid < 64) {
				const double tv0 = tid < nch ? ctot[tid] : 0.0;
				double run = 0.0;
				run += tv0;
				const double ri = wave_scan_incl(run, tid);
				double c0 = ri - run;
				c0 += tv0;
				if (tid < nch) table[tid] = c0;
			}
			__syncthreads();
			total = table[nch - 1];
		}
	} else if (resample) {
		if (in_lds) for (int j = tid; j < nch; j += kBlock) table[j] = r.chunk_incl[j];
		total = r.chunk_incl[nch - 1];
		__syncthreads();
	}
	double bv = -1.7976931348623157e308; int bi = -1;
	double acc[16], ns[8], nar[8];
#pragma unroll
	for (int s = 0; s < 16; ++s) acc[s] = 0.0;
#pragma unroll
	for (int s = 0; s < 8; ++s) ns[s] = nar[s] = 0.0;
	if (k < n) {
		int id = k;
		if (resample) {
			/* multinomial resampling (PF.cc:455-502 binary search; :505-536 linear search: both return the smallest index whose
			 * normalised cumulative weight reaches the draw): the draw is scaled by the total instead of every weight being divided */
			const double u = r.uniforms ? r.uniforms[k] : philox_uniform(a.seed, a.iter, (unsigned)k);
			const double tgt = u * total;
			int l = 0, h = nch - 1, c = (l + h) / 2;
			if (in_lds) { while (h > l) { if (table[c] >= tgt) h = c; else l = c + 1; c = (l + h) / 2; } }
			else { while (h > l) { if (r.chunk_incl[c] >= tgt) h = c; else l = c + 1; c = (l + h) / 2; } }
			const double off = c > 0 ? (in_lds ? table[c - 1] : r.chunk_incl[c - 1]) : 0.0;
			/* inside the chunk: the sixteen 16-particle sub-blocks (their end sums are contiguous: two cache lines), then the sixteen
			 * particles of the sub-block (two more) -- two rounds of independent loads, four lines; bisecting cum[] itself touched six
			 * and took eight dependent rounds */
			const double *cl = LOCAL ? lcum + (size_t)c * kPfChunk : r.cum + (size_t)c * kPfChunk;
			int lo;
			if constexpr (LOCAL) {   /* (the sub-block end sums are cum[16 q + 15]) */
				double v[16];
#pragma unroll
				for (int q = 0; q < 16; ++q) v[q] = cl[16 * q + 15];
				int cnt = 0;
#pragma unroll
				for (int q = 0; q < 16; ++q) cnt += (off + v[q] < tgt) ? 1 : 0;
				lo = 16 * min(cnt, 15);
			} else {
				const double2 *p2 = reinterpret_cast<const double2 *>(r.sub16 + (size_t)c * 16);
				double2 v[8];
#pragma unroll
				for (int q = 0; q < 8; ++q) v[q] = p2[q];
				int cnt = 0;
#pragma unroll
				for (int q = 0; q < 8; ++q) cnt += ((off + v[q].x < tgt) ? 1 : 0) + ((off + v[q].y < tgt) ? 1 : 0);
				lo = 16 * min(cnt, 15);
			}
			{
				const double2 *p2 = reinterpret_cast<const double2 *>(cl + lo);
				double2 v[8];
#pragma unroll
				for (int q = 0; q < 8; ++q) v[q] = p2[q];
				int cnt = 0;
#pragma unroll
				for (int q = 0; q < 8; ++q) cnt += ((off + v[q].x < tgt) ? 1 : 0) + ((off + v[q].y < tgt) ? 1 : 0);
				lo += min(cnt, 15);
			}
			id = min(c * kPfChunk + lo, n - 1);
			if (r.ids) r.ids[k] = id;
		}
		if (go && r.resampling_type == 3) id = r.ids[k];   /* residual resampling: the sources were laid out by k_pf_residual_map */
		if (!go && r.ids) r.ids[k] = k;
		/* the auto-regression terms travel with the particle only where a model reads them (AutoRegression1): under RandomWalk they
		 * stay what initializeParticles made them -- zero -- and three of the seven 64-byte rows this pass moves per particle go away
		 * (it is bound by those rows at a million particles: 832 -> 640 bytes of cache lines per particle) */
		const bool use_ar = a.dynamic_model == 1;
		pf_load_row<S>(r.prop, (size_t)id, ns);
		if (use_ar) pf_load_row<S>(r.prop_ar, (size_t)id, nar);
		pf_store_row<S>(r.st_out, (size_t)k, ns);
		if (use_ar) pf_store_row<S>(r.ar_out, (size_t)k, nar);
		if (r.estimate) { bv = r.wts[id]; bi = k; }
		if (go && r.forced_best && k != *r.forced_best) bv = -1.7976931348623157e308;   /* max_wt_id is handed down, not searched for */
		if (!r.estimate) {
		} else if (r.mean_type == 1) {
#pragma unroll
			for (int s = 0; s < 8; ++s) acc[s] = ns[s];
		} else if (r.mean_type == 2) {   /* updateMeanCorners :607-614 */
			double W[9];
			warp_from_state_dev<SSM>(ns, W);
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const double X = r.init_corners_hm[3 * q], Y = r.init_corners_hm[3 * q + 1], Z = r.init_corners_hm[3 * q + 2];
				double nx = W[0] * X + W[1] * Y + W[2] * Z, ny = W[3] * X + W[4] * Y + W[5] * Z;
				if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double d = W[6] * X + W[7] * Y + W[8] * Z; nx = nx / d; ny = ny / d; }
				acc[8 + 2 * q] = nx; acc[9 + 2 * q] = ny;
			}
		}
	}
	/* (estimate == 0: an iteration in the middle of a chained update() -- nobody reads its estimate, PfSelectArgs::estimate) */
	if (r.estimate) {
	/* the best of the (resampled) set, last index on ties (`>=` in index order, PF.cc:378-381, 487-490) */
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) {
		const double ov = __shfl_xor(bv, d); const int oi = __shfl_xor(bi, d);
		if (ov > bv || (ov == bv && oi > bi)) { bv = ov; bi = oi; }
	}
	if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
	__syncthreads();
	if (tid == 0) {
		for (int w = 1; w < kBlock / 64; ++w) if (red_v[w] > bv || (red_v[w] == bv && red_i[w] > bi)) { bv = red_v[w]; bi = red_i[w]; }
		red_v[0] = bv; wg_best = bi;
	}
	__syncthreads();
	if (k < n && k == wg_best) {
#pragma unroll
		for (int s = 0; s < 8; ++s) best_state[s] = ns[s];
	}
	double *part = r.parts + (size_t)blockIdx.x * kPfPart;
	block_reduce_store<16, true>(acc, part + 2, lds);   /* (its barrier also orders best_state) */
	if (tid == 0) { st_coh(part, red_v[0]); st_coh(part + 1, (double)wg_best); }
	if (tid >= 64 && tid < 72) st_coh(part + 18 + (tid - 64), best_state[tid - 64]);
	wait_stores_acked();
	__syncthreads();
	/* ---- the estimate (PF.cc:421-437).  The rows are folded by a two-level tree of last arrivers: workgroups form groups of 64,
	 * the last one of a group folds the group's rows (one row per lane, every load in flight at once -- a single workgroup
	 * walking thousands of rows through L2-bypassing loads was 340 of the 410 us of this kernel at a million particles), the last
	 * group to finish folds the group rows.  Sums are taken in a fixed tree, the same on every rank. ---- */
	const int nparts = r.nsel, ngroups = (nparts + 63) / 64, grp = blockIdx.x >> 6;
	const int gsize = min(64, nparts - grp * 64);
	if (tid == 0) is_last = __hip_atomic_fetch_add(r.counter + 1 + grp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize - 1;
	__syncthreads();
	auto estimate_tail = [&]() {
	const int lane = tid;
	PfRow row;
	pf_fold64(r.parts + (size_t)grp * 64 * kPfPart, gsize, lane, row);
	if (lane == 0) __hip_atomic_store(r.counter + 1 + grp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (ngroups > 1) {
		double *gp = r.gparts + (size_t)grp * kPfPart;
		if (lane == 0) { st_coh(gp, row.v); st_coh(gp + 1, (double)row.i); }
		if (lane < 16) st_coh(gp + 2 + lane, pf_pick(row.sum, lane));
		if (lane < 8) st_coh(gp + 18 + lane, pf_pick(row.st, lane));
		wait_stores_acked();
		int last = 0;
		if (lane == 0) last = __hip_atomic_fetch_add(r.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1;
		if (!__shfl(last, 0)) return;
		PfRow tot;
		pf_fold64(r.gparts, min(64, ngroups), lane, tot);
		for (int g0 = 64; g0 < ngroups; g0 += 64) {   /* more than 4096 workgroups (a million particles): passes of 64 group rows, combined in order */
			PfRow nx;
			pf_fold64(r.gparts + (size_t)g0 * kPfPart, min(64, ngroups - g0), lane, nx);
			if (nx.v > tot.v || (nx.v == tot.v && nx.i > tot.i)) {
				tot.v = nx.v; tot.i = nx.i;
#pragma unroll
				for (int q = 0; q < 8; ++q) tot.st[q] = nx.st[q];
			}
#pragma unroll
			for (int q = 0; q < 16; ++q) tot.sum[q] += nx.sum[q];
		}
		row = tot;
		if (lane == 0) __hip_atomic_store(r.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	/* lane q < 32 ends up holding out[q]: state estimate (8) | max_wt | max_wt_id | mean corners (8) | 0 ... */
	double o = 0.0;
	if (lane < 8) {
		if (r.mean_type == 1) o = lane < S ? pf_pick(row.sum, lane) / (double)n : 0.0;   /* estimateMeanOfSamples :311-317 */
		else o = lane < S ? pf_pick(row.st, lane) : 0.0;
	} else if (lane == 8) o = row.v;
	else if (lane == 9) o = (double)row.i;
	else if (lane >= 10 && lane < 18 && r.mean_type == 2) o = pf_pick(row.sum, lane - 2) / (double)n;   /* mean corners: sum slots 8..15 */
	if (lane < 32) r.out[lane] = o;
	if (r.pub.host) {
		if (lane < 32) __hip_atomic_store(r.pub.host + lane, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		/* (one wave: its write-through stores are performed, the flag is a posted write behind them -- no L2 write-back; fenced =
		 * MTFHIP_PUBLISH_FENCE=1, publish_fenced(): the system-scope release the memory model asks for) */
		if (r.pub.fenced) {
			__threadfence_system();
			if (lane == 0) __hip_atomic_store(r.pub.flag, r.pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		} else {
			wait_stores_acked();
			if (lane == 0) __hip_atomic_store(r.pub.flag, r.pub.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	};
	if (is_last && tid < 64) estimate_tail();
	}
	/* The next iteration's proposal of this thread's particle (PF.cc:207-245 one iteration ahead: a pure function of the resampled
	 * state and the particle's counter-based draws) -- AFTER the estimate has gone out: it is 5 of the 16 us of this launch at
	 * 10 000 particles (Philox + Box-Muller + the corner-based homography per particle), and the host, which only waits for the
	 * estimate, prepares and enqueues the next iteration meanwhile. */
	if (k < n && r.lookahead) {
		const bool use_ar = a.dynamic_model == 1;
		double ps[8], pa[8];
		if (have_pert) pf_dynamics<SSM>(a, pin, ns, nar, ps, pa);
		else pf_propose<SSM>(a, a.iter + 1, (unsigned)k, ns, nar, ps, pa);
		pf_store_row<S>(r.next, (size_t)k, ps);
		if (use_ar) pf_store_row<S>(r.next_ar, (size_t)k, pa);
	}
}

/* ---- residual resampling (PF.cc:538-582) ----
 * particle_wts /= particle_cum_wts[n - 1]; the particle indices EXCEPT THE LAST ONE are sorted by weight, highest first
 * (std::sort(idx, idx + n - 1): the reference's range ends one short); every particle in that order is copied
 * round(weight n) times until n slots are filled, the rest take the first of the order, which is also max_wt_id.
 * Here: normalise + keys (k_pf_residual_prep), a stable radix sort of the n - 1 keys (hipCUB; std::sort leaves the order of
 * equal weights unspecified, index order is one of its outcomes), copies (k_pf_residual_copies), their exclusive scan, and a
 * search of every destination slot in the scanned starts (k_pf_residual_map). */
__global__ __launch_bounds__(kBlock) void k_pf_residual_prep(int n, const double *total, double *wts, double *keys, int *idx, const int *flag) {
	const int k = blockIdx.x * kBlock + threadIdx.x;
	if (k >= n) return;
	const double w = wts[k] / *total;
	if (!flag || *flag) wts[k] = w;   /* (adaptive resampling: an iteration that does not resample keeps its weights as they are) */
	keys[k] = w; idx[k] = k;
}
__global__ __launch_bounds__(kBlock) void k_pf_residual_copies(int n, const double *wts, const int *order, int *copies) {
	const int j = blockIdx.x * kBlock + threadIdx.x;
	if (j >= n) return;
	copies[j] = (int)round(wts[order[j]] * (double)n);   /* static_cast<int>(round(particle_wts[resample_id] * n_particles)) */
}
__global__ __launch_bounds__(kBlock) void k_pf_residual_map(int n, const int *order, const int *copies, const int *starts, int *ids) {
	const int k = blockIdx.x * kBlock + threadIdx.x;
	if (k >= n) return;
	/* the last position j whose first slot is <= k (the starts are non-decreasing); it owns slot k if its copies reach that far */
	int lo = 0, hi = n;   /* first position with starts > k */
	while (lo < hi) { const int mid = (lo + hi) >> 1; if (starts[mid] > k) hi = mid; else lo = mid + 1; }
	const int j = lo - 1;
	const bool owned = j >= 0 && k < starts[j] + copies[j];
	ids[k] = owned ? order[j] : order[0];   /* leftovers: "duplicate particle with highest weight to get exactly same number again" */
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
static PfArgs pf_args(const PfLaunch &p, const PfBuffers &bf) {
	PfArgs a;
	a.n_distr = p.n_distr > 1 ? p.n_distr : 1;
	a.distr_sigma = bf.distr_sigma; a.distr_mean = bf.distr_mean; a.distr_cum = bf.distr_cum; a.distr_uniforms = p.distr_uniforms; a.distr_ids = bf.distr_ids;
	a.n = p.n; a.S = p.S; a.dynamic_model = p.dynamic_model; a.update_type = p.update_type; a.sampler = p.sampler; a.nz = p.nz; a.ar_coeff = p.ar_coeff;
	for (int k = 0; k < 8; ++k) { a.sigma[k] = p.sigma[k]; a.mean[k] = p.mean[k]; a.init_corners[k] = p.init_corners[k]; }
	for (int k = 0; k < 9; ++k) a.aux_inv[k] = p.aux_inv[k];
	for (int k = 0; k < 6; ++k) a.canon[k] = p.canon[k];
	a.seed = p.seed; a.iter = p.iter; a.normals = p.normals;
	return a;
}
void launch_pf_propose(int ssm, const PfLaunch &p, const PfBuffers &bf, const double *st_in, const double *ar_in, double *st_out, double *ar_out, hipStream_t st) {
	const PfArgs a = pf_args(p, bf);
	const dim3 g((p.n + kBlock - 1) / kBlock);
	if (ssm == MTFHIP_SSM_HOMOGRAPHY) MTFHIP_LAUNCH(k_pf_propose<MTFHIP_SSM_HOMOGRAPHY>, g, dim3(kBlock), 0, st, a, st_in, ar_in, st_out, ar_out);
	else MTFHIP_LAUNCH(k_pf_propose<MTFHIP_SSM_AFFINE>, g, dim3(kBlock), 0, st, a, st_in, ar_in, st_out, ar_out);
}
static void launch_pf_score_args(const BatchView &bv, const ImgView &im, const PfScoreArgs &s, int fast_math, hipStream_t st) {
	if (s.cnt <= 0) return;
	/* candidates per workgroup (see k_pf_score); MTFHIP_PF_K pins it (experiments) */
	static const int k_env = std::getenv("MTFHIP_PF_K") ? std::atoi(std::getenv("MTFHIP_PF_K")) : 0;
	const int kc = (k_env == 1 || k_env == 2 || k_env == 4) ? k_env : (s.cnt <= 1536 ? 1 : (s.cnt <= 4096 ? 2 : 4));
	const dim3 g((s.cnt + kc - 1) / kc);
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY, ncc = s.ncc_sc != nullptr, mc = bv.C > 1;
#define MTFHIP_PF_SCORE(SSM_, NCC_, FAST_, MC_, PAIR_) do { \
		if (kc == 1) MTFHIP_LAUNCH((k_pf_score<SSM_, NCC_, FAST_, MC_, 1, PAIR_>), g, dim3(kBlock), 0, st, bv, im, s); \
		else if (kc == 2) MTFHIP_LAUNCH((k_pf_score<SSM_, NCC_, FAST_, MC_, 2, PAIR_>), g, dim3(kBlock), 0, st, bv, im, s); \
		else MTFHIP_LAUNCH((k_pf_score<SSM_, NCC_, FAST_, MC_, 4, PAIR_>), g, dim3(kBlock), 0, st, bv, im, s); } while (0)
	const bool pair = fast_math && !mc && s.pair != nullptr;   /* tolerance mode, one channel: the cell from the row-pair copy of the frame */
#define MTFHIP_PF_SCORE_MC(SSM_, NCC_, FAST_) do { if (mc) MTFHIP_PF_SCORE(SSM_, NCC_, FAST_, true, false); \
		else if (pair && FAST_) MTFHIP_PF_SCORE(SSM_, NCC_, FAST_, false, (FAST_)); else MTFHIP_PF_SCORE(SSM_, NCC_, FAST_, false, false); } while (0)
	if (fast_math) {
		if (hom) { if (ncc) MTFHIP_PF_SCORE_MC(MTFHIP_SSM_HOMOGRAPHY, true, true); else MTFHIP_PF_SCORE_MC(MTFHIP_SSM_HOMOGRAPHY, false, true); }
		else { if (ncc) MTFHIP_PF_SCORE_MC(MTFHIP_SSM_AFFINE, true, true); else MTFHIP_PF_SCORE_MC(MTFHIP_SSM_AFFINE, false, true); }
	} else {
		if (hom) { if (ncc) MTFHIP_PF_SCORE_MC(MTFHIP_SSM_HOMOGRAPHY, true, false); else MTFHIP_PF_SCORE_MC(MTFHIP_SSM_HOMOGRAPHY, false, false); }
		else { if (ncc) MTFHIP_PF_SCORE_MC(MTFHIP_SSM_AFFINE, true, false); else MTFHIP_PF_SCORE_MC(MTFHIP_SSM_AFFINE, false, false); }
	}
#undef MTFHIP_PF_SCORE_MC
#undef MTFHIP_PF_SCORE
}
/* candidates [lo, lo + cnt) of `states`: weight and similarity at their global indices (the particle filter's scoring launch and
 * mtfhip_score_candidates: PF.cc:247-262, 341-365 per candidate) */
void launch_score_block(const BatchView &bv, const ImgView &im, const double *states, int lo, int cnt, double alpha, double norm_mult,
	double norm_add, const double *ncc_sc, double *wts, double *sim, int likelihood_func, double measurement_sigma, double max_similarity,
	int fast_math, const PfPeerPush *peer, const double *hull, const float *pair, hipStream_t st) {
	PfScoreArgs s;
	s.pair = (fast_math && bv.C == 1) ? pair : nullptr; s.pair_w = im.w;
	s.prop = states; s.lo = lo; s.cnt = cnt; s.alpha = alpha; s.norm_mult = norm_mult; s.norm_add = norm_add; s.ncc_sc = ncc_sc;
	s.likelihood_func = likelihood_func; s.measurement_sigma = measurement_sigma; s.max_similarity = max_similarity;
	s.wts = wts; s.sim = sim;
	if (peer) s.peer = *peer; else s.peer = PfPeerPush{};
	s.hull_ok = hull ? 1 : 0;
	{ static const bool lean_env = !(std::getenv("MTFHIP_PF_LEAN") && std::getenv("MTFHIP_PF_LEAN")[0] == '0'); s.lean_ok = lean_env ? 1 : 0; }
	for (int q = 0; q < 8; ++q) s.hull[q] = hull ? hull[q] : 0.0;
	launch_pf_score_args(bv, im, s, fast_math, st);
}
void launch_pf_peer_push(const PfPeerPush &peer, const double *wts, int lo, int cnt, hipStream_t st) {
	if (cnt <= 0) return;
	MTFHIP_LAUNCH(k_pf_peer_push, dim3((cnt + kPfPushPerGroup - 1) / kPfPushPerGroup), dim3(64), 0, st, peer, wts, lo, cnt);
}
void launch_pf_peer_wait(const PfPeerWait &w, hipStream_t st) { MTFHIP_LAUNCH(k_pf_peer_wait, dim3(1), dim3(kBlock), 0, st, w); }
void launch_pf_scan(const PfLaunch &p, const PfBuffers &bf, const PfPeerWait *wait, hipStream_t st) {
	const int nch = (p.n + kPfChunk - 1) / kPfChunk;
	PfScanArgs sc{p.n, nch, bf.wts, bf.cum, bf.sub16, bf.chunk_tot, bf.chunk_incl, bf.counters,
		bf.scan_stats, bf.distr_ids, p.n_distr > 1 ? p.n_distr : 1, p.min_distr_wt, p.min_eff_particles, bf.distr_cum, bf.distr_wts, bf.resample_flag,
		wait ? *wait : PfPeerWait{}};
	MTFHIP_LAUNCH(k_pf_scan, dim3((nch + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, st, sc);
}
void launch_pf_select(int ssm, const PfLaunch &p, const PfBuffers &bf, int lookahead, double *host_out, unsigned long long *host_flag,
	unsigned long long seq, const PfSelectPlan &plan, hipStream_t st) {
	const PfArgs a = pf_args(p, bf);
	PfSelectArgs r;
	r.resample_flag = bf.resample_flag;
	r.resampling_type = p.resampling_type; r.mean_type = p.mean_type; r.lookahead = lookahead; r.uniforms = p.uniforms;
	r.wts = bf.wts; r.cum = bf.cum; r.sub16 = bf.sub16; r.chunk_incl = bf.chunk_incl; r.prop = bf.prop; r.prop_ar = bf.prop_ar;
	r.st_out = bf.st; r.ar_out = bf.ar; r.next = bf.next; r.next_ar = bf.next_ar; r.ids = bf.ids;
	r.forced_best = p.resampling_type == 3 ? bf.res_order : nullptr;
	for (int k = 0; k < 12; ++k) r.init_corners_hm[k] = p.init_corners_hm[k];
	r.parts = bf.parts; r.gparts = bf.gparts; r.counter = bf.counters + 1; r.out = bf.out; r.pub = PfPublish{host_out, host_flag, seq, publish_fenced()};
	const int nsel = (p.n + kBlock - 1) / kBlock;
	r.estimate = plan.estimate; r.nsel = nsel; r.pert_in = plan.pert_in; r.pert_out = plan.pert_out; r.pert_out_iter = p.iter + 2;
	r.wait = plan.wait ? *plan.wait : PfPeerWait{};
	const dim3 g(plan.pert_out ? 2 * nsel : nsel);
	const bool hom = ssm == MTFHIP_SSM_HOMOGRAPHY;
	if (plan.local) {
		if (hom) MTFHIP_LAUNCH((k_pf_select<MTFHIP_SSM_HOMOGRAPHY, true>), g, dim3(kBlock), 0, st, a, r);
		else MTFHIP_LAUNCH((k_pf_select<MTFHIP_SSM_AFFINE, true>), g, dim3(kBlock), 0, st, a, r);
	} else {
		if (hom) MTFHIP_LAUNCH((k_pf_select<MTFHIP_SSM_HOMOGRAPHY, false>), g, dim3(kBlock), 0, st, a, r);
		else MTFHIP_LAUNCH((k_pf_select<MTFHIP_SSM_AFFINE, false>), g, dim3(kBlock), 0, st, a, r);
	}
}
int pf_local_max() { return kPfLocalMax; }
void launch_pf_fill(int n, int S, const double *dev_state, double *states, double *ars, hipStream_t st) {
	MTFHIP_LAUNCH(k_pf_fill, dim3((n + 255) / 256), dim3(256), 0, st, n, S, dev_state, states, ars);
}
void launch_pf_residual_prep(int n, const double *total, double *wts, double *keys, int *idx, const int *flag, hipStream_t st) {
	MTFHIP_LAUNCH(k_pf_residual_prep, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, total, wts, keys, idx, flag);
}
void launch_pf_residual_copies(int n, const double *wts, const int *order, int *copies, hipStream_t st) {
	MTFHIP_LAUNCH(k_pf_residual_copies, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, wts, order, copies);
}
void launch_pf_residual_map(int n, const int *order, const int *copies, const int *starts, int *ids, hipStream_t st) {
	MTFHIP_LAUNCH(k_pf_residual_map, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, n, order, copies, starts, ids);
}
#ifdef MTFHIP_PF_TRACE
void debug_pf_trace(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pf_trace), sizeof(unsigned long long) * 16); }
#endif
int pf_parts_per_block() { return kPfPart; }
int pf_chunk() { return kPfChunk; }

} // namespace mtfhip
#ifdef MTFHIP_PF_TRACE
namespace mtfhip { void debug_pf_trace(unsigned long long *out); }
extern "C" void mtfhip_debug_pf_trace(unsigned long long *out) { mtfhip::debug_pf_trace(out); }
#endif
