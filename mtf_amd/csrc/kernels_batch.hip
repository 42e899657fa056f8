/*
 * kernels_batch.hip -- the batch axis: NN candidate sampling, GridTracker's one-launch ICLK patch loop (candidate scoring: k_pf_score,
 * kernels_pf.hip)
 * (one of the translation units of libmtfhip.so; conventions and the shared device helpers: mtfhip_device.h)
 */
#include "mtfhip_device.h"
#include "mtfhip_grid_device.h"

namespace mtfhip {

/* ===================================================================== */
/* one-launch inverse-compositional tracker for small patches (GridTracker) */
/* ===================================================================== */
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

#ifdef MTFHIP_GRID_TRACE   /* tools/grid_trace.sh: wall-clock stamps (100 MHz) of one workgroup's phases */
__device__ unsigned long long g_grid_trace[32];
#define GRID_STAMP(k) do { if (blockIdx.x == 100 && threadIdx.x == 0) g_grid_trace[k] = wall_clock64(); } while (0)
#else
#define GRID_STAMP(k) do { } while (0)
#endif
template <int AM, int PPT, bool FAST>
__global__ __launch_bounds__(kBlock) void k_iclk_track(BatchView bv, ImgView im, mtfhip_sm_desc sm, TrackState ts,
	const double *h0inv_all, const double *ncc_sc_all, double norm_mult, double norm_add, HostPublish pub, RegionIngest rg) {
	__shared__ double red[4 * 8];
	__shared__ double sW[9], sSt[8], sHinv[64], sH8[64], sIc[12], sCr[8], sNc[8];
	__shared__ int sDone;
	GRID_STAMP(0);
	const int t = blockIdx.x, N = bv.N, S = bv.S, tid = threadIdx.x;
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY;
	const bool region = rg.corners != nullptr;   /* (uniform: a kernel argument) */
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * N;
	const double2 *ih = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_HXY]) + (size_t)t * N;
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * N;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N;
	const double *J0 = bv.buf[MTFHIP_BUF_J0] + (size_t)t * N * S;
	/* region mode: the patch's region and its template's NCC scalars from the pinned staging buffer -- one PCIe round trip per workgroup.
	 * The load is issued here, its LDS store (which waits for it) only behind the template operands' loads below, so that the two
	 * latencies overlap (program order is wait order: stored right away, the PCIe read was 1.1 us in front of everything else) */
	/* layout mode (RegionIngest::layout, r05): the patch's corners are computed HERE from the grid's region (kernel arguments: no memory
	 * round trip at all), and in tolerance mode the NCC scalars' PCIe read is not waited for before the first reduction of the loop */
	const bool lay = region && rg.layout != 0;
	const bool late_ncc = FAST && lay && AM == MTFHIP_AM_NCC;
	double q8lay[8];
#pragma unroll
	for (int q = 0; q < 8; ++q) q8lay[q] = 0.0;
	if (lay) grid_patch_corners_lanes(rg.grid, rg.region_map, t, q8lay);   /* (mtfhip_device.h) */
	double ingest = 0.0;
	if (region && tid < 16 && !(lay && tid < 8)) ingest = tid < 8 ? rg.corners[8 * (size_t)t + tid] : (AM == MTFHIP_AM_NCC ? rg.ncc[8 * (size_t)t + tid - 8] : 0.0);
	double m0 = (AM == MTFHIP_AM_NCC && !region) ? ncc_sc_all[t * 8 + 0] : 0.0;
	double cn = (AM == MTFHIP_AM_NCC && !region) ? ncc_sc_all[t * 8 + 1] : 1.0;
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
			if (!region) {
				hpv[k] = bv.unit_z ? ip[ic] : ih[ic];
				zv[k] = bv.unit_z ? 1.0 : iz[ic];
			}
		}
		i0v[k] = i < N ? I0[ic] : 0.0;
		if constexpr (HOIST_J) {
#pragma unroll
			for (int s = 0; s < 8; ++s) j0v[k][s] = (s < S && i < N) ? J0[(size_t)s * N + ic] : 0.0;
		}
	}
	/* NCC, tolerance mode: sum J0 | sum I0 J0 of this patch's template, reduced when the template was (ncc_template_moments: sum J0 | sum
	 * I0 J0 | Gram).  Requested HERE with the other operands (r05): they are uniform (scalar) loads, and behind the grid layout their
	 * round trip sat in front of the loop -- the first LDS read after them waits on the same counter */
	double tc[16];
#pragma unroll
	for (int q = 0; q < 16; ++q) tc[q] = 0.0;
	if constexpr (FAST && AM == MTFHIP_AM_NCC) {
		if (ts.ncc_tm) {
#pragma unroll
			for (int q = 0; q < 16; ++q) tc[q] = ts.ncc_tm[(size_t)t * 52 + q];
		}
	}
	if (tid < 64) sHinv[tid] = h0inv_all[(size_t)t * 64 + tid];
	if (tid < 64) { const int r = tid >> 3, c = tid & 7; sH8[tid] = (r < S && c < S) ? h0inv_all[(size_t)t * 64 + c * S + r] : 0.0; }   /* [r][c], zero padded */
	asm volatile("" ::: "memory");   /* (the loads above are issued before the ingest is waited for) */
	if (region && tid < 16) { if (tid < 8) { if (!lay) sCr[tid] = ingest; } else if (!late_ncc) sNc[tid - 8] = ingest; }
	if (lay && tid == 0) {
#pragma unroll
		for (int q = 0; q < 8; ++q) sCr[q] = q8lay[q];
	}
	if (!region) {
		if (tid < 12) sIc[tid] = ts.init_corners_hm[12 * t + tid];
		if (ts.fresh_reset) {   /* (uniform) behind a fused re-initialisation: identity warp, zero state, the template's own corners (x, y, 1 per corner) */
			if (tid < 8) sCr[tid] = ts.init_corners_hm[12 * t + 3 * (tid >> 1) + (tid & 1)];
			if (tid < 9) sW[tid] = (tid == 0 || tid == 4 || tid == 8) ? 1.0 : 0.0;
			if (tid < 8) sSt[tid] = 0.0;
		} else {
			if (tid < 8) sCr[tid] = ts.corners[8 * t + tid];
			if (tid < 9) sW[tid] = bv.warps[9 * t + tid];
			if (tid < 8) sSt[tid] = bv.states[8 * t + tid];
		}
	} else {
		/* the SSM's reset (setCorners: identity warp, zero state, init_corners_hm = (x, y, 1)) */
		if (tid < 9) sW[tid] = (tid == 0 || tid == 4 || tid == 8) ? 1.0 : 0.0;
		if (tid < 8) sSt[tid] = 0.0;
	}
	if (tid == 0) sDone = 0;
	GRID_STAMP(1);
	__syncthreads();
	GRID_STAMP(2);
	int n_it = 0;
	double f_last = 0;
	bool region_bad = false;
	/* Tolerance mode reads the texels of the whole loop from an LDS window around the corners (below).  Where the window sits depends on
	 * the corners only, so its fetch is REQUESTED here, in front of the map's divisions and the grid layout (~1.2 us of dependent FP64 in
	 * region mode), and stored behind them: the round trip runs under that arithmetic instead of after it (r05). */
	constexpr int kWinW = 64, kWinH = 64;
	int wx0 = 0, wy0 = 0;
	bool win_ok = false;
	float win_tv[FAST ? kWinW * kWinH / kBlock : 1];
#ifndef MTFHIP_GRID_NO_WINDOW
	if constexpr (FAST) {
		const double c0 = sCr[0], c1 = sCr[1], c2 = sCr[2], c3 = sCr[3], c4 = sCr[4], c5 = sCr[5], c6 = sCr[6], c7 = sCr[7];
		const double mnx = fmin(fmin(c0, c2), fmin(c4, c6)), mxx = fmax(fmax(c0, c2), fmax(c4, c6));
		const double mny = fmin(fmin(c1, c3), fmin(c5, c7)), mxy = fmax(fmax(c1, c3), fmax(c5, c7));
		const double cxm = 0.5 * (mnx + mxx), cym = 0.5 * (mny + mxy);
		/* (NaN corners fail every comparison) */
		win_ok = (mxx - mnx < kWinW - 6) & (mxy - mny < kWinH - 6) & (cxm > -1e6) & (cxm < 1e6) & (cym > -1e6) & (cym < 1e6) &
			(im.w >= kWinW) & (im.h >= kWinH);
		if (win_ok) {
			wx0 = min(max((int)floor(cxm) - kWinW / 2, 0), im.w - kWinW);
			wy0 = min(max((int)floor(cym) - kWinH / 2, 0), im.h - kWinH);
#pragma unroll
			for (int j = 0; j < kWinW * kWinH / kBlock; ++j) {
				const int idx = tid + j * kBlock;
				win_tv[j] = im.data[(unsigned)((wy0 + idx / kWinW) * im.stride + wx0 + idx % kWinW)];
			}
		}
	}
#endif
	if (region) {
		static_assert(PPT <= 8, "region mode keeps the grid in registers");
		/* every thread derives the same map from the same eight numbers (~40 flops and a dozen divisions: cheaper than a broadcast
		 * and its barrier), then its own grid points: ProjectiveBase::getPtsFromCorners + Homography / Affine::setCorners exactly as
		 * k_init_grid lays them out (same expressions, same order: the loop that follows must not depend on who built the grid) */
		double q8[8], W0[9];
#pragma unroll
		for (int q = 0; q < 8; ++q) q8[q] = sCr[q];
		region_bad = !rect_to_quad_hd(rg.lo_x, rg.lo_y, rg.hi_x, rg.hi_y, q8, W0);
		if (region_bad) {
#pragma unroll
			for (int q = 0; q < 9; ++q) W0[q] = (q == 0 || q == 4 || q == 8) ? 1.0 : 0.0;
		}
		/* (set_corners_core: a homography grid whose projective entries vanish is laid out with exact zeros) */
		if (hom && fabs(W0[6]) < 1e-15 && fabs(W0[7]) < 1e-15) { W0[6] = 0; W0[7] = 0; }
		if constexpr (AM == MTFHIP_AM_NCC) { if (!late_ncc) { m0 = sNc[0]; cn = sNc[1]; } }
		double2 *ipw = const_cast<double2 *>(ip), *ihw = const_cast<double2 *>(ih);
		double *izw = const_cast<double *>(iz);
#pragma unroll
		for (int k = 0; k < PPT; ++k) {
			const int i = tid + k * kBlock;
			const int ic = i < N ? i : N - 1;
			const int col = ic % rg.resx, row = ic / rg.resx;
			const double nx = (rg.resx == 1 || col == rg.resx - 1) ? rg.hi_x : rg.lo_x + col * ((rg.hi_x - rg.lo_x) / (rg.resx - 1));
			const double ny = (rg.resy == 1 || row == rg.resy - 1) ? rg.hi_y : rg.lo_y + row * ((rg.hi_y - rg.lo_y) / (rg.resy - 1));
			const double X = W0[0] * nx + W0[1] * ny + W0[2] * 1.0;
			const double Y = W0[3] * nx + W0[4] * ny + W0[5] * 1.0;
			const double Z = W0[6] * nx + W0[7] * ny + W0[8] * 1.0;
			/* (a parallelogram's map has Z = 1.0 exactly and x / 1.0 == x: the two divisions per point are skipped, same bits) */
			const double2 p = (W0[6] == 0 && W0[7] == 0 && W0[8] == 1.0) ? make_double2(X, Y) : make_double2(X / Z, Y / Z);
			const double z = rg.force_unit_z ? 1.0 : Z;
			const double2 hxy = rg.force_unit_z ? p : make_double2(X, Y);
			if constexpr (HOIST_P) { hpv[k] = bv.unit_z ? p : hxy; zv[k] = bv.unit_z ? 1.0 : z; }
			if (i < N) { ipw[i] = p; izw[i] = z; ihw[i] = hxy; }
		}
		/* the slab entries the ingest used to bring: this workgroup's piece, for the calls that come after the frame */
		if (tid < 9) rg.d_w0[9 * (size_t)t + tid] = W0[tid];
		if (tid < 12) { const double v = (tid % 3 == 2) ? 1.0 : q8[2 * (tid / 3) + tid % 3]; sIc[tid] = v; rg.d_init_corners_hm[12 * (size_t)t + tid] = v; }
		if (AM == MTFHIP_AM_NCC && tid < 8 && !late_ncc) rg.d_ncc[8 * (size_t)t + tid] = sNc[tid];
		__syncthreads();   /* sIc */
	}
	if constexpr (FAST) {
		/* Tolerance mode: ONE workgroup-wide reduction per iteration instead of four (NCC) / two (SSD), and no serial section.
		 * NCC's similarity and df_dI0 . J0 are functions of raw moments -- sum It, sum It^2, sum I0 It, sum It J0 plus the
		 * template's constants sum J0, sum I0 J0 (the fused NCC kernel's algebra, api_fused.hip::ncc_assemble) -- so the two-pass
		 * mean / norm / gradient-mean chain of NCC.cc:124-194 collapses into one pass; SSD accumulates r^2 and r J0 together.
		 * The S x S product with the pre-inverted Hessian, the inverse compositional update and the corner test are then
		 * evaluated redundantly by every thread from the broadcast sums: no thread-0 section, no barrier behind it, the warp stays
		 * in registers.  An iteration is two barriers instead of nine. */
		constexpr bool NCC = AM == MTFHIP_AM_NCC;
		constexpr int K = NCC ? 11 : 9;
		__shared__ double redk[2 * 4 * 16];   /* (two buffers: block_allsum_h12) */
		double W[9], St[8], Cr[8];
#pragma unroll
		for (int q = 0; q < 9; ++q) W[q] = sW[q];
#pragma unroll
		for (int q = 0; q < 8; ++q) { St[q] = sSt[q]; Cr[q] = sCr[q]; }
		if constexpr (NCC) {
			if (!ts.ncc_tm) {
#pragma unroll
				for (int k = 0; k < PPT; ++k) {
					const int i = tid + k * kBlock;
					if (i < N) {
#pragma unroll
						for (int s = 0; s < 8; ++s) {
							const double j = s < S ? (HOIST_J ? j0v[HOIST_J ? k : 0][s] : J0[(size_t)s * N + i]) : 0.0;
							tc[s] += j; tc[8 + s] = fma(i0v[k], j, tc[8 + s]);
						}
					}
				}
				block_allsum<16>(tc, redk);
			}
		}
		/* The texels of the whole loop from LDS: the patch moves by a fraction of its size over the iterations, so a window of the frame
		 * around its corners (kWinW x kWinH texels, centred on their bounding box, clamped into the image) is fetched ONCE -- one global
		 * round trip, what the first iteration's texel fetch cost anyway -- and every later iteration reads its four texels per sample
		 * from LDS (~100 ns) instead of L2 (~1 us of every 3.3 us iteration, section 4.4's phase trace).  A wave any of whose samples
		 * leaves the window (or a patch larger than it, or a frame smaller) takes the global path: same texels, same arithmetic, same bits. */
		__shared__ float win[kWinW * kWinH];
#ifndef MTFHIP_GRID_NO_WINDOW
		win_ok = win_ok & !region_bad;
		if (win_ok) {
#pragma unroll
			for (int j = 0; j < kWinW * kWinH / kBlock; ++j) win[tid + j * kBlock] = win_tv[j];
		}
		__syncthreads();   /* (win_ok is uniform: every thread holds the same corners) */
#endif
		/* constant over the iterations: this lane's row of H0^-1 and the template's homogeneous corners -- one LDS round trip each per
		 * iteration on a loop that is a chain of dependent latencies (r05) */
		double hrow[8], Ic[12];
#pragma unroll
		for (int q = 0; q < 8; ++q) hrow[q] = sH8[8 * (tid & 7) + q];
#pragma unroll
		for (int q = 0; q < 12; ++q) Ic[q] = sIc[q];
		const double winx0 = (double)wx0, winx1 = (double)(wx0 + kWinW - 1), winy0 = (double)wy0, winy1 = (double)(wy0 + kWinH - 1);
		const double nN = (double)N, inv_n = 1.0 / nN;
		double inv_cn = 1.0 / cn;
		GRID_STAMP(3);
		const int max_it = region_bad ? 0 : sm.max_iters;   /* degenerate region corners: no iteration, n_iters = -1 tells the host */
		for (int it = 0; it < max_it; ++it) {
			if (it < 12) GRID_STAMP(4 + it);
			double m[12];   /* K sums, padded to the twelve block_allsum_h12 takes */
#pragma unroll
			for (int q = 0; q < 12; ++q) m[q] = 0.0;
#pragma unroll
			for (int k = 0; k < PPT; ++k) {
				const int i = tid + k * kBlock;
				const int ick = i < N ? i : N - 1;
				const double2 hp = HOIST_P ? hpv[HOIST_P ? k : 0] : (bv.unit_z ? ip[ick] : ih[ick]);
				const double z = HOIST_P ? zv[HOIST_P ? k : 0] : (bv.unit_z ? 1.0 : iz[ick]);
				double wx = fma(W[0], hp.x, fma(W[1], hp.y, W[2] * z)), wy = fma(W[3], hp.x, fma(W[4], hp.y, W[5] * z));
				if (hom) { const double inv = rcp_fast(fma(W[6], hp.x, fma(W[7], hp.y, W[8] * z))); wx *= inv; wy *= inv; }
#if defined(MTFHIP_GRID_ABL) && (MTFHIP_GRID_ABL & 1)   /* ablation builds (tools/grid_ablation.sh): no texel fetch */
				const double v = i < N ? wx + wy : 0.0;
#else
				double pv;
				const bool inw = win_ok & (wx >= winx0) & (wx < winx1) & (wy >= winy0) & (wy < winy1);   /* lx + 1, ly + 1 inside as well */
				if (__builtin_amdgcn_ballot_w64(!inw) == 0) {
					const int lx = (int)wx, ly = (int)wy;
					const float *wp = win + ((ly - wy0) * kWinW + (lx - wx0));
					pv = bilin_val_fast(wp[0], wp[1], wp[kWinW], wp[kWinW + 1], wx - (double)lx, wy - (double)ly);
				} else pv = pix_val_fast(im, wx, wy);
				const double v = i < N ? fma(norm_mult, pv, norm_add) : 0.0;
#endif
				const double i0 = i0v[k];
				if constexpr (NCC) {
					m[0] += v; m[1] = fma(v, v, m[1]); m[2] = fma(i0, v, m[2]);
				} else {
					const double r = i < N ? v - i0 : 0.0;
					m[0] = fma(r, r, m[0]);
				}
				const double wgt = NCC ? v : (i < N ? v - i0 : 0.0);
#pragma unroll
				for (int s = 0; s < 8; ++s)
					if (s < S) m[K - 8 + s] = fma(wgt, HOIST_J ? j0v[HOIST_J ? k : 0][s] : J0[(size_t)s * N + ick], m[K - 8 + s]);
			}
			/* (layout mode: the template's NCC scalars come over PCIe and are first needed behind this reduction, whose barrier also
			 * publishes them to the workgroup -- by now the read has long returned) */
			if (late_ncc && it == 0 && tid >= 8 && tid < 16) sNc[tid - 8] = ingest;
#if !(defined(MTFHIP_GRID_ABL) && (MTFHIP_GRID_ABL & 2))   /* ablation: no workgroup reduction */
#ifdef MTFHIP_GRID_DPP_REDUCE   /* (r03 / early r04: one DPP wave sum per value) */
			block_allsum_dpp<K>(m, redk);
#else
			block_allsum_h12(m, redk + ((it + 1) & 1) * 64);   /* (round 0 in the buffer the template sums above did not use) */
#endif
#endif
			if (late_ncc && it == 0) {
				m0 = sNc[0]; cn = sNc[1]; inv_cn = 1.0 / cn;
				if (tid < 8) rg.d_ncc[8 * (size_t)t + tid] = sNc[tid];
			}
			double g[8];
			if constexpr (NCC) {
				/* (the whole section runs on one wave per SIMD with nothing to overlap: an IEEE division is ~30 dependent
				 * instructions, so the ~40 of the straightforward form were 3 us of every 7 us iteration -- reciprocals once) */
				/* (r05: 1 / b from v_rsq_f64 + two Newton steps and b = b2 / b, instead of sqrt -- itself a reciprocal square root with
				 * corrections -- followed by a reciprocal: ~15 dependent instructions less) */
				const double mt = m[0] * inv_n, b2 = fma(-nN * mt, mt, m[1]);
				const double inv_b = rsq_fast(b2), b = b2 * inv_b;
				const double inv_bc = inv_b * inv_cn, inv_b2 = inv_b * inv_b;
				const double f = fma(-nN * m0, mt, m[2]) * inv_bc;
				f_last = f;
				const double b_c = b * inv_cn;
#pragma unroll
				for (int s = 0; s < 8; ++s) {
					const double ut = fma(-mt, tc[s], m[3 + s]) * inv_b2, u0 = fma(-m0, tc[s], tc[8 + s]) * inv_bc;
					g[s] = b_c * fma(-f, u0, ut);   /* df_dI0 . J0, NCC.cc:163-194, 236-250 in moments */
				}
			} else {
				f_last = -m[0] / 2;
#pragma unroll
				for (int s = 0; s < 8; ++s) g[s] = m[1 + s];
			}
#if defined(MTFHIP_GRID_ABL) && (MTFHIP_GRID_ABL & 4)   /* ablation: no solve / update */
			W[2] += g[0] * 1e-30; ++n_it; continue;
#endif
			/* dp = -H0^-1 g, invertState, compositionalUpdate, corner test (NT/ICLK.cc:253-290) -- every thread, same values.
			 * This is a chain of dependent FP64 operations on a wave that has its SIMD to itself (~20 cycles per dependent
			 * operation): it is written for DEPTH -- pairwise sums, reciprocals instead of divisions, the affine inverse in closed
			 * form -- not for operation count (the straightforward form was 2.2 us of a 4.8 us iteration). */
			double dp[8];
#ifdef MTFHIP_GRID_REDUNDANT_SOLVE   /* (every thread forms all eight rows: 64 LDS reads + 120 FP64 instructions per iteration) */
#pragma unroll
			for (int r = 0; r < 8; ++r) {
				const double *h = sH8 + 8 * r;
				dp[r] = -(((h[0] * g[0] + h[1] * g[1]) + (h[2] * g[2] + h[3] * g[3])) + ((h[4] * g[4] + h[5] * g[5]) + (h[6] * g[6] + h[7] * g[7])));
			}
#else
			{
				/* lane l forms row l & 7 of -H0^-1 g (the same expression as above, so the same bits) and the eight results come back
				 * through the scalar unit: 8 LDS reads + 15 FP64 instructions + 16 v_readlane instead of 64 + 120 */
				const double *h = hrow;
				const double mine = -(((h[0] * g[0] + h[1] * g[1]) + (h[2] * g[2] + h[3] * g[3])) + ((h[4] * g[4] + h[5] * g[5]) + (h[6] * g[6] + h[7] * g[7])));
#pragma unroll
				for (int r = 0; r < 8; ++r) dp[r] = readlane_f64(mine, r);
			}
#endif
			double Wn[9];
			if (hom) {
				const double U0 = 1 + dp[0], U1 = dp[1], U2 = dp[2], U3 = dp[3], U4 = 1 + dp[4], U5 = dp[5], U6 = dp[6], U7 = dp[7];
				/* cofactors of U (U8 = 1); inverse / its (2, 2) entry: the determinant cancels */
				const double c0 = U4 - U5 * U7, c1 = U2 * U7 - U1, c2 = U1 * U5 - U2 * U4;
				const double c3 = U5 * U6 - U3, c4 = U0 - U2 * U6, c5 = U2 * U3 - U0 * U5;
				const double c6 = U3 * U7 - U4 * U6, c7 = U1 * U6 - U0 * U7, c8 = U0 * U4 - U1 * U3;
				const double ic8 = rcp_fast(c8);
				const double V[9] = {c0 * ic8, c1 * ic8, c2 * ic8, c3 * ic8, c4 * ic8, c5 * ic8, c6 * ic8, c7 * ic8, 1.0};
#pragma unroll
				for (int r = 0; r < 3; ++r)
#pragma unroll
					for (int c = 0; c < 3; ++c) Wn[3 * r + c] = fma(W[3 * r], V[c], fma(W[3 * r + 1], V[3 + c], W[3 * r + 2] * V[6 + c]));
				const double inv_w22 = rcp_fast(Wn[8]);
#pragma unroll
				for (int q = 0; q < 8; ++q) Wn[q] *= inv_w22;
				Wn[8] = 1;
				St[0] = Wn[0] - 1; St[1] = Wn[1]; St[2] = Wn[2]; St[3] = Wn[3]; St[4] = Wn[4] - 1; St[5] = Wn[5]; St[6] = Wn[6]; St[7] = Wn[7];
			} else {
				/* affine: U = [a b tx; c d ty; 0 0 1], inverse in closed form (Affine.cc:145-150) */
				const double a = 1 + dp[2], b = dp[3], tx = dp[0], c = dp[4], d = 1 + dp[5], ty = dp[1];
				const double idet = rcp_fast(a * d - b * c);
				const double ia = d * idet, ib = -b * idet, ic = -c * idet, id = a * idet;
				const double itx = -(ia * tx + ib * ty), ity = -(ic * tx + id * ty);
				Wn[0] = W[0] * ia + W[1] * ic; Wn[1] = W[0] * ib + W[1] * id; Wn[2] = (W[0] * itx + W[1] * ity) + W[2];
				Wn[3] = W[3] * ia + W[4] * ic; Wn[4] = W[3] * ib + W[4] * id; Wn[5] = (W[3] * itx + W[4] * ity) + W[5];
				Wn[6] = 0; Wn[7] = 0; Wn[8] = 1;
				St[0] = Wn[2]; St[1] = Wn[5]; St[2] = Wn[0] - 1; St[3] = Wn[1]; St[4] = Wn[3]; St[5] = Wn[4] - 1; St[6] = 0; St[7] = 0;
			}
			double ch[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const double X = Ic[3 * q], Y = Ic[3 * q + 1], Z = Ic[3 * q + 2];
				double nx = (Wn[0] * X + Wn[1] * Y) + Wn[2] * Z, ny = (Wn[3] * X + Wn[4] * Y) + Wn[5] * Z;
				if (hom) { const double idn = rcp_fast((Wn[6] * X + Wn[7] * Y) + Wn[8] * Z); nx *= idn; ny *= idn; }
				const double ddx = Cr[2 * q] - nx, ddy = Cr[2 * q + 1] - ny;
				ch[q] = ddx * ddx + ddy * ddy;
				Cr[2 * q] = nx; Cr[2 * q + 1] = ny;
			}
			const double change = (ch[0] + ch[1]) + (ch[2] + ch[3]);
			if (ts.trace && n_it < ts.trace_cap && tid == 0) {   /* debug trace (mtfhip_batch_track_trace); the Hessian is the constant H0 */
				double *trec = ts.trace + ((size_t)t * ts.trace_cap + n_it) * kTraceStride;
#pragma unroll
				for (int q = 0; q < 8; ++q) { trec[64 + q] = g[q]; trec[72 + q] = dp[q]; trec[80 + q] = Cr[q]; }
				trec[88] = f_last; trec[89] = (double)n_it; trec[90] = 0.0; trec[91] = 0.0; trec[92] = 0.0;
			}
#pragma unroll
			for (int q = 0; q < 9; ++q) W[q] = Wn[q];
			++n_it;
			if (change < sm.epsilon) break;   /* uniform: every thread holds the same numbers */
		}
		GRID_STAMP(16);
		/* every thread holds the same W / St / Cr: lane q takes entry q through selects (register arrays cannot be indexed by the lane
		 * id) and ONE store per array goes out.  (r03 stored entry by entry under `if (tid == q)`: 25 one-lane stores whose address
		 * register the compiler reused, each behind an s_waitcnt vmcnt(0) for the one before -- a chain of store round trips that was
		 * 5 us between the last iteration and the publish, r04 phase trace.) */
		if (region_bad) n_it = -1;
		if (tid < 64) {
			double wq = 0, sq = 0, cq = 0;
#pragma unroll
			for (int q = 0; q < 9; ++q) wq = tid == q ? W[q] : wq;
#pragma unroll
			for (int q = 0; q < 8; ++q) { sq = tid == q ? St[q] : sq; cq = tid == q ? Cr[q] : cq; }
			GRID_STAMP(17);
			if (pub.host) publish_target(pub, t, wq, sq, cq, n_it);   /* the host's copy first: it is what the caller waits for */
			if (tid < 9) bv.warps[9 * t + tid] = wq;
			if (tid < 8) { bv.states[8 * t + tid] = sq; ts.corners[8 * t + tid] = cq; }
			if (tid == 0) { ts.n_iters[t] = n_it; ts.acc[(size_t)t * ACC_COUNT + ACC_RR] = f_last; }
			GRID_STAMP(18);
		}
		return;
	}
	for (int it = 0; it < (region_bad ? 0 : sm.max_iters); ++it) {
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
			double wx, wy, v;   /* (replay arithmetic: the tolerance-mode loop above has returned) */
			if (hom) {
				const double cx = W[0] * hp.x + W[1] * hp.y + W[2] * z, cy = W[3] * hp.x + W[4] * hp.y + W[5] * z;
				const double d = W[6] * hp.x + W[7] * hp.y + W[8] * z;
				wx = cx / d; wy = cy / d;
			} else {
				wx = W[0] * hp.x + W[1] * hp.y + W[2] * z; wy = W[3] * hp.x + W[4] * hp.y + W[5] * z;
			}
			v = norm_mult * pix_val_select(im, wx, wy) + norm_add;
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
			if (ts.trace && n_it < ts.trace_cap) {
				double *trec = ts.trace + ((size_t)t * ts.trace_cap + n_it) * kTraceStride;
				for (int q = 0; q < 8; ++q) { trec[64 + q] = g[q]; trec[72 + q] = dp[q]; trec[80 + q] = sCr[q]; }
				trec[88] = f_last; trec[89] = (double)n_it; trec[90] = 0.0; trec[91] = 0.0; trec[92] = 0.0;
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
	if (region_bad) n_it = -1;
	if (tid == 0) { ts.n_iters[t] = n_it; ts.acc[(size_t)t * ACC_COUNT + ACC_RR] = f_last; }
	if (pub.host && tid < 64) publish_target(pub, t, tid < 9 ? sW[tid] : 0.0, tid < 8 ? sSt[tid] : 0.0, tid < 8 ? sCr[tid] : 0.0, n_it);
}


/* ===================================================================== */
/* launchers                                                              */
/* ===================================================================== */

template <int AM, bool FAST>
static bool launch_iclk_track_am(const BatchView &bv, const ImgView &im, const mtfhip_sm_desc &sm, const TrackState &ts,
	const double *h0inv, const double *ncc_sc, double norm_mult, double norm_add, const HostPublish &pub, const RegionIngest &rg, hipStream_t st) {
	const int ppt = (bv.N + kBlock - 1) / kBlock;
#define MTFHIP_ICLK_CASE(P) MTFHIP_LAUNCH((k_iclk_track<AM, P, FAST>), dim3(bv.B), dim3(kBlock), 0, st, bv, im, sm, ts, h0inv, ncc_sc, norm_mult, norm_add, pub, rg)
	if (ppt <= 1) MTFHIP_ICLK_CASE(1);
	else if (ppt <= 2) MTFHIP_ICLK_CASE(2);
	else if (ppt <= 3) MTFHIP_ICLK_CASE(3);
	else if (ppt <= 4) MTFHIP_ICLK_CASE(4);
	else if (ppt <= 8) MTFHIP_ICLK_CASE(8);
	else return false;
#undef MTFHIP_ICLK_CASE
	return true;
}
bool launch_iclk_track(const BatchView &bv, const ImgView &im, const mtfhip_sm_desc &sm, const TrackState &ts,
	const double *h0inv, const double *ncc_sc, double norm_mult, double norm_add, int fast_math, const HostPublish &pub, const RegionIngest &rg, hipStream_t st) {
	if (fast_math) {
		if (bv.am == MTFHIP_AM_NCC) return launch_iclk_track_am<MTFHIP_AM_NCC, true>(bv, im, sm, ts, h0inv, ncc_sc, norm_mult, norm_add, pub, rg, st);
		return launch_iclk_track_am<MTFHIP_AM_SSD, true>(bv, im, sm, ts, h0inv, ncc_sc, norm_mult, norm_add, pub, rg, st);
	}
	if (bv.am == MTFHIP_AM_NCC) return launch_iclk_track_am<MTFHIP_AM_NCC, false>(bv, im, sm, ts, h0inv, ncc_sc, norm_mult, norm_add, pub, rg, st);
	return launch_iclk_track_am<MTFHIP_AM_SSD, false>(bv, im, sm, ts, h0inv, ncc_sc, norm_mult, norm_add, pub, rg, st);
}
void launch_sample_candidates(const BatchView &bv, const ImgView &im, const double *dev_states, int C, double norm_mult,
	double norm_add, double *dev_feat, hipStream_t st) {
	MTFHIP_LAUNCH(k_sample_candidates, dim3(C), dim3(kBlock), 0, st, bv, im, dev_states, C, norm_mult, norm_add, dev_feat);
	if (bv.am == MTFHIP_AM_NCC) MTFHIP_LAUNCH(k_ncc_feature_rows, dim3(C), dim3(kBlock), 0, st, bv.N, dev_feat);
}

#ifdef MTFHIP_GRID_TRACE
void debug_grid_trace(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_grid_trace), sizeof(unsigned long long) * 32); }
#endif
} // namespace mtfhip
#ifdef MTFHIP_GRID_TRACE
extern "C" void mtfhip_debug_grid_trace(unsigned long long *out) { mtfhip::debug_grid_trace(out); }
#endif
