/*
 * kernels_init.hip -- a small patch's whole nt::ICLK::initialize in ONE launch (r05): GridTracker::resetTrackers(reinit = true)
 * re-initialises every patch tracker on the new grid after every frame (SM/src/GridTracker.cc:345-392 with the shipped
 * reset_at_each_frame = 1, Config/modules.cfg:80), i.e. per patch the body of NT/ICLK.cc:71-128:
 *     am->initializePixVals(pts)  am->initializePixGrad(pts)  ssm->cmptWarpedPixJacobian(J0, dI0_dx)
 *     am->initializeSimilarity()  am->initializeGrad()  am->initializeHess()  am->cmptSelfHessian(H0, J0)
 * Call by call that is ~15 launches and six host round trips (NCC's mean and norm, the Hessian read back for its inversion on the
 * host, the template moments of the fused NCC path): 385 us per frame of 256 patches against 42 us for tracking them.  Here one
 * workgroup per patch samples the template, takes its finite-difference gradient, writes the steepest-descent rows, reduces the
 * moments everything else is a function of -- sum I0, |I0 - mean|^2, sum J0, sum I0 J0, Gram(J0) -- forms the constant self
 * Hessian (SSD: -J0^T J0, SSDBase.cc:268-285; NCC: NCC.cc:337-389 in raw moments, the algebra of api_fused.hip::ncc_assemble),
 * inverts it (scaled, partially pivoted Gauss-Jordan on one wave) and leaves the small results both in device memory (for the
 * tracking launches that follow) and in a pinned host record the library folds into its mirrors when somebody asks for them.
 * One of the translation units of libmtfhip.so.
 *
 * Per-pixel arithmetic is the interface kernels' (k_sample, k_img_grad, k_pix_jacobian at the identity warp): I0, dI0_dx and J0 are
 * the same bits.  The reduced quantities are summed in another order (one workgroup instead of a grid + a fixed-order finish): they
 * agree to rounding (tests/test_gpu_grid.py::test_fused_template_init_equals_call_by_call).
 */
#include "mtfhip_device.h"
#include "mtfhip_grid_device.h"

namespace mtfhip {

template <int AM, int PPT>
__global__ __launch_bounds__(kBlock) void k_template_init(BatchView bv, ImgView im, double grad_eps, double norm_mult, double norm_add,
	double *h0_all /* [B][64] */, double *h0inv_all /* [B][64] */, double *ncc_all /* [B][8] */, double *ncc_tm_all /* [B][52] */, InitPublish pub,
	RegionIngest rg) {
	constexpr bool NCC = AM == MTFHIP_AM_NCC;
	__shared__ double red[4 * 16];
	__shared__ double sSum[56], sRed56[4 * 56], sH[64], sA[8 * 17], sRec[kInitRec];
	const double *sG = sSum, *sSJ = sSum + 36, *sIJ = sSum + 44;
	const int t = blockIdx.x, N = bv.N, S = bv.S, tid = threadIdx.x;
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY;
	const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * N;
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * N;
	double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N, *It = bv.buf[MTFHIP_BUF_IT] + (size_t)t * N;
	double *dI0 = bv.buf[MTFHIP_BUF_DI0_DX] + (size_t)t * 2 * N, *dIt = bv.buf[MTFHIP_BUF_DIT_DX] + (size_t)t * 2 * N;
	double *J0 = bv.buf[MTFHIP_BUF_J0] + (size_t)t * N * S;
	double *df0 = bv.buf[MTFHIP_BUF_DF_DI0] + (size_t)t * N, *dft = bv.buf[MTFHIP_BUF_DF_DIT] + (size_t)t * N;
	const double gmult = norm_mult / (2 * grad_eps);
	/* REGION mode (rg.corners != NULL; mtfhip_grid_reset): the patch's corners come straight from the pinned staging buffer and the
	 * workgroup lays out its own grid first -- ProjectiveBase::getPtsFromCorners + Homography / Affine::setCorners with the expressions of
	 * k_init_grid and of k_iclk_track's region mode (one set, rect_to_quad_hd) -- instead of a set_corners launch in front of this one */
	const bool region = rg.corners != nullptr;
	__shared__ double sCr[8];
	double W0[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	if (region) {
		double q8[8];
		if (rg.layout) {
			/* (r05) the patch laid out HERE from the grid's region -- kernel arguments only: no PCIe read in front of everything, and the host
			 * lays the patches out for its mirrors behind the launch (RegionIngest::layout, mtfhip_grid_reset) */
			grid_patch_corners_lanes(rg.grid, rg.region_map, t, q8);
		} else {
			if (tid < 8) sCr[tid] = rg.corners[8 * (size_t)t + tid];
			__syncthreads();
#pragma unroll
			for (int q = 0; q < 8; ++q) q8[q] = sCr[q];
		}
		const bool bad = !rect_to_quad_hd(rg.lo_x, rg.lo_y, rg.hi_x, rg.hi_y, q8, W0);
		if (bad) {
#pragma unroll
			for (int q = 0; q < 9; ++q) W0[q] = (q == 0 || q == 4 || q == 8) ? 1.0 : 0.0;
		}
		if (hom && fabs(W0[6]) < 1e-15 && fabs(W0[7]) < 1e-15) { W0[6] = 0; W0[7] = 0; }
		/* (degenerate corners are refused on the host before the launch -- set_corners_core, quad_degenerate_hd: the identity is only a safe stand-in) */
		if (tid < 9) rg.d_w0[9 * (size_t)t + tid] = W0[tid];
		if (tid < 12) rg.d_init_corners_hm[12 * (size_t)t + tid] = (tid % 3 == 2) ? 1.0 : q8[2 * (tid / 3) + tid % 3];
	}
	double2 *ipw = const_cast<double2 *>(ip), *ihw = reinterpret_cast<double2 *>(bv.buf[MTFHIP_BUF_INIT_HXY]) + (size_t)t * N;
	double *izw = const_cast<double *>(iz);
	double i0v[PPT], jv[PPT][8];
	/* phase 1: every pixel's grid point and the four texels of its bilinear cell, requested back to back (unconditional loads at a
	 * clamped address: a sample outside the frame keeps valid = false and takes the general sampler in phase 2) -- sample by sample
	 * the five pix_val calls of a pixel were a chain of memory round trips, 15 per thread (27.7 us for 256 patches of 25 x 25) */
	double2 pk[PPT]; double zk[PPT];
	Cell ck[PPT];
#pragma unroll
	for (int k = 0; k < PPT; ++k) {
		const int i = tid + k * kBlock;
		const int ic = i < N ? i : N - 1;
		double2 p; double zi = 1.0;
		if (region) {
			const int col = ic % rg.resx, row = ic / rg.resx;
			const double nx = (rg.resx == 1 || col == rg.resx - 1) ? rg.hi_x : rg.lo_x + col * ((rg.hi_x - rg.lo_x) / (rg.resx - 1));
			const double ny = (rg.resy == 1 || row == rg.resy - 1) ? rg.hi_y : rg.lo_y + row * ((rg.hi_y - rg.lo_y) / (rg.resy - 1));
			const double X = W0[0] * nx + W0[1] * ny + W0[2] * 1.0;
			const double Y = W0[3] * nx + W0[4] * ny + W0[5] * 1.0;
			const double Z = W0[6] * nx + W0[7] * ny + W0[8] * 1.0;
			p = (W0[6] == 0 && W0[7] == 0 && W0[8] == 1.0) ? make_double2(X, Y) : make_double2(X / Z, Y / Z);
			zi = rg.force_unit_z ? 1.0 : Z;
			if (i < N) { ipw[i] = p; izw[i] = zi; ihw[i] = rg.force_unit_z ? p : make_double2(X, Y); }
		} else { p = ip[ic]; zi = bv.unit_z ? 1.0 : iz[ic]; }
		pk[k] = p; zk[k] = zi;
		/* load_cell (mtfhip_device.h) with the loads hoisted out of its branches */
		Cell c;
		const double w = (double)(unsigned int)im.w, h = (double)(unsigned int)im.h;
		const bool in0 = !((p.x < 0) || (p.x >= w) || (p.y < 0) || (p.y >= h));
		const int lx = in0 ? (int)p.x : 0, ly = in0 ? (int)p.y : 0;
		const double dx = p.x - lx, dy = p.y - ly;
		const int ux = dx == 0 ? lx : lx + 1, uy = dy == 0 ? ly : ly + 1;
		c.valid = in0 && !(ux >= im.w || uy >= im.h);
		c.lx = c.valid ? lx : -1; c.ly = c.valid ? ly : -1; c.ux = c.valid ? ux : -1; c.uy = c.valid ? uy : -1;
		const int slx = c.valid ? lx : 0, sly = c.valid ? ly : 0, sux = c.valid ? ux : 0, suy = c.valid ? uy : 0;
		const float *r0 = im.data + (size_t)sly * im.stride, *r1 = im.data + (size_t)suy * im.stride;
		c.t00 = r0[slx]; c.t01 = r0[sux]; c.t10 = r1[slx]; c.t11 = r1[sux];
		ck[k] = c;
	}
#pragma unroll
	for (int k = 0; k < PPT; ++k) {
		const int i = tid + k * kBlock;
		i0v[k] = 0.0;
#pragma unroll
		for (int s = 0; s < 8; ++s) jv[k][s] = 0.0;
		if (i < N) {
			const double2 p = pk[k]; const double zi = zk[k];
			const Cell &c = ck[k];
			/* ImageBase::initializePixVals ImageBase.cc:62-99 (k_sample: pix_val; from the fetched cell it is the same expression) */
			const double v = norm_mult * pix_val_cell(im, c, p.x, p.y) + norm_add;
			/* ImageBase::initializePixGrad(pts) ImageBase.cc:101-132 -> utils::getImgGrad imgUtils.cc:233-254 (k_img_grad) */
			double inc = pix_val_cell(im, c, p.x + grad_eps, p.y), dec = pix_val_cell(im, c, p.x - grad_eps, p.y);
			const double gx = (inc - dec) * gmult;
			inc = pix_val_cell(im, c, p.x, p.y + grad_eps); dec = pix_val_cell(im, c, p.x, p.y - grad_eps);
			const double gy = (inc - dec) * gmult;
			/* cmptWarpedPixJacobian at the identity warp and zero state (Homography.cc:231-294, Affine.cc:213-242; k_pix_jacobian's
			 * expressions with W = I, so the same bits) */
			double r[8];
#pragma unroll
			for (int s = 0; s < 8; ++s) r[s] = 0.0;
			if (hom) {
				const double inv_det = 1.0 / (bv.unit_z ? 1.0 : zi);
				const double dwx_dx = (1.0 - 0.0 * p.x), dwx_dy = (0.0 - 0.0 * p.x), dwy_dx = (0.0 - 0.0 * p.y), dwy_dy = (1.0 - 0.0 * p.y);
				const double Ix = (dwx_dx * gx + dwy_dx * gy) * inv_det, Iy = (dwx_dy * gx + dwy_dy * gy) * inv_det;
				hom_row(r, Ix, Iy, p.x, p.y, p.x, p.y);
			} else {
				const double a = 0.0 + 1, b = 0.0, cc = 0.0, d = 0.0 + 1;
				const double Ixx = gx * p.x, Ixy = gx * p.y, Iyy = gy * p.y, Iyx = gy * p.x;
				r[0] = gx * a + gy * cc; r[1] = gx * b + gy * d;
				r[2] = Ixx * a + Iyx * cc; r[3] = Ixy * a + Iyy * cc; r[4] = Ixx * b + Iyx * d; r[5] = Ixy * b + Iyy * d;
			}
			I0[i] = v; It[i] = v;                                 /* (It = I0 and dIt_dx = dI0_dx on initialisation, ImageBase.cc:95, 128) */
			dI0[i] = gx; dI0[N + i] = gy; dIt[i] = gx; dIt[N + i] = gy;
			df0[i] = 0.0; dft[i] = 0.0;                          /* initializeSimilarity / initializeGrad: the gradient vectors start at zero (SSDBase.cc:29-63, NCC.cc:97-122) */
#pragma unroll
			for (int s = 0; s < 8; ++s) if (s < S) J0[(size_t)s * N + i] = r[s];
			i0v[k] = v;
#pragma unroll
			for (int s = 0; s < 8; ++s) jv[k][s] = r[s];
		}
	}
	/* the moments: sum I0 first (the norm is of the CENTRED template, NCC.cc:62-75: two passes as the reference) */
	double m0 = 0.0, cn = 1.0;
	if constexpr (NCC) {
		double s1[1] = {0.0};
#pragma unroll
		for (int k = 0; k < PPT; ++k) s1[0] += i0v[k];
		init_allsum<1>(s1, red);
		m0 = s1[0] / (double)N;
		double s2[1] = {0.0};
#pragma unroll
		for (int k = 0; k < PPT; ++k) { const int i = tid + k * kBlock; const double dv = i < N ? i0v[k] - m0 : 0.0; s2[0] = fma(dv, dv, s2[0]); }
		init_allsum<1>(s2, red);
		cn = sqrt(s2[0]);
	}
	/* Gram(J0) (upper triangle in ACC_H's order, 36) | sum J0 (8) | sum I0 J0 (8) | pad: one halving reduction of 56 values
	 * (block_reduce_store, mtfhip_device.h) instead of a wave sum per value */
	{
		double acc[56];
#pragma unroll
		for (int q = 0; q < 56; ++q) acc[q] = 0.0;
#pragma unroll
		for (int k = 0; k < PPT; ++k) {
			int idx = 0;
#pragma unroll
			for (int a = 0; a < 8; ++a)
#pragma unroll
				for (int c = a; c < 8; ++c) { acc[idx] = fma(jv[k][a], jv[k][c], acc[idx]); ++idx; }
#pragma unroll
			for (int s = 0; s < 8; ++s) { acc[36 + s] += jv[k][s]; acc[44 + s] = fma(i0v[k], jv[k][s], acc[44 + s]); }
		}
		__syncthreads();   /* (red: the readers of the reductions above are done) */
		block_reduce_store<56>(acc, sSum, sRed56);
	}
	__syncthreads();
	/* the constant self Hessian, column-major S x S packed (what init_template keeps in th[].h0 / d_h0) */
	if (tid < 64) {
		const int a = tid >> 3, c = tid & 7;
		if (a < S && c < S) {
			const int lo = a < c ? a : c, hi = a < c ? c : a;
			const double G = sG[lo * 8 - (lo * (lo - 1)) / 2 + (hi - lo)];
			double h;
			if constexpr (NCC) {
				/* cmptSelfHessian = -(Gram - sJ sJ^T / N) / b^2 + ut ut^T, ut = (sum It J - mt sJ) / b^2, at It = I0: b = c, mt = m0 (NCC.cc:337-389) */
				const double inv_b2 = 1.0 / (cn * cn);
				const double ua = (sIJ[a] - m0 * sSJ[a]) * inv_b2, uc = (sIJ[c] - m0 * sSJ[c]) * inv_b2;
				h = -(G - sSJ[a] * sSJ[c] / (double)N) * inv_b2 + ua * uc;
			} else h = -G;   /* SSDBase.cc:268-285 */
			sH[c * S + a] = h;
		}
	}
	__syncthreads();
	if (tid < 64) invert_definite_wave(S, sH, sA, h0inv_all + (size_t)t * 64);
	/* device copies + the host record: H0 64 | NCC scalars 8 (I0_mean, c, It_mean, b, f, gmean, 0, 0) | sum J0 8 | sum I0 J0 8 | Gram 36 */
	if (tid < kInitRec) {
		double v = 0.0;
		if (tid < 64) v = tid < S * S ? sH[tid] : 0.0;
		else if (tid < 72) { const int q = tid - 64; v = NCC ? (q == 0 ? m0 : (q == 1 ? cn : (q == 2 ? m0 : (q == 3 ? cn : (q == 4 ? 1.0 : 0.0))))) : 0.0; }
		else if (tid < 80) v = sSJ[tid - 72];
		else if (tid < 88) v = sIJ[tid - 80];
		else if (tid < 124) v = sG[tid - 88];
		sRec[tid] = v;
		if (tid < 64) h0_all[(size_t)t * 64 + tid] = v;
		if (NCC && tid >= 64 && tid < 72 && ncc_all) ncc_all[(size_t)t * 8 + tid - 64] = v;
		if (NCC && tid >= 72 && tid < 124 && ncc_tm_all) ncc_tm_all[(size_t)t * 52 + tid - 72] = v;
		if (pub.host) __hip_atomic_store(pub.host + (size_t)t * kInitRec + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	if (pub.host) {
		/* the hand-over of every host publisher (publish_fenced(), mtfhip_internal.h) */
		if (pub.fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else wait_stores_acked();
		__syncthreads();
		if (tid == 0) {
			const int done = pub.fenced ? __hip_atomic_fetch_add(pub.count, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
			                            : __hip_atomic_fetch_add(pub.count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (done == (int)gridDim.x - 1) {
				__hip_atomic_store(pub.count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (pub.fenced) { __threadfence_system(); __hip_atomic_store(pub.flag, pub.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
				else __hip_atomic_store(pub.flag, pub.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			}
		}
	}
}

template <int AM>
static void launch_init_am(const BatchView &bv, const ImgView &im, double grad_eps, double norm_mult, double norm_add, double *h0, double *h0inv,
	double *ncc, double *ncc_tm, const InitPublish &pub, const RegionIngest &rg, hipStream_t st) {
	const int ppt = (bv.N + kBlock - 1) / kBlock;
#define MTFHIP_INIT_CASE(P) MTFHIP_LAUNCH((k_template_init<AM, P>), dim3(bv.B), dim3(kBlock), 0, st, bv, im, grad_eps, norm_mult, norm_add, h0, h0inv, ncc, ncc_tm, pub, rg)
	if (ppt <= 1) MTFHIP_INIT_CASE(1); else if (ppt == 2) MTFHIP_INIT_CASE(2); else if (ppt == 3) MTFHIP_INIT_CASE(3); else MTFHIP_INIT_CASE(4);
#undef MTFHIP_INIT_CASE
}
/* N <= kTemplateInitMaxPix, single channel, SSD or NCC */
void launch_template_init(const BatchView &bv, const ImgView &im, double grad_eps, double norm_mult, double norm_add, double *h0, double *h0inv,
	double *ncc, double *ncc_tm, const InitPublish &pub, const RegionIngest &rg, hipStream_t st) {
	if (bv.am == MTFHIP_AM_NCC) launch_init_am<MTFHIP_AM_NCC>(bv, im, grad_eps, norm_mult, norm_add, h0, h0inv, ncc, ncc_tm, pub, rg, st);
	else launch_init_am<MTFHIP_AM_SSD>(bv, im, grad_eps, norm_mult, norm_add, h0, h0inv, ncc, ncc_tm, pub, rg, st);
}

} // namespace mtfhip
