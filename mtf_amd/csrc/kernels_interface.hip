/*
 * kernels_interface.hip -- the un-fused kernels behind the per-function AppearanceModel / StateSpaceModel entry points (SSD, NCC)
 * (one of the translation units of libmtfhip.so; conventions and the shared device helpers: mtfhip_device.h)
 */
#include "mtfhip_device.h"

namespace mtfhip {

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
/* (ing: the staged state slab, read from pinned host memory by the first workgroups of the same launch -- w0_all then points into
 * the host copy too, so nothing in this kernel depends on the ingest having landed) */
struct SlabIngest { const uint4 *src; uint4 *dst; unsigned n16; const unsigned *src_tail; unsigned *dst_tail; unsigned n_tail; };
/* write_curr = 0: only the template grid is laid out; CURR_PTS / CURR_HXY / CURR_Z (40 of the 64 bytes per point) are left to
 * k_apply_warp, which the caller's pts_stale flag triggers if an un-fused kernel ever asks for them (mtfhip_batch_track_region:
 * the loop that follows warps the template grid itself) */
__global__ __launch_bounds__(kBlock) void k_init_grid(BatchView bv, const double *w0_all, int resx, int resy,
	double lo_x, double lo_y, double hi_x, double hi_y, int force_unit_z, SlabIngest ing, int write_curr) {
	const int t = blockIdx.y;
	if (ing.src) {
		const unsigned i = (blockIdx.y * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
		if (i < ing.n16) ing.dst[i] = ing.src[i];
		if (i < ing.n_tail) ing.dst_tail[i] = ing.src_tail[i];
	}
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
		ip[i] = p; iz[i] = z; ih[i] = hxy;
		if (write_curr) { cp[i] = p; cz[i] = z; ch[i] = hxy; }
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

/* utils::getWeightedPixVals Utilities/src/imgUtils.cc:506-523 (SSD::updateModel AM/src/SSD.cc:49-75, NCC::updateModel
 * AM/src/NCC.cc:539-566): the template moves towards the patch at the given points -- running average over frame_count
 * frames, or weight alpha for the new patch (the reference applies the pixel normalisation only in the running average) */
__global__ __launch_bounds__(kBlock) void k_update_model(int N, ImgView im, const double *pts_all, double *I0_all,
	double mult, double add, double frame_count, double alpha, int running_avg) {
	const int t = blockIdx.y;
	const double2 *pts = reinterpret_cast<const double2 *>(pts_all) + (size_t)t * N;
	double *I0 = I0_all + (size_t)t * N;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		const double2 p = pts[i];
		const double old = I0[i];
		if (running_avg) {
			const double v = mult * pix_val(im, p.x, p.y) + add;
			I0[i] = old + (v - old) / frame_count;
		} else {
			const double v = pix_val(im, p.x, p.y);
			I0[i] = alpha * v + (1 - alpha) * old;
		}
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


/* the same for rows of any length (the NCC moment rows) */
__global__ __launch_bounds__(128) void k_finish_rows(const double *partials, int nblk, int row_len, double *out) {
	const int t = blockIdx.x, k = blockIdx.y * 128 + threadIdx.x;   /* grid.y covers rows longer than one workgroup */
	if (k >= row_len) return;
	out[(size_t)t * row_len + k] = column_sum(partials + (size_t)t * nblk * row_len + k, nblk, row_len);
}
/* the hand-over every publishing workgroup ends in, once its own results have been issued as system-scope stores: stores performed
 * (acknowledged write-through stores, or -- fenced -- an agent-scope release), the workgroup counted in, the flag raised by the last
 * arriver.  See publish_fenced() in mtfhip_internal.h for the two forms. */
__device__ __forceinline__ void publish_arrive(int *count, unsigned long long *flag_host, unsigned long long seq, int fenced) {
	if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else wait_stores_acked();
	__syncthreads();
	if (threadIdx.x == 0) {
		const int done = fenced ? __hip_atomic_fetch_add(count, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
		                        : __hip_atomic_fetch_add(count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (done == (int)gridDim.x - 1) {
			__hip_atomic_store(count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (fenced) { __threadfence_system(); __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
			else __hip_atomic_store(flag_host, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}
/* the same, delivered straight into host-coherent pinned memory: every target's row, then -- by the workgroup that
 * finishes last -- a sequence number the host is spinning on (system-scope release after the rows) */
__global__ __launch_bounds__(1024) void k_finish_host(const double *partials, int nblk, int row_len, double *out_host, int *count,
	unsigned long long *flag_host, unsigned long long seq, int fenced) {
	/* blockDim.x = 128 G: group j sums the rows j, j + G, ... of its column, group 0 adds the G partial sums in order -- a single
	 * target has hundreds of block rows, and one thread per column walking all of them was most of this kernel's 4 us */
	__shared__ double part[8][128];
	const int t = blockIdx.x, k = threadIdx.x & 127, j = threadIdx.x >> 7, G = blockDim.x >> 7;
	const double *p = partials + (size_t)t * nblk * row_len;
	double mine = 0.0;
	if (k < row_len && j < nblk) mine = column_sum(p + (size_t)j * row_len + k, (nblk - j + G - 1) / G, G * row_len);
	if (G > 1) {
		part[j][k] = mine;
		__syncthreads();
		if (j == 0) for (int g = 1; g < G; ++g) mine += part[g][k];
	}
	/* write-through (system-scope) stores, acknowledged before the workgroup counts itself in; the last arriver's flag is one more
	 * posted write of the same device behind them.  No fence: a system-scope release is a write-back of every L2 (2.5 us, measured on
	 * the grid kernel's publish in r04) for lines the host never reads -- the protocol of publish_target, kernels_batch.hip. */
	if (j == 0 && k < row_len) __hip_atomic_store(out_host + (size_t)t * row_len + k, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	publish_arrive(count, flag_host, seq, fenced);
}
/* a device buffer delivered straight into host-coherent pinned memory (32-bit words), then the sequence number the host is
 * spinning on: replaces a device-to-host copy + stream synchronisation at the end of the device-side loop */
__global__ __launch_bounds__(256) void k_publish_host(const unsigned *src, unsigned *dst_host, unsigned n_words, int *count,
	unsigned long long *flag_host, unsigned long long seq, int fenced) {
	for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n_words; i += gridDim.x * 256)
		__hip_atomic_store(dst_host + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	publish_arrive(count, flag_host, seq, fenced);   /* (as k_finish_host) */
}
/* the other direction: the staged state slab is read from pinned host memory by the kernel itself (16 bytes per lane, one PCIe
 * round trip) instead of through a copy-engine transfer and the cross-queue dependency that follows it */
/* [skip_lo, skip_hi): 16-byte units that are NOT copied (a section of the slab whose device copy is newer than the host's: the NCC scalars
 * behind a fused template initialisation whose record the host has not folded in yet) */
__global__ __launch_bounds__(256) void k_ingest_host(const uint4 *src_host, uint4 *dst, unsigned n16, const unsigned *src_tail, unsigned *dst_tail, unsigned n_tail,
	unsigned skip_lo, unsigned skip_hi) {
	const unsigned i = blockIdx.x * 256 + threadIdx.x;
	if (i < n16 && !(i >= skip_lo && i < skip_hi)) dst[i] = src_host[i];
	if (i < n_tail) dst_tail[i] = src_tail[i];
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


/* holds a queue for `ticks` of the constant-rate wall clock: the device-side loop starts its second queue half a period behind
 * the first (api_fused.hip, track_core) */
__global__ void k_queue_delay(unsigned long long ticks) {
	const unsigned long long t0 = wall_clock64();
	while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
void launch_queue_delay(double microseconds, hipStream_t st) {
	static double ticks_per_us = 0;
	if (ticks_per_us == 0) {
		int dev = 0, khz = 0;
		if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
		ticks_per_us = khz / 1000.0;
	}
	if (microseconds <= 0) return;
	MTFHIP_LAUNCH(k_queue_delay, dim3(1), dim3(64), 0, st, (unsigned long long)(microseconds * ticks_per_us));
}

/* ===================================================================== */
/* launchers                                                              */
/* ===================================================================== */
void launch_init_grid(const BatchView &bv, const double *dev_w0, int resx, int resy, double lo_x, double lo_y,
	double hi_x, double hi_y, int force_unit_z, hipStream_t st) {
	MTFHIP_LAUNCH(k_init_grid, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv, dev_w0, resx, resy,
		lo_x, lo_y, hi_x, hi_y, force_unit_z, SlabIngest{nullptr, nullptr, 0, nullptr, nullptr, 0}, 1);
}
/* the same with the slab ingest folded in; false: the launch is too small to carry it (the caller ingests separately) */
bool launch_init_grid_ingest(const BatchView &bv, const double *host_w0_dev, int resx, int resy, double lo_x, double lo_y,
	double hi_x, double hi_y, int force_unit_z, const void *src_host, void *dst, size_t bytes, int write_curr, hipStream_t st) {
	const unsigned n16 = (unsigned)(bytes / 16), n_tail = (unsigned)((bytes % 16) / 4);
	const dim3 g = grid2(simple_blocks_per_target(bv.N), bv.B);
	if ((size_t)g.x * g.y * kBlock < std::max(n16, n_tail)) return false;
	const SlabIngest ing{static_cast<const uint4 *>(src_host), static_cast<uint4 *>(dst), n16,
		reinterpret_cast<const unsigned *>(static_cast<const char *>(src_host) + 16 * (size_t)n16),
		reinterpret_cast<unsigned *>(static_cast<char *>(dst) + 16 * (size_t)n16), n_tail};
	MTFHIP_LAUNCH(k_init_grid, g, dim3(kBlock), 0, st, bv, host_w0_dev, resx, resy, lo_x, lo_y, hi_x, hi_y, force_unit_z, ing, write_curr);
	return true;
}
void launch_apply_warp(const BatchView &bv, hipStream_t st) {
	MTFHIP_LAUNCH(k_apply_warp, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv);
}
void launch_grad_pts(const BatchView &bv, double eps, hipStream_t st) {   /* per sample point */
	MTFHIP_LAUNCH(k_grad_pts, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv, eps);
}
void launch_sample(const BatchView &bv, const ImgView &im, const double *pts, double *out, double mult, double add,
	hipStream_t st) {
	if (bv.C > 1) { MTFHIP_LAUNCH(k_sample_mc, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.NP, bv.C, im, pts, out, mult, add); return; }
	MTFHIP_LAUNCH(k_sample, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, pts, out, mult, add);
}
void launch_update_model(const BatchView &bv, const ImgView &im, const double *pts, double *I0, double mult, double add,
	double frame_count, double alpha, int running_avg, hipStream_t st) {
	MTFHIP_LAUNCH(k_update_model, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, pts, I0, mult, add,
		frame_count, alpha, running_avg);
}
void launch_img_grad(const BatchView &bv, const ImgView &im, const double *pts, double *grad, double eps, double mult,
	hipStream_t st) {
	if (bv.C > 1) {
		MTFHIP_LAUNCH(k_img_grad_mc, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.NP, bv.C, im, pts, (const double *)nullptr, grad, eps, mult);
		return;
	}
	MTFHIP_LAUNCH(k_img_grad, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, pts, grad, eps, mult);
}
void launch_warped_img_grad(const BatchView &bv, const ImgView &im, const double *gp, double *grad, double eps,
	double mult, hipStream_t st) {
	if (bv.C > 1) {
		MTFHIP_LAUNCH(k_img_grad_mc, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.NP, bv.C, im, (const double *)nullptr, gp, grad, eps, mult);
		return;
	}
	MTFHIP_LAUNCH(k_warped_img_grad, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, gp, grad, eps, mult);
}
void launch_pix_jacobian(const BatchView &bv, int variant, const double *grad, double *J, hipStream_t st) {
	MTFHIP_LAUNCH(k_pix_jacobian, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv, variant, grad, J);
}
void launch_mean_jacobian(const BatchView &bv, hipStream_t st) {
	size_t n = (size_t)bv.B * bv.N * bv.S;
	int nb = (int)((n + kBlock * 4 - 1) / (kBlock * 4));
	MTFHIP_LAUNCH(k_mean_jacobian, dim3(nb), dim3(kBlock), 0, st, bv.buf[MTFHIP_BUF_J0], bv.buf[MTFHIP_BUF_JT],
		bv.buf[MTFHIP_BUF_JM], n);
}
void launch_negate(const double *src, double *dst, size_t n, hipStream_t st) {
	int nb = (int)((n + kBlock * 4 - 1) / (kBlock * 4));
	MTFHIP_LAUNCH(k_negate, dim3(nb), dim3(kBlock), 0, st, src, dst, n);
}
void launch_ssd_residual(const BatchView &bv, double *partials, int nblk, hipStream_t st) {
	MTFHIP_LAUNCH(k_ssd_residual, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv, partials, nblk);
}
void launch_gemv(const BatchView &bv, const double *v1, const double *J1, const double *v2, const double *J2,
	int sum_mode, double *partials, int nblk, hipStream_t st) {
	MTFHIP_LAUNCH(k_gemv, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv.N, bv.S, v1, J1, v2, J2, sum_mode, partials, nblk);
}
void launch_gram(const BatchView &bv, const double *J, double *partials, int nblk, hipStream_t st) {
	MTFHIP_LAUNCH(k_gram, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv.N, bv.S, J, partials, nblk);
}
void launch_vec_sum(const BatchView &bv, const double *v, double *partials, int nblk, hipStream_t st) {
	MTFHIP_LAUNCH(k_vec_sum, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv.N, v, partials, nblk);
}
void launch_ncc_centered(const BatchView &bv, const double *sc, double *partials, int nblk, hipStream_t st) {
	MTFHIP_LAUNCH(k_ncc_centered, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv, sc, partials, nblk);
}
void launch_ncc_grad(const BatchView &bv, const double *sc, int curr, double *out, double *partials, int nblk, hipStream_t st) {
	MTFHIP_LAUNCH(k_ncc_grad, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv, sc, curr, out, partials, nblk);
}
void launch_sub_mean(const BatchView &bv, double *v, const double *sc, hipStream_t st) {
	MTFHIP_LAUNCH(k_sub_mean, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, v, sc);
}
void launch_col_sum(const BatchView &bv, const double *J, double *partials, int nblk, hipStream_t st) {
	MTFHIP_LAUNCH(k_col_sum, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv.N, bv.S, J, partials, nblk);
}
void launch_ncc_hess(const BatchView &bv, const double *sc, const double *colmean, const double *J, double *partials,
	int nblk, hipStream_t st) {
	MTFHIP_LAUNCH(k_ncc_hess, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv, sc, colmean, J, partials, nblk);
}
void launch_finish_host(double *partials, int nblk, int row_len, double *out_host, int *count, unsigned long long *flag_host,
	unsigned long long seq, int B, hipStream_t st) {
	const int G = nblk >= 256 ? 8 : (nblk >= 64 ? 4 : 1);
	MTFHIP_LAUNCH(k_finish_host, dim3(B), dim3(128 * G), 0, st, partials, nblk, row_len, out_host, count, flag_host, seq, publish_fenced());
}
void launch_publish_host(const void *src, void *dst_host, size_t bytes, int *count, unsigned long long *flag_host,
	unsigned long long seq, hipStream_t st) {
	const unsigned n_words = (unsigned)(bytes / 4);
	const unsigned blocks = std::max(1u, std::min(64u, (n_words + 1023) / 1024));
	MTFHIP_LAUNCH(k_publish_host, dim3(blocks), dim3(256), 0, st, static_cast<const unsigned *>(src), static_cast<unsigned *>(dst_host),
		n_words, count, flag_host, seq, publish_fenced());
}
void launch_ingest_host(const void *src_host, void *dst, size_t bytes, hipStream_t st, size_t skip_off, size_t skip_len) {
	const unsigned n16 = (unsigned)(bytes / 16), n_tail = (unsigned)((bytes % 16) / 4);   /* (the slab is a multiple of 4 bytes) */
	const unsigned blocks = std::max(1u, (std::max(n16, n_tail) + 255) / 256);
	MTFHIP_LAUNCH(k_ingest_host, dim3(blocks), dim3(256), 0, st, static_cast<const uint4 *>(src_host), static_cast<uint4 *>(dst), n16,
		reinterpret_cast<const unsigned *>(static_cast<const char *>(src_host) + 16 * (size_t)n16),
		reinterpret_cast<unsigned *>(static_cast<char *>(dst) + 16 * (size_t)n16), n_tail, (unsigned)(skip_off / 16), (unsigned)((skip_off + skip_len) / 16));
}
void launch_finish_rows(double *partials, int nblk, int row_len, double *out, int B, hipStream_t st) {
	MTFHIP_LAUNCH(k_finish_rows, dim3(B, (row_len + 127) / 128), dim3(128), 0, st, partials, nblk, row_len, out);
}
void launch_finish(double *partials, int nblk, double *out, int B, hipStream_t st) {
	MTFHIP_LAUNCH(k_finish, dim3(B), dim3(64), 0, st, partials, nblk, out);
}

void launch_mean_planes(const double *a, const double *b, double *o, size_t n, hipStream_t st) {
	MTFHIP_LAUNCH(k_mean_jacobian, dim3((unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, 4096)), dim3(kBlock), 0, st, a, b, o, n);
}

} // namespace mtfhip
