/*
 * kernels_second_order.hip -- second-order Hessians (sec_ord_hess): image Hessians, SSM pixel Hessians, the fused second-order SSD term
 * (one of the translation units of libmtfhip.so; conventions and the shared device helpers: mtfhip_device.h)
 */
#include "mtfhip_device.h"
#include "mtfhip_mi_device.h"

namespace mtfhip {

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
 *   term  4: (MI only) hist_grad_term * Dt   cmptSelfHessian (2nd order), MI.cc:697-735 (CurrentSelf, SumOfSelf)
 * d0_variant: how init_pix_hessian was produced (Warped at the identity warp by initialize, Init after setRegion).
 * With nc.rows set the weights are NCC's: term 0 df_dIt Dt (NCC.cc:401-410), 1 df_dIt Dt + df_dI0 D0, 2 df_dIt (D0 + Dt) / 2,
 * 3 df_dI0 D0 (NCC.cc:391-400).  With mi.tb set they are MI's per-pixel gradients, same assignment (MI.cc:659-695). */
template <int SSM>
__global__ __launch_bounds__(kBlock) void k_second_order_ssd(BatchView bv, ImgView im, int term, int chained, int d0_variant,
	double grad_eps, double hess_eps, double norm_mult, double norm_add, double *partials, int nblk, int own_pts, SecondOrderNcc nc,
	SecondOrderMi mi) {
	constexpr int S = SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	__shared__ double lds[4 * S * S];
	const int t = blockIdx.y, N = bv.N;
	/* MI (MI.cc:659-695): the weights are its per-pixel gradients df_dIt = sum_r gradIt(r) sum_c matI0(c) T_curr(r, c) and df_dI0 =
	 * sum_r gradI0(r) sum_c matIt(c) T_init(r, c) (MI.cc:406-415, 432-441) -- the gradient-factor tables of this iteration, staged
	 * with the zero border the un-clamped windows index into ((bin + 1) in both directions, as pass 2 of the recompute form) */
	constexpr int kTR = 12;
	__shared__ double mi_tc[kTR * MI_NB], mi_ti[kTR * MI_NB];   /* (term 4: mi_tc holds the self table) */
	const bool is_mi = mi.tb != nullptr;
	if (is_mi) {
		const double *tb = mi.tb + (size_t)t * MI_SIZE;
		for (int k = threadIdx.x; k < kTR * MI_NB; k += kBlock) {
			const int r = k / MI_NB - 1, c = k % MI_NB - 1;
			const bool in = r >= 0 && r < 8 && c >= 0 && c < 8;
			mi_tc[k] = in ? tb[(term == 4 ? MI_T_SELF : MI_T_CURR) + r * MI_NB + c] : 0.0; mi_ti[k] = in ? tb[MI_T_INIT + r * MI_NB + c] : 0.0;
		}
		__syncthreads();
	}
	/* NCC (NCC.cc:391-410): the weights are its gradients df_dIt = (I0c / c - f Itc / b) / b and df_dI0 = (Itc / b - f I0c / c) / c
	 * (NCC.cc:163-234), functions of this pass's It moments: every workgroup sums the three moment columns of the fused pass's
	 * partial rows (same order everywhere: identical scalars in every workgroup) instead of waiting for a launch that would */
	const bool ncc = nc.rows != nullptr;
	double n_m0 = 0, n_c = 1, n_mt = 0, n_b = 1, n_f = 0;
	if (ncc) {
		__shared__ double mom_s[3];
		if (threadIdx.x < 64) {
			double s0 = 0, s1 = 0, s2 = 0;
			for (int r = threadIdx.x; r < nc.nblk; r += 64) {
				const double *row = nc.rows + ((size_t)t * nc.nblk + r) * NCC_ACC_COUNT;
				s0 += row[NCC_IT]; s1 += row[NCC_IT2]; s2 += row[NCC_I0IT];
			}
#pragma unroll
			for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
			if (threadIdx.x == 0) { mom_s[0] = s0; mom_s[1] = s1; mom_s[2] = s2; }
		}
		__syncthreads();
		const double nN = (double)N;
		n_m0 = nc.sc[(size_t)t * 8 + 0]; n_c = nc.sc[(size_t)t * 8 + 1];
		n_mt = mom_s[0] / nN;
		n_b = sqrt(mom_s[1] - nN * n_mt * n_mt);
		n_f = (mom_s[2] - nN * n_m0 * n_mt) / (n_b * n_c);
	}
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
	const double2 *h0 = (term != 0 && term != 4) ? reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_D2I0_DX2] + (size_t)t * N * 4) : nullptr;
	const double heps2 = 2 * hess_eps;
	const double hmult = norm_mult / (heps2 * heps2), gmult = norm_mult / (2 * grad_eps);
	double acc[S * S];
#pragma unroll
	for (int k = 0; k < S * S; ++k) acc[k] = 0.0;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += nblk * kBlock) {
		const double2 p0 = ip[i];
		double2 c, chv;
		double D;
		if (own_pts) {
			/* the device-side loop: curr_pts_hm = curr_warp * init_pts_hm and its dehomogenisation in registers, with k_apply_warp's
			 * expressions (Homography.cc:86-90, Affine.cc:104) -- CURR_PTS / CURR_HXY / CURR_Z are not refreshed between its passes */
			const double2 hp = bv.unit_z ? p0 : (reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_HXY]) + (size_t)t * N)[i];
			const double z = bv.unit_z ? 1.0 : (bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * N)[i];
			if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
				const double cx = W.m[0] * hp.x + W.m[1] * hp.y + W.m[2] * z, cy = W.m[3] * hp.x + W.m[4] * hp.y + W.m[5] * z;
				D = W.m[6] * hp.x + W.m[7] * hp.y + W.m[8] * z;
				c = make_double2(cx / D, cy / D); chv = make_double2(cx, cy);
			} else {
				c = make_double2(W.m[0] * hp.x + W.m[1] * hp.y + W.m[2] * z, W.m[3] * hp.x + W.m[4] * hp.y + W.m[5] * z);
				D = 1.0; chv = c;
			}
		} else {
			c = cp[i]; D = cz[i]; chv = ch[i];
		}
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
				if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) { const double2 h = chv; q0 = h.x; q1 = h.y; q2 = D; }
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
		double wt = term == 0 ? -r : (term == 1 ? r : (term == 2 ? -r / 2.0 : 0.0));
		double w0 = term == 1 ? r : (term == 2 ? -r / 2.0 : (term == 3 ? r : 0.0));
		if (ncc) {   /* the generic sum weights each block by its own gradient (AppearanceModel.h:209-219), not SSD's df_dI0 for both */
			const double i0c_c = (I0[i] - n_m0) / n_c, itc_b = ((norm_mult * cv + norm_add) - n_mt) / n_b;
			const double dft = (i0c_c - n_f * itc_b) / n_b, df0 = (itc_b - n_f * i0c_c) / n_c;
			wt = term == 0 ? dft : (term == 1 ? dft : (term == 2 ? dft / 2.0 : 0.0));
			w0 = term == 1 ? df0 : (term == 2 ? dft / 2.0 : (term == 3 ? df0 : 0.0));
		}
		if (is_mi) {
			const BsplWin4 a = bspl_window4<false>(norm_mult * cv + norm_add, 8, mi.hist_norm), c0 = bspl_window4<false>(I0[i], 8, mi.hist_norm);
			double dft = 0, df0 = 0;
			{
				const double *T0 = mi_tc + a.row0 * MI_NB + c0.row0;
#pragma unroll
				for (int r2 = 0; r2 < 4; ++r2) {
					const double *Tr = T0 + r2 * MI_NB;
					dft = fma(a.d[r2], fma(c0.w[3], Tr[3], fma(c0.w[2], Tr[2], fma(c0.w[1], Tr[1], c0.w[0] * Tr[0]))), dft);
				}
			}
			{
				const double *T0 = mi_ti + c0.row0 * MI_NB + a.row0;
#pragma unroll
				for (int r2 = 0; r2 < 4; ++r2) {
					const double *Tr = T0 + r2 * MI_NB;
					df0 = fma(c0.d[r2], fma(a.w[3], Tr[3], fma(a.w[2], Tr[2], fma(a.w[1], Tr[1], a.w[0] * Tr[0]))), df0);
				}
			}
			wt = term == 0 ? dft : (term == 1 ? dft : (term == 2 ? dft / 2.0 : 0.0));
			w0 = term == 1 ? df0 : (term == 2 ? dft / 2.0 : (term == 3 ? df0 : 0.0));
			if (term == 4) {   /* MI.cc:710-723: hist_grad_term = sum_r gradIt(r) sum_t matIt(t) self_grad_factor(r, t) */
				const double *T0 = mi_tc + a.row0 * MI_NB + a.row0;
				double hg = 0;
#pragma unroll
				for (int r2 = 0; r2 < 4; ++r2) {
					const double *Tr = T0 + r2 * MI_NB;
					hg = fma(a.d[r2], fma(a.w[3], Tr[3], fma(a.w[2], Tr[2], fma(a.w[1], Tr[1], a.w[0] * Tr[0]))), hg);
				}
				wt = hg; w0 = 0.0;
			}
		}
		if (term != 3) {
#pragma unroll
			for (int k = 0; k < S * S; ++k) acc[k] = fma(wt, d2[k], acc[k]);
		}
		if (term != 0 && term != 4) {
			const double2 ma = h0[2 * i], mb = h0[2 * i + 1];
			pix_hessian_block<SSM>(d2, d0_variant, Wid, st0, p0.x, p0.y, p0.x, p0.y, 1.0, ma.x, ma.y, mb.x, mb.y, g0[i], g0[N + i]);
#pragma unroll
			for (int k = 0; k < S * S; ++k) acc[k] = fma(w0, d2[k], acc[k]);
		}
	}
	block_reduce_store<S * S>(acc, partials + ((size_t)t * nblk + blockIdx.x) * (S * S), lds);
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

/* ===================================================================== */
/* launchers                                                              */
/* ===================================================================== */
void launch_hess_pts(const BatchView &bv, double eps, hipStream_t st) {
	MTFHIP_LAUNCH(k_hess_pts, dim3(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv, eps);
}
void launch_img_hess(const BatchView &bv, const ImgView &im, const double *pts, double *hess, double eps, double mult, hipStream_t st) {
	if (bv.C > 1) {
		MTFHIP_LAUNCH(k_img_hess_mc, dim3(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.NP, bv.C, im, pts, (const double *)nullptr, hess, eps, mult);
		return;
	}
	MTFHIP_LAUNCH(k_img_hess, dim3(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, pts, hess, eps, mult);
}
void launch_warped_img_hess(const BatchView &bv, const ImgView &im, const double *pts, const double *hp, double *hess, double eps,
	double mult, hipStream_t st) {
	if (bv.C > 1) {
		MTFHIP_LAUNCH(k_img_hess_mc, dim3(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.NP, bv.C, im, pts, hp, hess, eps, mult);
		return;
	}
	MTFHIP_LAUNCH(k_warped_img_hess, dim3(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, im, pts, hp, hess, eps, mult);
}
void launch_pix_hessian(const BatchView &bv, int variant, const double *hess, const double *grad, double *D, hipStream_t st) {
	const dim3 grid(simple_blocks_per_target(bv.N), bv.B);
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY) MTFHIP_LAUNCH(k_pix_hessian<MTFHIP_SSM_HOMOGRAPHY>, grid, dim3(kBlock), 0, st, bv, variant, hess, grad, D);
	else MTFHIP_LAUNCH(k_pix_hessian<MTFHIP_SSM_AFFINE>, grid, dim3(kBlock), 0, st, bv, variant, hess, grad, D);
}
void launch_weighted_plane_sum(const BatchView &bv, const double *d2a, const double *d2b, const double *w, double *partials, int nblk,
	double *out, hipStream_t st) {
	const dim3 grid(nblk, bv.B);
	if (bv.S == 8) MTFHIP_LAUNCH(k_weighted_plane_sum<64>, grid, dim3(kBlock), 0, st, bv.N, d2a, d2b, w, partials, nblk);
	else MTFHIP_LAUNCH(k_weighted_plane_sum<36>, grid, dim3(kBlock), 0, st, bv.N, d2a, d2b, w, partials, nblk);
	MTFHIP_LAUNCH(k_plane_sum_finish, dim3(bv.B), dim3(64), 0, st, partials, nblk, bv.S * bv.S, out);
}
void launch_second_order_ssd(const BatchView &bv, const ImgView &im, int term, int chained, int d0_variant, double grad_eps,
	double hess_eps, double norm_mult, double norm_add, double *partials, int nblk, double *out, hipStream_t st, int own_pts, SecondOrderNcc nc,
	SecondOrderMi mi) {
	const dim3 grid(nblk, bv.B);
	if (bv.ssm == MTFHIP_SSM_HOMOGRAPHY)
		MTFHIP_LAUNCH(k_second_order_ssd<MTFHIP_SSM_HOMOGRAPHY>, grid, dim3(kBlock), 0, st, bv, im, term, chained, d0_variant, grad_eps,
			hess_eps, norm_mult, norm_add, partials, nblk, own_pts, nc, mi);
	else
		MTFHIP_LAUNCH(k_second_order_ssd<MTFHIP_SSM_AFFINE>, grid, dim3(kBlock), 0, st, bv, im, term, chained, d0_variant, grad_eps,
			hess_eps, norm_mult, norm_add, partials, nblk, own_pts, nc, mi);
	MTFHIP_LAUNCH(k_plane_sum_finish, dim3(bv.B), dim3(64), 0, st, partials, nblk, bv.S * bv.S, out);
}

} // namespace mtfhip
