/*
 * mtf_oracle.cpp -- CPU parity oracle (TEST INFRASTRUCTURE ONLY, see mtf_oracle.h).
 *
 * A plain C++17 / FP64 restatement of the Lucas-Kanade hot path of
 * abhineet123/MTF without Eigen / OpenCV / Boost.  Every function cites the
 * reference file:line it follows (paths relative to the reference root).
 * PARITY UNPINNED (no reference tests / goldens, reference not buildable here).
 *
 * Single threaded on purpose: the reference's default build is single threaded
 * (CMakeLists.txt:170 WITH_OPENMP OFF) and this file doubles as the timed CPU
 * baseline of bench.py.
 */
#include "mtf_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

typedef std::vector<double> vecd;

/* ===================================================================== */
/* L1: pixel utilities                                                    */
/* ===================================================================== */

/* Utilities/include/mtf/Utilities/imgUtils.h:51-53 */
inline bool overflow(double x, double y, unsigned int h, unsigned int w) {
	return (x < 0) || (x >= w) || (y < 0) || (y >= h);
}

/* getPixVal<Linear, Constant>: Utilities/include/mtf/Utilities/imgUtils.h:91-113
 * img is the row-major float image (EigImgT), overflow_val = 128 */
inline double pix_val(const float *img, int h, int w, double x, double y) {
	const double overflow_val = 128.0;
	if (overflow(x, y, h, w)) return overflow_val;
	int lx = static_cast<int>(x);
	int ly = static_cast<int>(y);
	double dx = x - lx;
	double dy = y - ly;
	int ux = dx == 0 ? lx : lx + 1;
	int uy = dy == 0 ? ly : ly + 1;
	if (overflow(lx, ly, h, w) || overflow(ux, uy, h, w)) return overflow_val;
	const float *r0 = img + static_cast<size_t>(ly) * w;
	const float *r1 = img + static_cast<size_t>(uy) * w;
	return r0[lx] * (1 - dx) * (1 - dy) +
		r0[ux] * dx * (1 - dy) +
		r1[lx] * (1 - dx) * dy +
		r1[ux] * dx * dy;
}

/* utils::getPixVals: Utilities/src/imgUtils.cc:163-173 */
void pix_vals(double *out, const float *img, int h, int w, const double *pts, int n,
	double norm_mult, double norm_add) {
	for (int i = 0; i < n; ++i)
		out[i] = norm_mult * pix_val(img, h, w, pts[2 * i], pts[2 * i + 1]) + norm_add;
}

/* utils::getImgGrad: Utilities/src/imgUtils.cc:233-254 ; grad is N x 2 col-major */
void img_grad(double *grad, const float *img, int h, int w, const double *pts,
	double eps, int n, double pix_mult) {
	double mult = pix_mult / (2 * eps);
	for (int i = 0; i < n; ++i) {
		double cx = pts[2 * i], cy = pts[2 * i + 1];
		double inc = pix_val(img, h, w, cx + eps, cy);
		double dec = pix_val(img, h, w, cx - eps, cy);
		grad[i] = (inc - dec) * mult;
		inc = pix_val(img, h, w, cx, cy + eps);
		dec = pix_val(img, h, w, cx, cy - eps);
		grad[n + i] = (inc - dec) * mult;
	}
}

/* utils::getWarpedImgGrad: Utilities/src/imgUtils.cc:177-202 ; grad_pts is 8 x N */
void warped_img_grad(double *grad, const float *img, int h, int w, const double *gp,
	double eps, int n, double pix_mult) {
	double mult = pix_mult / (2 * eps);
	for (int i = 0; i < n; ++i) {
		const double *p = gp + 8 * i;
		double inc = pix_val(img, h, w, p[0], p[1]);
		double dec = pix_val(img, h, w, p[2], p[3]);
		grad[i] = (inc - dec) * mult;
		inc = pix_val(img, h, w, p[4], p[5]);
		dec = pix_val(img, h, w, p[6], p[7]);
		grad[n + i] = (inc - dec) * mult;
	}
}

/* utils::getImgHess: Utilities/src/imgUtils.cc:334-366 ; hess is 4 x N col-major (xx, xy, yx, yy per pixel) */
void img_hess(double *hess, const float *img, int h, int w, const double *pts,
	double eps, int n, double pix_mult) {
	double eps2 = 2 * eps;
	double mult = pix_mult / (eps2 * eps2);
	for (int i = 0; i < n; ++i) {
		double cx = pts[2 * i], cy = pts[2 * i + 1];
		double c = pix_val(img, h, w, cx, cy);
		double ix = pix_val(img, h, w, cx + eps2, cy);
		double dx = pix_val(img, h, w, cx - eps2, cy);
		hess[4 * i] = (ix + dx - 2 * c) * mult;
		double iy = pix_val(img, h, w, cx, cy + eps2);
		double dy = pix_val(img, h, w, cx, cy - eps2);
		hess[4 * i + 3] = (iy + dy - 2 * c) * mult;
		double inc_x = cx + eps, dec_x = cx - eps, inc_y = cy + eps, dec_y = cy - eps;
		double ixiy = pix_val(img, h, w, inc_x, inc_y);
		double dxdy = pix_val(img, h, w, dec_x, dec_y);
		double ixdy = pix_val(img, h, w, inc_x, dec_y);
		double iydx = pix_val(img, h, w, dec_x, inc_y);
		hess[4 * i + 1] = hess[4 * i + 2] = ((ixiy + dxdy) - (ixdy + iydx)) * mult;
	}
}

/* utils::getWarpedImgHess: Utilities/src/imgUtils.cc:259-289 ; hess_pts is 16 x N */
void warped_img_hess(double *hess, const float *img, int h, int w, const double *pts,
	const double *hp, double eps, int n, double pix_mult) {
	double eps2 = 2 * eps;
	double mult = pix_mult / (eps2 * eps2);
	for (int i = 0; i < n; ++i) {
		const double *p = hp + 16 * i;
		double c = pix_val(img, h, w, pts[2 * i], pts[2 * i + 1]);
		double inc = pix_val(img, h, w, p[0], p[1]);
		double dec = pix_val(img, h, w, p[2], p[3]);
		hess[4 * i] = (inc + dec - 2 * c) * mult;
		inc = pix_val(img, h, w, p[4], p[5]);
		dec = pix_val(img, h, w, p[6], p[7]);
		hess[4 * i + 3] = (inc + dec - 2 * c) * mult;
		inc = pix_val(img, h, w, p[8], p[9]);
		dec = pix_val(img, h, w, p[10], p[11]);
		double inc2 = pix_val(img, h, w, p[12], p[13]);
		double dec2 = pix_val(img, h, w, p[14], p[15]);
		hess[4 * i + 1] = hess[4 * i + 2] = ((inc + dec) - (inc2 + dec2)) * mult;
	}
}

/* ---- multi-channel (mc::) variants: image H x W x C float32 interleaved, outputs interleaved per pixel
 * (ch_pix_id = pix_id * C + channel).  mc::PixVal<Linear, Constant>::get Utilities/include/mtf/Utilities/imgUtils.h:505-551:
 * the four bilinear WEIGHTS are formed first and then applied per channel -- a different rounding order from the
 * single-channel expression, kept as is. ---- */
void pix_val_mc(double *out, const float *img, int h, int w, int C, double x, double y) {
	const double overflow_val = 128.0;
	if (overflow(x, y, h, w)) { for (int c = 0; c < C; ++c) out[c] = overflow_val; return; }
	int lx = static_cast<int>(x), ly = static_cast<int>(y);
	double dx = x - lx, dy = y - ly;
	int ux = dx == 0 ? lx : lx + 1, uy = dy == 0 ? ly : ly + 1;
	if (overflow(lx, ly, h, w) || overflow(ux, uy, h, w)) { for (int c = 0; c < C; ++c) out[c] = overflow_val; return; }
	double ly_lx = (1 - dx) * (1 - dy), ly_ux = dx * (1 - dy), uy_lx = (1 - dx) * dy, uy_ux = dx * dy;
	const float *p00 = img + (static_cast<size_t>(ly) * w + lx) * C, *p01 = img + (static_cast<size_t>(ly) * w + ux) * C;
	const float *p10 = img + (static_cast<size_t>(uy) * w + lx) * C, *p11 = img + (static_cast<size_t>(uy) * w + ux) * C;
	for (int c = 0; c < C; ++c) out[c] = p00[c] * ly_lx + p01[c] * ly_ux + p10[c] * uy_lx + p11[c] * uy_ux;
}
/* mc::getPixVals imgUtils.cc:867-882 */
void pix_vals_mc(double *out, const float *img, int h, int w, int C, const double *pts, int npix, double mult, double add) {
	for (int i = 0; i < npix; ++i) {
		double v[4];
		pix_val_mc(v, img, h, w, C, pts[2 * i], pts[2 * i + 1]);
		for (int c = 0; c < C; ++c) out[i * C + c] = mult * v[c] + add;
	}
}
/* mc::getImgGrad imgUtils.cc:977-1005 ; grad is (npix*C) x 2 col-major */
void img_grad_mc(double *grad, const float *img, int h, int w, int C, const double *pts, double eps, int npix, double pix_mult) {
	double mult = pix_mult / (2 * eps);
	const int P = npix * C;
	for (int i = 0; i < npix; ++i) {
		double cx = pts[2 * i], cy = pts[2 * i + 1];
		double ix[4], dx[4], iy[4], dy[4];
		pix_val_mc(ix, img, h, w, C, cx + eps, cy); pix_val_mc(dx, img, h, w, C, cx - eps, cy);
		pix_val_mc(iy, img, h, w, C, cx, cy + eps); pix_val_mc(dy, img, h, w, C, cx, cy - eps);
		for (int c = 0; c < C; ++c) { grad[i * C + c] = (ix[c] - dx[c]) * mult; grad[P + i * C + c] = (iy[c] - dy[c]) * mult; }
	}
}
/* mc::getWarpedImgGrad imgUtils.cc:914-944 */
void warped_img_grad_mc(double *grad, const float *img, int h, int w, int C, const double *gp, double eps, int npix, double pix_mult) {
	double mult = pix_mult / (2 * eps);
	const int P = npix * C;
	for (int i = 0; i < npix; ++i) {
		const double *p = gp + 8 * i;
		double ix[4], dx[4], iy[4], dy[4];
		pix_val_mc(ix, img, h, w, C, p[0], p[1]); pix_val_mc(dx, img, h, w, C, p[2], p[3]);
		pix_val_mc(iy, img, h, w, C, p[4], p[5]); pix_val_mc(dy, img, h, w, C, p[6], p[7]);
		for (int c = 0; c < C; ++c) { grad[i * C + c] = (ix[c] - dx[c]) * mult; grad[P + i * C + c] = (iy[c] - dy[c]) * mult; }
	}
}
/* mc::getImgHess imgUtils.cc:1127-1168 ; hess is 4 x (npix*C) */
void img_hess_mc(double *hess, const float *img, int h, int w, int C, const double *pts, double eps, int npix, double pix_mult) {
	double eps2 = 2 * eps, mult = pix_mult / (eps2 * eps2);
	for (int i = 0; i < npix; ++i) {
		double cx = pts[2 * i], cy = pts[2 * i + 1];
		double c0[4], ix[4], dx[4], iy[4], dy[4], a[4], b[4], c2[4], d[4];
		pix_val_mc(c0, img, h, w, C, cx, cy);
		pix_val_mc(ix, img, h, w, C, cx + eps2, cy); pix_val_mc(dx, img, h, w, C, cx - eps2, cy);
		pix_val_mc(iy, img, h, w, C, cx, cy + eps2); pix_val_mc(dy, img, h, w, C, cx, cy - eps2);
		double inc_x = cx + eps, dec_x = cx - eps, inc_y = cy + eps, dec_y = cy - eps;
		pix_val_mc(a, img, h, w, C, inc_x, inc_y); pix_val_mc(b, img, h, w, C, dec_x, dec_y);
		pix_val_mc(c2, img, h, w, C, inc_x, dec_y); pix_val_mc(d, img, h, w, C, dec_x, inc_y);
		for (int c = 0; c < C; ++c) {
			double *o = hess + 4 * (i * C + c);
			o[0] = (ix[c] + dx[c] - 2 * c0[c]) * mult;
			o[3] = (iy[c] + dy[c] - 2 * c0[c]) * mult;
			o[1] = o[2] = ((a[c] + b[c]) - (c2[c] + d[c])) * mult;
		}
	}
}
/* mc::getWarpedImgHess imgUtils.cc:1036-1075 */
void warped_img_hess_mc(double *hess, const float *img, int h, int w, int C, const double *pts, const double *hp, double eps,
	int npix, double pix_mult) {
	double eps2 = 2 * eps, mult = pix_mult / (eps2 * eps2);
	for (int i = 0; i < npix; ++i) {
		const double *p = hp + 16 * i;
		double c0[4], s[8][4];
		pix_val_mc(c0, img, h, w, C, pts[2 * i], pts[2 * i + 1]);
		for (int k = 0; k < 8; ++k) pix_val_mc(s[k], img, h, w, C, p[2 * k], p[2 * k + 1]);
		for (int c = 0; c < C; ++c) {
			double *o = hess + 4 * (i * C + c);
			o[0] = (s[0][c] + s[1][c] - 2 * c0[c]) * mult;
			o[3] = (s[2][c] + s[3][c] - 2 * c0[c]) * mult;
			o[1] = o[2] = ((s[4][c] + s[5][c]) - (s[6][c] + s[7][c])) * mult;
		}
	}
}

/* ===================================================================== */
/* small dense math (Eigen in the reference)                              */
/* ===================================================================== */

struct Mat3 {
	double m[9]; /* row-major */
	double &operator()(int r, int c) { return m[3 * r + c]; }
	double operator()(int r, int c) const { return m[3 * r + c]; }
};
Mat3 identity3() { Mat3 I = {{1, 0, 0, 0, 1, 0, 0, 0, 1}}; return I; }
Mat3 mul3(const Mat3 &a, const Mat3 &b) {
	Mat3 c;
	for (int i = 0; i < 3; ++i)
		for (int j = 0; j < 3; ++j)
			c(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
	return c;
}
/* Matrix3d::inverse() (cofactor expansion, as Eigen does for 3x3) */
Mat3 inv3(const Mat3 &a) {
	Mat3 c;
	c(0, 0) = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
	c(0, 1) = a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2);
	c(0, 2) = a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1);
	c(1, 0) = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2);
	c(1, 1) = a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0);
	c(1, 2) = a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2);
	c(2, 0) = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
	c(2, 1) = a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1);
	c(2, 2) = a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0);
	double det = a(0, 0) * c(0, 0) + a(0, 1) * c(1, 0) + a(0, 2) * c(2, 0);
	double inv_det = 1.0 / det;
	for (int i = 0; i < 9; ++i) c.m[i] *= inv_det;
	return c;
}
/* `mat /= scalar`: element-wise division (Eigen >= 3.3 semantics; 3.2.x multiplied by the reciprocal) */
void div3(Mat3 &a, double s) { for (int i = 0; i < 9; ++i) a.m[i] /= s; }

/* Null vector of an 8 x 9 matrix through a one-sided (Hestenes) Jacobi SVD:
 * stands in for JacobiSVD<Matrix89d>(...).matrixV().col(8) at
 * Utilities/src/warpUtils.cc:197-198.  A is row-major 8 x 9. */
void null_vector_8x9(const double *A, double *h) {
	const int m = 8, n = 9;
	double U[m * n], V[n * n];
	std::memcpy(U, A, sizeof(U));
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < n; ++j) V[i * n + j] = i == j ? 1.0 : 0.0;
	for (int sweep = 0; sweep < 60; ++sweep) {
		double off = 0;
		for (int p = 0; p < n - 1; ++p) {
			for (int q = p + 1; q < n; ++q) {
				double alpha = 0, beta = 0, gamma = 0;
				for (int i = 0; i < m; ++i) {
					alpha += U[i * n + p] * U[i * n + p];
					beta += U[i * n + q] * U[i * n + q];
					gamma += U[i * n + p] * U[i * n + q];
				}
				if (gamma == 0) continue;
				double lim = std::sqrt(alpha * beta);
				if (std::fabs(gamma) <= 1e-300 || std::fabs(gamma) <= 1e-17 * lim) continue;
				off = std::max(off, std::fabs(gamma) / (lim > 0 ? lim : 1));
				double zeta = (beta - alpha) / (2.0 * gamma);
				double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
				double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
				for (int i = 0; i < m; ++i) {
					double up = U[i * n + p], uq = U[i * n + q];
					U[i * n + p] = c * up - s * uq;
					U[i * n + q] = s * up + c * uq;
				}
				for (int i = 0; i < n; ++i) {
					double vp = V[i * n + p], vq = V[i * n + q];
					V[i * n + p] = c * vp - s * vq;
					V[i * n + q] = s * vp + c * vq;
				}
			}
		}
		if (off < 1e-15) break;
	}
	int best = 0;
	double best_norm = std::numeric_limits<double>::max();
	for (int j = 0; j < n; ++j) {
		double nn = 0;
		for (int i = 0; i < m; ++i) nn += U[i * n + j] * U[i * n + j];
		if (nn < best_norm) { best_norm = nn; best = j; }
	}
	for (int i = 0; i < n; ++i) h[i] = V[i * n + best];
}

/* utils::computeHomographyDLT(corners): Utilities/src/warpUtils.cc:171-224
 * corners are 2 x 4 column-major */
Mat3 homography_dlt(const double *in, const double *out) {
	double A[8 * 9];
	for (int i = 0; i < 4; ++i) {
		double ix = in[2 * i], iy = in[2 * i + 1];
		double ox = out[2 * i], oy = out[2 * i + 1];
		double *r1 = A + (2 * i) * 9;
		r1[0] = 0; r1[1] = 0; r1[2] = 0;
		r1[3] = -ix; r1[4] = -iy; r1[5] = -1;
		r1[6] = oy * ix; r1[7] = oy * iy; r1[8] = oy;
		double *r2 = A + (2 * i + 1) * 9;
		r2[0] = ix; r2[1] = iy; r2[2] = 1;
		r2[3] = 0; r2[4] = 0; r2[5] = 0;
		r2[6] = -ox * ix; r2[7] = -ox * iy; r2[8] = -ox;
	}
	double h[9];
	null_vector_8x9(A, h);
	Mat3 H;
	for (int i = 0; i < 9; ++i) H.m[i] = h[i] / h[8];
	return H;
}

/* x = A^{-1} b with a column-pivoted Householder QR; stands in for
 * hessian.colPivHouseholderQr().solve(...) (SM/src/NT/FCLK.cc:298, NT/ESM.cc:267,
 * NT/ICLK.cc:264).  A is n x n column-major. */
void colpiv_qr_solve(int n, const double *Ain, const double *b, double *x) {
	vecd A(Ain, Ain + n * n), rhs(b, b + n), cn(n);
	std::vector<int> perm(n);
	for (int j = 0; j < n; ++j) {
		perm[j] = j;
		double s = 0;
		for (int i = 0; i < n; ++i) s += A[j * n + i] * A[j * n + i];
		cn[j] = s;
	}
	int rank = n;
	double max_norm0 = 0;
	for (int j = 0; j < n; ++j) max_norm0 = std::max(max_norm0, cn[j]);
	for (int k = 0; k < n; ++k) {
		int piv = k;
		double best = -1;
		for (int j = k; j < n; ++j) {
			double s = 0;
			for (int i = k; i < n; ++i) s += A[j * n + i] * A[j * n + i];
			cn[j] = s;
			if (s > best) { best = s; piv = j; }
		}
		if (best <= max_norm0 * 1e-300 || best == 0) { rank = k; break; }
		if (piv != k) {
			for (int i = 0; i < n; ++i) std::swap(A[piv * n + i], A[k * n + i]);
			std::swap(perm[piv], perm[k]);
		}
		double *col = &A[k * n];
		double norm = std::sqrt(best);
		double alpha = col[k] > 0 ? -norm : norm;
		vecd v(n, 0.0);
		for (int i = k; i < n; ++i) v[i] = col[i];
		v[k] -= alpha;
		double vnorm2 = 0;
		for (int i = k; i < n; ++i) vnorm2 += v[i] * v[i];
		if (vnorm2 > 0) {
			for (int j = k; j < n; ++j) {
				double dot = 0;
				for (int i = k; i < n; ++i) dot += v[i] * A[j * n + i];
				double f = 2 * dot / vnorm2;
				for (int i = k; i < n; ++i) A[j * n + i] -= f * v[i];
			}
			double dot = 0;
			for (int i = k; i < n; ++i) dot += v[i] * rhs[i];
			double f = 2 * dot / vnorm2;
			for (int i = k; i < n; ++i) rhs[i] -= f * v[i];
		}
	}
	vecd y(n, 0.0);
	for (int k = rank - 1; k >= 0; --k) {
		double s = rhs[k];
		for (int j = k + 1; j < rank; ++j) s -= A[j * n + k] * y[j];
		y[k] = s / A[k * n + k];
	}
	for (int k = 0; k < n; ++k) x[perm[k]] = y[k];
}

/* utils::getNormUnitSquarePts: Utilities/src/warpUtils.cc:15-34
 * (VectorXd::LinSpaced(n, lo, hi): lo + i*step with the last element = hi) */
void norm_unit_square_pts(double *pts, double *corners, int resx, int resy,
	double min_x, double min_y, double max_x, double max_y) {
	auto lin = [](int i, int n, double lo, double hi) {
		if (n == 1) return hi;
		if (i == n - 1) return hi;
		return lo + i * ((hi - lo) / (n - 1));
	};
	int id = 0;
	for (int r = 0; r < resy; ++r)
		for (int c = 0; c < resx; ++c) {
			pts[2 * id] = lin(c, resx, min_x, max_x);
			pts[2 * id + 1] = lin(r, resy, min_y, max_y);
			++id;
		}
	double cx[4] = {min_x, max_x, max_x, min_x};
	double cy[4] = {min_y, min_y, max_y, max_y};
	for (int i = 0; i < 4; ++i) { corners[2 * i] = cx[i]; corners[2 * i + 1] = cy[i]; }
}

void homogenize(const double *p2, double *p3, int n) {
	for (int i = 0; i < n; ++i) { p3[3 * i] = p2[2 * i]; p3[3 * i + 1] = p2[2 * i + 1]; p3[3 * i + 2] = 1; }
}
/* utils::dehomogenize: Utilities/include/mtf/Utilities/warpUtils.h:16-22 */
void dehomogenize(const double *p3, double *p2, int n) {
	for (int i = 0; i < n; ++i) { p2[2 * i] = p3[3 * i] / p3[3 * i + 2]; p2[2 * i + 1] = p3[3 * i + 1] / p3[3 * i + 2]; }
}
void warp_hm(const Mat3 &W, const double *in3, double *out3, int n) {
	for (int i = 0; i < n; ++i) {
		double x = in3[3 * i], y = in3[3 * i + 1], z = in3[3 * i + 2];
		out3[3 * i] = W(0, 0) * x + W(0, 1) * y + W(0, 2) * z;
		out3[3 * i + 1] = W(1, 0) * x + W(1, 1) * y + W(1, 2) * z;
		out3[3 * i + 2] = W(2, 0) * x + W(2, 1) * y + W(2, 2) * z;
	}
}

} // namespace

/* ===================================================================== */
/* L2: state space models                                                 */
/* ===================================================================== */

struct mtfo_ssm {
	int kind, resx, resy, n, S;
	int C = 1, P = 0; /* channels of the paired AM (StateSpaceModel::initialize(corners, n_channels)) and rows n * C */
	vecd norm_pts, norm_pts_hm, norm_corners, norm_corners_hm;
	vecd init_pts, curr_pts, init_pts_hm, curr_pts_hm;
	vecd init_corners, curr_corners, init_corners_hm, curr_corners_hm;
	vecd grad_pts, hess_pts, state;
	Mat3 curr_warp;

	/* ProjectiveBase ctor (SSM/src/ProjectiveBase.cc:9-18); Affine re-does the
	 * normalised grid with pixel-like extents (SSM/src/Affine.cc:47-63) */
	mtfo_ssm(int _kind, int _resx, int _resy) : kind(_kind), resx(_resx), resy(_resy),
		n(_resx * _resy), S(_kind == MTFO_SSM_HOMOGRAPHY ? 8 : 6) {
		P = n;
		norm_pts.resize(2 * n); norm_pts_hm.resize(3 * n);
		norm_corners.resize(8); norm_corners_hm.resize(12);
		init_pts.resize(2 * n); curr_pts.resize(2 * n);
		init_pts_hm.resize(3 * n); curr_pts_hm.resize(3 * n);
		init_corners.resize(8); curr_corners.resize(8);
		init_corners_hm.resize(12); curr_corners_hm.resize(12);
		grad_pts.resize(8 * n); state.assign(S, 0.0);
		curr_warp = identity3();
		if (kind == MTFO_SSM_HOMOGRAPHY)
			norm_unit_square_pts(norm_pts.data(), norm_corners.data(), resx, resy, -0.5, -0.5, 0.5, 0.5);
		else
			norm_unit_square_pts(norm_pts.data(), norm_corners.data(), resx, resy,
				1 - resx / 2.0, 1 - resy / 2.0, resx / 2.0, resy / 2.0);
		homogenize(norm_pts.data(), norm_pts_hm.data(), n);
		homogenize(norm_corners.data(), norm_corners_hm.data(), 4);
		if (kind == MTFO_SSM_AFFINE) {
			init_corners = norm_corners; init_corners_hm = norm_corners_hm;
			init_pts = norm_pts; init_pts_hm = norm_pts_hm;
		}
	}

	/* Homography::getWarpFromState SSM/src/Homography.cc:94-107 ;
	 * Affine::getWarpFromState SSM/src/Affine.cc:116-130 */
	Mat3 warp_from_state(const double *p) const {
		Mat3 W;
		if (kind == MTFO_SSM_HOMOGRAPHY) {
			W(0, 0) = 1 + p[0]; W(0, 1) = p[1]; W(0, 2) = p[2];
			W(1, 0) = p[3]; W(1, 1) = 1 + p[4]; W(1, 2) = p[5];
			W(2, 0) = p[6]; W(2, 1) = p[7]; W(2, 2) = 1;
		} else {
			W(0, 0) = 1 + p[2]; W(0, 1) = p[3]; W(0, 2) = p[0];
			W(1, 0) = p[4]; W(1, 1) = 1 + p[5]; W(1, 2) = p[1];
			W(2, 0) = 0; W(2, 1) = 0; W(2, 2) = 1;
		}
		return W;
	}
	/* Homography::getStateFromWarp :116-132 ; Affine::getStateFromWarp :132-143 */
	void state_from_warp(double *p, const Mat3 &W) const {
		if (kind == MTFO_SSM_HOMOGRAPHY) {
			p[0] = W(0, 0) - 1; p[1] = W(0, 1); p[2] = W(0, 2);
			p[3] = W(1, 0); p[4] = W(1, 1) - 1; p[5] = W(1, 2);
			p[6] = W(2, 0); p[7] = W(2, 1);
		} else {
			p[0] = W(0, 2); p[1] = W(1, 2);
			p[2] = W(0, 0) - 1; p[3] = W(0, 1);
			p[4] = W(1, 0); p[5] = W(1, 1) - 1;
		}
	}

	/* ProjectiveBase::getPtsFromCorners SSM/src/ProjectiveBase.cc:20-25 */
	void pts_from_corners(Mat3 &warp, double *pts, double *pts_hm, const double *corners) {
		warp = homography_dlt(norm_corners.data(), corners);
		warp_hm(warp, norm_pts_hm.data(), pts_hm, n);
		dehomogenize(pts_hm, pts, n);
	}

	/* Homography::setCorners SSM/src/Homography.cc:50-71 (normalized_init = false,
	 * HOM_NORMALIZED_BASIS :7) -- note init_pts_hm keeps the un-normalised third
	 * row of the DLT product; Affine::setCorners SSM/src/Affine.cc:64-88
	 * (normalized_init = false) re-homogenises instead. */
	void set_corners(const double *corners) {
		std::copy(corners, corners + 8, curr_corners.begin());
		homogenize(curr_corners.data(), curr_corners_hm.data(), 4);
		pts_from_corners(curr_warp, curr_pts.data(), curr_pts_hm.data(), curr_corners.data());
		init_corners = curr_corners;
		init_pts = curr_pts;
		if (kind == MTFO_SSM_HOMOGRAPHY) {
			init_corners_hm = curr_corners_hm;
			init_pts_hm = curr_pts_hm;
		} else {
			homogenize(init_corners.data(), init_corners_hm.data(), 4);
			homogenize(init_pts.data(), init_pts_hm.data(), n);
		}
		curr_warp = identity3();
		std::fill(state.begin(), state.end(), 0.0);
	}

	void apply_curr_warp() {
		if (kind == MTFO_SSM_HOMOGRAPHY) {
			warp_hm(curr_warp, init_pts_hm.data(), curr_pts_hm.data(), n);
			warp_hm(curr_warp, init_corners_hm.data(), curr_corners_hm.data(), 4);
			dehomogenize(curr_pts_hm.data(), curr_pts.data(), n);
			dehomogenize(curr_corners_hm.data(), curr_corners.data(), 4);
		} else {
			/* curr_pts = curr_warp.topRows<2>() * init_pts_hm (Affine.cc:104-105,113-114) */
			for (int i = 0; i < n; ++i) {
				const double *q = &init_pts_hm[3 * i];
				curr_pts[2 * i] = curr_warp(0, 0) * q[0] + curr_warp(0, 1) * q[1] + curr_warp(0, 2) * q[2];
				curr_pts[2 * i + 1] = curr_warp(1, 0) * q[0] + curr_warp(1, 1) * q[1] + curr_warp(1, 2) * q[2];
			}
			for (int i = 0; i < 4; ++i) {
				const double *q = &init_corners_hm[3 * i];
				curr_corners[2 * i] = curr_warp(0, 0) * q[0] + curr_warp(0, 1) * q[1] + curr_warp(0, 2) * q[2];
				curr_corners[2 * i + 1] = curr_warp(1, 0) * q[0] + curr_warp(1, 1) * q[1] + curr_warp(1, 2) * q[2];
			}
		}
	}

	/* ProjectiveBase::setState SSM/src/ProjectiveBase.cc:41-49 ; Affine::setState Affine.cc:108-114 */
	void set_state(const double *p) {
		std::copy(p, p + S, state.begin());
		curr_warp = warp_from_state(p);
		apply_curr_warp();
	}

	/* Homography::compositionalUpdate SSM/src/Homography.cc:73-92 ;
	 * Affine::compositionalUpdate SSM/src/Affine.cc:90-106 */
	void compositional_update(const double *dp) {
		Mat3 upd = warp_from_state(dp);
		curr_warp = mul3(curr_warp, upd);
		if (kind == MTFO_SSM_HOMOGRAPHY) div3(curr_warp, curr_warp(2, 2));
		state_from_warp(state.data(), curr_warp);
		apply_curr_warp();
	}

	/* Homography::invertState SSM/src/Homography.cc:109-114 ; Affine.cc:145-150 */
	void invert_state(double *inv_p, const double *p) const {
		Mat3 W = warp_from_state(p);
		Mat3 Wi = inv3(W);
		div3(Wi, Wi(2, 2));
		state_from_warp(inv_p, Wi);
	}

	/* Homography::updateGradPts SSM/src/Homography.cc:803-827 ;
	 * Affine::updateGradPts SSM/src/Affine.cc:293-313 */
	void update_grad_pts(double eps) {
		if (kind == MTFO_SSM_HOMOGRAPHY) {
			double dx[3] = {curr_warp(0, 0) * eps, curr_warp(1, 0) * eps, curr_warp(2, 0) * eps};
			double dy[3] = {curr_warp(0, 1) * eps, curr_warp(1, 1) * eps, curr_warp(2, 1) * eps};
			for (int i = 0; i < n; ++i) {
				const double *q = &curr_pts_hm[3 * i];
				double *g = &grad_pts[8 * i];
				double a0 = q[0] + dx[0], a1 = q[1] + dx[1], a2 = q[2] + dx[2];
				g[0] = a0 / a2; g[1] = a1 / a2;
				a0 = q[0] - dx[0]; a1 = q[1] - dx[1]; a2 = q[2] - dx[2];
				g[2] = a0 / a2; g[3] = a1 / a2;
				a0 = q[0] + dy[0]; a1 = q[1] + dy[1]; a2 = q[2] + dy[2];
				g[4] = a0 / a2; g[5] = a1 / a2;
				a0 = q[0] - dy[0]; a1 = q[1] - dy[1]; a2 = q[2] - dy[2];
				g[6] = a0 / a2; g[7] = a1 / a2;
			}
		} else {
			double dx[2] = {curr_warp(0, 0) * eps, curr_warp(1, 0) * eps};
			double dy[2] = {curr_warp(0, 1) * eps, curr_warp(1, 1) * eps};
			for (int i = 0; i < n; ++i) {
				double cx = curr_pts[2 * i], cy = curr_pts[2 * i + 1];
				double *g = &grad_pts[8 * i];
				g[0] = cx + dx[0]; g[1] = cy + dx[1];
				g[2] = cx - dx[0]; g[3] = cy - dx[1];
				g[4] = cx + dy[0]; g[5] = cy + dy[1];
				g[6] = cx - dy[0]; g[7] = cy - dy[1];
			}
		}
	}

	/* row writers shared by the pixel-Jacobian variants */
	inline void hom_row(double *J, int q, double Ix, double Iy, double x, double y,
		double px, double py) const {
		double Ixx = Ix * x, Iyy = Iy * y, Ixy = Ix * y, Iyx = Iy * x;
		J[0 * P + q] = Ixx; J[1 * P + q] = Ixy; J[2 * P + q] = Ix;
		J[3 * P + q] = Iyx; J[4 * P + q] = Iyy; J[5 * P + q] = Iy;
		J[6 * P + q] = -px * Ixx - py * Iyx;
		J[7 * P + q] = -px * Ixy - py * Iyy;
	}

	/* Homography::cmptInitPixJacobian SSM/src/Homography.cc:157-191 ;
	 * Affine::cmptInitPixJacobian SSM/src/Affine.cc:160-182 */
	void init_pix_jacobian(double *J, const double *g) const {
		for (int i = 0; i < n; ++i) {
			for (int ch = 0; ch < C; ++ch) {
			const int q = i * C + ch;
			double x = init_pts[2 * i], y = init_pts[2 * i + 1];
			double Ix = g[q], Iy = g[P + q];
			if (kind == MTFO_SSM_HOMOGRAPHY) {
				hom_row(J, q, Ix, Iy, x, y, x, y);
			} else {
				J[0 * P + q] = Ix; J[1 * P + q] = Iy;
				J[2 * P + q] = Ix * x; J[3 * P + q] = Ix * y;
				J[4 * P + q] = Iy * x; J[5 * P + q] = Iy * y;
			}
			}
		}
	}
	/* Homography::cmptPixJacobian SSM/src/Homography.cc:193-229 ;
	 * Affine::cmptPixJacobian == cmptInitPixJacobian (Affine.h:35-37) */
	void pix_jacobian(double *J, const double *g) const {
		if (kind == MTFO_SSM_AFFINE) { init_pix_jacobian(J, g); return; }
		for (int i = 0; i < n; ++i) {
			for (int ch = 0; ch < C; ++ch) {
			const int q = i * C + ch;
			double x = init_pts[2 * i], y = init_pts[2 * i + 1];
			double cx = curr_pts[2 * i], cy = curr_pts[2 * i + 1];
			double inv_d = 1.0 / curr_pts_hm[3 * i + 2];
			double Ix = g[q] * inv_d, Iy = g[P + q] * inv_d;
			hom_row(J, q, Ix, Iy, x, y, cx, cy);
			}
		}
	}
	/* Homography::cmptWarpedPixJacobian SSM/src/Homography.cc:231-294 ;
	 * Affine::cmptWarpedPixJacobian SSM/src/Affine.cc:213-242 */
	void warped_pix_jacobian(double *J, const double *g) const {
		if (kind == MTFO_SSM_HOMOGRAPHY) {
			double a00 = curr_warp(0, 0), a01 = curr_warp(0, 1);
			double a10 = curr_warp(1, 0), a11 = curr_warp(1, 1);
			double a20 = curr_warp(2, 0), a21 = curr_warp(2, 1);
			for (int i = 0; i < n; ++i) {
				for (int ch = 0; ch < C; ++ch) {
				const int q = i * C + ch;
				double wx = curr_pts[2 * i], wy = curr_pts[2 * i + 1];
				double D = curr_pts_hm[3 * i + 2];
				double inv_det = 1.0 / D;
				double dwx_dx = (a00 - a20 * wx), dwx_dy = (a01 - a21 * wx);
				double dwy_dx = (a10 - a20 * wy), dwy_dy = (a11 - a21 * wy);
				double x = init_pts[2 * i], y = init_pts[2 * i + 1];
				double Ix = (dwx_dx * g[q] + dwy_dx * g[P + q]) * inv_det;
				double Iy = (dwx_dy * g[q] + dwy_dy * g[P + q]) * inv_det;
				hom_row(J, q, Ix, Iy, x, y, x, y);
				}
			}
		} else {
			double a = state[2] + 1, b = state[3], c = state[4], d = state[5] + 1;
			for (int i = 0; i < n; ++i) {
				for (int ch = 0; ch < C; ++ch) {
				const int q = i * C + ch;
				double x = init_pts[2 * i], y = init_pts[2 * i + 1];
				double Ix = g[q], Iy = g[P + q];
				double Ixx = Ix * x, Ixy = Ix * y, Iyy = Iy * y, Iyx = Iy * x;
				J[0 * P + q] = Ix * a + Iy * c;
				J[1 * P + q] = Ix * b + Iy * d;
				J[2 * P + q] = Ixx * a + Iyx * c;
				J[3 * P + q] = Ixy * a + Iyy * c;
				J[4 * P + q] = Ixx * b + Iyx * d;
				J[5 * P + q] = Ixy * b + Iyy * d;
				}
			}
		}
	}
	/* Homography::cmptApproxPixJacobian SSM/src/Homography.cc:296-358 ;
	 * Affine::cmptApproxPixJacobian SSM/src/Affine.cc:184-211 */
	void approx_pix_jacobian(double *J, const double *g) const {
		if (kind == MTFO_SSM_HOMOGRAPHY) {
			double h00 = curr_warp(0, 0), h01 = curr_warp(0, 1);
			double h10 = curr_warp(1, 0), h11 = curr_warp(1, 1);
			double h20 = curr_warp(2, 0), h21 = curr_warp(2, 1);
			for (int i = 0; i < n; ++i) {
				for (int ch = 0; ch < C; ++ch) {
				const int q = i * C + ch;
				double cx = curr_pts[2 * i], cy = curr_pts[2 * i + 1];
				double a = (h00 - h20 * cx), b = (h01 - h21 * cx);
				double c = (h10 - h20 * cy), d = (h11 - h21 * cy);
				double inv_factor = 1.0 / (a * d - b * c);
				double x = init_pts[2 * i], y = init_pts[2 * i + 1];
				double Ix = (d * g[q] - c * g[P + q]) * inv_factor;
				double Iy = (a * g[P + q] - b * g[q]) * inv_factor;
				hom_row(J, q, Ix, Iy, x, y, cx, cy);
				}
			}
		} else {
			double a = state[2] + 1, b = state[3], c = state[4], d = state[5] + 1;
			double inv_det = 1.0 / (a * d - b * c);
			for (int i = 0; i < n; ++i) {
				for (int ch = 0; ch < C; ++ch) {
				const int q = i * C + ch;
				double x = init_pts[2 * i], y = init_pts[2 * i + 1];
				double Ix = g[q], Iy = g[P + q];
				double Ixx = Ix * x, Ixy = Ix * y, Iyy = Iy * y, Iyx = Iy * x;
				J[0 * P + q] = (Ix * d - Iy * c) * inv_det;
				J[1 * P + q] = (Iy * a - Ix * b) * inv_det;
				J[2 * P + q] = (Ixx * d - Iyx * c) * inv_det;
				J[3 * P + q] = (Ixy * d - Iyy * c) * inv_det;
				J[4 * P + q] = (Iyx * a - Ixx * b) * inv_det;
				J[5 * P + q] = (Iyy * a - Ixy * b) * inv_det;
				}
			}
		}
	}
	/* Homography::updateHessPts SSM/src/Homography.cc:829-875 (= ProjectiveBase.cc:88-129) ;
	 * Affine::updateHessPts SSM/src/Affine.cc:315-350 ; hess_pts is 16 x N */
	void update_hess_pts(double eps) {
		hess_pts.resize(16 * static_cast<size_t>(n));
		double eps2 = 2 * eps;
		int R = kind == MTFO_SSM_HOMOGRAPHY ? 3 : 2;
		double xx[3], yy[3], xy[3], yx[3];
		for (int r = 0; r < R; ++r) {
			xx[r] = curr_warp(r, 0) * eps2;
			yy[r] = curr_warp(r, 1) * eps2;
			xy[r] = (curr_warp(r, 0) + curr_warp(r, 1)) * eps;
			yx[r] = (curr_warp(r, 0) - curr_warp(r, 1)) * eps;
		}
		const double *dv[4] = {xx, yy, xy, yx};
		for (int i = 0; i < n; ++i) {
			double *hp = &hess_pts[16 * static_cast<size_t>(i)];
			for (int k = 0; k < 4; ++k) {
				if (kind == MTFO_SSM_HOMOGRAPHY) {
					const double *q = &curr_pts_hm[3 * i];
					double a0 = q[0] + dv[k][0], a1 = q[1] + dv[k][1], a2 = q[2] + dv[k][2];
					hp[4 * k] = a0 / a2; hp[4 * k + 1] = a1 / a2;
					a0 = q[0] - dv[k][0]; a1 = q[1] - dv[k][1]; a2 = q[2] - dv[k][2];
					hp[4 * k + 2] = a0 / a2; hp[4 * k + 3] = a1 / a2;
				} else {
					double cx = curr_pts[2 * i], cy = curr_pts[2 * i + 1];
					hp[4 * k] = cx + dv[k][0]; hp[4 * k + 1] = cy + dv[k][1];
					hp[4 * k + 2] = cx - dv[k][0]; hp[4 * k + 3] = cy - dv[k][1];
				}
			}
		}
	}

	/* out(SxS, col-major) = dw_dp^T * M * dw_dp for a 2 x S dw_dp (rows r0, r1) and 2x2 M (m00 m01; m10 m11) */
	inline void sandwich(double *out, const double *r0, const double *r1, double m00, double m01, double m10, double m11) const {
		double a0[8], a1[8];
		for (int j = 0; j < S; ++j) { a0[j] = m00 * r0[j] + m01 * r1[j]; a1[j] = m10 * r0[j] + m11 * r1[j]; }
		for (int j = 0; j < S; ++j)
			for (int i = 0; i < S; ++i) out[j * S + i] = r0[i] * a0[j] + r1[i] * a1[j];
	}
	/* the "+= / -= third-order" tail shared by Homography's Init / Warped / Approx pixel Hessians:
	 * columns 6,7 of rows 0..5 get sgn*(..), the 2x2 corner gets corner*(..), then the 2x5 bottom-left block is
	 * overwritten by the transpose of the 5x2 top-right block (rows 0..4 only, as the reference does). */
	inline void hom_tail(double *d2, double Ix, double Iy, double x, double y, double sgn, double corner,
		double cx, double cy) const {
		double Ixx = Ix * x, Ixy = Ix * y, Iyy = Iy * y, Iyx = Iy * x;
		double Ixxx = Ixx * x, Ixxy = Ixx * y, Ixyy = Ixy * y;
		double Iyyy = Iyy * y, Iyyx = Iyy * x, Iyxx = Iyx * x;
#define D2(r, c) d2[(c) * 8 + (r)]
		D2(0, 6) += sgn * Ixxx; D2(0, 7) += sgn * Ixxy;
		D2(1, 6) += sgn * Ixxy; D2(1, 7) += sgn * Ixyy;
		D2(2, 6) += sgn * Ixx;  D2(2, 7) += sgn * Ixy;
		D2(3, 6) += sgn * Iyxx; D2(3, 7) += sgn * Iyyx;
		D2(4, 6) += sgn * Iyyx; D2(4, 7) += sgn * Iyyy;
		D2(5, 6) += sgn * Iyx;  D2(5, 7) += sgn * Iyy;
		D2(6, 6) += corner * (Ixxx * cx + Iyxx * cy);
		D2(6, 7) += corner * (Ixxy * cx + Iyyx * cy);
		D2(7, 6) += corner * (Ixxy * cx + Iyyx * cy);
		D2(7, 7) += corner * (Ixyy * cx + Iyyy * cy);
		for (int r = 0; r < 5; ++r) { D2(6, r) = D2(r, 6); D2(7, r) = D2(r, 7); }
#undef D2
	}
	inline void dw_dp_rows(double *r0, double *r1, double x, double y, double px, double py) const {
		if (kind == MTFO_SSM_HOMOGRAPHY) {
			double a[8] = {x, y, 1, 0, 0, 0, -px * x, -px * y};
			double b[8] = {0, 0, 0, x, y, 1, -py * x, -py * y};
			std::copy(a, a + 8, r0); std::copy(b, b + 8, r1);
		} else {
			double a[6] = {1, 0, x, y, 0, 0};
			double b[6] = {0, 1, 0, 0, x, y};
			std::copy(a, a + 6, r0); std::copy(b, b + 6, r1);
		}
	}
	/* Homography::cmptInitPixHessian SSM/src/Homography.cc:360-425 ; Affine::cmptInitPixHessian SSM/src/Affine.cc:243-263.
	 * d2 is S^2 x N (one column-major S x S block per pixel), pix_hess 4 x N, pix_grad N x 2 */
	void init_pix_hessian(double *d2, const double *ph, const double *g) const {
		double r0[8] = {0}, r1[8] = {0};
		for (int i = 0; i < n; ++i) {
			for (int ch = 0; ch < C; ++ch) {
			const int q = i * C + ch;
			double x = init_pts[2 * i], y = init_pts[2 * i + 1];
			double *out = d2 + static_cast<size_t>(q) * S * S;
			const double *m = ph + 4 * q;   /* Map<Matrix2d>: col-major (m0 m2; m1 m3) */
			dw_dp_rows(r0, r1, x, y, x, y);
			sandwich(out, r0, r1, m[0], m[2], m[1], m[3]);
			if (kind == MTFO_SSM_HOMOGRAPHY) hom_tail(out, g[q], g[P + q], x, y, -1.0, 2.0, x, y);
			}
		}
	}
	/* Homography::cmptPixHessian SSM/src/Homography.cc:427-513 (Affine: not implemented, StateSpaceModel.h:186-189) */
	int pix_hessian(double *d2, const double *ph, const double *g) const {
		if (kind != MTFO_SSM_HOMOGRAPHY) return -2;
		double r0[8] = {0}, r1[8] = {0};
		for (int i = 0; i < n; ++i) {
			for (int ch = 0; ch < C; ++ch) {
			const int q = i * C + ch;
			double x = init_pts[2 * i], y = init_pts[2 * i + 1];
			double cx = curr_pts[2 * i], cy = curr_pts[2 * i + 1];
			double D = curr_pts_hm[3 * i + 2];
			double *out = d2 + static_cast<size_t>(q) * 64;
			const double *m = ph + 4 * q;
			dw_dp_rows(r0, r1, x, y, cx, cy);
			for (int j = 0; j < 8; ++j) { r0[j] /= D; r1[j] /= D; }
			double inv_d2 = 1.0 / (D * D);
			sandwich(out, r0, r1, m[0], m[2], m[1], m[3]);
			/* the reference scales each third-order product by inv_d_squared (:490-507) */
			double Ix = g[q], Iy = g[P + q];
			double Ixx = Ix * x, Ixy = Ix * y, Iyy = Iy * y, Iyx = Iy * x;
			double Ixxx = Ixx * x, Ixxy = Ixx * y, Ixyy = Ixy * y;
			double Iyyy = Iyy * y, Iyyx = Iyy * x, Iyxx = Iyx * x;
#define D2(r, c) out[(c) * 8 + (r)]
			D2(0, 6) -= Ixxx * inv_d2; D2(1, 6) -= Ixxy * inv_d2; D2(2, 6) -= Ixx * inv_d2;
			D2(3, 6) -= Iyxx * inv_d2; D2(4, 6) -= Iyyx * inv_d2; D2(5, 6) -= Iyx * inv_d2;
			D2(6, 6) += 2 * (Ixxx * cx + Iyxx * cy) * inv_d2;
			D2(7, 6) += 2 * (Ixxy * cx + Iyyx * cy) * inv_d2;
			D2(0, 7) -= Ixxy * inv_d2; D2(1, 7) -= Ixyy * inv_d2; D2(2, 7) -= Ixy * inv_d2;
			D2(3, 7) -= Iyyx * inv_d2; D2(4, 7) -= Iyyy * inv_d2; D2(5, 7) -= Iyy * inv_d2;
			D2(6, 7) += 2 * (Ixxy * cx + Iyyx * cy) * inv_d2;
			D2(7, 7) += 2 * (Ixyy * cx + Iyyy * cy) * inv_d2;
			for (int r = 0; r < 5; ++r) { D2(6, r) = D2(r, 6); D2(7, r) = D2(r, 7); }
#undef D2
			}
		}
		return 0;
	}
	/* Homography::cmptWarpedPixHessian SSM/src/Homography.cc:515-618 ; Affine::cmptWarpedPixHessian SSM/src/Affine.cc:264-291 */
	void warped_pix_hessian(double *d2, const double *ph, const double *g) const {
		double r0[8] = {0}, r1[8] = {0};
		if (kind == MTFO_SSM_AFFINE) {
			double a2 = state[2] + 1, a3 = state[3], a4 = state[4], a5 = state[5] + 1;
			for (int i = 0; i < n; ++i) {
				for (int ch = 0; ch < C; ++ch) {
				const int q = i * C + ch;
				double x = init_pts[2 * i], y = init_pts[2 * i + 1];
				const double *m = ph + 4 * q;
				/* dw_dx^T * M * dw_dx with dw_dx = (a2 a3; a4 a5) */
				double t00 = m[0] * a2 + m[2] * a4, t01 = m[0] * a3 + m[2] * a5;
				double t10 = m[1] * a2 + m[3] * a4, t11 = m[1] * a3 + m[3] * a5;
				double q00 = a2 * t00 + a4 * t10, q01 = a2 * t01 + a4 * t11;
				double q10 = a3 * t00 + a5 * t10, q11 = a3 * t01 + a5 * t11;
				dw_dp_rows(r0, r1, x, y, x, y);
				sandwich(d2 + static_cast<size_t>(q) * 36, r0, r1, q00, q01, q10, q11);
				}
			}
			return;
		}
		double a00 = curr_warp(0, 0), a01 = curr_warp(0, 1), a10 = curr_warp(1, 0), a11 = curr_warp(1, 1);
		double a20 = curr_warp(2, 0), a21 = curr_warp(2, 1);
		for (int i = 0; i < n; ++i) {
			for (int ch = 0; ch < C; ++ch) {
			const int q = i * C + ch;
			double wx = curr_pts[2 * i], wy = curr_pts[2 * i + 1];
			double D = curr_pts_hm[3 * i + 2], D_inv = 1.0 / D;
			double dwx_dx = (a00 - a20 * wx) * D_inv, dwx_dy = (a01 - a21 * wx) * D_inv;
			double dwy_dx = (a10 - a20 * wy) * D_inv, dwy_dy = (a11 - a21 * wy) * D_inv;
			double d2wx_dx2 = -2 * a20 * dwx_dx * D_inv, d2wx_dxdy = -(a21 * dwx_dx + a20 * dwx_dy) * D_inv;
			double d2wx_dy2 = -2 * a21 * dwx_dy * D_inv;
			double d2wy_dx2 = -2 * a20 * dwy_dx * D_inv, d2wy_dxdy = -(a21 * dwy_dx + a20 * dwy_dy) * D_inv;
			double d2wy_dy2 = -2 * a21 * dwy_dy * D_inv;
			double x = init_pts[2 * i], y = init_pts[2 * i + 1];
			const double *m = ph + 4 * q;
			double gx = g[q], gy = g[P + q];
			/* dw_dX^T * M * dw_dX, dw_dX = (dwx_dx dwx_dy; dwy_dx dwy_dy) */
			double t00 = m[0] * dwx_dx + m[2] * dwy_dx, t01 = m[0] * dwx_dy + m[2] * dwy_dy;
			double t10 = m[1] * dwx_dx + m[3] * dwy_dx, t11 = m[1] * dwx_dy + m[3] * dwy_dy;
			double q00 = dwx_dx * t00 + dwy_dx * t10, q01 = dwx_dx * t01 + dwy_dx * t11;
			double q10 = dwx_dy * t00 + dwy_dy * t10, q11 = dwx_dy * t01 + dwy_dy * t11;
			q00 = q00 + gx * d2wx_dx2 + gy * d2wy_dx2;
			q01 = q01 + gx * d2wx_dxdy + gy * d2wy_dxdy;
			q10 = q10 + gx * d2wx_dxdy + gy * d2wy_dxdy;
			q11 = q11 + gx * d2wx_dy2 + gy * d2wy_dy2;
			double *out = d2 + static_cast<size_t>(q) * 64;
			dw_dp_rows(r0, r1, x, y, x, y);
			sandwich(out, r0, r1, q00, q01, q10, q11);
			double Ix = dwx_dx * gx + dwy_dx * gy;
			double Iy = dwx_dy * gx + dwy_dy * gy;
			hom_tail(out, Ix, Iy, x, y, -1.0, 2.0, x, y);
			}
		}
	}
	/* Homography::cmptApproxPixHessian SSM/src/Homography.cc:696-801 (Affine: not implemented) */
	int approx_pix_hessian(double *d2, const double *ph, const double *g) const {
		if (kind != MTFO_SSM_HOMOGRAPHY) return -2;
		double h00 = curr_warp(0, 0), h01 = curr_warp(0, 1), h10 = curr_warp(1, 0), h11 = curr_warp(1, 1);
		double h20 = curr_warp(2, 0), h21 = curr_warp(2, 1);
		double r0[8] = {0}, r1[8] = {0};
		for (int i = 0; i < n; ++i) {
			for (int ch = 0; ch < C; ++ch) {
			const int q = i * C + ch;
			double cx = curr_pts[2 * i], cy = curr_pts[2 * i + 1];
			double D = curr_pts_hm[3 * i + 2];
			double inv_det2 = 1.0 / (D * D), inv_det = 1.0 / D;
			double a = (h00 - h20 * cx) * inv_det, b = (h01 - h21 * cx) * inv_det;
			double c = (h10 - h20 * cy) * inv_det, d = (h11 - h21 * cy) * inv_det;
			double inv_factor = 1.0 / (a * d - b * c);
			double i00 = d * inv_factor, i01 = -b * inv_factor, i10 = -c * inv_factor, i11 = a * inv_factor; /* dw_dx_inv */
			double ax = -h20 * (h00 + a * D - h20 * cx) * inv_det2;
			double bx = -(h20 * h01 + h21 * (a * D - h20 * cx)) * inv_det2;
			double cxx = -h20 * (h10 + c * D - h20 * cy) * inv_det2;
			double dx = -(h20 * h11 + h21 * (c * D - h20 * cy)) * inv_det2;
			double ay = -(h21 * h00 + h20 * (b * D - h21 * cx)) * inv_det2;
			double by = -h21 * (h01 + b * D - h21 * cx) * inv_det2;
			double cyy = -(h21 * h10 + h20 * (d * D - h21 * cy)) * inv_det2;
			double dy = -h21 * (h11 + d * D - h21 * cy) * inv_det2;
			double x = init_pts[2 * i], y = init_pts[2 * i + 1];
			const double *m = ph + 4 * q;
			double Ix = (d * g[q] - c * g[P + q]) * inv_factor;
			double Iy = (a * g[P + q] - b * g[q]) * inv_factor;
			/* inner = M - (Ix * d2w_dx2_x + Iy * d2w_dx2_y); `<<` fills row-major: (ax bx; cx dx) */
			double n00 = m[0] - (Ix * ax + Iy * ay), n01 = m[2] - (Ix * bx + Iy * by);
			double n10 = m[1] - (Ix * cxx + Iy * cyy), n11 = m[3] - (Ix * dx + Iy * dy);
			/* dw_dx_inv^T * inner * dw_dx_inv */
			double t00 = n00 * i00 + n01 * i10, t01 = n00 * i01 + n01 * i11;
			double t10 = n10 * i00 + n11 * i10, t11 = n10 * i01 + n11 * i11;
			double q00 = i00 * t00 + i10 * t10, q01 = i00 * t01 + i10 * t11;
			double q10 = i01 * t00 + i11 * t10, q11 = i01 * t01 + i11 * t11;
			double *out = d2 + static_cast<size_t>(q) * 64;
			dw_dp_rows(r0, r1, x, y, x, y);
			sandwich(out, r0, r1, q00, q01, q10, q11);
			hom_tail(out, Ix, Iy, x, y, 1.0, -1.0, x, y);
			}
		}
		return 0;
	}

	/* ProjectiveBase::applyWarpToCorners SSM/src/ProjectiveBase.cc:137-144 ;
	 * Affine::applyWarpToCorners SSM/src/Affine.cc:366-375 */
	void apply_warp_to_corners(double *out, const double *in, const double *p) const {
		Mat3 W = warp_from_state(p);
		for (int i = 0; i < 4; ++i) {
			double x = in[2 * i], y = in[2 * i + 1];
			if (kind == MTFO_SSM_HOMOGRAPHY) {
				double discr = W(2, 0) * x + W(2, 1) * y + W(2, 2);
				out[2 * i] = (W(0, 0) * x + W(0, 1) * y + W(0, 2)) / discr;
				out[2 * i + 1] = (W(1, 0) * x + W(1, 1) * y + W(1, 2)) / discr;
			} else {
				out[2 * i] = W(0, 0) * x + W(0, 1) * y + W(0, 2);
				out[2 * i + 1] = W(1, 0) * x + W(1, 1) * y + W(1, 2);
			}
		}
	}
	/* ProjectiveBase::applyWarpToPts SSM/src/ProjectiveBase.cc:142-160 ; Affine::applyWarpToPts SSM/src/Affine.cc:382-393 */
	void apply_warp_to_pts(double *out, const double *in, int n_pts, const double *p) const {
		Mat3 W = warp_from_state(p);
		for (int i = 0; i < n_pts; ++i) {
			double x = in[2 * i], y = in[2 * i + 1];
			if (kind == MTFO_SSM_HOMOGRAPHY) {
				double discr = W(2, 0) * x + W(2, 1) * y + W(2, 2);
				out[2 * i] = (W(0, 0) * x + W(0, 1) * y + W(0, 2)) / discr;
				out[2 * i + 1] = (W(1, 0) * x + W(1, 1) * y + W(1, 2)) / discr;
			} else {
				out[2 * i] = W(0, 0) * x + W(0, 1) * y + W(0, 2);
				out[2 * i + 1] = W(1, 0) * x + W(1, 1) * y + W(1, 2);
			}
		}
	}
	/* ProjectiveBase::composeWarps SSM/src/ProjectiveBase.cc:324-331: warp_2 * warp_1, state read back without
	 * renormalising (2, 2) */
	void compose_warps(double *out, const double *p1, const double *p2) const {
		Mat3 W1 = warp_from_state(p1), W2 = warp_from_state(p2);
		state_from_warp(out, mul3(W2, W1));
	}
	/* Homography::estimateWarpFromCorners SSM/src/Homography.cc:877-883 (DLT, / (2, 2)) ;
	 * Affine::estimateWarpFromCorners SSM/src/Affine.cc:352-357 with utils::computeAffineDLT
	 * Utilities/src/warpUtils.cc:276-342: least-squares solution of the 8 x 6 system [x y 1 0 0 0; 0 0 0 x y 1] a = out
	 * (the reference takes it from the thin SVD; the normal equations give the same minimiser for a full-rank system) */
	void estimate_warp_from_corners(double *out, const double *in_c, const double *out_c) const {
		if (kind == MTFO_SSM_HOMOGRAPHY) {
			Mat3 H = homography_dlt(in_c, out_c);
			div3(H, H(2, 2));
			state_from_warp(out, H);
			return;
		}
		double AtA[36] = {0}, Atb[6] = {0}, x[6];
		for (int i = 0; i < 4; ++i) {
			double r1[6] = {in_c[2 * i], in_c[2 * i + 1], 1, 0, 0, 0}, r2[6] = {0, 0, 0, in_c[2 * i], in_c[2 * i + 1], 1};
			for (int a = 0; a < 6; ++a) {
				for (int b = 0; b < 6; ++b) AtA[b * 6 + a] += r1[a] * r1[b] + r2[a] * r2[b];
				Atb[a] += r1[a] * out_c[2 * i] + r2[a] * out_c[2 * i + 1];
			}
		}
		colpiv_qr_solve(6, AtA, Atb, x);
		Mat3 W = identity3();
		W(0, 0) = x[0]; W(0, 1) = x[1]; W(0, 2) = x[2]; W(1, 0) = x[3]; W(1, 1) = x[4]; W(1, 2) = x[5];
		state_from_warp(out, W);
	}
	/* ProjectiveBase::additiveUpdate SSM/src/ProjectiveBase.cc:51-55 */
	void additive_update(const double *dp) {
		vecd p(state);
		for (int i = 0; i < S; ++i) p[i] += dp[i];
		set_state(p.data());
	}
	/* Homography::compositionalRandomWalk SSM/src/Homography.cc:916-926 with the
	 * perturbation supplied by the caller (the Boost RNG is not reproducible) */
	void compositional_random_walk(double *out, const double *base, const double *pert) const {
		Mat3 B = warp_from_state(base), P = warp_from_state(pert);
		Mat3 W = mul3(B, P);
		if (kind == MTFO_SSM_HOMOGRAPHY) div3(W, W(2, 2));
		state_from_warp(out, W);
	}
	/* Affine::geomToState SSM/src/Affine.cc:393-410: (tx, ty, s, theta, r, phi) -> state */
	static void affine_geom_to_state(double *state, const double *geom) {
		double s = geom[2], r = geom[4];
		double theta = geom[3], phi = geom[5];
		double cos_theta = std::cos(theta), sin_theta = std::sin(theta);
		double cos_phi = std::cos(phi), sin_phi = std::sin(phi);
		double ccc = cos_theta * cos_phi * cos_phi;
		double ccs = cos_theta * cos_phi * sin_phi;
		double css = cos_theta * sin_phi * sin_phi;
		double scc = sin_theta * cos_phi * cos_phi;
		double scs = sin_theta * cos_phi * sin_phi;
		double sss = sin_theta * sin_phi * sin_phi;
		state[0] = geom[0];
		state[1] = geom[1];
		state[2] = s * (ccc + scs + r * (css - scs)) - 1;
		state[3] = s * (r * (ccs - scc) - ccs - sss);
		state[4] = s * (scc - ccs + r * (ccs + sss));
		state[5] = s * (r * (ccc + scs) - scs + css) - 1;
	}
	/* Affine::generatePerturbation SSM/src/Affine.cc:464-503 with the draws supplied: draw(j) = mean[j] + sigma[j] z.
	 * pt_based 1: coordinate j of (bottom right, bottom left, top centre) disturbed by distribution j (z[0..6));
	 * pt_based 2: every coordinate by distribution 1 (z[0..6)), then one translation from distribution 0 (z[6], z[7]);
	 * the perturbation is utils::computeAffineDLT of the three pairs (Utilities/src/warpUtils.cc:388-421: V (U^T b / S) of the
	 * 6 x 6 system = its exact solution for non-collinear points, taken here with the pivoted QR solve);
	 * pt_based 0: geomToState of six draws (z[0..6)). */
	void affine_generate_perturbation(double *pert, int pt_based, const double *mean, const double *sigma, const double *z) const {
		if (pt_based) {
			double orig[6], pts[6];   /* x, y interleaved per point: init_corners.col(2), col(3), (col(0) + col(1)) / 2 */
			orig[0] = init_corners[4]; orig[1] = init_corners[5];
			orig[2] = init_corners[6]; orig[3] = init_corners[7];
			orig[4] = (init_corners[0] + init_corners[2]) / 2.0; orig[5] = (init_corners[1] + init_corners[3]) / 2.0;
			if (pt_based == 1) {
				for (int j = 0; j < 6; ++j) pts[j] = orig[j] + (mean[j] + sigma[j] * z[j]);
			} else {
				const double tx = mean[0] + sigma[0] * z[6], ty = mean[0] + sigma[0] * z[7];
				for (int i = 0; i < 3; ++i) {
					pts[2 * i] = (orig[2 * i] + (mean[1] + sigma[1] * z[2 * i])) + tx;
					pts[2 * i + 1] = (orig[2 * i + 1] + (mean[1] + sigma[1] * z[2 * i + 1])) + ty;
				}
			}
			double A[36] = {0}, x[6];   /* column-major 6 x 6 */
			for (int i = 0; i < 3; ++i) {
				const int r1 = 2 * i, r2 = 2 * i + 1;
				A[0 * 6 + r1] = orig[2 * i]; A[1 * 6 + r1] = orig[2 * i + 1]; A[2 * 6 + r1] = 1;
				A[3 * 6 + r2] = orig[2 * i]; A[4 * 6 + r2] = orig[2 * i + 1]; A[5 * 6 + r2] = 1;
			}
			colpiv_qr_solve(6, A, pts, x);
			Mat3 W = identity3();
			W(0, 0) = x[0]; W(0, 1) = x[1]; W(0, 2) = x[2]; W(1, 0) = x[3]; W(1, 1) = x[4]; W(1, 2) = x[5];
			state_from_warp(pert, W);
		} else {
			double geom[6];
			for (int j = 0; j < 6; ++j) geom[j] = mean[j] + sigma[j] * z[j];
			affine_geom_to_state(pert, geom);
		}
	}
};

/* ===================================================================== */
/* L2: appearance models                                                  */
/* ===================================================================== */

namespace {
/* utils::bSpl3WithGrad Utilities/include/mtf/Utilities/histUtils.h:206-226
 * (keeps the truncated constant _2_BY_3 = 0.66666666666, histUtils.h:11) */
const double k2By3 = 0.66666666666;
inline void bspl3_with_grad(double &val, double &diff, double x) {
	if ((x > -2) && (x <= -1)) {
		double t = 2 + x; diff = (t * t) / 2; val = (diff * t) / 3;
	} else if ((x > -1) && (x <= 0)) {
		double t = x / 2; val = k2By3 - x * x * (1 + t); diff = -x * (t + x + 2);
	} else if ((x > 0) && (x <= 1)) {
		double t = x / 2; val = k2By3 - x * x * (1 - t); diff = x * (t + x - 2);
	} else if ((x > 1) && (x < 2)) {
		double t = 2 - x; diff = -(t * t) / 2; val = -(diff * t) / 3;
	}
}
/* utils::bSpl3Hess histUtils.h:271-283 */
inline double bspl3_hess(double x) {
	if ((x > -2) && (x <= -1)) return 2 + x;
	if ((x > -1) && (x <= 0)) return -(3 * x + 2);
	if ((x > 0) && (x <= 1)) return 3 * x - 2;
	if ((x > 1) && (x < 2)) return 2 - x;
	return 0;
}
} // namespace

struct mtfo_am {
	int kind, resx, resy, n;   /* n = patch_size = npix * C rows (ImageBase: patch_size = n_pix * n_channels) */
	int npix, C = 1;
	double grad_eps, likelihood_alpha;
	const float *img; int h, w;
	double norm_mult, norm_add;
	bool init_pix_vals, init_pix_grad, init_sim, init_grad, init_hess, init_pix_hess = false;
	unsigned int frame_count = 0;   /* ImageBase.h:169 */
	double hess_eps = 1.0; /* HESS_EPS AM/include/mtf/AM/ImageBase.h:9, Config/mtf.cfg:16 */
	vecd I0, It, dI0_dx, dIt_dx, df_dI0, df_dIt, d2I0_dx2, d2It_dx2;
	double f;
	/* NCC */
	double I0_mean, It_mean, a, b, c, bc, b2c;
	vecd I0_cntr, It_cntr, I0_cntr_c, It_cntr_b, df_dI0_ncntr, df_dIt_ncntr;
	/* MI */
	int n_bins; double pre_seed; int pou;
	double hist_pre_seed, hist_norm_mult, max_similarity;
	std::vector<int> std_ids, init_ids, curr_ids; /* [.. ,2] row-major */
	vecd init_hist, curr_hist, init_hist_log, curr_hist_log;
	vecd init_hist_mat, curr_hist_mat, init_hist_grad, curr_hist_grad; /* n_bins x N col-major */
	vecd init_hist_hess, curr_hist_hess;
	vecd joint_hist, joint_hist_log, init_grad_factor, curr_grad_factor; /* (r,c) -> r*n_bins+c */
	vecd self_joint_hist, self_joint_hist_log, self_grad_factor;

	mtfo_am(int _kind, int _resx, int _resy, double _grad_eps, double _alpha,
		int _n_bins, double _pre_seed, int _pou) : kind(_kind), resx(_resx), resy(_resy),
		n(_resx * _resy), npix(_resx * _resy), grad_eps(_grad_eps), likelihood_alpha(_alpha), img(nullptr), h(0), w(0),
		norm_mult(1), norm_add(0), init_pix_vals(false), init_pix_grad(false), init_sim(false),
		init_grad(false), init_hess(false), f(0), n_bins(_n_bins), pre_seed(_pre_seed), pou(_pou) {
		if (kind == MTFO_AM_MI) {
			/* MI ctor AM/src/MI.cc:80-122 */
			double norm_pix_min = 0, norm_pix_max = n_bins - 1;
			if (pou) { norm_pix_min = 1; norm_pix_max = n_bins - 2; }
			norm_mult = (norm_pix_max - norm_pix_min) / (255.0 - 0.0 + 1);
			norm_add = norm_pix_min;
			hist_pre_seed = n_bins * pre_seed;
			hist_norm_mult = 1.0 / (static_cast<double>(n) + hist_pre_seed * n_bins);
			std_ids.resize(2 * n_bins);
			for (int i = 0; i < n_bins; ++i) {
				std_ids[2 * i] = std::max(0, i - 1);
				std_ids[2 * i + 1] = std::min(n_bins - 1, i + 2);
			}
		}
	}

	/* MCSSD / MCNCC / MCMI: the same classes constructed with n_channels = 3 (AM/src/MCSSD.cc, MCNCC.cc, MCMI.cc);
	 * only the sampling differs (ImageBase.cc switches on MTF_32FC3 to the utils::mc:: functions) */
	void set_channels(int c) {
		C = c; n = npix * C;
		if (kind == MTFO_AM_MI) hist_norm_mult = 1.0 / (static_cast<double>(n) + hist_pre_seed * n_bins);   /* MI.cc:104 (patch_size) */
	}
	void s_vals(double *out, const double *pts) const {
		if (C == 1) pix_vals(out, img, h, w, pts, npix, norm_mult, norm_add);
		else pix_vals_mc(out, img, h, w, C, pts, npix, norm_mult, norm_add);
	}
	void s_grad(double *out, const double *pts) const {
		if (C == 1) img_grad(out, img, h, w, pts, grad_eps, npix, norm_mult);
		else img_grad_mc(out, img, h, w, C, pts, grad_eps, npix, norm_mult);
	}
	void s_wgrad(double *out, const double *gp) const {
		if (C == 1) warped_img_grad(out, img, h, w, gp, grad_eps, npix, norm_mult);
		else warped_img_grad_mc(out, img, h, w, C, gp, grad_eps, npix, norm_mult);
	}
	void s_hess(double *out, const double *pts) const {
		if (C == 1) img_hess(out, img, h, w, pts, hess_eps, npix, norm_mult);
		else img_hess_mc(out, img, h, w, C, pts, hess_eps, npix, norm_mult);
	}
	void s_whess(double *out, const double *pts, const double *hp) const {
		if (C == 1) warped_img_hess(out, img, h, w, pts, hp, hess_eps, npix, norm_mult);
		else warped_img_hess_mc(out, img, h, w, C, pts, hp, hess_eps, npix, norm_mult);
	}
	/* ImageBase::initializePixVals AM/src/ImageBase.cc:62-99 (MI: AM/src/MI.cc:124-158) */
	void initialize_pix_vals(const double *pts) {
		if (!init_pix_vals) { I0.resize(n); It.resize(n); }
		++frame_count;   /* ImageBase.cc:74 */
		s_vals(I0.data(), pts);
		if (!init_pix_vals) { It = I0; init_pix_vals = true; }
	}
	/* ImageBase::updatePixVals AM/src/ImageBase.cc:268-290 */
	void update_pix_vals(const double *pts) { s_vals(It.data(), pts); }
	/* SSD::updateModel AM/src/SSD.cc:49-75 / NCC::updateModel AM/src/NCC.cc:539-566 with utils::getWeightedPixVals
	 * Utilities/src/imgUtils.cc:506-523 (MTF_32FC1 branch: the pixel normalisation is applied in the running average only),
	 * then AppearanceModel::reinitialize AppearanceModel.h:119-123.  Returns -1 where the reference throws (MI). */
	int update_model(const double *pts, double learning_rate) {
		if (kind == MTFO_AM_MI || C > 1) return -1;
		++frame_count;
		const bool running = learning_rate < 0 || learning_rate > 1;
		for (int i = 0; i < n; ++i) {
			const double v = pix_val(img, h, w, pts[2 * i], pts[2 * i + 1]);
			if (running) I0[i] += (norm_mult * v + norm_add - I0[i]) / frame_count;
			else I0[i] = learning_rate * v + (1 - learning_rate) * I0[i];
		}
		if (init_sim) initialize_similarity();
		if (init_grad) initialize_grad();
		return 0;
	}
	/* ImageBase::initializePixGrad(PtsT) AM/src/ImageBase.cc:101-132 */
	void initialize_pix_grad_pts(const double *pts) {
		if (!init_pix_grad) { dI0_dx.resize(2 * n); dIt_dx.resize(2 * n); }
		s_grad(dI0_dx.data(), pts);
		if (!init_pix_grad) { dIt_dx = dI0_dx; init_pix_grad = true; }
	}
	/* ImageBase::initializePixGrad(GradPtsT) AM/src/ImageBase.cc:134-172 */
	void initialize_pix_grad_warped(const double *gp) {
		if (!init_pix_grad) { dI0_dx.resize(2 * n); dIt_dx.resize(2 * n); }
		s_wgrad(dI0_dx.data(), gp);
		if (!init_pix_grad) { dIt_dx = dI0_dx; init_pix_grad = true; }
	}
	/* ImageBase::updatePixGrad(PtsT) :292-314 ; (GradPtsT) :340-362 */
	void update_pix_grad_pts(const double *pts) { s_grad(dIt_dx.data(), pts); }
	void update_pix_grad_warped(const double *gp) { s_wgrad(dIt_dx.data(), gp); }

	/* ImageBase::initializePixHess(PtsT) AM/src/ImageBase.cc:208-240 ; (PtsT, HessPtsT) :174-206 ;
	 * updatePixHess :316-338 and :364-386 */
	void initialize_pix_hess_pts(const double *pts) {
		if (!init_pix_hess) { d2I0_dx2.resize(4 * static_cast<size_t>(n)); d2It_dx2.resize(4 * static_cast<size_t>(n)); }
		s_hess(d2I0_dx2.data(), pts);
		if (!init_pix_hess) { d2It_dx2 = d2I0_dx2; init_pix_hess = true; }
	}
	void initialize_pix_hess_warped(const double *pts, const double *hp) {
		if (!init_pix_hess) { d2I0_dx2.resize(4 * static_cast<size_t>(n)); d2It_dx2.resize(4 * static_cast<size_t>(n)); }
		s_whess(d2I0_dx2.data(), pts, hp);
		if (!init_pix_hess) { d2It_dx2 = d2I0_dx2; init_pix_hess = true; }
	}
	void update_pix_hess_pts(const double *pts) { s_hess(d2It_dx2.data(), pts); }
	void update_pix_hess_warped(const double *pts, const double *hp) {
		s_whess(d2It_dx2.data(), pts, hp);
	}

	/* ---------------- similarity ---------------- */
	void initialize_similarity() {
		switch (kind) {
		case MTFO_AM_SSD: /* SSDBase::initializeSimilarity AM/src/SSDBase.cc:29-45 */
			if (init_sim) return;
			df_dI0.assign(n, 0.0); f = 0; init_sim = true;
			break;
		case MTFO_AM_NCC: /* NCC::initializeSimilarity AM/src/NCC.cc:55-95 */
			if (!init_sim) { I0_cntr.resize(n); It_cntr.resize(n); }
			I0_mean = mean(I0);
			for (int i = 0; i < n; ++i) I0_cntr[i] = I0[i] - I0_mean;
			c = norm(I0_cntr);
			if (!init_sim) { f = 1; It_mean = I0_mean; It_cntr = I0_cntr; b = c; init_sim = true; }
			break;
		case MTFO_AM_MI: mi_initialize_similarity(); break;
		}
	}
	void initialize_grad() {
		switch (kind) {
		case MTFO_AM_SSD: /* SSDBase::initializeGrad AM/src/SSDBase.cc:47-63 */
			if (init_grad) return;
			df_dIt = df_dI0; init_grad = true;
			break;
		case MTFO_AM_NCC: /* NCC::initializeGrad AM/src/NCC.cc:97-122 */
			if (!init_grad) {
				df_dIt.assign(n, 0.0); df_dI0.assign(n, 0.0);
				df_dI0_ncntr.assign(n, 0.0); df_dIt_ncntr.assign(n, 0.0);
				I0_cntr_c.resize(n); It_cntr_b.resize(n);
			}
			for (int i = 0; i < n; ++i) I0_cntr_c[i] = I0_cntr[i] / c;
			if (!init_grad) { It_cntr_b = I0_cntr_c; init_grad = true; }
			break;
		case MTFO_AM_MI: mi_initialize_grad(); break;
		}
	}
	void initialize_hess() {
		if (kind == MTFO_AM_MI) mi_initialize_hess();
		init_hess = true;
	}
	void update_similarity(bool prereq_only) {
		switch (kind) {
		case MTFO_AM_SSD: { /* SSDBase::updateSimilarity AM/src/SSDBase.cc:75-96 ; I_diff aliases df_dI0 (:34) */
			for (int i = 0; i < n; ++i) df_dI0[i] = It[i] - I0[i];
			if (prereq_only) return;
			double s = 0;
			for (int i = 0; i < n; ++i) s += df_dI0[i] * df_dI0[i];
			f = -s / 2;
			break;
		}
		case MTFO_AM_NCC: { /* NCC::updateSimilarity AM/src/NCC.cc:124-161 */
			It_mean = mean(It);
			for (int i = 0; i < n; ++i) It_cntr[i] = It[i] - It_mean;
			double s = 0;
			for (int i = 0; i < n; ++i) s += I0_cntr[i] * It_cntr[i];
			a = s; b = norm(It_cntr);
			bc = b * c; b2c = bc * b; f = a / bc;
			break;
		}
		case MTFO_AM_MI: mi_update_similarity(prereq_only); break;
		}
	}
	void update_curr_grad() {
		switch (kind) {
		case MTFO_AM_SSD: /* SSDBase::updateCurrGrad AM/src/SSDBase.cc:115-121 */
			for (int i = 0; i < n; ++i) df_dIt[i] = -df_dI0[i];
			break;
		case MTFO_AM_NCC: { /* NCC::updateCurrGrad AM/src/NCC.cc:196-234 */
			double m = 0;
			for (int i = 0; i < n; ++i) {
				It_cntr_b[i] = It_cntr[i] / b;
				df_dIt_ncntr[i] = (I0_cntr_c[i] - f * It_cntr_b[i]) / b;
				m += df_dIt_ncntr[i];
			}
			m /= n;
			for (int i = 0; i < n; ++i) df_dIt[i] = df_dIt_ncntr[i] - m;
			break;
		}
		case MTFO_AM_MI: mi_update_curr_grad(); break;
		}
	}
	void update_init_grad() {
		switch (kind) {
		case MTFO_AM_SSD: break; /* SSDBase.h:67-72: nothing without an ILM */
		case MTFO_AM_NCC: { /* NCC::updateInitGrad AM/src/NCC.cc:163-194 */
			double m = 0;
			for (int i = 0; i < n; ++i) {
				It_cntr_b[i] = It_cntr[i] / b;
				df_dI0_ncntr[i] = (It_cntr_b[i] - f * I0_cntr_c[i]) / c;
				m += df_dI0_ncntr[i];
			}
			m /= n;
			for (int i = 0; i < n; ++i) df_dI0[i] = df_dI0_ncntr[i] - m;
			break;
		}
		case MTFO_AM_MI: mi_update_init_grad(); break;
		}
	}
	/* SSD::getLikelihood AM/include/mtf/AM/SSD.h:41-43 ; NCC::getLikelihood NCC.cc:50-53 ;
	 * MI::getLikelihood MI.cc:384-387 */
	double likelihood() const {
		if (kind == MTFO_AM_SSD) return std::exp(-likelihood_alpha * std::sqrt(-f / static_cast<double>(n)));
		double d = (1.0 / f) - 1;
		return std::exp(-likelihood_alpha * d * d);
	}

	/* ---------------- Jacobians: g = df_dI * J ---------------- */
	static void row_times_mat(double *g, const double *v, const double *J, int n, int S) {
		for (int s = 0; s < S; ++s) {
			double acc = 0;
			const double *col = J + static_cast<size_t>(s) * n;
			for (int i = 0; i < n; ++i) acc += v[i] * col[i];
			g[s] = acc;
		}
	}
	/* AppearanceModel::cmptInitJacobian AppearanceModel.h:146-149 ; SSDBase.cc:123-143 ; NCC.cc:236-250 */
	void cmpt_init_jacobian(double *g, const double *J0, int S) { row_times_mat(g, df_dI0.data(), J0, n, S); }
	/* cmptCurrJacobian AppearanceModel.h:150-153 ; SSDBase.cc:144-168 ; NCC.cc:252-266 */
	void cmpt_curr_jacobian(double *g, const double *Jt, int S) { row_times_mat(g, df_dIt.data(), Jt, n, S); }
	/* cmptDifferenceOfJacobians: SSD df_dIt*(J0+Jt) SSDBase.cc:169-191 ;
	 * generic (NCC.cc:268-280, AppearanceModel.h:161-164) df_dIt*Jt - df_dI0*J0 */
	void cmpt_difference_of_jacobians(double *g, const double *J0, const double *Jt, int S) {
		if (kind == MTFO_AM_SSD) {
			for (int s = 0; s < S; ++s) {
				double acc = 0;
				const double *c0 = J0 + static_cast<size_t>(s) * n, *ct = Jt + static_cast<size_t>(s) * n;
				for (int i = 0; i < n; ++i) acc += df_dIt[i] * (c0[i] + ct[i]);
				g[s] = acc;
			}
		} else {
			vecd g0(S), gt(S);
			row_times_mat(gt.data(), df_dIt.data(), Jt, n, S);
			row_times_mat(g0.data(), df_dI0.data(), J0, n, S);
			for (int s = 0; s < S; ++s) g[s] = gt[s] - g0[s];
		}
	}

	/* ---------------- Hessians ---------------- */
	static void neg_gram(double *H, const double *J, int n, int S) {
		for (int r = 0; r < S; ++r)
			for (int c2 = r; c2 < S; ++c2) {
				double acc = 0;
				const double *a = J + static_cast<size_t>(r) * n, *b2 = J + static_cast<size_t>(c2) * n;
				for (int i = 0; i < n; ++i) acc += a[i] * b2[i];
				H[c2 * S + r] = H[r * S + c2] = -acc;
			}
	}
	/* NCC helper: Jc = (J - colmean(J)) / b, u = Jc^T v */
	void ncc_centre(vecd &Jc, const double *J, int S) const {
		Jc.resize(static_cast<size_t>(n) * S);
		for (int s = 0; s < S; ++s) {
			const double *col = J + static_cast<size_t>(s) * n;
			double m = 0;
			for (int i = 0; i < n; ++i) m += col[i];
			m /= n;
			for (int i = 0; i < n; ++i) Jc[static_cast<size_t>(s) * n + i] = (col[i] - m) / b;
		}
	}
	static void mat_t_vec(double *u, const vecd &Jc, const vecd &v, int n, int S) {
		for (int s = 0; s < S; ++s) {
			double acc = 0;
			for (int i = 0; i < n; ++i) acc += Jc[static_cast<size_t>(s) * n + i] * v[i];
			u[s] = acc;
		}
	}
	void cmpt_init_hessian(double *H, const double *J0, int S) {
		switch (kind) {
		case MTFO_AM_SSD: neg_gram(H, J0, n, S); break; /* SSDBase.cc:251-267 */
		case MTFO_AM_NCC: { /* NCC::cmptInitHessian AM/src/NCC.cc:282-303 (divides by b, quirk kept) */
			vecd Jc; ncc_centre(Jc, J0, S);
			vecd G(S * S), ut(S), u0(S);
			neg_gram(G.data(), Jc.data(), n, S);
			mat_t_vec(ut.data(), Jc, It_cntr_b, n, S);
			mat_t_vec(u0.data(), Jc, I0_cntr_c, n, S);
			for (int r = 0; r < S; ++r)
				for (int c2 = 0; c2 < S; ++c2)
					H[c2 * S + r] = f * G[c2 * S + r] - ut[r] * u0[c2] - u0[r] * ut[c2] + 3 * u0[r] * u0[c2];
			break;
		}
		case MTFO_AM_MI: mi_cmpt_init_hessian(H, J0, S); break;
		}
	}
	void cmpt_curr_hessian(double *H, const double *Jt, int S) {
		switch (kind) {
		case MTFO_AM_SSD: neg_gram(H, Jt, n, S); break; /* SSDBase.cc:268-285 */
		case MTFO_AM_NCC: { /* NCC::cmptCurrHessian AM/src/NCC.cc:304-335 */
			vecd Jc; ncc_centre(Jc, Jt, S);
			vecd G(S * S), ut(S), u0(S);
			neg_gram(G.data(), Jc.data(), n, S);
			mat_t_vec(ut.data(), Jc, It_cntr_b, n, S);
			mat_t_vec(u0.data(), Jc, I0_cntr_c, n, S);
			for (int r = 0; r < S; ++r)
				for (int c2 = 0; c2 < S; ++c2)
					H[c2 * S + r] = f * G[c2 * S + r] - ut[r] * u0[c2] - u0[r] * ut[c2] + 3 * ut[r] * ut[c2];
			break;
		}
		case MTFO_AM_MI: mi_cmpt_curr_hessian(H, Jt, S); break;
		}
	}
	void cmpt_self_hessian(double *H, const double *Jt, int S) {
		switch (kind) {
		case MTFO_AM_SSD: neg_gram(H, Jt, n, S); break; /* SSDBase.h:91-94 */
		case MTFO_AM_NCC: { /* NCC::cmptSelfHessian AM/src/NCC.cc:337-389 (fast_hess = 0) */
			vecd Jc; ncc_centre(Jc, Jt, S);
			vecd G(S * S), ut(S);
			neg_gram(G.data(), Jc.data(), n, S);
			mat_t_vec(ut.data(), Jc, It_cntr_b, n, S);
			for (int r = 0; r < S; ++r)
				for (int c2 = 0; c2 < S; ++c2)
					H[c2 * S + r] = G[c2 * S + r] + ut[r] * ut[c2];
			break;
		}
		case MTFO_AM_MI: mi_cmpt_self_hessian(H, Jt, S); break;
		}
	}
	/* cmptSumOfHessians: SSDBase.cc:287-311 ; generic AppearanceModel.h:196-208 */
	void cmpt_sum_of_hessians(double *H, const double *J0, const double *Jt, int S) {
		vecd H0(S * S), Ht(S * S);
		cmpt_init_hessian(H0.data(), J0, S);
		cmpt_curr_hessian(Ht.data(), Jt, S);
		for (int i = 0; i < S * S; ++i) H[i] = H0[i] + Ht[i];
	}

	/* ---------------- second-order Hessians ---------------- */
	/* H += sum_p w[p] * d2I_dp2[:, p]  (d2 is S^2 x N, one column-major S x S block per pixel) */
	void add_weighted_pix_hess(double *H, const double *d2, const double *wt, int S) const {
		for (int p = 0; p < n; ++p) {
			const double *blk = d2 + static_cast<size_t>(p) * S * S;
			for (int k = 0; k < S * S; ++k) H[k] += blk[k] * wt[p];
		}
	}
	/* SSDBase::cmptInitHessian (second order) AM/src/SSDBase.cc:313-343 ; NCC AM/src/NCC.cc:391-400 ; MI AM/src/MI.cc:659-673 */
	int cmpt_init_hessian2(double *H, const double *J0, const double *d2I0, int S) {
		cmpt_init_hessian(H, J0, S);
		add_weighted_pix_hess(H, d2I0, df_dI0.data(), S);
		return 0;
	}
	/* SSDBase.cc:345-375 ; NCC.cc:401-410 ; MI.cc:680-694 */
	int cmpt_curr_hessian2(double *H, const double *Jt, const double *d2It, int S) {
		cmpt_curr_hessian(H, Jt, S);
		add_weighted_pix_hess(H, d2It, df_dIt.data(), S);
		return 0;
	}
	/* SSD: first order only (SSDBase.h:95-98) ; NCC: not overridden -> FunctonNotImplemented
	 * (AppearanceModel.h:188-191) ; MI::cmptSelfHessian (second order) AM/src/MI.cc:696-733 */
	int cmpt_self_hessian2(double *H, const double *Jt, const double *d2It, int S) {
		switch (kind) {
		case MTFO_AM_SSD: cmpt_self_hessian(H, Jt, S); return 0;
		case MTFO_AM_NCC: return -2;
		default: {
			mi_cmpt_self_hessian(H, Jt, S);
			vecd wt(n);
			for (int p = 0; p < n; ++p) {
				double grad_term = 0;
				for (int r = curr_ids[2 * p]; r <= curr_ids[2 * p + 1]; ++r) {
					double inner = 0;
					for (int t = curr_ids[2 * p]; t <= curr_ids[2 * p + 1]; ++t)
						inner += HMc(curr_hist_mat, t, p) * self_grad_factor[lin(r, t)];
					grad_term += HMc(curr_hist_grad, r, p) * inner;
				}
				wt[p] = grad_term;
			}
			add_weighted_pix_hess(H, d2It, wt.data(), S);
			return 0;
		}
		}
	}
	/* SSDBase::cmptSumOfHessians (second order) AM/src/SSDBase.cc:377-415 -- weights BOTH pixel Hessians by df_dI0 ;
	 * NCC / MI: the generic AppearanceModel.h:209-219 (init2 + curr2) */
	int cmpt_sum_of_hessians2(double *H, const double *J0, const double *Jt, const double *d2I0, const double *d2It, int S) {
		if (kind == MTFO_AM_SSD) {
			cmpt_sum_of_hessians(H, J0, Jt, S);
			add_weighted_pix_hess(H, d2I0, df_dI0.data(), S);
			add_weighted_pix_hess(H, d2It, df_dI0.data(), S);
			return 0;
		}
		vecd H0(S * S), Ht(S * S);
		cmpt_init_hessian2(H0.data(), J0, d2I0, S);
		cmpt_curr_hessian2(Ht.data(), Jt, d2It, S);
		for (int i = 0; i < S * S; ++i) H[i] = H0[i] + Ht[i];
		return 0;
	}

	/* ---------------- helpers ---------------- */
	static double mean(const vecd &v) { double s = 0; for (double x : v) s += x; return s / v.size(); }
	static double norm(const vecd &v) { double s = 0; for (double x : v) s += x * x; return std::sqrt(s); }

	/* ================= MI ================= */
	inline double &HM(vecd &m, int bin, int pix) { return m[static_cast<size_t>(pix) * n_bins + bin]; }
	inline double HMc(const vecd &m, int bin, int pix) const { return m[static_cast<size_t>(pix) * n_bins + bin]; }
	inline int lin(int r, int c2) const { return r * n_bins + c2; }

	/* MI::initializeSimilarity AM/src/MI.cc:207-287 */
	void mi_initialize_similarity() {
		size_t nb = n_bins;
		if (!init_sim) {
			init_hist.resize(nb); curr_hist.resize(nb); init_hist_log.resize(nb); curr_hist_log.resize(nb);
			init_hist_mat.resize(nb * n); curr_hist_mat.resize(nb * n);
			init_hist_grad.resize(nb * n); curr_hist_grad.resize(nb * n);
			joint_hist.resize(nb * nb); joint_hist_log.resize(nb * nb);
			init_ids.resize(2 * n); curr_ids.resize(2 * n);
		}
		std::fill(init_hist.begin(), init_hist.end(), hist_pre_seed);
		std::fill(init_hist_mat.begin(), init_hist_mat.end(), 0.0);
		std::fill(init_hist_grad.begin(), init_hist_grad.end(), 0.0);
		for (int p = 0; p < n; ++p) {
			int fl = static_cast<int>(I0[p]);
			init_ids[2 * p] = std_ids[2 * fl]; init_ids[2 * p + 1] = std_ids[2 * fl + 1];
			double diff = init_ids[2 * p] - I0[p];
			for (int id = init_ids[2 * p]; id <= init_ids[2 * p + 1]; ++id) {
				bspl3_with_grad(HM(init_hist_mat, id, p), HM(init_hist_grad, id, p), diff);
				HM(init_hist_grad, id, p) *= -hist_norm_mult;
				init_hist[id] += HM(init_hist_mat, id, p);
				++diff;
			}
		}
		for (int i = 0; i < n_bins; ++i) { init_hist[i] *= hist_norm_mult; init_hist_log[i] = std::log(init_hist[i]); }
		if (!init_sim) {
			std::fill(joint_hist.begin(), joint_hist.end(), pre_seed);
			for (int p = 0; p < n; ++p)
				for (int i1 = init_ids[2 * p]; i1 <= init_ids[2 * p + 1]; ++i1)
					for (int i2 = init_ids[2 * p]; i2 <= init_ids[2 * p + 1]; ++i2)
						joint_hist[lin(i1, i2)] += HMc(init_hist_mat, i1, p) * HMc(init_hist_mat, i2, p);
			for (size_t i = 0; i < nb * nb; ++i) { joint_hist[i] *= hist_norm_mult; joint_hist_log[i] = std::log(joint_hist[i]); }
			f = 0;
			for (int ci = 0; ci < n_bins; ++ci)
				for (int ii = 0; ii < n_bins; ++ii)
					f += joint_hist[lin(ci, ii)] * (joint_hist_log[lin(ci, ii)] - init_hist_log[ci] - init_hist_log[ii]);
			max_similarity = f;
			curr_ids = init_ids; curr_hist = init_hist; curr_hist_mat = init_hist_mat;
			curr_hist_log = init_hist_log; curr_hist_grad = init_hist_grad;
			init_sim = true;
		}
	}
	/* MI::initializeGrad AM/src/MI.cc:299-332 (the n_bins^2 x N joint-gradient matrices
	 * of :301-302 are never stored here: each entry is a product of two stored factors) */
	void mi_initialize_grad() {
		if (init_grad) return;
		size_t nb = n_bins;
		init_grad_factor.resize(nb * nb); curr_grad_factor.resize(nb * nb);
		df_dIt.resize(n); df_dI0.assign(n, 0.0);
		for (int ci = 0; ci < n_bins; ++ci)
			for (int ii = 0; ii < n_bins; ++ii)
				init_grad_factor[lin(ci, ii)] = 1 + joint_hist_log[lin(ci, ii)] - init_hist_log[ci];
		for (int p = 0; p < n; ++p)
			for (int ci = init_ids[2 * p]; ci <= init_ids[2 * p + 1]; ++ci)
				for (int ii = init_ids[2 * p]; ii <= init_ids[2 * p + 1]; ++ii)
					df_dI0[p] += HMc(init_hist_grad, ci, p) * HMc(init_hist_mat, ii, p) * init_grad_factor[lin(ci, ii)];
		curr_grad_factor = init_grad_factor;
		df_dIt = df_dI0;
		init_grad = true;
	}
	/* MI::updateSimilarity AM/src/MI.cc:346-382 */
	void mi_update_similarity(bool prereq_only) {
		size_t nb = n_bins;
		std::fill(curr_hist.begin(), curr_hist.end(), hist_pre_seed);
		std::fill(joint_hist.begin(), joint_hist.end(), pre_seed);
		std::fill(curr_hist_mat.begin(), curr_hist_mat.end(), 0.0);
		std::fill(curr_hist_grad.begin(), curr_hist_grad.end(), 0.0);
		for (int p = 0; p < n; ++p) {
			int fl = static_cast<int>(It[p]);
			curr_ids[2 * p] = std_ids[2 * fl]; curr_ids[2 * p + 1] = std_ids[2 * fl + 1];
			double diff = curr_ids[2 * p] - It[p];
			for (int ci = curr_ids[2 * p]; ci <= curr_ids[2 * p + 1]; ++ci) {
				bspl3_with_grad(HM(curr_hist_mat, ci, p), HM(curr_hist_grad, ci, p), diff);
				++diff;
				HM(curr_hist_grad, ci, p) *= -hist_norm_mult;
				curr_hist[ci] += HMc(curr_hist_mat, ci, p);
				for (int ii = init_ids[2 * p]; ii <= init_ids[2 * p + 1]; ++ii)
					joint_hist[lin(ci, ii)] += HMc(curr_hist_mat, ci, p) * HMc(init_hist_mat, ii, p);
			}
		}
		for (int i = 0; i < n_bins; ++i) { curr_hist[i] *= hist_norm_mult; curr_hist_log[i] = std::log(curr_hist[i]); }
		for (size_t i = 0; i < nb * nb; ++i) { joint_hist[i] *= hist_norm_mult; joint_hist_log[i] = std::log(joint_hist[i]); }
		if (prereq_only) return;
		f = 0;
		for (int ci = 0; ci < n_bins; ++ci)
			for (int ii = 0; ii < n_bins; ++ii)
				f += joint_hist[lin(ci, ii)] * (joint_hist_log[lin(ci, ii)] - curr_hist_log[ci] - init_hist_log[ii]);
	}
	/* MI::updateInitGrad AM/src/MI.cc:398-416 (init_grad_factor is indexed (init, curr) here) */
	void mi_update_init_grad() {
		for (int ii = 0; ii < n_bins; ++ii)
			for (int ci = 0; ci < n_bins; ++ci)
				init_grad_factor[lin(ii, ci)] = 1 + joint_hist_log[lin(ci, ii)] - init_hist_log[ii];
		for (int p = 0; p < n; ++p) {
			double acc = 0;
			for (int ii = init_ids[2 * p]; ii <= init_ids[2 * p + 1]; ++ii)
				for (int ci = curr_ids[2 * p]; ci <= curr_ids[2 * p + 1]; ++ci)
					acc += HMc(init_hist_grad, ii, p) * HMc(curr_hist_mat, ci, p) * init_grad_factor[lin(ii, ci)];
			df_dI0[p] = acc;
		}
	}
	/* MI::updateCurrGrad AM/src/MI.cc:426-442 */
	void mi_update_curr_grad() {
		for (int ci = 0; ci < n_bins; ++ci)
			for (int ii = 0; ii < n_bins; ++ii)
				curr_grad_factor[lin(ci, ii)] = 1 + joint_hist_log[lin(ci, ii)] - curr_hist_log[ci];
		for (int p = 0; p < n; ++p) {
			double acc = 0;
			for (int ci = curr_ids[2 * p]; ci <= curr_ids[2 * p + 1]; ++ci)
				for (int ii = init_ids[2 * p]; ii <= init_ids[2 * p + 1]; ++ii)
					acc += HMc(curr_hist_grad, ci, p) * HMc(init_hist_mat, ii, p) * curr_grad_factor[lin(ci, ii)];
			df_dIt[p] = acc;
		}
	}
	/* MI::initializeHess AM/src/MI.cc:443-459 ; utils::getBSplHistHess Utilities/src/histUtils.cc:234-256 */
	void mi_initialize_hess() {
		size_t nb = n_bins;
		if (!init_hess) {
			init_hist_hess.resize(nb * n); curr_hist_hess.resize(nb * n);
			self_joint_hist.resize(nb * nb); self_joint_hist_log.resize(nb * nb); self_grad_factor.resize(nb * nb);
		}
		std::fill(init_hist_hess.begin(), init_hist_hess.end(), 0.0);
		for (int p = 0; p < n; ++p) {
			double diff = init_ids[2 * p] - I0[p];
			for (int id = init_ids[2 * p]; id <= init_ids[2 * p + 1]; ++id)
				HM(init_hist_hess, id, p) = hist_norm_mult * bspl3_hess(diff++);
		}
		if (!init_hess) curr_hist_hess = init_hist_hess;
	}
	/* adds  sum_k row_k^T row_k * factor_k  for the n_bins^2 x S matrix Q */
	void mi_add_rank1_terms(double *H, const vecd &Q, const vecd &factor, int S) const {
		for (int k = 0; k < n_bins * n_bins; ++k) {
			const double *row = &Q[static_cast<size_t>(k) * S];
			for (int r = 0; r < S; ++r)
				for (int c2 = 0; c2 < S; ++c2)
					H[c2 * S + r] += row[r] * row[c2] * factor[k];
		}
	}
	static void add_outer(double *H, const double *J, int n, int S, int p, double wgt) {
		for (int r = 0; r < S; ++r) {
			double jr = J[static_cast<size_t>(r) * n + p] * wgt;
			for (int c2 = 0; c2 < S; ++c2) H[c2 * S + r] += jr * J[static_cast<size_t>(c2) * n + p];
		}
	}
	/* MI::cmptInitHessian AM/src/MI.cc:461-513 */
	void mi_cmpt_init_hessian(double *H, const double *J0, int S) {
		vecd Q(static_cast<size_t>(n_bins) * n_bins * S, 0.0), factor(n_bins * n_bins);
		std::fill(H, H + S * S, 0.0);
		for (int p = 0; p < n; ++p) {
			double hess_term = 0;
			for (int ii = init_ids[2 * p]; ii <= init_ids[2 * p + 1]; ++ii) {
				double inner = 0;
				for (int ci = curr_ids[2 * p]; ci <= curr_ids[2 * p + 1]; ++ci) {
					double gr = HMc(init_hist_grad, ii, p) * HMc(curr_hist_mat, ci, p); /* init_joint_hist_grad(lin(ii,ci),p) MI.cc:411 */
					double *row = &Q[static_cast<size_t>(lin(ci, ii)) * S];
					for (int s = 0; s < S; ++s) row[s] += gr * J0[static_cast<size_t>(s) * n + p];
					inner += HMc(curr_hist_mat, ci, p) * init_grad_factor[lin(ii, ci)];
				}
				hess_term += HMc(init_hist_hess, ii, p) * inner;
			}
			add_outer(H, J0, n, S, p, hess_term);
		}
		for (int ci = 0; ci < n_bins; ++ci)
			for (int ii = 0; ii < n_bins; ++ii)
				factor[lin(ci, ii)] = (1.0 / joint_hist[lin(ci, ii)]) - (1.0 / init_hist[ii]);
		mi_add_rank1_terms(H, Q, factor, S);
	}
	/* MI::cmptCurrHessian AM/src/MI.cc:603-637 */
	void mi_cmpt_curr_hessian(double *H, const double *Jt, int S) {
		vecd Q(static_cast<size_t>(n_bins) * n_bins * S, 0.0), factor(n_bins * n_bins);
		std::fill(H, H + S * S, 0.0);
		for (int p = 0; p < n; ++p) {
			double diff = curr_ids[2 * p] - It[p];
			double hess_term = 0;
			for (int ci = curr_ids[2 * p]; ci <= curr_ids[2 * p + 1]; ++ci) {
				HM(curr_hist_hess, ci, p) = hist_norm_mult * bspl3_hess(diff);
				++diff;
				double inner = 0;
				for (int ii = init_ids[2 * p]; ii <= init_ids[2 * p + 1]; ++ii) {
					double gr = HMc(curr_hist_grad, ci, p) * HMc(init_hist_mat, ii, p); /* curr_joint_hist_grad MI.cc:437 */
					double *row = &Q[static_cast<size_t>(lin(ci, ii)) * S];
					for (int s = 0; s < S; ++s) row[s] += gr * Jt[static_cast<size_t>(s) * n + p];
					inner += HMc(init_hist_mat, ii, p) * curr_grad_factor[lin(ci, ii)];
				}
				hess_term += HMc(curr_hist_hess, ci, p) * inner;
			}
			add_outer(H, Jt, n, S, p, hess_term);
		}
		for (int ci = 0; ci < n_bins; ++ci)
			for (int ii = 0; ii < n_bins; ++ii)
				factor[lin(ci, ii)] = (1.0 / joint_hist[lin(ci, ii)]) - (1.0 / curr_hist[ci]);
		mi_add_rank1_terms(H, Q, factor, S);
	}
	/* MI::cmptSelfHist AM/src/MI.cc:639-659 */
	void mi_cmpt_self_hist() {
		size_t nb = n_bins;
		std::fill(self_joint_hist.begin(), self_joint_hist.end(), pre_seed);
		for (int p = 0; p < n; ++p)
			for (int i1 = curr_ids[2 * p]; i1 <= curr_ids[2 * p + 1]; ++i1)
				for (int i2 = curr_ids[2 * p]; i2 <= curr_ids[2 * p + 1]; ++i2)
					self_joint_hist[lin(i1, i2)] += HMc(curr_hist_mat, i1, p) * HMc(curr_hist_mat, i2, p);
		for (size_t i = 0; i < nb * nb; ++i) { self_joint_hist[i] *= hist_norm_mult; self_joint_hist_log[i] = std::log(self_joint_hist[i]); }
		for (int ci = 0; ci < n_bins; ++ci)
			for (int ii = 0; ii < n_bins; ++ii)
				self_grad_factor[lin(ci, ii)] = 1 + self_joint_hist_log[lin(ci, ii)] - curr_hist_log[ci];
	}
	/* MI::cmptSelfHessian AM/src/MI.cc:515-601 -- the value the reference returns is the
	 * second pass (:565-590, `self_hessian`); the first pass (:524-562) is discarded there. */
	void mi_cmpt_self_hessian(double *H, const double *Jt, int S) {
		mi_cmpt_self_hist();
		vecd Q(static_cast<size_t>(n_bins) * n_bins * S, 0.0), factor(n_bins * n_bins);
		std::fill(H, H + S * S, 0.0);
		for (int p = 0; p < n; ++p) {
			double diff = curr_ids[2 * p] - It[p];
			double hess_term = 0;
			for (int ci = curr_ids[2 * p]; ci <= curr_ids[2 * p + 1]; ++ci) {
				HM(curr_hist_hess, ci, p) = hist_norm_mult * bspl3_hess(diff);
				++diff;
				double inner = 0;
				for (int ii = curr_ids[2 * p]; ii <= curr_ids[2 * p + 1]; ++ii) {
					double gr = HMc(curr_hist_grad, ci, p) * HMc(curr_hist_mat, ii, p);
					double *row = &Q[static_cast<size_t>(lin(ci, ii)) * S];
					for (int s = 0; s < S; ++s) row[s] += gr * Jt[static_cast<size_t>(s) * n + p];
					inner += HMc(curr_hist_mat, ii, p) * self_grad_factor[lin(ci, ii)];
				}
				hess_term += HMc(curr_hist_hess, ci, p) * inner;
			}
			add_outer(H, Jt, n, S, p, hess_term);
		}
		for (int ci = 0; ci < n_bins; ++ci)
			for (int ii = 0; ii < n_bins; ++ii)
				factor[lin(ci, ii)] = (1.0 / self_joint_hist[lin(ci, ii)]) - (1.0 / curr_hist[ci]);
		mi_add_rank1_terms(H, Q, factor, S);
	}
};

/* ===================================================================== */
/* L3: search methods (non-templated "NT" variants, virtual-dispatch path) */
/* ===================================================================== */

struct mtfo_tracker {
	int sm; mtfo_am *am; mtfo_ssm *ssm; mtfo_sm_params p;
	int S, N;
	vecd J0, Jt, Jmean, g, H, H0, dp, inv_dp, prev_corners;
	vecd D0, Dt, Dmean; /* init / curr / mean pixel Hessians, S^2 x N (sec_ord_hess) */
	int status = 0;     /* -2 once an AM/SSM call the reference does not implement was needed */
	vecd trace; int rec_len, n_rec;

	mtfo_tracker(int _sm, mtfo_am *_am, mtfo_ssm *_ssm, const mtfo_sm_params &_p) :
		sm(_sm), am(_am), ssm(_ssm), p(_p), S(_ssm->S), N(_am->n) {
		J0.resize(static_cast<size_t>(N) * S); Jt.resize(static_cast<size_t>(N) * S);
		g.assign(S, 0.0); H.assign(S * S, 0.0); H0.assign(S * S, 0.0);
		dp.assign(S, 0.0); inv_dp.assign(S, 0.0); prev_corners.resize(8);
		rec_len = 1 + S + S * S + S + 8; n_rec = 0;
		if (p.sec_ord_hess) { D0.assign(static_cast<size_t>(N) * S * S, 0.0); Dt = D0; }
	}

	void pix_jacobian_init(double *J) {
		/* ESM::initializePixJacobian NT/ESM.cc:379-388 ; FCLK NT/FCLK.cc:113-118,128-133 ; ICLK NT/ICLK.cc:78-95 */
		if (p.chained_warp) {
			am->initialize_pix_grad_pts(ssm->curr_pts.data());
			ssm->warped_pix_jacobian(J, am->dI0_dx.data());
		} else {
			ssm->update_grad_pts(am->grad_eps);
			am->initialize_pix_grad_warped(ssm->grad_pts.data());
			ssm->init_pix_jacobian(J, am->dI0_dx.data());
		}
	}
	void pix_jacobian_update(double *J) {
		/* ESM::updatePixJacobian NT/ESM.cc:390-408 ; FCLK NT/FCLK.cc:222-236 */
		if (p.chained_warp) {
			am->update_pix_grad_pts(ssm->curr_pts.data());
			ssm->warped_pix_jacobian(J, am->dIt_dx.data());
		} else {
			ssm->update_grad_pts(am->grad_eps);
			am->update_pix_grad_warped(ssm->grad_pts.data());
			ssm->init_pix_jacobian(J, am->dIt_dx.data());
		}
	}

	/* ESM::initializePixHessian NT/ESM.cc:406-416 (FCLK NT/FCLK.cc:121-142, ICLK NT/ICLK.cc:96-113 inline the same) */
	void pix_hess_init() {
		if (p.chained_warp) am->initialize_pix_hess_pts(ssm->curr_pts.data());
		else { ssm->update_hess_pts(am->hess_eps); am->initialize_pix_hess_warped(ssm->curr_pts.data(), ssm->hess_pts.data()); }
	}
	void pix_hessian_from_init(double *D) {
		if (p.chained_warp) ssm->warped_pix_hessian(D, am->d2I0_dx2.data(), am->dI0_dx.data());
		else ssm->init_pix_hessian(D, am->d2I0_dx2.data(), am->dI0_dx.data());
	}
	/* ESM::updatePixHessian NT/ESM.cc:418-432 ; FCLK NT/FCLK.cc:243-257 ; ICLK NT/ICLK.cc:223-237 */
	void pix_hessian_update(double *D) {
		if (p.chained_warp) {
			am->update_pix_hess_pts(ssm->curr_pts.data());
			ssm->warped_pix_hessian(D, am->d2It_dx2.data(), am->dIt_dx.data());
		} else {
			ssm->update_hess_pts(am->hess_eps);
			am->update_pix_hess_warped(ssm->curr_pts.data(), ssm->hess_pts.data());
			ssm->init_pix_hessian(D, am->d2It_dx2.data(), am->dIt_dx.data());
		}
	}
	void self_hessian(double *Hout, const double *J, const double *D) {
		if (p.sec_ord_hess) { if (am->cmpt_self_hessian2(Hout, J, D, S)) status = -2; }
		else am->cmpt_self_hessian(Hout, J, S);
	}

	void initialize(const double *corners) {
		am->init_pix_vals = am->init_pix_grad = am->init_sim = am->init_grad = am->init_hess = am->init_pix_hess = false;
		status = 0;
		ssm->set_corners(corners);
		am->initialize_pix_vals(ssm->curr_pts.data());
		switch (sm) {
		case MTFO_SM_ESM: /* nt::ESM::initialize SM/src/NT/ESM.cc:110-146 */
			pix_jacobian_init(J0.data());
			if (p.sec_ord_hess) { pix_hess_init(); pix_hessian_from_init(D0.data()); }
			am->initialize_similarity(); am->initialize_grad(); am->initialize_hess();
			if (p.hess_type == 0 /*InitialSelf*/ || p.hess_type == 2 /*SumOfSelf*/) {
				self_hessian(H.data(), J0.data(), D0.data());
				H0 = H;
			}
			break;
		case MTFO_SM_FCLK: /* nt::FCLK::initialize SM/src/NT/FCLK.cc:102-169 */
			am->initialize_similarity(); am->initialize_grad(); am->initialize_hess();
			if (p.chained_warp) am->initialize_pix_grad_pts(ssm->curr_pts.data());
			else { ssm->update_grad_pts(am->grad_eps); am->initialize_pix_grad_warped(ssm->grad_pts.data()); }
			if (p.sec_ord_hess) pix_hess_init();
			if (p.hess_type == 0 /*InitialSelf*/) {
				if (p.chained_warp) ssm->warped_pix_jacobian(J0.data(), am->dI0_dx.data());
				else ssm->init_pix_jacobian(J0.data(), am->dI0_dx.data());
				if (p.sec_ord_hess) pix_hessian_from_init(D0.data());
				self_hessian(H.data(), J0.data(), D0.data());
				if (p.leven_marq) H0 = H;
			}
			break;
		case MTFO_SM_ICLK: /* nt::ICLK::initialize SM/src/NT/ICLK.cc:71-128 */
			if (p.chained_warp) am->initialize_pix_grad_pts(ssm->curr_pts.data());
			else { ssm->update_grad_pts(am->grad_eps); am->initialize_pix_grad_warped(ssm->grad_pts.data()); }
			am->initialize_similarity(); am->initialize_grad(); am->initialize_hess();
			if (p.chained_warp) ssm->warped_pix_jacobian(J0.data(), am->dI0_dx.data());
			else ssm->init_pix_jacobian(J0.data(), am->dI0_dx.data());
			am->cmpt_init_jacobian(g.data(), J0.data(), S);
			if (p.sec_ord_hess) {
				pix_hess_init();
				if (p.hess_type != 1 /*CurrentSelf*/) pix_hessian_from_init(D0.data());
			}
			if (p.hess_type == 0 /*InitialSelf*/) {
				self_hessian(H.data(), J0.data(), D0.data());
				if (p.leven_marq) H0 = H;
			}
			break;
		}
	}

	/* nt::ESM::setRegion NT/ESM.cc:148-168 ; nt::ICLK::setRegion NT/ICLK.cc:131-157 (update_ssm=false) ;
	 * nt::FCLK::setRegion NT/FCLK.cc:360+ just resets the SSM */
	void set_region(const double *corners) {
		ssm->set_corners(corners);
		if (sm == MTFO_SM_ESM) {
			ssm->init_pix_jacobian(J0.data(), am->dI0_dx.data());
			if (p.sec_ord_hess) ssm->init_pix_hessian(D0.data(), am->d2I0_dx2.data(), am->dI0_dx.data());
			if (p.hess_type == 0 || p.hess_type == 2) { self_hessian(H.data(), J0.data(), D0.data()); H0 = H; }
		} else if (sm == MTFO_SM_FCLK && p.hess_type == 0) { /* NT/FCLK.cc:360-376: init_self_hessian is NOT refreshed */
			ssm->init_pix_jacobian(J0.data(), am->dI0_dx.data());
			if (p.sec_ord_hess) ssm->init_pix_hessian(D0.data(), am->d2I0_dx2.data(), am->dI0_dx.data());
			self_hessian(H.data(), J0.data(), D0.data());
		}
	}

	/* trace record = [f, g, H (before LM damping), dp, corners after the update] */
	void record(double f, const vecd &Hrec) {
		size_t off = trace.size();
		trace.resize(off + rec_len);
		double *r = &trace[off];
		r[0] = f;
		std::copy(g.begin(), g.end(), r + 1);
		std::copy(Hrec.begin(), Hrec.end(), r + 1 + S);
		std::copy(dp.begin(), dp.end(), r + 1 + S + S * S);
		std::copy(ssm->curr_corners.begin(), ssm->curr_corners.end(), r + 1 + 2 * S + S * S);
		++n_rec;
	}

	/* hessian += delta * diag(hessian) (NT/FCLK.cc:290-296, NT/ESM.cc:257-263, NT/ICLK.cc:253-259)
	 * then state_update = -hessian.colPivHouseholderQr().solve(jacobian^T) */
	void solve_and_negate(double delta) {
		if (p.leven_marq) for (int i = 0; i < S; ++i) H[i * S + i] += delta * H[i * S + i];
		colpiv_qr_solve(S, H.data(), g.data(), dp.data());
		for (int i = 0; i < S; ++i) dp[i] = -dp[i];
	}
	double corner_change() const {
		double s = 0;
		for (int i = 0; i < 8; ++i) { double d = prev_corners[i] - ssm->curr_corners[i]; s += d * d; }
		return s;
	}

	int update() {
		trace.clear(); n_rec = 0;
		switch (sm) {
		case MTFO_SM_ESM: return update_esm();
		case MTFO_SM_FCLK: return update_fclk();
		default: return update_iclk();
		}
	}

	/* nt::ESM::update SM/src/NT/ESM.cc:170-296, cmptJacobian :298-313, cmptHessian :315-377 */
	int update_esm() {
		double prev_sim = 0, delta = p.lm_delta_init;
		bool state_reset = false;
		int iters = 0;
		for (int it = 0; it < p.max_iters; ++it) {
			++iters;
			am->update_pix_vals(ssm->curr_pts.data());
			am->update_similarity(false);
			if (p.leven_marq && !state_reset) {
				double cur = am->f;
				if (it > 0) {
					if (cur < prev_sim) {
						delta *= p.lm_delta_update;
						ssm->invert_state(inv_dp.data(), dp.data());
						ssm->compositional_update(inv_dp.data());
						state_reset = true;
						continue;
					}
					if (cur > prev_sim) delta /= p.lm_delta_update;
				}
				prev_sim = cur;
			}
			state_reset = false;
			pix_jacobian_update(Jt.data()); /* DISABLE_SPI ordering, NT/ESM.cc:234-238 */
			bool need_mean = (p.jac_type == 0) || (p.hess_type == 3);
			if (need_mean) {
				Jmean.resize(Jt.size());
				for (size_t i = 0; i < Jt.size(); ++i) Jmean[i] = (J0[i] + Jt[i]) / 2.0;
			}
			if (p.sec_ord_hess && p.hess_type != 0) pix_hessian_update(Dt.data());
			am->update_curr_grad();
			am->update_init_grad();
			if (p.jac_type == 0) am->cmpt_curr_jacobian(g.data(), Jmean.data(), S);
			else {
				am->cmpt_difference_of_jacobians(g.data(), J0.data(), Jt.data(), S);
				for (int i = 0; i < S; ++i) g[i] *= 0.5;
			}
			switch (p.hess_type) {
			case 0: if (p.leven_marq) H = H0; break;                               /* InitialSelf */
			case 3: /* Original */
				if (p.sec_ord_hess) {
					Dmean.resize(Dt.size());
					for (size_t i = 0; i < Dt.size(); ++i) Dmean[i] = (D0[i] + Dt[i]) / 2.0;
					am->cmpt_curr_hessian2(H.data(), Jmean.data(), Dmean.data(), S);
				} else am->cmpt_curr_hessian(H.data(), Jmean.data(), S);
				break;
			case 4: /* SumOfStd */
				if (p.sec_ord_hess) am->cmpt_sum_of_hessians2(H.data(), J0.data(), Jt.data(), D0.data(), Dt.data(), S);
				else am->cmpt_sum_of_hessians(H.data(), J0.data(), Jt.data(), S);
				for (auto &v : H) { v *= 0.5; }
				break;
			case 2: /* SumOfSelf */
				self_hessian(H.data(), Jt.data(), Dt.data());
				for (int i = 0; i < S * S; ++i) { H[i] = (H[i] + H0[i]) * 0.5; }
				break;
			case 1: self_hessian(H.data(), Jt.data(), Dt.data()); break;             /* CurrentSelf */
			case 5: /* Std */
				if (p.sec_ord_hess) am->cmpt_curr_hessian2(H.data(), Jt.data(), Dt.data(), S);
				else am->cmpt_curr_hessian(H.data(), Jt.data(), S);
				break;
			}
			double f_now = am->f;
			vecd H_plain = H;
			solve_and_negate(delta);
			prev_corners = ssm->curr_corners;
			ssm->compositional_update(dp.data());
			record(f_now, H_plain);
			if (corner_change() < p.epsilon) break;
		}
		return iters;
	}

	/* nt::FCLK::update SM/src/NT/FCLK.cc:171-358 */
	int update_fclk() {
		double prev_sim = 0, delta = p.lm_delta_init;
		bool state_reset = false;
		int iters = 0, it = 0;
		while (it < p.max_iters) {
			++iters;
			am->update_pix_vals(ssm->curr_pts.data());
			am->update_similarity(false);
			if (p.leven_marq && !state_reset) {
				double cur = am->f;
				if (it > 0) {
					if (cur < prev_sim) {
						delta *= p.lm_delta_update;
						ssm->invert_state(inv_dp.data(), dp.data());
						ssm->compositional_update(inv_dp.data());
						state_reset = true;
						continue; /* iter_id is not advanced on a rejected step (while loop, NT/FCLK.cc:187,214) */
					}
					if (cur > prev_sim) delta /= p.lm_delta_update;
				}
				prev_sim = cur;
			}
			state_reset = false;
			am->update_curr_grad();
			pix_jacobian_update(Jt.data());
			if (p.sec_ord_hess && p.hess_type != 0) pix_hessian_update(Dt.data());
			am->cmpt_curr_jacobian(g.data(), Jt.data(), S);
			switch (p.hess_type) {
			case 0: if (p.leven_marq) H = H0; break;                       /* InitialSelf */
			case 1: self_hessian(H.data(), Jt.data(), Dt.data()); break;    /* CurrentSelf */
			case 2: /* Std */
				if (p.sec_ord_hess) am->cmpt_curr_hessian2(H.data(), Jt.data(), Dt.data(), S);
				else am->cmpt_curr_hessian(H.data(), Jt.data(), S);
				break;
			}
			double f_now = am->f;
			vecd H_plain = H;
			solve_and_negate(delta);
			prev_corners = ssm->curr_corners;
			ssm->compositional_update(dp.data());
			record(f_now, H_plain);
			if (corner_change() < p.epsilon) break;
			++it;
		}
		return iters;
	}

	/* nt::ICLK::update SM/src/NT/ICLK.cc:160-299 */
	int update_iclk() {
		double prev_sim = 0, delta = p.lm_delta_init;
		bool state_reset = false;
		int iters = 0;
		for (int it = 0; it < p.max_iters; ++it) {
			++iters;
			am->update_pix_vals(ssm->curr_pts.data());
			am->update_similarity(false);
			if (p.leven_marq && !state_reset) {
				double cur = am->f;
				if (it > 0) {
					if (cur < prev_sim) {
						delta *= p.lm_delta_update;
						ssm->compositional_update(dp.data()); /* undo of the inverse update, NT/ICLK.cc:188 */
						state_reset = true;
						continue;
					}
					if (cur > prev_sim) delta /= p.lm_delta_update;
				}
				prev_sim = cur;
			}
			state_reset = false;
			am->update_init_grad();
			am->cmpt_init_jacobian(g.data(), J0.data(), S);
			switch (p.hess_type) {
			case 0: if (p.leven_marq) H = H0; break;                      /* InitialSelf */
			case 1: pix_jacobian_update(Jt.data());                         /* CurrentSelf */
				if (p.sec_ord_hess) pix_hessian_update(Dt.data());
				self_hessian(H.data(), Jt.data(), Dt.data()); break;
			case 2: /* Std */
				if (p.sec_ord_hess) am->cmpt_init_hessian2(H.data(), J0.data(), D0.data(), S);
				else am->cmpt_init_hessian(H.data(), J0.data(), S);
				break;
			}
			double f_now = am->f;
			vecd H_plain = H;
			solve_and_negate(delta);
			prev_corners = ssm->curr_corners;
			ssm->invert_state(inv_dp.data(), dp.data());
			ssm->compositional_update(inv_dp.data());
			record(f_now, H_plain);
			if (corner_change() < p.epsilon) break;
		}
		return iters;
	}
};

/* ===================================================================== */
/* GridTracker (SM/src/GridTracker.cc): the patch layout and the frame loop */
/* ===================================================================== */
/* GridTracker<SSM> over patch trackers that are the restated nt:: search methods above.  The robust fit of the grid SSM to the
 * patch centroids (ssm.estimateWarpFromPts -> utils::estimateHomography / estimateAffine: RANSAC / LMedS, SURVEY.md section 2: out
 * of scope) is a callback the test supplies; everything else of initialize / update / setRegion / resetTrackers is restated. */
struct mtfo_grid {
	mtfo_grid_params p;
	int resx, resy, n;
	mtfo_ssm *ssm;                       /* the grid SSM: (resx x resy) points laid over the tracked region */
	std::vector<mtfo_tracker *> trackers;
	std::vector<int> linear_idx;         /* _linear_idx: (grid_size_y + 1) x (grid_size_x + 1), GridTracker.cc:139-146 */
	vecd patch_corners;                  /* n x 8: what resetTrackers handed tracker k (CornersT layout: x, y per corner) */
	std::vector<float> prev_pts, curr_pts;   /* std::vector<cv::Point2f> (GridTracker.h:102-103): centroids are rounded to float */
	vecd ssm_update, region;
	double centroid_dist_x, centroid_dist_y;
	bool reinit_at_each_frame;
	mtfo_grid_estimator est = nullptr; void *est_user = nullptr;
	/* forward-backward error estimation (GridTracker.h:104-106, GridTracker.cc:186-190) */
	bool enable_fb_err_est = false;
	const float *curr_img = nullptr; int img_h = 0, img_w = 0;   /* curr_img: a header on the caller's buffer (:227) */
	std::vector<float> prev_img;                                   /* prev_img = curr_img.clone() (:241-243, :266) */
	std::vector<float> fb_prev_pts; std::vector<unsigned char> fb_err_mask;
	vecd fb_locations, fb_regions;                                 /* n x 8 each (records for the parity tests) */
	std::vector<float> est_prev, est_curr;                         /* the pairs the estimator saw in the last update */

	/* GridTrackerParams::updateRes SM/src/GridTracker.cc:86-94 */
	static void update_res(const mtfo_grid_params &gp, int &rx, int &ry) {
		if (gp.dyn_patch_size || gp.patch_centroid_inside) { rx = gp.grid_size_x + 1; ry = gp.grid_size_y + 1; }
		else { rx = gp.grid_size_x; ry = gp.grid_size_y; }
	}

	/* constructor SM/src/GridTracker.cc:97-160 (the checks of :124-134 are made by mtfo_grid_create) */
	mtfo_grid(const mtfo_grid_params &gp, mtfo_ssm *grid_ssm, mtfo_tracker **trk, int n_trk) : p(gp), ssm(grid_ssm) {
		update_res(p, resx, resy);
		n = p.grid_size_x * p.grid_size_y;
		for (int i = 0; i < n_trk; ++i) trackers.push_back(trk[i]);
		reinit_at_each_frame = p.reset_at_each_frame == 1;                        /* :136 */
		const int sub_x = p.grid_size_x + 1, sub_y = p.grid_size_y + 1;            /* :139-146 */
		linear_idx.resize(static_cast<size_t>(sub_x) * sub_y);
		for (int idy = 0; idy < sub_y; ++idy)
			for (int idx = 0; idx < sub_x; ++idx) linear_idx[static_cast<size_t>(idy) * sub_x + idx] = idy * sub_x + idx;
		patch_corners.assign(static_cast<size_t>(8) * n, 0.0);
		prev_pts.assign(static_cast<size_t>(2) * n, 0.f); curr_pts = prev_pts;     /* :153-154 */
		ssm_update.assign(ssm->S, 0.0); region.assign(8, 0.0);
		centroid_dist_x = p.patch_size_x / 2.0;                                   /* :156-157 */
		centroid_dist_y = p.patch_size_y / 2.0;
		if (p.fb_err_thresh > 0) {                                                /* :186-190 */
			enable_fb_err_est = true;
			fb_prev_pts.assign(static_cast<size_t>(2) * n, 0.f);
			fb_err_mask.assign(n, 0);
			fb_locations.assign(static_cast<size_t>(8) * n, 0.0); fb_regions = fb_locations;
		}
	}
	/* GridTracker::setImage :205-231 without the pyramid branch */
	void set_image(const float *img, int h, int w) {
		for (auto *t : trackers) { t->am->img = img; t->am->h = h; t->am->w = w; }
		curr_img = img; img_h = h; img_w = w;
	}
	void clone_prev() { prev_img.assign(curr_img, curr_img + static_cast<size_t>(img_h) * img_w); }
	int lin(int r, int c) const { return linear_idx[static_cast<size_t>(r) * (p.grid_size_x + 1) + c]; }

	/* the corners resetTrackers builds for one patch, SM/src/GridTracker.cc:354-380 */
	void layout_patch(int tracker_id, double *pc /* 8: x, y per corner */) const {
		const int row_id = tracker_id / p.grid_size_x, col_id = tracker_id % p.grid_size_x;   /* :354-355 */
		const double *pts = ssm->curr_pts.data();                                              /* ssm.getPts(): 2 x (resx * resy) */
		/* :357-367 -- the four surrounding grid points TL, TR, BR, BL.  With dyn_patch_size = patch_centroid_inside = 0 the SSM has
		 * only grid_size^2 points and these reads index past what the layout means (the reference overwrites the result below, and its
		 * Eigen bounds asserts are compiled out by NDEBUG): they are skipped here. */
		if (p.dyn_patch_size || p.patch_centroid_inside) {
			const int id[4] = {lin(row_id, col_id), lin(row_id, col_id + 1), lin(row_id + 1, col_id + 1), lin(row_id + 1, col_id)};
			for (int q = 0; q < 4; ++q) { pc[2 * q] = pts[2 * id[q]]; pc[2 * q + 1] = pts[2 * id[q] + 1]; }
		}
		if (!p.dyn_patch_size) {                                                               /* :369-380 */
			double cx = pts[2 * tracker_id], cy = pts[2 * tracker_id + 1];                       /* ssm.getPts().col(tracker_id) */
			if (p.patch_centroid_inside) {                                                       /* utils::getCentroid miscUtils.h:481-487 */
				cx = (pc[0] + pc[2] + pc[4] + pc[6]) / 4.0;
				cy = (pc[1] + pc[3] + pc[5] + pc[7]) / 4.0;
			}
			/* utils::Corners(cv::Rect_<double>(x, y, w, h)) miscUtils.h:42-52 */
			const double min_x = cx - centroid_dist_x, min_y = cy - centroid_dist_y;
			const double max_x = min_x + p.patch_size_x, max_y = min_y + p.patch_size_y;
			pc[0] = pc[6] = min_x; pc[2] = pc[4] = max_x;
			pc[1] = pc[3] = min_y; pc[5] = pc[7] = max_y;
		}
	}
	/* utils::getCentroid(cv::Point2f&, corners) miscUtils.h:472-480: rounded to float */
	static void centroid_f(float *dst, const double *c) {
		dst[0] = static_cast<float>((c[0] + c[2] + c[4] + c[6]) / 4.0);
		dst[1] = static_cast<float>((c[1] + c[3] + c[5] + c[7]) / 4.0);
	}
	/* GridTracker::resetTrackers SM/src/GridTracker.cc:345-392 */
	void reset_trackers(bool reinit) {
		for (int tracker_id = 0; tracker_id < n; ++tracker_id) {
			double *pc = &patch_corners[static_cast<size_t>(8) * tracker_id];
			layout_patch(tracker_id, pc);
			if (!trackers.empty()) {
				if (reinit) trackers[tracker_id]->initialize(pc);                                /* :381-385 */
				else trackers[tracker_id]->set_region(pc);
				centroid_f(&prev_pts[2 * tracker_id], trackers[tracker_id]->ssm->curr_corners.data());   /* :387 */
			} else centroid_f(&prev_pts[2 * tracker_id], pc);
		}
	}
	/* GridTracker::initialize :233-246 */
	void initialize(const double *corners) {
		ssm->set_corners(corners);   /* ssm.initialize(corners) = setCorners + the init flag (StateSpaceModel.h:84-92) */
		reset_trackers(true);
		curr_pts = prev_pts;
		region = ssm->curr_corners;
		if (enable_fb_err_est) {                                                                 /* :241-243 */
			if (!curr_img) return;   /* (mtfo_grid_set_image was not used: mtfo_grid_update then reports -3) */
			clone_prev();
		}
	}
	/* the second half of GridTracker::backwardEstimation :307-332: which patch trackers came back to where they started, and the point
	 * pairs of those -- filled up in tracker order to n_model_pts when fewer survive */
	static void fb_mask(int n, const float *prev_pts, const float *curr_pts, const float *fb_prev_pts, double fb_err_thresh, int n_model_pts,
		unsigned char *fb_err_mask, std::vector<float> &prev_masked, std::vector<float> &curr_masked) {
		prev_masked.clear(); curr_masked.clear();                                                 /* :307 */
		auto push = [&](int id) {
			prev_masked.push_back(prev_pts[2 * id]); prev_masked.push_back(prev_pts[2 * id + 1]);
			curr_masked.push_back(curr_pts[2 * id]); curr_masked.push_back(curr_pts[2 * id + 1]);
		};
		for (int tracker_id = 0; tracker_id < n; ++tracker_id) {
			/* :309-310: Point2f members: the difference is taken in float, then widened */
			const double diff_x = fb_prev_pts[2 * tracker_id] - prev_pts[2 * tracker_id];
			const double diff_y = fb_prev_pts[2 * tracker_id + 1] - prev_pts[2 * tracker_id + 1];
			if (diff_x * diff_x + diff_y * diff_y > fb_err_thresh) fb_err_mask[tracker_id] = 0;   /* :312-313 */
			else { fb_err_mask[tracker_id] = 1; push(tracker_id); }                              /* :314-318 */
		}
		if (static_cast<int>(prev_masked.size() / 2) < n_model_pts) {                             /* :321-332 */
			for (int tracker_id = 0; tracker_id < n; ++tracker_id) {
				if (fb_err_mask[tracker_id]) continue;
				push(tracker_id);
				fb_err_mask[tracker_id] = 1;
				if (static_cast<int>(prev_masked.size() / 2) == n_model_pts) break;
			}
		}
	}
	/* GridTracker::backwardEstimation :294-343 */
	int backward_estimation() {
		for (int tracker_id = 0; tracker_id < n; ++tracker_id) {
			mtfo_tracker *t = trackers[tracker_id];
			double *loc = &fb_locations[static_cast<size_t>(8) * tracker_id];
			std::copy(t->ssm->curr_corners.begin(), t->ssm->curr_corners.end(), loc);             /* getRegion().clone() :296 */
			if (p.fb_reinit) t->initialize(loc);                                                  /* :297-299 */
			t->am->img = prev_img.data();                                                         /* setImage(prev_img) :300 */
			t->update();                                                                          /* :301 */
			centroid_f(&fb_prev_pts[2 * tracker_id], t->ssm->curr_corners.data());                /* :302 */
			std::copy(t->ssm->curr_corners.begin(), t->ssm->curr_corners.end(), &fb_regions[static_cast<size_t>(8) * tracker_id]);
			t->am->img = curr_img;                                                                /* setImage(curr_img) :304 */
			t->set_region(loc);                                                                   /* :305 */
		}
		fb_mask(n, prev_pts.data(), curr_pts.data(), fb_prev_pts.data(), p.fb_err_thresh, p.n_model_pts, fb_err_mask.data(), est_prev, est_curr);
		est(est_user, static_cast<int>(est_prev.size() / 2), est_prev.data(), est_curr.data(), ssm_update.data());   /* :334-335 */
		return 0;
	}
	/* GridTracker::update :247-285 */
	int update() {
		if (enable_fb_err_est && prev_img.empty()) return -3;
		for (int tracker_id = 0; tracker_id < n; ++tracker_id) {
			trackers[tracker_id]->update();
			centroid_f(&curr_pts[2 * tracker_id], trackers[tracker_id]->ssm->curr_corners.data());
		}
		if (!est) return -1;
		if (enable_fb_err_est) {                                                                 /* :263-266 */
			backward_estimation();
			clone_prev();
		} else {
			est_prev = prev_pts; est_curr = curr_pts;
			est(est_user, n, prev_pts.data(), curr_pts.data(), ssm_update.data());               /* ssm.estimateWarpFromPts :267 */
		}
		double opt_warped_corners[8];
		ssm->apply_warp_to_corners(opt_warped_corners, ssm->curr_corners.data(), ssm_update.data());   /* :270-271 */
		ssm->set_corners(opt_warped_corners);                                                    /* :272 */
		if (p.reset_at_each_frame) reset_trackers(reinit_at_each_frame);                         /* :273-274 */
		else prev_pts = curr_pts;
		region = ssm->curr_corners;
		return 0;
	}
	/* GridTracker::setRegion :287-292 */
	void set_region(const double *corners) {
		ssm->set_corners(corners);
		reset_trackers(reinit_at_each_frame);
		region = ssm->curr_corners;
	}
};

/* ===================================================================== */
/* C API                                                                  */
/* ===================================================================== */
extern "C" {

double mtfo_get_pix_val(const float *img, int h, int w, double x, double y) { return pix_val(img, h, w, x, y); }
void mtfo_get_pix_vals(double *out, const float *img, int h, int w, const double *pts, int n,
	double norm_mult, double norm_add) { pix_vals(out, img, h, w, pts, n, norm_mult, norm_add); }
void mtfo_get_img_grad(double *grad, const float *img, int h, int w, const double *pts,
	double grad_eps, int n, double pix_mult) { img_grad(grad, img, h, w, pts, grad_eps, n, pix_mult); }
void mtfo_get_warped_img_grad(double *grad, const float *img, int h, int w, const double *grad_pts,
	double grad_eps, int n, double pix_mult) { warped_img_grad(grad, img, h, w, grad_pts, grad_eps, n, pix_mult); }

void mtfo_get_img_hess(double *hess, const float *img, int h, int w, const double *pts, double hess_eps, int n, double pix_mult) {
	img_hess(hess, img, h, w, pts, hess_eps, n, pix_mult);
}
void mtfo_get_warped_img_hess(double *hess, const float *img, int h, int w, const double *pts, const double *hess_pts,
	double hess_eps, int n, double pix_mult) {
	warped_img_hess(hess, img, h, w, pts, hess_pts, hess_eps, n, pix_mult);
}
void mtfo_homography_dlt(const double *in_corners, const double *out_corners, double *warp9) {
	Mat3 H = homography_dlt(in_corners, out_corners);
	std::memcpy(warp9, H.m, sizeof(H.m));
}
void mtfo_colpiv_qr_solve(int n, const double *A, const double *b, double *x) { colpiv_qr_solve(n, A, b, x); }
void mtfo_norm_unit_square_pts(double *pts, double *corners, int resx, int resy,
	double min_x, double min_y, double max_x, double max_y) {
	norm_unit_square_pts(pts, corners, resx, resy, min_x, min_y, max_x, max_y);
}

mtfo_ssm *mtfo_ssm_create(int kind, int resx, int resy) { return new mtfo_ssm(kind, resx, resy); }
void mtfo_ssm_destroy(mtfo_ssm *s) { delete s; }
int mtfo_ssm_state_size(const mtfo_ssm *s) { return s->S; }
int mtfo_ssm_n_pts(const mtfo_ssm *s) { return s->n; }
void mtfo_ssm_set_corners(mtfo_ssm *s, const double *c) { s->set_corners(c); }
void mtfo_ssm_set_state(mtfo_ssm *s, const double *p) { s->set_state(p); }
/* ProjectiveBase::estimateStateSigma SSM/src/ProjectiveBase.cc:201-213 with Homography::getCurrPixGrad (Homography.cc:143-155) /
 * Affine::getCurrPixGrad = getInitPixGrad (Affine.h:30-32, Affine.cc:152-158) */
void mtfo_ssm_estimate_state_sigma(mtfo_ssm *s, double pix_sigma, double *state_sigma) {
	const int n = s->n, S = s->S;
	vecd mean(S, 0.0);
	for (int i = 0; i < n; ++i) {
		const double x = s->init_pts[2 * i], y = s->init_pts[2 * i + 1];
		double g[2][8] = {{0}};
		if (s->kind == MTFO_SSM_HOMOGRAPHY) {
			const double cx = s->curr_pts[2 * i], cy = s->curr_pts[2 * i + 1], inv_d = 1.0 / s->curr_pts_hm[3 * i + 2];
			const double r0[8] = {x, y, 1, 0, 0, 0, -x * cx, -y * cx}, r1[8] = {0, 0, 0, x, y, 1, -x * cy, -y * cy};
			for (int k = 0; k < 8; ++k) { g[0][k] = r0[k] * inv_d; g[1][k] = r1[k] * inv_d; }
		} else {
			const double r0[6] = {1, 0, x, y, 0, 0}, r1[6] = {0, 1, 0, 0, x, y};
			for (int k = 0; k < 6; ++k) { g[0][k] = r0[k]; g[1][k] = r1[k]; }
		}
		for (int k = 0; k < S; ++k) mean[k] += std::sqrt(g[0][k] * g[0][k] + g[1][k] * g[1][k]);
	}
	for (int k = 0; k < S; ++k) state_sigma[k] = pix_sigma / (mean[k] / n);
}
void mtfo_ssm_compositional_update(mtfo_ssm *s, const double *dp) { s->compositional_update(dp); }
void mtfo_ssm_invert_state(mtfo_ssm *s, double *inv, const double *p) { s->invert_state(inv, p); }
void mtfo_ssm_update_grad_pts(mtfo_ssm *s, double eps) { s->update_grad_pts(eps); }
void mtfo_ssm_cmpt_init_pix_jacobian(mtfo_ssm *s, double *J, const double *g) { s->init_pix_jacobian(J, g); }
void mtfo_ssm_cmpt_pix_jacobian(mtfo_ssm *s, double *J, const double *g) { s->pix_jacobian(J, g); }
void mtfo_ssm_cmpt_warped_pix_jacobian(mtfo_ssm *s, double *J, const double *g) { s->warped_pix_jacobian(J, g); }
void mtfo_ssm_cmpt_approx_pix_jacobian(mtfo_ssm *s, double *J, const double *g) { s->approx_pix_jacobian(J, g); }
void mtfo_ssm_update_hess_pts(mtfo_ssm *s, double eps) { s->update_hess_pts(eps); }
int mtfo_ssm_cmpt_init_pix_hessian(mtfo_ssm *s, double *d2, const double *ph, const double *g) { s->init_pix_hessian(d2, ph, g); return 0; }
int mtfo_ssm_cmpt_pix_hessian(mtfo_ssm *s, double *d2, const double *ph, const double *g) { return s->pix_hessian(d2, ph, g); }
int mtfo_ssm_cmpt_warped_pix_hessian(mtfo_ssm *s, double *d2, const double *ph, const double *g) { s->warped_pix_hessian(d2, ph, g); return 0; }
int mtfo_ssm_cmpt_approx_pix_hessian(mtfo_ssm *s, double *d2, const double *ph, const double *g) { return s->approx_pix_hessian(d2, ph, g); }
void mtfo_ssm_apply_warp_to_corners(mtfo_ssm *s, double *out, const double *in, const double *p) { s->apply_warp_to_corners(out, in, p); }
void mtfo_ssm_apply_warp_to_pts(mtfo_ssm *s, double *out, const double *in, int n_pts, const double *p) { s->apply_warp_to_pts(out, in, n_pts, p); }
void mtfo_ssm_compose_warps(mtfo_ssm *s, double *out, const double *p1, const double *p2) { s->compose_warps(out, p1, p2); }
void mtfo_ssm_estimate_warp_from_corners(mtfo_ssm *s, double *out, const double *in_c, const double *out_c) { s->estimate_warp_from_corners(out, in_c, out_c); }
void mtfo_ssm_additive_update(mtfo_ssm *s, const double *dp) { s->additive_update(dp); }
void mtfo_ssm_compositional_random_walk(mtfo_ssm *s, double *out, const double *base, const double *pert) {
	s->compositional_random_walk(out, base, pert);
}
void mtfo_ssm_get(const mtfo_ssm *s, int what, double *dst) {
	const vecd *v = nullptr;
	switch (what) {
	case 0: v = &s->curr_pts; break;
	case 1: v = &s->init_pts; break;
	case 2: v = &s->curr_corners; break;
	case 3: v = &s->init_corners; break;
	case 4: v = &s->state; break;
	case 5: std::memcpy(dst, s->curr_warp.m, sizeof(s->curr_warp.m)); return;
	case 6: v = &s->grad_pts; break;
	case 7: v = &s->curr_pts_hm; break;
	case 8: v = &s->init_pts_hm; break;
	case 9: v = &s->hess_pts; break;
	default: return;
	}
	std::copy(v->begin(), v->end(), dst);
}

mtfo_am *mtfo_am_create(int kind, int resx, int resy, double grad_eps, double alpha,
	int n_bins, double pre_seed, int pou) {
	return new mtfo_am(kind, resx, resy, grad_eps, alpha, n_bins, pre_seed, pou);
}
void mtfo_am_destroy(mtfo_am *a) { delete a; }
int mtfo_am_n_pix(const mtfo_am *a) { return a->n; }
void mtfo_am_set_curr_img(mtfo_am *a, const float *img, int h, int w) { a->img = img; a->h = h; a->w = w; }
void mtfo_am_set_channels(mtfo_am *a, int n_channels) { a->set_channels(n_channels); }
void mtfo_ssm_set_channels(mtfo_ssm *s, int n_channels) { s->C = n_channels; s->P = s->n * n_channels; }
int mtfo_am_patch_size(const mtfo_am *a) { return a->n; }
void mtfo_am_initialize_pix_vals(mtfo_am *a, const double *pts) { a->initialize_pix_vals(pts); }
void mtfo_am_update_pix_vals(mtfo_am *a, const double *pts) { a->update_pix_vals(pts); }
int mtfo_am_update_model(mtfo_am *a, const double *pts, double learning_rate) { return a->update_model(pts, learning_rate); }
void mtfo_am_initialize_pix_grad_pts(mtfo_am *a, const double *pts) { a->initialize_pix_grad_pts(pts); }
void mtfo_am_initialize_pix_grad_warped(mtfo_am *a, const double *gp) { a->initialize_pix_grad_warped(gp); }
void mtfo_am_update_pix_grad_pts(mtfo_am *a, const double *pts) { a->update_pix_grad_pts(pts); }
void mtfo_am_update_pix_grad_warped(mtfo_am *a, const double *gp) { a->update_pix_grad_warped(gp); }
void mtfo_am_set_hess_eps(mtfo_am *a, double eps) { a->hess_eps = eps; }
void mtfo_am_initialize_pix_hess_pts(mtfo_am *a, const double *pts) { a->initialize_pix_hess_pts(pts); }
void mtfo_am_initialize_pix_hess_warped(mtfo_am *a, const double *pts, const double *hp) { a->initialize_pix_hess_warped(pts, hp); }
void mtfo_am_update_pix_hess_pts(mtfo_am *a, const double *pts) { a->update_pix_hess_pts(pts); }
void mtfo_am_update_pix_hess_warped(mtfo_am *a, const double *pts, const double *hp) { a->update_pix_hess_warped(pts, hp); }
int mtfo_am_cmpt_init_hessian2(mtfo_am *a, double *H, const double *J0, const double *d2, int S) { return a->cmpt_init_hessian2(H, J0, d2, S); }
int mtfo_am_cmpt_curr_hessian2(mtfo_am *a, double *H, const double *Jt, const double *d2, int S) { return a->cmpt_curr_hessian2(H, Jt, d2, S); }
int mtfo_am_cmpt_self_hessian2(mtfo_am *a, double *H, const double *Jt, const double *d2, int S) { return a->cmpt_self_hessian2(H, Jt, d2, S); }
int mtfo_am_cmpt_sum_of_hessians2(mtfo_am *a, double *H, const double *J0, const double *Jt, const double *d20, const double *d2t, int S) {
	return a->cmpt_sum_of_hessians2(H, J0, Jt, d20, d2t, S);
}
void mtfo_am_initialize_similarity(mtfo_am *a) { a->initialize_similarity(); }
void mtfo_am_initialize_grad(mtfo_am *a) { a->initialize_grad(); }
void mtfo_am_initialize_hess(mtfo_am *a) { a->initialize_hess(); }
void mtfo_am_update_similarity(mtfo_am *a, int prereq_only) { a->update_similarity(prereq_only != 0); }
void mtfo_am_update_curr_grad(mtfo_am *a) { a->update_curr_grad(); }
void mtfo_am_update_init_grad(mtfo_am *a) { a->update_init_grad(); }
double mtfo_am_get_similarity(const mtfo_am *a) { return a->f; }
double mtfo_am_get_likelihood(const mtfo_am *a) { return a->likelihood(); }
void mtfo_am_cmpt_init_jacobian(mtfo_am *a, double *g, const double *J0, int S) { a->cmpt_init_jacobian(g, J0, S); }
void mtfo_am_cmpt_curr_jacobian(mtfo_am *a, double *g, const double *Jt, int S) { a->cmpt_curr_jacobian(g, Jt, S); }
void mtfo_am_cmpt_difference_of_jacobians(mtfo_am *a, double *g, const double *J0, const double *Jt, int S) {
	a->cmpt_difference_of_jacobians(g, J0, Jt, S);
}
void mtfo_am_cmpt_init_hessian(mtfo_am *a, double *H, const double *J0, int S) { a->cmpt_init_hessian(H, J0, S); }
void mtfo_am_cmpt_curr_hessian(mtfo_am *a, double *H, const double *Jt, int S) { a->cmpt_curr_hessian(H, Jt, S); }
void mtfo_am_cmpt_self_hessian(mtfo_am *a, double *H, const double *Jt, int S) { a->cmpt_self_hessian(H, Jt, S); }
void mtfo_am_cmpt_sum_of_hessians(mtfo_am *a, double *H, const double *J0, const double *Jt, int S) {
	a->cmpt_sum_of_hessians(H, J0, Jt, S);
}
void mtfo_am_get(const mtfo_am *a, int what, double *dst) {
	const vecd *v = nullptr;
	switch (what) {
	case 0: v = &a->I0; break;
	case 1: v = &a->It; break;
	case 2: v = &a->dI0_dx; break;
	case 3: v = &a->dIt_dx; break;
	case 4: v = &a->df_dI0; break;
	case 5: v = &a->df_dIt; break;
	case 6: v = &a->d2I0_dx2; break;
	case 7: v = &a->d2It_dx2; break;
	default: return;
	}
	std::copy(v->begin(), v->end(), dst);
}

/* utils::bSpl3 Utilities/include/mtf/Utilities/histUtils.h:161-175 */
static double bspl3_plain(double x) {
	if ((x > -2) && (x <= -1)) { double temp = 2 + x; return (temp * temp * temp) / 6; }
	else if ((x > -1) && (x <= 0)) return (4 - 3 * x * x * (2 + x)) / 6;
	else if ((x > 0) && (x <= 1)) return (4 - 3 * x * x * (2 - x)) / 6;
	else if ((x > 1) && (x < 2)) { double temp = 2 - x; return (temp * temp * temp) / 6; }
	return 0;
}
/* am->getDistFeatSize(): SSDBase.h:116-125 / NCC.cc:530-537 patch_size; MI.cc:122 5 * patch_size */
int mtfo_am_dist_feat_size(const mtfo_am *a) { return a->kind == MTFO_AM_MI ? 5 * a->n : a->n; }
/* AM::updateDistFeat(double *feat_addr) from the current pixel values It */
void mtfo_am_update_dist_feat(const mtfo_am *a, double *feat) {
	const int n = a->n;
	if (a->kind == MTFO_AM_SSD) {                 /* SSDBase.h:120-125: the patch itself */
		for (int i = 0; i < n; ++i) feat[i] = a->It[i];
	} else if (a->kind == MTFO_AM_NCC) {          /* NCC.cc:530-537: It - mean, then / norm */
		double mean = 0;
		for (int i = 0; i < n; ++i) mean += a->It[i];
		mean /= n;
		double sq = 0;
		for (int i = 0; i < n; ++i) { feat[i] = a->It[i] - mean; sq += feat[i] * feat[i]; }
		const double nrm = std::sqrt(sq);
		for (int i = 0; i < n; ++i) feat[i] /= nrm;
	} else {                                      /* MI.cc:736-747: row-major 5 x patch_size */
		for (int i = 0; i < n; ++i) {
			const int pix_val_floor = static_cast<int>(a->It[i]);
			double pix_diff = std::max(0, pix_val_floor - 1) - a->It[i];   /* std_bspl_ids(floor, 0) = max(0, floor - 1), MI.cc:115 */
			feat[i] = pix_val_floor;
			feat[n + i] = bspl3_plain(pix_diff);
			feat[2 * static_cast<size_t>(n) + i] = bspl3_plain(++pix_diff);
			feat[3 * static_cast<size_t>(n) + i] = bspl3_plain(++pix_diff);
			feat[4 * static_cast<size_t>(n) + i] = bspl3_plain(++pix_diff);
		}
	}
}
/* NN::generateDataset SM/src/NT/NN.cc:131-191 for given perturbations (generatePerturbation's draws are the caller's), compositional update:
 * the SSM walks W <- W inverse(P_k), the row is taken, W <- W P_k -- as the reference does, rounding of the round trip included */
void mtfo_nn_generate_dataset(mtfo_am *am, mtfo_ssm *ssm, const double *perturbations, int n_samples, double *dataset) {
	const int S = ssm->S, F = mtfo_am_dist_feat_size(am);
	vecd inv(S);
	for (int sample_id = 0; sample_id < n_samples; ++sample_id) {
		const double *pert = perturbations + static_cast<size_t>(sample_id) * S;
		ssm->invert_state(inv.data(), pert);                         /* :153 */
		ssm->compositional_update(inv.data());                       /* :154 */
		am->update_pix_vals(ssm->curr_pts.data());                   /* :158 */
		mtfo_am_update_dist_feat(am, dataset + static_cast<size_t>(sample_id) * F);   /* :159 */
		ssm->compositional_update(pert);                             /* :183-187 */
	}
}

mtfo_tracker *mtfo_tracker_create(int sm_kind, mtfo_am *am, mtfo_ssm *ssm, const mtfo_sm_params *params) {
	return new mtfo_tracker(sm_kind, am, ssm, *params);
}
void mtfo_tracker_destroy(mtfo_tracker *t) { delete t; }
void mtfo_tracker_initialize(mtfo_tracker *t, const double *corners) { t->initialize(corners); }
int mtfo_tracker_update(mtfo_tracker *t) { return t->update(); }
void mtfo_tracker_set_region(mtfo_tracker *t, const double *corners) { t->set_region(corners); }
void mtfo_tracker_get_region(const mtfo_tracker *t, double *corners) {
	std::copy(t->ssm->curr_corners.begin(), t->ssm->curr_corners.end(), corners);
}
int mtfo_tracker_status(const mtfo_tracker *t) { return t->status; }
int mtfo_tracker_trace_len(const mtfo_tracker *t) { return t->n_rec; }
int mtfo_tracker_trace(const mtfo_tracker *t, int iter, double *dst) {
	if (iter < 0 || iter >= t->n_rec) return 0;
	std::copy(t->trace.begin() + static_cast<size_t>(iter) * t->rec_len,
		t->trace.begin() + static_cast<size_t>(iter + 1) * t->rec_len, dst);
	return t->rec_len;
}

/* per-particle body of PF<AM,SSM>::update, SM/src/PF.cc:198-278 (likelihood_func = AM):
 * setState -> updatePixVals -> updateSimilarity(false) -> getLikelihood */
void mtfo_pf_score(mtfo_am *am, mtfo_ssm *ssm, const double *states, int n_particles,
	double *likelihoods, double *similarities) {
	int S = ssm->S;
	for (int k = 0; k < n_particles; ++k) {
		ssm->set_state(states + static_cast<size_t>(k) * S);
		am->update_pix_vals(ssm->curr_pts.data());
		am->update_similarity(false);
		if (likelihoods) likelihoods[k] = am->likelihood();
		if (similarities) similarities[k] = am->f;
	}
}

/* PF::binaryMultinomialResampling SM/src/PF.cc:345-394 with the uniforms supplied */
int mtfo_pf_binary_multinomial_resample(const double *wts, int n, const double *uniforms, int *resample_ids) {
	vecd cum(n);
	for (int i = 0; i < n; ++i) cum[i] = wts[i] + (i > 0 ? cum[i - 1] : 0.0);
	for (int i = 0; i < n; ++i) cum[i] /= cum[n - 1];
	double max_wt = std::numeric_limits<double>::lowest();
	int max_wt_id = 0;
	for (int k = 0; k < n; ++k) {
		double u = uniforms[k];
		int lo = 0, hi = n - 1, id = (lo + hi) / 2;
		while (hi > lo) {
			if (cum[id] >= u) hi = id; else lo = id + 1;
			id = (lo + hi) / 2;
		}
		resample_ids[k] = id;
		if (wts[id] >= max_wt) { max_wt = wts[id]; max_wt_id = k; }
	}
	return max_wt_id;
}

/* ---- one iteration of nt::PF::update's loop (SM/src/NT/PF.cc:260-447) with the random draws supplied by the caller
 * (the reference seeds boost::random from random_device, PF.cc:97-105, ProjectiveBase.cc:192-197: not reproducible).
 * normals: n x nz standard normals, nz = 10 (corner based homography sampling, Homography.cc:899-915: two draws of
 * distribution 0 for the common translation, then eight of distribution 1 for the four corners) or S (one per state
 * component, ProjectiveBase.cc:283-288); a draw of N(mean_k, sigma_k) is mean_k + sigma_k z.  uniforms: n draws for the
 * multinomial resampling.  states / ars: the particle set, replaced by the resampled one.  The SSM ends at the estimate. */
int mtfo_pf_iteration(mtfo_am *am, mtfo_ssm *ssm, const mtfo_pf_params *pp, double *states, double *ars, const double *normals,
	const double *uniforms, double max_similarity, double *wts_out, int *resample_ids, int *max_wt_id_out) {
	return mtfo_pf_iteration_ex(am, ssm, pp, nullptr, states, ars, normals, uniforms, max_similarity, wts_out, resample_ids, max_wt_id_out);
}
/* the same with the options of the shipped configuration (Config/modules.cfg:157-176): several sampler distributions whose
 * weights follow the average particle weight they produced (PF.cc:240-269, 345-369), adaptive resampling (PF.cc:114-118, 381-390).
 * mx == NULL: one distribution (pp->sigma / pp->mean), resampling every iteration.
 * The distribution of a particle is drawn from boost::random::discrete_distribution over distr_wts with its own generator
 * (PF.cc:111, 241, 262: an alias table over random_device seeds -- not reproducible); here the draw is supplied as a uniform
 * u in (0, 1] and inverted on the cumulative weights: the same distribution. */
int mtfo_pf_iteration_ex(mtfo_am *am, mtfo_ssm *ssm, const mtfo_pf_params *pp, mtfo_pf_mix *mx, double *states, double *ars,
	const double *normals, const double *uniforms, double max_similarity, double *wts_out, int *resample_ids, int *max_wt_id_out) {
	const int n = pp->n_particles, S = ssm->S;
	const int n_distr = mx ? mx->n_distr : 1;
	if (n_distr < 1 || n_distr > 8) return -4;
	/* n_distr == 1 switches update_distr_wts off (PF.cc:67); with several distributions and no update the reference zeroes the
	 * weights and then builds a discrete distribution from zeros (PF.cc:254-257, 241): a division by zero, not restated */
	if (n_distr > 1 && !mx->update_distr_wts) return -4;
	vecd distr_sum(n_distr, 0.0), distr_cum(n_distr, 0.0);
	std::vector<int> distr_cnt(n_distr, 0);
	if (n_distr > 1) { double c = 0; for (int i = 0; i < n_distr; ++i) { c += mx->distr_wts[i]; distr_cum[i] = c; } }
	const bool hom = ssm->kind == MTFO_SSM_HOMOGRAPHY;
	const bool corner_based = hom && pp->corner_based_sampling;
	const int pt_based = hom ? 0 : pp->pt_based_sampling;
	if (!hom) {
		/* Affine.cc:505-553: additive + point based and compositional RandomWalk + geometric throw FunctonNotImplemented in the
		 * reference; additive + geometric goes through Affine::stateToGeom (Affine.cc:411-462), a 2 x 2 JacobiSVD whose sign and
		 * ordering conventions select its branches -- not restated here (Eigen is absent, the conventions could not be pinned) */
		if (pp->update_type == 0) return -3;
		if (pp->dynamic_model == 0 && pt_based == 0) return -3;
	}
	const int nz = corner_based ? 10 : (hom ? S : (pt_based == 2 ? 8 : 6));
	vecd wts(n), cum(n), pert(S), ns(S), nar(S);
	int max_wt_id = 0;
	double max_wt = std::numeric_limits<double>::lowest();
	const double pi = 3.14159265358979323846;
	const double measurement_factor = 1.0 / std::sqrt(2 * pi * pp->measurement_sigma);   /* PF.cc:69-70 */
	for (int k = 0; k < n; ++k) {
		const double *z = normals + static_cast<size_t>(k) * nz;
		double *st = states + static_cast<size_t>(k) * S, *ar = ars + static_cast<size_t>(k) * S;
		/* the particle's sampler distribution: PF.cc:261-269 (ssm->setSampler(state_sigma[distr_id], state_mean[distr_id])) */
		const double *sg = pp->sigma, *mn = pp->mean;
		int distr_id = 0;
		if (n_distr > 1) {
			const double tgt = mx->distr_uniforms[k] * distr_cum[n_distr - 1];
			while (distr_id < n_distr - 1 && distr_cum[distr_id] < tgt) ++distr_id;
			sg = mx->sigma[distr_id]; mn = mx->mean[distr_id];
			if (mx->distr_ids_out) mx->distr_ids_out[k] = distr_id;
		}
		/* generatePerturbation: Homography.cc:899-915 / ProjectiveBase.cc:283-288 */
		if (corner_based) {
			double dc[8];
			const double tx = mn[0] + sg[0] * z[0], ty = mn[0] + sg[0] * z[1];
			for (int c = 0; c < 4; ++c) {
				dc[2 * c] = ssm->init_corners[2 * c] + (mn[1] + sg[1] * z[2 + 2 * c]) + tx;
				dc[2 * c + 1] = ssm->init_corners[2 * c + 1] + (mn[1] + sg[1] * z[3 + 2 * c]) + ty;
			}
			ssm->estimate_warp_from_corners(pert.data(), ssm->init_corners.data(), dc);
		} else if (!hom) {
			ssm->affine_generate_perturbation(pert.data(), pt_based, mn, sg, z);
		} else {
			for (int s2 = 0; s2 < S; ++s2) pert[s2] = mn[s2] + sg[s2] * z[s2];
		}
		/* dynamic model x update type: PF.cc:307-333 */
		if (pp->dynamic_model == 1) {
			if (pp->update_type == 0) {   /* ProjectiveBase::additiveAutoRegression1 :254-259 */
				for (int s2 = 0; s2 < S; ++s2) { ns[s2] = st[s2] + ar[s2] + pert[s2]; nar[s2] = pp->ar_coeff * (ns[s2] - st[s2]); }
			} else {                      /* Homography::compositionalAutoRegression1 Homography.cc:928-942; Affine: ProjectiveBase :260-276 (no normalisations) */
				Mat3 B = ssm->warp_from_state(st), P = ssm->warp_from_state(pert.data()), A = ssm->warp_from_state(ar);
				Mat3 W = mul3(mul3(B, A), P);
				if (hom) div3(W, W(2, 2));
				Mat3 AW = mul3(inv3(B), W);
				if (hom) div3(AW, AW(2, 2));
				ssm->state_from_warp(ns.data(), W); ssm->state_from_warp(nar.data(), AW);
				for (int s2 = 0; s2 < S; ++s2) nar[s2] *= pp->ar_coeff;
			}
			for (int s2 = 0; s2 < S; ++s2) ar[s2] = nar[s2];
		} else if (pp->update_type == 0) {   /* additiveRandomWalk :236-240 */
			for (int s2 = 0; s2 < S; ++s2) ns[s2] = st[s2] + pert[s2];
		} else {
			ssm->compositional_random_walk(ns.data(), st, pert.data());
		}
		for (int s2 = 0; s2 < S; ++s2) st[s2] = ns[s2];
		/* PF.cc:341-365 */
		ssm->set_state(st);
		am->update_pix_vals(ssm->curr_pts.data());
		am->update_similarity(false);
		const double val = max_similarity - am->f;
		double lik;
		if (pp->likelihood_func == 0) lik = am->likelihood();
		else if (pp->likelihood_func == 1) lik = measurement_factor * std::exp(-0.5 * val / pp->measurement_sigma);
		else lik = 1.0 / (1.0 + val);
		wts[k] = lik;
		cum[k] = k == 0 ? lik : lik + cum[k - 1];
		if (n_distr > 1) { distr_sum[distr_id] += lik; distr_cnt[distr_id] += 1; }   /* PF.cc:345-348 */
		if (lik >= max_wt) { max_wt = lik; max_wt_id = k; }
	}
	if (wts_out) std::memcpy(wts_out, wts.data(), sizeof(double) * n);
	if (n_distr > 1) {   /* PF.cc:354-369: average particle weight per distribution, normalised, floored at min_distr_wt */
		double wt_sum = 0;
		for (int i = 0; i < n_distr; ++i) {
			mx->distr_wts[i] = distr_sum[i];
			if (distr_cnt[i] > 0) { mx->distr_wts[i] /= distr_cnt[i]; wt_sum += mx->distr_wts[i]; }
		}
		for (int i = 0; i < n_distr; ++i) {
			mx->distr_wts[i] /= wt_sum;
			if (mx->distr_wts[i] < mx->min_distr_wt) mx->distr_wts[i] = mx->min_distr_wt;
		}
	}
	bool perform_resampling = true;
	if (mx && mx->adaptive_resampling_thresh > 0 && mx->adaptive_resampling_thresh <= 1) {   /* PF.cc:114-118, 381-390 */
		double sq = 0;
		for (int k = 0; k < n; ++k) { const double v = wts[k] / cum[n - 1]; sq += v * v; }
		const double n_eff = sq == 0 ? 0 : 1.0 / sq;
		if (n_eff > mx->adaptive_resampling_thresh * n) perform_resampling = false;
	}
	if (mx) mx->resampled = perform_resampling ? 1 : 0;
	if (!perform_resampling) {
		if (resample_ids) for (int k = 0; k < n; ++k) resample_ids[k] = k;
	} else if (pp->resampling_type == 1 || pp->resampling_type == 2) {   /* binary / linear multinomial: PF.cc:455-502, 505-536 */
		for (int k = 0; k < n; ++k) cum[k] /= cum[n - 1];
		vecd ns2(static_cast<size_t>(n) * S), na2(static_cast<size_t>(n) * S);
		max_wt = std::numeric_limits<double>::lowest();
		for (int k = 0; k < n; ++k) {
			const double u = uniforms[k];
			int id;
			if (pp->resampling_type == 1) {
				int lo = 0, hi = n - 1;
				id = (lo + hi) / 2;
				while (hi > lo) { if (cum[id] >= u) hi = id; else lo = id + 1; id = (lo + hi) / 2; }
			} else { id = 0; while (cum[id] < u) ++id; }
			if (resample_ids) resample_ids[k] = id;
			std::memcpy(&ns2[static_cast<size_t>(k) * S], states + static_cast<size_t>(id) * S, sizeof(double) * S);
			std::memcpy(&na2[static_cast<size_t>(k) * S], ars + static_cast<size_t>(id) * S, sizeof(double) * S);
			if (wts[id] >= max_wt) { max_wt = wts[id]; max_wt_id = k; }
		}
		std::memcpy(states, ns2.data(), sizeof(double) * ns2.size());
		std::memcpy(ars, na2.data(), sizeof(double) * na2.size());
	} else if (pp->resampling_type == 3) {   /* PF::residualResampling PF.cc:538-582 */
		/* normalize the weights */
		for (int k = 0; k < n; ++k) wts[k] /= cum[n - 1];
		if (wts_out) std::memcpy(wts_out, wts.data(), sizeof(double) * n);   /* (particle_wts stay normalised in the reference) */
		/* vector of particle indices; sort, with highest weight first -- std::sort(data(), data() + n - 1): the range ends one
		 * short of the last index.  std::sort leaves the order of equal weights unspecified; the stable order is one of its
		 * outcomes and the one fixed here. */
		std::vector<int> particle_idx(n);
		for (int k = 0; k < n; ++k) particle_idx[k] = k;
		std::stable_sort(particle_idx.begin(), particle_idx.begin() + (n - 1), [&](int a, int b) { return wts[a] > wts[b]; });
		vecd ns2(static_cast<size_t>(n) * S), na2(static_cast<size_t>(n) * S);
		auto put = [&](int dst, int src) {
			std::memcpy(&ns2[static_cast<size_t>(dst) * S], states + static_cast<size_t>(src) * S, sizeof(double) * S);
			std::memcpy(&na2[static_cast<size_t>(dst) * S], ars + static_cast<size_t>(src) * S, sizeof(double) * S);
			if (resample_ids) resample_ids[dst] = src;
		};
		int particles_found = 0;
		for (int particle_id = 0; particle_id < n; ++particle_id) {
			const int resample_id = particle_idx[particle_id];
			const int particle_copies = static_cast<int>(std::round(wts[resample_id] * n));
			for (int copy_id = 0; copy_id < particle_copies; ++copy_id) {
				put(particles_found, resample_id);
				if (++particles_found == n) break;
			}
			if (particles_found == n) break;
		}
		for (int particle_id = particles_found; particle_id < n; ++particle_id) put(particle_id, particle_idx[0]);   /* duplicate the highest weight */
		std::memcpy(states, ns2.data(), sizeof(double) * ns2.size());
		std::memcpy(ars, na2.data(), sizeof(double) * na2.size());
		max_wt_id = particle_idx[0];   /* (an index of the OLD set, used with the new one: PF.cc:581, kept) */
	} else if (pp->resampling_type != 0) return -2;
	/* mean type: PF.cc:421-437 */
	if (pp->mean_type == 0) ssm->set_state(states + static_cast<size_t>(max_wt_id) * S);
	else if (pp->mean_type == 1) {   /* ProjectiveBase::estimateMeanOfSamples :311-317 */
		vecd m(S, 0.0);
		for (int k = 0; k < n; ++k) for (int s2 = 0; s2 < S; ++s2) m[s2] += (states[static_cast<size_t>(k) * S + s2] - m[s2]) / (k + 1);
		ssm->set_state(m.data());
	} else {   /* updateMeanCorners :607-614, then setCorners */
		vecd mc(8, 0.0);
		for (int k = 0; k < n; ++k) {
			ssm->set_state(states + static_cast<size_t>(k) * S);
			for (int q = 0; q < 8; ++q) mc[q] += (ssm->curr_corners[q] - mc[q]) / (k + 1);
		}
		ssm->set_corners(mc.data());
	}
	if (max_wt_id_out) *max_wt_id_out = max_wt_id;
	return 0;
}

/* ---- GridTracker (SM/src/GridTracker.cc) ---- */
void mtfo_grid_res(const mtfo_grid_params *gp, int *resx, int *resy) { mtfo_grid::update_res(*gp, *resx, *resy); }
mtfo_grid *mtfo_grid_create(const mtfo_grid_params *gp, mtfo_ssm *grid_ssm, mtfo_tracker **trackers, int n_trackers) {
	int rx, ry; mtfo_grid::update_res(*gp, rx, ry);
	if (!grid_ssm || grid_ssm->resx != rx || grid_ssm->resy != ry) return nullptr;            /* GridTracker.cc:130-134 */
	if (n_trackers != 0 && n_trackers != gp->grid_size_x * gp->grid_size_y) return nullptr;  /* :124-129 (0: layout only) */
	return new mtfo_grid(*gp, grid_ssm, trackers, n_trackers);
}
void mtfo_grid_destroy(mtfo_grid *g) { delete g; }
void mtfo_grid_set_estimator(mtfo_grid *g, mtfo_grid_estimator est, void *user) { g->est = est; g->est_user = user; }
void mtfo_grid_set_image(mtfo_grid *g, const float *img, int h, int w) { g->set_image(img, h, w); }
int mtfo_grid_fb_mask(int n, const float *prev_pts, const float *curr_pts, const float *fb_prev_pts, double fb_err_thresh, int n_model_pts,
	unsigned char *fb_err_mask, float *prev_masked, float *curr_masked) {
	std::vector<float> a, b;
	mtfo_grid::fb_mask(n, prev_pts, curr_pts, fb_prev_pts, fb_err_thresh, n_model_pts, fb_err_mask, a, b);
	std::copy(a.begin(), a.end(), prev_masked); std::copy(b.begin(), b.end(), curr_masked);
	return static_cast<int>(a.size() / 2);
}
void mtfo_grid_initialize(mtfo_grid *g, const double *corners) { g->initialize(corners); }
int mtfo_grid_update(mtfo_grid *g) { return g->update(); }
void mtfo_grid_set_region(mtfo_grid *g, const double *corners) { g->set_region(corners); }
void mtfo_grid_get(const mtfo_grid *g, int what, double *dst) {
	switch (what) {
	case 0: std::copy(g->region.begin(), g->region.end(), dst); break;
	case 1: std::copy(g->patch_corners.begin(), g->patch_corners.end(), dst); break;
	case 2: for (size_t i = 0; i < g->prev_pts.size(); ++i) dst[i] = g->prev_pts[i]; break;
	case 3: for (size_t i = 0; i < g->curr_pts.size(); ++i) dst[i] = g->curr_pts[i]; break;
	case 4: std::copy(g->ssm_update.begin(), g->ssm_update.end(), dst); break;
	case 5: for (size_t i = 0; i < g->fb_prev_pts.size(); ++i) dst[i] = g->fb_prev_pts[i]; break;
	case 6: for (size_t i = 0; i < g->fb_err_mask.size(); ++i) dst[i] = g->fb_err_mask[i]; break;
	case 7: std::copy(g->fb_locations.begin(), g->fb_locations.end(), dst); break;
	case 8: std::copy(g->fb_regions.begin(), g->fb_regions.end(), dst); break;
	case 9: {
		const size_t c = g->est_prev.size() / 2;
		dst[0] = static_cast<double>(c);
		for (size_t i = 0; i < 2 * c; ++i) { dst[1 + i] = g->est_prev[i]; dst[1 + 2 * c + i] = g->est_curr[i]; }
		break;
	}
	default: break;
	}
}

} // extern "C"
