/* DeviceGrid.cpp -- see DeviceGrid.h */
#include "DeviceGrid.h"

#include <cmath>
#include <cstring>

namespace mtf {
namespace hip {

Grid::Grid(const GridTrackerParams &gp, int patch_sm, int patch_am, int patch_ssm, const nt::SMParams &pp, int _grid_ssm, int device, void *stream) :
	params(gp), gd(gp.desc()), n(gp.grid_size_x * gp.grid_size_y), grid_ssm(_grid_ssm) {
	if (gp.grid_size_x <= 0 || gp.grid_size_y <= 0 || gp.patch_size_x <= 0 || gp.patch_size_y <= 0)
		throw utils::InvalidArgument("GridTracker :: grid and patch sizes must be positive");
	if (patch_sm != MTFHIP_SM_ESM && patch_sm != MTFHIP_SM_FCLK && patch_sm != MTFHIP_SM_ICLK)
		throw utils::InvalidArgument("GridTracker :: unknown patch search method");
	if (grid_ssm != MTFHIP_SSM_HOMOGRAPHY && grid_ssm != MTFHIP_SSM_AFFINE) throw utils::InvalidArgument("GridTracker :: unknown grid SSM");
	reinit_at_each_frame = gp.reset_at_each_frame == 1;   /* GridTracker.cc:136 */
	std::memset(&d, 0, sizeof(d));
	d.sm = patch_sm;
	d.jac_type = pp.jac_type;
	d.hess_type = pp.hess_type >= 0 ? pp.hess_type : (patch_sm == MTFHIP_SM_ESM ? 2 : (patch_sm == MTFHIP_SM_FCLK ? 1 : 0));
	d.chained_warp = pp.chained_warp ? 1 : 0;
	d.materialize = 0;
	d.max_iters = pp.max_iters;
	d.epsilon = pp.epsilon;
	d.leven_marq = pp.leven_marq ? 1 : 0;
	d.lm_delta_init = pp.lm_delta_init;
	d.lm_delta_update = pp.lm_delta_update;
	d.sec_ord_hess = pp.sec_ord_hess ? 1 : 0;
	HipPair::check(mtfhip_ctx_create(device, stream, &ctx));
	mtfhip_patch_desc pd;
	std::memset(&pd, 0, sizeof(pd));
	pd.am = patch_am; pd.ssm = patch_ssm; pd.resx = gp.patch_size_x; pd.resy = gp.patch_size_y;   /* mtf.h:782-788 */
	pd.grad_eps = 1e-8; pd.likelihood_alpha = 1.0; pd.mi_n_bins = 8; pd.mi_pre_seed = 10; pd.mi_partition_of_unity = 0; pd.hess_eps = 1.0; pd.n_channels = 1;
	const int rc = mtfhip_batch_create(ctx, &pd, n, &b);
	if (rc != MTFHIP_OK) { mtfhip_ctx_destroy(ctx); ctx = nullptr; HipPair::check(rc); }
	prev_pts.resize(n); curr_pts.resize(n); cen.resize(2 * (size_t)n); n_iters.assign(n, 0);   /* :153-154 */
	patch_corners.assign(8 * (size_t)n, 0.0); patch_regions.assign(8 * (size_t)n, 0.0);
	ssm_update.resize(grid_ssm == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6);
	if (gp.fb_err_thresh > 0) {   /* :186-190 */
		enable_fb_err_est = true;
		fbd = mtfhip_grid_fb_desc{gp.fb_err_thresh, gp.fb_reinit ? 1 : 0, gp.n_model_pts};
		fb_prev_pts.resize(n); fb_err_mask.assign(n, 0);
		fb_cen.resize(2 * (size_t)n); prev_f.resize(2 * (size_t)n); prev_masked.resize(2 * (size_t)n); curr_masked.resize(2 * (size_t)n);
	}
	const int gs = grid_ssm;
	estimator = [gs](VectorXd &u, const std::vector<GridPt> &a, const std::vector<GridPt> &c) { leastSquaresFit(gs, u, a, c); };
}

Grid::~Grid() {
	if (b) mtfhip_batch_destroy(b);
	if (ctx) mtfhip_ctx_destroy(ctx);
}

void Grid::setImage(const ImageView &img) {   /* :205-231: every patch tracker shares the frame */
	if (img.channels != 1) throw utils::InvalidArgument("GridTracker :: single-channel patch trackers");
	HipPair::check(mtfhip_image_upload(ctx, img.data, img.rows, img.cols, img.step));
}

/* GridTracker::resetTrackers :345-392 */
void Grid::resetTrackers(bool reinit) {
	if (reinit || !have_template) {
		HipPair::check(mtfhip_grid_reset(b, &d, &gd, region.data(), 1, patch_corners.data(), cen.data()));
		have_pending = false;
	} else {
		/* setRegion only: it rides in the next frame's launch (mtfhip_grid_frame with a region) */
		HipPair::check(mtfhip_grid_layout(&gd, region.data(), nullptr, patch_corners.data()));
		for (int k = 0; k < n; ++k) {
			const double *c = &patch_corners[8 * (size_t)k];
			cen[2 * k] = static_cast<float>((c[0] + c[2] + c[4] + c[6]) / 4.0);
			cen[2 * k + 1] = static_cast<float>((c[1] + c[3] + c[5] + c[7]) / 4.0);
		}
		pending_region = region; have_pending = true;
	}
	have_template = true;
	for (int k = 0; k < n; ++k) { prev_pts[k].x = cen[2 * k]; prev_pts[k].y = cen[2 * k + 1]; }   /* :387 */
}

void Grid::initialize(const CornersT &corners) {
	region = corners;              /* ssm.initialize(corners) :234 */
	have_template = false;
	resetTrackers(true);           /* :235 */
	curr_pts = prev_pts;           /* :236-239 */
	if (enable_fb_err_est) HipPair::check(mtfhip_image_keep_prev(ctx));   /* prev_img = curr_img.clone() :241-243 */
}

void Grid::setRegion(const CornersT &corners) {
	region = corners;              /* ssm.setCorners(corners) */
	resetTrackers(reinit_at_each_frame);
}

void Grid::update() {
	if (!have_template) throw utils::LogicError("GridTracker :: update before initialize");
	/* :254-261 every patch tracker's update() + getCentroid(curr_pts[id], getRegion()) */
	if (enable_fb_err_est) {
		/* :263-266 backwardEstimation(); prev_img = curr_img.clone() */
		for (int k = 0; k < n; ++k) { prev_f[2 * k] = prev_pts[k].x; prev_f[2 * k + 1] = prev_pts[k].y; }
		int n_masked = 0;
		HipPair::check(mtfhip_grid_frame_fb(b, &d, &gd, &fbd, have_pending ? pending_region.data() : nullptr, prev_f.data(), n_iters.data(), patch_regions.data(),
			cen.data(), fb_cen.data(), fb_err_mask.data(), prev_masked.data(), curr_masked.data(), &n_masked));
		have_pending = false;
		HipPair::check(mtfhip_image_keep_prev(ctx));
		std::vector<GridPt> pm(n_masked), cm(n_masked);
		for (int k = 0; k < n_masked; ++k) { pm[k].x = prev_masked[2 * k]; pm[k].y = prev_masked[2 * k + 1]; cm[k].x = curr_masked[2 * k]; cm[k].y = curr_masked[2 * k + 1]; }
		for (int k = 0; k < n; ++k) { curr_pts[k].x = cen[2 * k]; curr_pts[k].y = cen[2 * k + 1]; fb_prev_pts[k].x = fb_cen[2 * k]; fb_prev_pts[k].y = fb_cen[2 * k + 1]; }
		estimator(ssm_update, pm, cm);                                                     /* :334-335 */
	} else {
		HipPair::check(mtfhip_grid_frame(b, &d, &gd, have_pending ? pending_region.data() : nullptr, n_iters.data(), patch_regions.data(), cen.data()));
		have_pending = false;
		for (int k = 0; k < n; ++k) { curr_pts[k].x = cen[2 * k]; curr_pts[k].y = cen[2 * k + 1]; }
		estimator(ssm_update, prev_pts, curr_pts);                                         /* :267 */
	}
	/* :270-272 ssm.applyWarpToCorners(opt_warped_corners, ssm.getCorners(), ssm_update); ssm.setCorners(opt_warped_corners) */
	CornersT warped;
	HipPair::check(mtfhip_ssm_apply_warp_to_pts(grid_ssm, region.data(), 4, ssm_update.data(), warped.data()));
	region = warped;
	if (params.reset_at_each_frame) resetTrackers(reinit_at_each_frame);                   /* :273-274 */
	else prev_pts = curr_pts;                                                              /* :275-280 */
}

/* all-points least squares: affine = two 3-unknown normal systems; homography = normalised DLT as an 8 x 8 normal system with
 * h22 = 1 in the normalised frame (stand-in for the out-of-scope robust estimators, see DeviceGrid.h) */
static bool solveSym(int m, std::vector<double> &A, std::vector<double> &rhs) {   /* Gaussian elimination with partial pivoting, m x m row-major */
	for (int i = 0; i < m; ++i) {
		int piv = i;
		for (int r = i + 1; r < m; ++r) if (std::fabs(A[r * m + i]) > std::fabs(A[piv * m + i])) piv = r;
		if (A[piv * m + i] == 0) return false;
		if (piv != i) { for (int c = 0; c < m; ++c) std::swap(A[i * m + c], A[piv * m + c]); std::swap(rhs[i], rhs[piv]); }
		for (int r = i + 1; r < m; ++r) {
			const double f = A[r * m + i] / A[i * m + i];
			for (int c = i; c < m; ++c) A[r * m + c] -= f * A[i * m + c];
			rhs[r] -= f * rhs[i];
		}
	}
	for (int i = m - 1; i >= 0; --i) {
		double s = rhs[i];
		for (int c = i + 1; c < m; ++c) s -= A[i * m + c] * rhs[c];
		rhs[i] = s / A[i * m + i];
	}
	return true;
}
void Grid::leastSquaresFit(int ssm, VectorXd &u, const std::vector<GridPt> &a, const std::vector<GridPt> &c) {
	const int n = (int)a.size();
	u.fill(0.0);
	if (ssm == MTFHIP_SSM_AFFINE) {
		std::vector<double> N(9, 0.0), bx(3, 0.0), by(3, 0.0);
		for (int i = 0; i < n; ++i) {
			const double r[3] = {a[i].x, a[i].y, 1.0};
			for (int p = 0; p < 3; ++p) { for (int q = 0; q < 3; ++q) N[p * 3 + q] += r[p] * r[q]; bx[p] += r[p] * c[i].x; by[p] += r[p] * c[i].y; }
		}
		std::vector<double> N2 = N;
		if (!solveSym(3, N, bx) || !solveSym(3, N2, by)) throw utils::InvalidTrackerState("GridTracker :: degenerate point set");
		u(0) = bx[2]; u(1) = by[2]; u(2) = bx[0] - 1; u(3) = bx[1]; u(4) = by[0]; u(5) = by[1] - 1;   /* Affine.cc:363-368 */
		return;
	}
	auto norm = [n](const std::vector<GridPt> &p, double &mx, double &my, double &sc) {
		mx = my = 0;
		for (int i = 0; i < n; ++i) { mx += p[i].x; my += p[i].y; }
		mx /= n; my /= n;
		double dist = 0;
		for (int i = 0; i < n; ++i) dist += std::sqrt((p[i].x - mx) * (p[i].x - mx) + (p[i].y - my) * (p[i].y - my));
		dist /= n;
		sc = dist > 0 ? std::sqrt(2.0) / dist : 1.0;
	};
	double amx, amy, asc, cmx, cmy, csc;
	norm(a, amx, amy, asc); norm(c, cmx, cmy, csc);
	/* normal equations of the rows [x y 1 0 0 0 -Xx -Xy | X], [0 0 0 x y 1 -Yx -Yy | Y]: their 8 x 8 matrix is made of 23 sums over the
	 * points (the two 3 x 3 diagonal blocks are the same, the coupling blocks share their entries), ~35 flops per point instead of the
	 * 144 of the two outer products */
	double sxx = 0, sxy = 0, sx = 0, syy = 0, sy = 0;
	double Xx = 0, Xy = 0, X1 = 0, Yx = 0, Yy = 0, Y1 = 0, Xxx = 0, Xxy = 0, Xyy = 0, Yxx = 0, Yxy = 0, Yyy = 0, Rxx = 0, Rxy = 0, Ryy = 0, Rx = 0, Ry = 0;
	for (int i = 0; i < n; ++i) {
		const double x = (a[i].x - amx) * asc, y = (a[i].y - amy) * asc, X = (c[i].x - cmx) * csc, Y = (c[i].y - cmy) * csc;
		const double xx = x * x, xy = x * y, yy = y * y, R = X * X + Y * Y;
		sxx += xx; sxy += xy; sx += x; syy += yy; sy += y;
		Xx += X * x; Xy += X * y; X1 += X; Yx += Y * x; Yy += Y * y; Y1 += Y;
		Xxx += X * xx; Xxy += X * xy; Xyy += X * yy; Yxx += Y * xx; Yxy += Y * xy; Yyy += Y * yy;
		Rxx += R * xx; Rxy += R * xy; Ryy += R * yy; Rx += R * x; Ry += R * y;
	}
	const double sn = (double)n;
	std::vector<double> N(64, 0.0), rhs(8, 0.0);
	const double A3[3][3] = {{sxx, sxy, sx}, {sxy, syy, sy}, {sx, sy, sn}};
	const double BX[3][2] = {{Xxx, Xxy}, {Xxy, Xyy}, {Xx, Xy}}, BY[3][2] = {{Yxx, Yxy}, {Yxy, Yyy}, {Yx, Yy}};
	for (int p = 0; p < 3; ++p) {
		for (int q = 0; q < 3; ++q) { N[p * 8 + q] = A3[p][q]; N[(3 + p) * 8 + 3 + q] = A3[p][q]; }
		for (int k = 0; k < 2; ++k) {
			N[p * 8 + 6 + k] = -BX[p][k]; N[(6 + k) * 8 + p] = -BX[p][k];
			N[(3 + p) * 8 + 6 + k] = -BY[p][k]; N[(6 + k) * 8 + 3 + p] = -BY[p][k];
		}
	}
	N[6 * 8 + 6] = Rxx; N[6 * 8 + 7] = Rxy; N[7 * 8 + 6] = Rxy; N[7 * 8 + 7] = Ryy;
	rhs[0] = Xx; rhs[1] = Xy; rhs[2] = X1; rhs[3] = Yx; rhs[4] = Yy; rhs[5] = Y1; rhs[6] = -Rx; rhs[7] = -Ry;
	if (!solveSym(8, N, rhs)) throw utils::InvalidTrackerState("GridTracker :: degenerate point set");
	/* H = Tc^-1 * Hn * Ta */
	const double Hn[9] = {rhs[0], rhs[1], rhs[2], rhs[3], rhs[4], rhs[5], rhs[6], rhs[7], 1.0};
	const double Ta[9] = {asc, 0, -asc * amx, 0, asc, -asc * amy, 0, 0, 1};
	const double Tci[9] = {1 / csc, 0, cmx, 0, 1 / csc, cmy, 0, 0, 1};
	double T1[9], H[9];
	for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) { double s = 0; for (int k = 0; k < 3; ++k) s += Hn[r * 3 + k] * Ta[k * 3 + q]; T1[r * 3 + q] = s; }
	for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) { double s = 0; for (int k = 0; k < 3; ++k) s += Tci[r * 3 + k] * T1[k * 3 + q]; H[r * 3 + q] = s; }
	for (int i = 0; i < 9; ++i) H[i] /= H[8];
	u(0) = H[0] - 1; u(1) = H[1]; u(2) = H[2]; u(3) = H[3]; u(4) = H[4] - 1; u(5) = H[5]; u(6) = H[6]; u(7) = H[7];   /* Homography.cc:889-896 */
}

} // namespace hip
} // namespace mtf
