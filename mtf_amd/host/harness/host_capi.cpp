/*
 * host_capi.cpp -- a small C wrapper around the C++ host layer (mtf::hip::HipAM / HipSSM driven by
 * mtf::nt::ESM / FCLK / ICLK), the equivalent of the reference's pyMTF create / setRegion / getRegion
 * (Examples/cpp/pyMTF.cc:35-62), so that the test-suite can drive the C++ objects.  HARNESS (libmtfharness.so): it constructs the
 * product's adapters and device drivers (libmtfhost.so) AND the restated reference callers that live next to it.
 */
#include <chrono>
#include <cstring>
#include <memory>
#include <string>

#include "../HipModels.h"
#include "../DeviceLK.h"
#include "../DevicePF.h"
#include "../DeviceGrid.h"
#include "SearchMethods.h"
#include "PF.h"
#include "TemplatedSM.h"

using namespace mtf;

struct mtfhost_tracker {
	std::shared_ptr<hip::HipPair> pair;
	std::shared_ptr<hip::HipAM> am;
	std::shared_ptr<hip::HipSSM> ssm;
	std::unique_ptr<nt::SearchMethod> sm;
};

static thread_local std::string g_err;

extern "C" {

const char *mtfhost_last_error(void) { return g_err.c_str(); }

mtfhost_tracker *mtfhost_create(int sm, int am, int ssm, int resx, int resy, int max_iters, double epsilon,
	int jac_type, int hess_type, int chained_warp, int leven_marq, double lm_delta_init, double lm_delta_update,
	int device, int sec_ord_hess, int n_channels) {
	try {
		std::unique_ptr<mtfhost_tracker> t(new mtfhost_tracker());   /* (a constructor below may throw: nothing leaks) */
		t->pair = std::make_shared<hip::HipPair>(am, ssm, resx, resy, 1e-8, 1.0, 8, 10.0, 0, device, nullptr, n_channels);
		t->am = std::make_shared<hip::HipAM>(t->pair);
		t->ssm = std::make_shared<hip::HipSSM>(t->pair);
		nt::SMParams p;
		p.max_iters = max_iters; p.epsilon = epsilon; p.jac_type = jac_type; p.hess_type = hess_type;
		p.chained_warp = chained_warp != 0; p.leven_marq = leven_marq != 0;
		p.lm_delta_init = lm_delta_init; p.lm_delta_update = lm_delta_update;
		p.sec_ord_hess = sec_ord_hess != 0;
		/* sm + 16: the same search method as ONE call per update() (mtf::hip::LK: the loop runs on the device) */
		if (sm >= 16) t->sm.reset(new hip::LK(sm - 16, t->am, t->ssm, p));
		else if (sm == MTFHIP_SM_ESM) t->sm.reset(new nt::ESM(t->am, t->ssm, p));
		else if (sm == MTFHIP_SM_FCLK) t->sm.reset(new nt::FCLK(t->am, t->ssm, p));
		else if (sm == MTFHIP_SM_ICLK) t->sm.reset(new nt::ICLK(t->am, t->ssm, p));
		else { g_err = "unknown search method"; return nullptr; }
		return t.release();
	} catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void mtfhost_destroy(mtfhost_tracker *t) { delete t; }
/* ESM / FC / IC_ENABLE_LEARNING + the AM's learning_rate: am->updateModel(ssm->getPts()) at the end of every update() */
int mtfhost_set_learning(mtfhost_tracker *t, int enable, double learning_rate) {
	if (!t) { g_err = "NULL tracker"; return -1; }
	t->sm->setLearning(enable != 0);
	t->am->setLearningRate(learning_rate);
	return 0;
}

static int guarded(mtfhost_tracker *t, void (*fn)(mtfhost_tracker *, const void *, void *), const void *in, void *out) {
	try { fn(t, in, out); return 0; }
	catch (const utils::Exception &e) { g_err = std::string(e.type()) + ": " + e.what(); return -1; }
	catch (const std::exception &e) { g_err = e.what(); return -2; }
}
int mtfhost_set_image(mtfhost_tracker *t, const float *img, int rows, int cols, int step) {
	ImageView v{img, rows, cols, step, (int)t->am->getNChannels()};
	return guarded(t, [](mtfhost_tracker *tt, const void *in, void *) { tt->sm->setImage(*(const ImageView *)in); }, &v, nullptr);
}
int mtfhost_initialize(mtfhost_tracker *t, const double *corners) {
	return guarded(t, [](mtfhost_tracker *tt, const void *in, void *) {
		CornersT c; std::memcpy(c.data(), in, sizeof(double) * 8); tt->sm->initialize(c); }, corners, nullptr);
}
int mtfhost_set_region(mtfhost_tracker *t, const double *corners) {
	return guarded(t, [](mtfhost_tracker *tt, const void *in, void *) {
		CornersT c; std::memcpy(c.data(), in, sizeof(double) * 8); tt->sm->setRegion(c); }, corners, nullptr);
}
int mtfhost_update(mtfhost_tracker *t, int *iters_done) {
	return guarded(t, [](mtfhost_tracker *tt, const void *, void *out) {
		tt->sm->update(); if (out) *(int *)out = tt->sm->getItersDone(); }, nullptr, iters_done);
}
int mtfhost_get_region(mtfhost_tracker *t, double *corners) {
	return guarded(t, [](mtfhost_tracker *tt, const void *, void *out) {
		std::memcpy(out, tt->sm->getRegion().data(), sizeof(double) * 8); }, nullptr, corners);
}
/* the SSM's host-side algebra through the StateSpaceModel virtuals (tests): what = 0 getIdentityWarp(out S), 1 composeWarps(out S;
 * a, b states), 2 estimateWarpFromCorners(out S; a, b corners 2 x 4), 3 applyWarpToCorners(out 8; a corners, b state),
 * 4 additiveUpdate(a) then getState(out S) */
int mtfhost_ssm_algebra(mtfhost_tracker *t, int what, const double *a, const double *b, double *out) {
	try {
		StateSpaceModel *ssm = t->ssm.get();
		const int S = (int)ssm->getStateSize();
		VectorXd r(S), va(S), vb(S);
		CornersT ca, cb;
		if (what == 1 || what == 4) std::memcpy(va.data(), a, sizeof(double) * S);
		if (what == 1 || what == 3) std::memcpy(vb.data(), b, sizeof(double) * S);
		if (what == 2 || what == 3) std::memcpy(ca.data(), a, sizeof(double) * 8);
		if (what == 2) std::memcpy(cb.data(), b, sizeof(double) * 8);
		switch (what) {
		case 0: ssm->getIdentityWarp(r); break;
		case 1: ssm->composeWarps(r, va, vb); break;
		case 2: ssm->estimateWarpFromCorners(r, ca, cb); break;
		case 3: { CornersT o; ssm->applyWarpToCorners(o, ca, vb); std::memcpy(out, o.data(), sizeof(double) * 8); return 0; }
		case 4: ssm->additiveUpdate(va); r = ssm->getState(); break;
		default: g_err = "mtfhost_ssm_algebra: unknown selector"; return -1;
		}
		std::memcpy(out, r.data(), sizeof(double) * S);
		return 0;
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}
/* AppearanceModel::updateDistFeat(double *) through the base-class pointer (tests) */
int mtfhost_dist_feat(mtfhost_tracker *t, double *feat, int *size) {
	try {
		AppearanceModel *am = t->am.get();
		if (size) *size = (int)am->getDistFeatSize();
		if (feat) { am->initializeDistFeat(); am->updateDistFeat(); std::memcpy(feat, am->getDistFeat(), sizeof(double) * am->getDistFeatSize()); }
		return 0;
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}
/* the particle filter: device = 1 -> mtf::hip::PF (mtfhip_pf_*), 0 -> mtf::nt::PF over the AM / SSM virtuals (one C-ABI round trip
 * per particle: the literal drop-in of SM/src/NT/PF.cc) */
mtfhost_tracker *mtfhost_pf_create(int device_filter, int am, int ssm, int resx, int resy, int n_particles, int max_iters, double epsilon,
	int dynamic_model, int update_type, int likelihood_func, int resampling_type, int mean_type, int corner_based_sampling,
	const double *ssm_sigma, double likelihood_alpha, unsigned long long seed, int device) {
	try {
		std::unique_ptr<mtfhost_tracker> t(new mtfhost_tracker());   /* (a constructor below may throw: nothing leaks) */
		t->pair = std::make_shared<hip::HipPair>(am, ssm, resx, resy, 1e-8, likelihood_alpha, 8, 10.0, 0, device, nullptr, 1);
		t->am = std::make_shared<hip::HipAM>(t->pair);
		t->ssm = std::make_shared<hip::HipSSM>(t->pair);
		t->ssm->setCornerBasedSampling(corner_based_sampling != 0);
		PFParams p;
		p.n_particles = n_particles; p.max_iters = max_iters; p.epsilon = epsilon;
		p.dynamic_model = (PFParams::DynamicModel)dynamic_model; p.update_type = (PFParams::UpdateType)update_type;
		p.likelihood_func = (PFParams::LikelihoodFunc)likelihood_func; p.resampling_type = (PFParams::ResamplingType)resampling_type;
		p.mean_type = (PFParams::MeanType)mean_type; p.seed = seed;
		p.ssm_sigma.assign(ssm_sigma, ssm_sigma + t->pair->S);
		if (device_filter) t->sm.reset(new hip::PF(t->am, t->ssm, p));
		else t->sm.reset(new nt::PF(t->am, t->ssm, p));
		return t.release();
	} catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
/* the same with the options of the shipped configuration: n_distr >= 1 sampler distributions (rows of 8), adaptive resampling,
 * jacobian_as_sigma (PFParams.h; Config/modules.cfg:157-176) */
/* pix_sigma != NULL: n_distr values, the sigma rows are then estimated at initialize() (PFParams.cc:105-116) */
mtfhost_tracker *mtfhost_pf_create_pix(int device_filter, int am, int ssm, int resx, int resy, int n_particles, int max_iters, double epsilon,
	int dynamic_model, int update_type, int likelihood_func, int resampling_type, int mean_type, int corner_based_sampling,
	int n_distr, const double *sigma_rows, const double *mean_rows, int update_distr_wts, double min_distr_wt, double adaptive_resampling_thresh,
	int jacobian_as_sigma, double likelihood_alpha, unsigned long long seed, int device, const double *pix_sigma);
mtfhost_tracker *mtfhost_pf_create_ex(int device_filter, int am, int ssm, int resx, int resy, int n_particles, int max_iters, double epsilon,
	int dynamic_model, int update_type, int likelihood_func, int resampling_type, int mean_type, int corner_based_sampling,
	int n_distr, const double *sigma_rows, const double *mean_rows, int update_distr_wts, double min_distr_wt, double adaptive_resampling_thresh,
	int jacobian_as_sigma, double likelihood_alpha, unsigned long long seed, int device) {
	return mtfhost_pf_create_pix(device_filter, am, ssm, resx, resy, n_particles, max_iters, epsilon, dynamic_model, update_type, likelihood_func,
		resampling_type, mean_type, corner_based_sampling, n_distr, sigma_rows, mean_rows, update_distr_wts, min_distr_wt, adaptive_resampling_thresh,
		jacobian_as_sigma, likelihood_alpha, seed, device, nullptr);
}
mtfhost_tracker *mtfhost_pf_create_pix(int device_filter, int am, int ssm, int resx, int resy, int n_particles, int max_iters, double epsilon,
	int dynamic_model, int update_type, int likelihood_func, int resampling_type, int mean_type, int corner_based_sampling,
	int n_distr, const double *sigma_rows, const double *mean_rows, int update_distr_wts, double min_distr_wt, double adaptive_resampling_thresh,
	int jacobian_as_sigma, double likelihood_alpha, unsigned long long seed, int device, const double *pix_sigma) {
	try {
		if (n_distr < 1) throw utils::InvalidArgument("mtfhost_pf_create_ex: n_distr must be positive");
		std::unique_ptr<mtfhost_tracker> t(new mtfhost_tracker());   /* (a constructor below may throw: nothing leaks) */
		t->pair = std::make_shared<hip::HipPair>(am, ssm, resx, resy, 1e-8, likelihood_alpha, 8, 10.0, 0, device, nullptr, 1);
		t->am = std::make_shared<hip::HipAM>(t->pair);
		t->ssm = std::make_shared<hip::HipSSM>(t->pair);
		t->ssm->setCornerBasedSampling(corner_based_sampling != 0);
		PFParams p;
		p.n_particles = n_particles; p.max_iters = max_iters; p.epsilon = epsilon;
		p.dynamic_model = (PFParams::DynamicModel)dynamic_model; p.update_type = (PFParams::UpdateType)update_type;
		p.likelihood_func = (PFParams::LikelihoodFunc)likelihood_func; p.resampling_type = (PFParams::ResamplingType)resampling_type;
		p.mean_type = (PFParams::MeanType)mean_type; p.seed = seed;
		const int S = t->pair->S;
		if (pix_sigma) p.pix_sigma.assign(pix_sigma, pix_sigma + n_distr);   /* (ssm_sigma is then not used: left empty, PFParams.h) */
		else {
			p.ssm_sigma.assign(sigma_rows, sigma_rows + S);
			for (int i = 1; i < n_distr; ++i) p.more_sigma.emplace_back(sigma_rows + 8 * i, sigma_rows + 8 * i + S);
		}
		p.ssm_mean.assign(mean_rows, mean_rows + S);
		for (int i = 1; i < n_distr; ++i) p.more_mean.emplace_back(mean_rows + 8 * i, mean_rows + 8 * i + S);
		p.update_distr_wts = update_distr_wts != 0; p.min_distr_wt = min_distr_wt;
		p.adaptive_resampling_thresh = adaptive_resampling_thresh; p.jacobian_as_sigma = jacobian_as_sigma != 0;
		if (device_filter) t->sm.reset(new hip::PF(t->am, t->ssm, p));
		else t->sm.reset(new nt::PF(t->am, t->ssm, p));
		return t.release();
	} catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
/* StateSpaceModel sampler virtuals through the base class (tests): n draws of compositionalRandomWalk from the current state */
int mtfhost_ssm_random_walk(mtfhost_tracker *t, unsigned long long seed, int n, const double *sigma, double *out) {
	try {
		StateSpaceModel *ssm = t->ssm.get();
		const int S = (int)ssm->getStateSize();
		VectorXd sg(S), mn(S), st(S);
		for (int s = 0; s < S; ++s) sg[s] = sigma[s];
		ssm->initializeSampler(sg, mn);
		t->ssm->setSamplerSeed(seed);
		const VectorXd base = ssm->getState();
		for (int k = 0; k < n; ++k) { ssm->compositionalRandomWalk(st, base); std::memcpy(out + (size_t)k * S, st.data(), sizeof(double) * S); }
		return 0;
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}
/* A reader that is not an adapter: compositionalUpdate(dp) and then the BYTES behind getPts() through the base class, with or
 * without eager getters (HipPair::eager_getters); out = 2 x N, x,y interleaved */
int mtfhost_ssm_pts_after_update(mtfhost_tracker *t, const double *dp, int eager, double *out) {
	try {
		t->pair->setEagerGetters(eager != 0);
		StateSpaceModel *ssm = t->ssm.get();
		VectorXd v((int)ssm->getStateSize());
		std::memcpy(v.data(), dp, sizeof(double) * v.size());
		ssm->compositionalUpdate(v);
		const PtsT &pts = ssm->getPts();
		std::memcpy(out, pts.data(), sizeof(double) * 2 * ssm->getNPts());
		t->pair->setEagerGetters(false);
		return 0;
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}
/* host-only helper exercised by the CPU tests */
int mtfhost_qr_solve(int n, const double *A_colmajor, const double *b, double *x) {
	try {
		MatrixXd A(n, n); VectorXd bb(n), xx;
		std::memcpy(A.data(), A_colmajor, sizeof(double) * n * n);
		std::memcpy(bb.data(), b, sizeof(double) * n);
		utils::colPivHouseholderQrSolve(A, bb, xx);
		std::memcpy(x, xx.data(), sizeof(double) * n);
		return 0;
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}

} // extern "C"

/* the reference's TEMPLATED search-method shape over the adapters (TemplatedSM.h): FCLK<HipAM, HipSSM> built from
 * `const AM::ParamType *` / `const SSM::ParamType *`, one initialize on frame0 and one update() on frame1; out: corners (2 x 4), *iters */
extern "C" int mtfhost_templated_fclk(int am, int ssm, int resx, int resy, int max_iters, double epsilon, int hess_type, int device,
	const float *frame0, const float *frame1, int rows, int cols, int step, const double *corners_2x4, double *out_corners_2x4, int *iters) {
	try {
		auto link = std::make_shared<hip::HipLink>();
		link->am = am; link->ssm = ssm; link->resx = resx; link->resy = resy; link->device = device;
		hip::HipAM::ParamType amp; amp.link = link;
		hip::HipSSM::ParamType ssmp; ssmp.link = link;
		templated::FCLKParams fp; fp.max_iters = max_iters; fp.epsilon = epsilon; fp.hess_type = hess_type;
		templated::FCLK<hip::HipAM, hip::HipSSM> sm(&fp, &amp, &ssmp);
		CornersT c; std::memcpy(c.data(), corners_2x4, sizeof(double) * 8);
		sm.setImage(ImageView{frame0, rows, cols, step});
		sm.initialize(c);
		sm.setImage(ImageView{frame1, rows, cols, step});
		const int n = sm.update();
		if (iters) *iters = n;
		std::memcpy(out_corners_2x4, sm.getRegion().data(), sizeof(double) * 8);
		return 0;
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* ---- mtf::hip::Grid (DeviceGrid.h): GridTracker<SSM> over one batch of patch trackers ---- */
struct mtfhost_grid { std::unique_ptr<hip::Grid> g; };
typedef void (*mtfhost_grid_estimator)(void *user, int n, const float *prev_pts, const float *curr_pts, double *ssm_update);
extern "C" {
mtfhost_grid *mtfhost_grid_create(int grid_size_x, int grid_size_y, int patch_size_x, int patch_size_y, int reset_at_each_frame, int dyn_patch_size,
	int patch_centroid_inside, int patch_sm, int patch_am, int patch_ssm, int grid_ssm, int max_iters, double epsilon, int hess_type, int leven_marq, int device,
	double fb_err_thresh, int fb_reinit, int n_model_pts) {
	try {
		GridTrackerParams gp;
		gp.fb_err_thresh = fb_err_thresh; gp.fb_reinit = fb_reinit != 0; gp.n_model_pts = n_model_pts;
		gp.grid_size_x = grid_size_x; gp.grid_size_y = grid_size_y; gp.patch_size_x = patch_size_x; gp.patch_size_y = patch_size_y;
		gp.reset_at_each_frame = reset_at_each_frame; gp.dyn_patch_size = dyn_patch_size != 0; gp.patch_centroid_inside = patch_centroid_inside != 0;
		nt::SMParams p;
		p.max_iters = max_iters; p.epsilon = epsilon; p.hess_type = hess_type; p.leven_marq = leven_marq != 0;
		std::unique_ptr<mtfhost_grid> h(new mtfhost_grid());
		h->g.reset(new hip::Grid(gp, patch_sm, patch_am, patch_ssm, p, grid_ssm, device));
		return h.release();
	} catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void mtfhost_grid_destroy(mtfhost_grid *h) { delete h; }
int mtfhost_grid_set_estimator(mtfhost_grid *h, mtfhost_grid_estimator est, void *user) {
	if (!h) { g_err = "NULL grid"; return -1; }
	if (!est) return 0;
	h->g->setEstimator([est, user](VectorXd &u, const std::vector<GridPt> &a, const std::vector<GridPt> &c) {
		std::vector<float> fa(2 * a.size()), fc(2 * c.size());
		for (size_t i = 0; i < a.size(); ++i) { fa[2 * i] = a[i].x; fa[2 * i + 1] = a[i].y; fc[2 * i] = c[i].x; fc[2 * i + 1] = c[i].y; }
		est(user, (int)a.size(), fa.data(), fc.data(), u.data());
	});
	return 0;
}
/* what: 0 setImage(img) 1 initialize(corners) 2 update() 3 setRegion(corners) */
int mtfhost_grid_call(mtfhost_grid *h, int what, const double *corners, const float *img, int rows, int cols, int step) {
	try {
		CornersT c;
		if (corners) std::memcpy(c.data(), corners, sizeof(double) * 8);
		switch (what) {
		case 0: h->g->setImage(ImageView{img, rows, cols, step}); break;
		case 1: h->g->initialize(c); break;
		case 2: h->g->update(); break;
		case 3: h->g->setRegion(c); break;
		default: g_err = "mtfhost_grid_call: unknown selector"; return -1;
		}
		return 0;
	} catch (const utils::Exception &e) { g_err = std::string(e.type()) + ": " + e.what(); return -1; }
	catch (const std::exception &e) { g_err = e.what(); return -2; }
}
/* what: 0 region (8) 1 patch corners of the last reset (n x 8) 2 prev_pts (n x 2) 3 curr_pts (n x 2) 4 ssm_update (S) 5 patch iteration
 * counts (n, as doubles) 6 the patch trackers' regions after the last update (n x 8) 7 fb_prev_pts (n x 2) 8 fb_err_mask (n, 0 / 1) */
int mtfhost_grid_get(mtfhost_grid *h, int what, double *dst) {
	try {
		hip::Grid &g = *h->g;
		switch (what) {
		case 0: std::memcpy(dst, g.getRegion().data(), sizeof(double) * 8); break;
		case 1: std::memcpy(dst, g.getPatchCorners().data(), sizeof(double) * g.getPatchCorners().size()); break;
		case 2: for (size_t i = 0; i < g.getPrevPts().size(); ++i) { dst[2 * i] = g.getPrevPts()[i].x; dst[2 * i + 1] = g.getPrevPts()[i].y; } break;
		case 3: for (size_t i = 0; i < g.getCurrPts().size(); ++i) { dst[2 * i] = g.getCurrPts()[i].x; dst[2 * i + 1] = g.getCurrPts()[i].y; } break;
		case 4: std::memcpy(dst, g.getSSMUpdate().data(), sizeof(double) * g.getSSMUpdate().size()); break;
		case 5: for (size_t i = 0; i < g.getPatchIters().size(); ++i) dst[i] = g.getPatchIters()[i]; break;
		case 6: std::memcpy(dst, g.getPatchRegions().data(), sizeof(double) * g.getPatchRegions().size()); break;
		case 7: for (size_t i = 0; i < g.getFbPrevPts().size(); ++i) { dst[2 * i] = g.getFbPrevPts()[i].x; dst[2 * i + 1] = g.getFbPrevPts()[i].y; } break;
		case 8: for (size_t i = 0; i < g.getFbErrMask().size(); ++i) dst[i] = g.getFbErrMask()[i]; break;
		default: g_err = "mtfhost_grid_get: unknown selector"; return -1;
		}
		return 0;
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}
}

/* the same for the templated ESM<HipAM, HipSSM> (sm = 0) and ICLK<HipAM, HipSSM> (sm = 2): initialize on frame0, update() on frame1; then --
 * move != NULL -- setRegion(the result shifted by move (dx, dy)) and a second update() on frame1.  out: corners after the first update
 * (2 x 4), out2: after the second (or NULL); iters: two counts */
extern "C" int mtfhost_templated_sm(int sm, int am, int ssm, int resx, int resy, int max_iters, double epsilon, int jac_type, int hess_type, int leven_marq,
	int device, const float *frame0, const float *frame1, int rows, int cols, int step, const double *corners_2x4, const double *move,
	double *out_corners_2x4, double *out2_corners_2x4, int *iters) {
	try {
		auto link = std::make_shared<hip::HipLink>();
		link->am = am; link->ssm = ssm; link->resx = resx; link->resy = resy; link->device = device;
		hip::HipAM::ParamType amp; amp.link = link;
		hip::HipSSM::ParamType ssmp; ssmp.link = link;
		CornersT c; std::memcpy(c.data(), corners_2x4, sizeof(double) * 8);
		auto run = [&](auto &tracker) {
			tracker.setImage(ImageView{frame0, rows, cols, step});
			tracker.initialize(c);
			tracker.setImage(ImageView{frame1, rows, cols, step});
			iters[0] = tracker.update();
			std::memcpy(out_corners_2x4, tracker.getRegion().data(), sizeof(double) * 8);
			if (move && out2_corners_2x4) {
				CornersT m = tracker.getRegion();
				for (int q = 0; q < 4; ++q) { m(0, q) += move[0]; m(1, q) += move[1]; }
				tracker.setRegion(m);
				iters[1] = tracker.update();
				std::memcpy(out2_corners_2x4, tracker.getRegion().data(), sizeof(double) * 8);
			}
		};
		if (sm == MTFHIP_SM_ESM) {
			templated::ESMParams p; p.max_iters = max_iters; p.epsilon = epsilon; p.jac_type = jac_type; p.hess_type = hess_type; p.leven_marq = leven_marq != 0;
			templated::ESM<hip::HipAM, hip::HipSSM> t(&p, &amp, &ssmp);
			run(t);
		} else if (sm == MTFHIP_SM_ICLK) {
			templated::ICLKParams p; p.max_iters = max_iters; p.epsilon = epsilon; p.hess_type = hess_type; p.leven_marq = leven_marq != 0;
			templated::ICLK<hip::HipAM, hip::HipSSM> t(&p, &amp, &ssmp);
			run(t);
		} else { g_err = "mtfhost_templated_sm: ESM (0) or ICLK (2)"; return -1; }
		return 0;
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* frame loops timed on the C++ side (no Python in the timed path; bench.py --workload grid reports them next to its own):
 * what = 0: n_frames calls of mtfhip_grid_frame(region) -- layout + setRegion + update of every patch in one C-ABI call
 * what = 1: n_frames of hip::Grid::update() -- the same launch + the estimator + the reset the parameters ask for (GridTracker.cc:247-285)
 * out: microseconds per frame (wall clock, steady_clock around the whole loop, a warm-up of n_frames / 10 + 5 frames in front) */
extern "C" int mtfhost_grid_bench(mtfhost_grid *h, int what, int n_frames, const double *region_corners, double *us_per_frame) {
	try {
		hip::Grid &g = *h->g;
		mtfhip_grid_desc gd;
		{   /* the driver's own description, from its parameters */
			CornersT c; std::memcpy(c.data(), region_corners, sizeof(double) * 8);
			g.setRegion(c);
		}
		const int B = mtfhip_batch_n_targets(g.batch());
		std::vector<int> it(B); std::vector<double> cr(8 * (size_t)B); std::vector<float> cen(2 * (size_t)B);
		gd = g.gridDesc();
		auto frame = [&]() {
			if (what == 0) hip::HipPair::check(mtfhip_grid_frame(g.batch(), &g.desc(), &gd, region_corners, it.data(), cr.data(), cen.data()));
			else g.update();
		};
		for (int k = 0; k < n_frames / 10 + 5; ++k) frame();
		const auto t0 = std::chrono::steady_clock::now();
		for (int k = 0; k < n_frames; ++k) frame();
		*us_per_frame = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n_frames;
		return 0;
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}

/* the video loop as runMTF drives a tracker (Examples/cpp/runMTF.cc:650-720): per frame setImage(next frame) -- the caller's buffer is
 * overwritten in place, so the device copy is refreshed -- then update().  Two frames alternate (the motion A -> B,
 * then B -> A: the region oscillates and the loop is stationary).  us[0] = per frame in update() alone, us[1] = per frame in setImage
 * (the host-to-device copy of the frame), both over exactly n_frames frames after n_frames / 10 + 5 untimed ones. */
extern "C" int mtfhost_grid_bench_video(mtfhost_grid *h, int n_frames, const float *frame_a, const float *frame_b, int rows, int cols, int step, double *us) {
	try {
		hip::Grid &g = *h->g;
		double t_upd = 0, t_img = 0;
		for (int k = -(n_frames / 10 + 5); k < n_frames; ++k) {
			const auto t0 = std::chrono::steady_clock::now();
			g.setImage(ImageView{(k & 1) ? frame_a : frame_b, rows, cols, step});
			const auto t1 = std::chrono::steady_clock::now();
			g.update();
			const auto t2 = std::chrono::steady_clock::now();
			if (k >= 0) {
				t_img += std::chrono::duration<double, std::micro>(t1 - t0).count();
				t_upd += std::chrono::duration<double, std::micro>(t2 - t1).count();
			}
		}
		us[0] = t_upd / n_frames; us[1] = t_img / n_frames;
		return 0;
	} catch (const std::exception &e) { g_err = e.what(); return -1; }
}
