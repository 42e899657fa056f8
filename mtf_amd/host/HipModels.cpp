#include "HipModels.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace mtf {
namespace hip {

void HipPair::check(int rc) {
	if (rc == MTFHIP_OK) return;
	const std::string msg = mtfhip_last_error();
	switch (rc) {
	case MTFHIP_ERR_INVALID_ARG: throw utils::InvalidArgument(msg);
	case MTFHIP_ERR_NOT_IMPLEMENTED: throw utils::FunctonNotImplemented(msg);
	case MTFHIP_ERR_LOGIC: throw utils::LogicError(msg);
	default: throw utils::Exception(msg);
	}
}

HipPair::HipPair(int _am, int _ssm, int _resx, int _resy, double _grad_eps, double likelihood_alpha, int mi_n_bins,
	double mi_pre_seed, int mi_pou, int device, void *stream, int _n_channels) :
	am(_am), ssm(_ssm), resx(_resx), resy(_resy), N(_resx * _resy * (_n_channels > 1 ? _n_channels : 1)),
	S(_ssm == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6), n_pix(_resx * _resy), n_channels(_n_channels > 1 ? _n_channels : 1),
	grad_eps(_grad_eps) {
	if (resx <= 0 || resy <= 0) throw utils::InvalidArgument("ImageBase::Invalid sampling resolution provided"); /* ImageBase.cc:33-35 */
	if (const char *e = std::getenv("MTFHIP_EAGER_GETTERS")) eager_getters = e[0] == '1';
	check(mtfhip_ctx_create(device, stream, &ctx));
	mtfhip_patch_desc d{am, ssm, resx, resy, grad_eps, likelihood_alpha, mi_n_bins, mi_pre_seed, mi_pou, hess_eps, n_channels};
	int rc = mtfhip_batch_create(ctx, &d, 1, &b);
	if (rc != MTFHIP_OK) { const std::string msg = mtfhip_last_error(); mtfhip_ctx_destroy(ctx); ctx = nullptr; throw utils::Exception(msg); }
}
HipPair::~HipPair() {
	if (b) mtfhip_batch_destroy(b);
	if (ctx) mtfhip_ctx_destroy(ctx);
}
/* the SM owns its N x S Jacobians (SM/include/mtf/SM/ESM.h:40-49) and passes them by reference: the first
 * matrix an SSM writes becomes J0, the second JT, the third JM */
/* A pixel Jacobian the SSM writes is given the device matrix that matches the gradient it is built from -- J0 for the
 * template's gradient, JT for the current one (the library's fused iteration reads / writes exactly those) -- and the
 * remaining free matrix otherwise; the host-side Eigen matrix is only the key. */
int HipPair::jacobianBuffer(const MatrixXd &J, bool may_register, int preferred) {
	auto it = jac_keys.find(J.data());
	if (it != jac_keys.end()) return it->second;
	if (!may_register) throw utils::LogicError("pixel Jacobian passed to the AM was not produced by the paired SSM");
	static const int order[3] = {MTFHIP_BUF_J0, MTFHIP_BUF_JT, MTFHIP_BUF_JM};
	auto taken = [&](int id) { for (auto &kv : jac_keys) if (kv.second == id) return true; return false; };
	int id = -1;
	if (preferred >= 0 && !taken(preferred)) id = preferred;
	for (int k = 0; k < 3 && id < 0; ++k) if (!taken(order[k]) && order[k] != MTFHIP_BUF_JM) id = order[k];
	if (id < 0 && !taken(MTFHIP_BUF_JM)) id = MTFHIP_BUF_JM;
	if (id < 0) throw utils::LogicError("more than three distinct pixel Jacobians in flight");
	jac_keys[J.data()] = id;
	return id;
}

/* same for the SM-owned S^2 x N pixel Hessians (SM/include/mtf/SM/NT/ESM.h: init / curr / mean_pix_hessian) */
int HipPair::hessianBuffer(const MatrixXd &D, bool may_register) {
	auto it = hess_keys.find(D.data());
	if (it != hess_keys.end()) return it->second;
	if (!may_register) throw utils::LogicError("pixel Hessian passed to the AM was not produced by the paired SSM");
	static const int order[3] = {MTFHIP_BUF_D2I0_DP2, MTFHIP_BUF_D2IT_DP2, MTFHIP_BUF_D2IM_DP2};
	if (next_hess >= 3) throw utils::LogicError("more than three distinct pixel Hessians in flight");
	int id = order[next_hess++];
	hess_keys[D.data()] = id;
	return id;
}

/* ------------------------------------------------------------------ AM */
HipAM::HipAM(std::shared_ptr<HipPair> pair) : p(pair) {
	name = p->am == MTFHIP_AM_SSD ? "ssd" : (p->am == MTFHIP_AM_NCC ? "ncc" : "mi");
	I0.resize(p->N); It.resize(p->N);
	dI0_dx.resize(p->N, 2); dIt_dx.resize(p->N, 2);
	d2I0_dx2.resize(4, p->N); d2It_dx2.resize(4, p->N);
	p->init_grad_key = dI0_dx.data();
	p->curr_grad_key = dIt_dx.data();
	p->init_hess_key = d2I0_dx2.data();
	p->curr_hess_key = d2It_dx2.data();
}
static std::shared_ptr<HipPair> pair_of(const std::shared_ptr<HipLink> &link, int n_channels, const char *who) {
	if (!link) throw utils::InvalidArgument(std::string(who) + " :: the parameter block carries no HipLink (the AM and the SSM of a tracker share one)");
	return link->pair(n_channels);
}
HipAM::HipAM(const ParamType *params, int n_channels) : HipAM(pair_of(params ? params->link : nullptr, n_channels, "HipAM")) {
	learning_rate = params->learning_rate;
}
const double *HipAM::hessPtsArg(const HessPtsT &pts) const { return pts.data() == p->hess_pts_key ? nullptr : pts.data(); }
const double *HipAM::ptsArg(const PtsT &pts) const { return pts.data() == p->pts_key ? nullptr : pts.data(); }
const double *HipAM::gradPtsArg(const GradPtsT &pts) const { return pts.data() == p->grad_pts_key ? nullptr : pts.data(); }

/* ImageBase::setCurrImg AM/src/ImageBase.cc:38-60: the buffer is borrowed and overwritten in place by the
 * caller every frame, so the device copy is refreshed here and again on setFirstIter() */
void HipAM::setCurrImg(const ImageView &im) {
	if (!im.data) throw utils::InvalidArgument("ImageBase::Input image is empty");
	img = im;
	HipPair::check(mtfhip_image_upload_mc(p->ctx, im.data, im.rows, im.cols, im.step, im.channels));
}
void HipAM::setFirstIter() {
	first_iter = true;
	if (img.data) HipPair::check(mtfhip_image_upload_mc(p->ctx, img.data, img.rows, img.cols, img.step, img.channels));
}
const PixValT &HipAM::getInitPixVals() { HipPair::check(mtfhip_batch_read(p->b, MTFHIP_BUF_I0, I0.data())); return I0; }
const PixValT &HipAM::getCurrPixVals() { HipPair::check(mtfhip_batch_read(p->b, MTFHIP_BUF_IT, It.data())); return It; }
void HipAM::syncPixGrad() {
	HipPair::check(mtfhip_batch_read(p->b, MTFHIP_BUF_DI0_DX, dI0_dx.data()));
	HipPair::check(mtfhip_batch_read(p->b, MTFHIP_BUF_DIT_DX, dIt_dx.data()));
}
void HipAM::initializePixVals(const PtsT &pts) { HipPair::check(mtfhip_am_initialize_pix_vals(p->b, ptsArg(pts))); }
void HipAM::updatePixVals(const PtsT &pts) { HipPair::check(mtfhip_am_update_pix_vals(p->b, ptsArg(pts))); }
void HipAM::initializePixGrad(const PtsT &pts) { HipPair::check(mtfhip_am_initialize_pix_grad(p->b, ptsArg(pts))); d_dI0 = true; }
void HipAM::updatePixGrad(const PtsT &pts) { HipPair::check(mtfhip_am_update_pix_grad(p->b, ptsArg(pts))); d_dIt = true; }
void HipAM::initializePixGrad(const GradPtsT &gp) { HipPair::check(mtfhip_am_initialize_pix_grad_warped(p->b, gradPtsArg(gp))); d_dI0 = true; }
void HipAM::updatePixGrad(const GradPtsT &gp) { HipPair::check(mtfhip_am_update_pix_grad_warped(p->b, gradPtsArg(gp))); d_dIt = true; }

void HipAM::initializePixHess(const PtsT &pts) { HipPair::check(mtfhip_am_initialize_pix_hess(p->b, ptsArg(pts))); d_h0 = true; }
void HipAM::updatePixHess(const PtsT &pts) { HipPair::check(mtfhip_am_update_pix_hess(p->b, ptsArg(pts))); d_ht = true; }
void HipAM::initializePixHess(const PtsT &pts, const HessPtsT &hp) {
	HipPair::check(mtfhip_am_initialize_pix_hess_warped(p->b, ptsArg(pts), hessPtsArg(hp))); d_h0 = true;
}
void HipAM::updatePixHess(const PtsT &pts, const HessPtsT &hp) {
	HipPair::check(mtfhip_am_update_pix_hess_warped(p->b, ptsArg(pts), hessPtsArg(hp))); d_ht = true;
}

double HipAM::getLikelihood() const { double l = 0; HipPair::check(mtfhip_am_get_likelihood(p->b, &l)); return l; }
void HipAM::initializeSimilarity() {
	HipPair::check(mtfhip_am_initialize_similarity(p->b));
	HipPair::check(mtfhip_am_get_similarity(p->b, &f));
}
void HipAM::initializeGrad() { HipPair::check(mtfhip_am_initialize_grad(p->b)); }
void HipAM::initializeHess() { HipPair::check(mtfhip_am_initialize_hess(p->b)); }
/* The value is fetched when getSimilarity() is called, not here: the library defers the pixel-level calls of an
 * iteration until something needs a number on the host, and reading f eagerly would cut every iteration in two. */
void HipAM::updateSimilarity(bool prereq_only) {
	HipPair::check(mtfhip_am_update_similarity(p->b, prereq_only ? 1 : 0));
	if (!prereq_only) f_fresh = false;
}
double HipAM::getSimilarity() const {
	if (!f_fresh) { HipPair::check(mtfhip_am_get_similarity(p->b, &f)); f_fresh = true; }
	return f;
}
void HipAM::updateInitGrad() { HipPair::check(mtfhip_am_update_init_grad(p->b)); }
void HipAM::updateCurrGrad() { HipPair::check(mtfhip_am_update_curr_grad(p->b)); }
void HipAM::updateDistFeat(double *feat_addr) {
	double st[8] = {0};
	HipPair::check(mtfhip_ssm_get_state(p->b, st));
	HipPair::check(mtfhip_sample_candidates(p->b, st, 1, feat_addr));
}
void HipAM::updateModel(const PtsT &pts) {
	HipPair::check(mtfhip_am_update_model(p->b, ptsArg(pts), learning_rate));
}

void HipAM::cmptInitJacobian(RowVectorXd &g, const MatrixXd &J0) {
	HipPair::check(mtfhip_am_cmpt_init_jacobian(p->b, p->jacobianBuffer(J0, false), g.data()));
}
void HipAM::cmptCurrJacobian(RowVectorXd &g, const MatrixXd &Jt) {
	HipPair::check(mtfhip_am_cmpt_curr_jacobian(p->b, p->jacobianBuffer(Jt, false), g.data()));
}
void HipAM::cmptDifferenceOfJacobians(RowVectorXd &g, const MatrixXd &J0, const MatrixXd &Jt) {
	HipPair::check(mtfhip_am_cmpt_difference_of_jacobians(p->b, p->jacobianBuffer(J0, false), p->jacobianBuffer(Jt, false), g.data()));
}
void HipAM::cmptInitHessian(MatrixXd &H, const MatrixXd &J0) {
	HipPair::check(mtfhip_am_cmpt_init_hessian(p->b, p->jacobianBuffer(J0, false), H.data()));
}
void HipAM::cmptCurrHessian(MatrixXd &H, const MatrixXd &Jt) {
	HipPair::check(mtfhip_am_cmpt_curr_hessian(p->b, p->jacobianBuffer(Jt, false), H.data()));
}
void HipAM::cmptSelfHessian(MatrixXd &H, const MatrixXd &Jt) {
	HipPair::check(mtfhip_am_cmpt_self_hessian(p->b, p->jacobianBuffer(Jt, false), H.data()));
}
void HipAM::cmptSumOfHessians(MatrixXd &H, const MatrixXd &J0, const MatrixXd &Jt) {
	HipPair::check(mtfhip_am_cmpt_sum_of_hessians(p->b, p->jacobianBuffer(J0, false), p->jacobianBuffer(Jt, false), H.data()));
}

void HipAM::cmptInitHessian(MatrixXd &H, const MatrixXd &J0, const MatrixXd &D0) {
	HipPair::check(mtfhip_am_cmpt_init_hessian2(p->b, p->jacobianBuffer(J0, false), p->hessianBuffer(D0, false), H.data()));
}
void HipAM::cmptCurrHessian(MatrixXd &H, const MatrixXd &Jt, const MatrixXd &Dt) {
	HipPair::check(mtfhip_am_cmpt_curr_hessian2(p->b, p->jacobianBuffer(Jt, false), p->hessianBuffer(Dt, false), H.data()));
}
void HipAM::cmptSelfHessian(MatrixXd &H, const MatrixXd &Jt, const MatrixXd &Dt) {
	HipPair::check(mtfhip_am_cmpt_self_hessian2(p->b, p->jacobianBuffer(Jt, false), p->hessianBuffer(Dt, false), H.data()));
}
void HipAM::cmptSumOfHessians(MatrixXd &H, const MatrixXd &J0, const MatrixXd &Jt, const MatrixXd &D0, const MatrixXd &Dt) {
	HipPair::check(mtfhip_am_cmpt_sum_of_hessians2(p->b, p->jacobianBuffer(J0, false), p->jacobianBuffer(Jt, false),
		p->hessianBuffer(D0, false), p->hessianBuffer(Dt, false), H.data()));
}
/* the SM's mean_pix_jacobian / mean_pix_hessian (NT/ESM.cc:239-242,325) of two device-resident matrices */
void HipAM::cmptMeanOf(MatrixXd &mean, const MatrixXd &a, const MatrixXd &b) {
	if (mean.rows() == p->N && mean.cols() == p->S) {
		if (p->jacobianBuffer(a, false) != MTFHIP_BUF_J0 || p->jacobianBuffer(b, false) != MTFHIP_BUF_JT)
			throw utils::LogicError("cmptMeanOf: expected (init_pix_jacobian, curr_pix_jacobian)");
		auto it = p->jac_keys.find(mean.data());
		if (it == p->jac_keys.end()) p->jac_keys[mean.data()] = MTFHIP_BUF_JM;
		else if (it->second != MTFHIP_BUF_JM) throw utils::LogicError("cmptMeanOf: the mean Jacobian aliases another device matrix");
		HipPair::check(mtfhip_sm_mean_jacobian(p->b));
	} else if (mean.rows() == p->S * p->S && mean.cols() == p->N) {
		if (p->hessianBuffer(a, false) != MTFHIP_BUF_D2I0_DP2 || p->hessianBuffer(b, false) != MTFHIP_BUF_D2IT_DP2)
			throw utils::LogicError("cmptMeanOf: expected (init_pix_hessian, curr_pix_hessian)");
		auto it = p->hess_keys.find(mean.data());
		if (it == p->hess_keys.end()) p->hess_keys[mean.data()] = MTFHIP_BUF_D2IM_DP2;
		else if (it->second != MTFHIP_BUF_D2IM_DP2) throw utils::LogicError("cmptMeanOf: the mean pixel Hessian aliases another device matrix");
		HipPair::check(mtfhip_sm_mean_pix_hessian(p->b));
	} else throw utils::InvalidArgument("cmptMeanOf: matrix is neither N x S nor S^2 x N");
}

/* ------------------------------------------------------------------ SSM */
HipSSM::HipSSM(std::shared_ptr<HipPair> pair) : p(pair) {
	name = p->ssm == MTFHIP_SSM_HOMOGRAPHY ? "homography" : "affine";
	curr_pts.resize(2, p->n_pix);
	grad_pts.resize(8, p->n_pix);
	hess_pts.resize(16, p->n_pix);
	p->hess_pts_key = hess_pts.data();
	curr_state.resize(p->S);
	std::memset(curr_corners.data(), 0, sizeof(double) * 8);
	p->pts_key = curr_pts.data();
	p->grad_pts_key = grad_pts.data();
}
HipSSM::HipSSM(const ParamType *params) : HipSSM(pair_of(params ? params->link : nullptr, 1, "HipSSM")) {
	corner_based_sampling = params->corner_based_sampling;
	pt_based_sampling = params->pt_based_sampling;
}
void HipSSM::syncSmall() {
	HipPair::check(mtfhip_ssm_get_corners(p->b, curr_corners.data()));
	HipPair::check(mtfhip_ssm_get_state(p->b, curr_state.data()));
}
void HipSSM::syncPts() { HipPair::check(mtfhip_batch_read(p->b, MTFHIP_BUF_CURR_PTS, curr_pts.data())); d_pts = false; }
void HipSSM::setCorners(const CornersT &c) { HipPair::check(mtfhip_ssm_set_corners(p->b, c.data())); syncSmall(); d_pts = d_gpts = d_hpts = true; }
void HipSSM::setState(const VectorXd &s) {
	if (s.size() != p->S) throw utils::InvalidArgument("setState: state has invalid size");   /* validate_ssm_state */
	HipPair::check(mtfhip_ssm_set_state(p->b, s.data())); syncSmall(); d_pts = d_gpts = d_hpts = true;
}
void HipSSM::compositionalUpdate(const VectorXd &dp) {
	if (dp.size() != p->S) throw utils::InvalidArgument("compositionalUpdate: state update has invalid size");
	HipPair::check(mtfhip_ssm_compositional_update(p->b, dp.data())); syncSmall(); d_pts = d_gpts = d_hpts = true;
}
void HipSSM::updateGradPts(double eps) { HipPair::check(mtfhip_ssm_update_grad_pts(p->b, eps)); d_gpts = true; }
void HipSSM::updateHessPts(double eps) { HipPair::check(mtfhip_ssm_update_hess_pts(p->b, eps)); d_hpts = true; }
void HipSSM::invertState(VectorXd &inv, const VectorXd &s) { HipPair::check(mtfhip_ssm_invert_state(p->b, s.data(), inv.data())); }
void HipSSM::applyWarpToCorners(CornersT &out, const CornersT &in, const VectorXd &s) {
	HipPair::check(mtfhip_ssm_apply_warp_to_corners(p->b, in.data(), s.data(), out.data()));
}
void HipSSM::additiveUpdate(const VectorXd &dp) {
	if (dp.size() != p->S) throw utils::InvalidArgument("additiveUpdate: state update has invalid size");
	HipPair::check(mtfhip_ssm_additive_update(p->b, dp.data())); syncSmall(); d_pts = d_gpts = d_hpts = true;
}
void HipSSM::applyWarpToPts(PtsT &out, const PtsT &in, const VectorXd &s) {
	if (out.rows() != in.rows() || out.cols() != in.cols()) out.resize(in.rows(), in.cols());
	HipPair::check(mtfhip_ssm_apply_warp_to_pts(p->ssm, in.data(), in.cols(), s.data(), out.data()));   /* 2 x n column-major = x, y interleaved */
}
void HipSSM::getIdentityWarp(VectorXd &w) {
	if (w.size() != p->S) w.resize(p->S);
	HipPair::check(mtfhip_ssm_identity_warp(p->ssm, w.data()));
}
void HipSSM::composeWarps(VectorXd &out, const VectorXd &s1, const VectorXd &s2) {
	if (s1.size() != p->S || s2.size() != p->S) throw utils::InvalidArgument("composeWarps: state has invalid size");
	if (out.size() != p->S) out.resize(p->S);
	HipPair::check(mtfhip_ssm_compose_warps(p->ssm, s1.data(), s2.data(), out.data()));
}
void HipSSM::estimateWarpFromCorners(VectorXd &out, const CornersT &in, const CornersT &oc) {
	if (out.size() != p->S) throw utils::InvalidArgument("estimateWarpFromCorners: state update has invalid size");   /* validate_ssm_state */
	HipPair::check(mtfhip_ssm_estimate_warp_from_corners(p->ssm, in.data(), oc.data(), out.data()));
}
int HipSSM::gradBuffer(const PixGradT &g) {
	if (g.data() == p->init_grad_key) return MTFHIP_BUF_DI0_DX;
	if (g.data() == p->curr_grad_key) return MTFHIP_BUF_DIT_DX;
	/* a foreign gradient: upload it into the current-gradient buffer */
	HipPair::check(mtfhip_batch_write(p->b, MTFHIP_BUF_DIT_DX, g.data()));
	return MTFHIP_BUF_DIT_DX;
}
void HipSSM::jac(int variant, MatrixXd &J, const PixGradT &g) {
	if (J.rows() != p->N || J.cols() != p->S) throw utils::InvalidArgument("pixel Jacobian has invalid size");   /* validate_ssm_jacobian */
	const int gb = gradBuffer(g);
	HipPair::check(mtfhip_ssm_cmpt_pix_jacobian(p->b, variant, gb, p->jacobianBuffer(J, true, gb == MTFHIP_BUF_DI0_DX ? MTFHIP_BUF_J0 : MTFHIP_BUF_JT)));
}
void HipSSM::pixHess(int variant, MatrixXd &D, const PixHessT &h, const PixGradT &g) {
	if (D.rows() != p->S * p->S || D.cols() != p->N) throw utils::InvalidArgument("pixel Hessian has invalid size");   /* validate_ssm_hessian */
	int hess_buf;
	if (h.data() == p->init_hess_key) hess_buf = MTFHIP_BUF_D2I0_DX2;
	else if (h.data() == p->curr_hess_key) hess_buf = MTFHIP_BUF_D2IT_DX2;
	else { HipPair::check(mtfhip_batch_write(p->b, MTFHIP_BUF_D2IT_DX2, h.data())); hess_buf = MTFHIP_BUF_D2IT_DX2; }
	HipPair::check(mtfhip_ssm_cmpt_pix_hessian(p->b, variant, hess_buf, gradBuffer(g), p->hessianBuffer(D, true)));
}

} // namespace hip
} // namespace mtf

/* ------------------------------------------------------------------ HipSSM: stochastic sampler (host side) */
#include <random>
namespace mtf {
namespace hip {

struct HipSSM::Rng {
	std::vector<std::mt19937_64> gen;   /* one generator per state component, as ProjectiveBase::initializeSampler :163-199 */
	std::vector<std::normal_distribution<double>> dist;
};
namespace {
struct W3 { double m[9]; };
W3 warpOf(int ssm, const double *p) {
	W3 W;
	if (ssm == MTFHIP_SSM_HOMOGRAPHY) { W = W3{{1 + p[0], p[1], p[2], p[3], 1 + p[4], p[5], p[6], p[7], 1}}; }
	else { W = W3{{1 + p[2], p[3], p[0], p[4], 1 + p[5], p[1], 0, 0, 1}}; }
	return W;
}
void stateOf(int ssm, double *p, const W3 &W) {
	if (ssm == MTFHIP_SSM_HOMOGRAPHY) { p[0] = W.m[0] - 1; p[1] = W.m[1]; p[2] = W.m[2]; p[3] = W.m[3]; p[4] = W.m[4] - 1; p[5] = W.m[5]; p[6] = W.m[6]; p[7] = W.m[7]; }
	else { p[0] = W.m[2]; p[1] = W.m[5]; p[2] = W.m[0] - 1; p[3] = W.m[1]; p[4] = W.m[3]; p[5] = W.m[4] - 1; }
}
W3 mul(const W3 &a, const W3 &b) {
	W3 c;
	for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
	return c;
}
W3 inv(const W3 &a) {
	const double *u = a.m;
	W3 c;
	c.m[0] = u[4] * u[8] - u[5] * u[7]; c.m[1] = u[2] * u[7] - u[1] * u[8]; c.m[2] = u[1] * u[5] - u[2] * u[4];
	c.m[3] = u[5] * u[6] - u[3] * u[8]; c.m[4] = u[0] * u[8] - u[2] * u[6]; c.m[5] = u[2] * u[3] - u[0] * u[5];
	c.m[6] = u[3] * u[7] - u[4] * u[6]; c.m[7] = u[1] * u[6] - u[0] * u[7]; c.m[8] = u[0] * u[4] - u[1] * u[3];
	const double id = 1.0 / (u[0] * c.m[0] + u[1] * c.m[3] + u[2] * c.m[6]);
	for (double &v : c.m) v *= id;
	return c;
}
void norm22(W3 &w) { const double s = w.m[8]; for (double &v : w.m) v /= s; }
}  // namespace

/* ProjectiveBase::initializeSampler :163-199 (a sigma / mean of size 1 is broadcast) */
void HipSSM::initializeSampler(const VectorXd &sg, const VectorXd &mn) {
	const int S = p->S;
	if (sg.size() != 1 && sg.size() != S) throw utils::InvalidArgument("ProjectiveBase::initializeSampler :: SSM sigma has invalid size " + std::to_string(sg.size()));
	if (mn.size() != 1 && mn.size() != S) throw utils::InvalidArgument("ProjectiveBase::initializeSampler :: SSM mean has invalid size " + std::to_string(mn.size()));
	sampler_sigma.resize(S); sampler_mean.resize(S);
	for (int s = 0; s < S; ++s) { sampler_sigma[s] = sg.size() == 1 ? sg[0] : sg[s]; sampler_mean[s] = mn.size() == 1 ? mn[0] : mn[s]; }
	rng = std::make_shared<Rng>();
	std::random_device rd;
	for (int s = 0; s < S; ++s) {
		std::seed_seq seq{rd(), rd(), rd(), rd(), rd(), rd(), rd(), rd()};
		rng->gen.emplace_back(seq);
		rng->dist.emplace_back(sampler_mean[s], sampler_sigma[s]);
	}
	sampler_ready = true;
}
void HipSSM::estimateStateSigma(VectorXd &state_sigma, double pix_sigma) {
	state_sigma.resize(p->S);
	HipPair::check(mtfhip_ssm_estimate_state_sigma(p->b, pix_sigma, state_sigma.data()));
}
void HipSSM::setSamplerSeed(unsigned long long seed) {
	if (!sampler_ready) throw utils::LogicError("setSamplerSeed before initializeSampler");
	for (int s = 0; s < p->S; ++s) { rng->gen[s].seed(seed * 1000003ULL + s); rng->dist[s].reset(); }
}
void HipSSM::setSampler(const VectorXd &sg, const VectorXd &mn) {   /* :208-215 */
	if (!sampler_ready) throw utils::LogicError("setSampler before initializeSampler");
	for (int s = 0; s < p->S; ++s) { sampler_sigma[s] = sg[s]; sampler_mean[s] = mn[s]; rng->dist[s] = std::normal_distribution<double>(mn[s], sg[s]); }
}
void HipSSM::setSamplerMean(const VectorXd &mn) { setSampler(sampler_sigma, mn); }     /* :217-223 */
void HipSSM::setSamplerSigma(const VectorXd &sg) { setSampler(sg, sampler_mean); }     /* :224-230 */
double HipSSM::draw(int state_id) {
	if (!sampler_ready) throw utils::LogicError("the SSM's sampler is used before initializeSampler");
	return rng->dist[state_id](rng->gen[state_id]);
}
/* Homography::generatePerturbation Homography.cc:899-915 (corner based: the warp that takes the template corners to randomly
 * displaced ones) / ProjectiveBase::generatePerturbation :283-288 / Affine::generatePerturbation Affine.cc:464-503 (point
 * based: bottom right, bottom left and top centre of the template disturbed, affine map of the three pairs; geometric:
 * geomToState of six draws, Affine.cc:393-410) */
void HipSSM::generatePerturbation(VectorXd &pert) {
	const int S = p->S;
	if (pert.size() != S) pert.resize(S);
	if (p->ssm == MTFHIP_SSM_HOMOGRAPHY && corner_based_sampling) {
		CornersT ic, dc;
		HipPair::check(mtfhip_ssm_get_init_corners(p->b, ic.data()));
		const double tx = draw(0), ty = draw(0);
		double rd[8];
		for (int c = 0; c < 4; ++c) { rd[2 * c] = draw(1); rd[2 * c + 1] = draw(1); }
		for (int c = 0; c < 4; ++c) { dc.data()[2 * c] = ic.data()[2 * c] + rd[2 * c] + tx; dc.data()[2 * c + 1] = ic.data()[2 * c + 1] + rd[2 * c + 1] + ty; }
		estimateWarpFromCorners(pert, ic, dc);
		return;
	}
	if (p->ssm == MTFHIP_SSM_AFFINE && pt_based_sampling) {
		CornersT ic;
		HipPair::check(mtfhip_ssm_get_init_corners(p->b, ic.data()));
		const double ox[3] = {ic.data()[4], ic.data()[6], (ic.data()[0] + ic.data()[2]) / 2.0}, oy[3] = {ic.data()[5], ic.data()[7], (ic.data()[1] + ic.data()[3]) / 2.0};
		double px[3], py[3];
		if (pt_based_sampling == 1) {
			for (int i = 0; i < 3; ++i) { px[i] = ox[i] + draw(2 * i); py[i] = oy[i] + draw(2 * i + 1); }
		} else {
			double rd[6];
			for (int i = 0; i < 3; ++i) { rd[2 * i] = draw(1); rd[2 * i + 1] = draw(1); }
			const double tx = draw(0), ty = draw(0);
			for (int i = 0; i < 3; ++i) { px[i] = (ox[i] + rd[2 * i]) + tx; py[i] = (oy[i] + rd[2 * i + 1]) + ty; }
		}
		/* utils::computeAffineDLT for three point pairs (warpUtils.cc:388-421): an exactly determined system */
		const W3 Mi = inv(W3{{ox[0], oy[0], 1, ox[1], oy[1], 1, ox[2], oy[2], 1}});
		W3 W{{0, 0, 0, 0, 0, 0, 0, 0, 1}};
		for (int j = 0; j < 3; ++j) {
			W.m[j] = Mi.m[3 * j] * px[0] + Mi.m[3 * j + 1] * px[1] + Mi.m[3 * j + 2] * px[2];
			W.m[3 + j] = Mi.m[3 * j] * py[0] + Mi.m[3 * j + 1] * py[1] + Mi.m[3 * j + 2] * py[2];
		}
		stateOf(p->ssm, pert.data(), W);
		return;
	}
	if (p->ssm == MTFHIP_SSM_AFFINE) {   /* geometric: Affine.cc:495-502 */
		double gm[6];
		for (int s = 0; s < 6; ++s) gm[s] = draw(s);
		const double sc = gm[2], r = gm[4], theta = gm[3], phi = gm[5];
		const double cos_theta = std::cos(theta), sin_theta = std::sin(theta), cos_phi = std::cos(phi), sin_phi = std::sin(phi);
		const double ccc = cos_theta * cos_phi * cos_phi, ccs = cos_theta * cos_phi * sin_phi, css = cos_theta * sin_phi * sin_phi;
		const double scc = sin_theta * cos_phi * cos_phi, scs = sin_theta * cos_phi * sin_phi, sss = sin_theta * sin_phi * sin_phi;
		pert[0] = gm[0]; pert[1] = gm[1];
		pert[2] = sc * (ccc + scs + r * (css - scs)) - 1;
		pert[3] = sc * (r * (ccs - scc) - ccs - sss);
		pert[4] = sc * (scc - ccs + r * (ccs + sss));
		pert[5] = sc * (r * (ccc + scs) - scs + css) - 1;
		return;
	}
	for (int s = 0; s < S; ++s) pert[s] = draw(s);
}
/* Affine's additive models perturb the geometric parametrisation (Affine.cc:507-538): stateToGeom is a 2 x 2 JacobiSVD whose
 * sign / ordering conventions select its branches -- not reproducible without Eigen, so they are refused rather than replaced
 * by raw-state noise; with point based sampling the reference itself throws */
void HipSSM::affineAdditiveRefused(const char *fn) const {
	if (pt_based_sampling) throw utils::FunctonNotImplemented(std::string("Affine::") + fn + " :: point based sampling is not implemented yet");
	throw utils::FunctonNotImplemented(std::string("Affine::") + fn + " :: geometric sampling needs Affine::stateToGeom (Eigen JacobiSVD conventions); use the compositional models");
}
void HipSSM::additiveRandomWalk(VectorXd &out, const VectorXd &base) {   /* :236-240 */
	if (p->ssm == MTFHIP_SSM_AFFINE) affineAdditiveRefused("additiveRandomWalk");
	VectorXd pert(p->S);
	generatePerturbation(pert);
	if (out.size() != p->S) out.resize(p->S);
	for (int s = 0; s < p->S; ++s) out[s] = base[s] + pert[s];
}
void HipSSM::compositionalRandomWalk(VectorXd &out, const VectorXd &base) {   /* Homography.cc:916-926, Affine.cc:539-553 */
	if (p->ssm == MTFHIP_SSM_AFFINE && !pt_based_sampling)
		throw utils::FunctonNotImplemented("Affine::compositionalRandomWalk :: geometric sampling is not implemented yet");
	VectorXd pert(p->S);
	generatePerturbation(pert);
	W3 W = mul(warpOf(p->ssm, base.data()), warpOf(p->ssm, pert.data()));
	if (p->ssm == MTFHIP_SSM_HOMOGRAPHY) norm22(W);
	if (out.size() != p->S) out.resize(p->S);
	stateOf(p->ssm, out.data(), W);
}
void HipSSM::additiveAutoRegression1(VectorXd &out, VectorXd &out_ar, const VectorXd &base, const VectorXd &base_ar, double a) {   /* :254-259 */
	if (p->ssm == MTFHIP_SSM_AFFINE) affineAdditiveRefused("additiveAutoRegression1");
	VectorXd pert(p->S);
	generatePerturbation(pert);
	if (out.size() != p->S) out.resize(p->S);
	if (out_ar.size() != p->S) out_ar.resize(p->S);
	for (int s = 0; s < p->S; ++s) { out[s] = base[s] + base_ar[s] + pert[s]; out_ar[s] = a * (out[s] - base[s]); }
}
void HipSSM::compositionalAutoRegression1(VectorXd &out, VectorXd &out_ar, const VectorXd &base, const VectorXd &base_ar, double a) {   /* Homography.cc:928-942 */
	VectorXd pert(p->S);
	generatePerturbation(pert);
	const bool hom = p->ssm == MTFHIP_SSM_HOMOGRAPHY;
	const W3 B = warpOf(p->ssm, base.data());
	W3 W = mul(mul(B, warpOf(p->ssm, base_ar.data())), warpOf(p->ssm, pert.data()));
	if (hom) norm22(W);
	W3 AW = mul(inv(B), W);
	if (hom) norm22(AW);
	if (out.size() != p->S) out.resize(p->S);
	if (out_ar.size() != p->S) out_ar.resize(p->S);
	stateOf(p->ssm, out.data(), W); stateOf(p->ssm, out_ar.data(), AW);
	for (int s = 0; s < p->S; ++s) out_ar[s] *= a;
}
void HipSSM::estimateMeanOfSamples(VectorXd &mean, const std::vector<VectorXd> &samples, int n) {   /* :311-317 */
	if (mean.size() != p->S) mean.resize(p->S);
	mean.fill(0.0);
	for (int k = 0; k < n; ++k) for (int s = 0; s < p->S; ++s) mean[s] += (samples[k][s] - mean[s]) / (k + 1);
}

} // namespace hip
} // namespace mtf
