/*
 * HipModels.h -- AppearanceModel / StateSpaceModel subclasses whose data lives in HBM behind the C ABI
 * (include/mtfhip.h).  One HipPair (context + a batch of one target) is shared by the AM and the SSM of
 * a tracker.  References the two objects hand each other through the search method (PtsT, GradPtsT,
 * PixGradT, the SM-owned N x S Jacobians) are recognised BY ADDRESS: the bytes stay on the device and the
 * host objects are only keys; anything else is uploaded.  Host mirrors behind the getters are refreshed
 * lazily, when somebody actually asks ("host-resident interface vs device-resident data").
 */
#ifndef MTF_AMD_HOST_HIP_MODELS_H
#define MTF_AMD_HOST_HIP_MODELS_H

#include <map>
#include <memory>

#include "AppearanceModel.h"
#include "StateSpaceModel.h"
#include "../../include/mtfhip.h"

namespace mtf {
namespace hip {

struct HipPair {
	mtfhip_ctx *ctx = nullptr;
	mtfhip_batch *b = nullptr;
	int am, ssm, resx, resy, N, S;   /* N = patch size = n_pix * n_channels */
	int n_pix, n_channels;
	double grad_eps, hess_eps = 1.0;
	/* addresses of host mirrors whose authoritative copy is a device buffer */
	const void *pts_key = nullptr, *grad_pts_key = nullptr, *init_grad_key = nullptr, *curr_grad_key = nullptr;
	const void *hess_pts_key = nullptr, *init_hess_key = nullptr, *curr_hess_key = nullptr;
	std::map<const void *, int> jac_keys;   /* SM-owned Jacobian matrices -> MTFHIP_BUF_J0 / _JT / _JM */
	std::map<const void *, int> hess_keys;  /* SM-owned pixel Hessians -> MTFHIP_BUF_D2I0_DP2 / _D2IT_DP2 / _D2IM_DP2 */
	int next_jac = 0, next_hess = 0;
	/* The host objects behind getPts(), getGradPts(), getCurrPixGrad() ... are keys: the adapters recognise them by address and
	 * their bytes stay on the device.  A reader that is NOT one of the adapters (the reference has a few: the SPI helpers next to
	 * NT/ESM.cc:466-472, NN's walk over ssm->getPts()) needs the bytes: with eager getters every such getter first reads its array
	 * back when a mutator has changed it since (a dirty flag per array) -- never stale, one D2H copy per changed array and getter
	 * call.  Off by default (the search methods call these getters every iteration only to pass the keys on);
	 * MTFHIP_EAGER_GETTERS=1 or setEagerGetters(true) turns it on. */
	bool eager_getters = false;
	void setEagerGetters(bool on) { eager_getters = on; }

	HipPair(int am, int ssm, int resx, int resy, double grad_eps, double likelihood_alpha, int mi_n_bins,
		double mi_pre_seed, int mi_pou, int device, void *stream, int n_channels = 1);
	~HipPair();
	static void check(int rc);               /* rethrows C-ABI failures as mtf::utils::Exception */
	int jacobianBuffer(const MatrixXd &J, bool may_register, int preferred = -1);
	int hessianBuffer(const MatrixXd &D, bool may_register);
};

/* What the AM and the SSM of ONE tracker share, in the form the reference's templated search methods can carry it.
 * SearchMethod<AM, SSM> holds its models BY VALUE and builds them from `const AM::ParamType *` / `const SSM::ParamType *`
 * (SM/include/mtf/SM/SearchMethod.h:13-19, SM/src/ESM.cc:79-118; the models' own constructors: AM/include/mtf/AM/SSD.h
 * `SSD(const ParamType *ssd_params = nullptr, const int _n_channels = 1)`, SSM/include/mtf/SSM/Homography.h
 * `Homography(const ParamType *params_in = nullptr)`), so the two adapter objects cannot be handed a shared pair by the caller: both
 * parameter blocks point at one HipLink, and whichever model is constructed first creates the pair (context + one-target batch). */
struct HipLink {
	int am = MTFHIP_AM_SSD, ssm = MTFHIP_SSM_HOMOGRAPHY, resx = 50, resy = 50;   /* AMParams / SSMParams: resx, resy */
	double grad_eps = 1e-8, likelihood_alpha = 1.0;                           /* AMParams::grad_eps; SSDParams / NCCParams / MIParams::likelihood_alpha */
	int mi_n_bins = 8; double mi_pre_seed = 10; int mi_pou = 0;                /* MIParams */
	int device = 0; void *stream = nullptr;
	std::shared_ptr<HipPair> pair(int n_channels = 1) {
		if (!p) p = std::make_shared<HipPair>(am, ssm, resx, resy, grad_eps, likelihood_alpha, mi_n_bins, mi_pre_seed, mi_pou, device, stream, n_channels);
		else if (n_channels > 1 && n_channels != p->n_channels)   /* (SearchMethod<AM, SSM> constructs the AM first: it fixes the channel count) */
			throw utils::InvalidArgument("HipLink :: the pair exists with another channel count");
		return p;
	}
private:
	std::shared_ptr<HipPair> p;
};
struct HipAMParams { std::shared_ptr<HipLink> link; double learning_rate = 0.5; };
struct HipSSMParams { std::shared_ptr<HipLink> link; bool corner_based_sampling = true; int pt_based_sampling = 0; };

class HipAM : public AppearanceModel {
public:
	typedef HipAMParams ParamType;
	HipAM(std::shared_ptr<HipPair> pair);
	/* the constructor SearchMethod<AM, SSM> calls (am(am_params): n_channels defaults as in the reference's models) */
	explicit HipAM(const ParamType *params, int n_channels = 1);
	unsigned int getResX() const override { return p->resx; }
	unsigned int getResY() const override { return p->resy; }
	unsigned int getNPix() const override { return p->n_pix; }
	unsigned int getNChannels() const override { return p->n_channels; }   /* MCSSD / MCNCC / MCMI: 3 */
	unsigned int getPatchSize() const override { return p->N; }
	double getGradOffset() const override { return p->grad_eps; }
	double getHessOffset() const override { return p->hess_eps; }
	const PixValT &getInitPixVals() override;
	const PixValT &getCurrPixVals() override;
	const PixGradT &getInitPixGrad() override { return fresh(dI0_dx, MTFHIP_BUF_DI0_DX, d_dI0); }   /* key; bytes on the device unless eager */
	const PixGradT &getCurrPixGrad() override { return fresh(dIt_dx, MTFHIP_BUF_DIT_DX, d_dIt); }
	const PixHessT &getInitPixHess() override { return fresh(d2I0_dx2, MTFHIP_BUF_D2I0_DX2, d_h0); }
	const PixHessT &getCurrPixHess() override { return fresh(d2It_dx2, MTFHIP_BUF_D2IT_DX2, d_ht); }
	void syncPixGrad();                                               /* explicit read-back of both gradients */

	void setCurrImg(const ImageView &img) override;
#ifdef MTF_AMD_USE_OPENCV
	void setCurrImg(const cv::Mat &img) { setCurrImg(imageView(img)); }   /* the reference's signature (ImageBase.h:92) */
#endif
	void initializePixVals(const PtsT &init_pts) override;
	void initializePixGrad(const GradPtsT &warped_offset_pts) override;
	void initializePixGrad(const PtsT &init_pts) override;
	void updatePixVals(const PtsT &curr_pts) override;
	void updatePixGrad(const GradPtsT &warped_offset_pts) override;
	void updatePixGrad(const PtsT &curr_pts) override;
	void initializePixHess(const PtsT &init_pts, const HessPtsT &warped_offset_pts) override;
	void initializePixHess(const PtsT &init_pts) override;
	void updatePixHess(const PtsT &curr_pts, const HessPtsT &warped_offset_pts) override;
	void updatePixHess(const PtsT &curr_pts) override;

	double getSimilarity() const override;
	double getLikelihood() const override;
	void initializeSimilarity() override;
	void initializeGrad() override;
	void initializeHess() override;
	void updateSimilarity(bool prereq_only = true) override;
	void updateInitGrad() override;
	void updateCurrGrad() override;
	/* online template update (enable_learning of the search methods); learning_rate as SSDParams / NCCParams (default 0.5,
	 * Config/parameters.h:200), outside [0, 1] = running average */
	void updateModel(const PtsT &curr_pts) override;
	void setLearningRate(double lr) { learning_rate = lr; }
	/* NN / FLANN distance feature of the patch at the SSM's current state: SSD = the pixel values (SSDBase.h:116-125), NCC =
	 * centred and normalised (NCC.cc:530-537); computed on the device (mtfhip_sample_candidates), MI has none here */
	void initializeDistFeat() override { dist_feat.resize((int)p->N); }
	void updateDistFeat() override { if (dist_feat.size() != (int)p->N) dist_feat.resize((int)p->N); updateDistFeat(dist_feat.data()); }
	void updateDistFeat(double *feat_addr) override;
	const double *getDistFeat() override { return dist_feat.data(); }
	unsigned int getDistFeatSize() override { return p->N; }
	void cmptInitJacobian(RowVectorXd &df_dp, const MatrixXd &dI0_dpssm) override;
	void cmptCurrJacobian(RowVectorXd &df_dp, const MatrixXd &dIt_dpssm) override;
	void cmptDifferenceOfJacobians(RowVectorXd &df_dp_diff, const MatrixXd &dI0_dpssm, const MatrixXd &dIt_dpssm) override;
	void cmptInitHessian(MatrixXd &H, const MatrixXd &dI0_dpssm) override;
	void cmptCurrHessian(MatrixXd &H, const MatrixXd &dIt_dpssm) override;
	void cmptSelfHessian(MatrixXd &H, const MatrixXd &dIt_dpssm) override;
	void cmptSumOfHessians(MatrixXd &H, const MatrixXd &dI0_dpssm, const MatrixXd &dIt_dpssm) override;
	void cmptInitHessian(MatrixXd &H, const MatrixXd &dI0_dpssm, const MatrixXd &d2I0_dpssm2) override;
	void cmptCurrHessian(MatrixXd &H, const MatrixXd &dIt_dpssm, const MatrixXd &d2It_dpssm2) override;
	void cmptSelfHessian(MatrixXd &H, const MatrixXd &dIt_dpssm, const MatrixXd &d2It_dpssm2) override;
	void cmptSumOfHessians(MatrixXd &H, const MatrixXd &dI0_dpssm, const MatrixXd &dIt_dpssm, const MatrixXd &d2I0_dpssm2,
		const MatrixXd &d2It_dpssm2) override;
	void cmptMeanOf(MatrixXd &mean, const MatrixXd &a, const MatrixXd &b) override;
	void setFirstIter() override;
	void clearInitStatus() override {}
private:
	double learning_rate = 0.5;
	VectorXd dist_feat;
	std::shared_ptr<HipPair> p;
public:
	const std::shared_ptr<HipPair> &pair() const { return p; }
	/* a device-side driver (hip::LK: mtfhip_batch_init_template / track / set_region) rewrote the AM's arrays behind this object */
	void markDeviceUpdated() { d_dI0 = d_dIt = d_h0 = d_ht = true; f_fresh = false; }
private:
	ImageView img{nullptr, 0, 0, 0};
	mutable double f = 0;
	mutable bool f_fresh = true;
	PixValT I0, It;
	PixGradT dI0_dx, dIt_dx;
	PixHessT d2I0_dx2, d2It_dx2;
	bool d_dI0 = false, d_dIt = false, d_h0 = false, d_ht = false;   /* host mirror older than the device array */
	template <typename T> const T &fresh(T &m, int buf, bool &dirty) {
		if (p->eager_getters && dirty) { HipPair::check(mtfhip_batch_read(p->b, buf, m.data())); dirty = false; }
		return m;
	}
	const double *hessPtsArg(const HessPtsT &pts) const;
	const double *ptsArg(const PtsT &pts) const;
	const double *gradPtsArg(const GradPtsT &pts) const;
	static void colMajorToHost(MatrixXd &H, const double *src, int S);
};

class HipSSM : public StateSpaceModel {
public:
	typedef HipSSMParams ParamType;
	HipSSM(std::shared_ptr<HipPair> pair);
	explicit HipSSM(const ParamType *params);   /* ssm(ssm_params) of SearchMethod<AM, SSM> */
	unsigned int getStateSize() override { return p->S; }
	unsigned int getResX() override { return p->resx; }
	unsigned int getResY() override { return p->resy; }
	unsigned int getNPts() override { return p->n_pix; }
	const PtsT &getPts() override { return fresh(curr_pts, MTFHIP_BUF_CURR_PTS, d_pts); }   /* key; syncPts() / eager getters refresh the bytes */
	const GradPtsT &getGradPts() override { return fresh(grad_pts, MTFHIP_BUF_GRAD_PTS, d_gpts); }
	const HessPtsT &getHessPts() override { return fresh(hess_pts, MTFHIP_BUF_HESS_PTS, d_hpts); }
	/* corners and state live in host memory behind the C ABI: always current, also after the device-side drivers
	 * (mtfhip_batch_track, mtfhip_pf_update) moved the SSM without going through this object */
	const CornersT &getCorners() override { syncSmall(); return curr_corners; }
	const VectorXd &getState() override { syncSmall(); return curr_state; }
	void syncPts();

	void setState(const VectorXd &ssm_state) override;
	void setCorners(const CornersT &corners) override;
	void compositionalUpdate(const VectorXd &state_update) override;
	void updateGradPts(double grad_eps) override;
	void updateHessPts(double hess_eps) override;
	void invertState(VectorXd &inv_state, const VectorXd &state) override;
	void cmptInitPixJacobian(MatrixXd &J, const PixGradT &g) override { jac(MTFHIP_JAC_INIT, J, g); }
	void cmptPixJacobian(MatrixXd &J, const PixGradT &g) override { jac(MTFHIP_JAC_PIX, J, g); }
	void cmptWarpedPixJacobian(MatrixXd &J, const PixGradT &g) override { jac(MTFHIP_JAC_WARPED, J, g); }
	void cmptApproxPixJacobian(MatrixXd &J, const PixGradT &g) override { jac(MTFHIP_JAC_APPROX, J, g); }
	void cmptInitPixHessian(MatrixXd &D, const PixHessT &h, const PixGradT &g) override { pixHess(MTFHIP_JAC_INIT, D, h, g); }
	void cmptPixHessian(MatrixXd &D, const PixHessT &h, const PixGradT &g) override { pixHess(MTFHIP_JAC_PIX, D, h, g); }
	void cmptWarpedPixHessian(MatrixXd &D, const PixHessT &h, const PixGradT &g) override { pixHess(MTFHIP_JAC_WARPED, D, h, g); }
	void cmptApproxPixHessian(MatrixXd &D, const PixHessT &h, const PixGradT &g) override { pixHess(MTFHIP_JAC_APPROX, D, h, g); }
	void applyWarpToCorners(CornersT &out, const CornersT &in, const VectorXd &state) override;
	/* 3 x 3 algebra on the host, no device work (ProjectiveBase.cc:51-55,142-160,321-331; Homography.cc:877-883; Affine.cc:352-357,382-393) */
	void additiveUpdate(const VectorXd &state_update) override;
	void applyWarpToPts(PtsT &out, const PtsT &in, const VectorXd &state) override;
	void getIdentityWarp(VectorXd &identity_warp) override;
	void composeWarps(VectorXd &composed, const VectorXd &state_1, const VectorXd &state_2) override;
	void estimateWarpFromCorners(VectorXd &state_update, const CornersT &in_corners, const CornersT &out_corners) override;
	/* stochastic sampler (ProjectiveBase.cc:163-317, Homography.cc:899-942) on the host, one state per call as the interface
	 * has it -- what the literal nt::PF / nt::NN loops call.  The device filter (hip::PF, mtfhip_pf_*) generates all particles
	 * of an iteration in one launch instead.  The reference draws from boost::mt11213b seeded by random_device; here
	 * std::mt19937_64, seeded the same way unless setSamplerSeed() fixes it. */
	void initializeSampler(const VectorXd &state_sigma, const VectorXd &state_mean) override;
	void estimateStateSigma(VectorXd &state_sigma, double pix_sigma) override;   /* ProjectiveBase.cc:201-213 */
	void setSampler(const VectorXd &state_sigma, const VectorXd &state_mean) override;
	void setSamplerMean(const VectorXd &mean) override;
	void setSamplerSigma(const VectorXd &sigma) override;
	VectorXd getSamplerSigma() override { return sampler_sigma; }
	VectorXd getSamplerMean() override { return sampler_mean; }
	void compositionalRandomWalk(VectorXd &perturbed_state, const VectorXd &base_state) override;
	void additiveRandomWalk(VectorXd &perturbed_state, const VectorXd &base_state) override;
	void compositionalAutoRegression1(VectorXd &perturbed_state, VectorXd &perturbed_ar, const VectorXd &base_state,
		const VectorXd &base_ar, double a = 0.5) override;
	void additiveAutoRegression1(VectorXd &perturbed_state, VectorXd &perturbed_ar, const VectorXd &base_state,
		const VectorXd &base_ar, double a = 0.5) override;
	void generatePerturbation(VectorXd &perturbation) override;
	void estimateMeanOfSamples(VectorXd &sample_mean, const std::vector<VectorXd> &samples, int n_samples) override;
	void setSamplerSeed(unsigned long long seed);
	void markMoved() { d_pts = d_gpts = d_hpts = true; }   /* a device-side driver (hip::PF, mtfhip_batch_track) moved the SSM */
	void setCornerBasedSampling(bool on) { corner_based_sampling = on; }   /* HomographyParams::corner_based_sampling */
	bool getCornerBasedSampling() const { return corner_based_sampling; }
	void setPtBasedSampling(int mode) { pt_based_sampling = mode; }        /* AffineParams::pt_based_sampling (0 geometric, 1, 2) */
	int getPtBasedSampling() const { return pt_based_sampling; }
	const std::shared_ptr<HipPair> &pair() const { return p; }
private:
	std::shared_ptr<HipPair> p;
	bool d_pts = true, d_gpts = true, d_hpts = true;   /* host mirror older than the device array */
	template <typename T> const T &fresh(T &m, int buf, bool &dirty) {
		if (p->eager_getters && dirty) { HipPair::check(mtfhip_batch_read(p->b, buf, m.data())); dirty = false; }
		return m;
	}
	VectorXd sampler_sigma, sampler_mean;
	bool corner_based_sampling = true, sampler_ready = false;
	int pt_based_sampling = 0;
	[[noreturn]] void affineAdditiveRefused(const char *fn) const;
	struct Rng;
	std::shared_ptr<Rng> rng;
	double draw(int state_id);   /* one draw of N(mean[state_id], sigma[state_id]) */
	PtsT curr_pts;
	GradPtsT grad_pts;
	HessPtsT hess_pts;
	CornersT curr_corners;
	VectorXd curr_state;
	void syncSmall();
	void jac(int variant, MatrixXd &J, const PixGradT &g);
	void pixHess(int variant, MatrixXd &D, const PixHessT &h, const PixGradT &g);
	int gradBuffer(const PixGradT &g);
};

} // namespace hip
} // namespace mtf
#endif
