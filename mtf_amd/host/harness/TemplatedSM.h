/*
 * TemplatedSM.h (harness) -- the shape of the reference's TEMPLATED search methods, to show that the adapters instantiate in it:
 * SearchMethod<AM, SSM> (SM/include/mtf/SM/SearchMethod.h:9-80) keeps its models by value and constructs them from
 * `const AM::ParamType *` / `const SSM::ParamType *`; FCLK<AM, SSM> (SM/src/FCLK.cc:66-224) is the forward compositional loop over
 * them, condensed to what the test needs (chained warp, first-order CurrentSelf / Std Hessian, no Levenberg-Marquardt).
 * mtf::hip::HipAM / HipSSM provide ParamType and those constructors (HipModels.h); an MTF build has the originals of this file.
 */
#ifndef MTF_AMD_HOST_HARNESS_TEMPLATED_SM_H
#define MTF_AMD_HOST_HARNESS_TEMPLATED_SM_H

#include "../AppearanceModel.h"
#include "../StateSpaceModel.h"

namespace mtf {
namespace templated {

template <class AM, class SSM>
class SearchMethod {
public:
	typedef typename AM::ParamType AMParams;
	typedef typename SSM::ParamType SSMParams;
	SearchMethod(const AMParams *am_params, const SSMParams *ssm_params) : am(am_params), ssm(ssm_params) {}
	virtual ~SearchMethod() {}
	void setImage(const ImageView &img) { am.setCurrImg(img); }
	virtual void setRegion(const CornersT &corners) { ssm.setCorners(corners); }
	const CornersT &getRegion() { return ssm.getCorners(); }
	virtual AM &getAM() { return am; }
	virtual SSM &getSSM() { return ssm; }
protected:
	AM am;
	SSM ssm;
};

struct FCLKParams { int max_iters = 30; double epsilon = 1e-4; int hess_type = 1; /* FCLKParams::HessType { InitialSelf, CurrentSelf, Std } */ };

template <class AM, class SSM>
class FCLK : public SearchMethod<AM, SSM> {
public:
	typedef typename SearchMethod<AM, SSM>::AMParams AMParams;
	typedef typename SearchMethod<AM, SSM>::SSMParams SSMParams;
	using SearchMethod<AM, SSM>::am;
	using SearchMethod<AM, SSM>::ssm;
	FCLK(const FCLKParams *fclk_params, const AMParams *am_params, const SSMParams *ssm_params) :
		SearchMethod<AM, SSM>(am_params, ssm_params), params(*fclk_params) {
		const int S = (int)ssm.getStateSize(), N = (int)am.getPatchSize();
		dIt_dpssm.resize(N, S); df_dp.resize(S); d2f_dp2.resize(S, S); ssm_update.resize(S);
	}
	void initialize(const CornersT &corners) {   /* FCLK.cc:66-104 */
		am.clearInitStatus(); ssm.clearInitStatus();
		ssm.initialize(corners, am.getNChannels());
		am.initializePixVals(ssm.getPts());
		am.initializePixGrad(ssm.getPts());
		am.initializeSimilarity();
		am.initializeGrad();
		am.initializeHess();
	}
	int update() {   /* FCLK.cc:107-224 without the Levenberg-Marquardt and second-order branches */
		am.setFirstIter();
		int iters = 0;
		for (int iter_id = 0; iter_id < params.max_iters; ++iter_id) {
			am.updatePixVals(ssm.getPts());
			am.updateSimilarity(false);
			am.updateCurrGrad();
			am.updatePixGrad(ssm.getPts());
			ssm.cmptWarpedPixJacobian(dIt_dpssm, am.getCurrPixGrad());
			am.cmptCurrJacobian(df_dp, dIt_dpssm);
			if (params.hess_type == 2) am.cmptCurrHessian(d2f_dp2, dIt_dpssm); else am.cmptSelfHessian(d2f_dp2, dIt_dpssm);
			VectorXd x;
			utils::colPivHouseholderQrSolve(d2f_dp2, df_dp, x);
			for (int i = 0; i < ssm_update.size(); ++i) ssm_update[i] = -x[i];
			prev_corners = ssm.getCorners();
			ssm.compositionalUpdate(ssm_update);
			++iters;
			const double update_norm = utils::squaredDistance(prev_corners, ssm.getCorners());
			if (update_norm < params.epsilon) break;
			am.clearFirstIter();
		}
		return iters;
	}
private:
	FCLKParams params;
	MatrixXd dIt_dpssm, d2f_dp2;
	RowVectorXd df_dp;
	VectorXd ssm_update;
	CornersT prev_corners;
};

} // namespace templated
} // namespace mtf
#endif
