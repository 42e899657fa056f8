/*
 * TemplatedSM.h (harness) -- the shape of the reference's TEMPLATED search methods, to show that the adapters instantiate in it:
 * SearchMethod<AM, SSM> (SM/include/mtf/SM/SearchMethod.h:9-80) keeps its models by value and constructs them from
 * `const AM::ParamType *` / `const SSM::ParamType *`; FCLK<AM, SSM> (SM/src/FCLK.cc:66-224) is the forward compositional loop over
 * them, condensed to what the test needs (chained warp, first-order CurrentSelf / Std Hessian, no Levenberg-Marquardt); ESM<AM, SSM>
 * (SM/src/ESM.cc:14-316) and ICLK<AM, SSM> (SM/src/ICLK.cc:13-263) follow with every first-order Jacobian / Hessian type and the
 * Levenberg-Marquardt branches -- the members they size from am.getPatchSize() / getStateSize() and the calls only they make
 * (cmptInitPixJacobian in setRegion, updateState / invertState of the AM) are exercised against the adapters too.
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


/* ESM<AM, SSM> (SM/src/ESM.cc:14-77 constructor, :79-118 initialize, :120-292 update, :294-316 setRegion): the templated ESM always
 * takes the chained-warp route (updatePixGrad(getPts) + cmptWarpedPixJacobian, :170-172); first-order Hessians of every type, both
 * Jacobian types, Levenberg-Marquardt with the undo through invertState (:138-160, :261-264). */
struct ESMParams {
	int max_iters = 30; double epsilon = 1e-4;
	int jac_type = 1;    /* ESMParams::JacType { Original, DiffOfJacs } */
	int hess_type = 2;   /* ESMParams::HessType { InitialSelf, CurrentSelf, SumOfSelf, Original, SumOfStd, Std } */
	bool leven_marq = false; double lm_delta_init = 0.01, lm_delta_update = 10;
};
template <class AM, class SSM>
class ESM : public SearchMethod<AM, SSM> {
public:
	typedef typename SearchMethod<AM, SSM>::AMParams AMParams;
	typedef typename SearchMethod<AM, SSM>::SSMParams SSMParams;
	using SearchMethod<AM, SSM>::am;
	using SearchMethod<AM, SSM>::ssm;
	enum { InitialSelf, CurrentSelf, SumOfSelf, Original, SumOfStd, Std };
	ESM(const ESMParams *esm_params, const AMParams *am_params, const SSMParams *ssm_params) :
		SearchMethod<AM, SSM>(am_params, ssm_params), params(*esm_params) {
		ssm_state_size = (int)ssm.getStateSize();               /* :43-46 */
		am_state_size = am.getStateSize();
		state_size = ssm_state_size + am_state_size;
		state_update.resize(state_size); ssm_update.resize(ssm_state_size); am_update.resize(am_state_size);   /* :50-54 */
		inv_ssm_update.resize(ssm_state_size); inv_am_update.resize(am_state_size);
		df_dp.resize(state_size); d2f_dp2.resize(state_size, state_size);                                       /* :56-57 */
		if (params.hess_type == SumOfSelf || params.hess_type == InitialSelf) init_d2f_dp2.resize(state_size, state_size);   /* :58-60 */
		dI0_dpssm.resize((int)am.getPatchSize(), ssm_state_size);                                                 /* :62-63 */
		dIt_dpssm.resize((int)am.getPatchSize(), ssm_state_size);
		if (params.jac_type == 0 || params.hess_type == Original) mean_dI_dpssm.resize((int)am.getPatchSize(), ssm_state_size);   /* :65-67 */
	}
	void initialize(const CornersT &corners) {   /* :79-118 */
		am.clearInitStatus(); ssm.clearInitStatus();
		ssm.initialize(corners, am.getNChannels());
		am.initializePixVals(ssm.getPts());
		am.initializeSimilarity();
		am.initializeGrad();
		am.initializeHess();
		am.initializePixGrad(ssm.getPts());
		ssm.cmptWarpedPixJacobian(dI0_dpssm, am.getInitPixGrad());
		if (params.hess_type == InitialSelf || params.hess_type == SumOfSelf) {
			am.cmptSelfHessian(d2f_dp2, dI0_dpssm);
			init_d2f_dp2 = d2f_dp2;   /* (:107-109 copies it for SumOfSelf only and then reads init_d2f_dp2 for InitialSelf + LM at :216: kept for both) */
		}
	}
	void setRegion(const CornersT &corners) override {   /* :294-316 */
		ssm.setCorners(corners);
		ssm.cmptInitPixJacobian(dI0_dpssm, am.getInitPixGrad());
		if (params.hess_type == InitialSelf || params.hess_type == SumOfSelf) {
			am.cmptSelfHessian(d2f_dp2, dI0_dpssm);
			init_d2f_dp2 = d2f_dp2;
		}
	}
	int update() {   /* :120-292 */
		double prev_f = 0, lm_delta = params.lm_delta_init;
		bool state_reset = false;
		int iters = 0;
		am.setFirstIter();
		for (int iter_id = 0; iter_id < params.max_iters; ++iter_id) {
			am.updatePixVals(ssm.getPts());
			am.updateSimilarity(false);
			if (params.leven_marq && !state_reset) {   /* :138-160 */
				const double f = am.getSimilarity();
				if (iter_id > 0) {
					if (f < prev_f) {
						lm_delta *= params.lm_delta_update;
						ssm.invertState(inv_ssm_update, ssm_update);
						ssm.compositionalUpdate(inv_ssm_update);
						am.invertState(inv_am_update, am_update);
						am.updateState(inv_am_update);
						state_reset = true;
						continue;
					}
					if (f > prev_f) lm_delta /= params.lm_delta_update;
				}
				prev_f = f;
			}
			state_reset = false;
			am.updateCurrGrad();
			am.updateInitGrad();
			am.updatePixGrad(ssm.getPts());
			ssm.cmptWarpedPixJacobian(dIt_dpssm, am.getCurrPixGrad());
			if (params.jac_type == 0 || params.hess_type == Original) am.cmptMeanOf(mean_dI_dpssm, dI0_dpssm, dIt_dpssm);   /* :174-177 (Eigen arithmetic there) */
			if (params.jac_type == 0) am.cmptCurrJacobian(df_dp, mean_dI_dpssm);   /* :185-202 */
			else {
				am.cmptDifferenceOfJacobians(df_dp, dI0_dpssm, dIt_dpssm);
				for (int i = 0; i < (int)df_dp.size(); ++i) df_dp.data()[i] *= 0.5;
			}
			switch (params.hess_type) {   /* :204-259 */
			case InitialSelf: d2f_dp2 = init_d2f_dp2; break;
			case Original: am.cmptCurrHessian(d2f_dp2, mean_dI_dpssm); break;
			case SumOfStd: am.cmptSumOfHessians(d2f_dp2, dI0_dpssm, dIt_dpssm); scale(d2f_dp2, 0.5); break;
			case SumOfSelf:
				am.cmptSelfHessian(d2f_dp2, dIt_dpssm);
				for (size_t i = 0; i < (size_t)d2f_dp2.size(); ++i) d2f_dp2.data()[i] = (d2f_dp2.data()[i] + init_d2f_dp2.data()[i]) * 0.5;
				break;
			case CurrentSelf: am.cmptSelfHessian(d2f_dp2, dIt_dpssm); break;
			default: am.cmptCurrHessian(d2f_dp2, dIt_dpssm); break;
			}
			if (params.leven_marq) for (int i = 0; i < state_size; ++i) d2f_dp2(i, i) += lm_delta * d2f_dp2(i, i);   /* :261-264 */
			VectorXd x;
			utils::colPivHouseholderQrSolve(d2f_dp2, df_dp, x);                                                        /* :266 */
			for (int i = 0; i < state_size; ++i) state_update[i] = -x[i];
			for (int i = 0; i < ssm_state_size; ++i) ssm_update[i] = state_update[i];                                   /* :269-270 */
			for (int i = 0; i < am_state_size; ++i) am_update[i] = state_update[ssm_state_size + i];
			prev_corners = ssm.getCorners();
			ssm.compositionalUpdate(ssm_update);
			am.updateState(am_update);
			++iters;
			if (utils::squaredDistance(prev_corners, ssm.getCorners()) < params.epsilon) break;
			am.clearFirstIter();
		}
		return iters;
	}
private:
	ESMParams params;
	int ssm_state_size, am_state_size, state_size;
	MatrixXd dI0_dpssm, dIt_dpssm, mean_dI_dpssm, d2f_dp2, init_d2f_dp2;
	RowVectorXd df_dp;
	VectorXd state_update, ssm_update, am_update, inv_ssm_update, inv_am_update;
	CornersT prev_corners;
	static void scale(MatrixXd &m, double f) { for (size_t i = 0; i < (size_t)m.size(); ++i) m.data()[i] *= f; }
};

/* ICLK<AM, SSM> (SM/src/ICLK.cc:13-72 constructor, :74-112 initialize, :114-143 setRegion, :145-263 update): chained route only
 * (:196-198); first-order Hessians InitialSelf / CurrentSelf / Std; Levenberg-Marquardt with the undo :166-173. */
struct ICLKParams {
	int max_iters = 30; double epsilon = 1e-4;
	int hess_type = 0;   /* ICLKParams::HessType { InitialSelf, CurrentSelf, Std } */
	bool leven_marq = false; double lm_delta_init = 0.01, lm_delta_update = 10;
	bool update_ssm = false;   /* ICLKParams::update_ssm: setRegion recomputes the template Jacobian on the new grid (:116-140) */
};
template <class AM, class SSM>
class ICLK : public SearchMethod<AM, SSM> {
public:
	typedef typename SearchMethod<AM, SSM>::AMParams AMParams;
	typedef typename SearchMethod<AM, SSM>::SSMParams SSMParams;
	using SearchMethod<AM, SSM>::am;
	using SearchMethod<AM, SSM>::ssm;
	enum { InitialSelf, CurrentSelf, Std };
	ICLK(const ICLKParams *iclk_params, const AMParams *am_params, const SSMParams *ssm_params) :
		SearchMethod<AM, SSM>(am_params, ssm_params), params(*iclk_params) {
		ssm_state_size = (int)ssm.getStateSize();               /* :37-39 */
		am_state_size = am.getStateSize();
		state_size = ssm_state_size + am_state_size;
		dI0_dpssm.resize((int)am.getPatchSize(), ssm_state_size);                                          /* :50 */
		if (params.hess_type == CurrentSelf) dIt_dpssm.resize((int)am.getPatchSize(), ssm_state_size);     /* :51-53 */
		df_dp.resize(state_size); d2f_dp2.resize(state_size, state_size);                                  /* :54-55 */
		state_update.resize(state_size); ssm_update.resize(ssm_state_size); am_update.resize(am_state_size);
		inv_ssm_update.resize(ssm_state_size); inv_am_update.resize(am_state_size);
	}
	void initialize(const CornersT &corners) {   /* :74-112 */
		am.clearInitStatus(); ssm.clearInitStatus();
		ssm.initialize(corners, am.getNChannels());
		am.initializePixVals(ssm.getPts());
		am.initializeSimilarity();
		am.initializeGrad();
		am.initializeHess();
		am.initializePixGrad(ssm.getPts());
		ssm.cmptWarpedPixJacobian(dI0_dpssm, am.getInitPixGrad());
		am.cmptInitJacobian(df_dp, dI0_dpssm);
		if (params.hess_type == InitialSelf) {
			am.cmptSelfHessian(d2f_dp2, dI0_dpssm);
			if (params.leven_marq) d2f_dp2_orig = d2f_dp2;
		}
	}
	void setRegion(const CornersT &corners) override {   /* :114-143 */
		ssm.setCorners(corners);
		if (params.update_ssm) {
			ssm.cmptWarpedPixJacobian(dI0_dpssm, am.getInitPixGrad());
			am.cmptInitJacobian(df_dp, dI0_dpssm);
			if (params.hess_type == InitialSelf) {
				am.cmptSelfHessian(d2f_dp2, dI0_dpssm);
				if (params.leven_marq) d2f_dp2_orig = d2f_dp2;
			}
		}
	}
	int update() {   /* :145-263 */
		am.setFirstIter();
		double prev_f = 0, lm_delta = params.lm_delta_init;
		bool state_reset = false;
		int iters = 0;
		for (int iter_id = 0; iter_id < params.max_iters; ++iter_id) {
			am.updatePixVals(ssm.getPts());
			am.updateSimilarity(false);
			if (params.leven_marq && !state_reset) {   /* :161-180 */
				const double f = am.getSimilarity();
				if (iter_id > 0) {
					if (f < prev_f) {
						lm_delta *= params.lm_delta_update;
						ssm.compositionalUpdate(ssm_update);   /* the undo: the forward update of the inverse that was applied */
						am.updateState(am_update);
						state_reset = true;
						continue;
					}
					if (f > prev_f) lm_delta /= params.lm_delta_update;
				}
				prev_f = f;
			}
			state_reset = false;
			am.updateInitGrad();
			am.cmptInitJacobian(df_dp, dI0_dpssm);
			switch (params.hess_type) {   /* :190-221 */
			case InitialSelf: if (params.leven_marq) d2f_dp2 = d2f_dp2_orig; break;
			case CurrentSelf:
				am.updatePixGrad(ssm.getPts());
				ssm.cmptWarpedPixJacobian(dIt_dpssm, am.getCurrPixGrad());
				am.cmptSelfHessian(d2f_dp2, dIt_dpssm);
				break;
			default: am.cmptInitHessian(d2f_dp2, dI0_dpssm); break;
			}
			if (params.leven_marq) for (int i = 0; i < state_size; ++i) d2f_dp2(i, i) += lm_delta * d2f_dp2(i, i);   /* :222-225 */
			VectorXd x;
			utils::colPivHouseholderQrSolve(d2f_dp2, df_dp, x);                                                        /* :227 */
			for (int i = 0; i < state_size; ++i) state_update[i] = -x[i];
			for (int i = 0; i < ssm_state_size; ++i) ssm_update[i] = state_update[i];
			for (int i = 0; i < am_state_size; ++i) am_update[i] = state_update[ssm_state_size + i];
			prev_corners = ssm.getCorners();
			ssm.invertState(inv_ssm_update, ssm_update);                                                                /* :235-245 */
			ssm.compositionalUpdate(inv_ssm_update);
			am.invertState(inv_am_update, am_update);
			am.updateState(inv_am_update);
			++iters;
			if (utils::squaredDistance(prev_corners, ssm.getCorners()) < params.epsilon) break;
			am.clearFirstIter();
		}
		return iters;
	}
private:
	ICLKParams params;
	int ssm_state_size, am_state_size, state_size;
	MatrixXd dI0_dpssm, dIt_dpssm, d2f_dp2, d2f_dp2_orig;
	RowVectorXd df_dp;
	VectorXd state_update, ssm_update, am_update, inv_ssm_update, inv_am_update;
	CornersT prev_corners;
};

} // namespace templated
} // namespace mtf
#endif
