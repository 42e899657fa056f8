#include "SearchMethods.h"

#include <cmath>

namespace mtf {

namespace nt {

LKSearchMethod::LKSearchMethod(AM _am, SSM _ssm, const SMParams &_params) : SearchMethod(_am, _ssm, _params) {
	const int n = (int)am->getPatchSize();
	init_pix_jacobian.resize(n, ssm_state_size);
	curr_pix_jacobian.resize(n, ssm_state_size);
	jacobian.resize(ssm_state_size);
	hessian.resize(ssm_state_size, ssm_state_size);
	init_self_hessian.resize(ssm_state_size, ssm_state_size);
	state_update.resize(ssm_state_size);
	inv_update.resize(ssm_state_size);
}

/* ESM::initializePixJacobian NT/ESM.cc:379-388 (same shape in FCLK :113-133 and ICLK :78-95) */
void LKSearchMethod::initPixJacobian(MatrixXd &J) {
	if (params.chained_warp) {
		am->initializePixGrad(ssm->getPts());
		ssm->cmptWarpedPixJacobian(J, am->getInitPixGrad());
	} else {
		ssm->initializeGradPts(am->getGradOffset());
		am->initializePixGrad(ssm->getGradPts());
		ssm->cmptInitPixJacobian(J, am->getInitPixGrad());
	}
}
/* ESM::updatePixJacobian NT/ESM.cc:390-408 */
void LKSearchMethod::updatePixJacobian(MatrixXd &J) {
	if (params.chained_warp) {
		am->updatePixGrad(ssm->getPts());
		ssm->cmptWarpedPixJacobian(J, am->getCurrPixGrad());
	} else {
		ssm->updateGradPts(am->getGradOffset());
		am->updatePixGrad(ssm->getGradPts());
		ssm->cmptInitPixJacobian(J, am->getCurrPixGrad());
	}
}
/* ESM::initializePixHessian NT/ESM.cc:406-416 ; FCLK NT/FCLK.cc:121-142 ; ICLK NT/ICLK.cc:96-113 */
void LKSearchMethod::initPixHess() {
	if (params.chained_warp) am->initializePixHess(ssm->getPts());
	else {
		ssm->initializeHessPts(am->getHessOffset());
		am->initializePixHess(ssm->getPts(), ssm->getHessPts());
	}
}
void LKSearchMethod::pixHessianFromInit(MatrixXd &D) {
	if (params.chained_warp) ssm->cmptWarpedPixHessian(D, am->getInitPixHess(), am->getInitPixGrad());
	else ssm->cmptInitPixHessian(D, am->getInitPixHess(), am->getInitPixGrad());
}
/* ESM::updatePixHessian NT/ESM.cc:418-432 ; FCLK NT/FCLK.cc:243-257 ; ICLK NT/ICLK.cc:223-237 */
void LKSearchMethod::updatePixHessian(MatrixXd &D) {
	if (params.chained_warp) {
		am->updatePixHess(ssm->getPts());
		ssm->cmptWarpedPixHessian(D, am->getCurrPixHess(), am->getCurrPixGrad());
	} else {
		ssm->updateHessPts(am->getHessOffset());
		am->updatePixHess(ssm->getPts(), ssm->getHessPts());
		ssm->cmptInitPixHessian(D, am->getCurrPixHess(), am->getCurrPixGrad());
	}
}
void LKSearchMethod::selfHessian(MatrixXd &H, const MatrixXd &J, const MatrixXd &D) {
	if (params.sec_ord_hess) am->cmptSelfHessian(H, J, D);
	else am->cmptSelfHessian(H, J);
}
void LKSearchMethod::dampAndSolve(double delta) {
	if (params.leven_marq)
		for (int i = 0; i < ssm_state_size; ++i) hessian(i, i) += delta * hessian(i, i);
	utils::colPivHouseholderQrSolve(hessian, jacobian, state_update);
	for (int i = 0; i < ssm_state_size; ++i) state_update[i] = -state_update[i];
}

/* ------------------------------------------------------------------ ESM */
ESM::ESM(AM a, SSM s, const SMParams &p) : LKSearchMethod(a, s, p) {
	name = "esm_nt";
	if (params.hess_type < 0) params.hess_type = SumOfSelf;
	const int n = (int)am->getPatchSize(), s2 = ssm_state_size * ssm_state_size;
	if (params.jac_type == 0 || params.hess_type == Original) mean_pix_jacobian.resize(n, ssm_state_size);
	if (params.sec_ord_hess) {   /* NT/ESM.cc:99-107 */
		init_pix_hessian.resize(s2, n);
		if (params.hess_type != InitialSelf) {
			curr_pix_hessian.resize(s2, n);
			if (params.hess_type == Original) mean_pix_hessian.resize(s2, n);
		}
	}
}
void ESM::initialize(const CornersT &corners) {
	am->clearInitStatus(); ssm->clearInitStatus();
	ssm->initialize(corners, am->getNChannels());
	am->initializePixVals(ssm->getPts());
	initPixJacobian(init_pix_jacobian);
	if (params.sec_ord_hess) { initPixHess(); pixHessianFromInit(init_pix_hessian); }
	am->initializeSimilarity(); am->initializeGrad(); am->initializeHess();
	if (params.hess_type == InitialSelf || params.hess_type == SumOfSelf) {
		selfHessian(hessian, init_pix_jacobian, init_pix_hessian);
		init_self_hessian = hessian;
	}
}
void ESM::setRegion(const CornersT &corners) {
	ssm->setCorners(corners);
	ssm->cmptInitPixJacobian(init_pix_jacobian, am->getInitPixGrad());
	if (params.sec_ord_hess) ssm->cmptInitPixHessian(init_pix_hessian, am->getInitPixHess(), am->getInitPixGrad());
	if (params.hess_type == InitialSelf || params.hess_type == SumOfSelf) {
		selfHessian(hessian, init_pix_jacobian, init_pix_hessian);
		init_self_hessian = hessian;
	}
}
void ESM::update() {
	double prev_similarity = 0, leven_marq_delta = params.lm_delta_init;
	bool state_reset = false;
	iters_done = 0;
	am->setFirstIter();
	for (int iter_id = 0; iter_id < params.max_iters; ++iter_id) {
		++iters_done;
		am->updatePixVals(ssm->getPts());
		am->updateSimilarity(false);
		if (params.leven_marq && !state_reset) {
			const double curr_similarity = am->getSimilarity();
			if (iter_id > 0) {
				if (curr_similarity < prev_similarity) {
					leven_marq_delta *= params.lm_delta_update;
					ssm->invertState(inv_update, state_update);
					ssm->compositionalUpdate(inv_update);
					state_reset = true;
					continue;
				}
				if (curr_similarity > prev_similarity) leven_marq_delta /= params.lm_delta_update;
			}
			prev_similarity = curr_similarity;
		}
		state_reset = false;
		updatePixJacobian(curr_pix_jacobian);
		/* mean_pix_jacobian = (init_pix_jacobian + curr_pix_jacobian) / 2.0, NT/ESM.cc:239-242 */
		if (params.jac_type == 0 || params.hess_type == Original) am->cmptMeanOf(mean_pix_jacobian, init_pix_jacobian, curr_pix_jacobian);
		if (params.sec_ord_hess && params.hess_type != InitialSelf) updatePixHessian(curr_pix_hessian);
		am->updateCurrGrad();
		am->updateInitGrad();
		if (params.jac_type == 0) am->cmptCurrJacobian(jacobian, mean_pix_jacobian);    /* cmptJacobian NT/ESM.cc:298-313 */
		else {
			am->cmptDifferenceOfJacobians(jacobian, init_pix_jacobian, curr_pix_jacobian);
			for (int i = 0; i < ssm_state_size; ++i) jacobian[i] *= 0.5;
		}
		switch (params.hess_type) {                                                       /* cmptHessian NT/ESM.cc:315-377 */
		case InitialSelf: if (params.leven_marq) hessian = init_self_hessian; break;
		case Original:
			if (params.sec_ord_hess) {
				am->cmptMeanOf(mean_pix_hessian, init_pix_hessian, curr_pix_hessian);
				am->cmptCurrHessian(hessian, mean_pix_jacobian, mean_pix_hessian);
			} else am->cmptCurrHessian(hessian, mean_pix_jacobian);
			break;
		case SumOfStd:
			if (params.sec_ord_hess) am->cmptSumOfHessians(hessian, init_pix_jacobian, curr_pix_jacobian, init_pix_hessian, curr_pix_hessian);
			else am->cmptSumOfHessians(hessian, init_pix_jacobian, curr_pix_jacobian);
			for (int i = 0; i < ssm_state_size; ++i) for (int j = 0; j < ssm_state_size; ++j) hessian(i, j) *= 0.5;
			break;
		case SumOfSelf:
			selfHessian(hessian, curr_pix_jacobian, curr_pix_hessian);
			for (int i = 0; i < ssm_state_size; ++i) for (int j = 0; j < ssm_state_size; ++j) hessian(i, j) = (hessian(i, j) + init_self_hessian(i, j)) * 0.5;
			break;
		case CurrentSelf: selfHessian(hessian, curr_pix_jacobian, curr_pix_hessian); break;
		default:
			if (params.sec_ord_hess) am->cmptCurrHessian(hessian, curr_pix_jacobian, curr_pix_hessian);
			else am->cmptCurrHessian(hessian, curr_pix_jacobian);
			break;
		}
		dampAndSolve(leven_marq_delta);
		prev_corners = ssm->getCorners();
		ssm->compositionalUpdate(state_update);
		const double update_norm = utils::squaredDistance(prev_corners, ssm->getCorners());
		if (update_norm < params.epsilon) break;
		am->clearFirstIter();
	}
	if (params.enable_learning) am->updateModel(ssm->getPts());
}

/* ------------------------------------------------------------------ FCLK */
FCLK::FCLK(AM a, SSM s, const SMParams &p) : LKSearchMethod(a, s, p) {
	name = "fclk_nt";
	if (params.hess_type < 0) params.hess_type = CurrentSelf;
	if (params.sec_ord_hess) {   /* NT/FCLK.cc:66-75 */
		const int n = (int)am->getPatchSize(), s2 = ssm_state_size * ssm_state_size;
		if (params.hess_type == InitialSelf) init_pix_hessian.resize(s2, n);
		else curr_pix_hessian.resize(s2, n);
	}
}
void FCLK::initialize(const CornersT &corners) {
	am->clearInitStatus(); ssm->clearInitStatus();
	ssm->initialize(corners, am->getNChannels());
	am->initializePixVals(ssm->getPts());
	am->initializeSimilarity(); am->initializeGrad(); am->initializeHess();
	if (params.chained_warp) am->initializePixGrad(ssm->getPts());
	else { ssm->initializeGradPts(am->getGradOffset()); am->initializePixGrad(ssm->getGradPts()); }
	if (params.sec_ord_hess) initPixHess();
	if (params.hess_type == InitialSelf) {
		if (params.chained_warp) ssm->cmptWarpedPixJacobian(init_pix_jacobian, am->getInitPixGrad());
		else ssm->cmptInitPixJacobian(init_pix_jacobian, am->getInitPixGrad());
		if (params.sec_ord_hess) pixHessianFromInit(init_pix_hessian);
		selfHessian(hessian, init_pix_jacobian, init_pix_hessian);
		if (params.leven_marq) init_self_hessian = hessian;
	}
}
/* NT/FCLK.cc:360-376: the recomputed Hessian is NOT copied to init_self_hessian there */
void FCLK::setRegion(const CornersT &corners) {
	ssm->setCorners(corners);
	if (params.hess_type == InitialSelf) {
		ssm->cmptInitPixJacobian(init_pix_jacobian, am->getInitPixGrad());
		if (params.sec_ord_hess) ssm->cmptInitPixHessian(init_pix_hessian, am->getInitPixHess(), am->getInitPixGrad());
		selfHessian(hessian, init_pix_jacobian, init_pix_hessian);
	}
}
void FCLK::update() {
	double prev_similarity = 0, leven_marq_delta = params.lm_delta_init;
	bool state_reset = false;
	iters_done = 0;
	am->setFirstIter();
	int iter_id = 0;
	while (iter_id < params.max_iters) {
		++iters_done;
		am->updatePixVals(ssm->getPts());
		am->updateSimilarity(false);
		if (params.leven_marq && !state_reset) {
			const double curr_similarity = am->getSimilarity();
			if (iter_id > 0) {
				if (curr_similarity < prev_similarity) {
					leven_marq_delta *= params.lm_delta_update;
					ssm->invertState(inv_update, state_update);
					ssm->compositionalUpdate(inv_update);
					state_reset = true;
					continue;
				}
				if (curr_similarity > prev_similarity) leven_marq_delta /= params.lm_delta_update;
			}
			prev_similarity = curr_similarity;
		}
		state_reset = false;
		am->updateCurrGrad();
		updatePixJacobian(curr_pix_jacobian);
		if (params.sec_ord_hess && params.hess_type != InitialSelf) updatePixHessian(curr_pix_hessian);
		am->cmptCurrJacobian(jacobian, curr_pix_jacobian);
		switch (params.hess_type) {
		case InitialSelf: if (params.leven_marq) hessian = init_self_hessian; break;
		case CurrentSelf: selfHessian(hessian, curr_pix_jacobian, curr_pix_hessian); break;
		default:
			if (params.sec_ord_hess) am->cmptCurrHessian(hessian, curr_pix_jacobian, curr_pix_hessian);
			else am->cmptCurrHessian(hessian, curr_pix_jacobian);
			break;
		}
		dampAndSolve(leven_marq_delta);
		prev_corners = ssm->getCorners();
		ssm->compositionalUpdate(state_update);
		const double update_norm = utils::squaredDistance(prev_corners, ssm->getCorners());
		if (update_norm < params.epsilon) break;
		am->clearFirstIter();
		++iter_id;
	}
	if (params.enable_learning) am->updateModel(ssm->getPts());   /* NT/FCLK.cc:352-354 */
}

/* ------------------------------------------------------------------ ICLK */
ICLK::ICLK(AM a, SSM s, const SMParams &p) : LKSearchMethod(a, s, p) {
	name = "iclk_nt";
	if (params.hess_type < 0) params.hess_type = InitialSelf;
	if (params.sec_ord_hess) {   /* NT/ICLK.cc:61-67 */
		const int n = (int)am->getPatchSize(), s2 = ssm_state_size * ssm_state_size;
		if (params.hess_type == CurrentSelf) curr_pix_hessian.resize(s2, n);
		else init_pix_hessian.resize(s2, n);
	}
}
void ICLK::initialize(const CornersT &corners) {
	am->clearInitStatus(); ssm->clearInitStatus();
	ssm->initialize(corners, am->getNChannels());
	am->initializePixVals(ssm->getPts());
	initPixJacobian(init_pix_jacobian);
	am->initializeSimilarity(); am->initializeGrad(); am->initializeHess();
	am->cmptInitJacobian(jacobian, init_pix_jacobian);
	if (params.sec_ord_hess) {
		initPixHess();
		if (params.hess_type != CurrentSelf) pixHessianFromInit(init_pix_hessian);
	}
	if (params.hess_type == InitialSelf) {
		selfHessian(hessian, init_pix_jacobian, init_pix_hessian);
		if (params.leven_marq) init_self_hessian = hessian;
	}
}
void ICLK::update() {
	double prev_similarity = 0, leven_marq_delta = params.lm_delta_init;
	bool state_reset = false;
	iters_done = 0;
	am->setFirstIter();
	for (int iter_id = 0; iter_id < params.max_iters; ++iter_id) {
		++iters_done;
		am->updatePixVals(ssm->getPts());
		am->updateSimilarity(false);
		if (params.leven_marq && !state_reset) {
			const double curr_similarity = am->getSimilarity();
			if (iter_id > 0) {
				if (curr_similarity < prev_similarity) {
					leven_marq_delta *= params.lm_delta_update;
					ssm->compositionalUpdate(state_update);   /* undo of the inverse update, NT/ICLK.cc:188 */
					state_reset = true;
					continue;
				}
				if (curr_similarity > prev_similarity) leven_marq_delta /= params.lm_delta_update;
			}
			prev_similarity = curr_similarity;
		}
		state_reset = false;
		am->updateInitGrad();
		am->cmptInitJacobian(jacobian, init_pix_jacobian);
		switch (params.hess_type) {
		case InitialSelf: if (params.leven_marq) hessian = init_self_hessian; break;
		case CurrentSelf:
			updatePixJacobian(curr_pix_jacobian);
			if (params.sec_ord_hess) updatePixHessian(curr_pix_hessian);
			selfHessian(hessian, curr_pix_jacobian, curr_pix_hessian);
			break;
		default:
			if (params.sec_ord_hess) am->cmptInitHessian(hessian, init_pix_jacobian, init_pix_hessian);
			else am->cmptInitHessian(hessian, init_pix_jacobian);
			break;
		}
		dampAndSolve(leven_marq_delta);
		prev_corners = ssm->getCorners();
		ssm->invertState(inv_update, state_update);
		ssm->compositionalUpdate(inv_update);
		const double update_norm = utils::squaredDistance(prev_corners, ssm->getCorners());
		if (update_norm < params.epsilon) break;
		am->clearFirstIter();
	}
	if (params.enable_learning) am->updateModel(ssm->getPts());
}

} // namespace nt
} // namespace mtf
