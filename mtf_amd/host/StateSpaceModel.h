/*
 * StateSpaceModel.h -- the slice of mtf::StateSpaceModel the search methods call
 * (SSM/include/mtf/SSM/StateSpaceModel.h:49-408), same names, argument meaning and error behaviour.
 */
#ifndef MTF_AMD_HOST_STATE_SPACE_MODEL_H
#define MTF_AMD_HOST_STATE_SPACE_MODEL_H

#include "mtf_types.h"

#define ssm_func_not_implemeted(func_name) \
	throw mtf::utils::FunctonNotImplemented(name + "::" + #func_name + ":: Not implemented Yet")

namespace mtf {

class StateSpaceModel {
public:
	std::string name;
	virtual ~StateSpaceModel() {}

	virtual unsigned int getStateSize() = 0;
	virtual unsigned int getResX() = 0;
	virtual unsigned int getResY() = 0;
	virtual unsigned int getNPts() = 0;
	virtual const PtsT &getPts() = 0;
	virtual const CornersT &getCorners() = 0;
	virtual const VectorXd &getState() = 0;
	virtual const GradPtsT &getGradPts() = 0;
	virtual const HessPtsT &getHessPts() = 0;

	virtual void setState(const VectorXd &) { ssm_func_not_implemeted(setState); }
	virtual void setCorners(const CornersT &) { ssm_func_not_implemeted(setCorners); }
	virtual void initialize(const CornersT &corners, int n_channels = 1) { (void)n_channels; setCorners(corners); }
	virtual void initializeGradPts(double grad_eps) { updateGradPts(grad_eps); }
	virtual void initializeHessPts(double hess_eps) { updateHessPts(hess_eps); }   /* StateSpaceModel.h:131-136 */
	virtual void additiveUpdate(const VectorXd &) { ssm_func_not_implemeted(additiveUpdate); }
	virtual void compositionalUpdate(const VectorXd &) { ssm_func_not_implemeted(compositionalUpdate); }
	virtual void updateGradPts(double) { ssm_func_not_implemeted(updateGradPts); }
	virtual void updateHessPts(double) { ssm_func_not_implemeted(updateHessPts); }
	virtual void invertState(VectorXd &, const VectorXd &) { ssm_func_not_implemeted(invertState); }

	virtual void cmptInitPixJacobian(MatrixXd &, const PixGradT &) { ssm_func_not_implemeted(cmptInitPixJacobian); }
	virtual void cmptPixJacobian(MatrixXd &, const PixGradT &) { ssm_func_not_implemeted(cmptPixJacobian); }
	virtual void cmptWarpedPixJacobian(MatrixXd &, const PixGradT &) { ssm_func_not_implemeted(cmptWarpedPixJacobian); }
	virtual void cmptApproxPixJacobian(MatrixXd &, const PixGradT &) { ssm_func_not_implemeted(cmptApproxPixJacobian); }
	/* StateSpaceModel.h:182-197 */
	virtual void cmptInitPixHessian(MatrixXd &, const PixHessT &, const PixGradT &) { ssm_func_not_implemeted(cmptInitPixHessian); }
	virtual void cmptPixHessian(MatrixXd &, const PixHessT &, const PixGradT &) { ssm_func_not_implemeted(cmptPixHessian); }
	virtual void cmptWarpedPixHessian(MatrixXd &, const PixHessT &, const PixGradT &) { ssm_func_not_implemeted(cmptWarpedPixHessian); }
	virtual void cmptApproxPixHessian(MatrixXd &, const PixHessT &, const PixGradT &) { ssm_func_not_implemeted(cmptApproxPixHessian); }
	virtual void applyWarpToCorners(CornersT &, const CornersT &, const VectorXd &) { ssm_func_not_implemeted(applyWarpToCorners); }
	virtual void applyWarpToPts(PtsT &, const PtsT &, const VectorXd &) { ssm_func_not_implemeted(applyWarpToPts); }
	virtual void getIdentityWarp(VectorXd &) { ssm_func_not_implemeted(getIdentityWarp); }                      /* StateSpaceModel.h:200-280 */
	virtual void composeWarps(VectorXd &, const VectorXd &, const VectorXd &) { ssm_func_not_implemeted(composeWarps); }
	virtual void estimateWarpFromCorners(VectorXd &, const CornersT &, const CornersT &) { ssm_func_not_implemeted(estimateWarpFromCorners); }

	/* ---- stochastic sampler (StateSpaceModel.h:286-338): what nt::PF and nt::NN call ---- */
	virtual void initializeSampler(const VectorXd &, const VectorXd &) { ssm_func_not_implemeted(initializeSampler(VectorXd, VectorXd)); }
	virtual void setSampler(const VectorXd &, const VectorXd &) { ssm_func_not_implemeted(setSampler); }
	virtual void setSamplerMean(const VectorXd &) { ssm_func_not_implemeted(setSamplerMean(VectorXd)); }
	virtual void setSamplerSigma(const VectorXd &) { ssm_func_not_implemeted(setSamplerSigma(VectorXd)); }
	virtual VectorXd getSamplerSigma() { ssm_func_not_implemeted(getSamplerSigma); }
	virtual VectorXd getSamplerMean() { ssm_func_not_implemeted(getSamplerMean); }
	virtual void compositionalRandomWalk(VectorXd &, const VectorXd &) { ssm_func_not_implemeted(compositionalRandomWalk); }
	virtual void additiveRandomWalk(VectorXd &, const VectorXd &) { ssm_func_not_implemeted(additiveRandomWalk); }
	virtual void compositionalAutoRegression1(VectorXd &, VectorXd &, const VectorXd &, const VectorXd &, double = 0.5) {
		ssm_func_not_implemeted(compositionalAutoRegression1);
	}
	virtual void additiveAutoRegression1(VectorXd &, VectorXd &, const VectorXd &, const VectorXd &, double = 0.5) {
		ssm_func_not_implemeted(additiveAutoRegression1);
	}
	virtual void generatePerturbation(VectorXd &) { ssm_func_not_implemeted(generatePerturbation); }
	virtual void estimateMeanOfSamples(VectorXd &, const std::vector<VectorXd> &, int) { ssm_func_not_implemeted(estimateMeanOfSamples); }
	virtual void estimateStateSigma(VectorXd &, double) { ssm_func_not_implemeted(estimateStateSigma); }

	virtual void setFirstIter() { first_iter = true; }
	virtual void clearFirstIter() { first_iter = false; }
	virtual void clearInitStatus() {}
protected:
	bool first_iter = false;
};

} // namespace mtf
#endif
