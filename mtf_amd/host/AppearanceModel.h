/*
 * AppearanceModel.h -- the slice of mtf::ImageBase / mtf::AppearanceModel the search methods call, with the
 * reference's virtual names, argument meaning and error behaviour
 * (AM/include/mtf/AM/ImageBase.h:51-191, AM/include/mtf/AM/AppearanceModel.h:63-396).
 */
#ifndef MTF_AMD_HOST_APPEARANCE_MODEL_H
#define MTF_AMD_HOST_APPEARANCE_MODEL_H

#include "mtf_types.h"

#define am_func_not_implemeted(func_name) \
	throw mtf::utils::FunctonNotImplemented(name + "::" + #func_name + ":: Not implemented Yet")

namespace mtf {

class AppearanceModel {
public:
	std::string name;
	virtual ~AppearanceModel() {}

	/* ImageBase accessors (ImageBase.h:74-89) */
	virtual unsigned int getResX() const = 0;
	virtual unsigned int getResY() const = 0;
	virtual unsigned int getNPix() const = 0;
	virtual unsigned int getNChannels() const { return 1; }
	virtual unsigned int getPatchSize() const { return getNPix(); }
	virtual double getGradOffset() const = 0;
	virtual double getHessOffset() const = 0;
	virtual const PixValT &getInitPixVals() = 0;
	virtual const PixValT &getCurrPixVals() = 0;
	virtual const PixGradT &getInitPixGrad() = 0;
	virtual const PixGradT &getCurrPixGrad() = 0;
	virtual const PixHessT &getInitPixHess() = 0;
	virtual const PixHessT &getCurrPixHess() = 0;

	/* ImageBase modifiers / updaters (ImageBase.h:92-123) */
	virtual void setCurrImg(const ImageView &img) = 0;
	virtual void initializePixVals(const PtsT &init_pts) = 0;
	virtual void initializePixGrad(const GradPtsT &warped_offset_pts) = 0; /* overload 1: gradient of the warped image */
	virtual void initializePixGrad(const PtsT &init_pts) = 0;                           /* overload 2: warp of the image gradient */
	virtual void updatePixVals(const PtsT &curr_pts) = 0;
	virtual void updatePixGrad(const GradPtsT &warped_offset_pts) = 0;
	virtual void updatePixGrad(const PtsT &curr_pts) = 0;
	/* ImageBase.h:112-114,122-123 */
	virtual void initializePixHess(const PtsT &init_pts, const HessPtsT &warped_offset_pts) = 0;
	virtual void initializePixHess(const PtsT &init_pts) = 0;
	virtual void updatePixHess(const PtsT &curr_pts, const HessPtsT &warped_offset_pts) = 0;
	virtual void updatePixHess(const PtsT &curr_pts) = 0;

	/* AppearanceModel (AppearanceModel.h:77-219) */
	virtual int getStateSize() const { return 0; }
	virtual double getSimilarity() const = 0;
	virtual double getLikelihood() const { am_func_not_implemeted(getLikelihood); }
	virtual void initializeSimilarity() { am_func_not_implemeted(initializeSimilarity); }
	virtual void initializeGrad() { am_func_not_implemeted(initializeGrad); }
	virtual void initializeHess() { am_func_not_implemeted(initializeHess); }
	virtual void updateSimilarity(bool prereq_only = true) { am_func_not_implemeted(updateSimilarity); }
	virtual void updateState(const VectorXd &) {}
	virtual void invertState(VectorXd &, const VectorXd &) {}
	virtual void updateInitGrad() { am_func_not_implemeted(updateInitGrad); }
	virtual void updateCurrGrad() { am_func_not_implemeted(updateCurrGrad); }
	virtual void updateModel(const PtsT &) { am_func_not_implemeted(updateModel); }   /* AppearanceModel.h:261; SSD.cc:49-75, NCC.cc:539-566 */
	/* selective pixel integration (AppearanceModel.h:228-236): the device path does not take a mask, as the reference's own AMs
	 * outside SSDBase do not */
	virtual void setSPIMask(const bool *) { am_func_not_implemeted(setSPIMask); }
	virtual void clearSPIMask() {}
	virtual bool supportsSPI() const { return false; }
	/* distance features for the NN / FLANN search (AppearanceModel.h:266-297) */
	virtual void initializeDistFeat() { am_func_not_implemeted(initializeDistFeat); }
	virtual void updateDistFeat() { am_func_not_implemeted(updateDistFeat); }
	virtual void updateDistFeat(double *) { am_func_not_implemeted(updateDistFeat); }
	virtual const double *getDistFeat() { am_func_not_implemeted(getDistFeat); }
	virtual unsigned int getDistFeatSize() { am_func_not_implemeted(getDistFeatSize); }
	virtual void cmptInitJacobian(RowVectorXd &df_dp, const MatrixXd &dI0_dpssm) = 0;
	virtual void cmptCurrJacobian(RowVectorXd &df_dp, const MatrixXd &dIt_dpssm) = 0;
	virtual void cmptDifferenceOfJacobians(RowVectorXd &df_dp_diff, const MatrixXd &dI0_dpssm, const MatrixXd &dIt_dpssm) = 0;
	virtual void cmptInitHessian(MatrixXd &, const MatrixXd &) { am_func_not_implemeted(cmptInitHessian(first order)); }
	virtual void cmptCurrHessian(MatrixXd &, const MatrixXd &) { am_func_not_implemeted(cmptCurrHessian(first order)); }
	virtual void cmptSelfHessian(MatrixXd &, const MatrixXd &) { am_func_not_implemeted(cmptSelfHessian(first order)); }
	virtual void cmptSumOfHessians(MatrixXd &, const MatrixXd &, const MatrixXd &) { am_func_not_implemeted(cmptSumOfHessians); }
	/* second order variants (AppearanceModel.h:180-192,209-219): d2I_dpssm2 is the SM's S^2 x N pixel Hessian */
	virtual void cmptInitHessian(MatrixXd &, const MatrixXd &, const MatrixXd &) { am_func_not_implemeted(cmptInitHessian(second order)); }
	virtual void cmptCurrHessian(MatrixXd &, const MatrixXd &, const MatrixXd &) { am_func_not_implemeted(cmptCurrHessian(second order)); }
	virtual void cmptSelfHessian(MatrixXd &, const MatrixXd &, const MatrixXd &) { am_func_not_implemeted(cmptSelfHessian(second order)); }
	virtual void cmptSumOfHessians(MatrixXd &, const MatrixXd &, const MatrixXd &, const MatrixXd &, const MatrixXd &) { am_func_not_implemeted(cmptSumOfHessians(second order)); }

	/* EXTENSION (not a reference virtual): the ESM variants Original average two SM-owned matrices with plain Eigen
	 * arithmetic (mean_pix_jacobian NT/ESM.cc:239-242, mean_pix_hessian NT/ESM.cc:325).  A device-resident AM overrides
	 * this so that those N x S / S^2 x N matrices never have to visit the host; the default is that arithmetic. */
	virtual void cmptMeanOf(MatrixXd &mean, const MatrixXd &a, const MatrixXd &b) {
		for (size_t i = 0; i < mean.size(); ++i) mean.data()[i] = (a.data()[i] + b.data()[i]) / 2.0;
	}

	virtual void setFirstIter() { first_iter = true; }
	virtual void clearFirstIter() { first_iter = false; }
	virtual void clearInitStatus() = 0;
	virtual bool isSymmetrical() const { return true; }
protected:
	bool first_iter = false;
};

} // namespace mtf
#endif
