/*
 * SearchMethods.h -- nt::ESM, nt::FCLK, nt::ICLK written against the abstract AppearanceModel /
 * StateSpaceModel interface only (SM/include/mtf/SM/NT/SearchMethod.h:15-84; update loops
 * SM/src/NT/ESM.cc:170-296, NT/FCLK.cc:171-358, NT/ICLK.cc:160-299), parameters with the reference's names,
 * enums and class defaults (SM/src/ESMParams.cc:4-15, FCLKParams.cc:4-17, ICLKParams.cc:4-14).
 * They are the callers of the hot path: nothing here knows whether the AM / SSM run on a GPU.
 */
#ifndef MTF_AMD_HOST_SEARCH_METHODS_H
#define MTF_AMD_HOST_SEARCH_METHODS_H

#include <memory>

#include "AppearanceModel.h"
#include "StateSpaceModel.h"

namespace mtf {
namespace nt {

struct SMParams {
	int max_iters = 30;
	double epsilon = 1e-4;
	int jac_type = 1;          /* ESMParams::JacType { Original, DiffOfJacs } */
	int hess_type = -1;        /* per-SM enum; -1 = the SM's class default */
	bool sec_ord_hess = false;
	bool chained_warp = true;
	bool leven_marq = true;
	double lm_delta_init = 0.01;
	double lm_delta_update = 10;
	bool enable_learning = false;   /* ESM / FC / IC_ENABLE_LEARNING: am->updateModel(ssm->getPts()) after update() (NT/ESM.cc:293-295) */
};

class SearchMethod {
public:
	typedef std::shared_ptr<AppearanceModel> AM;
	typedef std::shared_ptr<StateSpaceModel> SSM;
	std::string name;
	SearchMethod(AM _am, SSM _ssm, const SMParams &_params);
	virtual ~SearchMethod() {}
	virtual void initialize(const CornersT &corners) = 0;
	virtual void update() = 0;
	virtual void setRegion(const CornersT &corners) { ssm->setCorners(corners); }
	virtual const CornersT &getRegion() { return ssm->getCorners(); }
	void setLearning(bool on) { params.enable_learning = on; }
	virtual void setImage(const ImageView &img) { am->setCurrImg(img); }
	int getItersDone() const { return iters_done; }
protected:
	AM am;
	SSM ssm;
	SMParams params;
	int ssm_state_size, iters_done = 0;
	MatrixXd init_pix_jacobian, curr_pix_jacobian, mean_pix_jacobian;
	MatrixXd init_pix_hessian, curr_pix_hessian, mean_pix_hessian;   /* S^2 x N, sec_ord_hess only */
	RowVectorXd jacobian;
	MatrixXd hessian, init_self_hessian;
	VectorXd state_update, inv_update;
	CornersT prev_corners;
	void initPixJacobian(MatrixXd &J);
	void updatePixJacobian(MatrixXd &J);
	void initPixHess();                          /* am->initializePixHess, either overload */
	void pixHessianFromInit(MatrixXd &D);        /* ssm->cmpt{Warped,Init}PixHessian of the template */
	void updatePixHessian(MatrixXd &D);          /* ESM::updatePixHessian */
	void selfHessian(MatrixXd &H, const MatrixXd &J, const MatrixXd &D);
	void dampAndSolve(double delta);    /* hessian += delta*diag(hessian); state_update = -H^-1 g */
};

class ESM : public SearchMethod {
public:
	enum HessType { InitialSelf, CurrentSelf, SumOfSelf, Original, SumOfStd, Std };
	ESM(AM am, SSM ssm, const SMParams &params);
	void initialize(const CornersT &corners) override;
	void update() override;
	void setRegion(const CornersT &corners) override;
};
class FCLK : public SearchMethod {
public:
	enum HessType { InitialSelf, CurrentSelf, Std };
	FCLK(AM am, SSM ssm, const SMParams &params);
	void initialize(const CornersT &corners) override;
	void update() override;
	void setRegion(const CornersT &corners) override;
};
class ICLK : public SearchMethod {
public:
	enum HessType { InitialSelf, CurrentSelf, Std };
	ICLK(AM am, SSM ssm, const SMParams &params);
	void initialize(const CornersT &corners) override;
	void update() override;
};

} // namespace nt
} // namespace mtf
#endif
