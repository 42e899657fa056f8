/*
 * SearchMethods.h -- nt::ESM, nt::FCLK, nt::ICLK written against the abstract AppearanceModel /
 * StateSpaceModel interface only (SM/include/mtf/SM/NT/SearchMethod.h:15-84; update loops
 * SM/src/NT/ESM.cc:170-296, NT/FCLK.cc:171-358, NT/ICLK.cc:160-299), parameters with the reference's names,
 * enums and class defaults (SM/src/ESMParams.cc:4-15, FCLKParams.cc:4-17, ICLKParams.cc:4-14).
 * They are the callers of the hot path: nothing here knows whether the AM / SSM run on a GPU.
 * HARNESS (libmtfharness.so), not product: condensed restatements of the reference's callers, kept so that the adapters can be
 * driven through the reference's own virtual-call sequence in tests and in the drop-in bench line.  An MTF build has the originals.
 */
#ifndef MTF_AMD_HOST_SEARCH_METHODS_H
#define MTF_AMD_HOST_SEARCH_METHODS_H

#include <memory>

#include "../SearchMethod.h"

namespace mtf {
namespace nt {

/* what the three Lucas-Kanade loops share (the reference keeps these members in every NT class): the SM-owned pixel Jacobians /
 * Hessians and the steps ESM, FCLK and ICLK all take */
class LKSearchMethod : public SearchMethod {
public:
	LKSearchMethod(AM _am, SSM _ssm, const SMParams &_params);
protected:
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

class ESM : public LKSearchMethod {
public:
	enum HessType { InitialSelf, CurrentSelf, SumOfSelf, Original, SumOfStd, Std };
	ESM(AM am, SSM ssm, const SMParams &params);
	void initialize(const CornersT &corners) override;
	void update() override;
	void setRegion(const CornersT &corners) override;
};
class FCLK : public LKSearchMethod {
public:
	enum HessType { InitialSelf, CurrentSelf, Std };
	FCLK(AM am, SSM ssm, const SMParams &params);
	void initialize(const CornersT &corners) override;
	void update() override;
	void setRegion(const CornersT &corners) override;
};
class ICLK : public LKSearchMethod {
public:
	enum HessType { InitialSelf, CurrentSelf, Std };
	ICLK(AM am, SSM ssm, const SMParams &params);
	void initialize(const CornersT &corners) override;
	void update() override;
};

} // namespace nt
} // namespace mtf
#endif
