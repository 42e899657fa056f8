/*
 * SearchMethod.h -- the search-method interface the device-side drivers implement: mtf::nt::SearchMethod as the reference's
 * non-templated base (SM/include/mtf/SM/NT/SearchMethod.h:15-84: shared_ptr to the abstract AM / SSM, initialize / update /
 * setRegion / getRegion / setImage) and the parameter block with the reference's names and class defaults
 * (SM/src/ESMParams.cc:4-15, FCLKParams.cc:4-17, ICLKParams.cc:4-14).  Product: mtf::hip::LK and mtf::hip::PF derive from it.
 * The reference's own loops over the virtuals (nt::ESM / FCLK / ICLK / PF) are restated under harness/ -- test infrastructure
 * that drives the adapters the way an MTF build would, not part of libmtfhost.so.
 */
#ifndef MTF_AMD_HOST_SEARCH_METHOD_H
#define MTF_AMD_HOST_SEARCH_METHOD_H

#include <memory>
#include <string>

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
	SearchMethod(AM _am, SSM _ssm, const SMParams &_params) : am(_am), ssm(_ssm), params(_params), ssm_state_size((int)_ssm->getStateSize()) {}
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
};

} // namespace nt
} // namespace mtf
#endif
