/* DeviceLK.cpp -- see DeviceLK.h */
#include "DeviceLK.h"

#include <cstring>

namespace mtf {
namespace hip {

LK::LK(int sm_kind, std::shared_ptr<HipAM> a, std::shared_ptr<HipSSM> s, const nt::SMParams &pp) : nt::SearchMethod(a, s, pp), ham(a), hssm(s) {
	if (a->pair().get() != s->pair().get()) throw utils::InvalidArgument("hip::LK :: the AM and the SSM must share one HipPair");
	if (sm_kind != MTFHIP_SM_ESM && sm_kind != MTFHIP_SM_FCLK && sm_kind != MTFHIP_SM_ICLK)
		throw utils::InvalidArgument("hip::LK :: unknown search method");
	name = sm_kind == MTFHIP_SM_ESM ? "esm_hip" : (sm_kind == MTFHIP_SM_FCLK ? "fclk_hip" : "iclk_hip");
	std::memset(&d, 0, sizeof(d));
	d.sm = sm_kind;
	d.jac_type = params.jac_type;
	/* class defaults of the reference: ESMParams.cc:4-15 (SumOfSelf), FCLKParams.cc:4-17 (CurrentSelf), ICLKParams.cc:4-14 (InitialSelf) */
	d.hess_type = params.hess_type >= 0 ? params.hess_type : (sm_kind == MTFHIP_SM_ESM ? 2 : (sm_kind == MTFHIP_SM_FCLK ? 1 : 0));
	d.chained_warp = params.chained_warp ? 1 : 0;
	d.materialize = 0;   /* nobody reads It / dIt_dx / Jt between the iterations of a loop that runs on the device */
	d.max_iters = params.max_iters;
	d.epsilon = params.epsilon;
	d.leven_marq = params.leven_marq ? 1 : 0;
	d.lm_delta_init = params.lm_delta_init;
	d.lm_delta_update = params.lm_delta_update;
	d.sec_ord_hess = params.sec_ord_hess ? 1 : 0;
}

void LK::initialize(const CornersT &corners) {   /* NT/ESM.cc:110-146, NT/FCLK.cc:102-169, NT/ICLK.cc:71-128 */
	am->clearInitStatus(); ssm->clearInitStatus();
	ssm->initialize(corners, am->getNChannels());
	HipPair::check(mtfhip_batch_init_template(ham->pair()->b, &d));
	ham->markDeviceUpdated();
}

void LK::setRegion(const CornersT &corners) {   /* NT/ESM.cc:148-168, NT/FCLK.cc:360-376, NT/ICLK.cc:131-157 */
	ssm->setCorners(corners);                                                   /* (keeps the adapter's host mirrors current) */
	HipPair::check(mtfhip_batch_set_region(ham->pair()->b, corners.data(), &d));   /* same corners + the search method's refresh of J0 / H0 */
	hssm->markMoved();
	ham->markDeviceUpdated();
}

void LK::update() {
	am->setFirstIter();
	HipPair::check(mtfhip_batch_track(ham->pair()->b, &d, &iters_done, region.data()));
	hssm->markMoved();
	ham->markDeviceUpdated();
	if (params.enable_learning) am->updateModel(ssm->getPts());
}

const CornersT &LK::getRegion() {
	HipPair::check(mtfhip_ssm_get_corners(ham->pair()->b, region.data()));
	return region;
}

} // namespace hip
} // namespace mtf
