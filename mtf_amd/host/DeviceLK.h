/*
 * DeviceLK.h -- mtf::hip::LK: nt::ESM / nt::FCLK / nt::ICLK as ONE C-ABI call per update().
 *
 * mtf::nt::ESM / FCLK / ICLK (harness/SearchMethods.h: restated callers, test infrastructure) are the literal search methods: a loop over the AM / SSM virtuals, as
 * SM/src/NT/ESM.cc:170-296, NT/FCLK.cc:187-342 and NT/ICLK.cc:160-298 write it, which the library serves call by call
 * (20-37 us per loop pass for one target).  This class is the registration a maintainer adds next to them for a HipAM / HipSSM
 * pair (the counterpart of mtf::hip::PF): the same parameters -- the reference's class defaults included, Levenberg-Marquardt on --
 * and the same results, with the whole loop (pixel pass, g / H of the search method, damping, solve, compositional update,
 * corner test) on the device behind mtfhip_batch_init_template / mtfhip_batch_track / mtfhip_batch_set_region
 * (12-15 us per iteration for one target, and B targets per call for a batched pair).
 */
#ifndef MTF_AMD_HOST_DEVICE_LK_H
#define MTF_AMD_HOST_DEVICE_LK_H

#include "HipModels.h"
#include "SearchMethod.h"

namespace mtf {
namespace hip {

class LK : public nt::SearchMethod {
public:
	/* sm_kind: MTFHIP_SM_ESM / MTFHIP_SM_FCLK / MTFHIP_SM_ICLK */
	LK(int sm_kind, std::shared_ptr<HipAM> am, std::shared_ptr<HipSSM> ssm, const nt::SMParams &params);
	void initialize(const CornersT &corners) override;
	void update() override;
	void setRegion(const CornersT &corners) override;
	const CornersT &getRegion() override;
	const mtfhip_sm_desc &desc() const { return d; }
private:
	std::shared_ptr<HipAM> ham;
	std::shared_ptr<HipSSM> hssm;
	mtfhip_sm_desc d;
	CornersT region;
};

} // namespace hip
} // namespace mtf
#endif
