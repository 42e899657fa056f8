/*
 * DevicePF.h -- mtf::hip::PF: the particle filter search method (SM/include/mtf/SM/NT/PF.h, SM/src/NT/PF.cc:136-620) with the
 * reference's parameters over the device filter of the C ABI (mtfhip_pf_*): all particles of an iteration in three launches.  This
 * is what a maintainer registers next to nt::PF for HipAM / HipSSM pairs; optional sharding of the scoring over a
 * communicator.  (The literal nt::PF over the virtuals -- one C-ABI round trip per particle -- is restated in harness/PF.h.)
 */
#ifndef MTF_AMD_HOST_DEVICE_PF_H
#define MTF_AMD_HOST_DEVICE_PF_H

#include "HipModels.h"
#include "PFParams.h"
#include "SearchMethod.h"

namespace mtf {

namespace hip {
class PF : public nt::SearchMethod {
public:
	PF(std::shared_ptr<HipAM> am, std::shared_ptr<HipSSM> ssm, const PFParams &params);
	~PF() override;
	void initialize(const CornersT &corners) override;
	void update() override;
	void setRegion(const CornersT &corners) override;
	const CornersT &getRegion() override;
	void setComm(mtfhip_comm *comm);   /* shard the scoring over the communicator's ranks (one RCCL all-gather per iteration) */
	/* the weights as peer stores of the scoring kernel instead of the all-gather (mtfhip.h: mtfhip_pf_set_exchange); with a detached
	 * communicator the host program moves the 64-byte handles: exportMailbox(mine), its own transport, connectMailboxes(all) */
	void setPeerExchange(bool on = true);
	void exportMailbox(void *handle64);
	void connectMailboxes(const void *handles /* world x 64 bytes */);
	mtfhip_pf *handle() { return h; }
private:
	std::shared_ptr<HipAM> ham;
	std::shared_ptr<HipSSM> hssm;
	PFParams pf;
	mtfhip_pf *h = nullptr;
	CornersT region;
	MatrixXd dI_dp, d2f_dp2;      /* jacobian_as_sigma: through the adapters' virtuals, the solve on the host */
	RowVectorXd df_dp;   /* (a row vector in the reference: cmptCurrJacobian(RowVectorXd &, ...)) */
	void jacobianSigma(bool init);
};
} // namespace hip

} // namespace mtf
#endif
