/*
 * PF.h (harness) -- mtf::nt::PF written against the abstract AppearanceModel / StateSpaceModel interface only, as the reference's
 * (SM/include/mtf/SM/NT/PF.h, SM/src/NT/PF.cc:136-620): per particle setState -> updatePixVals -> updateSimilarity -> likelihood
 * through the virtuals.  Works with any AM / SSM pair, the device adapters included -- at one C-ABI round trip per particle, which is
 * the reference's cost model, not the device's.  A condensed restatement of the reference's caller: test infrastructure
 * (libmtfharness.so), not product; the product's filter is mtf::hip::PF (../DevicePF.h).
 */
#ifndef MTF_AMD_HOST_PF_H
#define MTF_AMD_HOST_PF_H

#include <random>

#include "../PFParams.h"
#include "SearchMethods.h"

namespace mtf {

namespace nt {
class PF : public SearchMethod {
public:
	PF(AM am, SSM ssm, const PFParams &params);
	void initialize(const CornersT &corners) override;
	void update() override;
	void setRegion(const CornersT &corners) override;
	const std::vector<VectorXd> &getParticles() const { return particle_states[curr_set_id]; }
	const VectorXd &getWeights() const { return particle_wts; }
protected:
	PFParams pf;
	std::vector<VectorXd> particle_states[2], particle_ar[2];
	int curr_set_id = 0, max_wt_id = 0;
	VectorXd particle_wts, particle_cum_wts, perturbed_state, perturbed_ar, mean_state;
	std::vector<VectorXd> state_sigma, state_mean;   /* [n_distr] */
	int n_distr = 1;
	bool using_pix_sigma = false;
	std::vector<double> distr_wts;
	std::vector<int> distr_n_particles, particle_distr;
	std::mt19937_64 distr_id_gen;
	MatrixXd dI_dp, d2f_dp2;      /* jacobian_as_sigma */
	RowVectorXd df_dp;   /* (a row vector in the reference: cmptCurrJacobian(RowVectorXd &, ...)) */
	bool enable_adaptive_resampling = false;
	double min_eff_particles = 0;
	void initializeDistributions();
	void jacobianSigma(bool init);
	CornersT mean_corners, prev_corners;
	double max_similarity = 0, measurement_factor = 1;
	std::mt19937_64 resample_gen;
	std::uniform_real_distribution<double> resample_dist{0.0, 1.0};
	void initializeParticles();
	void binaryMultinomialResampling();
	void linearMultinomialResampling();
	void updateMeanCorners();
};
} // namespace nt


} // namespace mtf
#endif
