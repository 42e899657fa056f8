/*
 * PF.h -- the particle filter search method.
 *   mtf::nt::PF   written against the abstract AppearanceModel / StateSpaceModel interface only, as the reference's
 *                 (SM/include/mtf/SM/NT/PF.h, SM/src/NT/PF.cc:136-620): per particle setState -> updatePixVals ->
 *                 updateSimilarity -> likelihood through the virtuals.  Works with any AM / SSM pair, the device adapters
 *                 included -- at one C-ABI round trip per particle, which is the reference's cost model, not the device's.
 *   mtf::hip::PF  the same search method with the same parameters over the device filter of the C ABI (mtfhip_pf_*):
 *                 all particles of an iteration in four launches.  This is what a maintainer registers next to nt::PF
 *                 for HipAM / HipSSM pairs; optional sharding of the scoring over a communicator.
 * Parameter names, enum values and defaults: SM/include/mtf/SM/PFParams.h, SM/src/PFParams.cc.
 */
#ifndef MTF_AMD_HOST_PF_H
#define MTF_AMD_HOST_PF_H

#include <random>

#include "HipModels.h"
#include "SearchMethods.h"

namespace mtf {

struct PFParams {
	enum class DynamicModel { RandomWalk, AutoRegression1 };
	enum class UpdateType { Additive, Compositional };
	enum class ResamplingType { None, BinaryMultinomial, LinearMultinomial, Residual };
	enum class LikelihoodFunc { AM, Gaussian, Reciprocal };
	enum class MeanType { None, SSM, Corners };
	int max_iters = 10;
	int n_particles = 200;
	double epsilon = 0.01;
	DynamicModel dynamic_model = DynamicModel::AutoRegression1;
	UpdateType update_type = UpdateType::Compositional;
	LikelihoodFunc likelihood_func = LikelihoodFunc::AM;
	ResamplingType resampling_type = ResamplingType::BinaryMultinomial;
	MeanType mean_type = MeanType::SSM;
	bool reset_to_mean = false;
	std::vector<double> ssm_sigma, ssm_mean;   /* the first distribution (the reference's vectorvd with a single entry) */
	/* further sampler distributions (PFParams::processDistributions, PFParams.cc:101-170: the shipped Config/modules.cfg:157 names
	 * five): with any, every particle draws its distribution from weights that follow the average particle weight each produced */
	std::vector<std::vector<double>> more_sigma, more_mean;
	std::vector<double> pix_sigma;             /* with pix_sigma[0] > 0: one sampler distribution per entry, its state sigma estimated by the
	                                              SSM at initialize() (PFParams.cc:105-116, PF.cc:142-149); ssm_sigma is then not used */
	bool update_distr_wts = false;             /* PFParams.h: update_distr_wts; switched off for a single distribution (PF.cc:67) */
	double min_distr_wt = 0.5;
	double adaptive_resampling_thresh = 0;     /* in (0, 1]: resample only when the effective particle count is <= thresh * n (PF.cc:381-390) */
	bool jacobian_as_sigma = false;            /* the sampler's sigma of every frame = the Gauss-Newton step (PF.cc:58-64, 156-165, 214-227) */
	double measurement_sigma = 0.1;
	bool enable_learning = false;
	unsigned long long seed = 0;               /* hip::PF: the device generator's key; nt::PF: 0 = random_device */
};

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
	VectorXd df_dp;
	bool enable_adaptive_resampling = false;
	double min_eff_particles = 0;
	void initializeDistributions();
	void jacobianSigma(bool init);
	CornersT mean_corners;
	double max_similarity = 0, measurement_factor = 1;
	std::mt19937_64 resample_gen;
	std::uniform_real_distribution<double> resample_dist{0.0, 1.0};
	void initializeParticles();
	void binaryMultinomialResampling();
	void linearMultinomialResampling();
	void updateMeanCorners();
};
} // namespace nt

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
	mtfhip_pf *handle() { return h; }
private:
	std::shared_ptr<HipAM> ham;
	std::shared_ptr<HipSSM> hssm;
	PFParams pf;
	mtfhip_pf *h = nullptr;
	CornersT region;
	MatrixXd dI_dp, d2f_dp2;      /* jacobian_as_sigma: through the adapters' virtuals, the solve on the host */
	VectorXd df_dp;
	void jacobianSigma(bool init);
};
} // namespace hip

} // namespace mtf
#endif
