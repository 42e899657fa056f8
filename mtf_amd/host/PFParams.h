/*
 * PFParams.h -- the particle filter's parameter block: names, enum values and defaults of SM/include/mtf/SM/PFParams.h,
 * SM/src/PFParams.cc (product: mtf::hip::PF takes it; the harness's nt::PF as well).
 */
#ifndef MTF_AMD_HOST_PF_PARAMS_H
#define MTF_AMD_HOST_PF_PARAMS_H

#include <vector>

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
	/* PFParams.h: update_distr_wts; switched off for a single distribution (PF.cc:67).  With SEVERAL distributions and the flag off
	 * the reference zeroes the distribution weights after the first iteration and then draws every particle's distribution from an
	 * all-zero discrete distribution (NT/PF.cc:241-257: undefined).  One behaviour for every front end -- C ABI
	 * (mtfhip_pf_set_distributions), mtf::hip::PF, the harness's nt::PF, the Python wrapper, the oracle: REFUSED with that reason. */
	bool update_distr_wts = false;
	double min_distr_wt = 0.5;
	double adaptive_resampling_thresh = 0;     /* in (0, 1]: resample only when the effective particle count is <= thresh * n (PF.cc:381-390) */
	bool jacobian_as_sigma = false;            /* the sampler's sigma of every frame = the Gauss-Newton step (PF.cc:58-64, 156-165, 214-227) */
	double measurement_sigma = 0.1;
	bool enable_learning = false;
	unsigned long long seed = 0;               /* hip::PF: the device generator's key; nt::PF: 0 = random_device */
};

} // namespace mtf
#endif
