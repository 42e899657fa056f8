/* DevicePF.cpp -- see DevicePF.h */
#include "DevicePF.h"

#include <algorithm>
#include <cstring>

namespace mtf {

/* ------------------------------------------------------------------ hip::PF: the same search method over mtfhip_pf_* */
namespace hip {
PF::PF(std::shared_ptr<HipAM> a, std::shared_ptr<HipSSM> s, const PFParams &pp) : nt::SearchMethod(a, s, nt::SMParams()), ham(a), hssm(s), pf(pp) {
	name = "pf_hip";
	if (a->pair().get() != s->pair().get()) throw utils::InvalidArgument("hip::PF :: the AM and the SSM must share one HipPair");
	mtfhip_pf_desc d;
	std::memset(&d, 0, sizeof(d));
	d.n_particles = pf.n_particles; d.max_iters = pf.max_iters; d.epsilon = pf.epsilon;
	d.dynamic_model = (int)pf.dynamic_model; d.update_type = (int)pf.update_type; d.likelihood_func = (int)pf.likelihood_func;
	d.resampling_type = (int)pf.resampling_type; d.mean_type = (int)pf.mean_type;
	d.corner_based_sampling = s->getCornerBasedSampling() ? 1 : 0;
	d.pt_based_sampling = s->getPtBasedSampling();
	d.reset_to_mean = pf.reset_to_mean ? 1 : 0; d.measurement_sigma = pf.measurement_sigma; d.ar_coeff = 0.5; d.seed = pf.seed;
	const int S = ssm_state_size;
	const bool pix = !pf.jacobian_as_sigma && !pf.pix_sigma.empty() && pf.pix_sigma[0] > 0;   /* sigmas estimated at initialize(): ssm_sigma is not used */
	if (!pix && pf.ssm_sigma.empty()) throw utils::InvalidArgument("hip::PF :: ssm_sigma is empty and no pix_sigma is given");
	/* an entry of a row: a single value serves every state component (the reference's vectorvd rows may be short) */
	auto entry = [](const std::vector<double> &row, int k, double fallback) { return row.empty() ? fallback : (row.size() == 1 ? row[0] : row.at((size_t)k)); };
	for (int k = 0; k < S; ++k) {
		d.ssm_sigma[k] = pix ? 1.0 : entry(pf.ssm_sigma, k, 1.0);   /* (placeholder until initialize() knows the points) */
		d.ssm_mean[k] = entry(pf.ssm_mean, k, 0.0);
	}
	std::vector<std::vector<double>> sg{pix ? std::vector<double>(1, 1.0) : pf.ssm_sigma}, mn{pf.ssm_mean.empty() ? std::vector<double>(1, 0.0) : pf.ssm_mean};
	for (const auto &r : pf.more_sigma) sg.push_back(r);
	for (const auto &r : pf.more_mean) mn.push_back(r);
	if (pix) { sg.assign(pf.pix_sigma.size(), std::vector<double>(1, 1.0)); mn.assign(pf.pix_sigma.size(), std::vector<double>(1, 0.0)); }
	const int n_distr = pf.jacobian_as_sigma ? 1 : (int)std::max(sg.size(), mn.size());
	/* everything that can be refused is refused BEFORE the device filter exists (a constructor that throws runs no destructor) */
	if (n_distr > 8) throw utils::InvalidArgument("hip::PF :: at most eight sampler distributions");
	if (n_distr > 1 && !pf.update_distr_wts)
		throw utils::InvalidArgument("hip::PF :: several sampler distributions need update_distr_wts (without it the reference zeroes the weights and "
			"draws from an all-zero discrete distribution, NT/PF.cc:241-257: refused by every front end -- see PFParams.h)");
	d.adaptive_resampling_thresh = pf.adaptive_resampling_thresh;
	d.update_distr_wts = n_distr > 1 ? 1 : 0;
	d.min_distr_wt = pf.min_distr_wt;
	HipPair::check(mtfhip_pf_create(a->pair()->b, &d, &h));
	try {
		if (n_distr > 1) {
			std::vector<double> fs(8 * (size_t)n_distr, 0.0), fm(8 * (size_t)n_distr, 0.0);
			for (int i = 0; i < n_distr; ++i) {
				const std::vector<double> &rs = sg[std::min<size_t>(i, sg.size() - 1)], &rm = mn[std::min<size_t>(i, mn.size() - 1)];
				for (int k = 0; k < S; ++k) { fs[8 * i + k] = entry(rs, k, 1.0); fm[8 * i + k] = entry(rm, k, 0.0); }
			}
			HipPair::check(mtfhip_pf_set_distributions(h, n_distr, fs.data(), fm.data()));
		}
	} catch (...) { mtfhip_pf_destroy(h); h = nullptr; throw; }
	if (pf.jacobian_as_sigma) { dI_dp.resize(am->getPatchSize(), S); df_dp.resize(S); d2f_dp2.resize(S, S); }
}
PF::~PF() { if (h) mtfhip_pf_destroy(h); }
void PF::jacobianSigma(bool init) {   /* NT/PF.cc:156-165, 214-227 through the adapters' virtuals; the S x S solve on the host */
	const bool additive = pf.update_type == PFParams::UpdateType::Additive;
	if (init) {
		am->initializeGrad();
		am->initializePixGrad(ssm->getPts());
		if (additive) ssm->cmptPixJacobian(dI_dp, am->getInitPixGrad()); else ssm->cmptWarpedPixJacobian(dI_dp, am->getInitPixGrad());
		am->cmptSelfHessian(d2f_dp2, dI_dp);
		return;
	}
	am->updatePixVals(ssm->getPts());
	am->updateSimilarity();
	am->updateCurrGrad();
	am->updatePixGrad(ssm->getPts());
	if (additive) ssm->cmptPixJacobian(dI_dp, am->getCurrPixGrad()); else ssm->cmptWarpedPixJacobian(dI_dp, am->getCurrPixGrad());
	am->cmptCurrJacobian(df_dp, dI_dp);
	VectorXd x;
	utils::colPivHouseholderQrSolve(d2f_dp2, df_dp, x);
	double sigma[8] = {0}, mean[8] = {0};
	for (int k = 0; k < ssm_state_size; ++k) {
		sigma[k] = -x[k];
		mean[k] = pf.ssm_mean.empty() ? 0.0 : (pf.ssm_mean.size() == 1 ? pf.ssm_mean[0] : pf.ssm_mean.at((size_t)k));
	}
	HipPair::check(mtfhip_pf_set_sampler(h, sigma, mean));
}
void PF::setComm(mtfhip_comm *comm) { HipPair::check(mtfhip_pf_set_comm(h, comm)); }
void PF::setPeerExchange(bool on) { HipPair::check(mtfhip_pf_set_exchange(h, on ? MTFHIP_PF_EXCHANGE_PEER : MTFHIP_PF_EXCHANGE_COLLECTIVE)); }
void PF::exportMailbox(void *handle64) { HipPair::check(mtfhip_pf_exchange_export(h, handle64)); }
void PF::connectMailboxes(const void *handles) { HipPair::check(mtfhip_pf_exchange_connect(h, handles)); }
void PF::initialize(const CornersT &corners) {   /* NT/PF.cc:136-183 */
	am->clearInitStatus(); ssm->clearInitStatus();
	ssm->initialize(corners, am->getNChannels());
	am->initializePixVals(ssm->getPts());
	am->initializeSimilarity();
	if (!pf.jacobian_as_sigma && !pf.pix_sigma.empty() && pf.pix_sigma[0] > 0) {   /* NT/PF.cc:142-149 */
		const int n_distr = (int)pf.pix_sigma.size(), S = ssm_state_size;
		std::vector<double> fs(8 * (size_t)n_distr, 0.0), fm(8 * (size_t)n_distr, 0.0);
		VectorXd sg;
		for (int i = 0; i < n_distr; ++i) { ssm->estimateStateSigma(sg, pf.pix_sigma[i]); for (int k = 0; k < S; ++k) fs[8 * i + k] = sg[k]; }
		if (n_distr > 1) HipPair::check(mtfhip_pf_set_distributions(h, n_distr, fs.data(), fm.data()));
		else HipPair::check(mtfhip_pf_set_sampler(h, fs.data(), fm.data()));
	}
	if (pf.jacobian_as_sigma) jacobianSigma(true);
	HipPair::check(mtfhip_pf_initialize(h));
}
void PF::setRegion(const CornersT &corners) {   /* NT/PF.cc:616-620 */
	ssm->setCorners(corners);                                    /* (keeps the adapter's host mirrors current) */
	HipPair::check(mtfhip_pf_set_region(h, corners.data()));     /* same corners again + initializeParticles */
	hssm->markMoved();
}
void PF::update() {
	am->setFirstIter();
	if (pf.jacobian_as_sigma) jacobianSigma(false);
	HipPair::check(mtfhip_pf_update(h, &iters_done));
	hssm->markMoved();
	if (pf.enable_learning) {   /* NT/PF.cc:443-446 */
		am->updateModel(ssm->getPts());
		HipPair::check(mtfhip_pf_set_max_similarity(h, am->getSimilarity()));
	}
}
const CornersT &PF::getRegion() {
	HipPair::check(mtfhip_ssm_get_corners(ham->pair()->b, region.data()));
	return region;
}
} // namespace hip
} // namespace mtf
