#include "PF.h"

#include <cmath>
#include <cstring>
#include <limits>

namespace mtf {
namespace nt {

/* ------------------------------------------------------------------ nt::PF over the virtual interface (NT/PF.cc) */
PF::PF(AM a, SSM s, const PFParams &pp) : SearchMethod(a, s, SMParams()), pf(pp) {
	name = "pf_nt";
	const int S = ssm_state_size;
	/* PFParams::processDistributions (PFParams.cc:101-170): n_distr = max(#sigma, #mean) rows; a single entry is broadcast, the last
	 * row is reused when one list is shorter; a wrong size is an error */
	std::vector<std::vector<double>> sg{pf.ssm_sigma}, mn{pf.ssm_mean.empty() ? std::vector<double>(1, 0.0) : pf.ssm_mean};
	for (const auto &r : pf.more_sigma) sg.push_back(r);
	for (const auto &r : pf.more_mean) mn.push_back(r);
	using_pix_sigma = !pf.pix_sigma.empty() && pf.pix_sigma[0] > 0;   /* PFParams.cc:105-116 */
	if (using_pix_sigma) { sg.assign(pf.pix_sigma.size(), std::vector<double>(1, 1.0)); mn.assign(pf.pix_sigma.size(), std::vector<double>(1, 0.0)); }
	if (pf.jacobian_as_sigma) using_pix_sigma = false;   /* PF.cc:58-64 */
	n_distr = pf.jacobian_as_sigma ? 1 : (int)std::max(sg.size(), mn.size());   /* PF.cc:58-63 */
	state_sigma.assign(n_distr, VectorXd(S)); state_mean.assign(n_distr, VectorXd(S));
	for (int i = 0; i < n_distr; ++i) {
		const std::vector<double> &rs = sg[std::min<size_t>(i, sg.size() - 1)], &rm = mn[std::min<size_t>(i, mn.size() - 1)];
		if (rs.size() != 1 && (int)rs.size() < S) throw utils::InvalidArgument("PFParams :: SSM sigma for distribution " + std::to_string(i) + " has invalid size: " + std::to_string(rs.size()));
		if (rm.size() != 1 && (int)rm.size() < S) throw utils::InvalidArgument("PFParams :: SSM mean for distribution " + std::to_string(i) + " has invalid size: " + std::to_string(rm.size()));
		for (int k = 0; k < S; ++k) { state_sigma[i][k] = rs.size() == 1 ? rs[0] : rs[k]; state_mean[i][k] = rm.size() == 1 ? rm[0] : rm[k]; }
	}
	if (n_distr == 1) pf.update_distr_wts = false;   /* PF.cc:67 */
	else if (!pf.update_distr_wts)   /* (one behaviour for every front end: PFParams.h) */
		throw utils::InvalidArgument("nt::PF :: several sampler distributions need update_distr_wts (the reference draws from an all-zero discrete distribution without it, NT/PF.cc:241-257)");
	distr_wts.assign(n_distr, 0.0); distr_n_particles.assign(n_distr, 0); particle_distr.assign(pf.n_particles, 0);
	if (pf.adaptive_resampling_thresh > 0 && pf.adaptive_resampling_thresh <= 1) {   /* PF.cc:114-118 */
		enable_adaptive_resampling = true;
		min_eff_particles = pf.adaptive_resampling_thresh * pf.n_particles;
	}
	if (pf.jacobian_as_sigma) { dI_dp.resize(am->getPatchSize(), S); df_dp.resize(S); d2f_dp2.resize(S, S); }
	const double pi = 3.14159265358979323846;
	measurement_factor = 1.0 / std::sqrt(2 * pi * pf.measurement_sigma);   /* :69-70 */
	for (int set_id = 0; set_id < 2; ++set_id) {
		particle_states[set_id].assign(pf.n_particles, VectorXd(S));
		particle_ar[set_id].assign(pf.n_particles, VectorXd(S));
	}
	particle_wts.resize(pf.n_particles); particle_cum_wts.resize(pf.n_particles);
	perturbed_state.resize(S); perturbed_ar.resize(S); mean_state.resize(S);
	if (pf.seed) { resample_gen.seed(pf.seed); distr_id_gen.seed(pf.seed ^ 0x9E3779B97F4A7C15ull); }
	else { std::random_device r; std::seed_seq seq{r(), r(), r(), r(), r(), r(), r(), r()}; resample_gen.seed(seq); distr_id_gen.seed(r()); }
}
void PF::initializeDistributions() {   /* :199-205 */
	for (int i = 0; i < n_distr; ++i) { distr_wts[i] = 1.0 / n_distr; distr_n_particles[i] = 0; }
}
/* jacobian_as_sigma: d2f_dp2 = the self Hessian of the template's pixel Jacobian once (:156-165); every frame the sampler's sigma =
 * -d2f_dp2^-1 df_dp on the current image (:214-227) */
void PF::jacobianSigma(bool init) {
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
	for (int k = 0; k < ssm_state_size; ++k) state_sigma[0][k] = -x[k];
	ssm->setSampler(state_sigma[0], state_mean[0]);
}
void PF::initialize(const CornersT &corners) {   /* :136-183 */
	am->clearInitStatus(); ssm->clearInitStatus();
	ssm->initialize(corners, am->getNChannels());
	if (using_pix_sigma)   /* :142-149 */
		for (int i = 0; i < n_distr; ++i) { state_mean[i].fill(0.0); ssm->estimateStateSigma(state_sigma[i], pf.pix_sigma[i]); }
	ssm->initializeSampler(state_sigma[0], state_mean[0]);
	am->initializePixVals(ssm->getPts());
	am->initializeSimilarity();
	max_similarity = am->getSimilarity();
	if (pf.jacobian_as_sigma) jacobianSigma(true);
	initializeParticles();
	initializeDistributions();
	prev_corners = ssm->getCorners();
}
void PF::initializeParticles() {   /* :185-197 */
	const double init_wt = 1.0 / pf.n_particles;
	for (int k = 0; k < pf.n_particles; ++k) {
		particle_states[curr_set_id][k] = ssm->getState();
		particle_wts[k] = init_wt;
		particle_cum_wts[k] = k > 0 ? particle_wts[k] + particle_cum_wts[k - 1] : particle_wts[k];
		particle_ar[curr_set_id][k].fill(0.0);
	}
}
void PF::setRegion(const CornersT &corners) {   /* :616-620 */
	ssm->setCorners(corners);
	initializeParticles();
}
void PF::update() {   /* :207-447 */
	am->setFirstIter();
	if (pf.jacobian_as_sigma) jacobianSigma(false);
	iters_done = 0;
	int distr_id = 0;
	for (int iter_id = 0; iter_id < pf.max_iters; ++iter_id) {
		++iters_done;
		std::discrete_distribution<int> distr_id_dist;
		if (n_distr > 1) {   /* :240-258 */
			distr_id_dist = std::discrete_distribution<int>(distr_wts.begin(), distr_wts.end());
			for (int i = 0; i < n_distr; ++i) { distr_wts[i] = 0; distr_n_particles[i] = 0; }
		}
		double max_wt = std::numeric_limits<double>::lowest();
		for (int k = 0; k < pf.n_particles; ++k) {
			if (n_distr > 1) {   /* :261-268: the SSM's sampler is reset only when the distribution changes */
				const int new_distr_id = distr_id_dist(distr_id_gen);
				if (new_distr_id != distr_id) { distr_id = new_distr_id; ssm->setSampler(state_sigma[distr_id], state_mean[distr_id]); }
			}
			particle_distr[k] = distr_id;
			VectorXd &st = particle_states[curr_set_id][k], &ar = particle_ar[curr_set_id][k];
			if (pf.dynamic_model == PFParams::DynamicModel::AutoRegression1) {
				if (pf.update_type == PFParams::UpdateType::Additive) ssm->additiveAutoRegression1(perturbed_state, perturbed_ar, st, ar);
				else ssm->compositionalAutoRegression1(perturbed_state, perturbed_ar, st, ar);
				ar = perturbed_ar;
			} else if (pf.update_type == PFParams::UpdateType::Additive) ssm->additiveRandomWalk(perturbed_state, st);
			else ssm->compositionalRandomWalk(perturbed_state, st);
			st = perturbed_state;
			ssm->setState(st);
			am->updatePixVals(ssm->getPts());
			am->updateSimilarity(false);
			const double measuremnt_val = max_similarity - am->getSimilarity();
			double lik;
			switch (pf.likelihood_func) {
			case PFParams::LikelihoodFunc::AM: lik = am->getLikelihood(); break;
			case PFParams::LikelihoodFunc::Gaussian: lik = measurement_factor * std::exp(-0.5 * measuremnt_val / pf.measurement_sigma); break;
			default: lik = 1.0 / (1.0 + measuremnt_val); break;
			}
			particle_wts[k] = lik;
			particle_cum_wts[k] = k == 0 ? lik : lik + particle_cum_wts[k - 1];
			if (pf.update_distr_wts) { distr_wts[distr_id] += lik; distr_n_particles[distr_id] += 1; }   /* :345-348 */
			if (lik >= max_wt) { max_wt = lik; max_wt_id = k; }
		}
		if (pf.update_distr_wts) {   /* :354-369: average particle weight per distribution, normalised, floored */
			double wt_sum = 0;
			for (int i = 0; i < n_distr; ++i) if (distr_n_particles[i] > 0) { distr_wts[i] /= distr_n_particles[i]; wt_sum += distr_wts[i]; }
			for (int i = 0; i < n_distr; ++i) { distr_wts[i] /= wt_sum; if (distr_wts[i] < pf.min_distr_wt) distr_wts[i] = pf.min_distr_wt; }
		}
		bool perform_resampling = true;
		if (enable_adaptive_resampling) {   /* :381-390 */
			const double tot = particle_cum_wts[pf.n_particles - 1];
			double sq = 0;
			for (int k = 0; k < pf.n_particles; ++k) { const double v = particle_wts[k] / tot; sq += v * v; }
			const double n_eff_particles = sq == 0 ? 0 : 1.0 / sq;
			if (n_eff_particles > min_eff_particles) perform_resampling = false;
		}
		if (perform_resampling) switch (pf.resampling_type) {
		case PFParams::ResamplingType::None: break;
		case PFParams::ResamplingType::BinaryMultinomial: binaryMultinomialResampling(); break;
		case PFParams::ResamplingType::LinearMultinomial: linearMultinomialResampling(); break;
		default: throw utils::FunctonNotImplemented("PF :: residual resampling");
		}
		switch (pf.mean_type) {
		case PFParams::MeanType::None: ssm->setState(particle_states[curr_set_id][max_wt_id]); break;
		case PFParams::MeanType::SSM:
			ssm->estimateMeanOfSamples(mean_state, particle_states[curr_set_id], pf.n_particles);
			ssm->setState(mean_state);
			break;
		default: updateMeanCorners(); ssm->setCorners(mean_corners); break;
		}
		const double update_norm = utils::squaredDistance(prev_corners, ssm->getCorners());
		prev_corners = ssm->getCorners();
		if (update_norm < pf.epsilon) break;
		am->clearFirstIter();
	}
	if (pf.reset_to_mean) initializeParticles();
	if (pf.enable_learning) { am->updateModel(ssm->getPts()); max_similarity = am->getSimilarity(); }
}
void PF::binaryMultinomialResampling() {   /* :455-502 */
	const int n = pf.n_particles;
	const double tot = particle_cum_wts[n - 1];
	for (int k = 0; k < n; ++k) particle_cum_wts[k] /= tot;
	double max_wt = std::numeric_limits<double>::lowest();
	for (int k = 0; k < n; ++k) {
		const double u = resample_dist(resample_gen);
		int lower_id = 0, upper_id = n - 1, resample_id = (lower_id + upper_id) / 2;
		while (upper_id > lower_id) {
			if (particle_cum_wts[resample_id] >= u) upper_id = resample_id; else lower_id = resample_id + 1;
			resample_id = (lower_id + upper_id) / 2;
		}
		particle_states[1 - curr_set_id][k] = particle_states[curr_set_id][resample_id];
		particle_ar[1 - curr_set_id][k] = particle_ar[curr_set_id][resample_id];
		if (particle_wts[resample_id] >= max_wt) { max_wt = particle_wts[resample_id]; max_wt_id = k; }
	}
	curr_set_id = 1 - curr_set_id;
}
void PF::linearMultinomialResampling() {   /* :505-536 */
	const int n = pf.n_particles;
	const double tot = particle_cum_wts[n - 1];
	for (int k = 0; k < n; ++k) particle_cum_wts[k] /= tot;
	double max_wt = std::numeric_limits<double>::lowest();
	for (int k = 0; k < n; ++k) {
		const double u = resample_dist(resample_gen);
		int resample_id = 0;
		while (resample_id < n - 1 && particle_cum_wts[resample_id] < u) ++resample_id;
		particle_states[1 - curr_set_id][k] = particle_states[curr_set_id][resample_id];
		particle_ar[1 - curr_set_id][k] = particle_ar[curr_set_id][resample_id];
		if (particle_wts[resample_id] >= max_wt) { max_wt = particle_wts[resample_id]; max_wt_id = k; }
	}
	curr_set_id = 1 - curr_set_id;
}
void PF::updateMeanCorners() {   /* :607-614 */
	for (int q = 0; q < 8; ++q) mean_corners.data()[q] = 0;
	for (int k = 0; k < pf.n_particles; ++k) {
		ssm->setState(particle_states[curr_set_id][k]);
		const CornersT &c = ssm->getCorners();
		for (int q = 0; q < 8; ++q) mean_corners.data()[q] += (c.data()[q] - mean_corners.data()[q]) / (k + 1);
	}
}
} // namespace nt

} // namespace mtf
