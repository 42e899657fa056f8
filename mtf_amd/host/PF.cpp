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
	state_sigma.resize(S); state_mean.resize(S);
	/* NT/PF.cc:27-52: a single sigma / mean entry is broadcast; a wrong size is an error */
	if (pf.ssm_sigma.size() != 1 && (int)pf.ssm_sigma.size() < S)
		throw utils::InvalidArgument("PF :: SSM sigma has too few values " + std::to_string(pf.ssm_sigma.size()));
	for (int k = 0; k < S; ++k) {
		state_sigma[k] = pf.ssm_sigma.size() == 1 ? pf.ssm_sigma[0] : pf.ssm_sigma[k];
		state_mean[k] = pf.ssm_mean.empty() ? 0.0 : (pf.ssm_mean.size() == 1 ? pf.ssm_mean[0] : pf.ssm_mean[k]);
	}
	const double pi = 3.14159265358979323846;
	measurement_factor = 1.0 / std::sqrt(2 * pi * pf.measurement_sigma);   /* :69-70 */
	for (int set_id = 0; set_id < 2; ++set_id) {
		particle_states[set_id].assign(pf.n_particles, VectorXd(S));
		particle_ar[set_id].assign(pf.n_particles, VectorXd(S));
	}
	particle_wts.resize(pf.n_particles); particle_cum_wts.resize(pf.n_particles);
	perturbed_state.resize(S); perturbed_ar.resize(S); mean_state.resize(S);
	if (pf.seed) resample_gen.seed(pf.seed);
	else { std::random_device r; std::seed_seq seq{r(), r(), r(), r(), r(), r(), r(), r()}; resample_gen.seed(seq); }
}
void PF::initialize(const CornersT &corners) {   /* :136-183 */
	am->clearInitStatus(); ssm->clearInitStatus();
	ssm->initialize(corners, am->getNChannels());
	ssm->initializeSampler(state_sigma, state_mean);
	am->initializePixVals(ssm->getPts());
	am->initializeSimilarity();
	max_similarity = am->getSimilarity();
	initializeParticles();
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
void PF::update() {   /* :207-447 (one sampler distribution, no jacobian_as_sigma) */
	am->setFirstIter();
	iters_done = 0;
	for (int iter_id = 0; iter_id < pf.max_iters; ++iter_id) {
		++iters_done;
		double max_wt = std::numeric_limits<double>::lowest();
		for (int k = 0; k < pf.n_particles; ++k) {
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
			if (lik >= max_wt) { max_wt = lik; max_wt_id = k; }
		}
		switch (pf.resampling_type) {
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
		const double update_norm = prev_corners.squaredDistance(ssm->getCorners());
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
	for (double &v : mean_corners.v) v = 0;
	for (int k = 0; k < pf.n_particles; ++k) {
		ssm->setState(particle_states[curr_set_id][k]);
		const CornersT &c = ssm->getCorners();
		for (int q = 0; q < 8; ++q) mean_corners.v[q] += (c.v[q] - mean_corners.v[q]) / (k + 1);
	}
}
} // namespace nt

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
	for (int k = 0; k < S; ++k) {
		d.ssm_sigma[k] = pf.ssm_sigma.size() == 1 ? pf.ssm_sigma[0] : pf.ssm_sigma.at(k);
		d.ssm_mean[k] = pf.ssm_mean.empty() ? 0.0 : (pf.ssm_mean.size() == 1 ? pf.ssm_mean[0] : pf.ssm_mean.at(k));
	}
	HipPair::check(mtfhip_pf_create(a->pair()->b, &d, &h));
}
PF::~PF() { mtfhip_pf_destroy(h); }
void PF::setComm(mtfhip_comm *comm) { HipPair::check(mtfhip_pf_set_comm(h, comm)); }
void PF::initialize(const CornersT &corners) {   /* NT/PF.cc:136-183 */
	am->clearInitStatus(); ssm->clearInitStatus();
	ssm->initialize(corners, am->getNChannels());
	am->initializePixVals(ssm->getPts());
	am->initializeSimilarity();
	HipPair::check(mtfhip_pf_initialize(h));
}
void PF::setRegion(const CornersT &corners) {   /* NT/PF.cc:616-620 */
	ssm->setCorners(corners);                                    /* (keeps the adapter's host mirrors current) */
	HipPair::check(mtfhip_pf_set_region(h, corners.data()));     /* same corners again + initializeParticles */
	hssm->markMoved();
}
void PF::update() {
	am->setFirstIter();
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
