/*
 * api_pf.hip -- the particle filter behind the C ABI (nt::PF, SM/src/NT/PF.cc) and the collective of its sharded form
 * (C-ABI implementation, include/mtfhip.h; shared declarations: mtfhip_api_internal.h)
 *
 * One iteration of nt::PF::update's loop is five launches: sample generation (k_pf_propagate), scoring
 * (k_score_candidates[_fast]; on R ranks each scores its contiguous block and ONE all-gather over RCCL puts every weight on
 * every rank, PF.cc:262-277), weights -> cumulative weights (k_pf_weights), resampling (k_pf_select), estimate
 * (k_pf_estimate, which also delivers its 32 doubles to host-coherent memory and raises the flag the host waits on).  The
 * reference does all of it per particle on the host, including a 4-corner DLT (8 x 9 JacobiSVD) per sample.
 *
 * RCCL is bound at run time (dlopen): libmtfhip.so has no link-time dependency on it, a process that already carries an
 * RCCL (PyTorch-ROCm bundles one) shares it, and the single-GPU library works where RCCL is absent.
 */
#include "mtfhip_api_internal.h"

#include <dlfcn.h>

/* ------------------------------------------------------------------ RCCL, bound at run time */
namespace {
typedef struct { char internal[128]; } rccl_unique_id;   /* ncclUniqueId, rccl.h:40-43 */
typedef void *rccl_comm_t;
enum { RCCL_FLOAT64 = 8 };                                /* ncclFloat64, rccl.h:467 */
struct Rccl {
	void *handle = nullptr;
	int (*GetUniqueId)(rccl_unique_id *) = nullptr;
	int (*CommInitRank)(rccl_comm_t *, int, rccl_unique_id, int) = nullptr;
	int (*CommDestroy)(rccl_comm_t) = nullptr;
	int (*AllGather)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(int) = nullptr;
	bool ok = false;
};
Rccl &rccl() {
	static Rccl r;
	static bool tried = false;
	if (tried) return r;
	tried = true;
	const char *names[] = {"librccl.so.1", "librccl.so"};
	for (const char *n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   /* the copy the process already has */
	for (const char *n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
	if (!r.handle) return r;
	r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
	r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
	r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
	r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
	r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
	r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
	return r;
}
}  // namespace

struct mtfhip_comm {
	int rank = 0, world = 1, device = 0;
	rccl_comm_t comm = nullptr;   /* NULL when world == 1: the all-gather is a device copy */
};

struct mtfhip_pf {
	mtfhip_batch *b = nullptr;
	mtfhip_pf_desc d;
	mtfhip_comm *comm = nullptr;
	int n = 0, S = 0, cur = 0;
	unsigned iter = 0;
	double max_similarity = 0;
	bool initialized = false;
	double *d_states[2] = {nullptr, nullptr}, *d_ars[2] = {nullptr, nullptr};
	double *d_lik = nullptr, *d_sim = nullptr, *d_wts = nullptr, *d_cum = nullptr, *d_out = nullptr, *d_normals = nullptr, *d_uniforms = nullptr;
	double *d_parts = nullptr;   /* per-workgroup partial results of the selection pass */
	double *d_send = nullptr, *d_recv = nullptr;   /* sharded scoring: [2 m] send, [2 m world] receive (likelihood | similarity) */
	int *d_ids = nullptr;
	double prev_corners[8];
};

extern "C" {

/* ------------------------------------------------------------------ the collective */
int mtfhip_comm_unique_id(void *id128) {
	if (!id128) return fail(MTFHIP_ERR_INVALID_ARG, "comm_unique_id: NULL argument");
	Rccl &r = rccl();
	if (!r.ok) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "RCCL (librccl.so) could not be loaded: %s", dlerror() ? dlerror() : "symbols missing");
	rccl_unique_id id;
	const int rc = r.GetUniqueId(&id);
	if (rc != 0) return fail(MTFHIP_ERR_HIP, "ncclGetUniqueId failed: %s", r.GetErrorString ? r.GetErrorString(rc) : "?");
	std::memcpy(id128, id.internal, sizeof(id.internal));
	return MTFHIP_OK;
}
int mtfhip_comm_create(const void *id128, int rank, int world, int device, mtfhip_comm **out) {
	if (!out || world < 1 || rank < 0 || rank >= world) return fail(MTFHIP_ERR_INVALID_ARG, "comm_create: invalid rank %d / world %d", rank, world);
	mtfhip_comm *c = new mtfhip_comm;
	c->rank = rank; c->world = world; c->device = device;
	if (world > 1) {
		if (!id128) { delete c; return fail(MTFHIP_ERR_INVALID_ARG, "comm_create: the unique id of rank 0 is required for world > 1"); }
		Rccl &r = rccl();
		if (!r.ok) { delete c; return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "RCCL (librccl.so) could not be loaded"); }
		if (hipSetDevice(device) != hipSuccess) { delete c; return fail(MTFHIP_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device); }
		rccl_unique_id id;
		std::memcpy(id.internal, id128, sizeof(id.internal));
		const int rc = r.CommInitRank(&c->comm, world, id, rank);
		if (rc != 0) { delete c; return fail(MTFHIP_ERR_HIP, "ncclCommInitRank failed: %s", r.GetErrorString ? r.GetErrorString(rc) : "?"); }
	}
	*out = c;
	return MTFHIP_OK;
}
void mtfhip_comm_destroy(mtfhip_comm *c) {
	if (!c) return;
	if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
	delete c;
}
int mtfhip_comm_rank(const mtfhip_comm *c) { return c ? c->rank : 0; }
int mtfhip_comm_world(const mtfhip_comm *c) { return c ? c->world : 1; }
/* every rank contributes `count` doubles; every rank receives world x count, rank-major (PF.cc:262-277's weights vector once
 * the particles are sharded).  world == 1: a device-to-device copy. */
int mtfhip_allgather_scores(mtfhip_comm *c, const double *dev_send, int count, double *dev_recv, void *hip_stream) {
	if (!c || !dev_send || !dev_recv || count <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "allgather_scores: invalid argument");
	hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
	if (c->world == 1 || !c->comm) {
		HIP_TRY(hipMemcpyAsync(dev_recv, dev_send, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, st));
		return MTFHIP_OK;
	}
	const int rc = rccl().AllGather(dev_send, dev_recv, (size_t)count, RCCL_FLOAT64, c->comm, st);
	if (rc != 0) return fail(MTFHIP_ERR_HIP, "ncclAllGather failed: %s", rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ the particle filter */
static void pf_free(mtfhip_pf *pf) {
	void *ptrs[] = {pf->d_states[0], pf->d_states[1], pf->d_ars[0], pf->d_ars[1], pf->d_lik, pf->d_sim, pf->d_wts, pf->d_cum, pf->d_out,
		pf->d_normals, pf->d_uniforms, pf->d_send, pf->d_recv, pf->d_ids, pf->d_parts};
	for (void *p : ptrs) if (p) (void)hipFree(p);
}
int mtfhip_pf_create(mtfhip_batch *b, const mtfhip_pf_desc *d, mtfhip_pf **out) {
	if (!b || !d || !out) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: NULL argument");
	if (b->B != 1) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: the particle filter tracks one target (batch of %d)", b->B);
	if (d->n_particles < 1) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: n_particles must be positive");
	if (d->dynamic_model < 0 || d->dynamic_model > 1 || d->update_type < 0 || d->update_type > 1 || d->likelihood_func < 0 || d->likelihood_func > 2 ||
		d->mean_type < 0 || d->mean_type > 2) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: enum value out of range (PFParams.h:10-33)");
	if (d->resampling_type < 0 || d->resampling_type > 3) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: unknown resampling type %d", d->resampling_type);
	if (d->resampling_type == 3) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "pf_create: residual resampling (PF.cc:538-582) is not available on the device");
	if (b->desc.am == MTFHIP_AM_MI) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "pf_create: candidate scoring covers SSD and NCC");
	if (b->desc.ssm == MTFHIP_SSM_AFFINE && d->update_type == 1 && d->corner_based_sampling == 0)
		/* Affine::compositionalRandomWalk throws for geometric sampling (Affine.cc:540-552); point based sampling is the DLT of three points */
		return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "pf_create: Affine compositional sampling needs point based perturbations, which are not implemented");
	mtfhip_pf *pf = new mtfhip_pf;
	pf->b = b; pf->d = *d; pf->n = d->n_particles; pf->S = b->S;
	const size_t nS = (size_t)pf->n * pf->S, n = (size_t)pf->n;
	bool okm = true;
	auto A = [&](auto &p, size_t bytes) { if (hipMalloc(reinterpret_cast<void **>(&p), bytes) != hipSuccess) okm = false; };
	for (int k = 0; k < 2; ++k) { A(pf->d_states[k], sizeof(double) * nS); A(pf->d_ars[k], sizeof(double) * nS); }
	A(pf->d_lik, sizeof(double) * n); A(pf->d_sim, sizeof(double) * n); A(pf->d_wts, sizeof(double) * n); A(pf->d_cum, sizeof(double) * n);
	A(pf->d_out, sizeof(double) * 32); A(pf->d_parts, sizeof(double) * 18 * ((n + 255) / 256)); A(pf->d_normals, sizeof(double) * n * 10); A(pf->d_uniforms, sizeof(double) * n); A(pf->d_ids, sizeof(int) * n);
	if (!okm) { pf_free(pf); delete pf; return fail(MTFHIP_ERR_HIP, "pf_create: hipMalloc failed"); }
	*out = pf;
	return MTFHIP_OK;
}
void mtfhip_pf_destroy(mtfhip_pf *pf) {
	if (!pf) return;
	pf_free(pf);
	delete pf;
}
/* shard the scoring over the communicator: rank r scores particles [r n / R, (r + 1) n / R) and one all-gather distributes the
 * weights; sample generation and resampling are replicated (identical draws on every rank) */
int mtfhip_pf_set_comm(mtfhip_pf *pf, mtfhip_comm *c) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_comm: NULL filter");
	pf->comm = c;
	if (pf->d_send) { (void)hipFree(pf->d_send); pf->d_send = nullptr; }
	if (pf->d_recv) { (void)hipFree(pf->d_recv); pf->d_recv = nullptr; }
	if (c && c->world > 1) {
		const size_t m = (size_t)(pf->n + c->world - 1) / c->world;
		HIP_TRY(hipMalloc(&pf->d_send, sizeof(double) * 2 * m));
		HIP_TRY(hipMalloc(&pf->d_recv, sizeof(double) * 2 * m * c->world));
	}
	return MTFHIP_OK;
}
/* ProjectiveBase::setSampler (ProjectiveBase.cc:208-215) */
int mtfhip_pf_set_sampler(mtfhip_pf *pf, const double *sigma, const double *mean) {
	if (!pf || !sigma || !mean) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_sampler: NULL argument");
	for (int s = 0; s < pf->S; ++s) { pf->d.ssm_sigma[s] = sigma[s]; pf->d.ssm_mean[s] = mean[s]; }
	return MTFHIP_OK;
}
/* PF::initializeParticles (PF.cc:185-197) */
static int pf_initialize_particles(mtfhip_pf *pf) {
	mtfhip_batch *b = pf->b;
	hipStream_t st = b->ctx->stream;
	(void)b->view();   /* a stale single-target warp is uploaded first: the fill reads the device copy of the state */
	launch_pf_fill(pf->n, pf->S, b->d_states, pf->d_states[pf->cur], pf->d_ars[pf->cur], st);
	return MTFHIP_OK;
}
/* the part of nt::PF::initialize that follows ssm->initialize, am->initializePixVals and am->initializeSimilarity
 * (PF.cc:136-183): max_similarity, initializeParticles, prev_corners */
int mtfhip_pf_initialize(mtfhip_pf *pf) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_initialize: NULL filter");
	mtfhip_batch *b = pf->b;
	FLUSH(b);
	if (!b->have_corners || !b->init_pix_vals || !b->init_sim) return fail(MTFHIP_ERR_LOGIC, "pf_initialize before ssm->initialize / am->initializePixVals / am->initializeSimilarity");
	double f = 0;
	TRY(mtfhip_am_get_similarity(b, &f));
	pf->max_similarity = f;
	pf->cur = 0; pf->iter = 0;
	TRY(pf_initialize_particles(pf));
	std::memcpy(pf->prev_corners, b->th[0].corners, sizeof(pf->prev_corners));
	pf->initialized = true;
	return MTFHIP_OK;
}
/* PF::setRegion (PF.cc:616-620) */
int mtfhip_pf_set_region(mtfhip_pf *pf, const double *corners) {
	if (!pf || !corners) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_region: NULL argument");
	TRY(mtfhip_ssm_set_corners(pf->b, corners));
	TRY(pf_initialize_particles(pf));
	std::memcpy(pf->prev_corners, pf->b->th[0].corners, sizeof(pf->prev_corners));
	return MTFHIP_OK;
}

/* One iteration of the loop of nt::PF::update (PF.cc:260-447).  normals: n x nz standard normals (nz = 10 with corner based
 * homography sampling, else the state size), uniforms: n draws in (0, 1]; host arrays, or NULL: the device generator
 * (Philox4x32-10 keyed by desc.seed, the iteration count and the particle).  update_norm: squared corner change of the
 * estimate (PF.cc:438-439). */
int mtfhip_pf_iteration(mtfhip_pf *pf, const double *normals, const double *uniforms, double *update_norm) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_iteration: NULL filter");
	if (!pf->initialized) return fail(MTFHIP_ERR_LOGIC, "pf_iteration before pf_initialize");
	mtfhip_batch *b = pf->b;
	FLUSH_AM(b);   /* (the candidates carry their own warps: the batch's CURR_PTS are not read) */
	TRY(need_image(b));
	hipStream_t st = b->ctx->stream;
	const int n = pf->n, S = pf->S;
	const bool hom = b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY;
	PfLaunch p;
	p.n = n; p.S = S; p.dynamic_model = pf->d.dynamic_model; p.update_type = pf->d.update_type;
	p.corner_based = (hom && pf->d.corner_based_sampling) ? 1 : 0;
	p.likelihood_func = pf->d.likelihood_func; p.resampling_type = pf->d.resampling_type; p.mean_type = pf->d.mean_type;
	p.ar_coeff = pf->d.ar_coeff; p.measurement_sigma = pf->d.measurement_sigma; p.max_similarity = pf->max_similarity;
	for (int k = 0; k < 8; ++k) { p.sigma[k] = pf->d.ssm_sigma[k]; p.mean[k] = pf->d.ssm_mean[k]; p.init_corners[k] = b->th[0].init_corners[k]; }
	for (int k = 0; k < 12; ++k) p.init_corners_hm[k] = b->th[0].init_corners_hm[k];
	{
		/* template corners -> unit square: the inverse of the closed-form square-to-quadrilateral map (rect_to_quad) */
		M3 sq;
		if (!rect_to_quad(0.0, 0.0, 1.0, 1.0, b->th[0].init_corners, sq)) return fail(MTFHIP_ERR_INVALID_ARG, "pf_iteration: degenerate template corners");
		const M3 inv = m3_inverse(sq);
		std::memcpy(p.sq_inv, inv.m, sizeof(p.sq_inv));
	}
	p.seed = pf->d.seed; p.iter = pf->iter;
	p.normals = nullptr; p.uniforms = nullptr;
	const int nz = p.corner_based ? 10 : S;
	if (normals) {
		HIP_TRY(hipMemcpyAsync(pf->d_normals, normals, sizeof(double) * (size_t)n * nz, hipMemcpyHostToDevice, st));
		p.normals = pf->d_normals;
	}
	if (uniforms) {
		HIP_TRY(hipMemcpyAsync(pf->d_uniforms, uniforms, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
		p.uniforms = pf->d_uniforms;
	}
	double *stc = pf->d_states[pf->cur], *arc = pf->d_ars[pf->cur];
	unsigned long long pub_seq = 0;
	{
		TimedScope ts(b->ctx, "pf_propagate");
		launch_pf_propagate(b->desc.ssm, p, stc, arc, st);
	}
	/* scoring: setState -> updatePixVals -> updateSimilarity -> likelihood per particle (PF.cc:341-365) */
	const mtfhip_comm *c = pf->comm;
	if (c && c->world > 1) {
		const int m = (n + c->world - 1) / c->world;
		const int lo = std::min(n, c->rank * m), hi = std::min(n, lo + m);
		if (hi > lo) TRY(mtfhip_score_candidates_dev(b, stc + (size_t)lo * S, hi - lo, pf->d_send, pf->d_send + m));
		TRY(mtfhip_allgather_scores(pf->comm, pf->d_send, 2 * m, pf->d_recv, st));
		/* rank-major [likelihood m | similarity m] blocks -> the two flat vectors */
		for (int r = 0; r < c->world; ++r) {
			const int rlo = std::min(n, r * m), cnt = std::min(n, rlo + m) - rlo;
			if (cnt <= 0) continue;
			HIP_TRY(hipMemcpyAsync(pf->d_lik + rlo, pf->d_recv + (size_t)2 * m * r, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
			HIP_TRY(hipMemcpyAsync(pf->d_sim + rlo, pf->d_recv + (size_t)2 * m * r + m, sizeof(double) * cnt, hipMemcpyDeviceToDevice, st));
		}
	} else {
		TRY(mtfhip_score_candidates_dev(b, stc, n, pf->d_lik, pf->d_sim));
	}
	{
		TimedScope ts(b->ctx, "pf_resample");
		if (b->h_acc_dev) pub_seq = ++b->acc_seq;
		launch_pf_resample(b->desc.ssm, p, pf->d_lik, pf->d_sim, pf->d_wts, pf->d_cum, stc, arc, pf->d_states[1 - pf->cur], pf->d_ars[1 - pf->cur],
			pf->d_ids, pf->d_out, pf->d_parts, pub_seq ? b->h_acc_dev : nullptr, b->h_flag_dev, pub_seq, st);
	}
	if (p.resampling_type == 1 || p.resampling_type == 2) pf->cur = 1 - pf->cur;   /* curr_set_id = 1 - curr_set_id (PF.cc:501) */
	/* the estimate (32 doubles) comes back through host-coherent pinned memory + the flag the host spins on, like every other
	 * per-iteration result of the library (a copy into pageable memory + stream synchronisation was 15 us of a 137 us iteration) */
	double out[32];
	if (pub_seq) {   /* k_pf_estimate delivered it itself */
		TRY(wait_host_flag(b, pub_seq));
		std::memcpy(out, b->h_acc, sizeof(out));
	} else {
		HIP_TRY(hipMemcpyAsync(out, pf->d_out, sizeof(out), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
	}
	++pf->iter;
	/* the estimate becomes the SSM's state (PF.cc:421-437) */
	if (p.mean_type == 2) TRY(mtfhip_ssm_set_corners(b, out + 10));
	else TRY(mtfhip_ssm_set_state(b, out));
	double un = 0;
	for (int q = 0; q < 8; ++q) { const double d = pf->prev_corners[q] - b->th[0].corners[q]; un += d * d; }
	std::memcpy(pf->prev_corners, b->th[0].corners, sizeof(pf->prev_corners));
	if (update_norm) *update_norm = un;
	return MTFHIP_OK;
}
/* nt::PF::update (PF.cc:207-447) with the device generator: up to max_iters iterations, stop when the estimate's corners move
 * by less than epsilon; reset_to_mean re-initialises the particles at the estimate */
int mtfhip_pf_update(mtfhip_pf *pf, int *n_iters) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_update: NULL filter");
	int it = 0;
	for (; it < pf->d.max_iters; ++it) {
		double un = 0;
		TRY(mtfhip_pf_iteration(pf, nullptr, nullptr, &un));
		if (un < pf->d.epsilon) { ++it; break; }
	}
	if (pf->d.reset_to_mean) TRY(pf_initialize_particles(pf));
	if (n_iters) *n_iters = it;
	return MTFHIP_OK;
}
int mtfhip_pf_get_particles(mtfhip_pf *pf, double *states, double *ars, double *wts, int *resample_ids) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_get_particles: NULL filter");
	hipStream_t st = pf->b->ctx->stream;
	const size_t nS = (size_t)pf->n * pf->S;
	if (states) HIP_TRY(hipMemcpyAsync(states, pf->d_states[pf->cur], sizeof(double) * nS, hipMemcpyDeviceToHost, st));
	if (ars) HIP_TRY(hipMemcpyAsync(ars, pf->d_ars[pf->cur], sizeof(double) * nS, hipMemcpyDeviceToHost, st));
	if (wts) HIP_TRY(hipMemcpyAsync(wts, pf->d_wts, sizeof(double) * pf->n, hipMemcpyDeviceToHost, st));
	if (resample_ids) HIP_TRY(hipMemcpyAsync(resample_ids, pf->d_ids, sizeof(int) * pf->n, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	return MTFHIP_OK;
}
int mtfhip_pf_set_particles(mtfhip_pf *pf, const double *states, const double *ars) {
	if (!pf || !states) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_particles: NULL argument");
	hipStream_t st = pf->b->ctx->stream;
	const size_t nS = (size_t)pf->n * pf->S;
	HIP_TRY(hipMemcpyAsync(pf->d_states[pf->cur], states, sizeof(double) * nS, hipMemcpyHostToDevice, st));
	if (ars) HIP_TRY(hipMemcpyAsync(pf->d_ars[pf->cur], ars, sizeof(double) * nS, hipMemcpyHostToDevice, st));
	else HIP_TRY(hipMemsetAsync(pf->d_ars[pf->cur], 0, sizeof(double) * nS, st));
	HIP_TRY(hipStreamSynchronize(st));
	return MTFHIP_OK;
}
double mtfhip_pf_max_similarity(const mtfhip_pf *pf) { return pf ? pf->max_similarity : 0.0; }

} /* extern "C" */
