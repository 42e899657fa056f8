/*
 * api_pf.hip -- the particle filter behind the C ABI (nt::PF, SM/src/NT/PF.cc) and the collective of its sharded form
 * (C-ABI implementation, include/mtfhip.h; shared declarations: mtfhip_api_internal.h)
 *
 * One iteration of nt::PF::update's loop is three launches (kernels_pf.hip): scoring + weight (k_pf_score; on R ranks each
 * scores its contiguous block and ONE in-place all-gather over RCCL leaves the flat weight vector on every rank,
 * PF.cc:262-277), chunk-local cumulative weights (k_pf_scan), resampling + estimate + the proposals of the next iteration
 * (k_pf_select, whose last workgroup also delivers the 32 doubles of the estimate to host-coherent memory and raises the flag
 * the host waits on); a fourth launch (k_pf_propose) only where the proposals could not be made ahead.  The reference does all
 * of it per particle on the host, including a 4-corner DLT (8 x 9 JacobiSVD) per sample.
 *
 * RCCL is bound at run time (dlopen): libmtfhip.so has no link-time dependency on it, a process that already carries an
 * RCCL (PyTorch-ROCm bundles one) shares it, and the single-GPU library works where RCCL is absent.
 */
#include "mtfhip_api_internal.h"

#include <dlfcn.h>
#include <hipcub/hipcub.hpp>   /* residual resampling: one radix sort + one exclusive scan per iteration (not the hot configuration) */
#include <condition_variable>
#include <mutex>

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
	std::string why = "symbols missing";   /* dlerror() is read once: a second call returns NULL */
};
Rccl &rccl() {
	static Rccl r;
	static bool tried = false;
	if (tried) return r;
	tried = true;
	const char *names[] = {"librccl.so.1", "librccl.so"};
	for (const char *n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   /* the copy the process already has */
	for (const char *n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
	if (!r.handle) { const char *e = dlerror(); r.why = e ? e : "librccl.so not found"; return r; }
	r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
	r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
	r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
	r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
	r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
	r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
	return r;
}
}  // namespace

/* A communicator is either an RCCL one (one process per GPU, the production form), a single rank (no exchange), or a member of a
 * LOOPBACK group: `world` ranks that are threads of one process sharing one GPU.  The loopback group exists so that the sharded
 * code path -- block bounds, ragged last blocks, the in-place all-gather layout, the re-evaluated proposals -- can be executed
 * and compared with the unsharded filter on a single GPU; its all-gather is a rendezvous of the member threads plus
 * device-to-device copies.  Everything outside mtfhip_allgather_scores is the same for the three kinds. */
struct LoopGroup {
	int world = 1, refs = 0;
	std::mutex mu;
	std::condition_variable cv;
	int arrived = 0;
	unsigned long gen = 0;
	std::vector<const double *> send;
	std::vector<void *> mailbox;   /* peer-store exchange: every member's mailbox (the same address space: no IPC mapping) */
	void barrier() {
		std::unique_lock<std::mutex> lk(mu);
		const unsigned long g = gen;
		if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); }
		else cv.wait(lk, [&] { return gen != g; });
	}
};
struct mtfhip_comm {
	int rank = 0, world = 1, device = 0;
	rccl_comm_t comm = nullptr;   /* NULL when world == 1, loopback or detached */
	LoopGroup *loop = nullptr;
	bool detached = false;        /* rank and world only: no collective (the sharded filter then exchanges through peer stores) */
};

struct mtfhip_pf {
	mtfhip_batch *b = nullptr;
	mtfhip_pf_desc d;
	mtfhip_comm *comm = nullptr;
	int n = 0, S = 0;
	int sampler = PF_SAMPLER_STATE, nz = 8;
	unsigned iter = 0;
	double max_similarity = 0;
	bool initialized = false;
	/* look-ahead: the selection pass of iteration t leaves the proposals of iteration t + 1 in d_prop[1 - pc] (device generator
	 * only); valid while nothing they depend on has changed -- the particle set, the sampler's distributions, the template
	 * corners.  MTFHIP_PF_LOOKAHEAD=0: every iteration proposes in a launch of its own. */
	bool lookahead_enabled = true, prop_valid = false;
	bool skip_unread_estimates = true;   /* MTFHIP_PF_SKIP_ESTIMATE=0 at creation: every iteration of a chained update() computes its estimate */
	bool local_enabled = true, pert_ahead_enabled = true;   /* MTFHIP_PF_LOCAL=0 / MTFHIP_PF_PERT_AHEAD=0 at creation: the scan launch / the draws inside the selection pass */
	unsigned prop_iter = 0;
	long prop_corners_epoch = -1;
	int pc = 0;
	size_t wts_capacity = 0;
	double *d_st = nullptr, *d_ar = nullptr;                                   /* the current (resampled) set */
	double *d_prop[2] = {nullptr, nullptr}, *d_prop_ar[2] = {nullptr, nullptr}; /* proposals: this iteration's | the next one's */
	double *d_wts = nullptr, *d_cum = nullptr, *d_chunk = nullptr, *d_out = nullptr, *d_normals = nullptr, *d_uniforms = nullptr;
	double *d_parts = nullptr, *d_gparts = nullptr;   /* per-workgroup rows of the selection pass and their per-group folds */
	/* perturbations drawn ahead (PfSelectPlan): buffer q holds those of iteration pert_iter[q] (-1: none) for the template corners of
	 * pert_epoch[q] and the sampler of pert_gen[q]; the selection pass of iteration t reads buffer (t + 1) & 1 and fills t & 1 with t + 2's */
	double *d_pert[2] = {nullptr, nullptr};
	long pert_iter[2] = {-1, -1}, pert_epoch[2] = {-1, -1};
	unsigned pert_gen[2] = {0, 0}, sampler_gen = 0;
	int *d_ids = nullptr, *d_counters = nullptr;
	/* residual resampling (PF.cc:538-582): sort keys in / out, particle order in / out, copies, their starts, hipCUB's scratch */
	double *d_res_keys = nullptr;
	int *d_res_idx = nullptr;
	void *d_res_tmp = nullptr; size_t res_tmp_bytes = 0;
	double prev_corners[8];
	/* several sampler distributions + adaptive resampling (PF.cc:240-269, 345-390): d_distr = sigma [8][8] | mean [8][8] | running sums
	 * of the distribution weights [8] | the weights [8]; the particles' distribution ids; the scan's per-chunk statistics; its verdict
	 * "this iteration resamples"; the next iteration's distribution draws when the caller hands them in */
	int n_distr = 1;
	double *d_distr = nullptr, *d_scan_stats = nullptr, *d_distr_u = nullptr;
	int *d_distr_ids = nullptr, *d_resample_flag = nullptr;
	std::vector<double> distr_u_next;
	/* the peer-store exchange (mtfhip_pf_set_exchange; PfPeerPush / PfPeerWait in mtfhip_internal.h).  mailbox: header (counters[kPfMaxPeers] |
	 * the storing launch's own arrival counter | seed | n | cap) | wts[2][cap] doubles; base[q]: rank q's mailbox as this process addresses it (own rank: mailbox; RCCL ranks: an IPC mapping;
	 * loopback ranks: the pointer itself) */
	struct Peer {
		bool on = false;
		void *mailbox = nullptr;
		size_t cap = 0;
		void *base[kPfMaxPeers] = {};
		bool mapped[kPfMaxPeers] = {};
		unsigned long long epoch = 0, expected[kPfMaxPeers] = {};
		int *h_err = nullptr, *h_err_dev = nullptr;
		/* a FIXED-SIZE header in front of the two weight vectors -- counters[kPfMaxPeers] | the storing launch's own arrival counter | the
		 * filter's seed | its particle count | its vector capacity -- so that a rank can read what a peer was created with BEFORE it
		 * addresses anything whose position depends on it (r04 advisor: ranks created with different n_particles / capacities stored
		 * outside the peer's allocation) */
		static constexpr int kHeadWords = (kPfMaxPeers + 4 + 15) & ~15;   /* whole 128-byte lines: the vectors stay line aligned */
		double *wts(int q, int parity) const { return static_cast<double *>(base[q]) + kHeadWords + (size_t)parity * cap; }
		unsigned long long *counters(int q) const { return static_cast<unsigned long long *>(base[q]); }
		unsigned long long *seed_slot(int q) const { return counters(q) + kPfMaxPeers + 1; }   /* seed | n_particles | cap */
	} peer;
	/* the weights of the last iteration: the mailbox vector of the last exchange, or d_wts */
	double *last_wts() const { return peer.on && peer.epoch ? peer.wts(comm->rank, (int)(peer.epoch & 1)) : d_wts; }
};

/* block of rank `rank`: [lo, lo + cnt) with m = ceil(n / world) particles per rank (the last blocks may be short or empty), so
 * that the all-gather of m weights per rank IS the flat weight vector */
static void pf_shard(int n, int world, int rank, int *lo, int *cnt, int *m) {
	const int mm = (n + world - 1) / world;
	const int l = std::min(n, rank * mm), h = std::min(n, l + mm);
	*lo = l; *cnt = h - l; *m = mm;
}

extern "C" {

int mtfhip_pf_shard_bounds(int n_particles, int world, int rank, int *lo, int *count, int *per_rank) {
	if (n_particles < 0 || world < 1 || rank < 0 || rank >= world || !lo || !count || !per_rank)
		return fail(MTFHIP_ERR_INVALID_ARG, "pf_shard_bounds: invalid argument");
	pf_shard(n_particles, world, rank, lo, count, per_rank);
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ the collective */
int mtfhip_comm_unique_id(void *id128) {
	if (!id128) return fail(MTFHIP_ERR_INVALID_ARG, "comm_unique_id: NULL argument");
	Rccl &r = rccl();
	if (!r.ok) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "RCCL (librccl.so) could not be loaded: %s", r.why.c_str());
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
		if (!r.ok) { delete c; return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "RCCL (librccl.so) could not be loaded: %s", r.why.c_str()); }
		if (hipSetDevice(device) != hipSuccess) { delete c; return fail(MTFHIP_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device); }
		rccl_unique_id id;
		std::memcpy(id.internal, id128, sizeof(id.internal));
		const int rc = r.CommInitRank(&c->comm, world, id, rank);
		if (rc != 0) { delete c; return fail(MTFHIP_ERR_HIP, "ncclCommInitRank failed: %s", r.GetErrorString ? r.GetErrorString(rc) : "?"); }
	}
	*out = c;
	return MTFHIP_OK;
}
/* rank `rank` of `world` with nothing behind it: the ranks' few set-up bytes (the mailbox handles of mtfhip_pf_exchange_export) are
 * moved by the host program, and the weights by the peer-store exchange.  For ranks that RCCL cannot or need not join -- and what
 * lets two processes share one GPU in tests/test_gpu_trackers.py (RCCL refuses duplicate devices). */
int mtfhip_comm_create_detached(int rank, int world, int device, mtfhip_comm **out) {
	if (!out || world < 1 || rank < 0 || rank >= world) return fail(MTFHIP_ERR_INVALID_ARG, "comm_create_detached: invalid rank %d / world %d", rank, world);
	mtfhip_comm *c = new mtfhip_comm;
	c->rank = rank; c->world = world; c->device = device; c->detached = world > 1;
	*out = c;
	return MTFHIP_OK;
}
/* `world` loopback ranks on one device: out[r] is rank r's communicator; each is used by its own host thread */
int mtfhip_comm_create_loopback(int world, int device, mtfhip_comm **out) {
	if (!out || world < 1) return fail(MTFHIP_ERR_INVALID_ARG, "comm_create_loopback: invalid world %d", world);
	LoopGroup *g = new LoopGroup;
	g->world = world; g->refs = world; g->send.assign((size_t)world, nullptr); g->mailbox.assign((size_t)world, nullptr);
	for (int r = 0; r < world; ++r) {
		mtfhip_comm *c = new mtfhip_comm;
		c->rank = r; c->world = world; c->device = device; c->loop = g;
		out[r] = c;
	}
	return MTFHIP_OK;
}
void mtfhip_comm_destroy(mtfhip_comm *c) {
	if (!c) return;
	if (c->comm && rccl().ok) (void)rccl().CommDestroy(c->comm);
	if (c->loop) {
		bool last;
		{ std::lock_guard<std::mutex> lk(c->loop->mu); last = --c->loop->refs == 0; }
		if (last) delete c->loop;
	}
	delete c;
}
int mtfhip_comm_rank(const mtfhip_comm *c) { return c ? c->rank : 0; }
int mtfhip_comm_world(const mtfhip_comm *c) { return c ? c->world : 1; }
/* every rank contributes `count` doubles; every rank receives world x count, rank-major (PF.cc:262-277's weights vector once
 * the particles are sharded).  In place when dev_send == dev_recv + rank * count (ncclAllGather's in-place form).
 * world == 1: a device-to-device copy (nothing when in place). */
int mtfhip_allgather_scores(mtfhip_comm *c, const double *dev_send, int count, double *dev_recv, void *hip_stream) {
	if (!c || !dev_send || !dev_recv || count <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "allgather_scores: invalid argument");
	hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
	const bool in_place = dev_send == dev_recv + (size_t)c->rank * count;
	if (c->loop) {
		LoopGroup *g = c->loop;
		HIP_TRY(hipStreamSynchronize(st));            /* this rank's block is complete */
		g->send[(size_t)c->rank] = dev_send;
		g->barrier();                                 /* ... and so is everybody's */
		for (int q = 0; q < c->world; ++q) {
			if (q == c->rank && in_place) continue;
			HIP_TRY(hipMemcpyAsync(dev_recv + (size_t)q * count, g->send[(size_t)q], sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, st));
		}
		HIP_TRY(hipStreamSynchronize(st));
		g->barrier();                                 /* nobody's send block is overwritten before everybody has read it */
		return MTFHIP_OK;
	}
	if (c->detached) return fail(MTFHIP_ERR_LOGIC, "allgather_scores: a detached communicator has no collective");
	if (c->world == 1 || !c->comm) {
		if (!in_place) HIP_TRY(hipMemcpyAsync(dev_recv, dev_send, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, st));
		return MTFHIP_OK;
	}
	const int rc = rccl().AllGather(dev_send, dev_recv, (size_t)count, RCCL_FLOAT64, c->comm, st);
	if (rc != 0) return fail(MTFHIP_ERR_HIP, "ncclAllGather failed: %s", rccl().GetErrorString ? rccl().GetErrorString(rc) : "?");
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ the particle filter */
static void pf_free(mtfhip_pf *pf) {
	void *ptrs[] = {pf->d_st, pf->d_ar, pf->d_prop[0], pf->d_prop[1], pf->d_prop_ar[0], pf->d_prop_ar[1], pf->d_wts, pf->d_cum, pf->d_chunk, pf->d_out,
		pf->d_normals, pf->d_uniforms, pf->d_ids, pf->d_parts, pf->d_gparts, pf->d_counters, pf->d_res_keys, pf->d_res_idx, pf->d_res_tmp,
		pf->d_distr, pf->d_scan_stats, pf->d_distr_u, pf->d_distr_ids, pf->d_resample_flag, pf->d_pert[0], pf->d_pert[1]};
	for (void *p : ptrs) if (p) (void)hipFree(p);
	for (int q = 0; q < kPfMaxPeers; ++q) if (pf->peer.mapped[q]) (void)hipIpcCloseMemHandle(pf->peer.base[q]);
	if (pf->peer.mailbox) (void)hipFree(pf->peer.mailbox);
	if (pf->peer.h_err) (void)hipHostFree(pf->peer.h_err);
}
/* which sampler the (SSM, update type, dynamic model, sampling switches) combination selects -- and which combinations the
 * reference itself refuses */
static int pf_pick_sampler(const mtfhip_batch *b, const mtfhip_pf_desc *d, int *sampler, int *nz) {
	if (b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY) {
		*sampler = d->corner_based_sampling ? PF_SAMPLER_HOM_CORNERS : PF_SAMPLER_STATE;
		*nz = d->corner_based_sampling ? 10 : 8;
		return MTFHIP_OK;
	}
	const int pt = d->pt_based_sampling;
	if (pt < 0 || pt > 2) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: pt_based_sampling must be 0, 1 or 2 (AffineParams, Affine.h)");
	if (d->update_type == 0) {
		if (pt) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "Affine::additive%s :: point based sampling is not implemented yet", d->dynamic_model ? "AutoRegression1" : "RandomWalk");   /* Affine.cc:509-511, 525-527 */
		/* Affine.cc:512-519, 528-538: base and AR state go through stateToGeom (Affine.cc:411-462), whose branches depend on the sign
		 * and ordering conventions of Eigen's 2 x 2 JacobiSVD -- Eigen is not in this image, so the result could not be pinned */
		return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "pf_create: Affine additive sampling perturbs the geometric parametrisation through Affine::stateToGeom "
			"(a JacobiSVD whose conventions cannot be reproduced without Eigen): use update_type Compositional");
	}
	if (d->dynamic_model == 0 && pt == 0)
		return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "Affine::compositionalRandomWalk :: geometric sampling is not implemented yet");   /* Affine.cc:550-552 */
	*sampler = pt == 1 ? PF_SAMPLER_AFF_PTS1 : pt == 2 ? PF_SAMPLER_AFF_PTS2 : PF_SAMPLER_AFF_GEOM;
	*nz = pt == 2 ? 8 : 6;
	return MTFHIP_OK;
}
static size_t pf_round_chunk(size_t n) { const size_t c = (size_t)pf_chunk(); return (n + c - 1) / c * c; }
int mtfhip_pf_create(mtfhip_batch *b, const mtfhip_pf_desc *d, mtfhip_pf **out) {
	if (!b || !d || !out) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: NULL argument");
	if (b->B != 1) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: the particle filter tracks one target (batch of %d)", b->B);
	if (d->n_particles < 1) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: n_particles must be positive");
	if (d->dynamic_model < 0 || d->dynamic_model > 1 || d->update_type < 0 || d->update_type > 1 || d->likelihood_func < 0 || d->likelihood_func > 2 ||
		d->mean_type < 0 || d->mean_type > 2) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: enum value out of range (PFParams.h:10-33)");
	if (d->resampling_type < 0 || d->resampling_type > 3) return fail(MTFHIP_ERR_INVALID_ARG, "pf_create: unknown resampling type %d", d->resampling_type);
	if (b->desc.am == MTFHIP_AM_MI && !(b->desc.mi_n_bins == 8 || (b->desc.mi_n_bins <= 10 && b->C == 1)))
		return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "pf_create: MI particles are scored with up to 10 bins (multi-channel: 8)");
	int sampler = 0, nz = 0;
	TRY(pf_pick_sampler(b, d, &sampler, &nz));
	mtfhip_pf *pf = new mtfhip_pf;
	pf->b = b; pf->d = *d; pf->n = d->n_particles; pf->S = b->S; pf->sampler = sampler; pf->nz = nz;
	{ const char *e = std::getenv("MTFHIP_PF_LOOKAHEAD"); pf->lookahead_enabled = !(e && e[0] == '0'); }
	{ const char *e = std::getenv("MTFHIP_PF_LOCAL"); pf->local_enabled = !(e && e[0] == '0'); }
	{ const char *e = std::getenv("MTFHIP_PF_SKIP_ESTIMATE"); pf->skip_unread_estimates = !(e && e[0] == '0'); }
	{ const char *e = std::getenv("MTFHIP_PF_PERT_AHEAD"); pf->pert_ahead_enabled = !(e && e[0] == '0'); }
	const size_t nS = (size_t)pf->n * pf->S, n = (size_t)pf->n, npad = pf_round_chunk(n), nch = npad / (size_t)pf_chunk();
	bool okm = true;
	auto A = [&](auto &p, size_t bytes) { if (hipMalloc(reinterpret_cast<void **>(&p), bytes) != hipSuccess) okm = false; };
	A(pf->d_st, sizeof(double) * nS); A(pf->d_ar, sizeof(double) * nS);
	for (int k = 0; k < 2; ++k) { A(pf->d_prop[k], sizeof(double) * nS); A(pf->d_prop_ar[k], sizeof(double) * nS); }
	for (int k = 0; k < 2; ++k) A(pf->d_pert[k], sizeof(double) * 8 * n);
	A(pf->d_wts, sizeof(double) * npad); pf->wts_capacity = npad;
	A(pf->d_cum, sizeof(double) * npad); A(pf->d_chunk, sizeof(double) * (2 + 16) * nch);   /* chunk totals | their prefix | sub-block sums */
	const size_t nblk = (n + 255) / 256, ngrp = (nblk + 63) / 64;
	A(pf->d_out, sizeof(double) * 32); A(pf->d_parts, sizeof(double) * pf_parts_per_block() * nblk); A(pf->d_gparts, sizeof(double) * pf_parts_per_block() * ngrp);
	A(pf->d_normals, sizeof(double) * n * 10); A(pf->d_uniforms, sizeof(double) * n); A(pf->d_ids, sizeof(int) * n); A(pf->d_counters, sizeof(int) * (2 + ngrp));
	if (okm && hipMemsetAsync(pf->d_counters, 0, sizeof(int) * (2 + ngrp), b->ctx->stream) != hipSuccess) okm = false;
	if (d->resampling_type == 3) {
		A(pf->d_res_keys, sizeof(double) * 2 * n); A(pf->d_res_idx, sizeof(int) * 4 * n);   /* keys in | out ; idx in | order | copies | starts */
		size_t t1 = 0, t2 = 0;
		(void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, t1, (const double *)nullptr, (double *)nullptr, (const int *)nullptr, (int *)nullptr, (int)n);
		(void)hipcub::DeviceScan::ExclusiveSum(nullptr, t2, (const int *)nullptr, (int *)nullptr, (int)n);
		pf->res_tmp_bytes = std::max(t1, t2) + 256;
		A(pf->d_res_tmp, pf->res_tmp_bytes);
	}
	if (okm && hipMemsetAsync(pf->d_out, 0, sizeof(double) * 32, b->ctx->stream) != hipSuccess) okm = false;
	if (d->adaptive_resampling_thresh > 0 && d->adaptive_resampling_thresh <= 1) {   /* PF.cc:114-118 */
		A(pf->d_scan_stats, sizeof(double) * 17 * nch); A(pf->d_resample_flag, sizeof(int));
	}
	if (!okm) { pf_free(pf); delete pf; return fail(MTFHIP_ERR_HIP, "pf_create: hipMalloc failed"); }
	*out = pf;
	return MTFHIP_OK;
}
/* PFParams::processDistributions + nt::PF's state_sigma / state_mean (PFParams.cc:101-170, PF.cc:55-56, 96-97): n_distr sampler
 * distributions, rows of 8; every particle draws its distribution from weights that follow the average particle weight each
 * distribution produced in the previous iteration (update_distr_wts, floored at min_distr_wt: PF.cc:345-369).  n_distr == 1 is
 * mtfhip_pf_set_sampler.  With several distributions and update_distr_wts == 0 the reference zeroes the weights and then builds a
 * discrete distribution from zeros (PF.cc:254-257, 241): refused. */
int mtfhip_pf_set_distributions(mtfhip_pf *pf, int n_distr, const double *sigma, const double *mean) {
	if (!pf || !sigma || !mean) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_distributions: NULL argument");
	if (n_distr < 1 || n_distr > 8) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_distributions: %d distributions (1 .. 8)", n_distr);
	if (n_distr > 1 && !pf->d.update_distr_wts)
		return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "pf_set_distributions: several distributions need update_distr_wts (the reference divides by a zero weight sum without it, PF.cc:241-257)");
	hipStream_t st = pf->b->ctx->stream;
	const size_t n = (size_t)pf->n, nch = pf_round_chunk(n) / (size_t)pf_chunk();
	if (!pf->d_distr) HIP_TRY(hipMalloc(&pf->d_distr, sizeof(double) * (64 + 64 + 8 + 8)));
	if (n_distr > 1) {
		if (!pf->d_scan_stats) HIP_TRY(hipMalloc(&pf->d_scan_stats, sizeof(double) * 17 * nch));
		if (!pf->d_resample_flag) HIP_TRY(hipMalloc(&pf->d_resample_flag, sizeof(int)));
		if (!pf->d_distr_ids) HIP_TRY(hipMalloc(&pf->d_distr_ids, sizeof(int) * n));
		if (!pf->d_distr_u) HIP_TRY(hipMalloc(&pf->d_distr_u, sizeof(double) * n));
	}
	double h[144];
	std::memset(h, 0, sizeof(h));
	for (int i = 0; i < n_distr; ++i)
		for (int s2 = 0; s2 < pf->S; ++s2) { h[8 * i + s2] = sigma[8 * i + s2]; h[64 + 8 * i + s2] = mean[8 * i + s2]; }
	for (int i = 0; i < n_distr; ++i) { h[136 + i] = 1.0 / n_distr; h[128 + i] = (i + 1.0) / n_distr; }   /* initializeDistributions PF.cc:199-205 */
	HIP_TRY(hipMemcpyAsync(pf->d_distr, h, sizeof(h), hipMemcpyHostToDevice, st));
	HIP_TRY(hipStreamSynchronize(st));
	pf->n_distr = n_distr;
	for (int s2 = 0; s2 < pf->S; ++s2) { pf->d.ssm_sigma[s2] = sigma[s2]; pf->d.ssm_mean[s2] = mean[s2]; }   /* initializeSampler(state_sigma[0], state_mean[0]) */
	pf->prop_valid = false;
	++pf->sampler_gen;
	return MTFHIP_OK;
}
/* the distribution draws (uniforms in (0, 1], one per particle) of the NEXT iteration, for callers that supply their own draws
 * (the parity tests); NULL: back to the device generator */
int mtfhip_pf_set_distr_draws(mtfhip_pf *pf, const double *u) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_distr_draws: NULL filter");
	if (u) pf->distr_u_next.assign(u, u + pf->n); else pf->distr_u_next.clear();
	return MTFHIP_OK;
}
/* the distribution weights the next iteration draws from (n_distr values), the distribution id of every particle's CURRENT proposal (n,
 * or NULL: with look-ahead on, k_pf_select has already drawn the pending iteration's -- see mtfhip.h), whether the last iteration
 * resampled (adaptive resampling, PF.cc:381-390) */
int mtfhip_pf_get_distributions(mtfhip_pf *pf, double *wts, int *ids, int *resampled) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_get_distributions: NULL filter");
	hipStream_t st = pf->b->ctx->stream;
	if (wts) {
		if (pf->n_distr > 1) HIP_TRY(hipMemcpyAsync(wts, pf->d_distr + 136, sizeof(double) * pf->n_distr, hipMemcpyDeviceToHost, st));
		else wts[0] = 1.0;
	}
	if (ids) {
		if (pf->n_distr > 1) HIP_TRY(hipMemcpyAsync(ids, pf->d_distr_ids, sizeof(int) * pf->n, hipMemcpyDeviceToHost, st));
		else std::memset(ids, 0, sizeof(int) * pf->n);
	}
	int flag = 1;
	const bool adaptive = pf->d.adaptive_resampling_thresh > 0 && pf->d.adaptive_resampling_thresh <= 1 && pf->d.resampling_type != 0;
	if (resampled && adaptive && pf->d_resample_flag && pf->iter > 0) HIP_TRY(hipMemcpyAsync(&flag, pf->d_resample_flag, sizeof(int), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	if (resampled) *resampled = pf->d.resampling_type == 0 ? 0 : flag;
	return MTFHIP_OK;
}
void mtfhip_pf_destroy(mtfhip_pf *pf) {
	if (!pf) return;
	pf_free(pf);
	delete pf;
}
/* shard the scoring over the communicator: rank r scores particles [r m, (r + 1) m), m = ceil(n / R), and one in-place all-gather
 * distributes the weights; proposals and resampling are replicated (identical draws and identical weights on every rank) */
int mtfhip_pf_set_comm(mtfhip_pf *pf, mtfhip_comm *c) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_comm: NULL filter");
	if (pf->peer.mailbox && c != pf->comm)
		return fail(MTFHIP_ERR_LOGIC, "pf_set_comm: this filter's peer exchange was set up over another communicator (its mailbox is sized and mapped for that one)");
	pf->comm = c;
	if (c && c->world > 1) {
		int lo, cnt, m;
		pf_shard(pf->n, c->world, c->rank, &lo, &cnt, &m);
		const size_t need = std::max(pf_round_chunk((size_t)pf->n), (size_t)m * c->world);
		if (need > pf->wts_capacity) {
			HIP_TRY(hipStreamSynchronize(pf->b->ctx->stream));
			(void)hipFree(pf->d_wts); pf->d_wts = nullptr;
			HIP_TRY(hipMalloc(&pf->d_wts, sizeof(double) * need));
			pf->wts_capacity = need;
		}
		/* Every rank scores a block of ITS OWN proposals and resamples from the gathered weights: the design relies on the ranks
		 * drawing identical proposals, i.e. on one Philox key.  One 8-byte all-gather at set-up makes a mismatch an error instead of a
		 * silently wrong estimate (r03 advisor finding: a front end that randomised seed 0 per process). */
		if (c->detached) return MTFHIP_OK;   /* (no collective: the seeds are compared through the mailboxes, mtfhip_pf_exchange_connect) */
		hipStream_t st = pf->b->ctx->stream;
		static_assert(sizeof(double) == sizeof(pf->d.seed), "the seed travels as the bits of one double");
		/* ... and the particle count: every rank lays the gathered weights out with its own n (r04 advisor) */
		for (int what = 0; what < 2; ++what) {
			double mine;
			if (what == 0) std::memcpy(&mine, &pf->d.seed, sizeof(mine)); else mine = (double)pf->n;
			std::vector<double> all((size_t)c->world);
			HIP_TRY(hipMemcpyAsync(pf->d_wts + c->rank, &mine, sizeof(double), hipMemcpyHostToDevice, st));
			HIP_TRY(hipStreamSynchronize(st));
			TRY(mtfhip_allgather_scores(c, pf->d_wts + c->rank, 1, pf->d_wts, st));
			HIP_TRY(hipMemcpyAsync(all.data(), pf->d_wts, sizeof(double) * (size_t)c->world, hipMemcpyDeviceToHost, st));
			HIP_TRY(hipStreamSynchronize(st));
			for (int q = 0; q < c->world; ++q)
				if (std::memcmp(&all[(size_t)q], &mine, sizeof(double)) != 0) {
					pf->comm = nullptr;
					if (what == 0)
						return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_comm: rank %d's filter was created with another seed than rank %d's: a sharded filter needs "
							"ONE seed on every rank (identical proposals)", q, c->rank);
					return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_comm: rank %d's filter holds %d particles, rank %d's %d: a sharded filter needs the same n_particles on every rank",
						q, (int)all[(size_t)q], c->rank, pf->n);
				}
		}
	}
	return MTFHIP_OK;
}
/* How a sharded filter's weights reach the other ranks.  COLLECTIVE (the default): one in-place all-gather per iteration,
 * enqueued by the host between the scoring and the scan.  PEER: the scoring kernel stores every weight into every rank's mailbox
 * and the scan waits for the ranks' arrival counters -- nothing is enqueued between the two kernels (mtfhip_internal.h, PfPeerPush).
 * The mailboxes are fine-grained device memory, mapped into the other ranks once, at set-up:
 *   mtfhip_pf_exchange_export   this rank's mailbox as a 64-byte hipIpc handle
 *   mtfhip_pf_exchange_connect  the handles of all ranks -> mapped; the seeds the ranks left in their mailboxes are compared
 *   mtfhip_pf_set_exchange      both, with the handles moved by the communicator itself (RCCL ranks: over its all-gather; loopback
 *                               ranks share an address space and exchange the pointers)
 * A DETACHED communicator (mtfhip_comm_create_detached: rank and world, no RCCL behind it) has no collective: the host program moves
 * the handles -- as it moves RCCL's unique id -- and only the peer exchange is available.  World sizes up to kPfMaxPeers (one node). */
static int pf_peer_alloc(mtfhip_pf *pf) {
	mtfhip_comm *c = pf->comm;
	if (!c || c->world < 2) return fail(MTFHIP_ERR_LOGIC, "pf exchange: the peer exchange belongs to a sharded filter (mtfhip_pf_set_comm with world > 1 first)");
	if (c->world > kPfMaxPeers) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "pf exchange: the peer exchange serves up to %d ranks (one node), not %d", kPfMaxPeers, c->world);
	mtfhip_pf::Peer &pr = pf->peer;
	if (pr.mailbox) return MTFHIP_OK;
	hipStream_t st = pf->b->ctx->stream;
	int lo, cnt, m;
	pf_shard(pf->n, c->world, c->rank, &lo, &cnt, &m);
	pr.cap = (std::max(pf->wts_capacity, (size_t)m * c->world) + 1) & ~(size_t)1;
	const size_t bytes = sizeof(double) * 2 * pr.cap + sizeof(unsigned long long) * mtfhip_pf::Peer::kHeadWords;
	HIP_TRY(hipExtMallocWithFlags(&pr.mailbox, bytes, hipDeviceMallocFinegrained));
	HIP_TRY(hipMemsetAsync(pr.mailbox, 0, bytes, st));
	pr.base[c->rank] = pr.mailbox;
	const unsigned long long ident[3] = {pf->d.seed, (unsigned long long)pf->n, (unsigned long long)pr.cap};
	HIP_TRY(hipMemcpyAsync(pr.seed_slot(c->rank), ident, sizeof(ident), hipMemcpyHostToDevice, st));
	HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&pr.h_err), sizeof(int), hipHostMallocMapped));
	*pr.h_err = 0;
	HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&pr.h_err_dev), pr.h_err, 0));
	HIP_TRY(hipStreamSynchronize(st));   /* the counters are zero and the seed is in place before anybody can learn where they are */
	return MTFHIP_OK;
}
/* every base[] is in place: the ranks must have been created with one seed (identical proposals: see mtfhip_pf_set_comm) */
static int pf_peer_finish(mtfhip_pf *pf) {
	mtfhip_comm *c = pf->comm;
	mtfhip_pf::Peer &pr = pf->peer;
	hipStream_t st = pf->b->ctx->stream;
	for (int q = 0; q < c->world; ++q) {
		unsigned long long theirs[3] = {0, 0, 0};   /* seed | n_particles | capacity: at fixed offsets of the peer's mailbox */
		HIP_TRY(hipMemcpyAsync(theirs, pr.seed_slot(q), sizeof(theirs), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		if (theirs[0] != pf->d.seed)
			return fail(MTFHIP_ERR_INVALID_ARG, "pf exchange: rank %d's filter was created with another seed than rank %d's: a sharded filter needs ONE seed "
				"on every rank (identical proposals)", q, c->rank);
		/* every rank addresses the peers' vectors with its own particle count and capacity: they must be the peers' too */
		if (theirs[1] != (unsigned long long)pf->n || theirs[2] != (unsigned long long)pr.cap)
			return fail(MTFHIP_ERR_INVALID_ARG, "pf exchange: rank %d's filter holds %llu particles in vectors of %llu, rank %d's %d in %zu: a sharded filter needs "
				"the same n_particles on every rank", q, theirs[1], theirs[2], c->rank, pf->n, pr.cap);
	}
	pr.on = true;
	return MTFHIP_OK;
}
int mtfhip_pf_exchange_export(mtfhip_pf *pf, void *handle64) {
	if (!pf || !handle64) return fail(MTFHIP_ERR_INVALID_ARG, "pf_exchange_export: NULL argument");
	static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI hands the mailbox handle over as 64 bytes");
	TRY(pf_peer_alloc(pf));
	hipIpcMemHandle_t h;
	HIP_TRY(hipIpcGetMemHandle(&h, pf->peer.mailbox));
	std::memcpy(handle64, &h, sizeof(h));
	return MTFHIP_OK;
}
int mtfhip_pf_exchange_connect(mtfhip_pf *pf, const void *handles /* world x 64 bytes, rank-major */) {
	if (!pf || !handles) return fail(MTFHIP_ERR_INVALID_ARG, "pf_exchange_connect: NULL argument");
	mtfhip_pf::Peer &pr = pf->peer;
	if (!pr.mailbox) return fail(MTFHIP_ERR_LOGIC, "pf_exchange_connect before pf_exchange_export");
	mtfhip_comm *c = pf->comm;
	for (int q = 0; q < c->world; ++q) {
		if (q == c->rank || pr.mapped[q]) continue;
		hipIpcMemHandle_t h;
		std::memcpy(&h, static_cast<const char *>(handles) + sizeof(h) * (size_t)q, sizeof(h));
		const hipError_t e = hipIpcOpenMemHandle(&pr.base[q], h, hipIpcMemLazyEnablePeerAccess);
		if (e != hipSuccess) return fail(MTFHIP_ERR_HIP, "pf_exchange_connect: rank %d cannot map rank %d's mailbox (hipIpcOpenMemHandle: %s)", c->rank, q, hipGetErrorString(e));
		pr.mapped[q] = true;
	}
	return pf_peer_finish(pf);
}
int mtfhip_pf_set_exchange(mtfhip_pf *pf, int mode) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_exchange: NULL filter");
	if (mode == MTFHIP_PF_EXCHANGE_COLLECTIVE) {
		if (pf->comm && pf->comm->detached) return fail(MTFHIP_ERR_LOGIC, "pf_set_exchange: a detached communicator has no collective");
		pf->peer.on = false;
		return MTFHIP_OK;
	}
	if (mode != MTFHIP_PF_EXCHANGE_PEER) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_exchange: unknown mode %d", mode);
	mtfhip_pf::Peer &pr = pf->peer;
	if (pr.mailbox && pr.base[pf->comm ? (pf->comm->rank + 1) % std::max(pf->comm->world, 1) : 0]) { pr.on = true; return MTFHIP_OK; }   /* (connected before: switched back on) */
	mtfhip_comm *c = pf->comm;
	if (c && c->detached) return fail(MTFHIP_ERR_LOGIC, "pf_set_exchange: a detached communicator cannot move the mailbox handles itself: "
		"mtfhip_pf_exchange_export, the host program's own transport, mtfhip_pf_exchange_connect");
	TRY(pf_peer_alloc(pf));
	hipStream_t st = pf->b->ctx->stream;
	if (c->loop) {
		LoopGroup *g = c->loop;
		g->mailbox[(size_t)c->rank] = pr.mailbox;
		g->barrier();
		for (int q = 0; q < c->world; ++q) pr.base[q] = g->mailbox[(size_t)q];
		g->barrier();   /* (the table may be reused by another filter of the group) */
		return pf_peer_finish(pf);
	}
	hipIpcMemHandle_t mine;
	static_assert(sizeof(hipIpcMemHandle_t) % sizeof(double) == 0, "the handle travels as doubles over the weight all-gather");
	constexpr int hw = (int)(sizeof(hipIpcMemHandle_t) / sizeof(double));
	HIP_TRY(hipIpcGetMemHandle(&mine, pr.mailbox));
	double *d_h = nullptr;
	HIP_TRY(hipMalloc(&d_h, sizeof(hipIpcMemHandle_t) * (size_t)c->world));
	std::vector<hipIpcMemHandle_t> all((size_t)c->world);
	int rc = MTFHIP_OK;
	if (hipMemcpyAsync(d_h + (size_t)c->rank * hw, &mine, sizeof(mine), hipMemcpyHostToDevice, st) != hipSuccess) rc = MTFHIP_ERR_HIP;
	if (rc == MTFHIP_OK) rc = mtfhip_allgather_scores(c, d_h + (size_t)c->rank * hw, hw, d_h, st);
	if (rc == MTFHIP_OK && hipMemcpyAsync(all.data(), d_h, sizeof(hipIpcMemHandle_t) * (size_t)c->world, hipMemcpyDeviceToHost, st) != hipSuccess) rc = MTFHIP_ERR_HIP;
	if (rc == MTFHIP_OK && hipStreamSynchronize(st) != hipSuccess) rc = MTFHIP_ERR_HIP;
	(void)hipFree(d_h);
	if (rc != MTFHIP_OK) return fail(rc, "pf_set_exchange: the exchange of the mailbox handles failed");
	return mtfhip_pf_exchange_connect(pf, all.data());
}
/* ProjectiveBase::setSampler (ProjectiveBase.cc:208-215) */
int mtfhip_pf_set_sampler(mtfhip_pf *pf, const double *sigma, const double *mean) {
	if (!pf || !sigma || !mean) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_sampler: NULL argument");
	for (int s = 0; s < pf->S; ++s) { pf->d.ssm_sigma[s] = sigma[s]; pf->d.ssm_mean[s] = mean[s]; }
	pf->prop_valid = false;   /* proposals made ahead used the old distributions */
	++pf->sampler_gen;        /* ... and so did the perturbations drawn ahead */
	if (pf->n_distr > 1 && pf->d_distr) {   /* distribution 0 of the set */
		double h[16];
		for (int s2 = 0; s2 < 8; ++s2) { h[s2] = s2 < pf->S ? sigma[s2] : 0.0; h[8 + s2] = s2 < pf->S ? mean[s2] : 0.0; }
		HIP_TRY(hipMemcpyAsync(pf->d_distr, h, sizeof(double) * 8, hipMemcpyHostToDevice, pf->b->ctx->stream));
		HIP_TRY(hipMemcpyAsync(pf->d_distr + 64, h + 8, sizeof(double) * 8, hipMemcpyHostToDevice, pf->b->ctx->stream));
		HIP_TRY(hipStreamSynchronize(pf->b->ctx->stream));
	}
	return MTFHIP_OK;
}
/* PF::initializeParticles (PF.cc:185-197) */
static int pf_initialize_particles(mtfhip_pf *pf) {
	mtfhip_batch *b = pf->b;
	hipStream_t st = b->ctx->stream;
	(void)b->view();   /* a stale single-target warp is uploaded first: the fill reads the device copy of the state */
	launch_pf_fill(pf->n, pf->S, b->d_states, pf->d_st, pf->d_ar, st);
	pf->prop_valid = false;
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
	pf->iter = 0;
	pf->pert_iter[0] = pf->pert_iter[1] = -1;
	if (pf->n_distr > 1) {   /* initializeDistributions PF.cc:199-205 */
		double h[16];
		for (int i = 0; i < 8; ++i) { h[8 + i] = i < pf->n_distr ? 1.0 / pf->n_distr : 0.0; h[i] = i < pf->n_distr ? (i + 1.0) / pf->n_distr : 0.0; }
		HIP_TRY(hipMemcpyAsync(pf->d_distr + 128, h, sizeof(h), hipMemcpyHostToDevice, b->ctx->stream));
		HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	}
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
/* max_similarity = am->getSimilarity() after am->updateModel (PF.cc:443-446, enable_learning) */
int mtfhip_pf_set_max_similarity(mtfhip_pf *pf, double max_similarity) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_max_similarity: NULL filter");
	pf->max_similarity = max_similarity;
	return MTFHIP_OK;
}

/* residual resampling (PF.cc:538-582): where every slot of the new set comes from -- d_ids -- and the order's first index, which is
 * max_wt_id (d_res_idx + n).  particle_wts are normalised in place as the reference does. */
static int pf_residual_sources(mtfhip_pf *pf, PfBuffers &bf, size_t nch, hipStream_t st) {
	const int n = pf->n;
	int *idx_in = pf->d_res_idx, *order = idx_in + n, *copies = order + n, *starts = copies + n;
	double *keys_in = pf->d_res_keys, *keys_out = keys_in + n;
	const double *total = bf.chunk_incl + (nch - 1);   /* particle_cum_wts[n - 1] */
	launch_pf_residual_prep(n, total, bf.wts, keys_in, idx_in, bf.resample_flag, st);
	if (n > 1) {
		size_t tb = pf->res_tmp_bytes;
		/* std::sort(idx, idx + n - 1, wts[a] > wts[b]): the last index is not part of the range */
		if (hipcub::DeviceRadixSort::SortPairsDescending(pf->d_res_tmp, tb, (const double *)keys_in, keys_out, (const int *)idx_in, order, n - 1, 0, 64, st) != hipSuccess)
			return fail(MTFHIP_ERR_HIP, "pf_iteration: radix sort of the particle weights failed");
	}
	HIP_TRY(hipMemcpyAsync(order + (n - 1), idx_in + (n - 1), sizeof(int), hipMemcpyDeviceToDevice, st));
	launch_pf_residual_copies(n, bf.wts, order, copies, st);
	{
		size_t tb = pf->res_tmp_bytes;
		if (hipcub::DeviceScan::ExclusiveSum(pf->d_res_tmp, tb, (const int *)copies, starts, n, st) != hipSuccess)
			return fail(MTFHIP_ERR_HIP, "pf_iteration: scan of the copy counts failed");
	}
	launch_pf_residual_map(n, order, copies, starts, pf->d_ids, st);
	bf.res_order = order;
	return MTFHIP_OK;
}

/* the launches of one iteration; publish: the estimate is also delivered to host-coherent memory (*pub_seq = the sequence
 * number to wait for, 0 when the read-back is a copy) */
static int pf_enqueue_iteration(mtfhip_pf *pf, const double *normals, const double *uniforms, bool publish, unsigned long long *pub_seq) {
	mtfhip_batch *b = pf->b;
	hipStream_t st = b->ctx->stream;
	const int n = pf->n, S = pf->S;
	const bool hom = b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY;
	PfLaunch p;
	p.n = n; p.S = S; p.dynamic_model = pf->d.dynamic_model; p.update_type = pf->d.update_type;
	p.sampler = pf->sampler; p.nz = pf->nz;
	p.likelihood_func = pf->d.likelihood_func; p.resampling_type = pf->d.resampling_type; p.mean_type = pf->d.mean_type;
	p.ar_coeff = pf->d.ar_coeff; p.measurement_sigma = pf->d.measurement_sigma; p.max_similarity = pf->max_similarity;
	for (int k = 0; k < 8; ++k) { p.sigma[k] = pf->d.ssm_sigma[k]; p.mean[k] = pf->d.ssm_mean[k]; p.init_corners[k] = b->th[0].init_corners[k]; }
	for (int k = 0; k < 12; ++k) p.init_corners_hm[k] = b->th[0].init_corners_hm[k];
	for (int k = 0; k < 9; ++k) p.aux_inv[k] = k % 4 == 0 ? 1.0 : 0.0;
	for (int k = 0; k < 6; ++k) p.canon[k] = 0.0;
	if (hom) {
		/* template corners -> unit square: the inverse of the closed-form square-to-quadrilateral map (rect_to_quad) */
		M3 sq;
		if (!rect_to_quad(0.0, 0.0, 1.0, 1.0, b->th[0].init_corners, sq)) return fail(MTFHIP_ERR_INVALID_ARG, "pf_iteration: degenerate template corners");
		const M3 inv = m3_inverse(sq);
		std::memcpy(p.aux_inv, inv.m, sizeof(p.aux_inv));
	} else if (pf->sampler == PF_SAMPLER_AFF_PTS1 || pf->sampler == PF_SAMPLER_AFF_PTS2) {
		/* the canonical points of Affine::generatePerturbation (Affine.cc:470-473): bottom right, bottom left, top centre */
		const double *ic = b->th[0].init_corners;
		p.canon[0] = ic[4]; p.canon[1] = ic[5]; p.canon[2] = ic[6]; p.canon[3] = ic[7];
		p.canon[4] = (ic[0] + ic[2]) / 2.0; p.canon[5] = (ic[1] + ic[3]) / 2.0;
		M3 m;
		for (int i = 0; i < 3; ++i) { m.m[3 * i] = p.canon[2 * i]; m.m[3 * i + 1] = p.canon[2 * i + 1]; m.m[3 * i + 2] = 1.0; }
		const double det = m.m[0] * (m.m[4] - m.m[7]) - m.m[1] * (m.m[3] - m.m[6]) + (m.m[3] * m.m[7] - m.m[4] * m.m[6]);
		if (det == 0) return fail(MTFHIP_ERR_INVALID_ARG, "pf_iteration: degenerate template corners");
		const M3 inv = m3_inverse(m);
		std::memcpy(p.aux_inv, inv.m, sizeof(p.aux_inv));
	}
	p.seed = pf->d.seed; p.iter = pf->iter;
	p.normals = nullptr; p.uniforms = nullptr;
	if (normals) {
		HIP_TRY(hipMemcpyAsync(pf->d_normals, normals, sizeof(double) * (size_t)n * pf->nz, hipMemcpyHostToDevice, st));
		p.normals = pf->d_normals;
	}
	if (uniforms) {
		HIP_TRY(hipMemcpyAsync(pf->d_uniforms, uniforms, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
		p.uniforms = pf->d_uniforms;
	}
	const size_t nch = pf_round_chunk((size_t)n) / (size_t)pf_chunk();
	const bool adaptive = pf->d.adaptive_resampling_thresh > 0 && pf->d.adaptive_resampling_thresh <= 1 && pf->d.resampling_type != 0;
	const bool mixture = pf->n_distr > 1;
	p.n_distr = pf->n_distr; p.distr_uniforms = nullptr;
	p.min_distr_wt = pf->d.min_distr_wt;
	p.min_eff_particles = adaptive ? pf->d.adaptive_resampling_thresh * n : 0.0;   /* PF.cc:117 */
	if (mixture && !pf->distr_u_next.empty()) {
		HIP_TRY(hipMemcpyAsync(pf->d_distr_u, pf->distr_u_next.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
		HIP_TRY(hipStreamSynchronize(st));   /* (the vector may be replaced before the copy has run) */
		p.distr_uniforms = pf->d_distr_u;
	}
	const mtfhip_comm *c = pf->comm;
	const bool sharded = c && c->world > 1;
	int lo = 0, cnt = n, m = n;
	if (sharded) pf_shard(n, c->world, c->rank, &lo, &cnt, &m);
	/* this iteration's proposals (PF.cc:307-335): left behind by the previous selection pass, or made now */
	PfBuffers bf;
	bf.distr_sigma = mixture ? pf->d_distr : nullptr; bf.distr_mean = mixture ? pf->d_distr + 64 : nullptr;
	bf.distr_cum = mixture ? pf->d_distr + 128 : nullptr; bf.distr_wts = mixture ? pf->d_distr + 136 : nullptr;
	bf.distr_ids = mixture ? pf->d_distr_ids : nullptr;
	bf.scan_stats = (mixture || adaptive) ? pf->d_scan_stats : nullptr;
	bf.resample_flag = adaptive ? pf->d_resample_flag : nullptr;
	const bool ahead = pf->prop_valid && !normals && !p.distr_uniforms && pf->prop_iter == pf->iter && pf->prop_corners_epoch == b->corners_epoch;
	if (!ahead) {
		TimedScope ts(b->ctx, "pf_propose");
		launch_pf_propose(b->desc.ssm, p, bf, pf->d_st, pf->d_ar, pf->d_prop[pf->pc], pf->d_prop_ar[pf->pc], st);
	}
	/* the next iteration's can be made by this one's selection pass when its draws are the device generator's and nothing the
	 * sampler reads moves in between (MeanType::Corners re-bases the SSM on the mean corners: setCorners, PF.cc:434-436) */
	const bool lookahead = pf->lookahead_enabled && !normals && !p.distr_uniforms && pf->d.mean_type != 2;
	bf.st = pf->d_st; bf.ar = pf->d_ar; bf.prop = pf->d_prop[pf->pc]; bf.prop_ar = pf->d_prop_ar[pf->pc];
	bf.next = pf->d_prop[1 - pf->pc]; bf.next_ar = pf->d_prop_ar[1 - pf->pc];
	/* peer exchange: this iteration's weights live in the mailbox vector of its parity, here and on every other rank */
	const bool peer = sharded && pf->peer.on;
	if (sharded && !peer && c->detached)
		return fail(MTFHIP_ERR_LOGIC, "pf_iteration: a filter sharded over a detached communicator needs the peer exchange connected first (mtfhip_pf_exchange_export / _connect)");
	PfPeerPush push{};
	PfPeerWait pwait{};
	/* (the exchange's epoch and the expected arrival counts are committed only once the storing launch has been enqueued: a call that
	 * fails before it must not leave this rank one exchange ahead of its peers -- every later wait would then spin to its limit) */
	unsigned long long new_epoch = pf->peer.epoch, new_expected[kPfMaxPeers] = {};
	if (peer) {
		mtfhip_pf::Peer &pr = pf->peer;
		new_epoch = pr.epoch + 1;
		const int parity = (int)(new_epoch & 1);
		push.world = pwait.world = c->world; push.rank = pwait.rank = c->rank;
		for (int q = 0; q < c->world; ++q) {
			int qlo, qcnt, qm;
			pf_shard(n, c->world, q, &qlo, &qcnt, &qm);
			new_expected[q] = pr.expected[q] + (qcnt > 0 ? 1u : 0u);   /* one arrival per storing launch; an empty block launches nothing */
			push.wts[q] = pr.wts(q, parity); push.counters[q] = pr.counters(q);
			push.arrive = reinterpret_cast<unsigned *>(pr.counters(c->rank) + kPfMaxPeers);   /* (the word behind the counters) */
			pwait.expected[q] = new_expected[q];
		}
		pwait.counters = pr.counters(c->rank); pwait.err = pr.h_err_dev;
	}
	bf.wts = peer ? pf->peer.wts(c->rank, (int)(new_epoch & 1)) : pf->d_wts; bf.sim = nullptr; bf.cum = pf->d_cum; bf.chunk_tot = pf->d_chunk; bf.chunk_incl = pf->d_chunk + nch; bf.sub16 = pf->d_chunk + 2 * nch;
	bf.res_order = nullptr;
	bf.parts = pf->d_parts; bf.gparts = pf->d_gparts; bf.out = pf->d_out; bf.ids = pf->d_ids; bf.counters = pf->d_counters;
	/* scoring: setState -> updatePixVals -> updateSimilarity -> likelihood per particle (PF.cc:341-365); sharded: this rank's block */
	{
		TimedScope ts(b->ctx, "pf_score");
		TRY(score_block_dev(b, bf.prop, lo, cnt, bf.wts, bf.sim, p.likelihood_func, p.measurement_sigma, p.max_similarity, peer ? &push : nullptr));
	}
	if (peer) {
		pf->peer.epoch = new_epoch;
		for (int q = 0; q < c->world; ++q) pf->peer.expected[q] = new_expected[q];
		/* Loopback ranks share one GPU and its few hardware queues: a spinning scan of one rank could sit in front of the scoring launch of
		 * another.  There the host threads meet once every rank's scoring has completed, and the waits below pass at once; the stores, the
		 * counters and the two mailbox vectors are exercised as between GPUs.  RCCL ranks (one GPU each) enqueue straight through. */
		TimedScope ts(b->ctx, "pf_peer_exchange");   /* (empty between GPUs: a marker that this exchange was the one in use) */
		if (c->loop) { HIP_TRY(hipStreamSynchronize(st)); c->loop->barrier(); }
	} else if (sharded) {
		TimedScope ts(b->ctx, "pf_allgather");
		TRY(mtfhip_allgather_scores(pf->comm, pf->d_wts + (size_t)c->rank * m, m, pf->d_wts, st));   /* in place: block r of m weights at r m */
	}
	{
		TimedScope ts(b->ctx, "pf_resample");
		unsigned long long seq = 0;
		if (publish && b->h_acc_dev) seq = ++b->acc_seq;
		/* small sets with multinomial resampling: the selection pass scans the weights itself (k_pf_select, LOCAL) -- MTFHIP_PF_LOCAL=0: the
		 * scan launch in front, as for every other case */
		const bool local_env = pf->local_enabled, pert_env = pf->pert_ahead_enabled;   /* (read when the filter was created) */
		PfSelectPlan plan;
		plan.estimate = (publish || !pf->skip_unread_estimates) ? 1 : 0;   /* (publish == false: a chained iteration whose estimate nobody reads) */
		plan.local = local_env && (p.resampling_type == 1 || p.resampling_type == 2) && !mixture && !bf.scan_stats && n <= pf_local_max();
		if (plan.local) { if (peer) plan.wait = &pwait; }
		else if (p.resampling_type != 0 || mixture) launch_pf_scan(p, bf, peer ? &pwait : nullptr, st);   /* (the distribution weights follow the particle weights whatever the resampling) */
		else if (peer) launch_pf_peer_wait(pwait, st);
		if (p.resampling_type == 3) TRY(pf_residual_sources(pf, bf, nch, st));
		/* the perturbations of the next iteration, drawn two launches ago; this launch draws those of the one after */
		const unsigned it = pf->iter;
		if (lookahead && pert_env && !mixture) {
			const int bi = (int)((it + 1) & 1u);
			if (pf->pert_iter[bi] == (long)it + 1 && pf->pert_epoch[bi] == b->corners_epoch && pf->pert_gen[bi] == pf->sampler_gen) plan.pert_in = pf->d_pert[bi];
			plan.pert_out = pf->d_pert[it & 1u];
		}
		launch_pf_select(b->desc.ssm, p, bf, lookahead ? 1 : 0, seq ? b->h_acc_dev : nullptr, b->h_flag_dev, seq, plan, st);
		if (plan.pert_out) { pf->pert_iter[it & 1u] = (long)it + 2; pf->pert_epoch[it & 1u] = b->corners_epoch; pf->pert_gen[it & 1u] = pf->sampler_gen; }
		if (pub_seq) *pub_seq = seq;
	}
	++pf->iter;
	pf->distr_u_next.clear();   /* (handed-in distribution draws are for one iteration) */
	pf->prop_valid = lookahead;
	if (lookahead) { pf->pc = 1 - pf->pc; pf->prop_iter = pf->iter; pf->prop_corners_epoch = b->corners_epoch; }
	return MTFHIP_OK;
}
/* the estimate (32 doubles) comes back through host-coherent pinned memory + the flag the host spins on, like every other
 * per-iteration result of the library, and becomes the SSM's state (PF.cc:421-437) */
static int pf_collect_estimate(mtfhip_pf *pf, unsigned long long pub_seq, double *update_norm) {
	mtfhip_batch *b = pf->b;
	hipStream_t st = b->ctx->stream;
	double out[32];
	if (pub_seq) {
		TRY(wait_host_flag(b, pub_seq));
		std::memcpy(out, b->h_acc, sizeof(out));
	} else {
		HIP_TRY(hipMemcpyAsync(out, pf->d_out, sizeof(out), hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
	}
	if (pf->peer.on && pf->peer.h_err && __atomic_load_n(pf->peer.h_err, __ATOMIC_ACQUIRE) != 0)
		return fail(MTFHIP_ERR_HIP, "pf_iteration: rank %d gave up waiting for another rank's weights (peer-store exchange): the estimate is not valid", pf->comm ? pf->comm->rank : 0);
	if (pf->d.mean_type == 2) TRY(mtfhip_ssm_set_corners(b, out + 10));
	else TRY(mtfhip_ssm_set_state(b, out));
	double un = 0;
	for (int q = 0; q < 8; ++q) { const double d = pf->prev_corners[q] - b->th[0].corners[q]; un += d * d; }
	std::memcpy(pf->prev_corners, b->th[0].corners, sizeof(pf->prev_corners));
	if (update_norm) *update_norm = un;
	return MTFHIP_OK;
}

/* One iteration of the loop of nt::PF::update (PF.cc:260-447).  normals: n x nz standard normals (nz = 10 with corner based
 * homography sampling, 6 / 8 for the affine samplers, else the state size), uniforms: n draws in (0, 1]; host arrays, or NULL:
 * the device generator (Philox4x32-10 keyed by desc.seed, the iteration count and the particle).  update_norm: squared corner
 * change of the estimate (PF.cc:438-439). */
int mtfhip_pf_iteration(mtfhip_pf *pf, const double *normals, const double *uniforms, double *update_norm) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_iteration: NULL filter");
	if (!pf->initialized) return fail(MTFHIP_ERR_LOGIC, "pf_iteration before pf_initialize");
	mtfhip_batch *b = pf->b;
	FLUSH_AM(b);   /* (the candidates carry their own warps: the batch's CURR_PTS are not read) */
	TRY(need_image(b));
	unsigned long long seq = 0;
	TRY(pf_enqueue_iteration(pf, normals, uniforms, true, &seq));
	return pf_collect_estimate(pf, seq, update_norm);
}
/* nt::PF::update (PF.cc:207-447) with the device generator: up to max_iters iterations, stop when the estimate's corners move
 * by less than epsilon; reset_to_mean re-initialises the particles at the estimate.  With a negative epsilon the test can never
 * fire and nothing on the host depends on an intermediate estimate (unless MeanType::Corners re-bases the SSM every iteration):
 * the iterations are then enqueued back to back and only the last one reports to the host. */
int mtfhip_pf_update(mtfhip_pf *pf, int *n_iters) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_update: NULL filter");
	if (!pf->initialized) return fail(MTFHIP_ERR_LOGIC, "pf_update before pf_initialize");
	int it = 0;
	if (pf->d.epsilon < 0 && pf->d.mean_type != 2 && pf->d.max_iters > 0) {
		mtfhip_batch *b = pf->b;
		FLUSH_AM(b);
		TRY(need_image(b));
		unsigned long long seq = 0;
		for (; it < pf->d.max_iters; ++it) TRY(pf_enqueue_iteration(pf, nullptr, nullptr, it == pf->d.max_iters - 1, &seq));
		TRY(pf_collect_estimate(pf, seq, nullptr));
	} else {
		for (; it < pf->d.max_iters; ++it) {
			double un = 0;
			TRY(mtfhip_pf_iteration(pf, nullptr, nullptr, &un));
			if (un < pf->d.epsilon) { ++it; break; }
		}
	}
	if (pf->d.reset_to_mean) TRY(pf_initialize_particles(pf));
	if (n_iters) *n_iters = it;
	return MTFHIP_OK;
}
int mtfhip_pf_get_particles(mtfhip_pf *pf, double *states, double *ars, double *wts, int *resample_ids) {
	if (!pf) return fail(MTFHIP_ERR_INVALID_ARG, "pf_get_particles: NULL filter");
	hipStream_t st = pf->b->ctx->stream;
	const size_t nS = (size_t)pf->n * pf->S;
	if (states) HIP_TRY(hipMemcpyAsync(states, pf->d_st, sizeof(double) * nS, hipMemcpyDeviceToHost, st));
	if (ars) HIP_TRY(hipMemcpyAsync(ars, pf->d_ar, sizeof(double) * nS, hipMemcpyDeviceToHost, st));
	if (wts) HIP_TRY(hipMemcpyAsync(wts, pf->last_wts(), sizeof(double) * pf->n, hipMemcpyDeviceToHost, st));
	if (resample_ids) HIP_TRY(hipMemcpyAsync(resample_ids, pf->d_ids, sizeof(int) * pf->n, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	return MTFHIP_OK;
}
int mtfhip_pf_set_particles(mtfhip_pf *pf, const double *states, const double *ars) {
	if (!pf || !states) return fail(MTFHIP_ERR_INVALID_ARG, "pf_set_particles: NULL argument");
	hipStream_t st = pf->b->ctx->stream;
	const size_t nS = (size_t)pf->n * pf->S;
	HIP_TRY(hipMemcpyAsync(pf->d_st, states, sizeof(double) * nS, hipMemcpyHostToDevice, st));
	if (ars) HIP_TRY(hipMemcpyAsync(pf->d_ar, ars, sizeof(double) * nS, hipMemcpyHostToDevice, st));
	else HIP_TRY(hipMemsetAsync(pf->d_ar, 0, sizeof(double) * nS, st));
	HIP_TRY(hipStreamSynchronize(st));
	pf->prop_valid = false;
	return MTFHIP_OK;
}
double mtfhip_pf_max_similarity(const mtfhip_pf *pf) { return pf ? pf->max_similarity : 0.0; }

} /* extern "C" */
