/*
 * api_am.hip -- the AppearanceModel entry points (ImageBase, SSD, NCC, MI, second order) and the deferred-fusion layer behind them
 * (C-ABI implementation, include/mtfhip.h; shared declarations: mtfhip_api_internal.h)
 *
 * No CPU fallback exists: every entry point either runs its HIP kernels or returns an error.
 */
#include "mtfhip_api_internal.h"

extern "C" {

/* ------------------------------------------------------------------ ImageBase */
int mtfhip_am_initialize_pix_vals(mtfhip_batch *b, const double *pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_pix_vals: NULL batch");
	TRY(need_image(b));
	const double *dp;
	TRY(resolve_pts(b, pts, MTFHIP_BUF_CURR_PTS, 2 * (size_t)b->NP, &dp));
	++b->frame_count;   /* ImageBase.cc:74 */
	{
		TimedScope ts(b->ctx, "sample");
		launch_sample(b->view(), b->ctx->img, dp, b->buf[MTFHIP_BUF_I0], b->norm_mult, b->norm_add, b->ctx->stream);
	}
	if (!b->init_pix_vals) {
		HIP_TRY(hipMemcpyAsync(b->buf[MTFHIP_BUF_IT], b->buf[MTFHIP_BUF_I0], sizeof(double) * b->N * b->B, hipMemcpyDeviceToDevice, b->ctx->stream));
		b->init_pix_vals = true;
		b->it_valid = true;
	}
	return MTFHIP_OK;
}
/* SSD::updateModel AM/src/SSD.cc:49-75, NCC::updateModel AM/src/NCC.cc:539-566 (the search methods call it at the end of update()
 * when enable_learning is set, NT/ESM.cc:293-295): template <- average with the patch at pts, then AppearanceModel::reinitialize
 * (AppearanceModel.h:119-123).  learning_rate outside [0, 1] = running average over the frames seen (SSD.cc:45). */
int mtfhip_am_update_model(mtfhip_batch *b, const double *pts, double learning_rate) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_model: NULL batch");
	if (b->desc.am == MTFHIP_AM_MI) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "updateModel :: MI has no online template update in the reference either");
	if (b->desc.am != MTFHIP_AM_SSD && b->desc.am != MTFHIP_AM_NCC) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "updateModel :: appearance model %d", b->desc.am);
	TRY(single_channel(b, "update_model"));
	if (!b->init_pix_vals) return fail(MTFHIP_ERR_LOGIC, "update_model before initializePixVals");
	TRY(need_image(b));
	TRY(ensure_df(b));
	const double *dp;
	TRY(resolve_pts(b, pts, MTFHIP_BUF_CURR_PTS, 2 * (size_t)b->NP, &dp));
	++b->frame_count;
	const int running = (learning_rate < 0 || learning_rate > 1) ? 1 : 0;
	{
		TimedScope ts(b->ctx, "sample");
		launch_update_model(b->view(), b->ctx->img, dp, b->buf[MTFHIP_BUF_I0], b->norm_mult, b->norm_add, (double)b->frame_count,
			learning_rate, running, b->ctx->stream);
	}
	/* everything cached about the template is void: buffer versions (Gram / moment caches), NCC's template scalars and moments */
	touch_all(b); b->lz.it_epoch = -1;
	/* reinitialize(): the initialize* functions called again refresh what depends on the template only (NCC: mean and norm of I0,
	 * NCC.cc:50-95; SSD: nothing; initializeGrad / initializeHess hold no template state for SSD and NCC) */
	if (b->init_sim) TRY(mtfhip_am_initialize_similarity(b));
	if (b->desc.am == MTFHIP_AM_NCC && b->d_ncc_tm && b->j0_is_template) TRY(ncc_template_moments(b));
	return MTFHIP_OK;
}
static int do_update_pix_vals(mtfhip_batch *b, const double *pts) {
	TRY(need_image(b));
	TRY(protect_stale(b, false, false));   /* IT is about to change: gradients skipped by a fused launch keep the old IT */
	const double *dp;
	TRY(resolve_pts(b, pts, MTFHIP_BUF_CURR_PTS, 2 * (size_t)b->NP, &dp));
	TimedScope ts(b->ctx, "sample");
	launch_sample(b->view(), b->ctx->img, dp, b->buf[MTFHIP_BUF_IT], b->norm_mult, b->norm_add, b->ctx->stream);
	touch(b, MTFHIP_BUF_IT);
	b->lz.it_epoch = pts ? -1 : b->lz.epoch;
	b->it_valid = true;
	return MTFHIP_OK;
}
int mtfhip_am_update_pix_vals(mtfhip_batch *b, const double *pts) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_pix_vals: NULL batch");
	if (b->lz.enabled && !pts && b->init_pix_vals && b->have_corners && b->ctx->img.data && b->ctx->img.channels == 1) {
		if (b->lz.pv || b->lz.sim) FLUSH(b);
		b->lz.pv = ++b->lz.seq;
		b->it_valid = true;
		return MTFHIP_OK;
	}
	FLUSH(b);
	return do_update_pix_vals(b, pts);
}
static int pix_grad_common(mtfhip_batch *b, const double *pts, bool warped, bool init) {
	TRY(need_image(b));
	const double *dp;
	if (warped) TRY(resolve_pts(b, pts, MTFHIP_BUF_GRAD_PTS, 8 * (size_t)b->NP, &dp));
	else TRY(resolve_pts(b, pts, MTFHIP_BUF_CURR_PTS, 2 * (size_t)b->NP, &dp));
	double *dst = b->buf[init ? MTFHIP_BUF_DI0_DX : MTFHIP_BUF_DIT_DX];
	{
		TimedScope ts(b->ctx, warped ? "warped_img_grad" : "img_grad");
		if (warped) launch_warped_img_grad(b->view(), b->ctx->img, dp, dst, b->desc.grad_eps, b->norm_mult, b->ctx->stream);
		else launch_img_grad(b->view(), b->ctx->img, dp, dst, b->desc.grad_eps, b->norm_mult, b->ctx->stream);
	}
	touch(b, init ? MTFHIP_BUF_DI0_DX : MTFHIP_BUF_DIT_DX);
	if (init) b->j0_is_template = false;
	if (init && !b->init_pix_grad) {
		HIP_TRY(hipMemcpyAsync(b->buf[MTFHIP_BUF_DIT_DX], b->buf[MTFHIP_BUF_DI0_DX], sizeof(double) * 2 * b->N * b->B, hipMemcpyDeviceToDevice, b->ctx->stream));
		b->init_pix_grad = true;
		b->dit_valid = true;
	}
	if (!init) b->dit_valid = true;
	return MTFHIP_OK;
}
int mtfhip_am_initialize_pix_grad(mtfhip_batch *b, const double *pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_pix_grad: NULL batch");
	return pix_grad_common(b, pts, false, true);
}
static int lazy_record_pix_grad(mtfhip_batch *b, int kind) {
	if (b->lz.pg || b->lz.pj) FLUSH(b);
	b->lz.pg = ++b->lz.seq; b->lz.pg_kind = kind;
	b->dit_valid = true;
	return MTFHIP_OK;
}
int mtfhip_am_update_pix_grad(mtfhip_batch *b, const double *pts) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_pix_grad: NULL batch");
	if (b->lz.enabled && !pts && b->have_corners && b->ctx->img.data && b->ctx->img.channels == 1)
		return lazy_record_pix_grad(b, 1);
	FLUSH(b);
	return pix_grad_common(b, pts, false, false);
}
int mtfhip_am_initialize_pix_grad_warped(mtfhip_batch *b, const double *gp) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_pix_grad_warped: NULL batch");
	return pix_grad_common(b, gp, true, true);
}
int mtfhip_am_update_pix_grad_warped(mtfhip_batch *b, const double *gp) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_pix_grad_warped: NULL batch");
	/* only after a deferred update_grad_pts: the fused kernel derives the warped gradient points from the current warp */
	if (b->lz.enabled && !gp && b->lz.gp && b->ctx->img.data && b->ctx->img.channels == 1)
		return lazy_record_pix_grad(b, 2);
	FLUSH(b);
	return pix_grad_common(b, gp, true, false);
}


/* ------------------------------------------------------------------ NCC (AM/src/NCC.cc) */
int push_ncc(mtfhip_batch *b) {
	std::vector<double> s(8 * (size_t)b->B, 0.0);
	for (int t = 0; t < b->B; ++t) {
		const TargetHost &h = b->th[t];
		double *p = &s[8 * t];
		p[0] = h.I0_mean; p[1] = h.c; p[2] = h.It_mean; p[3] = h.b; p[4] = h.f; p[5] = h.gmean;
	}
	HIP_TRY(hipMemcpyAsync(b->d_ncc, s.data(), sizeof(double) * s.size(), hipMemcpyHostToDevice, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	return MTFHIP_OK;
}
static int ncc_mean_of(mtfhip_batch *b, int buf, double TargetHost::*dst) {
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "ncc_stats");
		launch_vec_sum(b->view(), b->buf[buf], b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) b->th[t].*dst = b->h_acc[(size_t)t * ACC_COUNT + ACC_RR] / (double)b->N;
	return MTFHIP_OK;
}
/* NCC::initializeSimilarity NCC.cc:55-95 */
static int ncc_initialize_similarity(mtfhip_batch *b) {
	TRY(ncc_mean_of(b, MTFHIP_BUF_I0, &TargetHost::I0_mean));
	if (!b->init_sim)
		for (auto &h : b->th) h.It_mean = h.I0_mean;
	TRY(push_ncc(b));
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "ncc_stats");
		launch_ncc_centered(b->view(), b->d_ncc, b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) {
		TargetHost &h = b->th[t];
		h.c = std::sqrt(b->h_acc[(size_t)t * ACC_COUNT + ACC_G + 2]);
		if (!b->init_sim) { h.f = 1; h.b = h.c; }
	}
	b->init_sim = true;
	return push_ncc(b);
}
/* NCC::updateSimilarity NCC.cc:124-161 */
static int ncc_update_similarity(mtfhip_batch *b) {
	TRY(ncc_mean_of(b, MTFHIP_BUF_IT, &TargetHost::It_mean));
	TRY(push_ncc(b));
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "ncc_stats");
		launch_ncc_centered(b->view(), b->d_ncc, b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) {
		TargetHost &h = b->th[t];
		h.a = b->h_acc[(size_t)t * ACC_COUNT + ACC_G + 0];
		h.b = std::sqrt(b->h_acc[(size_t)t * ACC_COUNT + ACC_G + 1]);
		double bc = h.b * h.c;
		h.f = h.a / bc;
	}
	return push_ncc(b);
}
/* NCC::updateCurrGrad / updateInitGrad NCC.cc:163-234 */
static int ncc_update_grad(mtfhip_batch *b, int curr) {
	if (b->ncc_host_newer) { TRY(push_ncc(b)); b->ncc_host_newer = false; }
	int nblk = simple_blocks_per_target(b->N);
	double *dst = b->buf[curr ? MTFHIP_BUF_DF_DIT : MTFHIP_BUF_DF_DI0];
	{
		TimedScope ts(b->ctx, "ncc_grad");
		launch_ncc_grad(b->view(), b->d_ncc, curr, dst, b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) b->th[t].gmean = b->h_acc[(size_t)t * ACC_COUNT + ACC_RR] / (double)b->N;
	TRY(push_ncc(b));
	TimedScope ts(b->ctx, "ncc_grad");
	launch_sub_mean(b->view(), dst, b->d_ncc, b->ctx->stream);
	return MTFHIP_OK;
}
/* NCC::cmptInitHessian / cmptCurrHessian / cmptSelfHessian NCC.cc:282-389 (fast_hess = 0);
 * kind 0 init, 1 curr, 2 self.  H is column-major S x S per target. */
int ncc_hessian_from_cache(mtfhip_batch *b, int j_buf, int kind, double *H);
static int ncc_hessian(mtfhip_batch *b, int j_buf, int kind, double *H) {
	if (ncc_hessian_from_cache(b, j_buf, kind, H)) return MTFHIP_OK;
	if (b->ncc_host_newer) { TRY(push_ncc(b)); b->ncc_host_newer = false; }
	const int S = b->S;
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "ncc_hess");
		launch_col_sum(b->view(), b->buf[j_buf], b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	std::vector<double> cm(8 * (size_t)b->B, 0.0);
	for (int t = 0; t < b->B; ++t)
		for (int s = 0; s < S; ++s) cm[8 * t + s] = b->h_acc[(size_t)t * ACC_COUNT + ACC_G + s] / (double)b->N;
	HIP_TRY(hipMemcpyAsync(b->d_colmean, cm.data(), sizeof(double) * cm.size(), hipMemcpyHostToDevice, b->ctx->stream));
	TRY(push_ncc(b));
	{
		TimedScope ts(b->ctx, "ncc_hess");
		launch_ncc_hess(b->view(), b->d_ncc, b->d_colmean, b->buf[j_buf], b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) {
		const double *acc = b->h_acc + (size_t)t * ACC_COUNT;
		const double f = b->th[t].f;
		double *Ht = H + (size_t)t * S * S;
		int k = 0;
		for (int r = 0; r < 8; ++r)
			for (int c = r; c < 8; ++c) {
				if (r < S && c < S) {
					const double G = acc[ACC_H + k];
					const double ut_r = acc[ACC_G + r], ut_c = acc[ACC_G + c];
					const double u0_r = acc[ACC_G2 + r], u0_c = acc[ACC_G2 + c];
					double v;
					if (kind == 0) v = -f * G - ut_r * u0_c - u0_r * ut_c + 3 * u0_r * u0_c;
					else if (kind == 1) v = -f * G - ut_r * u0_c - u0_r * ut_c + 3 * ut_r * ut_c;
					else v = -G + ut_r * ut_c;
					Ht[c * S + r] = v; Ht[r * S + c] = v;
				}
				++k;
			}
	}
	return MTFHIP_OK;
}


/* ------------------------------------------------------------------ MI (AM/src/MI.cc) */
static int mi_read_f(mtfhip_batch *b) {
	std::vector<double> f(b->B);
	HIP_TRY(hipMemcpyAsync(f.data(), b->d_mi_f, sizeof(double) * b->B, hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	for (int t = 0; t < b->B; ++t) b->th[t].f = f[t];
	return MTFHIP_OK;
}
/* workgroups per target for the MI histogram / Hessian passes: enough to fill the chip (~4 per CU over the batch), few
 * enough that every wave amortises its register-resident bin accumulators over many 64-pixel chunks and that the
 * fixed-order finish has short columns to add */
int mi_blocks(const mtfhip_batch *b) {
	int nb = 1024 / b->B;
	if (nb < 1) nb = 1;
	return std::min(nb, simple_blocks_per_target(b->N));
}
/* mode 0 initialise (A = B = I0), 1 update (A = It, B = I0), 2 self (A = B = It) */
static int mi_hist_pass(mtfhip_batch *b, int mode, int first_init) {
	const int nb = b->desc.mi_n_bins, nblk = mi_blocks(b);
	const double *A = b->buf[mode == 0 ? MTFHIP_BUF_I0 : MTFHIP_BUF_IT];
	const double *Bv = b->buf[mode == 2 ? MTFHIP_BUF_IT : MTFHIP_BUF_I0];
	TimedScope ts(b->ctx, "mi_hist");
	launch_mi_hist(b->view(), nb, b->mi_hist_norm, A, Bv, b->d_mi_part, nblk, b->mi_row_len, b->ctx->stream);
	launch_mi_hist_finish(b->view(), nb, b->desc.mi_pre_seed, b->mi_hist_norm, mode, first_init, b->d_mi_part, nblk,
		b->mi_row_len, b->d_mi_tb, b->d_mi_f, b->ctx->stream);
	if (mode == 2) b->lz.mi_self_it = b->lz.ver[MTFHIP_BUF_IT];
	return MTFHIP_OK;
}
/* kind 0 init (MI.cc:461-513), 1 curr (:603-637), 2 self (:515-601, the returned second pass) */
static int mi_hessian(mtfhip_batch *b, int j_buf, int kind, double *H) {
	const int nb = b->desc.mi_n_bins, nblk = mi_blocks(b), S = b->S;
	if (kind == 2) {   /* cmptSelfHist MI.cc:639-659 -- unless the fused histogram pass of this iteration took it along */
		b->lz.mi_want_self = true;
		if (b->lz.no_cache || b->lz.mi_self_it != b->lz.ver[MTFHIP_BUF_IT]) TRY(mi_hist_pass(b, 2, 0));
	}
	const double *A = b->buf[kind == 0 ? MTFHIP_BUF_I0 : MTFHIP_BUF_IT];
	const double *Bv = b->buf[kind == 1 ? MTFHIP_BUF_I0 : MTFHIP_BUF_IT];
	const int table = kind == 0 ? MI_T_INIT : (kind == 1 ? MI_T_CURR : MI_T_SELF);
	const int joint = kind == 2 ? MI_SELF_JOINT : MI_JOINT;
	const int hist = kind == 0 ? MI_HIST_INIT : MI_HIST_CURR;
	{
		TimedScope ts(b->ctx, "mi_hess");
		launch_mi_hess(b->view(), nb, b->mi_hist_norm, A, Bv, b->d_mi_tb, table, kind == 0, b->buf[j_buf], b->d_mi_part, nblk,
			b->mi_row_len, b->ctx->stream);
		/* two short launches instead of one long one: the column sums are spread over the chip, the assembly reads one row */
		launch_finish_rows(b->d_mi_part, nblk, b->mi_row_len, b->d_mi_red, b->B, b->ctx->stream);
		launch_mi_hess_finish(b->view(), nb, b->d_mi_red, 1, b->mi_row_len, b->d_mi_tb, joint, hist, kind == 0, b->d_mi_H,
			b->ctx->stream);
	}
	std::vector<double> h(64 * (size_t)b->B);
	HIP_TRY(hipMemcpyAsync(h.data(), b->d_mi_H, sizeof(double) * h.size(), hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	for (int t = 0; t < b->B; ++t) std::memcpy(H + (size_t)t * S * S, &h[64 * t], sizeof(double) * S * S);
	return MTFHIP_OK;
}
static int mi_grad(mtfhip_batch *b, int curr) {
	const int nb = b->desc.mi_n_bins;
	TimedScope ts(b->ctx, "mi_grad");
	launch_mi_factor(b->view(), nb, curr, b->d_mi_tb, b->ctx->stream);
	if (curr) launch_mi_grad(b->view(), nb, b->mi_hist_norm, b->buf[MTFHIP_BUF_IT], b->buf[MTFHIP_BUF_I0], b->d_mi_tb, MI_T_CURR,
		b->buf[MTFHIP_BUF_DF_DIT], b->ctx->stream);
	else launch_mi_grad(b->view(), nb, b->mi_hist_norm, b->buf[MTFHIP_BUF_I0], b->buf[MTFHIP_BUF_IT], b->d_mi_tb, MI_T_INIT,
		b->buf[MTFHIP_BUF_DF_DI0], b->ctx->stream);
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ AppearanceModel */
static int am_supported(mtfhip_batch *b, const char *fn) {
	if (b->desc.am == MTFHIP_AM_SSD || b->desc.am == MTFHIP_AM_NCC || b->desc.am == MTFHIP_AM_MI) return MTFHIP_OK;
	return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "%s :: appearance model %d is not available on the device path yet", fn, b->desc.am);
}

int mtfhip_am_initialize_similarity(mtfhip_batch *b) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_similarity: NULL batch");
	TRY(am_supported(b, "initializeSimilarity"));
	if (b->desc.am == MTFHIP_AM_NCC) return ncc_initialize_similarity(b);
	if (b->desc.am == MTFHIP_AM_MI) {
		/* MI::initializeSimilarity MI.cc:207-287 */
		const int first = b->init_sim ? 0 : 1;
		TRY(mi_hist_pass(b, 0, first));
		if (first) TRY(mi_read_f(b));
		b->init_sim = true;
		return MTFHIP_OK;
	}
	if (b->init_sim) return MTFHIP_OK;
	HIP_TRY(hipMemsetAsync(b->buf[MTFHIP_BUF_DF_DI0], 0, sizeof(double) * b->N * b->B, b->ctx->stream));
	for (auto &h : b->th) h.f = 0;
	b->init_sim = true;
	return MTFHIP_OK;
}
int mtfhip_am_initialize_grad(mtfhip_batch *b) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_grad: NULL batch");
	TRY(am_supported(b, "initializeGrad"));
	if (b->desc.am == MTFHIP_AM_MI) {
		/* MI::initializeGrad MI.cc:299-332: df_dI0 from the initial tables, df_dIt = df_dI0 */
		if (b->init_grad) return MTFHIP_OK;
		{
			TimedScope ts(b->ctx, "mi_grad");
			launch_mi_grad(b->view(), b->desc.mi_n_bins, b->mi_hist_norm, b->buf[MTFHIP_BUF_I0], b->buf[MTFHIP_BUF_I0], b->d_mi_tb,
				MI_T_INIT, b->buf[MTFHIP_BUF_DF_DI0], b->ctx->stream);
		}
		HIP_TRY(hipMemcpyAsync(b->buf[MTFHIP_BUF_DF_DIT], b->buf[MTFHIP_BUF_DF_DI0], sizeof(double) * b->N * b->B, hipMemcpyDeviceToDevice, b->ctx->stream));
		b->init_grad = true;
		return MTFHIP_OK;
	}
	if (b->desc.am == MTFHIP_AM_NCC) {
		/* NCC::initializeGrad NCC.cc:97-122: gradient vectors start at zero */
		if (!b->init_grad) {
			HIP_TRY(hipMemsetAsync(b->buf[MTFHIP_BUF_DF_DI0], 0, sizeof(double) * b->N * b->B, b->ctx->stream));
			HIP_TRY(hipMemsetAsync(b->buf[MTFHIP_BUF_DF_DIT], 0, sizeof(double) * b->N * b->B, b->ctx->stream));
			b->init_grad = true;
		}
		return MTFHIP_OK;
	}
	if (b->init_grad) return MTFHIP_OK;
	HIP_TRY(hipMemcpyAsync(b->buf[MTFHIP_BUF_DF_DIT], b->buf[MTFHIP_BUF_DF_DI0], sizeof(double) * b->N * b->B, hipMemcpyDeviceToDevice, b->ctx->stream));
	b->init_grad = true;
	return MTFHIP_OK;
}
int mtfhip_am_initialize_hess(mtfhip_batch *b) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_hess: NULL batch");
	return am_supported(b, "initializeHess");
}
static int do_update_similarity(mtfhip_batch *b, int prereq_only);
int mtfhip_am_update_similarity(mtfhip_batch *b, int prereq_only) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_similarity: NULL batch");
	TRY(am_supported(b, "updateSimilarity"));
	if (!b->init_sim) return fail(MTFHIP_ERR_LOGIC, "updateSimilarity before initializeSimilarity");
	if (b->lz.enabled && b->lz.pv) {   /* only behind a deferred updatePixVals: otherwise nothing to fuse with */
		if (b->lz.sim || b->lz.cg) FLUSH(b);
		if (b->lz.pv) {
			b->lz.sim = ++b->lz.seq; b->lz.sim_need_f = !prereq_only;
			return MTFHIP_OK;
		}
	}
	FLUSH(b);
	return do_update_similarity(b, prereq_only);
}
static int do_update_similarity(mtfhip_batch *b, int prereq_only) {
	if (b->desc.am == MTFHIP_AM_NCC) { int rc = ncc_update_similarity(b); b->lz.df0_it_ver = b->lz.ver[MTFHIP_BUF_IT]; return rc; }
	if (b->desc.am == MTFHIP_AM_MI) {
		/* MI::updateSimilarity MI.cc:346-382 */
		TRY(mi_hist_pass(b, 1, 0));
		b->lz.df0_it_ver = b->lz.ver[MTFHIP_BUF_IT];
		if (!prereq_only) TRY(mi_read_f(b));
		return MTFHIP_OK;
	}
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "ssd_residual");
		launch_ssd_residual(b->view(), b->d_partials, nblk, b->ctx->stream);
	}
	stale_clear(b, true, false);
	b->lz.df0_it_ver = b->lz.ver[MTFHIP_BUF_IT];
	if (prereq_only) return MTFHIP_OK;
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) b->th[t].f = -b->h_acc[(size_t)t * ACC_COUNT + ACC_RR] / 2;
	return MTFHIP_OK;
}
static int do_update_curr_grad(mtfhip_batch *b) {
	if (b->desc.am == MTFHIP_AM_NCC) { int rc = ncc_update_grad(b, 1); stale_clear(b, false, true); return rc; }
	if (b->desc.am == MTFHIP_AM_MI) return mi_grad(b, 1);
	if (b->lz.df0_stale) TRY(ensure_one(b, false));
	TimedScope ts(b->ctx, "negate");
	launch_negate(b->buf[MTFHIP_BUF_DF_DI0], b->buf[MTFHIP_BUF_DF_DIT], (size_t)b->N * b->B, b->ctx->stream);
	stale_clear(b, false, true);
	return MTFHIP_OK;
}
int mtfhip_am_update_curr_grad(mtfhip_batch *b) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_curr_grad: NULL batch");
	TRY(am_supported(b, "updateCurrGrad"));
	if (b->lz.enabled) {
		if (b->lz.cg) FLUSH(b);
		b->lz.cg = ++b->lz.seq;
		return MTFHIP_OK;
	}
	FLUSH(b);
	return do_update_curr_grad(b);
}
int mtfhip_am_update_init_grad(mtfhip_batch *b) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_init_grad: NULL batch");
	TRY(am_supported(b, "updateInitGrad"));
	if (b->desc.am == MTFHIP_AM_SSD) return MTFHIP_OK;   /* SSD::updateInitGrad is empty: df_dI0 is updateSimilarity's residual */
	if (b->lz.enabled) {   /* NCC */
		if (b->lz.ig) FLUSH(b);
		b->lz.ig = ++b->lz.seq;
		return MTFHIP_OK;
	}
	FLUSH(b);
	if (b->desc.am == MTFHIP_AM_NCC) { int rc = ncc_update_grad(b, 0); stale_clear(b, true, false); return rc; }
	if (b->desc.am == MTFHIP_AM_MI) return mi_grad(b, 0);
	return MTFHIP_OK;
}
int mtfhip_am_get_similarity(mtfhip_batch *b, double *f) {
	if (!b || !f) return fail(MTFHIP_ERR_INVALID_ARG, "get_similarity: NULL argument");
	TRY(lazy_try_similarity(b));
	FLUSH_AM(b);
	for (int t = 0; t < b->B; ++t) f[t] = b->th[t].f;
	return MTFHIP_OK;
}
int mtfhip_am_get_likelihood(mtfhip_batch *b, double *l) {
	if (!b || !l) return fail(MTFHIP_ERR_INVALID_ARG, "get_likelihood: NULL argument");
	TRY(lazy_try_similarity(b));
	FLUSH_AM(b);
	for (int t = 0; t < b->B; ++t) {
		double f = b->th[t].f;
		if (b->desc.am == MTFHIP_AM_SSD) l[t] = std::exp(-b->desc.likelihood_alpha * std::sqrt(-f / (double)b->N));
		else { double d = (1.0 / f) - 1; l[t] = std::exp(-b->desc.likelihood_alpha * d * d); }
	}
	return MTFHIP_OK;
}

/* ---- deferred fusion: replay, refresh, and the fused execution of a recognised call sequence ---- */
static int pix_grad_common(mtfhip_batch *b, const double *pts, bool warped, bool init);
/* SSD's DF_DI0 = It - I0 and DF_DIT = -DF_DI0 (SSDBase.cc:75-121), NCC's gradient vectors (NCC.cc:163-234), when a fused
 * launch stood in for the calls that write them: derived from the IT (current or shadow) and scalars they belong to */
void stale_clear(mtfhip_batch *b, bool df0, bool dft) {
	mtfhip_batch::Lazy &L = b->lz;
	if (df0) L.df0_stale = L.df0_sh = false;
	if (dft) L.dft_stale = L.dft_sh = false;
	if (!L.df0_sh && !L.dft_sh) L.shadow_valid = false;
}
static void swap_shadow(mtfhip_batch *b) {
	std::swap(b->buf[MTFHIP_BUF_IT], b->d_it_shadow);
	if (b->desc.am == MTFHIP_AM_NCC)
		for (int t = 0; t < b->B; ++t) {
			TargetHost &h = b->th[t]; mtfhip_batch::Lazy::NccSave &v = b->lz.ncc_shadow[t];
			std::swap(h.It_mean, v.It_mean); std::swap(h.a, v.a); std::swap(h.b, v.b); std::swap(h.f, v.f);
		}
	b->ncc_host_newer = true;   /* d_ncc has to follow whichever set of scalars is current */
}
static int ensure_one(mtfhip_batch *b, bool curr) {
	mtfhip_batch::Lazy &L = b->lz;
	if (curr ? !L.dft_stale : !L.df0_stale) return MTFHIP_OK;
	const bool sh = curr ? L.dft_sh : L.df0_sh;
	if (sh) swap_shadow(b);
	int rc = MTFHIP_OK;
	if (b->desc.am == MTFHIP_AM_NCC) {
		rc = ncc_update_grad(b, curr ? 1 : 0);
	} else {
		/* the residual kernel writes It - I0 into the view's DF_DI0; for df_dIt it is pointed at DF_DIT and negated in place */
		BatchView v = b->view();
		if (curr) v.buf[MTFHIP_BUF_DF_DI0] = b->buf[MTFHIP_BUF_DF_DIT];
		{
			TimedScope ts(b->ctx, "ssd_residual");
			launch_ssd_residual(v, b->d_partials, simple_blocks_per_target(b->N), b->ctx->stream);
		}
		if (curr) {
			TimedScope ts(b->ctx, "negate");
			launch_negate(b->buf[MTFHIP_BUF_DF_DIT], b->buf[MTFHIP_BUF_DF_DIT], (size_t)b->N * b->B, b->ctx->stream);
		}
	}
	if (sh) swap_shadow(b);
	if (rc) return rc;
	stale_clear(b, !curr, curr);
	return MTFHIP_OK;
}
int ensure_df(mtfhip_batch *b) {
	TRY(ensure_one(b, false));
	return ensure_one(b, true);
}
/* IT is about to be overwritten by a launch that re-produces df_dI0 (w0) / df_dIt (wt) or not: stale vectors that it does
 * not re-produce keep their IT by a buffer swap instead of being derived now */
int protect_stale(mtfhip_batch *b, bool w0, bool wt) {
	mtfhip_batch::Lazy &L = b->lz;
	const bool cur0 = L.df0_stale && !w0 && !L.df0_sh, curt = L.dft_stale && !wt && !L.dft_sh;
	if (!cur0 && !curt) return MTFHIP_OK;
	if (L.shadow_valid) {   /* an older shadow is still referenced (rare): settle it first */
		if (L.df0_sh) TRY(ensure_one(b, false));
		if (L.dft_sh) TRY(ensure_one(b, true));
	}
	if (!b->d_it_shadow) HIP_TRY(hipMalloc(&b->d_it_shadow, sizeof(double) * b->per_target[MTFHIP_BUF_IT] * b->B));
	L.ncc_shadow.resize(b->B);
	for (int t = 0; t < b->B; ++t) { const TargetHost &h = b->th[t]; L.ncc_shadow[t] = {h.It_mean, h.a, h.b, h.f}; }
	std::swap(b->buf[MTFHIP_BUF_IT], b->d_it_shadow);   /* the launch fills the other buffer; th keeps the current scalars */
	L.shadow_valid = true;
	if (cur0) L.df0_sh = true;
	if (curt) L.dft_sh = true;
	return MTFHIP_OK;
}
/* replays the recorded calls through the un-fused kernels, in the order they were made */
int lazy_flush(mtfhip_batch *b, bool pts) {
	mtfhip_batch::Lazy &L = b->lz;
	if (b->init_mirror_seq && !b->hold_init_pull) TRY(pull_init_mirrors(b));   /* (a fused template initialisation left its small results in a pinned record) */
	/* whatever follows a full flush may launch a kernel that reads the current points; so may the replayed calls */
	if (pts || L.pv || L.gp || L.pg || L.pj) TRY(ensure_pts(b));
	if (!L.any()) return MTFHIP_OK;
	struct Op { long seq; int kind; };
	Op ops[8]; int n = 0;
	if (L.pv) ops[n++] = {L.pv, 0};
	if (L.gp) ops[n++] = {L.gp, 1};
	if (L.pg) ops[n++] = {L.pg, 2};
	if (L.pj) ops[n++] = {L.pj, 3};
	if (L.sim) ops[n++] = {L.sim, 4};
	if (L.cg) ops[n++] = {L.cg, 5};
	if (L.ig) ops[n++] = {L.ig, 6};
	if (L.jm) ops[n++] = {L.jm, 7};
	std::sort(ops, ops + n, [](const Op &x, const Op &y) { return x.seq < y.seq; });
	const int pg_kind = L.pg_kind, pj_variant = L.pj_variant; const bool need_f = L.sim_need_f;
	L.pv = L.gp = L.pg = L.pj = L.sim = L.cg = L.ig = L.jm = 0;   /* cleared first: the executors below may flush */
	for (int i = 0; i < n; ++i) {
		switch (ops[i].kind) {
		case 0: TRY(do_update_pix_vals(b, nullptr)); break;
		case 1: TRY(do_update_grad_pts(b, b->desc.grad_eps)); break;
		case 2: TRY(pix_grad_common(b, nullptr, pg_kind == 2, false)); break;
		case 3: TRY(do_cmpt_pix_jacobian(b, pj_variant, MTFHIP_BUF_DIT_DX, MTFHIP_BUF_JT)); break;
		case 4: TRY(do_update_similarity(b, need_f ? 0 : 1)); break;
		case 5: TRY(do_update_curr_grad(b)); break;
		case 6:          /* SSD::updateInitGrad is empty (SSDBase.h) */
			if (b->desc.am == MTFHIP_AM_NCC) { TRY(ncc_update_grad(b, 0)); stale_clear(b, true, false); }
			else if (b->desc.am == MTFHIP_AM_MI) TRY(mi_grad(b, 0));
			break;
		default: TRY(do_mean_jacobian(b)); break;
		}
	}
	return MTFHIP_OK;
}
int fused_args(const mtfhip_batch *b, const mtfhip_sm_desc *sm, FusedArgs &fa);
int ncc_template_moments(mtfhip_batch *b);
int ncc_lazy_outputs(mtfhip_batch *b, int trig, int j_a, bool hess_mean, double *g);
/* `*done` = 1 when the pending calls plus this Jacobian request were served by ONE fused launch (g filled with the AM's
 * raw Jacobian), 0 when the caller has to flush and take the un-fused route.
 *   FCLK  NT/FCLK.cc:171-358: updatePixVals, updateSimilarity, updateCurrGrad, pixel gradient + pixel Jacobian, cmptCurrJacobian(Jt)
 *   ESM   NT/ESM.cc:170-296: ... updateInitGrad, cmptDifferenceOfJacobians(J0, Jt)   (jac_type Original: cmptCurrJacobian(Jm))
 *   ICLK  NT/ICLK.cc:160-299: updatePixVals, updateSimilarity, updateInitGrad, cmptInitJacobian(J0) */
/* Deferred fusion, MI: the recognised sequence is served by the fused MI passes (api_fused.hip: mi_iterate describes
 * them) -- the fused LK kernel materialising It / dIt_dx / Jt, one histogram pass (with the self histogram when the search
 * method has been asking for self Hessians), one table kernel, one gradient + Jacobian-product pass that also writes the
 * gradient vectors the recorded update*Grad calls would have written.  `sm` carries what lazy_try_fused classified. */
static int mi_lazy_fused(mtfhip_batch *b, int trig, int j_a, const mtfhip_sm_desc &sm, bool replay, double *g, int *done) {
	mtfhip_batch::Lazy &L = b->lz;
	const bool iclk = sm.sm == MTFHIP_SM_ICLK;
	if (trig == LAZY_CURR_JAC && j_a != MTFHIP_BUF_JT) return MTFHIP_OK;   /* Original Jacobian: df_dIt . Jm is not accumulated */
	if (sm.sm != MTFHIP_SM_FCLK && !L.ig) return MTFHIP_OK;                /* df_dI0 comes from updateInitGrad */
	const int nb = b->desc.mi_n_bins, nblk = mi_blocks(b), S = b->S;
	hipStream_t st = b->ctx->stream;
	if (replay || !iclk) {   /* (ICLK on a current IT needs nothing from the image) */
		mtfhip_sm_desc s0 = sm;
		s0.sm = iclk ? MTFHIP_SM_ICLK : MTFHIP_SM_FCLK; s0.hess_type = iclk ? 0 : 1; s0.materialize = 1;
		FusedArgs fa;
		TRY(fused_args(b, &s0, fa));
		{
			TimedScope ts(b->ctx, "fused_lk");
			launch_fused_ssd(fused_view(b, fa), b->ctx->img, fa, b->d_partials, fused_blocks_per_target(b->N, b->B), st);
		}
		const bool self_was = L.mi_self_it == L.ver[MTFHIP_BUF_IT];
		touch(b, MTFHIP_BUF_IT);
		if (!replay && self_was) L.mi_self_it = L.ver[MTFHIP_BUF_IT];   /* same bits */
		b->it_valid = true;
		L.it_epoch = L.epoch;
		if (!iclk) { touch(b, MTFHIP_BUF_DIT_DX); touch(b, MTFHIP_BUF_JT); b->dit_valid = b->jt_valid = true; }
	}
	const double *It = b->buf[MTFHIP_BUF_IT], *I0 = b->buf[MTFHIP_BUF_I0];
	{
		TimedScope ts(b->ctx, "mi_hist");
		if (replay) {
			const bool self = L.mi_want_self && !L.no_cache;
			if (self) launch_mi_hist_self(b->view(), nb, b->mi_hist_norm, It, I0, b->d_mi_part, nblk, b->mi_row_len, st);
			else launch_mi_hist(b->view(), nb, b->mi_hist_norm, It, I0, b->d_mi_part, nblk, b->mi_row_len, st);
			launch_mi_tables_iter(b->view(), nb, b->desc.mi_pre_seed, b->mi_hist_norm, self ? 1 : 0, b->d_mi_part, nblk, b->mi_row_len, b->d_mi_tb,
				b->d_mi_f, st);
			L.mi_self_it = self ? L.ver[MTFHIP_BUF_IT] : -1;
		} else {   /* updateSimilarity already ran for this IT: only the factor tables update*Grad would refresh */
			launch_mi_factor(b->view(), nb, 1, b->d_mi_tb, st);
			launch_mi_factor(b->view(), nb, 0, b->d_mi_tb, st);
		}
	}
	L.df0_it_ver = L.ver[MTFHIP_BUF_IT];
	double *d_g = b->d_mi_H + 64 * (size_t)b->B;
	{
		TimedScope ts(b->ctx, "mi_grad");
		const int ng = std::min(simple_blocks_per_target(b->N), 64);
		launch_mi_grad_gemv(b->view(), nb, b->mi_hist_norm, It, I0, b->d_mi_tb, iclk ? nullptr : b->buf[MTFHIP_BUF_JT],
			sm.sm == MTFHIP_SM_FCLK ? nullptr : b->buf[MTFHIP_BUF_J0], mi_j0_rebuild(b), L.cg ? b->buf[MTFHIP_BUF_DF_DIT] : nullptr,
			L.ig ? b->buf[MTFHIP_BUF_DF_DI0] : nullptr, b->d_partials, ng, st);
		launch_finish_rows(b->d_partials, ng, 16, d_g, b->B, st);
	}
	const bool want_mean = L.jm != 0;
	L.pv = L.gp = L.pg = L.pj = L.sim = L.cg = L.ig = L.jm = 0;
	if (want_mean) TRY(do_mean_jacobian(b));
	std::vector<double> out((size_t)17 * b->B);
	HIP_TRY(hipMemcpyAsync(out.data(), d_g, sizeof(double) * 16 * b->B, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(out.data() + (size_t)16 * b->B, b->d_mi_f, sizeof(double) * b->B, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	for (int t = 0; t < b->B; ++t) {
		const double *gs = &out[(size_t)16 * t];
		if (replay) b->th[t].f = out[(size_t)16 * b->B + t];
		for (int s = 0; s < S; ++s)
			g[(size_t)t * S + s] = trig == LAZY_INIT_JAC ? gs[8 + s] : (trig == LAZY_CURR_JAC ? gs[s] : gs[s] - gs[8 + s]);
	}
	*done = 1;
	return MTFHIP_OK;
}
static int lazy_try_fused(mtfhip_batch *b, int trig, int j_a, int j_b, double *g, int *done) {
	*done = 0;
	mtfhip_batch::Lazy &L = b->lz;
	if (!L.enabled) return MTFHIP_OK;
	/* either updatePixVals + updateSimilarity are part of the pending set, or they already ran for this very warp and image */
	const bool replay = L.pv && L.sim && L.pv < L.sim;
	const bool current = !L.pv && !L.sim && L.it_epoch == L.epoch && L.df0_it_ver == L.ver[MTFHIP_BUF_IT];
	if (!replay && !current) return MTFHIP_OK;
	if (!b->init_pix_vals || !b->init_sim || !b->have_corners || !b->ctx->img.data || b->ctx->img.channels != 1) return MTFHIP_OK;
	mtfhip_sm_desc sm;
	std::memset(&sm, 0, sizeof(sm));
	sm.materialize = 1; sm.max_iters = 1; sm.chained_warp = 1;
	bool pixel_chain = false;
	if (L.pg || L.pj || L.gp) {
		if (!L.pg || !L.pj || L.pg > L.pj) return MTFHIP_OK;
		if (L.pg_kind == 1) { if (L.gp || L.pj_variant != MTFHIP_JAC_WARPED) return MTFHIP_OK; }
		else { if (!L.gp || L.gp > L.pg || L.pj_variant != MTFHIP_JAC_INIT) return MTFHIP_OK; sm.chained_warp = 0; }
		pixel_chain = true;
	}
	if (L.jm && (!pixel_chain || L.jm < L.pj)) return MTFHIP_OK;
	double gscale = 1.0;
	if (trig == LAZY_INIT_JAC) {
		if (pixel_chain || L.jm || j_a != MTFHIP_BUF_J0 || !b->buf[MTFHIP_BUF_J0]) return MTFHIP_OK;
		sm.sm = MTFHIP_SM_ICLK; sm.hess_type = 0;
	} else {
		if (!pixel_chain || !L.cg || L.cg < L.sim) return MTFHIP_OK;   /* (L.sim is 0 when it already ran) */
		if (trig == LAZY_DIFF_JAC) {
			if (j_a != MTFHIP_BUF_J0 || j_b != MTFHIP_BUF_JT || !b->buf[MTFHIP_BUF_J0]) return MTFHIP_OK;
			sm.sm = MTFHIP_SM_ESM; sm.hess_type = L.jm ? 3 : 5;
		} else if (j_a == MTFHIP_BUF_JT) {
			sm.sm = MTFHIP_SM_FCLK; sm.hess_type = 2;
		} else if (j_a == MTFHIP_BUF_JM && L.jm && b->buf[MTFHIP_BUF_J0]) {
			sm.sm = MTFHIP_SM_ESM; sm.hess_type = 3; gscale = 0.5;   /* df_dIt . (J0 + Jt) / 2, the halving is exact */
		} else return MTFHIP_OK;
	}
	if (b->desc.am == MTFHIP_AM_MI) return mi_lazy_fused(b, trig, j_a, sm, replay, g, done);
	const bool ncc = b->desc.am == MTFHIP_AM_NCC;
	if (ncc && trig == LAZY_INIT_JAC && !L.ig) return MTFHIP_OK;   /* NCC's df_dI0 comes from updateInitGrad */
	/* gradients a previous fused launch skipped and this one will not re-produce keep their IT (when IT is current the
	 * launch rewrites the same bits, nothing to protect) */
	if (replay) TRY(protect_stale(b, ncc ? L.ig != 0 : true, L.cg != 0));
	if (current && trig == LAZY_INIT_JAC && !L.no_cache) {
		/* the lean launch behind getSimilarity() already accumulated this Jacobian for the same IT and J0: no launch */
		bool served = false;
		if (!ncc && L.sim_g_it == L.ver[MTFHIP_BUF_IT] && L.sim_g_j0 == L.ver[MTFHIP_BUF_J0] && !L.sim_g.empty()) {
			for (int t = 0; t < b->B; ++t) std::memcpy(g + (size_t)t * b->S, &L.sim_g[(size_t)8 * t], sizeof(double) * b->S);
			served = true;
		} else if (ncc && !L.ncc_M.empty() && L.ncc_M_it == L.ver[MTFHIP_BUF_IT] && L.ncc_tm_ver == L.ver[MTFHIP_BUF_J0]) {
			std::memcpy(b->h_acc, L.ncc_M.data(), sizeof(double) * L.ncc_M.size());
			const long jt = L.ncc_M_jt, jm = L.ncc_M_jm; const bool mean = L.ncc_M_mean;
			TRY(ncc_lazy_outputs(b, trig, j_a, mean, g));
			L.ncc_M_jt = jt; L.ncc_M_jm = jm;   /* the rows are unchanged: what they hold about Jt / Jm stays as it was */
			served = true;
		}
		if (served) {
			if (ncc && L.ig) { L.df0_stale = true; L.df0_sh = false; if (!L.dft_sh) L.shadow_valid = false; }
			L.ig = 0;
			*done = 1;
			return MTFHIP_OK;
		}
	}
	if (ncc && sm.sm != MTFHIP_SM_FCLK && L.ncc_tm_ver != L.ver[MTFHIP_BUF_J0]) {   /* moments of the template's Jacobian */
		TRY(ncc_template_moments(b));
		L.ncc_tm_ver = L.ver[MTFHIP_BUF_J0];
	}
	FusedArgs fa;
	TRY(fused_args(b, &sm, fa));
	const int nblk = fused_blocks_per_target(b->N, b->B);
	{
		TimedScope ts(b->ctx, "fused_lk");
		launch_fused_ssd(fused_view(b, fa), b->ctx->img, fa, b->d_partials, nblk, b->ctx->stream);
	}
	touch(b, MTFHIP_BUF_IT);
	b->it_valid = true;
	L.it_epoch = L.epoch;
	if (fa.mode != 2) { touch(b, MTFHIP_BUF_DIT_DX); touch(b, MTFHIP_BUF_JT); b->dit_valid = b->jt_valid = true; }
	const bool want_mean = L.jm != 0;
	/* the N-sized gradient vectors the consumed calls would have written: SSD's df_dI0 is updateSimilarity's residual,
	 * NCC's comes from updateInitGrad; df_dIt from updateCurrGrad in both */
	if (ncc ? L.ig != 0 : replay) { L.df0_stale = true; L.df0_sh = false; }
	L.df0_it_ver = L.ver[MTFHIP_BUF_IT];          /* (when IT was current the launch rewrote the same bits) */
	if (L.cg) { L.dft_stale = true; L.dft_sh = false; }
	if (!L.df0_sh && !L.dft_sh) L.shadow_valid = false;
	L.pv = L.gp = L.pg = L.pj = L.sim = L.cg = L.ig = L.jm = 0;
	if (want_mean) TRY(do_mean_jacobian(b));
	if (ncc) {
		TRY(read_rows(b, nblk, NCC_ACC_COUNT));
		TRY(ncc_lazy_outputs(b, trig, j_a, fa.hess_mean != 0, g));
		*done = 1;
		return MTFHIP_OK;
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) {
		const double *acc = b->h_acc + (size_t)t * ACC_COUNT;
		b->th[t].f = -acc[ACC_RR] / 2;
		for (int s = 0; s < b->S; ++s) g[(size_t)t * b->S + s] = gscale * acc[ACC_G + s];
	}
	if (fa.mode != 2 && !L.no_cache) {   /* the Gram matrix the launch accumulated: Jt, or Jm with hess_mean */
		L.gram_buf = fa.hess_mean ? MTFHIP_BUF_JM : MTFHIP_BUF_JT;
		L.gram_ver = L.ver[L.gram_buf];
		L.gram.resize((size_t)36 * b->B);
		for (int t = 0; t < b->B; ++t) std::memcpy(&L.gram[(size_t)36 * t], b->h_acc + (size_t)t * ACC_COUNT + ACC_H, sizeof(double) * 36);
	}
	*done = 1;
	return MTFHIP_OK;
}

int gemv_to_host(mtfhip_batch *b, const double *v1, int j1, const double *v2, int j2, int sum_mode, double *g, int diff) {
	int nblk = simple_blocks_per_target(b->N);
	{
		TimedScope ts(b->ctx, "gemv");
		launch_gemv(b->view(), v1, b->buf[j1], v2, j2 >= 0 ? b->buf[j2] : nullptr, sum_mode, b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t)
		for (int s = 0; s < b->S; ++s) {
			double v = b->h_acc[(size_t)t * ACC_COUNT + ACC_G + s];
			if (diff) v -= b->h_acc[(size_t)t * ACC_COUNT + ACC_G2 + s];
			g[(size_t)t * b->S + s] = v;
		}
	return MTFHIP_OK;
}
static int j_ready(mtfhip_batch *b, int id, const char *fn) {
	if (!j_buf_ok(id)) return fail(MTFHIP_ERR_INVALID_ARG, "%s: Jacobian buffer id %d is not J0/JT/JM", fn, id);
	if (!b->buf[id]) return fail(MTFHIP_ERR_LOGIC, "%s: Jacobian buffer %d was never produced", fn, id);
	if (id == MTFHIP_BUF_JT && !b->jt_valid) return fail(MTFHIP_ERR_LOGIC, "%s: JT is not materialised", fn);
	return MTFHIP_OK;
}
int mtfhip_am_cmpt_init_jacobian(mtfhip_batch *b, int j0_buf, double *g) {
	if (!b || !g) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_init_jacobian: NULL argument");
	TRY(am_supported(b, "cmptInitJacobian"));
	TRY(j_ready(b, j0_buf, "cmptInitJacobian"));
	{ int done; TRY(lazy_try_fused(b, LAZY_INIT_JAC, j0_buf, -1, g, &done)); if (done) return MTFHIP_OK; }
	FLUSH_AM(b);
	TRY(ensure_df(b));
	return gemv_to_host(b, b->buf[MTFHIP_BUF_DF_DI0], j0_buf, nullptr, -1, 0, g, 0);
}
int mtfhip_am_cmpt_curr_jacobian(mtfhip_batch *b, int jt_buf, double *g) {
	if (!b || !g) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_curr_jacobian: NULL argument");
	TRY(am_supported(b, "cmptCurrJacobian"));
	TRY(j_ready(b, jt_buf, "cmptCurrJacobian"));
	{ int done; TRY(lazy_try_fused(b, LAZY_CURR_JAC, jt_buf, -1, g, &done)); if (done) return MTFHIP_OK; }
	FLUSH_AM(b);
	TRY(ensure_df(b));
	return gemv_to_host(b, b->buf[MTFHIP_BUF_DF_DIT], jt_buf, nullptr, -1, 0, g, 0);
}
int mtfhip_am_cmpt_difference_of_jacobians(mtfhip_batch *b, int j0_buf, int jt_buf, double *g) {
	if (!b || !g) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_difference_of_jacobians: NULL argument");
	TRY(am_supported(b, "cmptDifferenceOfJacobians"));
	TRY(j_ready(b, j0_buf, "cmptDifferenceOfJacobians"));
	TRY(j_ready(b, jt_buf, "cmptDifferenceOfJacobians"));
	{ int done; TRY(lazy_try_fused(b, LAZY_DIFF_JAC, j0_buf, jt_buf, g, &done)); if (done) return MTFHIP_OK; }
	FLUSH_AM(b);
	TRY(ensure_df(b));
	if (b->desc.am != MTFHIP_AM_SSD) /* (df_dIt * dIt_dp) - (df_dI0 * dI0_dp), NCC.cc:268-280, AppearanceModel.h:161-164 */
		return gemv_to_host(b, b->buf[MTFHIP_BUF_DF_DIT], jt_buf, b->buf[MTFHIP_BUF_DF_DI0], j0_buf, 0, g, 1);
	/* SSD: df_dIt * (dI0_dpssm + dIt_dpssm), SSDBase.cc:186 */
	return gemv_to_host(b, b->buf[MTFHIP_BUF_DF_DIT], jt_buf, nullptr, j0_buf, 1, g, 0);
}
/* J^T J of a pixel Jacobian, from the host-side copy when the buffer has not been written since that copy was made
 * (the fused launch of this iteration accumulated it; the template's J0 only changes with the template) */
static int gram_to_host(mtfhip_batch *b, int j_buf, double *H, double scale, bool accumulate) {
	mtfhip_batch::Lazy &L = b->lz;
	const double *src = nullptr;
	if (!L.no_cache) {
		if (j_buf == L.gram_buf && L.gram_ver == L.ver[j_buf] && !L.gram.empty()) src = L.gram.data();
		else if (j_buf == MTFHIP_BUF_J0 && L.gram0_ver == L.ver[j_buf] && !L.gram0.empty()) src = L.gram0.data();
	}
	if (!src) {
		int nblk = simple_blocks_per_target(b->N);
		{
			TimedScope ts(b->ctx, "gram");
			launch_gram(b->view(), b->buf[j_buf], b->d_partials, nblk, b->ctx->stream);
		}
		TRY(read_acc(b, nblk));
		std::vector<double> &dst = j_buf == MTFHIP_BUF_J0 ? L.gram0 : L.gram;
		dst.resize((size_t)36 * b->B);
		for (int t = 0; t < b->B; ++t) std::memcpy(&dst[(size_t)36 * t], b->h_acc + (size_t)t * ACC_COUNT + ACC_H, sizeof(double) * 36);
		if (j_buf == MTFHIP_BUF_J0) L.gram0_ver = L.ver[j_buf];
		else { L.gram_buf = j_buf; L.gram_ver = L.ver[j_buf]; }
		src = dst.data();
	}
	const int S = b->S;
	for (int t = 0; t < b->B; ++t) {
		int k = 0;
		for (int a = 0; a < 8; ++a)
			for (int c = a; c < 8; ++c) {
				if (a < S && c < S) {
					double v = scale * src[(size_t)36 * t + k];
					double *Ht = H + (size_t)t * S * S;
					if (accumulate) { Ht[c * S + a] += v; if (a != c) Ht[a * S + c] += v; }
					else { Ht[c * S + a] = v; Ht[a * S + c] = v; }
				}
				++k;
			}
	}
	return MTFHIP_OK;
}
int mtfhip_am_cmpt_init_hessian(mtfhip_batch *b, int j0_buf, double *H) {
	FLUSH_AM(b);
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_init_hessian: NULL argument");
	TRY(am_supported(b, "cmptInitHessian"));
	TRY(j_ready(b, j0_buf, "cmptInitHessian"));
	if (b->desc.am == MTFHIP_AM_NCC) return ncc_hessian(b, j0_buf, 0, H);
	if (b->desc.am == MTFHIP_AM_MI) return mi_hessian(b, j0_buf, 0, H);
	return gram_to_host(b, j0_buf, H, -1.0, false);
}
int mtfhip_am_cmpt_curr_hessian(mtfhip_batch *b, int jt_buf, double *H) {
	FLUSH_AM(b);
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_curr_hessian: NULL argument");
	TRY(am_supported(b, "cmptCurrHessian"));
	TRY(j_ready(b, jt_buf, "cmptCurrHessian"));
	if (b->desc.am == MTFHIP_AM_NCC) return ncc_hessian(b, jt_buf, 1, H);
	if (b->desc.am == MTFHIP_AM_MI) return mi_hessian(b, jt_buf, 1, H);
	return gram_to_host(b, jt_buf, H, -1.0, false);
}
int mtfhip_am_cmpt_self_hessian(mtfhip_batch *b, int jt_buf, double *H) {
	FLUSH_AM(b);
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_self_hessian: NULL argument");
	TRY(am_supported(b, "cmptSelfHessian"));
	TRY(j_ready(b, jt_buf, "cmptSelfHessian"));
	if (b->desc.am == MTFHIP_AM_NCC) return ncc_hessian(b, jt_buf, 2, H);
	if (b->desc.am == MTFHIP_AM_MI) return mi_hessian(b, jt_buf, 2, H);
	return gram_to_host(b, jt_buf, H, -1.0, false);
}
int mtfhip_am_cmpt_sum_of_hessians(mtfhip_batch *b, int j0_buf, int jt_buf, double *H) {
	FLUSH_AM(b);
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_sum_of_hessians: NULL argument");
	TRY(am_supported(b, "cmptSumOfHessians"));
	TRY(j_ready(b, j0_buf, "cmptSumOfHessians"));
	TRY(j_ready(b, jt_buf, "cmptSumOfHessians"));
	if (b->desc.am != MTFHIP_AM_SSD) {
		/* generic AppearanceModel::cmptSumOfHessians AppearanceModel.h:196-208 */
		std::vector<double> H0((size_t)b->B * b->S * b->S);
		if (b->desc.am == MTFHIP_AM_NCC) { TRY(ncc_hessian(b, j0_buf, 0, H0.data())); TRY(ncc_hessian(b, jt_buf, 1, H)); }
		else { TRY(mi_hessian(b, j0_buf, 0, H0.data())); TRY(mi_hessian(b, jt_buf, 1, H)); }
		for (size_t i = 0; i < H0.size(); ++i) H[i] += H0[i];
		return MTFHIP_OK;
	}
	TRY(gram_to_host(b, j0_buf, H, -1.0, false));
	return gram_to_host(b, jt_buf, H, -1.0, true);
}
static int do_mean_jacobian(mtfhip_batch *b) {
	TRY(ensure_buf(b, MTFHIP_BUF_JM));
	TimedScope ts(b->ctx, "mean_jacobian");
	launch_mean_jacobian(b->view(), b->ctx->stream);
	touch(b, MTFHIP_BUF_JM);
	return MTFHIP_OK;
}
int mtfhip_sm_mean_jacobian(mtfhip_batch *b) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "mean_jacobian: NULL batch");
	TRY(j_ready(b, MTFHIP_BUF_J0, "mean_jacobian"));
	TRY(j_ready(b, MTFHIP_BUF_JT, "mean_jacobian"));
	if (b->lz.enabled && b->lz.pj) {
		if (b->lz.jm) FLUSH(b);
		if (b->lz.pj) { TRY(ensure_buf(b, MTFHIP_BUF_JM)); b->lz.jm = ++b->lz.seq; return MTFHIP_OK; }
	}
	FLUSH(b);
	return do_mean_jacobian(b);
}

/* ------------------------------------------------------------------ second order (sec_ord_hess) */
static int hess_buf_ok(int id) { return id == MTFHIP_BUF_D2I0_DX2 || id == MTFHIP_BUF_D2IT_DX2; }
static int d2_buf_ok(int id) { return id == MTFHIP_BUF_D2I0_DP2 || id == MTFHIP_BUF_D2IT_DP2 || id == MTFHIP_BUF_D2IM_DP2; }

int mtfhip_ssm_update_hess_pts(mtfhip_batch *b, double hess_eps) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_hess_pts: NULL batch");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "update_hess_pts before set_corners");
	TRY(ensure_buf(b, MTFHIP_BUF_HESS_PTS));
	TimedScope ts(b->ctx, "hess_pts");
	launch_hess_pts(b->view(), hess_eps, b->ctx->stream);
	return MTFHIP_OK;
}

/* ImageBase::initializePixHess / updatePixHess, both overloads (AM/src/ImageBase.cc:174-240, 316-338, 364-386) */
static int pix_hess_common(mtfhip_batch *b, const double *pts, const double *hess_pts, bool warped, bool init) {
	TRY(need_image(b));
	TRY(ensure_buf(b, MTFHIP_BUF_D2I0_DX2));
	TRY(ensure_buf(b, MTFHIP_BUF_D2IT_DX2));
	const size_t N = b->N, NP = b->NP;
	const double *dp, *dh = nullptr;
	TRY(resolve_pts(b, pts, MTFHIP_BUF_CURR_PTS, 2 * NP, &dp));
	if (warped) {
		if (!hess_pts) {
			if (!b->buf[MTFHIP_BUF_HESS_PTS]) return fail(MTFHIP_ERR_LOGIC, "pix_hess: device hess_pts not available (call update_hess_pts)");
			dh = b->buf[MTFHIP_BUF_HESS_PTS];
		} else {
			double *stage = b->d_scratch_pts + 2 * NP * b->B;
			HIP_TRY(hipMemcpyAsync(stage, hess_pts, sizeof(double) * 16 * NP * b->B, hipMemcpyHostToDevice, b->ctx->stream));
			dh = stage;
		}
	}
	double *dst = b->buf[init ? MTFHIP_BUF_D2I0_DX2 : MTFHIP_BUF_D2IT_DX2];
	{
		TimedScope ts(b->ctx, warped ? "warped_img_hess" : "img_hess");
		if (warped) launch_warped_img_hess(b->view(), b->ctx->img, dp, dh, dst, b->hess_eps, b->norm_mult, b->ctx->stream);
		else launch_img_hess(b->view(), b->ctx->img, dp, dst, b->hess_eps, b->norm_mult, b->ctx->stream);
	}
	if (init && !b->init_pix_hess) {
		HIP_TRY(hipMemcpyAsync(b->buf[MTFHIP_BUF_D2IT_DX2], b->buf[MTFHIP_BUF_D2I0_DX2], sizeof(double) * 4 * N * b->B, hipMemcpyDeviceToDevice, b->ctx->stream));
		b->init_pix_hess = true;
	}
	return MTFHIP_OK;
}
int mtfhip_am_initialize_pix_hess(mtfhip_batch *b, const double *pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_pix_hess: NULL batch");
	return pix_hess_common(b, pts, nullptr, false, true);
}
int mtfhip_am_update_pix_hess(mtfhip_batch *b, const double *pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_pix_hess: NULL batch");
	return pix_hess_common(b, pts, nullptr, false, false);
}
int mtfhip_am_initialize_pix_hess_warped(mtfhip_batch *b, const double *pts, const double *hess_pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "initialize_pix_hess_warped: NULL batch");
	return pix_hess_common(b, pts, hess_pts, true, true);
}
int mtfhip_am_update_pix_hess_warped(mtfhip_batch *b, const double *pts, const double *hess_pts) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_pix_hess_warped: NULL batch");
	return pix_hess_common(b, pts, hess_pts, true, false);
}

int mtfhip_ssm_cmpt_pix_hessian(mtfhip_batch *b, int variant, int hess_buf, int grad_buf, int dst_buf) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_pix_hessian: NULL batch");
	if (variant < MTFHIP_JAC_INIT || variant > MTFHIP_JAC_APPROX) return fail(MTFHIP_ERR_INVALID_ARG, "unknown pixel Hessian variant %d", variant);
	if (!hess_buf_ok(hess_buf)) return fail(MTFHIP_ERR_INVALID_ARG, "hess_buf must be D2I0_DX2 or D2IT_DX2");
	if (grad_buf != MTFHIP_BUF_DI0_DX && grad_buf != MTFHIP_BUF_DIT_DX) return fail(MTFHIP_ERR_INVALID_ARG, "grad_buf must be DI0_DX or DIT_DX");
	if (!d2_buf_ok(dst_buf)) return fail(MTFHIP_ERR_INVALID_ARG, "dst_buf must be D2I0_DP2, D2IT_DP2 or D2IM_DP2");
	/* Affine implements Init and Warped only (SSM/include/mtf/SSM/Affine.h); the others are ssm_func_not_implemeted
	 * (StateSpaceModel.h:186-197) */
	if (b->desc.ssm == MTFHIP_SSM_AFFINE && (variant == MTFHIP_JAC_PIX || variant == MTFHIP_JAC_APPROX))
		return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "%s :: function not implemented yet", variant == MTFHIP_JAC_PIX ? "cmptPixHessian" : "cmptApproxPixHessian");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "cmpt_pix_hessian before set_corners");
	if (!b->buf[hess_buf]) return fail(MTFHIP_ERR_LOGIC, "cmpt_pix_hessian: image Hessian %d was never computed", hess_buf);
	TRY(ensure_buf(b, dst_buf));
	TimedScope ts(b->ctx, "pix_hessian");
	launch_pix_hessian(b->view(), variant, b->buf[hess_buf], b->buf[grad_buf], b->buf[dst_buf], b->ctx->stream);
	return MTFHIP_OK;
}

int mtfhip_sm_mean_pix_hessian(mtfhip_batch *b) {
	FLUSH_AM(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "mean_pix_hessian: NULL batch");
	if (!b->buf[MTFHIP_BUF_D2I0_DP2] || !b->buf[MTFHIP_BUF_D2IT_DP2]) return fail(MTFHIP_ERR_LOGIC, "mean_pix_hessian: init / curr pixel Hessians not computed");
	TRY(ensure_buf(b, MTFHIP_BUF_D2IM_DP2));
	TimedScope ts(b->ctx, "mean_pix_hessian");
	launch_mean_planes(b->buf[MTFHIP_BUF_D2I0_DP2], b->buf[MTFHIP_BUF_D2IT_DP2], b->buf[MTFHIP_BUF_D2IM_DP2],
		(size_t)b->B * b->N * b->S * b->S, b->ctx->stream);
	return MTFHIP_OK;
}

/* H[t] += sum_p w[p] * (d2a[:, p] (+ d2b[:, p])) */
static int add_second_order(mtfhip_batch *b, int d2a, int d2b, const double *dev_w, double *H) {
	if (!d2_buf_ok(d2a) || (d2b >= 0 && !d2_buf_ok(d2b))) return fail(MTFHIP_ERR_INVALID_ARG, "pixel-Hessian buffer must be D2I0_DP2, D2IT_DP2 or D2IM_DP2");
	if (!b->buf[d2a] || (d2b >= 0 && !b->buf[d2b])) return fail(MTFHIP_ERR_LOGIC, "second-order Hessian: pixel Hessian buffer was never computed");
	const int nblk = simple_blocks_per_target(b->N), S = b->S;
	if (!b->d_d2_part) {
		HIP_TRY(hipMalloc(&b->d_d2_part, sizeof(double) * 64 * (size_t)nblk * b->B));
		HIP_TRY(hipMalloc(&b->d_d2_out, sizeof(double) * 64 * (size_t)b->B));
	}
	{
		TimedScope ts(b->ctx, "pix_hess_weighted_sum");
		launch_weighted_plane_sum(b->view(), b->buf[d2a], d2b >= 0 ? b->buf[d2b] : nullptr, dev_w, b->d_d2_part, nblk, b->d_d2_out, b->ctx->stream);
	}
	std::vector<double> h((size_t)S * S * b->B);
	HIP_TRY(hipMemcpyAsync(h.data(), b->d_d2_out, sizeof(double) * h.size(), hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	for (size_t i = 0; i < h.size(); ++i) H[i] += h[i];
	return MTFHIP_OK;
}
/* SSDBase.cc:313-343 ; NCC.cc:391-400 ; MI.cc:659-673 */
int mtfhip_am_cmpt_init_hessian2(mtfhip_batch *b, int j0_buf, int d2_buf, double *H) {
	FLUSH_AM(b);
	if (b) TRY(ensure_df(b));   /* the second-order terms are weighted by df_dI */
	TRY(mtfhip_am_cmpt_init_hessian(b, j0_buf, H));
	return add_second_order(b, d2_buf, -1, b->buf[MTFHIP_BUF_DF_DI0], H);
}
/* SSDBase.cc:345-375 ; NCC.cc:401-410 ; MI.cc:680-694 */
int mtfhip_am_cmpt_curr_hessian2(mtfhip_batch *b, int jt_buf, int d2_buf, double *H) {
	FLUSH_AM(b);
	if (b) TRY(ensure_df(b));   /* the second-order terms are weighted by df_dI */
	TRY(mtfhip_am_cmpt_curr_hessian(b, jt_buf, H));
	return add_second_order(b, d2_buf, -1, b->buf[MTFHIP_BUF_DF_DIT], H);
}
/* SSD: first order only (SSDBase.h:95-98) ; NCC: am_func_not_implemeted (AppearanceModel.h:188-191) ; MI.cc:696-733 */
int mtfhip_am_cmpt_self_hessian2(mtfhip_batch *b, int jt_buf, int d2_buf, double *H) {
	FLUSH_AM(b);
	if (b) TRY(ensure_df(b));   /* the second-order terms are weighted by df_dI */
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_self_hessian (second order): NULL argument");
	if (b->desc.am == MTFHIP_AM_NCC) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "ncc :: cmptSelfHessian(second order) :: function not implemented yet");
	TRY(mtfhip_am_cmpt_self_hessian(b, jt_buf, H));
	if (b->desc.am == MTFHIP_AM_SSD) return MTFHIP_OK;
	/* MI: weight = sum_r curr_hist_grad(r) * sum_t curr_hist_mat(t) * self_grad_factor(r, t)  (MI.cc:710-723);
	 * the self table was filled by the first-order call above */
	if (!b->d_d2_w) HIP_TRY(hipMalloc(&b->d_d2_w, sizeof(double) * (size_t)b->N * b->B));
	launch_mi_grad(b->view(), b->desc.mi_n_bins, b->mi_hist_norm, b->buf[MTFHIP_BUF_IT], b->buf[MTFHIP_BUF_IT], b->d_mi_tb, MI_T_SELF,
		b->d_d2_w, b->ctx->stream);
	return add_second_order(b, d2_buf, -1, b->d_d2_w, H);
}
/* SSDBase.cc:377-415 (both pixel Hessians weighted by df_dI0) ; NCC / MI: generic AppearanceModel.h:209-219 */
int mtfhip_am_cmpt_sum_of_hessians2(mtfhip_batch *b, int j0_buf, int jt_buf, int d20_buf, int d2t_buf, double *H) {
	FLUSH_AM(b);
	if (b) TRY(ensure_df(b));   /* the second-order terms are weighted by df_dI */
	if (!b || !H) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_sum_of_hessians (second order): NULL argument");
	if (b->desc.am == MTFHIP_AM_SSD) {
		TRY(mtfhip_am_cmpt_sum_of_hessians(b, j0_buf, jt_buf, H));
		return add_second_order(b, d20_buf, d2t_buf, b->buf[MTFHIP_BUF_DF_DI0], H);
	}
	std::vector<double> H0((size_t)b->B * b->S * b->S);
	TRY(mtfhip_am_cmpt_init_hessian2(b, j0_buf, d20_buf, H0.data()));
	TRY(mtfhip_am_cmpt_curr_hessian2(b, jt_buf, d2t_buf, H));
	for (size_t i = 0; i < H0.size(); ++i) H[i] += H0[i];
	return MTFHIP_OK;
}


} /* extern "C" */
