/*
 * api_fused.hip -- the fused path: template initialisation, iterate / track, NCC moment forms, candidate scoring, NN dataset rows
 * (C-ABI implementation, include/mtfhip.h; shared declarations: mtfhip_api_internal.h)
 *
 * No CPU fallback exists: every entry point either runs its HIP kernels or returns an error.
 */
#include "mtfhip_api_internal.h"
#include <chrono>

extern "C" {

/* ------------------------------------------------------------------ fused path */
static int check_sm(const mtfhip_batch *b, const mtfhip_sm_desc *sm, const char *fn) {
	if (!b || !sm) return fail(MTFHIP_ERR_INVALID_ARG, "%s: NULL argument", fn);
	if (sm->sm < MTFHIP_SM_ESM || sm->sm > MTFHIP_SM_ICLK) return fail(MTFHIP_ERR_INVALID_ARG, "%s: unknown search method %d", fn, sm->sm);
	int max_h = sm->sm == MTFHIP_SM_ESM ? 5 : 2;
	if (sm->hess_type < 0 || sm->hess_type > max_h) return fail(MTFHIP_ERR_INVALID_ARG, "%s: hess_type %d invalid for search method %d", fn, sm->hess_type, sm->sm);
	if (b->desc.am == MTFHIP_AM_NCC) {
		/* NCC overrides the second-order cmptInit / CurrHessian (NCC.cc:391-410) but not cmptSelfHessian: the self types throw
		 * FunctonNotImplemented in the reference (AppearanceModel.h:188-191) */
		if (sm->sec_ord_hess && !(sm->sm == MTFHIP_SM_ESM ? sm->hess_type >= 3 : sm->hess_type == 2))
			return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "%s: NCC has no second-order self Hessian (AppearanceModel.h:188-191)", fn);
		return MTFHIP_OK;
	}
	if (b->desc.am == MTFHIP_AM_MI) {
		/* fused MI iteration: every first-order Jacobian / Hessian type of the three search methods; with sec_ord_hess the Std types
		 * (MI.cc:659-695: cmptInitHessian / cmptCurrHessian + sum_p df_dI(p) d2I_dp2(p)) and the self types (MI.cc:697-735: the
		 * current pixel Hessian under the self gradient factor; the initial self Hessian takes its second-order part at
		 * initialize / setRegion).  Eight bins (the weights are read from the 8-bin gradient-factor tables). */
		return MTFHIP_OK;
	}
	if (b->desc.am != MTFHIP_AM_SSD) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "%s: unknown appearance model", fn);
	return MTFHIP_OK;
}

/* inverse of a definite S x S matrix (column-major) by Gauss-Jordan on the diagonally scaled system */
static bool invert_definite(int S, const double *H, double *Hinv) {
	double A[8][16], sc[8];
	for (int i = 0; i < S; ++i) { double d = std::fabs(H[i * S + i]); sc[i] = d > 0 ? 1.0 / std::sqrt(d) : 1.0; }
	for (int i = 0; i < S; ++i)
		for (int j = 0; j < S; ++j) { A[i][j] = H[j * S + i] * sc[i] * sc[j]; A[i][S + j] = i == j ? 1.0 : 0.0; }
	for (int k = 0; k < S; ++k) {
		int piv = k;
		for (int i = k + 1; i < S; ++i) if (std::fabs(A[i][k]) > std::fabs(A[piv][k])) piv = i;
		if (A[piv][k] == 0) return false;
		if (piv != k) for (int j = 0; j < 2 * S; ++j) std::swap(A[piv][j], A[k][j]);
		const double p = A[k][k];
		for (int j = 0; j < 2 * S; ++j) A[k][j] /= p;
		for (int i = 0; i < S; ++i) {
			if (i == k) continue;
			const double f = A[i][k];
			if (f == 0) continue;
			for (int j = 0; j < 2 * S; ++j) A[i][j] -= f * A[k][j];
		}
	}
	for (int i = 0; i < S; ++i)
		for (int j = 0; j < S; ++j) Hinv[j * S + i] = A[i][S + j] * sc[i] * sc[j];
	return true;
}

/* ---- NCC on the fused path: everything NCC.cc derives from centred vectors, written in raw moments ----
 * With mt = mean(It), m0 = mean(I0), b = |It - mt|, c = |I0 - m0|, f = a / (b c)  (NCC.cc:124-161) and, for a pixel
 * Jacobian X with column sums sX, Gram(X), sum It X = itX, sum I0 X = i0X:
 *   Jc = (X - mean(X)) / b                       G(X)  = -Jc^T Jc            = -(Gram(X) - sX sX^T / N) / b^2
 *   ut(X) = Jc^T (It - mt) / b = (itX - mt sX) / b^2        u0(X) = Jc^T (I0 - m0) / c = (i0X - m0 sX) / (b c)
 *   df_dIt . X = u0 - f ut   (NCC.cc:196-234, 252-266)       df_dI0 . X = (b / c) (ut - f u0)   (NCC.cc:163-194, 236-250)
 *   cmptCurrHessian = f G - ut u0^T - u0 ut^T + 3 ut ut^T   (NCC.cc:304-335)    cmptInitHessian: ... + 3 u0 u0^T (NCC.cc:282-303)
 *   cmptSelfHessian = G + ut ut^T   (NCC.cc:337-389)
 * (the reference also subtracts the mean of the gradient vectors, which is zero up to rounding because the centred
 * vectors sum to zero; it does not survive into the moments).  Moments of the mean Jacobian (J0 + Jt) / 2 are the means of
 * the moments, except its Gram matrix, which the kernel accumulates itself when hess_mean is set. */
struct NccX { const double *gram; double s[8], it[8], i0[8]; };
struct NccScalars { double N, mt, m0, b, b2, c, f; };
static void ncc_vecs(const NccScalars &q, const NccX &X, int S, double *ut, double *u0) {
	for (int s = 0; s < S; ++s) {
		ut[s] = (X.it[s] - q.mt * X.s[s]) / q.b2;
		u0[s] = (X.i0[s] - q.m0 * X.s[s]) / (q.b * q.c);
	}
}
/* kind 0 init, 1 curr, 2 self; H column-major S x S */
static void ncc_hess_from_moments(const NccScalars &q, const NccX &X, int S, int kind, double *H) {
	double ut[8], u0[8];
	ncc_vecs(q, X, S, ut, u0);
	for (int r = 0; r < S; ++r)
		for (int c = 0; c < S; ++c) {
			const int a = r < c ? r : c, d = r < c ? c : r;
			const double G = -(X.gram[a * 8 - (a * (a - 1)) / 2 + (d - a)] - X.s[r] * X.s[c] / q.N) / q.b2;
			double v;
			if (kind == 2) v = G + ut[r] * ut[c];
			else v = q.f * G - ut[r] * u0[c] - u0[r] * ut[c] + 3 * (kind == 1 ? ut[r] * ut[c] : u0[r] * u0[c]);
			H[c * S + r] = v;
		}
}
static NccScalars ncc_scalars(const mtfhip_batch *b, const TargetHost &h, const double *M) {
	NccScalars q;
	q.N = (double)b->N; q.mt = M[NCC_IT] / q.N; q.m0 = h.I0_mean; q.c = h.c;
	const double a = M[NCC_I0IT] - q.N * q.m0 * q.mt;
	q.b2 = M[NCC_IT2] - q.N * q.mt * q.mt; q.b = std::sqrt(q.b2);
	q.f = a / (q.b * q.c);
	return q;
}
static void ncc_x(const mtfhip_batch *b, const TargetHost &h, const double *M, int which /* 0 J0, 1 Jt, 2 Jm */, bool gram_is_mean, NccX &X) {
	const int S = b->S;
	for (int s = 0; s < 8; ++s) X.s[s] = X.it[s] = X.i0[s] = 0;
	for (int s = 0; s < S; ++s) {
		const double s0 = h.ncc_sj0[s], it0 = M[NCC_ITJ0 + s], i00 = h.ncc_i0j0[s];
		const double st = M[NCC_SJ + s], itt = M[NCC_ITJ + s], i0t = M[NCC_I0J + s];
		if (which == 0) { X.s[s] = s0; X.it[s] = it0; X.i0[s] = i00; }
		else if (which == 1) { X.s[s] = st; X.it[s] = itt; X.i0[s] = i0t; }
		else { X.s[s] = (s0 + st) / 2; X.it[s] = (it0 + itt) / 2; X.i0[s] = (i00 + i0t) / 2; }
	}
	X.gram = which == 0 ? h.ncc_gram0 : ((which == 2) == gram_is_mean ? M + NCC_GRAM : nullptr);
}
/* one target's reduced moment row -> the SM's f, g, H (before LM damping); NT/ESM.cc:298-377, NT/FCLK.cc:260-288, NT/ICLK.cc:206-251 */
static int ncc_assemble(const mtfhip_batch *b, const mtfhip_sm_desc *sm, bool hess_mean, const double *M, TargetHost &h,
	double *f, double *g, double *H) {
	const int S = b->S;
	const NccScalars q = ncc_scalars(b, h, M);
	h.It_mean = q.mt; h.b = q.b; h.a = M[NCC_I0IT] - q.N * q.m0 * q.mt; h.f = q.f;
	if (f) *f = q.f;
	NccX X0, Xt, Xm;
	ncc_x(b, h, M, 0, hess_mean, X0); ncc_x(b, h, M, 1, hess_mean, Xt); ncc_x(b, h, M, 2, hess_mean, Xm);
	double ut[8], u0[8];
	auto curr_jac = [&](const NccX &X, double *o) { ncc_vecs(q, X, S, ut, u0); for (int s = 0; s < S; ++s) o[s] = u0[s] - q.f * ut[s]; };
	auto init_jac = [&](const NccX &X, double *o) { ncc_vecs(q, X, S, ut, u0); for (int s = 0; s < S; ++s) o[s] = (q.b / q.c) * (ut[s] - q.f * u0[s]); };
	if (sm->sm == MTFHIP_SM_FCLK) curr_jac(Xt, g);
	else if (sm->sm == MTFHIP_SM_ICLK) init_jac(X0, g);
	else if (sm->jac_type == 0) curr_jac(Xm, g);
	else { double gt[8], g0[8]; curr_jac(Xt, gt); init_jac(X0, g0); for (int s = 0; s < S; ++s) g[s] = 0.5 * (gt[s] - g0[s]); }
	const int ht = sm->hess_type;
	auto need = [&](const NccX &X) { return X.gram ? MTFHIP_OK : fail(MTFHIP_ERR_LOGIC, "fused NCC: the Gram matrix this Hessian needs was not accumulated"); };
	if (ht == 0) { std::memcpy(H, h.h0, sizeof(double) * S * S); return MTFHIP_OK; }
	if (sm->sm == MTFHIP_SM_ICLK) { ncc_hess_from_moments(q, X0, S, 0, H); return MTFHIP_OK; }   /* Std: cmptInitHessian(J0) */
	if (sm->sm == MTFHIP_SM_FCLK || ht == 1 || ht == 5) { TRY(need(Xt)); ncc_hess_from_moments(q, Xt, S, ht == 1 ? 2 : 1, H); return MTFHIP_OK; }
	if (ht == 2) {   /* SumOfSelf */
		TRY(need(Xt)); ncc_hess_from_moments(q, Xt, S, 2, H);
		for (int k = 0; k < S * S; ++k) H[k] = 0.5 * (H[k] + h.h0[k]);
		return MTFHIP_OK;
	}
	if (ht == 3) { TRY(need(Xm)); ncc_hess_from_moments(q, Xm, S, 1, H); return MTFHIP_OK; }   /* Original: cmptCurrHessian(mean) */
	/* SumOfStd: (cmptInitHessian(J0) + cmptCurrHessian(Jt)) / 2 */
	TRY(need(Xt));
	double Hi[64];
	ncc_hess_from_moments(q, X0, S, 0, Hi); ncc_hess_from_moments(q, Xt, S, 1, H);
	for (int k = 0; k < S * S; ++k) H[k] = 0.5 * (H[k] + Hi[k]);
	return MTFHIP_OK;
}
/* sum J0, sum I0 J0 and Gram(J0) of the template (after every change of J0) */
int gemv_to_host(mtfhip_batch *b, const double *v1, int j1, const double *v2, int j2, int sum_mode, double *g, int diff);
int ncc_template_moments(mtfhip_batch *b) {
	const int nblk = simple_blocks_per_target(b->N), S = b->S;
	{
		TimedScope ts(b->ctx, "ncc_hess");
		launch_col_sum(b->view(), b->buf[MTFHIP_BUF_J0], b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t)
		for (int s = 0; s < 8; ++s) b->th[t].ncc_sj0[s] = s < S ? b->h_acc[(size_t)t * ACC_COUNT + ACC_G + s] : 0.0;
	std::vector<double> g((size_t)b->B * S);
	TRY(gemv_to_host(b, b->buf[MTFHIP_BUF_I0], MTFHIP_BUF_J0, nullptr, -1, 0, g.data(), 0));
	for (int t = 0; t < b->B; ++t)
		for (int s = 0; s < 8; ++s) b->th[t].ncc_i0j0[s] = s < S ? g[(size_t)t * S + s] : 0.0;
	{
		TimedScope ts(b->ctx, "gram");
		launch_gram(b->view(), b->buf[MTFHIP_BUF_J0], b->d_partials, nblk, b->ctx->stream);
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) std::memcpy(b->th[t].ncc_gram0, b->h_acc + (size_t)t * ACC_COUNT + ACC_H, sizeof(double) * 36);
	/* device copy for the device-side finish (k_finish_track) */
	if (!b->d_ncc_tm) HIP_TRY(hipMalloc(&b->d_ncc_tm, sizeof(double) * 52 * (size_t)b->B));
	std::vector<double> tm((size_t)52 * b->B);
	for (int t = 0; t < b->B; ++t) {
		std::memcpy(&tm[52 * (size_t)t], b->th[t].ncc_sj0, sizeof(double) * 8);
		std::memcpy(&tm[52 * (size_t)t + 8], b->th[t].ncc_i0j0, sizeof(double) * 8);
		std::memcpy(&tm[52 * (size_t)t + 16], b->th[t].ncc_gram0, sizeof(double) * 36);
	}
	HIP_TRY(hipMemcpyAsync(b->d_ncc_tm, tm.data(), sizeof(double) * tm.size(), hipMemcpyHostToDevice, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	return MTFHIP_OK;
}

/* deferred fusion, NCC: the AM-level Jacobian the trigger asked for, and the moment rows kept for the Hessian calls */
int ncc_lazy_outputs(mtfhip_batch *b, int trig, int j_a, bool hess_mean, double *g) {
	mtfhip_batch::Lazy &L = b->lz;
	const int S = b->S;
	for (int t = 0; t < b->B; ++t) {
		const double *M = b->h_acc + (size_t)t * NCC_ACC_COUNT;
		TargetHost &h = b->th[t];
		const NccScalars q = ncc_scalars(b, h, M);
		h.It_mean = q.mt; h.b = q.b; h.a = M[NCC_I0IT] - q.N * q.m0 * q.mt; h.f = q.f;
		NccX X;
		double ut[8], u0[8], *o = g + (size_t)t * S;
		if (trig == LAZY_INIT_JAC) {
			ncc_x(b, h, M, 0, hess_mean, X); ncc_vecs(q, X, S, ut, u0);
			for (int s = 0; s < S; ++s) o[s] = (q.b / q.c) * (ut[s] - q.f * u0[s]);
		} else {
			ncc_x(b, h, M, (trig == LAZY_CURR_JAC && j_a == MTFHIP_BUF_JM) ? 2 : 1, hess_mean, X); ncc_vecs(q, X, S, ut, u0);
			for (int s = 0; s < S; ++s) o[s] = u0[s] - q.f * ut[s];
			if (trig == LAZY_DIFF_JAC) {   /* (df_dIt . Jt) - (df_dI0 . J0), NCC.cc:268-280 */
				ncc_x(b, h, M, 0, hess_mean, X); ncc_vecs(q, X, S, ut, u0);
				for (int s = 0; s < S; ++s) o[s] -= (q.b / q.c) * (ut[s] - q.f * u0[s]);
			}
		}
	}
	b->ncc_host_newer = true;
	if (!L.no_cache) {
		L.ncc_M.assign(b->h_acc, b->h_acc + (size_t)NCC_ACC_COUNT * b->B);
		L.ncc_M_mean = hess_mean;
		L.ncc_M_it = L.ver[MTFHIP_BUF_IT]; L.ncc_M_jt = L.ver[MTFHIP_BUF_JT]; L.ncc_M_jm = L.ver[MTFHIP_BUF_JM];
	}
	return MTFHIP_OK;
}
/* 1 when H was produced from the cached moment rows */
int ncc_hessian_from_cache(mtfhip_batch *b, int j_buf, int kind, double *H) {
	mtfhip_batch::Lazy &L = b->lz;
	if (L.no_cache || L.ncc_M.empty() || L.ncc_M_it != L.ver[MTFHIP_BUF_IT]) return 0;
	int which;
	if (j_buf == MTFHIP_BUF_J0) { if (L.ncc_tm_ver != L.ver[MTFHIP_BUF_J0]) return 0; which = 0; }
	else if (j_buf == MTFHIP_BUF_JT) { if (L.ncc_M_mean || L.ncc_M_jt != L.ver[MTFHIP_BUF_JT]) return 0; which = 1; }
	else { if (!L.ncc_M_mean || L.ncc_M_jm != L.ver[MTFHIP_BUF_JM] || L.ncc_M_jt != L.ver[MTFHIP_BUF_JT] || L.ncc_tm_ver != L.ver[MTFHIP_BUF_J0]) return 0; which = 2; }
	for (int t = 0; t < b->B; ++t) {
		const double *M = &L.ncc_M[(size_t)t * NCC_ACC_COUNT];
		const NccScalars q = ncc_scalars(b, b->th[t], M);
		NccX X;
		ncc_x(b, b->th[t], M, which, L.ncc_M_mean, X);
		if (!X.gram) return 0;
		ncc_hess_from_moments(q, X, b->S, kind, H + (size_t)t * b->S * b->S);
	}
	return 1;
}

/* getSimilarity() right after updatePixVals + updateSimilarity -- Levenberg-Marquardt's test in the middle of every
 * iteration (NT/ESM.cc:186-204, FCLK.cc:205-223, ICLK.cc:181-199): one launch of the lean (ICLK-type) fused kernel
 * writes IT and accumulates what f needs, instead of sample + residual (SSD) or sample + two reduction passes with
 * two host round trips (NCC).  Anything else pending, or nothing pending: not taken, the caller flushes. */
int lazy_try_similarity(mtfhip_batch *b) {
	mtfhip_batch::Lazy &L = b->lz;
	if (!L.enabled || !L.pv || !L.sim || L.pv > L.sim || L.gp || L.pg || L.pj || L.cg || L.ig || L.jm) return MTFHIP_OK;
	if (!b->init_pix_vals || !b->init_sim || !b->have_corners || !b->ctx->img.data || b->ctx->img.channels != 1) return MTFHIP_OK;
	const bool ncc = b->desc.am == MTFHIP_AM_NCC;
	mtfhip_sm_desc sm;
	std::memset(&sm, 0, sizeof(sm));
	sm.sm = MTFHIP_SM_ICLK; sm.hess_type = 0; sm.materialize = 1; sm.max_iters = 1; sm.chained_warp = 1;
	FusedArgs fa;
	TRY(fused_args(b, &sm, fa));
	if (b->desc.am == MTFHIP_AM_MI) {
		/* MI: the lean launch writes It (its SSD sums are ignored), then the histogram pass and the table kernel */
		const int nb = b->desc.mi_n_bins, nblk_mi = mi_blocks(b);
		hipStream_t st = b->ctx->stream;
		{
			TimedScope ts(b->ctx, "fused_lk");
			launch_fused_ssd(fused_view(b, fa), b->ctx->img, fa, b->d_partials, fused_blocks_per_target(b->N, b->B), st);
		}
		touch(b, MTFHIP_BUF_IT);
		b->it_valid = true;
		L.it_epoch = L.epoch;
		L.df0_it_ver = L.ver[MTFHIP_BUF_IT];
		L.pv = L.sim = 0;
		const bool self = L.mi_want_self && !L.no_cache;
		{
			TimedScope ts(b->ctx, "mi_hist");
			const double *It = b->buf[MTFHIP_BUF_IT], *I0 = b->buf[MTFHIP_BUF_I0];
			if (self) launch_mi_hist_self(b->view(), nb, b->mi_hist_norm, It, I0, b->d_mi_part, nblk_mi, b->mi_row_len, st);
			else launch_mi_hist(b->view(), nb, b->mi_hist_norm, It, I0, b->d_mi_part, nblk_mi, b->mi_row_len, st);
			launch_mi_tables_iter(b->view(), nb, b->desc.mi_pre_seed, b->mi_hist_norm, self ? 1 : 0, b->d_mi_part, nblk_mi, b->mi_row_len, b->d_mi_tb,
				b->d_mi_f, st);
		}
		L.mi_self_it = self ? L.ver[MTFHIP_BUF_IT] : -1;
		std::vector<double> fv(b->B);
		HIP_TRY(hipMemcpyAsync(fv.data(), b->d_mi_f, sizeof(double) * b->B, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
		for (int t = 0; t < b->B; ++t) b->th[t].f = fv[t];
		return MTFHIP_OK;
	}
	TRY(protect_stale(b, !ncc, false));   /* SSD's updateSimilarity re-produces df_dI0 */
	const int nblk = fused_blocks_per_target(b->N, b->B);
	{
		TimedScope ts(b->ctx, "fused_lk");
		launch_fused_ssd(fused_view(b, fa), b->ctx->img, fa, b->d_partials, nblk, b->ctx->stream);
	}
	touch(b, MTFHIP_BUF_IT);
	b->it_valid = true;
	L.it_epoch = L.epoch;
	L.df0_it_ver = L.ver[MTFHIP_BUF_IT];
	if (!ncc) { L.df0_stale = true; L.df0_sh = false; if (!L.dft_sh) L.shadow_valid = false; }
	L.pv = L.sim = 0;
	if (ncc) {
		TRY(read_rows(b, nblk, NCC_ACC_COUNT));
		for (int t = 0; t < b->B; ++t) {
			const double *M = b->h_acc + (size_t)t * NCC_ACC_COUNT;
			TargetHost &h = b->th[t];
			const NccScalars q = ncc_scalars(b, h, M);
			h.It_mean = q.mt; h.b = q.b; h.a = M[NCC_I0IT] - q.N * q.m0 * q.mt; h.f = q.f;
		}
		b->ncc_host_newer = true;
		if (!L.no_cache) {   /* sum It J0 and the scalars: enough for cmptInitJacobian / cmptInitHessian of this IT */
			L.ncc_M.assign(b->h_acc, b->h_acc + (size_t)NCC_ACC_COUNT * b->B);
			L.ncc_M_mean = false; L.ncc_M_it = L.ver[MTFHIP_BUF_IT]; L.ncc_M_jt = L.ncc_M_jm = -1;
		}
		return MTFHIP_OK;
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) b->th[t].f = -b->h_acc[(size_t)t * ACC_COUNT + ACC_RR] / 2;
	if (!L.no_cache) {
		L.sim_g.resize((size_t)8 * b->B);
		for (int t = 0; t < b->B; ++t) std::memcpy(&L.sim_g[(size_t)8 * t], b->h_acc + (size_t)t * ACC_COUNT + ACC_G, sizeof(double) * 8);
		L.sim_g_it = L.ver[MTFHIP_BUF_IT]; L.sim_g_j0 = L.ver[MTFHIP_BUF_J0];
	}
	return MTFHIP_OK;
}

/* The fused iteration serves the multi-channel models too: MCSSD / MCNCC through k_fused_mc, MCMI through the materialising MI
 * iteration (k_fused_mc + the histogram / gradient / Hessian kernels, which see (pixel, channel) rows like any other rows); the
 * MI recompute passes are single-channel (mi_fast_ok). */
static int fused_channels_ok(const mtfhip_batch *b, const char *fn) { (void)b; (void)fn; return MTFHIP_OK; }

/* the initial self Hessian a search method keeps (NT/ESM.cc:133-141, NT/FCLK.cc:120-128, NT/ICLK.cc:96-118).  MI with sec_ord_hess: its
 * second-order form (MI.cc:697-735) over the template's pixel Hessian, which is materialised for this one call. */
static int init_self_hessian(mtfhip_batch *b, const mtfhip_sm_desc *sm, double *H0) {
	if (b->desc.am == MTFHIP_AM_MI && sm->sec_ord_hess && b->C == 1) {
		TRY(mtfhip_ssm_cmpt_pix_hessian(b, b->d0_variant, MTFHIP_BUF_D2I0_DX2, MTFHIP_BUF_DI0_DX, MTFHIP_BUF_D2I0_DP2));
		return mtfhip_am_cmpt_self_hessian2(b, MTFHIP_BUF_J0, MTFHIP_BUF_D2I0_DP2, H0);
	}
	return mtfhip_am_cmpt_self_hessian(b, MTFHIP_BUF_J0, H0);
}
/* nt::ICLK::initialize of small single-channel SSD / NCC patches in ONE launch (kernels_init.hip): the grid tracker re-initialises its
 * 256 patch trackers after every frame with the shipped reset_at_each_frame = 1 (GridTracker.cc:273-274, 345-392), and call by call
 * that was 385 us per frame against 42 us for tracking them (r05, tools/grid_modes_probe.py).  Nothing is waited for: the small
 * results the host mirrors hold (H0, NCC scalars, template moments) arrive in a pinned record and are folded in by the next entry
 * point that flushes (pull_init_mirrors).  MTFHIP_INIT_FUSED=0 keeps the call-by-call form (A/B and the equality test). */
static bool template_init_fused_ok(const mtfhip_batch *b, const mtfhip_sm_desc *sm) {
	const char *e = std::getenv("MTFHIP_INIT_FUSED");   /* (read per call: the tests flip it) */
	if (e && e[0] == '0') return false;
	const int am = b->desc.am;
	if (am != MTFHIP_AM_SSD && am != MTFHIP_AM_NCC) return false;
	const bool const_h = sm->hess_type == 0 || (sm->hess_type == 2 && am == MTFHIP_AM_SSD);
	return sm->sm == MTFHIP_SM_ICLK && const_h && sm->chained_warp && !sm->sec_ord_hess && b->C == 1 && b->N <= kTemplateInitMaxPix &&
		b->h_flag_dev != nullptr && b->ctx->img.data != nullptr && b->ctx->img.channels == 1;
}
static int init_template_fused(mtfhip_batch *b, const mtfhip_sm_desc *sm, const RegionIngest *rg = nullptr, bool publish_host = true) {
	(void)sm;
	hipStream_t st = b->ctx->stream;
	const bool ncc = b->desc.am == MTFHIP_AM_NCC;
	TRY(need_image(b));
	for (int id : {MTFHIP_BUF_I0, MTFHIP_BUF_IT, MTFHIP_BUF_DI0_DX, MTFHIP_BUF_DIT_DX, MTFHIP_BUF_J0, MTFHIP_BUF_DF_DI0, MTFHIP_BUF_DF_DIT}) TRY(ensure_buf(b, id));
	if (!b->h_init_rec) {
		HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_init_rec), sizeof(double) * kInitRec * (size_t)b->B, hipHostMallocMapped));
		HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->h_init_rec_dev), b->h_init_rec, 0));
		HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_init_flag), sizeof(unsigned long long), hipHostMallocMapped));
		*b->h_init_flag = 0;
		HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->h_init_flag_dev), b->h_init_flag, 0));
	}
	if (ncc && !b->d_ncc_tm) HIP_TRY(hipMalloc(&b->d_ncc_tm, sizeof(double) * 52 * (size_t)b->B));
	++b->frame_count;   /* ImageBase.cc:74 */
	const unsigned long long seq = ++b->init_seq;
	{
		TimedScope tsc(b->ctx, "template_init");
		launch_template_init(b->view(), b->ctx->img, b->desc.grad_eps, b->norm_mult, b->norm_add, b->d_h0, b->d_h0inv, ncc ? b->d_ncc : nullptr,
			ncc ? b->d_ncc_tm : nullptr, publish_host ? InitPublish{b->h_init_rec_dev, b->d_fin_count, b->h_init_flag_dev, seq, publish_fenced()} : InitPublish{nullptr, nullptr, nullptr, 0, 0},
			rg ? *rg : RegionIngest{}, st);
	}
	/* (the kernel also zeroes the gradient vectors df_dI0 / df_dIt: initializeSimilarity / initializeGrad) */
	touch(b, MTFHIP_BUF_DI0_DX); touch(b, MTFHIP_BUF_J0);
	b->init_pix_vals = b->it_valid = true;
	b->init_pix_grad = b->dit_valid = true;
	b->init_sim = b->init_grad = true;
	for (auto &h : b->th) h.f = ncc ? 1.0 : 0.0;
	b->ncc_host_newer = false;       /* (the kernel wrote d_ncc itself) */
	b->init_mirror_seq = seq;
	b->init_rec_device = !publish_host;
	b->j0_is_template = true;
	b->j0_template_corners_epoch = b->corners_epoch;
	b->j0_variant = MTFHIP_JAC_WARPED;
	b->template_corners.resize(8 * (size_t)b->B);
	for (int t = 0; t < b->B; ++t) std::memcpy(&b->template_corners[8 * t], b->th[t].init_corners, sizeof(double) * 8);
	return MTFHIP_OK;
}
/* resetTrackers(reinit) for the patches of a grid: setCorners + initialize of every patch tracker in ONE launch -- the host half of the
 * reset (mirrors, staged corners; set_corners_core deferred) and k_template_init in region mode, which reads the patch corners from the
 * pinned staging buffer and lays out its own grid (as k_iclk_track does for the per-frame setRegion) */
static int grid_reinit_fused(mtfhip_batch *b, const mtfhip_sm_desc *sm, const double *patches, bool layout_later = false) {
	static const bool dbg = std::getenv("MTFHIP_TRACK_DEBUG_TIMING") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	/* a record of the PREVIOUS fused initialisation that nobody has asked for (reset-every-frame mode: mtfhip_grid_frame holds it back) is
	 * superseded by this one: every mirror it would fill is rewritten by the new record -- not folding it in saves the host 1 KB per patch of
	 * cold reads (12 us per frame at 256 patches).  With recorded interface calls pending the flush below still wants it. */
	/* (r05 advisor: a call that fails before the new launch is enqueued -- check_sm, need_image, degenerate corners in set_corners_core -- must not
	 * leave the mirrors older than d_h0 / d_ncc / d_ncc_tm with nothing pending: the dropped record is put back on those paths) */
	const unsigned long long dropped_seq = (b->init_mirror_seq && !b->lz.any()) ? b->init_mirror_seq : 0;
	const bool dropped_dev = b->init_rec_device;
	if (dropped_seq) b->init_mirror_seq = 0;
#define REINIT_TRY(expr) do { const int _rc = (expr); if (_rc != MTFHIP_OK) { if (dropped_seq && !b->init_mirror_seq) { b->init_mirror_seq = dropped_seq; b->init_rec_device = dropped_dev; } return _rc; } } while (0)
	REINIT_TRY(lazy_flush(b, false));   /* (the current points are about to be replaced: no apply_warp for them -- 7 us per frame when this was FLUSH) */
	touch_all(b); b->lz.it_epoch = -1; REINIT_TRY(ensure_df(b));
	REINIT_TRY(check_sm(b, sm, "init_template"));
	REINIT_TRY(need_image(b));
	const auto t1 = std::chrono::steady_clock::now();
	REINIT_TRY(set_corners_core(b, layout_later ? nullptr : patches, false, true, layout_later));   /* (layout_later: b->deferred_gdesc / _region / _region_map are set, mtfhip_grid_reset) */
#undef REINIT_TRY
	const auto t2 = std::chrono::steady_clock::now();
	b->init_pix_vals = b->init_pix_grad = b->init_sim = b->init_grad = false;
	const bool homg = b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY;
	RegionIngest rg{};
	const double *stage = reinterpret_cast<const double *>(b->h_stage_a_dev);
	rg.corners = stage + 17 * (size_t)b->B; rg.ncc = nullptr;
	rg.d_ncc = b->d_ncc; rg.d_w0 = b->d_w0; rg.d_init_corners_hm = b->d_init_corners_hm;
	rg.lo_x = homg ? -0.5 : 1 - b->desc.resx / 2.0; rg.lo_y = homg ? -0.5 : 1 - b->desc.resy / 2.0;
	rg.hi_x = homg ? 0.5 : b->desc.resx / 2.0; rg.hi_y = homg ? 0.5 : b->desc.resy / 2.0;
	rg.resx = b->desc.resx; rg.resy = b->desc.resy; rg.force_unit_z = homg ? 0 : 1;
	if (layout_later) {
		const mtfhip_grid_desc &gd = b->deferred_gdesc;
		rg.layout = 1;
		rg.grid = GridLayoutHD{gd.grid_size_x, gd.grid_size_y, gd.patch_size_x, gd.patch_size_y, gd.dyn_patch_size ? 1 : 0, gd.patch_centroid_inside ? 1 : 0};
		std::memcpy(rg.region_map, b->deferred_region_map, sizeof(rg.region_map));
	}
	/* (no host publish: a grid re-initialises every frame and its records are superseded unread -- the pinned stores and their acknowledgement
	 * were ~2 us at the tail of every workgroup; a caller that does read the mirrors copies d_h0 / d_ncc / d_ncc_tm, pull_init_mirrors) */
	static const bool rec_pinned = std::getenv("MTFHIP_GRID_INIT_PUBLISH") && std::getenv("MTFHIP_GRID_INIT_PUBLISH")[0] == '1';
	const int rc = init_template_fused(b, sm, &rg, rec_pinned);
	const auto t3 = std::chrono::steady_clock::now();
	set_corners_finish_deferred(b);   /* the host half of a deferred reset (a no-op when nothing was deferred): under the kernel */
	if (dbg) {
		const auto t4 = std::chrono::steady_clock::now();
		auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::micro>(c - a).count(); };
		static double a1 = 0, a2 = 0, a3 = 0, a4 = 0; static int n = 0;
		a1 += us(t0, t1); a2 += us(t1, t2); a3 += us(t2, t3); a4 += us(t3, t4);
		if (++n % 100 == 0) { std::fprintf(stderr, "[grid_reinit] flush + checks %.1f us, set_corners (deferred) %.1f, init_template_fused (launch) %.1f, deferred host half %.1f (mean of 100)\n", a1 / 100, a2 / 100, a3 / 100, a4 / 100); a1 = a2 = a3 = a4 = 0; }
	}
	/* (init_template_fused took the template corners from the mirrors, which the deferred half has only now brought up to date) */
	for (int t = 0; t < b->B; ++t) std::memcpy(&b->template_corners[8 * t], b->th[t].init_corners, sizeof(double) * 8);
	b->warps_dirty = true;   /* the device slab still holds the previous frame's warps: whoever needs them next uploads the (identity) mirrors */
	/* ... except the one-launch loop kernels, which start a freshly re-initialised patch from init_corners_hm (TrackState::fresh_reset) */
	b->fresh_reinit = rc == MTFHIP_OK && !(std::getenv("MTFHIP_GRID_FRESH") && std::getenv("MTFHIP_GRID_FRESH")[0] == '0');
	if (rc == MTFHIP_OK) { HIP_TRY(hipEventRecord(b->ev_a, b->ctx->stream)); b->stage_a_busy = true; }   /* the kernel reads the staging buffer */
	return rc;
}
int mtfhip_batch_init_template(mtfhip_batch *b, const mtfhip_sm_desc *sm) {
	FLUSH(b);
	if (b) { touch_all(b); b->lz.it_epoch = -1; TRY(ensure_df(b)); }
	TRY(check_sm(b, sm, "init_template"));
	TRY(fused_channels_ok(b, "init_template"));
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "init_template before set_corners");
	/* am->clearInitStatus() (NT/ESM.cc:113, NT/FCLK.cc:105, NT/ICLK.cc:74) */
	b->init_pix_vals = b->init_pix_grad = b->init_sim = b->init_grad = false;
	/* k_template_init samples INIT_PTS and builds J0 at the identity warp: that is the current image at the current points only while no
	 * setState / compositionalUpdate / update() has moved the warp since set_corners (the mirrors are exact: set_corners_core stores the
	 * identity itself, and every state change goes through them) */
	bool at_identity = true;
	{
		const M3 I = m3_identity();
		for (const TargetHost &h : b->th) if (std::memcmp(h.warp.m, I.m, sizeof(I.m)) != 0) { at_identity = false; break; }
	}
	if (at_identity && template_init_fused_ok(b, sm)) return init_template_fused(b, sm);
	TRY(mtfhip_am_initialize_pix_vals(b, nullptr));
	if (sm->chained_warp) {
		TRY(mtfhip_am_initialize_pix_grad(b, nullptr));
		TRY(mtfhip_ssm_cmpt_pix_jacobian(b, MTFHIP_JAC_WARPED, MTFHIP_BUF_DI0_DX, MTFHIP_BUF_J0));
	} else {
		TRY(mtfhip_ssm_update_grad_pts(b, b->desc.grad_eps));
		TRY(mtfhip_am_initialize_pix_grad_warped(b, nullptr));
		TRY(mtfhip_ssm_cmpt_pix_jacobian(b, MTFHIP_JAC_INIT, MTFHIP_BUF_DI0_DX, MTFHIP_BUF_J0));
	}
	if (sm->sec_ord_hess) {   /* initializePixHess, NT/ESM.cc:406-416 ; the template's pixel Hessian is rebuilt per pixel from
	                           * d2I0_dx2 and dI0_dx inside k_second_order_ssd instead of being stored as an S^2 x N matrix */
		b->init_pix_hess = false;
		if (sm->chained_warp) TRY(mtfhip_am_initialize_pix_hess(b, nullptr));
		else { TRY(mtfhip_ssm_update_hess_pts(b, b->hess_eps)); TRY(mtfhip_am_initialize_pix_hess_warped(b, nullptr, nullptr)); }
		b->d0_variant = sm->chained_warp ? MTFHIP_JAC_WARPED : MTFHIP_JAC_INIT;
	}
	TRY(mtfhip_am_initialize_similarity(b));
	TRY(mtfhip_am_initialize_grad(b));
	TRY(mtfhip_am_initialize_hess(b));
	std::vector<double> H0((size_t)b->B * b->S * b->S), h0dev((size_t)b->B * 64, 0.0);
	TRY(init_self_hessian(b, sm, H0.data()));
	for (int t = 0; t < b->B; ++t) {
		std::memset(b->th[t].h0, 0, sizeof(b->th[t].h0));
		std::memcpy(b->th[t].h0, &H0[(size_t)t * b->S * b->S], sizeof(double) * b->S * b->S);
		std::memcpy(&h0dev[(size_t)t * 64], b->th[t].h0, sizeof(double) * 64);
	}
	HIP_TRY(hipMemcpyAsync(b->d_h0, h0dev.data(), sizeof(double) * h0dev.size(), hipMemcpyHostToDevice, b->ctx->stream));
	std::vector<double> hinv((size_t)b->B * 64, 0.0);
	for (int t = 0; t < b->B; ++t)
		if (!invert_definite(b->S, b->th[t].h0, &hinv[(size_t)t * 64]))
			std::fill(hinv.begin() + (size_t)t * 64, hinv.begin() + (size_t)(t + 1) * 64, 0.0); /* flat template: no update */
	HIP_TRY(hipMemcpyAsync(b->d_h0inv, hinv.data(), sizeof(double) * hinv.size(), hipMemcpyHostToDevice, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	if (b->desc.am == MTFHIP_AM_NCC) TRY(ncc_template_moments(b));
	b->j0_is_template = true;
	b->j0_template_corners_epoch = b->corners_epoch;
	b->j0_variant = sm->chained_warp ? MTFHIP_JAC_WARPED : MTFHIP_JAC_INIT;
	b->template_corners.resize(8 * (size_t)b->B);
	for (int t = 0; t < b->B; ++t) std::memcpy(&b->template_corners[8 * t], b->th[t].init_corners, sizeof(double) * 8);
	return MTFHIP_OK;
}

/* nt::ESM::setRegion NT/ESM.cc:148-168, nt::FCLK::setRegion NT/FCLK.cc:360-376, nt::ICLK::setRegion NT/ICLK.cc:131-157 (update_ssm
 * off): the SSM is reset to the new corners; ESM (and FCLK with the InitialSelf Hessian) recompute init_pix_jacobian with
 * cmptInitPixJacobian on the new grid and, for the Hessian types that use it, the constant self Hessian; ICLK keeps its
 * template Jacobian.  The template (I0, dI0_dx) is kept in every case. */
static int set_region_core(mtfhip_batch *b, const double *corners, const mtfhip_sm_desc *sm, bool for_track, bool defer_grid = false, bool layout_later = false);
int mtfhip_batch_set_region(mtfhip_batch *b, const double *corners, const mtfhip_sm_desc *sm) { return set_region_core(b, corners, sm, false); }

/* the one-launch grid kernel (k_iclk_track: a patch's whole ICLK update() in one workgroup) takes ICLK with a constant Hessian -- up to
 * four pixels per thread, where every per-pixel operand of the loop stays in registers: 3.1-4.4 us per iteration at 25 x 25 and
 * 32 x 32 against 8.8-12 for a launch per pass.  Above that the template Jacobian is re-read in every iteration and the kernel
 * falls behind the launch-per-pass loop (40 x 40: 11.3-16.9 against 9.9-13.7 us; 50 x 50 x 256: 35.3 against 14.9;
 * profiles/r03_experiments.md), so larger patches take that loop.  MTFHIP_ICLK_ONE_LAUNCH_MAX moves the boundary (experiments). */
static int iclk_one_launch_max_pix() {
	static const int v = std::getenv("MTFHIP_ICLK_ONE_LAUNCH_MAX") ? std::atoi(std::getenv("MTFHIP_ICLK_ONE_LAUNCH_MAX")) : 4 * kBlock;
	return v < kIclkTrackMaxPix ? v : kIclkTrackMaxPix;
}
static bool iclk_one_launch(const mtfhip_batch *b, const mtfhip_sm_desc *sm) {
	return b->C == 1 && sm->sm == MTFHIP_SM_ICLK && (sm->hess_type == 0 || (sm->hess_type == 2 && b->desc.am == MTFHIP_AM_SSD)) &&
		b->N <= iclk_one_launch_max_pix();
}
static bool region_refreshes(const mtfhip_sm_desc *sm) { return sm->sm == MTFHIP_SM_ESM || (sm->sm == MTFHIP_SM_FCLK && sm->hess_type == 0); }

static int set_region_core(mtfhip_batch *b, const double *corners, const mtfhip_sm_desc *sm, bool for_track, bool defer_grid, bool layout_later) {
	FLUSH_AM(b);   /* (the current points are about to be replaced: only pending calls need them brought up to date) */
	if (b) { touch_all(b); b->lz.it_epoch = -1; TRY(ensure_df(b)); }
	TRY(check_sm(b, sm, "set_region"));
	TRY(fused_channels_ok(b, "set_region"));
	if (!b->init_pix_vals) return fail(MTFHIP_ERR_LOGIC, "set_region before init_template");
	TRY(set_corners_core(b, corners, for_track, defer_grid, layout_later));
	const bool refresh = region_refreshes(sm);
	if (layout_later) {
		if (refresh) return fail(MTFHIP_ERR_LOGIC, "set_region: a layout behind the launch with a search method that refreshes its template Jacobian");
		b->deferred_template_check = true;   /* (the comparison below, once the host has the corners: set_corners_finish_deferred) */
		return MTFHIP_OK;
	}
	if (!refresh) {
		/* back on exactly the grid the kept template Jacobian was computed on: its rows can still be rebuilt from dI0_dx */
		if (b->j0_is_template && b->template_corners.size() == 8 * (size_t)b->B &&
			std::memcmp(b->template_corners.data(), corners, sizeof(double) * 8 * b->B) == 0)
			b->j0_template_corners_epoch = b->corners_epoch;
		return MTFHIP_OK;
	}
	TRY(mtfhip_ssm_cmpt_pix_jacobian(b, MTFHIP_JAC_INIT, MTFHIP_BUF_DI0_DX, MTFHIP_BUF_J0));
	const bool need_h0 = sm->hess_type == 0 || (sm->sm == MTFHIP_SM_ESM && sm->hess_type == 2);
	if (need_h0) {
		std::vector<double> H0((size_t)b->B * b->S * b->S), h0dev((size_t)b->B * 64, 0.0), hinv((size_t)b->B * 64, 0.0);
		if (b->desc.am == MTFHIP_AM_MI && sm->sec_ord_hess) b->d0_variant = MTFHIP_JAC_INIT;   /* (setRegion: cmptInitPixHessian, NT/ESM.cc:160-163) */
		TRY(init_self_hessian(b, sm, H0.data()));
		for (int t = 0; t < b->B; ++t) {
			std::memset(b->th[t].h0, 0, sizeof(b->th[t].h0));
			std::memcpy(b->th[t].h0, &H0[(size_t)t * b->S * b->S], sizeof(double) * b->S * b->S);
			std::memcpy(&h0dev[(size_t)t * 64], b->th[t].h0, sizeof(double) * 64);
			if (!invert_definite(b->S, b->th[t].h0, &hinv[(size_t)t * 64]))
				std::fill(hinv.begin() + (size_t)t * 64, hinv.begin() + (size_t)(t + 1) * 64, 0.0);
		}
		HIP_TRY(hipMemcpyAsync(b->d_h0, h0dev.data(), sizeof(double) * h0dev.size(), hipMemcpyHostToDevice, b->ctx->stream));
		HIP_TRY(hipMemcpyAsync(b->d_h0inv, hinv.data(), sizeof(double) * hinv.size(), hipMemcpyHostToDevice, b->ctx->stream));
		HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	}
	if (b->desc.am == MTFHIP_AM_NCC) TRY(ncc_template_moments(b));
	b->j0_is_template = true;
	b->j0_template_corners_epoch = b->corners_epoch;
	b->j0_variant = MTFHIP_JAC_INIT;
	b->template_corners.assign(corners, corners + 8 * (size_t)b->B);
	return MTFHIP_OK;
}

int fused_args(const mtfhip_batch *b, const mtfhip_sm_desc *sm, FusedArgs &fa) {
	fa.chained = sm->chained_warp ? 1 : 0;
	fa.materialize = sm->materialize ? 1 : 0;
	fa.hess_mean = 0;
	fa.j0_recompute = (b->j0_is_template && b->j0_recompute_enabled && b->j0_template_corners_epoch == b->corners_epoch) ? 1 : 0;
	fa.j0_init_variant = b->j0_variant == MTFHIP_JAC_INIT ? 1 : 0;
	fa.grad_eps = b->desc.grad_eps;
	fa.norm_mult = b->norm_mult; fa.norm_add = b->norm_add;
	fa.active = nullptr;
	fa.inline_warp = 0;
	fa.fast_math = (b->math_mode == MTFHIP_MATH_FAST && !fa.materialize) ? 1 : 0;
	{ int nb; fused_decomposition(b->N, b->B, nb, fa.rows_per_block); }
	switch (sm->sm) {
	case MTFHIP_SM_FCLK: fa.mode = 0; break;
	case MTFHIP_SM_ESM: fa.mode = 1; fa.hess_mean = sm->hess_type == 3; break;
	default:
		if (sm->hess_type == 1) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "fused ICLK with hess_type CurrentSelf: use the un-fused entry points");
		fa.mode = 2;
	}
	return MTFHIP_OK;
}

/* The second-order term an SSD search method adds to its Hessian (k_second_order_ssd's `term`), -1 for none:
 * SSD's self Hessians are first order by definition (SSDBase.h:95-98) and InitialSelf never looks at the frame. */
/* MI (am = MTFHIP_AM_MI): its self Hessian has a second-order form of its own (MI.cc:697-735) -- term 4, the current pixel Hessian
 * weighted by sum_r gradIt(r) sum_t matIt(t) self_grad_factor(r, t) -- for CurrentSelf and ESM's SumOfSelf (whose other half, the
 * initial self Hessian, carries its second-order part since initialize / setRegion). */
static int second_order_term(const mtfhip_sm_desc *sm, int am = MTFHIP_AM_SSD) {
	if (!sm->sec_ord_hess) return -1;
	if (am == MTFHIP_AM_MI && (sm->hess_type == 1 || (sm->sm == MTFHIP_SM_ESM && sm->hess_type == 2))) return 4;
	switch (sm->sm) {
	case MTFHIP_SM_FCLK: return sm->hess_type == 2 ? 0 : -1;
	case MTFHIP_SM_ESM: return sm->hess_type == 5 ? 0 : (sm->hess_type == 4 ? 1 : (sm->hess_type == 3 ? 2 : -1));
	default: return sm->hess_type == 2 ? 3 : -1;
	}
}

/* turns one target's reduced accumulators into the SM's g and H (before LM damping):
 * NT/FCLK.cc:260-288 ; NT/ESM.cc:298-377 with SSDBase.cc:169-191,287-311 ; NT/ICLK.cc:206-251 */
static void assemble(const mtfhip_batch *b, const mtfhip_sm_desc *sm, const double *acc, const double *h0,
	double *f, double *g, double *H) {
	const int S = b->S;
	if (f) *f = -acc[ACC_RR] / 2;
	const double gscale = sm->sm == MTFHIP_SM_ESM ? 0.5 : 1.0;
	for (int s = 0; s < S; ++s) g[s] = gscale * acc[ACC_G + s];
	const bool use_h0 = (sm->hess_type == 0) || (sm->sm == MTFHIP_SM_ICLK);
	const bool sum_h0 = (sm->sm == MTFHIP_SM_ESM) && (sm->hess_type == 2 || sm->hess_type == 4);
	int k = 0;
	for (int a = 0; a < 8; ++a)
		for (int c = a; c < 8; ++c) {
			if (a < S && c < S) {
				double v = use_h0 ? h0[c * S + a] : -acc[ACC_H + k];
				if (sum_h0) v = (v + h0[c * S + a]) * 0.5;
				H[c * S + a] = v; H[a * S + c] = v;
			}
			++k;
		}
}

/* One ESM / FCLK / ICLK iteration with MI in four pixel-level launches instead of seventeen:
 *   0. the fused LK kernel (SSD instantiation, FCLK-type, materialising): warp -> It, dIt_dx, Jt in one pass (its SSD
 *      sums are ignored; MI's pixel scaling travels in norm_mult / norm_add).  ICLK: It only.
 *   1. k_mi_hist<MFMA, SELF>: histogram of It, joint (It, I0) and -- for the self Hessians -- joint (It, It), one pass;
 *      k_mi_tables_iter: pre-seeding, logs, similarity and the three gradient-factor tables in one launch
 *      (MI.cc:346-382, 399-403, 427-431, 651-658).
 *   2. k_mi_grad_gemv: both gradient vectors and df_dIt . Jt, df_dI0 . J0 in one pass (instead of 2 x k_mi_grad, k_gemv, k_finish).
 *   3. k_mi_hess<MFMA> + k_mi_hess_finish for the self Hessian (MI.cc:565-601) when the Hessian type needs it.
 * (Folding 2 into 3 was tried: 294 VGPRs, one wave per SIMD, 223 us instead of 95 + 35.)
 * g, H of the search method as in NT/ESM.cc:298-377, NT/FCLK.cc:260-288, NT/ICLK.cc:206-251. */
/* What an MI iteration of the search method needs from the AM (NT/ESM.cc:315-377, NT/FCLK.cc:262-283, NT/ICLK.cc:204-252) */
struct MiPlan {
	enum { H_CONST, H_SELF_JT, H_CURR_JT, H_CURR_JM, H_SUM_STD, H_INIT_J0 };
	int hk;
	bool iclk, esm, fclk, self, need_jt, orig_jac, need_mean;
	explicit MiPlan(const mtfhip_sm_desc *sm) {
		iclk = sm->sm == MTFHIP_SM_ICLK; esm = sm->sm == MTFHIP_SM_ESM; fclk = sm->sm == MTFHIP_SM_FCLK;
		const int ht = sm->hess_type;
		hk = ht == 0 ? H_CONST
			: esm ? (ht <= 2 ? H_SELF_JT : (ht == 3 ? H_CURR_JM : (ht == 4 ? H_SUM_STD : H_CURR_JT)))
			: fclk ? (ht == 1 ? H_SELF_JT : H_CURR_JT)
			: (ht == 1 ? H_SELF_JT : H_INIT_J0);
		self = hk == H_SELF_JT;                /* cmptSelfHessian(Jt): the self histogram rides along with pass 1 */
		need_jt = !iclk || self;               /* ICLK's CurrentSelf refreshes the current pixel Jacobian (NT/ICLK.cc:215-237) */
		orig_jac = esm && sm->jac_type == 0;   /* cmptCurrJacobian(mean Jacobian) */
		need_mean = orig_jac || hk == H_CURR_JM;
	}
};
/* Enqueues the passes of one fused MI iteration.  Results on the device: d_mi_f [B]; d_mi_H = [B][64] Hessian (column-major
 * S x S) | [B][16] df_dIt . J, df_dI0 . J0 | [B][64] cmptInitHessian(J0) of SumOfStd.  `active` (device, may be NULL): targets
 * whose flag is 0 keep their It / Jt (the device-side loop). */
static int mi_enqueue(mtfhip_batch *b, const mtfhip_sm_desc *sm, const MiPlan &pl, const int *active, bool reduce_g = true) {
	const int nb = b->desc.mi_n_bins, nblk = mi_blocks(b);
	hipStream_t st = b->ctx->stream;
	/* 0 */
	mtfhip_sm_desc s0 = *sm;
	s0.sm = pl.need_jt ? MTFHIP_SM_FCLK : MTFHIP_SM_ICLK; s0.hess_type = pl.need_jt ? 1 : 0; s0.materialize = 1; s0.sec_ord_hess = 0;
	FusedArgs fa;
	TRY(fused_args(b, &s0, fa));
	fa.active = active;
	{
		TimedScope ts(b->ctx, "fused_lk");
		launch_fused_ssd(fused_view(b, fa), b->ctx->img, fa, b->d_partials, fused_blocks_per_target(b->N, b->B), st);
	}
	b->it_valid = true;
	b->dit_valid = b->jt_valid = pl.need_jt;
	if (pl.need_mean) {
		TRY(ensure_buf(b, MTFHIP_BUF_JM));
		TimedScope ts(b->ctx, "mean_jacobian");
		launch_mean_jacobian(b->view(), st);
	}
	/* 1 */
	const double *It = b->buf[MTFHIP_BUF_IT], *I0 = b->buf[MTFHIP_BUF_I0];
	{
		TimedScope ts(b->ctx, "mi_hist");
		if (pl.self) launch_mi_hist_self(b->view(), nb, b->mi_hist_norm, It, I0, b->d_mi_part, nblk, b->mi_row_len, st);
		else launch_mi_hist(b->view(), nb, b->mi_hist_norm, It, I0, b->d_mi_part, nblk, b->mi_row_len, st);
		launch_mi_tables_iter(b->view(), nb, b->desc.mi_pre_seed, b->mi_hist_norm, pl.self ? 1 : 0, b->d_mi_part, nblk, b->mi_row_len, b->d_mi_tb,
			b->d_mi_f, st);
	}
	/* 2 */
	double *d_g = b->d_mi_H + 64 * (size_t)b->B;
	{
		TimedScope ts(b->ctx, "mi_grad");
		const int ng = simple_blocks_per_target(b->N) < 64 ? simple_blocks_per_target(b->N) : 64;
		launch_mi_grad_gemv(b->view(), nb, b->mi_hist_norm, It, I0, b->d_mi_tb,
			pl.iclk ? nullptr : b->buf[pl.orig_jac ? MTFHIP_BUF_JM : MTFHIP_BUF_JT],
			(pl.fclk || pl.orig_jac) ? nullptr : b->buf[MTFHIP_BUF_J0], mi_j0_rebuild(b), sm->materialize ? b->buf[MTFHIP_BUF_DF_DIT] : nullptr,
			sm->materialize ? b->buf[MTFHIP_BUF_DF_DI0] : nullptr, b->d_partials, ng, st);
		if (reduce_g) launch_finish_rows(b->d_partials, ng, 16, d_g, b->B, st);   /* (the device-side loop sums the rows in its finish) */
	}
	/* 3: kind 0 init (MI.cc:461-513), 1 curr (:603-637), 2 self (:515-601), as mi_hessian in api_am.hip */
	auto hess_pass = [&](int kind, int j_buf, double *out) {
		const double *A = b->buf[kind == 0 ? MTFHIP_BUF_I0 : MTFHIP_BUF_IT], *Bv = b->buf[kind == 1 ? MTFHIP_BUF_I0 : MTFHIP_BUF_IT];
		TimedScope ts(b->ctx, "mi_hess");
		launch_mi_hess(b->view(), nb, b->mi_hist_norm, A, Bv, b->d_mi_tb, kind == 0 ? MI_T_INIT : (kind == 1 ? MI_T_CURR : MI_T_SELF), kind == 0,
			b->buf[j_buf], b->d_mi_part, nblk, b->mi_row_len, st);
		launch_finish_rows(b->d_mi_part, nblk, b->mi_row_len, b->d_mi_red, b->B, st);
		launch_mi_hess_finish(b->view(), nb, b->d_mi_red, 1, b->mi_row_len, b->d_mi_tb, kind == 2 ? MI_SELF_JOINT : MI_JOINT,
			kind == 0 ? MI_HIST_INIT : MI_HIST_CURR, kind == 0, out, st);
	};
	switch (pl.hk) {
	case MiPlan::H_SUM_STD:   /* cmptSumOfHessians = cmptInitHessian(J0) + cmptCurrHessian(Jt) (MI.h) */
		hess_pass(0, MTFHIP_BUF_J0, b->d_mi_H + 80 * (size_t)b->B);
		hess_pass(1, MTFHIP_BUF_JT, b->d_mi_H);
		break;
	case MiPlan::H_SELF_JT: hess_pass(2, MTFHIP_BUF_JT, b->d_mi_H); break;
	case MiPlan::H_CURR_JT: hess_pass(1, MTFHIP_BUF_JT, b->d_mi_H); break;
	case MiPlan::H_CURR_JM: hess_pass(1, MTFHIP_BUF_JM, b->d_mi_H); break;
	case MiPlan::H_INIT_J0: hess_pass(0, MTFHIP_BUF_J0, b->d_mi_H); break;
	default: break;
	}
	return MTFHIP_OK;
}
/* The recompute form of the same iteration (kernels_mi_fused.hip): two pixel-level launches that read 28 + 44 B/px and write
 * nothing per pixel, instead of four that move 324 B/px.  Tolerance-mode arithmetic, the reference's 8 bins, nothing
 * materialised, every first-order type but SumOfStd (two Hessian passes: it keeps the materialising form). */
static bool mi_fast_ok(const mtfhip_batch *b, const mtfhip_sm_desc *sm, const MiPlan &pl) {
	static const bool enabled = !(std::getenv("MTFHIP_MI_RECOMPUTE") && std::getenv("MTFHIP_MI_RECOMPUTE")[0] == '0');
	/* r06: other bin counts up to ten (the shipped mi_n_bins 10, Config/modules.cfg:115) in the polynomial forms of pass 2 -- the constant and the
	 * self Hessian, single channel; 8 bins: every first-order form but SumOfStd */
	const int nb = b->desc.mi_n_bins;
	const bool bins_ok = nb == 8 || (nb <= 10 && b->C == 1 && (pl.hk == MiPlan::H_CONST || pl.hk == MiPlan::H_SELF_JT));
	return enabled && b->math_mode == MTFHIP_MATH_FAST && bins_ok && !sm->materialize && pl.hk != MiPlan::H_SUM_STD;
}
static MiFastPlan mi_fast_plan(const mtfhip_batch *b, const MiPlan &pl, const int *active, const mtfhip_sm_desc *sm) {
	MiFastPlan fp;
	fp.nb = b->desc.mi_n_bins;
	fp.nonchained = (sm && !sm->chained_warp) ? 1 : 0;
	fp.hk = pl.hk == MiPlan::H_CONST ? 0 : (pl.hk == MiPlan::H_SELF_JT ? 1 : (pl.hk == MiPlan::H_INIT_J0 ? 3 : 2));
	fp.hrow = pl.hk == MiPlan::H_CURR_JM ? 2 : (pl.hk == MiPlan::H_INIT_J0 ? 1 : 0);
	fp.need_dft = !pl.iclk; fp.need_df0 = !(pl.fclk || pl.orig_jac); fp.g_mean = pl.orig_jac;
	const bool need_j0 = fp.need_df0 || fp.g_mean || fp.hrow != 0;
	const MiJ0Rebuild rb = mi_j0_rebuild(b);
	fp.j0_mode = !need_j0 ? 0 : (rb.dI0 ? 1 : 2);
	fp.j0_init_variant = rb.init_variant;
	fp.grad_eps = b->desc.grad_eps; fp.norm_mult = b->norm_mult; fp.norm_add = b->norm_add; fp.hist_norm = b->mi_hist_norm;
	{
		/* (partition of unity, MI.cc:80-94: pixel values are mapped to [1, n_bins - 2], so every cubic B-spline window lies inside the bins and its
		 * weights sum to one: the histogram of It is the joint histogram's row sum to rounding -- MTFHIP_MI_HIST_ROWSUM=0: its own block product) */
		const char *e_rs = std::getenv("MTFHIP_MI_HIST_ROWSUM");   /* (read per plan: the parity test flips it) */
		const bool rowsum_env = !(e_rs && e_rs[0] == '0');
		fp.hist_from_joint = (rowsum_env && b->desc.mi_partition_of_unity) ? 1 : 0;
	}
	fp.active = active; fp.tb = b->d_mi_tb;
	return fp;
}
static int mi_gmode(const MiPlan &pl) { return pl.iclk ? 0 : (pl.fclk ? 1 : (pl.orig_jac ? 2 : 3)); }
/* enqueues pass 1, the tables, pass 2 and the finish; do_track: the finish also solves, updates and tests convergence */
/* sec_ord_hess: sum_p df_dI(p) d2I_dp2(p) with MI's own per-pixel gradients / self gradient factor, from this iteration's tables
 * (d_mi_tb: filled by launch_mi_tables_iter in both forms of the iteration) -- one more pixel pass into d_d2_out */
static void mi_second_order(mtfhip_batch *b, const mtfhip_sm_desc *sm, int own_pts) {
	TimedScope tsc(b->ctx, "second_order");
	const int nb2 = simple_blocks_per_target(b->N);
	launch_second_order_ssd(b->view(), b->ctx->img, second_order_term(sm, MTFHIP_AM_MI), sm->chained_warp ? 1 : 0, b->d0_variant, b->desc.grad_eps,
		b->hess_eps, b->norm_mult, b->norm_add, b->d_d2_part, nb2, b->d_d2_out, b->ctx->stream, own_pts, SecondOrderNcc{nullptr, 0, nullptr},
		SecondOrderMi{b->d_mi_tb, b->mi_hist_norm});
}
/* so_own_pts: -1 no second-order term; 1 inside the device loop (points re-derived from the warp), 0 from CURR_PTS (iterate) */
static int mi_enqueue_fast(mtfhip_batch *b, const mtfhip_sm_desc *sm, const MiPlan &pl, const int *active, const TrackState &ts, int do_track,
	int so_own_pts = -1) {
	const int nblk = mi_blocks(b);
	hipStream_t st = b->ctx->stream;
	MiFastPlan fp = mi_fast_plan(b, pl, active, sm);
	if (!b->d_mi_poly) HIP_TRY(hipMalloc(&b->d_mi_poly, sizeof(double) * (size_t)mi_poly_size(b->desc.mi_n_bins) * b->B));
	fp.poly = b->d_mi_poly;
	const BatchView bv = b->view();
	/* pass 1 holds 45.7 KB of LDS per workgroup: THREE workgroups per CU, so mi_blocks' ~4 per CU ran as one full round and a second one
	 * at a third of the occupancy (1024 workgroups over 768 slots; r05 ablation: the pass is bound by its sampling, 105 of 120 us, not
	 * by the block products).  Its own count: the largest multiple of the resident slots that the partial-row buffer holds. */
	/* (r05 advisor: nblk1 follows the device's resident slots, so tolerance mode's summation grouping -- and with it the last bits of its sums --
	 * depends on the CU count: results are reproducible run to run on one device, not bit for bit across devices; MTFHIP_MI_PASS1_BLOCKS pins it) */
	int nblk1 = nblk;
	{
		static const char *e_b1 = std::getenv("MTFHIP_MI_PASS1_BLOCKS");
		const int slots = 3 * std::max(b->ctx->n_cus, 1);
		if (e_b1) nblk1 = std::min(nblk, std::max(1, std::atoi(e_b1)));
		else if ((long)nblk * b->B > slots && slots / b->B >= 1) nblk1 = std::min(nblk, slots / b->B);
	}
	{
		TimedScope tsc(b->ctx, "mi_pass1");
		launch_mi_pass_hist(bv, b->ctx->img, fp, b->d_mi_part, nblk1, b->mi_row_len, st);
	}
	/* (the dense Hessian kinds read the tables themselves: no polynomial tables) */
	if (fp.hk <= 1) launch_mi_tables_poly(bv, fp.nb, b->desc.mi_pre_seed, b->mi_hist_norm, fp.hk == 1 ? 1 : 0, b->d_mi_part, nblk1, b->mi_row_len, b->d_mi_tb, b->d_mi_f, b->d_mi_poly, st);
	else launch_mi_tables_iter(bv, fp.nb, b->desc.mi_pre_seed, b->mi_hist_norm, fp.hk == 1 ? 1 : 0, b->d_mi_part, nblk1, b->mi_row_len, b->d_mi_tb, b->d_mi_f, st);
	{
		TimedScope tsc(b->ctx, "mi_pass2");
		launch_mi_pass_grad_hess(bv, b->ctx->img, fp, b->d_mi_part, nblk, st);
	}
	if (so_own_pts >= 0) mi_second_order(b, sm, so_own_pts);
	launch_mi_finish_fast(bv, *sm, ts, fp, mi_gmode(pl), do_track, b->d_mi_part, nblk, b->d_mi_H, b->d_mi_H + 64 * (size_t)b->B, b->d_mi_red, st);
	b->it_valid = b->dit_valid = b->jt_valid = false;
	return MTFHIP_OK;
}
static int mi_iterate(mtfhip_batch *b, const mtfhip_sm_desc *sm, double *f, double *g, double *H) {
	const int S = b->S;
	hipStream_t st = b->ctx->stream;
	const MiPlan pl(sm);
	const int term = second_order_term(sm, MTFHIP_AM_MI);
	const int S2 = S * S;
	std::vector<double> so;
	if (term >= 0) {
		if (b->desc.mi_n_bins != 8) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "iterate: second-order MI Hessians with other than 8 bins go through the per-function entry points");
		if (term != 0 && term != 4 && !b->init_pix_hess) return fail(MTFHIP_ERR_LOGIC, "iterate: init_template was run without sec_ord_hess");
		TRY(ensure_pts(b));
		if (!b->d_d2_part) {
			const int nb2 = simple_blocks_per_target(b->N);
			HIP_TRY(hipMalloc(&b->d_d2_part, sizeof(double) * 64 * (size_t)nb2 * b->B));
			HIP_TRY(hipMalloc(&b->d_d2_out, sizeof(double) * 64 * (size_t)b->B));
		}
	}
	if (mi_fast_ok(b, sm, pl)) {
		TrackState ts{b->d_acc, b->d_h0, b->d_corners, b->d_init_corners_hm, b->d_active, b->d_iters, nullptr, nullptr, 1, nullptr, nullptr};
		TRY(mi_enqueue_fast(b, sm, pl, nullptr, ts, 0, term >= 0 ? 0 : -1));
	} else {
		TRY(mi_enqueue(b, sm, pl, nullptr));
		if (term >= 0) mi_second_order(b, sm, 0);
	}
	if (term >= 0) {
		so.resize((size_t)S2 * b->B);
		HIP_TRY(hipMemcpyAsync(so.data(), b->d_d2_out, sizeof(double) * so.size(), hipMemcpyDeviceToHost, st));
	}
	const size_t B = (size_t)b->B;
	std::vector<double> out(B * 145);
	HIP_TRY(hipMemcpyAsync(out.data(), b->d_mi_H, sizeof(double) * (pl.hk == MiPlan::H_SUM_STD ? 144 : 80) * B, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(out.data() + 144 * B, b->d_mi_f, sizeof(double) * B, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	for (int t = 0; t < b->B; ++t) {
		const double *Hs = &out[64 * (size_t)t], *gs = &out[64 * B + 16 * (size_t)t], *H2 = &out[80 * B + 64 * (size_t)t];
		TargetHost &h = b->th[t];
		h.f = out[144 * B + t];
		if (f) f[t] = h.f;
		double *gt = g + (size_t)t * S, *Ht = H + (size_t)t * S * S;
		for (int s = 0; s < S; ++s) gt[s] = pl.iclk ? gs[8 + s] : ((pl.fclk || pl.orig_jac) ? gs[s] : 0.5 * (gs[s] - gs[8 + s]));
		for (int k = 0; k < S * S; ++k) {
			const int r = k % S, c = k / S;
			const double hv = Hs[c * S + r];   /* k_mi_hess_finish writes column-major S x S */
			Ht[k] = pl.hk == MiPlan::H_CONST ? h.h0[k]
				: (pl.esm && sm->hess_type == 2) ? 0.5 * (hv + h.h0[k])
				: pl.hk == MiPlan::H_SUM_STD ? 0.5 * (hv + H2[c * S + r])
				: hv;
			if (term >= 0) Ht[k] += ((term == 1 || (pl.esm && sm->hess_type == 2)) ? 0.5 : 1.0) * so[(size_t)t * S2 + k];   /* (k_plane_sum_finish: entry (r, c) at c S + r, as Ht) */
		}
	}
	return MTFHIP_OK;
}

int mtfhip_batch_iterate(mtfhip_batch *b, const mtfhip_sm_desc *sm, double *f, double *g, double *H) {
	FLUSH_AM(b);   /* the fused kernels derive the sample points from the warp: CURR_PTS may stay stale */
	if (b) { touch_all(b); b->lz.it_epoch = -1; TRY(ensure_df(b)); }
	TRY(check_sm(b, sm, "iterate"));
	TRY(fused_channels_ok(b, "iterate"));
	if (b->C != 1 && sm->sec_ord_hess) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "iterate: second-order Hessians of the multi-channel models use the per-function entry points");
	if (!g || !H) return fail(MTFHIP_ERR_INVALID_ARG, "iterate: NULL output");
	if (!b->init_pix_vals) return fail(MTFHIP_ERR_LOGIC, "iterate before init_template");
	TRY(need_image(b));
	if (b->desc.am == MTFHIP_AM_MI) return mi_iterate(b, sm, f, g, H);
	if (second_order_term(sm) >= 0) TRY(ensure_pts(b));   /* k_second_order_ssd reads the current points */
	FusedArgs fa;
	TRY(fused_args(b, sm, fa));
	int nblk = fused_blocks_per_target(b->N, b->B);
	{
		TimedScope ts(b->ctx, "fused_lk");
		launch_fused_ssd(fused_view(b, fa), b->ctx->img, fa, b->d_partials, nblk, b->ctx->stream);
	}
	b->it_valid = fa.materialize;
	b->dit_valid = fa.materialize && fa.mode != 2;
	b->jt_valid = fa.materialize && fa.mode != 2;
	const bool ncc = b->desc.am == MTFHIP_AM_NCC;
	const int term = second_order_term(sm);
	const int S2 = b->S * b->S;
	std::vector<double> so;
	if (term >= 0) {
		if (term != 0 && !b->init_pix_hess) return fail(MTFHIP_ERR_LOGIC, "iterate: init_template was run without sec_ord_hess");
		const int nb2 = simple_blocks_per_target(b->N);
		if (!b->d_d2_part) {
			HIP_TRY(hipMalloc(&b->d_d2_part, sizeof(double) * 64 * (size_t)nb2 * b->B));
			HIP_TRY(hipMalloc(&b->d_d2_out, sizeof(double) * 64 * (size_t)b->B));
		}
		if (ncc) TRY(push_ncc(b));   /* mean(I0), |I0 - mean| of the template */
		{
			TimedScope ts(b->ctx, "second_order");
			launch_second_order_ssd(b->view(), b->ctx->img, term, fa.chained, b->d0_variant, fa.grad_eps, b->hess_eps, b->norm_mult,
				b->norm_add, b->d_d2_part, nb2, b->d_d2_out, b->ctx->stream, 0,
				ncc ? SecondOrderNcc{b->d_partials, nblk, b->d_ncc} : SecondOrderNcc{nullptr, 0, nullptr});
		}
		so.resize((size_t)S2 * b->B);
		HIP_TRY(hipMemcpyAsync(so.data(), b->d_d2_out, sizeof(double) * so.size(), hipMemcpyDeviceToHost, b->ctx->stream));
	}
	if (ncc) {
		TRY(read_rows(b, nblk, NCC_ACC_COUNT));
		for (int t = 0; t < b->B; ++t) {
			double ft;
			double *Ht = H + (size_t)t * S2;
			TRY(ncc_assemble(b, sm, fa.hess_mean != 0, b->h_acc + (size_t)t * NCC_ACC_COUNT, b->th[t], &ft, g + (size_t)t * b->S, Ht));
			if (term >= 0) {   /* (NT/ESM.cc:339 halves the whole SumOfStd sum) */
				const double sc = term == 1 ? 0.5 : 1.0;
				for (int k = 0; k < S2; ++k) Ht[k] += sc * so[(size_t)t * S2 + k];
			}
			if (f) f[t] = ft;
		}
		b->ncc_host_newer = true;
		return MTFHIP_OK;
	}
	TRY(read_acc(b, nblk));
	for (int t = 0; t < b->B; ++t) {
		double ft;
		double *Ht = H + (size_t)t * S2;
		assemble(b, sm, b->h_acc + (size_t)t * ACC_COUNT, b->th[t].h0, &ft, g + (size_t)t * b->S, Ht);
		if (term >= 0) {   /* SumOfStd halves the whole sum (NT/ESM.cc:339) */
			const double sc = term == 1 ? 0.5 : 1.0;
			for (int k = 0; k < S2; ++k) Ht[k] += sc * so[(size_t)t * S2 + k];
		}
		b->th[t].f = ft;
		if (f) f[t] = ft;
	}
	return MTFHIP_OK;
}

/* Targets per launch of the device-side loop.  Chunking pays where an iteration both re-reads a large constant operand
 * set and writes as much again (ESM with materialisation: 88 B/px read, 88 B/px written): +15-17 % at B = 128-256.
 * FCLK reads only 24 B/px (fits anyway) and the lean / ICLK variants barely write, so for them a chunk only multiplies
 * the per-iteration finish launches (measured 7-20 % slower) and they keep one launch for all targets.
 * MTFHIP_TRACK_CHUNK_PX overrides the pixel budget (tests force tiny chunks with it, in every mode). */
static int track_chunk(const mtfhip_batch *b, const mtfhip_sm_desc *sm, const FusedArgs &fa) {
	const char *env_px = std::getenv("MTFHIP_TRACK_CHUNK_PX");
	if (!env_px && !(fa.mode == 1 && fa.materialize)) return b->B;
	const double chunk_px = env_px ? std::atof(env_px) : 2.6e6;
	int chunk = (int)(chunk_px / (double)b->N);
	if (chunk < 1) chunk = 1;
	if (chunk >= b->B || sm->max_iters == 1) return b->B;
	const int n_chunks = (b->B + chunk - 1) / chunk;
	return (b->B + n_chunks - 1) / n_chunks;   /* balanced: 100 targets -> 50 + 50, not 65 + 35 */
}
/* Queues of the device-side loop.  Two for the launches that materialise the interface arrays (HBM-bound: ESM / FCLK full mode, NCC,
 * the multi-channel models): the solve + update of one chunk of targets -- one-wave workgroups, a 6.5 us chain of dependent
 * latencies -- and the fill / drain of its pixel pass then run under the other chunk's pixel pass.  Measured at 200 x 200 x 64 (one
 * call): 61.5 -> 49-52 us per step in calls of >= 100 iterations, 64.5 -> 59-63 at 20; the lean / ICLK launches (issue-bound) gain
 * 0-4 %, small patches lose (50 x 50: -7 %): they keep one queue.  MTFHIP_TRACK_STREAMS=1 selects the single queue, 3 / 4 more
 * queues (measured slower), 12 two queues for every launch kind (A/B knob). */
static int track_queues(const mtfhip_batch *b, const FusedArgs &fa) {
	const char *e_want = std::getenv("MTFHIP_TRACK_STREAMS");   /* (read per call: the tests switch it) */
	const int want = e_want ? std::atoi(e_want) : 2;
	if (want < 2 || b->B < 2 || b->d_trace) return 1;
	/* launches that write nothing are issue-bound: ESM's lean pass gains 3-6 % in 200-iteration calls and loses 4-5 % in 20-iteration
	 * ones, FCLK's and ICLK's gain nothing */
	if (!fa.materialize && want < 12) return 1;
	/* small passes are launch- and latency-sized, not bandwidth-sized: 8 x 200 x 200 and 64 x 50 x 50 measured 5-14 % slower on two queues
	 * in 20-iteration calls, 16 / 32 / 48 x 200 x 200 6-22 % faster */
	static const double min_rows = std::getenv("MTFHIP_TRACK_STREAMS_MIN_ROWS") ? std::atof(std::getenv("MTFHIP_TRACK_STREAMS_MIN_ROWS")) : 0.5e6;
	if ((double)b->B * b->N < min_rows && want < 12) return 1;
	return std::min(std::min(want % 10, 4), b->B);
}
int mtfhip_batch_track_targets_per_launch(mtfhip_batch *b, const mtfhip_sm_desc *sm) {
	FLUSH(b);
	if (check_sm(b, sm, "track_targets_per_launch") != MTFHIP_OK) return 0;
	const bool one_launch = iclk_one_launch(b, sm);
	if (one_launch) return b->B;
	FusedArgs fa;
	if (fused_args(b, sm, fa) != MTFHIP_OK) return 0;
	const int chunk = track_chunk(b, sm, fa), nq = track_queues(b, fa);
	return nq >= 2 ? std::min(chunk, (b->B + nq - 1) / nq) : chunk;
}
int mtfhip_batch_track_queues(mtfhip_batch *b, const mtfhip_sm_desc *sm) {
	FLUSH(b);
	if (check_sm(b, sm, "track_queues") != MTFHIP_OK) return 0;
	if (b->desc.am == MTFHIP_AM_MI) return 1;
	const bool one_launch = iclk_one_launch(b, sm);
	if (one_launch) return 1;
	FusedArgs fa;
	if (fused_args(b, sm, fa) != MTFHIP_OK) return 0;
	return track_queues(b, fa);
}

static int track_core(mtfhip_batch *b, const mtfhip_sm_desc *sm, int *n_iters, double *corners, bool slab_uploaded, bool resume = false, bool region_mode = false);
static int track_validate(mtfhip_batch *b, const mtfhip_sm_desc *sm);
/* ---- the persistent one-launch loop (kernels_persist.hip) ---- */
/* rows per workgroup so that every target's workgroups are resident together: the default decomposition when it fits, else the
 * smallest number of rows that does */
static void persist_decomposition(const mtfhip_batch *b, int &nblk, int &rows) {
	fused_decomposition(b->N, b->B, nblk, rows);
	const int per_target = b->ctx->n_cus / b->B;
	if (per_target >= 1 && nblk > per_target) {
		const int total_rows = (b->N + kBlock - 1) / kBlock;
		rows = (total_rows + per_target - 1) / per_target;
		nblk = (total_rows + rows - 1) / rows;
	}
	static const int forced = std::getenv("MTFHIP_PERSIST_NBLK") ? std::atoi(std::getenv("MTFHIP_PERSIST_NBLK")) : 0;   /* experiments */
	if (forced > 0 && forced < nblk) {
		const int total_rows = (b->N + kBlock - 1) / kBlock;
		rows = (total_rows + forced - 1) / forced;
		nblk = (total_rows + rows - 1) / rows;
	}
}
static unsigned long long persist_timeout_ticks() {   /* 100 MHz ticks; MTFHIP_PERSIST_TIMEOUT_US for tests (default 20 ms) */
	const char *e = std::getenv("MTFHIP_PERSIST_TIMEOUT_US");
	const double us = e ? std::atof(e) : 20000.0;
	return (unsigned long long)(us * 100.0);
}
/* Opt-in (MTFHIP_PERSIST=1).  Measured on MI355X (profiles/README.md, r02): a hand-over between workgroups through memory costs what
 * the gap between two dependent launches costs (~2 us), so one launch per loop does not beat two launches per iteration --
 * 200 x 200 x 1: 17.1 us per iteration against 13.0, 50 x 50 x 1: 12.3 against 12.3 -- and the per-iteration time is the solve's
 * latency either way. */
static bool persist_fits(const mtfhip_batch *b, const mtfhip_sm_desc *sm, const FusedArgs &fa) {
	const char *e = std::getenv("MTFHIP_PERSIST");
	if (!(e && e[0] == '1') || !b->persist_ok || fa.materialize || b->ctx->n_cus <= 0 || b->B > b->ctx->n_cus || !b->h_pub_dev) return false;
	if (b->C != 1) return false;   /* (no multi-channel instantiation of the persistent kernel) */
	if (b->B > 8) return false;   /* a batch is better served by its own decomposition (eight workgroups per target) */
	if (sm->max_iters < 2) return false;
	int nblk, rows;
	persist_decomposition(b, nblk, rows);
	return (long)nblk * b->B <= b->ctx->n_cus && nblk <= b->nblk_max;
}
/* the rest of a loop the persistent launch left unfinished: the slab on the device is current (warps, flags, iteration counts) */
static int track_resume(mtfhip_batch *b, const mtfhip_sm_desc *sm, int *n_iters, double *corners) { return track_core(b, sm, n_iters, corners, true, true); }
int mtfhip_batch_track(mtfhip_batch *b, const mtfhip_sm_desc *sm, int *n_iters, double *corners) { return track_core(b, sm, n_iters, corners, false); }

/* setRegion + update of one frame in one call: what GridTracker::update does with every patch tracker (GridTracker.cc:345-363:
 * tracker->setRegion(patch corners); tracker->update()) and a pyramid level with the level above's result.  For the search
 * methods that keep their template Jacobian (ICLK; FCLK without the InitialSelf Hessian) the reset state and the loop's
 * active flags / iteration counts travel in ONE staged copy; the others take the two steps one after the other. */
/* MTFHIP_TRACK_DEBUG_TIMING: host-side stamps of a one-launch frame (before the launch call | after it | after the deferred host half) */
static const bool g_track_dbg_timing = std::getenv("MTFHIP_TRACK_DEBUG_TIMING") != nullptr;
static thread_local std::chrono::steady_clock::time_point g_track_dbg_t[3];
/* grid != NULL (mtfhip_grid_frame): region_corners is the GRID's region (8 doubles) and the patches are laid over it -- by the kernel
 * itself where the one-launch region mode applies and the patches are fixed-size rectangles (the host layout then runs behind the launch),
 * by mtfhip_grid_layout in front of the call otherwise */
static int track_region_impl(mtfhip_batch *b, const mtfhip_sm_desc *sm, const double *region_corners, int *n_iters, double *corners, const mtfhip_grid_desc *grid) {
	if (!b || !sm || !region_corners) return fail(MTFHIP_ERR_INVALID_ARG, "track_region: NULL argument");
	/* everything track_core would refuse is refused before the SSM is reset (the folded upload reads the pinned staging buffer
	 * without an event guard: the loop that follows is what the host waits for) */
	TRY(track_validate(b, sm));
	const bool folded = !region_refreshes(sm);
	static const bool dbg = std::getenv("MTFHIP_TRACK_DEBUG_TIMING") != nullptr;
	/* r04: in front of the one-launch ICLK kernel (the grid tracker's patches) the reset needs no launch of its own -- every workgroup
	 * ingests its patch's corners from the pinned staging buffer and lays out its own grid (RegionIngest, k_iclk_track).
	 * MTFHIP_GRID_FUSED=0 keeps the ingest + k_init_grid launch in front of the loop (A/B, and the bit-identity test). */
	const char *e_gf = std::getenv("MTFHIP_GRID_FUSED");   /* (read per call: the A/B test flips it) */
	const bool fused_ok = !(e_gf && e_gf[0] == '0');
	const bool region_mode = fused_ok && folded && b->desc.am != MTFHIP_AM_MI && !sm->leven_marq && iclk_one_launch(b, sm) && second_order_term(sm, b->desc.am) < 0 &&
		b->h_stage_a_dev && b->h_pub_dev;
	const auto t0 = std::chrono::steady_clock::now();
	static thread_local std::vector<double> patches;
	bool layout_later = false;
	if (grid) {
		const char *e_ld = std::getenv("MTFHIP_GRID_LAYOUT_DEV");   /* (=0: the host lays the patches out in front of the launch, the r05 first form) */
		layout_later = region_mode && b->desc.ssm != MTFHIP_SSM_HOMOGRAPHY && !grid->dyn_patch_size && !(e_ld && e_ld[0] == '0');
		if (layout_later) {
			M3 Wr;
			if (!rect_to_quad(-0.5, -0.5, 0.5, 0.5, region_corners, Wr)) return fail(MTFHIP_ERR_INVALID_ARG, "grid_layout: degenerate region corners");
			b->deferred_gdesc = *grid;
			std::memcpy(b->deferred_region, region_corners, sizeof(b->deferred_region));
			std::memcpy(b->deferred_region_map, Wr.m, sizeof(b->deferred_region_map));
		} else {
			patches.resize(8 * (size_t)b->B);
			TRY(mtfhip_grid_layout(grid, region_corners, nullptr, patches.data()));
			region_corners = patches.data();
		}
	}
	{
		const int rs = set_region_core(b, layout_later ? nullptr : region_corners, sm, folded, region_mode, layout_later);
		if (rs != MTFHIP_OK) { b->deferred_layout = false; b->deferred_template_check = false; set_corners_finish_deferred(b); return rs; }
	}
	const auto t1 = std::chrono::steady_clock::now();
	const int r = track_core(b, sm, n_iters, corners, folded, false, region_mode);
	set_corners_finish_deferred(b);   /* (a call that failed before its launch: nothing stays pending on the caller's buffer) */
	if (dbg) {
		const auto t2 = std::chrono::steady_clock::now();
		static double acc1 = 0, acc2 = 0, acc3 = 0, acc4 = 0, acc5 = 0, acc6 = 0; static int n = 0;
		auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::micro>(c - a).count(); };
		acc1 += us(t0, t1); acc2 += us(t1, t2); acc3 += us(t1, g_track_dbg_t[0]); acc4 += us(g_track_dbg_t[0], g_track_dbg_t[1]); acc5 += us(g_track_dbg_t[1], g_track_dbg_t[2]); acc6 += us(g_track_dbg_t[2], t2);
		if (++n % 100 == 0) {
			std::fprintf(stderr, "[track_region] set_region %.1f us, track %.1f us = before the launch %.1f + launch call %.1f + deferred host half %.1f + wait and copy-out %.1f (mean of 100)\n",
				acc1 / 100, acc2 / 100, acc3 / 100, acc4 / 100, acc5 / 100, acc6 / 100);
			acc1 = acc2 = acc3 = acc4 = acc5 = acc6 = 0;
		}
	}
	return r;
}
int mtfhip_batch_track_region(mtfhip_batch *b, const mtfhip_sm_desc *sm, const double *region_corners, int *n_iters, double *corners) {
	return track_region_impl(b, sm, region_corners, n_iters, corners, nullptr);
}

/* GridTracker::update's patch half as ONE call (SM/src/GridTracker.cc:345-363): every patch tracker is reset to its region and
 * runs its update(); regions and corners in the reference's CornersT layout as a row-major host array sees it (2 x 4: the x row,
 * then the y row), plus the patch centroids utils::getCentroid (miscUtils.h:473-480: the mean of the four corners) hands to the
 * robust estimator.  (The layout conversion and the centroids were ~9 of the ~14 us a frame spent in the Python layer.) */
int mtfhip_grid_update(mtfhip_batch *b, const mtfhip_sm_desc *sm, const double *regions_2x4, int *n_iters, double *corners_2x4, double *centroids) {
	if (!b || !sm || !regions_2x4) return fail(MTFHIP_ERR_INVALID_ARG, "grid_update: NULL argument");
	const size_t B = (size_t)b->B;
	static thread_local std::vector<double> in, out;
	in.resize(8 * B); out.resize(8 * B);
	for (size_t t = 0; t < B; ++t)
		for (int q = 0; q < 4; ++q) { in[8 * t + 2 * q] = regions_2x4[8 * t + q]; in[8 * t + 2 * q + 1] = regions_2x4[8 * t + 4 + q]; }
	TRY(mtfhip_batch_track_region(b, sm, in.data(), n_iters, out.data()));
	for (size_t t = 0; t < B; ++t) {
		const double *c = &out[8 * t];
		if (corners_2x4)
			for (int q = 0; q < 4; ++q) { corners_2x4[8 * t + q] = c[2 * q]; corners_2x4[8 * t + 4 + q] = c[2 * q + 1]; }
		if (centroids) { centroids[2 * t] = (c[0] + c[2] + c[4] + c[6]) * 0.25; centroids[2 * t + 1] = (c[1] + c[3] + c[5] + c[7]) * 0.25; }
	}
	return MTFHIP_OK;
}

/* utils::getCentroid(cv::Point2f &, corners) miscUtils.h:472-480: the mean of the four corners, rounded to float */
static inline void centroid_f(float *dst, const double *c) {
	dst[0] = static_cast<float>((c[0] + c[2] + c[4] + c[6]) / 4.0);
	dst[1] = static_cast<float>((c[1] + c[3] + c[5] + c[7]) / 4.0);
}
static int grid_batch_ok(const mtfhip_batch *b, const mtfhip_grid_desc *g, const char *fn) {
	if (!b || !g) return fail(MTFHIP_ERR_INVALID_ARG, "%s: NULL argument", fn);
	if (g->grid_size_x <= 0 || g->grid_size_y <= 0 || g->grid_size_x * g->grid_size_y != b->B)   /* GridTracker.cc:124-129 */
		return fail(MTFHIP_ERR_INVALID_ARG, "%s: mismatch between the grid dimensions (%d x %d) and the batch's %d patch trackers", fn, g->grid_size_x, g->grid_size_y, b->B);
	return MTFHIP_OK;
}
/* every patch tracker's update() behind a reset that may still be running (mtfhip_grid_frame without a region; the backward pass) */
static int grid_track_plain(mtfhip_batch *b, const mtfhip_sm_desc *sm, int *n_iters, double *out) {
	/* reset-every-frame mode: this call follows mtfhip_grid_reset(reinit), whose k_template_init may still be running.  The one-launch
	 * loop kernel reads nothing of that kernel's host record, so it is enqueued behind it right away (r05: the host used to wait for the
	 * record and copy 1 KB per patch first -- launch latency + 256 KB of memcpy exposed in every frame); the record is folded into the
	 * mirrors by the next call that flushes without this flag.  MTFHIP_GRID_HOLD_PULL=0: the r05 first form. */
	const size_t B = (size_t)b->B;
	const char *e_hp = std::getenv("MTFHIP_GRID_HOLD_PULL");
	const bool hold = b->init_mirror_seq != 0 && !(e_hp && e_hp[0] == '0') && b->h_stage_b_dev && b->h_pub_dev && b->desc.am != MTFHIP_AM_MI && !sm->leven_marq &&
		iclk_one_launch(b, sm) && second_order_term(sm, b->desc.am) < 0 && ((37 * sizeof(double) * B) % 16) == 0;
	b->hold_init_pull = hold;
	const int rc = mtfhip_batch_track(b, sm, n_iters, out);
	b->hold_init_pull = false;
	return rc;
}
/* GridTracker::update's patch loop (GridTracker.cc:254-261), with the reset that preceded it folded in when a region is given */
int mtfhip_grid_frame(mtfhip_batch *b, const mtfhip_sm_desc *sm, const mtfhip_grid_desc *g, const double *region, int *n_iters, double *corners, float *centroids) {
	if (!sm) return fail(MTFHIP_ERR_INVALID_ARG, "grid_frame: NULL argument");
	TRY(grid_batch_ok(b, g, "grid_frame"));
	const size_t B = (size_t)b->B;
	static thread_local std::vector<double> out;
	static thread_local std::vector<int> iters;
	out.resize(8 * B); iters.resize(B);
	if (region) TRY(track_region_impl(b, sm, region, n_iters ? n_iters : iters.data(), out.data(), g));
	else TRY(grid_track_plain(b, sm, n_iters ? n_iters : iters.data(), out.data()));
	if (corners) std::memcpy(corners, out.data(), sizeof(double) * 8 * B);
	if (centroids) for (size_t t = 0; t < B; ++t) centroid_f(centroids + 2 * t, &out[8 * t]);
	return MTFHIP_OK;
}

/* ---- forward-backward error estimation (GridTracker.cc:186-190, 263-266, 294-343) ---- */
/* the mask half of backwardEstimation (:307-332): host arithmetic, no device */
int mtfhip_grid_fb_mask(int n, const float *prev_pts, const float *curr_pts, const float *fb_prev_pts, const mtfhip_grid_fb_desc *fb,
	unsigned char *fb_err_mask, float *prev_masked, float *curr_masked, int *n_masked) {
	if (n < 0 || !prev_pts || !curr_pts || !fb_prev_pts || !fb || !fb_err_mask || !n_masked) return fail(MTFHIP_ERR_INVALID_ARG, "grid_fb_mask: NULL argument");
	int cnt = 0;
	auto keep = [&](int id) {
		if (prev_masked) { prev_masked[2 * cnt] = prev_pts[2 * id]; prev_masked[2 * cnt + 1] = prev_pts[2 * id + 1]; }
		if (curr_masked) { curr_masked[2 * cnt] = curr_pts[2 * id]; curr_masked[2 * cnt + 1] = curr_pts[2 * id + 1]; }
		++cnt;
	};
	for (int id = 0; id < n; ++id) {
		/* cv::Point2f members: the difference is a float, the squares and their sum doubles (:309-312) */
		const float dxf = fb_prev_pts[2 * id] - prev_pts[2 * id], dyf = fb_prev_pts[2 * id + 1] - prev_pts[2 * id + 1];
		const double dx = dxf, dy = dyf;
		if (dx * dx + dy * dy > fb->fb_err_thresh) fb_err_mask[id] = 0;
		else { fb_err_mask[id] = 1; keep(id); }
	}
	if (cnt < fb->n_model_pts) {   /* :321-332: filled up in tracker order to what the estimator needs */
		for (int id = 0; id < n; ++id) {
			if (fb_err_mask[id]) continue;
			keep(id);
			fb_err_mask[id] = 1;
			if (cnt == fb->n_model_pts) break;
		}
	}
	*n_masked = cnt;
	return MTFHIP_OK;
}
/* the patch half of backwardEstimation (:295-306) for every patch tracker of the batch at once: re-initialised at its tracked location on the
 * current frame (fb_reinit), run on the PREVIOUS frame (mtfhip_image_keep_prev), centroid of where it arrives, then back on the current
 * frame and setRegion(location) */
static int grid_backward_impl(mtfhip_batch *b, const mtfhip_sm_desc *sm, const mtfhip_grid_desc *g, const mtfhip_grid_fb_desc *fb, int *n_iters, double *fb_corners,
	float *fb_prev_pts, bool restore);
int mtfhip_grid_backward(mtfhip_batch *b, const mtfhip_sm_desc *sm, const mtfhip_grid_desc *g, const mtfhip_grid_fb_desc *fb, int *n_iters, double *fb_corners,
	float *fb_prev_pts) {
	return grid_backward_impl(b, sm, g, fb, n_iters, fb_corners, fb_prev_pts, true);
}
static int grid_backward_impl(mtfhip_batch *b, const mtfhip_sm_desc *sm, const mtfhip_grid_desc *g, const mtfhip_grid_fb_desc *fb, int *n_iters, double *fb_corners,
	float *fb_prev_pts, bool restore) {
	if (!sm || !fb) return fail(MTFHIP_ERR_INVALID_ARG, "grid_backward: NULL argument");
	TRY(grid_batch_ok(b, g, "grid_backward"));
	if (!b->init_pix_vals) return fail(MTFHIP_ERR_LOGIC, "grid_backward before the patch trackers were initialised");
	mtfhip_ctx *c = b->ctx;
	if (!c->prev.data) return fail(MTFHIP_ERR_LOGIC, "grid_backward: no previous image (mtfhip_image_keep_prev)");
	if (c->prev.h != c->img.h || c->prev.w != c->img.w || c->prev.channels != c->img.channels)
		return fail(MTFHIP_ERR_INVALID_ARG, "grid_backward: the previous image is %dx%dx%d, the current one %dx%dx%d", c->prev.h, c->prev.w, c->prev.channels, c->img.h, c->img.w, c->img.channels);
	TRY(track_validate(b, sm));
	const size_t B = (size_t)b->B;
	static thread_local std::vector<double> loc, out;
	static thread_local std::vector<int> iters;
	loc.resize(8 * B); out.resize(8 * B); iters.resize(B);
	FLUSH(b);
	for (size_t t = 0; t < B; ++t) std::memcpy(&loc[8 * t], b->th[t].corners, sizeof(double) * 8);   /* tracker_location = getRegion().clone() :296 */
	if (fb->fb_reinit) {                                                                               /* tracker->initialize(tracker_location) :297-299 */
		const char *e_gf = std::getenv("MTFHIP_GRID_FUSED");
		const bool fused = !(e_gf && e_gf[0] == '0') && b->h_stage_a_dev && template_init_fused_ok(b, sm);
		if (fused) TRY(grid_reinit_fused(b, sm, loc.data()));
		else {
			TRY(mtfhip_ssm_set_corners(b, loc.data()));
			TRY(mtfhip_batch_init_template(b, sm));
		}
	}
	TRY(mtfhip_image_swap_prev(c));                                                                    /* tracker->setImage(prev_img) :300 */
	const int rc = grid_track_plain(b, sm, n_iters ? n_iters : iters.data(), out.data());              /* tracker->update() :301 */
	const int rs = mtfhip_image_swap_prev(c);                                                          /* tracker->setImage(curr_img) :304 */
	if (rc != MTFHIP_OK) return rc;
	if (rs != MTFHIP_OK) return rs;
	if (fb_corners) std::memcpy(fb_corners, out.data(), sizeof(double) * 8 * B);
	if (fb_prev_pts) for (size_t t = 0; t < B; ++t) centroid_f(fb_prev_pts + 2 * t, &out[8 * t]);      /* getCentroid(fb_prev_pts[id], getRegion()) :302 */
	if (!restore) return MTFHIP_OK;
	return mtfhip_batch_set_region(b, loc.data(), sm);                                                 /* tracker->setRegion(tracker_location) :305 */
}
/* GridTracker::update's patch loop followed by backwardEstimation (:254-266): mtfhip_grid_frame, mtfhip_grid_backward and the mask in one call */
int mtfhip_grid_frame_fb(mtfhip_batch *b, const mtfhip_sm_desc *sm, const mtfhip_grid_desc *g, const mtfhip_grid_fb_desc *fb, const double *region,
	const float *prev_pts, int *n_iters, double *corners, float *centroids, float *fb_prev_pts, unsigned char *fb_err_mask, float *prev_masked, float *curr_masked,
	int *n_masked) {
	if (!fb || !prev_pts || !fb_prev_pts || !fb_err_mask || !n_masked) return fail(MTFHIP_ERR_INVALID_ARG, "grid_frame_fb: NULL argument");
	if (!(fb->fb_err_thresh > 0)) return fail(MTFHIP_ERR_INVALID_ARG, "grid_frame_fb: fb_err_thresh must be positive (GridTracker.cc:186: the estimation is off otherwise; use mtfhip_grid_frame)");
	TRY(grid_batch_ok(b, g, "grid_frame_fb"));
	static thread_local std::vector<float> cen;
	cen.resize(2 * (size_t)b->B);
	static const bool always_restore = std::getenv("MTFHIP_GRID_FB_RESTORE") && std::getenv("MTFHIP_GRID_FB_RESTORE")[0] == '1';
	/* The shipped configuration (reset_at_each_frame 1, fb_reinit 1; Config/modules.cfg:80-82) in ONE launch: k_grid_fb runs a patch's update(), its
	 * initialize(tracker_location) and its update() on the previous frame back to back in the patch's workgroup and leaves the trackers as the
	 * forward pass left them -- the caller's resetTrackers(reinit) (:273-274) re-initialises them next.  Tolerance mode, ICLK with a constant
	 * Hessian over SSD / NCC, an affine patch SSM (a tracked patch stays a parallelogram: a unit-z lattice), <= 1024 pixels.  MTFHIP_GRID_FB_FUSED=0:
	 * the three launches. */
	{
		const char *e_ff = std::getenv("MTFHIP_GRID_FB_FUSED");   /* (read per call: the tests compare the two forms in one process) */
		mtfhip_ctx *c = b->ctx;
		const bool reinit_ok = !fb->fb_reinit || (b->desc.ssm == MTFHIP_SSM_AFFINE && template_init_fused_ok(b, sm));   /* (fb_reinit 0: the backward loop keeps the forward pass's template and state) */
		/* reset_at_each_frame 1: the caller's reset follows, nothing to restore.  0 without fb_reinit: the template is untouched, setRegion(tracker_location)
		 * (:305) is one more call behind the launch.  (0 with fb_reinit would have to keep the backward template: the launch-by-launch form.) */
		const bool restore_after = g->reset_at_each_frame == 0 && !fb->fb_reinit;
		const bool fused = !(e_ff && e_ff[0] == '0') && !region && reinit_ok && (g->reset_at_each_frame == 1 || restore_after) && !always_restore &&
			b->math_mode == MTFHIP_MATH_FAST && (b->desc.am == MTFHIP_AM_SSD || b->desc.am == MTFHIP_AM_NCC) && b->C == 1 && sm->sm == MTFHIP_SM_ICLK && iclk_one_launch(b, sm) && !sm->leven_marq &&
			second_order_term(sm, b->desc.am) < 0 && b->N <= 4 * kBlock && b->h_pub_dev && !b->d_trace && b->init_pix_vals &&
			(b->desc.am != MTFHIP_AM_NCC || b->d_ncc_tm) && c->prev.data && c->img.data && c->prev.h == c->img.h && c->prev.w == c->img.w &&
			c->prev.channels == c->img.channels;
		if (fused) {
			const size_t B = (size_t)b->B;
			if (!b->h_fb) {
				HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&b->h_fb), sizeof(double) * 9 * B, hipHostMallocMapped));
				HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&b->h_fb_dev), b->h_fb, 0));
				HIP_TRY(hipMalloc(&b->d_fb, sizeof(double) * 9 * B));
			}
			b->fb_fused_req = true; b->fb_fused_reinit = fb->fb_reinit != 0;
			const int rc = mtfhip_grid_frame(b, sm, g, nullptr, n_iters, corners, cen.data());
			b->fb_fused_req = false;
			if (rc != MTFHIP_OK) return rc;
			if (centroids) std::memcpy(centroids, cen.data(), sizeof(float) * cen.size());
			for (size_t t = 0; t < B; ++t) {
				if (b->h_fb[9 * t + 8] < 0) return fail(MTFHIP_ERR_INVALID_ARG, "grid_frame_fb: degenerate tracked corners for patch %d", (int)t);
				centroid_f(fb_prev_pts + 2 * t, b->h_fb + 9 * t);                                        /* getCentroid(fb_prev_pts[id], getRegion()) :302 */
			}
			if (restore_after) {                                                                           /* tracker->setRegion(tracker_location) :305 */
				static thread_local std::vector<double> loc;
				loc.resize(8 * B);
				for (size_t t = 0; t < B; ++t) std::memcpy(&loc[8 * t], b->th[t].corners, sizeof(double) * 8);   /* (the forward pass's: what the kernel left) */
				TRY(mtfhip_batch_set_region(b, loc.data(), sm));
			}
			return mtfhip_grid_fb_mask(b->B, prev_pts, cen.data(), fb_prev_pts, fb, fb_err_mask, prev_masked, curr_masked, n_masked);
		}
	}
	TRY(mtfhip_grid_frame(b, sm, g, region, n_iters, corners, cen.data()));
	if (centroids) std::memcpy(centroids, cen.data(), sizeof(float) * cen.size());
	/* GridTracker::update goes on to resetTrackers when reset_at_each_frame != 0 (:273-274): every patch tracker is then initialize()d or
	 * setRegion()ed on the new grid, which replaces all that setRegion(tracker_location) (:305) would leave -- the SSM's state; with fb_reinit
	 * the template is the backward pass's either way -- so that call is left out here (MTFHIP_GRID_FB_RESTORE=1 keeps it) */
	TRY(grid_backward_impl(b, sm, g, fb, nullptr, nullptr, fb_prev_pts, always_restore || g->reset_at_each_frame == 0));
	return mtfhip_grid_fb_mask(b->B, prev_pts, cen.data(), fb_prev_pts, fb, fb_err_mask, prev_masked, curr_masked, n_masked);
}
/* GridTracker::resetTrackers(reinit) GridTracker.cc:345-392 */
int mtfhip_grid_reset(mtfhip_batch *b, const mtfhip_sm_desc *sm, const mtfhip_grid_desc *g, const double *region, int reinit, double *patch_corners, float *prev_pts) {
	if (!sm || !region) return fail(MTFHIP_ERR_INVALID_ARG, "grid_reset: NULL argument");
	TRY(grid_batch_ok(b, g, "grid_reset"));
	const size_t B = (size_t)b->B;
	static thread_local std::vector<double> patches;
	patches.resize(8 * B);
	const char *e_gf = std::getenv("MTFHIP_GRID_FUSED"), *e_ld = std::getenv("MTFHIP_GRID_LAYOUT_DEV");
	const bool fused = reinit && !(e_gf && e_gf[0] == '0') && b->h_stage_a_dev && template_init_fused_ok(b, sm);
	/* fixed-size patches of an affine patch SSM: k_template_init lays its patch out itself and the host layout runs behind the launch */
	const bool layout_later = fused && b->desc.ssm != MTFHIP_SSM_HOMOGRAPHY && !g->dyn_patch_size && !(e_ld && e_ld[0] == '0');
	if (layout_later) {
		M3 Wr;
		if (!rect_to_quad(-0.5, -0.5, 0.5, 0.5, region, Wr)) return fail(MTFHIP_ERR_INVALID_ARG, "grid_layout: degenerate region corners");
		b->deferred_gdesc = *g;
		std::memcpy(b->deferred_region, region, sizeof(b->deferred_region));
		std::memcpy(b->deferred_region_map, Wr.m, sizeof(b->deferred_region_map));
		const int rc = grid_reinit_fused(b, sm, nullptr, true);
		if (rc != MTFHIP_OK) { b->deferred_layout = false; return rc; }
		if (b->deferred_patches.size() != 8 * B) return fail(MTFHIP_ERR_LOGIC, "grid_reset: the deferred layout did not run");
		std::memcpy(patches.data(), b->deferred_patches.data(), sizeof(double) * 8 * B);
	} else {
		TRY(mtfhip_grid_layout(g, region, nullptr, patches.data()));
		if (!reinit) TRY(mtfhip_batch_set_region(b, patches.data(), sm));   /* tracker->setRegion(patch_corners) */
		else if (fused) TRY(grid_reinit_fused(b, sm, patches.data()));       /* tracker->initialize(patch_corners): NT/ICLK.cc:71-128 etc. */
		else {
			TRY(mtfhip_ssm_set_corners(b, patches.data()));
			TRY(mtfhip_batch_init_template(b, sm));
		}
	}
	if (patch_corners) std::memcpy(patch_corners, patches.data(), sizeof(double) * 8 * B);
	/* :387 getCentroid(prev_pts[id], tracker->getRegion()): both resets leave the tracker's region = the patch corners */
	if (prev_pts) for (size_t t = 0; t < B; ++t) centroid_f(prev_pts + 2 * t, &patches[8 * t]);
	return MTFHIP_OK;
}

/* the argument / state checks of the device loop, without side effects (track_region runs them before it resets the SSM) */
static int track_validate(mtfhip_batch *b, const mtfhip_sm_desc *sm) {
	TRY(check_sm(b, sm, "track"));
	TRY(fused_channels_ok(b, "track"));
	if (sm->max_iters <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "track: max_iters must be positive");
	const int so_term = second_order_term(sm, b->desc.am);
	if (b->desc.am == MTFHIP_AM_MI && so_term >= 0 && b->desc.mi_n_bins != 8)
		return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "track: second-order MI Hessians with other than 8 bins go through the per-function entry points");
	if (so_term >= 0 && b->C != 1) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "track: second-order Hessians of the multi-channel models use the per-function entry points");
	if (so_term > 0 && so_term != 4 && !b->init_pix_hess) return fail(MTFHIP_ERR_LOGIC, "track: init_template was run without sec_ord_hess");
	if (!b->init_pix_vals) return fail(MTFHIP_ERR_LOGIC, "track before init_template");
	return need_image(b);
}
static int track_core(mtfhip_batch *b, const mtfhip_sm_desc *sm, int *n_iters, double *corners, bool slab_uploaded, bool resume, bool region_mode) {
	FLUSH_AM(b);   /* (none of the loop's kernels reads CURR_PTS: they warp the template grid themselves) */
	if (b) { touch_all(b); b->lz.it_epoch = -1; TRY(ensure_df(b)); }
	TRY(track_validate(b, sm));
	const int so_term = second_order_term(sm, b->desc.am);
	hipStream_t st = b->ctx->stream;
	const bool mi = b->desc.am == MTFHIP_AM_MI;
	/* (the one-launch grid kernel has no Levenberg-Marquardt: with it ICLK takes the fused launch + finish per pass) */
	const bool one_launch = !mi && !sm->leven_marq && iclk_one_launch(b, sm) && so_term < 0;
	FusedArgs fa;
	if (!one_launch && !mi) TRY(fused_args(b, sm, fa));
	else { fa.materialize = 0; fa.mode = 2; fa.active = nullptr; fa.rows_per_block = 1; fa.j0_recompute = 0; fa.inline_warp = 0; fa.fast_math = 0; }
	/* active = 1, iters = 0, corners, warps, states, NCC scalars: one pinned async copy of the whole slab
	 * (w0 is copied along; init_grid consumed it long ago) */
	/* right behind a fused grid re-initialisation the one-launch kernels need nothing of the slab's warps / states / corners (identity, zero, the
	 * templates' own corners: TrackState::fresh_reset): no fill_stage, no ingest launch (5 us + its gap per frame of a reset-every-frame loop) */
	const bool fresh = b->fresh_reinit && one_launch && !region_mode && !slab_uploaded && !resume && b->h_pub_dev && b->d_trace == nullptr;
	b->fresh_reinit = false;
	if (!slab_uploaded && !fresh) {
		/* (h_stage_b needs no guard: every return path below has waited for the device to finish this call's work) */
		std::memcpy(b->h_stage_b + 45 * sizeof(double) * (size_t)b->B, b->h_stage_a + 45 * sizeof(double) * (size_t)b->B, 9 * sizeof(double) * (size_t)b->B);
		fill_stage(b, b->h_stage_b, nullptr, 1, true);
		/* (a pending fused initialisation, hold_init_pull: the mirrors' NCC scalars are older than d_ncc, which k_template_init wrote) */
		const bool keep_ncc = b->hold_init_pull && b->init_mirror_seq != 0;
		if (b->h_stage_b_dev) launch_ingest_host(b->h_stage_b_dev, b->d_slab, b->slab_bytes, st, keep_ncc ? 37 * sizeof(double) * (size_t)b->B : 0, keep_ncc ? 8 * sizeof(double) * (size_t)b->B : 0);
		else if (keep_ncc) return fail(MTFHIP_ERR_LOGIC, "track: a held template-initialisation record needs the host-visible staging buffer");
		else HIP_TRY(hipMemcpyAsync(b->d_slab, b->h_stage_b, b->slab_bytes, hipMemcpyHostToDevice, st));
	}
	b->warps_dirty = false;   /* the slab carries the warps */
	fa.active = b->d_active;
	const bool ncc = b->desc.am == MTFHIP_AM_NCC;
	const size_t RL = ncc ? NCC_ACC_COUNT : ACC_COUNT;   /* partial / reduced row length */
	if (ncc && !one_launch && !b->d_ncc_tm) return fail(MTFHIP_ERR_LOGIC, "track before init_template");
	/* The loop is enqueued without waiting for the device, so iterations after the last target has converged would still be
	 * launched (kernels that find every flag cleared, a few microseconds each).  With a reachable convergence test the flags are
	 * looked at every eighth iteration: one small copy + sync against up to seven idle iterations. */
	/* passes to enqueue: a rejected Levenberg-Marquardt step does not consume an iteration of FCLK's while loop (NT/FCLK.cc:193-223),
	 * and two rejections never follow each other (the pass after an undo skips the test) */
	const int max_passes = (sm->leven_marq && sm->sm == MTFHIP_SM_FCLK) ? 2 * sm->max_iters : sm->max_iters;
	std::vector<int> h_active;
	auto all_converged = [&](const int *d_flags, int n, int it, hipStream_t on = nullptr) -> bool {
		if (!(sm->epsilon > 0) || (it + 1) % 8 != 0 || it + 1 >= max_passes) return false;
		if (!on) on = st;
		h_active.resize(n);
		if (hipMemcpyAsync(h_active.data(), d_flags, sizeof(int) * n, hipMemcpyDeviceToHost, on) != hipSuccess) return false;
		if (hipStreamSynchronize(on) != hipSuccess) return false;
		for (int v : h_active) if (v) return false;
		return true;
	};
	TrackState ts{b->d_acc, b->d_h0, b->d_corners, b->d_init_corners_hm, b->d_active, b->d_iters, ncc ? b->d_ncc : nullptr, ncc ? b->d_ncc_tm : nullptr, 0, nullptr, nullptr,
		b->d_trace, b->trace_cap};
	ts.fresh_reset = fresh ? 1 : 0;
	if (b->d_trace && !resume) HIP_TRY(hipMemsetAsync(b->d_trace, 0, sizeof(double) * kTraceStride * (size_t)b->trace_cap * b->B, st));
	if (mi && b->d_trace) ts.f_ext = b->d_mi_f;   /* (the trace records the similarity; Levenberg-Marquardt sets it below as well) */
	if (sm->leven_marq && resume) { ts.lm = b->d_lm; if (mi) ts.f_ext = b->d_mi_f; }
	else if (sm->leven_marq) {
		/* per-target LM state: prev_similarity 0, leven_marq_delta = lm_delta_init, no pending reset, iteration 0 */
		if (!b->d_lm) HIP_TRY(hipMalloc(&b->d_lm, sizeof(double) * kLmStride * (size_t)b->B));
		std::vector<double> lm0((size_t)kLmStride * b->B, 0.0);
		for (int t = 0; t < b->B; ++t) lm0[(size_t)kLmStride * t + 1] = sm->lm_delta_init;
		HIP_TRY(hipMemcpyAsync(b->d_lm, lm0.data(), sizeof(double) * lm0.size(), hipMemcpyHostToDevice, st));
		HIP_TRY(hipStreamSynchronize(st));   /* lm0 is a stack-lifetime buffer */
		ts.lm = b->d_lm;
		if (mi) ts.f_ext = b->d_mi_f;
	}

	int nb2 = 0;
	if (so_term >= 0) {
		/* second-order term of SSD's Hessian inside the loop: one more pixel pass per iteration (k_second_order_ssd, the points
		 * re-derived from the warp), its S x S sums added by the finish, which then solves with pivoting */
		nb2 = simple_blocks_per_target(b->N);
		if (!b->d_d2_part) {
			HIP_TRY(hipMalloc(&b->d_d2_part, sizeof(double) * 64 * (size_t)nb2 * b->B));
			HIP_TRY(hipMalloc(&b->d_d2_out, sizeof(double) * 64 * (size_t)b->B));
		}
		/* (halved with the rest of the sum: ESM SumOfStd, NT/ESM.cc:339; MI's SumOfSelf, NT/ESM.cc:333) */
		ts.h_extra = b->d_d2_out; ts.h_extra_scale = (so_term == 1 || (so_term == 4 && sm->sm == MTFHIP_SM_ESM && sm->hess_type == 2)) ? 0.5 : 1.0;
	}
	{
		/* tolerance mode + a definite first-order system: the register-resident finish (finish_track_fast_body) */
		const char *e = std::getenv("MTFHIP_FAST_FINISH");   /* (read per call: the tests compare the two bodies in one process) */
		const bool enabled = !(e && e[0] == '0');
		ts.fast_finish = (enabled && b->math_mode == MTFHIP_MATH_FAST && !ncc && !mi && so_term < 0) ? 1 : 0;
	}
	BatchView bv = b->view();
	unsigned long long pub_seq = 0;   /* non-zero: the loop's own kernel delivers the results to the host */
	bool persisted = false;
	if (b->desc.am == MTFHIP_AM_MI) {
		/* the fused MI passes leave g and H on the device; k_finish_track_mi lays them out as one reduced row per target and
		 * runs the same finish (solve, compositional update, convergence test): no host round trip per iteration */
		const MiPlan pl(sm);
		const int gmode = pl.iclk ? 0 : (pl.fclk ? 1 : (pl.orig_jac ? 2 : 3));
		ts.h_from_acc = 1;
		const int ng = simple_blocks_per_target(b->N) < 64 ? simple_blocks_per_target(b->N) : 64;   /* as the gradient pass of mi_enqueue */
		const bool fast = mi_fast_ok(b, sm, pl);
		for (int it = 0; it < max_passes; ++it) {
			if (fast) {
				TRY(mi_enqueue_fast(b, sm, pl, b->d_active, ts, 1, so_term >= 0 ? 1 : -1));
			} else {
				TRY(mi_enqueue(b, sm, pl, b->d_active, false));
				if (so_term >= 0) mi_second_order(b, sm, 1);
				launch_finish_track_mi(bv, *sm, ts, pl.hk == MiPlan::H_SUM_STD, gmode, b->d_mi_H, b->d_partials, ng, b->d_mi_red, st);
			}
			if (all_converged(b->d_active, b->B, it)) break;
		}
	} else if (one_launch) {
		const bool fb_fused = b->fb_fused_req;   /* (mtfhip_grid_frame_fb: the frame's three per-patch steps in one launch, k_grid_fb) */
		TimedScope tsc(b->ctx, fb_fused ? "grid_fb" : "iclk_track");
		HostPublish pub{nullptr, 0, 0, nullptr, nullptr, 0, 0};
		if (b->h_pub_dev) {
			pub_seq = ++b->acc_seq;
			pub = HostPublish{b->h_pub_dev, b->slab_dbl_bytes, b->B, b->d_fin_count, b->h_flag_dev, pub_seq, publish_fenced()};
		}
		RegionIngest rg{};
		if (region_mode) {
			/* (the staging slab of set_corners_core: w 9 | s 8 | corners 8 | init_corners_hm 12 | NCC scalars 8 | w0 9 per target) */
			const double *stage = reinterpret_cast<const double *>(b->h_stage_a_dev);
			const bool homg = b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY;
			rg.corners = stage + 17 * (size_t)b->B; rg.ncc = stage + 37 * (size_t)b->B;
			rg.d_ncc = b->d_ncc; rg.d_w0 = b->d_w0; rg.d_init_corners_hm = b->d_init_corners_hm;
			rg.lo_x = homg ? -0.5 : 1 - b->desc.resx / 2.0; rg.lo_y = homg ? -0.5 : 1 - b->desc.resy / 2.0;
			rg.hi_x = homg ? 0.5 : b->desc.resx / 2.0; rg.hi_y = homg ? 0.5 : b->desc.resy / 2.0;
			rg.resx = b->desc.resx; rg.resy = b->desc.resy; rg.force_unit_z = homg ? 0 : 1;
			if (b->deferred_layout) {   /* the kernel lays its patches out itself (track_region_impl) */
				const mtfhip_grid_desc &gd = b->deferred_gdesc;
				rg.layout = 1;
				rg.grid = GridLayoutHD{gd.grid_size_x, gd.grid_size_y, gd.patch_size_x, gd.patch_size_y, gd.dyn_patch_size ? 1 : 0, gd.patch_centroid_inside ? 1 : 0};
				std::memcpy(rg.region_map, b->deferred_region_map, sizeof(rg.region_map));
			}
		}
		const bool dbg_t = g_track_dbg_timing;
		if (dbg_t) g_track_dbg_t[0] = std::chrono::steady_clock::now();
		if (fb_fused) {
			if (region_mode || !pub.host) return fail(MTFHIP_ERR_LOGIC, "track: the one-launch forward-backward frame takes the plain mode with a host record");
			const bool homg = b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY;
			RegionIngest geo{};   /* (the template lattice's geometry: what grid_reinit_fused hands k_template_init) */
			geo.lo_x = homg ? -0.5 : 1 - b->desc.resx / 2.0; geo.lo_y = homg ? -0.5 : 1 - b->desc.resy / 2.0;
			geo.hi_x = homg ? 0.5 : b->desc.resx / 2.0; geo.hi_y = homg ? 0.5 : b->desc.resy / 2.0;
			geo.resx = b->desc.resx; geo.resy = b->desc.resy; geo.force_unit_z = homg ? 0 : 1;
			if (!launch_grid_fb(bv, b->ctx->img, b->ctx->prev, *sm, ts, b->d_h0inv, b->d_ncc, b->norm_mult, b->norm_add, b->desc.grad_eps, pub, GridFbOut{b->h_fb_dev, b->d_fb, b->fb_fused_reinit ? 1 : 0}, geo, st))
				return fail(MTFHIP_ERR_LOGIC, "track: patch too large for the one-launch forward-backward frame");
		} else
		launch_iclk_track(bv, b->ctx->img, *sm, ts, b->d_h0inv, b->d_ncc, b->norm_mult, b->norm_add, b->math_mode == MTFHIP_MATH_FAST, pub, rg, st);
		if (dbg_t) g_track_dbg_t[1] = std::chrono::steady_clock::now();
		set_corners_finish_deferred(b);   /* the host half of a deferred reset, under the kernel */
		if (dbg_t) g_track_dbg_t[2] = std::chrono::steady_clock::now();
	} else if (so_term < 0 && persist_fits(b, sm, fa)) {
		/* a grid that fits the device at one workgroup per CU (a single large target, a few small ones): every pass of the loop in
		 * ONE launch, the workgroups meeting at an in-kernel barrier between the pixel pass and the solve (kernels_persist.hip) */
		persisted = true;
		if (!b->d_persist) {
			HIP_TRY(hipMalloc(&b->d_persist, 2 * sizeof(int) * (size_t)b->B));
			HIP_TRY(hipMemsetAsync(b->d_persist, 0, 2 * sizeof(int) * (size_t)b->B, st));
		}
		int nblk_p, rows_p;
		persist_decomposition(b, nblk_p, rows_p);
		FusedArgs fp = fa;
		fp.rows_per_block = rows_p;
		PersistState ps{b->d_persist, reinterpret_cast<unsigned *>(b->d_persist) + b->B, b->persist_gen, persist_timeout_ticks()};
		b->persist_gen += (unsigned)max_passes + 1u;
		{
			TimedScope tsc(b->ctx, "track_persist");
			launch_track_persist(bv, b->ctx->img, fp, *sm, ts, b->d_partials, nblk_p, ps, max_passes, st);
		}
	} else {
		/* Targets are independent, so the loops commute: all iterations of a chunk of targets run before the next chunk
		 * starts.  A chunk is sized so that what an iteration reads once (J0, I0, grid: 88 B/px for ESM) stays resident in
		 * the 256 MB Infinity Cache from one iteration to the next -- B = 64 at 200 x 200; larger batches used to fall back
		 * to plain HBM for both streams (0.62 instead of 0.75 of peak).  See track_chunk(). */
		int chunk = track_chunk(b, sm, fa);
		/* two chunks of targets in flight on two queues where that pays (track_queues); each chunk's pixel pass is then cut for half
		 * the resident workgroups, so that the two launches in flight fill the device once */
		int n_streams = track_queues(b, fa);
		if (n_streams >= 2) {
			mtfhip_ctx *c = b->ctx;
			if (!c->ev_fork && hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); n_streams = 1; }
			for (int q = 0; q + 1 < n_streams; ++q)
				if (!c->extra_streams[q] && (hipStreamCreateWithFlags(&c->extra_streams[q], hipStreamNonBlocking) != hipSuccess ||
					hipEventCreateWithFlags(&c->ev_join[q], hipEventDisableTiming) != hipSuccess)) { (void)hipGetLastError(); n_streams = 1; break; }
		}
		const char *e_ph = std::getenv("MTFHIP_TRACK_PHASE");   /* fraction of a period the queues are kept apart; 0 = no control */
		const double phase_frac = e_ph ? std::atof(e_ph) : 0.35;
		if (n_streams == 2 && phase_frac > 0) {
			if (!b->ctx->d_phase && hipMalloc(&b->ctx->d_phase, sizeof(unsigned long long) * 4) != hipSuccess) { (void)hipGetLastError(); b->ctx->d_phase = nullptr; }
			if (b->ctx->d_phase) HIP_TRY(hipMemsetAsync(b->ctx->d_phase, 0, sizeof(unsigned long long) * 4, st));
		}
		/* The queues start a quarter of a period apart (a spinning one-wave kernel in front of the later one; the period is estimated from
		 * the bytes a pass moves): started together they stay in lockstep on some boxes -- the fill and drain phases of the two pixel
		 * passes coincide and so do the two solves, 55 us per step of 64 x 200 x 200 against 49 out of phase (from there on the solve
		 * kernels keep them apart, PhaseCtl).  A/B at that size, three boxes: no delay 1.05-1.10 M iters/s in 20-iteration calls and
		 * 1.13-1.19 M in 200-iteration ones, 15 us 1.13-1.15 M and 1.28-1.30 M, 25 / 30 / 35 us in between and less repeatable.
		 * MTFHIP_TRACK_STAGGER_US: > 0 that many microseconds, 0 none.
		 * The context's own stream takes the LATER chunk of a pair: it is then the last to finish, and the join at the end of the call finds
		 * the other queue's event already signalled instead of paying a cross-queue wait (~12 us) in front of the result read-back. */
		static const double stagger_env = std::getenv("MTFHIP_TRACK_STAGGER_US") ? std::atof(std::getenv("MTFHIP_TRACK_STAGGER_US")) : -1.0;
		double stagger_us = stagger_env;
		if (stagger_env < 0) stagger_us = 0.25 * ((double)b->B * b->N * 130.0 / 6.5e6 + 8.0) * (2.0 / n_streams);
		if (n_streams >= 2) {
			const int part_sz = (b->B + n_streams - 1) / n_streams;
			if (chunk > part_sz) chunk = part_sz;
			HIP_TRY(hipEventRecord(b->ctx->ev_fork, st));   /* the slab upload */
			for (int q = 0; q + 1 < n_streams; ++q) HIP_TRY(hipStreamWaitEvent(b->ctx->extra_streams[q], b->ctx->ev_fork, 0));
		}
		/* small batches (a single tracker's target, a handful of them): one launch per pass instead of two -- the pixel pass's last
		 * workgroup runs the finish (kernels_step.hip).  MEASURED r05 (one box, ESM + SSD + homography, 200 iterations per call): 200 x 200
		 * full 12.22 -> 12.54 us per iteration, lean 10.92 -> 10.79, 50 x 50 lean 10.55 -> 10.44: nothing.  The r04 verdict's estimate (a
		 * launch boundary = the finish kernel's 4.9 us) does not hold: the in-kernel hand-over -- acknowledged stores, an agent-scope
		 * arrival, ~160 rows read back past the L2 -- costs what the boundary cost, as the persistent loop's did in r03.  Opt-in
		 * (MTFHIP_STEP=1) and bit-identical to the two-launch loop (test_one_launch_per_pass_equals_two_launch_loop);
		 * MTFHIP_STEP_MAX_TARGETS bounds the batch size it takes (default 8). */
		bool use_step = false;
		{
			const char *e_st = std::getenv("MTFHIP_STEP");   /* (read per call: the tests flip it) */
			const char *e_mx = std::getenv("MTFHIP_STEP_MAX_TARGETS");
			const int max_t = e_mx ? std::atoi(e_mx) : 8;
			use_step = (e_st && e_st[0] == '1') && so_term < 0 && n_streams == 1 && b->B <= max_t && track_step_available(bv, fa);
			if (use_step && !b->d_persist) {
				HIP_TRY(hipMalloc(&b->d_persist, 2 * sizeof(int) * (size_t)b->B));
				HIP_TRY(hipMemsetAsync(b->d_persist, 0, 2 * sizeof(int) * (size_t)b->B, st));
			}
		}
		struct ChunkRun { BatchView bc; FusedArgs fc; TrackState tc; int nblk_c, t0, nt; double *part; hipStream_t s; bool done; };
		std::vector<ChunkRun> runs;
		for (int t0 = 0; t0 < b->B; t0 += chunk) {
			const int nt = std::min(chunk, b->B - t0);
			BatchView bc = bv;
			bc.B = nt;
			for (int i = 0; i < MTFHIP_BUF_COUNT; ++i)
				if (bc.buf[i]) bc.buf[i] += (size_t)t0 * b->per_target[i];
			bc.warps += 9 * (size_t)t0; bc.states += 8 * (size_t)t0;
			FusedArgs fc = fa;
			fc.active = fa.active + t0;
			TrackState tc{ts.acc + (size_t)t0 * RL, ts.h0 + (size_t)t0 * 64, ts.corners + 8 * (size_t)t0,
				ts.init_corners_hm + 12 * (size_t)t0, ts.active + t0, ts.n_iters + t0, ncc ? ts.ncc + 8 * (size_t)t0 : nullptr,
				ncc ? ts.ncc_tm + 52 * (size_t)t0 : nullptr, 0, ts.lm ? ts.lm + (size_t)kLmStride * t0 : nullptr, nullptr,
				ts.trace ? ts.trace + (size_t)t0 * ts.trace_cap * kTraceStride : nullptr, ts.trace_cap,
				ts.h_extra ? ts.h_extra + (size_t)t0 * b->S * b->S : nullptr, ts.h_extra_scale, ts.fast_finish};
			int nblk_c; { int rows; fused_decomposition(b->N, nt, nblk_c, rows, MTFHIP_SLOTS / n_streams); fc.rows_per_block = rows; }
			if (nblk_c > b->nblk_max) { int rows; fused_decomposition(b->N, nt, nblk_c, rows); fc.rows_per_block = rows; }
			double *part = b->d_partials + (size_t)t0 * b->nblk_max * RL;
			/* MTFHIP_TRACK_SERIALIZE=1: the same chunks and the same cut of the pixel pass, one queue -- for the PMC passes, whose
			 * per-dispatch counters are device-wide and would include the launch in flight on the other queue */
			static const bool serialize = std::getenv("MTFHIP_TRACK_SERIALIZE") && std::getenv("MTFHIP_TRACK_SERIALIZE")[0] == '1';
			const int q = serialize ? n_streams - 1 : (int)(runs.size() % (size_t)n_streams);
			runs.push_back(ChunkRun{bc, fc, tc, nblk_c, t0, nt, part, q == n_streams - 1 ? st : b->ctx->extra_streams[q], false});
		}
		const auto dbg_t0 = std::chrono::steady_clock::now();
		/* Every way out of the loop below joins the extra queues: the normal path with an event the context's stream waits on; an
		 * early return (a failed launch or copy: TRY / HIP_TRY) by draining them here -- kernels still in flight on an extra queue
		 * would otherwise race with whatever the caller enqueues next on the context's stream (r03 advisor finding). */
		struct QueueJoin {
			mtfhip_ctx *c; int n; bool joined = false;
			~QueueJoin() { if (!joined) for (int q = 0; q + 1 < n; ++q) if (c->extra_streams[q]) (void)hipStreamSynchronize(c->extra_streams[q]); }
		} queue_join{b->ctx, n_streams};
		/* the chunks of a group (one per queue) advance together, pass by pass, so that both queues are fed from the start */
		for (size_t g0 = 0; g0 < runs.size(); g0 += (size_t)n_streams) {
			const size_t g1 = std::min(runs.size(), g0 + (size_t)n_streams);
			for (int it = 0; it < max_passes; ++it) {
				bool all_done = true;
				for (size_t k = g0; k < g1; ++k) {
					ChunkRun &r = runs[k];
					if (r.done) continue;
					if (n_streams >= 2 && it == 0 && k > g0 && stagger_us > 0) launch_queue_delay(stagger_us * (double)(k - g0), r.s);
					if (use_step) {
						/* one launch per pass: the last workgroup to arrive solves and updates (kernels_step.hip) */
						TimedScope tsc(b->ctx, "track_step", r.s);
						launch_track_step(r.bc, b->ctx->img, r.fc, *sm, r.tc, r.part, r.nblk_c, b->d_persist + r.t0, r.s);
						if (all_converged(r.tc.active, r.nt, it, r.s)) r.done = true;
						all_done = all_done && r.done;
						continue;
					}
					{
						TimedScope tsc(b->ctx, "fused_lk", r.s);
						launch_fused_ssd(r.bc, b->ctx->img, r.fc, r.part, r.nblk_c, r.s);
					}
					if (so_term >= 0) {
						TimedScope tsc(b->ctx, "second_order", r.s);
						launch_second_order_ssd(r.bc, b->ctx->img, so_term, fa.chained, b->d0_variant, fa.grad_eps, b->hess_eps, b->norm_mult, b->norm_add,
							b->d_d2_part + (size_t)r.t0 * nb2 * 64, nb2, b->d_d2_out + (size_t)r.t0 * b->S * b->S, r.s, 1,
							ncc ? SecondOrderNcc{r.part, r.nblk_c, r.tc.ncc} : SecondOrderNcc{nullptr, 0, nullptr});
					}
					{
						TimedScope tsc(b->ctx, "finish_track", r.s);
						PhaseCtl pc{nullptr, nullptr, 0.0};
						if (n_streams == 2 && phase_frac > 0 && b->ctx->d_phase) {
							const int qi = (int)((k - g0) & 1);
							pc = PhaseCtl{b->ctx->d_phase + qi, b->ctx->d_phase + (1 - qi), phase_frac};
						}
						launch_finish_track(r.bc, *sm, r.tc, r.part, r.nblk_c, r.s, pc);
					}
					if (all_converged(r.tc.active, r.nt, it, r.s)) r.done = true;
					all_done = all_done && r.done;
				}
				if (all_done) break;
			}
		}
		if (std::getenv("MTFHIP_TRACK_DEBUG_TIMING"))
			std::fprintf(stderr, "[track] %d queues, %d passes enqueued in %.1f us\n", n_streams, max_passes,
				std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - dbg_t0).count());
		for (int q = 0; q + 1 < n_streams; ++q) {
			HIP_TRY(hipEventRecord(b->ctx->ev_join[q], b->ctx->extra_streams[q]));
			HIP_TRY(hipStreamWaitEvent(st, b->ctx->ev_join[q], 0));
		}
		queue_join.joined = true;
	}
	/* the slab (warps, states, corners, iteration counts) comes back either through a kernel that writes it into host-coherent
	 * memory and raises a flag the host spins on, or as one copy + one sync (MTFHIP_ZERO_COPY=0) */
	const char *h_res = b->h_stage_b;
	if (pub_seq) {
		TRY(wait_host_flag(b, pub_seq));
		h_res = b->h_pub;
	} else if (b->h_pub_dev) {
		const unsigned long long seq = ++b->acc_seq;
		launch_publish_host(b->d_slab, b->h_pub_dev, b->slab_bytes, b->d_fin_count, b->h_flag_dev, seq, st);
		TRY(wait_host_flag(b, seq));
		h_res = b->h_pub;
	} else {
		HIP_TRY(hipMemcpyAsync(b->h_stage_b, b->d_slab, b->slab_bytes, hipMemcpyDeviceToHost, st));
		HIP_TRY(hipStreamSynchronize(st));
	}
	{
		const size_t Bt = (size_t)b->B;
		const double *p = reinterpret_cast<const double *>(h_res);
		const double *w = p, *s = p + 9 * Bt, *cr = p + 17 * Bt;
		const int *iters = reinterpret_cast<const int *>(h_res + b->slab_dbl_bytes) + Bt;
		for (int t = 0; t < b->B; ++t) {
			std::memcpy(b->th[t].warp.m, w + 9 * t, sizeof(double) * 9);
			std::memcpy(b->th[t].state, s + 8 * t, sizeof(double) * 8);
			std::memcpy(b->th[t].corners, cr + 8 * t, sizeof(double) * 8);
			if (n_iters) n_iters[t] = iters[t];
			if (corners) std::memcpy(corners + 8 * t, cr + 8 * t, sizeof(double) * 8);
		}
		if (region_mode)
			for (int t = 0; t < b->B; ++t)
				if (iters[t] < 0) return fail(MTFHIP_ERR_INVALID_ARG, "track_region: degenerate corners for target %d", t);
	}
	if (persisted) {
		/* a workgroup that could not wait any longer for its peers (CUs held by another process) leaves its target active with
		 * iterations to go: the two-launch loop takes the call from where the device stopped, and this batch stays with it */
		const int *act = reinterpret_cast<const int *>(h_res + b->slab_dbl_bytes);
		bool cut = false;
		for (int t = 0; t < b->B; ++t) cut = cut || act[t] != 0;
		if (cut) {
			b->persist_ok = false;
			HIP_TRY(hipStreamSynchronize(st));
			HIP_TRY(hipMemsetAsync(b->d_persist, 0, 2 * sizeof(int) * (size_t)b->B, st));
			return track_resume(b, sm, n_iters, corners);
		}
	}
	if (!mi) {   /* (mi_enqueue keeps the flags of its own passes) */
		b->it_valid = fa.materialize;
		b->dit_valid = b->jt_valid = fa.materialize && fa.mode != 2;
	}
	b->pts_stale = true;   /* CURR_PTS follow the final warp when an un-fused kernel next needs them */
	b->stage_a_busy = false;   /* the stream has drained: whatever set_corners staged has been consumed */
	return MTFHIP_OK;
}

/* debug trace of the device-side loop: with max_passes > 0 every pass of mtfhip_batch_track / _track_region also records what it
 * solved (H, g, the state update, the corners it produced, f) -- the per-iteration quantities the parity tests compare with the
 * CPU trackers' traces; 0 switches it off (the default: a NULL test per pass) */
int mtfhip_batch_track_trace(mtfhip_batch *b, int max_passes) {
	if (!b || max_passes < 0) return fail(MTFHIP_ERR_INVALID_ARG, "track_trace: invalid argument");
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	if (b->d_trace) { (void)hipFree(b->d_trace); b->d_trace = nullptr; }
	b->trace_cap = max_passes;
	if (max_passes > 0) {
		HIP_TRY(hipMalloc(&b->d_trace, sizeof(double) * kTraceStride * (size_t)max_passes * b->B));
		HIP_TRY(hipMemsetAsync(b->d_trace, 0, sizeof(double) * kTraceStride * (size_t)max_passes * b->B, b->ctx->stream));
	}
	return MTFHIP_OK;
}
int mtfhip_batch_track_trace_read(mtfhip_batch *b, double *dst) {
	if (!b || !dst) return fail(MTFHIP_ERR_INVALID_ARG, "track_trace_read: NULL argument");
	if (!b->d_trace) return fail(MTFHIP_ERR_LOGIC, "track_trace_read: tracing is off (mtfhip_batch_track_trace)");
	HIP_TRY(hipMemcpyAsync(dst, b->d_trace, sizeof(double) * kTraceStride * (size_t)b->trace_cap * b->B, hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ candidate scoring */
/* candidates [lo, lo + cnt) of dev_states: weight (the AM's likelihood, or PF's Gaussian / reciprocal mapping of the similarity) and
 * similarity at their global indices.  SSD / NCC (also multi-channel): k_pf_score; MI (8 bins): the histogram pass over the candidate
 * axis + k_mi_cand_score.  Shared by mtfhip_score_candidates_dev and the particle filter. */
int score_block_dev(mtfhip_batch *b, const double *dev_states, int lo, int cnt, double *wts, double *sim, int likelihood_func,
	double measurement_sigma, double max_similarity, const PfPeerPush *peer) {
	hipStream_t st = b->ctx->stream;
	if (b->desc.am == MTFHIP_AM_MI) {
		if (!(b->desc.mi_n_bins == 8 || (b->desc.mi_n_bins <= 10 && b->C == 1)))
			return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "score_candidates: MI candidates are scored with up to 10 bins (multi-channel: 8, the reference's default, parameters.h:344)");
		if (!b->init_sim) return fail(MTFHIP_ERR_LOGIC, "score_candidates before initializeSimilarity");
		const int nblk = 1;
		const size_t need = (size_t)std::max(cnt, 1) * nblk * b->mi_row_len;
		if (need > b->cand_mi_capacity) {
			HIP_TRY(hipStreamSynchronize(st));
			if (b->d_cand_mi) (void)hipFree(b->d_cand_mi);
			b->d_cand_mi = nullptr;
			HIP_TRY(hipMalloc(&b->d_cand_mi, sizeof(double) * need));
			b->cand_mi_capacity = need;
		}
		MiFastPlan fp;
		fp.nb = b->desc.mi_n_bins;
		fp.hk = 0; fp.hrow = 0; fp.j0_mode = 0; fp.j0_init_variant = 0; fp.need_dft = 0; fp.need_df0 = 0; fp.g_mean = 0;
		fp.grad_eps = b->desc.grad_eps; fp.norm_mult = b->norm_mult; fp.norm_add = b->norm_add; fp.hist_norm = b->mi_hist_norm;
		fp.active = nullptr; fp.tb = b->d_mi_tb;
		launch_mi_score_candidates(b->view_raw(), b->ctx->img, fp, dev_states, lo, cnt, b->d_cand_mi, nblk, b->mi_row_len, b->desc.mi_pre_seed,
			b->desc.likelihood_alpha, likelihood_func, measurement_sigma, max_similarity, wts, sim, st);
		if (peer && cnt > 0) launch_pf_peer_push(*peer, wts, lo, cnt, st);   /* (the MI scorer does not store to the peers itself) */
		return MTFHIP_OK;
	}
	const double *ncc_sc = nullptr;
	if (b->desc.am == MTFHIP_AM_NCC) {   /* mean(I0), |I0 - mean| of the template, as the un-fused NCC kernels read them */
		if (!b->init_sim) return fail(MTFHIP_ERR_LOGIC, "score_candidates before initializeSimilarity");
		TRY(push_ncc(b));
		ncc_sc = b->d_ncc;
	}
	/* the template grid's own corners (set_corners lays a unit-z grid out INSIDE them: the lattice's end points are the corners):
	 * what lets the scorer skip the border test of a candidate whose warped corners are inside the frame */
	double hull_buf[8];
	const double *hull = nullptr;
	if (b->unit_z && b->grid_from_corners && b->B >= 1) {
		const double *ic = b->th[0].init_corners_hm;
		bool unit = true;
		for (int q = 0; q < 4; ++q) { hull_buf[2 * q] = ic[3 * q]; hull_buf[2 * q + 1] = ic[3 * q + 1]; unit = unit && ic[3 * q + 2] == 1.0; }
		if (unit) hull = hull_buf;
	}
	/* (view_raw: the candidates bring their own warps; a stale device copy of the batch's warp is not uploaded for them) */
	launch_score_block(b->view_raw(), b->ctx->img, dev_states, lo, cnt, b->desc.likelihood_alpha, b->norm_mult, b->norm_add, ncc_sc, wts, sim,
		likelihood_func, measurement_sigma, max_similarity, b->math_mode == MTFHIP_MATH_FAST, peer, hull,
		(b->math_mode == MTFHIP_MATH_FAST && b->C == 1) ? pair_image_if_it_pays(b->ctx, cnt) : nullptr, st);
	return MTFHIP_OK;
}
int mtfhip_score_candidates_dev(mtfhip_batch *b, const double *dev_states, int C, double *dev_lik, double *dev_sim) {
	FLUSH_AM(b);   /* (every candidate warps the template grid itself: CURR_PTS are not read) */
	if (!b || !dev_states) return fail(MTFHIP_ERR_INVALID_ARG, "score_candidates: NULL argument");
	if (C <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "score_candidates: n_candidates must be positive");
	if (!b->init_pix_vals) return fail(MTFHIP_ERR_LOGIC, "score_candidates before the template was initialised");
	TRY(need_image(b));
	TimedScope ts(b->ctx, "score_candidates");
	return score_block_dev(b, dev_states, 0, C, dev_lik, dev_sim, 0, 1.0, 0.0);
}

int mtfhip_score_candidates(mtfhip_batch *b, const double *states, int C, double *lik, double *sim) {
	FLUSH(b);
	if (!b || !states) return fail(MTFHIP_ERR_INVALID_ARG, "score_candidates: NULL argument");
	if (C <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "score_candidates: n_candidates must be positive");
	size_t need = (size_t)C * (b->S + 2);
	if (need > b->cand_capacity) {
		if (b->d_cand) HIP_TRY(hipFree(b->d_cand));
		b->d_cand = nullptr;
		HIP_TRY(hipMalloc(&b->d_cand, sizeof(double) * need));
		b->cand_capacity = need;
	}
	double *d_states = b->d_cand, *d_lik = b->d_cand + (size_t)C * b->S, *d_sim = d_lik + C;
	HIP_TRY(hipMemcpyAsync(d_states, states, sizeof(double) * C * b->S, hipMemcpyHostToDevice, b->ctx->stream));
	TRY(mtfhip_score_candidates_dev(b, d_states, C, d_lik, d_sim));
	if (lik) HIP_TRY(hipMemcpyAsync(lik, d_lik, sizeof(double) * C, hipMemcpyDeviceToHost, b->ctx->stream));
	if (sim) HIP_TRY(hipMemcpyAsync(sim, d_sim, sizeof(double) * C, hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ NN dataset generation */
int mtfhip_sample_candidates_dev(mtfhip_batch *b, const double *dev_states, int C, double *dev_features) {
	FLUSH(b);
	if (!b || !dev_states || !dev_features) return fail(MTFHIP_ERR_INVALID_ARG, "sample_candidates: NULL argument");
	if (C <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "sample_candidates: n_samples must be positive");
	if (b->desc.am == MTFHIP_AM_MI) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "sample_candidates: MI distance features (5 x N B-spline rows) are not available");
	TRY(single_channel(b, "sample_candidates"));
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "sample_candidates before set_corners");
	TRY(need_image(b));
	TimedScope ts(b->ctx, "sample_candidates");
	launch_sample_candidates(b->view_raw(), b->ctx->img, dev_states, C, b->norm_mult, b->norm_add, dev_features, b->ctx->stream);
	return MTFHIP_OK;
}
int mtfhip_sample_candidates(mtfhip_batch *b, const double *states, int C, double *features) {
	FLUSH(b);
	if (!b || !states || !features) return fail(MTFHIP_ERR_INVALID_ARG, "sample_candidates: NULL argument");
	if (C <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "sample_candidates: n_samples must be positive");
	double *d_states = nullptr, *d_feat = nullptr;
	HIP_TRY(hipMalloc(&d_states, sizeof(double) * C * b->S));
	if (hipMalloc(&d_feat, sizeof(double) * (size_t)C * b->N) != hipSuccess) { (void)hipFree(d_states); return fail(MTFHIP_ERR_HIP, "hipMalloc of the %d x %d feature matrix failed", C, b->N); }
	int rc = MTFHIP_OK;
	if (hipMemcpyAsync(d_states, states, sizeof(double) * C * b->S, hipMemcpyHostToDevice, b->ctx->stream) != hipSuccess) rc = fail(MTFHIP_ERR_HIP, "state upload failed");
	if (rc == MTFHIP_OK) rc = mtfhip_sample_candidates_dev(b, d_states, C, d_feat);
	if (rc == MTFHIP_OK && hipMemcpyAsync(features, d_feat, sizeof(double) * (size_t)C * b->N, hipMemcpyDeviceToHost, b->ctx->stream) != hipSuccess) rc = fail(MTFHIP_ERR_HIP, "feature read-back failed");
	if (hipStreamSynchronize(b->ctx->stream) != hipSuccess && rc == MTFHIP_OK) rc = fail(MTFHIP_ERR_HIP, "stream synchronisation failed");
	(void)hipFree(d_states); (void)hipFree(d_feat);
	return rc;
}

/* NN::generateDataset (SM/src/NT/NN.cc:131-191) */
int mtfhip_nn_feature_size(mtfhip_batch *b, int *feat_size) {
	if (!b || !feat_size) return fail(MTFHIP_ERR_INVALID_ARG, "nn_feature_size: NULL argument");
	*feat_size = b->desc.am == MTFHIP_AM_MI ? 5 * b->N : b->N;   /* MI.cc:122: feat_size = 5 * patch_size; SSDBase.h:116-125, NCC.cc:530-537: patch_size */
	return MTFHIP_OK;
}
int mtfhip_nn_dataset_dev(mtfhip_batch *b, const mtfhip_nn_desc *d, const double *dev_perturbations_in, double *dev_perturbations_out, double *dev_features,
	int row_lo, int row_count) {
	FLUSH(b);
	if (!b || !d || !dev_features) return fail(MTFHIP_ERR_INVALID_ARG, "nn_dataset: NULL argument");
	if (d->n_samples <= 0 || row_lo < 0 || row_count < 0 || row_lo + row_count > d->n_samples)
		return fail(MTFHIP_ERR_INVALID_ARG, "nn_dataset: rows [%d, %d) of %d samples", row_lo, row_lo + row_count, d->n_samples);
	if (d->additive_update) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "nn_dataset: additive_update (NNParams, NT/NN.cc:150-152): the compositional form only");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "nn_dataset before set_corners");
	if (b->B != 1) return fail(MTFHIP_ERR_INVALID_ARG, "nn_dataset: one template per batch (the batch has %d targets)", b->B);
	TRY(need_image(b));
	for (int s = 0; s < b->S; ++s) if (!(d->sigma[s] >= 0)) return fail(MTFHIP_ERR_INVALID_ARG, "nn_dataset: sigma[%d] = %g", s, d->sigma[s]);
	NnArgs a;
	a.perts_in = dev_perturbations_in; a.perts_out = dev_perturbations_out;
	for (int s = 0; s < 8; ++s) { a.sigma[s] = s < b->S ? d->sigma[s] : 0.0; a.mean[s] = s < b->S ? d->mean[s] : 0.0; }
	a.seed = d->seed;
	std::memcpy(a.base, b->th[0].warp.m, sizeof(a.base));
	a.row_lo = row_lo; a.norm_mult = b->norm_mult; a.norm_add = b->norm_add;
	/* the template grid's own corners, as the candidate scorer takes them: a sample whose warped hull is inside the frame skips the border test */
	double hull_buf[8];
	const double *hull = nullptr;
	if (b->unit_z && b->grid_from_corners) {
		const double *ic = b->th[0].init_corners_hm;
		bool unit = true;
		for (int q = 0; q < 4; ++q) { hull_buf[2 * q] = ic[3 * q]; hull_buf[2 * q + 1] = ic[3 * q + 1]; unit = unit && ic[3 * q + 2] == 1.0; }
		if (unit) hull = hull_buf;
	}
	/* tolerance mode: the samples' warps go through a scratch array (k_nn_warps -> k_nn_rows), grown to the largest launch so far */
	double *warps = nullptr;
	if (nn_two_launch_ok(b->view_raw(), b->ctx->img, b->math_mode == MTFHIP_MATH_FAST)) {
		const size_t need = nn_warps_bytes(row_count);
		if (need > b->nn_warps_cap) {
			HIP_TRY(hipStreamSynchronize(b->ctx->stream));
			if (b->d_nn_warps) { (void)hipFree(b->d_nn_warps); b->d_nn_warps = nullptr; b->nn_warps_cap = 0; }
			HIP_TRY(hipMalloc(&b->d_nn_warps, need));
			b->nn_warps_cap = need;
		}
		warps = b->d_nn_warps;
	}
	TimedScope ts(b->ctx, "nn_dataset");
	launch_nn_dataset(b->view_raw(), b->ctx->img, a, row_count, dev_features, warps, hull, b->ctx->stream);
	return MTFHIP_OK;
}
int mtfhip_nn_dataset(mtfhip_batch *b, const mtfhip_nn_desc *d, const double *perturbations_in, double *perturbations_out, double *features) {
	if (!b || !d || !features) return fail(MTFHIP_ERR_INVALID_ARG, "nn_dataset: NULL argument");
	if (d->n_samples <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "nn_dataset: n_samples must be positive");
	int F = 0;
	TRY(mtfhip_nn_feature_size(b, &F));
	const size_t C = (size_t)d->n_samples;
	double *d_p = nullptr, *d_feat = nullptr;
	HIP_TRY(hipMalloc(&d_p, sizeof(double) * C * b->S));
	if (hipMalloc(&d_feat, sizeof(double) * C * F) != hipSuccess) { (void)hipFree(d_p); return fail(MTFHIP_ERR_HIP, "hipMalloc of the %d x %d feature matrix failed", d->n_samples, F); }
	int rc = MTFHIP_OK;
	hipStream_t st = b->ctx->stream;
	if (perturbations_in && hipMemcpyAsync(d_p, perturbations_in, sizeof(double) * C * b->S, hipMemcpyHostToDevice, st) != hipSuccess) rc = fail(MTFHIP_ERR_HIP, "perturbation upload failed");
	if (rc == MTFHIP_OK) rc = mtfhip_nn_dataset_dev(b, d, perturbations_in ? d_p : nullptr, d_p, d_feat, 0, d->n_samples);
	if (rc == MTFHIP_OK && perturbations_out && hipMemcpyAsync(perturbations_out, d_p, sizeof(double) * C * b->S, hipMemcpyDeviceToHost, st) != hipSuccess) rc = fail(MTFHIP_ERR_HIP, "perturbation read-back failed");
	if (rc == MTFHIP_OK && hipMemcpyAsync(features, d_feat, sizeof(double) * C * F, hipMemcpyDeviceToHost, st) != hipSuccess) rc = fail(MTFHIP_ERR_HIP, "feature read-back failed");
	if (hipStreamSynchronize(st) != hipSuccess && rc == MTFHIP_OK) rc = fail(MTFHIP_ERR_HIP, "stream synchronisation failed");
	(void)hipFree(d_p); (void)hipFree(d_feat);
	return rc;
}


} /* extern "C" */

#ifdef MTFHIP_FIN_TRACE
namespace mtfhip { void debug_fin_trace(unsigned long long *out); }
extern "C" void mtfhip_debug_fin_trace(unsigned long long *out) { mtfhip::debug_fin_trace(out); }
#endif
