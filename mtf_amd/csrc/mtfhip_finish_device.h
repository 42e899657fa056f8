/*
 * mtfhip_finish_device.h -- the device-side finish of an LK iteration (solve, compositional update, convergence test),
 * shared by the translation units that end an iteration with it (kernels_fused.hip: k_finish_track, k_finish_track_mi;
 * kernels_mi_fused.hip: k_mi_finish_fast).  No relocatable device code: the body is inlined into each kernel.
 */
#ifndef MTFHIP_FINISH_DEVICE_H
#define MTFHIP_FINISH_DEVICE_H
#include "mtfhip_device.h"

namespace mtfhip {

#ifdef MTFHIP_FIN_TRACE
__device__ unsigned long long g_fin_trace[16];
#define FIN_STAMP(k) do { if (t == 0 && threadIdx.x == 0) g_fin_trace[k] = clock64(); } while (0)
#else
#define FIN_STAMP(k) do { } while (0)
#endif

/* ===================================================================== */
/* on-device solve + compositional update (batched drivers only)          */
/* ===================================================================== */
/* One wave64 per target, one launch per LK iteration:
 *   (1) fixed-order sum of the per-workgroup partial rows (what k_finish does for the host-driven path),
 *   (2) g and H of the search method from the accumulators (NT/FCLK.cc:260-288, NT/ESM.cc:298-377 with
 *       SSDBase.cc:169-191,287-311, NT/ICLK.cc:206-251),
 *   (3) H dp = -g by Gauss-Jordan elimination spread over the 64 lanes (lane = matrix entry) on the
 *       symmetrically diagonal-scaled system; every SSD Hessian here is a negated Gram matrix, i.e.
 *       definite, so no pivoting is needed (the reference uses Eigen's colPivHouseholderQr, NT/FCLK.cc:298),
 *   (4) the (inverse) compositional update and the corner-change test on lane 0
 *       (Homography.cc:73-92,109-114, Affine.cc:90-106,145-150, NT/FCLK.cc:314-339). */
/* Body of the device-side finish, executed by the first wave of the calling workgroup (all threads of the workgroup
 * must call it: it contains workgroup barriers). */
/* COH (k_track_persist): everything the loop rewrites between passes -- partial rows, warp, state, corners, counters, flags, the
 * Levenberg-Marquardt block -- is read and written with the coherent accessors of mtfhip_device.h (st_coh / ld_coh). */
template <bool COH = false>
__device__ __forceinline__ void finish_track_body(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts,
	const double *partials, int nblk, int t) {
	auto LD = [](const double *p) -> double { if constexpr (COH) return ld_coh(p); else return *p; };
	auto LDI = [](const int *p) -> int { if constexpr (COH) return ld_coh(p); else return *p; };
	auto ST = [](double *p, double v) { if constexpr (COH) st_coh(p, v); else *p = v; };
	auto STI = [](int *p, int v) { if constexpr (COH) st_coh(p, v); else *p = v; };
	FIN_STAMP(0);
	__shared__ double acc_s[NCC_ACC_COUNT];   /* >= ACC_COUNT */
	__shared__ double A[8][9];
	__shared__ double dps[8];
	__shared__ double h0s[64], Ws[9], crs[8], ics[12], tms[52], ncs[2];
	const int lane = threadIdx.x;
	const bool wv0 = lane < 64;
	const int S = bv.S;
	/* NCC: the reduced row holds raw moments (NCC_* slots, NCC_ACC_COUNT wide); tms = sum J0 | sum I0 J0 | Gram(J0) of the
	 * template, ncs = mean(I0), |I0 - mean|.  The calling workgroup then has at least 128 threads. */
	const bool ncc = bv.am == MTFHIP_AM_NCC;
	const int RL = ncc ? (int)NCC_ACC_COUNT : (int)ACC_COUNT;
	/* every global operand of this target -- the `active` flag included -- is requested up front, in parallel across
	 * the lanes, and only then is the flag tested: one memory round trip instead of two (flag, then operands); the
	 * rest of the routine runs out of LDS / registers */
	const int act = LDI(ts.active + t);
	const int n_it_prev = LDI(ts.n_iters + t);   /* passes done so far (uniform: a scalar load) */
	double v_h0 = 0, v_w = 0, v_cr = 0, v_ic = 0, v_acc = 0, v_tm = 0, v_nc = 0;
	if (wv0) {
		v_h0 = ts.h0[(size_t)t * 64 + lane];
		if (lane < 9) v_w = LD(bv.warps + 9 * t + lane);
		if (lane < 8) v_cr = LD(ts.corners + 8 * t + lane);
		if (lane < 12) v_ic = ts.init_corners_hm[12 * t + lane];
		if (ncc) {
			if (lane < 52) v_tm = ts.ncc_tm[(size_t)t * 52 + lane];
			if (lane < 2) v_nc = ts.ncc[(size_t)t * 8 + lane];
		}
	}
	/* fixed-order sum of the block rows.  Up to eight rows (the batched decomposition) one lane per column sums them in one round
	 * trip; a single large target has 157: the rows are cut into three contiguous runs (multiples of eight rows) summed by
	 * three groups of 80 lanes and combined in run order -- 7 rounds instead of 20.  (Workgroups of fewer than 240 threads keep
	 * the single run.) */
	__shared__ double part_s[3][80];
	const int n_runs = (nblk > 8 && blockDim.x >= 240) ? 3 : 1;
	const int run_len = n_runs == 1 ? nblk : ((nblk + 3 * 8 - 1) / (3 * 8)) * 8;
	{
		const int run = n_runs == 1 ? 0 : lane / 80, col = n_runs == 1 ? lane : lane % 80;
		if (col < RL && run < n_runs) {
			const int b0 = run * run_len, b1 = (b0 + run_len < nblk) ? b0 + run_len : nblk;
			const double *p = partials + (size_t)t * nblk * RL + col;
			auto ld = [&](size_t off) -> double { return LD(p + off); };
			/* eight block rows in flight per lane */
			double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
			int b = b0;
			for (; b + 7 < b1; b += 8) {
				s0 += ld((size_t)b * RL); s1 += ld((size_t)(b + 1) * RL);
				s2 += ld((size_t)(b + 2) * RL); s3 += ld((size_t)(b + 3) * RL);
				s4 += ld((size_t)(b + 4) * RL); s5 += ld((size_t)(b + 5) * RL);
				s6 += ld((size_t)(b + 6) * RL); s7 += ld((size_t)(b + 7) * RL);
			}
			for (; b < b1; ++b) s0 += ld((size_t)b * RL);
			const double v = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
			if (n_runs == 1) v_acc = v; else part_s[run][col] = v;
		}
	}
	if (!act) return;
	FIN_STAMP(1);
	if (wv0) {
		h0s[lane] = v_h0;
		if (lane < 9) Ws[lane] = v_w;
		if (lane < 8) crs[lane] = v_cr;
		if (lane < 12) ics[lane] = v_ic;
		if (lane < 52) tms[lane] = v_tm;
		if (lane < 2) ncs[lane] = v_nc;
	}
	if (n_runs > 1) {
		__syncthreads();
		if (lane < RL) v_acc = (part_s[0][lane] + part_s[1][lane]) + part_s[2][lane];
	}
	if (lane < RL) { acc_s[lane] = v_acc; ST(ts.acc + (size_t)t * RL + lane, v_acc); }
	__syncthreads();
	FIN_STAMP(2);
	const int i = (lane >> 3) & 7, j = lane & 7;
	/* ---- Levenberg-Marquardt: the accept / undo test on the similarity of this pass (uniform over the workgroup) ---- */
	double *lmp = ts.lm ? ts.lm + (size_t)t * kLmStride : nullptr;
	double lm_delta = 0.0;
	int lm_iter_id = n_it_prev;   /* (LM keeps its own counter) */
	bool undo = false;
	if (lmp) {
		const double f_now = ts.f_ext ? LD(ts.f_ext + t) : (ncc ? 0.0 : -acc_s[ACC_RR] / 2);
		const double prev_f = LD(lmp + 0);
		lm_delta = LD(lmp + 1);
		const bool state_reset = LD(lmp + 2) != 0.0;
		lm_iter_id = (int)LD(lmp + 3);
		double f_use = f_now;
		if (ncc) {
			const double nN0 = (double)bv.N, mt0 = acc_s[NCC_IT] / nN0, b20 = acc_s[NCC_IT2] - nN0 * mt0 * mt0;
			f_use = (acc_s[NCC_I0IT] - nN0 * ncs[0] * mt0) / (sqrt(b20) * ncs[1]);
		}
		if (!state_reset) {
			if (lm_iter_id > 0) {
				if (f_use < prev_f) { lm_delta *= sm.lm_delta_update; undo = true; }
				else if (f_use > prev_f) lm_delta /= sm.lm_delta_update;
			}
		}
		__syncthreads();   /* every thread has read the LM block before lane 0 rewrites it */
		if (lane == 0) {
			ST(lmp + 1, lm_delta);
			if (undo) ST(lmp + 2, 1.0);
			else { ST(lmp + 2, 0.0); if (!state_reset) ST(lmp + 0, f_use); }
		}
	}
	const bool use_h0 = (sm.hess_type == 0) || (sm.sm == MTFHIP_SM_ICLK && !ts.h_from_acc);
	const bool sum_h0 = (sm.sm == MTFHIP_SM_ESM) && (sm.hess_type == 2 || (sm.hess_type == 4 && !ts.h_from_acc));   /* (MI's SumOfStd arrives summed) */
	const double gscale = (sm.sm == MTFHIP_SM_ESM) ? 0.5 : 1.0;
	/* NCC from its moments (ncc_assemble in api_fused.hip is the host twin; formulas and citations there) */
	const double nN = (double)bv.N;
	const double n_mt = ncc ? acc_s[NCC_IT] / nN : 0.0, n_m0 = ncs[0], n_c = ncc ? ncs[1] : 1.0;
	const double n_b2 = ncc ? acc_s[NCC_IT2] - nN * n_mt * n_mt : 1.0, n_b = ncc ? sqrt(n_b2) : 1.0;
	const double n_f = ncc ? (acc_s[NCC_I0IT] - nN * n_m0 * n_mt) / (n_b * n_c) : 0.0;
	auto mom = [&](int which, int c0, int ct, int s) -> double {   /* which: 0 J0, 1 Jt, 2 their mean */
		const double v0 = c0 >= 0 ? tms[c0 + s] : acc_s[NCC_ITJ0 + s], vt = acc_s[ct + s];
		return which == 0 ? v0 : (which == 1 ? vt : (v0 + vt) / 2);
	};
	auto n_ut = [&](int which, int s) { return (mom(which, -1, NCC_ITJ, s) - n_mt * mom(which, 0, NCC_SJ, s)) / n_b2; };
	auto n_u0 = [&](int which, int s) { return (mom(which, 8, NCC_I0J, s) - n_m0 * mom(which, 0, NCC_SJ, s)) / (n_b * n_c); };
	auto n_hess = [&](int kind, int which, int r, int c, int kk) -> double {   /* kind: 0 init, 1 curr, 2 self */
		const double gram = which == 0 ? tms[16 + kk] : acc_s[NCC_GRAM + kk];
		const double G = -(gram - mom(which, 0, NCC_SJ, r) * mom(which, 0, NCC_SJ, c) / nN) / n_b2;
		const double utr = n_ut(which, r), utc = n_ut(which, c);
		if (kind == 2) return G + utr * utc;
		const double u0r = n_u0(which, r), u0c = n_u0(which, c);
		return n_f * G - utr * u0c - u0r * utc + 3 * (kind == 1 ? utr * utc : u0r * u0c);
	};
	auto h_entry = [&](int r, int c) -> double {
		if (r >= S || c >= S) return r == c ? -1.0 : 0.0;
		const int a = r < c ? r : c, b2 = r < c ? c : r;
		const int kk = a * 8 - (a * (a - 1)) / 2 + (b2 - a);
		if (ncc) {
			const int ht = sm.hess_type;
			const double h0v = h0s[b2 * S + a];
			/* sec_ord_hess (NCC.cc:391-410): the weighted pixel-Hessian sums of k_second_order_ssd's NCC form, entry (r, c) */
			const double ex = ts.h_extra ? ts.h_extra_scale * ts.h_extra[(size_t)t * S * S + c * S + r] : 0.0;
			if (ht == 0) return h0v;
			if (sm.sm == MTFHIP_SM_ICLK) return n_hess(0, 0, r, c, kk) + ex;
			if (sm.sm == MTFHIP_SM_FCLK || ht == 1 || ht == 5) return n_hess(ht == 1 ? 2 : 1, 1, r, c, kk) + ex;
			if (ht == 2) return 0.5 * (n_hess(2, 1, r, c, kk) + h0v);
			if (ht == 3) return n_hess(1, 2, r, c, kk) + ex;
			return 0.5 * (n_hess(0, 0, r, c, kk) + n_hess(1, 1, r, c, kk)) + ex;
		}
		/* (the constant Hessian as entry (r, c), not from a triangle: MI's initial self Hessian with its second-order part is not
		 * symmetric in the homography's last two rows / columns; every other one is, and reads the same numbers) */
		double v = use_h0 ? h0s[c * S + r] : -acc_s[ACC_H + kk];
		if (sum_h0) v = (v + h0s[c * S + r]) * 0.5;
		/* sec_ord_hess: + sum_p df_dI[p] d2I_dp2[:, p] (SSDBase.cc:313-415); the homography blocks are not symmetric in their
		 * last two rows / columns (Homography.cc:421,613,796), so the entry is taken as (r, c), not from a triangle */
		if (ts.h_extra) v += ts.h_extra_scale * ts.h_extra[(size_t)t * S * S + c * S + r];
		return v;
	};
	auto g_entry = [&](int s) -> double {
		if (!ncc) return gscale * acc_s[ACC_G + s];
		auto cj = [&](int which) { return n_u0(which, s) - n_f * n_ut(which, s); };
		auto ij = [&](int which) { return (n_b / n_c) * (n_ut(which, s) - n_f * n_u0(which, s)); };
		if (sm.sm == MTFHIP_SM_FCLK) return cj(1);
		if (sm.sm == MTFHIP_SM_ICLK) return ij(0);
		if (sm.jac_type == 0) return cj(2);
		return 0.5 * (cj(1) - ij(0));
	};
	/* Symmetric diagonal equilibration by POWERS OF TWO: the products are exact, so the scaled elimination follows the unscaled
	 * one bit for bit while its pivots stay near one (and a frexp / ldexp pair replaces the square root and the division of
	 * 1 / sqrt|d|, a microsecond of dependent arithmetic on an idle workgroup). */
	auto pow2_scale = [](double d) -> double { return d != 0 ? ldexp(1.0, -(ilogb(fabs(d)) / 2)) : 1.0; };
	/* Gauss-Jordan; with partial pivoting where the system can be indefinite: the SSD Hessians are negated Gram matrices (definite;
	 * a flat template's zero pivot leaves its unknown at zero), but NCC's Std / SumOfStd and MI's Hessians can be indefinite away
	 * from convergence, where the reference's colPivHouseholderQr (NT/FCLK.cc:298) does not care either.  (A row-per-lane
	 * elimination in registers with v_readlane broadcasts was measured slower than this LDS form: 8.9 k against 6.9 k clocks.) */
	const bool pivoting = ncc || ts.h_from_acc || ts.h_extra != nullptr;   /* (the search and the row swap are two barriers and eight LDS reads per step: skipped for SSD) */
	const double si = pow2_scale(h_entry(i, i)), sj = pow2_scale(h_entry(j, j));
	double *trec = (ts.trace && n_it_prev < ts.trace_cap) ? ts.trace + ((size_t)t * ts.trace_cap + n_it_prev) * kTraceStride : nullptr;   /* debug trace */
	if (wv0) {
		/* hessian(i, i) += leven_marq_delta * hessian(i, i) (NT/ESM.cc:262-265) */
		const double hij = h_entry(i, j), gi = (j == 0 && i < S) ? g_entry(i) : 0.0;
		A[i][j] = hij * si * sj * ((lmp && i == j && i < S) ? 1.0 + lm_delta : 1.0);
		if (j == 0) A[i][8] = gi * si;
		if (trec) {
			trec[8 * i + j] = (i < S && j < S) ? hij : 0.0;
			if (j == 0) trec[64 + i] = gi;
			if (lane == 0) {
				trec[88] = ts.f_ext ? LD(ts.f_ext + t) : (ncc ? n_f : -acc_s[ACC_RR] / 2);
				trec[90] = undo ? 1.0 : 0.0; trec[91] = lm_delta; trec[92] = 1.0;
			}
		}
	}
	__syncthreads();
	FIN_STAMP(3);
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		if (pivoting) {
			int pr = k;
			double best = fabs(A[k][k]);
#pragma unroll
			for (int r = k + 1; r < 8; ++r) { const double v = fabs(A[r][k]); if (v > best) { best = v; pr = r; } }
			const int src = i == k ? pr : (i == pr ? k : i);   /* row i after the swap of rows k and pr */
			const double mine = A[src][j], rhs = A[src][8];
			__syncthreads();
			if (wv0) { A[i][j] = mine; if (j == 0) A[i][8] = rhs; }
			__syncthreads();
		}
		const double piv = A[k][k], aik = A[i][k], akj = A[k][j], bk = A[k][8];
		const double f = (i != k && piv != 0) ? aik / piv : 0.0;
		__syncthreads();
		if (wv0 && i != k) {
			A[i][j] -= f * akj;
			if (j == 0) A[i][8] -= f * bk;
		}
		__syncthreads();
	}
	if (wv0 && j == 0) {
		const double d = A[i][i];
		dps[i] = (i < S && d != 0) ? -(A[i][8] / d) * si : 0.0;
	}
	__syncthreads();
	FIN_STAMP(4);
	/* ---- compositional update and corner test.  The expressions (and their order) are those of the straightforward serial form;
	 * what is independent is spread over the lanes -- the 9 + 9 + 8 IEEE divisions of invertState, of the homography
	 * normalisation and of the corner dehomogenisation were ~35 dependent division latencies on one lane of an otherwise
	 * idle workgroup, now four. ---- */
	__shared__ double xs[9], nxy[8];
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY;
	if (lane < 8) {
		const double v = undo ? LD(lmp + 4 + lane) : dps[lane];   /* undo: the previous state_update is taken back */
		if (lmp && !undo) ST(lmp + 4 + lane, v);
		dps[lane] = v;
		if (trec) trec[72 + lane] = v;
	}
	__syncthreads();
	double dp[8];
#pragma unroll
	for (int s = 0; s < 8; ++s) dp[s] = dps[s];
	double U[9];
	if (hom) {
		U[0] = 1 + dp[0]; U[1] = dp[1]; U[2] = dp[2]; U[3] = dp[3]; U[4] = 1 + dp[4]; U[5] = dp[5];
		U[6] = dp[6]; U[7] = dp[7]; U[8] = 1;
	} else {
		U[0] = 1 + dp[2]; U[1] = dp[3]; U[2] = dp[0]; U[3] = dp[4]; U[4] = 1 + dp[5]; U[5] = dp[1];
		U[6] = 0; U[7] = 0; U[8] = 1;
	}
	/* forward step: ICLK applies the inverse of the solved update (NT/ICLK.cc:266-267); undo: ESM / FCLK apply the inverse of
	 * the previous update (NT/ESM.cc:194-195), ICLK re-applies it (NT/ICLK.cc:188) */
	if ((sm.sm == MTFHIP_SM_ICLK) != undo) {
		/* invertState: inverse through cofactors, normalised by (2,2) */
		double c[9];
		c[0] = U[4] * U[8] - U[5] * U[7]; c[1] = U[2] * U[7] - U[1] * U[8]; c[2] = U[1] * U[5] - U[2] * U[4];
		c[3] = U[5] * U[6] - U[3] * U[8]; c[4] = U[0] * U[8] - U[2] * U[6]; c[5] = U[2] * U[3] - U[0] * U[5];
		c[6] = U[3] * U[7] - U[4] * U[6]; c[7] = U[1] * U[6] - U[0] * U[7]; c[8] = U[0] * U[4] - U[1] * U[3];
		double det = U[0] * c[0] + U[1] * c[3] + U[2] * c[6];
		double inv_det = 1.0 / det;
#pragma unroll
		for (int q = 0; q < 9; ++q) c[q] *= inv_det;
		double n22 = c[8];
		double cq = 0;
#pragma unroll
		for (int q = 0; q < 9; ++q) if (lane == q) cq = c[q];
		if (lane < 9) xs[lane] = cq / n22;
		__syncthreads();
#pragma unroll
		for (int q = 0; q < 9; ++q) U[q] = xs[q];
		__syncthreads();   /* xs is reused below */
		/* round-trip through the state parameterisation as getStateFromWarp / getWarpFromState do */
		U[0] = 1 + (U[0] - 1); U[4] = 1 + (U[4] - 1); U[8] = 1;
		if (!hom) { U[6] = 0; U[7] = 0; }
	}
	double Wo[9], Wn[9];
#pragma unroll
	for (int q = 0; q < 9; ++q) Wo[q] = Ws[q];
#pragma unroll
	for (int r = 0; r < 3; ++r)
#pragma unroll
		for (int c2 = 0; c2 < 3; ++c2)
			Wn[3 * r + c2] = Wo[3 * r] * U[c2] + Wo[3 * r + 1] * U[3 + c2] + Wo[3 * r + 2] * U[6 + c2];
	double *Wp = bv.warps + 9 * t, *st = bv.states + 8 * t;
	if (hom) {
		double n22 = Wn[8];
		double wq = 0;
#pragma unroll
		for (int q = 0; q < 9; ++q) if (lane == q) wq = Wn[q];
		if (lane < 9) xs[lane] = wq / n22;
		__syncthreads();
#pragma unroll
		for (int q = 0; q < 9; ++q) Wn[q] = xs[q];
		if (lane == 0) {
			ST(st + 0, Wn[0] - 1); ST(st + 1, Wn[1]); ST(st + 2, Wn[2]); ST(st + 3, Wn[3]); ST(st + 4, Wn[4] - 1); ST(st + 5, Wn[5]);
			ST(st + 6, Wn[6]); ST(st + 7, Wn[7]);
		}
	} else if (lane == 0) {
		ST(st + 0, Wn[2]); ST(st + 1, Wn[5]); ST(st + 2, Wn[0] - 1); ST(st + 3, Wn[1]); ST(st + 4, Wn[3]); ST(st + 5, Wn[4] - 1);
	}
	{
		double wq = 0;
#pragma unroll
		for (int q = 0; q < 9; ++q) if (lane == q) wq = Wn[q];
		if (lane < 9) ST(Wp + lane, wq);
	}
	double *cr = ts.corners + 8 * t;
	if (lane < 4) {
		const int q = lane;
		double X = ics[3 * q], Y = ics[3 * q + 1], Z = ics[3 * q + 2];
		double nx = Wn[0] * X + Wn[1] * Y + Wn[2] * Z, ny = Wn[3] * X + Wn[4] * Y + Wn[5] * Z;
		if (hom) {
			double d = Wn[6] * X + Wn[7] * Y + Wn[8] * Z;
			nx = nx / d; ny = ny / d;
		}
		nxy[2 * q] = nx; nxy[2 * q + 1] = ny;
		ST(cr + 2 * q, nx); ST(cr + 2 * q + 1, ny);
		if (trec) { trec[80 + 2 * q] = nx; trec[81 + 2 * q] = ny; }
	}
	__syncthreads();
	FIN_STAMP(5);
	if (lane != 0) return;
	double change = 0;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		double ddx = crs[2 * q] - nxy[2 * q], ddy = crs[2 * q + 1] - nxy[2 * q + 1];
		change += ddx * ddx + ddy * ddy;
	}
	const int n_it = n_it_prev + 1;   /* passes done (the reference's iters_done) */
	STI(ts.n_iters + t, n_it);
	if (trec) trec[89] = (double)n_it_prev;
	if (lmp) {
		/* an undo pass skips the convergence test (`continue`); it consumes an iteration in the for loops of ESM and ICLK
		 * (NT/ESM.cc:179, NT/ICLK.cc:169) but not in FCLK's while loop (NT/FCLK.cc:193-223) */
		const int id = lm_iter_id + ((undo && sm.sm == MTFHIP_SM_FCLK) ? 0 : 1);
		ST(lmp + 3, (double)id);
		if ((!undo && change < sm.epsilon) || id >= sm.max_iters) STI(ts.active + t, 0);
	} else if (change < sm.epsilon || n_it >= sm.max_iters) STI(ts.active + t, 0);
	FIN_STAMP(6);
}


/* ---------------------------------------------------------------------------------------------------------------------
 * Tolerance-mode finish (TrackState::fast_finish: MTFHIP_MATH_FAST, SSD family, first-order Hessian, no pivoting needed).
 * finish_track_body keeps the reference's expressions and IEEE divisions and eliminates in LDS with two barriers per pivot --
 * ~3.7 us of dependent latency behind the row sums on a workgroup that has nothing else to do.  Here every lane of wave 0 holds
 * the whole 8 x 8 system in registers and solves it redundantly: LDL^T without pivoting (the SSD Hessians are negated Gram
 * matrices; a zero pivot leaves its unknown at zero, as there), reciprocals by v_rcp_f64 + two Newton steps, the inverse of
 * the update through cofactors with the determinant cancelled, no equilibration (scaling by powers of two changes nothing
 * without pivoting).  No LDS and no barrier between the reduced row and the stores.  Same g, H, Levenberg-Marquardt logic,
 * update and convergence test as finish_track_body; results agree with it to rounding (~1e-13 relative on dp).
 * --------------------------------------------------------------------------------------------------------------------- */
template <bool COH = false>
__device__ __forceinline__ void finish_track_fast_body(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts,
	const double *partials, int nblk, int t) {
	auto LD = [](const double *p) -> double { if constexpr (COH) return ld_coh(p); else return *p; };
	auto LDI = [](const int *p) -> int { if constexpr (COH) return ld_coh(p); else return *p; };
	auto ST = [](double *p, double v) { if constexpr (COH) st_coh(p, v); else *p = v; };
	auto STI = [](int *p, int v) { if constexpr (COH) st_coh(p, v); else *p = v; };
	__shared__ double f_acc[ACC_COUNT], f_h0[64], f_w[9], f_cr[8], f_ic[12], f_lm[kLmStride + 1], f_part[3][80];
	const int lane = threadIdx.x;
	const bool wv0 = lane < 64;
	const int S = bv.S;
	constexpr int RL = ACC_COUNT;
	const int act = LDI(ts.active + t);
	FIN_STAMP(0);
	const int n_it_prev = LDI(ts.n_iters + t);
	double *lmp = ts.lm ? ts.lm + (size_t)t * kLmStride : nullptr;
	double v_h0 = 0, v_w = 0, v_cr = 0, v_ic = 0, v_acc = 0, v_lm = 0, v_f = 0;
	if (wv0) {
		/* (the Levenberg-Marquardt block and an external similarity ride on the same round trip) */
		if (lmp && lane < kLmStride) v_lm = LD(lmp + lane);
		if (ts.f_ext && lane == 0) v_f = LD(ts.f_ext + t);
		v_h0 = ts.h0[(size_t)t * 64 + lane];
		if (lane < 9) v_w = LD(bv.warps + 9 * t + lane);
		if (lane < 8) v_cr = LD(ts.corners + 8 * t + lane);
		if (lane < 12) v_ic = ts.init_corners_hm[12 * t + lane];
	}
	/* fixed-order sum of the block rows: the same runs and the same order as finish_track_body */
	const int n_runs = (nblk > 8 && blockDim.x >= 240) ? 3 : 1;
	const int run_len = n_runs == 1 ? nblk : ((nblk + 3 * 8 - 1) / (3 * 8)) * 8;
	{
		const int run = n_runs == 1 ? 0 : lane / 80, col = n_runs == 1 ? lane : lane % 80;
		if (col < RL && run < n_runs) {
			const int b0 = run * run_len, b1 = (b0 + run_len < nblk) ? b0 + run_len : nblk;
			const double *p = partials + (size_t)t * nblk * RL + col;
			auto ld = [&](size_t off) -> double { return LD(p + off); };
			double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
			int b = b0;
			for (; b + 7 < b1; b += 8) {
				s0 += ld((size_t)b * RL); s1 += ld((size_t)(b + 1) * RL);
				s2 += ld((size_t)(b + 2) * RL); s3 += ld((size_t)(b + 3) * RL);
				s4 += ld((size_t)(b + 4) * RL); s5 += ld((size_t)(b + 5) * RL);
				s6 += ld((size_t)(b + 6) * RL); s7 += ld((size_t)(b + 7) * RL);
			}
			for (; b < b1; ++b) s0 += ld((size_t)b * RL);
			const double v = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
			if (n_runs == 1) v_acc = v; else f_part[run][col] = v;
		}
	}
	if (!act) return;
	FIN_STAMP(1);
	if (wv0) {
		f_h0[lane] = v_h0;
		if (lane < 9) f_w[lane] = v_w;
		if (lane < 8) f_cr[lane] = v_cr;
		if (lane < 12) f_ic[lane] = v_ic;
		if (lane < kLmStride) f_lm[lane] = v_lm;
		if (lane == 0) f_lm[kLmStride] = v_f;
	}
	if (n_runs > 1) {
		__syncthreads();
		if (lane < RL) v_acc = (f_part[0][lane] + f_part[1][lane]) + f_part[2][lane];
	}
	if (lane < RL) { f_acc[lane] = v_acc; ST(ts.acc + (size_t)t * RL + lane, v_acc); }
	__syncthreads();
	if (!wv0) return;   /* the rest is wave 0's, without barriers */
	FIN_STAMP(2);
	/* ---- Levenberg-Marquardt accept / undo (NT/ESM.cc:186-232, NT/FCLK.cc:205-250, NT/ICLK.cc:181-199) ---- */
	double lm_delta = 0.0;
	int lm_iter_id = n_it_prev;
	bool undo = false;
	const double f_now = ts.f_ext ? f_lm[kLmStride] : -f_acc[ACC_RR] / 2;
	if (lmp) {
		const double prev_f = f_lm[0];
		lm_delta = f_lm[1];
		const bool state_reset = f_lm[2] != 0.0;
		lm_iter_id = (int)f_lm[3];
		if (!state_reset && lm_iter_id > 0) {
			if (f_now < prev_f) { lm_delta *= sm.lm_delta_update; undo = true; }
			else if (f_now > prev_f) lm_delta /= sm.lm_delta_update;
		}
		if (lane == 0) {
			ST(lmp + 1, lm_delta);
			if (undo) ST(lmp + 2, 1.0);
			else { ST(lmp + 2, 0.0); if (!state_reset) ST(lmp + 0, f_now); }
		}
	}
	const bool use_h0 = (sm.hess_type == 0) || (sm.sm == MTFHIP_SM_ICLK);
	const bool sum_h0 = (sm.sm == MTFHIP_SM_ESM) && (sm.hess_type == 2 || sm.hess_type == 4);
	const double gscale = (sm.sm == MTFHIP_SM_ESM) ? 0.5 : 1.0;
	/* ---- the system, lower triangle, in the registers of every lane ---- */
	/* (every LDS operand is requested unconditionally and up front -- one wait; guarded reads were 36 round trips) */
	double aH[36], aG[8], hI[36];
#pragma unroll
	for (int k = 0; k < 36; ++k) aH[k] = f_acc[ACC_H + k];
#pragma unroll
	for (int k = 0; k < 8; ++k) aG[k] = f_acc[ACC_G + k];
#pragma unroll
	for (int r = 0; r < 8; ++r)
#pragma unroll
		for (int c = 0; c <= r; ++c) hI[r * (r + 1) / 2 + c] = f_h0[(r * S + c) & 63];
	double M[8][8], y[8];
#pragma unroll
	for (int r = 0; r < 8; ++r) {
#pragma unroll
		for (int c = 0; c <= r; ++c) {
			const int kk = c * 8 - (c * (c - 1)) / 2 + (r - c);
			const double h0v = hI[r * (r + 1) / 2 + c];
			double v = use_h0 ? h0v : -aH[kk];
			if (sum_h0) v = (v + h0v) * 0.5;
			M[r][c] = r < S ? v : (r == c ? -1.0 : 0.0);   /* (c <= r) */
		}
		y[r] = r < S ? gscale * aG[r] : 0.0;
	}
	FIN_STAMP(3);
	double *trec = (ts.trace && n_it_prev < ts.trace_cap) ? ts.trace + ((size_t)t * ts.trace_cap + n_it_prev) * kTraceStride : nullptr;   /* debug trace */
	if (trec && lane == 0) {
#pragma unroll
		for (int r = 0; r < 8; ++r) {
#pragma unroll
			for (int c = 0; c < 8; ++c) trec[8 * r + c] = (r < S && c < S) ? (c <= r ? M[r][c] : M[c][r]) : 0.0;
			trec[64 + r] = y[r];
		}
		trec[88] = f_now; trec[90] = undo ? 1.0 : 0.0; trec[91] = lm_delta; trec[92] = 1.0;
	}
	double dp[8];
	if (!undo) {
		if (lmp) {   /* hessian(i, i) += leven_marq_delta * hessian(i, i) (NT/ESM.cc:262-265) */
#pragma unroll
			for (int r = 0; r < 8; ++r) if (r < S) M[r][r] *= 1.0 + lm_delta;
		}
		/* L D L^T in place (L below the diagonal, D on it), then L z = g, D w = z, L^T x = w; dp = -x */
		double inv[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const double d = M[k][k];
			inv[k] = d != 0 ? rcp_fast(d) : 0.0;
			double l[8];
#pragma unroll
			for (int i = k + 1; i < 8; ++i) l[i] = M[i][k] * inv[k];
#pragma unroll
			for (int i = k + 1; i < 8; ++i) {
#pragma unroll
				for (int j = k + 1; j <= i; ++j) M[i][j] = fma(-l[i], M[j][k], M[i][j]);
			}
#pragma unroll
			for (int i = k + 1; i < 8; ++i) { M[i][k] = l[i]; y[i] = fma(-l[i], y[k], y[i]); }
		}
#pragma unroll
		for (int k = 7; k >= 0; --k) {
			double x = y[k] * inv[k];
#pragma unroll
			for (int i = k + 1; i < 8; ++i) x = fma(-M[i][k], dp[i], x);
			dp[k] = x;
		}
#pragma unroll
		for (int k = 0; k < 8; ++k) dp[k] = k < S ? -dp[k] : 0.0;
		if (lmp) {
#pragma unroll
			for (int q = 0; q < 8; ++q) if (lane == q) ST(lmp + 4 + q, dp[q]);
		}
	} else {
		/* undo: the previous state_update is taken back */
#pragma unroll
		for (int q = 0; q < 8; ++q) dp[q] = f_lm[4 + q];
	}
	FIN_STAMP(4);
	/* ---- compositional update (Homography.cc:73-92,109-114, Affine.cc:90-106,145-150) and the corner test ---- */
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY;
	double U[9];
	if (hom) {
		U[0] = 1 + dp[0]; U[1] = dp[1]; U[2] = dp[2]; U[3] = dp[3]; U[4] = 1 + dp[4]; U[5] = dp[5]; U[6] = dp[6]; U[7] = dp[7]; U[8] = 1;
	} else {
		U[0] = 1 + dp[2]; U[1] = dp[3]; U[2] = dp[0]; U[3] = dp[4]; U[4] = 1 + dp[5]; U[5] = dp[1]; U[6] = 0; U[7] = 0; U[8] = 1;
	}
	/* ICLK applies the inverse of the solved update (NT/ICLK.cc:266-267); undo: ESM / FCLK apply the inverse of the previous
	 * update (NT/ESM.cc:194-195), ICLK re-applies it (NT/ICLK.cc:188) */
	if ((sm.sm == MTFHIP_SM_ICLK) != undo) {
		double c[9];   /* cofactors; normalised by the (2, 2) entry: the determinant cancels */
		c[0] = U[4] * U[8] - U[5] * U[7]; c[1] = U[2] * U[7] - U[1] * U[8]; c[2] = U[1] * U[5] - U[2] * U[4];
		c[3] = U[5] * U[6] - U[3] * U[8]; c[4] = U[0] * U[8] - U[2] * U[6]; c[5] = U[2] * U[3] - U[0] * U[5];
		c[6] = U[3] * U[7] - U[4] * U[6]; c[7] = U[1] * U[6] - U[0] * U[7]; c[8] = U[0] * U[4] - U[1] * U[3];
		const double ic8 = rcp_fast(c[8]);
#pragma unroll
		for (int q = 0; q < 8; ++q) U[q] = c[q] * ic8;
		U[8] = 1;
		if (!hom) { U[6] = 0; U[7] = 0; }
	}
	double Wn[9];
#pragma unroll
	for (int r = 0; r < 3; ++r) {
		const double w0 = f_w[3 * r], w1 = f_w[3 * r + 1], w2 = f_w[3 * r + 2];
#pragma unroll
		for (int c2 = 0; c2 < 3; ++c2) Wn[3 * r + c2] = fma(w0, U[c2], fma(w1, U[3 + c2], w2 * U[6 + c2]));
	}
	if (hom) {
		const double in22 = rcp_fast(Wn[8]);
#pragma unroll
		for (int q = 0; q < 8; ++q) Wn[q] *= in22;
		Wn[8] = 1;
	}
	double St[8];
	if (hom) { St[0] = Wn[0] - 1; St[1] = Wn[1]; St[2] = Wn[2]; St[3] = Wn[3]; St[4] = Wn[4] - 1; St[5] = Wn[5]; St[6] = Wn[6]; St[7] = Wn[7]; }
	else { St[0] = Wn[2]; St[1] = Wn[5]; St[2] = Wn[0] - 1; St[3] = Wn[1]; St[4] = Wn[3]; St[5] = Wn[4] - 1; St[6] = 0; St[7] = 0; }
	double Cr[8], change = 0;
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		const double X = f_ic[3 * q], Y = f_ic[3 * q + 1], Z = f_ic[3 * q + 2];
		double nx = fma(Wn[0], X, fma(Wn[1], Y, Wn[2] * Z)), ny = fma(Wn[3], X, fma(Wn[4], Y, Wn[5] * Z));
		if (hom) { const double idn = rcp_fast(fma(Wn[6], X, fma(Wn[7], Y, Wn[8] * Z))); nx *= idn; ny *= idn; }
		const double ddx = f_cr[2 * q] - nx, ddy = f_cr[2 * q + 1] - ny;
		change += ddx * ddx + ddy * ddy;
		Cr[2 * q] = nx; Cr[2 * q + 1] = ny;
	}
	FIN_STAMP(5);
	/* (every lane holds the same numbers: lane q stores entry q) */
	double wq = 0, sq = 0, cq = 0;
#pragma unroll
	for (int q = 0; q < 9; ++q) if (lane == q) wq = Wn[q];
#pragma unroll
	for (int q = 0; q < 8; ++q) if (lane == q) { sq = St[q]; cq = Cr[q]; }
	if (lane < 9) ST(bv.warps + 9 * t + lane, wq);
	if (lane < S) ST(bv.states + 8 * t + lane, sq);
	if (lane < 8) ST(ts.corners + 8 * t + lane, cq);
	if (lane != 0) return;
	if (trec) {
#pragma unroll
		for (int q = 0; q < 8; ++q) { trec[72 + q] = dp[q]; trec[80 + q] = Cr[q]; }
		trec[89] = (double)n_it_prev;
	}
	const int n_it = n_it_prev + 1;
	STI(ts.n_iters + t, n_it);
	if (lmp) {
		const int id = lm_iter_id + ((undo && sm.sm == MTFHIP_SM_FCLK) ? 0 : 1);
		ST(lmp + 3, (double)id);
		if ((!undo && change < sm.epsilon) || id >= sm.max_iters) STI(ts.active + t, 0);
	} else if (change < sm.epsilon || n_it >= sm.max_iters) STI(ts.active + t, 0);
	FIN_STAMP(6);
}

} // namespace mtfhip
#endif
