/*
 * kernels_mi_fused.hip -- the recompute form of the MI Lucas-Kanade iteration (MTFHIP_MATH_FAST; 8 bins, or up to 10: r06)
 * (one of the translation units of libmtfhip.so; conventions and the shared device helpers: mtfhip_device.h, mtfhip_mi_device.h)
 *
 * The materialising form (api_fused.hip::mi_enqueue with the kernels of kernels_mi.hip) runs the fused LK kernel first
 * -- it writes It, dIt_dx and Jt, 88 B/px -- and three more passes read those arrays back: 324 B/px moved for 132 B/px
 * of algorithmic traffic (profiles/r01_mi_kernel_stats.csv: 1.09 ms per iteration of 64 x 400 x 400 pixels).  Here nothing
 * per-pixel is written at all:
 *   pass 1  k_mi_pass_hist   warp + sample It (lean), B-spline windows of It and I0, histogram of It, joint histogram with
 *                            I0 and (self Hessians) with itself                    reads 28 B/px: texels 4, I0 8, grid point 16
 *           k_mi_tables_iter pre-seeding, logs, similarity, the gradient-factor tables (unchanged, kernels_mi.hip)
 *   pass 2  k_mi_pass_grad_hess   warp + sample It AND its gradient again, the steepest-descent row of the pixel and the
 *                            template's row rebuilt from dI0_dx in registers, both gradient vectors (MI.cc:406-415, 432-441),
 *                            their Jacobian products, and the first-order Hessian sums (MI.cc:461-637)   reads 44 B/px (+16 dI0_dx)
 *           k_mi_finish_fast fixed-order sum of the block rows, the Hessian assembly (MI.cc:497-511, 590-600, 626-636) and the
 *                            body of k_finish_track (solve, compositional update, convergence test) in ONE launch
 * 72 B/px instead of 324.  The per-pixel quantities are those of the reference up to the tolerance-mode sampling arithmetic
 * (mtfhip_device.h); the windows themselves are bit-identical given the same pixel value.
 *
 * Bin mode of both passes runs on v_mfma_f64_4x4x4_4b_f64 as in kernels_mi.hip (operand lane l: row / column = l & 3, block =
 * (l >> 2) & 3, k = l >> 4; result lane l: column = l & 3, block = (l >> 2) & 3, row = l >> 4; tools/mfma_layout_test_4x4.hip);
 * the rank-one sum  sum_p hess_term(p) J_p J_p^T  rides on the same staged rows as a ninth block product instead of 36
 * register accumulators per lane.
 */

#include "mtfhip_mi_fused_device.h"

namespace mtfhip {

/* ---------------------------------------------------------------------------------------------
 * pass 1: histograms of the freshly sampled It (MI.cc:346-367 update, :639-659 self); block rows [8 | 64 | 64 self]
 * ------------------------------------------------------------------------------------------- */
/* CAND: the candidate axis (PF / NN, SM/src/PF.cc:247-262 with MI as the appearance model): blockIdx.y is a candidate of target 0 --
 * its warp comes from the candidate's state, the template arrays are target 0's */
/* NB (r06): 8 = the r03-r05 kernel (pa.nb == 8: the 8 x 8 histograms are the four 4 x 4 blocks of ONE block product per step).  10: up to
 * ten bins at run time (pa.nb; the shipped mi_n_bins 10): the histograms come out of ONE 16 x 16 tile product per four pixels. */
template <int SSM, bool SELF, bool CAND = false, bool MC = false, int NB = 8>
__global__ __launch_bounds__(kBlock) void k_mi_pass_hist(BatchView bv, ImgView im, MiPassArgs pa, double *partials, int nblk, int row_len) {
	constexpr int kWinRows = NB == 8 ? 11 : NB + 2;   /* rows (bin + 1): row 0 and the rows behind bin nb - 1 take the taps outside the histogram */
	__shared__ __attribute__((aligned(16))) double slabs[4 * 2 * kWinRows * kRS];
	const int nb = NB == 8 ? 8 : pa.nb;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int t = blockIdx.y;
	if (!CAND && pa.active && !pa.active[t]) return;
	double *wa = slabs + (size_t)wave * 2 * kWinRows * kRS, *wb = wa + kWinRows * kRS;
	const unsigned N = (unsigned)bv.N;            /* rows: (pixel, channel) pairs */
	const unsigned NPt = MC ? (unsigned)bv.NP : N, Cc = MC ? (unsigned)bv.C : 1u;
	auto pix_of = [&](unsigned i) -> unsigned { if constexpr (MC) return Cc == 3u ? i / 3u : i / Cc; else return i; };
	const bool uz = bv.unit_z != 0;
	Warp9 W;
	if constexpr (CAND) {   /* getWarpFromState (Homography.cc:94-107, Affine.cc:116-130) of candidate t */
		const double *p = pa.cand_states + (size_t)t * (SSM == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6);
		if constexpr (SSM == MTFHIP_SSM_HOMOGRAPHY) {
			W.m[0] = 1 + p[0]; W.m[1] = p[1]; W.m[2] = p[2]; W.m[3] = p[3]; W.m[4] = 1 + p[4]; W.m[5] = p[5]; W.m[6] = p[6]; W.m[7] = p[7]; W.m[8] = 1;
		} else {
			W.m[0] = 1 + p[2]; W.m[1] = p[3]; W.m[2] = p[0]; W.m[3] = p[4]; W.m[4] = 1 + p[5]; W.m[5] = p[1]; W.m[6] = 0; W.m[7] = 0; W.m[8] = 1;
		}
	} else {
		W = load_warp(bv.warps + 9 * t);
	}
	const size_t tt = CAND ? 0 : (size_t)t;   /* whose template */
	const double *pp = bv.buf[uz ? MTFHIP_BUF_INIT_PTS : MTFHIP_BUF_INIT_HXY] + tt * 2 * NPt;
	const double *iz = bv.buf[MTFHIP_BUF_INIT_Z] + tt * NPt;
	const double *I0 = bv.buf[MTFHIP_BUF_I0] + tt * N;
	for (int k2 = 0; k2 < 2 * kWinRows; ++k2) wa[k2 * kRS + lane] = 0.0;   /* the slabs start clean and every chunk leaves them clean */
	double bj8 = 0.0, bs8 = 0.0, bh8 = 0.0;
	double cj3[3] = {0.0, 0.0, 0.0};   /* NB != 8: the joint histogram's blocks (lb, 0 .. 2) */
	const bool hfj = !CAND && pa.hist_from_joint != 0;
	const int li = lane & 3, lb = (lane >> 2) & 3, lk = lane >> 4;
	const double one0 = li == 0 ? 1.0 : 0.0;
	const unsigned stride = (unsigned)nblk * kBlock;
	unsigned base = (blockIdx.x * (kBlock / 64) + wave) * 64;
	/* Software pipeline with build-time depths (MTFHIP_MI_OP_AHEAD / MTFHIP_MI_TEX_AHEAD): the operands of chunk n + TD + PF are requested
	 * while chunk n is processed, the texels of chunk n + TD.  r05 measured deeper pipelines on the hypothesis that the pass waits for
	 * memory round trips (its sampling alone is 105 of its 120 us: tools/mi1_ablation.sh): (1, 1) 117.4 us, (3, 1) 121.0, (3, 2) 122.8,
	 * (3, 3) 125.3, (5, 2) 127.1 -- it does not; the per-pixel time is the same from 8 to 192 targets (MALL-resident or not), so the
	 * pass is bound by instruction issue, not by latency or bandwidth.  (1, 1) = the r03 depths stay. */
	constexpr int PF = kMiOpAhead, TD = kMiTexAhead;
	double2 qf[PF]; double zf[PF], i0f[PF + TD]; unsigned chf[PF + TD];
	/* every load of the loop is issued unconditionally (the third homogeneous coordinate is fetched from the template when it is
	 * not needed): with guarded loads the compiler cannot count what is in flight and waits with vmcnt(0), which exposes the
	 * full memory latency of the just-issued prefetch in front of every chunk (65 % of the wave cycles in the first version) */
	const double *zsrc = uz ? I0 : iz;
	auto fetch = [&](unsigned i, double2 &q, double &z, double &i0v, unsigned &ch) {
		const unsigned pi = pix_of(i);
		ch = 0;
		if constexpr (MC) ch = i - pi * Cc;
		q = ld_off<double2>(pp, pi * 16u); i0v = ld_off<double>(I0, i * 8u);
		z = ld_off<double>(zsrc, (uz ? i : pi) * 8u);
	};
	/* prologue: texels of the first TD chunks, operands of the PF chunks behind them.  i0f / chf: [0 .. TD) belong to the chunks whose
	 * texels are in flight (tx[0] = the chunk about to be processed), [TD .. TD + PF) to the operand ring */
	MiTex tx[TD];
#pragma unroll
	for (int k = 0; k < TD; ++k) {
		double2 q; double z;
		fetch(min(base + lane + (unsigned)k * stride, N - 1), q, z, i0f[k], chf[k]);
		tx[k] = mi_issue<SSM, false, MC>(im, W, q.x, q.y, z, uz, pa.grad_eps, Cc, chf[k]);
	}
#pragma unroll
	for (int k = 0; k < PF; ++k) fetch(min(base + lane + (unsigned)(TD + k) * stride, N - 1), qf[k], zf[k], i0f[TD + k], chf[TD + k]);
	for (; base < N; base += stride) {
		const unsigned i = base + lane;
		const double vm = i < N ? 1.0 : 0.0;   /* lanes behind the end of the patch carry zero weights */
		const double i0 = i0f[0];
		const unsigned ch_here = chf[0];
		/* texels of chunk n + TD (its operands are the oldest of the ring), operands of chunk n + TD + PF */
		const MiTex tx_new = mi_issue<SSM, false, MC>(im, W, qf[0].x, qf[0].y, zf[0], uz, pa.grad_eps, Cc, chf[TD]);
#pragma unroll
		for (int k = 0; k + 1 < PF; ++k) { qf[k] = qf[k + 1]; zf[k] = zf[k + 1]; }
#pragma unroll
		for (int k = 0; k + 1 < PF + TD; ++k) { i0f[k] = i0f[k + 1]; chf[k] = chf[k + 1]; }
		fetch(min(i + (unsigned)(TD + PF) * stride, N - 1), qf[PF - 1], zf[PF - 1], i0f[PF + TD - 1], chf[PF + TD - 1]);
		const MiSample sp = mi_finish<SSM, false, MC>(im, tx[0], pa.grad_eps, pa.norm_mult, pa.norm_add, i < N, (int)ch_here);
#pragma unroll
		for (int k = 0; k + 1 < TD; ++k) tx[k] = tx[k + 1];
		tx[TD - 1] = tx_new;
#if defined(MTFHIP_MI1_ABL) && MTFHIP_MI1_ABL >= 4   /* ablation builds (tools/mi1_ablation.sh): 4 sampling only */
		bj8 += sp.it + i0 * vm;
#else
		const BsplWin4 a = bspl_window4<false>(sp.it, nb, pa.hist_norm);
		const BsplWin4 b = bspl_window4<false>(i0, nb, pa.hist_norm);
#if defined(MTFHIP_MI1_ABL) && MTFHIP_MI1_ABL == 3   /* 3: + windows, no staging, no products */
		bj8 += (a.w[0] + a.w[1] + a.w[2] + a.w[3]) * vm + b.w[0] + b.w[1] + b.w[2] + b.w[3] + a.row0 + b.row0;
#else
		double *ra = wa + a.row0 * kRS + lane, *rb = wb + b.row0 * kRS + lane;
		/* NB != 8: rows 0 .. nb + 1 only (NB + 2 rows: three workgroups per CU as the 8-bin kernel; with NB + 3 the allocation granule made it two and
		 * the pass 224 us instead of 115) -- the one tap that can reach bin nb + 1 (the last of a window at fl = nb - 1) is stored onto the row of
		 * bin nb, which nobody reads either */
		const int ka3 = (NB != 8 && a.row0 + 3 > nb + 1) ? 2 * kRS : 3 * kRS, kb3 = (NB != 8 && b.row0 + 3 > nb + 1) ? 2 * kRS : 3 * kRS;
#pragma unroll
		for (int k = 0; k < 3; ++k) { ra[k * kRS] = a.w[k] * vm; rb[k * kRS] = b.w[k]; }
		ra[ka3] = a.w[3] * vm; rb[kb3] = b.w[3];
		__builtin_amdgcn_wave_barrier();
#if !(defined(MTFHIP_MI1_ABL) && MTFHIP_MI1_ABL == 2)   /* 2: + staging, no products */
		if constexpr (NB != 8) {
			/* Up to twelve bins as three groups of four: lane group lb (0..2; group 3 idles) owns the It-bins 4 lb .. 4 lb + 3 and keeps its A operand
			 * over the step's instructions -- three 4 x 4 x 4 block products against the three I0-bin groups give the joint histogram (block (lb, t)),
			 * the operand against itself the diagonal blocks of the self-joint histogram and against the next group's rows its first off-diagonal
			 * blocks; the self-joint histogram is BANDED (a window covers four consecutive bins: |r - c| <= 3), so those five blocks and their
			 * mirror images are all of it.  Five LDS reads and five block instructions (~19 cycles each, profiles/r04_fp64_rates.txt) per four pixels;
			 * r06's first forms: 3 x 3 + 3 x 3 block products with six + six reads 214 us, two 16 x 16 x 4 tiles (120-160 cycles each whatever is in
			 * them) 218 us, one tile + the band blocks 193 us -- against the 8-bin kernel's 115. */
			const int rA = 1 + 4 * lb + li, rS = 1 + 4 * (lb + 1) + li;
			const double *pa = wa + (rA < kWinRows ? rA : kWinRows - 1) * kRS + lk, *ps = wa + (rS < kWinRows ? rS : kWinRows - 1) * kRS + lk;   /* (rows past bin nb + 1 do not exist: their blocks are dropped at the end) */
			const double *pb0 = wb + (1 + li) * kRS + lk, *pb1 = wb + (5 + li) * kRS + lk, *pb2 = wb + ((9 + li < kWinRows) ? 9 + li : kWinRows - 1) * kRS + lk;
#pragma unroll
			for (int ks = 0; ks < 16; ++ks) {
				const double av = pa[4 * ks];
				cj3[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, pb0[4 * ks], cj3[0], 0, 0, 0);
				cj3[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, pb1[4 * ks], cj3[1], 0, 0, 0);
				cj3[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, pb2[4 * ks], cj3[2], 0, 0, 0);
				if (!hfj) bh8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, one0, bh8, 0, 0, 0);   /* (uniform; with partition of unity the histogram is the joint histogram's row sums) */
				if constexpr (SELF) {
					bs8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, av, bs8, 0, 0, 0);
					bj8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, ps[4 * ks], bj8, 0, 0, 0);   /* (bj8 is free at NB != 8: the first off-diagonal blocks) */
				}
			}
		} else
		if (hfj) {   /* (uniform) the histogram comes out of the joint histogram's rows at the end: two block products per step instead of three */
#pragma unroll
			for (int qq = 0; qq < 16; ++qq) {
				const int p = 4 * qq + lk;
				const double av = wa[(1 + 4 * (lb >> 1) + li) * kRS + p], bvv = wb[(1 + 4 * (lb & 1) + li) * kRS + p];
				bj8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bvv, bj8, 0, 0, 0);
				if constexpr (SELF) bs8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, wa[(1 + 4 * (lb & 1) + li) * kRS + p], bs8, 0, 0, 0);
			}
		} else {
#pragma unroll
		for (int qq = 0; qq < 16; ++qq) {
			const int p = 4 * qq + lk;
			const double av = wa[(1 + 4 * (lb >> 1) + li) * kRS + p], bvv = wb[(1 + 4 * (lb & 1) + li) * kRS + p];
			bj8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bvv, bj8, 0, 0, 0);
#if !(defined(MTFHIP_MI1_ABL) && MTFHIP_MI1_ABL == 1)   /* 1: no histogram product */
			bh8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, one0, bh8, 0, 0, 0);
#endif
			if constexpr (SELF) bs8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, wa[(1 + 4 * (lb & 1) + li) * kRS + p], bs8, 0, 0, 0);
		}
		}
#endif
		__builtin_amdgcn_wave_barrier();
#pragma unroll
		for (int k = 0; k < 3; ++k) { ra[k * kRS] = 0.0; rb[k * kRS] = 0.0; }   /* leave the slabs clean: 8 stores instead of 22 */
		ra[ka3] = 0.0; rb[kb3] = 0.0;
#endif
#endif
	}
	__syncthreads();
	double *red = slabs;
	const int rl = nb + nb * nb + (SELF ? nb * nb : 0);
	if constexpr (NB == 8) {
		const int r = 4 * (lb >> 1) + (lane >> 4), c = 4 * (lb & 1) + (lane & 3);
		red[wave * rl + nb + r * nb + c] = bj8;
		if constexpr (SELF) red[wave * rl + nb + nb * nb + r * nb + c] = bs8;
		if ((lb & 1) == 0 && (lane & 3) == 0) red[wave * rl + r] = bh8;
	} else {
		/* block lb of every accumulator: rows 4 lb + (lane >> 4); what lies outside the band of the self-joint histogram is zero */
		double *rj = red + wave * rl + nb, *rs = rj + nb * nb;
		if constexpr (SELF) { for (int k2 = lane; k2 < nb * nb; k2 += 64) rs[k2] = 0.0; }
		__builtin_amdgcn_wave_barrier();
		const int r = 4 * lb + (lane >> 4), ci = lane & 3;
		if (lb < 3 && r < nb) {
#pragma unroll
			for (int tb = 0; tb < 3; ++tb) { const int c = 4 * tb + ci; if (c < nb) rj[r * nb + c] = cj3[tb]; }
			if (ci == 0) red[wave * rl + r] = bh8;   /* (hfj: overwritten by the row sums below) */
			if constexpr (SELF) {
				const int cd = 4 * lb + ci, co = 4 * (lb + 1) + ci;
				if (cd < nb) rs[r * nb + cd] = bs8;
				if (co < nb) { rs[r * nb + co] = bj8; rs[co * nb + r] = bj8; }
			}
		}
	}
	__syncthreads();
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * row_len;
	for (int k2 = threadIdx.x; k2 < rl; k2 += kBlock) {
		double v = (red[k2] + red[rl + k2]) + (red[2 * rl + k2] + red[3 * rl + k2]);
		if (hfj && k2 < nb) {   /* histogram row k2 = sum over the columns of the joint histogram's row k2 (each the four waves' sum, fixed order) */
			v = 0.0;
			if constexpr (NB == 8) {
#pragma unroll
				for (int c = 0; c < 8; ++c) {
					const int e = nb + k2 * nb + c;
					v += (red[e] + red[rl + e]) + (red[2 * rl + e] + red[3 * rl + e]);
				}
			} else {
				for (int c = 0; c < nb; ++c) {
					const int e = nb + k2 * nb + c;
					v += (red[e] + red[rl + e]) + (red[2 * rl + e] + red[3 * rl + e]);
				}
			}
		}
		dst[k2] = v;
	}
}

/* ---------------------------------------------------------------------------------------------
 * candidate mode: the histogram rows of a candidate -> MI (MI.cc:369-381) -> its likelihood (MI.cc:384-387) -> the particle
 * weight (PF.cc:341-365).  One wave per candidate, lane = bin pair; the wave sum is a fixed butterfly.
 * ------------------------------------------------------------------------------------------- */
__global__ __launch_bounds__(64) void k_mi_cand_score(int nb, int n, int lo, const double *partials, int nblk, int row_len, const double *tb0, double pre_seed,
	double hist_norm, double alpha, int likelihood_func, double measurement_sigma, double max_similarity, double *wts, double *sim) {
	const int cnd = blockIdx.x, lane = threadIdx.x;
	if (cnd >= n) return;
	const double *p = partials + (size_t)cnd * nblk * row_len;
	double part = 0.0;
	if (nb == 8) {   /* (the r03 form: one bin pair per lane) */
		const int r = lane >> 3, c = lane & 7;
		const double js = column_sum(p + nb + lane, nblk, row_len);
		const double hs = lane < nb ? column_sum(p + lane, nblk, row_len) : 0.0;
		const double jv = (js + pre_seed) * hist_norm;
		const double lhc_own = lane < nb ? log((hs + nb * pre_seed) * hist_norm) : 0.0;   /* log curr_hist(lane) */
		const double lhc = __shfl(lhc_own, r);
		part = jv * (log(jv) - lhc - tb0[MI_LOG_INIT + c]);
	} else {
		const double hs = lane < nb ? column_sum(p + lane, nblk, row_len) : 0.0;
		const double lhc_own = lane < nb ? log((hs + nb * pre_seed) * hist_norm) : 0.0;
		for (int e0 = 0; e0 < nb * nb; e0 += 64) {   /* (uniform trip count: the shuffles below are taken by every lane) */
			const int e = e0 + lane, ec = e < nb * nb ? e : 0, r = ec / nb, c = ec - r * nb;
			const double js = column_sum(p + nb + ec, nblk, row_len);
			const double jv = (js + pre_seed) * hist_norm;
			const double lhc = __shfl(lhc_own, r);
			if (e < nb * nb) part += jv * (log(jv) - lhc - tb0[MI_LOG_INIT + c]);
		}
	}
#pragma unroll
	for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d);
	if (lane == 0) {
		const double f = part;
		const double dd = (1.0 / f) - 1;
		double w = exp(-alpha * dd * dd);
		if (likelihood_func != 0) {
			const double pi = 3.14159265358979323846;
			const double val = max_similarity - f;
			w = likelihood_func == 1 ? (1.0 / sqrt(2 * pi * measurement_sigma)) * exp(-0.5 * val / measurement_sigma) : 1.0 / (1.0 + val);
		}
		if (wts) wts[lo + cnd] = w;
		if (sim) sim[lo + cnd] = f;
	}
}

/* ---------------------------------------------------------------------------------------------
 * finish: block rows -> g, H of the search method -> (device-side loop) solve + update + convergence test.
 * out_H [B][64] column-major S x S Hessian of the pass, out_g [B][16] the two Jacobian products (for iterate's host solve).
 * gmode: 0 ICLK (df_dI0 . J0), 1 FCLK (df_dIt . Jt), 2 ESM Original (df_dIt . Jm), 3 ESM DiffOfJacs.
 * ------------------------------------------------------------------------------------------- */
template <int NB>
__global__ __launch_bounds__(kBlock) void k_mi_finish_fast(BatchView bv, mtfhip_sm_desc sm, TrackState ts, int have_hess, int transpose_q,
	int joint_off, int hist_off, int nb_rt, int gmode, int do_track, const double *partials, int nblk, const double *tb_all,
	double *out_H, double *out_g, double *rows) {
	constexpr int kRowF = mi_fast_row_nb(NB);
	__shared__ double gs[16], Hs[64], Q[NB * NB * 8], fac[NB == 8 ? 64 : 128];
	const int nb = NB == 8 ? 8 : nb_rt;
	const int t = blockIdx.x, S = bv.S;
	if (do_track && !ts.active[t]) return;   /* (uniform per workgroup) */
	const double *p = partials + (size_t)t * nblk * kRowF;
	const double *tb = tb_all + (size_t)t * MI_SIZE;
	const int ncol = have_hess ? kRowF : 16;
	if constexpr (NB == 8) {
	if (have_hess && threadIdx.x >= kBlock - 64) {
		/* the 64 bin-pair factors (1 / joint - 1 / hist), one per thread of the last wave, while the others sum the block rows:
		 * evaluated inside the assembly loop below they were 64 dependent global loads and 128 divisions per entry of H */
		const int k = threadIdx.x - (kBlock - 64), rr = k >> 3, cc = k & 7;
		const double jv = tb[joint_off + rr * MI_NB + cc];
		const double hv = tb[hist_off + (transpose_q ? cc : rr)];
		fac[k] = (1.0 / jv) - (1.0 / hv);
	}
	} else {
	if (have_hess && threadIdx.x >= kBlock - 128) {
		const int k = threadIdx.x - (kBlock - 128), rr = k / NB, cc = k % NB;
		if (k < NB * NB) {
			double fv = 0.0;   /* (pairs of bins beyond pa.nb: their Q rows are zero) */
			if (rr < nb && cc < nb) {
				const double jv = tb[joint_off + rr * MI_NB + cc];
				const double hv = tb[hist_off + (transpose_q ? cc : rr)];
				fv = (1.0 / jv) - (1.0 / hv);
			}
			fac[k] = fv;
		}
	}
	}
	for (int k = threadIdx.x; k < ncol; k += kBlock) {
		const double s = column_sum(p + k, nblk, kRowF);
		if (k < 16) gs[k] = s; else if (k < 80) Hs[k - 16] = s; else Q[k - 80] = s;
	}
	__syncthreads();
	double *row = rows + (size_t)t * ACC_COUNT;
	if (threadIdx.x < 64) {
		const int r2 = threadIdx.x >> 3, c2 = threadIdx.x & 7;
		double h = 0.0;
		if (have_hess && r2 < S && c2 < S) {
			const int a = r2 < c2 ? r2 : c2, b2 = r2 < c2 ? c2 : r2;
			h = Hs[a * 8 + b2];
			for (int rr = 0; rr < (NB == 8 ? 8 : nb); ++rr)
				for (int cc = 0; cc < (NB == 8 ? 8 : nb); ++cc) {
					/* MI.cc:497-511, 590-600, 626-636: joint_hist_jacobian.row(idx)^T row(idx) * (1 / joint - 1 / hist) */
					const double *q = Q + (rr * NB + cc) * 8;
					h += q[r2] * q[c2] * fac[rr * NB + cc];
				}
			out_H[(size_t)t * 64 + c2 * S + r2] = h;
		}
		if (r2 <= c2) row[ACC_H + r2 * 8 - (r2 * (r2 - 1)) / 2 + (c2 - r2)] = -h;
		if (threadIdx.x < 16) out_g[(size_t)t * 16 + threadIdx.x] = gs[threadIdx.x];
		if (threadIdx.x < ACC_COUNT - 36) row[36 + threadIdx.x] = 0.0;
	}
	__syncthreads();
	if (threadIdx.x < S) {
		const double gt = gs[threadIdx.x], g0 = gs[8 + threadIdx.x];
		row[ACC_G + threadIdx.x] = gmode == 0 ? g0 : (gmode == 1 ? gt : (gmode == 2 ? 2.0 * gt : gt - g0));
	}
	if (!do_track) return;
	__syncthreads();   /* the row is read back by the same workgroup */
	finish_track_body(bv, sm, ts, rows, 1, t);
}

/* The gradient-factor tables of an iteration as per-class polynomials for pass 2 (kMiPoly*, mtfhip_mi_fused_device.h): one workgroup per
 * target, 2 x 64 x 12 + 8 x 5 coefficients of 16 multiply-adds each -- negligible next to the passes, and it runs once per target
 * instead of once per workgroup of pass 2.  Tables are indexed with (bin + 1) and have a zero border of one bin below and three above
 * (window rows fl .. fl + 3 for fl <= 7): the taps on the non-existent bins -1, 8, 9 contribute nothing (MI.cc:114-117). */
/* POLY_ONLY = false (k_mi_tables_poly, r05): k_mi_tables_iter's work first -- block rows of pass 1 -> histograms, logs, similarity, the
 * gradient-factor tables (mi_tables_iter_body) -- then, behind a barrier (the same workgroup wrote the tables it now reads), the
 * polynomial tables: one launch between the passes instead of two (6.3 + 5.7 us each with a launch gap, of a 430 us iteration) */
template <bool POLY_ONLY, int NB = 8>
__device__ __forceinline__ void mi_poly_tables_body(const double *tb_all, double hist_norm, int with_self, double *poly_all, double *Tc, double *Ti, double *Th, int nb_rt = 8) {
	const int t = blockIdx.x;
	const int nb = NB == 8 ? 8 : nb_rt;
	constexpr int TS = NB + 4;   /* table rows / columns: (bin + 1) with a zero border of one bin below and three above */
	constexpr int kPolyH = mi_poly_h(NB), kPolySz = mi_poly_size_nb(NB), NP = NB * NB;
	const double *tb = tb_all + (size_t)t * MI_SIZE;
	for (int k = threadIdx.x; k < TS * TS; k += kBlock) {
		const int r = k / TS - 1, c = k % TS - 1;
		const bool in = r >= 0 && r < nb && c >= 0 && c < nb;
		Tc[k] = in ? tb[MI_T_CURR + r * MI_NB + c] : 0.0; Ti[k] = in ? tb[MI_T_INIT + r * MI_NB + c] : 0.0;
		Th[k] = (in && with_self) ? tb[MI_T_SELF + r * MI_NB + c] : 0.0;
	}
	__syncthreads();
	constexpr MiPolyCoef CF = mi_poly_coef();
	double *out = poly_all + (size_t)t * kPolySz;
	for (int k = threadIdx.x; k < 2 * NP * kMiPolyPair; k += kBlock) {
		const int which = k / (NP * kMiPolyPair), e = k % (NP * kMiPolyPair);
		const int pair = e / kMiPolyPair, ab = e % kMiPolyPair, a = ab >> 2, b = ab & 3;
		const int f1 = pair / NB, f2 = pair % NB;   /* rows by the differentiated window's class, columns by the other's */
		const double *T = which ? Ti : Tc;
		double acc = 0.0;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			double dc = 0.0;
#pragma unroll
			for (int aa = 0; aa < 3; ++aa) dc = a == aa ? CF.d[r][aa] : dc;
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				double wc = 0.0;
#pragma unroll
				for (int bb = 0; bb < 4; ++bb) wc = b == bb ? CF.w[c][bb] : wc;
				acc = fma(dc * wc, T[(f1 + r) * TS + f2 + c], acc);
			}
		}
		out[k] = acc * hist_norm;
	}
	for (int k = threadIdx.x; k < NB * 8; k += kBlock) {
		const int fl = k >> 3, j = k & 7;
		double acc = 0.0;
		if (j < 5) {
#pragma unroll
			for (int r = 0; r < 4; ++r)
#pragma unroll
				for (int c = 0; c < 4; ++c)
#pragma unroll
					for (int a = 0; a < 2; ++a) {
						const int b = j - a;
						if (b >= 0 && b < 4) {
							double wc = 0.0;
#pragma unroll
							for (int bb = 0; bb < 4; ++bb) wc = b == bb ? CF.w[c][bb] : wc;
							acc = fma(CF.h[r][a] * wc, Th[(fl + r) * TS + fl + c], acc);
						}
					}
		}
		out[kPolyH + k] = acc * hist_norm;
	}
}
__global__ __launch_bounds__(kBlock) void k_mi_poly_tables(const double *tb_all, double hist_norm, int with_self, double *poly_all) {
	__shared__ double Tc[12 * 12], Ti[12 * 12], Th[12 * 12];
	mi_poly_tables_body<true>(tb_all, hist_norm, with_self, poly_all, Tc, Ti, Th);
}
template <int NB>
__global__ __launch_bounds__(kBlock) void k_mi_tables_poly(int nb, double pre_seed, double norm_mult, int with_self, const double *partials, int nblk, int row_len,
	double *tb_all, double *f_out, double *poly_all) {
	__shared__ double red[kBlock];
	__shared__ double Tc[(NB + 4) * (NB + 4)], Ti[(NB + 4) * (NB + 4)], Th[(NB + 4) * (NB + 4)];
	mi_tables_iter_body(nb, pre_seed, norm_mult, with_self, partials, nblk, row_len, tb_all, f_out, red);
	__threadfence_block();
	__syncthreads();
	mi_poly_tables_body<false, NB>(tb_all, norm_mult, with_self, poly_all, Tc, Ti, Th, nb);
}

/* ===================================================================== */
/* launchers                                                              */
/* ===================================================================== */
static MiPassArgs make_args(const MiFastPlan &pl) {
	MiPassArgs pa;
	pa.nb = pl.nb; pa.j0_mode = pl.j0_mode; pa.j0_init_variant = pl.j0_init_variant; pa.need_dft = pl.need_dft; pa.need_df0 = pl.need_df0;
	pa.g_mean = pl.g_mean; pa.table_off = 0; pa.transpose_q = pl.hk == 3; pa.nonchained = pl.nonchained; pa.hist_from_joint = pl.hist_from_joint;
	pa.grad_eps = pl.grad_eps; pa.norm_mult = pl.norm_mult; pa.norm_add = pl.norm_add; pa.hist_norm = pl.hist_norm;
	pa.active = pl.active; pa.tb = pl.tb; pa.poly = pl.poly; pa.cand_states = nullptr;
	return pa;
}
template <int SSM, bool MC>
static void launch_pass1(const BatchView &bv, const ImgView &im, const MiPassArgs &pa, bool self, double *partials, int nblk, int row_len, hipStream_t st) {
	const dim3 g = grid2(nblk, bv.B);
	if constexpr (!MC) {
		if (pa.nb != 8) {   /* up to ten bins (single channel) */
			if (self) MTFHIP_LAUNCH((k_mi_pass_hist<SSM, true, false, false, 10>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
			else MTFHIP_LAUNCH((k_mi_pass_hist<SSM, false, false, false, 10>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
			return;
		}
	}
	if (self) MTFHIP_LAUNCH((k_mi_pass_hist<SSM, true, false, MC>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	else MTFHIP_LAUNCH((k_mi_pass_hist<SSM, false, false, MC>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
}
void launch_mi_pass_hist(const BatchView &bv, const ImgView &im, const MiFastPlan &pl, double *partials, int nblk, int row_len, hipStream_t st) {
	const MiPassArgs pa = make_args(pl);
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY, self = pl.hk == 1, mc = bv.C > 1;
	if (hom && mc) launch_pass1<MTFHIP_SSM_HOMOGRAPHY, true>(bv, im, pa, self, partials, nblk, row_len, st);
	else if (hom) launch_pass1<MTFHIP_SSM_HOMOGRAPHY, false>(bv, im, pa, self, partials, nblk, row_len, st);
	else if (mc) launch_pass1<MTFHIP_SSM_AFFINE, true>(bv, im, pa, self, partials, nblk, row_len, st);
	else launch_pass1<MTFHIP_SSM_AFFINE, false>(bv, im, pa, self, partials, nblk, row_len, st);
}
/* MI over the candidate axis: candidates [lo, lo + cnt) of dev_states ([.][S]), weights / similarities at their global indices */
void launch_mi_score_candidates(const BatchView &bv, const ImgView &im, const MiFastPlan &pl, const double *dev_states, int lo, int cnt,
	double *partials, int nblk, int row_len, double pre_seed, double alpha, int likelihood_func, double measurement_sigma, double max_similarity,
	double *wts, double *sim, hipStream_t st) {
	if (cnt <= 0) return;
	MiPassArgs pa = make_args(pl);
	pa.active = nullptr;
	pa.cand_states = dev_states + (size_t)lo * bv.S;
	const dim3 g(nblk, cnt);
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY, mc = bv.C > 1;
	if (!mc && pa.nb != 8) {
		if (hom) MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_HOMOGRAPHY, false, true, false, 10>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
		else MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_AFFINE, false, true, false, 10>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	}
	else if (hom && mc) MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_HOMOGRAPHY, false, true, true>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	else if (hom) MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_HOMOGRAPHY, false, true, false>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	else if (mc) MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_AFFINE, false, true, true>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	else MTFHIP_LAUNCH((k_mi_pass_hist<MTFHIP_SSM_AFFINE, false, true, false>), g, dim3(kBlock), 0, st, bv, im, pa, partials, nblk, row_len);
	MTFHIP_LAUNCH(k_mi_cand_score, dim3(cnt), dim3(64), 0, st, pl.nb, cnt, lo, (const double *)partials, nblk, row_len, pl.tb, pre_seed, pl.hist_norm, alpha,
		likelihood_func, measurement_sigma, max_similarity, wts, sim);
}
void launch_mi_poly_tables(const BatchView &bv, const double *tb, double hist_norm, int with_self, double *poly, hipStream_t st) {
	MTFHIP_LAUNCH(k_mi_poly_tables, dim3(bv.B), dim3(kBlock), 0, st, tb, hist_norm, with_self, poly);
}
void launch_mi_tables_poly(const BatchView &bv, int nb, double pre_seed, double norm_mult, int with_self, const double *partials, int nblk, int row_len,
	double *tb, double *f_out, double *poly, hipStream_t st) {
	if (nb == 8) MTFHIP_LAUNCH(k_mi_tables_poly<8>, dim3(bv.B), dim3(kBlock), 0, st, nb, pre_seed, norm_mult, with_self, partials, nblk, row_len, tb, f_out, poly);
	else MTFHIP_LAUNCH(k_mi_tables_poly<10>, dim3(bv.B), dim3(kBlock), 0, st, nb, pre_seed, norm_mult, with_self, partials, nblk, row_len, tb, f_out, poly);
}
int mi_poly_size(int nb) { return nb == 8 ? mi_poly_size_nb(8) : mi_poly_size_nb(10); }
/* pass 2 is instantiated per (SSM, channels) in its own translation unit: kernels_mi_pass2_*.hip */
void launch_mi_pass2_hom(const BatchView &bv, const ImgView &im, const MiPassArgs &pa, int hk, int hrow, double *partials, int nblk, hipStream_t st);
void launch_mi_pass2_aff(const BatchView &bv, const ImgView &im, const MiPassArgs &pa, int hk, int hrow, double *partials, int nblk, hipStream_t st);
void launch_mi_pass2_hom_mc(const BatchView &bv, const ImgView &im, const MiPassArgs &pa, int hk, int hrow, double *partials, int nblk, hipStream_t st);
void launch_mi_pass2_aff_mc(const BatchView &bv, const ImgView &im, const MiPassArgs &pa, int hk, int hrow, double *partials, int nblk, hipStream_t st);
void launch_mi_pass2_hom_nb10(const BatchView &bv, const ImgView &im, const MiPassArgs &pa, int hk, int hrow, double *partials, int nblk, hipStream_t st);
void launch_mi_pass2_aff_nb10(const BatchView &bv, const ImgView &im, const MiPassArgs &pa, int hk, int hrow, double *partials, int nblk, hipStream_t st);
void launch_mi_pass_grad_hess(const BatchView &bv, const ImgView &im, const MiFastPlan &pl, double *partials, int nblk, hipStream_t st) {
	const MiPassArgs pa = make_args(pl);
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY, mc = bv.C > 1;
	if (pl.nb != 8) { (hom ? launch_mi_pass2_hom_nb10 : launch_mi_pass2_aff_nb10)(bv, im, pa, pl.hk, pl.hrow, partials, nblk, st); return; }
	(hom ? (mc ? launch_mi_pass2_hom_mc : launch_mi_pass2_hom) : (mc ? launch_mi_pass2_aff_mc : launch_mi_pass2_aff))(bv, im, pa, pl.hk, pl.hrow, partials, nblk, st);
}
void launch_mi_finish_fast(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts, const MiFastPlan &pl, int gmode, int do_track,
	const double *partials, int nblk, double *out_H, double *out_g, double *rows, hipStream_t st) {
	const int joint = pl.hk == 1 ? MI_SELF_JOINT : MI_JOINT, hist = pl.hk == 3 ? MI_HIST_INIT : MI_HIST_CURR;
	if (pl.nb == 8) MTFHIP_LAUNCH(k_mi_finish_fast<8>, dim3(bv.B), dim3(kBlock), 0, st, bv, sm, ts, pl.hk != 0, pl.hk == 3, joint, hist, 8, gmode, do_track,
		partials, nblk, pl.tb, out_H, out_g, rows);
	else MTFHIP_LAUNCH(k_mi_finish_fast<10>, dim3(bv.B), dim3(kBlock), 0, st, bv, sm, ts, pl.hk != 0, pl.hk == 3, joint, hist, pl.nb, gmode, do_track,
		partials, nblk, pl.tb, out_H, out_g, rows);
}
int mi_fast_row_len(int nb) { return nb == 8 ? mi_fast_row_nb(8) : mi_fast_row_nb(10); }

} // namespace mtfhip
