/*
 * kernels_mi.hip -- mutual information (AM/src/MI.cc): B-spline Parzen histograms, gradients, first-order Hessians
 * (one of the translation units of libmtfhip.so; conventions and the shared device helpers: mtfhip_device.h)
 */
#include "mtfhip_mi_device.h"

namespace mtfhip {

/* histogram of A and joint histogram A x B (MI.cc:222-235 init, :245-252 init joint, :352-367 update,
 * :641-649 self).  Block partial rows: [nb hist | nb*nb joint] */
/* SELF (MFMA only): the joint histogram of A with itself (cmptSelfHist MI.cc:639-659) is accumulated in the same pass and
 * written behind the ordinary row: [nb hist | nb*nb joint | nb*nb self] */
template <bool MFMA, bool SELF = false>
__global__ __launch_bounds__(kBlock) void k_mi_hist(int N, int nb, double norm_mult, const double *A_all,
	const double *B_all, double *partials, int nblk, int row_len) {
	extern __shared__ __attribute__((aligned(16))) double dyn[];
	constexpr int RS = MFMA ? kMiRowMfma : kMiRow;   /* as in k_mi_hess */
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	/* slabs are bin-major with rows of RS = 65 doubles: pixel-mode lanes write consecutive words, bin-mode lanes
	 * (different r, same p) land in different banks */
	double *wa = dyn + (size_t)wave * 2 * nb * RS;  /* [nb][65] dense A weights of this wave's chunk */
	double *wb = wa + nb * RS;                      /* [nb][65] dense B weights */
	const int t = blockIdx.y;
	const double *A = A_all + (size_t)t * N, *Bv = B_all + (size_t)t * N;
	double accj[kMiPairs], acch = 0.0;
	mfma_d4 cj = {0.0, 0.0, 0.0, 0.0}, cs = {0.0, 0.0, 0.0, 0.0};
	double bj8 = 0.0, bs8 = 0.0, bh8 = 0.0;   /* nb == 8: the 8 x 8 tables as four 4x4x4 blocks (Rg, Cg), one value per lane */
	const bool blocks8 = MFMA && nb == 8;
#pragma unroll
	for (int m = 0; m < kMiPairs; ++m) accj[m] = 0.0;
	int pr[kMiPairs], pc[kMiPairs];
#pragma unroll
	for (int m = 0; m < kMiPairs; ++m) { const int q = lane + 64 * m; pr[m] = q < nb * nb ? q / nb : -1; pc[m] = q < nb * nb ? q % nb : 0; }
	/* a wave walks only a handful of chunks and each needs its pixel values first: the next chunk's are requested
	 * before the current one is processed, otherwise every chunk starts with an exposed HBM round trip */
	int base = (blockIdx.x * (kBlock / 64) + wave) * 64;
	double a_nx = 0.0, b_nx = 0.0;
	if (base + lane < N) { a_nx = A[base + lane]; b_nx = Bv[base + lane]; }
	for (; base < N; base += nblk * kBlock) {
		const int i = base + lane;
		const double a_cur = a_nx, b_cur = b_nx;
		{
			const int in = i + nblk * kBlock;
			if (in < N) { a_nx = A[in]; b_nx = Bv[in]; }
		}
		for (int k2 = 0; k2 < nb; ++k2) { wa[k2 * RS + lane] = 0.0; wb[k2 * RS + lane] = 0.0; }
		if (i < N) {
			const BsplWin a = bspl_window(a_cur, nb, norm_mult, false);
			const BsplWin b = bspl_window(b_cur, nb, norm_mult, false);
			/* static indices only: a runtime-indexed window array would live in scratch memory */
#pragma unroll
			for (int r = 0; r < 4; ++r) if (r < a.n) wa[(a.lo + r) * RS + lane] = a.w[r];
#pragma unroll
			for (int c = 0; c < 4; ++c) if (c < b.n) wb[(b.lo + c) * RS + lane] = b.w[c];
		}
		__builtin_amdgcn_wave_barrier();
		if (blocks8) {
			/* v_mfma_f64_4x4x4_4b_f64 (operand layout: k_mi_hess): block b = (Rg, Cg) holds joint(4 Rg + i, 4 Cg + j) -- one
			 * instruction per four pixels gives the whole 8 x 8 table, where the 16 x 16 tile used a quarter of its outputs;
			 * a third operand of ones in column 0 gives the histogram */
			const int li = lane & 3, lb = (lane >> 2) & 3, lk = lane >> 4;
			const double one0 = li == 0 ? 1.0 : 0.0;
#pragma unroll
			for (int q = 0; q < 16; ++q) {
				const int p = 4 * q + lk;
				const double av = wa[(4 * (lb >> 1) + li) * RS + p], bv = wb[(4 * (lb & 1) + li) * RS + p];
				bj8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, bj8, 0, 0, 0);
				bh8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, one0, bh8, 0, 0, 0);
				if constexpr (SELF) bs8 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, wa[(4 * (lb & 1) + li) * RS + p], bs8, 0, 0, 0);
			}
		} else if constexpr (MFMA) {
			/* one 16x16 tile: rows r, columns c; when nb < 16 column nb of B is all ones, so D[r][nb] is the histogram */
			const int idx = lane & 15, kq = lane >> 4, row = idx < nb ? idx : nb - 1;
#pragma unroll
			for (int ks = 0; ks < 16; ++ks) {
				const int p = 4 * ks + kq;
				const double av = wa[row * RS + p], bv = wb[row * RS + p];
				cj = __builtin_amdgcn_mfma_f64_16x16x4f64(idx < nb ? av : 0.0, idx < nb ? bv : (idx == nb ? 1.0 : 0.0), cj, 0, 0, 0);
				if constexpr (SELF) cs = __builtin_amdgcn_mfma_f64_16x16x4f64(idx < nb ? av : 0.0, idx < nb ? av : 0.0, cs, 0, 0, 0);
			}
			if (nb == 16) {
#pragma unroll 8
				for (int p = 0; p < 64; ++p) if (lane < nb) acch += wa[lane * RS + p];
			}
		} else {
#pragma unroll 8
			for (int p = 0; p < 64; ++p) {
#pragma unroll
				for (int m = 0; m < kMiPairs; ++m)
					if (pr[m] >= 0) accj[m] = fma(wa[pr[m] * RS + p], wb[pc[m] * RS + p], accj[m]);
				if (lane < nb) acch += wa[lane * RS + p];
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	/* four waves -> one partial row per workgroup */
	__syncthreads();
	double *red = dyn;                                  /* [4][nb + nb*nb (+ nb*nb)], the slabs are free now */
	const int rl = nb + nb * nb + (SELF ? nb * nb : 0);
	if (blocks8) {
		/* result lane l: column j = l & 3, block = (l >> 2) & 3, row i = l >> 4 */
		const int lb = (lane >> 2) & 3, r = 4 * (lb >> 1) + (lane >> 4), c = 4 * (lb & 1) + (lane & 3);
		red[wave * rl + nb + r * nb + c] = bj8;
		if constexpr (SELF) red[wave * rl + nb + nb * nb + r * nb + c] = bs8;
		if ((lb & 1) == 0 && (lane & 3) == 0) red[wave * rl + r] = bh8;
	} else if constexpr (MFMA) {
		const int j = lane & 15;
#pragma unroll
		for (int v = 0; v < 4; ++v) {
			const int i = (lane >> 4) + 4 * v;
			if (i < nb && j < nb) red[wave * rl + nb + i * nb + j] = cj[v];
			if (i < nb && j == nb) red[wave * rl + i] = cj[v];
			if constexpr (SELF) { if (i < nb && j < nb) red[wave * rl + nb + nb * nb + i * nb + j] = cs[v]; }
		}
		if (nb == 16 && lane < nb) red[wave * rl + lane] = acch;
	} else {
		if (lane < nb) red[wave * rl + lane] = acch;
#pragma unroll
		for (int m = 0; m < kMiPairs; ++m)
			if (pr[m] >= 0) red[wave * rl + nb + pr[m] * nb + pc[m]] = accj[m];
	}
	__syncthreads();
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * row_len;
	for (int k2 = threadIdx.x; k2 < rl; k2 += kBlock) dst[k2] = (red[k2] + red[rl + k2]) + (red[2 * rl + k2] + red[3 * rl + k2]);
}
/* sums the block rows, applies pre-seeding and normalisation, logs, similarity and the gradient-factor
 * table of the requested flavour (MI.cc:237-262, 369-381, 310-314, 399-403, 427-431, 651-658).
 * mode 0: initialise (A = B = I0), 1: update (A = It, B = I0), 2: self (A = B = It) */
__global__ __launch_bounds__(kBlock) void k_mi_hist_finish(int nb, double pre_seed, double norm_mult, int mode, int first_init,
	const double *partials, int nblk, int row_len, double *tb_all, double *f_out) {
	__shared__ double red[kBlock];
	const int t = blockIdx.x;
	double *tb = tb_all + (size_t)t * MI_SIZE;
	const double *p = partials + (size_t)t * nblk * row_len;
	const double hist_seed = nb * pre_seed;
	for (int k = threadIdx.x; k < nb + nb * nb; k += kBlock) {
		const double s = column_sum(p + k, nblk, row_len);
		if (k < nb) {
			const double hv = (s + hist_seed) * norm_mult;
			if (mode == 0) { tb[MI_HIST_INIT + k] = hv; tb[MI_LOG_INIT + k] = log(hv); if (first_init) { tb[MI_HIST_CURR + k] = hv; tb[MI_LOG_CURR + k] = log(hv); } }
			else if (mode == 1) { tb[MI_HIST_CURR + k] = hv; tb[MI_LOG_CURR + k] = log(hv); }
		} else {
			const int q = k - nb, r = q / nb, c = q % nb;
			const double jv = (s + pre_seed) * norm_mult;
			if (mode == 2) tb[MI_SELF_JOINT + r * MI_NB + c] = jv;
			else if (mode == 1 || first_init) { tb[MI_JOINT + r * MI_NB + c] = jv; tb[MI_JOINT_LOG + r * MI_NB + c] = log(jv); }
		}
	}
	__syncthreads();
	double part = 0;
	for (int q = threadIdx.x; q < nb * nb; q += kBlock) {
		const int r = q / nb, c = q % nb;
		if (mode == 2) {
			const double lg = log(tb[MI_SELF_JOINT + r * MI_NB + c]);
			tb[MI_T_SELF + r * MI_NB + c] = 1 + lg - tb[MI_LOG_CURR + r];
		} else if (mode == 1 || first_init) {
			const double jv = tb[MI_JOINT + r * MI_NB + c], lg = tb[MI_JOINT_LOG + r * MI_NB + c];
			const double lr = mode == 0 ? tb[MI_LOG_INIT + r] : tb[MI_LOG_CURR + r];
			part += jv * (lg - lr - tb[MI_LOG_INIT + c]);
			if (mode == 0) {
				/* MI::initializeGrad MI.cc:310-314: both tables start as 1 + log(joint/init_hist(row)) */
				const double v = 1 + lg - tb[MI_LOG_INIT + r];
				tb[MI_T_INIT + r * MI_NB + c] = v; tb[MI_T_CURR + r * MI_NB + c] = v;
			}
		}
	}
	red[threadIdx.x] = part;
	__syncthreads();
	if (threadIdx.x == 0 && mode != 2) {
		double s = 0;
		for (int i = 0; i < kBlock; ++i) s += red[i];
		f_out[t] = s;
	}
}
/* Fused MI iteration: everything between the histogram pass and the gradient pass in ONE launch -- k_mi_hist_finish in
 * update mode (histogram of It, joint, logs, similarity: MI.cc:369-381), optionally in self mode (MI.cc:651-658), and the
 * two k_mi_factor tables (MI.cc:399-403, 427-431).  Rows: [nb hist | nb*nb joint | nb*nb self (with_self)]. */
__global__ __launch_bounds__(kBlock) void k_mi_tables_iter(int nb, double pre_seed, double norm_mult, int with_self, const double *partials,
	int nblk, int row_len, double *tb_all, double *f_out) {
	__shared__ double red[kBlock];
	mi_tables_iter_body(nb, pre_seed, norm_mult, with_self, partials, nblk, row_len, tb_all, f_out, red);
}
/* gradient-factor tables refreshed by updateCurrGrad / updateInitGrad (MI.cc:399-403, 427-431) */
__global__ __launch_bounds__(kBlock) void k_mi_factor(int nb, int curr, double *tb_all) {
	double *tb = tb_all + (size_t)blockIdx.x * MI_SIZE;
	for (int q = threadIdx.x; q < nb * nb; q += kBlock) {
		const int r = q / nb, c = q % nb;
		if (curr) tb[MI_T_CURR + r * MI_NB + c] = 1 + tb[MI_JOINT_LOG + r * MI_NB + c] - tb[MI_LOG_CURR + r];
		else tb[MI_T_INIT + r * MI_NB + c] = 1 + tb[MI_JOINT_LOG + c * MI_NB + r] - tb[MI_LOG_INIT + r]; /* (init, curr) indexing */
	}
}
/* df_dI[p] = sum_r sum_c gradA(r,p) * matB(c,p) * T(r,c)  (MI.cc:318-326, 406-415, 432-441) */
__global__ __launch_bounds__(kBlock) void k_mi_grad(int N, int nb, double norm_mult, const double *A_all,
	const double *B_all, const double *tb_all, int table_off, double *out_all) {
	__shared__ double T[MI_NB * MI_NB];
	const int t = blockIdx.y;
	const double *tb = tb_all + (size_t)t * MI_SIZE + table_off;
	for (int k = threadIdx.x; k < MI_NB * MI_NB; k += kBlock) T[k] = tb[k];
	__syncthreads();
	const double *A = A_all + (size_t)t * N, *Bv = B_all + (size_t)t * N;
	double *out = out_all + (size_t)t * N;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		const BsplWin a = bspl_window(A[i], nb, norm_mult, false);
		const BsplWin b = bspl_window(Bv[i], nb, norm_mult, false);
		double acc = 0;
#pragma unroll
		for (int r = 0; r < 4; ++r)
#pragma unroll
			for (int c = 0; c < 4; ++c)
				if (r < a.n && c < b.n) acc += a.d[r] * b.w[c] * T[(a.lo + r) * MI_NB + b.lo + c];
		out[i] = acc;
	}
}
/* Fused gradient pass of an MI iteration: both gradient vectors per pixel --
 *   df_dIt = sum gradIt(r) matI0(c) T_curr(r, c),  df_dI0 = sum gradI0(r) matIt(c) T_init(r, c)   (MI.cc:406-415, 432-441)
 * -- and the Jacobian products df_dIt . Jt, df_dI0 . J0 (cmptCurrJacobian / cmptInitJacobian) in the same pass, so that
 * 2 x k_mi_grad, k_gemv and its reduction are one launch and It, I0 are read once.  Block partial rows: [8 | 8]. */
__global__ __launch_bounds__(kBlock) void k_mi_grad_gemv(int N, int S, int nb, double norm_mult, const double *It_all, const double *I0_all,
	const double *tb_all, const double *Jt_all, const double *J0_all, MiJ0Rebuild rb, double *dft_out, double *df0_out, double *partials, int nblk) {
	__shared__ double Tc[MI_NB * MI_NB], Ti[MI_NB * MI_NB];
	__shared__ double lds[4 * 16];
	const int t = blockIdx.y;
	const double *tb = tb_all + (size_t)t * MI_SIZE;
	for (int k = threadIdx.x; k < MI_NB * MI_NB; k += kBlock) { Tc[k] = tb[MI_T_CURR + k]; Ti[k] = tb[MI_T_INIT + k]; }
	__syncthreads();
	const double *It = It_all + (size_t)t * N, *I0 = I0_all + (size_t)t * N;
	const double *Jt = Jt_all ? Jt_all + (size_t)t * N * S : nullptr, *J0 = J0_all ? J0_all + (size_t)t * N * S : nullptr;
	double acc[16];
#pragma unroll
	for (int k = 0; k < 16; ++k) acc[k] = 0.0;
	for (int i = blockIdx.x * kBlock + threadIdx.x; i < N; i += gridDim.x * kBlock) {
		double jt[kMaxS], j0[kMaxS];
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) { jt[s] = (Jt && s < S) ? Jt[(size_t)s * N + i] : 0.0; j0[s] = (J0 && !rb.dI0 && s < S) ? J0[(size_t)s * N + i] : 0.0; }
		if (J0 && rb.dI0) {
			/* the template's steepest-descent row from dI0_dx and the grid point (32-40 B instead of 8 S), as the fused LK
			 * kernel rebuilds it: cmptWarpedPixJacobian at the identity warp (Homography.cc:231-294: gradient / z) after a
			 * chained initialize, cmptInitPixJacobian (:157-191) otherwise; Affine.cc:160-182, 213-242 coincide there */
			const double *g0 = rb.dI0 + (size_t)t * 2 * N;
			const double g0x = g0[i], g0y = g0[(size_t)N + i];
			const double2 p0 = reinterpret_cast<const double2 *>(rb.pts + (size_t)t * 2 * N)[i];
			if (rb.hom) {
				double Ix0 = g0x, Iy0 = g0y;
				if (!rb.init_variant) { const double inv = 1.0 / (rb.z ? rb.z[(size_t)t * N + i] : 1.0); Ix0 = g0x * inv; Iy0 = g0y * inv; }
				hom_row(j0, Ix0, Iy0, p0.x, p0.y, p0.x, p0.y);
			} else {
				j0[0] = g0x; j0[1] = g0y; j0[2] = g0x * p0.x; j0[3] = g0x * p0.y; j0[4] = g0y * p0.x; j0[5] = g0y * p0.y; j0[6] = j0[7] = 0.0;
			}
		}
		const BsplWin a = bspl_window(It[i], nb, norm_mult, false);
		const BsplWin c0 = bspl_window(I0[i], nb, norm_mult, false);
		double dft = 0, df0 = 0;
#pragma unroll
		for (int r = 0; r < 4; ++r)
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				if (r < a.n && c < c0.n) dft += a.d[r] * c0.w[c] * Tc[(a.lo + r) * MI_NB + c0.lo + c];
				if (r < c0.n && c < a.n) df0 += c0.d[r] * a.w[c] * Ti[(c0.lo + r) * MI_NB + a.lo + c];
			}
		if (dft_out) dft_out[(size_t)t * N + i] = dft;
		if (df0_out) df0_out[(size_t)t * N + i] = df0;
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) { acc[s] = fma(dft, jt[s], acc[s]); acc[8 + s] = fma(df0, j0[s], acc[8 + s]); }
	}
	block_reduce_store<16>(acc, partials + ((size_t)t * nblk + blockIdx.x) * 16, lds);
}
/* first-order MI Hessians (MI.cc:461-513 init, 565-601 self (the returned pass), 603-637 curr):
 *   Hsum  += hess_term(p) * Jrow Jrow^T,  hess_term = sum_r hessA(r) * sum_c matB(c) T(r,c)
 *   Q[row(r,c)] += gradA(r) matB(c) Jrow          row(r,c) = (r,c), or (c,r) when transpose_q (init flavour)
 * Block partial rows: [36 Hsum | nb*nb*S Q] */
template <bool MFMA>   /* MFMA: nb == 8 (the reference's 8-bin histograms): 64 (r, c) rows x 8 columns out of 4x4x4 blocks */
__global__ __launch_bounds__(kBlock) void k_mi_hess(int N, int S, int nb, double norm_mult, const double *A_all,
	const double *B_all, const double *tb_all, int table_off, int transpose_q, const double *J_all,
	double *partials, int nblk, int row_len) {
	extern __shared__ __attribute__((aligned(16))) double dyn[];
	/* slab row stride: 65 keeps the VALU bin mode conflict-free; the MFMA form reads rows b, b + 1 .. at pixels p, p + 1 ..
	 * in one instruction and wants 8 b + 2 k distinct banks: 68 (136 dwords = 8 mod 64) */
	constexpr int RS = MFMA ? kMiRowMfma : kMiRow;
	double *T = dyn;                                    /* MI_NB*MI_NB gradient-factor table */
	double *red = dyn + MI_NB * MI_NB;                  /* 4 * 36 */
	double *slabs = red + 4 * 36;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int slab = RS * (2 * nb + kMaxS);
	double *gd = slabs + (size_t)wave * slab;           /* [nb][65] dense curr_hist_grad-type vector of A */
	double *wd = gd + nb * RS;                      /* [nb][65] dense weights of B */
	double *rw = wd + nb * RS;                      /* [kMaxS][65] J rows */
	const int t = blockIdx.y;
	const double *tb = tb_all + (size_t)t * MI_SIZE + table_off;
	for (int k2 = threadIdx.x; k2 < MI_NB * MI_NB; k2 += kBlock) T[k2] = tb[k2];
	__syncthreads();
	const double *A = A_all + (size_t)t * N, *Bv = B_all + (size_t)t * N;
	const double *J = J_all + (size_t)t * N * S;
	double acc[36];
#pragma unroll
	for (int k2 = 0; k2 < 36; ++k2) acc[k2] = 0.0;
	constexpr int NQ = MFMA ? 1 : kMiPairs;
	double accq[NQ][kMaxS];
	int pr[NQ], pc[NQ];
	double cq[8];   /* MFMA: accumulator (Rg, Cg, Sg) of the 4x4x4 blocks */
#pragma unroll
	for (int mt = 0; mt < 8; ++mt) cq[mt] = 0.0;
#pragma unroll
	for (int m = 0; m < NQ; ++m) {
		const int q = lane + 64 * m;
		pr[m] = q < nb * nb ? q / nb : -1; pc[m] = q < nb * nb ? q % nb : 0;
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) accq[m][s] = 0.0;
	}
	/* operands of the next chunk (two pixel values, S Jacobian entries) are requested before the current one is processed */
	int base = (blockIdx.x * (kBlock / 64) + wave) * 64;
	double a_nx = 0.0, b_nx = 0.0, row_nx[kMaxS];
#pragma unroll
	for (int s = 0; s < kMaxS; ++s) row_nx[s] = 0.0;
	if (base + lane < N) {
		a_nx = A[base + lane]; b_nx = Bv[base + lane];
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) if (s < S) row_nx[s] = J[(size_t)s * N + base + lane];
	}
	for (; base < N; base += nblk * kBlock) {
		const int i = base + lane;
		const double a_cur = a_nx, b_cur = b_nx;
		double row[kMaxS];
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) row[s] = i < N ? row_nx[s] : 0.0;
		{
			const int in = i + nblk * kBlock;
			if (in < N) {
				a_nx = A[in]; b_nx = Bv[in];
#pragma unroll
				for (int s = 0; s < kMaxS; ++s) if (s < S) row_nx[s] = J[(size_t)s * N + in];
			}
		}
		for (int k2 = 0; k2 < nb; ++k2) { gd[k2 * RS + lane] = 0.0; wd[k2 * RS + lane] = 0.0; }
		if (i < N) {
			/* pixel mode: windows, the scalar hess_term and its rank-1 contribution (MI.cc:478-496, 574-583, 620-629) */
			const BsplWin a = bspl_window(a_cur, nb, norm_mult, true);
			const BsplWin b = bspl_window(b_cur, nb, norm_mult, false);
			double hess_term = 0;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				if (r < a.n) {
					double inner = 0;
#pragma unroll
					for (int c = 0; c < 4; ++c) if (c < b.n) inner += b.w[c] * T[(a.lo + r) * MI_NB + b.lo + c];
					hess_term += a.h[r] * inner;
					gd[(a.lo + r) * RS + lane] = a.d[r];
				}
			}
#pragma unroll
			for (int c = 0; c < 4; ++c) if (c < b.n) wd[(b.lo + c) * RS + lane] = b.w[c];
			int k2 = 0;
#pragma unroll
			for (int x = 0; x < kMaxS; ++x) {
				const double hx = hess_term * row[x];
#pragma unroll
				for (int y = x; y < kMaxS; ++y) { acc[k2] = fma(hx, row[y], acc[k2]); ++k2; }
			}
		}
#pragma unroll
		for (int s = 0; s < kMaxS; ++s) rw[s * RS + lane] = row[s];
		__builtin_amdgcn_wave_barrier();
		/* bin mode: joint_hist_jacobian.row(r, c) += grad(r, p) * mat(c, p) * J.row(p)  (MI.cc:484-486, 576-577, 622-623) */
		if constexpr (MFMA) {
			/* v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 blocks per instruction, no idle output columns (the 16x16x4
			 * tile had 8 of 16): half the matrix-core time for the same LDS reads.  Operand lane l: row / column = l & 3,
			 * block = (l >> 2) & 3, k = l >> 4; result lane l: column = l & 3, block = (l >> 2) & 3, row = l >> 4
			 * (tools/mfma_layout_test_4x4.hip).  block <-> r = 4 Rg + b, row <-> c = 4 Cg + i, column <-> s = 4 Sg + j,
			 * k <-> pixel 4 q + k; eight accumulators (Rg, Cg, Sg). */
			const int li = lane & 3, lb = (lane >> 2) & 3, lk = lane >> 4;
#pragma unroll
			for (int q = 0; q < 16; ++q) {
				const int p = 4 * q + lk;
				const double g0 = gd[lb * RS + p], g1 = gd[(4 + lb) * RS + p];
				const double w0 = wd[li * RS + p], w1 = wd[(4 + li) * RS + p];
				const double j0 = rw[li * RS + p], j1 = rw[(4 + li) * RS + p];
				const double a00 = g0 * w0, a01 = g0 * w1, a10 = g1 * w0, a11 = g1 * w1;
				cq[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a00, j0, cq[0], 0, 0, 0);
				cq[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a00, j1, cq[1], 0, 0, 0);
				cq[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01, j0, cq[2], 0, 0, 0);
				cq[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a01, j1, cq[3], 0, 0, 0);
				cq[4] = __builtin_amdgcn_mfma_f64_4x4x4f64(a10, j0, cq[4], 0, 0, 0);
				cq[5] = __builtin_amdgcn_mfma_f64_4x4x4f64(a10, j1, cq[5], 0, 0, 0);
				cq[6] = __builtin_amdgcn_mfma_f64_4x4x4f64(a11, j0, cq[6], 0, 0, 0);
				cq[7] = __builtin_amdgcn_mfma_f64_4x4x4f64(a11, j1, cq[7], 0, 0, 0);
			}
		} else {
#pragma unroll 4
			for (int p = 0; p < 64; ++p) {
				double jr[kMaxS];
#pragma unroll
				for (int s = 0; s < kMaxS; ++s) jr[s] = rw[s * RS + p];
#pragma unroll
				for (int m = 0; m < NQ; ++m) {
					if (pr[m] >= 0) {
						const double gr = gd[pr[m] * RS + p] * wd[pc[m] * RS + p];
#pragma unroll
						for (int s = 0; s < kMaxS; ++s) accq[m][s] = fma(gr, jr[s], accq[m][s]);
					}
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	}
	double *dst = partials + ((size_t)t * nblk + blockIdx.x) * row_len;
	const int ql = nb * nb * S;
	block_reduce_store<36>(acc, dst, red);
	__syncthreads();
	/* the four waves' Q blocks through the (now free) slabs: [4][nb*nb*S], indexed as the finish expects */
	double *qred = slabs;
	if constexpr (MFMA) {
		const int li = lane & 3, lb = (lane >> 2) & 3, lk = lane >> 4;
#pragma unroll
		for (int a8 = 0; a8 < 8; ++a8) {
			const int r = 4 * (a8 >> 2) + lb, c = 4 * ((a8 >> 1) & 1) + lk, sx = 4 * (a8 & 1) + li;
			const int row_idx = transpose_q ? c * nb + r : r * nb + c;
			if (sx < S) qred[wave * ql + row_idx * S + sx] = cq[a8];
		}
	} else {
#pragma unroll
		for (int m = 0; m < NQ; ++m) {
			if (pr[m] >= 0) {
				const int row_idx = transpose_q ? pc[m] * nb + pr[m] : pr[m] * nb + pc[m];
#pragma unroll
				for (int s = 0; s < kMaxS; ++s) if (s < S) qred[wave * ql + row_idx * S + s] = accq[m][s];
			}
		}
	}
	__syncthreads();
	for (int k2 = threadIdx.x; k2 < ql; k2 += kBlock)
		dst[36 + k2] = (qred[k2] + qred[ql + k2]) + (qred[2 * ql + k2] + qred[3 * ql + k2]);
}
__global__ __launch_bounds__(kBlock) void k_mi_hess_finish(int S, int nb, const double *partials, int nblk, int row_len,
	const double *tb_all, int joint_off, int hist_off, int transpose_q, double *out) {
	extern __shared__ __attribute__((aligned(16))) double dyn[];
	double *Q = dyn;               /* nb*nb*S */
	double *Hs = dyn + nb * nb * S; /* 36 */
	const int t = blockIdx.x;
	const double *p = partials + (size_t)t * nblk * row_len;
	const double *tb = tb_all + (size_t)t * MI_SIZE;
	for (int k = threadIdx.x; k < 36 + nb * nb * S; k += kBlock) {
		const double s = column_sum(p + k, nblk, row_len);
		if (k < 36) Hs[k] = s; else Q[k - 36] = s;
	}
	__syncthreads();
	if (threadIdx.x < 64) {
		const int r2 = threadIdx.x >> 3, c2 = threadIdx.x & 7;
		if (r2 < S && c2 < S) {
			const int a = r2 < c2 ? r2 : c2, b2 = r2 < c2 ? c2 : r2;
			double h = Hs[a * 8 - (a * (a - 1)) / 2 + (b2 - a)];
			for (int rr = 0; rr < nb; ++rr)
				for (int cc = 0; cc < nb; ++cc) {
					/* Q row (rr,cc) is joint_hist_jacobian.row(linear_idx(rr,cc)); its factor uses joint(rr,cc) and the
					 * histogram of the image whose gradient was taken: rows for curr/self, columns for the init flavour */
					const double jv = tb[joint_off + rr * MI_NB + cc];
					const double hv = tb[hist_off + (transpose_q ? cc : rr)];
					const double fac = (1.0 / jv) - (1.0 / hv);
					const double *q = Q + (size_t)(rr * nb + cc) * S;
					h += q[r2] * q[c2] * fac;
				}
			out[(size_t)t * 64 + c2 * S + r2] = h;
		}
	}
}


/* ===================================================================== */
/* launchers                                                              */
/* ===================================================================== */
void launch_mi_hist(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, double *partials,
	int nblk, int row_len, hipStream_t st) {
	/* per-wave staging slabs [64][2 nb]; the same LDS later holds the four waves' [nb + nb^2] rows */
	const size_t lds = sizeof(double) * std::max<size_t>((size_t)4 * kMiRowMfma * 2 * nb, (size_t)4 * (nb + nb * nb));
	static const bool use_mfma = !(getenv("MTFHIP_MI_MFMA") && atoi(getenv("MTFHIP_MI_MFMA")) == 0);
	if (use_mfma) MTFHIP_LAUNCH((k_mi_hist<true, false>), grid2(nblk, bv.B), dim3(kBlock), lds, st, bv.N, nb, norm_mult, A, Bv, partials, nblk, row_len);
	else MTFHIP_LAUNCH((k_mi_hist<false, false>), grid2(nblk, bv.B), dim3(kBlock), lds, st, bv.N, nb, norm_mult, A, Bv, partials, nblk, row_len);
}
/* A = It against B = I0 and against itself in one pass (fused MI iteration); row: [nb | nb*nb | nb*nb] */
void launch_mi_hist_self(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, double *partials,
	int nblk, int row_len, hipStream_t st) {
	const size_t lds = sizeof(double) * std::max<size_t>((size_t)4 * kMiRowMfma * 2 * nb, (size_t)4 * (nb + 2 * nb * nb));
	MTFHIP_LAUNCH((k_mi_hist<true, true>), grid2(nblk, bv.B), dim3(kBlock), lds, st, bv.N, nb, norm_mult, A, Bv, partials, nblk, row_len);
}
void launch_mi_tables_iter(const BatchView &bv, int nb, double pre_seed, double norm_mult, int with_self, const double *partials, int nblk,
	int row_len, double *tb, double *f_out, hipStream_t st) {
	MTFHIP_LAUNCH(k_mi_tables_iter, dim3(bv.B), dim3(kBlock), 0, st, nb, pre_seed, norm_mult, with_self, partials, nblk, row_len, tb, f_out);
}
void launch_mi_hist_finish(const BatchView &bv, int nb, double pre_seed, double norm_mult, int mode, int first_init,
	const double *partials, int nblk, int row_len, double *tb, double *f_out, hipStream_t st) {
	MTFHIP_LAUNCH(k_mi_hist_finish, dim3(bv.B), dim3(kBlock), 0, st, nb, pre_seed, norm_mult, mode, first_init, partials,
		nblk, row_len, tb, f_out);
}
void launch_mi_factor(const BatchView &bv, int nb, int curr, double *tb, hipStream_t st) {
	MTFHIP_LAUNCH(k_mi_factor, dim3(bv.B), dim3(kBlock), 0, st, nb, curr, tb);
}
void launch_mi_grad(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, const double *tb,
	int table_off, double *out, hipStream_t st) {
	MTFHIP_LAUNCH(k_mi_grad, grid2(simple_blocks_per_target(bv.N), bv.B), dim3(kBlock), 0, st, bv.N, nb, norm_mult, A, Bv,
		tb, table_off, out);
}
void launch_mi_hess(const BatchView &bv, int nb, double norm_mult, const double *A, const double *Bv, const double *tb,
	int table_off, int transpose_q, const double *J, double *partials, int nblk, int row_len, hipStream_t st) {
	const size_t slabs = std::max<size_t>((size_t)4 * kMiRowMfma * (2 * nb + kMaxS), (size_t)4 * nb * nb * bv.S);
	const size_t lds = sizeof(double) * (MI_NB * MI_NB + 4 * 36 + slabs);
	static bool attr_set = false;
	if (!attr_set) {   /* 16 bins need 82 KB of dynamic LDS; the default cap is 64 KB */
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_mi_hess<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		(void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_mi_hess<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
		attr_set = true;
	}
	static const bool use_mfma = !(getenv("MTFHIP_MI_MFMA") && atoi(getenv("MTFHIP_MI_MFMA")) == 0);
	if (use_mfma && nb == 8)
		MTFHIP_LAUNCH(k_mi_hess<true>, grid2(nblk, bv.B), dim3(kBlock), lds, st, bv.N, bv.S, nb, norm_mult, A, Bv, tb, table_off,
			transpose_q, J, partials, nblk, row_len);
	else
		MTFHIP_LAUNCH(k_mi_hess<false>, grid2(nblk, bv.B), dim3(kBlock), lds, st, bv.N, bv.S, nb, norm_mult, A, Bv, tb, table_off,
			transpose_q, J, partials, nblk, row_len);
}
/* df_dIt, df_dI0 (optionally stored) and the two Jacobian products; partial rows of 16 (sum them with launch_finish_rows) */
void launch_mi_grad_gemv(const BatchView &bv, int nb, double norm_mult, const double *It, const double *I0, const double *tb,
	const double *Jt, const double *J0, const MiJ0Rebuild &rb, double *df_dIt, double *df_dI0, double *partials, int nblk, hipStream_t st) {
	MTFHIP_LAUNCH(k_mi_grad_gemv, grid2(nblk, bv.B), dim3(kBlock), 0, st, bv.N, bv.S, nb, norm_mult, It, I0, tb, Jt, J0, rb, df_dIt, df_dI0,
		partials, nblk);
}
void launch_mi_hess_finish(const BatchView &bv, int nb, const double *partials, int nblk, int row_len, const double *tb,
	int joint_off, int hist_off, int transpose_q, double *out, hipStream_t st) {
	size_t lds = sizeof(double) * ((size_t)nb * nb * bv.S + 36);
	MTFHIP_LAUNCH(k_mi_hess_finish, dim3(bv.B), dim3(kBlock), lds, st, bv.S, nb, partials, nblk, row_len, tb, joint_off,
		hist_off, transpose_q, out);
}

} // namespace mtfhip
