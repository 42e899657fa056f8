/*
 * kernels_fused.hip -- the fused Lucas-Kanade iteration (SSD and NCC) and the device-side solve + update
 * (one of the translation units of libmtfhip.so; conventions and the shared device helpers: mtfhip_device.h)
 */
#include "mtfhip_finish_device.h"
#include "mtfhip_fused_device.h"

namespace mtfhip {

template <int SSM, bool CHAINED, int MODE, bool MAT>
__global__ __launch_bounds__(kBlock, MTFHIP_FUSED_WAVES) void k_fused_ssd(BatchView bv, ImgView im, FusedArgs fa, double *partials, int nblk) {
	fused_lk_body<MTFHIP_AM_SSD, SSM, CHAINED, MODE, MAT>(bv, im, fa, partials, nblk);
}
template <int SSM, bool CHAINED, int MODE, bool MAT>
__global__ __launch_bounds__(kBlock, MTFHIP_FUSED_WAVES) void k_fused_ncc(BatchView bv, ImgView im, FusedArgs fa, double *partials, int nblk) {
	fused_lk_body<MTFHIP_AM_NCC, SSM, CHAINED, MODE, MAT>(bv, im, fa, partials, nblk);
}
/* tolerance-mode lean launches (see fused_lk_body) */
template <int AM, int SSM, int MODE, bool CHAINED>
__global__ __launch_bounds__(kBlock, MTFHIP_FAST_WAVES) void k_fused_fast(BatchView bv, ImgView im, FusedArgs fa, double *partials, int nblk) {
	fused_lk_body<AM, SSM, CHAINED, MODE, false, true>(bv, im, fa, partials, nblk);
}


/* stand-alone finish: one wave per target */
/* pc (two-queue loop): the queues run best half a period apart -- one's fill / drain / solve under the other's streaming (48-49 us per
 * step of 64 x 200 x 200, the two pixel passes starting 23-25 us apart) -- but started together, or on some boxes by themselves, they
 * stay close to lockstep (55-56 us).  Each queue's solve stamps the wall clock when it ends, and ends no sooner than `frac` of its own
 * last period after the other queue's stamp: a queue that runs too close behind the other is held back until it is not. */
__global__ __launch_bounds__(256) void k_finish_track(BatchView bv, mtfhip_sm_desc sm, TrackState ts,
	const double *partials, int nblk, PhaseCtl pc) {
	if (ts.fast_finish) finish_track_fast_body(bv, sm, ts, partials, nblk, blockIdx.x);
	else finish_track_body(bv, sm, ts, partials, nblk, blockIdx.x);
	if (pc.mine && blockIdx.x == 0 && threadIdx.x == 0) {
		const unsigned long long prev = ld_coh(pc.mine), other = ld_coh(pc.other);
		unsigned long long now = wall_clock64();   /* 100 MHz */
		if (prev && other && now > prev && now - prev < 50000ull) {   /* (a period of less than 500 us: the queue is in its stride) */
			const unsigned long long min_lag = (unsigned long long)((double)(now - prev) * pc.frac);
			while (now > other && now - other < min_lag) { __builtin_amdgcn_s_sleep(8); now = wall_clock64(); }
		}
		st_coh(pc.mine, now);
	}
}


/* MI device-side loop: g and H of the fused MI passes (mi_H = [B][64] H column-major | [B][16] unused here | [B][64] second H of
 * SumOfStd; gpart = the gradient pass's block rows [B][ng][16]) laid out as the reduced row the finish reads for SSD --
 * ACC_H = upper triangle of -H, ACC_G = the Jacobian sum the search method scales (ESM halves it, NT/ESM.cc:246-255) -- and
 * the finish itself, in one launch.  gmode: 0 ICLK, 1 FCLK, 2 ESM Original, 3 ESM DiffOfJacs. */
__global__ __launch_bounds__(64) void k_finish_track_mi(BatchView bv, mtfhip_sm_desc sm, TrackState ts, int sum_std, int gmode,
	const double *mi_H, const double *gpart, int ng, double *rows) {
	const int t = blockIdx.x, lane = threadIdx.x, S = bv.S, B = bv.B;
	const double *Hs = mi_H + 64 * (size_t)t, *H2 = mi_H + 80 * (size_t)B + 64 * (size_t)t;
	double *row = rows + (size_t)t * ACC_COUNT;
	__shared__ double gs[16];
	if (lane < 16) gs[lane] = column_sum(gpart + (size_t)t * ng * 16 + lane, ng, 16);
	if (lane < ACC_COUNT - 36) row[36 + lane] = 0.0;
	const int a = lane >> 3, c = lane & 7;
	if (a <= c) {
		double hv = 0.0;
		if (c < S) { hv = Hs[c * S + a]; if (sum_std) hv = 0.5 * (hv + H2[c * S + a]); }
		row[ACC_H + a * 8 - (a * (a - 1)) / 2 + (c - a)] = -hv;
	}
	__syncthreads();
	if (lane < S) {
		const double gt = gs[lane], g0 = gs[8 + lane];
		row[ACC_G + lane] = gmode == 0 ? g0 : (gmode == 1 ? gt : (gmode == 2 ? 2.0 * gt : gt - g0));
	}
	__syncthreads();   /* the row is read back by the same workgroup */
	finish_track_body(bv, sm, ts, rows, 1, t);
}

/* ===================================================================== */
/* launchers                                                              */
/* ===================================================================== */

template <int SSM, bool CHAINED, int MODE>
static void launch_fused_mat(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk,
	hipStream_t st) {
	dim3 g = grid2(nblk, bv.B);
	if (bv.am == MTFHIP_AM_NCC) {
		if (fa.materialize)
			MTFHIP_LAUNCH((k_fused_ncc<SSM, CHAINED, MODE, true>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
		else
			MTFHIP_LAUNCH((k_fused_ncc<SSM, CHAINED, MODE, false>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
		return;
	}
	if (fa.materialize)
		MTFHIP_LAUNCH((k_fused_ssd<SSM, CHAINED, MODE, true>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
	else
		MTFHIP_LAUNCH((k_fused_ssd<SSM, CHAINED, MODE, false>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
}
template <int SSM, bool CHAINED>
static void launch_fused_mode(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk,
	hipStream_t st) {
	if (fa.mode == 0) launch_fused_mat<SSM, CHAINED, 0>(bv, im, fa, partials, nblk, st);
	else if (fa.mode == 1) launch_fused_mat<SSM, CHAINED, 1>(bv, im, fa, partials, nblk, st);
	else launch_fused_mat<SSM, CHAINED, 2>(bv, im, fa, partials, nblk, st);
}
template <int AM, int SSM>
static void launch_fused_fast(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk, hipStream_t st) {
	dim3 g = grid2(nblk, bv.B);
	/* (ICLK takes no gradient: one instantiation) */
	if (fa.mode == 2) MTFHIP_LAUNCH((k_fused_fast<AM, SSM, 2, true>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
	else if (fa.mode == 0 && fa.chained) MTFHIP_LAUNCH((k_fused_fast<AM, SSM, 0, true>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
	else if (fa.mode == 0) MTFHIP_LAUNCH((k_fused_fast<AM, SSM, 0, false>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
	else if (fa.chained) MTFHIP_LAUNCH((k_fused_fast<AM, SSM, 1, true>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
	else MTFHIP_LAUNCH((k_fused_fast<AM, SSM, 1, false>), g, dim3(kBlock), 0, st, bv, im, fa, partials, nblk);
}
void launch_fused_ssd(const BatchView &bv, const ImgView &im, const FusedArgs &fa, double *partials, int nblk,
	hipStream_t st) {
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY;
	if (bv.C > 1) { launch_fused_mc(bv, im, fa, partials, nblk, st); return; }   /* MCSSD / MCNCC */
	if (fa.fast_math && !fa.materialize) {
		const bool ncc = bv.am == MTFHIP_AM_NCC;
		if (hom && ncc) launch_fused_fast<MTFHIP_AM_NCC, MTFHIP_SSM_HOMOGRAPHY>(bv, im, fa, partials, nblk, st);
		else if (hom) launch_fused_fast<MTFHIP_AM_SSD, MTFHIP_SSM_HOMOGRAPHY>(bv, im, fa, partials, nblk, st);
		else if (ncc) launch_fused_fast<MTFHIP_AM_NCC, MTFHIP_SSM_AFFINE>(bv, im, fa, partials, nblk, st);
		else launch_fused_fast<MTFHIP_AM_SSD, MTFHIP_SSM_AFFINE>(bv, im, fa, partials, nblk, st);
		return;
	}
	if (hom && fa.chained) launch_fused_mode<MTFHIP_SSM_HOMOGRAPHY, true>(bv, im, fa, partials, nblk, st);
	else if (hom) launch_fused_mode<MTFHIP_SSM_HOMOGRAPHY, false>(bv, im, fa, partials, nblk, st);
	else if (fa.chained) launch_fused_mode<MTFHIP_SSM_AFFINE, true>(bv, im, fa, partials, nblk, st);
	else launch_fused_mode<MTFHIP_SSM_AFFINE, false>(bv, im, fa, partials, nblk, st);
}
void launch_finish_track(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts, const double *partials,
	int nblk, hipStream_t st, PhaseCtl pc) {
	/* NCC rows are 72 wide: two waves load them, the first one solves; many block rows (a single large target): 240 lanes sum them */
	MTFHIP_LAUNCH(k_finish_track, dim3(bv.B), dim3(nblk > 8 ? 256 : (bv.am == MTFHIP_AM_NCC ? 128 : 64)), 0, st, bv, sm, ts, partials, nblk, pc);
}

void launch_finish_track_mi(const BatchView &bv, const mtfhip_sm_desc &sm, const TrackState &ts, int sum_std, int gmode,
	const double *mi_H, const double *gpart, int ng, double *rows, hipStream_t st) {
	MTFHIP_LAUNCH(k_finish_track_mi, dim3(bv.B), dim3(64), 0, st, bv, sm, ts, sum_std, gmode, mi_H, gpart, ng, rows);
}

/* The single-target launches read the warp and the state out of the kernel-argument segment at an offset computed from the C++
 * layout of (BatchView, ImgView, FusedArgs) -- fused_lk_body's static_asserts check the structs, not what the runtime actually
 * puts into the segment.  This probe has the fused kernels' leading parameters, reads the seventeen doubles with the same
 * arithmetic and hands them back: the library asks once per process and keeps the warp upload in front of every launch if the
 * answer is not what it passed in. */
__global__ void k_kernarg_probe(BatchView bv, ImgView im, FusedArgs fa, double *out) {
	const char *kernarg = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
	const double *kw = reinterpret_cast<const double *>(kernarg + sizeof(BatchView) + sizeof(ImgView) + offsetof(FusedArgs, iw));
	if (threadIdx.x < 17) out[threadIdx.x] = kw[threadIdx.x];
	if (threadIdx.x == 17) out[17] = (double)(bv.B + im.w + fa.mode);   /* (the arguments are live) */
}
bool kernarg_layout_verified(hipStream_t st) {
	static int state = -1;   /* -1 not asked, 0 no, 1 yes */
	if (state >= 0) return state == 1;
	state = 0;
	double *d_out = nullptr;
	if (hipMalloc(&d_out, sizeof(double) * 18) != hipSuccess) { (void)hipGetLastError(); return false; }
	BatchView bv{}; ImgView im{}; FusedArgs fa{};
	bv.B = 1; im.w = 2; fa.mode = 3; fa.inline_warp = 1;
	for (int q = 0; q < 9; ++q) fa.iw[q] = 0.5 + 1.25 * q;
	for (int q = 0; q < 8; ++q) fa.is[q] = -3.0 - 0.75 * q;
	double h[18] = {0};
	hipLaunchKernelGGL(k_kernarg_probe, dim3(1), dim3(64), 0, st, bv, im, fa, d_out);
	bool ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, st) == hipSuccess &&
		hipStreamSynchronize(st) == hipSuccess;
	(void)hipFree(d_out);
	for (int q = 0; ok && q < 9; ++q) ok = h[q] == fa.iw[q];
	for (int q = 0; ok && q < 8; ++q) ok = h[9 + q] == fa.is[q];
	state = ok ? 1 : 0;
	return ok;
}

#ifdef MTFHIP_FIN_TRACE
void debug_fin_trace(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fin_trace), sizeof(unsigned long long) * 16); }
#endif
} // namespace mtfhip
