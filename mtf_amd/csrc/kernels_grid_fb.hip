/*
 * kernels_grid_fb.hip -- GridTracker's frame with forward-backward error estimation in ONE launch (r06): the shipped configuration
 * (Config/modules.cfg:80-82: grid_reset_at_each_frame 1, grid_fb_err_thresh 2, grid_fb_reinit 1) runs, per patch and frame,
 *     tracker->update()                          on the current frame                 GridTracker.cc:254-261
 *     tracker->initialize(tracker_location)      on the current frame                 :296-299 (backwardEstimation, fb_reinit)
 *     tracker->setImage(prev_img); update()      on the previous frame                :300-301
 *     getCentroid(fb_prev_pts[id], getRegion())                                        :302
 * Launch by launch that is k_iclk_track + host wait + k_template_init + k_iclk_track + host wait (258 us per frame of 256 patches, of
 * which the three kernels are ~80).  The three steps of a patch depend on nothing but the patch: one workgroup runs them back to back --
 * the forward loop of k_iclk_track (tolerance mode, template in registers, texel window in LDS), the template initialisation of
 * k_template_init at the corners the loop arrived at with NOTHING written per pixel (the backward template lives in the registers the
 * forward one has just vacated: the reset that follows the frame -- resetTrackers(reinit), :273-274 -- replaces it anyway), the
 * backward loop on the previous frame's texel window -- and hands the host the forward results (k_iclk_track's record) plus the corners
 * the backward pass arrived at.  Expressions, summation orders and reductions are those of the two kernels: the same bits as the three
 * launches (tests/test_gpu_grid.py::test_grid_fb_one_launch_equals_three).
 * The patch trackers are left as the FORWARD pass left them (warp, state, corners; template untouched).
 * One of the translation units of libmtfhip.so.
 */
#include "mtfhip_device.h"
#include "mtfhip_grid_device.h"

namespace mtfhip {

template <int AM, int PPT>
__global__ __launch_bounds__(kBlock) void k_grid_fb(BatchView bv, ImgView im, ImgView imp, mtfhip_sm_desc sm, TrackState ts,
	const double *h0inv_all, const double *ncc_sc_all, double norm_mult, double norm_add, double grad_eps, HostPublish pub, GridFbOut fo, RegionIngest rg) {
	static_assert(PPT <= 4, "every per-pixel operand of both loops stays in registers");
	constexpr bool NCC = AM == MTFHIP_AM_NCC;
	constexpr int K = NCC ? 11 : 9;
	constexpr int kWinW = 64, kWinH = 64;
	__shared__ double redk[2 * 4 * 16];
	__shared__ double sW[9], sSt[8], sH8[64], sIc[12], sCr[8], sFwd[28];
	__shared__ float win[kWinW * kWinH];
	__shared__ double red[4 * 16], sSum[56], sRed56[4 * 56], sH[64], sA[8 * 17], sHinv[64];
	const double *sG = sSum, *sSJ = sSum + 36, *sIJ = sSum + 44;
	const int t = blockIdx.x, N = bv.N, S = bv.S, tid = threadIdx.x;
	const bool hom = bv.ssm == MTFHIP_SSM_HOMOGRAPHY;

	/* one ICLK update() of the patch (k_iclk_track's tolerance-mode loop, NT/ICLK.cc:161-290) on `image`, from (W, St, Cr) */
	auto run_loop = [&](const ImgView &image, const double2 (&hpv)[PPT], const double (&zv)[PPT], const double (&i0v)[PPT], const double (&j0v)[PPT][8],
		const double (&tc)[16], double m0, double cn, double (&W)[9], double (&St)[8], double (&Cr)[8], const double (&Ic)[12], const double (&hrow)[8],
		int &n_it, double &f_last) {
		/* the texels of the whole loop from an LDS window around the corners (k_iclk_track) */
		int wx0 = 0, wy0 = 0;
		bool win_ok;
		{
			const double mnx = fmin(fmin(Cr[0], Cr[2]), fmin(Cr[4], Cr[6])), mxx = fmax(fmax(Cr[0], Cr[2]), fmax(Cr[4], Cr[6]));
			const double mny = fmin(fmin(Cr[1], Cr[3]), fmin(Cr[5], Cr[7])), mxy = fmax(fmax(Cr[1], Cr[3]), fmax(Cr[5], Cr[7]));
			const double cxm = 0.5 * (mnx + mxx), cym = 0.5 * (mny + mxy);
			win_ok = (mxx - mnx < kWinW - 6) & (mxy - mny < kWinH - 6) & (cxm > -1e6) & (cxm < 1e6) & (cym > -1e6) & (cym < 1e6) &
				(image.w >= kWinW) & (image.h >= kWinH);
			if (win_ok) {
				wx0 = min(max((int)floor(cxm) - kWinW / 2, 0), image.w - kWinW);
				wy0 = min(max((int)floor(cym) - kWinH / 2, 0), image.h - kWinH);
				float win_tv[kWinW * kWinH / kBlock];
#pragma unroll
				for (int j = 0; j < kWinW * kWinH / kBlock; ++j) {
					const int idx = tid + j * kBlock;
					win_tv[j] = image.data[(unsigned)((wy0 + idx / kWinW) * image.stride + wx0 + idx % kWinW)];
				}
#pragma unroll
				for (int j = 0; j < kWinW * kWinH / kBlock; ++j) win[tid + j * kBlock] = win_tv[j];
			}
		}
		__syncthreads();   /* (win_ok is uniform: every thread holds the same corners) */
		const double winx0 = (double)wx0, winx1 = (double)(wx0 + kWinW - 1), winy0 = (double)wy0, winy1 = (double)(wy0 + kWinH - 1);
		const double nN = (double)N, inv_n = 1.0 / nN;
		const double inv_cn = 1.0 / cn;
		for (int it = 0; it < sm.max_iters; ++it) {
			double m[12];
#pragma unroll
			for (int q = 0; q < 12; ++q) m[q] = 0.0;
#pragma unroll
			for (int k = 0; k < PPT; ++k) {
				const int i = tid + k * kBlock;
				const double2 hp = hpv[k];
				const double z = zv[k];
				double wx = fma(W[0], hp.x, fma(W[1], hp.y, W[2] * z)), wy = fma(W[3], hp.x, fma(W[4], hp.y, W[5] * z));
				if (hom) { const double inv = rcp_fast(fma(W[6], hp.x, fma(W[7], hp.y, W[8] * z))); wx *= inv; wy *= inv; }
				double pv;
				const bool inw = win_ok & (wx >= winx0) & (wx < winx1) & (wy >= winy0) & (wy < winy1);
				if (__builtin_amdgcn_ballot_w64(!inw) == 0) {
					const int lx = (int)wx, ly = (int)wy;
					const float *wp = win + ((ly - wy0) * kWinW + (lx - wx0));
					pv = bilin_val_fast(wp[0], wp[1], wp[kWinW], wp[kWinW + 1], wx - (double)lx, wy - (double)ly);
				} else pv = pix_val_fast(image, wx, wy);
				const double v = i < N ? fma(norm_mult, pv, norm_add) : 0.0;
				const double i0 = i0v[k];
				if constexpr (NCC) {
					m[0] += v; m[1] = fma(v, v, m[1]); m[2] = fma(i0, v, m[2]);
				} else {
					const double r = i < N ? v - i0 : 0.0;
					m[0] = fma(r, r, m[0]);
				}
				const double wgt = NCC ? v : (i < N ? v - i0 : 0.0);
#pragma unroll
				for (int s = 0; s < 8; ++s)
					if (s < S) m[K - 8 + s] = fma(wgt, j0v[k][s], m[K - 8 + s]);
			}
			block_allsum_h12(m, redk + ((it + 1) & 1) * 64);
			double g[8];
			if constexpr (NCC) {
				const double mt = m[0] * inv_n, b2 = fma(-nN * mt, mt, m[1]);
				const double inv_b = rsq_fast(b2), b = b2 * inv_b;
				const double inv_bc = inv_b * inv_cn, inv_b2 = inv_b * inv_b;
				const double f = fma(-nN * m0, mt, m[2]) * inv_bc;
				f_last = f;
				const double b_c = b * inv_cn;
#pragma unroll
				for (int s = 0; s < 8; ++s) {
					const double ut = fma(-mt, tc[s], m[3 + s]) * inv_b2, u0 = fma(-m0, tc[s], tc[8 + s]) * inv_bc;
					g[s] = b_c * fma(-f, u0, ut);
				}
			} else {
				f_last = -m[0] / 2;
#pragma unroll
				for (int s = 0; s < 8; ++s) g[s] = m[1 + s];
			}
			double dp[8];
			{
				const double *h = hrow;
				const double mine = -(((h[0] * g[0] + h[1] * g[1]) + (h[2] * g[2] + h[3] * g[3])) + ((h[4] * g[4] + h[5] * g[5]) + (h[6] * g[6] + h[7] * g[7])));
#pragma unroll
				for (int r = 0; r < 8; ++r) dp[r] = readlane_f64(mine, r);
			}
			double Wn[9];
			if (hom) {
				const double U0 = 1 + dp[0], U1 = dp[1], U2 = dp[2], U3 = dp[3], U4 = 1 + dp[4], U5 = dp[5], U6 = dp[6], U7 = dp[7];
				const double c0 = U4 - U5 * U7, c1 = U2 * U7 - U1, c2 = U1 * U5 - U2 * U4;
				const double c3 = U5 * U6 - U3, c4 = U0 - U2 * U6, c5 = U2 * U3 - U0 * U5;
				const double c6 = U3 * U7 - U4 * U6, c7 = U1 * U6 - U0 * U7, c8 = U0 * U4 - U1 * U3;
				const double ic8 = rcp_fast(c8);
				const double V[9] = {c0 * ic8, c1 * ic8, c2 * ic8, c3 * ic8, c4 * ic8, c5 * ic8, c6 * ic8, c7 * ic8, 1.0};
#pragma unroll
				for (int r = 0; r < 3; ++r)
#pragma unroll
					for (int c = 0; c < 3; ++c) Wn[3 * r + c] = fma(W[3 * r], V[c], fma(W[3 * r + 1], V[3 + c], W[3 * r + 2] * V[6 + c]));
				const double inv_w22 = rcp_fast(Wn[8]);
#pragma unroll
				for (int q = 0; q < 8; ++q) Wn[q] *= inv_w22;
				Wn[8] = 1;
				St[0] = Wn[0] - 1; St[1] = Wn[1]; St[2] = Wn[2]; St[3] = Wn[3]; St[4] = Wn[4] - 1; St[5] = Wn[5]; St[6] = Wn[6]; St[7] = Wn[7];
			} else {
				const double a = 1 + dp[2], b = dp[3], tx = dp[0], c = dp[4], d = 1 + dp[5], ty = dp[1];
				const double idet = rcp_fast(a * d - b * c);
				const double ia = d * idet, ib = -b * idet, ic = -c * idet, id = a * idet;
				const double itx = -(ia * tx + ib * ty), ity = -(ic * tx + id * ty);
				Wn[0] = W[0] * ia + W[1] * ic; Wn[1] = W[0] * ib + W[1] * id; Wn[2] = (W[0] * itx + W[1] * ity) + W[2];
				Wn[3] = W[3] * ia + W[4] * ic; Wn[4] = W[3] * ib + W[4] * id; Wn[5] = (W[3] * itx + W[4] * ity) + W[5];
				Wn[6] = 0; Wn[7] = 0; Wn[8] = 1;
				St[0] = Wn[2]; St[1] = Wn[5]; St[2] = Wn[0] - 1; St[3] = Wn[1]; St[4] = Wn[3]; St[5] = Wn[4] - 1; St[6] = 0; St[7] = 0;
			}
			double ch[4];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const double X = Ic[3 * q], Y = Ic[3 * q + 1], Z = Ic[3 * q + 2];
				double nx = (Wn[0] * X + Wn[1] * Y) + Wn[2] * Z, ny = (Wn[3] * X + Wn[4] * Y) + Wn[5] * Z;
				if (hom) { const double idn = rcp_fast((Wn[6] * X + Wn[7] * Y) + Wn[8] * Z); nx *= idn; ny *= idn; }
				const double ddx = Cr[2 * q] - nx, ddy = Cr[2 * q + 1] - ny;
				ch[q] = ddx * ddx + ddy * ddy;
				Cr[2 * q] = nx; Cr[2 * q + 1] = ny;
			}
			const double change = (ch[0] + ch[1]) + (ch[2] + ch[3]);
#pragma unroll
			for (int q = 0; q < 9; ++q) W[q] = Wn[q];
			++n_it;
			if (change < sm.epsilon) break;   /* uniform */
		}
	};

	/* ---- phase A: tracker->update() on the current frame (k_iclk_track, plain mode) ---- */
	double Cf[8];   /* where the forward pass arrives: the backward pass's template region */
	double Cb[8];   /* where the backward pass arrives */
	int n_it_b = 0;
	{
		const double2 *ip = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_PTS]) + (size_t)t * N;
		const double2 *ih = reinterpret_cast<const double2 *>(bv.buf[MTFHIP_BUF_INIT_HXY]) + (size_t)t * N;
		const double *iz = bv.buf[MTFHIP_BUF_INIT_Z] + (size_t)t * N;
		const double *I0 = bv.buf[MTFHIP_BUF_I0] + (size_t)t * N;
		const double *J0 = bv.buf[MTFHIP_BUF_J0] + (size_t)t * N * S;
		const double m0 = NCC ? ncc_sc_all[t * 8 + 0] : 0.0, cn = NCC ? ncc_sc_all[t * 8 + 1] : 1.0;
		double2 hpv[PPT];
		double zv[PPT], i0v[PPT], j0v[PPT][8];
#pragma unroll
		for (int k = 0; k < PPT; ++k) {
			const int i = tid + k * kBlock;
			const int ic = i < N ? i : N - 1;
			hpv[k] = bv.unit_z ? ip[ic] : ih[ic];
			zv[k] = bv.unit_z ? 1.0 : iz[ic];
			i0v[k] = i < N ? I0[ic] : 0.0;
#pragma unroll
			for (int s = 0; s < 8; ++s) j0v[k][s] = (s < S && i < N) ? J0[(size_t)s * N + ic] : 0.0;
		}
		double tc[16];
#pragma unroll
		for (int q = 0; q < 16; ++q) tc[q] = 0.0;
		if constexpr (NCC) {
#pragma unroll
			for (int q = 0; q < 16; ++q) tc[q] = ts.ncc_tm[(size_t)t * 52 + q];
		}
		if (tid < 64) { const int r = tid >> 3, c = tid & 7; sH8[tid] = (r < S && c < S) ? h0inv_all[(size_t)t * 64 + c * S + r] : 0.0; }
		if (tid < 12) sIc[tid] = ts.init_corners_hm[12 * t + tid];
		if (ts.fresh_reset) {   /* (uniform) behind a fused re-initialisation: identity warp, zero state, the template's own corners (k_iclk_track) */
			if (tid < 8) sCr[tid] = ts.init_corners_hm[12 * t + 3 * (tid >> 1) + (tid & 1)];
			if (tid < 9) sW[tid] = (tid == 0 || tid == 4 || tid == 8) ? 1.0 : 0.0;
			if (tid < 8) sSt[tid] = 0.0;
		} else {
			if (tid < 8) sCr[tid] = ts.corners[8 * t + tid];
			if (tid < 9) sW[tid] = bv.warps[9 * t + tid];
			if (tid < 8) sSt[tid] = bv.states[8 * t + tid];
		}
		__syncthreads();
		double W[9], St[8], Ic[12], hrow[8];
#pragma unroll
		for (int q = 0; q < 9; ++q) W[q] = sW[q];
#pragma unroll
		for (int q = 0; q < 8; ++q) { St[q] = sSt[q]; Cf[q] = sCr[q]; hrow[q] = sH8[8 * (tid & 7) + q]; }
#pragma unroll
		for (int q = 0; q < 12; ++q) Ic[q] = sIc[q];
		int n_it = 0;
		double f_last = 0;
		run_loop(im, hpv, zv, i0v, j0v, tc, m0, cn, W, St, Cf, Ic, hrow, n_it, f_last);
		/* every thread holds the same W / St / Cf: the record goes to LDS until the tail (the registers are the backward pass's now) */
		if (tid < 28) {
			double v = 0;
#pragma unroll
			for (int q = 0; q < 9; ++q) v = tid == q ? W[q] : v;
#pragma unroll
			for (int q = 0; q < 8; ++q) { v = tid == 9 + q ? St[q] : v; v = tid == 17 + q ? Cf[q] : v; }
			v = tid == 25 ? f_last : v;
			v = tid == 26 ? (double)n_it : v;
			sFwd[tid] = v;
		}
		if (!fo.reinit) {
			/* fb_reinit = 0 (:297-299 skipped): setImage(prev_img); update() with the template and from the state the forward pass left */
#pragma unroll
			for (int q = 0; q < 8; ++q) Cb[q] = Cf[q];
			double f_b = 0;
			run_loop(imp, hpv, zv, i0v, j0v, tc, m0, cn, W, St, Cb, Ic, hrow, n_it_b, f_b);
		}
	}
	if (fo.reinit) {   /* (uniform) */

	/* ---- phase B: tracker->initialize(tracker_location) on the current frame (k_template_init, region mode; nothing written per pixel) ---- */
	double W0[9];
	const bool bad = !rect_to_quad_hd(rg.lo_x, rg.lo_y, rg.hi_x, rg.hi_y, Cf, W0);
	if (bad) {
#pragma unroll
		for (int q = 0; q < 9; ++q) W0[q] = (q == 0 || q == 4 || q == 8) ? 1.0 : 0.0;
	}
	if (hom && fabs(W0[6]) < 1e-15 && fabs(W0[7]) < 1e-15) { W0[6] = 0; W0[7] = 0; }
	double2 hpb[PPT];
	double zb[PPT], i0b[PPT], jb[PPT][8];
	{
		const double gmult = norm_mult / (2 * grad_eps);
		double2 pk[PPT]; double zk[PPT];
		Cell ck[PPT];
#pragma unroll
		for (int k = 0; k < PPT; ++k) {
			const int i = tid + k * kBlock;
			const int ic = i < N ? i : N - 1;
			const int col = ic % rg.resx, row = ic / rg.resx;
			const double nx = (rg.resx == 1 || col == rg.resx - 1) ? rg.hi_x : rg.lo_x + col * ((rg.hi_x - rg.lo_x) / (rg.resx - 1));
			const double ny = (rg.resy == 1 || row == rg.resy - 1) ? rg.hi_y : rg.lo_y + row * ((rg.hi_y - rg.lo_y) / (rg.resy - 1));
			const double X = W0[0] * nx + W0[1] * ny + W0[2] * 1.0;
			const double Y = W0[3] * nx + W0[4] * ny + W0[5] * 1.0;
			const double Z = W0[6] * nx + W0[7] * ny + W0[8] * 1.0;
			const double2 p = (W0[6] == 0 && W0[7] == 0 && W0[8] == 1.0) ? make_double2(X, Y) : make_double2(X / Z, Y / Z);
			const double zi = rg.force_unit_z ? 1.0 : Z;
			const double2 hxy = rg.force_unit_z ? p : make_double2(X, Y);
			pk[k] = p; zk[k] = zi;
			hpb[k] = rg.force_unit_z ? p : hxy; zb[k] = rg.force_unit_z ? 1.0 : zi;   /* (the loop's operands: INIT_PTS | INIT_HXY, INIT_Z as the launcher's unit_z picks them) */
			Cell c;
			const double w = (double)(unsigned int)im.w, h = (double)(unsigned int)im.h;
			const bool in0 = !((p.x < 0) || (p.x >= w) || (p.y < 0) || (p.y >= h));
			const int lx = in0 ? (int)p.x : 0, ly = in0 ? (int)p.y : 0;
			const double dx = p.x - lx, dy = p.y - ly;
			const int ux = dx == 0 ? lx : lx + 1, uy = dy == 0 ? ly : ly + 1;
			c.valid = in0 && !(ux >= im.w || uy >= im.h);
			c.lx = c.valid ? lx : -1; c.ly = c.valid ? ly : -1; c.ux = c.valid ? ux : -1; c.uy = c.valid ? uy : -1;
			const int slx = c.valid ? lx : 0, sly = c.valid ? ly : 0, sux = c.valid ? ux : 0, suy = c.valid ? uy : 0;
			const float *r0 = im.data + (size_t)sly * im.stride, *r1 = im.data + (size_t)suy * im.stride;
			c.t00 = r0[slx]; c.t01 = r0[sux]; c.t10 = r1[slx]; c.t11 = r1[sux];
			ck[k] = c;
		}
#pragma unroll
		for (int k = 0; k < PPT; ++k) {
			const int i = tid + k * kBlock;
			i0b[k] = 0.0;
#pragma unroll
			for (int s = 0; s < 8; ++s) jb[k][s] = 0.0;
			if (i < N) {
				const double2 p = pk[k]; const double zi = zk[k];
				const Cell &c = ck[k];
				const double v = norm_mult * pix_val_cell(im, c, p.x, p.y) + norm_add;
				double inc = pix_val_cell(im, c, p.x + grad_eps, p.y), dec = pix_val_cell(im, c, p.x - grad_eps, p.y);
				const double gx = (inc - dec) * gmult;
				inc = pix_val_cell(im, c, p.x, p.y + grad_eps); dec = pix_val_cell(im, c, p.x, p.y - grad_eps);
				const double gy = (inc - dec) * gmult;
				double r[8];
#pragma unroll
				for (int s = 0; s < 8; ++s) r[s] = 0.0;
				if (hom) {
					const double inv_det = 1.0 / (rg.force_unit_z ? 1.0 : zi);
					const double dwx_dx = (1.0 - 0.0 * p.x), dwx_dy = (0.0 - 0.0 * p.x), dwy_dx = (0.0 - 0.0 * p.y), dwy_dy = (1.0 - 0.0 * p.y);
					const double Ix = (dwx_dx * gx + dwy_dx * gy) * inv_det, Iy = (dwx_dy * gx + dwy_dy * gy) * inv_det;
					hom_row(r, Ix, Iy, p.x, p.y, p.x, p.y);
				} else {
					const double a = 0.0 + 1, b = 0.0, cc = 0.0, d = 0.0 + 1;
					const double Ixx = gx * p.x, Ixy = gx * p.y, Iyy = gy * p.y, Iyx = gy * p.x;
					r[0] = gx * a + gy * cc; r[1] = gx * b + gy * d;
					r[2] = Ixx * a + Iyx * cc; r[3] = Ixy * a + Iyy * cc; r[4] = Ixx * b + Iyx * d; r[5] = Ixy * b + Iyy * d;
				}
				i0b[k] = v;
#pragma unroll
				for (int s = 0; s < 8; ++s) jb[k][s] = r[s];
			}
		}
	}
	double m0b = 0.0, cnb = 1.0;
	if constexpr (NCC) {
		double s1[1] = {0.0};
#pragma unroll
		for (int k = 0; k < PPT; ++k) s1[0] += i0b[k];
		init_allsum<1>(s1, red);
		m0b = s1[0] / (double)N;
		double s2[1] = {0.0};
#pragma unroll
		for (int k = 0; k < PPT; ++k) { const int i = tid + k * kBlock; const double dv = i < N ? i0b[k] - m0b : 0.0; s2[0] = fma(dv, dv, s2[0]); }
		init_allsum<1>(s2, red);
		cnb = sqrt(s2[0]);
	}
	{
		double acc[56];
#pragma unroll
		for (int q = 0; q < 56; ++q) acc[q] = 0.0;
#pragma unroll
		for (int k = 0; k < PPT; ++k) {
			int idx = 0;
#pragma unroll
			for (int a = 0; a < 8; ++a)
#pragma unroll
				for (int c = a; c < 8; ++c) { acc[idx] = fma(jb[k][a], jb[k][c], acc[idx]); ++idx; }
#pragma unroll
			for (int s = 0; s < 8; ++s) { acc[36 + s] += jb[k][s]; acc[44 + s] = fma(i0b[k], jb[k][s], acc[44 + s]); }
		}
		__syncthreads();
		block_reduce_store<56>(acc, sSum, sRed56);
	}
	__syncthreads();
	if (tid < 64) {
		const int a = tid >> 3, c = tid & 7;
		if (a < S && c < S) {
			const int lo = a < c ? a : c, hi = a < c ? c : a;
			const double G = sG[lo * 8 - (lo * (lo - 1)) / 2 + (hi - lo)];
			double h;
			if constexpr (NCC) {
				const double inv_b2 = 1.0 / (cnb * cnb);
				const double ua = (sIJ[a] - m0b * sSJ[a]) * inv_b2, uc = (sIJ[c] - m0b * sSJ[c]) * inv_b2;
				h = -(G - sSJ[a] * sSJ[c] / (double)N) * inv_b2 + ua * uc;
			} else h = -G;
			sH[c * S + a] = h;
		}
	}
	__syncthreads();
	if (tid < 64) invert_definite_wave(S, sH, sA, sHinv);
	__syncthreads();

	/* ---- phase C: tracker->setImage(prev_img); update() from the identity warp at tracker_location ---- */
	{
		if (tid < 64) { const int r = tid >> 3, c = tid & 7; sH8[tid] = (r < S && c < S) ? sHinv[c * S + r] : 0.0; }
		__syncthreads();
		double tc[16];
#pragma unroll
		for (int q = 0; q < 8; ++q) { tc[q] = NCC ? sSJ[q] : 0.0; tc[8 + q] = NCC ? sIJ[q] : 0.0; }
		double W[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, St[8] = {0, 0, 0, 0, 0, 0, 0, 0}, Ic[12], hrow[8];
#pragma unroll
		for (int q = 0; q < 8; ++q) { Cb[q] = Cf[q]; hrow[q] = sH8[8 * (tid & 7) + q]; }
#pragma unroll
		for (int q = 0; q < 12; ++q) Ic[q] = (q % 3 == 2) ? 1.0 : Cf[2 * (q / 3) + q % 3];
		double f_b = 0;
		if (!bad) run_loop(imp, hpb, zb, i0b, jb, tc, m0b, cnb, W, St, Cb, Ic, hrow, n_it_b, f_b);
		else n_it_b = -1;
	}
	}   /* fo.reinit */

	/* ---- results: the backward corners, then k_iclk_track's record of the forward pass (device slab, host mirror, flag) ---- */
	if (tid < 64) {
		double cb = 0;
#pragma unroll
		for (int q = 0; q < 8; ++q) cb = tid == q ? Cb[q] : cb;
		if (tid == 8) cb = (double)n_it_b;
		if (tid < 9) {
			if (fo.host) __hip_atomic_store(fo.host + 9 * (size_t)t + tid, cb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			if (fo.dev) fo.dev[9 * (size_t)t + tid] = cb;
		}
		const double wq = tid < 9 ? sFwd[tid] : 0.0, sq = tid < 8 ? sFwd[9 + tid] : 0.0, cq = tid < 8 ? sFwd[17 + tid] : 0.0;
		const int n_it = (int)sFwd[26];
		if (pub.host) publish_target(pub, t, wq, sq, cq, n_it);
		if (tid < 9) bv.warps[9 * t + tid] = wq;
		if (tid < 8) { bv.states[8 * t + tid] = sq; ts.corners[8 * t + tid] = cq; }
		if (tid == 0) { ts.n_iters[t] = n_it; ts.acc[(size_t)t * ACC_COUNT + ACC_RR] = sFwd[25]; }
	}
}

template <int AM>
static bool launch_grid_fb_am(const BatchView &bv, const ImgView &im, const ImgView &imp, const mtfhip_sm_desc &sm, const TrackState &ts, const double *h0inv,
	const double *ncc_sc, double norm_mult, double norm_add, double grad_eps, const HostPublish &pub, const GridFbOut &fo, const RegionIngest &rg, hipStream_t st) {
	const int ppt = (bv.N + kBlock - 1) / kBlock;
#define MTFHIP_FB_CASE(P) MTFHIP_LAUNCH((k_grid_fb<AM, P>), dim3(bv.B), dim3(kBlock), 0, st, bv, im, imp, sm, ts, h0inv, ncc_sc, norm_mult, norm_add, grad_eps, pub, fo, rg)
	if (ppt <= 1) MTFHIP_FB_CASE(1);
	else if (ppt <= 2) MTFHIP_FB_CASE(2);
	else if (ppt <= 3) MTFHIP_FB_CASE(3);
	else if (ppt <= 4) MTFHIP_FB_CASE(4);
	else return false;
#undef MTFHIP_FB_CASE
	return true;
}
/* tolerance mode, SSD / NCC, N <= 4 * kBlock; rg carries the template lattice's geometry only (lo / hi, resx / resy, force_unit_z) */
bool launch_grid_fb(const BatchView &bv, const ImgView &im, const ImgView &imp, const mtfhip_sm_desc &sm, const TrackState &ts, const double *h0inv,
	const double *ncc_sc, double norm_mult, double norm_add, double grad_eps, const HostPublish &pub, const GridFbOut &fo, const RegionIngest &rg, hipStream_t st) {
	if (bv.am == MTFHIP_AM_NCC) return launch_grid_fb_am<MTFHIP_AM_NCC>(bv, im, imp, sm, ts, h0inv, ncc_sc, norm_mult, norm_add, grad_eps, pub, fo, rg, st);
	return launch_grid_fb_am<MTFHIP_AM_SSD>(bv, im, imp, sm, ts, h0inv, ncc_sc, norm_mult, norm_add, grad_eps, pub, fo, rg, st);
}

} // namespace mtfhip
