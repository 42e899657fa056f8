/*
 * kernels_preproc.hip -- pre-processing and pyramid levels (the step upstream of the path)
 * (one of the translation units of libmtfhip.so; conventions and the shared device helpers: mtfhip_device.h)
 */
#include "mtfhip_device.h"
#include <algorithm>

namespace mtfhip {

/* ===================================================================== */
/* pre-processing and pyramid levels (the step upstream of the path)      */
/* ===================================================================== */
/* What the reference gets from OpenCV (Utilities/src/preprocUtils.cc:108-127 for the default CV_32FC1 output):
 *   frame_raw.convertTo(float) -> cvtColor(BGR2GRAY) when the input has 3 channels -> GaussianBlur(5x5, sigma 3)
 * and for PyramidalTracker (SM/src/PyramidalTracker.cc:88-97) cv::pyrDown (scale 0.5) or cv::resize + GaussianBlur.
 * OpenCV is a third-party dependency that is absent here; these kernels follow its published float32 algorithms
 * (operation order of the symmetric separable filter engine, BORDER_REFLECT_101, INTER_LINEAR with pixel-centre
 * alignment), restated in NumPy in oracle/preproc_ref.py.  All arithmetic is float32, no contraction. */
__device__ __forceinline__ int reflect101(int p, int n) {
	if (n == 1) return 0;
	while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
	return p;
}
/* convertTo(CV_32F) + cvtColor(BGR2GRAY): gray = B*0.114f + G*0.587f + R*0.299f (float, left to right) */
__global__ __launch_bounds__(kBlock) void k_to_gray_f32(const unsigned char *raw, int rows, int cols, size_t stride_bytes, int channels,
	int depth_f32, float *out) {
	const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
	if (x >= cols) return;
	const unsigned char *row = raw + (size_t)y * stride_bytes;
	float v;
	if (channels == 1) {
		v = depth_f32 ? reinterpret_cast<const float *>(row)[x] : (float)row[x];
	} else {
		float b, g, r;
		if (depth_f32) { const float *p = reinterpret_cast<const float *>(row) + 3 * x; b = p[0]; g = p[1]; r = p[2]; }
		else { const unsigned char *p = row + 3 * x; b = (float)p[0]; g = (float)p[1]; r = (float)p[2]; }
		v = b * 0.114f + g * 0.587f + r * 0.299f;
	}
	out[(size_t)y * cols + x] = v;
}
/* symmetric 5-tap row pass: S[0]*k0 + (S[-1]+S[1])*k1 + (S[-2]+S[2])*k2 (SymmRowSmallFilter, ksize 5) */
__global__ __launch_bounds__(kBlock) void k_sym5_rows(const float *src, int rows, int cols, float k0, float k1, float k2, float *dst) {
	const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
	if (x >= cols) return;
	const float *S = src + (size_t)y * cols;
	const float s0 = S[x], m1 = S[reflect101(x - 1, cols)], p1 = S[reflect101(x + 1, cols)];
	const float m2 = S[reflect101(x - 2, cols)], p2 = S[reflect101(x + 2, cols)];
	dst[(size_t)y * cols + x] = s0 * k0 + (m1 + p1) * k1 + (m2 + p2) * k2;
}
/* symmetric 5-tap column pass: s = k0*S0 + 0; s += k1*(S+1 + S-1); s += k2*(S+2 + S-2) (SymmColumnFilter) */
__global__ __launch_bounds__(kBlock) void k_sym5_cols(const float *src, int rows, int cols, float k0, float k1, float k2, float *dst) {
	const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
	if (x >= cols) return;
	const float c0 = src[(size_t)y * cols + x];
	const float m1 = src[(size_t)reflect101(y - 1, rows) * cols + x], p1 = src[(size_t)reflect101(y + 1, rows) * cols + x];
	const float m2 = src[(size_t)reflect101(y - 2, rows) * cols + x], p2 = src[(size_t)reflect101(y + 2, rows) * cols + x];
	float s = k0 * c0 + 0.0f;
	s += k1 * (p1 + m1);
	s += k2 * (p2 + m2);
	dst[(size_t)y * cols + x] = s;
}
/* cv::pyrDown, float: rows  r[x] = S[2x]*6 + (S[2x-1]+S[2x+1])*4 + S[2x-2] + S[2x+2] ; columns the same on the five row buffers,
 * times 1/256 */
__global__ __launch_bounds__(kBlock) void k_pyr_down(const float *src, int srows, int scols, int drows, int dcols, float *dst) {
	const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
	if (x >= dcols) return;
	float r[5];
#pragma unroll
	for (int j = 0; j < 5; ++j) {
		const float *S = src + (size_t)reflect101(2 * y - 2 + j, srows) * scols;
		const float c = S[reflect101(2 * x, scols)], m1 = S[reflect101(2 * x - 1, scols)], p1 = S[reflect101(2 * x + 1, scols)];
		const float m2 = S[reflect101(2 * x - 2, scols)], p2 = S[reflect101(2 * x + 2, scols)];
		r[j] = c * 6.0f + (m1 + p1) * 4.0f + m2 + p2;
	}
	dst[(size_t)y * dcols + x] = (r[2] * 6.0f + (r[1] + r[3]) * 4.0f + r[0] + r[4]) * (1.0f / 256.0f);
}
/* cv::resize INTER_LINEAR, float: fx = (float)((dx + 0.5) * scale - 0.5), clamped like resizeGeneric's index tables */
__device__ __forceinline__ void lin_coord(int d, double scale, int n, int &s, float &f) {
	f = (float)(((double)d + 0.5) * scale - 0.5);
	s = (int)floorf(f);
	f -= (float)s;
	if (s < 0) { f = 0.0f; s = 0; }
	if (s >= n - 1) { f = 0.0f; s = n - 1; }
}
__global__ __launch_bounds__(kBlock) void k_resize_linear(const float *src, int srows, int scols, int drows, int dcols, float *dst) {
	const int x = blockIdx.x * kBlock + threadIdx.x, y = blockIdx.y;
	if (x >= dcols) return;
	int sx, sy; float fx, fy;
	lin_coord(x, (double)scols / dcols, scols, sx, fx);
	lin_coord(y, (double)srows / drows, srows, sy, fy);
	const int sx1 = sx + 1 < scols ? sx + 1 : sx, sy1 = sy + 1 < srows ? sy + 1 : sy;
	const float *S0 = src + (size_t)sy * scols, *S1 = src + (size_t)sy1 * scols;
	const float a0 = 1.0f - fx, a1 = fx, b0 = 1.0f - fy, b1 = fy;
	const float h0 = sx >= scols - 1 ? S0[sx] * 1.0f : S0[sx] * a0 + S0[sx1] * a1;
	const float h1 = sx >= scols - 1 ? S1[sx] * 1.0f : S1[sx] * a0 + S1[sx1] * a1;
	dst[(size_t)y * dcols + x] = h0 * b0 + h1 * b1;
}

/* ---- hist_eq (Utilities/src/preprocUtils.cc:120-125): frame_gs.convertTo(CV_8UC1) -> cv::equalizeHist -> convertTo(CV_32FC1).
 * convertTo rounds to nearest even and saturates (cvRound + saturate_cast<uchar>); equalizeHist (imgproc/histogram.cpp): the 256-bin
 * histogram, i = first occupied bin, a constant image is left alone, otherwise scale = 255.f / (total - hist[i]) in float and
 * lut[j] = saturate_cast<uchar>(sum_{i < k <= j} hist[k] * scale), lut[i] = 0.  All integer / float32 work: bit-exact. ---- */
__device__ __forceinline__ int to_u8(float v) { const int r = __float2int_rn(v); return r < 0 ? 0 : (r > 255 ? 255 : r); }
__global__ __launch_bounds__(kBlock) void k_hist_u8(const float *gray, size_t n, unsigned *hist) {
	__shared__ unsigned h[256];
	h[threadIdx.x] = 0;
	__syncthreads();
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) atomicAdd(&h[to_u8(gray[i])], 1u);
	__syncthreads();
	if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_hist_lut(const unsigned *hist, unsigned total, float *lut) {
	if (threadIdx.x || blockIdx.x) return;
	int i = 0;
	while (!hist[i]) ++i;
	if (hist[i] == total) { for (int j = 0; j < 256; ++j) lut[j] = (float)j; return; }   /* dst.setTo(i): the only value that occurs maps to itself */
	const float scale = 255.f / (float)(total - hist[i]);
	int sum = 0;
	for (int j = 0; j <= i; ++j) lut[j] = 0.f;
	for (int j = i + 1; j < 256; ++j) { sum += (int)hist[j]; lut[j] = (float)to_u8((float)sum * scale); }
}
__global__ __launch_bounds__(kBlock) void k_apply_lut(float *gray, size_t n, const float *lut) {
	const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
	if (i < n) gray[i] = lut[to_u8(gray[i])];
}

/* ===================================================================== */
/* launchers                                                              */
/* ===================================================================== */
void launch_hist_eq(float *gray, int rows, int cols, unsigned *hist256, float *lut256, hipStream_t st) {
	const size_t n = (size_t)rows * cols;
	(void)hipMemsetAsync(hist256, 0, 256 * sizeof(unsigned), st);
	const unsigned nb = (unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, 1024);
	MTFHIP_LAUNCH(k_hist_u8, dim3(nb), dim3(kBlock), 0, st, (const float *)gray, n, hist256);
	MTFHIP_LAUNCH(k_hist_lut, dim3(1), dim3(1), 0, st, (const unsigned *)hist256, (unsigned)n, lut256);
	MTFHIP_LAUNCH(k_apply_lut, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, gray, n, (const float *)lut256);
}
void launch_to_gray(const void *raw, int rows, int cols, size_t stride_bytes, int channels, int depth_f32, float *out, hipStream_t st) {
	MTFHIP_LAUNCH(k_to_gray_f32, dim3((cols + kBlock - 1) / kBlock, rows), dim3(kBlock), 0, st, (const unsigned char *)raw, rows, cols,
		stride_bytes, channels, depth_f32, out);
}
void launch_sym5(const float *src, float *tmp, float *dst, int rows, int cols, const float kx[3], const float ky[3], hipStream_t st) {
	const dim3 grid((cols + kBlock - 1) / kBlock, rows);
	MTFHIP_LAUNCH(k_sym5_rows, grid, dim3(kBlock), 0, st, src, rows, cols, kx[0], kx[1], kx[2], tmp);
	MTFHIP_LAUNCH(k_sym5_cols, grid, dim3(kBlock), 0, st, (const float *)tmp, rows, cols, ky[0], ky[1], ky[2], dst);
}
void launch_pyr_down(const float *src, int srows, int scols, float *dst, int drows, int dcols, hipStream_t st) {
	MTFHIP_LAUNCH(k_pyr_down, dim3((dcols + kBlock - 1) / kBlock, drows), dim3(kBlock), 0, st, src, srows, scols, drows, dcols, dst);
}
void launch_resize_linear(const float *src, int srows, int scols, float *dst, int drows, int dcols, hipStream_t st) {
	MTFHIP_LAUNCH(k_resize_linear, dim3((dcols + kBlock - 1) / kBlock, drows), dim3(kBlock), 0, st, src, srows, scols, drows, dcols, dst);
}

/* pair[2 (y W + x)] = I[y][x], pair[2 (y W + x) + 1] = I[min(y + 1, H - 1)][x]: a bilinear cell's four texels in 16 contiguous bytes */
__global__ __launch_bounds__(kBlock) void k_pair_image(ImgView im, float2 *pair) {
	const size_t n = (size_t)im.w * im.h;
	for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
		const int y = (int)(i / (size_t)im.w), x = (int)(i - (size_t)y * im.w);
		const int y1 = y + 1 < im.h ? y + 1 : y;
		pair[i] = make_float2(im.data[(size_t)y * im.stride + x], im.data[(size_t)y1 * im.stride + x]);
	}
}
void launch_pair_image(const ImgView &im, float *pair, hipStream_t st) {
	const size_t n = (size_t)im.w * im.h;
	const unsigned blocks = (unsigned)std::min<size_t>((n + kBlock - 1) / kBlock, 4096);
	MTFHIP_LAUNCH(k_pair_image, dim3(blocks), dim3(kBlock), 0, st, im, reinterpret_cast<float2 *>(pair));
}

} // namespace mtfhip
