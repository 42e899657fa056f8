/*
 * api_core.hip -- context, image / pre-processing, batch lifecycle and buffers, the StateSpaceModel entry points, timing
 * (C-ABI implementation, include/mtfhip.h; shared declarations: mtfhip_api_internal.h)
 *
 * No CPU fallback exists: every entry point either runs its HIP kernels or returns an error.
 */
#include "mtfhip_api_internal.h"
#include <algorithm>

/* Kernel arguments in device memory instead of host-coherent memory: the fused kernel's first instruction is a
 * scalar load of its 400-byte argument block, and every step is two launches, so the PCIe round trip of that load is
 * 2-3 us of a 65 us step (measured: 68.6 -> 65.6 us/step at 64 targets, 22.0 -> 17.9 us at one).  The HIP runtime
 * reads the variable when it initialises (first HIP call of the process), so this has to run at load time; a value
 * already present in the environment wins. */
__attribute__((constructor)) static void mtfhip_runtime_defaults() { setenv("HIP_FORCE_DEV_KERNARG", "1", 0); }

thread_local std::string g_last_error;
/* sticky launch error (MTFHIP_LAUNCH): reported by the next call that waits for the device */
namespace mtfhip {
static char g_launch_msg[320];
static volatile int g_launch_failed = 0;
void note_launch_error(hipError_t e, const char *file, int line) {
	if (g_launch_failed) return;
	snprintf(g_launch_msg, sizeof(g_launch_msg), "kernel launch failed: %s (%s:%d)", hipGetErrorString(e), file, line);
	g_launch_failed = 1;
	fprintf(stderr, "libmtfhip: %s\n", g_launch_msg);
}
}
int launch_error_pending() {
	if (!mtfhip::g_launch_failed) return MTFHIP_OK;
	mtfhip::g_launch_failed = 0;
	return fail(MTFHIP_ERR_HIP, "%s", mtfhip::g_launch_msg);
}

extern "C" {

/* ------------------------------------------------------------------ context */
const char *mtfhip_last_error(void) { return g_last_error.c_str(); }

int mtfhip_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int mtfhip_ctx_create(int device, void *hip_stream, mtfhip_ctx **out) {
	if (!out) return fail(MTFHIP_ERR_INVALID_ARG, "ctx_create: out is NULL");
	int n = mtfhip_device_count();
	if (n <= 0) return fail(MTFHIP_ERR_NO_DEVICE, "no HIP device visible");
	if (device < 0 || device >= n) return fail(MTFHIP_ERR_INVALID_ARG, "device %d out of range [0,%d)", device, n);
	HIP_TRY(hipSetDevice(device));
	mtfhip_ctx *c = new mtfhip_ctx();
	c->device = device;
	if (hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); c->n_cus = 0; }
	if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->own_stream = false; }
	else {
		hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
		if (e != hipSuccess) { delete c; return fail(MTFHIP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
		c->own_stream = true;
	}
	*out = c;
	return MTFHIP_OK;
}

/* The destroy calls may run from a host-language finaliser after the HIP runtime has begun tearing itself
 * down at process exit (its calls then throw from inside the runtime); nothing may escape a C entry point. */
void mtfhip_ctx_destroy(mtfhip_ctx *c) {
	if (!c) return;
	try {
		(void)hipSetDevice(c->device);
		(void)hipStreamSynchronize(c->stream);
		for (auto &kv : c->timers)
			for (auto &p : kv.second.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
		for (auto e : c->free_events) (void)hipEventDestroy(e);
		if (c->img_owned) (void)hipFree(c->img_owned);
		if (c->prev_owned) (void)hipFree(c->prev_owned);
		if (c->pair_owned) (void)hipFree(c->pair_owned);
		if (c->raw) (void)hipFree(c->raw);
		if (c->tmp_a) (void)hipFree(c->tmp_a);
		if (c->tmp_b) (void)hipFree(c->tmp_b);
		for (int q = 0; q < 3; ++q) {
			if (c->extra_streams[q]) (void)hipStreamDestroy(c->extra_streams[q]);
			if (c->ev_join[q]) (void)hipEventDestroy(c->ev_join[q]);
		}
		if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
		if (c->d_phase) (void)hipFree(c->d_phase);
		if (c->ev_ref) (void)hipEventDestroy(c->ev_ref);
		if (c->own_stream) (void)hipStreamDestroy(c->stream);
	} catch (...) {
	}
	delete c;
}

int mtfhip_ctx_synchronize(mtfhip_ctx *c) {
	if (!c) return fail(MTFHIP_ERR_INVALID_ARG, "ctx is NULL");
	TRY(launch_error_pending());
	HIP_TRY(hipStreamSynchronize(c->stream));
	return MTFHIP_OK;
}
void *mtfhip_ctx_stream(mtfhip_ctx *c) { return c ? (void *)c->stream : nullptr; }

int mtfhip_image_upload(mtfhip_ctx *c, const float *host_img, int height, int width, int row_stride) {
	return mtfhip_image_upload_mc(c, host_img, height, width, row_stride, 1);
}
/* CV_32FC3: `channels` interleaved floats per pixel, row_stride in floats */
int mtfhip_image_upload_mc(mtfhip_ctx *c, const float *host_img, int height, int width, int row_stride, int channels) {
	if (!c || !host_img) return fail(MTFHIP_ERR_INVALID_ARG, "image_upload: NULL argument");
	TRY(lazy_flush_ctx(c));
	if (channels != 1 && channels != 3) return fail(MTFHIP_ERR_INVALID_ARG, "image_upload: %d channels (1 or 3 expected)", channels);
	if (height <= 0 || width <= 0 || row_stride < width * channels) return fail(MTFHIP_ERR_INVALID_ARG, "image_upload: bad shape %dx%d stride %d", height, width, row_stride);
	if ((double)height * width * channels * 4.0 >= 4294967296.0) return fail(MTFHIP_ERR_INVALID_ARG, "image_upload: %dx%dx%d floats exceed the 4 GiB a 32-bit texel offset can address", height, width, channels);
	if (height >= (1 << 24) || (double)width * channels >= (double)(1 << 24)) return fail(MTFHIP_ERR_INVALID_ARG, "image_upload: %d rows of %d floats: 2^24 or more of either (the samplers' 24-bit row x pitch product)", height, width * channels);
	HIP_TRY(hipSetDevice(c->device));
	const int logical_width = width;
	width *= channels;   /* floats per row */
	size_t need = (size_t)height * width;
	if (need > c->img_capacity) {
		if (c->img_owned) HIP_TRY(hipFree(c->img_owned));
		c->img_owned = nullptr;
		HIP_TRY(hipMalloc(&c->img_owned, need * sizeof(float)));
		c->img_capacity = need;
	}
	HIP_TRY(hipMemcpy2DAsync(c->img_owned, (size_t)width * sizeof(float), host_img, (size_t)row_stride * sizeof(float),
		(size_t)width * sizeof(float), (size_t)height, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream)); /* the caller may overwrite its buffer right after (TrackerBase.h:22-26) */
	c->img = ImgView{c->img_owned, height, logical_width, width, channels};
	++c->img_serial;
	return MTFHIP_OK;
}

int mtfhip_image_borrow(mtfhip_ctx *c, const float *dev_img, int height, int width, int row_stride) {
	if (c) TRY(lazy_flush_ctx(c));
	if (!c || !dev_img) return fail(MTFHIP_ERR_INVALID_ARG, "image_borrow: NULL argument");
	if (height <= 0 || width <= 0 || row_stride < width) return fail(MTFHIP_ERR_INVALID_ARG, "image_borrow: bad shape");
	if ((double)height * row_stride * 4.0 >= 4294967296.0 || height >= (1 << 24) || row_stride >= (1 << 24))
		return fail(MTFHIP_ERR_INVALID_ARG, "image_borrow: %d rows of %d floats exceed what a 32-bit texel offset / a 24-bit row x pitch product can address", height, row_stride);
	c->img = ImgView{dev_img, height, width, row_stride};
	++c->img_serial;
	return MTFHIP_OK;
}

/* the row-pair copy of the current image (mtfhip_ctx::pair_owned): one launch per image change, on the context's stream */
const float *ensure_pair_image(mtfhip_ctx *c) {
	const char *e = std::getenv("MTFHIP_PAIR_IMAGE");   /* (read per call: the tests compare the two forms in one process) */
	if (e && e[0] == '0') return nullptr;
	const ImgView &im = c->img;
	if (!im.data || im.channels != 1 || (im.data != c->img_owned && im.data != c->prev_owned)) return nullptr;
	const size_t n = (size_t)im.w * im.h;
	if (n * 8 >= 4294967296ull || im.w < 2 || im.h < 2) return nullptr;   /* (32-bit byte offsets into the pair image) */
	if (c->pair_serial == c->img_serial && c->pair_owned) return c->pair_owned;
	if (hipSetDevice(c->device) != hipSuccess) return nullptr;
	if (2 * n > c->pair_capacity) {
		if (c->pair_owned) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->pair_owned); }
		c->pair_owned = nullptr; c->pair_capacity = 0;
		if (hipMalloc(&c->pair_owned, 2 * n * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
		c->pair_capacity = 2 * n;
	}
	launch_pair_image(im, c->pair_owned, c->stream);
	c->pair_serial = c->img_serial;
	return c->pair_owned;
}

/* The copy costs a launch of its own per image (~8 us for a 1024 x 1024 frame) and saves ~3 us per 10 000 candidates scored on it: a filter that
 * scores 500 particles once per frame (the shipped pf_n_particles 500, pf_max_iters 1) must not pay for it, one that scores 10 000 particles
 * ten times per frame should.  So the copy of an image is built once 30 000 candidates have been scored on that image without it
 * (MTFHIP_PAIR_IMAGE_AFTER moves the threshold; 0 = at the first launch). */
const float *pair_image_if_it_pays(mtfhip_ctx *c, int n_candidates) {
	if (c->pair_serial == c->img_serial && c->pair_owned) return ensure_pair_image(c);   /* (there already; the call still honours MTFHIP_PAIR_IMAGE=0) */
	const char *e_after = std::getenv("MTFHIP_PAIR_IMAGE_AFTER");   /* (read per call: the tests set it) */
	const long after = e_after ? std::atol(e_after) : 30000;
	if (c->pair_demand_serial != c->img_serial) { c->pair_demand_serial = c->img_serial; c->pair_demand = 0; }
	if ((long)c->pair_demand >= after) return ensure_pair_image(c);
	c->pair_demand += (size_t)(n_candidates > 0 ? n_candidates : 0);
	return nullptr;
}

/* prev_img = curr_img.clone() (SM/src/GridTracker.cc:241-243, 266) */
int mtfhip_image_keep_prev(mtfhip_ctx *c) {
	if (!c) return fail(MTFHIP_ERR_INVALID_ARG, "image_keep_prev: NULL argument");
	if (!c->img.data) return fail(MTFHIP_ERR_LOGIC, "image_keep_prev: no current image");
	if (c->img.data == c->prev.data) { c->prev = c->img; return MTFHIP_OK; }   /* kept already, no frame since */
	if (c->img.data == c->img_owned) {
		/* the context's own buffer: it becomes the previous frame's and the next upload / pre-processing pass fills the other one */
		std::swap(c->img_owned, c->prev_owned);
		std::swap(c->img_capacity, c->prev_capacity);
		c->prev = c->img;
		return MTFHIP_OK;
	}
	HIP_TRY(hipSetDevice(c->device));
	const size_t row = (size_t)c->img.w * c->img.channels, need = row * c->img.h;
	if (need > c->prev_capacity) {
		if (c->prev_owned) HIP_TRY(hipFree(c->prev_owned));
		c->prev_owned = nullptr; c->prev_capacity = 0;
		HIP_TRY(hipMalloc(&c->prev_owned, need * sizeof(float)));
		c->prev_capacity = need;
	}
	HIP_TRY(hipMemcpy2DAsync(c->prev_owned, row * sizeof(float), c->img.data, (size_t)c->img.stride * sizeof(float), row * sizeof(float), (size_t)c->img.h,
		hipMemcpyDeviceToDevice, c->stream));
	c->prev = ImgView{c->prev_owned, c->img.h, c->img.w, (int)row, c->img.channels};
	return MTFHIP_OK;
}
int mtfhip_image_has_prev(mtfhip_ctx *c) { return c && c->prev.data ? 1 : 0; }
/* tracker->setImage(prev_img) ... tracker->setImage(curr_img) (GridTracker.cc:300-304): the two views change places */
int mtfhip_image_swap_prev(mtfhip_ctx *c) {
	if (!c) return fail(MTFHIP_ERR_INVALID_ARG, "image_swap_prev: NULL argument");
	if (!c->prev.data || !c->img.data) return fail(MTFHIP_ERR_LOGIC, "image_swap_prev: no previous image (mtfhip_image_keep_prev)");
	TRY(lazy_flush_ctx(c));
	std::swap(c->img, c->prev);
	++c->img_serial;
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ pre-processing / pyramid */
static int ensure_image(mtfhip_ctx *c, int rows, int cols) {
	const size_t need = (size_t)rows * cols;
	if (need > c->img_capacity) {
		if (c->img_owned) HIP_TRY(hipFree(c->img_owned));
		c->img_owned = nullptr;
		HIP_TRY(hipMalloc(&c->img_owned, need * sizeof(float)));
		c->img_capacity = need;
	}
	return MTFHIP_OK;
}
static int ensure_tmp(mtfhip_ctx *c, size_t need) {
	if (need > c->tmp_capacity) {
		if (c->tmp_a) HIP_TRY(hipFree(c->tmp_a));
		if (c->tmp_b) HIP_TRY(hipFree(c->tmp_b));
		c->tmp_a = c->tmp_b = nullptr;
		HIP_TRY(hipMalloc(&c->tmp_a, need * sizeof(float)));
		HIP_TRY(hipMalloc(&c->tmp_b, need * sizeof(float)));
		c->tmp_capacity = need;
	}
	return MTFHIP_OK;
}
/* cv::getGaussianKernel(ksize, sigma, CV_32F) for sigma > 0: exp(-x^2 / (2 sigma^2)) rounded to float, normalised by the
 * double sum of those floats; k[0] is the centre tap, k[1], k[2] the taps one and two samples out */
static void gaussian5(double sigma, float k[3]) {
	const double scale2x = -0.5 / (sigma * sigma);
	float cf[5];
	double sum = 0;
	for (int i = 0; i < 5; ++i) { const double x = i - 2.0; cf[i] = (float)std::exp(scale2x * x * x); sum += cf[i]; }
	sum = 1. / sum;
	for (int i = 0; i < 5; ++i) cf[i] = (float)(cf[i] * sum);
	k[0] = cf[2]; k[1] = cf[3]; k[2] = cf[4];
}

int mtfhip_image_preprocess(mtfhip_ctx *c, const void *host_raw, int rows, int cols, int row_stride_bytes, int channels, int depth,
	int ksize, double sigma_x, double sigma_y) {
	return mtfhip_image_preprocess_ex(c, host_raw, rows, cols, row_stride_bytes, channels, depth, ksize, sigma_x, sigma_y, 0, 1.0);
}
/* PreProcBase::processFrame for the CV_32FC1 output with all of its switches (Utilities/src/preprocUtils.cc:108-137): gray ->
 * [hist_eq: to 8 bit, cv::equalizeHist, back to float] -> apply() (the Gaussian smoothing, or nothing) -> [resize_factor != 1:
 * cv::resize(INTER_LINEAR) to (int)(rows f) x (int)(cols f)] */
int mtfhip_image_preprocess_ex(mtfhip_ctx *c, const void *host_raw, int rows, int cols, int row_stride_bytes, int channels, int depth,
	int ksize, double sigma_x, double sigma_y, int hist_eq, double resize_factor) {
	if (c) TRY(lazy_flush_ctx(c));
	if (!c || !host_raw) return fail(MTFHIP_ERR_INVALID_ARG, "image_preprocess: NULL argument");
	if (rows <= 0 || cols <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "image_preprocess: bad shape %dx%d", rows, cols);
	if (channels != 1 && channels != 3) return fail(MTFHIP_ERR_INVALID_ARG, "image_preprocess: %d channels (1 or 3 expected)", channels);
	if (depth != MTFHIP_DEPTH_U8 && depth != MTFHIP_DEPTH_F32) return fail(MTFHIP_ERR_INVALID_ARG, "PreProcBase::processFrame : Invalid input image depth provided: %d", depth);
	const size_t px = (size_t)channels * (depth == MTFHIP_DEPTH_F32 ? 4 : 1);
	if ((size_t)row_stride_bytes < px * cols) return fail(MTFHIP_ERR_INVALID_ARG, "image_preprocess: row stride %d shorter than a row", row_stride_bytes);
	if (ksize != 0 && ksize != 5) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "image_preprocess: Gaussian kernel size %d (5, or 0 for no smoothing)", ksize);
	if (ksize == 5 && sigma_x <= 0) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "image_preprocess: sigma <= 0 selects OpenCV's fixed kernel table, not available");
	if (!(resize_factor > 0)) return fail(MTFHIP_ERR_INVALID_ARG, "image_preprocess: resize_factor must be positive");
	const bool resize = resize_factor != 1.0;
	const int orows = resize ? (int)(rows * resize_factor) : rows, ocols = resize ? (int)(cols * resize_factor) : cols;   /* static_cast<int>(rows * resize_factor), :63-64 */
	if (orows <= 0 || ocols <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "image_preprocess: resize_factor %g leaves no pixels", resize_factor);
	HIP_TRY(hipSetDevice(c->device));
	const size_t raw_bytes = px * cols * rows;
	if (raw_bytes > c->raw_capacity) {
		if (c->raw) HIP_TRY(hipFree(c->raw));
		c->raw = nullptr;
		HIP_TRY(hipMalloc(&c->raw, raw_bytes));
		c->raw_capacity = raw_bytes;
	}
	TRY(ensure_image(c, std::max(rows, orows), std::max(cols, ocols)));
	TRY(ensure_tmp(c, (size_t)rows * cols + 512));   /* (+ the 256-bin histogram and the look-up table of hist_eq behind tmp_b) */
	HIP_TRY(hipMemcpy2DAsync(c->raw, px * cols, host_raw, (size_t)row_stride_bytes, px * cols, (size_t)rows, hipMemcpyHostToDevice, c->stream));
	{
		TimedScope ts(c, "preprocess");
		/* stages ping-pong between tmp_a and the image buffer so that the last one lands in the image */
		const int n_after = (ksize ? 1 : 0) + (resize ? 1 : 0);
		float *gray = n_after == 1 ? c->tmp_a : c->img_owned;   /* 0 or 2 later stages: start in the image buffer */
		launch_to_gray(c->raw, rows, cols, px * cols, channels, depth == MTFHIP_DEPTH_F32, gray, c->stream);
		if (hist_eq) {
			unsigned *hist = reinterpret_cast<unsigned *>(c->tmp_b + (size_t)rows * cols);
			launch_hist_eq(gray, rows, cols, hist, reinterpret_cast<float *>(hist + 256), c->stream);
		}
		float *cur = gray;
		if (ksize) {
			float kx[3], ky[3];
			gaussian5(sigma_x, kx);
			gaussian5(sigma_y > 0 ? sigma_y : sigma_x, ky);   /* sigma2 <= 0 -> sigma2 = sigma1 (createGaussianKernels) */
			float *dst = cur == c->tmp_a ? c->img_owned : c->tmp_a;
			launch_sym5(cur, c->tmp_b, dst, rows, cols, kx, ky, c->stream);
			cur = dst;
		}
		if (resize) {
			float *dst = cur == c->tmp_a ? c->img_owned : c->tmp_a;
			launch_resize_linear(cur, rows, cols, dst, orows, ocols, c->stream);
			cur = dst;
		}
		if (cur != c->img_owned) return fail(MTFHIP_ERR_LOGIC, "image_preprocess: stage bookkeeping");
	}
	HIP_TRY(hipStreamSynchronize(c->stream)); /* the caller may reuse its frame buffer */
	c->img = ImgView{c->img_owned, orows, ocols, ocols};
	++c->img_serial;
	return MTFHIP_OK;
}

int mtfhip_image_pyramid_level(mtfhip_ctx *dst, mtfhip_ctx *src, int dst_rows, int dst_cols, int use_pyr_down) {
	if (dst) TRY(lazy_flush_ctx(dst));
	if (!dst || !src) return fail(MTFHIP_ERR_INVALID_ARG, "image_pyramid_level: NULL argument");
	if (!src->img.data) return fail(MTFHIP_ERR_LOGIC, "image_pyramid_level: the source context has no image");
	if (dst == src) return fail(MTFHIP_ERR_INVALID_ARG, "image_pyramid_level: source and destination contexts must differ");
	if (dst->device != src->device) return fail(MTFHIP_ERR_INVALID_ARG, "image_pyramid_level: contexts live on different devices");
	if (dst_rows <= 0 || dst_cols <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "image_pyramid_level: bad destination shape");
	if (src->img.stride != src->img.w) return fail(MTFHIP_ERR_NOT_IMPLEMENTED, "image_pyramid_level: padded source rows");
	const int sr = src->img.h, sc = src->img.w;
	if (use_pyr_down && (std::abs(dst_cols * 2 - sc) > 2 || std::abs(dst_rows * 2 - sr) > 2))   /* cv::pyrDown's own assertion */
		return fail(MTFHIP_ERR_INVALID_ARG, "pyrDown: destination %dx%d is not half of %dx%d", dst_rows, dst_cols, sr, sc);
	HIP_TRY(hipSetDevice(dst->device));
	HIP_TRY(hipStreamSynchronize(src->stream));
	TRY(ensure_image(dst, dst_rows, dst_cols));
	{
		TimedScope ts(dst, "pyramid_level");
		if (use_pyr_down) launch_pyr_down(src->img.data, sr, sc, dst->img_owned, dst_rows, dst_cols, dst->stream);
		else {   /* cv::resize + GaussianBlur(5x5, 3), PyramidalTracker.cc:93-94 */
			TRY(ensure_tmp(dst, (size_t)dst_rows * dst_cols));
			float k[3];
			gaussian5(3.0, k);
			launch_resize_linear(src->img.data, sr, sc, dst->tmp_a, dst_rows, dst_cols, dst->stream);
			launch_sym5(dst->tmp_a, dst->tmp_b, dst->img_owned, dst_rows, dst_cols, k, k, dst->stream);
		}
	}
	dst->img = ImgView{dst->img_owned, dst_rows, dst_cols, dst_cols};
	++dst->img_serial;
	return MTFHIP_OK;
}

int mtfhip_image_download(mtfhip_ctx *c, float *host_img, int rows, int cols) {
	if (!c || !host_img) return fail(MTFHIP_ERR_INVALID_ARG, "image_download: NULL argument");
	if (!c->img.data) return fail(MTFHIP_ERR_LOGIC, "image_download: no current image");
	if (rows != c->img.h || cols != c->img.w) return fail(MTFHIP_ERR_INVALID_ARG, "image_download: the image is %dx%d", c->img.h, c->img.w);
	HIP_TRY(hipMemcpy2DAsync(host_img, (size_t)cols * sizeof(float), c->img.data, (size_t)c->img.stride * sizeof(float),
		(size_t)cols * sizeof(float), (size_t)rows, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(hipStreamSynchronize(c->stream));
	return MTFHIP_OK;
}
int mtfhip_image_shape(mtfhip_ctx *c, int *rows, int *cols) {
	if (!c || !rows || !cols) return fail(MTFHIP_ERR_INVALID_ARG, "image_shape: NULL argument");
	*rows = c->img.h; *cols = c->img.w;
	return MTFHIP_OK;
}

/* ------------------------------------------------------------------ batch */
int mtfhip_batch_create(mtfhip_ctx *c, const mtfhip_patch_desc *d, int n_targets, mtfhip_batch **out) {
	if (!c || !d || !out) return fail(MTFHIP_ERR_INVALID_ARG, "batch_create: NULL argument");
	/* ImageBase ctor AM/src/ImageBase.cc:33-35, StateSpaceModel ctor StateSpaceModel.h:58-60 */
	if (d->resx <= 0 || d->resy <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "Invalid sampling resolution provided");
	if (n_targets <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "batch_create: n_targets must be positive");
	/* the fused kernel addresses a target's arrays with 32-bit byte offsets (ld_off / st_off): 8 columns of N doubles */
	if ((double)d->resx * d->resy * 3.0 >= (double)(1u << 26)) return fail(MTFHIP_ERR_INVALID_ARG, "batch_create: %dx%d sample points per target exceed the 2^26-row limit", d->resx, d->resy);
	if (d->grad_eps <= 0 || d->hess_eps < 0) return fail(MTFHIP_ERR_INVALID_ARG, "batch_create: grad_eps must be positive (got %g)", d->grad_eps);
	if (d->am < MTFHIP_AM_SSD || d->am > MTFHIP_AM_MI) return fail(MTFHIP_ERR_INVALID_ARG, "unknown appearance model %d", d->am);
	if (d->am == MTFHIP_AM_MI && (d->mi_n_bins < 2 || d->mi_n_bins > MI_NB)) return fail(MTFHIP_ERR_INVALID_ARG, "MI: n_bins %d outside [2, %d]", d->mi_n_bins, (int)MI_NB);
	if (d->am == MTFHIP_AM_MI && d->mi_partition_of_unity && d->mi_n_bins < 4) /* MI.cc:83-87 */
		return fail(MTFHIP_ERR_INVALID_ARG, "MI::Too few bins %d specified to enforce partition of unity constraint", d->mi_n_bins);
	if (d->ssm != MTFHIP_SSM_HOMOGRAPHY && d->ssm != MTFHIP_SSM_AFFINE) return fail(MTFHIP_ERR_INVALID_ARG, "unknown state space model %d", d->ssm);
	if (d->n_channels != 0 && d->n_channels != 1 && d->n_channels != 3) return fail(MTFHIP_ERR_INVALID_ARG, "n_channels %d (1 or 3 expected)", d->n_channels);
	HIP_TRY(hipSetDevice(c->device));
	mtfhip_batch *b = new mtfhip_batch();
	b->ctx = c; b->desc = *d; b->B = n_targets;
	b->C = d->n_channels > 1 ? d->n_channels : 1;
	b->NP = d->resx * d->resy; b->N = b->NP * b->C;   /* ImageBase: patch_size = n_pix * n_channels */
	b->S = d->ssm == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	if (d->am == MTFHIP_AM_MI) {
		/* MI ctor AM/src/MI.cc:80-94 */
		double lo = 0, hi = d->mi_n_bins - 1;
		if (d->mi_partition_of_unity) { lo = 1; hi = d->mi_n_bins - 2; }
		b->norm_mult = (hi - lo) / (255.0 - 0.0 + 1);
		b->norm_add = lo;
	}
	const size_t N = b->N, S = b->S, NP = b->NP;
	size_t per[MTFHIP_BUF_COUNT] = {N, N, 2 * N, 2 * N, N, N, N * S, N * S, N * S, 2 * NP, 2 * NP, 8 * NP, NP, NP, 2 * NP, 2 * NP,
		4 * N, 4 * N, 16 * NP, S * S * N, S * S * N, S * S * N};
	b->hess_eps = d->hess_eps > 0 ? d->hess_eps : 1.0; /* HESS_EPS, AM/include/mtf/AM/ImageBase.h:9 */
	for (int i = 0; i < MTFHIP_BUF_COUNT; ++i) { b->per_target[i] = per[i]; b->buf[i] = nullptr; }
	b->th.resize(n_targets);
	for (auto &h : b->th) { std::memset(&h, 0, sizeof(h)); h.warp = m3_identity(); }
	b->nblk_max = simple_blocks_per_target(b->N);
	int nf = fused_blocks_per_target(b->N, 1);   /* the finest decomposition is the single-target one */
	if (nf > b->nblk_max) b->nblk_max = nf;
	auto cleanup = [&](int code) { mtfhip_batch_destroy(b); return code; };
	const int eager[] = {MTFHIP_BUF_I0, MTFHIP_BUF_IT, MTFHIP_BUF_DI0_DX, MTFHIP_BUF_DIT_DX, MTFHIP_BUF_DF_DI0,
		MTFHIP_BUF_DF_DIT, MTFHIP_BUF_J0, MTFHIP_BUF_JT, MTFHIP_BUF_INIT_PTS, MTFHIP_BUF_CURR_PTS,
		MTFHIP_BUF_INIT_Z, MTFHIP_BUF_CURR_Z, MTFHIP_BUF_INIT_HXY, MTFHIP_BUF_CURR_HXY};
	for (int id : eager) { int r = ensure_buf(b, id); if (r) return cleanup(r); }
	{
		const size_t Bt = (size_t)n_targets, d = sizeof(double);
		b->slab_dbl_bytes = 54 * Bt * d;
		b->slab_bytes = b->slab_dbl_bytes + 2 * sizeof(int) * Bt;
		if (hipMalloc(&b->d_slab, b->slab_bytes) != hipSuccess || hipHostMalloc(&b->h_stage_a, b->slab_bytes) != hipSuccess ||
			hipHostMalloc(&b->h_stage_b, b->slab_bytes) != hipSuccess || hipEventCreateWithFlags(&b->ev_a, hipEventDisableTiming) != hipSuccess ||
			hipEventCreateWithFlags(&b->ev_b, hipEventDisableTiming) != hipSuccess ||
			hipHostMalloc(&b->h_wstage[0], 17 * Bt * d) != hipSuccess || hipHostMalloc(&b->h_wstage[1], 17 * Bt * d) != hipSuccess ||
			hipEventCreateWithFlags(&b->ev_w[0], hipEventDisableTiming) != hipSuccess ||
			hipEventCreateWithFlags(&b->ev_w[1], hipEventDisableTiming) != hipSuccess)
			return cleanup(fail(MTFHIP_ERR_HIP, "allocation of the per-target state slab failed"));
		double *p = reinterpret_cast<double *>(b->d_slab);
		b->d_warps = p; b->d_states = p + 9 * Bt; b->d_corners = p + 17 * Bt; b->d_init_corners_hm = p + 25 * Bt;
		b->d_ncc = p + 37 * Bt; b->d_w0 = p + 45 * Bt;
		b->d_active = reinterpret_cast<int *>(b->d_slab + b->slab_dbl_bytes); b->d_iters = b->d_active + Bt;
		(void)hipMemsetAsync(b->d_slab, 0, b->slab_bytes, c->stream);
	}
#define ALLOC(ptr, bytes) do { if (hipMalloc(&(ptr), (bytes)) != hipSuccess) return cleanup(fail(MTFHIP_ERR_HIP, "hipMalloc(%zu) failed", (size_t)(bytes))); } while (0)
	ALLOC(b->d_partials, sizeof(double) * kAccRowMax * b->nblk_max * n_targets);
	ALLOC(b->d_acc, sizeof(double) * kAccRowMax * n_targets);
	ALLOC(b->d_scratch_pts, sizeof(double) * 18 * NP * n_targets); /* largest upload: pts (2 NP) + hess_pts (16 NP) */
	ALLOC(b->d_h0, sizeof(double) * 64 * n_targets);
	ALLOC(b->d_h0inv, sizeof(double) * 64 * n_targets);
	ALLOC(b->d_colmean, sizeof(double) * 8 * n_targets);
	if (d->am == MTFHIP_AM_MI) {
		const int nb = d->mi_n_bins;
		b->mi_row_len = std::max(std::max(nb + 2 * nb * nb, 36 + nb * nb * b->S), nb <= 10 ? mi_fast_row_len(nb) : 0);   /* widest of the MI partial rows (fused passes incl.) */
		/* hist_norm_mult = 1 / (patch_size + hist_pre_seed * n_bins), hist_pre_seed = n_bins * pre_seed (MI.cc:97,104) */
		b->mi_hist_norm = 1.0 / ((double)b->N + (nb * d->mi_pre_seed) * nb);
		ALLOC(b->d_mi_tb, sizeof(double) * MI_SIZE * n_targets);
		ALLOC(b->d_mi_part, sizeof(double) * (size_t)b->mi_row_len * b->nblk_max * n_targets);
		ALLOC(b->d_mi_f, sizeof(double) * n_targets);
		ALLOC(b->d_mi_red, sizeof(double) * (size_t)b->mi_row_len * n_targets);
		ALLOC(b->d_mi_H, sizeof(double) * (64 + 16 + 64) * n_targets);   /* [B][64] Hessians, [B][16] Jacobian sums of the fused pass, [B][64] second Hessian (SumOfStd) */
		(void)hipMemsetAsync(b->d_mi_tb, 0, sizeof(double) * MI_SIZE * n_targets, c->stream);
	}
#undef ALLOC
	if (hipHostMalloc(&b->h_acc, sizeof(double) * kAccRowMax * n_targets, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
		return cleanup(fail(MTFHIP_ERR_HIP, "hipHostMalloc failed"));
	{
		const char *zc = std::getenv("MTFHIP_ZERO_COPY");
		void *dp = nullptr, *fp = nullptr;
		if (!(zc && zc[0] == '0') && hipHostGetDevicePointer(&dp, b->h_acc, 0) == hipSuccess &&
			hipHostMalloc(&b->h_flag, sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
			hipHostGetDevicePointer(&fp, b->h_flag, 0) == hipSuccess && hipMalloc(&b->d_fin_count, sizeof(int)) == hipSuccess) {
			*b->h_flag = 0;
			(void)hipMemsetAsync(b->d_fin_count, 0, sizeof(int), c->stream);
			b->h_acc_dev = static_cast<double *>(dp); b->h_flag_dev = static_cast<unsigned long long *>(fp);
			void *pp = nullptr;
			if (hipHostMalloc(&b->h_pub, b->slab_bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
				hipHostGetDevicePointer(&pp, b->h_pub, 0) == hipSuccess) b->h_pub_dev = static_cast<char *>(pp);
			else (void)hipGetLastError();
			void *sp = nullptr;
			if (hipHostGetDevicePointer(&sp, b->h_stage_a, 0) == hipSuccess) b->h_stage_a_dev = static_cast<char *>(sp);
			else (void)hipGetLastError();
			if (hipHostGetDevicePointer(&sp, b->h_stage_b, 0) == hipSuccess) b->h_stage_b_dev = static_cast<char *>(sp);
			else (void)hipGetLastError();
		} else (void)hipGetLastError();
	}
	(void)hipMemsetAsync(b->d_partials, 0, sizeof(double) * kAccRowMax * b->nblk_max * n_targets, c->stream);
	int r = push_warps(b);
	if (r) return cleanup(r);
	{
		const char *iw_env = std::getenv("MTFHIP_INLINE_WARP");
		b->inline_warp_ok = b->B == 1 && !(iw_env && iw_env[0] == '0') && kernarg_layout_verified(c->stream);
		const char *lazy_env = std::getenv("MTFHIP_LAZY");
		/* SSD and NCC have a fused kernel each; MI has its fused passes */
		b->lz.enabled = (d->am == MTFHIP_AM_SSD || d->am == MTFHIP_AM_NCC || d->am == MTFHIP_AM_MI) && b->C == 1 &&
			!(lazy_env && lazy_env[0] == '0');
	}
	c->batches.push_back(b);
	*out = b;
	return MTFHIP_OK;
}

void mtfhip_batch_destroy(mtfhip_batch *b) {
	if (!b) return;
	try {
		auto &reg = b->ctx->batches;
		reg.erase(std::remove(reg.begin(), reg.end(), b), reg.end());
		(void)hipSetDevice(b->ctx->device);
		(void)hipStreamSynchronize(b->ctx->stream);
		for (int i = 0; i < MTFHIP_BUF_COUNT; ++i)
			if (b->buf[i]) (void)hipFree(b->buf[i]);
		void *ptrs[] = {b->d_slab, b->d_partials, b->d_acc, b->d_scratch_pts, b->d_h0,
			b->d_cand, b->d_colmean, b->d_mi_tb, b->d_mi_part,
			b->d_mi_f, b->d_mi_H, b->d_h0inv, b->d_d2_part, b->d_d2_out, b->d_d2_w, b->d_it_shadow, b->d_ncc_tm, b->d_mi_red, b->d_lm, b->d_persist, b->d_trace, b->d_cand_mi, b->d_mi_poly, b->d_nn_warps, b->d_fb};
		for (void *p : ptrs)
			if (p) (void)hipFree(p);
		if (b->h_fb) (void)hipHostFree(b->h_fb);
		if (b->h_init_rec) (void)hipHostFree(b->h_init_rec);
		if (b->h_init_flag) (void)hipHostFree(b->h_init_flag);
		if (b->h_acc) (void)hipHostFree(b->h_acc);
		if (b->h_flag) (void)hipHostFree(b->h_flag);
		if (b->h_pub) (void)hipHostFree(b->h_pub);
		if (b->d_fin_count) (void)hipFree(b->d_fin_count);
		if (b->h_stage_a) (void)hipHostFree(b->h_stage_a);
		if (b->h_stage_b) (void)hipHostFree(b->h_stage_b);
		if (b->ev_a) (void)hipEventDestroy(b->ev_a);
		if (b->ev_b) (void)hipEventDestroy(b->ev_b);
		for (int k = 0; k < 2; ++k) {
			if (b->h_wstage[k]) (void)hipHostFree(b->h_wstage[k]);
			if (b->ev_w[k]) (void)hipEventDestroy(b->ev_w[k]);
		}
	} catch (...) {
	}
	delete b;
}

int mtfhip_batch_n_targets(const mtfhip_batch *b) { return b ? b->B : 0; }
int mtfhip_batch_set_math_mode(mtfhip_batch *b, int mode) {
	FLUSH(b);
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "set_math_mode: NULL batch");
	if (mode != MTFHIP_MATH_REPLAY && mode != MTFHIP_MATH_FAST) return fail(MTFHIP_ERR_INVALID_ARG, "set_math_mode: unknown mode %d", mode);
	b->math_mode = mode;
	return MTFHIP_OK;
}
int mtfhip_batch_get_math_mode(const mtfhip_batch *b) { return b ? b->math_mode : -1; }
int mtfhip_batch_n_pix(const mtfhip_batch *b) { return b ? b->NP : 0; }          /* ImageBase::getNPix */
int mtfhip_batch_patch_size(const mtfhip_batch *b) { return b ? b->N : 0; }      /* ImageBase::getPatchSize = n_pix * n_channels */
int mtfhip_batch_state_size(const mtfhip_batch *b) { return b ? b->S : 0; }

int mtfhip_batch_read(mtfhip_batch *b, int id, double *dst) {
	if (!b || !dst || id < 0 || id >= MTFHIP_BUF_COUNT) return fail(MTFHIP_ERR_INVALID_ARG, "batch_read: bad argument");
	FLUSH(b);
	if (id == MTFHIP_BUF_DF_DI0 || id == MTFHIP_BUF_DF_DIT) TRY(ensure_df(b));
	if (!b->buf[id]) return fail(MTFHIP_ERR_LOGIC, "batch_read: buffer %d was never produced", id);
	if ((id == MTFHIP_BUF_IT && !b->it_valid) || (id == MTFHIP_BUF_DIT_DX && !b->dit_valid) || (id == MTFHIP_BUF_JT && !b->jt_valid))
		return fail(MTFHIP_ERR_LOGIC, "batch_read: buffer %d is not materialised (last fused iteration ran with materialize=0)", id);
	HIP_TRY(hipMemcpyAsync(dst, b->buf[id], sizeof(double) * b->per_target[id] * b->B, hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	return MTFHIP_OK;
}

int mtfhip_batch_write(mtfhip_batch *b, int id, const double *src) {
	if (!b || !src || id < 0 || id >= MTFHIP_BUF_COUNT) return fail(MTFHIP_ERR_INVALID_ARG, "batch_write: bad argument");
	FLUSH(b);
	TRY(ensure_df(b));
	if (id == MTFHIP_BUF_DF_DI0 || id == MTFHIP_BUF_DF_DIT) stale_clear(b, true, true);
	touch(b, id); ++b->lz.epoch;
	TRY(ensure_buf(b, id));
	HIP_TRY(hipMemcpyAsync(b->buf[id], src, sizeof(double) * b->per_target[id] * b->B, hipMemcpyHostToDevice, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	if (id == MTFHIP_BUF_IT) b->it_valid = true;
	if (id == MTFHIP_BUF_DIT_DX) b->dit_valid = true;
	if (id == MTFHIP_BUF_JT) b->jt_valid = true;
	if (id == MTFHIP_BUF_J0 || id == MTFHIP_BUF_DI0_DX || id == MTFHIP_BUF_INIT_PTS || id == MTFHIP_BUF_INIT_Z) b->j0_is_template = false;
	/* a caller that supplies its own homogeneous grid gets the general (non unit-z) kernels */
	if (id == MTFHIP_BUF_INIT_Z || id == MTFHIP_BUF_INIT_HXY) b->unit_z = 0;
	if (id == MTFHIP_BUF_INIT_PTS || id == MTFHIP_BUF_INIT_Z || id == MTFHIP_BUF_INIT_HXY) b->grid_from_corners = false;
	return MTFHIP_OK;
}

void *mtfhip_batch_device_ptr(mtfhip_batch *b, int id) {
	if (!b || id < 0 || id >= MTFHIP_BUF_COUNT) return nullptr;
	/* a raw pointer lets the caller write behind the library's back: no more deferral or host-side caches for this batch */
	if (lazy_flush(b) != MTFHIP_OK || ensure_df(b) != MTFHIP_OK) return nullptr;
	b->lz.enabled = false; b->lz.no_cache = true;
	if (ensure_buf(b, id) != MTFHIP_OK) return nullptr;
	if (id == MTFHIP_BUF_INIT_PTS || id == MTFHIP_BUF_INIT_Z || id == MTFHIP_BUF_INIT_HXY) b->grid_from_corners = false;   /* (the caller may lay out its own grid) */
	return b->buf[id];
}

/* ------------------------------------------------------------------ SSM */
int mtfhip_ssm_set_corners(mtfhip_batch *b, const double *corners) { return set_corners_core(b, corners, false); }

/* for_track: the upload also carries active = 1 / iteration counts = 0, i.e. it is the slab mtfhip_batch_track would
 * upload next (mtfhip_batch_track_region: one staging pass and one copy per frame instead of two) */
/* defer_grid (mtfhip_batch_track_region in front of the one-launch ICLK kernel): the host half only -- mirrors, the staged corners and NCC
 * scalars -- the kernel that follows ingests them and lays out the grid itself (RegionIngest); an affine SSM then needs no map on the
 * host at all (5.9 us of closed-form homographies per 256-patch frame), a homography one still has to know whether the grids are affine */
/* the host half of a deferred affine reset (set_corners_core): mirrors and the staged slab entries the launch did not need.  Called once
 * the loop kernel is enqueued; a no-op when nothing is pending. */
void set_corners_finish_deferred(mtfhip_batch *b) {
	const size_t Bt = (size_t)b->B;
	if (b->deferred_layout) {
		/* the patches the kernel laid out for itself, once more on the host (the same expressions: grid_pt_hd) -- under the kernel */
		b->deferred_layout = false;
		b->deferred_patches.resize(8 * Bt);
		if (mtfhip_grid_layout(&b->deferred_gdesc, b->deferred_region, nullptr, b->deferred_patches.data()) == MTFHIP_OK) {
			b->deferred_corners = b->deferred_patches.data();
			std::memcpy(reinterpret_cast<double *>(b->h_stage_a) + 17 * Bt, b->deferred_patches.data(), sizeof(double) * 8 * Bt);   /* the slab's corners */
			if (b->deferred_template_check && b->j0_is_template && b->template_corners.size() == 8 * Bt &&
				std::memcmp(b->template_corners.data(), b->deferred_patches.data(), sizeof(double) * 8 * Bt) == 0)
				b->j0_template_corners_epoch = b->corners_epoch;   /* (set_region_core's test, with the corners it did not have yet) */
		}
		b->deferred_template_check = false;
	}
	const double *corners = b->deferred_corners;
	if (!corners) return;
	b->deferred_corners = nullptr;
	double *sp = reinterpret_cast<double *>(b->h_stage_a);
	double *s_w = sp, *s_s = sp + 9 * Bt, *s_ic = sp + 25 * Bt, *s_w0 = sp + 45 * Bt;
	int *s_act = reinterpret_cast<int *>(b->h_stage_a + b->slab_dbl_bytes), *s_it = s_act + Bt;
	static const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	for (int t = 0; t < b->B; ++t) {
		TargetHost &h = b->th[t];
		std::memcpy(h.corners, corners + 8 * t, sizeof(double) * 8);
		std::memcpy(h.init_corners, corners + 8 * t, sizeof(double) * 8);
		for (int q = 0; q < 4; ++q) {
			h.init_corners_hm[3 * q] = corners[8 * t + 2 * q];
			h.init_corners_hm[3 * q + 1] = corners[8 * t + 2 * q + 1];
			h.init_corners_hm[3 * q + 2] = 1;
		}
		h.warp = m3_identity();
		std::memset(h.state, 0, sizeof(h.state));
		std::memcpy(s_w + 9 * t, ident, sizeof(ident));
		std::memset(s_s + 8 * t, 0, sizeof(double) * 8);
		std::memcpy(s_ic + 12 * t, h.init_corners_hm, sizeof(double) * 12);
		std::memcpy(s_w0 + 9 * t, ident, sizeof(ident));   /* (affine + deferred: the map is the kernel's business; the slab keeps the identity as before) */
		s_act[t] = b->deferred_for_track ? 1 : 0;
		if (b->deferred_for_track) s_it[t] = 0;
	}
}
int set_corners_core(mtfhip_batch *b, const double *corners, bool for_track, bool defer_grid, bool layout_later) {
	b->fresh_reinit = false;   /* (grid_reinit_fused sets it again behind this call) */
	FLUSH_AM(b);   /* pending calls are replayed (with the points they need); the points themselves are about to change */
	if (b) ++b->lz.epoch;
	if (b) { touch_all(b); b->lz.it_epoch = -1; TRY(ensure_df(b)); }
	if (!b || (!corners && !layout_later)) return fail(MTFHIP_ERR_INVALID_ARG, "set_corners: NULL argument");
	const bool hom = b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY;
	if (layout_later && !(defer_grid && !hom)) return fail(MTFHIP_ERR_LOGIC, "set_corners: a layout behind the launch needs the deferred affine reset");
	/* normalised grid extents: ProjectiveBase.cc:14 (unit square) ; Affine.cc:56-57 */
	double lo_x = -0.5, lo_y = -0.5, hi_x = 0.5, hi_y = 0.5;
	if (!hom) { lo_x = 1 - b->desc.resx / 2.0; lo_y = 1 - b->desc.resy / 2.0; hi_x = b->desc.resx / 2.0; hi_y = b->desc.resy / 2.0; }
	/* the staging buffer is protected by an event instead of a stream sync; one pass over the targets fills the host mirrors
	 * AND the staged slab (w | s | corners | init_corners_hm | NCC scalars | w0 | flags: the layout of fill_stage) */
	if (b->stage_a_busy) HIP_TRY(hipEventSynchronize(b->ev_a));
	const size_t Bt = (size_t)b->B;
	double *sp = reinterpret_cast<double *>(b->h_stage_a);
	double *s_w = sp, *s_s = sp + 9 * Bt, *s_cr = sp + 17 * Bt, *s_ic = sp + 25 * Bt, *s_nc = sp + 37 * Bt, *s_w0 = sp + 45 * Bt;
	int *s_act = reinterpret_cast<int *>(b->h_stage_a + b->slab_dbl_bytes), *s_it = s_act + Bt;
	static const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	int unit_z = 1;
	/* Deferred + affine (the grid tracker's frame, one launch): the kernel that follows reads only the corners and the template's
	 * NCC scalars from the staging buffer and starts every patch from the identity -- those two go in now, the host mirrors and
	 * the rest of the staged slab are written by set_corners_finish_deferred() AFTER the launch, while the device works
	 * (2-2.5 us of a 45 us frame at 256 patches). */
	if (defer_grid && !hom) {
		/* (r04 advisor: nothing is committed before the corners have been looked at -- the cheap part of the map's degeneracy test) */
		/* (layout_later: fixed-size rectangles the kernel lays out itself -- never degenerate -- reach the slab in set_corners_finish_deferred) */
		if (!layout_later) {
			for (int t = 0; t < b->B; ++t)
				if (quad_degenerate_hd(corners + 8 * t)) return fail(MTFHIP_ERR_INVALID_ARG, "set_corners: degenerate corners for target %d", t);
			std::memcpy(s_cr, corners, sizeof(double) * 8 * Bt);
		}
		for (int t = 0; t < b->B; ++t) {
			const TargetHost &h = b->th[t];
			double *q8 = s_nc + 8 * t;
			q8[0] = h.I0_mean; q8[1] = h.c; q8[2] = h.It_mean; q8[3] = h.b; q8[4] = h.f; q8[5] = h.gmean; q8[6] = q8[7] = 0;
		}
		b->deferred_corners = layout_later ? nullptr : corners; b->deferred_for_track = for_track;
		b->deferred_layout = layout_later;
		b->unit_z = 1;
		b->grid_from_corners = true;
		b->warps_dirty = false;
		b->have_corners = true;
		b->pts_stale = true;
		++b->corners_epoch;
		return MTFHIP_OK;
	}
	for (int t = 0; t < b->B; ++t) {
		M3 W0 = m3_identity();
		if (!(defer_grid && !hom)) {   /* (deferred + affine: the kernel reports degenerate corners through n_iters = -1) */
			if (!rect_to_quad(lo_x, lo_y, hi_x, hi_y, corners + 8 * t, W0)) return fail(MTFHIP_ERR_INVALID_ARG, "set_corners: degenerate corners for target %d", t);
			if (!hom || (std::fabs(W0.m[6]) < 1e-15 && std::fabs(W0.m[7]) < 1e-15)) {
				if (hom) { W0.m[6] = 0; W0.m[7] = 0; }
			} else unit_z = 0;
		}
		TargetHost &h = b->th[t];
		std::memcpy(h.corners, corners + 8 * t, sizeof(double) * 8);
		std::memcpy(h.init_corners, corners + 8 * t, sizeof(double) * 8);
		for (int q = 0; q < 4; ++q) {
			h.init_corners_hm[3 * q] = corners[8 * t + 2 * q];
			h.init_corners_hm[3 * q + 1] = corners[8 * t + 2 * q + 1];
			h.init_corners_hm[3 * q + 2] = 1;
		}
		h.warp = m3_identity();
		std::memset(h.state, 0, sizeof(h.state));
		std::memcpy(s_w + 9 * t, ident, sizeof(ident));
		std::memset(s_s + 8 * t, 0, sizeof(double) * 8);
		std::memcpy(s_cr + 8 * t, corners + 8 * t, sizeof(double) * 8);
		std::memcpy(s_ic + 12 * t, h.init_corners_hm, sizeof(double) * 12);
		double *q8 = s_nc + 8 * t;
		q8[0] = h.I0_mean; q8[1] = h.c; q8[2] = h.It_mean; q8[3] = h.b; q8[4] = h.f; q8[5] = h.gmean; q8[6] = q8[7] = 0;
		std::memcpy(s_w0 + 9 * t, W0.m, sizeof(double) * 9);
		s_act[t] = for_track ? 1 : 0;
		if (for_track) s_it[t] = 0;
	}
	b->unit_z = hom ? unit_z : 1;
	b->grid_from_corners = true;   /* (k_init_grid, or the grid kernel's region mode, lays the lattice out inside these corners) */
	if (defer_grid) {
		b->warps_dirty = false;   /* the kernel that follows starts every target from the identity */
		b->have_corners = true;
		b->pts_stale = true;
		++b->corners_epoch;
		return MTFHIP_OK;
	}
	const size_t up_bytes = for_track ? b->slab_bytes : b->slab_dbl_bytes;
	bool grid_done = false;
	if (b->h_stage_a_dev) {
		/* the kernel that lays out the grid reads the slab from the pinned staging buffer itself: no copy-engine transfer, no
		 * second launch (w0 is taken from the host copy, 45 B doubles into the slab) */
		TimedScope ts(b->ctx, "init_grid");
		grid_done = launch_init_grid_ingest(b->view(), reinterpret_cast<const double *>(b->h_stage_a_dev) + 45 * (size_t)b->B, b->desc.resx, b->desc.resy,
			lo_x, lo_y, hi_x, hi_y, hom ? 0 : 1, b->h_stage_a_dev, b->d_slab, up_bytes, for_track ? 0 : 1, b->ctx->stream);
		if (!grid_done) launch_ingest_host(b->h_stage_a_dev, b->d_slab, up_bytes, b->ctx->stream);
	} else HIP_TRY(hipMemcpyAsync(b->d_slab, b->h_stage_a, up_bytes, hipMemcpyHostToDevice, b->ctx->stream));
	b->warps_dirty = false;   /* the slab carries the (identity) warps */
	if (!grid_done) {
		TimedScope ts(b->ctx, "init_grid");
		launch_init_grid(b->view(), b->d_w0, b->desc.resx, b->desc.resy, lo_x, lo_y, hi_x, hi_y, hom ? 0 : 1, b->ctx->stream);
	}
	if (!for_track) { HIP_TRY(hipEventRecord(b->ev_a, b->ctx->stream)); b->stage_a_busy = true; }   /* (for_track: the caller waits for the loop that follows) */
	b->have_corners = true;
	b->pts_stale = grid_done && for_track;   /* k_init_grid writes the current points too, except in front of a device loop */
	++b->corners_epoch;
	return MTFHIP_OK;
}

int ensure_pts(mtfhip_batch *b) {
	if (!b->pts_stale) return MTFHIP_OK;
	b->pts_stale = false;
	TimedScope ts(b->ctx, "apply_warp");
	launch_apply_warp(b->view(), b->ctx->stream);
	return MTFHIP_OK;
}
static int apply_states(mtfhip_batch *b) {
	b->fresh_reinit = false;
	if (b->inline_warp_ok) b->warps_dirty = true;   /* uploaded by whoever needs it, or carried by the next fused launch */
	else TRY(push_warps(b));
	b->pts_stale = true;   /* refreshed by the next entry point that may read them (lazy_flush) */
	return MTFHIP_OK;
}

int mtfhip_ssm_set_state(mtfhip_batch *b, const double *states) {
	FLUSH_AM(b);   /* pending calls are replayed (with the points they need); the points themselves are about to change */
	if (b) ++b->lz.epoch;
	if (!b || !states) return fail(MTFHIP_ERR_INVALID_ARG, "set_state: NULL argument");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "set_state before set_corners");
	for (int t = 0; t < b->B; ++t) {
		TargetHost &h = b->th[t];
		std::memset(h.state, 0, sizeof(h.state));
		std::memcpy(h.state, states + (size_t)t * b->S, sizeof(double) * b->S);
		h.warp = warp_from_state(b->desc.ssm, h.state);
		update_corners(b, t);
	}
	return apply_states(b);
}

int mtfhip_ssm_compositional_update(mtfhip_batch *b, const double *dps) {
	FLUSH_AM(b);   /* pending calls are replayed (with the points they need); the points themselves are about to change */
	if (b) ++b->lz.epoch;
	if (!b || !dps) return fail(MTFHIP_ERR_INVALID_ARG, "compositional_update: NULL argument");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "compositional_update before set_corners");
	for (int t = 0; t < b->B; ++t) {
		TargetHost &h = b->th[t];
		double dp[8] = {0};
		std::memcpy(dp, dps + (size_t)t * b->S, sizeof(double) * b->S);
		M3 upd = warp_from_state(b->desc.ssm, dp);
		h.warp = m3_mul(h.warp, upd);
		if (b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY) {
			double s = h.warp.m[8];
			for (int i = 0; i < 9; ++i) h.warp.m[i] /= s;
		}
		state_from_warp(b->desc.ssm, h.state, h.warp);
		update_corners(b, t);
	}
	return apply_states(b);
}

int mtfhip_ssm_invert_state(mtfhip_batch *b, const double *states, double *inv_states) {
	if (!b || !states || !inv_states) return fail(MTFHIP_ERR_INVALID_ARG, "invert_state: NULL argument");
	for (int t = 0; t < b->B; ++t) {
		double p[8] = {0}, q[8];
		std::memcpy(p, states + (size_t)t * b->S, sizeof(double) * b->S);
		M3 Wi = m3_inverse(warp_from_state(b->desc.ssm, p));
		double s = Wi.m[8];
		for (int i = 0; i < 9; ++i) Wi.m[i] /= s;
		state_from_warp(b->desc.ssm, q, Wi);
		std::memcpy(inv_states + (size_t)t * b->S, q, sizeof(double) * b->S);
	}
	return MTFHIP_OK;
}

int do_update_grad_pts(mtfhip_batch *b, double grad_eps) {
	TRY(ensure_buf(b, MTFHIP_BUF_GRAD_PTS));
	TimedScope ts(b->ctx, "grad_pts");
	launch_grad_pts(b->view(), grad_eps, b->ctx->stream);
	return MTFHIP_OK;
}
int mtfhip_ssm_update_grad_pts(mtfhip_batch *b, double grad_eps) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "update_grad_pts: NULL batch");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "update_grad_pts before set_corners");
	if (b->lz.enabled && grad_eps == b->desc.grad_eps) {
		if (b->lz.gp || b->lz.pg) FLUSH(b);
		b->lz.gp = ++b->lz.seq;
		return MTFHIP_OK;
	}
	FLUSH(b);
	return do_update_grad_pts(b, grad_eps);
}

int do_cmpt_pix_jacobian(mtfhip_batch *b, int variant, int grad_buf, int dst_buf) {
	TRY(ensure_buf(b, dst_buf));
	TimedScope ts(b->ctx, "pix_jacobian");
	launch_pix_jacobian(b->view(), variant, b->buf[grad_buf], b->buf[dst_buf], b->ctx->stream);
	touch(b, dst_buf);
	if (dst_buf == MTFHIP_BUF_JT) b->jt_valid = true;
	if (dst_buf == MTFHIP_BUF_J0) b->j0_is_template = false;
	return MTFHIP_OK;
}
int mtfhip_ssm_cmpt_pix_jacobian(mtfhip_batch *b, int variant, int grad_buf, int dst_buf) {
	if (!b) return fail(MTFHIP_ERR_INVALID_ARG, "cmpt_pix_jacobian: NULL batch");
	if (variant < MTFHIP_JAC_INIT || variant > MTFHIP_JAC_APPROX) return fail(MTFHIP_ERR_INVALID_ARG, "unknown Jacobian variant %d", variant);
	if (grad_buf != MTFHIP_BUF_DI0_DX && grad_buf != MTFHIP_BUF_DIT_DX) return fail(MTFHIP_ERR_INVALID_ARG, "grad_buf must be DI0_DX or DIT_DX");
	if (!j_buf_ok(dst_buf)) return fail(MTFHIP_ERR_INVALID_ARG, "dst_buf must be J0, JT or JM");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "cmpt_pix_jacobian before set_corners");
	if (b->lz.enabled && grad_buf == MTFHIP_BUF_DIT_DX && dst_buf == MTFHIP_BUF_JT &&
		(variant == MTFHIP_JAC_WARPED || variant == MTFHIP_JAC_INIT)) {
		if (b->lz.pj || b->lz.jm) FLUSH(b);
		TRY(ensure_buf(b, dst_buf));
		b->lz.pj = ++b->lz.seq; b->lz.pj_variant = variant;
		b->jt_valid = true;
		return MTFHIP_OK;
	}
	FLUSH(b);
	return do_cmpt_pix_jacobian(b, variant, grad_buf, dst_buf);
}

int mtfhip_ssm_get_corners(mtfhip_batch *b, double *corners) {
	if (!b || !corners) return fail(MTFHIP_ERR_INVALID_ARG, "get_corners: NULL argument");
	for (int t = 0; t < b->B; ++t) std::memcpy(corners + 8 * t, b->th[t].corners, sizeof(double) * 8);
	return MTFHIP_OK;
}
/* ProjectiveBase::estimateStateSigma (SSM/src/ProjectiveBase.cc:201-213): state_sigma[k] = pix_sigma / mean over the points of the
 * norm of column k of the 2 x S point Jacobian dw/dp (Homography::getCurrPixGrad Homography.cc:143-155, Affine::getInitPixGrad
 * Affine.cc:152-158) -- how nt::PF turns pix_sigma into sampler sigmas (PF.cc:142-149).  Host arithmetic on the points read back. */
int mtfhip_ssm_estimate_state_sigma(mtfhip_batch *b, double pix_sigma, double *state_sigma) {
	if (!b || !state_sigma) return fail(MTFHIP_ERR_INVALID_ARG, "estimate_state_sigma: NULL argument");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "estimate_state_sigma before set_corners");
	FLUSH(b);
	TRY(ensure_pts(b));
	const size_t NP = (size_t)b->NP;
	std::vector<double> ip(2 * NP * b->B), cp(2 * NP * b->B), cz(NP * b->B);
	hipStream_t st = b->ctx->stream;
	HIP_TRY(hipMemcpyAsync(ip.data(), b->buf[MTFHIP_BUF_INIT_PTS], sizeof(double) * ip.size(), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(cp.data(), b->buf[MTFHIP_BUF_CURR_PTS], sizeof(double) * cp.size(), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(cz.data(), b->buf[MTFHIP_BUF_CURR_Z], sizeof(double) * cz.size(), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	const bool hom = b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY;
	for (int t = 0; t < b->B; ++t) {
		double mean[8] = {0, 0, 0, 0, 0, 0, 0, 0};
		for (size_t i = 0; i < NP; ++i) {
			const double x = ip[2 * (t * NP + i)], y = ip[2 * (t * NP + i) + 1];
			double c0[8], c1[8];   /* the two rows of dw/dp */
			if (hom) {
				const double cx = cp[2 * (t * NP + i)], cy = cp[2 * (t * NP + i) + 1], inv_d = 1.0 / cz[t * NP + i];
				const double r0[8] = {x, y, 1, 0, 0, 0, -x * cx, -y * cx}, r1[8] = {0, 0, 0, x, y, 1, -x * cy, -y * cy};
				for (int k = 0; k < 8; ++k) { c0[k] = r0[k] * inv_d; c1[k] = r1[k] * inv_d; }
			} else {
				const double r0[8] = {1, 0, x, y, 0, 0, 0, 0}, r1[8] = {0, 1, 0, 0, x, y, 0, 0};
				for (int k = 0; k < 8; ++k) { c0[k] = r0[k]; c1[k] = r1[k]; }
			}
			for (int k = 0; k < b->S; ++k) mean[k] += std::sqrt(c0[k] * c0[k] + c1[k] * c1[k]);
		}
		for (int k = 0; k < b->S; ++k) state_sigma[(size_t)t * b->S + k] = pix_sigma / (mean[k] / (double)NP);
	}
	return MTFHIP_OK;
}
int mtfhip_ssm_get_init_corners(mtfhip_batch *b, double *corners) {
	if (!b || !corners) return fail(MTFHIP_ERR_INVALID_ARG, "get_init_corners: NULL argument");
	for (int t = 0; t < b->B; ++t) std::memcpy(corners + 8 * t, b->th[t].init_corners, sizeof(double) * 8);
	return MTFHIP_OK;
}
int mtfhip_ssm_get_state(mtfhip_batch *b, double *states) {
	if (!b || !states) return fail(MTFHIP_ERR_INVALID_ARG, "get_state: NULL argument");
	for (int t = 0; t < b->B; ++t) std::memcpy(states + (size_t)t * b->S, b->th[t].state, sizeof(double) * b->S);
	return MTFHIP_OK;
}
int mtfhip_ssm_get_warp(mtfhip_batch *b, double *warps) {
	if (!b || !warps) return fail(MTFHIP_ERR_INVALID_ARG, "get_warp: NULL argument");
	for (int t = 0; t < b->B; ++t) std::memcpy(warps + 9 * t, b->th[t].warp.m, sizeof(double) * 9);
	return MTFHIP_OK;
}
int mtfhip_ssm_apply_warp_to_corners(mtfhip_batch *b, const double *in_corners, const double *states, double *out_corners) {
	if (!b || !in_corners || !states || !out_corners) return fail(MTFHIP_ERR_INVALID_ARG, "apply_warp_to_corners: NULL argument");
	for (int t = 0; t < b->B; ++t) {
		double p[8] = {0};
		std::memcpy(p, states + (size_t)t * b->S, sizeof(double) * b->S);
		M3 W = warp_from_state(b->desc.ssm, p);
		for (int q = 0; q < 4; ++q) {
			double x = in_corners[8 * t + 2 * q], y = in_corners[8 * t + 2 * q + 1];
			double nx = W.m[0] * x + W.m[1] * y + W.m[2], ny = W.m[3] * x + W.m[4] * y + W.m[5];
			if (b->desc.ssm == MTFHIP_SSM_HOMOGRAPHY) {
				double d = W.m[6] * x + W.m[7] * y + W.m[8];
				nx = nx / d; ny = ny / d;
			}
			out_corners[8 * t + 2 * q] = nx; out_corners[8 * t + 2 * q + 1] = ny;
		}
	}
	return MTFHIP_OK;
}

/* ---- SSM functions that are 3 x 3 algebra on the host: no device, no context (callable on a machine without a GPU) ---- */
static int ssm_kind_ok(int ssm, const char *fn) {
	if (ssm != MTFHIP_SSM_HOMOGRAPHY && ssm != MTFHIP_SSM_AFFINE) return fail(MTFHIP_ERR_INVALID_ARG, "%s: unknown state space model %d", fn, ssm);
	return MTFHIP_OK;
}
int mtfhip_ssm_identity_warp(int ssm, double *state) {
	TRY(ssm_kind_ok(ssm, "identity_warp"));
	if (!state) return fail(MTFHIP_ERR_INVALID_ARG, "identity_warp: NULL argument");
	std::memset(state, 0, sizeof(double) * (ssm == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6));
	return MTFHIP_OK;
}
int mtfhip_ssm_compose_warps(int ssm, const double *state_1, const double *state_2, double *composed) {
	TRY(ssm_kind_ok(ssm, "compose_warps"));
	if (!state_1 || !state_2 || !composed) return fail(MTFHIP_ERR_INVALID_ARG, "compose_warps: NULL argument");
	const int S = ssm == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6;
	double p1[8] = {0}, p2[8] = {0}, out[8] = {0};
	std::memcpy(p1, state_1, sizeof(double) * S); std::memcpy(p2, state_2, sizeof(double) * S);
	/* warp_2 * warp_1, read back entry by entry: the reference does not renormalise by (2, 2) here (ProjectiveBase.cc:324-331) */
	state_from_warp(ssm, out, m3_mul(warp_from_state(ssm, p2), warp_from_state(ssm, p1)));
	std::memcpy(composed, out, sizeof(double) * S);
	return MTFHIP_OK;
}
int mtfhip_ssm_estimate_warp_from_corners(int ssm, const double *in_corners, const double *out_corners, double *state_update) {
	TRY(ssm_kind_ok(ssm, "estimate_warp_from_corners"));
	if (!in_corners || !out_corners || !state_update) return fail(MTFHIP_ERR_INVALID_ARG, "estimate_warp_from_corners: NULL argument");
	double out[8] = {0};
	if (ssm == MTFHIP_SSM_HOMOGRAPHY) {
		/* the 4-point DLT (warpUtils.cc:171-224) is the unique homography through the four pairs: unit square -> out composed
		 * with the inverse of unit square -> in, scaled to (2, 2) = 1 (Homography.cc:877-883) */
		M3 Qi, Qo;
		if (!rect_to_quad(0, 0, 1, 1, in_corners, Qi) || !rect_to_quad(0, 0, 1, 1, out_corners, Qo))
			return fail(MTFHIP_ERR_INVALID_ARG, "estimate_warp_from_corners: degenerate corners");
		M3 H = m3_mul(Qo, m3_inverse(Qi));
		if (H.m[8] == 0 || !std::isfinite(H.m[8])) return fail(MTFHIP_ERR_INVALID_ARG, "estimate_warp_from_corners: degenerate corners");
		for (int i = 0; i < 9; ++i) H.m[i] /= H.m[8];
		H.m[8] = 1;
		state_from_warp(ssm, out, H);
	} else {
		/* least-squares affine map of the four pairs (computeAffineDLT, warpUtils.cc:276-342: pseudo-inverse of the 8 x 6
		 * system, whose x and y halves share the 3 x 3 normal matrix) */
		M3 G{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
		double bx[3] = {0, 0, 0}, by[3] = {0, 0, 0};
		for (int q = 0; q < 4; ++q) {
			const double r[3] = {in_corners[2 * q], in_corners[2 * q + 1], 1.0};
			for (int i = 0; i < 3; ++i) {
				for (int j = 0; j < 3; ++j) G.m[3 * i + j] += r[i] * r[j];
				bx[i] += r[i] * out_corners[2 * q]; by[i] += r[i] * out_corners[2 * q + 1];
			}
		}
		const double det = G.m[0] * (G.m[4] * G.m[8] - G.m[5] * G.m[7]) - G.m[1] * (G.m[3] * G.m[8] - G.m[5] * G.m[6]) +
			G.m[2] * (G.m[3] * G.m[7] - G.m[4] * G.m[6]);
		if (det == 0 || !std::isfinite(det)) return fail(MTFHIP_ERR_INVALID_ARG, "estimate_warp_from_corners: degenerate corners");
		const M3 Gi = m3_inverse(G);
		M3 W = m3_identity();
		for (int i = 0; i < 3; ++i) {
			W.m[i] = Gi.m[3 * i] * bx[0] + Gi.m[3 * i + 1] * bx[1] + Gi.m[3 * i + 2] * bx[2];
			W.m[3 + i] = Gi.m[3 * i] * by[0] + Gi.m[3 * i + 1] * by[1] + Gi.m[3 * i + 2] * by[2];
		}
		state_from_warp(ssm, out, W);
	}
	std::memcpy(state_update, out, sizeof(double) * (ssm == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6));
	return MTFHIP_OK;
}
int mtfhip_ssm_apply_warp_to_pts(int ssm, const double *in_pts, int n_pts, const double *state, double *out_pts) {
	TRY(ssm_kind_ok(ssm, "apply_warp_to_pts"));
	if (!in_pts || !state || !out_pts || n_pts < 0) return fail(MTFHIP_ERR_INVALID_ARG, "apply_warp_to_pts: invalid argument");
	double p[8] = {0};
	std::memcpy(p, state, sizeof(double) * (ssm == MTFHIP_SSM_HOMOGRAPHY ? 8 : 6));
	const M3 W = warp_from_state(ssm, p);
	for (int i = 0; i < n_pts; ++i) {
		const double x = in_pts[2 * i], y = in_pts[2 * i + 1];
		double nx = W.m[0] * x + W.m[1] * y + W.m[2], ny = W.m[3] * x + W.m[4] * y + W.m[5];
		if (ssm == MTFHIP_SSM_HOMOGRAPHY) { const double d = W.m[6] * x + W.m[7] * y + W.m[8]; nx = nx / d; ny = ny / d; }
		out_pts[2 * i] = nx; out_pts[2 * i + 1] = ny;
	}
	return MTFHIP_OK;
}
/* ------------------------------------------------------------------ GridTracker's patch layout (host arithmetic only) */
static int grid_desc_ok(const mtfhip_grid_desc *g, const char *fn) {
	if (!g) return fail(MTFHIP_ERR_INVALID_ARG, "%s: NULL grid description", fn);
	if (g->grid_size_x <= 0 || g->grid_size_y <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "%s: grid_size must be positive", fn);
	if (g->patch_size_x <= 0 || g->patch_size_y <= 0) return fail(MTFHIP_ERR_INVALID_ARG, "%s: patch_size must be positive", fn);
	return MTFHIP_OK;
}
/* (Eigen's LinSpaced as utils::getNormUnitSquarePts uses it, warpUtils.cc:15-34, is lin_spaced_hd in mtfhip_internal.h) */
/* GridTrackerParams::updateRes SM/src/GridTracker.cc:86-94 */
int mtfhip_grid_res(const mtfhip_grid_desc *g, int *resx, int *resy) {
	TRY(grid_desc_ok(g, "grid_res"));
	if (!resx || !resy) return fail(MTFHIP_ERR_INVALID_ARG, "grid_res: NULL argument");
	const int extra = (g->dyn_patch_size || g->patch_centroid_inside) ? 1 : 0;
	*resx = g->grid_size_x + extra; *resy = g->grid_size_y + extra;
	return MTFHIP_OK;
}
/* GridTracker::resetTrackers' geometry SM/src/GridTracker.cc:345-380 over ssm.getPts() of the grid SSM after setCorners(region)
 * (ProjectiveBase::setCorners -> getPtsFromCorners ProjectiveBase.cc:20-36: the resx x resy grid of the unit square through the
 * 4-corner homography; Affine::setCorners Affine.cc:75-79 takes the same route with normalized_init = 0 -- its pixel-sized
 * normalised square instead of the unit one is the same uniform grid).  The grid point is evaluated, not interpolated between
 * the region's corners: for a region that is not a parallelogram the two differ. */
int mtfhip_grid_layout(const mtfhip_grid_desc *g, const double *region, double *grid_pts, double *patch_corners) {
	int resx, resy;
	TRY(mtfhip_grid_res(g, &resx, &resy));
	if (!region || !patch_corners) return fail(MTFHIP_ERR_INVALID_ARG, "grid_layout: NULL argument");
	M3 W;
	if (!rect_to_quad(-0.5, -0.5, 0.5, 0.5, region, W)) return fail(MTFHIP_ERR_INVALID_ARG, "grid_layout: degenerate region corners");
	/* every grid point once (grid_pt_hd: the expression k_iclk_track's own layout evaluates per patch, mtfhip_internal.h), then the patches
	 * from them as grid_patch_corners_hd assembles them */
	static thread_local std::vector<double> pts;
	pts.resize(2 * (size_t)resx * resy);
	for (int i = 0; i < resx * resy; ++i) grid_pt_hd(W.m, resx, resy, i, &pts[2 * (size_t)i], &pts[2 * (size_t)i + 1]);
	if (grid_pts) std::memcpy(grid_pts, pts.data(), sizeof(double) * pts.size());
	const bool surround = g->dyn_patch_size || g->patch_centroid_inside;
	const int sub_x = g->grid_size_x + 1;   /* _linear_idx(idy, idx) = idy * (grid_size_x + 1) + idx, :139-146 */
	const double half_x = g->patch_size_x / 2.0, half_y = g->patch_size_y / 2.0;   /* centrod_dist_x / _y :156-157 */
	for (int k = 0; k < g->grid_size_x * g->grid_size_y; ++k) {
		const int row = k / g->grid_size_x, col = k % g->grid_size_x;   /* :354-355 */
		double *pc = patch_corners + 8 * (size_t)k;
		if (surround) {   /* :357-367 TL, TR, BR, BL of the cell (without the extra row / column the reads have no meaning: the reference overwrites them) */
			const int id[4] = {row * sub_x + col, row * sub_x + col + 1, (row + 1) * sub_x + col + 1, (row + 1) * sub_x + col};
			for (int q = 0; q < 4; ++q) { pc[2 * q] = pts[2 * (size_t)id[q]]; pc[2 * q + 1] = pts[2 * (size_t)id[q] + 1]; }
		}
		if (!g->dyn_patch_size) {   /* :369-380 */
			double cx = pts[2 * (size_t)k], cy = pts[2 * (size_t)k + 1];   /* ssm.getPts().col(tracker_id) */
			if (g->patch_centroid_inside) {   /* utils::getCentroid miscUtils.h:481-487 */
				cx = (pc[0] + pc[2] + pc[4] + pc[6]) / 4.0;
				cy = (pc[1] + pc[3] + pc[5] + pc[7]) / 4.0;
			}
			const double min_x = cx - half_x, min_y = cy - half_y;   /* utils::Corners(cv::Rect_<double>) miscUtils.h:42-52 */
			const double max_x = min_x + g->patch_size_x, max_y = min_y + g->patch_size_y;
			pc[0] = pc[6] = min_x; pc[2] = pc[4] = max_x;
			pc[1] = pc[3] = min_y; pc[5] = pc[7] = max_y;
		}
	}
	return MTFHIP_OK;
}

/* ProjectiveBase::additiveUpdate SSM/src/ProjectiveBase.cc:51-55: curr_state += update; setState(curr_state) */
int mtfhip_ssm_additive_update(mtfhip_batch *b, const double *state_updates) {
	if (!b || !state_updates) return fail(MTFHIP_ERR_INVALID_ARG, "additive_update: NULL argument");
	if (!b->have_corners) return fail(MTFHIP_ERR_LOGIC, "additive_update before set_corners");
	std::vector<double> st((size_t)b->B * b->S);
	for (int t = 0; t < b->B; ++t)
		for (int s = 0; s < b->S; ++s) st[(size_t)t * b->S + s] = b->th[t].state[s] + state_updates[(size_t)t * b->S + s];
	return mtfhip_ssm_set_state(b, st.data());
}

/* ------------------------------------------------------------------ timing */
int mtfhip_timing_enable(mtfhip_ctx *c, int on) {
	if (!c) return fail(MTFHIP_ERR_INVALID_ARG, "timing_enable: NULL ctx");
	c->timing = on != 0;
	c->timing_stride = on > 1 ? on : 1;
	return MTFHIP_OK;
}
static void drain(mtfhip_ctx *c) {
	(void)hipStreamSynchronize(c->stream);
	for (hipStream_t s : c->extra_streams) if (s) (void)hipStreamSynchronize(s);
	for (auto &kv : c->timers) {
		for (auto &p : kv.second.pending) {
			float ms = 0, t0 = 0;
			if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
				kv.second.total_ms += ms; kv.second.n += 1;
				if (c->ev_ref && hipEventElapsedTime(&t0, c->ev_ref, p.first) == hipSuccess) kv.second.spans.emplace_back((double)t0, (double)t0 + ms);
				else (void)hipGetLastError();
			}
			c->free_events.push_back(p.first);
			c->free_events.push_back(p.second);
		}
		kv.second.pending.clear();
	}
}
int mtfhip_timing_reset(mtfhip_ctx *c) {
	if (!c) return fail(MTFHIP_ERR_INVALID_ARG, "timing_reset: NULL ctx");
	drain(c);
	for (auto &kv : c->timers) { kv.second.total_ms = 0; kv.second.n = 0; kv.second.launches = 0; kv.second.spans.clear(); }
	if (!c->ev_ref && hipEventCreate(&c->ev_ref) != hipSuccess) { (void)hipGetLastError(); c->ev_ref = nullptr; }
	if (c->ev_ref) { (void)hipEventRecord(c->ev_ref, c->stream); (void)hipStreamSynchronize(c->stream); }
	return MTFHIP_OK;
}
/* time during which at least one launch of the family was executing (the union of its launches' intervals) and their number:
 * equal to n x the average of mtfhip_timing_get while launches follow each other on one queue, smaller when they overlap */
int mtfhip_timing_get_busy(mtfhip_ctx *c, const char *family, double *busy_ms, int *n_launches) {
	if (!c || !family) return fail(MTFHIP_ERR_INVALID_ARG, "timing_get_busy: NULL argument");
	drain(c);
	auto it = c->timers.find(family);
	double busy = 0; int n = 0;
	if (it != c->timers.end()) {
		std::vector<std::pair<double, double>> sp = it->second.spans;
		std::sort(sp.begin(), sp.end());
		double cur_a = 0, cur_b = -1;
		for (const auto &iv : sp) {
			if (cur_b < cur_a || iv.first > cur_b) { if (cur_b >= cur_a) busy += cur_b - cur_a; cur_a = iv.first; cur_b = iv.second; }
			else if (iv.second > cur_b) cur_b = iv.second;
		}
		if (cur_b >= cur_a) busy += cur_b - cur_a;
		n = (int)sp.size();
	}
	if (busy_ms) *busy_ms = busy;
	if (n_launches) *n_launches = n;
	return MTFHIP_OK;
}
int mtfhip_timing_get(mtfhip_ctx *c, const char *family, double *avg_ms, int *n_launches) {
	if (!c || !family) return fail(MTFHIP_ERR_INVALID_ARG, "timing_get: NULL argument");
	drain(c);
	auto it = c->timers.find(family);
	double avg = 0; int n = 0;
	if (it != c->timers.end() && it->second.n > 0) { avg = it->second.total_ms / it->second.n; n = it->second.n; }
	if (avg_ms) *avg_ms = avg;
	if (n_launches) *n_launches = n;
	return MTFHIP_OK;
}
int mtfhip_batch_inline_warp(const mtfhip_batch *b) { return b && b->inline_warp_ok ? 1 : 0; }

} /* extern "C" */
