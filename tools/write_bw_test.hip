// Pure streaming-write bandwidth of one MI355X (r06: what the NN dataset's rows kernel can be priced against): a kernel that only stores --
// 8-byte / 16-byte per lane, plain / non-temporal, over 0.2 GB (the 10 000-sample dataset) and 2 GB (100 000 samples) -- and the same with
// a read stream of equal size beside it.   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/write_bw tools/write_bw_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));
template <int WIDTH, bool NT, bool READ>
__global__ __launch_bounds__(256) void k_write(double *dst, const double *src, size_t n /* doubles */) {
	const size_t stride = (size_t)gridDim.x * 256 * (WIDTH / 8);
	for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * (WIDTH / 8); i < n; i += stride) {
		double v = (double)i;
		if (READ) v += WIDTH == 16 ? src[i] + src[i + 1] : src[i];
		if constexpr (WIDTH == 16) {
			d2 p; p.x = v; p.y = v + 1;
			if (NT) __builtin_nontemporal_store(p, reinterpret_cast<d2 *>(dst + i)); else *reinterpret_cast<d2 *>(dst + i) = p;
		} else {
			if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
		}
	}
}
template <int WIDTH, bool NT, bool READ>
static int run(const char *name, double *dst, const double *src, size_t n, int blocks) {
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_write<WIDTH, NT, READ>), dim3(blocks), dim3(256), 0, 0, dst, src, n);
	CK(hipEventRecord(a));
	const int reps = 20;
	for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_write<WIDTH, NT, READ>), dim3(blocks), dim3(256), 0, 0, dst, src, n);
	CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
	float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
	const double us = ms * 1e3 / reps, gb = n * 8.0 / 1e9;
	printf("%-44s %6.2f GB written%s  %8.1f us  %6.2f TB/s written\n", name, gb, READ ? " + as much read" : "", us, gb / us * 1e3);   /* GB / us = 1000 TB/s */
	return 0;
}
int main() {
	for (size_t bytes : {(size_t)200000000, (size_t)2000000000}) {
		const size_t n = bytes / 8;
		double *dst, *src;
		CK(hipMalloc(&dst, bytes + 64)); CK(hipMalloc(&src, bytes + 64));
		CK(hipMemset(src, 0, bytes));
		for (int blocks : {2048, 8192}) {
			printf("-- %zu bytes, %d workgroups\n", bytes, blocks);
			if (run<8, false, false>("8 B per lane, plain", dst, src, n, blocks)) return 1;
			if (run<8, true, false>("8 B per lane, non-temporal", dst, src, n, blocks)) return 1;
			if (run<16, false, false>("16 B per lane, plain", dst, src, n, blocks)) return 1;
			if (run<16, true, false>("16 B per lane, non-temporal", dst, src, n, blocks)) return 1;
			if (run<16, true, true>("16 B per lane, non-temporal, + read stream", dst, src, n, blocks)) return 1;
		}
		CK(hipFree(dst)); CK(hipFree(src));
	}
	return 0;
}
