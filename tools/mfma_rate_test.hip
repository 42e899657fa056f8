// Issue rate of the FP64 MFMA forms on gfx950: cycles per instruction per SIMD (s_memtime), 8 independent accumulators.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate_test tools/mfma_rate_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, long long *cyc, int iters) {
	double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
	double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
	d4 e0 = {0, 0, 0, 0}, e1 = e0, e2 = e0, e3 = e0;
	double f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	long long t0 = __builtin_readcyclecounter();
	for (int i = 0; i < iters; ++i) {
		if (KIND == 0) {
			c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
			c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
			c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0); c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
			c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0); c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
		} else if (KIND == 1) {
			e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e0, 0, 0, 0); e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e1, 0, 0, 0);
			e2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e2, 0, 0, 0); e3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, e3, 0, 0, 0);
		} else {
#pragma unroll
			for (int j = 0; j < 8; ++j) f[j] = fma(a, b, f[j]);
		}
	}
	long long t1 = __builtin_readcyclecounter();
	out[blockIdx.x * 256 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + e0[0] + e1[1] + e2[2] + e3[3] + f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7];
	if (threadIdx.x == 0 && blockIdx.x == 0) cyc[KIND] = t1 - t0;
}
int main() {
	double *out; long long *cyc;
	hipMalloc(&out, 8 * 1024 * 256 * 8); hipMalloc(&cyc, 64);
	const int iters = 4096;
	const char *names[3] = {"v_mfma_f64_4x4x4_4b (8 per iteration)", "v_mfma_f64_16x16x4 (4 per iteration)", "v_fma_f64 (8 per iteration)"};
	const int per[3] = {8, 4, 8};
	for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu)
		for (int kind = 0; kind < 3; ++kind) {
			hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
			dim3 g(256 * wg_per_cu);
			if (kind == 0) hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, out, cyc, 16);
			hipDeviceSynchronize();
			hipEventRecord(a);
			if (kind == 0) hipLaunchKernelGGL(k<0>, g, dim3(256), 0, 0, out, cyc, iters);
			else if (kind == 1) hipLaunchKernelGGL(k<1>, g, dim3(256), 0, 0, out, cyc, iters);
			else hipLaunchKernelGGL(k<2>, g, dim3(256), 0, 0, out, cyc, iters);
			hipEventRecord(b); hipEventSynchronize(b);
			float ms; hipEventElapsedTime(&ms, a, b);
			long long h[3]; hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
			const double n = (double)iters * per[kind];
			printf("%-42s waves/SIMD %d: %.1f ns-clock cycles (100 MHz s_memtime ticks x?) per instr by counter %.2f, wall %.3f ms -> %.1f cycles/instr/SIMD at 2.4 GHz\n",
				names[kind], wg_per_cu, 0.0, (double)h[kind] / n, ms, ms * 1e-3 * 2.4e9 / (n * wg_per_cu));
		}
	return 0;
}
