// Issue rate of FP64 vector FMA and the FP64 matrix forms on gfx950: wall time per instruction per SIMD for 8 / 16 independent accumulators
// and 1 / 2 / 4 waves per SIMD (r04: what "fraction of the FP64 peak" has to be read against).
// build: hipcc --offload-arch=gfx950 -O3 -o fp64_rate_test tools/fp64_rate_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND, int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters) {
	double a = threadIdx.x * 1e-3 + 1.0, b = 1.0 - threadIdx.x * 1e-4;
	double f[NACC];
#pragma unroll
	for (int j = 0; j < NACC; ++j) f[j] = j;
	for (int i = 0; i < iters; ++i) {
#pragma unroll
		for (int j = 0; j < NACC; ++j) {
			if (KIND == 0) f[j] = fma(a, b, f[j]);
			else if (KIND == 1) f[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, f[j], 0, 0, 0);
			else f[j] = f[j] * a;   // v_mul_f64
		}
	}
	double s = 0;
#pragma unroll
	for (int j = 0; j < NACC; ++j) s += f[j];
	out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND, int NACC>
static void run(const char *name, double *out, int waves_per_simd, double ghz) {
	const int iters = 4096;
	dim3 g(256 * waves_per_simd);   // 256 CUs x (4 waves per workgroup = 1 per SIMD) x waves_per_simd
	hipLaunchKernelGGL((k<KIND, NACC>), g, dim3(256), 0, 0, out, 16);
	hipDeviceSynchronize();
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	hipEventRecord(a);
	hipLaunchKernelGGL((k<KIND, NACC>), g, dim3(256), 0, 0, out, iters);
	hipEventRecord(b); hipEventSynchronize(b);
	float ms; hipEventElapsedTime(&ms, a, b);
	const double n = (double)iters * NACC * waves_per_simd;   // instructions per SIMD
	const double cyc = ms * 1e-3 * ghz * 1e9 / n;
	const double flop_per_instr = KIND == 1 ? 512.0 : (KIND == 0 ? 128.0 : 64.0);
	printf("%-22s %2d accumulators, %d wave(s)/SIMD: %.3f ms, %.2f cycles/instr/SIMD at %.1f GHz -> %.1f TFLOP/s device-wide\n", name, NACC, waves_per_simd, ms, cyc,
		ghz, flop_per_instr * n * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
	double *out; hipMalloc(&out, 8 * 2048 * 256);
	const double ghz = 2.4;
	for (int w : {1, 2, 4}) {
		if (w == 1) { run<0, 8>("v_fma_f64", out, 1, ghz); run<0, 16>("v_fma_f64", out, 1, ghz); run<2, 16>("v_mul_f64", out, 1, ghz); run<1, 8>("v_mfma_f64_4x4x4_4b", out, 1, ghz); run<1, 16>("v_mfma_f64_4x4x4_4b", out, 1, ghz); }
		if (w == 2) { run<0, 8>("v_fma_f64", out, 2, ghz); run<0, 16>("v_fma_f64", out, 2, ghz); run<2, 16>("v_mul_f64", out, 2, ghz); run<1, 8>("v_mfma_f64_4x4x4_4b", out, 2, ghz); run<1, 16>("v_mfma_f64_4x4x4_4b", out, 2, ghz); }
		if (w == 4) { run<0, 8>("v_fma_f64", out, 4, ghz); run<0, 16>("v_fma_f64", out, 4, ghz); run<2, 16>("v_mul_f64", out, 4, ghz); run<1, 8>("v_mfma_f64_4x4x4_4b", out, 4, ghz); run<1, 16>("v_mfma_f64_4x4x4_4b", out, 4, ghz); }
	}
	return 0;
}
