#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *a_in, const double *b_in, double *d_out) {
	const int l = threadIdx.x;
	d4 c = {0, 0, 0, 0};
	c = __builtin_amdgcn_mfma_f64_16x16x4f64(a_in[l], b_in[l], c, 0, 0, 0);
	for (int v = 0; v < 4; ++v) d_out[l * 4 + v] = c[v];
}
int main() {
	double ha[64], hb[64], hd[256], *dA, *dB, *dD;
	for (int l = 0; l < 64; ++l) { ha[l] = 1 + std::sin(l * 1.7); hb[l] = 2 + std::cos(l * 0.9); }
	(void)hipMalloc(&dA, sizeof ha); (void)hipMalloc(&dB, sizeof hb); (void)hipMalloc(&dD, sizeof hd);
	(void)hipMemcpy(dA, ha, sizeof ha, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hb, sizeof hb, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	(void)hipMemcpy(hd, dD, sizeof hd, hipMemcpyDeviceToHost);
	for (int HA = 0; HA < 2; ++HA) for (int HB = 0; HB < 2; ++HB) for (int HD = 0; HD < 4; ++HD) {
		double A[16][4], B[4][16];
		for (int l = 0; l < 64; ++l) {
			if (HA == 0) A[l % 16][l / 16] = ha[l]; else A[l / 4][l % 4] = ha[l];
			if (HB == 0) B[l / 16][l % 16] = hb[l]; else B[l % 4][l / 4] = hb[l];
		}
		double err = 0;
		for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
			int i, j;
			if (HD == 0) { i = 4 * (l / 16) + v; j = l % 16; }
			else if (HD == 1) { i = l / 16 + 4 * v; j = l % 16; }
			else if (HD == 2) { i = l % 16; j = 4 * (l / 16) + v; }
			else { i = l % 16; j = l / 16 + 4 * v; }
			double r = 0; for (int kk = 0; kk < 4; ++kk) r += A[i][kk] * B[kk][j];
			err = std::fmax(err, std::fabs(r - hd[l * 4 + v]));
		}
		printf("HA=%d HB=%d HD=%d err=%g %s\n", HA, HB, HD, err, err < 1e-9 ? "<== MATCH" : "");
	}
	return 0;
}
