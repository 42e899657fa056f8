// Operand layout of v_mfma_f64_4x4x4_4b_f64 on gfx950, found by brute force: which base-4 digit of the lane index is the
// block, the row / column and the k index of A, B and D.   hipcc --offload-arch=gfx950 -o /tmp/t tools/mfma_layout_test_4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *a_in, const double *b_in, double *d_out) {
	const int l = threadIdx.x;
	d_out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a_in[l], b_in[l], 0.0, 0, 0, 0);
}
int main() {
	double ha[64], hb[64], hd[64], *dA, *dB, *dD;
	for (int l = 0; l < 64; ++l) { ha[l] = 1 + std::sin(l * 1.7); hb[l] = 2 + std::cos(l * 0.9); }
	(void)hipMalloc(&dA, sizeof ha); (void)hipMalloc(&dB, sizeof hb); (void)hipMalloc(&dD, sizeof hd);
	(void)hipMemcpy(dA, ha, sizeof ha, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hb, sizeof hb, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	(void)hipMemcpy(hd, dD, sizeof hd, hipMemcpyDeviceToHost);
	const int perm[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
	for (int pa = 0; pa < 6; ++pa) for (int pb = 0; pb < 6; ++pb) for (int pd = 0; pd < 6; ++pd) {
		double A[4][4][4], B[4][4][4];   // A[block][i][k], B[block][k][j]
		for (int l = 0; l < 64; ++l) {
			const int d[3] = {l % 4, (l / 4) % 4, l / 16};
			A[d[perm[pa][0]]][d[perm[pa][1]]][d[perm[pa][2]]] = ha[l];   // digits -> (block, i, k)
			B[d[perm[pb][0]]][d[perm[pb][1]]][d[perm[pb][2]]] = hb[l];   // digits -> (block, k, j)
		}
		double err = 0;
		for (int l = 0; l < 64; ++l) {
			const int d[3] = {l % 4, (l / 4) % 4, l / 16};
			const int bl = d[perm[pd][0]], i = d[perm[pd][1]], j = d[perm[pd][2]];
			double r = 0; for (int kk = 0; kk < 4; ++kk) r += A[bl][i][kk] * B[bl][kk][j];
			err = std::fmax(err, std::fabs(r - hd[l]));
		}
		if (err < 1e-9) printf("A digits(block,i,k)=(%d,%d,%d) B digits(block,k,j)=(%d,%d,%d) D digits(block,i,j)=(%d,%d,%d) MATCH\n",
			perm[pa][0], perm[pa][1], perm[pa][2], perm[pb][0], perm[pb][1], perm[pb][2], perm[pd][0], perm[pd][1], perm[pd][2]);
	}
	printf("done\n");
	return 0;
}
