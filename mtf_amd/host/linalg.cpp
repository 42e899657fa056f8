/* linalg.cpp -- the one dense solve the host layer needs (product: mtf::hip::PF's jacobian_as_sigma step and the harness loops use it) */
#include "mtf_types.h"
#include "AppearanceModel.h"

#include <cmath>
#include <vector>

namespace mtf {

/* column-pivoted Householder QR solve, the algorithm behind Eigen's ColPivHouseholderQR::solve */
void utils::colPivHouseholderQrSolve(const MatrixXd &Ain, const double *b, int nb, VectorXd &x) {
	const int n = (int)Ain.rows();
	if ((int)Ain.cols() != n || nb != n) throw InvalidArgument("colPivHouseholderQrSolve: size mismatch");
	MatrixXd A = Ain;
	std::vector<double> rhs(b, b + n), v(n);
	std::vector<int> perm(n);
	for (int j = 0; j < n; ++j) perm[j] = j;
	int rank = n;
	for (int k = 0; k < n; ++k) {
		int piv = k;
		double best = -1;
		for (int j = k; j < n; ++j) {
			double s = 0;
			for (int i = k; i < n; ++i) s += A(i, j) * A(i, j);
			if (s > best) { best = s; piv = j; }
		}
		if (best <= 0) { rank = k; break; }
		if (piv != k) {
			for (int i = 0; i < n; ++i) std::swap(A(i, piv), A(i, k));
			std::swap(perm[piv], perm[k]);
		}
		const double norm = std::sqrt(best);
		const double alpha = A(k, k) > 0 ? -norm : norm;
		double vnorm2 = 0;
		for (int i = k; i < n; ++i) { v[i] = A(i, k); if (i == k) v[i] -= alpha; vnorm2 += v[i] * v[i]; }
		if (vnorm2 > 0) {
			for (int j = k; j < n; ++j) {
				double dot = 0;
				for (int i = k; i < n; ++i) dot += v[i] * A(i, j);
				const double f = 2 * dot / vnorm2;
				for (int i = k; i < n; ++i) A(i, j) -= f * v[i];
			}
			double dot = 0;
			for (int i = k; i < n; ++i) dot += v[i] * rhs[i];
			const double f = 2 * dot / vnorm2;
			for (int i = k; i < n; ++i) rhs[i] -= f * v[i];
		}
	}
	std::vector<double> y(n, 0.0);
	for (int k = rank - 1; k >= 0; --k) {
		double s = rhs[k];
		for (int j = k + 1; j < rank; ++j) s -= A(k, j) * y[j];
		y[k] = s / A(k, k);
	}
	x.resize(n);
	for (int k = 0; k < n; ++k) x[perm[k]] = y[k];
}

} // namespace mtf
