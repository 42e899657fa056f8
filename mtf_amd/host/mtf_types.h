/*
 * mtf_types.h -- the handful of dense types the AM / SSM interface passes around, with the reference's
 * names and memory layouts (Macros/include/mtf/Macros/common.h:190-258: Eigen, column-major).
 *
 * With Eigen on the include path these ARE the reference's typedefs (the adapters then compile against the types an MTF build
 * uses: SURVEY.md section 7).  Eigen is not available in this image, so without it they are plain storage classes -- enough for the
 * interface and for the search-method loops, which only need element access, `data()`, and a few S x S operations.
 * -DMTF_AMD_NO_EIGEN forces the storage classes, -DMTF_AMD_USE_EIGEN forces Eigen (a compile error if it is absent).
 * Code above this header uses only what both offer: (i, j) / (i) access, data(), rows(), cols(), size(), resize(), fill(), and the
 * free helpers below (squaredDistance, colPivHouseholderQrSolve).
 */
#ifndef MTF_AMD_HOST_TYPES_H
#define MTF_AMD_HOST_TYPES_H

#include <cassert>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

#if !defined(MTF_AMD_NO_EIGEN) && !defined(MTF_AMD_USE_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Dense>)
#define MTF_AMD_USE_EIGEN 1
#endif
#endif

#ifdef MTF_AMD_USE_EIGEN
#include <Eigen/Dense>
namespace mtf {
/* Macros/include/mtf/Macros/common.h:190-258 */
using Eigen::MatrixXd;
using Eigen::VectorXd;
using Eigen::RowVectorXd;
typedef Eigen::Matrix<double, 2, Eigen::Dynamic> PtsT;        /* Matrix2Xd */
typedef Eigen::Matrix<double, 8, Eigen::Dynamic> GradPtsT;    /* Matrix8Xd */
typedef Eigen::Matrix<double, 16, Eigen::Dynamic> HessPtsT;   /* Matrix16Xd */
typedef Eigen::Matrix<double, Eigen::Dynamic, 2> PixGradT;    /* MatrixX2d */
typedef Eigen::Matrix<double, 4, Eigen::Dynamic> PixHessT;    /* Matrix4Xd */
typedef VectorXd PixValT;
typedef Eigen::Matrix<double, 2, 4> CornersT;                 /* Matrix24d: TL TR BR BL */
namespace utils {
inline double squaredDistance(const CornersT &a, const CornersT &b) { return (a - b).squaredNorm(); }
}
#else
namespace mtf {

/* column-major dynamic matrix of doubles */
class MatrixXd {
public:
	MatrixXd() : r_(0), c_(0) {}
	MatrixXd(int rows, int cols) : r_(rows), c_(cols), d_((size_t)rows * cols, 0.0) {}
	void resize(int rows, int cols) { r_ = rows; c_ = cols; d_.assign((size_t)rows * cols, 0.0); }
	int rows() const { return r_; }
	int cols() const { return c_; }
	size_t size() const { return d_.size(); }
	double *data() { return d_.data(); }
	const double *data() const { return d_.data(); }
	double &operator()(int i, int j) { return d_[(size_t)j * r_ + i]; }
	double operator()(int i, int j) const { return d_[(size_t)j * r_ + i]; }
	void fill(double v) { std::fill(d_.begin(), d_.end(), v); }
private:
	int r_, c_;
	std::vector<double> d_;
};

class VectorXd {
public:
	VectorXd() {}
	explicit VectorXd(int n) : d_(n, 0.0) {}
	void resize(int n) { d_.assign(n, 0.0); }
	int size() const { return (int)d_.size(); }
	double *data() { return d_.data(); }
	const double *data() const { return d_.data(); }
	double &operator()(int i) { return d_[i]; }
	double operator()(int i) const { return d_[i]; }
	double &operator[](int i) { return d_[i]; }
	double operator[](int i) const { return d_[i]; }
	void fill(double v) { std::fill(d_.begin(), d_.end(), v); }
	double squaredNorm() const { double s = 0; for (double v : d_) s += v * v; return s; }
private:
	std::vector<double> d_;
};
typedef VectorXd RowVectorXd;

/* The reference's fixed-row typedefs (common.h:190-258) are DISTINCT Eigen types -- Matrix2Xd, Matrix8Xd, Matrix16Xd, MatrixX2d,
 * Matrix4Xd -- and ImageBase overloads on them (initializePixGrad(const PtsT&) vs (const GradPtsT&), ImageBase.h:107-109,
 * 119-120).  They are distinct types here too, so the interface keeps the reference's overload set. */
struct PtsT : MatrixXd { using MatrixXd::MatrixXd; };       /* 2 x N points, x,y interleaved (Matrix2Xd) */
struct GradPtsT : MatrixXd { using MatrixXd::MatrixXd; };   /* 8 x N (Matrix8Xd) */
struct HessPtsT : MatrixXd { using MatrixXd::MatrixXd; };   /* 16 x N (Matrix16Xd) */
struct PixGradT : MatrixXd { using MatrixXd::MatrixXd; };   /* N x 2 (MatrixX2d) */
struct PixHessT : MatrixXd { using MatrixXd::MatrixXd; };   /* 4 x N (Matrix4Xd) */
typedef VectorXd PixValT;

/* 2 x 4 corners, TL TR BR BL (Matrix24d) */
struct CornersT {
	double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	double &operator()(int r, int c) { return v[2 * c + r]; }
	double operator()(int r, int c) const { return v[2 * c + r]; }
	double *data() { return v; }
	const double *data() const { return v; }
};

namespace utils {
inline double squaredDistance(const CornersT &a, const CornersT &b) {
	double s = 0;
	for (int i = 0; i < 8; ++i) { const double d = a.data()[i] - b.data()[i]; s += d * d; }
	return s;
}
}
#endif   /* MTF_AMD_USE_EIGEN */

/* the float32 image the AM borrows (cv::Mat CV_32FC1 / CV_32FC3 in the reference) */
struct ImageView {
	const float *data;
	int rows, cols, step; /* step in elements (floats) */
	int channels = 1;     /* 1: CV_32FC1, 3: CV_32FC3 interleaved */
};
} // namespace mtf
/* With OpenCV on the include path an ImageView is made from the `const cv::Mat &` the reference's setCurrImg / setImage take
 * (AM/include/mtf/AM/ImageBase.h:92, include/mtf/TrackerBase.h:22-26): `am.setCurrImg(mtf::imageView(img))`, or directly through the
 * cv::Mat overloads the adapters then declare (HipModels.h).  -DMTF_AMD_NO_OPENCV switches it off. */
#if !defined(MTF_AMD_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#define MTF_AMD_USE_OPENCV 1
#endif
#endif
namespace mtf {
#ifdef MTF_AMD_USE_OPENCV
inline ImageView imageView(const cv::Mat &img) {
	if (img.depth() != CV_32F || (img.channels() != 1 && img.channels() != 3))
		throw std::invalid_argument("ImageBase::setCurrImg: a CV_32FC1 / CV_32FC3 image is expected");   /* ImageBase.cc:55-59 */
	ImageView v{reinterpret_cast<const float *>(img.data), img.rows, img.cols, (int)(img.step / sizeof(float))};
	v.channels = img.channels();
	return v;
}
#endif

namespace utils {
/* Utilities/include/mtf/Utilities/excpUtils.h:8-55 */
class Exception : public std::runtime_error {
public:
	explicit Exception(const std::string &what, const char *type = "Generic") : std::runtime_error(what), type_(type) {}
	const char *type() const { return type_; }
private:
	const char *type_;
};
struct InvalidArgument : Exception { explicit InvalidArgument(const std::string &w) : Exception(w, "InvalidArgument") {} };
struct FunctonNotImplemented : Exception { explicit FunctonNotImplemented(const std::string &w) : Exception(w, "FunctonNotImplemented") {} };
struct LogicError : Exception { explicit LogicError(const std::string &w) : Exception(w, "LogicError") {} };
struct InvalidTrackerState : Exception { explicit InvalidTrackerState(const std::string &w) : Exception(w, "InvalidTrackerState") {} };

/* x = A^{-1} b through a column-pivoted Householder QR (what the SMs call on Eigen:
 * hessian.colPivHouseholderQr().solve(...), SM/src/NT/FCLK.cc:298) */
void colPivHouseholderQrSolve(const MatrixXd &A, const double *b, int n, VectorXd &x);
inline void colPivHouseholderQrSolve(const MatrixXd &A, const VectorXd &b, VectorXd &x) { colPivHouseholderQrSolve(A, b.data(), (int)b.size(), x); }
#ifdef MTF_AMD_USE_EIGEN   /* (without Eigen RowVectorXd IS VectorXd) */
inline void colPivHouseholderQrSolve(const MatrixXd &A, const RowVectorXd &b, VectorXd &x) { colPivHouseholderQrSolve(A, b.data(), (int)b.size(), x); }
#endif
} // namespace utils

} // namespace mtf
#endif
