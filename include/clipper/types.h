/**
 * @file types.h
 * @brief Types of the clipper:: facade (mirror of the reference include/clipper/types.h:15-23
 *        and invariants/abstract.h:19-20).
 *
 * The reference's public types ARE Eigen types. Eigen is not available in the image this was
 * built in, so the facade is written against the small common subset
 *     rows(), cols(), size(), data(), operator()(i,j), operator()(i)
 * and two interchangeable type families provide it:
 *   - with Eigen present (`__has_include(<Eigen/Dense>)`, or -DCLIPPER_USE_EIGEN): exactly the
 *     reference's aliases, so existing call sites compile unchanged;
 *   - otherwise: the minimal column-major containers below (same memory layout as Eigen's
 *     defaults: column-major, Association = m x 2 int, each datum of Data contiguous).
 */
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

#if !defined(CLIPPER_NO_EIGEN) && (defined(CLIPPER_USE_EIGEN) || __has_include(<Eigen/Dense>))
#define CLIPPER_HAVE_EIGEN 1
#include <Eigen/Dense>
#include <Eigen/Sparse>
#endif

namespace clipper {

#ifdef CLIPPER_HAVE_EIGEN

using SpMat = Eigen::SparseMatrix<double>;
using SpTriplet = Eigen::Triplet<double>;
using Association = Eigen::Matrix<int, Eigen::Dynamic, 2>;
using Affinity = Eigen::MatrixXd;
using Constraint = Eigen::MatrixXd;
using SpAffinity = SpMat;
using SpConstraint = SpMat;
using MatrixXd = Eigen::MatrixXd;
using VectorXd = Eigen::VectorXd;
using VectorXi = Eigen::VectorXi;

#else

/// Column-major dense matrix with the subset of the Eigen::Matrix interface the facade uses.
template <typename T>
class DenseMatrix {
 public:
  DenseMatrix() = default;
  DenseMatrix(std::ptrdiff_t rows, std::ptrdiff_t cols)
      : rows_(rows), cols_(cols), d_(static_cast<size_t>(rows * cols)) {}
  static DenseMatrix Zero(std::ptrdiff_t rows, std::ptrdiff_t cols) {
    DenseMatrix m(rows, cols);
    std::fill(m.d_.begin(), m.d_.end(), T(0));
    return m;
  }
  static DenseMatrix Identity(std::ptrdiff_t rows, std::ptrdiff_t cols) {
    DenseMatrix m = Zero(rows, cols);
    for (std::ptrdiff_t i = 0; i < rows && i < cols; ++i) m(i, i) = T(1);
    return m;
  }
  std::ptrdiff_t rows() const { return rows_; }
  std::ptrdiff_t cols() const { return cols_; }
  std::ptrdiff_t size() const { return rows_ * cols_; }
  T* data() { return d_.data(); }
  const T* data() const { return d_.data(); }
  T& operator()(std::ptrdiff_t i, std::ptrdiff_t j) { return d_[static_cast<size_t>(i + j * rows_)]; }
  const T& operator()(std::ptrdiff_t i, std::ptrdiff_t j) const {
    return d_[static_cast<size_t>(i + j * rows_)];
  }
  void resize(std::ptrdiff_t rows, std::ptrdiff_t cols) {
    rows_ = rows;
    cols_ = cols;
    d_.assign(static_cast<size_t>(rows * cols), T(0));
  }
  bool operator==(const DenseMatrix& o) const {
    return rows_ == o.rows_ && cols_ == o.cols_ && d_ == o.d_;
  }

 protected:
  std::ptrdiff_t rows_ = 0, cols_ = 0;
  std::vector<T> d_;
};

/// Column vector (Eigen::VectorXd / VectorXi stand-in).
template <typename T>
class DenseVector : public DenseMatrix<T> {
 public:
  DenseVector() = default;
  explicit DenseVector(std::ptrdiff_t n) : DenseMatrix<T>(n, 1) {}
  DenseVector(const T* p, std::ptrdiff_t n) : DenseMatrix<T>(n, 1) {
    for (std::ptrdiff_t i = 0; i < n; ++i) this->d_[static_cast<size_t>(i)] = p[i];
  }
  static DenseVector Zero(std::ptrdiff_t n) {
    DenseVector v(n);
    std::fill(v.d_.begin(), v.d_.end(), T(0));
    return v;
  }
  using DenseMatrix<T>::operator();
  T& operator()(std::ptrdiff_t i) { return this->d_[static_cast<size_t>(i)]; }
  const T& operator()(std::ptrdiff_t i) const { return this->d_[static_cast<size_t>(i)]; }
  T& operator[](std::ptrdiff_t i) { return this->d_[static_cast<size_t>(i)]; }
  const T& operator[](std::ptrdiff_t i) const { return this->d_[static_cast<size_t>(i)]; }
  void resize(std::ptrdiff_t n) { DenseMatrix<T>::resize(n, 1); }
};

/// Compressed sparse column matrix (Eigen::SparseMatrix<double> stand-in for
/// setSparseMatrixData: strictly upper triangular, no diagonal, reference clipper.h:137-146).
struct SpMat {
  std::ptrdiff_t nrows = 0, ncols = 0;
  std::vector<int64_t> colptr;  ///< ncols + 1
  std::vector<int32_t> rowidx;
  std::vector<double> values;
  std::ptrdiff_t rows() const { return nrows; }
  std::ptrdiff_t cols() const { return ncols; }
  std::ptrdiff_t nonZeros() const { return static_cast<std::ptrdiff_t>(values.size()); }
};

using MatrixXd = DenseMatrix<double>;
using VectorXd = DenseVector<double>;
using VectorXi = DenseVector<int>;
using Association = DenseMatrix<int>;  ///< m x 2
using Affinity = MatrixXd;
using Constraint = MatrixXd;
using SpAffinity = SpMat;
using SpConstraint = SpMat;

#endif  // CLIPPER_HAVE_EIGEN

}  // namespace clipper
