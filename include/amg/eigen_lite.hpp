// eigen_lite.hpp -- container types for the AMG:: drop-in headers.
//
// The reference's public API is written against Eigen 3.4 containers
// (Eigen::SparseMatrix<EleType>, Eigen::Matrix<EleType,-1,1>; multigrid.hpp:5-7).
// When <Eigen/Sparse> is on the include path the drop-in headers use the real
// thing and this file only pulls it in.  When it is not (this build image has no
// Eigen and no network), a minimal column-major SparseMatrix / dense Vector with
// the subset of the Eigen API that the reference's headers and test driver use is
// provided under the same names, so that user code compiles unchanged.
// Storage only: every multigrid operation runs on the GPU through amg_hip.h.
#pragma once

#if defined(__has_include)
#if __has_include(<Eigen/Sparse>) && !defined(AMG_FORCE_EIGEN_LITE)
#define AMG_HAVE_EIGEN 1
#endif
#endif

#ifdef AMG_HAVE_EIGEN
#include <Eigen/Core>
#include <Eigen/Sparse>
#include <Eigen/SparseCholesky>
#else

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <ostream>
#include <stdexcept>
#include <vector>

namespace Eigen {

typedef std::ptrdiff_t Index;
const int Dynamic = -1;

// ---- dense column vector: Eigen::Matrix<T, -1, 1> -----------------------------
template <class T, int Rows = Dynamic, int Cols = 1>
class Matrix {
  static_assert(Rows == Dynamic && Cols == 1, "eigen_lite: only dynamic column vectors");
  std::vector<T> v_;

 public:
  typedef T Scalar;
  Matrix() {}
  explicit Matrix(Index n) : v_((size_t)n) {}
  Index size() const { return (Index)v_.size(); }
  Index rows() const { return (Index)v_.size(); }
  Index cols() const { return 1; }
  void resize(Index n) { v_.resize((size_t)n); }
  T& operator[](Index i) { return v_[(size_t)i]; }
  const T& operator[](Index i) const { return v_[(size_t)i]; }
  T& operator()(Index i) { return v_[(size_t)i]; }
  const T& operator()(Index i) const { return v_[(size_t)i]; }
  T* data() { return v_.data(); }
  const T* data() const { return v_.data(); }
  Matrix& setZero() { std::fill(v_.begin(), v_.end(), T(0)); return *this; }
  T squaredNorm() const { T s = 0; for (const T& x : v_) s += x * x; return s; }
  T norm() const { return std::sqrt(squaredNorm()); }
  // Eigen: ||a-b||^2 <= prec^2 * min(||a||^2, ||b||^2)
  bool isApprox(const Matrix& o, T prec = T(1e-12)) const {
    if (o.size() != size()) return false;
    T d = 0;
    for (Index i = 0; i < size(); ++i) d += (v_[i] - o[i]) * (v_[i] - o[i]);
    return d <= prec * prec * std::min(squaredNorm(), o.squaredNorm());
  }
  Matrix operator+(const Matrix& o) const { Matrix r(size()); for (Index i = 0; i < size(); ++i) r[i] = v_[i] + o[i]; return r; }
  Matrix operator-(const Matrix& o) const { Matrix r(size()); for (Index i = 0; i < size(); ++i) r[i] = v_[i] - o[i]; return r; }
  static Matrix Zero(Index n) { Matrix r(n); r.setZero(); return r; }
};
typedef Matrix<double, Dynamic, 1> VectorXd;

template <class T>
std::ostream& operator<<(std::ostream& os, const Matrix<T, Dynamic, 1>& v) {
  for (Index i = 0; i < v.size(); ++i) os << v[i] << (i + 1 < v.size() ? "\n" : "");
  return os;
}

template <class T>
class Triplet {
  int r_, c_;
  T v_;

 public:
  Triplet(Index r, Index c, const T& v) : r_((int)r), c_((int)c), v_(v) {}
  int row() const { return r_; }
  int col() const { return c_; }
  const T& value() const { return v_; }
};

// ---- column-major sparse matrix: Eigen::SparseMatrix<T> ------------------------
template <class T>
class SparseMatrix {
  Index rows_ = 0, cols_ = 0;
  std::vector<int> outer_;  // cols_ + 1
  std::vector<int> inner_;
  std::vector<T> val_;

 public:
  typedef T Scalar;
  typedef int StorageIndex;
  SparseMatrix() : outer_(1, 0) {}
  SparseMatrix(Index r, Index c) : rows_(r), cols_(c), outer_((size_t)c + 1, 0) {}
  Index rows() const { return rows_; }
  Index cols() const { return cols_; }
  Index size() const { return rows_ * cols_; }
  Index nonZeros() const { return (Index)inner_.size(); }
  Index outerSize() const { return cols_; }
  bool isCompressed() const { return true; }
  void makeCompressed() {}
  template <class X> void reserve(const X&) {}
  const int* outerIndexPtr() const { return outer_.data(); }
  const int* innerIndexPtr() const { return inner_.data(); }
  const T* valuePtr() const { return val_.data(); }
  int* outerIndexPtr() { return outer_.data(); }
  int* innerIndexPtr() { return inner_.data(); }
  T* valuePtr() { return val_.data(); }

  // adopt compressed arrays (what Eigen::Map<const SparseMatrix> + assignment does)
  static SparseMatrix from_arrays(Index r, Index c, const int* outer, const int* inner,
                                  const T* val) {
    SparseMatrix M(r, c);
    M.outer_.assign(outer, outer + c + 1);
    M.inner_.assign(inner, inner + outer[c]);
    M.val_.assign(val, val + outer[c]);
    return M;
  }

  template <class It>
  void setFromTriplets(It first, It last) {  // no duplicates expected
    std::vector<int> cnt((size_t)cols_ + 1, 0);
    for (It it = first; it != last; ++it) cnt[(size_t)it->col() + 1]++;
    for (Index c = 0; c < cols_; ++c) cnt[c + 1] += cnt[c];
    outer_ = cnt;
    inner_.assign((size_t)cnt[cols_], 0);
    val_.assign((size_t)cnt[cols_], T(0));
    std::vector<int> cur(outer_.begin(), outer_.end() - 1);
    for (It it = first; it != last; ++it) {
      const int q = cur[it->col()]++;
      inner_[q] = it->row();
      val_[q] = it->value();
    }
    for (Index c = 0; c < cols_; ++c) {  // ascending rows inside a column
      std::vector<std::pair<int, T>> tmp;
      for (int q = outer_[c]; q < outer_[c + 1]; ++q) tmp.emplace_back(inner_[q], val_[q]);
      std::sort(tmp.begin(), tmp.end(), [](const std::pair<int, T>& a, const std::pair<int, T>& b) { return a.first < b.first; });
      for (int q = outer_[c], k = 0; q < outer_[c + 1]; ++q, ++k) { inner_[q] = tmp[k].first; val_[q] = tmp[k].second; }
    }
  }

  T coeff(Index r, Index c) const {
    for (int q = outer_[c]; q < outer_[c + 1]; ++q)
      if (inner_[q] == r) return val_[q];
    return T(0);
  }

  SparseMatrix transpose() const {
    SparseMatrix R(cols_, rows_);
    std::vector<int> cnt((size_t)rows_ + 1, 0);
    for (int i : inner_) cnt[(size_t)i + 1]++;
    for (Index r = 0; r < rows_; ++r) cnt[r + 1] += cnt[r];
    R.outer_ = cnt;
    R.inner_.assign(inner_.size(), 0);
    R.val_.assign(val_.size(), T(0));
    std::vector<int> cur(cnt.begin(), cnt.end() - 1);
    for (Index c = 0; c < cols_; ++c)
      for (int q = outer_[c]; q < outer_[c + 1]; ++q) {
        const int p = cur[inner_[q]]++;
        R.inner_[p] = (int)c;
        R.val_[p] = val_[q];
      }
    return R;
  }
};

template <class T>
std::ostream& operator<<(std::ostream& os, const SparseMatrix<T>& A) {
  for (Index r = 0; r < A.rows(); ++r) {
    for (Index c = 0; c < A.cols(); ++c) os << A.coeff(r, c) << (c + 1 < A.cols() ? " " : "");
    os << "\n";
  }
  return os;
}

}  // namespace Eigen
#endif  // AMG_HAVE_EIGEN

#ifndef AMG_HAVE_EIGEN
#include "../amg_hip.h"
namespace Eigen {
// Direct solver used by the reference's test driver for the exact solution
// (testlib.cpp:31-35).  Lite stand-in: the device banded LDL^T of amg_hip.h.
template <class MatrixType>
class SimplicialLDLT {
  MatrixType A_;

 public:
  void analyzePattern(const MatrixType& A) { A_ = A; }
  void factorize(const MatrixType& A) { A_ = A; }
  template <class Vec>
  Vec solve(const Vec& b) const {
    Vec x(b.size());
    if (amg_hip_coarse_solve(A_.rows(), A_.outerIndexPtr(), A_.innerIndexPtr(), A_.valuePtr(),
                             b.data(), x.data(), nullptr) != AMG_HIP_OK)
      throw std::runtime_error(std::string("amg_hip: ") + amg_hip_last_error());
    return x;
  }
};
}  // namespace Eigen
#endif

#include <stdexcept>
#include <string>

#include "../amg_hip.h"

namespace AMG {
namespace detail {

template <class T>
inline Eigen::SparseMatrix<T> make_sparse(std::ptrdiff_t rows, std::ptrdiff_t cols,
                                          const int* outer, const int* inner, const T* val) {
#ifdef AMG_HAVE_EIGEN
  Eigen::SparseMatrix<T> M = Eigen::Map<const Eigen::SparseMatrix<T>>(rows, cols, outer[cols], outer, inner, val);
  return M;
#else
  return Eigen::SparseMatrix<T>::from_arrays(rows, cols, outer, inner, val);
#endif
}

// compressed copy (Eigen matrices may be uncompressed after insert(), grid.hpp:56-69)
template <class T>
inline Eigen::SparseMatrix<T> compressed(const Eigen::SparseMatrix<T>& A) {
  Eigen::SparseMatrix<T> C = A;
  C.makeCompressed();
  return C;
}

// C-ABI status -> the exceptions the reference's callers see
inline void check(amg_hip_status st) {
  if (st == AMG_HIP_OK) return;
  const std::string msg = amg_hip_last_error();
  if (st == AMG_HIP_EINVAL) throw std::invalid_argument(msg);  // multigrid.hpp:165-178
  throw std::runtime_error("amg_hip: " + msg);
}

}  // namespace detail
}  // namespace AMG
