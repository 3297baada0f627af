// Drop-in for include/amg/smoother.hpp: same classes, same constructors, same
// public fields; smooth() runs on the GPU through amg_hip.h.
#pragma once
#include <iostream>
#include <string>

#include <amg/common.hpp>

namespace AMG {

// reference smoother.hpp:18-66
template <class EleType>
class SmootherBase {
 public:
  EleType tolerance{1e-9};
  size_t compute_error_every_n_iters{100};
  size_t n_iters{1};

  SmootherBase() {}
  SmootherBase(size_t n_iters_) : n_iters(n_iters_) {}
  SmootherBase(double tolerance_, size_t compute_error_every_n_iters_, size_t n_iters_)
      : tolerance(tolerance_), compute_error_every_n_iters(compute_error_every_n_iters_),
        n_iters(n_iters_) {}
  virtual ~SmootherBase() {}

  virtual void smooth(const Eigen::SparseMatrix<EleType>& A, Eigen::Matrix<EleType, -1, 1>& u,
                      const Eigen::Matrix<EleType, -1, 1>& b) = 0;
};

namespace detail {
template <class EleType>
inline void device_smooth(int kind, const Eigen::SparseMatrix<EleType>& A,
                          Eigen::Matrix<EleType, -1, 1>& u, const Eigen::Matrix<EleType, -1, 1>& b,
                          double omega, double tol, size_t every, size_t n_iters, int64_t* iters,
                          int32_t* converged) {
  static_assert(sizeof(EleType) == sizeof(double), "the MI355X path is fp64 only");
  const Eigen::SparseMatrix<EleType> C = compressed(A);
  check(amg_hip_smooth(kind, C.rows(), C.outerIndexPtr(), C.innerIndexPtr(), C.valuePtr(),
                       u.data(), b.data(), omega, tol, (int64_t)every, (int64_t)n_iters, iters,
                       converged));
}
}  // namespace detail

// reference smoother.hpp:86-216: symmetric (forward + backward) lexicographic
// Gauss-Seidel, executed on the device in exact sequential order.
template <class EleType>
class SparseGaussSeidel : public SmootherBase<EleType> {
 public:
  using SmootherBase<EleType>::SmootherBase;
  SparseGaussSeidel() {  // smoother.hpp:183-187
    this->tolerance = 1e-9;
    this->compute_error_every_n_iters = 0;
    this->n_iters = 1;
  }
  void smooth(const Eigen::SparseMatrix<EleType>& A, Eigen::Matrix<EleType, -1, 1>& u,
              const Eigen::Matrix<EleType, -1, 1>& b) override {
    int64_t iter = 0;
    int32_t conv = 0;
    detail::device_smooth(AMG_HIP_SM_SPGS, A, u, b, 1.0, this->tolerance,
                          this->compute_error_every_n_iters, this->n_iters, &iter, &conv);
    if (this->compute_error_every_n_iters != 0) {  // smoother.hpp:205-212
      if (conv) std::cout << "SPGS converged after " << iter << " iterations." << std::endl;
      else std::cout << "SPGS did not converge after " << iter << " iterations." << std::endl;
    }
  }
};

// reference smoother.hpp:223-264 (an in-place forward Gauss-Seidel)
template <class EleType>
class Jacobi : public SmootherBase<EleType> {
 public:
  using SmootherBase<EleType>::SmootherBase;
  Jacobi() {}
  void smooth(const Eigen::SparseMatrix<EleType>& A, Eigen::Matrix<EleType, -1, 1>& u,
              const Eigen::Matrix<EleType, -1, 1>& b) override {
    detail::device_smooth(AMG_HIP_SM_REF_JACOBI, A, u, b, 1.0, this->tolerance,
                          this->compute_error_every_n_iters, this->n_iters, nullptr, nullptr);
  }
};

// reference smoother.hpp:271-373
template <class EleType>
class SuccessiveOverRelaxation : public SmootherBase<EleType> {
  double omega{1.0};
  void validate_omega() {  // smoother.hpp:286-293
    if (omega > 2 || omega < 0) {
      std::string msg = "`omega` must be in [0, 2] but got omega=" + std::to_string(omega) + "\n";
      throw std::invalid_argument(msg);
    }
  }

 public:
  using SmootherBase<EleType>::SmootherBase;
  SuccessiveOverRelaxation() {}
  SuccessiveOverRelaxation(double omega_) : omega(omega_) { validate_omega(); }
  SuccessiveOverRelaxation(double omega_, double tolerance_, size_t compute_error_every_n_iters_,
                           size_t n_iters_)
      : SmootherBase<EleType>(tolerance_, compute_error_every_n_iters_, n_iters_), omega(omega_) {
    validate_omega();
  }
  double get_omega() const { return omega; }
  void smooth(const Eigen::SparseMatrix<EleType>& A, Eigen::Matrix<EleType, -1, 1>& u,
              const Eigen::Matrix<EleType, -1, 1>& b) override {
    detail::device_smooth(AMG_HIP_SM_SOR, A, u, b, omega, this->tolerance,
                          this->compute_error_every_n_iters, this->n_iters, nullptr, nullptr);
  }
};

// Build-side addition (no reference counterpart): true two-buffer weighted Jacobi,
// u <- u + omega*(D^-1 (b - (A-D) u) - u), `n_iters` sweeps per smooth().  Fully
// parallel: this is the throughput smoother of the MI355X path.  omega must stay
// below 2/lambda_max(D^-1 A) (~0.67 on the reference's coarse Galerkin levels).
template <class EleType>
class TrueJacobi : public SmootherBase<EleType> {
  double omega{0.6};

 public:
  TrueJacobi() { this->n_iters = 2; this->compute_error_every_n_iters = 0; }
  TrueJacobi(double omega_, size_t n_sweeps = 2) : omega(omega_) {
    this->n_iters = n_sweeps;
    this->compute_error_every_n_iters = 0;
  }
  double get_omega() const { return omega; }
  void smooth(const Eigen::SparseMatrix<EleType>& A, Eigen::Matrix<EleType, -1, 1>& u,
              const Eigen::Matrix<EleType, -1, 1>& b) override {
    detail::device_smooth(AMG_HIP_SM_JACOBI, A, u, b, omega, 0.0, 0, this->n_iters, nullptr, nullptr);
  }
};

}  // namespace AMG
