// Drop-in for include/amg/interpolator.hpp.
#pragma once
#include <vector>

#include <amg/eigen_lite.hpp>

namespace AMG {

// reference interpolator.hpp:15-87
template <class EleType>
class InterpolatorBase {
  std::vector<Eigen::SparseMatrix<EleType>> level_to_P;
  std::vector<Eigen::SparseMatrix<EleType>> level_to_R;

  static Eigen::Matrix<EleType, -1, 1> apply(const Eigen::SparseMatrix<EleType>& M,
                                             const Eigen::Matrix<EleType, -1, 1>& v) {
    static_assert(sizeof(EleType) == sizeof(double), "the MI355X path is fp64 only");
    const Eigen::SparseMatrix<EleType> C = detail::compressed(M);
    Eigen::Matrix<EleType, -1, 1> result(C.rows());
    detail::check(amg_hip_spmv(C.rows(), C.cols(), C.outerIndexPtr(), C.innerIndexPtr(),
                               C.valuePtr(), v.data(), result.data()));
    return result;
  }

 public:
  InterpolatorBase(size_t n_levels) {
    level_to_P.resize(n_levels - 1);
    level_to_R.resize(n_levels - 1);
  }
  InterpolatorBase() {}
  virtual ~InterpolatorBase() {}

  virtual void make_operators(size_t n_h_dofs, size_t n_H_dofs, size_t level) = 0;

  // P_level * v and R_level * v, on the device (interpolator.hpp:52-56, :64-68)
  Eigen::Matrix<EleType, -1, 1> prolongation(const Eigen::Matrix<EleType, -1, 1>& v, size_t level) {
    return apply(get_P(level), v);
  }
  Eigen::Matrix<EleType, -1, 1> restriction(const Eigen::Matrix<EleType, -1, 1>& v, size_t level) {
    return apply(get_R(level), v);
  }

  const Eigen::SparseMatrix<EleType>& get_P(size_t level) const { return level_to_P[level]; }
  const Eigen::SparseMatrix<EleType>& get_R(size_t level) const { return level_to_R[level]; }
  void set_level_to_P(size_t level, Eigen::SparseMatrix<EleType>& P) { level_to_P[level] = P; }
  void set_level_to_R(size_t level, Eigen::SparseMatrix<EleType>& R) { level_to_R[level] = R; }
  size_t n_operator_levels() const { return level_to_P.size(); }
};

// reference interpolator.hpp:98-142
template <class EleType>
class LinearInterpolator : public InterpolatorBase<EleType> {
  const size_t n_elements_per_columns = 3;

 public:
  using InterpolatorBase<EleType>::InterpolatorBase;

  void make_operators(size_t n_h_dofs, size_t n_H_dofs, size_t level) override {
    Eigen::SparseMatrix<EleType> P(n_h_dofs, n_H_dofs);
    std::vector<Eigen::Triplet<EleType>> P_coefficients;
    P_coefficients.reserve(n_H_dofs * n_elements_per_columns);
    size_t i = 0;
    for (size_t j = 0; j < n_H_dofs; ++j) {
      if (i < n_h_dofs) P_coefficients.push_back(Eigen::Triplet<EleType>(i, j, 0.5));
      if (i + 1 < n_h_dofs) P_coefficients.push_back(Eigen::Triplet<EleType>(i + 1, j, 1.0));
      if (i + 2 < n_h_dofs) P_coefficients.push_back(Eigen::Triplet<EleType>(i + 2, j, 0.5));
      i += n_elements_per_columns - 1;
    }
    P.setFromTriplets(P_coefficients.begin(), P_coefficients.end());
    Eigen::SparseMatrix<EleType> R(n_H_dofs, n_h_dofs);
    R = P.transpose();
    this->set_level_to_P(level, P);
    this->set_level_to_R(level, R);
  }
};

}  // namespace AMG
