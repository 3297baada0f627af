// interpolator.hpp -- AMG::InterpolatorBase / AMG::LinearInterpolator of the drop-in.
//
// Same public interface as the reference's interpolator.hpp (constructors, the pure
// virtual make_operators(n_h, n_H, level), prolongation / restriction, the P / R getters
// and setters); the products run on the device through the C ABI.  Inside the V-cycle
// the operators of LinearInterpolator are recognised and replaced by matrix-free kernels.
#pragma once
#include <array>
#include <stdexcept>
#include <vector>

#include <amg/eigen_lite.hpp>

namespace AMG {

template <class EleType>
class InterpolatorBase {
  using Sparse = Eigen::SparseMatrix<EleType>;
  using Vector = Eigen::Matrix<EleType, -1, 1>;

  std::vector<Sparse> prolong_;   // level l -> P_l (fine x coarse)
  std::vector<Sparse> restrict_;  // level l -> R_l (coarse x fine)

  // M * v as one device SpMV (interpolator.hpp:52-56 and :64-68 of the reference)
  static Vector device_product(const Sparse& M, const Vector& v) {
    static_assert(sizeof(EleType) == sizeof(double), "the MI355X path is fp64 only");
    const Sparse packed = detail::compressed(M);
    Vector out(packed.rows());
    detail::check(amg_hip_spmv(packed.rows(), packed.cols(), packed.outerIndexPtr(),
                               packed.innerIndexPtr(), packed.valuePtr(), v.data(), out.data()));
    return out;
  }

 public:
  InterpolatorBase() = default;
  InterpolatorBase(size_t n_levels) : prolong_(n_levels - 1), restrict_(n_levels - 1) {}
  virtual ~InterpolatorBase() = default;

  // fills P and R of `level` for a fine level of n_h_dofs and a coarse one of n_H_dofs
  virtual void make_operators(size_t n_h_dofs, size_t n_H_dofs, size_t level) = 0;

  Vector prolongation(const Vector& v, size_t level) { return device_product(prolong_[level], v); }
  Vector restriction(const Vector& v, size_t level) { return device_product(restrict_[level], v); }

  const Sparse& get_P(size_t level) const { return prolong_[level]; }
  const Sparse& get_R(size_t level) const { return restrict_[level]; }
  void set_level_to_P(size_t level, Sparse& P) { prolong_[level] = P; }
  void set_level_to_R(size_t level, Sparse& R) { restrict_[level] = R; }
  size_t n_operator_levels() const { return prolong_.size(); }
};

// 1-D linear interpolation in the flat index: coarse dof c feeds fine rows 2c, 2c+1, 2c+2
// with weights 1/2, 1, 1/2 (rows past the fine level are dropped); R is P transposed.
template <class EleType>
class LinearInterpolator : public InterpolatorBase<EleType> {
 public:
  using InterpolatorBase<EleType>::InterpolatorBase;

  void make_operators(size_t n_h_dofs, size_t n_H_dofs, size_t level) override {
    static const std::array<EleType, 3> weight = {EleType(0.5), EleType(1.0), EleType(0.5)};
    std::vector<Eigen::Triplet<EleType>> entries;
    entries.reserve(weight.size() * n_H_dofs);
    for (size_t coarse = 0; coarse < n_H_dofs; ++coarse)
      for (size_t k = 0; k < weight.size(); ++k) {
        const size_t fine = 2 * coarse + k;
        if (fine < n_h_dofs) entries.emplace_back(fine, coarse, weight[k]);
      }
    Eigen::SparseMatrix<EleType> P(n_h_dofs, n_H_dofs);
    P.setFromTriplets(entries.begin(), entries.end());
    Eigen::SparseMatrix<EleType> R = P.transpose();
    this->set_level_to_P(level, P);
    this->set_level_to_R(level, R);
  }
};

// Strength-based C/F coarsening with direct interpolation -- the alternative the reference's
// README.md:104-109 names and does not build (NO reference counterpart).  Its operators depend
// on the level MATRIX, which make_operators(n_h, n_H, level) is never told, so AMG::Multigrid
// recognises the class and lets the library build the hierarchy (amg_hip_create_rs): `n_levels`
// is then an upper bound, Multigrid::get_n_levels() tells how many were built, and get_P /
// get_R hold the operators afterwards.
template <class EleType>
class RugeStuebenInterpolator : public InterpolatorBase<EleType> {
  EleType theta_;
  size_t min_coarse_;

 public:
  explicit RugeStuebenInterpolator(size_t max_levels, EleType theta = 0.25, size_t min_coarse = 500)
      : InterpolatorBase<EleType>(max_levels), theta_(theta), min_coarse_(min_coarse) {}
  EleType theta() const { return theta_; }
  size_t min_coarse() const { return min_coarse_; }
  void make_operators(size_t, size_t, size_t) override {
    throw std::logic_error("RugeStuebenInterpolator: the operators depend on the level matrix; "
                           "AMG::Multigrid builds them");
  }
};

}  // namespace AMG
