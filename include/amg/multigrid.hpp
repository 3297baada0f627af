// Drop-in for include/amg/multigrid.hpp: AMG::Multigrid<double> with the
// reference's constructor, vcycle(), solve() and getters; the hierarchy lives on
// the GPU behind the C ABI (amg_hip.h).
#pragma once
#include <iostream>
#include <string>
#include <vector>

#include <amg/common.hpp>
#include <amg/grid.hpp>
#include <amg/interpolator.hpp>
#include <amg/smoother.hpp>

namespace AMG {

template <class EleType>
class Multigrid {
  static_assert(sizeof(EleType) == sizeof(double), "the MI355X path is fp64 only");
  typedef Eigen::SparseMatrix<EleType> Sparse;
  typedef Eigen::Matrix<EleType, -1, 1> Vector;

  InterpolatorBase<EleType>* interpolator;  // non-owning, as in the reference (:29-31)
  SmootherBase<EleType>* smoother;
  EleType tolerance;
  size_t compute_error_every_n_iters;
  size_t n_iters;
  size_t n_levels;
  amg_hip_solver* handle = nullptr;
  std::vector<size_t> level_to_n_dofs;
  std::vector<Sparse> level_to_coefficient_matrix;  // host copies for the getters
  mutable std::vector<Vector> level_to_soln, level_to_rhs;
  bool display_error{false};
  bool custom_smoother{false};

  // host round trip for a user-defined smoother on one level
  void host_smooth(size_t level) {
    Vector u = fetch(level_to_soln, level, 0);
    const Vector& f = fetch(level_to_rhs, level, 1);
    smoother->smooth(level_to_coefficient_matrix[level], u, f);
    detail::check(amg_hip_set_vec(handle, (int32_t)level, 0, u.data()));
  }

  size_t n_H_dofs_from_n_h_dofs(size_t h_dofs) { return (h_dofs + 1) / 2 - 1; }  // :127-130

  void fetch_level_matrices() {
    level_to_coefficient_matrix.resize(n_levels);
    for (size_t l = 0; l < n_levels; ++l) {
      const size_t n = level_to_n_dofs[l];
      const int64_t nnz = amg_hip_get_level_nnz(handle, (int32_t)l);
      std::vector<int32_t> cp(n + 1), ri((size_t)nnz);
      std::vector<double> v((size_t)nnz);
      detail::check(amg_hip_get_level_matrix(handle, (int32_t)l, cp.data(), ri.data(), v.data()));
      level_to_coefficient_matrix[l] = detail::make_sparse<EleType>(n, n, cp.data(), ri.data(), v.data());
    }
    level_to_soln.resize(n_levels);
    level_to_rhs.resize(n_levels);
  }

  const Vector& fetch(std::vector<Vector>& cache, size_t level, int which) const {
    Vector& v = cache[level];
    if ((size_t)v.size() != level_to_n_dofs[level]) v.resize(level_to_n_dofs[level]);
    detail::check(amg_hip_get_vec(handle, (int32_t)level, which, v.data()));
    return v;
  }

 public:
  ~Multigrid() { amg_hip_destroy(handle); }
  Multigrid() = delete;
  Multigrid(const Multigrid&) = delete;
  Multigrid& operator=(const Multigrid&) = delete;

  // reference multigrid.hpp:151-244
  Multigrid(AMG::InterpolatorBase<EleType>* interpolator_, AMG::SmootherBase<EleType>* smoother_,
            const Sparse& A, const Vector& b, size_t n_levels_, EleType tolerance_ = 1e-9,
            size_t compute_error_every_n_iters_ = 10, size_t n_iters_ = 100)
      : interpolator(interpolator_), smoother(smoother_), tolerance(tolerance_),
        compute_error_every_n_iters(compute_error_every_n_iters_), n_iters(n_iters_),
        n_levels(n_levels_) {
    std::string errormsg;
    if (compute_error_every_n_iters > n_iters) {  // :165-171
      errormsg = "`compute_error_every_n_iters` must be leq to `n_iters`, got " +
                 std::to_string(compute_error_every_n_iters) + " and " + std::to_string(n_iters);
      throw(std::invalid_argument(errormsg));
    }
    if (A.rows() != b.rows()) {  // :173-178
      errormsg = "`A` and `b` must have the same number of degrees of freedom, got " +
                 std::to_string(A.rows()) + " and " + std::to_string(b.rows());
      throw(std::invalid_argument(errormsg));
    }

    amg_hip_options opt;
    amg_hip_default_options(&opt);
    opt.smoother_iters = (int32_t)smoother->n_iters;
    if (dynamic_cast<SparseGaussSeidel<EleType>*>(smoother)) {
      opt.smoother = AMG_HIP_SM_SPGS;
    } else if (dynamic_cast<Jacobi<EleType>*>(smoother)) {
      opt.smoother = AMG_HIP_SM_REF_JACOBI;
    } else if (auto* sor = dynamic_cast<SuccessiveOverRelaxation<EleType>*>(smoother)) {
      opt.smoother = AMG_HIP_SM_SOR;
      opt.omega = sor->get_omega();
    } else if (auto* tj = dynamic_cast<TrueJacobi<EleType>*>(smoother)) {
      opt.smoother = AMG_HIP_SM_JACOBI;
      opt.omega = tj->get_omega();
    } else {
      // user-defined SmootherBase: its smooth() runs on the host, everything else of the
      // V-cycle on the device; the device-side smoother is never used
      custom_smoother = true;
    }
    // A reference smoother built with the (tolerance, every, n_iters) constructors checks rss
    // every `compute_error_every_n_iters` sweeps, may leave its loop early and prints its
    // convergence line on EVERY smooth() call (smoother.hpp:195-212).  The fused device
    // V-cycle runs a fixed number of sweeps and prints nothing, so such a smoother goes
    // through its own smooth() level by level: same sweep counts, same lines.
    if (!custom_smoother && opt.smoother != AMG_HIP_SM_JACOBI &&
        smoother->compute_error_every_n_iters != 0)
      custom_smoother = true;
    if (custom_smoother) {
      opt.smoother = AMG_HIP_SM_JACOBI;
      opt.smoother_iters = 0;
    }

    if (auto* rs = dynamic_cast<RugeStuebenInterpolator<EleType>*>(interpolator)) {
      // strength-based coarsening: the library derives level sizes and operators from A
      const Sparse A0 = detail::compressed(A);
      detail::check(amg_hip_create_rs(A0.rows(), A0.outerIndexPtr(), A0.innerIndexPtr(), A0.valuePtr(),
                                      b.data(), (int32_t)n_levels, (double)rs->theta(),
                                      (int64_t)rs->min_coarse(), &opt, &handle));
      try {
        n_levels = (size_t)amg_hip_n_levels(handle);
        level_to_n_dofs.resize(n_levels);
        for (size_t l = 0; l < n_levels; ++l)
          level_to_n_dofs[l] = (size_t)amg_hip_get_n_dofs(handle, (int32_t)l);
        for (size_t l = 0; l + 1 < n_levels; ++l)
          for (int which = 0; which < 2; ++which) {  // 0 = P (n_l x n_{l+1}), 1 = R
            const int64_t nnz = amg_hip_get_transfer_nnz(handle, (int32_t)l, which);
            const size_t rows = which == 0 ? level_to_n_dofs[l] : level_to_n_dofs[l + 1];
            const size_t cols = which == 0 ? level_to_n_dofs[l + 1] : level_to_n_dofs[l];
            std::vector<int32_t> cp(cols + 1), ri((size_t)nnz);
            std::vector<double> v((size_t)nnz);
            detail::check(amg_hip_get_transfer(handle, (int32_t)l, which, cp.data(), ri.data(), v.data()));
            Sparse M = detail::make_sparse<EleType>(rows, cols, cp.data(), ri.data(), v.data());
            if (which == 0) interpolator->set_level_to_P(l, M);
            else interpolator->set_level_to_R(l, M);
          }
        fetch_level_matrices();
      } catch (...) {
        amg_hip_destroy(handle);
        handle = nullptr;
        throw;
      }
      return;
    }

    // level sizes + transfer operators: make_operators runs on the host and
    // overwrites levels 0..L-2 of the caller's interpolator, as in the
    // reference (:211-216); P/R go to the device as CSC triples.
    level_to_n_dofs.resize(n_levels);
    level_to_n_dofs[0] = (size_t)A.rows();
    std::vector<Sparse> Ps, Rs;
    std::vector<const int32_t*> Pc, Pr, Rc, Rr;
    std::vector<const double*> Pv, Rv;
    for (size_t level = 1; level < n_levels; ++level) {
      const size_t n_h = level_to_n_dofs[level - 1];
      const size_t n_H = n_H_dofs_from_n_h_dofs(n_h);
      if (n_H < 1 || n_H >= n_h)
        throw std::invalid_argument("level " + std::to_string(level) +
                                    " would have no degrees of freedom; reduce `n_levels`");
      level_to_n_dofs[level] = n_H;
      interpolator->make_operators(n_h, n_H, level - 1);
      Ps.push_back(detail::compressed(interpolator->get_P(level - 1)));
      Rs.push_back(detail::compressed(interpolator->get_R(level - 1)));
    }
    for (size_t l = 0; l + 1 < n_levels; ++l) {
      Pc.push_back(Ps[l].outerIndexPtr()); Pr.push_back(Ps[l].innerIndexPtr()); Pv.push_back(Ps[l].valuePtr());
      Rc.push_back(Rs[l].outerIndexPtr()); Rr.push_back(Rs[l].innerIndexPtr()); Rv.push_back(Rs[l].valuePtr());
    }
    const Sparse A0 = detail::compressed(A);
    detail::check(amg_hip_create_custom(A0.rows(), A0.outerIndexPtr(), A0.innerIndexPtr(),
                                        A0.valuePtr(), b.data(), (int32_t)n_levels, Pc.data(),
                                        Pr.data(), Pv.data(), Rc.data(), Rr.data(), Rv.data(), &opt,
                                        &handle));
    // host copies of the level matrices (get_coefficient_matrix returns const&); the
    // destructor does not run when a constructor throws, so the handle is released here
    try {
      fetch_level_matrices();
    } catch (...) {
      amg_hip_destroy(handle);
      handle = nullptr;
      throw;
    }
  }

  // reference multigrid.hpp:263-305
  void vcycle() {
    if (!custom_smoother) {
      detail::check(amg_hip_vcycle(handle));
      return;
    }
    for (size_t level = 0; level < n_levels; ++level) {
      host_smooth(level);                                                   // :268
      detail::check(amg_hip_level_op(handle, (int32_t)level, 1));          // :272-274
      if (level + 1 != n_levels)
        detail::check(amg_hip_level_op(handle, (int32_t)level, 2));        // :278-282
    }
    detail::check(amg_hip_level_op(handle, (int32_t)n_levels - 1, 4));     // :287-288
    for (int level = (int)n_levels - 2; level >= 0; --level) {
      detail::check(amg_hip_level_op(handle, level, 3));                   // :294-296
      host_smooth((size_t)level);                                           // :300
    }
  }

  // reference multigrid.hpp:311-337
  const Vector& solve() {
    size_t iter = 0;
    EleType error = 100;
    while (iter < n_iters && error > tolerance) {
      vcycle();
      iter += 1;
      if ((iter % compute_error_every_n_iters) == 0) {
        double e = 0;
        detail::check(amg_hip_rss(handle, &e));
        error = e;
        if (display_error) std::cout << "Iter: " << iter << " | Error: " << error << std::endl;
      }
    }
    if (error <= tolerance) std::cout << "AMG converged after " << iter << " iterations." << std::endl;
    else std::cout << "AMG did not converge after " << iter << " iterations." << std::endl;
    return fetch(level_to_soln, 0, 0);
  }

  const Sparse& get_coefficient_matrix(size_t level) const { return level_to_coefficient_matrix[level]; }
  const Vector& get_soln(size_t level) const { return fetch(level_to_soln, level, 0); }
  const Vector& get_rhs(size_t level) const { return fetch(level_to_rhs, level, 1); }
  const size_t get_n_dofs(size_t level) const { return level_to_n_dofs[level]; }
  size_t get_n_levels() const { return n_levels; }  // extension: RugeStuebenInterpolator decides
  const EleType get_tolerance() const { return tolerance; }
  void display_error_on() { display_error = true; }
  void display_error_off() { display_error = true; }  // sic: reference multigrid.hpp:361-364

  amg_hip_solver* native_handle() { return handle; }
  // false when smooth() of a user-defined (or rss-checking) smoother runs on the host
  bool runs_on_device() const { return !custom_smoother; }
};

}  // namespace AMG

namespace amg = AMG;  // BASELINE.json spells the namespace in lower case
