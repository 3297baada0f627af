// AMG::PCG<double> -- conjugate gradients preconditioned with one V-cycle of an
// AMG::Multigrid<double> (M^-1 v).  A build-side addition: the reference names this use of its
// V-cycle (README.md:127, its ref [7]: "a single V-cycle used for preconditioner to get M^-1 v")
// but ships no Krylov driver.  Everything runs on the GPU behind the C ABI (amg_hip_pcg).
#pragma once
#include <iostream>
#include <stdexcept>

#include <amg/multigrid.hpp>

namespace AMG {

template <class EleType>
class PCG {
  typedef Eigen::Matrix<EleType, -1, 1> Vector;
  Multigrid<EleType>* mg;  // non-owning, like the plug-ins of Multigrid itself
  EleType rtol;
  size_t max_iters;
  size_t n_done{0};
  EleType relres{0};

 public:
  // mg: its smoother must make the cycle symmetric (SparseGaussSeidel, TrueJacobi); a
  // user-defined SmootherBase runs on the host and cannot take part in the device iteration
  PCG(Multigrid<EleType>* mg_, EleType rtol_ = 1e-10, size_t max_iters_ = 100)
      : mg(mg_), rtol(rtol_), max_iters(max_iters_) {
    if (!mg) throw std::invalid_argument("`mg` must not be null");
    if (!mg->runs_on_device())
      throw std::invalid_argument("PCG needs a Multigrid whose smoother runs on the device");
  }
  // ||b - A x|| <= rtol ||b||, starting from mg's current solution; returns it
  const Vector& solve() {
    int64_t it = 0;
    double rel = 0;
    detail::check(amg_hip_pcg(mg->native_handle(), (double)rtol, (int64_t)max_iters, &it, &rel));
    n_done = (size_t)it;
    relres = (EleType)rel;
    if (relres <= rtol) std::cout << "PCG converged after " << n_done << " iterations." << std::endl;
    else std::cout << "PCG did not converge after " << n_done << " iterations." << std::endl;
    return mg->get_soln(0);
  }
  size_t iterations() const { return n_done; }
  EleType relative_residual() const { return relres; }
};

}  // namespace AMG
