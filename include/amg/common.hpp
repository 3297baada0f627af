// Drop-in for include/amg/common.hpp of jfdev001/algebraic-multigrid.
#pragma once
#include <amg/eigen_lite.hpp>

namespace AMG {

/**
 * Residual sum of squares of `A u` against `b` (reference common.hpp:17-27).
 * Runs on the GPU (amg_hip_rss_host): per-row terms are bit-identical to the
 * reference's, the sum is a tree reduction (agrees to ~1e-15 relative) and costs
 * one SpMV instead of the reference's accidental O(N nnz).
 */
template <class EleType>
EleType rss(const Eigen::SparseMatrix<EleType>& A, const Eigen::Matrix<EleType, -1, 1>& u,
            const Eigen::Matrix<EleType, -1, 1>& b) {
  static_assert(sizeof(EleType) == sizeof(double), "the MI355X path is fp64 only");
  const Eigen::SparseMatrix<EleType> C = detail::compressed(A);
  double out = 0;
  detail::check(amg_hip_rss_host(C.rows(), C.outerIndexPtr(), C.innerIndexPtr(), C.valuePtr(),
                                 u.data(), b.data(), &out));
  return out;
}

}  // namespace AMG
