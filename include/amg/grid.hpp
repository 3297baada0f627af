// Drop-in for include/amg/grid.hpp (Poisson test problem with Dirichlet BCs).
#pragma once
#include <cmath>
#include <functional>

#include <amg/eigen_lite.hpp>

namespace AMG {

template <class EleType>
class Grid {
  static const size_t n_boundary_points = 2;

 public:
  // reference grid.hpp:31
  static EleType grid_spacing_h(size_t n) { return 2.0 / (n + 1); }
  // reference grid.hpp:39-41
  static size_t points_n_from_grid_spacing_h(EleType h = 1. / 50) {
    return static_cast<size_t>((2 / h) - 1);
  }

  // reference grid.hpp:50-75: tridiag(1,-2,1)/(h*h)
  static Eigen::SparseMatrix<EleType> second_order_central_difference(size_t n) {
    const EleType h = grid_spacing_h(n);
    std::vector<Eigen::Triplet<EleType>> t;
    for (size_t i = 0; i < n; ++i) {
      t.push_back(Eigen::Triplet<EleType>(i, i, -2.0 / (h * h)));
      if (i > 0) t.push_back(Eigen::Triplet<EleType>(i, i - 1, 1.0 / (h * h)));
      if (i + 1 < n) t.push_back(Eigen::Triplet<EleType>(i, i + 1, 1.0 / (h * h)));
    }
    Eigen::SparseMatrix<EleType> D(n, n);
    D.setFromTriplets(t.begin(), t.end());
    return D;
  }

  // reference grid.hpp:88-98: kron(I,D) + kron(D,I); values and pattern generated
  // by the library (amg_hip_laplacian), bit-identical to the Eigen construction.
  static Eigen::SparseMatrix<EleType> laplacian(size_t n) { return laplacian_nd(2, n); }
  // build-side extension: 7-point operator on an n^3 grid (BASELINE config 5)
  static Eigen::SparseMatrix<EleType> laplacian3d(size_t n) { return laplacian_nd(3, n); }

  // reference grid.hpp:108-140
  static Eigen::Matrix<EleType, -1, 1> rhs(
      size_t n, std::function<EleType(EleType, EleType)> f = nullptr) {
    Eigen::Matrix<EleType, -1, 1> b(n * n);
    if (!f) {  // default Gaussian forcing, grid.hpp:110-112
      detail::check(amg_hip_rhs(2, (int64_t)n, b.data()));
      return b;
    }
    const size_t np = n + n_boundary_points;  // LinSpaced(np, -1, 1)
    const EleType step = (EleType(1) - EleType(-1)) / EleType(np - 1);
    auto x = [&](size_t i) { return i == np - 1 ? EleType(1) : EleType(-1) + EleType(i) * step; };
    size_t dof = 0;
    for (size_t j = 1; j <= n; ++j)
      for (size_t i = 1; i <= n; ++i) b[dof++] = f(x(j), x(i));
    return b;
  }

 private:
  static Eigen::SparseMatrix<EleType> laplacian_nd(int dim, size_t n) {
    static_assert(sizeof(EleType) == sizeof(double), "the MI355X path is fp64 only");
    const int64_t nnz = amg_hip_laplacian(dim, (int64_t)n, nullptr, nullptr, nullptr);
    if (nnz < 0) throw std::invalid_argument(amg_hip_last_error());
    size_t N = n * n;
    if (dim == 3) N *= n;
    std::vector<int> cp(N + 1), ri((size_t)nnz);
    std::vector<double> v((size_t)nnz);
    amg_hip_laplacian(dim, (int64_t)n, cp.data(), ri.data(), v.data());
    return detail::make_sparse<EleType>(N, N, cp.data(), ri.data(), v.data());
  }
};

}  // namespace AMG
