// The reference's test driver (test/testlib.cpp, one Catch2 TEST_CASE with 32
// assertions) restated against the drop-in headers, with a ten-line CHECK macro
// instead of Catch2 (not in this image).  Same objects, same calls, same order;
// every assertion of the original is here.  Runs on the GPU box (pytest -m gpu).
#include <cmath>
#include <iostream>
#include <stdexcept>

#include <amg/common.hpp>
#include <amg/grid.hpp>
#include <amg/interpolator.hpp>
#include <amg/multigrid.hpp>
#include <amg/pcg.hpp>
#include <amg/smoother.hpp>

static int n_checks = 0, n_failed = 0;
#define CHECK(cond) do { ++n_checks; if (!(cond)) { ++n_failed; std::cout << "FAILED " << __LINE__ << ": " #cond << std::endl; } } while (0)
#define CHECK_THROWS_AS(expr, type) do { ++n_checks; bool ok_ = false; try { expr; } catch (const type&) { ok_ = true; } catch (...) {} \
  if (!ok_) { ++n_failed; std::cout << "FAILED " << __LINE__ << ": " #expr " did not throw " #type << std::endl; } } while (0)

int main() {
  size_t n_interior_points = 2;
  Eigen::SparseMatrix<double> A = AMG::Grid<double>::laplacian(n_interior_points);
  Eigen::VectorXd b = AMG::Grid<double>::rhs(n_interior_points);
  size_t ndofs = n_interior_points * n_interior_points;
  CHECK(b.size() == (Eigen::Index)ndofs);                                  // testlib.cpp:28

  Eigen::SimplicialLDLT<Eigen::SparseMatrix<double>> direct_solver;          // :31-35
  Eigen::VectorXd exact_u(ndofs);
  direct_solver.analyzePattern(A);
  direct_solver.factorize(A);
  exact_u = direct_solver.solve(b);
  double rss = AMG::rss(A, exact_u, b);
  std::cout << "RSS: " << rss << std::endl;

  double h_from_n = AMG::Grid<double>::grid_spacing_h(n_interior_points);   // :60-62
  size_t n_from_h = AMG::Grid<double>::points_n_from_grid_spacing_h(h_from_n);
  CHECK(n_from_h == n_interior_points);

  using bad_sor = AMG::SuccessiveOverRelaxation<double>;                     // :65-71
  CHECK_THROWS_AS(bad_sor(-0.01), std::invalid_argument);
  CHECK_THROWS_AS(bad_sor(2.01), std::invalid_argument);

  size_t niters = 100;
  Eigen::VectorXd jacobi_u(ndofs);                                           // :77-81
  jacobi_u.setZero();
  AMG::Jacobi<double> jacobi(niters);
  jacobi.smooth(A, jacobi_u, b);
  CHECK(jacobi_u.isApprox(exact_u, jacobi.tolerance));

  Eigen::VectorXd sor_u(ndofs);                                              // :90-94
  sor_u.setZero();
  AMG::SuccessiveOverRelaxation<double> sor(niters);
  sor.smooth(A, sor_u, b);
  CHECK(sor_u.isApprox(exact_u, sor.tolerance));

  AMG::SparseGaussSeidel<double> spgs(niters);                               // :103-107
  Eigen::VectorXd spgs_u(ndofs);
  spgs_u.setZero();
  spgs.smooth(A, spgs_u, b);
  CHECK(spgs_u.isApprox(exact_u, spgs.tolerance));

  double tolerance = 1e-10;                                                  // :110-115
  size_t compute_error_every_n_iters = 100;
  AMG::Jacobi<double> jacobi_base(tolerance, compute_error_every_n_iters, niters);
  AMG::SuccessiveOverRelaxation<double> sor_base(tolerance, compute_error_every_n_iters, niters);

  std::cout << "Linear interpolator:" << std::endl;                          // :118-128
  size_t n_levels = 8;
  AMG::LinearInterpolator<double> linear_interpolator(n_levels);
  linear_interpolator.make_operators(7, 3, 0);
  std::cout << "nh = 7, nH = 3:" << std::endl << linear_interpolator.get_P(0) << std::endl;
  linear_interpolator.make_operators(24, 11, 0);
  CHECK(linear_interpolator.get_P(0).rows() == 24 && linear_interpolator.get_P(0).cols() == 11);

  using bad_amg = AMG::Multigrid<double>;                                    // :131-144
  size_t bad_compute_error_every_n_iters = 100;
  size_t bad_n_iters = 10;
  CHECK_THROWS_AS(bad_amg(&linear_interpolator, &spgs, A, b, n_levels, 1e-9,
                          bad_compute_error_every_n_iters, bad_n_iters), std::invalid_argument);
  Eigen::SparseMatrix<double> bad_A(10, 10);
  Eigen::VectorXd bad_b(11);
  CHECK_THROWS_AS(bad_amg(&linear_interpolator, &spgs, bad_A, bad_b, n_levels, 1e-9,
                          bad_compute_error_every_n_iters, bad_n_iters), std::invalid_argument);

  size_t n_fine_nodes = 35;                                                  // :147-159
  Eigen::SparseMatrix<double> amg_A = AMG::Grid<double>::laplacian(n_fine_nodes);
  Eigen::VectorXd amg_b = AMG::Grid<double>::rhs(n_fine_nodes);
  AMG::SparseGaussSeidel<double> amg_spgs;
  AMG::Multigrid<double> amg(&linear_interpolator, &amg_spgs, amg_A, amg_b, n_levels, 1e-9, 5, 100);

  std::cout << "Dofs at Levels in Multigrid:" << std::endl;                  // :161-181
  std::cout << amg.get_coefficient_matrix(0).rows() << std::endl;
  const size_t want_sizes[8] = {1225, 612, 305, 152, 75, 37, 18, 8};       // output.png
  for (size_t level = 1; level < n_levels; ++level) {
    auto finer_A = amg.get_coefficient_matrix(level - 1);
    auto coarser_A = amg.get_coefficient_matrix(level);
    std::cout << coarser_A.rows() << std::endl;
    auto finer_u = amg.get_soln(level - 1);
    auto coarser_u = amg.get_soln(level);
    auto finer_b = amg.get_rhs(level - 1);
    auto coarser_b = amg.get_rhs(level);
    CHECK(finer_A.size() > coarser_A.size());
    CHECK(finer_u.size() > coarser_u.size());
    CHECK(finer_b.size() > coarser_b.size());
    CHECK((size_t)coarser_A.rows() == want_sizes[level]);
  }

  std::cout << "Checking sparse gaussian solver:" << std::endl;              // :188-196
  AMG::SparseGaussSeidel<double> realistic_spgs(1e-9, 100, 1000);
  auto A_h = AMG::Grid<double>::laplacian(n_fine_nodes);
  auto rhs_h = AMG::Grid<double>::rhs(n_fine_nodes);
  Eigen::VectorXd spgs_u_h(rhs_h.rows());
  spgs_u_h.setZero();
  realistic_spgs.smooth(A_h, spgs_u_h, rhs_h);
  auto spgs_error = AMG::rss(A_h, spgs_u_h, rhs_h);
  std::cout << "SPGS error: " << spgs_error << std::endl;
  CHECK(spgs_error < realistic_spgs.tolerance);
  CHECK(std::fabs(spgs_error - 8.69692e-10) < 1e-15);                        // output.png

  std::cout << "Checking AMG solver:" << std::endl;                          // :203-212
  auto amg_u = amg.solve();
  auto amg_error = AMG::rss(A_h, amg_u, rhs_h);
  std::cout << "AMG error: " << amg_error << std::endl;
  CHECK(amg_error < amg.get_tolerance());
  CHECK(std::fabs(amg_error - 7.19199e-11) < 1e-16);                         // output.png
  CHECK(amg_u.isApprox(spgs_u_h, 1e-6));

  // build-side smoother through the same constructor
  AMG::TrueJacobi<double> tj(0.6, 2);
  AMG::LinearInterpolator<double> interp2(4);
  AMG::Multigrid<double> amg2(&interp2, &tj, amg_A, amg_b, 4, 1e-9, 5, 50);
  amg2.vcycle();
  double r1 = AMG::rss(amg_A, amg2.get_soln(0), amg_b);
  for (int i = 0; i < 10; ++i) amg2.vcycle();
  CHECK(AMG::rss(amg_A, amg2.get_soln(0), amg_b) < 0.1 * r1);

  // a user-defined SmootherBase subclass (plug-in API, smoother.hpp:18-66): runs on the
  // host between device steps.  Damped Richardson-Jacobi written against the public
  // SparseMatrix interface only; must give the TrueJacobi result (same arithmetic).
  struct UserJacobi : public AMG::SmootherBase<double> {
    double omega;
    explicit UserJacobi(double w) : omega(w) { n_iters = 2; }
    void smooth(const Eigen::SparseMatrix<double>& A, Eigen::Matrix<double, -1, 1>& u,
                const Eigen::Matrix<double, -1, 1>& b) override {
      const int* cp = A.outerIndexPtr(); const int* ri = A.innerIndexPtr(); const double* v = A.valuePtr();
      const Eigen::Index n = A.cols();
      for (size_t it = 0; it < n_iters; ++it) {
        Eigen::Matrix<double, -1, 1> t(n);
        for (Eigen::Index c = 0; c < n; ++c) {
          double rsum = 0, diag = 0;
          for (int p = cp[c]; p < cp[c + 1]; ++p) { if (ri[p] == c) diag = v[p]; else rsum += v[p] * u[ri[p]]; }
          t[c] = diag == 0 ? u[c] : u[c] + omega * ((b[c] - rsum) / diag - u[c]);
        }
        u = t;
      }
    }
  };
  {
    UserJacobi uj(0.6);
    AMG::LinearInterpolator<double> interp3(4), interp4(4);
    AMG::TrueJacobi<double> tj2(0.6, 2);
    AMG::Multigrid<double> amg_user(&interp3, &uj, amg_A, amg_b, 4, 1e-9, 5, 50);
    AMG::Multigrid<double> amg_dev(&interp4, &tj2, amg_A, amg_b, 4, 1e-9, 5, 50);
    for (int i = 0; i < 3; ++i) { amg_user.vcycle(); amg_dev.vcycle(); }
    auto ua = amg_user.get_soln(0), ub = amg_dev.get_soln(0);
    bool same = ua.size() == ub.size();
    for (Eigen::Index i = 0; same && i < ua.size(); ++i) same = (ua[i] == ub[i]);
    CHECK(same);
  }

  // the V-cycle as a preconditioner (reference README.md:127): AMG::PCG on the same problem
  {
    AMG::LinearInterpolator<double> interp5(8);
    AMG::SparseGaussSeidel<double> pre;
    AMG::Multigrid<double> amg_pc(&interp5, &pre, amg_A, amg_b, n_levels, 1e-9, 5, 100);
    AMG::PCG<double> pcg(&amg_pc, 1e-10, 50);
    auto x = pcg.solve();
    CHECK(pcg.relative_residual() <= 1e-10);
    CHECK(pcg.iterations() < 35);                       // plain V-cycles needed 35
    CHECK(x.isApprox(amg_u, 1e-6));
    UserJacobi uj2(0.6);
    AMG::LinearInterpolator<double> interp6(8);
    AMG::Multigrid<double> amg_host(&interp6, &uj2, amg_A, amg_b, 4, 1e-9, 5, 50);
    auto bad_pcg = [&]() { AMG::PCG<double> p2(&amg_host); };
    CHECK_THROWS_AS(bad_pcg(), std::invalid_argument);
  }

  // strength-based C/F coarsening (reference README.md:104-109: considered, not built): the
  // same Multigrid object on a Ruge-Stueben hierarchy needs far fewer cycles than the 35 above
  {
    AMG::RugeStuebenInterpolator<double> rs(12, 0.25, 20);
    AMG::SparseGaussSeidel<double> sm;
    AMG::Multigrid<double> amg_rs(&rs, &sm, amg_A, amg_b, 12, 1e-9, 1, 100);
    CHECK(amg_rs.get_n_levels() >= 3 && amg_rs.get_n_levels() <= 12);
    CHECK(amg_rs.get_n_dofs(1) < amg_rs.get_n_dofs(0) && amg_rs.get_n_dofs(1) >= amg_rs.get_n_dofs(0) / 2 - 1);
    CHECK((size_t)rs.get_P(0).rows() == amg_rs.get_n_dofs(0) && (size_t)rs.get_P(0).cols() == amg_rs.get_n_dofs(1));
    CHECK((size_t)amg_rs.get_coefficient_matrix(1).rows() == amg_rs.get_n_dofs(1));
    size_t cycles = 0;
    double err = AMG::rss(amg_A, amg_rs.get_soln(0), amg_b);
    while (err > 1e-9 && cycles < 35) { amg_rs.vcycle(); ++cycles; err = AMG::rss(amg_A, amg_rs.get_soln(0), amg_b); }
    CHECK(err <= 1e-9);
    CHECK(cycles <= 12);
    CHECK(amg_rs.get_soln(0).isApprox(amg_u, 1e-6));
    auto direct = [&]() { rs.make_operators(10, 4, 0); };
    CHECK_THROWS_AS(direct(), std::logic_error);
  }

  std::cout << (n_failed ? "SOME TESTS FAILED" : "All tests passed") << " (" << n_checks
            << " assertions)" << std::endl;
  return n_failed ? 1 : 0;
}
