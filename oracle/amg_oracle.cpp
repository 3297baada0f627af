// =============================================================================
// oracle/amg_oracle.cpp  --  TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the V-cycle hot path of jfdev001/algebraic-multigrid
// (header-only C++17 on Eigen 3.4.0; reference mounted at /root/reference).
// It is the *checker* for the HIP path and the "port" CPU baseline of bench.py.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// it.  The product (algebraic-multigrid_amd/) never includes, links or calls
// anything in this directory.
//
// Parity pin: the reference cannot be compiled here (needs Eigen 3.4 + Catch2,
// both fetched from the network by its CMake; neither is in the image), and its
// tests hold no golden vectors.  The oracle is pinned by the only known-answer
// values the reference publishes (image/README/output.png, produced by
// test/testlib.cpp:166-170,188-195,203-205):
//   * level sizes 1225,612,305,152,75,37,18,8
//   * AMG: "converged after 35 iterations", rss 7.19199e-11
//   * SPGS: "converged after 900 iterations", rss 8.69692e-10
// and by SURVEY.md KAT-4 (config 1 trajectory).  tests/test_oracle_kat.py checks
// all of them.  Third-party arithmetic (Eigen 3.4.0 SpMV / SpGEMM / LinSpaced /
// kroneckerProduct order of operations) is restated from Eigen's published
// algorithm at the reference's call sites, cited per function below.
// The coarse direct solve (Eigen::SimplicialLDLT, AMD ordering) is replaced by
// an un-permuted banded LDL^T: equal to Eigen's only to rounding (~1e-13 rel).
//
// Build: g++ -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
// All arithmetic is separate IEEE multiply / add / divide, no FMA
// (reference default build: no optimisation flags, plain x86-64).
// =============================================================================
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

typedef int32_t idx_t;

struct Csc {  // Eigen::SparseMatrix<double> default = column-major, int indices
  int64_t rows = 0, cols = 0;
  std::vector<idx_t> colptr;  // cols+1
  std::vector<idx_t> rowind;  // nnz, ascending within a column
  std::vector<double> val;
  int64_t nnz() const { return (int64_t)rowind.size(); }
};

// ---------------------------------------------------------------- grid.hpp ---
// grid.hpp:31
double grid_spacing_h(size_t n) { return 2.0 / (n + 1); }
// grid.hpp:39-41
size_t points_n_from_grid_spacing_h(double h) {
  return static_cast<size_t>((2 / h) - 1);
}

// grid.hpp:50-75 (second_order_central_difference) + :88-98 (laplacian).
// D = tridiag(1,-2,1); D = D / (h*h)  (each stored value divided, :72);
// A = kron(I,D) + kron(D,I).  Dof = j*n + i (i fast).  The diagonal is the sum
// D_ii + D_jj of two equal values (exact doubling); every off-diagonal comes
// from exactly one of the two Kronecker terms (times 1.0, exact).
Csc laplacian(size_t n) {
  const double h = grid_spacing_h(n);
  const double hh = h * h;
  const double dm = -2.0 / hh;  // D diagonal
  const double dp = 1.0 / hh;   // D off-diagonal
  Csc A;
  A.rows = A.cols = (int64_t)(n * n);
  A.colptr.assign(n * n + 1, 0);
  A.rowind.reserve(5 * n * n);
  A.val.reserve(5 * n * n);
  for (size_t j = 0; j < n; ++j) {
    for (size_t i = 0; i < n; ++i) {
      const size_t c = j * n + i;
      // column c of A (A symmetric): rows c-n, c-1, c, c+1, c+n when present
      if (j > 0) { A.rowind.push_back((idx_t)(c - n)); A.val.push_back(dp * 1.0); }
      if (i > 0) { A.rowind.push_back((idx_t)(c - 1)); A.val.push_back(1.0 * dp); }
      A.rowind.push_back((idx_t)c); A.val.push_back(1.0 * dm + dm * 1.0);
      if (i + 1 < n) { A.rowind.push_back((idx_t)(c + 1)); A.val.push_back(1.0 * dp); }
      if (j + 1 < n) { A.rowind.push_back((idx_t)(c + n)); A.val.push_back(dp * 1.0); }
      A.colptr[c + 1] = (idx_t)A.rowind.size();
    }
  }
  return A;
}

// Build-side extension (no reference counterpart, SURVEY 8(c) last row):
// 3-D 7-point A = I(x)I(x)D + I(x)D(x)I + D(x)I(x)I, same D and h.
Csc laplacian3d(size_t n) {
  const double h = grid_spacing_h(n);
  const double hh = h * h;
  const double dm = -2.0 / hh, dp = 1.0 / hh;
  const size_t n2 = n * n, N = n2 * n;
  Csc A;
  A.rows = A.cols = (int64_t)N;
  A.colptr.assign(N + 1, 0);
  A.rowind.reserve(7 * N);
  A.val.reserve(7 * N);
  for (size_t k = 0; k < n; ++k)
    for (size_t j = 0; j < n; ++j)
      for (size_t i = 0; i < n; ++i) {
        const size_t c = k * n2 + j * n + i;
        if (k > 0) { A.rowind.push_back((idx_t)(c - n2)); A.val.push_back(dp); }
        if (j > 0) { A.rowind.push_back((idx_t)(c - n)); A.val.push_back(dp); }
        if (i > 0) { A.rowind.push_back((idx_t)(c - 1)); A.val.push_back(dp); }
        A.rowind.push_back((idx_t)c); A.val.push_back((dm + dm) + dm);
        if (i + 1 < n) { A.rowind.push_back((idx_t)(c + 1)); A.val.push_back(dp); }
        if (j + 1 < n) { A.rowind.push_back((idx_t)(c + n)); A.val.push_back(dp); }
        if (k + 1 < n) { A.rowind.push_back((idx_t)(c + n2)); A.val.push_back(dp); }
        A.colptr[c + 1] = (idx_t)A.rowind.size();
      }
  return A;
}

// Eigen 3.4.0 DenseBase::LinSpaced(size, low, high) for floating point
// (linspaced_op_impl, |high| < |low| is false here so no flip):
//   step = (high-low)/(size-1);  x_i = (i == size-1) ? high : low + i*step.
// grid.hpp:118-120.
double linspaced(size_t size, double low, double high, size_t i) {
  const size_t size1 = size == 1 ? 1 : size - 1;
  const double step = size == 1 ? 0.0 : (high - low) / (double)(size - 1);
  return (i == size1) ? high : low + (double)i * step;
}

// grid.hpp:108-140 with the default forcing (:110-112).
void rhs(size_t n, double* b) {
  const size_t np = n + 2;
  size_t dof = 0;
  for (size_t j = 1; j <= n; ++j) {
    const double xj = linspaced(np, -1.0, 1.0, j);
    for (size_t i = 1; i <= n; ++i) {
      const double xi = linspaced(np, -1.0, 1.0, i);
      b[dof++] = 5 * exp(-10 * (xj * xj + xi * xi));
    }
  }
}

void rhs3d(size_t n, double* b) {  // build-side analogue for laplacian3d
  const size_t np = n + 2;
  size_t dof = 0;
  for (size_t k = 1; k <= n; ++k) {
    const double xk = linspaced(np, -1.0, 1.0, k);
    for (size_t j = 1; j <= n; ++j) {
      const double xj = linspaced(np, -1.0, 1.0, j);
      for (size_t i = 1; i <= n; ++i) {
        const double xi = linspaced(np, -1.0, 1.0, i);
        b[dof++] = 5 * exp(-10 * ((xk * xk + xj * xj) + xi * xi));
      }
    }
  }
}

// -------------------------------------------------------- interpolator.hpp ---
// interpolator.hpp:106-129: P (n_h x n_H), column j holds 0.5,1.0,0.5 on rows
// 2j,2j+1,2j+2 (each guarded by "< n_h").
Csc make_P(size_t n_h, size_t n_H) {
  Csc P;
  P.rows = (int64_t)n_h;
  P.cols = (int64_t)n_H;
  P.colptr.assign(n_H + 1, 0);
  size_t i = 0;
  for (size_t j = 0; j < n_H; ++j) {
    if (i < n_h) { P.rowind.push_back((idx_t)i); P.val.push_back(0.5); }
    if (i + 1 < n_h) { P.rowind.push_back((idx_t)(i + 1)); P.val.push_back(1.0); }
    if (i + 2 < n_h) { P.rowind.push_back((idx_t)(i + 2)); P.val.push_back(0.5); }
    i += 2;
    P.colptr[j + 1] = (idx_t)P.rowind.size();
  }
  return P;
}

// interpolator.hpp:132-134: R = P.transpose() (CSC of the transpose; inner
// indices come out ascending).
Csc transpose(const Csc& M) {
  Csc T;
  T.rows = M.cols;
  T.cols = M.rows;
  T.colptr.assign(T.cols + 1, 0);
  T.rowind.resize(M.nnz());
  T.val.resize(M.nnz());
  for (int64_t p = 0; p < M.nnz(); ++p) T.colptr[M.rowind[p] + 1]++;
  for (int64_t c = 0; c < T.cols; ++c) T.colptr[c + 1] += T.colptr[c];
  std::vector<idx_t> next(T.colptr.begin(), T.colptr.end() - 1);
  for (int64_t c = 0; c < M.cols; ++c)
    for (idx_t p = M.colptr[c]; p < M.colptr[c + 1]; ++p) {
      const idx_t q = next[M.rowind[p]]++;
      T.rowind[q] = (idx_t)c;
      T.val[q] = M.val[p];
    }
  return T;
}

// ------------------------------------------------------------ Eigen SpGEMM ---
// multigrid.hpp:222  A_H = R_h * (A_h * P_h).  Eigen 3.4.0
// conservative_sparse_sparse_product_impl (col-major x col-major): for each
// result column j, for each nonzero (k, y) of rhs column j in storage order,
// for each nonzero (i, x) of lhs column k: first touch values[i] = x*y, later
// values[i] += x*y.  Structural entries whose sum is exactly 0.0 are KEPT
// (SURVEY F5).  Result columns are sorted ascending afterwards.
Csc spgemm(const Csc& L, const Csc& Rm) {
  Csc C;
  C.rows = L.rows;
  C.cols = Rm.cols;
  C.colptr.assign(C.cols + 1, 0);
  std::vector<char> mask(L.rows, 0);
  std::vector<double> values(L.rows, 0.0);
  std::vector<idx_t> indices;
  for (int64_t j = 0; j < Rm.cols; ++j) {
    indices.clear();
    for (idx_t q = Rm.colptr[j]; q < Rm.colptr[j + 1]; ++q) {
      const double y = Rm.val[q];
      const idx_t k = Rm.rowind[q];
      for (idx_t p = L.colptr[k]; p < L.colptr[k + 1]; ++p) {
        const idx_t i = L.rowind[p];
        const double x = L.val[p];
        if (!mask[i]) {
          mask[i] = 1;
          values[i] = x * y;
          indices.push_back(i);
        } else {
          values[i] += x * y;
        }
      }
    }
    // sort indices ascending (insertion into the sorted result)
    for (size_t a = 1; a < indices.size(); ++a) {
      idx_t v = indices[a];
      size_t b = a;
      while (b > 0 && indices[b - 1] > v) { indices[b] = indices[b - 1]; --b; }
      indices[b] = v;
    }
    for (idx_t i : indices) {
      C.rowind.push_back(i);
      C.val.push_back(values[i]);
      mask[i] = 0;
    }
    C.colptr[j + 1] = (idx_t)C.rowind.size();
  }
  return C;
}

// ------------------------------------------------------------- Eigen SpMV ----
// Eigen sparse_time_dense_product_impl<..., ColMajor>: for each column j:
// rhs_j = alpha * x[j]; for each (i, a) in column j: res[i] += a * rhs_j.

// multigrid.hpp:272-274 (also :204, :236):  r = f - A*u  is evaluated as
// r = f; r += (-1) * A * u  => per row ((f_i - a_i1 u_1) - a_i2 u_2) ... in
// ascending column order.
void residual(const Csc& A, const double* u, const double* f, double* r) {
  for (int64_t i = 0; i < A.rows; ++i) r[i] = f[i];
  for (int64_t j = 0; j < A.cols; ++j) {
    const double rhs_j = -1.0 * u[j];
    for (idx_t p = A.colptr[j]; p < A.colptr[j + 1]; ++p)
      r[A.rowind[p]] += A.val[p] * rhs_j;
  }
}

// interpolator.hpp:52-56 / :64-68:  result = M * v  (dst = 0; dst += 1*M*v).
void spmv(const Csc& M, const double* v, double* out) {
  for (int64_t i = 0; i < M.rows; ++i) out[i] = 0.0;
  for (int64_t j = 0; j < M.cols; ++j) {
    const double rhs_j = 1.0 * v[j];
    for (idx_t p = M.colptr[j]; p < M.colptr[j + 1]; ++p)
      out[M.rowind[p]] += M.val[p] * rhs_j;
  }
}

// common.hpp:17-27.  bhat = A*u (one evaluation is enough: the O(N nnz)
// re-evaluation of the reference, SURVEY F10, changes cost not values);
// error += (b_i - bhat_i)*(b_i - bhat_i) sequentially in i.
double rss(const Csc& A, const double* u, const double* b) {
  std::vector<double> bhat(A.rows);
  spmv(A, u, bhat.data());
  double error = 0.0;
  for (int64_t i = 0; i < A.rows; ++i)
    error += (b[i] - bhat[i]) * (b[i] - bhat[i]);
  return error;
}

// ------------------------------------------------------------ smoother.hpp ---
// smoother.hpp:101-117 + :129-138.  Walks COLUMN col of the CSC matrix as if it
// were the row (assumes A = A^T).
inline void spgs_update(const Csc& A, const double* b, double* u, idx_t col) {
  const double z = 0;
  double rsum = z, diag = z;
  for (idx_t p = A.colptr[col]; p < A.colptr[col + 1]; ++p) {
    const idx_t row = A.rowind[p];
    const double val = A.val[p];
    diag = (col == row) ? val : diag;
    rsum += (col == row) ? z : val * u[row];
  }
  u[col] = diag == z ? u[col] : (b[col] - rsum) / diag;
}
// smoother.hpp:148-157
void spgs_forward(const Csc& A, const double* b, double* u) {
  for (idx_t c = 0; c < (idx_t)A.cols; ++c) spgs_update(A, b, u, c);
}
// smoother.hpp:167-174
void spgs_backward(const Csc& A, const double* b, double* u) {
  for (idx_t c = (idx_t)A.cols - 1; c >= 0; --c) spgs_update(A, b, u, c);
}

// smoother.hpp:189-215.  Returns iterations done; *converged as the reference
// would print it (only meaningful when every != 0).
size_t spgs_smooth(const Csc& A, double* u, const double* b, double tol,
                   size_t every, size_t n_iters, int* converged) {
  size_t iter = 0;
  double error = 100;
  while (iter < n_iters && error > tol) {
    spgs_forward(A, b, u);
    spgs_backward(A, b, u);
    iter += 1;
    if (every != 0 && iter % every == 0) error = rss(A, u, b);
  }
  if (converged) *converged = (error <= tol);
  return iter;
}

// CSR view of a CSC matrix (row-major gather order) for the two dense-loop
// smoothers, which address A.coeff(i, j) by ROW i.
struct Csr {
  std::vector<idx_t> rowptr, col;
  std::vector<double> val;
};
Csr to_csr(const Csc& A) {
  Csc T = transpose(A);
  Csr R;
  R.rowptr = T.colptr;
  R.col = T.rowind;
  R.val = T.val;
  return R;
}

// smoother.hpp:239-263 (AMG::Jacobi, which is an in-place forward Gauss-Seidel,
// SURVEY F6).  sigma sums A.coeff(i,j)*u[j] over ALL j != i ascending; absent
// entries contribute 0.0*u[j] = +-0 which leaves sigma unchanged for finite u,
// so the sparse ascending sum is the same number.
size_t refjacobi_smooth(const Csc& A, double* u, const double* b, double tol,
                        size_t every, size_t n_iters) {
  const Csr R = to_csr(A);
  const size_t ndofs = (size_t)A.rows;
  size_t iter = 0;
  double error = 100;
  while (iter < n_iters && error > tol) {
    for (size_t i = 0; i < ndofs; ++i) {
      double sigma = 0, aii = 0;
      for (idx_t p = R.rowptr[i]; p < R.rowptr[i + 1]; ++p) {
        if ((size_t)R.col[p] != i) sigma += R.val[p] * u[R.col[p]];
        else aii = R.val[p];
      }
      u[i] = (b[i] - sigma) / aii;
    }
    iter += 1;
    if (every != 0 && iter % every == 0) error = rss(A, u, b);
  }
  return iter;
}

// smoother.hpp:339-372 (SuccessiveOverRelaxation): two partial sums j<i and
// j>i, then (b - s1 - s2)/aii, then uk + omega*(gs - uk).
size_t sor_smooth(const Csc& A, double* u, const double* b, double omega,
                  double tol, size_t every, size_t n_iters) {
  const Csr R = to_csr(A);
  const size_t ndofs = (size_t)A.rows;
  size_t iter = 0;
  double error = 100;
  while (iter < n_iters && error > tol) {
    for (size_t i = 0; i < ndofs; ++i) {
      double s_less = 0, s_greater = 0, aii = 0;
      for (idx_t p = R.rowptr[i]; p < R.rowptr[i + 1]; ++p) {
        const size_t j = (size_t)R.col[p];
        if (j < i) s_less += R.val[p] * u[j];
        else if (j > i) s_greater += R.val[p] * u[j];
        else aii = R.val[p];
      }
      const double gs = (b[i] - s_less - s_greater) / aii;
      const double uk = u[i];
      u[i] = uk + omega * (gs - uk);
    }
    iter += 1;
    if (every != 0 && iter % every == 0) error = rss(A, u, b);
  }
  return iter;
}

// Build-side twins (no reference counterpart): the per-row
// arithmetic is the SpGS row (column-as-row walk, ascending, diagonal skipped,
// IEEE divide), then the SOR blend  u + omega*(g - u).
// True (two-buffer) weighted Jacobi, `sweeps` passes; u is updated in place at
// the end of every pass.
void true_jacobi(const Csc& A, double* u, const double* b, double omega,
                 size_t sweeps) {
  std::vector<double> tmp(A.cols);
  for (size_t s = 0; s < sweeps; ++s) {
    for (idx_t c = 0; c < (idx_t)A.cols; ++c) {
      double rsum = 0, diag = 0;
      for (idx_t p = A.colptr[c]; p < A.colptr[c + 1]; ++p) {
        const idx_t row = A.rowind[p];
        if (row == c) diag = A.val[p];
        else rsum += A.val[p] * u[row];
      }
      const double uk = u[c];
      tmp[c] = diag == 0 ? uk : uk + omega * ((b[c] - rsum) / diag - uk);
    }
    std::memcpy(u, tmp.data(), sizeof(double) * A.cols);
  }
}

// Multicolour symmetric Gauss-Seidel: colours 0..nc-1 ascending then nc-1..0
// descending = one iteration.  Rows of one colour are mutually independent, so
// the result does not depend on the order inside a colour.
void multicolor_gs(const Csc& A, double* u, const double* b, const idx_t* color,
                   idx_t n_colors, size_t n_iters) {
  std::vector<std::vector<idx_t>> rows_of(n_colors);
  for (idx_t c = 0; c < (idx_t)A.cols; ++c) rows_of[color[c]].push_back(c);
  for (size_t it = 0; it < n_iters; ++it) {
    for (idx_t k = 0; k < n_colors; ++k)
      for (idx_t c : rows_of[k]) spgs_update(A, b, u, c);
    for (idx_t k = n_colors - 1; k >= 0; --k)
      for (idx_t c : rows_of[k]) spgs_update(A, b, u, c);
  }
}

// First-fit colouring in row order; neighbours = numerically non-zero off-diagonal entries
// of column k and of row k (so that a structurally non-symmetric A is handled too).
void greedy_colors(const Csc& A, std::vector<idx_t>* color, idx_t* n_colors) {
  const Csc T = transpose(A);
  const int64_t n = A.cols;
  color->assign((size_t)n, -1);
  idx_t nc = 0;
  std::vector<int64_t> mark;
  for (int64_t k = 0; k < n; ++k) {
    for (const Csc* S : {&A, &T})
      for (idx_t p = S->colptr[k]; p < S->colptr[k + 1]; ++p) {
        const idx_t j = S->rowind[p];
        if (j == k || S->val[p] == 0.0) continue;
        const idx_t c = (*color)[j];
        if (c >= 0) mark[c] = k;
      }
    idx_t c = 0;
    while (c < nc && mark[c] == k) ++c;
    if (c == nc) {
      ++nc;
      mark.push_back(-1);
    }
    (*color)[k] = c;
  }
  *n_colors = nc;
}

// --------------------------------------------------- coarse direct solver ----
// multigrid.hpp:33,240-243,287-288 use Eigen::SimplicialLDLT (AMD ordering).
// Restated as an un-permuted banded LDL^T (A_L is symmetric negative definite,
// so no pivoting is needed).  band[i*(w+1)+d] = L[i,i-d] (d>=1), D[i] at d=0.
struct BandLDL {
  int64_t n = 0, w = 0;
  std::vector<double> band;
};
BandLDL band_factor(const Csc& A) {
  BandLDL F;
  F.n = A.rows;
  int64_t w = 0;
  for (int64_t c = 0; c < A.cols; ++c)
    for (idx_t p = A.colptr[c]; p < A.colptr[c + 1]; ++p) {
      const int64_t d = (int64_t)A.rowind[p] - c;
      if (d > w) w = d;
      if (-d > w) w = -d;
    }
  F.w = w;
  const int64_t W = w + 1;
  F.band.assign((size_t)(F.n * W), 0.0);
  // load the lower triangle (row i, column c <= i) from column c of the CSC
  for (int64_t c = 0; c < A.cols; ++c)
    for (idx_t p = A.colptr[c]; p < A.colptr[c + 1]; ++p) {
      const int64_t i = A.rowind[p];
      if (i >= c) F.band[i * W + (i - c)] = A.val[p];
    }
  std::vector<double> v(W);
  for (int64_t i = 0; i < F.n; ++i) {
    const int64_t j0 = i - w > 0 ? i - w : 0;
    for (int64_t j = j0; j < i; ++j) {
      // s = A[i,j] - sum_{k=max(j0, j-w)}^{j-1} (L[i,k] D[k]) L[j,k]
      double s = F.band[i * W + (i - j)];
      const int64_t k0 = (j - w > j0) ? j - w : j0;
      for (int64_t k = k0; k < j; ++k)
        s -= v[k - j0] * F.band[j * W + (j - k)];
      v[j - j0] = s;                                // L[i,j]*D[j]
      F.band[i * W + (i - j)] = s / F.band[j * W];  // L[i,j]
    }
    double d = F.band[i * W];
    for (int64_t k = j0; k < i; ++k) d -= v[k - j0] * F.band[i * W + (i - k)];
    F.band[i * W] = d;
  }
  return F;
}
// Forward: y_i = ((f_i - L[i,i-w] y_{i-w}) - ...) - L[i,i-1] y_{i-1}
// Diagonal: z_i = y_i / D_i
// Backward: x_i = ((z_i - L[i+w,i] x_{i+w}) - ...) - L[i+1,i] x_{i+1}
void band_solve(const BandLDL& F, const double* f, double* x) {
  const int64_t n = F.n, w = F.w, W = w + 1;
  for (int64_t i = 0; i < n; ++i) {
    double s = f[i];
    const int64_t k0 = i - w > 0 ? i - w : 0;
    for (int64_t k = k0; k < i; ++k) s -= F.band[i * W + (i - k)] * x[k];
    x[i] = s;
  }
  for (int64_t i = 0; i < n; ++i) x[i] = x[i] / F.band[i * W];
  for (int64_t i = n - 1; i >= 0; --i) {
    double s = x[i];
    const int64_t k1 = i + w < n - 1 ? i + w : n - 1;
    for (int64_t k = k1; k > i; --k) s -= F.band[k * W + (k - i)] * x[k];
    x[i] = s;
  }
}

// ----------------------------------------------------------- multigrid.hpp ---
enum Smoother {
  SM_SPGS = 0,        // smoother.hpp:86-216  (symmetric lexicographic GS)
  SM_REF_JACOBI = 1,  // smoother.hpp:223-264 (forward GS, dense loops)
  SM_SOR = 2,         // smoother.hpp:271-373
  SM_TRUE_JACOBI = 3, // build-side twin
  SM_MULTICOLOR = 4   // build-side twin
};

struct Multigrid {
  size_t n_levels = 0;
  std::vector<Csc> A, P, R;
  std::vector<std::vector<double>> u, f, r;
  std::vector<std::vector<idx_t>> color;  // per level, only for SM_MULTICOLOR
  std::vector<idx_t> n_colors;
  BandLDL coarse;
  int smoother = SM_SPGS;
  size_t sm_iters = 1;  // SmootherBase::n_iters (SparseGaussSeidel() => 1)
  double omega = 1.0;
  std::string err;
};

// multigrid.hpp:127-130
size_t n_H_dofs_from_n_h_dofs(size_t h_dofs) { return (h_dofs + 1) / 2 - 1; }

// multigrid.hpp:181-244 with LinearInterpolator (interpolator.hpp:106-141).
// custom_P: transfer operators handed over (a user InterpolatorBase, interpolator.hpp:43-44,
// or the strength-based coarsening twin in oracle.py); else LinearInterpolator's.
Multigrid* mg_create(const Csc& A0, const double* b, size_t n_levels,
                     const std::vector<Csc>* custom_P = nullptr) {
  Multigrid* M = new Multigrid;
  M->n_levels = n_levels;
  M->A.resize(n_levels);
  M->P.resize(n_levels > 0 ? n_levels - 1 : 0);
  M->R.resize(n_levels > 0 ? n_levels - 1 : 0);
  M->u.resize(n_levels);
  M->f.resize(n_levels);
  M->r.resize(n_levels);
  M->color.resize(n_levels);
  M->n_colors.assign(n_levels, 0);
  M->A[0] = A0;
  const size_t n0 = (size_t)A0.rows;
  M->u[0].assign(n0, 0.0);
  M->f[0].assign(b, b + n0);
  M->r[0].assign(n0, 0.0);
  residual(A0, M->u[0].data(), M->f[0].data(), M->r[0].data());
  for (size_t l = 1; l < n_levels; ++l) {
    const size_t n_h = (size_t)M->A[l - 1].rows;
    const size_t n_H = custom_P ? (size_t)(*custom_P)[l - 1].cols : n_H_dofs_from_n_h_dofs(n_h);
    M->P[l - 1] = custom_P ? (*custom_P)[l - 1] : make_P(n_h, n_H);
    M->R[l - 1] = transpose(M->P[l - 1]);
    Csc AP = spgemm(M->A[l - 1], M->P[l - 1]);
    M->A[l] = spgemm(M->R[l - 1], AP);
    M->u[l].assign(n_H, 0.0);
    M->f[l].assign(n_H, 0.0);
    M->r[l].assign(n_H, 0.0);
  }
  M->coarse = band_factor(M->A[n_levels - 1]);
  return M;
}

void mg_smooth(Multigrid* M, size_t l) {
  const Csc& A = M->A[l];
  double* u = M->u[l].data();
  const double* f = M->f[l].data();
  switch (M->smoother) {
    case SM_SPGS: spgs_smooth(A, u, f, 1e-9, 0, M->sm_iters, nullptr); break;
    case SM_REF_JACOBI: refjacobi_smooth(A, u, f, 1e-9, 0, M->sm_iters); break;
    case SM_SOR: sor_smooth(A, u, f, M->omega, 1e-9, 0, M->sm_iters); break;
    case SM_TRUE_JACOBI: true_jacobi(A, u, f, M->omega, M->sm_iters); break;
    case SM_MULTICOLOR:
      // no colouring handed over (orc_mg_set_colors): colour greedily here -- first fit in
      // row order over the numerically non-zero pattern of A and A^T
      if ((int64_t)M->color[l].size() != A.rows) greedy_colors(A, &M->color[l], &M->n_colors[l]);
      multicolor_gs(A, u, f, M->color[l].data(), M->n_colors[l], M->sm_iters);
      break;
  }
}

// multigrid.hpp:263-305
void mg_vcycle(Multigrid* M) {
  const size_t L = M->n_levels;
  for (size_t l = 0; l < L; ++l) {
    mg_smooth(M, l);                                              // :268
    residual(M->A[l], M->u[l].data(), M->f[l].data(), M->r[l].data());  // :272
    if (l + 1 != L) {
      std::fill(M->u[l + 1].begin(), M->u[l + 1].end(), 0.0);     // :278
      spmv(M->R[l], M->r[l].data(), M->f[l + 1].data());          // :281
    }
  }
  {                                                               // :287-288
    std::vector<double> x(M->u[L - 1].size());
    band_solve(M->coarse, M->f[L - 1].data(), x.data());
    M->u[L - 1] = x;
  }
  for (int l = (int)L - 2; l >= 0; --l) {                         // :291
    std::vector<double> t(M->u[l].size());
    spmv(M->P[l], M->u[l + 1].data(), t.data());                  // :296
    for (size_t i = 0; i < t.size(); ++i) M->u[l][i] = M->u[l][i] + t[i];  // :294
    mg_smooth(M, (size_t)l);                                      // :300
  }
}

// multigrid.hpp:311-337.  Returns iterations; *converged = (error <= tol);
// *last_rss = the last computed error (100 if never computed).
size_t mg_solve(Multigrid* M, double tol, size_t every, size_t n_iters,
                int* converged, double* last_rss, double* trajectory,
                size_t traj_cap) {
  size_t iter = 0, nt = 0;
  double error = 100;
  while (iter < n_iters && error > tol) {
    mg_vcycle(M);
    iter += 1;
    if ((iter % every) == 0) {
      error = rss(M->A[0], M->u[0].data(), M->f[0].data());
      if (trajectory && nt < traj_cap) trajectory[nt++] = error;
    }
  }
  if (converged) *converged = (error <= tol);
  if (last_rss) *last_rss = error;
  return iter;
}

Csc csc_from_raw(int64_t rows, int64_t cols, const idx_t* colptr,
                 const idx_t* rowind, const double* val) {
  Csc A;
  A.rows = rows;
  A.cols = cols;
  A.colptr.assign(colptr, colptr + cols + 1);
  const int64_t nnz = colptr[cols];
  A.rowind.assign(rowind, rowind + nnz);
  A.val.assign(val, val + nnz);
  return A;
}

void csc_to_raw(const Csc& A, idx_t* colptr, idx_t* rowind, double* val) {
  if (colptr) std::memcpy(colptr, A.colptr.data(), sizeof(idx_t) * A.colptr.size());
  if (rowind) std::memcpy(rowind, A.rowind.data(), sizeof(idx_t) * A.rowind.size());
  if (val) std::memcpy(val, A.val.data(), sizeof(double) * A.val.size());
}

}  // namespace

// =============================================================================
// C interface for ctypes (tests / smoke / bench cpu_baseline only)
// =============================================================================
extern "C" {

double orc_grid_spacing_h(uint64_t n) { return grid_spacing_h(n); }
uint64_t orc_points_n_from_grid_spacing_h(double h) {
  return points_n_from_grid_spacing_h(h);
}

// dim = 2 or 3.  Returns nnz; arrays may be NULL to query sizes.
int64_t orc_laplacian(int dim, uint64_t n, int32_t* colptr, int32_t* rowind,
                      double* val) {
  Csc A = dim == 3 ? laplacian3d(n) : laplacian(n);
  csc_to_raw(A, colptr, rowind, val);
  return A.nnz();
}
void orc_rhs(int dim, uint64_t n, double* b) {
  if (dim == 3) rhs3d(n, b);
  else rhs(n, b);
}

uint64_t orc_n_H_from_n_h(uint64_t n_h) { return n_H_dofs_from_n_h_dofs(n_h); }

int64_t orc_make_P(uint64_t n_h, uint64_t n_H, int32_t* colptr, int32_t* rowind,
                   double* val) {
  Csc P = make_P(n_h, n_H);
  csc_to_raw(P, colptr, rowind, val);
  return P.nnz();
}

// out arrays sized by the caller (nnz of the transpose = nnz of the input)
void orc_transpose(int64_t rows, int64_t cols, const int32_t* colptr,
                   const int32_t* rowind, const double* val, int32_t* t_colptr,
                   int32_t* t_rowind, double* t_val) {
  Csc T = transpose(csc_from_raw(rows, cols, colptr, rowind, val));
  csc_to_raw(T, t_colptr, t_rowind, t_val);
}

void orc_residual(int64_t n, const int32_t* colptr, const int32_t* rowind,
                  const double* val, const double* u, const double* f, double* r) {
  residual(csc_from_raw(n, n, colptr, rowind, val), u, f, r);
}
void orc_spmv(int64_t rows, int64_t cols, const int32_t* colptr,
              const int32_t* rowind, const double* val, const double* v,
              double* out) {
  spmv(csc_from_raw(rows, cols, colptr, rowind, val), v, out);
}
double orc_rss(int64_t n, const int32_t* colptr, const int32_t* rowind,
               const double* val, const double* u, const double* b) {
  return rss(csc_from_raw(n, n, colptr, rowind, val), u, b);
}

// kind: see enum Smoother.  color/n_colors only for kind 4.  Returns iters.
uint64_t orc_smooth(int kind, int64_t n, const int32_t* colptr,
                    const int32_t* rowind, const double* val, double* u,
                    const double* b, double omega, double tol, uint64_t every,
                    uint64_t n_iters, const int32_t* color, int32_t n_colors,
                    int* converged) {
  Csc A = csc_from_raw(n, n, colptr, rowind, val);
  if (converged) *converged = 0;
  switch (kind) {
    case SM_SPGS: return spgs_smooth(A, u, b, tol, every, n_iters, converged);
    case SM_REF_JACOBI: return refjacobi_smooth(A, u, b, tol, every, n_iters);
    case SM_SOR: return sor_smooth(A, u, b, omega, tol, every, n_iters);
    case SM_TRUE_JACOBI: true_jacobi(A, u, b, omega, n_iters); return n_iters;
    case SM_MULTICOLOR:
      multicolor_gs(A, u, b, color, n_colors, n_iters);
      return n_iters;
  }
  return 0;
}
// one forward (dir=+1) or backward (dir=-1) lexicographic sweep
void orc_spgs_sweep(int dir, int64_t n, const int32_t* colptr,
                    const int32_t* rowind, const double* val, double* u,
                    const double* b) {
  Csc A = csc_from_raw(n, n, colptr, rowind, val);
  if (dir > 0) spgs_forward(A, b, u);
  else spgs_backward(A, b, u);
}

// banded LDL^T solve of a symmetric system (test helper + coarse-solve twin)
int64_t orc_band_solve(int64_t n, const int32_t* colptr, const int32_t* rowind,
                       const double* val, const double* f, double* x) {
  BandLDL F = band_factor(csc_from_raw(n, n, colptr, rowind, val));
  band_solve(F, f, x);
  return F.w;
}

void* orc_mg_create(int64_t n, const int32_t* colptr, const int32_t* rowind,
                    const double* val, const double* b, uint64_t n_levels) {
  if (n_levels == 0) return nullptr;
  return mg_create(csc_from_raw(n, n, colptr, rowind, val), b, n_levels);
}
// P_l (n_l x n_{l+1}) for l = 0 .. n_levels-2 as CSC triples; R_l = P_l^T
void* orc_mg_create_custom(int64_t n, const int32_t* colptr, const int32_t* rowind,
                           const double* val, const double* b, uint64_t n_levels,
                           const int64_t* P_cols, const int32_t* const* P_colptr,
                           const int32_t* const* P_rowind, const double* const* P_val) {
  if (n_levels == 0) return nullptr;
  std::vector<Csc> Ps;
  int64_t rows = n;
  for (uint64_t l = 0; l + 1 < n_levels; ++l) {
    Ps.push_back(csc_from_raw(rows, P_cols[l], P_colptr[l], P_rowind[l], P_val[l]));
    rows = P_cols[l];
  }
  return mg_create(csc_from_raw(n, n, colptr, rowind, val), b, n_levels, &Ps);
}
// C = A B in Eigen's conservative order (spgemm above); arrays may be null: returns nnz(C)
int64_t orc_spgemm(int64_t a_rows, int64_t a_cols, const int32_t* a_colptr, const int32_t* a_rowind,
                   const double* a_val, int64_t b_cols, const int32_t* b_colptr,
                   const int32_t* b_rowind, const double* b_val, int32_t* c_colptr,
                   int32_t* c_rowind, double* c_val) {
  Csc Cm = spgemm(csc_from_raw(a_rows, a_cols, a_colptr, a_rowind, a_val),
                  csc_from_raw(a_cols, b_cols, b_colptr, b_rowind, b_val));
  if (c_colptr && c_rowind && c_val) csc_to_raw(Cm, c_colptr, c_rowind, c_val);
  return Cm.nnz();
}
void orc_mg_destroy(void* h) { delete (Multigrid*)h; }
void orc_mg_set_smoother(void* h, int kind, uint64_t iters, double omega) {
  Multigrid* M = (Multigrid*)h;
  M->smoother = kind;
  M->sm_iters = iters;
  M->omega = omega;
}
void orc_mg_set_colors(void* h, uint64_t level, const int32_t* color,
                       int32_t n_colors) {
  Multigrid* M = (Multigrid*)h;
  M->color[level].assign(color, color + M->A[level].cols);
  M->n_colors[level] = n_colors;
}
uint64_t orc_mg_n_dofs(void* h, uint64_t level) {
  return (uint64_t)((Multigrid*)h)->A[level].rows;
}
int64_t orc_mg_level_nnz(void* h, uint64_t level) {
  return ((Multigrid*)h)->A[level].nnz();
}
void orc_mg_level_matrix(void* h, uint64_t level, int32_t* colptr,
                         int32_t* rowind, double* val) {
  csc_to_raw(((Multigrid*)h)->A[level], colptr, rowind, val);
}
// which: 0 = P, 1 = R
int64_t orc_mg_transfer_nnz(void* h, uint64_t level, int which) {
  Multigrid* M = (Multigrid*)h;
  return (which ? M->R[level] : M->P[level]).nnz();
}
void orc_mg_transfer(void* h, uint64_t level, int which, int32_t* colptr,
                     int32_t* rowind, double* val) {
  Multigrid* M = (Multigrid*)h;
  csc_to_raw(which ? M->R[level] : M->P[level], colptr, rowind, val);
}
// which: 0 = u, 1 = f (rhs), 2 = r (residual)
void orc_mg_get_vec(void* h, uint64_t level, int which, double* out) {
  Multigrid* M = (Multigrid*)h;
  const std::vector<double>& v =
      which == 0 ? M->u[level] : (which == 1 ? M->f[level] : M->r[level]);
  std::memcpy(out, v.data(), sizeof(double) * v.size());
}
void orc_mg_set_vec(void* h, uint64_t level, int which, const double* in) {
  Multigrid* M = (Multigrid*)h;
  std::vector<double>& v =
      which == 0 ? M->u[level] : (which == 1 ? M->f[level] : M->r[level]);
  std::memcpy(v.data(), in, sizeof(double) * v.size());
}
int64_t orc_mg_coarse_halfbw(void* h) { return ((Multigrid*)h)->coarse.w; }
void orc_mg_vcycle(void* h) { mg_vcycle((Multigrid*)h); }
double orc_mg_rss(void* h) {
  Multigrid* M = (Multigrid*)h;
  return rss(M->A[0], M->u[0].data(), M->f[0].data());
}
uint64_t orc_mg_solve(void* h, double tol, uint64_t every, uint64_t n_iters,
                      int* converged, double* last_rss, double* trajectory,
                      uint64_t traj_cap) {
  return mg_solve((Multigrid*)h, tol, every, n_iters, converged, last_rss,
                  trajectory, traj_cap);
}
// The V-cycle as a preconditioner (README.md:127 of the reference, its ref [7]): z = M^-1 v
// is one mg_vcycle() from the zero vector with v as the level-0 right-hand side.  The
// reference only names this use; the twin below is what the product's amg_hip_apply /
// amg_hip_pcg are compared with ("parity unpinned" against the reference itself).
void orc_mg_apply(void* h, const double* v, double* z) {
  Multigrid* M = (Multigrid*)h;
  const std::vector<double> f_save = M->f[0], u_save = M->u[0];
  M->f[0].assign(v, v + f_save.size());
  std::fill(M->u[0].begin(), M->u[0].end(), 0.0);
  mg_vcycle(M);
  std::memcpy(z, M->u[0].data(), sizeof(double) * f_save.size());
  M->f[0] = f_save;
  M->u[0] = u_save;
}
// Textbook PCG on A_0 x = b (x starts at the level-0 solution and replaces it); stops when
// ||r|| <= rtol ||b|| or after max_iters.  Sequential dot products, separate multiply / add.
uint64_t orc_mg_pcg(void* h, double rtol, uint64_t max_iters, double* relres) {
  Multigrid* M = (Multigrid*)h;
  const Csc& A = M->A[0];
  const size_t n = (size_t)A.rows;
  const std::vector<double> b = M->f[0];
  std::vector<double> x = M->u[0], r(n), z(n), p(n), q(n);
  auto dot = [&](const std::vector<double>& a, const std::vector<double>& c) {
    double s = 0.0;
    for (size_t i = 0; i < n; ++i) s += a[i] * c[i];
    return s;
  };
  const double bnorm = std::sqrt(dot(b, b));
  residual(A, x.data(), b.data(), r.data());
  uint64_t it = 0;
  double rel = bnorm > 0 ? std::sqrt(dot(r, r)) / bnorm : std::sqrt(dot(r, r));
  if (rel > rtol && max_iters > 0) {
    orc_mg_apply(h, r.data(), z.data());
    p = z;
    double rz = dot(r, z);
    while (it < max_iters) {
      spmv(A, p.data(), q.data());
      const double alpha = rz / dot(p, q);
      for (size_t i = 0; i < n; ++i) {
        x[i] = x[i] + alpha * p[i];
        r[i] = r[i] - alpha * q[i];
      }
      it += 1;
      const double rr = dot(r, r);
      rel = bnorm > 0 ? std::sqrt(rr) / bnorm : std::sqrt(rr);
      if (!(rel > rtol) || it >= max_iters) break;
      orc_mg_apply(h, r.data(), z.data());
      const double rzn = dot(r, z);
      const double beta = rzn / rz;
      for (size_t i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
      rz = rzn;
    }
  }
  M->u[0] = x;
  if (relres) *relres = rel;
  return it;
}
// CPU baseline: wall seconds for n consecutive vcycle() calls (single thread;
// the reference has no threading).  rss is not called.
double orc_mg_time_vcycles(void* h, uint64_t n) {
  auto t0 = std::chrono::steady_clock::now();
  for (uint64_t i = 0; i < n; ++i) mg_vcycle((Multigrid*)h);
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
