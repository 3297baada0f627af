// Host-side SETUP of the multigrid hierarchy (runs once per solver; the
// reference does the same work in Multigrid's constructor, multigrid.hpp:151-244).
// Product code: shares nothing with oracle/.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace amg_hip {

// Compressed sparse storage.  As "CSC": outer = column, inner = row index.
// As "CSR": outer = row, inner = column index.  Inner indices ascending.
struct Sparse {
  int64_t n_outer = 0, n_inner = 0;
  std::vector<int32_t> ptr;  // n_outer + 1
  std::vector<int32_t> idx;  // nnz
  std::vector<double> val;   // nnz
  int64_t nnz() const { return (int64_t)idx.size(); }
};

Sparse from_raw(int64_t n_outer, int64_t n_inner, const int32_t* ptr,
                const int32_t* idx, const double* val);
// Validates a compressed matrix (monotone ptr, indices in range and strictly
// ascending).  Returns "" when fine.
std::string validate(const Sparse& M, const char* name);

// Storage transpose: CSC(M) -> CSC(M^T) == CSR(M).
Sparse transpose(const Sparse& M);
bool same_arrays(const Sparse& a, const Sparse& b);  // bitwise equality

// interpolator.hpp:106-141, LinearInterpolator::make_operators: returns P in
// CSC (n_h x n_H); R = transpose(P).
Sparse linear_P(int64_t n_h, int64_t n_H);
// true when P (CSC, n_h x n_H) is exactly what linear_P produces
bool is_linear_P(const Sparse& P, int64_t n_h, int64_t n_H);

// multigrid.hpp:127-130
inline int64_t coarse_dofs(int64_t n_h) { return (n_h + 1) / 2 - 1; }

// Strength-based C/F coarsening with direct interpolation (classical Ruge-Stueben first pass;
// the alternative the reference names and does not build, README.md:104-109).  A_rows: CSR(A).
// Couplings are measured against the sign of the diagonal (the reference's operators are
// NEGATIVE definite: diagonal < 0, couplings > 0): s_ij = -sign(a_ii) a_ij; row i depends
// strongly on j when s_ij > 0 and s_ij >= theta * max_k s_ik.  First pass: repeatedly make
// the undecided point with the most strong dependants (ties: lowest index) a C point, its
// undecided dependants F points, and raise / lower the measures of their neighbours.  Points
// without strong couplings are F points with an empty row.  Direct interpolation for an F
// point i over C_i = its strong C neighbours, sums in ascending column order:
//   w_ij = -(sum_{k: s_ik > 0} a_ik / sum_{k in C_i} a_ik) * a_ij / (a_ii + sum_{k: s_ik < 0} a_ik).
// Returns P in CSC (n_h x n_H, C points numbered in ascending fine index); R = transpose(P).
// is_c (optional): 1 for C points.
Sparse ruge_stueben_P(const Sparse& A_rows, double theta, std::vector<uint8_t>* is_c = nullptr);

// C = A * B, all three in row-major CSR, Gustavson row by row.  For a fixed
// output entry (i, J) the products are added in ascending k, which is the
// order Eigen's conservative column-major product uses (multigrid.hpp:222);
// entries that cancel to exactly 0.0 stay in the pattern.  Multi-threaded over
// row blocks (results do not depend on the thread count).
Sparse spgemm_csr(const Sparse& A, const Sparse& B, int n_threads);

// Galerkin coarse operator A_H = R (A P) given the row-major forms
// CSR(R), CSR(A), CSR(P).  Returns CSR(A_H).
Sparse galerkin_csr(const Sparse& Rr, const Sparse& Ar, const Sparse& Pr,
                    int n_threads);

// ---- SELL-64: CSR sliced into 64-row panels, lane-interleaved ----------------
struct Sell64 {
  int64_t n = 0;
  int32_t max_width = 0;
  std::vector<int64_t> soff;   // n_panels + 1
  std::vector<int32_t> col;    // -1 = padding
  std::vector<double> val;
  int64_t slots() const { return (int64_t)col.size(); }
};
// rows_as: matrix whose outer index is the row.  Entry order inside a row kept.
void to_sell64(const Sparse& rows_as, Sell64* out);

// ---- dictionary-coded rows (K-Dict) -------------------------------------------
// The Galerkin hierarchies of constant-coefficient stencils hold a handful of
// distinct (column - diagonal column, value) pairs (2-D Poisson: 5 on level 0, <= 60
// below).  Row r is stored as `words` 64-bit words of byte codes, code k of entry j
// in byte j (ascending column order kept), 0xFF = no entry; the pairs themselves sit
// in a table of <= 255 entries.  8 or 16 bytes per row instead of 10-12 per entry,
// and the decoded values are the original doubles, so results are bit-identical.
struct DictMat {
  int64_t n = 0;
  int32_t words = 1;            // 64-bit code words per row (1: <= 8 entries, 2: <= 16)
  int32_t max_width = 0;
  std::vector<uint64_t> codes;  // n * words
  std::vector<int32_t> doff;    // table: column offset from the row's diagonal column
  std::vector<double> dval;     // table: value
  // second level: when the rows repeat as whole code words (<= 255 distinct), one byte
  // per row into a word table of 256 * words entries (entry 255 = empty row, all 0xFF)
  std::vector<uint8_t> rtype;   // n, empty when the matrix has more distinct rows
  std::vector<uint64_t> rwords; // 256 * words
};
// false when the matrix does not qualify (more than 255 pairs or a row > 16 entries).
// rowid (optional): dof of every storage row (colour-permuted copies; -1 = empty padding
// row); offsets are then taken from column rowid[r] instead of r + diag_shift.
bool to_dict(const Sparse& rows_as, int64_t diag_shift, DictMat* out,
             const int32_t* rowid = nullptr);

// ---- coarsest level: banded LDL^T (replaces Eigen::SimplicialLDLT) ---------
struct BandFactor {
  int64_t n = 0, w = 0;
  // Column-major-by-diagonal layout chosen for the device solve kernel:
  // lcol[k*w + (d-1)] = L[k+d, k]  (d = 1..w), zero past the matrix end;
  // dinv is NOT stored: the solve divides by d[k] (IEEE divide, no reciprocal).
  std::vector<double> lcol;
  std::vector<double> d;
};
// `A` symmetric, CSC or CSR (same thing).  Fails (returns message) on a zero
// pivot or when the band would need more than `max_bytes`.
std::string band_factor(const Sparse& A, size_t max_bytes, BandFactor* out);

// Device layout of the factor for the one-wave substitution kernel: m = power of
// two > w (4..64) lanes; row i lives in lane i % m (forward) / (n-1-i) % m
// (backward).  sched_f[s*m + l] = L[s+d, s], d = (l - s) mod m, when 1<=d<=w and
// s+d < n, else 0;  sched_b[t*m + l] = L[k, k-d], k = n-1-t, d = (l - t) mod m,
// when 1<=d<=w and k-d >= 0, else 0.
struct BandSchedule {
  int64_t n = 0;
  int32_t m = 0;
  std::vector<double> sched_f, sched_b, d;
};
// fails when w > 63
std::string band_schedule(const BandFactor& F, BandSchedule* out);

// Device layout for the narrow-band LDS kernel (K-BandChain, w <= 3): per step s the w
// operands of the recurrence, farthest column first.
//   cf[s*w + t] = L[s, s-w+t]           (0 when the column is negative)
//   cb[s*w + t] = L[i+w-t, i], i = n-1-s (0 when the row is past the end)
struct BandChain {
  int64_t n = 0, w = 0;
  std::vector<double> cf, cb, d;
};
void band_chain_schedule(const BandFactor& F, BandChain* out);

// Device layout for the general-bandwidth kernel (K-BandWide, w > 63): rows in blocks of
// 64.  Per block b a panel of (w + 64) x 64 doubles, [t][lane]:
//   t < w : L[i, i-w+t] for row i = 64 b + lane when that column lies in an EARLIER block
//   t >= w: L[i, 64 b + (t-w)] -- the block's own strictly lower triangle
// (zero elsewhere).  sched_b is the same thing for the mirrored system L'[i',k'] =
// L[n-1-k', n-1-i'], i.e. the L^T solve walked from the last row.
struct BandWide {
  int64_t n = 0, w = 0;
  std::vector<double> sched_f, sched_b, d;
};
// fails when the panels would need more than max_bytes or the LDS ring (w + 64 > 8192)
std::string band_wide_schedule(const BandFactor& F, size_t max_bytes, BandWide* out);

// Partitioned ("spike") form of the same factor for the parallel coarse solve:
// the n rows are cut into P partitions of c rows (c a multiple of 64, c >= w).
// Every partition solves its own triangular system with the one-wave substitution
// kernel (one workgroup per partition, all in parallel), a short serial recurrence
// over the partition boundaries (w unknowns each) couples them, and every row
// corrects itself with its spike row.  Depth ~ 2c + P steps instead of n.
struct SpikeFactor {
  int64_t n = 0;
  int32_t w = 0, c = 0, P = 0, m = 0;
  int64_t sched_stride = 0;        // doubles between the schedules of two partitions
  std::vector<double> sched_f;     // P local forward schedules (BandSchedule layout)
  std::vector<double> sched_b;     // P local backward schedules
  std::vector<double> d;           // n
  std::vector<double> V;           // [n][w]  forward spikes  L_pp^-1 B_p     (rows of partition 0: 0)
  std::vector<double> W;           // [n][w]  backward spikes L_pp^-T B_{p+1}^T (last partition: 0)
  std::vector<double> Vt;          // [P][m(j)][m(k)]: tail rows of V, k fastest, zero-padded
  std::vector<double> Wh;          // [P][m(j)][m(k)]: head rows of W, k fastest, zero-padded
};
std::string spike_factor(const BandFactor& F, SpikeFactor* out);

// ---- exact lexicographic Gauss-Seidel schedule ------------------------------
// Rows are executed in `order`; rows inside one window of `block` consecutive
// slots that depend on an earlier slot of the same window get `depth` > 0.
struct LexSchedule {
  int32_t block = 64;    // window size = threads per workgroup
  int32_t width = 0;     // ELL width (max entries per row)
  int64_t n = 0;         // rows
  int64_t n_slots = 0;   // n rounded up to a multiple of block
  std::vector<int32_t> row;     // n_slots: row of each slot (-1 = padding)
  std::vector<int16_t> depth;   // n_slots
  std::vector<int32_t> win_depth;  // per window: max depth
  // ELL arrays, entry e of slot s at [e*n_slots + s]; col = -1 pads.
  std::vector<int32_t> col;
  std::vector<double> val;
  std::vector<int16_t> src;     // producing lane inside the window, or -1
  int64_t n_sets = 0;           // DAG depth of the sweep
};
// rows_as = matrix whose OUTER index is the "row" the sweep walks (for SpGS
// that is the CSC itself, smoother.hpp:101-117).  backward: row order n-1..0.
// Numerically zero entries do not create dependencies (x + 0*u == x).
std::string build_lex_schedule(const Sparse& rows_as, bool backward,
                               int32_t max_width, LexSchedule* out);

// ---- multicolour ordering -----------------------------------------------------
// Greedy first-fit colouring in row order over the numerically non-zero
// pattern.  color[i] in [0, n_colors).
void greedy_coloring(const Sparse& rows_as, std::vector<int32_t>* color,
                     int32_t* n_colors);

// Colour-permuted copy for the multicolour kernel: the rows of colour c are
// stored contiguously from storage row start[c] (a multiple of 64, padded with
// empty rows); rowid[p] = dof of storage row p or -1.  Exact-zero entries are
// dropped (they change nothing: x + 0*u == x for finite u).
struct ColorPerm {
  Sparse rows;                  // n_outer = storage rows, columns = natural dof indices
  std::vector<int32_t> rowid;   // storage rows
  std::vector<int64_t> start;   // n_colors + 1
};
void build_color_perm(const Sparse& rows_as, const std::vector<int32_t>& color,
                      int32_t n_colors, ColorPerm* out);

// ---- Grid<double> (grid.hpp) ------------------------------------------------
Sparse laplacian(int dim, int64_t n, int64_t n_last = -1);  // CSC (== CSR, symmetric); n_last: units of the slowest axis (window)
void rhs(int dim, int64_t n, double* b);
void rhs_range(int dim, int64_t n, double* b, int64_t dof0, int64_t dof1);  // entries [dof0, dof1)

}  // namespace amg_hip
