// Launchers of the gfx950 kernels (kernels.hip).  Device pointers only.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace amg_hip {

enum CsrMode {
  CSR_RESID = 0,   // out = f - A x   (ascending column order per row)
  CSR_JACOBI = 1,  // out = x_i + omega*((f_i - sum_{j!=i} a_ij x_j)/a_ii - x_i)
  CSR_SPMV = 2,    // out = A x
  CSR_RSSQ = 3,    // out_i = (f_i - (A x)_i)^2
  CSR_GS = 4,      // in-place Gauss-Seidel update of one colour
  CSR_JACOBI_P = 5,// Jacobi sweep whose input is x + P*uH (linear P), K-SELL only
  CSR_SPMV_ADD = 6 // out = f + A x (f may be out: the prolongation-and-add u_h = u_h + P u_H of a
                   // custom interpolator, multigrid.hpp:294-296, in one launch), K-CSR only
};

// one workgroup runs the whole symmetric pass over all colours (small levels); starts_dev:
// nc + 1 storage-row offsets on the device
hipError_t launch_dict_gs_sweep(int nc, const int32_t* starts_dev, int64_t n_storage, int words, int wmax,
                                const uint64_t* codes, const int32_t* rowid, const int32_t* doff,
                                const double* dval, int ntab, const double* f, double* u,
                                hipStream_t st);
// max_block_nnz: max entries in any 256-row block; max_row_nnz: longest row.
hipError_t launch_csr(int mode, int64_t n, int64_t nnz, int max_block_nnz,
                      int max_row_nnz, const int32_t* rowptr, const int32_t* col,
                      const double* val, const double* x, const double* f,
                      double* out, double omega, int64_t diag_shift, hipStream_t st);
// Same operations on a SELL-64 matrix (64-row panels, lane-interleaved):
// soff[n/64 + 1] panel offsets, scol/sval padded with col = -1.
// idx16 bit 0: scol holds int16 offsets from the diagonal column (pad -32768);
// bit 1: stream the matrix / f / out with non-temporal accesses
hipError_t launch_sell(int mode, int64_t n, int idx16, const int64_t* soff,
                       const void* scol, const double* sval, const double* x,
                       const double* f, double* out, double omega, int64_t diag_shift,
                       hipStream_t st);
// out = Jacobi sweep applied to (u + P uH), P = LinearInterpolator prolongation:
// prolongation + add (multigrid.hpp:294-296) fused into the first post-smoothing
// sweep; u itself is not modified.
// K-Dict: dictionary-coded rows (host_setup.hpp: DictMat); modes as launch_sell.
struct DictRef {  // device pointers of one dictionary-coded matrix
  int words = 1, wmax = 0, nt = 0, ntab = 0;
  const uint64_t* codes = nullptr;   // n * words code words (unused when rtype is set)
  const uint8_t* rtype = nullptr;    // optional: one byte per row into rwords
  const uint64_t* rwords = nullptr;  // 256 * words, entry 255 = all 0xFF
  const int32_t* doff = nullptr;     // pair table: column offset from the diagonal column
  const double* dval = nullptr;      //             value
  int scan_new = 99;                 // most entries of a row on one side beyond +-1 (K-GS-scan)
  int hb = 0;                        // largest |column offset| (3-D levels: ~ one grid plane): tile order
  // per row type, for waves whose rows all share one 7- / 15-point stencil type (scalar loads): utd =
  // {off-diagonal value or +0.0 [16], value [16], diagonal}, uti = {offset in rows [16], mask of
  // slots in use, pattern 7 / 15 / 0}; null: none
  const double* utd = nullptr;
  const int32_t* uti = nullptr;
};
void set_xcd_mapping(int on);  // contiguous run of tiles per XCD (default on)
void set_dict_rows_per_lane(int r);  // 1 or 2 (default), tuning / test switch
void set_dict_stencil(int on);       // paired-load path for wave-uniform 7- / 15-point rows (default on)
hipError_t launch_dict(int mode, int64_t n, const DictRef& D, const double* x, const double* f,
                       double* out, double omega, int64_t diag_shift, hipStream_t st);
void dict_kernel_name(int mode, int64_t n, const DictRef& D, const void* f, const void* out,
                      char* buf, size_t cap);
// one colour of the multicolour GS sweep on a dictionary-coded colour-permuted copy
hipError_t launch_dict_gs_color(int64_t p0, int64_t count, int words, int wmax,
                                const uint64_t* codes, const int32_t* rowid, const int32_t* doff,
                                const double* dval, int ntab, const double* f, double* u,
                                hipStream_t st);
// fused forms for the true-Jacobi V-cycle on linear-interpolation levels (kernels.hip):
// r = f - A x (written), f_H = R r, and either uH1 = first Jacobi sweep of the coarse level
// from zero (needs diagH) or, with uH1 == nullptr, uH0 = 0
hipError_t launch_dict_resid_restrict(int64_t n, const DictRef& D, const double* x,
                                      const double* f, double* r_out, int64_t nH, double* fH,
                                      const double* diagH, double* uH1, double* uH0,
                                      double omega, hipStream_t st);
// out = Jacobi sweep of x on this (coarse) level, then uh += P out on the finer level
hipError_t launch_dict_jacobi_prolong(int64_t n, const DictRef& D, const double* x,
                                      const double* f, double* out, double omega, int64_t n_h,
                                      const double* uh_in, double* uh_out, hipStream_t st);
// Two sweeps in one launch on small levels (narrow band, hb = half-bandwidth in rows):
//   down: u_out = Jacobi(a); r = f - A u_out (r_out optional); f_H = R r; uH1 = first coarse sweep
//   up:   t = Jacobi(a) (not stored); u_out = Jacobi(t); uh_out = uh_in + P u_out
bool dict_pair_ok(int64_t n, const DictRef& D, int hb, const void* a, const void* b, const void* c);
hipError_t launch_dict_pair_down(int64_t n, const DictRef& D, int hb, const double* a,
                                 const double* f, double* u_out, double* r_out, int64_t nH,
                                 double* fH, const double* diagH, double* uH1, double omega,
                                 hipStream_t st);
hipError_t launch_dict_pair_up(int64_t n, const DictRef& D, int hb, const double* a,
                               const double* f, double* u_out, double omega, int64_t n_h,
                               const double* uh_in, double* uh_out, hipStream_t st);
// K-Tail (kernels.hip): the deepest levels of the 2+2 true-Jacobi cycle -- down-legs of levels
// lt .. L-2, the coarsest solve (K-BandChain operands), up-legs and the prolongation into level
// lt - 1 -- in ONE launch of one workgroup.
constexpr int TAIL_LEVELS_MAX = 6;
struct TailLevelRef {
  int n, hbw, words, ntab;
  const uint8_t* rtype;
  const uint64_t* rwords;
  const int32_t* doff;
  const double* dval;
  const double* f;
  double* u;
  double* tmp;
  const double* diag;
};
struct TailRef {
  int nlev;
  TailLevelRef L[TAIL_LEVELS_MAX];
  int nc, wc;
  const double *cf, *cb, *dg;
  double *fc, *uc, *tmpc;
  const double* diagc;
  int n_fine;
  const double* uf_in;
  double* uf_out;
  double omega;
};
bool tail_level_ok(int64_t n, const DictRef& D, int hb);
int tail_max_coarse();
size_t tail_lds_bytes(const TailRef& A);  // LDS the level vectors, tables and the coarsest factor need
size_t tail_lds_capacity();               // ... and what the kernel has (static: 144 KB)
hipError_t launch_tail(const TailRef& A, hipStream_t st);

// K-Patch (kernels.hip): a level's whole down-leg / up-leg of the 2+2 true-Jacobi cycle in
// one launch over 2-D patches of the level (temporal blocking).  ptab: per (row type, slot)
// {off-diagonal value or +0.0, diagonal value or +0.0, value, LDS offset dj * pitch + di or
// -1e9 for an unused slot} as 4 doubles, `un` slots per type (patch_un(longest row)),
// nent = ntypes * un; m = pitch of the band (line length of the flat index).
struct PatchRef {
  const uint8_t* rtype = nullptr;
  const double* ptab = nullptr;
  // the same table per type for the wave-uniform path (scalar loads): utabd = per type
  // {aJ[un], value[un], diagonal}, utabi = per type {LDS offset[un], mask of off-diagonal
  // slots, mask of slots in use}
  const double* utabd = nullptr;
  const int32_t* utabi = nullptr;
  // per tile of the level: the row type every row of the tile and its halo has, or 255
  // (launch_patch_tile_flags); null = always take the general path
  const uint8_t* tflag = nullptr;
  int nent = 0, ntypes = 0, un = 0, nt = 0;
  int umask = 0;  // slots (3 x 3, bit = slot) of the level's interior row type: selects the kernel kind
  // down-leg: per tile 1 when every coarse row under it has the diagonal dHu (launch_patch_coarse_flags)
  const uint8_t* cflag = nullptr;
  double dHu = 0.0;
};
hipError_t launch_patch_coarse_flags(int64_t n, int64_t m, int64_t nH, const double* diagH, double dref,
                                     uint8_t* cflag, hipStream_t st);
// flag == null: only *n_tiles is computed (the size of the array)
hipError_t launch_patch_tile_flags(int64_t n, int64_t m, const uint8_t* rtype, int ntypes, uint8_t* flag,
                                   int64_t* n_tiles, hipStream_t st);
bool patch_geometry_ok(int64_t n, int64_t m);
int patch_un(int longest_row);
int patch_default_umask(int un);
int patch_kind_umask(int un, int umask);
int patch_lds_pitch();
int patch_max_entries();
// first: both pre-sweeps (x = u) else the second only (x = result of the first sweep);
// u_out = smoothed level vector (never x), r_out optional, f_H / uH1 as launch_dict_resid_restrict
hipError_t launch_patch_down(bool first, int64_t n, int64_t m, const PatchRef& P, const double* x,
                             const double* f, double* u_out, double* r_out, int64_t nH, double* fH,
                             const double* diagH, double* uH1, double omega, hipStream_t st,
                             int64_t line_lo = 0, int64_t line_hi = -1);
// The patch launchers take a range of grid lines [line_lo, line_hi) (line_hi < 0: the whole
// level): only the tiles that meet it run (row-block sharding, solver.cpp "slab").
int patch_tile_lines();
// u_out = two Jacobi sweeps of (x + P uH); u_out must not be x
hipError_t launch_patch_up(int64_t n, int64_t m, const PatchRef& P, const double* x, const double* f,
                           const double* uH, int64_t nH, double* u_out, double omega,
                           hipStream_t st, int64_t line_lo = 0, int64_t line_hi = -1);
// Multicolour Gauss-Seidel on patches: up to patch_rb_max_stages(tail) colour stages per launch
// (stages = patch_rb_stages(colours, count)), u_out != x.  prolong: the input is x + P uH;
// tail: followed by the residual (r_out optional), the restriction into fH and the zeroing of
// uH_zero (optional).  colour(row) = 2-bit entry ((row / m) & 1) * 6 + column class of ctab; column
// classes of c = row % m: 0: c == 0, 1: c == 1, 2: c == m-2, 3: c == m-1, else 4 + (c & 1).
int patch_rb_max_stages(bool tail);
uint32_t patch_rb_stages(const int* colors, int count);
hipError_t launch_patch_rb(bool prolong, bool tail, int64_t n, int64_t m, const PatchRef& P, const double* x,
                           const double* f, const double* uH, int64_t nH, double* u_out, double* r_out,
                           double* fH, double* uH_zero, uint32_t stages, uint32_t ctab, hipStream_t st,
                           int64_t line_lo = 0, int64_t line_hi = -1);
// K-March (kernels.hip): two true-Jacobi sweeps x -> out of a 3-D 7-point level (rows = plane * m *
// lines + line * m + column) in one plane-marching pass.  wtab: per row type 8 doubles {w(-M), w(-m),
// w(-1), w(+1), w(+m), w(+M), diagonal, 0} (+0.0 where the type has no entry); tint / wint / dint: the
// interior type and its values.  out != x.
struct MarchRef {
  int m = 0, lines = 0, planes = 0, ntypes = 0, tint = -1, chunk_planes = 0;
  double omega = 1.0, dint = 0.0;
  double wint[6] = {0, 0, 0, 0, 0, 0};
  const double* wtab = nullptr;
  const uint8_t* rtype = nullptr;
  const double* x = nullptr;
  const double* f = nullptr;
  double* out = nullptr;
};
bool march_ok(const MarchRef& A);
hipError_t launch_march(MarchRef A, hipStream_t st);
// K-Strip (kernels.hip): the colour stages of a narrow level's whole leg in one launch over strips
// of T rows (+ halo), any colouring (colour byte per row, < 16 colours); stages: 4 bits per stage.
// prolong: the input is x + P uH; tail: followed by the residual (r_out optional), the restriction
// into fH and the zeroing of uH_zero (optional).  u_out != x.  The dictionary must be row-typed.
struct StripRef {
  int n = 0, nH = 0, T = 0, H = 0, hbw = 0, nst = 0, ntab = 0;
  uint64_t stages = 0;
  const uint8_t* rtype = nullptr;
  const uint64_t* rwords = nullptr;
  const int32_t* doff = nullptr;
  const double* dval = nullptr;
  const uint8_t* color = nullptr;
  const double* x = nullptr;
  const double* f = nullptr;
  double* u_out = nullptr;
  double* r_out = nullptr;
  const double* uH = nullptr;
  double* fH = nullptr;
  double* uH_zero = nullptr;
};
int strip_window_max();
hipError_t launch_mc_strip(bool prolong, bool tail, StripRef A, const DictRef& D, hipStream_t st);
// uh_out = uh_in + P uH for the linear interpolation pair (16-byte aligned vectors)
hipError_t launch_linear_prolong_to(int64_t n_h, int64_t n_H, const double* uH,
                                    const double* uh_in, double* uh_out, hipStream_t st);
hipError_t launch_sell_jacobi_prolong(int64_t n, int idx16, const int64_t* soff,
                                      const void* scol, const double* sval, const double* u,
                                      const double* uH, int64_t nH, const double* f, double* out,
                                      double omega, hipStream_t st);
// out = Jacobi sweep applied to u == 0: needs only the diagonal and f.
hipError_t launch_jacobi_from_zero(int64_t n, const double* diag, const double* f, double* out,
                                   double omega, hipStream_t st);
// One colour of the multicolour Gauss-Seidel: storage rows [row0, row0+count) of a
// colour-permuted SELL-64 matrix (row0 % 64 == 0), rowid = dof of each storage
// row (-1 pads); u[rowid] = (f - sum_{j != i} a_ij u_j) / a_ii in place.
hipError_t launch_sell_gs_color(int64_t n_storage, int max_width, const int64_t* soff,
                                const int32_t* scol, const double* sval, const int32_t* rowid,
                                int64_t row0, int64_t count, const double* f, double* u,
                                hipStream_t st);

// uH_zero (may be null): coarse solution vector zero-filled in the same pass
hipError_t launch_linear_restrict(int64_t n_h, int64_t n_H, const double* r, double* fH,
                                  double* uH_zero, hipStream_t st);
hipError_t launch_linear_prolong_add(int64_t n_h, int64_t n_H, const double* uH,
                                     double* uh, hipStream_t st);
hipError_t launch_add_inplace(int64_t n, const double* x, double* y, hipStream_t st);
// *out = sum x_i (square=0) or sum x_i^2 (square=1); scratch: 1024 doubles
hipError_t launch_sum(int64_t n, const double* x, double* out, double* scratch, int square,
                      hipStream_t st);

// K-PCG: *out = sum x_i y_i (deterministic two-stage tree; scratch: 1024 doubles), and the
// two vector updates with alpha = *num / *den resp. beta = *num / *den formed on the device
hipError_t launch_dot(int64_t n, const double* x, const double* y, double* out, double* scratch,
                      hipStream_t st);
hipError_t launch_pcg_update_xr(int64_t n, const double* num, const double* den, double* x, double* r,
                                const double* p, const double* q, hipStream_t st);
hipError_t launch_pcg_update_p(int64_t n, const double* num, const double* den, double* p,
                               const double* z, hipStream_t st);

// In-graph neighbour exchange (K-Halo).  All pointers are device pointers; *_at_*
// and dst_* point into a neighbour's hipIpc-mapped arena.
struct HaloArgs {
  double* dst_prev; const double* src_prev; int64_t cnt_prev;
  double* dst_next; const double* src_next; int64_t cnt_next;
  uint32_t* my_free_from_prev; uint32_t* my_free_from_next;
  uint32_t* data_at_prev; uint32_t* data_at_next;
  uint32_t* my_data_from_prev; uint32_t* my_data_from_next;
  int32_t recv_prev, recv_next;
  uint32_t* timeout;
};
hipError_t launch_halo_exchange(const HaloArgs& a, hipStream_t st);
hipError_t launch_halo_ack(uint32_t* free_at_prev, uint32_t* free_at_next, hipStream_t st);
constexpr int GATHER_MAX_RANKS = 16;
struct GatherArgs {
  int32_t rank, world;
  const double* src; int64_t cnt, off;
  double* dst[GATHER_MAX_RANKS];          // every rank's full vector (own included)
  uint32_t* data_at[GATHER_MAX_RANKS];    // word [me] in rank g's DATA block
  uint32_t* free_at[GATHER_MAX_RANKS];    // word [me] in rank g's FREE block
  uint32_t* my_data_from;                 // my DATA block, one word per source rank
  uint32_t* my_free_from;                 // my FREE block, one word per destination rank
  uint32_t* timeout;
};
hipError_t launch_gather(const GatherArgs& a, hipStream_t st);
hipError_t launch_gather_ack(const GatherArgs& a, hipStream_t st);

struct LexDev {  // device copy of a LexSchedule
  int32_t block = 0, width = 0;
  int64_t n_slots = 0;
  const int32_t* row = nullptr;
  const int16_t* depth = nullptr;
  const int32_t* win_depth = nullptr;
  const int32_t* col = nullptr;
  const double* val = nullptr;
  const int16_t* src = nullptr;
};
// mode 0 SpGS, 1 reference "Jacobi" (forward GS), 2 SOR
hipError_t launch_gs_lex(const LexDev& S, const double* b, double* u, int mode, double omega,
                         hipStream_t st);

// Line-scan form of the lexicographic sweeps on a dictionary-coded matrix (K-GS-scan):
// chunks of C rows, the in-chunk chain u_k = c_k + q_k u_{k-1} solved by an affine scan.
// mode 0 SpGS update, 1 reference "Jacobi" (forward GS), 2 SOR.  ring: power of two.
void set_scan_typed(int on);  // K-GS-scan: per-type fast path on row-typed matrices (default on)
// s_old: scratch of n doubles (the sums over rows the sweep has not reached, formed by a
// parallel pre-pass).
// chunks of up to gs_scan_max_chunk() rows (else 1024): row-typed matrices whose rows have at most 4
// new-side entries besides the chain
bool gs_scan_long_chunks_ok(const DictRef& D);
int gs_scan_max_chunk();
hipError_t launch_gs_scan(int64_t n, const DictRef& D, const double* b, double* u, double* s_old,
                          bool backward, int mode, double omega, int C, int ring, hipStream_t st);

// m: lanes taking part (power of two > half-bandwidth, 4..64); sched_f/sched_b:
// per-(step, lane) L operands (host_setup: band_schedule); y: scratch, n doubles
hipError_t launch_band_solve(int64_t n, int m, const double* sched_f, const double* sched_b,
                             const double* dg, const double* f, double* y, double* x,
                             hipStream_t st);

// K-BandChain: narrow band (w <= 3), small n (<= 2048), everything in LDS; same bits as K-Band.
// cf / cb: n x w operands per step (host_setup.hpp: band_chain_schedule)
bool band_chain_ok(int64_t n, int64_t w);
hipError_t launch_band_chain(int64_t n, int w, const double* cf, const double* cb, const double* dg,
                             const double* f, double* x, hipStream_t st, int64_t n_h = 0,
                             const double* uh_in = nullptr, double* uh_out = nullptr);
// Any half-bandwidth (K-BandWide): rows in blocks of 64; sched_f/sched_b hold per block a
// [w][64] panel of the operands that reach into earlier blocks followed by the block's own
// [64][64] triangle (host_setup: band_wide_schedule).  w + 64 <= 8192 (LDS ring).
hipError_t launch_band_wide(int64_t n, int64_t w, const double* sched_f, const double* sched_b,
                            const double* dg, const double* f, double* y, double* x,
                            hipStream_t st);

// Partitioned coarse solve (K-Spike); arrays as in host_setup.hpp: SpikeFactor.
struct SpikeArgs {
  int64_t n;
  int32_t w, c, P, m;
  int64_t sched_stride;
  const double *sched_f, *sched_b, *d, *V, *W, *Vt, *Wh;
  const double* f;
  double* x;
  double *G, *Z;   // scratch, n doubles each
  double *T, *H;   // scratch, P * max(w,1)
};
hipError_t launch_spike_solve(const SpikeArgs& a, hipStream_t st);

// K-Galerkin (setup): A_H = R (A P) for the linear interpolation pair, on CSR arrays that
// live on the device; count pass (fill = false: cnt[row]) / fill pass (orp = scanned
// counts) around launch_exclusive_scan.  Same entry order and bits as spgemm_csr.
hipError_t launch_exclusive_scan(int64_t n, const int32_t* counts, int32_t* offsets, int64_t* bsum,
                                 int64_t* total, hipStream_t st);
// general CSR x CSR product, Eigen's conservative order (count pass: cnt; fill pass: orp); rows
// of A with more than 32 entries set *overflow
hipError_t launch_spgemm(bool fill, int64_t n_rows, const int32_t* arp, const int32_t* acol,
                         const double* aval, const int32_t* brp, const int32_t* bcol, const double* bval,
                         int32_t* cnt, const int32_t* orp, int32_t* ocol, double* oval,
                         int32_t* overflow, hipStream_t st);
hipError_t launch_galerkin_ap(bool fill, int64_t n_h, int64_t n_H, const int32_t* arp,
                              const int32_t* acol, const double* aval, int32_t* cnt,
                              const int32_t* orp, int32_t* ocol, double* oval, hipStream_t st);
hipError_t launch_galerkin_rap(bool fill, int64_t n_h, int64_t n_H, const int32_t* prp,
                               const int32_t* pcol, const double* pval, int32_t* cnt,
                               const int32_t* orp, int32_t* ocol, double* oval, hipStream_t st);

// K-Setup: Grid generators and the dictionary encoder on the device (kernels.hip)
hipError_t launch_laplacian_count(int dim, int64_t n, int64_t n_last, int64_t N, int32_t* cnt, hipStream_t st);
hipError_t launch_laplacian_fill(int dim, int64_t n, int64_t n_last, int64_t N, const int32_t* rowptr, int32_t* col,
                                 double* val, double off, double diag, hipStream_t st);
// stats[0] = longest row (exact zeros skipped when prune), stats[1] = 1 when not bitwise
// symmetric (both zero-initialised by the caller); diag (optional) = a_ii
hipError_t launch_csr_inspect(int64_t n, const int32_t* rowptr, const int32_t* col, const double* val,
                              bool prune, int32_t* stats, double* diag, hipStream_t st);
// fail: 64 int32, zero-initialised: [0] = rows that did not encode, [1..62] = their indices
hipError_t launch_dict_encode(int64_t n, const int32_t* rowptr, const int32_t* col, const double* val,
                              bool prune, const int32_t* doff, const double* dval, int ntab, int words,
                              uint64_t* codes, int32_t* fail, hipStream_t st);
hipError_t launch_dict_types(int64_t n, const uint64_t* codes, int words, const uint64_t* rwords,
                             int ntypes, uint8_t* rtype, int32_t* fail, hipStream_t st);

}  // namespace amg_hip
