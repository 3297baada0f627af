// =============================================================================
// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the V-cycle.
//
// All kernels are HBM-bandwidth bound sparse fp64 kernels (~0.17 flop/byte):
// no MFMA.  Compile with -ffp-contract=off: parity with the reference's
// arithmetic (separate IEEE multiply/add, true divide) is part of
// the contract and several kernels are bit-exact against the CPU oracle.
//
// K-Dict (dict_kernel, dict_resid_restrict_kernel, dict_jacobi_prolong_kernel,
//   dict_gs_color_kernel)   -- residual / true Jacobi / SpMV / rss terms / one colour of
//   the multicolour GS on dictionary-coded matrices: 8 or 16 bytes of byte codes per
//   row into an LDS table of the matrix's distinct (column offset, value) pairs, and
//   the forms fused across a level boundary (residual + restriction + first coarse
//   sweep; last sweep + prolongation).  The default layout wherever a matrix qualifies.
// K-SELL (sell_kernel)      -- the same operations on CSR sliced into 64-row
//   lane-interleaved panels, 16-bit relative column indices, non-temporal matrix
//   stream (see the comment at the kernel).  Fallback layout; the fastest plain-CSR
//   form measured.
// K-CSR (csr_stage_kernel)  -- the same operations on plain CSR arrays (device-pointer
//   API, custom interpolators, fallback when SELL would pad too much).
//   One 256-thread workgroup owns 256 consecutive rows.  The workgroup's slice
//   of the CSR column-index and value arrays is contiguous in HBM, so it is
//   streamed with 16-byte-per-lane coalesced loads into LDS (the per-row
//   non-zeros are staged, never gathered).  Then one lane per row walks its
//   entries out of LDS in ascending column order (bit-exact summation order of
//   Eigen's column-major SpMV, multigrid.hpp:272-274) and gathers x[col]; for
//   banded stencil matrices consecutive lanes gather consecutive x, i.e. the
//   gathers coalesce into a few 128-B lines served by L1/L2.  LDS strides of 5
//   or 9 entries per lane are bank-conflict free for ds_read_b64 / ds_read_b32.
// K-Restrict / K-ProlongAdd -- matrix-free LinearInterpolator transfers.
// K-SumSq                   -- wave-shuffle + LDS tree reduction (deterministic).
// K-GS-lex (gs_lex_window)  -- exact lexicographic Gauss-Seidel, dependency
//   scheduled (parity mode, latency bound by construction).
// K-Band   (band_solve)     -- coarsest-level banded LDL^T solve, one wave.
// K-Spike  (spike_*)        -- the same solve partitioned over many waves (opt-in).
// K-Galerkin (galerkin_*)   -- setup: R (A P) for the linear interpolation pair.
// K-Halo / K-Gather         -- multi-GPU neighbour exchange and all-gather as
//   graph-capturable kernels over hipIpc-mapped peer memory.
// =============================================================================
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include <cstdio>
#include <type_traits>
#include <utility>

#include "kernels.hpp"

namespace amg_hip {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every
// outstanding GLOBAL access of the wave (s_waitcnt vmcnt(0): prefetched operands, streamed-out
// results); kernels whose waves talk to each other through LDS alone use this one and keep
// their global loads and stores in flight across the barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------ K-CSR ----
constexpr int CSR_BLOCK = 256;

template <int MODE, int K, int U>
__global__ __launch_bounds__(CSR_BLOCK) void csr_stage_kernel(
    int64_t n, int64_t nnz, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const double* __restrict__ val,
    const double* __restrict__ x, const double* f,  // f may be out (CSR_SPMV_ADD in place)
    double* out, double omega, int64_t diag_shift) {
  constexpr int CAP = CSR_BLOCK * K;  // entries staged per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds_val = reinterpret_cast<double*>(smem);                    // CAP + 4
  int32_t* lds_col = reinterpret_cast<int32_t*>(smem + (CAP + 4) * 8);  // CAP + 4

  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * CSR_BLOCK;
  const int64_t row = r0 + tid;
  const bool live = row < n;
  const int64_t rlast = (r0 + CSR_BLOCK < n) ? r0 + CSR_BLOCK : n;
  const int64_t p0 = rowptr[r0];
  const int64_t p1 = rowptr[rlast];
  int64_t rs = 0, re = 0;
  double fi = 0.0, xi = 0.0;
  if (live) {
    rs = rowptr[row];
    re = rowptr[row + 1];
    if (MODE != CSR_SPMV) fi = f[row];
    if (MODE == CSR_JACOBI) xi = x[row + diag_shift];
  }
  double acc = (MODE == CSR_RESID) ? fi : 0.0;
  double diag = 0.0;
  const int64_t drow = row + diag_shift;

  for (int64_t c0 = p0 & ~(int64_t)3; c0 < p1; c0 += CAP) {
    const int64_t c1 = (c0 + CAP < p1) ? c0 + CAP : p1;  // chunk = [c0, c1)
    const int cnt = (int)(c1 - c0);
    // ---- stage: coalesced 16-B loads HBM -> LDS ----
#pragma unroll
    for (int it = 0; it < K / 4 + 1; ++it) {
      const int i = (it * CSR_BLOCK + tid) * 4;
      if (i < cnt) {
        if (c0 + i + 4 <= nnz) {
          const int4 cc = *reinterpret_cast<const int4*>(col + c0 + i);
          *reinterpret_cast<int4*>(lds_col + i) = cc;
        } else {
          for (int t = 0; t < 4 && c0 + i + t < nnz; ++t) lds_col[i + t] = col[c0 + i + t];
        }
      }
    }
#pragma unroll
    for (int it = 0; it < K / 2 + 1; ++it) {
      const int i = (it * CSR_BLOCK + tid) * 2;
      if (i < cnt) {
        if (c0 + i + 2 <= nnz) {
          const double2 vv = *reinterpret_cast<const double2*>(val + c0 + i);
          *reinterpret_cast<double2*>(lds_val + i) = vv;
        } else {
          lds_val[i] = val[c0 + i];
        }
      }
    }
    __syncthreads();
    // ---- one lane per row, ascending column order ----
    int64_t p = rs > c0 ? rs : c0;
    const int64_t pe = re < c1 ? re : c1;
    while (p < pe) {
      int32_t c[U];
      double v[U], xx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = p + u < pe;
        const int o = ok ? (int)(p + u - c0) : (int)(p - c0);
        c[u] = lds_col[o];
        v[u] = lds_val[o];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) xx[u] = x[c[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (p + u < pe) {
          if (MODE == CSR_RESID) {
            acc -= v[u] * xx[u];
          } else if (MODE == CSR_JACOBI) {
            if ((int64_t)c[u] == drow) diag = v[u];
            else acc += v[u] * xx[u];
          } else {
            acc += v[u] * xx[u];
          }
        }
      }
      p += U;
    }
    __syncthreads();
  }
  if (live) {
    if (MODE == CSR_RESID || MODE == CSR_SPMV) {
      out[row] = acc;
    } else if (MODE == CSR_SPMV_ADD) {
      out[row] = fi + acc;  // t = P u_H summed from 0, then u_h + t: the reference's two statements
    } else if (MODE == CSR_JACOBI) {
      out[row] = (diag == 0.0) ? xi : xi + omega * ((fi - acc) / diag - xi);
    } else {  // CSR_RSSQ: (b - bhat)^2, common.hpp:24
      const double d = fi - acc;
      out[row] = d * d;
    }
  }
}

template <int MODE, int K, int U>
static hipError_t launch_csr_ku(int64_t n, int64_t nnz, const int32_t* rowptr,
                                const int32_t* col, const double* val,
                                const double* x, const double* f, double* out,
                                double omega, int64_t diag_shift, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  const size_t lds = (size_t)(CSR_BLOCK * K + 4) * 12;
  const unsigned grid = (unsigned)((n + CSR_BLOCK - 1) / CSR_BLOCK);
  hipLaunchKernelGGL((csr_stage_kernel<MODE, K, U>), dim3(grid), dim3(CSR_BLOCK), lds,
                     st, n, nnz, rowptr, col, val, x, f, out, omega, diag_shift);
  return hipGetLastError();
}

template <int MODE>
static hipError_t launch_csr_mode(int64_t n, int64_t nnz, int max_block_nnz,
                                  int max_row_nnz, const int32_t* rowptr,
                                  const int32_t* col, const double* val,
                                  const double* x, const double* f, double* out,
                                  double omega, int64_t diag_shift, hipStream_t st) {
  // +3: the chunk start is rounded down to a multiple of 4 entries.  U (gathers
  // issued per pass of a lane over its row) is matched to the longest row: a
  // 5-point row walked with U = 8 wastes 3 LDS reads + 3 gathers (measured
  // 4.9 -> 5.6 TB/s on the 4096^2 fine level, tools/kbench.hip).
  const int need = max_block_nnz + 3;
  if (need <= CSR_BLOCK * 4 && max_row_nnz <= 3)
    return launch_csr_ku<MODE, 4, 3>(n, nnz, rowptr, col, val, x, f, out, omega, diag_shift, st);
  if (need <= CSR_BLOCK * 6 && max_row_nnz <= 5)
    return launch_csr_ku<MODE, 6, 5>(n, nnz, rowptr, col, val, x, f, out, omega, diag_shift, st);
  if (need <= CSR_BLOCK * 8 && max_row_nnz <= 7)
    return launch_csr_ku<MODE, 8, 7>(n, nnz, rowptr, col, val, x, f, out, omega, diag_shift, st);
  if (need <= CSR_BLOCK * 10 && max_row_nnz <= 9)
    return launch_csr_ku<MODE, 10, 9>(n, nnz, rowptr, col, val, x, f, out, omega, diag_shift, st);
  return launch_csr_ku<MODE, 16, 8>(n, nnz, rowptr, col, val, x, f, out, omega, diag_shift, st);
}

hipError_t launch_csr(int mode, int64_t n, int64_t nnz, int max_block_nnz,
                      int max_row_nnz, const int32_t* rowptr, const int32_t* col,
                      const double* val, const double* x, const double* f,
                      double* out, double omega, int64_t diag_shift, hipStream_t st) {
  switch (mode) {
    case CSR_RESID:
      return launch_csr_mode<CSR_RESID>(n, nnz, max_block_nnz, max_row_nnz, rowptr, col,
                                        val, x, f, out, omega, diag_shift, st);
    case CSR_JACOBI:
      return launch_csr_mode<CSR_JACOBI>(n, nnz, max_block_nnz, max_row_nnz, rowptr, col,
                                         val, x, f, out, omega, diag_shift, st);
    case CSR_SPMV:
      return launch_csr_mode<CSR_SPMV>(n, nnz, max_block_nnz, max_row_nnz, rowptr, col,
                                       val, x, f, out, omega, diag_shift, st);
    case CSR_RSSQ:
      return launch_csr_mode<CSR_RSSQ>(n, nnz, max_block_nnz, max_row_nnz, rowptr, col,
                                       val, x, f, out, omega, diag_shift, st);
    case CSR_SPMV_ADD:
      return launch_csr_mode<CSR_SPMV_ADD>(n, nnz, max_block_nnz, max_row_nnz, rowptr, col,
                                           val, x, f, out, omega, diag_shift, st);
  }
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------- K-SELL -----
// Same four operations on the solver's own level matrices, stored as CSR sliced
// into 64-row panels with each panel lane-interleaved (SELL-64): entry j of row
// r sits at soff[r/64] + 64*j + (r%64), rows padded to the panel's longest row
// with col = -1.  One lane per row; every load of a wave is one contiguous
// 256-B (col) / 512-B (val) segment, no row pointer, no LDS round trip, no
// barrier, ascending-column order per row kept (bit-identical results to K-CSR).
// Measured on the 4096^2 fine level: 6.2 TB/s algorithmic vs 5.6 TB/s for K-CSR
// (tools/kbench.hip) -- the panel layout is what "CSR laid out for coalesced
// HBM reads" comes to on wave64 hardware.
// MODE CSR_GS (multicolour Gauss-Seidel): rows are the colour-permuted rows
// [row0, row0 + count) of the matrix, rowid[p] is the dof each one updates
// (-1 = padding), x is updated IN PLACE -- rows of one colour do not reference
// each other, so the launch is race-free.
// IDX16: column indices stored as 16-bit offsets from the row's diagonal column
// (col - (row + dshift)), pad = -32768.  Every level of a banded hierarchy with
// half-bandwidth < 32768 qualifies; 10 instead of 12 bytes per entry, and a wave's
// index load is exactly one 128-B line.
// NT: the matrix stream (index, value), f and the output use non-temporal
// accesses, so that the once-read stream does not evict the x lines the gathers
// re-use from L2 (measured 229 -> 208 us on the 4096^2 fine level).  Only for
// matrices far larger than the Infinity Cache; small levels keep the default
// policy because their matrix is re-read from cache by the next sweep.
template <int MODE, bool IDX16, bool NT>
__global__ __launch_bounds__(256) void sell_kernel(
    int n, const int64_t* __restrict__ soff, const void* __restrict__ scol_v,
    const double* __restrict__ sval, const double* x, const double* __restrict__ f,
    double* out, double omega, const int32_t* __restrict__ rowid, int row0,
    const double* __restrict__ uH, int nH, int dshift) {
  // CSR_JACOBI_P: x is not the smoother's input yet -- the input is x + P uH
  // (LinearInterpolator prolongation, interpolator.hpp:106-129), formed on the
  // fly per gathered entry in the same order as K-ProlongAdd:
  //   t[2j+1] = 0 + 1.0 uH[j];  t[2j] = (0 + 0.5 uH[j-1]) + 0.5 uH[j].
  auto corrected = [&](int c) -> double {
    const double xc = x[c];
    const int j = c >> 1;
    const double a = uH[(j >= 1 && j - 1 < nH) ? j - 1 : 0];
    const double b = uH[j < nH ? j : 0];
    double t = 0.0;
    if (c & 1) {
      if (j < nH) t += 1.0 * b;
    } else {
      if (j >= 1 && j - 1 < nH) t += 0.5 * a;
      if (j < nH) t += 0.5 * b;
    }
    return xc + t;
  };
  const int32_t* __restrict__ scol = static_cast<const int32_t*>(scol_v);
  const int16_t* __restrict__ scol16 = static_cast<const int16_t*>(scol_v);
  const int p = row0 + blockIdx.x * 256 + threadIdx.x;  // storage row
  const int s = p >> 6;
  if ((s << 6) >= n) return;  // whole wave past the end
  const int64_t o0 = soff[s], o1 = soff[s + 1];
  const int w = (int)((o1 - o0) >> 6);  // wave-uniform panel width
  const int64_t base = o0 + (p & 63);
  int row = p;
  bool live = p < n;
  if (MODE == CSR_GS) {
    row = live ? rowid[p] : -1;
    live = row >= 0;
  }
  // column of row's diagonal: row itself, or row + dshift when x is a rank's
  // halo-extended vector and the columns are numbered in it (multi-GPU shards)
  const int drow = row + dshift;
  double fi = 0.0, xi = 0.0;
  if (live) {
    if (MODE != CSR_SPMV) fi = NT ? __builtin_nontemporal_load(f + row) : f[row];
    if (MODE == CSR_JACOBI || MODE == CSR_GS) xi = x[drow];
    if (MODE == CSR_JACOBI_P) xi = corrected(row);
  }
  double acc = (MODE == CSR_RESID) ? fi : 0.0;
  double diag = 0.0;
  // one pass over U entries: all loads of the pass are issued before the first use
  auto pass = [&](int j0, auto UU) {
    constexpr int UN = decltype(UU)::value;
    int32_t c[UN];
    double v[UN], xx[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int j = j0 + u < w ? j0 + u : j0;
      const int64_t at = base + ((int64_t)j << 6);
      if (IDX16) {
        const int d = NT ? __builtin_nontemporal_load(scol16 + at) : scol16[at];
        c[u] = (d == -32768) ? -1 : drow + d;
      } else {
        c[u] = NT ? __builtin_nontemporal_load(scol + at) : scol[at];
      }
      v[u] = NT ? __builtin_nontemporal_load(sval + at) : sval[at];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (MODE == CSR_JACOBI_P) xx[u] = corrected(c[u] >= 0 ? c[u] : 0);
      else xx[u] = x[c[u] >= 0 ? c[u] : 0];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (j0 + u < w && c[u] >= 0) {
        if (MODE == CSR_RESID) {
          acc -= v[u] * xx[u];
        } else if (MODE == CSR_JACOBI || MODE == CSR_GS || MODE == CSR_JACOBI_P) {
          if (c[u] == drow) diag = v[u];
          else acc += v[u] * xx[u];
        } else {
          acc += v[u] * xx[u];
        }
      }
    }
  };
  // the pass length follows the PANEL's width (wave-uniform branch): a 7-wide
  // panel walked with a 9-entry pass would issue 2 dead loads + 2 dead gathers
  if (w <= 3) pass(0, std::integral_constant<int, 3>{});
  else if (w <= 5) pass(0, std::integral_constant<int, 5>{});
  else if (w <= 7) pass(0, std::integral_constant<int, 7>{});
  else if (w <= 9) pass(0, std::integral_constant<int, 9>{});
  else
    for (int j0 = 0; j0 < w; j0 += 8) pass(j0, std::integral_constant<int, 8>{});
  if (live) {
    double res;
    bool store = true;
    if (MODE == CSR_RESID || MODE == CSR_SPMV) {
      res = acc;
    } else if (MODE == CSR_JACOBI || MODE == CSR_JACOBI_P) {
      res = (diag == 0.0) ? xi : xi + omega * ((fi - acc) / diag - xi);
    } else if (MODE == CSR_GS) {
      res = (fi - acc) / diag;  // smoother.hpp:136
      store = diag != 0.0;
    } else {
      const double d = fi - acc;
      res = d * d;
    }
    if (store) {
      if (NT && MODE != CSR_GS) __builtin_nontemporal_store(res, out + row);
      else out[row] = res;
    }
  }
}

template <int MODE>
static hipError_t launch_sell_mode(int64_t n, int idx16, const int64_t* soff,
                                   const void* scol, const double* sval, const double* x,
                                   const double* f, double* out, double omega,
                                   const int32_t* rowid, int64_t row0, int64_t count,
                                   const double* uH, int64_t nH, hipStream_t st,
                                   int64_t diag_shift = 0) {
  const unsigned grid = (unsigned)((count + 255) / 256);
  // idx16: bit 0 = 16-bit relative columns, bit 1 = non-temporal matrix stream
#define AMG_SELL_LAUNCH(I16, NTF)                                                              \
  hipLaunchKernelGGL((sell_kernel<MODE, I16, NTF>), dim3(grid), dim3(256), 0, st, (int)n, soff, \
                     scol, sval, x, f, out, omega, rowid, (int)row0, uH, (int)nH, (int)diag_shift)
  switch (idx16 & 3) {
    case 0: AMG_SELL_LAUNCH(false, false); break;
    case 1: AMG_SELL_LAUNCH(true, false); break;
    case 2: AMG_SELL_LAUNCH(false, true); break;
    default: AMG_SELL_LAUNCH(true, true); break;
  }
#undef AMG_SELL_LAUNCH
  return hipGetLastError();
}
hipError_t launch_sell(int mode, int64_t n, int idx16, const int64_t* soff,
                       const void* scol, const double* sval, const double* x,
                       const double* f, double* out, double omega, int64_t diag_shift,
                       hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (n >= ((int64_t)1 << 31) - 256 || diag_shift >= ((int64_t)1 << 30)) return hipErrorInvalidValue;
  switch (mode) {
    case CSR_RESID: return launch_sell_mode<CSR_RESID>(n, idx16, soff, scol, sval, x, f, out, omega, nullptr, 0, n, nullptr, 0, st, diag_shift);
    case CSR_JACOBI: return launch_sell_mode<CSR_JACOBI>(n, idx16, soff, scol, sval, x, f, out, omega, nullptr, 0, n, nullptr, 0, st, diag_shift);
    case CSR_SPMV: return launch_sell_mode<CSR_SPMV>(n, idx16, soff, scol, sval, x, f, out, omega, nullptr, 0, n, nullptr, 0, st, diag_shift);
    case CSR_RSSQ: return launch_sell_mode<CSR_RSSQ>(n, idx16, soff, scol, sval, x, f, out, omega, nullptr, 0, n, nullptr, 0, st, diag_shift);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_sell_jacobi_prolong(int64_t n, int idx16, const int64_t* soff,
                                      const void* scol, const double* sval, const double* u,
                                      const double* uH, int64_t nH, const double* f, double* out,
                                      double omega, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (n >= ((int64_t)1 << 31) - 256 || nH < 1) return hipErrorInvalidValue;
  return launch_sell_mode<CSR_JACOBI_P>(n, idx16, soff, scol, sval, u, f, out, omega, nullptr,
                                        0, n, uH, nH, st);
}

hipError_t launch_sell_gs_color(int64_t n_storage, int /*idx16: rows are permuted*/,
                                const int64_t* soff, const int32_t* scol, const double* sval,
                                const int32_t* rowid,
                                int64_t row0, int64_t count, const double* f, double* u,
                                hipStream_t st) {
  if (count <= 0) return hipSuccess;
  if (n_storage >= ((int64_t)1 << 31) - 256 || (row0 & 63)) return hipErrorInvalidValue;
  // the kernel's row bound is the END OF THIS COLOUR: a 256-thread workgroup must not run on
  // into the storage rows of the next colour (they reference this colour's rows)
  const int64_t end = row0 + count < n_storage ? row0 + count : n_storage;
  return launch_sell_mode<CSR_GS>(end, 0, soff, scol, sval, u, f, u, 1.0, rowid, row0,
                                  count, nullptr, 0, st);
}

static int g_dict_rows_per_lane = 2;
static int g_xcd_map = 1;
// ---------------------------------------------------------------- K-Dict -----
// The same four operations on a dictionary-coded matrix (host_setup.hpp: DictMat).
// Lane per row; the row's structure is ONE 8- or 16-byte load (byte j = code of
// entry j, 0xFF = none), the <= 255 (column offset, value) pairs live in LDS.  The
// matrix stream shrinks from 10-12 bytes per entry to 8/16 bytes per row, which on
// the 4096^2 fine level leaves f, x and the output as the bulk of the HBM traffic.
// Entry order (ascending column) and the decoded doubles are the stored ones, so
// results are bit-identical to K-SELL / K-CSR.
// R = rows per lane (1 or 2).  R = 2: a lane owns rows 2t and 2t+1, so codes, f and
// the output move as 16-byte (32-byte for two-word codes) lane accesses and twice as
// many bytes are in flight per wave (the launcher checks the 16-byte alignment).
// (Persistent workgroups that prefetch their next tile were measured 10 % SLOWER on the
// 4096^2 fine level, 108 vs 99 us, and are not kept.)
// Row types (second-level dictionary): the rows of such a matrix also repeat as whole
// code words (2-D Poisson hierarchy: <= 12 distinct rows per level, 3-D: <= 40), so
// when there are at most 255 distinct words a row is ONE byte indexing a word table in
// LDS (type 255 = empty row), and the matrix stream is 1 byte per row.
template <int WORDS, int R>
struct DictStream {  // streamed operands of R rows of one lane
  uint64_t cw[R][2];
  uint32_t ty;  // row types, byte r = row r
  double fi[R], xi[R];
  bool live[R];
};
template <int MODE, int WORDS, bool NT, int R>
__device__ __forceinline__ void dict_fetch(DictStream<WORDS, R>& s, int row0, int n,
                                           const uint64_t* __restrict__ codes,
                                           const uint8_t* __restrict__ rtype,
                                           const double* __restrict__ f, const double* x,
                                           int dshift) {
  typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
  typedef double f64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s.live[r] = row0 + r < n;
    s.cw[r][0] = s.cw[r][1] = ~(uint64_t)0;
    s.fi[r] = s.xi[r] = 0.0;
  }
  s.ty = 0xFFFFu;  // both rows empty; unpacked in dict_expand, after the other loads are out
  if (R == 2 && s.live[R - 1]) {
    if (rtype) {  // wave-uniform
      s.ty = *reinterpret_cast<const uint16_t*>(rtype + row0);
    } else {
      const u64x2* vp = reinterpret_cast<const u64x2*>(codes + (int64_t)row0 * WORDS);
      const u64x2 a = NT ? __builtin_nontemporal_load(vp) : *vp;
      if (WORDS == 2) {
        const u64x2 b2 = NT ? __builtin_nontemporal_load(vp + 1) : vp[1];
        s.cw[0][0] = a.x; s.cw[0][1] = a.y; s.cw[R - 1][0] = b2.x; s.cw[R - 1][1] = b2.y;
      } else {
        s.cw[0][0] = a.x; s.cw[R - 1][0] = a.y;
      }
    }
    if (MODE != CSR_SPMV) {
      const f64x2* fp = reinterpret_cast<const f64x2*>(f + row0);
      const f64x2 t = NT ? __builtin_nontemporal_load(fp) : *fp;
      s.fi[0] = t.x; s.fi[R - 1] = t.y;
    }
    if (MODE == CSR_JACOBI) { s.xi[0] = x[row0 + dshift]; s.xi[R - 1] = x[row0 + 1 + dshift]; }
  } else if (s.live[0]) {
    if (rtype) {
      s.ty = 0xFF00u | rtype[row0];
    } else {
      const uint64_t* cp = codes + (int64_t)row0 * WORDS;
      if (WORDS == 2) {
        const u64x2* vp = reinterpret_cast<const u64x2*>(cp);
        const u64x2 t = NT ? __builtin_nontemporal_load(vp) : *vp;
        s.cw[0][0] = t.x; s.cw[0][1] = t.y;
      } else {
        s.cw[0][0] = NT ? __builtin_nontemporal_load(cp) : cp[0];
      }
    }
    if (MODE != CSR_SPMV) s.fi[0] = NT ? __builtin_nontemporal_load(f + row0) : f[row0];
    if (MODE == CSR_JACOBI) s.xi[0] = x[row0 + dshift];
  }
}
// word table of the row types: 256 entries of WORDS words, entry 255 = all 0xFF (host pads)
template <int WORDS>
__device__ __forceinline__ void dict_stage_words(uint64_t* wtab, const uint64_t* __restrict__ rwords) {
#pragma unroll
  for (int k = 0; k < WORDS; ++k) wtab[threadIdx.x * WORDS + k] = rwords[threadIdx.x * WORDS + k];
}
template <int WORDS, int R>
__device__ __forceinline__ void dict_expand(DictStream<WORDS, R>& s, const uint64_t* wtab) {
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int k = 0; k < WORDS; ++k) s.cw[r][k] = wtab[((s.ty >> (8 * r)) & 0xFFu) * WORDS + k];
}
// One 32-byte LDS entry per code.  Entry 255 (= no entry) is all zero: its gather reads
// the row's diagonal column (always a valid address).
// Jacobi sweeps stage  a = value of an OFF-diagonal pair (else +0.0),
//                      d = value of a diagonal pair (else +0.0);
// the other modes      a = value.
struct DictEntry {
  double a;
  double d;
  int32_t off8;  // BYTE offset of the column from the row's diagonal column
  int32_t pad[3];
};
template <int MODE>
__device__ __forceinline__ void dict_stage_table(DictEntry* tab, const int32_t* __restrict__ doff,
                                                 const double* __restrict__ dval, int ntab) {
  const int t = threadIdx.x;  // 256 threads, 256 entries
  const double v = t < ntab ? dval[t] : 0.0;
  const int32_t o = t < ntab ? doff[t] : 0;
  DictEntry e;
  constexpr bool split = MODE == CSR_JACOBI || MODE == CSR_GS;
  e.a = (split && o == 0) ? 0.0 : v;
  e.d = (split && o == 0 && t < ntab) ? v : 0.0;
  e.off8 = o * 8;
  e.pad[0] = e.pad[1] = e.pad[2] = 0;
  tab[t] = e;
}
// decode + gather + row arithmetic of the R rows of one lane.  The kernels are issue
// bound as much as bandwidth bound (a wave64 VALU instruction occupies its SIMD for 4
// cycles), so the row walk is branch-free and short: per entry one byte extract, one
// 16-byte + one 4-byte LDS read, one 32-bit add for the byte offset (the gather uses
// the scalar-base + 32-bit-offset addressing mode), mul, add.
//
// Skipped terms and bit-identity.  A Jacobi sweep adds (+0.0) * x for "no entry" slots
// and for the diagonal pair, i.e. +-0.0 for finite x: acc + (+-0.0) == acc because a sum
// that starts at +0.0 is never -0.0, and diag + (+0.0) == diag, so for finite vectors
// the result has the same bits as skipping those terms (the x read there is the row's
// own u_i, which enters the result anyway).  The other modes skip "no entry" slots
// with a select: acc - (+0.0) == acc holds for every acc.
template <int MODE, int WORDS, int UN, int R>
__device__ __forceinline__ void dict_rows(const DictStream<WORDS, R>& s, int row0,
                                          const DictEntry* tab, const double* x, double omega,
                                          int dshift, double (&res)[R]) {
  uint32_t c8[R][UN];
  double v[R][UN], vd[R][UN], xx[R][UN];
  bool ok[R][UN];
  const char* xb = reinterpret_cast<const char*>(x);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const uint32_t drow8 = s.live[r] ? (uint32_t)(row0 + r + dshift) * 8u : 0u;
    const uint32_t w32[4] = {(uint32_t)s.cw[r][0], (uint32_t)(s.cw[r][0] >> 32),
                             (uint32_t)s.cw[r][1], (uint32_t)(s.cw[r][1] >> 32)};
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const uint32_t code = (w32[u >> 2] >> (8 * (u & 3))) & 0xFFu;
      ok[r][u] = code != 0xFFu;
      c8[r][u] = drow8 + (uint32_t)tab[code].off8;
      v[r][u] = tab[code].a;
      vd[r][u] = (MODE == CSR_JACOBI || MODE == CSR_GS) ? tab[code].d : 0.0;
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int u = 0; u < UN; ++u)
      xx[r][u] = *reinterpret_cast<const double*>(xb + c8[r][u]);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    double acc = (MODE == CSR_RESID) ? s.fi[r] : 0.0;
    double diag = 0.0;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      double t = v[r][u] * xx[r][u];
      if (MODE == CSR_JACOBI || MODE == CSR_GS) {
        acc += t;
        diag += vd[r][u];
      } else {
        // the empty asm keeps the product unconditional, so that clang does not turn
        // the select into an exec-masked branch with the gather (and its wait) inside
        asm volatile("" : "+v"(t));
        t = ok[r][u] ? t : 0.0;
        if (MODE == CSR_RESID) acc -= t;
        else acc += t;
      }
    }
    if (MODE == CSR_RESID || MODE == CSR_SPMV) {
      res[r] = acc;
    } else if (MODE == CSR_JACOBI || MODE == CSR_GS) {
      // The division runs for every row (a missing diagonal divides by 1 and is then
      // discarded): with it inside `diag != 0` clang puts the whole row walk, gathers and
      // their waits included, into that branch and the rows of a lane run one after
      // the other instead of overlapping.
      const bool nod = diag == 0.0;
      double q = (s.fi[r] - acc) / (nod ? 1.0 : diag);  // smoother.hpp:136
      asm volatile("" : "+v"(q));
      if (MODE == CSR_GS) res[r] = nod ? s.xi[r] : q;
      else res[r] = nod ? s.xi[r] : s.xi[r] + omega * (q - s.xi[r]);
    } else {
      const double d = s.fi[r] - acc;
      res[r] = d * d;
    }
  }
}
// Wave-uniform STENCIL rows (the interior of a 3-D level: almost every wave).  When all 128 rows of
// a wave share one row type whose columns are the 7-point pattern {-M, -m, -1, 0, +1, +m, +M} or the
// 15-point pattern {-M, -m, 0, +m, +M} x {-1, 0, +1} (m, M even), a lane's two consecutive rows need
// 12 / 20 distinct entries of x that sit in aligned pairs: 7 / 15 load instructions (16-byte pairs
// plus the two outer singles of a triple) instead of the 16 / 32 eight-byte gathers of dict_rows --
// the sweeps of these levels are bound by the gathers through the vector L1, not by HBM (traffic
// 1.01 x must-move at 3.9 / 2.4 TB/s) and not by the decode (a scalar-table path with the same
// gathers measured no faster).  Values come through scalar loads from a per-type table (utd: per
// type {off-diagonal value or +0.0 [16], value [16], diagonal}, uti: {offset in rows [16], mask of
// slots in use, pattern 7 / 15 / 0}), built by the host from the same pairs.  Same products in the
// same slot order: Jacobi adds (+0.0) x for the diagonal slot exactly as dict_rows does -> same
// bits.  A mixed wave, or one whose type has no pattern, takes dict_rows (dict_stencil_can).
// (wave-uniform) does this wave qualify?  (Deciding this per WORKGROUP before the LDS tables are
// staged, so that a workgroup of stencil waves skips the staging -- 12 KB of loads and LDS stores
// per 512 rows --, was measured no faster: 512^3 +3.2 % against +5.2 % for this form; the early
// workgroup-wide vote waits for the row types with nothing else in flight.)
template <int UN>
__device__ __forceinline__ bool dict_stencil_can(uint32_t ty, const bool (&live)[2],
                                                 const int32_t* __restrict__ uti) {
  constexpr int PAT = UN == 7 ? 7 : 15;
  const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ty & 0xFFu));
  const bool ok = ty == (t0 | (t0 << 8)) && live[0] && live[1];
  if (t0 == 255u || __builtin_amdgcn_ballot_w64(!ok) != 0) return false;
  return uti[t0 * 18u + 17u] == PAT;
}
template <int MODE, int UN>
__device__ __forceinline__ void dict_rows_stencil(uint32_t ty, const double (&fi)[2],
                                                  const double (&xi)[2], int row0,
                                                  const double* __restrict__ utd,
                                                  const int32_t* __restrict__ uti, const double* x,
                                                  double omega, double (&res)[2]) {
  static_assert(MODE == CSR_JACOBI || MODE == CSR_RESID, "fast path of the two hot modes");
  static_assert(UN == 7 || UN == 16, "7-point rows / 15-point rows");
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  constexpr int PAT = UN == 7 ? 7 : 15;
  const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ty & 0xFFu));
  const int32_t* ti = uti + t0 * 18u;
  const double* td = utd + t0 * 33u + (MODE == CSR_RESID ? 16u : 0u);
  const double* xr = x + row0;  // row0 is even, x 16-byte aligned, m and M even
  double xx[2][PAT];
  if (PAT == 7) {
    const int g[4] = {0, 1, 5, 6};  // slots of -M, -m, +m, +M
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f64x2 v = *reinterpret_cast<const f64x2*>(xr + ti[g[k]]);
      xx[0][g[k]] = v.x;
      xx[1][g[k]] = v.y;
    }
    const f64x2 c = *reinterpret_cast<const f64x2*>(xr);
    const double lo = xr[-1], hi = xr[2];
    xx[0][2] = lo;  xx[0][3] = c.x; xx[0][4] = c.y;
    xx[1][2] = c.x; xx[1][3] = c.y; xx[1][4] = hi;
  } else {
#pragma unroll
    for (int k = 0; k < 5; ++k) {  // triple k: slots 3k, 3k+1, 3k+2 around the offset of slot 3k+1
      const double* xc = xr + ti[3 * k + 1];
      const f64x2 c = *reinterpret_cast<const f64x2*>(xc);
      const double lo = xc[-1], hi = xc[2];
      xx[0][3 * k] = lo;  xx[0][3 * k + 1] = c.x; xx[0][3 * k + 2] = c.y;
      xx[1][3 * k] = c.x; xx[1][3 * k + 1] = c.y; xx[1][3 * k + 2] = hi;
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    double acc = (MODE == CSR_RESID) ? fi[r] : 0.0;
#pragma unroll
    for (int u = 0; u < PAT; ++u) {
      const double t = td[u] * xx[r][u];
      if (MODE == CSR_JACOBI) acc += t;   // td[u] = +0.0 for the diagonal slot (dict_stage_table<CSR_JACOBI>)
      else acc -= t;                      // every slot of the pattern is in use
    }
    if (MODE == CSR_RESID) {
      res[r] = acc;
    } else {
      const double diag = td[32];
      const bool nod = diag == 0.0;
      const double q = (fi[r] - acc) / (nod ? 1.0 : diag);  // smoother.hpp:136
      res[r] = nod ? xi[r] : xi[r] + omega * (q - xi[r]);
    }
  }
}
template <int WORDS, bool NT, int R>
__device__ __forceinline__ void dict_store(const DictStream<WORDS, R>& s, int row0,
                                           const double (&res)[R], double* out) {
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  if (R == 2 && s.live[R - 1]) {
    f64x2 t;
    t.x = res[0]; t.y = res[R - 1];
    f64x2* op = reinterpret_cast<f64x2*>(out + row0);
    if (NT) __builtin_nontemporal_store(t, op);
    else *op = t;
  } else if (s.live[0]) {
    if (NT) __builtin_nontemporal_store(res[0], out + row0);
    else out[row0] = res[0];
  }
}

// Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an L2).  A banded
// operator re-reads x at +-(half-bandwidth) rows, i.e. a few tiles away, so every XCD
// gets ONE contiguous run of tiles: the re-reads then hit the L2 that fetched the line
// (measured on level 1 of the 4096^2 hierarchy: x fetched 2.1x -> see DESIGN.md).
// on >= 16: a WIDE band (3-D levels: x is re-read a whole grid plane = `on` tiles away, far beyond
// what an L2 keeps of a contiguous run).  Every XCD then takes the same slab of on / 8 tiles out of
// EVERY plane, plane after plane: the +-plane re-reads come back after one slab (a few hundred KB)
// of the XCD's own traffic, the +-line ones sit inside the slab.
__device__ __forceinline__ int xcd_tile(unsigned b, unsigned nb, int on) {
  if (!on) return (int)b;
  if (on >= 16) {
    const unsigned tp = (unsigned)on, s = tp >> 3;
    if (b >= nb / tp * tp) return (int)b;  // the last, partial plane: plain order
    const unsigned k = b & 7u, i = b >> 3, p = i / s;
    return (int)(p * tp + k * s + (i - p * s));
  }
  const unsigned k = b & 7u, i = b >> 3, q = nb >> 3, r = nb & 7u;
  return (int)(k * q + (k < r ? k : r) + i);
}
// (Two tiles per workgroup, i.e. the streamed operands of twice as many rows in flight per
// lane, measured SLOWER on the 4096^2 fine level: 99.5 vs 86.8 us, 90 VGPRs.)
template <int MODE, int WORDS, int UN, bool NT, int R>
__global__ __launch_bounds__(256) void dict_kernel(
    int n, const uint64_t* __restrict__ codes, const uint8_t* __restrict__ rtype,
    const uint64_t* __restrict__ rwords, const int32_t* __restrict__ doff,
    const double* __restrict__ dval, int ntab, const double* x, const double* __restrict__ f,
    double* out, double omega, int dshift, int xcd_map, const double* __restrict__ utd,
    const int32_t* __restrict__ uti) {
  __shared__ DictEntry tab[256];
  __shared__ uint64_t wtab[256 * WORDS];
  const int row0 = (xcd_tile(blockIdx.x, gridDim.x, xcd_map) * 256 + (int)threadIdx.x) * R;
  DictStream<WORDS, R> s;
  dict_fetch<MODE, WORDS, NT, R>(s, row0, n, codes, rtype, f, x, dshift);  // in flight while the tables are staged
  dict_stage_table<MODE>(tab, doff, dval, ntab);
  if (rtype) dict_stage_words<WORDS>(wtab, rwords);
  __syncthreads();
  double res[R];
  bool fast = false;
  if constexpr ((MODE == CSR_JACOBI || MODE == CSR_RESID) && R == 2 && (UN == 7 || UN == 16)) {
    if (utd) fast = dict_stencil_can<UN>(s.ty, s.live, uti);
    if (fast) dict_rows_stencil<MODE, UN>(s.ty, s.fi, s.xi, row0, utd, uti, x, omega, res);
  }
  if (!fast) {
    if (rtype) dict_expand<WORDS, R>(s, wtab);
    dict_rows<MODE, WORDS, UN, R>(s, row0, tab, x, omega, dshift, res);
  }
  dict_store<WORDS, NT, R>(s, row0, res, out);
}

// ---- fused forms for the true-Jacobi V-cycle on linear-interpolation levels --------
// Tiles of 256 R rows that OVERLAP by two rows (stride 256 R - 2): the row results of
// a tile are parked in LDS, and everything a coarse point j needs from the fine rows
// 2j, 2j+1, 2j+2 (restriction) or from its neighbour j-1 (prolongation) is then inside
// one tile.  The two duplicated rows are recomputed (0.4 %) and stored twice with the
// same bits.  Same per-row arithmetic and the same transfer expressions as the
// separate kernels, so the V-cycle is bit-identical with and without these fusions.
//
// (1) residual + restriction + first coarse Jacobi sweep (multigrid.hpp:272-282, then
// :268 on level l+1 whose u is zero): r = f - A u is written, f_H = R r is written, and
// the from-zero sweep of the coarse level (jacobi_from_zero_kernel) is written to uH1.
// Other smoothers (uH1 == nullptr): the coarse u is zero-filled instead (uH0, :278).
// restriction of a tile's residuals rs[0 .. 256 R) (rows tile * (256 R - 2) ...) and, when
// uH1 is given, the first coarse Jacobi sweep from zero; else the coarse u is zero-filled
template <int R>
__device__ __forceinline__ void dict_restrict_tail(int tile, int n, const double* rs, int nH,
                                                   double* __restrict__ fH,
                                                   const double* __restrict__ diagH,
                                                   double* __restrict__ uH1,
                                                   double* __restrict__ uH0, double omega) {
  for (int q = threadIdx.x; q < 128 * R - 1; q += 256) {
    const int j = tile * (128 * R - 1) + q;
    if (j >= nH) break;
    const int64_t i = 2 * (int64_t)j;  // linear_restrict_kernel, same guards and order
    double sum = 0.0;
    if (i < n) sum += 0.5 * rs[2 * q];
    if (i + 1 < n) sum += 1.0 * rs[2 * q + 1];
    if (i + 2 < n) sum += 0.5 * rs[2 * q + 2];
    fH[j] = sum;
    if (uH1) {
      const double xi = 0.0, acc = 0.0;  // jacobi_from_zero_kernel
      const double d = diagH[j];
      uH1[j] = (d == 0.0) ? xi : xi + omega * ((sum - acc) / d - xi);
    } else {
      uH0[j] = 0.0;  // multigrid.hpp:278
    }
  }
}
// prolongation of a tile's new coarse values rs[] into the finer level:
// uh_out[2j] = uh_in[2j] + (0.5 u_H[j-1] + 0.5 u_H[j]), uh_out[2j+1] = uh_in[2j+1] + u_H[j]
// (linear_prolong_add2_kernel).  The tile prolongs the coarse points q = 1 .. stride (q = 0
// too in the first tile); j = n is the virtual point whose fine rows only receive the left
// neighbour / + 0.0.  uh_in == uh_out is the in-place form.
template <int R>
__device__ __forceinline__ void dict_prolong_tail(int tile, int n, const double* rs, int n_h,
                                                  const double* uh_in, double* uh_out) {
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  const int stride = 256 * R - 2;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int q = (int)threadIdx.x * R + r;
    const int j = tile * stride + q;
    if ((q == 0 && tile != 0) || q > stride || j > n) continue;
    double t0 = 0.0, t1 = 0.0;
    const double b = (j < n) ? rs[q] : 0.0;
    if (j >= 1 && j - 1 < n) t0 += 0.5 * rs[q - 1];
    if (j < n) {
      t0 += 0.5 * b;
      t1 += 1.0 * b;
    }
    const int64_t i = 2 * (int64_t)j;
    if (i + 1 < n_h) {
      f64x2 u = *reinterpret_cast<const f64x2*>(uh_in + i);
      u.x = u.x + t0;
      u.y = u.y + t1;
      *reinterpret_cast<f64x2*>(uh_out + i) = u;
    } else if (i < n_h) {
      uh_out[i] = uh_in[i] + t0;
    }
  }
}
template <int WORDS, int UN, bool NT, int R>
__global__ __launch_bounds__(256) void dict_resid_restrict_kernel(
    int n, const uint64_t* __restrict__ codes, const uint8_t* __restrict__ rtype,
    const uint64_t* __restrict__ rwords, const int32_t* __restrict__ doff,
    const double* __restrict__ dval, int ntab, const double* x, const double* __restrict__ f,
    double* r_out, int nH, double* __restrict__ fH, const double* __restrict__ diagH,
    double* __restrict__ uH1, double* __restrict__ uH0, double omega, int xcd_map,
    const double* __restrict__ utd, const int32_t* __restrict__ uti) {
  __shared__ DictEntry tab[256];
  __shared__ uint64_t wtab[256 * WORDS];
  __shared__ double rs[256 * R];
  const int tile = xcd_tile(blockIdx.x, gridDim.x, xcd_map);
  const int row0 = tile * (256 * R - 2) + (int)threadIdx.x * R;
  DictStream<WORDS, R> s;
  dict_fetch<CSR_RESID, WORDS, NT, R>(s, row0, n, codes, rtype, f, x, 0);
  dict_stage_table<CSR_RESID>(tab, doff, dval, ntab);
  if (rtype) dict_stage_words<WORDS>(wtab, rwords);
  __syncthreads();
  double res[R];
  bool fast = false;
  if constexpr (R == 2 && (UN == 7 || UN == 16)) {
    if (utd) fast = dict_stencil_can<UN>(s.ty, s.live, uti);
    if (fast) dict_rows_stencil<CSR_RESID, UN>(s.ty, s.fi, s.xi, row0, utd, uti, x, omega, res);
  }
  if (!fast) {
    if (rtype) dict_expand<WORDS, R>(s, wtab);
    dict_rows<CSR_RESID, WORDS, UN, R>(s, row0, tab, x, omega, 0, res);
  }
  if (r_out) dict_store<WORDS, NT, R>(s, row0, res, r_out);  // nullptr: r is dead after this kernel
#pragma unroll
  for (int r = 0; r < R; ++r) rs[threadIdx.x * R + r] = s.live[r] ? res[r] : 0.0;
  __syncthreads();
  dict_restrict_tail<R>(tile, n, rs, nH, fH, diagH, uH1, uH0, omega);
}
// (2) Jacobi sweep on level H + prolongation of its result into the finer level
// (multigrid.hpp:300 on level l+1, then :294-296 on level l):
// u_h[2j] += 0.5 u_H[j-1] + 0.5 u_H[j], u_h[2j+1] += u_H[j] (linear_prolong_add2_kernel).
template <int WORDS, int UN, bool NT, int R>
__global__ __launch_bounds__(256) void dict_jacobi_prolong_kernel(
    int n, const uint64_t* __restrict__ codes, const uint8_t* __restrict__ rtype,
    const uint64_t* __restrict__ rwords, const int32_t* __restrict__ doff,
    const double* __restrict__ dval, int ntab, const double* x, const double* __restrict__ f,
    double* out, double omega, int n_h, const double* uh_in, double* uh_out, int xcd_map,
    const double* __restrict__ utd, const int32_t* __restrict__ uti) {
  __shared__ DictEntry tab[256];
  __shared__ uint64_t wtab[256 * WORDS];
  __shared__ double rs[256 * R];
  const int tile = xcd_tile(blockIdx.x, gridDim.x, xcd_map);
  const int stride = 256 * R - 2;
  const int row0 = tile * stride + (int)threadIdx.x * R;
  DictStream<WORDS, R> s;
  dict_fetch<CSR_JACOBI, WORDS, NT, R>(s, row0, n, codes, rtype, f, x, 0);
  dict_stage_table<CSR_JACOBI>(tab, doff, dval, ntab);
  if (rtype) dict_stage_words<WORDS>(wtab, rwords);
  __syncthreads();
  double res[R];
  bool fast = false;
  if constexpr (R == 2 && (UN == 7 || UN == 16)) {
    if (utd) fast = dict_stencil_can<UN>(s.ty, s.live, uti);
    if (fast) dict_rows_stencil<CSR_JACOBI, UN>(s.ty, s.fi, s.xi, row0, utd, uti, x, omega, res);
  }
  if (!fast) {
    if (rtype) dict_expand<WORDS, R>(s, wtab);
    dict_rows<CSR_JACOBI, WORDS, UN, R>(s, row0, tab, x, omega, 0, res);
  }
  dict_store<WORDS, NT, R>(s, row0, res, out);
#pragma unroll
  for (int r = 0; r < R; ++r) rs[threadIdx.x * R + r] = s.live[r] ? res[r] : 0.0;
  __syncthreads();
  dict_prolong_tail<R>(tile, n, rs, n_h, uh_in, uh_out);
}

// ---- two sweeps in one launch on small levels ----------------------------------------
// Below ~300 K rows a launch costs more than its work (~5 us each), and the band is narrow
// (half-bandwidth hb <= 64 rows).  A tile then computes its first Jacobi sweep on its rows
// PLUS hb rows on either side (recomputed by the neighbouring tiles, <= 25 %) into an LDS
// window, and the operation that follows reads its x from that window:
//   down: sweep + (residual + restriction + first coarse sweep)   [dict_pair_down_kernel]
//   up:   sweep + (sweep + prolongation into the finer level)     [dict_pair_up_kernel]
// Same row arithmetic (dict_rows) and the same tails, so the V-cycle stays bit-identical.
// R = 2 rows per lane, plain loads/stores.  w0 = tile start - hbw (hbw even >= hb).
constexpr int PAIR_HB = 66;                       // largest half-window, in rows
constexpr int PAIR_W = 512 + 2 * PAIR_HB;         // window capacity

// first stage: Jacobi sweep of the window rows [w0, w0 + 512 + 2 hbw) from global `a` into
// uw (0.0 for rows outside the matrix); rows of the tile are also stored to u_out if given
template <int WORDS, int UN>
__device__ __forceinline__ void dict_pair_stage1(int n, int w0, int hbw, const uint64_t* codes,
                                                 const uint8_t* rtype, const DictEntry* tabJ,
                                                 const uint64_t* wtab, const double* a,
                                                 const double* f, double omega, double* uw,
                                                 double* u_out) {
  const int W = 512 + 2 * hbw;
  for (int base = 0; base < W; base += 512) {  // 2 passes, the second one over the 2 hbw rows left
    const int lw = base + (int)threadIdx.x * 2;  // index in the window
    const int row0 = w0 + lw;
    DictStream<WORDS, 2> s;
    // rows before the matrix (first tile) or past the window: an empty stream, no loads.
    // (row0 is even, so a pair is either entirely before row 0 or not at all.)
    const bool inside = row0 >= 0 && lw < W;
    dict_fetch<CSR_JACOBI, WORDS, false, 2>(s, inside ? row0 : 0, inside ? n : 0, codes, rtype, f, a, 0);
    if (rtype) dict_expand<WORDS, 2>(s, wtab);
    double res[2];
    dict_rows<CSR_JACOBI, WORDS, UN, 2>(s, row0, tabJ, a, omega, 0, res);
    if (lw < W) {
      uw[lw] = s.live[0] ? res[0] : 0.0;
      uw[lw + 1] = s.live[1] ? res[1] : 0.0;
    }
    if (u_out && lw >= hbw && lw < hbw + 512) dict_store<WORDS, false, 2>(s, row0, res, u_out);
  }
}
template <int WORDS, int UN>
__global__ __launch_bounds__(256) void dict_pair_down_kernel(
    int n, const uint64_t* __restrict__ codes, const uint8_t* __restrict__ rtype,
    const uint64_t* __restrict__ rwords, const int32_t* __restrict__ doff,
    const double* __restrict__ dval, int ntab, const double* a, const double* __restrict__ f,
    double* u_out, double* r_out, int nH, double* __restrict__ fH,
    const double* __restrict__ diagH, double* __restrict__ uH1, double omega, int hbw,
    int xcd_map) {
  __shared__ DictEntry tabJ[256];
  __shared__ DictEntry tabR[256];
  __shared__ uint64_t wtab[256 * WORDS];
  __shared__ double uw[PAIR_W];
  __shared__ double rs[512];
  const int tile = xcd_tile(blockIdx.x, gridDim.x, xcd_map);
  const int t0 = tile * 510, w0 = t0 - hbw;
  dict_stage_table<CSR_JACOBI>(tabJ, doff, dval, ntab);
  dict_stage_table<CSR_RESID>(tabR, doff, dval, ntab);
  if (rtype) dict_stage_words<WORDS>(wtab, rwords);
  __syncthreads();
  dict_pair_stage1<WORDS, UN>(n, w0, hbw, codes, rtype, tabJ, wtab, a, f, omega, uw, u_out);
  __syncthreads();
  const int row0 = t0 + (int)threadIdx.x * 2;
  DictStream<WORDS, 2> s;
  dict_fetch<CSR_RESID, WORDS, false, 2>(s, row0, n, codes, rtype, f, a, 0);
  if (rtype) dict_expand<WORDS, 2>(s, wtab);
  double res[2];
  dict_rows<CSR_RESID, WORDS, UN, 2>(s, row0 - w0, tabR, uw, omega, 0, res);  // x = the LDS window
  if (r_out) dict_store<WORDS, false, 2>(s, row0, res, r_out);
  rs[threadIdx.x * 2] = s.live[0] ? res[0] : 0.0;
  rs[threadIdx.x * 2 + 1] = s.live[1] ? res[1] : 0.0;
  __syncthreads();
  dict_restrict_tail<2>(tile, n, rs, nH, fH, diagH, uH1, nullptr, omega);
}
template <int WORDS, int UN>
__global__ __launch_bounds__(256) void dict_pair_up_kernel(
    int n, const uint64_t* __restrict__ codes, const uint8_t* __restrict__ rtype,
    const uint64_t* __restrict__ rwords, const int32_t* __restrict__ doff,
    const double* __restrict__ dval, int ntab, const double* a, const double* __restrict__ f,
    double* u_out, double omega, int hbw, int n_h, const double* uh_in, double* uh_out,
    int xcd_map) {
  __shared__ DictEntry tabJ[256];
  __shared__ uint64_t wtab[256 * WORDS];
  __shared__ double uw[PAIR_W];
  __shared__ double rs[512];
  const int tile = xcd_tile(blockIdx.x, gridDim.x, xcd_map);
  const int t0 = tile * 510, w0 = t0 - hbw;
  dict_stage_table<CSR_JACOBI>(tabJ, doff, dval, ntab);
  if (rtype) dict_stage_words<WORDS>(wtab, rwords);
  __syncthreads();
  dict_pair_stage1<WORDS, UN>(n, w0, hbw, codes, rtype, tabJ, wtab, a, f, omega, uw, nullptr);
  __syncthreads();
  const int row0 = t0 + (int)threadIdx.x * 2;
  DictStream<WORDS, 2> s;
  dict_fetch<CSR_RESID, WORDS, false, 2>(s, row0, n, codes, rtype, f, a, 0);  // types + f; xi from the window
  s.xi[0] = uw[row0 - w0];
  s.xi[1] = uw[row0 - w0 + 1];
  if (rtype) dict_expand<WORDS, 2>(s, wtab);
  double res[2];
  dict_rows<CSR_JACOBI, WORDS, UN, 2>(s, row0 - w0, tabJ, uw, omega, 0, res);
  dict_store<WORDS, false, 2>(s, row0, res, u_out);
  rs[threadIdx.x * 2] = s.live[0] ? res[0] : 0.0;
  rs[threadIdx.x * 2 + 1] = s.live[1] ? res[1] : 0.0;
  __syncthreads();
  dict_prolong_tail<2>(tile, n, rs, n_h, uh_in, uh_out);
}

// ---- K-Patch: the level's whole down-leg / up-leg in ONE pass over the level ------------
// On the big levels (0 and 1 of the 4096^2 hierarchy) every sweep streams f, x and the
// output again, 25 B per row and sweep.  The 2+2 true-Jacobi cycle touches such a level
// five times on the way down (sweep, sweep, residual + restriction) and three times on the
// way up (prolongation, sweep, sweep).  These kernels do each leg in one launch (temporal
// blocking): a workgroup owns a 2-D PATCH of the level -- PATCH_TH grid lines x PATCH_TW
// columns of the flat index (row = line * m + column; m = the pitch of the band: 4096,
// 2048, ...) -- loads it once with the halo the later stages need (one ring of lines and
// columns per stage), and runs the stages LDS -> LDS with a barrier in between.  The halo
// cells are recomputed by the neighbouring patches (same expressions, same bits):
//   down: [sweep 1] -> sweep 2 (stored) -> residual -> restriction + first coarse sweep
//   up:   u + P u_H (formed while loading) -> sweep 1 -> sweep 2 (stored)
// All addressing is flat-index arithmetic: cell (line lj, column li) of a patch is row
// (j0 + lj) m + i0 + li even when i0 + li runs past the end of a grid line (the flat-index
// smear of the reference's coarsening), so a column offset o = dj m + di of the
// pair table is the LDS offset dj * pitch + di.  Row arithmetic = dict_rows' (entries in
// ascending column order, diagonal split off for Jacobi, IEEE divide), transfers =
// dict_restrict_tail / linear_prolong_add2_kernel: the V-cycle stays bit-identical.
// The row's structure comes from its TYPE alone: the host expands type -> code word ->
// pairs into one table of (value, diagonal value, LDS offset) per type and slot, so a
// cell costs one byte of matrix and no decode chain.  Interior rows share one type: the
// table reads of a wave are broadcasts.
constexpr int PATCH_TW = 64;                 // columns of a patch (output)
constexpr int PATCH_TH = 42;                 // lines of a patch (output)
constexpr int PATCH_EH = PATCH_TH + 6;       // lines held, 3 rings
constexpr int PATCH_EC = PATCH_TW + 8;       // columns held: [-4, TW + 4)
constexpr int PATCH_NT = 432;                // threads = 6 line groups x 72 columns (4 workgroups per CU)
constexpr int PATCH_K = PATCH_EH * PATCH_EC / PATCH_NT;  // cells per thread: 8 consecutive lines
constexpr int PATCH_BUF = (PATCH_EH + 2) * PATCH_EC;     // + one guard line above and below
constexpr int PATCH_MAXTAB = 192;            // table capacity: (types + 1) x slots
static_assert(PATCH_EH * PATCH_EC == PATCH_K * PATCH_NT && PATCH_NT % PATCH_EC == 0, "patch geometry");

struct PatchJ { double a, d; };              // off-diagonal value (else +0.0), diagonal value (else +0.0)
struct PatchR { double a; int32_t loff, ok; };  // value, LDS offset, slot in use

// The PATCH_K cells one thread owns: column li of the consecutive lines lj0 .. lj0 + 7; cell k
// sits at LDS index cell0 + k * PATCH_EC and is flat row row0 + k m.  Adjacent lanes own
// adjacent columns (conflict-free 8-byte LDS accesses, coalesced global ones), and a thread's
// cells are vertical neighbours, so the wave-uniform path walks them with a sliding 3 x 3
// register window: 3 LDS reads per cell instead of one per matrix entry.
struct PatchCells {
  int cell0, row0, lj0, li;
  double f[PATCH_K];
  uint32_t ty[PATCH_K];     // row type, clamped to the all-absent table row `ntypes`
  bool live[PATCH_K];       // row inside the matrix
  bool uniform;             // every cell of the wave has the type tu (then the table comes
  uint32_t tu;              // through scalar loads instead of per-lane LDS reads)
};

// The table of ONE row type held in scalar registers (the wave-uniform fast path: in the
// interior of a level every row has the same type).  Loaded once per workgroup through
// scalar loads from the per-type arrays the host lays out (solver.cpp: build_patch_table):
//   utabd: per type aj[UN] (off-diagonal values, +0.0 elsewhere), ar[UN] (values), diag
//   utabi: per type lo[UN] (LDS offsets), jmask (slots with an off-diagonal entry), rmask
//          (slots in use)
// slot s = (dj + 1) * 3 + (di + 1) of the 3 x 3 neighbourhood = ascending column offset
// (-m-1, -m, -m+1, -1, 0, +1, m-1, m, m+1): walking the slots in order IS the reference's
// ascending-column summation order (smoother.hpp:101-117).
//   utabd: per type wj[9] (off-diagonal values, +0.0 where there is no entry and on the
//          diagonal), wr[9] (values, +0.0 where there is no entry), diag
//   utabi: per type rmask (slots in use), corners (1 when a corner slot 0, 2, 6, 8 is in use)
struct PatchU {
  double wj[9], wr[9], diag;
#ifdef AMG_PATCH_FASTDIV
  double rdiag;
#endif
  uint32_t rmask, corners;
};
__device__ __forceinline__ void patch_load_u(PatchU& U, uint32_t tu, const double* __restrict__ utabd,
                                             const int32_t* __restrict__ utabi) {
  const double* d = utabd + (size_t)tu * 19;
  const int32_t* i = utabi + (size_t)tu * 2;
#pragma unroll
  for (int e = 0; e < 9; ++e) {
    U.wj[e] = d[e];
    U.wr[e] = d[9 + e];
  }
  U.diag = d[18];
#ifdef AMG_PATCH_FASTDIV
  U.rdiag = 1.0 / (U.diag == 0.0 ? 1.0 : U.diag);
#endif
  U.rmask = (uint32_t)i[0];
  U.corners = (uint32_t)i[1];
}
// Uniform-type evaluation of the thread's PATCH_K vertically consecutive cells with a sliding
// window: w[r][c] = value of line (k - 1 + r), column (li - 1 + c).  Slots that hold no entry
// have weight +0.0: adding (+0.0) x leaves a Jacobi accumulator's bits alone for finite x (see
// dict_rows); the residual selects them away.  CORNERS = false: the four corner slots are
// not even read (5-point level).
// MASK >= 0: the type's slots in use are exactly MASK (the interior row types of a Poisson
// hierarchy: 0x0BA = 5-point level 0, 0x1D7 = level 1 with its exact-zero +-1 entries pruned,
// 0x1FF = 9-point levels): absent slots are not even multiplied, and the residual needs no
// selects.  Same bits as the generic form (MASK < 0): a Jacobi accumulator starts at +0.0 and
// never becomes -0.0, so skipping a (+0.0) x term changes nothing for finite x; the residual's
// generic form selects absent terms away, which is what not computing them does.
template <bool RESID, bool CORNERS, bool GS = false, int MASK = -1>
__device__ __forceinline__ void patch_eval_u(const double* buf, int cell0, const PatchU& U,
                                             const double (&f)[PATCH_K], double omega,
                                             double (&res)[PATCH_K]) {
  const double* p = buf + cell0 - PATCH_EC;  // line above the first cell
  if (!CORNERS && !RESID && U.diag == 0.0) {  // rows without a diagonal keep their value (smoother.hpp:133)
#pragma unroll
    for (int k = 0; k < PATCH_K; ++k) res[k] = p[(k + 1) * PATCH_EC];
    return;
  }
  // (5-point rows: the test above is wave-uniform and sits OUTSIDE the cell loop; inside it, it
  // cuts the loop into one basic block per cell and the eight independent dependency chains --
  // LDS read, sums, IEEE division -- run one after the other instead of interleaved.  For the
  // 9-point rows that is what the 72-register budget of four workgroups per CU wants: there the
  // test stays inside the loop, measured 10 % faster on levels 1-3.)
  double w[3][3];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) w[r][c] = (CORNERS || c == 1 || r == 1) ? p[r * PATCH_EC + c - 1] : 0.0;
#pragma unroll
  for (int k = 0; k < PATCH_K; ++k) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      w[2][c] = (CORNERS || c == 1) ? p[(k + 2) * PATCH_EC + c - 1] : 0.0;
    if (!CORNERS) {  // left / right of line k + 1 become the middle row's of the next cell
      w[2][0] = p[(k + 2) * PATCH_EC - 1];
      w[2][2] = p[(k + 2) * PATCH_EC + 1];
    }
    const double xi = w[1][1];
    if (RESID) {
      double acc = f[k];
#pragma unroll
      for (int s9 = 0; s9 < 9; ++s9) {
        const int r = s9 / 3, c = s9 % 3;
        if (!CORNERS && r != 1 && c != 1) continue;
        if (MASK >= 0) {
          if ((MASK >> s9) & 1) acc -= U.wr[s9] * w[r][c];
        } else {
          double t = U.wr[s9] * w[r][c];
          t = ((U.rmask >> s9) & 1u) ? t : 0.0;
          acc -= t;
        }
      }
      res[k] = acc;
    } else {
      double acc = 0.0;
#pragma unroll
      for (int s9 = 0; s9 < 9; ++s9) {
        const int r = s9 / 3, c = s9 % 3;
        if (s9 == 4 || (!CORNERS && r != 1 && c != 1)) continue;
        if (MASK >= 0 && !((MASK >> s9) & 1)) continue;
        acc += U.wj[s9] * w[r][c];
      }
      if (CORNERS && U.diag == 0.0) {
        res[k] = xi;
      } else {
#ifdef AMG_PATCH_FASTDIV
        // experiment: Markstein's correctly rounded quotient from the (wave-uniform) reciprocal
        const double nm = f[k] - acc;
        const double q0 = nm * U.rdiag;
        const double r0 = __builtin_fma(-U.diag, q0, nm);
        const double q1 = __builtin_fma(r0, U.rdiag, q0);
        const double r1 = __builtin_fma(-U.diag, q1, nm);
        const double q = __builtin_fma(r1, U.rdiag, q1);
#else
        const double q = (f[k] - acc) / U.diag;  // smoother.hpp:136
#endif
        res[k] = GS ? q : xi + omega * (q - xi);      // GS: the Gauss-Seidel update itself (K-SELL CSR_GS)
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      w[0][c] = w[1][c];
      w[1][c] = w[2][c];
    }
  }
}
// Row arithmetic of dict_rows, per-lane table entries from the LDS copy (mixed row types:
// level boundaries).
template <int UN, bool RESID, bool GS = false>
__device__ __forceinline__ double patch_eval(const double* buf, int cell, uint32_t type,
                                             const PatchJ* tabJ, const PatchR* tabR, double fi,
                                             double omega) {
  const int t0 = (int)type * UN;
  double xx[UN];
  if (RESID) {
    double av[UN];
    bool ok[UN];
#pragma unroll
    for (int e = 0; e < UN; ++e) {
      const PatchR r = tabR[t0 + e];
      av[e] = r.a;
      ok[e] = r.ok != 0;
      xx[e] = buf[cell + r.loff];
    }
    double acc = fi;
#pragma unroll
    for (int e = 0; e < UN; ++e) {
      double t = av[e] * xx[e];
      t = ok[e] ? t : 0.0;
      acc -= t;
    }
    return acc;
  }
  double aj[UN], dj[UN];
#pragma unroll
  for (int e = 0; e < UN; ++e) {
    const PatchJ j = tabJ[t0 + e];
    aj[e] = j.a; dj[e] = j.d;
    xx[e] = buf[cell + tabR[t0 + e].loff];
  }
  double acc = 0.0, diag = 0.0;
  const double xi = buf[cell];
#pragma unroll
  for (int e = 0; e < UN; ++e) {
    acc += aj[e] * xx[e];
    diag += dj[e];
  }
  const bool nod = diag == 0.0;
  const double q = (fi - acc) / (nod ? 1.0 : diag);  // smoother.hpp:136
  return nod ? xi : (GS ? q : xi + omega * (q - xi));
}

// one stage over the region lines [l0, l1) x columns [c0, c1), IN PLACE in the workgroup's one
// LDS buffer: every thread evaluates its PATCH_K cells into registers (branch-free, so the
// LDS reads overlap), the workgroup meets at a barrier, then the results are written back
// (one buffer instead of two: four workgroups of 432 threads fit a CU instead of one, which is what these
// latency-bound stages need).  Cells outside the region or outside the matrix keep their
// value; ZERO: such cells inside the region are set to 0.0 instead (the residual that the
// restriction reads).  out (optional): rows of the patch proper also go to global memory.
template <int UN, int UM, bool RESID, bool NT, bool ZERO>
__device__ __forceinline__ void patch_stage(const PatchCells& pc, int m, int ntypes, double* buf,
                                            const PatchU& U, const PatchJ* tabJ, const PatchR* tabR,
                                            double omega, int l0, int l1, int c0, int c1, double* out) {
  const bool inc = pc.li >= c0 && pc.li < c1;
  double res[PATCH_K];
  bool did[PATCH_K], inr[PATCH_K];
#pragma unroll
  for (int k = 0; k < PATCH_K; ++k) {
    const int lj = pc.lj0 + k;
    inr[k] = inc && lj >= l0 && lj < l1;
    did[k] = inr[k] && pc.live[k];
  }
  if (pc.uniform && U.rmask == (uint32_t)UM) {
    // interior of the level (UM = the slots its row type uses, chosen at launch): one row type
    // for the whole wave, weights in scalar registers; cells outside the region are evaluated
    // with it too (their reads stay inside the guard lines) and dropped.  Every other wave
    // (level boundaries) takes the per-lane table.
    patch_eval_u<RESID, (UM & 0x145) != 0, false, (UM & 0x145) ? -1 : UM>(buf, pc.cell0, U, pc.f, omega, res);
  } else {
#pragma unroll
    for (int k = 0; k < PATCH_K; ++k)
      res[k] = patch_eval<UN, RESID>(buf, pc.cell0 + k * PATCH_EC,
                                     did[k] ? pc.ty[k] : (uint32_t)ntypes, tabJ, tabR, pc.f[k], omega);
  }
  lds_barrier();  // everybody has read the old values
  const bool outc = out != nullptr && pc.li >= 0 && pc.li < PATCH_TW;
#pragma unroll
  for (int k = 0; k < PATCH_K; ++k) {
    if (did[k]) buf[pc.cell0 + k * PATCH_EC] = res[k];
    else if (ZERO && inr[k]) buf[pc.cell0 + k * PATCH_EC] = 0.0;
    const int lj = pc.lj0 + k;
    if (outc && did[k] && lj >= 0 && lj < PATCH_TH) {
      double* op = out + (pc.row0 + k * m);
      if (NT) __builtin_nontemporal_store(res[k], op);
      else *op = res[k];
    }
  }
}

// The smoothed patch leaves through LDS: 64 consecutive threads write one 512-byte line of the
// patch, so every store of a wave covers whole, aligned 128-byte lines.  (Stored straight from
// the stage's registers, a line was split 60 + 4 entries between two waves -- the thread map
// has 72 columns -- and the PMC write traffic was 1.5 x the vector.)
template <bool NT>
__device__ __forceinline__ void patch_copy_out(const double* buf, double* __restrict__ out, int n, int m,
                                               int j0, int i0) {
  for (int q = threadIdx.x; q < PATCH_TH * PATCH_TW; q += PATCH_NT) {
    const int lj = q / PATCH_TW, li = q - lj * PATCH_TW;
    const int64_t row = (int64_t)(j0 + lj) * m + i0 + li;
    if (row < (int64_t)n) {
      const double v = buf[(lj + 4) * PATCH_EC + li + 4];
      if (NT) __builtin_nontemporal_store(v, out + row);
      else out[row] = v;
    }
  }
}

struct __attribute__((aligned(8))) PatchPair { double x, y; };  // 16-byte load, 8-byte aligned
// PROLONG: the loaded vector is x + P uH (up-leg); else plain x.
// UNI: every row of the patch and its halo has the row type `tf` (patch_tile_flags_kernel found
// that out at setup): no row-type loads, no bounds checks, the wave-uniform path for all waves.
template <bool PROLONG, bool UNI>
__device__ __forceinline__ void patch_load(PatchCells& pc, uint32_t tf, int n, int m, int j0, int i0,
                                           const double* __restrict__ x,
                                           const double* __restrict__ f,
                                           const uint8_t* __restrict__ rtype, int ntypes,
                                           const double* __restrict__ uH, int nH, double* buf) {
  const int g = (int)threadIdx.x / PATCH_EC, col = (int)threadIdx.x - g * PATCH_EC;
  const int le = g * PATCH_K;            // first of the thread's PATCH_K consecutive lines
  pc.lj0 = le - 3;
  pc.li = col - 4;
  pc.cell0 = (le + 1) * PATCH_EC + col;  // + 1: guard line
  const int64_t r0 = (int64_t)(j0 + pc.lj0) * m + i0 + pc.li;
  pc.row0 = (int)(r0 < -((int64_t)1 << 30) ? -((int64_t)1 << 30) : r0);
  uint32_t mism = 0;
  double xv[PATCH_K];
#pragma unroll
  for (int k = 0; k < PATCH_K; ++k) {
    const int64_t r64 = r0 + (int64_t)k * m;
    pc.live[k] = UNI || (r64 >= 0 && r64 < (int64_t)n);
    const int row = pc.live[k] ? (int)r64 : 0;
    xv[k] = x[row];
    pc.f[k] = f[row];
    pc.ty[k] = UNI ? tf : (uint32_t)rtype[row];
  }
#pragma unroll
  for (int k = 0; k < PATCH_K; ++k) {
    const int row = pc.row0 + k * m;
    double x0 = pc.live[k] ? xv[k] : 0.0;
    if (PROLONG) {  // linear_prolong_add_kernel, same guards and order, written with selects
      const int j = pc.live[k] ? (row >> 1) : 0;
      const bool odd = (row & 1) != 0;
      const bool a_ok = pc.live[k] && !odd && j >= 1 && j - 1 < nH;
      const bool b_ok = pc.live[k] && j < nH;
      // uH[j - 1] and uH[j] as ONE 16-byte load of the pair (jc - 1, jc), jc = j clamped to
      // [1, nH - 1] (nH >= 2 on a patch level): half the load instructions of two gathers
      const int jc = j < 1 ? 1 : (j > nH - 1 ? nH - 1 : j);
      const PatchPair pr = *reinterpret_cast<const PatchPair*>(uH + (jc - 1));
      const double a = (j == jc) ? pr.x : pr.y;   // j - 1 == jc - 1, or j - 1 == jc (j == nH)
      const double b = (j == jc) ? pr.y : pr.x;   // j == jc, or j == jc - 1 (j == 0)
      double t = 0.0;
      t = a_ok ? t + 0.5 * a : t;
      t = b_ok ? t + (odd ? 1.0 : 0.5) * b : t;
      x0 = pc.live[k] ? x0 + t : x0;
    }
    buf[pc.cell0 + k * PATCH_EC] = x0;
    const uint32_t nty = (uint32_t)ntypes;
    pc.ty[k] = (pc.live[k] && pc.ty[k] < nty) ? pc.ty[k] : nty;  // 255 = empty row -> absent row
    pc.f[k] = pc.live[k] ? pc.f[k] : 0.0;
  }
  if (UNI) {
    pc.tu = tf;
    pc.uniform = true;
    return;
  }
  pc.tu = (uint32_t)__builtin_amdgcn_readfirstlane((int)pc.ty[0]);
#pragma unroll
  for (int k = 0; k < PATCH_K; ++k) mism |= pc.ty[k] ^ pc.tu;
  pc.uniform = __builtin_amdgcn_ballot_w64(mism != 0) == 0 && pc.tu < (uint32_t)ntypes;
}

// flag[tile] = the row type shared by EVERY row a patch kernel loads for the tile (lines
// -3 .. TH+3, columns -4 .. TW+4 of the flat index, all inside the matrix), else 255.
__global__ __launch_bounds__(256) void patch_tile_flags_kernel(int n, int m, int px_count,
                                                               const uint8_t* __restrict__ rtype,
                                                               int ntypes, uint8_t* __restrict__ flag) {
  const int tile = blockIdx.x;
  const int py = tile / px_count, px = tile - py * px_count;
  const int64_t base = (int64_t)(py * PATCH_TH - 3) * m + px * PATCH_TW - 4;
  const int64_t first = base < 0 ? 0 : (base < n ? base : n - 1);
  const uint32_t t0 = rtype[first];
  int ok = base >= 0 && t0 < (uint32_t)ntypes;
  for (int q = threadIdx.x; q < PATCH_EH * PATCH_EC; q += 256) {
    const int le = q / PATCH_EC, c = q - le * PATCH_EC;
    const int64_t r = base + (int64_t)le * m + c;
    ok = ok && r >= 0 && r < (int64_t)n && rtype[r < 0 ? 0 : (r < n ? r : n - 1)] == t0;
  }
  ok = __syncthreads_and(ok);
  if (threadIdx.x == 0) flag[tile] = ok ? (uint8_t)t0 : (uint8_t)255;
}
hipError_t launch_patch_tile_flags(int64_t n, int64_t m, const uint8_t* rtype, int ntypes, uint8_t* flag,
                                   int64_t* n_tiles, hipStream_t st) {
  if (!patch_geometry_ok(n, m)) return hipErrorInvalidValue;
  const int64_t lines = (n + m - 1) / m;
  const int pxc = (int)(m / PATCH_TW);
  const int64_t tiles = (lines + PATCH_TH - 1) / PATCH_TH * pxc;
  if (n_tiles) *n_tiles = tiles;
  if (!flag) return hipSuccess;
  hipLaunchKernelGGL(patch_tile_flags_kernel, dim3((unsigned)tiles), dim3(256), 0, st, (int)n, (int)m, pxc,
                     rtype, ntypes, flag);
  return hipGetLastError();
}

// cflag[tile] = 1 when every coarse row the down-leg of the tile produces (c = i / 2 for the even
// fine rows i of the patch) has the diagonal dref, bit for bit.
__global__ __launch_bounds__(256) void patch_coarse_flags_kernel(int n, int m, int px_count, int nH,
                                                                 const double* __restrict__ diagH, double dref,
                                                                 uint8_t* __restrict__ cflag) {
  const int tile = blockIdx.x;
  const int py = tile / px_count, px = tile - py * px_count;
  const int j0 = py * PATCH_TH, i0 = px * PATCH_TW;
  int ok = 1;
  for (int q = threadIdx.x; q < PATCH_TH * (PATCH_TW / 2); q += 256) {
    const int lj = q / (PATCH_TW / 2), cx = q - lj * (PATCH_TW / 2);
    const int64_t i = (int64_t)(j0 + lj) * m + i0 + 2 * cx;
    const int64_t c = i >> 1;
    if (i >= (int64_t)n || c >= nH) continue;
    ok = ok && __double_as_longlong(diagH[c]) == __double_as_longlong(dref);
  }
  ok = __syncthreads_and(ok);
  if (threadIdx.x == 0) cflag[tile] = ok ? 1 : 0;
}
hipError_t launch_patch_coarse_flags(int64_t n, int64_t m, int64_t nH, const double* diagH, double dref,
                                     uint8_t* cflag, hipStream_t st) {
  if (!patch_geometry_ok(n, m) || !diagH || !cflag) return hipErrorInvalidValue;
  const int64_t lines = (n + m - 1) / m;
  const int pxc = (int)(m / PATCH_TW);
  const int64_t tiles = (lines + PATCH_TH - 1) / PATCH_TH * pxc;
  hipLaunchKernelGGL(patch_coarse_flags_kernel, dim3((unsigned)tiles), dim3(256), 0, st, (int)n, (int)m, pxc,
                     (int)nH, diagH, dref, cflag);
  return hipGetLastError();
}

__device__ __forceinline__ void patch_stage_tables(PatchJ* tabJ, PatchR* tabR,
                                                   const double* __restrict__ ptab, int nent) {
  // ptab: nent x {aJ, d, aR, loff (as double; -1e9 = unused slot)}; slots past nent: absent
  for (int t = threadIdx.x; t < PATCH_MAXTAB; t += PATCH_NT) {
    PatchJ j = {0.0, 0.0};
    PatchR r = {0.0, 0, 0};
    if (t < nent) {
      j.a = ptab[4 * t];
      j.d = ptab[4 * t + 1];
      r.a = ptab[4 * t + 2];
      const double lo = ptab[4 * t + 3];
      r.ok = lo > -1.0e8 ? 1 : 0;
      r.loff = r.ok ? (int)lo : 0;
    }
    tabJ[t] = j;
    tabR[t] = r;
  }
}
// guard lines: finite values for the reads of dropped cells
__device__ __forceinline__ void patch_clear_guards(double* buf) {
  for (int t = threadIdx.x; t < 2 * PATCH_EC; t += PATCH_NT)
    buf[t < PATCH_EC ? t : PATCH_BUF - 2 * PATCH_EC + t] = 0.0;
}

// common prologue: tables, guards and the uniform type's table (scalar loads)
__device__ __forceinline__ void patch_prologue(const PatchCells& pc, double* buf, PatchJ* tabJ,
                                               PatchR* tabR, const double* __restrict__ ptab,
                                               int nent, const double* __restrict__ utabd,
                                               const int32_t* __restrict__ utabi, PatchU& U) {
  patch_stage_tables(tabJ, tabR, ptab, nent);
  patch_clear_guards(buf);
  patch_load_u(U, pc.uniform ? pc.tu : 0u, utabd, utabi);
}

// FIRST: the input is the level's u and both pre-sweeps run here (level 0); else the input
// is the result of the first sweep (done by the finer level's kernel) and one sweep runs.
// (Tried and dropped: at most 1024 workgroups per launch, each walking several patches of its XCD's
// run with the tables staged once -- the loop alone cost 20 % (165 vs 134 us on level 0), with
// 1024 workgroups 204 us: the slots of a CU standing empty 40 % of the time
// (SQ_WAVE_CYCLES / (slots x duration) = 0.59) is not workgroup turnover.)
#ifdef AMG_PATCH_STAMPS
// DIAGNOSTIC BUILD ONLY (never shipped, never timed): per workgroup the 100 MHz wall clock at the
// phase boundaries of the down-leg and the hardware id of the CU it ran on, into a buffer of its
// own (tools/patch_stamps.py reads it: phase shares, workgroups resident per CU, dispatch gaps).
__device__ unsigned long long* g_patch_stamps = nullptr;
__device__ __forceinline__ void patch_stamp(int k) {
  if (threadIdx.x == 0 && g_patch_stamps) {
    unsigned long long t = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    g_patch_stamps[(size_t)blockIdx.x * 8 + k] = t;
    if (k == 0) {
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
      const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID
      g_patch_stamps[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
    }
  }
}
hipError_t debug_set_patch_stamps(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_patch_stamps), &p, sizeof(p));
}
#define PATCH_STAMP(k) patch_stamp(k)
#else
#define PATCH_STAMP(k)
#endif
template <int UN, int UM, bool FIRST, bool NT>
// (four workgroups per CU = 7 waves per SIMD = 72 registers: measured against 6 and 5 waves again in
// round 3 -- level-0 down-leg 131-135 us at 7, 138-144 at 6, 142-143 at 5.  The down-legs of the
// 7- and 9-point levels spill 2-7 registers at 72 and run at 6 waves (80 registers, three
// workgroups per CU): cycle 1288 / 1290 / 1291 -> 1315 / 1305 / 1308 V-cycles/s in alternating runs;
// their up-legs, which fit, stay at 7: 1296 / 1298 with both at 6.)
#ifndef AMG_PATCH_WAVES
#define AMG_PATCH_WAVES 7
#endif
__global__ __launch_bounds__(PATCH_NT, (UN == 5 ? AMG_PATCH_WAVES : AMG_PATCH_WAVES - 1)) void patch_down_kernel(
    int n, int m, int px_count, const uint8_t* __restrict__ rtype, const double* __restrict__ ptab,
    const double* __restrict__ utabd, const int32_t* __restrict__ utabi,
    int nent, int ntypes, const double* x, const double* __restrict__ f, double* u_out, double* r_out, int nH,
    double* __restrict__ fH, const double* __restrict__ diagH, double* __restrict__ uH1,
    double omega, int xcd_map, int py0, const uint8_t* __restrict__ tflag,
    const uint8_t* __restrict__ cflag, double dHu) {
  __shared__ double buf[PATCH_BUF];
  __shared__ PatchJ tabJ[PATCH_MAXTAB];
  __shared__ PatchR tabR[PATCH_MAXTAB];
  PATCH_STAMP(0);
  const int tile = xcd_tile(blockIdx.x, gridDim.x, xcd_map);
  const int py = tile / px_count, px = tile - py * px_count;
  const int j0 = (py + py0) * PATCH_TH, i0 = px * PATCH_TW;
  PatchCells pc;
  PatchU U;
  const uint32_t tf = tflag ? (uint32_t)tflag[(py + py0) * px_count + px] : 255u;
  // every coarse row under this patch has the diagonal dHu (launch_patch_coarse_flags found that
  // out at setup): the first coarse sweep then needs no load of the coarse diagonal at the very
  // end of the workgroup's life, where nothing is left to hide its latency behind
  const bool cuni = cflag && cflag[(py + py0) * px_count + px] != 0;
  if (tf != 255u) patch_load<false, true>(pc, tf, n, m, j0, i0, x, f, rtype, ntypes, nullptr, 0, buf);
  else patch_load<false, false>(pc, tf, n, m, j0, i0, x, f, rtype, ntypes, nullptr, 0, buf);
  patch_prologue(pc, buf, tabJ, tabR, ptab, nent, utabd, utabi, U);
  lds_barrier();
  PATCH_STAMP(1);
  if (FIRST) {
    patch_stage<UN, UM, false, NT, false>(pc, m, ntypes, buf, U, tabJ, tabR, omega, -2, PATCH_TH + 2, -2,
                                          PATCH_TW + 3, nullptr);
    lds_barrier();
  }
  PATCH_STAMP(2);
  patch_stage<UN, UM, false, NT, false>(pc, m, ntypes, buf, U, tabJ, tabR, omega, -1, PATCH_TH + 1, -1,
                                        PATCH_TW + 2, nullptr);
  lds_barrier();
  PATCH_STAMP(3);
  patch_copy_out<NT>(buf, u_out, n, m, j0, i0);  // the residual stage below only reads until its barrier
  // residual; rows outside the matrix read as 0.0 for the restriction (ZERO)
  patch_stage<UN, UM, true, NT, true>(pc, m, ntypes, buf, U, tabJ, tabR, omega, 0, PATCH_TH, 0, PATCH_TW + 1,
                                      r_out);
  lds_barrier();
  PATCH_STAMP(4);
  const double* rsb = buf;
  // restriction + first coarse sweep: coarse row c <-> even fine row 2c of the patch
  for (int q = threadIdx.x; q < PATCH_TH * (PATCH_TW / 2); q += PATCH_NT) {
    const int lj = q / (PATCH_TW / 2), cx = q - lj * (PATCH_TW / 2);
    const int64_t i = (int64_t)(j0 + lj) * m + i0 + 2 * cx;   // fine row 2c
    const int64_t c = i >> 1;
    if (i >= (int64_t)n || c >= nH) continue;
    const double* rs = rsb + (lj + 4) * PATCH_EC + 2 * cx + 4;
    double sum = 0.0;  // dict_restrict_tail / linear_restrict_kernel, same guards and order
    if (i < n) sum += 0.5 * rs[0];
    if (i + 1 < n) sum += 1.0 * rs[1];
    if (i + 2 < n) sum += 0.5 * rs[2];
    fH[c] = sum;
    const double xi = 0.0, acc = 0.0;  // jacobi_from_zero_kernel
    const double d = cuni ? dHu : diagH[c];
    uH1[c] = (d == 0.0) ? xi : xi + omega * ((sum - acc) / d - xi);
  }
  PATCH_STAMP(5);
}

template <int UN, int UM, bool NT>
__global__ __launch_bounds__(PATCH_NT, AMG_PATCH_WAVES) void patch_up_kernel(
    int n, int m, int px_count, const uint8_t* __restrict__ rtype, const double* __restrict__ ptab,
    const double* __restrict__ utabd, const int32_t* __restrict__ utabi,
    int nent, int ntypes, const double* x, const double* __restrict__ f, const double* __restrict__ uH, int nH,
    double* u_out, double omega, int xcd_map, int py0, const uint8_t* __restrict__ tflag) {
  __shared__ double buf[PATCH_BUF];
  __shared__ PatchJ tabJ[PATCH_MAXTAB];
  __shared__ PatchR tabR[PATCH_MAXTAB];
  const int tile = xcd_tile(blockIdx.x, gridDim.x, xcd_map);
  const int py = tile / px_count, px = tile - py * px_count;
  const int j0 = (py + py0) * PATCH_TH, i0 = px * PATCH_TW;
  PatchCells pc;
  PatchU U;
  const uint32_t tf = tflag ? (uint32_t)tflag[(py + py0) * px_count + px] : 255u;
  if (tf != 255u) patch_load<true, true>(pc, tf, n, m, j0, i0, x, f, rtype, ntypes, uH, nH, buf);
  else patch_load<true, false>(pc, tf, n, m, j0, i0, x, f, rtype, ntypes, uH, nH, buf);
  patch_prologue(pc, buf, tabJ, tabR, ptab, nent, utabd, utabi, U);
  lds_barrier();
  patch_stage<UN, UM, false, NT, false>(pc, m, ntypes, buf, U, tabJ, tabR, omega, -1, PATCH_TH + 1, -1,
                                        PATCH_TW + 1, nullptr);
  lds_barrier();
  patch_stage<UN, UM, false, NT, false>(pc, m, ntypes, buf, U, tabJ, tabR, omega, 0, PATCH_TH, 0, PATCH_TW,
                                        nullptr);
  lds_barrier();
  patch_copy_out<NT>(buf, u_out, n, m, j0, i0);
}

// ---- multicolour Gauss-Seidel on a patch (colours laid out on the 2 x 2 cells of the grid) ----
// One colour of a Gauss-Seidel half-sweep as a patch stage: cells of colour `c` inside the
// region take (f - sum of off-diagonal terms) / diagonal from the CURRENT neighbours (which all
// have other colours), every other cell keeps its value.  cbits: bits 2k, 2k+1 = colour of cell k.
template <int UN, int UM>
__device__ __forceinline__ void patch_stage_color(const PatchCells& pc, int ntypes, double* buf,
                                                  const PatchU& U, const PatchJ* tabJ, const PatchR* tabR,
                                                  int l0, int l1, int c0, int c1, uint32_t cbits,
                                                  uint32_t c) {
  const bool inc = pc.li >= c0 && pc.li < c1;
  double res[PATCH_K];
  bool did[PATCH_K];
#pragma unroll
  for (int k = 0; k < PATCH_K; ++k) {
    const int lj = pc.lj0 + k;
    did[k] = inc && lj >= l0 && lj < l1 && pc.live[k] && ((cbits >> (2 * k)) & 3u) == c;
  }
  // A wave none of whose cells has the stage's colour has nothing to do (the colours of the line
  // ends on level 1: almost every wave; -9 % on that level).  Otherwise every cell is evaluated and
  // the other colours' results are dropped: evaluating only the lines that hold the colour (every
  // other one, for the colourings by lines and by products) was measured no faster -- a stage is
  // bound by its LDS round trips and barriers, not by the arithmetic -- and any finer branching
  // (a ballot per cell) 35 % slower.
  bool any = false;
#pragma unroll
  for (int k = 0; k < PATCH_K; ++k) any |= did[k];
  if (__builtin_amdgcn_ballot_w64(any) == 0) return;
  if (pc.uniform && U.rmask == (uint32_t)UM) {
    constexpr bool CN = (UM & 0x145) != 0;
    constexpr int MK = (UM & 0x145) ? -1 : UM;
    patch_eval_u<false, CN, true, MK>(buf, pc.cell0, U, pc.f, 1.0, res);
  } else {
#pragma unroll
    for (int k = 0; k < PATCH_K; ++k)
      res[k] = patch_eval<UN, false, true>(buf, pc.cell0 + k * PATCH_EC,
                                           did[k] ? pc.ty[k] : (uint32_t)ntypes, tabJ, tabR, pc.f[k], 1.0);
  }
  // cells of one colour do not read each other: no barrier between the reads and the writes
#pragma unroll
  for (int k = 0; k < PATCH_K; ++k)
    if (did[k]) buf[pc.cell0 + k * PATCH_EC] = res[k];
}

// Up to three colour stages on the loaded patch -- a piece of the symmetric pass 0 .. nc-1, nc-1 .. 0
// (a colour that directly follows itself is dropped by the host: its rows read no row of their own
// colour, so the repeat would write the bits that are already there) -- in the frame of the Jacobi
// kernels above:
//   !TAIL: like the up-leg: [u + P u_H while loading when PROLONG], up to 3 stages, store;
//    TAIL: like the level-0 down-leg: up to 2 stages, store, residual, restriction (f_H; the
//          coarse u is zeroed, multigrid.hpp:278).
// stages: bits 0-2 = number of stages, bits 4+2s, 5+2s = colour of stage s.
// colour(row) = ctab entry (line & 1) * 6 + column class, 2 bits each; column classes: 0, 1, m-2,
// m-1 (the line ends, whose rows differ) and the even / odd columns between them (what the greedy
// colouring yields on these stencils: the checkerboard on the 5-point level; line parity with
// colours of their own at the line ends on level 1, where only the lines above and below are
// coupled; the product colouring on the 9-point levels -- verified on the host against the
// colouring itself).  Same row arithmetic as the colour kernels
// (dict_rows<CSR_GS>): ascending-column sum of the off-diagonal terms, IEEE divide.
template <int UN, int UM, bool NT, bool PROLONG, bool TAIL>
// (Register budget: with the stage loop, the colour bits and the residual stage the kernel does not
// fit the 72 registers of four workgroups per CU -- 26-118 spilled registers, and the scratch
// traffic showed as 1.6x the write bytes in the PMC record.  At 96 registers (two workgroups per
// CU) nothing spills: 8192^2 multicolour 190.9 -> 203.2 V-cycles/s, 4096^2 576.9 -> 607.0; three
// workgroups / 80 registers: 196.5 / 593.1.  The Jacobi kernels, which fit 72, are fastest at four.)
#ifndef AMG_RB_WAVES
#define AMG_RB_WAVES 5
#endif
__global__ __launch_bounds__(PATCH_NT, AMG_RB_WAVES) void patch_rb_kernel(
    int n, int m, int px_count, const uint8_t* __restrict__ rtype, const double* __restrict__ ptab,
    const double* __restrict__ utabd, const int32_t* __restrict__ utabi, int nent, int ntypes,
    const double* x, const double* __restrict__ f, const double* __restrict__ uH, int nH, double* u_out,
    double* r_out, double* __restrict__ fH, double* __restrict__ uH_zero, uint32_t stages, uint32_t ctab,
    int xcd_map, int py0, const uint8_t* __restrict__ tflag) {
  __shared__ double buf[PATCH_BUF];
  __shared__ PatchJ tabJ[PATCH_MAXTAB];
  __shared__ PatchR tabR[PATCH_MAXTAB];
  const int tile = xcd_tile(blockIdx.x, gridDim.x, xcd_map);
  const int py = tile / px_count, px = tile - py * px_count;
  const int j0 = (py + py0) * PATCH_TH, i0 = px * PATCH_TW;
  PatchCells pc;
  PatchU U;
  const uint32_t tf = tflag ? (uint32_t)tflag[(py + py0) * px_count + px] : 255u;
  if (tf != 255u) patch_load<PROLONG, true>(pc, tf, n, m, j0, i0, x, f, rtype, ntypes, uH, nH, buf);
  else patch_load<PROLONG, false>(pc, tf, n, m, j0, i0, x, f, rtype, ntypes, uH, nH, buf);
  patch_prologue(pc, buf, tabJ, tabR, ptab, nent, utabd, utabi, U);
  uint32_t cbits = 0;
  {
    // the cell is flat row (j0 + lj) * m + i0 + li: beyond a line's end that is a row of the next / previous line
    const int col = i0 + pc.li;
    const uint32_t line = (uint32_t)(j0 + pc.lj0 + (col < 0 ? -1 : (col >= m ? 1 : 0)) + 64);
    const int tc = col < 0 ? col + m : (col >= m ? col - m : col);
    const uint32_t cls = tc == 0 ? 0u : (tc == 1 ? 1u : (tc == m - 2 ? 2u : (tc == m - 1 ? 3u : 4u + ((uint32_t)tc & 1u))));
#pragma unroll
    for (int k = 0; k < PATCH_K; ++k)
      cbits |= ((ctab >> (2u * (((line + (uint32_t)k) & 1u) * 6u + cls))) & 3u) << (2 * k);
  }
  lds_barrier();
  constexpr int E = TAIL ? 1 : 0;  // the residual stage needs one more ring (and the restriction one more column)
  const int nst = (int)(stages & 7u);
  for (int sg = 0; sg < nst; ++sg) {
    const int ext = nst - 1 - sg + E;  // rings the later stages still read
    patch_stage_color<UN, UM>(pc, ntypes, buf, U, tabJ, tabR, -ext, PATCH_TH + ext, -ext, PATCH_TW + ext + E,
                              cbits, (stages >> (4 + 2 * sg)) & 3u);
    lds_barrier();
  }
  patch_copy_out<NT>(buf, u_out, n, m, j0, i0);
  if (TAIL) {
    patch_stage<UN, UM, true, NT, true>(pc, m, ntypes, buf, U, tabJ, tabR, 1.0, 0, PATCH_TH, 0, PATCH_TW + 1,
                                    r_out);
    lds_barrier();
    for (int q = threadIdx.x; q < PATCH_TH * (PATCH_TW / 2); q += PATCH_NT) {
      const int lj = q / (PATCH_TW / 2), cx = q - lj * (PATCH_TW / 2);
      const int64_t i = (int64_t)(j0 + lj) * m + i0 + 2 * cx;   // fine row 2c
      const int64_t c = i >> 1;
      if (i >= (int64_t)n || c >= nH) continue;
      const double* rs = buf + (lj + 4) * PATCH_EC + 2 * cx + 4;
      double sum = 0.0;  // dict_restrict_tail / linear_restrict_kernel, same guards and order
      if (i < n) sum += 0.5 * rs[0];
      if (i + 1 < n) sum += 1.0 * rs[1];
      if (i + 2 < n) sum += 0.5 * rs[2];
      fH[c] = sum;
      if (uH_zero) uH_zero[c] = 0.0;  // multigrid.hpp:278
    }
  }
}
int patch_un(int un) { return un <= 5 ? 5 : un <= 7 ? 7 : 9; }
bool patch_geometry_ok(int64_t n, int64_t m) {
  return m >= 2 * PATCH_TW && (m % PATCH_TW) == 0 && n >= m && n < ((int64_t)1 << 31) - 4 * m - 64;
}
// tiles of the lines [line_lo, line_hi) (line_hi < 0: all); *py0 = first tile row
static unsigned patch_grid(int64_t n, int64_t m, int64_t line_lo, int64_t line_hi, int* px_count,
                           int* py0) {
  const int64_t lines = (n + m - 1) / m;
  if (line_hi < 0 || line_hi > lines) line_hi = lines;
  if (line_lo < 0) line_lo = 0;
  if (line_lo > line_hi) line_lo = line_hi;
  *px_count = (int)(m / PATCH_TW);
  *py0 = (int)(line_lo / PATCH_TH);
  return (unsigned)(((line_hi + PATCH_TH - 1) / PATCH_TH - *py0) * *px_count);
}
int patch_tile_lines() { return PATCH_TH; }
// kernel kinds: (row width of the per-lane path, slots of the interior row type -- the scalar path)
//   (5, 0x0BA) the 5-point level 0; (7, 0x1D7) / (9, 0x1D7) level 1, whose +-1 entries are exact
//   zeros and pruned (its line ends keep rows of 9); (9, 0x1FF) the 9-point levels
int patch_default_umask(int un) { return un <= 5 ? 0x0BA : (un <= 7 ? 0x1D7 : 0x1FF); }
// the mask of the kernel kind patch_dispatch picks for (un, umask)
int patch_kind_umask(int un, int umask) { return (un > 7 && un <= 9 && umask == 0x1D7) ? 0x1D7 : patch_default_umask(un); }
template <class F>
static hipError_t patch_dispatch(int un, int umask, bool nt, F&& go) {
  using std::integral_constant;
  auto with_nt = [&](auto U, auto M) {
    if (nt) go(U, M, integral_constant<bool, true>{});
    else go(U, M, integral_constant<bool, false>{});
  };
  if (un <= 5) with_nt(integral_constant<int, 5>{}, integral_constant<int, 0x0BA>{});
  else if (un <= 7) with_nt(integral_constant<int, 7>{}, integral_constant<int, 0x1D7>{});
  else if (un <= 9 && umask == 0x1D7) with_nt(integral_constant<int, 9>{}, integral_constant<int, 0x1D7>{});
  else if (un <= 9) with_nt(integral_constant<int, 9>{}, integral_constant<int, 0x1FF>{});
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
hipError_t launch_patch_down(bool first, int64_t n, int64_t m, const PatchRef& P, const double* x,
                             const double* f, double* u_out, double* r_out, int64_t nH, double* fH,
                             const double* diagH, double* uH1, double omega, hipStream_t st,
                             int64_t line_lo, int64_t line_hi) {
  if (!patch_geometry_ok(n, m) || !P.rtype || !P.ptab || !P.utabd || !P.utabi || (P.ntypes + 1) * patch_un(P.un) > PATCH_MAXTAB ||
      P.nent != P.ntypes * patch_un(P.un) || !fH || !diagH || !uH1 || !u_out || u_out == x)
    return hipErrorInvalidValue;
  int pxc = 0, py0 = 0;
  const unsigned grid = patch_grid(n, m, line_lo, line_hi, &pxc, &py0);
  if (grid == 0) return hipSuccess;
  const int xm = g_xcd_map ? 1 : 0;  // halo lines of neighbouring patches meet in one L2
  return patch_dispatch(P.un, P.umask, P.nt != 0, [&](auto U, auto M, auto NTF) {
    if (first)
      hipLaunchKernelGGL((patch_down_kernel<decltype(U)::value, decltype(M)::value, true, decltype(NTF)::value>), dim3(grid),
                         dim3(PATCH_NT), 0, st, (int)n, (int)m, pxc, P.rtype, P.ptab, P.utabd, P.utabi, P.nent, P.ntypes, x, f, u_out,
                         r_out, (int)nH, fH, diagH, uH1, omega, xm, py0, P.tflag, P.cflag, P.dHu);
    else
      hipLaunchKernelGGL((patch_down_kernel<decltype(U)::value, decltype(M)::value, false, decltype(NTF)::value>), dim3(grid),
                         dim3(PATCH_NT), 0, st, (int)n, (int)m, pxc, P.rtype, P.ptab, P.utabd, P.utabi, P.nent, P.ntypes, x, f, u_out,
                         r_out, (int)nH, fH, diagH, uH1, omega, xm, py0, P.tflag, P.cflag, P.dHu);
  });
}
hipError_t launch_patch_up(int64_t n, int64_t m, const PatchRef& P, const double* x, const double* f,
                           const double* uH, int64_t nH, double* u_out, double omega,
                           hipStream_t st, int64_t line_lo, int64_t line_hi) {
  if (!patch_geometry_ok(n, m) || !P.rtype || !P.ptab || !P.utabd || !P.utabi || (P.ntypes + 1) * patch_un(P.un) > PATCH_MAXTAB ||
      P.nent != P.ntypes * patch_un(P.un) || !uH || !u_out || u_out == x)
    return hipErrorInvalidValue;
  int pxc = 0, py0 = 0;
  const unsigned grid = patch_grid(n, m, line_lo, line_hi, &pxc, &py0);
  if (grid == 0) return hipSuccess;
  const int xm = g_xcd_map ? 1 : 0;  // halo lines of neighbouring patches meet in one L2
  return patch_dispatch(P.un, P.umask, P.nt != 0, [&](auto U, auto M, auto NTF) {
    hipLaunchKernelGGL((patch_up_kernel<decltype(U)::value, decltype(M)::value, decltype(NTF)::value>), dim3(grid),
                       dim3(PATCH_NT), 0, st, (int)n, (int)m, pxc, P.rtype, P.ptab, P.utabd, P.utabi, P.nent, P.ntypes, x, f, uH,
                       (int)nH, u_out, omega, xm, py0, P.tflag);
  });
}
// the halo a patch is loaded with (3 lines above and below, 4 columns left and right; the lines
// beyond are guard lines of zeros) bounds the dependent stages of one launch: stage s of S reads
// ring S - s, and the residual + restriction take a ring and a column of their own
int patch_rb_max_stages(bool tail) { return tail ? 2 : 3; }
uint32_t patch_rb_stages(const int* colors, int count) {
  uint32_t v = (uint32_t)count & 7u;
  for (int q = 0; q < count && q < 4; ++q) v |= ((uint32_t)colors[q] & 3u) << (4 + 2 * q);
  return v;
}
hipError_t launch_patch_rb(bool prolong, bool tail, int64_t n, int64_t m, const PatchRef& P, const double* x,
                           const double* f, const double* uH, int64_t nH, double* u_out, double* r_out,
                           double* fH, double* uH_zero, uint32_t stages, uint32_t ctab, hipStream_t st,
                           int64_t line_lo, int64_t line_hi) {
  const int nst = (int)(stages & 7u);
  if (!patch_geometry_ok(n, m) || !P.rtype || !P.ptab || !P.utabd || !P.utabi || (P.ntypes + 1) * patch_un(P.un) > PATCH_MAXTAB ||
      P.nent != P.ntypes * patch_un(P.un) || !u_out || u_out == x || (prolong && (!uH || tail)) || (tail && !fH) || nH < 2 ||
      nst < 1 || nst > patch_rb_max_stages(tail) || (m & 1))
    return hipErrorInvalidValue;
  int pxc = 0, py0 = 0;
  const unsigned grid = patch_grid(n, m, line_lo, line_hi, &pxc, &py0);
  if (grid == 0) return hipSuccess;
  const int xm = g_xcd_map ? 1 : 0;
  return patch_dispatch(P.un, P.umask, P.nt != 0, [&](auto U, auto M, auto NTF) {
#define AMG_RB(PRO, TL)                                                                              \
  hipLaunchKernelGGL((patch_rb_kernel<decltype(U)::value, decltype(M)::value, decltype(NTF)::value, PRO, TL>), dim3(grid), \
                     dim3(PATCH_NT), 0, st, (int)n, (int)m, pxc, P.rtype, P.ptab, P.utabd, P.utabi,    \
                     P.nent, P.ntypes, x, f, uH, (int)nH, u_out, r_out, fH, uH_zero, stages, ctab, xm, py0,  \
                     P.tflag)
    if (tail) AMG_RB(false, true);
    else if (prolong) AMG_RB(true, false);
    else AMG_RB(false, false);
#undef AMG_RB
  });
}
int patch_lds_pitch() { return PATCH_EC; }
int patch_max_entries() { return PATCH_MAXTAB; }

// ---- one colour of the multicolour Gauss-Seidel sweep, dictionary-coded -----------------
// Storage rows [p0, p0 + count) are the rows of one colour (host_setup.hpp: ColorPerm),
// rowid[p] the dof each one updates (-1 = padding); codes are indexed by storage row,
// offsets are relative to the dof.  u is updated in place: rows of one colour do not
// reference each other.  Same arithmetic as K-SELL's CSR_GS mode.
template <int WORDS, int UN>
__global__ __launch_bounds__(256) void dict_gs_color_kernel(
    int p0, int count, const uint64_t* __restrict__ codes, const int32_t* __restrict__ rowid,
    const int32_t* __restrict__ doff, const double* __restrict__ dval, int ntab,
    const double* __restrict__ f, double* u, int xcd_map) {
  typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
  __shared__ DictEntry tab[256];
  const int k = xcd_tile(blockIdx.x, gridDim.x, xcd_map) * 256 + (int)threadIdx.x;
  const int p = p0 + k;
  const int row = k < count ? rowid[p] : -1;
  DictStream<WORDS, 1> s;
  s.live[0] = row >= 0;
  s.cw[0][0] = s.cw[0][1] = ~(uint64_t)0;
  s.ty = 0xFFFFu;
  s.fi[0] = s.xi[0] = 0.0;
  if (s.live[0]) {
    const uint64_t* cp = codes + (int64_t)p * WORDS;
    if (WORDS == 2) {
      const u64x2 t = *reinterpret_cast<const u64x2*>(cp);
      s.cw[0][0] = t.x; s.cw[0][1] = t.y;
    } else {
      s.cw[0][0] = cp[0];
    }
    s.fi[0] = f[row];
    s.xi[0] = u[row];
  }
  dict_stage_table<CSR_GS>(tab, doff, dval, ntab);
  __syncthreads();
  double res[1];
  dict_rows<CSR_GS, WORDS, UN, 1>(s, s.live[0] ? row : 0, tab, u, 1.0, 0, res);
  if (s.live[0]) u[row] = res[0];
}
// The whole symmetric pass (colours 0 .. nc-1, then nc-1 .. 0) of a SMALL level in one launch:
// one workgroup, a barrier between colours (__syncthreads also orders the global accesses of
// the workgroup).  On the launch-bound levels of a deep hierarchy a pass is 2 nc launches of
// ~4.6 us each with almost nothing to do; here it is 2 nc barriers.  Same rows, same order of
// the colours, same row arithmetic: same bits.
template <int WORDS, int UN>
__global__ __launch_bounds__(1024) void dict_gs_sweep_kernel(
    int nc, const int32_t* __restrict__ starts, const uint64_t* __restrict__ codes,
    const int32_t* __restrict__ rowid, const int32_t* __restrict__ doff,
    const double* __restrict__ dval, int ntab, const double* __restrict__ f, double* u) {
  typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
  __shared__ DictEntry tab[256];
  if (threadIdx.x < 256) dict_stage_table<CSR_GS>(tab, doff, dval, ntab);
  __syncthreads();
  for (int phase = 0; phase < 2 * nc; ++phase) {
    const int c = phase < nc ? phase : 2 * nc - 1 - phase;
    const int p0 = starts[c], count = starts[c + 1] - p0;
    for (int k = threadIdx.x; k < count; k += 1024) {
      const int p = p0 + k;
      const int row = rowid[p];
      DictStream<WORDS, 1> s;
      s.live[0] = row >= 0;
      s.cw[0][0] = s.cw[0][1] = ~(uint64_t)0;
      s.ty = 0xFFFFu;
      s.fi[0] = s.xi[0] = 0.0;
      if (s.live[0]) {
        const uint64_t* cp = codes + (int64_t)p * WORDS;
        if (WORDS == 2) {
          const u64x2 t = *reinterpret_cast<const u64x2*>(cp);
          s.cw[0][0] = t.x; s.cw[0][1] = t.y;
        } else {
          s.cw[0][0] = cp[0];
        }
        s.fi[0] = f[row];
        s.xi[0] = u[row];
      }
      double res[1];
      dict_rows<CSR_GS, WORDS, UN, 1>(s, s.live[0] ? row : 0, tab, u, 1.0, 0, res);
      if (s.live[0]) u[row] = res[0];
    }
    __syncthreads();  // colour c is complete (and visible) before the next one reads it
  }
}
hipError_t launch_dict_gs_sweep(int nc, const int32_t* starts_dev, int64_t n_storage, int words, int wmax,
                                const uint64_t* codes, const int32_t* rowid, const int32_t* doff,
                                const double* dval, int ntab, const double* f, double* u,
                                hipStream_t st) {
  if (nc <= 0) return hipSuccess;
  if (!starts_dev || n_storage >= ((int64_t)1 << 31) - 512 || ntab > 255 || (words != 1 && words != 2) ||
      wmax > 8 * words)
    return hipErrorInvalidValue;
#define AMG_DICT_GSS(W, U)                                                                   \
  hipLaunchKernelGGL((dict_gs_sweep_kernel<W, U>), dim3(1), dim3(1024), 0, st, nc, starts_dev, \
                     codes, rowid, doff, dval, ntab, f, u)
  if (words == 1) {
    if (wmax <= 3) AMG_DICT_GSS(1, 3);
    else if (wmax <= 5) AMG_DICT_GSS(1, 5);
    else if (wmax <= 7) AMG_DICT_GSS(1, 7);
    else AMG_DICT_GSS(1, 8);
  } else {
    if (wmax <= 9) AMG_DICT_GSS(2, 9);
    else if (wmax <= 12) AMG_DICT_GSS(2, 12);
    else AMG_DICT_GSS(2, 16);
  }
#undef AMG_DICT_GSS
  return hipGetLastError();
}
hipError_t launch_dict_gs_color(int64_t p0, int64_t count, int words, int wmax,
                                const uint64_t* codes, const int32_t* rowid, const int32_t* doff,
                                const double* dval, int ntab, const double* f, double* u,
                                hipStream_t st) {
  if (count <= 0) return hipSuccess;
  if (p0 + count >= ((int64_t)1 << 31) - 512 || ntab > 255 || (words != 1 && words != 2) ||
      wmax > 8 * words)
    return hipErrorInvalidValue;
  const unsigned grid = (unsigned)((count + 255) / 256);
#define AMG_DICT_GS(W, U)                                                                       \
  hipLaunchKernelGGL((dict_gs_color_kernel<W, U>), dim3(grid), dim3(256), 0, st, (int)p0,        \
                     (int)count, codes, rowid, doff, dval, ntab, f, u, g_xcd_map)
  if (words == 1) {
    if (wmax <= 3) AMG_DICT_GS(1, 3);
    else if (wmax <= 5) AMG_DICT_GS(1, 5);
    else if (wmax <= 7) AMG_DICT_GS(1, 7);
    else AMG_DICT_GS(1, 8);
  } else {
    if (wmax <= 9) AMG_DICT_GS(2, 9);
    else if (wmax <= 12) AMG_DICT_GS(2, 12);
    else AMG_DICT_GS(2, 16);
  }
#undef AMG_DICT_GS
  return hipGetLastError();
}

// (W, U) dispatch shared by the fused launchers: F is a generic lambda taking
// std::integral_constant<int, W>, <int, U>, <bool, NT>, <int, R>
template <class F>
static hipError_t dict_dispatch(int words, int wmax, bool nt, bool two, F&& go) {
  using std::integral_constant;
  auto with_nr = [&](auto W, auto U) -> hipError_t {
    if (two) {
      if (nt) go(W, U, integral_constant<bool, true>{}, integral_constant<int, 2>{});
      else go(W, U, integral_constant<bool, false>{}, integral_constant<int, 2>{});
    } else {
      if (nt) go(W, U, integral_constant<bool, true>{}, integral_constant<int, 1>{});
      else go(W, U, integral_constant<bool, false>{}, integral_constant<int, 1>{});
    }
    return hipGetLastError();
  };
  if (words == 1) {
    if (wmax <= 3) return with_nr(integral_constant<int, 1>{}, integral_constant<int, 3>{});
    if (wmax <= 5) return with_nr(integral_constant<int, 1>{}, integral_constant<int, 5>{});
    if (wmax <= 7) return with_nr(integral_constant<int, 1>{}, integral_constant<int, 7>{});
    return with_nr(integral_constant<int, 1>{}, integral_constant<int, 8>{});
  }
  if (wmax <= 9) return with_nr(integral_constant<int, 2>{}, integral_constant<int, 9>{});
  if (wmax <= 12) return with_nr(integral_constant<int, 2>{}, integral_constant<int, 12>{});
  return with_nr(integral_constant<int, 2>{}, integral_constant<int, 16>{});
}
static bool dict_args_ok(int64_t n, int words, int wmax, int ntab) {
  return n < ((int64_t)1 << 31) - 1024 && ntab <= 255 && (words == 1 || words == 2) &&
         wmax <= 8 * words;
}
static bool aligned16(const void* a, const void* b, const void* c) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) |
           reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}
// The contiguous-run-per-XCD tile order pays where the level lives in the caches (levels
// 2+ of the 4096^2 hierarchy: -1..3 us per launch); on the streamed levels beyond the
// Infinity Cache the plain order is faster (level 0: 91 vs 94 us per sweep, fused
// residual 108 vs 114 us), although it fetches x 2x at the L2 -- measured per level.
// tile_rows: rows a workgroup advances by.  A band of >= 16 tiles (the 3-D levels) takes the
// slab-per-plane order whatever the size of the level (xcd_tile).
static int g_xcd_slab = 1;
// paired-load path of wave-uniform 7- / 15-point rows (set_dict_stencil / AMG_HIP_DICT_STENCIL=0: A/B switch)
static int g_dict_stencil = [] {
  const char* e = std::getenv("AMG_HIP_DICT_STENCIL");
  return (e && *e == '0') ? 0 : 1;
}();
static const double* dict_stencil_tab(const DictRef& D, const double* x) {
  return (g_dict_stencil && D.utd && D.uti && D.rtype && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? D.utd : nullptr;
}
static int dict_xcd_map(const DictRef& D, int64_t tile_rows = 512) {
  if (!g_xcd_map) return 0;
  const int64_t tp = D.hb / tile_rows / 8 * 8;
  if (g_xcd_slab && tp >= 16 && tp <= (1 << 20)) return (int)tp;
  return D.nt ? 0 : 1;
}
// two rows per lane need 16-byte aligned f / out / codes (vector lane accesses)
static bool dict_two_rows(int64_t n, const DictRef& D, const void* f, const void* out) {
  return g_dict_rows_per_lane == 2 && n >= 4096 && aligned16(f, out, D.rtype ? nullptr : D.codes) &&
         (!D.rtype || (reinterpret_cast<uintptr_t>(D.rtype) & 1) == 0);
}
hipError_t launch_dict_resid_restrict(int64_t n, const DictRef& D, const double* x,
                                      const double* f, double* r_out, int64_t nH, double* fH,
                                      const double* diagH, double* uH1, double* uH0,
                                      double omega, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (!dict_args_ok(n, D.words, D.wmax, D.ntab) || nH > n || !fH || (uH1 ? !diagH : !uH0))
    return hipErrorInvalidValue;
  const bool two = dict_two_rows(n, D, f, r_out);  // r_out may be nullptr (not stored)
  const int64_t stride = 256 * (two ? 2 : 1) - 2;
  const unsigned tiles = (unsigned)((n + stride - 1) / stride);
  return dict_dispatch(D.words, D.wmax, D.nt != 0, two, [&](auto W, auto U, auto NTF, auto RR) {
    hipLaunchKernelGGL((dict_resid_restrict_kernel<decltype(W)::value, decltype(U)::value,
                                                   decltype(NTF)::value, decltype(RR)::value>),
                       dim3(tiles), dim3(256), 0, st, (int)n, D.codes, D.rtype, D.rwords, D.doff,
                       D.dval, D.ntab, x, f, r_out, (int)nH, fH, diagH, uH1, uH0, omega, dict_xcd_map(D, stride),
                       dict_stencil_tab(D, x), D.uti);
  });
}
hipError_t launch_dict_jacobi_prolong(int64_t n, const DictRef& D, const double* x,
                                      const double* f, double* out, double omega, int64_t n_h,
                                      const double* uh_in, double* uh_out, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (!dict_args_ok(n_h, D.words, D.wmax, D.ntab) || n > n_h || !uh_in || !uh_out ||
      ((reinterpret_cast<uintptr_t>(uh_in) | reinterpret_cast<uintptr_t>(uh_out)) & 15) != 0)
    return hipErrorInvalidValue;
  const bool two = dict_two_rows(n, D, f, out);
  const int64_t stride = 256 * (two ? 2 : 1) - 2;
  const unsigned tiles = (unsigned)((n + stride - 1) / stride);
  return dict_dispatch(D.words, D.wmax, D.nt != 0, two, [&](auto W, auto U, auto NTF, auto RR) {
    hipLaunchKernelGGL((dict_jacobi_prolong_kernel<decltype(W)::value, decltype(U)::value,
                                                   decltype(NTF)::value, decltype(RR)::value>),
                       dim3(tiles), dim3(256), 0, st, (int)n, D.codes, D.rtype, D.rwords, D.doff,
                       D.dval, D.ntab, x, f, out, omega, (int)n_h, uh_in, uh_out, dict_xcd_map(D, stride),
                       dict_stencil_tab(D, x), D.uti);
  });
}
// the two-sweep forms for small levels; see dict_pair_down_kernel / dict_pair_up_kernel
bool dict_pair_ok(int64_t n, const DictRef& D, int hb, const void* a, const void* b, const void* c) {
  return hb >= 0 && hb <= PAIR_HB - 1 && n >= 256 && n <= 400000 && !D.nt &&
         dict_args_ok(n, D.words, D.wmax, D.ntab) && g_dict_rows_per_lane == 2 &&
         aligned16(a, b, c) && aligned16(D.rtype ? nullptr : D.codes, nullptr, nullptr) &&
         (!D.rtype || (reinterpret_cast<uintptr_t>(D.rtype) & 1) == 0);
}
template <class F>
static hipError_t dict_dispatch_wu(int words, int wmax, F&& go) {
  using std::integral_constant;
  if (words == 1) {
    if (wmax <= 3) go(integral_constant<int, 1>{}, integral_constant<int, 3>{});
    else if (wmax <= 5) go(integral_constant<int, 1>{}, integral_constant<int, 5>{});
    else if (wmax <= 7) go(integral_constant<int, 1>{}, integral_constant<int, 7>{});
    else go(integral_constant<int, 1>{}, integral_constant<int, 8>{});
  } else {
    if (wmax <= 9) go(integral_constant<int, 2>{}, integral_constant<int, 9>{});
    else if (wmax <= 12) go(integral_constant<int, 2>{}, integral_constant<int, 12>{});
    else go(integral_constant<int, 2>{}, integral_constant<int, 16>{});
  }
  return hipGetLastError();
}
hipError_t launch_dict_pair_down(int64_t n, const DictRef& D, int hb, const double* a,
                                 const double* f, double* u_out, double* r_out, int64_t nH,
                                 double* fH, const double* diagH, double* uH1, double omega,
                                 hipStream_t st) {
  if (!dict_pair_ok(n, D, hb, a, f, u_out) || !fH || !diagH || !uH1 || nH > n ||
      (r_out && (reinterpret_cast<uintptr_t>(r_out) & 15)))
    return hipErrorInvalidValue;
  const int hbw = (hb + 1) & ~1;
  const unsigned tiles = (unsigned)((n + 509) / 510);
  return dict_dispatch_wu(D.words, D.wmax, [&](auto W, auto U) {
    hipLaunchKernelGGL((dict_pair_down_kernel<decltype(W)::value, decltype(U)::value>), dim3(tiles),
                       dim3(256), 0, st, (int)n, D.codes, D.rtype, D.rwords, D.doff, D.dval, D.ntab,
                       a, f, u_out, r_out, (int)nH, fH, diagH, uH1, omega, hbw, dict_xcd_map(D));
  });
}
hipError_t launch_dict_pair_up(int64_t n, const DictRef& D, int hb, const double* a,
                               const double* f, double* u_out, double omega, int64_t n_h,
                               const double* uh_in, double* uh_out, hipStream_t st) {
  if (!dict_pair_ok(n, D, hb, a, f, u_out) || n > n_h || !uh_in || !uh_out ||
      ((reinterpret_cast<uintptr_t>(uh_in) | reinterpret_cast<uintptr_t>(uh_out)) & 15) != 0)
    return hipErrorInvalidValue;
  const int hbw = (hb + 1) & ~1;
  const unsigned tiles = (unsigned)((n + 509) / 510);
  return dict_dispatch_wu(D.words, D.wmax, [&](auto W, auto U) {
    hipLaunchKernelGGL((dict_pair_up_kernel<decltype(W)::value, decltype(U)::value>), dim3(tiles),
                       dim3(256), 0, st, (int)n, D.codes, D.rtype, D.rwords, D.doff, D.dval, D.ntab,
                       a, f, u_out, omega, hbw, (int)n_h, uh_in, uh_out, dict_xcd_map(D));
  });
}
// ---- K-March: two true-Jacobi sweeps of a 3-D 7-point level in ONE pass ---------------------
// On the finest level of a 3-D hierarchy a sweep streams x, f and the output again (25 B per row),
// and the +-plane neighbours defeat the 2-D patch form (three rings around planes x lines x columns
// leave a quarter of a patch as output).  Here a workgroup owns a tile of TL lines x TC columns and
// MARCHES through the planes: x of plane q arrives (tile + 2 rings in the plane), the first sweep is
// formed on plane q-1 (tile + 1 ring) from the x planes q-2, q-1, q, the second sweep on plane q-2
// (tile) from the first-sweep planes q-3, q-2, q-1 -- two rings of three planes in LDS, two barriers
// per plane, only the two in-plane directions pay a halo.  Row structure from the row TYPE: the
// values of every type laid out by pattern position {-M, -m, -1, (0), +1, +m, +M} by the host
// (positions a boundary row does not have: +0.0, its diagonal apart); waves of interior cells take
// the interior type's values from kernel arguments.  Same products in the same (ascending column)
// order, the same IEEE division and relaxation expression as dict_rows<CSR_JACOBI>, adding (+0.0) x
// where dict_rows adds it for absent slots -> same bits as two launches of the sweep.
constexpr int MARCH_TL = 16, MARCH_TC = 64, MARCH_NT = 512;  // (8 lines x 256 threads: 0.8 % slower)
constexpr int MARCH_XL = MARCH_TL + 4, MARCH_XC = MARCH_TC + 4;  // x tile: 20 x 68
constexpr int MARCH_SL = MARCH_TL + 2, MARCH_SC = MARCH_TC + 2;  // first-sweep tile: 18 x 66
constexpr int MARCH_XN = MARCH_XL * MARCH_XC, MARCH_SN = MARCH_SL * MARCH_SC;
constexpr int MARCH_XS = (MARCH_XN + MARCH_NT - 1) / MARCH_NT;   // x cells per thread (3)
constexpr int MARCH_SS = (MARCH_SN + MARCH_NT - 1) / MARCH_NT;   // first-sweep cells per thread (3)
constexpr int MARCH_OS = MARCH_TL * MARCH_TC / MARCH_NT;         // output cells per thread (2)
constexpr int MARCH_TYPES = 64;
typedef MarchRef MarchArgs;
// one Jacobi update of a cell from its six neighbours nb[] (-M, -m, -1, +1, +m, +M) and itself
__device__ __forceinline__ double march_update(const double (&w)[6], double diag, const double (&nb)[6],
                                               double xi, double fi, double omega) {
  double acc = 0.0;
#pragma unroll
  for (int u = 0; u < 6; ++u) acc += w[u] * nb[u];
  const bool nod = diag == 0.0;
  const double q = (fi - acc) / (nod ? 1.0 : diag);  // smoother.hpp:136
  return nod ? xi : xi + omega * (q - xi);
}
__global__ __launch_bounds__(MARCH_NT) void march_kernel(MarchArgs A) {
  __shared__ double X[3][MARCH_XN];
  __shared__ double S[3][MARCH_SN];
  __shared__ double wt[MARCH_TYPES * 8];  // per type: w[6] (pattern order without the diagonal), diagonal, -
  const int tid = (int)threadIdx.x;
  const int tiles_c = A.m / MARCH_TC, tiles_l = A.lines / MARCH_TL;
  const int tile = (int)blockIdx.x % (tiles_c * tiles_l), chunk = (int)blockIdx.x / (tiles_c * tiles_l);
  const int l0 = (tile / tiles_c) * MARCH_TL, c0 = (tile % tiles_c) * MARCH_TC;
  const int pa = chunk * A.chunk_planes, pb = min(pa + A.chunk_planes, A.planes);
  for (int q = tid; q < A.ntypes * 8; q += MARCH_NT) wt[q] = A.wtab[q];
  // cells of this thread
  int xoff[MARCH_XS];   // global row offset inside a plane (line * m + col), or -1 outside the domain
#pragma unroll
  for (int k = 0; k < MARCH_XS; ++k) {
    const int q = k * MARCH_NT + tid;
    const int line = l0 - 2 + q / MARCH_XC, col = c0 - 2 + q % MARCH_XC;
    xoff[k] = (q < MARCH_XN && line >= 0 && line < A.lines && col >= 0 && col < A.m) ? line * A.m + col : -1;
  }
  // first-sweep cells of this thread: its two OUTPUT cells (so that their f and row type are still in
  // registers when the second sweep reaches the plane, one step later) and one cell of the ring
  // around the tile (164 cells: the first 164 threads).  soff: row offset in the plane or -1, si / sx:
  // index in the first-sweep tile / in the x tile.
  static_assert(MARCH_SS == MARCH_OS + 1 && 2 * MARCH_SC + 2 * MARCH_TL <= MARCH_NT, "first-sweep cell map");
  int soff[MARCH_SS], si[MARCH_SS], sx[MARCH_SS];
  int ooff[MARCH_OS];
#pragma unroll
  for (int k = 0; k < MARCH_OS; ++k) {
    const int q = k * MARCH_NT + tid;
    const int ol = q / MARCH_TC, oc = q % MARCH_TC;
    ooff[k] = (l0 + ol) * A.m + c0 + oc;   // always inside the domain
    soff[k] = ooff[k];
    si[k] = (ol + 1) * MARCH_SC + oc + 1;
    sx[k] = (ol + 2) * MARCH_XC + oc + 2;
  }
  {
    const int r = tid;
    int sl, sc;
    if (r < MARCH_SC) { sl = 0; sc = r; }
    else if (r < 2 * MARCH_SC) { sl = MARCH_SL - 1; sc = r - MARCH_SC; }
    else if (r < 2 * MARCH_SC + MARCH_TL) { sl = r - 2 * MARCH_SC + 1; sc = 0; }
    else { sl = r - 2 * MARCH_SC - MARCH_TL + 1; sc = MARCH_SC - 1; }
    const bool ring = r < 2 * MARCH_SC + 2 * MARCH_TL;
    if (!ring) { sl = 1; sc = 1; }
    const int line = l0 - 1 + sl, col = c0 - 1 + sc;
    soff[MARCH_OS] = (ring && line >= 0 && line < A.lines && col >= 0 && col < A.m) ? line * A.m + col : -1;
    si[MARCH_OS] = ring ? sl * MARCH_SC + sc : -1;   // -1: no cell (nothing is stored)
    sx[MARCH_OS] = (sl + 1) * MARCH_XC + sc + 1;
  }
  const int64_t M = (int64_t)A.m * A.lines;
  double xr[MARCH_XS];  // x of the plane that arrives next
  auto fetch_x = [&](int plane) {
#pragma unroll
    for (int k = 0; k < MARCH_XS; ++k)
      xr[k] = (plane >= 0 && plane < A.planes && xoff[k] >= 0) ? A.x[(int64_t)plane * M + xoff[k]] : 0.0;
  };
  double wi[6];
#pragma unroll
  for (int u = 0; u < 6; ++u) wi[u] = A.wint[u];
  const double di = A.dint;
  const uint32_t ti = (uint32_t)A.tint;
  // operands (f, row type) of the first sweep on plane q - 1 and of the second on plane q - 2:
  // requested one plane step ahead, like x (a step is two barriers and a few dozen LDS reads; a
  // global round trip inside it would be most of its time)
  double nf1[MARCH_SS];
  uint32_t nt1[MARCH_SS];
  auto fetch_ops = [&](int q) {
    const int p1 = q - 1;
    const bool live1 = p1 >= pa - 1 && p1 <= pb && p1 >= 0 && p1 < A.planes;
#pragma unroll
    for (int k = 0; k < MARCH_SS; ++k) {
      const bool live = live1 && soff[k] >= 0;
      const int64_t r = live ? (int64_t)p1 * M + soff[k] : 0;
      nf1[k] = live ? A.f[r] : 0.0;
      nt1[k] = live ? (uint32_t)A.rtype[r] : 255u;
    }
  };
  double f2[MARCH_OS];   // operands of the second sweep: what the first sweep used one step ago
  uint32_t t2[MARCH_OS];
#pragma unroll
  for (int k = 0; k < MARCH_OS; ++k) {
    f2[k] = 0.0;
    t2[k] = 255u;
  }
  fetch_x(pa - 2);
  fetch_ops(pa - 2);
  __syncthreads();
  for (int q = pa - 2; q <= pb + 1; ++q) {
    // x of plane q into its ring slot; the next plane is requested
    {
      double* Xq = X[(q + 3) % 3];
#pragma unroll
      for (int k = 0; k < MARCH_XS; ++k)
        if (k * MARCH_NT + tid < MARCH_XN) Xq[k * MARCH_NT + tid] = xr[k];
    }
    const int p1 = q - 1, p2 = q - 2;
    const bool do1 = p1 >= pa - 1 && p1 <= pb;
    const bool do2 = p2 >= pa && p2 < pb;
    double f1[MARCH_SS];
    uint32_t t1[MARCH_SS];
#pragma unroll
    for (int k = 0; k < MARCH_SS; ++k) {
      f1[k] = nf1[k];
      t1[k] = nt1[k];
    }
    fetch_x(q + 1);
    fetch_ops(q + 1);
    lds_barrier();
    if (do1) {  // first sweep on plane q - 1 from the x planes q - 2, q - 1, q
      const double* Xm = X[(q + 1) % 3];  // plane q - 2
      const double* Xc = X[(q + 2) % 3];  // plane q - 1
      const double* Xp = X[(q + 3) % 3];  // plane q
      double* Sq = S[(p1 + 3) % 3];
#pragma unroll
      for (int k = 0; k < MARCH_SS; ++k) {
        const int i = sx[k];
        const double nb[6] = {Xm[i], Xc[i - MARCH_XC], Xc[i - 1], Xc[i + 1], Xc[i + MARCH_XC], Xp[i]};
        const double xi = Xc[i];
        const bool live = t1[k] != 255u;
        double r;
        if (__builtin_amdgcn_ballot_w64(live && t1[k] != ti) == 0) {  // interior (or dead) cells only
          r = march_update(wi, di, nb, xi, f1[k], A.omega);
        } else {
          const uint32_t tt = t1[k] < (uint32_t)MARCH_TYPES ? t1[k] : 0u;
          double w[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) w[u] = wt[tt * 8 + u];
          r = march_update(w, wt[tt * 8 + 6], nb, xi, f1[k], A.omega);
        }
        if (si[k] >= 0) Sq[si[k]] = live ? r : 0.0;
      }
    }
    lds_barrier();
    if (do2) {  // second sweep on plane q - 2 from the first-sweep planes q - 3, q - 2, q - 1
      const double* Sm = S[(p2 + 2) % 3];
      const double* Sc = S[(p2 + 3) % 3];
      const double* Sp = S[(p2 + 4) % 3];
#pragma unroll
      for (int k = 0; k < MARCH_OS; ++k) {
        const int i = si[k];
        const double nb[6] = {Sm[i], Sc[i - MARCH_SC], Sc[i - 1], Sc[i + 1], Sc[i + MARCH_SC], Sp[i]};
        const double xi = Sc[i];
        double r;
        if (__builtin_amdgcn_ballot_w64(t2[k] != ti) == 0) {
          r = march_update(wi, di, nb, xi, f2[k], A.omega);
        } else {
          const uint32_t tt = t2[k] < (uint32_t)MARCH_TYPES ? t2[k] : 0u;
          double w[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) w[u] = wt[tt * 8 + u];
          r = march_update(w, wt[tt * 8 + 6], nb, xi, f2[k], A.omega);
        }
        A.out[(int64_t)p2 * M + ooff[k]] = r;
      }
    }
#pragma unroll
    for (int k = 0; k < MARCH_OS; ++k) {  // plane q - 1 is the second sweep's plane of the next step
      f2[k] = f1[k];
      t2[k] = t1[k];
    }
  }
}
bool march_ok(const MarchRef& A) {
  return A.m >= MARCH_TC && A.m % MARCH_TC == 0 && A.lines >= MARCH_TL && A.lines % MARCH_TL == 0 && A.planes >= 3 &&
         (int64_t)A.m * A.lines * A.planes < ((int64_t)1 << 31) - 1024 && A.ntypes >= 1 && A.ntypes <= MARCH_TYPES &&
         A.tint >= 0 && A.tint < A.ntypes && A.chunk_planes >= 1 && A.wtab && A.rtype && A.x && A.f && A.out &&
         A.out != A.x;
}
hipError_t launch_march(MarchRef A, hipStream_t st) {
  // one wave of workgroups (2 per CU x 256 CUs): as many plane chunks as that takes
  const int64_t tiles = (int64_t)(A.m / MARCH_TC) * (A.lines / MARCH_TL);
  int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(A.planes / 8, (512 + tiles - 1) / tiles));
  A.chunk_planes = (A.planes + chunks - 1) / chunks;
  chunks = (A.planes + A.chunk_planes - 1) / A.chunk_planes;
  if (!march_ok(A)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(march_kernel, dim3((unsigned)(tiles * chunks)), dim3(MARCH_NT), 0, st, A);
  return hipGetLastError();
}

// ---- K-Strip: the multicolour smoother's whole leg of a NARROW level in one launch ----------
// Below the K-Patch levels (pitch < 128) the symmetric pass of a 4-colour level is 7 colour
// launches of ~5 us each with almost nothing to do, plus the residual + restriction or the
// prolongation: 16 launches per level and cycle.  Here a workgroup takes a strip of T consecutive
// rows plus a halo of (stages [+ 1]) x half-bandwidth rows on either side into LDS (u, f, row
// types, colours), runs the colour stages LDS -> LDS in place (rows of one colour do not read each
// other; a barrier between stages), and finishes with the residual + restriction (TAIL) or starts
// from u + P u_H (PROLONG): the 1-D counterpart of patch_rb_kernel, for any colouring (one colour
// byte per row).  The halo rows are recomputed by the neighbouring strips with the same
// expressions; row arithmetic = dict_rows<CSR_GS> / <CSR_RESID> on the level's own dictionary
// (ascending columns; the colour kernels walk the same entries of the symmetric matrix), transfers
// = patch_rb_kernel's: same bits as one launch per colour.
constexpr int STRIP_NT = 512;
constexpr int STRIP_WMAX = 4608;  // rows of a window (u and f: 2 x 36 KB of LDS)
typedef StripRef StripArgs;
int strip_window_max() { return STRIP_WMAX; }
template <int WORDS, int UN, bool PROLONG, bool TAIL>
__global__ __launch_bounds__(STRIP_NT) void mc_strip_kernel(StripArgs A) {
  __shared__ __attribute__((aligned(16))) double uw[STRIP_WMAX];
  __shared__ __attribute__((aligned(16))) double fw[STRIP_WMAX];
  __shared__ __attribute__((aligned(16))) uint8_t tyw[STRIP_WMAX];
  __shared__ __attribute__((aligned(16))) uint8_t cw8[STRIP_WMAX];
  __shared__ DictEntry tabG[256];
  __shared__ DictEntry tabR[TAIL ? 256 : 1];
  __shared__ uint64_t wtab[256 * WORDS];
  const int tid = (int)threadIdx.x;
  const int t0 = (int)blockIdx.x * A.T, w0 = t0 - A.H;
  const int W = A.T + 2 * A.H + 2;
  const int n = A.n;
  if (tid < 256) {
    dict_stage_table<CSR_GS>(tabG, A.doff, A.dval, A.ntab);
    if (TAIL) dict_stage_table<CSR_RESID>(tabR, A.doff, A.dval, A.ntab);
    dict_stage_words<WORDS>(wtab, A.rwords);
  }
  for (int lw = tid; lw < W; lw += STRIP_NT) {
    const int r = w0 + lw;
    const bool live = r >= 0 && r < n;
    const int row = live ? r : 0;
    double x0 = live ? A.x[row] : 0.0;
    if (PROLONG) {  // linear_prolong_add_kernel, same guards and order (patch_load)
      const int nH = A.nH;
      const int j = row >> 1;
      const bool odd = (row & 1) != 0;
      const bool a_ok = live && !odd && j >= 1 && j - 1 < nH;
      const bool b_ok = live && j < nH;
      const double a = A.uH[a_ok ? j - 1 : 0];
      const double b = A.uH[b_ok ? j : 0];
      double t = 0.0;
      t = a_ok ? t + 0.5 * a : t;
      t = b_ok ? t + (odd ? 1.0 : 0.5) * b : t;
      x0 = live ? x0 + t : x0;
    }
    uw[lw] = x0;
    fw[lw] = live ? A.f[row] : 0.0;
    tyw[lw] = live ? A.rtype[row] : (uint8_t)255;
    cw8[lw] = live ? A.color[row] : (uint8_t)255;
  }
  __syncthreads();
  constexpr int E = TAIL ? 1 : 0;
  for (int sg = 0; sg < A.nst; ++sg) {
    const uint32_t c = (uint32_t)(A.stages >> (4 * sg)) & 15u;
    const int ext = (A.nst - 1 - sg + E) * A.hbw;
    const int lo = A.H - ext, hi = A.H + A.T + 2 + ext;
    for (int lw = tid * 2; lw < W; lw += 2 * STRIP_NT) {
      DictStream<WORDS, 2> s;
      bool did[2];
      uint32_t ty = 0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int l = lw + r;
        did[r] = l >= lo && l < hi && (uint32_t)cw8[l] == c;
        s.live[r] = did[r];
        s.fi[r] = fw[l];
        s.xi[r] = uw[l];
        ty |= (did[r] ? (uint32_t)tyw[l] : 255u) << (8 * r);
      }
      s.ty = ty;
      dict_expand<WORDS, 2>(s, wtab);
      double res[2];
      dict_rows<CSR_GS, WORDS, UN, 2>(s, lw, tabG, uw, 1.0, 0, res);
      if (did[0]) uw[lw] = res[0];
      if (did[1]) uw[lw + 1] = res[1];
    }
    __syncthreads();
  }
  if (TAIL) {  // r = f - A u on the rows the restriction reads, left in the f window
    for (int lw = tid * 2; lw < W; lw += 2 * STRIP_NT) {
      DictStream<WORDS, 2> s;
      bool inr[2];
      uint32_t ty = 0;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int l = lw + r;
        inr[r] = l >= A.H && l < A.H + A.T + 2 && tyw[l] != (uint8_t)255;
        s.live[r] = inr[r];
        s.fi[r] = fw[l];
        s.xi[r] = 0.0;
        ty |= (inr[r] ? (uint32_t)tyw[l] : 255u) << (8 * r);
      }
      s.ty = ty;
      dict_expand<WORDS, 2>(s, wtab);
      double res[2];
      dict_rows<CSR_RESID, WORDS, UN, 2>(s, lw, tabR, uw, 1.0, 0, res);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int l = lw + r;
        if (l >= A.H && l < A.H + A.T + 2) fw[l] = inr[r] ? res[r] : 0.0;
      }
    }
    __syncthreads();
  }
  for (int q = tid; q < A.T; q += STRIP_NT) {
    const int r = t0 + q;
    if (r < n) {
      A.u_out[r] = uw[A.H + q];
      if (TAIL && A.r_out) A.r_out[r] = fw[A.H + q];
    }
  }
  if (TAIL) {
    for (int qj = tid; 2 * qj < A.T; qj += STRIP_NT) {
      const int64_t i = (int64_t)t0 + 2 * qj;  // fine row 2c
      const int64_t cj = i >> 1;
      if (i >= (int64_t)n || cj >= A.nH) continue;
      const double* rs = fw + A.H + 2 * qj;
      double sum = 0.0;  // dict_restrict_tail / linear_restrict_kernel, same guards and order
      if (i < n) sum += 0.5 * rs[0];
      if (i + 1 < n) sum += 1.0 * rs[1];
      if (i + 2 < n) sum += 0.5 * rs[2];
      A.fH[cj] = sum;
      if (A.uH_zero) A.uH_zero[cj] = 0.0;  // multigrid.hpp:278
    }
  }
}
bool strip_ok(const StripRef& A, const DictRef& D) {
  return D.rtype && D.rwords && D.doff && D.dval && dict_args_ok(A.n, D.words, D.wmax, D.ntab) && A.n >= 2 &&
         A.T >= 2 && (A.T & 1) == 0 && A.hbw >= D.hb && A.hbw >= 1 && A.nst >= 1 && A.nst <= 16 &&
         A.T + 2 * A.H + 2 <= STRIP_WMAX && A.color && A.x && A.f && A.u_out && A.u_out != A.x;
}
hipError_t launch_mc_strip(bool prolong, bool tail, StripRef A, const DictRef& D, hipStream_t st) {
  A.H = (A.nst + (tail ? 1 : 0)) * A.hbw;
  A.rtype = D.rtype; A.rwords = D.rwords; A.doff = D.doff; A.dval = D.dval; A.ntab = D.ntab;
  if (!strip_ok(A, D) || (prolong && (tail || !A.uH)) || (tail && !A.fH) || A.nH < 1) return hipErrorInvalidValue;
  const unsigned tiles = (unsigned)((A.n + A.T - 1) / A.T);
  return dict_dispatch_wu(D.words, D.wmax, [&](auto W, auto U) {
#define AMG_STRIP(PRO, TL) \
  hipLaunchKernelGGL((mc_strip_kernel<decltype(W)::value, decltype(U)::value, PRO, TL>), dim3(tiles), dim3(STRIP_NT), 0, st, A)
    if (tail) AMG_STRIP(false, true);
    else if (prolong) AMG_STRIP(true, false);
    else AMG_STRIP(false, false);
#undef AMG_STRIP
  });
}

void set_xcd_mapping(int on) {
  g_xcd_map = on ? 1 : 0;
  g_xcd_slab = on == 1 ? 1 : 0;  // 2: contiguous runs only (the round-2 order), an A/B switch
}
void set_dict_rows_per_lane(int r) { g_dict_rows_per_lane = r == 1 ? 1 : 2; }
void set_dict_stencil(int on) { g_dict_stencil = on ? 1 : 0; }
template <int MODE>
static hipError_t launch_dict_mode(int64_t n, const DictRef& D, const double* x, const double* f,
                                   double* out, double omega, int64_t dshift, hipStream_t st) {
  const bool two = dict_two_rows(n, D, f, out);
  const unsigned tiles = (unsigned)((n + 256 * (two ? 2 : 1) - 1) / (256 * (two ? 2 : 1)));
  return dict_dispatch(D.words, D.wmax, D.nt != 0, two, [&](auto W, auto U, auto NTF, auto RR) {
    hipLaunchKernelGGL((dict_kernel<MODE, decltype(W)::value, decltype(U)::value,
                                    decltype(NTF)::value, decltype(RR)::value>),
                       dim3(tiles), dim3(256), 0, st, (int)n, D.codes, D.rtype, D.rwords, D.doff,
                       D.dval, D.ntab, x, f, out, omega, (int)dshift, dict_xcd_map(D, 256 * (two ? 2 : 1)),
                       dshift == 0 ? dict_stencil_tab(D, x) : nullptr, D.uti);
  });
}
// the name rocprofv3 prints for the launch launch_dict(mode, ...) makes
void dict_kernel_name(int mode, int64_t n, const DictRef& D, const void* f, const void* out,
                      char* buf, size_t cap) {
  const int u = D.words == 1 ? (D.wmax <= 3 ? 3 : D.wmax <= 5 ? 5 : D.wmax <= 7 ? 7 : 8)
                             : (D.wmax <= 9 ? 9 : D.wmax <= 12 ? 12 : 16);
  std::snprintf(buf, cap, "dict_kernel<%d, %d, %d, %s, %d>", mode, D.words, u,
                D.nt ? "true" : "false", dict_two_rows(n, D, f, out) ? 2 : 1);
}
hipError_t launch_dict(int mode, int64_t n, const DictRef& D, const double* x, const double* f,
                       double* out, double omega, int64_t diag_shift, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (!dict_args_ok(n, D.words, D.wmax, D.ntab) || diag_shift >= ((int64_t)1 << 28))
    return hipErrorInvalidValue;
  switch (mode) {
    case CSR_RESID: return launch_dict_mode<CSR_RESID>(n, D, x, f, out, omega, diag_shift, st);
    case CSR_JACOBI: return launch_dict_mode<CSR_JACOBI>(n, D, x, f, out, omega, diag_shift, st);
    case CSR_SPMV: return launch_dict_mode<CSR_SPMV>(n, D, x, f, out, omega, diag_shift, st);
    case CSR_RSSQ: return launch_dict_mode<CSR_RSSQ>(n, D, x, f, out, omega, diag_shift, st);
  }
  return hipErrorInvalidValue;
}

// ------------------------------------------------- K-Restrict / K-ProlongAdd ---
// f_H[j] = ((0 + 0.5 r[2j]) + 1.0 r[2j+1]) + 0.5 r[2j+2]   (Eigen column-major
// scatter order of R*v, interpolator.hpp:64-68 with R = P^T, :132-134).
// uH (optional) is zero-filled in the same pass (multigrid.hpp:278).
__global__ __launch_bounds__(256) void linear_restrict_kernel(
    int64_t n_h, int64_t n_H, const double* __restrict__ r, double* __restrict__ fH,
    double* __restrict__ uH) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_H) return;
  if (uH) uH[j] = 0.0;
  const int64_t i = 2 * j;
  double s = 0.0;  // r is dead after this kernel: stream it
  if (i < n_h) s += 0.5 * __builtin_nontemporal_load(r + i);
  if (i + 1 < n_h) s += 1.0 * __builtin_nontemporal_load(r + i + 1);
  if (i + 2 < n_h) s += 0.5 * r[i + 2];
  fH[j] = s;
}
// u_h[i] = u_h[i] + t[i], t = P u_H: t[2j+1] = 0 + 1.0 u_H[j];
// t[2j] = (0 + 0.5 u_H[j-1]) + 0.5 u_H[j]; rows past 2 n_H get t = 0
// (interpolator.hpp:52-56, multigrid.hpp:294-296).
__global__ __launch_bounds__(256) void linear_prolong_add_kernel(
    int64_t n_h, int64_t n_H, const double* __restrict__ uH, double* __restrict__ uh) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_h) return;
  double t = 0.0;
  const int64_t j = i >> 1;
  if (i & 1) {
    if (j < n_H) t += 1.0 * uH[j];
  } else {
    if (j >= 1 && j - 1 < n_H) t += 0.5 * uH[j - 1];  // column j-1, row 2(j-1)+2
    if (j < n_H) t += 0.5 * uH[j];                    // column j,   row 2j
  }
  uh[i] = uh[i] + t;
}
hipError_t launch_linear_restrict(int64_t n_h, int64_t n_H, const double* r, double* fH,
                                  double* uH_zero, hipStream_t st) {
  if (n_H <= 0) return hipSuccess;
  hipLaunchKernelGGL(linear_restrict_kernel, dim3((unsigned)((n_H + 255) / 256)),
                     dim3(256), 0, st, n_h, n_H, r, fH, uH_zero);
  return hipGetLastError();
}
// Two fine rows (2j, 2j+1) per lane: 16-byte load/store of u, same arithmetic.
__global__ __launch_bounds__(256) void linear_prolong_add2_kernel(
    int64_t n_h, int64_t n_H, const double* __restrict__ uH, const double* uh_in, double* uh) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = 2 * j;
  if (i >= n_h) return;
  double t0 = 0.0, t1 = 0.0;
  const double b = (j < n_H) ? uH[j] : 0.0;
  if (j >= 1 && j - 1 < n_H) t0 += 0.5 * uH[j - 1];
  if (j < n_H) {
    t0 += 0.5 * b;
    t1 += 1.0 * b;
  }
  if (i + 1 < n_h) {
    double2 u = *reinterpret_cast<const double2*>(uh_in + i);
    u.x = u.x + t0;
    u.y = u.y + t1;
    *reinterpret_cast<double2*>(uh + i) = u;
  } else {
    uh[i] = uh_in[i] + t0;
  }
}
// uh_out = uh_in + P uH (16-byte aligned vectors); uh_out == uh_in is the in-place form
hipError_t launch_linear_prolong_to(int64_t n_h, int64_t n_H, const double* uH,
                                    const double* uh_in, double* uh_out, hipStream_t st) {
  if (n_h <= 0) return hipSuccess;
  if (((reinterpret_cast<uintptr_t>(uh_in) | reinterpret_cast<uintptr_t>(uh_out)) & 15) != 0)
    return hipErrorInvalidValue;
  const int64_t nt = (n_h + 1) / 2;
  hipLaunchKernelGGL(linear_prolong_add2_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0,
                     st, n_h, n_H, uH, uh_in, uh_out);
  return hipGetLastError();
}
hipError_t launch_linear_prolong_add(int64_t n_h, int64_t n_H, const double* uH,
                                     double* uh, hipStream_t st) {
  if (n_h <= 0) return hipSuccess;
  if ((reinterpret_cast<uintptr_t>(uh) & 15) == 0) {
    const int64_t nt = (n_h + 1) / 2;
    hipLaunchKernelGGL(linear_prolong_add2_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256),
                       0, st, n_h, n_H, uH, uh, uh);
  } else {
    hipLaunchKernelGGL(linear_prolong_add_kernel, dim3((unsigned)((n_h + 255) / 256)),
                       dim3(256), 0, st, n_h, n_H, uH, uh);
  }
  return hipGetLastError();
}

// First Jacobi sweep from a zero guess (coarse levels on the way down,
// multigrid.hpp:278 then :268): with u == 0 every a_ij*u_j is +-0 and the row sum
// is +0.0, so the sweep needs only f and the diagonal -- same operations, same
// bits as the full kernel on a zero vector, without streaming the matrix.
__global__ __launch_bounds__(256) void jacobi_from_zero_kernel(
    int64_t n, const double* __restrict__ diag, const double* __restrict__ f,
    double* __restrict__ out, double omega) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double xi = 0.0, acc = 0.0;
  const double d = diag[i], fi = f[i];
  out[i] = (d == 0.0) ? xi : xi + omega * ((fi - acc) / d - xi);
}
hipError_t launch_jacobi_from_zero(int64_t n, const double* diag, const double* f, double* out,
                                   double omega, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(jacobi_from_zero_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     st, n, diag, f, out, omega);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void add_inplace_kernel(int64_t n,
                                                          const double* __restrict__ x,
                                                          double* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = y[i] + x[i];
}
hipError_t launch_add_inplace(int64_t n, const double* x, double* y, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     st, n, x, y);
  return hipGetLastError();
}

// ---------------------------------------------------------------- K-SumSq ----
// Deterministic two-stage reduction: grid-stride per-thread sums, wave shuffle
// tree, LDS across the 4 waves, one partial per block; the last stage is one
// block over <= 1024 partials.  square=1: sum of x^2, square=0: plain sum.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__global__ __launch_bounds__(256) void sum_kernel(int64_t n, const double* __restrict__ x,
                                                  double* __restrict__ out, int square) {
  __shared__ double part[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double v = x[i];
    s += square ? v * v : v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ((part[0] + part[1]) + part[2]) + part[3];
}
hipError_t launch_sum(int64_t n, const double* x, double* out, double* scratch, int square,
                      hipStream_t st) {
  int64_t g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(sum_kernel, dim3((unsigned)g), dim3(256), 0, st, n, x, scratch, square);
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, g, scratch, out, 0);
  return hipGetLastError();
}

// ------------------------------------------------------------------ K-PCG -----
// Vector kernels of the preconditioned conjugate-gradient driver (the V-cycle as M^-1,
// README.md:127 of the reference): deterministic two-stage dot product and the two
// updates; the scalars alpha / beta are formed on the device from the dot products in
// sc[], so an iteration needs no host round trip.
__global__ __launch_bounds__(256) void dot_kernel(int64_t n, const double* __restrict__ x,
                                                  const double* __restrict__ y,
                                                  double* __restrict__ out) {
  __shared__ double part[4];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    s += x[i] * y[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = ((part[0] + part[1]) + part[2]) + part[3];
}
hipError_t launch_dot(int64_t n, const double* x, const double* y, double* out, double* scratch,
                      hipStream_t st) {
  int64_t g = (n + 255) / 256;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(dot_kernel, dim3((unsigned)g), dim3(256), 0, st, n, x, y, scratch);
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, g, scratch, out, 0);
  return hipGetLastError();
}
// alpha = num / den;  x = x + alpha p;  r = r - alpha q
__global__ __launch_bounds__(256) void pcg_update_xr_kernel(int64_t n, const double* __restrict__ num,
                                                            const double* __restrict__ den, double* x,
                                                            double* r, const double* __restrict__ p,
                                                            const double* __restrict__ q) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double alpha = *num / *den;
  x[i] = x[i] + alpha * p[i];
  r[i] = r[i] - alpha * q[i];
}
// beta = num / den;  p = z + beta p
__global__ __launch_bounds__(256) void pcg_update_p_kernel(int64_t n, const double* __restrict__ num,
                                                           const double* __restrict__ den, double* p,
                                                           const double* __restrict__ z) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double beta = *num / *den;
  p[i] = z[i] + beta * p[i];
}
hipError_t launch_pcg_update_xr(int64_t n, const double* num, const double* den, double* x, double* r,
                                const double* p, const double* q, hipStream_t st) {
  hipLaunchKernelGGL(pcg_update_xr_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, num,
                     den, x, r, p, q);
  return hipGetLastError();
}
hipError_t launch_pcg_update_p(int64_t n, const double* num, const double* den, double* p,
                               const double* z, hipStream_t st) {
  hipLaunchKernelGGL(pcg_update_p_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, num,
                     den, p, z);
  return hipGetLastError();
}

// ---------------------------------------------------------------- K-GS-lex ---
// Exact lexicographic Gauss-Seidel (smoother.hpp:101-174) under a dependency
// schedule built on the host (host_setup.cpp: build_lex_schedule).  ONE
// workgroup walks the level window by window (BLOCK slots per window).  Inside
// a window every lane first gathers the u values that do not come from this
// window (old values and values finished by earlier windows), then the window
// is resolved in `win_depth+1` steps: a lane at dependency depth d sums its
// row in ascending column order, taking in-window producers from LDS.  Any
// execution that honours the dependencies and keeps each row's summation order
// reproduces the sequential sweep bit for bit.
//   MODE 0: SparseGaussSeidel update  (b - rsum)/diag, unchanged if diag == 0
//   MODE 1: AMG::Jacobi               (b - sigma)/aii  (aii = 0 if absent)
//   MODE 2: SOR  uk + omega*((b - s_less - s_greater)/aii - uk)
template <int BLOCK, int MAXE>
__global__ __launch_bounds__(BLOCK) void gs_lex_window(
    int64_t n_slots, int width, const int32_t* __restrict__ slot_row,
    const int16_t* __restrict__ slot_depth, const int32_t* __restrict__ win_depth,
    const int32_t* __restrict__ ecol, const double* __restrict__ eval,
    const int16_t* __restrict__ esrc, const double* __restrict__ b, double* u,
    int mode, double omega) {
  __shared__ double pub[BLOCK];
  const int lane = threadIdx.x;
  const int64_t n_win = n_slots / BLOCK;
  for (int64_t w = 0; w < n_win; ++w) {
    const int64_t slot = w * BLOCK + lane;
    const int row = slot_row[slot];
    const int depth = slot_depth[slot];
    const int wd = win_depth[w];
    // per entry: kind 0 = absent / diagonal, 1 = term of the (first) sum, 2 = term
    // of SOR's second sum (j > i, smoother.hpp:354-357); term = product that is
    // already known (operand outside this window) or v * (value published in LDS).
    // Everything is written branch-free (clamped addresses + selects) so that the
    // loads of a phase are issued back to back instead of one wait per entry.
    int sidx[MAXE];   // LDS slot of the in-window producer (0 when none)
    bool dep[MAXE];
    int kind[MAXE];
    double v[MAXE], term[MAXE];
    const bool live = row >= 0;
    const int rowc = live ? row : 0;
    const double bi = b[rowc];
    const double uk = u[rowc];
    double diag = 0.0;
    int32_t c[MAXE];
    int32_t sr[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const int ee = e < width ? e : 0;
      const int64_t at = (int64_t)ee * n_slots + slot;
      c[e] = ecol[at];
      v[e] = eval[at];
      sr[e] = esrc[at];
    }
    double ug[MAXE];
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const bool entry = live && e < width && c[e] >= 0;
      ug[e] = u[entry ? c[e] : 0];
    }
#pragma unroll
    for (int e = 0; e < MAXE; ++e) {
      const bool entry = live && e < width && c[e] >= 0;
      const bool isdiag = entry && c[e] == row;
      diag = isdiag ? v[e] : diag;
      const bool off = entry && !isdiag;
      dep[e] = off && sr[e] >= 0;
      sidx[e] = dep[e] ? sr[e] : 0;
      kind[e] = off ? ((mode == 2 && c[e] > row) ? 2 : 1) : 0;
      term[e] = v[e] * ug[e];  // operand final (earlier window) or old value
    }
    double unew = uk;
    for (int d = 0; d <= wd; ++d) {
      double pv[MAXE];
#pragma unroll
      for (int e = 0; e < MAXE; ++e) pv[e] = pub[sidx[e]];
      double s = 0.0, s2 = 0.0;
#pragma unroll
      for (int e = 0; e < MAXE; ++e) {
        const double t = dep[e] ? v[e] * pv[e] : term[e];
        s = (kind[e] == 1) ? s + t : s;
        s2 = (kind[e] == 2) ? s2 + t : s2;
      }
      double cand;
      if (mode == 0) cand = (diag == 0.0) ? uk : (bi - s) / diag;
      else if (mode == 1) cand = (bi - s) / diag;
      else cand = uk + omega * ((bi - s - s2) / diag - uk);
      unew = (live && depth == d) ? cand : unew;
      __syncthreads();       // everybody has read pub of the previous step
      pub[lane] = unew;      // final once depth <= d; earlier values are never consumed
      __syncthreads();
    }
    if (live) u[row] = unew;
    __threadfence_block();
    __syncthreads();
  }
}

template <int BLOCK>
static hipError_t launch_gs_lex_b(const LexDev& S, const double* b, double* u, int mode,
                                  double omega, hipStream_t st) {
#define AMG_LEX(M) hipLaunchKernelGGL((gs_lex_window<BLOCK, M>), dim3(1), dim3(BLOCK), 0, st, S.n_slots, \
                                      S.width, S.row, S.depth, S.win_depth, S.col, S.val, S.src, b, u, mode, omega)
  if (S.width <= 5) AMG_LEX(5);
  else if (S.width <= 7) AMG_LEX(7);
  else if (S.width <= 9) AMG_LEX(9);
  else if (S.width <= 16) AMG_LEX(16);
  else if (S.width <= 32) AMG_LEX(32);  // irregular coarse operators (strength-based coarsening, 3-D)
  else AMG_LEX(64);
#undef AMG_LEX
  return hipGetLastError();
}
hipError_t launch_gs_lex(const LexDev& S, const double* b, double* u, int mode, double omega,
                         hipStream_t st) {
  if (S.n_slots <= 0) return hipSuccess;
  if (S.width > 64) return hipErrorInvalidValue;
  switch (S.block) {
    case 64: return launch_gs_lex_b<64>(S, b, u, mode, omega, st);
    case 256: return launch_gs_lex_b<256>(S, b, u, mode, omega, st);
    case 1024: return launch_gs_lex_b<1024>(S, b, u, mode, omega, st);
  }
  return hipErrorInvalidValue;
}

static int g_scan_typed = 1;
static int g_scan_long = [] {  // chunks of more than 1024 rows (AMG_HIP_SCAN_LONG=0: the round-2 walk)
  const char* e = std::getenv("AMG_HIP_SCAN_LONG");
  return (e && *e == '0') ? 0 : 1;
}();
bool gs_scan_long_chunks_ok(const DictRef& D) { return g_scan_long && D.rtype && g_scan_typed && D.scan_new <= 4; }
void set_scan_typed(int on) { g_scan_typed = on ? 1 : 0; }
// --------------------------------------------------------------- K-GS-scan ---
// Lexicographic Gauss-Seidel at size (smoother.hpp:148-174).  On the reference's Galerkin
// levels row k of a forward sweep needs the NEW values of k-1 (the flat-index chain, SURVEY
// F9) and of rows at least g = m-1 back (the previous grid line); everything else is old.
// Inside a chunk of C <= g consecutive rows only the chain is unresolved, and it is a
// first-order linear recurrence
//     u_k = c_k + q_k u_{k-1},   c_k = (b_k - S_new - S_old) / a_kk,   q_k = -a_{k,k-1} / a_kk
// (SOR: c, q blended with omega and the old u_k) that a parallel affine scan solves in
// log C steps: (q, c) o (Q, C) = (q Q, c + q C).  ONE workgroup of 1024 threads walks the
// level chunk by chunk: S_old (terms of rows not yet swept) is formed one chunk ahead from
// global memory, S_new from an LDS ring of the last g + C new values, wave-level scan with
// DPP shuffles, wave totals combined through LDS.  Same sweep, different rounding order
// (|q| ~ 0.13-0.23: errors do not grow): agrees with the sequential sweep to ~1e-13, which
// is why the exact dependency-scheduled kernel stays the default on small problems and
// under opt.exact_gs.  The backward sweep is the same walk from the last row with the
// roles of lower and upper entries swapped.
//   mode 0: SparseGaussSeidel update (rows without diagonal keep their value)
//   mode 1: AMG::Jacobi (forward GS)      mode 2: SOR with omega
struct ScanEntry { double v; int32_t off; int32_t pad; };
// row structure of a dictionary-coded row: code words from the row type or the per-row codes
template <int WORDS>
__device__ __forceinline__ void scan_row_words(uint64_t (&cw)[2], int k, const uint64_t* __restrict__ codes,
                                               const uint8_t* __restrict__ rtype, const uint64_t* wtab) {
  cw[0] = cw[1] = ~(uint64_t)0;
  if (rtype) {
    const uint32_t ty = rtype[k];
#pragma unroll
    for (int w = 0; w < WORDS; ++w) cw[w] = wtab[ty * WORDS + w];
  } else {
#pragma unroll
    for (int w = 0; w < WORDS; ++w) cw[w] = codes[(int64_t)k * WORDS + w];
  }
}
__device__ __forceinline__ void scan_stage_tables(ScanEntry* tab, uint64_t* wtab, int words,
                                                  const uint8_t* rtype, const uint64_t* __restrict__ rwords,
                                                  const int32_t* __restrict__ doff,
                                                  const double* __restrict__ dval, int ntab) {
  const int t = threadIdx.x;
  if (t < 256) {
    ScanEntry e;
    e.v = t < ntab ? dval[t] : 0.0;
    e.off = t < ntab ? doff[t] : 0;
    e.pad = 0;
    tab[t] = e;
    if (rtype)
      for (int k = 0; k < words; ++k) wtab[t * words + k] = rwords[t * words + k];
  }
}
// Pre-pass, parallel over the whole level: the part of every row sum that a sweep takes from
// rows it has not reached yet (forward: columns above the diagonal, backward: below) -- old
// values, known before the sweep starts.  s_old[k] = sum in ascending column order.
template <int WORDS, int UN>
__global__ __launch_bounds__(256) void gs_scan_prep_kernel(
    int n, const uint64_t* __restrict__ codes, const uint8_t* __restrict__ rtype,
    const uint64_t* __restrict__ rwords, const int32_t* __restrict__ doff,
    const double* __restrict__ dval, int ntab, const double* __restrict__ u, int backward,
    double* __restrict__ s_old) {
  __shared__ ScanEntry tab[256];
  __shared__ uint64_t wtab[256 * WORDS];
  scan_stage_tables(tab, wtab, WORDS, rtype, rwords, doff, dval, ntab);
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  uint64_t cw[2];
  scan_row_words<WORDS>(cw, k, codes, rtype, wtab);
  double uv[UN], vv[UN];
  bool side[UN];
#pragma unroll
  for (int e = 0; e < UN; ++e) {
    const uint32_t code = (uint32_t)(cw[e >> 3] >> (8 * (e & 7))) & 0xFFu;
    const ScanEntry en = tab[code == 0xFFu ? 0 : code];
    side[e] = code != 0xFFu && (backward ? en.off < 0 : en.off > 0);
    vv[e] = en.v;
    uv[e] = u[side[e] ? k + en.off : k];
  }
  double s = 0.0;
#pragma unroll
  for (int e = 0; e < UN; ++e) s = side[e] ? s + vv[e] * uv[e] : s;
  s_old[k] = s;
}

// wave-level inclusive scan of affine maps with DPP row shifts / broadcasts (no LDS crossbar
// round trips): after it lane i holds the composition of lanes 0..i
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double ident, double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(ident), __double2loint(v), CTRL, ROWMASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(ident), __double2hiint(v), CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void affine_step(double& Q, double& Cc) {
  const double Qp = dpp_f64<CTRL, ROWMASK>(1.0, Q), Cp = dpp_f64<CTRL, ROWMASK>(0.0, Cc);
  Cc = Cc + Q * Cp;   // (q, c) o (Qp, Cp): lanes without a source see the identity (1, 0)
  Q = Q * Qp;
}
__device__ __forceinline__ void affine_scan_wave(double& Q, double& Cc) {
  affine_step<0x111, 0xF>(Q, Cc);  // row_shr:1
  affine_step<0x112, 0xF>(Q, Cc);  // row_shr:2
  affine_step<0x114, 0xF>(Q, Cc);  // row_shr:4
  affine_step<0x118, 0xF>(Q, Cc);  // row_shr:8
  affine_step<0x142, 0xA>(Q, Cc);  // row_bcast:15 into rows 1 and 3
  affine_step<0x143, 0xC>(Q, Cc);  // row_bcast:31 into rows 2 and 3
}

template <int WORDS, int UN>
__global__ __launch_bounds__(1024) void gs_scan_kernel(
    int n, const uint64_t* __restrict__ codes, const uint8_t* __restrict__ rtype,
    const uint64_t* __restrict__ rwords, const int32_t* __restrict__ doff,
    const double* __restrict__ dval, int ntab, const double* __restrict__ b, double* u,
    const double* __restrict__ s_old, int backward, int mode, double omega, int C, int ringmask) {
  extern __shared__ double ring[];           // new values of the last rows, by row & ringmask
  __shared__ ScanEntry tab[256];
  __shared__ uint64_t wtab[256 * WORDS];
  __shared__ double totQ[16], totC[16], carry_in[16];
  __shared__ double last_u;                  // new value of the row before this chunk
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  scan_stage_tables(tab, wtab, WORDS, rtype, rwords, doff, dval, ntab);
  for (int i = t; i <= ringmask; i += (int)blockDim.x) ring[i] = 0.0;
  if (t == 0) last_u = 0.0;
  __syncthreads();
  const int nchunks = (n + C - 1) / C;
  const int chain = backward ? 1 : -1;       // offset of the chained neighbour
  // operands of one row, fetched two chunks ahead
  // (only global loads here, nothing that depends on them: the type -> code word lookup
  // happens when the row is used, two chunks later)
  struct Row {
    uint64_t cw[2];
    uint32_t ty;
    double bk, uk, so;
    int k;
    bool live;
  };
  auto fetch = [&](int chunk, Row& r) {
    const int idx = chunk * C + t;
    r.live = chunk < nchunks && t < C && idx < n;
    r.k = backward ? n - 1 - idx : idx;
    r.cw[0] = r.cw[1] = ~(uint64_t)0;
    r.ty = 255;
    r.bk = r.uk = r.so = 0.0;
    if (!r.live) return;
    if (rtype) {
      r.ty = rtype[r.k];
    } else {
#pragma unroll
      for (int w = 0; w < WORDS; ++w) r.cw[w] = codes[(int64_t)r.k * WORDS + w];
    }
    r.bk = b[r.k];
    r.uk = u[r.k];
    r.so = s_old[r.k];
  };
  // waves beyond the chunk length only take part in the barriers
  const bool active = wave * 64 < C;
  const int nwaves = (C + 63) >> 6;
  Row cur, nx1, nx2;
  cur.live = nx1.live = nx2.live = false;
  if (active) {
    fetch(0, cur);
    fetch(1, nx1);
  }
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    double Q = 1.0, Cc = 0.0;
    if (active) {
      fetch(chunk + 2, nx2);  // in flight during two scans
      // (q, c) of this row from the new values of earlier chunks
      double q = 0.0, c = 0.0;
      if (cur.live) {
        if (rtype) {
#pragma unroll
          for (int w = 0; w < WORDS; ++w) cur.cw[w] = wtab[cur.ty * WORDS + w];
        }
        double s_new = 0.0, diag = 0.0, achain = 0.0;
#pragma unroll
        for (int e = 0; e < UN; ++e) {
          const uint32_t code = (uint32_t)(cur.cw[e >> 3] >> (8 * (e & 7))) & 0xFFu;
          const ScanEntry en = tab[code == 0xFFu ? 0 : code];
          const bool used = code != 0xFFu;
          const bool isdiag = used && en.off == 0;
          const bool ischain = used && en.off == chain;
          const bool new_side = used && !ischain && (backward ? en.off > 0 : en.off < 0);
          diag = isdiag ? en.v : diag;
          achain = ischain ? en.v : achain;
          const double rv = ring[(cur.k + (new_side ? en.off : 0)) & ringmask];
          s_new = new_side ? s_new + en.v * rv : s_new;
        }
        const double rsum = s_new + cur.so;
        if (mode == 0 && diag == 0.0) {       // smoother.hpp:134-137: row left alone
          c = cur.uk;
          q = 0.0;
        } else {
          const double g = (cur.bk - rsum) / diag;
          const double qq = -achain / diag;
          if (mode == 2) {                    // u_k + omega (g - u_k), smoother.hpp:360-362
            c = cur.uk + omega * (g - cur.uk);
            q = omega * qq;
          } else {
            c = g;
            q = qq;
          }
        }
      }
      // inclusive affine scan inside the wave: (Q, Cc) maps the value before the wave's
      // first row to this row's value
      Q = q;
      Cc = c;
      affine_scan_wave(Q, Cc);
      if (lane == 63) {
        totQ[wave] = Q;
        totC[wave] = Cc;
      }
    }
    lds_barrier();
    // second level: wave 0 turns the wave totals into the value BEFORE each wave's first row
    if (wave == 0) {
      double tq = lane < nwaves ? totQ[lane & 15] : 1.0, tc = lane < nwaves ? totC[lane & 15] : 0.0;
      affine_scan_wave(tq, tc);
      const double after = tc + tq * last_u;     // value of wave `lane`'s last row
      if (lane < 15) carry_in[lane + 1] = after;
      if (lane == 0) carry_in[0] = last_u;
    }
    lds_barrier();
    if (active) {
      const double unew = Cc + Q * carry_in[wave];
      if (cur.live) {
        ring[cur.k & ringmask] = unew;
        u[cur.k] = unew;
      }
      const int last_t = (chunk * C + C <= n ? C : n - chunk * C) - 1;
      if (t == last_t) last_u = unew;
    }
    lds_barrier();                             // ring and last_u are ready for the next chunk
    cur = nx1;
    nx1 = nx2;
  }
}

// ---- row-typed matrices: everything that does not depend on new values moves to the pre-pass
// The chain coefficient, the diagonal and the weights of the new-side entries are properties
// of the row TYPE.  The pre-pass therefore leaves ONE number per row,
//   P_k = (b_k - s_old_k) / a_kk          (SOR: u_k + omega (that - u_k); rows that SpGS leaves
//                                          alone: u_k, with all weights 0),
// and the serial kernel evaluates  c_k = P_k - sum_e w_e u_new[k + off_e],  q_k = Q  with
// w_e = a_e / a_kk (x omega), Q = -a_chain / a_kk (x omega) from a per-type LDS table: no code
// words, no division and at most SCAN_NEW ring reads on the serial path.
__device__ __forceinline__ double readlane_f64(double v, int lane);  // K-Band section below
constexpr int SCAN_NEW = 4;  // new-side entries other than the chain, per row type (2-D: 3)
// per-type table as three LDS arrays: tq[256], tw[256 * SCAN_NEW], toff[256 * SCAN_NEW];
// thread t < 256 builds the entry of type t from its code words (type 255 / unused: zeros)
template <int WORDS, int UN>
__device__ __forceinline__ void scan_build_types(double* tq, double* tw, int32_t* toff,
                                                 const ScanEntry* tab, const uint64_t* wtab,
                                                 int backward, int mode, double omega) {
  const int t = threadIdx.x;
  if (t >= 256) return;
  const int chain = backward ? 1 : -1;
  double diag = 0.0, achain = 0.0;
  for (int e = 0; e < UN; ++e) {
    const uint32_t code = (uint32_t)(wtab[t * WORDS + (e >> 3)] >> (8 * (e & 7))) & 0xFFu;
    if (code != 0xFFu) {
      const ScanEntry en = tab[code];
      if (en.off == 0) diag = en.v;
      if (en.off == chain) achain = en.v;
    }
  }
  const bool frozen = mode == 0 && diag == 0.0;  // smoother.hpp:134-137
  const double sc = mode == 2 ? omega : 1.0;
  for (int e = 0; e < SCAN_NEW; ++e) {
    tw[t * SCAN_NEW + e] = 0.0;
    toff[t * SCAN_NEW + e] = 0;
  }
  tq[t] = frozen ? 0.0 : sc * (-achain / diag);
  int nn = 0;
  for (int e = 0; e < UN; ++e) {
    const uint32_t code = (uint32_t)(wtab[t * WORDS + (e >> 3)] >> (8 * (e & 7))) & 0xFFu;
    if (code == 0xFFu || frozen) continue;
    const ScanEntry en = tab[code];
    const bool new_side = en.off != chain && (backward ? en.off > 0 : en.off < 0);
    if (new_side && nn < SCAN_NEW) {
      tw[t * SCAN_NEW + nn] = sc * (en.v / diag);
      toff[t * SCAN_NEW + nn] = en.off;
      ++nn;
    }
  }
}
template <int WORDS, int UN>
__global__ __launch_bounds__(256) void gs_scan_prep_typed_kernel(
    int n, const uint8_t* __restrict__ rtype, const uint64_t* __restrict__ rwords,
    const int32_t* __restrict__ doff, const double* __restrict__ dval, int ntab,
    const double* __restrict__ b, const double* __restrict__ u, int backward, int mode, double omega,
    double* __restrict__ P) {
  __shared__ ScanEntry tab[256];
  __shared__ uint64_t wtab[256 * WORDS];
  scan_stage_tables(tab, wtab, WORDS, rtype, rwords, doff, dval, ntab);
  __syncthreads();
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  uint64_t cw[2];
  scan_row_words<WORDS>(cw, k, rwords, rtype, wtab);  // (codes unused: the matrix is row-typed)
  double uv[UN], vv[UN];
  bool side[UN];
  double diag = 0.0;
#pragma unroll
  for (int e = 0; e < UN; ++e) {
    const uint32_t code = (uint32_t)(cw[e >> 3] >> (8 * (e & 7))) & 0xFFu;
    const ScanEntry en = tab[code == 0xFFu ? 0 : code];
    side[e] = code != 0xFFu && (backward ? en.off < 0 : en.off > 0);
    diag = (code != 0xFFu && en.off == 0) ? en.v : diag;
    vv[e] = en.v;
    uv[e] = u[side[e] ? k + en.off : k];
  }
  double s = 0.0;
#pragma unroll
  for (int e = 0; e < UN; ++e) s = side[e] ? s + vv[e] * uv[e] : s;
  const double uk = u[k];
  double p;
  if (mode == 0 && diag == 0.0) {
    p = uk;
  } else {
    const double g = (b[k] - s) / diag;
    p = mode == 2 ? uk + omega * (g - uk) : g;
  }
  P[k] = p;
}
template <int WORDS, int UN>
__global__ __launch_bounds__(1024) void gs_scan_typed_kernel(
    int n, const uint8_t* __restrict__ rtype, const uint64_t* __restrict__ rwords,
    const int32_t* __restrict__ doff, const double* __restrict__ dval, int ntab, double* u,
    const double* __restrict__ P, int backward, int mode, double omega, int C, int ringmask) {
  extern __shared__ double ring[];           // new values of the last rows, by row & ringmask
  __shared__ ScanEntry tab[256];
  __shared__ uint64_t wtab[256 * WORDS];
  __shared__ double tq[256];
  __shared__ double tw[256 * SCAN_NEW];
  __shared__ int32_t toff[256 * SCAN_NEW];
  __shared__ double totQ[16], totC[16], carry_in[16];
  __shared__ double last_u;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  scan_stage_tables(tab, wtab, WORDS, rtype, rwords, doff, dval, ntab);
  for (int i = t; i <= ringmask; i += (int)blockDim.x) ring[i] = 0.0;
  if (t == 0) last_u = 0.0;
  __syncthreads();
  scan_build_types<WORDS, UN>(tq, tw, toff, tab, wtab, backward, mode, omega);
  __syncthreads();
  const int nchunks = (n + C - 1) / C;
  struct Row {
    double p;
    uint32_t ty;
    int k;
    bool live;
  };
  auto fetch = [&](int chunk, Row& r) {
    const int idx = chunk * C + t;
    r.live = chunk < nchunks && t < C && idx < n;
    r.k = backward ? n - 1 - idx : idx;
    r.ty = 255;
    r.p = 0.0;
    if (!r.live) return;
    r.ty = rtype[r.k];
    r.p = P[r.k];
  };
  const bool active = wave * 64 < C;
  const int nwaves = (C + 63) >> 6;
  Row cur, nx1, nx2;
  cur.live = nx1.live = nx2.live = false;
  if (active) {
    fetch(0, cur);
    fetch(1, nx1);
  }
  if (nwaves == 1) {
    // A chunk of at most 64 rows (the deep levels: short grid lines) is ONE wave's scan: no
    // cross-wave combine, no barrier -- the wave's LDS accesses keep their order -- and the
    // carry travels in a register.  Same products and sums as the general path below.
    if (wave != 0) return;
    double carry = 0.0;  // wave-uniform: the last new value of the previous chunk
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      fetch(chunk + 2, nx2);
      double q = 0.0, c = 0.0;
      if (cur.live) {
        const int tb = (int)cur.ty * SCAN_NEW;
        double rv[SCAN_NEW], wv[SCAN_NEW];
#pragma unroll
        for (int e = 0; e < SCAN_NEW; ++e) {
          wv[e] = tw[tb + e];
          rv[e] = ring[(cur.k + toff[tb + e]) & ringmask];
        }
        c = cur.p;
#pragma unroll
        for (int e = 0; e < SCAN_NEW; ++e) c -= wv[e] * rv[e];
        q = tq[cur.ty];
      }
      double Q = q, Cc = c;
      affine_scan_wave(Q, Cc);
      const double unew = Cc + Q * carry;
      if (cur.live) {
        ring[cur.k & ringmask] = unew;
        u[cur.k] = unew;
      }
      const int last_t = (chunk * C + C <= n ? C : n - chunk * C) - 1;
      carry = readlane_f64(unew, last_t);
      cur = nx1;
      nx1 = nx2;
    }
    return;
  }
  uint32_t cty = 0xFFFFFFFFu;  // row type whose weights the lane holds
  double cwv[SCAN_NEW], cqv = 0.0;
  int32_t cof[SCAN_NEW];
#pragma unroll
  for (int e = 0; e < SCAN_NEW; ++e) {
    cwv[e] = 0.0;
    cof[e] = 0;
  }
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    double Q = 1.0, Cc = 0.0;
    if (active) {
      fetch(chunk + 2, nx2);  // in flight during two scans
      double q = 0.0, c = 0.0;
      // the weights of the lane's row type stay in registers from chunk to chunk: a lane walks
      // down one grid column, whose rows share a type except at the first and last lines, so the
      // nine table reads per row (of thirteen LDS reads) are skipped while no lane of the wave
      // changes type -- one workgroup's LDS instruction rate is what bounds this kernel
      if (__builtin_amdgcn_ballot_w64(cur.ty != cty) != 0) {
        const int tb = (int)cur.ty * SCAN_NEW;
#pragma unroll
        for (int e = 0; e < SCAN_NEW; ++e) {
          cwv[e] = tw[tb + e];
          cof[e] = toff[tb + e];
        }
        cqv = tq[cur.ty];
        cty = cur.ty;
      }
      if (cur.live) {
        double rv[SCAN_NEW];
#pragma unroll
        for (int e = 0; e < SCAN_NEW; ++e) rv[e] = ring[(cur.k + cof[e]) & ringmask];
        c = cur.p;
#pragma unroll
        for (int e = 0; e < SCAN_NEW; ++e) c -= cwv[e] * rv[e];   // unused slots: weight 0
        q = cqv;
      }
      Q = q;
      Cc = c;
      affine_scan_wave(Q, Cc);
      if (lane == 63) {
        totQ[wave] = Q;
        totC[wave] = Cc;
      }
    }
    lds_barrier();
    if (wave == 0) {
      double tq = lane < nwaves ? totQ[lane & 15] : 1.0, tc = lane < nwaves ? totC[lane & 15] : 0.0;
      affine_scan_wave(tq, tc);
      const double after = tc + tq * last_u;
      if (lane < 15) carry_in[lane + 1] = after;
      if (lane == 0) carry_in[0] = last_u;
    }
    lds_barrier();
    if (active) {
      const double unew = Cc + Q * carry_in[wave];
      if (cur.live) {
        ring[cur.k & ringmask] = unew;
        u[cur.k] = unew;
      }
      const int last_t = (chunk * C + C <= n ? C : n - chunk * C) - 1;
      if (t == last_t) last_u = unew;
    }
    lds_barrier();
    cur = nx1;
    nx1 = nx2;
  }
}

// The same walk with R = 2 or 4 CONSECUTIVE rows per thread, for chunks of more than 1024 rows (grid
// lines of 2048 / 4096 and more: the finest levels of the big grids).  A chunk may be as long as
// the distance to the previous grid line, and the serial walk pays per CHUNK (three barriers, two
// scan levels), so a 4096-wide level takes a quarter of the chunks.  A thread folds its R affine
// maps into one, the wave and workgroup scans run on the folded maps as before, and the thread
// then walks its rows from the value before its block (the exclusive prefix: the inclusive result
// of the lane below, through one cross-lane shift).  Same recurrence; the association of the
// products differs from the one-row kernel (as that one's differs from the sequential sweep):
// 1e-13 class.
template <int WORDS, int UN, int R>
__global__ __launch_bounds__(1024) void gs_scan_typed_rows_kernel(
    int n, const uint8_t* __restrict__ rtype, const uint64_t* __restrict__ rwords,
    const int32_t* __restrict__ doff, const double* __restrict__ dval, int ntab, double* u,
    const double* __restrict__ P, int backward, int mode, double omega, int C, int ringmask) {
  extern __shared__ double ring[];
  __shared__ ScanEntry tab[256];
  __shared__ uint64_t wtab[256 * WORDS];
  __shared__ double tq[256];
  __shared__ double tw[256 * SCAN_NEW];
  __shared__ int32_t toff[256 * SCAN_NEW];
  __shared__ double totQ[16], totC[16], carry_in[16];
  __shared__ double last_u;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  scan_stage_tables(tab, wtab, WORDS, rtype, rwords, doff, dval, ntab);
  for (int i = t; i <= ringmask; i += (int)blockDim.x) ring[i] = 0.0;
  if (t == 0) last_u = 0.0;
  __syncthreads();
  scan_build_types<WORDS, UN>(tq, tw, toff, tab, wtab, backward, mode, omega);
  __syncthreads();
  const int nchunks = (n + C - 1) / C;
  struct Row {
    double p;
    uint32_t ty;
    int k;
    bool live;
  };
  auto fetch = [&](int chunk, Row (&rw)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int j = t * R + r;
      const int idx = chunk * C + j;
      rw[r].live = chunk < nchunks && j < C && idx < n;
      rw[r].k = backward ? n - 1 - idx : idx;
      rw[r].ty = 255;
      rw[r].p = 0.0;
      if (rw[r].live) {
        rw[r].ty = rtype[rw[r].k];
        rw[r].p = P[rw[r].k];
      }
    }
  };
  const int nwaves = ((C + R - 1) / R + 63) >> 6;
  const bool active = wave < nwaves;
  Row cur[R], nx1[R], nx2[R];
#pragma unroll
  for (int r = 0; r < R; ++r) cur[r].live = nx1[r].live = nx2[r].live = false;
  if (active) {
    fetch(0, cur);
    fetch(1, nx1);
  }
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    double q[R], c[R];
    double Q = 1.0, Cc = 0.0;
    if (active) {
      fetch(chunk + 2, nx2);  // in flight during two scans
#pragma unroll
      for (int r = 0; r < R; ++r) {
        q[r] = 0.0;
        c[r] = 0.0;
        // (the one-row kernel keeps the weights of a lane's row type in registers from chunk to
        // chunk; with R rows per lane that costs more registers than it saves LDS reads: 4096^2
        // 221 -> 241 ms per cycle, not kept here)
        if (cur[r].live) {
          const int tb = (int)cur[r].ty * SCAN_NEW;
          double rv[SCAN_NEW], wv[SCAN_NEW];
#pragma unroll
          for (int e = 0; e < SCAN_NEW; ++e) {
            wv[e] = tw[tb + e];
            rv[e] = ring[(cur[r].k + toff[tb + e]) & ringmask];
          }
          double cc = cur[r].p;
#pragma unroll
          for (int e = 0; e < SCAN_NEW; ++e) cc -= wv[e] * rv[e];   // unused slots: weight 0
          c[r] = cc;
          q[r] = tq[cur[r].ty];
        }
      }
      // the thread's R maps folded into one: (q_r, c_r) o (Q, Cc) = (q_r Q, c_r + q_r Cc)
      Q = q[0];
      Cc = c[0];
#pragma unroll
      for (int r = 1; r < R; ++r) {
        Cc = c[r] + q[r] * Cc;
        Q = q[r] * Q;
      }
      affine_scan_wave(Q, Cc);
      if (lane == 63) {
        totQ[wave] = Q;
        totC[wave] = Cc;
      }
    }
    lds_barrier();
    if (wave == 0) {
      double sq = lane < nwaves ? totQ[lane & 15] : 1.0, sc = lane < nwaves ? totC[lane & 15] : 0.0;
      affine_scan_wave(sq, sc);
      const double after = sc + sq * last_u;
      if (lane < 15) carry_in[lane + 1] = after;
      if (lane == 0) carry_in[0] = last_u;
    }
    lds_barrier();
    if (active) {
      // value before this thread's block: the inclusive result of the lane below applied to the
      // wave's carry (lane 0: the carry itself)
      double Qp = __shfl_up(Q, 1), Cp = __shfl_up(Cc, 1);
      if (lane == 0) {
        Qp = 1.0;
        Cp = 0.0;
      }
      double v = Cp + Qp * carry_in[wave];
      const int len = (chunk * C + C <= n ? C : n - chunk * C);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        v = c[r] + q[r] * v;
        if (cur[r].live) {
          ring[cur[r].k & ringmask] = v;
          u[cur[r].k] = v;
        }
        if (t * R + r == len - 1) last_u = v;
      }
    }
    lds_barrier();                             // ring and last_u are ready for the next chunk
#pragma unroll
    for (int r = 0; r < R; ++r) {
      cur[r] = nx1[r];
      nx1[r] = nx2[r];
    }
  }
}
constexpr int GS_SCAN_MAX_CHUNK = 4096;
int gs_scan_max_chunk() { return GS_SCAN_MAX_CHUNK; }

template <int WORDS, int UN>
static hipError_t launch_gs_scan_wu(int64_t n, const DictRef& D, const double* b, double* u,
                                    double* s_old, int backward, int mode, double omega, int C,
                                    int ring, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gs_scan_kernel<WORDS, UN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    attr_set = true;
  }
  if (D.rtype && g_scan_typed && D.scan_new <= SCAN_NEW) {
    static bool attr2 = false;
    if (!attr2) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gs_scan_typed_kernel<WORDS, UN>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      attr2 = true;
    }
    hipLaunchKernelGGL((gs_scan_prep_typed_kernel<WORDS, UN>), dim3((unsigned)((n + 255) / 256)), dim3(256),
                       0, st, (int)n, D.rtype, D.rwords, D.doff, D.dval, D.ntab, b, u, backward, mode, omega,
                       s_old);
    if (C > 1024) {  // long grid lines: 2 or 4 consecutive rows per thread
      static bool attr3 = false;
      if (!attr3) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gs_scan_typed_rows_kernel<WORDS, UN, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gs_scan_typed_rows_kernel<WORDS, UN, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        attr3 = true;
      }
      if (C <= 2048)
        hipLaunchKernelGGL((gs_scan_typed_rows_kernel<WORDS, UN, 2>), dim3(1), dim3(1024), (size_t)ring * 8, st,
                           (int)n, D.rtype, D.rwords, D.doff, D.dval, D.ntab, u, s_old, backward, mode, omega, C,
                           ring - 1);
      else
        hipLaunchKernelGGL((gs_scan_typed_rows_kernel<WORDS, UN, 4>), dim3(1), dim3(1024), (size_t)ring * 8, st,
                           (int)n, D.rtype, D.rwords, D.doff, D.dval, D.ntab, u, s_old, backward, mode, omega, C,
                           ring - 1);
      return hipGetLastError();
    }
    // as many waves as the chunk has rows (at least the four that stage the tables): the three
    // barriers per chunk cost what the waves they hold up cost, and a deep level's chunk is short
    const int nt = std::max(256, (C + 63) / 64 * 64);
    hipLaunchKernelGGL((gs_scan_typed_kernel<WORDS, UN>), dim3(1), dim3(nt), (size_t)ring * 8, st, (int)n,
                       D.rtype, D.rwords, D.doff, D.dval, D.ntab, u, s_old, backward, mode, omega, C,
                       ring - 1);
    return hipGetLastError();
  }
  if (C > 1024) return hipErrorInvalidValue;  // only the row-typed form walks long chunks
  hipLaunchKernelGGL((gs_scan_prep_kernel<WORDS, UN>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                     (int)n, D.codes, D.rtype, D.rwords, D.doff, D.dval, D.ntab, u, backward, s_old);
  const int nt = std::max(256, (C + 63) / 64 * 64);
  hipLaunchKernelGGL((gs_scan_kernel<WORDS, UN>), dim3(1), dim3(nt), (size_t)ring * 8, st, (int)n,
                     D.codes, D.rtype, D.rwords, D.doff, D.dval, D.ntab, b, u, s_old, backward, mode,
                     omega, C, ring - 1);
  return hipGetLastError();
}
// C: rows per chunk (<= the distance of the nearest non-chain dependency; <= 1024, or <= 4096 on a
// row-typed matrix with at most SCAN_NEW new-side entries per row: gs_scan_long_chunks_ok);
// ring: power of two >= largest dependency distance + C + 1, at most 16384 doubles
hipError_t launch_gs_scan(int64_t n, const DictRef& D, const double* b, double* u, double* s_old,
                          bool backward, int mode, double omega, int C, int ring, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (!dict_args_ok(n, D.words, D.wmax, D.ntab) || C < 1 || C > GS_SCAN_MAX_CHUNK || ring < 2 || ring > 16384 ||
      (ring & (ring - 1)) || !s_old)
    return hipErrorInvalidValue;
  hipError_t r = hipSuccess;
  (void)dict_dispatch_wu(D.words, D.wmax, [&](auto W, auto U) {
    r = launch_gs_scan_wu<decltype(W)::value, decltype(U)::value>(n, D, b, u, s_old, backward ? 1 : 0,
                                                                  mode, omega, C, ring, st);
  });
  return r;
}

// ------------------------------------------------------- K-Halo (in-graph comm) ---
// Neighbour exchange as ONE graph-capturable kernel: the host-driven version
// (hipMemcpyAsync + hipStreamWrite/WaitValue32, solver.cpp) costs ~3 us of host
// time per stream operation and those operations cannot be captured in a hipGraph.
// Protocol per channel and direction, words in the RECEIVER's arena, values 0/1:
//   DATA  set to 1 by the sender after its boundary values landed in the
//         receiver's halo slots; reset to 0 by the receiver when it has seen it;
//   FREE  set to 1 by the receiver (halo_ack_kernel) once the kernel that read the
//         halo is done; reset to 0 by the sender before it overwrites the slots.
// One workgroup: lane 0 waits for FREE, all lanes copy (stores to the peer's
// hipIpc-mapped memory), system fence, lane 0 publishes DATA and waits for its
// own.  All flag accesses are system-scope atomics; every spin is bounded
// (~2 s of wall clock) and reports through `timeout`.
__device__ __forceinline__ bool spin_until_one(uint32_t* w, uint32_t* timeout) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(w, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != 1u) {
    if (wall_clock64() - t0 > 200000000ull) {  // 100 MHz constant clock
      __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(8);
  }
  __hip_atomic_store(w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return true;
}

__global__ __launch_bounds__(256) void halo_exchange_kernel(HaloArgs a) {
  const int t = threadIdx.x;
  if (t == 0) {
    if (a.cnt_prev > 0) spin_until_one(a.my_free_from_prev, a.timeout);
    if (a.cnt_next > 0) spin_until_one(a.my_free_from_next, a.timeout);
  }
  __syncthreads();
  for (int64_t i = t; i < a.cnt_prev; i += 256)
    __builtin_nontemporal_store(a.src_prev[i], a.dst_prev + i);
  for (int64_t i = t; i < a.cnt_next; i += 256)
    __builtin_nontemporal_store(a.src_next[i], a.dst_next + i);
  __threadfence_system();
  __syncthreads();
  if (t == 0) {
    if (a.cnt_prev > 0)
      __hip_atomic_store(a.data_at_prev, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.cnt_next > 0)
      __hip_atomic_store(a.data_at_next, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.recv_prev) spin_until_one(a.my_data_from_prev, a.timeout);
    if (a.recv_next) spin_until_one(a.my_data_from_next, a.timeout);
    __threadfence_system();
  }
}
__global__ void halo_ack_kernel(uint32_t* free_at_prev, uint32_t* free_at_next) {
  if (threadIdx.x == 0) {
    if (free_at_prev) __hip_atomic_store(free_at_prev, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (free_at_next) __hip_atomic_store(free_at_next, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// Large halos (3-D: one x-y plane, MBs): same protocol split over three launches so
// that the copy can use many workgroups.
__global__ void halo_wait_free_kernel(HaloArgs a) {
  if (threadIdx.x == 0) {
    if (a.cnt_prev > 0) spin_until_one(a.my_free_from_prev, a.timeout);
    if (a.cnt_next > 0) spin_until_one(a.my_free_from_next, a.timeout);
  }
}
__global__ __launch_bounds__(256) void halo_copy_kernel(HaloArgs a) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x, nt = (int64_t)gridDim.x * 256;
  for (int64_t i = t; i < a.cnt_prev; i += nt) __builtin_nontemporal_store(a.src_prev[i], a.dst_prev + i);
  for (int64_t i = t; i < a.cnt_next; i += nt) __builtin_nontemporal_store(a.src_next[i], a.dst_next + i);
  __threadfence_system();
}
__global__ void halo_publish_wait_kernel(HaloArgs a) {
  if (threadIdx.x == 0) {
    if (a.cnt_prev > 0)
      __hip_atomic_store(a.data_at_prev, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.cnt_next > 0)
      __hip_atomic_store(a.data_at_next, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.recv_prev) spin_until_one(a.my_data_from_prev, a.timeout);
    if (a.recv_next) spin_until_one(a.my_data_from_next, a.timeout);
    __threadfence_system();
  }
}
hipError_t launch_halo_exchange(const HaloArgs& a, hipStream_t st) {
  const int64_t big = a.cnt_prev > a.cnt_next ? a.cnt_prev : a.cnt_next;
  if (big <= 16384) {
    hipLaunchKernelGGL(halo_exchange_kernel, dim3(1), dim3(256), 0, st, a);
  } else {
    int64_t g = (big + 2047) / 2048;
    if (g > 256) g = 256;
    hipLaunchKernelGGL(halo_wait_free_kernel, dim3(1), dim3(64), 0, st, a);
    hipLaunchKernelGGL(halo_copy_kernel, dim3((unsigned)g), dim3(256), 0, st, a);
    hipLaunchKernelGGL(halo_publish_wait_kernel, dim3(1), dim3(64), 0, st, a);
  }
  return hipGetLastError();
}
hipError_t launch_halo_ack(uint32_t* free_at_prev, uint32_t* free_at_next, hipStream_t st) {
  if (!free_at_prev && !free_at_next) return hipSuccess;
  hipLaunchKernelGGL(halo_ack_kernel, dim3(1), dim3(64), 0, st, free_at_prev, free_at_next);
  return hipGetLastError();
}

// All-gather of the agglomeration level's right-hand side by direct pushes:
// rank r copies its slice into every rank's full vector at offset `off` (its own
// included).  k1: wait until every destination released the slot (FREE); k2: copy;
// k3: publish DATA everywhere and wait for everybody's DATA.
__global__ void gather_wait_free_kernel(GatherArgs a) {
  const int g = threadIdx.x;
  if (g < a.world && g != a.rank) spin_until_one(a.my_free_from + g, a.timeout);
}
__global__ __launch_bounds__(256) void gather_copy_kernel(GatherArgs a) {
  const int g = blockIdx.y;
  double* dst = a.dst[g] + a.off;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.cnt; i += (int64_t)gridDim.x * 256)
    dst[i] = a.src[i];
  __threadfence_system();
}
__global__ void gather_publish_wait_kernel(GatherArgs a) {
  const int g = threadIdx.x;
  if (g < a.world && g != a.rank) {
    __hip_atomic_store(a.data_at[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    spin_until_one(a.my_data_from + g, a.timeout);
    __threadfence_system();
  }
}
__global__ void gather_ack_kernel(GatherArgs a) {
  const int g = threadIdx.x;
  if (g < a.world && g != a.rank)
    __hip_atomic_store(a.free_at[g], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_gather(const GatherArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(gather_wait_free_kernel, dim3(1), dim3(64), 0, st, a);
  int64_t gx = (a.cnt + 255) / 256;
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(gather_copy_kernel, dim3((unsigned)gx, (unsigned)a.world), dim3(256), 0, st, a);
  hipLaunchKernelGGL(gather_publish_wait_kernel, dim3(1), dim3(64), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_gather_ack(const GatherArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(gather_ack_kernel, dim3(1), dim3(64), 0, st, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------ K-Band ---
// x = A^-1 f with A = L D L^T banded (half-bandwidth w <= 63), ONE wave -- the
// substitution is a serial chain, so the design goal is the fewest instructions
// per step.  M = power of two > w lanes take part; lane l keeps the running
// right-hand side of the rows == l (mod M) in a register.  Forward step s: y_s
// is complete in lane s%M -> v_readlane broadcast -> every lane subtracts its
// pre-scheduled L entry times y_s (one multiply, one subtract, no FMA: the
// updates reach a row in ascending s, the order of a row-oriented substitution).
// The host lays L out per (step, lane) (host_setup.cpp: band_schedule) so a
// step's operands are ONE coalesced load with an immediate offset, fetched a
// whole 32-step chunk ahead.  Then z = y / D (IEEE divide) and the mirrored
// backward pass (row i lives in lane (n-1-i) % M).
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// v_writelane_b32 has no clang builtin on this toolchain.  dst[lane] = uniform
// value; `lane` is wave-uniform.  (SALU-written lane select, SGPR data operand:
// no software wait states needed on gfx9-family for this pair.)
template <int LANE>  // immediate lane select: an SGPR one would be a 2nd constant-bus read
__device__ __forceinline__ void writelane_f64(double& dst, double uniform_v) {
  int lo = __double2loint(dst), hi = __double2hiint(dst);
  const int ulo = __builtin_amdgcn_readfirstlane(__double2loint(uniform_v));
  const int uhi = __builtin_amdgcn_readfirstlane(__double2hiint(uniform_v));
  asm volatile("v_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
               : "+v"(lo), "+v"(hi)
               : "s"(ulo), "s"(uhi), "n"(LANE));
  dst = __hiloint2double(hi, lo);
}

constexpr int BAND_CH = 32;  // steps per register-prefetched chunk

struct BandChunk {
  double lc[BAND_CH];  // lane l < M: its L operand for each of the 32 steps
  double nx;           // lane i < 32: right-hand side of the row picked up after step i
};

// FWD: step s handles row s; else row n-1-s.  The schedule is zero-padded to a
// multiple of 64 steps (+64), so there are no bounds checks on it.
template <int M, bool FWD>
__device__ __forceinline__ void band_load(BandChunk& c, int64_t s0, int64_t n, int lane,
                                          const double* __restrict__ sched,
                                          const double* __restrict__ rhs) {
  if (lane < M) {
    const double* p = sched + s0 * M + lane;
#pragma unroll
    for (int i = 0; i < BAND_CH; ++i) c.lc[i] = p[i * M];
  } else {
#pragma unroll
    for (int i = 0; i < BAND_CH; ++i) c.lc[i] = 0.0;
  }
  // after step s0+i its owner lane picks up row s0+i+M; rows past the end read
  // as 0.0 so that the padded steps broadcast 0.0, never garbage
  const int64_t s2 = s0 + lane + M;
  const bool ok = lane < BAND_CH && s2 < n;
  const double v = rhs[ok ? (FWD ? s2 : n - 1 - s2) : 0];
  c.nx = ok ? v : 0.0;
}

// PH = (s0 / 32) % 2 (the owner of step s0+i is lane (PH*32 + i) % M).
template <int M, int PH, int I>
__device__ __forceinline__ void band_step(const BandChunk& c, double& acc, double& done) {
  constexpr int owner = (PH * BAND_CH + I) & (M - 1);
  const double vk = readlane_f64(acc, owner);
  acc -= c.lc[I] * vk;  // lc == 0 outside the band and for the owner itself
  writelane_f64<I>(done, vk);
  writelane_f64<owner>(acc, readlane_f64(c.nx, I));
}
template <int M, int PH, int... I>
__device__ __forceinline__ void band_step_seq(const BandChunk& c, double& acc, double& done,
                                              std::integer_sequence<int, I...>) {
  (band_step<M, PH, I>(c, acc, done), ...);
}
template <int M, bool FWD, int PH>
__device__ __forceinline__ void band_steps(const BandChunk& c, double& acc, int64_t s0,
                                           int64_t n, int lane, double* __restrict__ x) {
  double done = 0.0;  // lane i collects the value finished at step s0+i
  band_step_seq<M, PH>(c, acc, done, std::make_integer_sequence<int, BAND_CH>{});
  const int64_t s = s0 + lane;
  if (lane < BAND_CH && s < n) x[FWD ? s : n - 1 - s] = done;
}

template <int M, bool FWD>
__device__ __forceinline__ void band_pass(int64_t n, int lane, const double* __restrict__ sched,
                                          const double* __restrict__ rhs, double* __restrict__ x) {
  double acc = 0.0;
  if (lane < M && lane < n) acc = FWD ? rhs[lane] : rhs[n - 1 - lane];
  BandChunk A, B;
  band_load<M, FWD>(A, 0, n, lane, sched, rhs);
  for (int64_t s0 = 0; s0 < n; s0 += 2 * BAND_CH) {  // s0 % 64 == 0
    band_load<M, FWD>(B, s0 + BAND_CH, n, lane, sched, rhs);
    band_steps<M, FWD, 0>(A, acc, s0, n, lane, x);
    band_load<M, FWD>(A, s0 + 2 * BAND_CH, n, lane, sched, rhs);
    band_steps<M, FWD, 1>(B, acc, s0 + BAND_CH, n, lane, x);
  }
}

// y: scratch vector of n doubles (forward result, then z = y / D).
template <int M>
__global__ __launch_bounds__(64) void band_solve_kernel(
    int64_t n, const double* __restrict__ sched_f, const double* __restrict__ sched_b,
    const double* __restrict__ dg, const double* __restrict__ f, double* __restrict__ y,
    double* __restrict__ x) {
  const int lane = threadIdx.x;
  band_pass<M, true>(n, lane, sched_f, f, y);           // L y = f
  __threadfence_block();
  for (int64_t i = lane; i < n; i += 64) y[i] = y[i] / dg[i];  // z = y / D
  __threadfence_block();
  band_pass<M, false>(n, lane, sched_b, y, x);          // L^T x = z
}

// ------------------------------------------------------------- K-BandChain ----
// The same substitution for the NARROW coarsest operators of deep hierarchies (half-bandwidth
// w <= 3, a few hundred rows: 511 rows / w = 2 on the 16-level 4096^2 hierarchy), where the
// broadcast-and-update scheme above spends ~48 ns per step on lane traffic.  Here the whole
// problem sits in LDS and every lane walks the recurrence redundantly as plain scalar code:
//   y_s = rhs_s - sum_t c[s][t] * y_{s-w+t},  t = 0 .. w-1  (ascending k, farthest first:
// the order in which K-Band's updates reach a row, so the bits are the same),
// with the operands of the next 8 steps read ahead of the dependent multiply / subtract chain
// (one multiply and one subtract per step on the critical path).  The backward pass runs the
// same code on the mirrored system (step s = row n-1-s; host_setup.cpp: band_chain_schedule).
constexpr int CHAIN_CH = 8;  // steps per read-ahead chunk
__host__ __device__ inline int band_chain_pad(int n) { return (n + 2 * CHAIN_CH - 1) / (2 * CHAIN_CH) * (2 * CHAIN_CH) + 2 * CHAIN_CH; }
template <int W>
struct ChainChunk {
  double o[CHAIN_CH][W], r[CHAIN_CH];
};
// ops / v are padded to band_chain_pad(n) steps (zero operands): no bounds checks on reads
template <int W>
__device__ __forceinline__ void band_chain_fetch(ChainChunk<W>& c, const double* op, const double* rv) {
#pragma unroll
  for (int i = 0; i < CHAIN_CH; ++i) {
#pragma unroll
    for (int t = 0; t < W; ++t) c.o[i][t] = op[i * W + t];
    c.r[i] = rv[i];
  }
}
// One chunk of CHAIN_CH steps of y_i = ((r_i - o[i][0] y_{i-W}) - ...) - o[i][W-1] y_{i-1} (ascending
// distance order: the row-oriented substitution's).  Only the LAST product depends on the previous
// step: a step's critical path is one multiply and one subtract (two dependent fp64 operations,
// ~11 cycles each on gfx950: tools/fp64_chain.hip).  Everything before that product -- the
// "early" part r_i - sum_{t < W-1} o[i][t] y_{i-W+t} -- only needs results that are at least two
// steps old, so the early part of step i+1 is formed while step i's chain is in flight, in a
// PINNED instruction order (an in-order wave executes what it is given; left to the scheduler
// the four operations of a step came out as one dependent sequence, 31 ns per step):
//   m = o[i][W-1] * y_{i-1};  products of early_{i+1};  y_i = early_i - m;  subtractions of early_{i+1}
// Same operations on the same operands in the same order per row: same bits.
// nxt: the chunk after c (its first step's early part is formed by c's last step).
template <int W>
__device__ __forceinline__ void band_chain_steps(const ChainChunk<W>& c, const ChainChunk<W>& nxt,
                                                 double (&prev)[W], double& early, int lane, double* out) {
  double res[CHAIN_CH];
#pragma unroll
  for (int i = 0; i < CHAIN_CH; ++i) {
    const double m = c.o[i][W - 1] * prev[W - 1];
    __builtin_amdgcn_sched_barrier(0);
    const double* on = i + 1 < CHAIN_CH ? c.o[i + 1] : nxt.o[0];
    double pr[W > 1 ? W - 1 : 1];
#pragma unroll
    for (int t = 0; t + 1 < W; ++t) pr[t] = on[t] * prev[t + 1];   // y_{i+1-W+t} = prev[t + 1]
    __builtin_amdgcn_sched_barrier(0);
    const double acc = early - m;
    __builtin_amdgcn_sched_barrier(0);
    double en = i + 1 < CHAIN_CH ? c.r[i + 1] : nxt.r[0];
#pragma unroll
    for (int t = 0; t + 1 < W; ++t) en -= pr[t];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t + 1 < W; ++t) prev[t] = prev[t + 1];
    prev[W - 1] = acc;
    res[i] = acc;
    early = en;
  }
  if (lane == 0) {  // every lane holds all eight results; one lane stores them (padded array)
#pragma unroll
    for (int i = 0; i < CHAIN_CH; ++i) out[i] = res[i];
  }
}
template <int W>
__device__ __forceinline__ void band_chain_pass(int n, const double* ops, double* v) {
  const int lane = threadIdx.x;
  double prev[W];
#pragma unroll
  for (int t = 0; t < W; ++t) prev[t] = 0.0;
  ChainChunk<W> A, B;
  band_chain_fetch<W>(A, ops, v);
  double early = A.r[0];  // step 0: the same subtractions of (operand x 0.0) the plain loop makes
#pragma unroll
  for (int t = 0; t + 1 < W; ++t) early -= A.o[0][t] * prev[t];
  for (int s0 = 0; s0 < n; s0 += 2 * CHAIN_CH) {
    const double* op = ops + s0 * W;
    double* rv = v + s0;
    // the next chunk's operands are read BEFORE this chunk's dependent chain starts (and stay
    // there: sched_barrier), so the LDS latency hides behind it
    band_chain_fetch<W>(B, op + CHAIN_CH * W, rv + CHAIN_CH);
    __builtin_amdgcn_sched_barrier(0);
    band_chain_steps<W>(A, B, prev, early, lane, rv);
    __builtin_amdgcn_sched_barrier(0);
    band_chain_fetch<W>(A, op + 2 * CHAIN_CH * W, rv + 2 * CHAIN_CH);
    __builtin_amdgcn_sched_barrier(0);
    band_chain_steps<W>(B, A, prev, early, lane, rv + CHAIN_CH);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// global -> LDS with the loads of a round issued TOGETHER (one memory round trip per 16 x 64
// entries; written as a plain loop the compiler waited for every load before the next one:
// seventeen dependent round trips for the 511-row level, more than the substitution itself)
__device__ __forceinline__ void band_chain_stage(const double* __restrict__ src, double* dst, int valid,
                                                 int total, int lane) {
  constexpr int R = 16;
  for (int base = 0; base < total; base += 64 * R) {
    double t[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int k = base + j * 64 + lane;
      t[j] = k < valid ? src[k] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int k = base + j * 64 + lane;
      if (k < total) dst[k] = t[j];
    }
  }
}
template <int W>
__global__ __launch_bounds__(64) void band_chain_kernel(
    int n, const double* __restrict__ cf, const double* __restrict__ cb,
    const double* __restrict__ dg, const double* __restrict__ f, double* __restrict__ x, int n_h,
    const double* uh_in, double* uh_out, int pre) {
  extern __shared__ double chain_lds[];
  const int np = band_chain_pad(n);
  double* ops = chain_lds;           // np x W: forward operands
  double* v = chain_lds + np * W;    // np: right-hand side in step order, then the result
  // backward operands: staged with the forward ones when the LDS allows (pre: no global round
  // trip between the two passes), else over them after the first pass
  double* opb = pre ? v + np : ops;
  const int lane = threadIdx.x;
  double dgv[32];                    // the diagonal, mirrored, requested before the first pass
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int s = lane + 64 * q;
    dgv[q] = s < n ? dg[n - 1 - s] : 1.0;
  }
  // the fine vector the result is prolonged into: its first 1024 rows are requested now, the solve
  // hides the round trip
  double a0[8], a1[8];
  auto fetch_fine = [&](int j0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int i = 2 * (j0 + q * 64 + lane);
      a0[q] = i < n_h ? uh_in[i] : 0.0;
      a1[q] = i + 1 < n_h ? uh_in[i + 1] : 0.0;
    }
  };
  if (uh_out) fetch_fine(0);
  band_chain_stage(cf, ops, n * W, np * W, lane);
  band_chain_stage(f, v, n, np, lane);
  if (pre) band_chain_stage(cb, opb, n * W, np * W, lane);
  lds_barrier();
  band_chain_pass<W>(n, ops, v);                 // L y = f
  lds_barrier();
  double z[32];                                  // n <= 2048: z = y / D, mirrored
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int s = lane + 64 * q;
    z[q] = s < n ? v[n - 1 - s] / dgv[q] : 0.0;
  }
  if (!pre) band_chain_stage(cb, opb, n * W, n * W, lane);
  lds_barrier();
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    const int s = lane + 64 * q;
    if (s < n) v[s] = z[q];
  }
  lds_barrier();
  band_chain_pass<W>(n, opb, v);                 // L^T x = z, step s = row n-1-s
  lds_barrier();
  for (int s = lane; s < n; s += 64) x[n - 1 - s] = v[s];
  if (uh_out) {
    // the prolongation into the level above (multigrid.hpp:294-296 for l = L-2), which would
    // otherwise be a launch of its own: uh_out = uh_in + P x, linear_prolong_add2_kernel's
    // expressions and order; x sits in v[] mirrored (x_j = v[n-1-j])
    // (the loads of eight pairs issued together: one global round trip per 1024 fine rows)
    for (int j0 = 0; 2 * j0 < n_h; j0 += 64 * 8) {
      if (j0 > 0) fetch_fine(j0);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = j0 + q * 64 + lane, i = 2 * j;
        if (i >= n_h) continue;
        double t0 = 0.0, t1 = 0.0;
        const double b = (j < n) ? v[n - 1 - j] : 0.0;
        if (j >= 1 && j - 1 < n) t0 += 0.5 * v[n - j];
        if (j < n) {
          t0 += 0.5 * b;
          t1 += 1.0 * b;
        }
        uh_out[i] = a0[q] + t0;
        if (i + 1 < n_h) uh_out[i + 1] = a1[q] + t1;
      }
    }
  }
}
bool band_chain_ok(int64_t n, int64_t w) { return w >= 1 && w <= 3 && n >= 1 && n <= 2048; }
hipError_t launch_band_chain(int64_t n, int w, const double* cf, const double* cb, const double* dg,
                             const double* f, double* x, hipStream_t st, int64_t n_h, const double* uh_in,
                             double* uh_out) {
  if (!band_chain_ok(n, w) || (uh_out && (!uh_in || n_h < 1 || n_h > 2 * n + 2))) return hipErrorInvalidValue;
  const size_t np = (size_t)band_chain_pad((int)n);
  const int pre = sizeof(double) * np * (size_t)(2 * w + 1) <= (size_t)48 * 1024 ? 1 : 0;
  const size_t lds = sizeof(double) * np * (size_t)(pre ? 2 * w + 1 : w + 1);
  switch (w) {
    case 1: hipLaunchKernelGGL(band_chain_kernel<1>, dim3(1), dim3(64), lds, st, (int)n, cf, cb, dg, f, x, (int)n_h, uh_in, uh_out, pre); break;
    case 2: hipLaunchKernelGGL(band_chain_kernel<2>, dim3(1), dim3(64), lds, st, (int)n, cf, cb, dg, f, x, (int)n_h, uh_in, uh_out, pre); break;
    default: hipLaunchKernelGGL(band_chain_kernel<3>, dim3(1), dim3(64), lds, st, (int)n, cf, cb, dg, f, x, (int)n_h, uh_in, uh_out, pre); break;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------ K-Tail -----
// The deepest levels of a 2+2 true-Jacobi cycle in ONE launch of ONE workgroup (1024 threads),
// with every vector of those levels RESIDENT IN LDS: the down-legs of the levels lt .. L-2 (second
// pre-sweep, residual, restriction, first sweep of the next level), the coarsest solve
// (K-BandChain's recurrence on wave 0), the prolongation out of the coarsest level, the up-legs back
// to level lt and the prolongation into level lt - 1.  Every one of those steps is a launch of
// 5-7 us otherwise -- and almost all of that is latency: the launch itself and two or three
// dependent global round trips per step for a few thousand rows.  (A first form of this kernel
// that kept the vectors in global memory and replayed the pair kernels tile by tile inside one
// workgroup took 134 us for what eleven launches do in 70: the round trips, not the launches,
// are the cost.)  Here a step is an LDS pass and a workgroup barrier: row types, tables, f and
// the first-sweep result of level lt are staged once, everything else is produced in LDS; the
// level vectors leave for global memory once, at the end (the getters and the finer level read
// them there).  Same row arithmetic (dict_fetch / dict_rows on the same operands in the same
// order), same transfer expressions, same recurrence: same bits.  All levels run with the widest
// instantiation (two code words, nine slots); absent slots are no entries.
constexpr int TAIL_MAX_LEVELS = TAIL_LEVELS_MAX;
constexpr int TAIL_MAX_COARSE = 1024;
constexpr int TAIL_LDS_DOUBLES = 18432;   // 147 456 B
typedef TailLevelRef TailLevel;
typedef TailRef TailArgs;
int tail_max_coarse() { return TAIL_MAX_COARSE; }
__host__ __device__ inline int tail_pad(int n) { return (n + 3) & ~1; }   // even, one spare entry
// LDS layout in doubles: per level {U, F, tabJ (1024), tabR (1024), wtab (512), row types (n bytes)},
// then T and R (scratch of the largest level), Uc, the chain arrays
__host__ __device__ inline size_t tail_lds_doubles(const TailArgs& A) {
  size_t d = 0;
  int nmax = 0;
  for (int q = 0; q < A.nlev; ++q) {
    d += 2 * (size_t)tail_pad(A.L[q].n) + 1024 + 1024 + 512 + (size_t)((A.L[q].n + 15) / 16) * 2;
    nmax = A.L[q].n > nmax ? A.L[q].n : nmax;
  }
  d += 2 * (size_t)tail_pad(nmax) + (size_t)tail_pad(A.nc) + (size_t)band_chain_pad(A.nc) * (size_t)(A.wc + 1);
  return d;
}
// offsets (in doubles) into the kernel's one LDS array: pointers are formed from the array at
// every use, so the compiler knows the address space (ds_ instructions, not flat ones)
struct TailLds {
  int U, F, tabJ, tabR, wtab, rt;
};
// out[row] = the MODE operation on rows [0, n) of the level, x / f / out in LDS
template <int MODE>
__device__ __forceinline__ void tail_rows(int n, const DictEntry* tab, const uint64_t* wtab, const uint8_t* rt,
                                          const double* x, const double* f, double* out, double omega) {
  for (int row0 = (int)threadIdx.x * 2; row0 < n; row0 += 2048) {
    DictStream<2, 2> s;
    dict_fetch<MODE, 2, false, 2>(s, row0, n, nullptr, rt, f, x, 0);
    dict_expand<2, 2>(s, wtab);
    double res[2];
    dict_rows<MODE, 2, 9, 2>(s, row0, tab, x, omega, 0, res);
    if (s.live[0]) out[row0] = res[0];
    if (s.live[1]) out[row0 + 1] = res[1];
  }
}
// out[i] = base[i] + (P uH)[i] for i in [0, n_h) (linear_prolong_add2_kernel)
__device__ __forceinline__ void tail_prolong(int n_h, int n_H, const double* uH, const double* base, double* out) {
  for (int j = threadIdx.x; 2 * j < n_h; j += 1024) {
    const int i = 2 * j;
    double t0 = 0.0, t1 = 0.0;
    const double b = (j < n_H) ? uH[j] : 0.0;
    if (j >= 1 && j - 1 < n_H) t0 += 0.5 * uH[j - 1];
    if (j < n_H) {
      t0 += 0.5 * b;
      t1 += 1.0 * b;
    }
    out[i] = base[i] + t0;
    if (i + 1 < n_h) out[i + 1] = base[i + 1] + t1;
  }
}
template <int W>
__global__ __launch_bounds__(1024) void tail_kernel(TailArgs A) {
  __shared__ __attribute__((aligned(16))) double tail_lds[TAIL_LDS_DOUBLES];   // static: 144 KB of the CU's 160
  TailLds P[TAIL_MAX_LEVELS];
  int p = 0, nmax = 0;
  for (int q = 0; q < A.nlev; ++q) {
    const int n = A.L[q].n, np = tail_pad(n);
    P[q].U = p; p += np;
    P[q].F = p; p += np;
    P[q].tabJ = p; p += 1024;
    P[q].tabR = p; p += 1024;
    P[q].wtab = p; p += 512;
    P[q].rt = p; p += ((n + 15) / 16) * 2;
    nmax = n > nmax ? n : nmax;
  }
  double* const T = tail_lds + p; p += tail_pad(nmax);
  double* const R = tail_lds + p; p += tail_pad(nmax);
  const int Uc = p; p += tail_pad(A.nc);
  const int npc = band_chain_pad(A.nc);
  double* const ops = tail_lds + p;      // npc x W
  const int voff = p + npc * W;
  double* const v = tail_lds + voff;     // npc: coarsest right-hand side in step order, then the result
  if (voff + npc > TAIL_LDS_DOUBLES) return;   // launch_tail refuses such a hierarchy; never reached
  const double omega = A.omega;
  const int tid = (int)threadIdx.x;
#define TAIL_D(off) (tail_lds + (off))
#define TAIL_TAB(off) reinterpret_cast<DictEntry*>(tail_lds + (off))
#define TAIL_W(off) reinterpret_cast<uint64_t*>(tail_lds + (off))
#define TAIL_B(off) reinterpret_cast<uint8_t*>(tail_lds + (off))
  // ---- stage: tables and row types of every level, f and the first-sweep result of level lt, the
  // forward operands of the coarsest factor -- all requested before the first wait ----
  for (int q = 0; q < A.nlev; ++q) {
    const TailLevel& L = A.L[q];
    if (tid < 256) {
      const int t = tid;
      const double val = t < L.ntab ? L.dval[t] : 0.0;
      const int32_t o = t < L.ntab ? L.doff[t] : 0;
      DictEntry e;
      e.a = o == 0 ? 0.0 : val;                    // dict_stage_table<CSR_JACOBI>
      e.d = (o == 0 && t < L.ntab) ? val : 0.0;
      e.off8 = o * 8;
      e.pad[0] = e.pad[1] = e.pad[2] = 0;
      TAIL_TAB(P[q].tabJ)[t] = e;
      e.a = val;                                   // dict_stage_table<CSR_RESID>
      e.d = 0.0;
      TAIL_TAB(P[q].tabR)[t] = e;
      TAIL_W(P[q].wtab)[2 * t] = L.rwords[(size_t)t * L.words];
      TAIL_W(P[q].wtab)[2 * t + 1] = L.words == 2 ? L.rwords[(size_t)t * 2 + 1] : ~(uint64_t)0;
    }
    for (int r = tid; r < ((L.n + 15) / 16) * 16; r += 1024) TAIL_B(P[q].rt)[r] = r < L.n ? L.rtype[r] : (uint8_t)255;
  }
  for (int r = tid; r < tail_pad(A.L[0].n); r += 1024) {
    TAIL_D(P[0].F)[r] = r < A.L[0].n ? A.L[0].f[r] : 0.0;
    T[r] = r < A.L[0].n ? A.L[0].tmp[r] : 0.0;
  }
  for (int k = tid; k < npc * W; k += 1024) ops[k] = k < A.nc * W ? A.cf[k] : 0.0;
  __syncthreads();
  // ---- down-legs (multigrid.hpp:265-283) ----
  for (int li = 0; li < A.nlev; ++li) {
    const int n = A.L[li].n;
    const bool last = li + 1 == A.nlev;
    const int nH = last ? A.nc : A.L[li + 1].n;
    const double* dH = last ? A.diagc : A.L[li + 1].diag;
    const TailLds Q = P[li];
    tail_rows<CSR_JACOBI>(n, TAIL_TAB(Q.tabJ), TAIL_W(Q.wtab), TAIL_B(Q.rt), T, TAIL_D(Q.F), TAIL_D(Q.U),
                          omega);                                          // second pre-sweep (T = the first)
    __syncthreads();
    tail_rows<CSR_RESID>(n, TAIL_TAB(Q.tabR), TAIL_W(Q.wtab), TAIL_B(Q.rt), TAIL_D(Q.U), TAIL_D(Q.F), R,
                         omega);                                           // r = f - A u
    if (tid < 2) R[n + tid] = 0.0;
    __syncthreads();
    double* FH = TAIL_D(last ? voff : P[li + 1 < A.nlev ? li + 1 : li].F);
    for (int j = tid; j < nH; j += 1024) {                                // linear_restrict_kernel + jacobi_from_zero
      const int i = 2 * j;
      double sum = 0.0;
      if (i < n) sum += 0.5 * R[i];
      if (i + 1 < n) sum += 1.0 * R[i + 1];
      if (i + 2 < n) sum += 0.5 * R[i + 2];
      FH[j] = sum;
      const double xi = 0.0, acc = 0.0;
      const double d = dH[j];
      T[j] = (d == 0.0) ? xi : xi + omega * ((sum - acc) / d - xi);
    }
    if (last)
      for (int j = nH + tid; j < npc; j += 1024) v[j] = 0.0;
    __syncthreads();
  }
  // ---- coarsest solve (multigrid.hpp:287-288): band_chain_kernel, wave 0 walks the recurrence ----
  {
    const int n = A.nc;
    if (tid >= 64)   // the level's right-hand side and first sweep go home while wave 0 solves
      for (int j = tid - 64; j < n; j += 960) {
        A.fc[j] = v[j];
        A.tmpc[j] = T[j];
      }
    __syncthreads();
    if (tid < 64) band_chain_pass<W>(n, ops, v);       // L y = f
    __syncthreads();
    double z = 0.0;
    if (tid < n) z = v[n - 1 - tid] / A.dg[n - 1 - tid];
    for (int k = tid; k < n * W; k += 1024) ops[k] = A.cb[k];
    __syncthreads();
    if (tid < n) v[tid] = z;
    __syncthreads();
    if (tid < 64) band_chain_pass<W>(n, ops, v);       // L^T x = z, step s = row n-1-s
    __syncthreads();
    if (tid < n) {
      const double x = v[tid];
      TAIL_D(Uc)[n - 1 - tid] = x;
      A.uc[n - 1 - tid] = x;
    }
    __syncthreads();
  }
  // ---- up-legs (multigrid.hpp:291-302) ----
  int uH = Uc;
  int nH = A.nc;
  for (int li = A.nlev - 1; li >= 0; --li) {
    const int n = A.L[li].n;
    const TailLds Q = P[li];
    tail_prolong(n, nH, TAIL_D(uH), TAIL_D(Q.U), T);                       // T = u + P u_H
    __syncthreads();
    tail_rows<CSR_JACOBI>(n, TAIL_TAB(Q.tabJ), TAIL_W(Q.wtab), TAIL_B(Q.rt), T, TAIL_D(Q.F), R, omega);
    __syncthreads();
    tail_rows<CSR_JACOBI>(n, TAIL_TAB(Q.tabJ), TAIL_W(Q.wtab), TAIL_B(Q.rt), R, TAIL_D(Q.F), TAIL_D(Q.U), omega);
    __syncthreads();
    uH = Q.U;
    nH = n;
  }
  // ---- the prolongation into level lt - 1, and the level vectors go home ----
  tail_prolong(A.n_fine, nH, TAIL_D(uH), A.uf_in, A.uf_out);
  for (int li = 0; li < A.nlev; ++li) {
    const TailLevel& L = A.L[li];
    for (int r = tid; r < L.n; r += 1024) {
      L.u[r] = TAIL_D(P[li].U)[r];
      if (li > 0) const_cast<double*>(L.f)[r] = TAIL_D(P[li].F)[r];
    }
  }
#undef TAIL_D
#undef TAIL_TAB
#undef TAIL_W
#undef TAIL_B
}
bool tail_level_ok(int64_t n, const DictRef& D, int hb) {
  return D.rtype && D.rwords && !D.nt && hb >= 0 && n >= 256 && n <= 4095 && D.wmax <= 9 && D.words >= 1 &&
         D.words <= 2 && D.ntab <= 255;
}
size_t tail_lds_bytes(const TailRef& A) { return sizeof(double) * tail_lds_doubles(A); }
size_t tail_lds_capacity() { return sizeof(double) * (size_t)TAIL_LDS_DOUBLES; }
hipError_t launch_tail(const TailArgs& A, hipStream_t st) {
  if (A.nlev < 1 || A.nlev > TAIL_MAX_LEVELS || A.nc < 1 || A.nc > TAIL_MAX_COARSE || !band_chain_ok(A.nc, A.wc) ||
      tail_lds_bytes(A) > tail_lds_capacity() || A.n_fine < 1 || !A.uf_in || !A.uf_out)
    return hipErrorInvalidValue;
  for (int q = 0; q < A.nlev; ++q)
    if (A.L[q].n < 2 || !A.L[q].rtype || !A.L[q].rwords || !A.L[q].doff || !A.L[q].dval || !A.L[q].f || !A.L[q].u ||
        !A.L[q].tmp || !A.L[q].diag || A.L[q].ntab < 1 || A.L[q].ntab > 255)
      return hipErrorInvalidValue;
  switch (A.wc) {
    case 1: hipLaunchKernelGGL(tail_kernel<1>, dim3(1), dim3(1024), 0, st, A); break;
    case 2: hipLaunchKernelGGL(tail_kernel<2>, dim3(1), dim3(1024), 0, st, A); break;
    default: hipLaunchKernelGGL(tail_kernel<3>, dim3(1), dim3(1024), 0, st, A); break;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------- K-BandWide -----
// The same LDL^T substitution for ANY half-bandwidth w (the reference's SimplicialLDLT
// factors whatever coarsest matrix it is given, multigrid.hpp:240-243): rows in blocks of
// 64, one wave, lane l owns row r0 + l of the current block.
//   phase 1: the terms that come from earlier blocks, k = i-w .. r0-1 in ascending k --
//            a private loop per lane over the block's [t][lane] operand panel (coalesced
//            512-B loads), y_k from an LDS ring that holds the last w + 64 results;
//   phase 2: the 64 x 64 triangle of the block itself, step s broadcasts the finished
//            y_{r0+s} with v_readlane (ascending k again).
// Every row therefore subtracts its products in exactly the order of the row-oriented
// substitution (oracle band_solve), so the result has the same bits.  The backward pass
// runs the same code on the mirrored system (row n-1-i, host_setup.cpp: band_wide_schedule).
template <bool FWD>
__device__ __forceinline__ void band_wide_pass(int64_t n, int w, int ringmask, double* ring,
                                               const double* __restrict__ sched,
                                               const double* rhs, double* out) {
  const int lane = threadIdx.x;
  const int64_t nb = (n + 63) / 64;
  const int64_t stride = (int64_t)(w + 64) * 64;
  for (int k = lane; k <= ringmask; k += 64) ring[k] = 0.0;
  __syncthreads();
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t r0 = b * 64, i = r0 + lane;
    const bool live = i < n;
    double acc = live ? rhs[FWD ? i : n - 1 - i] : 0.0;
    const double* __restrict__ Lp = sched + b * stride + lane;
    const int64_t tl = (int64_t)w - i;
    const int tlo = tl > 0 ? (int)tl : 0, thi = w - lane;  // k >= 0 and k < r0
    if (b > 0) {
#pragma unroll 4
      for (int t = 0; t < w; ++t) {
        const double v = Lp[(int64_t)t * 64];
        const double yk = ring[(int)((i - w + t) & ringmask)];
        const double p = v * yk;
        acc = (t >= tlo && t < thi) ? acc - p : acc;
      }
    }
    const double* __restrict__ Ld = Lp + (int64_t)w * 64;
#pragma unroll 16
    for (int s = 0; s < 64; ++s) {
      const double vk = readlane_f64(acc, s);
      const double p = Ld[s * 64] * vk;
      acc = (lane > s) ? acc - p : acc;
    }
    if (live) {
      ring[(int)(i & ringmask)] = acc;
      out[FWD ? i : n - 1 - i] = acc;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(64) void band_wide_kernel(
    int64_t n, int w, int ringmask, const double* __restrict__ sched_f,
    const double* __restrict__ sched_b, const double* __restrict__ dg, const double* f, double* y,
    double* x) {
  extern __shared__ double ring[];
  band_wide_pass<true>(n, w, ringmask, ring, sched_f, f, y);   // L y = f
  __threadfence_block();
  for (int64_t i = threadIdx.x; i < n; i += 64) y[i] = y[i] / dg[i];  // z = y / D
  __threadfence_block();
  __syncthreads();
  band_wide_pass<false>(n, w, ringmask, ring, sched_b, y, x);  // L^T x = z
}
hipError_t launch_band_wide(int64_t n, int64_t w, const double* sched_f, const double* sched_b,
                            const double* dg, const double* f, double* y, double* x,
                            hipStream_t st) {
  if (n <= 0) return hipSuccess;
  int64_t R = 128;
  while (R < w + 64) R *= 2;
  if (R * 8 > 65536 || w < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(band_wide_kernel, dim3(1), dim3(64), (size_t)R * 8, st, n, (int)w, (int)(R - 1),
                     sched_f, sched_b, dg, f, y, x);
  return hipGetLastError();
}

// ---------------------------------------------------------------- K-Spike -----
// Parallel form of the same LDL^T solve (host_setup.cpp: spike_factor): P partitions
// of c rows.  (A) every partition runs the one-wave substitution above on its own
// triangular block, one workgroup per partition; (B) one wave walks the P partition
// boundaries (w unknowns each: t_p = tail(g_p) - Vt_p t_{p-1}); (C) every row
// subtracts its spike row times the neighbouring boundary vector and divides by D;
// (D)-(F) mirror it for L^T.  Depth ~ 2c + P steps instead of n.  Same direct solve,
// different rounding order, so the V-cycle uses it only on request.
template <int M, bool FWD>
__global__ __launch_bounds__(64) void spike_local_kernel(SpikeArgs a, const double* rhs,
                                                         double* out) {
  const int lane = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * a.c;
  const int64_t nl = (a.n - r0 < a.c) ? a.n - r0 : a.c;
  band_pass<M, FWD>(nl, lane, (FWD ? a.sched_f : a.sched_b) + (int64_t)blockIdx.x * a.sched_stride,
                    rhs + r0, out + r0);
}
// forward: q = 0..P-1, block = last w rows of partition q, couples to q-1
// backward: q = P-1..0, block = first w rows of partition q, couples to q+1
// Vt/Wh blocks are zero-padded to M x M ([j][k], k fastest), so the matvec is a
// fixed-length unrolled loop and the next block is fetched while this one is used.
template <int M, bool FWD>
__global__ __launch_bounds__(64) void spike_boundary_kernel(SpikeArgs a, const double* g,
                                                            double* bnd) {
  const int k = threadIdx.x;
  const int w = a.w, wq = a.w > 0 ? a.w : 1;
  const bool act = k < w;
  const int kc = k < M ? k : 0;
  const double* blocks = FWD ? a.Vt : a.Wh;
  double cur[M], nxt[M];
#pragma unroll
  for (int j = 0; j < M; ++j) cur[j] = 0.0;
  double t = 0.0;
  for (int s = 0; s < a.P; ++s) {
    const int q = FWD ? s : a.P - 1 - s;
    const int64_t r0 = (int64_t)q * a.c;
    const int64_t nl = (a.n - r0 < a.c) ? a.n - r0 : a.c;
    const int64_t row = FWD ? r0 + nl - w + k : r0 + k;
    const bool ok = act && row >= r0 && row < r0 + nl;
    double v = g[ok ? row : 0];
    v = ok ? v : 0.0;
    if (s + 1 < a.P) {
      const double* blk = blocks + (int64_t)(FWD ? q + 1 : q - 1) * M * M + kc;
#pragma unroll
      for (int j = 0; j < M; ++j) nxt[j] = blk[j * M];
    }
#pragma unroll
    for (int j = 0; j < M; ++j) v -= cur[j] * readlane_f64(t, j);
    v = ok ? v : 0.0;
    if (act) bnd[(int64_t)q * wq + k] = v;
    t = v;
#pragma unroll
    for (int j = 0; j < M; ++j) cur[j] = nxt[j];
  }
}
// forward: out_i = (g_i - V_i . T[p-1]) / D_i ; backward: out_i = g_i - W_i . H[p+1]
template <bool FWD>
__global__ __launch_bounds__(256) void spike_correct_kernel(SpikeArgs a, const double* g,
                                                            const double* bnd, double* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int w = a.w, wq = a.w > 0 ? a.w : 1;
  const int64_t p = i / a.c;
  double v = g[i];
  const int64_t nb = FWD ? p - 1 : p + 1;
  if (nb >= 0 && nb < a.P) {
    const double* __restrict__ sp = (FWD ? a.V : a.W) + i * wq;
    const double* __restrict__ t = bnd + nb * wq;
#pragma unroll 8
    for (int j = 0; j < w; ++j) v -= sp[j] * t[j];
  }
  out[i] = FWD ? v / a.d[i] : v;
}
template <int M>
static hipError_t launch_spike_m(const SpikeArgs& a, hipStream_t st) {
  const unsigned P = (unsigned)a.P, g = (unsigned)((a.n + 255) / 256);
  hipLaunchKernelGGL((spike_local_kernel<M, true>), dim3(P), dim3(64), 0, st, a, a.f, a.G);
  hipLaunchKernelGGL((spike_boundary_kernel<M, true>), dim3(1), dim3(64), 0, st, a, a.G, a.T);
  hipLaunchKernelGGL((spike_correct_kernel<true>), dim3(g), dim3(256), 0, st, a, a.G, a.T, a.Z);
  hipLaunchKernelGGL((spike_local_kernel<M, false>), dim3(P), dim3(64), 0, st, a, a.Z, a.G);
  hipLaunchKernelGGL((spike_boundary_kernel<M, false>), dim3(1), dim3(64), 0, st, a, a.G, a.H);
  hipLaunchKernelGGL((spike_correct_kernel<false>), dim3(g), dim3(256), 0, st, a, a.G, a.H, a.x);
  return hipGetLastError();
}
hipError_t launch_spike_solve(const SpikeArgs& a, hipStream_t st) {
  if (a.n <= 0) return hipSuccess;
  switch (a.m) {
    case 4: return launch_spike_m<4>(a, st);
    case 8: return launch_spike_m<8>(a, st);
    case 16: return launch_spike_m<16>(a, st);
    case 32: return launch_spike_m<32>(a, st);
    case 64: return launch_spike_m<64>(a, st);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_band_solve(int64_t n, int m, const double* sched_f, const double* sched_b,
                             const double* dg, const double* f, double* y, double* x,
                             hipStream_t st) {
  if (n <= 0) return hipSuccess;
  switch (m) {
    case 4: hipLaunchKernelGGL(band_solve_kernel<4>, dim3(1), dim3(64), 0, st, n, sched_f, sched_b, dg, f, y, x); break;
    case 8: hipLaunchKernelGGL(band_solve_kernel<8>, dim3(1), dim3(64), 0, st, n, sched_f, sched_b, dg, f, y, x); break;
    case 16: hipLaunchKernelGGL(band_solve_kernel<16>, dim3(1), dim3(64), 0, st, n, sched_f, sched_b, dg, f, y, x); break;
    case 32: hipLaunchKernelGGL(band_solve_kernel<32>, dim3(1), dim3(64), 0, st, n, sched_f, sched_b, dg, f, y, x); break;
    case 64: hipLaunchKernelGGL(band_solve_kernel<64>, dim3(1), dim3(64), 0, st, n, sched_f, sched_b, dg, f, y, x); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------ K-Galerkin -----
// Setup on the device: A_H = R (A P) for the LinearInterpolator pair (P column j =
// {0.5, 1, 0.5} on rows 2j..2j+2, R = P^T; multigrid.hpp:219-223), in Eigen's
// conservative-product order: an output entry exists as soon as one product touches
// it (exact zeros are kept), its first product is assigned and the others are added in
// ascending inner index -- the same bits as host_setup.cpp: spgemm_csr.  Thread per
// output row, two passes (count, fill) around an exclusive scan.
//
// Row i of A P.  Walking A's row in ascending k, the coarse columns that P's row k
// holds (k odd: (k-1)/2; k even: k/2 - 1 and k/2) come out in non-decreasing order, so
// one open accumulator is enough.
template <bool FILL>
__global__ __launch_bounds__(256) void galerkin_ap_kernel(
    int64_t n_h, int64_t n_H, const int32_t* __restrict__ arp, const int32_t* __restrict__ acol,
    const double* __restrict__ aval, int32_t* __restrict__ cnt, const int32_t* __restrict__ orp,
    int32_t* __restrict__ ocol, double* __restrict__ oval) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_h) return;
  int32_t n_out = 0;
  int64_t cur = -1;
  double acc = 0.0;
  const int64_t base = FILL ? orp[i] : 0;
  auto touch = [&](int64_t c, double t) {
    if (c == cur) {
      acc += t;
    } else {
      if (cur >= 0) {
        if (FILL) { ocol[base + n_out] = (int32_t)cur; oval[base + n_out] = acc; }
        ++n_out;
      }
      cur = c;
      acc = t;  // first touch: assign
    }
  };
  for (int32_t p = arp[i]; p < arp[i + 1]; ++p) {
    const int64_t k = acol[p];
    const double a = aval[p];
    if (k & 1) {
      const int64_t c = (k - 1) >> 1;
      if (c < n_H) touch(c, a * 1.0);
    } else {
      const int64_t c = k >> 1;
      if (c >= 1 && c - 1 < n_H) touch(c - 1, a * 0.5);  // P(2(c-1)+2, c-1)
      if (c < n_H) touch(c, a * 0.5);                    // P(2c, c)
    }
  }
  if (cur >= 0) {
    if (FILL) { ocol[base + n_out] = (int32_t)cur; oval[base + n_out] = acc; }
    ++n_out;
  }
  if (!FILL) cnt[i] = n_out;
}
// Row j of R (A P): three-way merge of rows 2j, 2j+1, 2j+2 of A P, products taken in
// that order (R's row j in ascending column).
template <bool FILL>
__global__ __launch_bounds__(256) void galerkin_rap_kernel(
    int64_t n_h, int64_t n_H, const int32_t* __restrict__ prp, const int32_t* __restrict__ pcol,
    const double* __restrict__ pval, int32_t* __restrict__ cnt, const int32_t* __restrict__ orp,
    int32_t* __restrict__ ocol, double* __restrict__ oval) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n_H) return;
  const double w[3] = {0.5, 1.0, 0.5};
  int32_t q[3], e[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int64_t i = 2 * j + r;
    q[r] = i < n_h ? prp[i] : 0;
    e[r] = i < n_h ? prp[i + 1] : 0;
  }
  int32_t n_out = 0;
  const int64_t base = FILL ? orp[j] : 0;
  for (;;) {
    int32_t c = INT32_MAX;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (q[r] < e[r]) { const int32_t cc = pcol[q[r]]; c = cc < c ? cc : c; }
    if (c == INT32_MAX) break;
    double v = 0.0;
    bool started = false;
#pragma unroll
    for (int r = 0; r < 3; ++r)
      if (q[r] < e[r] && pcol[q[r]] == c) {
        const double t = w[r] * pval[q[r]];
        v = started ? v + t : t;
        started = true;
        ++q[r];
      }
    if (FILL) { ocol[base + n_out] = c; oval[base + n_out] = v; }
    ++n_out;
  }
  if (!FILL) cnt[j] = n_out;
}
// General C = A B on CSR arrays (custom interpolators: A P and R (A P), multigrid.hpp:219-223)
// in Eigen's conservative-product order: row i of C is the K-way merge of the rows B[k, :] for
// the K entries k of A's row i; an output entry exists as soon as one product touches it, its
// first product is assigned and the later ones are added in ascending k (= the order of A's
// row) -- the same bits as host_setup.cpp: spgemm_csr.  Thread per output row, K <= 32 cursors
// (rows of A longer than that raise *overflow and the host product is used instead).
constexpr int SPGEMM_K = 32;
template <bool FILL>
__global__ __launch_bounds__(128) void spgemm_kway_kernel(
    int64_t n_rows, const int32_t* __restrict__ arp, const int32_t* __restrict__ acol,
    const double* __restrict__ aval, const int32_t* __restrict__ brp, const int32_t* __restrict__ bcol,
    const double* __restrict__ bval, int32_t* __restrict__ cnt, const int32_t* __restrict__ orp,
    int32_t* __restrict__ ocol, double* __restrict__ oval, int32_t* __restrict__ overflow) {
  const int64_t i = (int64_t)blockIdx.x * 128 + threadIdx.x;
  if (i >= n_rows) return;
  const int32_t a0 = arp[i], K = arp[i + 1] - a0;
  if (K > SPGEMM_K) {
    if (!FILL) cnt[i] = 0;
    atomicOr(overflow, 1);
    return;
  }
  int32_t q[SPGEMM_K], e[SPGEMM_K];
  for (int t = 0; t < K; ++t) {
    const int32_t k = acol[a0 + t];
    q[t] = brp[k];
    e[t] = brp[k + 1];
  }
  int32_t n_out = 0;
  const int64_t base = FILL ? orp[i] : 0;
  for (;;) {
    int32_t c = INT32_MAX;
    for (int t = 0; t < K; ++t)
      if (q[t] < e[t]) {
        const int32_t cc = bcol[q[t]];
        c = cc < c ? cc : c;
      }
    if (c == INT32_MAX) break;
    double v = 0.0;
    bool started = false;
    for (int t = 0; t < K; ++t)
      if (q[t] < e[t] && bcol[q[t]] == c) {
        if (FILL) {
          const double p = aval[a0 + t] * bval[q[t]];
          v = started ? v + p : p;
          started = true;
        }
        ++q[t];
      }
    if (FILL) {
      ocol[base + n_out] = c;
      oval[base + n_out] = v;
    }
    ++n_out;
  }
  if (!FILL) cnt[i] = n_out;
}
hipError_t launch_spgemm(bool fill, int64_t n_rows, const int32_t* arp, const int32_t* acol,
                         const double* aval, const int32_t* brp, const int32_t* bcol, const double* bval,
                         int32_t* cnt, const int32_t* orp, int32_t* ocol, double* oval,
                         int32_t* overflow, hipStream_t st) {
  const unsigned grid = (unsigned)((n_rows + 127) / 128);
  if (fill)
    hipLaunchKernelGGL(spgemm_kway_kernel<true>, dim3(grid), dim3(128), 0, st, n_rows, arp, acol, aval, brp,
                       bcol, bval, cnt, orp, ocol, oval, overflow);
  else
    hipLaunchKernelGGL(spgemm_kway_kernel<false>, dim3(grid), dim3(128), 0, st, n_rows, arp, acol, aval,
                       brp, bcol, bval, cnt, orp, ocol, oval, overflow);
  return hipGetLastError();
}
// exclusive scan of n int32 counts into n + 1 offsets (three small kernels)
__global__ __launch_bounds__(256) void scan_block_sums_kernel(int64_t n, const int32_t* __restrict__ in,
                                                              int64_t* __restrict__ bsum) {
  __shared__ int64_t red[256];
  const int64_t b0 = (int64_t)blockIdx.x * 1024;
  int64_t sum = 0;
  for (int k = 0; k < 4; ++k) {
    const int64_t i = b0 + threadIdx.x * 4 + k;
    if (i < n) sum += in[i];
  }
  red[threadIdx.x] = sum;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = red[0];
}
// exclusive scan of the <= 2^21 block sums by ONE workgroup: thread t owns a contiguous
// run of ceil(nb / 1024) sums, the 1024 run totals are scanned in LDS
__global__ __launch_bounds__(1024) void scan_block_offsets_kernel(int64_t nb, int64_t* bsum,
                                                                  int64_t* total) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t per = (nb + 1023) / 1024;
  const int64_t b0 = (int64_t)t * per;
  const int64_t b1 = b0 + per < nb ? b0 + per : nb;
  int64_t s = 0;
  for (int64_t b = b0; b < b1; ++b) s += bsum[b];
  part[t] = s;
  __syncthreads();
  for (int st = 1; st < 1024; st <<= 1) {
    const int64_t add = t >= st ? part[t - st] : 0;
    __syncthreads();
    part[t] += add;
    __syncthreads();
  }
  int64_t run = part[t] - s;
  for (int64_t b = b0; b < b1; ++b) {
    const int64_t v = bsum[b];
    bsum[b] = run;
    run += v;
  }
  if (t == 1023) *total = part[1023];
}
__global__ __launch_bounds__(256) void scan_finish_kernel(int64_t n, const int32_t* __restrict__ in,
                                                          const int64_t* __restrict__ bsum,
                                                          int32_t* __restrict__ out) {
  __shared__ int64_t pre[256];
  const int64_t b0 = (int64_t)blockIdx.x * 1024;
  int32_t v[4];
  int64_t sum = 0;
  for (int k = 0; k < 4; ++k) {
    const int64_t i = b0 + threadIdx.x * 4 + k;
    v[k] = i < n ? in[i] : 0;
    sum += v[k];
  }
  pre[threadIdx.x] = sum;
  __syncthreads();
  for (int st = 1; st < 256; st <<= 1) {  // inclusive scan of the per-thread sums
    const int64_t add = (int)threadIdx.x >= st ? pre[threadIdx.x - st] : 0;
    __syncthreads();
    pre[threadIdx.x] += add;
    __syncthreads();
  }
  int64_t run = bsum[blockIdx.x] + pre[threadIdx.x] - sum;
  for (int k = 0; k < 4; ++k) {
    const int64_t i = b0 + threadIdx.x * 4 + k;
    if (i < n) out[i] = (int32_t)run;
    run += v[k];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = (int32_t)run;
}
// counts[0..n) -> offsets[0..n]; *total (device int64) = offsets[n]; bsum: ceil(n/1024) int64
hipError_t launch_exclusive_scan(int64_t n, const int32_t* counts, int32_t* offsets, int64_t* bsum,
                                 int64_t* total, hipStream_t st) {
  if (n <= 0) return hipErrorInvalidValue;
  const int64_t nb = (n + 1023) / 1024;
  hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, st, n, counts, bsum);
  hipLaunchKernelGGL(scan_block_offsets_kernel, dim3(1), dim3(1024), 0, st, nb, bsum, total);
  hipLaunchKernelGGL(scan_finish_kernel, dim3((unsigned)nb), dim3(256), 0, st, n, counts, bsum,
                     offsets);
  return hipGetLastError();
}
hipError_t launch_galerkin_ap(bool fill, int64_t n_h, int64_t n_H, const int32_t* arp,
                              const int32_t* acol, const double* aval, int32_t* cnt,
                              const int32_t* orp, int32_t* ocol, double* oval, hipStream_t st) {
  const unsigned grid = (unsigned)((n_h + 255) / 256);
  if (fill)
    hipLaunchKernelGGL(galerkin_ap_kernel<true>, dim3(grid), dim3(256), 0, st, n_h, n_H, arp, acol,
                       aval, cnt, orp, ocol, oval);
  else
    hipLaunchKernelGGL(galerkin_ap_kernel<false>, dim3(grid), dim3(256), 0, st, n_h, n_H, arp, acol,
                       aval, cnt, orp, ocol, oval);
  return hipGetLastError();
}
hipError_t launch_galerkin_rap(bool fill, int64_t n_h, int64_t n_H, const int32_t* prp,
                               const int32_t* pcol, const double* pval, int32_t* cnt,
                               const int32_t* orp, int32_t* ocol, double* oval, hipStream_t st) {
  const unsigned grid = (unsigned)((n_H + 255) / 256);
  if (fill)
    hipLaunchKernelGGL(galerkin_rap_kernel<true>, dim3(grid), dim3(256), 0, st, n_h, n_H, prp, pcol,
                       pval, cnt, orp, ocol, oval);
  else
    hipLaunchKernelGGL(galerkin_rap_kernel<false>, dim3(grid), dim3(256), 0, st, n_h, n_H, prp, pcol,
                       pval, cnt, orp, ocol, oval);
  return hipGetLastError();
}

// --------------------------------------------------------------- K-Setup ------
// Setup on the device end to end: the Grid generators
// (grid.hpp:88-98 and the 7-point analogue) as kernels, and the dictionary encoder of the
// level matrices, so that a Poisson hierarchy never exists as host arrays unless a getter
// asks for one.
// A = sum over axes of I (x) .. D .. (x) I, D = tridiag(1, -2, 1) / h^2: row c holds its lower
// neighbours (axes dim-1 .. 0), the diagonal, its upper neighbours (axes 0 .. dim-1) --
// ascending columns, the order host_setup.cpp: laplacian() produces.
// n_last: units (grid lines in 2-D, x-y planes in 3-D) of the slowest axis -- n for the whole
// problem, fewer for the window of a sharded solver (host_setup.cpp: laplacian).
__global__ __launch_bounds__(256) void lap_count_kernel(int dim, int64_t n, int64_t n_last, int64_t N,
                                                        int32_t* __restrict__ cnt) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  const int64_t co[3] = {c % n, dim == 2 ? c / n : (c / n) % n, c / (n * n)};
  const int64_t ext[3] = {n, dim == 2 ? n_last : n, n_last};
  int k = 1;
  for (int a = 0; a < dim; ++a) k += (co[a] > 0) + (co[a] + 1 < ext[a]);
  cnt[c] = k;
}
__global__ __launch_bounds__(256) void lap_fill_kernel(int dim, int64_t n, int64_t n_last, int64_t N,
                                                       const int32_t* __restrict__ rowptr,
                                                       int32_t* __restrict__ col,
                                                       double* __restrict__ val, double off,
                                                       double diag) {
  const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  const int64_t co[3] = {c % n, dim == 2 ? c / n : (c / n) % n, c / (n * n)};
  const int64_t ext[3] = {n, dim == 2 ? n_last : n, n_last};
  const int64_t st[3] = {1, n, n * n};
  int64_t p = rowptr[c];
  for (int a = dim - 1; a >= 0; --a)
    if (co[a] > 0) { col[p] = (int32_t)(c - st[a]); val[p] = off; ++p; }
  col[p] = (int32_t)c; val[p] = diag; ++p;
  for (int a = 0; a < dim; ++a)
    if (co[a] + 1 < ext[a]) { col[p] = (int32_t)(c + st[a]); val[p] = off; ++p; }
}
hipError_t launch_laplacian_count(int dim, int64_t n, int64_t n_last, int64_t N, int32_t* cnt, hipStream_t st) {
  hipLaunchKernelGGL(lap_count_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dim, n, n_last, N, cnt);
  return hipGetLastError();
}
hipError_t launch_laplacian_fill(int dim, int64_t n, int64_t n_last, int64_t N, const int32_t* rowptr, int32_t* col,
                                 double* val, double off, double diag, hipStream_t st) {
  hipLaunchKernelGGL(lap_fill_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dim, n, n_last, N,
                     rowptr, col, val, off, diag);
  return hipGetLastError();
}

// stats[0] = longest row (entries that are kept), stats[1] = 1 when the matrix is not
// bitwise symmetric, diag[i] = a_ii (0.0 when absent).  prune: exact zeros do not count.
__global__ __launch_bounds__(256) void csr_inspect_kernel(int64_t n, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col,
                                                          const double* __restrict__ val, int prune,
                                                          int32_t* __restrict__ stats,
                                                          double* __restrict__ diag) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int len = 0;
  double d = 0.0;
  bool asym = false;
  for (int32_t p = rowptr[i]; p < rowptr[i + 1]; ++p) {
    const int32_t j = col[p];
    const double v = val[p];
    if (!prune || v != 0.0) ++len;
    if (j == i) { d = v; continue; }
    // (j, i) must hold the same bits: binary search in row j
    int32_t lo = rowptr[j], hi = rowptr[j + 1] - 1;
    bool found = false;
    while (lo <= hi) {
      const int32_t mid = (lo + hi) >> 1;
      const int32_t cm = col[mid];
      if (cm == (int32_t)i) {
        found = __double_as_longlong(val[mid]) == __double_as_longlong(v);
        break;
      }
      if (cm < (int32_t)i) lo = mid + 1;
      else hi = mid - 1;
    }
    asym = asym || !found;
  }
  if (diag) diag[i] = d;
  atomicMax(&stats[0], len);
  if (asym) atomicOr(&stats[1], 1);
}
hipError_t launch_csr_inspect(int64_t n, const int32_t* rowptr, const int32_t* col, const double* val,
                              bool prune, int32_t* stats, double* diag, hipStream_t st) {
  hipLaunchKernelGGL(csr_inspect_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, rowptr,
                     col, val, prune ? 1 : 0, stats, diag);
  return hipGetLastError();
}

// Dictionary encoder: row r -> `words` 64-bit words of byte codes into the pair table
// (doff, dval: <= 255 entries in LDS), entries in ascending column order, exact zeros skipped
// when prune.  A pair that is not in the table makes the row a FAILURE: its index goes to
// fail[1 + slot] (first 62 of them, fail[0] counts all) and the host, which proposed the
// table from a sample of rows, adds the row's pairs and runs the pass again -- the encoding is
// exact because every row is checked, however the table was guessed.
__global__ __launch_bounds__(256) void dict_encode_kernel(
    int64_t n, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const double* __restrict__ val, int prune, const int32_t* __restrict__ doff,
    const double* __restrict__ dval, int ntab, int words, uint64_t* __restrict__ codes,
    int32_t* __restrict__ fail) {
  __shared__ int32_t toff[256];
  __shared__ long long tval[256];
  if ((int)threadIdx.x < ntab) {
    toff[threadIdx.x] = doff[threadIdx.x];
    tval[threadIdx.x] = __double_as_longlong(dval[threadIdx.x]);
  }
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  uint64_t w[2] = {~(uint64_t)0, ~(uint64_t)0};
  int k = 0;
  bool bad = false;
  for (int32_t p = rowptr[r]; p < rowptr[r + 1]; ++p) {
    const double v = val[p];
    if (prune && v == 0.0) continue;
    const int32_t o = (int32_t)((int64_t)col[p] - r);
    const long long vb = __double_as_longlong(v);
    int code = -1;
    for (int t = 0; t < ntab; ++t)
      if (toff[t] == o && tval[t] == vb) { code = t; break; }
    if (code < 0 || k >= 8 * words) { bad = true; break; }
    w[k >> 3] = (w[k >> 3] & ~((uint64_t)0xFF << (8 * (k & 7)))) | ((uint64_t)code << (8 * (k & 7)));
    ++k;
  }
  if (bad) {
    const int32_t slot = atomicAdd(&fail[0], 1);
    if (slot < 62) fail[1 + slot] = (int32_t)r;
    return;
  }
  for (int q = 0; q < words; ++q) codes[r * words + q] = w[q];
}
// second level: code words -> one byte per row into the word table (<= 255 types; a row
// without entries is type 255); unknown words are failures as above
__global__ __launch_bounds__(256) void dict_type_kernel(int64_t n, const uint64_t* __restrict__ codes,
                                                        int words, const uint64_t* __restrict__ rwords,
                                                        int ntypes, uint8_t* __restrict__ rtype,
                                                        int32_t* __restrict__ fail) {
  __shared__ uint64_t tw[512];
  for (int t = threadIdx.x; t < ntypes * words; t += 256) tw[t] = rwords[t];
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint64_t a = codes[r * words], b = words == 2 ? codes[r * 2 + 1] : ~(uint64_t)0;
  if (a == ~(uint64_t)0 && b == ~(uint64_t)0) { rtype[r] = 255; return; }
  int ty = -1;
  for (int t = 0; t < ntypes; ++t)
    if (tw[t * words] == a && (words == 1 || tw[t * 2 + 1] == b)) { ty = t; break; }
  if (ty < 0) {
    const int32_t slot = atomicAdd(&fail[0], 1);
    if (slot < 62) fail[1 + slot] = (int32_t)r;
    return;
  }
  rtype[r] = (uint8_t)ty;
}
hipError_t launch_dict_encode(int64_t n, const int32_t* rowptr, const int32_t* col, const double* val,
                              bool prune, const int32_t* doff, const double* dval, int ntab, int words,
                              uint64_t* codes, int32_t* fail, hipStream_t st) {
  if (ntab > 255 || (words != 1 && words != 2)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dict_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, rowptr,
                     col, val, prune ? 1 : 0, doff, dval, ntab, words, codes, fail);
  return hipGetLastError();
}
hipError_t launch_dict_types(int64_t n, const uint64_t* codes, int words, const uint64_t* rwords,
                             int ntypes, uint8_t* rtype, int32_t* fail, hipStream_t st) {
  if (ntypes > 255 || (words != 1 && words != 2)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(dict_type_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, codes,
                     words, rwords, ntypes, rtype, fail);
  return hipGetLastError();
}

}  // namespace amg_hip
