// =============================================================================
// solver.cpp -- the C ABI of include/amg_hip.h: solver handle, V-cycle
// orchestration on one HIP stream (replayed as a hipGraph), stand-alone plug-in
// operations and device-pointer launchers.  No CPU fallback anywhere: every
// compute entry point needs a HIP device and fails with AMG_HIP_EHIP otherwise.
// =============================================================================
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>

#include "../../include/amg_hip.h"
#include "host_setup.hpp"
#include "kernels.hpp"

using namespace amg_hip;
#ifdef AMG_PATCH_STAMPS
namespace amg_hip { hipError_t debug_set_patch_stamps(unsigned long long* p); }
#endif

namespace {

thread_local std::string g_err;

amg_hip_status fail(amg_hip_status st, const std::string& msg) {
  g_err = msg;
  return st;
}

#define HIP_TRY(expr)                                                          \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess)                                                      \
      return fail(_e == hipErrorOutOfMemory ? AMG_HIP_ENOMEM : AMG_HIP_EHIP,   \
                  std::string(#expr) + ": " + hipGetErrorString(_e));          \
  } while (0)

int host_threads() {
  unsigned h = std::thread::hardware_concurrency();
  if (h == 0) h = 1;
  return (int)std::min(h, 16u);
}

// ---- device buffers ---------------------------------------------------------
struct DevMem {  // owns one hipMalloc allocation
  void* p = nullptr;
  size_t bytes = 0;
  DevMem() = default;
  DevMem(const DevMem&) = delete;
  DevMem& operator=(const DevMem&) = delete;
  DevMem(DevMem&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
  DevMem& operator=(DevMem&& o) noexcept {
    if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
    return *this;
  }
  ~DevMem() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  // 64 bytes of slack so that 16-byte vector loads of a tail never leave the
  // allocation
  hipError_t alloc(size_t n) {
    release();
    bytes = n;
    return hipMalloc(&p, n + 64);
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

template <class T>
hipError_t upload(DevMem& m, const T* src, size_t count) {
  hipError_t e = m.alloc(count * sizeof(T));
  if (e != hipSuccess) return e;
  if (count == 0) return hipSuccess;
  return hipMemcpy(m.p, src, count * sizeof(T), hipMemcpyHostToDevice);
}

struct DevCsr {  // row-major view on the device
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int max_block_nnz = 0, max_row_nnz = 0;
  DevMem ptr, idx, val;
  const int32_t* rowptr() const { return ptr.as<int32_t>(); }
  const int32_t* col() const { return idx.as<int32_t>(); }
  const double* v() const { return val.as<double>(); }
};

void csr_shape(const Sparse& M, int* max_block, int* max_row) {
  int mb = 0, mr = 0;
  for (int64_t r = 0; r < M.n_outer; ++r) mr = std::max(mr, M.ptr[r + 1] - M.ptr[r]);
  for (int64_t r = 0; r < M.n_outer; r += 256) {
    const int64_t e = std::min<int64_t>(r + 256, M.n_outer);
    mb = std::max(mb, M.ptr[e] - M.ptr[r]);
  }
  *max_block = mb;
  *max_row = mr;
}

hipError_t upload_csr(const Sparse& M, DevCsr* D) {
  D->n_rows = M.n_outer;
  D->n_cols = M.n_inner;
  D->nnz = M.nnz();
  csr_shape(M, &D->max_block_nnz, &D->max_row_nnz);
  hipError_t e;
  if ((e = upload(D->ptr, M.ptr.data(), M.ptr.size())) != hipSuccess) return e;
  if ((e = upload(D->idx, M.idx.data(), M.idx.size())) != hipSuccess) return e;
  return upload(D->val, M.val.data(), M.val.size());
}

// Setup on the device (K-Galerkin): AH = R (A P) for the linear interpolation pair from
// CSR(A) that already sits on the device; AH stays there for the next level and is
// copied to the host (the hierarchy's host copy, layout encoding, getters).
hipError_t device_galerkin(const DevCsr& A, int64_t n_H, DevCsr* AH, Sparse* host) {
  const int64_t n_h = A.n_rows;
  hipError_t e;
  DevMem cnt, bsum, total, ap_ptr, ap_idx, ap_val;
  const int64_t nmax = std::max(n_h, n_H);
  if ((e = cnt.alloc(sizeof(int32_t) * (nmax + 1))) != hipSuccess) return e;
  if ((e = bsum.alloc(sizeof(int64_t) * ((nmax + 1023) / 1024 + 1))) != hipSuccess) return e;
  if ((e = total.alloc(sizeof(int64_t))) != hipSuccess) return e;
  auto scan = [&](int64_t n, DevMem& ptr, int64_t* tot) -> hipError_t {
    hipError_t q;
    if ((q = ptr.alloc(sizeof(int32_t) * (n + 1))) != hipSuccess) return q;
    if ((q = launch_exclusive_scan(n, cnt.as<int32_t>(), ptr.as<int32_t>(), bsum.as<int64_t>(),
                                   total.as<int64_t>(), nullptr)) != hipSuccess)
      return q;
    return hipMemcpy(tot, total.p, sizeof(int64_t), hipMemcpyDeviceToHost);
  };
  // A P
  if ((e = launch_galerkin_ap(false, n_h, n_H, A.rowptr(), A.col(), A.v(), cnt.as<int32_t>(),
                              nullptr, nullptr, nullptr, nullptr)) != hipSuccess)
    return e;
  int64_t nnz_ap = 0;
  if ((e = scan(n_h, ap_ptr, &nnz_ap)) != hipSuccess) return e;
  if (nnz_ap >= ((int64_t)1 << 31) - 1) return hipErrorInvalidValue;
  if ((e = ap_idx.alloc(sizeof(int32_t) * std::max<int64_t>(nnz_ap, 1))) != hipSuccess) return e;
  if ((e = ap_val.alloc(sizeof(double) * std::max<int64_t>(nnz_ap, 1))) != hipSuccess) return e;
  if ((e = launch_galerkin_ap(true, n_h, n_H, A.rowptr(), A.col(), A.v(), nullptr,
                              ap_ptr.as<int32_t>(), ap_idx.as<int32_t>(), ap_val.as<double>(),
                              nullptr)) != hipSuccess)
    return e;
  // R (A P)
  if ((e = launch_galerkin_rap(false, n_h, n_H, ap_ptr.as<int32_t>(), ap_idx.as<int32_t>(),
                               ap_val.as<double>(), cnt.as<int32_t>(), nullptr, nullptr, nullptr,
                               nullptr)) != hipSuccess)
    return e;
  int64_t nnz = 0;
  if ((e = scan(n_H, AH->ptr, &nnz)) != hipSuccess) return e;
  if (nnz >= ((int64_t)1 << 31) - 1) return hipErrorInvalidValue;
  if ((e = AH->idx.alloc(sizeof(int32_t) * std::max<int64_t>(nnz, 1))) != hipSuccess) return e;
  if ((e = AH->val.alloc(sizeof(double) * std::max<int64_t>(nnz, 1))) != hipSuccess) return e;
  if ((e = launch_galerkin_rap(true, n_h, n_H, ap_ptr.as<int32_t>(), ap_idx.as<int32_t>(),
                               ap_val.as<double>(), nullptr, AH->ptr.as<int32_t>(),
                               AH->idx.as<int32_t>(), AH->val.as<double>(), nullptr)) != hipSuccess)
    return e;
  AH->n_rows = AH->n_cols = n_H;
  AH->nnz = nnz;
  if (!host) return hipSuccess;  // device-only setup: the host copy is made when a getter asks
  host->n_outer = host->n_inner = n_H;
  host->ptr.resize(n_H + 1);
  host->idx.resize(nnz);
  host->val.resize(nnz);
  if ((e = hipMemcpy(host->ptr.data(), AH->ptr.p, sizeof(int32_t) * (n_H + 1), hipMemcpyDeviceToHost)) != hipSuccess) return e;
  if (nnz > 0) {
    if ((e = hipMemcpy(host->idx.data(), AH->idx.p, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost)) != hipSuccess) return e;
    if ((e = hipMemcpy(host->val.data(), AH->val.p, sizeof(double) * nnz, hipMemcpyDeviceToHost)) != hipSuccess) return e;
  }
  return hipSuccess;
}

// C = A B on the device (general CSR operands: custom interpolators), count -> scan -> fill.
// *ok = false: a row of A is longer than the kernel's merge width, or an index overflow:
// the caller takes the host product.
hipError_t device_spgemm(const DevCsr& A, const DevCsr& B, DevCsr* C, bool* ok) {
  *ok = false;
  const int64_t n = A.n_rows;
  hipError_t e;
  DevMem cnt, bsum, total, ovf;
  if ((e = cnt.alloc(sizeof(int32_t) * (n + 1))) != hipSuccess) return e;
  if ((e = bsum.alloc(sizeof(int64_t) * ((n + 1023) / 1024 + 1))) != hipSuccess) return e;
  if ((e = total.alloc(sizeof(int64_t))) != hipSuccess) return e;
  if ((e = ovf.alloc(sizeof(int32_t))) != hipSuccess) return e;
  if ((e = hipMemset(ovf.p, 0, sizeof(int32_t))) != hipSuccess) return e;
  if ((e = launch_spgemm(false, n, A.rowptr(), A.col(), A.v(), B.rowptr(), B.col(), B.v(),
                         cnt.as<int32_t>(), nullptr, nullptr, nullptr, ovf.as<int32_t>(), nullptr)) != hipSuccess)
    return e;
  int32_t over = 0;
  if ((e = hipMemcpy(&over, ovf.p, sizeof(int32_t), hipMemcpyDeviceToHost)) != hipSuccess) return e;
  if (over) return hipSuccess;
  if ((e = C->ptr.alloc(sizeof(int32_t) * (n + 1))) != hipSuccess) return e;
  if ((e = launch_exclusive_scan(n, cnt.as<int32_t>(), C->ptr.as<int32_t>(), bsum.as<int64_t>(),
                                 total.as<int64_t>(), nullptr)) != hipSuccess)
    return e;
  int64_t nnz = 0;
  if ((e = hipMemcpy(&nnz, total.p, sizeof(int64_t), hipMemcpyDeviceToHost)) != hipSuccess) return e;
  if (nnz >= ((int64_t)1 << 31) - 1) return hipSuccess;
  if ((e = C->idx.alloc(sizeof(int32_t) * std::max<int64_t>(nnz, 1))) != hipSuccess) return e;
  if ((e = C->val.alloc(sizeof(double) * std::max<int64_t>(nnz, 1))) != hipSuccess) return e;
  if ((e = launch_spgemm(true, n, A.rowptr(), A.col(), A.v(), B.rowptr(), B.col(), B.v(), nullptr,
                         C->ptr.as<int32_t>(), C->idx.as<int32_t>(), C->val.as<double>(),
                         ovf.as<int32_t>(), nullptr)) != hipSuccess)
    return e;
  C->n_rows = n;
  C->n_cols = B.n_cols;
  C->nnz = nnz;
  *ok = true;
  return hipSuccess;
}
// A_H = R (A P) for arbitrary transfer operators (CSR(R), CSR(P) on the device); host copy of
// CSR(A_H) as for device_galerkin
hipError_t device_galerkin_generic(const DevCsr& A, const DevCsr& P, const DevCsr& R, DevCsr* AH,
                                   Sparse* host, bool* ok) {
  DevCsr AP;
  hipError_t e = device_spgemm(A, P, &AP, ok);
  if (e != hipSuccess || !*ok) return e;
  e = device_spgemm(R, AP, AH, ok);
  if (e != hipSuccess || !*ok) return e;
  const int64_t n_H = AH->n_rows, nnz = AH->nnz;
  host->n_outer = host->n_inner = n_H;
  host->ptr.resize(n_H + 1);
  host->idx.resize(nnz);
  host->val.resize(nnz);
  if ((e = hipMemcpy(host->ptr.data(), AH->ptr.p, sizeof(int32_t) * (n_H + 1), hipMemcpyDeviceToHost)) != hipSuccess) return e;
  if (nnz > 0) {
    if ((e = hipMemcpy(host->idx.data(), AH->idx.p, sizeof(int32_t) * nnz, hipMemcpyDeviceToHost)) != hipSuccess) return e;
    if ((e = hipMemcpy(host->val.data(), AH->val.p, sizeof(double) * nnz, hipMemcpyDeviceToHost)) != hipSuccess) return e;
  }
  return hipSuccess;
}

int g_patch_tile_flags = 1;  // amg_hip_set_patch_tile_flags: A/B switch (same bits either way)

// A level matrix on the device in one of the two layouts the kernels take.
struct DevMat {
  bool sell = false;
  int64_t n_rows = 0, nnz = 0;
  DevCsr csr;                 // K-CSR (LDS-staged CSR)
  int max_width = 0;          // K-SELL (64-row panels, lane-interleaved)
  int64_t slots = 0;
  int idx16 = 0;              // scol = int16 offsets from the diagonal column
  DevMem soff, scol, sval;
  bool dict = false;          // K-Dict (dictionary-coded rows)
  int dict_words = 0, dict_wmax = 0, dict_ntab = 0, dict_nt = 0;
  int64_t dict_shift = 0;
  int dict_hb = 0;            // largest |column offset| of the table (half-bandwidth in rows)
  bool dict_typed = false;    // second level: one byte per row into a table of code words
  DevMem dcodes, doff, dval, drtype, drwords;
  DevMem dutd, duti;  // per row type: values / diagonal, offsets + slot mask + stencil pattern (DictRef::utd, uti)
  // K-March (3-D 7-point level: every row type's columns within {-M, -m, -1, 0, 1, m, M}): per type
  // {w(-M), w(-m), w(-1), w(+1), w(+m), w(+M), diagonal, 0}; the interior type and its values
  DevMem dmarch;
  int march_m = 0, march_M = 0, march_ntypes = 0, march_tint = -1;
  double march_wint[6] = {0, 0, 0, 0, 0, 0}, march_dint = 0.0;
  // K-GS-scan: nearest dependency of a lexicographic sweep other than the chained neighbour
  // (min |offset| over the pairs with |offset| >= 2) and the farthest one; 0 = not usable
  int64_t scan_gap = 0, scan_far = 0;
  int scan_new = 99;  // most entries of a row type beyond the chained neighbour on one side
  // K-Patch (temporal blocking): per (row type, slot) table, pitch of the band
  bool patch = false;
  int64_t patch_m = 0;
  int patch_un = 0, patch_ntypes = 0, patch_umask = 0;
  DevMem patch_tab, patch_utabd, patch_utabi, patch_flags;
  DevMem patch_cflags;      // per tile: the coarse rows under it share the diagonal patch_dH (set_patch_coarse_flags)
  double patch_dH = 0.0;
  double patch_cfrac = 0.0;  // fraction of the tiles whose flag is set (bytes accounting)
  PatchRef patch_ref() const {
    PatchRef P;
    P.rtype = drtype.as<uint8_t>();
    P.tflag = g_patch_tile_flags ? patch_flags.as<uint8_t>() : nullptr;
    P.cflag = g_patch_tile_flags ? patch_cflags.as<uint8_t>() : nullptr;
    P.dHu = patch_dH;
    P.ptab = patch_tab.as<double>();
    P.utabd = patch_utabd.as<double>();
    P.utabi = patch_utabi.as<int32_t>();
    P.ntypes = patch_ntypes;
    P.un = patch_un;
    P.umask = patch_umask;
    P.nent = patch_ntypes * patch_un;
    P.nt = dict_nt;
    return P;
  }
  DictRef dict_ref() const {
    DictRef D;
    D.words = dict_words; D.wmax = dict_wmax; D.nt = dict_nt; D.ntab = dict_ntab;
    D.codes = dcodes.as<uint64_t>();
    D.rtype = dict_typed ? drtype.as<uint8_t>() : nullptr;
    D.rwords = dict_typed ? drwords.as<uint64_t>() : nullptr;
    D.doff = doff.as<int32_t>();
    D.dval = dval.as<double>();
    D.scan_new = scan_new;
    D.hb = dict_hb;
    D.utd = (dict_typed && dutd.p) ? dutd.as<double>() : nullptr;
    D.uti = (dict_typed && duti.p) ? duti.as<int32_t>() : nullptr;
    return D;
  }
};

int g_row_types = 1;  // use the second-level (row type) coding when a matrix allows it

int g_index16 = 1;  // use 16-bit relative column indices when a matrix allows it
int g_nontemporal = 1;  // stream large matrices with non-temporal loads

int g_default_layout = AMG_HIP_LAYOUT_AUTO;

// Device copies drop entries that are exactly 0.0 (the Galerkin product keeps
// them structurally, SURVEY F5: 22 % of level 1).  x + 0*u == x for every finite
// u, so results are bit-identical; the host hierarchy (getters) keeps them.
Sparse without_exact_zeros(const Sparse& M) {
  Sparse R;
  R.n_outer = M.n_outer;
  R.n_inner = M.n_inner;
  R.ptr.assign(M.n_outer + 1, 0);
  R.idx.reserve(M.idx.size());
  R.val.reserve(M.val.size());
  for (int64_t o = 0; o < M.n_outer; ++o) {
    for (int32_t p = M.ptr[o]; p < M.ptr[o + 1]; ++p)
      if (M.val[p] != 0.0) {
        R.idx.push_back(M.idx[p]);
        R.val.push_back(M.val[p]);
      }
    R.ptr[o + 1] = (int32_t)R.idx.size();
  }
  return R;
}

hipError_t upload_mat(const Sparse& M, int layout, DevMat* D, int64_t diag_shift, bool allow16);

// K-Patch table of a dictionary-coded, row-typed matrix: finds the pitch m of the band (every
// column offset must be dj * m + di with dj, di in {-1, 0, 1}: the 5-point level and the
// 9-point Galerkin levels of a 2-D grid) and expands type -> code word -> pairs into
// {off-diagonal value, diagonal value, value, LDS offset} per (type, slot).
bool build_patch_table(const DictMat& T, int64_t n, int64_t* m_out, int* un_out, int* ntypes_out,
                       std::vector<double>* tab) {
  if (T.rwords.empty() || T.max_width > 9 || T.doff.empty()) return false;
  int64_t omax = 0;
  for (int32_t o : T.doff) omax = std::max<int64_t>(omax, o < 0 ? -(int64_t)o : o);
  int ntypes = 0;
  for (int t = 0; t < 255; ++t) {  // used types are numbered from 0; unused words are all 0xFF
    bool used = false;
    for (int k = 0; k < T.words; ++k) used = used || T.rwords[(size_t)t * T.words + k] != ~(uint64_t)0;
    if (used) ntypes = t + 1;
  }
  const int un = patch_un(T.max_width);
  if (ntypes < 1 || (ntypes + 1) * un > patch_max_entries()) return false;
  const int pitch = patch_lds_pitch();
  for (int64_t m : {omax, omax - 1}) {
    if (!patch_geometry_ok(n, m)) continue;
    bool ok = true;
    std::vector<double> out((size_t)ntypes * un * 4, 0.0);
    for (int t = 0; t < ntypes && ok; ++t)
      for (int e = 0; e < un; ++e) {
        double* q = &out[((size_t)t * un + e) * 4];
        const int code = e < 8 * T.words ? (int)((T.rwords[(size_t)t * T.words + e / 8] >> (8 * (e % 8))) & 0xFF) : 0xFF;
        if (code == 0xFF) { q[3] = -1.0e9; continue; }
        if (code >= (int)T.doff.size()) { ok = false; break; }
        const int64_t o = T.doff[code];
        const double v = T.dval[code];
        int64_t dj = 0;
        if (o > m / 2) dj = 1;
        else if (o < -(m / 2)) dj = -1;
        const int64_t di = o - dj * m;
        if (di < -1 || di > 1) { ok = false; break; }
        q[0] = o == 0 ? 0.0 : v;
        q[1] = o == 0 ? v : 0.0;
        q[2] = v;
        q[3] = (double)(dj * pitch + di);
      }
    if (!ok) continue;
    *m_out = m;
    *un_out = un;
    *ntypes_out = ntypes;
    tab->swap(out);
    return true;
  }
  return false;
}

// Common tail of the dictionary upload (host encoder: upload_mat; device encoder:
// device_dict_encode): the tables, the layout flags, the K-GS-scan distances and the K-Patch
// tables.  D->dict_typed and the per-row arrays (drtype or dcodes) are set by the caller;
// T needs words, max_width, doff, dval and, when typed, rwords.
hipError_t finish_dict(const DictMat& T, int64_t n, int64_t diag_shift, DevMat* D) {
  hipError_t e;
  D->dict = true;
  D->sell = false;
  D->dict_words = T.words;
  D->dict_wmax = T.max_width;
  D->dict_ntab = (int)T.doff.size();
  D->dict_shift = diag_shift;
  D->dict_hb = 0;
  for (int32_t o : T.doff) D->dict_hb = std::max<int>(D->dict_hb, o < 0 ? -o : o);
  // one sweep streams codes + f + out and gathers x: non-temporal stream when
  // that is well beyond the 256 MiB Infinity Cache
  const double stream_bytes = (double)n * ((D->dict_typed ? 1.0 : 8.0 * T.words) + 24.0);
  D->dict_nt = (g_nontemporal && stream_bytes > 192.0e6) ? 1 : 0;
  if (D->dict_typed)
    if ((e = upload(D->drwords, T.rwords.data(), T.rwords.size())) != hipSuccess) return e;
  if ((e = upload(D->doff, T.doff.data(), T.doff.size())) != hipSuccess) return e;
  if ((e = upload(D->dval, T.dval.data(), T.dval.size())) != hipSuccess) return e;
  if (D->dict_typed && diag_shift == 0 && T.rwords.size() >= (size_t)256 * T.words && T.words <= 2) {
    // The pairs of every row type laid out per slot, and whether its columns form the 7-point
    // pattern {-M, -m, -1, 0, 1, m, M} or the 15-point pattern {-M, -m, 0, m, M} x {-1, 0, 1} with
    // even m < M (kernels.hip: dict_rows_stencil): same values, same slot order as the LDS tables.
    std::vector<double> ud((size_t)256 * 33, 0.0);
    std::vector<int32_t> ui((size_t)256 * 18, 0);
    bool any = false;
    for (int t = 0; t < 255; ++t) {
      double diag = 0.0;
      int32_t mask = 0;
      int cnt = 0;
      int64_t o[16];
      bool dense = true;  // the slots in use are 0 .. cnt-1
      for (int sl = 0; sl < 8 * T.words; ++sl) {
        const int code = (int)((T.rwords[(size_t)t * T.words + sl / 8] >> (8 * (sl % 8))) & 0xFF);
        if (code == 0xFF || code >= (int)T.doff.size()) continue;
        const int32_t off = T.doff[(size_t)code];
        const double v = T.dval[(size_t)code];
        dense = dense && sl == cnt;
        o[cnt++] = off;
        ui[(size_t)t * 18 + sl] = off;
        ud[(size_t)t * 33 + 16 + sl] = v;
        ud[(size_t)t * 33 + sl] = off == 0 ? 0.0 : v;
        if (off == 0) diag = diag + v;
        mask |= (int32_t)1 << sl;
      }
      ud[(size_t)t * 33 + 32] = diag;
      ui[(size_t)t * 18 + 16] = mask;
      int pat = 0;
      if (dense && cnt == 7) {
        const int64_t m = o[5], M = o[6];
        if (o[0] == -M && o[1] == -m && o[2] == -1 && o[3] == 0 && o[4] == 1 && m > 1 && M > m && m % 2 == 0 &&
            M % 2 == 0)
          pat = 7;
      } else if (dense && cnt == 15) {
        const int64_t m = o[10], M = o[13];
        bool okp = m > 2 && M > m + 2 && m % 2 == 0 && M % 2 == 0;
        const int64_t c[5] = {-M, -m, 0, m, M};
        for (int k = 0; k < 5 && okp; ++k)
          okp = o[3 * k] == c[k] - 1 && o[3 * k + 1] == c[k] && o[3 * k + 2] == c[k] + 1;
        if (okp) pat = 15;
      }
      ui[(size_t)t * 18 + 17] = pat;
      any = any || pat != 0;
    }
    if (any) {
      if ((e = upload(D->dutd, ud.data(), ud.size())) != hipSuccess) return e;
      if ((e = upload(D->duti, ui.data(), ui.size())) != hipSuccess) return e;
    }
  }
  D->scan_gap = D->scan_far = 0;
  if (diag_shift == 0) {
    int64_t gap = INT64_MAX, far = 1;
    for (int32_t o : T.doff) {
      const int64_t a = o < 0 ? -(int64_t)o : o;
      if (a >= 2) gap = std::min(gap, a);
      far = std::max(far, a);
    }
    D->scan_gap = gap == INT64_MAX ? (int64_t)1 << 20 : gap;
    D->scan_far = far;
    D->scan_new = 99;
    if (!T.rwords.empty()) {  // per row type: entries below -1 / above +1
      int mx = 0;
      for (int t = 0; t < 255; ++t) {
        int lo = 0, hi = 0;
        for (int e = 0; e < 8 * T.words; ++e) {
          const int code = (int)((T.rwords[(size_t)t * T.words + e / 8] >> (8 * (e % 8))) & 0xFF);
          if (code == 0xFF || code >= (int)T.doff.size()) continue;
          if (T.doff[code] < -1) ++lo;
          if (T.doff[code] > 1) ++hi;
        }
        mx = std::max(mx, std::max(lo, hi));
      }
      D->scan_new = mx;
    }
  }
  D->march_ntypes = 0;
  if (D->dict_typed && diag_shift == 0 && T.rwords.size() >= (size_t)256 * T.words && T.words <= 2) {
    // K-March table: find the interior 7-point type, then lay every type out by pattern position
    int64_t pm = 0, pM = 0;
    int tint = -1;
    std::vector<std::vector<std::pair<int64_t, double>>> rows(255);
    int last = -1;
    for (int t = 0; t < 255; ++t) {
      for (int sl = 0; sl < 8 * T.words; ++sl) {
        const int code = (int)((T.rwords[(size_t)t * T.words + sl / 8] >> (8 * (sl % 8))) & 0xFF);
        if (code == 0xFF || code >= (int)T.doff.size()) continue;
        rows[(size_t)t].push_back({(int64_t)T.doff[(size_t)code], T.dval[(size_t)code]});
      }
      const auto& r = rows[(size_t)t];
      if (!r.empty()) last = t;
      if (tint < 0 && r.size() == 7 && r[2].first == -1 && r[3].first == 0 && r[4].first == 1 &&
          r[5].first > 1 && r[6].first > r[5].first && r[1].first == -r[5].first && r[0].first == -r[6].first &&
          r[6].first % r[5].first == 0) {
        tint = t;
        pm = r[5].first;
        pM = r[6].first;
      }
    }
    if (tint >= 0 && last < 64 && pM < ((int64_t)1 << 30) && n % pM == 0) {
      const int64_t P[6] = {-pM, -pm, -1, 1, pm, pM};
      std::vector<double> wt((size_t)(last + 1) * 8, 0.0);
      bool ok = true;
      for (int t = 0; t <= last && ok; ++t) {
        double diag = 0.0;
        for (const auto& e : rows[(size_t)t]) {
          if (e.first == 0) {
            diag = diag + e.second;
            continue;
          }
          int at = -1;
          for (int u = 0; u < 6; ++u)
            if (P[u] == e.first) at = u;
          if (at < 0) { ok = false; break; }
          wt[(size_t)t * 8 + at] = e.second;
        }
        wt[(size_t)t * 8 + 6] = diag;
      }
      if (ok) {
        if ((e = upload(D->dmarch, wt.data(), wt.size())) != hipSuccess) return e;
        D->march_m = (int)pm;
        D->march_M = (int)pM;
        D->march_ntypes = last + 1;
        D->march_tint = tint;
        for (int u = 0; u < 6; ++u) D->march_wint[u] = wt[(size_t)tint * 8 + u];
        D->march_dint = wt[(size_t)tint * 8 + 6];
      }
    }
  }
  D->patch = false;
  if (D->dict_typed && diag_shift == 0) {
    std::vector<double> tab;
    if (build_patch_table(T, n, &D->patch_m, &D->patch_un, &D->patch_ntypes, &tab)) {
      if ((e = upload(D->patch_tab, tab.data(), tab.size())) != hipSuccess) return e;
      // per-type 3 x 3 slot form for the wave-uniform path (+ the all-absent type `ntypes`):
      // slot = (dj + 1) * 3 + (di + 1) = ascending column offset
      const int un = D->patch_un, nty = D->patch_ntypes, pitch = patch_lds_pitch();
      std::vector<double> ud((size_t)(nty + 1) * 19, 0.0);
      std::vector<int32_t> ui((size_t)(nty + 1) * 2, 0);
      for (int t = 0; t < nty; ++t) {
        double diag = 0.0;
        uint32_t rm = 0;
        for (int k = 0; k < un; ++k) {
          const double* q = &tab[((size_t)t * un + k) * 4];
          if (!(q[3] > -1.0e8)) continue;
          const int lo = (int)q[3];
          const int dj = lo > pitch / 2 ? 1 : (lo < -pitch / 2 ? -1 : 0), di = lo - dj * pitch;
          const int slot = (dj + 1) * 3 + (di + 1);
          ud[(size_t)t * 19 + slot] = q[0];
          ud[(size_t)t * 19 + 9 + slot] = q[2];
          diag += q[1];  // dict_rows' order: +0.0 except the diagonal slot
          rm |= 1u << slot;
        }
        ud[(size_t)t * 19 + 18] = diag;
        ui[(size_t)t * 2] = (int32_t)rm;
        ui[(size_t)t * 2 + 1] = (rm & 0x145u) ? 1 : 0;  // corner slots 0, 2, 6, 8
      }
      if ((e = upload(D->patch_utabd, ud.data(), ud.size())) != hipSuccess) return e;
      if ((e = upload(D->patch_utabi, ui.data(), ui.size())) != hipSuccess) return e;
      // which patches consist of one row type only (the interior of the level)
      int64_t tiles = 0;
      if ((e = launch_patch_tile_flags(n, D->patch_m, nullptr, nty, nullptr, &tiles, nullptr)) != hipSuccess) return e;
      if ((e = D->patch_flags.alloc((size_t)tiles)) != hipSuccess) return e;
      if ((e = launch_patch_tile_flags(n, D->patch_m, D->drtype.as<uint8_t>(), nty,
                                       D->patch_flags.as<uint8_t>(), nullptr, nullptr)) != hipSuccess) return e;
      if ((e = hipDeviceSynchronize()) != hipSuccess) return e;
      // the interior row type = the type most tiles consist of; its slots choose the kernel kind
      {
        std::vector<uint8_t> fl((size_t)tiles);
        if ((e = hipMemcpy(fl.data(), D->patch_flags.p, fl.size(), hipMemcpyDeviceToHost)) != hipSuccess) return e;
        std::vector<int64_t> cnt(256, 0);
        for (uint8_t f : fl) ++cnt[f];
        int best = -1;
        for (int t = 0; t < nty; ++t)
          if (cnt[t] > 0 && (best < 0 || cnt[t] > cnt[best])) best = t;
        D->patch_umask = best >= 0 ? ui[(size_t)best * 2] : patch_default_umask(patch_un(un));
      }
      D->patch = true;
    }
  }
  return hipSuccess;
}

hipError_t upload_mat_pruned(const Sparse& M, int layout, bool prune, DevMat* D,
                             int64_t diag_shift = 0) {
  bool any = false;
  if (prune)
    for (double v : M.val)
      if (v == 0.0) { any = true; break; }
  if (!any) return upload_mat(M, layout, D, diag_shift, true);
  return upload_mat(without_exact_zeros(M), layout, D, diag_shift, true);
}

// layout: AMG_HIP_LAYOUT_*; AUTO takes SELL-64 unless padding exceeds 25 %.
// diag_shift: column of row i's diagonal is i + diag_shift (0 except for the
// halo-extended local blocks of the multi-GPU driver); only used to decide
// whether 16-bit relative column indices fit.
hipError_t upload_mat(const Sparse& M, int layout, DevMat* D, int64_t diag_shift, bool allow16);
hipError_t upload_mat(const Sparse& M, int layout, DevMat* D) {
  return upload_mat(M, layout, D, 0, true);
}
hipError_t upload_mat(const Sparse& M, int layout, DevMat* D, int64_t diag_shift, bool allow16) {
  D->n_rows = M.n_outer;
  D->nnz = M.nnz();
  D->dict = false;
  if ((layout == AMG_HIP_LAYOUT_AUTO || layout == AMG_HIP_LAYOUT_DICT) && allow16 &&
      M.n_outer < ((int64_t)1 << 31) - 512) {
    DictMat T;
    if (to_dict(M, diag_shift, &T)) {
      hipError_t e;
      D->dict_typed = g_row_types != 0 && !T.rtype.empty();
      if (D->dict_typed) {  // the kernels then never touch the per-row code words
        if ((e = upload(D->drtype, T.rtype.data(), T.rtype.size())) != hipSuccess) return e;
      } else {
        if ((e = upload(D->dcodes, T.codes.data(), T.codes.size())) != hipSuccess) return e;
      }
      return finish_dict(T, M.n_outer, diag_shift, D);
    }
    if (layout == AMG_HIP_LAYOUT_DICT) layout = AMG_HIP_LAYOUT_SELL;
  }
  if (layout == AMG_HIP_LAYOUT_DICT) layout = AMG_HIP_LAYOUT_SELL;
  bool sell = layout == AMG_HIP_LAYOUT_SELL;
  Sell64 S;
  if (layout != AMG_HIP_LAYOUT_CSR && M.n_outer < ((int64_t)1 << 31) - 512) {
    to_sell64(M, &S);
    if (layout == AMG_HIP_LAYOUT_AUTO) sell = (double)S.slots() <= 1.25 * (double)M.nnz() + 4096.0;
  } else {
    sell = false;
  }
  D->sell = sell;
  if (!sell) return upload_csr(M, &D->csr);
  D->max_width = S.max_width;
  D->slots = S.slots();
  hipError_t e;
  if ((e = upload(D->soff, S.soff.data(), S.soff.size())) != hipSuccess) return e;
  // 16-bit relative indices when every entry is within +-32767 of its row's diagonal
  bool fits = allow16 && g_index16 != 0;
  std::vector<int16_t> c16;
  if (fits) {
    c16.assign(S.col.size(), (int16_t)-32768);
    const int64_t np = (int64_t)S.soff.size() - 1;
    for (int64_t p = 0; p < np && fits; ++p) {
      const int64_t w = (S.soff[p + 1] - S.soff[p]) / 64;
      for (int64_t j = 0; j < w && fits; ++j)
        for (int64_t l = 0; l < 64; ++l) {
          const int64_t at = S.soff[p] + j * 64 + l;
          const int32_t c = S.col[at];
          if (c < 0) continue;
          const int64_t d = (int64_t)c - (p * 64 + l + diag_shift);
          if (d < -32767 || d > 32767) { fits = false; break; }
          c16[at] = (int16_t)d;
        }
    }
  }
  // non-temporal matrix stream for matrices well beyond the 256 MiB Infinity Cache
  const double stream_bytes = (double)S.slots() * (fits ? 10.0 : 12.0);
  D->idx16 = (fits ? 1 : 0) | ((g_nontemporal && stream_bytes > 192.0e6) ? 2 : 0);
  if (fits) {
    if ((e = upload(D->scol, c16.data(), c16.size())) != hipSuccess) return e;
  } else {
    if ((e = upload(D->scol, S.col.data(), S.col.size())) != hipSuccess) return e;
  }
  return upload(D->sval, S.val.data(), S.val.size());
}

hipError_t launch_mat(int mode, const DevMat& A, const double* x, const double* f, double* out,
                      double omega, hipStream_t st, int64_t diag_shift = 0) {
  if (A.dict) {
    if (diag_shift != A.dict_shift) return hipErrorInvalidValue;  // offsets are baked in
    return launch_dict(mode, A.n_rows, A.dict_ref(), x, f, out, omega, diag_shift, st);
  }
  if (A.sell)
    return launch_sell(mode, A.n_rows, A.idx16, A.soff.as<int64_t>(), A.scol.p,
                       A.sval.as<double>(), x, f, out, omega, diag_shift, st);
  const DevCsr& C = A.csr;
  return launch_csr(mode, C.n_rows, C.nnz, C.max_block_nnz, C.max_row_nnz, C.rowptr(), C.col(),
                    C.v(), x, f, out, omega, diag_shift, st);
}

struct SpikeOnDev {  // device copy of a SpikeFactor + scratch
  SpikeArgs a{};
  DevMem sf, sb, d, V, W, Vt, Wh, G, Z, T, H;
};
hipError_t upload_spike(const SpikeFactor& S, SpikeOnDev* D) {
  hipError_t e;
  if ((e = upload(D->sf, S.sched_f.data(), S.sched_f.size())) != hipSuccess) return e;
  if ((e = upload(D->sb, S.sched_b.data(), S.sched_b.size())) != hipSuccess) return e;
  if ((e = upload(D->d, S.d.data(), S.d.size())) != hipSuccess) return e;
  if ((e = upload(D->V, S.V.data(), S.V.size())) != hipSuccess) return e;
  if ((e = upload(D->W, S.W.data(), S.W.size())) != hipSuccess) return e;
  if ((e = upload(D->Vt, S.Vt.data(), S.Vt.size())) != hipSuccess) return e;
  if ((e = upload(D->Wh, S.Wh.data(), S.Wh.size())) != hipSuccess) return e;
  const size_t wq = S.w > 0 ? S.w : 1;
  if ((e = D->G.alloc(sizeof(double) * (size_t)S.n)) != hipSuccess) return e;
  if ((e = D->Z.alloc(sizeof(double) * (size_t)S.n)) != hipSuccess) return e;
  if ((e = D->T.alloc(sizeof(double) * (size_t)(S.P + 1) * wq)) != hipSuccess) return e;
  if ((e = D->H.alloc(sizeof(double) * (size_t)(S.P + 1) * wq)) != hipSuccess) return e;
  SpikeArgs& a = D->a;
  a.n = S.n; a.w = S.w; a.c = S.c; a.P = S.P; a.m = S.m; a.sched_stride = S.sched_stride;
  a.sched_f = D->sf.as<double>(); a.sched_b = D->sb.as<double>(); a.d = D->d.as<double>();
  a.V = D->V.as<double>(); a.W = D->W.as<double>(); a.Vt = D->Vt.as<double>();
  a.Wh = D->Wh.as<double>(); a.G = D->G.as<double>(); a.Z = D->Z.as<double>();
  a.T = D->T.as<double>(); a.H = D->H.as<double>();
  return hipSuccess;
}

// The factored coarsest operator on the device in one of three forms:
//   BAND  one-wave substitution, half-bandwidth <= 63, bit-exact against the oracle
//   SPIKE partitioned (parallel) form of the same factor, <= 63, agrees to ~1e-14
//   WIDE  blocked one-wave substitution for any half-bandwidth, bit-exact
enum CoarseKind { COARSE_BAND = 0, COARSE_SPIKE = 1, COARSE_WIDE = 2, COARSE_CHAIN = 3 };
int g_no_band_chain = 0;  // amg_hip_set_band_chain
struct CoarseOnDev {
  int kind = COARSE_BAND;
  int64_t n = 0, w = 0;
  int m = 0;
  DevMem sf, sb, d;                  // BAND / WIDE schedules
  std::unique_ptr<SpikeOnDev> spike;
};
// level-0 rows from which the lexicographic smoothers take the line-scan form by default
constexpr int64_t GS_SCAN_MIN_ROWS = 65536;
// ... and the shortest chunk worth it.  A chunk of at most 64 rows is one wave's scan without a
// barrier (~0.3 us), a serial row of the exact kernel costs ~0.33 us: from two rows per chunk on the
// scan form wins (round 3: 4 -> 2; 4096^2 / 16 levels 217 -> 198 ms per cycle, 1024^2 / 12 levels
// 30.4 -> 26.1 ms, and a hierarchy without exact-kernel levels is set up on the device: 2.75 ->
// 0.48 s).  AMG_HIP_SCAN_MIN_GAP overrides.
static const int64_t GS_SCAN_MIN_GAP = [] {
  const char* e = std::getenv("AMG_HIP_SCAN_MIN_GAP");
  return e ? (int64_t)std::atoi(e) : (int64_t)2;
}();
// serial substitution costs ~44 ns per row and pass; from this size on the partitioned
// solve is the default (opt.exact_coarse_solve keeps the bit-exact one)
constexpr int64_t COARSE_SPIKE_MIN_ROWS = 4096;
// multicolour GS: levels of at most this many rows run a whole symmetric pass as ONE launch
// (one workgroup, barriers between the colours) instead of one launch per colour and direction
// Multicolour levels of at most this many rows run the whole symmetric pass in ONE launch (one
// workgroup, a barrier between colours) instead of one launch per colour and direction; larger
// levels lose more to the single CU than the launches cost.  AMG_HIP_MC_ONE_LAUNCH_ROWS overrides
// (tuning / A-B; same bits).
int64_t mc_one_launch_max_rows() {
  static const int64_t v = [] {
    const char* e = std::getenv("AMG_HIP_MC_ONE_LAUNCH_ROWS");
    return e ? std::atoll(e) : (int64_t)8192;
  }();
  return v;
}
// want_fast: 1 = partitioned whenever it applies, 0 = by size, -1 = never
amg_hip_status upload_coarse(const Sparse& A, int want_fast, CoarseOnDev* C) {
  BandFactor F;
  std::string e = band_factor(A, (size_t)8 << 30, &F);
  if (!e.empty()) return fail(AMG_HIP_EINVAL, e);
  C->n = F.n;
  C->w = F.w;
  if (F.w > 63) {
    BandWide W;
    e = band_wide_schedule(F, (size_t)16 << 30, &W);
    if (!e.empty()) return fail(AMG_HIP_EUNSUPPORTED, e);
    C->kind = COARSE_WIDE;
    HIP_TRY(upload(C->sf, W.sched_f.data(), W.sched_f.size()));
    HIP_TRY(upload(C->sb, W.sched_b.data(), W.sched_b.size()));
    HIP_TRY(upload(C->d, W.d.data(), W.d.size()));
    return AMG_HIP_OK;
  }
  if (want_fast > 0 || (want_fast == 0 && F.n >= COARSE_SPIKE_MIN_ROWS)) {
    SpikeFactor SF;
    e = spike_factor(F, &SF);
    if (e.empty()) {
      C->kind = COARSE_SPIKE;
      C->spike.reset(new SpikeOnDev);
      HIP_TRY(upload_spike(SF, C->spike.get()));
      return AMG_HIP_OK;
    }
  }
  if (band_chain_ok(F.n, F.w) && !g_no_band_chain) {
    BandChain S;
    band_chain_schedule(F, &S);
    C->kind = COARSE_CHAIN;
    HIP_TRY(upload(C->sf, S.cf.data(), S.cf.size()));
    HIP_TRY(upload(C->sb, S.cb.data(), S.cb.size()));
    HIP_TRY(upload(C->d, S.d.data(), S.d.size()));
    return AMG_HIP_OK;
  }
  BandSchedule S;
  e = band_schedule(F, &S);
  if (!e.empty()) return fail(AMG_HIP_EUNSUPPORTED, e);
  C->kind = COARSE_BAND;
  C->m = S.m;
  HIP_TRY(upload(C->sf, S.sched_f.data(), S.sched_f.size()));
  HIP_TRY(upload(C->sb, S.sched_b.data(), S.sched_b.size()));
  HIP_TRY(upload(C->d, S.d.data(), S.d.size()));
  return AMG_HIP_OK;
}
// x = A^-1 f; y: scratch of n doubles (BAND / WIDE)
// uh_out (K-BandChain only): also uh_out = uh_in + P x, the prolongation into the level above
hipError_t launch_coarse(const CoarseOnDev& C, const double* f, double* y, double* x,
                         hipStream_t st, int64_t n_h = 0, const double* uh_in = nullptr,
                         double* uh_out = nullptr) {
  switch (C.kind) {
    case COARSE_SPIKE: {
      SpikeArgs a = C.spike->a;
      a.f = f;
      a.x = x;
      return launch_spike_solve(a, st);
    }
    case COARSE_WIDE:
      return launch_band_wide(C.n, C.w, C.sf.as<double>(), C.sb.as<double>(), C.d.as<double>(), f,
                              y, x, st);
    case COARSE_CHAIN:
      return launch_band_chain(C.n, (int)C.w, C.sf.as<double>(), C.sb.as<double>(), C.d.as<double>(),
                               f, x, st, n_h, uh_in, uh_out);
    default:
      return launch_band_solve(C.n, C.m, C.sf.as<double>(), C.sb.as<double>(), C.d.as<double>(),
                               f, y, x, st);
  }
}

struct LexOnDev {
  LexDev d;
  DevMem row, depth, win_depth, col, val, src;
  int64_t n_sets = 0;
};

hipError_t upload_lex(const LexSchedule& S, LexOnDev* L) {
  hipError_t e;
  if ((e = upload(L->row, S.row.data(), S.row.size())) != hipSuccess) return e;
  if ((e = upload(L->depth, S.depth.data(), S.depth.size())) != hipSuccess) return e;
  if ((e = upload(L->win_depth, S.win_depth.data(), S.win_depth.size())) != hipSuccess) return e;
  if ((e = upload(L->col, S.col.data(), S.col.size())) != hipSuccess) return e;
  if ((e = upload(L->val, S.val.data(), S.val.size())) != hipSuccess) return e;
  if ((e = upload(L->src, S.src.data(), S.src.size())) != hipSuccess) return e;
  L->d.block = S.block;
  L->d.width = S.width;
  L->d.n_slots = S.n_slots;
  L->d.row = L->row.as<int32_t>();
  L->d.depth = L->depth.as<int16_t>();
  L->d.win_depth = L->win_depth.as<int32_t>();
  L->d.col = L->col.as<int32_t>();
  L->d.val = L->val.as<double>();
  L->d.src = L->src.as<int16_t>();
  L->n_sets = S.n_sets;
  return hipSuccess;
}

struct Level {
  int64_t n = 0;
  int64_t nnz_struct = 0;  // structural entries (exact zeros of the Galerkin product included)
  // host copy, what get_coefficient_matrix returns.  Device-only setup (amg_hip_create_poisson)
  // leaves it empty and keeps CSR(A) on the device instead; ensure_host_matrix() fills it on demand.
  mutable Sparse A_csc;
  DevCsr A_dev;
  hipError_t ensure_host_matrix() const {
    if (!A_csc.ptr.empty() || A_dev.n_rows <= 0) return hipSuccess;
    Sparse H;  // symmetric: the CSR arrays are the CSC arrays
    H.n_outer = H.n_inner = A_dev.n_rows;
    H.ptr.resize((size_t)A_dev.n_rows + 1);
    H.idx.resize((size_t)A_dev.nnz);
    H.val.resize((size_t)A_dev.nnz);
    hipError_t e = hipMemcpy(H.ptr.data(), A_dev.ptr.p, sizeof(int32_t) * H.ptr.size(), hipMemcpyDeviceToHost);
    if (e == hipSuccess && A_dev.nnz > 0) {
      e = hipMemcpy(H.idx.data(), A_dev.idx.p, sizeof(int32_t) * H.idx.size(), hipMemcpyDeviceToHost);
      if (e == hipSuccess)
        e = hipMemcpy(H.val.data(), A_dev.val.p, sizeof(double) * H.val.size(), hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess) A_csc = std::move(H);
    return e;
  }
  DevMat A_rows;           // rows of A: residual, rss
  DevMat A_cols_own;       // CSC arrays walked as rows (column-as-row smoothers,
  bool symmetric = false;  //   smoother.hpp:101-117); aliases A_rows when bitwise equal
  const DevMat& A_cols() const { return symmetric ? A_rows : A_cols_own; }
  DevMem u, f, r, tmp;
  DevMem diag;             // a_ii (true-Jacobi smoother only)
  // transfers to level+1 (absent on the coarsest level)
  // host copies; for the built-in LinearInterpolator they are only materialised when a
  // getter, the CSR transfer kernels or the host Galerkin product ask for them
  mutable Sparse P_csc, R_csc;
  bool lazy_linear = false;
  int64_t n_coarse = 0;
  const Sparse& P() const {
    if (lazy_linear && P_csc.ptr.empty()) P_csc = linear_P(n, n_coarse);  // interpolator.hpp:106-129
    return P_csc;
  }
  const Sparse& R() const {
    if (lazy_linear && R_csc.ptr.empty()) R_csc = transpose(P());         // :132-134
    return R_csc;
  }
  bool linear = false;
  DevCsr P_rows, R_rows;   // CSR(P), CSR(R)
  // exact lexicographic schedules
  std::unique_ptr<LexOnDev> lex_fwd, lex_bwd;
  int scan_C = 0, scan_ring = 0;  // > 0: the lexicographic sweeps run as K-GS-scan
  // multicolour: colour-permuted SELL-64 copy + dof of every storage row
  std::vector<int32_t> color;
  // <= 4 colours that are a function of (line parity, column class) on the level's 2-D band: the table
  // as 2-bit entries (line & 1) * 6 + class, else -1 (kernels.hip: patch_rb_kernel; classes: the
  // columns 0, 1, m-2, m-1 and the even / odd columns between them)
  int mc_ctab = -1;
  DevMem mc_aux;  // third vector of the patch colour stages when r is kept (opt.keep_residual)
  DevMem mc_color8;  // colour of every row as one byte (K-Strip, kernels.hip: mc_strip_kernel)
  int32_t n_colors = 0;
  DevMat mc_mat;
  DevMem mc_rowid;
  std::vector<int64_t> mc_start;
  DevMem mc_start_dev;  // the same offsets as int32 on the device (small levels: one-launch pass)
  bool mc_dict = false;    // ... dictionary-coded instead (K-Dict colour kernel)
  int mc_words = 0, mc_wmax = 0, mc_ntab = 0;
  DevMem mc_codes, mc_doff, mc_dval;
};

}  // namespace

// Row-block ("slab") sharding of the K-Patch levels: every rank holds the whole
// hierarchy and full-size vectors, but runs the patch kernels of levels 0..levels-1 only over
// its own grid lines plus a halo deep enough that NO exchange is needed inside those legs
// (the halo lines are recomputed redundantly: each kernel stage makes one more line at either
// end of the range stale, and the ranges below are what is left valid where it is needed).
// The level below the slab levels is all-gathered and the rest of the cycle runs replicated.
struct Slab {
  int levels = 0;               // 0: not configured
  int rank = 0, world = 1;
  int64_t lines = 0, chunk = 0, line0 = 0, line1 = 0;   // grid lines; lines per rank; owned [line0, line1)
  int halo = 0;                 // lines of level-0 u a neighbour supplies before each cycle
  std::vector<int64_t> down_lo, down_hi, up_lo, up_hi;  // lines each leg runs over, per level
  hipGraph_t graph[3] = {nullptr, nullptr, nullptr};
  hipGraphExec_t exec[3] = {nullptr, nullptr, nullptr};
  void reset_graphs() {
    for (int i = 0; i < 3; ++i) {
      if (exec[i]) (void)hipGraphExecDestroy(exec[i]);
      if (graph[i]) (void)hipGraphDestroy(graph[i]);
      exec[i] = nullptr;
      graph[i] = nullptr;
    }
  }
};

// amg_hip_set_tail_fusion / AMG_HIP_TAIL_FUSION=1: K-Tail (deepest levels + coarsest solve in one
// launch).  Off by default: bit-identical, and measured 0.5 % SLOWER than one launch per step on the
// 4096^2 cycle (1282 / 1291 / 1293 against 1299 / 1301 / 1294 V-cycles/s, alternating runs).
// K-March on / off (AMG_HIP_MARCH=0: two launches of the sweep)
int g_march = [] {
  const char* e = std::getenv("AMG_HIP_MARCH");
  return (e && *e == '0') ? 0 : 1;
}();
int g_tail_fusion = [] {
  const char* e = std::getenv("AMG_HIP_TAIL_FUSION");
  return (e && *e == '1') ? 1 : 0;
}();
int64_t g_patch_min_rows = 1000000;  // amg_hip_set_patch_min_rows (level 4 of 4096^2 has 1 048 575 rows)

struct amg_hip_solver {
  amg_hip_options opt;
  int64_t patch_min_rows = g_patch_min_rows;  // the process-wide value when the solver was made
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;
  std::vector<Level> lv;
  CoarseOnDev coarse;  // coarsest level factor
  DevMem scratch;   // 1024 doubles + 1 result
  DevMem pcg_x, pcg_p, pcg_q, pcg_b;  // PCG work vectors (allocated on first use)
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  bool graph_ready = false;
  double cycle_bytes = 0, fine_sweep_bytes = 0;
  // bytes the launches of one V-cycle HAVE TO move (what each kernel reads and writes once in the
  // layout it really streams), summed while the cycle is enqueued; [part] as enqueue_vcycle's
  double must_move[4] = {0, 0, 0, 0};
  int acct_part = -1;  // -1: not inside enqueue_vcycle
  void acct(double bytes) { if (acct_part >= 0) must_move[acct_part] += bytes; }
  Slab slab;

  ~amg_hip_solver() {
    slab.reset_graphs();
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (stream && own_stream) (void)hipStreamDestroy(stream);
  }
};

namespace {

void layout_of(const DevMat& A, int32_t* layout, int64_t* matrix_stream_bytes);  // below
// matrix stream of one pass over a level matrix in its device layout (dictionary: 1 B row type per
// row + tables; SELL / CSR: indices + values), the figure amg_hip_level_layout reports
double mat_bytes(const DevMat& A) {
  int64_t b = 0;
  layout_of(A, nullptr, &b);
  return (double)b;
}

// ---- smoother on one level --------------------------------------------------
amg_hip_status enqueue_multicolor(amg_hip_solver* s, Level& L, hipStream_t st, bool again);

// phase 0: u_l is arbitrary.  phase 1: u_l is known to be zero (pre-smoothing of a
// coarse level, multigrid.hpp:278).  phase 2: u_l still lacks the correction
// P u_{l+1} (multigrid.hpp:294-296), the smoother must apply it.  Only the true
// Jacobi smoother has shortcuts for phases 1/2; callers fall back otherwise.
bool jacobi_fuses_zero(const amg_hip_solver* s) {
  return s->opt.smoother == AMG_HIP_SM_JACOBI && !s->opt.no_fusion && s->opt.smoother_iters >= 1;
}
bool jacobi_fuses_prolong(const amg_hip_solver* s, int l) {
  const Level& L = s->lv[l];
  // measured slower than the separate prolongation kernel (3 gathers per entry:
  // 337 vs 248+70 us on the 4096^2 fine level), so only on request
  return s->opt.fuse_prolong && jacobi_fuses_zero(s) && L.linear && s->opt.stencil_transfers &&
         L.A_cols().sell;
}

// Fusions of the true-Jacobi V-cycle across a level boundary (kernels.hip, "fused
// forms"): both need the dictionary-coded layout and the matrix-free linear transfers.
// down: residual of level l + restriction + first (from-zero) sweep of level l+1
bool fuses_resid_restrict(const amg_hip_solver* s, int l) {
  if (l + 1 >= (int)s->lv.size()) return false;
  const Level& L = s->lv[l];
  if (s->opt.no_fusion || !L.linear || !s->opt.stencil_transfers || !L.A_rows.dict ||
      L.A_rows.dict_shift != 0)
    return false;
  // true Jacobi: the kernel also does the first coarse sweep and needs the coarse diagonal
  return !jacobi_fuses_zero(s) || s->lv[l + 1].diag.p != nullptr;
}
// K-Patch: the whole down-leg / up-leg of a big level in one launch each (kernels.hip).
// Levels 0 .. k of a 2+2 true-Jacobi cycle whose matrices are row-typed dictionaries with a
// 2-D band (the finer level's kernel hands the first sweep of the next one over).
bool patch_level_ok(const amg_hip_solver* s, int l) {
  // needs a smoothed coarser level (a window's last level is one: it is smoothed by the tail solver)
  if (l < 0 || l + (s->opt.window ? 1 : 2) >= (int)s->lv.size()) return false;
  const Level& L = s->lv[l];
  const DevMat& A = L.A_rows;
  if (!(jacobi_fuses_zero(s) && !s->opt.fuse_prolong && s->opt.smoother_iters == 2 && L.symmetric &&
        A.dict && A.dict_shift == 0 && A.patch && L.linear && s->opt.stencil_transfers &&
        L.n >= s->patch_min_rows && s->lv[l + 1].diag.p != nullptr))
    return false;
  return l == 0 || patch_level_ok(s, l - 1);
}

// Multicolour smoother on a level whose colours repeat on the 2 x 2 cells of its 2-D band (the
// checkerboard of the 5-point level, line parity, the product colouring of the 9-point levels): the
// symmetric passes as patch stages, the residual + restriction and the prolongation fused into them
// (kernels.hip: patch_rb_kernel)
bool mc_patch_ok(const amg_hip_solver* s, int l) {
  if (l < 0 || l + 1 >= (int)s->lv.size()) return false;
  const Level& L = s->lv[l];
  const DevMat& A = L.A_rows;
  return s->opt.smoother == AMG_HIP_SM_MULTICOLOR_GS && s->opt.smoother_iters >= 1 && !s->opt.no_fusion &&
         L.symmetric && A.dict && A.dict_shift == 0 && A.patch && L.linear && s->opt.stencil_transfers &&
         L.n >= s->patch_min_rows && L.mc_ctab >= 0 && s->lv[l + 1].n >= 2 &&
         (!s->opt.keep_residual || L.mc_aux.p != nullptr);
}
// The colour stages of smoother_iters symmetric passes (0 .. nc-1, nc-1 .. 0 each).  A colour that
// directly follows itself is dropped: its rows read no row of their own colour (that is what the
// colouring means; explicit zeros add +-0.0), so the repeat would store the bits already there.
static std::vector<int> mc_sequence(int nc, int iters) {
  std::vector<int> q;
  for (int it = 0; it < iters; ++it) {
    for (int c = 0; c < nc; ++c)
      if (q.empty() || q.back() != c) q.push_back(c);
    for (int c = nc - 1; c >= 0; --c)
      if (q.back() != c) q.push_back(c);
  }
  return q;
}
// The launches of one level and cycle in the patch form: the down-leg (the last launch carries the
// residual + restriction and at most patch_rb_max_stages(true) stages), then the up-leg (the first
// launch prolongs while loading).  A launch cannot write the vector it reads (tiles overlap), so
// the level vector travels u -> tmp -> r -> tmp ... -> u (r, or mc_aux when r is kept).
struct McLaunch {
  uint32_t stages = 0;
  bool prolong = false, tail = false;
  double *in = nullptr, *out = nullptr;
};
static std::vector<McLaunch> mc_plan(amg_hip_solver* s, int l, int* n_down) {
  Level& L = s->lv[l];
  const std::vector<int> q = mc_sequence(L.n_colors, s->opt.smoother_iters);
  const int n = (int)q.size();
  std::vector<McLaunch> P;
  auto chunks = [&](int from, int count, int cap, bool prolong_first) {
    if (count <= 0) return;
    const int k = (count + cap - 1) / cap;
    int at = from;
    for (int c = 0; c < k; ++c) {
      const int len = count / k + (c < count % k ? 1 : 0);
      McLaunch M;
      M.stages = patch_rb_stages(q.data() + at, len);
      M.prolong = prolong_first && c == 0;
      P.push_back(M);
      at += len;
    }
  };
  const int tail_cnt = std::min(n, patch_rb_max_stages(true));
  chunks(0, n - tail_cnt, patch_rb_max_stages(false), false);
  {
    McLaunch M;
    M.stages = patch_rb_stages(q.data() + (n - tail_cnt), tail_cnt);
    M.tail = true;
    P.push_back(M);
  }
  *n_down = (int)P.size();
  chunks(0, n, patch_rb_max_stages(false), true);
  double* third = s->opt.keep_residual ? L.mc_aux.as<double>() : L.r.as<double>();
  double* cur = L.u.as<double>();
  for (size_t i = 0; i < P.size(); ++i) {
    P[i].in = cur;
    P[i].out = i + 1 == P.size() ? L.u.as<double>() : ((i & 1) ? third : L.tmp.as<double>());
    cur = P[i].out;
  }
  return P;
}

// up: last post-smoothing sweep of level l+1 + prolongation into level l
bool fuses_jacobi_prolong(const amg_hip_solver* s, int l) {
  if (l + 2 >= (int)s->lv.size()) return false;  // the coarsest level is solved, not smoothed
  if (patch_level_ok(s, l)) return false;  // a patch level prolongs into itself while loading
  const Level& L = s->lv[l];
  const DevMat& AC = s->lv[l + 1].A_cols();
  return jacobi_fuses_zero(s) && !jacobi_fuses_prolong(s, l) && L.linear &&
         s->opt.stencil_transfers && AC.dict && AC.dict_shift == 0;
}

// two sweeps in one launch (kernels.hip: dict_pair_*_kernel): small, narrow-band,
// symmetric levels of the 2+2 true-Jacobi cycle
bool pair_level_ok(const amg_hip_solver* s, int l) {
  if (l < 1 || l + 1 >= (int)s->lv.size()) return false;
  const Level& L = s->lv[l];
  const DevMat& A = L.A_rows;
  return jacobi_fuses_zero(s) && s->opt.smoother_iters == 2 && L.symmetric && A.dict &&
         A.dict_shift == 0 && fuses_resid_restrict(s, l) &&
         dict_pair_ok(L.n, A.dict_ref(), A.dict_hb, L.u.p, L.f.p, L.tmp.p);
}
bool pair_up_ok(const amg_hip_solver* s, int l) {  // its second sweep prolongs into level l-1
  return pair_level_ok(s, l) && fuses_jacobi_prolong(s, l - 1);
}

// K-Tail: first level lt such that the levels lt .. L-2, the coarsest solve and the prolongation
// into lt - 1 run as ONE launch (kernels.hip: tail_kernel), or -1.
int tail_from(const amg_hip_solver* s) {
  const int nl = (int)s->lv.size();
  if (!g_tail_fusion || s->opt.keep_residual || s->opt.window || nl < 3 || s->coarse.kind != COARSE_CHAIN ||
      s->coarse.n > tail_max_coarse())
    return -1;
  int lt = nl - 1;
  while (lt - 1 >= 1 && nl - 1 - (lt - 1) <= TAIL_LEVELS_MAX) {
    const int l = lt - 1;
    const Level& L = s->lv[l];
    if (!(pair_level_ok(s, l) && pair_up_ok(s, l) && tail_level_ok(L.n, L.A_rows.dict_ref(), L.A_rows.dict_hb) &&
          L.diag.p && s->lv[l + 1].diag.p))
      break;
    lt = l;
  }
  // the level vectors live in LDS: drop levels from the top until the launch fits
  for (; lt <= nl - 2; ++lt) {
    TailRef T;
    std::memset(&T, 0, sizeof(T));
    T.nlev = nl - 1 - lt;
    for (int q = 0; q < T.nlev; ++q) T.L[q].n = (int)s->lv[lt + q].n;
    T.nc = (int)s->coarse.n;
    T.wc = (int)s->coarse.w;
    if (tail_lds_bytes(T) <= tail_lds_capacity()) return lt;
  }
  return -1;
}

// K-March: the two plain Jacobi sweeps u -> tmp -> u of a 3-D 7-point level as ONE launch u -> tmp
// (kernels.hip: march_kernel); the level's two vectors then trade places (a level is swept twice
// per cycle this way -- down-leg and up-leg --, so every cycle ends with them where it found them).
static bool march_fill(const amg_hip_solver* s, int l, MarchRef* R) {
  const Level& L = s->lv[l];
  const DevMat& A = L.A_cols();
  if (!(s->opt.smoother == AMG_HIP_SM_JACOBI && s->opt.smoother_iters == 2 && !s->opt.no_fusion && !s->opt.window &&
        s->slab.levels == 0 && A.dict && A.dict_typed && A.march_ntypes > 0 && A.dmarch.p && A.march_m > 0 &&
        L.n % A.march_M == 0 && A.march_M % A.march_m == 0 && g_march))
    return false;
  R->m = A.march_m;
  R->lines = A.march_M / A.march_m;
  R->planes = (int)(L.n / A.march_M);
  R->ntypes = A.march_ntypes;
  R->tint = A.march_tint;
  R->omega = s->opt.omega;
  R->dint = A.march_dint;
  for (int u = 0; u < 6; ++u) R->wint[u] = A.march_wint[u];
  R->wtab = A.dmarch.as<double>();
  R->rtype = A.drtype.as<uint8_t>();
  R->f = L.f.as<double>();
  R->x = L.u.as<double>();
  R->out = L.tmp.as<double>();
  R->chunk_planes = 1;
  return march_ok(*R);
}
// phase 3: the first sweep was already done by the fused kernel of the finer level
// (result in tmp).  prolong_into >= 0: the last sweep also adds P u_l to that level's u.
amg_hip_status enqueue_smooth(amg_hip_solver* s, int l, int phase = 0, int prolong_into = -1,
                              double* prolong_out = nullptr) {
  Level& L = s->lv[l];
  hipStream_t st = s->stream;
  const int iters = s->opt.smoother_iters;
  switch (s->opt.smoother) {
    case AMG_HIP_SM_SPGS:
      // forward + backward sweep: matrix, f, u in, u out each (the scan form's pre-pass also
      // writes and re-reads one number per row: + 16 n)
      s->acct(iters * 2.0 * (mat_bytes(L.A_rows) + 24.0 * L.n + (L.scan_C > 0 ? 16.0 * L.n : 0.0)));
      for (int it = 0; it < iters; ++it) {
        if (L.scan_C > 0) {  // line-scan form (kernels.hip: K-GS-scan)
          const DictRef D = L.A_rows.dict_ref();
          HIP_TRY(launch_gs_scan(L.n, D, L.f.as<double>(), L.u.as<double>(), L.tmp.as<double>(), false, 0,
                                 1.0, L.scan_C, L.scan_ring, st));
          HIP_TRY(launch_gs_scan(L.n, D, L.f.as<double>(), L.u.as<double>(), L.tmp.as<double>(), true, 0,
                                 1.0, L.scan_C, L.scan_ring, st));
          continue;
        }
        HIP_TRY(launch_gs_lex(L.lex_fwd->d, L.f.as<double>(), L.u.as<double>(), 0, 1.0, st));
        HIP_TRY(launch_gs_lex(L.lex_bwd->d, L.f.as<double>(), L.u.as<double>(), 0, 1.0, st));
      }
      return AMG_HIP_OK;
    case AMG_HIP_SM_REF_JACOBI:
    case AMG_HIP_SM_SOR: {
      const int mode = s->opt.smoother == AMG_HIP_SM_SOR ? 2 : 1;
      s->acct(iters * (mat_bytes(L.A_rows) + 24.0 * L.n + (L.scan_C > 0 ? 16.0 * L.n : 0.0)));
      for (int it = 0; it < iters; ++it) {
        if (L.scan_C > 0) {
          HIP_TRY(launch_gs_scan(L.n, L.A_rows.dict_ref(), L.f.as<double>(), L.u.as<double>(),
                                 L.tmp.as<double>(), false, mode, s->opt.omega, L.scan_C, L.scan_ring, st));
          continue;
        }
        HIP_TRY(launch_gs_lex(L.lex_fwd->d, L.f.as<double>(), L.u.as<double>(), mode,
                              s->opt.omega, st));
      }
      return AMG_HIP_OK;
    }
    case AMG_HIP_SM_JACOBI: {
      const DevMat& A = L.A_cols();
      double* a = L.u.as<double>();
      double* b = L.tmp.as<double>();
      int it = 0;
      if (phase == 3) {
        std::swap(a, b);
        it = 1;
      } else if (phase == 1 && jacobi_fuses_zero(s)) {
        HIP_TRY(launch_jacobi_from_zero(L.n, L.diag.as<double>(), L.f.as<double>(), b,
                                        s->opt.omega, st));
        s->acct(24.0 * L.n);  // diagonal, f, out
        std::swap(a, b);
        it = 1;
      } else if (phase == 2 && jacobi_fuses_prolong(s, l)) {
        Level& C = s->lv[l + 1];
        HIP_TRY(launch_sell_jacobi_prolong(A.n_rows, A.idx16, A.soff.as<int64_t>(),
                                           A.scol.p, A.sval.as<double>(), a,
                                           C.u.as<double>(), C.n, L.f.as<double>(), b,
                                           s->opt.omega, st));
        s->acct(mat_bytes(A) + 24.0 * L.n + 8.0 * C.n);
        std::swap(a, b);
        it = 1;
      }
      if (it == 0 && prolong_into < 0) {
        MarchRef R;
        // (only inside a whole V-cycle, which sweeps the level twice: a lone smooth() must leave the
        // vectors where a captured graph expects them)
        if (s->acct_part == 0 /* CYCLE_ALL */ && march_fill(s, l, &R)) {  // both sweeps in one pass: u -> tmp, then trade
          HIP_TRY(launch_march(R, st));
          s->acct(mat_bytes(A) + 24.0 * L.n);
          std::swap(L.u, L.tmp);
          return AMG_HIP_OK;
        }
      }
      for (; it < iters; ++it) {
        if (prolong_into >= 0 && it == iters - 1) {
          Level& F = s->lv[prolong_into];
          HIP_TRY(launch_dict_jacobi_prolong(A.n_rows, A.dict_ref(), a, L.f.as<double>(), b,
                                             s->opt.omega, F.n, F.u.as<double>(),
                                             prolong_out ? prolong_out : F.u.as<double>(), st));
          s->acct(mat_bytes(A) + 24.0 * L.n + 16.0 * F.n);  // sweep + u_F in, u_F + P u out
        } else {
          HIP_TRY(launch_mat(CSR_JACOBI, A, a, L.f.as<double>(), b, s->opt.omega, st));
          s->acct(mat_bytes(A) + 24.0 * L.n);  // matrix, f, x, out
        }
        std::swap(a, b);
      }
      if (iters & 1) {  // result sits in tmp: bring it home (keeps the graph static)
        HIP_TRY(hipMemcpyAsync(L.u.p, L.tmp.p, sizeof(double) * L.n,
                               hipMemcpyDeviceToDevice, st));
        s->acct(16.0 * L.n);
      }
      return AMG_HIP_OK;
    }
    case AMG_HIP_SM_MULTICOLOR_GS:
      for (int it = 0; it < iters; ++it) {
        amg_hip_status r = enqueue_multicolor(s, L, st, it > 0);
        if (r != AMG_HIP_OK) return r;
      }
      return AMG_HIP_OK;
  }
  return fail(AMG_HIP_EINVAL, "unknown smoother kind");
}

// K-Strip: the whole leg of a narrow multicolour level (below the K-Patch pitch) in one launch
// over strips of rows.  *T_out: rows per strip -- about n / 256 (one wave of workgroups), within
// what the LDS window leaves after the halo of (stages + 1) half-bandwidths on either side.
bool mc_strip_ok(const amg_hip_solver* s, int l, int* T_out = nullptr, int* nst_out = nullptr) {
  if (l < 0 || l + 1 >= (int)s->lv.size()) return false;
  const Level& L = s->lv[l];
  const DevMat& A = L.A_rows;
  if (!(s->opt.smoother == AMG_HIP_SM_MULTICOLOR_GS && s->opt.smoother_iters >= 1 && !s->opt.no_fusion &&
        L.symmetric && A.dict && A.dict_typed && A.dict_shift == 0 && L.linear && s->opt.stencil_transfers &&
        L.mc_color8.p != nullptr && L.n_colors >= 1 && L.n_colors <= 15 && s->lv[l + 1].n >= 1 && L.n >= 2 &&
        !mc_patch_ok(s, l)))
    return false;
  const int nst = 2 * L.n_colors * s->opt.smoother_iters - (2 * s->opt.smoother_iters - 1);  // mc_sequence
  if (nst < 1 || nst > 16) return false;
  const int hbw = (A.dict_hb + 1) & ~1;
  const int64_t T_max = ((int64_t)strip_window_max() - 2 - 2 * (int64_t)(nst + 1) * std::max(hbw, 2)) & ~(int64_t)1;
  if (T_max < 256) return false;
  int64_t T = ((L.n + 255) / 256 + 1) & ~(int64_t)1;
  T = std::min<int64_t>(std::max<int64_t>(T, 256), T_max);
  if (T_out) *T_out = (int)T;
  if (nst_out) *nst_out = nst;
  return true;
}
static StripRef mc_strip_ref(amg_hip_solver* s, int l) {
  Level& L = s->lv[l];
  StripRef R;
  int T = 0, nst = 0;
  (void)mc_strip_ok(s, l, &T, &nst);
  const std::vector<int> q = mc_sequence(L.n_colors, s->opt.smoother_iters);
  R.n = (int)L.n;
  R.nH = (int)s->lv[l + 1].n;
  R.T = T;
  R.hbw = std::max((L.A_rows.dict_hb + 1) & ~1, 2);
  R.nst = (int)q.size();
  for (size_t i = 0; i < q.size() && i < 16; ++i) R.stages |= (uint64_t)(q[i] & 15) << (4 * i);
  R.color = L.mc_color8.as<uint8_t>();
  R.f = L.f.as<double>();
  return R;
}
// one symmetric pass: colours 0..nc-1 then nc-1..0; a colour that directly follows itself is not
// launched again (mc_sequence: the repeat would store the bits already there).  again: the pass
// follows another pass of the same smoothing call (its colour 0 directly follows colour 0).
amg_hip_status enqueue_multicolor(amg_hip_solver* s, Level& L, hipStream_t st, bool again) {
  const DevMat& A = L.mc_mat;
  if (L.mc_dict && L.mc_start_dev.p && !s->opt.no_fusion) {  // small level: the whole pass in one launch
    // every row twice: code words 8 B per word and row + 4 B dof id, f, u written
    s->acct(2.0 * ((8.0 * L.mc_words + 4.0) * L.n + 16.0 * L.n));
    HIP_TRY(launch_dict_gs_sweep(L.n_colors, L.mc_start_dev.as<int32_t>(), L.mc_start.back(), L.mc_words,
                                 L.mc_wmax, L.mc_codes.as<uint64_t>(), L.mc_rowid.as<int32_t>(),
                                 L.mc_doff.as<int32_t>(), L.mc_dval.as<double>(), L.mc_ntab,
                                 L.f.as<double>(), L.u.as<double>(), st));
    return AMG_HIP_OK;
  }
  // per row of a launched colour: matrix stream (dictionary: code words + 4 B dof id; SELL panels:
  // indices + values + 4 B dof id), f, u written
  const double row_bytes = (L.mc_dict ? 8.0 * L.mc_words + 4.0 : (mat_bytes(A) + 4.0 * L.n) / (double)std::max<int64_t>(L.n, 1)) + 16.0;
  auto one = [&](int c) -> hipError_t {
    s->acct(row_bytes * (double)(L.mc_start[c + 1] - L.mc_start[c]));
    if (L.mc_dict)
      return launch_dict_gs_color(L.mc_start[c], L.mc_start[c + 1] - L.mc_start[c], L.mc_words,
                                  L.mc_wmax, L.mc_codes.as<uint64_t>(), L.mc_rowid.as<int32_t>(),
                                  L.mc_doff.as<int32_t>(), L.mc_dval.as<double>(), L.mc_ntab,
                                  L.f.as<double>(), L.u.as<double>(), st);
    return launch_sell_gs_color(A.n_rows, A.max_width, A.soff.as<int64_t>(), A.scol.as<int32_t>(),
                                A.sval.as<double>(), L.mc_rowid.as<int32_t>(), L.mc_start[c],
                                L.mc_start[c + 1] - L.mc_start[c], L.f.as<double>(),
                                L.u.as<double>(), st);
  };
  const bool dedupe = !s->opt.no_fusion;
  for (int c = (again && dedupe) ? 1 : 0; c < L.n_colors; ++c) HIP_TRY(one(c));
  for (int c = L.n_colors - (dedupe ? 2 : 1); c >= 0; --c) HIP_TRY(one(c));
  return AMG_HIP_OK;
}

amg_hip_status enqueue_residual(amg_hip_solver* s, int l) {
  Level& L = s->lv[l];
  HIP_TRY(launch_mat(CSR_RESID, L.A_rows, L.u.as<double>(), L.f.as<double>(), L.r.as<double>(),
                     1.0, s->stream));
  s->acct(mat_bytes(L.A_rows) + 24.0 * L.n);  // matrix, f, u, r
  return AMG_HIP_OK;
}

// multigrid.hpp:263-305.  part: the whole cycle, or one of the three pieces a slab-sharded
// cycle is cut into at its two exchange points (struct Slab): the down-legs of the slab levels
// over this rank's lines, the replicated rest below them (it begins by redoing the from-zero
// sweep of its first level on the gathered right-hand side), the up-legs of the slab levels.
enum { CYCLE_ALL = 0, CYCLE_SLAB_DOWN = 1, CYCLE_SLAB_TAIL = 2, CYCLE_SLAB_UP = 3 };
// roctx ranges per level and leg, AMG_HIP_ROCTX=1: "L<l> down" / "L<l> up" /
// "coarse solve" around what is ENQUEUED for them (librocprofiler-sdk-roctx by dlopen; the ranges
// bracket the launches, so they line up with the kernels in a `rocprofv3 --marker-trace
// --kernel-trace` run of an eager cycle, use_graph = 0 / bench.py --no-graph; under a captured
// graph they mark the capture only).
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (!std::getenv("AMG_HIP_ROCTX")) return;
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
    pop = (int (*)())dlsym(h, "roctxRangePop");
    if (!push || !pop) push = nullptr;
  }
};
struct RoctxRange {
  static Roctx& api() {
    static Roctx r;
    return r;
  }
  bool on = false;
  RoctxRange(const char* what, int level) {
    if (!api().push) return;
    char name[48];
    if (level >= 0) std::snprintf(name, sizeof(name), "L%d %s", level, what);
    else std::snprintf(name, sizeof(name), "%s", what);
    api().push(name);
    on = true;
  }
  ~RoctxRange() {
    if (on) api().pop();
  }
};

amg_hip_status enqueue_vcycle_body(amg_hip_solver* s, int part);
amg_hip_status enqueue_vcycle(amg_hip_solver* s, int part = CYCLE_ALL) {
  s->must_move[part] = 0;
  s->acct_part = part;
  const amg_hip_status r = enqueue_vcycle_body(s, part);
  s->acct_part = -1;
  return r;
}
amg_hip_status enqueue_vcycle_body(amg_hip_solver* s, int part) {
  const int nl = (int)s->lv.size();
  hipStream_t st = s->stream;
  const Slab& sb = s->slab;
  const int k = sb.levels;
  const bool ranged = part == CYCLE_SLAB_DOWN || part == CYCLE_SLAB_UP;
  const bool zero_known = jacobi_fuses_zero(s);  // coarse pre-smoothing starts from u == 0
  bool first_sweep_done = false;  // by the fused residual+restrict kernel of level l-1
  const int tail_lt = (part == CYCLE_ALL || part == CYCLE_SLAB_TAIL) ? tail_from(s) : -1;
  bool tail_done = false;
  if (part == CYCLE_SLAB_TAIL) {
    Level& L = s->lv[k];
    HIP_TRY(launch_jacobi_from_zero(L.n, L.diag.as<double>(), L.f.as<double>(), L.tmp.as<double>(),
                                    s->opt.omega, st));
    s->acct(24.0 * L.n);
    first_sweep_done = true;
  }
  const int down_from = part == CYCLE_SLAB_TAIL ? k : 0;
  const int down_to = part == CYCLE_SLAB_DOWN ? k : (part == CYCLE_SLAB_UP ? 0 : nl);
  for (int l = down_from; l < down_to; ++l) {
    RoctxRange range("down", l);
    // On the coarsest level the reference smooths and forms the residual, then overwrites
    // u with the direct solve of the level's rhs (multigrid.hpp:268-274, :287-288): unless
    // the residual is to be kept, neither has an observable effect.
    if (l == nl - 1 && nl > 1 && !s->opt.keep_residual) break;
    if (mc_patch_ok(s, l)) {  // :268 (the colour stages of the symmetric passes) + :272-282
      Level& L = s->lv[l];
      Level& C = s->lv[l + 1];
      const DevMat& A = L.A_rows;
      int nd = 0;
      const std::vector<McLaunch> plan = mc_plan(s, l, &nd);
      for (int q = 0; q < nd; ++q) {
        const McLaunch& M = plan[(size_t)q];
        HIP_TRY(launch_patch_rb(false, M.tail, L.n, A.patch_m, A.patch_ref(), M.in, L.f.as<double>(), nullptr, C.n,
                                M.out, (M.tail && s->opt.keep_residual) ? L.r.as<double>() : nullptr,
                                M.tail ? C.f.as<double>() : nullptr, M.tail ? C.u.as<double>() : nullptr, M.stages,
                                (uint32_t)L.mc_ctab, st));
      }
      // each launch: row types + x + f + out; the last also f_H, zeroed u_H (and r when kept)
      s->acct(nd * 25.0 * L.n + 16.0 * C.n + (s->opt.keep_residual ? 8.0 * L.n : 0.0));
      first_sweep_done = false;
      continue;
    }
    if (mc_strip_ok(s, l)) {  // :268 + :272-282 of a narrow level, one launch: u -> tmp
      Level& L = s->lv[l];
      Level& C = s->lv[l + 1];
      StripRef R = mc_strip_ref(s, l);
      R.x = L.u.as<double>();
      R.u_out = L.tmp.as<double>();
      R.r_out = s->opt.keep_residual ? L.r.as<double>() : nullptr;
      R.fH = C.f.as<double>();
      R.uH_zero = C.u.as<double>();
      HIP_TRY(launch_mc_strip(false, true, R, L.A_rows.dict_ref(), st));
      // row types + colours + x + f + out; f_H, zeroed u_H (and r when kept)
      s->acct(26.0 * L.n + 16.0 * C.n + (s->opt.keep_residual ? 8.0 * L.n : 0.0));
      first_sweep_done = false;
      continue;
    }
    if (patch_level_ok(s, l)) {  // :268 (both sweeps) + :272-282 + :268 of l+1, one launch
      Level& L = s->lv[l];
      Level& C = s->lv[l + 1];
      const DevMat& A = L.A_rows;
      const bool first = l == 0;  // l >= 1: the first sweep came from level l-1's kernel (in tmp)
      HIP_TRY(launch_patch_down(first, L.n, A.patch_m, A.patch_ref(),
                                first ? L.u.as<double>() : L.tmp.as<double>(), L.f.as<double>(),
                                first ? L.tmp.as<double>() : L.u.as<double>(),
                                s->opt.keep_residual ? L.r.as<double>() : nullptr, C.n,
                                C.f.as<double>(), C.diag.as<double>(), C.tmp.as<double>(),
                                s->opt.omega, st, ranged ? sb.down_lo[l] : 0,
                                ranged ? sb.down_hi[l] : -1));
      {  // row types + x + f + smoothed u; f_H, first coarse sweep, coarse diagonal
        double frac = 1.0;
        if (ranged && sb.down_hi[l] >= 0) {
          const double lines = (double)((L.n + A.patch_m - 1) / A.patch_m);
          frac = std::min(1.0, (double)(sb.down_hi[l] - sb.down_lo[l]) / lines);
        }
        // (the coarse diagonal is not read under the tiles that take it as an argument)
        s->acct(frac * (25.0 * L.n + (24.0 - 8.0 * A.patch_cfrac) * C.n + (s->opt.keep_residual ? 8.0 * L.n : 0.0)));
      }
      first_sweep_done = true;
      continue;
    }
    if (first_sweep_done && l == tail_lt) {  // levels l .. L-2, the solve and the way back: ONE launch
      TailRef T;
      std::memset(&T, 0, sizeof(T));
      T.nlev = nl - 1 - l;
      for (int q = 0; q < T.nlev; ++q) {
        Level& Q = s->lv[l + q];
        const DictRef D = Q.A_rows.dict_ref();
        TailLevelRef& R = T.L[q];
        R.n = (int)Q.n;
        R.hbw = (Q.A_rows.dict_hb + 1) & ~1;
        R.words = D.words;
        R.ntab = D.ntab;
        R.rtype = D.rtype;
        R.rwords = D.rwords;
        R.doff = D.doff;
        R.dval = D.dval;
        R.f = Q.f.as<double>();
        R.u = Q.u.as<double>();
        R.tmp = Q.tmp.as<double>();
        R.diag = Q.diag.as<double>();
      }
      Level& C = s->lv[nl - 1];
      T.nc = (int)C.n;
      T.wc = (int)s->coarse.w;
      T.cf = s->coarse.sf.as<double>();
      T.cb = s->coarse.sb.as<double>();
      T.dg = s->coarse.d.as<double>();
      T.fc = C.f.as<double>();
      T.uc = C.u.as<double>();
      T.tmpc = C.tmp.as<double>();
      T.diagc = C.diag.as<double>();
      Level& F = s->lv[l - 1];
      T.n_fine = (int)F.n;
      T.uf_in = F.u.as<double>();
      T.uf_out = pair_up_ok(s, l - 1) ? F.tmp.as<double>() : F.u.as<double>();
      T.omega = s->opt.omega;
      HIP_TRY(launch_tail(T, st));
      for (int q = 0; q < T.nlev; ++q) {  // what the pair kernels of these levels would have moved
        const Level& Q = s->lv[l + q];
        const double nH = (double)s->lv[l + q + 1].n, nF = (double)s->lv[l + q - 1].n;
        s->acct(2.0 * (mat_bytes(Q.A_rows) + 24.0 * Q.n) + 24.0 * nH + 16.0 * nF);
      }
      s->acct(16.0 * (double)C.n * (double)std::max<int64_t>(s->coarse.w, 1) + 24.0 * C.n);
      tail_done = true;
      break;
    }
    if (first_sweep_done && pair_level_ok(s, l)) {  // second pre-sweep + residual + restriction
      Level& L = s->lv[l];
      Level& C = s->lv[l + 1];
      const DevMat& A = L.A_rows;
      HIP_TRY(launch_dict_pair_down(A.n_rows, A.dict_ref(), A.dict_hb, L.tmp.as<double>(),
                                    L.f.as<double>(), L.u.as<double>(),
                                    s->opt.keep_residual ? L.r.as<double>() : nullptr, C.n,
                                    C.f.as<double>(), C.diag.as<double>(), C.tmp.as<double>(),
                                    s->opt.omega, st));
      s->acct(mat_bytes(A) + 24.0 * L.n + 24.0 * C.n + (s->opt.keep_residual ? 8.0 * L.n : 0.0));
      continue;  // first_sweep_done stays true for level l+1
    }
    amg_hip_status r =
        enqueue_smooth(s, l, first_sweep_done ? 3 : (l >= 1 && zero_known) ? 1 : 0);  // :268
    if (r != AMG_HIP_OK) return r;
    first_sweep_done = false;
    if (fuses_resid_restrict(s, l)) {                              // :272-282 + :268 of l+1
      Level& L = s->lv[l];
      Level& C = s->lv[l + 1];
      const DevMat& A = L.A_rows;
      HIP_TRY(launch_dict_resid_restrict(A.n_rows, A.dict_ref(), L.u.as<double>(),
                                         L.f.as<double>(),
                                         s->opt.keep_residual ? L.r.as<double>() : nullptr, C.n,
                                         C.f.as<double>(),
                                         zero_known ? C.diag.as<double>() : nullptr,
                                         zero_known ? C.tmp.as<double>() : nullptr,
                                         C.u.as<double>(), s->opt.omega, st));
      // matrix, u, f (r when kept); f_H and either the first coarse sweep + diagonal or the zeroed u_H
      s->acct(mat_bytes(A) + 16.0 * L.n + (s->opt.keep_residual ? 8.0 * L.n : 0.0) +
              (zero_known ? 24.0 : 16.0) * C.n);
      first_sweep_done = zero_known;
      continue;
    }
    if ((r = enqueue_residual(s, l)) != AMG_HIP_OK) return r;      // :272-274
    if (l + 1 != nl) {
      Level& L = s->lv[l];
      Level& C = s->lv[l + 1];
      // :278 -- when the smoother starts from "u == 0" without reading u, the
      // fill itself is dead (u_{l+1} is fully overwritten before anyone reads it)
      if (L.linear && s->opt.stencil_transfers) {                  // :278 + :281-282
        HIP_TRY(launch_linear_restrict(L.n, C.n, L.r.as<double>(), C.f.as<double>(),
                                       zero_known ? nullptr : C.u.as<double>(), st));
        s->acct(8.0 * L.n + (zero_known ? 8.0 : 16.0) * C.n);
      } else {
        if (!zero_known) HIP_TRY(hipMemsetAsync(C.u.p, 0, sizeof(double) * C.n, st)); // :278
        const DevCsr& R = L.R_rows;
        HIP_TRY(launch_csr(CSR_SPMV, R.n_rows, R.nnz, R.max_block_nnz, R.max_row_nnz,
                           R.rowptr(), R.col(), R.v(), L.r.as<double>(), nullptr,
                           C.f.as<double>(), 1.0, 0, st));
        s->acct(12.0 * R.nnz + 4.0 * C.n + 8.0 * L.n + (zero_known ? 8.0 : 16.0) * C.n);
      }
    }
  }
  // where the prolongation INTO level l lands: a level whose two post-sweeps run as one
  // launch reads u + P u_{l+1} from tmp (its second sweep then writes u; no in-place race
  // between the tiles), every other level takes it in place
  auto up_target = [&](int l) -> double* {
    return pair_up_ok(s, l) ? s->lv[l].tmp.as<double>() : s->lv[l].u.as<double>();
  };
  int prolonged_by_solve = -1;  // level whose :294-296 the coarsest solve's launch did (K-BandChain)
  if (!ranged && !tail_done) {                                     // :287-288
    RoctxRange range("coarse solve", -1);
    if (s->opt.window)
      return fail(AMG_HIP_EINVAL, "a window solver (opt.window) runs by parts: amg_hip_window_run");
    Level& C = s->lv[nl - 1];
    const int lp = nl - 2;  // plain stencil prolongation into level L-2: one launch less when fused
    const bool fuse_p = lp >= 0 && s->coarse.kind == COARSE_CHAIN && !s->opt.no_fusion && s->lv[lp].linear &&
                        s->opt.stencil_transfers && !mc_patch_ok(s, lp) && !mc_strip_ok(s, lp) &&
                        !patch_level_ok(s, lp) && !jacobi_fuses_prolong(s, lp) && !fuses_jacobi_prolong(s, lp) &&
                        (part == CYCLE_ALL || (part == CYCLE_SLAB_TAIL && lp >= k));
    if (fuse_p) {
      Level& Lp = s->lv[lp];
      HIP_TRY(launch_coarse(s->coarse, C.f.as<double>(), C.tmp.as<double>(), C.u.as<double>(), st, Lp.n,
                            Lp.u.as<double>(), up_target(lp)));
      s->acct(16.0 * Lp.n);
      prolonged_by_solve = lp;
    } else {
      HIP_TRY(launch_coarse(s->coarse, C.f.as<double>(), C.tmp.as<double>(), C.u.as<double>(), st));
    }
    // banded L D L^T solve: the band of L forwards and backwards, D, f, u
    s->acct(16.0 * (double)C.n * (double)std::max<int64_t>(s->coarse.w, 1) + 24.0 * C.n);
  }
  const int up_from = part == CYCLE_SLAB_UP ? k - 1 : (part == CYCLE_SLAB_DOWN ? -1 : (tail_done ? tail_lt - 1 : nl - 2));
  const int up_to = part == CYCLE_SLAB_TAIL ? k : 0;
  for (int l = up_from; l >= up_to; --l) {                         // :291
    RoctxRange range("up", l);
    Level& L = s->lv[l];
    Level& C = s->lv[l + 1];
    if (mc_patch_ok(s, l)) {  // :294-296 + :300 (the colour stages of the symmetric passes)
      const DevMat& A = L.A_rows;
      int nd = 0;
      const std::vector<McLaunch> plan = mc_plan(s, l, &nd);
      for (size_t q = (size_t)nd; q < plan.size(); ++q) {
        const McLaunch& M = plan[q];
        HIP_TRY(launch_patch_rb(M.prolong, false, L.n, A.patch_m, A.patch_ref(), M.in, L.f.as<double>(),
                                M.prolong ? C.u.as<double>() : nullptr, C.n, M.out, nullptr, nullptr, nullptr,
                                M.stages, (uint32_t)L.mc_ctab, st));
      }
      s->acct((double)(plan.size() - (size_t)nd) * 25.0 * L.n + 8.0 * C.n);
      continue;
    }
    if (mc_strip_ok(s, l)) {  // :294-296 + :300 of a narrow level, one launch: tmp + P u_H -> u
      StripRef R = mc_strip_ref(s, l);
      R.x = L.tmp.as<double>();
      R.u_out = L.u.as<double>();
      R.uH = C.u.as<double>();
      HIP_TRY(launch_mc_strip(true, false, R, L.A_rows.dict_ref(), st));
      s->acct(26.0 * L.n + 8.0 * C.n);
      continue;
    }
    if (patch_level_ok(s, l)) {  // :294-296 + :300 (both sweeps), one launch
      // level 0 rests in u (its down-leg wrote tmp); coarser patch levels end in tmp
      const DevMat& A = L.A_rows;
      const bool top = l == 0;
      const double* uH = patch_level_ok(s, l + 1) ? C.tmp.as<double>() : C.u.as<double>();
      HIP_TRY(launch_patch_up(L.n, A.patch_m, A.patch_ref(),
                              top ? L.tmp.as<double>() : L.u.as<double>(), L.f.as<double>(), uH, C.n,
                              top ? L.u.as<double>() : L.tmp.as<double>(), s->opt.omega, st,
                              ranged ? sb.up_lo[l] : 0, ranged ? sb.up_hi[l] : -1));
      {
        double frac = 1.0;
        if (ranged && sb.up_hi[l] >= 0) {
          const double lines = (double)((L.n + A.patch_m - 1) / A.patch_m);
          frac = std::min(1.0, (double)(sb.up_hi[l] - sb.up_lo[l]) / lines);
        }
        s->acct(frac * (25.0 * L.n + 8.0 * C.n));  // row types + x + f + out; u_H
      }
      continue;
    }
    // the last sweep of this level also prolongs into level l-1 when that fuses
    const int into = (l >= 1 && fuses_jacobi_prolong(s, l - 1)) ? l - 1 : -1;
    if (jacobi_fuses_prolong(s, l)) {                              // :294-296 inside :300
      amg_hip_status r = enqueue_smooth(s, l, 2, into, into >= 0 ? up_target(into) : nullptr);
      if (r != AMG_HIP_OK) return r;
      continue;
    }
    if (fuses_jacobi_prolong(s, l)) {
      // :294-296 was done by the last sweep of level l+1
    } else if (l == prolonged_by_solve) {
      // :294-296 was done by the launch of the coarsest solve
    } else if (L.linear && s->opt.stencil_transfers) {             // :294-296
      if (up_target(l) != L.u.as<double>())
        HIP_TRY(launch_linear_prolong_to(L.n, C.n, C.u.as<double>(), L.u.as<double>(), up_target(l), st));
      else
        HIP_TRY(launch_linear_prolong_add(L.n, C.n, C.u.as<double>(), L.u.as<double>(), st));
      s->acct(8.0 * C.n + 16.0 * L.n);
    } else {
      const DevCsr& P = L.P_rows;  // u_h = u_h + P u_H in one launch
      HIP_TRY(launch_csr(CSR_SPMV_ADD, P.n_rows, P.nnz, P.max_block_nnz, P.max_row_nnz,
                         P.rowptr(), P.col(), P.v(), C.u.as<double>(), L.u.as<double>(),
                         L.u.as<double>(), 1.0, 0, st));
      s->acct(12.0 * P.nnz + 4.0 * L.n + 8.0 * C.n + 16.0 * L.n);
    }
    if (pair_up_ok(s, l)) {                                        // :300, both sweeps
      Level& F = s->lv[l - 1];
      const DevMat& A = L.A_cols();
      HIP_TRY(launch_dict_pair_up(A.n_rows, A.dict_ref(), A.dict_hb, L.tmp.as<double>(),
                                  L.f.as<double>(), L.u.as<double>(), s->opt.omega, F.n,
                                  F.u.as<double>(), up_target(l - 1), st));
      s->acct(mat_bytes(A) + 24.0 * L.n + 16.0 * F.n);
      continue;
    }
    amg_hip_status r = enqueue_smooth(s, l, 0, into, into >= 0 ? up_target(into) : nullptr);  // :300
    if (r != AMG_HIP_OK) return r;
  }
  return AMG_HIP_OK;
}

amg_hip_status ensure_graph(amg_hip_solver* s) {
  if (s->graph_ready) return AMG_HIP_OK;
  HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
  amg_hip_status r = enqueue_vcycle(s);
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(s->stream, &g);
  if (r != AMG_HIP_OK) {
    if (g) (void)hipGraphDestroy(g);
    return r;
  }
  if (e != hipSuccess) return fail(AMG_HIP_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  s->graph = g;
  HIP_TRY(hipGraphInstantiate(&s->graph_exec, s->graph, nullptr, nullptr, 0));
  s->graph_ready = true;
  return AMG_HIP_OK;
}

amg_hip_status set_device(const amg_hip_solver* s) {
  if (s->opt.host_only)
    return fail(AMG_HIP_EINVAL, "solver was created with host_only = 1: no device state");
  HIP_TRY(hipSetDevice(s->device));
  return AMG_HIP_OK;
}

// K-Patch down-legs: which tiles sit over coarse rows that all share one diagonal value (the
// interior of the coarse level) -- those workgroups take it as a kernel argument instead of loading
// it per coarse row at the end of their life.  Called once the whole hierarchy is on the device.
amg_hip_status set_patch_coarse_flags(amg_hip_solver* s) {
  for (size_t l = 0; l + 1 < s->lv.size(); ++l) {
    Level& L = s->lv[l];
    Level& C = s->lv[l + 1];
    DevMat& A = L.A_rows;
    if (!A.patch || !C.diag.p || C.n < 1) continue;
    int64_t tiles = 0;
    HIP_TRY(launch_patch_tile_flags(L.n, A.patch_m, nullptr, A.patch_ntypes, nullptr, &tiles, nullptr));
    // reference value: the coarse row under the middle of the middle line (C.n / 2 would be a line end)
    const int64_t lines = (L.n + A.patch_m - 1) / A.patch_m;
    const int64_t cref = std::min<int64_t>(C.n - 1, ((lines / 2) * A.patch_m + A.patch_m / 2) >> 1);
    double dref = 0.0;
    HIP_TRY(hipMemcpy(&dref, C.diag.as<double>() + cref, sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(A.patch_cflags.alloc((size_t)tiles));
    HIP_TRY(launch_patch_coarse_flags(L.n, A.patch_m, C.n, C.diag.as<double>(), dref,
                                      A.patch_cflags.as<uint8_t>(), nullptr));
    A.patch_dH = dref;
    std::vector<uint8_t> fl((size_t)tiles);
    HIP_TRY(hipMemcpy(fl.data(), A.patch_cflags.p, fl.size(), hipMemcpyDeviceToHost));
    int64_t on = 0;
    for (uint8_t f : fl) on += f != 0;
    A.patch_cfrac = g_patch_tile_flags && tiles > 0 ? (double)on / (double)tiles : 0.0;
  }
  HIP_TRY(hipDeviceSynchronize());
  return AMG_HIP_OK;
}

// ---- bytes model ----------------------------------------------
void compute_bytes(amg_hip_solver* s) {
  const int nl = (int)s->lv.size();
  double total = 0;
  const int iters = s->opt.smoother_iters;
  int sweeps_per_smooth = iters;
  if (s->opt.smoother == AMG_HIP_SM_SPGS || s->opt.smoother == AMG_HIP_SM_MULTICOLOR_GS)
    sweeps_per_smooth = 2 * iters;
  for (int l = 0; l < nl; ++l) {
    const Level& L = s->lv[l];
    const double sweep = 12.0 * (double)L.nnz_struct + 28.0 * (double)L.n;
    if (l == 0) s->fine_sweep_bytes = sweep;
    // pre-smooth + residual on every level; post-smooth on all but the coarsest
    total += sweep * (sweeps_per_smooth + 1);
    if (l + 1 != nl) {
      total += sweep * sweeps_per_smooth;
      const double nH = (double)s->lv[l + 1].n, nh = (double)L.n;
      total += 8.0 * nH;                                   // zero
      total += 12.0 * 3 * nH + 4 * nH + 8 * nh + 8 * nH;   // restrict (CSR R)
      total += 12.0 * 3 * nH + 4 * nh + 8 * nH + 16 * nh;  // prolong + add (CSR P)
    }
  }
  s->cycle_bytes = total;
}

// ---- setup --------------------------------------------------------------------
hipError_t device_dict_encode(const DevCsr& A, bool prune, int maxlen, DevMat* D, bool* ok);

amg_hip_status build_solver(int64_t n, const int32_t* colptr, const int32_t* rowind,
                            const double* val, const double* b, int32_t n_levels,
                            const int32_t* const* Pc, const int32_t* const* Pr,
                            const double* const* Pv, const int32_t* const* Rc,
                            const int32_t* const* Rr, const double* const* Rv,
                            const amg_hip_options* opts, amg_hip_solver** out,
                            double rs_theta = -1.0, int64_t rs_min_coarse = 0) {
  if (!out) return fail(AMG_HIP_EINVAL, "out handle pointer is null");
  *out = nullptr;
  if (!colptr || !rowind || !val || !b) return fail(AMG_HIP_EINVAL, "null input array");
  const bool rs = rs_theta >= 0.0;  // strength-based C/F coarsening: n_levels is an upper bound
  if (n <= 0) return fail(AMG_HIP_EINVAL, "`A` must have at least one degree of freedom");
  if (n_levels < 1) return fail(AMG_HIP_EINVAL, "`n_levels` must be at least 1");
  std::unique_ptr<amg_hip_solver> s(new amg_hip_solver);
  if (opts) s->opt = *opts;
  else amg_hip_default_options(&s->opt);
  if (s->opt.smoother < 0 || s->opt.smoother > AMG_HIP_SM_MULTICOLOR_GS)
    return fail(AMG_HIP_EINVAL, "unknown smoother kind");
  if (s->opt.smoother_iters < 0) return fail(AMG_HIP_EINVAL, "`smoother_iters` must be >= 0");
  if (s->opt.layout < AMG_HIP_LAYOUT_AUTO || s->opt.layout > AMG_HIP_LAYOUT_DICT)
    return fail(AMG_HIP_EINVAL, "unknown matrix layout");
  if (s->opt.smoother == AMG_HIP_SM_SOR && (s->opt.omega > 2 || s->opt.omega < 0))
    return fail(AMG_HIP_EINVAL, "`omega` must be in [0, 2] but got omega=" +
                                    std::to_string(s->opt.omega) + "\n");  // smoother.hpp:286-293
  const bool dev = !s->opt.host_only;
  if (dev) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return fail(AMG_HIP_EHIP, "no HIP device available (this library has no CPU fallback)");
    if (s->opt.device >= 0) {
      if (s->opt.device >= ndev) return fail(AMG_HIP_EINVAL, "device ordinal out of range");
      s->device = s->opt.device;
    } else {
      HIP_TRY(hipGetDevice(&s->device));
    }
    HIP_TRY(hipSetDevice(s->device));
    if (s->opt.stream) {
      s->stream = (hipStream_t)s->opt.stream;
      s->own_stream = false;
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    }
  }

  // AMG_HIP_TIMING=1: wall time of the setup phases on stderr
  struct PhaseTimer {
    bool on = std::getenv("AMG_HIP_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    std::vector<std::pair<const char*, double>> acc;
    void lap(const char* name) {
      if (!on) return;
      const auto t1 = std::chrono::steady_clock::now();
      const double dt = std::chrono::duration<double>(t1 - t0).count();
      t0 = t1;
      for (auto& a : acc)
        if (a.first == name) { a.second += dt; return; }
      acc.emplace_back(name, dt);
    }
    ~PhaseTimer() {
      if (!on) return;
      std::fprintf(stderr, "amg_hip setup:");
      for (auto& a : acc) std::fprintf(stderr, " %s %.2fs", a.first, a.second);
      std::fprintf(stderr, "\n");
    }
  } timer;
  static const char *T_IN = "input+transpose", *T_ENC = "encode+upload", *T_SM = "smoother-setup",
                    *T_TR = "transfers", *T_RAP = "galerkin", *T_CSC = "csc-copy", *T_BAND = "band";
  const int nt = host_threads();
  s->lv.resize(n_levels);
  {
    Level& L0 = s->lv[0];
    L0.n = n;
    L0.A_csc = from_raw(n, n, colptr, rowind, val);
    std::string v = validate(L0.A_csc, "A");
    if (!v.empty()) return fail(AMG_HIP_EINVAL, v);
  }
  // ---- hierarchy (multigrid.hpp:211-237) ----
  Sparse A_r = transpose(s->lv[0].A_csc);  // CSR(A_0)
  DevCsr galerkin_A;             // CSR(A_l) on the device while the Galerkin chain runs there
  bool galerkin_on_dev = false;
  timer.lap(T_IN);
  for (int l = 0; l < n_levels; ++l) {
    Level& L = s->lv[l];
    L.symmetric = same_arrays(A_r, L.A_csc);
    L.nnz_struct = L.A_csc.nnz();
    if (dev) {
    const bool prune = !s->opt.keep_structural_zeros;
    // Symmetric levels headed for the dictionary layout are encoded ON THE DEVICE from CSR(A_l)
    // (K-Setup: the encoder kernels check every row against a table proposed from a sample of
    // rows); the arrays are there anyway for the Galerkin chain.  Everything else takes the
    // host encoder.
    bool encoded = false;
    if (L.symmetric && (s->opt.layout == AMG_HIP_LAYOUT_AUTO || s->opt.layout == AMG_HIP_LAYOUT_DICT) &&
        L.n >= (1 << 16) && A_r.nnz() < ((int64_t)1 << 31) - 1) {
      if (!galerkin_on_dev) {
        HIP_TRY(upload_csr(A_r, &galerkin_A));
        galerkin_on_dev = true;
      }
      DevMem stats;
      HIP_TRY(stats.alloc(sizeof(int32_t) * 2));
      HIP_TRY(hipMemset(stats.p, 0, sizeof(int32_t) * 2));
      HIP_TRY(L.diag.alloc(sizeof(double) * L.n));
      HIP_TRY(launch_csr_inspect(L.n, galerkin_A.rowptr(), galerkin_A.col(), galerkin_A.v(), prune,
                                 stats.as<int32_t>(), L.diag.as<double>(), nullptr));
      int32_t st2[2];
      HIP_TRY(hipMemcpy(st2, stats.p, sizeof(st2), hipMemcpyDeviceToHost));
      if (st2[1] == 0) HIP_TRY(device_dict_encode(galerkin_A, prune, st2[0], &L.A_rows, &encoded));
      if (!encoded || s->opt.smoother != AMG_HIP_SM_JACOBI) L.diag.release();
    }
    if (!encoded) {
      HIP_TRY(upload_mat_pruned(A_r, s->opt.layout, prune, &L.A_rows));
      if (!L.symmetric && s->opt.smoother >= AMG_HIP_SM_JACOBI)
        HIP_TRY(upload_mat_pruned(L.A_csc, s->opt.layout, prune, &L.A_cols_own));
    }
    if (s->opt.smoother == AMG_HIP_SM_JACOBI && !L.diag.p) {  // diagonal of the column-as-row walk
      std::vector<double> dg(L.n, 0.0);
      for (int64_t c = 0; c < L.n; ++c)
        for (int32_t p = L.A_csc.ptr[c]; p < L.A_csc.ptr[c + 1]; ++p)
          if (L.A_csc.idx[p] == c) dg[c] = L.A_csc.val[p];
      HIP_TRY(upload(L.diag, dg.data(), dg.size()));
    }
    HIP_TRY(L.u.alloc(sizeof(double) * L.n));
    HIP_TRY(L.f.alloc(sizeof(double) * L.n));
    HIP_TRY(L.r.alloc(sizeof(double) * L.n));
    HIP_TRY(L.tmp.alloc(sizeof(double) * L.n));
    HIP_TRY(hipMemset(L.u.p, 0, sizeof(double) * L.n));
    HIP_TRY(hipMemset(L.f.p, 0, sizeof(double) * L.n));
    HIP_TRY(hipMemset(L.r.p, 0, sizeof(double) * L.n));
    timer.lap(T_ENC);
    // smoother-specific structures
    // Lexicographic sweeps at size: the line-scan form, unless opt.exact_gs or a small
    // problem (the exact dependency-scheduled kernel is bit-exact and fast enough there)
    if (s->opt.smoother <= AMG_HIP_SM_SOR && !s->opt.exact_gs && s->lv[0].n > GS_SCAN_MIN_ROWS &&
        L.symmetric) {
      const DevMat& A = L.A_rows;
      if (A.dict && A.dict_shift == 0 && A.scan_gap >= GS_SCAN_MIN_GAP) {
        int ring = 128;
        int C = (int)std::min<int64_t>(A.scan_gap, gs_scan_long_chunks_ok(A.dict_ref()) ? gs_scan_max_chunk() : 1024);
        if (C > 1024 && A.scan_far + C + 1 > 16384) C = 1024;  // the ring of new values must fit the LDS
        while (ring < A.scan_far + C + 1 && ring <= 16384) ring *= 2;
        if (ring <= 16384) {
          L.scan_C = C;
          L.scan_ring = ring;
        }
      }
    }
    if (L.scan_C > 0) {
      // no schedule needed
    } else if (s->opt.smoother == AMG_HIP_SM_SPGS) {
      LexSchedule F, B;
      std::string e = build_lex_schedule(L.A_csc, false, 64, &F);
      if (e.empty()) e = build_lex_schedule(L.A_csc, true, 64, &B);
      if (!e.empty()) return fail(AMG_HIP_EUNSUPPORTED, e);
      L.lex_fwd.reset(new LexOnDev);
      L.lex_bwd.reset(new LexOnDev);
      HIP_TRY(upload_lex(F, L.lex_fwd.get()));
      HIP_TRY(upload_lex(B, L.lex_bwd.get()));
    } else if (s->opt.smoother == AMG_HIP_SM_REF_JACOBI || s->opt.smoother == AMG_HIP_SM_SOR) {
      LexSchedule F;  // these two address A by ROW (A.coeff(i,j), smoother.hpp:251,351)
      std::string e = build_lex_schedule(A_r, false, 64, &F);
      if (!e.empty()) return fail(AMG_HIP_EUNSUPPORTED, e);
      L.lex_fwd.reset(new LexOnDev);
      HIP_TRY(upload_lex(F, L.lex_fwd.get()));
    } else if (s->opt.smoother == AMG_HIP_SM_MULTICOLOR_GS) {
      greedy_coloring(L.A_csc, &L.color, &L.n_colors);
      // K-Patch form of the passes (patch_rb_kernel): the colour of a row is a function of the
      // parities of its grid line and column
      L.mc_ctab = -1;
      if (L.n_colors <= 4 && L.A_rows.patch && L.symmetric && (L.A_rows.patch_m % 2) == 0 &&
          L.A_rows.patch_m >= 8 && L.n >= 2 * L.A_rows.patch_m) {
        const int64_t m = L.A_rows.patch_m;
        auto cls = [m](int64_t c) -> int {
          return c == 0 ? 0 : (c == 1 ? 1 : (c == m - 2 ? 2 : (c == m - 1 ? 3 : 4 + (int)(c & 1))));
        };
        int32_t tab[12];
        const int64_t sample[6] = {0, 1, m - 2, m - 1, 2, 3};
        for (int p = 0; p < 2; ++p)
          for (int q = 0; q < 6; ++q) tab[p * 6 + q] = L.color[(size_t)(p * m + sample[q])];
        bool ok = true;
        for (int64_t i0 = 0, line = 0; i0 < L.n && ok; i0 += m, ++line) {
          const int32_t* t = tab + 6 * (line & 1);
          const int32_t* c = L.color.data() + i0;
          const int64_t len = std::min<int64_t>(m, L.n - i0);
          for (int64_t j = 0; j < len; ++j) ok &= c[j] == t[(j < 2 || j >= m - 2) ? cls(j) : 4 + (j & 1)];
        }
        if (ok) {
          L.mc_ctab = 0;
          for (int q = 0; q < 12; ++q) L.mc_ctab |= tab[q] << (2 * q);
          if (s->opt.keep_residual && L.n >= s->patch_min_rows) HIP_TRY(L.mc_aux.alloc(sizeof(double) * L.n));
        }
      }
      if (L.n_colors <= 15 && !(L.mc_ctab >= 0 && L.n >= s->patch_min_rows)) {  // K-Strip's colour bytes
        std::vector<uint8_t> c8((size_t)L.n);
        for (int64_t i = 0; i < L.n; ++i) c8[(size_t)i] = (uint8_t)L.color[(size_t)i];
        HIP_TRY(upload(L.mc_color8, c8.data(), c8.size()));
      }
      ColorPerm CP;
      build_color_perm(L.A_csc, L.color, L.n_colors, &CP);  // column-as-row walk, like SpGS
      if (CP.rows.n_outer >= ((int64_t)1 << 31) - 512)
        return fail(AMG_HIP_EUNSUPPORTED, "multicolour smoother: level too large for int32 rows");
      DictMat T;
      L.mc_dict = (s->opt.layout == AMG_HIP_LAYOUT_AUTO || s->opt.layout == AMG_HIP_LAYOUT_DICT) &&
                  to_dict(CP.rows, 0, &T, CP.rowid.data());
      if (L.mc_dict) {
        L.mc_words = T.words;
        L.mc_wmax = T.max_width;
        L.mc_ntab = (int)T.doff.size();
        HIP_TRY(upload(L.mc_codes, T.codes.data(), T.codes.size()));
        HIP_TRY(upload(L.mc_doff, T.doff.data(), T.doff.size()));
        HIP_TRY(upload(L.mc_dval, T.dval.data(), T.dval.size()));
      } else {
        HIP_TRY(upload_mat(CP.rows, AMG_HIP_LAYOUT_SELL, &L.mc_mat, 0, false));
      }
      HIP_TRY(upload(L.mc_rowid, CP.rowid.data(), CP.rowid.size()));
      L.mc_start = CP.start;
      if (L.mc_dict && L.n <= mc_one_launch_max_rows()) {
        std::vector<int32_t> st32(CP.start.begin(), CP.start.end());
        HIP_TRY(upload(L.mc_start_dev, st32.data(), st32.size()));
      }
    }
    }  // dev
    if (!dev && s->opt.smoother == AMG_HIP_SM_MULTICOLOR_GS)
      greedy_coloring(L.A_csc, &L.color, &L.n_colors);
    timer.lap(T_SM);
    if (l + 1 == n_levels) break;
    // ---- transfer operators for level l -> l+1 ----
    const int64_t n_h = L.n;
    int64_t n_H = coarse_dofs(n_h);  // multigrid.hpp:214
    if (rs) {
      // host_setup.hpp: ruge_stueben_P.  The hierarchy ends where a level is small enough for
      // the direct solve or no longer coarsens.
      bool last = n_h <= rs_min_coarse;
      if (!last) {
        L.P_csc = ruge_stueben_P(A_r, rs_theta);
        n_H = L.P_csc.n_outer;
        last = n_H < 1 || n_H >= n_h;
      }
      if (last) {
        L.P_csc = Sparse();
        n_levels = l + 1;
        s->lv.resize((size_t)n_levels);
        break;
      }
      L.R_csc = transpose(L.P_csc);
      L.linear = false;
    } else if (n_H < 1) {
      return fail(AMG_HIP_EINVAL, "level " + std::to_string(l + 1) +
                                      " would have no degrees of freedom; reduce `n_levels`");
    } else if (Pc) {
      L.P_csc = from_raw(n_H, n_h, Pc[l], Pr[l], Pv[l]);
      L.R_csc = from_raw(n_h, n_H, Rc[l], Rr[l], Rv[l]);
      std::string v = validate(L.P_csc, "P");
      if (v.empty()) v = validate(L.R_csc, "R");
      if (!v.empty()) return fail(AMG_HIP_EINVAL, v);
      L.linear = is_linear_P(L.P_csc, n_h, n_H) && same_arrays(transpose(L.P_csc), L.R_csc);
    } else {
      L.lazy_linear = true;
      L.n_coarse = n_H;
      L.linear = true;
    }
    // the matrix-free kernels need no device copy of the linear operators, and the
    // device Galerkin product no host copy
    const bool dev_galerkin = dev && L.linear && !s->opt.host_galerkin;
    const bool dev_rows = dev && !(L.linear && s->opt.stencil_transfers);
    Sparse P_r, R_r;
    if (dev_rows || !dev_galerkin) {
      P_r = transpose(L.P());  // CSR(P)
      R_r = transpose(L.R());  // CSR(R)
    }
    if (dev_rows) {
      HIP_TRY(upload_csr(P_r, &L.P_rows));
      HIP_TRY(upload_csr(R_r, &L.R_rows));
    }
    timer.lap(T_TR);
    // Galerkin (multigrid.hpp:219-223), row-major, Eigen's summation order: on the device
    // for the linear interpolation pair (K-Galerkin), else on the host (same bits)
    Sparse AH_r;
    bool done = false;
    if (dev_galerkin) {
      if (!galerkin_on_dev) HIP_TRY(upload_csr(A_r, &galerkin_A));
      DevCsr next;
      HIP_TRY(device_galerkin(galerkin_A, n_H, &next, &AH_r));
      galerkin_A = std::move(next);
      galerkin_on_dev = true;
      done = true;
    } else if (dev && dev_rows && !s->opt.host_galerkin) {
      // custom interpolator: general K-way-merge product on the device (CSR(P), CSR(R) were
      // uploaded for the transfer kernels anyway)
      if (!galerkin_on_dev) HIP_TRY(upload_csr(A_r, &galerkin_A));
      DevCsr next;
      HIP_TRY(device_galerkin_generic(galerkin_A, L.P_rows, L.R_rows, &next, &AH_r, &done));
      if (done) {
        galerkin_A = std::move(next);
        galerkin_on_dev = true;
      }
    }
    if (!done) {
      AH_r = galerkin_csr(R_r, A_r, P_r, nt);
      galerkin_on_dev = false;
    }
    timer.lap(T_RAP);
    Level& C = s->lv[l + 1];
    C.n = n_H;
    C.A_csc = transpose(AH_r);
    timer.lap(T_CSC);
    A_r.ptr.swap(AH_r.ptr);
    A_r.idx.swap(AH_r.idx);
    A_r.val.swap(AH_r.val);
    A_r.n_outer = A_r.n_inner = n_H;
  }
  if (!dev) {
    compute_bytes(s.get());
    *out = s.release();
    return AMG_HIP_OK;
  }
  // rhs / initial state (multigrid.hpp:196-204)
  HIP_TRY(hipMemcpy(s->lv[0].f.p, b, sizeof(double) * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->lv[0].r.p, b, sizeof(double) * n, hipMemcpyHostToDevice));  // b - A*0
  // ---- coarsest factor (multigrid.hpp:240-243) ----
  if (!s->opt.window) {
    const int want = s->opt.fast_coarse_solve ? 1 : (s->opt.exact_coarse_solve ? -1 : 0);
    amg_hip_status r = upload_coarse(s->lv[n_levels - 1].A_csc, want, &s->coarse);
    if (r != AMG_HIP_OK) return r;
  }
  timer.lap(T_BAND);
  HIP_TRY(s->scratch.alloc(sizeof(double) * 1100));
  {
    amg_hip_status r = set_patch_coarse_flags(s.get());
    if (r != AMG_HIP_OK) return r;
  }
  compute_bytes(s.get());
  HIP_TRY(hipDeviceSynchronize());
  *out = s.release();
  return AMG_HIP_OK;
}

// ---- setup on the device end to end (amg_hip_create_poisson) -------------------
// Dictionary encoding of CSR(A) that sits on the device (K-Setup: dict_encode_kernel,
// dict_type_kernel).  The pair table and the word table are PROPOSED by the host from a
// sample of rows (first and last 64 K) and VERIFIED by the kernels on every row; rows that
// do not encode are reported back, their pairs / words are added, and the pass is repeated.
// Returns false (D untouched in the sense of dict = false) when the matrix does not qualify.
struct HostRows {  // rows [r0, r1) of a device CSR matrix
  int64_t r0 = 0;
  std::vector<int32_t> ptr, col;
  std::vector<double> val;
};
hipError_t fetch_rows(const DevCsr& A, int64_t r0, int64_t r1, HostRows* R) {
  R->r0 = r0;
  R->ptr.resize((size_t)(r1 - r0 + 1));
  hipError_t e = hipMemcpy(R->ptr.data(), A.rowptr() + r0, sizeof(int32_t) * R->ptr.size(), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return e;
  const int64_t p0 = R->ptr.front(), p1 = R->ptr.back();
  R->col.resize((size_t)(p1 - p0));
  R->val.resize((size_t)(p1 - p0));
  if (p1 > p0) {
    if ((e = hipMemcpy(R->col.data(), A.col() + p0, sizeof(int32_t) * (p1 - p0), hipMemcpyDeviceToHost)) != hipSuccess) return e;
    if ((e = hipMemcpy(R->val.data(), A.v() + p0, sizeof(double) * (p1 - p0), hipMemcpyDeviceToHost)) != hipSuccess) return e;
  }
  return hipSuccess;
}
hipError_t device_dict_encode(const DevCsr& A, bool prune, int maxlen, DevMat* D, bool* ok) {
  *ok = false;
  const int64_t n = A.n_rows;
  if (maxlen > 16 || n >= ((int64_t)1 << 28) || n < 1) return hipSuccess;
  hipError_t e;
  DictMat T;
  T.n = n;
  T.max_width = maxlen;
  T.words = maxlen > 8 ? 2 : 1;
  const int nw = T.words;
  std::map<std::pair<int32_t, uint64_t>, int> pairs;
  auto add_pairs = [&](const HostRows& R) {
    for (size_t r = 0; r + 1 < R.ptr.size(); ++r)
      for (int32_t p = R.ptr[r] - R.ptr[0]; p < R.ptr[r + 1] - R.ptr[0]; ++p) {
        if (prune && R.val[p] == 0.0) continue;
        uint64_t bits;
        std::memcpy(&bits, &R.val[p], 8);
        const auto key = std::make_pair((int32_t)((int64_t)R.col[p] - (R.r0 + (int64_t)r)), bits);
        if (pairs.emplace(key, (int)T.doff.size()).second) {
          T.doff.push_back(key.first);
          T.dval.push_back(R.val[p]);
        }
      }
  };
  const int64_t S = std::min<int64_t>(n, 65536);
  HostRows R;
  if ((e = fetch_rows(A, 0, S, &R)) != hipSuccess) return e;
  add_pairs(R);
  if (n > S) {
    if ((e = fetch_rows(A, n - S, n, &R)) != hipSuccess) return e;
    add_pairs(R);
  }
  DevMem dfail, dcodes, doff, dval;
  int32_t fail[64];
  if ((e = dfail.alloc(sizeof(fail))) != hipSuccess) return e;
  if ((e = dcodes.alloc(sizeof(uint64_t) * (size_t)n * nw)) != hipSuccess) return e;
  for (int round = 0;; ++round) {
    if (T.doff.size() > 255 || round > 64) return hipSuccess;  // does not qualify
    if ((e = upload(doff, T.doff.data(), T.doff.size())) != hipSuccess) return e;
    if ((e = upload(dval, T.dval.data(), T.dval.size())) != hipSuccess) return e;
    if ((e = hipMemset(dfail.p, 0, sizeof(fail))) != hipSuccess) return e;
    if ((e = launch_dict_encode(n, A.rowptr(), A.col(), A.v(), prune, doff.as<int32_t>(),
                                dval.as<double>(), (int)T.doff.size(), nw, dcodes.as<uint64_t>(),
                                dfail.as<int32_t>(), nullptr)) != hipSuccess)
      return e;
    if ((e = hipMemcpy(fail, dfail.p, sizeof(fail), hipMemcpyDeviceToHost)) != hipSuccess) return e;
    if (fail[0] == 0) break;
    const size_t before = T.doff.size();
    for (int k = 0; k < std::min(fail[0], 62); ++k) {
      if ((e = fetch_rows(A, fail[1 + k], (int64_t)fail[1 + k] + 1, &R)) != hipSuccess) return e;
      add_pairs(R);
    }
    if (T.doff.size() == before) return hipSuccess;  // a row longer than the code words hold
  }
  // second level: row types
  T.rwords.assign((size_t)256 * nw, ~(uint64_t)0);
  std::map<std::pair<uint64_t, uint64_t>, int> types;
  auto add_words = [&](const std::vector<uint64_t>& cw) {
    for (size_t r = 0; r * nw < cw.size(); ++r) {
      const std::pair<uint64_t, uint64_t> key(cw[r * nw], nw == 2 ? cw[r * 2 + 1] : ~(uint64_t)0);
      if (key.first == ~(uint64_t)0 && key.second == ~(uint64_t)0) continue;  // empty row: type 255
      if (types.size() < 256 && types.emplace(key, (int)types.size()).second && types.size() <= 255) {
        const int ty = (int)types.size() - 1;
        T.rwords[(size_t)ty * nw] = key.first;
        if (nw == 2) T.rwords[(size_t)ty * 2 + 1] = key.second;
      }
    }
  };
  auto fetch_codes = [&](int64_t r0, int64_t r1, std::vector<uint64_t>* cw) -> hipError_t {
    cw->resize((size_t)(r1 - r0) * nw);
    return hipMemcpy(cw->data(), dcodes.as<uint64_t>() + r0 * nw, sizeof(uint64_t) * cw->size(), hipMemcpyDeviceToHost);
  };
  bool typed = g_row_types != 0;
  DevMem drtype, drwords;
  if (typed) {
    std::vector<uint64_t> cw;
    if ((e = fetch_codes(0, S, &cw)) != hipSuccess) return e;
    add_words(cw);
    if (n > S) {
      if ((e = fetch_codes(n - S, n, &cw)) != hipSuccess) return e;
      add_words(cw);
    }
    if ((e = drtype.alloc((size_t)n)) != hipSuccess) return e;
    for (int round = 0; typed; ++round) {
      if (types.size() > 255 || round > 64) { typed = false; break; }
      if ((e = upload(drwords, T.rwords.data(), T.rwords.size())) != hipSuccess) return e;
      if ((e = hipMemset(dfail.p, 0, sizeof(fail))) != hipSuccess) return e;
      if ((e = launch_dict_types(n, dcodes.as<uint64_t>(), nw, drwords.as<uint64_t>(), (int)types.size(),
                                 drtype.as<uint8_t>(), dfail.as<int32_t>(), nullptr)) != hipSuccess)
        return e;
      if ((e = hipMemcpy(fail, dfail.p, sizeof(fail), hipMemcpyDeviceToHost)) != hipSuccess) return e;
      if (fail[0] == 0) break;
      for (int k = 0; k < std::min(fail[0], 62); ++k) {
        if ((e = fetch_codes(fail[1 + k], (int64_t)fail[1 + k] + 1, &cw)) != hipSuccess) return e;
        add_words(cw);
      }
    }
  }
  D->n_rows = n;
  D->nnz = A.nnz;
  D->dict_typed = typed;
  if (typed) {
    D->drtype = std::move(drtype);
  } else {
    T.rwords.clear();
    D->dcodes = std::move(dcodes);
  }
  if ((e = finish_dict(T, n, 0, D)) != hipSuccess) return e;
  *ok = true;
  return hipSuccess;
}

void rhs_threads(int dim, int64_t n, double* b, int nt, int64_t d0 = 0, int64_t d1 = -1);  // below

// AMG::Multigrid's constructor (multigrid.hpp:151-244) for A = Grid::laplacian(n), b =
// Grid::rhs(n) without host matrices: generator, Galerkin chain, dictionary encoder, diagonal
// and symmetry check all run on the device; only b (n^dim exp() calls, kept on the host so that
// the bits are libm's, like the reference's) and the coarsest operator (factored on the host)
// cross PCIe.  *unsupported = true: the options need host structures (exact lexicographic
// schedules, multicolouring, non-dictionary layouts): the caller takes the host path instead.
// [unit0, unit1): the grid lines (2-D) / x-y planes (3-D) of a window solver, else unit1 < 0.
amg_hip_status build_poisson_device(int dim, int64_t n, int32_t n_levels, const amg_hip_options* opts,
                                    amg_hip_solver** out, bool* unsupported, int64_t unit0 = 0,
                                    int64_t unit1 = -1) {
  *unsupported = true;
  amg_hip_options o;
  if (opts) o = *opts;
  else amg_hip_default_options(&o);
  if (unit1 < 0) { unit0 = 0; unit1 = n; }
  const int64_t n_last = unit1 - unit0;
  const int64_t unit_rows = dim == 3 ? n * n : n;
  const int64_t N = unit_rows * n_last;
  const bool lex = o.smoother <= AMG_HIP_SM_SOR;
  if (o.host_only || o.host_galerkin || !o.stencil_transfers || o.fuse_prolong ||
      (o.layout != AMG_HIP_LAYOUT_AUTO && o.layout != AMG_HIP_LAYOUT_DICT) ||
      o.smoother == AMG_HIP_SM_MULTICOLOR_GS || (lex && (o.exact_gs || N <= GS_SCAN_MIN_ROWS)) ||
      n_levels < 2 || N >= ((int64_t)1 << 28))
    return AMG_HIP_OK;
  if (o.smoother_iters < 0 || (o.smoother == AMG_HIP_SM_SOR && (o.omega > 2 || o.omega < 0)))
    return AMG_HIP_OK;  // the host path words the argument error
  std::unique_ptr<amg_hip_solver> s(new amg_hip_solver);
  s->opt = o;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(AMG_HIP_EHIP, "no HIP device available (this library has no CPU fallback)");
  if (o.device >= 0) {
    if (o.device >= ndev) return fail(AMG_HIP_EINVAL, "device ordinal out of range");
    s->device = o.device;
  } else {
    HIP_TRY(hipGetDevice(&s->device));
  }
  HIP_TRY(hipSetDevice(s->device));
  if (o.stream) {
    s->stream = (hipStream_t)o.stream;
    s->own_stream = false;
  } else {
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  }
  const bool timing = std::getenv("AMG_HIP_TIMING") != nullptr;
  auto t_start = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    (void)hipDeviceSynchronize();
    const auto t1 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "amg_hip device setup: %s %.3fs\n", what, std::chrono::duration<double>(t1 - t_start).count());
    t_start = t1;
  };
  // b on the host threads while the device builds the hierarchy
  std::vector<double> b((size_t)N);
  std::thread rhs_thread([&] { rhs_threads(dim, n, b.data(), host_threads(), unit0 * unit_rows, unit1 * unit_rows); });
  struct Joiner { std::thread& t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{rhs_thread};
  // A_0 (grid.hpp:88-98)
  DevCsr cur;
  {
    DevMem cnt, bsum, total;
    HIP_TRY(cnt.alloc(sizeof(int32_t) * (N + 1)));
    HIP_TRY(bsum.alloc(sizeof(int64_t) * ((N + 1023) / 1024 + 1)));
    HIP_TRY(total.alloc(sizeof(int64_t)));
    HIP_TRY(launch_laplacian_count(dim, n, n_last, N, cnt.as<int32_t>(), nullptr));
    HIP_TRY(cur.ptr.alloc(sizeof(int32_t) * (N + 1)));
    HIP_TRY(launch_exclusive_scan(N, cnt.as<int32_t>(), cur.ptr.as<int32_t>(), bsum.as<int64_t>(),
                                  total.as<int64_t>(), nullptr));
    int64_t nnz = 0;
    HIP_TRY(hipMemcpy(&nnz, total.p, sizeof(int64_t), hipMemcpyDeviceToHost));
    if (nnz >= ((int64_t)1 << 31) - 1) return AMG_HIP_OK;
    HIP_TRY(cur.idx.alloc(sizeof(int32_t) * nnz));
    HIP_TRY(cur.val.alloc(sizeof(double) * nnz));
    const double h = 2.0 / (double)(n + 1), hh = h * h;
    const double off = 1.0 / hh, dg = -2.0 / hh;
    double diag = dg + dg;
    if (dim == 3) diag = diag + dg;
    HIP_TRY(launch_laplacian_fill(dim, n, n_last, N, cur.ptr.as<int32_t>(), cur.idx.as<int32_t>(),
                                  cur.val.as<double>(), off, diag, nullptr));
    cur.n_rows = cur.n_cols = N;
    cur.nnz = nnz;
  }
  lap("generator");
  s->lv.resize(n_levels);
  DevMem stats;
  HIP_TRY(stats.alloc(sizeof(int32_t) * 2));
  const bool prune = !o.keep_structural_zeros;
  for (int l = 0; l < n_levels; ++l) {
    Level& L = s->lv[l];
    L.n = cur.n_rows;
    L.nnz_struct = cur.nnz;
    HIP_TRY(L.diag.alloc(sizeof(double) * L.n));
    HIP_TRY(hipMemset(stats.p, 0, sizeof(int32_t) * 2));
    HIP_TRY(launch_csr_inspect(L.n, cur.rowptr(), cur.col(), cur.v(), prune, stats.as<int32_t>(),
                               L.diag.as<double>(), nullptr));
    int32_t st[2];
    HIP_TRY(hipMemcpy(st, stats.p, sizeof(st), hipMemcpyDeviceToHost));
    L.symmetric = st[1] == 0;
    bool ok = false;
    if (!L.symmetric) {
      // Galerkin sums of a deep 3-D level can differ in the last bit between (i, j) and (j, i):
      // the smoothers then walk the columns of A (smoother.hpp:101-117), the residual its rows.
      // Both forms of this one level are made on the host (transpose) and uploaded; the chain
      // itself stays on the device.
      if (lex) return AMG_HIP_OK;
      if (timing) std::fprintf(stderr, "amg_hip device setup: level %d (%lld rows) not bitwise symmetric -> rows and columns uploaded separately\n",
                               l, (long long)L.n);
      Sparse H;
      H.n_outer = H.n_inner = L.n;
      H.ptr.resize((size_t)L.n + 1);
      H.idx.resize((size_t)cur.nnz);
      H.val.resize((size_t)cur.nnz);
      HIP_TRY(hipMemcpy(H.ptr.data(), cur.ptr.p, sizeof(int32_t) * H.ptr.size(), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(H.idx.data(), cur.idx.p, sizeof(int32_t) * H.idx.size(), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(H.val.data(), cur.val.p, sizeof(double) * H.val.size(), hipMemcpyDeviceToHost));
      HIP_TRY(upload_mat_pruned(H, o.layout, prune, &L.A_rows));       // H = CSR(A)
      L.A_csc = transpose(H);                                           // CSC(A)
      HIP_TRY(upload_mat_pruned(L.A_csc, o.layout, prune, &L.A_cols_own));
      ok = true;
    } else {
      HIP_TRY(device_dict_encode(cur, prune, st[0], &L.A_rows, &ok));
    }
    if (!ok) {
      // this level does not qualify for the dictionary (rows of more than 16 entries on the
      // deep 3-D levels, more than 255 pairs): its CSR arrays go to the host once and take
      // the general upload (SELL-64 / CSR panels); the chain itself stays on the device
      if (timing) std::fprintf(stderr, "amg_hip device setup: level %d (%lld rows, longest row %d) -> panel layout\n",
                               l, (long long)L.n, (int)st[0]);
      L.A_dev.n_rows = 0;
      Sparse H;
      H.n_outer = H.n_inner = L.n;
      H.ptr.resize((size_t)L.n + 1);
      H.idx.resize((size_t)cur.nnz);
      H.val.resize((size_t)cur.nnz);
      HIP_TRY(hipMemcpy(H.ptr.data(), cur.ptr.p, sizeof(int32_t) * H.ptr.size(), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(H.idx.data(), cur.idx.p, sizeof(int32_t) * H.idx.size(), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(H.val.data(), cur.val.p, sizeof(double) * H.val.size(), hipMemcpyDeviceToHost));
      HIP_TRY(upload_mat_pruned(H, AMG_HIP_LAYOUT_SELL, prune, &L.A_rows));
      L.A_csc = std::move(H);  // symmetric: the CSR arrays are the CSC arrays
    }
    if (o.smoother != AMG_HIP_SM_JACOBI) L.diag.release();
    HIP_TRY(L.u.alloc(sizeof(double) * L.n));
    HIP_TRY(L.f.alloc(sizeof(double) * L.n));
    HIP_TRY(L.r.alloc(sizeof(double) * L.n));
    HIP_TRY(L.tmp.alloc(sizeof(double) * L.n));
    HIP_TRY(hipMemset(L.u.p, 0, sizeof(double) * L.n));
    HIP_TRY(hipMemset(L.f.p, 0, sizeof(double) * L.n));
    HIP_TRY(hipMemset(L.r.p, 0, sizeof(double) * L.n));
    if (lex && (l + 1 < n_levels || o.keep_residual)) {  // K-GS-scan on every smoothed level, or the host path
      const DevMat& A = L.A_rows;
      int ring = 128;
      int C = (int)std::min<int64_t>(A.scan_gap, gs_scan_long_chunks_ok(A.dict_ref()) ? gs_scan_max_chunk() : 1024);
      if (C > 1024 && A.scan_far + C + 1 > 16384) C = 1024;  // the ring of new values must fit the LDS
      while (ring < A.scan_far + C + 1 && ring <= 16384) ring *= 2;
      if (A.scan_gap < GS_SCAN_MIN_GAP || ring > 16384) return AMG_HIP_OK;
      L.scan_C = C;
      L.scan_ring = ring;
    }
    if (l + 1 == n_levels) {
      if (L.A_csc.ptr.empty()) L.A_dev = std::move(cur);
      break;
    }
    const int64_t n_H = coarse_dofs(L.n);  // multigrid.hpp:214
    if (n_H < 1)
      return fail(AMG_HIP_EINVAL, "level " + std::to_string(l + 1) +
                                      " would have no degrees of freedom; reduce `n_levels`");
    L.lazy_linear = true;
    L.n_coarse = n_H;
    L.linear = true;
    DevCsr next;
    {
      const hipError_t ge = device_galerkin(cur, n_H, &next, nullptr);
      if (ge == hipErrorInvalidValue) {  // an intermediate product beyond int32 indices: host path
        if (timing) std::fprintf(stderr, "amg_hip device setup: Galerkin product of level %d exceeds int32 indexing -> host path\n", l);
        return AMG_HIP_OK;
      }
      HIP_TRY(ge);
    }
    if (L.A_csc.ptr.empty()) L.A_dev = std::move(cur);
    cur = std::move(next);
  }
  lap("hierarchy");
  joiner.t.join();
  HIP_TRY(hipMemcpy(s->lv[0].f.p, b.data(), sizeof(double) * N, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->lv[0].r.p, b.data(), sizeof(double) * N, hipMemcpyHostToDevice));  // b - A*0
  lap("rhs");
  if (!o.window) {
    const int want = o.fast_coarse_solve ? 1 : (o.exact_coarse_solve ? -1 : 0);
    HIP_TRY(s->lv[n_levels - 1].ensure_host_matrix());
    amg_hip_status r = upload_coarse(s->lv[n_levels - 1].A_csc, want, &s->coarse);
    if (r != AMG_HIP_OK) return r;
  }
  lap("coarse factor");
  HIP_TRY(s->scratch.alloc(sizeof(double) * 1100));
  {
    amg_hip_status r = set_patch_coarse_flags(s.get());
    if (r != AMG_HIP_OK) return r;
  }
  compute_bytes(s.get());
  HIP_TRY(hipDeviceSynchronize());
  *unsupported = false;
  *out = s.release();
  return AMG_HIP_OK;
}

// ---- helpers for the stand-alone host-array entry points ---------------------
struct Scoped {
  hipStream_t st = nullptr;
  ~Scoped() { if (st) (void)hipStreamDestroy(st); }
};

// Grid::rhs on several host threads (the values are libm's exp(), as in the reference);
// entries [d0, d1) of the right-hand side into b[0 .. d1 - d0) (d1 < 0: all)
void rhs_threads(int dim, int64_t n, double* b, int nt, int64_t d0, int64_t d1) {
  int64_t N = n * n;
  if (dim == 3) N *= n;
  if (d1 < 0) { d0 = 0; d1 = N; }
  const int64_t M = d1 - d0;
  if (nt <= 1 || M < (1 << 18)) {
    rhs_range(dim, n, b - d0, d0, d1);
    return;
  }
  std::vector<std::thread> th;
  for (int t = 0; t < nt; ++t)
    th.emplace_back([=] { rhs_range(dim, n, b - d0, d0 + M * t / nt, d0 + M * (t + 1) / nt); });
  for (auto& x : th) x.join();
}

amg_hip_status need_device() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(AMG_HIP_EHIP, "no HIP device available (this library has no CPU fallback)");
  return AMG_HIP_OK;
}

}  // namespace

namespace amg_hip {
// for the other translation units of the library (comm.cpp): sets amg_hip_last_error
amg_hip_status fail_status(amg_hip_status st, const std::string& msg) { return fail(st, msg); }
}  // namespace amg_hip

// =============================================================================
extern "C" {

const char* amg_hip_last_error(void) { return g_err.c_str(); }

void amg_hip_default_options(amg_hip_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->smoother = AMG_HIP_SM_SPGS;
  o->smoother_iters = 1;
  o->omega = 1.0;
  o->device = -1;
  o->use_graph = 1;
  o->stencil_transfers = 1;
  o->layout = g_default_layout;
  o->keep_structural_zeros = 0;
  o->no_fusion = 0;
  o->fuse_prolong = 0;
  o->fast_coarse_solve = 0;
  o->stream = nullptr;
  o->window = 0;
  o->reserved0 = 0;
}

void amg_hip_set_index16(int32_t on) { g_index16 = on ? 1 : 0; }
void amg_hip_set_nontemporal(int32_t on) { g_nontemporal = on ? 1 : 0; }
void amg_hip_set_xcd_mapping(int32_t on) { set_xcd_mapping(on); }
void amg_hip_set_row_types(int32_t on) { g_row_types = on ? 1 : 0; }
void amg_hip_set_dict_rows(int32_t rows_per_lane) { set_dict_rows_per_lane(rows_per_lane); }
void amg_hip_set_dict_stencil(int32_t on) { set_dict_stencil(on); }
void amg_hip_set_patch_tile_flags(int32_t on) { g_patch_tile_flags = on ? 1 : 0; }
void amg_hip_set_band_chain(int32_t on) { g_no_band_chain = on ? 0 : 1; }
void amg_hip_set_tail_fusion(int32_t on) { g_tail_fusion = on ? 1 : 0; }
void amg_hip_set_patch_min_rows(int64_t rows) { g_patch_min_rows = rows < 0 ? INT64_MAX : rows; }

void amg_hip_set_default_layout(int32_t layout) {
  if (layout >= AMG_HIP_LAYOUT_AUTO && layout <= AMG_HIP_LAYOUT_DICT) g_default_layout = layout;
}

int amg_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

amg_hip_status amg_hip_create(int64_t n, const int32_t* colptr, const int32_t* rowind,
                              const double* val, const double* b, int32_t n_levels,
                              const amg_hip_options* opts, amg_hip_solver** out) {
  return build_solver(n, colptr, rowind, val, b, n_levels, nullptr, nullptr, nullptr, nullptr,
                      nullptr, nullptr, opts, out);
}

amg_hip_status amg_hip_create_custom(int64_t n, const int32_t* colptr, const int32_t* rowind,
                                     const double* val, const double* b, int32_t n_levels,
                                     const int32_t* const* P_colptr,
                                     const int32_t* const* P_rowind, const double* const* P_val,
                                     const int32_t* const* R_colptr,
                                     const int32_t* const* R_rowind, const double* const* R_val,
                                     const amg_hip_options* opts, amg_hip_solver** out) {
  if (n_levels > 1 && (!P_colptr || !P_rowind || !P_val || !R_colptr || !R_rowind || !R_val))
    return fail(AMG_HIP_EINVAL, "custom transfer operator arrays are null");
  return build_solver(n, colptr, rowind, val, b, n_levels, P_colptr, P_rowind, P_val, R_colptr,
                      R_rowind, R_val, opts, out);
}

amg_hip_status amg_hip_create_rs(int64_t n, const int32_t* colptr, const int32_t* rowind,
                                 const double* val, const double* b, int32_t max_levels,
                                 double theta, int64_t min_coarse, const amg_hip_options* opts,
                                 amg_hip_solver** out) {
  if (!(theta >= 0.0 && theta <= 1.0)) return fail(AMG_HIP_EINVAL, "`theta` must be in [0, 1]");
  if (min_coarse < 1) return fail(AMG_HIP_EINVAL, "`min_coarse` must be at least 1");
  return build_solver(n, colptr, rowind, val, b, max_levels, nullptr, nullptr, nullptr, nullptr,
                      nullptr, nullptr, opts, out, theta, min_coarse);
}

amg_hip_status amg_hip_create_poisson(int32_t dim, int64_t n, int32_t n_levels,
                                      const amg_hip_options* opts, amg_hip_solver** out) {
  if (!out) return fail(AMG_HIP_EINVAL, "out handle pointer is null");
  *out = nullptr;
  if ((dim != 2 && dim != 3) || n < 1) return fail(AMG_HIP_EINVAL, "dim must be 2 or 3 and n >= 1");
  bool unsupported = true;
  amg_hip_status r = build_poisson_device(dim, n, n_levels, opts, out, &unsupported);
  if (r != AMG_HIP_OK || !unsupported) return r;
  // options that need host structures: generate on the host and take the general constructor
  Sparse A = laplacian(dim, n);
  std::vector<double> b((size_t)A.n_outer);
  rhs_threads(dim, n, b.data(), host_threads());
  return amg_hip_create(A.n_outer, A.ptr.data(), A.idx.data(), A.val.data(), b.data(), n_levels, opts, out);
}

amg_hip_status amg_hip_create_poisson_window(int32_t dim, int64_t n, int64_t unit_begin, int64_t unit_end,
                                             int32_t n_levels, const amg_hip_options* opts,
                                             amg_hip_solver** out) {
  if (!out) return fail(AMG_HIP_EINVAL, "out handle pointer is null");
  *out = nullptr;
  if ((dim != 2 && dim != 3) || n < 1) return fail(AMG_HIP_EINVAL, "dim must be 2 or 3 and n >= 1");
  if (unit_begin < 0 || unit_end > n || unit_end - unit_begin < 1)
    return fail(AMG_HIP_EINVAL, "window units must satisfy 0 <= unit_begin < unit_end <= n");
  const int64_t unit_rows = dim == 3 ? n * n : n;
  if ((unit_begin * unit_rows) & 1)
    return fail(AMG_HIP_EINVAL, "a window must begin at an even flat index (coarse dof j <-> fine dof 2j+1, multigrid.hpp:127-130)");
  if (n_levels < 2) return fail(AMG_HIP_EINVAL, "a window solver needs at least one distributed level (n_levels >= 2)");
  amg_hip_options o;
  if (opts) o = *opts;
  else amg_hip_default_options(&o);
  o.window = 1;
  bool unsupported = true;
  amg_hip_status r = build_poisson_device(dim, n, n_levels, &o, out, &unsupported, unit_begin, unit_end);
  if (r != AMG_HIP_OK || !unsupported) return r;
  // options that need host structures (multicolouring, ...): generate the window on the host
  Sparse A = laplacian(dim, n, unit_end - unit_begin);
  std::vector<double> b((size_t)A.n_outer);
  rhs_threads(dim, n, b.data(), host_threads(), unit_begin * unit_rows, unit_end * unit_rows);
  return amg_hip_create(A.n_outer, A.ptr.data(), A.idx.data(), A.val.data(), b.data(), n_levels, &o, out);
}

void amg_hip_destroy(amg_hip_solver* s) {
  if (!s) return;
  if (!s->opt.host_only) {
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
  }
  delete s;
}

amg_hip_status amg_hip_vcycles(amg_hip_solver* s, int32_t n) {
  if (!s) return fail(AMG_HIP_EINVAL, "null solver");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  if (s->opt.use_graph) {
    if ((r = ensure_graph(s)) != AMG_HIP_OK) return r;
    for (int i = 0; i < n; ++i) HIP_TRY(hipGraphLaunch(s->graph_exec, s->stream));
  } else {
    for (int i = 0; i < n; ++i)
      if ((r = enqueue_vcycle(s)) != AMG_HIP_OK) return r;
  }
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_vcycle(amg_hip_solver* s) { return amg_hip_vcycles(s, 1); }

// ---- slab sharding (struct Slab) ---------------------------------------------
namespace {
// a bigger allocation with the same leading contents
amg_hip_status grow(DevMem& m, size_t bytes, hipStream_t st) {
  if (m.bytes >= bytes) return AMG_HIP_OK;
  DevMem big;
  HIP_TRY(big.alloc(bytes));
  HIP_TRY(hipMemsetAsync(big.p, 0, bytes, st));
  if (m.bytes) HIP_TRY(hipMemcpyAsync(big.p, m.p, m.bytes, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  m = std::move(big);
  return AMG_HIP_OK;
}
}  // namespace

amg_hip_status amg_hip_slab_plan(int64_t lines, int32_t rank, int32_t world, int32_t levels,
                                 amg_hip_slab_info* out) {
  if (!out || lines < 1 || world < 1 || rank < 0 || rank >= world || levels < 1 ||
      levels > AMG_HIP_SLAB_MAX_LEVELS)
    return fail(AMG_HIP_EINVAL, "amg_hip_slab_plan: bad argument");
  const int k = levels;
  std::memset(out, 0, sizeof(*out));
  const int64_t chunk = (lines + world - 1) / world;
  const int64_t line0 = std::min<int64_t>(lines, chunk * rank);
  const int64_t line1 = std::min<int64_t>(lines, chunk * (rank + 1));
  // Lines beyond the owned block on which each leg's INPUT has to be valid.  A Jacobi sweep or
  // a residual makes one line plus one entry (the corner couplings of the 9-point coarse
  // operators) at either end stale; the restriction and the prolongation reach two / one
  // entries across a line end.  Counted in whole lines:
  //   up-leg of level l (prolongation + two sweeps) has to leave u_l valid 3 l lines out (what
  //   level l-1 loads: two sweeps' worth plus the neighbouring coarse entry), so the smoothed
  //   u_l the down-leg left must be valid 3 l + 2 lines and 2 entries out: 3 l + 3 lines;
  //   down-leg of level l: input valid I_l lines out gives f_{l+1} valid I_l - sweeps - 2 lines
  //   out (sweeps, residual, and the restriction's entries: less than a line), where sweeps = 2
  //   on level 0 and 1 below it (the first sweep of a coarser level is done by the finer
  //   level's kernel); the gathered level needs f only on the owned lines.
  std::vector<int64_t> need(k + 1, 0);
  for (int l = k - 1; l >= 0; --l) {
    const int sw = l == 0 ? 2 : 1;
    const int64_t own = 3 * l + 3 + sw;
    const int64_t feed = (l + 1 < k ? need[l + 1] : 0) + 2 + sw;
    need[l] = std::max(own, feed);
  }
  const int64_t last = lines - chunk * (world - 1);
  if (world > 1 && (last < need[0] || chunk < need[0]))
    return fail(AMG_HIP_EUNSUPPORTED, "amg_hip_slab_plan: fewer grid lines per rank than the halo depth");
  auto lo = [&](int64_t h) { return world == 1 ? 0 : std::max<int64_t>(0, line0 - h); };
  auto hi = [&](int64_t h) { return world == 1 ? lines : std::min<int64_t>(lines, line1 + h); };
  for (int l = 0; l < k; ++l) {
    out->down_lo[l] = lo(need[l]);
    out->down_hi[l] = hi(need[l]);
    out->up_lo[l] = lo(3 * l);
    out->up_hi[l] = hi(3 * l);
  }
  out->levels = k;
  out->halo_lines = (int32_t)need[0];
  out->lines = lines;
  out->chunk_lines = chunk;
  out->line_begin = line0;
  out->line_end = line1;
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_slab_setup(amg_hip_solver* s, int32_t rank, int32_t world,
                                  int32_t max_levels, amg_hip_slab_info* info) {
  if (!s || !info || world < 1 || rank < 0 || rank >= world)
    return fail(AMG_HIP_EINVAL, "amg_hip_slab_setup: bad argument");
  if (s->opt.window) return fail(AMG_HIP_EINVAL, "amg_hip_slab_setup: a window solver is cut already (amg_hip_window_setup)");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  int k = 0;
  while (patch_level_ok(s, k)) ++k;
  if (max_levels >= 0 && k > max_levels) k = max_levels;
  if (k > AMG_HIP_SLAB_MAX_LEVELS) k = AMG_HIP_SLAB_MAX_LEVELS;
  if (k < 1)
    return fail(AMG_HIP_EUNSUPPORTED,
                "amg_hip_slab_setup: no K-Patch level (needs the 2+2 true-Jacobi cycle on a "
                "dictionary-coded 2-D hierarchy of at least patch_min_rows rows)");
  HIP_TRY(hipStreamSynchronize(s->stream));
  Slab& sb = s->slab;
  sb.reset_graphs();
  sb.levels = 0;  // configured again only when everything below succeeds
  const Level& L0 = s->lv[0];
  const int64_t m0 = L0.A_rows.patch_m;
  sb.lines = (L0.n + m0 - 1) / m0;
  for (int l = 0; l <= k; ++l) {  // the lines of the slab levels coincide (x-only coarsening)
    const Level& L = s->lv[l];
    const int64_t m = l < k ? L.A_rows.patch_m : s->lv[k - 1].A_rows.patch_m / 2;
    if (m != (m0 >> l) || (L.n + m - 1) / m != sb.lines)
      return fail(AMG_HIP_EUNSUPPORTED, "amg_hip_slab_setup: level lines do not coincide");
  }
  amg_hip_slab_info pl;
  if ((r = amg_hip_slab_plan(sb.lines, rank, world, k, &pl)) != AMG_HIP_OK) return r;
  sb.rank = rank;
  sb.world = world;
  sb.chunk = pl.chunk_lines;
  sb.line0 = pl.line_begin;
  sb.line1 = pl.line_end;
  sb.halo = pl.halo_lines;
  sb.down_lo.assign(pl.down_lo, pl.down_lo + k);
  sb.down_hi.assign(pl.down_hi, pl.down_hi + k);
  sb.up_lo.assign(pl.up_lo, pl.up_lo + k);
  sb.up_hi.assign(pl.up_hi, pl.up_hi + k);
  sb.levels = k;
  // in-place all-gathers need room for world equal blocks
  Level& G = s->lv[k];
  const int64_t mk = m0 >> k;
  if ((r = grow(G.f, sizeof(double) * (size_t)(sb.chunk * world * mk), s->stream)) != AMG_HIP_OK ||
      (r = grow(s->lv[0].u, sizeof(double) * (size_t)(sb.chunk * world * m0), s->stream)) != AMG_HIP_OK) {
    sb.levels = 0;  // not configured: amg_hip_slab_run refuses
    return r;
  }
  if (s->graph_ready) {  // the whole-cycle graph holds the old pointers
    (void)hipGraphExecDestroy(s->graph_exec);
    (void)hipGraphDestroy(s->graph);
    s->graph_exec = nullptr;
    s->graph = nullptr;
    s->graph_ready = false;
  }
  *info = pl;
  info->pitch0 = m0;
  info->gather_pitch = mk;
  info->gather_rows = G.n;
  info->u0 = s->lv[0].u.as<double>();
  info->f_gather = G.f.as<double>();
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_slab_run(amg_hip_solver* s, int32_t part) {
  if (!s || part < CYCLE_SLAB_DOWN || part > CYCLE_SLAB_UP)
    return fail(AMG_HIP_EINVAL, "amg_hip_slab_run: part is 1 (down-legs), 2 (replicated rest) or 3 (up-legs)");
  Slab& sb = s->slab;
  if (sb.levels < 1) return fail(AMG_HIP_EINVAL, "amg_hip_slab_run: amg_hip_slab_setup has not succeeded");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  if (!s->opt.use_graph) return enqueue_vcycle(s, part);
  const int gi = part - 1;
  if (!sb.exec[gi]) {
    HIP_TRY(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    r = enqueue_vcycle(s, part);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s->stream, &g);
    if (r != AMG_HIP_OK) {
      if (g) (void)hipGraphDestroy(g);
      return r;
    }
    if (e != hipSuccess) return fail(AMG_HIP_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    sb.graph[gi] = g;
    HIP_TRY(hipGraphInstantiate(&sb.exec[gi], g, nullptr, nullptr, 0));
  }
  HIP_TRY(hipGraphLaunch(sb.exec[gi], s->stream));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_window_setup(amg_hip_solver* s, const int64_t* down_lo, const int64_t* down_hi,
                                    const int64_t* up_lo, const int64_t* up_hi) {
  if (!s) return fail(AMG_HIP_EINVAL, "null solver");
  if (!s->opt.window) return fail(AMG_HIP_EINVAL, "amg_hip_window_setup: not a window solver (opt.window)");
  if ((down_lo || down_hi || up_lo || up_hi) && !(down_lo && down_hi && up_lo && up_hi))
    return fail(AMG_HIP_EINVAL, "amg_hip_window_setup: give all four range arrays or none");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  HIP_TRY(hipStreamSynchronize(s->stream));
  const int k = (int)s->lv.size() - 1;
  Slab& sb = s->slab;
  sb.reset_graphs();
  sb.levels = k;
  sb.rank = 0;
  sb.world = 1;
  sb.down_lo.assign(k, 0);
  sb.down_hi.assign(k, -1);  // -1: every line
  sb.up_lo.assign(k, 0);
  sb.up_hi.assign(k, -1);
  if (down_lo)
    for (int l = 0; l < k; ++l) {
      if (down_lo[l] < 0 || up_lo[l] < 0 || down_hi[l] < down_lo[l] || up_hi[l] < up_lo[l])
        return fail(AMG_HIP_EINVAL, "amg_hip_window_setup: bad line range");
      sb.down_lo[l] = down_lo[l];
      sb.down_hi[l] = down_hi[l];
      sb.up_lo[l] = up_lo[l];
      sb.up_hi[l] = up_hi[l];
    }
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_window_run(amg_hip_solver* s, int32_t part) {
  if (!s || (part != CYCLE_SLAB_DOWN && part != CYCLE_SLAB_UP))
    return fail(AMG_HIP_EINVAL, "amg_hip_window_run: part is 1 (down-legs) or 3 (up-legs)");
  if (!s->opt.window) return fail(AMG_HIP_EINVAL, "amg_hip_window_run: not a window solver (opt.window)");
  if (s->slab.levels < 1) {
    amg_hip_status r = amg_hip_window_setup(s, nullptr, nullptr, nullptr, nullptr);
    if (r != AMG_HIP_OK) return r;
  }
  return amg_hip_slab_run(s, part);
}

#ifdef AMG_PATCH_STAMPS
amg_hip_status amg_hip_debug_patch_stamps(void* dev_ptr) {  // diagnostic build only (tools/patch_stamps.py)
  HIP_TRY(debug_set_patch_stamps((unsigned long long*)dev_ptr));
  return AMG_HIP_OK;
}
#endif

amg_hip_status amg_hip_get_stream(amg_hip_solver* s, void** stream) {
  if (!s || !stream) return fail(AMG_HIP_EINVAL, "bad argument");
  if (s->opt.host_only) return fail(AMG_HIP_EINVAL, "solver was created with host_only = 1: no device state");
  *stream = (void*)s->stream;
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_vec_dev_ptr(amg_hip_solver* s, int32_t level, int32_t which, void** ptr, int64_t* n) {
  if (!s || !ptr || level < 0 || level >= (int32_t)s->lv.size() || which < 0 || which > 2)
    return fail(AMG_HIP_EINVAL, "bad argument");
  if (s->opt.host_only) return fail(AMG_HIP_EINVAL, "solver was created with host_only = 1: no device state");
  Level& L = s->lv[level];
  *ptr = which == 0 ? L.u.p : (which == 1 ? L.f.p : L.r.p);
  if (n) *n = L.n;
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_sync(amg_hip_solver* s) {
  if (!s) return fail(AMG_HIP_EINVAL, "null solver");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  HIP_TRY(hipStreamSynchronize(s->stream));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_copy_vec_dev(amg_hip_solver* s, int32_t level, int32_t which,
                                    void* dev_ptr, int32_t to_solver) {
  if (!s || !dev_ptr || level < 0 || level >= (int32_t)s->lv.size() || which < 0 || which > 2)
    return fail(AMG_HIP_EINVAL, "bad argument");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  Level& L = s->lv[level];
  void* v = which == 0 ? L.u.p : (which == 1 ? L.f.p : L.r.p);
  if (to_solver)
    HIP_TRY(hipMemcpyAsync(v, dev_ptr, sizeof(double) * L.n, hipMemcpyDeviceToDevice, s->stream));
  else
    HIP_TRY(hipMemcpyAsync(dev_ptr, v, sizeof(double) * L.n, hipMemcpyDeviceToDevice, s->stream));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_zero_vec(amg_hip_solver* s, int32_t level, int32_t which) {
  if (!s || level < 0 || level >= (int32_t)s->lv.size() || which < 0 || which > 2)
    return fail(AMG_HIP_EINVAL, "bad argument");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  Level& L = s->lv[level];
  void* v = which == 0 ? L.u.p : (which == 1 ? L.f.p : L.r.p);
  HIP_TRY(hipMemsetAsync(v, 0, sizeof(double) * L.n, s->stream));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_level_op(amg_hip_solver* s, int32_t level, int32_t op) {
  if (!s) return fail(AMG_HIP_EINVAL, "null solver");
  const int nl = (int)s->lv.size();
  if (level < 0 || level >= nl) return fail(AMG_HIP_EINVAL, "level out of range");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  hipStream_t st = s->stream;
  Level& L = s->lv[level];
  switch (op) {
    case 0: return enqueue_smooth(s, level, 0);
    case 1: return enqueue_residual(s, level);
    case 2: {
      if (level + 1 >= nl) return fail(AMG_HIP_EINVAL, "no coarser level");
      Level& C = s->lv[level + 1];
      if (L.linear && s->opt.stencil_transfers) {
        HIP_TRY(launch_linear_restrict(L.n, C.n, L.r.as<double>(), C.f.as<double>(), C.u.as<double>(), st));
      } else {
        HIP_TRY(hipMemsetAsync(C.u.p, 0, sizeof(double) * C.n, st));
        const DevCsr& R = L.R_rows;
        HIP_TRY(launch_csr(CSR_SPMV, R.n_rows, R.nnz, R.max_block_nnz, R.max_row_nnz, R.rowptr(),
                           R.col(), R.v(), L.r.as<double>(), nullptr, C.f.as<double>(), 1.0, 0, st));
      }
      return AMG_HIP_OK;
    }
    case 3: {
      if (level + 1 >= nl) return fail(AMG_HIP_EINVAL, "no coarser level");
      Level& C = s->lv[level + 1];
      if (L.linear && s->opt.stencil_transfers) {
        HIP_TRY(launch_linear_prolong_add(L.n, C.n, C.u.as<double>(), L.u.as<double>(), st));
      } else {
        const DevCsr& P = L.P_rows;
        HIP_TRY(launch_csr(CSR_SPMV_ADD, P.n_rows, P.nnz, P.max_block_nnz, P.max_row_nnz, P.rowptr(),
                           P.col(), P.v(), C.u.as<double>(), L.u.as<double>(), L.u.as<double>(), 1.0, 0, st));
      }
      return AMG_HIP_OK;
    }
    case 4:
      if (level != nl - 1) return fail(AMG_HIP_EINVAL, "the direct solve belongs to the coarsest level");
      HIP_TRY(launch_coarse(s->coarse, L.f.as<double>(), L.tmp.as<double>(), L.u.as<double>(), st));
      return AMG_HIP_OK;
  }
  return fail(AMG_HIP_EINVAL, "unknown level operation");
}

amg_hip_status amg_hip_rss(amg_hip_solver* s, double* out) {
  if (!s || !out) return fail(AMG_HIP_EINVAL, "null argument");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  Level& L = s->lv[0];
  HIP_TRY(launch_mat(CSR_RSSQ, L.A_rows, L.u.as<double>(), L.f.as<double>(), L.tmp.as<double>(),
                     1.0, s->stream));
  double* sc = s->scratch.as<double>();
  HIP_TRY(launch_sum(L.n, L.tmp.as<double>(), sc + 1024, sc, 0, s->stream));
  HIP_TRY(hipMemcpyAsync(out, sc + 1024, sizeof(double), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return AMG_HIP_OK;
}

// z = M^-1 v: one V-cycle from a zero guess on the right-hand side v (README.md:127, ref [7]
// of the reference: "a single V-cycle used for preconditioner").  Level 0's f and u are the
// cycle's operands, so they are parked in the PCG work vectors and put back.
static amg_hip_status pcg_alloc(amg_hip_solver* s) {
  const size_t bytes = sizeof(double) * (size_t)s->lv[0].n;
  if (s->pcg_x.p) return AMG_HIP_OK;
  HIP_TRY(s->pcg_x.alloc(bytes));
  HIP_TRY(s->pcg_p.alloc(bytes));
  HIP_TRY(s->pcg_q.alloc(bytes));
  HIP_TRY(s->pcg_b.alloc(bytes));
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_apply(amg_hip_solver* s, const double* v_dev, double* z_dev) {
  if (!s || !v_dev || !z_dev) return fail(AMG_HIP_EINVAL, "null argument");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  if ((r = pcg_alloc(s)) != AMG_HIP_OK) return r;
  Level& L = s->lv[0];
  const size_t bytes = sizeof(double) * (size_t)L.n;
  hipStream_t st = s->stream;
  HIP_TRY(hipMemcpyAsync(s->pcg_b.p, L.f.p, bytes, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(s->pcg_x.p, L.u.p, bytes, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(L.f.p, v_dev, bytes, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemsetAsync(L.u.p, 0, bytes, st));
  if ((r = amg_hip_vcycles(s, 1)) != AMG_HIP_OK) return r;
  HIP_TRY(hipMemcpyAsync(z_dev, L.u.p, bytes, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(L.f.p, s->pcg_b.p, bytes, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(L.u.p, s->pcg_x.p, bytes, hipMemcpyDeviceToDevice, st));
  return AMG_HIP_OK;
}

// Preconditioned conjugate gradients on A_0 x = b, M^-1 = one V-cycle from zero.  During the
// iteration level 0's f holds the residual r and level 0's u receives z = M^-1 r (no copies
// around the cycle); x, p, q = A p live in the work vectors.  Start: x = the level-0
// solution; on return it holds the result and f is b again.
amg_hip_status amg_hip_pcg(amg_hip_solver* s, double rtol, int64_t max_iters, int64_t* iters,
                           double* relres) {
  if (!s) return fail(AMG_HIP_EINVAL, "null solver");
  if (!(rtol >= 0) || max_iters < 0) return fail(AMG_HIP_EINVAL, "bad tolerance / iteration limit");
  amg_hip_status st0 = set_device(s);
  if (st0 != AMG_HIP_OK) return st0;
  if ((st0 = pcg_alloc(s)) != AMG_HIP_OK) return st0;
  Level& L = s->lv[0];
  const int64_t n = L.n;
  const size_t bytes = sizeof(double) * (size_t)n;
  hipStream_t st = s->stream;
  double* x = s->pcg_x.as<double>();
  double* p = s->pcg_p.as<double>();
  double* q = s->pcg_q.as<double>();
  double* bsv = s->pcg_b.as<double>();
  double* r = L.f.as<double>();   // residual lives where the cycle expects its right-hand side
  double* z = L.u.as<double>();   // and the cycle leaves M^-1 r here
  double* sc = s->scratch.as<double>();
  double* part = sc;              // 1024 partials
  double* d_rz = sc + 1030;       // device scalars: r.z, p.q, new r.z, r.r
  double* d_pq = sc + 1031;
  double* d_rzn = sc + 1032;
  double* d_rr = sc + 1033;
  double h[2];
  // b . b
  HIP_TRY(launch_dot(n, r, r, d_rr, part, st));
  HIP_TRY(hipMemcpyAsync(h, d_rr, sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  const double bnorm = std::sqrt(h[0]);
  HIP_TRY(hipMemcpyAsync(bsv, r, bytes, hipMemcpyDeviceToDevice, st));   // park b
  HIP_TRY(hipMemcpyAsync(x, z, bytes, hipMemcpyDeviceToDevice, st));     // x = u_0
  // r = b - A x (multigrid.hpp:272-274 arithmetic), in place of b
  HIP_TRY(launch_mat(CSR_RESID, L.A_rows, x, bsv, r, 1.0, st));
  int64_t it = 0;
  double rel = 0.0;
  auto finish = [&]() -> amg_hip_status {
    HIP_TRY(hipMemcpyAsync(L.f.p, bsv, bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(L.u.p, x, bytes, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (iters) *iters = it;
    if (relres) *relres = rel;
    return AMG_HIP_OK;
  };
  HIP_TRY(launch_dot(n, r, r, d_rr, part, st));
  HIP_TRY(hipMemcpyAsync(h, d_rr, sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  rel = bnorm > 0 ? std::sqrt(h[0]) / bnorm : std::sqrt(h[0]);
  if (rel <= rtol || max_iters == 0) return finish();
  // z = M^-1 r; p = z; rz = r . z
  HIP_TRY(hipMemsetAsync(z, 0, bytes, st));
  amg_hip_status rc = amg_hip_vcycles(s, 1);
  if (rc != AMG_HIP_OK) return rc;
  HIP_TRY(hipMemcpyAsync(p, z, bytes, hipMemcpyDeviceToDevice, st));
  HIP_TRY(launch_dot(n, r, z, d_rz, part, st));
  while (it < max_iters) {
    HIP_TRY(launch_mat(CSR_SPMV, L.A_rows, p, nullptr, q, 1.0, st));     // q = A p
    HIP_TRY(launch_dot(n, p, q, d_pq, part, st));
    HIP_TRY(launch_pcg_update_xr(n, d_rz, d_pq, x, r, p, q, st));        // alpha = rz / pq
    HIP_TRY(launch_dot(n, r, r, d_rr, part, st));
    HIP_TRY(hipMemcpyAsync(h, d_rr, sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    it += 1;
    rel = bnorm > 0 ? std::sqrt(h[0]) / bnorm : std::sqrt(h[0]);
    if (!(rel > rtol) || it >= max_iters) break;                         // NaN leaves too
    HIP_TRY(hipMemsetAsync(z, 0, bytes, st));
    if ((rc = amg_hip_vcycles(s, 1)) != AMG_HIP_OK) return rc;           // z = M^-1 r
    HIP_TRY(launch_dot(n, r, z, d_rzn, part, st));
    HIP_TRY(launch_pcg_update_p(n, d_rzn, d_rz, p, z, st));              // beta = rz_new / rz
    HIP_TRY(hipMemcpyAsync(d_rz, d_rzn, sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  return finish();
}

amg_hip_status amg_hip_solve(amg_hip_solver* s, double tol, int64_t every, int64_t n_iters,
                             int64_t* iters, double* last_rss, int32_t* converged) {
  if (!s) return fail(AMG_HIP_EINVAL, "null solver");
  if (every > n_iters)  // multigrid.hpp:165-171
    return fail(AMG_HIP_EINVAL, "`compute_error_every_n_iters` must be leq to `n_iters`, got " +
                                    std::to_string(every) + " and " + std::to_string(n_iters));
  if (every <= 0) return fail(AMG_HIP_EINVAL, "`compute_error_every_n_iters` must be positive");
  int64_t iter = 0;
  double error = 100;  // multigrid.hpp:313
  while (iter < n_iters && error > tol) {
    amg_hip_status r = amg_hip_vcycles(s, 1);
    if (r != AMG_HIP_OK) return r;
    iter += 1;
    if ((iter % every) == 0) {
      if ((r = amg_hip_rss(s, &error)) != AMG_HIP_OK) return r;
    }
  }
  HIP_TRY(hipStreamSynchronize(s->stream));
  if (iters) *iters = iter;
  if (last_rss) *last_rss = error;
  if (converged) *converged = error <= tol;
  return AMG_HIP_OK;
}

int32_t amg_hip_n_levels(const amg_hip_solver* s) { return s ? (int32_t)s->lv.size() : 0; }
int64_t amg_hip_get_n_dofs(const amg_hip_solver* s, int32_t level) {
  if (!s || level < 0 || level >= (int32_t)s->lv.size()) return -1;
  return s->lv[level].n;
}
int64_t amg_hip_get_level_nnz(const amg_hip_solver* s, int32_t level) {
  if (!s || level < 0 || level >= (int32_t)s->lv.size()) return -1;
  return s->lv[level].nnz_struct;
}
amg_hip_status amg_hip_get_level_matrix(const amg_hip_solver* s, int32_t level,
                                        int32_t* colptr, int32_t* rowind, double* val) {
  if (!s || level < 0 || level >= (int32_t)s->lv.size())
    return fail(AMG_HIP_EINVAL, "level out of range");
  if (!s->opt.host_only) HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(s->lv[level].ensure_host_matrix());
  const Sparse& A = s->lv[level].A_csc;
  if (colptr) std::memcpy(colptr, A.ptr.data(), sizeof(int32_t) * A.ptr.size());
  if (rowind) std::memcpy(rowind, A.idx.data(), sizeof(int32_t) * A.idx.size());
  if (val) std::memcpy(val, A.val.data(), sizeof(double) * A.val.size());
  return AMG_HIP_OK;
}
int64_t amg_hip_get_transfer_nnz(const amg_hip_solver* s, int32_t level, int32_t which) {
  if (!s || level < 0 || level + 1 >= (int32_t)s->lv.size()) return -1;
  return (which ? s->lv[level].R() : s->lv[level].P()).nnz();
}
amg_hip_status amg_hip_get_transfer(const amg_hip_solver* s, int32_t level, int32_t which,
                                    int32_t* colptr, int32_t* rowind, double* val) {
  if (!s || level < 0 || level + 1 >= (int32_t)s->lv.size())
    return fail(AMG_HIP_EINVAL, "level out of range");
  const Sparse& M = which ? s->lv[level].R() : s->lv[level].P();
  if (colptr) std::memcpy(colptr, M.ptr.data(), sizeof(int32_t) * M.ptr.size());
  if (rowind) std::memcpy(rowind, M.idx.data(), sizeof(int32_t) * M.idx.size());
  if (val) std::memcpy(val, M.val.data(), sizeof(double) * M.val.size());
  return AMG_HIP_OK;
}

static DevMem* pick_vec(amg_hip_solver* s, int32_t level, int32_t which) {
  if (!s || level < 0 || level >= (int32_t)s->lv.size()) return nullptr;
  Level& L = s->lv[level];
  // a coarse K-Patch level leaves its solution in the second buffer (its up-leg cannot run in place)
  if (which == 0 && level >= 1 && !s->opt.host_only && patch_level_ok(s, level)) return &L.tmp;
  return which == 0 ? &L.u : (which == 1 ? &L.f : (which == 2 ? &L.r : nullptr));
}
amg_hip_status amg_hip_get_vec(amg_hip_solver* s, int32_t level, int32_t which, double* out) {
  DevMem* m = pick_vec(s, level, which);
  if (!m || !out) return fail(AMG_HIP_EINVAL, "bad level / vector selector");
  if (s->opt.host_only) return fail(AMG_HIP_EINVAL, "host_only solver has no vectors");
  if (which == 2 && !s->opt.keep_residual &&
      (fuses_resid_restrict(s, level) || patch_level_ok(s, level) || mc_patch_ok(s, level) || mc_strip_ok(s, level) ||
       (level == (int32_t)s->lv.size() - 1 && level > 0)))
    return fail(AMG_HIP_EINVAL, "the residual of this level is not kept (create the solver with "
                                "opt.keep_residual = 1)");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(out, m->p, sizeof(double) * s->lv[level].n, hipMemcpyDeviceToHost));
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_set_vec(amg_hip_solver* s, int32_t level, int32_t which,
                               const double* in) {
  DevMem* m = pick_vec(s, level, which);
  if (!m || !in) return fail(AMG_HIP_EINVAL, "bad level / vector selector");
  if (s->opt.host_only) return fail(AMG_HIP_EINVAL, "host_only solver has no vectors");
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  HIP_TRY(hipMemcpy(m->p, in, sizeof(double) * s->lv[level].n, hipMemcpyHostToDevice));
  return AMG_HIP_OK;
}
int64_t amg_hip_coarse_halfbw(const amg_hip_solver* s) { return s ? s->coarse.w : -1; }
int32_t amg_hip_coarse_solve_kind(const amg_hip_solver* s) { return s ? s->coarse.kind : -1; }

namespace {
void layout_of(const DevMat& A, int32_t* layout, int64_t* matrix_stream_bytes) {
  int32_t lay;
  int64_t bytes;
  if (A.dict) {
    lay = AMG_HIP_LAYOUT_DICT;
    bytes = (A.dict_typed ? A.n_rows + 256 * 8 * A.dict_words : A.n_rows * 8 * A.dict_words) +
            (int64_t)A.dict_ntab * 12;
  } else if (A.sell) {
    lay = AMG_HIP_LAYOUT_SELL;
    bytes = A.slots * ((A.idx16 & 1) ? 10 : 12) + (A.n_rows + 63) / 64 * 8;
  } else {
    lay = AMG_HIP_LAYOUT_CSR;
    bytes = A.csr.nnz * 12 + (A.csr.n_rows + 1) * 4;
  }
  if (layout) *layout = lay;
  if (matrix_stream_bytes) *matrix_stream_bytes = bytes;
}
}  // namespace
amg_hip_status amg_hip_level_layout(const amg_hip_solver* s, int32_t level, int32_t* layout,
                                    int64_t* matrix_stream_bytes) {
  if (!s || level < 0 || level >= (int32_t)s->lv.size())
    return fail(AMG_HIP_EINVAL, "level out of range");
  if (s->opt.host_only) return fail(AMG_HIP_EINVAL, "host_only solver has no device matrices");
  layout_of(s->lv[level].A_rows, layout, matrix_stream_bytes);
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_get_colors(const amg_hip_solver* s, int32_t level, int32_t* color,
                                  int32_t* n_colors) {
  if (!s || level < 0 || level >= (int32_t)s->lv.size())
    return fail(AMG_HIP_EINVAL, "level out of range");
  const Level& L = s->lv[level];
  if (L.color.empty()) return fail(AMG_HIP_EINVAL, "solver was not built with the multicolour smoother");
  if (color) std::memcpy(color, L.color.data(), sizeof(int32_t) * L.color.size());
  if (n_colors) *n_colors = L.n_colors;
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_cycle_bytes(const amg_hip_solver* s, double* cycle_bytes,
                                   double* fine_sweep_bytes) {
  if (!s) return fail(AMG_HIP_EINVAL, "null solver");
  if (cycle_bytes) *cycle_bytes = s->cycle_bytes;
  if (fine_sweep_bytes) *fine_sweep_bytes = s->fine_sweep_bytes;
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_profile_fine_sweep(amg_hip_solver* s, int32_t n_launches,
                                          double* avg_ms, double* min_ms) {
  if (!s || n_launches < 1) return fail(AMG_HIP_EINVAL, "bad argument");
  if (s->opt.smoother != AMG_HIP_SM_JACOBI && s->opt.smoother != AMG_HIP_SM_MULTICOLOR_GS)
    return fail(AMG_HIP_EUNSUPPORTED, "profile_fine_sweep: only for the Jacobi and multicolour smoothers");
  amg_hip_status r = set_device(s);
  if (r != AMG_HIP_OK) return r;
  Level& L = s->lv[0];
  const DevMat& A = L.A_cols();
  const bool mc = s->opt.smoother == AMG_HIP_SM_MULTICOLOR_GS;
  const bool mc_patch = mc && mc_patch_ok(s, 0);
  // multicolour, colour kernels: the launch updates u in place; keep a copy in r
  if (mc && !mc_patch)
    HIP_TRY(hipMemcpyAsync(L.r.p, L.u.p, sizeof(double) * L.n, hipMemcpyDeviceToDevice, s->stream));
  std::vector<hipEvent_t> ev(2 * (size_t)n_launches);
  for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  // keep u: sweep u -> tmp only (u itself is never written).  With K-Patch the launch is the
  // level's whole down-leg (2 sweeps + residual + restriction + first coarse sweep); it also
  // rewrites f and tmp of level 1, which every V-cycle recomputes before use.
  const bool patch = !mc && patch_level_ok(s, 0);
  MarchRef MR;
  const bool march = !mc && !patch && march_fill(s, 0, &MR);
  // slab-sharded solver: this rank's launch, i.e. its own lines + halo only
  const bool slab = patch && s->slab.levels > 0 && s->slab.world > 1;
  for (int i = 0; i < n_launches; ++i) {
    HIP_TRY(hipEventRecord(ev[2 * i], s->stream));
    if (mc_patch) {  // the first launch of the level's down-leg (its colour stages): u -> tmp
      int nd = 0;
      const std::vector<McLaunch> plan = mc_plan(s, 0, &nd);
      HIP_TRY(launch_patch_rb(false, false, L.n, L.A_rows.patch_m, L.A_rows.patch_ref(), L.u.as<double>(),
                              L.f.as<double>(), nullptr, s->lv[1].n, L.tmp.as<double>(), nullptr, nullptr,
                              nullptr, plan[0].stages, (uint32_t)L.mc_ctab, s->stream));
    } else if (mc) {  // colour 0 of the forward half
      if (L.mc_dict)
        HIP_TRY(launch_dict_gs_color(L.mc_start[0], L.mc_start[1] - L.mc_start[0], L.mc_words, L.mc_wmax,
                                     L.mc_codes.as<uint64_t>(), L.mc_rowid.as<int32_t>(),
                                     L.mc_doff.as<int32_t>(), L.mc_dval.as<double>(), L.mc_ntab,
                                     L.f.as<double>(), L.u.as<double>(), s->stream));
      else
        HIP_TRY(launch_sell_gs_color(L.mc_mat.n_rows, L.mc_mat.max_width, L.mc_mat.soff.as<int64_t>(),
                                     L.mc_mat.scol.as<int32_t>(), L.mc_mat.sval.as<double>(),
                                     L.mc_rowid.as<int32_t>(), L.mc_start[0], L.mc_start[1] - L.mc_start[0],
                                     L.f.as<double>(), L.u.as<double>(), s->stream));
    } else if (march) {  // both sweeps of level 0 in one plane-marching pass: u -> tmp
      HIP_TRY(launch_march(MR, s->stream));
    } else if (patch) {
      Level& C = s->lv[1];
      HIP_TRY(launch_patch_down(true, L.n, A.patch_m, A.patch_ref(), L.u.as<double>(),
                                L.f.as<double>(), L.tmp.as<double>(), nullptr, C.n, C.f.as<double>(),
                                C.diag.as<double>(), C.tmp.as<double>(), s->opt.omega, s->stream,
                                slab ? s->slab.down_lo[0] : 0, slab ? s->slab.down_hi[0] : -1));
    } else {
      HIP_TRY(launch_mat(CSR_JACOBI, A, L.u.as<double>(), L.f.as<double>(), L.tmp.as<double>(),
                         s->opt.omega, s->stream));
    }
    HIP_TRY(hipEventRecord(ev[2 * i + 1], s->stream));
  }
  if (mc && !mc_patch)
    HIP_TRY(hipMemcpyAsync(L.u.p, L.r.p, sizeof(double) * L.n, hipMemcpyDeviceToDevice, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  double sum = 0, mn = 1e30;
  for (int i = 0; i < n_launches; ++i) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
    sum += ms;
    mn = std::min<double>(mn, ms);
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (avg_ms) *avg_ms = sum / n_launches;
  if (min_ms) *min_ms = mn;
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_cycle_must_move(amg_hip_solver* s, int32_t part, double* bytes) {
  if (!s || !bytes || part < 0 || part > 3) return fail(AMG_HIP_EINVAL, "bad argument");
  if (s->opt.host_only) return fail(AMG_HIP_EINVAL, "host_only solver has no device cycle");
  if (part == CYCLE_ALL && s->must_move[0] == 0.0 && !s->opt.window) {
    // not enqueued yet: capturing the graph walks the cycle without running it
    amg_hip_status r = set_device(s);
    if (r != AMG_HIP_OK) return r;
    if (s->opt.use_graph) {
      if ((r = ensure_graph(s)) != AMG_HIP_OK) return r;
    } else {
      return fail(AMG_HIP_EINVAL, "amg_hip_cycle_must_move: run a V-cycle first");
    }
  }
  *bytes = s->must_move[part];
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_fine_sweep_info(const amg_hip_solver* s, char* name, int32_t name_cap,
                                       int32_t* sweeps_per_launch, double* bytes_per_launch) {
  if (!s || !name || name_cap < 32) return fail(AMG_HIP_EINVAL, "bad argument");
  if (s->opt.host_only) return fail(AMG_HIP_EINVAL, "host_only solver has no device matrices");
  const Level& L = s->lv[0];
  const DevMat& A = L.A_cols();
  int32_t lay = 0;
  int64_t mat = 0;
  layout_of(A, &lay, &mat);
  int sweeps = 1;
  // bytes one launch has to move: what it reads and writes once, not a layout it does not stream
  double bytes = 12.0 * (double)L.nnz_struct + 28.0 * (double)L.n;  // SELL / CSR: SURVEY 8(d)
  if (s->opt.smoother == AMG_HIP_SM_MULTICOLOR_GS) {
    // one launch of the symmetric pass: two colour stages over the level (patch form), or one
    // colour of one direction (colour kernels: half the rows)
    if (mc_patch_ok(s, 0)) {
      std::snprintf(name, (size_t)name_cap, "patch_rb_kernel<%d, %d, %s, false, false>", patch_un(A.patch_un),
                    patch_kind_umask(A.patch_un, A.patch_umask), A.dict_nt ? "true" : "false");
      bytes = 25.0 * (double)L.n;
    } else {
      std::snprintf(name, (size_t)name_cap, "%s", L.mc_dict ? "dict_gs_color_kernel" : "sell_kernel<5>");
      const double rows = (double)(L.mc_start.size() > 1 ? L.mc_start[1] - L.mc_start[0] : L.n);
      // rows of colour 0: code words + dof id (dictionary) or their share of the panels, f, u written
      bytes = L.mc_dict ? rows * (8.0 * L.mc_words + 4.0 + 16.0)
                        : rows / (double)L.n * (12.0 * (double)L.nnz_struct + 28.0 * (double)L.n);
    }
  } else if (A.dict && patch_level_ok(s, 0)) {
    // row types + x + f + smoothed u per fine row; f_H, first coarse sweep, coarse diagonal
    std::snprintf(name, (size_t)name_cap, "patch_down_kernel<%d, %d, true, %s>", patch_un(A.patch_un),
                  patch_kind_umask(A.patch_un, A.patch_umask), A.dict_nt ? "true" : "false");
    sweeps = 2;
    bytes = 25.0 * (double)L.n + (24.0 - 8.0 * A.patch_cfrac) * (double)s->lv[1].n;
    if (s->slab.levels > 0 && s->slab.world > 1) {  // the tiles this rank's launch covers
      const int64_t th = patch_tile_lines();
      const int64_t l0 = s->slab.down_lo[0] / th * th;
      const int64_t l1 = std::min<int64_t>(s->slab.lines, (s->slab.down_hi[0] + th - 1) / th * th);
      bytes *= (double)(l1 - l0) / (double)s->slab.lines;
    }
  } else if (MarchRef MR; A.dict && march_fill(s, 0, &MR)) {
    // both sweeps of the level in one pass: row types + x + f + out
    std::snprintf(name, (size_t)name_cap, "march_kernel");
    sweeps = 2;
    bytes = (double)mat + 24.0 * (double)L.n;
  } else if (A.dict) {
    dict_kernel_name(CSR_JACOBI, A.n_rows, A.dict_ref(), L.f.p, L.tmp.p, name, (size_t)name_cap);
    bytes = (double)mat + 24.0 * (double)L.n;  // matrix stream (1 B / row) + f + x + out
  } else if (A.sell) {
    std::snprintf(name, (size_t)name_cap, "sell_kernel<1, %s, %s>", (A.idx16 & 1) ? "true" : "false",
                  (A.idx16 & 2) ? "true" : "false");
  } else {
    std::snprintf(name, (size_t)name_cap, "csr_stage_kernel<1>");
  }
  if (sweeps_per_launch) *sweeps_per_launch = sweeps;
  if (bytes_per_launch) *bytes_per_launch = bytes;
  return AMG_HIP_OK;
}

// ---- stand-alone operations on host arrays -------------------------------------
amg_hip_status amg_hip_residual(int64_t n, const int32_t* colptr, const int32_t* rowind,
                                const double* val, const double* u, const double* f,
                                double* r) {
  amg_hip_status st = need_device();
  if (st != AMG_HIP_OK) return st;
  if (n <= 0 || !colptr || !rowind || !val || !u || !f || !r)
    return fail(AMG_HIP_EINVAL, "bad argument");
  Sparse A = from_raw(n, n, colptr, rowind, val);
  std::string v = validate(A, "A");
  if (!v.empty()) return fail(AMG_HIP_EINVAL, v);
  Sparse Ar = transpose(A);
  DevMat D;
  DevMem du, df, dr;
  HIP_TRY(upload_mat(Ar, g_default_layout, &D));
  HIP_TRY(upload(du, u, (size_t)n));
  HIP_TRY(upload(df, f, (size_t)n));
  HIP_TRY(dr.alloc(sizeof(double) * n));
  HIP_TRY(launch_mat(CSR_RESID, D, du.as<double>(), df.as<double>(), dr.as<double>(), 1.0, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(r, dr.p, sizeof(double) * n, hipMemcpyDeviceToHost));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_spmv(int64_t rows, int64_t cols, const int32_t* colptr,
                            const int32_t* rowind, const double* val, const double* v,
                            double* out) {
  amg_hip_status st = need_device();
  if (st != AMG_HIP_OK) return st;
  if (rows <= 0 || cols <= 0 || !colptr || !rowind || !val || !v || !out)
    return fail(AMG_HIP_EINVAL, "bad argument");
  Sparse M = from_raw(cols, rows, colptr, rowind, val);
  std::string e = validate(M, "M");
  if (!e.empty()) return fail(AMG_HIP_EINVAL, e);
  Sparse Mr = transpose(M);
  DevMat D;
  DevMem dv, dout;
  HIP_TRY(upload_mat(Mr, g_default_layout, &D));
  HIP_TRY(upload(dv, v, (size_t)cols));
  HIP_TRY(dout.alloc(sizeof(double) * rows));
  HIP_TRY(launch_mat(CSR_SPMV, D, dv.as<double>(), nullptr, dout.as<double>(), 1.0, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, dout.p, sizeof(double) * rows, hipMemcpyDeviceToHost));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_linear_restrict(int64_t n_h, int64_t n_H, const double* r, double* f_H) {
  amg_hip_status st = need_device();
  if (st != AMG_HIP_OK) return st;
  if (n_h <= 0 || n_H <= 0 || !r || !f_H) return fail(AMG_HIP_EINVAL, "bad argument");
  DevMem dr, df;
  HIP_TRY(upload(dr, r, (size_t)n_h));
  HIP_TRY(df.alloc(sizeof(double) * n_H));
  HIP_TRY(launch_linear_restrict(n_h, n_H, dr.as<double>(), df.as<double>(), nullptr, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(f_H, df.p, sizeof(double) * n_H, hipMemcpyDeviceToHost));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_linear_prolong_add(int64_t n_h, int64_t n_H, const double* u_H,
                                          double* u_h) {
  amg_hip_status st = need_device();
  if (st != AMG_HIP_OK) return st;
  if (n_h <= 0 || n_H <= 0 || !u_H || !u_h) return fail(AMG_HIP_EINVAL, "bad argument");
  DevMem dH, dh;
  HIP_TRY(upload(dH, u_H, (size_t)n_H));
  HIP_TRY(upload(dh, u_h, (size_t)n_h));
  HIP_TRY(launch_linear_prolong_add(n_h, n_H, dH.as<double>(), dh.as<double>(), nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(u_h, dh.p, sizeof(double) * n_h, hipMemcpyDeviceToHost));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_rss_host(int64_t n, const int32_t* colptr, const int32_t* rowind,
                                const double* val, const double* u, const double* b,
                                double* out) {
  amg_hip_status st = need_device();
  if (st != AMG_HIP_OK) return st;
  if (n <= 0 || !colptr || !rowind || !val || !u || !b || !out)
    return fail(AMG_HIP_EINVAL, "bad argument");
  Sparse A = from_raw(n, n, colptr, rowind, val);
  std::string v = validate(A, "A");
  if (!v.empty()) return fail(AMG_HIP_EINVAL, v);
  Sparse Ar = transpose(A);
  DevMat D;
  DevMem du, db, dt, sc;
  HIP_TRY(upload_mat(Ar, g_default_layout, &D));
  HIP_TRY(upload(du, u, (size_t)n));
  HIP_TRY(upload(db, b, (size_t)n));
  HIP_TRY(dt.alloc(sizeof(double) * n));
  HIP_TRY(sc.alloc(sizeof(double) * 1100));
  HIP_TRY(launch_mat(CSR_RSSQ, D, du.as<double>(), db.as<double>(), dt.as<double>(), 1.0, nullptr));
  HIP_TRY(launch_sum(n, dt.as<double>(), sc.as<double>() + 1024, sc.as<double>(), 0, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, sc.as<double>() + 1024, sizeof(double), hipMemcpyDeviceToHost));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_coarse_solve(int64_t n, const int32_t* colptr, const int32_t* rowind,
                                    const double* val, const double* f, double* x,
                                    int64_t* halfbw) {
  amg_hip_status st = need_device();
  if (st != AMG_HIP_OK) return st;
  if (n <= 0 || !colptr || !rowind || !val || !f || !x) return fail(AMG_HIP_EINVAL, "bad argument");
  Sparse A = from_raw(n, n, colptr, rowind, val);
  std::string v = validate(A, "A");
  if (!v.empty()) return fail(AMG_HIP_EINVAL, v);
  CoarseOnDev C;
  if ((st = upload_coarse(A, -1, &C)) != AMG_HIP_OK) return st;  // bit-exact forms only
  if (halfbw) *halfbw = C.w;
  DevMem df, dy, dx;
  HIP_TRY(upload(df, f, (size_t)n));
  HIP_TRY(dy.alloc(sizeof(double) * n));
  HIP_TRY(dx.alloc(sizeof(double) * n));
  HIP_TRY(launch_coarse(C, df.as<double>(), dy.as<double>(), dx.as<double>(), nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(x, dx.p, sizeof(double) * n, hipMemcpyDeviceToHost));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_coarse_solve_fast(int64_t n, const int32_t* colptr, const int32_t* rowind,
                                         const double* val, const double* f, double* x,
                                         int64_t* halfbw, int32_t* partition_rows) {
  amg_hip_status st = need_device();
  if (st != AMG_HIP_OK) return st;
  if (n <= 0 || !colptr || !rowind || !val || !f || !x) return fail(AMG_HIP_EINVAL, "bad argument");
  Sparse A = from_raw(n, n, colptr, rowind, val);
  std::string v = validate(A, "A");
  if (!v.empty()) return fail(AMG_HIP_EINVAL, v);
  BandFactor F;
  std::string e = band_factor(A, (size_t)8 << 30, &F);
  if (!e.empty()) return fail(AMG_HIP_EINVAL, e);
  if (halfbw) *halfbw = F.w;
  SpikeFactor SF;
  e = spike_factor(F, &SF);
  if (!e.empty()) return fail(AMG_HIP_EUNSUPPORTED, e);
  if (partition_rows) *partition_rows = SF.c;
  SpikeOnDev D;
  DevMem df, dx;
  HIP_TRY(upload_spike(SF, &D));
  HIP_TRY(upload(df, f, (size_t)n));
  HIP_TRY(dx.alloc(sizeof(double) * n));
  SpikeArgs a = D.a;
  a.f = df.as<double>();
  a.x = dx.as<double>();
  HIP_TRY(launch_spike_solve(a, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(x, dx.p, sizeof(double) * n, hipMemcpyDeviceToHost));
  return AMG_HIP_OK;
}

// SmootherBase::smooth for the built-in kinds, on host arrays.
amg_hip_status amg_hip_smooth(int32_t kind, int64_t n, const int32_t* colptr,
                              const int32_t* rowind, const double* val, double* u,
                              const double* b, double omega, double tol, int64_t every,
                              int64_t n_iters, int64_t* iters, int32_t* converged) {
  amg_hip_status st = need_device();
  if (st != AMG_HIP_OK) return st;
  if (n <= 0 || !colptr || !rowind || !val || !u || !b) return fail(AMG_HIP_EINVAL, "bad argument");
  if (kind == AMG_HIP_SM_SOR && (omega > 2 || omega < 0))
    return fail(AMG_HIP_EINVAL, "`omega` must be in [0, 2] but got omega=" + std::to_string(omega) + "\n");
  if (kind == AMG_HIP_SM_REF_JACOBI && every == 0)
    return fail(AMG_HIP_EINVAL, "AMG::Jacobi divides by compute_error_every_n_iters (smoother.hpp:258); it must be non-zero");
  if (kind == AMG_HIP_SM_SOR && every == 0)
    return fail(AMG_HIP_EINVAL, "AMG::SuccessiveOverRelaxation divides by compute_error_every_n_iters (smoother.hpp:367); it must be non-zero");
  Sparse A = from_raw(n, n, colptr, rowind, val);
  std::string v = validate(A, "A");
  if (!v.empty()) return fail(AMG_HIP_EINVAL, v);
  Sparse Ar = transpose(A);
  DevMat Drows;  // rows of A: rss
  DevMat Dcols;  // CSC arrays as rows: true Jacobi
  const bool sym = same_arrays(A, Ar);
  HIP_TRY(upload_mat(Ar, g_default_layout, &Drows));
  if (!sym) HIP_TRY(upload_mat(A, g_default_layout, &Dcols));
  const DevMat& Dc = sym ? Drows : Dcols;
  DevMem du, db, dt, sc;
  HIP_TRY(upload(du, u, (size_t)n));
  HIP_TRY(upload(db, b, (size_t)n));
  HIP_TRY(dt.alloc(sizeof(double) * n));
  HIP_TRY(sc.alloc(sizeof(double) * 1100));
  LexOnDev Lf, Lb;
  if (kind == AMG_HIP_SM_SPGS) {
    LexSchedule F, B;
    std::string e = build_lex_schedule(A, false, 64, &F);
    if (e.empty()) e = build_lex_schedule(A, true, 64, &B);
    if (!e.empty()) return fail(AMG_HIP_EUNSUPPORTED, e);
    HIP_TRY(upload_lex(F, &Lf));
    HIP_TRY(upload_lex(B, &Lb));
  } else if (kind == AMG_HIP_SM_REF_JACOBI || kind == AMG_HIP_SM_SOR) {
    LexSchedule F;
    std::string e = build_lex_schedule(Ar, false, 64, &F);
    if (!e.empty()) return fail(AMG_HIP_EUNSUPPORTED, e);
    HIP_TRY(upload_lex(F, &Lf));
  } else if (kind == AMG_HIP_SM_JACOBI) {
  } else {
    return fail(AMG_HIP_EUNSUPPORTED, "smoother kind not available as a stand-alone call");
  }
  int64_t iter = 0;
  double error = 100;  // smoother.hpp:194
  double* cur = du.as<double>();
  double* alt = dt.as<double>();
  const bool check = (kind <= AMG_HIP_SM_SOR);
  while (iter < n_iters && (!check || error > tol)) {
    if (kind == AMG_HIP_SM_SPGS) {
      HIP_TRY(launch_gs_lex(Lf.d, db.as<double>(), cur, 0, 1.0, nullptr));
      HIP_TRY(launch_gs_lex(Lb.d, db.as<double>(), cur, 0, 1.0, nullptr));
    } else if (kind == AMG_HIP_SM_REF_JACOBI) {
      HIP_TRY(launch_gs_lex(Lf.d, db.as<double>(), cur, 1, 1.0, nullptr));
    } else if (kind == AMG_HIP_SM_SOR) {
      HIP_TRY(launch_gs_lex(Lf.d, db.as<double>(), cur, 2, omega, nullptr));
    } else {
      HIP_TRY(launch_mat(CSR_JACOBI, Dc, cur, db.as<double>(), alt, omega, nullptr));
      std::swap(cur, alt);
    }
    iter += 1;
    if (check && every != 0 && iter % every == 0) {
      HIP_TRY(launch_mat(CSR_RSSQ, Drows, cur, db.as<double>(), alt, 1.0, nullptr));
      HIP_TRY(launch_sum(n, alt, sc.as<double>() + 1024, sc.as<double>(), 0, nullptr));
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipMemcpy(&error, sc.as<double>() + 1024, sizeof(double), hipMemcpyDeviceToHost));
    }
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(u, cur, sizeof(double) * n, hipMemcpyDeviceToHost));
  if (iters) *iters = iter;
  if (converged) *converged = error <= tol;
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_spgs_sweep(int32_t dir, int64_t n, const int32_t* colptr,
                                  const int32_t* rowind, const double* val, double* u,
                                  const double* b) {
  amg_hip_status st = need_device();
  if (st != AMG_HIP_OK) return st;
  if (n <= 0 || !colptr || !rowind || !val || !u || !b) return fail(AMG_HIP_EINVAL, "bad argument");
  Sparse A = from_raw(n, n, colptr, rowind, val);
  std::string v = validate(A, "A");
  if (!v.empty()) return fail(AMG_HIP_EINVAL, v);
  LexSchedule S;
  std::string e = build_lex_schedule(A, dir < 0, 64, &S);
  if (!e.empty()) return fail(AMG_HIP_EUNSUPPORTED, e);
  LexOnDev L;
  HIP_TRY(upload_lex(S, &L));
  DevMem du, db;
  HIP_TRY(upload(du, u, (size_t)n));
  HIP_TRY(upload(db, b, (size_t)n));
  HIP_TRY(launch_gs_lex(L.d, db.as<double>(), du.as<double>(), 0, 1.0, nullptr));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(u, du.p, sizeof(double) * n, hipMemcpyDeviceToHost));
  return AMG_HIP_OK;
}

// ---- hipIpc arena + stream-ordered halo exchange ---------------------------------
struct amg_hip_arena_impl {
  void* base = nullptr;
  int64_t bytes = 0;
  int device = 0;
};

amg_hip_status amg_hip_arena_create(int64_t bytes, int32_t device, amg_hip_arena** out) {
  if (!out || bytes <= 0) return fail(AMG_HIP_EINVAL, "bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(AMG_HIP_EHIP, "no HIP device available (this library has no CPU fallback)");
  if (device < 0) HIP_TRY(hipGetDevice(&device));
  HIP_TRY(hipSetDevice(device));
  std::unique_ptr<amg_hip_arena_impl> a(new amg_hip_arena_impl);
  a->bytes = bytes;
  a->device = device;
  HIP_TRY(hipMalloc(&a->base, (size_t)bytes));
  HIP_TRY(hipMemset(a->base, 0, (size_t)bytes));
  HIP_TRY(hipDeviceSynchronize());
  *out = reinterpret_cast<amg_hip_arena*>(a.release());
  return AMG_HIP_OK;
}
void amg_hip_arena_destroy(amg_hip_arena* h) {
  auto* a = reinterpret_cast<amg_hip_arena_impl*>(h);
  if (!a) return;
  if (a->base) {
    (void)hipSetDevice(a->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(a->base);
  }
  delete a;
}
void* amg_hip_arena_base(const amg_hip_arena* h) {
  return h ? reinterpret_cast<const amg_hip_arena_impl*>(h)->base : nullptr;
}
amg_hip_status amg_hip_arena_export(const amg_hip_arena* h, uint8_t handle[64]) {
  auto* a = reinterpret_cast<const amg_hip_arena_impl*>(h);
  if (!a || !handle) return fail(AMG_HIP_EINVAL, "bad argument");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  hipIpcMemHandle_t hd;
  HIP_TRY(hipIpcGetMemHandle(&hd, a->base));
  std::memcpy(handle, &hd, 64);
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_arena_open_peer(const uint8_t handle[64], void** peer_base) {
  if (!handle || !peer_base) return fail(AMG_HIP_EINVAL, "bad argument");
  hipIpcMemHandle_t hd;
  std::memcpy(&hd, handle, 64);
  HIP_TRY(hipIpcOpenMemHandle(peer_base, hd, hipIpcMemLazyEnablePeerAccess));
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_arena_close_peer(void* peer_base) {
  if (!peer_base) return AMG_HIP_OK;
  HIP_TRY(hipIpcCloseMemHandle(peer_base));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_halo_push_wait(const amg_hip_halo_desc* d, void* stream) {
  if (!d || d->epoch == 0) return fail(AMG_HIP_EINVAL, "bad halo descriptor");
  hipStream_t st = (hipStream_t)stream;
  const uint32_t e = d->epoch;
  const bool to_prev = d->dst_prev && d->bytes_prev > 0, to_next = d->dst_next && d->bytes_next > 0;
  // 1. my previous message on this channel must have been consumed
  if (e > 1) {
    if (to_prev && d->my_ack_from_prev)
      HIP_TRY(hipStreamWaitValue32(st, d->my_ack_from_prev, e - 1, hipStreamWaitValueGte, 0xffffffffu));
    if (to_next && d->my_ack_from_next)
      HIP_TRY(hipStreamWaitValue32(st, d->my_ack_from_next, e - 1, hipStreamWaitValueGte, 0xffffffffu));
  }
  // 2. push + publish
  if (to_prev) {
    HIP_TRY(hipMemcpyAsync(d->dst_prev, d->src_prev, (size_t)d->bytes_prev, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamWriteValue32(st, d->data_flag_at_prev, e, 0));
  }
  if (to_next) {
    HIP_TRY(hipMemcpyAsync(d->dst_next, d->src_next, (size_t)d->bytes_next, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamWriteValue32(st, d->data_flag_at_next, e, 0));
  }
  // 3. wait for the neighbours' data of this epoch
  if (d->recv_from_prev)
    HIP_TRY(hipStreamWaitValue32(st, d->my_data_from_prev, e, hipStreamWaitValueGte, 0xffffffffu));
  if (d->recv_from_next)
    HIP_TRY(hipStreamWaitValue32(st, d->my_data_from_next, e, hipStreamWaitValueGte, 0xffffffffu));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_halo_ack(const amg_hip_halo_desc* d, void* stream) {
  if (!d || d->epoch == 0) return fail(AMG_HIP_EINVAL, "bad halo descriptor");
  hipStream_t st = (hipStream_t)stream;
  if (d->recv_from_prev && d->ack_flag_at_prev)
    HIP_TRY(hipStreamWriteValue32(st, d->ack_flag_at_prev, d->epoch, 0));
  if (d->recv_from_next && d->ack_flag_at_next)
    HIP_TRY(hipStreamWriteValue32(st, d->ack_flag_at_next, d->epoch, 0));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_halo_exchange_kernel(const amg_hip_halo_kdesc* d, void* stream) {
  if (!d || !d->timeout) return fail(AMG_HIP_EINVAL, "bad halo descriptor");
  static_assert(sizeof(amg_hip_halo_kdesc) == sizeof(HaloArgs), "layout");
  HaloArgs a;
  std::memcpy(&a, d, sizeof(a));
  HIP_TRY(launch_halo_exchange(a, (hipStream_t)stream));
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_halo_ack_kernel(uint32_t* free_at_prev, uint32_t* free_at_next,
                                       void* stream) {
  HIP_TRY(launch_halo_ack(free_at_prev, free_at_next, (hipStream_t)stream));
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_gather_kernel(const amg_hip_gather_kdesc* d, void* stream) {
  if (!d || d->world < 1 || d->world > GATHER_MAX_RANKS) return fail(AMG_HIP_EINVAL, "bad gather descriptor");
  static_assert(sizeof(amg_hip_gather_kdesc) == sizeof(GatherArgs), "layout");
  GatherArgs a;
  std::memcpy(&a, d, sizeof(a));
  HIP_TRY(launch_gather(a, (hipStream_t)stream));
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_gather_ack_kernel(const amg_hip_gather_kdesc* d, void* stream) {
  if (!d || d->world < 1 || d->world > GATHER_MAX_RANKS) return fail(AMG_HIP_EINVAL, "bad gather descriptor");
  GatherArgs a;
  std::memcpy(&a, d, sizeof(a));
  HIP_TRY(launch_gather_ack(a, (hipStream_t)stream));
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_fill_u32(uint32_t* dev_ptr, int64_t count, uint32_t value, void* stream) {
  if (!dev_ptr || count < 0) return fail(AMG_HIP_EINVAL, "bad argument");
  HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)dev_ptr, (int)value, (size_t)count, (hipStream_t)stream));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_capture_begin(void* stream) {
  HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeRelaxed));
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_capture_end(void* stream, void** graph_exec) {
  if (!graph_exec) return fail(AMG_HIP_EINVAL, "bad argument");
  hipGraph_t g = nullptr;
  HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &g));
  hipGraphExec_t ex = nullptr;
  hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return fail(AMG_HIP_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  *graph_exec = ex;
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) return fail(AMG_HIP_EINVAL, "bad argument");
  HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return AMG_HIP_OK;
}
void amg_hip_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
}

// ---- generators -----------------------------------------------------------------
int64_t amg_hip_laplacian(int32_t dim, int64_t n, int32_t* colptr, int32_t* rowind,
                          double* val) {
  if ((dim != 2 && dim != 3) || n <= 0) {
    g_err = "laplacian: dim must be 2 or 3 and n positive";
    return -1;
  }
  const double N = std::pow((double)n, dim);
  if (N * (2 * dim + 1) >= 2147483647.0) {
    g_err = "laplacian: nnz exceeds int32 indices";
    return -1;
  }
  if (!colptr && !rowind && !val) {  // size query
    const int64_t nn = (dim == 2) ? n * n : n * n * n;
    const int64_t faces = (dim == 2) ? 2 * n * (n - 1) : 3 * n * n * (n - 1);
    return nn + 2 * faces;
  }
  Sparse A = laplacian(dim, n);
  if (colptr) std::memcpy(colptr, A.ptr.data(), sizeof(int32_t) * A.ptr.size());
  if (rowind) std::memcpy(rowind, A.idx.data(), sizeof(int32_t) * A.idx.size());
  if (val) std::memcpy(val, A.val.data(), sizeof(double) * A.val.size());
  return A.nnz();
}
amg_hip_status amg_hip_rhs(int32_t dim, int64_t n, double* b) {
  if ((dim != 2 && dim != 3) || n <= 0 || !b) return fail(AMG_HIP_EINVAL, "bad argument");
  rhs(dim, n, b);
  return AMG_HIP_OK;
}

// ---- device-pointer launchers ------------------------------------------------------
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static amg_hip_status dev_csr(int mode, int64_t nrows, int64_t nnz, int32_t max_block_nnz,
                              int32_t max_row_nnz, const int32_t* rowptr, const int32_t* col,
                              const double* val, const double* x, const double* f, double* out,
                              double omega, int64_t diag_shift, void* stream) {
  if (nrows < 0 || nnz < 0 || !rowptr || !col || !val || !x || !out)
    return fail(AMG_HIP_EINVAL, "bad argument");
  if (!aligned16(col) || !aligned16(val))
    return fail(AMG_HIP_EINVAL, "col/val device pointers must be 16-byte aligned");
  if (nrows == 0) return AMG_HIP_OK;
  HIP_TRY(launch_csr(mode, nrows, nnz, max_block_nnz, max_row_nnz, rowptr, col, val, x, f, out,
                     omega, diag_shift, (hipStream_t)stream));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_dict_probe(int64_t nrows, int64_t ncols, const int32_t* rowptr,
                                  const int32_t* col, const double* val, int64_t diag_shift,
                                  int32_t* n_pairs, int32_t* n_row_types, int32_t* words) {
  if (nrows < 0 || ncols < 0 || !rowptr || (rowptr[nrows] > 0 && (!col || !val)))
    return fail(AMG_HIP_EINVAL, "bad argument");
  Sparse M = from_raw(nrows, ncols, rowptr, col, val);
  std::string v = validate(M, "matrix");
  if (!v.empty()) return fail(AMG_HIP_EINVAL, v);
  DictMat T;
  if (!to_dict(M, diag_shift, &T))
    return fail(AMG_HIP_EUNSUPPORTED, "the matrix does not qualify for the dictionary coding");
  // decode: row type -> code words -> (offset, value) pairs, compare with the input
  const bool typed = !T.rtype.empty();
  for (int64_t r = 0; r < nrows; ++r) {
    uint64_t w[2] = {~(uint64_t)0, ~(uint64_t)0};
    for (int k = 0; k < T.words; ++k)
      w[k] = typed ? T.rwords[(size_t)T.rtype[r] * T.words + k] : T.codes[(size_t)r * T.words + k];
    if (typed)
      for (int k = 0; k < T.words; ++k)
        if (w[k] != T.codes[(size_t)r * T.words + k])
          return fail(AMG_HIP_EHIP, "row type table does not reproduce the code words");
    int j = 0;
    for (int32_t q = rowptr[r]; q < rowptr[r + 1]; ++q, ++j) {
      const int code = (int)((w[j >> 3] >> (8 * (j & 7))) & 0xFF);
      if (code == 0xFF || code >= (int)T.doff.size() ||
          (int64_t)T.doff[code] + r + diag_shift != (int64_t)col[q] ||
          std::memcmp(&T.dval[code], &val[q], sizeof(double)) != 0)
        return fail(AMG_HIP_EHIP, "dictionary coding does not round-trip");
    }
    for (; j < 8 * T.words; ++j)
      if (((w[j >> 3] >> (8 * (j & 7))) & 0xFF) != 0xFF)
        return fail(AMG_HIP_EHIP, "dictionary coding has entries past the end of a row");
  }
  if (n_pairs) *n_pairs = (int32_t)T.doff.size();
  if (n_row_types) {
    int32_t nt = 0;
    if (typed)
      for (uint8_t t : T.rtype) nt = std::max<int32_t>(nt, (int32_t)t + 1);
    *n_row_types = nt;
  }
  if (words) *words = T.words;
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_csr_shape(int64_t nrows, const int32_t* rowptr_host,
                                 int32_t* max_block_nnz, int32_t* max_row_nnz) {
  if (nrows < 0 || !rowptr_host) return fail(AMG_HIP_EINVAL, "bad argument");
  int mb = 0, mr = 0;
  for (int64_t r = 0; r < nrows; ++r) mr = std::max(mr, rowptr_host[r + 1] - rowptr_host[r]);
  for (int64_t r = 0; r < nrows; r += 256) {
    const int64_t e = std::min<int64_t>(r + 256, nrows);
    mb = std::max(mb, rowptr_host[e] - rowptr_host[r]);
  }
  if (max_block_nnz) *max_block_nnz = mb;
  if (max_row_nnz) *max_row_nnz = mr;
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_dev_residual(int64_t nrows, int64_t nnz, int32_t max_block_nnz,
                                    int32_t max_row_nnz, const int32_t* rowptr,
                                    const int32_t* col, const double* val, const double* u,
                                    const double* f, double* r, void* stream) {
  if (!f) return fail(AMG_HIP_EINVAL, "bad argument");
  return dev_csr(CSR_RESID, nrows, nnz, max_block_nnz, max_row_nnz, rowptr, col, val, u, f, r,
                 1.0, 0, stream);
}
amg_hip_status amg_hip_dev_jacobi(int64_t nrows, int64_t nnz, int32_t max_block_nnz,
                                  int32_t max_row_nnz, const int32_t* rowptr,
                                  const int32_t* col, const double* val, const double* u_in,
                                  const double* b, double* u_out, double omega,
                                  int64_t diag_shift, void* stream) {
  if (!b) return fail(AMG_HIP_EINVAL, "bad argument");
  return dev_csr(CSR_JACOBI, nrows, nnz, max_block_nnz, max_row_nnz, rowptr, col, val, u_in, b,
                 u_out, omega, diag_shift, stream);
}
amg_hip_status amg_hip_dev_spmv(int64_t nrows, int64_t nnz, int32_t max_block_nnz,
                                int32_t max_row_nnz, const int32_t* rowptr, const int32_t* col,
                                const double* val, const double* v, double* out, void* stream) {
  return dev_csr(CSR_SPMV, nrows, nnz, max_block_nnz, max_row_nnz, rowptr, col, val, v, nullptr,
                 out, 1.0, 0, stream);
}
amg_hip_status amg_hip_dev_jacobi_from_zero(int64_t nrows, const double* diag, const double* b,
                                            double* u_out, double omega, void* stream) {
  if (nrows < 0 || !diag || !b || !u_out) return fail(AMG_HIP_EINVAL, "bad argument");
  HIP_TRY(launch_jacobi_from_zero(nrows, diag, b, u_out, omega, (hipStream_t)stream));
  return AMG_HIP_OK;
}
// ---- device matrix object: a local CSR block uploaded in the solver's own layout ----
struct amg_hip_devmat_impl {
  DevMat m;
  int device = 0;
  int64_t diag_shift = 0;  // the relative column indices were built against this
};
amg_hip_status amg_hip_devmat_create(int64_t nrows, int64_t ncols, const int32_t* rowptr,
                                     const int32_t* col, const double* val, int32_t layout,
                                     int64_t diag_shift, int32_t device, amg_hip_devmat** out) {
  if (!out || nrows < 0 || ncols < 0 || !rowptr || (rowptr[nrows] > 0 && (!col || !val)))
    return fail(AMG_HIP_EINVAL, "bad argument");
  amg_hip_status st0 = need_device();
  if (st0 != AMG_HIP_OK) return st0;
  Sparse M = from_raw(nrows, ncols, rowptr, col, val);
  std::string v = validate(M, "local matrix");
  if (!v.empty()) return fail(AMG_HIP_EINVAL, v);
  std::unique_ptr<amg_hip_devmat_impl> d(new amg_hip_devmat_impl);
  if (device < 0) HIP_TRY(hipGetDevice(&device));
  HIP_TRY(hipSetDevice(device));
  d->device = device;
  d->diag_shift = diag_shift;
  HIP_TRY(upload_mat_pruned(M, layout, true, &d->m, diag_shift));
  *out = reinterpret_cast<amg_hip_devmat*>(d.release());
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_devmat_layout(const amg_hip_devmat* h, int32_t* layout,
                                     int64_t* matrix_stream_bytes) {
  auto* d = reinterpret_cast<const amg_hip_devmat_impl*>(h);
  if (!d) return fail(AMG_HIP_EINVAL, "bad argument");
  layout_of(d->m, layout, matrix_stream_bytes);
  return AMG_HIP_OK;
}
void amg_hip_devmat_destroy(amg_hip_devmat* h) {
  auto* d = reinterpret_cast<amg_hip_devmat_impl*>(h);
  if (!d) return;
  (void)hipSetDevice(d->device);
  delete d;
}
amg_hip_status amg_hip_devmat_apply(const amg_hip_devmat* h, int32_t op, const double* x,
                                    const double* f, double* out, double omega,
                                    int64_t diag_shift, void* stream) {
  auto* d = reinterpret_cast<const amg_hip_devmat_impl*>(h);
  if (!d || !x || !out) return fail(AMG_HIP_EINVAL, "bad argument");
  int mode;
  switch (op) {
    case 0: mode = CSR_RESID; break;
    case 1: mode = CSR_JACOBI; break;
    case 2: mode = CSR_SPMV; break;
    default: return fail(AMG_HIP_EINVAL, "unknown operation");
  }
  if (mode != CSR_SPMV && !f) return fail(AMG_HIP_EINVAL, "bad argument");
  // residual / SpMV do not look at the diagonal: they decode the relative column
  // indices against the shift the matrix was created with
  if (mode != CSR_JACOBI) diag_shift = d->diag_shift;
  if (diag_shift != d->diag_shift)
    return fail(AMG_HIP_EINVAL, "diag_shift differs from the one the matrix was created with");
  HIP_TRY(launch_mat(mode, d->m, x, f, out, omega, (hipStream_t)stream, diag_shift));
  return AMG_HIP_OK;
}

amg_hip_status amg_hip_dev_axpy1(int64_t n, const double* x, double* y, void* stream) {
  if (n < 0 || !x || !y) return fail(AMG_HIP_EINVAL, "bad argument");
  HIP_TRY(launch_add_inplace(n, x, y, (hipStream_t)stream));
  return AMG_HIP_OK;
}
amg_hip_status amg_hip_dev_sumsq(int64_t n, const double* r, double* out, double* scratch,
                                 void* stream) {
  if (n < 0 || !r || !out || !scratch) return fail(AMG_HIP_EINVAL, "bad argument");
  HIP_TRY(launch_sum(n, r, out, scratch, 1, (hipStream_t)stream));
  return AMG_HIP_OK;
}

}  // extern "C"
