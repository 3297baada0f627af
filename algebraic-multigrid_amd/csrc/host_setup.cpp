// Host-side SETUP of the multigrid hierarchy (see host_setup.hpp).
#include "host_setup.hpp"

#include <algorithm>
#include <climits>
#include <map>
#include <queue>
#include <cmath>
#include <cstring>
#include <thread>

namespace amg_hip {

Sparse from_raw(int64_t n_outer, int64_t n_inner, const int32_t* ptr,
                const int32_t* idx, const double* val) {
  Sparse M;
  M.n_outer = n_outer;
  M.n_inner = n_inner;
  M.ptr.assign(ptr, ptr + n_outer + 1);
  const int64_t nnz = ptr[n_outer];
  M.idx.assign(idx, idx + nnz);
  M.val.assign(val, val + nnz);
  return M;
}

std::string validate(const Sparse& M, const char* name) {
  const std::string nm(name);
  if (M.n_outer < 0 || M.n_inner < 0) return nm + ": negative dimension";
  if ((int64_t)M.ptr.size() != M.n_outer + 1) return nm + ": bad pointer array";
  if (M.ptr[0] != 0) return nm + ": pointer array must start at 0";
  for (int64_t o = 0; o < M.n_outer; ++o) {
    if (M.ptr[o + 1] < M.ptr[o]) return nm + ": pointer array not monotone";
    for (int32_t p = M.ptr[o]; p < M.ptr[o + 1]; ++p) {
      if (M.idx[p] < 0 || M.idx[p] >= M.n_inner)
        return nm + ": inner index out of range";
      if (p > M.ptr[o] && M.idx[p] <= M.idx[p - 1])
        return nm + ": inner indices must be strictly ascending (compressed, "
                    "no duplicates)";
    }
  }
  return "";
}

static Sparse transpose_serial(const Sparse& M) {
  Sparse T;
  T.n_outer = M.n_inner;
  T.n_inner = M.n_outer;
  T.ptr.assign(T.n_outer + 1, 0);
  T.idx.resize(M.nnz());
  T.val.resize(M.nnz());
  for (int32_t i : M.idx) T.ptr[i + 1]++;
  for (int64_t o = 0; o < T.n_outer; ++o) T.ptr[o + 1] += T.ptr[o];
  std::vector<int32_t> cursor(T.ptr.begin(), T.ptr.end() - 1);
  for (int64_t o = 0; o < M.n_outer; ++o)
    for (int32_t p = M.ptr[o]; p < M.ptr[o + 1]; ++p) {
      const int32_t q = cursor[M.idx[p]]++;
      T.idx[q] = (int32_t)o;
      T.val[q] = M.val[p];
    }
  return T;
}

// Threaded form for the large, banded matrices of the hierarchy: the outer range is cut
// into chunks; a chunk touches a narrow window of inner indices, so its histogram and
// its scatter cursors are window-sized.  Entries of an output row keep ascending outer
// order (chunk 0 first, then chunk 1, ...), exactly as in the serial form.
Sparse transpose(const Sparse& M) {
  const int64_t nnz = M.nnz();
  unsigned hw = std::thread::hardware_concurrency();
  const int T0 = (int)std::min<unsigned>(hw ? hw : 1, 16);
  if (T0 < 2 || nnz < (1 << 22) || M.n_outer < 64 * T0) return transpose_serial(M);
  const int nt = T0;
  struct Chunk {
    int64_t o0 = 0, o1 = 0;
    int32_t imin = 0, imax = -1;
    std::vector<int32_t> cnt;  // per inner index of the window, then the scatter cursor
  };
  std::vector<Chunk> ch(nt);
  auto run = [&](auto&& fn) {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(fn, t);
    for (auto& x : th) x.join();
  };
  // chunks of about equal nnz
  for (int t = 0; t < nt; ++t) {
    const int64_t target0 = nnz * t / nt, target1 = nnz * (t + 1) / nt;
    auto at = [&](int64_t target) {
      return (int64_t)(std::lower_bound(M.ptr.begin(), M.ptr.end(), (int32_t)target) - M.ptr.begin());
    };
    ch[t].o0 = t == 0 ? 0 : std::min<int64_t>(at(target0), M.n_outer);
    ch[t].o1 = t == nt - 1 ? M.n_outer : std::min<int64_t>(at(target1), M.n_outer);
  }
  for (int t = 1; t < nt; ++t) ch[t].o0 = ch[t - 1].o1;
  run([&](int t) {
    Chunk& c = ch[t];
    int32_t lo = INT32_MAX, hi = -1;
    for (int32_t p = M.ptr[c.o0]; p < M.ptr[c.o1]; ++p) {
      lo = std::min(lo, M.idx[p]);
      hi = std::max(hi, M.idx[p]);
    }
    c.imin = lo;
    c.imax = hi;
    if (hi < lo) return;
    c.cnt.assign((size_t)hi - lo + 1, 0);
    for (int32_t p = M.ptr[c.o0]; p < M.ptr[c.o1]; ++p) c.cnt[M.idx[p] - lo]++;
  });
  int64_t windows = 0;
  for (auto& c : ch) windows += (int64_t)c.cnt.size();
  if (windows > 4 * M.n_inner + (1 << 20)) return transpose_serial(M);  // not banded: too much scratch
  Sparse T;
  T.n_outer = M.n_inner;
  T.n_inner = M.n_outer;
  T.ptr.assign(T.n_outer + 1, 0);
  T.idx.resize(nnz);
  T.val.resize(nnz);
  for (auto& c : ch)
    for (size_t k = 0; k < c.cnt.size(); ++k) T.ptr[(size_t)c.imin + k + 1] += c.cnt[k];
  for (int64_t o = 0; o < T.n_outer; ++o) T.ptr[o + 1] += T.ptr[o];
  {  // cursors: chunk t starts behind what the chunks before it put into the row
    std::vector<int32_t> filled(T.n_outer, 0);
    for (auto& c : ch)
      for (size_t k = 0; k < c.cnt.size(); ++k) {
        const size_t i = (size_t)c.imin + k;
        const int32_t n_here = c.cnt[k];
        c.cnt[k] = T.ptr[i] + filled[i];
        filled[i] += n_here;
      }
  }
  run([&](int t) {
    Chunk& c = ch[t];
    for (int64_t o = c.o0; o < c.o1; ++o)
      for (int32_t p = M.ptr[o]; p < M.ptr[o + 1]; ++p) {
        const int32_t q = c.cnt[M.idx[p] - c.imin]++;
        T.idx[q] = (int32_t)o;
        T.val[q] = M.val[p];
      }
  });
  return T;
}

bool same_arrays(const Sparse& a, const Sparse& b) {
  return a.n_outer == b.n_outer && a.n_inner == b.n_inner && a.ptr == b.ptr &&
         a.idx == b.idx &&
         std::memcmp(a.val.data(), b.val.data(), sizeof(double) * a.val.size()) == 0;
}

Sparse linear_P(int64_t n_h, int64_t n_H) {
  Sparse P;  // CSC: outer = coarse column j
  P.n_outer = n_H;
  P.n_inner = n_h;
  P.ptr.resize(n_H + 1);
  P.ptr[0] = 0;
  P.idx.reserve(3 * n_H);
  P.val.reserve(3 * n_H);
  static const double w3[3] = {0.5, 1.0, 0.5};
  for (int64_t j = 0; j < n_H; ++j) {
    for (int t = 0; t < 3; ++t) {
      const int64_t i = 2 * j + t;
      if (i < n_h) {
        P.idx.push_back((int32_t)i);
        P.val.push_back(w3[t]);
      }
    }
    P.ptr[j + 1] = (int32_t)P.idx.size();
  }
  return P;
}

bool is_linear_P(const Sparse& P, int64_t n_h, int64_t n_H) {
  return same_arrays(P, linear_P(n_h, n_H));
}

// ------------------------------------------------------------------ SpGEMM ---
namespace {
struct RowBlockOut {
  std::vector<int32_t> cnt;  // entries per row of the block
  std::vector<int32_t> idx;
  std::vector<double> val;
};

void spgemm_rows(const Sparse& A, const Sparse& B, int64_t r0, int64_t r1,
                 RowBlockOut* out) {
  const int64_t nc = B.n_inner;
  std::vector<int32_t> stamp(nc, -1);
  std::vector<double> acc(nc);
  std::vector<int32_t> touched;
  out->cnt.assign(r1 - r0, 0);
  for (int64_t i = r0; i < r1; ++i) {
    touched.clear();
    for (int32_t p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
      const int32_t k = A.idx[p];
      const double a = A.val[p];
      for (int32_t q = B.ptr[k]; q < B.ptr[k + 1]; ++q) {
        const int32_t J = B.idx[q];
        const double t = a * B.val[q];
        if (stamp[J] != (int32_t)(i - r0)) {  // first touch: assign
          stamp[J] = (int32_t)(i - r0);
          acc[J] = t;
          touched.push_back(J);
        } else {
          acc[J] += t;
        }
      }
    }
    std::sort(touched.begin(), touched.end());
    out->cnt[i - r0] = (int32_t)touched.size();
    for (int32_t J : touched) {
      out->idx.push_back(J);
      out->val.push_back(acc[J]);
    }
  }
}
}  // namespace

Sparse spgemm_csr(const Sparse& A, const Sparse& B, int n_threads) {
  const int64_t n = A.n_outer;
  if (n_threads < 1) n_threads = 1;
  // row blocks small enough that the per-block stamp (int32 of local row) works
  int64_t nblk = std::max<int64_t>(n_threads, 1);
  if (n < 4096) nblk = 1;
  std::vector<RowBlockOut> parts(nblk);
  std::vector<std::thread> th;
  auto work = [&](int64_t b) {
    const int64_t r0 = n * b / nblk, r1 = n * (b + 1) / nblk;
    spgemm_rows(A, B, r0, r1, &parts[b]);
  };
  if (nblk == 1) {
    work(0);
  } else {
    for (int64_t b = 0; b < nblk; ++b) th.emplace_back(work, b);
    for (auto& t : th) t.join();
  }
  Sparse C;
  C.n_outer = n;
  C.n_inner = B.n_inner;
  C.ptr.resize(n + 1);
  C.ptr[0] = 0;
  int64_t total = 0;
  for (auto& p : parts) total += (int64_t)p.idx.size();
  C.idx.resize(total);
  C.val.resize(total);
  int64_t row = 0, off = 0;
  for (auto& p : parts) {
    for (int32_t c : p.cnt) {
      C.ptr[row + 1] = C.ptr[row] + c;
      ++row;
    }
    std::memcpy(C.idx.data() + off, p.idx.data(), sizeof(int32_t) * p.idx.size());
    std::memcpy(C.val.data() + off, p.val.data(), sizeof(double) * p.val.size());
    off += (int64_t)p.idx.size();
    RowBlockOut().cnt.swap(p.cnt);
    std::vector<int32_t>().swap(p.idx);
    std::vector<double>().swap(p.val);
  }
  return C;
}

Sparse galerkin_csr(const Sparse& Rr, const Sparse& Ar, const Sparse& Pr,
                    int n_threads) {
  Sparse AP = spgemm_csr(Ar, Pr, n_threads);
  return spgemm_csr(Rr, AP, n_threads);
}

void to_sell64(const Sparse& M, Sell64* S) {
  const int64_t n = M.n_outer;
  const int64_t np = (n + 63) / 64;
  S->n = n;
  S->soff.assign(np + 1, 0);
  S->max_width = 0;
  for (int64_t p = 0; p < np; ++p) {
    int32_t w = 0;
    for (int64_t r = p * 64; r < std::min(n, p * 64 + 64); ++r)
      w = std::max(w, M.ptr[r + 1] - M.ptr[r]);
    S->soff[p + 1] = S->soff[p] + (int64_t)w * 64;
    S->max_width = std::max(S->max_width, w);
  }
  S->col.assign((size_t)S->soff[np], -1);
  S->val.assign((size_t)S->soff[np], 0.0);
  for (int64_t r = 0; r < n; ++r) {
    const int64_t base = S->soff[r >> 6] + (r & 63);
    int64_t j = 0;
    for (int32_t q = M.ptr[r]; q < M.ptr[r + 1]; ++q, ++j) {
      S->col[base + j * 64] = M.idx[q];
      S->val[base + j * 64] = M.val[q];
    }
  }
}

namespace {
// rows [r0, r1) with a chunk-local pair table (codes are local numbers)
struct DictChunk {
  std::vector<int32_t> doff;
  std::vector<double> dval;
  std::vector<uint64_t> words;  // chunk-local distinct rows (row types), `nw` words each
  bool ok = true, typed = true;
};
void dict_encode_rows(const Sparse& M, int64_t diag_shift, const int32_t* rowid, int64_t r0,
                      int64_t r1, int nw, uint64_t* codes, DictChunk* C) {
  std::map<std::pair<int64_t, uint64_t>, int> table;
  int prev[16];  // codes of the previous row, tried first (interior rows repeat)
  for (int j = 0; j < 16; ++j) prev[j] = -1;
  for (int64_t r = r0; r < r1; ++r) {
    uint64_t w[2] = {~(uint64_t)0, ~(uint64_t)0};
    int j = 0;
    for (int32_t q = M.ptr[r]; q < M.ptr[r + 1]; ++q, ++j) {
      const int64_t d = (int64_t)M.idx[q] - (rowid ? (int64_t)rowid[r] : r + diag_shift);
      if (d < INT32_MIN / 2 || d > INT32_MAX / 2) { C->ok = false; return; }
      uint64_t bits;
      std::memcpy(&bits, &M.val[q], 8);
      int code = prev[j];
      bool hit = false;
      if (code >= 0 && C->doff[code] == (int32_t)d) {
        uint64_t tb;
        std::memcpy(&tb, &C->dval[code], 8);
        hit = tb == bits;
      }
      if (!hit) {
        auto key = std::make_pair(d, bits);
        auto it = table.find(key);
        if (it == table.end()) {
          if (C->doff.size() >= 255) { C->ok = false; return; }
          code = (int)C->doff.size();
          table.emplace(key, code);
          C->doff.push_back((int32_t)d);
          C->dval.push_back(M.val[q]);
        } else {
          code = it->second;
        }
        prev[j] = code;
      }
      w[j >> 3] = (w[j >> 3] & ~((uint64_t)0xFF << (8 * (j & 7)))) | ((uint64_t)code << (8 * (j & 7)));
    }
    codes[(size_t)r * nw] = w[0];
    if (nw == 2) codes[(size_t)r * 2 + 1] = w[1];
  }
}
// local code numbers -> global ones in every code word of rows [r0, r1)
void dict_remap_rows(int64_t r0, int64_t r1, int nw, const uint8_t* lut, uint64_t* codes) {
  for (size_t k = (size_t)r0 * nw; k < (size_t)r1 * nw; ++k) {
    uint64_t w = codes[k], o = 0;
    for (int b = 0; b < 8; ++b) o |= (uint64_t)lut[(w >> (8 * b)) & 0xFF] << (8 * b);
    codes[k] = o;
  }
}
void dict_type_rows(int64_t r0, int64_t r1, int nw, const uint64_t* codes, uint8_t* rtype,
                    DictChunk* C) {
  std::map<std::pair<uint64_t, uint64_t>, int> types;
  int last = -1;
  std::pair<uint64_t, uint64_t> last_key(0, 0);
  for (int64_t r = r0; r < r1; ++r) {
    const std::pair<uint64_t, uint64_t> key(codes[(size_t)r * nw], nw == 2 ? codes[(size_t)r * 2 + 1] : 0);
    int t;
    if (last >= 0 && key == last_key) {
      t = last;
    } else {
      auto it = types.find(key);
      if (it == types.end()) {
        if (types.size() >= 255) { C->typed = false; return; }
        t = (int)types.size();
        types.emplace(key, t);
        C->words.push_back(key.first);
        if (nw == 2) C->words.push_back(key.second);
      } else {
        t = it->second;
      }
      last = t;
      last_key = key;
    }
    rtype[(size_t)r] = (uint8_t)t;
  }
}
}  // namespace

// Threaded over row chunks: every chunk numbers its pairs (and then its distinct rows)
// locally, the tables are merged in chunk order and the codes renumbered.
bool to_dict(const Sparse& M, int64_t diag_shift, DictMat* D, const int32_t* rowid) {
  const int64_t n = M.n_outer;
  int32_t mw = 0;
  for (int64_t r = 0; r < n; ++r) mw = std::max(mw, M.ptr[r + 1] - M.ptr[r]);
  if (mw > 16) return false;
  // byte offsets are 32-bit, and the gather of a "no entry" slot reads the row's
  // diagonal column, which therefore has to exist
  if (n >= ((int64_t)1 << 28) || M.n_inner >= ((int64_t)1 << 28) || diag_shift < 0 ||
      (!rowid && n + diag_shift > M.n_inner))
    return false;
  D->n = n;
  D->max_width = mw;
  D->words = mw > 8 ? 2 : 1;
  const int nw = D->words;
  D->doff.clear();
  D->dval.clear();
  D->rtype.clear();
  D->rwords.clear();
  D->codes.assign((size_t)n * nw, ~(uint64_t)0);
  unsigned hw = std::thread::hardware_concurrency();
  const int nt = n < (1 << 20) ? 1 : (int)std::min<unsigned>(hw ? hw : 1, 16);
  std::vector<DictChunk> ch(nt);
  auto run = [&](auto&& fn) {
    if (nt == 1) { fn(0); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(fn, t);
    for (auto& x : th) x.join();
  };
  auto lo = [&](int t) { return n * t / nt; };
  run([&](int t) { dict_encode_rows(M, diag_shift, rowid, lo(t), lo(t + 1), nw, D->codes.data(), &ch[t]); });
  // merge the pair tables
  std::map<std::pair<int32_t, uint64_t>, int> table;
  std::vector<std::vector<uint8_t>> lut(nt, std::vector<uint8_t>(256, 0xFF));
  for (int t = 0; t < nt; ++t) {
    if (!ch[t].ok) return false;
    for (size_t k = 0; k < ch[t].doff.size(); ++k) {
      uint64_t bits;
      std::memcpy(&bits, &ch[t].dval[k], 8);
      auto key = std::make_pair(ch[t].doff[k], bits);
      auto it = table.find(key);
      int code;
      if (it == table.end()) {
        if (D->doff.size() >= 255) return false;
        code = (int)D->doff.size();
        table.emplace(key, code);
        D->doff.push_back(ch[t].doff[k]);
        D->dval.push_back(ch[t].dval[k]);
      } else {
        code = it->second;
      }
      lut[t][k] = (uint8_t)code;
    }
  }
  if (nt > 1) run([&](int t) { dict_remap_rows(lo(t), lo(t + 1), nw, lut[t].data(), D->codes.data()); });
  // row types
  D->rtype.assign((size_t)n, 255);
  run([&](int t) { dict_type_rows(lo(t), lo(t + 1), nw, D->codes.data(), D->rtype.data(), &ch[t]); });
  D->rwords.assign((size_t)256 * nw, ~(uint64_t)0);
  std::map<std::pair<uint64_t, uint64_t>, int> types;
  bool typed = true;
  for (int t = 0; t < nt && typed; ++t) {
    if (!ch[t].typed) { typed = false; break; }
    std::fill(lut[t].begin(), lut[t].end(), 0xFF);
    for (size_t k = 0; k * nw < ch[t].words.size(); ++k) {
      const std::pair<uint64_t, uint64_t> key(ch[t].words[k * nw], nw == 2 ? ch[t].words[k * 2 + 1] : 0);
      auto it = types.find(key);
      int ty;
      if (it == types.end()) {
        if (types.size() >= 255) { typed = false; break; }
        ty = (int)types.size();
        types.emplace(key, ty);
        D->rwords[(size_t)ty * nw] = key.first;
        if (nw == 2) D->rwords[(size_t)ty * 2 + 1] = key.second;
      } else {
        ty = it->second;
      }
      lut[t][k] = (uint8_t)ty;
    }
  }
  if (!typed) {  // too many distinct rows: first level only
    D->rtype.clear();
    D->rwords.clear();
    return true;
  }
  if (nt > 1)
    run([&](int t) {
      const uint8_t* l = lut[t].data();
      for (int64_t r = lo(t); r < lo(t + 1); ++r) D->rtype[(size_t)r] = l[D->rtype[(size_t)r]];
    });
  return true;
}

// ------------------------------------------------------------- banded LDL^T ---
std::string band_factor(const Sparse& A, size_t max_bytes, BandFactor* out) {
  const int64_t n = A.n_outer;
  int64_t w = 0;
  for (int64_t o = 0; o < n; ++o)
    for (int32_t p = A.ptr[o]; p < A.ptr[o + 1]; ++p)
      w = std::max<int64_t>(w, std::llabs((int64_t)A.idx[p] - o));
  const int64_t W = w + 1;
  if ((size_t)n * (size_t)W * sizeof(double) * 3 > max_bytes)
    return "coarsest operator: band storage n*(w+1) = " + std::to_string(n) + "*" +
           std::to_string(W) + " doubles exceeds the limit; use more levels";
  // row-oriented work band: rb[i*W + d] = L[i, i-d], d = 0 holds D[i]
  std::vector<double> rb((size_t)(n * W), 0.0);
  for (int64_t o = 0; o < n; ++o)
    for (int32_t p = A.ptr[o]; p < A.ptr[o + 1]; ++p) {
      const int64_t i = A.idx[p];  // (i, o) with i >= o: lower triangle
      if (i >= o) rb[i * W + (i - o)] = A.val[p];
    }
  std::vector<double> ld(W);  // ld[k-j0] = L[i,k]*D[k]
  for (int64_t i = 0; i < n; ++i) {
    const int64_t j0 = std::max<int64_t>(0, i - w);
    for (int64_t j = j0; j < i; ++j) {
      double s = rb[i * W + (i - j)];
      for (int64_t k = std::max(j0, j - w); k < j; ++k)
        s -= ld[k - j0] * rb[j * W + (j - k)];
      ld[j - j0] = s;
      rb[i * W + (i - j)] = s / rb[j * W];
    }
    double d = rb[i * W];
    for (int64_t k = j0; k < i; ++k) d -= ld[k - j0] * rb[i * W + (i - k)];
    if (d == 0.0 || !std::isfinite(d))
      return "coarsest operator: zero or non-finite pivot in LDL^T at row " +
             std::to_string(i);
    rb[i * W] = d;
  }
  out->n = n;
  out->w = w;
  out->d.resize(n);
  out->lcol.assign((size_t)(n * std::max<int64_t>(w, 1)), 0.0);
  for (int64_t i = 0; i < n; ++i) {
    out->d[i] = rb[i * W];
    for (int64_t d = 1; d <= w && i - d >= 0; ++d)
      out->lcol[(i - d) * w + (d - 1)] = rb[i * W + d];  // L[i, i-d] = L[(i-d)+d, i-d]
  }
  return "";
}

std::string band_schedule(const BandFactor& F, BandSchedule* S) {
  if (F.w > 63)
    return "coarsest operator has half-bandwidth " + std::to_string(F.w) +
           " > 63 (device band solve limit); use more levels";
  int32_t m = 4;
  while (m <= F.w) m *= 2;
  const int64_t n = F.n, w = F.w;
  S->n = n;
  S->m = m;
  S->d = F.d;
  // zero-padded to a multiple of 64 steps plus one prefetched chunk pair
  const int64_t n_pad = (n + 63) / 64 * 64 + 64;
  S->sched_f.assign((size_t)(n_pad * m), 0.0);
  S->sched_b.assign((size_t)(n_pad * m), 0.0);
  for (int64_t s = 0; s < n; ++s)
    for (int64_t d = 1; d <= w; ++d) {
      if (s + d < n)  // forward step s: row s+d, entry L[s+d, s]
        S->sched_f[s * m + ((s + d) % m)] = F.lcol[s * w + (d - 1)];
      const int64_t k = n - 1 - s;  // backward step s: row k-d, entry L[k, k-d]
      if (k - d >= 0)
        S->sched_b[s * m + ((s + d) % m)] = F.lcol[(k - d) * w + (d - 1)];
    }
  return "";
}

void band_chain_schedule(const BandFactor& F, BandChain* S) {
  const int64_t n = F.n, w = F.w;
  S->n = n;
  S->w = w;
  S->d = F.d;
  S->cf.assign((size_t)(n * w), 0.0);
  S->cb.assign((size_t)(n * w), 0.0);
  auto L = [&](int64_t i, int64_t j) -> double { return F.lcol[j * w + (i - j - 1)]; };  // i > j
  for (int64_t s = 0; s < n; ++s)
    for (int64_t t = 0; t < w; ++t) {
      const int64_t k = s - w + t;            // forward: row s, column k
      if (k >= 0) S->cf[s * w + t] = L(s, k);
      const int64_t i = n - 1 - s, r = i + w - t;  // backward: unknown i, row r > i
      if (r < n) S->cb[s * w + t] = L(r, i);
    }
}

std::string band_wide_schedule(const BandFactor& F, size_t max_bytes, BandWide* S) {
  const int64_t n = F.n, w = F.w;
  if (w + 64 > 8192)
    return "coarsest operator has half-bandwidth " + std::to_string(w) +
           " > 8128 (device band solve limit); use more levels";
  const int64_t nb = (n + 63) / 64, stride = (w + 64) * 64;
  if ((size_t)nb * (size_t)stride * sizeof(double) * 2 > max_bytes)
    return "coarsest operator: band solve panels of " + std::to_string(nb) + " x " +
           std::to_string(stride) + " doubles exceed the limit; use more levels";
  S->n = n;
  S->w = w;
  S->d = F.d;
  S->sched_f.assign((size_t)(nb * stride), 0.0);
  S->sched_b.assign((size_t)(nb * stride), 0.0);
  auto L = [&](int64_t i, int64_t j) -> double {  // L[i, j], i > j, inside the band
    return F.lcol[j * w + (i - j - 1)];
  };
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t r0 = b * 64;
    double* pf = S->sched_f.data() + b * stride;
    double* pb = S->sched_b.data() + b * stride;
    for (int64_t l = 0; l < 64 && r0 + l < n; ++l) {
      const int64_t i = r0 + l;          // forward: row i; mirrored: row n-1-i
      for (int64_t t = 0; t < w; ++t) {
        const int64_t k = i - w + t;
        if (k < 0 || k >= r0) continue;
        pf[t * 64 + l] = L(i, k);
        pb[t * 64 + l] = L(n - 1 - k, n - 1 - i);
      }
      for (int64_t s = 0; s < l; ++s) {
        const int64_t k = r0 + s;
        if (i - k > w) continue;
        pf[(w + s) * 64 + l] = L(i, k);
        pb[(w + s) * 64 + l] = L(n - 1 - k, n - 1 - i);
      }
    }
  }
  return "";
}

std::string spike_factor(const BandFactor& F, SpikeFactor* S) {
  const int64_t n = F.n;
  const int w = (int)F.w;
  if (w > 63) return "coarsest operator has half-bandwidth > 63";
  const int wq = w > 0 ? w : 1;
  int32_t m = 4;
  while (m <= w) m *= 2;
  // a local step costs ~1/8 of an m = 64 boundary step: balance c against n / c
  int c = 64 * (int)std::ceil(std::sqrt((double)n * m / 8.0) / 64.0);
  if (c < 64) c = 64;
  if (c < w) c = ((w + 63) / 64) * 64;
  const int64_t P = (n + c - 1) / c;
  S->n = n; S->w = w; S->c = c; S->P = (int32_t)P;
  S->d = F.d;
  auto L = [&](int64_t i, int64_t j) -> double {  // L[i, j], i > j, inside the band
    const int64_t d = i - j;
    if (j < 0 || i >= n || d < 1 || d > w) return 0.0;
    return F.lcol[j * w + (d - 1)];
  };
  // local schedules: partition p as a stand-alone banded system
  S->m = m;
  const int64_t n_pad = (int64_t)(c + 63) / 64 * 64 + 64;
  S->sched_stride = n_pad * m;
  S->sched_f.assign((size_t)(P * S->sched_stride), 0.0);
  S->sched_b.assign((size_t)(P * S->sched_stride), 0.0);
  for (int64_t p = 0; p < P; ++p) {
    const int64_t r0 = p * c, nl = std::min<int64_t>(c, n - r0);
    BandFactor loc;
    loc.n = nl;
    loc.w = w;
    loc.d.assign(F.d.begin() + r0, F.d.begin() + r0 + nl);
    loc.lcol.assign((size_t)(nl * wq), 0.0);
    for (int64_t k = 0; k < nl; ++k)
      for (int d = 1; d <= w && k + d < nl; ++d) loc.lcol[k * wq + (d - 1)] = L(r0 + k + d, r0 + k);
    BandSchedule bs;
    std::string e = band_schedule(loc, &bs);
    if (!e.empty()) return e;
    std::copy(bs.sched_f.begin(), bs.sched_f.end(), S->sched_f.begin() + p * S->sched_stride);
    std::copy(bs.sched_b.begin(), bs.sched_b.end(), S->sched_b.begin() + p * S->sched_stride);
  }
  // spikes
  S->V.assign((size_t)n * wq, 0.0);
  S->W.assign((size_t)n * wq, 0.0);
  S->Vt.assign((size_t)P * m * m, 0.0);   // zero-padded M x M blocks
  S->Wh.assign((size_t)P * m * m, 0.0);
  std::vector<double> col(c);
  for (int64_t p = 0; p < P; ++p) {
    const int64_t r0 = p * c, nl = std::min<int64_t>(c, n - r0);
    if (p > 0)
      for (int k = 0; k < w; ++k) {      // V_p[:, k] = L_pp^-1 (B_p e_k), e_k = k-th tail entry of p-1
        const int64_t gj = r0 - w + k;
        for (int64_t i = 0; i < nl; ++i) {
          double s = L(r0 + i, gj);
          for (int d = 1; d <= w && i - d >= 0; ++d) s -= L(r0 + i, r0 + i - d) * col[i - d];
          col[i] = s;
          S->V[(size_t)(r0 + i) * wq + k] = s;
        }
      }
    if (p + 1 < P)
      for (int k = 0; k < w; ++k) {      // W_p[:, k] = L_pp^-T (B_{p+1}^T e_k), e_k = k-th head entry of p+1
        const int64_t gr = r0 + c + k;
        for (int64_t i = nl - 1; i >= 0; --i) {
          double s = L(gr, r0 + i);
          for (int d = 1; d <= w && i + d < nl; ++d) s -= L(r0 + i + d, r0 + i) * col[i + d];
          col[i] = s;
          S->W[(size_t)(r0 + i) * wq + k] = s;
        }
      }
    // boundary blocks, transposed so that lane k reads contiguously: [p][j][k]
    for (int k = 0; k < w; ++k)
      for (int j = 0; j < w; ++j) {
        const int64_t it = r0 + nl - w + k;   // k-th tail row of partition p
        if (it >= r0) S->Vt[((size_t)p * m + j) * m + k] = S->V[(size_t)it * wq + j];
        const int64_t ih = r0 + k;            // k-th head row
        if (ih < r0 + nl) S->Wh[((size_t)p * m + j) * m + k] = S->W[(size_t)ih * wq + j];
      }
  }
  return "";
}

// ------------------------------------------------- exact lexicographic order ---
std::string build_lex_schedule(const Sparse& M, bool backward, int32_t max_width,
                               LexSchedule* S) {
  const int64_t n = M.n_outer;
  if (M.n_inner != n) return "lexicographic Gauss-Seidel needs a square matrix";
  int32_t width = 0;
  for (int64_t k = 0; k < n; ++k)
    width = std::max(width, M.ptr[k + 1] - M.ptr[k]);
  if (width > max_width)
    return "exact lexicographic Gauss-Seidel kernel supports at most " +
           std::to_string(max_width) + " entries per row, matrix has " +
           std::to_string(width);
  // Dependencies of row k in sweep order: every j "before" k that k reads
  // (true dependency) or that reads k's OLD value (anti-dependency, matters
  // only for structurally non-symmetric matrices).  Exact zeros are skipped.
  const Sparse T = transpose(M);
  std::vector<int32_t> lev(n, 0);
  int32_t maxlev = 0;
  auto before = [&](int64_t j, int64_t k) { return backward ? j > k : j < k; };
  for (int64_t t = 0; t < n; ++t) {
    const int64_t k = backward ? n - 1 - t : t;
    int32_t l = 0;
    for (int32_t p = M.ptr[k]; p < M.ptr[k + 1]; ++p) {
      const int64_t j = M.idx[p];
      if (j != k && M.val[p] != 0.0 && before(j, k)) l = std::max(l, lev[j] + 1);
    }
    for (int32_t p = T.ptr[k]; p < T.ptr[k + 1]; ++p) {
      const int64_t j = T.idx[p];
      if (j != k && T.val[p] != 0.0 && before(j, k)) l = std::max(l, lev[j] + 1);
    }
    lev[k] = l;
    maxlev = std::max(maxlev, l);
  }
  const int64_t n_sets = n == 0 ? 0 : (int64_t)maxlev + 1;
  const double avg = n_sets ? (double)n / (double)n_sets : 0.0;
  int32_t block;
  bool by_sets;
  if (avg >= 256.0) { block = 1024; by_sets = true; }
  else if (avg >= 32.0) { block = 256; by_sets = true; }
  else { block = 64; by_sets = false; }

  std::vector<int32_t> order(n);
  if (by_sets) {  // counting sort by level, sweep order inside a level
    std::vector<int64_t> start(n_sets + 1, 0);
    for (int64_t k = 0; k < n; ++k) start[lev[k] + 1]++;
    for (int64_t s = 0; s < n_sets; ++s) start[s + 1] += start[s];
    for (int64_t t = 0; t < n; ++t) {
      const int64_t k = backward ? n - 1 - t : t;
      order[start[lev[k]]++] = (int32_t)k;
    }
  } else {
    for (int64_t t = 0; t < n; ++t) order[t] = (int32_t)(backward ? n - 1 - t : t);
  }
  std::vector<int32_t> pos(n);
  for (int64_t p = 0; p < n; ++p) pos[order[p]] = (int32_t)p;

  S->block = block;
  S->width = width;
  S->n = n;
  S->n_sets = n_sets;
  const int64_t n_win = (n + block - 1) / block;
  S->n_slots = n_win * block;
  S->row.assign(S->n_slots, -1);
  S->depth.assign(S->n_slots, 0);
  S->win_depth.assign(n_win, 0);
  S->col.assign((size_t)S->n_slots * width, -1);
  S->val.assign((size_t)S->n_slots * width, 0.0);
  S->src.assign((size_t)S->n_slots * width, -1);
  for (int64_t p = 0; p < n; ++p) {
    const int64_t k = order[p];
    const int64_t w0 = (p / block) * block;
    S->row[p] = (int32_t)k;
    int32_t dep = 0;
    int32_t e = 0;
    for (int32_t q = M.ptr[k]; q < M.ptr[k + 1]; ++q, ++e) {
      const int64_t j = M.idx[q];
      const size_t at = (size_t)e * S->n_slots + p;
      S->col[at] = (int32_t)j;
      S->val[at] = M.val[q];
      if (j != k && M.val[q] != 0.0 && before(j, k) && pos[j] >= w0) {
        // producer sits in this window (necessarily in an earlier slot)
        S->src[at] = (int16_t)(pos[j] - w0);
        dep = std::max<int32_t>(dep, S->depth[pos[j]] + 1);
      }
    }
    S->depth[p] = (int16_t)dep;
    int32_t& wd = S->win_depth[p / block];
    wd = std::max(wd, dep);
  }
  return "";
}

void greedy_coloring(const Sparse& M, std::vector<int32_t>* color,
                     int32_t* n_colors) {
  const int64_t n = M.n_outer;
  const Sparse T = transpose(M);
  color->assign(n, -1);
  int32_t nc = 0;
  std::vector<int32_t> mark;  // mark[c] = last row that saw colour c adjacent
  for (int64_t k = 0; k < n; ++k) {
    auto visit = [&](const Sparse& S) {
      for (int32_t p = S.ptr[k]; p < S.ptr[k + 1]; ++p) {
        const int64_t j = S.idx[p];
        if (j == k || S.val[p] == 0.0) continue;
        const int32_t c = (*color)[j];
        if (c >= 0) mark[c] = (int32_t)k;
      }
    };
    visit(M);
    visit(T);
    int32_t c = 0;
    while (c < nc && mark[c] == (int32_t)k) ++c;
    if (c == nc) {
      ++nc;
      mark.push_back(-1);
    }
    (*color)[k] = c;
  }
  *n_colors = nc;
}

void build_color_perm(const Sparse& M, const std::vector<int32_t>& color, int32_t n_colors,
                      ColorPerm* P) {
  const int64_t n = M.n_outer;
  std::vector<int64_t> cnt(n_colors, 0);
  for (int64_t i = 0; i < n; ++i) cnt[color[i]]++;
  P->start.assign(n_colors + 1, 0);
  for (int32_t c = 0; c < n_colors; ++c) P->start[c + 1] = P->start[c] + (cnt[c] + 63) / 64 * 64;
  const int64_t ns = P->start[n_colors];
  P->rowid.assign(ns, -1);
  std::vector<int64_t> cur(P->start.begin(), P->start.end() - 1);
  for (int64_t i = 0; i < n; ++i) P->rowid[cur[color[i]]++] = (int32_t)i;  // ascending inside a colour
  Sparse& R = P->rows;
  R.n_outer = ns;
  R.n_inner = M.n_inner;
  R.ptr.assign(ns + 1, 0);
  R.idx.clear();
  R.val.clear();
  R.idx.reserve(M.idx.size());
  R.val.reserve(M.val.size());
  for (int64_t p = 0; p < ns; ++p) {
    const int32_t i = P->rowid[p];
    if (i >= 0)
      for (int32_t q = M.ptr[i]; q < M.ptr[i + 1]; ++q)
        if (M.val[q] != 0.0 || M.idx[q] == i) {
          R.idx.push_back(M.idx[q]);
          R.val.push_back(M.val[q]);
        }
    R.ptr[p + 1] = (int32_t)R.idx.size();
  }
}

// ---------------------------------------------------------------- grid.hpp ---
Sparse laplacian(int dim, int64_t n, int64_t n_last) {
  // grid.hpp:31,50-75,88-98.  D = tridiag(1,-2,1)/(h*h) with h = 2/(n+1);
  // A = sum over axes of I (x) .. D .. (x) I.  Diagonal = D_ii added once per
  // axis; every off-diagonal appears in exactly one Kronecker term.
  // n_last >= 1: only n_last units (grid lines in 2-D, x-y planes in 3-D) of the slowest axis,
  // i.e. the principal submatrix of a block of whole units -- the WINDOW of a sharded solver
  // (amg_hip_create_poisson_window); h stays that of the n^dim problem.
  const double h = 2.0 / (double)(n + 1);
  const double hh = h * h;
  const double off = 1.0 / hh;
  const double dg = -2.0 / hh;
  double diag = dg + dg;
  if (dim == 3) diag = diag + dg;
  if (n_last < 1) n_last = n;
  const int64_t ext[3] = {n, dim == 2 ? n_last : n, dim == 3 ? n_last : 1};
  const int64_t N = ext[0] * ext[1] * ext[2];
  const int64_t strides[3] = {1, n, n * n};
  Sparse A;
  A.n_outer = A.n_inner = N;
  A.ptr.resize(N + 1);
  A.ptr[0] = 0;
  A.idx.reserve((size_t)N * (2 * dim + 1));
  A.val.reserve((size_t)N * (2 * dim + 1));
  for (int64_t c = 0; c < N; ++c) {
    int64_t coord[3] = {c % n, dim == 2 ? c / n : (c / n) % n, c / (n * n)};
    for (int a = dim - 1; a >= 0; --a)  // lower neighbours, ascending index
      if (coord[a] > 0) {
        A.idx.push_back((int32_t)(c - strides[a]));
        A.val.push_back(off);
      }
    A.idx.push_back((int32_t)c);
    A.val.push_back(diag);
    for (int a = 0; a < dim; ++a)
      if (coord[a] + 1 < ext[a]) {
        A.idx.push_back((int32_t)(c + strides[a]));
        A.val.push_back(off);
      }
    A.ptr[c + 1] = (int32_t)A.idx.size();
  }
  return A;
}

void rhs_range(int dim, int64_t n, double* b, int64_t d0, int64_t d1) {
  // grid.hpp:108-140 with Eigen's LinSpaced(n+2, -1, 1):
  // x_i = (i == n+1) ? 1 : -1 + i*step, step = 2/(n+1).
  const int64_t np = n + 2;
  const double step = (1.0 - (-1.0)) / (double)(np - 1);
  std::vector<double> x(np);
  for (int64_t i = 0; i < np; ++i) x[i] = (i == np - 1) ? 1.0 : -1.0 + (double)i * step;
  for (int64_t dof = d0; dof < d1; ++dof) {
    const int64_t i = dof % n + 1, j = (dof / n) % n + 1;
    if (dim == 2) {
      b[dof] = 5 * std::exp(-10 * (x[j] * x[j] + x[i] * x[i]));
    } else {
      const int64_t k = dof / (n * n) + 1;
      b[dof] = 5 * std::exp(-10 * ((x[k] * x[k] + x[j] * x[j]) + x[i] * x[i]));
    }
  }
}

void rhs(int dim, int64_t n, double* b) {
  int64_t N = n * n;
  if (dim == 3) N *= n;
  rhs_range(dim, n, b, 0, N);
}


// ---- strength-based C/F coarsening (host_setup.hpp) -----------------------------------
Sparse ruge_stueben_P(const Sparse& A, double theta, std::vector<uint8_t>* is_c_out) {
  const int64_t n = A.n_outer;
  // strength graph S (CSR, column indices ascending like A's) and its transpose
  std::vector<int32_t> sp(n + 1, 0), stp(n + 1, 0);
  std::vector<int32_t> sj;
  sj.reserve((size_t)A.nnz());
  std::vector<double> sgn(n, 1.0);
  for (int64_t i = 0; i < n; ++i) {
    double d = 0.0;
    for (int32_t p = A.ptr[i]; p < A.ptr[i + 1]; ++p)
      if (A.idx[p] == i) d = A.val[p];
    sgn[i] = d < 0.0 ? -1.0 : 1.0;
    double mx = 0.0;
    for (int32_t p = A.ptr[i]; p < A.ptr[i + 1]; ++p)
      if (A.idx[p] != i) mx = std::max(mx, -sgn[i] * A.val[p]);
    if (mx > 0.0) {
      const double thr = theta * mx;
      for (int32_t p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
        const double sij = -sgn[i] * A.val[p];
        if (A.idx[p] != i && sij > 0.0 && sij >= thr) {
          sj.push_back(A.idx[p]);
          ++stp[A.idx[p] + 1];
        }
      }
    }
    sp[i + 1] = (int32_t)sj.size();
  }
  for (int64_t i = 0; i < n; ++i) stp[i + 1] += stp[i];
  std::vector<int32_t> stj(sj.size());
  {
    std::vector<int32_t> fill(stp.begin(), stp.end() - 1);
    for (int64_t i = 0; i < n; ++i)
      for (int32_t p = sp[i]; p < sp[i + 1]; ++p) stj[fill[sj[p]]++] = (int32_t)i;
  }
  // first pass
  enum : uint8_t { U = 0, C = 1, F = 2 };
  std::vector<uint8_t> st(n, U);
  std::vector<int32_t> lam(n);
  using Key = std::pair<int32_t, int32_t>;  // (measure, -index): max measure, then lowest index
  std::priority_queue<Key> heap;
  for (int64_t i = 0; i < n; ++i) {
    lam[i] = stp[i + 1] - stp[i];
    if (sp[i + 1] == sp[i]) st[i] = F;  // no strong coupling: nothing to interpolate from
    else heap.emplace(lam[i], -(int32_t)i);
  }
  while (!heap.empty()) {
    const Key top = heap.top();
    heap.pop();
    const int32_t i = -top.second;
    if (st[i] != U || lam[i] != top.first) continue;  // stale entry
    st[i] = C;
    for (int32_t p = stp[i]; p < stp[i + 1]; ++p) {
      const int32_t j = stj[p];
      if (st[j] != U) continue;
      st[j] = F;
      for (int32_t q = sp[j]; q < sp[j + 1]; ++q) {
        const int32_t k = sj[q];
        if (st[k] == U) heap.emplace(++lam[k], -k);
      }
    }
    for (int32_t p = sp[i]; p < sp[i + 1]; ++p) {
      const int32_t j = sj[p];
      if (st[j] == U) heap.emplace(--lam[j], -j);
    }
  }
  // coarse numbering and direct interpolation, row-wise (CSR(P)), then to CSC
  std::vector<int32_t> cidx(n, -1);
  int32_t nc = 0;
  for (int64_t i = 0; i < n; ++i)
    if (st[i] == C) cidx[i] = nc++;
  Sparse Pr;  // CSR(P): n x nc
  Pr.n_outer = n;
  Pr.n_inner = nc;
  Pr.ptr.assign(n + 1, 0);
  for (int64_t i = 0; i < n; ++i) {
    if (st[i] == C) {
      Pr.idx.push_back(cidx[i]);
      Pr.val.push_back(1.0);
    } else {
      double num = 0.0, den = 0.0, dg = 0.0;
      for (int32_t p = A.ptr[i]; p < A.ptr[i + 1]; ++p)
        if (A.idx[p] == i) dg = A.val[p];
      for (int32_t p = A.ptr[i]; p < A.ptr[i + 1]; ++p) {
        if (A.idx[p] == i) continue;
        const double sij = -sgn[i] * A.val[p];
        if (sij > 0.0) num += A.val[p];
        else if (sij < 0.0) dg += A.val[p];
      }
      for (int32_t q = sp[i]; q < sp[i + 1]; ++q)
        if (st[sj[q]] == C) {
          // the value of a_{i, sj[q]}: S keeps A's column order, walk A alongside
          for (int32_t p = A.ptr[i]; p < A.ptr[i + 1]; ++p)
            if (A.idx[p] == sj[q]) den += A.val[p];
        }
      if (den != 0.0 && dg != 0.0) {
        const double alpha = num / den;
        for (int32_t q = sp[i]; q < sp[i + 1]; ++q) {
          if (st[sj[q]] != C) continue;
          double aij = 0.0;
          for (int32_t p = A.ptr[i]; p < A.ptr[i + 1]; ++p)
            if (A.idx[p] == sj[q]) aij = A.val[p];
          Pr.idx.push_back(cidx[sj[q]]);
          Pr.val.push_back(-alpha * aij / dg);
        }
      }
    }
    Pr.ptr[i + 1] = (int32_t)Pr.idx.size();
  }
  if (is_c_out) {
    is_c_out->assign(n, 0);
    for (int64_t i = 0; i < n; ++i) (*is_c_out)[i] = st[i] == C;
  }
  return transpose(Pr);  // CSC(P): n_outer = nc columns, n_inner = n rows
}

}  // namespace amg_hip
