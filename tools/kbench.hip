// tools/kbench.hip -- kernel-variant micro-benchmark for the fine-level sweep.
// Builds the n x n 5-point Poisson CSR on the device, runs Jacobi-sweep kernel
// variants back to back (interleaved rounds, one process: guide rule 24) and
// prints algorithmic GB/s (12*nnz + 28*n bytes per sweep) next to a float4 copy
// ceiling.  Not part of the product; used to choose the kernel structure.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/kbench.hip -o kbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// ---------------------------------------------------------------- generators --
__global__ void gen_rowptr(int64_t n, int64_t N, int32_t* rowptr) {
  // row c = j*n+i has 5 - (i==0) - (i==n-1) - (j==0) - (j==n-1) entries; exclusive scan in closed form
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > N) return;
  // entries before row c: 5c - (#rows<c with i==0) - (#rows<c with i==n-1) - (#rows<c with j==0) - (# with j==n-1)
  int64_t j = c / n, i = c % n;
  int64_t i0 = j + (i > 0 ? 1 : 0);              // rows < c with i == 0
  int64_t in1 = j;                               // rows < c with i == n-1 (complete lines only; current line's last not < c unless ... )
  int64_t j0 = c < n ? c : n;                    // rows < c with j == 0
  int64_t jn = c > N - n ? c - (N - n) : 0;      // rows < c with j == n-1
  if (c == N) { i0 = n; in1 = n; j0 = n; jn = n; }
  rowptr[c] = (int32_t)(5 * c - i0 - in1 - j0 - jn);
}
__global__ void gen_entries(int64_t n, int64_t N, const int32_t* rowptr, int32_t* col, double* val,
                            double off, double diag) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  int64_t j = c / n, i = c % n;
  int32_t p = rowptr[c];
  if (j > 0) { col[p] = (int32_t)(c - n); val[p++] = off; }
  if (i > 0) { col[p] = (int32_t)(c - 1); val[p++] = off; }
  col[p] = (int32_t)c; val[p++] = diag;
  if (i + 1 < n) { col[p] = (int32_t)(c + 1); val[p++] = off; }
  if (j + 1 < n) { col[p] = (int32_t)(c + n); val[p++] = off; }
}
__global__ void gen_vec(int64_t N, double* x, double scale) {
  int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c < N) x[c] = scale * (double)((c * 2654435761u) & 0xffff) / 65536.0 - 0.3;
}

// ------------------------------------------------------------------- copy ----
__global__ __launch_bounds__(256) void copy_f4(const float4* __restrict__ a, float4* __restrict__ b, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// read-mostly stream with the sweep's byte mix: read R bytes, write 1/11 of it
__global__ __launch_bounds__(256) void read_f4(const float4* __restrict__ a, float4* __restrict__ b, int64_t n4) {
  float4 s = {0, 0, 0, 0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = a[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  b[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// FETCH_SIZE calibration: the same 1 GiB read with 8-byte and 4-byte lane loads
__global__ __launch_bounds__(256) void read_f2(const double* __restrict__ a, double* __restrict__ b, int64_t n) {
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += a[i];
  b[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void read_f1(const float* __restrict__ a, float* __restrict__ b, int64_t n) {
  float s = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += a[i];
  b[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// --------------------------------------------------- V0: block-staged (product) --
template <int K, int U>
__global__ __launch_bounds__(256) void jac_v0(int64_t n, int64_t nnz, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x,
    const double* __restrict__ f, double* __restrict__ out, double omega) {
  constexpr int CAP = 256 * K;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* lds_val = reinterpret_cast<double*>(smem);
  int32_t* lds_col = reinterpret_cast<int32_t*>(smem + (CAP + 4) * 8);
  const int tid = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 256;
  const int64_t row = r0 + tid;
  const bool live = row < n;
  const int64_t rlast = (r0 + 256 < n) ? r0 + 256 : n;
  const int64_t p0 = rowptr[r0], p1 = rowptr[rlast];
  int64_t rs = 0, re = 0; double fi = 0, xi = 0;
  if (live) { rs = rowptr[row]; re = rowptr[row + 1]; fi = f[row]; xi = x[row]; }
  double acc = 0, diag = 0;
  for (int64_t c0 = p0 & ~(int64_t)3; c0 < p1; c0 += CAP) {
    const int64_t c1 = (c0 + CAP < p1) ? c0 + CAP : p1;
    const int cnt = (int)(c1 - c0);
#pragma unroll
    for (int it = 0; it < K / 4 + 1; ++it) { const int i = (it * 256 + tid) * 4; if (i < cnt) *reinterpret_cast<int4*>(lds_col + i) = *reinterpret_cast<const int4*>(col + c0 + i); }
#pragma unroll
    for (int it = 0; it < K / 2 + 1; ++it) { const int i = (it * 256 + tid) * 2; if (i < cnt) *reinterpret_cast<double2*>(lds_val + i) = *reinterpret_cast<const double2*>(val + c0 + i); }
    __syncthreads();
    int64_t p = rs > c0 ? rs : c0; const int64_t pe = re < c1 ? re : c1;
    while (p < pe) {
      int32_t c[U]; double v[U], xx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const bool ok = p + u < pe; const int o = ok ? (int)(p + u - c0) : (int)(p - c0); c[u] = lds_col[o]; v[u] = lds_val[o]; }
#pragma unroll
      for (int u = 0; u < U; ++u) xx[u] = x[c[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) if (p + u < pe) { if ((int64_t)c[u] == row) diag = v[u]; else acc += v[u] * xx[u]; }
      p += U;
    }
    __syncthreads();
  }
  if (live) out[row] = (diag == 0.0) ? xi : xi + omega * ((fi - acc) / diag - xi);
}

// ------------------------------------- V1: wave-private staging, no block barrier --
// Each wave owns 64 consecutive rows and its own LDS slice; 32-bit offsets.
template <int K, int U, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void jac_v1(int n, int nnz, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x,
    const double* __restrict__ f, double* __restrict__ out, double omega) {
  constexpr int CAP = 64 * K;  // entries per wave
  __shared__ __attribute__((aligned(16))) double s_val[WAVES][CAP + 4];
  __shared__ __attribute__((aligned(16))) int32_t s_col[WAVES][CAP + 4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int r0 = (blockIdx.x * WAVES + wv) * 64;
  if (r0 >= n) return;
  const int row = r0 + lane;
  const bool live = row < n;
  const int rl = r0 + 64 < n ? r0 + 64 : n;
  const int rs = rowptr[live ? row : rl], re = rowptr[live ? row + 1 : rl];
  const int p0 = __builtin_amdgcn_readfirstlane(rs);
  const int p1 = rowptr[rl];
  const int c0 = p0 & ~3;
  const int cnt = p1 - c0;
  double* lv = s_val[wv];
  int32_t* lc = s_col[wv];
  double fi = 0, xi = 0;
  if (live) { fi = f[row]; xi = x[row]; }
  if (cnt <= CAP) {
#pragma unroll
    for (int it = 0; it < (K + 3) / 4; ++it) { const int i = (it * 64 + lane) * 4; if (i < cnt) *reinterpret_cast<int4*>(lc + i) = *reinterpret_cast<const int4*>(col + c0 + i); }
#pragma unroll
    for (int it = 0; it < (K + 1) / 2; ++it) { const int i = (it * 64 + lane) * 2; if (i < cnt) *reinterpret_cast<double2*>(lv + i) = *reinterpret_cast<const double2*>(val + c0 + i); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  double acc = 0, diag = 0;
  if (cnt <= CAP) {
    int p = rs - c0; const int pe = re - c0;
    while (p < pe) {
      int32_t c[U]; double v[U], xx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int o = p + u < pe ? p + u : p; c[u] = lc[o]; v[u] = lv[o]; }
#pragma unroll
      for (int u = 0; u < U; ++u) xx[u] = x[c[u]];
#pragma unroll
      for (int u = 0; u < U; ++u) if (p + u < pe) { if (c[u] == row) diag = v[u]; else acc += v[u] * xx[u]; }
      p += U;
    }
  } else {  // long rows: straight from global
    for (int p = rs; p < re; ++p) { const int c = col[p]; const double v = val[p]; if (c == row) diag = v; else acc += v * x[c]; }
  }
  if (live) out[row] = (diag == 0.0) ? xi : xi + omega * ((fi - acc) / diag - xi);
}

// ------------------------------------------------- V2: direct CSR, lane per row --
template <int U>
__global__ __launch_bounds__(256) void jac_v2(int n, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x,
    const double* __restrict__ f, double* __restrict__ out, double omega) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  const int rs = rowptr[row], re = rowptr[row + 1];
  const double fi = f[row], xi = x[row];
  double acc = 0, diag = 0;
  int p = rs;
  while (p < re) {
    int32_t c[U]; double v[U], xx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int o = p + u < re ? p + u : p; c[u] = col[o]; v[u] = val[o]; }
#pragma unroll
    for (int u = 0; u < U; ++u) xx[u] = x[c[u]];
#pragma unroll
    for (int u = 0; u < U; ++u) if (p + u < re) { if (c[u] == row) diag = v[u]; else acc += v[u] * xx[u]; }
    p += U;
  }
  out[row] = (diag == 0.0) ? xi : xi + omega * ((fi - acc) / diag - xi);
}

// ------------------------------------------------------- V3: SELL-64 (sliced) --
// slice s = rows [64s, 64s+64); entry j of row r at soff[s] + j*64 + (r & 63).
__global__ void sell_width(int n, const int32_t* rowptr, int32_t* width) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s * 64 >= n) return;
  int w = 0;
  for (int r = s * 64; r < s * 64 + 64 && r < n; ++r) w = max(w, rowptr[r + 1] - rowptr[r]);
  width[s] = w;
}
__global__ void sell_fill(int n, const int32_t* rowptr, const int32_t* col, const double* val,
                          const int64_t* soff, const int32_t* width, int32_t* scol, double* sval) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int s = r >> 6;
  const int w = width[s];
  const int rs = rowptr[r], re = rowptr[r + 1];
  for (int j = 0; j < w; ++j) {
    const int64_t at = soff[s] + (int64_t)j * 64 + (r & 63);
    if (rs + j < re) { scol[at] = col[rs + j]; sval[at] = val[rs + j]; }
    else { scol[at] = -1; sval[at] = 0.0; }
  }
}
template <int U>
__global__ __launch_bounds__(256) void jac_v3(int n, const int64_t* __restrict__ soff, const int32_t* __restrict__ width,
    const int32_t* __restrict__ scol, const double* __restrict__ sval, const double* __restrict__ x,
    const double* __restrict__ f, double* __restrict__ out, double omega) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  const int s = row >> 6;
  if (s * 64 >= n) return;
  const int w = width[s];
  const int64_t base = soff[s] + (row & 63);
  const bool live = row < n;
  double fi = 0, xi = 0;
  if (live) { fi = f[row]; xi = x[row]; }
  double acc = 0, diag = 0;
  for (int j0 = 0; j0 < w; j0 += U) {
    int32_t c[U]; double v[U], xx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int j = j0 + u < w ? j0 + u : j0; c[u] = scol[base + (int64_t)j * 64]; v[u] = sval[base + (int64_t)j * 64]; }
#pragma unroll
    for (int u = 0; u < U; ++u) xx[u] = x[c[u] >= 0 ? c[u] : (live ? row : 0)];
#pragma unroll
    for (int u = 0; u < U; ++u) if (j0 + u < w && c[u] >= 0) { if (c[u] == row) diag = v[u]; else acc += v[u] * xx[u]; }
  }
  if (live) out[row] = (diag == 0.0) ? xi : xi + omega * ((fi - acc) / diag - xi);
}

// ----------------------------------- V4: SELL-64 with 16-bit relative columns --
__global__ void sell16_fill(int n, const int64_t* soff, const int32_t* width, const int32_t* scol, int16_t* s16) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int s = r >> 6; const int w = width[s];
  for (int j = 0; j < w; ++j) { const int64_t at = soff[s] + (int64_t)j * 64 + (r & 63); const int c = scol[at]; s16[at] = c < 0 ? (int16_t)-32768 : (int16_t)(c - r); }
}
template <int U, bool NT, int BLOCK>
__global__ __launch_bounds__(BLOCK) void jac_v4(int n, const int64_t* __restrict__ soff,
    const int16_t* __restrict__ s16, const double* __restrict__ sval, const double* __restrict__ x,
    const double* __restrict__ f, double* __restrict__ out, double omega) {
  const int row = blockIdx.x * BLOCK + threadIdx.x;
  const int s = row >> 6;
  if (s * 64 >= n) return;
  const int64_t o0 = soff[s], o1 = soff[s + 1];
  const int w = (int)((o1 - o0) >> 6);
  const int64_t base = o0 + (row & 63);
  const bool live = row < n;
  double fi = 0, xi = 0;
  if (live) { fi = NT ? __builtin_nontemporal_load(f + row) : f[row]; xi = x[row]; }
  double acc = 0, diag = 0;
  for (int j0 = 0; j0 < w; j0 += U) {
    int32_t c[U]; double v[U], xx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u < w ? j0 + u : j0;
      const int d = NT ? __builtin_nontemporal_load(s16 + base + (int64_t)j * 64) : s16[base + (int64_t)j * 64];
      c[u] = d == -32768 ? -1 : row + d;
      v[u] = NT ? __builtin_nontemporal_load(sval + base + (int64_t)j * 64) : sval[base + (int64_t)j * 64];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) xx[u] = x[c[u] >= 0 ? c[u] : 0];
#pragma unroll
    for (int u = 0; u < U; ++u) if (j0 + u < w && c[u] >= 0) { if (c[u] == row) diag = v[u]; else acc += v[u] * xx[u]; }
  }
  if (live) {
    const double r = (diag == 0.0) ? xi : xi + omega * ((fi - acc) / diag - xi);
    if (NT) __builtin_nontemporal_store(r, out + row); else out[row] = r;
  }
}

// ------------------------------------------------------------------- driver ----
struct Timer { hipEvent_t a, b; Timer() { CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); } };

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 4096;
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  const int64_t N = n * n;
  const int64_t nnz = 5 * N - 4 * n;
  printf("grid %lld^2: N=%lld nnz=%lld\n", (long long)n, (long long)N, (long long)nnz);
  int32_t *rowptr, *col; double *val, *x, *f, *out, *out_ref;
  CHECK(hipMalloc(&rowptr, (N + 1) * 4 + 64)); CHECK(hipMalloc(&col, nnz * 4 + 64)); CHECK(hipMalloc(&val, nnz * 8 + 64));
  CHECK(hipMalloc(&x, N * 8)); CHECK(hipMalloc(&f, N * 8)); CHECK(hipMalloc(&out, N * 8)); CHECK(hipMalloc(&out_ref, N * 8));
  const double h = 2.0 / (n + 1); const double off = 1.0 / (h * h), diag = -4.0 / (h * h);
  gen_rowptr<<<(N + 256) / 256, 256>>>(n, N, rowptr);
  gen_entries<<<(N + 255) / 256, 256>>>(n, N, rowptr, col, val, off, diag);
  gen_vec<<<(N + 255) / 256, 256>>>(N, x, 1.0);
  gen_vec<<<(N + 255) / 256, 256>>>(N, f, 3.0);
  CHECK(hipDeviceSynchronize());
  { int32_t last; CHECK(hipMemcpy(&last, rowptr + N, 4, hipMemcpyDeviceToHost)); if (last != nnz) { printf("rowptr bug %d vs %lld\n", last, (long long)nnz); return 1; } }
  // SELL
  const int ns = (int)((N + 63) / 64);
  int32_t* width; int64_t* soff; CHECK(hipMalloc(&width, ns * 4)); CHECK(hipMalloc(&soff, (ns + 1) * 8));
  sell_width<<<(ns + 255) / 256, 256>>>((int)N, rowptr, width);
  std::vector<int32_t> hw(ns); CHECK(hipMemcpy(hw.data(), width, ns * 4, hipMemcpyDeviceToHost));
  std::vector<int64_t> ho(ns + 1); ho[0] = 0; for (int s = 0; s < ns; ++s) ho[s + 1] = ho[s] + (int64_t)hw[s] * 64;
  CHECK(hipMemcpy(soff, ho.data(), (ns + 1) * 8, hipMemcpyHostToDevice));
  int32_t* scol; double* sval; CHECK(hipMalloc(&scol, ho[ns] * 4 + 64)); CHECK(hipMalloc(&sval, ho[ns] * 8 + 64));
  sell_fill<<<(N + 255) / 256, 256>>>((int)N, rowptr, col, val, soff, width, scol, sval);
  CHECK(hipDeviceSynchronize());
  printf("SELL slots %lld (padding %.3f%%)\n", (long long)ho[ns], 100.0 * (ho[ns] - nnz) / nnz);

  const double bytes = 12.0 * nnz + 28.0 * N;
  const unsigned grid256 = (unsigned)((N + 255) / 256);
  const double omega = 0.6;
  // copy buffers (1 GiB each)
  const int64_t n4 = (int64_t)64 << 20; float4 *ca, *cb; CHECK(hipMalloc(&ca, n4 * 16)); CHECK(hipMalloc(&cb, n4 * 16));
  CHECK(hipMemset(ca, 1, n4 * 16));

  int16_t* s16; CHECK(hipMalloc(&s16, ho[ns] * 2 + 64));
  sell16_fill<<<(N + 255) / 256, 256>>>((int)N, soff, width, scol, s16);
  CHECK(hipDeviceSynchronize());
  enum { COPY, READ, READ8, READ4, V0, V0b, V1a, V1b, V1c, V2, V3a, V3b, V4a, V4b, V4c, V4d, NV };
  const char* names[NV] = {"copy_f4 (1GiB->1GiB, bytes=2GiB)", "read_f4 (1GiB read)", "read 8B/lane (1GiB read)", "read 4B/lane (1GiB read)", "v0 block-staged K6 U8 (product)", "v0 block-staged K6 U5",
                           "v1 wave-private K6 U8 W4", "v1 wave-private K6 U5 W4", "v1 wave-private K6 U5 W8", "v2 direct CSR U5", "v3 SELL-64 U5", "v3 SELL-64 U8",
                           "v4 SELL-64 idx16 U5 (product)", "v4 idx16 nontemporal", "v4 idx16 block 512", "v4 idx16 nt block 1024"};
  std::vector<std::vector<float>> t(NV);
  Timer tm;
  auto run = [&](int v) {
    switch (v) {
      case COPY: copy_f4<<<2048, 256>>>(ca, cb, n4); break;
      case READ: read_f4<<<2048, 256>>>(ca, cb, n4); break;
      case READ8: read_f2<<<2048, 256>>>((const double*)ca, (double*)cb, n4 * 2); break;
      case READ4: read_f1<<<2048, 256>>>((const float*)ca, (float*)cb, n4 * 4); break;
      case V0: jac_v0<6, 8><<<grid256, 256, (256 * 6 + 4) * 12>>>(N, nnz, rowptr, col, val, x, f, out, omega); break;
      case V0b: jac_v0<6, 5><<<grid256, 256, (256 * 6 + 4) * 12>>>(N, nnz, rowptr, col, val, x, f, out, omega); break;
      case V1a: jac_v1<6, 8, 4><<<grid256, 256>>>((int)N, (int)nnz, rowptr, col, val, x, f, out, omega); break;
      case V1b: jac_v1<6, 5, 4><<<grid256, 256>>>((int)N, (int)nnz, rowptr, col, val, x, f, out, omega); break;
      case V1c: jac_v1<6, 5, 8><<<(unsigned)((N + 511) / 512), 512>>>((int)N, (int)nnz, rowptr, col, val, x, f, out, omega); break;
      case V2: jac_v2<5><<<grid256, 256>>>((int)N, rowptr, col, val, x, f, out, omega); break;
      case V3a: jac_v3<5><<<grid256, 256>>>((int)N, soff, width, scol, sval, x, f, out, omega); break;
      case V3b: jac_v3<8><<<grid256, 256>>>((int)N, soff, width, scol, sval, x, f, out, omega); break;
      case V4a: jac_v4<5, false, 256><<<grid256, 256>>>((int)N, soff, s16, sval, x, f, out, omega); break;
      case V4b: jac_v4<5, true, 256><<<grid256, 256>>>((int)N, soff, s16, sval, x, f, out, omega); break;
      case V4c: jac_v4<5, false, 512><<<(unsigned)((N + 511) / 512), 512>>>((int)N, soff, s16, sval, x, f, out, omega); break;
      case V4d: jac_v4<5, true, 1024><<<(unsigned)((N + 1023) / 1024), 1024>>>((int)N, soff, s16, sval, x, f, out, omega); break;
    }
  };
  // correctness: every variant == V2 (plain CSR) bit for bit
  run(V2); CHECK(hipDeviceSynchronize()); CHECK(hipMemcpy(out_ref, out, N * 8, hipMemcpyDeviceToDevice));
  std::vector<double> href(N), hout(N);
  CHECK(hipMemcpy(href.data(), out_ref, N * 8, hipMemcpyDeviceToHost));
  for (int v = V0; v < NV; ++v) {
    CHECK(hipMemset(out, 0, N * 8)); run(v); CHECK(hipDeviceSynchronize());
    hipError_t e = hipGetLastError(); if (e != hipSuccess) { printf("%s: launch error %s\n", names[v], hipGetErrorString(e)); continue; }
    CHECK(hipMemcpy(hout.data(), out, N * 8, hipMemcpyDeviceToHost));
    int64_t bad = 0; for (int64_t i = 0; i < N; ++i) bad += memcmp(&href[i], &hout[i], 8) != 0;
    printf("check %-36s mismatches %lld\n", names[v], (long long)bad);
  }
  for (int r = 0; r < rounds + 1; ++r)
    for (int v = 0; v < NV; ++v) {
      CHECK(hipEventRecord(tm.a)); for (int k = 0; k < 5; ++k) run(v); CHECK(hipEventRecord(tm.b)); CHECK(hipEventSynchronize(tm.b));
      float ms; CHECK(hipEventElapsedTime(&ms, tm.a, tm.b)); if (r > 0) t[v].push_back(ms / 5);
    }
  for (int v = 0; v < NV; ++v) {
    std::sort(t[v].begin(), t[v].end());
    const float med = t[v][t[v].size() / 2], mn = t[v][0];
    const double b = v == COPY ? 2.0 * n4 * 16 : ((v == READ || v == READ8 || v == READ4) ? 1.0 * n4 * 16 : bytes);
    printf("%-38s median %8.1f us  min %8.1f us   %7.0f GB/s (median)  %5.1f%% of 8 TB/s\n", names[v], med * 1e3, mn * 1e3, b / med / 1e6, b / med / 1e6 / 80.0);
  }
  return 0;
}
