// triad.hip -- what the memory system gives the traffic mix of a Jacobi sweep on the
// dictionary-coded layout (two read streams + one write stream of 8 bytes per row), and
// which ingredient of the real kernel costs what.  Standalone: one 16.7 M-row problem
// (the 4096^2 fine level), kernels launched back to back as in the V-cycle.
//   hipcc --offload-arch=gfx950 -O3 tools/triad.hip -o tools/triad && tools/triad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef double f64x2 __attribute__((ext_vector_type(2)));

// V: 0 = out = f + s*x (2 rows per lane, 16-byte accesses)
//    1 = + one byte per row of "row type"
//    2 = x read as the 5-point stencil's five 8-byte gathers per row instead of one stream
//    3 = 2 + a 10 KB LDS table staged per workgroup + barrier + one LDS lookup per entry
//    4 = 3 + an fp64 division per row
//    5 = 2 + barrier only (64 B staged), no lookups
//    6 = 2 + 64 B staged + barrier + lookups
//    7 = 2 + lookups in a wave-private table, no workgroup barrier
//    8 = 6, with the gathers depending on the looked-up value (offset from the table)
//   10 = 2 with a 2-D patch of rows per workgroup (8 grid lines x 64 columns) instead of
//        512 consecutive rows: the +-4096 neighbours are rows of the same workgroup
//    9 = the tile's three x windows staged in LDS with dense 16-byte loads (3 per lane
//        instead of 10 gathers), stencil read from LDS, weights looked up as in 6
template <int V>
__global__ __launch_bounds__(256) void k(int n, const double* __restrict__ f, const double* x,
                                         const uint8_t* __restrict__ ty,
                                         const double* __restrict__ tabg, double* __restrict__ out,
                                         double s) {
  __shared__ double tab[1280];
  int row0 = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (V == 10) {  // tile -> (column block of 64, block of 8 lines); wave w owns lines 2w, 2w+1
    const int S = 4096, nb = S / 64;
    const int jb = blockIdx.x % nb, kb = blockIdx.x / nb;
    const int line = kb * 8 + ((int)threadIdx.x >> 5);          // 32 lanes x 2 rows per line
    row0 = line * S + jb * 64 + ((int)threadIdx.x & 31) * 2;
  }
  if (row0 + 1 >= n) return;
  const f64x2 fi = __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(f + row0));
  uint32_t t = 0;
  if (V >= 1) t = *reinterpret_cast<const uint16_t*>(ty + row0);
  if (V == 3 || V == 4) {
    for (int i = threadIdx.x; i < 1280; i += 256) tab[i] = tabg[i];
    __syncthreads();
  }
  if (V == 5 || V == 6 || V == 8) {
    if (threadIdx.x < 8) tab[threadIdx.x] = tabg[threadIdx.x];
    __syncthreads();
  }
  double* wtab = tab + (threadIdx.x >> 6) * 64;
  if (V == 7) {
    if ((threadIdx.x & 63) < 8) wtab[threadIdx.x & 63] = tabg[threadIdx.x & 63];
    __builtin_amdgcn_wave_barrier();
  }
  double acc[2];
  if (V == 9) {
    __shared__ double xs[3][528];
    const int t0 = blockIdx.x * 512;
    const int lo[3] = {-4096, -8, 4096};   // window starts (even, 16-byte aligned)
    {
      const int i = threadIdx.x * 2;
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int g = t0 + lo[kk] + i;
        f64x2 v2 = {0.0, 0.0};
        if (g >= 0 && g + 1 < n) v2 = *reinterpret_cast<const f64x2*>(x + g);
        xs[kk][i] = v2.x;
        xs[kk][i + 1] = v2.y;
      }
      if (threadIdx.x < 8) {   // the middle window is 528 wide
        const int g = t0 + lo[1] + 512 + i;
        f64x2 v2 = {0.0, 0.0};
        if (g >= 0 && g + 1 < n) v2 = *reinterpret_cast<const f64x2*>(x + g);
        xs[1][512 + i] = v2.x;
        xs[1][512 + i + 1] = v2.y;
      }
      if (threadIdx.x < 8) tab[threadIdx.x] = tabg[threadIdx.x];
    }
    __syncthreads();
    const int l0 = threadIdx.x * 2;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int l = l0 + r;
      const double w = tab[((t >> (8 * r)) & 0x7)];
      acc[r] = w * xs[0][l];
      acc[r] += w * xs[1][l + 7];
      acc[r] += w * xs[1][l + 8];
      acc[r] += w * xs[1][l + 9];
      acc[r] += w * xs[2][l];
    }
  } else if (V < 2) {
    const f64x2 xi = *reinterpret_cast<const f64x2*>(x + row0);
    acc[0] = xi.x; acc[1] = xi.y;
  } else {
    const int off[5] = {-4096, -1, 0, 1, 4096};
    double xx[2][5];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        int c = row0 + r + off[u];
        if (V == 8) c += (int)tab[((t >> (8 * r)) & 0x7) + (u & 1)];  // 0.0 in the table
        c = c < 0 ? 0 : (c >= n ? n - 1 : c);
        xx[r][u] = x[c];
      }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      acc[r] = 0.0;
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        double w = 0.25;
        if (V == 3 || V == 4) w = tab[((t >> (8 * r)) & 0xFF) * 5 + u];
        if (V == 6 || V == 8) w = tab[((t >> (8 * r)) & 0x7) + (u & 1)];
        if (V == 7) w = wtab[((t >> (8 * r)) & 0x7) + (u & 1)];
        acc[r] += w * xx[r][u];
      }
    }
  }
  f64x2 o;
  o.x = fi.x + s * acc[0] + (double)(t & 0xFF) * 1e-300;
  o.y = fi.y + s * acc[1] + (double)(t >> 8) * 1e-300;
  if (V >= 4) { o.x = o.x / (4.0 + acc[0] * 1e-300); o.y = o.y / (4.0 + acc[1] * 1e-300); }
  __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(out + row0));
}

template <int V>
double run(int n, double* f, double* xa, double* xb, uint8_t* ty, double* tab) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const unsigned grid = (unsigned)((n / 2 + 255) / 256);
  const int reps = 40;
  for (int k2 = 0; k2 < reps + 4; ++k2) {
    if (k2 == 4) hipEventRecord(a);
    // ping-pong like consecutive sweeps: x of one launch is the output of the previous
    hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, n, f, (k2 & 1) ? xb : xa, ty, tab,
                       (k2 & 1) ? xa : xb, 0.5);
  }
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps * 1e3;  // us
}

int main() {
  const int n = 16777216;
  double *f, *xa, *xb, *tab;
  uint8_t* ty;
  hipMalloc(&f, (size_t)n * 8);
  hipMalloc(&xa, (size_t)n * 8);
  hipMalloc(&xb, (size_t)n * 8);
  hipMalloc(&ty, (size_t)n);
  hipMalloc(&tab, 1280 * 8);
  hipMemset(f, 0, (size_t)n * 8);
  hipMemset(xa, 0, (size_t)n * 8);
  hipMemset(xb, 0, (size_t)n * 8);
  hipMemset(ty, 0, (size_t)n);
  hipMemset(tab, 0, 1280 * 8);
  const double us[11] = {run<0>(n, f, xa, xb, ty, tab), run<1>(n, f, xa, xb, ty, tab),
                        run<2>(n, f, xa, xb, ty, tab), run<3>(n, f, xa, xb, ty, tab),
                        run<4>(n, f, xa, xb, ty, tab), run<5>(n, f, xa, xb, ty, tab),
                        run<6>(n, f, xa, xb, ty, tab), run<7>(n, f, xa, xb, ty, tab),
                        run<8>(n, f, xa, xb, ty, tab), run<9>(n, f, xa, xb, ty, tab),
                        run<10>(n, f, xa, xb, ty, tab)};
  const char* name[11] = {"stream triad out = f + s*x (24 B/row)", "+ 1 B/row of row types (25 B/row)",
                         "x as five 8-byte gathers per row", "+ 10 KB LDS table per workgroup, lookups",
                         "+ fp64 division per row", "gathers + barrier only (64 B staged)",
                         "gathers + 64 B staged + barrier + lookups", "gathers + wave-private table, no barrier",
                         "as before, gather address from the table",
                         "x windows staged in LDS with 16-byte loads",
                         "five gathers, 8 lines x 64 columns per workgroup"};
  for (int v = 0; v < 11; ++v)
    std::printf("%-46s %7.1f us  %5.2f TB/s of the %d B/row\n", name[v], us[v],
                (v == 0 ? 24.0 : 25.0) * n / us[v] / 1e6, v == 0 ? 24 : 25);
  return 0;
}
