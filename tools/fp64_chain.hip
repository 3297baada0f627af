// Microbenchmark: latency of a dependent fp64 chain on one wave (what bounds the coarsest banded
// solve, kernels.hip: band_chain_kernel).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void chain(int n, const double* __restrict__ a, double* out, long long* cyc) {
  __shared__ double la[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) la[i] = a[i];
  __syncthreads();
  double x = a[0], y = a[1];
  const long long t0 = __builtin_readcyclecounter();
  const long long w0 = wall_clock64();
  if (MODE == 0) {  // mul -> add, fully dependent
#pragma unroll 8
    for (int i = 0; i < n; ++i) x = y - la[i & 4095] * x;
  } else if (MODE == 1) {  // two-term recurrence as in band_chain (W = 2)
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
      double acc = la[(3 * i) & 4095];
      acc -= la[(3 * i + 1) & 4095] * x;
      acc -= la[(3 * i + 2) & 4095] * y;
      x = y;
      y = acc;
    }
  } else {  // fma chain
#pragma unroll 8
    for (int i = 0; i < n; ++i) x = __builtin_fma(-la[i & 4095], x, y);
  }
  const long long t1 = __builtin_readcyclecounter();
  const long long w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = x + y; cyc[0] = t1 - t0; cyc[1] = w1 - w0; }
}

int main() {
  std::vector<double> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 1e-3 * ((i * 7919) % 1000) / 1000.0;
  double *a, *o; long long* c;
  hipMalloc(&a, 4096 * 8); hipMalloc(&o, 64); hipMalloc(&c, 16);
  hipMemcpy(a, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  const int n = 100000;
  for (int mode = 0; mode < 3; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(1), dim3(64), 0, 0, n, a, o, c);
      else if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(1), dim3(64), 0, 0, n, a, o, c);
      else hipLaunchKernelGGL(chain<2>, dim3(1), dim3(64), 0, 0, n, a, o, c);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
      printf("mode %d: %.1f ns/step (events), %.1f shader cycles/step, %.1f ns/step (100 MHz wall clock)\n", mode,
             ms * 1e6 / n, (double)hc[0] / n, (double)hc[1] * 10.0 / n);
    }
  return 0;
}
