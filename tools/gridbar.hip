// Microbenchmark: cost of a device-wide barrier between G co-resident workgroups on MI355X
// (atomic counter in global memory, agent-scope fences), and a visibility check across XCDs.
// Decides whether a persistent multi-level kernel for the launch-bound tail of the V-cycle can
// beat one graph launch per leg (~6.5 us each).  build: hipcc --offload-arch=gfx950 -O3 tools/gridbar.hip -o gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned G, unsigned& epoch, unsigned limit) {
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __threadfence();
    epoch += 1;
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    const unsigned want = epoch * G;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want && spins < limit) {
      __builtin_amdgcn_s_sleep(1);
      ++spins;
    }
    ok = spins < limit;
    __threadfence();
  }
  __syncthreads();
  return ok != 0;
}

// every round: WG b writes data[b] = round * 1000 + b, barrier, reads data[(b + 1) % G] and checks
__global__ void bar_kernel(unsigned* ctr, int rounds, int* data, int* errors, unsigned limit) {
  unsigned epoch = 0;
  const unsigned G = gridDim.x, b = blockIdx.x;
  for (int r = 1; r <= rounds; ++r) {
    if (threadIdx.x == 0) data[b] = r * 1000 + (int)b;
    if (!grid_barrier(ctr, G, epoch, limit)) { if (threadIdx.x == 0) atomicAdd(errors, 1000000); return; }
    if (threadIdx.x == 0) {
      const int got = __hip_atomic_load(&data[(b + 1) % G], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (got != r * 1000 + (int)((b + 1) % G)) atomicAdd(errors, 1);
    }
    if (!grid_barrier(ctr, G, epoch, limit)) { if (threadIdx.x == 0) atomicAdd(errors, 1000000); return; }
  }
}

int main() {
  unsigned* ctr;
  int *data, *errors;
  hipMalloc(&ctr, 4);
  hipMalloc(&data, 4 * 1024);
  hipMalloc(&errors, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int rounds = 200;
  for (int threads : {256, 1024})
    for (int G : {1, 2, 4, 8, 16, 32, 64, 128, 256}) {
      float best = 1e30f;
      int err = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(ctr, 0, 4);
        hipMemset(errors, 0, 4);
        hipEventRecord(a);
        hipLaunchKernelGGL(bar_kernel, dim3(G), dim3(threads), 0, 0, ctr, rounds, data, errors, 2000000u);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost);
      }
      printf("threads %4d  G %3d: %.2f us per barrier (2 per round), errors %d\n", threads, G,
             best * 1e3f / (2 * rounds), err);
      fflush(stdout);
    }
  return 0;
}
