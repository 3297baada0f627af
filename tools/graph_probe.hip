// Can hipStreamWaitValue32 / hipStreamWriteValue32 / peer hipMemcpyAsync be captured
// into a hipGraph and replayed?  Also times the host cost of the stream memory ops.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(2);} } while (0)
__global__ void bump(double* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0; }
int main() {
  double *a, *b; unsigned* flags;
  const int N = 4096;
  CHECK(hipMalloc(&a, N * 8)); CHECK(hipMalloc(&b, N * 8)); CHECK(hipMalloc(&flags, 256));
  CHECK(hipMemset(a, 0, N * 8)); CHECK(hipMemset(b, 0, N * 8)); CHECK(hipMemset(flags, 0, 256));
  hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  // host cost of stream memory ops
  unsigned one = 1; CHECK(hipMemcpy(flags, &one, 4, hipMemcpyHostToDevice));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 2000; ++i) { CHECK(hipStreamWriteValue32(st, flags + 1, i, 0)); }
  auto t1 = std::chrono::steady_clock::now();
  CHECK(hipStreamSynchronize(st));
  auto t1b = std::chrono::steady_clock::now();
  for (int i = 0; i < 2000; ++i) { CHECK(hipStreamWaitValue32(st, flags, 1, hipStreamWaitValueGte, 0xffffffffu)); }
  auto t2 = std::chrono::steady_clock::now();
  CHECK(hipStreamSynchronize(st));
  auto t2b = std::chrono::steady_clock::now();
  for (int i = 0; i < 2000; ++i) { hipLaunchKernelGGL(bump, dim3(16), dim3(256), 0, st, a, N); }
  auto t3 = std::chrono::steady_clock::now();
  CHECK(hipStreamSynchronize(st));
  auto t3b = std::chrono::steady_clock::now();
  for (int i = 0; i < 2000; ++i) { CHECK(hipMemcpyAsync(b, a, 4096, hipMemcpyDeviceToDevice, st)); }
  auto t4 = std::chrono::steady_clock::now();
  CHECK(hipStreamSynchronize(st));
  auto t4b = std::chrono::steady_clock::now();
  auto us = [](auto x, auto y) { return std::chrono::duration<double, std::micro>(y - x).count() / 2000; };
  printf("host enqueue cost per op (us) / incl. execution: WriteValue32 %.2f / %.2f   WaitValue32 %.2f / %.2f   kernel %.2f / %.2f   memcpyD2D(4KB) %.2f / %.2f\n",
         us(t0, t1), us(t0, t1b), us(t1b, t2), us(t1b, t2b), us(t2b, t3), us(t2b, t3b), us(t3b, t4), us(t3b, t4b));
  // capture
  hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  printf("BeginCapture: %s\n", hipGetErrorString(e));
  e = hipStreamWaitValue32(st, flags, 1, hipStreamWaitValueGte, 0xffffffffu);
  printf("  capture WaitValue32: %s\n", hipGetErrorString(e));
  hipLaunchKernelGGL(bump, dim3(16), dim3(256), 0, st, a, N);
  e = hipMemcpyAsync(b, a, N * 8, hipMemcpyDeviceToDevice, st);
  printf("  capture memcpy: %s\n", hipGetErrorString(e));
  e = hipStreamWriteValue32(st, flags + 2, 7, 0);
  printf("  capture WriteValue32: %s\n", hipGetErrorString(e));
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(st, &g);
  printf("EndCapture: %s\n", hipGetErrorString(e));
  if (e == hipSuccess && g) {
    hipGraphExec_t ex;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    printf("Instantiate: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
      size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn); printf("nodes: %zu\n", nn);
      for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(ex, st));
      CHECK(hipStreamSynchronize(st));
      unsigned f; CHECK(hipMemcpy(&f, flags + 2, 4, hipMemcpyDeviceToHost));
      double v; CHECK(hipMemcpy(&v, b, 8, hipMemcpyDeviceToHost));
      printf("after 3 replays: flag2=%u b[0]=%.1f (expect 7, 2003)\n", f, v);
    }
  }
  return 0;
}
