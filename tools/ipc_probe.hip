// Feasibility probe: two processes on ONE GPU share a hipMalloc buffer through
// hipIpc; the producer pushes data with hipMemcpyAsync and publishes a flag with
// hipStreamWriteValue32; the consumer waits with hipStreamWaitValue32 (stream
// ordered, no host sync, no spinning kernel) and checks the data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("[%d] %s: %s\n", getpid(), #x, hipGetErrorString(e)); exit(2);} } while (0)

__global__ void fill(double* p, int n, double v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v + i; }
__global__ void spin_wait(volatile unsigned* flag, unsigned want, unsigned* timeout) {
  unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load((unsigned*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
    if (wall_clock64() - t0 > 200000000ull) { *timeout = 1; return; }   // ~2 s at 100 MHz
    __builtin_amdgcn_s_sleep(32);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

int main() {
  int p2c[2], c2p[2];
  if (pipe(p2c) || pipe(c2p)) return 1;
  const int N = 4096;
  pid_t pid = fork();
  if (pid == 0) {  // consumer: owns the buffer
    double* buf; unsigned* flag;
    CHECK(hipMalloc(&buf, N * 8 + 256));
    CHECK(hipMemset(buf, 0, N * 8 + 256));
    flag = (unsigned*)(buf + N);
    hipIpcMemHandle_t h;
    CHECK(hipIpcGetMemHandle(&h, buf));
    if (write(c2p[1], &h, sizeof h) != sizeof h) return 3;
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // round 1: stream memory op wait
    hipError_t e = hipStreamWaitValue32(st, flag, 1, hipStreamWaitValueGte, 0xffffffffu);
    printf("[consumer] hipStreamWaitValue32 enqueue: %s\n", hipGetErrorString(e));
    double* host = (double*)malloc(N * 8);
    if (e == hipSuccess) {
      CHECK(hipMemcpyAsync(host, buf, N * 8, hipMemcpyDeviceToHost, st));
      CHECK(hipStreamSynchronize(st));
      int bad = 0; for (int i = 0; i < N; ++i) bad += host[i] != 100.0 + i;
      printf("[consumer] round 1 (WaitValue32): mismatches %d\n", bad);
    }
    // round 2: spinning kernel on the flag
    unsigned* tmo; CHECK(hipMalloc(&tmo, 4)); CHECK(hipMemset(tmo, 0, 4));
    hipLaunchKernelGGL(spin_wait, dim3(1), dim3(64), 0, st, flag, 2u, tmo);
    CHECK(hipMemcpyAsync(host, buf, N * 8, hipMemcpyDeviceToHost, st));
    CHECK(hipStreamSynchronize(st));
    unsigned t; CHECK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < N; ++i) bad += host[i] != 200.0 + i;
    printf("[consumer] round 2 (spin kernel): timeout %u mismatches %d\n", t, bad);
    char c = 1; if (write(c2p[1], &c, 1) != 1) return 3;
    return 0;
  }
  // producer
  hipIpcMemHandle_t h;
  if (read(c2p[0], &h, sizeof h) != sizeof h) return 3;
  double* peer; 
  CHECK(hipIpcOpenMemHandle((void**)&peer, h, hipIpcMemLazyEnablePeerAccess));
  unsigned* flag = (unsigned*)(peer + N);
  double* mine; CHECK(hipMalloc(&mine, N * 8));
  hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  usleep(300000);  // let the consumer enqueue its wait first
  hipLaunchKernelGGL(fill, dim3(N / 256), dim3(256), 0, st, mine, N, 100.0);
  CHECK(hipMemcpyAsync(peer, mine, N * 8, hipMemcpyDeviceToDevice, st));
  hipError_t e = hipStreamWriteValue32(st, flag, 1, 0);
  printf("[producer] hipStreamWriteValue32: %s\n", hipGetErrorString(e));
  if (e != hipSuccess) { unsigned one = 1; CHECK(hipMemcpyAsync(flag, &one, 4, hipMemcpyHostToDevice, st)); }
  CHECK(hipStreamSynchronize(st));
  usleep(300000);
  hipLaunchKernelGGL(fill, dim3(N / 256), dim3(256), 0, st, mine, N, 200.0);
  CHECK(hipMemcpyAsync(peer, mine, N * 8, hipMemcpyDeviceToDevice, st));
  e = hipStreamWriteValue32(st, flag, 2, 0);
  if (e != hipSuccess) { unsigned two = 2; CHECK(hipMemcpyAsync(flag, &two, 4, hipMemcpyHostToDevice, st)); }
  CHECK(hipStreamSynchronize(st));
  char c; if (read(c2p[0], &c, 1) != 1) printf("[producer] consumer died\n");
  int status = 0; waitpid(pid, &status, 0);
  printf("[producer] consumer exit %d\n", WEXITSTATUS(status));
  CHECK(hipIpcCloseMemHandle(peer));
  return 0;
}
