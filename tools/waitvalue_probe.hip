// Micro-probe (diagnostic, not product): is hipStreamWaitValue32 usable as a host doorbell on this stack, and how long is
// host write -> start of the kernel queued behind the wait? Compared with launching the kernel after the write.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

__global__ void k_stamp(volatile unsigned long long* out) {
  if (threadIdx.x == 0) *out = 1ull;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned* flag = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&flag, 64, hipMallocSignalMemory);
  std::printf("hipExtMallocWithFlags(signal) -> %s\n", hipGetErrorString(e));
  if (e != hipSuccess) CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
  volatile unsigned long long* done = nullptr;
  CK(hipHostMalloc((void**)&done, 64, hipHostMallocDefault));
  *flag = 0;
  double best_wait = 1e9, best_launch = 1e9;
  for (int it = 1; it <= 200; ++it) {
    *done = 0;
    hipError_t w = hipStreamWaitValue32(s, flag, (unsigned)it, hipStreamWaitValueEq, 0xFFFFFFFFu);
    if (w != hipSuccess) { std::printf("hipStreamWaitValue32: %s\n", hipGetErrorString(w)); return 1; }
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, done);
    std::this_thread::sleep_for(std::chrono::microseconds(200));  // the wait is parked in the queue
    const auto t0 = std::chrono::steady_clock::now();
    __atomic_store_n(flag, (unsigned)it, __ATOMIC_RELEASE);
    while (*done == 0) {}
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (it > 20 && us < best_wait) best_wait = us;
    CK(hipStreamSynchronize(s));
    // reference: launch after the "glue"
    *done = 0;
    const auto t1 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_stamp, dim3(1), dim3(64), 0, s, done);
    while (*done == 0) {}
    const double us2 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count();
    if (it > 20 && us2 < best_launch) best_launch = us2;
    CK(hipStreamSynchronize(s));
  }
  std::printf("host write -> kernel behind hipStreamWaitValue32 done: best %.2f us\n", best_wait);
  std::printf("launch -> kernel done (idle stream):                  best %.2f us\n", best_launch);
  return 0;
}
