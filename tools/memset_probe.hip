// Is hipMemset on device memory synchronous with the host on this runtime? (alloc_map clears fresh edge maps with it while the
// context's non-blocking streams are running.)  hipcc --offload-arch=gfx950 tools/memset_probe.hip -o tools/memset_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
  const size_t n = (size_t)1 << 30;
  char* d = nullptr;
  if (hipMalloc(&d, n) != hipSuccess) return 1;
  (void)hipMemset(d, 0, n);
  (void)hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    (void)hipMemset(d, 0xFF, n);
    const auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    const auto t2 = std::chrono::steady_clock::now();
    std::printf("hipMemset of 1 GiB: call returned after %.1f us, device idle %.1f us later\n",
                std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count());
  }
  for (int rep = 0; rep < 3; ++rep) {
    const auto t0 = std::chrono::steady_clock::now();
    (void)hipMemset(d, 0xFF, 1228800);
    const auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    const auto t2 = std::chrono::steady_clock::now();
    std::printf("hipMemset of 1.2 MB: call returned after %.1f us, device idle %.1f us later\n",
                std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count());
  }
  return 0;
}
