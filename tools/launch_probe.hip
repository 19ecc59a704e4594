// Host cost of one kernel launch on this runtime, by entry point: hipLaunchKernelGGL, hipExtLaunchKernelGGL with a stop event,
// hipModuleLaunchKernel on a cached hipFunction_t (kernelParams array / packed argument buffer).
// hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_probe tools/launch_probe.hip && /tmp/launch_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstring>
struct Big { float v[24]; };
__global__ void k_small(int* p, int a, int b) { if (p && a == -1) p[0] = b; }
__global__ void k_big(int* p, Big x, Big y, const float* q, int a, int b, int c, int d) { if (p && a == -1) p[0] = (int)x.v[0] + (int)y.v[1] + b + c + d + (q ? 1 : 0); }
#define CHK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
template <typename F> double per_call_us(int n, hipStream_t s, F f) {
  (void)hipStreamSynchronize(s);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f(i);
  const auto t1 = std::chrono::steady_clock::now();
  (void)hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
  hipStream_t s;
  CHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t ev[64];
  for (auto& e : ev) CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  Big x{}, y{};
  const int N = 400;
  for (int rep = 0; rep < 3; ++rep) {
    const double a = per_call_us(N, s, [&](int) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, s, (int*)nullptr, 0, 1); });
    const double b = per_call_us(N, s, [&](int) { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, (int*)nullptr, x, y, (const float*)nullptr, 0, 1, 2, 3); });
    const double c = per_call_us(N, s, [&](int i) { hipExtLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, nullptr, ev[i & 63], 0, (int*)nullptr, x, y, (const float*)nullptr, 0, 1, 2, 3); });
    const double d = per_call_us(N, s, [&](int i) { hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, s, (int*)nullptr, x, y, (const float*)nullptr, 0, 1, 2, 3); (void)hipEventRecord(ev[i & 63], s); });
    hipFunction_t f;
    CHK(hipGetFuncBySymbol(&f, reinterpret_cast<const void*>(&k_big)));
    int* p = nullptr; const float* q = nullptr; int i0 = 0, i1 = 1, i2 = 2, i3 = 3;
    void* args[] = {&p, &x, &y, &q, &i0, &i1, &i2, &i3};
    const double e = per_call_us(N, s, [&](int) { (void)hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, s, args, nullptr); });
    struct __attribute__((packed, aligned(8))) Packed { int* p; Big x; Big y; const float* q; int a, b, c, d; } pk{nullptr, x, y, nullptr, 0, 1, 2, 3};
    size_t sz = sizeof(pk);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pk, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    const double g = per_call_us(N, s, [&](int) { (void)hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, s, nullptr, extra); });
    hipEvent_t w = ev[0];
    CHK(hipEventRecord(w, s));
    CHK(hipStreamSynchronize(s));
    hipStream_t s2;
    CHK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const double h = per_call_us(N, s2, [&](int) { (void)hipStreamWaitEvent(s2, w, 0); });  // an event that has completed
    const double q2 = per_call_us(N, s2, [&](int) { (void)hipEventQuery(w); });
    CHK(hipStreamDestroy(s2));
    std::printf("us per call: GGL small %.2f | GGL 220-byte args %.2f | ExtGGL + stop event %.2f | GGL + hipEventRecord %.2f | hipModuleLaunchKernel kernelParams %.2f, packed buffer %.2f | "
                "hipStreamWaitEvent(completed) %.2f | hipEventQuery %.2f\n", a, b, c, d, e, g, h, q2);
  }
  return 0;
}
