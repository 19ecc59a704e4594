// Micro-probe (diagnostic, not product): effective clock for a single wave's dependent fp32 add chain, dependent
// LDS read->add->write chain, dependent global-load chain, and empty-kernel launch cost, in the idle/bursty regime
// this pipeline runs in.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_addchain(float* out, int n) {
  float s = out[threadIdx.x];
  for (int i = 0; i < n; ++i) s = s + 1.0f;
  out[threadIdx.x] = s;
}
__global__ void k_ldschain(float* out, int n) {
  __shared__ float t[64 * 33];
  for (int i = threadIdx.x; i < 64 * 33; i += 64) t[i] = 1.0f;
  __syncthreads();
  float s = 0.f;
  volatile float* p = t + threadIdx.x;
#pragma unroll 8
  for (int i = 0; i < n; ++i) {
    s = p[(i & 31) * 64 % (64 * 32)] + s;
    p[(i & 31) * 64 % (64 * 32)] = s;
  }
  out[threadIdx.x] = s;
}
__global__ void k_chase(const int* next, int* out, int n) {
  int j = threadIdx.x;
  for (int i = 0; i < n; ++i) j = next[j];
  out[threadIdx.x] = j;
}
__global__ void k_empty() {}
template <int CHAINS>
__global__ void k_dep(unsigned long long* out, float* sink, int reps) {
  float s[CHAINS];
  for (int c = 0; c < CHAINS; ++c) s[c] = sink[threadIdx.x + c];
  const float one = sink[100];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < reps; ++i) {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) s[c] = s[c] + one;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int c = 0; c < CHAINS; ++c) sink[threadIdx.x + c] = s[c];
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
__global__ void k_clock(unsigned long long* out, int n, float* sink) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  float s = sink[threadIdx.x];
  for (int i = 0; i < n; ++i) s = s + 1.0f;
  sink[threadIdx.x] = s;
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

template <class F>
double time_us(F f, int reps) {
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) f();
  hipDeviceSynchronize();
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  hipMemset(d, 0, 4096);
  const int N = 1 << 20;
  std::vector<int> h(N);
  for (int i = 0; i < N; ++i) h[i] = (int)(((long long)i * 7919 + 104729) % N);
  int *dn, *dout;
  hipMalloc(&dn, N * 4);
  hipMalloc(&dout, 4096);
  hipMemcpy(dn, h.data(), N * 4, hipMemcpyHostToDevice);
  hipStream_t s;
  hipStreamCreate(&s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto ev = [&](auto f) {
    hipEventRecord(e0, s);
    f();
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3;
  };
  for (int rep = 0; rep < 3; ++rep) {
    double a = ev([&] { hipLaunchKernelGGL(k_addchain, dim3(1), dim3(64), 0, s, d, 100000); });
    double l = ev([&] { hipLaunchKernelGGL(k_ldschain, dim3(1), dim3(64), 0, s, d, 20000); });
    double c = ev([&] { hipLaunchKernelGGL(k_chase, dim3(1), dim3(64), 0, s, dn, dout, 2000); });
    double e = ev([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s); });
    double e10 = ev([&] { for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s); });
    printf("addchain 100k: %.1f us (%.2f ns/add)  ldschain 20k: %.1f us (%.1f ns/step)  chase 2000: %.1f us (%.0f ns/load)  empty: %.1f us  10 empties: %.1f us\n",
           a, a * 1e3 / 100000, l, l * 1e3 / 20000, c, c * 1e3 / 2000, e, e10);
  }
  {
    unsigned long long* dc; hipMalloc(&dc, 16);
    for (int n : {2000, 20000, 200000}) {
      hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, s, dc, n, d);
      unsigned long long h[2]; hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
      printf("k_clock n=%d: shader cycles %llu, realtime ticks(100MHz) %llu -> %.0f MHz, %.2f cycles/iter\n", n, h[0], h[1], (double)h[0] / ((double)h[1] / 100.0), (double)h[0] / n);
    }
  }
  {
    unsigned long long* dc; hipMalloc(&dc, 16); unsigned long long h;
    hipLaunchKernelGGL(k_dep<1>, dim3(1), dim3(64), 0, s, dc, d, 100); hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
    printf("dependent v_add chain (1 wave, 64 lanes): %.2f cycles/add\n", (double)h / 6400);
    hipLaunchKernelGGL(k_dep<1>, dim3(1), dim3(16), 0, s, dc, d, 100); hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
    printf("dependent v_add chain (16 lanes): %.2f cycles/add\n", (double)h / 6400);
    hipLaunchKernelGGL(k_dep<2>, dim3(1), dim3(64), 0, s, dc, d, 100); hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
    printf("2 independent chains: %.2f cycles/add-pair\n", (double)h / 6400);
    hipLaunchKernelGGL(k_dep<4>, dim3(1), dim3(64), 0, s, dc, d, 100); hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost);
    printf("4 independent chains: %.2f cycles/4 adds\n", (double)h / 6400);
  }
  double host1 = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s); }, 2000);
  printf("host launch cost (async, queued): %.2f us per launch\n", host1);
  double sync1 = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s); hipStreamSynchronize(s); }, 500);
  printf("launch + hipStreamSynchronize round trip: %.2f us\n", sync1);
  hipEvent_t e2;
  hipEventCreateWithFlags(&e2, hipEventDisableTiming);
  double sync2 = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s); hipEventRecord(e2, s); hipEventSynchronize(e2); }, 500);
  printf("launch + event record + hipEventSynchronize: %.2f us\n", sync2);
  volatile unsigned* flag;
  hipHostMalloc((void**)&flag, 64, hipHostMallocDefault);
  *flag = 0;
  unsigned gen = 0;
  double sync3 = time_us([&] {
    ++gen;
    hipLaunchKernelGGL(k_empty, dim3(64), dim3(256), 0, s);
    hipStreamWriteValue32(s, (void*)flag, gen, 0);
    while (*flag != gen) {}
  }, 500);
  printf("launch + hipStreamWriteValue32 + host spin: %.2f us\n", sync3);
  return 0;
}
