// Probe: cost of a sequential fp32 prefix over the 64 lanes of a wave as 63 dependent v_add_f32_dpp wave_shr:1 steps
// (lane i ends with ((x0 + x1) + ...) + xi, the reference's left-to-right order), and its correctness.
//   hipcc --offload-arch=gfx950 -O2 tools/dpp_chain_probe.hip -o tools/dpp_chain_probe && ./tools/dpp_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float chain64(float x) {
  float s = x;
#pragma unroll
  for (int k = 0; k < 63; ++k) asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(x));
  return s;
}

__global__ void k_chain(const float* __restrict__ in, float* __restrict__ out, int chunks, unsigned long long* ticks) {
  const int lane = threadIdx.x & 63;
  const float* row = in + (size_t)blockIdx.x * chunks * 64;
  float* orow = out + (size_t)blockIdx.x * chunks * 64;
  float carry = 0.f;
  float xs[10];
#pragma unroll
  for (int c = 0; c < 10; ++c) xs[c] = row[c * 64 + lane];  // all loads in flight before the chain starts
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int c = 0; c < 10; ++c) {
    float x = xs[c];
    if (lane == 0) x = carry + x;
    const float s = chain64(x);
    orow[c * 64 + lane] = s;
    carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 63));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && lane == 0) *ticks = t1 - t0;
}

int main() {
  const int rows = 480, chunks = 10, n = rows * chunks * 64;
  std::vector<float> h(n), ref(n), got(n);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = (float)(rand() % 100000) * 0.37f + 0.001f * (float)(rand() % 1000);
  for (int r = 0; r < rows; ++r) {
    float s = 0.f;
    for (int c = 0; c < chunks * 64; ++c) {
      s = s + h[r * chunks * 64 + c];
      ref[r * chunks * 64 + c] = s;
    }
  }
  float *din, *dout;
  unsigned long long* ticks;
  CK(hipMalloc(&din, n * 4));
  CK(hipMalloc(&dout, n * 4));
  CK(hipHostMalloc(&ticks, 8));
  CK(hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_chain, dim3(rows), dim3(64), 0, 0, din, dout, chunks, ticks);
  CK(hipEventRecord(e0, 0));
  for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_chain, dim3(rows), dim3(64), 0, 0, din, dout, chunks, ticks);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(got.data(), dout, n * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i)
    if (got[i] != ref[i]) ++bad;
  std::printf("480 rows x 640: %d mismatches vs the sequential CPU prefix; %.2f us per launch (back to back); wave 0: %llu shader clocks for %d steps = %.1f clocks/element\n",
              bad, ms * 1000.0 / 20.0, *ticks, chunks * 64, (double)*ticks / (chunks * 64));
  return bad != 0;
}
