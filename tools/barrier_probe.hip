// Micro-probe (diagnostic, not product): cost of a grid-wide barrier among G co-resident workgroups of 256 threads on
// gfx950, with the data exchange a phase boundary of the tracking chain needs (every block publishes 16 floats, every
// block reads all G records after the barrier), against the same exchange done as back-to-back kernel launches.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);  // agent scope by default for device code
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void k_phases(unsigned* ctr, float* rec, float* out, int phases, unsigned base) {
  const int G = gridDim.x;
  float acc = (float)threadIdx.x;
  for (int p = 0; p < phases; ++p) {
    if (threadIdx.x < 16) __hip_atomic_store(rec + ((size_t)(p & 1) * G + blockIdx.x) * 16 + threadIdx.x, acc + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    grid_barrier(ctr, base + (unsigned)(p + 1) * G);
    float s = 0.f;
    for (int b = threadIdx.x >> 4; b < G; b += 16)
      s += __hip_atomic_load(rec + ((size_t)(p & 1) * G + b) * 16 + (threadIdx.x & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc += s * 1e-9f;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_one(const float* rec_in, float* rec_out, float* out, int G) {
  float s = 0.f;
  for (int b = threadIdx.x >> 4; b < G; b += 16) s += rec_in[(size_t)b * 16 + (threadIdx.x & 15)];
  const float acc = (float)threadIdx.x + s * 1e-9f;
  if (threadIdx.x < 16) rec_out[(size_t)blockIdx.x * 16 + threadIdx.x] = acc;
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int phases = 64;
  unsigned* ctr; float *rec, *out;
  CK(hipMalloc(&ctr, 4)); CK(hipMemset(ctr, 0, 4));
  CK(hipMalloc(&rec, 2 * 512 * 16 * 4)); CK(hipMemset(rec, 0, 2 * 512 * 16 * 4));
  CK(hipMalloc(&out, 512 * 256 * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {16, 60, 120, 240}) {
    unsigned base = 0;
    CK(hipMemsetAsync(ctr, 0, 4, s));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_phases, dim3(G), dim3(256), 0, s, ctr, rec, out, phases, base);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      base += (unsigned)phases * G;
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    float bestk = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, s));
      for (int p = 0; p < phases; ++p)
        hipLaunchKernelGGL(k_one, dim3(G), dim3(256), 0, s, rec + (size_t)(p & 1) * 512 * 16, rec + (size_t)((p + 1) & 1) * 512 * 16, out, G);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < bestk) bestk = ms;
    }
    std::printf("G=%3d blocks: grid barrier phase %.2f us | kernel-per-phase %.2f us\n", G, best * 1e3f / phases, bestk * 1e3f / phases);
  }
  return 0;
}
