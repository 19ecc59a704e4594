// Micro-probe (diagnostic, not product): one exchange round of the persistent LM kernels - every workgroup publishes 16 tagged
// 64-bit words, every workgroup polls the words of all G workgroups - (A) as the kernels do it today: agent-scope atomics,
// workgroups dealt to all 8 XCDs; (B) agent scope, but the G workgroups all on ONE XCD (grid of 8 G, only the blocks with
// blockIdx % 8 == 0 take part); (C) the same placement with hand-written sc0 accesses (coherent at the XCD's L2, which is
// all that workgroups of one XCD need). Reports us per round and the XCC ids the participants saw.
//   hipcc --offload-arch=gfx950 -O2 tools/xcd_exchange_probe.hip -o tools/xcd_exchange_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// SCOPE -1: hand-written sc0 accesses (the compiler gives workgroup scope no cache bits outside tgsplit mode: a workgroup lives
// on one CU and its L1 is enough - which is exactly what several workgroups of one XCD must bypass)
template <int SCOPE>
__device__ __forceinline__ void pub(unsigned long long* w, unsigned tag, float v) {
  const unsigned long long x = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
  if constexpr (SCOPE == -1)
    asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(w), "v"(x) : "memory");
  else
    __hip_atomic_store(w, x, __ATOMIC_RELAXED, SCOPE);
}
template <int SCOPE>
__device__ __forceinline__ unsigned long long ldw(const unsigned long long* w) {
  if constexpr (SCOPE == -1) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(w) : "memory");
    return v;
  } else {
    return __hip_atomic_load(w, __ATOMIC_RELAXED, SCOPE);
  }
}
template <int SCOPE>
__device__ __forceinline__ float waitw(const unsigned long long* w, unsigned tag, int* err) {
  unsigned spins = 0;
  unsigned long long v = ldw<SCOPE>(w);
  while ((unsigned)(v >> 32) != tag) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 18)) {
      *err = 1;
      break;
    }
    v = ldw<SCOPE>(w);
  }
  return __uint_as_float((unsigned)v);
}

template <int SCOPE>
__global__ __launch_bounds__(512) void k_rounds(unsigned long long* xch, int G, int stride, int rounds, unsigned tag0, float* out, int* xcc, int* err) {
  if (blockIdx.x % stride != 0) return;
  const int g = blockIdx.x / stride;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[g] = (int)(id & 15u);
  }
  float acc = (float)g;
  for (int r = 0; r < rounds; ++r) {
    unsigned long long* base = xch + (size_t)(r & 1) * G * 16;
    if (threadIdx.x < 16) pub<SCOPE>(base + g * 16 + threadIdx.x, tag0 + r + 1, acc + threadIdx.x);
    float s = 0.f;
    for (int i = threadIdx.x; i < G * 16; i += 512) s += waitw<SCOPE>(base + i, tag0 + r + 1, err);
    // workgroup total (so that every thread's next value depends on every word)
    __shared__ float red[512];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 256; o > 0; o >>= 1) {
      if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    acc = red[0] * 1e-3f;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[g] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int G = 30, rounds = 64;
  unsigned long long* xch; float* out; int *xcc, *err;
  CK(hipMalloc(&xch, 2 * G * 16 * 8)); CK(hipMemset(xch, 0, 2 * G * 16 * 8));
  CK(hipMalloc(&out, G * 4)); CK(hipMalloc(&xcc, G * 4)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipDeviceSynchronize());
  unsigned tag = 0;
  std::vector<float> ref(G);
  for (int mode = 0; mode < 3; ++mode) {
    const int stride = mode == 0 ? 1 : 8;
    float best = 1e9f;
    std::vector<float> h(G);
    std::vector<int> hx(G);
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, s));
      if (mode < 2)
        hipLaunchKernelGGL(k_rounds<__HIP_MEMORY_SCOPE_AGENT>, dim3(G * stride), dim3(512), 0, s, xch, G, stride, rounds, tag, out, xcc, err);
      else
        hipLaunchKernelGGL(k_rounds<-1>, dim3(G * stride), dim3(512), 0, s, xch, G, stride, rounds, tag, out, xcc, err);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
      tag += rounds + 8;
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    CK(hipMemcpy(h.data(), out, G * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost));
    int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    if (mode == 0) ref = h;
    bool same = true;
    for (int g = 0; g < G; ++g) same = same && (h[g] == h[0]);
    std::printf("%s: %.2f us per round (%d rounds, %d workgroups of 512), poll time-outs %d, all workgroups agree %d, value %g, XCC ids:",
                mode == 0 ? "A agent scope, 8 XCDs" : (mode == 1 ? "B agent scope, one XCD" : "C sc0 accesses (coherent at the XCD's L2), one XCD"), best * 1e3f / rounds, rounds, G,
                herr, (int)same, h[0]);
    for (int g = 0; g < G; ++g) std::printf(" %d", hx[g]);
    std::printf("\n");
  }
  return 0;
}
