// Probe: can the host store directly into device memory (fine-grained VRAM through the PCIe BAR), and what does a kernel
// pay to read a small parameter block from it compared with pinned host memory?
//   hipcc --offload-arch=gfx950 -O2 tools/hostvram_probe.hip -o tools/hostvram_probe && ./tools/hostvram_probe
#include <hip/hip_runtime.h>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_read(const int* __restrict__ src, int nwords, int* out, unsigned long long* ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int acc = 0;
  for (int i = 0; i < nwords; ++i) acc += __builtin_nontemporal_load(src + i);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *out = acc;
    *ticks = t1 - t0;
  }
}

static int run(const char* name, int* buf_dev_visible, int* host_ptr) {
  int* out;
  unsigned long long* ticks;
  CK(hipHostMalloc(&out, sizeof(int)));
  CK(hipHostMalloc(&ticks, sizeof(unsigned long long)));
  double sum = 0;
  int bad = 0;
  for (int it = 0; it < 200; ++it) {
    for (int i = 0; i < 44; ++i) host_ptr[i] = it * 100 + i;  // host stores (posted writes for BAR memory)
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    hipLaunchKernelGGL(k_read, dim3(60), dim3(256), 0, 0, buf_dev_visible, 44, out, ticks);
    CK(hipDeviceSynchronize());
    int expect = 0;
    for (int i = 0; i < 44; ++i) expect += it * 100 + i;
    if (*out != expect) ++bad;
    if (it >= 20) sum += (double)*ticks;
  }
  std::printf("%-28s kernel-side read of 44 words: %.2f us mean, %d stale reads of 200\n", name, sum / 180.0 * 0.01, bad);
  return 0;
}

int main() {
  int* pinned;
  CK(hipHostMalloc(&pinned, 4096));
  if (run("pinned host memory", pinned, pinned)) return 1;
  int* fg = nullptr;
  hipError_t e = hipExtMallocWithFlags((void**)&fg, 4096, hipDeviceMallocFinegrained);
  std::printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
  if (e == hipSuccess) {
    // the host access may fault: catch it (a forked child cannot be used, GPU mappings are not inherited)
    static sigjmp_buf jb;
    struct sigaction sa {}, old {};
    sa.sa_handler = [](int) { siglongjmp(jb, 1); };
    sigaction(SIGSEGV, &sa, &old);
    bool ok = false;
    if (sigsetjmp(jb, 1) == 0) {
      volatile int* v = fg;
      v[0] = 1;
      ok = v[0] == 1;
    }
    sigaction(SIGSEGV, &old, nullptr);
    if (ok) {
      std::printf("host can store to fine-grained device memory\n");
      if (run("fine-grained device memory", fg, fg)) return 1;
    } else {
      std::printf("host access to fine-grained device memory faults\n");
    }
  }
  int* dev = nullptr;
  CK(hipMalloc(&dev, 4096));
  int* scratch;
  CK(hipHostMalloc(&scratch, 4096));
  CK(hipMemcpy(dev, scratch, 4096, hipMemcpyHostToDevice));
  // reference point: plain device memory (host cannot write it; the kernel reads zeros)
  {
    int* out;
    unsigned long long* ticks;
    CK(hipHostMalloc(&out, sizeof(int)));
    CK(hipHostMalloc(&ticks, sizeof(unsigned long long)));
    double sum = 0;
    for (int it = 0; it < 200; ++it) {
      hipLaunchKernelGGL(k_read, dim3(60), dim3(256), 0, 0, dev, 44, out, ticks);
      CK(hipDeviceSynchronize());
      if (it >= 20) sum += (double)*ticks;
    }
    std::printf("%-28s kernel-side read of 44 words: %.2f us mean\n", "plain device memory", sum / 180.0 * 0.01);
  }
  return 0;
}
