// Probe: host cost and device throughput of a 12-kernel, two-stream frame chain enqueued kernel by kernel vs. launched as one
// captured hipGraph (with and without per-launch kernel-parameter updates). Build: hipcc --offload-arch=gfx950 -O2 -o graph_probe graph_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(float* p, int n, int spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = i < n ? p[i] : 0.f;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) v += 1e-9f;
  if (i < n) p[i] = v;
}
int main() {
  const int n = 1 << 16, iters = 3000, spin = 500;  // 100 MHz clock: 5 us per kernel
  float *a, *b;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t e_fork, e_join;
  CK(hipEventCreateWithFlags(&e_fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e_join, hipEventDisableTiming));
  auto frame = [&](hipStream_t x, hipStream_t y, float* pa, float* pb) {
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, x, pa, n, spin);
    hipEventRecord(e_fork, x);
    hipStreamWaitEvent(y, e_fork, 0);
    for (int k = 0; k < 7; ++k) hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, y, pb, n, spin);
    hipEventRecord(e_join, y);
    hipStreamWaitEvent(x, e_join, 0);
  };
  // direct
  for (int i = 0; i < 200; ++i) frame(s0, s1, a, b);
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  double host = 0;
  for (int i = 0; i < iters; ++i) {
    auto h0 = std::chrono::steady_clock::now();
    frame(s0, s1, a, b);
    host += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
  }
  CK(hipDeviceSynchronize());
  double tot = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  std::printf("direct : host %.1f us/frame, wall %.1f us/frame (12 kernels x 5 us, 5 + 7 on two streams, serial chain = 60 us)\n", host / iters, tot / iters);
  // graph
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
  frame(s0, s1, a, b);
  CK(hipStreamEndCapture(s0, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 200; ++i) CK(hipGraphLaunch(ge, s0));
  CK(hipDeviceSynchronize());
  t0 = std::chrono::steady_clock::now();
  host = 0;
  for (int i = 0; i < iters; ++i) {
    auto h0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, s0));
    host += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
  }
  CK(hipDeviceSynchronize());
  tot = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  std::printf("graph  : host %.1f us/frame, wall %.1f us/frame\n", host / iters, tot / iters);
  // graph with parameter updates of every kernel node
  size_t nn = 0;
  CK(hipGraphGetNodes(g, nullptr, &nn));
  std::vector<hipGraphNode_t> nodes(nn);
  CK(hipGraphGetNodes(g, nodes.data(), &nn));
  std::vector<hipGraphNode_t> kn;
  for (auto nd : nodes) {
    hipGraphNodeType t;
    CK(hipGraphNodeGetType(nd, &t));
    if (t == hipGraphNodeTypeKernel) kn.push_back(nd);
  }
  std::printf("graph nodes %zu, kernel nodes %zu\n", nn, kn.size());
  t0 = std::chrono::steady_clock::now();
  host = 0;
  for (int i = 0; i < iters; ++i) {
    auto h0 = std::chrono::steady_clock::now();
    for (auto nd : kn) {
      hipKernelNodeParams kp;
      CK(hipGraphKernelNodeGetParams(nd, &kp));
      float* pp = (i & 1) ? a : b;
      int nnv = n, sp = spin;
      void* args[3] = {&pp, &nnv, &sp};
      kp.kernelParams = args;
      CK(hipGraphExecKernelNodeSetParams(ge, nd, &kp));
    }
    CK(hipGraphLaunch(ge, s0));
    host += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
  }
  CK(hipDeviceSynchronize());
  tot = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  std::printf("graph + 12 param updates: host %.1f us/frame, wall %.1f us/frame\n", host / iters, tot / iters);
  return 0;
}
