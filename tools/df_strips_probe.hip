// Stand-alone timing of the keyline-driven distance-field kernel on synthetic raster-ordered keylines (no API, no other
// streams): N back-to-back launches between two events, plus s_memrealtime stamps of one workgroup's phases.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../include -I../rebvio_amd/csrc df_strips_probe.hip -o df_strips_probe
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#define RH_DF_PROBE 1
#include "../rebvio_amd/csrc/detect.hip"

// profiler hooks of common.hpp (api.hip defines them in the library)
namespace rh {
void prof_begin(hipStream_t, const char*) {}
void prof_end(hipStream_t) {}
void prof_group_begin(hipStream_t, const char*, int) {}
void prof_group_end(hipStream_t) {}
}  // namespace rh

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  using namespace rh;
  const int W = argc > 1 ? std::atoi(argv[1]) : 640, H = argc > 2 ? std::atoi(argv[2]) : 480, N = argc > 3 ? std::atoi(argv[3]) : 15000;
  KParams p{};
  p.rows = H; p.cols = W; p.kmax = N + 1000; p.df_nr = 80; p.search_range = 40;
  // keylines: N random pixels in raster order, random gradient directions
  std::mt19937 rng(5);
  std::vector<int> pix(N);
  for (auto& v : pix) v = (int)(rng() % (unsigned)((H - 4) * (W - 4)));
  std::sort(pix.begin(), pix.end());
  pix.erase(std::unique(pix.begin(), pix.end()), pix.end());
  const int n = (int)pix.size();
  std::vector<float2> pos(n), grad(n), unit(n);
  std::vector<float> gn(n);
  std::vector<int> row_start(H + 1, 0);
  std::uniform_real_distribution<float> U(-0.5f, 0.5f), A(0.f, 6.2831853f), G(5.f, 50.f);
  for (int i = 0; i < n; ++i) {
    const int r = pix[i] / (W - 4) + 2, c = pix[i] % (W - 4) + 2;
    pos[i] = make_float2(c + U(rng), r + U(rng));
    const float a = A(rng), g = G(rng);
    grad[i] = make_float2(g * std::cos(a), g * std::sin(a));
    gn[i] = std::sqrt(grad[i].x * grad[i].x + grad[i].y * grad[i].y);
    unit[i] = make_float2(grad[i].x / gn[i], grad[i].y / gn[i]);
    row_start[r + 1]++;
  }
  for (int r = 0; r < H; ++r) row_start[r + 1] += row_start[r];
  MapDev m{};
  const size_t M = ((size_t)p.kmax + 1023) / 1024 * 1024;
  CK(hipMalloc(&m.pos, M * 8)); CK(hipMalloc(&m.grad, M * 8)); CK(hipMalloc(&m.unit, M * 8)); CK(hipMalloc(&m.gnorm, M * 4));
  CK(hipMalloc(&m.df, (size_t)W * H * 4)); CK(hipMalloc(&m.row_start, (H + 1) * 4)); CK(hipMalloc(&m.st, sizeof(MapState)));
  CK(hipMalloc(&m.mask, (size_t)W * H * 4));
  CK(hipMemset(m.mask, 0xFF, (size_t)W * H * 4));
  CK(hipMalloc(&m.id_prev, M * 4)); CK(hipMalloc(&m.id_next, M * 4));
  const DfGrid dg = df_grid(H, W);
  CK(hipMalloc(&m.tile_cnt, (size_t)dg.ntx * dg.nty * 4));
  CK(hipMalloc(&m.tile_list, (size_t)dg.ntx * dg.nty * kDfTileCap * 2 * sizeof(float4)));
  CK(hipMemcpy(m.pos, pos.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(m.grad, grad.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(m.unit, unit.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(m.gnorm, gn.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(m.row_start, row_start.data(), (H + 1) * 4, hipMemcpyHostToDevice));
  MapState st{};
  st.n = n; st.total = n; st.gmin_bits = 0; st.gmax_bits = 0;  // auto threshold 0: nothing is skipped
  CK(hipMemcpy(m.st, &st, sizeof(st), hipMemcpyHostToDevice));
  DetState ds{0.01f, n, 0.0f, 0};
  DetState* d_ds;
  CK(hipMalloc(&d_ds, sizeof(ds)));
  CK(hipMemcpy(d_ds, &ds, sizeof(ds), hipMemcpyHostToDevice));
  unsigned long long* d_stamps;
  CK(hipMalloc(&d_stamps, 64 * 8));
  g_df_stamps_host = d_stamps;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<unsigned> ref;
  {  // joinEdges + binning: timed with the counter reset it needs in front (the emit kernel does that in the pipeline)
    auto join = [&] {
      (void)hipMemsetAsync(m.tile_cnt, 0, (size_t)dg.ntx * dg.nty * 4, s);
      hipLaunchKernelGGL(k_join_edges, dim3(div_up(p.kmax, 256)), dim3(256), (size_t)dg.ntx * dg.nty * sizeof(int), s, p, m, dg.T, dg.ntx, dg.nty);
    };
    for (int i = 0; i < 10; ++i) join();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 200; ++i) join();
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<int> cnt((size_t)dg.ntx * dg.nty);
    CK(hipMemcpy(cnt.data(), m.tile_cnt, cnt.size() * 4, hipMemcpyDeviceToHost));
    long tot = 0;
    int mx = 0;
    for (int v : cnt) { tot += v; mx = std::max(mx, v); }
    st.gmin_bits = 0; st.gmax_bits = 0;  // (joinEdges accumulated min/max: back to "no threshold")
    CK(hipMemcpy(m.st, &st, sizeof(st), hipMemcpyHostToDevice));
    std::printf("memset + k_join_edges (with binning): %.2f us; %d tiles of %d, %ld entries (%.2f per keyline), fullest %d\n", ms * 1e3 / 200,
                dg.ntx * dg.nty, dg.T, tot, (double)tot / n, mx);
  }
  for (int rep = 0; rep < 2; ++rep) {
    // reference field from the scatter kernel (first round), then the strips kernel
    for (int i = 0; i < 20; ++i) launch_df_build(s, p, m, d_ds, true);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    const int L = 200;
    for (int i = 0; i < L; ++i) launch_df_build(s, p, m, d_ds, true);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long st_h[64] = {0};
    CK(hipMemcpy(st_h, d_stamps, sizeof(st_h), hipMemcpyDeviceToHost));
    std::printf("%dx%d n=%d: %.2f us per launch;", W, H, n, ms * 1e3 / L);
    if (st_h[0]) {
      std::printf(" stamps (us since start, one workgroup):");
      for (int i = 1; i < 24 && st_h[i]; ++i) std::printf(" %.2f", (double)(st_h[i] - st_h[0]) * 0.01);
    }
    std::printf("\n");
    if (rep == 0) {
      ref.resize((size_t)W * H);
      CK(hipMemcpy(ref.data(), m.df, ref.size() * 4, hipMemcpyDeviceToHost));
      // scatter reference
      CK(hipMemset(m.df, 0xFF, (size_t)W * H * 4));
      hipLaunchKernelGGL(k_df_build, dim3(128), dim3(256), 0, s, p, m, (const DetState*)d_ds);
      CK(hipStreamSynchronize(s));
      std::vector<unsigned> sc((size_t)W * H);
      CK(hipMemcpy(sc.data(), m.df, sc.size() * 4, hipMemcpyDeviceToHost));
      size_t bad = 0;
      for (size_t i = 0; i < sc.size(); ++i) bad += sc[i] != ref[i];
      std::printf("  strips vs scatter kernel: %zu differing cells of %zu\n", bad, sc.size());
    }
  }
  return 0;
}
