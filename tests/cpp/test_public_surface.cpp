// Exercises the parts of the reference's public C++ surface that no ros_rebvio caller reaches (missing #1):
// rebvio::DistanceField (core.hpp:20-79), Core::testfk / calculatefJ / updateInverseDepthARLU (core.hpp:122,138,187),
// EdgeMap::searchMatch (edge_map.hpp:93-94), ScaleSpace / FastGaussian (scale_space.hpp:22-96), the REBVIO_TIMER macros
// (util/timer.hpp:18-32). Results are dumped as raw arrays; tests/test_host_api.py compares them with the CPU oracle.
//   test_public_surface frames.u8 W H fm cx cy keylines_ref keylines_max outdir
#define TIMER
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "rebvio/core.hpp"
#include "rebvio/edge_detector.hpp"
#include "rebvio/scale_space.hpp"
#include "rebvio/util/timer.hpp"

namespace {
template <typename T>
void dump(const std::string& path, const T* p, size_t n) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(p), (std::streamsize)(n * sizeof(T)));
}
int timed_section(int x) {
  REBVIO_TIMER_TICK();
  REBVIO_NAMED_TIMER_TICK(inner);
  x = x * 3 + 1;
  REBVIO_NAMED_TIMER_TOCK(inner);
  REBVIO_TIMER_TOCK();
  return x;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 10) return 2;
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
  const float fm = std::atof(argv[4]), cx = std::atof(argv[5]), cy = std::atof(argv[6]);
  const std::string out = argv[9];
  std::vector<unsigned char> buf((size_t)W * H * 2);
  std::ifstream f(argv[1], std::ios::binary);
  if (!f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size())) return 2;

  auto cam = std::make_shared<rebvio::Camera>(H, W, fm, fm, cx, cy);
  auto dcfg = std::make_shared<rebvio::EdgeDetectorConfig>();
  dcfg->keylines_ref = std::atoi(argv[7]);
  dcfg->keylines_max = std::atoi(argv[8]);
  rebvio::EdgeDetector detector(cam, dcfg);
  rebvio::Core core(cam);

  rebvio::EdgeMap::SharedPtr maps[2];
  for (int i = 0; i < 2; ++i) {
    cv::Mat u8(H, W, CV_8UC1, buf.data() + (size_t)i * W * H);
    rebvio::types::Image img{(uint64_t)i * 50000ull, cv::Mat()};
    u8.convertTo(img.data, CV_32FC1, 3.0);  // what Rebvio::imageCallback hands the detector (rebvio.cpp:43)
    maps[i] = detector.detect(img);
  }
  const int n0 = maps[0]->size(), n1 = maps[1]->size();
  if (n0 < 100 || n1 < 100) return 3;
  dump(out + "/kl0.bin", maps[0]->keylines().data(), (size_t)n0);
  dump(out + "/kl1.bin", maps[1]->keylines().data(), (size_t)n1);

  // --- rebvio::DistanceField: build on the device, read through operator[] ---
  rebvio::DistanceField df(H, W, core.config()->search_range);
  if (df[0].id != -1) return 4;  // never built: empty
  df.build(maps[1]);
  if (df.map() != maps[1]) return 4;
  std::vector<int> ids((size_t)W * H), dist((size_t)W * H);
  for (int i = 0; i < W * H; ++i) {
    ids[i] = df[i].id;
    dist[i] = df[i].distance;
  }
  dump(out + "/df_id.bin", ids.data(), ids.size());
  dump(out + "/df_dist.bin", dist.data(), dist.size());

  // --- Core::calculatefJ / testfk on the field of frame 1, queried with the keylines of frame 0 at their own pixel ---
  core.buildDistanceField(maps[1]);
  const int K = 400;
  std::vector<float> fj((size_t)K * 4);
  std::vector<int> fji((size_t)K * 3);
  for (int k = 0; k < K; ++k) {
    rebvio::types::KeyLine kl = (*maps[0])[(k * 37) % n0];
    kl.sigma_rho = 0.5f + 0.01f * k;
    const int x = (int)(kl.pos[0] + 0.5f), y = (int)(kl.pos[1] + 0.5f);
    rebvio::types::Float dx = -1, dy = -1, fi = 123.0f;
    int mnum = 0;
    const rebvio::types::Float r = core.calculatefJ(maps[0], y * W + x, dx, dy, kl, kl.pos[0], kl.pos[1], mnum, fi);
    fj[k * 4 + 0] = r; fj[k * 4 + 1] = dx; fj[k * 4 + 2] = dy; fj[k * 4 + 3] = fi;
    fji[k * 3 + 0] = mnum; fji[k * 3 + 1] = kl.match_id_forward;
    fji[k * 3 + 2] = rebvio::Core::testfk((*maps[1])[(k * 11) % n1], kl, core.config()->match_treshold) ? 1 : 0;
  }
  dump(out + "/fj.bin", fj.data(), fj.size());
  dump(out + "/fji.bin", fji.data(), fji.size());

  // --- EdgeMap::searchMatch: keylines of frame 1 searched in frame 0 ---
  const rebvio::types::Vector3f vel = TooN::makeVector(0.004f, -0.002f, 0.003f);
  rebvio::types::Matrix3f Rvel = TooN::Zeros, Rback = TooN::Zeros;
  for (int i = 0; i < 3; ++i) {
    Rvel(i, i) = 1e-6f * (i + 1);
    Rback(i, i) = 1.0f;
  }
  Rback(0, 2) = 2e-3f;
  Rback(2, 0) = -2e-3f;
  std::vector<int> sm(K);
  for (int k = 0; k < K; ++k) {
    rebvio::types::KeyLine q = (*maps[1])[(k * 29) % n1];
    q.rho = 0.4f + 0.002f * k;
    q.sigma_rho = 0.05f + 0.01f * (k % 50);
    sm[k] = maps[0]->searchMatch(q, vel, Rvel, Rback, 40.0f);
  }
  dump(out + "/sm.bin", sm.data(), sm.size());
  // degenerate direction (|t| <= 1e-6: search along the keyline's own gradient)
  {
    const rebvio::types::Vector3f v0 = TooN::Zeros;
    rebvio::types::Matrix3f I = TooN::Zeros;
    for (int i = 0; i < 3; ++i) I(i, i) = 1.0f;
    std::vector<int> sm0(K);
    for (int k = 0; k < K; ++k) sm0[k] = maps[0]->searchMatch((*maps[1])[(k * 29) % n1], v0, Rvel, I, 40.0f);
    dump(out + "/sm0.bin", sm0.data(), sm0.size());
  }

  // --- Core::updateInverseDepthARLU on keylines with a synthetic match ---
  std::vector<float> ekf((size_t)K * 2);
  // three motions: a generic one, one that pushes rho above RHO_MAX (k % 7 == 0) and one that pushes it below RHO_MIN (== 1)
  rebvio::types::Vector3f vs[3] = {TooN::makeVector(0.01f, -0.004f, 0.02f), TooN::makeVector(0.01f, -0.004f, -0.01f),
                                   TooN::makeVector(0.01f, -0.004f, 200.0f)};
  for (int k = 0; k < K; ++k) {
    rebvio::types::Vector3f& v = vs[k % 7 == 0 ? 1 : (k % 7 == 1 ? 2 : 0)];
    rebvio::types::KeyLine kl = (*maps[1])[(k * 13) % n1];
    kl.match_pos_img = kl.pos_img + TooN::makeVector(0.5f - 0.01f * (k % 90), -0.3f + 0.02f * (k % 40));
    kl.match_gradient = kl.gradient;
    kl.match_gradient_norm = kl.gradient_norm;
    kl.match_id = 1;
    kl.rho = (k % 7 == 0) ? 19.99f : (k % 7 == 1 ? 0.0011f : 0.002f + 0.03f * k);
    kl.sigma_rho = (k % 7 <= 1) ? 0.001f : 0.01f + 0.05f * (k % 100);
    core.updateInverseDepthARLU(kl, v);
    ekf[k * 2] = kl.rho;
    ekf[k * 2 + 1] = kl.sigma_rho;
  }
  dump(out + "/ekf.bin", ekf.data(), ekf.size());

  // --- ScaleSpace / FastGaussian ---
  cv::Mat u8(H, W, CV_8UC1, buf.data());
  cv::Mat img;
  u8.convertTo(img, CV_32FC1, 3.0);
  rebvio::ScaleSpace space(cam);
  space.build(img);
  dump(out + "/dog.bin", space.dog().ptr<float>(0), (size_t)W * H);
  dump(out + "/mag.bin", space.mag().ptr<float>(0), (size_t)W * H);
  rebvio::FastGaussian fg(cam, 2.2f);
  cv::Mat sm_img = fg.smooth(img);
  dump(out + "/smooth.bin", sm_img.ptr<float>(0), (size_t)W * H);
  const float meta[6] = {(float)fg.n_, fg.sigma_, fg.sigma_true_, (float)fg.widths_[0], (float)fg.widths_[1], (float)fg.widths_[2]};
  dump(out + "/smooth_meta.bin", meta, 6);
  dump(out + "/div0.bin", fg.divisors_[0].ptr<float>(0), (size_t)W * H);

  if (timed_section(1) != 4) return 5;
  std::printf("n0=%d n1=%d\n", n0, n1);
  return 0;
}
