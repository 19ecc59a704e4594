// An EdgeMap::SharedPtr handed to an edge-image callback (rebvio.hpp:104-109; ros_rebvio.cpp:32-51 keeps such pointers for
// as long as its publisher wants) must stay usable after the pipeline that produced it is gone, and releasing it then must
// not touch freed memory. Also the C-ABI form of the same: create -> detect -> destroy -> queries / release on the handle.
//   test_map_lifetime frames.u8 width height n_frames fm cx cy
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "rebvio/rebvio.hpp"
#include "rebvio_hip.h"

int main(int argc, char** argv) {
  if (argc < 8) return 2;
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]);
  std::vector<unsigned char> buf((size_t)W * H * N);
  std::ifstream f(argv[1], std::ios::binary);
  if (!f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size())) return 2;
  const float fm = std::atof(argv[5]), cx = std::atof(argv[6]), cy = std::atof(argv[7]);

  std::vector<rebvio::EdgeMap::SharedPtr> kept;
  {
    rebvio::RebvioConfig config;
    config.camera = rebvio::Camera(H, W, fm, fm, cx, cy);
    config.edge_detector.keylines_ref = 2500;
    config.edge_detector.keylines_max = 3500;
    config.core.global_min_matches_threshold = 50;
    rebvio::Rebvio rebvio(config);
    std::mutex mu;
    rebvio.registerEdgeImageCallback([&](cv::Mat&, rebvio::EdgeMap::SharedPtr& map) {
      std::lock_guard<std::mutex> g(mu);
      kept.push_back(map);  // the consumer holds on to every map
    });
    for (int i = 0; i < N; ++i) {
      cv::Mat frame(H, W, CV_8UC1, buf.data() + (size_t)i * W * H);
      rebvio.imageCallback(rebvio::types::Image{(uint64_t)i * 50000ull, frame.clone()});
      for (int k = 0; k < 10; ++k)
        rebvio.imuCallback(rebvio::types::Imu{(uint64_t)i * 50000ull + (uint64_t)k * 5000ull + 1ull, TooN::makeVector(0.0f, 0.0f, 0.0f),
                                              TooN::makeVector(0.0f, 9.81f, 0.0f)});
    }
    rebvio.waitIdle();
  }  // ~Rebvio: workers joined, detector and tracker gone
  if ((int)kept.size() != N) {
    std::fprintf(stderr, "kept %zu maps of %d\n", kept.size(), N);
    return 1;
  }
  int total = 0;
  for (auto& m : kept) {  // still readable: the maps keep their backend context alive
    const int n = m->size();
    if (n <= 0) return 1;
    total += n;
    (void)(*m)[n - 1].pos[0];
  }
  kept.clear();  // releases every handle, then the context with the last one
  std::fprintf(stderr, "kept maps held %d keylines\n", total);

  // C-ABI: a handle that outlives its context is inert - queries fail with -10, release frees the husk
  rebvio_hip_params p;
  rebvio_hip_default_params(&p, H, W);
  p.fm = fm; p.cx = cx; p.cy = cy;
  p.keylines_ref = 2500; p.keylines_max = 3500;
  rebvio_hip_ctx* ctx = nullptr;
  if (rebvio_hip_create(&p, &ctx) != 0) return 1;
  {
    // Maps acquired on one thread (detect) while another thread releases earlier ones - what an edge-image consumer that lets
    // go of its EdgeMap::SharedPtr does to the acquisition thread's pool (ros_rebvio.cpp:32-51). The pool's bookkeeping is
    // under one mutex per context; the keyline counts must be those of the same frames detected and released on one thread.
    const int rounds = 6 * N;
    std::vector<int> serial((size_t)rounds), threaded((size_t)rounds, -1);
    for (int i = 0; i < rounds; ++i) {
      rebvio_hip_map* m = nullptr;
      if (rebvio_hip_detect_u8(ctx, buf.data() + (size_t)(i % N) * W * H, 0, (uint64_t)i * 50000ull, &m) != 0) return 1;
      serial[(size_t)i] = rebvio_hip_map_size(m);
      rebvio_hip_map_release(m);
    }
    rebvio_hip_ctx* ctx2 = nullptr;
    if (rebvio_hip_create(&p, &ctx2) != 0) return 1;
    std::mutex qmu;
    std::condition_variable qcv;
    std::deque<std::pair<int, rebvio_hip_map*>> q;
    bool done = false;
    std::thread consumer([&] {
      for (;;) {
        std::pair<int, rebvio_hip_map*> e;
        {
          std::unique_lock<std::mutex> lk(qmu);
          qcv.wait(lk, [&] { return done || !q.empty(); });
          if (q.empty()) return;
          e = q.front();
          q.pop_front();
        }
        threaded[(size_t)e.first] = rebvio_hip_map_size(e.second);
        rebvio_hip_map_release(e.second);
      }
    });
    int rc_det = 0;
    for (int i = 0; i < rounds && rc_det == 0; ++i) {
      rebvio_hip_map* m = nullptr;
      rc_det = rebvio_hip_detect_u8(ctx2, buf.data() + (size_t)(i % N) * W * H, 0, (uint64_t)i * 50000ull, &m);
      if (rc_det == 0) {
        std::lock_guard<std::mutex> lk(qmu);
        q.emplace_back(i, m);
      }
      qcv.notify_one();
    }
    {
      std::lock_guard<std::mutex> lk(qmu);
      done = true;
    }
    qcv.notify_one();
    consumer.join();
    rebvio_hip_destroy(ctx2);
    if (rc_det != 0 || serial != threaded) {
      std::fprintf(stderr, "maps released on another thread: detect rc %d, counts %s\n", rc_det, serial == threaded ? "equal" : "differ");
      return 1;
    }
  }
  rebvio_hip_map *m0 = nullptr, *m1 = nullptr;
  if (rebvio_hip_detect_u8(ctx, buf.data(), 0, 0, &m0) != 0) return 1;
  if (rebvio_hip_detect_u8(ctx, buf.data() + (size_t)W * H, 0, 50000, &m1) != 0) return 1;
  if (rebvio_hip_map_size(m0) <= 0) return 1;
  rebvio_hip_map_release(m1);  // released before: nothing left of it afterwards
  rebvio_hip_destroy(ctx);
  std::vector<rebvio_hip_keyline> kl(4000);
  if (rebvio_hip_map_size(m0) != -10) return 1;
  if (!std::isnan(rebvio_hip_map_threshold(m0))) return 1;
  if (rebvio_hip_map_download(m0, kl.data(), nullptr) != -10) return 1;
  if (rebvio_hip_map_ts(m0) != 0) return 1;
  rebvio_hip_map_release(m0);
  std::printf("ok\n");
  return 0;
}
