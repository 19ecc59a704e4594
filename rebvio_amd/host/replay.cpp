// rebvio_replay: run rebvio::Rebvio over a dataset on disk and write the odometry in the reference's regression format.
//   rebvio_replay --asl <mav0 dir> --euroc --out odometry.txt [--first N --count M]
//   rebvio_replay --raw frames.u8 --size W H [--imu imu.bin] [--dt 50000] [--camera fm cx cy [k1 k2 p1 p2 k3]] --out odometry.txt
// --euroc selects the reference's built-in EuRoC MH cam0 model (camera.hpp:25-45). With the real MH_03 data
// (first = the frame at 15 s) this replays what ros_rebvio/test/test_ros_rebvio.cpp checks against its golden file.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>

#include "rebvio/io/stream_io.hpp"
#include "rebvio/rebvio.hpp"

int main(int argc, char** argv) {
  std::string asl, raw, imu, out;
  int W = 0, H = 0;
  size_t first = 0, count = (size_t)-1;
  uint64_t dt = 50000;
  bool euroc = false;
  float cam[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int ncam = 0, kref = 0, kmax = 0, min_matches = -1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) {
        std::fprintf(stderr, "missing value after %s\n", a.c_str());
        std::exit(2);
      }
      return argv[++i];
    };
    if (a == "--asl") asl = next();
    else if (a == "--raw") raw = next();
    else if (a == "--imu") imu = next();
    else if (a == "--out") out = next();
    else if (a == "--size") { W = std::atoi(next()); H = std::atoi(next()); }
    else if (a == "--first") first = (size_t)std::atoll(next());
    else if (a == "--count") count = (size_t)std::atoll(next());
    else if (a == "--dt") dt = (uint64_t)std::atoll(next());
    else if (a == "--euroc") euroc = true;
    else if (a == "--keylines") { kref = std::atoi(next()); kmax = std::atoi(next()); }
    else if (a == "--min-matches") min_matches = std::atoi(next());
    else if (a == "--camera") {
      while (ncam < 8 && i + 1 < argc && (std::isdigit((unsigned char)argv[i + 1][0]) || argv[i + 1][0] == '-' || argv[i + 1][0] == '.') &&
             std::strncmp(argv[i + 1], "--", 2) != 0)
        cam[ncam++] = (float)std::atof(argv[++i]);
    } else {
      std::fprintf(stderr, "unknown option %s\n", a.c_str());
      return 2;
    }
  }
  if ((asl.empty() == raw.empty()) || out.empty()) {
    std::fprintf(stderr, "usage: %s (--asl mav0 | --raw frames.u8 --size W H [--imu imu.bin]) [--euroc | --camera fm cx cy [k1 k2 p1 p2 k3]] --out file\n", argv[0]);
    return 2;
  }
  try {
    std::unique_ptr<rebvio::io::StreamSource> src;
    if (!asl.empty()) src.reset(new rebvio::io::EurocReader(asl));
    else src.reset(new rebvio::io::RawReader(raw, H, W, 0, dt, imu));
    if (src->numFrames() == 0) {
      std::fprintf(stderr, "no frames\n");
      return 1;
    }
    if (W == 0) {
      const cv::Mat f0 = src->frame(0);
      W = f0.cols;
      H = f0.rows;
    }
    rebvio::RebvioConfig config;
    if (!euroc) {
      const float fm = ncam > 0 ? cam[0] : 458.0f * W / 640.0f;
      config.camera = rebvio::Camera(H, W, fm, fm, ncam > 1 ? cam[1] : W / 2.0f, ncam > 2 ? cam[2] : H / 2.0f);
      if (ncam > 3) config.camera.k1_ = cam[3];
      if (ncam > 4) config.camera.k2_ = cam[4];
      if (ncam > 5) config.camera.p1_ = cam[5];
      if (ncam > 6) config.camera.p2_ = cam[6];
      if (ncam > 7) config.camera.k3_ = cam[7];
    }
    if (kref > 0) {
      config.edge_detector.keylines_ref = kref;
      config.edge_detector.keylines_max = kmax;
    }
    if (min_matches >= 0) config.core.global_min_matches_threshold = (unsigned)min_matches;
    rebvio::io::OdometryWriter writer(out);
    rebvio::Rebvio rebvio(config);
    std::mutex mu;
    size_t n_odo = 0;
    std::chrono::steady_clock::time_point t_first, t_last;
    rebvio.registerOdometryCallback([&](rebvio::types::Odometry& o) {
      std::lock_guard<std::mutex> g(mu);
      writer.write(o);
      t_last = std::chrono::steady_clock::now();
      if (n_odo == 0) t_first = t_last;
      ++n_odo;
    });
    const size_t n = rebvio::io::replay(
        *src, [&](rebvio::types::Image&& im) { rebvio.imageCallback(std::move(im)); },
        [&](rebvio::types::Imu&& s) { rebvio.imuCallback(std::move(s)); }, first, count);
    rebvio.waitIdle();
    std::fprintf(stderr, "frames=%zu odometry=%zu running=%d\n", n, n_odo, (int)rebvio.running());
    if (n_odo > 1) {  // wall clock between the first and the last published odometry: the whole class, input thread included
      const double sec = std::chrono::duration<double>(t_last - t_first).count();
      std::fprintf(stderr, "[replay] %zu odometry records in %.3f s = %.0f frames/s (wall clock, first to last record)\n", n_odo, sec,
                   (double)(n_odo - 1) / sec);
    }
    return (n_odo + 1 == n || n == 0) ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
