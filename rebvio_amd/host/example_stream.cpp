// Drives rebvio::Rebvio exactly the way ros_rebvio does (ros_rebvio/src/ros_rebvio.cpp:15-82): register an odometry
// and an edge-image callback, push MONO8 frames (and optionally IMU samples), print "ts wx wy wz px py pz" lines in
// the format of the reference's golden odometry file.
//   rebvio_stream_example frames.u8 width height n_frames [fm cx cy keylines_ref keylines_max [imu.bin [min_matches]]]
// imu.bin: records of {int64 ts_us, float gyro[3], float acc[3]} (32 bytes); without it a still 200 Hz IMU is synthesised.
// Each line carries, after the reference's seven columns, the scale K, gravity estimate, gyro bias and match count.
// REBVIO_EXAMPLE_FRAME_DT_US: camera period in microseconds (default 50000 = the 20 Hz of the reference's regression data).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <vector>

#include "rebvio/rebvio.hpp"

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s frames.u8 width height n_frames [fm cx cy keylines_ref keylines_max]\n", argv[0]);
    return 2;
  }
  const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), N = std::atoi(argv[4]);
  std::vector<unsigned char> buf((size_t)W * H * N);
  std::ifstream f(argv[1], std::ios::binary);
  if (!f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size())) {
    std::fprintf(stderr, "cannot read %s\n", argv[1]);
    return 2;
  }
  rebvio::RebvioConfig config;
  const float fm = argc > 5 ? std::atof(argv[5]) : 458.0f * W / 640.0f;
  const float cx = argc > 6 ? std::atof(argv[6]) : W / 2.0f, cy = argc > 7 ? std::atof(argv[7]) : H / 2.0f;
  config.camera = rebvio::Camera(H, W, fm, fm, cx, cy);
  if (const char* d = std::getenv("REBVIO_EXAMPLE_DISTORTION")) {  // "k1,k2,p1,p2,k3" of the rad-tan lens model
    float k[5] = {0, 0, 0, 0, 0};
    if (std::sscanf(d, "%f,%f,%f,%f,%f", &k[0], &k[1], &k[2], &k[3], &k[4]) < 1) {
      std::fprintf(stderr, "bad REBVIO_EXAMPLE_DISTORTION\n");
      return 2;
    }
    config.camera.k1_ = k[0];
    config.camera.k2_ = k[1];
    config.camera.p1_ = k[2];
    config.camera.p2_ = k[3];
    config.camera.k3_ = k[4];
  }
  if (argc > 8) config.edge_detector.keylines_ref = std::atoi(argv[8]);
  if (argc > 9) config.edge_detector.keylines_max = std::atoi(argv[9]);
  if (W * H < 640 * 480) config.core.global_min_matches_threshold = 50;
  if (argc > 11) config.core.global_min_matches_threshold = (unsigned)std::atoi(argv[11]);
  struct ImuRec {
    long long ts;
    float gyro[3], acc[3];
    int pad;
  };
  static_assert(sizeof(ImuRec) == 40 || sizeof(ImuRec) == 32, "record layout");
  std::vector<ImuRec> imu;
  if (argc > 10 && argv[10][0] != '-') {
    std::ifstream fi(argv[10], std::ios::binary);
    char rec[32];
    while (fi.read(rec, 32)) {
      ImuRec r;
      std::memcpy(&r.ts, rec, 8);
      std::memcpy(r.gyro, rec + 8, 12);
      std::memcpy(r.acc, rec + 20, 12);
      imu.push_back(r);
    }
    if (imu.empty()) {
      std::fprintf(stderr, "cannot read %s\n", argv[10]);
      return 2;
    }
  }

  const uint64_t frame_dt_us = std::getenv("REBVIO_EXAMPLE_FRAME_DT_US") ? std::strtoull(std::getenv("REBVIO_EXAMPLE_FRAME_DT_US"), nullptr, 10) : 50000ull;
  rebvio::Rebvio rebvio(config);
  std::mutex mu;
  int n_odo = 0, n_edge = 0, last_keylines = 0;
  rebvio.registerOdometryCallback([&](rebvio::types::Odometry& o) {
    std::lock_guard<std::mutex> g(mu);
    // REBVIO_EXAMPLE_PRECISE: nine significant digits, enough to give every float back bit for bit
    static const bool precise = std::getenv("REBVIO_EXAMPLE_PRECISE") != nullptr;
    std::printf(precise ? "%llu %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %d\n"
                        : "%llu %.6f %.6f %.6f %.6f %.6f %.6f %.6f %.5f %.5f %.5f %.7f %.7f %.7f %d\n", (unsigned long long)o.ts_us,
                o.orientation[0], o.orientation[1], o.orientation[2], o.position[0], o.position[1], o.position[2], o.scale,
                o.gravity[0], o.gravity[1], o.gravity[2], o.gyro_bias[0], o.gyro_bias[1], o.gyro_bias[2], o.klm_num);
    ++n_odo;
  });
  rebvio.registerEdgeImageCallback([&](cv::Mat& img, rebvio::EdgeMap::SharedPtr& map) {
    std::lock_guard<std::mutex> g(mu);
    ++n_edge;
    last_keylines = map->size();  // as ros_rebvio.cpp:44 does: size() and (*map)[i].pos[...]
    if (last_keylines > 0) (void)(*map)[0].pos[0];
    (void)img;
  });
  size_t k_imu = 0;
  for (int i = 0; i < N; ++i) {
    cv::Mat frame(H, W, CV_8UC1, buf.data() + (size_t)i * W * H);
    // samples up to this frame's stamp are queued before the frame, as a time-ordered bag replay delivers them
    for (; k_imu < imu.size() && (uint64_t)imu[k_imu].ts <= (uint64_t)i * frame_dt_us; ++k_imu)
      rebvio.imuCallback(rebvio::types::Imu{(uint64_t)imu[k_imu].ts,
                                            TooN::makeVector(imu[k_imu].gyro[0], imu[k_imu].gyro[1], imu[k_imu].gyro[2]),
                                            TooN::makeVector(imu[k_imu].acc[0], imu[k_imu].acc[1], imu[k_imu].acc[2])});
    rebvio.imageCallback(rebvio::types::Image{(uint64_t)i * frame_dt_us, frame.clone()});
    for (int k = 0; imu.empty() && k < (int)(frame_dt_us / 5000ull); ++k)  // 200 Hz IMU next to the camera: a still gyro, gravity on y
      rebvio.imuCallback(rebvio::types::Imu{(uint64_t)i * frame_dt_us + (uint64_t)k * 5000ull + 1ull, TooN::makeVector(0.0f, 0.0f, 0.0f),
                                            TooN::makeVector(0.0f, 9.81f, 0.0f)});
  }
  rebvio.waitIdle();
  std::fprintf(stderr, "frames=%d edge_callbacks=%d odometry_callbacks=%d last_keylines=%d running=%d\n", N, n_edge, n_odo,
               last_keylines, (int)rebvio.running());
  return (n_odo == N - 1 && n_edge == N) ? 0 : 1;
}
