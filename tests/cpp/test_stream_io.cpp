// Host stream IO (no GPU): PNG decoding, EuRoC folder parsing, raw reader, replay ordering, odometry text format.
//   test_stream_io <mav0 dir> <raw frames file> <W> <H> <N> <golden odometry file>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "rebvio/io/stream_io.hpp"

static int fails = 0;
#define CHECK(c)                                              \
  do {                                                        \
    if (!(c)) {                                               \
      std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
      ++fails;                                                \
    }                                                         \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), N = std::atoi(argv[5]);
  rebvio::io::EurocReader asl(argv[1]);
  rebvio::io::RawReader raw(argv[2], H, W, 1000000ull, 50000ull);
  CHECK((int)asl.numFrames() == N && (int)raw.numFrames() == N);
  for (int i = 0; i < N; ++i) {
    cv::Mat a = asl.frame(i), b = raw.frame(i);
    CHECK(a.rows == H && a.cols == W && a.type() == CV_8UC1);
    CHECK(asl.frameTs(i) == raw.frameTs(i));
    bool same = true;
    for (int r = 0; r < H && same; ++r) same = std::memcmp(a.ptr<unsigned char>(r), b.ptr<unsigned char>(r), W) == 0;
    CHECK(same);
  }
  // the csv rows were written shuffled: both streams come back time ordered
  for (size_t i = 1; i < asl.imu().size(); ++i) CHECK(asl.imu()[i - 1].ts <= asl.imu()[i].ts);
  CHECK(asl.imu().size() == (size_t)(N - 1) * 10);
  CHECK(std::fabs(asl.imu()[0].gyro[1] - 0.25f) < 1e-6f && std::fabs(asl.imu()[0].acc[2] + 9.5f) < 1e-6f);
  // replay: every sample stamped <= a frame arrives before that frame, none is lost or duplicated
  std::vector<std::pair<int, uint64_t>> order;  // (0 = imu, 1 = image)
  const size_t played = rebvio::io::replay(
      asl, [&](rebvio::types::Image&& im) { order.push_back({1, im.ts_us}); },
      [&](rebvio::types::Imu&& s) { order.push_back({0, s.ts}); });
  CHECK((int)played == N);
  uint64_t last_img = 0;
  size_t n_imu = 0;
  for (size_t i = 0; i < order.size(); ++i) {
    if (order[i].first == 1) {
      last_img = order[i].second;
    } else {
      ++n_imu;
      CHECK(order[i].second > last_img || last_img == 0);
      size_t j = i;
      while (j < order.size() && order[j].first == 0) ++j;
      CHECK(j < order.size() && order[i].second <= order[j].second);
    }
  }
  CHECK(n_imu == asl.imu().size());
  // odometry text: parsing the reference's regression file and writing it back reproduces it byte for byte
  std::ifstream g(argv[6]);
  std::string line;
  int n_lines = 0, first_nonzero = -1;
  while (std::getline(g, line)) {
    std::istringstream iss(line);
    rebvio::types::Odometry o;
    unsigned long long ts;
    double v[6];
    iss >> ts >> v[0] >> v[1] >> v[2] >> v[3] >> v[4] >> v[5];
    o.ts_us = ts;
    for (int k = 0; k < 3; ++k) {
      o.orientation[k] = (float)v[k];
      o.position[k] = (float)v[3 + k];
    }
    CHECK(rebvio::io::OdometryWriter::format(o) == line);
    if (first_nonzero < 0 && (v[0] != 0 || v[3] != 0)) first_nonzero = n_lines;
    ++n_lines;
  }
  CHECK(n_lines == 299);
  CHECK(first_nonzero == 15);  // pose integration starts with the 16th pair: num_frames > 4 + init_bias_frame_num (rebvio.cpp:263)
  // optional: extra PNG flavours, each followed by the raw bytes it must decode to
  for (int a = 7; a + 1 < argc; a += 2) {
    cv::Mat m = rebvio::io::readPngGray(argv[a]);
    std::ifstream ef(argv[a + 1], std::ios::binary);
    std::vector<char> want((size_t)m.rows * m.cols);
    CHECK((bool)ef.read(want.data(), (std::streamsize)want.size()));
    bool same = true;
    for (int r = 0; r < m.rows && same; ++r) same = std::memcmp(m.ptr<unsigned char>(r), want.data() + (size_t)r * m.cols, m.cols) == 0;
    CHECK(same);
  }
  bool threw = false;
  try {
    rebvio::io::readPngGray(argv[2]);  // raw bytes are not a PNG
  } catch (const std::exception&) {
    threw = true;
  }
  CHECK(threw);
  if (!fails) std::printf("ok\n");
  return fails ? 1 : 0;
}
