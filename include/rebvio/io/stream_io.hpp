// Stream sources and sinks around rebvio::Rebvio without ROS: what ros_rebvio's bag player and the
// reference's TESTING odometry logger do (ros_rebvio.cpp:89-126; log.cpp:26-41, rebvio.cpp:279-286), for data on disk.
//   EurocReader      - EuRoC / ASL dataset folder (mav0/cam0/data.csv + data/*.png, mav0/imu0/data.csv)
//   RawReader        - dense MONO8 frames (rows*cols bytes each) + optional IMU records {int64 ts_us, float gyro[3], acc[3]}
//   OdometryWriter   - "ts_us wx wy wz px py pz" lines, %.6f, the format of the reference's regression file
//                      ros_rebvio/test/data/MH_03_medium_test_15s-30s_odometry.txt
//   replay()         - time-ordered playback into imageCallback / imuCallback
#pragma once

#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "rebvio/types/image.hpp"
#include "rebvio/types/imu.hpp"
#include "rebvio/types/odometry.hpp"

namespace rebvio {
namespace io {

// 8-bit grey image from a PNG file (grey 8/16 bit, RGB/RGBA 8 bit -> luma; non-interlaced). Throws std::runtime_error.
cv::Mat readPngGray(const std::string& path);

struct FrameRef {
  uint64_t ts_us;
  std::string path;    // EuRoC: image file; raw: empty
  size_t raw_index;    // raw: frame number
};

class StreamSource {
 public:
  virtual ~StreamSource() {}
  virtual size_t numFrames() const = 0;
  virtual uint64_t frameTs(size_t i) const = 0;
  virtual cv::Mat frame(size_t i) = 0;  // CV_8UC1
  const std::vector<rebvio::types::Imu>& imu() const { return imu_; }

 protected:
  std::vector<rebvio::types::Imu> imu_;  // time ordered
};

class EurocReader : public StreamSource {
 public:
  // `mav0_dir` = the folder holding cam0/ and imu0/; timestamps in the csv files are nanoseconds
  explicit EurocReader(const std::string& mav0_dir, const std::string& cam = "cam0", const std::string& imu = "imu0");
  size_t numFrames() const override { return frames_.size(); }
  uint64_t frameTs(size_t i) const override { return frames_[i].ts_us; }
  cv::Mat frame(size_t i) override { return readPngGray(frames_[i].path); }

 private:
  std::vector<FrameRef> frames_;
};

class RawReader : public StreamSource {
 public:
  RawReader(const std::string& frames_file, int rows, int cols, uint64_t first_ts_us, uint64_t frame_dt_us,
            const std::string& imu_file = "");
  size_t numFrames() const override { return n_; }
  uint64_t frameTs(size_t i) const override { return t0_ + (uint64_t)i * dt_; }
  cv::Mat frame(size_t i) override;

 private:
  std::string path_;
  int rows_, cols_;
  size_t n_;
  uint64_t t0_, dt_;
};

class OdometryWriter {
 public:
  explicit OdometryWriter(const std::string& path);
  ~OdometryWriter();
  void write(const rebvio::types::Odometry& o);
  static std::string format(const rebvio::types::Odometry& o);  // one line without the newline

 private:
  std::FILE* f_;
};

// Plays frames [first, first+count) in time order: every IMU sample with ts <= the frame's stamp is delivered before the
// frame (a time-ordered bag delivers them that way, ros_rebvio.cpp:108-121). Returns the number of frames delivered.
size_t replay(StreamSource& src, const std::function<void(rebvio::types::Image&&)>& image_cb,
              const std::function<void(rebvio::types::Imu&&)>& imu_cb, size_t first = 0, size_t count = (size_t)-1);

}  // namespace io
}  // namespace rebvio
