// rebvio::Rebvio — pipeline orchestrator with the reference's public surface (rebvio.hpp:38-62): two worker threads
// (edge detection || state estimation), image / IMU callbacks in, edge-image / odometry callbacks out.
// ros_rebvio.cpp builds against this header unchanged.
#pragma once

#include <atomic>
#include <functional>
#include <mutex>
#include <deque>
#include <queue>
#include <thread>
#include <vector>

#include "rebvio/camera.hpp"
#include "rebvio/core.hpp"
#include "rebvio/edge_detector.hpp"
#include "rebvio/sab_estimator.hpp"
#include "rebvio/types/definitions.hpp"
#include "rebvio/types/image.hpp"
#include "rebvio/types/imu.hpp"
#include "rebvio/types/odometry.hpp"

namespace rebvio {

struct RebvioConfig {
  rebvio::EdgeDetectorConfig edge_detector;
  rebvio::CoreConfig core;
  rebvio::types::ImuStateConfig imu_state;
  rebvio::Camera camera;  // addition: the reference hard-codes EuRoC's camera; default-constructed = the same camera
  int device_id{0};       // addition: GPU ordinal of this camera stream
};

class Rebvio {
 public:
  Rebvio(rebvio::RebvioConfig& config);
  ~Rebvio();

  void imageCallback(rebvio::types::Image&& image);
  void imuCallback(rebvio::types::Imu&& imu);
  void registerEdgeImageCallback(std::function<void(cv::Mat&, rebvio::EdgeMap::SharedPtr&)> cb);
  void registerOdometryCallback(std::function<void(rebvio::types::Odometry&)> cb);

  // additions for embedding without ROS: block until every queued frame has been processed / current status
  void waitIdle();
  bool running() const { return run_; }
  unsigned int framesProcessed() const { return num_frames_; }

 private:
  void dataAcquisitionProcess();
  void stateEstimationProcess();

  rebvio::RebvioConfig config_;
  std::atomic<bool> run_;
  std::atomic<unsigned int> num_frames_, num_detected_, num_images_, num_published_{0};
  rebvio::Camera camera_;
  rebvio::EdgeDetector edge_detector_;
  rebvio::Core core_;
  rebvio::types::ImuState imu_state_;
  rebvio::SABEstimator::State sab_state_;  // after core_ and config_ in construction order

  std::queue<rebvio::types::Image> image_buffer_;
  std::mutex image_buffer_mutex_;
  std::queue<rebvio::types::Imu> imu_buffer_;
  std::mutex imu_buffer_mutex_;
  std::deque<rebvio::EdgeMap::SharedPtr> edge_map_buffer_;  // (a deque: the tracker looks one map ahead for the next pair's gyro prior)
  std::mutex edge_map_buffer_mutex_;
  std::vector<std::function<void(cv::Mat&, rebvio::EdgeMap::SharedPtr&)>> edge_image_callbacks_;
  std::vector<std::function<void(rebvio::types::Odometry&)>> odometry_callbacks_;
  std::thread data_acquisition_thread_, state_estimation_thread_;
};

}  // namespace rebvio
