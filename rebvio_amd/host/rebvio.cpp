// rebvio::Rebvio over the gfx950 backend. Same threading model and observable behaviour as the reference
// (rebvio.cpp:17-313): the caller thread converts/undistorts and queues images; worker 1 detects edges and attaches the
// pre-integrated IMU data; worker 2 runs one tracking step per frame pair and publishes odometry.
//
// Differences, stated once: (i) the frame-pair step is ONE call into the backend (rebvio_hip_track_pair: distance
// field, rotate, minimizeVel, forwardMatch, extRotVel, gyroBiasCorrection, rotate, directedMatch, regularize, depth
// EKF - rebvio.cpp:142-259); (ii) the accelerometer / scale-attitude-bias fusion (estimateBias + SABEstimator,
// rebvio.cpp:206-224) is not built yet: the branch of rebvio.cpp:225-233 is always taken, scale K = 1
// and the global attitude is integrated from the gyro/visual rotation alone; (iii) the reference's latent races
// (unlocked queue peeks, plain-bool run flag) are not inherited.
#include "rebvio/rebvio.hpp"

#include <chrono>
#include <iostream>

#include "rebvio/util/log.hpp"
#include "session.hpp"

namespace rebvio {

namespace {
struct ScopeOpener {  // runs before the detector/tracker members are constructed
  explicit ScopeOpener(int device) { backend::Session::newScope(device); }
};
}  // namespace

Rebvio::Rebvio(rebvio::RebvioConfig& config)
    : config_((ScopeOpener(config.device_id), config)), run_(true), num_frames_(0), num_detected_(0), num_images_(0),
      camera_(config.camera),
      edge_detector_(std::make_shared<rebvio::Camera>(camera_), std::make_shared<rebvio::EdgeDetectorConfig>(config.edge_detector)),
      core_(std::make_shared<rebvio::Camera>(camera_), std::make_shared<rebvio::CoreConfig>(config.core)) {
  core_.session()->setImuNoise(config_.imu_state.gyro_std_dev, config_.imu_state.gyro_bias_std_dev);
  core_.session()->ctx();  // create the device context now: fail loudly here, not in a worker thread
  data_acquisition_thread_ = std::thread(&Rebvio::dataAcquisitionProcess, this);
  state_estimation_thread_ = std::thread(&Rebvio::stateEstimationProcess, this);
}

Rebvio::~Rebvio() {
  run_ = false;
  if (data_acquisition_thread_.joinable()) data_acquisition_thread_.join();
  if (state_estimation_thread_.joinable()) state_estimation_thread_.join();
}

void Rebvio::imageCallback(rebvio::types::Image&& image) {
  std::lock_guard<std::mutex> guard(image_buffer_mutex_);
  cv::Mat img;
  image.data.convertTo(img, CV_FLOAT_PRECISION, 3.0);  // 0..765, matches max_image_value_ (edge_detector.cpp:21)
  image.data = camera_.undistort(img);
  image_buffer_.push(image);
  ++num_images_;
}

void Rebvio::imuCallback(rebvio::types::Imu&& imu) {
  std::lock_guard<std::mutex> guard(imu_buffer_mutex_);
  imu_buffer_.push(imu);
}

void Rebvio::registerEdgeImageCallback(std::function<void(cv::Mat&, rebvio::EdgeMap::SharedPtr&)> cb) {
  edge_image_callbacks_.push_back(cb);
}

void Rebvio::registerOdometryCallback(std::function<void(rebvio::types::Odometry&)> cb) { odometry_callbacks_.push_back(cb); }

void Rebvio::waitIdle() {
  while (run_) {
    const unsigned imgs = num_images_;
    if (num_detected_ >= imgs && (imgs < 2 || num_frames_ + 1 >= imgs)) return;
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
}

void Rebvio::dataAcquisitionProcess() {
  REBVIO_INFO("Starting Data Acquisition Process..");
  while (run_) {
    rebvio::types::Image img;
    bool have = false;
    {
      std::lock_guard<std::mutex> guard(image_buffer_mutex_);
      if (!image_buffer_.empty()) {
        img = image_buffer_.front();
        image_buffer_.pop();
        have = true;
      }
    }
    if (!have) {
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
      continue;
    }
    rebvio::EdgeMap::SharedPtr edge_map = edge_detector_.detect(img);
    for (auto& cb : edge_image_callbacks_) cb(img.data, edge_map);
    {
      // integrate the IMU samples up to this frame BEFORE the map becomes visible to the tracker
      std::lock_guard<std::mutex> guard(imu_buffer_mutex_);
      while (!imu_buffer_.empty() && imu_buffer_.front().ts <= img.ts_us) {
        edge_map->imu().add(imu_buffer_.front(), camera_.getRc2i());
        imu_buffer_.pop();
      }
    }
    {
      std::lock_guard<std::mutex> guard(edge_map_buffer_mutex_);
      edge_map_buffer_.push(edge_map);
    }
    ++num_detected_;
  }
}

void Rebvio::stateEstimationProcess() {
  REBVIO_INFO("Starting State Estimation Process..");
  rebvio_hip_ctx* ctx = core_.session()->ctx();
  types::Vector3f Pos = TooN::Zeros;
  types::Matrix3f R_global = TooN::Identity;
  int num_gyro_init = 0;
  types::Vector3f gyro_init = TooN::Zeros;

  while (run_) {
    rebvio::EdgeMap::SharedPtr new_edge_map, old_edge_map;
    {
      std::lock_guard<std::mutex> guard(edge_map_buffer_mutex_);
      if (edge_map_buffer_.size() >= 2) {
        old_edge_map = edge_map_buffer_.front();
        edge_map_buffer_.pop();
        new_edge_map = edge_map_buffer_.front();
      }
    }
    if (!new_edge_map) {
      std::this_thread::sleep_for(std::chrono::microseconds(200));
      continue;
    }

    // IMU state initialisation (rebvio.cpp:145-160)
    const rebvio::types::IntegratedImu& imu = new_edge_map->imu().get(camera_.getRc2i(), camera_.getTc2i());
    if (!imu_state_.initialized && num_frames_ > 0) {
      if (config_.imu_state.init_bias > 0) {
        gyro_init += imu.gyro() * imu.dt_s();
        if (++num_gyro_init > config_.imu_state.init_bias_frame_num) {
          imu_state_.Bg = gyro_init / types::Float(num_gyro_init);
          imu_state_.W_Bg = types::invert(imu_state_.RGBias * types::Float(1e2));
          imu_state_.initialized = true;
        }
      } else {
        imu_state_.initialized = true;
        imu_state_.Bg = config_.imu_state.init_bias_guess * imu.dt_s();
      }
      if (imu_state_.initialized) {
        float bg[3] = {imu_state_.Bg[0], imu_state_.Bg[1], imu_state_.Bg[2]}, wb[9];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) wb[i * 3 + j] = imu_state_.W_Bg(i, j);
        rebvio_hip_set_gyro_state(ctx, bg, wb);
      }
    }

    float Rp[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Rp[i * 3 + j] = imu.R()(i, j);
    const types::Float frame_dt = types::Float(new_edge_map->ts_us() - old_edge_map->ts_us()) / 1000000.0;
    rebvio_hip_pair_out out;
    backend::check("rebvio_hip_track_pair", rebvio_hip_track_pair(ctx, old_edge_map->handle(), new_edge_map->handle(), Rp, frame_dt, &out));
    old_edge_map->invalidateMirror();
    new_edge_map->invalidateMirror();
    {
      float bg[3], wb[9];
      rebvio_hip_get_gyro_state(ctx, bg, wb);
      imu_state_.Bg = TooN::makeVector(bg[0], bg[1], bg[2]);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) imu_state_.W_Bg(i, j) = wb[i * 3 + j];
    }
    imu_state_.Vg = TooN::makeVector(out.Vg[0], out.Vg[1], out.Vg[2]);
    imu_state_.Vgva = TooN::makeVector(out.V[0], out.V[1], out.V[2]);

    if (out.status == 1) {  // rebvio.cpp:236-241
      std::cerr << "Minimization Error occured!\n";
      run_ = false;
    } else if (out.status == 2) {  // rebvio.cpp:247-252
      std::cerr << "Insufficient number of keylines matches!\n";
      run_ = false;
    }

    // incremental pose (rebvio.cpp:263-271 without the gravity-aligned frame of the SAB filter)
    if (num_frames_ > 4u + (unsigned)config_.imu_state.init_bias_frame_num) {
      types::Matrix3f Rgva;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rgva(i, j) = out.R[i * 3 + j];
      R_global = R_global * Rgva.T();
      Pos += -(R_global * imu_state_.Vgva);
    }
    types::Odometry odometry;
    odometry.ts_us = new_edge_map->ts_us();
    odometry.orientation = TooN::SO3<types::Float>(R_global).ln();
    odometry.position = Pos;
    for (auto& cb : odometry_callbacks_) cb(odometry);
    ++num_frames_;
  }
}

}  // namespace rebvio
