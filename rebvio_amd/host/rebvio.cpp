// rebvio::Rebvio over the gfx950 backend. Same threading model and observable behaviour as the reference
// (rebvio.cpp:17-313): the caller thread converts/undistorts and queues images; worker 1 detects edges and attaches the
// pre-integrated IMU data; worker 2 runs one tracking step per frame pair and publishes odometry.
//
// The frame-pair step is calls into the backend with the O(1) inertial fusion between them (and the previous pair's
// counters fetched while this pair's first half is already queued behind that pair's second half, so that the GPU goes from
// one pair to the next without waiting for the host):
//   rebvio_hip_track_pair_begin  - distance field, rotate by the gyro prior, minimizeVel, forwardMatch, extRotVel,
//                                  gyroBiasCorrection (rebvio.cpp:142-192)
//   host                         - Ls4 / mean acceleration, Core::estimateBias + SABEstimator, second rotation
//                                  (rebvio.cpp:195-233)
//   rebvio_hip_track_pair_finish_async / _result - rotate, directedMatch, regularize, depth EKF (rebvio.cpp:222-259)
// followed by the gravity-aligned pose integration (rebvio.cpp:263-271). The reference's latent races (unlocked queue
// peeks, plain-bool run flag) are not inherited.
#include "rebvio/rebvio.hpp"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <limits>

#include "../csrc/hostmath.hpp"
#include "rebvio/util/log.hpp"
#include "session.hpp"

namespace rebvio {

namespace {
struct ScopeOpener {  // runs before the detector/tracker members are constructed
  explicit ScopeOpener(int device) { backend::Session::newScope(device); }
};

// The reference library throws nothing from its workers: a failed step prints to stderr and clears run_
// (rebvio.cpp:236-252). A backend error surfacing on a worker thread (edge-map pool exhausted, a HIP failure) does the
// same here instead of ending the process through std::terminate.
template <class Body>
void run_worker(const char* name, std::atomic<bool>& run, Body&& body) {
  try {
    body();
  } catch (const std::exception& e) {
    std::cerr << name << " stopped: " << e.what() << "\n";
    run = false;
  } catch (...) {
    std::cerr << name << " stopped: unknown error\n";
    run = false;
  }
}
}  // namespace

Rebvio::Rebvio(rebvio::RebvioConfig& config)
    : config_((ScopeOpener(config.device_id), config)), run_(true), num_frames_(0), num_detected_(0), num_images_(0),
      camera_(config.camera),
      edge_detector_(std::make_shared<rebvio::Camera>(camera_), std::make_shared<rebvio::EdgeDetectorConfig>(config.edge_detector)),
      core_(std::make_shared<rebvio::Camera>(camera_), std::make_shared<rebvio::CoreConfig>(config.core)),
      sab_state_(config_.imu_state) {
  core_.session()->setImuNoise(config_.imu_state.gyro_std_dev, config_.imu_state.gyro_bias_std_dev);
  core_.session()->ctx();  // create the device context now: fail loudly here, not in a worker thread
  data_acquisition_thread_ = std::thread([this] { run_worker("Data Acquisition Process", run_, [this] { dataAcquisitionProcess(); }); });
  state_estimation_thread_ = std::thread([this] { run_worker("State Estimation Process", run_, [this] { stateEstimationProcess(); }); });
}

Rebvio::~Rebvio() {
  run_ = false;
  if (data_acquisition_thread_.joinable()) data_acquisition_thread_.join();
  if (state_estimation_thread_.joinable()) state_estimation_thread_.join();
}

void Rebvio::imageCallback(rebvio::types::Image&& image) {
  std::lock_guard<std::mutex> guard(image_buffer_mutex_);
  if (image.data.type() != CV_8UC1) {
    // not a MONO8 frame: convert and undistort here like the reference does (rebvio.cpp:43-47)
    cv::Mat img;
    image.data.convertTo(img, CV_FLOAT_PRECISION, 3.0);  // 0..765, matches max_image_value_ (edge_detector.cpp:21)
    image.data = camera_.undistort(img);
  }  // else: the u8 frame goes to the device as it is; x3 + undistort run there (1 byte/pixel over PCIe instead of 4)
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
    if (num_detected_ >= imgs && (imgs < 2 || num_published_ + 1 >= imgs)) return;
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
    // back-pressure: the reference's edge-map queue is unbounded (rebvio.cpp:86-90); here every queued map holds device
    // memory from a bounded pool, so detection waits while the tracker is more than a few maps behind
    for (;;) {
      size_t queued;
      {
        std::lock_guard<std::mutex> guard(edge_map_buffer_mutex_);
        queued = edge_map_buffer_.size();
      }
      if (queued < 8 || !run_) break;
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    static const bool acq_timers = std::getenv("REBVIO_HOST_TIMERS") != nullptr;
    const auto td0 = std::chrono::steady_clock::now();
    rebvio::EdgeMap::SharedPtr edge_map = edge_detector_.detect(img);
    if (acq_timers) {
      static double acc_us = 0;
      static unsigned acc_n = 0;
      acc_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - td0).count();
      if (++acc_n % 2000 == 0) std::fprintf(stderr, "[Rebvio] EdgeDetector::detect (staging + enqueue) %.1f us per frame on the acquisition thread\n", acc_us / acc_n);
    }
    if (!edge_image_callbacks_.empty()) {
      // callbacks see the undistorted frame, as in the reference; for a raw u8 frame of a distorting lens it is fetched
      // from the device front end (only when somebody listens)
      if (img.data.type() == CV_8UC1 && core_.session()->hasDistortion()) {
        cv::Mat und(img.data.rows, img.data.cols, CV_32FC1);
        cv::Mat dense = (img.data.step == (size_t)img.data.cols) ? img.data : img.data.clone();
        backend::check("rebvio_hip_front_end_u8",
                       rebvio_hip_front_end_u8(core_.session()->ctx(), dense.ptr<unsigned char>(0), und.ptr<float>(0)));
        img.data = und;
      }
      for (auto& cb : edge_image_callbacks_) cb(img.data, edge_map);
    }
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
      edge_map_buffer_.push_back(edge_map);
    }
    ++num_detected_;
  }
}

namespace {
// stand-in for the reference's REBVIO_TIMER accumulators (util/timer.hpp): per-stage host times of the tracking worker,
// printed when the worker ends if REBVIO_HOST_TIMERS is set
struct StageTimers {
  bool on = std::getenv("REBVIO_HOST_TIMERS") != nullptr;
  double t[4] = {0, 0, 0, 0};
  double t_bias = 0;  // Core::estimateBias alone (inside t[1])
  double t_in = 0;  // from the end of a pair to the next pair's first device call (input queue, IMU pre-integration read)
  std::chrono::steady_clock::time_point pair_end;
  bool have_end = false;
  unsigned n = 0;
  std::chrono::steady_clock::time_point last;
  void start() {
    if (!on) return;
    last = std::chrono::steady_clock::now();
    if (have_end) t_in += std::chrono::duration<double, std::micro>(last - pair_end).count();
  }
  void end_pair() {
    if (!on) return;
    pair_end = std::chrono::steady_clock::now();
    have_end = true;
  }
  void lap(int i) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    t[i] += std::chrono::duration<double, std::micro>(now - last).count();
    last = now;
  }
  ~StageTimers() {
    if (on && n)
      std::fprintf(stderr, "[Rebvio] per pair (us): first half on device %.1f  acceleration + bias/scale filter %.1f  second half on device %.1f  pose + callbacks %.1f  (between pairs: input queue + IMU read %.1f)\n",
                   t[0] / n, t[1] / n, t[2] / n, t[3] / n, t_in / n);
    const backend::FusionCounters& f = backend::t_fusion;
    if (on && n && f.gn_calls)
      std::fprintf(stderr, "[Rebvio]   estimateBias %.1f us per pair: %.2f Gauss-Newton iterations per call, %lu of %lu solves through the pseudo-inverse\n",
                   t_bias / n, (double)f.gn_iterations / f.gn_calls, f.pinv_solves, f.gn_iterations);
  }
};
// REBVIO_DUMP_FUSION=<file>: every Core::estimateBias call of the run as 237 raw floats - 100 inputs (sacc, facc, kP, Rot, Qg,
// Qrot, Qbias, QKp, Rg, g_norm, Rs, Rf, Wvw), the 68 state words (X, P, g_est, b_est, Xvw) before the call, the scale it
// returned and the state words after it. tests/golden/estimate_bias_calls.npz was recorded this way
// (tools/record_fusion_calls.py); tests/test_fusion_math.py replays it on the CPU.
struct FusionDump {
  FILE* f = std::getenv("REBVIO_DUMP_FUSION") ? std::fopen(std::getenv("REBVIO_DUMP_FUSION"), "wb") : nullptr;
  void put(float v) { std::fwrite(&v, 4, 1, f); }
  void put3(const types::Matrix3f& m) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) put(m(i, j));
  }
  void state(const SABEstimator::State& s, const types::Vector6f& Xvw) {
    for (int i = 0; i < 7; ++i) put(s.X[i]);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) put(s.P(i, j));
    for (int i = 0; i < 3; ++i) put(s.g_est[i]);
    for (int i = 0; i < 3; ++i) put(s.b_est[i]);
    for (int i = 0; i < 6; ++i) put(Xvw[i]);
  }
  void before(const types::Vector3f& sacc, const types::Vector3f& facc, float kP, const types::Matrix3f& Rot,
              const SABEstimator::State& s, const types::Matrix6f& Wvw, const types::Vector6f& Xvw, float g_norm) {
    if (!f) return;
    for (int i = 0; i < 3; ++i) put(sacc[i]);
    for (int i = 0; i < 3; ++i) put(facc[i]);
    put(kP);
    put3(Rot);
    put3(s.Qg);
    put3(s.Qrot);
    put3(s.Qbias);
    put(s.QKp);
    put(s.Rg);
    put(g_norm);
    put3(s.Rs);
    put3(s.Rv);
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) put(Wvw(i, j));
    state(s, Xvw);
  }
  void after(float k, const SABEstimator::State& s, const types::Vector6f& Xvw) {
    if (!f) return;
    put(k);
    state(s, Xvw);
  }
  ~FusionDump() {
    if (f) std::fclose(f);
  }
};
}  // namespace

void Rebvio::stateEstimationProcess() {
  REBVIO_INFO("Starting State Estimation Process..");
  StageTimers timers;
  FusionDump fusion_dump;
  rebvio_hip_ctx* ctx = core_.session()->ctx();
  const types::Float FMAX = std::numeric_limits<types::Float>::max();
  types::Vector3f Pos = TooN::Zeros;
  types::Matrix3f R_global = TooN::Identity;
  types::Float K = 1.0;
  types::Float P_Kp = 5e-6;
  int num_gyro_init = 0;
  types::Vector3f gyro_init = TooN::Zeros;
  types::Vector3f g_init = TooN::Zeros;

  auto load3 = [](const float* p) {
    types::Matrix3f m;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) m(i, j) = p[i * 3 + j];
    return m;
  };
  auto store3 = [](const types::Matrix3f& m, float* p) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) p[i * 3 + j] = m(i, j);
  };

  // The odometry of a pair is complete once its second half has reported its match count (rebvio.cpp:245-252). That is
  // fetched AFTER the next pair's first half has been queued, so the record waits here for one loop turn (or until the
  // input runs dry). Returns false when the pair ended the run (minimization error / too few matches).
  struct Pending {
    bool have = false;     // a pair whose counters have not been fetched yet
    bool fetched = false;  // counters in, record not yet handed to the callbacks
    types::Odometry odometry;
    rebvio::EdgeMap::SharedPtr old_map, new_map;
  } pending;
  // Two steps: the counters (and with them the stop conditions of rebvio.cpp:236-252) are fetched as soon as the next pair's
  // first half is back; the callbacks run after that pair's second half has been launched, off the path between its halves.
  auto publish_pending = [&]() {
    if (!pending.fetched) return;
    pending.fetched = false;
    for (auto& cb : odometry_callbacks_) cb(pending.odometry);
    pending.old_map.reset();
    pending.new_map.reset();
    ++num_published_;
  };
  auto fetch_pending = [&]() -> bool {
    if (!pending.have) return true;
    pending.have = false;
    pending.fetched = true;
    int klm_num = 0, kf_matches = 0, reg_num = 0, status = 0;
    backend::check("rebvio_hip_track_pair_result", rebvio_hip_track_pair_result(ctx, &klm_num, &kf_matches, &reg_num, &status));
    pending.old_map->invalidateMirror();
    pending.new_map->invalidateMirror();
    bool ok = true;
    if (status == 1) {  // rebvio.cpp:236-241
      P_Kp = FMAX;
      std::cerr << "Minimization Error occured!\n";
      run_ = false;
      ok = false;
    } else if (status == 2) {  // rebvio.cpp:247-252
      P_Kp = FMAX;
      std::cerr << "Insufficient number of keylines matches!\n";
      run_ = false;
      ok = false;
    }
    pending.odometry.klm_num = klm_num;
    return ok;
  };
  auto complete_pending = [&]() -> bool {
    const bool ok = fetch_pending();
    publish_pending();
    return ok;
  };

  while (run_) {
    rebvio::EdgeMap::SharedPtr new_edge_map, old_edge_map;
    {
      std::lock_guard<std::mutex> guard(edge_map_buffer_mutex_);
      if (edge_map_buffer_.size() >= 2) {
        old_edge_map = edge_map_buffer_.front();
        edge_map_buffer_.pop_front();
        new_edge_map = edge_map_buffer_.front();
      }
    }
    if (!new_edge_map) {
      if (pending.have) {  // nothing to overlap with: finish the last pair now
        complete_pending();
        continue;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(200));
      continue;
    }

    types::Matrix3f P_V = TooN::Identity * FMAX, P_W = TooN::Identity * FMAX;

    // IMU state initialisation (rebvio.cpp:145-160)
    const rebvio::types::IntegratedImu& imu = new_edge_map->imu().get(camera_.getRc2i(), camera_.getTc2i());
    if (!imu_state_.initialized && num_frames_ > 0) {
      if (config_.imu_state.init_bias > 0) {
        gyro_init += imu.gyro() * imu.dt_s();
        g_init -= imu.cacc();
        if (++num_gyro_init > config_.imu_state.init_bias_frame_num) {
          imu_state_.Bg = gyro_init / types::Float(num_gyro_init);
          imu_state_.W_Bg = types::invert(imu_state_.RGBias * types::Float(1e2));
          const types::Vector3f g0 = g_init / types::Float(num_gyro_init);
          for (int i = 0; i < 3; ++i) sab_state_.X[1 + i] = g0[i];
          imu_state_.initialized = true;
        }
      } else {
        imu_state_.initialized = true;
        imu_state_.Bg = config_.imu_state.init_bias_guess * imu.dt_s();
      }
      if (imu_state_.initialized) {
        float bg[3] = {imu_state_.Bg[0], imu_state_.Bg[1], imu_state_.Bg[2]}, wb[9];
        store3(imu_state_.W_Bg, wb);
        rebvio_hip_set_gyro_state(ctx, bg, wb);
      }
    }

    // first half on the device (rebvio.cpp:142-192)
    float Rp[9];
    store3(imu.R(), Rp);
    const types::Float frame_dt = types::Float(new_edge_map->ts_us() - old_edge_map->ts_us()) / 1000000.0;
    rebvio_hip_pair_mid mid;
    timers.start();
    backend::check("rebvio_hip_track_pair_begin",
                   rebvio_hip_track_pair_begin(ctx, old_edge_map->handle(), new_edge_map->handle(), Rp, frame_dt, &mid));
    // this pair's first half (and its parked second half) are queued behind the previous pair's second half: now that pair's
    // counters can be fetched without leaving the GPU idle
    if (!fetch_pending()) {
      publish_pending();
      const float nanv[3] = {std::numeric_limits<float>::quiet_NaN(), 0.f, 0.f}, I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      int st = 0;  // the run has ended: let this pair's parked second half fall through
      (void)rebvio_hip_track_pair_finish(ctx, old_edge_map->handle(), new_edge_map->handle(), nanv, I9, I9, I9, nullptr, nullptr, nullptr, &st);
      break;
    }
    {
      float bg[3], wb[9];
      rebvio_hip_get_gyro_state(ctx, bg, wb);
      imu_state_.Bg = TooN::makeVector(bg[0], bg[1], bg[2]);
      imu_state_.W_Bg = load3(wb);
    }
    // the noise models the backend applied in gyroBiasCorrection (rebvio.cpp:186-187); RGBias feeds next frame's W_Bg init
    imu_state_.RGBias = TooN::Identity * (double)(config_.imu_state.gyro_bias_std_dev * config_.imu_state.gyro_bias_std_dev * frame_dt * frame_dt);
    imu_state_.RGyro = TooN::Identity * (double)(config_.imu_state.gyro_std_dev * config_.imu_state.gyro_std_dev * frame_dt * frame_dt);
    timers.lap(0);
    imu_state_.Vg = TooN::makeVector(mid.Vg[0], mid.Vg[1], mid.Vg[2]);
    imu_state_.P_Vg = load3(mid.P_Vg);
    types::Vector6f Xgv, Xgva;
    types::Matrix6f W_Xgv;
    for (int i = 0; i < 6; ++i) {
      Xgv[i] = mid.Xgv[i];
      for (int j = 0; j < 6; ++j) W_Xgv(i, j) = mid.W_Xgv[i * 6 + j];
    }
    imu_state_.dVgv = Xgv.slice<0, 3>();
    imu_state_.dWgv = Xgv.slice<3, 3>();

    // rebvio.cpp:195-203
    types::Matrix3f R = load3(mid.R);
    types::Matrix3f Rgva = R;
    const types::Matrix3f R0 = TooN::SO3<types::Float>(imu_state_.dWgv).get_matrix();
    R = (R0 * R.T()).T();
    imu_state_.Vgv = R0 * imu_state_.Vg + imu_state_.dVgv;
    {
      float A6[36], inv[36];
      for (int i = 0; i < 36; ++i) A6[i] = mid.W_Xgv[i];
      rh::hm::cholesky6_inverse(A6, inv);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          P_V(i, j) = inv[i * 6 + j];
          P_W(i, j) = inv[(3 + i) * 6 + 3 + j];
        }
    }

    // rebvio.cpp:206-209
    core_.estimateLs4Acceleration(-imu_state_.Vgv / frame_dt, imu_state_.Av, R, frame_dt);
    core_.estimateMeanAcceleration(imu.cacc(), imu_state_.As, R);
    Xgva = Xgv;
    sab_state_.Rv = P_V / (frame_dt * frame_dt * frame_dt * frame_dt);
    sab_state_.Qrot = P_W;
    sab_state_.QKp = P_Kp;

    types::Matrix3f R_second;
    if (num_frames_ > 4u + (unsigned)config_.imu_state.init_bias_frame_num) {  // rebvio.cpp:210-224
      const auto tb0 = timers.on ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
      fusion_dump.before(imu_state_.As, imu_state_.Av, 1.0, R, sab_state_, W_Xgv, Xgva, config_.imu_state.g_norm);
      K = core_.estimateBias(imu_state_.As, imu_state_.Av, 1.0, R, sab_state_.X, sab_state_.P, sab_state_.Qg, sab_state_.Qrot,
                             sab_state_.Qbias, sab_state_.QKp, sab_state_.Rg, sab_state_.Rs, sab_state_.Rv, sab_state_.g_est,
                             sab_state_.b_est, W_Xgv, Xgva, config_.imu_state.g_norm);
      if (timers.on) timers.t_bias += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tb0).count();
      fusion_dump.after(K, sab_state_, Xgva);
      imu_state_.dVgva = Xgva.slice<0, 3>();
      imu_state_.dWgva = Xgva.slice<3, 3>();
      const types::Matrix3f R0gva = TooN::SO3<types::Float>(imu_state_.dWgva).get_matrix();
      Rgva = (R0gva * Rgva.T()).T();
      imu_state_.Vgva = R0gva * imu_state_.Vg + imu_state_.dVgva;
      R_second = R0gva;
    } else {  // rebvio.cpp:225-233
      Rgva = R;
      imu_state_.Vgva = imu_state_.Vgv;
      R_second = R0;
    }

    // second half on the device (rebvio.cpp:222/232, 236-259)
    float V[3] = {imu_state_.Vgva[0], imu_state_.Vgva[1], imu_state_.Vgva[2]}, pv[9], rg[9], r2[9];
    store3(P_V, pv);
    store3(Rgva, rg);
    store3(R_second, r2);
    timers.lap(1);
    // If the following frame is already queued, its gyro pre-integration (complete before the map was queued) is the next
    // pair's prior: that pair's first rotateKeylines then rides in this pair's last kernel. Only once the gyro bias is
    // initialised (until then the bias still changes between pairs, rebvio.cpp:146-160).
    float Rnext[9];
    bool have_next = false;
    rebvio::EdgeMap::SharedPtr next_map;
    {
      std::lock_guard<std::mutex> guard(edge_map_buffer_mutex_);
      if (edge_map_buffer_.size() >= 2) {
        next_map = edge_map_buffer_[1];
        if (imu_state_.initialized) {
          store3(next_map->imu().R(), Rnext);
          have_next = true;
        }
      }
    }
    // the track stream's wait for the next frame's detection goes in ahead of this pair's second half
    if (next_map) backend::check("rebvio_hip_track_pair_hint_next", rebvio_hip_track_pair_hint_next(ctx, next_map->handle()));
    backend::check("rebvio_hip_track_pair_finish_async",
                   rebvio_hip_track_pair_finish_async(ctx, old_edge_map->handle(), new_edge_map->handle(), V, pv, rg, r2,
                                                      have_next ? Rnext : nullptr));
    publish_pending();  // the previous pair's record: its callbacks run while the device works on this pair's second half
    timers.lap(2);

    // gravity-aligned pose integration (rebvio.cpp:263-271)
    if (num_frames_ > 4u + (unsigned)config_.imu_state.init_bias_frame_num) {
      imu_state_.u_est = Rgva.T() * imu_state_.u_est;
      imu_state_.u_est =
          imu_state_.u_est - (imu_state_.u_est * sab_state_.g_est) / (sab_state_.g_est * sab_state_.g_est) * sab_state_.g_est;
      imu_state_.u_est = imu_state_.u_est / std::sqrt(imu_state_.u_est * imu_state_.u_est);
      const types::Matrix3f R1 = TooN::SO3<types::Float>(sab_state_.g_est, TooN::makeVector(0.0f, 1.0f, 0.0f)).get_matrix();
      const types::Matrix3f R2 = TooN::SO3<types::Float>(R1 * imu_state_.u_est, TooN::makeVector(1.0f, 0.0f, 0.0f)).get_matrix();
      R_global = R2 * R1;
      Pos += -(R_global * imu_state_.Vgva) * K;
    }
    types::Odometry odometry;
    odometry.ts_us = new_edge_map->ts_us();
    odometry.orientation = TooN::SO3<types::Float>(R_global).ln();
    odometry.position = Pos;
    odometry.scale = K;
    odometry.gravity = sab_state_.g_est;
    odometry.gyro_bias = imu_state_.Bg;
    pending.have = true;  // published by complete_pending() once the match count is in
    pending.odometry = odometry;
    pending.old_map = old_edge_map;
    pending.new_map = new_edge_map;
    timers.lap(3);
    timers.end_pair();
    timers.n++;
    ++num_frames_;
  }
  complete_pending();
}

}  // namespace rebvio
