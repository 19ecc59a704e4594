// One gfx950 backend context (rebvio_hip_ctx) shared by the EdgeDetector, the Core and the EdgeMaps of ONE camera
// stream. The reference constructs detector and tracker independently, each with its own Camera copy and config
// (rebvio.cpp:23-24); the backend needs all parameters at creation, so the context is created lazily at first use,
// after both constructors have registered their configuration.
#pragma once

#include <memory>
#include <mutex>

#include "rebvio/core.hpp"
#include "rebvio/edge_detector.hpp"
#include "rebvio_hip.h"

namespace rebvio {
namespace backend {

class Session {
 public:
  // Sessions are keyed by (scope, rows, cols, fm, cx, cy). newScope() starts a fresh camera stream on `device_id`
  // for objects constructed afterwards on this thread (Rebvio's constructor calls it).
  static int newScope(int device_id);
  static std::shared_ptr<Session> forCamera(const Camera& cam);

  void setDetectorConfig(const EdgeDetectorConfig& c);
  void setCoreConfig(const CoreConfig& c);
  void setImuNoise(float gyro_std_dev, float gyro_bias_std_dev);
  rebvio_hip_ctx* ctx();  // throws std::runtime_error when no GPU / library error: there is no CPU fallback
  const rebvio_hip_params& params() const { return p_; }
  bool hasDistortion() const { return D5_[0] != 0 || D5_[1] != 0 || D5_[2] != 0 || D5_[3] != 0 || D5_[4] != 0; }
  ~Session();

 private:
  Session() = default;
  rebvio_hip_params p_{};
  float K4_[4]{}, D5_[5]{};  // lens model of the device front end (camera.hpp:39-40)
  rebvio_hip_ctx* ctx_ = nullptr;
  std::mutex mu_;
};

// REBVIO_HOST_TIMERS diagnostics of the fusion thread (sab_estimator.cpp counts, rebvio.cpp prints)
struct FusionCounters {
  unsigned long gn_calls = 0, gn_iterations = 0, pinv_solves = 0;
};
extern thread_local FusionCounters t_fusion;

[[noreturn]] void fail(const char* what, int rc);
inline void check(const char* what, int rc) {
  if (rc != 0) fail(what, rc);
}

}  // namespace backend
}  // namespace rebvio
