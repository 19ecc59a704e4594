#include "session.hpp"

#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>

namespace rebvio {
namespace backend {

namespace {
thread_local int t_scope = 0;
thread_local int t_device = 0;
std::mutex g_mu;
int g_next_scope = 1;
using Key = std::tuple<int, unsigned, unsigned, float, float, float>;
std::map<Key, std::weak_ptr<Session>> g_sessions;
}  // namespace

void fail(const char* what, int rc) {
  throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + rebvio_hip_last_error());
}

int Session::newScope(int device_id) {
  std::lock_guard<std::mutex> g(g_mu);
  t_scope = g_next_scope++;
  t_device = device_id;
  return t_scope;
}

std::shared_ptr<Session> Session::forCamera(const Camera& cam) {
  std::lock_guard<std::mutex> g(g_mu);
  const Key k{t_scope, cam.rows_, cam.cols_, cam.fm_, cam.cx_, cam.cy_};
  auto it = g_sessions.find(k);
  if (it != g_sessions.end())
    if (auto s = it->second.lock()) return s;
  std::shared_ptr<Session> s(new Session);
  rebvio_hip_default_params(&s->p_, (int)cam.rows_, (int)cam.cols_);
  s->p_.fm = cam.fm_;
  s->p_.cx = cam.cx_;
  s->p_.cy = cam.cy_;
  s->p_.device_id = t_device;
  const float K4[4] = {cam.fm_, cam.fm_, cam.cx_, cam.cy_};  // K_ uses the mean focal length on both axes (camera.hpp:39)
  const float D5[5] = {cam.k1_, cam.k2_, cam.p1_, cam.p2_, cam.k3_};
  std::memcpy(s->K4_, K4, sizeof(K4));
  std::memcpy(s->D5_, D5, sizeof(D5));
  g_sessions[k] = s;
  return s;
}

void Session::setDetectorConfig(const EdgeDetectorConfig& c) {
  std::lock_guard<std::mutex> g(mu_);
  if (ctx_) return;  // parameters are frozen once the device context exists
  p_.keylines_ref = c.keylines_ref;
  p_.keylines_max = c.keylines_max;
  p_.pos_neg_threshold = c.pos_neg_threshold;
  p_.dog_threshold = c.dog_threshold;
  p_.threshold = c.threshold;
  p_.gain = c.gain;
  p_.max_threshold = c.max_threshold;
  p_.min_threshold = c.min_threshold;
}

void Session::setCoreConfig(const CoreConfig& c) {
  std::lock_guard<std::mutex> g(mu_);
  if (ctx_) return;
  p_.search_range = c.search_range;
  p_.reweight_distance = c.reweight_distance;
  p_.match_treshold = c.match_treshold;
  p_.min_match_threshold = c.min_match_threshold;
  p_.iterations = c.iterations;
  p_.global_min_matches_threshold = c.global_min_matches_threshold;
  p_.pixel_uncertainty = c.pixel_uncertainty;
  p_.quantile_cutoff = c.quantile_cutoff;
  p_.quantile_num_bins = c.quantile_num_bins;
  p_.reshape_q_abs = c.reshape_q_abs;
}

void Session::setImuNoise(float gyro_std_dev, float gyro_bias_std_dev) {
  std::lock_guard<std::mutex> g(mu_);
  if (ctx_) return;
  p_.gyro_std_dev = gyro_std_dev;
  p_.gyro_bias_std_dev = gyro_bias_std_dev;
}

rebvio_hip_ctx* Session::ctx() {
  std::lock_guard<std::mutex> g(mu_);
  if (!ctx_) {
    check("rebvio_hip_create", rebvio_hip_create(&p_, &ctx_));
    check("rebvio_hip_set_undistort", rebvio_hip_set_undistort(ctx_, K4_, D5_));
  }
  return ctx_;
}

Session::~Session() {
  if (ctx_) rebvio_hip_destroy(ctx_);
}

}  // namespace backend
}  // namespace rebvio
