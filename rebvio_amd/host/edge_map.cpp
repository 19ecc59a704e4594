#include "rebvio/edge_map.hpp"

#include <cstring>

#include "session.hpp"

namespace rebvio {

using backend::check;

EdgeMap::EdgeMap(rebvio::Camera::SharedPtr camera, int size, uint64_t ts_us, rebvio::EdgeMapConfig::SharedPtr config)
    : config_(config), camera_(camera), ts_us_(ts_us), threshold_(-1.0) {
  keylines_.reserve(size);
}

EdgeMap::~EdgeMap() {
  if (handle_) rebvio_hip_map_release(handle_);
}

void EdgeMap::attach(rebvio_hip_ctx* ctx, rebvio_hip_map* handle, std::shared_ptr<void> keepalive) {
  keepalive_ = std::move(keepalive);
  ctx_ = ctx;
  handle_ = handle;
  invalidateMirror();
}

void EdgeMap::syncMirror() {
  if (mirror_valid_ || !handle_) return;
  const int n = rebvio_hip_map_size(handle_);
  if (n < 0) backend::fail("rebvio_hip_map_size", n);
  keylines_.resize((size_t)n);
  static_assert(sizeof(types::KeyLine) == sizeof(rebvio_hip_keyline), "AoS mirror layout");
  check("rebvio_hip_map_download",
        rebvio_hip_map_download(handle_, reinterpret_cast<rebvio_hip_keyline*>(keylines_.data()), nullptr));
  mirror_valid_ = true;
}

rebvio::types::KeyLine& EdgeMap::operator[](int idx) {
  syncMirror();
  return keylines_[idx];
}

int EdgeMap::size() {
  if (!handle_) return (int)keylines_.size();
  const int n = rebvio_hip_map_size(handle_);
  if (n < 0) backend::fail("rebvio_hip_map_size", n);
  return n;
}

std::vector<rebvio::types::KeyLine>& EdgeMap::keylines() {
  syncMirror();
  return keylines_;
}

uint64_t EdgeMap::ts_us() { return ts_us_; }

const types::Float& EdgeMap::threshold() const {
  if (handle_) threshold_ = rebvio_hip_map_threshold(handle_);
  return threshold_;
}

void EdgeMap::threshold(const types::Float& t) { threshold_ = t; }  // device maps take their threshold from detect()

rebvio::types::IntegratedImu& EdgeMap::imu() { return imu_; }

std::unordered_map<unsigned int, unsigned int>& EdgeMap::mask() {
  if (!mask_valid_ && handle_) {
    std::vector<int> dense((size_t)camera_->rows_ * camera_->cols_);
    check("rebvio_hip_map_download", rebvio_hip_map_download(handle_, nullptr, dense.data()));
    keylines_mask_.clear();
    for (size_t i = 0; i < dense.size(); ++i)
      if (dense[i] >= 0) keylines_mask_.emplace((unsigned)i, (unsigned)dense[i]);
    mask_valid_ = true;
  }
  return keylines_mask_;
}

types::Float EdgeMap::estimateQuantile(types::Float percentile, int num_bins) {
  float out = 1e3f;
  check("rebvio_hip_quantile", rebvio_hip_quantile(ctx_, handle_, percentile, num_bins, &out));
  return out;
}

void EdgeMap::rotateKeylines(const rebvio::types::Matrix3f& R) {
  float r[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r[i * 3 + j] = R(i, j);
  check("rebvio_hip_rotate", rebvio_hip_rotate(ctx_, handle_, r));
  invalidateMirror();
}

int EdgeMap::forwardMatch(rebvio::EdgeMap::SharedPtr map) {
  check("rebvio_hip_forward_match", rebvio_hip_forward_match(ctx_, handle_, map->handle_));
  map->invalidateMirror();
  return 0;  // the reference's count is order dependent and unused (rebvio.cpp:172)
}

int EdgeMap::searchMatch(const rebvio::types::KeyLine& keyline, const rebvio::types::Vector3f& vel, const rebvio::types::Matrix3f& Rvel,
                         const rebvio::types::Matrix3f& Rback, types::Float max_radius) {
  float v[3] = {vel[0], vel[1], vel[2]}, rv[9], rb[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      rv[i * 3 + j] = Rvel(i, j);
      rb[i * 3 + j] = Rback(i, j);
    }
  int idx = -1;
  check("rebvio_hip_search_match",
        rebvio_hip_search_match(ctx_, handle_, reinterpret_cast<const rebvio_hip_keyline*>(&keyline), v, rv, rb, max_radius, &idx));
  return idx;
}

int EdgeMap::directedMatch(rebvio::EdgeMap::SharedPtr map, const rebvio::types::Vector3f& vel, const rebvio::types::Matrix3f& Rvel,
                           const rebvio::types::Matrix3f& Rback, int& kf_matches, types::Float max_radius) {
  float v[3] = {vel[0], vel[1], vel[2]}, rv[9], rb[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      rv[i * 3 + j] = Rvel(i, j);
      rb[i * 3 + j] = Rback(i, j);
    }
  int n = 0, kf = 0;
  check("rebvio_hip_directed_match", rebvio_hip_directed_match(ctx_, handle_, map->handle_, v, rv, rb, max_radius, &n, &kf));
  kf_matches = kf;
  invalidateMirror();
  return n;
}

int EdgeMap::regularize1Iter() {
  int n = 0;
  check("rebvio_hip_regularize", rebvio_hip_regularize(ctx_, handle_, &n));
  invalidateMirror();
  return n;
}

}  // namespace rebvio
