// rebvio::EdgeMap — per-frame keyline container + the per-keyline tracking operations (reference edge_map.hpp:28-135).
// Here the keylines live on the GPU (SoA, owned by a pooled rebvio_hip_map); the host-visible std::vector<KeyLine>,
// operator[] and mask() are lazily downloaded mirrors, so callers such as ros_rebvio.cpp:44-46 keep working and
// nobody pays a device->host copy unless they look.
#pragma once

#include <memory>
#include <unordered_map>
#include <vector>

#include "rebvio/types/imu.hpp"
#include "rebvio/types/keyline.hpp"

struct rebvio_hip_map;
struct rebvio_hip_ctx;

namespace rebvio {

struct EdgeMapConfig {
  types::Float pixel_uncertainty_match{2.0};
  types::Float match_threshold_norm{1.0};
  types::Float match_threshold_angle{45.0};
  types::Float regularization_threshold{0.5};
  using SharedPtr = std::shared_ptr<rebvio::EdgeMapConfig>;
};

class EdgeMap {
 public:
  using SharedPtr = std::shared_ptr<rebvio::EdgeMap>;

  // Reference signature (edge_map.hpp:34): an empty host-side map. Maps with keylines come from EdgeDetector::detect.
  EdgeMap(rebvio::Camera::SharedPtr camera, int size, uint64_t ts_us,
          rebvio::EdgeMapConfig::SharedPtr config = std::make_shared<rebvio::EdgeMapConfig>());
  ~EdgeMap();
  EdgeMap(const EdgeMap&) = delete;
  EdgeMap& operator=(const EdgeMap&) = delete;

  rebvio::types::KeyLine& operator[](int idx);
  int size();
  std::vector<rebvio::types::KeyLine>& keylines();
  uint64_t ts_us();
  const types::Float& threshold() const;
  void threshold(const types::Float& t);
  rebvio::types::IntegratedImu& imu();
  std::unordered_map<unsigned int, unsigned int>& mask();

  types::Float estimateQuantile(types::Float percentile, int num_bins);
  void rotateKeylines(const rebvio::types::Matrix3f& R);
  int forwardMatch(rebvio::EdgeMap::SharedPtr map);
  // One keyline of ANOTHER map searched in this one (reference edge_map.hpp:93-94); index of the match or -1.
  int searchMatch(const rebvio::types::KeyLine& keyline, const rebvio::types::Vector3f& vel, const rebvio::types::Matrix3f& Rvel,
                  const rebvio::types::Matrix3f& Rback, types::Float max_radius);
  int directedMatch(rebvio::EdgeMap::SharedPtr map, const rebvio::types::Vector3f& vel, const rebvio::types::Matrix3f& Rvel,
                    const rebvio::types::Matrix3f& Rback, int& kf_matches, types::Float max_radius);
  int regularize1Iter();

  // --- backend plumbing (not part of the reference surface) ---
  // `keepalive` owns the backend context the handle belongs to (backend::Session): a map kept by an edge-image consumer
  // (ros_rebvio.cpp:32-51) past ~Rebvio keeps its context alive and releases into it.
  void attach(rebvio_hip_ctx* ctx, rebvio_hip_map* handle, std::shared_ptr<void> keepalive = nullptr);
  rebvio_hip_map* handle() const { return handle_; }
  rebvio_hip_ctx* ctx() const { return ctx_; }
  void invalidateMirror() { mirror_valid_ = false; mask_valid_ = false; }

 private:
  void syncMirror();
  rebvio::EdgeMapConfig::SharedPtr config_;
  rebvio::Camera::SharedPtr camera_;
  uint64_t ts_us_;
  std::shared_ptr<void> keepalive_;  // declared before the handle: destroyed after the destructor body has released it
  rebvio_hip_ctx* ctx_ = nullptr;
  rebvio_hip_map* handle_ = nullptr;
  std::vector<rebvio::types::KeyLine> keylines_;
  std::unordered_map<unsigned int, unsigned int> keylines_mask_;
  bool mirror_valid_ = true, mask_valid_ = true;
  mutable types::Float threshold_;
  rebvio::types::IntegratedImu imu_;
};

}  // namespace rebvio
