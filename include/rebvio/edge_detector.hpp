// rebvio::EdgeDetector — DoG edge detector (reference edge_detector.hpp:19-97): detect(Image) -> EdgeMap.
// The work (scale space, keyline extraction with ordered compaction, chaining, auto threshold, distance field)
// runs as HIP kernels behind rebvio_hip_detect; this class is the thin host.
#pragma once

#include <memory>

#include "rebvio/camera.hpp"
#include "rebvio/edge_map.hpp"
#include "rebvio/types/image.hpp"

namespace rebvio {

struct EdgeDetectorConfig {
  int keylines_ref{12000};
  int keylines_max{16000};
  static constexpr int plane_fit_size{2};
  types::Float pos_neg_threshold{0.4};
  types::Float dog_threshold{0.095259868922420};
  types::Float threshold{0.01};
  types::Float gain{5e-7};
  types::Float max_threshold{0.5};
  types::Float min_threshold{0.005};
  static constexpr int num_bins{100};
  using SharedPtr = std::shared_ptr<rebvio::EdgeDetectorConfig>;
};

namespace backend {
class Session;
}

class EdgeDetector {
 public:
  EdgeDetector(rebvio::Camera::SharedPtr camera,
               rebvio::EdgeDetectorConfig::SharedPtr config = std::make_shared<rebvio::EdgeDetectorConfig>());
  EdgeDetector() = delete;
  ~EdgeDetector();

  // Asynchronous on the GPU: returns as soon as the kernels are enqueued; size()/keylines() of the map synchronise.
  rebvio::EdgeMap::SharedPtr detect(rebvio::types::Image& image);

 private:
  EdgeDetectorConfig::SharedPtr config_;
  rebvio::Camera::SharedPtr camera_;
  std::shared_ptr<backend::Session> session_;
};

}  // namespace rebvio
