#include "rebvio/edge_detector.hpp"

#include "session.hpp"

namespace rebvio {

EdgeDetector::EdgeDetector(rebvio::Camera::SharedPtr camera, rebvio::EdgeDetectorConfig::SharedPtr config)
    : config_(config), camera_(camera), session_(backend::Session::forCamera(*camera)) {
  session_->setDetectorConfig(*config_);
}

EdgeDetector::~EdgeDetector() {}

rebvio::EdgeMap::SharedPtr EdgeDetector::detect(rebvio::types::Image& image) {
  rebvio_hip_ctx* ctx = session_->ctx();
  if (image.data.type() != CV_32FC1 || image.data.rows != (int)camera_->rows_ || image.data.cols != (int)camera_->cols_)
    backend::fail("EdgeDetector::detect: image must be CV_32FC1 of the camera's size", -1);
  rebvio_hip_map* h = nullptr;
  backend::check("rebvio_hip_detect", rebvio_hip_detect(ctx, image.data.ptr<float>(0), image.data.step, image.ts_us, &h));
  auto map = std::make_shared<rebvio::EdgeMap>(camera_, config_->keylines_max, image.ts_us);
  map->attach(ctx, h);
  return map;
}

}  // namespace rebvio
