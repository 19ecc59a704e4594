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
  if (image.data.rows != (int)camera_->rows_ || image.data.cols != (int)camera_->cols_)
    backend::fail("EdgeDetector::detect: image must have the camera's size", -1);
  rebvio_hip_map* h = nullptr;
  if (image.data.type() == CV_8UC1) {
    // raw MONO8 frame: convertTo(CV_32F, 3.0) and undistort run on the device in front of the scale space
    backend::check("rebvio_hip_detect_u8", rebvio_hip_detect_u8(ctx, image.data.ptr<unsigned char>(0), image.data.step, image.ts_us, &h));
  } else if (image.data.type() == CV_32FC1) {
    // the reference's contract: an already converted and undistorted fp32 frame (rebvio.cpp:43-47)
    backend::check("rebvio_hip_detect", rebvio_hip_detect(ctx, image.data.ptr<float>(0), image.data.step, image.ts_us, &h));
  } else {
    backend::fail("EdgeDetector::detect: image must be CV_8UC1 or CV_32FC1", -1);
  }
  auto map = std::make_shared<rebvio::EdgeMap>(camera_, config_->keylines_max, image.ts_us);
  map->attach(ctx, h, session_);
  return map;
}

}  // namespace rebvio
