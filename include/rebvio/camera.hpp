// Pinhole camera constants + camera->IMU extrinsics. Public fields as in the reference (camera.hpp:61-72); unlike the
// reference (EuRoC 752x480 hard-coded, camera.hpp:25-45) the resolution / intrinsics can be set, which the
// 640x480 and 1280x960 configurations need.
#pragma once

#include <memory>

#include "rebvio/types/definitions.hpp"

namespace rebvio {

class Camera {
 public:
  using SharedPtr = std::shared_ptr<rebvio::Camera>;

  Camera();  // EuRoC MH cam0 defaults of the reference
  Camera(unsigned int rows, unsigned int cols, types::Float fx, types::Float fy, types::Float cx, types::Float cy);

  const types::Matrix3f& getRc2i() const { return R_c2i_; }
  const types::Vector3f& getTc2i() const { return t_c2i_; }
  void setExtrinsics(const types::Matrix3f& R_c2i, const types::Vector3f& t_c2i) {
    R_c2i_ = R_c2i;
    t_c2i_ = t_c2i;
  }

  // Rad-tan undistortion with the camera matrix K(fm, fm, cx, cy), like cv::undistort(in, out, K_, D_) (camera.hpp:54-58).
  // Identity when all distortion coefficients are zero. Host bilinear remap (tolerance-level parity with OpenCV only).
  cv::Mat undistort(cv::Mat& in);

  types::Float fx_, fy_, fm_, cx_, cy_;
  types::Float k1_, k2_, k3_, p1_, p2_;
  unsigned int rows_, cols_;

 private:
  types::Matrix3f R_c2i_;
  types::Vector3f t_c2i_;
};

}  // namespace rebvio
