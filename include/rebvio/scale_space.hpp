// rebvio::FastGaussian / rebvio::ScaleSpace — Kovesi's repeated box filtering over fp32 integral images and the
// two-scale DoG + squared gradient magnitude built from it (reference scale_space.hpp:22-96, scale_space.cpp). The
// filtering runs on the GPU behind rebvio_hip_smooth / rebvio_hip_scale_space, with the reference's evaluation order
// (bit-identical integral images, box averages, DoG and gradient); these classes are the thin hosts.
// EdgeDetector does not go through them: its detect() keeps the images on the device.
#pragma once

#include <memory>

#include "rebvio/camera.hpp"

namespace rebvio {

namespace backend {
class Session;
}

class FastGaussian {
 public:
  // any n in 1..16 whose Kovesi widths stay in 3..11 (scale_space.cpp:14-41; the reference itself only constructs n = 3, :186)
  FastGaussian(rebvio::Camera::SharedPtr cam, types::Float sigma, int n = 3);
  FastGaussian() = delete;
  FastGaussian(const FastGaussian&) = delete;
  FastGaussian& operator=(const FastGaussian&) = delete;
  ~FastGaussian();

  // CV_32FC1 in, CV_32FC1 out (camera size)
  cv::Mat smooth(cv::Mat& image);

  int n_;                     // number of box passes
  types::Float sigma_;        // requested standard deviation
  types::Float sigma_true_;   // standard deviation actually realised by the integer box widths
  int* widths_;               // [n_] box widths
  cv::Mat* divisors_;         // [n_] per-pixel reciprocal box areas (host copies; the kernels use an 81-entry table)

 private:
  rebvio::Camera::SharedPtr camera_;
  std::shared_ptr<backend::Session> session_;
};

class ScaleSpace {
 public:
  explicit ScaleSpace(rebvio::Camera::SharedPtr camera);
  ScaleSpace() = delete;
  ~ScaleSpace();

  cv::Mat dog() const;  // scale1 - scale0 of the last build()
  cv::Mat mag() const;  // squared central-difference gradient of scale0 (border 0)
  void build(cv::Mat& image);  // CV_32FC1, values 0..765

 private:
  rebvio::Camera::SharedPtr camera_;
  std::shared_ptr<backend::Session> session_;
  cv::Mat dog_, gradient_mag_;
};

}  // namespace rebvio
