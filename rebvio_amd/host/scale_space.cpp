#include "rebvio/scale_space.hpp"

#include <cstring>
#include <stdexcept>
#include <vector>

#include "../csrc/hostmath.hpp"
#include "session.hpp"

namespace rebvio {

using backend::check;
namespace hm = rh::hm;

namespace {
// reciprocal of the number of image pixels under a width-d box centred on every pixel (reference scale_space.cpp:130-171):
// the count is the product of the clipped extents along x and y, the reciprocal is taken in double and rounded once
cv::Mat reciprocal_box_areas(int rows, int cols, int d) {
  cv::Mat out(rows, cols, CV_32FC1);
  const int h = d / 2;
  auto extent = [h, d](int i, int n) { return i <= h ? i + h + 1 : (i >= n - h ? n - i + h : d); };
  for (int r = 0; r < rows; ++r) {
    float* o = out.ptr<float>(r);
    const int ny = extent(r, rows);
    for (int c = 0; c < cols; ++c) o[c] = (float)(1.0 / (double)(float)(extent(c, cols) * ny));
  }
  return out;
}

void require_f32(const cv::Mat& m, const Camera& cam, const char* who) {
  if (m.type() != CV_32FC1 || m.rows != (int)cam.rows_ || m.cols != (int)cam.cols_)
    backend::fail(who, -1);
}

// dense copy of a (possibly strided) fp32 image
std::vector<float> dense(const cv::Mat& m) {
  std::vector<float> v((size_t)m.rows * m.cols);
  for (int r = 0; r < m.rows; ++r) std::memcpy(v.data() + (size_t)r * m.cols, m.ptr<float>(r), (size_t)m.cols * sizeof(float));
  return v;
}
}  // namespace

FastGaussian::FastGaussian(rebvio::Camera::SharedPtr cam, types::Float sigma, int n)
    : n_(n), sigma_(sigma), sigma_true_(0), widths_(nullptr), divisors_(nullptr), camera_(cam),
      session_(backend::Session::forCamera(*cam)) {
  if (n < 1 || n > 16) throw std::invalid_argument("rebvio::FastGaussian: 1..16 box passes");
  widths_ = new int[n_];
  hm::kovesi_widths(sigma_, n_, widths_, &sigma_true_);
  for (int i = 0; i < n_; ++i)
    if (widths_[i] < 3 || widths_[i] > 11) {
      delete[] widths_;
      widths_ = nullptr;
      throw std::invalid_argument("rebvio::FastGaussian: sigma / n ask for a box width outside 3..11 (the device kernels' tile halo)");
    }
  divisors_ = new cv::Mat[n_];
  for (int i = 0; i < n_; ++i) divisors_[i] = reciprocal_box_areas((int)cam->rows_, (int)cam->cols_, widths_[i]);
}

FastGaussian::~FastGaussian() {
  delete[] widths_;
  delete[] divisors_;
}

cv::Mat FastGaussian::smooth(cv::Mat& image) {
  require_f32(image, *camera_, "FastGaussian::smooth: CV_32FC1 image of the camera's size expected");
  const std::vector<float> in = dense(image);
  cv::Mat out((int)camera_->rows_, (int)camera_->cols_, CV_32FC1);
  check("rebvio_hip_smooth_n", rebvio_hip_smooth_n(session_->ctx(), in.data(), widths_, n_, out.ptr<float>(0)));
  return out;
}

ScaleSpace::ScaleSpace(rebvio::Camera::SharedPtr camera) : camera_(camera), session_(backend::Session::forCamera(*camera)) {
  dog_.create((int)camera->rows_, (int)camera->cols_, CV_32FC1);
  gradient_mag_.create((int)camera->rows_, (int)camera->cols_, CV_32FC1);
  std::memset(dog_.data, 0, (size_t)dog_.rows * dog_.step);
  std::memset(gradient_mag_.data, 0, (size_t)gradient_mag_.rows * gradient_mag_.step);
}

ScaleSpace::~ScaleSpace() {}

cv::Mat ScaleSpace::dog() const { return dog_; }
cv::Mat ScaleSpace::mag() const { return gradient_mag_; }

void ScaleSpace::build(cv::Mat& image) {
  require_f32(image, *camera_, "ScaleSpace::build: CV_32FC1 image of the camera's size expected");
  const std::vector<float> in = dense(image);
  check("rebvio_hip_scale_space",
        rebvio_hip_scale_space(session_->ctx(), in.data(), nullptr, nullptr, dog_.ptr<float>(0), gradient_mag_.ptr<float>(0)));
}

}  // namespace rebvio
