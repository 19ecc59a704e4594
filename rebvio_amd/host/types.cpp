// Camera, IntegratedImu and the small dense helpers of the host API.
#include <cmath>
#include <vector>
#include <cstring>

#include "../csrc/hostmath.hpp"
#include "rebvio/camera.hpp"
#include "rebvio/types/imu.hpp"

namespace rebvio {

namespace types {
Matrix3f invert(const Matrix3f& in) {
  rh::hm::M3 m;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m.a[i][j] = in(i, j);
  const rh::hm::M3 o = rh::hm::invert3(m);
  Matrix3f out;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out(i, j) = o.a[i][j];
  return out;
}
}  // namespace types

Camera::Camera()
    : fx_(458.654), fy_(457.296), fm_(0.5 * (fx_ + fy_)), cx_(367.215), cy_(248.375), k1_(-0.28340811), k2_(0.07395907),
      k3_(0.0), p1_(0.00019359), p2_(1.76187114e-05), rows_(480), cols_(752) {
  R_c2i_ = TooN::Data(0.0148655429818, -0.999880929698, 0.00414029679422, 0.999557249008, 0.0149672133247, 0.025715529948,
                      -0.0257744366974, 0.00375618835797, 0.999660727178);
  t_c2i_ = TooN::makeVector(-0.0216401454975f, -0.064676986768f, 0.00981073058949f);
}

Camera::Camera(unsigned int rows, unsigned int cols, types::Float fx, types::Float fy, types::Float cx, types::Float cy)
    : fx_(fx), fy_(fy), fm_(0.5 * (fx + fy)), cx_(cx), cy_(cy), k1_(0), k2_(0), k3_(0), p1_(0), p2_(0), rows_(rows), cols_(cols) {
  R_c2i_ = TooN::Identity;
  t_c2i_ = TooN::Zeros;
}

cv::Mat Camera::undistort(cv::Mat& in) {
  // cv::undistort(in, out, K(fm,0,cx;0,fm,cy), D) on an fp32 frame: fixed-point maps (1/32 px) + bilinear remap with a
  // constant-zero border (camera.hpp:54-58). Host version for callers of the public API; rebvio::Rebvio itself sends
  // MONO8 frames through the device front end (rebvio_hip_set_undistort), which implements the same map.
  if (k1_ == 0 && k2_ == 0 && k3_ == 0 && p1_ == 0 && p2_ == 0) return in;
  std::vector<int> map((size_t)2 * in.rows * in.cols);
  rh::hm::undistort_fixed_map(in.rows, in.cols, fm_, fm_, cx_, cy_, k1_, k2_, p1_, p2_, k3_, map.data());
  cv::Mat out(in.rows, in.cols, CV_32FC1);
  for (int v = 0; v < in.rows; ++v) {
    float* o = out.ptr<float>(v);
    for (int u = 0; u < in.cols; ++u) {
      const int iu = map[((size_t)v * in.cols + u) * 2], iv = map[((size_t)v * in.cols + u) * 2 + 1];
      const int sx = iu >> 5, sy = iv >> 5;
      const float ax = (float)(iu & 31) * 0.03125f, ay = (float)(iv & 31) * 0.03125f;
      auto tap = [&](int y, int x) -> float {
        return (x >= 0 && x < in.cols && y >= 0 && y < in.rows) ? in.ptr<float>(y)[x] : 0.0f;
      };
      o[u] = tap(sy, sx) * ((1.0f - ay) * (1.0f - ax)) + tap(sy, sx + 1) * ((1.0f - ay) * ax) + tap(sy + 1, sx) * (ay * (1.0f - ax)) +
             tap(sy + 1, sx + 1) * (ay * ax);
    }
  }
  return out;
}

namespace types {

IntegratedImu::IntegratedImu()
    : n_(0), last_ts_(0), init_ts_(0), dt_(0), R_(TooN::Identity), gyro_(TooN::Zeros), gyro_init_(TooN::Zeros),
      gyro_last_(TooN::Zeros), acc_(TooN::Zeros), dgyro_(TooN::Zeros), cacc_(TooN::Zeros) {}

void IntegratedImu::add(Imu& imu, const Matrix3f& R_c2i) {
  const Vector3f tmp = R_c2i.T() * imu.gyro;
  Float dt;
  if (last_ts_ == 0) {
    n_ = 1;
    init_ts_ = imu.ts;
    last_ts_ = init_ts_;
    dt = 0.005;
    gyro_init_ = imu.gyro;
    gyro_last_ = gyro_init_;
    R_ = TooN::Identity;
    gyro_ = tmp;
    acc_ = R_c2i.T() * imu.acc;
    dgyro_ = TooN::Zeros;
    cacc_ = TooN::Zeros;
  } else {
    ++n_;
    dt = Float(imu.ts - last_ts_) / 1000000.0;
    gyro_ += tmp;
    acc_ += R_c2i.T() * imu.acc;
  }
  R_ = R_ * TooN::SO3<Float>(tmp * dt).get_matrix();
  last_ts_ = imu.ts;
  gyro_last_ = imu.gyro;
}

const IntegratedImu& IntegratedImu::get(const Matrix3f& R_c2i, const Vector3f t_c2i) {
  // unsigned arithmetic as in the reference: with no samples (n_ = 0) this evaluates to dt_ = 0 (imu.hpp:81); with
  // exactly one sample the reference divides 0 by 0 (SIGFPE on x86) - defined here as dt_ = 0
  dt_ = (n_ == 1) ? 0 : (last_ts_ - init_ts_) / (uint64_t)(n_ - 1) * n_;
  if (n_ > 1) {
    gyro_ /= Float(n_);
    acc_ /= Float(n_);
    dgyro_ = R_c2i.T() * (gyro_last_ - gyro_init_) / dt_s();
  }
  cacc_ = acc_ + (dgyro_ ^ (-(R_c2i.T() * t_c2i)));
  n_ = 0;
  init_ts_ = 0;
  last_ts_ = 0;
  return *this;
}

}  // namespace types
}  // namespace rebvio
