// IMU sample, inter-frame IMU pre-integration and the gyro/visual fusion state (reference types/imu.hpp).
// Host-side O(1) work at IMU rate; nothing here runs on the GPU.
#pragma once

#include <cstdint>
#include <limits>

#include "rebvio/camera.hpp"
#include "rebvio/types/definitions.hpp"

namespace rebvio {
namespace types {

struct Imu {
  uint64_t ts;    // [us]
  Vector3f gyro;  // [rad/s]
  Vector3f acc;   // [m/s^2]
};

class IntegratedImu {
 public:
  IntegratedImu();
  // queue one sample (gyro pre-integration happens here, means are finished by get()) - reference imu.hpp:52-75
  void add(Imu& imu, const Matrix3f& R_c2i);
  // finish the interval: mean rates, angular acceleration, lever-arm compensated acceleration - reference imu.hpp:80-94
  const IntegratedImu& get(const Matrix3f& R_c2i, const Vector3f t_c2i);

  const uint64_t& dt_us() const { return dt_; }
  Float dt_s() const { return Float(dt_) / 1000000.0; }
  const Vector3f& gyro() const { return gyro_; }
  const Vector3f& acc() const { return acc_; }
  const Vector3f& cacc() const { return cacc_; }
  const Matrix3f& R() const { return R_; }

 private:
  unsigned int n_;
  uint64_t last_ts_, init_ts_, dt_;
  Matrix3f R_;
  Vector3f gyro_, gyro_init_, gyro_last_, acc_, dgyro_, cacc_;
};

struct ImuStateConfig {
  Float g_norm{9.81};
  Float g_uncertainty{2e-3};
  Float g_norm_uncertainty{0.2e3};
  Float acc_std_dev{2.0e-3};
  Float gyro_std_dev{1.6968e-04};
  Float gyro_bias_std_dev{1.9393e-05};
  Float vbias_std_dev{1e-7};
  Float scale_std_dev_mult{1e-2};
  Float scale_std_dev_max{1e-4};
  Float scale_stdd_dev_init{1.2e-3};
  int init_bias{1};
  int init_bias_frame_num{10};
  Vector3f init_bias_guess{TooN::makeVector(0.0188f, 0.0037f, 0.0776f)};
};

struct ImuState {
  Vector3f Vg{TooN::Zeros};
  Matrix3f P_Vg{TooN::Identity * std::numeric_limits<Float>::max()};
  Vector3f Vgv{TooN::Zeros}, dVgv{TooN::Zeros}, dWgv{TooN::Zeros};
  Vector3f Vgva{TooN::Zeros}, dVgva{TooN::Zeros}, dWgva{TooN::Zeros};
  Vector3f Bg{TooN::Zeros};
  Matrix3f RGBias{TooN::Identity};
  Matrix3f W_Bg{TooN::Identity * 0.01};  // invert(100 * RGBias)
  Matrix3f RGyro{TooN::Identity};
  Vector3f Av{TooN::Zeros}, As{TooN::Zeros};
  Vector3f u_est{TooN::makeVector(1.0f, 0.0f, 0.0f)};
  bool initialized{false};
};

}  // namespace types
}  // namespace rebvio
