// Stamped pose sample delivered to odometry callbacks (reference types/odometry.hpp).
#pragma once

#include <cstdint>

#include "rebvio/types/definitions.hpp"

namespace rebvio {
namespace types {

struct Odometry {
  uint64_t ts_us;        // [us]
  Vector3f orientation;  // so(3) logarithm of the global rotation
  Vector3f position;
  // diagnostics appended after the reference's three fields (readers of those are unaffected)
  Float scale{1.0};                  // K of the scale filter
  Vector3f gravity{TooN::Zeros};     // g_est
  Vector3f gyro_bias{TooN::Zeros};   // Bg [rad/frame]
  int klm_num{0};                    // directedMatch count of this pair
};

}  // namespace types
}  // namespace rebvio
