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
};

}  // namespace types
}  // namespace rebvio
