// Stamped camera frame handed to Rebvio::imageCallback (reference types/image.hpp).
#pragma once

#include <cstdint>

#include "rebvio/types/definitions.hpp"

namespace rebvio {
namespace types {

struct Image {
  uint64_t ts_us;  // [us]
  cv::Mat data;    // MONO8 on input; fp32 x3.0 after Rebvio::imageCallback
};

}  // namespace types
}  // namespace rebvio
