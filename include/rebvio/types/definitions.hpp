// Scalar / small-matrix typedefs of the rebvio API (mirrors the names of the reference's types/definitions.hpp:17-53).
// Real TooN / OpenCV are used when installed; otherwise the minimal stand-ins in rebvio/compat/.
#pragma once

#if __has_include(<TooN/TooN.h>)
#include <TooN/TooN.h>
#include <TooN/so3.h>
#else
#include "rebvio/compat/toon_min.hpp"
#endif
#if __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
#else
#include "rebvio/compat/cv_min.hpp"
#endif

#define CV_FLOAT_PRECISION CV_32FC1

namespace rebvio {
namespace types {

using Float = float;

using Vector2f = TooN::Vector<2, Float>;
using Vector3f = TooN::Vector<3, Float>;
using Vector6f = TooN::Vector<6, Float>;
using Vector7f = TooN::Vector<7, Float>;
using Matrix3f = TooN::Matrix<3, 3, Float>;
using Matrix6f = TooN::Matrix<6, 6, Float>;
using Matrix7f = TooN::Matrix<7, 7, Float>;
using Point2Df = TooN::Vector<2, Float>;
using Point3Df = TooN::Vector<3, Float>;

// 3x3 inverse by adjugate over determinant (reference types/definitions.hpp:40-53); defined in host/linalg.cpp
Matrix3f invert(const Matrix3f& in);

}  // namespace types
}  // namespace rebvio
