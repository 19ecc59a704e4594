// rebvio::types::KeyLine — the 84-byte host record of one edge pixel. Field names and order are the reference's
// (types/keyline.hpp:24-40) because callers index them directly ((*map)[i].pos[0], ros_rebvio.cpp:44-45); on the
// device keylines live as SoA arrays and this AoS form is only the lazily downloaded mirror.
#pragma once

#include <cmath>

#include "rebvio/types/definitions.hpp"

namespace rebvio {
namespace types {

constexpr Float RHO_MAX = 20.0;
constexpr Float RHO_MIN = 1e-3;
constexpr Float RHO_INIT = 1.0;

struct KeyLine {
  Point2Df pos;
  Point2Df pos_img;
  Point2Df match_pos_img;
  Vector2f gradient;
  Vector2f match_gradient;
  Float gradient_norm;
  Float match_gradient_norm;
  Float rho;
  Float sigma_rho;
  int id;
  int id_prev;
  int id_next;
  int match_id;
  int match_id_forward;
  int match_id_keyframe;
  unsigned int matches;

  KeyLine() = default;  // the mirror is filled by a bulk download (the reference deletes this ctor)
  KeyLine(const Point2Df& p, const Vector2f& g, const Point2Df& p_img)
      : pos(p), pos_img(p_img), match_pos_img(p_img), gradient(g), match_gradient(TooN::Zeros),
        gradient_norm(std::sqrt(g[0] * g[0] + g[1] * g[1])), match_gradient_norm(0.0), rho(RHO_INIT), sigma_rho(RHO_MAX),
        id(-1), id_prev(-1), id_next(-1), match_id(-1), match_id_forward(-1), match_id_keyframe(-1), matches(0) {}
};
static_assert(sizeof(KeyLine) == 84, "KeyLine must stay layout-compatible with rebvio_hip_keyline");

}  // namespace types
}  // namespace rebvio
