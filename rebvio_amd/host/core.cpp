#include "rebvio/core.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>

#include "../csrc/hostmath.hpp"
#include "session.hpp"

namespace rebvio {

using backend::check;
namespace hm = rh::hm;

// ---- DistanceField (reference core.hpp:20-79) ------------------------------------------------------------------------
DistanceField::DistanceField(int rows, int cols, types::Float search_range)
    : rows_((unsigned)rows), cols_((unsigned)cols), search_range_(search_range) {}

DistanceField::~DistanceField() {}

void DistanceField::build(rebvio::EdgeMap::SharedPtr map) {
  if (!map || !map->handle()) backend::fail("DistanceField::build: the map has no device keylines (maps come from EdgeDetector::detect)", -1);
  check("rebvio_hip_build_distance_field", rebvio_hip_build_distance_field(map->ctx(), map->handle()));
  map_ = map;
  mirror_valid_ = false;
}

void DistanceField::syncMirror() {
  if (mirror_valid_) return;
  const size_t n = (size_t)rows_ * cols_;
  std::vector<int> ids(n), dist(n);
  // (a size or search range that differs from the stream's context shows up here: the ABI call sizes by the context)
  check("rebvio_hip_map_distance_field", rebvio_hip_map_distance_field(map_->handle(), ids.data(), dist.data()));
  field_.resize(n);
  for (size_t i = 0; i < n; ++i) {
    field_[i].id = ids[i];
    field_[i].distance = dist[i];
  }
  mirror_valid_ = true;
}

DistanceFieldElement& DistanceField::operator[](int index) {
  if (!map_) {  // never built: every cell empty, like the reference's freshly constructed field
    if (field_.empty()) field_.resize((size_t)rows_ * cols_);
    return field_[index];
  }
  syncMirror();
  return field_[index];
}

// ---- Core ------------------------------------------------------------------------------------------------------------
Core::Core(rebvio::Camera::SharedPtr camera, rebvio::CoreConfig::SharedPtr config)
    : config_(config), camera_(camera), session_(backend::Session::forCamera(*camera)),
      distance_field_((int)camera->rows_, (int)camera->cols_, config->search_range) {
  session_->setCoreConfig(*config_);
  for (auto& v : ls4_V_) v = TooN::Zeros;
  for (auto& v : mean_A_) v = TooN::Zeros;
  std::memset(ls4_T_, 0, sizeof(ls4_T_));
  std::memset(ls4_Dt_, 0, sizeof(ls4_Dt_));
}

Core::~Core() {}

CoreConfig::SharedPtr Core::config() { return config_; }

void Core::buildDistanceField(rebvio::EdgeMap::SharedPtr map) {
  if (map->ctx() != session_->ctx()) backend::fail("Core::buildDistanceField: the map belongs to another camera stream", -1);
  distance_field_.build(map);
}

// |g1.g2 - g2.g2| / (g2.g2) <= threshold (reference core.cpp:39-44)
bool Core::testfk(const rebvio::types::KeyLine& keyline1, const rebvio::types::KeyLine& keyline2,
                  const types::Float& similarity_threshold) {
  const types::Float g22 = keyline2.gradient_norm * keyline2.gradient_norm;
  const types::Float g12 = keyline1.gradient[0] * keyline2.gradient[0] + keyline1.gradient[1] * keyline2.gradient[1];
  return !(std::fabs(g12 - g22) > similarity_threshold * g22);
}

// Distance-field lookup of one reprojected keyline (reference core.cpp:46-76): weighted residual along the matched
// keyline's unit gradient and its two image-space derivatives; no cell / dissimilar gradients -> the range penalty with
// zero derivatives (and `fi` untouched, which is what tryVel's carry-forward rule rests on).
types::Float Core::calculatefJ(rebvio::EdgeMap::SharedPtr, int f_inx, types::Float& df_dx, types::Float& df_dy,
                               rebvio::types::KeyLine& keyline, const types::Float& px, const types::Float& py, int& mnum,
                               types::Float& fi) {
  const int id = distance_field_[f_inx].id;
  const types::KeyLine* target = id >= 0 ? &(*distance_field_.map())[id] : nullptr;
  if (!target || !Core::testfk(*target, keyline, config_->match_treshold)) {
    df_dx = 0.0;
    df_dy = 0.0;
    return config_->search_range / keyline.sigma_rho;
  }
  const types::Float ux = target->gradient[0] / target->gradient_norm;
  const types::Float uy = target->gradient[1] / target->gradient_norm;
  fi = (px - target->pos[0]) * ux + (py - target->pos[1]) * uy;
  df_dx = ux / keyline.sigma_rho;
  df_dy = uy / keyline.sigma_rho;
  ++mnum;
  keyline.match_id_forward = id;
  return fi / keyline.sigma_rho;
}

// Scalar depth EKF of one matched keyline (reference core.cpp:424-456), operation for operation what k_depth_ekf /
// k_regularize_ekf evaluate per thread (the same float / double promotions).
void Core::updateInverseDepthARLU(rebvio::types::KeyLine& k, rebvio::types::Vector3f& vel) {
  const float fm = camera_->fm_;
  const float vx = vel[0], vy = vel[1], vz = vel[2];
  float v_rho = k.sigma_rho * k.sigma_rho;
  const float ux = k.match_gradient[0] / k.match_gradient_norm;
  const float uy = k.match_gradient[1] / k.match_gradient_norm;
  const float Y = ux * (k.pos_img[0] - k.match_pos_img[0]) + uy * (k.pos_img[1] - k.match_pos_img[1]);
  const float H = ux * (vx * fm - vz * k.match_pos_img[0]) + uy * (vy * fm - vz * k.match_pos_img[1]);
  const float rho_p = (float)(1.0 / (1.0 / (double)k.rho + (double)vz));
  float F = (float)(1.0 / (1.0 + (double)(k.rho * vz)));
  F *= F;
  const float p_p = F * v_rho * F + config_->reshape_q_abs * config_->reshape_q_abs;
  const float e = Y - H * rho_p;
  const float S = H * p_p * H + config_->pixel_uncertainty * config_->pixel_uncertainty;
  const float K = (float)((double)(p_p * H) * (1.0 / (double)S));
  float rho = rho_p + K * e;
  v_rho = (float)((1.0 - (double)(K * H)) * (double)p_p);
  float sig = std::sqrt(v_rho);
  if (rho < types::RHO_MIN) {
    sig += types::RHO_MIN - rho;
    rho = types::RHO_MIN;
  } else if (rho > types::RHO_MAX) {
    rho = types::RHO_MAX;
  } else if (std::isnan(rho) || std::isnan(sig) || std::isinf(rho) || std::isinf(sig)) {
    std::fprintf(stderr, "ERROR NaN or INF RHO in updateInverseDepthARLU()!\n");
    rho = types::RHO_INIT;
    sig = types::RHO_MAX;
  }
  k.rho = rho;
  k.sigma_rho = sig;
}

types::Float Core::tryVel(rebvio::EdgeMap::SharedPtr map, rebvio::types::Matrix3f& JtJ, rebvio::types::Vector3f& JtF,
                          const rebvio::types::Vector3f& vel, types::Float sigma_rho_min, types::Float* residuals) {
  const float v[3] = {vel[0], vel[1], vel[2]};
  float o[10];
  check("rebvio_hip_try_vel", rebvio_hip_try_vel(session_->ctx(), map->handle(), v, sigma_rho_min, residuals, o));
  JtJ(0, 0) = o[1]; JtJ(1, 1) = o[2]; JtJ(2, 2) = o[3];
  JtJ(0, 1) = JtJ(1, 0) = o[4];
  JtJ(0, 2) = JtJ(2, 0) = o[5];
  JtJ(1, 2) = JtJ(2, 1) = o[6];
  JtF[0] = o[7]; JtF[1] = o[8]; JtF[2] = o[9];
  map->invalidateMirror();
  return o[0];
}

types::Float Core::minimizeVel(rebvio::EdgeMap::SharedPtr map, rebvio::types::Vector3f& vel, rebvio::types::Matrix3f& Rvel) {
  float v[3] = {vel[0], vel[1], vel[2]}, r[9], F = 0.f;
  check("rebvio_hip_minimize_vel", rebvio_hip_minimize_vel(session_->ctx(), map->handle(), v, r, &F, nullptr, nullptr));
  for (int i = 0; i < 3; ++i) {
    vel[i] = v[i];
    for (int j = 0; j < 3; ++j) Rvel(i, j) = r[i * 3 + j];
  }
  map->invalidateMirror();
  return F;
}

bool Core::extRotVel(rebvio::EdgeMap::SharedPtr, const rebvio::types::Vector3f& vel, rebvio::types::Matrix6f& Wx,
                     rebvio::types::Vector6f& X) {
  // like the reference (core.cpp:196-200) this works on the distance field's map, the argument is unused
  const float v[3] = {vel[0], vel[1], vel[2]};
  float w[36], x[6];
  int ok = 0;
  check("rebvio_hip_ext_rot_vel", rebvio_hip_ext_rot_vel(session_->ctx(), v, w, nullptr, x, &ok));
  for (int i = 0; i < 6; ++i) {
    X[i] = x[i];
    for (int j = 0; j < 6; ++j) Wx(i, j) = w[i * 6 + j];
  }
  return ok != 0;
}

rebvio::types::Vector3f Core::gyroBiasCorrection(rebvio::types::Vector6f& X, rebvio::types::Matrix6f& Wx, rebvio::types::Matrix3f& Wb,
                                                 const rebvio::types::Matrix3f& Rg, const rebvio::types::Matrix3f& Rb) {
  float x[6], w[36], dg[3];
  hm::M3 wb, rg, rb;
  for (int i = 0; i < 6; ++i) {
    x[i] = X[i];
    for (int j = 0; j < 6; ++j) w[i * 6 + j] = Wx(i, j);
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      wb.a[i][j] = Wb(i, j);
      rg.a[i][j] = Rg(i, j);
      rb.a[i][j] = Rb(i, j);
    }
  hm::gyro_bias_correction(x, w, wb, rg, rb, dg);
  for (int i = 0; i < 6; ++i) {
    X[i] = x[i];
    for (int j = 0; j < 6; ++j) Wx(i, j) = w[i * 6 + j];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Wb(i, j) = wb.a[i][j];
  return TooN::makeVector(dg[0], dg[1], dg[2]);
}

void Core::estimateLs4Acceleration(const rebvio::types::Vector3f& vel, rebvio::types::Vector3f& acc, const rebvio::types::Matrix3f& R,
                                   types::Float dt) {
  // least-squares slope of the last five velocity samples (reference core.cpp:285-333), history kept per instance
  types::Vector3f* V = ls4_V_;  // V[0] newest ... V[4] oldest
  const types::Matrix3f RT = R.T();
  V[4] = RT * V[3];
  V[3] = RT * V[2];
  V[2] = RT * V[1];
  V[1] = RT * V[0];
  V[0] = vel;
  types::Float* T = ls4_T_;
  types::Float* Dt = ls4_Dt_;
  for (int i = 0; i < 3; ++i) Dt[i] = Dt[i + 1];
  Dt[3] = dt;
  T[0] = 0.0;
  types::Float mt = 0.0;
  for (int i = 0; i < 4; ++i) {
    T[i + 1] = T[i] + Dt[i];
    mt += T[i + 1];
  }
  mt /= 5.0;
  types::Float den = 0.0;
  for (int i = 0; i < 5; ++i) den += (T[i] - mt) * (T[i] - mt);
  for (int i = 0; i < 3; ++i) {
    // the reference's mean reads V[3] out of bounds for its fifth term (core.cpp:321); the weights sum to zero, so the
    // mean cancels in `num` whatever that value is - use 0
    const types::Float vm = (V[0][i] + V[1][i] + V[2][i] + V[3][i] + 0.0f) / 5.0;
    types::Float num = (V[0][i] - vm) * (T[4] - mt);
    num += (V[1][i] - vm) * (T[3] - mt);
    num += (V[2][i] - vm) * (T[2] - mt);
    num += (V[3][i] - vm) * (T[1] - mt);
    num += (V[4][i] - vm) * (T[0] - mt);
    if (den > 0.0) acc[i] = num / den;
  }
}

void Core::estimateMeanAcceleration(const rebvio::types::Vector3f sacc, rebvio::types::Vector3f& acc, const rebvio::types::Matrix3f& R) {
  const types::Matrix3f RT = R.T();  // reference core.cpp:335-347
  mean_A_[3] = RT * mean_A_[2];
  mean_A_[2] = RT * mean_A_[1];
  mean_A_[1] = RT * mean_A_[0];
  mean_A_[0] = sacc;
  acc = (mean_A_[0] + mean_A_[1] + mean_A_[2] + mean_A_[3]) * types::Float(0.25);
}

void Core::updateInverseDepth(rebvio::types::Vector3f& vel) {
  const float v[3] = {vel[0], vel[1], vel[2]};
  check("rebvio_hip_update_inverse_depth", rebvio_hip_update_inverse_depth(session_->ctx(), v));
  if (distance_field_.map()) distance_field_.map()->invalidateMirror();
}

}  // namespace rebvio
