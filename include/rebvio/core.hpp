// rebvio::Core — the reference's "edge tracker" (core.hpp:82-205): distance field, translation LM (tryVel /
// minimizeVel), 6-DoF linear step (extRotVel), per-keyline depth EKF; plus the O(1) inertial helpers that stay on
// the host. Hot methods forward to the gfx950 backend through the C-ABI.
#pragma once

#include <limits>
#include <memory>
#include <vector>

#include "rebvio/edge_map.hpp"

namespace rebvio {

// One cell of the distance field (reference core.hpp:15-18): the keyline whose +-search_range gradient segment passes
// through the pixel at the smallest |r|, and that |r|.
struct DistanceFieldElement {
  int id = {-1};
  int distance = {std::numeric_limits<int>::max()};
};

// rebvio::DistanceField (reference core.hpp:20-79). build() runs on the device (rebvio_hip_build_distance_field: the
// field belongs to the map it is built from and stays in HBM, where tryVel reads it); operator[] reads a host mirror that
// is downloaded on first use after a build. rows / cols / search_range must be those of the camera stream the map
// belongs to (the device context is created with them); build() throws std::runtime_error otherwise.
class DistanceField {
 public:
  DistanceField(int rows, int cols, types::Float search_range);
  ~DistanceField();
  DistanceField(const DistanceField&) = delete;
  DistanceField& operator=(const DistanceField&) = delete;

  void build(rebvio::EdgeMap::SharedPtr map);
  DistanceFieldElement& operator[](int index);
  rebvio::EdgeMap::SharedPtr map() { return map_; }

 private:
  void syncMirror();
  std::vector<DistanceFieldElement> field_;
  unsigned int rows_, cols_;
  types::Float search_range_;
  rebvio::EdgeMap::SharedPtr map_;
  bool mirror_valid_ = true;
};

struct CoreConfig {
  types::Float search_range{40.0};
  types::Float reweight_distance{2.0};
  types::Float match_treshold{0.5};
  unsigned int min_match_threshold{0};
  unsigned int iterations{5};
  unsigned int global_min_matches_threshold{500};
  types::Float pixel_uncertainty{1};
  types::Float quantile_cutoff{0.9};
  int quantile_num_bins{100};
  types::Float reshape_q_abs{1e-4};
  using SharedPtr = std::shared_ptr<CoreConfig>;
};

namespace backend {
class Session;
}

class Core {
 public:
  Core(rebvio::Camera::SharedPtr camera, rebvio::CoreConfig::SharedPtr config = std::make_shared<rebvio::CoreConfig>());
  Core() = delete;
  ~Core();

  CoreConfig::SharedPtr config();
  void buildDistanceField(rebvio::EdgeMap::SharedPtr map);
  // Single-keyline forms of the reference's public helpers (core.hpp:122,138,187), evaluated on the host with the
  // kernels' arithmetic; calculatefJ reads the distance field of buildDistanceField through its host mirror.
  static bool testfk(const rebvio::types::KeyLine& keyline1, const rebvio::types::KeyLine& keyline2,
                     const types::Float& similarity_threshold);
  types::Float calculatefJ(rebvio::EdgeMap::SharedPtr map, int f_inx, types::Float& df_dx, types::Float& df_dy,
                           rebvio::types::KeyLine& keyline, const types::Float& px, const types::Float& py, int& mnum,
                           types::Float& fi);
  void updateInverseDepthARLU(rebvio::types::KeyLine& keyline, rebvio::types::Vector3f& vel);
  types::Float tryVel(rebvio::EdgeMap::SharedPtr map, rebvio::types::Matrix3f& JtJ, rebvio::types::Vector3f& JtF,
                      const rebvio::types::Vector3f& vel, types::Float sigma_rho_min, types::Float* residuals);
  types::Float minimizeVel(rebvio::EdgeMap::SharedPtr map, rebvio::types::Vector3f& vel, rebvio::types::Matrix3f& Rvel);
  bool extRotVel(rebvio::EdgeMap::SharedPtr map, const rebvio::types::Vector3f& vel, rebvio::types::Matrix6f& Wx,
                 rebvio::types::Vector6f& X);
  rebvio::types::Vector3f gyroBiasCorrection(rebvio::types::Vector6f& X, rebvio::types::Matrix6f& Wx, rebvio::types::Matrix3f& Wb,
                                             const rebvio::types::Matrix3f& Rg, const rebvio::types::Matrix3f& Rb);
  void estimateLs4Acceleration(const rebvio::types::Vector3f& vel, rebvio::types::Vector3f& acc, const rebvio::types::Matrix3f& R,
                               types::Float dt);
  void estimateMeanAcceleration(const rebvio::types::Vector3f sacc, rebvio::types::Vector3f& acc, const rebvio::types::Matrix3f& R);
  types::Float estimateBias(const rebvio::types::Vector3f& sacc, const rebvio::types::Vector3f& facc, types::Float kP,
                            const rebvio::types::Matrix3f Rot, rebvio::types::Vector7f& X, rebvio::types::Matrix7f& P,
                            const rebvio::types::Matrix3f& Qg, const rebvio::types::Matrix3f& Qrot, const rebvio::types::Matrix3f& Qbias,
                            types::Float QKp, types::Float Rg, const rebvio::types::Matrix3f& Rs, const rebvio::types::Matrix3f& Rf,
                            rebvio::types::Vector3f& g_est, rebvio::types::Vector3f& b_est, const rebvio::types::Matrix6f& Wvw,
                            rebvio::types::Vector6f& Xvw, types::Float g_gravit);
  void updateInverseDepth(rebvio::types::Vector3f& vel);

  std::shared_ptr<backend::Session> session() { return session_; }

 private:
  rebvio::CoreConfig::SharedPtr config_;
  rebvio::Camera::SharedPtr camera_;
  std::shared_ptr<backend::Session> session_;
  rebvio::DistanceField distance_field_;
  // per-instance history of the two acceleration estimators (function-static in the reference, core.cpp:287-293,335-338)
  types::Vector3f ls4_V_[5];
  types::Float ls4_T_[5], ls4_Dt_[4];
  types::Vector3f mean_A_[4];
};

}  // namespace rebvio
