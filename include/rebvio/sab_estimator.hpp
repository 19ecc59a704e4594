// rebvio::SABEstimator — 7-state scale / attitude / visual-bias Gauss-Newton filter (reference sab_estimator.hpp:20-86,
// Tarrio & Pedre 2017 eq. 40). O(1) host math, called once per frame pair by Core::estimateBias.
#pragma once

#include "rebvio/types/definitions.hpp"
#include "rebvio/types/imu.hpp"

namespace rebvio {

class SABEstimator {
 public:
  struct Config {
    rebvio::types::Vector3f a_v;  // visual acceleration
    rebvio::types::Vector3f a_s;  // gravity-corrected (sensed) acceleration
    rebvio::types::Float G;       // standard gravity
    rebvio::types::Vector7f x_p;  // prior state
    rebvio::types::Matrix7f Pp;   // prior covariance
    rebvio::types::Matrix3f Rv, Rs;
    rebvio::types::Float Rg;
    Config(const rebvio::types::Vector3f& a_v_, const rebvio::types::Vector3f& a_s_, rebvio::types::Float G_,
           const rebvio::types::Vector7f& x_p_, const rebvio::types::Matrix3f& Rv_, const rebvio::types::Matrix3f& Rs_,
           rebvio::types::Float Rg_, const rebvio::types::Matrix7f& Pp_)
        : a_v(a_v_), a_s(a_s_), G(G_), x_p(x_p_), Pp(Pp_), Rv(Rv_), Rs(Rs_), Rg(Rg_) {}
    Config() = delete;
  };

  struct State {
    rebvio::types::Vector7f X;  // [angle(scale), gravity(3), visual rotation bias(3)]
    rebvio::types::Vector3f g_est, b_est;
    rebvio::types::Matrix7f P;
    rebvio::types::Matrix3f Qrot, Qg, Qbias;
    types::Float QKp;
    types::Float Rg;
    rebvio::types::Matrix3f Rs, Rv;
    explicit State(rebvio::types::ImuStateConfig& config);
  };

  explicit SABEstimator(SABEstimator::Config& config);
  SABEstimator() = delete;
  ~SABEstimator();
  bool problem(rebvio::types::Matrix7f& JtJ, rebvio::types::Vector7f& JtF, const rebvio::types::Vector7f& X);
  int gaussNewton(rebvio::types::Vector7f& X, int iter_max, rebvio::types::Float a_tol = 0.0, rebvio::types::Float r_tol = 0.0);

 private:
  SABEstimator::Config config_;
  // inverse of the prior covariance Pp: constant over the Gauss-Newton iterations of one estimator (computed on first use)
  rebvio::types::Matrix7f Wp_;
  bool Wp_valid_ = false;
};

}  // namespace rebvio
