// The O(1) glue of one frame pair between extRotVel and directedMatch (rebvio.cpp:177-233, accelerometer / SAB branch
// excluded): sum of the extRotVel block records, 6x6 solve (core.cpp:244-248), gyroBiasCorrection (core.cpp:264-284), SO3
// correction, Cholesky covariance, the inputs of directedMatch rotated by Rback (edge_map.cpp:193-194) and the next pair's
// prior rotation (rebvio.cpp:163-164). ONE source for both places it runs:
//   * on the host, for the per-pair API (rebvio_hip_track_pair; rebvio::Rebvio runs its own fusion between the halves);
//   * on the device, for the streaming and batch drivers: workgroup 0 of the persistent LM kernel runs it at the kernel's tail
//     (track.hip lm_tail_glue -> glue_dev.hpp glue_workgroup, the same statements dealt to the lanes of four waves; k_pair_glue
//     in per-call mode) - no host round trip sits between a pair's two halves there.
// Same statements in the same order, fp contraction off on both sides, and no library call whose rounding could differ
// between host and device: the sine / cosine of SO3::exp come from hm::sincos_det (double polynomial, hostmath.hpp), square
// roots and divisions are IEEE on both. The two therefore give identical BITS, which tests/test_parity_gpu.py checks
// (test_device_glue_equals_host_glue on a stream, test_glue_probe_random_inputs on random and degenerate inputs through
// rebvio_hip_test_glue). TooN's own SO3 / Cholesky / SVD remain tolerance-only against this restatement.
#pragma once

#include "common.hpp"
#include "hostmath.hpp"

namespace rh {

namespace hm {

// R = imu.R(); R.T() = SO3(Bg) * R.T()  (rebvio.cpp:163-164); R_prior = IMU inter-frame rotation (identity without one)
RH_HD inline M3 prior_rotation(const float Bg[3], const M3& R_prior) { return transpose(mul(so3_exp(Bg), transpose(R_prior))); }

// sum of the extRotVel block records in block order, accumulated in double (21 upper-triangle products, 6 JtF, count)
RH_HD inline void sum_xrv(const float* xrv, int nblocks, float Wx[36], float JtF[6]) {
  double acc[27];
  for (int k = 0; k < 27; ++k) acc[k] = 0.0;
  for (int b = 0; b < nblocks; ++b)
    for (int k = 0; k < 27; ++k) acc[k] += (double)xrv[(size_t)b * kXrvStride + k];
  int k = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) {
      Wx[i * 6 + j] = (float)acc[k];
      Wx[j * 6 + i] = (float)acc[k];
      ++k;
    }
  for (int i = 0; i < 6; ++i) JtF[i] = (float)acc[21 + i];
}

// _Rvel = invert(JtJ) (core.cpp:186) from the packed LM state
RH_HD inline void lm_rvel(const LmState& s, float Rvel[9]) {
  M3 J;
  J.a[0][0] = s.JtJ[0]; J.a[1][1] = s.JtJ[1]; J.a[2][2] = s.JtJ[2];
  J.a[0][1] = J.a[1][0] = s.JtJ[3];
  J.a[0][2] = J.a[2][0] = s.JtJ[4];
  J.a[1][2] = J.a[2][1] = s.JtJ[5];
  store3(invert3(J), Rvel);
}

// The whole glue. `st` is the filter state: read, then replaced by the state after this pair (st.R = the prior rotation of
// the NEXT pair under "no IMU prior", which is what the streaming drivers use). `gl` is what the second half of the pair
// reads; `out` the record for the caller.
RH_HD inline void pair_glue_core(const LmState& lm, const float* xrv, int n_new, const GlueParams& gp, GlueState& st, GlueDev& gl,
                                 rebvio_hip_pair_out& out) {
  float Vg[3], P_Vg[9];
  for (int i = 0; i < 3; ++i) Vg[i] = lm.vel[i];
  lm_rvel(lm, P_Vg);
  out.F = lm.F;
  out.lm_accept_mask = lm.accept_mask;
  out.sigma_rho_min = lm.sigma_rho_min;
  float Xv[6], W_Xv[36], JtF6[6];
  sum_xrv(xrv, (n_new + 255) / 256, W_Xv, JtF6);
  {
    double ws[72];
    sym6_solve_ws(W_Xv, JtF6, Xv, ws);
  }
  out.ext_ok = 1;
  for (int i = 0; i < 6; ++i)
    if (Xv[i] != Xv[i]) out.ext_ok = 0;
  float Xgv[6], W_Xgv[36];
  for (int i = 0; i < 6; ++i) Xgv[i] = Xv[i];
  for (int i = 0; i < 36; ++i) W_Xgv[i] = W_Xv[i];
  // rebvio.cpp:186-191
  const float s_b = gp.gyro_bias_std_dev * gp.gyro_bias_std_dev * gp.frame_dt * gp.frame_dt;
  const float s_g = gp.gyro_std_dev * gp.gyro_std_dev * gp.frame_dt * gp.frame_dt;
  const M3 RGBias = diag3(s_b), RGyro = diag3(s_g);
  M3 W_Bg = load3(st.W_Bg);
  float dg[3];
  gyro_bias_correction(Xgv, W_Xgv, W_Bg, RGyro, RGBias, dg);
  for (int i = 0; i < 3; ++i) st.Bg[i] += dg[i];
  store3(W_Bg, st.W_Bg);
  const float dVgv[3] = {Xgv[0], Xgv[1], Xgv[2]};
  const float dWgv[3] = {Xgv[3], Xgv[4], Xgv[5]};
  // rebvio.cpp:195-203
  const M3 R0 = so3_exp(dWgv);
  const M3 R = transpose(mul(R0, transpose(load3(st.R))));
  float V[3];
  mulv(R0, Vg, V);
  for (int i = 0; i < 3; ++i) V[i] += dVgv[i];
  float R_Xgv[36], P_V[9];
  cholesky6_inverse(W_Xgv, R_Xgv);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) P_V[i * 3 + j] = R_Xgv[i * 6 + j];
  // second half: directedMatch prologue (edge_map.cpp:193-194) with Rback = R (rebvio.cpp:228)
  mulv(R, V, gl.vel_r);
  store3(mul(mul(R, load3(P_V)), transpose(R)), gl.Rvel_r);
  store3(R, gl.Rgva);
  store3(R0, gl.R0a);
  for (int i = 0; i < 3; ++i) gl.V[i] = V[i];
  gl.nan_v = (V[0] != V[0] || V[1] != V[1] || V[2] != V[2]) ? 1 : 0;
  // the new map becomes the next pair's old map: its first rotation (prior after this pair's bias update) rides along
  const M3 Rn = prior_rotation(st.Bg, identity3());
  store3(Rn, st.R);
  store3(transpose(Rn), gl.RT_next);
  gl.has_next = 1;
  for (int i = 0; i < 3; ++i) {
    out.Vg[i] = Vg[i];
    out.V[i] = V[i];
  }
  for (int i = 0; i < 9; ++i) {
    out.P_Vg[i] = P_Vg[i];
    out.R[i] = gl.Rgva[i];
    out.P_V[i] = P_V[i];
  }
  for (int i = 0; i < 6; ++i) {
    out.Xv[i] = Xv[i];
    out.Xgv[i] = Xgv[i];
  }
  for (int i = 0; i < 36; ++i) out.W_Xv[i] = W_Xv[i];
  out.klm_num = out.kf_matches = out.reg_num = 0;
  out.status = gl.nan_v ? 1 : 0;
}

}  // namespace hm
}  // namespace rh
