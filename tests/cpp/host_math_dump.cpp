// Runs the host-side fusion math on inputs read from a file of float32 and writes the results as float32,
// so that tests/test_fusion_math.py can check them against INDEPENDENT float64 statements (finite differences of the
// weighted cost, joint normal equations, matrix exponentials) instead of against the oracle's restatement of the same text.
//   host_math_dump <mode> in.f32 out.f32
//   sab   : a_v3 a_s3 G x_p7 Pp49 Rv9 Rs9 Rg X7            -> JtJ49 JtF7 Xgn7 iterations1
//   gbc   : X6 Wx36 Wb9 Rg9 Rb9                            -> X6 Wx36 Wb9 dg3          (Core::gyroBiasCorrection + the C-ABI's hostmath form)
//   gpre  : W_Bg9 s_g s_b n                                 -> n x (gyro_pre's W_Bg9, gyro_bias_correction's W_Bg9)
//   chol6 : A36                                            -> inv36
//   bias  : n x (sacc3 facc3 kP Rot9 Qg9 Qrot9 Qbias9 QKp Rg g_norm Rs9 Rf9 Wvw36 | X7 P49 g_est3 b_est3 Xvw6)
//                                                          -> n x (K | X7 P49 g_est3 b_est3 Xvw6)   (Core::estimateBias; the
//           record layout of REBVIO_DUMP_FUSION, rebvio_amd/host/rebvio.cpp)
//   so3   : w3 a3 b3                                       -> exp(w)9 ln(exp(w))3 R(a->b)9 hostmath exp(w)9
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "../../rebvio_amd/csrc/hostmath.hpp"
#include "rebvio/core.hpp"
#include "rebvio/sab_estimator.hpp"

using namespace rebvio;
namespace hm = rh::hm;

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const std::string mode = argv[1];
  std::ifstream f(argv[2], std::ios::binary);
  std::vector<float> in((std::istreambuf_iterator<char>(f)), {});
  {
    std::ifstream g(argv[2], std::ios::binary | std::ios::ate);
    const size_t bytes = (size_t)g.tellg();
    g.seekg(0);
    in.resize(bytes / sizeof(float));
    g.read(reinterpret_cast<char*>(in.data()), (std::streamsize)bytes);
  }
  std::vector<float> out;
  const float* p = in.data();
  auto v3 = [&](const float* q) { return TooN::makeVector(q[0], q[1], q[2]); };
  auto m3 = [&](const float* q) {
    types::Matrix3f M;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) M(i, j) = q[i * 3 + j];
    return M;
  };
  if (mode == "sab") {
    if (in.size() != 3 + 3 + 1 + 7 + 49 + 9 + 9 + 1 + 7) return 3;
    types::Vector7f xp, X;
    types::Matrix7f Pp;
    for (int i = 0; i < 7; ++i) xp[i] = p[7 + i];
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) Pp(i, j) = p[14 + i * 7 + j];
    for (int i = 0; i < 7; ++i) X[i] = p[82 + i];
    SABEstimator::Config cfg(v3(p), v3(p + 3), p[6], xp, m3(p + 63), m3(p + 72), p[81], Pp);
    SABEstimator sab(cfg);
    types::Matrix7f JtJ;
    types::Vector7f JtF;
    sab.problem(JtJ, JtF, X);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) out.push_back(JtJ(i, j));
    for (int i = 0; i < 7; ++i) out.push_back(JtF[i]);
    types::Vector7f Xg = X;
    const int it = sab.gaussNewton(Xg, 20);
    for (int i = 0; i < 7; ++i) out.push_back(Xg[i]);
    out.push_back((float)it);
  } else if (mode == "gbc") {
    if (in.size() != 6 + 36 + 9 + 9 + 9) return 3;
    Core core(std::make_shared<Camera>());
    types::Vector6f X;
    types::Matrix6f Wx;
    for (int i = 0; i < 6; ++i) X[i] = p[i];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) Wx(i, j) = p[6 + i * 6 + j];
    types::Matrix3f Wb = m3(p + 42);
    const types::Vector3f dg = core.gyroBiasCorrection(X, Wx, Wb, m3(p + 51), m3(p + 60));
    for (int i = 0; i < 6; ++i) out.push_back(X[i]);
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) out.push_back(Wx(i, j));
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) out.push_back(Wb(i, j));
    for (int i = 0; i < 3; ++i) out.push_back(dg[i]);
  } else if (mode == "gpre") {
    // in: W_Bg9 s_g s_b n; out per step k < n: pre[2] of hm::gyro_pre (the information matrix after the pair) and the W_Bg that
    // hm::gyro_bias_correction leaves for the same inputs (9 + 9 floats), the shadow advanced like the streaming driver does
    if (in.size() != 12) return 3;
    hm::M3 shadow = hm::load3(p), W = hm::load3(p);
    const float s_g = p[9], s_b = p[10];
    const int n = (int)p[11];
    for (int k = 0; k < n; ++k) {
      float pre[6][9];
      hm::gyro_pre(shadow, s_g, s_b, pre);
      shadow = hm::load3(pre[2]);
      float X[6] = {0.01f, -0.02f, 0.005f, 1e-3f, -2e-3f, 5e-4f}, Wx[36], dg[3];
      for (int i = 0; i < 36; ++i) Wx[i] = (i % 7 == 0) ? 1000.0f + 10.0f * (float)k : 1.0f;
      hm::gyro_bias_correction(X, Wx, W, hm::diag3(s_g), hm::diag3(s_b), dg);
      for (int i = 0; i < 9; ++i) out.push_back(pre[2][i]);
      float w9[9];
      hm::store3(W, w9);
      for (int i = 0; i < 9; ++i) out.push_back(w9[i]);
    }
  } else if (mode == "bias") {
    if (in.size() % 168 != 0) return 3;
    Core core(std::make_shared<Camera>());
    for (size_t c = 0; c < in.size() / 168; ++c) {
      const float* q = p + c * 168;
      types::Matrix6f W;
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) W(i, j) = q[64 + i * 6 + j];
      const float* s = q + 100;
      types::Vector7f X;
      types::Matrix7f P;
      types::Vector3f g, b;
      types::Vector6f Xvw;
      for (int i = 0; i < 7; ++i) X[i] = s[i];
      for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 7; ++j) P(i, j) = s[7 + i * 7 + j];
      for (int i = 0; i < 3; ++i) {
        g[i] = s[56 + i];
        b[i] = s[59 + i];
      }
      for (int i = 0; i < 6; ++i) Xvw[i] = s[62 + i];
      const float k = core.estimateBias(v3(q), v3(q + 3), q[6], m3(q + 7), X, P, m3(q + 16), m3(q + 25), m3(q + 34), q[43], q[44], m3(q + 46),
                                        m3(q + 55), g, b, W, Xvw, q[45]);
      out.push_back(k);
      for (int i = 0; i < 7; ++i) out.push_back(X[i]);
      for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 7; ++j) out.push_back(P(i, j));
      for (int i = 0; i < 3; ++i) out.push_back(g[i]);
      for (int i = 0; i < 3; ++i) out.push_back(b[i]);
      for (int i = 0; i < 6; ++i) out.push_back(Xvw[i]);
    }
  } else if (mode == "chol6") {
    if (in.size() != 36) return 3;
    float h[36];  // what the host classes and the C-ABI glue use for TooN::Cholesky<6,float>::get_inverse (rebvio.cpp:198)
    hm::cholesky6_inverse(p, h);
    out.insert(out.end(), h, h + 36);
  } else if (mode == "so3") {
    if (in.size() != 9) return 3;
    const TooN::SO3<types::Float> E(v3(p));
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) out.push_back(E.get_matrix()(i, j));
    const types::Vector3f l = E.ln();
    for (int i = 0; i < 3; ++i) out.push_back(l[i]);
    const TooN::SO3<types::Float> AB(v3(p + 3), v3(p + 6));
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) out.push_back(AB.get_matrix()(i, j));
    const hm::M3 H = hm::so3_exp(p);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) out.push_back(H.a[i][j]);
  } else {
    return 2;
  }
  std::ofstream o(argv[3], std::ios::binary);
  o.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)(out.size() * sizeof(float)));
  return 0;
}
