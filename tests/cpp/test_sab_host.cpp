// Analytic cases for rebvio::SABEstimator on the host (no GPU): measurement model of sab_estimator.cpp:38-60,
// F[0..2] = (a_s + g) cos(a) - a_v sin(a), scale = tan(a).
#include <cmath>
#include <cstdio>

#include "rebvio/sab_estimator.hpp"

using namespace rebvio;

static int fails = 0;
#define CHECK(c)                                          \
  do {                                                    \
    if (!(c)) {                                           \
      std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
      ++fails;                                            \
    }                                                     \
  } while (0)

static SABEstimator::Config make(const types::Vector3f& a_v, const types::Vector3f& a_s, const types::Vector7f& xp, float p_scale) {
  types::Matrix3f Rv = TooN::Identity * 1e-6, Rs = TooN::Identity * 1e-6;
  types::Matrix7f Pp = TooN::Zeros;
  Pp(0, 0) = p_scale;
  for (int i = 1; i < 4; ++i) Pp(i, i) = 1e-8f;  // gravity pinned by the prior
  for (int i = 4; i < 7; ++i) Pp(i, i) = 1e-12f;
  return SABEstimator::Config(a_v, a_s, 9.81f, xp, Rv, Rs, 4e4f, Pp);
}

int main() {
  types::Vector7f xp = TooN::Zeros;
  xp[0] = (float)M_PI_4;
  xp[2] = 9.81f;
  const types::Vector3f a_s = TooN::makeVector(0.1f, -9.81f, 0.0f);
  {  // consistent measurements at scale 1: the prior is the fixed point, the gradient vanishes
    auto cfg = make(TooN::makeVector(0.1f, 0.0f, 0.0f), a_s, xp, 1e-2f);
    SABEstimator sab(cfg);
    types::Vector7f X = xp;
    types::Matrix7f JtJ;
    types::Vector7f JtF;
    sab.problem(JtJ, JtF, X);
    for (int i = 0; i < 7; ++i) CHECK(std::fabs(JtF[i]) < 1e-2f * std::sqrt(JtJ(i, i)) + 1e-6f);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) CHECK(std::fabs(JtJ(i, j) - JtJ(j, i)) <= 1e-3f * std::fabs(JtJ(i, j)) + 1e-3f);
    sab.gaussNewton(X, 20);
    CHECK(std::fabs(X[0] - (float)M_PI_4) < 1e-4f);
    CHECK(std::fabs(X[2] - 9.81f) < 1e-3f);
  }
  {  // visual acceleration twice the metric one with a loose scale prior: tan(a) -> 0.5
    auto cfg = make(TooN::makeVector(0.2f, 0.0f, 0.0f), a_s, xp, 1e2f);
    SABEstimator sab(cfg);
    types::Vector7f X = xp;
    const int it = sab.gaussNewton(X, 20);
    CHECK(it == 20);  // tolerances default to 0: all iterations run (sab_estimator.cpp:140-163)
    CHECK(std::fabs(std::tan(X[0]) - 0.5f) < 2e-3f);
    CHECK(std::fabs(X[2] - 9.81f) < 1e-2f);
    for (int i = 4; i < 7; ++i) CHECK(std::fabs(X[i]) <= 5e-1f / 25);  // bias saturation
  }
  {  // State defaults (sab_estimator.hpp:50-64)
    types::ImuStateConfig ic;
    SABEstimator::State st(ic);
    CHECK(std::fabs(st.X[0] - (float)M_PI_4) < 1e-7f && st.X[2] == ic.g_norm && st.X[1] == 0.0f);
    CHECK(st.P(1, 1) == 100.0f && std::fabs(st.P(0, 0) - ic.scale_stdd_dev_init * ic.scale_stdd_dev_init) < 1e-12f);
    CHECK(std::fabs(st.Rg - ic.g_norm_uncertainty * ic.g_norm_uncertainty) < 1e-3f && st.QKp == 5e-6f);
  }
  if (fails == 0) std::printf("ok\n");
  return fails ? 1 : 0;
}
