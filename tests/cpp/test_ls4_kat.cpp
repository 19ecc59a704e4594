// The reference's one unit test, ported verbatim in substance (rebvio/test/test_rebvio.cpp:6-18), against the host
// C++ API. Needs no GPU: Core creates its device context lazily.
#include <cmath>
#include <cstdio>

#include "rebvio/rebvio.hpp"

int main() {
  rebvio::Core core(std::make_shared<rebvio::Camera>());
  rebvio::types::Vector3f Vgv = TooN::makeVector(-4.06833e-05f, 9.40667e-05f, 5.70767e-05f);
  rebvio::types::Float dt = 0.05;
  rebvio::types::Vector3f Av = TooN::makeVector(0.0f, 0.0f, 0.0f);
  rebvio::types::Matrix3f R = TooN::Data(1, 8.83134e-05, -7.48149e-05, -8.831e-05, 1, 4.57494e-05, 7.4819e-05, -4.57428e-05, 1);
  core.estimateLs4Acceleration(-Vgv / dt, Av, R, dt);
  const double exp[3] = {0.0162734, -0.0376267, -0.0228307};
  int bad = 0;
  for (int i = 0; i < 3; ++i)
    if (std::fabs(Av[i] - exp[i]) > 0.00001) ++bad;
  std::printf("Av = %.7f %.7f %.7f  (%s)\n", Av[0], Av[1], Av[2], bad ? "FAIL" : "ok");
  // gyroBiasCorrection on an SPD system keeps the state finite and symmetric information
  rebvio::types::Vector6f X;
  rebvio::types::Matrix6f Wx = TooN::Zeros;
  for (int i = 0; i < 6; ++i) {
    X[i] = 0.001f * (i + 1);
    Wx(i, i) = 1e6f * (i + 1);
  }
  rebvio::types::Matrix3f Wb = TooN::Identity * 0.01, Rg = TooN::Identity * 7.2e-11, Rb = TooN::Identity * 9.4e-13;
  rebvio::types::Vector3f dg = core.gyroBiasCorrection(X, Wx, Wb, Rg, Rb);
  for (int i = 0; i < 3; ++i)
    if (!std::isfinite(dg[i]) || !std::isfinite(X[3 + i])) ++bad;
  return bad ? 1 : 0;
}
