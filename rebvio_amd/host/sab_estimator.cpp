// Scale-attitude-bias estimator and Core::estimateBias on the host (reference sab_estimator.cpp:21-165,
// core.cpp:350-414). Dense 7x7 / 11x11 fp32 algebra in TooN's evaluation order; SVD<7> (double in the reference,
// sab_estimator.cpp:31) is a Jacobi pseudo-inverse in double with TooN's 1e9 condition cut.
#include "rebvio/sab_estimator.hpp"

#include <cmath>

#include "../csrc/hostmath.hpp"
#include "rebvio/core.hpp"
#include "session.hpp"

namespace rebvio {
namespace backend {
thread_local FusionCounters t_fusion;
}

namespace {
template <int R, int C>
struct Mx {
  float a[R][C];
  static Mx zeros() {
    Mx m;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) m.a[i][j] = 0;
    return m;
  }
};
template <int R, int K, int C>
Mx<R, C> mul(const Mx<R, K>& x, const Mx<K, C>& y) {
  Mx<R, C> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      float s = 0;
      for (int k = 0; k < K; ++k) s += x.a[i][k] * y.a[k][j];
      r.a[i][j] = s;
    }
  return r;
}
template <int R, int C>
Mx<C, R> tr(const Mx<R, C>& x) {
  Mx<C, R> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) r.a[j][i] = x.a[i][j];
  return r;
}
template <int R, int C>
void mulv(const Mx<R, C>& m, const float* v, float* out) {
  float t[R];
  for (int i = 0; i < R; ++i) {
    float s = 0;
    for (int k = 0; k < C; ++k) s += m.a[i][k] * v[k];
    t[i] = s;
  }
  for (int i = 0; i < R; ++i) out[i] = t[i];
}
template <int N>
float dot(const float* a, const float* b) {
  float s = 0;
  for (int i = 0; i < N; ++i) s += a[i] * b[i];
  return s;
}
template <int N>
float quad(const float* x, const Mx<N, N>& M, const float* y) {
  float t[N];
  for (int j = 0; j < N; ++j) {
    float s = 0;
    for (int k = 0; k < N; ++k) s += x[k] * M.a[k][j];
    t[j] = s;
  }
  return dot<N>(t, y);
}
// TooN Cholesky<N,float>::get_inverse (LDL^T)
template <int N>
Mx<N, N> chol_inverse(const Mx<N, N>& A) {
  float L[N][N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) L[i][j] = A.a[i][j];
  for (int col = 0; col < N; ++col) {
    float inv_diag = 1;
    for (int row = col; row < N; ++row) {
      float val = L[row][col];
      for (int c2 = 0; c2 < col; ++c2) val -= L[c2][col] * L[row][c2];
      if (row == col) {
        L[row][col] = val;
        if (val == 0) break;
        inv_diag = 1 / val;
      } else {
        L[col][row] = val;
        L[row][col] = val * inv_diag;
      }
    }
  }
  Mx<N, N> inv;
  for (int c = 0; c < N; ++c) {
    float y[N], res[N];
    for (int i = 0; i < N; ++i) {
      float val = (i == c) ? 1.0f : 0.0f;
      for (int j = 0; j < i; ++j) val -= L[i][j] * y[j];
      y[i] = val;
    }
    for (int i = 0; i < N; ++i) y[i] /= L[i][i];
    for (int i = N - 1; i >= 0; --i) {
      float val = y[i];
      for (int j = i + 1; j < N; ++j) val -= L[j][i] * res[j];
      res[i] = val;
    }
    for (int i = 0; i < N; ++i) inv.a[i][c] = res[i];
  }
  return inv;
}
// pseudo-inverse solve of a symmetric system in double (Jacobi eigen-decomposition, 1e9 condition cut)
template <int N>
void sym_pinv_solve_d(const double* A_, const double* b_, double* x_) {
  double A[N][N], V[N][N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      A[i][j] = 0.5 * (A_[i * N + j] + A_[j * N + i]);
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 80; ++sweep) {
    // cyclic Jacobi with the relative criterion: an off-diagonal element that is negligible against its two diagonal
    // elements is left alone, a sweep without any rotation ends the iteration (an absolute threshold never triggers when
    // the spectrum spans 1e15, as the information matrix of this filter does)
    int rotations = 0;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        if (A[p][q] == 0.0 || std::fabs(A[p][q]) <= 1e-17 * std::sqrt(std::fabs(A[p][p] * A[q][q]))) continue;
        ++rotations;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = ((theta >= 0) ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < N; ++k) {
          const double akp = A[k][p], akq = A[k][q];
          A[k][p] = cs * akp - sn * akq;
          A[k][q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < N; ++k) {
          const double apk = A[p][k], aqk = A[q][k];
          A[p][k] = cs * apk - sn * aqk;
          A[q][k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < N; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = cs * vkp - sn * vkq;
          V[k][q] = sn * vkp + cs * vkq;
        }
      }
    if (rotations == 0) break;
  }
  double dmax = 0;
  for (int i = 0; i < N; ++i) dmax = std::max(dmax, std::fabs(A[i][i]));
  for (int i = 0; i < N; ++i) x_[i] = 0;
  for (int k = 0; k < N; ++k) {
    const double lam = A[k][k];
    if (!(std::fabs(lam) * 1e9 > dmax)) continue;
    double proj = 0;
    for (int i = 0; i < N; ++i) proj += V[i][k] * b_[i];
    proj /= lam;
    for (int i = 0; i < N; ++i) x_[i] += V[i][k] * proj;
  }
}
// h = pinv(A) b as SVD<7>::backsub gives it (sab_estimator.cpp:31-32). For a well-conditioned symmetric positive definite A
// the pseudo-inverse is the inverse: LDL^T in double (~200 flops); when a pivot falls below the 1e9 condition cut relative
// to the largest diagonal entry, the Jacobi pseudo-inverse (minimum-norm solution, like the SVD) takes over.
template <int N>
void sym_solve_d(const double* A_, const double* b_, double* x_) {
  double L[N][N], d[N], dmax = 0;
  bool ok = true;
  for (int i = 0; i < N; ++i) dmax = std::max(dmax, std::fabs(A_[i * N + i]));
  for (int j = 0; j < N && ok; ++j) {
    double v = A_[j * N + j];
    for (int k = 0; k < j; ++k) v -= L[j][k] * L[j][k] * d[k];
    if (!(v * 1e7 > dmax)) ok = false;
    d[j] = v;
    for (int i = j + 1; i < N && ok; ++i) {
      double s = 0.5 * (A_[i * N + j] + A_[j * N + i]);
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k] * d[k];
      L[i][j] = s / v;
    }
  }
  if (!ok) {
    ++backend::t_fusion.pinv_solves;
    sym_pinv_solve_d<N>(A_, b_, x_);
    return;
  }
  double y[N];
  for (int i = 0; i < N; ++i) {
    double s = b_[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
    y[i] = s;
  }
  for (int i = 0; i < N; ++i) y[i] /= d[i];
  for (int i = N - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < N; ++k) s -= L[k][i] * x_[k];
    x_[i] = s;
  }
}
inline float saturate(float t, float limit) { return (t > limit) ? limit : ((t < -limit) ? -limit : t); }
}  // namespace

SABEstimator::State::State(rebvio::types::ImuStateConfig& cfg) {
  Qg = TooN::Identity * (double)(cfg.g_uncertainty * cfg.g_uncertainty);
  Rg = cfg.g_norm_uncertainty * cfg.g_norm_uncertainty;
  Rs = TooN::Identity * (double)(cfg.acc_std_dev * cfg.acc_std_dev);
  Qbias = TooN::Identity * (double)(cfg.vbias_std_dev * cfg.vbias_std_dev);
  Qrot = TooN::Identity;
  Rv = TooN::Identity;
  QKp = 5e-6;
  g_est = TooN::Zeros;
  b_est = TooN::Zeros;
  X = TooN::Zeros;
  X[0] = M_PI_4;
  X[2] = cfg.g_norm;
  P = TooN::Zeros;
  P(0, 0) = cfg.scale_stdd_dev_init * cfg.scale_stdd_dev_init;
  P(1, 1) = P(2, 2) = P(3, 3) = 100.0;
  P(4, 4) = P(5, 5) = P(6, 6) = cfg.vbias_std_dev * cfg.vbias_std_dev * 1e1;
}

SABEstimator::SABEstimator(SABEstimator::Config& config) : config_(config) {}
SABEstimator::~SABEstimator() {}

bool SABEstimator::problem(rebvio::types::Matrix7f& JtJ_, rebvio::types::Vector7f& JtF_, const rebvio::types::Vector7f& X) {
  const Config& cfg = config_;
  const float a = X[0];
  const float g[3] = {X[1], X[2], X[3]}, b[3] = {X[4], X[5], X[6]};
  float F[11];
  for (int i = 0; i < 11; ++i) F[i] = 0;
  const float ca = std::cos(a), sa = std::sin(a);
  for (int i = 0; i < 3; ++i) F[i] = (cfg.a_s[i] + g[i]) * ca - cfg.a_v[i] * sa;
  F[3] = dot<3>(g, g) - cfg.G * cfg.G;
  F[4] = X[0] - cfg.x_p[0];
  if (F[4] > M_PI) F[4] -= 2.0 * M_PI;
  else if (F[4] < -M_PI) F[4] += 2.0 * M_PI;
  const rh::hm::M3 Rb = rh::hm::so3_exp(b);
  float Rg3[3];
  rh::hm::mulv(Rb, g, Rg3);
  for (int i = 0; i < 3; ++i) F[5 + i] = Rg3[i] - cfg.x_p[1 + i];
  for (int i = 0; i < 3; ++i) F[8 + i] = b[i] - cfg.x_p[4 + i];
  float dFda[11];
  for (int i = 0; i < 11; ++i) dFda[i] = 0;
  for (int i = 0; i < 3; ++i) dFda[i] = -(cfg.a_s[i] + g[i]) * sa - cfg.a_v[i] * ca;
  dFda[4] = 1.0;
  // dF/dx1 (11 x 6) has 24 structural non-zeros - rows 0-2: ca on the diagonal; row 3: 2 g; rows 5-7: [Rb | (Rb g)x]; rows 8-10:
  // [0 | I] - and W = blockdiag(Wz, 1 / Rg, Wp), dW/da = blockdiag(dWz, 0, 0). The reference multiplies the dense 11 x 11 /
  // 11 x 6 matrices (sab_estimator.cpp:100-160); every term skipped below is a product with one of those exact zeros, and
  // a sum that starts at +0 does not change when +-0 is added to it, so the sums keep their terms in the dense order and their
  // bits (checked against recorded calls of the dense form: tests/test_fusion_math.py).
  const float Gx[3][3] = {{0.0f, Rg3[2], -Rg3[1]}, {-Rg3[2], 0.0f, Rg3[0]}, {Rg3[1], -Rg3[0], 0.0f}};
  float g2[3];
  for (int j = 0; j < 3; ++j) g2[j] = 2.0 * g[j];
  Mx<3, 3> Pz;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Pz.a[i][j] = sa * sa * cfg.Rv(i, j) + ca * ca * cfg.Rs(i, j);
  const Mx<3, 3> Wz = chol_inverse<3>(Pz);
  const float W33 = 1.0 / cfg.Rg;
  if (!Wp_valid_) {  // Cholesky<7>(Pp).get_inverse() does not depend on X: once per estimator, not once per iteration
    Mx<7, 7> Pp;
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) Pp.a[i][j] = cfg.Pp(i, j);
    const Mx<7, 7> Wp = chol_inverse<7>(Pp);
    for (int i = 0; i < 7; ++i)
      for (int j = 0; j < 7; ++j) Wp_(i, j) = Wp.a[i][j];
    Wp_valid_ = true;
  }
  float Wp[7][7];
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) Wp[i][j] = Wp_(i, j);
  // dP/da is non-zero in the 3x3 measurement block only, so dW/da = -W dP/da W and dW/da P dW/da live there too
  Mx<3, 3> dPz;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) dPz.a[i][j] = 2.0 * sa * ca * (cfg.Rv(i, j) - cfg.Rs(i, j));
  Mx<3, 3> dWz = mul(mul(Wz, dPz), Wz);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) dWz.a[i][j] = -dWz.a[i][j];
  // x^T W y restricted to the non-zeros of x (rows 0-2 and row 4 of W's blocks): t = x^T W, then the dot product with y
  float tdW[11];  // dFda^T W: columns 0-2 through Wz, column 3 is 0 (dFda[3] = 0), columns 4-10 = 1.0 * Wp[0][.]
  for (int j = 0; j < 3; ++j) {
    float acc = 0;
    for (int k = 0; k < 3; ++k) acc += dFda[k] * Wz.a[k][j];
    tdW[j] = acc;
  }
  tdW[3] = 0;
  for (int j = 0; j < 7; ++j) tdW[4 + j] = dFda[4] * Wp[0][j];
  {
    const Mx<3, 3> Mz = mul(mul(dWz, Pz), dWz);
    float q3 = 0;  // quad<11>(dFda, W, dFda): dFda is non-zero at 0-2 and 4
    for (int j = 0; j < 3; ++j) q3 += tdW[j] * dFda[j];
    q3 += tdW[4] * dFda[4];
    JtJ_(0, 0) = 0.25 * quad<3>(F, Mz, F) + quad<3>(dFda, dWz, F) + q3;
  }
  float t1[3], t2[11];  // dWda F (rows 0-2; the rest is 0) and W dFda
  for (int i = 0; i < 3; ++i) {
    float a1 = 0, a2 = 0;
    for (int k = 0; k < 3; ++k) {
      a1 += dWz.a[i][k] * F[k];
      a2 += Wz.a[i][k] * dFda[k];
    }
    t1[i] = a1;
    t2[i] = a2;
  }
  t2[3] = 0;
  for (int i = 0; i < 7; ++i) t2[4 + i] = Wp[i][0] * dFda[4];
  // dT v for the transpose of dF/dx1 (6 x 11): row i < 3 has ca at k = i, 2 g[i] at k = 3, Rb[k - 5][i] at k = 5-7; row 3 + i has
  // Gx[k - 5][i] at k = 5-7 and 1 at k = 8 + i. Terms in increasing k, as the dense product meets them.
  auto dT_times = [&](const float* v, bool v3_zero, float* out) {
    for (int i = 0; i < 3; ++i) {
      float acc = 0;
      acc += ca * v[i];
      if (!v3_zero) acc += g2[i] * v[3];
      for (int k = 0; k < 3; ++k) acc += Rb.a[k][i] * v[5 + k];
      out[i] = acc;
    }
    for (int i = 0; i < 3; ++i) {
      float acc = 0;
      for (int k = 0; k < 3; ++k) acc += Gx[k][i] * v[5 + k];
      acc += 1.0f * v[8 + i];
      out[3 + i] = acc;
    }
  };
  {
    float c2[6];
    dT_times(t2, true, c2);
    for (int i = 0; i < 6; ++i) {
      const float c1 = (i < 3) ? ca * t1[i] : 0.0f;  // dT (dWda F): only k = i < 3 contributes
      JtJ_(1 + i, 0) = 0.5 * c1 + c2[i];
      JtJ_(0, 1 + i) = JtJ_(1 + i, 0);
    }
  }
  {
    // T1 = dT W (6 x 11), then B = T1 dF/dx1 (6 x 6)
    float T1[6][11];
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) T1[i][j] = ca * Wz.a[i][j];
      T1[i][3] = g2[i] * W33;
      for (int j = 0; j < 7; ++j) {
        float acc = 0;
        for (int k = 0; k < 3; ++k) acc += Rb.a[k][i] * Wp[1 + k][j];
        T1[i][4 + j] = acc;
      }
    }
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 4; ++j) T1[3 + i][j] = 0;
      for (int j = 0; j < 7; ++j) {
        float acc = 0;
        for (int k = 0; k < 3; ++k) acc += Gx[k][i] * Wp[1 + k][j];
        acc += 1.0f * Wp[4 + i][j];
        T1[3 + i][4 + j] = acc;
      }
    }
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j < 3; ++j) {
        float acc = 0;
        if (i < 3) {
          acc += T1[i][j] * ca;
          acc += T1[i][3] * g2[j];
        }
        for (int k = 0; k < 3; ++k) acc += T1[i][5 + k] * Rb.a[k][j];
        JtJ_(1 + i, 1 + j) = acc;
      }
      for (int j = 0; j < 3; ++j) {
        float acc = 0;
        for (int k = 0; k < 3; ++k) acc += T1[i][5 + k] * Gx[k][j];
        acc += T1[i][8 + j] * 1.0f;
        JtJ_(1 + i, 4 + j) = acc;
      }
    }
  }
  {
    float q = 0;  // quad<11>(dFda, W, F) = (dFda^T W) . F
    for (int j = 0; j < 3; ++j) q += tdW[j] * F[j];
    for (int j = 4; j < 11; ++j) q += tdW[j] * F[j];
    JtF_[0] = 0.5 * quad<3>(F, dWz, F) + q;
  }
  {
    float t[11], c[6];  // W F, then dT (W F)
    for (int i = 0; i < 3; ++i) {
      float acc = 0;
      for (int k = 0; k < 3; ++k) acc += Wz.a[i][k] * F[k];
      t[i] = acc;
    }
    t[3] = W33 * F[3];
    for (int i = 0; i < 7; ++i) {
      float acc = 0;
      for (int k = 0; k < 7; ++k) acc += Wp[i][k] * F[4 + k];
      t[4 + i] = acc;
    }
    dT_times(t, false, c);
    for (int i = 0; i < 6; ++i) JtF_[1 + i] = c[i];
  }
  return true;
}

int SABEstimator::gaussNewton(rebvio::types::Vector7f& X, int iter_max, types::Float a_tol, types::Float r_tol) {
  int i = 0;
  ++backend::t_fusion.gn_calls;
  for (; i < iter_max; ++i) {
    ++backend::t_fusion.gn_iterations;
    types::Matrix7f JtJ;
    types::Vector7f JtF;
    problem(JtJ, JtF, X);
    double A[49], b[7], h[7];
    for (int r = 0; r < 7; ++r) {
      b[r] = -(double)JtF[r];
      for (int c = 0; c < 7; ++c) A[r * 7 + c] = (double)JtJ(r, c);
    }
    sym_solve_d<7>(A, b, h);
    float Xn[7];
    for (int r = 0; r < 7; ++r) Xn[r] = (float)((double)X[r] + h[r]);
    Xn[0] = std::atan2(std::sin(Xn[0]), std::cos(Xn[0]));
    for (int r = 4; r < 7; ++r) Xn[r] = saturate(Xn[r], 5e-1 / 25);
    bool moved = false;
    for (int r = 0; r < 7; ++r) {
      moved = moved || !(Xn[r] == X[r]);
      X[r] = Xn[r];
    }
    if (!moved && a_tol <= 0 && r_tol <= 0) {
      // fixed point in fp32: every further iteration would evaluate the same problem at the same X and leave it where it is,
      // so running them (the reference does: its default tolerances are zero) cannot change the result
      i = iter_max;
      break;
    }
    double nh = 0;
    for (int r = 0; r < 7; ++r) nh += h[r] * h[r];
    nh = std::sqrt(nh);
    float nx = 0;
    for (int r = 0; r < 7; ++r) nx += X[r] * X[r];
    if (nh < a_tol || nh / (std::sqrt(nx) + 1e-20) < r_tol) break;
  }
  return i;
}

// Core::estimateBias (core.cpp:350-414)
types::Float Core::estimateBias(const rebvio::types::Vector3f& sacc, const rebvio::types::Vector3f& facc, types::Float kP,
                                const rebvio::types::Matrix3f Rot, rebvio::types::Vector7f& X, rebvio::types::Matrix7f& P,
                                const rebvio::types::Matrix3f& Qg, const rebvio::types::Matrix3f& Qrot,
                                const rebvio::types::Matrix3f& Qbias, types::Float QKp, types::Float Rg,
                                const rebvio::types::Matrix3f& Rs, const rebvio::types::Matrix3f& Rf, rebvio::types::Vector3f& g_est,
                                rebvio::types::Vector3f& b_est, const rebvio::types::Matrix6f& Wvw, rebvio::types::Vector6f& Xvw,
                                types::Float g_gravit) {
  Mx<7, 7> F = Mx<7, 7>::zeros();
  F.a[0][0] = kP;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) F.a[1 + i][1 + j] = Rot(j, i);
  for (int i = 0; i < 3; ++i) F.a[4 + i][4 + i] = 1.0f;
  const float G[3] = {X[1], X[2], X[3]};
  Mx<3, 3> GProd = {{{0.0f, G[2], -G[1]}, {-G[2], 0.0f, G[0]}, {G[1], -G[0], 0.0f}}};
  Mx<3, 3> Qr, Qgm;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Qr.a[i][j] = Qrot(i, j);
      Qgm.a[i][j] = Qg(i, j);
    }
  Mx<7, 7> Q = Mx<7, 7>::zeros();
  const float tn = std::tan(X[0]);
  Q.a[0][0] = QKp / (1.0 + tn * tn);
  const Mx<3, 3> Qg2 = mul(mul(tr(GProd), Qr), GProd);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Q.a[1 + i][1 + j] = Qg2.a[i][j] + Qgm.a[i][j];
      Q.a[4 + i][4 + j] = Qbias(i, j);
    }
  // F = blockdiag(kP, Rot^T, I): the dense products F X, (F P) F^T of core.cpp:372-376 restricted to F's non-zeros, terms in the
  // dense order (a skipped term is a product with an exact zero, see SABEstimator::problem)
  float Xa[7], Xp[7];
  for (int i = 0; i < 7; ++i) Xa[i] = X[i];
  Xp[0] = F.a[0][0] * Xa[0];
  for (int i = 1; i < 4; ++i) {
    float acc = 0;
    for (int k = 1; k < 4; ++k) acc += F.a[i][k] * Xa[k];
    Xp[i] = acc;
  }
  for (int i = 4; i < 7; ++i) Xp[i] = F.a[i][i] * Xa[i];
  for (int i = 0; i < 7; ++i) X[i] = Xp[i];
  Mx<7, 7> Pm;
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) Pm.a[i][j] = P(i, j);
  Mx<7, 7> FP, Ppm;
  for (int j = 0; j < 7; ++j) {
    FP.a[0][j] = F.a[0][0] * Pm.a[0][j];
    for (int i = 1; i < 4; ++i) {
      float acc = 0;
      for (int k = 1; k < 4; ++k) acc += F.a[i][k] * Pm.a[k][j];
      FP.a[i][j] = acc;
    }
    for (int i = 4; i < 7; ++i) FP.a[i][j] = F.a[i][i] * Pm.a[i][j];
  }
  for (int i = 0; i < 7; ++i) {
    Ppm.a[i][0] = FP.a[i][0] * F.a[0][0];
    for (int j = 1; j < 4; ++j) {
      float acc = 0;
      for (int k = 1; k < 4; ++k) acc += FP.a[i][k] * F.a[j][k];
      Ppm.a[i][j] = acc;
    }
    for (int j = 4; j < 7; ++j) Ppm.a[i][j] = FP.a[i][j] * F.a[j][j];
  }
  types::Matrix7f Pp;
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) Pp(i, j) = Ppm.a[i][j] + Q.a[i][j];

  rebvio::SABEstimator::Config params(facc, sacc, g_gravit, X, Rf, Rs, Rg, Pp);
  rebvio::SABEstimator sab(params);
  sab.gaussNewton(X, 20);
  types::Matrix7f JtJ;
  types::Vector7f JtF;
  sab.problem(JtJ, JtF, X);
  Mx<7, 7> J;
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) J.a[i][j] = JtJ(i, j);
  const Mx<7, 7> Pi = chol_inverse<7>(J);
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) P(i, j) = Pi.a[i][j];
  types::Float k = std::tan(X[0]);
  if (k < 0 || std::isnan(k) || std::isinf(k)) k = 0;
  for (int i = 0; i < 3; ++i) {
    g_est[i] = X[1 + i];
    b_est[i] = X[4 + i];
  }
  float A6[36], rhs[6], inv[36];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) A6[i * 6 + j] = 0.0f + Wvw(i, j);
  float wc[3], wx[3];
  for (int i = 0; i < 3; ++i) wc[i] = Xvw[3 + i] - b_est[i];
  for (int i = 0; i < 3; ++i) {
    float s = 0;
    for (int j = 0; j < 3; ++j) {
      s += JtJ(4 + i, 4 + j) * wc[j];
      A6[(3 + i) * 6 + 3 + j] = JtJ(4 + i, 4 + j) + Wvw(3 + i, 3 + j);
    }
    wx[i] = s;
  }
  for (int i = 0; i < 6; ++i) {
    float s = 0;
    for (int j = 0; j < 6; ++j) s += Wvw(i, j) * Xvw[j];
    rhs[i] = s + ((i >= 3) ? wx[i - 3] : 0.0f);
  }
  rh::hm::cholesky6_inverse(A6, inv);
  for (int i = 0; i < 6; ++i) {
    float s = 0;
    for (int j = 0; j < 6; ++j) s += inv[i * 6 + j] * rhs[j];
    Xvw[i] = s;
  }
  return k;
}


}  // namespace rebvio
