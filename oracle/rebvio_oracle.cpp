/*
 * rebvio_oracle.cpp — CPU oracle (TEST INFRASTRUCTURE ONLY; see rebvio_oracle.h).
 *
 * Scalar, sequential restatement of the reference hot path with the reference's evaluation
 * order and its float/double promotion rules. Build with -ffp-contract=off (no FMA fusing) so
 * results are machine independent and bit-comparable with the HIP parity path.
 *
 * PARITY UNPINNED (no reference golden vectors exist for this path).
 *
 * Decisions for undefined behaviour in the reference, applied here AND in the
 * HIP kernels:
 *   - float->int conversions of NaN / out-of-range values follow x86 cvttss2si/cvttsd2si
 *     (result INT_MIN), which is what the reference does de facto on its only platform;
 *   - tryVel: `fi` on calculatefJ's two early-return paths (core.cpp:49-60) is the value last
 *     written by an earlier matched keyline of the same tryVel call (0 before the first);
 *   - tuneThreshold on an empty map keeps the previous auto threshold; the histogram read one
 *     past the end (edge_detector.cpp:183) is read as 0 (it cannot change the result);
 *   - TooN pieces (absent submodule, unpinned) are restated from TooN's published algorithms:
 *     dot products accumulate from 0 in index order, mixed float/double makeVector promotes to
 *     double, determinant by pivoted elimination, LDL^T Cholesky, Rodrigues SO3::exp;
 *     SVD<6>::backsub (LAPACK gesvd) is replaced by a Jacobi eigen-solve with the same 1e9
 *     condition cut -> compared with tolerance only.
 */
#include "rebvio_oracle.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr float RHO_MAX = 20.0;   // types/keyline.hpp:17
constexpr float RHO_MIN = 1e-3;   // types/keyline.hpp:18
constexpr float RHO_INIT = 1.0;   // types/keyline.hpp:19
constexpr int PLANE_FIT = 2;      // edge_detector.hpp:22
constexpr int NUM_BINS = 100;     // edge_detector.hpp:29
constexpr int MAX_IMAGE_VALUE = 765;  // edge_detector.cpp:21

// x86 cvttss2si / cvttsd2si: NaN and out-of-range give the "integer indefinite" value.
inline int cvtt(float v) {
  if (!(v >= -2147483648.0f && v < 2147483648.0f)) return INT_MIN;
  return (int)v;
}
inline int cvtt(double v) {
  if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
  return (int)v;
}

// ---- TooN-style small dense helpers (float, accumulate from 0 in index order) ----------------
struct M3 {
  float a[3][3];
};
inline M3 m3_identity() {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = (i == j) ? 1.0f : 0.0f;
  return r;
}
inline M3 m3_from(const float* p) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = p[i * 3 + j];
  return r;
}
inline void m3_to(const M3& m, float* p) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) p[i * 3 + j] = m.a[i][j];
}
inline M3 m3_T(const M3& m) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = m.a[j][i];
  return r;
}
inline M3 m3_mul(const M3& x, const M3& y) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 3; ++k) s += x.a[i][k] * y.a[k][j];
      r.a[i][j] = s;
    }
  return r;
}
inline void m3_vec(const M3& m, const float v[3], float out[3]) {
  float t[3];
  for (int i = 0; i < 3; ++i) {
    float s = 0;
    for (int k = 0; k < 3; ++k) s += m.a[i][k] * v[k];
    t[i] = s;
  }
  out[0] = t[0];
  out[1] = t[1];
  out[2] = t[2];
}
inline M3 m3_add(const M3& x, const M3& y) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][j] + y.a[i][j];
  return r;
}
inline M3 m3_sub(const M3& x, const M3& y) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][j] - y.a[i][j];
  return r;
}
inline M3 m3_scale(const M3& x, float s) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][j] * s;
  return r;
}

// TooN::determinant for N>2: Gaussian elimination with partial pivoting (TooN determinant.h).
template <int N>
float det_elim(const float* A_) {
  float A[N][N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) A[i][j] = A_[i * N + j];
  float det = 1;
  for (int i = 0; i < N; ++i) {
    int argmax = i;
    float mx = std::fabs(A[i][i]);
    for (int ii = i + 1; ii < N; ++ii)
      if (std::fabs(A[ii][i]) > mx) {
        mx = std::fabs(A[ii][i]);
        argmax = ii;
      }
    float pivot = A[argmax][i];
    if (argmax != i) {
      det *= -1;
      for (int ii = i; ii < N; ++ii) std::swap(A[i][ii], A[argmax][ii]);
    }
    det *= A[i][i];
    if (det == 0) return 0;
    for (int u = i + 1; u < N; ++u) {
      float factor = A[u][i] / pivot;
      for (int uu = i; uu < N; ++uu) A[u][uu] = A[u][uu] - factor * A[i][uu];
    }
  }
  return det;
}

// types::invert (types/definitions.hpp:40-53): adjugate / determinant.
inline M3 m3_invert(const M3& in) {
  M3 o;
  const float(*m)[3] = in.a;
  o.a[0][0] = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  o.a[0][1] = m[0][2] * m[2][1] - m[0][1] * m[2][2];
  o.a[0][2] = m[0][1] * m[1][2] - m[0][2] * m[1][1];
  o.a[1][0] = m[1][2] * m[2][0] - m[1][0] * m[2][2];
  o.a[1][1] = m[0][0] * m[2][2] - m[0][2] * m[2][0];
  o.a[1][2] = m[0][2] * m[1][0] - m[0][0] * m[1][2];
  o.a[2][0] = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  o.a[2][1] = m[0][1] * m[2][0] - m[0][0] * m[2][1];
  o.a[2][2] = m[0][0] * m[1][1] - m[0][1] * m[1][0];
  float d = det_elim<3>(&in.a[0][0]);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o.a[i][j] = o.a[i][j] / d;
  return o;
}

// TooN SO3::exp (so3.h rodrigues_so3_exp), precision float.
inline M3 so3_exp(const float w[3]) {
  const float one_6th = 1.0 / 6.0;
  const float one_20th = 1.0 / 20.0;
  float theta_sq = 0;
  for (int i = 0; i < 3; ++i) theta_sq += w[i] * w[i];
  float A, B;
  if (theta_sq < 1e-8) {
    A = 1.0 - one_6th * theta_sq;
    B = 0.5;
  } else if (theta_sq < 1e-6) {
    B = 0.5 - 0.25 * one_6th * theta_sq;
    A = 1.0 - theta_sq * one_6th * (1.0 - one_20th * theta_sq);
  } else {
    const float theta = std::sqrt(theta_sq);
    const float inv_theta = 1.0 / theta;
    A = std::sin(theta) * inv_theta;
    B = (1 - std::cos(theta)) * (inv_theta * inv_theta);
  }
  M3 R;
  {
    const float wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
    R.a[0][0] = 1.0 - B * (wy2 + wz2);
    R.a[1][1] = 1.0 - B * (wx2 + wz2);
    R.a[2][2] = 1.0 - B * (wx2 + wy2);
  }
  {
    const float a = A * w[2], b = B * (w[0] * w[1]);
    R.a[0][1] = b - a;
    R.a[1][0] = b + a;
  }
  {
    const float a = A * w[1], b = B * (w[0] * w[2]);
    R.a[0][2] = b + a;
    R.a[2][0] = b - a;
  }
  {
    const float a = A * w[0], b = B * (w[1] * w[2]);
    R.a[1][2] = b - a;
    R.a[2][1] = b + a;
  }
  return R;
}

// TooN Cholesky<N> (LDL^T, Cholesky.h) get_inverse().
template <int N>
void cholesky_inverse(const float* A, float* inv) {
  float L[N][N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) L[i][j] = A[i * N + j];
  for (int col = 0; col < N; ++col) {
    float inv_diag = 1;
    for (int row = col; row < N; ++row) {
      float val = L[row][col];
      for (int col2 = 0; col2 < col; ++col2) val -= L[col2][col] * L[row][col2];
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
    for (int i = 0; i < N; ++i) inv[i * N + c] = res[i];
  }
}

// Replacement for TooN::SVD<6,6,float>::backsub on a symmetric matrix: cyclic Jacobi in double,
// singular values below max/1e9 dropped (TooN condition_no).
template <int N>
void sym_pinv_solve(const float* A_, const float* b_, float* x_) {
  double A[N][N], V[N][N];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      A[i][j] = 0.5 * ((double)A_[i * N + j] + (double)A_[j * N + i]);
      V[i][j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) off += A[p][q] * A[p][q];
    if (off < 1e-300) break;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        if (A[p][q] == 0.0) continue;
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = ((theta >= 0) ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < N; ++k) {
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = cs * akp - sn * akq;
          A[k][q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < N; ++k) {
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = cs * apk - sn * aqk;
          A[q][k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < N; ++k) {
          double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = cs * vkp - sn * vkq;
          V[k][q] = sn * vkp + cs * vkq;
        }
      }
  }
  double dmax = 0;
  for (int i = 0; i < N; ++i) dmax = std::max(dmax, std::fabs(A[i][i]));
  double x[N];
  for (int i = 0; i < N; ++i) x[i] = 0;
  for (int k = 0; k < N; ++k) {
    double lam = A[k][k];
    if (!(std::fabs(lam) * 1e9 > dmax)) continue;
    double proj = 0;
    for (int i = 0; i < N; ++i) proj += V[i][k] * (double)b_[i];
    proj /= lam;
    for (int i = 0; i < N; ++i) x[i] += V[i][k] * proj;
  }
  bool bad = false;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j)
      if (std::isnan(A_[i * N + j])) bad = true;
  for (int i = 0; i < N; ++i) x_[i] = bad ? std::numeric_limits<float>::quiet_NaN() : (float)x[i];
}


// ---- generic small dense helpers for the inertial glue (float, TooN evaluation order) -----------------------------
template <int R, int C>
struct Mat {
  float a[R][C];
  static Mat zeros() {
    Mat m;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) m.a[i][j] = 0;
    return m;
  }
};
template <int R, int K, int C>
Mat<R, C> mmul(const Mat<R, K>& x, const Mat<K, C>& y) {
  Mat<R, C> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      float s = 0;
      for (int k = 0; k < K; ++k) s += x.a[i][k] * y.a[k][j];
      r.a[i][j] = s;
    }
  return r;
}
template <int R, int C>
Mat<C, R> mT(const Mat<R, C>& x) {
  Mat<C, R> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) r.a[j][i] = x.a[i][j];
  return r;
}
template <int R, int C>
void mvec(const Mat<R, C>& m, const float* v, float* out) {
  float t[R];
  for (int i = 0; i < R; ++i) {
    float s = 0;
    for (int k = 0; k < C; ++k) s += m.a[i][k] * v[k];
    t[i] = s;
  }
  for (int i = 0; i < R; ++i) out[i] = t[i];
}
template <int N>
float vdot(const float* a, const float* b) {
  float s = 0;
  for (int i = 0; i < N; ++i) s += a[i] * b[i];
  return s;
}
// x^T M y
template <int N>
float quad(const float* x, const Mat<N, N>& M, const float* y) {
  float t[N];
  for (int j = 0; j < N; ++j) {
    float s = 0;
    for (int k = 0; k < N; ++k) s += x[k] * M.a[k][j];
    t[j] = s;
  }
  return vdot<N>(t, y);
}
template <int N>
Mat<N, N> chol_inv(const Mat<N, N>& A) {
  Mat<N, N> r;
  cholesky_inverse<N>(&A.a[0][0], &r.a[0][0]);
  return r;
}

// TooN SVD<N>(A).backsub(b) in DOUBLE precision (SVD<7> without a precision argument, sab_estimator.cpp:31) for a
// symmetric A: Jacobi eigen-decomposition, singular values below max/1e9 dropped.
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
        double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        double t = ((theta >= 0) ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < N; ++k) {
          double akp = A[k][p], akq = A[k][q];
          A[k][p] = cs * akp - sn * akq;
          A[k][q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < N; ++k) {
          double apk = A[p][k], aqk = A[q][k];
          A[p][k] = cs * apk - sn * aqk;
          A[q][k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < N; ++k) {
          double vkp = V[k][p], vkq = V[k][q];
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
    double lam = A[k][k];
    if (!(std::fabs(lam) * 1e9 > dmax)) continue;
    double proj = 0;
    for (int i = 0; i < N; ++i) proj += V[i][k] * b_[i];
    proj /= lam;
    for (int i = 0; i < N; ++i) x_[i] += V[i][k] * proj;
  }
}

// TooN SO3(a, b): minimal rotation taking a to b (so3.h two-vector constructor)
inline M3 so3_from_two(const float a[3], const float b[3]) {
  auto cross = [](const float* x, const float* y, float* o) {
    o[0] = x[1] * y[2] - x[2] * y[1];
    o[1] = x[2] * y[0] - x[0] * y[2];
    o[2] = x[0] * y[1] - x[1] * y[0];
  };
  auto unit = [](const float* x, float* o) {
    float n = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    for (int i = 0; i < 3; ++i) o[i] = x[i] / n;
  };
  float n[3];
  cross(a, b, n);
  if (n[0] * n[0] + n[1] * n[1] + n[2] * n[2] == 0) return m3_identity();
  unit(n, n);
  M3 R1, Rm;
  float ua[3], ub[3], c1[3], c2[3];
  unit(a, ua);
  unit(b, ub);
  cross(n, ua, c1);
  cross(n, ub, c2);
  for (int i = 0; i < 3; ++i) {  // columns: unit(a) | n | n ^ unit(a)
    R1.a[i][0] = ua[i]; R1.a[i][1] = n[i]; R1.a[i][2] = c1[i];
    Rm.a[i][0] = ub[i]; Rm.a[i][1] = n[i]; Rm.a[i][2] = c2[i];
  }
  return m3_mul(Rm, m3_T(R1));
}

// TooN SO3::ln()
inline void so3_ln(const M3& R, float out[3]) {
  const float cos_angle = (R.a[0][0] + R.a[1][1] + R.a[2][2] - 1.0f) * 0.5f;
  float r[3] = {(R.a[2][1] - R.a[1][2]) / 2, (R.a[0][2] - R.a[2][0]) / 2, (R.a[1][0] - R.a[0][1]) / 2};
  float sin_abs = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  if (cos_angle > (float)M_SQRT1_2) {
    if (sin_abs > 0) {
      float k = std::asin(sin_abs) / sin_abs;
      for (int i = 0; i < 3; ++i) r[i] *= k;
    }
  } else if (cos_angle > -(float)M_SQRT1_2) {
    if (sin_abs > 0) {
      float k = std::acos(cos_angle) / sin_abs;
      for (int i = 0; i < 3; ++i) r[i] *= k;
    }
  } else {
    const float angle = (float)M_PI - std::asin(sin_abs);
    const float d0 = R.a[0][0] - cos_angle, d1 = R.a[1][1] - cos_angle, d2 = R.a[2][2] - cos_angle;
    float r2[3];
    if (d0 * d0 > d1 * d1 && d0 * d0 > d2 * d2) {
      r2[0] = d0; r2[1] = (R.a[1][0] + R.a[0][1]) / 2; r2[2] = (R.a[0][2] + R.a[2][0]) / 2;
    } else if (d1 * d1 > d2 * d2) {
      r2[0] = (R.a[1][0] + R.a[0][1]) / 2; r2[1] = d1; r2[2] = (R.a[2][1] + R.a[1][2]) / 2;
    } else {
      r2[0] = (R.a[0][2] + R.a[2][0]) / 2; r2[1] = (R.a[2][1] + R.a[1][2]) / 2; r2[2] = d2;
    }
    if (r2[0] * r[0] + r2[1] * r[1] + r2[2] * r[2] < 0)
      for (int i = 0; i < 3; ++i) r2[i] = -r2[i];
    float nrm = std::sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
    for (int i = 0; i < 3; ++i) r[i] = r2[i] * (angle / nrm);
  }
  for (int i = 0; i < 3; ++i) out[i] = r[i];
}

// types::IntegratedImu (types/imu.hpp:35-151)
struct IntImu {
  unsigned n = 0;
  uint64_t last_ts = 0, init_ts = 0, dt = 0;
  M3 R = m3_identity();
  float gyro[3] = {0, 0, 0}, gyro_init[3] = {0, 0, 0}, gyro_last[3] = {0, 0, 0}, acc[3] = {0, 0, 0}, dgyro[3] = {0, 0, 0},
        cacc[3] = {0, 0, 0};
  float dt_s() const { return float(dt) / 1000000.0; }
  void add(uint64_t ts, const float g[3], const float a[3], const M3& Rc2i) {
    float tmp[3], at[3];
    M3 RT = m3_T(Rc2i);
    m3_vec(RT, g, tmp);
    float dts;
    if (last_ts == 0) {
      n = 1;
      init_ts = ts;
      last_ts = init_ts;
      dts = 0.005;
      for (int i = 0; i < 3; ++i) gyro_init[i] = gyro_last[i] = g[i];
      R = m3_identity();
      for (int i = 0; i < 3; ++i) gyro[i] = tmp[i];
      m3_vec(RT, a, acc);
      for (int i = 0; i < 3; ++i) dgyro[i] = cacc[i] = 0;
    } else {
      ++n;
      dts = float(ts - last_ts) / 1000000.0;
      m3_vec(RT, a, at);
      for (int i = 0; i < 3; ++i) {
        gyro[i] += tmp[i];
        acc[i] += at[i];
      }
    }
    float w[3] = {tmp[0] * dts, tmp[1] * dts, tmp[2] * dts};
    R = m3_mul(R, so3_exp(w));
    last_ts = ts;
    for (int i = 0; i < 3; ++i) gyro_last[i] = g[i];
  }
  void get(const M3& Rc2i, const float tc2i[3]) {
    dt = (n == 1) ? 0 : (last_ts - init_ts) / (uint64_t)(unsigned)(n - 1) * n;  // n == 1: the reference divides 0 by 0
    M3 RT = m3_T(Rc2i);
    if (n > 1) {
      for (int i = 0; i < 3; ++i) {
        gyro[i] /= float(n);
        acc[i] /= float(n);
      }
      float d[3] = {gyro_last[0] - gyro_init[0], gyro_last[1] - gyro_init[1], gyro_last[2] - gyro_init[2]}, t[3];
      m3_vec(RT, d, t);
      for (int i = 0; i < 3; ++i) dgyro[i] = t[i] / dt_s();
    }
    float rt[3];
    m3_vec(RT, tc2i, rt);
    for (int i = 0; i < 3; ++i) rt[i] = -rt[i];
    cacc[0] = acc[0] + (dgyro[1] * rt[2] - dgyro[2] * rt[1]);
    cacc[1] = acc[1] + (dgyro[2] * rt[0] - dgyro[0] * rt[2]);
    cacc[2] = acc[2] + (dgyro[0] * rt[1] - dgyro[1] * rt[0]);
    n = 0;
    init_ts = 0;
    last_ts = 0;
  }
};

// SABEstimator::problem (sab_estimator.cpp:41-165): weighted least squares on the 11-vector F
struct SabCfg {
  float a_v[3], a_s[3], G, x_p[7], Rg;
  M3 Rv, Rs;
  Mat<7, 7> Pp;
};
inline void sab_problem(const SabCfg& cfg, Mat<7, 7>& JtJ, float JtF[7], const float X[7]) {
  const float a = X[0];
  const float g[3] = {X[1], X[2], X[3]}, b[3] = {X[4], X[5], X[6]};
  float F[11];
  for (int i = 0; i < 11; ++i) F[i] = 0;
  const float ca = std::cos(a), sa = std::sin(a);
  for (int i = 0; i < 3; ++i) F[i] = (cfg.a_s[i] + g[i]) * ca - cfg.a_v[i] * sa;
  F[3] = vdot<3>(g, g) - cfg.G * cfg.G;
  F[4] = X[0] - cfg.x_p[0];
  if (F[4] > M_PI) F[4] -= 2.0 * M_PI;
  else if (F[4] < -M_PI) F[4] += 2.0 * M_PI;
  const M3 Rb = so3_exp(b);
  float Rg3[3];
  m3_vec(Rb, g, Rg3);
  for (int i = 0; i < 3; ++i) F[5 + i] = Rg3[i] - cfg.x_p[1 + i];
  for (int i = 0; i < 3; ++i) F[8 + i] = b[i] - cfg.x_p[4 + i];
  float dFda[11];
  for (int i = 0; i < 11; ++i) dFda[i] = 0;
  for (int i = 0; i < 3; ++i) dFda[i] = -(cfg.a_s[i] + g[i]) * sa - cfg.a_v[i] * ca;
  dFda[4] = 1.0;
  Mat<11, 6> dFdx1 = Mat<11, 6>::zeros();
  const float Gx[3][3] = {{0.0f, Rg3[2], -Rg3[1]}, {-Rg3[2], 0.0f, Rg3[0]}, {Rg3[1], -Rg3[0], 0.0f}};
  for (int i = 0; i < 3; ++i) dFdx1.a[i][i] = ca;
  for (int j = 0; j < 3; ++j) dFdx1.a[3][j] = 2.0 * g[j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      dFdx1.a[5 + i][j] = Rb.a[i][j];
      dFdx1.a[5 + i][3 + j] = Gx[i][j];
    }
  for (int i = 0; i < 3; ++i) dFdx1.a[8 + i][3 + i] = 1.0f;
  Mat<11, 11> P = Mat<11, 11>::zeros(), W = Mat<11, 11>::zeros(), dPda = Mat<11, 11>::zeros();
  Mat<3, 3> Pz;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Pz.a[i][j] = sa * sa * cfg.Rv.a[i][j] + ca * ca * cfg.Rs.a[i][j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) P.a[i][j] = Pz.a[i][j];
  P.a[3][3] = cfg.Rg;
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) P.a[4 + i][4 + j] = cfg.Pp.a[i][j];
  const Mat<3, 3> Wz = chol_inv<3>(Pz);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) W.a[i][j] = Wz.a[i][j];
  W.a[3][3] = 1.0 / cfg.Rg;
  const Mat<7, 7> Wp = chol_inv<7>(cfg.Pp);
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) W.a[4 + i][4 + j] = Wp.a[i][j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) dPda.a[i][j] = 2.0 * sa * ca * (cfg.Rv.a[i][j] - cfg.Rs.a[i][j]);
  Mat<11, 11> dWda = mmul(mmul(W, dPda), W);
  for (int i = 0; i < 11; ++i)
    for (int j = 0; j < 11; ++j) dWda.a[i][j] = -dWda.a[i][j];
  // _JtJ(0,0) = 0.25*F*dWda*P*dWda*F + dFda*dWda*F + dFda*W*dFda
  {
    const Mat<11, 11> M = mmul(mmul(dWda, P), dWda);
    JtJ.a[0][0] = 0.25 * quad<11>(F, M, F) + quad<11>(dFda, dWda, F) + quad<11>(dFda, W, dFda);
  }
  const Mat<6, 11> dT = mT(dFdx1);
  {
    float t1[11], t2[11], c1[6], c2[6];
    mvec(dWda, F, t1);
    mvec(W, dFda, t2);
    mvec(dT, t1, c1);
    mvec(dT, t2, c2);
    for (int i = 0; i < 6; ++i) {
      JtJ.a[1 + i][0] = 0.5 * c1[i] + c2[i];
      JtJ.a[0][1 + i] = JtJ.a[1 + i][0];
    }
  }
  {
    const Mat<6, 6> B = mmul(mmul(dT, W), dFdx1);
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) JtJ.a[1 + i][1 + j] = B.a[i][j];
  }
  JtF[0] = 0.5 * quad<11>(F, dWda, F) + quad<11>(dFda, W, F);
  {
    float t[11], c[6];
    mvec(W, F, t);
    mvec(dT, t, c);
    for (int i = 0; i < 6; ++i) JtF[1 + i] = c[i];
  }
}

inline float sab_saturate(float t, float limit) { return (t > limit) ? limit : ((t < -limit) ? -limit : t); }

// SABEstimator::gaussNewton (sab_estimator.cpp:21-39), SVD<7> in double
inline int sab_gauss_newton(const SabCfg& cfg, float X[7], int iter_max) {
  int i = 0;
  for (; i < iter_max; ++i) {
    Mat<7, 7> JtJ;
    float JtF[7];
    sab_problem(cfg, JtJ, JtF, X);
    double A[49], b[7], h[7];
    for (int r = 0; r < 7; ++r) {
      b[r] = -(double)JtF[r];
      for (int c2 = 0; c2 < 7; ++c2) A[r * 7 + c2] = (double)JtJ.a[r][c2];
    }
    sym_pinv_solve_d<7>(A, b, h);
    for (int r = 0; r < 7; ++r) X[r] = (float)((double)X[r] + h[r]);  // _X += h with h a double vector
    X[0] = std::atan2(std::sin(X[0]), std::cos(X[0]));
    for (int r = 4; r < 7; ++r) X[r] = sab_saturate(X[r], 5e-1 / 25);
    double nh = 0;
    for (int r = 0; r < 7; ++r) nh += h[r] * h[r];
    nh = std::sqrt(nh);
    float nx = 0;
    for (int r = 0; r < 7; ++r) nx += X[r] * X[r];
    if (nh < 0.0 || nh / (std::sqrt(nx) + 1e-20) < 0.0) break;  // tolerances default to 0: never breaks early
  }
  return i;
}

// ---- scale space (scale_space.cpp) -------------------------------------------------------------
struct BoxGaussian {
  int n = 3;
  float sigma = 0, sigma_true = 0;
  std::vector<int> widths;                   // [n] (the reference allocates new int[n_], scale_space.cpp:17)
  std::vector<std::vector<float>> divisors;  // [n]
};

// FastGaussian::precomputeDivisors (scale_space.cpp:130-171)
void make_divisors(int rows, int cols, int d, std::vector<float>& div) {
  div.assign((size_t)rows * cols, 0.f);
  const int d2 = d / 2;
  const float a = (float)d * d;
  auto at = [&](int r, int c) -> float& { return div[(size_t)r * cols + c]; };
  for (int row = 0; row < d2 + 1; ++row) {
    for (int col = 0; col < d2 + 1; ++col) at(row, col) = (col + d2 + 1) * (row + d2 + 1);
    for (int col = d2 + 1; col < cols - d2; ++col) at(row, col) = (float)d * (row + d2 + 1);
    for (int col = cols - d2; col < cols; ++col) at(row, col) = (cols - col + d2) * (row + d2 + 1);
  }
  for (int row = d2 + 1; row < rows - d2; ++row) {
    for (int col = 0; col < d2 + 1; ++col) at(row, col) = (col + d2 + 1) * (float)d;
    for (int col = d2 + 1; col < cols - d2; ++col) at(row, col) = a;
    for (int col = cols - d2; col < cols; ++col) at(row, col) = (cols - col + d2) * (float)d;
  }
  for (int row = rows - d2; row < rows; ++row) {
    for (int col = 0; col < d2 + 1; ++col) at(row, col) = (rows - row + d2) * (col + d2 + 1);
    for (int col = d2 + 1; col < cols - d2; ++col) at(row, col) = (rows - row + d2) * (float)d;
    for (int col = cols - d2; col < cols; ++col) at(row, col) = (rows - row + d2) * (cols - col + d2);
  }
  for (size_t i = 0; i < div.size(); ++i) div[i] = 1.0 / div[i];
}

// FastGaussian::FastGaussian (scale_space.cpp:14-41)
void make_filter(BoxGaussian& f, int rows, int cols, float sigma, int n) {
  f.n = n;
  f.sigma = sigma;
  f.widths.assign((size_t)n, 0);
  f.divisors.assign((size_t)n, std::vector<float>());
  float w_ideal = sqrt(12.0 * sigma * sigma / float(n + 1));
  int w_l = int(w_ideal);
  if (int(w_l / 2) * 2 == w_l) --w_l;
  int m = std::round((3 * n + 4 * n * w_l + n * w_l * w_l - 12 * sigma * sigma) / (4 + 4 * w_l));
  int i;
  for (i = 0; i < m; i++) f.widths[i] = w_l;
  for (; i < n; i++) f.widths[i] = w_l + 2;
  f.sigma_true = sqrt((m * w_l * w_l + (n - m) * (w_l + 2.0) * (w_l + 2.0) - n) / 12.0);
  for (int k = 0; k < n; ++k) make_divisors(rows, cols, f.widths[k], f.divisors[k]);
}

// FastGaussian::createIntegralImage (scale_space.cpp:48-67)
void integral_image(int rows, int cols, const float* in, float* ii) {
  for (int row = 0; row < rows; ++row) {
    float* o = ii + (size_t)row * cols;
    const float* p = in + (size_t)row * cols;
    o[0] = p[0];
    for (int col = 1; col < cols; ++col) o[col] = o[col - 1] + p[col];
  }
  for (int row = 1; row < rows; ++row) {
    const float* prev = ii + (size_t)(row - 1) * cols;
    float* o = ii + (size_t)row * cols;
    for (int col = 0; col < cols; ++col) o[col] += prev[col];
  }
}

// FastGaussian::average (scale_space.cpp:69-128): nine regions, operand orders kept.
void box_average(int rows, int cols, int d, const float* ii, const float* div, float* out) {
  const int d2 = d / 2;
  const float a = 1.0 / (d * d);
  auto II = [&](int r) { return ii + (size_t)r * cols; };
  for (int row = 0; row < d2 + 1; ++row) {
    const float* dv = div + (size_t)row * cols;
    const float* p = II(row + d2);
    float* o = out + (size_t)row * cols;
    for (int col = 0; col < d2 + 1; ++col) o[col] = p[col + d2] * dv[col];
    for (int col = d2 + 1; col < cols - d2; ++col) o[col] = (p[col + d2] - p[col - d2 - 1]) * dv[col];
    for (int col = cols - d2; col < cols; ++col) o[col] = (p[cols - 1] - p[col - d2 - 1]) * dv[col];
  }
  for (int row = d2 + 1; row < rows - d2; ++row) {
    const float* dv = div + (size_t)row * cols;
    const float* p1 = II(row + d2);
    const float* p2 = II(row - d2 - 1);
    float* o = out + (size_t)row * cols;
    for (int col = 0; col < d2 + 1; ++col) o[col] = (p1[col + d2] - p2[col + d2]) * dv[col];
    for (int col = d2 + 1; col < cols - d2; ++col) {
      int c1 = col + d2, c2 = col - d2 - 1;
      o[col] = (p1[c1] - p1[c2] - p2[c1] + p2[c2]) * a;
    }
    for (int col = cols - d2; col < cols; ++col) {
      int c1 = cols - 1, c2 = col - d2 - 1;
      o[col] = (p1[c1] - p1[c2] - p2[c1] + p2[c2]) * dv[col];
    }
  }
  for (int row = rows - d2; row < rows; ++row) {
    const float* dv = div + (size_t)row * cols;
    const float* p1 = II(rows - 1);
    const float* p2 = II(row - d2 - 1);
    float* o = out + (size_t)row * cols;
    for (int col = 0; col < d2 + 1; ++col) o[col] = (p1[col + d2] - p2[col + d2]) * dv[col];
    for (int col = d2 + 1; col < cols - d2; ++col) {
      int c1 = col + d2, c2 = col - d2 - 1;
      o[col] = (p1[c1] - p2[c1] - p1[c2] + p2[c2]) * dv[col];
    }
    for (int col = cols - d2; col < cols; ++col) {
      int c1 = cols - 1, c2 = col - d2 - 1;
      o[col] = (p1[c1] - p2[c1] - p1[c2] + p2[c2]) * dv[col];
    }
  }
}

}  // namespace

// ---- containers ---------------------------------------------------------------------------------
struct orc_map {
  IntImu imu;
  int rows = 0, cols = 0;
  uint64_t ts_us = 0;
  float threshold = -1.0f;  // edge_map.cpp:17
  unsigned matches = 0;
  std::vector<orc_keyline> kl;
  std::vector<int> mask;  // dense image-index -> keyline-index (the reference keeps a hash map, edge_map.hpp:131)
};

struct orc_ctx {
  orc_params p;
  BoxGaussian filter[2];
  std::vector<float> scale[2], dog, mag, tmp_a, tmp_b;
  // EdgeDetector state (edge_detector.hpp:84-92)
  float det_threshold;  // config_->threshold (servo state)
  int keylines_count = 0;
  float auto_threshold;
  std::vector<int> det_mask;
  // DistanceField (core.hpp:73-78)
  std::vector<int> df_id, df_dist;
  orc_map* df_map = nullptr;
  unsigned frame_count = 0;
  // DIAGNOSTIC, not the reference: 1 = the keyline sums of tryVel / extRotVel are accumulated in double (terms still
  // fp32). Used by tools/tolerance_probe.py to measure how much of the reference's own result is summation rounding.
  // 2 = the fp32 terms added in the HIP kernels' order (struct Acc).
  int wide_sums = 0;
  // glue state (imu.hpp:171-187)
  float Bg[3];
  M3 W_Bg, RGBias, RGyro;
  // full VIO glue state (rebvio.cpp:95-117, types/imu.hpp ImuState, sab_estimator.hpp State)
  M3 R_c2i;
  float t_c2i[3];
  unsigned num_frames;
  int initialized, num_gyro_init;
  float gyro_init[3], g_init[3];
  float Kscale, P_Kp;
  float Pos[3], u_est[3];
  M3 R_global;
  float sabX[7], g_est[3], b_est[3];
  Mat<7, 7> sabP;
  M3 Qrot, Qg, Qbias, Rs, Rv;
  float QKp, Rg_sab;
  float Av[3], As[3];
  float meanA[4][3];
  // estimateLs4Acceleration statics (core.cpp:287-293), per instance here
  float ls4_V[3], ls4_V0[3], ls4_V1[3], ls4_V2[3], ls4_V3[3], ls4_T[5], ls4_Dt[4];
  // wall seconds accumulated at the reference's REBVIO_TIMER tick sites (orc_stage_seconds): 0 detect
  // (edge_detector.cpp:31,41), 1 buildDistanceField (core.cpp:34-36), 2 minimizeVel (core.cpp:152,187), 3 extRotVel
  // (core.cpp:193,258), 4 directedMatch (edge_map.cpp:189,216), 5 everything else of the pair step (untimed in the reference)
  double stage_acc[6] = {0, 0, 0, 0, 0, 0};
};

namespace {
struct StageTimer {
  double* acc;
  std::chrono::steady_clock::time_point t0;
  explicit StageTimer(double* a) : acc(a), t0(std::chrono::steady_clock::now()) {}
  ~StageTimer() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

namespace {

void smooth(orc_ctx* c, BoxGaussian& f, const float* in, std::vector<float>& out) {
  // FastGaussian::smooth (scale_space.cpp:173-182)
  const int R = c->p.rows, C = c->p.cols;
  c->tmp_a.resize((size_t)R * C);
  c->tmp_b.resize((size_t)R * C);
  out.resize((size_t)R * C);
  integral_image(R, C, in, c->tmp_a.data());
  for (int i = 0; i < f.n - 1; ++i) {
    box_average(R, C, f.widths[i], c->tmp_a.data(), f.divisors[i].data(), c->tmp_b.data());
    integral_image(R, C, c->tmp_b.data(), c->tmp_a.data());
  }
  box_average(R, C, f.widths[f.n - 1], c->tmp_a.data(), f.divisors[f.n - 1].data(), out.data());
}

void scale_space_build(orc_ctx* c, const float* img) {
  // ScaleSpace::build (scale_space.cpp:203-233)
  const int R = c->p.rows, C = c->p.cols;
  smooth(c, c->filter[0], img, c->scale[0]);
  smooth(c, c->filter[1], img, c->scale[1]);
  for (int row = 0; row < R; ++row)
    for (int col = 0; col < C; ++col) {
      size_t i = (size_t)row * C + col;
      c->dog[i] = c->scale[1][i] - c->scale[0][i];
    }
  for (int row = 1; row < R - 1; ++row) {
    const float* ic = &c->scale[0][(size_t)row * C];
    const float* il = &c->scale[0][(size_t)(row - 1) * C];
    const float* iu = &c->scale[0][(size_t)(row + 1) * C];
    float* m = &c->mag[(size_t)row * C];
    for (int col = 1; col < C - 1; ++col) {
      float dx = ic[col + 1] - ic[col - 1];
      float dy = iu[col] - il[col];
      m[col] = dx * dx + dy * dy;
    }
  }
}

inline orc_keyline make_keyline(float px, float py, float gx, float gy, float ix, float iy) {
  // KeyLine ctor (types/keyline.hpp:42-59)
  orc_keyline k;
  k.pos[0] = px; k.pos[1] = py;
  k.pos_img[0] = ix; k.pos_img[1] = iy;
  k.match_pos_img[0] = ix; k.match_pos_img[1] = iy;
  k.gradient[0] = gx; k.gradient[1] = gy;
  k.match_gradient[0] = 0; k.match_gradient[1] = 0;
  k.gradient_norm = std::sqrt(gx * gx + gy * gy);
  k.match_gradient_norm = 0.0;
  k.rho = 1.0;
  k.sigma_rho = 20.0;
  k.id = -1; k.id_prev = -1; k.id_next = -1;
  k.match_id = -1; k.match_id_forward = -1; k.match_id_keyframe = -1;
  k.matches = 0;
  return k;
}

// EdgeDetector::buildEdgeMap (edge_detector.cpp:45-123)
orc_map* build_edge_map(orc_ctx* c, const float* img, uint64_t ts) {
  const orc_params& P = c->p;
  const int R = P.rows, C = P.cols;
  scale_space_build(c, img);
  orc_map* map = new orc_map;
  map->rows = R; map->cols = C; map->ts_us = ts;
  map->kl.reserve(P.keylines_max);
  c->keylines_count = 0;

  // Pinv = invert(Phi^T Phi) * Phi^T (edge_detector.cpp:55-68)
  static_assert(PLANE_FIT == 2, "window is 5x5");
  float Phi[25][3];
  for (int row = -PLANE_FIT, k = 0; row <= PLANE_FIT; ++row)
    for (int col = -PLANE_FIT; col <= PLANE_FIT; ++col, ++k) {
      Phi[k][0] = col; Phi[k][1] = row; Phi[k][2] = 1;
    }
  M3 PtP;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 25; ++k) s += Phi[k][i] * Phi[k][j];
      PtP.a[i][j] = s;
    }
  M3 inv = m3_invert(PtP);
  float Pinv[3][25];
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 25; ++k) {
      float s = 0;
      for (int j = 0; j < 3; ++j) s += inv.a[i][j] * Phi[k][j];
      Pinv[i][k] = s;
    }

  float pn_threshold = float((2.0 * PLANE_FIT + 1.0) * (2.0 * PLANE_FIT + 1.0)) * P.pos_neg_threshold;
  float thr = c->det_threshold;
  float gradient_threshold_squared = (thr * MAX_IMAGE_VALUE * P.dog_threshold) * (thr * MAX_IMAGE_VALUE * P.dog_threshold);
  float mag_threshold = (thr * MAX_IMAGE_VALUE) * (thr * MAX_IMAGE_VALUE);

  int* mask = c->det_mask.data();
  for (int row = PLANE_FIT; row < R - PLANE_FIT; ++row) {
    int* km = mask + (size_t)row * C;
    const float* mg = &c->mag[(size_t)row * C];
    for (int col = PLANE_FIT; col < C - PLANE_FIT; ++col) {
      int idx = col + row * C;
      km[col] = -1;
      if (mg[col] < mag_threshold) continue;

      int pn = 0;
      float Y[25];
      for (int r = -PLANE_FIT, k = 0; r <= PLANE_FIT; ++r) {
        const float* dp = &c->dog[(size_t)(row + r) * C];
        for (int cc = -PLANE_FIT; cc <= PLANE_FIT; ++cc, ++k) {
          float dog = dp[col + cc];
          Y[k] = dog;
          pn = (dog > 0.0) ? pn + 1 : pn - 1;
        }
      }
      if (fabs((double)pn) > pn_threshold) continue;

      float theta[3];
      for (int i = 0; i < 3; ++i) {
        float s = 0;
        for (int k = 0; k < 25; ++k) s += Pinv[i][k] * Y[k];
        theta[i] = s;
      }
      float tmp = theta[2] / (theta[0] * theta[0] + theta[1] * theta[1]);
      float xs = -theta[0] * tmp;
      float ys = -theta[1] * tmp;
      if (fabs((double)xs) > 0.5 || fabs((double)ys) > 0.5) continue;
      if (theta[0] * theta[0] + theta[1] * theta[1] < gradient_threshold_squared) continue;

      float px = float(col) + xs, py = float(row) + ys;
      map->kl.push_back(make_keyline(px, py, theta[0], theta[1], px - P.cx, py - P.cy));
      km[col] = c->keylines_count;
      if (++c->keylines_count >= P.keylines_max) {
        int boundary = R * C;
        for (++idx; idx < boundary; ++idx) mask[idx] = -1;
        return map;
      }
    }
  }
  return map;
}

// EdgeDetector::nextKeylineIdx (edge_detector.cpp:138-165)
int next_keyline_idx(const orc_ctx* c, const orc_map* map, int x, int y, int idx) {
  const int C = c->p.cols;
  const int* mask = c->det_mask.data();
  auto M = [&](int yy, int xx) { return mask[(size_t)yy * C + xx]; };
  float tx = -map->kl[idx].gradient[1];
  float ty = map->kl[idx].gradient[0];
  int i;
  if (ty > 0.0) {
    if (tx > 0.0) {
      if ((i = M(y, x + 1)) >= 0) return i;
      if ((i = M(y + 1, x)) >= 0) return i;
      if ((i = M(y + 1, x + 1)) >= 0) return i;
    } else {
      if ((i = M(y, x - 1)) >= 0) return i;
      if ((i = M(y + 1, x)) >= 0) return i;
      if ((i = M(y + 1, x - 1)) >= 0) return i;
    }
  } else {
    if (tx < 0.0) {
      if ((i = M(y, x - 1)) >= 0) return i;
      if ((i = M(y - 1, x)) >= 0) return i;
      if ((i = M(y - 1, x - 1)) >= 0) return i;
    } else {
      if ((i = M(y, x + 1)) >= 0) return i;
      if ((i = M(y - 1, x)) >= 0) return i;
      if ((i = M(y - 1, x + 1)) >= 0) return i;
    }
  }
  return -1;
}

// EdgeDetector::joinEdges (edge_detector.cpp:125-136)
void join_edges(orc_ctx* c, orc_map* map) {
  for (int idx = 0; idx < (int)map->kl.size(); ++idx) {
    orc_keyline& k = map->kl[idx];
    int x = cvtt(k.pos[0] + 0.5);
    int y = cvtt(k.pos[1] + 0.5);
    int id_next = next_keyline_idx(c, map, x, y, idx);
    if (id_next < 0) continue;
    map->kl[id_next].id_prev = idx;
    k.id_next = id_next;
  }
}

// EdgeDetector::tuneThreshold (edge_detector.cpp:167-186)
void tune_threshold(orc_ctx* c, orc_map* map) {
  const int n = (int)map->kl.size();
  if (n == 0) {  // reference dereferences element 0 of an empty vector: keep previous value
    map->threshold = c->auto_threshold;
    return;
  }
  float max_dog = map->kl[0].gradient_norm;
  float min_dog = max_dog;
  for (int idx = 1; idx < n; ++idx) {
    const orc_keyline& k = map->kl[idx];
    if (max_dog < k.gradient_norm) max_dog = k.gradient_norm;
    if (min_dog > k.gradient_norm) min_dog = k.gradient_norm;
  }
  int histogram[NUM_BINS + 1] = {0};  // +1: the reference reads one past the end
  for (int idx = 0; idx < n; ++idx) {
    int i = cvtt(NUM_BINS * (max_dog - map->kl[idx].gradient_norm) / (max_dog - min_dog));
    i = (i > NUM_BINS - 1) ? NUM_BINS - 1 : i;
    i = (i < 0) ? 0 : i;
    ++histogram[i];
  }
  int i = 0;
  for (int a = 0; i < NUM_BINS && a < c->p.keylines_max; i++, a += histogram[i]);
  c->auto_threshold = max_dog - float(i * (max_dog - min_dog)) / float(NUM_BINS);
  map->threshold = c->auto_threshold;
}

inline int get_index(const orc_params& P, float frow, float fcol) {
  // EdgeMap::getIndex / DistanceField::getIndex (edge_map.hpp:119-124, core.hpp:66-71);
  // rows_/cols_ are unsigned there, so negative ints fail the first comparison.
  int row = cvtt(std::round(frow));
  int col = cvtt(std::round(fcol));
  if ((unsigned)row >= (unsigned)P.rows || row < 0 || (unsigned)col >= (unsigned)P.cols || col < 0) return -1;
  return row * P.cols + col;
}

// EdgeMap::searchMatch (edge_map.cpp:101-184) on `old_map`, for keyline `kq` of the other map.
int search_match(const orc_ctx* c, const orc_map* old_map, const orc_keyline& kq, const float vel[3],
                 const M3& Rvel, const M3& Rback, float max_radius) {
  const orc_params& P = c->p;
  const float cang_min_edge = std::cos(P.match_threshold_angle * M_PI / 180.0);

  float v3[3] = {kq.pos_img[0], kq.pos_img[1], P.fm};
  float p_m3[3];
  m3_vec(Rback, v3, p_m3);
  float pmx = p_m3[0] * P.fm / p_m3[2];
  float pmy = p_m3[1] * P.fm / p_m3[2];
  float k_rho = kq.rho * P.fm / p_m3[2];

  float pi0x = pmx + P.cx;
  float pi0y = pmy + P.cy;

  float t_x = -(vel[0] * P.fm - vel[2] * pmx);
  float t_y = -(vel[1] * P.fm - vel[2] * pmy);
  float norm_t = std::sqrt(t_x * t_x + t_y * t_y);

  float DrDv[3] = {P.fm, P.fm, -(pmx + pmy)};
  float rowv[3];
  for (int j = 0; j < 3; ++j) {
    float s = 0;
    for (int k = 0; k < 3; ++k) s += DrDv[k] * Rvel.a[k][j];
    rowv[j] = s;
  }
  float sigma2_t = 0;
  for (int j = 0; j < 3; ++j) sigma2_t += rowv[j] * DrDv[j];

  float dq_min = 0.0, dq_max = 0.0, dq_rho = 0.0;
  int t_steps = 0;
  if (norm_t > 1e-6) {
    t_x /= norm_t;
    t_y /= norm_t;
    dq_rho = norm_t * k_rho;
    dq_min = std::max(float(0.0), norm_t * (k_rho - kq.sigma_rho)) - P.pixel_uncertainty_match;
    dq_max = std::min(max_radius, norm_t * (k_rho + kq.sigma_rho)) + P.pixel_uncertainty_match;
    if (dq_rho > dq_max) {
      dq_rho = 0.5 * (dq_max + dq_min);
      t_steps = cvtt(dq_rho + 0.5);
    } else {
      t_steps = cvtt(std::max(dq_max - dq_rho, dq_rho - dq_min));
    }
  } else {
    t_x = kq.gradient[0];
    t_y = kq.gradient[1];
    norm_t = kq.gradient_norm;
    t_x /= norm_t;
    t_y /= norm_t;
    norm_t = 1.0;
    dq_min = -max_radius - P.pixel_uncertainty_match;
    dq_max = max_radius + P.pixel_uncertainty_match;
    dq_rho = 0.0;
    t_steps = cvtt(dq_max);
  }

  float tn = dq_rho;
  float tp = dq_rho + 1;
  for (int t_i = 0; t_i < t_steps; ++t_i, ++tp, --tn) {
    for (int i_idx = 0; i_idx < 2; ++i_idx) {
      float t;
      if (i_idx) {
        t = tp;
        if (t > dq_max) continue;
      } else {
        t = tn;
        if (t < dq_min) continue;
      }
      int idx = get_index(P, t_y * t + pi0y, t_x * t + pi0x);
      if (idx < 0) continue;
      int cand = old_map->mask[idx];
      if (cand < 0) continue;
      const orc_keyline& k = old_map->kl[cand];
      float cang = (k.gradient[0] * kq.gradient[0] + k.gradient[1] * kq.gradient[1]) / (k.gradient_norm * kq.gradient_norm);
      if (cang < cang_min_edge || std::fabs(k.gradient_norm / kq.gradient_norm - 1.0) > P.match_threshold_norm) continue;
      float v_rho_dr = (P.pixel_uncertainty_match * P.pixel_uncertainty_match + k.sigma_rho * k.sigma_rho * norm_t * norm_t +
                        sigma2_t * k.rho * k.rho);
      if ((t - norm_t * k.rho) * (t - norm_t * k.rho) > v_rho_dr) continue;
      return cand;
    }
  }
  return -1;
}

// The keyline sums of tryVel / extRotVel. mode 0: fp32 running sum in index order, as the reference adds. Modes 1 and 2 are
// DIAGNOSTICS, not the reference (orc_ctx::wide_sums): 1 = double accumulation; 2 = the fp32 terms associated the way the HIP
// kernels associate them (dev_* below) - with it the oracle's sums, and everything minimizeVel / extRotVel derive from them,
// have to equal the device's bit for bit, which isolates the order of these additions as the ONLY fp32 difference between
// the two (tests/test_parity_gpu.py::test_lm_sums_in_device_order_are_bit_exact).
// Device order (rebvio_amd/csrc/track.hip): keyline idx is lane idx % 64 of wave idx / 64; a wave's total is a Hillis-Steele
// scan inside each row of 16 lanes (shifts 1, 2, 4, 8, zero fill), rows combined as (r3 + r2) + (r1 + r0) (wave_total63_f);
// the four waves of a group of 256 keylines are added in order from 0 (do_eval / k_try_vel); tryVel's group records are dealt
// to 16 lanes (record b to lane b % 16, added in ascending b from 0) and those lanes scanned like a row (reduce_staged_records);
// extRotVel's group records are added in ascending order in double and rounded once (hm::sum_xrv / lm_tail_glue).
static float dev_row16(float* s) {
  for (int d = 1; d < 16; d <<= 1)
    for (int i = 15; i >= 0; --i) s[i] = s[i] + (i >= d ? s[i - d] : 0.0f);
  return s[15];
}
static float dev_wave64(const float* x) {
  float r[4];
  for (int row = 0; row < 4; ++row) {
    float s[16];
    for (int i = 0; i < 16; ++i) s[i] = x[row * 16 + i];
    r[row] = dev_row16(s);
  }
  return (r[3] + r[2]) + (r[1] + r[0]);
}
static float dev_group256(const std::vector<float>& t, int b) {
  float x[256];
  for (int i = 0; i < 256; ++i) x[i] = ((size_t)b * 256 + i < t.size()) ? t[(size_t)b * 256 + i] : 0.0f;
  float acc = 0.0f;
  for (int w = 0; w < 4; ++w) acc += dev_wave64(x + 64 * w);
  return acc;
}
struct Acc {
  float f = 0.0f;
  double d = 0.0;
  int mode = 0;
  bool groups_in_double = false;  // mode 2: extRotVel's last stage (see above)
  std::vector<float> terms;       // mode 2: the term of keyline idx (0 where it has none)
  void init(int mode_, size_t n, bool groups_in_double_) {
    mode = mode_;
    groups_in_double = groups_in_double_;
    if (mode == 2) terms.assign(n, 0.0f);
  }
  inline void add(float x, int idx) {
    if (mode == 1) d += (double)x;
    else if (mode == 2) terms[idx] = x;
    else f += x;
  }
  float get() const {
    if (mode == 1) return (float)d;
    if (mode != 2) return f;
    const int nblocks = ((int)terms.size() + 255) / 256;
    if (groups_in_double) {
      double acc = 0.0;
      for (int b = 0; b < nblocks; ++b) acc += (double)dev_group256(terms, b);
      return (float)acc;
    }
    float part[16];
    for (int j = 0; j < 16; ++j) {
      float acc = 0.0f;
      for (int b = j; b < nblocks; b += 16) acc += dev_group256(terms, b);
      part[j] = acc;
    }
    return dev_row16(part);
  }
};

// Core::tryVel + calculatefJ + testfk (core.cpp:39-148)
float try_vel(orc_ctx* c, orc_map* map, float JtJ[9], float JtF[3], const float vel[3], float sigma_rho_min,
              float* residuals) {
  const orc_params& P = c->p;
  const orc_map* nm = c->df_map;
  Acc score, J00, J11, J22, J01, J02, J12, F0, F1, F2;
  for (Acc* a : {&score, &J00, &J11, &J22, &J01, &J02, &J12, &F0, &F1, &F2}) a->init(c->wide_sums, map->kl.size(), false);
  float fi = 0.0f;  // reference leaves this uninitialised: carry-forward semantics (header note)
  const unsigned min_matches = std::min(P.min_match_threshold, c->frame_count);
  for (int idx = 0; idx < (int)map->kl.size(); ++idx) {
    orc_keyline& k = map->kl[idx];
    k.match_id_forward = -1;
    if (map->threshold > 0.0 && k.gradient_norm < map->threshold) continue;
    if (k.sigma_rho > sigma_rho_min || k.matches < min_matches) continue;

    float weight = 1.0;
    if (residuals[idx] > P.reweight_distance) weight = P.reweight_distance / residuals[idx];

    float z_p = 1.0 / k.rho + vel[2];
    float f;
    if (z_p <= 0.0) {
      f = (1.0 / k.sigma_rho) * P.search_range * weight;
      score.add(f * f, idx);
      continue;
    }
    float rho_p = 1.0 / z_p;
    float p_x = rho_p * (vel[0] * P.fm - vel[2] * k.pos_img[0]) + k.pos_img[0];
    float p_y = rho_p * (vel[1] * P.fm - vel[2] * k.pos_img[1]) + k.pos_img[1];
    float p_xc = p_x + P.cx;
    float p_yc = p_y + P.cy;
    int x = cvtt(p_xc + 0.5);
    int y = cvtt(p_yc + 0.5);
    if (x < 1 || y < 1 || (unsigned)x >= (unsigned)P.cols - 1 || (unsigned)y >= (unsigned)P.rows - 1) {
      f = (1.0 / k.sigma_rho) * P.search_range * weight;
      score.add(f * f, idx);
      continue;
    }
    float df_dx, df_dy;
    {  // calculatefJ
      int f_inx = y * P.cols + x;
      int id = c->df_id[f_inx];
      bool matched = false;
      if (id >= 0) {
        const orc_keyline& kn = nm->kl[id];
        float norm_squared = k.gradient_norm * k.gradient_norm;  // testfk(kn, k, thr): _keyline2 = k
        float dot_product = kn.gradient[0] * k.gradient[0] + kn.gradient[1] * k.gradient[1];
        if (!(std::fabs(dot_product - norm_squared) > P.match_treshold * norm_squared)) {
          float dx = p_xc - kn.pos[0];
          float dy = p_yc - kn.pos[1];
          float gnx = kn.gradient[0] / kn.gradient_norm;
          float gny = kn.gradient[1] / kn.gradient_norm;
          fi = (dx * gnx + dy * gny);
          df_dx = gnx / k.sigma_rho;
          df_dy = gny / k.sigma_rho;
          k.match_id_forward = id;
          f = fi / k.sigma_rho;
          matched = true;
        }
      }
      if (!matched) {
        df_dx = 0.0;
        df_dy = 0.0;
        f = P.search_range / k.sigma_rho;
      }
    }
    f *= weight;
    score.add(f * f, idx);
    float jx = rho_p * P.fm * df_dx * weight;
    float jy = rho_p * P.fm * df_dy * weight;
    float jz = -rho_p * (p_x * df_dx + p_y * df_dy) * weight;
    J00.add(jx * jx, idx); J11.add(jy * jy, idx); J22.add(jz * jz, idx);
    J01.add(jx * jy, idx); J02.add(jx * jz, idx); J12.add(jy * jz, idx);
    F0.add(jx * f, idx); F1.add(jy * f, idx); F2.add(jz * f, idx);
    residuals[idx] = std::fabs(fi);
  }
  JtJ[0] = J00.get(); JtJ[1] = J01.get(); JtJ[2] = J02.get();
  JtJ[3] = J01.get(); JtJ[4] = J11.get(); JtJ[5] = J12.get();
  JtJ[6] = J02.get(); JtJ[7] = J12.get(); JtJ[8] = J22.get();
  JtF[0] = F0.get(); JtF[1] = F1.get(); JtF[2] = F2.get();
  return score.get();
}

float estimate_quantile(const orc_map* m, float percentile, int num_bins) {
  // EdgeMap::estimateQuantile (edge_map.cpp:39-56)
  std::vector<int> histogram(num_bins, 0);
  const int n = (int)m->kl.size();
  for (int idx = 0; idx < n; ++idx) {
    int i = cvtt(num_bins * (m->kl[idx].sigma_rho - RHO_MIN) / (RHO_MAX - RHO_MIN));
    i = (i > num_bins - 1) ? (num_bins - 1) : i;
    i = (i < 0) ? 0 : i;
    ++histogram[i];
  }
  float sigma_rho = 1e3;
  for (int i = 0, a = 0; i < num_bins; ++i) {
    if (a > percentile * n) {
      sigma_rho = float(i) * (RHO_MAX - RHO_MIN) / float(num_bins) + RHO_MIN;
      break;
    }
    a += histogram[i];
  }
  return sigma_rho;
}

float minimize_vel(orc_ctx* c, orc_map* map, float vel[3], M3& Rvel, int* accept_mask, float* srm_out) {
  // Core::minimizeVel (core.cpp:150-189)
  const orc_params& P = c->p;
  float sigma_rho_min = estimate_quantile(map, P.quantile_cutoff, P.quantile_num_bins);
  if (srm_out) *srm_out = sigma_rho_min;
  M3 JtJ, ApI, JtJnew;
  float JtF[3], JtFnew[3], h[3], Vnew[3];
  std::vector<float> residuals(map->kl.size() + 1, 0.0f);
  float F = try_vel(c, map, &JtJ.a[0][0], JtF, vel, sigma_rho_min, residuals.data());
  float v = 2.0;
  float tau = 1e-3;
  float mx = JtJ.a[0][0];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      if (JtJ.a[i][j] > mx) mx = JtJ.a[i][j];
  float u = tau * mx;
  float gain;
  int mask = 0;
  for (unsigned iter = 0; iter < P.iterations; ++iter) {
    ApI = JtJ;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) ApI.a[i][j] = JtJ.a[i][j] + ((i == j) ? 1.0f * u : 0.0f * u);
    float neg[3] = {-JtF[0], -JtF[1], -JtF[2]};
    M3 inv = m3_invert(ApI);
    m3_vec(inv, neg, h);
    for (int i = 0; i < 3; ++i) Vnew[i] = vel[i] + h[i];
    float Fnew = try_vel(c, map, &JtJnew.a[0][0], JtFnew, Vnew, sigma_rho_min, residuals.data());
    double den = 0;
    for (int i = 0; i < 3; ++i) den += (0.5 * h[i]) * (double)(u * h[i] - JtF[i]);
    gain = (F - Fnew) / den;
    if (gain > 0.0) {
      F = Fnew;
      for (int i = 0; i < 3; ++i) { vel[i] = Vnew[i]; JtF[i] = JtFnew[i]; }
      JtJ = JtJnew;
      u *= std::max(0.33, 1.0 - ((2.0 * gain - 1.0) * (2.0 * gain - 1.0) * (2.0 * gain - 1.0)));
      v = 2.0;
      mask |= (1 << iter);
    } else {
      u *= v;
      v *= 2.0;
    }
  }
  Rvel = m3_invert(JtJ);
  if (accept_mask) *accept_mask = mask;
  return F;
}

int forward_match(orc_map* old_map, orc_map* new_map) {
  // EdgeMap::forwardMatch (edge_map.cpp:73-99)
  unsigned num = 0;
  for (int idx = 0; idx < (int)old_map->kl.size(); ++idx) {
    const orc_keyline& k = old_map->kl[idx];
    const int idx_f = k.match_id_forward;
    if (idx_f < 0) continue;
    orc_keyline& t = new_map->kl[idx_f];
    if (t.match_id >= 0 && t.rho > k.rho) continue;
    t.rho = k.rho;
    t.sigma_rho = k.sigma_rho;
    t.matches = k.matches + 1;
    t.match_id = idx;
    t.match_pos_img[0] = k.pos_img[0]; t.match_pos_img[1] = k.pos_img[1];
    t.match_gradient[0] = k.gradient[0]; t.match_gradient[1] = k.gradient[1];
    t.match_gradient_norm = k.gradient_norm;
    t.match_id_keyframe = k.match_id_keyframe;
    ++num;
  }
  new_map->matches = num;
  return num;
}

// Core::extRotVel (core.cpp:191-261); sums in row order like Phi.T()*Phi.
int ext_rot_vel(orc_ctx* c, const float vel[3], float Wx[36], float X[6], float JtF_out[6]) {
  const orc_params& P = c->p;
  const orc_map* m = c->df_map;
  Acc JtJ[6][6], JtF[6];
  for (int i = 0; i < 6; ++i) {
    JtF[i].init(c->wide_sums, m->kl.size(), true);
    for (int j = 0; j < 6; ++j) JtJ[i][j].init(c->wide_sums, m->kl.size(), true);
  }
  for (int idx = 0; idx < (int)m->kl.size(); ++idx) {
    const orc_keyline& k = m->kl[idx];
    if (k.match_id < 0) continue;
    float u_x = k.gradient[0] / k.gradient_norm;
    float u_y = k.gradient[1] / k.gradient_norm;
    float rho_t = 1.0 / (1.0 / k.rho + vel[2]);
    float qt_x = k.match_pos_img[0] + rho_t * (vel[0] * P.fm - vel[2] * k.match_pos_img[0]);
    float qt_y = k.match_pos_img[1] + rho_t * (vel[1] * P.fm - vel[2] * k.match_pos_img[1]);
    float q_x = k.pos_img[0];
    float q_y = k.pos_img[1];
    float row[6];
    row[0] = u_x * rho_t * P.fm;
    row[1] = u_y * rho_t * P.fm;
    row[2] = u_x * (-rho_t * q_x) + u_y * (-rho_t * q_y);
    row[3] = -u_x * q_x * q_y / P.fm - u_y * (P.fm + q_y * q_y / P.fm);
    row[4] = u_y * q_x * q_y / P.fm + u_x * (P.fm + q_x * q_x / P.fm);
    row[5] = -u_x * q_y + u_y * q_x;
    float Y = u_x * (q_x - qt_x) + u_y * (q_y - qt_y);
    float dqvel = u_x * (vel[0] * P.fm - vel[2] * k.match_pos_img[0]) + u_y * (vel[1] * P.fm - vel[2] * k.match_pos_img[1]);
    float s_y = std::sqrt(k.sigma_rho * k.sigma_rho * dqvel * dqvel + P.pixel_uncertainty * P.pixel_uncertainty);
    float weight = 1.0;
    if (std::fabs(Y) > P.reweight_distance) weight = std::fabs(Y) / P.reweight_distance;
    float dv = s_y * weight;
    for (int i = 0; i < 6; ++i) row[i] /= dv;
    Y /= dv;
    for (int i = 0; i < 6; ++i) {
      for (int j = 0; j < 6; ++j) JtJ[i][j].add(row[i] * row[j], idx);
      JtF[i].add(row[i] * Y, idx);
    }
  }
  float JtF_f[6];
  for (int i = 0; i < 6; ++i) {
    JtF_f[i] = JtF[i].get();
    for (int j = 0; j < 6; ++j) Wx[i * 6 + j] = JtJ[i][j].get();
  }
  if (JtF_out)
    for (int i = 0; i < 6; ++i) JtF_out[i] = JtF_f[i];
  sym_pinv_solve<6>(Wx, JtF_f, X);
  for (int i = 0; i < 6; ++i)
    if (std::isnan(X[i])) return 0;
  return 1;
}

int directed_match(orc_ctx* c, orc_map* nmap, orc_map* omap, const float vel_[3], const M3& Rvel_, const M3& Rback,
                   int* kf_matches, float max_radius) {
  // EdgeMap::directedMatch (edge_map.cpp:186-218)
  nmap->matches = 0;
  *kf_matches = 0;
  float vel[3];
  m3_vec(Rback, vel_, vel);
  M3 Rvel = m3_mul(m3_mul(Rback, Rvel_), m3_T(Rback));
  for (int idx = 0; idx < (int)nmap->kl.size(); ++idx) {
    orc_keyline& k = nmap->kl[idx];
    int im = search_match(c, omap, k, vel, Rvel, Rback, max_radius);
    if (im < 0) continue;
    const orc_keyline& mk = omap->kl[im];
    k.rho = mk.rho;
    k.sigma_rho = mk.sigma_rho;
    k.match_id = im;
    k.matches = mk.matches + 1;
    k.match_pos_img[0] = mk.pos_img[0]; k.match_pos_img[1] = mk.pos_img[1];
    k.match_gradient[0] = mk.gradient[0]; k.match_gradient[1] = mk.gradient[1];
    k.match_gradient_norm = mk.gradient_norm;
    k.match_id_keyframe = mk.match_id_keyframe;
    if (k.match_id_keyframe >= 0) ++(*kf_matches);
    ++nmap->matches;
  }
  return nmap->matches;
}

int regularize_1iter(const orc_ctx* c, orc_map* m, float thr) {
  // EdgeMap::regularize1Iter (edge_map.cpp:220-259)
  (void)c;
  const int n = (int)m->kl.size();
  int r_num = 0;
  std::vector<float> r(n), s(n);
  std::vector<char> set(n, 0);
  for (int idx = 0; idx < n; ++idx) {
    orc_keyline& k = m->kl[idx];
    if (k.id_next < 0 || k.id_prev < 0) continue;
    const orc_keyline& kn = m->kl[k.id_next];
    const orc_keyline& kp = m->kl[k.id_prev];
    if ((kn.rho - kp.rho) * (kn.rho - kp.rho) > (kn.sigma_rho * kn.sigma_rho + kp.sigma_rho * kp.sigma_rho)) continue;
    float alpha = (kn.gradient[0] * kp.gradient[0] + kn.gradient[1] * kp.gradient[1]) / (kn.gradient_norm * kp.gradient_norm);
    if (alpha < thr) continue;
    alpha = (alpha - thr) / (1.0 - thr);
    alpha /= std::fabs(kn.rho - kp.rho) / (kn.sigma_rho + kp.sigma_rho) + 1.0;
    float wr = 1.0 / (k.sigma_rho * k.sigma_rho);
    float wrn = alpha / (kn.sigma_rho * kn.sigma_rho);
    float wrp = alpha / (kp.sigma_rho * kp.sigma_rho);
    r[idx] = (k.rho * wr + kn.rho * wrn + kp.rho * wrp) / (wr + wrn + wrp);
    s[idx] = (k.sigma_rho * wr + kn.sigma_rho * wrn + kp.sigma_rho * wrp) / (wr + wrn + wrp);
    set[idx] = 1;
    ++r_num;
  }
  for (int idx = 0; idx < n; ++idx)
    if (set[idx]) {
      m->kl[idx].rho = r[idx];
      m->kl[idx].sigma_rho = s[idx];
    }
  return r_num;
}

void update_inverse_depth_arlu(const orc_params& P, orc_keyline& k, const float vel[3]) {
  // Core::updateInverseDepthARLU (core.cpp:424-456)
  float qx = k.pos_img[0], qy = k.pos_img[1];
  float q0x = k.match_pos_img[0], q0y = k.match_pos_img[1];
  float v_rho = k.sigma_rho * k.sigma_rho;
  float ux = k.match_gradient[0] / k.match_gradient_norm;
  float uy = k.match_gradient[1] / k.match_gradient_norm;
  float Y = ux * (qx - q0x) + uy * (qy - q0y);
  float H = ux * (vel[0] * P.fm - vel[2] * q0x) + uy * (vel[1] * P.fm - vel[2] * q0y);
  float rho_p = 1.0 / (1.0 / k.rho + vel[2]);
  float F = 1.0 / (1.0 + k.rho * vel[2]);
  F *= F;
  float p_p = F * v_rho * F + P.reshape_q_abs * P.reshape_q_abs;
  float e = Y - H * rho_p;
  float S = H * p_p * H + P.pixel_uncertainty * P.pixel_uncertainty;
  float K = p_p * H * (1.0 / S);
  k.rho = rho_p + K * e;
  v_rho = (1.0 - K * H) * p_p;
  k.sigma_rho = std::sqrt(v_rho);
  if (k.rho < RHO_MIN) {
    k.sigma_rho += RHO_MIN - k.rho;
    k.rho = RHO_MIN;
  } else if (k.rho > RHO_MAX) {
    k.rho = RHO_MAX;
  } else if (std::isnan(k.rho) || std::isnan(k.sigma_rho) || std::isinf(k.rho) || std::isinf(k.sigma_rho)) {
    k.rho = RHO_INIT;
    k.sigma_rho = RHO_MAX;
  }
}

// Core::gyroBiasCorrection (core.cpp:264-284), dgbias == 0 on entry as in the reference.
void gyro_bias_correction(float X[6], float Wx[36], M3& Wb, const M3& Rg, const M3& Rb, float dgbias_out[3]) {
  float dgbias[3] = {0, 0, 0};
  const M3 Wg = m3_invert(Rg);
  Wb = m3_invert(m3_add(m3_invert(Wb), Rb));
  float Wxb[36];
  std::memcpy(Wxb, Wx, sizeof(Wxb));
  M3 iWgWb = m3_invert(m3_add(Wg, Wb));
  M3 add = m3_mul(Wg, m3_sub(m3_identity(), m3_mul(iWgWb, Wg)));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Wxb[(3 + i) * 6 + 3 + j] += add.a[i][j];
  float X1[6];
  for (int i = 0; i < 6; ++i) {
    float s = 0;
    for (int k = 0; k < 6; ++k) s += Wx[i * 6 + k] * X[k];
    X1[i] = s;
  }
  {
    float t[3];
    M3 m = m3_mul(m3_mul(Wg, iWgWb), Wb);
    m3_vec(m, dgbias, t);
    for (int i = 0; i < 3; ++i) X1[3 + i] += t[i];
  }
  float inv[36];
  cholesky_inverse<6>(Wxb, inv);
  for (int i = 0; i < 6; ++i) {
    float s = 0;
    for (int k = 0; k < 6; ++k) s += inv[i * 6 + k] * X1[k];
    X[i] = s;
  }
  {
    float a[3], b[3], sum[3];
    float xw[3] = {X[3], X[4], X[5]};
    m3_vec(Wg, xw, a);
    m3_vec(Wb, dgbias, b);
    for (int i = 0; i < 3; ++i) sum[i] = a[i] + b[i];
    m3_vec(iWgWb, sum, dgbias);
  }
  Wb = m3_add(Wg, Wb);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Wx[(3 + i) * 6 + 3 + j] += Wg.a[i][j];
  for (int i = 0; i < 3; ++i) dgbias_out[i] = dgbias[i];
}

}  // namespace

// ---- C API ---------------------------------------------------------------------------------------
extern "C" {

void orc_default_params(orc_params* p, int rows, int cols) {
  // camera.hpp:25-45 (EuRoC) scaled is NOT done here: caller sets fm/cx/cy; defaults are EuRoC's.
  p->rows = rows; p->cols = cols;
  float fx = 458.654, fy = 457.296;
  p->fm = 0.5 * (fx + fy);
  p->cx = 367.215; p->cy = 248.375;
  p->keylines_ref = 12000; p->keylines_max = 16000;
  p->pos_neg_threshold = 0.4; p->dog_threshold = 0.095259868922420;
  p->threshold = 0.01; p->gain = 5e-7; p->max_threshold = 0.5; p->min_threshold = 0.005;
  p->search_range = 40.0; p->reweight_distance = 2.0; p->match_treshold = 0.5;
  p->min_match_threshold = 0; p->iterations = 5; p->global_min_matches_threshold = 500;
  p->pixel_uncertainty = 1; p->quantile_cutoff = 0.9; p->quantile_num_bins = 100; p->reshape_q_abs = 1e-4;
  p->pixel_uncertainty_match = 2.0; p->match_threshold_norm = 1.0; p->match_threshold_angle = 45.0;
  p->regularization_threshold = 0.5;
  p->gyro_std_dev = 1.6968e-04; p->gyro_bias_std_dev = 1.9393e-05;
}

void orc_reset_state(orc_ctx* c) {
  c->Bg[0] = c->Bg[1] = c->Bg[2] = 0;
  c->RGBias = m3_identity();
  c->RGyro = m3_identity();
  c->W_Bg = m3_invert(m3_scale(c->RGBias, 100.0f));  // imu.hpp:181
  orc_ls4_reset(c);
}

orc_ctx* orc_create(const orc_params* p) {
  orc_ctx* c = new orc_ctx;
  c->p = *p;
  const size_t n = (size_t)p->rows * p->cols;
  make_filter(c->filter[0], p->rows, p->cols, 3.56359, 3);                           // scale_space.cpp:186
  make_filter(c->filter[1], p->rows, p->cols, c->filter[0].sigma_true * 1.2599, 3);  // scale_space.cpp:186
  c->dog.assign(n, 0.f);
  c->mag.assign(n, 0.f);
  c->det_threshold = p->threshold;
  c->auto_threshold = p->threshold;  // edge_detector.cpp:20
  c->det_mask.assign(n, -1);
  c->df_id.assign(n, -1);
  c->df_dist.assign(n, std::numeric_limits<int>::max());
  orc_reset_state(c);
  return c;
}

void orc_destroy(orc_ctx* c) { delete c; }

void orc_scale_space(orc_ctx* c, const float* img, float* s0, float* s1, float* dog, float* mag) {
  scale_space_build(c, img);
  const size_t nb = (size_t)c->p.rows * c->p.cols * sizeof(float);
  if (s0) std::memcpy(s0, c->scale[0].data(), nb);
  if (s1) std::memcpy(s1, c->scale[1].data(), nb);
  if (dog) std::memcpy(dog, c->dog.data(), nb);
  if (mag) std::memcpy(mag, c->mag.data(), nb);
}

void orc_integral_image(int rows, int cols, const float* in, float* out) { integral_image(rows, cols, in, out); }

void orc_box_average(int rows, int cols, int d, const float* ii, float* out) {
  std::vector<float> div;
  make_divisors(rows, cols, d, div);
  box_average(rows, cols, d, ii, div.data(), out);
}

int orc_filter_width(orc_ctx* c, int filter, int pass) { return c->filter[filter].widths[pass]; }

orc_map* orc_detect(orc_ctx* c, const float* img, uint64_t ts_us) {
  // EdgeDetector::detect (edge_detector.cpp:30-43)
  StageTimer tick(&c->stage_acc[0]);
  orc_params& P = c->p;
  if (P.gain > 0) {
    c->det_threshold -= P.gain * float(P.keylines_ref - c->keylines_count);
    c->det_threshold = (c->det_threshold > P.max_threshold) ? P.max_threshold
                       : ((c->det_threshold < P.min_threshold) ? P.min_threshold : c->det_threshold);
  }
  orc_map* map = build_edge_map(c, img, ts_us);
  join_edges(c, map);
  tune_threshold(c, map);
  map->mask = c->det_mask;  // the map's own image-index -> keyline-index table (edge_detector.cpp:112)
  return map;
}

float orc_detector_threshold(orc_ctx* c) { return c->det_threshold; }
float orc_detector_auto_threshold(orc_ctx* c) { return c->auto_threshold; }
void orc_detector_mask(orc_ctx* c, int* out) { std::memcpy(out, c->det_mask.data(), c->det_mask.size() * sizeof(int)); }

int orc_map_size(orc_map* m) { return (int)m->kl.size(); }
float orc_map_threshold(orc_map* m) { return m->threshold; }
void orc_map_set_threshold(orc_map* m, float t) { m->threshold = t; }
void orc_map_get_keylines(orc_map* m, orc_keyline* out) {
  if (!m->kl.empty()) std::memcpy(out, m->kl.data(), m->kl.size() * sizeof(orc_keyline));
}
void orc_map_set_keylines(orc_map* m, const orc_keyline* in, int n) { m->kl.assign(in, in + n); }
void orc_map_get_mask(orc_map* m, int* out) { std::memcpy(out, m->mask.data(), m->mask.size() * sizeof(int)); }
orc_map* orc_map_clone(orc_map* m) { return new orc_map(*m); }
void orc_map_free(orc_map* m) { delete m; }

void orc_build_distance_field(orc_ctx* c, orc_map* map) {
  // DistanceField::build (core.hpp:37-59)
  const orc_params& P = c->p;
  c->df_map = map;
  const int size = P.rows * P.cols;
  for (int i = 0; i < size; ++i) c->df_id[i] = -1;
  for (int idx = 0; idx < (int)map->kl.size(); ++idx) {
    const orc_keyline& k = map->kl[idx];
    if (map->threshold > 0.0 && k.gradient_norm < map->threshold) continue;
    for (int r = cvtt(-P.search_range); r < P.search_range; ++r) {
      int fi = get_index(P, (k.gradient[1] / k.gradient_norm) * float(r) + k.pos[1],
                         (k.gradient[0] / k.gradient_norm) * float(r) + k.pos[0]);
      if (fi < 0) continue;
      if (c->df_id[fi] >= 0 && c->df_dist[fi] < std::abs(r)) continue;
      c->df_dist[fi] = std::abs(r);
      c->df_id[fi] = idx;
    }
  }
}

void orc_distance_field(orc_ctx* c, int* id_out, int* dist_out) {
  const size_t n = c->df_id.size();
  if (id_out) std::memcpy(id_out, c->df_id.data(), n * sizeof(int));
  if (dist_out) std::memcpy(dist_out, c->df_dist.data(), n * sizeof(int));
}

void orc_rotate_keylines(orc_ctx* c, orc_map* m, const float R_[9]) {
  // EdgeMap::rotateKeylines (edge_map.cpp:58-71). makeVector(float,float,1.0) / (..,0.0) mix
  // float and double arguments -> TooN's double overload -> the product is formed in double.
  const float fm = c->p.fm;
  for (orc_keyline& k : m->kl) {
    double v[3] = {(double)(k.pos_img[0] / fm), (double)(k.pos_img[1] / fm), 1.0};
    float q[3];
    for (int i = 0; i < 3; ++i) {
      double s = 0;
      for (int j = 0; j < 3; ++j) s += (double)R_[i * 3 + j] * v[j];
      q[i] = (float)s;
    }
    if (fabs((double)q[2]) > 0.0) {
      k.pos_img[0] = q[0] / q[2] * fm;
      k.pos_img[1] = q[1] / q[2] * fm;
      k.rho /= q[2];
      k.sigma_rho /= q[2];
    }
    double g[3] = {(double)k.gradient[0], (double)k.gradient[1], 0.0};
    for (int i = 0; i < 3; ++i) {
      double s = 0;
      for (int j = 0; j < 3; ++j) s += (double)R_[i * 3 + j] * g[j];
      q[i] = (float)s;
    }
    k.gradient[0] = q[0];
    k.gradient[1] = q[1];
  }
}

float orc_estimate_quantile(orc_map* m, float percentile, int num_bins) { return estimate_quantile(m, percentile, num_bins); }

void orc_set_wide_sums(orc_ctx* c, int on) { c->wide_sums = on; }

float orc_try_vel(orc_ctx* c, orc_map* m, const float vel[3], float sigma_rho_min, float* residuals, float JtJ[9],
                  float JtF[3]) {
  return try_vel(c, m, JtJ, JtF, vel, sigma_rho_min, residuals);
}

float orc_minimize_vel(orc_ctx* c, orc_map* m, float vel[3], float Rvel[9], int* accept_mask, float* srm) {
  M3 R;
  float F = minimize_vel(c, m, vel, R, accept_mask, srm);
  m3_to(R, Rvel);
  return F;
}

int orc_forward_match(orc_map* o, orc_map* n) { return forward_match(o, n); }

int orc_ext_rot_vel(orc_ctx* c, const float vel[3], float Wx[36], float X[6], float JtF_out[6]) {
  return ext_rot_vel(c, vel, Wx, X, JtF_out);
}

int orc_directed_match(orc_ctx* c, orc_map* nm, orc_map* om, const float vel[3], const float Rvel[9], const float Rback[9],
                       int* kf, float max_radius) {
  return directed_match(c, nm, om, vel, m3_from(Rvel), m3_from(Rback), kf, max_radius);
}

int orc_regularize(orc_map* m) {
  // config is default-constructed per map in the reference (edge_map.hpp:34): threshold 0.5
  return regularize_1iter(nullptr, m, 0.5f);
}

void orc_update_inverse_depth(orc_ctx* c, const float vel[3]) {
  // Core::updateInverseDepth (core.cpp:417-422)
  orc_map* m = c->df_map;
  for (orc_keyline& k : m->kl)
    if (k.match_id >= 0) update_inverse_depth_arlu(c->p, k, vel);
}

int orc_track_pair(orc_ctx* c, orc_map* old_map, orc_map* new_map, const float* R_prior, float frame_dt, orc_pair_out* out) {
  // Rebvio::stateEstimationProcess body (rebvio.cpp:135-259), accelerometer/SAB branch excluded.
  const orc_params& P = c->p;
  std::memset(out, 0, sizeof(*out));
  StageTimer tick_pair(&c->stage_acc[5]);  // whole pair step; the four timed stages are subtracted in orc_stage_seconds
  M3 P_V = m3_scale(m3_identity(), std::numeric_limits<float>::max());
  M3 R = m3_identity();

  {
    StageTimer tick(&c->stage_acc[1]);
    orc_build_distance_field(c, new_map);  // rebvio.cpp:142
  }

  if (R_prior) R = m3_from(R_prior);  // rebvio.cpp:163
  {                                   // R.T() = SO3(Bg)*R.T()  (rebvio.cpp:164)
    M3 RT = m3_mul(so3_exp(c->Bg), m3_T(R));
    R = m3_T(RT);
  }
  {
    float RT[9];
    m3_to(m3_T(R), RT);
    orc_rotate_keylines(c, old_map, RT);  // rebvio.cpp:165
  }

  float Vg[3] = {0, 0, 0};
  M3 P_Vg;
  {
    StageTimer tick(&c->stage_acc[2]);
    out->F = minimize_vel(c, old_map, Vg, P_Vg, &out->lm_accept_mask, &out->sigma_rho_min);  // rebvio.cpp:169
  }
  forward_match(old_map, new_map);  // rebvio.cpp:172

  float Xv[6], W_Xv[36];
  {
    StageTimer tick(&c->stage_acc[3]);
    out->ext_ok = ext_rot_vel(c, Vg, W_Xv, Xv, nullptr);  // rebvio.cpp:177
  }
  float Xgv[6], W_Xgv[36];
  std::memcpy(Xgv, Xv, sizeof(Xv));
  std::memcpy(W_Xgv, W_Xv, sizeof(W_Xv));

  // rebvio.cpp:186-191
  float s_b = P.gyro_bias_std_dev * P.gyro_bias_std_dev * frame_dt * frame_dt;
  float s_g = P.gyro_std_dev * P.gyro_std_dev * frame_dt * frame_dt;
  c->RGBias = m3_scale(m3_identity(), 1.0f);
  c->RGyro = m3_scale(m3_identity(), 1.0f);
  for (int i = 0; i < 3; ++i) {
    c->RGBias.a[i][i] = s_b;
    c->RGyro.a[i][i] = s_g;
  }
  float dg[3];
  gyro_bias_correction(Xgv, W_Xgv, c->W_Bg, c->RGyro, c->RGBias, dg);
  for (int i = 0; i < 3; ++i) c->Bg[i] += dg[i];
  float dVgv[3] = {Xgv[0], Xgv[1], Xgv[2]};
  float dWgv[3] = {Xgv[3], Xgv[4], Xgv[5]};

  // rebvio.cpp:195-203
  M3 R0 = so3_exp(dWgv);
  {
    M3 RT = m3_mul(R0, m3_T(R));
    R = m3_T(RT);
  }
  float Vgv[3];
  m3_vec(R0, Vg, Vgv);
  for (int i = 0; i < 3; ++i) Vgv[i] += dVgv[i];
  float V[3] = {Vgv[0], Vgv[1], Vgv[2]};
  float R_Xgv[36];
  cholesky_inverse<6>(W_Xgv, R_Xgv);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) P_V.a[i][j] = R_Xgv[i * 6 + j];

  M3 Rgva = R;  // rebvio.cpp:228
  {
    float R0a[9];
    m3_to(R0, R0a);
    orc_rotate_keylines(c, old_map, R0a);  // rebvio.cpp:232
  }

  for (int i = 0; i < 3; ++i) { out->Vg[i] = Vg[i]; out->V[i] = V[i]; }
  m3_to(P_Vg, out->P_Vg);
  std::memcpy(out->Xv, Xv, sizeof(Xv));
  std::memcpy(out->W_Xv, W_Xv, sizeof(W_Xv));
  std::memcpy(out->Xgv, Xgv, sizeof(Xgv));
  m3_to(Rgva, out->R);
  m3_to(P_V, out->P_V);

  if (std::isnan(V[0]) || std::isnan(V[1]) || std::isnan(V[2])) {  // rebvio.cpp:236
    out->status = 1;
    return 1;
  }
  {
    StageTimer tick(&c->stage_acc[4]);
    out->klm_num = directed_match(c, new_map, old_map, V, P_V, Rgva, &out->kf_matches, P.search_range);  // rebvio.cpp:245
  }
  if ((unsigned)out->klm_num < P.global_min_matches_threshold) {                                         // rebvio.cpp:247
    out->status = 2;
    return 2;
  }
  out->reg_num = orc_regularize(new_map);  // rebvio.cpp:256
  orc_update_inverse_depth(c, V);          // rebvio.cpp:259
  return 0;
}

void orc_ls4_reset(orc_ctx* c) {
  for (int i = 0; i < 3; ++i) c->ls4_V[i] = c->ls4_V0[i] = c->ls4_V1[i] = c->ls4_V2[i] = c->ls4_V3[i] = 0;
  for (int i = 0; i < 5; ++i) c->ls4_T[i] = 0;
  for (int i = 0; i < 4; ++i) c->ls4_Dt[i] = 0;
}

void orc_estimate_ls4_acceleration(orc_ctx* c, const float vel[3], float acc[3], const float R_[9], float dt) {
  // Core::estimateLs4Acceleration (core.cpp:285-333); the V[3] read out of bounds at core.cpp:321
  // multiplies a term whose weights sum to zero -> replaced by 0.
  M3 RT = m3_T(m3_from(R_));
  m3_vec(RT, c->ls4_V2, c->ls4_V3);
  m3_vec(RT, c->ls4_V1, c->ls4_V2);
  m3_vec(RT, c->ls4_V0, c->ls4_V1);
  m3_vec(RT, c->ls4_V, c->ls4_V0);
  for (int i = 0; i < 3; ++i) c->ls4_V[i] = vel[i];
  float* T = c->ls4_T;
  float* Dt = c->ls4_Dt;
  for (int i = 0; i < 3; ++i) Dt[i] = Dt[i + 1];
  Dt[3] = dt;
  T[0] = 0.0;
  float mt = 0.0;
  for (int i = 0; i < 4; ++i) {
    T[i + 1] = T[i] + Dt[i];
    mt += T[i + 1];
  }
  mt /= 5.0;
  float den = 0.0;
  for (int i = 0; i < 5; ++i) den += (T[i] - mt) * (T[i] - mt);
  for (int i = 0; i < 3; ++i) {
    float vm = (c->ls4_V[i] + c->ls4_V0[i] + c->ls4_V1[i] + c->ls4_V2[i] + 0.0f) / 5.0;
    float num = (c->ls4_V[i] - vm) * (T[4] - mt);
    num += (c->ls4_V0[i] - vm) * (T[3] - mt);
    num += (c->ls4_V1[i] - vm) * (T[2] - mt);
    num += (c->ls4_V2[i] - vm) * (T[1] - mt);
    num += (c->ls4_V3[i] - vm) * (T[0] - mt);
    if (den > 0.0) acc[i] = num / den;
  }
}

void orc_so3_exp(const float w[3], float R[9]) { m3_to(so3_exp(w), R); }
void orc_sym6_solve(const float A[36], const float b[6], float x[6]) { sym_pinv_solve<6>(A, b, x); }

// ---- single-keyline forms of the reference's public methods (checkers for the C++ surface) ------------------------
int orc_search_match(orc_ctx* c, orc_map* searched, const orc_keyline* query, const float vel[3], const float Rvel[9],
                     const float Rback[9], float max_radius) {
  return search_match(c, searched, *query, vel, m3_from(Rvel), m3_from(Rback), max_radius);
}

int orc_test_fk(const orc_keyline* k1, const orc_keyline* k2, float similarity_threshold) {
  // Core::testfk (core.cpp:39-44)
  float norm_squared = k2->gradient_norm * k2->gradient_norm;
  float dot_product = k1->gradient[0] * k2->gradient[0] + k1->gradient[1] * k2->gradient[1];
  if (std::fabs(dot_product - norm_squared) > similarity_threshold * norm_squared) return 0;
  return 1;
}

float orc_calculate_fj(orc_ctx* c, int f_inx, float* df_dx, float* df_dy, orc_keyline* keyline, float px, float py, int* mnum,
                       float* fi) {
  // Core::calculatefJ (core.cpp:46-76) on the distance field last built (orc_build_distance_field)
  const orc_params& P = c->p;
  const int id = c->df_id[f_inx];
  if (id < 0) {
    *df_dx = 0.0;
    *df_dy = 0.0;
    return P.search_range / keyline->sigma_rho;
  }
  const orc_keyline& k = c->df_map->kl[id];
  if (!orc_test_fk(&k, keyline, P.match_treshold)) {
    *df_dx = 0.0;
    *df_dy = 0.0;
    return P.search_range / keyline->sigma_rho;
  }
  float dx = px - k.pos[0];
  float dy = py - k.pos[1];
  float gnx = k.gradient[0] / k.gradient_norm;
  float gny = k.gradient[1] / k.gradient_norm;
  *fi = (dx * gnx + dy * gny);
  *df_dx = gnx / keyline->sigma_rho;
  *df_dy = gny / keyline->sigma_rho;
  ++*mnum;
  keyline->match_id_forward = id;
  return *fi / keyline->sigma_rho;
}

void orc_update_inverse_depth_arlu(orc_ctx* c, orc_keyline* keyline, const float vel[3]) {
  update_inverse_depth_arlu(c->p, *keyline, vel);
}

void orc_smooth(orc_ctx* c, const float* img, float sigma, int n, float* out, int widths_out[3]) {
  // FastGaussian(camera, sigma, n).smooth(img) (scale_space.cpp:14-41,173-182)
  BoxGaussian f;
  make_filter(f, c->p.rows, c->p.cols, sigma, n);
  std::vector<float> o;
  smooth(c, f, img, o);
  std::memcpy(out, o.data(), o.size() * sizeof(float));
  if (widths_out)
    for (int i = 0; i < 3; ++i) widths_out[i] = i < n ? f.widths[i] : 0;
}

void orc_smooth_n(orc_ctx* c, const float* img, float sigma, int n, float* out, int* widths_out) {
  // the same for any number of box passes (scale_space.cpp:14-41 takes n; the reference itself constructs n = 3 only)
  BoxGaussian f;
  make_filter(f, c->p.rows, c->p.cols, sigma, n);
  std::vector<float> o;
  smooth(c, f, img, o);
  std::memcpy(out, o.data(), o.size() * sizeof(float));
  if (widths_out)
    for (int i = 0; i < n; ++i) widths_out[i] = f.widths[i];
}

void orc_stage_seconds(orc_ctx* c, double out[6], int reset) {
  for (int i = 0; i < 5; ++i) out[i] = c->stage_acc[i];
  out[5] = c->stage_acc[5] - c->stage_acc[1] - c->stage_acc[2] - c->stage_acc[3] - c->stage_acc[4];
  if (reset)
    for (double& v : c->stage_acc) v = 0.0;
}

double orc_run_stream(orc_ctx* c, const uint8_t* frames, const int* idx, int nframes, int threads, int* keyline_counts,
                      int* match_counts, float* pose_out) {
  return orc_run_stream_ex(c, frames, idx, nframes, threads, keyline_counts, match_counts, pose_out, nullptr);
}

double orc_run_stream_ex(orc_ctx* c, const uint8_t* frames, const int* idx, int nframes, int threads, int* keyline_counts,
                         int* match_counts, float* pose_out, double* frame_done_s) {
  const size_t npx = (size_t)c->p.rows * c->p.cols;
  auto to_float = [&](int i, std::vector<float>& img) {
    const uint8_t* src = frames + (size_t)idx[i] * npx;
    img.resize(npx);
    for (size_t k = 0; k < npx; ++k) img[k] = (float)src[k] * 3.0f;  // convertTo(CV_32F, 3.0) rebvio.cpp:43
  };
  auto track = [&](orc_map* o, orc_map* n, int i) {
    orc_pair_out po;
    orc_track_pair(c, o, n, nullptr, 0.05f, &po);
    if (match_counts) match_counts[i] = po.klm_num;
    if (pose_out) {
      for (int k = 0; k < 3; ++k) pose_out[i * 6 + k] = po.Vg[k];
      for (int k = 0; k < 3; ++k) pose_out[i * 6 + 3 + k] = po.Xgv[3 + k];
    }
  };
  auto t0 = std::chrono::steady_clock::now();
  // frame_done_s[i] = seconds since the start at which frame i had been detected AND tracked against its predecessor
  auto stamp = [&](int i) {
    if (frame_done_s) frame_done_s[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  if (threads <= 1) {
    std::vector<float> img;
    orc_map* prev = nullptr;
    for (int i = 0; i < nframes; ++i) {
      to_float(i, img);
      orc_map* m = orc_detect(c, img.data(), (uint64_t)i * 50000);
      if (keyline_counts) keyline_counts[i] = (int)m->kl.size();
      if (prev) {
        track(prev, m, i);
        orc_map_free(prev);
      }
      stamp(i);
      prev = m;
    }
    if (prev) orc_map_free(prev);
  } else {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::pair<int, orc_map*>> q;
    bool done = false;
    std::thread det([&] {
      std::vector<float> img;
      for (int i = 0; i < nframes; ++i) {
        to_float(i, img);
        orc_map* m = orc_detect(c, img.data(), (uint64_t)i * 50000);
        if (keyline_counts) keyline_counts[i] = (int)m->kl.size();
        {
          std::lock_guard<std::mutex> g(mu);
          q.emplace_back(i, m);
        }
        cv.notify_one();
      }
      {
        std::lock_guard<std::mutex> g(mu);
        done = true;
      }
      cv.notify_one();
    });
    orc_map* prev = nullptr;
    for (;;) {
      std::pair<int, orc_map*> it;
      {
        std::unique_lock<std::mutex> g(mu);
        cv.wait(g, [&] { return !q.empty() || done; });
        if (q.empty()) break;
        it = q.front();
        q.pop_front();
      }
      if (prev) {
        track(prev, it.second, it.first);
        orc_map_free(prev);
      }
      stamp(it.first);
      prev = it.second;
    }
    det.join();
    if (prev) orc_map_free(prev);
  }
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"

// ---- full VIO glue (rebvio.cpp:92-293, core.cpp:335-414) ---------------------------------------------------------
namespace {
// ImuStateConfig defaults (types/imu.hpp:154-168)
constexpr float kGNorm = 9.81, kGUnc = 2e-3, kGNormUnc = 0.2e3, kAccStd = 2.0e-3, kVbiasStd = 1e-7, kScaleStdInit = 1.2e-3;
constexpr int kInitBiasFrameNum = 10;

void mean_acceleration(orc_ctx* c, const float sacc[3], float acc[3], const M3& R) {
  // Core::estimateMeanAcceleration (core.cpp:335-347)
  M3 RT = m3_T(R);
  m3_vec(RT, c->meanA[2], c->meanA[3]);
  m3_vec(RT, c->meanA[1], c->meanA[2]);
  m3_vec(RT, c->meanA[0], c->meanA[1]);
  for (int i = 0; i < 3; ++i) c->meanA[0][i] = sacc[i];
  for (int i = 0; i < 3; ++i) acc[i] = 0.25 * (((c->meanA[0][i] + c->meanA[1][i]) + c->meanA[2][i]) + c->meanA[3][i]);
}

// Core::estimateBias (core.cpp:350-414)
float estimate_bias(orc_ctx* c, const float sacc[3], const float facc[3], float kP, const M3& Rot, const float Wvw[36],
                    float Xvw[6]) {
  float* X = c->sabX;
  Mat<7, 7> F = Mat<7, 7>::zeros();
  F.a[0][0] = kP;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) F.a[1 + i][1 + j] = Rot.a[j][i];
  for (int i = 0; i < 3; ++i) F.a[4 + i][4 + i] = 1.0f;
  const float G[3] = {X[1], X[2], X[3]};
  M3 GProd;
  const float gp[9] = {0.0f, G[2], -G[1], -G[2], 0.0f, G[0], G[1], -G[0], 0.0f};
  GProd = m3_from(gp);
  Mat<7, 7> Q = Mat<7, 7>::zeros();
  const float tn = std::tan(X[0]);
  Q.a[0][0] = c->QKp / (1.0 + tn * tn);
  const M3 Qg2 = m3_add(m3_mul(m3_mul(m3_T(GProd), c->Qrot), GProd), c->Qg);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      Q.a[1 + i][1 + j] = Qg2.a[i][j];
      Q.a[4 + i][4 + j] = c->Qbias.a[i][j];
    }
  float Xp[7];
  mvec(F, X, Xp);
  for (int i = 0; i < 7; ++i) X[i] = Xp[i];
  Mat<7, 7> Pp = mmul(mmul(F, c->sabP), mT(F));
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) Pp.a[i][j] = Pp.a[i][j] + Q.a[i][j];
  SabCfg cfg;
  for (int i = 0; i < 3; ++i) {
    cfg.a_v[i] = facc[i];
    cfg.a_s[i] = sacc[i];
  }
  cfg.G = kGNorm;
  for (int i = 0; i < 7; ++i) cfg.x_p[i] = X[i];
  cfg.Rv = c->Rv;
  cfg.Rs = c->Rs;
  cfg.Rg = c->Rg_sab;
  cfg.Pp = Pp;
  sab_gauss_newton(cfg, X, 20);
  Mat<7, 7> JtJ;
  float JtF[7];
  sab_problem(cfg, JtJ, JtF, X);
  c->sabP = chol_inv<7>(JtJ);
  float k = std::tan(X[0]);
  if (k < 0 || std::isnan(k) || std::isinf(k)) k = 0;
  for (int i = 0; i < 3; ++i) {
    c->g_est[i] = X[1 + i];
    c->b_est[i] = X[4 + i];
  }
  M3 WVBias;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) WVBias.a[i][j] = JtJ.a[4 + i][4 + j];
  float A6[36];
  for (int i = 0; i < 36; ++i) A6[i] = 0.0f + Wvw[i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A6[(3 + i) * 6 + 3 + j] = WVBias.a[i][j] + Wvw[(3 + i) * 6 + 3 + j];
  const float wc[3] = {Xvw[3] - c->b_est[0], Xvw[4] - c->b_est[1], Xvw[5] - c->b_est[2]};
  float wx[3];
  m3_vec(WVBias, wc, wx);
  float rhs[6];
  for (int i = 0; i < 6; ++i) {
    float s2 = 0;
    for (int k2 = 0; k2 < 6; ++k2) s2 += Wvw[i * 6 + k2] * Xvw[k2];
    rhs[i] = s2 + ((i >= 3) ? wx[i - 3] : 0.0f);
  }
  float inv[36];
  cholesky_inverse<6>(A6, inv);
  for (int i = 0; i < 6; ++i) {
    float s2 = 0;
    for (int k2 = 0; k2 < 6; ++k2) s2 += inv[i * 6 + k2] * rhs[k2];
    Xvw[i] = s2;
  }
  return k;
}
}  // namespace

// ---- front end: convertTo(CV_32F, 3.0) + cv::undistort (rebvio.cpp:43-47, camera.hpp:39-40,54-58) -----
// OpenCV is a third-party dependency absent from /root/reference (found via find_package(OpenCV), version unpinned); this
// restates its published algorithm: cv::undistort builds, per stripe of max(1, 4096/cols) rows, CV_16SC2 + CV_16UC1 maps
// with initUndistortRectifyMap (double arithmetic, normalised x advanced by repeated addition of 1/fx, principal point
// moved by the stripe offset, coordinates quantised to 1/32 px with cvRound = round-half-even) and calls
// remap(INTER_LINEAR, BORDER_CONSTANT 0), whose float path weighs the four taps with the table tab_y[k1]*tab_x[k2].
// Parity unpinned (no OpenCV here); exactness of the interpolation itself: every tap*weight product and their sum are
// exactly representable for 8-bit*3 sources, so only the double-precision map can differ from a real OpenCV.
extern "C" void orc_front_end_u8(orc_ctx* c, const uint8_t* img, const float K4[4], const float D5[5], float* out) {
  const int R = c->p.rows, C = c->p.cols;
  std::vector<float> src((size_t)R * C);
  for (size_t i = 0; i < src.size(); ++i) src[i] = float(img[i]) * 3.0f;  // convertTo(CV_32F, 3.0)
  const double fx = K4[0], fy = K4[1], u0 = K4[2], v0 = K4[3];
  const double k1 = D5[0], k2 = D5[1], p1 = D5[2], p2 = D5[3], k3 = D5[4];
  const int stripe0 = std::min(std::max(1, (1 << 12) / std::max(C, 1)), R);
  std::vector<short> m1((size_t)stripe0 * C * 2);
  std::vector<unsigned short> m2((size_t)stripe0 * C);
  float tab[32][2];
  for (int t = 0; t < 32; ++t) {
    const float x = float(t) * (1.0f / 32.0f);
    tab[t][0] = 1.0f - x;
    tab[t][1] = x;
  }
  for (int y = 0; y < R; y += stripe0) {
    const int sh = std::min(stripe0, R - y);
    // inverse of Ar = [fx 0 u0; 0 fy v0-y; 0 0 1]
    const double ir[9] = {1.0 / fx, 0.0, -u0 / fx, 0.0, 1.0 / fy, -(v0 - y) / fy, 0.0, 0.0, 1.0};
    for (int i = 0; i < sh; ++i) {
      double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
      for (int j = 0; j < C; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
        const double w = 1.0 / _w, x = _x * w, yy = _y * w;
        const double x2 = x * x, y2 = yy * yy;
        const double r2 = x2 + y2, _2xy = 2 * x * yy;
        const double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
        const double xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2);
        const double yd = yy * kr + p1 * (r2 + 2 * y2) + p2 * _2xy;
        const double u = fx * xd + u0, v = fy * yd + v0;
        const int iu = (int)std::lrint(u * 32), iv = (int)std::lrint(v * 32);
        m1[((size_t)i * C + j) * 2] = (short)(iu >> 5);
        m1[((size_t)i * C + j) * 2 + 1] = (short)(iv >> 5);
        m2[(size_t)i * C + j] = (unsigned short)((iv & 31) * 32 + (iu & 31));
      }
    }
    for (int i = 0; i < sh; ++i)
      for (int j = 0; j < C; ++j) {
        const int sx = m1[((size_t)i * C + j) * 2], sy = m1[((size_t)i * C + j) * 2 + 1];
        const int fxy = m2[(size_t)i * C + j];
        const float* ty = tab[fxy >> 5];
        const float* tx = tab[fxy & 31];
        const float w4[4] = {ty[0] * tx[0], ty[0] * tx[1], ty[1] * tx[0], ty[1] * tx[1]};
        float v4[4];
        for (int k = 0; k < 4; ++k) {
          const int xx = sx + (k & 1), yy = sy + (k >> 1);
          v4[k] = (xx >= 0 && xx < C && yy >= 0 && yy < R) ? src[(size_t)yy * C + xx] : 0.0f;
        }
        out[(size_t)(y + i) * C + j] = v4[0] * w4[0] + v4[1] * w4[1] + v4[2] * w4[2] + v4[3] * w4[3];
      }
  }
}

extern "C" void orc_vio_reset(orc_ctx* c, const float R_c2i[9], const float t_c2i[3]) {
  orc_reset_state(c);
  c->R_c2i = m3_from(R_c2i);
  for (int i = 0; i < 3; ++i) c->t_c2i[i] = t_c2i[i];
  c->num_frames = 0;
  c->initialized = 0;
  c->num_gyro_init = 0;
  for (int i = 0; i < 3; ++i) c->gyro_init[i] = c->g_init[i] = c->Pos[i] = c->Av[i] = c->As[i] = c->g_est[i] = c->b_est[i] = 0;
  c->Kscale = 1.0;
  c->P_Kp = 5e-6;
  c->u_est[0] = 1; c->u_est[1] = 0; c->u_est[2] = 0;
  c->R_global = m3_identity();
  // SABEstimator::State (sab_estimator.hpp:50-64)
  const float x0[7] = {(float)M_PI_4, 0.0f, kGNorm, 0.0f, 0.0f, 0.0f, 0.0f};
  for (int i = 0; i < 7; ++i) c->sabX[i] = x0[i];
  c->sabP = Mat<7, 7>::zeros();
  const float pd[7] = {kScaleStdInit * kScaleStdInit, 100.0f, 100.0f, 100.0f, (float)(kVbiasStd * kVbiasStd * 1e1),
                       (float)(kVbiasStd * kVbiasStd * 1e1), (float)(kVbiasStd * kVbiasStd * 1e1)};
  for (int i = 0; i < 7; ++i) c->sabP.a[i][i] = pd[i];
  c->Qg = m3_scale(m3_identity(), kGUnc * kGUnc);
  c->Rg_sab = kGNormUnc * kGNormUnc;
  c->Rs = m3_scale(m3_identity(), kAccStd * kAccStd);
  c->Qbias = m3_scale(m3_identity(), kVbiasStd * kVbiasStd);
  c->Qrot = m3_identity();
  c->Rv = m3_identity();
  c->QKp = 5e-6;
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 3; ++i) c->meanA[k][i] = 0;
}

extern "C" void orc_vio_add_imu(orc_ctx* c, orc_map* m, uint64_t ts_us, const float gyro[3], const float acc[3]) {
  m->imu.add(ts_us, gyro, acc, c->R_c2i);
}

extern "C" int orc_vio_step(orc_ctx* c, orc_map* old_map, orc_map* new_map, orc_vio_out* out) {
  const orc_params& P = c->p;
  std::memset(out, 0, sizeof(*out));
  orc_pair_out* po = &out->pair;
  const float FMAX = std::numeric_limits<float>::max();
  M3 P_V = m3_scale(m3_identity(), FMAX), P_W = m3_scale(m3_identity(), FMAX);
  orc_build_distance_field(c, new_map);  // rebvio.cpp:142

  IntImu& imu = new_map->imu;  // rebvio.cpp:145-160
  imu.get(c->R_c2i, c->t_c2i);
  if (!c->initialized && c->num_frames > 0) {
    for (int i = 0; i < 3; ++i) {
      c->gyro_init[i] += imu.gyro[i] * imu.dt_s();
      c->g_init[i] -= imu.cacc[i];
    }
    if (++c->num_gyro_init > kInitBiasFrameNum) {
      for (int i = 0; i < 3; ++i) c->Bg[i] = c->gyro_init[i] / c->num_gyro_init;
      c->W_Bg = m3_invert(m3_scale(c->RGBias, 1e2f));
      for (int i = 0; i < 3; ++i) c->sabX[1 + i] = c->g_init[i] / c->num_gyro_init;
      c->initialized = 1;
    }
  }
  M3 R = imu.R;  // rebvio.cpp:163-165
  R = m3_T(m3_mul(so3_exp(c->Bg), m3_T(R)));
  {
    float RT[9];
    m3_to(m3_T(R), RT);
    orc_rotate_keylines(c, old_map, RT);
  }
  float Vg[3] = {0, 0, 0};
  M3 P_Vg;
  po->F = minimize_vel(c, old_map, Vg, P_Vg, &po->lm_accept_mask, &po->sigma_rho_min);
  forward_match(old_map, new_map);
  float Xv[6], W_Xv[36];
  po->ext_ok = ext_rot_vel(c, Vg, W_Xv, Xv, nullptr);
  float Xgv[6], W_Xgv[36];
  std::memcpy(Xgv, Xv, sizeof(Xv));
  std::memcpy(W_Xgv, W_Xv, sizeof(W_Xv));
  const float frame_dt = float(new_map->ts_us - old_map->ts_us) / 1000000.0;  // rebvio.cpp:183
  const float s_b = P.gyro_bias_std_dev * P.gyro_bias_std_dev * frame_dt * frame_dt;
  const float s_g = P.gyro_std_dev * P.gyro_std_dev * frame_dt * frame_dt;
  c->RGBias = m3_scale(m3_identity(), s_b);
  c->RGyro = m3_scale(m3_identity(), s_g);
  float dg[3];
  gyro_bias_correction(Xgv, W_Xgv, c->W_Bg, c->RGyro, c->RGBias, dg);
  for (int i = 0; i < 3; ++i) c->Bg[i] += dg[i];
  const float dVgv[3] = {Xgv[0], Xgv[1], Xgv[2]}, dWgv[3] = {Xgv[3], Xgv[4], Xgv[5]};
  M3 Rgva = R;  // rebvio.cpp:196 (before the visual correction)
  const M3 R0 = so3_exp(dWgv);
  R = m3_T(m3_mul(R0, m3_T(R)));
  float Vgv[3];
  m3_vec(R0, Vg, Vgv);
  for (int i = 0; i < 3; ++i) Vgv[i] += dVgv[i];
  float V[3] = {Vgv[0], Vgv[1], Vgv[2]};
  float R_Xgv[36];
  cholesky_inverse<6>(W_Xgv, R_Xgv);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      P_V.a[i][j] = R_Xgv[i * 6 + j];
      P_W.a[i][j] = R_Xgv[(3 + i) * 6 + 3 + j];
    }
  // rebvio.cpp:206-208
  {
    float negv[3] = {-Vgv[0] / frame_dt, -Vgv[1] / frame_dt, -Vgv[2] / frame_dt}, Rr[9];
    m3_to(R, Rr);
    orc_estimate_ls4_acceleration(c, negv, c->Av, Rr, frame_dt);
    mean_acceleration(c, imu.cacc, c->As, R);
  }
  float Xgva[6];
  std::memcpy(Xgva, Xgv, sizeof(Xgv));
  {
    const float d4 = frame_dt * frame_dt * frame_dt * frame_dt;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) c->Rv.a[i][j] = P_V.a[i][j] / d4;
  }
  c->Qrot = P_W;
  c->QKp = c->P_Kp;
  float Vgva[3];
  if (c->num_frames > 4u + (unsigned)kInitBiasFrameNum) {  // rebvio.cpp:213-224
    out->sab_active = 1;
    c->Kscale = estimate_bias(c, c->As, c->Av, 1.0f, R, W_Xgv, Xgva);
    const float dVgva[3] = {Xgva[0], Xgva[1], Xgva[2]}, dWgva[3] = {Xgva[3], Xgva[4], Xgva[5]};
    const M3 R0gva = so3_exp(dWgva);
    Rgva = m3_T(m3_mul(R0gva, m3_T(Rgva)));
    m3_vec(R0gva, Vg, Vgva);
    for (int i = 0; i < 3; ++i) Vgva[i] += dVgva[i];
    for (int i = 0; i < 3; ++i) V[i] = Vgva[i];
    float r[9];
    m3_to(R0gva, r);
    orc_rotate_keylines(c, old_map, r);
  } else {  // rebvio.cpp:225-233
    Rgva = R;
    for (int i = 0; i < 3; ++i) Vgva[i] = Vgv[i];
    float r[9];
    m3_to(R0, r);
    orc_rotate_keylines(c, old_map, r);
  }
  for (int i = 0; i < 3; ++i) { po->Vg[i] = Vg[i]; po->V[i] = V[i]; }
  m3_to(P_Vg, po->P_Vg);
  std::memcpy(po->Xv, Xv, sizeof(Xv));
  std::memcpy(po->W_Xv, W_Xv, sizeof(W_Xv));
  std::memcpy(po->Xgv, Xgv, sizeof(Xgv));
  m3_to(Rgva, po->R);
  m3_to(P_V, po->P_V);
  if (std::isnan(V[0]) || std::isnan(V[1]) || std::isnan(V[2])) {  // rebvio.cpp:236-241
    c->P_Kp = FMAX;
    po->status = 1;
  } else {
    po->klm_num = directed_match(c, new_map, old_map, V, P_V, Rgva, &po->kf_matches, P.search_range);
    if ((unsigned)po->klm_num < P.global_min_matches_threshold) {  // rebvio.cpp:247-252
      c->P_Kp = FMAX;
      po->status = 2;
    } else {
      po->reg_num = orc_regularize(new_map);
      orc_update_inverse_depth(c, V);
    }
  }
  if (c->num_frames > 4u + (unsigned)kInitBiasFrameNum) {  // rebvio.cpp:263-271
    float u[3];
    m3_vec(m3_T(Rgva), c->u_est, u);
    const float k = vdot<3>(u, c->g_est) / vdot<3>(c->g_est, c->g_est);
    for (int i = 0; i < 3; ++i) u[i] = u[i] - k * c->g_est[i];
    const float nu = std::sqrt(vdot<3>(u, u));
    for (int i = 0; i < 3; ++i) c->u_est[i] = u[i] / nu;
    const float ey[3] = {0.0f, 1.0f, 0.0f}, ex[3] = {1.0f, 0.0f, 0.0f};
    const M3 R1 = so3_from_two(c->g_est, ey);
    float r1u[3];
    m3_vec(R1, c->u_est, r1u);
    const M3 R2 = so3_from_two(r1u, ex);
    c->R_global = m3_mul(R2, R1);
    float d[3];
    m3_vec(c->R_global, Vgva, d);
    for (int i = 0; i < 3; ++i) c->Pos[i] += -d[i] * c->Kscale;
  }
  so3_ln(c->R_global, out->orientation);
  for (int i = 0; i < 3; ++i) {
    out->position[i] = c->Pos[i];
    out->g_est[i] = c->g_est[i];
    out->b_est[i] = c->b_est[i];
    out->Bg[i] = c->Bg[i];
  }
  out->K = c->Kscale;
  out->initialized = c->initialized;
  ++c->num_frames;
  return po->status;
}
