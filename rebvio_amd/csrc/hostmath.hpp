// Host-side O(1) math of the frame-pair glue (3x3 / 6x6): stays on the CPU exactly like the reference keeps
// it in TooN on the tracking thread (rebvio.cpp:163-233, core.cpp:244-248,264-284). fp32 with TooN's evaluation
// order (dot products accumulate from 0 in index order); TooN itself is an absent submodule, so its published
// algorithms are restated: determinant by pivoted elimination, LDL^T Cholesky inverse, Rodrigues SO3::exp.
// SVD<6>::backsub (LAPACK gesvd underneath) is replaced by a cyclic-Jacobi eigen solve with TooN's 1e9
// condition cut.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

// The 3x3 / 6x6 functions below also run on the device (the streaming and batch drivers keep the glue of a pair on the GPU,
// glue.hpp): one source, the same operation order on both sides. RH_HD is empty for plain C++ translation units
// (rebvio_amd/host/rebvio.cpp uses cholesky6_inverse).
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RH_HD __host__ __device__
#else
#define RH_HD
#endif

namespace rh {
namespace hm {

struct M3 {
  float a[3][3];
};

RH_HD inline M3 identity3() {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = (i == j) ? 1.0f : 0.0f;
  return r;
}
RH_HD inline M3 load3(const float* p) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = p[i * 3 + j];
  return r;
}
RH_HD inline void store3(const M3& m, float* p) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) p[i * 3 + j] = m.a[i][j];
}
RH_HD inline M3 transpose(const M3& m) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = m.a[j][i];
  return r;
}
RH_HD inline M3 mul(const M3& x, const M3& y) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 3; ++k) s += x.a[i][k] * y.a[k][j];
      r.a[i][j] = s;
    }
  return r;
}
RH_HD inline void mulv(const M3& m, const float v[3], float out[3]) {
  float t[3];
  for (int i = 0; i < 3; ++i) {
    float s = 0;
    for (int k = 0; k < 3; ++k) s += m.a[i][k] * v[k];
    t[i] = s;
  }
  out[0] = t[0]; out[1] = t[1]; out[2] = t[2];
}
RH_HD inline M3 add(const M3& x, const M3& y) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][j] + y.a[i][j];
  return r;
}
RH_HD inline M3 sub(const M3& x, const M3& y) {
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.a[i][j] = x.a[i][j] - y.a[i][j];
  return r;
}
RH_HD inline M3 diag3(float v) {
  M3 r = identity3();
  for (int i = 0; i < 3; ++i) r.a[i][i] = v;
  return r;
}

RH_HD inline float det3(const M3& m) {
  float A[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A[i][j] = m.a[i][j];
  float det = 1;
  for (int i = 0; i < 3; ++i) {
    int arg = i;
    float mx = std::fabs(A[i][i]);
    for (int ii = i + 1; ii < 3; ++ii)
      if (std::fabs(A[ii][i]) > mx) {
        mx = std::fabs(A[ii][i]);
        arg = ii;
      }
    const float pivot = A[arg][i];
    if (arg != i) {
      det *= -1;
      for (int ii = i; ii < 3; ++ii) {
        const float t = A[i][ii];
        A[i][ii] = A[arg][ii];
        A[arg][ii] = t;
      }
    }
    det *= A[i][i];
    if (det == 0) return 0;
    for (int u = i + 1; u < 3; ++u) {
      const float factor = A[u][i] / pivot;
      for (int uu = i; uu < 3; ++uu) A[u][uu] = A[u][uu] - factor * A[i][uu];
    }
  }
  return det;
}

// types::invert (types/definitions.hpp:40-53)
RH_HD inline M3 invert3(const M3& in) {
  const float(*m)[3] = in.a;
  M3 o;
  o.a[0][0] = m[1][1] * m[2][2] - m[1][2] * m[2][1];
  o.a[0][1] = m[0][2] * m[2][1] - m[0][1] * m[2][2];
  o.a[0][2] = m[0][1] * m[1][2] - m[0][2] * m[1][1];
  o.a[1][0] = m[1][2] * m[2][0] - m[1][0] * m[2][2];
  o.a[1][1] = m[0][0] * m[2][2] - m[0][2] * m[2][0];
  o.a[1][2] = m[0][2] * m[1][0] - m[0][0] * m[1][2];
  o.a[2][0] = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  o.a[2][1] = m[0][1] * m[2][0] - m[0][0] * m[2][1];
  o.a[2][2] = m[0][0] * m[1][1] - m[0][1] * m[1][0];
  const float d = det3(in);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) o.a[i][j] = o.a[i][j] / d;
  return o;
}

// sin / cos of a float argument, rounded from a double evaluation that uses +, -, * only (Cody-Waite reduction by pi/2,
// Taylor polynomials on [-pi/4, pi/4], Horner form, no fused operations): the SAME bits on the host and on the device, where
// libm's sinf and the device library's differ in the last place now and then. The double result is within ~2e-16 of the true
// value, so the float is the correctly rounded one except on ~1e-9 of the arguments. Arguments beyond 1e5 (meaningless as
// an inter-frame rotation) go to the platform's double sin / cos.
RH_HD inline void sincos_det(float xf, float* s_out, float* c_out) {
  const double x = (double)xf;
  if (!(std::fabs(x) < 1.0e5)) {
    *s_out = (float)std::sin(x);
    *c_out = (float)std::cos(x);
    return;
  }
  const double two_over_pi = 0.63661977236758134308;
  const double pio2_hi = 1.57079632673412561417e+00, pio2_lo = 6.07710050650619224932e-11;  // 33 bits of pi/2 (k * hi exact) + the rest
  const double kd = std::nearbyint(x * two_over_pi);
  const long long k = (long long)kd;
  double r = x - kd * pio2_hi;
  r = r - kd * pio2_lo;
  const double r2 = r * r;
  // sin r = r (1 - r2/3! + r2^2/5! - ... - r2^8/17!)
  double ps = -1.0 / 355687428096000.0;
  ps = ps * r2 + 1.0 / 1307674368000.0;
  ps = ps * r2 - 1.0 / 6227020800.0;
  ps = ps * r2 + 1.0 / 39916800.0;
  ps = ps * r2 - 1.0 / 362880.0;
  ps = ps * r2 + 1.0 / 5040.0;
  ps = ps * r2 - 1.0 / 120.0;
  ps = ps * r2 + 1.0 / 6.0;
  const double sr = r - r * (r2 * ps);
  // cos r = 1 - r2/2! + r2^2/4! - ... + r2^9/18!
  double pc = -1.0 / 6402373705728000.0;
  pc = pc * r2 + 1.0 / 20922789888000.0;
  pc = pc * r2 - 1.0 / 87178291200.0;
  pc = pc * r2 + 1.0 / 479001600.0;
  pc = pc * r2 - 1.0 / 3628800.0;
  pc = pc * r2 + 1.0 / 40320.0;
  pc = pc * r2 - 1.0 / 720.0;
  pc = pc * r2 + 1.0 / 24.0;
  pc = pc * r2 - 1.0 / 2.0;
  const double cr = 1.0 + r2 * pc;
  double sv, cv;
  switch ((int)(k & 3)) {
    case 0: sv = sr; cv = cr; break;
    case 1: sv = cr; cv = -sr; break;
    case 2: sv = -sr; cv = -cr; break;
    default: sv = -cr; cv = sr; break;
  }
  *s_out = (float)sv;
  *c_out = (float)cv;
}

// TooN SO3<float>::exp
RH_HD inline M3 so3_exp(const float w[3]) {
  const float one_6th = 1.0 / 6.0, one_20th = 1.0 / 20.0;
  float tsq = 0;
  for (int i = 0; i < 3; ++i) tsq += w[i] * w[i];
  float A, B;
  if (tsq < 1e-8) {
    A = 1.0 - one_6th * tsq;
    B = 0.5;
  } else if (tsq < 1e-6) {
    B = 0.5 - 0.25 * one_6th * tsq;
    A = 1.0 - tsq * one_6th * (1.0 - one_20th * tsq);
  } else {
    const float th = std::sqrt(tsq), inv = 1.0 / th;
    float sn, cs;
    sincos_det(th, &sn, &cs);  // TooN: sin(theta), cos(theta) of a float (libm); see sincos_det
    A = sn * inv;
    B = (1 - cs) * (inv * inv);
  }
  M3 R;
  const float wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
  R.a[0][0] = 1.0 - B * (wy2 + wz2);
  R.a[1][1] = 1.0 - B * (wx2 + wz2);
  R.a[2][2] = 1.0 - B * (wx2 + wy2);
  float a = A * w[2], b = B * (w[0] * w[1]);
  R.a[0][1] = b - a; R.a[1][0] = b + a;
  a = A * w[1]; b = B * (w[0] * w[2]);
  R.a[0][2] = b + a; R.a[2][0] = b - a;
  a = A * w[0]; b = B * (w[1] * w[2]);
  R.a[1][2] = b - a; R.a[2][1] = b + a;
  return R;
}

// TooN Cholesky<6,float>::get_inverse (LDL^T), in two steps so that the device can give every column of the inverse to a
// lane of its own (each lane factorises for itself: the same operations either way).
RH_HD inline void cholesky6_factor(const float* A, float L[6][6]) {
  constexpr int N = 6;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) L[i][j] = A[i * N + j];
  for (int col = 0; col < N; ++col) {
    float inv_diag = 1;
    bool stop = false;  // TooN leaves the column at a zero pivot (`break`): a flag keeps the loop bounds static for the device
    for (int row = col; row < N; ++row) {
      if (stop) continue;
      float val = L[row][col];
      for (int c2 = 0; c2 < col; ++c2) val -= L[c2][col] * L[row][c2];
      if (row == col) {
        L[row][col] = val;
        if (val == 0)
          stop = true;
        else
          inv_diag = 1 / val;
      } else {
        L[col][row] = val;
        L[row][col] = val * inv_diag;
      }
    }
  }
}
// column c of the inverse: solve L D L^T x = e_c
RH_HD inline void cholesky6_inverse_col(const float L[6][6], int c, float res[6]) {
  constexpr int N = 6;
  float y[N];
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
}
RH_HD inline void cholesky6_inverse(const float* A, float* inv) {
  constexpr int N = 6;
  float L[N][N];
  cholesky6_factor(A, L);
  for (int c = 0; c < N; ++c) {
    float res[N];
    cholesky6_inverse_col(L, c, res);
    for (int i = 0; i < N; ++i) inv[i * N + c] = res[i];
  }
}

// x = pinv(A) b for symmetric 6x6 A (stands in for SVD<6,6,float>(A).backsub(b), core.cpp:247-248): cyclic Jacobi with
// TooN's 1e9 condition cut. ws = 72 doubles of workspace (on the device: LDS; its loops index dynamically, which a register
// array would turn into scratch memory).
RH_HD inline void sym6_pinv_solve_ws(const float* A_, const float* b_, float* x_, double* ws) {
  constexpr int N = 6;
  double* A = ws;       // [N][N]
  double* V = ws + 36;  // [N][N]
  bool bad = false;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) {
      if (A_[i * N + j] != A_[i * N + j]) bad = true;
      A[i * N + j] = 0.5 * ((double)A_[i * N + j] + (double)A_[j * N + i]);
      V[i * N + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 60 && !bad; ++sweep) {
    double off = 0;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) off += A[p * N + q] * A[p * N + q];
    if (off < 1e-300) break;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        if (A[p * N + q] == 0.0) continue;
        const double theta = (A[q * N + q] - A[p * N + p]) / (2.0 * A[p * N + q]);
        const double t = ((theta >= 0) ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < N; ++k) {
          const double akp = A[k * N + p], akq = A[k * N + q];
          A[k * N + p] = cs * akp - sn * akq;
          A[k * N + q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < N; ++k) {
          const double apk = A[p * N + k], aqk = A[q * N + k];
          A[p * N + k] = cs * apk - sn * aqk;
          A[q * N + k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < N; ++k) {
          const double vkp = V[k * N + p], vkq = V[k * N + q];
          V[k * N + p] = cs * vkp - sn * vkq;
          V[k * N + q] = sn * vkp + cs * vkq;
        }
      }
  }
  double dmax = 0, x[N] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < N; ++i) dmax = (std::fabs(A[i * N + i]) > dmax) ? std::fabs(A[i * N + i]) : dmax;
  for (int k = 0; k < N && !bad; ++k) {
    const double lam = A[k * N + k];
    if (!(std::fabs(lam) * 1e9 > dmax)) continue;
    double proj = 0;
    for (int i = 0; i < N; ++i) proj += V[i * N + k] * (double)b_[i];
    proj /= lam;
    for (int i = 0; i < N; ++i) x[i] += V[i * N + k] * proj;
  }
  for (int i = 0; i < N; ++i) x_[i] = bad ? std::numeric_limits<float>::quiet_NaN() : (float)x[i];
}
inline void sym6_pinv_solve(const float* A_, const float* b_, float* x_) {
  double ws[72];
  sym6_pinv_solve_ws(A_, b_, x_, ws);
}

// SVD<6>::backsub stand-in used by the pipeline: for a well-conditioned SPD JtJ (the normal case) the pseudo-inverse
// IS the inverse, so solve by an LDL^T factorisation in double (~300 flops); if a pivot falls below the 1e9 condition
// cut relative to the largest diagonal, use the Jacobi pseudo-inverse (minimum-norm solution like the SVD).
// sym6_ldlt_solve returns false (x untouched) when the fall-back is needed.
RH_HD inline bool sym6_ldlt_solve(const float* A_, const float* b_, float* x_) {
  constexpr int N = 6;
  double L[N][N], d[N], dmax = 0;
  bool ok = true;
  for (int i = 0; i < N; ++i) dmax = (std::fabs((double)A_[i * N + i]) > dmax) ? std::fabs((double)A_[i * N + i]) : dmax;
  for (int j = 0; j < N; ++j) {  // (loop bounds static, `ok` tested inside: the device keeps L and d in registers)
    if (!ok) continue;
    double v = 0.5 * ((double)A_[j * N + j] + (double)A_[j * N + j]);
    for (int k = 0; k < j; ++k) v -= L[j][k] * L[j][k] * d[k];
    if (!(v * 1e7 > dmax)) ok = false;
    d[j] = v;
    for (int i = j + 1; i < N; ++i) {
      if (!ok) continue;
      double s = 0.5 * ((double)A_[i * N + j] + (double)A_[j * N + i]);
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k] * d[k];
      L[i][j] = s / v;
    }
  }
  if (!ok) return false;
  double y[N], x[N];
  for (int i = 0; i < N; ++i) {
    double s = (double)b_[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
    y[i] = s;
  }
  for (int i = 0; i < N; ++i) y[i] /= d[i];
  for (int i = N - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < N; ++k) s -= L[k][i] * x[k];
    x[i] = s;
  }
  for (int i = 0; i < N; ++i) x_[i] = (float)x[i];
  return true;
}
RH_HD inline void sym6_solve_ws(const float* A_, const float* b_, float* x_, double* ws72) {
  if (!sym6_ldlt_solve(A_, b_, x_)) sym6_pinv_solve_ws(A_, b_, x_, ws72);
}
inline void sym6_solve(const float* A_, const float* b_, float* x_) {
  double ws[72];
  sym6_solve_ws(A_, b_, x_, ws);
}

// The data-independent part of gyroBiasCorrection below (same statements): pre[0] Wg, [1] Wb' , [2] Wg + Wb' (the information
// matrix after the pair), [3] iWgWb, [4] upd, [5] (Wg iWgWb) Wb' - see GlueParams.
RH_HD inline void gyro_pre(const M3& W_Bg, float s_g, float s_b, float pre[6][9]) {
  const M3 Wg = invert3(diag3(s_g));
  const M3 Wb = invert3(add(invert3(W_Bg), diag3(s_b)));
  const M3 iWgWb = invert3(add(Wg, Wb));
  const M3 upd = mul(Wg, sub(identity3(), mul(iWgWb, Wg)));
  store3(Wg, pre[0]);
  store3(Wb, pre[1]);
  store3(add(Wg, Wb), pre[2]);
  store3(iWgWb, pre[3]);
  store3(upd, pre[4]);
  store3(mul(mul(Wg, iWgWb), Wb), pre[5]);
}

// Core::gyroBiasCorrection (core.cpp:264-284); dgbias is zero on entry, as in the reference.
RH_HD inline void gyro_bias_correction(float X[6], float Wx[36], M3& Wb, const M3& Rg, const M3& Rb, float dgbias_out[3]) {
  float dgbias[3] = {0, 0, 0};
  const M3 Wg = invert3(Rg);
  Wb = invert3(add(invert3(Wb), Rb));
  float Wxb[36];
  for (int i = 0; i < 36; ++i) Wxb[i] = Wx[i];
  const M3 iWgWb = invert3(add(Wg, Wb));
  const M3 upd = mul(Wg, sub(identity3(), mul(iWgWb, Wg)));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Wxb[(3 + i) * 6 + 3 + j] += upd.a[i][j];
  float X1[6];
  for (int i = 0; i < 6; ++i) {
    float s = 0;
    for (int k = 0; k < 6; ++k) s += Wx[i * 6 + k] * X[k];
    X1[i] = s;
  }
  {
    float t[3];
    mulv(mul(mul(Wg, iWgWb), Wb), dgbias, t);
    for (int i = 0; i < 3; ++i) X1[3 + i] += t[i];
  }
  float inv[36];
  cholesky6_inverse(Wxb, inv);
  for (int i = 0; i < 6; ++i) {
    float s = 0;
    for (int k = 0; k < 6; ++k) s += inv[i * 6 + k] * X1[k];
    X[i] = s;
  }
  {
    float a[3], b[3], sum[3];
    const float xw[3] = {X[3], X[4], X[5]};
    mulv(Wg, xw, a);
    mulv(Wb, dgbias, b);
    for (int i = 0; i < 3; ++i) sum[i] = a[i] + b[i];
    mulv(iWgWb, sum, dgbias);
  }
  Wb = add(Wg, Wb);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Wx[(3 + i) * 6 + 3 + j] += Wg.a[i][j];
  for (int i = 0; i < 3; ++i) dgbias_out[i] = dgbias[i];
}

// Kovesi box widths of FastGaussian::FastGaussian (scale_space.cpp:19-35)
// Fixed-point source coordinates of cv::undistort(src, dst, K, D) with K = (fx,0,cx; 0,fy,cy), D = (k1,k2,p1,p2,k3), as
// the reference calls it (camera.hpp:39-40,54-58): OpenCV builds CV_16SC2 maps in double, stripe by stripe
// (stripe = max(1, 4096/cols) rows, principal point shifted by the stripe's first row), u and v quantised to 1/32 pixel
// with round-half-even; remap then interpolates bilinearly with constant-zero border. out[2*i] = iu, out[2*i+1] = iv.
inline void undistort_fixed_map(int rows, int cols, double fx, double fy, double cx, double cy, double k1, double k2, double p1,
                                double p2, double k3, int* out) {
  const int stripe = std::min(std::max(1, (1 << 12) / std::max(cols, 1)), rows);
  const double ir0 = 1.0 / fx, ir4 = 1.0 / fy, ir2 = -cx / fx;
  for (int y0 = 0; y0 < rows; y0 += stripe) {
    const int h = std::min(stripe, rows - y0);
    const double ir5 = -(cy - y0) / fy;
    for (int i = 0; i < h; ++i) {
      double x_ = ir2;
      const double y = i * ir4 + ir5;
      int* o = out + (size_t)(y0 + i) * cols * 2;
      for (int j = 0; j < cols; ++j, x_ += ir0) {
        const double x = x_;
        const double x2 = x * x, y2 = y * y;
        const double r2 = x2 + y2, xy2 = 2 * x * y;
        const double kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2;
        const double xd = x * kr + p1 * xy2 + p2 * (r2 + 2 * x2);
        const double yd = y * kr + p1 * (r2 + 2 * y2) + p2 * xy2;
        const double u = fx * xd + cx, v = fy * yd + cy;
        o[2 * j] = (int)std::nearbyint(u * 32.0);
        o[2 * j + 1] = (int)std::nearbyint(v * 32.0);
      }
    }
  }
}

inline void kovesi_widths(float sigma, int n, int* widths, float* sigma_true) {
  const float w_ideal = std::sqrt(12.0 * sigma * sigma / float(n + 1));
  int w_l = int(w_ideal);
  if (int(w_l / 2) * 2 == w_l) --w_l;
  const int m = std::round((3 * n + 4 * n * w_l + n * w_l * w_l - 12 * sigma * sigma) / (4 + 4 * w_l));
  int i = 0;
  for (; i < m; i++) widths[i] = w_l;
  for (; i < n; i++) widths[i] = w_l + 2;
  *sigma_true = std::sqrt((m * w_l * w_l + (n - m) * (w_l + 2.0) * (w_l + 2.0) - n) / 12.0);
}

// Pinv = invert(Phi^T Phi) Phi^T for the 5x5 plane fit (edge_detector.cpp:55-68)
inline void plane_fit_pinv(float out[75]) {
  float Phi[25][3];
  for (int row = -2, k = 0; row <= 2; ++row)
    for (int col = -2; col <= 2; ++col, ++k) {
      Phi[k][0] = col; Phi[k][1] = row; Phi[k][2] = 1;
    }
  M3 PtP;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0;
      for (int k = 0; k < 25; ++k) s += Phi[k][i] * Phi[k][j];
      PtP.a[i][j] = s;
    }
  const M3 inv = invert3(PtP);
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 25; ++k) {
      float s = 0;
      for (int j = 0; j < 3; ++j) s += inv.a[i][j] * Phi[k][j];
      out[i * 25 + k] = s;
    }
}

}  // namespace hm
}  // namespace rh
