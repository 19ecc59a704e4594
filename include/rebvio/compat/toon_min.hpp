// Minimal stand-in for the subset of TooN that rebvio's public API exposes (Vector / Matrix value types,
// makeVector, Zeros, Identity, Data, SO3). Used ONLY when <TooN/TooN.h> is not installed (the reference vendors
// TooN as a git submodule that is empty in its repository). With real TooN on the include path these headers
// pick it up instead and this file is not compiled.
#pragma once

#include <cmath>
#include <cstddef>
#include <initializer_list>

namespace TooN {

struct ZerosTag {};
struct IdentityTag {
  double scale = 1.0;
};
static constexpr ZerosTag Zeros{};
static const IdentityTag Identity{};
inline IdentityTag operator*(const IdentityTag& i, double s) { return IdentityTag{i.scale * s}; }

template <int N, typename P = double>
struct Vector {
  P v[N];
  static constexpr int SizeParameter = N;
  Vector() = default;
  Vector(ZerosTag) {
    for (int i = 0; i < N; ++i) v[i] = 0;
  }
  template <typename Q>
  Vector(const Vector<N, Q>& o) {
    for (int i = 0; i < N; ++i) v[i] = static_cast<P>(o.v[i]);
  }
  P& operator[](int i) { return v[i]; }
  const P& operator[](int i) const { return v[i]; }
  int size() const { return N; }
  Vector& operator+=(const Vector& o) {
    for (int i = 0; i < N; ++i) v[i] += o.v[i];
    return *this;
  }
  Vector& operator-=(const Vector& o) {
    for (int i = 0; i < N; ++i) v[i] -= o.v[i];
    return *this;
  }
  Vector& operator/=(P s) {
    for (int i = 0; i < N; ++i) v[i] /= s;
    return *this;
  }
  template <int Start, int Len>
  Vector<Len, P> slice() const {
    Vector<Len, P> r;
    for (int i = 0; i < Len; ++i) r.v[i] = v[Start + i];
    return r;
  }
};

template <int N, typename P>
Vector<N, P> operator+(Vector<N, P> a, const Vector<N, P>& b) { return a += b; }
template <int N, typename P>
Vector<N, P> operator-(Vector<N, P> a, const Vector<N, P>& b) { return a -= b; }
template <int N, typename P>
Vector<N, P> operator-(const Vector<N, P>& a) {
  Vector<N, P> r;
  for (int i = 0; i < N; ++i) r.v[i] = -a.v[i];
  return r;
}
template <int N, typename P>
Vector<N, P> operator*(const Vector<N, P>& a, P s) {
  Vector<N, P> r;
  for (int i = 0; i < N; ++i) r.v[i] = a.v[i] * s;
  return r;
}
template <int N, typename P>
Vector<N, P> operator*(P s, const Vector<N, P>& a) { return a * s; }
template <int N, typename P>
Vector<N, P> operator/(const Vector<N, P>& a, P s) {
  Vector<N, P> r;
  for (int i = 0; i < N; ++i) r.v[i] = a.v[i] / s;
  return r;
}
template <int N, typename P>  // dot product: accumulate from 0 in index order
P operator*(const Vector<N, P>& a, const Vector<N, P>& b) {
  P r = 0;
  for (int i = 0; i < N; ++i) r += a.v[i] * b.v[i];
  return r;
}
template <typename P>
Vector<3, P> operator^(const Vector<3, P>& a, const Vector<3, P>& b) {
  Vector<3, P> r;
  r.v[0] = a.v[1] * b.v[2] - a.v[2] * b.v[1];
  r.v[1] = a.v[2] * b.v[0] - a.v[0] * b.v[2];
  r.v[2] = a.v[0] * b.v[1] - a.v[1] * b.v[0];
  return r;
}
template <int N, typename P>
bool isnan(const Vector<N, P>& a) {
  for (int i = 0; i < N; ++i)
    if (std::isnan(a.v[i])) return true;
  return false;
}

template <typename P>
Vector<2, P> makeVector(P a, P b) {
  Vector<2, P> r;
  r.v[0] = a; r.v[1] = b;
  return r;
}
template <typename P>
Vector<3, P> makeVector(P a, P b, P c) {
  Vector<3, P> r;
  r.v[0] = a; r.v[1] = b; r.v[2] = c;
  return r;
}
inline Vector<3, double> makeVector(double a, double b, double c) {
  Vector<3, double> r;
  r.v[0] = a; r.v[1] = b; r.v[2] = c;
  return r;
}

template <int R, int C = R, typename P = double>
struct Matrix {
  P m[R][C];
  Matrix() = default;
  Matrix(ZerosTag) {
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) m[i][j] = 0;
  }
  Matrix(const IdentityTag& id) {
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) m[i][j] = (i == j) ? static_cast<P>(id.scale) : P(0);
  }
  P& operator()(int r, int c) { return m[r][c]; }
  const P& operator()(int r, int c) const { return m[r][c]; }
  Matrix<C, R, P> T() const {
    Matrix<C, R, P> t;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) t.m[j][i] = m[i][j];
    return t;
  }
};

template <int R, int C, typename P>
Matrix<R, C, P> operator*(const Matrix<R, C, P>& a, P s) {
  Matrix<R, C, P> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) r.m[i][j] = a.m[i][j] * s;
  return r;
}
template <int R, int C, typename P>
Matrix<R, C, P> operator/(const Matrix<R, C, P>& a, P s) {
  Matrix<R, C, P> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) r.m[i][j] = a.m[i][j] / s;
  return r;
}
template <int R, int C, typename P>
Matrix<R, C, P> operator+(const Matrix<R, C, P>& a, const Matrix<R, C, P>& b) {
  Matrix<R, C, P> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
  return r;
}
template <int R, int K, int C, typename P>
Matrix<R, C, P> operator*(const Matrix<R, K, P>& a, const Matrix<K, C, P>& b) {
  Matrix<R, C, P> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      P s = 0;
      for (int k = 0; k < K; ++k) s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
template <int R, int C, typename P>
Vector<R, P> operator*(const Matrix<R, C, P>& a, const Vector<C, P>& v) {
  Vector<R, P> r;
  for (int i = 0; i < R; ++i) {
    P s = 0;
    for (int k = 0; k < C; ++k) s += a.m[i][k] * v.v[k];
    r.v[i] = s;
  }
  return r;
}

// TooN::Data(a, b, ...) fills a matrix row-major
struct DataFill {
  double d[49];
  int n;
  template <int R, int C, typename P>
  operator Matrix<R, C, P>() const {
    Matrix<R, C, P> m;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) m.m[i][j] = static_cast<P>(d[i * C + j]);
    return m;
  }
};
template <typename... A>
DataFill Data(A... a) {
  DataFill f{{static_cast<double>(a)...}, static_cast<int>(sizeof...(A))};
  return f;
}

// SO3: Rodrigues exponential / logarithm
template <typename P = double>
class SO3 {
 public:
  SO3() : R_(Identity) {}
  explicit SO3(const Vector<3, P>& w) { R_ = exp(w); }
  explicit SO3(const Matrix<3, 3, P>& R) : R_(R) {}
  // minimal rotation taking a to b
  SO3(const Vector<3, P>& a, const Vector<3, P>& b) {
    Vector<3, P> n = a ^ b;
    if (n * n == 0) {
      R_ = Matrix<3, 3, P>(Identity);
      return;
    }
    n = n / std::sqrt(n * n);
    const Vector<3, P> ua = a / std::sqrt(a * a), ub = b / std::sqrt(b * b);
    const Vector<3, P> c1 = n ^ ua, c2 = n ^ ub;
    Matrix<3, 3, P> R1, Rm;
    for (int i = 0; i < 3; ++i) {
      R1(i, 0) = ua[i]; R1(i, 1) = n[i]; R1(i, 2) = c1[i];
      Rm(i, 0) = ub[i]; Rm(i, 1) = n[i]; Rm(i, 2) = c2[i];
    }
    R_ = Rm * R1.T();
  }
  static Matrix<3, 3, P> exp(const Vector<3, P>& w) {
    const P one_6th = P(1.0 / 6.0), one_20th = P(1.0 / 20.0);
    P tsq = w * w, A, B;
    if (tsq < 1e-8) {
      A = P(1.0 - one_6th * tsq);
      B = P(0.5);
    } else if (tsq < 1e-6) {
      B = P(0.5 - 0.25 * one_6th * tsq);
      A = P(1.0 - tsq * one_6th * (1.0 - one_20th * tsq));
    } else {
      const P th = std::sqrt(tsq), inv = P(1.0 / th);
      A = std::sin(th) * inv;
      B = (1 - std::cos(th)) * (inv * inv);
    }
    Matrix<3, 3, P> R;
    const P wx2 = w[0] * w[0], wy2 = w[1] * w[1], wz2 = w[2] * w[2];
    R(0, 0) = P(1.0 - B * (wy2 + wz2));
    R(1, 1) = P(1.0 - B * (wx2 + wz2));
    R(2, 2) = P(1.0 - B * (wx2 + wy2));
    P a = A * w[2], b = B * (w[0] * w[1]);
    R(0, 1) = b - a; R(1, 0) = b + a;
    a = A * w[1]; b = B * (w[0] * w[2]);
    R(0, 2) = b + a; R(2, 0) = b - a;
    a = A * w[0]; b = B * (w[1] * w[2]);
    R(1, 2) = b - a; R(2, 1) = b + a;
    return R;
  }
  const Matrix<3, 3, P>& get_matrix() const { return R_; }
  Vector<3, P> operator*(const Vector<3, P>& v) const { return R_ * v; }
  Vector<3, P> ln() const {
    Vector<3, P> r;
    const P cos_angle = (R_(0, 0) + R_(1, 1) + R_(2, 2) - P(1.0)) * P(0.5);
    r[0] = (R_(2, 1) - R_(1, 2)) / 2;
    r[1] = (R_(0, 2) - R_(2, 0)) / 2;
    r[2] = (R_(1, 0) - R_(0, 1)) / 2;
    const P sin_abs = std::sqrt(r * r);
    if (cos_angle > P(M_SQRT1_2)) {
      if (sin_abs > 0) r = r * (std::asin(sin_abs) / sin_abs);
    } else if (cos_angle > -P(M_SQRT1_2)) {
      if (sin_abs > 0) r = r * (std::acos(cos_angle) / sin_abs);
    } else {
      const P angle = P(M_PI) - std::asin(sin_abs);
      const P d0 = R_(0, 0) - cos_angle, d1 = R_(1, 1) - cos_angle, d2 = R_(2, 2) - cos_angle;
      Vector<3, P> r2;
      if (d0 * d0 > d1 * d1 && d0 * d0 > d2 * d2) {
        r2[0] = d0; r2[1] = (R_(1, 0) + R_(0, 1)) / 2; r2[2] = (R_(0, 2) + R_(2, 0)) / 2;
      } else if (d1 * d1 > d2 * d2) {
        r2[0] = (R_(1, 0) + R_(0, 1)) / 2; r2[1] = d1; r2[2] = (R_(2, 1) + R_(1, 2)) / 2;
      } else {
        r2[0] = (R_(0, 2) + R_(2, 0)) / 2; r2[1] = (R_(2, 1) + R_(1, 2)) / 2; r2[2] = d2;
      }
      if (r2 * r < 0) r2 = r2 * P(-1);
      const P nrm = std::sqrt(r2 * r2);
      r = r2 * (angle / nrm);
    }
    return r;
  }

 private:
  Matrix<3, 3, P> R_;
};

}  // namespace TooN
