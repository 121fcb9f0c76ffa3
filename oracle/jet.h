// oracle/jet.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never linked into the product.
//
// Forward-mode dual number of fixed width 4 = what the reference differentiates with:
//   `using Jet = ceres::Jet<double, kStride>` with `kStride = 4` (reference lib/ValueTransform.h:31-34),
// evaluated in passes of 4 parameters by ceres::DynamicAutoDiffCostFunction<F, 4>
// (reference lib/PoseOptimizer.cpp:1198).  Ceres itself is not vendored in /root/reference, so the
// operator semantics below restate the published ceres/jet.h rules (sum/product/quotient rules;
// comparisons look at the scalar part only; abs() flips the sign of value and derivative).
#pragma once
#include <cmath>

namespace cvdo {

constexpr int kStride = 4;

struct Jet {
  double a;
  double v[kStride];
  Jet() : a(0.0), v{0.0, 0.0, 0.0, 0.0} {}
  Jet(double x) : a(x), v{0.0, 0.0, 0.0, 0.0} {}  // NOLINT: implicit like ceres::Jet(T)
};

inline Jet operator-(const Jet& f) {
  Jet r;
  r.a = -f.a;
  for (int i = 0; i < kStride; ++i) r.v[i] = -f.v[i];
  return r;
}
inline Jet operator+(const Jet& f, const Jet& g) {
  Jet r;
  r.a = f.a + g.a;
  for (int i = 0; i < kStride; ++i) r.v[i] = f.v[i] + g.v[i];
  return r;
}
inline Jet operator-(const Jet& f, const Jet& g) {
  Jet r;
  r.a = f.a - g.a;
  for (int i = 0; i < kStride; ++i) r.v[i] = f.v[i] - g.v[i];
  return r;
}
inline Jet operator*(const Jet& f, const Jet& g) {
  Jet r;
  r.a = f.a * g.a;
  for (int i = 0; i < kStride; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return r;
}
inline Jet operator/(const Jet& f, const Jet& g) {
  // ceres/jet.h: a/b with g_a_inverse = 1/g.a; (f.v - f.a/g.a * g.v) * g_a_inverse
  Jet r;
  const double gi = 1.0 / g.a;
  const double q = f.a * gi;
  r.a = q;
  for (int i = 0; i < kStride; ++i) r.v[i] = (f.v[i] - q * g.v[i]) * gi;
  return r;
}
inline Jet operator+(const Jet& f, double s) { Jet r = f; r.a += s; return r; }
inline Jet operator+(double s, const Jet& f) { Jet r = f; r.a += s; return r; }
inline Jet operator-(const Jet& f, double s) { Jet r = f; r.a -= s; return r; }
inline Jet operator-(double s, const Jet& f) { Jet r = -f; r.a += s; return r; }
inline Jet operator*(const Jet& f, double s) {
  Jet r;
  r.a = f.a * s;
  for (int i = 0; i < kStride; ++i) r.v[i] = f.v[i] * s;
  return r;
}
inline Jet operator*(double s, const Jet& f) { return f * s; }
inline Jet operator/(const Jet& f, double s) {
  const double si = 1.0 / s;
  return f * si;
}
inline Jet operator/(double s, const Jet& g) {
  Jet r;
  const double gi = 1.0 / g.a;
  r.a = s * gi;
  const double m = -s * gi * gi;
  for (int i = 0; i < kStride; ++i) r.v[i] = m * g.v[i];
  return r;
}
inline Jet& operator+=(Jet& f, const Jet& g) { f = f + g; return f; }
inline Jet& operator-=(Jet& f, const Jet& g) { f = f - g; return f; }
inline Jet& operator*=(Jet& f, const Jet& g) { f = f * g; return f; }
inline Jet& operator*=(Jet& f, double s) { f = f * s; return f; }

inline bool operator<(const Jet& f, const Jet& g) { return f.a < g.a; }
inline bool operator>(const Jet& f, const Jet& g) { return f.a > g.a; }
inline bool operator<=(const Jet& f, const Jet& g) { return f.a <= g.a; }
inline bool operator>=(const Jet& f, const Jet& g) { return f.a >= g.a; }

inline Jet jsqrt(const Jet& f) {
  Jet r;
  r.a = std::sqrt(f.a);
  const double m = 1.0 / (2.0 * r.a);
  for (int i = 0; i < kStride; ++i) r.v[i] = m * f.v[i];
  return r;
}
inline Jet jsin(const Jet& f) {
  Jet r;
  r.a = std::sin(f.a);
  const double m = std::cos(f.a);
  for (int i = 0; i < kStride; ++i) r.v[i] = m * f.v[i];
  return r;
}
inline Jet jcos(const Jet& f) {
  Jet r;
  r.a = std::cos(f.a);
  const double m = -std::sin(f.a);
  for (int i = 0; i < kStride; ++i) r.v[i] = m * f.v[i];
  return r;
}
inline Jet jlog(const Jet& f) {
  Jet r;
  r.a = std::log(f.a);
  const double m = 1.0 / f.a;
  for (int i = 0; i < kStride; ++i) r.v[i] = m * f.v[i];
  return r;
}
inline Jet jabs(const Jet& f) { return f.a < 0.0 ? -f : f; }

// Generic spellings so that the functors can be written once for T in {double, Jet}.
inline double tsqrt(double x) { return std::sqrt(x); }
inline Jet tsqrt(const Jet& x) { return jsqrt(x); }
inline double tsin(double x) { return std::sin(x); }
inline Jet tsin(const Jet& x) { return jsin(x); }
inline double tcos(double x) { return std::cos(x); }
inline Jet tcos(const Jet& x) { return jcos(x); }
inline double tlog(double x) { return std::log(x); }
inline Jet tlog(const Jet& x) { return jlog(x); }
inline double tabs(double x) { return std::abs(x); }
inline Jet tabs(const Jet& x) { return jabs(x); }
// std::max / std::min semantics (what `max(a, b)` resolves to in the reference through
// `using namespace cv` -> `using std::max`): ties return the FIRST operand, value and derivative of
// the selected operand are propagated.
template <typename T>
inline T tmax(const T& x, const T& y) { return (x < y) ? y : x; }
template <typename T>
inline T tmin(const T& x, const T& y) { return (y < x) ? y : x; }

inline double scalar(double x) { return x; }
inline double scalar(const Jet& x) { return x.a; }

}  // namespace cvdo
