// TEST INFRASTRUCTURE ONLY -- CPU oracle for the NID registration hot path.
// Nothing under oracle/ may be imported, linked or executed by the product path
// (direct_visual_lidar_calibration_amd/); only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg use it, and only as the checker.
//
// PARITY: the reference (koide3/direct_visual_lidar_calibration @ 2025-05-23) ships no tests, golden
// vectors or fixtures for this path, and its own build cannot run here (cmake + ROS + Eigen / Ceres /
// Sophus / OpenCV / GTSAM / PCL, none installed, none vendored).  What pins this restatement:
//   (1) oracle/_ref/libref.so (`make -C oracle ref`): the reference's OWN hot-path source files compiled
//       unmodified, in place, against stand-in third-party headers (oracle/shim/); the restatement equals
//       it bit for bit (tests/test_reference_build.py) and its outputs are committed as fixtures
//       (tests/golden/reference_cases.npz, tests/test_reference_golden.py);
//   (2) an independent second oracle (tests/pyoracle.py: numpy + torch.autograd) and finite differences.
// STILL UNPINNED: the arithmetic of the absent third-party libraries themselves (ceres::Jet chain rules,
// Sophus' point action, Eigen's reduction order, OpenCV's casts), restated from their published
// definitions in this file and in oracle/shim/ -- see DESIGN.md section 5.
//
// jet.hpp -- forward-mode dual number with N partials.  Restates the arithmetic the
// reference gets from ceres::Jet<double, 7> (Ceres Solver @ e47a42c2, un-vendored
// third-party dependency; used at nid_cost.hpp:36-107, generic_camera_base.hpp:40 and in
// every camera functor).  Formulas follow the published ceres/jet.h definitions:
//   f*g  = (f.a g.a,  f.a g.v + f.v g.a)          f/g = (f.a/g.a, (f.v - f.a/g.a g.v)/g.a)
//   sqrt = (sqrt a, v / (2 sqrt a))               log = (log a, v / a)
//   atan2(g,f) = (atan2(g.a,f.a), (-g.a f.v + f.a g.v)/(f.a^2+g.a^2))
//   asin = (asin a, v / sqrt(1-a^2))   atan = (atan a, v/(1+a^2))   tan = (tan a, v (1+tan^2 a))
//   pow(f,p) = (a^p, p a^(p-1) v)                 abs = (|a|, copysign(1,a) v)
#pragma once
#include <cmath>

namespace oracle {

template <int N>
struct Jet {
  double a;
  double v[N];

  Jet() : a(0.0) {
    for (int i = 0; i < N; i++) v[i] = 0.0;
  }
  Jet(double value) : a(value) {  // NOLINT (implicit on purpose, like ceres::Jet)
    for (int i = 0; i < N; i++) v[i] = 0.0;
  }
  Jet(double value, int k) : a(value) {
    for (int i = 0; i < N; i++) v[i] = 0.0;
    v[k] = 1.0;
  }

  Jet& operator+=(const Jet& y) {
    a += y.a;
    for (int i = 0; i < N; i++) v[i] += y.v[i];
    return *this;
  }
  Jet& operator-=(const Jet& y) {
    a -= y.a;
    for (int i = 0; i < N; i++) v[i] -= y.v[i];
    return *this;
  }
  Jet& operator*=(const Jet& y) {
    *this = *this * y;
    return *this;
  }
  Jet& operator/=(const Jet& y) {
    *this = *this / y;
    return *this;
  }
};

template <int N>
inline Jet<N> operator-(const Jet<N>& f) {
  Jet<N> r;
  r.a = -f.a;
  for (int i = 0; i < N; i++) r.v[i] = -f.v[i];
  return r;
}
template <int N>
inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> r;
  r.a = f.a + g.a;
  for (int i = 0; i < N; i++) r.v[i] = f.v[i] + g.v[i];
  return r;
}
template <int N>
inline Jet<N> operator+(const Jet<N>& f, double s) {
  Jet<N> r = f;
  r.a = f.a + s;
  return r;
}
template <int N>
inline Jet<N> operator+(double s, const Jet<N>& f) {
  Jet<N> r = f;
  r.a = f.a + s;
  return r;
}
template <int N>
inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> r;
  r.a = f.a - g.a;
  for (int i = 0; i < N; i++) r.v[i] = f.v[i] - g.v[i];
  return r;
}
template <int N>
inline Jet<N> operator-(const Jet<N>& f, double s) {
  Jet<N> r = f;
  r.a = f.a - s;
  return r;
}
template <int N>
inline Jet<N> operator-(double s, const Jet<N>& f) {
  Jet<N> r;
  r.a = s - f.a;
  for (int i = 0; i < N; i++) r.v[i] = -f.v[i];
  return r;
}
template <int N>
inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> r;
  r.a = f.a * g.a;
  for (int i = 0; i < N; i++) r.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return r;
}
template <int N>
inline Jet<N> operator*(const Jet<N>& f, double s) {
  Jet<N> r;
  r.a = f.a * s;
  for (int i = 0; i < N; i++) r.v[i] = f.v[i] * s;
  return r;
}
template <int N>
inline Jet<N> operator*(double s, const Jet<N>& f) {
  return f * s;
}
template <int N>
inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  const double g_a_inverse = 1.0 / g.a;
  const double f_a_by_g_a = f.a * g_a_inverse;
  Jet<N> r;
  r.a = f_a_by_g_a;
  for (int i = 0; i < N; i++) r.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
  return r;
}
template <int N>
inline Jet<N> operator/(double s, const Jet<N>& g) {
  const double minus_s_g_a_inverse2 = -s / (g.a * g.a);
  Jet<N> r;
  r.a = s / g.a;
  for (int i = 0; i < N; i++) r.v[i] = g.v[i] * minus_s_g_a_inverse2;
  return r;
}
template <int N>
inline Jet<N> operator/(const Jet<N>& f, double s) {
  const double s_inverse = 1.0 / s;
  Jet<N> r;
  r.a = f.a * s_inverse;
  for (int i = 0; i < N; i++) r.v[i] = f.v[i] * s_inverse;
  return r;
}

// comparisons look at the scalar part only (ceres/jet.h)
template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> inline bool operator<(const Jet<N>& f, double g) { return f.a < g; }
template <int N> inline bool operator<(double f, const Jet<N>& g) { return f < g.a; }
template <int N> inline bool operator>(const Jet<N>& f, const Jet<N>& g) { return f.a > g.a; }
template <int N> inline bool operator>(const Jet<N>& f, double g) { return f.a > g; }
template <int N> inline bool operator>(double f, const Jet<N>& g) { return f > g.a; }

template <int N>
inline Jet<N> sqrt(const Jet<N>& f) {
  const double tmp = std::sqrt(f.a);
  const double two_a_inverse = 1.0 / (2.0 * tmp);
  Jet<N> r;
  r.a = tmp;
  for (int i = 0; i < N; i++) r.v[i] = f.v[i] * two_a_inverse;
  return r;
}
template <int N>
inline Jet<N> log(const Jet<N>& f) {
  const double a_inverse = 1.0 / f.a;
  Jet<N> r;
  r.a = std::log(f.a);
  for (int i = 0; i < N; i++) r.v[i] = f.v[i] * a_inverse;
  return r;
}
template <int N>
inline Jet<N> abs(const Jet<N>& f) {
  const double s = std::copysign(1.0, f.a);
  Jet<N> r;
  r.a = std::abs(f.a);
  for (int i = 0; i < N; i++) r.v[i] = s * f.v[i];
  return r;
}
template <int N>
inline Jet<N> atan2(const Jet<N>& g, const Jet<N>& f) {
  const double tmp = 1.0 / (f.a * f.a + g.a * g.a);
  Jet<N> r;
  r.a = std::atan2(g.a, f.a);
  for (int i = 0; i < N; i++) r.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
  return r;
}
template <int N>
inline Jet<N> asin(const Jet<N>& f) {
  const double tmp = 1.0 / std::sqrt(1.0 - f.a * f.a);
  Jet<N> r;
  r.a = std::asin(f.a);
  for (int i = 0; i < N; i++) r.v[i] = tmp * f.v[i];
  return r;
}
template <int N>
inline Jet<N> atan(const Jet<N>& f) {
  const double tmp = 1.0 / (1.0 + f.a * f.a);
  Jet<N> r;
  r.a = std::atan(f.a);
  for (int i = 0; i < N; i++) r.v[i] = tmp * f.v[i];
  return r;
}
template <int N>
inline Jet<N> tan(const Jet<N>& f) {
  const double tan_a = std::tan(f.a);
  const double tmp = 1.0 + tan_a * tan_a;
  Jet<N> r;
  r.a = tan_a;
  for (int i = 0; i < N; i++) r.v[i] = tmp * f.v[i];
  return r;
}
template <int N>
inline Jet<N> pow(const Jet<N>& f, double p) {
  const double tmp = p * std::pow(f.a, p - 1.0);
  Jet<N> r;
  r.a = std::pow(f.a, p);
  for (int i = 0; i < N; i++) r.v[i] = tmp * f.v[i];
  return r;
}

// double overloads so the scalar-generic camera functors resolve the same names for T=double
inline double sqrt(double x) { return std::sqrt(x); }
inline double log(double x) { return std::log(x); }
inline double abs(double x) { return std::abs(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }
inline double asin(double x) { return std::asin(x); }
inline double atan(double x) { return std::atan(x); }
inline double tan(double x) { return std::tan(x); }
inline double pow(double x, double p) { return std::pow(x, p); }

// nid_cost.hpp:11-19 get_real
template <int N> inline double get_real(const Jet<N>& x) { return x.a; }
inline double get_real(const double& x) { return x; }

}  // namespace oracle
