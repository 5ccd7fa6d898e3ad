// nid_device.hpp -- device-side building blocks of the NID registration kernels (gfx950 / CDNA4).
//
// What the reference computes per point per evaluation (include/vlcal/costs/nid_cost.hpp:46-84,
// src/vlcal/calib/cost_calculator_nid.cpp:30-52) is re-designed here as a gather/reduce pipeline:
//   * points are pre-bucketed by their pose-independent histogram column (bin_points), so a
//     workgroup owns a small tile of the joint histogram in LDS and never touches global atomics
//     inside the point loop;
//   * weights are accumulated as 64-bit FIXED-POINT integers (ds_add_u64), which makes the
//     histogram independent of thread order, workgroup count and GPU count (bit-reproducible,
//     all-reducible as int64);
//   * the Jacobian is obtained in reverse mode: dNID/dh (a B x B table) from the finished
//     histogram, then one more streaming pass that contracts it with d(weight)/d(pose); no
//     8-wide Jet histogram is ever formed;
//   * camera models are template parameters (one kernel instantiation per model), written once
//     for T = real (value) and T = Dual3<real> (value + d/d(x,y,z)).
// No MFMA: nothing here is a dense contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include <cmath>
#include <cstring>
#include "nid_atan_table.hpp"
#include "nid_log_table.hpp"

namespace nidreg {

typedef unsigned long long u64;

// the scalar math (camera models, B-spline basis, dual numbers) is also compiled for the host so that
// tests/cxx/test_device_math.cpp can check it without a GPU; the loads / LDS code below is device only
#define NID_HD __host__ __device__ __forceinline__

// ------------------------------------------------------------------------------------------
// scalar helpers for float / double
NID_HD float m_sqrt(float x) { return sqrtf(x); }
NID_HD double m_sqrt(double x) { return sqrt(x); }
NID_HD float m_atan2(float y, float x) { return atan2f(y, x); }
NID_HD double m_atan2(double y, double x) { return atan2(y, x); }
NID_HD float m_asin(float x) { return asinf(x); }
NID_HD double m_asin(double x) { return asin(x); }
NID_HD float m_atan(float x) { return atanf(x); }
NID_HD double m_atan(double x) { return atan(x); }
NID_HD float m_abs(float x) { return fabsf(x); }
NID_HD double m_abs(double x) { return fabs(x); }
NID_HD float m_floor(float x) { return floorf(x); }
NID_HD double m_floor(double x) { return floor(x); }
NID_HD float m_val(float x) { return x; }
NID_HD double m_val(double x) { return x; }
// x - floor(x) for x >= 0 in one instruction (v_fract); for such x the subtraction is exact, so this equals
// x - floor(x) bit for bit, and int(x) (truncation) is floor(x): knot and fraction cost two operations, not three
NID_HD double m_fract(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fract(x);
#else
  return x - floor(x);
#endif
}
NID_HD float m_fract(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fractf(x);
#else
  return x - floorf(x);
#endif
}

// ------------------------------------------------------------------------------------------
// forward-mode dual number with three partials (d/dx, d/dy, d/dz of the camera-frame point).
// Same chain rules as the reference's ceres::Jet arithmetic, three slots instead of seven: the
// remaining 3x7 factor d(p_cam)/d(pose) is linear in the LiDAR point and is folded into a
// 3x3 + 3 accumulator (see k_spline_grad).
template <typename real>
struct Dual3 {
  real a, d0, d1, d2;
  NID_HD Dual3() {}
  NID_HD Dual3(real v) : a(v), d0(0), d1(0), d2(0) {}
  NID_HD Dual3(real v, real x, real y, real z) : a(v), d0(x), d1(y), d2(z) {}
};

template <typename real>
NID_HD real m_val(const Dual3<real>& x) { return x.a; }

#define NID_D Dual3<real>
template <typename real> NID_HD NID_D operator+(const NID_D& f, const NID_D& g) { return NID_D(f.a + g.a, f.d0 + g.d0, f.d1 + g.d1, f.d2 + g.d2); }
template <typename real> NID_HD NID_D operator+(const NID_D& f, real s) { return NID_D(f.a + s, f.d0, f.d1, f.d2); }
template <typename real> NID_HD NID_D operator+(real s, const NID_D& f) { return NID_D(s + f.a, f.d0, f.d1, f.d2); }
template <typename real> NID_HD NID_D operator-(const NID_D& f, const NID_D& g) { return NID_D(f.a - g.a, f.d0 - g.d0, f.d1 - g.d1, f.d2 - g.d2); }
template <typename real> NID_HD NID_D operator-(const NID_D& f, real s) { return NID_D(f.a - s, f.d0, f.d1, f.d2); }
template <typename real> NID_HD NID_D operator-(real s, const NID_D& f) { return NID_D(s - f.a, -f.d0, -f.d1, -f.d2); }
template <typename real> NID_HD NID_D operator-(const NID_D& f) { return NID_D(-f.a, -f.d0, -f.d1, -f.d2); }
template <typename real> NID_HD NID_D operator*(const NID_D& f, const NID_D& g) {
  return NID_D(f.a * g.a, fma(f.a, g.d0, f.d0 * g.a), fma(f.a, g.d1, f.d1 * g.a), fma(f.a, g.d2, f.d2 * g.a));
}
template <typename real> NID_HD NID_D operator*(const NID_D& f, real s) { return NID_D(f.a * s, f.d0 * s, f.d1 * s, f.d2 * s); }
template <typename real> NID_HD NID_D operator*(real s, const NID_D& f) { return NID_D(f.a * s, f.d0 * s, f.d1 * s, f.d2 * s); }
template <typename real> NID_HD NID_D operator/(const NID_D& f, const NID_D& g) {
  const real gi = real(1) / g.a;
  const real q = f.a * gi;
  return NID_D(q, fma(-q, g.d0, f.d0) * gi, fma(-q, g.d1, f.d1) * gi, fma(-q, g.d2, f.d2) * gi);
}
template <typename real> NID_HD NID_D operator/(const NID_D& f, real s) {
  const real si = real(1) / s;
  return NID_D(f.a * si, f.d0 * si, f.d1 * si, f.d2 * si);
}
template <typename real> NID_HD bool operator<(const NID_D& f, real s) { return f.a < s; }
template <typename real> NID_HD bool operator>(const NID_D& f, real s) { return f.a > s; }
template <typename real> NID_HD NID_D m_sqrt(const NID_D& f) {
  const real t = m_sqrt(f.a);
  const real k = real(1) / (real(2) * t);
  return NID_D(t, f.d0 * k, f.d1 * k, f.d2 * k);
}
template <typename real> NID_HD NID_D m_atan2(const NID_D& g, const NID_D& f) {
  const real k = real(1) / fma(f.a, f.a, g.a * g.a);
  return NID_D(m_atan2(g.a, f.a), k * fma(-g.a, f.d0, f.a * g.d0), k * fma(-g.a, f.d1, f.a * g.d1), k * fma(-g.a, f.d2, f.a * g.d2));
}
template <typename real> NID_HD NID_D m_asin(const NID_D& f) {
  const real k = real(1) / m_sqrt(real(1) - f.a * f.a);
  return NID_D(m_asin(f.a), k * f.d0, k * f.d1, k * f.d2);
}
template <typename real> NID_HD NID_D m_atan(const NID_D& f) {
  const real k = real(1) / (real(1) + f.a * f.a);
  return NID_D(m_atan(f.a), k * f.d0, k * f.d1, k * f.d2);
}
template <typename real> NID_HD NID_D m_abs(const NID_D& f) {
  const real s = f.a < real(0) ? real(-1) : real(1);
  return NID_D(m_abs(f.a), s * f.d0, s * f.d1, s * f.d2);
}
#undef NID_D

template <typename T> struct scalar_of { typedef T type; };
template <typename real> struct scalar_of<Dual3<real> > { typedef real type; };

// ------------------------------------------------------------------------------------------
// kernel-argument PODs
template <typename real>
struct CamParams {
  real intr[5];
  real dist[8];
};
// p_cam = R p + t with R = I + 2 w [v]x + 2 [v]x^2 built on the host from the UN-normalised
// quaternion exactly as Sophus' SO3 * point expands (nid_cost.hpp:47)
// `atan` model: the distortion slots behind its one coefficient carry the two constants its projection derives from it
constexpr int kAtanD1 = 6, kAtanD2 = 7;
template <typename real>
inline void cam_derive(int model, CamParams<real>& c);  // (host side: defined below the model ids)
template <typename real>
struct PoseParams {
  real R[9];
  real t[3];
};
// rows 0..2 of a 4x4 row-major isometry (cost_calculator_nid.cpp:31)
template <typename real>
struct IsoParams {
  real m[12];
};

enum { MODEL_PLUMB_BOB = 0, MODEL_FISHEYE = 1, MODEL_OMNIDIR = 2, MODEL_EQUIRECT = 3, MODEL_ATAN = 4, MODEL_RATIONAL = 5 };
// constants a model derives from its coefficients, in the reference's expressions, once per camera on the host
template <typename real>
inline void cam_derive(int model, CamParams<real>& c) {
  if (model == MODEL_ATAN) {  // atan.hpp:21-22
    const double d0 = double(c.dist[0]);
    c.dist[kAtanD1] = real(1.0 / d0);
    c.dist[kAtanD2] = real(2.0 * std::tan(d0 / 2.0));
  }
}

// ------------------------------------------------------------------------------------------
// a*b + c: one fused multiply-add in the FAST (SPLINE) instantiation when all operands are plain
// scalars, an unfused multiply then add otherwise (NEAREST path: bit-identical to the CPU, which has
// no fma; Dual3 operands: their operators).  Every use below nests the terms so that the unfused
// form reproduces the reference's left-to-right association exactly (fp addition is commutative).
template <bool FAST, typename A, typename B, typename C>
NID_HD auto mad(const A& a, const B& b, const C& c) -> decltype(a * b + c) {
  if constexpr (FAST && std::is_floating_point<A>::value && std::is_floating_point<B>::value && std::is_floating_point<C>::value) {
    return fma(a, b, c);
  } else {
    return a * b + c;
  }
}

// 1/z by v_rcp_f64 + Newton steps r += r (1 - z r) (instead of the ~12 instructions of the IEEE division
// sequence).  The hardware seed is good to <= 2^29 ulp, i.e. 2^-23 relative (measured: tools/ubench_rcp.hip), and
// every step squares the error: kNewtonSteps = 1 leaves <= 2^-46 = 1.4e-14 relative -- 3e-11 px on a 2000-px
// coordinate, three orders below the histogram tolerance --, 2 gives the correctly rounded quotient to ~1 ulp.
// |z| is a camera-frame depth in metres, never denormal / inf in range of interest, and a NaN / zero z still
// yields a NaN / inf projection, i.e. an outlier.
#ifndef NID_NEWTON_STEPS
#define NID_NEWTON_STEPS 1
#endif
constexpr int kNewtonSteps = NID_NEWTON_STEPS;
NID_HD double rcp_seed(double z) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcp(z);
#else
  return double(1.0f / float(z));  // host build (tests): a 24-bit seed, like the hardware's
#endif
}
NID_HD double rsq_seed(double z) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rsq(z);
#else
  return double(1.0f / sqrtf(float(z)));
#endif
}
NID_HD double fast_rcp(double z) {
  double r = rcp_seed(z);
#pragma unroll
  for (int k = 0; k < kNewtonSteps; k++) r = fma(fma(-z, r, 1.0), r, r);
  return r;
}
NID_HD float fast_rcp(float z) { return 1.0f / z; }
// 1/sqrt(z) by v_rsq_f64 + the same number of Newton steps r += r (1 - z r^2) / 2.  z = 0 -> NaN (inf * 0), z < 0 -> NaN.
NID_HD double fast_rsq(double z) {
  double r = rsq_seed(z);
#pragma unroll
  for (int k = 0; k < kNewtonSteps; k++) r = fma(0.5 * r, fma(-z * r, r, 1.0), r);
  return r;
}
NID_HD float fast_rsq(float z) { return 1.0f / sqrtf(z); }

// atan2 for the SPLINE kernels' fisheye / equirectangular projections: octant reduction to mn / mx in [0, 1], then a
// 257-entry table (nid_atan_table.hpp: atan(i / 256) correctly rounded; 2 KB, read through the vector L1, where it stays
// resident) and the addition theorem on the UN-DIVIDED pair,
//   atan(mn / mx) = atan(t0) + atan(d),   d = (mn - t0 mx) / (mx + t0 mn),   t0 = i / 256 the table point next to mn / mx.
// Round 5: ONE double-precision reciprocal per call instead of two.  Until round 4 the quotient t = mn / mx was formed first
// (fast_rcp) and d = (t - t0) / (1 + t t0) needed a second one; but t is only used to PICK i -- any table point within 1/256
// of it will do -- so a single-precision quotient (v_rcp_f32: 1 ulp) picks it, and the one remaining fp64 division acts on d,
// which is <= 1/511: its 2^-46 relative error (one Newton step) is 3e-17 absolute in the angle, where the old form carried the
// reciprocal's 1.4e-14 into t itself (7e-15 in the angle).  The numerator is ONE fma of exact operands (t0 has 9 significant
// bits): rounded once.  atan(d) = d (1 - d^2/3 + d^4/5) (next term d^7/7 < 2e-20).  Max abs error 2.3e-16 on the host
// (test_device_math).  atan2(0, 0) = 0 like libm (equirectangular.hpp:21 relies on it for points on the vertical axis): the
// denominator is held above 1e-30 (lengths in metres; v_max_f64 returns the non-NaN operand, so a NaN reaches the result through the numerator).
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const double g_atan_tab[kAtanTableN + 1] = {NID_ATAN_TABLE_VALUES};
#else
static const double g_atan_tab[kAtanTableN + 1] = {NID_ATAN_TABLE_VALUES};
#endif
NID_HD float rcp_f32(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);
#else
  return 1.0f / x;
#endif
}
NID_HD double fast_atan2(double y, double x) {
  const double ax = fabs(x), ay = fabs(y);
  const bool swap = ay > ax;
  const double mx = swap ? ay : ax, mn = swap ? ax : ay;  // NaN operands propagate (no maxNum semantics)
  // table point: single precision is plenty (|256 t - i| <= 0.5 + 2^-15); 0/0, inf/inf, NaN -> some i in range after the clamp
  const float ti = rintf((float(mn) * float(kAtanTableN)) * rcp_f32(float(mx)));
#if defined(__HIP_DEVICE_COMPILE__)
  int i = int(ti);  // NaN -> 0 (v_cvt_i32_f32): the result is NaN through `num`
#else
  int i = ti == ti ? int(ti) : 0;  // (a NaN conversion is undefined behaviour in the host build)
#endif
  i = i < 0 ? 0 : (i > kAtanTableN ? kAtanTableN : i);
  const double t0 = double(i) * (1.0 / double(kAtanTableN));
  const double num = fma(-t0, mx, mn);
  const double den = fmax(fma(t0, mn, mx), 1e-30);
  const double d = num * fast_rcp(den);
  const double d2 = d * d;
  double a = fma(d, fma(d2, fma(d2, 0.2, -1.0 / 3.0), 1.0), g_atan_tab[i]);
  a = swap ? 1.57079632679489661923 - a : a;
  a = x < 0.0 ? 3.14159265358979323846 - a : a;
  return copysign(a, y);
}
NID_HD float fast_atan2(float y, float x) { return atan2f(y, x); }

// Natural logarithm for the entropy terms and the G tile (p log(p + 1e-6), log(q + 1e-6): arguments in [1e-6, 1.000001]; any
// positive normal double works).  The library's log is ~95 dependent VALU instructions, 44 of them double-double additions,
// and the entropy tail that every gradient workgroup runs in its prologue is a chain of three to four of them at one wave per
// SIMD (2.8 + 0.9 us of a small evaluation, profiles/archive/r04n_stage_times.json).  Here: x = 2^k m with m in [0.6875, 1.375) --
// so that x near 1 gives k = 0 and no cancellation against k ln 2 --, a 128-entry table (nid_log_table.hpp: r_i = 1 / centre of
// m's sub-interval, -log(r_i) correctly rounded; 2 KB, read through the vector L1 like the atan table),
//   log(x) = k ln 2 - log(r_i) + log1p(z),   z = m r_i - 1 (one fma: |z| <= 0.004),
//   log1p(z) = z - z^2/2 + z^3/3 - z^4/4 + z^5/5 - z^6/6   (next term z^7/7 < 2.4e-18):
// ~19 instructions.  Accuracy (test_device_math, on the host against a 60-digit reference): RELATIVE error of about one ulp of
// the result (<= 2.6e-16) over [1e-9, 4]; in absolute terms < 3e-16 only where |log x| <= 1 -- half an ulp of log(1e-9) = -20.7 is
// already 1.8e-15, and the single-double ln 2 adds |k| x 2.3e-17.  What matters downstream is p log(p + 1e-6) with p <= 1, whose
// absolute error stays below 4e-16: 1e-5 of the cost's parity bar.  Every route of the entropy work calls this one function: the
// cost stays bit-identical across routes.  PRECONDITION: x positive and NORMAL (the callers add 1e-6 first); zero, subnormals,
// negatives, inf, NaN (only reached when there is no inlier, where entropy_scalars overrides the result) make the exponent /
// table split below meaningless: some finite or NaN value comes back, never a trap.
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const double g_log_tab[2 * kLogTableN] = {NID_LOG_TABLE_VALUES};
#else
static const double g_log_tab[2 * kLogTableN] = {NID_LOG_TABLE_VALUES};
#endif
NID_HD double fast_log(double x) {
  uint32_t hi, lo;
#if defined(__HIP_DEVICE_COMPILE__)
  hi = uint32_t(__double2hiint(x)), lo = uint32_t(__double2loint(x));
#else
  uint64_t bits;
  std::memcpy(&bits, &x, sizeof(bits));
  hi = uint32_t(bits >> 32), lo = uint32_t(bits);
#endif
  const uint32_t tmp = hi - 0x3fe60000u;
  const int k = int(tmp) >> 20;                 // (arithmetic shift: x < 0.6875 gives negative k)
  const uint32_t i = (tmp >> 13) & 127u;
  const uint32_t mhi = hi - (tmp & 0xfff00000u);  // exponent of m: 0x3fe or 0x3ff
  double m;
#if defined(__HIP_DEVICE_COMPILE__)
  m = __hiloint2double(int(mhi), int(lo));
#else
  bits = (uint64_t(mhi) << 32) | lo;
  std::memcpy(&m, &bits, sizeof(m));
#endif
  const double r = g_log_tab[2 * i], t = g_log_tab[2 * i + 1];
  const double z = fma(m, r, -1.0);
  const double q = fma(fma(fma(fma(-1.0 / 6.0, z, 0.2), z, -0.25), z, 1.0 / 3.0), z, -0.5);
  const double l1p = fma(z * z, q, z);
  return fma(double(k), 0.69314718055994530942, t + l1p);
}

// perspective division: exact x/z, y/z (NEAREST path) or one reciprocal and two multiplies (SPLINE
// kernels; both passes use the same form, so they agree on every knot)
template <bool FAST, typename real>
NID_HD void persp(real x, real y, real z, real& px, real& py) {
  if (FAST) {
    const real iz = fast_rcp(z);
    px = x * iz;
    py = y * iz;
  } else {
    px = x / z;
    py = y / z;
  }
}
template <bool FAST, typename real>
NID_HD void persp(const Dual3<real>& x, const Dual3<real>& y, const Dual3<real>& z, Dual3<real>& px, Dual3<real>& py) {
  px = x / z;  // Dual3 division is reciprocal-multiply already
  py = y / z;
}

// Radial-tangential distortion of the SPLINE kernels (plumb_bob / rational_polynomial / omnidir), factored so that the
// value and -- in the gradient pass -- its Jacobian share every product (rc = the radial factor):
//   dx = px i0 + p2 py^2,   i0 = rc + 2 p1 py + 3 p2 px      ( = rc px + 2 p1 px py + p2 (r2 + 2 px^2), pinhole.hpp:33-47 )
//   dy = py i1 + p1 px^2,   i1 = rc + 2 p2 px + 3 p1 py
//   d(dx)/d(px) = i0 + 3 p2 px + px^2 rc2,   d(dy)/d(py) = i1 + 3 p1 py + py^2 rc2,   rc2 = 2 d(rc)/d(r2)
//   d(dx)/d(py) = d(dy)/d(px) = px py rc2 + 2 p1 px + 2 p2 py
// -- 8 fused operations for the value (10 in the reference's association) and 9 more for the Jacobian (20 before).
// The two functions repeat i0 / i1 with identical expressions: after inlining they are computed once.
template <typename real>
NID_HD void radtan_value(real p1, real p2, real px, real py, real x2, real y2, real rc, real& dx, real& dy) {
  const real i0 = fma(real(3) * p2, px, fma(real(2) * p1, py, rc));
  const real i1 = fma(real(3) * p1, py, fma(real(2) * p2, px, rc));
  dx = fma(px, i0, p2 * y2);
  dy = fma(py, i1, p1 * x2);
}
template <typename real>
NID_HD void radtan_partials(real p1, real p2, real px, real py, real x2, real y2, real rc, real rc2, real& a00, real& off, real& a11) {
  const real i0 = fma(real(3) * p2, px, fma(real(2) * p1, py, rc));
  const real i1 = fma(real(3) * p1, py, fma(real(2) * p2, px, rc));
  a00 = fma(x2, rc2, fma(real(3) * p2, px, i0));
  off = fma(px * py, rc2, fma(real(2) * p1, px, (real(2) * p2) * py));
  a11 = fma(y2, rc2, fma(real(3) * p1, py, i1));
}

// ---- the FAST (SPLINE-kernel) forms of the three wide-angle models, each as ONE core shared by the histogram pass
// (project<..., FAST>), the gradient pass (project_fwd) and the Jacobian utility (project_jac): the projected point is the
// same expression everywhere, so every pass agrees on every knot.
//
// fisheye.hpp:14-36:  (u, v) = f .* s (x, y) + c,  s = theta_d(theta) / r,  theta = atan2(r, |z|),  r = |(x, y)|.
// 1/r from one rsqrt (r = r2 / r), atan2 without an IEEE division, theta_d / r as a multiply.  r2 = 0 gives 1/r = NaN,
// hence a NaN projection, like the reference's 0/0 (fisheye.hpp:31-33).
template <typename real>
struct FisheyeCore {
  real r2, ir, az, th2, s;
};
template <typename real>
NID_HD FisheyeCore<real> fisheye_core(const CamParams<real>& c, real x, real y, real z) {
  FisheyeCore<real> k;
  k.r2 = fma(x, x, y * y);
  k.ir = fast_rsq(k.r2);
  k.az = m_abs(z);
  const real theta = fast_atan2(k.r2 * k.ir, k.az);
  k.th2 = theta * theta;
  const real theta_d = theta * fma(k.th2, fma(k.th2, fma(k.th2, fma(k.th2, c.dist[3], c.dist[2]), c.dist[1]), c.dist[0]), real(1));
  k.s = theta_d * k.ir;
  return k;
}
// omnidir.hpp:14-41:  m = s_xy / (s_z + xi) with s = p / |p|, i.e. m = p_xy / (p_z + xi |p|): ONE rsqrt for |p| = n2 rsq(n2) and
// ONE reciprocal of the un-normalised denominator (round 5; rounds 1-4 normalised the bearing first: three more multiplies and
// two more values alive in the gradient pass).  |p| = 0 (omnidir.hpp:19 skips the normalisation): den = xi, m = 0 -- n = 1 there.
template <typename real>
struct OmnidirCore {
  real in, n, iden, ux, uy;
};
template <typename real>
NID_HD OmnidirCore<real> omnidir_core(const CamParams<real>& c, real x, real y, real z) {
  OmnidirCore<real> k;
  const real n2 = fma(z, z, fma(y, y, x * x));
  const bool pos = n2 > real(0);
  k.in = pos ? fast_rsq(n2) : real(1);
  k.n = pos ? n2 * k.in : real(1);
  k.iden = fast_rcp(fma(c.intr[4], k.n, z));
  k.ux = x * k.iden;
  k.uy = y * k.iden;
  return k;
}
// equirectangular.hpp:14-28:  lat = -asin(y / |p|) = -atan2(y, rho), rho = |(x, z)|: no normalisation of the bearing is needed
// at all, and both angles share the division-free atan2.  |p|^2 < 1e-3 -> the image centre (:16-18).
template <typename real>
struct EquirectCore {
  real rho2, irho;
  bool tiny;
};
template <typename real>
NID_HD EquirectCore<real> equirect_core(const CamParams<real>& c, real x, real y, real z, real& u, real& v) {
  EquirectCore<real> k;
  const real n2 = fma(z, z, fma(y, y, x * x));
  k.rho2 = fma(x, x, z * z);
  const bool rpos = k.rho2 > real(0);
  k.irho = rpos ? fast_rsq(k.rho2) : real(0);
  const real rho = k.rho2 * k.irho;
  const real lon = fast_atan2(x, z);
  const real nlat = fast_atan2(y, rho);  // = -lat
  k.tiny = n2 < real(1e-3);
  const real uu = fma(c.intr[0] * real(0.15915494309189533577), lon, c.intr[0] * real(0.5));
  const real vv = fma(c.intr[1] * real(0.31830988618379067154), nlat, c.intr[1] * real(0.5));
  u = k.tiny ? c.intr[0] * real(0.5) : uu;
  v = k.tiny ? c.intr[1] * real(0.5) : vv;
  return k;
}

// projection models (reference: include/camera/{pinhole,fisheye,omnidir,equirectangular,atan,
// rational_polynomial}.hpp), T = real or Dual3<real>.
template <int MODEL, typename T, typename real, bool FAST = false>
NID_HD void project(const CamParams<real>& c, const T& x, const T& y, const T& z, T& u, T& v) {
  if (MODEL == MODEL_PLUMB_BOB) {  // pinhole.hpp:13-51, distortion k1 k2 p1 p2 k3
    const real k1 = c.dist[0], k2 = c.dist[1], p1 = c.dist[2], p2 = c.dist[3], k3 = c.dist[4];
    T px, py;
    persp<FAST>(x, y, z, px, py);
    const T x2 = px * px, y2 = py * py;
    const T r2 = mad<FAST>(px, px, y2);                                   // x2 + y2
    const T r4 = r2 * r2;
    const T r6 = r2 * r4;
    const T rc = mad<FAST>(k3, r6, mad<FAST>(k2, r4, mad<FAST>(k1, r2, real(1))));  // 1 + k1 r2 + k2 r4 + k3 r6
    T dx, dy;
    if constexpr (FAST && std::is_floating_point<T>::value) {
      radtan_value<T>(p1, p2, px, py, x2, y2, rc, dx, dy);
    } else {
      const T t1 = real(2) * px * py;
      const T t2 = mad<FAST>(real(2), x2, r2);                            // r2 + 2 x2
      const T t3 = mad<FAST>(real(2), y2, r2);
      dx = mad<FAST>(p2, t2, mad<FAST>(p1, t1, rc * px));                 // rc px + p1 t1 + p2 t2
      dy = mad<FAST>(p2, t1, mad<FAST>(p1, t3, rc * py));                 // rc py + p1 t3 + p2 t1
    }
    u = mad<FAST>(c.intr[0], dx, c.intr[2]);
    v = mad<FAST>(c.intr[1], dy, c.intr[3]);
  } else if (MODEL == MODEL_FISHEYE) {  // fisheye.hpp:14-36 (abs(z) at :16)
    const real k1 = c.dist[0], k2 = c.dist[1], k3 = c.dist[2], k4 = c.dist[3];
    if constexpr (FAST && std::is_floating_point<T>::value) {
      const FisheyeCore<T> k = fisheye_core<T>(c, x, y, z);
      u = fma(c.intr[0], k.s * x, c.intr[2]);
      v = fma(c.intr[1], k.s * y, c.intr[3]);
    } else {
      const T r = m_sqrt(mad<FAST>(x, x, y * y));
      const T theta = m_atan2(r, m_abs(z));
      const T th2 = theta * theta;
      const T th4 = th2 * th2;
      const T th6 = th4 * th2;
      const T th8 = th4 * th4;
      const T theta_d = theta * mad<FAST>(k4, th8, mad<FAST>(k3, th6, mad<FAST>(k2, th4, mad<FAST>(k1, th2, real(1)))));
      const T s = theta_d / r;
      u = mad<FAST>(c.intr[0], s * x, c.intr[2]);
      v = mad<FAST>(c.intr[1], s * y, c.intr[3]);
    }
  } else if (MODEL == MODEL_OMNIDIR) {  // omnidir.hpp:14-41
    const real xi = c.intr[4];
    const real k1 = c.dist[0], k2 = c.dist[1], p1 = c.dist[2], p2 = c.dist[3];
    T ux, uy;
    if constexpr (FAST && std::is_floating_point<T>::value) {
      const OmnidirCore<T> k = omnidir_core<T>(c, x, y, z);
      ux = k.ux;
      uy = k.uy;
    } else {
      const T n2 = mad<FAST>(z, z, mad<FAST>(y, y, x * x));               // x x + y y + z z
      T sx = x, sy = y, sz = z;
      if (n2 > real(0)) {
        const T n = m_sqrt(n2);
        sx = x / n;
        sy = y / n;
        sz = z / n;
      }
      const T den = sz + xi;
      ux = sx / den;
      uy = sy / den;
    }
    const T x2 = ux * ux, y2 = uy * uy, xy = ux * uy;
    const T r2 = mad<FAST>(ux, ux, y2);
    const T r4 = r2 * r2;
    const T dr = mad<FAST>(k2, r4, mad<FAST>(k1, r2, real(1)));
    T nx, ny;
    if constexpr (FAST && std::is_floating_point<T>::value) {
      radtan_value<T>(p1, p2, ux, uy, x2, y2, dr, nx, ny);
    } else {
      nx = mad<FAST>(p2, mad<FAST>(real(2), x2, r2), mad<FAST>(real(2) * p1, xy, ux * dr));
      ny = mad<FAST>(real(2) * p2, xy, mad<FAST>(p1, mad<FAST>(real(2), y2, r2), uy * dr));
    }
    u = mad<FAST>(c.intr[0], nx, c.intr[2]);
    v = mad<FAST>(c.intr[1], ny, c.intr[3]);
  } else if (MODEL == MODEL_EQUIRECT) {  // equirectangular.hpp:14-28, intr = [W H]
    if constexpr (FAST && std::is_floating_point<T>::value) {
      (void)equirect_core<T>(c, x, y, z, u, v);
    } else {
      const T n2 = mad<FAST>(z, z, mad<FAST>(y, y, x * x));
      if (n2 < real(1e-3)) {
        u = T(c.intr[0] / real(2));
        v = T(c.intr[1] / real(2));
      } else {
        const T n = m_sqrt(n2);
        const T bx = x / n, by = y / n, bz = z / n;
        const T lat = -m_asin(by);
        const T lon = m_atan2(bx, bz);
        u = c.intr[0] * (real(0.5) + lon / real(2.0 * 3.14159265358979323846));
        v = c.intr[1] * (real(0.5) - lat / real(3.14159265358979323846));
      }
    }
  } else if (MODEL == MODEL_ATAN) {  // atan.hpp:14-39
    const real d0 = c.dist[0];
    T px, py;
    persp<FAST>(x, y, z, px, py);
    const T r = m_sqrt(mad<FAST>(px, px, py * py));
    T dx = px, dy = py;
    if (!(r < real(1e-3) || d0 < real(1e-7))) {
      // d1 = 1 / d0 and d2 = 2 tan(d0 / 2) (atan.hpp:21-22) are camera constants: computed once on the HOST (cam_derive below) --
      // the reference's own libm, and no call into the device's tan() in a point kernel (the call alone pinned the gradient
      // kernel of this model at 165 VGPRs, whatever its arithmetic)
      const real d1 = c.dist[kAtanD1];
      const real d2 = c.dist[kAtanD2];
      T factor;
      if constexpr (FAST && std::is_floating_point<T>::value) {
        // SPLINE kernels: the table atan2 of the wide-angle models (4.4e-16 absolute) instead of the library's atan -- which held
        // this model's gradient kernel at 166 VGPRs where the others need 124-130
        factor = d1 * fast_atan2(r * d2, T(1)) * fast_rcp(r);
      } else {
        factor = d1 * m_atan(r * d2) / r;
      }
      dx = factor * px;
      dy = factor * py;
    }
    u = mad<FAST>(c.intr[0], dx, c.intr[2]);
    v = mad<FAST>(c.intr[1], dy, c.intr[3]);
  } else {  // rational_polynomial.hpp:11-58, k1 k2 p1 p2 k3 k4 k5 k6
    const real k1 = c.dist[0], k2 = c.dist[1], p1 = c.dist[2], p2 = c.dist[3];
    const real k3 = c.dist[4], k4 = c.dist[5], k5 = c.dist[6], k6 = c.dist[7];
    T px, py;
    persp<FAST>(x, y, z, px, py);
    const T x2 = px * px, y2 = py * py;
    const T r2 = mad<FAST>(px, px, y2);
    const T r4 = r2 * r2;
    const T r6 = r2 * r4;
    const T num = mad<FAST>(k3, r6, mad<FAST>(k2, r4, mad<FAST>(k1, r2, real(1))));
    const T den = mad<FAST>(k6, r6, mad<FAST>(k5, r4, mad<FAST>(k4, r2, real(1))));
    const T rc = den > real(1e-8) ? num / den : num;
    T dx, dy;
    if constexpr (FAST && std::is_floating_point<T>::value) {
      radtan_value<T>(p1, p2, px, py, x2, y2, rc, dx, dy);
    } else {
      const T t1 = real(2) * px * py;
      const T t2 = mad<FAST>(real(2), x2, r2);
      const T t3 = mad<FAST>(real(2), y2, r2);
      dx = mad<FAST>(p2, t2, mad<FAST>(p1, t1, rc * px));
      dy = mad<FAST>(p2, t1, mad<FAST>(p1, t3, rc * py));
    }
    u = mad<FAST>(c.intr[0], dx, c.intr[2]);
    v = mad<FAST>(c.intr[1], dy, c.intr[3]);
  }
}

// `atan` (FOV) model, atan.hpp:14-39: (dx, dy) = f(r) (px, py), f = atan(d2 r) / (d0 r), d2 = 2 tan(d0 / 2), p = (x, y) / z -- and the
// identity where r < 1e-3 or d0 < 1e-7.  d(dx, dy)/d(px, py) = f I + q p p^T with q = f'(r) / r = (d2 / (d0 (1 + d2^2 r^2)) - f) / r^2.
// Round 6: the gradient pass contracts (gx, gy) through these two scalars like the other models do through theirs; until round 5
// it pushed three partials through every operation (Dual3, 164-170 VGPRs: three waves per SIMD).
template <typename real>
struct AtanCore {
  real iz, px, py, f, q;
};
template <typename real>
NID_HD AtanCore<real> atan_core(const CamParams<real>& c, real x, real y, real z) {
  AtanCore<real> k;
  k.iz = fast_rcp(z);
  k.px = x * k.iz;
  k.py = y * k.iz;
  const real r2 = fma(k.px, k.px, k.py * k.py);
  const real r = m_sqrt(r2);
  const real d0 = c.dist[0];
  k.f = real(1);
  k.q = real(0);
  if (!(r < real(1e-3) || d0 < real(1e-7))) {
    const real d1 = c.dist[kAtanD1], d2 = c.dist[kAtanD2];
    const real ir2 = fast_rcp(r2);
    k.f = d1 * fast_atan2(r * d2, real(1)) * fast_rcp(r);  // the value pass's own factor (project<MODEL_ATAN, FAST>)
    k.q = fma(d1 * d2, fast_rcp(fma(d2 * d2, r2, real(1))), -k.f) * ir2;
  }
  return k;
}

// what the wide-angle models' Jacobians need beyond their cores
//   fisheye: q = (theta_d' |z| / |p|^2 - s) / r^2 and wz = -theta_d' sgn(z) / |p|^2 (see project_jac)
template <typename real>
NID_HD void fisheye_partials(const CamParams<real>& c, const FisheyeCore<real>& k, real z, real& q, real& wz) {
  const real k1 = c.dist[0], k2 = c.dist[1], k3 = c.dist[2], k4 = c.dist[3];
  const real dtheta_d = fma(k.th2, fma(k.th2, fma(k.th2, fma(k.th2, real(9) * k4, real(7) * k3), real(5) * k2), real(3) * k1), real(1));
  const real in2 = fast_rcp(fma(z, z, k.r2));
  q = fma(dtheta_d * k.az, in2, -k.s) * (k.ir * k.ir);
  wz = (z < real(0) ? dtheta_d : -dtheta_d) * in2;
}
//   equirectangular: du = ku (z, 0, -x), dv = (t x, kvr, t z) with ku = W / (2 pi rho^2), kvr = H rho / (pi |p|^2),
//   t = -H y / (pi |p|^2 rho); all zero for a point the value sends to the image centre (:16)
template <typename real>
NID_HD void equirect_partials(const CamParams<real>& c, const EquirectCore<real>& k, real y, real& ku, real& kvr, real& t) {
  const real in2 = fast_rcp(fma(y, y, k.rho2));
  const real kv = k.tiny ? real(0) : (c.intr[1] * real(0.31830988618379067154)) * in2;
  ku = k.tiny ? real(0) : (c.intr[0] * real(0.15915494309189533577)) * (k.irho * k.irho);
  kvr = kv * (k.rho2 * k.irho);
  t = -(kv * y) * k.irho;
}

// Jacobian of the radial-tangential distortion shared by plumb_bob / rational_polynomial / omnidir (radtan_partials
// above; it is symmetric up to fx / fy): a = diag(fx, fy) d(dx, dy)/d(px, py).  rc and the powers of r2 are the value
// pass's own expressions (project<..., FAST>), so that after inlining they are computed once.
template <int MODEL, typename real>
NID_HD void radtan_jac(const CamParams<real>& c, real px, real py, real& a00, real& a01, real& a10, real& a11) {
  const real x2 = px * px, y2 = py * py;
  const real r2 = fma(px, px, y2);
  const real r4 = r2 * r2;
  const real p1 = c.dist[2], p2 = c.dist[3];
  real rc, rc2;  // radial factor and 2 d(rc)/d(r2)
  if (MODEL == MODEL_PLUMB_BOB) {
    const real k1 = c.dist[0], k2 = c.dist[1], k3 = c.dist[4];
    const real r6 = r2 * r4;
    rc = fma(k3, r6, fma(k2, r4, fma(k1, r2, real(1))));
    rc2 = fma(real(6) * k3, r4, fma(real(4) * k2, r2, real(2) * k1));
  } else if (MODEL == MODEL_OMNIDIR) {
    const real k1 = c.dist[0], k2 = c.dist[1];
    rc = fma(k2, r4, fma(k1, r2, real(1)));
    rc2 = fma(real(4) * k2, r2, real(2) * k1);
  } else {
    const real k1 = c.dist[0], k2 = c.dist[1], k3 = c.dist[4], k4 = c.dist[5], k5 = c.dist[6], k6 = c.dist[7];
    const real r6 = r2 * r4;
    const real num = fma(k3, r6, fma(k2, r4, fma(k1, r2, real(1))));
    const real den = fma(k6, r6, fma(k5, r4, fma(k4, r2, real(1))));
    const real nump = fma(real(3) * k3, r4, fma(real(2) * k2, r2, k1));
    const real denp = fma(real(3) * k6, r4, fma(real(2) * k5, r2, k4));
    if (den > real(1e-8)) {
      const real id = real(1) / den;
      rc = num / den;  // the value pass's own quotient (project<MODEL_RATIONAL>)
      rc2 = real(2) * ((nump - rc * denp) * id);
    } else {
      rc = num;
      rc2 = real(2) * nump;
    }
  }
  real b00, off, b11;
  radtan_partials<real>(p1, p2, px, py, x2, y2, rc, rc2, b00, off, b11);
  a00 = c.intr[0] * b00;
  a01 = c.intr[0] * off;
  a10 = c.intr[1] * off;
  a11 = c.intr[1] * b11;
}

// projection value + 2x3 Jacobian d(u,v)/d(x,y,z) for the gradient pass.  The value is the same
// expression as in the histogram pass (both passes agree on every knot).  The Jacobians of the five
// models the configs use are written out by hand -- 2.5x (plumb_bob) to ~6x (equirectangular) fewer
// fp64 operations than pushing three partials through every operation; `atan` keeps the generic
// Dual3 forward mode.  Same chain rules as the reference's Jets (a1-a10 in SURVEY.md section 8).
template <int MODEL, typename real>
NID_HD void project_jac(const CamParams<real>& c, real x, real y, real z, real& u, real& v, real* du, real* dv) {
  if (MODEL == MODEL_PLUMB_BOB || MODEL == MODEL_RATIONAL) {
    project<MODEL, real, real, true>(c, x, y, z, u, v);
    const real iz = fast_rcp(z);
    const real px = x * iz, py = y * iz;
    real a00, a01, a10, a11;
    radtan_jac<MODEL, real>(c, px, py, a00, a01, a10, a11);
    du[0] = a00 * iz;
    du[1] = a01 * iz;
    du[2] = -fma(du[0], px, du[1] * py);
    dv[0] = a10 * iz;
    dv[1] = a11 * iz;
    dv[2] = -fma(dv[0], px, dv[1] * py);
  } else if (MODEL == MODEL_OMNIDIR) {
    // m = p_xy / den, den = p_z + xi |p|:  d(m)/d(p) = ([I2 | 0] - m (e_z + xi p / |p|)^T) / den
    project<MODEL, real, real, true>(c, x, y, z, u, v);
    const OmnidirCore<real> k = omnidir_core<real>(c, x, y, z);
    real a00, a01, a10, a11;
    radtan_jac<MODEL, real>(c, k.ux, k.uy, a00, a01, a10, a11);
    // an un-normalised point (|p| = 0, omnidir.hpp:19) has den = xi, a constant: the xi p / |p| term vanishes with p
    const real xin = c.intr[4] * k.in;
    {
      const real b0 = a00 * k.iden, b1 = a01 * k.iden, t = fma(b0, k.ux, b1 * k.uy);
      du[0] = fma(-(t * xin), x, b0);
      du[1] = fma(-(t * xin), y, b1);
      du[2] = fma(-(t * xin), z, -t);
    }
    {
      const real b0 = a10 * k.iden, b1 = a11 * k.iden, t = fma(b0, k.ux, b1 * k.uy);
      dv[0] = fma(-(t * xin), x, b0);
      dv[1] = fma(-(t * xin), y, b1);
      dv[2] = fma(-(t * xin), z, -t);
    }
  } else if (MODEL == MODEL_FISHEYE) {
    //   d(s x)/dx = s + x^2 q,  d(s x)/dy = x y q,  d(s x)/dz = -x theta_d' sgn(z) / |p|^2,
    //   q = (theta_d' |z| / |p|^2 - s) / r^2
    project<MODEL, real, real, true>(c, x, y, z, u, v);
    const FisheyeCore<real> k = fisheye_core<real>(c, x, y, z);
    real q, wz;
    fisheye_partials<real>(c, k, z, q, wz);
    const real fx = c.intr[0], fy = c.intr[1];
    const real xq = x * q, yq = y * q;
    du[0] = fx * fma(x, xq, k.s);
    du[1] = fx * (x * yq);
    du[2] = fx * (x * wz);
    dv[0] = fy * (y * xq);
    dv[1] = fy * fma(y, yq, k.s);
    dv[2] = fy * (y * wz);
  } else if (MODEL == MODEL_EQUIRECT) {
    // u = W (1/2 + atan2(x, z) / 2 pi),  v = H (1/2 + atan2(y, rho) / pi),  rho = |(x, z)|
    const EquirectCore<real> k = equirect_core<real>(c, x, y, z, u, v);
    real ku, kvr, t;
    equirect_partials<real>(c, k, y, ku, kvr, t);
    du[0] = ku * z;
    du[1] = real(0);
    du[2] = -ku * x;
    dv[0] = t * x;
    dv[1] = kvr;
    dv[2] = t * z;
  } else if (MODEL == MODEL_ATAN) {
    project<MODEL, real, real, true>(c, x, y, z, u, v);  // value: the histogram pass's own expression
    const AtanCore<real> k = atan_core<real>(c, x, y, z);
    const real fx = c.intr[0] * k.iz, fy = c.intr[1] * k.iz;
    const real qxy = k.q * (k.px * k.py);
    du[0] = fx * fma(k.q * k.px, k.px, k.f);
    du[1] = fx * qxy;
    du[2] = -fma(du[0], k.px, du[1] * k.py);
    dv[0] = fy * qxy;
    dv[1] = fy * fma(k.q * k.py, k.py, k.f);
    dv[2] = -fma(dv[0], k.px, dv[1] * k.py);
  } else {
    typedef Dual3<real> D;
    D uu, vv;
    project<MODEL, real, real, true>(c, x, y, z, u, v);  // value: the histogram pass's own expression
    project<MODEL, D, real, true>(c, D(x, real(1), real(0), real(0)), D(y, real(0), real(1), real(0)), D(z, real(0), real(0), real(1)), uu, vv);
    du[0] = uu.d0;
    du[1] = uu.d1;
    du[2] = uu.d2;
    dv[0] = vv.d0;
    dv[1] = vv.d1;
    dv[2] = vv.d2;
  }
}

// Gradient pass, two halves around the tap loop.  project_fwd: the projected point (the histogram pass's own
// expression, so both passes agree on every knot) plus what the backward half needs; project_bwd: the
// vector-Jacobian product gp = (gx, gy) . d(u, v)/d(x, y, z).  No model forms its 2x3 Jacobian here (round 5; until round 4
// only the pinhole family did not) -- each contracts (gx, gy) through the factors of its chain rule:
//   pinhole family   A = d(dx, dy)/d(px, py) (symmetric off-diagonal):  h = A^T (fx gx, fy gy),  gp = (h0 / z, h1 / z, -(gp0 px + gp1 py))
//   omnidir          m = p_xy / den, den = p_z + xi |p|:  b = (A / den)^T (fx gx, fy gy), t = b . m,
//                    gp = (b0, b1, -t) - xi t p / |p| = (b0 - cd ux, b1 - cd uy, (xi^2 - 1) t - cd),  cd = xi t den / |p|
//                    (p = den m in x and y, p_z = den - xi |p|): nothing of the camera-frame point stays alive across the taps
//   fisheye          g' = (fx gx, fy gy), k = x g'0 + y g'1:  gp = (s g'0 + x q k, s g'1 + y q k, wz k)
//   equirectangular  gp = (a z + b x, gy kvr, b z - a x),  a = gx ku, b = gy t
//   atan (FOV)       g' = (fx gx, fy gy), k = px g'0 + py g'1:  h = f g' + q k (px, py),  gp = (h0 / z, h1 / z, -(gp0 px + gp1 py))   (round 6)
// -- 10 / 14 / 10 / 7 / 11 operations where the explicit Jacobian and its contraction took 18 / 36 / 20 / 14.  A model without a
// branch of its own (none of the six today) would fall back to the generic 2x3 Jacobian (project_jac) contracted here.
template <typename real>
struct ProjCtx {
  real a[6];  // pinhole family: iz, px, py, A00, off, A11;  omnidir: (A00, off, A11) / den, ux, uy, xi den / |p|;
              // fisheye: s, q, wz, x, y;  equirectangular: ku, kvr, t, x, z;  atan: iz, px, py, f, q
};
template <int MODEL, typename real>
NID_HD void project_fwd(const CamParams<real>& c, real x, real y, real z, real& u, real& v, ProjCtx<real>& ctx) {
  if (MODEL == MODEL_PLUMB_BOB || MODEL == MODEL_RATIONAL) {
    project<MODEL, real, real, true>(c, x, y, z, u, v);
    const real iz = fast_rcp(z);
    const real px = x * iz, py = y * iz;
    real a00, a01, a10, a11;
    CamParams<real> unit = c;  // the distortion Jacobian without the focal lengths (they scale gx, gy instead)
    unit.intr[0] = real(1);
    unit.intr[1] = real(1);
    radtan_jac<MODEL, real>(unit, px, py, a00, a01, a10, a11);
    ctx.a[0] = iz;
    ctx.a[1] = px;
    ctx.a[2] = py;
    ctx.a[3] = a00;
    ctx.a[4] = a01;  // == a10
    ctx.a[5] = a11;
  } else if (MODEL == MODEL_OMNIDIR) {
    project<MODEL, real, real, true>(c, x, y, z, u, v);
    const OmnidirCore<real> k = omnidir_core<real>(c, x, y, z);
    real a00, a01, a10, a11;
    CamParams<real> unit = c;
    unit.intr[0] = real(1);
    unit.intr[1] = real(1);
    radtan_jac<MODEL, real>(unit, k.ux, k.uy, a00, a01, a10, a11);
    ctx.a[0] = a00 * k.iden;  // (six values stay alive across the taps, like every other model: a seventh spilled in the looped kernels)
    ctx.a[1] = a01 * k.iden;
    ctx.a[2] = a11 * k.iden;
    ctx.a[3] = k.ux;
    ctx.a[4] = k.uy;
    // xi den / |p|; for |p| = 0 (den = xi, a constant) the term it multiplies has to vanish: in = 1, n = 1, but t = 0 there
    // (m = 0), so whatever this is, cd = 0
    ctx.a[5] = c.intr[4] * (fma(c.intr[4], k.n, z) * k.in);
  } else if (MODEL == MODEL_FISHEYE) {
    project<MODEL, real, real, true>(c, x, y, z, u, v);
    const FisheyeCore<real> k = fisheye_core<real>(c, x, y, z);
    real q, wz;
    fisheye_partials<real>(c, k, z, q, wz);
    ctx.a[0] = k.s;
    ctx.a[1] = q;
    ctx.a[2] = wz;
    ctx.a[3] = x;
    ctx.a[4] = y;
  } else if (MODEL == MODEL_EQUIRECT) {
    const EquirectCore<real> k = equirect_core<real>(c, x, y, z, u, v);
    equirect_partials<real>(c, k, y, ctx.a[0], ctx.a[1], ctx.a[2]);
    ctx.a[3] = x;
    ctx.a[4] = z;
  } else if (MODEL == MODEL_ATAN) {
    project<MODEL, real, real, true>(c, x, y, z, u, v);
    const AtanCore<real> k = atan_core<real>(c, x, y, z);
    ctx.a[0] = k.iz;
    ctx.a[1] = k.px;
    ctx.a[2] = k.py;
    ctx.a[3] = k.f;
    ctx.a[4] = k.q;
  } else {
    project_jac<MODEL, real>(c, x, y, z, u, v, ctx.a, ctx.a + 3);
  }
}
template <int MODEL, typename real>
NID_HD void project_bwd(const CamParams<real>& c, const ProjCtx<real>& ctx, real gx, real gy, real* gp) {
  if (MODEL == MODEL_PLUMB_BOB || MODEL == MODEL_RATIONAL) {
    const real gxf = gx * c.intr[0], gyf = gy * c.intr[1];
    const real h0 = fma(gxf, ctx.a[3], gyf * ctx.a[4]);
    const real h1 = fma(gxf, ctx.a[4], gyf * ctx.a[5]);
    gp[0] = h0 * ctx.a[0];
    gp[1] = h1 * ctx.a[0];
    gp[2] = -fma(gp[0], ctx.a[1], gp[1] * ctx.a[2]);
  } else if (MODEL == MODEL_OMNIDIR) {
    const real xi = c.intr[4];
    const real gxf = gx * c.intr[0], gyf = gy * c.intr[1];
    const real b0 = fma(gxf, ctx.a[0], gyf * ctx.a[1]);
    const real b1 = fma(gxf, ctx.a[1], gyf * ctx.a[2]);
    const real t = fma(b0, ctx.a[3], b1 * ctx.a[4]);
    const real cd = t * ctx.a[5];
    gp[0] = fma(-cd, ctx.a[3], b0);
    gp[1] = fma(-cd, ctx.a[4], b1);
    gp[2] = fma(t, fma(xi, xi, real(-1)), -cd);
  } else if (MODEL == MODEL_FISHEYE) {
    const real g0 = gx * c.intr[0], g1 = gy * c.intr[1];
    const real k = fma(ctx.a[3], g0, ctx.a[4] * g1);
    const real kq = k * ctx.a[1];
    gp[0] = fma(ctx.a[3], kq, ctx.a[0] * g0);
    gp[1] = fma(ctx.a[4], kq, ctx.a[0] * g1);
    gp[2] = ctx.a[2] * k;
  } else if (MODEL == MODEL_EQUIRECT) {
    const real a = gx * ctx.a[0], b = gy * ctx.a[2];
    gp[0] = fma(a, ctx.a[4], b * ctx.a[3]);
    gp[1] = gy * ctx.a[1];
    gp[2] = fma(b, ctx.a[4], -(a * ctx.a[3]));
  } else if (MODEL == MODEL_ATAN) {
    const real g0 = gx * c.intr[0], g1 = gy * c.intr[1];
    const real kq = fma(ctx.a[1], g0, ctx.a[2] * g1) * ctx.a[4];
    gp[0] = fma(kq, ctx.a[1], ctx.a[3] * g0) * ctx.a[0];
    gp[1] = fma(kq, ctx.a[2], ctx.a[3] * g1) * ctx.a[0];
    gp[2] = -fma(gp[0], ctx.a[1], gp[1] * ctx.a[2]);
  } else {
    gp[0] = fma(gx, ctx.a[0], gy * ctx.a[3]);
    gp[1] = fma(gx, ctx.a[1], gy * ctx.a[4]);
    gp[2] = fma(gx, ctx.a[2], gy * ctx.a[5]);
  }
}

// ------------------------------------------------------------------------------------------
// device point records (written once per handle by the device-side build, nid_build.hip)
struct Rec32 {  // 16 B: one PLY record (float xyz) + the pose-independent histogram column
  float x, y, z;
  uint32_t bin;
};
struct Rec64 {  // 32 B: used only when the caller's doubles do not round-trip through float
  double x, y, z;
  uint64_t bin;
};

template <typename real>
__device__ __forceinline__ void load_rec(const Rec32* p, real& x, real& y, real& z, uint32_t& bin) {
  const float4 v = *reinterpret_cast<const float4*>(p);  // one 16 B/lane coalesced load
  x = real(v.x);
  y = real(v.y);
  z = real(v.z);
  bin = __float_as_uint(v.w);
}
template <typename real>
__device__ __forceinline__ void load_rec(const Rec64* p, real& x, real& y, real& z, uint32_t& bin) {
  const double2 a = reinterpret_cast<const double2*>(p)[0];
  const double2 b = reinterpret_cast<const double2*>(p)[1];
  x = real(a.x);
  y = real(a.y);
  z = real(b.x);
  bin = uint32_t(__double_as_longlong(b.y));
}

// kUnroll-deep batch of RAW records: what the loads return, not yet converted -- so that a batch can be in flight across a
// workgroup's prologue (tile zeroing, entropy tail, G tile) without the conversion forcing the wait (nid_kernels.hpp)
template <typename Rec, int N>
struct RawBatch;
template <int N>
struct RawBatch<Rec32, N> {
  float4 v[N];
  __device__ __forceinline__ void load(const char* rec_base, uint32_t byte_off, int k) { v[k] = *reinterpret_cast<const float4*>(rec_base + size_t(byte_off)); }
  template <typename real>
  __device__ __forceinline__ void get(int k, real& x, real& y, real& z, uint32_t& bin) const {
    x = real(v[k].x), y = real(v[k].y), z = real(v[k].z), bin = __float_as_uint(v[k].w);
  }
  __device__ __forceinline__ void store(int k, Rec32* dst) const { *reinterpret_cast<float4*>(dst) = v[k]; }  // the raw record, as loaded (nid_fused.hpp: into LDS)
};
template <int N>
struct RawBatch<Rec64, N> {
  double2 a[N], b[N];
  __device__ __forceinline__ void load(const char* rec_base, uint32_t byte_off, int k) {
    a[k] = reinterpret_cast<const double2*>(rec_base + size_t(byte_off))[0];
    b[k] = reinterpret_cast<const double2*>(rec_base + size_t(byte_off))[1];
  }
  template <typename real>
  __device__ __forceinline__ void get(int k, real& x, real& y, real& z, uint32_t& bin) const {
    x = real(a[k].x), y = real(a[k].y), z = real(b[k].x), bin = uint32_t(__double_as_longlong(b[k].y));
  }
  __device__ __forceinline__ void store(int k, Rec64* dst) const {
    reinterpret_cast<double2*>(dst)[0] = a[k];
    reinterpret_cast<double2*>(dst)[1] = b[k];
  }
};

struct Chunk {  // one workgroup's slice of the bucketed cloud
  uint32_t start;
  uint32_t count;
  uint32_t group;  // histogram columns [group*GW, (group+1)*GW)
  uint32_t pad;
};

// uniform cubic B-spline basis, the reference's 4x4 coefficient matrix / 6 (nid_cost.hpp:29-33), written in the
// two mirror variables s and t = 1 - s (b2(s) = b1(t), b3(s) = b0(t)) and WITHOUT the 1/6 -- bspline6 returns 6 b:
//   6 b0 = t^3,  6 b1 = 4 - 6 s^2 + 3 s^3,  6 b2 = 4 - 6 t^2 + 3 t^3,  6 b3 = s^3
// -- 9 operations for the four weights; bspline_deriv2 returns 2 db/ds = (-t^2, s (3 s - 4), t (4 - 3 t), s^2), 4 more
// (s^2, t^2 are shared; the sign of -t^2 folds into the consuming fma).  The consumers carry the constant factors once
// per workgroup instead of once per point: the histogram pass in the fixed-point unit of the x-weights
// (BsplineScale), the gradient pass in its G tile (1/12 = 1/6 * 1/2 for both gx and gy).
// Every operation is an explicit mul / fma: all translation units are built with -ffp-contract=off so that a point
// gets the same arithmetic whichever unrolled slot / chunk / GPU processes it (the histogram is bit-identical across
// tilings).  Every weight is >= +0 by construction (products of non-negative factors; 6 b1, 6 b2 >= 1): the
// subnormal fixed-point trick (to_fixed_dn) needs sign bit 0 on every weight.
template <typename real>
NID_HD void bspline6(real s, real* b) {
  const real t = real(1) - s;
  const real s2 = s * s, t2 = t * t;
  b[0] = t2 * t;
  b[1] = fma(s2, fma(real(3), s, real(-6)), real(4));
  b[2] = fma(t2, fma(real(3), t, real(-6)), real(4));
  b[3] = s2 * s;
}
template <typename real>
NID_HD void bspline_deriv2(real s, real* d) {
  const real t = real(1) - s;
  d[0] = -(t * t);
  d[1] = s * fma(real(3), s, real(-4));
  d[2] = t * fma(real(-3), t, real(4));
  d[3] = s * s;
}

// The x-weights of the histogram pass come out of the polynomial ALREADY in fixed-point units: its constants are
// pre-multiplied by U/36 (one set per kernel, uniform -- or zeroed per lane for an outlier / padding slot, which
// then adds exact zeros), so b'[a] = 6 bx[a] * U/36 needs no multiply of its own and bits(b'[a] * 6 by[b]) is the
// integer weight bx by U (to_fixed_dn).  All of this arithmetic happens in the SUBNORMAL range, i.e. on the integer
// grid of 2^-1074: the constants are k, 3k, 4k, 6k grid steps with k = round(2^frac / 36), so they are exact, the
// fixed-point unit is U = 36k (within 18 of 2^frac; nidreg_internal.hpp fixed_unit) and no constant carries a rounding bias
// into the histogram; each operation rounds to the grid (a few grid steps per weight after the multiplication by
// 6 by <= 4 -- unbiased, deterministic, order independent; bound checked by tests/cxx/test_device_math.cpp).
struct BsplineScale {
  double k16, k46, k05, k1;  // k, 4k, 3k, 6k grid steps as subnormal doubles, k = U/36
};
NID_HD BsplineScale bspline_scale(double k16) {
  BsplineScale K;
  K.k16 = k16;
  K.k46 = 4.0 * k16;  // exact: small integers times a grid value
  K.k05 = 3.0 * k16;
  K.k1 = 6.0 * k16;
  return K;
}
NID_HD void bspline_scaled(double s, const BsplineScale& K, double* b) {
  const double t = 1.0 - s;
  const double s2 = s * s, t2 = t * t;
  b[0] = t2 * (t * K.k16);
  b[1] = fma(s2, fma(K.k05, s, -K.k1), K.k46);
  b[2] = fma(t2, fma(K.k05, t, -K.k1), K.k46);
  b[3] = s2 * (s * K.k16);
}

// p_cam = R p + t with fused multiply-adds (SPLINE kernels)
template <typename real>
NID_HD void transform_fma(const PoseParams<real>& pose, real x, real y, real z, real& cx, real& cy, real& cz) {
  cx = fma(pose.R[2], z, fma(pose.R[1], y, fma(pose.R[0], x, pose.t[0])));
  cy = fma(pose.R[5], z, fma(pose.R[4], y, fma(pose.R[3], x, pose.t[1])));
  cz = fma(pose.R[8], z, fma(pose.R[7], y, fma(pose.R[6], x, pose.t[2])));
}

// weight -> unsigned fixed point in ONE instruction: the x-weights are born in grid steps of 2^-1074 (bspline_scaled),
// so the product bxs * by6 is a SUBNORMAL double whose bit pattern (exponent field 0) IS the integer
// round-to-nearest(bx by U) -- no magic add, no mask.  gfx950 handles fp64 denormals at full rate.
NID_HD u64 to_fixed_dn(double bx_scaled, double by) {
#if defined(__HIP_DEVICE_COMPILE__)
  return u64(__double_as_longlong(bx_scaled * by));
#else
  const double p = bx_scaled * by;  // host build (tests): same IEEE product, subnormals honoured by default on x86-64
  u64 bits;
  __builtin_memcpy(&bits, &p, sizeof bits);
  return bits;
#endif
}


// The bin image is stored in STRIPS of four rows with the four vertically adjacent pixels of a
// column contiguous: byte address of padded pixel (x, y) = (y >> 2) * 4 * pitch + 4 * x + (y & 3).
// The 4x4 tap patch [kx..kx+3] x [ky..ky+3] is then two 16-byte loads (one per strip it touches)
// instead of four row gathers -- half the L1 lookups, which co-limited both spline kernels -- and a
// v_alignbyte per column puts rows ky..ky+3 of column kx+a into the four bytes of cols[a].
struct __attribute__((aligned(4))) StripQuad {
  uint32_t c[4];
};
// v_alignbyte_b32: bytes [sh, sh + 3] of the 8-byte value hi:lo
NID_HD uint32_t align_bytes(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbyte(hi, lo, sh);
#else
  return uint32_t(((uint64_t(hi) << 32) | lo) >> (8u * (sh & 3u)));
#endif
}
// 24-bit multiply (v_mul_u32_u24, full rate; v_mul_lo_u32 is quarter rate): strip index < 2^24, strip bytes < 2^24
NID_HD uint32_t mul24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return a * b;
#endif
}
NID_HD void load_patch(const uint8_t* __restrict__ img, int pitch, int kx, int ky, uint32_t* cols) {
  const uint32_t stride = uint32_t(pitch) * 4u;
  const uint32_t base = mul24(uint32_t(ky) >> 2, stride) + uint32_t(kx) * 4u;
  // both loads: uniform base (SGPR pair) + the same 32-bit lane offset -- no 64-bit address arithmetic per point
  const uint8_t* __restrict__ img_next = img + stride;
  const StripQuad s0 = *reinterpret_cast<const StripQuad*>(img + base);
  const StripQuad s1 = *reinterpret_cast<const StripQuad*>(img_next + base);
  const uint32_t sh = uint32_t(ky) & 3u;
#pragma unroll
  for (int a = 0; a < 4; a++) cols[a] = align_bytes(s1.c[a], s0.c[a], sh);
}
NID_HD uint32_t load_pixel(const uint8_t* __restrict__ img, int pitch, int x, int y) {
  return img[(uint32_t(y) >> 2) * uint32_t(pitch) * 4u + uint32_t(x) * 4u + (uint32_t(y) & 3u)];
}

}  // namespace nidreg
