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

namespace nidreg {

typedef unsigned long long u64;

// ------------------------------------------------------------------------------------------
// scalar helpers for float / double
__device__ __forceinline__ float m_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double m_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float m_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double m_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float m_asin(float x) { return asinf(x); }
__device__ __forceinline__ double m_asin(double x) { return asin(x); }
__device__ __forceinline__ float m_atan(float x) { return atanf(x); }
__device__ __forceinline__ double m_atan(double x) { return atan(x); }
__device__ __forceinline__ float m_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double m_abs(double x) { return fabs(x); }
__device__ __forceinline__ float m_floor(float x) { return floorf(x); }
__device__ __forceinline__ double m_floor(double x) { return floor(x); }
__device__ __forceinline__ float m_val(float x) { return x; }
__device__ __forceinline__ double m_val(double x) { return x; }

// ------------------------------------------------------------------------------------------
// forward-mode dual number with three partials (d/dx, d/dy, d/dz of the camera-frame point).
// Same chain rules as the reference's ceres::Jet arithmetic, three slots instead of seven: the
// remaining 3x7 factor d(p_cam)/d(pose) is linear in the LiDAR point and is folded into a
// 3x3 + 3 accumulator (see k_spline_grad).
template <typename real>
struct Dual3 {
  real a, d0, d1, d2;
  __device__ __forceinline__ Dual3() {}
  __device__ __forceinline__ Dual3(real v) : a(v), d0(0), d1(0), d2(0) {}
  __device__ __forceinline__ Dual3(real v, real x, real y, real z) : a(v), d0(x), d1(y), d2(z) {}
};

template <typename real>
__device__ __forceinline__ real m_val(const Dual3<real>& x) { return x.a; }

#define NID_D Dual3<real>
template <typename real> __device__ __forceinline__ NID_D operator+(const NID_D& f, const NID_D& g) { return NID_D(f.a + g.a, f.d0 + g.d0, f.d1 + g.d1, f.d2 + g.d2); }
template <typename real> __device__ __forceinline__ NID_D operator+(const NID_D& f, real s) { return NID_D(f.a + s, f.d0, f.d1, f.d2); }
template <typename real> __device__ __forceinline__ NID_D operator+(real s, const NID_D& f) { return NID_D(s + f.a, f.d0, f.d1, f.d2); }
template <typename real> __device__ __forceinline__ NID_D operator-(const NID_D& f, const NID_D& g) { return NID_D(f.a - g.a, f.d0 - g.d0, f.d1 - g.d1, f.d2 - g.d2); }
template <typename real> __device__ __forceinline__ NID_D operator-(const NID_D& f, real s) { return NID_D(f.a - s, f.d0, f.d1, f.d2); }
template <typename real> __device__ __forceinline__ NID_D operator-(real s, const NID_D& f) { return NID_D(s - f.a, -f.d0, -f.d1, -f.d2); }
template <typename real> __device__ __forceinline__ NID_D operator-(const NID_D& f) { return NID_D(-f.a, -f.d0, -f.d1, -f.d2); }
template <typename real> __device__ __forceinline__ NID_D operator*(const NID_D& f, const NID_D& g) {
  return NID_D(f.a * g.a, fma(f.a, g.d0, f.d0 * g.a), fma(f.a, g.d1, f.d1 * g.a), fma(f.a, g.d2, f.d2 * g.a));
}
template <typename real> __device__ __forceinline__ NID_D operator*(const NID_D& f, real s) { return NID_D(f.a * s, f.d0 * s, f.d1 * s, f.d2 * s); }
template <typename real> __device__ __forceinline__ NID_D operator*(real s, const NID_D& f) { return NID_D(f.a * s, f.d0 * s, f.d1 * s, f.d2 * s); }
template <typename real> __device__ __forceinline__ NID_D operator/(const NID_D& f, const NID_D& g) {
  const real gi = real(1) / g.a;
  const real q = f.a * gi;
  return NID_D(q, fma(-q, g.d0, f.d0) * gi, fma(-q, g.d1, f.d1) * gi, fma(-q, g.d2, f.d2) * gi);
}
template <typename real> __device__ __forceinline__ NID_D operator/(const NID_D& f, real s) {
  const real si = real(1) / s;
  return NID_D(f.a * si, f.d0 * si, f.d1 * si, f.d2 * si);
}
template <typename real> __device__ __forceinline__ bool operator<(const NID_D& f, real s) { return f.a < s; }
template <typename real> __device__ __forceinline__ bool operator>(const NID_D& f, real s) { return f.a > s; }
template <typename real> __device__ __forceinline__ NID_D m_sqrt(const NID_D& f) {
  const real t = m_sqrt(f.a);
  const real k = real(1) / (real(2) * t);
  return NID_D(t, f.d0 * k, f.d1 * k, f.d2 * k);
}
template <typename real> __device__ __forceinline__ NID_D m_atan2(const NID_D& g, const NID_D& f) {
  const real k = real(1) / fma(f.a, f.a, g.a * g.a);
  return NID_D(m_atan2(g.a, f.a), k * fma(-g.a, f.d0, f.a * g.d0), k * fma(-g.a, f.d1, f.a * g.d1), k * fma(-g.a, f.d2, f.a * g.d2));
}
template <typename real> __device__ __forceinline__ NID_D m_asin(const NID_D& f) {
  const real k = real(1) / m_sqrt(real(1) - f.a * f.a);
  return NID_D(m_asin(f.a), k * f.d0, k * f.d1, k * f.d2);
}
template <typename real> __device__ __forceinline__ NID_D m_atan(const NID_D& f) {
  const real k = real(1) / (real(1) + f.a * f.a);
  return NID_D(m_atan(f.a), k * f.d0, k * f.d1, k * f.d2);
}
template <typename real> __device__ __forceinline__ NID_D m_abs(const NID_D& f) {
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

// ------------------------------------------------------------------------------------------
// a*b + c: one fused multiply-add in the FAST (SPLINE) instantiation when all operands are plain
// scalars, an unfused multiply then add otherwise (NEAREST path: bit-identical to the CPU, which has
// no fma; Dual3 operands: their operators).  Every use below nests the terms so that the unfused
// form reproduces the reference's left-to-right association exactly (fp addition is commutative).
template <bool FAST, typename A, typename B, typename C>
__device__ __forceinline__ auto mad(const A& a, const B& b, const C& c) -> decltype(a * b + c) {
  if constexpr (FAST && std::is_floating_point<A>::value && std::is_floating_point<B>::value && std::is_floating_point<C>::value) {
    return fma(a, b, c);
  } else {
    return a * b + c;
  }
}

// 1/z by v_rcp_f64 + two Newton steps (5 instructions instead of the ~12 of the IEEE division
// sequence); |z| is a camera-frame depth in metres, never denormal / inf in range of interest, and a
// NaN / zero z still yields a NaN / inf projection, i.e. an outlier.
__device__ __forceinline__ double fast_rcp(double z) {
  double r = __builtin_amdgcn_rcp(z);
  r = fma(fma(-z, r, 1.0), r, r);
  r = fma(fma(-z, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float fast_rcp(float z) { return 1.0f / z; }

// perspective division: exact x/z, y/z (NEAREST path) or one reciprocal and two multiplies (SPLINE
// kernels; both passes use the same form, so they agree on every knot)
template <bool FAST, typename real>
__device__ __forceinline__ void persp(real x, real y, real z, real& px, real& py) {
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
__device__ __forceinline__ void persp(const Dual3<real>& x, const Dual3<real>& y, const Dual3<real>& z, Dual3<real>& px, Dual3<real>& py) {
  px = x / z;  // Dual3 division is reciprocal-multiply already
  py = y / z;
}

// projection models (reference: include/camera/{pinhole,fisheye,omnidir,equirectangular,atan,
// rational_polynomial}.hpp), T = real or Dual3<real>.
template <int MODEL, typename T, typename real, bool FAST = false>
__device__ __forceinline__ void project(const CamParams<real>& c, const T& x, const T& y, const T& z, T& u, T& v) {
  if (MODEL == MODEL_PLUMB_BOB) {  // pinhole.hpp:13-51, distortion k1 k2 p1 p2 k3
    const real k1 = c.dist[0], k2 = c.dist[1], p1 = c.dist[2], p2 = c.dist[3], k3 = c.dist[4];
    T px, py;
    persp<FAST>(x, y, z, px, py);
    const T x2 = px * px, y2 = py * py;
    const T r2 = mad<FAST>(px, px, y2);                                   // x2 + y2
    const T r4 = r2 * r2;
    const T r6 = r2 * r4;
    const T rc = mad<FAST>(k3, r6, mad<FAST>(k2, r4, mad<FAST>(k1, r2, real(1))));  // 1 + k1 r2 + k2 r4 + k3 r6
    const T t1 = real(2) * px * py;
    const T t2 = mad<FAST>(real(2), x2, r2);                              // r2 + 2 x2
    const T t3 = mad<FAST>(real(2), y2, r2);
    const T dx = mad<FAST>(p2, t2, mad<FAST>(p1, t1, rc * px));           // rc px + p1 t1 + p2 t2
    const T dy = mad<FAST>(p2, t1, mad<FAST>(p1, t3, rc * py));           // rc py + p1 t3 + p2 t1
    u = mad<FAST>(c.intr[0], dx, c.intr[2]);
    v = mad<FAST>(c.intr[1], dy, c.intr[3]);
  } else if (MODEL == MODEL_FISHEYE) {  // fisheye.hpp:14-36 (abs(z) at :16)
    const real k1 = c.dist[0], k2 = c.dist[1], k3 = c.dist[2], k4 = c.dist[3];
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
  } else if (MODEL == MODEL_OMNIDIR) {  // omnidir.hpp:14-41
    const real xi = c.intr[4];
    const real k1 = c.dist[0], k2 = c.dist[1], p1 = c.dist[2], p2 = c.dist[3];
    const T n2 = mad<FAST>(z, z, mad<FAST>(y, y, x * x));                 // x x + y y + z z
    T sx = x, sy = y, sz = z;
    if (n2 > real(0)) {
      const T n = m_sqrt(n2);
      sx = x / n;
      sy = y / n;
      sz = z / n;
    }
    const T den = sz + xi;
    const T ux = sx / den, uy = sy / den;
    const T x2 = ux * ux, y2 = uy * uy, xy = ux * uy;
    const T r2 = mad<FAST>(ux, ux, y2);
    const T r4 = r2 * r2;
    const T dr = mad<FAST>(k2, r4, mad<FAST>(k1, r2, real(1)));
    const T nx = mad<FAST>(p2, mad<FAST>(real(2), x2, r2), mad<FAST>(real(2) * p1, xy, ux * dr));
    const T ny = mad<FAST>(real(2) * p2, xy, mad<FAST>(p1, mad<FAST>(real(2), y2, r2), uy * dr));
    u = mad<FAST>(c.intr[0], nx, c.intr[2]);
    v = mad<FAST>(c.intr[1], ny, c.intr[3]);
  } else if (MODEL == MODEL_EQUIRECT) {  // equirectangular.hpp:14-28, intr = [W H]
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
  } else if (MODEL == MODEL_ATAN) {  // atan.hpp:14-39
    const real d0 = c.dist[0];
    T px, py;
    persp<FAST>(x, y, z, px, py);
    const T r = m_sqrt(mad<FAST>(px, px, py * py));
    T dx = px, dy = py;
    if (!(r < real(1e-3) || d0 < real(1e-7))) {
      const real d1 = real(1) / d0;
      const real d2 = real(2) * tan(d0 / real(2));
      const T factor = d1 * m_atan(r * d2) / r;
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
    const T t1 = real(2) * px * py;
    const T t2 = mad<FAST>(real(2), x2, r2);
    const T t3 = mad<FAST>(real(2), y2, r2);
    const T dx = mad<FAST>(p2, t2, mad<FAST>(p1, t1, rc * px));
    const T dy = mad<FAST>(p2, t1, mad<FAST>(p1, t3, rc * py));
    u = mad<FAST>(c.intr[0], dx, c.intr[2]);
    v = mad<FAST>(c.intr[1], dy, c.intr[3]);
  }
}

// projection value + 2x3 Jacobian d(u,v)/d(x,y,z) for the gradient pass.  Generic route: Dual3
// forward mode (3 partials through every operation).  plumb_bob / rational_polynomial: the Jacobian of
// the distortion polynomial is written out by hand (it is symmetric) and chained with the closed-form
// Jacobian of the perspective division, ~2.5x fewer fp64 operations than Dual3.
template <int MODEL, typename real>
__device__ __forceinline__ void project_jac(const CamParams<real>& c, real x, real y, real z, real& u, real& v, real* du, real* dv) {
  if (MODEL == MODEL_PLUMB_BOB || MODEL == MODEL_RATIONAL) {
    project<MODEL, real, real, true>(c, x, y, z, u, v);  // value: identical expression to the histogram pass
    const real iz = fast_rcp(z);
    const real px = x * iz, py = y * iz;
    const real x2 = px * px, y2 = py * py, xy = px * py;
    const real r2 = fma(px, px, y2);
    const real r4 = r2 * r2;
    real rc, rcp, p1, p2;  // radial factor and d(rc)/d(r2)
    if (MODEL == MODEL_PLUMB_BOB) {
      const real k1 = c.dist[0], k2 = c.dist[1], k3 = c.dist[4];
      p1 = c.dist[2];
      p2 = c.dist[3];
      const real r6 = r2 * r4;
      rc = fma(k3, r6, fma(k2, r4, fma(k1, r2, real(1))));
      rcp = fma(real(3) * k3, r4, fma(real(2) * k2, r2, k1));
    } else {
      const real k1 = c.dist[0], k2 = c.dist[1], k3 = c.dist[4], k4 = c.dist[5], k5 = c.dist[6], k6 = c.dist[7];
      p1 = c.dist[2];
      p2 = c.dist[3];
      const real r6 = r2 * r4;
      const real num = fma(k3, r6, fma(k2, r4, fma(k1, r2, real(1))));
      const real den = fma(k6, r6, fma(k5, r4, fma(k4, r2, real(1))));
      const real nump = fma(real(3) * k3, r4, fma(real(2) * k2, r2, k1));
      const real denp = fma(real(3) * k6, r4, fma(real(2) * k5, r2, k4));
      if (den > real(1e-8)) {
        const real id = real(1) / den;
        rc = num * id;
        rcp = (nump - rc * denp) * id;
      } else {
        rc = num;
        rcp = nump;
      }
    }
    const real off = fma(real(2) * xy, rcp, real(2) * fma(p1, px, p2 * py));  // d(dx)/d(py) = d(dy)/d(px)
    const real a00 = c.intr[0] * fma(real(2) * x2, rcp, rc + real(2) * fma(p1, py, real(3) * p2 * px));
    const real a01 = c.intr[0] * off;
    const real a10 = c.intr[1] * off;
    const real a11 = c.intr[1] * fma(real(2) * y2, rcp, rc + real(2) * fma(real(3) * p1, py, p2 * px));
    du[0] = a00 * iz;
    du[1] = a01 * iz;
    du[2] = -fma(du[0], px, du[1] * py);
    dv[0] = a10 * iz;
    dv[1] = a11 * iz;
    dv[2] = -fma(dv[0], px, dv[1] * py);
  } else {
    typedef Dual3<real> D;
    D uu, vv;
    project<MODEL, D, real, true>(c, D(x, real(1), real(0), real(0)), D(y, real(0), real(1), real(0)), D(z, real(0), real(0), real(1)), uu, vv);
    u = uu.a;
    v = vv.a;
    du[0] = uu.d0;
    du[1] = uu.d1;
    du[2] = uu.d2;
    dv[0] = vv.d0;
    dv[1] = vv.d1;
    dv[2] = vv.d2;
  }
}

// ------------------------------------------------------------------------------------------
// device point records (written once per handle by the host-side bucketing)
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

struct Chunk {  // one workgroup's slice of the bucketed cloud
  uint32_t start;
  uint32_t count;
  uint32_t group;  // histogram columns [group*GW, (group+1)*GW)
  uint32_t pad;
};

// uniform cubic B-spline basis, the reference's 4x4 coefficient matrix / 6 (nid_cost.hpp:29-33),
// evaluated as C * [1 s s^2 s^3]^T in the same term order
template <typename real>
__device__ __forceinline__ void bspline(real s, real* b) {
  // explicit fma: all translation units are built with -ffp-contract=off so that every point gets
  // the same arithmetic no matter which unrolled slot / chunk / GPU processes it (the histogram is
  // bit-identical across tilings); the fusions we want are therefore written out
  const real s2 = s * s, s3 = s2 * s;
  const real k16 = real(1.0 / 6.0), k36 = real(3.0 / 6.0), k46 = real(4.0 / 6.0);
  const real t = real(1) - s;  // b0 = (1-s)^3 / 6, written so that it can never round below +0: the
  b[0] = k16 * (t * t * t);     // subnormal fixed-point trick (to_fixed_dn) needs sign bit 0 on every weight
  b[1] = fma(k36, s3, k46 - s2);
  b[2] = fma(-k36, s3, fma(k36, s2, fma(k36, s, k16)));
  b[3] = k16 * s3;
}
template <typename real>
__device__ __forceinline__ void bspline_deriv(real s, real* d) {
  const real s2 = s * s;
  d[0] = fma(real(-0.5), s2, s - real(0.5));
  d[1] = fma(real(1.5), s2, real(-2) * s);
  d[2] = fma(real(-1.5), s2, s + real(0.5));
  d[3] = real(0.5) * s2;
}

// p_cam = R p + t with fused multiply-adds (SPLINE kernels)
template <typename real>
__device__ __forceinline__ void transform_fma(const PoseParams<real>& pose, real x, real y, real z, real& cx, real& cy, real& cz) {
  cx = fma(pose.R[2], z, fma(pose.R[1], y, fma(pose.R[0], x, pose.t[0])));
  cy = fma(pose.R[5], z, fma(pose.R[4], y, fma(pose.R[3], x, pose.t[1])));
  cz = fma(pose.R[8], z, fma(pose.R[7], y, fma(pose.R[6], x, pose.t[2])));
}

// weight -> unsigned fixed point with `frac` fractional bits in ONE instruction: the x-weights are
// pre-multiplied by dn = 2^(frac - 1074), so the product bxs * by is a SUBNORMAL double whose bit
// pattern (exponent field 0) IS the integer round-to-nearest(bx' * by * 2^frac) -- no magic add, no mask.
// (bx' = bx rounded to 2^-frac by the pre-scaling: total quantisation <= 0.75 units of 2^-frac instead
// of 0.5; still deterministic and order independent.)  gfx950 handles fp64 denormals at full rate.
__device__ __forceinline__ u64 to_fixed_dn(double bx_scaled, double by) { return u64(__double_as_longlong(bx_scaled * by)); }


// The bin image is stored in STRIPS of four rows with the four vertically adjacent pixels of a
// column contiguous: byte address of padded pixel (x, y) = (y >> 2) * 4 * pitch + 4 * x + (y & 3).
// The 4x4 tap patch [kx..kx+3] x [ky..ky+3] is then two 16-byte loads (one per strip it touches)
// instead of four row gathers -- half the L1 lookups, which co-limited both spline kernels -- and a
// v_alignbyte per column puts rows ky..ky+3 of column kx+a into the four bytes of cols[a].
struct __attribute__((aligned(4))) StripQuad {
  uint32_t c[4];
};
__device__ __forceinline__ void load_patch(const uint8_t* __restrict__ img, int pitch, int kx, int ky, uint32_t* cols) {
  const uint32_t stride = uint32_t(pitch) * 4u;
  const uint32_t base = (uint32_t(ky) >> 2) * stride + uint32_t(kx) * 4u;
  const StripQuad s0 = *reinterpret_cast<const StripQuad*>(img + base);
  const StripQuad s1 = *reinterpret_cast<const StripQuad*>(img + base + stride);
  const uint32_t sh = uint32_t(ky) & 3u;
#pragma unroll
  for (int a = 0; a < 4; a++) cols[a] = __builtin_amdgcn_alignbyte(s1.c[a], s0.c[a], sh);
}
__device__ __forceinline__ uint32_t load_pixel(const uint8_t* __restrict__ img, int pitch, int x, int y) {
  return img[(uint32_t(y) >> 2) * uint32_t(pitch) * 4u + uint32_t(x) * 4u + (uint32_t(y) & 3u)];
}

}  // namespace nidreg
