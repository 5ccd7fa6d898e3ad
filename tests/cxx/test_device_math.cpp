// Host-side check of the scalar device math in csrc/nid_device.hpp (compiled by hipcc as a HIP
// source, run on the CPU -- no kernel is launched): the SPLINE kernels' FAST projections
// (reciprocal / rsqrt with Newton steps, division-free atan2) and the hand-derived 2x3 projection
// Jacobians are compared with the oracle's camera functors instantiated with double and Jet<7>
// (the reference's own route to those derivatives, generic_camera_base.hpp:34-40).
// Prints one line per model: max |uv - uv_ref| (pixels), max relative Jacobian error; exit 1 on failure.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <random>
#include <vector>

#include "../../direct_visual_lidar_calibration_amd/csrc/nid_device.hpp"
#include "../../oracle/cameras.hpp"

using namespace nidreg;

template <int MODEL, typename Proj>
static int check(const char* name, const double* intr, int ni, const double* dist, int nd, double tol_uv, double tol_jac) {
  CamParams<double> c;
  double I[5] = {0, 0, 0, 0, 0}, D[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 5; i++) c.intr[i] = I[i] = i < ni ? intr[i] : 0.0;
  for (int i = 0; i < 8; i++) c.dist[i] = D[i] = i < nd ? dist[i] : 0.0;
  cam_derive<double>(MODEL, c);  // (the constants a model derives from its coefficients on the host, as the library's make_cam does)
  std::mt19937_64 rng(1234 + MODEL);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  Proj proj;
  double e_uv = 0, e_fast_exact = 0, e_j = 0;
  int n_checked = 0;
  for (int it = 0; it < 200000; it++) {
    double x, y, z;
    if (MODEL == MODEL_EQUIRECT || MODEL == MODEL_OMNIDIR) {  // full sphere (omnidir: in front of the mirror singularity)
      x = 20 * U(rng), y = 20 * U(rng), z = 20 * U(rng);
      if (MODEL == MODEL_OMNIDIR && z < 0.2 * std::sqrt(x * x + y * y)) continue;
    } else {
      z = 0.5 + 15 * (U(rng) + 1.0);
      x = z * 1.2 * U(rng), y = z * 0.8 * U(rng);
    }
    if (it % 1000 == 7 && MODEL == MODEL_FISHEYE) z = -z;  // fisheye.hpp:16 takes abs(z)
    // reference value and Jacobian (Jets seeded on x, y, z)
    oracle::V3<oracle::Jet7> pj{oracle::Jet7(x, 0), oracle::Jet7(y, 1), oracle::Jet7(z, 2)};
    const oracle::V2<oracle::Jet7> rj = proj(I, D, pj);
    // exact-order device expression (NEAREST path) and FAST expression (SPLINE path)
    double ue, ve, uf, vf, u, v, du[3], dv[3];
    project<MODEL, double, double, false>(c, x, y, z, ue, ve);
    project<MODEL, double, double, true>(c, x, y, z, uf, vf);
    project_jac<MODEL, double>(c, x, y, z, u, v, du, dv);
    if (!(std::isfinite(rj.x.a) && std::isfinite(rj.y.a))) continue;
    if (u != uf || v != vf) {
      std::printf("%s: project_jac value differs from the histogram pass's value\n", name);
      return 1;
    }
    if (ue != rj.x.a || ve != rj.y.a) {  // the NEAREST path must be bit-identical to the CPU expression
      // pow(theta, k) in the reference vs products here can differ in the last bit for fisheye
      if (std::fabs(ue - rj.x.a) > 1e-9 || std::fabs(ve - rj.y.a) > 1e-9) {
        std::printf("%s: exact-order projection differs: %.17g %.17g vs %.17g %.17g\n", name, ue, ve, rj.x.a, rj.y.a);
        return 1;
      }
    }
    e_fast_exact = std::fmax(e_fast_exact, std::fmax(std::fabs(ue - rj.x.a), std::fabs(ve - rj.y.a)));
    e_uv = std::fmax(e_uv, std::fmax(std::fabs(uf - rj.x.a), std::fabs(vf - rj.y.a)));
    double scale = 1e-300;
    for (int k = 0; k < 3; k++) scale = std::fmax(scale, std::fmax(std::fabs(rj.x.v[k]), std::fabs(rj.y.v[k])));
    for (int k = 0; k < 3; k++) {
      e_j = std::fmax(e_j, std::fabs(du[k] - rj.x.v[k]) / scale);
      e_j = std::fmax(e_j, std::fabs(dv[k] - rj.y.v[k]) / scale);
    }
    n_checked++;
  }
  std::printf("%s checked=%d exact_vs_ref=%.3g fast_vs_ref=%.3g jac_rel=%.3g\n", name, n_checked, e_fast_exact, e_uv, e_j);
  if (!(e_uv <= tol_uv) || !(e_j <= tol_jac) || n_checked < 50000) return 1;
  return 0;
}

int main() {
  int bad = 0;
  {
    const double intr[4] = {1100, 1100, 960, 540}, dist[5] = {-0.04, 0.08, 1e-4, -3e-4, -0.04};
    bad += check<MODEL_PLUMB_BOB, oracle::PinholeProjection>("plumb_bob", intr, 4, dist, 5, 1e-9, 1e-11);
  }
  {
    const double intr[4] = {600, 600, 960, 540}, dist[4] = {-0.01, 0.002, -1e-4, 1e-5};
    bad += check<MODEL_FISHEYE, oracle::FisheyeProjection>("fisheye", intr, 4, dist, 4, 1e-9, 1e-11);
  }
  {
    const double intr[5] = {600, 600, 1024, 1024, 1.0}, dist[4] = {-0.02, 0.003, 2e-4, -1e-4};
    bad += check<MODEL_OMNIDIR, oracle::OmnidirProjection>("omnidir", intr, 5, dist, 4, 1e-9, 1e-11);
  }
  {
    const double intr[2] = {2048, 2048};
    bad += check<MODEL_EQUIRECT, oracle::EquirectProjection>("equirectangular", intr, 2, nullptr, 0, 1e-9, 1e-10);
  }
  {
    const double intr[4] = {500, 500, 320, 240}, dist[1] = {0.9};
    bad += check<MODEL_ATAN, oracle::AtanProjection>("atan", intr, 4, dist, 1, 1e-9, 1e-11);
  }
  {
    const double intr[4] = {1100, 1100, 960, 540}, dist[8] = {0.1, -0.05, 1e-4, -2e-4, 0.01, 0.05, -0.02, 0.003};
    bad += check<MODEL_RATIONAL, oracle::RationalProjection>("rational_polynomial", intr, 4, dist, 8, 1e-9, 1e-11);
  }
  // the poles / axis of the equirectangular model: atan2(0, 0) = 0 like libm
  {
    CamParams<double> c{};
    c.intr[0] = 2048, c.intr[1] = 1024;
    double u, v;
    project<MODEL_EQUIRECT, double, double, true>(c, 0.0, 3.0, 0.0, u, v);
    oracle::EquirectProjection P;
    const double I[2] = {2048, 1024};
    const oracle::V2<double> r = P(I, nullptr, oracle::V3<double>{0.0, 3.0, 0.0});
    std::printf("equirect axis: %.17g %.17g vs %.17g %.17g\n", u, v, r.x, r.y);
    if (std::fabs(u - r.x) > 1e-9 || std::fabs(v - r.y) > 1e-9) bad++;
    project<MODEL_EQUIRECT, double, double, true>(c, 0.01, 0.0, 0.01, u, v);  // |p|^2 < 1e-3 -> image centre
    if (u != 1024.0 || v != 512.0) bad++;
    // fisheye on the optical axis: NaN like the reference's 0/0
    CamParams<double> f{};
    f.intr[0] = f.intr[1] = 600, f.intr[2] = 960, f.intr[3] = 540;
    project<MODEL_FISHEYE, double, double, true>(f, 0.0, 0.0, 5.0, u, v);
    if (u == u || v == v) {
      std::printf("fisheye axis: expected NaN, got %g %g\n", u, v);
      bad++;
    }
  }
  // B-spline basis (nid_cost.hpp:29-33: C/6 applied to [1 s s^2 s^3]) and its derivative; pose transform
  {
    const double Cm[4][4] = {{1.0, -3.0, 3.0, -1.0}, {4.0, 0.0, -6.0, 3.0}, {1.0, 3.0, 3.0, -3.0}, {0.0, 0.0, 0.0, 1.0}};
    double eb = 0, ed = 0, es = 0, emin = 1.0;
    for (int i = 0; i <= 100000; i++) {
      const double t = i == 100000 ? std::nextafter(1.0, 0.0) : i / 100000.0;
      double b[4], d[4];
      bspline6<double>(t, b);       // 6 b
      bspline_deriv2<double>(t, d);  // 2 db/ds
      double sum = 0;
      for (int k = 0; k < 4; k++) {
        const double ref = Cm[k][0] + Cm[k][1] * t + Cm[k][2] * t * t + Cm[k][3] * t * t * t;
        const double dref = (Cm[k][1] + 2.0 * Cm[k][2] * t + 3.0 * Cm[k][3] * t * t) / 3.0;
        eb = std::fmax(eb, std::fabs(b[k] - ref));
        ed = std::fmax(ed, std::fabs(d[k] - dref));
        emin = std::fmin(emin, b[k]);
        sum += b[k];
      }
      es = std::fmax(es, std::fabs(sum - 6.0));
    }
    std::printf("bspline6 max err %.3g, derivative (x2) %.3g, partition of unity (x6) %.3g, min weight %.3g\n", eb, ed, es, emin);
    // every weight must be >= +0: the fixed-point conversion reads the product's bit pattern (to_fixed_dn)
    if (eb > 4e-15 || ed > 4e-15 || es > 4e-15 || emin < 0.0 || std::signbit(emin)) bad++;
    float bf[4];
    bspline6<float>(0.37f, bf);
    double b64[4];
    bspline6<double>(double(0.37f), b64);
    for (int k = 0; k < 4; k++)
      if (std::fabs(bf[k] - b64[k]) > 1e-6) bad++;
    // p_cam = R p + t (fma chain) against the plain expression
    PoseParams<double> pose;
    std::mt19937_64 rng(5);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int k = 0; k < 9; k++) pose.R[k] = U(rng);
    for (int k = 0; k < 3; k++) pose.t[k] = U(rng);
    double et = 0;
    for (int i = 0; i < 100000; i++) {
      const double x = 30 * U(rng), y = 30 * U(rng), z = 30 * U(rng);
      double cx, cy, cz;
      transform_fma<double>(pose, x, y, z, cx, cy, cz);
      et = std::fmax(et, std::fabs(cx - (pose.R[0] * x + pose.R[1] * y + pose.R[2] * z + pose.t[0])));
      et = std::fmax(et, std::fabs(cy - (pose.R[3] * x + pose.R[4] * y + pose.R[5] * z + pose.t[1])));
      et = std::fmax(et, std::fabs(cz - (pose.R[6] * x + pose.R[7] * y + pose.R[8] * z + pose.t[2])));
    }
    std::printf("transform_fma max err %.3g\n", et);
    if (et > 2e-14) bad++;
  }
  // fixed point in one multiply (to_fixed_dn): the x-weights come out of bspline_scaled in grid steps of 2^-1074 (6 bx U/36),
  // so that bits(bx' * 6 by) is the integer round(bx by U); exact zeros, sign bit never set
  {
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    double worst = 0, worst_sum = 0;
    for (int frac = 34; frac <= 40; frac += 2) {
      // the fixed-point unit is U = 36 round(2^frac / 36) grid steps (nidreg_internal.hpp fixed_unit), so that k = U/36, 3k, 4k, 6k are integers
      const double unit = 36.0 * std::rint(std::ldexp(1.0, frac) / 36.0), dn = std::ldexp(unit, -1074);
      const BsplineScale KS = bspline_scale(std::ldexp(unit / 36.0, -1074));
      if (KS.k46 != std::ldexp(unit / 9.0, -1074) || KS.k05 != std::ldexp(unit / 12.0, -1074) || KS.k1 != std::ldexp(unit / 6.0, -1074)) bad++;  // exact constants
      const BsplineScale KZ = {0.0, 0.0, 0.0, 0.0};  // an outlier / padding slot: zeroed constants
      {
        double z4[4];
        bspline_scaled(0.37, KZ, z4);
        for (int a = 0; a < 4; a++)
          if (to_fixed_dn(z4[a], 3.6) != 0) bad++;  // adds exact zeros, sign bit clear
      }
      for (int i = 0; i < 200000; i++) {
        double bx[4], by[4], bxs4[4];
        const double sxv = i == 0 ? 0.0 : (i == 1 ? std::nextafter(1.0, 0.0) : U(rng));
        bspline6<double>(sxv, bx);
        bspline_scaled(sxv, KS, bxs4);  // the polynomial evaluated with constants in grid steps (subnormal arithmetic)
        bspline6<double>(i == 2 ? 0.0 : U(rng), by);
        double sum = 0;
        for (int a = 0; a < 4; a++) {
          const double bxs = bxs4[a];
          if (std::signbit(bxs)) bad++;
          for (int b = 0; b < 4; b++) {
            const u64 bits = to_fixed_dn(bxs, by[b]);
            if (bits >> 63) bad++;
            worst = std::fmax(worst, std::fabs(double(bits) - bx[a] * by[b] * (unit / 36.0)));
            sum += double(bits);
          }
        }
        worst_sum = std::fmax(worst_sum, std::fabs(sum - unit));  // partition of unity: 16 taps add up to 1.0
      }
      if (to_fixed_dn(0.0 * dn, 0.7) != 0 || to_fixed_dn(0.3 * dn, 0.0) != 0 || to_fixed_dn(0.5 * 0.0, 0.25) != 0) bad++;  // outliers add exact zeros
    }
    std::printf("to_fixed_dn: max |fixed - exact| %.3f units, max |sum of 16 taps - 1| %.1f units\n", worst, worst_sum);
    // each x-weight carries <= ~1.6 grid steps of rounding (two roundings of the polynomial; the constants are exact),
    // times 6 by <= 4, plus half a step for the product: <= 7 of ~2.7e11 steps per unit weight, unbiased
    if (worst > 7.0 || worst_sum > 40.0) bad++;
  }
  // strip-tiled padded bin image: padded pixel (x, y) at (y >> 2) * 4 * pitch + 4 x + (y & 3); load_patch returns, for
  // knot (kx, ky), byte b of cols[a] = padded pixel (kx + a, ky + b) = source pixel (clamp(kx + a - 1), clamp(ky + b - 1))
  {
    const int W = 53, H = 38;
    const int pitch = ((W + 8) + 3) & ~3, nstrips = (H + 3 + 3) / 4 + 1;
    std::mt19937 rng(12);
    std::vector<uint8_t> src(size_t(W) * H), img(size_t(pitch) * 4 * nstrips + 64, 0);
    for (auto& v : src) v = uint8_t(rng() & 255);
    for (int py = 0; py < nstrips * 4; py++)
      for (int px = 0; px < pitch; px++) {
        const int sy = std::min(std::max(py - 1, 0), H - 1), sx = std::min(std::max(px - 1, 0), W - 1);
        img[size_t(py >> 2) * pitch * 4 + size_t(px) * 4 + (py & 3)] = src[size_t(sy) * W + sx];
      }
    int wrong = 0;
    for (int ky = 0; ky < H; ky++)
      for (int kx = 0; kx < W; kx++) {
        uint32_t cols[4];
        load_patch(img.data(), pitch, kx, ky, cols);
        for (int a = 0; a < 4; a++)
          for (int b = 0; b < 4; b++) {
            const int sx = std::min(std::max(kx + a - 1, 0), W - 1), sy = std::min(std::max(ky + b - 1, 0), H - 1);
            if (((cols[a] >> (8 * b)) & 0xffu) != src[size_t(sy) * W + sx]) wrong++;
          }
        if (load_pixel(img.data(), pitch, kx + 1, ky + 1) != src[size_t(ky) * W + kx]) wrong++;
      }
    std::printf("load_patch / load_pixel on the strip-tiled image: %d mismatches\n", wrong);
    if (wrong) bad++;
  }
  // fast_atan2 / fast_rcp / fast_rsq accuracy
  {
    std::mt19937_64 rng(99);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    double ea = 0, er = 0, es = 0;
    for (int i = 0; i < 2000000; i++) {
      const double a = 30 * U(rng), b = 30 * U(rng);
      ea = std::fmax(ea, std::fabs(fast_atan2(a, b) - std::atan2(a, b)));
      const double z = 0.01 + 100 * std::fabs(U(rng));
      er = std::fmax(er, std::fabs(fast_rcp(z) * z - 1.0));
      es = std::fmax(es, std::fabs(fast_rsq(z) * std::sqrt(z) - 1.0));
    }
    std::printf("fast_atan2 max abs err %.3g, fast_rcp rel %.3g, fast_rsq rel %.3g\n", ea, er, es);
    // one Newton step on a 2^-23 seed leaves <= 2^-46 (rcp) / 1.5 * 2^-46 (rsq); the atan2 argument inherits the reciprocal's
    const double tol = kNewtonSteps >= 2 ? 5e-16 : 3e-14;
    if (ea > 1e-15 + tol || er > tol || es > tol) bad++;
    if (fast_atan2(0.0, 0.0) != 0.0 || fast_atan2(0.0, -1.0) != std::atan2(0.0, -1.0) || fast_atan2(-0.0, 1.0) != 0.0) bad++;
  }
  // fast_log (the entropy terms' and the G tile's logarithm): against the long-double library logarithm over [1e-9, 4], dense
  // around 1 (where k = 0 must leave no cancellation) and on the arguments the kernels really pass (p + 1e-6, p in [0, 1])
  {
    std::mt19937_64 rng(123);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    double erel = 0, eabs1 = 0;
    for (int i = 0; i < 3000000; i++) {
      double x;
      switch (i % 4) {
        case 0: x = std::exp(std::log(1e-9) + U(rng) * (std::log(4.0) - std::log(1e-9))); break;
        case 1: x = U(rng) + 1e-6; break;
        case 2: x = 1.0 + (U(rng) - 0.5) * 1e-3; break;
        default: x = 0.6875 + U(rng) * 0.6875; break;
      }
      const long double ref = logl((long double)x);
      const double got = fast_log(x);
      const long double d = fabsl((long double)got - ref);
      erel = std::fmax(erel, double(d / std::fmax(1e-2L, fabsl(ref))));
      if (std::fabs(x - 1.0) < 1e-3) eabs1 = std::fmax(eabs1, double(d));
    }
    std::printf("fast_log: max error relative to max(|log x|, 1e-2) %.3g; absolute within 1e-3 of x = 1: %.3g\n", erel, eabs1);
    if (erel > 4e-16 || eabs1 > 1e-17) bad++;
    if (std::fabs(fast_log(1.0)) > 5e-18) bad++;  // (x = 1 sits at the edge of a table interval: the two halves cancel to ~2e-18, not to 0)
  }
  return bad ? 1 : 0;
}
