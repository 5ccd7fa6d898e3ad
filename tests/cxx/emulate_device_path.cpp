// Host emulation of the DEVICE ALGORITHM of one NIDCost evaluation (k_spline_hist -> k_entropy -> k_spline_grad in
// csrc/nid_kernels.hpp), built from the very same scalar helpers the kernels call -- transform_fma, project<.., FAST>,
// project_jac, bspline6, bspline_deriv2, load_patch on the strip-tiled padded bin image, to_fixed_dn -- which
// csrc/nid_device.hpp compiles for the host too.  The per-point bodies below MIRROR the kernels' (they are not shared
// code: the kernels interleave them with LDS / wave plumbing); what this pins on a machine without a GPU is the
// arithmetic design: 64-bit fixed-point accumulation (order independent, so the serial loop here and the GPU's
// workgroups produce the same integers up to the seed of fast_rcp / fast_rsq), the entropy tail with its
// dNID/dh coefficients, and the reverse-mode gradient with its 3x3 + 3 accumulator and quaternion chain rule.
// Reads the scene file of tests/test_cxx_dropin.py; prints "cost g0..g6 inliers".  tests/test_device_emulation.py
// compares with the oracle at the GPU parity bars (1e-10 / rtol 1e-7).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../direct_visual_lidar_calibration_amd/csrc/nid_device.hpp"

using namespace nidreg;

template <int MODEL>
static int run(const CamParams<double>& cam, int W, int H, int B, const std::vector<uint8_t>& src_bins, const std::vector<double>& pts, const std::vector<double>& ints, const double* se3) {
  const int64_t N = int64_t(ints.size());
  // ---- padded, strip-tiled bin image (nidreg_plan.hip create_impl: 1 px left / top, >= 2 right / bottom, edge replicated)
  const int pitch = ((W + 8) + 3) & ~3, nstrips = (H + 3 + 3) / 4 + 1;
  std::vector<uint8_t> img(size_t(pitch) * 4 * nstrips + 64, 0);
  for (int py = 0; py < nstrips * 4; py++)
    for (int px = 0; px < pitch; px++) {
      const int sy = std::min(std::max(py - 1, 0), H - 1), sx = std::min(std::max(px - 1, 0), W - 1);
      img[size_t(py >> 2) * pitch * 4 + size_t(px) * 4 + (py & 3)] = src_bins[size_t(sy) * W + sx];
    }
  // ---- pose: R from the UN-normalised quaternion exactly as p + 2 w (v x p) + 2 v x (v x p) expands (nidreg_plan.hip pose_from_se3)
  const double qx = se3[0], qy = se3[1], qz = se3[2], qw = se3[3];
  PoseParams<double> pose;
  pose.R[0] = 1 - 2 * (qy * qy + qz * qz), pose.R[1] = 2 * (qx * qy - qz * qw), pose.R[2] = 2 * (qx * qz + qy * qw);
  pose.R[3] = 2 * (qx * qy + qz * qw), pose.R[4] = 1 - 2 * (qx * qx + qz * qz), pose.R[5] = 2 * (qy * qz - qx * qw);
  pose.R[6] = 2 * (qx * qz - qy * qw), pose.R[7] = 2 * (qy * qz + qx * qw), pose.R[8] = 1 - 2 * (qx * qx + qy * qy);
  pose.t[0] = se3[4], pose.t[1] = se3[5], pose.t[2] = se3[6];
  int nbits = 1;
  while ((int64_t(1) << nbits) <= N) nbits++;
  const int frac = std::min(40, 62 - nbits);
  const double unit = 36.0 * std::rint(std::ldexp(1.0, frac) / 36.0), inv_unit = 1.0 / unit;  // nidreg_plan.hip fixed_unit
  const BsplineScale KS = bspline_scale(std::ldexp(unit / 36.0, -1074));

  // ---- pass A (k_spline_hist body): fixed-point joint histogram [bin_points][bin_image], inlier count
  std::vector<u64> hist(size_t(B) * B, 0);
  std::vector<int> bin_pts(static_cast<size_t>(N));
  u64 inliers = 0;
  const double fW = W, fH = H;
  for (int64_t i = 0; i < N; i++) {
    const double x = double(float(pts[4 * i])), y = double(float(pts[4 * i + 1])), z = double(float(pts[4 * i + 2]));  // Rec32 records
    bin_pts[size_t(i)] = std::max(0, std::min(B - 1, int(ints[size_t(i)] * B)));
    double cx, cy, cz, u, v;
    transform_fma<double>(pose, x, y, z, cx, cy, cz);
    project<MODEL, double, double, true>(cam, cx, cy, cz, u, v);
    const bool in = (u >= 0.0) && (u < fW) && (v >= 0.0) && (v < fH);
    if (!in) continue;
    inliers++;
    double bxs[4], by[4];
    bspline_scaled(std::fabs(m_fract(u)), KS, bxs);  // x-weights straight in fixed-point units (k_spline_hist's taps)
    bspline6<double>(std::fabs(m_fract(v)), by);
    uint32_t cols[4];
    load_patch(img.data(), pitch, int(u), int(v), cols);
    for (int b = 0; b < 4; b++)
      for (int a = 0; a < 4; a++) hist[size_t(bin_pts[size_t(i)]) * B + ((cols[a] >> (8 * b)) & 0xffu)] += to_fixed_dn(bxs[a], by[b]);
  }

  // ---- entropy tail (k_entropy + entropy_final_body): hist_image = row sums, hist_points = column sums / unit
  const double S = double(inliers);
  std::vector<double> phi_q(static_cast<size_t>(B));
  double hi = 0, hp = 0, hj = 0;
  for (int r = 0; r < B; r++) {
    u64 t = 0;
    for (int c = 0; c < B; c++) t += hist[size_t(c) * B + r];
    const double q = double(t) * inv_unit / S;
    hi += q * fast_log(q + 1e-6);
    phi_q[size_t(r)] = fast_log(q + 1e-6) + q / (q + 1e-6);
  }
  for (int c = 0; c < B; c++) {
    u64 t = 0;
    for (int r = 0; r < B; r++) t += hist[size_t(c) * B + r];
    const double p = std::rint(double(t) * inv_unit) / S;  // exact inlier count of the column (partition of unity)
    hp += p * fast_log(p + 1e-6);
  }
  const double scale = inv_unit / S;
  for (size_t k = 0; k < hist.size(); k++)
    if (hist[k]) {
      const double p = double(hist[k]) * scale;
      hj += p * fast_log(p + 1e-6);
    }
  const double Hi = -hi, Hp = -hp, Hj = -hj;
  const double MI = Hi + Hp - Hj, nid = (Hj - MI) / Hj;
  const double coefA = -(Hi + Hp) / (Hj * Hj * S), coefB = 1.0 / (Hj * S);

  // ---- pass B (k_spline_grad body): G = dNID/dh, per point (gx, gy) -> gp -> M += gp p^T, gt += gp
  std::vector<double> G(size_t(B) * B);
  for (int c = 0; c < B; c++)
    for (int r = 0; r < B; r++) {
      const double p = double(hist[size_t(c) * B + r]) * scale;
      G[size_t(c) * B + r] = (coefA * (fast_log(p + 1e-6) + p / (p + 1e-6)) + coefB * phi_q[size_t(r)]) * (1.0 / 12.0);  // the taps use 6 b and 2 db/ds
    }
  double acc[12] = {0};
  for (int64_t i = 0; i < N; i++) {
    const double x = double(float(pts[4 * i])), y = double(float(pts[4 * i + 1])), z = double(float(pts[4 * i + 2]));
    double cx, cy, cz, uu, vv;
    ProjCtx<double> ctx;
    transform_fma<double>(pose, x, y, z, cx, cy, cz);
    project_fwd<MODEL, double>(cam, cx, cy, cz, uu, vv, ctx);  // k_spline_grad's two halves around the tap loop
    const bool in = (uu >= 0.0) && (uu < fW) && (vv >= 0.0) && (vv < fH);
    if (!in) continue;
    const double sx = std::fabs(m_fract(uu)), sy = std::fabs(m_fract(vv));
    double bx[4], by[4], dbx[4], dby[4];
    bspline6<double>(sx, bx);
    bspline6<double>(sy, by);
    bspline_deriv2<double>(sx, dbx);
    bspline_deriv2<double>(sy, dby);
    uint32_t cols[4];
    load_patch(img.data(), pitch, int(uu), int(vv), cols);
    const double* gcol = G.data() + size_t(bin_pts[size_t(i)]) * B;
    double gx = 0, gy = 0;
    for (int b = 0; b < 4; b++) {
      double sa = 0, sb = 0;
      for (int a = 0; a < 4; a++) {
        const double g = gcol[(cols[a] >> (8 * b)) & 0xffu];
        sa = fma(g, dbx[a], sa);
        sb = fma(g, bx[a], sb);
      }
      gx = fma(sa, by[b], gx);
      gy = fma(sb, dby[b], gy);
    }
    double gp[3];
    project_bwd<MODEL, double>(cam, ctx, gx, gy, gp);
    const double p[3] = {x, y, z};
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) acc[3 * r + c] = fma(gp[r], p[c], acc[3 * r + c]);
      acc[9 + r] += gp[r];
    }
  }
  // grad_final_body: chain M, gt to d/d[qx qy qz qw tx ty tz]
  const double* M = acc;
  const double A0 = M[7] - M[5], A1 = M[2] - M[6], A2 = M[3] - M[1];
  const double tr = M[0] + M[4] + M[8];
  const double Mv0 = M[0] * qx + M[1] * qy + M[2] * qz, Mv1 = M[3] * qx + M[4] * qy + M[5] * qz, Mv2 = M[6] * qx + M[7] * qy + M[8] * qz;
  const double Mt0 = M[0] * qx + M[3] * qy + M[6] * qz, Mt1 = M[1] * qx + M[4] * qy + M[7] * qz, Mt2 = M[2] * qx + M[5] * qy + M[8] * qz;
  double g[7];
  g[0] = 2.0 * qw * A0 + 2.0 * (Mv0 + Mt0 - 2.0 * tr * qx);
  g[1] = 2.0 * qw * A1 + 2.0 * (Mv1 + Mt1 - 2.0 * tr * qy);
  g[2] = 2.0 * qw * A2 + 2.0 * (Mv2 + Mt2 - 2.0 * tr * qz);
  g[3] = 2.0 * (qx * A0 + qy * A1 + qz * A2);
  g[4] = M[9], g[5] = M[10], g[6] = M[11];
  std::printf("%.17g", nid);
  for (int k = 0; k < 7; k++) std::printf(" %.17g", g[k]);
  std::printf(" %llu\n", (unsigned long long)inliers);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int W, H, N, bins, nintr, ndist;
  char model[64] = {0};
  double intr[5], dist[8], se3[7], max_fov, T[16];
  if (fread(model, 1, 64, f) != 64) return 4;
  if (fread(&W, 4, 1, f) != 1 || fread(&H, 4, 1, f) != 1 || fread(&N, 4, 1, f) != 1 || fread(&bins, 4, 1, f) != 1 || fread(&nintr, 4, 1, f) != 1 || fread(&ndist, 4, 1, f) != 1) return 4;
  if (fread(intr, 8, 5, f) != 5 || fread(dist, 8, 8, f) != 8 || fread(se3, 8, 7, f) != 7 || fread(&max_fov, 8, 1, f) != 1 || fread(T, 8, 16, f) != 16) return 4;
  std::vector<uint8_t> img8(size_t(W) * H), src_bins(size_t(W) * H);
  if (fread(img8.data(), 1, img8.size(), f) != img8.size()) return 4;
  std::vector<double> pts(size_t(N) * 4), ints(static_cast<size_t>(N));
  if (fread(pts.data(), 8, pts.size(), f) != pts.size() || fread(ints.data(), 8, ints.size(), f) != ints.size()) return 4;
  fclose(f);
  // bin image from the CV_64FC1 image the functor receives: min(int(pix * B), B - 1), pix = u8 * (1 / 255)  (nid_cost.hpp:78-79)
  for (size_t k = 0; k < img8.size(); k++) src_bins[k] = uint8_t(std::max(0, std::min(int(img8[k] * (1.0 / 255.0) * bins), bins - 1)));
  CamParams<double> cam;
  for (int i = 0; i < 5; i++) cam.intr[i] = intr[i];
  for (int i = 0; i < 8; i++) cam.dist[i] = dist[i];
  const std::string m(model);
  if (m == "atan") cam_derive<double>(MODEL_ATAN, cam);
  if (m == "plumb_bob") return run<MODEL_PLUMB_BOB>(cam, W, H, bins, src_bins, pts, ints, se3);
  if (m == "fisheye" || m == "equidistant") return run<MODEL_FISHEYE>(cam, W, H, bins, src_bins, pts, ints, se3);
  if (m == "omnidir") return run<MODEL_OMNIDIR>(cam, W, H, bins, src_bins, pts, ints, se3);
  if (m == "equirectangular") return run<MODEL_EQUIRECT>(cam, W, H, bins, src_bins, pts, ints, se3);
  if (m == "atan") return run<MODEL_ATAN>(cam, W, H, bins, src_bins, pts, ints, se3);
  if (m == "rational_polynomial") return run<MODEL_RATIONAL>(cam, W, H, bins, src_bins, pts, ints, se3);
  return 5;
}
