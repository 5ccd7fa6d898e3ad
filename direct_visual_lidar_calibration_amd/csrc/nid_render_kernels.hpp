// nid_render_kernels.hpp -- the two point-to-image consumers that sit either side of the NID path in
// the reference (SURVEY.md "next" rows N2 and N4), sharing the CostCalculatorNID front end
// (transform, FoV gate on the normalised 3-vector, projection, truncating cast, in-image test):
//   k_colorize        PointsColorUpdater::update (src/vlcal/common/points_color_updater.cpp:37-61):
//                     per point, blend the image pixel with the point's intensity colour; zeros when
//                     the point is out of the FoV / image.
//   k_lidar_zmin      generate_lidar_image (src/vlcal/preprocess/generate_lidar_image.cpp:7-41), pass 1:
//                     per pixel atomicMin of the squared distance (fp64 bits order like u64 for d >= 0).
//   k_lidar_argmax    pass 2: among the points that attain the pixel's minimum, the LARGEST index wins --
//                     the reference's sequential loop overwrites on `!(stored < sq_dist)`, i.e. also on
//                     ties, so the last such point in index order is what it ends with.
//   k_lidar_resolve   pass 3: per pixel, intensity of the winner (0 where no point landed, index -1).
// The result is independent of thread order and identical to the sequential CPU loop.
// Compiled in the -ffp-contract=off translation unit with the exact-order projection: every
// +,-,*,/,sqrt matches the CPU bit for bit, so pixel assignments are identical.
#pragma once
#include "nid_device.hpp"

namespace nidreg {

// pixel index (py * W + px) of a LiDAR point, or -1 (out of FoV / out of image / non-finite).
// cx, cy, cz: the camera-frame point (Eigen 4x4 * (x y z w), summed left to right).
template <int MODEL>
__device__ __forceinline__ int point_to_pixel(const IsoParams<double>& iso, const CamParams<double>& cam, double x, double y, double z, double w, int W, int H, double min_nz, double& cx,
                                              double& cy, double& cz) {
  cx = ((iso.m[0] * x + iso.m[1] * y) + iso.m[2] * z) + iso.m[3] * w;
  cy = ((iso.m[4] * x + iso.m[5] * y) + iso.m[6] * z) + iso.m[7] * w;
  cz = ((iso.m[8] * x + iso.m[9] * y) + iso.m[10] * z) + iso.m[11] * w;
  const double n2 = (cx * cx + cy * cy) + cz * cz;
  const double zn = n2 > 0.0 ? cz / sqrt(n2) : cz;  // Eigen normalized(): unchanged when the norm is 0
  if (zn < min_nz) return -1;
  double u, v;
  project<MODEL, double, double, false>(cam, cx, cy, cz, u, v);
  // .cast<int>() truncates toward zero; NaN / overflow become INT_MIN on x86 (rejected by the < 0 test)
  if (!((u > -1.0) && (u < double(W)) && (v > -1.0) && (v < double(H)))) return -1;
  return int(v) * W + int(u);
}

template <int MODEL>
__global__ __launch_bounds__(256) void k_colorize(
  const double* __restrict__ pts, long long stride_d, long long n, IsoParams<double> iso, CamParams<double> cam, const uint8_t* __restrict__ img, int W, int H, double min_nz,
  const float4* __restrict__ icolor, float wf, float omwf, float4* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* p = pts + i * stride_d;
  double cx, cy, cz;
  const int q = point_to_pixel<MODEL>(iso, cam, p[0], p[1], p[2], p[3], W, H, min_nz, cx, cy, cz);
  float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q >= 0) {
    const float g = float(img[q]) / 255.0f;
    const float4 ic = icolor ? icolor[i] : make_float4(1.f, 1.f, 1.f, 1.f);
    // Vector4f(g, g, g, 1) * float(w) + intensity_color * float(1 - w), unfused
    c.x = g * wf + ic.x * omwf;
    c.y = g * wf + ic.y * omwf;
    c.z = g * wf + ic.z * omwf;
    c.w = 1.0f * wf + ic.w * omwf;
  }
  out[i] = c;
}

template <int MODEL>
__global__ __launch_bounds__(256) void k_lidar_zmin(
  const double* __restrict__ pts, long long stride_d, long long n, IsoParams<double> iso, CamParams<double> cam, int W, int H, double min_nz, int* __restrict__ pix,
  u64* __restrict__ zmin) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* p = pts + i * stride_d;
  double cx, cy, cz;
  const int q = point_to_pixel<MODEL>(iso, cam, p[0], p[1], p[2], p[3], W, H, min_nz, cx, cy, cz);
  pix[i] = q;
  if (q >= 0) {
    const double sq = (cx * cx + cy * cy) + cz * cz;
    atomicMin(&zmin[q], u64(__double_as_longlong(sq)));  // sq >= 0: the bit pattern is monotone
  }
}

__global__ __launch_bounds__(256) void k_lidar_argmax(
  const double* __restrict__ pts, long long stride_d, long long n, IsoParams<double> iso, const int* __restrict__ pix, const u64* __restrict__ zmin, int* __restrict__ index_image) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int q = pix[i];
  if (q < 0) return;
  const double* p = pts + i * stride_d;
  const double x = p[0], y = p[1], z = p[2], w = p[3];
  const double cx = ((iso.m[0] * x + iso.m[1] * y) + iso.m[2] * z) + iso.m[3] * w;
  const double cy = ((iso.m[4] * x + iso.m[5] * y) + iso.m[6] * z) + iso.m[7] * w;
  const double cz = ((iso.m[8] * x + iso.m[9] * y) + iso.m[10] * z) + iso.m[11] * w;
  const double sq = (cx * cx + cy * cy) + cz * cz;
  if (u64(__double_as_longlong(sq)) == zmin[q]) atomicMax(&index_image[q], int(i));
}

__global__ __launch_bounds__(256) void k_lidar_resolve(const double* __restrict__ intensities, const int* __restrict__ index_image, long long npix, double* __restrict__ intensity_image) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= npix) return;
  const int i = index_image[q];
  intensity_image[q] = i >= 0 ? intensities[i] : 0.0;
}

}  // namespace nidreg
