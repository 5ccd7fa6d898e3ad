// double-precision NEAREST (CostCalculatorNID) kernels.  Build with -ffp-contract=off: the arithmetic
// follows the reference's expression tree, so +,-,*,/,sqrt are bit-identical to the CPU's and the
// integer histogram is exactly reproducible.
#include "nid_launch_impl.hpp"
#include "nid_cull_kernels.hpp"
#include "nid_render_kernels.hpp"

namespace nidreg {

template <> hipError_t launch_nearest_hist<double>(const PassArgs& a) {
  if (a.nchunks == 0) return hipSuccess;
  return a.rec64 ? launch_nearest_hist_rec<double, Rec64>(a) : launch_nearest_hist_rec<double, Rec32>(a);
}
template <> int occupancy_nearest_hist<double>(const PassArgs& a) { return a.rec64 ? occupancy_nearest_hist_rec<double, Rec64>(a) : occupancy_nearest_hist_rec<double, Rec32>(a); }
hipError_t launch_cull(int model, const double* intr, const double* dist, const double* d_pts, long long stride_d, long long n, const double* T, int W, int H, double min_z,
                       int depth, int* d_pix, unsigned int* d_zbuf, unsigned char* d_keep, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  struct { int model; } a{model};
  const CamParams<double> cam = make_cam<double>(model, intr, dist);
  IsoParams<double> iso;
  for (int k = 0; k < 12; k++) iso.m[k] = T[k];
  const unsigned grid = unsigned((n + 255) / 256);
#define NID_LAUNCH(M) hipLaunchKernelGGL((k_cull_zbuf<M>), dim3(grid), dim3(256), 0, stream, d_pts, stride_d, n, iso, cam, W, H, min_z, depth, d_pix, d_zbuf)
  NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_cull_keep, dim3(grid), dim3(256), 0, stream, d_pts, stride_d, n, iso, depth, d_pix, d_zbuf, d_keep);
  return hipGetLastError();
}


hipError_t launch_colorize(int model, const double* intr, const double* dist, const double* d_pts, long long stride_d, long long n, const double* T, const uint8_t* d_img, int W, int H,
                           double min_nz, const float* d_icolor, double blend_weight, float* d_out, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  struct { int model; } a{model};
  const CamParams<double> cam = make_cam<double>(model, intr, dist);
  IsoParams<double> iso;
  for (int k = 0; k < 12; k++) iso.m[k] = T[k];
  // Eigen: Vector4f * double converts the scalar to float first (points_color_updater.cpp:57)
  const float wf = float(blend_weight), omwf = float(1.0 - blend_weight);
  const unsigned grid = unsigned((n + 255) / 256);
#define NID_LAUNCH(M)                                                                                                                                                  \
  hipLaunchKernelGGL((k_colorize<M>), dim3(grid), dim3(256), 0, stream, d_pts, stride_d, n, iso, cam, d_img, W, H, min_nz, reinterpret_cast<const float4*>(d_icolor), wf, omwf, \
                     reinterpret_cast<float4*>(d_out))
  NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
  return hipGetLastError();
}

hipError_t launch_lidar_image(int model, const double* intr, const double* dist, const double* d_pts, long long stride_d, const double* d_intensities, long long n, const double* T, int W,
                              int H, double min_nz, int* d_pix, u64* d_zmin, int* d_index_image, double* d_intensity_image, hipStream_t stream) {
  struct { int model; } a{model};
  const CamParams<double> cam = make_cam<double>(model, intr, dist);
  IsoParams<double> iso;
  for (int k = 0; k < 12; k++) iso.m[k] = T[k];
  const long long npix = (long long)W * H;
  hipError_t e = hipMemsetAsync(d_zmin, 0xff, size_t(npix) * sizeof(u64), stream);       // > every finite distance
  if (e == hipSuccess) e = hipMemsetAsync(d_index_image, 0xff, size_t(npix) * sizeof(int), stream);  // -1
  if (e != hipSuccess) return e;
  if (n > 0) {
    const unsigned grid = unsigned((n + 255) / 256);
#define NID_LAUNCH(M) hipLaunchKernelGGL((k_lidar_zmin<M>), dim3(grid), dim3(256), 0, stream, d_pts, stride_d, n, iso, cam, W, H, min_nz, d_pix, d_zmin)
    NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_lidar_argmax, dim3(grid), dim3(256), 0, stream, d_pts, stride_d, n, iso, d_pix, d_zmin, d_index_image);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_lidar_resolve, dim3(unsigned((npix + 255) / 256)), dim3(256), 0, stream, d_intensities, d_index_image, npix, d_intensity_image);
  return hipGetLastError();
}

}  // namespace nidreg
