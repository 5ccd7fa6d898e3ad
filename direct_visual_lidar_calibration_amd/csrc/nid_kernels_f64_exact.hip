// double-precision NEAREST (CostCalculatorNID) kernels.  Build with -ffp-contract=off: the arithmetic
// follows the reference's expression tree, so +,-,*,/,sqrt are bit-identical to the CPU's and the
// integer histogram is exactly reproducible.
#include "nid_launch_impl.hpp"
#include "nid_cull_kernels.hpp"

namespace nidreg {

template <> hipError_t launch_nearest_hist<double>(const PassArgs& a) {
  if (a.nchunks == 0) return hipSuccess;
  return a.rec64 ? launch_nearest_hist_rec<double, Rec64>(a) : launch_nearest_hist_rec<double, Rec32>(a);
}
hipError_t launch_cull(int model, const double* intr, const double* dist, const double* d_pts, long long stride_d, long long n, const double* T, int W, int H, double min_z,
                       int depth, int* d_pix, unsigned int* d_zbuf, unsigned char* d_keep, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  struct { int model; } a{model};
  const CamParams<double> cam = make_cam<double>(intr, dist);
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

}  // namespace nidreg
