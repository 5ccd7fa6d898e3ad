// single-precision-geometry instantiations (throughput mode): float transform / projection / weights,
// float32 point records; the histogram is still accumulated in 64-bit fixed point.
#include "nid_launch_impl.hpp"

namespace nidreg {

template <> hipError_t launch_spline_hist<float>(const PassArgs& a) {
  if (a.nchunks == 0) return hipSuccess;
  if (a.rec64) return hipErrorInvalidValue;
  return launch_spline_hist_rec<float, Rec32>(a);
}
template <> hipError_t launch_spline_grad<float>(const PassArgs& a) {
  if (a.nchunks == 0) return hipSuccess;
  if (a.rec64) return hipErrorInvalidValue;
  return launch_spline_grad_rec<float, Rec32>(a);
}
template <> int occupancy_spline_hist<float>(const PassArgs& a) { return a.rec64 ? 0 : occupancy_spline_hist_rec<float, Rec32>(a); }
template <> int occupancy_spline_grad<float>(const PassArgs& a) { return a.rec64 ? 0 : occupancy_spline_grad_rec<float, Rec32>(a); }
template <> hipError_t launch_nearest_hist<float>(const PassArgs& a) {
  if (a.nchunks == 0) return hipSuccess;
  if (a.rec64) return hipErrorInvalidValue;
  return launch_nearest_hist_rec<float, Rec32>(a);
}
template <> hipError_t launch_project<float>(int model, const double* intr, const double* dist, const double* p3, long long n, double* uv, double* jac, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  struct { int model; } a{model};
  const CamParams<float> cam = make_cam<float>(intr, dist);
  const unsigned grid = unsigned((n + 255) / 256);
#define NID_LAUNCH(M) hipLaunchKernelGGL((k_project<M, float>), dim3(grid), dim3(256), 0, stream, p3, n, cam, uv, jac)
  NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
  return hipGetLastError();
}

}  // namespace nidreg
