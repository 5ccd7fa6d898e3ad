// double-precision SPLINE (NIDCost) kernels + projection utility.  Built with -ffp-contract=off like every other
// translation unit (csrc/Makefile): the fusions wanted are explicit fma calls, so a point gets the same arithmetic
// whichever unrolled slot / chunk / GPU processes it.
#include "nid_launch_impl.hpp"

namespace nidreg {

template <> hipError_t launch_spline_hist<double>(const PassArgs& a) {
  if (a.nchunks == 0) return hipSuccess;
  return a.rec64 ? launch_spline_hist_rec<double, Rec64>(a) : launch_spline_hist_rec<double, Rec32>(a);
}
template <> hipError_t launch_spline_grad<double>(const PassArgs& a) {
  if (a.nchunks == 0) return hipSuccess;
  return a.rec64 ? launch_spline_grad_rec<double, Rec64>(a) : launch_spline_grad_rec<double, Rec32>(a);
}
template <> int occupancy_spline_hist<double>(const PassArgs& a) { return a.rec64 ? occupancy_spline_hist_rec<double, Rec64>(a) : occupancy_spline_hist_rec<double, Rec32>(a); }
template <> int occupancy_spline_grad<double>(const PassArgs& a) { return a.rec64 ? occupancy_spline_grad_rec<double, Rec64>(a) : occupancy_spline_grad_rec<double, Rec32>(a); }
template <> hipError_t launch_project<double>(int model, const double* intr, const double* dist, const double* p3, long long n, double* uv, double* jac, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  struct { int model; } a{model};
  const CamParams<double> cam = make_cam<double>(intr, dist);
  const unsigned grid = unsigned((n + 255) / 256);
#define NID_LAUNCH(M) hipLaunchKernelGGL((k_project<M, double>), dim3(grid), dim3(256), 0, stream, p3, n, cam, uv, jac)
  NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
  return hipGetLastError();
}

}  // namespace nidreg


#ifdef NID_STAMP
// development aid (tools/stage_times.py, an instrumented build loaded through NIDREG_LIB): the stage stamps of the last launch
extern "C" int nidreg_debug_stage_stamps(unsigned long long* out, int words) {
  return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(nidreg::g_stage), size_t(words) * sizeof(unsigned long long)));
}
extern "C" int nidreg_debug_wg_stamps(unsigned long long* out, int words) {
  return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(nidreg::g_stamp), size_t(words) * sizeof(unsigned long long)));
}
#endif
