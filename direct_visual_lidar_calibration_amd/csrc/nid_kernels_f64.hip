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
  const CamParams<double> cam = make_cam<double>(model, intr, dist);
  const unsigned grid = unsigned((n + 255) / 256);
#define NID_LAUNCH(M) hipLaunchKernelGGL((k_project<M, double>), dim3(grid), dim3(256), 0, stream, p3, n, cam, uv, jac)
  NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
  return hipGetLastError();
}

#ifdef NID_EXP_HANDOFF
hipError_t set_handoff_buffer(void* p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_uv_handoff), &p, sizeof(p)); }
#endif

// the same projection code on the HOST (nid_device.hpp's scalar math is __host__ __device__): for callers that project a
// handful of points at a time -- estimate_camera_fov inverts the projection at three pixels with NelderMead<2>, ~240 probes
// of ONE point (src/vlcal/common/estimate_fov.cpp:17-51), host work in the reference as well
int project_host(int model, const double* intr, const double* dist, const double* p3, long long n, double* uv, double* jac) {
  struct { int model; } a{model};
  const CamParams<double> cam = make_cam<double>(model, intr, dist);
#define NID_LAUNCH(M)                                                                                      \
  for (long long i = 0; i < n; i++) {                                                                      \
    double u, v, du[3], dv[3];                                                                             \
    project_jac<M, double>(cam, p3[3 * i], p3[3 * i + 1], p3[3 * i + 2], u, v, du, dv);                     \
    uv[2 * i] = u, uv[2 * i + 1] = v;                                                                      \
    if (jac)                                                                                               \
      for (int k = 0; k < 3; k++) jac[6 * i + k] = du[k], jac[6 * i + 3 + k] = dv[k];                      \
  }
  switch (a.model) {
    case MODEL_PLUMB_BOB: NID_LAUNCH(MODEL_PLUMB_BOB) break;
    case MODEL_FISHEYE: NID_LAUNCH(MODEL_FISHEYE) break;
    case MODEL_OMNIDIR: NID_LAUNCH(MODEL_OMNIDIR) break;
    case MODEL_EQUIRECT: NID_LAUNCH(MODEL_EQUIRECT) break;
    case MODEL_ATAN: NID_LAUNCH(MODEL_ATAN) break;
    case MODEL_RATIONAL: NID_LAUNCH(MODEL_RATIONAL) break;
    default: return -1;
  }
#undef NID_LAUNCH
  return 0;
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
