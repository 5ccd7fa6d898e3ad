// the fused one-launch evaluation (nid_fused.hpp): instantiations for every camera model, both record types, both stash formats.
// Built with -ffp-contract=off like every other translation unit: the per-point arithmetic is k_spline_hist's / k_spline_grad's.
#include "nid_fused.hpp"
#include "nid_launch_impl.hpp"

namespace nidreg {

size_t fused_lds_bytes_for(const PassArgs& a, int full, int cap) { return fused_lds_bytes(a.B, a.GW, a.cshift, full != 0, a.rec64 ? sizeof(Rec64) : sizeof(Rec32), cap); }

template <typename Rec, bool FULL>
static hipError_t launch_fused_rec(const PassArgs& a, const FusedArgs& f, bool occupancy_only, int* occ) {
  const PoseParams<double> pose = make_pose<double>(a);
  const CamParams<double> cam = make_cam<double>(a.model, a.intr, a.dist);
  GradTail gt;
  gt.phi_q = a.gt_phi_q;
  gt.hist_image = a.gt_hist_image;
  gt.hist_points = a.gt_hist_points;
  gt.scal = a.gt_scal;
  gt.from_partials = 2;
  gt.zero_buf = static_cast<u64*>(a.gt_zero_buf);
  gt.zero_words = a.gt_zero_words;
  const size_t lds = fused_lds_bytes(a.B, a.GW, a.cshift, FULL, sizeof(Rec), f.cap);
#define NID_LAUNCH(M)                                                                                                                                  \
  {                                                                                                                                                    \
    auto k = k_spline_fused<M, Rec, FULL>;                                                                                                             \
    hipError_t e = ensure_lds(k, lds);                                                                                                                 \
    if (e != hipSuccess) return e;                                                                                                                     \
    if (occupancy_only) {                                                                                                                              \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(occ, reinterpret_cast<const void*>(k), kThreads, lds) != hipSuccess) *occ = 0;                  \
    } else {                                                                                                                                           \
      hipLaunchKernelGGL(k, dim3(a.nchunks), dim3(kThreads), lds, a.stream, static_cast<const Rec*>(a.pts), a.chunks, a.img, a.pitch, a.W, a.H, pose, cam, a.B, a.GW, a.cshift, \
                         a.magic, a.inv_unit, a.hist, gt, a.partials, a.q[0], a.q[1], a.q[2], a.q[3], a.out, a.out_host, a.tag, a.counter, f.barrier, f.barrier_target,         \
                         f.flags, f.epoch, f.timeout_ticks, f.cap);                                                                                                      \
    }                                                                                                                                                  \
  }
  NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
  return occupancy_only ? hipSuccess : hipGetLastError();
}

hipError_t launch_spline_fused(const PassArgs& a, const FusedArgs& f) {
  if (a.nchunks == 0) return hipErrorInvalidValue;
  if (a.rec64) return f.full ? launch_fused_rec<Rec64, true>(a, f, false, nullptr) : launch_fused_rec<Rec64, false>(a, f, false, nullptr);
  return f.full ? launch_fused_rec<Rec32, true>(a, f, false, nullptr) : launch_fused_rec<Rec32, false>(a, f, false, nullptr);
}
int occupancy_spline_fused(const PassArgs& a, const FusedArgs& f) {
  int n = 0;
  hipError_t e;
  if (a.rec64) e = f.full ? launch_fused_rec<Rec64, true>(a, f, true, &n) : launch_fused_rec<Rec64, false>(a, f, true, &n);
  else e = f.full ? launch_fused_rec<Rec32, true>(a, f, true, &n) : launch_fused_rec<Rec32, false>(a, f, true, &n);
  return e == hipSuccess ? n : 0;
}

}  // namespace nidreg

#ifdef NID_STAMP
// development aid (tools/fused_stage_times.py, an instrumented build loaded through NIDREG_LIB): the fused kernel's stage stamps
extern "C" int nidreg_debug_fused_stage_stamps(unsigned long long* out, int words) {
  return int(hipMemcpyFromSymbol(out, HIP_SYMBOL(nidreg::g_stage), size_t(words) * sizeof(unsigned long long)));
}
#endif
