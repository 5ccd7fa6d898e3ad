// nid_launch_impl.hpp -- bodies of the launch wrappers; included by exactly one TU per `real`.
#pragma once
#include <mutex>
#include <utility>
#include <vector>

#include "nid_kernels.hpp"
#include "nid_launch.hpp"

namespace nidreg {

template <typename real>
static CamParams<real> make_cam(int model, const double* intr, const double* dist) {
  CamParams<real> c;
  for (int i = 0; i < 5; i++) c.intr[i] = real(intr[i]);
  for (int i = 0; i < 8; i++) c.dist[i] = real(dist[i]);
  cam_derive<real>(model, c);  // (constants a model derives from its coefficients: nid_device.hpp)
  return c;
}
template <typename real>
static PoseParams<real> make_pose(const PassArgs& a) {
  PoseParams<real> p;
  for (int i = 0; i < 9; i++) p.R[i] = real(a.R[i]);
  for (int i = 0; i < 3; i++) p.t[i] = real(a.t[i]);
  return p;
}
template <typename real>
static IsoParams<real> make_iso(const PassArgs& a) {
  IsoParams<real> p;
  for (int i = 0; i < 12; i++) p.m[i] = real(a.iso[i]);
  return p;
}

// A kernel that needs more than 64 KB of dynamic LDS must be told so once per (kernel, device); doing it on every
// launch put a runtime call on the critical path of each evaluation of the headline configuration.
template <typename K>
static hipError_t ensure_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  static std::mutex mu;
  static std::vector<std::pair<const void*, std::pair<int, size_t>>> done;  // (kernel, (device, bytes granted))
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const void* key = reinterpret_cast<const void*>(kernel);
  std::lock_guard<std::mutex> lk(mu);
  for (const auto& d : done)
    if (d.first == key && d.second.first == dev && d.second.second >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes));
  if (e == hipSuccess) done.push_back(std::make_pair(key, std::make_pair(dev, bytes)));
  return e;
}

#define NID_MODEL_SWITCH(MACRO)                       \
  switch (a.model) {                                  \
    case MODEL_PLUMB_BOB: MACRO(MODEL_PLUMB_BOB); break; \
    case MODEL_FISHEYE: MACRO(MODEL_FISHEYE); break;  \
    case MODEL_OMNIDIR: MACRO(MODEL_OMNIDIR); break;  \
    case MODEL_EQUIRECT: MACRO(MODEL_EQUIRECT); break; \
    case MODEL_ATAN: MACRO(MODEL_ATAN); break;        \
    case MODEL_RATIONAL: MACRO(MODEL_RATIONAL); break; \
    default: return hipErrorInvalidValue;             \
  }

template <typename real, typename Rec>
static hipError_t launch_spline_hist_rec(const PassArgs& a) {
  const PoseParams<real> pose = make_pose<real>(a);
  const CamParams<real> cam = make_cam<real>(a.model, a.intr, a.dist);
  // SEG: the table has chunks that run across column-group boundaries (nid_kernels.hpp Segments)
#define NID_LAUNCH_W(M, WIDE, THREADS, SEG)                                                                                                            \
  if (a.multi) {                                                                                                                                       \
    auto k = k_spline_hist<M, Rec, real, WIDE, true, SEG>;                                                                                             \
    hipError_t e = ensure_lds(k, a.lds_hist);                                                                                                          \
    if (e != hipSuccess) return e;                                                                                                                     \
    hipLaunchKernelGGL(k, dim3(a.nchunks), dim3(THREADS), a.lds_hist, a.stream, static_cast<const Rec*>(a.pts), a.chunks, a.gend, a.img, a.pitch, a.W, a.H, pose, \
                       cam, a.B, a.GW, a.cshift, a.magic, a.hist, a.prio, a.multi, a.dyn);                             \
  } else {                                                                                                                                             \
    auto k = k_spline_hist<M, Rec, real, WIDE, false, SEG>;                                                                                            \
    hipError_t e = ensure_lds(k, a.lds_hist);                                                                                                          \
    if (e != hipSuccess) return e;                                                                                                                     \
    hipLaunchKernelGGL(k, dim3(a.nchunks), dim3(THREADS), a.lds_hist, a.stream, static_cast<const Rec*>(a.pts), a.chunks, a.gend, a.img, a.pitch, a.W, a.H, pose, \
                       cam, a.B, a.GW, a.cshift, a.magic, a.hist, a.prio, a.multi, NoMultiDyn());                      \
  }
  if (a.wide) {  // B = 256, GW = 1, 32 copies, 512 threads (see k_spline_hist)
    if (a.seg) {
#define NID_LAUNCH(M) NID_LAUNCH_W(M, true, kWideThreads, true)
      NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
    } else {
#define NID_LAUNCH(M) NID_LAUNCH_W(M, true, kWideThreads, false)
      NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
    }
  } else {
    if (a.seg) {
#define NID_LAUNCH(M) NID_LAUNCH_W(M, false, kThreads, true)
      NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
    } else {
#define NID_LAUNCH(M) NID_LAUNCH_W(M, false, kThreads, false)
      NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
    }
  }
#undef NID_LAUNCH_W
  return hipGetLastError();
}

template <typename real, typename Rec>
static hipError_t launch_spline_grad_rec(const PassArgs& a) {
  const PoseParams<real> pose = make_pose<real>(a);
  const CamParams<real> cam = make_cam<real>(a.model, a.intr, a.dist);
  GradTail gt;
  gt.phi_q = a.gt_phi_q;
  gt.hist_image = a.gt_hist_image;
  gt.hist_points = a.gt_hist_points;
  gt.scal = a.gt_scal;
  gt.from_partials = a.gt_from_partials;
  gt.zero_buf = static_cast<u64*>(a.gt_zero_buf);
  gt.zero_words = a.gt_zero_words;
  if (a.seg && a.GW != 1) return hipErrorInvalidValue;  // multi-segment tables are built for the single-column kernels only
#define NID_LAUNCH_G(M, GW1, SEG)                                                                                                                      \
  if (a.multi) {                                                                                                                                       \
    auto k = k_spline_grad<M, Rec, real, GW1, true, SEG>;                                                                                              \
    hipError_t e = ensure_lds(k, a.lds_grad);                                                                                                          \
    if (e != hipSuccess) return e;                                                                                                                     \
    hipLaunchKernelGGL(k, dim3(a.nchunks), dim3(kThreads), a.lds_grad, a.stream, static_cast<const Rec*>(a.pts), a.chunks, a.gend, a.img, a.pitch, a.W, a.H, pose, \
                       cam, a.B, a.GW, a.cshift, a.inv_unit, a.hist, a.phi_q, a.scal, gt, a.partials, unsigned(a.nslots), a.q[0], a.q[1], a.q[2], a.q[3], a.out, a.out_host, a.tag, \
                       a.counter, a.prio, a.multi, a.dyn);                                                                                             \
  } else {                                                                                                                                             \
    auto k = k_spline_grad<M, Rec, real, GW1, false, SEG>;                                                                                             \
    hipError_t e = ensure_lds(k, a.lds_grad);                                                                                                          \
    if (e != hipSuccess) return e;                                                                                                                     \
    hipLaunchKernelGGL(k, dim3(a.nchunks), dim3(kThreads), a.lds_grad, a.stream, static_cast<const Rec*>(a.pts), a.chunks, a.gend, a.img, a.pitch, a.W, a.H, pose, \
                       cam, a.B, a.GW, a.cshift, a.inv_unit, a.hist, a.phi_q, a.scal, gt, a.partials, unsigned(a.nslots), a.q[0], a.q[1], a.q[2], a.q[3], a.out, a.out_host, a.tag, \
                       a.counter, a.prio, a.multi, NoMultiDyn());                                                                                      \
  }
  if (a.GW == 1) {
    if (a.seg) {
#define NID_LAUNCH(M) NID_LAUNCH_G(M, true, true)
      NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
    } else {
#define NID_LAUNCH(M) NID_LAUNCH_G(M, true, false)
      NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
    }
  } else {
#define NID_LAUNCH(M) NID_LAUNCH_G(M, false, false)
    NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
  }
#undef NID_LAUNCH_G
  return hipGetLastError();
}

template <typename real, typename Rec>
static int occupancy_spline_hist_rec(const PassArgs& a) {
  int n = 0;
#define NID_OCC_W(M, WIDE, THREADS)                                                                                                   \
  {                                                                                                                                   \
    auto k = k_spline_hist<M, Rec, real, WIDE, false, false>;                                                                         \
    if (ensure_lds(k, a.lds_hist) != hipSuccess) return 0;                                                                            \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(k), THREADS, a.lds_hist) != hipSuccess) n = 0; \
  }
  if (a.wide) {
#define NID_LAUNCH(M) NID_OCC_W(M, true, kWideThreads)
    switch (a.model) {
      case MODEL_PLUMB_BOB: NID_LAUNCH(MODEL_PLUMB_BOB); break;
      case MODEL_FISHEYE: NID_LAUNCH(MODEL_FISHEYE); break;
      case MODEL_OMNIDIR: NID_LAUNCH(MODEL_OMNIDIR); break;
      case MODEL_EQUIRECT: NID_LAUNCH(MODEL_EQUIRECT); break;
      case MODEL_ATAN: NID_LAUNCH(MODEL_ATAN); break;
      case MODEL_RATIONAL: NID_LAUNCH(MODEL_RATIONAL); break;
      default: return 0;
    }
#undef NID_LAUNCH
  } else {
#define NID_LAUNCH(M) NID_OCC_W(M, false, kThreads)
    switch (a.model) {
      case MODEL_PLUMB_BOB: NID_LAUNCH(MODEL_PLUMB_BOB); break;
      case MODEL_FISHEYE: NID_LAUNCH(MODEL_FISHEYE); break;
      case MODEL_OMNIDIR: NID_LAUNCH(MODEL_OMNIDIR); break;
      case MODEL_EQUIRECT: NID_LAUNCH(MODEL_EQUIRECT); break;
      case MODEL_ATAN: NID_LAUNCH(MODEL_ATAN); break;
      case MODEL_RATIONAL: NID_LAUNCH(MODEL_RATIONAL); break;
      default: return 0;
    }
#undef NID_LAUNCH
  }
#undef NID_OCC_W
  return n;
}

template <typename real, typename Rec>
static int occupancy_spline_grad_rec(const PassArgs& a) {
  int n = 0;
#define NID_OCC_G(M, GW1)                                                                                                             \
  {                                                                                                                                   \
    auto k = k_spline_grad<M, Rec, real, GW1, false, false>;                                                                          \
    if (ensure_lds(k, a.lds_grad) != hipSuccess) return 0;                                                                            \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(k), kThreads, a.lds_grad) != hipSuccess) n = 0; \
  }
  if (a.GW == 1) {
#define NID_LAUNCH(M) NID_OCC_G(M, true)
    switch (a.model) {
      case MODEL_PLUMB_BOB: NID_LAUNCH(MODEL_PLUMB_BOB); break;
      case MODEL_FISHEYE: NID_LAUNCH(MODEL_FISHEYE); break;
      case MODEL_OMNIDIR: NID_LAUNCH(MODEL_OMNIDIR); break;
      case MODEL_EQUIRECT: NID_LAUNCH(MODEL_EQUIRECT); break;
      case MODEL_ATAN: NID_LAUNCH(MODEL_ATAN); break;
      case MODEL_RATIONAL: NID_LAUNCH(MODEL_RATIONAL); break;
      default: return 0;
    }
#undef NID_LAUNCH
  } else {
#define NID_LAUNCH(M) NID_OCC_G(M, false)
    switch (a.model) {
      case MODEL_PLUMB_BOB: NID_LAUNCH(MODEL_PLUMB_BOB); break;
      case MODEL_FISHEYE: NID_LAUNCH(MODEL_FISHEYE); break;
      case MODEL_OMNIDIR: NID_LAUNCH(MODEL_OMNIDIR); break;
      case MODEL_EQUIRECT: NID_LAUNCH(MODEL_EQUIRECT); break;
      case MODEL_ATAN: NID_LAUNCH(MODEL_ATAN); break;
      case MODEL_RATIONAL: NID_LAUNCH(MODEL_RATIONAL); break;
      default: return 0;
    }
#undef NID_LAUNCH
  }
#undef NID_OCC_G
  return n;
}

// workgroups of the (straight-line, single-pair) NEAREST kernel of this model that fit on one CU at once
template <typename real, typename Rec>
static int occupancy_nearest_hist_rec(const PassArgs& a) {
  int n = 0;
#define NID_LAUNCH(M)                                                                                                                 \
  {                                                                                                                                   \
    auto k = k_nearest_hist<M, Rec, real, false, false>;                                                                              \
    if (ensure_lds(k, a.lds_hist) != hipSuccess) return 0;                                                                            \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(k), kThreads, a.lds_hist) != hipSuccess) n = 0; \
  }
  switch (a.model) {
    case MODEL_PLUMB_BOB: NID_LAUNCH(MODEL_PLUMB_BOB); break;
    case MODEL_FISHEYE: NID_LAUNCH(MODEL_FISHEYE); break;
    case MODEL_OMNIDIR: NID_LAUNCH(MODEL_OMNIDIR); break;
    case MODEL_EQUIRECT: NID_LAUNCH(MODEL_EQUIRECT); break;
    case MODEL_ATAN: NID_LAUNCH(MODEL_ATAN); break;
    case MODEL_RATIONAL: NID_LAUNCH(MODEL_RATIONAL); break;
    default: return 0;
  }
#undef NID_LAUNCH
  return n;
}

template <typename real, typename Rec>
static hipError_t launch_nearest_hist_rec(const PassArgs& a) {
  const IsoParams<real> iso = make_iso<real>(a);
  const CamParams<real> cam = make_cam<real>(a.model, a.intr, a.dist);
  NearestFast fast;
  fast.er = a.nfast.er, fast.et = a.nfast.et, fast.A = a.nfast.A, fast.Bc = a.nfast.Bc, fast.C = a.nfast.C, fast.D = a.nfast.D, fast.Bc2 = a.nfast.Bc2, fast.on = a.nfast.on;
  fast.tab_c = a.nfast.tab_c, fast.tab_r = a.nfast.tab_r, fast.kmax = a.nfast.kmax, fast.jmax = a.nfast.jmax;
#define NID_LAUNCH_N(M, SEG)                                                                                                                           \
  if (a.multi) {                                                                                                                                       \
    auto k = k_nearest_hist<M, Rec, real, true, SEG>;                                                                                                       \
    hipError_t e = ensure_lds(k, a.lds_hist);                                                                                                          \
    if (e != hipSuccess) return e;                                                                                                                     \
    hipLaunchKernelGGL(k, dim3(a.nchunks), dim3(kThreads), a.lds_hist, a.stream, static_cast<const Rec*>(a.pts), a.chunks, a.gend, a.img, a.pitch, a.W, a.H, iso, cam,  \
                       a.B, a.GW, a.cshift, real(a.cos_fov), fast, a.hist, a.multi, a.dyn);                                                                   \
  } else {                                                                                                                                             \
    auto k = k_nearest_hist<M, Rec, real, false, SEG>;                                                                                                      \
    hipError_t e = ensure_lds(k, a.lds_hist);                                                                                                          \
    if (e != hipSuccess) return e;                                                                                                                     \
    hipLaunchKernelGGL(k, dim3(a.nchunks), dim3(kThreads), a.lds_hist, a.stream, static_cast<const Rec*>(a.pts), a.chunks, a.gend, a.img, a.pitch, a.W, a.H, iso, cam,  \
                       a.B, a.GW, a.cshift, real(a.cos_fov), fast, a.hist, a.multi, NoMultiDyn());                                                            \
  }
  if (a.seg) {
#define NID_LAUNCH(M) NID_LAUNCH_N(M, true)
    NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
  } else {
#define NID_LAUNCH(M) NID_LAUNCH_N(M, false)
    NID_MODEL_SWITCH(NID_LAUNCH)
#undef NID_LAUNCH
  }
#undef NID_LAUNCH_N
  return hipGetLastError();
}

}  // namespace nidreg
