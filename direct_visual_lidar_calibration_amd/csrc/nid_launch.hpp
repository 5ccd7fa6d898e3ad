// nid_launch.hpp -- host-visible launch wrappers around the templated kernels.  The double and the
// float instantiations live in separate translation units because they are compiled with different
// floating-point contraction rules (nid_kernels_f64.hip: -ffp-contract=off so +,-,*,/,sqrt match the
// CPU bit for bit; nid_kernels_f32.hip: fused multiply-adds allowed).
#pragma once
#include <string>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <vector>

#include "nid_multi.hpp"

namespace nidreg {

struct Chunk;
struct EntropyScalars;
struct ShardTable;
// k_nearest_hist's fast decision tier (nid_kernels.hpp): error-bound coefficients of this pose and camera, from the host
struct NearestFastArgs {
  double er, et, A, Bc;
  double C, D, Bc2;
  int on;
  const double* tab_c;  // equirectangular: boundary tables of the handle (device memory; nid_kernels.hpp NearestFast)
  const double* tab_r;
  int kmax, jmax;
};

struct PassArgs {
  int model;
  int rec64;  // 1: Rec64 records (double xyz), 0: Rec32
  const void* pts;
  const Chunk* chunks;
  int nchunks;
  int nslots;  // segments of the gradient pass's table: one 12-double partial each
  int seg;     // the table of THIS launch has chunks that run across column groups: the looped (SEG) kernel instantiations
  const uint32_t* gend;  // end offsets of the column groups among the records (a chunk may run across group boundaries)
  const uint8_t* img;  // padded, edge-replicated bin image
  int pitch, W, H, B, GW, cshift;
  int wide;  // SPLINE histogram pass: the 512-thread / 32-copy single-column specialisation (B = 256, GW = 1)
  double R[9], t[3];  // SPLINE pose
  double iso[12];     // NEAREST pose (rows 0..2 of the 4x4)
  double intr[5], dist[8];
  double magic;     // 2^(frac_bits - 1074): subnormal pre-scale of the x-weights
  double inv_unit;  // 2^(-frac_bits)
  double cos_fov;
  NearestFastArgs nfast;  // NEAREST: the fast decision tier's band coefficients (on = 0: exact tier only)
  unsigned long long* hist;
  const double* phi_q;
  const EntropyScalars* scal;
  double* partials;
  double q[4];          // quaternion of the pose (gradient chain rule)
  double* out;          // device result block
  double* out_host;     // host-mapped mirror (nullable)
  double tag;           // completion tag written behind the results (host polls it)
  unsigned int* counter;  // last-workgroup ticket
  hipStream_t stream;
  size_t lds_hist, lds_grad;
  const MultiEntry* multi;  // non-NULL: one grid over several pairs (chunks / nchunks are then the combined table)
  MultiDyn dyn;
  // the gradient kernel runs the entropy tail itself (k_entropy launched with tail = 0): nid_kernels.hpp GradTail
  double* gt_phi_q;
  double* gt_hist_image;
  double* gt_hist_points;
  EntropyScalars* gt_scal;
  int gt_from_partials;
  void* gt_zero_buf;       // gt_from_partials == 2 (GradTail): the next evaluation's histogram buffer, cleared by the gradient kernel
  long long gt_zero_words;
  int prio;  // progress priority (s_setprio) in the spline passes: set when the evaluation has its device to itself
};

template <typename real> hipError_t launch_spline_hist(const PassArgs& a);
template <typename real> hipError_t launch_spline_grad(const PassArgs& a);
template <typename real> hipError_t launch_nearest_hist(const PassArgs& a);
// workgroups of the selected kernel instantiation (model, record type, tiling) that fit on one CU at once
// (hipOccupancyMaxActiveBlocksPerMultiprocessor; 0 on error): a pass gets exactly one round of co-resident workgroups
template <typename real> int occupancy_spline_hist(const PassArgs& a);
template <typename real> int occupancy_spline_grad(const PassArgs& a);
template <typename real> int occupancy_nearest_hist(const PassArgs& a);
// ONE launch per cost+Jacobian evaluation (nid_fused.hpp; small tables, grids of co-resident workgroups): a.chunks / a.nchunks =
// the fused table, a.hist = this evaluation's histogram buffer, a.gt_zero_buf the next one's
struct FusedArgs {
  unsigned long long* barrier;         // arrival counter of the grid barrier (device memory, counts over the handle's lifetime)
  unsigned long long barrier_target;   // arrivals once every workgroup of THIS launch has arrived
  unsigned long long* flags;           // one release word per workgroup, kFusedFlagStride (16) words apart
  unsigned long long epoch;            // this launch's number: what the releasing workgroup stores into the flags
  unsigned long long timeout_ticks;    // 100 MHz wall-clock ticks a workgroup waits at the barrier before it gives up
  int cap;                             // points of LDS stash per workgroup (>= the longest chunk)
  int full;                            // 1: (u, v) + projection context + patch + record per point, 0: (u, v) only
};
size_t fused_lds_bytes_for(const PassArgs& a, int full, int cap);
hipError_t launch_spline_fused(const PassArgs& a, const FusedArgs& f);
int occupancy_spline_fused(const PassArgs& a, const FusedArgs& f);  // workgroups of that instantiation per CU (0 on error)
template <typename real> hipError_t launch_project(int model, const double* intr, const double* dist, const double* p3, long long n, double* uv, double* jac, hipStream_t stream);

#ifdef NID_EXP_HANDOFF
hipError_t set_handoff_buffer(void* p);  // nid_kernels_f64.hip (experiment)
#endif
// the same scalar projection code run on the host (nid_kernels_f64.hip): host arrays, fp64, 0 = ok
int project_host(int model, const double* intr, const double* dist, const double* p3, long long n, double* uv, double* jac);

// ViewCulling::cull (nid_cull_kernels.hpp); all pointers are device memory
hipError_t launch_cull(int model, const double* intr, const double* dist, const double* d_pts, long long stride_d, long long n, const double* T, int W, int H, double min_z,
                       int depth, int* d_pix, unsigned int* d_zbuf, unsigned char* d_keep, hipStream_t stream);

// PointsColorUpdater::update / generate_lidar_image (nid_render_kernels.hpp); all pointers are device memory,
// T = rows of the 4x4 T_camera_lidar
hipError_t launch_colorize(int model, const double* intr, const double* dist, const double* d_pts, long long stride_d, long long n, const double* T, const uint8_t* d_img, int W, int H,
                           double min_nz, const float* d_icolor, double blend_weight, float* d_out, hipStream_t stream);
hipError_t launch_lidar_image(int model, const double* intr, const double* dist, const double* d_pts, long long stride_d, const double* d_intensities, long long n, const double* T, int W,
                              int H, double min_nz, int* d_pix, u64* d_zmin, int* d_index_image, double* d_intensity_image, hipStream_t stream);

// preprocess.cpp:464-473 rank equalisation in place on a device array (nid_build.hip; synchronises the stream)
hipError_t equalize_intensities_device(double* d_intensities, long long n, hipStream_t stream);

// error text of the calling thread (nidreg_last_error); returns `code`
int fail(int code, const std::string& msg);

// view-culling parameters for the device-side record build (all host values; T = rows of the 4x4)
struct CullArgs {
  int model;
  double intr[5], dist[8];
  double T[16];
  int W, H;
  double min_z;
  int depth;
};

// Per-device scratch arena for handle construction: the reference builds a new NIDCost per pair per outer
// iteration (visual_camera_calibration.cpp:199-208), and hipMalloc / hipFree of the temporaries (upload
// staging, sort keys, rocPRIM scratch) used to cost more than the kernels.  One grow-only allocation per
// device, carved linearly; the lock is held for the whole construction (the reference constructs its cost
// objects sequentially).  nidreg_trim() releases it.
class ScratchArena {
 public:
  static ScratchArena& of(int device);
  void lock() { mu_.lock(); }
  void unlock() { mu_.unlock(); }
  hipError_t reserve(size_t bytes);  // grow to >= bytes (invalidates earlier carves), rewinds the cursor
  void* carve(size_t bytes);         // 256-byte aligned; nullptr when the reservation is exhausted
  void release();                    // hipFree (called with the lock held)
 private:
  std::mutex mu_;
  void* base_ = nullptr;
  size_t cap_ = 0, cur_ = 0;
};

// scratch bytes build_records_device carves for n input points (+ a W x H depth buffer when culling)
size_t build_scratch_bytes(long long n, bool cull, int W, int H);

// [cull ->] bucket -> Morton sort -> gather on the device (nid_build.hip).  d_pts: n x 4 doubles (x y z 1),
// cull nullable.  input_order: keep the caller's order inside each column group (stable sort on the group bits
// only) instead of the Morton order.  Temporaries come from `arena` (reserved by the caller for at least
// build_scratch_bytes).  Returns the record buffer (hipMalloc, caller owns), its type, and the column-group offsets.
// Bsrc = the caller's bin count (bin_points = clamp(int(intensity * Bsrc))); d_lut (nullable; bins > 256): occupied bin -> compact bin.
hipError_t build_records_device(
  const double* d_pts, const double* d_intensities, long long n, const CullArgs* cull, int Bsrc, const uint16_t* d_lut, int GW, int NG, bool force_rec32, bool input_order, ScratchArena& arena,
  void** d_recs_out, int* rec64_out, std::vector<int64_t>& gcount, hipStream_t stream);
// which of the B bins the n device-resident values occupy (used_host: B bytes)
hipError_t mark_bins_device(const double* d_v, long long n, int B, unsigned char* used_host);

// bin image (nid_device.hpp load_patch layout: strips of four rows, padded by 1 left/top and >= 2 right/bottom,
// edge replicated) from the caller's CV_64FC1 / CV_8UC1 image already uploaded to d_src (row stride in bytes)
hipError_t build_bin_image_device(const void* d_src, int is_f64, long long row_stride, int W, int H, int B, const uint16_t* d_lut, int pitch, int nstrips, uint8_t* d_img, hipStream_t stream);

}  // namespace nidreg
