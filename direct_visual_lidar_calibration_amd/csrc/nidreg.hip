// nidreg.hip -- host side of the C ABI declared in include/nidreg.h.
//
// Owns: device residency of one LiDAR-camera pair (bucketed point records, padded bin image,
// fixed-point histogram, scratch), the per-evaluation launch sequence, and the multi-handle
// (multi-pair / multi-GPU) fan-out.  No CPU compute path exists here: every evaluation runs the
// HIP kernels of nid_kernels.hpp, and creation fails when no gfx950 device is usable.
#define NID_COMMON_KERNELS
#include "nid_kernels.hpp"
#include "nid_launch.hpp"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/nidreg.h"

using namespace nidreg;

namespace {

thread_local std::string g_last_error;
}  // namespace

namespace nidreg {
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
}  // namespace nidreg

namespace {

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) return fail(NIDREG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

// double -> int exactly as the reference's x86-64 build converts (cvttsd2si): NaN / overflow -> INT_MIN
inline int cast_int(double d) {
  if (!(d > -2147483649.0 && d < 2147483648.0)) return INT_MIN;
  return static_cast<int>(d);
}

constexpr int kEntropyCols = 16;  // histogram columns per k_entropy_partial workgroup
const int kNumIntr[6] = {4, 4, 5, 2, 4, 4};
const int kNumDist[6] = {5, 4, 4, 0, 1, 8};

}  // namespace

struct nidreg_handle {
  int device = 0;
  int model = 0, mode = 0, precision = 0, bins = 0;
  int W = 0, H = 0, pitch = 0;
  int GW = 0, NG = 0, cshift = 0;
  int wide = 0;  // k_spline_hist<.., WIDE>: B = 256, GW = 1, 32 copies, 512 threads
  int NEB = 0;  // entropy column blocks
  int frac_bits = 0;
  int rec64 = 0;
  int64_t num_points = 0;
  int nchunks = 0;       // gradient pass / generic histogram kernels
  int nchunks_hist = 0;  // WIDE histogram kernel's own table (0 = shares d_chunks)
  double intr[5] = {0}, dist[8] = {0};
  double max_fov = 0.0;

  hipStream_t stream = nullptr;
  bool own_stream = false;
  void* d_pts = nullptr;
  Chunk* d_chunks = nullptr;
  Chunk* d_chunks_hist = nullptr;
  uint8_t* d_img = nullptr;
  u64* d_hist = nullptr;      // histogram of the current / most recent evaluation
  // double buffering of the histogram (own buffers only): evaluation k accumulates into one buffer and
  // its k_entropy zeroes the OTHER one for evaluation k + 1, so no memset sits on the critical path
  u64* d_hist_buf[2] = {nullptr, nullptr};
  bool hist_zeroed[2] = {false, false};
  int hist_cur = 0;
  bool own_hist = false;
  double* d_out = nullptr;
  bool own_out = false;
  double* d_part_hj = nullptr;
  u64* d_row_part = nullptr;
  double* d_phi_q = nullptr;
  double* d_hist_image = nullptr;
  double* d_hist_points = nullptr;
  EntropyScalars* d_scal = nullptr;
  double* d_partials = nullptr;
  double* h_out = nullptr;       // pinned, host-mapped
  double* d_out_host = nullptr;  // device address of h_out (NULL when results live in ext_out)
  unsigned int* d_counters = nullptr;  // [0] entropy ticket, [1] gradient ticket
  double seq = 0.0;                    // completion tag of the evaluation in flight (host-mapped polling)
  unsigned int evals_since_reap = 0;

  size_t lds_hist = 0, lds_grad = 0, lds_entropy = 0;
  int64_t hist_words = 0;

  bool timing = false;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool ev_grad = false;
  double last_q[4] = {0, 0, 0, 1};
  double last_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double last_t[3] = {0, 0, 0};
};

namespace {

void free_handle(nidreg_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->d_pts) (void)hipFree(h->d_pts);
  if (h->d_chunks) (void)hipFree(h->d_chunks);
  if (h->d_chunks_hist) (void)hipFree(h->d_chunks_hist);
  if (h->d_img) (void)hipFree(h->d_img);
  if (h->own_hist) {
    if (h->d_hist_buf[0]) (void)hipFree(h->d_hist_buf[0]);
    if (h->d_hist_buf[1]) (void)hipFree(h->d_hist_buf[1]);
  }
  if (h->own_out && h->d_out) (void)hipFree(h->d_out);
  if (h->d_part_hj) (void)hipFree(h->d_part_hj);
  if (h->d_row_part) (void)hipFree(h->d_row_part);
  if (h->d_phi_q) (void)hipFree(h->d_phi_q);
  if (h->d_hist_image) (void)hipFree(h->d_hist_image);
  if (h->d_hist_points) (void)hipFree(h->d_hist_points);
  if (h->d_scal) (void)hipFree(h->d_scal);
  if (h->d_partials) (void)hipFree(h->d_partials);
  if (h->h_out) (void)hipHostFree(h->h_out);
  if (h->d_counters) (void)hipFree(h->d_counters);
  for (int i = 0; i < 6; i++)
    if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

void fill_pass_args(const nidreg_handle* h, PassArgs& a) {
  std::memset(&a, 0, sizeof(a));
  a.model = h->model;
  a.rec64 = h->rec64;
  a.pts = h->d_pts;
  a.chunks = h->d_chunks;
  a.nchunks = h->nchunks;
  a.img = h->d_img;
  a.pitch = h->pitch;
  a.W = h->W;
  a.H = h->H;
  a.B = h->bins;
  a.GW = h->GW;
  a.cshift = h->cshift;
  a.wide = h->wide;
  std::memcpy(a.intr, h->intr, sizeof(a.intr));
  std::memcpy(a.dist, h->dist, sizeof(a.dist));
  a.magic = std::ldexp(1.0, h->frac_bits - 1074);  // subnormal pre-scale of the x-weights (to_fixed_dn)
  a.inv_unit = std::ldexp(1.0, -h->frac_bits);
  a.cos_fov = std::cos(h->max_fov);
  a.hist = h->d_hist;
  a.phi_q = h->d_phi_q;
  a.scal = h->d_scal;
  a.partials = h->d_partials;
  for (int k = 0; k < 4; k++) a.q[k] = h->last_q[k];
  a.out = h->d_out;
  a.out_host = h->d_out_host;
  a.tag = h->seq;
  a.counter = h->d_counters + 1;
  a.stream = h->stream;
  a.lds_hist = h->lds_hist;
  a.lds_grad = h->lds_grad;
}

// R = I + 2 w [v]x + 2 [v]x^2 from the un-normalised quaternion (Sophus SO3 * point expanded)
void pose_from_se3(const double* se3, double* R, double* t) {
  const double x = se3[0], y = se3[1], z = se3[2], w = se3[3];
  R[0] = 1.0 - 2.0 * (y * y + z * z);
  R[1] = 2.0 * (x * y - w * z);
  R[2] = 2.0 * (x * z + w * y);
  R[3] = 2.0 * (x * y + w * z);
  R[4] = 1.0 - 2.0 * (x * x + z * z);
  R[5] = 2.0 * (y * z - w * x);
  R[6] = 2.0 * (x * z - w * y);
  R[7] = 2.0 * (y * z + w * x);
  R[8] = 1.0 - 2.0 * (x * x + y * y);
  t[0] = se3[4];
  t[1] = se3[5];
  t[2] = se3[6];
}

// Select the buffer this evaluation accumulates into and make sure it is zero.  With own (double)
// buffers the previous evaluation's k_entropy has already zeroed it; a caller-provided buffer
// (ext_hist: the sharded protocol all-reduces it in place) or a buffer left dirty by a failed launch
// is cleared with a memset.
hipError_t begin_histogram(nidreg_handle* h) {
  if (h->own_hist) {
    h->hist_cur ^= 1;
    h->d_hist = h->d_hist_buf[h->hist_cur];
    if (!h->hist_zeroed[h->hist_cur]) {
      hipError_t e = hipMemsetAsync(h->d_hist, 0, size_t(h->hist_words) * sizeof(u64), h->stream);
      if (e != hipSuccess) return e;
    }
    h->hist_zeroed[h->hist_cur] = false;  // about to be written
    return hipSuccess;
  }
  return hipMemsetAsync(h->d_hist, 0, size_t(h->hist_words) * sizeof(u64), h->stream);
}

int launch_hist_spline(nidreg_handle* h, const double* se3) {
  PassArgs a;
  fill_pass_args(h, a);
  if (h->d_chunks_hist) {
    a.chunks = h->d_chunks_hist;
    a.nchunks = h->nchunks_hist;
  }
  pose_from_se3(se3, a.R, a.t);
  for (int k = 0; k < 4; k++) h->last_q[k] = se3[k];
  std::memcpy(h->last_R, a.R, sizeof(a.R));
  std::memcpy(h->last_t, a.t, sizeof(a.t));
  HIP_TRY(begin_histogram(h));
  a.hist = h->d_hist;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[1], h->stream));
  if (h->precision == NIDREG_PREC_FP32) {
    HIP_TRY(launch_spline_hist<float>(a));
  } else {
    HIP_TRY(launch_spline_hist<double>(a));
  }
  return NIDREG_OK;
}

int launch_hist_nearest(nidreg_handle* h, const double* T) {
  PassArgs a;
  fill_pass_args(h, a);
  for (int k = 0; k < 12; k++) a.iso[k] = T[k];
  HIP_TRY(begin_histogram(h));
  a.hist = h->d_hist;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[1], h->stream));
  if (h->precision == NIDREG_PREC_FP32) {
    HIP_TRY(launch_nearest_hist<float>(a));
  } else {
    HIP_TRY(launch_nearest_hist<double>(a));
  }
  return NIDREG_OK;
}

int launch_entropy(nidreg_handle* h, double tag) {
  const double inv_unit = std::ldexp(1.0, -h->frac_bits);
  hipLaunchKernelGGL(
    k_entropy, dim3(h->NEB), dim3(kThreads), 0, h->stream, h->d_hist, h->bins, kEntropyCols, inv_unit, h->d_part_hj, h->d_row_part, h->d_phi_q, h->d_hist_image,
    h->d_hist_points, h->d_scal, h->d_out, h->d_out_host, tag, h->d_counters, h->own_hist ? h->d_hist_buf[h->hist_cur ^ 1] : nullptr, h->hist_words);
  HIP_TRY(hipGetLastError());
  if (h->own_hist) h->hist_zeroed[h->hist_cur ^ 1] = true;  // zeroed by this k_entropy for the next evaluation
  return NIDREG_OK;
}

int launch_grad(nidreg_handle* h) {
  PassArgs a;
  fill_pass_args(h, a);
  // same pose as the histogram pass of this evaluation
  std::memcpy(a.R, h->last_R, sizeof(a.R));
  std::memcpy(a.t, h->last_t, sizeof(a.t));
  if (h->precision == NIDREG_PREC_FP32) {
    HIP_TRY(launch_spline_grad<float>(a));
  } else {
    HIP_TRY(launch_spline_grad<double>(a));
  }
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[4], h->stream));
  if (h->nchunks == 0) {  // empty cloud: no gradient workgroups ran, finalise (zeros) stand-alone
    hipLaunchKernelGGL(k_grad_final, dim3(1), dim3(kThreads), 0, h->stream, h->d_partials, 0, h->last_q[0], h->last_q[1], h->last_q[2], h->last_q[3], h->d_out, h->d_out_host, h->seq);
    HIP_TRY(hipGetLastError());
  }
  return NIDREG_OK;
}

// asynchronous part of nidreg_eval
int eval_launch(nidreg_handle* h, const double* se3, bool want_grad) {
  if (h->mode != NIDREG_MODE_SPLINE) return fail(NIDREG_ERR_INVALID, "nidreg_eval: handle was created in NEAREST mode");
  HIP_TRY(hipSetDevice(h->device));
  h->seq += 1.0;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  int rc = launch_hist_spline(h, se3);
  if (rc) return rc;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[2], h->stream));
  rc = launch_entropy(h, want_grad ? 0.0 : h->seq);
  if (rc) return rc;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[3], h->stream));
  h->ev_grad = want_grad;
  if (want_grad) {
    rc = launch_grad(h);
    if (rc) return rc;
  } else if (h->timing) {
    HIP_TRY(hipEventRecord(h->ev[4], h->stream));
  }
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[5], h->stream));
  if (!h->d_out_host) HIP_TRY(hipMemcpyAsync(h->h_out, h->d_out, NIDREG_OUT_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return NIDREG_OK;
}

int eval_finish(nidreg_handle* h, double* cost, double* grad7) {
  HIP_TRY(hipSetDevice(h->device));
  if (h->d_out_host) {
    // the finalising workgroup wrote the results and then this evaluation's tag into host-mapped memory:
    // poll the tag (a few us cheaper than hipStreamSynchronize); look at the stream now and then so that a
    // faulted kernel cannot hang the caller, and so the runtime can retire finished commands
    volatile double* flag = h->h_out + 15;
    unsigned long long spins = 0;
    while (*flag != h->seq) {
      if ((++spins & 0x3fffull) == 0) {
        const hipError_t q = hipStreamQuery(h->stream);
        if (q == hipSuccess) {
          if (*flag != h->seq) HIP_TRY(hipStreamSynchronize(h->stream));
          if (*flag != h->seq) return fail(NIDREG_ERR_HIP, "nidreg_eval: stream drained but the completion tag is missing");
          break;
        }
        if (q != hipErrorNotReady) return fail(NIDREG_ERR_HIP, std::string("nidreg_eval: ") + hipGetErrorString(q));
      }
    }
    if (++h->evals_since_reap >= 256) {
      h->evals_since_reap = 0;
      (void)hipStreamQuery(h->stream);
    }
  } else {
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  if (cost) *cost = h->h_out[0];
  if (grad7)
    for (int k = 0; k < 7; k++) grad7[k] = h->h_out[1 + k];
  return h->h_out[8] != 0.0 ? NIDREG_FALSE : NIDREG_OK;
}

int iso_launch(nidreg_handle* h, const double* T) {
  if (h->mode != NIDREG_MODE_NEAREST) return fail(NIDREG_ERR_INVALID, "nidreg_eval_iso: handle was created in SPLINE mode");
  HIP_TRY(hipSetDevice(h->device));
  h->seq += 1.0;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  int rc = launch_hist_nearest(h, T);
  if (rc) return rc;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[2], h->stream));
  rc = launch_entropy(h, h->seq);
  if (rc) return rc;
  if (h->timing) {
    HIP_TRY(hipEventRecord(h->ev[3], h->stream));
    HIP_TRY(hipEventRecord(h->ev[4], h->stream));
    HIP_TRY(hipEventRecord(h->ev[5], h->stream));
  }
  h->ev_grad = false;
  if (!h->d_out_host) HIP_TRY(hipMemcpyAsync(h->h_out, h->d_out, NIDREG_OUT_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return NIDREG_OK;
}

// visual_camera_calibration.cpp:149-156: delta = init^-1 * T; reject when |t| > 0.2 m or angle > 2 deg
bool trust_gate_ok(const double* init, const double* se3) {
  const double x0 = -init[0], y0 = -init[1], z0 = -init[2], w0 = init[3];
  const double x1 = se3[0], y1 = se3[1], z1 = se3[2], w1 = se3[3];
  const double qw = w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1;
  const double qx = w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1;
  const double qy = w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1;
  const double qz = w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1;
  // R0^T (t - t0) with the unit-quaternion rotation of init
  const double ix = init[0], iy = init[1], iz = init[2], iw = init[3];
  const double R0[9] = {1 - 2 * (iy * iy + iz * iz), 2 * (ix * iy - iz * iw),     2 * (ix * iz + iy * iw),     2 * (ix * iy + iz * iw),    1 - 2 * (ix * ix + iz * iz),
                        2 * (iy * iz - ix * iw),     2 * (ix * iz - iy * iw),     2 * (iy * iz + ix * iw),     1 - 2 * (ix * ix + iy * iy)};
  const double d[3] = {se3[4] - init[4], se3[5] - init[5], se3[6] - init[6]};
  const double tx = R0[0] * d[0] + R0[3] * d[1] + R0[6] * d[2];
  const double ty = R0[1] * d[0] + R0[4] * d[1] + R0[7] * d[2];
  const double tz = R0[2] * d[0] + R0[5] * d[1] + R0[8] * d[2];
  const double tn = std::sqrt(tx * tx + ty * ty + tz * tz);
  const double ang = 2.0 * std::atan2(std::sqrt(qx * qx + qy * qy + qz * qz), std::fabs(qw));
  return !(tn > 0.2 || ang > 2.0 * M_PI / 180.0);
}

}  // namespace

extern "C" {

const char* nidreg_last_error(void) { return g_last_error.c_str(); }
const char* nidreg_version(void) { return "nidreg 0.1 (gfx950, hand-written HIP)"; }

int nidreg_model_from_name(const char* name, int* num_intrinsics, int* num_distortion) {
  if (!name) return -1;
  const std::string s(name);
  int id = -1;
  if (s == "plumb_bob") id = NIDREG_MODEL_PLUMB_BOB;
  else if (s == "fisheye" || s == "equidistant") id = NIDREG_MODEL_FISHEYE;
  else if (s == "atan") id = NIDREG_MODEL_ATAN;
  else if (s == "omnidir") id = NIDREG_MODEL_OMNIDIR;
  else if (s == "equirectangular") id = NIDREG_MODEL_EQUIRECTANGULAR;
  else if (s == "rational_polynomial") id = NIDREG_MODEL_RATIONAL_POLYNOMIAL;
  if (id < 0) return -1;
  if (num_intrinsics) *num_intrinsics = kNumIntr[id];
  if (num_distortion) *num_distortion = kNumDist[id];
  return id;
}

int nidreg_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int64_t nidreg_hist_words(int bins) {
  // room for a partially filled last column group plus the tail words
  return int64_t(bins) * bins + kTailWords + ((bins + 7) & ~7);  // joint histogram, tail, column sums
}

}  // extern "C"

// device-resident cloud: uploaded once per pair, re-culled / re-bucketed on the GPU for every handle
struct nidreg_cloud {
  int device = 0;
  int64_t n = 0;
  double* d_pts = nullptr;  // n x 4 doubles (x y z 1)
  double* d_int = nullptr;  // n doubles
};

namespace {

int create_impl(const nidreg_desc* d, const nidreg_cloud* cloud, const double* T_cull, double min_z, int enable_depth, nidreg_handle** out) {
  if (!d || !out) return fail(NIDREG_ERR_INVALID, "nidreg_create: null argument");
  *out = nullptr;
  if (d->struct_size != int32_t(sizeof(nidreg_desc))) return fail(NIDREG_ERR_INVALID, "nidreg_create: struct_size mismatch");
  if (d->model_id < 0 || d->model_id > 5) return fail(NIDREG_ERR_INVALID, "nidreg_create: unknown camera model");
  if (d->bins < 2 || d->bins > NIDREG_MAX_BINS) return fail(NIDREG_ERR_INVALID, "nidreg_create: bins must be in [2, 256]");
  if (d->width < 1 || d->height < 1 || !d->image) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad image");
  const int64_t n_in = cloud ? cloud->n : d->num_points;
  if (n_in < 0 || n_in > int64_t(INT_MAX)) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad num_points");
  if (!cloud && n_in > 0 && (!d->points || !d->intensities)) return fail(NIDREG_ERR_INVALID, "nidreg_create: null points");
  if (cloud && cloud->device != d->device_id) return fail(NIDREG_ERR_INVALID, "nidreg_create_from_cloud: cloud lives on another device");
  if (d->mode != NIDREG_MODE_SPLINE && d->mode != NIDREG_MODE_NEAREST) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad mode");
  if (d->precision != NIDREG_PREC_FP64 && d->precision != NIDREG_PREC_FP32) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad precision");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NIDREG_ERR_NO_DEVICE, "nidreg_create: no HIP device (the NID core has no CPU path)");
  if (d->device_id < 0 || d->device_id >= ndev) return fail(NIDREG_ERR_INVALID, "nidreg_create: device_id out of range");
  HIP_TRY(hipSetDevice(d->device_id));

  nidreg_handle* h = new nidreg_handle();
  h->device = d->device_id;
  h->model = d->model_id;
  h->mode = d->mode;
  h->precision = d->precision;
  h->bins = d->bins;
  h->W = d->width;
  h->H = d->height;
  h->num_points = n_in;
  h->max_fov = d->max_fov;
  std::memcpy(h->intr, d->intrinsics, sizeof(h->intr));
  std::memcpy(h->dist, d->distortion, sizeof(h->dist));
  const int B = h->bins;
  int64_t N = h->num_points;  // becomes the number of records (after culling on the cloud path)

#define CREATE_TRY(expr)                                                                     \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      free_handle(h);                                                                        \
      return fail(NIDREG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    }                                                                                        \
  } while (0)

  // ---- tiling: a workgroup owns GW histogram columns (= GW * B cells) in LDS, each cell replicated
  // 2^cshift times (lane-private copies, see k_spline_hist).  Default: ~256 cells x 16 copies = 32 KB.
  int GW = d->columns_per_group > 0 ? d->columns_per_group : std::max(1, 256 / B);
  GW = std::min(GW, B);
  int copies = d->lds_copies > 0 ? d->lds_copies : 16;
  int cshift = 0;
  while ((2 << cshift) <= copies && cshift < 4) cshift++;
  while (size_t(GW) * B * 8 > 128 * 1024 && GW > 1) GW /= 2;
  while ((size_t(GW) * B * 8 << cshift) > 64 * 1024 && cshift > 0) cshift--;
  // the headline shape (256 bins, one column per workgroup, default tuning) takes the WIDE histogram
  // kernel: 512 threads, 32 copies, one-instruction tap address (k_spline_hist); an explicit
  // lds_copies keeps the generic kernel (tests compare the two bit for bit)
  h->wide = (d->mode == NIDREG_MODE_SPLINE && B == 256 && GW == 1 && d->lds_copies == 0) ? 1 : 0;
  if (h->wide) cshift = kWideShift;
  h->GW = GW;
  h->cshift = cshift;
  h->NG = (B + GW - 1) / GW;
  h->NEB = (B + kEntropyCols - 1) / kEntropyCols;
  h->lds_hist = (size_t(GW) * B * 8 << cshift) + size_t(GW) * 8 + 16;
  // gradient pass: a single-column workgroup (GW = 1) keeps ONE copy of its G column (k_spline_grad<.., GW1>)
  h->lds_grad = (GW == 1 ? size_t(B) * 8 : (size_t(GW) * B * 8 << cshift)) + size_t(kWaves) * 12 * 8 + 16;
  h->lds_entropy = size_t(B) * 8 + size_t(GW) * 8 + size_t(kWaves) * 8;

  // ---- fixed point: sum over a bin <= N * 2^frac must stay below 2^63
  const int64_t scaleN = std::max<int64_t>(N, d->scale_points);
  int nbits = 1;
  while ((int64_t(1) << nbits) <= scaleN) nbits++;
  h->frac_bits = d->mode == NIDREG_MODE_NEAREST ? 0 : std::min(40, 62 - nbits);

  // ---- bin image, padded by 1 (left/top) and >= 2 (right/bottom), edge replicated:
  // bin_image = min(int(pix * bins), bins - 1) (nid_cost.hpp:78-79) for CV_64FC1 input;
  // max(0, min(bins-1, int(u8 / 255.0 * bins))) (cost_calculator_nid.cpp:43-46) for CV_8UC1 input.
  const int W = h->W, H = h->H;
  h->pitch = ((W + 8) + 3) & ~3;  // padded width in pixels
  const int PH = H + 3;
  const int nstrips = (PH + 3) / 4 + 1;  // rows are stored in strips of four (nid_device.hpp load_patch)
  std::vector<uint8_t> img(size_t(h->pitch) * 4 * nstrips + 64, 0);
  {
    uint8_t lut[256];
    for (int k = 0; k < 256; k++) lut[k] = uint8_t(std::max(0, std::min(B - 1, cast_int(k / 255.0 * B))));
    const uint8_t* base = static_cast<const uint8_t*>(d->image);
    for (int py = 0; py < nstrips * 4; py++) {
      const int sy = std::min(std::max(py - 1, 0), H - 1);
      uint8_t* dst = img.data() + size_t(py >> 2) * size_t(h->pitch) * 4 + size_t(py & 3);
      if (d->image_dtype == NIDREG_IMAGE_F64) {
        const double* row = reinterpret_cast<const double*>(base + size_t(sy) * d->image_row_stride);
        for (int px = 0; px < h->pitch; px++) {
          const int sx = std::min(std::max(px - 1, 0), W - 1);
          dst[size_t(px) * 4] = uint8_t(std::max(0, std::min(cast_int(row[sx] * B), B - 1)));
        }
      } else {
        const uint8_t* row = base + size_t(sy) * d->image_row_stride;
        for (int px = 0; px < h->pitch; px++) {
          const int sx = std::min(std::max(px - 1, 0), W - 1);
          dst[size_t(px) * 4] = lut[row[sx]];
        }
      }
    }
  }
  CREATE_TRY(hipMalloc(&h->d_img, img.size()));
  CREATE_TRY(hipMemcpy(h->d_img, img.data(), img.size(), hipMemcpyHostToDevice));

  // ---- points: bin_points = max(0, min(bins-1, int(intensity * bins))) (nid_cost.hpp:49,
  // cost_calculator_nid.cpp:47) is pose independent -> bucket by column group (stable), so a
  // workgroup owns GW histogram columns.  Records are float32 when that is lossless or when the
  // caller asked for FP32 geometry; otherwise double.
  std::vector<int64_t> gcount;
  if (cloud) {
    // device path: [ViewCulling::cull ->] bucket -> Morton sort -> gather, all on the GPU (nid_build.hip)
    CullArgs ca;
    if (T_cull) {
      ca.model = d->model_id;
      std::memcpy(ca.intr, d->intrinsics, sizeof(ca.intr));
      std::memcpy(ca.dist, d->distortion, sizeof(ca.dist));
      std::memcpy(ca.T, T_cull, sizeof(ca.T));
      ca.W = d->width;
      ca.H = d->height;
      ca.min_z = min_z;
      ca.depth = enable_depth ? 1 : 0;
    }
    void* recs = nullptr;
    int rec64 = 0;
    CREATE_TRY(build_records_device(cloud->d_pts, cloud->d_int, cloud->n, T_cull ? &ca : nullptr, B, GW, h->NG, d->precision == NIDREG_PREC_FP32, &recs, &rec64, gcount, nullptr));
    h->d_pts = recs;
    h->rec64 = rec64;
    N = gcount[size_t(h->NG)];
    h->num_points = N;
  } else {
    const char* pbase = reinterpret_cast<const char*>(d->points);
    const int64_t pstride = d->point_stride > 0 ? d->point_stride : 32;
    const bool spatial = !(d->flags & NIDREG_FLAG_INPUT_ORDER);
    const int nthreads = int(std::max(1u, std::min(32u, std::thread::hardware_concurrency())));
    auto parallel_for = [&](int64_t n, const std::function<void(int64_t, int64_t, int)>& fn) {
      const int T = int(std::min<int64_t>(nthreads, std::max<int64_t>(1, n / 65536)));
      if (T <= 1) {
        fn(0, n, 0);
        return;
      }
      std::vector<std::thread> th;
      for (int t = 0; t < T; t++) th.emplace_back([&, t]() { fn(n * t / T, n * (t + 1) / T, t); });
      for (auto& x : th) x.join();
    };

    // pass 1 (parallel): histogram column, float-representability, Morton code of the bearing
    std::vector<uint32_t> bin(N), mort(spatial ? N : 0);
    std::vector<int> lossless_t(nthreads, 1);
    std::vector<std::vector<int64_t>> gcount_t(nthreads, std::vector<int64_t>(h->NG, 0));
    parallel_for(N, [&](int64_t lo, int64_t hi, int t) {
      bool ll = true;
      std::vector<int64_t>& gc = gcount_t[t];
      for (int64_t i = lo; i < hi; i++) {
        const double* p = reinterpret_cast<const double*>(pbase + i * pstride);
        if (ll) {
          for (int k = 0; k < 3; k++)
            if (double(float(p[k])) != p[k] && p[k] == p[k]) ll = false;
        }
        const int b = std::max(0, std::min(B - 1, cast_int(d->intensities[i] * B)));
        bin[i] = uint32_t(b);
        gc[b / GW]++;
        if (spatial) {
          // bearing cell: the sums are order independent (fixed point), so any order gives the same
          // bits; a spatially coherent one makes the 64 lanes of a wave gather from neighbouring image
          // lines for ANY pose (camera and LiDAR are rigidly mounted: a compact patch of bearings stays a
          // compact patch of pixels)
          const double az = std::atan2(p[1], p[0]);
          const double el = std::atan2(p[2], std::sqrt(p[0] * p[0] + p[1] * p[1]));
          uint32_t qa = uint32_t(std::min(65535.0, std::max(0.0, (az + M_PI) * (65535.0 / (2.0 * M_PI)))));
          uint32_t qe = uint32_t(std::min(65535.0, std::max(0.0, (el + 0.5 * M_PI) * (65535.0 / M_PI))));
          if (!(az == az) || !(el == el)) qa = qe = 0;
          uint32_t m = 0;
          for (int bb = 0; bb < 16; bb++) m |= (((qa >> bb) & 1u) << (2 * bb)) | (((qe >> bb) & 1u) << (2 * bb + 1));
          mort[i] = m;
        }
      }
      lossless_t[t] = ll ? 1 : 0;
    });
    bool lossless = true;
    for (int t = 0; t < nthreads; t++) lossless = lossless && lossless_t[t];
    gcount.assign(size_t(h->NG) + 1, 0);
    for (int g = 0; g < h->NG; g++) {
      int64_t c = 0;
      for (int t = 0; t < nthreads; t++) c += gcount_t[t][g];
      gcount[g + 1] = gcount[g] + c;
    }
    h->rec64 = (d->precision == NIDREG_PREC_FP64 && !lossless) ? 1 : 0;
    const size_t rec_bytes = h->rec64 ? sizeof(Rec64) : sizeof(Rec32);
    {
      // order[k] = source index of the k-th device record: stable bucketing by column group, then (default)
      // each group sorted by Morton code; NIDREG_FLAG_INPUT_ORDER keeps the caller's order inside groups.
      std::vector<uint32_t> order(N);
      {
        std::vector<int64_t> cursor(gcount.begin(), gcount.end() - 1);
        for (int64_t i = 0; i < N; i++) order[cursor[bin[i] / GW]++] = uint32_t(i);
      }
      if (spatial) {
        std::atomic<int> next_group(0);
        auto worker = [&]() {
          std::vector<std::pair<uint32_t, uint32_t>> tmp;
          for (;;) {
            const int g = next_group.fetch_add(1);
            if (g >= h->NG) break;
            const int64_t lo = gcount[g], hi = gcount[g + 1];
            tmp.resize(size_t(hi - lo));
            for (int64_t k = lo; k < hi; k++) tmp[size_t(k - lo)] = std::make_pair(mort[order[k]], order[k]);
            std::sort(tmp.begin(), tmp.end());
            for (int64_t k = lo; k < hi; k++) order[k] = tmp[size_t(k - lo)].second;
          }
        };
        const int T = N > 200000 ? std::min(nthreads, h->NG) : 1;
        if (T <= 1) {
          worker();
        } else {
          std::vector<std::thread> th;
          for (int t = 0; t < T; t++) th.emplace_back(worker);
          for (auto& x : th) x.join();
        }
      }
      std::vector<unsigned char> recs(size_t(std::max<int64_t>(N, 1)) * rec_bytes);
      parallel_for(N, [&](int64_t lo, int64_t hi, int) {
        for (int64_t dst = lo; dst < hi; dst++) {
          const int64_t i = order[dst];
          const double* p = reinterpret_cast<const double*>(pbase + i * pstride);
          if (h->rec64) {
            Rec64 r;
            r.x = p[0];
            r.y = p[1];
            r.z = p[2];
            r.bin = bin[i];
            std::memcpy(recs.data() + size_t(dst) * rec_bytes, &r, rec_bytes);
          } else {
            Rec32 r;
            r.x = float(p[0]);
            r.y = float(p[1]);
            r.z = float(p[2]);
            r.bin = bin[i];
            std::memcpy(recs.data() + size_t(dst) * rec_bytes, &r, rec_bytes);
          }
        }
      });
      CREATE_TRY(hipMalloc(&h->d_pts, recs.size() + 64));
      CREATE_TRY(hipMemcpy(h->d_pts, recs.data(), recs.size(), hipMemcpyHostToDevice));
    }
  }

  // ---- chunk tables: each chunk = one workgroup, points of one column group only.  By default a pass
  // gets ONE round of co-resident workgroups (measured on cfg 2: the per-workgroup prologue / flush is
  // amortised over more points and no partial last round is left -- 2048 chunks +4 %, 4096 +12 %):
  // 4 workgroups per CU for the 256-thread kernels, 2 per CU for the WIDE histogram kernel (64 KB LDS
  // each), which therefore has its own table.  A column group is split EVENLY into its chunks.
  {
    int num_cus = 256;
    if (hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || num_cus <= 0) num_cus = 256;
    auto build_chunks = [&](int target, int threads, std::vector<Chunk>& chunks) {
      int64_t CH = (N + target - 1) / std::max(target, 1);
      CH = std::max<int64_t>(threads, ((CH + threads - 1) / threads) * threads);
      for (int g = 0; g < h->NG; g++) {
        const int64_t cnt = gcount[g + 1] - gcount[g];
        if (cnt <= 0) continue;
        const int64_t parts = (cnt + CH - 1) / CH;
        const int64_t size = (((cnt + parts - 1) / parts + 63) / 64) * 64;  // 64 records = 1 KB: chunk starts stay aligned
        for (int64_t s = gcount[g]; s < gcount[g + 1]; s += size) {
          Chunk c;
          c.start = uint32_t(s);
          c.count = uint32_t(std::min<int64_t>(size, gcount[g + 1] - s));
          c.group = uint32_t(g);
          c.pad = 0;
          chunks.push_back(c);
        }
      }
    };
    std::vector<Chunk> chunks;
    build_chunks(d->target_blocks > 0 ? d->target_blocks : 4 * num_cus, kThreads, chunks);
    h->nchunks = int(chunks.size());
    CREATE_TRY(hipMalloc(&h->d_chunks, std::max<size_t>(chunks.size(), 1) * sizeof(Chunk)));
    if (!chunks.empty()) CREATE_TRY(hipMemcpy(h->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
    if (h->wide) {
      std::vector<Chunk> wide_chunks;
      build_chunks(d->target_blocks > 0 ? d->target_blocks : 2 * num_cus, kWideThreads, wide_chunks);
      h->nchunks_hist = int(wide_chunks.size());
      CREATE_TRY(hipMalloc(&h->d_chunks_hist, std::max<size_t>(wide_chunks.size(), 1) * sizeof(Chunk)));
      if (!wide_chunks.empty()) CREATE_TRY(hipMemcpy(h->d_chunks_hist, wide_chunks.data(), wide_chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
    }
  }

  // ---- per-evaluation scratch
  h->hist_words = nidreg_hist_words(B);
  if (d->ext_stream || (d->flags & NIDREG_FLAG_EXT_STREAM)) {
    h->stream = static_cast<hipStream_t>(d->ext_stream);
  } else {
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  if (d->ext_hist) {
    h->d_hist = static_cast<u64*>(d->ext_hist);
  } else {
    CREATE_TRY(hipMalloc(&h->d_hist_buf[0], size_t(h->hist_words) * sizeof(u64)));
    CREATE_TRY(hipMalloc(&h->d_hist_buf[1], size_t(h->hist_words) * sizeof(u64)));
    CREATE_TRY(hipMemset(h->d_hist_buf[1], 0, size_t(h->hist_words) * sizeof(u64)));
    h->d_hist = h->d_hist_buf[0];
    h->hist_cur = 0;
    h->hist_zeroed[1] = true;  // [0] is zeroed below and read by nidreg_get_hist before the first evaluation
    h->own_hist = true;
  }
  if (d->ext_out) {
    h->d_out = static_cast<double*>(d->ext_out);
  } else {
    CREATE_TRY(hipMalloc(&h->d_out, NIDREG_OUT_DOUBLES * sizeof(double)));
    h->own_out = true;
  }
  CREATE_TRY(hipMemset(h->d_out, 0, NIDREG_OUT_DOUBLES * sizeof(double)));
  CREATE_TRY(hipMemset(h->d_hist, 0, size_t(h->hist_words) * sizeof(u64)));
  CREATE_TRY(hipMalloc(&h->d_part_hj, size_t(h->NEB) * sizeof(double)));
  CREATE_TRY(hipMalloc(&h->d_row_part, size_t(h->NEB) * B * sizeof(u64)));
  CREATE_TRY(hipMalloc(&h->d_phi_q, size_t(B) * sizeof(double)));
  CREATE_TRY(hipMalloc(&h->d_hist_image, size_t(B) * sizeof(double)));
  CREATE_TRY(hipMalloc(&h->d_hist_points, size_t(B) * sizeof(double)));
  CREATE_TRY(hipMalloc(&h->d_scal, sizeof(EntropyScalars)));
  CREATE_TRY(hipMalloc(&h->d_partials, std::max<size_t>(h->nchunks, 1) * 12 * sizeof(double)));
  CREATE_TRY(hipHostMalloc(&h->h_out, NIDREG_OUT_DOUBLES * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
  std::memset(h->h_out, 0, NIDREG_OUT_DOUBLES * sizeof(double));
  if (!d->ext_out) {
    // results are written straight into host-mapped memory by the finalising workgroups: no D2H copy
    void* dp = nullptr;
    CREATE_TRY(hipHostGetDevicePointer(&dp, h->h_out, 0));
    h->d_out_host = static_cast<double*>(dp);
  }
  CREATE_TRY(hipMalloc(&h->d_counters, 4 * sizeof(unsigned int)));
  CREATE_TRY(hipMemset(h->d_counters, 0, 4 * sizeof(unsigned int)));
  for (int i = 0; i < 6; i++) CREATE_TRY(hipEventCreate(&h->ev[i]));
#undef CREATE_TRY
  *out = h;
  return NIDREG_OK;
}

}  // namespace

extern "C" {

int nidreg_create(const nidreg_desc* d, nidreg_handle** out) { return create_impl(d, nullptr, nullptr, 0.0, 0, out); }

int nidreg_cloud_create(int device_id, const double* points, int64_t point_stride, const double* intensities, int64_t num_points, nidreg_cloud** out) {
  if (!out || num_points < 0 || num_points > int64_t(INT_MAX) || (num_points > 0 && (!points || !intensities))) return fail(NIDREG_ERR_INVALID, "nidreg_cloud_create: bad argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NIDREG_ERR_NO_DEVICE, "nidreg_cloud_create: no HIP device");
  if (device_id < 0 || device_id >= ndev) return fail(NIDREG_ERR_INVALID, "nidreg_cloud_create: device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  nidreg_cloud* c = new nidreg_cloud();
  c->device = device_id;
  c->n = num_points;
  const size_t n1 = size_t(std::max<int64_t>(num_points, 1));
  hipError_t e = hipMalloc(&c->d_pts, n1 * 32);
  if (e == hipSuccess) e = hipMalloc(&c->d_int, n1 * 8);
  const int64_t stride = point_stride > 0 ? point_stride : 32;
  if (e == hipSuccess && num_points > 0) {
    if (stride == 32) {
      e = hipMemcpy(c->d_pts, points, size_t(num_points) * 32, hipMemcpyHostToDevice);
    } else {
      e = hipMemcpy2D(c->d_pts, 32, points, size_t(stride), 32, size_t(num_points), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMemcpy(c->d_int, intensities, size_t(num_points) * 8, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    if (c->d_pts) (void)hipFree(c->d_pts);
    if (c->d_int) (void)hipFree(c->d_int);
    delete c;
    return fail(NIDREG_ERR_HIP, std::string("nidreg_cloud_create: ") + hipGetErrorString(e));
  }
  *out = c;
  return NIDREG_OK;
}

void nidreg_cloud_destroy(nidreg_cloud* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->d_pts) (void)hipFree(c->d_pts);
  if (c->d_int) (void)hipFree(c->d_int);
  delete c;
}

int nidreg_create_from_cloud(const nidreg_desc* d, const nidreg_cloud* cloud, const double* T_camera_lidar, double min_z, int enable_depth_buffer_culling, nidreg_handle** out) {
  if (!cloud) return fail(NIDREG_ERR_INVALID, "nidreg_create_from_cloud: null cloud");
  return create_impl(d, cloud, T_camera_lidar, min_z, enable_depth_buffer_culling, out);
}

void nidreg_destroy(nidreg_handle* h) { free_handle(h); }

int nidreg_eval(nidreg_handle* h, const double* se3, double* cost, double* grad7) {
  if (!h || !se3) return fail(NIDREG_ERR_INVALID, "nidreg_eval: null argument");
  const int rc = eval_launch(h, se3, grad7 != nullptr);
  if (rc) return rc;
  return eval_finish(h, cost, grad7);
}

int nidreg_eval_iso(nidreg_handle* h, const double* T, double* cost) {
  if (!h || !T) return fail(NIDREG_ERR_INVALID, "nidreg_eval_iso: null argument");
  const int rc = iso_launch(h, T);
  if (rc) return rc;
  return eval_finish(h, cost, nullptr) < 0 ? NIDREG_ERR_HIP : NIDREG_OK;  // CostCalculatorNID has no finite check
}

int nidreg_eval_multi(nidreg_handle* const* handles, int n, const double* init_se3, const double* se3, double* cost, double* grad7) {
  if (!handles || n <= 0 || !se3) return fail(NIDREG_ERR_INVALID, "nidreg_eval_multi: bad argument");
  if (init_se3 && !trust_gate_ok(init_se3, se3)) return NIDREG_FALSE;
  for (int i = 0; i < n; i++) {
    const int rc = eval_launch(handles[i], se3, grad7 != nullptr);
    if (rc) return rc;
  }
  double csum = 0.0, gsum[7] = {0, 0, 0, 0, 0, 0, 0};
  bool all_ok = true;
  for (int i = 0; i < n; i++) {
    double c = 0.0, g[7];
    const int rc = eval_finish(handles[i], &c, grad7 ? g : nullptr);
    if (rc < 0) return rc;
    if (rc == NIDREG_FALSE) all_ok = false;
    csum += c;
    if (grad7)
      for (int k = 0; k < 7; k++) gsum[k] += g[k];
  }
  if (cost) *cost = csum;
  if (grad7)
    for (int k = 0; k < 7; k++) grad7[k] = gsum[k];
  return all_ok ? NIDREG_OK : NIDREG_FALSE;
}

int nidreg_eval_iso_multi(nidreg_handle* const* handles, int n, const double* T, double* cost) {
  if (!handles || n <= 0 || !T) return fail(NIDREG_ERR_INVALID, "nidreg_eval_iso_multi: bad argument");
  for (int i = 0; i < n; i++) {
    const int rc = iso_launch(handles[i], T);
    if (rc) return rc;
  }
  double csum = 0.0;
  for (int i = 0; i < n; i++) {
    double c = 0.0;
    const int rc = eval_finish(handles[i], &c, nullptr);
    if (rc < 0) return rc;
    csum += c;
  }
  if (cost) *cost = csum;
  return NIDREG_OK;
}

int nidreg_get_hist_fixed(nidreg_handle* h, int64_t* joint, int64_t* inliers, int* frac_bits) {
  if (!h) return fail(NIDREG_ERR_INVALID, "nidreg_get_hist_fixed: null handle");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int B = h->bins;
  std::vector<u64> tmp(size_t(h->hist_words));
  HIP_TRY(hipMemcpy(tmp.data(), h->d_hist, tmp.size() * sizeof(u64), hipMemcpyDeviceToHost));
  if (joint) {
    // device layout [bin_points][bin_image] -> [bin_image][bin_points]
    for (int c = 0; c < B; c++)
      for (int r = 0; r < B; r++) joint[size_t(r) * B + c] = int64_t(tmp[size_t(c) * B + r]);
  }
  if (inliers) *inliers = int64_t(tmp[size_t(B) * B + kTailInliers]);
  if (frac_bits) *frac_bits = h->frac_bits;
  return NIDREG_OK;
}

int nidreg_get_hist(nidreg_handle* h, double* joint, double* hist_image, double* hist_points) {
  if (!h) return fail(NIDREG_ERR_INVALID, "nidreg_get_hist: null handle");
  const int B = h->bins;
  if (joint) {
    std::vector<int64_t> fx(size_t(B) * B);
    const int rc = nidreg_get_hist_fixed(h, fx.data(), nullptr, nullptr);
    if (rc) return rc;
    const double inv_unit = std::ldexp(1.0, -h->frac_bits);
    for (size_t k = 0; k < fx.size(); k++) joint[k] = double(fx[k]) * inv_unit;
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (hist_image) HIP_TRY(hipMemcpy(hist_image, h->d_hist_image, size_t(B) * sizeof(double), hipMemcpyDeviceToHost));
  if (hist_points) HIP_TRY(hipMemcpy(hist_points, h->d_hist_points, size_t(B) * sizeof(double), hipMemcpyDeviceToHost));
  return NIDREG_OK;
}

int nidreg_project(nidreg_handle* h, const double* p3, int64_t n, double* uv, double* jac) {
  if (!h || !p3 || !uv || n < 0) return fail(NIDREG_ERR_INVALID, "nidreg_project: bad argument");
  if (n == 0) return NIDREG_OK;
  HIP_TRY(hipSetDevice(h->device));
  double *d_p = nullptr, *d_uv = nullptr, *d_j = nullptr;
  hipError_t e = hipMalloc(&d_p, size_t(n) * 3 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&d_uv, size_t(n) * 2 * sizeof(double));
  if (e == hipSuccess && jac) e = hipMalloc(&d_j, size_t(n) * 6 * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(d_p, p3, size_t(n) * 3 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = h->precision == NIDREG_PREC_FP32 ? launch_project<float>(h->model, h->intr, h->dist, d_p, n, d_uv, d_j, h->stream) : launch_project<double>(h->model, h->intr, h->dist, d_p, n, d_uv, d_j, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(uv, d_uv, size_t(n) * 2 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && jac) e = hipMemcpy(jac, d_j, size_t(n) * 6 * sizeof(double), hipMemcpyDeviceToHost);
  if (d_p) (void)hipFree(d_p);
  if (d_uv) (void)hipFree(d_uv);
  if (d_j) (void)hipFree(d_j);
  if (e != hipSuccess) return fail(NIDREG_ERR_HIP, std::string("nidreg_project: ") + hipGetErrorString(e));
  return NIDREG_OK;
}

int nidreg_project_model(int model_id, const double* intrinsics, const double* distortion, int device_id, int precision, const double* p3, int64_t n, double* uv, double* jac) {
  if (model_id < 0 || model_id > 5 || !intrinsics || !distortion || !p3 || !uv || n < 0) return fail(NIDREG_ERR_INVALID, "nidreg_project_model: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NIDREG_ERR_NO_DEVICE, "nidreg_project_model: no HIP device");
  nidreg_handle tmp;
  tmp.device = device_id;
  tmp.model = model_id;
  tmp.precision = precision;
  std::memcpy(tmp.intr, intrinsics, sizeof(tmp.intr));
  std::memcpy(tmp.dist, distortion, sizeof(tmp.dist));
  tmp.stream = nullptr;  // default stream
  return nidreg_project(&tmp, p3, n, uv, jac);
}

int64_t nidreg_view_culling(int model_id, const double* intrinsics, const double* distortion, int device_id, int width, int height, double min_z, int enable_depth_buffer_culling,
                            const double* points, int64_t point_stride, int64_t num_points, const double* T_camera_lidar, int32_t* indices_out) {
  if (model_id < 0 || model_id > 5 || !intrinsics || !distortion || width < 1 || height < 1 || num_points < 0 || !T_camera_lidar || (num_points > 0 && (!points || !indices_out)))
    return fail(NIDREG_ERR_INVALID, "nidreg_view_culling: bad argument");
  if (num_points == 0) return 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NIDREG_ERR_NO_DEVICE, "nidreg_view_culling: no HIP device");
  HIP_TRY(hipSetDevice(device_id));
  const int64_t stride = point_stride > 0 ? point_stride : 32;
  if (stride % 8 != 0) return fail(NIDREG_ERR_INVALID, "nidreg_view_culling: point_stride must be a multiple of 8");
  double* d_pts = nullptr;
  int* d_pix = nullptr;
  unsigned int* d_zbuf = nullptr;
  unsigned char* d_keep = nullptr;
  std::vector<unsigned char> keep(static_cast<size_t>(num_points));
  hipError_t e = hipMalloc(&d_pts, size_t(num_points) * size_t(stride));
  if (e == hipSuccess) e = hipMalloc(&d_pix, size_t(num_points) * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&d_zbuf, size_t(width) * height * sizeof(unsigned int));
  if (e == hipSuccess) e = hipMalloc(&d_keep, size_t(num_points));
  if (e == hipSuccess) e = hipMemcpy(d_pts, points, size_t(num_points) * size_t(stride), hipMemcpyHostToDevice);
  // CV_32FC1 filled with saturate_cast<float>(DBL_MAX) = +inf (view_culling.cpp:40) = 0x7f800000
  if (e == hipSuccess) e = hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(d_zbuf), 0x7f800000, size_t(width) * height);
  if (e == hipSuccess)
    e = launch_cull(model_id, intrinsics, distortion, d_pts, stride / 8, num_points, T_camera_lidar, width, height, min_z, enable_depth_buffer_culling ? 1 : 0, d_pix, d_zbuf, d_keep, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(keep.data(), d_keep, size_t(num_points), hipMemcpyDeviceToHost);
  if (d_pts) (void)hipFree(d_pts);
  if (d_pix) (void)hipFree(d_pix);
  if (d_zbuf) (void)hipFree(d_zbuf);
  if (d_keep) (void)hipFree(d_keep);
  if (e != hipSuccess) return fail(NIDREG_ERR_HIP, std::string("nidreg_view_culling: ") + hipGetErrorString(e));
  int64_t m = 0;
  for (int64_t i = 0; i < num_points; i++)
    if (keep[size_t(i)]) indices_out[m++] = int32_t(i);
  return m;
}

int nidreg_shard_hist(nidreg_handle* h, const double* se3) {
  if (!h || !se3) return fail(NIDREG_ERR_INVALID, "nidreg_shard_hist: null argument");
  if (h->mode != NIDREG_MODE_SPLINE) return fail(NIDREG_ERR_INVALID, "nidreg_shard_hist: SPLINE handles only");
  HIP_TRY(hipSetDevice(h->device));
  return launch_hist_spline(h, se3);
}

int nidreg_shard_entropy(nidreg_handle* h) {
  if (!h) return fail(NIDREG_ERR_INVALID, "nidreg_shard_entropy: null handle");
  HIP_TRY(hipSetDevice(h->device));
  return launch_entropy(h, 0.0);
}

int nidreg_shard_grad(nidreg_handle* h) {
  if (!h) return fail(NIDREG_ERR_INVALID, "nidreg_shard_grad: null handle");
  HIP_TRY(hipSetDevice(h->device));
  return launch_grad(h);
}

int nidreg_shard_finish(nidreg_handle* h, double* cost, double* grad7) {
  if (!h) return fail(NIDREG_ERR_INVALID, "nidreg_shard_finish: null handle");
  HIP_TRY(hipSetDevice(h->device));
  if (!h->d_out_host) HIP_TRY(hipMemcpyAsync(h->h_out, h->d_out, NIDREG_OUT_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (cost) *cost = h->h_out[0];
  if (grad7)
    for (int k = 0; k < 7; k++) grad7[k] = h->h_out[1 + k];
  return h->h_out[8] != 0.0 ? NIDREG_FALSE : NIDREG_OK;
}

int nidreg_set_timing(nidreg_handle* h, int enable) {
  if (!h) return fail(NIDREG_ERR_INVALID, "nidreg_set_timing: null handle");
  h->timing = enable != 0;
  return NIDREG_OK;
}

int nidreg_get_timing(nidreg_handle* h, float* ms6) {
  if (!h || !ms6) return fail(NIDREG_ERR_INVALID, "nidreg_get_timing: null argument");
  if (!h->timing) return fail(NIDREG_ERR_INVALID, "nidreg_get_timing: timing not enabled");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipEventSynchronize(h->ev[5]));
  HIP_TRY(hipEventElapsedTime(&ms6[0], h->ev[0], h->ev[5]));
  for (int k = 0; k < 5; k++) HIP_TRY(hipEventElapsedTime(&ms6[1 + k], h->ev[k], h->ev[k + 1]));
  return NIDREG_OK;
}

int nidreg_get_info(nidreg_handle* h, int64_t* info8) {
  if (!h || !info8) return fail(NIDREG_ERR_INVALID, "nidreg_get_info: null argument");
  info8[0] = h->rec64 ? int64_t(sizeof(Rec64)) : int64_t(sizeof(Rec32));
  info8[1] = h->nchunks;
  info8[2] = h->GW;
  info8[3] = h->frac_bits;
  info8[4] = int64_t(h->lds_hist);
  info8[5] = h->pitch;
  info8[6] = h->num_points;
  info8[7] = (h->rec64 ? 0 : 1) | (int64_t(1 << h->cshift) << 8);
  return NIDREG_OK;
}

}  // extern "C"
