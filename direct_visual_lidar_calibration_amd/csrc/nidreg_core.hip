#define NID_COMMON_KERNELS
#define NID_FINAL_KERNEL
#include "nidreg_internal.hpp"

thread_local std::string g_last_error;
namespace nidreg {
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
}  // namespace nidreg

namespace nidreg_detail {
std::atomic<int> g_inflight[NIDREG_MAX_DEVICES];  // evaluations in flight per device (InflightGuard below)
ResourcePool g_pool[NIDREG_MAX_DEVICES];

hipError_t pool_stream(int device, hipStream_t* out) {
  if (device >= 0 && device < NIDREG_MAX_DEVICES) {
    std::lock_guard<std::mutex> lk(g_pool[device].mu);
    if (!g_pool[device].streams.empty()) {
      *out = g_pool[device].streams.back();
      g_pool[device].streams.pop_back();
      return hipSuccess;
    }
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
// (the stream must be idle)
void unpool_stream(int device, hipStream_t s) {
  if (!s) return;
  if (device >= 0 && device < NIDREG_MAX_DEVICES) {
    std::lock_guard<std::mutex> lk(g_pool[device].mu);
    if (g_pool[device].streams.size() < kPoolCap) {
      g_pool[device].streams.push_back(s);
      return;
    }
  }
  (void)hipStreamDestroy(s);
}
hipError_t pool_host_block(int device, bool ring, size_t bytes, void** out) {
  if (device >= 0 && device < NIDREG_MAX_DEVICES) {
    std::lock_guard<std::mutex> lk(g_pool[device].mu);
    std::vector<void*>& v = ring ? g_pool[device].ring_blocks : g_pool[device].out_blocks;
    if (!v.empty()) {
      *out = v.back();
      v.pop_back();
      return hipSuccess;
    }
  }
  return hipHostMalloc(out, bytes, hipHostMallocMapped | hipHostMallocCoherent);
}
// (no kernel that writes the block may still be running)
void unpool_host_block(int device, bool ring, void* p) {
  if (!p) return;
  if (device >= 0 && device < NIDREG_MAX_DEVICES) {
    std::lock_guard<std::mutex> lk(g_pool[device].mu);
    std::vector<void*>& v = ring ? g_pool[device].ring_blocks : g_pool[device].out_blocks;
    if (v.size() < kPoolCap) {
      v.push_back(p);
      return;
    }
  }
  (void)hipHostFree(p);
}
SmallProject g_small_project[NIDREG_MAX_DEVICES];

void pool_release(int device) {
  std::vector<hipStream_t> streams;
  std::vector<void*> blocks;
  {
    std::lock_guard<std::mutex> lk(g_pool[device].mu);
    streams.swap(g_pool[device].streams);
    blocks.swap(g_pool[device].out_blocks);
    blocks.insert(blocks.end(), g_pool[device].ring_blocks.begin(), g_pool[device].ring_blocks.end());
    g_pool[device].ring_blocks.clear();
  }
  for (hipStream_t st : streams) (void)hipStreamDestroy(st);
  for (void* b : blocks) (void)hipHostFree(b);
  SmallProject& sp = g_small_project[device];
  std::lock_guard<std::mutex> lk(sp.mu);
  if (sp.host) (void)hipHostFree(sp.host);
  sp.host = sp.dev = nullptr;
}

void free_handle(nidreg_handle* h) {
  if (!h) return;
  if (h->set) {
    free_shard_set(h->set);  // stops the workers and frees the other shards
    h->set = nullptr;
  }
  drop_groups_of(h);
  cohort_leave(h);
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  rccl_release(h);
  for (auto& p : h->pending)  // tickets never collected: give their in-flight counts back to the device
    if (p.ticket != 0 && p.counted && h->device >= 0 && h->device < NIDREG_MAX_DEVICES) g_inflight[h->device].fetch_sub(1, std::memory_order_acq_rel);
  if (h->d_pts) (void)hipFree(h->d_pts);
  if (h->d_chunks) (void)hipFree(h->d_chunks);
  if (h->d_chunks_hist) (void)hipFree(h->d_chunks_hist);
  if (h->d_fused_scratch) (void)hipFree(h->d_fused_scratch);
  if (h->d_eq_tab) (void)hipFree(h->d_eq_tab);
  if (h->d_gend) (void)hipFree(h->d_gend);
  if (h->d_img) (void)hipFree(h->d_img);
  if (h->own_hist) {
    if (h->d_hist_buf[0]) (void)hipFree(h->d_hist_buf[0]);
    if (h->d_hist_buf[1]) (void)hipFree(h->d_hist_buf[1]);
  }
  if (h->d_shard_tab) (void)hipFree(h->d_shard_tab);
  if (h->own_out && h->d_out) (void)hipFree(h->d_out);
  if (h->d_scratch) (void)hipFree(h->d_scratch);
  // (a multi-pair group that evaluated this handle on its own stream was drained and freed by drop_groups_of above)
  unpool_host_block(h->device, false, h->h_out);
  unpool_host_block(h->device, true, h->h_ring);
  for (int i = 0; i < 6; i++)
    if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
  if (h->own_stream && h->stream) unpool_stream(h->device, h->stream);
  delete h;
}

void fill_pass_args(const nidreg_handle* h, PassArgs& a) {
  std::memset(&a, 0, sizeof(a));
  a.model = h->model;
  a.rec64 = h->rec64;
  a.pts = h->d_pts;
  a.chunks = h->d_chunks;
  a.nchunks = h->nchunks;
  a.nslots = h->nslots;
  a.seg = h->seg;
  a.gend = h->d_gend;
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
  a.magic = fixed_unit_k(h);  // the x-weight constants (bspline_scale)
  a.inv_unit = 1.0 / fixed_unit(h);
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
  a.gt_phi_q = h->d_phi_q;
  a.gt_hist_image = h->d_hist_image;
  a.gt_hist_points = h->d_hist_points;
  a.gt_scal = h->d_scal;
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
hipError_t begin_histogram(nidreg_handle* h, hipStream_t stream) {
  if (h->own_hist) {
    h->hist_cur ^= 1;
    h->d_hist = h->d_hist_buf[h->hist_cur];
    // The buffer was cleared by the PREVIOUS evaluation's kernels (plain stores of k_entropy / the gradient prologue), and the
    // host may be here before that kernel has ended: it proceeds on the completion tag.  On the same stream the kernel boundary
    // orders the clears against this evaluation's atomics; on another stream (a cohort round on a different group's stream, a
    // caller mixing nidreg_eval and nidreg_eval_multi on one handle) nothing does -- the clears sit in the old kernel's XCD-local
    // L2 until it ends and could land on top of the new counts.  Rare path: drain the old stream first.
    if (h->hist_zeroed[h->hist_cur] && h->zero_stream && h->zero_stream != stream) {
      hipError_t e = hipStreamSynchronize(h->zero_stream);
      if (e != hipSuccess) return e;
    }
    if (!h->hist_zeroed[h->hist_cur]) {
      hipError_t e = hipMemsetAsync(h->d_hist, 0, size_t(h->hist_words) * sizeof(u64), stream);
      if (e != hipSuccess) return e;
    }
    h->hist_zeroed[h->hist_cur] = false;  // about to be written
    return hipSuccess;
  }
  return hipMemsetAsync(h->d_hist, 0, size_t(h->hist_words) * sizeof(u64), stream);
}
hipError_t begin_histogram(nidreg_handle* h) { return begin_histogram(h, h->stream); }

int launch_hist_spline(nidreg_handle* h, const double* se3, bool alone) {
  PassArgs a;
  fill_pass_args(h, a);
  a.prio = alone ? 1 : 0;
  if (h->d_chunks_hist) {
    a.chunks = h->d_chunks_hist;
    a.nchunks = h->nchunks_hist;
    a.seg = h->seg_hist;
  }
  pose_from_se3(se3, a.R, a.t);
  for (int k = 0; k < 4; k++) h->last_q[k] = se3[k];
  std::memcpy(h->last_R, a.R, sizeof(a.R));
  std::memcpy(h->last_t, a.t, sizeof(a.t));
  HIP_TRY(begin_histogram(h));
  a.hist = h->d_hist;
  if (h->timing == 1) HIP_TRY(hipEventRecord(h->ev[1], h->stream));
  HIP_TRY(launch_spline_hist<double>(a));
  return NIDREG_OK;
}

// k_nearest_hist's fast decision tier (nid_kernels.hpp NearestFast): the coefficients of its error bound for this pose and
// this camera (derivations at the kernel).  plumb_bob and omnidir: only for a FoV cone over which the normalised image
// coordinates stay bounded (tan(max_fov), resp. sin / (cos + xi)); fisheye and equirectangular: any cone (their bands are
// per point); atan and rational_polynomial keep the exact tier.  NIDREG_NEAREST_EXACT=1 switches the tier off (A/B runs).
NearestFastArgs nearest_fast_args(const nidreg_handle* h, const double* T) {
  NearestFastArgs f;
  std::memset(&f, 0, sizeof(f));
  static const bool off = [] {
    const char* e = std::getenv("NIDREG_NEAREST_EXACT");
    return e && *e && *e != '0';
  }();
  if (off || h->nearest_exact || h->precision != NIDREG_PREC_FP64) return f;
  const double eps = std::ldexp(1.0, -52);
  const double pi = 3.14159265358979323846;
  double rmax = 0.0, tmax = 0.0;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) rmax = std::max(rmax, std::fabs(T[4 * r + c]));
    tmax = std::max(tmax, std::fabs(T[4 * r + 3]));
  }
  f.er = 8.0 * eps * rmax;
  f.et = 8.0 * eps * tmax;
  const double cos_fov = std::cos(h->max_fov);
  const double fmax = std::max(std::fabs(h->intr[0]), std::fabs(h->intr[1]));
  const double frame = double(h->W) + double(h->H) + std::fabs(h->intr[2]) + std::fabs(h->intr[3]);
  // sup of the radial factor, of its derivative and of the row sums of d(dx, dy)/d(px, py) of the radial-tangential distortion
  // on |p| <= pmax (plumb_bob: k1 k2 p1 p2 k3; omnidir: k1 k2 p1 p2)
  auto radtan_sup = [&](double pmax, double k3, double& R, double& K) {
    const double r2 = pmax * pmax;
    const double k1 = std::fabs(h->dist[0]), k2 = std::fabs(h->dist[1]), p1 = std::fabs(h->dist[2]), p2 = std::fabs(h->dist[3]);
    R = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3));
    const double Rp = k1 + r2 * (2.0 * k2 + r2 * 3.0 * k3);
    const double Kx = R + 4.0 * r2 * Rp + 4.0 * p1 * pmax + 8.0 * p2 * pmax;
    const double Ky = R + 4.0 * r2 * Rp + 8.0 * p1 * pmax + 4.0 * p2 * pmax;
    K = 1.5 * std::max(Kx, Ky);
  };
  if (h->model == NIDREG_MODEL_PLUMB_BOB) {
    if (!(cos_fov > 0.1)) return f;
    const double pmax = std::tan(h->max_fov) * 1.001 + 1e-6;
    double R, K;
    radtan_sup(pmax, std::fabs(h->dist[4]), R, K);
    f.A = 2.0 * fmax * K * (1.0 + pmax);
    f.Bc = fmax * (4e-14 * K * pmax + 2e-14 * R * pmax) + 1e-15 * frame;
  } else if (h->model == NIDREG_MODEL_OMNIDIR) {
    const double xi = std::fabs(h->intr[4]);
    if (!(cos_fov + xi > 0.1) || !(h->intr[4] >= 0.0)) return f;
    // |m| = sin(theta) / (cos(theta) + xi): d/dtheta = (1 + xi cos(theta)) / (cos(theta) + xi)^2.  For xi <= 1 it grows with theta
    // on [0, max_fov] while the denominator stays positive; for xi > 1 it peaks at theta* = acos(-1 / xi) with the value
    // 1 / sqrt(xi^2 - 1), and a cone that reaches past theta* has THAT as its supremum, not the value at its rim
    const double fov_c = std::min(h->max_fov, pi);
    const double m_rim = std::sin(fov_c) / (cos_fov + xi);
    const double m_sup = (xi > 1.0 && fov_c > std::acos(-1.0 / xi)) ? 1.0 / std::sqrt(xi * xi - 1.0) : m_rim;
    const double mmax = m_sup * 1.001 + 1e-6;
    double R, K;
    radtan_sup(mmax, 0.0, R, K);
    f.A = 2.0 * fmax * K * (1.0 + mmax * (1.0 + 1.74 * xi));
    const double rel_m = 2.1e-14 * xi / (cos_fov + xi) + 1.4e-14 + 8.0 * eps;  // 1 / (cz + xi |c|): one-step rsqrt inside, one-step reciprocal
    f.Bc = fmax * (4.0 * rel_m * K * mmax + 2e-14 * R * mmax) + 1e-15 * frame;
  } else if (h->model == NIDREG_MODEL_FISHEYE) {
    const double th = 0.5 * pi, t2 = th * th;
    const double D = 1.0 + t2 * (3.0 * std::fabs(h->dist[0]) + t2 * (5.0 * std::fabs(h->dist[1]) + t2 * (7.0 * std::fabs(h->dist[2]) + t2 * 9.0 * std::fabs(h->dist[3]))));
    f.A = 2.0 * 1.5 * fmax * D;  // e1 (1.5 D / |c| + 2 s) fmax, doubled
    f.C = 2.0 * 2.0 * fmax;
    f.Bc = 2e-13 * std::max(1.0, D);  // relative: the one-step rsqrt (2.1e-14), its share of theta through atan2, theta_d's four fmas, s x
  } else if (h->model == NIDREG_MODEL_EQUIRECTANGULAR) {
    // round 6: decided on the pixel boundaries (nid_kernels.hpp below NearestFast): the tables built at creation, bands per point
    if (!h->d_eq_tab) return f;
    f.tab_c = h->d_eq_tab;
    f.tab_r = h->d_eq_tab + 2 * size_t(h->eq_kmax + 1);
    f.kmax = h->eq_kmax;
    f.jmax = h->eq_jmax;
  } else {
    return f;
  }
  f.on = std::isfinite(f.A) && std::isfinite(f.Bc) && std::isfinite(f.C) && std::isfinite(f.D) && std::isfinite(f.Bc2) && std::isfinite(f.er) && std::isfinite(f.et) ? 1 : 0;
  return f;
}

int launch_hist_nearest(nidreg_handle* h, const double* T) {
  PassArgs a;
  fill_pass_args(h, a);
  a.nfast = nearest_fast_args(h, T);
  for (int k = 0; k < 12; k++) a.iso[k] = T[k];
  HIP_TRY(begin_histogram(h));
  a.hist = h->d_hist;
  if (h->timing == 1) HIP_TRY(hipEventRecord(h->ev[1], h->stream));
  HIP_TRY(launch_nearest_hist<double>(a));
  return NIDREG_OK;
}

// tail = false: partials only -- the gradient kernel that follows runs the entropy tail in its prologue (launch_grad with
// from_partials); tail = true: the last workgroup finalises (cost-only evaluations, the split-phase ABI, empty clouds)
int launch_entropy(nidreg_handle* h, double tag, bool tail) {
  const double inv_unit = 1.0 / fixed_unit(h);
  hipLaunchKernelGGL(
    k_entropy<false>, dim3(h->NEB), dim3(kEntropyThreads), 0, h->stream, h->d_hist, h->bins, kEntropyCols, inv_unit, h->d_part_hj, h->d_row_part, h->d_phi_q, h->d_hist_image,
    h->d_hist_points, h->d_scal, h->d_out, h->d_out_host, tag, h->d_counters, h->own_hist ? h->d_hist_buf[h->hist_cur ^ 1] : nullptr, h->hist_words, tail ? 1 : 0,
    static_cast<const MultiEntry*>(nullptr), NoMultiDyn());
  HIP_TRY(hipGetLastError());
  if (h->own_hist) {
    h->hist_zeroed[h->hist_cur ^ 1] = true;  // zeroed by this k_entropy for the next evaluation
    h->zero_stream = h->stream;
  }
  return NIDREG_OK;
}

// small tables (B <= 32): a cost+Jacobian evaluation launches no entropy kernel, the gradient workgroups sum the table
// themselves (nid_kernels.hpp kSelfEntropyCells).  NIDREG_NO_SELF_ENTROPY=1: always k_entropy (A/B runs)
bool grad_sums_table(const nidreg_handle* h) {
  if (h->self_entropy < 0) {  // decided at the handle's first cost+Jacobian evaluation
    const char* e = std::getenv("NIDREG_NO_SELF_ENTROPY");
    const bool off = e && *e && *e != '0';
    const_cast<nidreg_handle*>(h)->self_entropy = (!off && h->mode == NIDREG_MODE_SPLINE && h->GW != 1 && h->bins * h->bins <= kSelfEntropyCells && !h->is_shard && !h->set) ? 1 : 0;
  }
  return h->self_entropy == 1;
}

// stand-alone finalisation of an empty pair (no gradient workgroup exists to do it): zeros through k_grad_final
hipError_t launch_grad_final(hipStream_t stream, const double* partials, const double* q4, double* out, double* out_host, double tag) {
  hipLaunchKernelGGL(k_grad_final, dim3(1), dim3(kThreads), 0, stream, partials, 0, q4[0], q4[1], q4[2], q4[3], out, out_host, tag);
  return hipGetLastError();
}

int launch_grad(nidreg_handle* h, bool alone, int from_partials) {
  PassArgs a;
  fill_pass_args(h, a);
  a.prio = alone ? 1 : 0;
  a.gt_from_partials = from_partials;
  if (from_partials == 2) {
    a.gt_zero_buf = h->own_hist ? h->d_hist_buf[h->hist_cur ^ 1] : nullptr;
    a.gt_zero_words = h->hist_words;
    if (h->own_hist) {
      h->hist_zeroed[h->hist_cur ^ 1] = true;  // zeroed by this evaluation's gradient kernel for the next one
      h->zero_stream = h->stream;
    }
  }
  a.hist = h->d_hist;  // the finished histogram (for a shard: its own columns)
  // same pose as the histogram pass of this evaluation
  std::memcpy(a.R, h->last_R, sizeof(a.R));
  std::memcpy(a.t, h->last_t, sizeof(a.t));
  HIP_TRY(launch_spline_grad<double>(a));
  if (h->timing == 1) HIP_TRY(hipEventRecord(h->ev[4], h->stream));
  if (h->nchunks == 0) {  // empty cloud: no gradient workgroups ran, finalise (zeros) stand-alone
    HIP_TRY(launch_grad_final(h->stream, h->d_partials, h->last_q, h->d_out, h->d_out_host, h->seq));
  }
  return NIDREG_OK;
}

// asynchronous part of nidreg_eval, in two steps so that a multi-handle caller can put every GPU to work before it
// queues the rest: eval_launch_first = the histogram pass, eval_launch_rest = entropy (+ gradient)
int eval_launch_first(nidreg_handle* h, const double* se3, bool alone) {
  if (h->mode != NIDREG_MODE_SPLINE) return fail(NIDREG_ERR_INVALID, "nidreg_eval: handle was created in NEAREST mode");
  HIP_TRY(hipSetDevice(h->device));
  bump_seq(h);
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  const int rc = launch_hist_spline(h, se3, alone);
  if (rc) return rc;
  if (h->timing == 1) HIP_TRY(hipEventRecord(h->ev[2], h->stream));
  return NIDREG_OK;
}
int eval_launch_rest(nidreg_handle* h, bool want_grad, bool alone) {
  HIP_TRY(hipSetDevice(h->device));
  // cost + Jacobian on a non-empty cloud: k_entropy stores its partials and ends; every gradient workgroup runs the tail
  const bool grad_runs_tail = want_grad && h->nchunks > 0;
  const bool no_entropy_kernel = grad_runs_tail && grad_sums_table(h);
  int rc = no_entropy_kernel ? NIDREG_OK : launch_entropy(h, want_grad ? 0.0 : h->seq, !grad_runs_tail);
  if (rc) return rc;
  if (h->timing == 1) HIP_TRY(hipEventRecord(h->ev[3], h->stream));
  h->ev_grad = want_grad;
  if (want_grad) {
    rc = launch_grad(h, alone, no_entropy_kernel ? 2 : (grad_runs_tail ? 1 : 0));
    if (rc) return rc;
  } else if (h->timing == 1) {
    HIP_TRY(hipEventRecord(h->ev[4], h->stream));
  }
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[5], h->stream));
  if (!h->d_out_host) HIP_TRY(hipMemcpyAsync(h->h_out, h->d_out, NIDREG_OUT_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  return NIDREG_OK;
}

// ---- one launch per cost+Jacobian evaluation (nid_fused.hpp) -----------------------------------------------------------------
// Planned at the handle's first eligible evaluation.  The fused kernel runs over the handle's OWN gradient-pass chunk table with
// the thread <-> point mapping, per-point arithmetic and reductions of k_spline_grad, so the route changes nothing in the results:
// cost AND gradient have the bits of the three-kernel route (which route runs depends on whether the evaluation has its device to
// itself -- results must not).  Usable when every chunk lies inside one column group, the whole table is one round of
// co-resident workgroups of the fused kernel, and the longest chunk fits the LDS stash (the full format where it does, (u, v)
// only otherwise).  NIDREG_FUSED=0 switches the route off, NIDREG_FUSED_STASH=uv|full forces a format (A/B runs).
void plan_fused(nidreg_handle* h) {
  h->fused = -1;
  const char* off = std::getenv("NIDREG_FUSED");
  if (off && *off == '0') return;
  if (h->mode != NIDREG_MODE_SPLINE || !grad_sums_table(h) || h->is_shard || h->set || !h->own_hist || h->cohort || h->nchunks <= 0 || h->seg || h->nslots != h->nchunks ||
      h->longest_chunk <= 0)
    return;
  if (hipSetDevice(h->device) != hipSuccess) return;
  int lds_max = 64 * 1024;
  if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) != hipSuccess) return;
  PassArgs a;
  fill_pass_args(h, a);
  const char* st = std::getenv("NIDREG_FUSED_STASH");
  const int cap = int((uint32_t(h->longest_chunk) + 63u) & ~63u);
  for (int f = 1; f >= 0; f--) {
    if (st && ((f == 1 && st[0] == 'u') || (f == 0 && st[0] == 'f'))) continue;
    if (fused_lds_bytes_for(a, f, cap) > size_t(lds_max)) continue;
    const FusedArgs fa{nullptr, 0, nullptr, 0, 0, cap, f};
    const int occ = occupancy_spline_fused(a, fa);
    if (occ <= 0 || int64_t(occ) * h->num_cus < int64_t(h->nchunks)) continue;
    const size_t sbytes = 256 + size_t(h->nchunks) * 128;  // the arrival counter, then one release word per workgroup in a line of its own
    if (hipMalloc(&h->d_fused_scratch, sbytes) != hipSuccess) return;
    if (hipMemset(h->d_fused_scratch, 0, sbytes) != hipSuccess) return;
    if (hipDeviceSynchronize() != hipSuccess) return;  // (the null-stream memset against the handle's non-blocking stream)
    h->d_fused_barrier = static_cast<u64*>(h->d_fused_scratch);
    h->fused_launches = 0;
    h->fused_cap = cap;
    h->fused_full = f;
    h->fused_arrivals = 0;
    h->fused = 1;
    return;
  }
}
bool fused_planned(nidreg_handle* h) {
  if (h->fused == 0) plan_fused(h);
  return h->fused == 1;
}
bool fused_usable(nidreg_handle* h) {
  if (h->fused == 0) plan_fused(h);
  return h->fused == 1 && h->timing != 1 && !h->cohort && !h->rccl_comm && h->d_out_host != nullptr;
}
// after a barrier that timed out (nid_fused.hpp: two half-resident grids of different processes): the route is off for this
// handle, its counters are cleared (workgroups that gave up never drew their tickets)
void fused_give_up(nidreg_handle* h) {
  h->fused = -1;
  h->fused_last = false;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  (void)hipMemsetAsync(h->d_counters, 0, 8 * sizeof(unsigned int), h->stream);
  h->hist_zeroed[0] = h->hist_zeroed[1] = false;  // (whatever the aborted kernel cleared or did not: memset before use)
  (void)hipStreamSynchronize(h->stream);
}
int eval_launch_fused(nidreg_handle* h, const double* se3) {
  HIP_TRY(hipSetDevice(h->device));
  bump_seq(h);
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  for (int k = 0; k < 4; k++) h->last_q[k] = se3[k];
  pose_from_se3(se3, h->last_R, h->last_t);
  HIP_TRY(begin_histogram(h));
  PassArgs a;
  fill_pass_args(h, a);  // (after begin_histogram: a.hist = this evaluation's buffer; a.q = the pose's quaternion; a.tag = its sequence number)
  std::memcpy(a.R, h->last_R, sizeof(a.R));
  std::memcpy(a.t, h->last_t, sizeof(a.t));
  a.gt_zero_buf = h->d_hist_buf[h->hist_cur ^ 1];
  a.gt_zero_words = h->hist_words;
  h->hist_zeroed[h->hist_cur ^ 1] = true;  // cleared by this launch for the next evaluation
  h->zero_stream = h->stream;
  h->fused_arrivals += uint64_t(h->nchunks);
  h->fused_launches += 1;
  const char* tmo = std::getenv("NIDREG_FUSED_TIMEOUT_US");  // (default 5 ms: far beyond any barrier wait of a co-resident grid)
  const unsigned long long timeout_ticks = (unsigned long long)(100.0 * (tmo ? std::max(10.0, std::strtod(tmo, nullptr)) : 5000.0));
  // test hook (tests/test_gpu_parity.py): a barrier target no launch can reach -- every workgroup times out, the kernel ends without
  // its tag, eval_one falls back to the three kernels
  const char* hang = std::getenv("NIDREG_FUSED_TEST_HANG");
  const FusedArgs f{h->d_fused_barrier, h->fused_arrivals + ((hang && *hang == '1') ? 1u : 0u), h->d_fused_barrier + 32, h->fused_launches, timeout_ticks, h->fused_cap, h->fused_full};
  HIP_TRY(launch_spline_fused(a, f));
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[5], h->stream));
  h->ev_grad = true;
  h->fused_last = true;
  return NIDREG_OK;
}

int eval_launch(nidreg_handle* h, const double* se3, bool want_grad, bool alone) {
  h->fused_last = false;
  const int rc = eval_launch_first(h, se3, alone);
  if (rc) return rc;
  return eval_launch_rest(h, want_grad, alone);
}
// `stream` = the stream the evaluation's kernels were queued on (the handle's own, or a multi-pair group's)
int eval_finish_on(nidreg_handle* h, hipStream_t stream, double* cost, double* grad7) { return eval_finish_block(h, stream, h->h_out, h->seq_bits, h->d_out_host != nullptr, cost, grad7); }
int eval_finish(nidreg_handle* h, double* cost, double* grad7) { return eval_finish_on(h, h->stream, cost, grad7); }
// `block` = the host-mapped result block the evaluation writes (the handle's own, or a slot of its asynchronous ring) and
// `seq_bits` the completion tag expected in its last word
int eval_finish_block(nidreg_handle* h, hipStream_t stream, const double* block, uint64_t seq_bits, bool polled, double* cost, double* grad7) {
  HIP_TRY(hipSetDevice(h->device));
  if (polled) {
    // the finalising workgroup wrote the results and then this evaluation's tag into host-mapped memory:
    // poll the tag (a few us cheaper than hipStreamSynchronize); look at the stream now and then so that a
    // faulted kernel cannot hang the caller, and so the runtime can retire finished commands
    // acquire load of the tag, then plain loads of the payload.  Back-off by elapsed TIME, not by spin count (a count
    // means a different wait on every host CPU: with naps starting too early, their ~60 us granularity added 40-50 us
    // to every 100-200 us evaluation): `pause` spinning for the first 2 ms -- every evaluation up to ~100M points --,
    // then 50 us naps, so that N in-flight handles (one OpenMP thread per pair in the reference) do not burn N cores
    // through a long wait; the stream is looked at once per millisecond so that a faulted kernel cannot hang the caller.
    const double* flag = block + 15;
    auto tag_seen = [&]() { return __atomic_load_n(reinterpret_cast<const uint64_t*>(flag), __ATOMIC_ACQUIRE) == seq_bits; };
    if (!tag_seen()) {
      struct timespec t0;
      clock_gettime(CLOCK_MONOTONIC, &t0);
      double next_query_us = 1000.0;
      unsigned spins = 0;
      for (;;) {
        for (int k = 0; k < 32 && !tag_seen(); k++) __builtin_ia32_pause();
        if (tag_seen()) break;
        if ((++spins & 7u) != 0) continue;  // look at the clock every ~256 pauses
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double us = (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3;
        if (us >= next_query_us) {
          next_query_us = us + 1000.0;
          const hipError_t q = hipStreamQuery(stream);
          if (q == hipSuccess) {
            if (!tag_seen()) HIP_TRY(hipStreamSynchronize(stream));
            if (!tag_seen()) return fail(NIDREG_ERR_HIP, "nidreg_eval: stream drained but the completion tag is missing");
            break;
          }
          if (q != hipErrorNotReady) return fail(NIDREG_ERR_HIP, std::string("nidreg_eval: ") + hipGetErrorString(q));
        }
        if (us > 2000.0) {
          struct timespec ts = {0, 50000};
          nanosleep(&ts, nullptr);
        }
      }
    }
    if (++h->evals_since_reap >= 256) {
      h->evals_since_reap = 0;
      (void)hipStreamQuery(stream);
    }
  } else {
    HIP_TRY(hipStreamSynchronize(stream));
  }
  if (cost) *cost = block[0];
  if (grad7)
    for (int k = 0; k < 7; k++) grad7[k] = block[1 + k];
  return block[8] != 0.0 ? NIDREG_FALSE : NIDREG_OK;
}

// one synchronous evaluation of a plain handle: the fused single launch when the evaluation is a cost+Jacobian one on a small
// table, has its device to itself and fits on chip; the three-kernel route otherwise -- and again when the fused kernel's grid
// barrier gave up (the kernel then ends without its completion tag: eval_finish_block reports the drained stream)
int eval_one(nidreg_handle* h, const double* se3, double* cost, double* grad7) {
  InflightGuard guard(h->device);
  if (grad7 && guard.alone && h->async_outstanding == 0 && fused_usable(h)) {
    int rc = eval_launch_fused(h, se3);
    if (rc) return rc;
    rc = eval_finish(h, cost, grad7);
    if (rc >= 0) return rc;
    fused_give_up(h);  // ... and fall through to the three kernels
  }
  const int rc = eval_launch(h, se3, grad7 != nullptr, guard.alone);
  if (rc) return rc;
  return eval_finish(h, cost, grad7);
}

int iso_launch(nidreg_handle* h, const double* T) {
  if (h->mode != NIDREG_MODE_NEAREST) return fail(NIDREG_ERR_INVALID, "nidreg_eval_iso: handle was created in SPLINE mode");
  HIP_TRY(hipSetDevice(h->device));
  bump_seq(h);
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

}  // namespace nidreg_detail
