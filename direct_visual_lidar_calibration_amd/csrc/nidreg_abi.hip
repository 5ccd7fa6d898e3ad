#include "nidreg_internal.hpp"

extern "C" {

const char* nidreg_last_error(void) { return g_last_error.c_str(); }
const char* nidreg_version(void) { return "nidreg 0.6 (gfx950, hand-written HIP)"; }
#ifndef NIDREG_KERNEL_BUILD
#define NIDREG_KERNEL_BUILD "unstamped"
#endif
const char* nidreg_kernel_build(void) { return NIDREG_KERNEL_BUILD; }

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
  return int64_t(bins) * bins + kTailWords + 2 * ((bins + 7) & ~7);  // joint histogram, tail, column sums, row sums
}

}  // extern "C"

extern "C" {

static int create_done(int rc, nidreg_handle** out) {
  if (rc == NIDREG_OK && out && *out) cohort_join(*out);
  return rc;
}

int nidreg_create(const nidreg_desc* d, nidreg_handle** out) {
  if (d && out && d->struct_size == int32_t(sizeof(nidreg_desc)) && wants_shards(d)) return create_sharded(d, nullptr, nullptr, 0.0, 0, out);
  return create_done(create_impl(d, nullptr, nullptr, 0.0, 0, CreateOpts(), out), out);
}

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
  // desc.device_ids / NIDREG_DEVICES: cull + bucket + sort on the cloud's GPU, then every shard takes its column groups
  // device to device -- the per-outer-iteration `cull -> new NIDCost` stays on the GPUs
  if (d && out && d->struct_size == int32_t(sizeof(nidreg_desc)) && wants_shards(d)) return create_sharded(d, cloud, T_camera_lidar, min_z, enable_depth_buffer_culling, out);
  return create_done(create_impl(d, cloud, T_camera_lidar, min_z, enable_depth_buffer_culling, CreateOpts(), out), out);
}

void nidreg_destroy(nidreg_handle* h) { free_handle(h); }

int nidreg_eval(nidreg_handle* h, const double* se3, double* cost, double* grad7) {
  if (!h || !se3) return fail(NIDREG_ERR_INVALID, "nidreg_eval: null argument");
  if (h->set) return set_eval(h->set, NIDREG_MODE_SPLINE, se3, cost, grad7);
  if (h->rccl_comm) return rccl_eval(h, NIDREG_MODE_SPLINE, se3, cost, grad7);
  cohort_check(h);
  if (h->cohort && h->cohort->members.size() >= 2 && h->mode == NIDREG_MODE_SPLINE && !h->timing && h->async_outstanding == 0) {
    const int rc = cohort_eval(h, se3, grad7 != nullptr, cost, grad7);
    if (rc != kNotJoined) return rc;
  }
  return eval_one(h, se3, cost, grad7);
}

// ---- asynchronous evaluations: nidreg_submit* queue the kernels of one evaluation and return, nidreg_wait collects.  The
// kernels of consecutive evaluations of a handle run back to back on its stream (the histogram double buffer and the
// scratch are only ever touched in stream order), each evaluation writes its results and completion tag into its own block
// of a host-mapped ring, and the host turnaround between evaluations (7-14 us of every synchronous one) disappears for a
// caller that holds several independent poses: Nelder-Mead's initial simplex (nelder_mead.hpp:32-57), multi-start, batches.
constexpr int kAsyncDepth = 8;
static int async_submit(nidreg_handle* h, int mode, const double* pose, bool want_grad, int64_t* ticket) {
  if (!h || !pose || !ticket) return fail(NIDREG_ERR_INVALID, "nidreg_submit: null argument");
  if (h->mode != mode) return fail(NIDREG_ERR_INVALID, mode == NIDREG_MODE_SPLINE ? "nidreg_submit: handle was created in NEAREST mode" : "nidreg_submit_iso: handle was created in SPLINE mode");
  if (h->async_outstanding >= kAsyncDepth) return fail(NIDREG_ERR_INVALID, "nidreg_submit: too many evaluations in flight on this handle (8): nidreg_wait first");
  const int64_t t = h->next_ticket + 1;
  nidreg_handle::Pending& p = h->pending[t % kAsyncDepth];
  if (p.ticket != 0) return fail(NIDREG_ERR_INVALID, "nidreg_submit: ticket ring collision (wait for the oldest evaluation first)");
  if (h->set || h->rccl_comm || !h->d_out_host || h->timing) {
    // a handle sharded over several GPUs (its shards hand-shake inside the kernels), results in a caller's buffer, or
    // per-kernel timing: evaluated here and now, the ticket just carries the results
    double c = 0.0, g[7] = {0};
    const int rc = mode == NIDREG_MODE_SPLINE ? nidreg_eval(h, pose, &c, want_grad ? g : nullptr) : nidreg_eval_iso(h, pose, &c);
    if (rc < 0) return rc;
    p.ticket = t;
    p.done = true;
    p.counted = false;
    p.rc = rc;
    p.grad = want_grad;
    p.res[0] = c;
    for (int k = 0; k < 7; k++) p.res[1 + k] = g[k];
    h->next_ticket = t;
    h->async_outstanding++;
    *ticket = t;
    return NIDREG_OK;
  }
  cohort_check(h);
  HIP_TRY(hipSetDevice(h->device));
  if (!h->h_ring) {
    {
      void* blk = nullptr;
      HIP_TRY(pool_host_block(h->device, true, size_t(kAsyncDepth) * NIDREG_OUT_DOUBLES * sizeof(double), &blk));
      h->h_ring = static_cast<double*>(blk);
    }
    std::memset(h->h_ring, 0, size_t(kAsyncDepth) * NIDREG_OUT_DOUBLES * sizeof(double));
    void* dp = nullptr;
    HIP_TRY(hipHostGetDevicePointer(&dp, h->h_ring, 0));
    h->d_ring = static_cast<double*>(dp);
  }
  // the launch helpers write to h->d_out_host with tag h->seq: point them at this evaluation's ring block for the launch
  double* const own = h->d_out_host;
  h->d_out_host = h->d_ring + size_t(t % kAsyncDepth) * NIDREG_OUT_DOUBLES;
  // alone on the device = nothing in flight but this handle's own earlier submissions (they run before it, in stream order)
  const bool alone = h->device >= 0 && h->device < NIDREG_MAX_DEVICES && g_inflight[h->device].fetch_add(1, std::memory_order_acq_rel) == h->async_outstanding;
  const int rc = mode == NIDREG_MODE_SPLINE ? eval_launch(h, pose, want_grad, alone) : iso_launch(h, pose);
  h->d_out_host = own;
  if (rc) {
    if (h->device >= 0 && h->device < NIDREG_MAX_DEVICES) g_inflight[h->device].fetch_sub(1, std::memory_order_acq_rel);
    return rc;
  }
  p.ticket = t;
  p.bits = h->seq_bits;  // the tag this evaluation's final workgroup writes behind its results
  p.grad = want_grad;
  p.done = false;
  p.counted = true;
  h->next_ticket = t;
  h->async_outstanding++;
  *ticket = t;
  return NIDREG_OK;
}

int nidreg_submit(nidreg_handle* h, const double* se3, int want_grad, int64_t* ticket) { return async_submit(h, NIDREG_MODE_SPLINE, se3, want_grad != 0, ticket); }
int nidreg_submit_iso(nidreg_handle* h, const double* T, int64_t* ticket) { return async_submit(h, NIDREG_MODE_NEAREST, T, false, ticket); }

int nidreg_wait(nidreg_handle* h, int64_t ticket, double* cost, double* grad7) {
  if (!h || ticket <= 0) return fail(NIDREG_ERR_INVALID, "nidreg_wait: bad argument");
  nidreg_handle::Pending& p = h->pending[ticket % kAsyncDepth];
  if (p.ticket != ticket) return fail(NIDREG_ERR_INVALID, "nidreg_wait: unknown ticket (already collected, or never issued by this handle)");
  int rc;
  if (p.done) {
    rc = p.rc;
    if (cost) *cost = p.res[0];
    if (grad7 && p.grad)
      for (int k = 0; k < 7; k++) grad7[k] = p.res[1 + k];
  } else {
    rc = eval_finish_block(h, h->stream, h->h_ring + size_t(ticket % kAsyncDepth) * NIDREG_OUT_DOUBLES, p.bits, true, cost, p.grad ? grad7 : nullptr);
  }
  if (p.counted && h->device >= 0 && h->device < NIDREG_MAX_DEVICES) g_inflight[h->device].fetch_sub(1, std::memory_order_acq_rel);
  p = nidreg_handle::Pending();
  h->async_outstanding--;
  if (h->mode == NIDREG_MODE_NEAREST && rc >= 0) rc = NIDREG_OK;  // CostCalculatorNID has no finite check
  return rc;
}

int nidreg_eval_batch(nidreg_handle* h, const double* se3s, int n, double* costs, double* grads7) {
  if (!h || !se3s || n < 0) return fail(NIDREG_ERR_INVALID, "nidreg_eval_batch: bad argument");
  int worst = NIDREG_OK;
  for (int i = 0; i < n; i++) {
    double c = 0.0;
    const int rc = nidreg_eval(h, se3s + 7 * size_t(i), &c, grads7 ? grads7 + 7 * size_t(i) : nullptr);
    if (rc < 0) return rc;
    if (rc != NIDREG_OK) worst = rc;
    if (costs) costs[i] = c;
  }
  return worst;
}

// n INDEPENDENT poses through the submit / wait pair: up to kAsyncDepth evaluations queued ahead of the one being collected
int nidreg_eval_pipelined(nidreg_handle* h, const double* se3s, int n, double* costs, double* grads7) {
  if (!h || !se3s || n < 0) return fail(NIDREG_ERR_INVALID, "nidreg_eval_pipelined: bad argument");
  if (h->async_outstanding != 0) return fail(NIDREG_ERR_INVALID, "nidreg_eval_pipelined: collect the handle's outstanding tickets first");
  int worst = NIDREG_OK;
  int64_t tickets[kAsyncDepth];
  int head = 0, tail = 0;  // poses submitted / collected
  const int depth = kAsyncDepth - 1;
  auto drain = [&]() {  // after a failure: collect what is still in flight so that the handle stays usable
    for (; tail < head; tail++) (void)nidreg_wait(h, tickets[tail % kAsyncDepth], nullptr, nullptr);
  };
  while (tail < n) {
    while (head < n && head - tail < depth) {
      const int rc = nidreg_submit(h, se3s + 7 * size_t(head), grads7 != nullptr, &tickets[head % kAsyncDepth]);
      if (rc < 0) {
        drain();
        return rc;
      }
      head++;
    }
    double c = 0.0;
    const int rc = nidreg_wait(h, tickets[tail % kAsyncDepth], &c, grads7 ? grads7 + 7 * size_t(tail) : nullptr);
    tail++;
    if (rc < 0) {
      drain();
      return rc;
    }
    if (rc != NIDREG_OK) worst = rc;
    if (costs) costs[tail - 1] = c;
  }
  return worst;
}

int nidreg_eval_iso(nidreg_handle* h, const double* T, double* cost) {
  if (!h || !T) return fail(NIDREG_ERR_INVALID, "nidreg_eval_iso: null argument");
  if (h->set) return set_eval(h->set, NIDREG_MODE_NEAREST, T, cost, nullptr) < 0 ? NIDREG_ERR_HIP : NIDREG_OK;
  if (h->rccl_comm) return rccl_eval(h, NIDREG_MODE_NEAREST, T, cost, nullptr) < 0 ? NIDREG_ERR_HIP : NIDREG_OK;
  cohort_check(h);
  const int rc = iso_launch(h, T);
  if (rc) return rc;
  return eval_finish(h, cost, nullptr) < 0 ? NIDREG_ERR_HIP : NIDREG_OK;  // CostCalculatorNID has no finite check
}

int nidreg_eval_multi(nidreg_handle* const* handles, int n, const double* init_se3, const double* se3, double* cost, double* grad7) {
  if (!handles || n <= 0 || !se3) return fail(NIDREG_ERR_INVALID, "nidreg_eval_multi: bad argument");
  if (init_se3 && !trust_gate_ok(init_se3, se3)) return NIDREG_FALSE;
  for (int i = 0; i < n; i++)
    if (!handles[i]) return fail(NIDREG_ERR_INVALID, "nidreg_eval_multi: null handle");
  for (int i = 0; i < n; i++)
    if (handles[i]->rccl_comm) return fail(NIDREG_ERR_INVALID, "nidreg_eval_multi: a handle with a communicator (nidreg_shard_attach_rccl) is a collective of its own: evaluate it with nidreg_eval");
  for (int i = 0; i < n; i++) cohort_check(handles[i]);
  // several compatible pairs on ONE GPU: a single grid per pass over all pairs (group_eval)
  if (handles[0]->mode == NIDREG_MODE_SPLINE) {
    if (can_group(handles, n)) {
      MultiGroup* g = find_or_make_group(handles, n);
      if (g) {
        double costs[kMaxMulti], grads[kMaxMulti * 7];
        bool all_ok = true;
        const int rc = group_eval(g, se3, grad7 != nullptr, costs, grad7 ? grads : nullptr, &all_ok);
        release_group(g);
        if (rc < 0) return rc;
        double csum = 0.0, gsum[7] = {0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < n; i++) {
          csum += costs[i];
          if (grad7)
            for (int k = 0; k < 7; k++) gsum[k] += grads[7 * i + k];
        }
        if (cost) *cost = csum;
        if (grad7)
          for (int k = 0; k < 7; k++) grad7[k] = gsum[k];
        return all_ok ? NIDREG_OK : NIDREG_FALSE;
      }
    }
  }
  if (n == 1 && !handles[0]->set && handles[0]->mode == NIDREG_MODE_SPLINE) return eval_one(handles[0], se3, cost, grad7);  // (the trust gate has passed above)
  // progress priority only for a pair that is alone on its device
  std::vector<std::unique_ptr<InflightGuard>> guards(static_cast<size_t>(n));
  std::vector<char> alone(static_cast<size_t>(n), 0);
  for (int i = 0; i < n; i++) {
    if (handles[i]->set) continue;
    guards[size_t(i)].reset(new InflightGuard(handles[i]->device));
    int same = 0;
    for (int j = 0; j < n; j++) same += (!handles[j]->set && handles[j]->device == handles[i]->device) ? 1 : 0;
    alone[size_t(i)] = guards[size_t(i)]->alone && same == 1;
  }
  for (int i = 0; i < n; i++) {
    if (handles[i]->set) continue;  // a pair sharded over several GPUs: evaluated through its set below
    const int rc = eval_launch_first(handles[i], se3, alone[size_t(i)] != 0);  // every pair's (every GPU's) histogram pass is running ...
    if (rc) return rc;
  }
  for (int i = 0; i < n; i++) {
    if (handles[i]->set) continue;
    const int rc = eval_launch_rest(handles[i], grad7 != nullptr, alone[size_t(i)] != 0);  // ... while the rest is queued behind it
    if (rc) return rc;
  }
  double csum = 0.0, gsum[7] = {0, 0, 0, 0, 0, 0, 0};
  bool all_ok = true;
  for (int i = 0; i < n; i++) {
    double c = 0.0, g[7];
    const int rc = handles[i]->set ? set_eval(handles[i]->set, NIDREG_MODE_SPLINE, se3, &c, grad7 ? g : nullptr) : eval_finish(handles[i], &c, grad7 ? g : nullptr);
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
  for (int i = 0; i < n; i++)
    if (!handles[i]) return fail(NIDREG_ERR_INVALID, "nidreg_eval_iso_multi: null handle");
  for (int i = 0; i < n; i++)
    if (handles[i]->rccl_comm) return fail(NIDREG_ERR_INVALID, "nidreg_eval_iso_multi: a handle with a communicator (nidreg_shard_attach_rccl) is a collective of its own: evaluate it with nidreg_eval_iso");
  for (int i = 0; i < n; i++) cohort_check(handles[i]);
  if (handles[0]->mode == NIDREG_MODE_NEAREST && can_group(handles, n)) {  // several pairs on one GPU: one grid per pass
    MultiGroup* g = find_or_make_group(handles, n);
    if (g) {
      double costs[kMaxMulti];
      const int rc = group_eval_iso(g, T, costs);
      release_group(g);
      if (rc < 0) return rc;
      double csum = 0.0;
      for (int i = 0; i < n; i++) csum += costs[i];
      if (cost) *cost = csum;
      return NIDREG_OK;
    }
  }
  for (int i = 0; i < n; i++) {
    if (!handles[i]) return fail(NIDREG_ERR_INVALID, "nidreg_eval_iso_multi: null handle");
    if (handles[i]->set) continue;
    const int rc = iso_launch(handles[i], T);
    if (rc) return rc;
  }
  double csum = 0.0;
  for (int i = 0; i < n; i++) {
    double c = 0.0;
    const int rc = handles[i]->set ? set_eval(handles[i]->set, NIDREG_MODE_NEAREST, T, &c, nullptr) : eval_finish(handles[i], &c, nullptr);
    if (rc < 0) return rc;
    csum += c;
  }
  if (cost) *cost = csum;
  return NIDREG_OK;
}

int nidreg_get_hist_fixed(nidreg_handle* h, int64_t* joint, int64_t* inliers, int* frac_bits) {
  if (!h) return fail(NIDREG_ERR_INVALID, "nidreg_get_hist_fixed: null handle");
  if (h->set) {  // every shard holds a replica of the whole histogram once an evaluation has run: read the leader's
    for (nidreg_handle* sh : h->set->shards) {
      HIP_TRY(hipSetDevice(sh->device));
      HIP_TRY(hipStreamSynchronize(sh->stream));
    }
  }
  HIP_TRY(hipSetDevice(h->device));
  // the marginals / scalars are plain stores of a gradient workgroup: the host sees the completion tag before the kernel has
  // ended, so drain the stream the evaluation really ran on (a multi-pair group's stream is not the handle's)
  HIP_TRY(hipStreamSynchronize(h->last_stream ? h->last_stream : h->stream));
  const int B = h->bins;
  std::vector<u64> tmp(size_t(h->hist_words));
  HIP_TRY(hipMemcpy(tmp.data(), h->d_hist, tmp.size() * sizeof(u64), hipMemcpyDeviceToHost));
  if (joint) {
    // device layout [bin_points][bin_image] -> [bin_image][bin_points]
    if (h->bins_user) {  // bins > 256: the compact bins back to the caller's (every other cell is empty)
      const size_t Bu = size_t(h->bins_user);
      std::fill(joint, joint + Bu * Bu, int64_t(0));
      for (size_t c = 0; c < h->inv_pts.size(); c++)
        for (size_t r = 0; r < h->inv_img.size(); r++) joint[size_t(h->inv_img[r]) * Bu + size_t(h->inv_pts[c])] = int64_t(tmp[c * size_t(B) + r]);
    } else {
      for (int c = 0; c < B; c++)
        for (int r = 0; r < B; r++) joint[size_t(r) * B + c] = int64_t(tmp[size_t(c) * B + r]);
    }
  }
  if (inliers) *inliers = int64_t(tmp[size_t(B) * B + kTailInliers]);
  if (frac_bits) *frac_bits = h->frac_bits;
  return NIDREG_OK;
}

int nidreg_get_hist(nidreg_handle* h, double* joint, double* hist_image, double* hist_points) {
  if (!h) return fail(NIDREG_ERR_INVALID, "nidreg_get_hist: null handle");
  const int B = h->bins, Bu = h->bins_user ? h->bins_user : h->bins;
  if (joint) {
    std::vector<int64_t> fx(size_t(Bu) * Bu);
    const int rc = nidreg_get_hist_fixed(h, fx.data(), nullptr, nullptr);
    if (rc) return rc;
    const double inv_unit = 1.0 / fixed_unit(h);
    for (size_t k = 0; k < fx.size(); k++) joint[k] = double(fx[k]) * inv_unit;
  }
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipStreamSynchronize(h->last_stream ? h->last_stream : h->stream));  // (see nidreg_get_hist_fixed)
  if (h->bins_user) {
    std::vector<double> hi(static_cast<size_t>(B)), hp(static_cast<size_t>(B));
    HIP_TRY(hipMemcpy(hi.data(), h->d_hist_image, size_t(B) * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(hp.data(), h->d_hist_points, size_t(B) * sizeof(double), hipMemcpyDeviceToHost));
    if (hist_image) {
      std::fill(hist_image, hist_image + Bu, 0.0);
      for (size_t r = 0; r < h->inv_img.size(); r++) hist_image[h->inv_img[r]] = hi[r];
    }
    if (hist_points) {
      std::fill(hist_points, hist_points + Bu, 0.0);
      for (size_t c = 0; c < h->inv_pts.size(); c++) hist_points[h->inv_pts[c]] = hp[c];
    }
    return NIDREG_OK;
  }
  if (hist_image) HIP_TRY(hipMemcpy(hist_image, h->d_hist_image, size_t(B) * sizeof(double), hipMemcpyDeviceToHost));
  if (hist_points) HIP_TRY(hipMemcpy(hist_points, h->d_hist_points, size_t(B) * sizeof(double), hipMemcpyDeviceToHost));
  return NIDREG_OK;
}

int nidreg_project(nidreg_handle* h, const double* p3, int64_t n, double* uv, double* jac) {
  if (!h || !p3 || !uv || n < 0) return fail(NIDREG_ERR_INVALID, "nidreg_project: bad argument");
  if (n == 0) return NIDREG_OK;
  HIP_TRY(hipSetDevice(h->device));
  if (n <= kSmallProject && h->device >= 0 && h->device < NIDREG_MAX_DEVICES) {
    // A handful of points (estimate_camera_fov inverts the projection at three pixels with NelderMead<2>: ~240 calls of ONE
    // point, src/vlcal/common/estimate_fov.cpp:17-51): the same kernel on a host-mapped staging block kept per device -- no
    // hipMalloc / hipMemcpy / hipFree per call (60 -> ~15 us; those calls were 16 of the 24 ms a whole configs[0] calibration
    // took, profiles/archive/r04q_profile_1bag_bfgs.txt).
    SmallProject& sp = g_small_project[h->device];
    std::lock_guard<std::mutex> lk(sp.mu);
    if (!sp.host) {
      void* blk = nullptr;
      HIP_TRY(hipHostMalloc(&blk, size_t(kSmallProject) * 11 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
      void* dp = nullptr;
      hipError_t e0 = hipHostGetDevicePointer(&dp, blk, 0);
      if (e0 != hipSuccess) {
        (void)hipHostFree(blk);
        return fail(NIDREG_ERR_HIP, std::string("nidreg_project: ") + hipGetErrorString(e0));
      }
      sp.host = static_cast<double*>(blk);
      sp.dev = static_cast<double*>(dp);
    }
    std::memcpy(sp.host, p3, size_t(n) * 3 * sizeof(double));
    double* d_uv = sp.dev + 3 * kSmallProject;
    double* d_j = jac ? sp.dev + 5 * kSmallProject : nullptr;
    hipError_t e = launch_project<double>(h->model, h->intr, h->dist, sp.dev, n, d_uv, d_j, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(NIDREG_ERR_HIP, std::string("nidreg_project: ") + hipGetErrorString(e));
    std::memcpy(uv, sp.host + 3 * kSmallProject, size_t(n) * 2 * sizeof(double));
    if (jac) std::memcpy(jac, sp.host + 5 * kSmallProject, size_t(n) * 6 * sizeof(double));
    return NIDREG_OK;
  }
  double *d_p = nullptr, *d_uv = nullptr, *d_j = nullptr;
  hipError_t e = hipMalloc(&d_p, size_t(n) * 3 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&d_uv, size_t(n) * 2 * sizeof(double));
  if (e == hipSuccess && jac) e = hipMalloc(&d_j, size_t(n) * 6 * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(d_p, p3, size_t(n) * 3 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = launch_project<double>(h->model, h->intr, h->dist, d_p, n, d_uv, d_j, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(uv, d_uv, size_t(n) * 2 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && jac) e = hipMemcpy(jac, d_j, size_t(n) * 6 * sizeof(double), hipMemcpyDeviceToHost);
  if (d_p) (void)hipFree(d_p);
  if (d_uv) (void)hipFree(d_uv);
  if (d_j) (void)hipFree(d_j);
  if (e != hipSuccess) return fail(NIDREG_ERR_HIP, std::string("nidreg_project: ") + hipGetErrorString(e));
  return NIDREG_OK;
}

/* vlcal::estimate_camera_fov (src/vlcal/common/estimate_fov.cpp:17-51) on the host: for each of the pixels (0, 0), (W/2, 0),
 * (0, H/2) the bearing that projects onto it, found by NelderMead<2> (include/dfo/nelder_mead.hpp:32-113, defaults) over two
 * rotation angles, then the largest angle to the optical axis.  ~240 projections of ONE point: host work in the reference
 * and here (the device's scalar projection code compiled for the host, project_host) -- through Python and the GPU it was
 * 8 of the 13 ms a whole configs[0] calibration took. */
int nidreg_estimate_camera_fov(int model_id, const double* intrinsics, const double* distortion, int width, int height, double* max_fov) {
  if (model_id < 0 || model_id > 5 || !intrinsics || !distortion || !max_fov) return fail(NIDREG_ERR_INVALID, "nidreg_estimate_camera_fov: bad argument");
  double intr5[5], dist8[8];
  std::memcpy(intr5, intrinsics, sizeof(intr5));
  std::memcpy(dist8, distortion, sizeof(dist8));
  // AngleAxis(x0, X) * AngleAxis(x1, Y) * UnitZ through quaternions, as Eigen evaluates it (estimate_fov.cpp:19-21)
  auto to_dir = [](const double* x, double* d) {
    const double aw = std::cos(0.5 * x[0]), ax = std::sin(0.5 * x[0]);
    const double bw = std::cos(0.5 * x[1]), by = std::sin(0.5 * x[1]);
    const double qw = aw * bw, qx = ax * bw, qy = aw * by, qz = ax * by;
    const double ux = 2.0 * qy, uy = -2.0 * qx, uz = 0.0;  // 2 (vec x ez)
    d[0] = qw * ux + (qy * uz - qz * uy);
    d[1] = qw * uy + (qz * ux - qx * uz);
    d[2] = (1.0 + qw * uz) + (qx * uy - qy * ux);
  };
  const double corners[3][2] = {{0.0, 0.0}, {double(width / 2), 0.0}, {0.0, double(height / 2)}};
  double best = 0.0;
  for (int c = 0; c < 3; c++) {
    const double pu = corners[c][0], pv = corners[c][1];
    auto f = [&](const double* x) {
      double d[3], uv[2];
      to_dir(x, d);
      if (project_host(model_id, intr5, dist8, d, 1, uv, nullptr) != 0) return std::numeric_limits<double>::max();
      const double e = (pu - uv[0]) * (pu - uv[0]) + (pv - uv[1]) * (pv - uv[1]);
      return std::isfinite(e) ? e : std::numeric_limits<double>::max();
    };
    // NelderMead<2>: rows (y, x0, x1), init_step 0.1, (alpha, gamma, rho) = (1, 2, 0.5), 1024 iterations, variance threshold 1e-5
    std::array<std::array<double, 3>, 3> x;
    for (int i = 0; i < 3; i++) {
      x[size_t(i)] = {0.0, 0.0, 0.0};
      if (i > 0) x[size_t(i)][size_t(i)] += 0.1;
      x[size_t(i)][0] = f(&x[size_t(i)][1]);
    }
    for (int it = 0; it < 1024; it++) {
      std::stable_sort(x.begin(), x.end(), [](const std::array<double, 3>& a, const std::array<double, 3>& b) { return a[0] < b[0]; });
      double var = 0.0;
      for (int k = 1; k < 3; k++) {
        const double m = ((x[0][size_t(k)] + x[1][size_t(k)]) + x[2][size_t(k)]) / 3.0;
        double v = 0.0;
        for (int i = 0; i < 3; i++) v += (x[size_t(i)][size_t(k)] - m) * (x[size_t(i)][size_t(k)] - m);
        var += v;
      }
      if (var < 1e-5) break;
      std::array<double, 3> xo, xr;
      for (int k = 1; k < 3; k++) xo[size_t(k)] = (x[0][size_t(k)] + x[1][size_t(k)]) / 2.0;
      xo[0] = f(&xo[1]);
      for (int k = 1; k < 3; k++) xr[size_t(k)] = xo[size_t(k)] + 1.0 * (xo[size_t(k)] - x[2][size_t(k)]);
      xr[0] = f(&xr[1]);
      if (x[0][0] <= xr[0] && xr[0] < x[1][0]) {
        x[2] = xr;
      } else if (xr[0] < x[0][0]) {
        std::array<double, 3> xe;
        for (int k = 1; k < 3; k++) xe[size_t(k)] = xo[size_t(k)] + 2.0 * (xo[size_t(k)] - x[2][size_t(k)]);
        xe[0] = f(&xe[1]);
        x[2] = xe[0] < xr[0] ? xe : xr;
      } else {
        std::array<double, 3> xc;
        for (int k = 1; k < 3; k++) xc[size_t(k)] = xo[size_t(k)] + 0.5 * (xo[size_t(k)] - x[2][size_t(k)]);
        xc[0] = f(&xc[1]);
        if (xc[0] < x[2][0]) {
          x[2] = xc;
        } else {
          for (int j = 1; j < 3; j++) {
            for (int k = 1; k < 3; k++) x[size_t(j)][size_t(k)] = x[0][size_t(k)] + 0.5 * (x[size_t(j)][size_t(k)] - x[0][size_t(k)]);
            x[size_t(j)][0] = f(&x[size_t(j)][1]);
          }
        }
      }
    }
    // result.x = x[0] of the LAST SORT INSIDE the loop (nelder_mead.hpp:97-98): after 1024 iterations without convergence the
    // reference does not sort again, and neither does this
    double d[3];
    to_dir(&x[0][1], d);
    const double n = std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    const double fov = std::acos(n > 0.0 ? d[2] / n : d[2]);
    if (fov > best) best = fov;
  }
  *max_fov = best;
  return NIDREG_OK;
}

int nidreg_project_model(int model_id, const double* intrinsics, const double* distortion, int device_id, int precision, const double* p3, int64_t n, double* uv, double* jac) {
  if (model_id < 0 || model_id > 5 || !intrinsics || !distortion || !p3 || !uv || n < 0) return fail(NIDREG_ERR_INVALID, "nidreg_project_model: bad argument");
  if (device_id == NIDREG_DEVICE_HOST) {  // the device's scalar projection code compiled for the host: no GPU involved (fp64 whatever `precision` says)
    double intr5[5], dist8[8];
    std::memcpy(intr5, intrinsics, sizeof(intr5));
    std::memcpy(dist8, distortion, sizeof(dist8));
    return project_host(model_id, intr5, dist8, p3, n, uv, jac) == 0 ? NIDREG_OK : fail(NIDREG_ERR_INVALID, "nidreg_project_model: unknown camera model");
  }
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
  if (device_id < 0 || device_id >= ndev) return fail(NIDREG_ERR_INVALID, "nidreg_view_culling: device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  const int64_t stride = point_stride > 0 ? point_stride : 32;
  if (stride % 8 != 0 || stride < 32) return fail(NIDREG_ERR_INVALID, "nidreg_view_culling: point_stride must be a multiple of 8 and >= 32 ((x y z 1) doubles)");
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

int nidreg_rccl_unique_id(unsigned char* id128) {
  if (!id128) return fail(NIDREG_ERR_INVALID, "nidreg_rccl_unique_id: null argument");
  RcclApi* api = rccl_api();
  if (!api->lib || !api->error.empty()) return fail(NIDREG_ERR_HIP, "nidreg_rccl_unique_id: " + api->error);
  static_assert(sizeof(ncclUniqueId) == NIDREG_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  RCCL_TRY(api->GetUniqueId(&id));
  std::memcpy(id128, &id, sizeof(id));
  return NIDREG_OK;
}

int nidreg_shard_comm_init(nidreg_handle* h, int world_size, int rank, const unsigned char* id128) {
  int rc = rccl_attachable(h, "nidreg_shard_comm_init");
  if (rc) return rc;
  if (!id128 || world_size < 1 || rank < 0 || rank >= world_size) return fail(NIDREG_ERR_INVALID, "nidreg_shard_comm_init: bad argument");
  RcclApi* api = rccl_api();
  if (!api->lib || !api->error.empty()) return fail(NIDREG_ERR_HIP, "nidreg_shard_comm_init: " + api->error);
  HIP_TRY(hipSetDevice(h->device));
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclComm_t comm = nullptr;
  RCCL_TRY(api->CommInitRank(&comm, world_size, id, rank));
  rc = rccl_check_agreement(h, comm, "nidreg_shard_comm_init");
  if (rc) {
    if (api->CommDestroy) (void)api->CommDestroy(comm);
    return rc;
  }
  rccl_release(h);
  cohort_leave(h);  // (NIDREG_COHORT=1: a handle that evaluates collectively is nobody's sibling on this GPU)
  drop_groups_of(h);
  h->rccl_comm = comm;
  h->rccl_owned = true;
  return NIDREG_OK;
}

int nidreg_shard_attach_rccl(nidreg_handle* h, void* nccl_comm) {
  int rc = rccl_attachable(h, "nidreg_shard_attach_rccl");
  if (rc) return rc;
  if (!nccl_comm) {  // detach: the handle evaluates on its own again
    rccl_release(h);
    return NIDREG_OK;
  }
  RcclApi* api = rccl_api();
  if (!api->lib || !api->error.empty()) return fail(NIDREG_ERR_HIP, "nidreg_shard_attach_rccl: " + api->error);
  int count = 0;
  RCCL_TRY(api->CommCount(static_cast<ncclComm_t>(nccl_comm), &count));  // (also rejects a pointer that is not a communicator of this RCCL)
  HIP_TRY(hipSetDevice(h->device));
  rc = rccl_check_agreement(h, static_cast<ncclComm_t>(nccl_comm), "nidreg_shard_attach_rccl");
  if (rc) return rc;
  rccl_release(h);
  cohort_leave(h);
  drop_groups_of(h);
  h->rccl_comm = nccl_comm;
  h->rccl_owned = false;
  return NIDREG_OK;
}

int nidreg_shard_hist(nidreg_handle* h, const double* se3) {
  if (!h || !se3) return fail(NIDREG_ERR_INVALID, "nidreg_shard_hist: null argument");
  if (h->mode != NIDREG_MODE_SPLINE) return fail(NIDREG_ERR_INVALID, "nidreg_shard_hist: SPLINE handles only");
  if (h->set || h->is_shard) return fail(NIDREG_ERR_INVALID, "nidreg_shard_hist: the handle is already sharded inside the library (desc.device_ids / NIDREG_DEVICES)");
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
  h->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);  // 1: per-kernel events (three-kernel path); 2: the whole evaluation, whichever path runs
  return NIDREG_OK;
}

int nidreg_get_timing(nidreg_handle* h, float* ms6) {
  if (!h || !ms6) return fail(NIDREG_ERR_INVALID, "nidreg_get_timing: null argument");
  if (!h->timing) return fail(NIDREG_ERR_INVALID, "nidreg_get_timing: timing not enabled");
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipEventSynchronize(h->ev[5]));
  HIP_TRY(hipEventElapsedTime(&ms6[0], h->ev[0], h->ev[5]));
  for (int k = 0; k < 5; k++) ms6[1 + k] = 0.f;
  if (h->timing == 1)
    for (int k = 0; k < 5; k++) HIP_TRY(hipEventElapsedTime(&ms6[1 + k], h->ev[k], h->ev[k + 1]));
  return NIDREG_OK;
}

int nidreg_num_shards(nidreg_handle* h) {
  if (!h) return 0;
  return h->set ? int(h->set->shards.size()) : 1;
}

int nidreg_shard_devices(nidreg_handle* h, int* device_ids, int capacity) {
  if (!h || !device_ids) return fail(NIDREG_ERR_INVALID, "nidreg_shard_devices: null argument");
  if (!h->set) {
    if (capacity > 0) device_ids[0] = h->device;
    return 1;
  }
  const int n = int(h->set->shards.size());
  for (int g = 0; g < n && g < capacity; g++) device_ids[g] = h->set->shards[size_t(g)]->device;
  return n;
}

/* test hook (tests/test_host_logic.py; not part of the drop-in surface): the column-group partition a pair spread over n
 * GPUs uses -- gcount[NG + 1] record offsets of the column groups -> cut[n + 1] group boundaries */
int nidreg_debug_partition_groups(const int64_t* gcount, int NG, int n, int* cut_out) {
  if (!gcount || !cut_out || NG < 1 || n < 1) return NIDREG_ERR_INVALID;
  const std::vector<int64_t> g(gcount, gcount + NG + 1);
  const std::vector<int> cut = partition_groups(g, NG, n);
  for (int k = 0; k <= n; k++) cut_out[k] = cut[size_t(k)];
  return NIDREG_OK;
}

/* test hook (tests/test_host_logic.py; not part of the drop-in surface): the chunk table split_groups builds for a share
 * `target` of a round, a per-segment cost of `overhead` records and at most max_segs segments per chunk -- gcount[NG + 1] record offsets of the column groups ->
 * up to cap rows {start, count, group, pad};
 * returns the number of chunks (also when it exceeds cap) */
int nidreg_debug_chunk_table(const int64_t* gcount, int NG, int target, int overhead, int max_segs, int pair, uint32_t* rows_out, int cap) {
  if (!gcount || NG < 1 || overhead < 0 || max_segs < 1) return NIDREG_ERR_INVALID;
  std::vector<Chunk> chunks;
  split_groups(gcount, NG, target, overhead, max_segs, pair, chunks);
  for (size_t k = 0; k < chunks.size() && int(k) < cap && rows_out; k++) {
    rows_out[4 * k] = chunks[k].start, rows_out[4 * k + 1] = chunks[k].count, rows_out[4 * k + 2] = chunks[k].group, rows_out[4 * k + 3] = chunks[k].pad;
  }
  return int(chunks.size());
}

/* test hook (tests/test_host_logic.py; not part of the drop-in surface): the number of chunks a handle's own table of a pass
 * gets -- round_chunks (the square-root rule) snapped to whole multiples of the non-empty column groups, as create_impl does */
int nidreg_debug_round_chunks(int per_cu, int num_cus, const int64_t* gcount, int NG) {
  if (!gcount || NG < 1 || per_cu < 1 || num_cus < 1) return NIDREG_ERR_INVALID;
  return int(snap_to_groups(round_chunks(per_cu, num_cus, gcount[NG] - gcount[0]), gcount, NG, int64_t(per_cu) * num_cus));
}

void nidreg_trim(void) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) return;
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (int dev = 0; dev < ndev && dev < 64; dev++) {
    ScratchArena& a = ScratchArena::of(dev);
    std::lock_guard<ScratchArena> guard(a);
    (void)hipSetDevice(dev);
    a.release();
    if (dev < NIDREG_MAX_DEVICES) pool_release(dev);
  }
  (void)hipSetDevice(cur);
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
  if (h->set) {
    info8[6] = 0;
    for (nidreg_handle* sh : h->set->shards) info8[6] += sh->num_points;
  }
  const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  const int nfast = h->mode == NIDREG_MODE_NEAREST ? nearest_fast_args(h, ident).on : 0;
  info8[7] = (h->rec64 ? 0 : 1) | (h->seg ? 2 : 0) | (h->seg_hist ? 4 : 0) | (nfast ? 8 : 0) | (grad_sums_table(h) ? 16 : 0) | (int64_t(1 << h->cshift) << 8) |
             (fused_planned(h) ? (int64_t(32) | (int64_t(h->nchunks) << 16) | (int64_t(h->fused_full) << 28)) : 0);
  return NIDREG_OK;
}

}  // extern "C"
