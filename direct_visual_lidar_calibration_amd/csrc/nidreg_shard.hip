#define NID_COMMON_KERNELS
#define NID_SHARD_KERNELS
#include "nidreg_internal.hpp"

namespace nidreg_detail {

// ---- sharded pairs ----------------------------------------------------------------------------------------------

// NIDREG_DEVICES="0,1,2,3": spread every SPLINE / NEAREST handle over these devices without touching the caller -- this is
// how the reference's unchanged `new NIDCost(proj, image, points, bins)` (visual_camera_calibration.cpp:206) uses all GPUs
// of a node for a one-bag dataset
std::vector<int> shard_devices(const nidreg_desc* d) {
  std::vector<int> ids;
  if (d->num_devices > 1) {
    for (int i = 0; i < d->num_devices && i < NIDREG_MAX_DEVICES; i++) ids.push_back(d->device_ids[i]);
    return ids;
  }
  if (d->num_devices == 1) return ids;  // explicit single device
  if (const char* env = std::getenv("NIDREG_DEVICES")) {
    const char* p = env;
    while (*p) {
      char* end = nullptr;
      const long v = std::strtol(p, &end, 10);
      if (end == p) break;
      ids.push_back(int(v));
      p = (*end == ',') ? end + 1 : end;
      if (*end != ',' && *end != 0) break;
    }
    if (ids.size() < 2) ids.clear();
  }
  return ids;
}
bool wants_shards(const nidreg_desc* d) { return !d->ext_hist && !d->ext_out && !d->ext_stream && !(d->flags & NIDREG_FLAG_EXT_STREAM) && shard_devices(d).size() > 1; }

// Sets of one process are evaluated one after the other on every device they share: an in-kernel wait of set X must never
// sit in a hardware queue behind a kernel of set Y that waits for X on another device (the reference calls the pairs of a
// multi-bag dataset from an OpenMP loop, visual_camera_calibration.cpp:161 -- with NIDREG_DEVICES every one of them is a
// set).  One mutex per device, taken in ascending device order.
std::mutex g_shard_device_mu[NIDREG_MAX_DEVICES];

// one shard's launches of one evaluation, phase by phase: 0 histogram; 1 k_entropy_repl (shards on devices of their own: PUSH | REDUCE
// in one launch; shards that share a device: PUSH only); 2 REDUCE where phase 1 only pushed; 3 gradient
int shard_launch_phase(ShardSet* set, int g, int phase, bool alone) {
  nidreg_handle* h = set->shards[size_t(g)];
  HIP_TRY(hipSetDevice(h->device));
  const bool grad = set->job_mode == NIDREG_MODE_SPLINE && set->job_grad;
  const bool grad_runs_tail = grad && h->nchunks > 0;  // (a shard without points has no gradient workgroup to run the entropy tail)
  if (phase == 0) {
    bump_seq(h);
    h->h_out[10] = 0.0;
    if (h->timing) HIP_TRY(hipEventRecord(h->ev[0], h->stream));
    if (h->nchunks == 0) {  // no points in this shard's columns: no histogram kernel runs, its columns of the replica stay zero
      HIP_TRY(begin_histogram(h));
      if (set->job_mode == NIDREG_MODE_SPLINE) {
        for (int k = 0; k < 4; k++) h->last_q[k] = set->job_pose[k];
        pose_from_se3(set->job_pose, h->last_R, h->last_t);
      }
    } else {
      const int rc = set->job_mode == NIDREG_MODE_SPLINE ? launch_hist_spline(h, set->job_pose, alone) : launch_hist_nearest(h, set->job_pose);
      if (rc) return rc;
    }
    if (h->timing) HIP_TRY(hipEventRecord(h->ev[2], h->stream));
    return NIDREG_OK;
  }
  if (phase == 1 || phase == 2) {
    const int role = set->colocated ? (phase == 1 ? SHARD_PUSH : SHARD_REDUCE) : (SHARD_PUSH | SHARD_REDUCE);
    if (phase == 2 && !set->colocated) return NIDREG_OK;
    const bool reduces = (role & SHARD_REDUCE) != 0;
    hipLaunchKernelGGL(k_entropy_repl, dim3(set->nblocks), dim3(kEntropyThreads), 0, h->stream, h->d_hist, h->bins, 1.0 / fixed_unit(h), h->d_shard_tab, set->seq, h->hist_cur, role, h->d_phi_q,
                       h->d_hist_image, h->d_hist_points, h->d_scal, h->d_out, h->d_out_host, grad ? 0.0 : h->seq, h->d_counters, reduces ? h->d_hist_buf[h->hist_cur ^ 1] : nullptr, h->hist_words,
                       grad_runs_tail ? 0 : 1, h->d_out + 10, h->d_out_host ? h->d_out_host + 10 : nullptr, set->timeout_ticks);
    HIP_TRY(hipGetLastError());
    if (reduces) {
      h->hist_zeroed[h->hist_cur ^ 1] = true;
      h->zero_stream = h->stream;
      if (h->timing) HIP_TRY(hipEventRecord(h->ev[3], h->stream));
    }
    return NIDREG_OK;
  }
  if (grad) {
    const int rc = launch_grad(h, alone, grad_runs_tail ? 1 : 0);  // records ev[4]; an empty shard finalises zeros stand-alone
    if (rc) return rc;
  } else if (h->timing) {
    HIP_TRY(hipEventRecord(h->ev[4], h->stream));
  }
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[5], h->stream));
  h->ev_grad = grad;
  return NIDREG_OK;
}

int shard_finish(ShardSet* set, int g) {
  nidreg_handle* h = set->shards[size_t(g)];
  const bool grad = set->job_mode == NIDREG_MODE_SPLINE && set->job_grad;
  std::array<double, 8>& r = set->res[size_t(g)];
  const int rc = eval_finish(h, &r[0], grad ? &r[1] : nullptr);
  if (rc >= 0 && h->h_out[10] != 0.0) return fail(NIDREG_ERR_HIP, "sharded evaluation: timed out waiting for a peer GPU's partials");
  return rc;
}

// one shard's part of one evaluation: launches + completion wait; called concurrently for different shards (one shard per device)
int run_shard(ShardSet* set, int g) {
  InflightGuard guard(set->shards[size_t(g)]->device);
  for (int phase = 0; phase < 4; phase++) {
    const int rc = shard_launch_phase(set, g, phase, guard.alone);
    if (rc) return rc;
  }
  return shard_finish(set, g);
}

void shard_worker(ShardSet* set, int g) {
  uint64_t seen = 0;
  for (;;) {
    // wait for the next generation: a few hundred microseconds of paused spinning (an optimiser calls back to back),
    // then sleep
    unsigned spins = 0;
    while (set->gen.load(std::memory_order_acquire) == seen && !set->stop.load(std::memory_order_acquire)) {
      if (++spins < 20000) {
        for (int k = 0; k < 8; k++) __builtin_ia32_pause();
      } else {
        std::unique_lock<std::mutex> lk(set->mu);
        set->sleepers.fetch_add(1);
        set->cv.wait(lk, [&] { return set->gen.load(std::memory_order_acquire) != seen || set->stop.load(std::memory_order_acquire); });
        set->sleepers.fetch_sub(1);
      }
    }
    if (set->stop.load(std::memory_order_acquire)) return;
    seen = set->gen.load(std::memory_order_acquire);
    set->rc[size_t(g)] = run_shard(set, g);
    set->pending.fetch_sub(1, std::memory_order_release);
  }
}

// the whole set: one evaluation.  mode SPLINE: pose = se3[7]; NEAREST: pose = row-major 4x4
int set_eval(ShardSet* set, int mode, const double* pose, double* cost, double* grad7) {
  const int n = int(set->shards.size());
  if (set->shards[0]->mode != mode) return fail(NIDREG_ERR_INVALID, mode == NIDREG_MODE_SPLINE ? "nidreg_eval: handle was created in NEAREST mode" : "nidreg_eval_iso: handle was created in SPLINE mode");
  if (set->poisoned) return fail(NIDREG_ERR_HIP, "sharded handle: an earlier evaluation failed half way; destroy and re-create the handle");
  // one set at a time per device (see g_shard_device_mu)
  struct Unlock {
    const std::vector<int>& devs;
    ~Unlock() {
      for (size_t i = devs.size(); i-- > 0;) g_shard_device_mu[devs[i]].unlock();
    }
  };
  for (int dev : set->lock_devices) g_shard_device_mu[dev].lock();
  Unlock unlock{set->lock_devices};
  set->seq++;
  set->job_mode = mode;
  set->job_grad = grad7 != nullptr;
  std::memcpy(set->job_pose, pose, (mode == NIDREG_MODE_SPLINE ? 7 : 16) * sizeof(double));
  if (set->colocated) {
    // several shards on one device (a test configuration): their streams may share an in-order hardware queue, so the
    // kernels are launched by this thread phase by phase -- every in-kernel wait then targets a kernel that sits AHEAD of
    // it in whatever queue they share
    for (int phase = 0; phase < 4; phase++)
      for (int g = 0; g < n; g++) {
        set->rc[size_t(g)] = shard_launch_phase(set, g, phase, false);
        if (set->rc[size_t(g)] < 0) {
          set->poisoned = true;
          return fail(set->rc[size_t(g)], "sharded evaluation: launch failed on shard " + std::to_string(g) + ": " + g_last_error);
        }
      }
    for (int g = 0; g < n; g++) set->rc[size_t(g)] = shard_finish(set, g);
  } else {
    set->pending.store(n - 1, std::memory_order_relaxed);
    set->gen.fetch_add(1, std::memory_order_release);
    if (set->sleepers.load() > 0) {
      std::lock_guard<std::mutex> lk(set->mu);
      set->cv.notify_all();
    }
    set->rc[0] = run_shard(set, 0);
    // the other shards finish within microseconds of this one: pause-spin (2 ms by the clock), then nap
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    unsigned spins = 0;
    while (set->pending.load(std::memory_order_acquire) > 0) {
      __builtin_ia32_pause();
      if ((++spins & 0xffu) == 0) {
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3 > 2000.0) {
          struct timespec ts = {0, 50000};
          nanosleep(&ts, nullptr);
        }
      }
    }
  }
  bool all_ok = true;
  for (int g = 0; g < n; g++) {
    if (set->rc[size_t(g)] < 0) {
      set->poisoned = true;
      return fail(set->rc[size_t(g)], "sharded evaluation failed on shard " + std::to_string(g) + " (device " + std::to_string(set->shards[size_t(g)]->device) + ")");
    }
    if (set->rc[size_t(g)] == NIDREG_FALSE) all_ok = false;
  }
  // every shard computed the cost from the same gathered integers: any difference means a shard read stale or torn data
  for (int g = 1; g < n; g++) {
    if (std::memcmp(&set->res[size_t(g)][0], &set->res[0][0], sizeof(double)) != 0) {
      set->poisoned = true;
      return fail(NIDREG_ERR_HIP, "sharded evaluation: shard " + std::to_string(g) + " (device " + std::to_string(set->shards[size_t(g)]->device) +
                                    ") computed a different cost than shard 0 from the gathered partials -- cross-device visibility failure");
    }
  }
  if (cost) *cost = set->res[0][0];
  if (grad7) {
    for (int k = 0; k < 7; k++) {
      double t = 0.0;
      for (int g = 0; g < n; g++) t += set->res[size_t(g)][size_t(1 + k)];  // fixed order: run-to-run reproducible
      grad7[k] = t;
    }
  }
  return all_ok ? NIDREG_OK : NIDREG_FALSE;
}

void free_shard_set(ShardSet* set) {
  if (!set) return;
  set->stop.store(true, std::memory_order_release);
  {
    std::lock_guard<std::mutex> lk(set->mu);
    set->cv.notify_all();
  }
  for (auto& t : set->workers)
    if (t.joinable()) t.join();
  for (size_t g = 0; g < set->shards.size(); g++) {
    nidreg_handle* h = set->shards[g];
    if (h) {
      (void)hipSetDevice(h->device);
      if (h->stream) (void)hipStreamSynchronize(h->stream);
    }
  }
  for (size_t g = 0; g < set->shards.size(); g++) {
    if (!set->shards[g]) continue;
    (void)hipSetDevice(set->shards[g]->device);
    if (g < set->flags.size() && set->flags[g]) (void)hipFree(set->flags[g]);
    if (g < set->gather.size() && set->gather[g]) (void)hipFree(set->gather[g]);
  }
  for (size_t g = 1; g < set->shards.size(); g++) free_handle(set->shards[g]);
  delete set;
}

// Cut the NG column groups into n contiguous ranges with (nearly) equal point counts: boundary g is the first group index
// at which the running count reaches g / n of the total (the intensities are rank-equalised upstream, preprocess.cpp:464-473,
// so the groups are close to uniform; view culling skews them a little).  Ranges may be empty when n > NG.
std::vector<int> partition_groups(const std::vector<int64_t>& gcount, int NG, int n) {
  std::vector<int> cut(static_cast<size_t>(n) + 1, 0);
  const int64_t total = gcount[size_t(NG)];
  cut[size_t(n)] = NG;
  int g = 0;
  for (int k = 1; k < n; k++) {
    const int64_t want = total * k / n;
    while (g < NG && gcount[size_t(g) + 1] <= want) g++;
    // group g straddles the target: cut on the nearer side
    if (g < NG && want - gcount[size_t(g)] > gcount[size_t(g) + 1] - want) g++;
    g = std::max(g, cut[size_t(k) - 1]);
    cut[size_t(k)] = std::min(g, NG);
  }
  if (total == 0)  // nothing to balance: equal column ranges
    for (int k = 1; k < n; k++) cut[size_t(k)] = NG * k / n;
  return cut;
}

int create_sharded(const nidreg_desc* d, const nidreg_cloud* cloud, const double* T_cull, double min_z, int enable_depth, nidreg_handle** out) {
  *out = nullptr;
  const std::vector<int> ids = shard_devices(d);
  const int n = int(ids.size());
  if (n > kMaxShards) return fail(NIDREG_ERR_INVALID, "nidreg_create: at most 16 shards");
  if (!cloud && (d->num_points < 0 || (d->num_points > 0 && (!d->points || !d->intensities)))) return fail(NIDREG_ERR_INVALID, "nidreg_create: null points");
  if (d->bins < 2 || d->bins > kMaxWideBins) return fail(NIDREG_ERR_INVALID, "nidreg_create: bins must be in [2, " + std::to_string(kMaxWideBins) + "]");
  if (d->image_dtype != NIDREG_IMAGE_F64 && d->image_dtype != NIDREG_IMAGE_U8) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad image_dtype");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NIDREG_ERR_NO_DEVICE, "nidreg_create: no HIP device (the NID core has no CPU path)");
  for (int id : ids)
    if (id < 0 || id >= ndev) return fail(NIDREG_ERR_INVALID, "nidreg_create: device id " + std::to_string(id) + " out of range (NIDREG_DEVICES / desc.device_ids)");
  // peer mappings, both directions, before any buffer is allocated
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) {
      if (ids[size_t(i)] == ids[size_t(j)]) continue;
      int can = 0;
      HIP_TRY(hipDeviceCanAccessPeer(&can, ids[size_t(i)], ids[size_t(j)]));
      if (!can) return fail(NIDREG_ERR_HIP, "nidreg_create: device " + std::to_string(ids[size_t(i)]) + " cannot map the memory of device " + std::to_string(ids[size_t(j)]));
      HIP_TRY(hipSetDevice(ids[size_t(i)]));
      const hipError_t e = hipDeviceEnablePeerAccess(ids[size_t(j)], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(NIDREG_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
      (void)hipGetLastError();
    }
  }
  // ---- the master: the complete pair on the owner device (where the cloud lives / the first listed device): upload,
  // [ViewCulling::cull,] bucketing by column group, Morton sort, gather, bin image -- once; the shards then take their
  // column groups' records (already in their final order) device to device
  WideBins wide;  // bins > 256: the shards run on the occupied bins, compacted (resolve_wide_bins)
  if (d->bins > NIDREG_MAX_BINS) {
    const int rc = resolve_wide_bins(d, cloud, wide);
    if (rc) return rc;
  }
  const int B = wide.user_bins ? wide.compact_bins : d->bins;
  nidreg_desc md = *d;
  md.num_devices = 1;
  md.device_id = cloud ? cloud->device : ids[0];
  if (md.columns_per_group <= 0) md.columns_per_group = std::max(1, std::min(std::max(1, 256 / B), B / (4 * n)));  // >= 4 column groups per shard where B allows
  md.scale_points = std::max<int64_t>(cloud ? cloud->n : d->num_points, d->scale_points);  // the fixed-point unit of the unsharded handle
  nidreg_handle* master = nullptr;
  {
    CreateOpts mo;
    if (wide.user_bins) mo.wide = &wide;
    const int rc = create_impl(&md, cloud, T_cull, min_z, enable_depth, mo, &master);
    if (rc) return rc;
  }
  md.bins = B;  // (the shards below are created with the compact count)
  // every cut between shards is a whole number of k_entropy_repl column blocks: CB columns per block, the largest of 8, 4, 2, 1
  // that tiles with the column groups (GW columns each) and still leaves every shard a cut unit of its own
  const int GWm = master->GW, NGm = master->NG;
  int CB = 1, unit = 1;  // unit = column groups per cut unit
  for (int cb = kEntropyCols; cb >= 1; cb /= 2) {
    if (cb % GWm != 0 && GWm % cb != 0) continue;
    const int u = std::max(1, cb / GWm);
    if (cb > 1 && (NGm + u - 1) / u < n) continue;
    CB = cb;
    unit = u;
    break;
  }
  std::vector<int> cut(size_t(n) + 1, 0);
  {
    const int NU = (NGm + unit - 1) / unit;
    std::vector<int64_t> ucount(size_t(NU) + 1, 0);
    for (int u = 0; u <= NU; u++) ucount[size_t(u)] = master->gcount[size_t(std::min(NGm, u * unit))];
    const std::vector<int> ucut = partition_groups(ucount, NU, n);
    for (int k = 0; k <= n; k++) cut[size_t(k)] = std::min(NGm, ucut[size_t(k)] * unit);
  }

  ShardSet* set = new ShardSet();
  set->CB = CB;
  set->nblocks = (B + CB - 1) / CB;
  set->shards.assign(size_t(n), nullptr);
  set->flags.assign(size_t(n), nullptr);
  set->gather.assign(size_t(n), nullptr);
  set->rc.assign(size_t(n), 0);
  set->res.assign(size_t(n), std::array<double, 8>());
  set->lock_devices = ids;
  std::sort(set->lock_devices.begin(), set->lock_devices.end());
  set->lock_devices.erase(std::unique(set->lock_devices.begin(), set->lock_devices.end()), set->lock_devices.end());
  set->colocated = set->lock_devices.size() != ids.size();
  // measurement knob (tools/shard_cost.py): co-located shards driven by their worker threads like shards on different
  // devices -- only valid when the caller has made sure their streams do not share a hardware queue (GPU_MAX_HW_QUEUES)
  if (set->colocated && std::getenv("NIDREG_SHARD_COLOCATED_WORKERS")) set->colocated = false;
  if (const char* t = std::getenv("NIDREG_SHARD_TIMEOUT_MS")) set->timeout_ticks = 100000ull * (unsigned long long)std::max(1L, std::strtol(t, nullptr, 10));
  auto bail = [&](int rc) {
    const std::string msg = g_last_error;
    free_handle(master);
    nidreg_handle* lead = set->shards[0];
    if (lead) {
      lead->set = set;
      free_handle(lead);  // frees the set (and through it the other shards)
    } else {
      free_shard_set(set);
    }
    g_last_error = msg;
    return rc;
  };
  {
    std::vector<std::thread> th;
    std::vector<int> rcs(size_t(n), 0);
    std::vector<std::string> errs(static_cast<size_t>(n));
    for (int g = 0; g < n; g++) {
      th.emplace_back([&, g]() {
        nidreg_desc sd = md;
        sd.device_id = ids[size_t(g)];
        sd.columns_per_group = master->GW;
        CreateOpts o;
        o.shard = true;
        o.master = master;
        o.group_lo = cut[size_t(g)];
        o.group_hi = cut[size_t(g) + 1];
        rcs[size_t(g)] = create_impl(&sd, nullptr, nullptr, 0.0, 0, o, &set->shards[size_t(g)]);
        if (rcs[size_t(g)]) errs[size_t(g)] = g_last_error;
      });
    }
    for (auto& t : th) t.join();
    for (int g = 0; g < n; g++)
      if (rcs[size_t(g)]) return bail(fail(rcs[size_t(g)], "shard " + std::to_string(g) + ": " + errs[size_t(g)]));
  }
  // flag and gather blocks: fine-grained (coherent) device memory, mapped into every peer
  const size_t gw = size_t(kGatherWords);
  for (int g = 0; g < n; g++) {
    nidreg_handle* h = set->shards[size_t(g)];
    h->shard_index = g;
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipExtMallocWithFlags(reinterpret_cast<void**>(&set->flags[size_t(g)]), kFlagWords * sizeof(u64), hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(set->flags[size_t(g)], 0, kFlagWords * sizeof(u64));
    if (e == hipSuccess) e = hipExtMallocWithFlags(reinterpret_cast<void**>(&set->gather[size_t(g)]), gw * sizeof(u64), hipDeviceMallocFinegrained);
    if (e == hipSuccess) e = hipMemset(set->gather[size_t(g)], 0, gw * sizeof(u64));
    if (e != hipSuccess) return bail(fail(NIDREG_ERR_HIP, std::string("sharded handle: flag / gather block: ") + hipGetErrorString(e)));
  }
  for (int g = 0; g < n; g++) {
    nidreg_handle* h = set->shards[size_t(g)];
    ShardTable tab;
    std::memset(&tab, 0, sizeof(tab));
    for (int p = 0; p < n; p++) {
      tab.flags[p] = set->flags[size_t(p)];
      tab.gather[p] = set->gather[size_t(p)];
      tab.hist[p][0] = set->shards[size_t(p)]->d_hist_buf[0];
      tab.hist[p][1] = set->shards[size_t(p)]->d_hist_buf[1];
      tab.cut[p] = set->shards[size_t(p)]->col_lo;
    }
    tab.cut[n] = B;
    tab.CB = set->CB;
    tab.n = n;
    tab.me = g;
    tab.col_lo = h->col_lo;
    tab.col_hi = h->col_hi;
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipMalloc(&h->d_shard_tab, sizeof(ShardTable));
    if (e == hipSuccess) e = hipMemcpy(h->d_shard_tab, &tab, sizeof(ShardTable), hipMemcpyHostToDevice);
    if (e != hipSuccess) return bail(fail(NIDREG_ERR_HIP, std::string("sharded handle: shard table: ") + hipGetErrorString(e)));
  }
  free_handle(master);
  master = nullptr;
  // NIDREG_SHARD_SELFTEST=1: every ordered pair of shards exchanges a flag and a payload once, now, with a short timeout -- a
  // set whose peer mappings, flag ordering or queues do not work fails HERE, with the pair named, instead of timing out in
  // the middle of the first evaluation.  Shards on the same device (a 1-GPU box exercising the protocol) are skipped: their
  // two kernels may share an in-order hardware queue.
  if (const char* st = std::getenv("NIDREG_SHARD_SELFTEST"); st && *st && *st != '0') {
    const unsigned long long ticks = 20000000ull;  // 200 ms of the 100 MHz wall clock
    std::string report;
    u64 seq = 1;
    for (int a = 0; a < n; a++)
      for (int b = 0; b < n; b++) {
        nidreg_handle *ha = set->shards[size_t(a)], *hb = set->shards[size_t(b)];
        if (a == b || ha->device == hb->device) continue;
        u64 *oa = nullptr, *ob = nullptr;  // host-mapped result words
        hipError_t e = hipHostMalloc(&oa, 2 * sizeof(u64), hipHostMallocMapped);
        if (e == hipSuccess) e = hipHostMalloc(&ob, 2 * sizeof(u64), hipHostMallocMapped);
        if (e != hipSuccess) return bail(fail(NIDREG_ERR_HIP, std::string("shard self-test: ") + hipGetErrorString(e)));
        oa[0] = oa[1] = ob[0] = ob[1] = 0;
        const u64 pattern = 0x5e1f7e5700000000ull | (u64(a) << 8) | u64(b);
        (void)hipSetDevice(hb->device);
        hipLaunchKernelGGL(k_shard_selftest_pong, dim3(1), dim3(64), 0, hb->stream, hb->d_shard_tab, a, seq, pattern, ob, ticks);
        (void)hipSetDevice(ha->device);
        hipLaunchKernelGGL(k_shard_selftest_ping, dim3(1), dim3(64), 0, ha->stream, ha->d_shard_tab, b, seq, pattern, oa, ticks);
        e = hipStreamSynchronize(ha->stream);
        (void)hipSetDevice(hb->device);
        if (e == hipSuccess) e = hipStreamSynchronize(hb->stream);
        const u64 ra = oa[0], rb = ob[0], rtt = oa[1];
        (void)hipHostFree(oa);
        (void)hipHostFree(ob);
        char line[200];
        std::snprintf(line, sizeof(line), "nidreg shard self-test: device %d -> device %d: %s%s, round trip %.1f us\n", ha->device, hb->device,
                      rb == 1 ? "flag and payload visible" : (rb == 3 ? "PAYLOAD NOT VISIBLE BEHIND THE FLAG" : "FLAG NEVER ARRIVED"), ra == 1 ? ", answer seen" : ", NO ANSWER", double(rtt) * 0.01);
        report += line;
        if (e != hipSuccess || ra != 1 || rb != 1) {
          std::fputs(report.c_str(), stderr);
          return bail(fail(NIDREG_ERR_HIP, std::string("sharded handle: self-test of the GPU-to-GPU exchange failed: ") + line));
        }
        seq++;
      }
    std::fputs(report.empty() ? "nidreg shard self-test: no pair of shards on different devices (nothing to test)\n" : report.c_str(), stderr);
    // flags and payload words back to zero: the evaluations' sequence numbers start at 1
    for (int g = 0; g < n; g++) {
      (void)hipSetDevice(set->shards[size_t(g)]->device);
      (void)hipMemset(set->flags[size_t(g)], 0, kFlagWords * sizeof(u64));
      (void)hipMemset(set->gather[size_t(g)], 0, gw * sizeof(u64));
    }
  }
  if (!set->colocated)
    for (int g = 1; g < n; g++) set->workers.emplace_back(shard_worker, set, g);
  nidreg_handle* lead = set->shards[0];
  lead->set = set;
  *out = lead;
  return NIDREG_OK;
}

}  // namespace nidreg_detail
