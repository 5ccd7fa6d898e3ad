#define NID_COMMON_KERNELS
#include "nidreg_internal.hpp"

namespace nidreg_detail {
std::mutex g_groups_mu;
std::vector<MultiGroup*> g_groups;
std::vector<std::vector<nidreg_handle*>> g_rejected;
uint64_t g_group_clock = 0;
void release_group(MultiGroup* g) { g->users.fetch_sub(1, std::memory_order_acq_rel); }

void free_group(MultiGroup* g) {
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  for (nidreg_handle* m : g->hs)  // (ADVICE r3 made the getters drain the stream of the last evaluation: never a freed group's)
  {
    if (m->last_stream == g->stream) m->last_stream = m->stream;
    if (m->zero_stream == g->stream) m->zero_stream = nullptr;  // drained above: nothing left to order against
  }
  if (g->d_table) (void)hipFree(g->d_table);
  if (g->d_chunks) (void)hipFree(g->d_chunks);
  if (g->d_chunks_hist) (void)hipFree(g->d_chunks_hist);
  if (g->stream) unpool_stream(g->device, g->stream);  // (synchronised above)
  delete g;
}
// called by free_handle: a group dies with any of its members
void drop_groups_of(const nidreg_handle* h) {
  std::vector<MultiGroup*> dead;
  {
    std::lock_guard<std::mutex> lk(g_groups_mu);
    for (size_t i = 0; i < g_groups.size();) {
      if (std::find(g_groups[i]->hs.begin(), g_groups[i]->hs.end(), h) != g_groups[i]->hs.end()) {
        dead.push_back(g_groups[i]);
        g_groups.erase(g_groups.begin() + long(i));
      } else {
        i++;
      }
    }
    for (size_t i = 0; i < g_rejected.size();) {
      if (std::find(g_rejected[i].begin(), g_rejected[i].end(), h) != g_rejected[i].end()) {
        g_rejected.erase(g_rejected.begin() + long(i));
      } else {
        i++;
      }
    }
  }
  for (MultiGroup* g : dead) {
    while (g->users.load(std::memory_order_acquire) > 0) std::this_thread::yield();  // an evaluation of another thread still runs on it
    free_group(g);
  }
}

bool groupable(const nidreg_handle* a, const nidreg_handle* b) {
  return a->device == b->device && a->model == b->model && a->mode == b->mode && a->max_fov == b->max_fov && a->precision == b->precision && a->bins == b->bins &&
         a->W == b->W && a->H == b->H && a->pitch == b->pitch && a->GW == b->GW && a->cshift == b->cshift && a->wide == b->wide && a->rec64 == b->rec64 &&
         std::memcmp(a->intr, b->intr, sizeof(a->intr)) == 0 && std::memcmp(a->dist, b->dist, sizeof(a->dist)) == 0 && a->own_hist && b->own_hist && a->d_out_host && b->d_out_host &&
         !a->set && !b->set && !a->is_shard && !b->is_shard && !a->timing && !b->timing;
}

// chunks of one pair for a share `target` of the round (same rule as create_impl's tables: split_groups)
int64_t pair_chunks(const nidreg_handle* h, int pair, int64_t target, bool wide_hist, std::vector<Chunk>& chunks) {
  return split_groups(h->gcount.data(), h->NG, target, segment_overhead(wide_hist), max_segments(h->mode, h->GW), pair, chunks);
}

// ---- cohorts (NIDREG_COHORT=1): the unchanged reference caller and ONE round of workgroups ------------------------------
// MultiNIDCost evaluates its pairs from an OpenMP loop (visual_camera_calibration.cpp:147-173): k threads, each calling its own
// NIDCost at the same pose.  Every handle's chunk tables are sized for a whole round of co-resident workgroups, so k
// concurrent callers queue k rounds of small chunks and pay every workgroup's prologue k times (8 x 1.25M points: 282 us
// against 172 us for nidreg_eval_multi's single grid).  A cohort gives the unchanged caller the
// single grid's geometry: the handles created one after the other on a device, compatible (same camera, image size, bins,
// precision), BEFORE any of them is evaluated -- visual_camera_calibration.cpp:199-208 builds all pairs' cost objects, then
// solves -- form a cohort; the first evaluation of any member seals it and rebuilds every member's chunk tables as its SHARE
// of one round (in proportion to its points, like the single grid's table).  The k callers' kernels then fill the GPU
// together.  Deterministic by construction: a member's table -- hence the order of its gradient partials -- is a function of
// the cohort (which handles were created together), never of timing; cost and histogram bits do not depend on tables at all.
// Opt-in, because a caller that evaluates the members ONE AT A TIME (OMP_NUM_THREADS=1) gets 1/k of the GPU per evaluation.
std::mutex g_cohort_mu;
Cohort* g_open_cohort[NIDREG_MAX_DEVICES];

bool cohorts_enabled() {
  const char* e = std::getenv("NIDREG_COHORT");
  return e && *e && *e != '0';
}

// the member's tables as its share of one round (called once per member, before its first evaluation, under the cohort's lock)
int cohort_reshape(nidreg_handle* h, int64_t total_points) {
  HIP_TRY(hipSetDevice(h->device));
  const int64_t mine = std::max<int64_t>(h->num_points, 1);
  // both tables are built on the host first and committed together: a member whose share does not fit keeps its own tables whole
  auto build = [&](int per_cu, bool wide_hist, std::vector<Chunk>& chunks) -> int64_t {
    const int64_t share = std::max<int64_t>(1, round_chunks(per_cu, h->num_cus, total_points) * mine / total_points);
    return split_groups(h->gcount.data(), h->NG, share, segment_overhead(wide_hist), max_segments(h->mode, h->GW), -1, chunks);
  };
  auto upload = [&](const std::vector<Chunk>& chunks, Chunk*& d_tab, size_t& cap) -> int {
    if (chunks.size() > cap) {
      Chunk* fresh = nullptr;
      HIP_TRY(hipMalloc(&fresh, chunks.size() * sizeof(Chunk)));
      if (d_tab) (void)hipFree(d_tab);
      d_tab = fresh;
      cap = chunks.size();
    }
    if (!chunks.empty()) HIP_TRY(hipMemcpy(d_tab, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
    return NIDREG_OK;
  };
  std::vector<Chunk> grad_chunks, hist_chunks;
  const int64_t slots = build(h->per_cu_grad, false, grad_chunks);
  if (slots > int64_t(h->partials_cap)) return fail(NIDREG_ERR_INVALID, "cohort: a member's share table needs more gradient-partial slots than its scratch holds");
  const bool wide = h->wide && h->d_chunks_hist;
  int64_t wslots = 0;
  if (wide) wslots = build(h->per_cu_hist, true, hist_chunks);
  // each table is committed together with the fields that describe it, right after its upload
  int rc = upload(grad_chunks, h->d_chunks, h->chunks_cap);
  if (rc) return rc;
  h->nchunks = int(grad_chunks.size());
  h->fused = -1;  // (a cohort member's table is the cohort's: the fused single launch is for handles on their own)
  h->nslots = int(slots);
  h->seg = h->nslots > h->nchunks ? 1 : 0;
  h->lds_grad = spline_grad_lds_bytes(h->bins, h->GW, h->cshift, h->seg != 0);
  h->cohort_chunks = std::move(grad_chunks);
  if (wide) {
    rc = upload(hist_chunks, h->d_chunks_hist, h->chunks_hist_cap);
    if (rc) return rc;
    h->nchunks_hist = int(hist_chunks.size());
    h->seg_hist = wslots > int64_t(h->nchunks_hist) ? 1 : 0;
    h->cohort_chunks_hist = std::move(hist_chunks);
  }
  return NIDREG_OK;
}

void cohort_seal(Cohort* c) {
  {
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->sealed.load(std::memory_order_acquire)) return;
    if (c->members.size() >= 2) {
      int64_t total = 0;
      for (const nidreg_handle* m : c->members) total += std::max<int64_t>(m->num_points, 1);
      for (nidreg_handle* m : c->members)
        if (cohort_reshape(m, total) != NIDREG_OK) std::fprintf(stderr, "nidreg: cohort member keeps its own chunk tables (%s)\n", g_last_error.c_str());
    }
    c->sealed.store(true, std::memory_order_release);
  }
  std::lock_guard<std::mutex> gl(g_cohort_mu);
  if (c->device >= 0 && c->device < NIDREG_MAX_DEVICES && g_open_cohort[c->device] == c) g_open_cohort[c->device] = nullptr;
}
void cohort_join(nidreg_handle* h) {
  if (!cohorts_enabled() || h->mode != NIDREG_MODE_SPLINE || h->set || h->is_shard || !h->own_hist || !h->d_out_host || !h->own_stream || h->device < 0 || h->device >= NIDREG_MAX_DEVICES) return;
  Cohort* to_seal = nullptr;
  {
    std::lock_guard<std::mutex> gl(g_cohort_mu);
    Cohort*& open = g_open_cohort[h->device];
    if (open) {
      std::lock_guard<std::mutex> lk(open->mu);
      if (!open->sealed.load(std::memory_order_acquire) && !open->members.empty() && open->members.size() < size_t(kMaxMulti) && groupable(open->members[0], h)) {
        open->members.push_back(h);
        h->cohort = open;
        return;
      }
    }
    to_seal = open;  // a handle of another kind ends the cohort that was forming
    open = new Cohort();
    open->device = h->device;
    open->members.push_back(h);
    h->cohort = open;
  }
  if (to_seal) cohort_seal(to_seal);
}

void cohort_leave(nidreg_handle* h) {
  Cohort* c = h->cohort;
  if (!c) return;
  h->cohort = nullptr;
  bool empty = false;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->members.erase(std::remove(c->members.begin(), c->members.end(), h), c->members.end());
    empty = c->members.empty();
  }
  if (empty) {
    {
      std::lock_guard<std::mutex> gl(g_cohort_mu);
      if (c->device >= 0 && c->device < NIDREG_MAX_DEVICES && g_open_cohort[c->device] == c) g_open_cohort[c->device] = nullptr;
    }
    delete c;
  }
}

// returns the group with its use count raised (release_group when the evaluation is over), or nullptr
MultiGroup* find_or_make_group(nidreg_handle* const* handles, int n) {
  // evicted groups are drained and freed AFTER the lock is released (free_group synchronises a stream: with the lock held every
  // concurrent caller on the device waited behind it -- ADVICE r4)
  std::vector<MultiGroup*> evicted;
  struct FreeEvicted {
    std::vector<MultiGroup*>& v;
    ~FreeEvicted() {
      for (MultiGroup* g : v) free_group(g);
    }
  } free_evicted{evicted};  // (declared before the lock: destroyed after it)
  std::lock_guard<std::mutex> lk(g_groups_mu);
  for (MultiGroup* g : g_groups)
    if (int(g->hs.size()) == n && std::equal(g->hs.begin(), g->hs.end(), handles)) {
      g->users.fetch_add(1, std::memory_order_acq_rel);
      g->last_use = ++g_group_clock;
      return g;
    }
  for (const auto& r : g_rejected)
    if (int(r.size()) == n && std::equal(r.begin(), r.end(), handles)) return nullptr;
  MultiGroup* g = new MultiGroup();
  g->hs.assign(handles, handles + n);
  g->device = handles[0]->device;
  int64_t total = 0;
  for (int i = 0; i < n; i++) total += std::max<int64_t>(handles[i]->num_points, 1);
  std::vector<std::vector<Chunk>> pair_grad(static_cast<size_t>(n)), pair_hist(static_cast<size_t>(n));
  std::vector<MultiEntry> table(static_cast<size_t>(n));
  const nidreg_handle* h0 = handles[0];
  // members of ONE sealed cohort bring their fixed share tables (cohort_reshape): a pair's chunks -- hence the order of its
  // gradient partials -- are then the same whether it is evaluated alone, in this group, or in a group of any other subset
  bool fixed_tables = true;
  for (int i = 0; i < n; i++)
    fixed_tables = fixed_tables && handles[i]->cohort && handles[i]->cohort == handles[0]->cohort && handles[i]->cohort->sealed.load(std::memory_order_acquire) &&
                   (!handles[i]->cohort_chunks.empty() || handles[i]->num_points == 0);
  for (int i = 0; i < n; i++) {
    nidreg_handle* h = handles[i];
    int64_t pair_slots = 0;
    if (fixed_tables) {
      pair_grad[size_t(i)] = h->cohort_chunks;
      for (Chunk& c : pair_grad[size_t(i)]) c.pad = (c.pad & ~0xffu) | uint32_t(i);
      pair_slots = h->nslots;
      if (h->seg) g->seg = 1;
      if (h0->wide) {
        pair_hist[size_t(i)] = h->cohort_chunks_hist;
        for (Chunk& c : pair_hist[size_t(i)]) c.pad = (c.pad & ~0xffu) | uint32_t(i);
        if (h->seg_hist) g->seg_hist = 1;
      }
    } else {
      const int64_t share_grad = std::max<int64_t>(1, round_chunks(h0->per_cu_grad, h0->num_cus, total) * std::max<int64_t>(h->num_points, 1) / total);
      pair_slots = pair_chunks(h, i, share_grad, false, pair_grad[size_t(i)]);
      if (pair_slots > int64_t(pair_grad[size_t(i)].size())) g->seg = 1;
      if (h0->wide) {
        const int64_t share_hist = std::max<int64_t>(1, round_chunks(h0->per_cu_hist, h0->num_cus, total) * std::max<int64_t>(h->num_points, 1) / total);
        if (pair_chunks(h, i, share_hist, true, pair_hist[size_t(i)]) > int64_t(pair_hist[size_t(i)].size())) g->seg_hist = 1;
      }
    }
    MultiEntry& e = table[size_t(i)];
    e.pts = h->d_pts;
    e.gend = h->d_gend;
    e.img = h->d_img;
    e.hist_buf[0] = h->d_hist_buf[0];
    e.hist_buf[1] = h->d_hist_buf[1];
    e.k16 = fixed_unit_k(h);
    e.inv_unit = 1.0 / fixed_unit(h);
    e.part_hj = h->d_part_hj;
    e.row_part = h->d_row_part;
    e.phi_q = h->d_phi_q;
    e.hist_image = h->d_hist_image;
    e.hist_points = h->d_hist_points;
    e.scal = h->d_scal;
    e.partials = h->d_partials;
    e.out = h->d_out;
    e.out_host = h->d_out_host;
    e.counters = h->d_counters;
    e.zero_words = h->hist_words;
    e.nslots = int(pair_slots);
    e.nchunks = int(pair_grad[size_t(i)].size());
    if (e.nslots > h->partials_cap) {  // the pair's partial buffer (12 doubles per slot) was sized at its creation
      delete g;
      if (g_rejected.size() >= kMaxRejected) g_rejected.erase(g_rejected.begin());
      g_rejected.emplace_back(handles, handles + n);
      return nullptr;
    }
  }
  // the pairs' chunks one after the other.  (An XCD-aware order -- workgroup b runs on XCD b mod 8, so XCD x would only see
  // the bin image of pair floor(x n / 8) -- was measured and changed nothing: 2 / 4 / 8 pairs 204 / 183 / 176 us against
  // 191 / 185 / 178 us, profiles/archive/r03g_multi_pair_patterns.jsonl: the images' L2 footprint is not what slows the group down.)
  std::vector<Chunk> chunks, wide_chunks;
  for (int i = 0; i < n; i++) chunks.insert(chunks.end(), pair_grad[size_t(i)].begin(), pair_grad[size_t(i)].end());
  for (int i = 0; i < n; i++) wide_chunks.insert(wide_chunks.end(), pair_hist[size_t(i)].begin(), pair_hist[size_t(i)].end());
  hipError_t err = hipSetDevice(g->device);
  if (err == hipSuccess) err = pool_stream(g->device, &g->stream);
  if (err == hipSuccess) err = hipMalloc(&g->d_table, table.size() * sizeof(MultiEntry));
  if (err == hipSuccess) err = hipMemcpy(g->d_table, table.data(), table.size() * sizeof(MultiEntry), hipMemcpyHostToDevice);
  if (err == hipSuccess) err = hipMalloc(&g->d_chunks, std::max<size_t>(chunks.size(), 1) * sizeof(Chunk));
  if (err == hipSuccess && !chunks.empty()) err = hipMemcpy(g->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice);
  if (err == hipSuccess && h0->wide) {
    err = hipMalloc(&g->d_chunks_hist, std::max<size_t>(wide_chunks.size(), 1) * sizeof(Chunk));
    if (err == hipSuccess && !wide_chunks.empty()) err = hipMemcpy(g->d_chunks_hist, wide_chunks.data(), wide_chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice);
  }
  if (err != hipSuccess) {
    evicted.push_back(g);  // (freed once the lock is released)
    return nullptr;
  }
  g->nchunks = int(chunks.size());
  g->nchunks_hist = int(wide_chunks.size());
  g->lds_grad = spline_grad_lds_bytes(h0->bins, h0->GW, h0->cshift, g->seg != 0);
  // least recently used out (never one that is being evaluated)
  while (g_groups.size() >= kMaxGroups) {
    size_t victim = g_groups.size();
    for (size_t i = 0; i < g_groups.size(); i++)
      if (g_groups[i]->users.load(std::memory_order_acquire) == 0 && (victim == g_groups.size() || g_groups[i]->last_use < g_groups[victim]->last_use)) victim = i;
    if (victim == g_groups.size()) break;
    evicted.push_back(g_groups[victim]);
    g_groups.erase(g_groups.begin() + long(victim));
  }
  g->users.store(1, std::memory_order_release);
  g->last_use = ++g_group_clock;
  g_groups.push_back(g);
  return g;
}

// one evaluation of a group: three launches for all pairs, then every pair's completion tag
int group_eval(MultiGroup* g, const double* se3, bool want_grad, double* costs, double* grads /* n x 7 or null */, bool* all_ok, int* rcs) {
  const int n = int(g->hs.size());
  nidreg_handle* h0 = g->hs[0];
  HIP_TRY(hipSetDevice(g->device));
  PassArgs a;
  fill_pass_args(h0, a);
  a.stream = g->stream;
  a.multi = g->d_table;
  a.dyn.want_grad = want_grad ? 1 : 0;
  a.dyn.neb = h0->NEB;
  InflightGuard guard(g->device);
  a.prio = guard.alone ? 1 : 0;
  pose_from_se3(se3, a.R, a.t);
  for (int k = 0; k < 4; k++) a.q[k] = se3[k];
  for (int i = 0; i < n; i++) {
    nidreg_handle* h = g->hs[size_t(i)];
    bump_seq(h);
    HIP_TRY(begin_histogram(h, g->stream));
    a.dyn.cur[i] = h->hist_cur;
    a.dyn.tag[i] = h->seq;
    h->last_stream = g->stream;
    for (int k = 0; k < 4; k++) h->last_q[k] = se3[k];
    std::memcpy(h->last_R, a.R, sizeof(a.R));
    std::memcpy(h->last_t, a.t, sizeof(a.t));
    h->ev_grad = want_grad;
    }
  // pass A
  a.chunks = h0->wide ? g->d_chunks_hist : g->d_chunks;
  a.nchunks = h0->wide ? g->nchunks_hist : g->nchunks;
  a.seg = h0->wide ? g->seg_hist : g->seg;
  HIP_TRY(launch_spline_hist<double>(a));
  // entropy: NEB workgroups per pair -- none for small tables when every pair has gradient workgroups (they sum the table
  // themselves and clear the next evaluation's buffers, grad_sums_table)
  bool no_entropy_kernel = want_grad && grad_sums_table(h0);
  for (int i = 0; i < n && no_entropy_kernel; i++) no_entropy_kernel = g->hs[size_t(i)]->num_points > 0 && g->hs[size_t(i)]->own_hist;
  if (!no_entropy_kernel) {
    hipLaunchKernelGGL(
      k_entropy<true>, dim3(h0->NEB * n), dim3(kEntropyThreads), 0, g->stream, static_cast<u64*>(nullptr), h0->bins, kEntropyCols, 0.0, static_cast<long long*>(nullptr),
      static_cast<u64*>(nullptr), static_cast<double*>(nullptr), static_cast<double*>(nullptr), static_cast<double*>(nullptr), static_cast<EntropyScalars*>(nullptr), static_cast<double*>(nullptr),
      static_cast<double*>(nullptr), 0.0, static_cast<unsigned int*>(nullptr), static_cast<u64*>(nullptr), 0ll, 1, static_cast<const MultiEntry*>(g->d_table), a.dyn);
    HIP_TRY(hipGetLastError());
  }
  for (int i = 0; i < n; i++) {
    g->hs[size_t(i)]->hist_zeroed[g->hs[size_t(i)]->hist_cur ^ 1] = true;
    g->hs[size_t(i)]->zero_stream = g->stream;
  }
  // pass B (k_entropy<true> ran without its tail for every pair that has gradient workgroups: they run it)
  if (want_grad) {
    a.chunks = g->d_chunks;
    a.nchunks = g->nchunks;
    a.seg = g->seg;
    a.lds_grad = g->lds_grad;
    a.gt_from_partials = no_entropy_kernel ? 2 : 1;
    HIP_TRY(launch_spline_grad<double>(a));
    for (int i = 0; i < n; i++) {
      nidreg_handle* h = g->hs[size_t(i)];
      if (h->nchunks == 0 || h->num_points == 0) {  // an empty pair has no gradient workgroups: finalise (zeros) stand-alone
        HIP_TRY(launch_grad_final(g->stream, h->d_partials, se3, h->d_out, h->d_out_host, h->seq));
      }
    }
  }
  *all_ok = true;
  for (int i = 0; i < n; i++) {
    const int rc = eval_finish_on(g->hs[size_t(i)], g->stream, costs + i, grads ? grads + 7 * i : nullptr);
    if (rc < 0 && std::getenv("NIDREG_DEBUG_GROUP")) {
      (void)hipStreamSynchronize(g->stream);
      for (int j = 0; j < n; j++) {
        nidreg_handle* hj = g->hs[size_t(j)];
        unsigned int c[8];
        (void)hipMemcpy(c, hj->d_counters, sizeof(c), hipMemcpyDeviceToHost);
        std::vector<MultiEntry> tab(static_cast<size_t>(n));
        (void)hipMemcpy(tab.data(), g->d_table, tab.size() * sizeof(MultiEntry), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "group debug: pair %d seq %.0f tag %.0f cost %.6f counters %u %u %u %u nchunks(table) %d group nchunks %d/%d\n", j, hj->seq, hj->h_out[15], hj->h_out[0], c[0], c[1], c[2], c[3],
                     tab[size_t(j)].nchunks, g->nchunks, g->nchunks_hist);
      }
    }
    if (rc < 0) return rc;
    if (rc == NIDREG_FALSE) *all_ok = false;
    if (rcs) rcs[i] = rc;
  }
  return NIDREG_OK;
}
CohortTrace g_cohort_trace;
int cohort_eval(nidreg_handle* h, const double* se3, bool want_grad, double* cost, double* grad7) {
  Cohort* c = h->cohort;
  const int k = int(c->members.size());  // (fixed once sealed, except for members being destroyed -- not while their siblings evaluate)
  static const double wait_us = [] {
    const char* e = std::getenv("NIDREG_COHORT_WAIT_US");
    return e ? std::max(0.0, std::strtod(e, nullptr)) : 100.0;
  }();
  bool leader = false;
  {
    std::lock_guard<std::mutex> lk(c->rv_mu);
    if (c->round_open) {
      if (std::memcmp(c->round_pose, se3, sizeof(c->round_pose)) != 0 || c->round_grad != want_grad) return kNotJoined;
      const int n = c->n_arrived.load(std::memory_order_relaxed);
      for (int i = 0; i < n; i++)
        if (c->arrivals[i].h == h) return kNotJoined;
      if (n >= 16) return kNotJoined;
    } else {
      c->round_open = true;
      std::memcpy(c->round_pose, se3, sizeof(c->round_pose));
      c->round_grad = want_grad;
      c->n_arrived.store(0, std::memory_order_relaxed);
      leader = true;
    }
    const int me = c->n_arrived.load(std::memory_order_relaxed);
    c->arrivals[me] = Cohort::Arrival{h, cost, grad7};
    h->rv_done.store(0, std::memory_order_relaxed);
    c->n_arrived.store(me + 1, std::memory_order_release);
  }
  if (!leader) {  // the round's leader evaluates; spin (an evaluation takes 100-300 us), then yield
    unsigned spins = 0;
    while (h->rv_done.load(std::memory_order_acquire) == 0) {
      if (++spins < 200000) {
        __builtin_ia32_pause();
      } else {
        std::this_thread::yield();
      }
    }
    return h->rv_rc;
  }
  // leader: wait for the siblings, close the round
  const long long tr0 = g_cohort_trace.on ? mono_ns() : 0;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  while (c->n_arrived.load(std::memory_order_acquire) < k) {
    for (int i = 0; i < 16; i++) __builtin_ia32_pause();
    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if ((t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3 > wait_us) break;
  }
  Cohort::Arrival arr[16];
  int n = 0;
  {
    std::lock_guard<std::mutex> lk(c->rv_mu);
    n = c->n_arrived.load(std::memory_order_acquire);
    for (int i = 0; i < n; i++) arr[i] = c->arrivals[i];
    c->round_open = false;  // later arrivals open the next round
  }
  // member order (not arrival order): the same subset is the same group, whoever came first
  std::sort(arr, arr + n, [c](const Cohort::Arrival& a, const Cohort::Arrival& b) {
    return std::find(c->members.begin(), c->members.end(), a.h) < std::find(c->members.begin(), c->members.end(), b.h);
  });
  int my_rc = NIDREG_OK;
  bool done = false;
  const long long tr1 = g_cohort_trace.on ? mono_ns() : 0;
  long long tr2 = 0;
  if (n >= 2) {
    nidreg_handle* hs[16];
    for (int i = 0; i < n; i++) hs[i] = arr[i].h;
    MultiGroup* g = find_or_make_group(hs, n);
    if (g) {
      double costs[kMaxMulti], grads[kMaxMulti * 7];
      int rcs[kMaxMulti];
      for (int i = 0; i < n; i++) rcs[i] = NIDREG_OK;
      bool all_ok = true;
      const int rc = group_eval(g, se3, want_grad, costs, want_grad ? grads : nullptr, &all_ok, rcs);
      release_group(g);
      tr2 = g_cohort_trace.on ? mono_ns() : 0;
      const std::string err = rc < 0 ? g_last_error : std::string();
      for (int i = 0; i < n; i++) {
        if (rc >= 0) {
          if (arr[i].cost) *arr[i].cost = costs[i];
          if (want_grad && arr[i].grad7)
            for (int q = 0; q < 7; q++) arr[i].grad7[q] = grads[7 * i + q];
        }
        const int r = rc < 0 ? rc : rcs[i];
        if (arr[i].h == h) {
          my_rc = r;
        } else {
          arr[i].h->rv_rc = r;
          arr[i].h->rv_done.store(1, std::memory_order_release);
        }
      }
      if (rc < 0) g_last_error = err;
      done = true;
    }
  }
  if (!done) {  // alone in the round (or the group could not be built): everybody evaluates by itself, the leader for all
    for (int i = 0; i < n; i++) {
      nidreg_handle* m = arr[i].h;
      InflightGuard guard(m->device);
      int rc = eval_launch(m, se3, want_grad, guard.alone && n == 1);
      if (!rc) rc = eval_finish(m, arr[i].cost, want_grad ? arr[i].grad7 : nullptr);
      if (m == h) {
        my_rc = rc;
      } else {
        m->rv_rc = rc;
        m->rv_done.store(1, std::memory_order_release);
      }
    }
  }
  if (g_cohort_trace.on) {
    const long long tr3 = mono_ns();
    g_cohort_trace.rounds.fetch_add(1);
    if (n == k) g_cohort_trace.full.fetch_add(1);
    g_cohort_trace.wait_ns.fetch_add(tr1 - tr0);
    g_cohort_trace.eval_ns.fetch_add((tr2 ? tr2 : tr3) - tr1);
    g_cohort_trace.tail_ns.fetch_add(tr2 ? tr3 - tr2 : 0);
  }
  return my_rc;
}

// the Nelder-Mead objective's sum over pairs (visual_camera_calibration.cpp:103-119) the same way: two launches in all
int group_eval_iso(MultiGroup* g, const double* T, double* costs) {
  const int n = int(g->hs.size());
  nidreg_handle* h0 = g->hs[0];
  HIP_TRY(hipSetDevice(g->device));
  PassArgs a;
  fill_pass_args(h0, a);
  a.stream = g->stream;
  a.multi = g->d_table;
  a.dyn.want_grad = 0;
  a.dyn.neb = h0->NEB;
  for (int k = 0; k < 12; k++) a.iso[k] = T[k];
  a.nfast = nearest_fast_args(h0, T);
  for (int i = 0; i < n; i++) {
    nidreg_handle* h = g->hs[size_t(i)];
    bump_seq(h);
    HIP_TRY(begin_histogram(h, g->stream));
    a.dyn.cur[i] = h->hist_cur;
    a.dyn.tag[i] = h->seq;
    h->last_stream = g->stream;
    h->ev_grad = false;
  }
  a.chunks = g->d_chunks;
  a.nchunks = g->nchunks;
  a.seg = g->seg;
  HIP_TRY(launch_nearest_hist<double>(a));
  hipLaunchKernelGGL(
    k_entropy<true>, dim3(h0->NEB * n), dim3(kEntropyThreads), 0, g->stream, static_cast<u64*>(nullptr), h0->bins, kEntropyCols, 0.0, static_cast<long long*>(nullptr),
    static_cast<u64*>(nullptr), static_cast<double*>(nullptr), static_cast<double*>(nullptr), static_cast<double*>(nullptr), static_cast<EntropyScalars*>(nullptr), static_cast<double*>(nullptr),
    static_cast<double*>(nullptr), 0.0, static_cast<unsigned int*>(nullptr), static_cast<u64*>(nullptr), 0ll, 1, static_cast<const MultiEntry*>(g->d_table), a.dyn);
  HIP_TRY(hipGetLastError());
  for (int i = 0; i < n; i++) {
    g->hs[size_t(i)]->hist_zeroed[g->hs[size_t(i)]->hist_cur ^ 1] = true;
    g->hs[size_t(i)]->zero_stream = g->stream;
  }
  for (int i = 0; i < n; i++) {
    const int rc = eval_finish_on(g->hs[size_t(i)], g->stream, costs + i, nullptr);
    if (rc < 0) return rc;
  }
  return NIDREG_OK;
}

// handles[0..n) all distinct, compatible and on one device?
// Measured on the same clouds in one harness (tools/archive/omp_pairs.cpp on the 10M-point scene split into n pairs,
// profiles/archive/r03q_omp_pairs.jsonl), microseconds per evaluation of all pairs: single grid 152 / 149 / 172 at 2 / 4 / 8
// pairs, per-pair launches 154 / 198 / 274, one OpenMP caller per pair 173 / 215 / 282.  (Until the chunk tables were made
// to fit one round -- split_groups -- the single grid took 191 / 185 / 178 and two or three pairs ran as per-pair launches.)
// The single grid (three launches instead of 3 n) is therefore used from two pairs on; NIDREG_MULTI_GRID_MIN=n moves the
// threshold, NIDREG_NO_MULTI_GRID=1 keeps per-pair launches (every pair's histogram pass queued before the rest).
int multi_grid_min() {
  static const int v = [] {
    const char* e = std::getenv("NIDREG_MULTI_GRID_MIN");
    const long m = e ? std::strtol(e, nullptr, 10) : 2;
    return int(std::max(2L, std::min(m, long(kMaxMulti) + 1)));
  }();
  return v;
}
bool can_group(nidreg_handle* const* handles, int n) {
  if (n < multi_grid_min() || n > kMaxMulti || std::getenv("NIDREG_NO_MULTI_GRID")) return false;
  if (!groupable(handles[0], handles[0])) return false;
  for (int i = 1; i < n; i++)
    if (!groupable(handles[0], handles[i])) return false;
  for (int i = 0; i < n; i++)
    for (int j = i + 1; j < n; j++)
      if (handles[i] == handles[j]) return false;
  return true;
}
}  // namespace nidreg_detail
