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
#include <array>
#include <climits>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>
// RCCL is an OPTIONAL, run-time dependency (dlopen in rccl_api below): the handful of types and enumerator values of its public C
// ABI (rccl/rccl.h = NCCL 2.x: stable across releases) are declared here, so that libnidreg.so builds on a ROCm install without
// the RCCL development headers.  Where the header is present the values are checked against it at compile time.
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
enum { ncclSuccess = 0, ncclSum = 0, ncclMax = 2, ncclInt64 = 4, ncclFloat64 = 8 };
#if __has_include(<rccl/rccl.h>)
#include <hip/hip_fp16.h>  // (what rccl.h includes itself: seen here first so that the namespace below holds RCCL's declarations only)
#include <limits.h>
namespace rccl_header_check {
#include <rccl/rccl.h>
static_assert(int(ncclSuccess) == 0 && int(ncclSum) == 0 && int(ncclMax) == 2 && int(ncclInt64) == 4 && int(ncclFloat64) == 8 && sizeof(ncclUniqueId) == 128,
              "the RCCL ABI values declared above differ from this install's rccl/rccl.h");
}  // namespace rccl_header_check
#endif

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

constexpr int kEntropyCols = kEntropyColsMax;  // histogram columns per k_entropy workgroup (nid_kernels.hpp)
const int kNumIntr[6] = {4, 4, 5, 2, 4, 4};
const int kNumDist[6] = {5, 4, 4, 0, 1, 8};

}  // namespace

struct nidreg_handle {
  int device = 0;
  int model = 0, mode = 0, precision = 0, bins = 0;
  // bins > 256 (WideBins below): `bins` is the compact count the kernels run on, bins_user the caller's; inv_*[compact] = the
  // caller's bin (the getters expand with them).  bins_user == 0: the two are the same.
  int bins_user = 0;
  std::vector<uint16_t> inv_img, inv_pts;
  bool nearest_exact = false;  // NIDREG_FLAG_NEAREST_EXACT
  int W = 0, H = 0, pitch = 0;
  int GW = 0, NG = 0, cshift = 0;
  int wide = 0;  // k_spline_hist<.., WIDE>: B = 256, GW = 1, 32 copies, 512 threads
  int NEB = 0;  // entropy column blocks
  int self_entropy = -1;  // 1: cost+Jacobian evaluations launch no entropy kernel (grad_sums_table); -1: not decided yet
  int frac_bits = 0;
  int rec64 = 0;
  int64_t num_points = 0;
  int nchunks = 0;       // gradient pass / generic histogram kernels
  int nslots = 0;        // segments in that table = 12-double partials of the gradient pass (>= nchunks)
  int partials_cap = 0;  // 12-double slots allocated behind d_partials at creation (cohort / multi-pair tables must fit)
  int seg = 0, seg_hist = 0;  // the table (d_chunks / d_chunks_hist) has chunks that run across column groups: SEG kernels
  size_t chunks_cap = 0, chunks_hist_cap = 0;  // entries allocated behind d_chunks / d_chunks_hist
  struct Cohort* cohort = nullptr;  // NIDREG_COHORT=1: the handles created together for one MultiNIDCost share ONE round of workgroups
  std::vector<Chunk> cohort_chunks, cohort_chunks_hist;  // host copies of a sealed cohort member's share tables (the single grid concatenates them)
  std::atomic<int> rv_done{0};  // rendezvous: the round's leader has stored this member's results
  int rv_rc = 0;
  int nchunks_hist = 0;  // WIDE histogram kernel's own table (0 = shares d_chunks)
  double intr[5] = {0}, dist[8] = {0};
  double max_fov = 0.0;

  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipStream_t last_stream = nullptr;  // the stream the most recent evaluation's kernels ran on (a multi-pair group's, else `stream`): what the histogram getters drain
  void* d_pts = nullptr;
  Chunk* d_chunks = nullptr;
  Chunk* d_chunks_hist = nullptr;
  uint32_t* d_gend = nullptr;  // [NG] end offsets of the column groups among the records (nid_kernels.hpp Segments)
  uint8_t* d_img = nullptr;
  u64* d_hist = nullptr;      // histogram of the current / most recent evaluation (accumulation target of pass A)
  // a shard of a ShardSet owns a range of histogram COLUMN GROUPS: it holds the points of those columns only, and its
  // histogram is the pair's histogram restricted to them (the other columns stay zero)
  struct ShardSet* set = nullptr;  // non-NULL on the leader (shard 0) of a set: nidreg_eval* fan out over the shards
  bool is_shard = false;
  int shard_index = 0;
  ShardTable* d_shard_tab = nullptr;  // device copy of this shard's ShardTable (peer flag / gather blocks, owned columns)
  // one process per GPU (nidreg_shard_attach_rccl / nidreg_shard_comm_init): this handle holds an index-range slice of the pair,
  // every evaluation all-reduces the integer histogram (and the 7-double gradient partial) over this communicator
  void* rccl_comm = nullptr;
  bool rccl_owned = false;
  int col_lo = 0, col_hi = 0;         // owned histogram columns
  size_t img_bytes = 0;
  // double buffering of the histogram (own buffers only): evaluation k accumulates into one buffer and
  // its k_entropy zeroes the OTHER one for evaluation k + 1, so no memset sits on the critical path
  u64* d_hist_buf[2] = {nullptr, nullptr};
  bool hist_zeroed[2] = {false, false};
  hipStream_t zero_stream = nullptr;  // the stream of the kernel that cleared the idle buffer (begin_histogram orders a launch on another stream behind it)
  int hist_cur = 0;
  bool own_hist = false;
  double* d_out = nullptr;
  bool own_out = false;
  void* d_scratch = nullptr;  // ONE allocation carved into the per-evaluation scratch below (zeroed at creation)
  long long* d_part_hj = nullptr;  // fixed-point entropy partials (nid_kernels.hpp ent_fixed)
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
  uint64_t seq_bits = 0;               // its bit pattern (what the acquire load of the tag compares against)
  unsigned int evals_since_reap = 0;
  // asynchronous evaluations (nidreg_submit / nidreg_wait): a ring of host-mapped result blocks, one per evaluation in flight
  double* h_ring = nullptr;   // [kAsyncDepth][NIDREG_OUT_DOUBLES], pinned, host-mapped (allocated at the first submit)
  double* d_ring = nullptr;   // its device address
  struct Pending {
    int64_t ticket = 0;      // 0: free
    uint64_t bits = 0;       // completion tag of the evaluation (the handle's sequence number at its launch)
    bool grad = false, done = false, counted = false;  // done: evaluated synchronously inside nidreg_submit (sharded handles, ext_out); counted: holds an in-flight count of its device
    int rc = 0;
    double res[8] = {0};
  };
  Pending pending[8];
  int async_outstanding = 0;
  int64_t next_ticket = 0;  // tickets are numbered by a counter of their own: the completion tags (seq) advance by more than one per
                            // submit on handles whose submit evaluates synchronously (shards bump the leader's seq themselves)

  size_t lds_hist = 0, lds_grad = 0, lds_entropy = 0;
  int64_t hist_words = 0;
  std::vector<int64_t> gcount;  // record offsets of the column groups (host copy: multi-pair groups build their chunk tables from it)
  int num_cus = 256, per_cu_grad = 4, per_cu_hist = 2;

  // one launch per cost+Jacobian evaluation (nid_fused.hpp): small tables, clouds whose chunks fit the LDS stash
  int fused = 0;               // 0: not planned yet, 1: usable, -1: not applicable / switched off (a barrier that timed out)
  int fused_cap = 0, fused_full = 0; // points of LDS stash per workgroup; 1: full stash format, 0: (u, v) only
  int64_t longest_chunk = 0;         // records of the longest chunk of the gradient-pass table (0: unknown -> no fused route)
  void* d_fused_scratch = nullptr;   // the grid barrier's arrival counter (256 B), then its per-workgroup release words (128 B each)
  u64* d_fused_barrier = nullptr;
  uint64_t fused_arrivals = 0;       // arrivals the counter holds once every launch so far has passed its barrier
  uint64_t fused_launches = 0;       // fused launches so far (the barrier's epoch)
  bool fused_last = false;           // the evaluation in flight (or last finished) ran on the fused route

  // NEAREST, equirectangular: (cos, sin) of the column-boundary longitudes, then the signed squared sines of the row-boundary
  // latitudes (nid_kernels.hpp NearestFast); eq_kmax / eq_jmax = ceil of the intrinsics' W / H
  double* d_eq_tab = nullptr;
  int eq_kmax = 0, eq_jmax = 0;

  int timing = 0;  // 1: per-kernel events (three-kernel path), 2: events around whichever path runs
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool ev_grad = false;
  double last_q[4] = {0, 0, 0, 1};
  double last_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double last_t[3] = {0, 0, 0};
};

// the handles created together for one MultiNIDCost (NIDREG_COHORT=1; see "cohorts" below)
struct Cohort {
  std::mutex mu;
  std::vector<nidreg_handle*> members;
  std::atomic<bool> sealed{false};
  int device = 0;
  // rendezvous of concurrent callers (cohort_eval): the round that is collecting arrivals
  std::mutex rv_mu;
  bool round_open = false;
  double round_pose[7] = {0};
  bool round_grad = false;
  std::atomic<int> n_arrived{0};
  struct Arrival {
    nidreg_handle* h;
    double* cost;
    double* grad7;
  } arrivals[16];
};

// One LiDAR-camera pair spread over several GPUs (BASELINE north_star: "disjoint point slices with a final all-reduce of the
// 2D histogram over xGMI"), driven by ONE host process.  The slices are cut along the pose-independent histogram column
// ("shard by histogram column"): shard g holds the points of a contiguous range of column groups, chosen from
// the groups' point counts so that the shards are balanced to within one cut unit.  The shards' histograms then have disjoint
// support: the all-reduce of the B x B table is an all-gather of columns by plain stores -- every shard keeps a replica of the
// whole integer histogram in fine-grained memory and the owners of a column block store it into every replica
// (nid_kernels.hpp k_entropy_repl: ONE exchange per evaluation).  Per evaluation every shard runs  histogram -> k_entropy_repl
// -> gradient  on its own stream -- the kernels of an unsharded handle with one exchange inside the middle one --, launched by
// its own host thread (the caller for shard 0), and the host adds the n 7-double gradient partials.  Every shard computes the
// cost from the same integers: bit-identical, which the host CHECKS after every evaluation (a stale cross-device read cannot
// go unnoticed).
struct ShardSet {
  std::vector<nidreg_handle*> shards;  // [0] = the leader (owns this set), the rest are owned by the set
  std::vector<u64*> flags;             // per shard: fine-grained flag block on its device (kFlagWords)
  std::vector<u64*> gather;            // per shard: fine-grained gather block (kGatherWords)
  int CB = kEntropyCols, nblocks = 0;  // k_entropy_repl: columns per workgroup, workgroups (every cut between shards is a multiple of CB columns)
  std::vector<int> lock_devices;       // distinct devices of the set, ascending: set_eval locks them in this order
  bool colocated = false;              // a device is listed more than once (a 1-GPU box exercising the protocol)
  bool poisoned = false;               // an evaluation failed half way: the flag sequence is no longer trustworthy
  u64 seq = 0;
  unsigned long long timeout_ticks = 300000000ull;  // 3 s of the 100 MHz wall clock
  // worker threads (one per shard >= 1): spin briefly on `gen`, then sleep on the condition variable
  std::vector<std::thread> workers;
  std::atomic<uint64_t> gen{0};
  std::atomic<int> pending{0};
  std::atomic<int> sleepers{0};
  std::atomic<bool> stop{false};
  std::mutex mu;
  std::condition_variable cv;
  // job of the current generation
  int job_mode = 0;  // NIDREG_MODE_*
  bool job_grad = false;
  double job_pose[16];
  std::vector<int> rc;
  std::vector<std::array<double, 8>> res;  // cost, grad7
};

namespace {

void free_shard_set(ShardSet* set);
void rccl_release(nidreg_handle* h);
void drop_groups_of(const nidreg_handle* h);
void cohort_leave(nidreg_handle* h);
std::atomic<int> g_inflight[NIDREG_MAX_DEVICES];  // evaluations in flight per device (InflightGuard below)

// ---- streams and host-mapped result blocks are kept between handles ---------------------------------------------------------
// The reference builds a new NIDCost per pair in every outer iteration (visual_camera_calibration.cpp:199-208).  Creating and
// destroying a handle for a 100k-point cloud took 0.98 ms, of which hipStreamCreate + hipStreamDestroy 0.5 + 0.4 ms and
// hipHostFree 0.2 ms (rocprofv3 --hip-trace, profiles/archive/r04m_hip_api_stats.csv) -- thirty evaluations' worth.  A destroyed
// handle's stream (idle: free_handle synchronises it) and result blocks go to a per-device free list and the next handle on
// that device takes them; nidreg_trim() releases them.
struct ResourcePool {
  std::mutex mu;
  std::vector<hipStream_t> streams;
  std::vector<void*> out_blocks;   // NIDREG_OUT_DOUBLES doubles, mapped + coherent
  std::vector<void*> ring_blocks;  // kAsyncDepth of them
};
ResourcePool g_pool[NIDREG_MAX_DEVICES];
constexpr size_t kPoolCap = 64;

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
// staging of nidreg_project for a handful of points: [3 n | 2 n | 6 n] doubles, host-mapped, one block per device
constexpr int64_t kSmallProject = 64;
struct SmallProject {
  std::mutex mu;
  double* host = nullptr;
  double* dev = nullptr;
};
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


// The fixed-point unit of the SPLINE histogram: U = 36 round(2^frac / 36) -- within 18 of 2^frac, and a multiple of 36
// so that the constants U/36, 3U/36, 4U/36, 6U/36 of the x-weight polynomial are integers (nid_device.hpp
// bspline_scale; both axes produce 6 b).  NEAREST counts: 1.
inline double fixed_unit(const nidreg_handle* h) {
  if (h->mode == NIDREG_MODE_NEAREST || h->frac_bits == 0) return 1.0;
  return 36.0 * std::rint(std::ldexp(1.0, h->frac_bits) / 36.0);
}
// U/36 grid steps of 2^-1074 as a subnormal double (the kernels' dn_scale / MultiEntry::k16)
inline double fixed_unit_k(const nidreg_handle* h) { return std::ldexp(fixed_unit(h) / 36.0, -1074); }

// capacity of a handle's gradient-partial buffer, in 12-double slots: its own table's segments, and room for any table a
// multi-pair group builds for it (at most one slot per chunk plus one per column group)
inline int partial_slots(const nidreg_handle* h) { return std::max(std::max(h->nchunks, h->nchunks_hist), 1) + h->NG + 1; }

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

inline void bump_seq(nidreg_handle* h) {
  h->last_stream = h->stream;  // (a multi-pair group overrides this after the call)
  h->seq += 1.0;
  std::memcpy(&h->seq_bits, &h->seq, sizeof(h->seq_bits));
}

// Evaluations in flight per device (this process).  An evaluation that has its device to itself runs with progress
// priority in the spline passes; with several callers on one GPU (the reference's OpenMP loop over pairs,
// visual_camera_calibration.cpp:161) it is off: the rule made competing kernels 5-16 % slower
// (profiles/archive/r02h_multi_pair_threads.txt).
struct InflightGuard {
  int dev;
  bool alone;
  explicit InflightGuard(int d) : dev(d >= 0 && d < NIDREG_MAX_DEVICES ? d : -1), alone(false) {
    if (dev >= 0) alone = g_inflight[dev].fetch_add(1, std::memory_order_acq_rel) == 0;
  }
  ~InflightGuard() {
    if (dev >= 0) g_inflight[dev].fetch_sub(1, std::memory_order_acq_rel);
  }
  InflightGuard(const InflightGuard&) = delete;
  InflightGuard& operator=(const InflightGuard&) = delete;
};

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

int launch_hist_spline(nidreg_handle* h, const double* se3, bool alone = false) {
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
int launch_entropy(nidreg_handle* h, double tag, bool tail = true) {
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

int launch_grad(nidreg_handle* h, bool alone = false, int from_partials = 0) {
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
    hipLaunchKernelGGL(k_grad_final, dim3(1), dim3(kThreads), 0, h->stream, h->d_partials, 0, h->last_q[0], h->last_q[1], h->last_q[2], h->last_q[3], h->d_out, h->d_out_host, h->seq);
    HIP_TRY(hipGetLastError());
  }
  return NIDREG_OK;
}

// asynchronous part of nidreg_eval, in two steps so that a multi-handle caller can put every GPU to work before it
// queues the rest: eval_launch_first = the histogram pass, eval_launch_rest = entropy (+ gradient)
int eval_launch_first(nidreg_handle* h, const double* se3, bool alone = false) {
  if (h->mode != NIDREG_MODE_SPLINE) return fail(NIDREG_ERR_INVALID, "nidreg_eval: handle was created in NEAREST mode");
  HIP_TRY(hipSetDevice(h->device));
  bump_seq(h);
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  const int rc = launch_hist_spline(h, se3, alone);
  if (rc) return rc;
  if (h->timing == 1) HIP_TRY(hipEventRecord(h->ev[2], h->stream));
  return NIDREG_OK;
}
int eval_launch_rest(nidreg_handle* h, bool want_grad, bool alone = false) {
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

int eval_launch(nidreg_handle* h, const double* se3, bool want_grad, bool alone = false) {
  h->fused_last = false;
  const int rc = eval_launch_first(h, se3, alone);
  if (rc) return rc;
  return eval_launch_rest(h, want_grad, alone);
}

int eval_finish_block(nidreg_handle* h, hipStream_t stream, const double* block, uint64_t seq_bits, bool polled, double* cost, double* grad7);
int eval_one(nidreg_handle* h, const double* se3, double* cost, double* grad7);
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

}  // namespace

extern "C" {

const char* nidreg_last_error(void) { return g_last_error.c_str(); }
const char* nidreg_version(void) { return "nidreg 0.5 (gfx950, hand-written HIP)"; }

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

// device-resident cloud: uploaded once per pair, re-culled / re-bucketed on the GPU for every handle
struct nidreg_cloud {
  int device = 0;
  int64_t n = 0;
  double* d_pts = nullptr;  // n x 4 doubles (x y z 1)
  double* d_int = nullptr;  // n doubles
};

namespace {

// ---- chunk tables: each chunk = one workgroup's slice of the bucketed record array.  A pass should be ONE round of
// co-resident workgroups (`target` of them; measured on cfg 2: the per-workgroup prologue / flush is amortised over more
// points and no partial last round is left -- 2048 chunks +4 %, 4096 +12 %), and the round ends with its LONGEST chunk.
// Rounds 1-3 gave every chunk the points of one column group only (a group split evenly into an integer number of chunks):
// exact on a cloud whose columns are equally full -- the rank equalisation of preprocess.cpp:464-473 makes them so for a
// WHOLE cloud --, but the clouds `calibrate` evaluates are view-culled (visual_camera_calibration.cpp:201-206) and a pair of
// a multi-pair set is a subset: with columns of 0.5 ... 1.5 x the mean an integer split leaves chunks of 3/4 ... 3/2 of the
// mean (measured: +20 % per point, profiles/archive/r03r_culled_cloud_ab*.json).  Since round 4 a chunk is a CONTIGUOUS RANGE of
// records that may run across group boundaries (nid_kernels.hpp Segments): the workgroup flushes / rebuilds its tile at
// every boundary, which costs about `overhead` records' worth of time (pipeline drain, 64 KB of LDS traffic, the first
// load latency of the next segment).  The table minimises the longest chunk under that cost model:
//     cost(chunk) = sum over its segments of (overhead + records),
// smallest bound C for which a greedy left-to-right fill needs <= target chunks (binary search; the fill never opens a
// segment shorter than `overhead` at the end of a chunk, and cuts inside a group at multiples of 64 records).  A pure function
// of (group offsets, target, overhead, max_segs): the gradient's partial sums keep a fixed order, run to run.
// max_segs = segments a chunk may hold: kMaxSegs for the single-column kernels (B > 128) and every NEAREST table, 1 where a
// workgroup's tile spans several columns (B <= 128: few groups, large tiles -- a chunk then ends at the group boundary as it
// always did).  Tables without a multi-segment chunk run the straight-line kernels of rounds 1-3, the others the looped
// (SEG) instantiations; NIDREG_MAX_SEGS=1 keeps every table on one segment per chunk (A/B runs).
// Chunk::pad = pair | slot << 8: pair = index into the multi-pair table (0 in a single-pair table, pair < 0), slot = number of
// segments in the table's earlier chunks -- the gradient pass stores one 12-double partial per SEGMENT, in this order.
// *nslots_out (nullable) = segments in the whole table.
int64_t fill_chunks(const int64_t* gcount, int NG, int64_t C, int64_t overhead, int max_segs, int pair, std::vector<Chunk>* out, int64_t* nslots_out = nullptr) {
  int64_t nchunks = 0, cur = 0;  // cur = cost already in the open chunk (0: none open)
  int64_t nslots = 0, first_slot = 0;
  Chunk c{0, 0, 0, 0};
  auto close = [&]() {
    if (cur > 0 && out) {
      c.pad = uint32_t(pair < 0 ? 0 : pair) | (uint32_t(first_slot) << 8);
      out->push_back(c);
    }
    cur = 0;
  };
  for (int g = 0; g < NG; g++) {
    int64_t pos = gcount[g];
    const int64_t hi = gcount[g + 1];
    while (pos < hi) {
      const int64_t rem = hi - pos;
      int64_t room = C - cur - overhead;  // records of this group the open chunk can still take
      if (cur > 0 && (room < std::min(rem, std::max<int64_t>(overhead, 64)) || nslots - first_slot >= max_segs)) {  // not worth a segment (or the chunk is at its segment limit): next chunk
        close();
        continue;
      }
      if (cur == 0) {
        nchunks++;
        c.start = uint32_t(pos);
        c.count = 0;
        c.group = uint32_t(g);
        first_slot = nslots;
        room = std::max<int64_t>(room, 64);  // an empty chunk always makes progress
      }
      int64_t take = std::min(rem, room);
      if (take < rem) take = std::max<int64_t>(64, take / 64 * 64);  // cut inside a group: whole waves
      take = std::min(take, rem);
      c.count += uint32_t(take);
      cur += overhead + take;
      nslots++;
      pos += take;
      if (pos < hi) close();  // the group goes on in the next chunk
    }
  }
  close();
  if (nslots_out) *nslots_out = nslots;
  return nchunks;
}
// returns the number of segments (= gradient partial slots) of the table appended to `chunks`
// smallest cost bound for which the fill needs <= target chunks (or, when even one chunk per group is too many, the bound that
// gives one chunk per group) -- and the number of chunks at that bound
int64_t best_bound(const int64_t* gcount, int NG, int64_t target, int64_t overhead, int max_segs, int64_t N, int64_t nonempty, int64_t* nchunks_out) {
  // fill_chunks(C) is non-increasing in C; C = everything in one chunk always fits (as far as max_segs allows)
  int64_t lo = overhead + 63, hi = N + nonempty * overhead + 64;  // lo: too small (or just feasible -- checked first), hi: feasible
  if (fill_chunks(gcount, NG, lo + 1, overhead, max_segs, -1, nullptr) <= target) {
    hi = lo + 1;
  } else {
    while (hi - lo > 1) {
      const int64_t mid = lo + (hi - lo) / 2;
      if (fill_chunks(gcount, NG, mid, overhead, max_segs, -1, nullptr) <= target) {
        hi = mid;
      } else {
        lo = mid;
      }
    }
  }
  *nchunks_out = fill_chunks(gcount, NG, hi, overhead, max_segs, -1, nullptr);
  return hi;
}
int64_t split_groups(const int64_t* gcount, int NG, int64_t target, int64_t overhead, int max_segs, int pair, std::vector<Chunk>& chunks) {
  const int64_t N = gcount[NG] - gcount[0];
  if (N <= 0) return 0;
  target = std::max<int64_t>(target, 1);
  overhead = std::max<int64_t>(overhead, 0);
  max_segs = std::max(1, std::min(max_segs, kMaxSegs));
  int64_t nonempty = 0;
  for (int g = 0; g < NG; g++) nonempty += gcount[g + 1] > gcount[g] ? 1 : 0;
  int64_t n_one = 0;
  const int64_t c_one = best_bound(gcount, NG, target, overhead, 1, N, nonempty, &n_one);
  int64_t bound = c_one;
  int segs = 1;
  if (max_segs > 1) {
    // Chunks across groups only where they PAY: the looped kernel instantiations run 2-6 % slower per point than the
    // straight-line ones (measured, profiles/archive/r04c_culled_cloud_ab.jsonl: on clouds whose columns are nearly equally full the
    // better balance did not make up for it), so the segmented table must beat the best one-group-per-chunk table by
    // NIDREG_SEG_MIN_GAIN (default 10 %) in the cost model -- rounds x longest chunk -- to be chosen.
    const char* mg = std::getenv("NIDREG_SEG_MIN_GAIN");
    const double min_gain = mg ? std::max(0.0, std::strtod(mg, nullptr)) : 0.10;
    int64_t n_seg = 0;
    const int64_t c_seg = best_bound(gcount, NG, target, overhead, max_segs, N, nonempty, &n_seg);
    const double t_one = double(c_one) * double((n_one + target - 1) / target), t_seg = double(c_seg) * double((n_seg + target - 1) / target);
    if (t_seg * (1.0 + min_gain) < t_one) {
      bound = c_seg;
      segs = max_segs;
    }
  }
  int64_t nslots = 0;
  fill_chunks(gcount, NG, bound, overhead, segs, pair, &chunks, &nslots);
  return nslots;
}
// records' worth of time one more segment costs a workgroup of the given kernel (measured orders of magnitude: a WIDE
// histogram workgroup streams ~400 records/us and a boundary costs it ~2.5 us; a gradient / generic workgroup ~150-200
// records/us and ~2 us).  NIDREG_SEG_OVERHEAD=<records> overrides both (A/B runs).
int max_segments(int mode, int GW) {
  int m = (mode == NIDREG_MODE_NEAREST || GW == 1) ? kMaxSegs : 1;
  if (const char* e = std::getenv("NIDREG_MAX_SEGS")) m = std::max(1, std::min(m, int(std::strtol(e, nullptr, 10))));
  return m;
}
int64_t segment_overhead(bool wide_hist) {
  if (const char* e = std::getenv("NIDREG_SEG_OVERHEAD")) return std::max<int64_t>(0, std::strtoll(e, nullptr, 10));
  return wide_hist ? 1024 : 384;
}

// How many chunks (= workgroups) a pass over `points` records gets.  A FULL round -- every co-resident slot of the GPU, per_cu
// workgroups on each CU -- is right for the 10M-point headline and wrong for the clouds `calibrate` usually sees: a workgroup
// costs a fixed prologue + epilogue (tile zeroing, the G columns' logarithms, the flush's atomics on the cells every other
// workgroup flushes too, the partial reduction), and the workgroups of a CU share its issue slots, so a pass costs about
//     a x chunks / CUs  +  b x points / chunks          (prologues on the busiest CU + the sweeps of one workgroup's slice),
// smallest at chunks ~ sqrt(points).  Measured (profiles/archive/r04i_small_cloud_sweep.jsonl, synchronous cost+Jacobian evaluation,
// best chunk count against the full round's): 30k points 64-128 chunks, 28 us against 37; 100k 128, 29 against 48; 300k
// 192-256, 35 against 51; 1M 384-512, 45 against 54; 3M 512-768, 71 against 76; 10M 1024 (the full round).  The rule is the
// square root through those points -- CUs/2 chunks at 100k points --, capped by the full round (reached at 6.4M points).
// NIDREG_FULL_ROUND=1 restores the full round at every size (A/B runs).
int64_t round_chunks(int per_cu, int num_cus, int64_t points) {
  const int64_t full = std::max<int64_t>(1, int64_t(per_cu) * num_cus);
  static const bool always_full = [] {
    const char* e = std::getenv("NIDREG_FULL_ROUND");
    return e && *e && *e != '0';
  }();
  if (always_full) return full;
  double t = 0.5 * double(num_cus) * std::sqrt(double(std::max<int64_t>(points, 1)) / 1.0e5);
  // Round 5: from 0.6 CUs on, whole multiples of the CU count (the nearest one on a logarithmic scale).  With the per-workgroup
  // overhead of this round's kernels (fast_log, the reduction through LDS) a count between two multiples -- 569 chunks on 256 CUs
  // -- leaves some CUs a workgroup more than the others and loses 3-8 % against the multiple next to it: 500k points 280 -> 256
  // chunks 35.5 -> 32.7 us, 2M 569 -> 512 57.3 -> 53.6, 3M 700 -> 768 70.9 -> 67.1, 4M 802 -> 768 80.3 -> 77.1
  // (profiles/r05z_small_cloud_sweep_b16.jsonl; 256-bin tables were multiples of their 256 column groups already).
  if (t >= 0.6 * double(num_cus)) {
    int k = 1;
    while (t > double(num_cus) * std::sqrt(double(k) * double(k + 1))) k++;
    t = double(k) * double(num_cus);
  }
  return std::max<int64_t>(1, std::min<int64_t>(full, int64_t(t + 0.5)));
}
// ... for a handle's own tables, in whole multiples of its non-empty column groups where it has several: a target between
// two multiples splits SOME groups once more and leaves the longest chunk as it was (B = 256, 1M points: 384 chunks 53 us,
// 256 chunks 47 us, 512 chunks 47 us), and a target below the group count would put several groups into every chunk --
// the looped kernels, 60 us against 34 us for 256 one-group chunks at 30k points.
int64_t snap_to_groups(int64_t target, const int64_t* gcount, int NG, int64_t cap) {
  int64_t nonempty = 0;
  for (int g = 0; g < NG; g++) nonempty += gcount[g + 1] > gcount[g] ? 1 : 0;
  if (nonempty <= 1) return target;
  int64_t per_group = std::max<int64_t>(1, (target + nonempty / 2) / nonempty);
  while (per_group > 1 && per_group * nonempty > cap) per_group--;
  return per_group * nonempty;
}

// ---- bins > 256.  The reference takes any --nid_bins (src/calibrate.cpp:175, nid_cost.hpp:23).  The kernels' layouts -- an
// 8-bit bin image, one histogram column of <= 256 cells per LDS tile -- hold 256 bins per axis, and the reference's own data
// path never OCCUPIES more: the camera image is 8-bit (pix = k / 255, visual_camera_calibration.cpp:204) and the LiDAR
// intensities are rank-equalised to floor(256 i / n) / 256 (preprocess.cpp:464-473), so a B x B histogram with B > 256 has at
// most 256 non-empty rows and 256 non-empty columns.  The NID is a function of the MULTISET of cell values and of the row /
// column sums (three entropies; an empty cell, row or column contributes p log(p + eps) = 0 exactly), so relabelling the
// occupied bins 0, 1, 2, ... changes nothing -- not the cost, not the gradient; with the integer accumulation here not even
// the bits.  At creation the occupied image bins and point bins (computed with the CALLER's bin count, the reference's own
// expressions) are collected; if either axis occupies more than 256 the request is refused (never truncated), otherwise the
// handle runs on the compact bins and the getters expand them back to the caller's B x B / B layout.
struct WideBins {
  int user_bins = 0, compact_bins = 0;
  std::vector<uint16_t> lut_img, lut_pts;  // [user_bins]: the caller's bin -> compact bin (unoccupied: 0, never read)
  std::vector<uint16_t> inv_img, inv_pts;  // [occupied]: compact bin -> the caller's bin
};
constexpr int kMaxWideBins = NIDREG_MAX_BINS_WIDE;

int resolve_wide_bins(const nidreg_desc* d, const nidreg_cloud* cloud, WideBins& wb) {
  const int B = d->bins;
  if (d->ext_hist) return fail(NIDREG_ERR_INVALID, "nidreg_create: bins > 256 with a caller-provided histogram buffer (ext_hist) is not supported");
  if (!d->image || d->width < 1 || d->height < 1) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad image");
  if (d->image_row_stride < int64_t(d->width) * (d->image_dtype == NIDREG_IMAGE_F64 ? 8 : 1)) return fail(NIDREG_ERR_INVALID, "nidreg_create: image_row_stride smaller than a row");
  std::vector<unsigned char> used_img(size_t(B), 0), used_pts(size_t(B), 0);
  const bool f64 = d->image_dtype == NIDREG_IMAGE_F64;
  for (int y = 0; y < d->height; y++) {
    const unsigned char* row = static_cast<const unsigned char*>(d->image) + size_t(y) * size_t(d->image_row_stride);
    for (int x = 0; x < d->width; x++) {
      int b;
      if (f64) {
        double v;
        std::memcpy(&v, row + size_t(x) * 8, 8);
        b = std::max(0, std::min(cast_int(v * double(B)), B - 1));  // nid_cost.hpp:78-79 (k_build_bin_image)
      } else {
        b = std::max(0, std::min(B - 1, cast_int(double(row[x]) / 255.0 * double(B))));  // cost_calculator_nid.cpp:43-46
      }
      used_img[size_t(b)] = 1;
    }
  }
  if (cloud) {
    HIP_TRY(hipSetDevice(cloud->device));
    HIP_TRY(mark_bins_device(cloud->d_int, cloud->n, B, used_pts.data()));
  } else {
    for (int64_t i = 0; i < d->num_points; i++) used_pts[size_t(std::max(0, std::min(B - 1, cast_int(d->intensities[i] * double(B)))))] = 1;  // nid_cost.hpp:49
  }
  wb.user_bins = B;
  wb.lut_img.assign(size_t(B), 0);
  wb.lut_pts.assign(size_t(B), 0);
  wb.inv_img.clear();
  wb.inv_pts.clear();
  for (int b = 0; b < B; b++) {
    if (used_img[size_t(b)]) {
      wb.lut_img[size_t(b)] = uint16_t(wb.inv_img.size() & 0xffff);
      wb.inv_img.push_back(uint16_t(b));
    }
    if (used_pts[size_t(b)]) {
      wb.lut_pts[size_t(b)] = uint16_t(wb.inv_pts.size() & 0xffff);
      wb.inv_pts.push_back(uint16_t(b));
    }
  }
  if (wb.inv_img.size() > size_t(NIDREG_MAX_BINS) || wb.inv_pts.size() > size_t(NIDREG_MAX_BINS))
    return fail(NIDREG_ERR_INVALID, "nidreg_create: bins = " + std::to_string(B) + " with " + std::to_string(wb.inv_img.size()) + " occupied image bins and " + std::to_string(wb.inv_pts.size()) +
                                      " occupied intensity bins: more than 256 bins per axis are supported only while at most 256 of them are occupied (8-bit images and 256-level "
                                      "equalised intensities, what the reference's own pipeline produces, always are); refused, not truncated");
  wb.compact_bins = int(std::max<size_t>(2, std::max(wb.inv_img.size(), wb.inv_pts.size())));
  return NIDREG_OK;
}

struct CreateOpts {
  const WideBins* wide = nullptr;  // bins > 256, already resolved by the caller (create_sharded); else create_impl resolves it itself
  // one shard of a ShardSet: built from the column groups [group_lo, group_hi) of `master` (a complete handle of the pair
  // on the owner device) -- its bin image and that slice of its bucketed records are copied device to device
  bool shard = false;
  const nidreg_handle* master = nullptr;
  int group_lo = 0, group_hi = 0;
};

int create_impl(const nidreg_desc* d, const nidreg_cloud* cloud, const double* T_cull, double min_z, int enable_depth, const CreateOpts& opts, nidreg_handle** out) {
  if (!d || !out) return fail(NIDREG_ERR_INVALID, "nidreg_create: null argument");
  *out = nullptr;
  if (d->struct_size != int32_t(sizeof(nidreg_desc))) return fail(NIDREG_ERR_INVALID, "nidreg_create: struct_size mismatch");
  if (d->model_id < 0 || d->model_id > 5) return fail(NIDREG_ERR_INVALID, "nidreg_create: unknown camera model");
  // The reference takes any int (src/calibrate.cpp:175 --nid_bins, nid_cost.hpp:23); its own data path quantises BOTH inputs
  // to 256 levels before the cost sees them -- the camera image is 8-bit (pix = k / 255, visual_camera_calibration.cpp:204),
  // the LiDAR intensities are rank-equalised to floor(256 i / n) / 256 (preprocess.cpp:464-473) -- so more than 256 bins
  // only adds rows and columns that stay empty.  The kernels' layouts (8-bit bin image, one histogram column of <= 256 cells
  // per LDS tile) are built on that bound: refused, not truncated.
  if (d->bins < 2 || d->bins > kMaxWideBins) return fail(NIDREG_ERR_INVALID, "nidreg_create: bins must be in [2, " + std::to_string(kMaxWideBins) + "]");
  if (d->width < 1 || d->height < 1 || (!d->image && !opts.shard)) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad image");
  const nidreg_handle* master = opts.shard ? opts.master : nullptr;
  if (opts.shard && !master) return fail(NIDREG_ERR_INVALID, "nidreg_create: shard without a master");
  // bins > 256: run on the occupied bins, compacted (WideBins above); a shard takes its master's compact layout
  WideBins wide_here;
  const WideBins* wide = opts.wide;
  nidreg_desc dd;
  if (!opts.shard && d->bins > NIDREG_MAX_BINS) {
    if (!wide) {
      if (d->image_dtype != NIDREG_IMAGE_F64 && d->image_dtype != NIDREG_IMAGE_U8) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad image_dtype");
      if (!cloud && (d->num_points < 0 || (d->num_points > 0 && !d->intensities))) return fail(NIDREG_ERR_INVALID, "nidreg_create: null points");
      const int rc = resolve_wide_bins(d, cloud, wide_here);
      if (rc) return rc;
      wide = &wide_here;
    }
    dd = *d;
    dd.bins = wide->compact_bins;
    d = &dd;
  } else if (opts.shard && d->bins > NIDREG_MAX_BINS) {
    return fail(NIDREG_ERR_INVALID, "nidreg_create: a shard is created with its master's compact bin count");
  }
  const int64_t n_in = master ? master->gcount[size_t(opts.group_hi)] - master->gcount[size_t(opts.group_lo)] : (cloud ? cloud->n : d->num_points);
  if (n_in < 0 || n_in > int64_t(INT_MAX)) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad num_points");
  if (!master && !cloud && n_in > 0 && (!d->points || !d->intensities)) return fail(NIDREG_ERR_INVALID, "nidreg_create: null points");
  if (cloud && cloud->device != d->device_id) return fail(NIDREG_ERR_INVALID, "nidreg_create_from_cloud: cloud lives on another device");
  if (d->mode != NIDREG_MODE_SPLINE && d->mode != NIDREG_MODE_NEAREST) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad mode");
  // (NIDREG_PREC_FP32 -- float transform / projection, everything else as now -- existed until round 4: +8 % on the headline, for
  // |dNID| <= 2e-5; a mode that cheap to lose was not worth its kernel instantiations and was removed rather than kept half-built)
  if (d->precision != NIDREG_PREC_FP64)
    return fail(NIDREG_ERR_INVALID, d->precision == NIDREG_PREC_FP32 ? "nidreg_create: NIDREG_PREC_FP32 was removed (it bought 8 %); the core computes in double" : "nidreg_create: bad precision");

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
  h->nearest_exact = (d->flags & NIDREG_FLAG_NEAREST_EXACT) != 0;
  if (wide) {
    h->bins_user = wide->user_bins;
    h->inv_img = wide->inv_img;
    h->inv_pts = wide->inv_pts;
  } else if (master && master->bins_user) {
    h->bins_user = master->bins_user;
    h->inv_img = master->inv_img;
    h->inv_pts = master->inv_pts;
  }
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
  h->lds_hist = d->mode == NIDREG_MODE_NEAREST ? nearest_hist_lds_bytes(B, GW, cshift) : (size_t(GW) * B * 8 << cshift) + size_t(GW) * 8 + 16;
  // gradient pass: a single-column workgroup (GW = 1) keeps ONE copy of its G column (k_spline_grad<.., GW1>)
  h->lds_grad = spline_grad_lds_bytes(B, GW, cshift, false);  // G tile, reduction scratch, phi(q_r), flag (+ staged columns once the table is known to need them)
  h->lds_entropy = size_t(B) * 8 + size_t(GW) * 8 + size_t(kWaves) * 8;

  // ---- fixed point: sum over a bin <= N * 2^frac must stay below 2^63
  const int64_t scaleN = std::max<int64_t>(N, d->scale_points);
  int nbits = 1;
  while ((int64_t(1) << nbits) <= scaleN) nbits++;
  h->frac_bits = d->mode == NIDREG_MODE_NEAREST ? 0 : std::min(40, 62 - nbits);

  // ---- everything below is built ON THE DEVICE from one upload of the caller's arrays (the reference constructs
  // a cost object per pair per outer iteration, visual_camera_calibration.cpp:199-208, so construction time counts):
  // the bin image, then [ViewCulling::cull ->] histogram column + Morton key -> rocPRIM radix sort -> record gather
  // (nid_build.hip).  Temporaries live in the per-device scratch arena.
  const int W = h->W, H = h->H;
  h->pitch = ((W + 8) + 3) & ~3;  // padded width in pixels
  const int PH = H + 3;
  const int nstrips = (PH + 3) / 4 + 1;  // rows are stored in strips of four (nid_device.hpp load_patch)
  const size_t img_bytes = size_t(h->pitch) * 4 * nstrips + 64;
  const bool img_f64 = d->image_dtype == NIDREG_IMAGE_F64;
  if (d->image_dtype != NIDREG_IMAGE_F64 && d->image_dtype != NIDREG_IMAGE_U8) {
    free_handle(h);
    return fail(NIDREG_ERR_INVALID, "nidreg_create: bad image_dtype");
  }
  const size_t src_row = size_t(W) * (img_f64 ? 8 : 1);
  if (d->image_row_stride < int64_t(src_row)) {
    free_handle(h);
    return fail(NIDREG_ERR_INVALID, "nidreg_create: image_row_stride smaller than a row");
  }
  const int64_t pstride = d->point_stride > 0 ? d->point_stride : 32;
  if (!cloud && n_in > 0 && (pstride < 32 || pstride % 8 != 0)) {
    free_handle(h);
    return fail(NIDREG_ERR_INVALID, "nidreg_create: point_stride must be a multiple of 8 and >= 32 ((x y z 1) doubles)");
  }
  std::vector<int64_t> gcount;
  h->img_bytes = img_bytes;
  if (master) {
    // a shard: the master handle (same tiling, same bins) has built the padded bin image and the bucketed, Morton-ordered
    // records on the owner device; this shard takes the records of its column groups and a copy of the image
    if (master->GW != h->GW || master->NG != h->NG || master->bins != B || master->pitch != h->pitch || master->img_bytes != img_bytes) {
      free_handle(h);
      return fail(NIDREG_ERR_INVALID, "nidreg_create: shard / master layout mismatch");
    }
    const size_t rec_bytes = master->rec64 ? sizeof(Rec64) : sizeof(Rec32);
    const int64_t lo = master->gcount[size_t(opts.group_lo)], hi = master->gcount[size_t(opts.group_hi)];
    CREATE_TRY(hipMalloc(&h->d_img, img_bytes));
    CREATE_TRY(hipMemcpyPeer(h->d_img, h->device, master->d_img, master->device, img_bytes));
    CREATE_TRY(hipMalloc(&h->d_pts, std::max<size_t>(size_t(hi - lo), 1) * rec_bytes));
    if (hi > lo) CREATE_TRY(hipMemcpyPeer(h->d_pts, h->device, static_cast<const char*>(master->d_pts) + size_t(lo) * rec_bytes, master->device, size_t(hi - lo) * rec_bytes));
    h->rec64 = master->rec64;
    gcount.assign(master->gcount.size(), 0);
    for (size_t g = 0; g < master->gcount.size(); g++) gcount[g] = std::min(std::max(master->gcount[g], lo), hi) - lo;
    N = hi - lo;
    h->num_points = N;
    h->is_shard = true;
    h->col_lo = std::min(B, opts.group_lo * h->GW);
    h->col_hi = std::min(B, opts.group_hi * h->GW);
  } else
  {
    ScratchArena& arena = ScratchArena::of(h->device);
    std::lock_guard<ScratchArena> guard(arena);
    const size_t up_img = ((src_row * size_t(H)) + 255) & ~size_t(255);
    const size_t up_pts = cloud ? 0 : ((size_t(std::max<int64_t>(n_in, 1)) * 32 + 255) & ~size_t(255));
    const size_t up_int = cloud ? 0 : ((size_t(std::max<int64_t>(n_in, 1)) * 8 + 255) & ~size_t(255));
    const size_t up_lut = wide ? ((size_t(wide->user_bins) * 2 + 255) & ~size_t(255)) : 0;
    CREATE_TRY(arena.reserve(up_img + up_pts + up_int + 2 * up_lut + build_scratch_bytes(n_in, T_cull != nullptr, W, H) + 4096));
    const int Bsrc = wide ? wide->user_bins : B;  // the bin count the caller's values are binned with
    uint16_t *d_lut_img = nullptr, *d_lut_pts = nullptr;
    if (wide) {
      d_lut_img = static_cast<uint16_t*>(arena.carve(up_lut));
      d_lut_pts = static_cast<uint16_t*>(arena.carve(up_lut));
      CREATE_TRY(hipMemcpy(d_lut_img, wide->lut_img.data(), size_t(wide->user_bins) * 2, hipMemcpyHostToDevice));
      CREATE_TRY(hipMemcpy(d_lut_pts, wide->lut_pts.data(), size_t(wide->user_bins) * 2, hipMemcpyHostToDevice));
    }

    // bin image: bin_image = min(int(pix * bins), bins - 1) (nid_cost.hpp:78-79) for CV_64FC1 input;
    // max(0, min(bins-1, int(u8 / 255.0 * bins))) (cost_calculator_nid.cpp:43-46) for CV_8UC1 input;
    // padded by 1 (left/top) and >= 2 (right/bottom), edge replicated (= the clamp of knots_x / knots_y, :70-73)
    void* d_src = arena.carve(up_img);
    CREATE_TRY(hipMalloc(&h->d_img, img_bytes));
    CREATE_TRY(hipMemcpy2D(d_src, src_row, d->image, size_t(d->image_row_stride), src_row, size_t(H), hipMemcpyHostToDevice));
    CREATE_TRY(build_bin_image_device(d_src, img_f64 ? 1 : 0, (long long)src_row, W, H, Bsrc, d_lut_img, h->pitch, nstrips, h->d_img, nullptr));

    // points: bin_points = max(0, min(bins-1, int(intensity * bins))) (nid_cost.hpp:49, cost_calculator_nid.cpp:47)
    // is pose independent -> records are bucketed by column group, so a workgroup owns GW histogram columns;
    // inside a group they follow a Morton curve of the LiDAR-frame bearing (any order gives the same bits -- the
    // sums are integers --, a spatially coherent one makes a wave's gathers share cache lines for ANY pose).
    // Records are float32 when that is lossless (PLY data is float32 at source); otherwise double.
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
    const double* d_cloud_pts = cloud ? cloud->d_pts : nullptr;
    const double* d_cloud_int = cloud ? cloud->d_int : nullptr;
    if (!cloud) {
      double* up_p = static_cast<double*>(arena.carve(up_pts));
      double* up_i = static_cast<double*>(arena.carve(up_int));
      if (n_in > 0) {
        if (pstride == 32) {
          CREATE_TRY(hipMemcpy(up_p, d->points, size_t(n_in) * 32, hipMemcpyHostToDevice));
        } else {
          CREATE_TRY(hipMemcpy2D(up_p, 32, d->points, size_t(pstride), 32, size_t(n_in), hipMemcpyHostToDevice));
        }
        CREATE_TRY(hipMemcpy(up_i, d->intensities, size_t(n_in) * 8, hipMemcpyHostToDevice));
      }
      d_cloud_pts = up_p;
      d_cloud_int = up_i;
    }
    void* recs = nullptr;
    int rec64 = 0;
    CREATE_TRY(build_records_device(d_cloud_pts, d_cloud_int, n_in, T_cull ? &ca : nullptr, Bsrc, d_lut_pts, GW, h->NG, false, (d->flags & NIDREG_FLAG_INPUT_ORDER) != 0,
                                    arena, &recs, &rec64, gcount, nullptr));
    h->d_pts = recs;
    h->rec64 = rec64;
    N = gcount[size_t(h->NG)];
    h->num_points = N;
    CREATE_TRY(hipStreamSynchronize(nullptr));  // the bin-image kernel, before the arena is handed to the next construction
  }

  // ---- chunk tables (split_groups): by default 4 workgroups per CU for the 256-thread kernels, 2 per CU for the WIDE
  // histogram kernel (64 KB LDS each), which therefore has its own table
  {
    std::vector<uint32_t> gend(static_cast<size_t>(h->NG));
    for (int g = 0; g < h->NG; g++) gend[size_t(g)] = uint32_t(gcount[size_t(g) + 1]);
    CREATE_TRY(hipMalloc(&h->d_gend, gend.size() * sizeof(uint32_t)));
    CREATE_TRY(hipMemcpy(h->d_gend, gend.data(), gend.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    int num_cus = 256;
    if (hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || num_cus <= 0) num_cus = 256;
    // (Tried and dropped: emitting a group's parts part-major -- part j of every group in dispatch slot j of the CUs -- and
    // sizing them by slot weights.  The first-dispatched workgroup of a CU does run ~8 % faster than the second, systematically
    // (profiles/archive/r03d_workgroup_spread.txt), but the workgroups of one CU share its issue capacity: what ends a pass is the
    // slowest CU, not the slowest workgroup, and no weighting moved the kernel times (profiles/archive/r03e_slot_weights_no_gain.txt;
    // the part-major order itself cost 3 us in the gradient pass).)
    const int max_segs = max_segments(d->mode, GW);
    auto build_chunks = [&](int target, bool wide_hist, std::vector<Chunk>& chunks) { return split_groups(gcount.data(), h->NG, target, segment_overhead(wide_hist), max_segs, -1, chunks); };
    // workgroups per CU that are really co-resident for THIS kernel instantiation: 4 for the pinhole family, 3 for the
    // fisheye / equirectangular gradient kernels (154-161 VGPRs) -- 1024 chunks there meant 1.33 rounds
    int per_cu_grad = 4, per_cu_hist = h->wide ? 2 : 4;
    if (d->mode == NIDREG_MODE_SPLINE) {
      PassArgs oa;
      fill_pass_args(h, oa);
      const int og = occupancy_spline_grad<double>(oa);
      const int oh = occupancy_spline_hist<double>(oa);
      if (og > 0) per_cu_grad = std::min(og, 8);
      if (oh > 0) per_cu_hist = std::min(oh, 8);
    } else {  // NEAREST: the fast-tier kernels of the wide-angle models hold three (equirectangular) or four waves per SIMD
      PassArgs oa;
      fill_pass_args(h, oa);
      const int on = occupancy_nearest_hist<double>(oa);
      if (on > 0) per_cu_grad = per_cu_hist = std::min(on, 4);
    }
    h->gcount = gcount;
    h->num_cus = num_cus;
    h->per_cu_grad = h->wide ? per_cu_grad : std::min(per_cu_grad, per_cu_hist);
    h->per_cu_hist = per_cu_hist;
    std::vector<Chunk> chunks;
    auto own_target = [&](int per_cu) {
      if (d->target_blocks > 0) return int(d->target_blocks);
      const int64_t full = int64_t(per_cu) * num_cus;
      return int(snap_to_groups(round_chunks(per_cu, num_cus, N), gcount.data(), h->NG, full));
    };
    h->nslots = int(build_chunks(own_target(h->per_cu_grad), false, chunks));
    h->nchunks = int(chunks.size());
    for (const Chunk& c : chunks) h->longest_chunk = std::max<int64_t>(h->longest_chunk, c.count);
    h->seg = h->nslots > h->nchunks ? 1 : 0;
    if (h->seg && GW == 1) h->lds_grad = spline_grad_lds_bytes(B, GW, cshift, true);
    h->chunks_cap = std::max<size_t>(chunks.size(), 1);
    CREATE_TRY(hipMalloc(&h->d_chunks, h->chunks_cap * sizeof(Chunk)));
    if (!chunks.empty()) CREATE_TRY(hipMemcpy(h->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
    if (h->wide) {
      std::vector<Chunk> wide_chunks;
      const int64_t wide_slots = build_chunks(own_target(per_cu_hist), true, wide_chunks);
      h->nchunks_hist = int(wide_chunks.size());
      h->seg_hist = wide_slots > int64_t(wide_chunks.size()) ? 1 : 0;
      h->chunks_hist_cap = std::max<size_t>(wide_chunks.size(), 1);
      CREATE_TRY(hipMalloc(&h->d_chunks_hist, h->chunks_hist_cap * sizeof(Chunk)));
      if (!wide_chunks.empty()) CREATE_TRY(hipMemcpy(h->d_chunks_hist, wide_chunks.data(), wide_chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
    }
  }

  // ---- NEAREST on an equirectangular camera: the pixel-boundary tables of the fast decision tier (nid_kernels.hpp NearestFast).
  // Boundary u = k sits at longitude theta_k = 2 pi (k / W - 1/2), boundary v = j at latitude pi (j / H - 1/2), W and H the
  // INTRINSICS (equirectangular.hpp:14-28 projects with them; the image size only enters the in-image test).
  if (d->mode == NIDREG_MODE_NEAREST && h->model == NIDREG_MODEL_EQUIRECTANGULAR && h->intr[0] >= 8.0 && h->intr[1] >= 8.0 && h->intr[0] <= 65536.0 && h->intr[1] <= 65536.0) {
    const double pi = 3.14159265358979323846;
    h->eq_kmax = int(std::ceil(h->intr[0]));
    h->eq_jmax = int(std::ceil(h->intr[1]));
    std::vector<double> tab(2 * size_t(h->eq_kmax + 1) + size_t(h->eq_jmax + 1));
    for (int k = 0; k <= h->eq_kmax; k++) {
      const double th = 2.0 * pi * (double(k) / h->intr[0] - 0.5);
      tab[2 * size_t(k)] = std::cos(th);
      tab[2 * size_t(k) + 1] = std::sin(th);
    }
    for (int j = 0; j <= h->eq_jmax; j++) {
      const double sj = std::sin(pi * (double(j) / h->intr[1] - 0.5));
      tab[2 * size_t(h->eq_kmax + 1) + size_t(j)] = sj * std::fabs(sj);
    }
    CREATE_TRY(hipMalloc(&h->d_eq_tab, tab.size() * sizeof(double)));
    CREATE_TRY(hipMemcpy(h->d_eq_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
  }

#ifdef NID_EXP_HANDOFF
  {  // EXPERIMENT: the (u, v) hand-off buffer of the process' one handle (leaked at destruction: an experiment build)
    void* uvb = nullptr;
    CREATE_TRY(hipMalloc(&uvb, size_t(std::max<int64_t>(N, 1)) * 16 + 64));
    CREATE_TRY(hipMemset(uvb, 0, size_t(std::max<int64_t>(N, 1)) * 16 + 64));
    CREATE_TRY(set_handoff_buffer(uvb));
  }
#endif
  // ---- per-evaluation scratch
  h->hist_words = nidreg_hist_words(B);
  if (d->ext_stream || (d->flags & NIDREG_FLAG_EXT_STREAM)) {
    h->stream = static_cast<hipStream_t>(d->ext_stream);
  } else {
    CREATE_TRY(pool_stream(h->device, &h->stream));
    h->own_stream = true;
  }
  if (d->ext_hist) {
    h->d_hist = static_cast<u64*>(d->ext_hist);
  } else {
    if (opts.shard) {
      // a shard's two buffers are replicas of the WHOLE pair's histogram: the owners of the other columns store into them from
      // their own devices (nid_kernels.hpp k_entropy_repl) -- fine-grained (coherent) device memory, mapped into every peer
      CREATE_TRY(hipExtMallocWithFlags(reinterpret_cast<void**>(&h->d_hist_buf[0]), size_t(h->hist_words) * sizeof(u64), hipDeviceMallocFinegrained));
      CREATE_TRY(hipExtMallocWithFlags(reinterpret_cast<void**>(&h->d_hist_buf[1]), size_t(h->hist_words) * sizeof(u64), hipDeviceMallocFinegrained));
    } else {
      CREATE_TRY(hipMalloc(&h->d_hist_buf[0], size_t(h->hist_words) * sizeof(u64)));
      CREATE_TRY(hipMalloc(&h->d_hist_buf[1], size_t(h->hist_words) * sizeof(u64)));
    }
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
  {
    // one allocation, carved (256-byte aligned) and zeroed: nidreg_get_hist before the first evaluation then
    // reads zeros, not uninitialised memory
    size_t off = 0;
    auto carve = [&](size_t bytes) {
      const size_t at = off;
      off = (off + bytes + 255) & ~size_t(255);
      return at;
    };
    const size_t o_part_hj = carve(size_t(h->NEB) * sizeof(long long));   // (split-phase ABI scratch: per column block)
    const size_t o_row_part = carve(size_t(h->NEB) * B * sizeof(u64));    // [NEB][B]
    const size_t o_phi_q = carve(size_t(B) * sizeof(double));
    const size_t o_hist_image = carve(size_t(B) * sizeof(double));
    const size_t o_hist_points = carve(size_t(B) * sizeof(double));
    const size_t o_scal = carve(sizeof(EntropyScalars));
    h->partials_cap = partial_slots(h);
    const size_t o_partials = carve(size_t(h->partials_cap) * 12 * sizeof(double));
    const size_t o_counters = carve(8 * sizeof(unsigned int));
    CREATE_TRY(hipMalloc(&h->d_scratch, off));
    CREATE_TRY(hipMemset(h->d_scratch, 0, off));
    char* base = static_cast<char*>(h->d_scratch);
    h->d_part_hj = reinterpret_cast<long long*>(base + o_part_hj);
    h->d_row_part = reinterpret_cast<u64*>(base + o_row_part);
    h->d_phi_q = reinterpret_cast<double*>(base + o_phi_q);
    h->d_hist_image = reinterpret_cast<double*>(base + o_hist_image);
    h->d_hist_points = reinterpret_cast<double*>(base + o_hist_points);
    h->d_scal = reinterpret_cast<EntropyScalars*>(base + o_scal);
    h->d_partials = reinterpret_cast<double*>(base + o_partials);
    h->d_counters = reinterpret_cast<unsigned int*>(base + o_counters);
  }
  {
    void* blk = nullptr;
    CREATE_TRY(pool_host_block(h->device, false, NIDREG_OUT_DOUBLES * sizeof(double), &blk));
    h->h_out = static_cast<double*>(blk);
  }
  std::memset(h->h_out, 0, NIDREG_OUT_DOUBLES * sizeof(double));
  if (!d->ext_out) {
    // results are written straight into host-mapped memory by the finalising workgroups: no D2H copy
    void* dp = nullptr;
    CREATE_TRY(hipHostGetDevicePointer(&dp, h->h_out, 0));
    h->d_out_host = static_cast<double*>(dp);
  }
  for (int i = 0; i < 6; i++) CREATE_TRY(hipEventCreate(&h->ev[i]));
  // The clears above (result block, histogram buffers, scratch incl. the ticket counters) are hipMemset calls on the null
  // stream, which return before they have run (2.9 us per call in the API trace, profiles/archive/r04m_hip_api_stats.csv), and the
  // handle's own stream is non-blocking: the first evaluation must not be able to overtake them.
  CREATE_TRY(hipStreamSynchronize(nullptr));
#undef CREATE_TRY
  *out = h;
  return NIDREG_OK;
}

// ---- several pairs on one GPU: one grid per pass over all pairs' chunks -----------------------------------------
// MultiNIDCost evaluates every pair at the same pose (visual_camera_calibration.cpp:147-173).  Launching three kernels
// per pair makes k small pairs on one GPU launch- and prologue-bound (8 x 1.25M points: 337 us against 160 us for one
// 10M-point pair); a group launches THREE kernels in all, whose combined chunk tables give every pair a share of the
// one round of co-resident workgroups in proportion to its points.  Every pair's cost and histogram are bit-identical to
// evaluating the handles one by one (each pair keeps its own histogram buffers, fixed-point unit, scratch and result block);
// the gradient's workgroup partials follow the group's chunk table: equal up to summation order.
struct MultiGroup {
  std::vector<nidreg_handle*> hs;
  int device = 0;
  hipStream_t stream = nullptr;
  MultiEntry* d_table = nullptr;
  Chunk* d_chunks = nullptr;       // gradient pass / generic histogram kernels
  Chunk* d_chunks_hist = nullptr;  // WIDE histogram kernel
  int seg = 0, seg_hist = 0;       // the combined tables have chunks that run across column groups (SEG kernels)
  size_t lds_grad = 0;
  int nchunks = 0, nchunks_hist = 0;
  std::atomic<int> users{0};  // evaluations running on this group (acquire_group / release_group)
  uint64_t last_use = 0;
};
// The cache of groups: keyed by the exact handle list, at most kMaxGroups entries (least recently used first out), plus a
// short list of handle lists that cannot be grouped (a pair's partial buffer too small for its share of the table) so that
// the chunk tables are not rebuilt on every call.  A group in use is never freed: acquire_group / release_group count the
// evaluations running on it, and a handle's destruction waits for them.
constexpr size_t kMaxGroups = 64, kMaxRejected = 32;  // (a cohort of k members may be evaluated as any of its subsets: 16 entries thrashed at k >= 5)
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
// every evaluation entry point, before it reads the handle's tables
inline void cohort_check(nidreg_handle* h) {
  if (h->cohort && !h->cohort->sealed.load(std::memory_order_acquire)) cohort_seal(h->cohort);
}

bool groupable(const nidreg_handle* a, const nidreg_handle* b);
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
int group_eval(MultiGroup* g, const double* se3, bool want_grad, double* costs, double* grads /* n x 7 or null */, bool* all_ok, int* rcs = nullptr /* per pair */) {
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
        hipLaunchKernelGGL(k_grad_final, dim3(1), dim3(kThreads), 0, g->stream, h->d_partials, 0, se3[0], se3[1], se3[2], se3[3], h->d_out, h->d_out_host, h->seq);
        HIP_TRY(hipGetLastError());
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

// Rendezvous of a cohort's concurrent callers (NIDREG_COHORT=1).  The reference's MultiNIDCost calls its pairs' functors from an
// OpenMP loop at ONE pose (visual_camera_calibration.cpp:161-165); the members of a sealed cohort that arrive in nidreg_eval
// at that pose within NIDREG_COHORT_WAIT_US (default 100) of the first are evaluated as ONE grid per pass -- three launches
// for all of them -- by the first arriver, the others wait for their results.  Which callers make it into a round depends on
// timing; what they get does not: every member's chunk table is fixed by the cohort (cohort_reshape), and the single grid
// of any subset concatenates those tables, so a member's cost, histogram AND gradient are the same bits whether it ran
// alone, with all of its siblings or with some of them (tests/test_gpu_parity.py test_cohort_...).
// Returns kNotJoined when the caller should evaluate by itself (another pose is collecting, or the handle is in the round).
constexpr int kNotJoined = -1000;
// NIDREG_COHORT_TRACE=1: where a round's time goes (leader's clock), printed when the process ends
struct CohortTrace {
  std::atomic<long long> rounds{0}, full{0}, wait_ns{0}, eval_ns{0}, tail_ns{0};
  bool on = false;
  CohortTrace() {
    const char* e = std::getenv("NIDREG_COHORT_TRACE");
    on = e && *e && *e != '0';
  }
  ~CohortTrace() {
    const long long r = rounds.load();
    if (on && r > 0)
      std::fprintf(stderr, "nidreg cohort trace: %lld rounds (%lld with every member), per round: waiting for the siblings %.1f us, evaluation %.1f us, handing the results out %.1f us\n", r, full.load(),
                   1e-3 * double(wait_ns.load()) / double(r), 1e-3 * double(eval_ns.load()) / double(r), 1e-3 * double(tail_ns.load()) / double(r));
  }
};
CohortTrace g_cohort_trace;
inline long long mono_ns() {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (long long)t.tv_sec * 1000000000ll + t.tv_nsec;
}
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
// Measured on the same clouds in one harness (tools/omp_pairs.cpp on the 10M-point scene split into n pairs,
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

// ---- one pair, one process per GPU: the north star's literal form -- "disjoint point slices with a final RCCL all-reduce of
// the 2D histogram over xGMI" -- inside the library.  Every rank creates a plain handle over ITS slice of the cloud
// (desc.scale_points = the pair's total point count: the same fixed-point unit on every rank) and attaches a communicator;
// nidreg_eval / nidreg_eval_iso then run
//     histogram kernel -> ncclAllReduce(int64, sum; hist_words words in place) -> k_entropy (tail) ->
//     gradient kernel  -> ncclAllReduce(float64, sum; 7 words of the result block in place)
// on the handle's stream -- no host synchronisation between the steps.  The histogram is integer, so the all-reduce is exact
// and order independent: every rank computes the same cost, bit for bit, whatever the ring order.  librccl.so is opened at
// run time (dlopen; a process that has torch loaded gets the copy torch already mapped): libnidreg.so does not link it and a
// caller that never attaches a communicator never touches it.
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};
RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {std::getenv("NIDREG_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {
      if (!nm || !*nm) continue;
      api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) {
      const char* e = dlerror();
      api.error = std::string("cannot open librccl.so (set NIDREG_RCCL_LIB): ") + (e ? e : "?");
      return;
    }
    auto sym = [&](const char* nm) {
      void* p = dlsym(api.lib, nm);
      if (!p && api.error.empty()) api.error = std::string("librccl.so has no symbol ") + nm;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return &api;
}
int rccl_fail(const char* what, ncclResult_t r) {
  RcclApi* api = rccl_api();
  return fail(NIDREG_ERR_HIP, std::string(what) + ": " + (api->GetErrorString ? api->GetErrorString(r) : "RCCL error"));
}
#define RCCL_TRY(expr)                                \
  do {                                                \
    const ncclResult_t _r = (expr);                   \
    if (_r != ncclSuccess) return rccl_fail(#expr, _r); \
  } while (0)

void rccl_release(nidreg_handle* h) {
  if (h->rccl_comm && h->rccl_owned) {
    RcclApi* api = rccl_api();
    if (api->CommDestroy) (void)api->CommDestroy(static_cast<ncclComm_t>(h->rccl_comm));
  }
  h->rccl_comm = nullptr;
  h->rccl_owned = false;
}

// Every rank of the communicator must run the SAME fixed-point histogram: the all-reduce adds the ranks' int64 words as they are.
// One collective at attach time -- max over the ranks of {v, -v} for the handle's table parameters -- and a refusal on every rank
// alike when they differ (a rank created without desc.scale_points = the pair's total, or with other bins / mode, would
// otherwise produce a silently wrong cost, identical on all ranks).
int rccl_check_agreement(nidreg_handle* h, ncclComm_t comm, const char* who) {
  RcclApi* api = rccl_api();
  const int kN = 5;
  long long v[2 * kN] = {h->frac_bits, h->bins, (long long)h->hist_words, h->mode, h->bins_user};
  for (int k = 0; k < kN; k++) v[kN + k] = -v[k];
  long long* d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(v)));
  hipError_t e = hipMemcpyAsync(d, v, sizeof(v), hipMemcpyHostToDevice, h->stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) r = api->AllReduce(d, d, size_t(2 * kN), ncclInt64, ncclMax, comm, h->stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(v, d, sizeof(v), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d);
  if (r != ncclSuccess) return rccl_fail(who, r);
  if (e != hipSuccess) return fail(NIDREG_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
  static const char* names[kN] = {"frac_bits (desc.scale_points must be the pair's TOTAL point count on every rank)", "bins", "hist_words", "mode", "bins (caller's count)"};
  for (int k = 0; k < kN; k++)
    if (v[k] != -v[kN + k])
      return fail(NIDREG_ERR_INVALID, std::string(who) + ": the ranks of the communicator disagree on " + names[k] + ": max " + std::to_string(v[k]) + ", min " + std::to_string(-v[kN + k]) +
                                        "; the handle stays detached");
  return NIDREG_OK;
}

int rccl_attachable(const nidreg_handle* h, const char* who) {
  if (!h) return fail(NIDREG_ERR_INVALID, std::string(who) + ": null handle");
  if (h->set || h->is_shard) return fail(NIDREG_ERR_INVALID, std::string(who) + ": the handle is already sharded inside the library (desc.device_ids / NIDREG_DEVICES)");
  if (!h->own_hist || !h->own_out) return fail(NIDREG_ERR_INVALID, std::string(who) + ": the handle must own its histogram and result buffers (no ext_hist / ext_out)");
  if (h->async_outstanding != 0) return fail(NIDREG_ERR_INVALID, std::string(who) + ": collect the handle's outstanding tickets first");
  return NIDREG_OK;
}

// one evaluation of a handle with a communicator; mode SPLINE: pose = se3[7], NEAREST: row-major 4x4.  Collective: every rank
// of the communicator calls it with the same pose.
int rccl_eval(nidreg_handle* h, int mode, const double* pose, double* cost, double* grad7) {
  if (h->mode != mode) return fail(NIDREG_ERR_INVALID, mode == NIDREG_MODE_SPLINE ? "nidreg_eval: handle was created in NEAREST mode" : "nidreg_eval_iso: handle was created in SPLINE mode");
  RcclApi* api = rccl_api();
  ncclComm_t comm = static_cast<ncclComm_t>(h->rccl_comm);
  HIP_TRY(hipSetDevice(h->device));
  InflightGuard guard(h->device);
  bump_seq(h);
  const bool grad = mode == NIDREG_MODE_SPLINE && grad7 != nullptr;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  if (h->num_points == 0) {  // a rank without points launches no histogram kernel: its (cleared) buffer still takes part in the sum
    HIP_TRY(begin_histogram(h));
    if (h->timing == 1) HIP_TRY(hipEventRecord(h->ev[1], h->stream));  // (the marker launch_hist_* would have recorded: nidreg_get_timing reads it)
    if (mode == NIDREG_MODE_SPLINE) {
      for (int k = 0; k < 4; k++) h->last_q[k] = pose[k];
      pose_from_se3(pose, h->last_R, h->last_t);
    }
  } else {
    const int rc = mode == NIDREG_MODE_SPLINE ? launch_hist_spline(h, pose, guard.alone) : launch_hist_nearest(h, pose);
    if (rc) return rc;
  }
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[2], h->stream));
  RCCL_TRY(api->AllReduce(h->d_hist, h->d_hist, size_t(h->hist_words), ncclInt64, ncclSum, comm, h->stream));
  int rc = launch_entropy(h, 0.0, true);
  if (rc) return rc;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[3], h->stream));
  if (grad) {
    rc = launch_grad(h, guard.alone, 0);
    if (rc) return rc;
    RCCL_TRY(api->AllReduce(h->d_out + 1, h->d_out + 1, 7, ncclFloat64, ncclSum, comm, h->stream));
  } else if (h->timing) {
    HIP_TRY(hipEventRecord(h->ev[4], h->stream));
  }
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[5], h->stream));
  h->ev_grad = grad;
  HIP_TRY(hipMemcpyAsync(h->h_out, h->d_out, NIDREG_OUT_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (cost) *cost = h->h_out[0];
  if (grad)
    for (int k = 0; k < 7; k++) grad7[k] = h->h_out[1 + k];
  return h->h_out[8] != 0.0 ? NIDREG_FALSE : NIDREG_OK;
}

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

}  // namespace

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
