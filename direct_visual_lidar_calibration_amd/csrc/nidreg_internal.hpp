// nidreg_internal.hpp -- what the translation units of libnidreg.so's host side share: the handle and its companions, the
// per-process tables, and the prototypes of every function that crosses a file boundary.  (Until round 5 all of this was ONE
// 3 400-line file, nidreg.hip; round 6 split it along its section comments, no behaviour change, same exported symbols:
//   nidreg_core.hip   pools, pass arguments, launches, the fused route, completion
//   nidreg_plan.hip   chunk tables, bins > 256, handle construction
//   nidreg_multi.hip  several pairs as one grid, cohorts
//   nidreg_rccl.hip   RCCL inside the library
//   nidreg_shard.hip  one pair over several GPUs by one process
//   nidreg_abi.hip    the extern "C" entry points of include/nidreg.h)
#pragma once
// The host side of the C ABI declared in include/nidreg.h
// owns: device residency of one LiDAR-camera pair (bucketed point records, padded bin image,
// fixed-point histogram, scratch), the per-evaluation launch sequence, and the multi-handle
// (multi-pair / multi-GPU) fan-out.  No CPU compute path exists here: every evaluation runs the
// HIP kernels of nid_kernels.hpp, and creation fails when no gfx950 device is usable.
// (NID_COMMON_KERNELS / NID_SHARD_KERNELS / NID_FINAL_KERNEL -- which non-point kernels of nid_kernels.hpp a translation unit instantiates --
// are defined by the .hip file BEFORE it includes this header)
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

extern thread_local std::string g_last_error;  // nidreg_last_error() of the calling thread (nidreg_core.hip; set by nidreg::fail)

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


// device-resident cloud: uploaded once per pair, re-culled / re-bucketed on the GPU for every handle
struct nidreg_cloud {
  int device = 0;
  int64_t n = 0;
  double* d_pts = nullptr;  // n x 4 doubles (x y z 1)
  double* d_int = nullptr;  // n doubles
};

// ---- everything below: declarations shared by the translation units of libnidreg.so's host side (nidreg_core / _plan / _multi / _rccl / _shard / _abi .hip)
namespace nidreg_detail {

extern std::atomic<int> g_inflight[NIDREG_MAX_DEVICES];

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
extern ResourcePool g_pool[NIDREG_MAX_DEVICES];
constexpr size_t kPoolCap = 64;
// staging of nidreg_project for a handful of points: [3 n | 2 n | 6 n] doubles, host-mapped, one block per device
constexpr int64_t kSmallProject = 64;
struct SmallProject {
  std::mutex mu;
  double* host = nullptr;
  double* dev = nullptr;
};
extern SmallProject g_small_project[NIDREG_MAX_DEVICES];

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

struct CreateOpts {
  const WideBins* wide = nullptr;  // bins > 256, already resolved by the caller (create_sharded); else create_impl resolves it itself
  // one shard of a ShardSet: built from the column groups [group_lo, group_hi) of `master` (a complete handle of the pair
  // on the owner device) -- its bin image and that slice of its bucketed records are copied device to device
  bool shard = false;
  const nidreg_handle* master = nullptr;
  int group_lo = 0, group_hi = 0;
};

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
extern std::mutex g_groups_mu;
extern std::vector<MultiGroup*> g_groups;
extern std::vector<std::vector<nidreg_handle*>> g_rejected;
extern uint64_t g_group_clock;
extern std::mutex g_cohort_mu;
extern Cohort* g_open_cohort[NIDREG_MAX_DEVICES];

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
extern CohortTrace g_cohort_trace;

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
#define RCCL_TRY(expr)                                \
  do {                                                \
    const ncclResult_t _r = (expr);                   \
    if (_r != ncclSuccess) return rccl_fail(#expr, _r); \
  } while (0)
extern std::mutex g_shard_device_mu[NIDREG_MAX_DEVICES];

hipError_t pool_stream(int device, hipStream_t* out);
void unpool_stream(int device, hipStream_t s);
hipError_t pool_host_block(int device, bool ring, size_t bytes, void** out);
void unpool_host_block(int device, bool ring, void* p);
void pool_release(int device);
void free_handle(nidreg_handle* h);
void fill_pass_args(const nidreg_handle* h, PassArgs& a);
void pose_from_se3(const double* se3, double* R, double* t);
hipError_t begin_histogram(nidreg_handle* h, hipStream_t stream);
hipError_t begin_histogram(nidreg_handle* h);
int launch_hist_spline(nidreg_handle* h, const double* se3, bool alone = false);
NearestFastArgs nearest_fast_args(const nidreg_handle* h, const double* T);
int launch_hist_nearest(nidreg_handle* h, const double* T);
int launch_entropy(nidreg_handle* h, double tag, bool tail = true);
bool grad_sums_table(const nidreg_handle* h);
int launch_grad(nidreg_handle* h, bool alone = false, int from_partials = 0);
hipError_t launch_grad_final(hipStream_t stream, const double* partials, const double* q4, double* out, double* out_host, double tag);
int eval_launch_first(nidreg_handle* h, const double* se3, bool alone = false);
int eval_launch_rest(nidreg_handle* h, bool want_grad, bool alone = false);
void plan_fused(nidreg_handle* h);
bool fused_planned(nidreg_handle* h);
bool fused_usable(nidreg_handle* h);
void fused_give_up(nidreg_handle* h);
int eval_launch_fused(nidreg_handle* h, const double* se3);
int eval_launch(nidreg_handle* h, const double* se3, bool want_grad, bool alone = false);
int eval_finish_on(nidreg_handle* h, hipStream_t stream, double* cost, double* grad7);
int eval_finish(nidreg_handle* h, double* cost, double* grad7);
int eval_finish_block(nidreg_handle* h, hipStream_t stream, const double* block, uint64_t seq_bits, bool polled, double* cost, double* grad7);
int eval_one(nidreg_handle* h, const double* se3, double* cost, double* grad7);
int iso_launch(nidreg_handle* h, const double* T);
bool trust_gate_ok(const double* init, const double* se3);
int64_t fill_chunks(const int64_t* gcount, int NG, int64_t C, int64_t overhead, int max_segs, int pair, std::vector<Chunk>* out, int64_t* nslots_out = nullptr);
int64_t best_bound(const int64_t* gcount, int NG, int64_t target, int64_t overhead, int max_segs, int64_t N, int64_t nonempty, int64_t* nchunks_out);
int64_t split_groups(const int64_t* gcount, int NG, int64_t target, int64_t overhead, int max_segs, int pair, std::vector<Chunk>& chunks);
int max_segments(int mode, int GW);
int64_t segment_overhead(bool wide_hist);
int64_t round_chunks(int per_cu, int num_cus, int64_t points);
int64_t snap_to_groups(int64_t target, const int64_t* gcount, int NG, int64_t cap);
int resolve_wide_bins(const nidreg_desc* d, const nidreg_cloud* cloud, WideBins& wb);
int create_impl(const nidreg_desc* d, const nidreg_cloud* cloud, const double* T_cull, double min_z, int enable_depth, const CreateOpts& opts, nidreg_handle** out);
void release_group(MultiGroup* g);
void free_group(MultiGroup* g);
void drop_groups_of(const nidreg_handle* h);
bool groupable(const nidreg_handle* a, const nidreg_handle* b);
int64_t pair_chunks(const nidreg_handle* h, int pair, int64_t target, bool wide_hist, std::vector<Chunk>& chunks);
bool cohorts_enabled();
int cohort_reshape(nidreg_handle* h, int64_t total_points);
void cohort_seal(Cohort* c);
void cohort_join(nidreg_handle* h);
void cohort_leave(nidreg_handle* h);
MultiGroup* find_or_make_group(nidreg_handle* const* handles, int n);
int group_eval(MultiGroup* g, const double* se3, bool want_grad, double* costs, double* grads /* n x 7 or null */, bool* all_ok, int* rcs = nullptr /* per pair */);
int cohort_eval(nidreg_handle* h, const double* se3, bool want_grad, double* cost, double* grad7);
int group_eval_iso(MultiGroup* g, const double* T, double* costs);
int multi_grid_min();
bool can_group(nidreg_handle* const* handles, int n);
RcclApi* rccl_api();
int rccl_fail(const char* what, ncclResult_t r);
void rccl_release(nidreg_handle* h);
int rccl_check_agreement(nidreg_handle* h, ncclComm_t comm, const char* who);
int rccl_attachable(const nidreg_handle* h, const char* who);
int rccl_eval(nidreg_handle* h, int mode, const double* pose, double* cost, double* grad7);
std::vector<int> shard_devices(const nidreg_desc* d);
bool wants_shards(const nidreg_desc* d);
int shard_launch_phase(ShardSet* set, int g, int phase, bool alone);
int shard_finish(ShardSet* set, int g);
int run_shard(ShardSet* set, int g);
void shard_worker(ShardSet* set, int g);
int set_eval(ShardSet* set, int mode, const double* pose, double* cost, double* grad7);
void free_shard_set(ShardSet* set);
std::vector<int> partition_groups(const std::vector<int64_t>& gcount, int NG, int n);
int create_sharded(const nidreg_desc* d, const nidreg_cloud* cloud, const double* T_cull, double min_z, int enable_depth, nidreg_handle** out);



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

inline void bump_seq(nidreg_handle* h) {
  h->last_stream = h->stream;  // (a multi-pair group overrides this after the call)
  h->seq += 1.0;
  std::memcpy(&h->seq_bits, &h->seq, sizeof(h->seq_bits));
}
// every evaluation entry point, before it reads the handle's tables
inline void cohort_check(nidreg_handle* h) {
  if (h->cohort && !h->cohort->sealed.load(std::memory_order_acquire)) cohort_seal(h->cohort);
}
inline long long mono_ns() {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (long long)t.tv_sec * 1000000000ll + t.tv_nsec;
}

}  // namespace nidreg_detail
using namespace nidreg_detail;

