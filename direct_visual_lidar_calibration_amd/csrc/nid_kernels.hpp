// nid_kernels.hpp -- the HIP kernels of the NID registration core (gfx950, wave64).
//
// One NIDCost evaluation (include/vlcal/costs/nid_cost.hpp:36-107 redesigned) -- the histogram buffer is zero on entry:
// double-buffered, cleared by the previous evaluation's kernels; a caller-provided buffer is cleared with a memset:
//   k_spline_hist   <model,rec,real,WIDE,MULTI,SEG>  stream the point records once; LDS-tiled fixed-point joint histogram
//   k_entropy       <MULTI>                          per block of 8 columns: sum p log(p + eps) (fixed point) and row sums, added
//                                                    behind the histogram; cost-only evaluations: its last workgroup runs the
//                                                    entropy tail (H_image, H_points, H_joint -> NID).  Not launched for
//                                                    tables of <= 1024 cells in a cost+Jacobian evaluation (kSelfEntropyCells)
//   k_spline_grad   <model,rec,real,GW1,MULTI,SEG>   prologue: the entropy tail on those sums (every workgroup, same integers),
//                                                    the G tile; stream the records again; contract dNID/dh with
//                                                    d(weight)/d(p_cam), fold d(p_cam)/d(pose) into a 3x3 + 3 accumulator per
//                                                    segment; last workgroup: reduce the partials, chain to
//                                                    d/d[qx qy qz qw tx ty tz], results + completion tag to the host
// One CostCalculatorNID evaluation (src/vlcal/calib/cost_calculator_nid.cpp:21-67):  k_nearest_hist; k_entropy (tail).
// A pair spread over several GPUs by one process:  k_*_hist; k_entropy_repl (k_entropy + one exchange); k_spline_grad.
// k_grad_final: stand-alone finalisation, launched only for an empty cloud (no gradient workgroup exists to do it).
//
// Point kernels: 256 threads (4 waves of 64; the WIDE histogram kernel 512); dynamic LDS only, base 16-B aligned.
#pragma once
#include "nid_device.hpp"
#include "nid_multi.hpp"

namespace nidreg {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
#ifdef NID_EXP_HANDOFF
// EXPERIMENT (VERDICT r4 #6 ii; profiles/r05_experiments.md): the histogram pass hands every record's projected (u, v) to the
// gradient pass as an fp64 pair, 16 B / point, instead of the gradient pass recomputing the projection's value.  One buffer per
// process (the experiment runs one handle), set by nidreg::set_handoff_buffer.
static __device__ double2* g_uv_handoff;
#endif
#ifndef NID_UNROLL
#define NID_UNROLL 4
#endif
constexpr int kUnroll = NID_UNROLL;  // point records in flight per thread

// tail words behind the B*B joint histogram
constexpr int kTailInliers = 0;  // number of inlier points (plain count)
constexpr int kTailHj = 1;       // sum over all cells of p log(p + eps), fixed point (ent_fixed), accumulated by k_entropy
constexpr int kTailWords = 8;     // followed by B column sums (sum_r h[c][r], fixed point) written by the flush,
                                  // then (from the next multiple of 8) B row sums (sum_c h[c][r]) accumulated by k_entropy
__host__ __device__ __forceinline__ size_t hist_row_sums_at(int B) { return size_t(B) * size_t(B) + kTailWords + size_t((B + 7) & ~7); }

// scalars written by k_entropy_final for k_spline_grad / the host
struct EntropyScalars {
  double nid;      // the cost
  double S;        // inlier count
  double coefA;    // -(Hi + Hp) / (Hj^2 * S)
  double coefB;    // 1 / (Hj * S)
  double Hi, Hp, Hj;
  double status;   // 0 = finite, 1 = non-finite NID (functor returns false)
};

#ifdef NID_STAMP
// development aid (tools/wg_timeline.hip): per-workgroup start / end wall-clock stamps (100 MHz) and hardware ids
__device__ unsigned long long g_stamp[4 * 8192];
__device__ __forceinline__ void stamp_begin() {
  if (threadIdx.x == 0) {
    g_stamp[4 * blockIdx.x + 0] = wall_clock64();
    g_stamp[4 * blockIdx.x + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
  }
}
__device__ __forceinline__ void stamp_end() {
  if (threadIdx.x == 0) g_stamp[4 * blockIdx.x + 1] = wall_clock64();
}
// stage stamps of one workgroup (tools/stage_times.py): slot i of workgroup b = g_stage[8 b + i]
__device__ unsigned long long g_stage[8 * 8192];
__device__ __forceinline__ void stamp_stage(int i) {
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_stage[8 * blockIdx.x + i] = wall_clock64();
}
#else
__device__ __forceinline__ void stamp_begin() {}
__device__ __forceinline__ void stamp_end() {}
__device__ __forceinline__ void stamp_stage(int) {}
#endif

// Sum over the 64 lanes of a wave, returned to every lane.  Cross-lane moves by DPP (v_mov_b32_dpp on the two halves of
// the double: row_shr 1 / 2 / 4 / 8 with out-of-row sources reading 0, then row_bcast:15 into rows 1 and 3 and row_bcast:31
// into rows 2 and 3) instead of __shfl_down's ds_bpermute pairs: 18 VALU instructions and two v_readlane per sum, no LDS
// crossbar traffic -- the gradient kernel's epilogue runs twelve of these in every wave (144 ds_bpermute before, all of a
// CU's workgroups at about the same time).  Fixed association: lane 63 ends up with ((rows 0 + 1) + (rows 2 + 3)).
#ifndef NID_WAVE_SUM_SHFL
template <int CTRL, int ROW_MASK, bool BOUND_ZERO>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
  return __hiloint2double(hi, lo);  // lanes the control does not write (disabled rows, out-of-row sources) get +0.0
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0x111, 0xf, true>(v);   // row_shr:1
  v += dpp_move<0x112, 0xf, true>(v);   // row_shr:2
  v += dpp_move<0x114, 0xf, true>(v);   // row_shr:4
  v += dpp_move<0x118, 0xf, true>(v);   // row_shr:8 -> lane 15 of every row holds its row's sum
  v += dpp_move<0x142, 0xa, false>(v);  // row_bcast:15 -> rows 1 and 3
  v += dpp_move<0x143, 0xc, false>(v);  // row_bcast:31 -> rows 2 and 3: lane 63 holds the total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
// the same for a 32-bit count (the inlier counters of the histogram passes)
__device__ __forceinline__ unsigned int wave_sum(unsigned int v) {
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x111, 0xf, 0xf, true));
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x112, 0xf, 0xf, true));
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x114, 0xf, 0xf, true));
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x118, 0xf, 0xf, true));
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x142, 0xa, 0xf, false));
  v += unsigned(__builtin_amdgcn_update_dpp(0, int(v), 0x143, 0xc, 0xf, false));
  return unsigned(__builtin_amdgcn_readlane(int(v), 63));
}
#else
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ unsigned int wave_sum(unsigned int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
#endif

// the same for a 64-bit integer (fixed-point entropy partials, row sums): integer sums are order independent
#ifndef NID_WAVE_SUM_SHFL
template <int CTRL, int ROW_MASK, bool BOUND_ZERO>
__device__ __forceinline__ long long dpp_move_i64(long long v) {
  const int lo = __builtin_amdgcn_update_dpp(0, int(uint32_t(u64(v))), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
  const int hi = __builtin_amdgcn_update_dpp(0, int(uint32_t(u64(v) >> 32)), CTRL, ROW_MASK, 0xf, BOUND_ZERO);
  return (long long)((u64(uint32_t(hi)) << 32) | u64(uint32_t(lo)));
}
__device__ __forceinline__ long long wave_sum(long long v) {
  v += dpp_move_i64<0x111, 0xf, true>(v);
  v += dpp_move_i64<0x112, 0xf, true>(v);
  v += dpp_move_i64<0x114, 0xf, true>(v);
  v += dpp_move_i64<0x118, 0xf, true>(v);
  v += dpp_move_i64<0x142, 0xa, false>(v);
  v += dpp_move_i64<0x143, 0xc, false>(v);
  const int lo = __builtin_amdgcn_readlane(int(uint32_t(u64(v))), 63), hi = __builtin_amdgcn_readlane(int(uint32_t(u64(v) >> 32)), 63);
  return (long long)((u64(uint32_t(hi)) << 32) | u64(uint32_t(lo)));
}
#else
__device__ __forceinline__ long long wave_sum(long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
#endif

// Entropy terms t = p log(p + 1e-6) (|t| < 0.37; sum over a distribution <= log(bins^2) < 12) are accumulated as 64-bit
// FIXED POINT with 50 fractional bits: t + 6 lies in [4, 8), where a double's ulp is 2^-50, so bits(t + 6) - bits(6) is
// round(t 2^50) -- one fp64 add and one 64-bit integer subtract.  Integer sums do not depend on their order, so every
// decomposition of the entropy work (k_entropy's column blocks, the shards of a pair spread over
// several GPUs, a multi-pair group) gives the SAME three entropies bit for bit, hence the same NID.  Quantisation: <= 2^-51
// per term, 65 536 terms -> |dH| < 3e-11 worst case (1e-13 typical) on H ~ 5-10, far inside the 1e-10 parity bar.
__device__ __forceinline__ long long ent_fixed(double t) { return __double_as_longlong(t + 6.0) - __double_as_longlong(6.0); }
__device__ __forceinline__ double ent_value(long long k) { return double(k) * 0x1p-50; }

// ------------------------------------------------------------------------------------------
// "last workgroup finalises" hand-off (cdna_hip_programming.md Guideline 16): every workgroup's
// stores -> __syncthreads -> one lane: agent-scope release, drained vmcnt, relaxed agent-scope ticket.
// The workgroup that draws the last ticket does one agent-scope acquire (invalidates its L1) and may
// then read everybody's partials with plain loads.  The counter is reset for the next launch.
// s_flag must be an LDS word owned by the caller.
// WRITE_THROUGH = the payload was stored by wave 0 with agent-scope (sc1, write-through) stores, so
// no release fence is needed -- a buffer_wbl2 per workgroup costs ~2 us and, with thousands of
// workgroups, made the fused gradient kernel 50 % slower than a separate finalisation launch.
// ACQUIRE = false: the caller reads the others' payload with agent-scope loads instead (nid_fused.hpp: no L2 invalidation).
template <bool WRITE_THROUGH, bool ACQUIRE = true>
__device__ __forceinline__ bool last_workgroup_arrives(unsigned int* counter, unsigned int nblocks, int* s_flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!WRITE_THROUGH) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == nblocks - 1u) ? 1 : 0;
    if (last) {
      if (ACQUIRE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *s_flag = last;
  }
  __syncthreads();
  return *s_flag != 0;
}

// reduce workgroup partials ([12][nblocks], fixed order -> run-to-run reproducible) and chain M, gt to
// the ambient 7-gradient of p_cam = p + 2 w (v x p) + 2 v x (v x p) + t  (SURVEY.md Appendix C):
//   A = (M21 - M12, M02 - M20, M10 - M01);  grad_w = 2 v.A;
//   grad_v = 2 w A + 2 (M v + M^T v - 2 tr(M) v);  grad_t = gt.
// out[1..7] = d NID / d [qx qy qz qw tx ty tz]; out_host (nullable) = host-mapped mirror.
// s_red: kWaves * 12 doubles of LDS.
// COH: the partials are read with agent-scope loads (written by this kernel's other workgroups with agent-scope stores; no acquire fence)
template <int kT, bool COH = false>
__device__ __forceinline__ void grad_final_body(const double* partials, int nblocks, double qx, double qy, double qz, double qw, double* out, double* out_host, double tag, double* s_red,
                                                const double* s_fin = nullptr) {
  const int tid = threadIdx.x;
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.0;
  // (Round 6 tried four rounds of these 12 loads in flight per thread -- the full round's 1024 partials take four dependent trips to
  // the coherence point here --: the finalising workgroup's stage 4.9 -> 4.2 us at 10M points, the gradient kernel 73.9 -> 73.4 us,
  // and +1 us on a 100k-point evaluation (the predicated rounds still issue); profiles/r06_experiments.md section 5.  One round per trip.)
  for (int b = tid; b < nblocks; b += kT) {
#pragma unroll
    for (int k = 0; k < 12; k++) acc[k] += COH ? __hip_atomic_load(&partials[size_t(k) * nblocks + b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : partials[size_t(k) * nblocks + b];
  }
  __syncthreads();  // s_red may still be in use by the caller's own reduction
#pragma unroll
  for (int k = 0; k < 12; k++) {
    const double t = wave_sum(acc[k]);
    if ((tid & 63) == 0) s_red[(tid >> 6) * 12 + k] = t;
  }
  __syncthreads();
  if (tid == 0) {
    double M[12];
    for (int k = 0; k < 12; k++) {
      double t = 0.0;
      for (int w = 0; w < kT / 64; w++) t += s_red[w * 12 + k];
      M[k] = t;
    }
    const double A0 = M[7] - M[5], A1 = M[2] - M[6], A2 = M[3] - M[1];
    const double tr = M[0] + M[4] + M[8];
    const double Mv0 = M[0] * qx + M[1] * qy + M[2] * qz;
    const double Mv1 = M[3] * qx + M[4] * qy + M[5] * qz;
    const double Mv2 = M[6] * qx + M[7] * qy + M[8] * qz;
    const double Mtv0 = M[0] * qx + M[3] * qy + M[6] * qz;
    const double Mtv1 = M[1] * qx + M[4] * qy + M[7] * qz;
    const double Mtv2 = M[2] * qx + M[5] * qy + M[8] * qz;
    double g[7];
    g[0] = 2.0 * qw * A0 + 2.0 * (Mv0 + Mtv0 - 2.0 * tr * qx);
    g[1] = 2.0 * qw * A1 + 2.0 * (Mv1 + Mtv1 - 2.0 * tr * qy);
    g[2] = 2.0 * qw * A2 + 2.0 * (Mv2 + Mtv2 - 2.0 * tr * qz);
    g[3] = 2.0 * (qx * A0 + qy * A1 + qz * A2);
    g[4] = M[9];
    g[5] = M[10];
    g[6] = M[11];
    for (int k = 0; k < 7; k++) out[1 + k] = g[k];
    if (out_host) {
      for (int k = 0; k < 7; k++) out_host[1 + k] = g[k];
      // cost / status / inlier count: k_entropy's own tail has mirrored them already; when the gradient kernel runs the tail
      // (grad_scalars_from_partials) another workgroup of THIS kernel wrote them to `out` (agent scope, before its ticket)
      // and this is their way to the host --
      // every host-visible word of an evaluation is then written by ONE thread, in order, ahead of the tag.  (Round 4 let the
      // pair's first workgroup write the three words to the host itself, ahead of its ticket, to save this dependent round
      // trip -- 3.0 -> 2.7 us for the final stage: with five callers on the GPU a host saw the tag BEFORE the cost once in a
      // full suite -- test_concurrent_callers, the previous pose's cost.  Writes of different workgroups (different XCDs) to
      // one host block are not ordered on their way to the host by the writer's s_waitcnt + the ticket.  Reverted.)
      // Round 5: when this kernel ran the tail, the finalising workgroup holds the same three values itself (s_fin: computed in its
      // own prologue from the same integers) -- no re-read.
      if (s_fin) {
        out_host[0] = s_fin[0];
        out_host[8] = s_fin[1];
        out_host[9] = s_fin[2];
      } else {
        out_host[0] = __hip_atomic_load(&out[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out_host[8] = __hip_atomic_load(&out[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        out_host[9] = __hip_atomic_load(&out[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __threadfence_system();
      // completion tag of this evaluation: the host polls this word instead of synchronising the stream
      __hip_atomic_store(&out_host[15], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Issue priority by progress (s_setprio).  The SIMD's arbiter serves the oldest ready wave first, so of the workgroups
// co-resident on a CU the first-dispatched one runs at its solo speed and the last one is left to finish alone, at
// single-wave issue rate (PMC: on average only 2.4 of the 4 resident waves per SIMD are alive over the gradient kernel).
// A wave that has done less of its chunk gets the higher priority: the waves of a SIMD stay within a quarter of a chunk
// of each other.  Measured on cfg 2 (profiles/archive/r02g_variants_prio.txt): histogram pass 57.5 -> 53.0 us, gradient pass
// 84.0 -> 78.6 us; the thresholds hardly matter (quarters, 1/2-3/4-7/8, per point instead of per batch: all within 1 us); a
// purely phase-based toggle (either direction) or a rotating priority gain half as much -- what helps is that co-resident
// waves stop being served strictly by age.
// `done` / `total` are uniform: scalar compares only.  -DNID_NO_PRIO builds the kernels without it (A/B runs).
// `on` (uniform, a kernel argument): the host sets it when the evaluation is the only one in flight on its device -- with
// several callers' kernels sharing the GPU (the reference's OpenMP loop over pairs, visual_camera_calibration.cpp:161) the
// rule made the last kernels 5-16 % slower (profiles/archive/r02h_multi_pair_threads.txt), so they run without it.
__device__ __forceinline__ void set_progress_priority(bool on, uint32_t done, uint32_t total) {
#ifndef NID_NO_PRIO
  if (!on) return;
  const uint32_t quarter = total >> 2;
  if (done < quarter) __builtin_amdgcn_s_setprio(3);
  else if (done < 2u * quarter) __builtin_amdgcn_s_setprio(2);
  else if (done < 3u * quarter) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
#endif
}

// ---- one pair spread over several GPUs (protocol: see the k_entropy_repl section below)
constexpr int kMaxShards = 16;
constexpr int kShardBlockFlags = 256;  // one flag per entropy column block (at most B / 1 = 256 of them)
struct ShardTable {            // device-resident, one per shard, constant over the set's lifetime
  u64* flags[kMaxShards];      // flag block of every shard (fine-grained): [kShardBlockFlags] "column block j has arrived", then the
                               // self-test's [2][kMaxShards] words
  u64* gather[kMaxShards];     // gather block of every shard (fine-grained): the self-test's payload words
  u64* hist[kMaxShards][2];    // both histogram buffers of every shard (fine-grained): an owner stores its columns into all of them
  int n, me;
  int col_lo, col_hi;          // histogram columns this shard owns (= cut[me], cut[me + 1])
  int cut[kMaxShards + 1];     // shard g owns the columns [cut[g], cut[g + 1]); every cut is a multiple of CB
  int CB;                      // columns per entropy block
};
constexpr int kGatherS = 0;                      // [kMaxShards] inlier count of shard g
constexpr int kGatherTest = kMaxShards;          // [kMaxShards] self-test payload
constexpr int kGatherWords = 2 * kMaxShards;
constexpr int kFlagTest = kShardBlockFlags;      // [2][kMaxShards] self-test flags
constexpr int kFlagWords = kShardBlockFlags + 2 * kMaxShards;

__device__ __forceinline__ u64 load_sys(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void store_sys(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// waits until flag word `p` reaches `seq`; false after ~timeout_ticks of the 100 MHz wall clock
__device__ __forceinline__ bool wait_flag(const u64* p, u64 seq, unsigned long long timeout_ticks) {
  const unsigned long long t0 = wall_clock64();
  while (load_sys(p) < seq) {
    __builtin_amdgcn_s_sleep(2);
    if (wall_clock64() - t0 > timeout_ticks) return false;
  }
  return true;
}
// A column block's flag word: low half = the evaluation's sequence number, high half = a payload that rides on it (the
// owning shard's inlier count on its first block: it arrives WITH the flag instead of behind a second load).  Polls back to
// back; the wall clock (a scalar memory read of its own) is looked at every 64th poll only.
__device__ __forceinline__ u64 block_flag(u64 seq, u64 payload) { return (payload << 32) | (seq & 0xffffffffull); }
__device__ __forceinline__ bool wait_block_flag(const u64* p, u64 seq, unsigned long long timeout_ticks, u64* payload) {
  const uint32_t want = uint32_t(seq);
  u64 w = load_sys(p);
  if (uint32_t(w) != want) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 1;; spins++) {
      w = load_sys(p);
      if (uint32_t(w) == want) break;
      if ((spins & 63u) == 0) {
        if (wall_clock64() - t0 > timeout_ticks) return false;
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  *payload = w >> 32;
  return true;
}

// ------------------------------------------------------------------------------------------
// pass A (SPLINE): joint histogram with bicubic B-spline soft assignment.
// LDS: tile[GW*B << cshift] u64 (this workgroup's GW histogram columns, 2^cshift copies) + 1 u32 inlier counter.
//
// WIDE = the B = 256 / GW = 1 specialisation (the headline configuration): 512 threads share one
// histogram column with 32 lane-private copies, so a cell lives at LDS byte address
// (bin_image << 8) | (copy << 3) and ONE v_perm_b32 builds it from the packed bin-image word (the
// generic path needs a byte extract + shift and an add per tap); the tile then spans 64 KB, hence
// the 8-wave workgroup (2 workgroups = 16 waves per CU, the same wave occupancy as 4 x 4 waves).
// The tile must start at LDS address 0: this kernel has no static LDS, so the dynamic segment does.
// Measured on cfg 2: 74.3 -> 71.5 us.
#ifndef NID_WIDE_THREADS
#define NID_WIDE_THREADS 512
#endif
constexpr int kWideThreads = NID_WIDE_THREADS;
constexpr int kWideShift = 5;
// Chunks and segments.  A chunk is one workgroup's slice [start, start + count) of the bucketed records; `group` is the
// column group its first record belongs to.  Since round 4 the chunks of a table are EQUAL slices of the whole record array
// (nidreg_plan.hip split_groups), so a chunk may run across column-group boundaries: the workgroup then works through it segment
// by segment -- the records of one group each --, flushing (and re-zeroing) its histogram tile, or rebuilding its G tile,
// at every boundary.  gend[g] = index one past the last record of group g (the groups' end offsets; empty groups repeat the
// previous value and are skipped).  Everything about a segment is wave-uniform (scalar loads, scalar branches).
// The segment loop costs registers (the hot loops sit at the 128-VGPR edge of four waves per SIMD) and ~3 VALU instructions
// per point of re-materialised constants, so every point kernel exists twice: SEG = false is the straight-line kernel of
// rounds 1-3 for tables whose chunks all lie inside one group (any cloud with equally full columns: the headline), SEG =
// true the loop; the host picks per table (nidreg_plan.hip).  A chunk holds at most kMaxSegs segments (the gradient kernel
// stages the G columns of all of them in LDS up front).
constexpr int kMaxSegs = 4;
struct Segments {
  const uint32_t* __restrict__ gend;
  uint32_t pos, end, g;
  __device__ __forceinline__ Segments(const uint32_t* __restrict__ gend_, const Chunk& ch) : gend(gend_), pos(ch.start), end(ch.start + ch.count), g(ch.group) {}
  // records of the current segment: [pos, seg_end())
  __device__ __forceinline__ uint32_t seg_end() const { return min(end, gend[g]); }
  // moves to the next segment; false when the chunk is finished
  __device__ __forceinline__ bool advance(uint32_t seg_end_) {
    pos = seg_end_;
    if (pos >= end) return false;
    do { g++; } while (gend[g] <= pos);  // skip empty groups
    return true;
  }
};

// Between two segments: wait for this wave's outstanding global stores / atomics (the flush, the partial sums).  gfx9 counts
// loads and stores on the one vmcnt and completes them out of order with respect to each other, so with stores possibly in
// flight on the segment loop's back edge the compiler has to wait for vmcnt(0) at EVERY use of a prefetched record -- the
// four record loads of an iteration are then no longer consumed one by one (measured: the looped histogram kernel +20-30 %).
// An s_waitcnt it can see on that edge gives the point loop its vmcnt(3) / (2) / (1) back.
__device__ __forceinline__ void drain_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }  // vmcnt(0), expcnt / lgkmcnt untouched

// The pass as a device function.  `smem` = the workgroup's dynamic LDS (tile at its start); kT = threads per workgroup.
template <int MODEL, typename Rec, typename real, bool WIDE, bool SEG, int kT>
__device__ __forceinline__ void spline_hist_body(
  const Rec* __restrict__ pts, const Chunk ch, const uint32_t* __restrict__ gend, const uint8_t* __restrict__ img, int pitch, int W, int H, const PoseParams<real>& pose,
  const CamParams<real>& cam, int B, int GW, int cshift, double dn_scale, u64* __restrict__ hist, unsigned char* smem, bool prio) {
  u64* tile = reinterpret_cast<u64*>(smem);
  if (WIDE) {  // the specialisation's tiling is fixed: compile-time constants instead of three SGPRs
    B = 256;
    GW = 1;
    cshift = kWideShift;
  }
  const int tile_n = GW * B;
  // every histogram cell has 2^cshift lane-private copies, interleaved so that lane l only ever
  // touches copy (l mod 2^cshift): with 16 copies the 16 lanes the LDS services per cycle hit 16
  // different 8-byte bank pairs -> ds_add_u64 is conflict free by construction (measured before:
  // 80 % of LDS cycles were bank-conflict cycles)
  const int tile_w = tile_n << cshift;
  const uint32_t cmask = (1u << cshift) - 1u;
  u64* s_colsum = tile + tile_w;  // GW words: this workgroup's contribution to each column sum
  unsigned int* s_inl = reinterpret_cast<unsigned int*>(s_colsum + GW);

  const int tid = threadIdx.x;
  stamp_begin();
  stamp_stage(0);
  if (tid == 0) *s_inl = 0;

  const real fW = real(W), fH = real(H);
  const uint32_t lane_copy = uint32_t(tid) & cmask;
  unsigned int inl = 0;
  const BsplineScale KU = bspline_scale(dn_scale);  // uniform: the x-weight polynomial's constants in fixed-point units

  Segments seg(gend, ch);
  // (Requesting the chunk's first batch of records before the tile is zeroed -- and, in the gradient kernel, before the
  // entropy tail / G tile -- was measured in round 4: histogram kernel unchanged, gradient kernel +2 us (the compiler peels
  // the first iteration: 3963 instead of 2960 instructions); profiles/archive/r04d_variants.txt.  Not kept.)
  RawBatch<Rec, kUnroll> rb;
  for (;;) {
    // a zeroed tile for every segment (coalesced stores; zeroing inside the flush below -- 32 more LDS addresses per thread in the
    // WIDE kernel -- took the looped kernel from 93 to 161 VGPRs)
    for (int k = tid; k < tile_w; k += kT) tile[k] = 0;
    if (tid < GW) s_colsum[tid] = 0;
    __syncthreads();
    stamp_stage(1);
    const uint32_t seg_end = SEG ? seg.seg_end() : seg.end;
    const uint32_t cnt = seg_end - seg.pos;  // >= 1
    const uint32_t col0 = seg.g * uint32_t(GW);
    // record stream: uniform base (SGPR pair) + 32-bit byte offsets per lane
    const char* rec_base = reinterpret_cast<const char*>(pts + seg.pos);

    // One point: floor knot, cubic B-spline weights per axis (x-weights already in fixed-point units), the 4x4 tap
    // patch of the strip-tiled bin image as two 16-byte loads, 16 ds_add_u64 into the lane-private copy.
    // K is the uniform constant set when every lane of the wave holds an inlier (the common case after view
    // culling), a per-lane set zeroed for outliers / padding slots otherwise: the same arithmetic either way, so
    // the bits a point contributes do not depend on its wave neighbours (tiling independence).
    auto taps = [&](real uc, real vc, uint32_t bin, const BsplineScale& K) {
      const int kx = int(uc), ky = int(vc);  // uc, vc >= 0: truncation is the floor knot (nid_cost.hpp:52)
      double bxs[4];
      real by[4];
      // |.|: a projected coordinate of exactly -0.0 passes `>= 0`, and a -0.0 fraction would put a sign bit into a
      // weight (to_fixed_dn reads bit patterns); the modifier folds into the consuming instructions
      bspline_scaled(double(m_abs(m_fract(uc))), K, bxs);
      bspline6<real>(m_abs(m_fract(vc)), by);
      u64* col = tile + ((((bin - col0) * uint32_t(B)) << cshift) + lane_copy);
      // padded bin image: tap (a,b) of knot (kx,ky) is padded pixel (kx + a, ky + b) (edge-replicated,
      // which is the reference's clamp of knots_x / knots_y, nid_cost.hpp:70-73)
      uint32_t cols[4];  // the two strip loads are issued before the first LDS atomic
      load_patch(img, pitch, kx, ky, cols);
#pragma unroll
      for (int b = 0; b < 4; b++) {
#pragma unroll
        for (int a = 0; a < 4; a++) {
          if (WIDE) {
            typedef __attribute__((address_space(3))) u64 lds_u64_t;
            const uint32_t addr = __builtin_amdgcn_perm(cols[a], lane_copy << 3, 0x0c0c0000u | (uint32_t(4 + b) << 8));  // [0, 0, byte b of cols[a], copy * 8]
            __hip_atomic_fetch_add((lds_u64_t*)(uintptr_t)addr, to_fixed_dn(bxs[a], double(by[b])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          } else {
            const uint32_t r = (cols[a] >> (8 * b)) & 0xffu;
            atomicAdd(&col[r << cshift], to_fixed_dn(bxs[a], double(by[b])));  // v_mul_f64 + ds_add_u64
          }
        }
      }
    };

    // kUnroll records per thread are fetched before any of them is processed (memory-level parallelism); the
    // geometry of all of them comes first, then -- wave-uniform choice -- the tap code with uniform or per-lane
    // constants.  Both are branch-free per point, so the scheduler interleaves the kUnroll independent points.
    // GUARDED = the batch may reach past the end of the segment (clamped loads, per-slot validity): the last batch of a segment only
    auto batch = [&](uint32_t base, auto guarded) {
      constexpr bool GUARDED = decltype(guarded)::value;
      set_progress_priority(prio, seg.pos - ch.start + base, ch.count);
      real xs[kUnroll], ys[kUnroll], zs[kUnroll];
      uint32_t bins_[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; k++) rb.load(rec_base, (GUARDED ? min(base + uint32_t(k) * kT + tid, cnt - 1u) : base + uint32_t(k) * kT + tid) * uint32_t(sizeof(Rec)), k);
#pragma unroll
      for (int k = 0; k < kUnroll; k++) rb.template get<real>(k, xs[k], ys[k], zs[k], bins_[k]);
      real us[kUnroll], vs[kUnroll];
      bool ins[kUnroll];
      bool all_in = true;
      // Round 6: a slot of the batch that lies past the end of the segment for EVERY lane of this wave is skipped (a scalar compare on
      // the wave's first slot index).  The last batch of a chunk used to cost a full batch whatever it held -- all four slots run the
      // same instructions on clamped records and add exact zeros --: a fifth of the WIDE kernel's point loop at 1.25M points / 256 bins
      // (2441 records per workgroup = one full batch + 393 records: 9.5 -> 6 us), and slot 3 of seven waves out of eight in the last batch of
      // the 10M-point headline (9766 = 4 x 2048 + 1574).  The skipped slots contributed zeros: the histogram's bits are unchanged.
      bool live[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; k++) live[k] = !GUARDED || uint32_t(__builtin_amdgcn_readfirstlane(int(base + uint32_t(k) * kT + tid))) < cnt;
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
        if (!live[k]) {
          us[k] = vs[k] = real(0);
          ins[k] = false;
          all_in = false;  // (the batch takes the per-lane-constant tap path below, which skips this slot)
          continue;
        }
        const bool valid = !GUARDED || base + uint32_t(k) * kT + tid < cnt;
        real cx, cy, cz;
        transform_fma<real>(pose, xs[k], ys[k], zs[k], cx, cy, cz);
        project<MODEL, real, real, true>(cam, cx, cy, cz, us[k], vs[k]);
        // floor(u) in [0,W) and floor(v) in [0,H); NaN / inf / overflow compare false -> outlier
        // (the reference's int conversion sends those to INT_MIN, nid_cost.hpp:52-58)
        ins[k] = bool(int(valid) & int(us[k] >= real(0)) & int(us[k] < fW) & int(vs[k] >= real(0)) & int(vs[k] < fH));  // no short circuit: branch-free
#ifdef NID_EXP_HANDOFF
        if (valid) g_uv_handoff[seg.pos + base + uint32_t(k) * kT + tid] = make_double2(double(us[k]), double(vs[k]));
#endif
        inl += ins[k] ? 1u : 0u;
        all_in = bool(int(all_in) & int(ins[k]));
      }
      if (__builtin_amdgcn_ballot_w64(!all_in) == 0) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) taps(us[k], vs[k], bins_[k], KU);
      } else {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
          if (!live[k]) continue;
          // an outlier (or a slot past the end of the segment) runs the same instructions with its knot at pixel (0,0)
          // and zeroed constants: it adds exact zeros
          const bool in = ins[k];
          BsplineScale KL;
          KL.k16 = in ? KU.k16 : 0.0;
          KL.k46 = in ? KU.k46 : 0.0;
          KL.k05 = in ? KU.k05 : 0.0;
          KL.k1 = in ? KU.k1 : 0.0;
          taps(in ? us[k] : real(0), in ? vs[k] : real(0), bins_[k], KL);
        }
      }
    };
    // every full batch without the bounds checks (branch-free: the four points interleave), the last one with them and with the
    // per-wave skip of dead slots.  (Until round 5 one guarded form ran every batch -- the split alone measured the same at 10M points,
    // profiles/archive/r04g_variants.txt; putting the dead-slot branches into EVERY batch cost the fisheye kernel 6 %: its four atan2
    // chains no longer interleaved, 164 -> 173 instructions per point, profiles/r06_experiments.md.)
    {
      uint32_t base = 0;
      for (; base + uint32_t(kT * kUnroll) <= cnt; base += kT * kUnroll) batch(base, std::false_type());
      if (base < cnt) batch(base, std::true_type());
    }
    __syncthreads();
    stamp_stage(2);

    // flush the tile: contiguous in the [bin_points][bin_image] device layout
    u64* dst = hist + size_t(seg.g) * size_t(tile_n);
    for (int k = tid; k < tile_n; k += kT) {
      u64 vv = 0;
      for (uint32_t j = 0; j <= cmask; j++) vv += tile[(uint32_t(k) << cshift) + ((j + uint32_t(k)) & cmask)];  // rotated: conflict-free reads
      if (vv) {
        atomicAdd(&dst[k], vv);
        atomicAdd(&s_colsum[k / B], vv);
      }
    }
    __syncthreads();
    if (tid < GW && s_colsum[tid]) atomicAdd(&hist[size_t(B) * size_t(B) + kTailWords + col0 + uint32_t(tid)], s_colsum[tid]);
    if (!SEG || !seg.advance(seg_end)) break;
    drain_vmem();
  }

  // inlier count: wave-reduce, one LDS add per wave, one global add per workgroup
  const unsigned int winl = wave_sum(inl);
  if ((tid & 63) == 0 && winl) atomicAdd(s_inl, winl);
  __syncthreads();
  if (tid == 0 && *s_inl) atomicAdd(&hist[size_t(B) * size_t(B) + kTailInliers], u64(*s_inl));
  stamp_stage(3);
  stamp_end();
}

// LDS bytes spline_hist_body uses (nidreg_plan.hip sizes lds_hist the same way)
__host__ __device__ __forceinline__ size_t spline_hist_lds_bytes(int B, int GW, int cshift) { return (size_t(GW) * size_t(B) * 8 << cshift) + size_t(GW) * 8 + 16; }

// waves per SIMD the WIDE kernel is compiled for: two 8-wave workgroups per CU need <= 128 VGPRs; both the straight-line and
// the looped kernels land there by themselves (the `atan` model's Dual3 forward mode holds ~164 either way), so no bound is
// imposed (a forced bound made the compiler give up the four-deep record prefetch: +20-30 % on the looped kernel).
template <int MODEL, typename Rec, typename real, bool WIDE, bool MULTI, bool SEG>
__global__ __launch_bounds__(WIDE ? kWideThreads : kThreads) void k_spline_hist(
  const Rec* __restrict__ pts, const Chunk* __restrict__ chunks, const uint32_t* __restrict__ gend, const uint8_t* __restrict__ img, int pitch, int W, int H, PoseParams<real> pose, CamParams<real> cam, int B,
  int GW, int cshift, double dn_scale, u64* __restrict__ hist, int prio,
  const MultiEntry* __restrict__ multi, typename multi_dyn_of<MULTI>::type dyn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kT = WIDE ? kWideThreads : kThreads;
  const Chunk ch = chunks[blockIdx.x];
  if constexpr (MULTI) {  // one grid over several pairs: this chunk's pair brings its own records, bin image, histogram and unit
    const uint32_t pair = ch.pad & 0xffu;  // Chunk::pad = pair | (index of the chunk among its pair's chunks) << 8
    const MultiEntry& e = multi[pair];
    pts = as_global(static_cast<const Rec*>(e.pts));
    gend = as_global(e.gend);
    img = as_global(e.img);
    hist = as_global(e.hist_buf[dyn.cur[pair]]);
    dn_scale = e.k16;
  }
  spline_hist_body<MODEL, Rec, real, WIDE, SEG, kT>(pts, ch, gend, img, pitch, W, H, pose, cam, B, GW, cshift, dn_scale, hist, smem, prio != 0);
}

// ------------------------------------------------------------------------------------------
// pass A (NEAREST): CostCalculatorNID's hard assignment.  FoV gate on the normalised camera-frame
// point, truncating int cast, one count per inlier.  The arithmetic order matches the reference
// expression tree so that, compiled with -ffp-contract=off, +,-,*,/,sqrt results are bit-identical
// to the CPU's and the integer histogram is exactly reproducible.
// Round 4: the kernel decides in two tiers.  What the histogram needs from a point is three DECISIONS -- inside the FoV cone,
// inside the image, which pixel --, not the projected coordinates themselves.  The FAST tier (plumb_bob, double precision)
// computes zn, u, v the way the SPLINE kernels do (fused multiply-adds, v_rcp / v_rsq + one Newton step: ~75 instead of ~137
// VALU instructions per point) together with a bound on how far these values can be from the reference's (below); a point
// whose decisions all lie outside that band keeps them, and only lanes inside the band (a few per 10^7 points; `ballot`)
// repeat the point in the reference's exact expression order.  The integer histogram stays bit-exact against the CPU
// (tests: every model and bin count, 10M points, and points placed 1e-13 ... 1e-9 around every kind of decision boundary).
//
// The bound.  Both tiers evaluate the same real-valued functions of the same inputs; with eps = 2^-52:
//   c = M p + t:   |c_fast - c_exact| <= e1 = 8 eps (rmax (|x| + |y| + |z|) + tmax)   per component (three products, three
//                  sums, either association; rmax / tmax = largest |entry| of the rotation block / the translation);
//   zn = cz / |c|: |d zn| <= sqrt(3) e1 / |c| + 16 eps                                 -> band bz = 8 e1 / |c| + 4e-14
//                  (the one-Newton-step rsqrt may be 2^-46 = 1.4e-14 relative off by the documented worst case of its seed:
//                  <= 2.1e-14 on zn, and the constant carries a factor ~2 over that);
//   px = cx / cz:  |d px| <= e1 (1 + |px|) / |cz| + 1.5e-14 |px|  (one-Newton-step reciprocal: 2^-46 relative); same for py;
//   (u, v) = f D(px, py) + c0 with |px|, |py| <= pmax = tan(max_fov) for a point inside the cone, D's Jacobian row sums <= K
//                  there (from the distortion coefficients, on the host): |d u| <= A e1 / |cz| + Bc with
//                  A = 2 fmax K (1 + pmax) and Bc = fmax (4e-14 K pmax + 2e-14 R pmax) + 1e-15 (W + H + |cx| + |cy|),
//                  R = sup |radial factor|                                               -> band bu = A e1 / |cz| + Bc.
// Every factor carries a margin of >= 2 over the first-order term; second-order terms are below 2^-20 of them whenever the
// band is small enough to matter (a band >= 0.5 px sends the lane to the exact tier by itself).  NaN / inf anywhere compare
// false, i.e. "inside the band": those points take the exact tier and behave like the reference.
// Round 5: the three wide-angle models of the BASELINE configs have a fast tier as well -- the same structure (values by the SPLINE
// kernels' cores, a per-point band, exact tier by ballot inside it), each with the band of its own chain rule:
//   omnidir   m = c_xy / den, den = cz + xi |c|:  |d m| <= e1 (1 + |m| (1 + 1.74 xi)) / |den|, then the distortion as for plumb_bob
//             with |m| <= mmax = sin(fov) / (cos(fov) + xi) inside the cone:      bu = A e1 / |den| + Bc,  A = 2 fmax K (1 + mmax (1 + 1.74 xi));
//             Bc also carries the relative error of 1 / den (one rsqrt and one reciprocal with one Newton step each);
//   fisheye   u - cx = fx s x, s = theta_d(theta) / r, theta = atan2(r, |z|):  |d (s x)| <= e1 (1.5 D / |c| + 2 s), D = sup |theta_d'|
//             on [0, pi/2] (host):                                                  bu = e1 (A / |c| + C s) + Brel (|u - cx| + |v - cy| + 1);
//   equirect  (round 5: u, v by two full-precision atan2 with bands A e1 / rho + Bc and C e1 / |c| + Bc2 + D |c| / rho -- 84 us for 10M
//             points; round 6 decides on the pixel BOUNDARIES instead, without the angles: see below NearestFast)
struct NearestFast {
  double er, et;  // 8 eps rmax, 8 eps tmax
  double A, Bc;
  double C, D, Bc2;  // (wide-angle models, see above)
  int on;         // 0: exact tier only (atan / rational_polynomial, or a cone too wide for the model's bound)
  // equirectangular (round 6): pixel BOUNDARIES instead of pixel coordinates (below)
  const double* tab_c;  // [kmax + 1][2]: (cos, sin) of theta_k = 2 pi (k / W - 1/2), the longitude of column boundary u = k
  const double* tab_r;  // [jmax + 1]: t_j = s_j |s_j|, s_j = sin(pi (j / H - 1/2)): the signed squared sine of row boundary v = j
  int kmax, jmax;       // ceil(W), ceil(H) of the intrinsics
};

// Round 6, equirectangular: the decisions WITHOUT the angles.  u = W (1/2 + lon / 2 pi) with lon = atan2(x, z) and v = H (1/2 + lat' / pi)
// with lat' = asin(y / |c|) are monotone in their angle, so "which column" is "between which two boundary longitudes", and a point lies
// on the far side of the boundary theta_k exactly when  d_k = x cos(theta_k) - z sin(theta_k) = rho sin(lon - theta_k) >= 0;  "which row" is
// e_j = y |y| - t_j |c|^2 >= 0 (s -> s |s| is monotone: no square root); the cone  z / |c| >= cos(fov)  is  z |z| - cf |cf| |c|^2 >= 0.
// Products and sums only -- the round-5 tier spent 68 of its instructions on two full-precision atan2 and 16 on two reciprocal square
// roots, at 163 VGPRs (84 us for 10M points against 45 for plumb_bob).  A CANDIDATE column / row comes from a single-precision atan2
// (a degree-9 odd polynomial after the octant reduction: ~1e-5 rad, a few hundredths of a pixel); it is then confirmed against FOUR
// consecutive boundaries k - 1 ... k + 2, which fixes a candidate that is one off and needs nothing from the approximation but "within a
// pixel": d_{k-1} > 0 and d_{k+2} < 0 put lon inside three columns (sin changes sign once over less than pi), the signs of d_k, d_{k+1} say
// which.  A lane is DECIDED when every one of its tests is outside its band; the bands bound how far the value can be from zero while the
// REFERENCE's floating-point evaluation (cost_calculator_nid.cpp:30-47 through equirectangular.hpp:14-28) still decides otherwise:
//   d:  4 e1 + 7e-15 (|x| + |z|)     e1 = the camera-frame error of the fused transform (above); the reference's column flips within
//                                    W 4e-16 of an integer u (atan2 and the bearing's normalisation 1-1.5 ulp each, three more roundings),
//                                    i.e. 2.5e-15 rad, times rho <= |x| + |z|; 3 eps (|x| + |z|) for this side's table entries and products;
//   e:  8 e1 (|x|+|y|+|z|) + 60 eps |c|^2   the reference's asin(y / |c|) carries 1.5 ulp of its argument amplified by |c| / rho -- which the
//                                    factor 2 |b| cos(lat) of d e / d lat takes back: 13 eps |c|^2 in all; 5 eps for this side;
//   g:  8 e1 (|x|+|y|+|z|) + 16 eps |c|^2   (the cone; z / sqrt(n2) on the reference's side: 2 eps relative, squared).
// Undecided lanes (a few per 10^7 points), |c|^2 <= 2e-3 (the model's centre rule) and the vertical axis go to the exact tier.
__device__ __forceinline__ float approx_atan2f(float y, float x) {  // |error| ~ 1e-5 rad; NaN / garbage in -> a candidate the tests reject
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float t = mn * __builtin_amdgcn_rcpf(mx);
  const float t2 = t * t;
  float p = t * fmaf(t2, fmaf(t2, fmaf(t2, fmaf(t2, 0.0208351f, -0.0851330f), 0.1801410f), -0.3302995f), 0.9998660f);
  p = ay > ax ? 1.57079637f - p : p;
  p = x < 0.0f ? 3.14159274f - p : p;
  return copysignf(p, y);
}

// LDS of the NEAREST kernel: the tile counts POINTS, and a workgroup's chunk holds fewer than 2^32 of them -- 32-bit cells
// (ds_add_u32, half the LDS of the SPLINE tiles: 16 KB at 256 cells x 16 copies), summed to 64 bits in the flush
__host__ __device__ __forceinline__ size_t nearest_hist_lds_bytes(int B, int GW, int cshift) { return (((size_t(GW) * size_t(B) * 4 << cshift) + 7) & ~size_t(7)) + size_t(GW) * 8 + 16; }
// five waves per SIMD for the plumb_bob fast-tier instantiation (96 VGPRs, nothing spilled; 104-106 without the bound)
// (round 5: fisheye lands at 132 and equirectangular at 173 by themselves -- asked for four and three waves)
#ifndef NID_NEAREST_EQUIRECT_WAVES
#define NID_NEAREST_EQUIRECT_WAVES 4
#endif
constexpr int nearest_min_waves(int model, bool is_double, bool rec32, bool seg) {
  return (is_double && rec32 && !seg) ? (model == MODEL_PLUMB_BOB ? 5 : (model == MODEL_FISHEYE ? 4 : (model == MODEL_EQUIRECT ? NID_NEAREST_EQUIRECT_WAVES : 1))) : 1;
}
template <int MODEL, typename Rec, typename real, bool MULTI, bool SEG>
__global__ __launch_bounds__(kThreads, nearest_min_waves(MODEL, std::is_same<real, double>::value, sizeof(Rec) == sizeof(Rec32), SEG)) void k_nearest_hist(
  const Rec* __restrict__ pts, const Chunk* __restrict__ chunks, const uint32_t* __restrict__ gend, const uint8_t* __restrict__ img, int pitch, int W, int H, IsoParams<real> iso,
  CamParams<real> cam, int B, int GW, int cshift, real cos_fov, NearestFast fast, u64* __restrict__ hist,
  const MultiEntry* __restrict__ multi, typename multi_dyn_of<MULTI>::type dyn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* tile = reinterpret_cast<uint32_t*>(smem);
  const int tile_n = GW * B;
  const int tile_w = tile_n << cshift;
  const uint32_t cmask = (1u << cshift) - 1u;
  u64* s_colsum = reinterpret_cast<u64*>(smem + ((size_t(tile_w) * 4 + 7) & ~size_t(7)));  // GW words: this workgroup's contribution to each column sum
  unsigned int* s_inl = reinterpret_cast<unsigned int*>(s_colsum + GW);

  const int tid = threadIdx.x;
  const Chunk ch = chunks[blockIdx.x];
  if constexpr (MULTI) {  // one grid over several pairs (see k_spline_hist)
    const uint32_t pair = ch.pad & 0xffu;  // Chunk::pad = pair | (first gradient-partial slot of the chunk) << 8
    const MultiEntry& e = multi[pair];
    pts = as_global(static_cast<const Rec*>(e.pts));
    gend = as_global(e.gend);
    img = as_global(e.img);
    hist = as_global(e.hist_buf[dyn.cur[pair]]);
  }
  if (tid == 0) *s_inl = 0;

  const real fW = real(W), fH = real(H);
  unsigned int inl = 0;
  constexpr bool kFastTier = (MODEL == MODEL_PLUMB_BOB || MODEL == MODEL_FISHEYE || MODEL == MODEL_OMNIDIR || MODEL == MODEL_EQUIRECT) && std::is_same<real, double>::value;

  // the reference's expression order (cost_calculator_nid.cpp:30-47): +, -, *, /, sqrt bit-identical to the CPU's
  auto exact = [&](real x, real y, real z, bool& in, int& px, int& py) {
    // Eigen 4x4 * (x y z 1): ((m0 x + m1 y) + m2 z) + m3 * 1
    const real cx = ((iso.m[0] * x + iso.m[1] * y) + iso.m[2] * z) + iso.m[3];
    const real cy = ((iso.m[4] * x + iso.m[5] * y) + iso.m[6] * z) + iso.m[7];
    const real cz = ((iso.m[8] * x + iso.m[9] * y) + iso.m[10] * z) + iso.m[11];
    const real n2 = (cx * cx + cy * cy) + cz * cz;
    const real zn = n2 > real(0) ? cz / m_sqrt(n2) : cz;
    const bool in_fov = !(zn < cos_fov);  // out of FoV otherwise (cost_calculator_nid.cpp:32)
    real u, v;
    project<MODEL, real, real>(cam, cx, cy, cz, u, v);
    // trunc(u) in [0,W)  <=>  -1 < u < W ; NaN false (cost_calculator_nid.cpp:37-41)
    in = in_fov && (u > real(-1)) && (u < fW) && (v > real(-1)) && (v < fH);
    px = in ? int(u) : 0;  // truncation toward zero
    py = in ? int(v) : 0;
  };

  Segments seg(gend, ch);
  for (;;) {
    for (int k = tid; k < tile_w; k += kThreads) tile[k] = 0;  // a zeroed tile for every segment
    if (tid < GW) s_colsum[tid] = 0;
    __syncthreads();
    const uint32_t seg_end = SEG ? seg.seg_end() : seg.end;
    const uint32_t cnt = seg_end - seg.pos;
    const uint32_t col0 = seg.g * uint32_t(GW);
    const Rec* __restrict__ recs = pts + seg.pos;
    // kUnroll records per thread are fetched before any of them is processed, the decisions of all of them come before the
    // first pixel gather, and the gathers before the first LDS add: every memory latency of an iteration is paid once.
    // (Until round 3 the four points ran one after the other, each through its own gather: 40 % of the wave cycles waited
    // on memory, profiles/archive/r04a_pmc_summary_nearest.txt.)  No progress priority here (profiles/archive/r02g_kernel_gaps.txt).
    // (Round 6 tried the SPLINE kernels' split here -- full batches without the bounds checks, a guarded last one --: two copies of
    // the body under this kernel's register bounds spill: plumb_bob 60 -> 98 us per evaluation, fisheye 53 -> 57, omnidir 66 -> 72,
    // equirectangular 87 -> 85; profiles/r06_experiments.md.  One guarded loop.)
    for (uint32_t base = 0; base < cnt; base += kThreads * kUnroll) {
      real xs[kUnroll], ys[kUnroll], zs[kUnroll];
      uint32_t bins_[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
        const uint32_t ii = min(base + uint32_t(k) * kThreads + tid, cnt - 1u);
        load_rec<real>(recs + ii, xs[k], ys[k], zs[k], bins_[k]);
      }
      bool ins[kUnroll], need[kUnroll];
      int pxs[kUnroll], pys[kUnroll];
      bool any_need = false;
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
        ins[k] = false;
        pxs[k] = pys[k] = 0;
        need[k] = base + uint32_t(k) * kThreads + tid < cnt;  // (a valid slot)
      }
      bool fast_on = false;
      if constexpr (kFastTier) fast_on = fast.on != 0;  // uniform
      if (fast_on) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {  // branch-free: the four points interleave
          const bool valid = need[k];
          const double x = xs[k], y = ys[k], z = zs[k];
          const double cx = fma(double(iso.m[2]), z, fma(double(iso.m[1]), y, fma(double(iso.m[0]), x, double(iso.m[3]))));
          const double cy = fma(double(iso.m[6]), z, fma(double(iso.m[5]), y, fma(double(iso.m[4]), x, double(iso.m[7]))));
          const double cz = fma(double(iso.m[10]), z, fma(double(iso.m[9]), y, fma(double(iso.m[8]), x, double(iso.m[11]))));
          const double n2 = fma(cx, cx, fma(cy, cy, cz * cz));
          const double e1 = fma(fast.er, (fabs(x) + fabs(y)) + fabs(z), fast.et);
          if constexpr (MODEL == MODEL_EQUIRECT) {
            // the bands from ONE per-point magnitude, S = |x| + |y| + |z| of the LiDAR-frame point (e1 is linear in it already; the
            // camera-frame magnitudes are bounded through it: |c|_1 <= 3 (rmax S + tmax), with er = 8 eps rmax, et = 8 eps tmax):
            //   b1 >= 8 e1 |c|_1,  bd >= 4 e1 + 7e-15 (|cx| + |cz|)   -- looser by the factor |c|_1 bound / |c|_1 <= ~3, still 1e-13 of a pixel
            const double eps = 0x1p-52;
            const double S = (fabs(x) + fabs(y)) + fabs(z);
            const double m1 = fma(fast.er, S, fast.et) * (3.0 / (8.0 * eps));  // >= |c|_1
            const double b1 = (8.0 * e1) * m1;
            const double bd = fma(7e-15, m1, 4.0 * e1);
            // the cone
            const double cf = double(cos_fov);
            const double g = fma(-(cf * fabs(cf)), n2, cz * fabs(cz));
            const bool fov_safe = fabs(g) > fma(16.0 * eps, n2, b1);
            const bool in_fov = g >= 0.0;
            // candidates from single precision
            const float xf = float(cx), yf = float(cy), zf = float(cz);
            const float lonf = approx_atan2f(xf, zf);
            const float latf = approx_atan2f(yf, __builtin_amdgcn_sqrtf(fmaf(xf, xf, zf * zf)));
            const float uf = fmaf(float(cam.intr[0]) * 0.159154943f, lonf, float(cam.intr[0]) * 0.5f);
            const float vf = fmaf(float(cam.intr[1]) * 0.318309886f, latf, float(cam.intr[1]) * 0.5f);
            const int kc = min(max(int(uf), 1), fast.kmax - 2), jc = min(max(int(vf), 1), fast.jmax - 2);  // (NaN converts to 0)
            // columns: boundaries kc - 1 ... kc + 2 (uniform base + 32-bit byte offset per lane)
            const char* tcb = reinterpret_cast<const char*>(fast.tab_c) + uint32_t(kc - 1) * 16u;
            const double2 c0 = reinterpret_cast<const double2*>(tcb)[0], c1 = reinterpret_cast<const double2*>(tcb)[1], c2 = reinterpret_cast<const double2*>(tcb)[2],
                          c3 = reinterpret_cast<const double2*>(tcb)[3];
            const double d0 = fma(cx, c0.x, -(cz * c0.y)), d1 = fma(cx, c1.x, -(cz * c1.y)), d2 = fma(cx, c2.x, -(cz * c2.y)), d3 = fma(cx, c3.x, -(cz * c3.y));
            // decided when d0, -d3, |d1|, |d2| are all beyond the band (with d0 > 0 > d3 the signs of d1, d2 can only be + +, + - or - -:
            // sin(lon - theta) changes sign once over three columns; rho = 0 gives d = 0: undecided by itself)
            const bool col_ok = fmin(fmin(d0, -d3), fmin(fabs(d1), fabs(d2))) > bd;
            const int col = (kc - 1) + int(d1 > 0.0) + int(d2 > 0.0);
            // rows: boundaries jc - 1 ... jc + 2
            const char* trb = reinterpret_cast<const char*>(fast.tab_r) + uint32_t(jc - 1) * 8u;
            const double t0 = reinterpret_cast<const double*>(trb)[0], t1 = reinterpret_cast<const double*>(trb)[1], t2 = reinterpret_cast<const double*>(trb)[2],
                         t3 = reinterpret_cast<const double*>(trb)[3];
            const double yy = cy * fabs(cy);
            const double q0 = fma(-t0, n2, yy), q1 = fma(-t1, n2, yy), q2 = fma(-t2, n2, yy), q3 = fma(-t3, n2, yy);
            const bool row_ok = fmin(fmin(q0, -q3), fmin(fabs(q1), fabs(q2))) > fma(60.0 * eps, n2, b1);
            const int row = (jc - 1) + int(q1 > 0.0) + int(q2 > 0.0);
            const bool uv_safe = bool(int(n2 > 2e-3) & int(col_ok) & int(row_ok));
            const bool in_rng = bool(int(col < W) & int(row < H));  // (col, row >= 0 by construction: u, v >= 0 for this model)
            const bool decided = bool(int(fov_safe) & (int(!in_fov) | int(uv_safe)));
            need[k] = bool(int(valid) & int(!decided));
            ins[k] = bool(int(valid) & int(decided) & int(in_fov) & int(in_rng));
            pxs[k] = ins[k] ? col : 0;
            pys[k] = ins[k] ? row : 0;
          } else {
          const double rs = fast_rsq(n2);  // NaN for n2 == 0
          const double dz = fma(cz, rs, -double(cos_fov));
          const bool fov_safe = fabs(dz) > fma(8.0 * e1, rs, 4e-14);
          const bool in_fov = !(dz < 0.0);
          real u, v;
          double bu, bv;
          bool model_ok = true;  // (false: a rule of the model the bands do not cover -> exact tier)
          if constexpr (MODEL == MODEL_OMNIDIR) {
            const OmnidirCore<real> kc = omnidir_core<real>(cam, real(cx), real(cy), real(cz));
            project<MODEL, real, real, true>(cam, real(cx), real(cy), real(cz), u, v);  // (the same core: computed once after inlining)
            bu = bv = fma(fast.A * e1, fabs(double(kc.iden)), fast.Bc);
            model_ok = n2 > 0.0;
          } else if constexpr (MODEL == MODEL_FISHEYE) {
            const FisheyeCore<real> kc = fisheye_core<real>(cam, real(cx), real(cy), real(cz));
            u = fma(cam.intr[0], kc.s * real(cx), cam.intr[2]);
            v = fma(cam.intr[1], kc.s * real(cy), cam.intr[3]);
            bu = bv = fma(e1, fma(fast.A, rs, fast.C * fabs(double(kc.s))), fast.Bc * ((fabs(double(u) - double(cam.intr[2])) + fabs(double(v) - double(cam.intr[3]))) + 1.0));
          } else {
            project<MODEL, real, real, true>(cam, real(cx), real(cy), real(cz), u, v);
            bu = bv = fma(fast.A * e1, fabs(fast_rcp(cz)), fast.Bc);
          }
          const bool uv_safe = bool(int(model_ok) & int(fabs(double(u) - rint(double(u))) > bu) & int(fabs(double(v) - rint(double(v))) > bv));
          const bool in_rng = bool(int(u > real(-1)) & int(u < fW) & int(v > real(-1)) & int(v < fH));
          // decided: safely outside the cone (u, v do not matter), or safely inside it with every pixel decision safe
          const bool decided = bool(int(fov_safe) & (int(!in_fov) | int(uv_safe)));
          need[k] = bool(int(valid) & int(!decided));
          ins[k] = bool(int(valid) & int(decided) & int(in_fov) & int(in_rng));
          pxs[k] = ins[k] ? int(u) : 0;
          pys[k] = ins[k] ? int(v) : 0;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < kUnroll; k++) any_need = bool(int(any_need) | int(need[k]));
      if (__builtin_amdgcn_ballot_w64(any_need) != 0) {  // a lane of this wave sits inside a band (or the fast tier is off)
#pragma unroll
        for (int k = 0; k < kUnroll; k++)
          if (need[k]) exact(xs[k], ys[k], zs[k], ins[k], pxs[k], pys[k]);
      }
      uint32_t rs_[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; k++) rs_[k] = load_pixel(img, pitch, pxs[k] + 1, pys[k] + 1);  // (an outlier reads padded pixel (1, 1))
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
        if (ins[k]) {
          inl++;
          atomicAdd(&tile[((((bins_[k] - col0) * uint32_t(B)) + rs_[k]) << cshift) + (uint32_t(tid) & cmask)], 1u);
        }
      }
    }
    __syncthreads();

    u64* dst = hist + size_t(seg.g) * size_t(tile_n);
    for (int k = tid; k < tile_n; k += kThreads) {
      u64 vv = 0;
      for (uint32_t j = 0; j <= cmask; j++) vv += tile[(uint32_t(k) << cshift) + ((j + uint32_t(k)) & cmask)];
      if (vv) {
        atomicAdd(&dst[k], vv);
        atomicAdd(&s_colsum[k / B], vv);
      }
    }
    __syncthreads();
    if (tid < GW && s_colsum[tid]) atomicAdd(&hist[size_t(B) * size_t(B) + kTailWords + col0 + uint32_t(tid)], s_colsum[tid]);
    if (!SEG || !seg.advance(seg_end)) break;
    drain_vmem();
  }

  const unsigned int winl = wave_sum(inl);
  if ((tid & 63) == 0 && winl) atomicAdd(s_inl, winl);
  __syncthreads();
  if (tid == 0 && *s_inl) atomicAdd(&hist[size_t(B) * size_t(B) + kTailInliers], u64(*s_inl));
}

// nid_cost.hpp:86-104 from the three entropies (fixed point, see ent_fixed) and the inlier count: NID = (Hj - MI) / Hj,
// MI = Hi + Hp - Hj; the gradient pass's coefficients; status 1 when the NID is not finite (sum == 0 -> 0/0 in the
// reference, nid_cost.hpp:98-102: the functor returns false).  One definition for every path, so that they agree bit for bit.
__device__ __forceinline__ EntropyScalars entropy_scalars(long long hi_k, long long hp_k, long long hj_k, double S) {
  const bool empty = !(S > 0.0);  // the reference divides the histograms by sum = 0: every entropy is NaN
  const double qnan = __longlong_as_double(0x7ff8000000000000ll);
  const double Hi = empty ? qnan : -ent_value(hi_k), Hp = empty ? qnan : -ent_value(hp_k), Hj = empty ? qnan : -ent_value(hj_k);
  const double MI = Hi + Hp - Hj;
  const double nid = (Hj - MI) / Hj;
  EntropyScalars e;
  e.nid = nid;
  e.S = S;
  e.coefA = -(Hi + Hp) / (Hj * Hj * S);
  e.coefB = 1.0 / (Hj * S);
  e.Hi = Hi;
  e.Hp = Hp;
  e.Hj = Hj;
  e.status = isfinite(nid) ? 0.0 : 1.0;
  return e;
}

#ifdef NID_COMMON_KERNELS
// ------------------------------------------------------------------------------------------
// entropy: one workgroup per block of CB histogram columns, the last one to finish runs the tail (independent of the point
// kernels' tiling).  hist layout [c][r] (c = bin_points, r = bin_image); thread r walks the block's
// columns (coalesced).  Writes part_hj[j] = sum p log(p + 1e-6) over the block, row_part[j][r] =
// sum_c h[c][r] (fixed point); the column sums sum_r h[c][r] come from the histogram kernels' flush.
// entropy, part 2 (run by the last workgroup of k_entropy):   hist_image = row sums, hist_points = column sums / unit
// (partition of unity: the 16 weights of an inlier sum to 1), S = inlier count.
// nid_cost.hpp:86-104: NID = (Hj - MI) / Hj, MI = Hi + Hp - Hj.
__device__ __forceinline__ void entropy_final_body(
  double S, int B, int NG, double inv_unit, const long long* part_hj, const u64* row_part, const u64* col_sum, double* phi_q, double* hist_image_out, double* hist_points_out,
  EntropyScalars* scal, double* out, double* out_host, double tag, long long* s_red) {
  const int tid = threadIdx.x;
  long long hi_acc = 0, hp_acc = 0, hj_acc = 0;
  for (int r = tid; r < B; r += kThreads) {
    u64 t = 0;
    for (int g = 0; g < NG; g++) t += row_part[size_t(g) * size_t(B) + r];
    const double raw = double(t) * inv_unit;  // raw (un-normalised) hist_image[r]
    const double q = raw / S;
    hi_acc += ent_fixed(q * fast_log(q + 1e-6));
    phi_q[r] = fast_log(q + 1e-6) + q / (q + 1e-6);
    hist_image_out[r] = raw;
  }
  for (int c = tid; c < B; c += kThreads) {
    const double cnt = rint(double(col_sum[c]) * inv_unit);  // exact inlier count of column c
    const double p = cnt / S;
    hp_acc += ent_fixed(p * fast_log(p + 1e-6));
    hist_points_out[c] = cnt;
  }
  for (int g = tid; g < NG; g += kThreads) hj_acc += part_hj[g];
  hi_acc = wave_sum(hi_acc);
  hp_acc = wave_sum(hp_acc);
  hj_acc = wave_sum(hj_acc);
  if ((tid & 63) == 0) {
    s_red[(tid >> 6) * 3 + 0] = hi_acc;
    s_red[(tid >> 6) * 3 + 1] = hp_acc;
    s_red[(tid >> 6) * 3 + 2] = hj_acc;
  }
  __syncthreads();
  if (tid == 0) {
    long long a = 0, b = 0, c = 0;
    for (int w = 0; w < int(blockDim.x) / 64; w++) {
      a += s_red[w * 3 + 0];
      b += s_red[w * 3 + 1];
      c += s_red[w * 3 + 2];
    }
    const EntropyScalars e = entropy_scalars(a, b, c, S);
    *scal = e;
    out[0] = e.nid;
    out[8] = e.status;
    out[9] = S;
    if (out_host) {
      out_host[0] = e.nid;
      out_host[8] = e.status;
      out_host[9] = S;
      __threadfence_system();
      if (tag != 0.0) __hip_atomic_store(&out_host[15], tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // final kernel of a cost-only evaluation
    }
  }
}

#endif  // NID_COMMON_KERNELS
// NID_ENTROPY_COLS columns per workgroup of NID_ENTROPY_THREADS threads: thread (r = tid & 255, q = tid >> 8) takes row r of
// columns 4q .. 4q+3, i.e. 4 loads and 4 logs per thread (with one 256-thread workgroup per 16 columns the 16 dependent log
// chains of a wave ran at single-wave latency: 13.8 us); the quarter-row partials meet in LDS.
// Round 4: 8 columns x 512 threads (32 workgroups at B = 256) instead of 16 x 1024 (16 workgroups): 8.8 -> 7.2 us event-timed
// on cfg 2, 4 x 256 (64 workgroups) the same (profiles/archive/r04d_variants.txt).
#ifndef NID_ENTROPY_COLS
#define NID_ENTROPY_COLS 8
#endif
#ifndef NID_ENTROPY_THREADS
#define NID_ENTROPY_THREADS 512
#endif
constexpr int kEntropyColsMax = NID_ENTROPY_COLS;
constexpr int kEntropyThreads = NID_ENTROPY_THREADS;
static_assert(kEntropyThreads % 256 == 0 && kEntropyThreads <= 1024 && kEntropyColsMax % (kEntropyThreads / 256) == 0, "k_entropy: thread (r, q) takes row r of columns q kPer ... q kPer + kPer - 1");
constexpr int kEntropyWaves = kEntropyThreads / 64;
#ifdef NID_COMMON_KERNELS
template <bool MULTI>
__global__ __launch_bounds__(kEntropyThreads) void k_entropy(
  u64* __restrict__ hist, int B, int CB, double inv_unit, long long* part_hj, u64* row_part, double* phi_q, double* hist_image_out, double* hist_points_out,
  EntropyScalars* scal, double* out, double* out_host, double tag, unsigned int* counter, u64* __restrict__ zero_buf, long long zero_words, int tail,
  const MultiEntry* __restrict__ multi, typename multi_dyn_of<MULTI>::type dyn) {
  __shared__ long long s_red[3 * kEntropyWaves];
  __shared__ u64 s_row[3][256];
  __shared__ int s_flag;
  const int tid = threadIdx.x;
  int j = blockIdx.x;
  int nblocks = int(gridDim.x);
  if constexpr (MULTI) {  // dyn.neb workgroups per pair, pair after pair
    const int pair = int(blockIdx.x) / dyn.neb;
    j = int(blockIdx.x) % dyn.neb;
    nblocks = dyn.neb;
    const MultiEntry& e = multi[pair];
    hist = as_global(e.hist_buf[dyn.cur[pair]]);
    zero_buf = as_global(e.hist_buf[dyn.cur[pair] ^ 1]);
    zero_words = e.zero_words;
    inv_unit = e.inv_unit;
    part_hj = as_global(e.part_hj);
    row_part = as_global(e.row_part);
    phi_q = as_global(e.phi_q);
    hist_image_out = as_global(e.hist_image);
    hist_points_out = as_global(e.hist_points);
    scal = as_global(e.scal);
    out = as_global(e.out);
    out_host = as_global(e.out_host);
    counter = as_global(e.counters);
    tag = dyn.want_grad ? 0.0 : dyn.tag[pair];
    tail = (!dyn.want_grad || e.nchunks == 0) ? 1 : 0;  // a pair without points has no gradient workgroup to run the tail
  }
  // double-buffered histogram: clear the buffer the NEXT evaluation accumulates into (nidreg_core.hip,
  // begin_histogram) -- ~0.5 MB of stores that replace a memset launch per evaluation
  if (zero_buf)
    for (long long k = (long long)j * kEntropyThreads + tid; k < zero_words; k += (long long)nblocks * kEntropyThreads) zero_buf[k] = 0;
  const int r = tid & 255, q = tid >> 8;
  const int c0 = j * CB;
  const int ncols = min(CB, B - c0);
  constexpr int kPer = kEntropyColsMax / (kEntropyThreads / 256);  // columns per thread
  u64 v[kPer];
#pragma unroll
  for (int c = 0; c < kPer; c++) {
    const int col = q * kPer + c;
    v[c] = (r < B && col < ncols) ? hist[size_t(c0 + col) * size_t(B) + r] : 0;  // independent loads
  }
  const double S = double(hist[size_t(B) * size_t(B) + kTailInliers]);
  const double scale = inv_unit / S;  // fixed-point word -> probability
  long long acc = 0;
  u64 row = 0;
#pragma unroll
  for (int c = 0; c < kPer; c++) {
    if (v[c]) {
      const double p = double(v[c]) * scale;
      acc += ent_fixed(p * fast_log(p + 1e-6));
    }
    row += v[c];
  }
  if (q > 0) s_row[q - 1][r] = row;
  acc = wave_sum(acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  // this block's row sums and entropy partial are ADDED (device-scope atomics) to the B row-sum words and the Hj word behind
  // the histogram -- zero on entry like the histogram itself: they are part of the buffer the previous evaluation cleared.
  // Whoever runs the tail (this kernel's last workgroup, or every workgroup of k_spline_grad) reads B + 1 finished words
  // instead of summing NEB x B partials (the gradient prologue read 32 KB per workgroup, 32 MB over the grid, before).
  if (q == 0 && r < B) {
    u64 t = row;
#pragma unroll
    for (int i = 0; i < kEntropyThreads / 256 - 1; i++) t += s_row[i][r];
    if (t) atomicAdd(&hist[hist_row_sums_at(B) + size_t(r)], t);
  }
  if (tid < 64) {  // the sixteen wave partials, summed by one wave instead of a serial loop of LDS reads
    const long long t = wave_sum(tid < kEntropyWaves ? s_red[tid] : 0ll);
    if (tid == 0 && t) atomicAdd(&hist[size_t(B) * size_t(B) + kTailHj], u64(t));
  }
  (void)part_hj;
  (void)row_part;
  // cost + Jacobian: the tail (three entropies -> NID, coefficients, phi(q_r)) is run by every workgroup of k_spline_grad in
  // its prologue, in parallel on all CUs, from the partials stored above (grad_scalars_from_partials) -- the ticket, the
  // acquire and one workgroup's serial tail (~6.8 us of this kernel, profiles/archive/r02g_variants_prio.txt) leave the critical path
  if (!tail) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's atomics have been performed
  const u64* col_sum = hist + size_t(B) * size_t(B) + kTailWords;  // accumulated by the histogram kernels' flush
  // the tail's loops stride by kThreads = 256 over B <= 256 items: threads beyond 255 find nothing to do but take part
  // in its barriers and (with zeros) in its wave reductions -- s_red holds 3 slots for each of the 16 waves
  if (last_workgroup_arrives<true>(counter, unsigned(nblocks), &s_flag))
    entropy_final_body(S, B, 1, inv_unit, reinterpret_cast<const long long*>(hist + size_t(B) * size_t(B) + kTailHj), hist + hist_row_sums_at(B), col_sum, phi_q, hist_image_out,
                       hist_points_out, scal, out, out_host, tag, s_red);
}

#endif  // NID_COMMON_KERNELS

#ifdef NID_SHARD_KERNELS  // (defined by the one translation unit that launches them: nidreg_shard.hip)
// ------------------------------------------------------------------------------------------
// One pair spread over several GPUs by ONE host process (nidreg_internal.hpp ShardSet, nidreg_shard.hip): the points are partitioned by
// their pose-independent histogram column -- GPU g owns a contiguous range of column groups, balanced by their point
// counts -- so the shards' joint histograms have DISJOINT support: the "all-reduce of the 2-D histogram" is an all-gather of
// columns, and since no two shards write the same word it needs no reduction at all -- plain stores.
//
// Round 5: ONE exchange per evaluation (replicated histogram).  Rounds 3-4 kept the B x B table at home and exchanged
// (2 + B + B/n) words of partial sums instead -- but the inlier count S sits inside every logarithm (p = h / S), so that took
// TWO dependent exchanges (S first, then the partials) and two more kernels; their round trips were the whole protocol cost
// (32 -> 50-55 us per evaluation at 2-3 shards, profiles/archive/r04k_shard_phases.json).  Now every shard holds a replica of the whole
// integer histogram (fine-grained device memory, mapped into every peer):
//   k_*_hist          the shard's own points into its own columns of its own replica (as an unsharded handle does)
//   k_entropy_repl    one workgroup per block of CB columns, as k_entropy.  PUSH: the workgroups whose block this shard owns
//                     store the block (B x CB words) and its column sums into every peer's replica, then raise that block's
//                     flag at every peer; the flag of a shard's first block carries the shard's inlier count S_g.
//                     REDUCE: every workgroup waits for its block's flag (nothing to wait for when it owns the block) and for
//                     the first flag of every shard (S = sum S_g), then does exactly what k_entropy does on the local replica.
//                     B^2 (n - 1) / n words leave every GPU (458 KB at B = 256, n = 8: 64 KB per link), one flag per block.
//   k_spline_grad     unchanged: the entropy tail in its prologue, the shard's points against its own columns of G.
// Every shard computes the cost from the same integers -- bit-identical, which the host CHECKS after every evaluation.
// Shards on different devices run PUSH | REDUCE as one launch: the owners never wait before they push, so the waits of one
// device only depend on kernels that are already running on the others.  Shards that share a device (a 1-GPU box exercising
// the protocol) may share an in-order hardware queue: there the host launches PUSH for every shard, then REDUCE for every
// shard -- every wait then targets a kernel launched earlier.  Waits are bounded by the wall clock: a lost peer becomes an
// error code, not a hung GPU.

// NIDREG_SHARD_SELFTEST=1 (creation-time check of a set's exchange paths, before the first evaluation could hang on them):
// one ordered pair of shards at a time.  `ping` (on shard a's device) writes a pattern into b's gather block, releases, raises
// b's test flag [0][a] and waits for b's answer in its own test flag [1][b]; `pong` (on b's device) waits for the flag, checks
// that the pattern is visible behind it, answers.  out[0] = 1 ok / 2 flag wait timed out / 3 payload not visible behind the
// flag; out[1] (ping) = round trip in ticks of the 100 MHz wall clock.
__global__ void k_shard_selftest_ping(const ShardTable* tab, int peer, u64 seq, u64 pattern, u64* out, unsigned long long timeout_ticks) {
  if (threadIdx.x != 0) return;
  const int me = tab->me;
  const unsigned long long t0 = wall_clock64();
  store_sys(tab->gather[peer] + kGatherTest + me, pattern);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  store_sys(&tab->flags[peer][kFlagTest + 0 * kMaxShards + me], seq);
  const bool ok = wait_flag(&tab->flags[me][kFlagTest + 1 * kMaxShards + peer], seq, timeout_ticks);
  out[1] = wall_clock64() - t0;
  out[0] = ok ? 1 : 2;
}
__global__ void k_shard_selftest_pong(const ShardTable* tab, int peer, u64 seq, u64 pattern, u64* out, unsigned long long timeout_ticks) {
  if (threadIdx.x != 0) return;
  const int me = tab->me;
  const bool ok = wait_flag(&tab->flags[me][kFlagTest + 0 * kMaxShards + peer], seq, timeout_ticks);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  const u64 seen = load_sys(tab->gather[me] + kGatherTest + peer);
  store_sys(&tab->flags[peer][kFlagTest + 1 * kMaxShards + me], seq);
  out[0] = !ok ? 2 : (seen == pattern ? 1 : 3);
}

enum { SHARD_PUSH = 1, SHARD_REDUCE = 2 };
// k_entropy for a shard of a replicated histogram (thread layout, partial sums and tail are k_entropy's; see above).
// hist = this shard's replica of the current evaluation (= tab->hist[me][cur]); err_out / err_host: set to `seq` when a wait
// timed out.
__global__ __launch_bounds__(kEntropyThreads) void k_entropy_repl(
  u64* __restrict__ hist, int B, double inv_unit, const ShardTable* __restrict__ tab, u64 seq, int cur, int role, double* phi_q, double* hist_image_out, double* hist_points_out,
  EntropyScalars* scal, double* out, double* out_host, double tag, unsigned int* counter, u64* __restrict__ zero_buf, long long zero_words, int tail, double* err_out, double* err_host,
  unsigned long long timeout_ticks) {
  __shared__ long long s_red[3 * kEntropyWaves];
  __shared__ u64 s_row[3][256];
  __shared__ int s_flag;
  __shared__ int s_bad;
  const int tid = threadIdx.x;
  const int j = blockIdx.x, nblocks = int(gridDim.x);
  const int n = tab->n, me = tab->me, CB = tab->CB;
  const int c0 = j * CB;
  const int ncols = min(CB, B - c0);
  int owner = 0;
  while (owner + 1 < n && c0 >= tab->cut[owner + 1]) owner++;  // the last shard whose range starts at or before c0 (empty ranges are skipped)
  const bool mine = owner == me;
  const int r = tid & 255, q = tid >> 8;
  constexpr int kPer = kEntropyColsMax / (kEntropyThreads / 256);  // columns per thread
  const size_t tail_at = size_t(B) * size_t(B);
  u64 v[kPer];
#pragma unroll
  for (int c = 0; c < kPer; c++) v[c] = 0;
  if (mine) {
#pragma unroll
    for (int c = 0; c < kPer; c++) {
      const int col = q * kPer + c;
      if (r < B && col < ncols) v[c] = hist[size_t(c0 + col) * size_t(B) + r];  // accumulated by this shard's histogram kernel (device-scope atomics)
    }
  }
  if ((role & SHARD_PUSH) && mine) {
    for (int p = 0; p < n; p++) {
      if (p == me) continue;
      u64* dst = tab->hist[p][cur];
#pragma unroll
      for (int c = 0; c < kPer; c++) {
        const int col = q * kPer + c;
        if (r < B && col < ncols) store_sys(dst + size_t(c0 + col) * size_t(B) + r, v[c]);
      }
      if (tid < ncols) store_sys(dst + tail_at + kTailWords + c0 + tid, hist[tail_at + kTailWords + c0 + tid]);  // the block's column sums (the flush's)
    }
    // the shard's first block carries its inlier count, in the flag word itself -- to every shard, itself included
    const u64 Sg = (tid == 0 && c0 == tab->col_lo) ? hist[tail_at + kTailInliers] : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores have been performed at system scope
    __syncthreads();
    if (tid == 0) {
#ifdef NID_SHARD_LIGHT_FENCE  // (A/B: every payload word above is a write-through system-scope store that has been waited for)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
#else
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      const u64 word = block_flag(seq, Sg);
      for (int p = 0; p < n; p++) store_sys(&tab->flags[p][j], word);
    }
  }
  if (!(role & SHARD_REDUCE)) return;
  __shared__ u64 s_S[kMaxShards];
  if (tid == 0) s_bad = 0;
  __syncthreads();
  // wave 0: the first block of every shard that owns columns (its inlier count rides on the flag); wave 1: this block
  if (tid < n) {
    u64 Sg = 0;
    if (tab->cut[tid] < tab->cut[tid + 1] && !wait_block_flag(&tab->flags[me][tab->cut[tid] / CB], seq, timeout_ticks, &Sg)) s_bad = 1;
    s_S[tid] = Sg;
  } else if (tid == 64 && !mine) {
    u64 unused;
    if (!wait_block_flag(&tab->flags[me][j], seq, timeout_ticks, &unused)) s_bad = 1;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  if (s_bad && tid == 0) {
    *err_out = double(seq);
    if (err_host) *err_host = double(seq);
  }
  u64 Sn = 0;
  for (int g = 0; g < n; g++) Sn += s_S[g];
  const double S = double(Sn);
  // the whole pair's inlier count where k_spline_grad's prologue / the tail / the getters look for it (this shard's own count
  // was read by its first block before that block raised the flag waited for above)
  if (j == 0 && tid == 0) hist[tail_at + kTailInliers] = Sn;
  // double-buffered histogram: clear the replica the NEXT evaluation accumulates into (the peers write into it only after the
  // host has collected this evaluation from every shard)
  if (zero_buf)
    for (long long k = (long long)j * kEntropyThreads + tid; k < zero_words; k += (long long)nblocks * kEntropyThreads) zero_buf[k] = 0;
  if (!mine) {
#pragma unroll
    for (int c = 0; c < kPer; c++) {
      const int col = q * kPer + c;
      if (r < B && col < ncols) v[c] = load_sys(hist + size_t(c0 + col) * size_t(B) + r);  // stored by the block's owner
    }
  }
  const double scale = inv_unit / S;  // fixed-point word -> probability
  long long acc = 0;
  u64 row = 0;
#pragma unroll
  for (int c = 0; c < kPer; c++) {
    if (v[c]) {
      const double p = double(v[c]) * scale;
      acc += ent_fixed(p * fast_log(p + 1e-6));
    }
    row += v[c];
  }
  if (q > 0) s_row[q - 1][r] = row;
  acc = wave_sum(acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (q == 0 && r < B) {
    u64 t = row;
#pragma unroll
    for (int i = 0; i < kEntropyThreads / 256 - 1; i++) t += s_row[i][r];
    if (t) atomicAdd(&hist[hist_row_sums_at(B) + size_t(r)], t);
  }
  if (tid < 64) {
    const long long t = wave_sum(tid < kEntropyWaves ? s_red[tid] : 0ll);
    if (tid == 0 && t) atomicAdd(&hist[tail_at + kTailHj], u64(t));
  }
  if (!tail) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (last_workgroup_arrives<true>(counter, unsigned(nblocks), &s_flag)) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the other blocks' column sums were stored by their owners on other devices
    entropy_final_body(S, B, 1, inv_unit, reinterpret_cast<const long long*>(hist + tail_at + kTailHj), hist + hist_row_sums_at(B), hist + tail_at + kTailWords, phi_q, hist_image_out,
                       hist_points_out, scal, out, out_host, tag, s_red);
  }
}

#endif  // NID_SHARD_KERNELS

// ------------------------------------------------------------------------------------------
// pass B (SPLINE): gradient.  G[c][r] = coefA * phi(h[c][r] / S) + coefB * phi_q[r] with
// phi(p) = log(p + eps) + p / (p + eps) is dNID/d(raw weight in bin (r,c)) (DESIGN.md derivation).
// Per point: (gx, gy) = sum_taps G * d(w_tap)/d(u, v); gp = (gx gy) * d(uv)/d(p_cam) (Dual3
// projection); accumulate M += gp p^T (3x3) and gt += gp (3).  One 12-double partial per workgroup.
// LDS: gtile[GW*B] doubles + kWaves*12 doubles.
//
// GW1 = the single-column specialisation (GW = 1, i.e. B > 128): every point of the workgroup reads the
// same column of G, so the table needs no per-point column offset, and -- reads of one address
// broadcast, unlike atomics -- no lane-private copies either: the tap's LDS byte address is just
// bin_image << 3, ONE v_lshlrev_b32_sdwa instead of extract + shift + add (gtile sits at LDS
// address 0: no static LDS in this kernel).  Measured on cfg 2: 103.4 -> 99.5 us.
#ifndef NID_GRAD_MIN_WAVES
#define NID_GRAD_MIN_WAVES 1
#endif
// how a tap finds its G value in LDS
enum { TAP_COPIES = 0,   // gtile[(cell << cshift) + lane copy]: lane-private copies like the histogram tile (several columns per workgroup)
       TAP_SINGLE = 1 }; // one column, ONE copy at LDS address 0: byte address = bin_image << 3 (one SDWA shift)
                         // (a 32-copy conflict-free tile with one v_perm_b32 per tap was measured in round 3: the same loop time)

// The point loop of the pass over ONE segment of the workgroup's chunk (`cnt` records from `recs`, all of column group
// col0 / GW, whose G columns sit in gtile): accumulates M += gp p^T, gt += gp into acc[12].  `done` / `total`: the chunk's
// progress (issue priority).
template <int MODEL, typename Rec, typename real, int TAP, int kT, bool SPLIT>
__device__ __forceinline__ void spline_grad_loop(
  const Rec* __restrict__ recs, uint32_t cnt, uint32_t col0, uint32_t done, uint32_t total, const uint8_t* __restrict__ img, int pitch, int W, int H, const PoseParams<real>& pose,
  const CamParams<real>& cam, int B, int cshift, const double* gtile, double* acc, bool prio, uint32_t rec0 = 0) {
  const int tid = threadIdx.x;
  if (TAP == TAP_SINGLE) cshift = 0;
  const uint32_t cmask = (1u << cshift) - 1u;
  const real fW = real(W), fH = real(H);
  const uint32_t lane_copy = uint32_t(tid) & cmask;

  const char* rec_base = reinterpret_cast<const char*>(recs);  // uniform base + 32-bit byte offsets per lane
  auto batch = [&](uint32_t base, auto guarded) {  // GUARDED: see spline_hist_body
    constexpr bool GUARDED = decltype(guarded)::value;
    set_progress_priority(prio, done + base, total);
    real xs[kUnroll], ys[kUnroll], zs[kUnroll];
    uint32_t bins_[kUnroll];
    RawBatch<Rec, kUnroll> rb;
#pragma unroll
    for (int k = 0; k < kUnroll; k++) rb.load(rec_base, (GUARDED ? min(base + uint32_t(k) * kT + tid, cnt - 1u) : base + uint32_t(k) * kT + tid) * uint32_t(sizeof(Rec)), k);
#ifdef NID_EXP_HANDOFF
    double2 huv[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; k++) huv[k] = g_uv_handoff[rec0 + (GUARDED ? min(base + uint32_t(k) * kT + tid, cnt - 1u) : base + uint32_t(k) * kT + tid)];
#endif
#pragma unroll
    for (int k = 0; k < kUnroll; k++) rb.template get<real>(k, xs[k], ys[k], zs[k], bins_[k]);
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      // (a branch-free body like k_spline_hist's was measured 15 % slower here: 173 VGPRs, 2 waves/SIMD)
      if (GUARDED && base + uint32_t(k) * kT + tid >= cnt) break;
      const real x = xs[k], y = ys[k], z = zs[k];
      real cx, cy, cz;
      transform_fma<real>(pose, x, y, z, cx, cy, cz);
      real uu, vv;
      ProjCtx<real> ctx;
      project_fwd<MODEL, real>(cam, cx, cy, cz, uu, vv, ctx);
#ifdef NID_EXP_HANDOFF
      uu = real(huv[k].x), vv = real(huv[k].y);  // the value half of project_fwd above is dead code now
#endif
      const bool in = (uu >= real(0)) && (uu < fW) && (vv >= real(0)) && (vv < fH);
      if (in) {
        const int kx = int(uu), ky = int(vv);  // uu, vv >= 0 here: truncation is the floor knot
        const real sx = m_abs(m_fract(uu)), sy = m_abs(m_fract(vv));  // |.| as in the histogram pass: identical fractions
        real bx[4], by[4], dbx[4], dby[4];
        bspline6<real>(sx, bx);
        bspline6<real>(sy, by);
        bspline_deriv2<real>(sx, dbx);
        bspline_deriv2<real>(sy, dby);
        typedef __attribute__((address_space(3))) const double lds_f64_t;
        const double* gcol = gtile + ((((bins_[k] - col0) * uint32_t(B)) << cshift) + lane_copy);
        uint32_t cols[4];
#ifdef NID_EXP_GRAD_NOGATHER
        cols[0] = uint32_t(kx) * 0x01010101u, cols[1] = cols[0] + 0x01010101u, cols[2] = uint32_t(ky) * 0x01010101u, cols[3] = cols[2] + 0x01010101u;
#else
        load_patch(img, pitch, kx, ky, cols);
#endif
#ifdef NID_EXP_GRAD_PAD
        {  // experiment: NID_EXP_GRAD_PAD independent fp64 fmas per point (marginal cost of a VALU instruction)
          double pa = double(sx), pb = double(sy), pc = double(uu), pd = double(vv);
#pragma unroll
          for (int q = 0; q < NID_EXP_GRAD_PAD / 4; q++) {
            pa = fma(pa, 0.999, 0.25), pb = fma(pb, 0.999, 0.25), pc = fma(pc, 0.999, 0.25), pd = fma(pd, 0.999, 0.25);
          }
          acc[11] += (pa + pb + pc + pd) * 1e-300;
        }
#endif
        real gx = real(0), gy = real(0);
#pragma unroll
        for (int b = 0; b < 4; b++) {
          real sa = real(0), sb = real(0);
#pragma unroll
          for (int a = 0; a < 4; a++) {
            const uint32_t r = (cols[a] >> (8 * b)) & 0xffu;
            const real g = TAP == TAP_SINGLE ? real(*(lds_f64_t*)(uintptr_t)(r << 3)) : real(gcol[r << cshift]);
            sa = fma(g, dbx[a], sa);
            sb = fma(g, bx[a], sb);
          }
          gx = fma(sa, by[b], gx);
          gy = fma(sb, dby[b], gy);
        }
        real gpr[3];
        project_bwd<MODEL, real>(cam, ctx, gx, gy, gpr);
        const double gp0 = double(gpr[0]), gp1 = double(gpr[1]), gp2 = double(gpr[2]);
        const double dx = double(x), dy = double(y), dz = double(z);
        acc[0] = fma(gp0, dx, acc[0]);
        acc[1] = fma(gp0, dy, acc[1]);
        acc[2] = fma(gp0, dz, acc[2]);
        acc[3] = fma(gp1, dx, acc[3]);
        acc[4] = fma(gp1, dy, acc[4]);
        acc[5] = fma(gp1, dz, acc[5]);
        acc[6] = fma(gp2, dx, acc[6]);
        acc[7] = fma(gp2, dy, acc[7]);
        acc[8] = fma(gp2, dz, acc[8]);
        acc[9] += gp0;
        acc[10] += gp1;
        acc[11] += gp2;
      }
    }
  };
  // every full batch without the bounds checks, the last one with them: 18 of ~920 instructions per batch less, 75.3-75.8 ->
  // 74.8-75.0 us on cfg 2 (profiles/archive/r04g_variants.txt; -DNID_GRAD_ALWAYS_GUARDED: the single loop)
  // (SPLIT = false: the looped kernel instantiations, which have no registers to spare for a second copy of the body)
#ifdef NID_GRAD_ALWAYS_GUARDED
  constexpr bool kSplit = false;
#else
  constexpr bool kSplit = SPLIT;
#endif
  uint32_t base = 0;
  if constexpr (kSplit)
    for (; base + uint32_t(kT * kUnroll) <= cnt; base += kT * kUnroll) batch(base, std::false_type());
  for (; base < cnt; base += kT * kUnroll) batch(base, std::true_type());
}

// the workgroup's 12 sums -> partials[k][my_block] ([12][nchunks]: coalesced for the final reduction), stored write-through
// at agent scope.
// Round 5: through LDS, transposed.  Until round 4 every wave summed each of its 12 accumulators over its 64 lanes by DPP (twelve
// wave_sums = 314 VALU instructions per thread) and 12 threads added the four waves' results -- the largest single item of the
// per-workgroup overhead (8 instructions per point at 10M points, 30 % of a small evaluation's workgroup).  Now every thread
// stores its 12 accumulators ([12][kT], conflict-free), 12 x 16 threads each add 16 of them (stride 16: conflict-free reads)
// and finish inside their DPP row of 16 lanes (row_shr 1 / 2 / 4 / 8): ~60 instructions per thread, a fixed association like
// before.  `scratch`: 12 * kT doubles of LDS that may alias the G tile (dead once every wave has left the point loop: hence the
// barrier up front).  -DNID_GRAD_REDUCE_DPP restores the round-4 reduction (A/B).
constexpr int kGradScratchDoubles = 12 * kThreads;
template <int kT>
__device__ __forceinline__ void grad_reduce_store(const double* acc, double* s_red, double* scratch, double* partials, unsigned int my_block, unsigned int my_blocks) {
  const int tid = threadIdx.x;
#ifdef NID_GRAD_REDUCE_DPP
#pragma unroll
  for (int k = 0; k < 12; k++) {
    const double t = wave_sum(acc[k]);
    if ((tid & 63) == 0) s_red[(tid >> 6) * 12 + k] = t;
  }
  __syncthreads();
  if (tid < 12) {
    double t = 0.0;
    for (int w = 0; w < kT / 64; w++) t += s_red[w * 12 + tid];
    __hip_atomic_store(&partials[size_t(tid) * my_blocks + my_block], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#else
  static_assert(kT == 256, "12 accumulators x 16 lanes = 192 threads of the workgroup's 256");
  __syncthreads();  // every wave has left the point loop: the G tile under `scratch` is dead
#pragma unroll
  for (int k = 0; k < 12; k++) scratch[k * kT + tid] = acc[k];
  __syncthreads();
  if (tid < 12 * 16) {
    const int k = tid >> 4, j = tid & 15;
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < kT / 16; i++) t += scratch[k * kT + i * 16 + j];
    t += dpp_move<0x111, 0xf, true>(t);  // row_shr:1, 2, 4, 8: lane 15 of the row of 16 holds the row's sum
    t += dpp_move<0x112, 0xf, true>(t);
    t += dpp_move<0x114, 0xf, true>(t);
    t += dpp_move<0x118, 0xf, true>(t);
    if (j == 15) __hip_atomic_store(&partials[size_t(k) * my_blocks + my_block], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  (void)s_red;
#endif
}

// what the gradient kernel needs to run the entropy tail itself (k_entropy launched with tail = 0)
struct GradTail {
  double* phi_q;             // outputs, written by the pair's first workgroup (nidreg_get_hist, the cost's way to the host)
  double* hist_image;
  double* hist_points;
  EntropyScalars* scal;
  int from_partials;         // 1: run the tail on the row sums / Hj k_entropy left behind the histogram; 0: read scal / phi_q
                             // as k_entropy's own tail wrote them; 2: no k_entropy ran at all -- the
                             // workgroup sums the B x B cells itself (small tables only, grad_entropy_partials)
  u64* zero_buf;             // from_partials == 2: the histogram buffer of the NEXT evaluation, cleared here (k_entropy's other duty)
  long long zero_words;
};
// Tables of at most this many cells (B <= 32; the reference's default is 16 bins) need no entropy kernel in a cost+Jacobian
// evaluation: every gradient workgroup reads the whole table -- 1 to 4 cells per thread, about what its G tile costs it
// anyway -- and gets the same integers k_entropy would have left behind the histogram (sums of ent_fixed terms and of
// fixed-point cells: the order does not matter).  One launch and one kernel boundary less per evaluation.
constexpr int kSelfEntropyCells = 1024;

// A histogram word as the prologue of a gradient workgroup reads it.  COH = false: a plain load -- the words were written by an
// EARLIER kernel, the kernel boundary made them visible.  COH = true (nid_fused.hpp: the same kernel's other workgroups wrote them
// with device-scope atomics, a grid barrier lies in between): an agent-scope load, served at the coherence point -- no cache of this
// CU / XCD is consulted, and none needs to be invalidated (an acquire fence would drop the XCD's whole L2, records and image included,
// once per workgroup).
template <bool COH>
__device__ __forceinline__ u64 ld_hist(const u64* p) {
  if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

// row sums (into s_rows[B], LDS) and the fixed-point sum of p log(p + eps) over ALL cells of a small table, by every thread of
// the workgroup; s_redk: kT / 64 words.  The caller's next barrier publishes both.
template <int kT, bool COH = false>
__device__ __forceinline__ void grad_entropy_partials(const u64* __restrict__ hist, int B, double inv_unit, unsigned long long* s_rows, long long* s_redk) {
  const int tid = threadIdx.x;
  for (int r = tid; r < B; r += kT) s_rows[r] = 0;
  __syncthreads();
  const double S = double(ld_hist<COH>(&hist[size_t(B) * size_t(B) + kTailInliers]));
  const double scale = inv_unit / S;  // the same two operations as k_entropy: identical terms
  long long acc = 0;
  for (int k = tid; k < B * B; k += kT) {
    const u64 v = ld_hist<COH>(&hist[k]);
    if (v) {
      const double p = double(v) * scale;
      acc += ent_fixed(p * fast_log(p + 1e-6));
      atomicAdd(&s_rows[k % B], (unsigned long long)v);  // device layout [bin_points][bin_image]: k % B = the image bin
    }
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) s_redk[tid >> 6] = acc;
}

// The entropy tail in the gradient kernel's prologue: every workgroup computes the three entropies from the sums k_entropy
// left behind the histogram (integers: identical bits everywhere), NID, coefA / coefB, and phi(q_r) into s_phi.  `writer` (one
// workgroup per pair) publishes hist_image / hist_points / phi_q / scal and -- at agent scope, read by grad_final_body in
// another workgroup -- cost, status and inlier count.  s_redk: 3 * (kT / 64) words of LDS.
template <int kT, bool SELF, bool COH = false>
__device__ __forceinline__ EntropyScalars grad_scalars_from_partials(const u64* hist, int B, double inv_unit, const GradTail& gt, double* s_phi, long long* s_redk, bool writer, double* out) {
  static_assert(!COH || SELF, "coherent loads are wired for the small-table path only (column sums below; the row sums come from LDS there)");
  const int tid = threadIdx.x;
  const double S = double(ld_hist<COH>(&hist[size_t(B) * size_t(B) + kTailInliers]));
  const u64* col_sum = hist + size_t(B) * size_t(B) + kTailWords;
  const u64* row_sum = hist + hist_row_sums_at(B);
  long long hi_k = 0, hp_k = 0;
  long long hj_all = 0;
  if constexpr (SELF) {  // no k_entropy ran: row sums in LDS (behind phi: B <= 32), wave partials of Hj in s_redk[2 kT/64 ...]
    unsigned long long* s_rows = reinterpret_cast<unsigned long long*>(s_phi + 128);
    grad_entropy_partials<kT, COH>(hist, B, inv_unit, s_rows, s_redk + 2 * (kT / 64));
    __syncthreads();
    row_sum = reinterpret_cast<const u64*>(s_rows);
    for (int w = 0; w < kT / 64; w++) hj_all += s_redk[2 * (kT / 64) + w];
  } else {
    hj_all = (long long)hist[size_t(B) * size_t(B) + kTailHj];
  }
  for (int r = tid; r < B; r += kT) {
    const double raw = double(row_sum[r]) * inv_unit;  // raw (un-normalised) hist_image[r]
    const double qv = raw / S;
    const double lq = fast_log(qv + 1e-6);
    hi_k += ent_fixed(qv * lq);
    const double ph = lq + qv / (qv + 1e-6);
    s_phi[r] = ph;
    const double cnt = rint(double(ld_hist<COH>(&col_sum[r])) * inv_unit);  // exact inlier count of column r
    const double p = cnt / S;
    hp_k += ent_fixed(p * fast_log(p + 1e-6));
    if (writer) {
      gt.phi_q[r] = ph;
      gt.hist_image[r] = raw;
      gt.hist_points[r] = cnt;
    }
  }
  hi_k = wave_sum(hi_k);
  hp_k = wave_sum(hp_k);
  if ((tid & 63) == 0) {
    s_redk[(tid >> 6) * 2 + 0] = hi_k;
    s_redk[(tid >> 6) * 2 + 1] = hp_k;
  }
  __syncthreads();
  long long A = 0, Bk = 0;
  for (int w = 0; w < kT / 64; w++) {
    A += s_redk[w * 2 + 0];
    Bk += s_redk[w * 2 + 1];
  }
  const EntropyScalars e = entropy_scalars(A, Bk, hj_all, S);
  if (writer && tid == 0) {
    *gt.scal = e;
    __hip_atomic_store(&out[0], e.nid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&out[8], e.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&out[9], S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return e;
}

// The G columns of column group g into the workgroup's LDS tile: G[c][r] = (coefA phi(h[c][r] / S) + coefB phi(q_r)) / 12 --
// the 1/12 because the tap loop works with 6 b and 2 db/ds (bspline6 / bspline_deriv2) --, every cell in 2^cshift copies.
// (dst: the tile, or a staging area of the same layout)
template <bool COH = false>
__device__ __forceinline__ void build_gtile(const u64* __restrict__ hist, uint32_t g, int B, int GW, int cshift, double scale, double coefA, double coefB, const double* phi_q, double* gtile) {
  const int tid = threadIdx.x;
  const uint32_t cmask = (1u << cshift) - 1u;
  const int tile_n = GW * B;
  const u64* src = hist + size_t(g) * size_t(tile_n);
  const int ncols = min(GW, B - int(g) * GW);
  const int n = ncols * B;
  for (int k = tid; k < n; k += kThreads) {
    const double p = double(ld_hist<COH>(&src[k])) * scale;
    const double gval = (coefA * (fast_log(p + 1e-6) + p / (p + 1e-6)) + coefB * phi_q[k % B]) * (1.0 / 12.0);
    for (uint32_t j = 0; j <= cmask; j++) gtile[(uint32_t(k) << cshift) + ((j + uint32_t(k)) & cmask)] = gval;
  }
}
// (until round 5 the `atan` model ran the generic Dual3 forward mode at 164-170 VGPRs and was asked for three waves; with its own
// vector-Jacobian product -- nid_device.hpp atan_core -- it lands where the other models do)
// (the looped instantiations on float records are told to stay at four -- they land at 126-130 by themselves)
constexpr int grad_min_waves(int model, bool seg = false, bool rec32 = false) { return (seg && rec32) ? 4 : NID_GRAD_MIN_WAVES; }
// LDS of the gradient kernel: G tile (one copy of one column when GW = 1, else 2^cshift copies of GW columns), reduction scratch,
// phi(q_r), flag, the workgroup's own copy of cost / status / inlier count (s_fin); SEG (GW = 1 only): kMaxSegs - 1 staged G columns behind them
// the region at the start of the gradient kernel's LDS: the G tile during the point loop, the reduction's scratch after it
__host__ __device__ __forceinline__ size_t grad_tile_region_bytes(int B, int GW, int cshift) {
  const size_t tile = GW == 1 ? size_t(B) * 8 : (size_t(GW) * size_t(B) * 8 << cshift);
#ifdef NID_GRAD_REDUCE_DPP
  return tile;
#else
  return tile > size_t(12 * 256 * 8) ? tile : size_t(12 * 256 * 8);
#endif
}
__host__ __device__ __forceinline__ size_t spline_grad_lds_bytes(int B, int GW, int cshift, bool seg) {
  return grad_tile_region_bytes(B, GW, cshift) + size_t(kWaves) * 12 * 8 + 256 * 8 + 16 + 32 + (seg && GW == 1 ? size_t(kMaxSegs - 1) * size_t(B) * 8 : 0);
}
template <int MODEL, typename Rec, typename real, bool GW1, bool MULTI, bool SEG>
__global__ __launch_bounds__(kThreads, grad_min_waves(MODEL, SEG, sizeof(Rec) == sizeof(Rec32))) void k_spline_grad(
  const Rec* __restrict__ pts, const Chunk* __restrict__ chunks, const uint32_t* __restrict__ gend, const uint8_t* __restrict__ img, int pitch, int W, int H, PoseParams<real> pose,
  CamParams<real> cam, int B, int GW, int cshift, double inv_unit, const u64* __restrict__ hist, const double* __restrict__ phi_q, const EntropyScalars* __restrict__ scal, GradTail gt,
  double* partials, unsigned int nslots, double qx, double qy, double qz, double qw, double* out, double* out_host, double tag, unsigned int* counter, int prio,
  const MultiEntry* __restrict__ multi, typename multi_dyn_of<MULTI>::type dyn) {
  static_assert(!SEG || GW1, "a chunk runs across column groups only in the single-column kernels (the host builds one-segment tables otherwise)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* gtile = reinterpret_cast<double*>(smem);
  if (GW1) cshift = 0;
  double* s_red = reinterpret_cast<double*>(smem + grad_tile_region_bytes(B, GW, cshift));  // (cshift = 0 for GW1 above)
  double* s_phi = s_red + kWaves * 12;  // [256] phi(q_r) when this kernel runs the entropy tail itself
  int* s_flag = reinterpret_cast<int*>(s_phi + 256);
  // cost, status, inlier count as THIS workgroup's prologue computed them (the same integers everywhere: identical bits): the
  // workgroup that ends up finalising writes them to the host from here instead of re-reading what the pair's first workgroup
  // stored (three dependent agent-scope loads, ~1 us of the final stage; round 5)
  double* s_fin = reinterpret_cast<double*>(s_flag + 4);
  double* s_stage = s_fin + 4;  // SEG: [kMaxSegs - 1][B] G columns of the chunk's later segments

  const int tid = threadIdx.x;
  stamp_stage(0);
  const Chunk ch = chunks[blockIdx.x];
  unsigned int my_block = blockIdx.x, my_blocks = gridDim.x;
  if constexpr (MULTI) {
    const uint32_t pair = ch.pad & 0xffu;  // Chunk::pad = pair | (first gradient-partial slot of the chunk among its pair's) << 8
    const MultiEntry& e = multi[pair];
    pts = as_global(static_cast<const Rec*>(e.pts));
    gend = as_global(e.gend);
    img = as_global(e.img);
    hist = as_global(e.hist_buf[dyn.cur[pair]]);
    inv_unit = e.inv_unit;
    phi_q = as_global(e.phi_q);
    scal = as_global(e.scal);
    partials = as_global(e.partials);
    out = as_global(e.out);
    out_host = as_global(e.out_host);
    tag = dyn.tag[pair];
    counter = as_global(e.counters) + 1;
    my_block = ch.pad >> 8;  // (only its being 0 matters: the pair's first chunk publishes the marginals)
    my_blocks = unsigned(e.nchunks);
    nslots = unsigned(e.nslots);
    gt.phi_q = as_global(e.phi_q);
    gt.hist_image = as_global(e.hist_image);
    gt.hist_points = as_global(e.hist_points);
    gt.scal = as_global(e.scal);
  }
  {
    double coefA = 0.0, coefB = 0.0, S = 1.0;
    bool self = false;
    if constexpr (!GW1) self = gt.from_partials == 2;  // (small tables: always several columns per workgroup)
    if (self) {
      if constexpr (!GW1) {
        u64* zero_buf = gt.zero_buf;
        long long zero_words = gt.zero_words;
        if constexpr (MULTI) {
          const MultiEntry& e = multi[ch.pad & 0xffu];
          zero_buf = as_global(e.hist_buf[dyn.cur[ch.pad & 0xffu] ^ 1]);
          zero_words = e.zero_words;
        }
        if (zero_buf)
          for (long long k = (long long)my_block * kThreads + tid; k < zero_words; k += (long long)my_blocks * kThreads) zero_buf[k] = 0;
        // (Requesting everything the three prologue steps read -- the cells, S, the column sums, the G tile's cells -- at
        // entry, in one round trip, changed nothing: stage stamps 2.8 + 0.9 us before, 3.4 + 0.2 us after.  The prologue is a
        // chain of dependent fp64 logarithms / divisions at one wave per SIMD, not of loads; profiles/archive/r04n_stage_times*.json.)
        const EntropyScalars es = grad_scalars_from_partials<kThreads, true>(hist, B, inv_unit, gt, s_phi, reinterpret_cast<long long*>(s_red), my_block == 0, out);
        coefA = es.coefA, coefB = es.coefB, S = es.S;
        if (tid == 0) s_fin[0] = es.nid, s_fin[1] = es.status, s_fin[2] = es.S;
        phi_q = s_phi;
      }
    } else if (gt.from_partials) {
      const EntropyScalars es = grad_scalars_from_partials<kThreads, false>(hist, B, inv_unit, gt, s_phi, reinterpret_cast<long long*>(s_red), my_block == 0, out);
      coefA = es.coefA, coefB = es.coefB, S = es.S;
      if (tid == 0) s_fin[0] = es.nid, s_fin[1] = es.status, s_fin[2] = es.S;
      phi_q = s_phi;  // LDS through a generic pointer: B reads per tile
    } else {
      coefA = scal->coefA, coefB = scal->coefB, S = scal->S;
    }
    const double scale = inv_unit / S;
    stamp_stage(1);
    // The G column(s) of EVERY segment of the chunk are built here, before the point loops: the first one in place, the others
    // into the staging area, from where a boundary costs one LDS copy.  (Rebuilding inside the segment loop let the compiler
    // hoist the logarithm's constants out of that loop and keep them through the point loop: 167 VGPRs; a call that is not
    // inlined pinned the loop's values to the callee-saved registers: 145.)
    Segments sg(gend, ch);
    build_gtile(hist, sg.g, B, GW, cshift, scale, coefA, coefB, phi_q, gtile);
    if constexpr (SEG) {
      for (int s = 0; s < kMaxSegs - 1 && sg.advance(sg.seg_end()); s++) build_gtile(hist, sg.g, B, 1, 0, scale, coefA, coefB, phi_q, s_stage + s * B);
    }
  }
  __syncthreads();
  stamp_stage(2);

  // One 12-double partial PER SEGMENT (slot = the chunk's first slot, Chunk::pad >> 8, + the segment's ordinal): the final
  // reduction sums the slots in table order -- a fixed order, run to run.
  unsigned int slot = SEG || MULTI ? ch.pad >> 8 : blockIdx.x;
  Segments seg(gend, ch);
  for (int s = 0;; s++) {
    double acc[12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
      double zero = 0.0;
      if (SEG) asm volatile("" : "+v"(zero));  // twelve scalars: as a zero VECTOR the initialisation took a 32-register tuple at the top of the loop
      acc[k] = zero;
    }
    const uint32_t seg_end = SEG ? seg.seg_end() : seg.end;
    spline_grad_loop<MODEL, Rec, real, GW1 ? TAP_SINGLE : TAP_COPIES, kThreads, !SEG>(pts + seg.pos, seg_end - seg.pos, seg.g * uint32_t(GW), seg.pos - ch.start, ch.count, img, pitch, W, H, pose,
                                                                               cam, B, cshift, gtile, acc, prio != 0, seg.pos);
    stamp_stage(3);
    grad_reduce_store<kThreads>(acc, s_red, gtile, partials, slot, nslots);
    stamp_stage(4);
    if (!SEG || !seg.advance(seg_end)) break;
    slot++;
    drain_vmem();
    __syncthreads();  // every wave has left the point loop (and s_red) before the tile changes
    for (int k = tid; k < B; k += kThreads) gtile[k] = s_stage[s * B + k];
    __syncthreads();
  }
  const bool last = last_workgroup_arrives<true>(counter, my_blocks, s_flag);
  stamp_stage(5);
  if (last) {
    grad_final_body<kThreads>(partials, int(nslots), qx, qy, qz, qw, out, out_host, tag, s_red, gt.from_partials ? s_fin : nullptr);
    stamp_stage(6);
  }
}

#ifdef NID_FINAL_KERNEL  // (nidreg_core.hip: launch_grad_final)
// standalone finalisation (only launched for an empty cloud, where k_spline_grad has no workgroups)
__global__ __launch_bounds__(kThreads) void k_grad_final(const double* partials, int nblocks, double qx, double qy, double qz, double qw, double* out, double* out_host, double tag) {
  __shared__ double s_red[kWaves * 12];
  grad_final_body<kThreads>(partials, nblocks, qx, qy, qz, qw, out, out_host, tag, s_red);
}
#endif  // NID_FINAL_KERNEL

// GenericCameraBase::project on the device (test / utility path): uv and the 2x3 Jacobian
template <int MODEL, typename real>
__global__ void k_project(const double* __restrict__ p3, long long n, CamParams<real> cam, double* __restrict__ uv, double* __restrict__ jac) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  real u, v, du[3], dv[3];
  project_jac<MODEL, real>(cam, real(p3[3 * i]), real(p3[3 * i + 1]), real(p3[3 * i + 2]), u, v, du, dv);
  uv[2 * i] = double(u);
  uv[2 * i + 1] = double(v);
  if (jac) {
    jac[6 * i + 0] = double(du[0]);
    jac[6 * i + 1] = double(du[1]);
    jac[6 * i + 2] = double(du[2]);
    jac[6 * i + 3] = double(dv[0]);
    jac[6 * i + 4] = double(dv[1]);
    jac[6 * i + 5] = double(dv[2]);
  }
}

}  // namespace nidreg
