// nid_fused.hpp -- ONE kernel per NIDCost evaluation (include/vlcal/costs/nid_cost.hpp:36-107 = histogram, entropy tail,
// Jacobian) instead of three: the launch boundaries between k_spline_hist, k_entropy and k_spline_grad and k_entropy's
// latency chain (16 workgroups, 12.7 us for 1.3 MB of traffic on the headline configuration) are replaced by two grid
// barriers and an entropy step spread over every workgroup.
//
//   phase 1  spline_hist_body on the workgroup's chunk (the histogram pass's own table: one round of co-resident
//            workgroups by construction -- nidreg.hip sizes it from the kernel's real occupancy)
//   barrier  every flush has reached the device's coherence point (the flush is device-scope atomics)
//   phase 2  the workgroup's share of the B x B cells, a SEGMENT OF ONE ROW of hist_image: fixed-point sum of
//            p log(p + eps) and the row-segment sum (no atomics: one store each); zeroes its share of the OTHER
//            histogram buffer for the next evaluation
//   barrier  (cost + Jacobian; a cost-only evaluation ends here: last workgroup by ticket -> scalars, cost, tag)
//   phase 3  every workgroup: the three entropies from the partials (integers: every workgroup gets the same bits) ->
//            NID, coefA / coefB, phi(q_r) in LDS -> its G tile -> spline_grad_loop on the same chunk -> 12 partial sums;
//            last workgroup by ticket: chain rule, gradient + cost + tag to the host
//
// Everything a workgroup reads that another workgroup of the SAME launch wrote goes through agent-scope (sc1) loads of
// data stored by device-scope atomics or agent-scope write-through stores: the barriers need no cache write-back /
// invalidate (a buffer_wbl2 per workgroup costs ~2 us, nid_kernels.hpp last_workgroup_arrives).
//
// A grid barrier needs every workgroup resident.  That holds when the evaluation has the GPU to itself (nidreg.hip launches
// this kernel only then); when something else holds CUs -- another process's kernels, a second fused kernel -- the late
// workgroups start once those finish, and if they do not within the timeout the waiting workgroups raise the abort flag and
// leave: the kernel always terminates, and the host repeats the evaluation with the three-kernel path.
#pragma once
#include "nid_kernels.hpp"

namespace nidreg {

// Everything the kernel needs only AFTER the histogram phase lives in device memory (written once per handle) and is read
// through `st` where it is used: kernel arguments are all loaded at the kernel's entry and would stay live -- in SGPRs, or
// spilled to VGPR lanes and restored inside the hot loops -- through the histogram loop (the first version, with
// everything by value, spilled 81 SGPRs and restored 20 of them per loop iteration).
struct FusedStatic {
  u64* hist_buf[2];
  long long hist_words;
  long long* part_hj;   // [gridDim.x] fixed-point entropy partials
  u64* row_part;        // [kFusedMaxSegs][B] row-segment sums
  double* phi_q;        // [B]  (written by workgroup 0: nidreg_get_hist / tests)
  double* hist_image;   // [B]
  double* hist_points;  // [B]
  EntropyScalars* scal;
  double* partials;     // [12][gridDim.x]
  double* out;
  double* out_host;
  unsigned int* counters;  // [1] gradient ticket, [4] entropy ticket (cost-only evaluations)
  double* abort_host;      // host-mapped mirror of the abort flag (nullable)
};
struct FusedArgs {
  const void* pts;
  const Chunk* chunks;
  const uint8_t* img;
  int pitch, W, H, B, GW, cshift;
  double dn_scale;  // U/36 as a subnormal double (bspline_scale)
  double inv_unit;  // 1 / U
  u64* hist;        // this evaluation's histogram (= st->hist_buf[cur]), zero on entry
  unsigned int* barrier;      // grid-barrier block (kBarrierWords words, zero at handle creation)
  unsigned int bar_base;      // number of grid barriers this handle has completed before this launch
  unsigned int* abort_flag;   // device word; non-zero = a barrier timed out, the evaluation was abandoned
  unsigned long long timeout_ticks;  // of the 100 MHz wall clock
  const FusedStatic* st;
  double q[4];
  double tag;
  int cur;  // index of this evaluation's histogram buffer
  int want_grad;
  int prio;
};

constexpr int kFusedMaxSegs = 8;

__device__ __forceinline__ u64 load_agent(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ long long load_agent(const long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_agent(long long* p, long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_agent(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Grid barrier, two levels.  Device-scope atomics on ONE word are serialised at the memory side (tens of nanoseconds each):
// 512 workgroups arriving at one counter within a few microseconds queue up for ~20 us (measured: the first version of this
// kernel, with a flat counter per barrier, was 23 us slower than the three kernels it replaces,
// profiles/r03a_fused_flat_barrier_kernel_stats.csv).  Here a workgroup arrives at one of kBarrierGroups group counters
// (blockIdx mod kBarrierGroups; 128 bytes apart), the last arrival of a group arrives at the top counter, and the last of
// those publishes the barrier's number in a `go` word that everybody polls (plain sc1 loads, no read-modify-write).
// Counters run on and are never reset: barrier number b (1, 2, ... over the handle's lifetime; the host passes how many
// were completed before the launch) is complete for a group of m members when its counter reaches m b.
// Precondition: everything this workgroup published was stored by device-scope atomics or agent-scope stores.
// Returns false when the evaluation is abandoned (timeout here or in another workgroup).
constexpr int kBarrierGroups = 32;
constexpr int kBarrierStride = 32;  // words between counters: 128 bytes
constexpr int kBarrierWords = (kBarrierGroups + 2) * kBarrierStride;  // groups, top, go
__device__ __forceinline__ bool grid_barrier(unsigned int* bar, unsigned int b, unsigned int* abort_flag, unsigned long long timeout_ticks, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this thread's stores / atomics have been acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int nW = gridDim.x, w = blockIdx.x;
    const unsigned int groups = nW < unsigned(kBarrierGroups) ? nW : unsigned(kBarrierGroups);
    const unsigned int g = w % groups;
    const unsigned int members = nW / groups + (g < nW % groups ? 1u : 0u);
    unsigned int* go = bar + (kBarrierGroups + 1) * kBarrierStride;
    if (__hip_atomic_fetch_add(bar + g * kBarrierStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == members * b) {
      if (__hip_atomic_fetch_add(bar + kBarrierGroups * kBarrierStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == groups * b)
        __hip_atomic_store(go, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int ok = 1;
    const unsigned long long t0 = wall_clock64();
    unsigned int spins = 0;
    while (int(__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - b) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if ((spins++ & 15u) == 0u) {
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || wall_clock64() - t0 > timeout_ticks) {
          ok = 0;
          break;
        }
      }
    }
    // workgroups that gave up had already been counted: a late workgroup can complete the count of an abandoned barrier
    if (ok && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ok = 0;
    if (!ok) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

#ifdef NID_FUSED_STAMP
// development aid (tools/fused_stamps.py): wall-clock stamps (100 MHz) of every workgroup at the phase boundaries
__device__ unsigned long long g_fused_stamp[8 * 2048];
#define NID_FSTAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_fused_stamp[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define NID_FSTAMP(k) do { } while (0)
#endif

// Row segments of phase 2: with nW >= B workgroups a row of hist_image is cut into `segs` segments of L columns, one per
// workgroup (workgroups beyond B * segs have no entropy work); with fewer workgroups each takes ceil(B / nW) whole rows.
struct FusedSplit {
  int segs, L, rows_per_wg;
};
__host__ __device__ __forceinline__ FusedSplit fused_split(int nW, int B) {
  FusedSplit s;
  if (nW >= B) {
    s.segs = nW / B < kFusedMaxSegs ? nW / B : kFusedMaxSegs;
    s.L = (B + s.segs - 1) / s.segs;
    s.rows_per_wg = 1;
  } else {
    s.segs = 1;
    s.L = B;
    s.rows_per_wg = (B + nW - 1) / nW;
  }
  return s;
}

// What the entropy tail needs is split around the second barrier so that as little as possible sits behind it:
//   fused_pre   (histogram complete): the column marginal's entropy term of column `tid` -- hist_points is final
//   fused_post  (every workgroup's row-segment sums and entropy partial published): the three entropies -- integer sums,
//               identical bits in every workgroup -- and phi(q_r) into s_phi.  `writer`: this workgroup also publishes
//               hist_image / phi_q / scal / cost.
// s_redk: 3 * (kT / 64) words of LDS.
__device__ __forceinline__ long long fused_pre(double inv_unit, const u64* hist, const FusedStatic& st, int B, double S, bool writer) {
  const int tid = threadIdx.x;
  long long hp_k = 0;
  if (tid < B) {  // B <= 256 <= threads per workgroup
    const u64* col_sum = hist + size_t(B) * size_t(B) + kTailWords;
    const double cnt = rint(double(load_agent(col_sum + tid)) * inv_unit);  // exact inlier count of column c
    const double p = cnt / S;
    hp_k = ent_fixed(p * log(p + 1e-6));
    if (writer) st.hist_points[tid] = cnt;
  }
  return hp_k;
}
template <int kT>
__device__ __forceinline__ EntropyScalars fused_post(double inv_unit, const FusedStatic& st, int B, int nW, double S, long long hp_k, double* s_phi, long long* s_redk, bool writer) {
  const int tid = threadIdx.x;
  const FusedSplit sp = fused_split(nW, B);
  long long hi_k = 0, hj_k = 0;
  u64 t = 0;
  if (tid < B)
    for (int s = 0; s < sp.segs; s++) t += load_agent(st.row_part + size_t(s) * size_t(B) + tid);
  for (int g = tid; g < nW; g += kT) hj_k += load_agent(st.part_hj + g);
  if (tid < B) {
    const double raw = double(t) * inv_unit;  // raw (un-normalised) hist_image[r]
    const double qv = raw / S;
    const double lq = log(qv + 1e-6);
    hi_k = ent_fixed(qv * lq);
    const double ph = lq + qv / (qv + 1e-6);
    s_phi[tid] = ph;
    if (writer) {
      st.phi_q[tid] = ph;
      st.hist_image[tid] = raw;
    }
  }
  hi_k = wave_sum(hi_k);
  hp_k = wave_sum(hp_k);
  hj_k = wave_sum(hj_k);
  __syncthreads();  // s_redk may still be in use
  if ((tid & 63) == 0) {
    s_redk[(tid >> 6) * 3 + 0] = hi_k;
    s_redk[(tid >> 6) * 3 + 1] = hp_k;
    s_redk[(tid >> 6) * 3 + 2] = hj_k;
  }
  __syncthreads();
  long long A = 0, Bk = 0, C = 0;
  for (int w = 0; w < kT / 64; w++) {
    A += s_redk[w * 3 + 0];
    Bk += s_redk[w * 3 + 1];
    C += s_redk[w * 3 + 2];
  }
  const EntropyScalars e = entropy_scalars(A, Bk, C, S);
  if (writer && tid == 0) {
    *st.scal = e;
    // agent scope: grad_final_body (another workgroup) mirrors these to the host behind the gradient
    store_agent(&st.out[0], e.nid);
    store_agent(&st.out[8], e.status);
    store_agent(&st.out[9], S);
  }
  return e;
}

// 16 waves per CU (two 8-wave or four 4-wave workgroups: the chunk tables are one such round) need <= 128 VGPRs: asked for
// explicitly where the loops fit (the fisheye / equirectangular gradient loops hold 154-161: those instantiations keep the
// compiler's choice and nidreg.hip leaves their handles on the three-kernel route)
constexpr int fused_min_waves(int model) { return (model == MODEL_FISHEYE || model == MODEL_EQUIRECT) ? 1 : 4; }
template <int MODEL, typename Rec, typename real, bool WIDE>
__global__ __launch_bounds__(WIDE ? kWideThreads : kThreads, fused_min_waves(MODEL)) void k_fused(PoseParams<real> pose, CamParams<real> cam, FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int kT = WIDE ? kWideThreads : kThreads;
  constexpr int kNW = kT / 64;
  int B = a.B, GW = a.GW, cshift = a.cshift;
  if (WIDE) {
    B = 256;
    GW = 1;
    cshift = kWideShift;
  }
  const int tid = threadIdx.x;
  const int nW = int(gridDim.x), w = int(blockIdx.x);
  const Chunk ch = a.chunks[w];
  const Rec* pts = static_cast<const Rec*>(a.pts);

  // LDS behind the histogram / G tile
  unsigned char* extra = smem + ((spline_hist_lds_bytes(B, GW, cshift) + 15) & ~size_t(15));
  double* s_phi = reinterpret_cast<double*>(extra);                  // [256]
  double* s_red = s_phi + 256;                                       // [kNW * 12] (also 3 * kNW 64-bit integers)
  long long* s_redk = reinterpret_cast<long long*>(s_red);
  u64* s_row = reinterpret_cast<u64*>(s_red + kNW * 12);             // [kNW]
  int* s_flag = reinterpret_cast<int*>(s_row + kNW);

  // ---- phase 1: joint histogram of this chunk
  NID_FSTAMP(0);
  spline_hist_body<MODEL, Rec, real, WIDE, kT>(pts, ch, a.img, a.pitch, a.W, a.H, pose, cam, B, GW, cshift, a.dn_scale, a.hist, smem, a.prio != 0);
  NID_FSTAMP(1);
  // the arguments used from here on are re-read from the kernel-argument segment where they are needed (see FusedStatic)
  typedef __attribute__((address_space(4))) const unsigned char kernarg_bytes_t;
  kernarg_bytes_t* ka_base = (kernarg_bytes_t*)__builtin_amdgcn_kernarg_segment_ptr();
  constexpr size_t ka_off = (sizeof(PoseParams<real>) + sizeof(CamParams<real>) + 7) & ~size_t(7);
  const __attribute__((address_space(4))) FusedArgs& al = *reinterpret_cast<const __attribute__((address_space(4))) FusedArgs*>(ka_base + ka_off);
  const FusedStatic& st = *al.st;
  if (!grid_barrier(a.barrier, a.bar_base + 1u, a.abort_flag, a.timeout_ticks, s_flag)) {
    if (tid == 0 && st.abort_host) *st.abort_host = 1.0;
    return;
  }
  NID_FSTAMP(2);

  // ---- phase 2: this workgroup's row segment(s): sum p log(p + eps) (fixed point), row-segment sums; and everything of
  // phase 3's prologue that needs only the finished histogram: the column-marginal term, phi(p) of the own column(s)
  const int want_grad = al.want_grad;
  const double inv_unit = al.inv_unit;
  u64* hist = st.hist_buf[al.cur];
  const double S = double(load_agent(hist + size_t(B) * size_t(B) + kTailInliers));
  const double scale = inv_unit / S;  // fixed-point word -> probability
  const int tile_n = GW * B;
  const u64* own = hist + size_t(ch.group) * size_t(tile_n);
  const int own_n = min(GW, B - int(ch.group) * GW) * B;
  double* gtile = reinterpret_cast<double*>(smem);  // the histogram tile's LDS, free since the flush
  double phi_own = 0.0;  // WIDE: phi(p) of cell tid & 255 of the own column
  if (want_grad) {
    if (WIDE) {
      const double p = double(load_agent(own + (tid & 255))) * scale;
      phi_own = log(p + 1e-6) + p / (p + 1e-6);
    } else {
      for (int k = tid; k < own_n; k += kT) {
        const double p = double(load_agent(own + k)) * scale;
        gtile[uint32_t(k) << cshift] = log(p + 1e-6) + p / (p + 1e-6);  // parked in the cell's first copy until coefA / coefB are known
      }
    }
  }
  const long long hp_k = fused_pre(inv_unit, hist, st, B, S, w == 0);
  {
    const FusedSplit sp = fused_split(nW, B);
    long long ek = 0;
    int r0, r1, seg;
    if (nW >= B) {
      r0 = w / sp.segs;
      seg = w % sp.segs;
      r1 = r0 < B ? r0 + 1 : r0;  // workgroups beyond B * segs: no rows
    } else {
      seg = 0;
      r0 = min(B, w * sp.rows_per_wg);
      r1 = min(B, r0 + sp.rows_per_wg);
    }
    const int c0 = seg * sp.L, c1 = min(B, c0 + sp.L);
    for (int r = r0; r < r1; r++) {
      u64 row = 0;
      for (int c = c0 + tid; c < c1; c += kT) {
        const u64 v = load_agent(hist + size_t(c) * size_t(B) + size_t(r));
        if (v) {
          const double p = double(v) * scale;
          ek += ent_fixed(p * log(p + 1e-6));
        }
        row += v;
      }
      row = u64(wave_sum((long long)row));
      __syncthreads();  // s_row of the previous row has been read
      if ((tid & 63) == 0) s_row[tid >> 6] = row;
      __syncthreads();
      if (tid == 0) {
        u64 t = 0;
        for (int k = 0; k < kNW; k++) t += s_row[k];
        store_agent(st.row_part + size_t(seg) * size_t(B) + size_t(r), t);
      }
    }
    ek = wave_sum(ek);
    __syncthreads();
    if ((tid & 63) == 0) s_redk[tid >> 6] = ek;
    __syncthreads();
    if (tid == 0) {
      long long t = 0;
      for (int k = 0; k < kNW; k++) t += s_redk[k];
      store_agent(st.part_hj + w, t);
    }
    // double-buffered histogram: clear the buffer the NEXT evaluation accumulates into
    u64* zero_buf = st.hist_buf[al.cur ^ 1];
    const long long zero_words = st.hist_words;
    for (long long k = (long long)w * kT + tid; k < zero_words; k += (long long)nW * kT) store_agent(zero_buf + k, u64(0));
  }

  NID_FSTAMP(3);
  if (!want_grad) {
    // cost only: the last workgroup to get here finalises (every partial above was stored at agent scope)
    if (last_workgroup_arrives<true>(st.counters + 4, unsigned(nW), s_flag)) {
      const EntropyScalars e = fused_post<kT>(inv_unit, st, B, nW, S, hp_k, s_phi, s_redk, true);
      double* out_host = st.out_host;
      if (tid == 0 && out_host) {
        out_host[0] = e.nid;
        out_host[8] = e.status;
        out_host[9] = e.S;
        __threadfence_system();
        __hip_atomic_store(&out_host[15], al.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    return;
  }
  if (!grid_barrier(al.barrier, al.bar_base + 2u, al.abort_flag, al.timeout_ticks, s_flag)) {
    if (tid == 0 && st.abort_host) *st.abort_host = 1.0;
    return;
  }
  NID_FSTAMP(4);

  // ---- phase 3: entropies -> G tile -> gradient of this chunk
  const EntropyScalars e = fused_post<kT>(inv_unit, st, B, nW, S, hp_k, s_phi, s_redk, w == 0);
  {
    // G = (coefA phi(p) + coefB phi(q_r)) / 12: the tap loop works with 6 b and 2 db/ds (bspline6 / bspline_deriv2)
    const uint32_t cmask = (1u << cshift) - 1u;
    if (WIDE) {
      // 256 cells x 32 copies at byte address (cell << 8) | (copy << 3): thread t takes cell t & 255, the 16 copies of half t >> 8
      const int cell = tid & 255, half = tid >> 8;
      const double gval = (e.coefA * phi_own + e.coefB * s_phi[cell]) * (1.0 / 12.0);
      static_assert(kWideThreads == 512, "the G tile fill assumes two threads per cell");
#pragma unroll
      for (int j = 0; j < 16; j++) gtile[(cell << kWideShift) + half * 16 + j] = gval;
    } else {
      for (int k = tid; k < own_n; k += kT) {
        const double gval = (e.coefA * gtile[uint32_t(k) << cshift] + e.coefB * s_phi[k % B]) * (1.0 / 12.0);
        for (uint32_t j = 0; j <= cmask; j++) gtile[(uint32_t(k) << cshift) + ((j + uint32_t(k)) & cmask)] = gval;
      }
    }
  }
  __syncthreads();
  NID_FSTAMP(5);

  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.0;
  spline_grad_loop<MODEL, Rec, real, WIDE ? TAP_WIDE : TAP_COPIES, kT>(pts, ch, a.img, a.pitch, a.W, a.H, pose, cam, B, GW, cshift, gtile, acc, a.prio != 0);
  NID_FSTAMP(6);
  __syncthreads();  // s_red aliases the reduction scratch of fused_scalars
  grad_reduce_store<kT>(acc, s_red, st.partials, unsigned(w), unsigned(nW));
  NID_FSTAMP(7);
  if (last_workgroup_arrives<true>(st.counters + 1, unsigned(nW), s_flag))
    grad_final_body<kT>(st.partials, nW, al.q[0], al.q[1], al.q[2], al.q[3], st.out, st.out_host, al.tag, s_red);
}

// dynamic LDS of k_fused: the histogram / G tile + phi table + reduction scratch
__host__ __device__ __forceinline__ size_t fused_lds_bytes(int B, int GW, int cshift, int threads) {
  return ((spline_hist_lds_bytes(B, GW, cshift) + 15) & ~size_t(15)) + 256 * 8 + size_t(threads / 64) * 12 * 8 + size_t(threads / 64) * 8 + 16;
}

}  // namespace nidreg
