// nid_fused.hpp -- ONE launch per cost+Jacobian evaluation for clouds that fit on chip (round 6).
//
// The three-kernel route pays, on a small cloud, mostly for things that are not arithmetic: two kernel boundaries, a second sweep
// over the records (load, transform, project -- everything up to the knot is computed twice), a second image gather.  CDNA4 has
// 160 KB of LDS per CU: a grid of at most one round of co-resident workgroups can keep what pass A knows about its points ON CHIP
// and hand it to pass B across a grid barrier:
//
//   phase A   k_spline_hist's arithmetic, bit for bit (the integer histogram does not depend on the tiling), per point additionally
//             the STASH into LDS: FULL = (u, v), the six values project_bwd needs (ProjCtx), the 4 x 4 patch of image bins, the raw
//             record -- 80 B + sizeof(Rec) per point; UV = (u, v) only, 16 B per point, for chunks too long for the full stash;
//   flush     the workgroup's tile into the global integer histogram (device-scope atomics, as before);
//   barrier   every workgroup of the grid has flushed (one agent-scope counter: arrive, spin; bounded by the wall clock);
//   phase P   the entropy tail + G tile exactly as k_spline_grad's prologue for small tables (every workgroup sums the table itself:
//             grad_scalars_from_partials<SELF>) -- so the cost has the bits of the three-kernel route;
//   phase B   the gradient taps from the stash: FULL reads nothing from global memory at all and runs no transform / projection;
//             UV re-reads its records and re-runs project_fwd for the Jacobian context but takes (u, v) from LDS;
//   end       partial sums, ticket, last workgroup finalises -- k_spline_grad's epilogue unchanged.
//
// Tables of at most kSelfEntropyCells cells only (B <= 32; the reference's default is 16 bins): the entropy work then needs no
// second barrier.  Chunks lie inside one column group (the host builds the table that way).
//
// CO-RESIDENCY is what the barrier rests on: the host launches this kernel only with a grid that fits the device at once
// (occupancy x CUs, checked at table-build time) and only for an evaluation that has the device to itself (InflightGuard: no
// other evaluation of this process in flight there).  Another PROCESS running the same kernel on the same GPU could still
// interleave two half-resident grids; the barrier therefore gives up after `timeout_ticks` of the 100 MHz wall clock, the
// kernel ends without its completion tag, and the host re-runs the evaluation on the three-kernel route and stops using this
// one for the handle (nidreg_core.hip eval_one).
#pragma once
#include "nid_kernels.hpp"

namespace nidreg {

// LDS layout (bytes from the start of the dynamic segment):
//   [0, region)                 histogram tile (u64, 2^cshift copies) -> G tile (double, same layout) -> reduction scratch
//   s_red   kWaves * 12 doubles
//   s_phi   256 doubles          (phi(q_r); the SELF entropy path keeps its row sums behind the first 128)
//   s_flag  4 ints, s_fin 4 doubles, s_cols 32 u64, s_rows 32 u64 (column / row sums of the finished table, phase P)
//   stash   FULL: d8[8][cap] doubles, c4[cap] uint4, rec[cap] Rec;  UV: d2[2][cap] doubles
__host__ __device__ __forceinline__ size_t fused_fixed_lds_bytes(int B, int GW, int cshift) {
  return grad_tile_region_bytes(B, GW, cshift) + size_t(kWaves) * 12 * 8 + 256 * 8 + 16 + 32 + 32 * 8 + 32 * 8;
}
__host__ __device__ __forceinline__ size_t fused_stash_bytes_per_point(bool full, size_t rec_bytes) { return full ? 64 + 16 + rec_bytes : 16; }
__host__ __device__ __forceinline__ size_t fused_lds_bytes(int B, int GW, int cshift, bool full, size_t rec_bytes, int cap) {
  return fused_fixed_lds_bytes(B, GW, cshift) + fused_stash_bytes_per_point(full, rec_bytes) * size_t(cap);
}

// The grid barrier.  `counter` counts arrivals over the handle's lifetime, `target` = arrivals once every workgroup of THIS launch
// has arrived, `flags` = one word per workgroup, each in a cache line of its own (stride kFusedFlagStride words), `epoch` = this
// launch's number.  Everything the other workgroups must see was written with device-scope atomics (performed at the coherence
// point) and has been waited for (vmcnt) before the arrival.
//   * The workgroup whose arrival completes the count RELEASES the others by storing the epoch into every workgroup's own flag word;
//     a waiting workgroup polls its OWN word.  (First build: everybody polled the counter itself -- several hundred agent-scope loads
//     of one address in flight at the coherence point, each poll queued behind all the others: 16 ns per workgroup of the grid, 8 us
//     at 505 workgroups; profiles/r06_experiments.md.)
//   * NO acquire fence behind the wait: at agent scope it invalidates the XCD's L2 -- once per workgroup, i.e. dozens of times per
//     XCD while its other workgroups are reading records, image patches and their G tiles (1M points: 53 -> 48.8 us without it).
//     What phase P reads of the other workgroups' writes, it reads with agent-scope loads instead (ld_hist<true>): a few hundred
//     words per workgroup.
constexpr int kFusedFlagStride = 16;  // u64 words: 128 bytes
__device__ __forceinline__ bool fused_grid_barrier(u64* counter, u64 target, u64* flags, u64 epoch, unsigned long long timeout_ticks, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's flush atomics have been performed
  __syncthreads();
  if (threadIdx.x == 0) {
    const u64 old = __hip_atomic_fetch_add(counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = (old + 1 == target) ? 2 : 1;
  }
  __syncthreads();
  const int role = *s_flag;
  __syncthreads();  // (every thread has read its role before thread 0 reuses the word below)
  if (role == 2) {
    for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x) __hip_atomic_store(&flags[size_t(i) * kFusedFlagStride], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  if (threadIdx.x == 0) {
    const u64* mine = &flags[size_t(blockIdx.x) * kFusedFlagStride];
    int ok = 1;
    if (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
      const unsigned long long t0 = wall_clock64();
      for (unsigned spins = 1;; spins++) {
        if (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch) break;
        if ((spins & 15u) == 0) {
          if (wall_clock64() - t0 > timeout_ticks) {
            ok = 0;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

template <int MODEL, typename Rec, bool FULL>
__global__ __launch_bounds__(kThreads) void k_spline_fused(
  const Rec* __restrict__ pts, const Chunk* __restrict__ chunks, const uint8_t* __restrict__ img, int pitch, int W, int H, PoseParams<double> pose, CamParams<double> cam, int B, int GW,
  int cshift, double dn_scale, double inv_unit, u64* __restrict__ hist, GradTail gt, double* partials, double qx, double qy, double qz, double qw, double* out, double* out_host, double tag,
  unsigned int* counter, u64* barrier, u64 barrier_target, u64* barrier_flags, u64 barrier_epoch, unsigned long long timeout_ticks, int cap) {
  typedef double real;
  constexpr int kT = kThreads;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const size_t region = grad_tile_region_bytes(B, GW, cshift);
  u64* tile = reinterpret_cast<u64*>(smem);
  double* gtile = reinterpret_cast<double*>(smem);
  double* s_red = reinterpret_cast<double*>(smem + region);
  double* s_phi = s_red + kWaves * 12;
  int* s_flag = reinterpret_cast<int*>(s_phi + 256);
  double* s_fin = reinterpret_cast<double*>(s_flag + 4);
  unsigned long long* s_cols = reinterpret_cast<unsigned long long*>(s_fin + 4);
  unsigned long long* s_rows = s_cols + 32;
  unsigned char* stash = reinterpret_cast<unsigned char*>(s_rows + 32);
  double* sd = reinterpret_cast<double*>(stash);                                       // FULL: [8][cap], UV: [2][cap]
  uint4* sc = reinterpret_cast<uint4*>(stash + size_t(8) * size_t(cap) * 8);            // FULL: [cap]
  Rec* sr = reinterpret_cast<Rec*>(stash + size_t(8) * size_t(cap) * 8 + size_t(cap) * 16);  // FULL: [cap]

  const int tid = threadIdx.x;
  const Chunk ch = chunks[blockIdx.x];
  const uint32_t cnt = ch.count;  // <= cap (host)
  const uint32_t col0 = ch.group * uint32_t(GW);
  const int tile_n = GW * B;
  const int tile_w = tile_n << cshift;
  const uint32_t cmask = (1u << cshift) - 1u;
  const uint32_t lane_copy = uint32_t(tid) & cmask;
  const real fW = real(W), fH = real(H);

  // ------------------------------------------------------------------ phase A (spline_hist_body, generic tiling, one segment)
  stamp_stage(0);  // (-DNID_STAMP builds, tools/fused_stage_times.py: 0 entry, 1 points done, 2 flush issued, 3 barrier passed, 4 G tile built, 5 taps done, 6 ticket drawn, 7 results written)
  for (int k = tid; k < tile_w; k += kT) tile[k] = 0;
  if (tid < 64) s_cols[tid] = 0;  // (s_cols and s_rows, for phase P)
  __syncthreads();
  {
    const BsplineScale KU = bspline_scale(dn_scale);
    const char* rec_base = reinterpret_cast<const char*>(pts + ch.start);
    // the same instructions per point as k_spline_hist's `taps` (the bits a point contributes do not depend on the route);
    // additionally hands the 4 x 4 patch back for the stash
    auto taps = [&](real uc, real vc, uint32_t bin, const BsplineScale& K, uint32_t* cols) {
      const int kx = int(uc), ky = int(vc);
      double bxs[4];
      real by[4];
      bspline_scaled(double(m_abs(m_fract(uc))), K, bxs);
      bspline6<real>(m_abs(m_fract(vc)), by);
      u64* col = tile + ((((bin - col0) * uint32_t(B)) << cshift) + lane_copy);
      load_patch(img, pitch, kx, ky, cols);
#pragma unroll
      for (int b = 0; b < 4; b++) {
#pragma unroll
        for (int a = 0; a < 4; a++) {
          const uint32_t r = (cols[a] >> (8 * b)) & 0xffu;
          atomicAdd(&col[r << cshift], to_fixed_dn(bxs[a], double(by[b])));
        }
      }
    };
    RawBatch<Rec, kUnroll> rb;
    for (uint32_t base = 0; base < cnt; base += kT * kUnroll) {
      real xs[kUnroll], ys[kUnroll], zs[kUnroll];
      uint32_t bins_[kUnroll];
#pragma unroll
      for (int k = 0; k < kUnroll; k++) rb.load(rec_base, min(base + uint32_t(k) * kT + tid, cnt - 1u) * uint32_t(sizeof(Rec)), k);
#pragma unroll
      for (int k = 0; k < kUnroll; k++) rb.template get<real>(k, xs[k], ys[k], zs[k], bins_[k]);
      real us[kUnroll], vs[kUnroll];
      bool ins[kUnroll];
      bool all_in = true;
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
        const uint32_t slot = base + uint32_t(k) * kT + tid;
        const bool valid = slot < cnt;
        real cx, cy, cz;
        transform_fma<real>(pose, xs[k], ys[k], zs[k], cx, cy, cz);
        if constexpr (FULL) {
          ProjCtx<real> ctx;
          project_fwd<MODEL, real>(cam, cx, cy, cz, us[k], vs[k], ctx);  // (u, v): the very call k_spline_grad makes -- and k_spline_hist's value
          if (valid) {
#pragma unroll
            for (int j = 0; j < 6; j++) sd[size_t(2 + j) * cap + slot] = double(ctx.a[j]);
            rb.store(k, &sr[slot]);  // (the raw record: x y z for M += gp p^T, the column)
          }
        } else {
          project<MODEL, real, real, true>(cam, cx, cy, cz, us[k], vs[k]);
        }
        ins[k] = bool(int(valid) & int(us[k] >= real(0)) & int(us[k] < fW) & int(vs[k] >= real(0)) & int(vs[k] < fH));
        if (valid) {
          sd[size_t(0) * cap + slot] = ins[k] ? double(us[k]) : -1.0;  // an outlier fails phase B's range test by its stashed u
          sd[size_t(1) * cap + slot] = double(vs[k]);
        }
        all_in = bool(int(all_in) & int(ins[k]));
      }
      if (__builtin_amdgcn_ballot_w64(!all_in) == 0) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
          uint32_t cols[4];
          taps(us[k], vs[k], bins_[k], KU, cols);
          if constexpr (FULL) sc[base + uint32_t(k) * kT + tid] = make_uint4(cols[0], cols[1], cols[2], cols[3]);  // (all lanes valid here)
        }
      } else {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) {
          const bool in = ins[k];
          BsplineScale KL;
          KL.k16 = in ? KU.k16 : 0.0;
          KL.k46 = in ? KU.k46 : 0.0;
          KL.k05 = in ? KU.k05 : 0.0;
          KL.k1 = in ? KU.k1 : 0.0;
          uint32_t cols[4];
          taps(in ? us[k] : real(0), in ? vs[k] : real(0), bins_[k], KL, cols);
          if constexpr (FULL) {
            const uint32_t slot = base + uint32_t(k) * kT + tid;
            if (slot < cnt) sc[slot] = make_uint4(cols[0], cols[1], cols[2], cols[3]);
          }
        }
      }
    }
    __syncthreads();
    stamp_stage(1);
    // flush: contiguous in the [bin_points][bin_image] device layout
    u64* dst = hist + size_t(ch.group) * size_t(tile_n);
    for (int k = tid; k < tile_n; k += kT) {
      u64 vv = 0;
      for (uint32_t j = 0; j <= cmask; j++) vv += tile[(uint32_t(k) << cshift) + ((j + uint32_t(k)) & cmask)];
      if (vv) atomicAdd(&dst[k], vv);
    }
    // (no column sums, no inlier count here: on this route every workgroup derives both from the finished cells in phase P --
    // two dependent rounds of global atomics less on the way to the barrier)
  }
  // the next evaluation's histogram buffer (k_entropy's other duty on the three-kernel route): independent of the barrier
  if (gt.zero_buf)
    for (long long k = (long long)blockIdx.x * kT + tid; k < gt.zero_words; k += (long long)gridDim.x * kT) gt.zero_buf[k] = 0;

  stamp_stage(2);
  // ------------------------------------------------------------------ grid barrier: the histogram is complete
  if (!fused_grid_barrier(barrier, barrier_target, barrier_flags, barrier_epoch, timeout_ticks, s_flag)) return;  // no tag: the host falls back (see the header)

  stamp_stage(3);
  // ------------------------------------------------------------------ phase P: entropy tail + G tile from ONE round of loads
  // The integers and the floating-point expressions are those of grad_scalars_from_partials<SELF> / build_gtile (nid_kernels.hpp), so
  // the cost, the marginals and the G values have the bits of the three-kernel route; what differs is where the integers come from:
  // every thread fetches its (<= 4) cells of the finished table once, with agent-scope loads; row sums, column sums and -- from the
  // column sums: rint(colsum / unit) is a column's exact inlier count -- the inlier count S are LDS sums of those cells; the G tile
  // reuses the cells and their logarithms from registers.
  {
    const int ncell = B * B;
    u64 vc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + i * kT;
      vc[i] = c < ncell ? ld_hist<true>(&hist[c]) : 0;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + i * kT;
      if (vc[i]) {  // device layout [bin_points][bin_image]: c / B = the column (points bin), c % B = the row (image bin)
        atomicAdd(&s_cols[c / B], (unsigned long long)vc[i]);
        atomicAdd(&s_rows[c % B], (unsigned long long)vc[i]);
      }
    }
    __syncthreads();
    double* s_S = s_fin + 3;
    if (tid < 64) {
      const double cnt = tid < B ? rint(double(s_cols[tid]) * inv_unit) : 0.0;
      const double Sw = wave_sum(cnt);  // integers below 2^53: exact in any order
      if (tid == 0) *s_S = Sw;
    }
    __syncthreads();
    const double S = *s_S;
    const double scale = inv_unit / S;
    long long hj = 0, hi_k = 0, hp_k = 0;
    double lg[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double p = double(vc[i]) * scale;
      lg[i] = fast_log(p + 1e-6);
      if (vc[i]) hj += ent_fixed(p * lg[i]);
    }
    if (tid < B) {
      const int r = tid;
      const double raw = double(s_rows[r]) * inv_unit;
      const double qv = raw / S;
      const double lq = fast_log(qv + 1e-6);
      hi_k = ent_fixed(qv * lq);
      const double ph = lq + qv / (qv + 1e-6);
      s_phi[r] = ph;
      const double cnt = rint(double(s_cols[r]) * inv_unit);
      const double p = cnt / S;
      hp_k = ent_fixed(p * fast_log(p + 1e-6));
      if (blockIdx.x == 0) {
        gt.phi_q[r] = ph;
        gt.hist_image[r] = raw;
        gt.hist_points[r] = cnt;
        hist[size_t(ncell) + kTailWords + r] = u64(s_cols[r]);  // where the getters (and a reader of the raw buffer) look for them
      }
    }
    long long* s_redk = reinterpret_cast<long long*>(s_red);
    hj = wave_sum(hj);
    hi_k = wave_sum(hi_k);
    hp_k = wave_sum(hp_k);
    if ((tid & 63) == 0) {
      s_redk[(tid >> 6) * 3 + 0] = hi_k;
      s_redk[(tid >> 6) * 3 + 1] = hp_k;
      s_redk[(tid >> 6) * 3 + 2] = hj;
    }
    __syncthreads();
    long long A = 0, Bk = 0, C = 0;
    for (int w = 0; w < kT / 64; w++) {
      A += s_redk[w * 3 + 0];
      Bk += s_redk[w * 3 + 1];
      C += s_redk[w * 3 + 2];
    }
    const EntropyScalars es = entropy_scalars(A, Bk, C, S);
    if (tid == 0) {
      s_fin[0] = es.nid, s_fin[1] = es.status, s_fin[2] = es.S;
      if (blockIdx.x == 0) {
        *gt.scal = es;
        __hip_atomic_store(&out[0], es.nid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out[8], es.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&out[9], S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hist[size_t(ncell) + kTailInliers] = u64(S);
      }
    }
    // G tile of this workgroup's column group (build_gtile's expression; the cell's logarithm is the one computed above)
    const int ncols = min(GW, B - int(ch.group) * GW);
    const int lo = int(ch.group) * tile_n, n = ncols * B;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int k = tid + i * kT - lo;
      if (k >= 0 && k < n) {
        const double p = double(vc[i]) * scale;
        const double gval = (es.coefA * (lg[i] + p / (p + 1e-6)) + es.coefB * s_phi[k % B]) * (1.0 / 12.0);
        for (uint32_t j = 0; j <= cmask; j++) gtile[(uint32_t(k) << cshift) + ((j + uint32_t(k)) & cmask)] = gval;
      }
    }
  }
  __syncthreads();

  stamp_stage(4);
  // ------------------------------------------------------------------ phase B (spline_grad_loop's point body, fed from the stash)
  double acc[12];
#pragma unroll
  for (int k = 0; k < 12; k++) acc[k] = 0.0;
  {
    const char* rec_base = reinterpret_cast<const char*>(pts + ch.start);
    // (A branch-free form of this loop -- the thread's four points of a batch interleaved, outliers masked by selects -- was measured on
    // the FULL stash: 100k points 23.5 -> 23.5 us, 300k 29.4 -> 31.5 (182 VGPRs); profiles/r06_experiments.md.  Not kept.)
    for (uint32_t base = 0; base < cnt; base += kT * kUnroll) {
      RawBatch<Rec, kUnroll> rb;
      if constexpr (!FULL) {
#pragma unroll
        for (int k = 0; k < kUnroll; k++) rb.load(rec_base, min(base + uint32_t(k) * kT + tid, cnt - 1u) * uint32_t(sizeof(Rec)), k);
      }
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
        const uint32_t slot = base + uint32_t(k) * kT + tid;
        if (slot >= cnt) break;
        real x, y, z;
        uint32_t bin;
        ProjCtx<real> ctx;
        uint32_t cols[4];
        const real uu = sd[size_t(0) * cap + slot], vv = sd[size_t(1) * cap + slot];
        const bool in = (uu >= real(0)) && (uu < fW) && (vv >= real(0)) && (vv < fH);
        if constexpr (FULL) {
          load_rec<real>(&sr[slot], x, y, z, bin);
        } else {
          rb.template get<real>(k, x, y, z, bin);
        }
        if (in) {
          const int kx = int(uu), ky = int(vv);
          if constexpr (FULL) {
#pragma unroll
            for (int j = 0; j < 6; j++) ctx.a[j] = sd[size_t(2 + j) * cap + slot];
            const uint4 c4 = sc[slot];
            cols[0] = c4.x, cols[1] = c4.y, cols[2] = c4.z, cols[3] = c4.w;
          } else {
            real cx, cy, cz, u2, v2;
            transform_fma<real>(pose, x, y, z, cx, cy, cz);
            project_fwd<MODEL, real>(cam, cx, cy, cz, u2, v2, ctx);
            load_patch(img, pitch, kx, ky, cols);
          }
          const real sx = m_abs(m_fract(uu)), sy = m_abs(m_fract(vv));
          real bx[4], by[4], dbx[4], dby[4];
          bspline6<real>(sx, bx);
          bspline6<real>(sy, by);
          bspline_deriv2<real>(sx, dbx);
          bspline_deriv2<real>(sy, dby);
          const double* gcol = gtile + ((((bin - col0) * uint32_t(B)) << cshift) + lane_copy);
          real gx = real(0), gy = real(0);
#pragma unroll
          for (int b = 0; b < 4; b++) {
            real sa = real(0), sb = real(0);
#pragma unroll
            for (int a = 0; a < 4; a++) {
              const uint32_t r = (cols[a] >> (8 * b)) & 0xffu;
              const real g = gcol[r << cshift];
              sa = fma(g, dbx[a], sa);
              sb = fma(g, bx[a], sb);
            }
            gx = fma(sa, by[b], gx);
            gy = fma(sb, dby[b], gy);
          }
          real gpr[3];
          project_bwd<MODEL, real>(cam, ctx, gx, gy, gpr);
          const double gp0 = gpr[0], gp1 = gpr[1], gp2 = gpr[2];
          acc[0] = fma(gp0, x, acc[0]);
          acc[1] = fma(gp0, y, acc[1]);
          acc[2] = fma(gp0, z, acc[2]);
          acc[3] = fma(gp1, x, acc[3]);
          acc[4] = fma(gp1, y, acc[4]);
          acc[5] = fma(gp1, z, acc[5]);
          acc[6] = fma(gp2, x, acc[6]);
          acc[7] = fma(gp2, y, acc[7]);
          acc[8] = fma(gp2, z, acc[8]);
          acc[9] += gp0;
          acc[10] += gp1;
          acc[11] += gp2;
        }
      }
    }
  }
  // ------------------------------------------------------------------ end (k_spline_grad's epilogue)
  stamp_stage(5);
  grad_reduce_store<kT>(acc, s_red, gtile, partials, blockIdx.x, gridDim.x);
  // (the finalising workgroup reads the partials with agent-scope loads instead of an acquire fence + plain loads: no difference
  // measured -- 23.1 / 38.9 us either way at 100k / 1M points --, one L2 invalidation less)
  const bool last = last_workgroup_arrives<true, false>(counter, gridDim.x, s_flag);
  stamp_stage(6);
  if (last) {
    grad_final_body<kT, true>(partials, int(gridDim.x), qx, qy, qz, qw, out, out_host, tag, s_red, s_fin);
    stamp_stage(7);
  }
}

}  // namespace nidreg
