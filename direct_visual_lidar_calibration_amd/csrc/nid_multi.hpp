// nid_multi.hpp -- per-pair table of the single-grid multi-pair evaluation (shared by the kernels and the host side).
#pragma once
#include <stdint.h>

#include <type_traits>

namespace nidreg {

typedef unsigned long long u64;
struct EntropyScalars;

// MultiNIDCost on one GPU (visual_camera_calibration.cpp:141-178: the same pose for every pair): ONE grid per pass over
// the chunks of ALL pairs instead of three launches per pair.  Chunk::pad carries the pair's index into this table (low 8
// bits; above them the first of the chunk's slots in the pair's gradient partials, one per segment) (device
// memory, built once per set of handles); what changes per evaluation -- which of the two histogram buffers is current,
// the completion tag -- travels by value (MultiDyn).  multi == nullptr: the single-pair launch, unchanged.
constexpr int kMaxMulti = 16;
struct MultiEntry {
  const void* pts;
  const uint32_t* gend;  // end offsets of the pair's column groups among its records (nid_kernels.hpp Segments)
  const uint8_t* img;
  u64* hist_buf[2];
  double k16;       // U/36 as a subnormal double (bspline_scale)
  double inv_unit;  // 1 / U
  long long* part_hj;  // fixed-point entropy partials (ent_fixed)
  u64* row_part;
  double* phi_q;
  double* hist_image;
  double* hist_points;
  EntropyScalars* scal;
  double* partials;
  double* out;
  double* out_host;
  unsigned int* counters;  // [0] entropy ticket, [1] gradient ticket
  long long zero_words;
  int nslots;   // segments (= 12-double gradient partials) of this pair in the combined gradient-pass table; a chunk's first: Chunk::pad >> 8
  int nchunks;  // chunks of this pair in that table (the last-workgroup ticket counts them)
};
struct NoMultiDyn {};  // what the single-pair instantiations take in its place (no kernel-argument bytes, no branch)
struct MultiDyn {
  double tag[kMaxMulti];
  // index of the histogram buffer this evaluation accumulates into.  32-bit on purpose: with a byte array the compiler
  // folded `&dyn + pair` into a common base for cur[pair] and tag[pair] and read the tag with a scalar load at base + 7 pair
  // -- scalar loads ignore the two low address bits, so pair 1 got the bytes at offset 4 (found on hardware, round 3)
  int cur[kMaxMulti];
  int want_grad;
  int neb;  // entropy workgroups per pair
};


// A pointer READ FROM DEVICE MEMORY (this table) is a generic pointer to the compiler: every access through it becomes a
// FLAT instruction -- 64-bit per-lane addresses instead of an SGPR base + 32-bit lane offset, and, worse, FLAT loads count
// on lgkmcnt as well as vmcnt, so each wait for an LDS atomic of the tap loop also waited for the record prefetch: the
// multi-pair kernels ran 1.35-1.65x slower than the single-pair kernels on the same points (round 3, 85 + 100 us against
// 51 + 74 us).  Kernel-argument pointers are known to be global; these are told to be.
#if defined(__HIPCC__)
template <typename T>
__device__ __forceinline__ T* as_global(T* p) {
  // through an integer: a plain generic -> global -> generic cast pair is folded away again before the address-space
  // inference runs (and an assume(!is_shared && !is_private) did not reach it either; both checked in the ISA)
  typedef __attribute__((address_space(1))) T global_t;
  return (T*)(global_t*)(unsigned long long)p;
}
#endif

template <bool MULTI>
struct multi_dyn_of {
  typedef typename std::conditional<MULTI, MultiDyn, NoMultiDyn>::type type;
};

}  // namespace nidreg
