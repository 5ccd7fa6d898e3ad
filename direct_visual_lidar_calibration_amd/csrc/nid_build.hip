// nid_build.hip -- device-side construction of a handle's point records from a device-resident cloud:
//   [view culling (nid_cull_kernels.hpp)] -> histogram column + Morton key per surviving point ->
//   rocPRIM radix sort by (column group, Morton code) -> gather into Rec32 / Rec64.
// This is the host bucketing of nidreg_create (nidreg_plan.hip) moved onto the GPU, so that the reference's
// per-outer-iteration sequence  ViewCulling::cull -> new NIDCost  (visual_camera_calibration.cpp:201-206)
// never round-trips the cloud through the host.  The record ORDER may differ from the host path (device
// vs host atan2 rounding in the Morton key); the histogram does not (fixed point, order independent).
// Built with -ffp-contract=off (the culling arithmetic must match the CPU bit for bit).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <vector>

#include "nid_device.hpp"
#include "nid_launch.hpp"

namespace nidreg {

namespace {

__device__ __forceinline__ int cast_int_dev(double d) {  // x86 cvttsd2si semantics: NaN / overflow -> INT_MIN
  if (!(d > -2147483649.0 && d < 2147483648.0)) return int(0x80000000u);
  return int(d);
}

// bin_points = max(0, min(bins - 1, int(intensity * bins))) (nid_cost.hpp:49, cost_calculator_nid.cpp:47) with the CALLER's bin
// count; `lut` (nullable: bins > 256, nidreg_plan.hip WideBins) then maps the occupied bins onto the compact range the kernels use
__device__ __forceinline__ int point_bin(double intensity, int Bsrc, const uint16_t* __restrict__ lut) {
  const int b = max(0, min(Bsrc - 1, cast_int_dev(intensity * double(Bsrc))));
  return lut ? int(lut[b]) : b;
}

__global__ __launch_bounds__(256) void k_mark_bins(const double* __restrict__ v, long long n, int B, unsigned char* __restrict__ used) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  used[max(0, min(B - 1, cast_int_dev(v[i] * double(B))))] = 1;
}

__global__ __launch_bounds__(256) void k_build_keys(
  const double* __restrict__ pts, const double* __restrict__ intensities, const unsigned char* __restrict__ keep, long long n, int Bsrc, const uint16_t* __restrict__ lut, int GW, int input_order,
  unsigned long long* __restrict__ keys, unsigned int* __restrict__ idx, int* __restrict__ not_lossless) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  idx[i] = (unsigned int)i;
  if (keep && !keep[i]) {
    keys[i] = ~0ull;
    return;
  }
  const double x = pts[4 * i], y = pts[4 * i + 1], z = pts[4 * i + 2];
  if ((double(float(x)) != x && x == x) || (double(float(y)) != y && y == y) || (double(float(z)) != z && z == z)) *not_lossless = 1;
  const int b = point_bin(intensities[i], Bsrc, lut);
  if (input_order) {  // NIDREG_FLAG_INPUT_ORDER: the caller's order inside each column group
    keys[i] = ((unsigned long long)(unsigned int)(b / GW) << 41) | (unsigned long long)(unsigned int)i;
    return;
  }
  const double az = atan2(y, x);
  const double el = atan2(z, sqrt(x * x + y * y));
  unsigned int qa = (unsigned int)fmin(65535.0, fmax(0.0, (az + 3.14159265358979323846) * (65535.0 / (2.0 * 3.14159265358979323846))));
  unsigned int qe = (unsigned int)fmin(65535.0, fmax(0.0, (el + 0.5 * 3.14159265358979323846) * (65535.0 / 3.14159265358979323846)));
  if (!(az == az) || !(el == el)) qa = qe = 0;
  unsigned int m = 0;
#pragma unroll
  for (int bb = 0; bb < 16; bb++) m |= (((qa >> bb) & 1u) << (2 * bb)) | (((qe >> bb) & 1u) << (2 * bb + 1));
  // [column group : 23][column : 9][Morton code : 32]
  keys[i] = ((unsigned long long)(unsigned int)(b / GW) << 41) | ((unsigned long long)(unsigned int)b << 32) | m;
}

// first[g] = first sorted position whose column group is g (g == NG: the removed points)
__global__ __launch_bounds__(256) void k_build_bounds(const unsigned long long* __restrict__ keys, long long n, int NG, int* __restrict__ first) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  const int g = k == ~0ull ? NG : int(k >> 41);
  if (i == 0) {
    first[g] = 0;
  } else {
    const unsigned long long kp = keys[i - 1];
    const int gp = kp == ~0ull ? NG : int(kp >> 41);
    if (gp != g) first[g] = int(i);
  }
}

template <typename Rec>
__global__ __launch_bounds__(256) void k_build_gather(const double* __restrict__ pts, const double* __restrict__ intensities, const unsigned int* __restrict__ idx, long long kept, int Bsrc,
                                                      const uint16_t* __restrict__ lut, Rec* __restrict__ recs) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kept) return;
  const unsigned int s = idx[i];
  Rec r;
  r.x = decltype(r.x)(pts[4 * (long long)s]);
  r.y = decltype(r.y)(pts[4 * (long long)s + 1]);
  r.z = decltype(r.z)(pts[4 * (long long)s + 2]);
  r.bin = decltype(r.bin)(point_bin(intensities[s], Bsrc, lut));  // as in k_build_keys
  recs[i] = r;
}

}  // namespace

#define BUILD_TRY(expr)           \
  do {                            \
    e = (expr);                   \
    if (e != hipSuccess) goto done; \
  } while (0)

// ---- scratch arena ------------------------------------------------------------------------
ScratchArena& ScratchArena::of(int device) {
  static ScratchArena arenas[64];
  return arenas[device >= 0 && device < 64 ? device : 0];
}
hipError_t ScratchArena::reserve(size_t bytes) {
  cur_ = 0;
  if (bytes <= cap_) return hipSuccess;
  if (base_) (void)hipFree(base_);
  base_ = nullptr;
  cap_ = 0;
  const size_t want = bytes + bytes / 8 + (1u << 20);  // head-room: the next outer iteration's cloud is about the same size
  hipError_t e = hipMalloc(&base_, want);
  if (e != hipSuccess) return e;
  cap_ = want;
  return hipSuccess;
}
void* ScratchArena::carve(size_t bytes) {
  const size_t at = (cur_ + 255) & ~size_t(255);
  if (at + bytes > cap_) return nullptr;
  cur_ = at + bytes;
  return static_cast<char*>(base_) + at;
}
void ScratchArena::release() {
  if (base_) (void)hipFree(base_);
  base_ = nullptr;
  cap_ = cur_ = 0;
}

namespace {
size_t sort_scratch_bytes(long long n) {
  size_t tmp_bytes = 0;
  unsigned long long* k = nullptr;
  unsigned int* v = nullptr;
  if (n > 0) (void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, k, k, v, v, size_t(n), 0, 64, hipStream_t(nullptr));
  return tmp_bytes;
}
inline size_t al(size_t b) { return (b + 255) & ~size_t(255); }
}  // namespace

size_t build_scratch_bytes(long long n, bool cull, int W, int H) {
  if (n <= 0) return 4096;
  size_t b = 0;
  if (cull) b += al(size_t(n)) + al(size_t(n) * 4) + al(size_t(W) * H * 4);
  b += 2 * al(size_t(n) * 8) + 2 * al(size_t(n) * 4) + al(4096) + al(sort_scratch_bytes(n));
  return b + 4096;
}

hipError_t build_records_device(
  const double* d_pts, const double* d_intensities, long long n, const CullArgs* cull, int Bsrc, const uint16_t* d_lut, int GW, int NG, bool force_rec32, bool input_order, ScratchArena& arena,
  void** d_recs_out, int* rec64_out, std::vector<int64_t>& gcount, hipStream_t stream) {
  hipError_t e = hipSuccess;
  unsigned char* d_keep = nullptr;
  int* d_pix = nullptr;
  unsigned int* d_zbuf = nullptr;
  unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
  unsigned int *d_idx = nullptr, *d_idx2 = nullptr;
  int* d_first = nullptr;  // NG + 1 firsts, then the not-lossless flag
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  void* d_recs = nullptr;
  std::vector<int> first(size_t(NG) + 2, -1);
  const unsigned grid = unsigned((n + 255) / 256);
  long long kept = 0;
  int rec64 = 0;
  gcount.assign(size_t(NG) + 1, 0);
  *d_recs_out = nullptr;
  *rec64_out = 0;
  if (size_t(NG) + 2 > 1024) return hipErrorInvalidValue;  // d_first is one 4 KB carve

#define CARVE(ptr, type, bytes)                              \
  do {                                                       \
    ptr = static_cast<type>(arena.carve(bytes));             \
    if (!ptr) {                                              \
      e = hipErrorOutOfMemory;                               \
      goto done;                                             \
    }                                                        \
  } while (0)

  if (n > 0) {
    if (cull) {
      CARVE(d_keep, unsigned char*, size_t(n));
      CARVE(d_pix, int*, size_t(n) * sizeof(int));
      CARVE(d_zbuf, unsigned int*, size_t(cull->W) * cull->H * sizeof(unsigned int));
      BUILD_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(d_zbuf), 0x7f800000, size_t(cull->W) * cull->H, stream));
      BUILD_TRY(launch_cull(cull->model, cull->intr, cull->dist, d_pts, 4, n, cull->T, cull->W, cull->H, cull->min_z, cull->depth, d_pix, d_zbuf, d_keep, stream));
    }
    CARVE(d_keys, unsigned long long*, size_t(n) * 8);
    CARVE(d_keys2, unsigned long long*, size_t(n) * 8);
    CARVE(d_idx, unsigned int*, size_t(n) * 4);
    CARVE(d_idx2, unsigned int*, size_t(n) * 4);
    CARVE(d_first, int*, 4096);
    BUILD_TRY(hipMemsetAsync(d_first, 0xff, (size_t(NG) + 1) * sizeof(int), stream));
    BUILD_TRY(hipMemsetAsync(d_first + NG + 1, 0, sizeof(int), stream));
    hipLaunchKernelGGL(k_build_keys, dim3(grid), dim3(256), 0, stream, d_pts, d_intensities, d_keep, n, Bsrc, d_lut, GW, input_order ? 1 : 0, d_keys, d_idx, d_first + NG + 1);
    BUILD_TRY(hipGetLastError());
    // key = [group : 23][column : 9][Morton code : 32], or [group : 23][input index : 41] to keep the caller's order inside
    // a group; removed points carry all-ones and sort last
    // Only the key bits that can differ are sorted: 41 low bits, the group's bits, and ONE bit above them -- 0 in every real key, 1 in
    // the all-ones key of a removed point, which therefore still sorts last.  For 256 column groups that is 50 of 64 bits: seven
    // radix passes instead of eight (96 us each at 10M points, profiles/r06v_build_kernel_stats.csv); the order is the same.
    int gbits = 1;
    while ((1 << gbits) < NG) gbits++;
    const unsigned end_bit = unsigned(std::min(64, 41 + gbits + 1));
    BUILD_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_keys2, d_idx, d_idx2, size_t(n), 0, end_bit, stream));
    CARVE(d_tmp, void*, tmp_bytes > 0 ? tmp_bytes : 256);
    BUILD_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_idx, d_idx2, size_t(n), 0, end_bit, stream));
    hipLaunchKernelGGL(k_build_bounds, dim3(grid), dim3(256), 0, stream, d_keys2, n, NG, d_first);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipMemcpyAsync(first.data(), d_first, (size_t(NG) + 2) * sizeof(int), hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    kept = first[size_t(NG)] >= 0 ? first[size_t(NG)] : n;
    {
      long long next = kept;
      for (int g = NG - 1; g >= 0; g--) {
        if (first[size_t(g)] < 0) first[size_t(g)] = int(next);
        next = first[size_t(g)];
      }
      for (int g = 0; g < NG; g++) gcount[size_t(g)] = first[size_t(g)];
      gcount[size_t(NG)] = kept;
    }
    rec64 = (!force_rec32 && first[size_t(NG) + 1] != 0) ? 1 : 0;
  }
  {
    const size_t rec_bytes = rec64 ? sizeof(Rec64) : sizeof(Rec32);
    BUILD_TRY(hipMalloc(&d_recs, size_t(kept > 0 ? kept : 1) * rec_bytes + 64));
    if (kept > 0) {
      const unsigned g2 = unsigned((kept + 255) / 256);
      if (rec64)
        hipLaunchKernelGGL(k_build_gather<Rec64>, dim3(g2), dim3(256), 0, stream, d_pts, d_intensities, d_idx2, kept, Bsrc, d_lut, static_cast<Rec64*>(d_recs));
      else
        hipLaunchKernelGGL(k_build_gather<Rec32>, dim3(g2), dim3(256), 0, stream, d_pts, d_intensities, d_idx2, kept, Bsrc, d_lut, static_cast<Rec32*>(d_recs));
      BUILD_TRY(hipGetLastError());
      BUILD_TRY(hipStreamSynchronize(stream));
    }
  }
  *d_recs_out = d_recs;
  d_recs = nullptr;
  *rec64_out = rec64;
done:
#undef CARVE
  if (d_recs) (void)hipFree(d_recs);
  return e;
}

// ------------------------------------------------------------------------------------------
// bin image: bin_image = min(int(pix * bins), bins - 1) with a lower clamp at 0 where the reference would index
// out of bounds (nid_cost.hpp:78-79) for CV_64FC1 input; max(0, min(bins - 1, int(u8 / 255.0 * bins)))
// (cost_calculator_nid.cpp:43-46) for CV_8UC1 input -- the same IEEE double operations as the host code they
// replace.  One thread per padded pixel, written in the strip-tiled layout of load_patch.
namespace {
__global__ __launch_bounds__(256) void k_build_bin_image(const unsigned char* __restrict__ src, int is_f64, long long row_stride, int W, int H, int B, const uint16_t* __restrict__ lut, int pitch, int rows,
                                                         uint8_t* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)pitch * rows) return;
  const int py = int(i / pitch), px = int(i % pitch);
  const int sy = min(max(py - 1, 0), H - 1), sx = min(max(px - 1, 0), W - 1);
  int b;
  if (is_f64) {
    const double v = *reinterpret_cast<const double*>(src + size_t(sy) * size_t(row_stride) + size_t(sx) * 8);
    b = max(0, min(cast_int_dev(v * double(B)), B - 1));
  } else {
    const double v = double(src[size_t(sy) * size_t(row_stride) + size_t(sx)]) / 255.0;
    b = max(0, min(B - 1, cast_int_dev(v * double(B))));
  }
  if (lut) b = int(lut[b]);  // bins > 256: the occupied bins, compacted (nidreg_plan.hip WideBins)
  dst[size_t(py >> 2) * size_t(pitch) * 4 + size_t(px) * 4 + size_t(py & 3)] = uint8_t(b);
}
}  // namespace

hipError_t build_bin_image_device(const void* d_src, int is_f64, long long row_stride, int W, int H, int B, const uint16_t* d_lut, int pitch, int nstrips, uint8_t* d_img, hipStream_t stream) {
  const long long total = (long long)pitch * nstrips * 4;
  hipLaunchKernelGGL(k_build_bin_image, dim3(unsigned((total + 255) / 256)), dim3(256), 0, stream, static_cast<const unsigned char*>(d_src), is_f64, row_stride, W, H, B, d_lut, pitch, nstrips * 4, d_img);
  return hipGetLastError();
}

// which of the B bins the n device-resident values v occupy (bins > 256 on a device-resident cloud: nidreg_plan.hip resolve_wide_bins)
hipError_t mark_bins_device(const double* d_v, long long n, int B, unsigned char* used_host) {
  unsigned char* d_used = nullptr;
  hipError_t e = hipMalloc(&d_used, size_t(B));
  if (e != hipSuccess) return e;
  e = hipMemset(d_used, 0, size_t(B));
  if (e == hipSuccess && n > 0) {
    hipLaunchKernelGGL(k_mark_bins, dim3(unsigned((n + 255) / 256)), dim3(256), 0, hipStream_t(nullptr), d_v, n, B, d_used);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(used_host, d_used, size_t(B), hipMemcpyDeviceToHost);
  (void)hipFree(d_used);
  return e;
}

// ------------------------------------------------------------------------------------------
// Intensity rank equalisation (src/vlcal/preprocess/preprocess.cpp:464-473, preprocess_map.cpp): sort the
// indices by intensity, then intensity[indices[i]] = floor(256 * double(i) / n) / 256.  The reference's
// std::sort is unstable, so the rank order inside a group of EQUAL intensities is unspecified there;
// the radix sort is stable (ties keep index order), one of the orders the reference can produce.
namespace {
__global__ __launch_bounds__(256) void k_eq_keys(const double* __restrict__ v, long long n, unsigned long long* __restrict__ keys, unsigned int* __restrict__ idx) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long b = (unsigned long long)__double_as_longlong(v[i] + 0.0);  // -0 -> +0: they compare equal
  keys[i] = (b >> 63) ? ~b : (b | 0x8000000000000000ull);                              // total order of the doubles
  idx[i] = (unsigned int)i;
}
__global__ __launch_bounds__(256) void k_eq_scatter(const unsigned int* __restrict__ idx_sorted, long long n, double* __restrict__ v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int bins = 256;
  v[idx_sorted[i]] = floor(bins * double(i) / double(n)) / bins;
}
}  // namespace

hipError_t equalize_intensities_device(double* d_intensities, long long n, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipError_t e = hipSuccess;
  unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
  unsigned int *d_idx = nullptr, *d_idx2 = nullptr;
  void* d_tmp = nullptr;
  size_t tmp_bytes = 0;
  const unsigned grid = unsigned((n + 255) / 256);
  BUILD_TRY(hipMalloc(&d_keys, size_t(n) * 8));
  BUILD_TRY(hipMalloc(&d_keys2, size_t(n) * 8));
  BUILD_TRY(hipMalloc(&d_idx, size_t(n) * 4));
  BUILD_TRY(hipMalloc(&d_idx2, size_t(n) * 4));
  hipLaunchKernelGGL(k_eq_keys, dim3(grid), dim3(256), 0, stream, d_intensities, n, d_keys, d_idx);
  BUILD_TRY(hipGetLastError());
  BUILD_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_keys, d_keys2, d_idx, d_idx2, size_t(n), 0, 64, stream));
  BUILD_TRY(hipMalloc(&d_tmp, tmp_bytes));
  BUILD_TRY(rocprim::radix_sort_pairs(d_tmp, tmp_bytes, d_keys, d_keys2, d_idx, d_idx2, size_t(n), 0, 64, stream));
  hipLaunchKernelGGL(k_eq_scatter, dim3(grid), dim3(256), 0, stream, d_idx2, n, d_intensities);
  BUILD_TRY(hipGetLastError());
  BUILD_TRY(hipStreamSynchronize(stream));
done:
  if (d_keys) (void)hipFree(d_keys);
  if (d_keys2) (void)hipFree(d_keys2);
  if (d_idx) (void)hipFree(d_idx);
  if (d_idx2) (void)hipFree(d_idx2);
  if (d_tmp) (void)hipFree(d_tmp);
  return e;
}

}  // namespace nidreg
