// Development aid: per-workgroup timeline of k_spline_hist (WIDE) on a synthetic 10M-point workload.
// Builds with -DNID_STAMP, launches the production kernel directly (no library), and prints when each
// workgroup started / ended and on which XCD / CU it ran.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DNID_STAMP -I direct_visual_lidar_calibration_amd/csrc tools/wg_timeline.hip -o /tmp/wg_timeline
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <numeric>
#include <random>
#include <vector>

#define NID_COMMON_KERNELS
#include "nid_kernels.hpp"

using namespace nidreg;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                   \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? std::atoll(argv[1]) : 10000000;
  const int nchunks_target = argc > 2 ? std::atoi(argv[2]) : 512;
  const int W = 1920, H = 1080, B = 256;
  const double fx = 1100, fy = 1100, cx = 960, cy = 540;
  // records: 256 column groups of N/256 points, each sweeping the image in raster order with jitter
  std::vector<Rec32> recs(static_cast<size_t>(N));
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(0.f, 1.f);
  const int64_t per = N / B;
  for (int g = 0; g < B; g++) {
    for (int64_t i = 0; i < per; i++) {
      const float t = (float(i) + U(rng)) / float(per);
      const float v = t * (H - 2) + 0.5f;
      const float u = std::fmod(t * 997.0f, 1.0f) * (W - 2) + 0.5f;
      const float z = 3.0f + 15.0f * U(rng);
      Rec32 r;
      r.x = float((u - cx) / fx) * z;
      r.y = float((v - cy) / fy) * z;
      r.z = z;
      r.bin = uint32_t(g);
      recs[size_t(g) * per + i] = r;
    }
  }
  const int64_t NN = per * B;
  std::vector<Chunk> chunks;
  const int parts = std::max(1, nchunks_target / B);
  const double first_frac = argc > 3 ? std::atof(argv[3]) : 0.0;  // > 0 with 2 parts: share of a group given to its FIRST-dispatched chunk
  if (parts == 2 && first_frac > 0.0) {
    // all "A" chunks first (dispatched first = older on their CU = faster), then all "B" chunks
    const int64_t a = int64_t(per * first_frac) / 64 * 64;
    for (int g = 0; g < B; g++) chunks.push_back(Chunk{uint32_t(g * per), uint32_t(a), uint32_t(g), 0});
    for (int g = 0; g < B; g++) chunks.push_back(Chunk{uint32_t(g * per + a), uint32_t(per - a), uint32_t(g), 0});
  } else {
    for (int g = 0; g < B; g++) {
      const int64_t size = ((per + parts - 1) / parts + 63) / 64 * 64;
      for (int64_t s = 0; s < per; s += size) chunks.push_back(Chunk{uint32_t(g * per + s), uint32_t(std::min<int64_t>(size, per - s)), uint32_t(g), 0});
    }
  }
  const int pitch = ((W + 8) + 3) & ~3;
  const int nstrips = (H + 3 + 3) / 4 + 1;
  std::vector<uint8_t> img(size_t(pitch) * 4 * nstrips + 64);
  for (auto& b : img) b = uint8_t(rng() & 255);

  Rec32* d_recs;
  Chunk* d_chunks;
  uint32_t* d_gend;  // end offsets of the column groups (nid_kernels.hpp Segments; every chunk here lies inside one group)
  std::vector<uint32_t> gend(static_cast<size_t>(B));
  for (int g = 0; g < B; g++) gend[size_t(g)] = uint32_t((g + 1) * per);
  CK(hipMalloc(&d_gend, gend.size() * sizeof(uint32_t)));
  CK(hipMemcpy(d_gend, gend.data(), gend.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  uint8_t* d_img;
  u64* d_hist;
  const size_t hist_words = size_t(B) * B + 8 + 2 * size_t((B + 7) & ~7);  // nidreg_hist_words(B): joint cells, tail, column sums, row sums
  CK(hipMalloc(&d_recs, recs.size() * sizeof(Rec32) + 64));
  CK(hipMalloc(&d_chunks, chunks.size() * sizeof(Chunk)));
  CK(hipMalloc(&d_img, img.size()));
  CK(hipMalloc(&d_hist, hist_words * 8));
  CK(hipMemcpy(d_recs, recs.data(), recs.size() * sizeof(Rec32), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_chunks, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice));

  PoseParams<double> pose{};
  pose.R[0] = pose.R[4] = pose.R[8] = 1.0;
  CamParams<double> cam{};
  cam.intr[0] = fx, cam.intr[1] = fy, cam.intr[2] = cx, cam.intr[3] = cy;
  cam.dist[0] = -0.04, cam.dist[1] = 0.08, cam.dist[2] = 1e-4, cam.dist[3] = -3e-4, cam.dist[4] = -0.04;
  const int frac = 38;
  const double dn = std::ldexp(1.0, frac - 1074);
  auto k = k_spline_hist<MODEL_PLUMB_BOB, Rec32, double, true, false, false>;
  const size_t lds = (size_t(B) * 8 << kWideShift) + 8 + 16;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0;
  for (int it = 0; it < 6; it++) {
    CK(hipMemset(d_hist, 0, hist_words * 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(unsigned(chunks.size())), dim3(kWideThreads), lds, 0, d_recs, d_chunks, d_gend, d_img, pitch, W, H, pose, cam, B, 1, kWideShift, dn, d_hist, 1, static_cast<const ShardTable*>(nullptr), u64(0),
                       static_cast<unsigned int*>(nullptr), static_cast<const MultiEntry*>(nullptr), NoMultiDyn());
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    CK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::vector<unsigned long long> st(4 * chunks.size());
  CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), st.size() * 8));
  std::vector<u64> hist(hist_words);
  CK(hipMemcpy(hist.data(), d_hist, hist_words * 8, hipMemcpyDeviceToHost));
  const size_t n = chunks.size();
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t i = 0; i < n; i++) {
    t0 = std::min(t0, st[4 * i]);
    t1 = std::max(t1, st[4 * i + 1]);
  }
  std::vector<double> start(n), end(n), dur(n);
  std::map<unsigned, int> per_cu;
  for (size_t i = 0; i < n; i++) {
    start[i] = (st[4 * i] - t0) * 0.01;
    end[i] = (st[4 * i + 1] - t0) * 0.01;
    dur[i] = end[i] - start[i];
    const unsigned hw = unsigned(st[4 * i + 2]), xcc = unsigned(st[4 * i + 2] >> 32) & 15u;
    const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
    per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
  }
  auto pct = [](std::vector<double> v, double p) {
    std::sort(v.begin(), v.end());
    return v[size_t(p * (v.size() - 1))];
  };
  std::printf("points %lld chunks %zu inliers %llu  event time %.1f us  span(first start..last end) %.1f us\n", (long long)NN, n, hist[size_t(B) * B], ms * 1e3, (t1 - t0) * 0.01);
  std::printf("start  us: min %.1f p50 %.1f p90 %.1f max %.1f\n", pct(start, 0), pct(start, .5), pct(start, .9), pct(start, 1));
  std::printf("end    us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f\n", pct(end, 0), pct(end, .1), pct(end, .5), pct(end, .9), pct(end, 1));
  std::printf("dur    us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f  mean %.1f\n", pct(dur, 0), pct(dur, .1), pct(dur, .5), pct(dur, .9), pct(dur, 1),
              std::accumulate(dur.begin(), dur.end(), 0.0) / n);
  std::map<int, int> occ;
  for (auto& kv : per_cu) occ[kv.second]++;
  std::printf("distinct CUs used %zu; workgroups per CU histogram:", per_cu.size());
  for (auto& kv : occ) std::printf("  %d WG x %d CUs", kv.first, kv.second);
  std::printf("\n");
  // mean duration per XCD and, within XCD 0, per shader engine / CU: is the spread systematic?
  {
    std::map<unsigned, std::pair<double, int>> by_xcc, by_cu0;
    for (size_t i = 0; i < n; i++) {
      const unsigned hw = unsigned(st[4 * i + 2]), xcc = unsigned(st[4 * i + 2] >> 32) & 15u;
      by_xcc[xcc].first += dur[i];
      by_xcc[xcc].second++;
      if (xcc == 0) {
        by_cu0[(hw >> 8) & 0xffu].first += dur[i];
        by_cu0[(hw >> 8) & 0xffu].second++;
      }
    }
    std::printf("mean dur per XCD:");
    for (auto& kv : by_xcc) std::printf("  x%u: %.1f (%d)", kv.first, kv.second.first / kv.second.second, kv.second.second);
    std::printf("\nXCD 0, mean dur per (se.sh.cu):");
    for (auto& kv : by_cu0) std::printf(" %02x:%.0f", kv.first, kv.second.first / kv.second.second);
    std::printf("\n");
    // by column group (image / record position): first vs second half of the chunk list
    double a = 0, b = 0;
    for (size_t i = 0; i < n; i++) (i < n / 2 ? a : b) += dur[i];
    std::printf("mean dur first half of the chunk list %.1f, second half %.1f\n", a / (n / 2), b / (n - n / 2));
  }
  {
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return end[a] > end[b]; });
    std::printf("slowest:");
    for (int k = 0; k < 8; k++) {
      const size_t i = order[k];
      std::printf("  [wg %zu x%u cu %02x end %.1f]", i, unsigned(st[4 * i + 2] >> 32) & 15u, (unsigned(st[4 * i + 2]) >> 8) & 0xffu, end[i]);
    }
    std::printf("\nmean dur by 64-chunk block:");
    for (size_t b0 = 0; b0 < n; b0 += 64) {
      double a = 0;
      for (size_t i = b0; i < std::min(n, b0 + 64); i++) a += dur[i];
      std::printf(" %.0f", a / std::min<size_t>(64, n - b0));
    }
    std::printf("\n");
  }
  // workgroups that started late (after the first one ended) = second round
  const double first_end = pct(end, 0);
  int late = 0;
  for (size_t i = 0; i < n; i++) late += start[i] > first_end ? 1 : 0;
  std::printf("workgroups started after the first one ended: %d\n", late);
  return 0;
}
