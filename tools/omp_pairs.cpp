// The reference's MultiNIDCost pattern against the C ABI: k NIDCost handles on one GPU, evaluated at the same pose from an
// OpenMP loop with one thread per pair (visual_camera_calibration.cpp:161), against nidreg_eval_multi over the same
// handles.  Synthetic data (timing only).  Run with and without NIDREG_COMBINE=1.
//   g++ -O2 -fopenmp -I include tools/omp_pairs.cpp -o tools/omp_pairs.bin -L direct_visual_lidar_calibration_amd/csrc -lnidreg -Wl,-rpath,'$ORIGIN/../direct_visual_lidar_calibration_amd/csrc'
#include <omp.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "nidreg.h"

int main(int argc, char** argv) {
  const long total = argc > 1 ? std::atol(argv[1]) : 10000000;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 200;
  const int W = 1920, H = 1080, B = 256;
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  std::vector<double> img(size_t(W) * H);
  for (size_t i = 0; i < img.size(); i++) img[i] = 0.5 + 0.5 * std::sin(0.013 * double(i % W)) * std::cos(0.017 * double(i / W));
  std::printf("{\"combine\": \"%s\"", std::getenv("NIDREG_COMBINE") ? std::getenv("NIDREG_COMBINE") : "");
  for (int k : {1, 2, 4, 8}) {
    const long n = total / k;
    std::vector<nidreg_handle*> hs;
    for (int p = 0; p < k; p++) {
      std::vector<double> pts(size_t(n) * 4), ints(static_cast<size_t>(n));
      for (long i = 0; i < n; i++) {
        const double z = 3.0 + 15.0 * U(rng), u = 20.0 + (W - 40.0) * U(rng), v = 20.0 + (H - 40.0) * U(rng);
        pts[4 * i] = float((u - 960.0) / 1100.0 * z), pts[4 * i + 1] = float((v - 540.0) / 1100.0 * z), pts[4 * i + 2] = float(z), pts[4 * i + 3] = 1.0;
        ints[size_t(i)] = U(rng);
      }
      nidreg_desc d{};
      d.struct_size = sizeof(d);
      d.model_id = NIDREG_MODEL_PLUMB_BOB;
      d.mode = NIDREG_MODE_SPLINE;
      d.bins = B;
      d.intrinsics[0] = d.intrinsics[1] = 1100, d.intrinsics[2] = 960, d.intrinsics[3] = 540;
      d.distortion[0] = -0.04, d.distortion[1] = 0.08, d.distortion[2] = 1e-4, d.distortion[3] = -3e-4, d.distortion[4] = -0.04;
      d.width = W, d.height = H, d.image_dtype = NIDREG_IMAGE_F64, d.image = img.data(), d.image_row_stride = W * 8;
      d.num_points = n, d.points = pts.data(), d.point_stride = 32, d.intensities = ints.data();
      nidreg_handle* h = nullptr;
      if (nidreg_create(&d, &h) != NIDREG_OK) {
        std::printf("create failed: %s\n", nidreg_last_error());
        return 1;
      }
      hs.push_back(h);
    }
    std::vector<double> cost(static_cast<size_t>(k)), grad(size_t(k) * 7);
    auto pose = [&](int r, double* se3) {
      se3[0] = 1e-3 * std::sin(0.1 * r), se3[1] = 1e-3 * std::cos(0.2 * r), se3[2] = 5e-4 * std::sin(0.3 * r), se3[3] = 1.0;
      se3[4] = 0.01 * std::sin(0.05 * r), se3[5] = 0.01 * std::cos(0.07 * r), se3[6] = 0.005;
    };
    double t_omp = 0, t_multi = 0;
    const bool multi_first = std::getenv("OMP_PAIRS_MULTI_FIRST") != nullptr;  // order of the two measurements
    for (int ph = 0; ph < 2; ph++) {
      const int phase = multi_first ? 1 - ph : ph;
      for (int r = -10; r < reps; r++) {
        double se3[7];
        pose(r, se3);
        const auto t0 = std::chrono::steady_clock::now();
        if (phase == 0) {
#pragma omp parallel for num_threads(k) schedule(static, 1)
          for (int p = 0; p < k; p++) nidreg_eval(hs[size_t(p)], se3, &cost[size_t(p)], &grad[size_t(p) * 7]);
        } else {
          double c, g[7];
          nidreg_eval_multi(hs.data(), k, nullptr, se3, &c, g);
        }
        const double dt = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (r >= 0) (phase == 0 ? t_omp : t_multi) += dt;
      }
    }
    std::printf(", \"pairs_%d\": {\"points_per_pair\": %ld, \"omp_us\": %.1f, \"eval_multi_us\": %.1f, \"cost0\": %.12f}", k, n, t_omp / reps, t_multi / reps, cost[0]);
    for (nidreg_handle* h : hs) nidreg_destroy(h);
  }
  std::printf("}\n");
  return 0;
}
