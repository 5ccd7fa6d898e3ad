// The reference's MultiNIDCost pattern against the C ABI: k NIDCost handles on one GPU, evaluated at the same pose from an
// OpenMP loop with one thread per pair (visual_camera_calibration.cpp:161), against nidreg_eval_multi over the same
// handles -- as one grid per pass over all pairs (the default for 2..16 compatible pairs on one GPU) and, with
// NIDREG_NO_MULTI_GRID=1 in the environment, as per-pair launches.  Data: a scene written by tools/dump_scene_raw.py
// (argv[3]; pair p takes the points i = p mod k, so every pair sees the scene's distribution) or, without it, uniformly
// random points in the image (timing only).
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
  // optional real scene
  std::vector<float> sxyz, sint;
  std::vector<unsigned char> simg;
  double T_true[7] = {0, 0, 0, 1, 0, 0, 0}, sintr[4] = {1100, 1100, 960, 540}, sdist[5] = {-0.04, 0.08, 1e-4, -3e-4, -0.04};
  long scene_n = 0;
  int sW = W, sH = H;
  if (argc > 3) {
    FILE* f = std::fopen(argv[3], "rb");
    if (!f) {
      std::printf("cannot open %s\n", argv[3]);
      return 1;
    }
    long long n64 = 0;
    bool ok = std::fread(&n64, 8, 1, f) == 1 && std::fread(&sW, 4, 1, f) == 1 && std::fread(&sH, 4, 1, f) == 1 && std::fread(T_true, 8, 7, f) == 7 && std::fread(sintr, 8, 4, f) == 4 &&
              std::fread(sdist, 8, 5, f) == 5;
    scene_n = long(n64);
    sxyz.resize(size_t(scene_n) * 3);
    sint.resize(size_t(scene_n));
    simg.resize(size_t(sW) * sH);
    ok = ok && std::fread(sxyz.data(), 4, sxyz.size(), f) == sxyz.size() && std::fread(sint.data(), 4, sint.size(), f) == sint.size() && std::fread(simg.data(), 1, simg.size(), f) == simg.size();
    std::fclose(f);
    if (!ok || sW != W || sH != H) {
      std::printf("bad scene file (need a %dx%d scene)\n", W, H);
      return 1;
    }
    for (size_t i = 0; i < img.size(); i++) img[i] = double(simg[i]) * (1.0 / 255.0);
  }
  std::printf("{\"data\": \"%s\", \"multi_grid\": \"%s\"", scene_n ? "scene" : "uniform random", std::getenv("NIDREG_NO_MULTI_GRID") ? "off (per-pair launches)" : "on");
  std::vector<int> ks = {1, 2, 4, 8};
  if (argc > 4) ks.assign(1, std::atoi(argv[4]));  // one pair count only (kernel traces)
  for (int k : ks) {
    const long n = (scene_n ? std::min(total, scene_n) : total) / k;
    std::vector<nidreg_handle*> hs;
    for (int p = 0; p < k; p++) {
      std::vector<double> pts(size_t(n) * 4), ints(static_cast<size_t>(n));
      for (long i = 0; i < n; i++) {
        if (scene_n) {
          const size_t j = size_t(i) * size_t(k) + size_t(p);
          pts[4 * i] = sxyz[3 * j], pts[4 * i + 1] = sxyz[3 * j + 1], pts[4 * i + 2] = sxyz[3 * j + 2], pts[4 * i + 3] = 1.0;
          ints[size_t(i)] = sint[j];
          continue;
        }
        const double z = 3.0 + 15.0 * U(rng), u = 20.0 + (W - 40.0) * U(rng), v = 20.0 + (H - 40.0) * U(rng);
        pts[4 * i] = float((u - 960.0) / 1100.0 * z), pts[4 * i + 1] = float((v - 540.0) / 1100.0 * z), pts[4 * i + 2] = float(z), pts[4 * i + 3] = 1.0;
        ints[size_t(i)] = U(rng);
      }
      nidreg_desc d{};
      d.struct_size = sizeof(d);
      d.model_id = NIDREG_MODEL_PLUMB_BOB;
      d.mode = NIDREG_MODE_SPLINE;
      d.bins = B;
      for (int q = 0; q < 4; q++) d.intrinsics[q] = sintr[q];
      for (int q = 0; q < 5; q++) d.distortion[q] = sdist[q];
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
      if (scene_n) {  // small perturbations of the scene's extrinsic
        for (int q = 0; q < 7; q++) se3[q] = T_true[q];
        se3[0] += 1e-3 * std::sin(0.1 * r), se3[1] += 1e-3 * std::cos(0.2 * r), se3[4] += 0.01 * std::sin(0.05 * r), se3[5] += 0.01 * std::cos(0.07 * r);
        return;
      }
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
    // what the caller's own pattern costs without any evaluation: one parallel region of k threads whose body spins ~1 us
    // (fork, the threads' arrival skew, join) -- the part of the omp column no library can take out
    double t_region = 0;
    for (int r = -10; r < reps; r++) {
      const auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for num_threads(k) schedule(static, 1)
      for (int p = 0; p < k; p++) {
        const auto s0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - s0).count() < 1.0) {
        }
      }
      const double dt = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (r >= 0) t_region += dt;
    }
    std::printf(", \"pairs_%d\": {\"points_per_pair\": %ld, \"omp_us\": %.1f, \"eval_multi_us\": %.1f, \"empty_omp_region_us\": %.1f, \"cohort\": \"%s\", \"cost0\": %.12f}", k, n, t_omp / reps, t_multi / reps,
                t_region / reps - 1.0, std::getenv("NIDREG_COHORT") ? std::getenv("NIDREG_COHORT") : "", cost[0]);
    for (nidreg_handle* h : hs) nidreg_destroy(h);
  }
  std::printf("}\n");
  return 0;
}
