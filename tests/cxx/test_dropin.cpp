// Exercises the C++ drop-in classes (include/vlcal_amd/*.hpp) the way
// src/vlcal/calib/visual_camera_calibration.cpp:141-178,206-215 uses the reference classes:
// a MultiNIDCost-style functor evaluated once with doubles and once with Jet<double,7>.
// Prints "cost g0..g6 cost_double nearest_cost" for the pytest wrapper, which compares with the oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <cmath>

#include "vlcal_amd/cost_calculator_nid.hpp"
#include "vlcal_amd/generate_lidar_image.hpp"
#include "vlcal_amd/nid_cost.hpp"
#include "vlcal_amd/points_color_updater.hpp"
#include "vlcal_amd/view_culling.hpp"

struct MultiNIDCost {  // visual_camera_calibration.cpp:141-178 without the trust gate / OpenMP
  std::vector<std::shared_ptr<vlcal::NIDCost>> costs;
  template <typename T>
  bool operator()(const T* params, T* residual) const {
    std::vector<T> residuals(costs.size());
    bool ok = true;
    for (size_t i = 0; i < costs.size(); i++) ok = (*costs[i])(params, &residuals[i]) && ok;
    for (size_t i = 1; i < costs.size(); i++) {
      residuals[0].a += residuals[i].a;
      for (int k = 0; k < 7; k++) residuals[0].v[k] += residuals[i].v[k];
    }
    *residual = residuals[0];
    return ok;
  }
};

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int W, H, N, bins, nintr, ndist;
  char model[64] = {0};
  double intr[5], dist[8], se3[7], max_fov, T[16];
  if (fread(model, 1, 64, f) != 64) return 4;
  if (fread(&W, 4, 1, f) != 1 || fread(&H, 4, 1, f) != 1 || fread(&N, 4, 1, f) != 1 || fread(&bins, 4, 1, f) != 1 || fread(&nintr, 4, 1, f) != 1 || fread(&ndist, 4, 1, f) != 1) return 4;
  if (fread(intr, 8, 5, f) != 5 || fread(dist, 8, 8, f) != 8 || fread(se3, 8, 7, f) != 7 || fread(&max_fov, 8, 1, f) != 1 || fread(T, 8, 16, f) != 16) return 4;
  cv::Mat img8(H, W, cv::CV_8UC1_), img64(H, W, cv::CV_64FC1_);
  if (fread(img8.data, 1, size_t(W) * H, f) != size_t(W) * H) return 4;
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) img64.at<double>(y, x) = img8.at<unsigned char>(y, x) * (1.0 / 255.0);
  std::vector<Eigen::Vector4d> pts(N);
  std::vector<double> ints(N);
  if (fread(pts.data(), 32, N, f) != size_t(N) || fread(ints.data(), 8, N, f) != size_t(N)) return 4;
  fclose(f);

  auto bad = camera::create_camera("plumb_bob", {1, 2, 3}, {});
  if (bad) return 5;  // intrinsic-count mismatch must give nullptr
  auto proj = camera::create_camera(model, std::vector<double>(intr, intr + nintr), std::vector<double>(dist, dist + ndist));
  if (!proj) return 6;
  auto frame = std::make_shared<vlcal::Frame>();
  frame->num_points = N;
  frame->points = pts.data();
  frame->intensities = ints.data();

  MultiNIDCost multi;
  multi.costs.emplace_back(new vlcal::NIDCost(proj, img64, frame, bins));

  typedef ceres::Jet<double, 7> J;
  J params[7], res;
  for (int k = 0; k < 7; k++) {
    params[k] = J(se3[k]);
    params[k].v[k] = 1.0;
  }
  if (!multi(params, &res)) return 7;
  double c2 = 0.0;
  if (!(*multi.costs[0])(se3, &c2)) return 8;
  // device-resident route (no culling): must give the very same cost
  vlcal::DeviceCloud cloud(frame);
  vlcal::NIDCost from_cloud(proj, img64, cloud, nullptr, 0.0, false, bins);
  double c3 = 0.0;
  if (!from_cloud(se3, &c3) || c3 != c2) return 9;
  // the pair sharded over "two GPUs" inside the library (the device list extension of the constructor; the same device
  // twice is what a 1-GPU box offers): bit-identical cost, gradient equal to summation order
  {
    vlcal::NIDCost sharded(proj, img64, frame, bins, 0, NIDREG_PREC_FP64, std::vector<int>{0, 0});
    if (nidreg_num_shards(sharded.native_handle()) != 2) return 10;
    J res_s;
    if (!sharded(params, &res_s) || res_s.a != res.a) return 11;
    for (int k = 0; k < 7; k++)
      if (std::fabs(res_s.v[k] - res.v[k]) > 1e-12 * (1.0 + std::fabs(res.v[k]))) return 12;
  }

  auto data = std::make_shared<vlcal::VisualLiDARData>(img8, frame);
  vlcal::NIDCostParams np;
  np.bins = bins;
  vlcal::CostCalculatorNID calc(proj, data, np, max_fov);
  Eigen::Isometry3d Tm;
  for (int k = 0; k < 16; k++) Tm.m[k] = T[k];
  const double cn = calc.calculate(Tm);

  // PointsColorUpdater::update and generate_lidar_image through their drop-in headers (checksums)
  vlcal::PointsColorUpdater updater(proj, img8, frame, nullptr, std::cos(max_fov + 0.5 * M_PI / 180.0));
  const std::vector<float>& colors = updater.update(Tm, 0.7);
  double color_sum = 0.0;
  long n_colored = 0;
  for (int i = 0; i < N; i++) {
    for (int k = 0; k < 4; k++) color_sum += colors[size_t(i) * 4 + k];
    n_colored += colors[size_t(i) * 4 + 3] > 0.0f ? 1 : 0;
  }
  std::vector<double> lidar_intensity(size_t(W) * H);
  std::vector<int32_t> lidar_index(size_t(W) * H);
  vlcal::generate_lidar_image(proj, W, H, Tm, frame, std::cos(max_fov), lidar_intensity.data(), lidar_index.data());
  long idx_count = 0;
  double idx_sum = 0.0, inten_sum = 0.0;
  for (size_t q = 0; q < lidar_index.size(); q++) {
    if (lidar_index[q] >= 0) {
      idx_count++;
      idx_sum += lidar_index[q];
    }
    inten_sum += lidar_intensity[q];
  }

  // ViewCulling through its drop-in header (count + index checksum, depth buffer on)
  const vlcal::ViewCulling culling(proj, W, H, vlcal::ViewCullingParams(), std::cos(max_fov));
  const std::vector<int> kept = culling.cull_indices(frame, Tm);
  double kept_sum = 0.0;
  for (int i : kept) kept_sum += i;

  printf("%.17g", res.a);
  for (int k = 0; k < 7; k++) printf(" %.17g", res.v[k]);
  printf(" %.17g %.17g", c2, cn);
  printf(" %.17g %ld %ld %.17g %.17g", color_sum, n_colored, idx_count, idx_sum, inten_sum);
  printf(" %zu %.17g\n", kept.size(), kept_sum);
  return 0;
}
