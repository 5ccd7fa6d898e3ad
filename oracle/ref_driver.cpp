// TEST INFRASTRUCTURE ONLY -- C ABI around the REFERENCE's own hot-path sources, compiled unmodified and in
// place from /root/reference against the stand-in headers in oracle/shim/ (Eigen, Sophus, ceres::Jet,
// OpenCV, Boost are not installed in this image; see oracle/shim/Eigen/Core for what that does and does
// not pin).  Built by `make -C oracle ref` into oracle/_ref/libref.so (git-ignored; never shipped, never
// imported by the product).  tests/test_reference_build.py compares oracle_* (the hand-written
// restatement, nid_oracle.cpp) with ref_* (this file) on seeded inputs.
//
// Reference translation units linked in: src/camera/create_camera.cpp, src/vlcal/calib/cost_calculator_nid.cpp,
// src/vlcal/calib/view_culling.cpp, src/vlcal/preprocess/generate_lidar_image.cpp,
// src/vlcal/common/points_color_updater.cpp (glk / guik = stand-ins of the viewer side),
// src/vlcal/calib/visual_camera_calibration.cpp (outer loop, Nelder-Mead inner solve, MultiNIDCost; its BFGS solve
// needs Ceres: ceres::Solve is an evaluate-and-record stand-in, gtsam::Pose3::Expmap / Sophus::SE3d restated); header-only:
// include/camera/*.hpp, include/vlcal/costs/nid_cost.hpp, include/dfo/nelder_mead.hpp.
// src/vlcal/common/estimate_fov.cpp is compiled too (estimate_direction / estimate_camera_fov, :17-51, are what the
// path uses; its estimate_lidar_fov needs PCL, whose stand-ins only make the file compile).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include <ceres/ceres.h>
#include <ceres/jet.h>

#include <camera/create_camera.hpp>
#include <dfo/nelder_mead.hpp>
#include <vlcal/calib/cost_calculator_nid.hpp>
#include <vlcal/calib/view_culling.hpp>
#include <vlcal/calib/visual_camera_calibration.hpp>
#include <vlcal/common/estimate_fov.hpp>
#include <vlcal/common/points_color_updater.hpp>
#include <vlcal/costs/nid_cost.hpp>
#include <vlcal/preprocess/generate_lidar_image.hpp>

// ceres::Solve stand-in (oracle/shim/ceres/ceres.h): evaluate, record, leave the parameters alone
namespace ceres {
ProbeLog& probe_log() {
  static ProbeLog log;
  return log;
}
void Solve(const GradientProblemSolver::Options&, const GradientProblem& problem, double* parameters, GradientProblemSolver::Summary* summary) {
  ProbeLog& log = probe_log();
  const int n = problem.function->NumParameters();
  std::vector<std::vector<double>> points;
  points.emplace_back(parameters, parameters + n);
  for (const auto& p : log.probes) points.push_back(p);
  for (const auto& x : points) {
    ProbeLog::Entry e;
    e.cost_value = e.cost_grad = std::numeric_limits<double>::quiet_NaN();
    e.grad.assign(n, std::numeric_limits<double>::quiet_NaN());
    e.ok_value = problem.function->Evaluate(x.data(), &e.cost_value, nullptr);  // T = double instantiation
    e.ok_grad = problem.function->Evaluate(x.data(), &e.cost_grad, e.grad.data());  // T = Jet<double, 7>
    log.entries.push_back(e);
  }
  if (summary) summary->final_cost = log.entries.empty() ? 0.0 : log.entries.front().cost_grad;
}
}  // namespace ceres

// Iridescence's TURBO table is not available; the constructor's call (points_color_updater.cpp:34) gets a
// placeholder, and the driver overwrites the public `intensity_colors` with the caller's colours
Eigen::Vector4f glk::colormapf(glk::COLORMAP, float x) { return Eigen::Vector4f(x, x, x, 1.0f); }

namespace vlcal {

// visual_lidar_data.cpp:29 (that file also holds the PNG / PLY loading constructor: not compiled)
VisualLiDARData::~VisualLiDARData() {}

}  // namespace vlcal

namespace {

camera::GenericCameraBase::ConstPtr make_camera(const char* model, const double* intr, int n_intr, const double* dist, int n_dist) {
  return camera::create_camera(std::string(model), std::vector<double>(intr, intr + n_intr), std::vector<double>(dist, dist + n_dist));
}

std::shared_ptr<vlcal::FrameCPU> make_frame(const double* points, const double* intensities, int64_t n) {
  return std::make_shared<vlcal::FrameCPU>(points, intensities, static_cast<size_t>(n));
}

template <int N>
int run_nelder_mead(double init_step, double conv, int max_iter, double (*fn)(const double*, void*), void* user, const double* x0, double* x_out, double* y_out, int* iters_out) {
  typename dfo::NelderMead<N>::Params params;
  params.init_step = init_step;
  params.convergence_var_thresh = conv;
  params.max_iterations = max_iter;
  dfo::NelderMead<N> optimizer(params);
  Eigen::Matrix<double, N, 1> start;
  for (int i = 0; i < N; i++) start[i] = x0[i];
  const auto f = [&](const Eigen::Matrix<double, N, 1>& x) { return fn(x.data(), user); };
  const auto result = optimizer.optimize(f, start);
  for (int i = 0; i < N; i++) x_out[i] = result.x[i];
  *y_out = result.y;
  *iters_out = result.num_iterations;
  return result.converged ? 1 : 0;
}

}  // namespace

extern "C" {

int ref_project(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, const double* p3, int64_t n, double* uv_out) {
  auto cam = make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  for (int64_t i = 0; i < n; i++) {
    const Eigen::Vector2d uv = cam->project(Eigen::Vector3d(p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]));
    uv_out[2 * i] = uv[0];
    uv_out[2 * i + 1] = uv[1];
  }
  return 0;
}

// value + d(u,v)/d(x,y,z) through the Jet overload (generic_camera_base.hpp:40)
int ref_project_jacobian(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, const double* p3, int64_t n, double* uv_out, double* jac_out) {
  auto cam = make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  typedef ceres::Jet<double, 7> J;
  for (int64_t i = 0; i < n; i++) {
    Eigen::Matrix<J, 3, 1> p;
    for (int k = 0; k < 3; k++) p[k] = J(p3[3 * i + k], k);
    const Eigen::Matrix<J, 2, 1> uv = (*cam)(p);
    for (int r = 0; r < 2; r++) {
      uv_out[2 * i + r] = uv[r].a;
      for (int k = 0; k < 3; k++) jac_out[6 * i + 3 * r + k] = uv[r].v[k];
    }
  }
  return 0;
}

// vlcal::NIDCost::operator()<double | Jet<double,7>>.  Returns 1 = true, 0 = false (non-finite NID), -1 = bad camera.
int ref_nid_cost(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, const double* image_f64, int rows, int cols, const double* points,
                 const double* intensities, int64_t n, int bins, const double* se3, int want_grad, double* cost_out, double* grad_out) {
  auto cam = make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  const cv::Mat image(rows, cols, CV_64FC1, const_cast<double*>(image_f64));
  const vlcal::NIDCost cost(cam, image, make_frame(points, intensities, n), bins);
  if (want_grad) {
    typedef ceres::Jet<double, 7> J;
    J params[7], residual;
    for (int k = 0; k < 7; k++) params[k] = J(se3[k], k);
    if (!cost(params, &residual)) return 0;
    *cost_out = residual.a;
    for (int k = 0; k < 7; k++) grad_out[k] = residual.v[k];
    return 1;
  }
  double residual = 0.0;
  if (!cost(se3, &residual)) return 0;
  *cost_out = residual;
  return 1;
}

int ref_estimate_camera_fov(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, int width, int height, double* max_fov) {
  auto cam = make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  Eigen::Vector2i size;
  size[0] = width;
  size[1] = height;
  *max_fov = vlcal::estimate_camera_fov(cam, size);
  return 0;
}

// vlcal::CostCalculatorNID::calculate; T: row-major 4x4 T_camera_lidar
int ref_cost_calculator_nid(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, const uint8_t* image, int rows, int cols, const double* points,
                            const double* intensities, int64_t n, int bins, const double* T, double* cost_out) {
  auto cam = make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  const cv::Mat img(rows, cols, CV_8UC1, const_cast<uint8_t*>(image));
  auto data = std::make_shared<vlcal::VisualLiDARData>(img, make_frame(points, intensities, n));
  vlcal::NIDCostParams params;
  params.bins = bins;
  vlcal::CostCalculatorNID calc(cam, data, params);
  *cost_out = calc.calculate(Eigen::Isometry3d::FromRowMajor(T));
  return 0;
}

// vlcal::ViewCulling::cull; returns the number of surviving points, their indices in indices_out
int64_t ref_view_culling(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, int width, int height, int enable_depth_buffer_culling,
                         const double* points, int64_t n, const double* T, int* indices_out) {
  auto cam = make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  std::vector<double> zeros(static_cast<size_t>(n), 0.0);
  Eigen::Vector2i size;
  size[0] = width;
  size[1] = height;
  vlcal::ViewCullingParams params;
  params.enable_depth_buffer_culling = enable_depth_buffer_culling != 0;
  const vlcal::ViewCulling culling(cam, size, params);
  const auto culled = culling.cull(make_frame(points, zeros.data(), n), Eigen::Isometry3d::FromRowMajor(T));
  for (size_t i = 0; i < culled->indices.size(); i++) indices_out[i] = culled->indices[i];
  return static_cast<int64_t>(culled->indices.size());
}

// vlcal::generate_lidar_image
int ref_generate_lidar_image(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, int width, int height, const double* points,
                             const double* intensities, int64_t n, const double* T, double* intensity_image, int32_t* index_image) {
  auto cam = make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  Eigen::Vector2i size;
  size[0] = width;
  size[1] = height;
  const auto images = vlcal::generate_lidar_image(cam, size, Eigen::Isometry3d::FromRowMajor(T), make_frame(points, intensities, n));
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) {
      intensity_image[static_cast<size_t>(y) * width + x] = images.first.at<double>(y, x);
      index_image[static_cast<size_t>(y) * width + x] = images.second.at<std::int32_t>(y, x);
    }
  return 0;
}

// vlcal::PointsColorUpdater::update (constructor :26-35 incl. its min_nz; colours read back from the stand-in
// PointCloudBuffer).  intensity_colors: n x 4 floats.  colors_out: n x 4 floats.
int ref_points_color_update(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, const uint8_t* image, int rows, int cols, const double* points, int64_t n,
                            const float* intensity_colors, const double* T, double blend_weight, float* colors_out, double* min_nz_out) {
  auto cam = make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  const cv::Mat img(rows, cols, CV_8UC1, const_cast<uint8_t*>(image));
  std::vector<double> zeros(static_cast<size_t>(n), 0.0);
  vlcal::PointsColorUpdater updater(cam, img, make_frame(points, zeros.data(), n));
  for (int64_t i = 0; i < n; i++)
    for (int k = 0; k < 4; k++) updater.intensity_colors[static_cast<size_t>(i)][k] = intensity_colors[4 * i + k];
  if (min_nz_out) *min_nz_out = updater.min_nz;
  updater.update(Eigen::Isometry3d::FromRowMajor(T), blend_weight);
  const auto& colors = updater.cloud_buffer->last_colors;
  if (static_cast<int64_t>(colors.size()) != n) return -2;
  for (int64_t i = 0; i < n; i++)
    for (int k = 0; k < 4; k++) colors_out[4 * i + k] = colors[static_cast<size_t>(i)][k];
  return 0;
}

namespace {
// the dataset of a VisualCameraCalibration from flat arrays: n_pairs images (rows x cols, 8 bit) and clouds
std::vector<vlcal::VisualLiDARData::ConstPtr> make_dataset(int n_pairs, const uint8_t* const* images, int rows, int cols, const double* const* points, const double* const* intensities,
                                                           const int64_t* num_points) {
  std::vector<vlcal::VisualLiDARData::ConstPtr> dataset;
  for (int i = 0; i < n_pairs; i++) {
    const cv::Mat img(rows, cols, CV_8UC1, const_cast<uint8_t*>(images[i]));
    dataset.emplace_back(std::make_shared<vlcal::VisualLiDARData>(img.clone(), make_frame(points[i], intensities[i], num_points[i])));
  }
  return dataset;
}
}  // namespace

// vlcal::VisualCameraCalibration::calibrate with registration_type NID_NELDER_MEAD: the reference's own outer loop
// (visual_camera_calibration.cpp:35-68) around its Nelder-Mead inner solve (:70-139).  T_in / T_out: row-major 4x4
// T_camera_lidar.  Returns the number of callback invocations (new best cost found).
int ref_calibrate_nelder_mead(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, int n_pairs, const uint8_t* const* images, int rows, int cols,
                              const double* const* points, const double* const* intensities, const int64_t* num_points, int bins, int max_outer_iterations, int max_inner_iterations,
                              double init_step, double convergence, int disable_culling, const double* T_in, double* T_out) {
  auto cam = make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  int callbacks = 0;
  vlcal::VisualCameraCalibrationParams params;
  params.registration_type = vlcal::RegistrationType::NID_NELDER_MEAD;
  params.nid_bins = bins;
  params.max_outer_iterations = max_outer_iterations;
  params.max_inner_iterations = max_inner_iterations;
  params.nelder_mead_init_step = init_step;
  params.nelder_mead_convergence_criteria = convergence;
  params.disable_z_buffer_culling = disable_culling != 0;
  params.callback = [&](const Eigen::Isometry3d&) { callbacks++; };
  vlcal::VisualCameraCalibration calib(cam, make_dataset(n_pairs, images, rows, cols, points, intensities, num_points), params);
  const Eigen::Isometry3d T = calib.calibrate(Eigen::Isometry3d::FromRowMajor(T_in));
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) T_out[i * 4 + j] = T(i, j);
  return callbacks;
}

// The reference's MultiNIDCost functor (a struct private to visual_camera_calibration.cpp:141-178), reached through
// estimate_pose_bfgs: view culling at T_init, NIDCost per pair, MultiNIDCost(init) wrapped in
// AutoDiffFirstOrderFunction<MultiNIDCost, 7>, handed to ceres::Solve -- which in this build only evaluates at the
// start and at the `n_probes` probe parameter vectors ([qx qy qz qw tx ty tz], 7 doubles each) and records.
// Outputs per evaluated point (1 + n_probes): ok_value, ok_grad (0/1), cost_value, cost_grad, grad[7].
int ref_multi_nid_cost_probes(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, int n_pairs, const uint8_t* const* images, int rows, int cols,
                              const double* const* points, const double* const* intensities, const int64_t* num_points, int bins, int disable_culling, const double* T_init,
                              const double* probes, int n_probes, int* ok_value, int* ok_grad, double* cost_value, double* cost_grad, double* grads, double* start_params) {
  auto cam = make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  ceres::ProbeLog& log = ceres::probe_log();
  log.probes.clear();
  log.entries.clear();
  for (int k = 0; k < n_probes; k++) log.probes.emplace_back(probes + 7 * k, probes + 7 * k + 7);
  vlcal::VisualCameraCalibrationParams params;
  params.registration_type = vlcal::RegistrationType::NID_BFGS;
  params.nid_bins = bins;
  params.max_outer_iterations = 1;
  params.disable_z_buffer_culling = disable_culling != 0;
  params.callback = [](const Eigen::Isometry3d&) {};
  vlcal::VisualCameraCalibration calib(cam, make_dataset(n_pairs, images, rows, cols, points, intensities, num_points), params);
  calib.calibrate(Eigen::Isometry3d::FromRowMajor(T_init));
  if (static_cast<int>(log.entries.size()) != n_probes + 1) return -2;
  for (int k = 0; k <= n_probes; k++) {
    const auto& e = log.entries[static_cast<size_t>(k)];
    ok_value[k] = e.ok_value ? 1 : 0;
    ok_grad[k] = e.ok_grad ? 1 : 0;
    cost_value[k] = e.cost_value;
    cost_grad[k] = e.cost_grad;
    for (int i = 0; i < 7; i++) grads[7 * k + i] = e.grad[static_cast<size_t>(i)];
  }
  // the starting parameters the reference derived from the 4x4 (Sophus::SE3d(matrix).data())
  const Sophus::SE3d start(Eigen::Isometry3d::FromRowMajor(T_init).matrix());
  for (int i = 0; i < 7; i++) start_params[i] = start.data()[i];
  return 0;
}

// dfo::NelderMead<N>::optimize for N = 2 (estimate_fov.cpp) and N = 6 (visual_camera_calibration.cpp:121)
int ref_nelder_mead(int n, double init_step, double conv_thresh, int max_iterations, double (*fn)(const double*, void*), void* user, const double* x0, double* x_out, double* y_out,
                    int* iters_out) {
  if (n == 2) return run_nelder_mead<2>(init_step, conv_thresh, max_iterations, fn, user, x0, x_out, y_out, iters_out);
  if (n == 6) return run_nelder_mead<6>(init_step, conv_thresh, max_iterations, fn, user, x0, x_out, y_out, iters_out);
  return -1;
}

}  // extern "C"
