// TEST INFRASTRUCTURE ONLY (see jet.hpp header).  PARITY: bit-identical to the reference's own sources
// compiled against stand-in third-party headers (oracle/_ref, tests/test_reference_build.py); the third-party
// arithmetic itself (Ceres / Sophus / Eigen / OpenCV, absent here) is restated and UNPINNED.
//
// nid_oracle.cpp -- CPU restatement, statement by statement, of the reference hot path:
//   include/vlcal/costs/nid_cost.hpp:21-116                 -> nid_cost_ref<T>()
//   src/vlcal/calib/cost_calculator_nid.cpp:13-67           -> cost_calculator_nid_ref()
//   src/vlcal/calib/visual_camera_calibration.cpp:147-173   -> trust gate (multi_nid_gate)
//   src/vlcal/calib/view_culling.cpp:13-92                  -> view_culling_ref()
//   src/vlcal/common/estimate_fov.cpp:17-51                 -> estimate_camera_fov_ref()
//   include/dfo/nelder_mead.hpp:11-113                      -> nelder_mead_ref()
//   src/vlcal/common/points_color_updater.cpp:37-61         -> points_color_update_ref()
//   src/vlcal/preprocess/generate_lidar_image.cpp:7-41      -> generate_lidar_image_ref()
//   src/vlcal/preprocess/preprocess.cpp:464-473             -> equalize_intensities_ref()
// Third-party arithmetic restated from published definitions (sources absent from
// /root/reference): ceres::Jet (jet.hpp), Sophus SO3/SE3 `operator*(point)`
// (uv = 2 q.vec x p; p' = p + q.w uv + q.vec x uv + t, no quaternion normalisation),
// Eigen normalized()/AngleAxis(angle = 2 atan2(|vec|, |w|)), OpenCV Mat::at / saturate to float.
// Exposed through a C ABI (ctypes) to tests/, smoke() and bench.py's cpu_baseline only.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <numeric>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "cameras.hpp"

namespace oracle {

// double -> int as the reference's x86-64 build does it (cvttsd2si): NaN / out of range -> INT_MIN
static inline int cast_int(double d) {
  if (!(d > -2147483649.0 && d < 2147483648.0)) return INT_MIN;
  return static_cast<int>(d);
}

// Sophus SE3<T> * Vector3d with storage [qx qy qz qw tx ty tz] (nid_cost.hpp:38,47)
template <typename T>
static inline V3<T> se3_mul_point(const T* params, const double* p) {
  const T qx = params[0], qy = params[1], qz = params[2], qw = params[3];
  // uv = q.vec().cross(p); uv += uv;
  T uvx = qy * p[2] - qz * p[1];
  T uvy = qz * p[0] - qx * p[2];
  T uvz = qx * p[1] - qy * p[0];
  uvx += uvx;
  uvy += uvy;
  uvz += uvz;
  // p + q.w() * uv + q.vec().cross(uv)
  const T cx = qy * uvz - qz * uvy;
  const T cy = qz * uvx - qx * uvz;
  const T cz = qx * uvy - qy * uvx;
  V3<T> r;
  r.x = (p[0] + qw * uvx + cx) + params[4];
  r.y = (p[1] + qw * uvy + cy) + params[5];
  r.z = (p[2] + qw * uvz + cz) + params[6];
  return r;
}

struct NIDInputs {
  const CameraBase* proj;
  const double* image;  // CV_64FC1, rows x cols, [0,1]
  int rows, cols;
  const double* points;  // xyz1, stride 4 doubles (Eigen::Vector4d, frame.hpp:66)
  const double* intensities;
  int64_t num_points;
  int bins;
};

// nid_cost.hpp:29-33 -- spline_coeffs (4x4) / 6
static const double SPLINE[4][4] = {
  {1.0 / 6.0, -3.0 / 6.0, 3.0 / 6.0, -1.0 / 6.0},
  {4.0 / 6.0, 0.0 / 6.0, -6.0 / 6.0, 3.0 / 6.0},
  {1.0 / 6.0, 3.0 / 6.0, 3.0 / 6.0, -3.0 / 6.0},
  {0.0 / 6.0, 0.0 / 6.0, 0.0 / 6.0, 1.0 / 6.0}};

// nid_cost.hpp:46-84 -- the per-point accumulation loop over [begin, end)
template <typename T>
static void nid_accumulate(const NIDInputs& in, const T* params, int64_t begin, int64_t end, T* hist, T* hist_image, double* hist_points, int64_t* num_outliers) {
  const int bins = in.bins;
  for (int64_t i = begin; i < end; i++) {
    const V3<T> pt_camera = se3_mul_point<T>(params, in.points + 4 * i);
    const double intensity = in.intensities[i];
    const int bin_points = std::max<int>(0, std::min<int>(bins - 1, cast_int(intensity * bins)));

    const V2<T> projected = (*in.proj)(pt_camera);
    const int knot_x = cast_int(std::floor(get_real(projected.x)));
    const int knot_y = cast_int(std::floor(get_real(projected.y)));
    const T sx = projected.x - static_cast<double>(knot_x);
    const T sy = projected.y - static_cast<double>(knot_y);

    if (knot_x < 0 || knot_y < 0 || knot_x >= in.cols || knot_y >= in.rows) {
      (*num_outliers)++;
      continue;
    }

    hist_points[bin_points]++;

    T se[4][2];
    se[0][0] = T(1.0);
    se[0][1] = T(1.0);
    se[1][0] = sx;
    se[1][1] = sy;
    se[2][0] = sx * sx;
    se[2][1] = sy * sy;
    se[3][0] = se[2][0] * sx;
    se[3][1] = se[2][1] * sy;

    T beta[4][2];
    for (int r = 0; r < 4; r++) {
      for (int c = 0; c < 2; c++) {
        T acc = SPLINE[r][0] * se[0][c];
        acc += SPLINE[r][1] * se[1][c];
        acc += SPLINE[r][2] * se[2][c];
        acc += SPLINE[r][3] * se[3][c];
        beta[r][c] = acc;
      }
    }

    int knots_x[4] = {knot_x - 1, knot_x, knot_x + 1, knot_x + 2};
    int knots_y[4] = {knot_y - 1, knot_y, knot_y + 1, knot_y + 2};
    for (int k = 0; k < 4; k++) {
      knots_x[k] = std::min(std::max(knots_x[k], 0), in.cols - 1);
      knots_y[k] = std::min(std::max(knots_y[k], 0), in.rows - 1);
    }

    for (int a = 0; a < 4; a++) {
      for (int b = 0; b < 4; b++) {
        const T w = beta[a][0] * beta[b][1];
        const double pix = in.image[static_cast<int64_t>(knots_y[b]) * in.cols + knots_x[a]];
        const int bin_image = std::min<int>(cast_int(pix * bins), bins - 1);
        hist[static_cast<int64_t>(bin_image) * bins + bin_points] += w;
        hist_image[bin_image] += w;
      }
    }
  }
}

// nid_cost.hpp:86-104 -- entropy tail.  Returns false on non-finite NID.
template <typename T>
static bool nid_entropy_tail(int bins, T* hist, T* hist_image, double* hist_points, T* residual) {
  double sum = 0.0;
  for (int i = 0; i < bins; i++) sum += hist_points[i];

  for (int i = 0; i < bins; i++) hist_image[i] = hist_image[i] / sum;
  for (int i = 0; i < bins; i++) hist_points[i] = hist_points[i] / sum;
  for (int64_t i = 0; i < static_cast<int64_t>(bins) * bins; i++) hist[i] = hist[i] / sum;

  T H_image_acc(0.0);
  for (int i = 0; i < bins; i++) H_image_acc += hist_image[i] * log(hist_image[i] + 1e-6);
  const T H_image = -H_image_acc;
  double H_points_acc = 0.0;
  for (int i = 0; i < bins; i++) H_points_acc += hist_points[i] * std::log(hist_points[i] + 1e-6);
  const double H_points = -H_points_acc;
  T H_joint_acc(0.0);
  // Eigen's default storage is column-major: (hist.array() * log).sum() walks columns
  // (bin_points) outermost.  Follow the same visiting order.
  for (int c = 0; c < bins; c++) {
    for (int r = 0; r < bins; r++) {
      const T& h = hist[static_cast<int64_t>(r) * bins + c];
      H_joint_acc += h * log(h + 1e-6);
    }
  }
  const T H_image_points = -H_joint_acc;
  const T MI = H_image + H_points - H_image_points;
  const T NID = (H_image_points - MI) / H_image_points;

  if (!std::isfinite(get_real(NID))) {
    return false;
  }
  residual[0] = NID;
  return true;
}

// nid_cost.hpp:36-107 -- NIDCost::operator()<T>.  hist is row-major [bin_image][bin_points].
// threads <= 1: the reference's serial loop.  threads > 1: "generous" CPU baseline
// (same arithmetic, OpenMP over point slices with per-thread private histograms).
template <typename T>
static bool nid_cost_ref(const NIDInputs& in, const T* params, T* residual, T* hist_out, T* hist_image_out, double* hist_points_out, int64_t* outliers_out, int threads) {
  const int bins = in.bins;
  const int64_t nb2 = static_cast<int64_t>(bins) * bins;
  std::vector<T> hist(nb2, T(0.0));
  std::vector<T> hist_image(bins, T(0.0));
  std::vector<double> hist_points(bins, 0.0);
  int64_t num_outliers = 0;

  if (threads <= 1) {
    nid_accumulate<T>(in, params, 0, in.num_points, hist.data(), hist_image.data(), hist_points.data(), &num_outliers);
  } else {
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
    {
      std::vector<T> h(nb2, T(0.0));
      std::vector<T> hi(bins, T(0.0));
      std::vector<double> hp(bins, 0.0);
      int64_t no = 0;
      const int tid = omp_get_thread_num();
      const int nt = omp_get_num_threads();
      const int64_t b = in.num_points * tid / nt;
      const int64_t e = in.num_points * (tid + 1) / nt;
      nid_accumulate<T>(in, params, b, e, h.data(), hi.data(), hp.data(), &no);
#pragma omp critical
      {
        for (int64_t k = 0; k < nb2; k++) hist[k] += h[k];
        for (int k = 0; k < bins; k++) hist_image[k] += hi[k];
        for (int k = 0; k < bins; k++) hist_points[k] += hp[k];
        num_outliers += no;
      }
    }
#else
    nid_accumulate<T>(in, params, 0, in.num_points, hist.data(), hist_image.data(), hist_points.data(), &num_outliers);
#endif
  }

  if (outliers_out) *outliers_out = num_outliers;
  // raw (un-normalised) histograms for the tests
  if (hist_out) std::copy(hist.begin(), hist.end(), hist_out);
  if (hist_image_out) std::copy(hist_image.begin(), hist_image.end(), hist_image_out);
  if (hist_points_out) std::copy(hist_points.begin(), hist_points.end(), hist_points_out);

  return nid_entropy_tail<T>(bins, hist.data(), hist_image.data(), hist_points.data(), residual);
}

// ---------------------------------------------------------------------------------------------
// 4x4 row-major isometry helpers
static inline void iso_mul_point4(const double* T, const double* p, double* out) {
  // Eigen 4x4 * 4x1 coefficient product, summed left to right
  for (int r = 0; r < 4; r++) {
    out[r] = ((T[r * 4 + 0] * p[0] + T[r * 4 + 1] * p[1]) + T[r * 4 + 2] * p[2]) + T[r * 4 + 3] * p[3];
  }
}

// cost_calculator_nid.cpp:21-67.  hist_out is row-major [image_bin][lidar_bin] (int64).
static double cost_calculator_nid_ref(
  const CameraBase* proj, const uint8_t* image, int rows, int cols, const double* points, const double* intensities, int64_t num_points, int bins, double max_fov,
  const double* T_camera_lidar, int64_t* hist_out) {
  const int64_t nb2 = static_cast<int64_t>(bins) * bins;
  std::vector<int64_t> hist(nb2, 0);
  std::vector<int64_t> hist_image(bins, 0);
  std::vector<int64_t> hist_points(bins, 0);
  const double cos_fov = std::cos(max_fov);

  for (int64_t i = 0; i < num_points; i++) {
    double pc[4];
    iso_mul_point4(T_camera_lidar, points + 4 * i, pc);
    const V3<double> p3{pc[0], pc[1], pc[2]};
    if (normalized(p3).z < cos_fov) {
      continue;  // out of FoV
    }
    const V2<double> uv = proj->project(p3);
    const int px = cast_int(uv.x);  // .cast<int>() truncates toward zero
    const int py = cast_int(uv.y);
    if (px < 0 || py < 0 || px >= cols || py >= rows) {
      continue;  // out of image
    }
    const double pixel = image[static_cast<int64_t>(py) * cols + px] / 255.0;
    const double lidar_intensity = intensities[i];
    const int image_bin = std::max<int>(0, std::min<int>(bins - 1, cast_int(pixel * bins)));
    const int lidar_bin = std::max<int>(0, std::min<int>(bins - 1, cast_int(lidar_intensity * bins)));
    hist[static_cast<int64_t>(image_bin) * bins + lidar_bin]++;
    hist_image[image_bin]++;
    hist_points[lidar_bin]++;
  }
  if (hist_out) std::copy(hist.begin(), hist.end(), hist_out);

  int64_t sum_i = 0;
  for (int i = 0; i < bins; i++) sum_i += hist_image[i];
  const int sum = static_cast<int>(sum_i);  // `const int sum = hist_image.sum()` (:54)

  // cast<double>() / sum -- integer sum converted to double by the division
  double Hr_acc = 0.0, Hs_acc = 0.0, Hrs_acc = 0.0;
  for (int i = 0; i < bins; i++) {
    const double p = static_cast<double>(hist_image[i]) / sum;
    Hr_acc += p * std::log(p + 1e-6);
  }
  for (int i = 0; i < bins; i++) {
    const double p = static_cast<double>(hist_points[i]) / sum;
    Hs_acc += p * std::log(p + 1e-6);
  }
  for (int c = 0; c < bins; c++) {  // column-major visiting order, as Eigen's .sum()
    for (int r = 0; r < bins; r++) {
      const double p = static_cast<double>(hist[static_cast<int64_t>(r) * bins + c]) / sum;
      Hrs_acc += p * std::log(p + 1e-6);
    }
  }
  const double Hr = -Hr_acc, Hs = -Hs_acc, Hrs = -Hrs_acc;
  const double MI = Hr + Hs - Hrs;
  const double NID = (Hrs - MI) / Hrs;
  return NID;
}

// ---------------------------------------------------------------------------------------------
// include/dfo/nelder_mead.hpp:32-113, N-generic.  f maps N doubles -> double.
struct NMParams {
  double init_step = 0.1, alpha = 1.0, gamma = 2.0, rho = 0.5, sigma = 0.5;
  int max_iterations = 1024;
  double convergence_var_thresh = 1e-5;
};
struct NMResult {
  bool converged = false;
  int num_iterations = 0;
  std::vector<double> x;
  double y = 0.0;
};

static NMResult nelder_mead_ref(int N, const NMParams& params, const std::function<double(const double*)>& function, const double* x0) {
  using VecM = std::vector<double>;  // [value, x_0..x_{N-1}]
  NMResult result;
  std::vector<VecM> x(1, VecM(N + 1));
  x[0][0] = function(x0);
  for (int k = 0; k < N; k++) x[0][1 + k] = x0[k];

  for (int i = 0; i < N; i++) {
    VecM xi(N + 1);
    for (int k = 0; k < N; k++) xi[1 + k] = x0[k];
    xi[1 + i] += params.init_step;
    xi[0] = function(xi.data() + 1);
    x.push_back(xi);
  }

  const auto is_converged = [&](const std::vector<VecM>& xs) {
    VecM mean(N + 1, 0.0);
    for (const auto& xi : xs)
      for (int k = 0; k <= N; k++) mean[k] += xi[k];
    for (int k = 0; k <= N; k++) mean[k] /= xs.size();
    VecM var(N + 1, 0.0);
    for (const auto& xi : xs)
      for (int k = 0; k <= N; k++) var[k] = var[k] + (xi[k] - mean[k]) * (xi[k] - mean[k]);
    double s = 0.0;
    for (int k = 1; k <= N; k++) s += var[k];
    return s < params.convergence_var_thresh;
  };

  for (int i = 0; i < params.max_iterations; i++) {
    result.num_iterations = i;
    std::sort(x.begin(), x.end(), [](const VecM& lhs, const VecM& rhs) { return lhs[0] < rhs[0]; });
    if (is_converged(x)) {
      result.converged = true;
      break;
    }

    VecM xo(N + 1, 0.0);
    for (size_t j = 0; j + 1 < x.size(); j++)
      for (int k = 0; k <= N; k++) xo[k] += x[j][k];
    for (int k = 0; k <= N; k++) xo[k] /= (x.size() - 1);
    xo[0] = function(xo.data() + 1);

    VecM xr(N + 1);
    for (int k = 0; k <= N; k++) xr[k] = xo[k] + params.alpha * (xo[k] - x.back()[k]);
    xr[0] = function(xr.data() + 1);

    if (x[0][0] <= xr[0] && xr[0] < x[N - 1][0]) {
      x.back() = xr;
    } else if (xr[0] < x[0][0]) {
      VecM xe(N + 1);
      for (int k = 0; k <= N; k++) xe[k] = xo[k] + params.gamma * (xo[k] - x.back()[k]);
      xe[0] = function(xe.data() + 1);
      if (xe[0] < xr[0]) {
        x.back() = xe;
      } else {
        x.back() = xr;
      }
    } else {
      VecM xc(N + 1);
      for (int k = 0; k <= N; k++) xc[k] = xo[k] + params.rho * (xo[k] - x.back()[k]);
      xc[0] = function(xc.data() + 1);
      if (xc[0] < x.back()[0]) {
        x.back() = xc;
      } else {
        for (size_t j = 1; j < x.size(); j++) {
          for (int k = 0; k <= N; k++) x[j][k] = x[0][k] + params.rho * (x[j][k] - x[0][k]);
          x[j][0] = function(x[j].data() + 1);
        }
      }
    }
  }

  result.x.assign(x[0].begin() + 1, x[0].end());
  result.y = x[0][0];
  return result;
}

// estimate_fov.cpp:17-34
static V3<double> estimate_direction_ref(const CameraBase* proj, double u, double v) {
  // AngleAxis(x0, UnitX) * AngleAxis(x1, UnitY) * UnitZ, evaluated the way Eigen does: each AngleAxis becomes a
  // quaternion (w = cos(angle/2), vec = sin(angle/2) axis), the two are multiplied (generic quaternion product),
  // and the product rotates UnitZ by uv = vec x v; uv += uv; v + w uv + vec x uv
  const auto to_dir = [](const double* x) {
    const double ha = 0.5 * x[0], hb = 0.5 * x[1];
    const double aw = std::cos(ha), sa = std::sin(ha), bw = std::cos(hb), sb = std::sin(hb);
    const double ax = sa * 1.0, ay = sa * 0.0, az = sa * 0.0;
    const double bx = sb * 0.0, by = sb * 1.0, bz = sb * 0.0;
    const double qw = aw * bw - ax * bx - ay * by - az * bz;
    const double qx = aw * bx + ax * bw + ay * bz - az * by;
    const double qy = aw * by + ay * bw + az * bx - ax * bz;
    const double qz = aw * bz + az * bw + ax * by - ay * bx;
    const double v[3] = {0.0, 0.0, 1.0};
    double ux = qy * v[2] - qz * v[1], uy = qz * v[0] - qx * v[2], uz = qx * v[1] - qy * v[0];
    ux += ux;
    uy += uy;
    uz += uz;
    return V3<double>{(v[0] + qw * ux) + (qy * uz - qz * uy), (v[1] + qw * uy) + (qz * ux - qx * uz), (v[2] + qw * uz) + (qx * uy - qy * ux)};
  };
  const auto f = [&](const double* x) {
    const V3<double> dir = to_dir(x);
    const V2<double> pr = proj->project(dir);
    const double ex = u - pr.x, ey = v - pr.y;
    const double err = ex * ex + ey * ey;
    return std::isfinite(err) ? err : std::numeric_limits<double>::max();
  };
  NMParams params;  // defaults: init_step 0.1, max_iter 1024, conv 1e-5 (nelder_mead.hpp:12)
  const double x0[2] = {0.0, 0.0};
  const NMResult result = nelder_mead_ref(2, params, f, x0);
  return to_dir(result.x.data());
}

// estimate_fov.cpp:36-51
static double estimate_camera_fov_ref(const CameraBase* proj, int width, int height) {
  const double corners[3][2] = {{0.0, 0.0}, {static_cast<double>(width / 2), 0.0}, {0.0, static_cast<double>(height / 2)}};
  double max_fov = 0.0;
  for (int k = 0; k < 3; k++) {
    const V3<double> dir = estimate_direction_ref(proj, corners[k][0], corners[k][1]);
    const double fov = std::acos(normalized(dir).z);
    if (fov > max_fov) max_fov = fov;
  }
  return max_fov;
}

// view_culling.cpp:21-92.  Returns surviving indices (into the input cloud), in order.
static std::vector<int> view_culling_ref(
  const CameraBase* proj, int width, int height, double min_z, bool enable_depth_buffer_culling, const double* points, int64_t num_points, const double* T_camera_lidar) {
  std::vector<double> points_camera(4 * num_points);
  for (int64_t i = 0; i < num_points; i++) iso_mul_point4(T_camera_lidar, points + 4 * i, points_camera.data() + 4 * i);

  std::vector<int> indices;
  std::vector<int> proj_x, proj_y;
  indices.reserve(num_points);
  // CV_32FC1 filled with saturate_cast<float>(DBL_MAX) = +inf
  std::vector<float> dist_map(static_cast<size_t>(width) * height, static_cast<float>(std::numeric_limits<double>::max()));
  std::vector<int> index_map(static_cast<size_t>(width) * height, -1);

  for (int64_t i = 0; i < num_points; i++) {
    const double* pc = points_camera.data() + 4 * i;
    // normalises the 4-vector (x, y, z, 1) -- view_culling.cpp:45
    const double n4 = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2] + pc[3] * pc[3]);
    if (pc[2] / n4 < min_z) continue;

    const V2<double> uv = proj->project(V3<double>{pc[0], pc[1], pc[2]});
    const int px = cast_int(uv.x), py = cast_int(uv.y);
    if (px < 0 || py < 0 || px >= width || py >= height) continue;

    indices.push_back(static_cast<int>(i));
    proj_x.push_back(px);
    proj_y.push_back(py);

    if (enable_depth_buffer_culling) {
      const double dist = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
      float& d = dist_map[static_cast<size_t>(py) * width + px];
      if (dist > d) continue;
      d = static_cast<float>(dist);
      index_map[static_cast<size_t>(py) * width + px] = static_cast<int>(i);
    }
  }

  if (enable_depth_buffer_culling) {
    std::vector<int> new_indices;
    new_indices.reserve(indices.size());
    for (size_t k = 0; k < indices.size(); k++) {
      const int index = indices[k];
      const double* pc = points_camera.data() + 4 * static_cast<int64_t>(index);
      const double dist = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
      if (dist > dist_map[static_cast<size_t>(proj_y[k]) * width + proj_x[k]] + 0.1) continue;
      new_indices.push_back(index);
    }
    indices = std::move(new_indices);
  }
  return indices;
}

// points_color_updater.cpp:37-61.  intensity_colors / colors_out: n x 4 floats (Eigen::Vector4f).
// `color * blend_weight`: Eigen converts the double scalar to the expression's float scalar first.
static void points_color_update_ref(
  const CameraBase* proj, const uint8_t* image, int rows, int cols, const double* points, int64_t num_points, const float* intensity_colors, double min_nz, const double* T_camera_lidar,
  double blend_weight, float* colors_out) {
  const float wf = static_cast<float>(blend_weight);
  const float omwf = static_cast<float>(1.0 - blend_weight);
  for (int64_t i = 0; i < num_points; i++) {
    float* out = colors_out + 4 * i;
    out[0] = out[1] = out[2] = out[3] = 0.0f;  // Vector4f::Zero() for skipped points
    double pc[4];
    iso_mul_point4(T_camera_lidar, points + 4 * i, pc);
    const V3<double> p3{pc[0], pc[1], pc[2]};
    if (normalized(p3).z < min_nz) continue;  // out of FoV
    const V2<double> uv = proj->project(p3);
    const int px = cast_int(uv.x), py = cast_int(uv.y);
    if (px < 0 || py < 0 || px >= cols || py >= rows) continue;  // out of image
    const unsigned char pix = image[static_cast<int64_t>(py) * cols + px];
    const float color[4] = {pix / 255.0f, pix / 255.0f, pix / 255.0f, 1.0f};
    const float* ic = intensity_colors + 4 * i;
    for (int k = 0; k < 4; k++) out[k] = color[k] * wf + ic[k] * omwf;
  }
}

// generate_lidar_image.cpp:7-41.  intensity_image: rows x cols doubles, index_image: rows x cols int32.
static void generate_lidar_image_ref(
  const CameraBase* proj, int width, int height, double min_z, const double* points, const double* intensities, int64_t num_points, const double* T_camera_lidar, double* intensity_image,
  int32_t* index_image) {
  const size_t npix = static_cast<size_t>(width) * height;
  std::vector<double> sq_dist_image(npix, std::numeric_limits<double>::max());
  std::fill(intensity_image, intensity_image + npix, 0.0);
  std::fill(index_image, index_image + npix, -1);
  for (int64_t i = 0; i < num_points; i++) {
    double pc[4];
    iso_mul_point4(T_camera_lidar, points + 4 * i, pc);
    const V3<double> p3{pc[0], pc[1], pc[2]};
    if (normalized(p3).z < min_z) continue;
    const V2<double> uv = proj->project(p3);
    const int px = cast_int(uv.x), py = cast_int(uv.y);
    if (px < 0 || py < 0 || px >= width || py >= height) continue;
    const double sq_dist = pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2];
    const size_t q = static_cast<size_t>(py) * width + px;
    if (sq_dist_image[q] < sq_dist) continue;
    sq_dist_image[q] = sq_dist;
    intensity_image[q] = intensities[i];
    index_image[q] = static_cast<int32_t>(i);
  }
}

// preprocess.cpp:464-473: rank equalisation of the integrated cloud's intensities into 256 levels.
// std::sort is not stable; among EQUAL intensities the rank order is unspecified in the reference, so
// only the multiset of values per tie group is defined -- the restatement uses a stable sort.
static void equalize_intensities_ref(double* intensities, int64_t n) {
  std::vector<int> indices(n);
  std::iota(indices.begin(), indices.end(), 0);
  std::stable_sort(indices.begin(), indices.end(), [&](const int lhs, const int rhs) { return intensities[lhs] < intensities[rhs]; });
  const int bins = 256;
  std::vector<double> out(n);
  for (int64_t i = 0; i < n; i++) out[indices[i]] = std::floor(bins * static_cast<double>(i) / n) / bins;
  std::copy(out.begin(), out.end(), intensities);
}

// visual_camera_calibration.cpp:149-156 -- MultiNIDCost trust gate.
// delta = init^-1 * T ; reject if |delta.t| > 0.2 or angle(delta.R) > 2 deg.
static void quat_to_rot(const double* q, double* R) {  // q = [x y z w], unit
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  R[0] = 1 - 2 * (y * y + z * z);
  R[1] = 2 * (x * y - z * w);
  R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);
  R[4] = 1 - 2 * (x * x + z * z);
  R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);
  R[7] = 2 * (y * z + x * w);
  R[8] = 1 - 2 * (x * x + y * y);
}
static bool multi_nid_gate(const double* init_se3, const double* se3) {
  // q_delta = conj(q0) * q ; t_delta = R0^T (t - t0)
  const double x0 = -init_se3[0], y0 = -init_se3[1], z0 = -init_se3[2], w0 = init_se3[3];
  const double x1 = se3[0], y1 = se3[1], z1 = se3[2], w1 = se3[3];
  const double qw = w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1;
  const double qx = w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1;
  const double qy = w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1;
  const double qz = w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1;
  double R0[9];
  quat_to_rot(init_se3, R0);
  const double dt[3] = {se3[4] - init_se3[4], se3[5] - init_se3[5], se3[6] - init_se3[6]};
  const double tx = R0[0] * dt[0] + R0[3] * dt[1] + R0[6] * dt[2];
  const double ty = R0[1] * dt[0] + R0[4] * dt[1] + R0[7] * dt[2];
  const double tz = R0[2] * dt[0] + R0[5] * dt[1] + R0[8] * dt[2];
  const double tnorm = std::sqrt(tx * tx + ty * ty + tz * tz);
  const double angle = 2.0 * std::atan2(std::sqrt(qx * qx + qy * qy + qz * qz), std::abs(qw));
  return !(tnorm > 0.2 || angle > 2.0 * M_PI / 180.0);
}

static std::shared_ptr<const CameraBase> make_camera(const char* model, const double* intr, int n_intr, const double* dist, int n_dist) {
  return create_camera(std::string(model), std::vector<double>(intr, intr + n_intr), std::vector<double>(dist, dist + n_dist));
}

}  // namespace oracle

// ---------------------------------------------------------------------------------------------
// C ABI for ctypes (tests/, smoke(), bench.py cpu_baseline).  All pointers are host memory.
extern "C" {

// returns 0 on success, -1 on unknown model / intrinsic-count mismatch (create_camera -> nullptr)
int oracle_project(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, const double* p3, int64_t n, double* uv_out) {
  auto cam = oracle::make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  for (int64_t i = 0; i < n; i++) {
    const oracle::V2<double> uv = cam->project(oracle::V3<double>{p3[3 * i], p3[3 * i + 1], p3[3 * i + 2]});
    uv_out[2 * i] = uv.x;
    uv_out[2 * i + 1] = uv.y;
  }
  return 0;
}

// projection with the 2x3 Jacobian d(uv)/d(p) obtained through Jet<7> (first three slots seeded)
int oracle_project_jacobian(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, const double* p3, int64_t n, double* uv_out, double* jac_out) {
  auto cam = oracle::make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  for (int64_t i = 0; i < n; i++) {
    oracle::V3<oracle::Jet7> p{oracle::Jet7(p3[3 * i], 0), oracle::Jet7(p3[3 * i + 1], 1), oracle::Jet7(p3[3 * i + 2], 2)};
    const oracle::V2<oracle::Jet7> uv = (*cam)(p);
    uv_out[2 * i] = uv.x.a;
    uv_out[2 * i + 1] = uv.y.a;
    for (int k = 0; k < 3; k++) {
      jac_out[6 * i + k] = uv.x.v[k];
      jac_out[6 * i + 3 + k] = uv.y.v[k];
    }
  }
  return 0;
}

// NIDCost::operator() for T=double (grad7 == NULL) or T=Jet<double,7> (grad7 != NULL).
// image: rows x cols doubles in [0,1]; points: xyz1 stride 4; se3 = [qx qy qz qw tx ty tz].
// Optional outputs (may be NULL): hist [bins*bins] row-major [bin_image][bin_points] RAW sums,
// hist_image [bins] raw, hist_points [bins] raw counts, hist_grad [7*bins*bins] (Jet partials of
// the raw joint histogram, slot-major), outliers.
// returns 1 = true, 0 = false (non-finite NID), -1 = bad camera.
int oracle_nid_cost(
  const char* model, const double* intr, int n_intr, const double* dist, int n_dist, const double* image, int rows, int cols, const double* points, const double* intensities,
  int64_t num_points, int bins, const double* se3, int threads, double* cost, double* grad7, double* hist, double* hist_image, double* hist_points, double* hist_grad,
  int64_t* outliers) {
  auto cam = oracle::make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  oracle::NIDInputs in{cam.get(), image, rows, cols, points, intensities, num_points, bins};
  const int64_t nb2 = static_cast<int64_t>(bins) * bins;
  if (grad7 == nullptr && hist_grad == nullptr) {
    double residual = std::numeric_limits<double>::quiet_NaN();
    const bool ok = oracle::nid_cost_ref<double>(in, se3, &residual, hist, hist_image, hist_points, outliers, threads);
    *cost = residual;
    return ok ? 1 : 0;
  }
  oracle::Jet7 params[7];
  for (int k = 0; k < 7; k++) params[k] = oracle::Jet7(se3[k], k);
  oracle::Jet7 residual(std::numeric_limits<double>::quiet_NaN());
  std::vector<oracle::Jet7> h(hist || hist_grad ? nb2 : 0), hi(hist_image ? bins : 0);
  const bool ok = oracle::nid_cost_ref<oracle::Jet7>(in, params, &residual, h.empty() ? nullptr : h.data(), hi.empty() ? nullptr : hi.data(), hist_points, outliers, threads);
  *cost = residual.a;
  if (grad7)
    for (int k = 0; k < 7; k++) grad7[k] = residual.v[k];
  if (hist)
    for (int64_t i = 0; i < nb2; i++) hist[i] = h[i].a;
  if (hist_grad)
    for (int k = 0; k < 7; k++)
      for (int64_t i = 0; i < nb2; i++) hist_grad[k * nb2 + i] = h[i].v[k];
  if (hist_image)
    for (int i = 0; i < bins; i++) hist_image[i] = hi[i].a;
  return ok ? 1 : 0;
}

// MultiNIDCost::operator() (visual_camera_calibration.cpp:147-173) over n_pairs pairs sharing one
// camera: trust gate against init_se3, plain sum of per-pair NIDs (+gradients), false if any failed.
// OpenMP over pairs only, exactly like the reference.  Per-pair arrays are passed as pointer tables.
int oracle_multi_nid_cost(
  const char* model, const double* intr, int n_intr, const double* dist, int n_dist, int n_pairs, const double* const* images, int rows, int cols, const double* const* points,
  const double* const* intensities, const int64_t* num_points, int bins, const double* init_se3, const double* se3, double* cost, double* grad7) {
  auto cam = oracle::make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  if (!oracle::multi_nid_gate(init_se3, se3)) return 0;
  std::vector<int> results(n_pairs, 0);
  std::vector<double> costs(n_pairs, 0.0);
  std::vector<double> grads(7 * static_cast<size_t>(n_pairs), 0.0);
#pragma omp parallel for
  for (int i = 0; i < n_pairs; i++) {
    oracle::NIDInputs in{cam.get(), images[i], rows, cols, points[i], intensities[i], num_points[i], bins};
    if (grad7) {
      oracle::Jet7 params[7];
      for (int k = 0; k < 7; k++) params[k] = oracle::Jet7(se3[k], k);
      oracle::Jet7 residual(0.0);
      results[i] = oracle::nid_cost_ref<oracle::Jet7>(in, params, &residual, nullptr, nullptr, nullptr, nullptr, 1);
      costs[i] = residual.a;
      for (int k = 0; k < 7; k++) grads[7 * i + k] = residual.v[k];
    } else {
      double residual = 0.0;
      results[i] = oracle::nid_cost_ref<double>(in, se3, &residual, nullptr, nullptr, nullptr, nullptr, 1);
      costs[i] = residual;
    }
  }
  for (int i = 1; i < n_pairs; i++) {
    costs[0] += costs[i];
    for (int k = 0; k < 7; k++) grads[k] += grads[7 * i + k];
  }
  *cost = costs[0];
  if (grad7)
    for (int k = 0; k < 7; k++) grad7[k] = grads[k];
  return std::count(results.begin(), results.end(), 0) == 0 ? 1 : 0;
}

int oracle_trust_gate(const double* init_se3, const double* se3) {
  return oracle::multi_nid_gate(init_se3, se3) ? 1 : 0;
}

// CostCalculatorNID::calculate; image: rows x cols uint8; T: 4x4 row-major T_camera_lidar;
// hist (optional): int64 [bins*bins] row-major [image_bin][lidar_bin].
int oracle_cost_calculator_nid(
  const char* model, const double* intr, int n_intr, const double* dist, int n_dist, const uint8_t* image, int rows, int cols, const double* points, const double* intensities,
  int64_t num_points, int bins, double max_fov, const double* T, double* cost, int64_t* hist) {
  auto cam = oracle::make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  *cost = oracle::cost_calculator_nid_ref(cam.get(), image, rows, cols, points, intensities, num_points, bins, max_fov, T, hist);
  return 0;
}

int oracle_estimate_camera_fov(const char* model, const double* intr, int n_intr, const double* dist, int n_dist, int width, int height, double* max_fov) {
  auto cam = oracle::make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  *max_fov = oracle::estimate_camera_fov_ref(cam.get(), width, height);
  return 0;
}

// ViewCulling::cull.  indices_out must hold num_points ints; returns the number kept (or -1).
int64_t oracle_view_culling(
  const char* model, const double* intr, int n_intr, const double* dist, int n_dist, int width, int height, int enable_depth_buffer_culling, const double* points,
  int64_t num_points, const double* T, int* indices_out) {
  auto cam = oracle::make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  const double min_z = std::cos(oracle::estimate_camera_fov_ref(cam.get(), width, height));
  const std::vector<int> idx = oracle::view_culling_ref(cam.get(), width, height, min_z, enable_depth_buffer_culling != 0, points, num_points, T);
  std::copy(idx.begin(), idx.end(), indices_out);
  return static_cast<int64_t>(idx.size());
}

// PointsColorUpdater::update; min_nz as the constructor computes it: cos(estimate_camera_fov + 0.5 deg)
int oracle_points_color_update(
  const char* model, const double* intr, int n_intr, const double* dist, int n_dist, const uint8_t* image, int rows, int cols, const double* points, int64_t num_points,
  const float* intensity_colors, const double* T, double blend_weight, float* colors_out, double* min_nz_out) {
  auto cam = oracle::make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  const double min_nz = std::cos(oracle::estimate_camera_fov_ref(cam.get(), cols, rows) + 0.5 * M_PI / 180.0);
  if (min_nz_out) *min_nz_out = min_nz;
  oracle::points_color_update_ref(cam.get(), image, rows, cols, points, num_points, intensity_colors, min_nz, T, blend_weight, colors_out);
  return 0;
}

// generate_lidar_image
int oracle_generate_lidar_image(
  const char* model, const double* intr, int n_intr, const double* dist, int n_dist, int width, int height, const double* points, const double* intensities, int64_t num_points,
  const double* T, double* intensity_image, int32_t* index_image) {
  auto cam = oracle::make_camera(model, intr, n_intr, dist, n_dist);
  if (!cam) return -1;
  const double min_z = std::cos(oracle::estimate_camera_fov_ref(cam.get(), width, height));
  oracle::generate_lidar_image_ref(cam.get(), width, height, min_z, points, intensities, num_points, T, intensity_image, index_image);
  return 0;
}

void oracle_equalize_intensities(double* intensities, int64_t n) { oracle::equalize_intensities_ref(intensities, n); }

// Nelder-Mead with a C callback (used by the parity driver in tests).
typedef double (*oracle_nm_fn)(const double* x, void* user);
int oracle_nelder_mead(int n, double init_step, double conv_thresh, int max_iterations, oracle_nm_fn fn, void* user, const double* x0, double* x_out, double* y_out, int* iters_out) {
  oracle::NMParams params;
  params.init_step = init_step;
  params.convergence_var_thresh = conv_thresh;
  params.max_iterations = max_iterations;
  const auto f = [&](const double* x) { return fn(x, user); };
  const oracle::NMResult r = oracle::nelder_mead_ref(n, params, f, x0);
  std::copy(r.x.begin(), r.x.end(), x_out);
  *y_out = r.y;
  *iters_out = r.num_iterations;
  return r.converged ? 1 : 0;
}

int oracle_num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
