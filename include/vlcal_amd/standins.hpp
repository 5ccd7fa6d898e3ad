// standins.hpp -- minimal stand-ins for the third-party types that appear in the signatures of the
// reference classes this repository replaces (Eigen, OpenCV, ceres::Jet, camera::GenericCameraBase,
// vlcal::Frame).  None of those libraries exist in this build image, so the drop-in headers are
// compiled and tested against these PODs; a reference checkout defines NIDREG_WITH_REFERENCE_DEPS
// and gets the real headers instead.  Member names are the ones the
// reference's call sites use (frame.hpp:63-69, generic_camera_base.hpp:18-41, cv::Mat rows/cols/
// data/step, ceres::Jet a/v).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace Eigen {
struct Vector4d {
  double v[4];
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
  const double* data() const { return v; }
};
struct Vector3d {
  double v[3];
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
};
struct Vector2d {
  double v[2];
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
};
// row-major 4x4 here; the real Eigen::Isometry3d is column-major (the wrapper transposes, see
// cost_calculator_nid.hpp)
struct Isometry3d {
  double m[16];
  static Isometry3d Identity() {
    Isometry3d T{};
    T.m[0] = T.m[5] = T.m[10] = T.m[15] = 1.0;
    return T;
  }
  double operator()(int r, int c) const { return m[r * 4 + c]; }
  double& operator()(int r, int c) { return m[r * 4 + c]; }
};
}  // namespace Eigen

namespace ceres {
template <typename T, int N>
struct Jet {
  T a;
  T v[N];
  Jet() : a(), v() {}
  explicit Jet(const T& value) : a(value), v() {}
};
}  // namespace ceres

namespace cv {
enum { CV_8UC1_ = 0, CV_64FC1_ = 6 };
struct Mat {
  int rows = 0, cols = 0;
  int type_ = CV_8UC1_;
  unsigned char* data = nullptr;
  size_t step = 0;  // bytes per row
  std::shared_ptr<std::vector<unsigned char>> storage;
  Mat() {}
  Mat(int r, int c, int type) : rows(r), cols(c), type_(type) {
    const size_t es = type == CV_64FC1_ ? 8 : 1;
    step = size_t(c) * es;
    storage = std::make_shared<std::vector<unsigned char>>(size_t(r) * step);
    data = storage->data();
  }
  int type() const { return type_; }
  template <typename T> T& at(int y, int x) { return *reinterpret_cast<T*>(data + size_t(y) * step + size_t(x) * sizeof(T)); }
  template <typename T> const T& at(int y, int x) const { return *reinterpret_cast<const T*>(data + size_t(y) * step + size_t(x) * sizeof(T)); }
};
}  // namespace cv

namespace vlcal {
// frame.hpp:12-74 (only what the cost functions read)
struct Frame {
  using Ptr = std::shared_ptr<Frame>;
  using ConstPtr = std::shared_ptr<const Frame>;
  size_t size() const { return num_points; }
  size_t num_points = 0;
  Eigen::Vector4d* points = nullptr;
  double* intensities = nullptr;
};
// visual_lidar_data.hpp:10-23
struct VisualLiDARData {
  using Ptr = std::shared_ptr<VisualLiDARData>;
  using ConstPtr = std::shared_ptr<const VisualLiDARData>;
  VisualLiDARData(const cv::Mat& image, const Frame::ConstPtr& points) : image(image), points(points) {}
  cv::Mat image;
  Frame::ConstPtr points;
};
}  // namespace vlcal
