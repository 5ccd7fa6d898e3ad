// TEST INFRASTRUCTURE ONLY -- stand-in for <opencv2/core.hpp> (OpenCV is not installed here): the slice of
// cv::Mat the reference's hot-path sources touch (rows / cols / data / at<T>(y, x) / clone(), the
// (rows, cols, type, Scalar) constructor with saturating fill).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8UC1 0
#define CV_32SC1 4
#define CV_32FC1 5
#define CV_64FC1 6
#define CV_8UC4 24

namespace cv {

struct Scalar {
  double v[4];
  static Scalar all(double x) { return Scalar{{x, x, x, x}}; }
};

class Mat {
public:
  int rows = 0, cols = 0;
  unsigned char* data = nullptr;
  size_t step = 0;

  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, const Scalar& s) {
    create(r, c, type);
    for (int y = 0; y < r; y++)
      for (int x = 0; x < c; x++) {
        if (type == CV_64FC1) at<double>(y, x) = s.v[0];
        else if (type == CV_32FC1) at<float>(y, x) = static_cast<float>(s.v[0]);  // saturate_cast<float>(double): DBL_MAX -> +inf
        else if (type == CV_32SC1) at<int>(y, x) = static_cast<int>(s.v[0]);
        else at<unsigned char>(y, x) = static_cast<unsigned char>(s.v[0]);
      }
  }
  // non-owning view of caller memory (the driver's way in)
  Mat(int r, int c, int type, void* ptr) : rows(r), cols(c), data(static_cast<unsigned char*>(ptr)), step(size_t(c) * elem_size(type)), type_(type) {}

  int type() const { return type_; }
  template <typename T>
  T& at(int y, int x) { return *reinterpret_cast<T*>(data + size_t(y) * step + size_t(x) * sizeof(T)); }
  template <typename T>
  const T& at(int y, int x) const { return *reinterpret_cast<const T*>(data + size_t(y) * step + size_t(x) * sizeof(T)); }
  template <typename T>
  T* ptr(int y = 0) { return reinterpret_cast<T*>(data + size_t(y) * step); }
  template <typename T>
  const T* ptr(int y = 0) const { return reinterpret_cast<const T*>(data + size_t(y) * step); }
  // convertTo(dst, CV_64FC1, alpha) from 8-bit: saturate_cast<double>(src * alpha)
  void convertTo(Mat& dst, int rtype, double alpha = 1.0) const {
    dst = Mat(rows, cols, rtype);
    for (int y = 0; y < rows; y++)
      for (int x = 0; x < cols; x++) {
        const double v = (type_ == CV_8UC1 ? double(at<unsigned char>(y, x)) : at<double>(y, x)) * alpha;
        if (rtype == CV_64FC1) dst.at<double>(y, x) = v;
        else dst.at<unsigned char>(y, x) = static_cast<unsigned char>(v);
      }
  }
  Mat clone() const {
    Mat m(rows, cols, type_);
    for (int y = 0; y < rows; y++) std::memcpy(m.data + size_t(y) * m.step, data + size_t(y) * step, m.step);
    return m;
  }

private:
  static size_t elem_size(int type) { return type == CV_64FC1 ? 8 : (type == CV_32FC1 || type == CV_32SC1 || type == CV_8UC4) ? 4 : 1; }
  void create(int r, int c, int type) {
    rows = r;
    cols = c;
    type_ = type;
    step = size_t(c) * elem_size(type);
    storage = std::make_shared<std::vector<unsigned char>>(size_t(r) * step);
    data = storage->data();
  }
  int type_ = CV_8UC1;
  std::shared_ptr<std::vector<unsigned char>> storage;
};

}  // namespace cv
