// TEST INFRASTRUCTURE ONLY -- stand-in for <sophus/se3.hpp> (thirdparty/Sophus is an empty submodule in
// the reference tree).  What the reference's sources use:
//  * include/vlcal/costs/nid_cost.hpp:38,47: Eigen::Map<Sophus::SE3<T> const>(params), storage
//    [qx qy qz qw tx ty tz], and `T_camera_lidar * point`, evaluated as the published Sophus code does
//    (SO3::operator*(Point): uv = q.vec x p; uv += uv; p + q.w * uv + q.vec x uv -- no quaternion
//    normalisation; SE3: so3 * p + translation);
//  * src/vlcal/calib/visual_camera_calibration.cpp:147-156,195,214-216,238: Sophus::SE3d from a 4x4 matrix,
//    matrix(), data(), num_parameters, inverse(), `SE3d * Map<const SE3d>`, translation(), rotationMatrix().
#pragma once
#include <Eigen/Core>
#include <Eigen/Geometry>

namespace Sophus {

template <typename T, int Options = 0>
class SE3 {};

template <int Options>
class SE3<double, Options> {
public:
  static constexpr int num_parameters = 7;
  SE3() {
    p[0] = p[1] = p[2] = 0.0;
    p[3] = 1.0;
    p[4] = p[5] = p[6] = 0.0;
  }
  explicit SE3(const Eigen::Matrix4d& m) {
    Eigen::Matrix3d r;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) r(i, j) = m(i, j);
    Eigen::rotation_to_quaternion(r, p);
    for (int i = 0; i < 3; i++) p[4 + i] = m(i, 3);
  }
  static SE3 FromParams(const double* q) {
    SE3 s;
    for (int i = 0; i < 7; i++) s.p[i] = q[i];
    return s;
  }
  double* data() { return p; }
  const double* data() const { return p; }
  Eigen::Matrix3d rotationMatrix() const {
    const double x = p[0], y = p[1], z = p[2], w = p[3];
    Eigen::Matrix3d R;
    R(0, 0) = 1 - 2 * (y * y + z * z);
    R(0, 1) = 2 * (x * y - z * w);
    R(0, 2) = 2 * (x * z + y * w);
    R(1, 0) = 2 * (x * y + z * w);
    R(1, 1) = 1 - 2 * (x * x + z * z);
    R(1, 2) = 2 * (y * z - x * w);
    R(2, 0) = 2 * (x * z - y * w);
    R(2, 1) = 2 * (y * z + x * w);
    R(2, 2) = 1 - 2 * (x * x + y * y);
    return R;
  }
  Eigen::Vector3d translation() const { return Eigen::Vector3d(p[4], p[5], p[6]); }
  Eigen::Matrix4d matrix() const {
    Eigen::Matrix4d m = Eigen::Matrix4d::Zero();
    const Eigen::Matrix3d R = rotationMatrix();
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) m(i, j) = R(i, j);
      m(i, 3) = p[4 + i];
    }
    m(3, 3) = 1.0;
    return m;
  }
  SE3 inverse() const {  // (conj(q), -R^T t)
    SE3 r;
    r.p[0] = -p[0], r.p[1] = -p[1], r.p[2] = -p[2], r.p[3] = p[3];
    const Eigen::Matrix3d R = r.rotationMatrix();
    for (int i = 0; i < 3; i++) r.p[4 + i] = -(R(i, 0) * p[4] + R(i, 1) * p[5] + R(i, 2) * p[6]);
    return r;
  }
  // group product with anything exposing the 7 parameters through data()
  template <typename Other>
  SE3 operator*(const Other& o) const {
    const double* a = p;
    const double* b = o.data();
    SE3 r;
    r.p[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r.p[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r.p[1] = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
    r.p[2] = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
    const Eigen::Matrix3d R = rotationMatrix();
    for (int i = 0; i < 3; i++) r.p[4 + i] = (R(i, 0) * b[4] + R(i, 1) * b[5] + R(i, 2) * b[6]) + a[4 + i];
    return r;
  }

private:
  double p[7];  // [qx qy qz qw tx ty tz]
};

typedef SE3<double> SE3d;

}  // namespace Sophus

namespace Eigen {
template <typename T>
class Map<Sophus::SE3<T> const> {
public:
  explicit Map(const T* params) : q(params) {}
  const T* data() const { return q; }
  Matrix<T, 3, 1> operator*(const Matrix<double, 3, 1>& p) const {
    const T &qx = q[0], &qy = q[1], &qz = q[2], &qw = q[3];
    T uvx = qy * p[2] - qz * p[1];
    T uvy = qz * p[0] - qx * p[2];
    T uvz = qx * p[1] - qy * p[0];
    uvx += uvx;
    uvy += uvy;
    uvz += uvz;
    const T cx = qy * uvz - qz * uvy;
    const T cy = qz * uvx - qx * uvz;
    const T cz = qx * uvy - qy * uvx;
    Matrix<T, 3, 1> r;
    r[0] = (p[0] + qw * uvx + cx) + q[4];
    r[1] = (p[1] + qw * uvy + cy) + q[5];
    r[2] = (p[2] + qw * uvz + cz) + q[6];
    return r;
  }

private:
  const T* q;
};
}  // namespace Eigen
