// TEST INFRASTRUCTURE ONLY -- stand-in for <sophus/se3.hpp> (thirdparty/Sophus is an empty submodule in
// the reference tree).  Only what include/vlcal/costs/nid_cost.hpp:38,47 uses:
// Eigen::Map<Sophus::SE3<T> const>(params) with storage [qx qy qz qw tx ty tz] and `T_camera_lidar * point`,
// evaluated as the published Sophus code does (SO3::operator*(Point): uv = q.vec x p; uv += uv;
// p + q.w * uv + q.vec x uv -- no quaternion normalisation; SE3: so3 * p + translation).
#pragma once
#include <Eigen/Core>

namespace Sophus {
template <typename T>
class SE3 {};
}  // namespace Sophus

namespace Eigen {
template <typename T>
class Map<Sophus::SE3<T> const> {
public:
  explicit Map(const T* params) : q(params) {}
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
