// TEST INFRASTRUCTURE ONLY (see jet.hpp header).  PARITY: bit-identical to include/camera/*.hpp compiled
// from the reference tree against stand-in Eigen / ceres::Jet headers (oracle/_ref); third-party arithmetic UNPINNED.
//
// cameras.hpp -- CPU restatement of the reference's six projection functors and the
// type-erased wrapper / string factory around them:
//   include/camera/pinhole.hpp:11-52            -> PinholeProjection   ("plumb_bob")
//   include/camera/fisheye.hpp:12-37            -> FisheyeProjection   ("fisheye"|"equidistant")
//   include/camera/omnidir.hpp:12-42            -> OmnidirProjection   ("omnidir")
//   include/camera/equirectangular.hpp:12-29    -> EquirectProjection  ("equirectangular")
//   include/camera/atan.hpp:12-40               -> AtanProjection      ("atan")
//   include/camera/rational_polynomial.hpp:9-59 -> RationalProjection  ("rational_polynomial")
//   include/camera/generic_camera_base.hpp:18-41, generic_camera.hpp:16-38 -> CameraBase / Camera<P>
//   src/camera/create_camera.cpp:17-51          -> create_camera()
// Scalar-generic (T = double or Jet<7>), one virtual call per point like the reference.
#pragma once
#include <memory>
#include <string>
#include <vector>
#include "jet.hpp"

namespace oracle {

using Jet7 = Jet<7>;

template <typename T>
struct V2 {
  T x, y;
};
template <typename T>
struct V3 {
  T x, y, z;
};

// pinhole.hpp:13-38 (distortion storage order k1 k2 p1 p2 k3) and :41-51
struct PinholeProjection {
  static constexpr int num_intrinsic = 4;
  static constexpr int num_distortion = 5;
  template <typename T>
  V2<T> operator()(const double* intrinsic, const double* distortion, const V3<T>& p) const {
    const T px = p.x / p.z;
    const T py = p.y / p.z;
    const double k1 = distortion[0], k2 = distortion[1], k3 = distortion[4];
    const double p1 = distortion[2], p2 = distortion[3];
    const T x2 = px * px;
    const T y2 = py * py;
    const T r2 = x2 + y2;
    const T r4 = r2 * r2;
    const T r6 = r2 * r4;
    const T r_coeff = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const T t_coeff1 = 2.0 * px * py;
    const T t_coeff2 = r2 + 2.0 * x2;
    const T t_coeff3 = r2 + 2.0 * y2;
    const T x = r_coeff * px + p1 * t_coeff1 + p2 * t_coeff2;
    const T y = r_coeff * py + p1 * t_coeff3 + p2 * t_coeff1;
    return {intrinsic[0] * x + intrinsic[2], intrinsic[1] * y + intrinsic[3]};
  }
};

// fisheye.hpp:14-36 -- note abs(z) at :16 and the pow() calls at :17-20
struct FisheyeProjection {
  static constexpr int num_intrinsic = 4;
  static constexpr int num_distortion = 4;
  template <typename T>
  V2<T> operator()(const double* intrinsic, const double* distortion, const V3<T>& p) const {
    const T r = sqrt(p.x * p.x + p.y * p.y);
    const T theta = atan2(r, abs(p.z));
    const T theta2 = pow(theta, 2);
    const T theta4 = pow(theta, 4);
    const T theta6 = pow(theta, 6);
    const T theta8 = pow(theta, 8);
    const double k1 = distortion[0], k2 = distortion[1], k3 = distortion[2], k4 = distortion[3];
    const T theta_d = theta * (1.0 + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
    const T s = theta_d / r;
    const T dx = s * p.x;
    const T dy = s * p.y;
    return {intrinsic[0] * dx + intrinsic[2], intrinsic[1] * dy + intrinsic[3]};
  }
};

// Eigen's normalized(): v / sqrt(squaredNorm) when squaredNorm > 0, else v unchanged
template <typename T>
inline V3<T> normalized(const V3<T>& p) {
  const T z = p.x * p.x + p.y * p.y + p.z * p.z;
  if (z > 0.0) {
    const T n = sqrt(z);
    return {p.x / n, p.y / n, p.z / n};
  }
  return p;
}

// omnidir.hpp:14-41
struct OmnidirProjection {
  static constexpr int num_intrinsic = 5;
  static constexpr int num_distortion = 4;
  template <typename T>
  V2<T> operator()(const double* intrinsic, const double* distortion, const V3<T>& p) const {
    const double xi = intrinsic[4];
    const double k1 = distortion[0], k2 = distortion[1], p1 = distortion[2], p2 = distortion[3];
    const V3<T> s = normalized(p);
    const T den = s.z + xi;
    const T ux = s.x / den;
    const T uy = s.y / den;
    const T r2 = ux * ux + uy * uy;
    const T r4 = r2 * r2;
    const T dr = (1.0 + k1 * r2 + k2 * r4);
    const T x2 = ux * ux;
    const T y2 = uy * uy;
    const T xy = ux * uy;
    const T nx = ux * dr + 2.0 * p1 * xy + p2 * (r2 + 2.0 * x2);
    const T ny = uy * dr + p1 * (r2 + 2.0 * y2) + 2.0 * p2 * xy;
    return {intrinsic[0] * nx + intrinsic[2], intrinsic[1] * ny + intrinsic[3]};
  }
};

// equirectangular.hpp:14-28 (intrinsic = [W, H])
struct EquirectProjection {
  static constexpr int num_intrinsic = 2;
  static constexpr int num_distortion = 0;
  template <typename T>
  V2<T> operator()(const double* intrinsic, const double* /*distortion*/, const V3<T>& p) const {
    if (p.x * p.x + p.y * p.y + p.z * p.z < 1e-3) {
      return {T(intrinsic[0] / 2), T(intrinsic[1] / 2)};
    }
    const V3<T> b = normalized(p);
    const T lat = -asin(b.y);
    const T lon = atan2(b.x, b.z);
    const T x = intrinsic[0] * (0.5 + lon / (2.0 * M_PI));
    const T y = intrinsic[1] * (0.5 - lat / M_PI);
    return {x, y};
  }
};

// atan.hpp:14-39
struct AtanProjection {
  static constexpr int num_intrinsic = 4;
  static constexpr int num_distortion = 1;
  template <typename T>
  V2<T> operator()(const double* intrinsic, const double* distortion, const V3<T>& p) const {
    const T px = p.x / p.z;
    const T py = p.y / p.z;
    const double d0 = distortion[0];
    const T r = sqrt(px * px + py * py);
    T dx = px, dy = py;
    if (!(r < 1e-3 || d0 < 1e-7)) {
      const double d1 = 1.0 / d0;
      const double d2 = 2.0 * std::tan(d0 / 2.0);
      const T factor = d1 * atan(r * d2) / r;
      dx = factor * px;
      dy = factor * py;
    }
    return {intrinsic[0] * dx + intrinsic[2], intrinsic[1] * dy + intrinsic[3]};
  }
};

// rational_polynomial.hpp:11-58 (storage order k1 k2 p1 p2 k3 k4 k5 k6)
struct RationalProjection {
  static constexpr int num_intrinsic = 4;
  static constexpr int num_distortion = 8;
  template <typename T>
  V2<T> operator()(const double* intrinsic, const double* distortion, const V3<T>& p) const {
    const T px = p.x / p.z;
    const T py = p.y / p.z;
    const double k1 = distortion[0], k2 = distortion[1], p1 = distortion[2], p2 = distortion[3];
    const double k3 = distortion[4], k4 = distortion[5], k5 = distortion[6], k6 = distortion[7];
    const T x2 = px * px;
    const T y2 = py * py;
    const T r2 = x2 + y2;
    const T r4 = r2 * r2;
    const T r6 = r2 * r4;
    const T numerator = 1.0 + k1 * r2 + k2 * r4 + k3 * r6;
    const T denominator = 1.0 + k4 * r2 + k5 * r4 + k6 * r6;
    const T r_coeff = denominator > 1e-8 ? numerator / denominator : numerator;
    const T t_coeff1 = 2.0 * px * py;
    const T t_coeff2 = r2 + 2.0 * x2;
    const T t_coeff3 = r2 + 2.0 * y2;
    const T x = r_coeff * px + p1 * t_coeff1 + p2 * t_coeff2;
    const T y = r_coeff * py + p1 * t_coeff3 + p2 * t_coeff1;
    return {intrinsic[0] * x + intrinsic[2], intrinsic[1] * y + intrinsic[3]};
  }
};

// generic_camera_base.hpp:18-41
class CameraBase {
public:
  virtual ~CameraBase() {}
  virtual V2<double> project(const V3<double>& p) const = 0;
  virtual V2<double> operator()(const V3<double>& p) const = 0;
  virtual V2<Jet7> operator()(const V3<Jet7>& p) const = 0;
};

// generic_camera.hpp:16-38
template <typename Projection>
class Camera : public CameraBase {
public:
  Camera(const std::vector<double>& intrinsic, const std::vector<double>& distortion) : intrinsic(intrinsic), distortion(distortion) {
    this->distortion.resize(8, 0.0);  // so .data() is never null for the 0-parameter model
  }
  V2<double> project(const V3<double>& p) const override { return (*this)(p); }
  V2<double> operator()(const V3<double>& p) const override {
    Projection proj;
    return proj(intrinsic.data(), distortion.data(), p);
  }
  V2<Jet7> operator()(const V3<Jet7>& p) const override {
    Projection proj;
    return proj(intrinsic.data(), distortion.data(), p);
  }

private:
  std::vector<double> intrinsic;
  std::vector<double> distortion;
};

// create_camera.cpp:17-32
template <typename Projection>
std::shared_ptr<const CameraBase> create_camera_t(const std::vector<double>& intrinsics, const std::vector<double>& distortion_coeffs) {
  if (static_cast<int>(intrinsics.size()) != Projection::num_intrinsic) {
    return nullptr;
  }
  std::vector<double> dist(Projection::num_distortion, 0.0);
  for (size_t i = 0; i < std::min(distortion_coeffs.size(), dist.size()); i++) {
    dist[i] = distortion_coeffs[i];
  }
  return std::make_shared<Camera<Projection>>(intrinsics, dist);
}

// create_camera.cpp:34-51
inline std::shared_ptr<const CameraBase>
create_camera(const std::string& camera_model, const std::vector<double>& intrinsics, const std::vector<double>& distortion_coeffs) {
  if (camera_model == "plumb_bob") {
    return create_camera_t<PinholeProjection>(intrinsics, distortion_coeffs);
  } else if (camera_model == "fisheye" || camera_model == "equidistant") {
    return create_camera_t<FisheyeProjection>(intrinsics, distortion_coeffs);
  } else if (camera_model == "atan") {
    return create_camera_t<AtanProjection>(intrinsics, distortion_coeffs);
  } else if (camera_model == "omnidir") {
    return create_camera_t<OmnidirProjection>(intrinsics, distortion_coeffs);
  } else if (camera_model == "equirectangular") {
    return create_camera_t<EquirectProjection>(intrinsics, distortion_coeffs);
  } else if (camera_model == "rational_polynomial") {
    return create_camera_t<RationalProjection>(intrinsics, distortion_coeffs);
  }
  return nullptr;
}

}  // namespace oracle
