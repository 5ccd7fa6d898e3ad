// camera.hpp -- camera::create_camera / camera::GenericCameraBase with the reference's interface
// (include/camera/generic_camera_base.hpp:18-41, include/camera/create_camera.hpp:14,
// src/camera/create_camera.cpp:17-51), extended by the three accessors a GPU cost function needs
// (the reference hides model and parameters inside GenericCamera<Projection>, generic_camera.hpp:35-37).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../nidreg.h"

// What a GPU cost function needs from a camera: which model it is (NIDREG_MODEL_*) and its parameters, zero padded to
// 5 intrinsics / 8 distortion coefficients.  The reference hides both inside GenericCamera<Projection>
// (generic_camera.hpp:35-37); a camera can expose them in two ways, and nidreg_camera_params() accepts either:
//   * as this side interface, which the cameras made by integration/src/create_camera.cpp (linked INSTEAD of the
//     reference's src/camera/create_camera.cpp; no reference header is touched) implement next to GenericCamera<P>;
//   * as three virtuals on GenericCameraBase itself (integration/reference_camera.patch, which defines
//     NIDREG_CAMERA_PATCHED; the stand-in GenericCameraBase below has them too).
namespace camera {
class NidregCameraInfo {
public:
  virtual ~NidregCameraInfo() {}
  virtual int nidreg_model_id() const = 0;
  virtual const double* nidreg_intrinsics() const = 0;  // 5 doubles, zero padded
  virtual const double* nidreg_distortion() const = 0;  // 8 doubles, zero padded
};
}  // namespace camera

#ifdef NIDREG_WITH_REFERENCE_DEPS
// inside a reference checkout: the reference's own camera classes; CPU project() stays the reference's
#include <Eigen/Core>
#include <ceres/jet.h>
#include <camera/create_camera.hpp>
#include <camera/generic_camera_base.hpp>
#else
#define NIDREG_CAMERA_PATCHED 1
#include "standins.hpp"

namespace camera {

class GenericCameraBase {
public:
  using Ptr = std::shared_ptr<GenericCameraBase>;
  using ConstPtr = std::shared_ptr<const GenericCameraBase>;
  virtual ~GenericCameraBase() {}

  // generic_camera_base.hpp:29,34 -- one point at a time is host work (set-up code such as the reference's
  // estimate_camera_fov, ~240 calls): the device's scalar projection code compiled for the host (NIDREG_DEVICE_HOST);
  // per-point loops over a cloud belong in the kernels, not here
  virtual Eigen::Vector2d project(const Eigen::Vector3d& point_3d) const = 0;
  virtual Eigen::Vector2d operator()(const Eigen::Vector3d& point_3d) const = 0;

  // additions
  virtual int nidreg_model_id() const = 0;
  virtual const double* nidreg_intrinsics() const = 0;  // 5 doubles, zero padded
  virtual const double* nidreg_distortion() const = 0;  // 8 doubles, zero padded
};

class NidregCamera : public GenericCameraBase {
public:
  NidregCamera(int model_id, const std::vector<double>& intrinsics, const std::vector<double>& distortion) : model_id(model_id) {
    for (int i = 0; i < 5; i++) intr[i] = i < int(intrinsics.size()) ? intrinsics[i] : 0.0;
    for (int i = 0; i < 8; i++) dist[i] = i < int(distortion.size()) ? distortion[i] : 0.0;
  }
  Eigen::Vector2d project(const Eigen::Vector3d& p) const override { return (*this)(p); }
  Eigen::Vector2d operator()(const Eigen::Vector3d& p) const override {
    const double p3[3] = {p[0], p[1], p[2]};
    double uv[2] = {0.0, 0.0};
    nidreg_project_model(model_id, intr, dist, NIDREG_DEVICE_HOST, NIDREG_PREC_FP64, p3, 1, uv, nullptr);
    Eigen::Vector2d r;
    r[0] = uv[0];
    r[1] = uv[1];
    return r;
  }
  int nidreg_model_id() const override { return model_id; }
  const double* nidreg_intrinsics() const override { return intr; }
  const double* nidreg_distortion() const override { return dist; }

private:
  int model_id;
  double intr[5];
  double dist[8];
};

// create_camera.cpp:34-51: nullptr on unknown model or intrinsic-count mismatch (:19-22); distortion
// zero-padded / truncated to the model's count (:24-27)
inline GenericCameraBase::ConstPtr create_camera(const std::string& camera_model, const std::vector<double>& intrinsics, const std::vector<double>& distortion_coeffs) {
  int ni = 0, nd = 0;
  const int id = nidreg_model_from_name(camera_model.c_str(), &ni, &nd);
  if (id < 0) return nullptr;
  if (int(intrinsics.size()) != ni) return nullptr;
  std::vector<double> dist(nd, 0.0);
  for (int i = 0; i < nd && i < int(distortion_coeffs.size()); i++) dist[i] = distortion_coeffs[i];
  return std::make_shared<NidregCamera>(id, intrinsics, dist);
}

}  // namespace camera
#endif  // NIDREG_WITH_REFERENCE_DEPS

#include <stdexcept>

namespace camera {
struct NidregCameraParams {
  int model_id;
  const double* intrinsics;  // 5
  const double* distortion;  // 8
};
inline NidregCameraParams nidreg_camera_params(const GenericCameraBase& cam) {
#ifdef NIDREG_CAMERA_PATCHED
  return NidregCameraParams{cam.nidreg_model_id(), cam.nidreg_intrinsics(), cam.nidreg_distortion()};
#else
  const NidregCameraInfo* info = dynamic_cast<const NidregCameraInfo*>(&cam);
  if (!info)
    throw std::runtime_error(
      "nidreg: this camera does not expose its model and parameters -- link integration/src/create_camera.cpp instead of the reference's "
      "src/camera/create_camera.cpp, or apply integration/reference_camera.patch");
  return NidregCameraParams{info->nidreg_model_id(), info->nidreg_intrinsics(), info->nidreg_distortion()};
#endif
}
}  // namespace camera
