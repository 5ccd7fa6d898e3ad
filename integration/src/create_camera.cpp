// create_camera.cpp -- drop-in for the reference's src/camera/create_camera.cpp (compile THIS file instead of that
// one; no reference header is modified).  Same contract (create_camera.cpp:17-51): the model string selects the
// projection, a wrong intrinsic count or an unknown model prints an error and returns nullptr, the distortion vector is
// zero padded / truncated to the model's count.  The cameras it returns are the reference's own
// camera::GenericCamera<Projection> objects -- project() and the Jet overload are untouched, culling / viewer /
// initial guess keep calling them on the CPU -- which additionally implement camera::NidregCameraInfo
// (include/vlcal_amd/camera.hpp), so that the GPU cost functions can read the model id and the padded parameters the
// reference keeps private (generic_camera.hpp:35-37).
#include <algorithm>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <ceres/jet.h>

#include <camera/atan.hpp>
#include <camera/create_camera.hpp>
#include <camera/equirectangular.hpp>
#include <camera/fisheye.hpp>
#include <camera/generic_camera.hpp>
#include <camera/omnidir.hpp>
#include <camera/pinhole.hpp>
#include <camera/rational_polynomial.hpp>

#include <vlcal_amd/camera.hpp>
#include <vlcal_amd/reference_camera_ids.hpp>

namespace camera {
namespace {

template <typename Projection>
class NidregGenericCamera : public GenericCamera<Projection>, public NidregCameraInfo {
public:
  NidregGenericCamera(const Eigen::VectorXd& intrinsic, const Eigen::VectorXd& distortion) : GenericCamera<Projection>(intrinsic, distortion) {
    for (int i = 0; i < 5; i++) intr[i] = i < intrinsic.size() ? intrinsic[i] : 0.0;
    for (int i = 0; i < 8; i++) dist[i] = i < distortion.size() ? distortion[i] : 0.0;
  }
  int nidreg_model_id() const override { return NidregModelId<Projection>::value; }
  const double* nidreg_intrinsics() const override { return intr; }
  const double* nidreg_distortion() const override { return dist; }

private:
  double intr[5];
  double dist[8];
};

template <typename Projection>
GenericCameraBase::ConstPtr make(const std::vector<double>& intrinsics, const std::vector<double>& distortion_coeffs) {
  const size_t want_intr = CameraModelTraits<Projection>::num_intrinsic_params;
  const size_t want_dist = CameraModelTraits<Projection>::num_distortion_params;
  if (intrinsics.size() != want_intr) {
    std::cerr << "error: num of intrinsic parameters mismatch!!" << std::endl;
    return nullptr;
  }
  Eigen::VectorXd intr(static_cast<int>(want_intr)), dist(static_cast<int>(want_dist));
  for (size_t i = 0; i < want_intr; i++) intr[static_cast<int>(i)] = intrinsics[i];
  for (size_t i = 0; i < want_dist; i++) dist[static_cast<int>(i)] = i < distortion_coeffs.size() ? distortion_coeffs[i] : 0.0;
  return std::make_shared<NidregGenericCamera<Projection>>(intr, dist);
}

}  // namespace

GenericCameraBase::ConstPtr create_camera(const std::string& camera_model, const std::vector<double>& intrinsics, const std::vector<double>& distortion_coeffs) {
  struct Entry {
    const char* name;
    GenericCameraBase::ConstPtr (*factory)(const std::vector<double>&, const std::vector<double>&);
  };
  static const Entry table[] = {
    {"plumb_bob", &make<PinholeProjection>},           {"fisheye", &make<FisheyeProjection>},
    {"equidistant", &make<FisheyeProjection>},         {"atan", &make<ATANProjection>},
    {"omnidir", &make<OmnidirectionalProjection>},     {"equirectangular", &make<EquirectangularProjection>},
    {"rational_polynomial", &make<RationalPolynomialProjection>},
  };
  for (const Entry& e : table)
    if (camera_model == e.name) return e.factory(intrinsics, distortion_coeffs);
  std::cerr << "error: unknown camera model " << camera_model << std::endl;
  return nullptr;
}

}  // namespace camera
