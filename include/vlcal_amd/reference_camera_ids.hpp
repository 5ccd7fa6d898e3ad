// reference_camera_ids.hpp -- NIDREG_MODEL_* of each of the reference's projection functors
// (include/camera/{pinhole,fisheye,omnidir,equirectangular,atan,rational_polynomial}.hpp), for the
// accessors integration/reference_camera.patch adds to camera::GenericCamera<Projection>
// (include/camera/generic_camera.hpp).  Same assignment as camera::create_camera's strings
// (src/camera/create_camera.cpp:34-51) and nidreg_model_from_name().
#pragma once
#include "../nidreg.h"

namespace camera {

struct PinholeProjection;
struct FisheyeProjection;
struct OmnidirectionalProjection;
struct EquirectangularProjection;
struct ATANProjection;
struct RationalPolynomialProjection;

template <typename Projection>
struct NidregModelId;
template <>
struct NidregModelId<PinholeProjection> {
  static constexpr int value = NIDREG_MODEL_PLUMB_BOB;
};
template <>
struct NidregModelId<FisheyeProjection> {
  static constexpr int value = NIDREG_MODEL_FISHEYE;
};
template <>
struct NidregModelId<OmnidirectionalProjection> {
  static constexpr int value = NIDREG_MODEL_OMNIDIR;
};
template <>
struct NidregModelId<EquirectangularProjection> {
  static constexpr int value = NIDREG_MODEL_EQUIRECTANGULAR;
};
template <>
struct NidregModelId<ATANProjection> {
  static constexpr int value = NIDREG_MODEL_ATAN;
};
template <>
struct NidregModelId<RationalPolynomialProjection> {
  static constexpr int value = NIDREG_MODEL_RATIONAL_POLYNOMIAL;
};

}  // namespace camera
