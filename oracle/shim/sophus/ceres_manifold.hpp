// TEST INFRASTRUCTURE ONLY -- stand-in for <sophus/ceres_manifold.hpp>: the type only
// (visual_camera_calibration.cpp:216 hands `new Sophus::Manifold<Sophus::SE3>()` to ceres::GradientProblem).
#pragma once
#include <ceres/ceres.h>
#include <sophus/se3.hpp>

namespace Sophus {
template <template <typename, int> class LieGroup>
class Manifold : public ceres::Manifold {};
}  // namespace Sophus
