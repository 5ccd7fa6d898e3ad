// points_color_updater.hpp -- drop-in for include/vlcal/common/points_color_updater.hpp +
// src/vlcal/common/points_color_updater.cpp: the per-frame recolouring of the cloud under a candidate
// extrinsic, evaluated by the HIP engine (nidreg_colorizer_*).  Differences from the reference class,
// both forced by dependencies that are not part of its tree:
//   * the OpenGL side (glk::PointCloudBuffer, guik viewer invoke) stays with the caller: update()
//     returns the colours (RGBA float per point) instead of pushing them to the viewer;
//   * intensity colours (glk::colormapf(TURBO, intensity), points_color_updater.cpp:33-35) and
//     min_nz = cos(estimate_camera_fov(proj, size) + 0.5 deg) (:12, :28) are passed in -- the caller
//     has both (estimate_fov.cpp is unchanged by the integration).
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "camera.hpp"
#ifdef NIDREG_WITH_REFERENCE_DEPS
#include <Eigen/Geometry>
#include <opencv2/core.hpp>
#include <vlcal/common/frame_cpu.hpp>
#endif

namespace vlcal {

class PointsColorUpdater {
public:
  // intensity_colors: 4 floats (RGBA) per point, or nullptr for (1,1,1,1) (the icosahedron constructor, :24)
  PointsColorUpdater(const camera::GenericCameraBase::ConstPtr& proj, const cv::Mat& image /* CV_8UC1 */, const Frame::ConstPtr& points, const float* intensity_colors,
                     const double min_nz, const int device_id = 0)
  : proj(proj), min_nz(min_nz), image(image), points(points) {
    nidreg_colorizer* c = nullptr;
    const camera::NidregCameraParams cp = camera::nidreg_camera_params(*proj);
    const int rc = nidreg_colorizer_create(device_id, cp.model_id, cp.intrinsics, cp.distortion, image.cols, image.rows, image.data,
                                           static_cast<int64_t>(image.step), static_cast<int64_t>(points->size()), reinterpret_cast<const double*>(points->points),
                                           sizeof(points->points[0]), intensity_colors, min_nz, &c);
    if (rc != NIDREG_OK) throw std::runtime_error(std::string("vlcal::PointsColorUpdater: ") + nidreg_last_error());
    handle = std::shared_ptr<nidreg_colorizer>(c, &nidreg_colorizer_destroy);
    colors.resize(points->size() * 4);
  }

  // points_color_updater.cpp:37-61; returns size() x RGBA floats, valid until the next update
  const std::vector<float>& update(const Eigen::Isometry3d& T_camera_lidar, const double blend_weight) {
    double T[16];
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) T[r * 4 + c] = r < 3 ? T_camera_lidar(r, c) : (c == 3 ? 1.0 : 0.0);
    if (nidreg_colorizer_update(handle.get(), T, blend_weight, colors.data()) != NIDREG_OK)
      throw std::runtime_error(std::string("vlcal::PointsColorUpdater::update: ") + nidreg_last_error());
    return colors;
  }

public:
  camera::GenericCameraBase::ConstPtr proj;
  double min_nz;
  cv::Mat image;
  Frame::ConstPtr points;
  std::vector<float> colors;

private:
  std::shared_ptr<nidreg_colorizer> handle;
};

}  // namespace vlcal
