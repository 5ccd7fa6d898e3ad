// view_culling.hpp -- drop-in for include/vlcal/calib/view_culling.hpp + src/vlcal/calib/view_culling.cpp:
// vlcal::ViewCullingParams / vlcal::ViewCulling with the reference's constructor shape, evaluated by the
// HIP engine (nidreg_view_culling: FoV gate on the normalised 4-vector, in-image test, float depth buffer,
// +0.1 m keep threshold; the surviving index list is identical to the CPU loop's).
// Differences, both because FrameCPU / estimate_fov.cpp stay with the reference:
//   * min_z = cos(estimate_camera_fov(proj, image_size)) (view_culling.cpp:17) is passed in;
//   * cull_indices() returns the surviving indices; the reference's cull() is then
//     `sample(points, cull_indices(points, T))` (view_culling.cpp:31-32, frame_cpu.cpp).
// When the cost object is built right after culling (visual_camera_calibration.cpp:201-206), prefer
// vlcal::DeviceCloud + the second NIDCost constructor (nid_cost.hpp): culling then never leaves the GPU.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "camera.hpp"
#ifdef NIDREG_WITH_REFERENCE_DEPS
#include <cmath>

#include <Eigen/Geometry>
#include <vlcal/common/estimate_fov.hpp>
#include <vlcal/common/frame.hpp>
#include <vlcal/common/frame_cpu.hpp>
#endif

namespace vlcal {

struct ViewCullingParams {
public:
  ViewCullingParams() { enable_depth_buffer_culling = true; }
  bool enable_depth_buffer_culling;  ///< If true, perform depth-buffer-based hidden points removal
};

class ViewCulling {
public:
  ViewCulling(const camera::GenericCameraBase::ConstPtr& proj, const int width, const int height, const ViewCullingParams& params, const double min_z, const int device_id = 0)
  : params(params), proj(proj), width(width), height(height), min_z(min_z), device_id(device_id) {}
#ifdef NIDREG_WITH_REFERENCE_DEPS
  // the reference's own signatures (view_culling.hpp:23-27): min_z from the reference's estimate_camera_fov and
  // cull() = sample(points, surviving indices), so visual_camera_calibration.cpp:73,78,193,201 compile unchanged
  ViewCulling(const camera::GenericCameraBase::ConstPtr& proj, const Eigen::Vector2i& image_size, const ViewCullingParams& params)
  : ViewCulling(proj, image_size[0], image_size[1], params, std::cos(estimate_camera_fov(proj, image_size))) {}
  FrameCPU::Ptr cull(const Frame::ConstPtr& points, const Eigen::Isometry3d& T_camera_lidar) const { return sample(points, cull_indices(points, T_camera_lidar)); }
#endif
  ~ViewCulling() {}

  std::vector<int> cull_indices(const Frame::ConstPtr& points, const Eigen::Isometry3d& T_camera_lidar) const {
    double T[16];
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) T[r * 4 + c] = r < 3 ? T_camera_lidar(r, c) : (c == 3 ? 1.0 : 0.0);
    std::vector<int32_t> idx(points->size());
    const camera::NidregCameraParams cp = camera::nidreg_camera_params(*proj);
    const int64_t n = nidreg_view_culling(cp.model_id, cp.intrinsics, cp.distortion, device_id, width, height, min_z,
                                          params.enable_depth_buffer_culling ? 1 : 0, reinterpret_cast<const double*>(points->points), sizeof(points->points[0]),
                                          static_cast<int64_t>(points->size()), T, idx.data());
    if (n < 0) throw std::runtime_error(std::string("vlcal::ViewCulling: ") + nidreg_last_error());
    return std::vector<int>(idx.begin(), idx.begin() + n);
  }

private:
  const ViewCullingParams params;
  const camera::GenericCameraBase::ConstPtr proj;
  const int width, height;
  const double min_z;
  const int device_id;
};

}  // namespace vlcal
