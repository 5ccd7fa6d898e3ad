// generate_lidar_image.hpp -- drop-in for include/vlcal/preprocess/generate_lidar_image.hpp +
// src/vlcal/preprocess/generate_lidar_image.cpp:7-41, evaluated by the HIP engine.  Same arguments plus
// min_z = cos(estimate_camera_fov(proj, image_size)), which the reference computes inside (:10-11)
// with estimate_fov.cpp (unchanged by the integration, so the caller passes it).
// Returns {intensity_image CV_64FC1, index_image CV_32SC1}; identical to the CPU loop's output.
#pragma once
#include <stdexcept>
#include <string>
#include <utility>

#include "camera.hpp"
#ifdef NIDREG_WITH_REFERENCE_DEPS
#include <Eigen/Geometry>
#include <opencv2/core.hpp>
#include <vlcal/common/frame.hpp>
#endif

namespace vlcal {

inline void generate_lidar_image(const camera::GenericCameraBase::ConstPtr& proj, const int width, const int height, const Eigen::Isometry3d& T_camera_lidar, const Frame::ConstPtr& points,
                                 const double min_z, double* intensity_image /* height x width */, int32_t* index_image /* height x width */, const int device_id = 0) {
  double T[16];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) T[r * 4 + c] = r < 3 ? T_camera_lidar(r, c) : (c == 3 ? 1.0 : 0.0);
  const camera::NidregCameraParams cp = camera::nidreg_camera_params(*proj);
  const int rc = nidreg_generate_lidar_image(cp.model_id, cp.intrinsics, cp.distortion, device_id, width, height, min_z,
                                             reinterpret_cast<const double*>(points->points), sizeof(points->points[0]), points->intensities, static_cast<int64_t>(points->size()), T,
                                             intensity_image, index_image);
  if (rc != NIDREG_OK) throw std::runtime_error(std::string("vlcal::generate_lidar_image: ") + nidreg_last_error());
}

#ifdef NIDREG_WITH_REFERENCE_DEPS
// the reference signature (generate_lidar_image.hpp): std::pair<cv::Mat, cv::Mat>
inline std::pair<cv::Mat, cv::Mat> generate_lidar_image(const camera::GenericCameraBase::ConstPtr& proj, const Eigen::Vector2i& image_size, const Eigen::Isometry3d& T_camera_lidar,
                                                        const Frame::ConstPtr& points, const double min_z) {
  cv::Mat intensity_image(image_size[1], image_size[0], CV_64FC1);
  cv::Mat index_image(image_size[1], image_size[0], CV_32SC1);
  generate_lidar_image(proj, image_size[0], image_size[1], T_camera_lidar, points, min_z, intensity_image.ptr<double>(), index_image.ptr<int32_t>());
  return std::make_pair(intensity_image, index_image);
}
#endif

}  // namespace vlcal
