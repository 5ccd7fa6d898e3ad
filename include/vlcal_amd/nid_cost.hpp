// nid_cost.hpp -- drop-in for the reference's include/vlcal/costs/nid_cost.hpp: same class name,
// constructor and functor signature (nid_cost.hpp:23, :36-37), so that
// src/vlcal/calib/visual_camera_calibration.cpp:206,215 (MultiNIDCost +
// ceres::AutoDiffFirstOrderFunction<MultiNIDCost, 7>) compiles unchanged.  The body forwards to the
// HIP engine through the C ABI (include/nidreg.h): T = double -> cost only, T = Jet<double,7> ->
// cost + the ambient 7-gradient written into residual[0].v.
#pragma once
#include <cmath>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <type_traits>

#include "camera.hpp"
#ifdef NIDREG_WITH_REFERENCE_DEPS
#include <opencv2/core.hpp>
#include <vlcal/common/frame.hpp>
#endif

namespace vlcal {

template <typename T>
inline double get_real(const T& x) { return x.a; }
template <>
inline double get_real(const double& x) { return x; }

// One LiDAR cloud resident on the GPU (upload once per pair).  Optional extension over the reference:
// with it, `ViewCulling::cull -> new NIDCost` of visual_camera_calibration.cpp:201-206 runs entirely on
// the device (see the second NIDCost constructor).
class DeviceCloud {
public:
  DeviceCloud(const Frame::ConstPtr& points, const int device_id = 0) {
    nidreg_cloud* c = nullptr;
    if (nidreg_cloud_create(device_id, reinterpret_cast<const double*>(points->points), sizeof(points->points[0]), points->intensities, static_cast<int64_t>(points->size()), &c) != NIDREG_OK)
      throw std::runtime_error(std::string("vlcal::DeviceCloud: ") + nidreg_last_error());
    cloud = std::shared_ptr<nidreg_cloud>(c, &nidreg_cloud_destroy);
  }
  const nidreg_cloud* get() const { return cloud.get(); }

private:
  std::shared_ptr<nidreg_cloud> cloud;
};

class NIDCost {
public:
  NIDCost(const camera::GenericCameraBase::ConstPtr& proj, const cv::Mat& normalized_image, const Frame::ConstPtr& points, const int bins = 16, const int device_id = 0,
          const int precision = NIDREG_PREC_FP64)
  {
    nidreg_desc d{};
    d.struct_size = sizeof(nidreg_desc);
    d.device_id = device_id;
    d.model_id = proj->nidreg_model_id();
    d.mode = NIDREG_MODE_SPLINE;
    d.precision = precision;
    d.bins = bins;
    for (int i = 0; i < 5; i++) d.intrinsics[i] = proj->nidreg_intrinsics()[i];
    for (int i = 0; i < 8; i++) d.distortion[i] = proj->nidreg_distortion()[i];
    d.width = normalized_image.cols;
    d.height = normalized_image.rows;
    d.image_dtype = NIDREG_IMAGE_F64;  // CV_64FC1, made by convertTo(..., 1/255) (visual_camera_calibration.cpp:204)
    d.image = normalized_image.data;
    d.image_row_stride = static_cast<int64_t>(normalized_image.step);
    d.num_points = static_cast<int64_t>(points->size());
    d.points = reinterpret_cast<const double*>(points->points);
    d.point_stride = sizeof(points->points[0]);
    d.intensities = points->intensities;
    nidreg_handle* h = nullptr;
    if (nidreg_create(&d, &h) != NIDREG_OK) throw std::runtime_error(std::string("vlcal::NIDCost: ") + nidreg_last_error());
    handle = std::shared_ptr<nidreg_handle>(h, &nidreg_destroy);
  }

  // cull (at T_camera_lidar, row-major 4x4; nullptr = no culling) + build on the device
  NIDCost(const camera::GenericCameraBase::ConstPtr& proj, const cv::Mat& normalized_image, const DeviceCloud& cloud, const double* T_camera_lidar, const double min_z,
          const bool enable_depth_buffer_culling, const int bins = 16, const int device_id = 0, const int precision = NIDREG_PREC_FP64) {
    nidreg_desc d{};
    d.struct_size = sizeof(nidreg_desc);
    d.device_id = device_id;
    d.model_id = proj->nidreg_model_id();
    d.mode = NIDREG_MODE_SPLINE;
    d.precision = precision;
    d.bins = bins;
    for (int i = 0; i < 5; i++) d.intrinsics[i] = proj->nidreg_intrinsics()[i];
    for (int i = 0; i < 8; i++) d.distortion[i] = proj->nidreg_distortion()[i];
    d.width = normalized_image.cols;
    d.height = normalized_image.rows;
    d.image_dtype = NIDREG_IMAGE_F64;
    d.image = normalized_image.data;
    d.image_row_stride = static_cast<int64_t>(normalized_image.step);
    nidreg_handle* h = nullptr;
    if (nidreg_create_from_cloud(&d, cloud.get(), T_camera_lidar, min_z, enable_depth_buffer_culling ? 1 : 0, &h) != NIDREG_OK)
      throw std::runtime_error(std::string("vlcal::NIDCost: ") + nidreg_last_error());
    handle = std::shared_ptr<nidreg_handle>(h, &nidreg_destroy);
  }

  template <typename T>
  bool operator()(const T* T_camera_lidar_params, T* residual) const {
    double se3[7];
    for (int i = 0; i < 7; i++) se3[i] = get_real(T_camera_lidar_params[i]);
    double cost = 0.0;
    if constexpr (std::is_same<T, double>::value) {
      const int rc = nidreg_eval(handle.get(), se3, &cost, nullptr);
      if (rc < 0) throw std::runtime_error(std::string("vlcal::NIDCost: ") + nidreg_last_error());
      if (rc == NIDREG_FALSE) return false;  // non-finite NID (nid_cost.hpp:98-102)
      residual[0] = cost;
    } else {
      double grad[7];
      const int rc = nidreg_eval(handle.get(), se3, &cost, grad);
      if (rc < 0) throw std::runtime_error(std::string("vlcal::NIDCost: ") + nidreg_last_error());
      if (rc == NIDREG_FALSE) return false;
      // chain rule through whatever partials the caller seeded: residual.v = sum_k grad[k] * params[k].v
      T r = T_camera_lidar_params[0];
      r.a = cost;
      constexpr int N = sizeof(r.v) / sizeof(r.v[0]);
      for (int j = 0; j < N; j++) {
        double s = 0.0;
        for (int k = 0; k < 7; k++) s += grad[k] * T_camera_lidar_params[k].v[j];
        r.v[j] = s;
      }
      residual[0] = r;
    }
    return true;
  }

  nidreg_handle* native_handle() const { return handle.get(); }

private:
  std::shared_ptr<nidreg_handle> handle;
};

}  // namespace vlcal
