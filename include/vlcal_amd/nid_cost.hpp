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
#include <string>
#include <type_traits>
#include <vector>

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
  // device_ids (optional extension): more than one entry shards the pair's points over those GPUs inside the library
  // (single process; the cloud is cut along the histogram column and every GPU stores its columns of the integer histogram
  // into every other GPU's replica: one GPU-to-GPU exchange per evaluation).  Without it the environment variable NIDREG_DEVICES
  // does the same for a caller that cannot be changed (visual_camera_calibration.cpp:206).
  NIDCost(const camera::GenericCameraBase::ConstPtr& proj, const cv::Mat& normalized_image, const Frame::ConstPtr& points, const int bins = 16, const int device_id = 0,
          const int precision = NIDREG_PREC_FP64, const std::vector<int>& device_ids = std::vector<int>())
  {
    nidreg_desc d{};
    d.num_devices = static_cast<int32_t>(device_ids.size() > NIDREG_MAX_DEVICES ? NIDREG_MAX_DEVICES : device_ids.size());
    for (int i = 0; i < d.num_devices; i++) d.device_ids[i] = device_ids[static_cast<size_t>(i)];
    d.struct_size = sizeof(nidreg_desc);
    d.device_id = device_id;
    const camera::NidregCameraParams cp = camera::nidreg_camera_params(*proj);
    d.model_id = cp.model_id;
    d.mode = NIDREG_MODE_SPLINE;
    d.precision = precision;
    d.bins = bins;
    for (int i = 0; i < 5; i++) d.intrinsics[i] = cp.intrinsics[i];
    for (int i = 0; i < 8; i++) d.distortion[i] = cp.distortion[i];
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
    const camera::NidregCameraParams cp = camera::nidreg_camera_params(*proj);
    d.model_id = cp.model_id;
    d.mode = NIDREG_MODE_SPLINE;
    d.precision = precision;
    d.bins = bins;
    for (int i = 0; i < 5; i++) d.intrinsics[i] = cp.intrinsics[i];
    for (int i = 0; i < 8; i++) d.distortion[i] = cp.distortion[i];
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
      if (rc != NIDREG_OK) return reject(rc);  // non-finite NID (nid_cost.hpp:98-102), or an engine error
      residual[0] = cost;
    } else {
      double grad[7];
      const int rc = nidreg_eval(handle.get(), se3, &cost, grad);
      if (rc != NIDREG_OK) return reject(rc);
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
  // text of the last engine error this functor swallowed (empty = none)
  const std::string& last_error() const { return *error; }

private:
  // The functor contract is `return false` (an invalid step for Ceres' line search): the reference calls it inside
  // `#pragma omp parallel for` (visual_camera_calibration.cpp:161), where an escaping exception is std::terminate.
  // An engine error (rc < 0) is therefore reported on stderr, kept in last_error(), and rejected like a non-finite NID.
  bool reject(const int rc) const {
    if (rc < 0) {
      *error = nidreg_last_error();
      std::cerr << "vlcal::NIDCost: " << *error << std::endl;
    }
    return false;
  }
  std::shared_ptr<std::string> error = std::make_shared<std::string>();

private:
  std::shared_ptr<nidreg_handle> handle;
};

}  // namespace vlcal
