// cost_calculator_nid.hpp -- drop-in for include/vlcal/calib/cost_calculator{,_nid}.hpp +
// src/vlcal/calib/cost_calculator_nid.cpp: vlcal::CostCalculator / NIDCostParams / CostCalculatorNID with
// the reference's signatures (cost_calculator.hpp:9-18, cost_calculator_nid.hpp:9-28), evaluated by
// the HIP engine.  max_fov comes from estimate_camera_fov (estimate_fov.cpp:36-51), which the
// reference computes in the constructor (cost_calculator_nid.cpp:13-17); here it is passed in or
// estimated by the caller-supplied functor so this header stays free of the optimiser dependency.
#pragma once
#include <iostream>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>

#include "camera.hpp"
#ifdef NIDREG_WITH_REFERENCE_DEPS
#include <Eigen/Geometry>
#include <vlcal/common/estimate_fov.hpp>
#include <vlcal/common/visual_lidar_data.hpp>
#endif

namespace vlcal {

class CostCalculator {
public:
  using Ptr = std::shared_ptr<CostCalculator>;
  using ConstPtr = std::shared_ptr<const CostCalculator>;
  CostCalculator() {}
  virtual ~CostCalculator() {}
  virtual double calculate(const Eigen::Isometry3d& T_camera_lidar) = 0;
};

struct NIDCostParams {
  NIDCostParams() : bins(16) {}
  ~NIDCostParams() {}
  int bins;
};

class CostCalculatorNID : public CostCalculator {
public:
#ifdef NIDREG_WITH_REFERENCE_DEPS
  // the reference's own constructor signature (cost_calculator_nid.hpp:20, cost_calculator_nid.cpp:13-17):
  // max_fov from the reference's estimate_camera_fov, so visual_camera_calibration.cpp:84 compiles unchanged
  CostCalculatorNID(const camera::GenericCameraBase::ConstPtr& proj, const VisualLiDARData::ConstPtr& data, const NIDCostParams& params = NIDCostParams())
  : CostCalculatorNID(proj, data, params, estimate_camera_fov(proj, {data->image.cols, data->image.rows})) {}
#endif
  CostCalculatorNID(const camera::GenericCameraBase::ConstPtr& proj, const VisualLiDARData::ConstPtr& data, const NIDCostParams& params, const double max_fov,
                    const int device_id = 0, const int precision = NIDREG_PREC_FP64)
  {
    nidreg_desc d{};
    d.struct_size = sizeof(nidreg_desc);
    d.device_id = device_id;
    const camera::NidregCameraParams cp = camera::nidreg_camera_params(*proj);
    d.model_id = cp.model_id;
    d.mode = NIDREG_MODE_NEAREST;
    d.precision = precision;
    d.bins = params.bins;
    for (int i = 0; i < 5; i++) d.intrinsics[i] = cp.intrinsics[i];
    for (int i = 0; i < 8; i++) d.distortion[i] = cp.distortion[i];
    d.width = data->image.cols;
    d.height = data->image.rows;
    d.image_dtype = NIDREG_IMAGE_U8;
    d.image = data->image.data;
    d.image_row_stride = static_cast<int64_t>(data->image.step);
    d.num_points = static_cast<int64_t>(data->points->size());
    d.points = reinterpret_cast<const double*>(data->points->points);
    d.point_stride = sizeof(data->points->points[0]);
    d.intensities = data->points->intensities;
    d.max_fov = max_fov;
    nidreg_handle* h = nullptr;
    if (nidreg_create(&d, &h) != NIDREG_OK) throw std::runtime_error(std::string("vlcal::CostCalculatorNID: ") + nidreg_last_error());
    handle = std::shared_ptr<nidreg_handle>(h, &nidreg_destroy);
  }
  ~CostCalculatorNID() override {}

  double calculate(const Eigen::Isometry3d& T_camera_lidar) override {
    double T[16];
    for (int r = 0; r < 4; r++)
      for (int c = 0; c < 4; c++) T[r * 4 + c] = r < 3 ? T_camera_lidar(r, c) : (c == 3 ? 1.0 : 0.0);  // (r,c) access works for Eigen's column-major storage too
    double cost = 0.0;
    if (nidreg_eval_iso(handle.get(), T, &cost) < 0) {
      // called inside `#pragma omp parallel for` (visual_camera_calibration.cpp:107), where an escaping exception is
      // std::terminate: report, and hand the simplex a value it will never accept
      std::cerr << "vlcal::CostCalculatorNID: " << nidreg_last_error() << std::endl;
      return std::numeric_limits<double>::max();
    }
    return cost;
  }

private:
  std::shared_ptr<nidreg_handle> handle;
};

}  // namespace vlcal
