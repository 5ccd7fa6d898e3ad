// TEST INFRASTRUCTURE ONLY -- shadows the reference's include/vlcal/common/frame_cpu.hpp in the
// oracle/_ref build.  The real FrameCPU (frame_cpu.hpp:1-206, frame_cpu.cpp) drags in PCL / nanoflann
// features that have nothing to do with the NID path; the hot-path sources only need a Frame that owns
// its points / intensities and `sample(frame, indices)` (frame_cpu.cpp: copy of the selected points and
// intensities).  `indices` is an addition: it lets the driver read which points ViewCulling::cull kept.
#pragma once
#include <memory>
#include <vector>

#include <vlcal/common/frame.hpp>

namespace vlcal {

struct FrameCPU : public Frame {
  using Ptr = std::shared_ptr<FrameCPU>;
  using ConstPtr = std::shared_ptr<const FrameCPU>;
  FrameCPU() {}
  FrameCPU(const double* xyzw, const double* inten, size_t n) : points_storage(n), intensities_storage(inten, inten + n) {
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 4; k++) points_storage[i][k] = xyzw[4 * i + k];
    bind();
  }
  // frame_cpu.hpp:29: from a vector of D-vectors (w = 1 for D = 3)
  template <typename T, int D, typename Alloc>
  FrameCPU(const std::vector<Eigen::Matrix<T, D, 1>, Alloc>& pts) : points_storage(pts.size()) {
    for (size_t i = 0; i < pts.size(); i++) {
      for (int k = 0; k < 4; k++) points_storage[i][k] = k < D ? double(pts[i][k]) : 1.0;
    }
    bind();
  }
  void bind() {
    num_points = points_storage.size();
    points = points_storage.data();
    intensities = intensities_storage.empty() ? nullptr : intensities_storage.data();
  }
  std::vector<Eigen::Vector4d> points_storage;
  std::vector<double> intensities_storage;
  std::vector<int> indices;  // set by sample()
};

inline FrameCPU::Ptr sample(const Frame::ConstPtr& frame, const std::vector<int>& indices) {
  auto out = std::make_shared<FrameCPU>();
  out->points_storage.resize(indices.size());
  out->intensities_storage.resize(indices.size());
  for (size_t i = 0; i < indices.size(); i++) {
    out->points_storage[i] = frame->points[indices[i]];
    out->intensities_storage[i] = frame->intensities ? frame->intensities[indices[i]] : 0.0;
  }
  out->indices = indices;
  out->bind();
  return out;
}

}  // namespace vlcal
