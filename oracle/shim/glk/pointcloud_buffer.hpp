// TEST INFRASTRUCTURE ONLY -- stand-in for Iridescence's glk::PointCloudBuffer / glk::colormapf (OpenGL side of
// the viewer, not in the reference tree).  add_color() keeps the colours so the driver can read what
// PointsColorUpdater::update hands to the viewer (points_color_updater.cpp:60).
#pragma once
#include <vector>

#include <Eigen/Core>

namespace glk {

enum class COLORMAP { TURBO };
Eigen::Vector4f colormapf(COLORMAP map, float x);  // defined in oracle/ref_driver.cpp (the real table is Iridescence's)

class PointCloudBuffer {
public:
  template <typename P>
  PointCloudBuffer(const P*, int) {}
  void add_color(const std::vector<Eigen::Vector4f>& colors) { last_colors = colors; }
  std::vector<Eigen::Vector4f> last_colors;
};

}  // namespace glk
