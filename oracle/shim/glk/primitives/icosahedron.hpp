// TEST INFRASTRUCTURE ONLY -- stand-in for Iridescence's glk::Icosahedron (not in the reference tree); only the
// members points_color_updater.cpp:14-21 names.  The icosahedron constructor is compiled, never exercised.
#pragma once
#include <vector>

#include <Eigen/Core>

namespace glk {
class Icosahedron {
public:
  void subdivide() {}
  void spherize() {}
  std::vector<Eigen::Vector3f> vertices;
};
}  // namespace glk
