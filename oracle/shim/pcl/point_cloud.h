// TEST INFRASTRUCTURE ONLY -- stand-in for the PCL types src/vlcal/common/estimate_fov.cpp:53-92 (estimate_lidar_fov)
// names, so that the file compiles; PCL is not installed.  estimate_lidar_fov is NOT exercised: VoxelGrid::filter
// copies its input and ConvexHull::reconstruct returns it unchanged.
#pragma once
#include <memory>
#include <vector>

#include <Eigen/Core>

namespace pcl {

struct PointXYZ {
  PointXYZ() : v(0.f, 0.f, 0.f) {}
  PointXYZ(float x, float y, float z) : v(x, y, z) {}
  const Eigen::Vector3f& getVector3fMap() const { return v; }
  Eigen::Vector3f v;
};

template <typename PointT>
class PointCloud {
public:
  typedef typename std::vector<PointT>::iterator iterator;
  void resize(size_t n) { points.resize(n); }
  size_t size() const { return points.size(); }
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  iterator erase(iterator a, iterator b) { return points.erase(a, b); }
  const PointT& at(size_t i) const { return points.at(i); }
  std::vector<PointT> points;
};

template <typename T, typename... Args>
std::shared_ptr<T> make_shared(Args&&... args) {
  return std::make_shared<T>(std::forward<Args>(args)...);
}

template <typename PointT>
class VoxelGrid {
public:
  void setLeafSize(float, float, float) {}
  void setInputCloud(const std::shared_ptr<PointCloud<PointT>>& c) { in = c; }
  void filter(PointCloud<PointT>& out) { out = *in; }
  std::shared_ptr<PointCloud<PointT>> in;
};

template <typename PointT>
class ConvexHull {
public:
  void setInputCloud(const std::shared_ptr<PointCloud<PointT>>& c) { in = c; }
  void reconstruct(PointCloud<PointT>& out) { out = *in; }
  std::shared_ptr<PointCloud<PointT>> in;
};

}  // namespace pcl
