// MOCK of pcl/point_cloud.h for tests/test_interop_headers.py (PCL is not in this image): the two members of
// pcl::PointCloud<PointT> that include/UnifiedCvo/pcl_interop.hpp reads.  Test scaffolding only.
#pragma once
#include <cstddef>
#include <vector>
namespace pcl {
template <typename PointT>
struct PointCloud {
  std::vector<PointT> points;
  std::size_t size() const { return points.size(); }
};
}  // namespace pcl
