// cvo::CvoPoint: the 192-byte AoS record the reference moves to the GPU, layout-identical to upstream's
// pcl::PointSegmentedDistribution<FEATURE_DIMENSIONS = 5, NUM_CLASSES = 19> (utils/PointSegmentedDistribution.hpp:17-99,
// utils/CvoPoint.hpp:8) without the PCL dependency.  An array of these IS what pcl::PointCloud<CvoPoint>::points holds,
// so the pcl overloads of CvoGPU take (const CvoPoint*, int n); where PCL exists, UnifiedCvo/pcl_interop.hpp forwards
// pcl::PointCloud<CvoPoint> objects unchanged.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "utils/CvoPointCloud.hpp"

namespace cvo {

struct alignas(16) CvoPoint {
  float x, y, z, data_w;              // PCL_ADD_POINT4D
  std::uint32_t rgba;                 // PCL_ADD_RGB (b, g, r, a bytes)
  float features[5];
  int label;
  float label_distribution[19];
  float geometric_type[2];
  float normal[3];
  float covariance[9];
  float cov_eigenvalues[3];
};
static_assert(sizeof(CvoPoint) == 192, "CvoPoint must be 192 bytes");
static_assert(offsetof(CvoPoint, rgba) == 16 && offsetof(CvoPoint, features) == 20 && offsetof(CvoPoint, label) == 40 &&
                  offsetof(CvoPoint, label_distribution) == 44 && offsetof(CvoPoint, geometric_type) == 120 &&
                  offsetof(CvoPoint, normal) == 128 && offsetof(CvoPoint, covariance) == 140 &&
                  offsetof(CvoPoint, cov_eigenvalues) == 176,
              "CvoPoint field offsets differ from pcl::PointSegmentedDistribution<5, 19>");

// What upstream's CvoPointCloud_to_gpu builds per point on the host before the copy (CvoGPU_impl.cu:206-263; also
// CvoPointCloud_to_pcl, CvoPointCloud.cpp): xyz, features (+ r, g, b bytes = min(255, f * 255)), label_distribution
// (+ label = argmax), geometric_type.
std::vector<CvoPoint> CvoPointCloud_to_cvo_points(const CvoPointCloud& pc);

}  // namespace cvo
