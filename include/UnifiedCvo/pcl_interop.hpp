// Forwards pcl::PointCloud<CvoPoint> objects to the pcl overloads of cvo::CvoGPU (upstream CvoGPU.hpp:91-99, 167-171,
// 220-224), for code bases that have PCL.  Header-only; compiles to nothing where PCL is absent.  The point type only
// has to be the 192-byte record of utils/CvoPoint.hpp (upstream's pcl::PointSegmentedDistribution<5, 19> is).
#pragma once
#include "cvo/CvoGPU.hpp"

#if defined(__has_include)
#if __has_include(<pcl/point_cloud.h>)
#define UNIFIEDCVO_HAS_PCL 1
#include <pcl/point_cloud.h>

namespace cvo {

template <typename PointT>
int align(const CvoGPU& cvo, const pcl::PointCloud<PointT>& source, const pcl::PointCloud<PointT>& target,
          const Mat4f& T_target_frame_to_source_frame, Mat4f& transform, Association* association = nullptr,
          double* registration_seconds = nullptr) {
  static_assert(sizeof(PointT) == sizeof(CvoPoint), "the point type must be the 192-byte CvoPoint record");
  return cvo.align(source.points.data(), (int)source.size(), target.points.data(), (int)target.size(),
                   T_target_frame_to_source_frame, transform, association, registration_seconds);
}
template <typename PointT>
float inner_product_gpu(const CvoGPU& cvo, const pcl::PointCloud<PointT>& source, const pcl::PointCloud<PointT>& target,
                        const Mat4f& T, float ell) {
  static_assert(sizeof(PointT) == sizeof(CvoPoint), "the point type must be the 192-byte CvoPoint record");
  return cvo.inner_product_gpu(source.points.data(), (int)source.size(), target.points.data(), (int)target.size(), T, ell);
}
template <typename PointT>
float function_angle(const CvoGPU& cvo, const pcl::PointCloud<PointT>& source, const pcl::PointCloud<PointT>& target,
                     const Mat4f& T, float ell, bool is_approximate = true) {
  static_assert(sizeof(PointT) == sizeof(CvoPoint), "the point type must be the 192-byte CvoPoint record");
  return cvo.function_angle(source.points.data(), (int)source.size(), target.points.data(), (int)target.size(), T, ell,
                            is_approximate);
}

}  // namespace cvo
#endif
#endif
