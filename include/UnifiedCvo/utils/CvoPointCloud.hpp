// Host point-cloud container: the accessor subset of upstream utils/CvoPointCloud.hpp:126-188 that the
// align() path and its drivers use.  Image / LiDAR constructors are out of scope.
#pragma once
#include <string>
#include <vector>

#include "utils/data_type.hpp"

#ifndef FEATURE_DIMENSIONS
#define FEATURE_DIMENSIONS 5
#endif
#ifndef NUM_CLASSES
#define NUM_CLASSES 19
#endif

namespace cvo {

class CvoPointCloud {
 public:
  enum GeometryType { EDGE = 0, SURFACE = 1 };

  CvoPointCloud();
  CvoPointCloud(int feature_dimensions, int num_classes);
  // ASCII .pcd with FIELDS "x y z" or "x y z rgb": the pcl::PointXYZ / pcl::PointXYZRGB constructors
  // (upstream CvoPointCloud.cpp:569-594, 633-652) applied to what pcl::io::loadPCDFile returns.
  explicit CvoPointCloud(const std::string& pcd_filename);

  static CvoPointCloud from_xyz(const float* xyz, int n);                                // type (1,0), F = 0
  static CvoPointCloud from_xyzrgb(const float* xyz, const unsigned char* rgb, int n);   // type (0,1), F = 5

  static void transform(const Mat4f& pose, const CvoPointCloud& input, CvoPointCloud& output);
  friend CvoPointCloud operator+(CvoPointCloud a, const CvoPointCloud& b);

  int num_points() const { return num_points_; }
  int size() const { return num_points_; }
  int num_classes() const { return num_classes_; }
  int num_features() const { return feature_dimensions_; }
  int feature_dimensions() const { return feature_dimensions_; }
  const std::vector<Vec3f>& positions() const { return positions_; }
  Vec3f at(unsigned int index) const { return positions_[index]; }
  const MatXf& labels() const { return labels_; }
  const MatXf& semantics() const { return labels_; }
  const MatXf& features() const { return features_; }
  const std::vector<float>& geometric_types() const { return geometric_types_; }

  void reserve(int num_points, int feature_dims, int num_classes);
  // returns -1 if the cloud was not reserved, the index is out of range or geometric_type.size() != 2
  int add_point(int index, const Vec3f& xyz, const std::vector<float>& feature, const std::vector<float>& label,
                const std::vector<float>& geometric_type);

  void write_to_color_pcd(const std::string& name) const;

 private:
  int num_points_ = 0;
  int num_classes_ = 0;
  int feature_dimensions_ = 0;
  std::vector<Vec3f> positions_;
  MatXf features_;
  MatXf labels_;
  std::vector<float> geometric_types_;
};

}  // namespace cvo
