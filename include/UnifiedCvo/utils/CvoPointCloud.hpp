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
  // A file on disk.  A text file in upstream's own format (CvoPointCloud.cpp:89-148: "N F C" then per point
  // "x y z f_1..f_F l_1..l_C", points farther than 55 m dropped) is read as upstream's string constructor does.
  // A file that starts with a PCD header is read as the drivers do with pcl::io::loadPCDFile + the pcl::PointXYZ /
  // PointXYZRGB / PointXYZI constructors (CvoPointCloud.cpp:569-594, 633-652): ASCII, FIELDS "x y z", "x y z rgb"
  // or "x y z intensity" (one feature: the cvo_gpu_lidar_lib flavour, FEATURE_DIMENSIONS = 1).
  explicit CvoPointCloud(const std::string& filename);
  // upstream CvoPointCloud.cpp:1157-1199: "N F C" then per point "u v idepth f_1..f_F x y z l_1..l_C"; 0 / -1
  int read_cvo_pointcloud_from_file(const std::string& filename);

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
  // upstream CvoPointCloud.hpp:141-143 (CvoPointCloud.cpp:1282-1286): one point's class distribution / feature row /
  // (edge, surface) pair by value
  VecXf label_at(unsigned int index) const { return labels_.row((int)index); }
  VecXf feature_at(unsigned int index) const { return features_.row((int)index); }
  Vec2f geometry_type_at(unsigned int index) const {
    return Vec2f{{geometric_types_[(size_t)index * 2], geometric_types_[(size_t)index * 2 + 1]}};
  }
  const MatXf& semantics() const { return labels_; }
  const MatXf& features() const { return features_; }
  const std::vector<float>& geometric_types() const { return geometric_types_; }

  void reserve(int num_points, int feature_dims, int num_classes);
  // returns -1 if the cloud was not reserved, the index is out of range or geometric_type.size() != 2
  int add_point(int index, const Vec3f& xyz, const std::vector<float>& feature, const std::vector<float>& label,
                const std::vector<float>& geometric_type);

  // upstream CvoPointCloud.cpp:1289-1362; PCD files in the layout of pcl::io::savePCDFileASCII (PCL 1.9.1)
  void write_to_color_pcd(const std::string& name) const;      // PointXYZRGB: r,g,b <- features 2,1,0 as upstream
  void write_to_pcd(const std::string& name) const;            // PointXYZ
  void write_to_label_pcd(const std::string& name) const;      // PointXYZL: argmax label (nothing if no classes)
  void write_to_intensity_pcd(const std::string& name) const;  // PointXYZI: first feature
  void write_to_txt(const std::string& name) const;            // "N C" header, then xyz line + features/labels line

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
