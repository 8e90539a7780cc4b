// File I/O of cvo::CvoPointCloud:
//   cvo_pointcloud_io <cloud file: upstream text format or ASCII .pcd> <output directory>
//   cvo_pointcloud_io --raw <upstream "u v idepth features xyz labels" file> <output directory>
// Prints a one-line summary and writes xyz.pcd, color.pcd, label.pcd, intensity.pcd and cloud.txt.
#include <cstdio>
#include <string>

#include "utils/CvoPointCloud.hpp"

int main(int argc, char* argv[]) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s [--raw] cloud_file out_dir\n", argv[0]);
    return 2;
  }
  const bool raw = std::string(argv[1]) == "--raw";
  const std::string in = argv[raw ? 2 : 1], dir = argv[raw ? 3 : 2];
  cvo::CvoPointCloud pc;
  if (raw) {
    const int rc = pc.read_cvo_pointcloud_from_file(in);
    if (rc != 0) {
      std::printf("read_cvo_pointcloud_from_file %d\n", rc);
      return 1;
    }
  } else {
    pc = cvo::CvoPointCloud(in);
  }
  double sx = 0, sf = 0, sl = 0;
  for (int i = 0; i < pc.num_points(); i++) {
    for (int c = 0; c < 3; c++) sx += pc.positions()[i][c];
    for (int j = 0; j < pc.features().cols(); j++) sf += pc.features()(i, j);
    for (int j = 0; j < pc.labels().cols() && pc.labels().rows() == pc.num_points(); j++) sl += pc.labels()(i, j);
  }
  std::printf("points %d features %d classes %d geometric_types %zu sum_xyz %.9g sum_features %.9g sum_labels %.9g\n",
              pc.num_points(), pc.feature_dimensions(), pc.num_classes(), pc.geometric_types().size(), sx, sf, sl);
  pc.write_to_pcd(dir + "/xyz.pcd");
  pc.write_to_color_pcd(dir + "/color.pcd");
  pc.write_to_label_pcd(dir + "/label.pcd");
  pc.write_to_intensity_pcd(dir + "/intensity.pcd");
  pc.write_to_txt(dir + "/cloud.txt");
  return 0;
}
