// cvo::CvoPointCloud accessor subset (see include/UnifiedCvo/utils/CvoPointCloud.hpp).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include "utils/CvoPointCloud.hpp"

namespace cvo {

Mat4f Mat4f::inverse_rigid() const {
  Mat4f o = Mat4f::Identity();
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) o(r, c) = (*this)(c, r);
  for (int r = 0; r < 3; r++) {
    float s = 0;
    for (int c = 0; c < 3; c++) s += o(r, c) * (*this)(c, 3);
    o(r, 3) = -s;
  }
  return o;
}

CvoPointCloud::CvoPointCloud() {}
CvoPointCloud::CvoPointCloud(int feature_dimensions, int num_classes)
    : num_points_(0), num_classes_(num_classes), feature_dimensions_(feature_dimensions) {}

CvoPointCloud CvoPointCloud::from_xyz(const float* xyz, int n) {
  CvoPointCloud pc(0, 0);
  pc.num_points_ = n;
  pc.positions_.resize(n);
  pc.geometric_types_.resize(2 * (size_t)n);
  for (int i = 0; i < n; i++) {
    pc.positions_[i] = Vec3f{{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}};
    pc.geometric_types_[2 * i] = 1;
    pc.geometric_types_[2 * i + 1] = 0;
  }
  return pc;
}

CvoPointCloud CvoPointCloud::from_xyzrgb(const float* xyz, const unsigned char* rgb, int n) {
  CvoPointCloud pc(5, 0);
  pc.num_points_ = n;
  pc.positions_.resize(n);
  pc.features_.resize(n, 5);
  pc.geometric_types_.resize(2 * (size_t)n);
  for (int i = 0; i < n; i++) {
    pc.positions_[i] = Vec3f{{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]}};
    for (int c = 0; c < 3; c++) pc.features_(i, c) = ((float)(int)rgb[3 * i + c]) / 255.0f;
    pc.geometric_types_[2 * i] = 0;
    pc.geometric_types_[2 * i + 1] = 1;
  }
  return pc;
}

namespace {

bool is_good_point(const Vec3f& p) {  // upstream CvoPointCloud.cpp:47-52
  const float n = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  return !(n > 55.f);
}

bool looks_like_pcd(const std::string& filename) {
  std::ifstream f(filename);
  std::string line;
  for (int k = 0; k < 4 && std::getline(f, line); k++) {
    if (line.rfind("# .PCD", 0) == 0 || line.rfind("VERSION", 0) == 0 || line.rfind("FIELDS", 0) == 0) return true;
    if (!line.empty() && line[0] != '#') break;
  }
  return false;
}

}  // namespace

CvoPointCloud::CvoPointCloud(const std::string& filename) {
  if (!looks_like_pcd(filename)) {
    // upstream's text format (CvoPointCloud.cpp:89-148)
    std::ifstream in(filename);
    if (!in) throw std::runtime_error("cannot open " + filename);
    int total = 0, F = 0, C = 0;
    in >> total >> F >> C;
    if (!in || total < 0 || F < 0 || C < 0) throw std::runtime_error("not a CvoPointCloud text file: " + filename);
    std::vector<Vec3f> pos;
    std::vector<float> feat, lab;
    for (int i = 0; i < total; i++) {
      Vec3f p;
      in >> p[0] >> p[1] >> p[2];
      std::vector<float> f(F), l(C);
      for (int j = 0; j < F; j++) in >> f[j];
      for (int j = 0; j < C; j++) in >> l[j];
      if (!in) throw std::runtime_error("truncated CvoPointCloud text file: " + filename);
      if (!is_good_point(p)) continue;
      pos.push_back(p);
      feat.insert(feat.end(), f.begin(), f.end());
      lab.insert(lab.end(), l.begin(), l.end());
    }
    num_points_ = (int)pos.size();
    feature_dimensions_ = F;
    num_classes_ = C;
    positions_ = pos;
    features_.resize(num_points_, F);
    labels_.resize(num_points_, C);
    for (int i = 0; i < num_points_; i++) {
      for (int j = 0; j < F; j++) features_(i, j) = feat[(size_t)i * F + j];
      for (int j = 0; j < C; j++) labels_(i, j) = lab[(size_t)i * C + j];
    }
    // upstream leaves geometric_types_ empty here; CvoPointCloud_to_gpu then reads zeros (CvoGPU_impl.cu:250-253)
    return;
  }
  std::ifstream f(filename);
  if (!f) throw std::runtime_error("cannot open " + filename);
  std::string line;
  std::vector<std::string> fields;
  bool data = false;
  std::vector<float> xyz, intensity;
  std::vector<unsigned char> rgb;
  while (std::getline(f, line)) {
    if (!data) {
      std::istringstream ss(line);
      std::string tag;
      ss >> tag;
      if (tag == "FIELDS") {
        std::string w;
        while (ss >> w) fields.push_back(w);
      } else if (tag == "DATA") {
        std::string kind;
        ss >> kind;
        if (kind != "ascii") throw std::runtime_error("only ASCII .pcd files are supported: " + filename);
        data = true;
      }
      continue;
    }
    std::istringstream ss(line);
    std::vector<std::string> tok;
    std::string w;
    while (ss >> w) tok.push_back(w);
    if (tok.size() < 3 || fields.size() < 3) continue;
    for (int c = 0; c < 3; c++) xyz.push_back(std::strtof(tok[c].c_str(), nullptr));
    if (fields.size() >= 4 && fields[3] == "rgb" && tok.size() >= 4) {
      const unsigned long u = std::strtoul(tok[3].c_str(), nullptr, 10);
      rgb.push_back((unsigned char)((u >> 16) & 255));
      rgb.push_back((unsigned char)((u >> 8) & 255));
      rgb.push_back((unsigned char)(u & 255));
    } else if (fields.size() >= 4 && fields[3] == "intensity" && tok.size() >= 4) {
      intensity.push_back(std::strtof(tok[3].c_str(), nullptr));
    }
  }
  const int n = (int)(xyz.size() / 3);
  if (intensity.size() == (size_t)n && n > 0) {
    // LiDAR flavour: one feature (intensity), geometric type left to the caller's selection (edge / surface); the
    // plain loader marks every point as a surface point like the colour constructor does
    *this = CvoPointCloud(1, 0);
    num_points_ = n;
    positions_.resize(n);
    features_.resize(n, 1);
    geometric_types_.assign(2 * (size_t)n, 0.f);
    for (int i = 0; i < n; i++) {
      for (int c = 0; c < 3; c++) positions_[i][c] = xyz[3 * (size_t)i + c];
      features_(i, 0) = intensity[i];
      geometric_types_[2 * (size_t)i + 1] = 1.f;
    }
    return;
  }
  *this = (rgb.size() == 3 * (size_t)n && n > 0) ? from_xyzrgb(xyz.data(), rgb.data(), n) : from_xyz(xyz.data(), n);
}

int CvoPointCloud::read_cvo_pointcloud_from_file(const std::string& filename) {
  std::ifstream in(filename);
  if (!in.is_open()) return -1;
  in >> num_points_ >> feature_dimensions_ >> num_classes_;
  if (!in || num_points_ < 0 || feature_dimensions_ < 0 || num_classes_ < 0) return -1;
  positions_.assign(num_points_, Vec3f{{0, 0, 0}});
  features_.resize(num_points_, feature_dimensions_);
  labels_.resize(num_classes_ ? num_points_ : 0, num_classes_);
  for (int i = 0; i < num_points_; i++) {
    float u, v, idepth;
    in >> u >> v >> idepth;
    for (int j = 0; j < feature_dimensions_; j++) in >> features_(i, j);
    for (int j = 0; j < 3; j++) in >> positions_[i][j];
    for (int j = 0; j < num_classes_; j++) in >> labels_(i, j);
  }
  return in ? 0 : -1;
}

void CvoPointCloud::transform(const Mat4f& pose, const CvoPointCloud& input, CvoPointCloud& output) {
  // num_points_, num_classes_, features, labels, geometric types are copied; feature_dimensions_ is not
  output.num_points_ = input.num_points_;
  output.num_classes_ = input.num_classes_;
  output.features_ = input.features_;
  output.labels_ = input.labels_;
  output.positions_.resize(input.num_points_);
  for (int j = 0; j < input.num_points_; j++) {
    const Vec3f& p = input.positions_[j];
    Vec3f q;
    for (int r = 0; r < 3; r++) q[r] = pose(r, 0) * p[0] + pose(r, 1) * p[1] + pose(r, 2) * p[2] + pose(r, 3);
    output.positions_[j] = q;
  }
  output.geometric_types_ = input.geometric_types_;
}

CvoPointCloud operator+(CvoPointCloud a, const CvoPointCloud& b) {
  const int na = a.num_points_, nb = b.num_points_;
  a.positions_.insert(a.positions_.end(), b.positions_.begin(), b.positions_.end());
  auto vcat = [&](const MatXf& A, const MatXf& B) {
    const int cols = A.cols() ? A.cols() : B.cols();
    MatXf R(na + nb, cols);
    for (int c = 0; c < cols; c++) {
      for (int i = 0; i < na && c < A.cols() && i < A.rows(); i++) R(i, c) = A(i, c);
      for (int i = 0; i < nb && c < B.cols() && i < B.rows(); i++) R(na + i, c) = B(i, c);
    }
    return R;
  };
  if (a.features_.cols() || b.features_.cols()) a.features_ = vcat(a.features_, b.features_);
  if (a.labels_.cols() || b.labels_.cols()) a.labels_ = vcat(a.labels_, b.labels_);
  a.geometric_types_.insert(a.geometric_types_.end(), b.geometric_types_.begin(), b.geometric_types_.end());
  a.num_points_ = na + nb;
  return a;
}

void CvoPointCloud::reserve(int num_points, int feature_dims, int num_classes) {
  num_points_ = num_points;
  num_classes_ = num_classes;
  feature_dimensions_ = feature_dims;
  positions_.assign(num_points, Vec3f{{0, 0, 0}});
  if (feature_dims) features_.resize(num_points, feature_dims);
  if (num_classes) labels_.resize(num_points, num_classes);
  geometric_types_.assign(2 * (size_t)num_points, 0.f);
}

int CvoPointCloud::add_point(int index, const Vec3f& xyz, const std::vector<float>& feature,
                             const std::vector<float>& label, const std::vector<float>& geometric_type) {
  if (index >= num_points_ || index < 0) return -1;
  if ((int)positions_.size() < num_points_ || features_.rows() < num_points_ ||
      features_.cols() != feature_dimensions_ || geometric_type.size() != 2)
    return -1;
  positions_[index] = xyz;
  for (int c = 0; c < feature_dimensions_ && c < (int)feature.size(); c++) features_(index, c) = feature[c];
  for (int c = 0; c < num_classes_ && c < (int)label.size(); c++) labels_(index, c) = label[c];
  geometric_types_[2 * index] = geometric_type[0];
  geometric_types_[2 * index + 1] = geometric_type[1];
  return 0;
}

namespace {

// The header pcl::io::savePCDFileASCII writes (PCL 1.9.1 PCDWriter::generateHeader) for an unorganised cloud.
FILE* open_pcd(const std::string& name, const char* fields, const char* sizes, const char* types, const char* counts, int n) {
  FILE* f = std::fopen(name.c_str(), "w");
  if (!f) throw std::runtime_error("cannot write " + name);
  std::fprintf(f, "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\n"
                  "WIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA ascii\n",
               fields, sizes, types, counts, n, n);
  return f;
}

}  // namespace

void CvoPointCloud::write_to_color_pcd(const std::string& name) const {
  // export_to_pcd<pcl::PointXYZRGB> (upstream CvoPointCloud.cpp:1232-1259): r, g, b <- features 2, 1, 0,
  // min(255, (int)(f * 255)); PCL stores the packed colour with alpha 255 and prints it as an unsigned integer
  FILE* f = open_pcd(name, "x y z rgb", "4 4 4 4", "F F F U", "1 1 1 1", num_points_);
  for (int i = 0; i < num_points_; i++) {
    unsigned r = 0, g = 0, b = 0;
    if (features_.rows() == num_points_ && features_.cols() >= 3) {
      auto q = [&](float v) { return (unsigned)(unsigned char)std::min(255, (int)(v * 255)); };
      r = q(features_(i, 2));
      g = q(features_(i, 1));
      b = q(features_(i, 0));
    }
    std::fprintf(f, "%.8g %.8g %.8g %u\n", positions_[i][0], positions_[i][1], positions_[i][2],
                 (255u << 24) | (r << 16) | (g << 8) | b);
  }
  std::fclose(f);
}

void CvoPointCloud::write_to_pcd(const std::string& name) const {
  FILE* f = open_pcd(name, "x y z", "4 4 4", "F F F", "1 1 1", num_points_);
  for (int i = 0; i < num_points_; i++)
    std::fprintf(f, "%.8g %.8g %.8g\n", positions_[i][0], positions_[i][1], positions_[i][2]);
  std::fclose(f);
}

void CvoPointCloud::write_to_label_pcd(const std::string& name) const {
  if (num_classes_ < 1) return;
  FILE* f = open_pcd(name, "x y z label", "4 4 4 4", "F F F U", "1 1 1 1", num_points_);
  for (int i = 0; i < num_points_; i++) {
    int l = 0;  // Eigen maxCoeff(&l): first index of the maximum
    for (int j = 1; j < num_classes_; j++)
      if (labels_(i, j) > labels_(i, l)) l = j;
    std::fprintf(f, "%.8g %.8g %.8g %u\n", positions_[i][0], positions_[i][1], positions_[i][2], (unsigned)l);
  }
  std::fclose(f);
}

void CvoPointCloud::write_to_intensity_pcd(const std::string& name) const {
  FILE* f = open_pcd(name, "x y z intensity", "4 4 4 4", "F F F F", "1 1 1 1", num_points_);
  for (int i = 0; i < num_points_; i++) {
    const float it = (features_.rows() == num_points_ && features_.cols() >= 1) ? features_(i, 0) : 0.f;
    std::fprintf(f, "%.8g %.8g %.8g %.8g\n", positions_[i][0], positions_[i][1], positions_[i][2], it);
  }
  std::fclose(f);
}

void CvoPointCloud::write_to_txt(const std::string& name) const {
  // upstream CvoPointCloud.cpp:1327-1348 (note the two-number header: N and C, no F), default ostream formatting
  std::ofstream out(name);
  if (!out.is_open()) return;
  out << num_points_ << " " << num_classes_ << "\n";
  for (int i = 0; i < num_points_; i++) {
    out << positions_[i][0] << " " << positions_[i][1] << " " << positions_[i][2] << std::endl;
    for (int j = 0; j < feature_dimensions_; j++) out << features_(i, j) << " ";
    if (num_classes_)
      for (int j = 0; j < num_classes_; j++) out << labels_(i, j) << " ";
    out << "\n";
  }
}

}  // namespace cvo
