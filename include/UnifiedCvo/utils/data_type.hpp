// Minimal linear-algebra value types for the drop-in C++ API.  The reference exposes Eigen types at
// this boundary (Eigen::Matrix4f, Eigen::Vector3f, Eigen::MatrixXf); Eigen is not a dependency here, so
// layout-compatible stand-ins are used and UnifiedCvo/eigen_interop.hpp converts where Eigen exists.
#pragma once
#include <array>
#include <cstddef>
#include <vector>

namespace cvo {

// 3-vector of floats, same layout as Eigen::Vector3f (12 bytes, packed).
struct Vec3f {
  float v[3];
  float& operator()(int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
};

// 2-vector of floats, same layout as Eigen::Vector2f (CvoPointCloud::geometry_type_at).
struct Vec2f {
  float v[2];
  float& operator()(int i) { return v[i]; }
  float operator()(int i) const { return v[i]; }
  float& operator[](int i) { return v[i]; }
  float operator[](int i) const { return v[i]; }
};

// Dynamic float vector in the role of Eigen::VectorXf (CvoPointCloud::label_at / feature_at return one row of the
// N x C / N x F matrices by value).
class VecXf {
 public:
  VecXf() = default;
  explicit VecXf(int n) : d_((size_t)n, 0.f) {}
  int size() const { return (int)d_.size(); }
  int rows() const { return (int)d_.size(); }
  float& operator()(int i) { return d_[(size_t)i]; }
  float operator()(int i) const { return d_[(size_t)i]; }
  float& operator[](int i) { return d_[(size_t)i]; }
  float operator[](int i) const { return d_[(size_t)i]; }
  const float* data() const { return d_.data(); }
  float* data() { return d_.data(); }

 private:
  std::vector<float> d_;
};

// 4x4 float matrix, COLUMN-major like Eigen::Matrix4f: element (r, c) is m[4*c + r].
struct Mat4f {
  float m[16];
  static Mat4f Identity() {
    Mat4f I{};
    I.m[0] = I.m[5] = I.m[10] = I.m[15] = 1.f;
    return I;
  }
  float& operator()(int r, int c) { return m[4 * c + r]; }
  float operator()(int r, int c) const { return m[4 * c + r]; }
  float* data() { return m; }
  const float* data() const { return m; }
  Mat4f inverse_rigid() const;  // [R | t]^-1 = [R^T | -R^T t]
};

// 3x3 float matrix, COLUMN-major like Eigen::Matrix3f: element (r, c) is m[3*c + r].
struct Mat3f {
  float m[9];
  static Mat3f Identity() {
    Mat3f I{};
    I.m[0] = I.m[4] = I.m[8] = 1.f;
    return I;
  }
  float& operator()(int r, int c) { return m[3 * c + r]; }
  float operator()(int r, int c) const { return m[3 * c + r]; }
  float* data() { return m; }
  const float* data() const { return m; }
};

// Dynamic float matrix, COLUMN-major like Eigen::MatrixXf (features_: N x F, labels_: N x C).
class MatXf {
 public:
  MatXf() = default;
  MatXf(int rows, int cols) { resize(rows, cols); }
  void resize(int rows, int cols) {
    rows_ = rows;
    cols_ = cols;
    d_.assign((size_t)rows * cols, 0.f);
  }
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  float& operator()(int r, int c) { return d_[(size_t)c * rows_ + r]; }
  float operator()(int r, int c) const { return d_[(size_t)c * rows_ + r]; }
  const float* data() const { return d_.data(); }
  VecXf row(int r) const {  // (Eigen: `VectorXf v = m.row(r)`)
    VecXf out(cols_);
    for (int c = 0; c < cols_; c++) out(c) = (*this)(r, c);
    return out;
  }

 private:
  int rows_ = 0, cols_ = 0;
  std::vector<float> d_;
};

// Row-major sparse float matrix in CSR form (the role of Eigen::SparseMatrix<float, RowMajor>).
struct SparseRowMat {
  int rows = 0, cols = 0;
  std::vector<int> row_ptr;  // rows + 1
  std::vector<int> col;
  std::vector<float> val;
  size_t nonZeros() const { return val.size(); }
};

}  // namespace cvo
