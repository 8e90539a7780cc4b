// Conversions between the dependency-free value types of utils/data_type.hpp and Eigen, for code bases that have
// Eigen (the reference's callers do).  Header-only; compiles to nothing where Eigen is absent.
//   cvo::Mat4f  <-> Eigen::Matrix4f        (both 16 floats, column-major: the conversion is a memcpy)
//   cvo::Mat3f  <-> Eigen::Matrix3f
//   cvo::Vec3f  <-> Eigen::Vector3f
//   cvo::Vec2f  <-> Eigen::Vector2f,  cvo::VecXf <-> Eigen::VectorXf   (CvoPointCloud::geometry_type_at / label_at / feature_at)
//   cvo::SparseRowMat -> Eigen::SparseMatrix<float, Eigen::RowMajor>   (Association::pairs, upstream Association.hpp:9)
#pragma once
#include "utils/data_type.hpp"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>) && __has_include(<Eigen/Sparse>)
#define UNIFIEDCVO_HAS_EIGEN 1
#include <Eigen/Dense>
#include <Eigen/Sparse>
#include <cstring>
#include <vector>

namespace cvo {

inline Eigen::Matrix4f to_eigen(const Mat4f& m) {
  Eigen::Matrix4f e;
  std::memcpy(e.data(), m.data(), sizeof(float) * 16);
  return e;
}
inline Mat4f from_eigen(const Eigen::Matrix4f& e) {
  Mat4f m;
  std::memcpy(m.data(), e.data(), sizeof(float) * 16);
  return m;
}
inline Eigen::Matrix3f to_eigen(const Mat3f& m) {
  Eigen::Matrix3f e;
  std::memcpy(e.data(), m.data(), sizeof(float) * 9);
  return e;
}
inline Mat3f from_eigen(const Eigen::Matrix3f& e) {
  Mat3f m;
  std::memcpy(m.data(), e.data(), sizeof(float) * 9);
  return m;
}
inline Eigen::Vector3f to_eigen(const Vec3f& v) { return Eigen::Vector3f(v[0], v[1], v[2]); }
inline Vec3f from_eigen(const Eigen::Vector3f& e) { return Vec3f{{e[0], e[1], e[2]}}; }

inline Eigen::Vector2f to_eigen(const Vec2f& v) { return Eigen::Vector2f(v[0], v[1]); }
inline Vec2f from_eigen(const Eigen::Vector2f& e) { return Vec2f{{e[0], e[1]}}; }
inline Eigen::VectorXf to_eigen(const VecXf& v) {
  Eigen::VectorXf e(v.size());
  for (int i = 0; i < v.size(); i++) e[i] = v[i];
  return e;
}
inline VecXf from_eigen(const Eigen::VectorXf& e) {
  VecXf v((int)e.size());
  for (int i = 0; i < v.size(); i++) v[i] = e[i];
  return v;
}

inline Eigen::SparseMatrix<float, Eigen::RowMajor> to_eigen(const SparseRowMat& s) {
  std::vector<Eigen::Triplet<float>> t;
  t.reserve(s.val.size());
  for (int i = 0; i < s.rows; i++)
    for (int q = s.row_ptr[i]; q < s.row_ptr[i + 1]; q++) t.emplace_back(i, s.col[q], s.val[q]);
  Eigen::SparseMatrix<float, Eigen::RowMajor> m(s.rows, s.cols);
  m.setFromTriplets(t.begin(), t.end());
  m.makeCompressed();
  return m;
}

}  // namespace cvo
#endif
#endif
