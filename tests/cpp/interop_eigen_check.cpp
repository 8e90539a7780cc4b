// Compiles include/UnifiedCvo/eigen_interop.hpp against the mock Eigen of tests/mock_include and EXECUTES every conversion
// (layout claims: Mat4f / Mat3f are column-major like Eigen's fixed matrices), plus the by-value accessors of
// cvo::CvoPointCloud (upstream CvoPointCloud.hpp:141-143).  Built and run by tests/test_interop_headers.py with g++.
#include <cstdio>
#include <vector>

#include "eigen_interop.hpp"
#include "utils/CvoPointCloud.hpp"

#ifndef UNIFIEDCVO_HAS_EIGEN
#error "the mock Eigen headers were not found: eigen_interop.hpp compiled to nothing"
#endif

#define CHECK(c)                                              \
  do {                                                        \
    if (!(c)) {                                               \
      std::printf("FAILED line %d: %s\n", __LINE__, #c);      \
      return 1;                                               \
    }                                                         \
  } while (0)

int main() {
  cvo::Mat4f m{};
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) m(r, c) = 10.f * r + c;
  const Eigen::Matrix4f e = cvo::to_eigen(m);
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) CHECK(e(r, c) == 10.f * r + c);  // same (row, column) element, both column-major
  const cvo::Mat4f back = cvo::from_eigen(e);
  for (int q = 0; q < 16; q++) CHECK(back.m[q] == m.m[q]);

  cvo::Mat3f m3{};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) m3(r, c) = 7.f * r - c;
  const Eigen::Matrix3f e3 = cvo::to_eigen(m3);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) CHECK(e3(r, c) == 7.f * r - c);
  const cvo::Mat3f b3 = cvo::from_eigen(e3);
  for (int q = 0; q < 9; q++) CHECK(b3.m[q] == m3.m[q]);

  const cvo::Vec3f v{{1.f, 2.f, 3.f}};
  const Eigen::Vector3f ev = cvo::to_eigen(v);
  CHECK(ev[0] == 1.f && ev[1] == 2.f && ev[2] == 3.f);
  const cvo::Vec3f bv = cvo::from_eigen(ev);
  CHECK(bv[0] == 1.f && bv[1] == 2.f && bv[2] == 3.f);

  const cvo::Vec2f g{{0.f, 1.f}};
  const Eigen::Vector2f eg = cvo::to_eigen(g);
  CHECK(eg[0] == 0.f && eg[1] == 1.f);
  const cvo::Vec2f bg = cvo::from_eigen(eg);
  CHECK(bg[0] == 0.f && bg[1] == 1.f);

  cvo::SparseRowMat s;
  s.rows = 3;
  s.cols = 4;
  s.row_ptr = {0, 2, 2, 3};
  s.col = {1, 3, 0};
  s.val = {0.5f, 0.25f, 2.f};
  const Eigen::SparseMatrix<float, Eigen::RowMajor> es = cvo::to_eigen(s);
  CHECK(es.rows() == 3 && es.cols() == 4 && es.nonZeros() == 3 && es.compressed);
  CHECK(es.coeff(0, 1) == 0.5f && es.coeff(0, 3) == 0.25f && es.coeff(2, 0) == 2.f && es.coeff(1, 1) == 0.f);

  // label_at / feature_at / geometry_type_at
  cvo::CvoPointCloud pc(5, 19);
  pc.reserve(3, 5, 19);
  std::vector<float> f{0.1f, 0.2f, 0.3f, 0.4f, 0.5f}, l(19, 0.f), gt{0.f, 1.f};
  l[4] = 1.f;
  CHECK(pc.add_point(1, cvo::Vec3f{{1.f, 2.f, 3.f}}, f, l, gt) == 0);
  const cvo::VecXf la = pc.label_at(1), fa = pc.feature_at(1), l0 = pc.label_at(0);
  CHECK(la.size() == 19 && fa.size() == 5 && la(4) == 1.f && la(3) == 0.f && l0(4) == 0.f);
  for (int q = 0; q < 5; q++) CHECK(fa[q] == f[(size_t)q]);
  const cvo::Vec2f ga = pc.geometry_type_at(1), g0 = pc.geometry_type_at(0);
  CHECK(ga(0) == 0.f && ga(1) == 1.f && g0(0) == 0.f && g0(1) == 0.f);
  const Eigen::VectorXf ela = cvo::to_eigen(la);
  CHECK(ela.size() == 19 && ela[4] == 1.f);
  const cvo::VecXf bla = cvo::from_eigen(ela);
  CHECK(bla.size() == 19 && bla[4] == 1.f);
  std::printf("interop ok\n");
  return 0;
}
