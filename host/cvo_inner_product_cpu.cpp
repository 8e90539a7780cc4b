// cvo::CvoGPU::inner_product_cpu: the reference's HOST inner product (upstream src/cvo/CvoGPU.cpp:95-213,
// se_kernel_init_ell_cpu + inner_product_cpu), part of the public API (CvoGPU.hpp:225-229) and the is_gpu = false branch
// of function_angle.  It is a different function from the GPU path by upstream's own design - plain ell (no range
// factor), no neighbour cap, radius search (nanoflann) instead of the ordered scan, and NO cut-off on the colour and
// semantic kernels - and it is host code upstream too; nothing in the GPU path falls back to it.
//
// Upstream: moving points <- T^-1 (4x4 inverse, float); for every fixed point a nanoflann radiusSearch with squared radius
// d2_thres = -2 l^2 log(sp_thres / sigma^2); per match k = sigma^2 exp(-d2 / (2 l^2)) [is_using_geometry],
// ck = c_sigma^2 exp(-d2c / (2 c_ell^2)) over ALL feature columns [is_using_intensity], sk = s_sigma^2 exp(-d2s / (2 s_ell^2))
// over all classes [is_using_semantics], a = ck k sk, kept if a > sp_thres; result = sum of the kept a (Eigen sparse sum,
// filled from a tbb::concurrent_vector: the summation order is not defined upstream).  Here: a uniform grid with cell
// = radius replaces the kd-tree (same neighbour set), the sum is accumulated in double and rounded once.
#include <algorithm>
#include <cmath>
#include <vector>

#include "cvo/CvoGPU.hpp"

namespace cvo {
namespace {

// general 4x4 inverse (upstream: Eigen::Matrix4f::inverse()), Gauss-Jordan with partial pivoting in double
Mat4f inverse4(const Mat4f& m) {
  double a[4][8];
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) {
      a[r][c] = m(r, c);
      a[r][4 + c] = r == c ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; c++) {
    int piv = c;
    for (int r = c + 1; r < 4; r++)
      if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
    for (int k = 0; k < 8; k++) std::swap(a[c][k], a[piv][k]);
    const double d = a[c][c];
    for (int k = 0; k < 8; k++) a[c][k] /= d;
    for (int r = 0; r < 4; r++) {
      if (r == c) continue;
      const double f = a[r][c];
      for (int k = 0; k < 8; k++) a[r][k] -= f * a[c][k];
    }
  }
  Mat4f out{};
  for (int r = 0; r < 4; r++)
    for (int c = 0; c < 4; c++) out(r, c) = (float)a[r][4 + c];
  return out;
}

}  // namespace

float CvoGPU::inner_product_cpu(const CvoPointCloud& source_points, const CvoPointCloud& target_points,
                                const Mat4f& t2s_frame_transform, float ell) const {
  const int n = source_points.num_points(), m = target_points.num_points();
  if (n == 0 || m == 0) return 0.f;  // upstream CvoGPU.cpp:190-192
  const Mat4f s2t = inverse4(t2s_frame_transform);
  std::vector<Vec3f> moving(m);
  for (int j = 0; j < m; j++) {  // moving_positions[j] = rot * moving_positions[j] + trans (float)
    const Vec3f& p = target_points.positions()[j];
    for (int r = 0; r < 3; r++) moving[j][r] = (s2t(r, 0) * p[0] + s2t(r, 1) * p[1] + s2t(r, 2) * p[2]) + s2t(r, 3);
  }
  const CvoParams& P = params;
  const float s2 = P.sigma * P.sigma;
  const float l = ell;
  const float d2_thres = (float)(-2.0 * l * l * std::log(P.sp_thres / s2));
  if (!(d2_thres > 0.f)) return 0.f;  // empty search radius
  const double radius = std::sqrt((double)d2_thres);
  // uniform grid over the moving points, cell = radius
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int j = 0; j < m; j++)
    for (int c = 0; c < 3; c++) {
      lo[c] = std::min(lo[c], (double)moving[j][c]);
      hi[c] = std::max(hi[c], (double)moving[j][c]);
    }
  double cell = radius;
  int dim[3];
  for (;;) {
    double cells = 1;
    for (int c = 0; c < 3; c++) {
      dim[c] = (int)std::min(1e6, (hi[c] - lo[c]) / cell) + 1;
      cells *= dim[c];
    }
    if (cells <= 4e6) break;
    cell *= 2;
  }
  auto cell_of = [&](double v, int c) {
    int q = (int)std::floor((v - lo[c]) / cell);
    return q < 0 ? 0 : (q >= dim[c] ? dim[c] - 1 : q);
  };
  const size_t nc = (size_t)dim[0] * dim[1] * dim[2];
  std::vector<int> start(nc + 1, 0), items(m), which(m);
  for (int j = 0; j < m; j++) {
    which[j] = (cell_of(moving[j][2], 2) * dim[1] + cell_of(moving[j][1], 1)) * dim[0] + cell_of(moving[j][0], 0);
    start[which[j] + 1]++;
  }
  for (size_t c = 0; c < nc; c++) start[c + 1] += start[c];
  {
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int j = 0; j < m; j++) items[fill[which[j]]++] = j;
  }
  const MatXf &Fa = source_points.features(), &Fb = target_points.features();
  const MatXf &La = source_points.labels(), &Lb = target_points.labels();
  const int nf = std::min(Fa.cols(), Fb.cols()), nl = std::min(La.cols(), Lb.cols());
  const int reach = (int)std::ceil(radius / cell);
  double total = 0;
  for (int i = 0; i < n; i++) {
    const Vec3f& x = source_points.positions()[i];
    const int cx = cell_of(x[0], 0), cy = cell_of(x[1], 1), cz = cell_of(x[2], 2);
    for (int z = std::max(0, cz - reach); z <= std::min(dim[2] - 1, cz + reach); z++)
      for (int y = std::max(0, cy - reach); y <= std::min(dim[1] - 1, cy + reach); y++)
        for (int xx = std::max(0, cx - reach); xx <= std::min(dim[0] - 1, cx + reach); xx++) {
          const size_t c = ((size_t)z * dim[1] + y) * dim[0] + xx;
          for (int q = start[c]; q < start[c + 1]; q++) {
            const int idx = items[q];
            const float dx = x[0] - moving[idx][0], dy = x[1] - moving[idx][1], dz = x[2] - moving[idx][2];
            const float d2 = dx * dx + dy * dy + dz * dz;
            if (!(d2 < d2_thres)) continue;  // nanoflann radiusSearch: squared distance below the squared radius
            float k = 1, ck = 1, sk = 1;
            if (P.is_using_semantics) {
              float d2s = 0;
              for (int c2 = 0; c2 < nl; c2++) {
                const float t = La(i, c2) - Lb(idx, c2);
                d2s += t * t;
              }
              sk = (float)((double)(P.s_sigma * P.s_sigma) * std::exp(-(double)d2s / (2.0 * P.s_ell * P.s_ell)));
            }
            if (P.is_using_geometry) k = (float)((double)s2 * std::exp(-(double)d2 / (2.0 * l * l)));
            if (P.is_using_intensity) {
              float d2c = 0;
              for (int c2 = 0; c2 < nf; c2++) {
                const float t = Fa(i, c2) - Fb(idx, c2);
                d2c += t * t;
              }
              ck = (float)((double)(P.c_sigma * P.c_sigma) * std::exp(-(double)d2c / (2.0 * P.c_ell * P.c_ell)));
            }
            const float a = ck * k * sk;
            if (a > P.sp_thres) total += (double)a;
          }
        }
  }
  return (float)total;
}

}  // namespace cvo
