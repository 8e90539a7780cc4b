/*
 * cvo_oracle.cpp -- CPU restatement of unified_cvo's pairwise CvoGPU::align() path.
 *
 * TEST INFRASTRUCTURE ONLY (see cvo_oracle.h).  PARITY UNPINNED: no reference golden
 * vectors exist and the reference cannot be built here; pins are tests/test_oracle_*.py.
 *
 * Every function cites the reference file:line (relative to the upstream repository root)
 * it follows.  The restatement follows the reference's CUDA path (src/cvo/CvoGPU.cu), not
 * its two CPU variants, which differ by design.
 *
 * Floating-point conventions (the reference's exact rounding is compiler dependent and
 * unknowable here, so they are fixed explicitly and mirrored by the HIP kernels):
 *   * code that the reference runs ON THE DEVICE (kernels, thrust functors) is compiled by
 *     nvcc with its default -fmad=true: "a*b + c" patterns are written as explicit fmaf();
 *     Eigen's fixed-size 3-term reductions are a0 + (a1 + a2) (redux_novec_unroller).
 *   * code that the reference runs ON THE HOST (LieGroup.cpp, align_impl scalar maths) is
 *     written with plain, unfused arithmetic in source order.
 *   * this file must be compiled with -ffp-contract=off so that only the explicit fmaf()
 *     calls fuse.
 *   * float/double promotion follows the C++ expression types of the reference exactly
 *     (e.g. exp() is evaluated in double because of the 2.0 literal, CvoGPU.cu:551).
 */
#include "cvo_oracle.h"

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <queue>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

// ---- build-time convention switches (oracle/Makefile: variants; tests/test_oracle_conventions.py) -------------
// The default build fixes the conventions listed above.  Because no reference build exists to pin them, the
// oracle can be rebuilt under the other plausible choices and the tests show that the results the north_star
// promises (discrete decisions over the well-conditioned prefix, final pose to 1e-4 / 2e-4) do not depend on them:
//   -DORACLE_FMA_NONE          no contraction anywhere: every FMAF(a, b, c) is a rounded product plus a rounded sum
//   -ffp-contract=fast         ("all") the explicit FMAFs stay and the compiler also fuses the host-side expressions
//   -DORACLE_SUM_ALT           3-term reductions associate the other way: (a0 + a1) + a2 where the default has
//                              a0 + (a1 + a2), and vice versa
#if defined(ORACLE_FMA_NONE)
static inline float FMAF(float a, float b, float c) {
  volatile float p = a * b;  // (volatile: the product is rounded even when the file is built with contraction on)
  return p + c;
}
#else
static inline float FMAF(float a, float b, float c) { return std::fmaf(a, b, c); }
#endif
#if defined(ORACLE_SUM_ALT)
#define SUM3(a0, a1, a2) (((a0) + (a1)) + (a2))
#define SUM3L(a0, a1, a2) ((a0) + ((a1) + (a2)))
#else
#define SUM3(a0, a1, a2) ((a0) + ((a1) + (a2)))
#define SUM3L(a0, a1, a2) (((a0) + (a1)) + (a2))
#endif

namespace {

constexpr int FD = ORACLE_FEATURE_DIMENSIONS;
constexpr int NC = ORACLE_NUM_CLASSES;

// ---- small helpers encoding the conventions above ------------------------------------

// Eigen fixed-size 3-term dot product compiled for the device: a0*b0 + (a1*b1 + a2*b2).
inline float dot3_dev(float a0, float a1, float a2, float b0, float b1, float b2) {
#if defined(ORACLE_SUM_ALT)
  return FMAF(a2, b2, FMAF(a1, b1, a0 * b0));  // (a0*b0 + a1*b1) + a2*b2
#else
  return FMAF(a0, b0, FMAF(a1, b1, a2 * b2));
#endif
}
// a*b - c*d on the device.
inline float dop_dev(float a, float b, float c, float d) { return FMAF(a, b, -(c * d)); }

struct V3 {
  float x, y, z;
};
inline V3 cross_dev(const V3& a, const V3& b) {  // Eigen cross(), OrthoMethods.h
  return {dop_dev(a.y, b.z, a.z, b.y), dop_dev(a.z, b.x, a.x, b.z), dop_dev(a.x, b.y, a.y, b.x)};
}
struct M3 {
  float m[3][3];
};
inline M3 skew(const V3& v) {  // gpu_utils.cuh:9-15
  M3 r;
  r.m[0][0] = 0;    r.m[0][1] = -v.z; r.m[0][2] = v.y;
  r.m[1][0] = v.z;  r.m[1][1] = 0;    r.m[1][2] = -v.x;
  r.m[2][0] = -v.y; r.m[2][1] = v.x;  r.m[2][2] = 0;
  return r;
}
inline M3 matmul_dev(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      r.m[i][j] = dot3_dev(a.m[i][0], a.m[i][1], a.m[i][2], b.m[0][j], b.m[1][j], b.m[2][j]);
  return r;
}
inline V3 matvec_dev(const M3& a, const V3& v) {
  return {dot3_dev(a.m[0][0], a.m[0][1], a.m[0][2], v.x, v.y, v.z),
          dot3_dev(a.m[1][0], a.m[1][1], a.m[1][2], v.x, v.y, v.z),
          dot3_dev(a.m[2][0], a.m[2][1], a.m[2][2], v.x, v.y, v.z)};
}

// squared_dist(const T&, const T&), gpu_utils.cuh:72-78 (device)
inline float squared_dist_xyz(const float* a, const float* b) {
  float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
  return FMAF(dz, dz, FMAF(dy, dy, dx * dx));
}
// squared_dist(const T*, const T*, int), gpu_utils.cuh:33-41 (device)
inline float squared_dist_n(const float* a, const float* b, int dim) {
  float result = 0;
  for (int i = 0; i < dim; i++) {
    float tmp = a[i] - b[i];
    result = FMAF(tmp, tmp, result);
  }
  return result;
}
inline float square_norm_n(const float* a, int dim) {  // gpu_utils.cuh:96-104
  float result = 0;
  for (int j = 0; j < dim; j++) result = FMAF(a[j], a[j], result);
  return result;
}
inline float dot_n(const float* a, const float* b, int dim) {  // gpu_utils.cuh:23-30
  float result = 0;
  for (int i = 0; i < dim; i++) result = FMAF(a[i], b[i], result);
  return result;
}
// compute_geometric_type_ip, CvoGPU.cu:203-215
inline float geometric_type_ip(const float* ga, const float* gb) {
  float norm2_a = square_norm_n(ga, 2);
  float norm2_b = square_norm_n(gb, 2);
  float dot_ab = dot_n(ga, gb, 2);
  return dot_ab * dot_ab / (norm2_a * norm2_b);
}
// compute_range_ell, CvoGPU.cu:86-90: ((dist)/500.0 + 1.0) * ell evaluated in double
inline float compute_range_ell(float curr_ell, float curr_dist_to_sensor) {
  return (float)(((double)curr_dist_to_sensor / 500.0 + 1.0) * (double)curr_ell);
}

const float kZeros[32] = {0};

struct CloudView {
  int n;
  const float *xyz, *feat, *label, *geo;
  const float* p(int i) const { return xyz + 3 * i; }
  const float* f(int i) const { return feat ? feat + FD * i : kZeros; }
  const float* l(int i) const { return label ? label + NC * i : kZeros; }
  const float* g(int i) const { return geo ? geo + 2 * i : kZeros; }
};
inline CloudView view(const OracleCloud* c) { return {c->n, c->xyz, c->feat, c->label, c->geo}; }

struct RowConsts {
  float sp_thres, sigma2, c2, c_sigma2, s_ell, s_sigma2;
  float l, d2_thres, d2_c_thres, d2_s_thres;
};

// Prologue of fill_in_A_mat_gpu, CvoGPU.cu:494-515
inline RowConsts row_consts(const OracleParams& P, const float* pa, float ell) {
  RowConsts r;
  r.sp_thres = P.sp_thres;
  r.sigma2 = P.sigma * P.sigma;
  r.c2 = P.c_ell * P.c_ell;
  r.c_sigma2 = P.c_sigma * P.c_sigma;
  r.s_ell = P.s_ell;
  r.s_sigma2 = P.s_sigma * P.s_sigma;
  float a_to_sensor = std::sqrt(FMAF(pa[2], pa[2], FMAF(pa[1], pa[1], pa[0] * pa[0])));
  r.l = compute_range_ell(ell, a_to_sensor);
  r.d2_thres = 1;
  r.d2_c_thres = 1;
  r.d2_s_thres = 1;
  if (P.is_using_geometry)
    r.d2_thres = (float)(-2.0 * r.l * r.l * (double)std::log(P.sp_thres / r.sigma2));
  if (P.is_using_intensity)
    r.d2_c_thres = (float)(-2.0 * r.c2 * (double)std::log(P.sp_thres / r.c_sigma2));
  if (P.is_using_semantics)
    r.d2_s_thres = (float)(-2.0 * r.s_ell * r.s_ell * (double)std::log(P.sp_thres / r.s_sigma2));
  return r;
}

// Body of the j-loop of fill_in_A_mat_gpu for one (i, j), CvoGPU.cu:528-573.
// Returns true and sets a if the pair survives all `continue`s (the a > sp_thres test is
// done by the caller, CvoGPU.cu:576).
inline bool pair_value(const OracleParams& P, const RowConsts& rc, const CloudView& X, int i,
                       const CloudView& Y, int j, float* a_out) {
  float a = 1, sk = 1, ck = 1, k = 1, geo_sim = 1;
  if (P.is_using_geometric_type) {
    geo_sim = geometric_type_ip(X.g(i), Y.g(j));
    if (geo_sim < 0.01) return false;
  }
  if (P.is_using_geometry) {
    float d2 = squared_dist_xyz(Y.p(j), X.p(i));
    if (d2 < rc.d2_thres)
      k = (float)((double)rc.sigma2 * std::exp((double)(-d2) / (2.0 * rc.l * rc.l)));
    else
      return false;
  }
  if (P.is_using_intensity) {
    float d2_color = squared_dist_n(X.f(i), Y.f(j), FD);
    if (d2_color < rc.d2_c_thres)
      ck = (float)((double)rc.c_sigma2 * std::exp((double)(-d2_color) / (2.0 * rc.c2)));
    else
      return false;
  }
  if (P.is_using_semantics) {
    float d2_semantic = squared_dist_n(X.l(i), Y.l(j), NC);
    if (d2_semantic < rc.d2_s_thres)
      sk = (float)((double)(P.s_sigma * P.s_sigma) *
                   std::exp((double)(-d2_semantic) / (2.0 * rc.s_ell * rc.s_ell)));
    else
      return false;
  }
  a = ck * k * sk * geo_sim;
  *a_out = a;
  return true;
}

// fill_in_A_mat_gpu for one row, literal form: CvoGPU.cu:477-593
inline unsigned se_row_literal(const OracleParams& P, const CloudView& X, int i, const CloudView& Y,
                               int K, float ell, float* mat_row, int* ind_row) {
  RowConsts rc = row_consts(P, X.p(i), ell);
  unsigned num_inds = 0;
  for (int j = 0; j < Y.n; j++) {
    if (num_inds == (unsigned)K) break;
    float a;
    if (!pair_value(P, rc, X, i, Y, j, &a)) continue;
    if (a > P.sp_thres) {
      mat_row[num_inds] = a;
      ind_row[num_inds] = j;
      num_inds++;
    }
  }
  return num_inds;
}

// Same result as se_row_literal, but the geometric cut-off (the only O(N*M) work) is
// evaluated blockwise so the compiler can vectorise it; survivors go through pair_value()
// in ascending j.  Valid because a pair failing `d2 < d2_thres` has no side effect.
// yx/yy/yz: SoA copies of the transformed target coordinates.
inline unsigned se_row_blocked(const OracleParams& P, const CloudView& X, int i, const CloudView& Y,
                               const float* yx, const float* yy, const float* yz, int K, float ell,
                               float* mat_row, int* ind_row) {
  RowConsts rc = row_consts(P, X.p(i), ell);
  unsigned num_inds = 0;
  const float ax = X.p(i)[0], ay = X.p(i)[1], az = X.p(i)[2];
  const float thr = rc.d2_thres;
  constexpr int BLK = 64;
  const int m = Y.n;
  for (int j0 = 0; j0 < m; j0 += BLK) {
    if (num_inds == (unsigned)K) break;
    const int jn = std::min(BLK, m - j0);
    unsigned char hit[BLK];
    int any = 0;
    for (int t = 0; t < jn; t++) {
      float dx = yx[j0 + t] - ax, dy = yy[j0 + t] - ay, dz = yz[j0 + t] - az;
      float d2 = FMAF(dz, dz, FMAF(dy, dy, dx * dx));
      unsigned char h = d2 < thr;
      hit[t] = h;
      any |= h;
    }
    if (!any) continue;
    for (int t = 0; t < jn; t++) {
      if (!hit[t]) continue;
      if (num_inds == (unsigned)K) break;
      float a;
      if (!pair_value(P, rc, X, i, Y, j0 + t, &a)) continue;
      if (a > P.sp_thres) {
        mat_row[num_inds] = a;
        ind_row[num_inds] = j0 + t;
        num_inds++;
      }
    }
  }
  return num_inds;
}

// ---- "best-effort CPU" variant: a uniform grid over the transformed targets ------------------
// The reference's own CPU code searches neighbours with a kd-tree (Cvo.cpp:368-381, CvoGPU.cpp:115-125); so that the
// GPU is not only timed against a dense scan, the oracle can also generate each row's candidates from the grid cells
// its cut-off sphere touches.  Candidates are visited in ascending j and go through the same pair_value(), so the
// result is identical to se_row_literal (checked in tests/test_oracle_numpy.py); only the time differs.
static int g_use_grid = 0;
static double g_scan_seconds = 0;  // time spent in se_kernel (the association scan) since the last reset: bench.py's per-stage split

struct TargetGrid {
  double ox, oy, oz, inv_cell;
  int nx, ny, nz;
  std::vector<int> start;  // CSR over cells
  std::vector<int> items;  // target indices, ascending inside a cell (stable counting sort)
  int cell_of(double v, double o, int n) const {
    int c = (int)std::floor((v - o) * inv_cell);
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
  }
};

static bool build_grid(const CloudView& Y, double cell, TargetGrid& G) {
  if (!(cell > 0) || Y.n == 0) return false;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int j = 0; j < Y.n; j++)
    for (int c = 0; c < 3; c++) {
      const double v = Y.p(j)[c];
      if (!(v == v) || std::isinf(v)) return false;
      lo[c] = std::min(lo[c], v);
      hi[c] = std::max(hi[c], v);
    }
  double ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
  while ((ext[0] / cell + 1) * (ext[1] / cell + 1) * (ext[2] / cell + 1) > 4e6) cell *= 2;  // bound the cell count
  G.ox = lo[0];
  G.oy = lo[1];
  G.oz = lo[2];
  G.inv_cell = 1.0 / cell;
  G.nx = (int)(ext[0] / cell) + 1;
  G.ny = (int)(ext[1] / cell) + 1;
  G.nz = (int)(ext[2] / cell) + 1;
  const size_t nc = (size_t)G.nx * G.ny * G.nz;
  G.start.assign(nc + 1, 0);
  std::vector<int> cell_of(Y.n);
  for (int j = 0; j < Y.n; j++) {
    const int c = (G.cell_of(Y.p(j)[2], G.oz, G.nz) * G.ny + G.cell_of(Y.p(j)[1], G.oy, G.ny)) * G.nx +
                  G.cell_of(Y.p(j)[0], G.ox, G.nx);
    cell_of[j] = c;
    G.start[c + 1]++;
  }
  for (size_t c = 0; c < nc; c++) G.start[c + 1] += G.start[c];
  G.items.resize(Y.n);
  std::vector<int> fill(G.start.begin(), G.start.end() - 1);
  for (int j = 0; j < Y.n; j++) G.items[fill[cell_of[j]]++] = j;
  return true;
}

// xq: the row's point in the frame the grid was built in; slack: absolute allowance on top of the cut-off radius for
// whatever separates that frame from the one the exact test runs in.
inline unsigned se_row_grid(const OracleParams& P, const CloudView& X, int i, const CloudView& Y, const TargetGrid& G,
                            int K, float ell, float* mat_row, int* ind_row, std::vector<int>& cand, const double xq[3],
                            double slack) {
  RowConsts rc = row_consts(P, X.p(i), ell);
  // every j with (float) d2 < d2_thres lies within this radius of x_i (1e-5 relative + absolute slack for the float
  // evaluation of d2)
  const double r = std::sqrt(std::max((double)rc.d2_thres, 0.0)) * (1.0 + 1e-5) + 1e-6 + slack;
  const double* x = xq;
  const int x0 = G.cell_of(x[0] - r, G.ox, G.nx), x1 = G.cell_of(x[0] + r, G.ox, G.nx);
  const int y0 = G.cell_of(x[1] - r, G.oy, G.ny), y1 = G.cell_of(x[1] + r, G.oy, G.ny);
  const int z0 = G.cell_of(x[2] - r, G.oz, G.nz), z1 = G.cell_of(x[2] + r, G.oz, G.nz);
  cand.clear();
  for (int cz = z0; cz <= z1; cz++)
    for (int cy = y0; cy <= y1; cy++) {
      const size_t row = ((size_t)cz * G.ny + cy) * G.nx;
      cand.insert(cand.end(), G.items.begin() + G.start[row + x0], G.items.begin() + G.start[row + x1 + 1]);
    }
  std::sort(cand.begin(), cand.end());  // ascending j: the reference's truncation / accumulation order
  unsigned num_inds = 0;
  for (int j : cand) {
    if (num_inds == (unsigned)K) break;
    float a;
    if (!pair_value(P, rc, X, i, Y, j, &a)) continue;
    if (a > P.sp_thres) {
      mat_row[num_inds] = a;
      ind_row[num_inds] = j;
      num_inds++;
    }
  }
  return num_inds;
}

// se_kernel (CvoGPU.cu:648-683) + reset_state_at_new_iter (CvoState.cu:143-157): rows are
// written from slot 0; slot nnz holds ind = -1, mat = 0 when nnz < K (what memset leaves).
// A grid that outlives the iteration (the align loop's): built over the INITIAL targets y0, queried with the row's point
// mapped into that frame, x' = R x + T (y_t = R^T (y0 - T) is an isometry of y0, so |y0 - x'| = |y_t - x| up to the
// float rounding of the transform: a few 1e-6 of |y|, covered by `slack`).  Candidates still go through pair_value()
// against the TRANSFORMED targets in ascending j: identical results, no O(M) rebuild per iteration.
struct PersistentGrid {
  TargetGrid G;
  double cell = 0;
  double ymax = 0;  // largest |coordinate| of the initial targets (sizes the rounding allowance of the frame change)
  bool valid = false;
};
void se_kernel_impl(const OracleParams& P, const CloudView& X, const CloudView& Y, int K, float ell,
                    float* mat, int* ind, unsigned* nonzeros, bool literal, PersistentGrid* pg = nullptr,
                    const CloudView* Y0 = nullptr, const float* Rpose = nullptr, const float* Tpose = nullptr) {
  const int n = X.n, m = Y.n;
  if (g_use_grid && !literal && P.is_using_geometry && n > 0 && m > 0) {
    // cell = the largest cut-off radius of any row: at most 3 x 3 x 3 cells per query
    double rmax = 0;
    for (int i = 0; i < n; i++) rmax = std::max(rmax, (double)row_consts(P, X.p(i), ell).d2_thres);
    const double rad = std::sqrt(std::max(rmax, 0.0));
    TargetGrid local;
    const TargetGrid* G = nullptr;
    double slack = 0, ymax = 0;
    if (pg && Y0 && Rpose && Tpose && rad > 0) {
      // (re)build when the cut-off outgrew the cells or shrank to well below them (ell decays during a solve)
      if (!pg->valid || rad > pg->cell || rad < 0.4 * pg->cell) {
        pg->valid = build_grid(*Y0, rad, pg->G);
        pg->cell = rad;
        pg->ymax = 0;
        for (int j = 0; j < m; j++)
          for (int c = 0; c < 3; c++) pg->ymax = std::max(pg->ymax, (double)std::fabs(Y0->p(j)[c]));
      }
      if (pg->valid) {
        G = &pg->G;
        ymax = pg->ymax;
        // The grid is queried at R x + T while the exact test measures |R^T (y0 - T) - x| (float transform of the targets,
        // float (R, T) <-> (R^T, -R^T T) pair): the two distances differ by the rounding of the float transform and by
        // however far the float R has drifted from orthonormality over the iterations so far - |R^T R - I|_F times the
        // extent of whatever it multiplies (source points, targets, T).  Both measured here, in double, per call.
        double xmax = 0, tmax = 0, drift = 0;
        for (int i = 0; i < n; i++)
          for (int c = 0; c < 3; c++) xmax = std::max(xmax, (double)std::fabs(X.p(i)[c]));
        for (int c = 0; c < 3; c++) tmax = std::max(tmax, (double)std::fabs(Tpose[c]));
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) {
            double d = (a == b) ? -1.0 : 0.0;
            for (int c = 0; c < 3; c++) d += (double)Rpose[3 * c + a] * (double)Rpose[3 * c + b];
            drift += d * d;
          }
        drift = std::sqrt(drift);
        slack = 1e-5 * (ymax + xmax + tmax + 1.0) + 1e-5 + 2.0 * drift * (ymax + xmax + tmax + 1.0);
      }
    } else if (rad > 0 && build_grid(Y, rad, local)) {
      G = &local;
    }
    if (G) {
#pragma omp parallel
      {
        std::vector<int> cand;
#pragma omp for schedule(dynamic, 64)
        for (int i = 0; i < n; i++) {
          float* mr = mat + (size_t)i * K;
          int* ir = ind + (size_t)i * K;
          const float* x = X.p(i);
          double xq[3] = {x[0], x[1], x[2]};
          if (pg && G == &pg->G)
            for (int c = 0; c < 3; c++)
              xq[c] = (double)Rpose[3 * c] * x[0] + (double)Rpose[3 * c + 1] * x[1] + (double)Rpose[3 * c + 2] * x[2] + (double)Tpose[c];
          unsigned nn = se_row_grid(P, X, i, Y, *G, K, ell, mr, ir, cand, xq, slack);
          if ((int)nn < K) {
            ir[nn] = -1;
            mr[nn] = 0;
          }
          nonzeros[i] = nn;
        }
      }
      return;
    }
  }
  std::vector<float> yx, yy, yz;
  const bool blocked = !literal && P.is_using_geometry;
  if (blocked) {
    yx.resize(m);
    yy.resize(m);
    yz.resize(m);
    for (int j = 0; j < m; j++) {
      yx[j] = Y.p(j)[0];
      yy[j] = Y.p(j)[1];
      yz[j] = Y.p(j)[2];
    }
  }
#pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < n; i++) {
    float* mr = mat + (size_t)i * K;
    int* ir = ind + (size_t)i * K;
    unsigned nn = blocked ? se_row_blocked(P, X, i, Y, yx.data(), yy.data(), yz.data(), K, ell, mr, ir)
                          : se_row_literal(P, X, i, Y, K, ell, mr, ir);
    if ((int)nn < K) {
      ir[nn] = -1;
      mr[nn] = 0;
    }
    nonzeros[i] = nn;
  }
}

// update_tf, CvoGPU.cu:94-126 (host; R row-major here)
void update_tf_impl(const float R[9], const float T[3], float Ri[9], float Ti[3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Ri[3 * i + j] = R[3 * j + i];
  for (int i = 0; i < 3; i++) {
    // (-R_inv) * T, Eigen 3-term product a0 + (a1 + a2) on the host (unfused)
    float a0 = (-Ri[3 * i + 0]) * T[0], a1 = (-Ri[3 * i + 1]) * T[1], a2 = (-Ri[3 * i + 2]) * T[2];
    Ti[i] = SUM3(a0, a1, a2);
  }
}

// transform_point_R_T::operator(), CvoGPU_impl.cu:31-82 (device functor, xyz only:
// update_normal_and_cov is false on the path, CvoGPU.cu:1408)
inline void transform_point(const float Ri[9], const float Ti[3], const float* in, float* out) {
  for (int i = 0; i < 3; i++)
    out[i] = dot3_dev(Ri[3 * i + 0], Ri[3 * i + 1], Ri[3 * i + 2], in[0], in[1], in[2]) + Ti[i];
}

struct FlowOut {
  float omega[3], v[3];
};

// compute_flow_gpu_no_eigen (CvoGPU.cu:729-790) + compute_flow host half (793-848)
FlowOut compute_flow_impl(const OracleParams& P, const CloudView& X, const CloudView& Y, int K,
                          const float* mat, const int* ind) {
  const int n = X.n;
  static thread_local std::vector<double> flow_buf;  // (reused across iterations; workers get raw pointers)
  flow_buf.resize(6 * (size_t)n);
  double* const om = flow_buf.data();
  double* const vv = om + 3 * (size_t)n;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    const float* px = X.p(i);
    V3 pxe{px[0], px[1], px[2]};
    float o0 = 0, o1 = 0, o2 = 0, v0 = 0, v1 = 0, v2 = 0;
    for (int j = 0; j < K; j++) {
      int idx = ind[(size_t)i * K + j];
      if (idx == -1) break;
      const float* py = Y.p(idx);
      V3 pye{py[0], py[1], py[2]};
      V3 cr = cross_dev(pxe, pye);
      float dx = pye.x - pxe.x, dy = pye.y - pxe.y, dz = pye.z - pxe.z;
      float a = mat[(size_t)i * K + j];
      o0 = FMAF(cr.x, a, o0);
      o1 = FMAF(cr.y, a, o1);
      o2 = FMAF(cr.z, a, o2);
      v0 = FMAF(dx, a, v0);
      v1 = FMAF(dy, a, v1);
      v2 = FMAF(dz, a, v2);
    }
    om[3 * (size_t)i + 0] = (double)(o0 / P.c);
    om[3 * (size_t)i + 1] = (double)(o1 / P.c);
    om[3 * (size_t)i + 2] = (double)(o2 / P.c);
    vv[3 * (size_t)i + 0] = (double)(v0 / P.d);
    vv[3 * (size_t)i + 1] = (double)(v1 / P.d);
    vv[3 * (size_t)i + 2] = (double)(v2 / P.d);
  }
  // thrust::reduce in double (order unspecified upstream; sequential here), CvoGPU.cu:824-825
  double so[3] = {0, 0, 0}, sv[3] = {0, 0, 0};
  for (int i = 0; i < n; i++)
    for (int c = 0; c < 3; c++) {
      so[c] += om[3 * (size_t)i + c];
      sv[c] += vv[3 * (size_t)i + c];
    }
  float ov[6] = {(float)so[0], (float)so[1], (float)so[2], (float)sv[0], (float)sv[1], (float)sv[2]};
  // Eigen::MatrixBase::normalize(): z = squaredNorm(); if (z > 0) *this /= sqrt(z);  (827-832)
  float z = 0;
  for (int c = 0; c < 6; c++) z = z + ov[c] * ov[c];
  if (z > 0) {
    float s = std::sqrt(z);
    for (int c = 0; c < 6; c++) ov[c] = ov[c] / s;
  }
  FlowOut f;
  for (int c = 0; c < 3; c++) {
    f.omega[c] = ov[c];
    f.v[c] = ov[3 + c];
  }
  return f;
}

struct XiMats {
  M3 oh, m2, m3, m4;
  V3 ohv, m2v, m3v;
};
inline XiMats xi_mats(const float omega[3], const float v[3]) {
  XiMats x;
  V3 w{omega[0], omega[1], omega[2]}, vv{v[0], v[1], v[2]};
  x.oh = skew(w);
  x.m2 = matmul_dev(x.oh, x.oh);  // (omega_hat*omega_hat) evaluated into a temporary
  x.m3 = matmul_dev(x.m2, x.oh);
  x.m4 = matmul_dev(x.m3, x.oh);
  x.ohv = matvec_dev(x.oh, vv);
  x.m2v = matvec_dev(x.m2, vv);
  x.m3v = matvec_dev(x.m3, vv);
  return x;
}
struct XiZ {
  V3 xiz, xi2z, xi3z, xi4z;
  float normxiz2, xiz_dot_xi2z, epsil_const;
};
// compute_step_size_xi for one target, CvoGPU.cu:953-998
inline XiZ xi_point(const XiMats& M, const float omega[3], const float v[3], const float* y) {
  XiZ r;
  V3 w{omega[0], omega[1], omega[2]}, yy{y[0], y[1], y[2]};
  V3 c = cross_dev(w, yy);
  r.xiz = {c.x + v[0], c.y + v[1], c.z + v[2]};
  V3 a = matvec_dev(M.m2, yy);
  r.xi2z = {a.x + M.ohv.x, a.y + M.ohv.y, a.z + M.ohv.z};
  a = matvec_dev(M.m3, yy);
  r.xi3z = {a.x + M.m2v.x, a.y + M.m2v.y, a.z + M.m2v.z};
  a = matvec_dev(M.m4, yy);
  r.xi4z = {a.x + M.m3v.x, a.y + M.m3v.y, a.z + M.m3v.z};
  r.normxiz2 = dot3_dev(r.xiz.x, r.xiz.y, r.xiz.z, r.xiz.x, r.xiz.y, r.xiz.z);
  r.xiz_dot_xi2z = -dot3_dev(r.xiz.x, r.xiz.y, r.xiz.z, r.xi2z.x, r.xi2z.y, r.xi2z.z);
  r.epsil_const = FMAF(2.0f, dot3_dev(r.xiz.x, r.xiz.y, r.xiz.z, r.xi3z.x, r.xi3z.y, r.xi3z.z),
                            dot3_dev(r.xi2z.x, r.xi2z.y, r.xi2z.z, r.xi2z.x, r.xi2z.y, r.xi2z.z));
  return r;
}

struct Coefs {
  double B, C, D, E;
};
// compute_step_size_poly_coeff (CvoGPU.cu:1001-1082) + the four thrust::reduce (1118-1121)
Coefs poly_coeff_impl(const OracleParams& P, const CloudView& X, const CloudView& Y, int K, float ell,
                      const float* mat, const int* ind, const float omega[3], const float v[3]) {
  const int n = X.n, m = Y.n;
  XiMats M = xi_mats(omega, v);
  // compute_step_size_xi runs over ALL targets upstream (CvoGPU.cu:953-998); the "best-effort CPU" variant evaluates it
  // per nonzero instead (the same function of the same inputs: identical values, O(nnz) instead of O(M))
  const bool per_entry = g_use_grid != 0;
  // (buffers of the calling thread, reused across iterations; the OpenMP workers get raw pointers: a thread_local
  // named inside a parallel region would be the WORKER's own, empty, instance)
  static thread_local std::vector<XiZ> xz_buf;
  static thread_local std::vector<double> coef_buf;
  if (!per_entry) xz_buf.resize(m);
  coef_buf.resize(4 * (size_t)n);
  XiZ* const xz = xz_buf.data();
  double* const Bv = coef_buf.data();
  double* const Cv = Bv + n;
  double* const Dv = Cv + n;
  double* const Ev = Dv + n;
  if (!per_entry) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < m; j++) xz[j] = xi_point(M, omega, v, Y.p(j));
  }
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) {
    double Bi = 0, Ci = 0, Di = 0, Ei = 0;
    const float* px = X.p(i);
    float d2_sqrt = std::sqrt(dot3_dev(px[0], px[1], px[2], px[0], px[1], px[2]));  // px.norm()
    float temp_ell = ell;
    if (P.is_using_range_ell) temp_ell = compute_range_ell(ell, d2_sqrt);
    for (int j = 0; j < K; j++) {
      int idx = ind[(size_t)i * K + j];
      if (idx == -1) break;
      float temp_coef = (float)(1 / (2.0 * temp_ell * temp_ell));
      const float* py = Y.p(idx);
      float dfx = px[0] - py[0], dfy = px[1] - py[1], dfz = px[2] - py[2];
      const XiZ zloc = per_entry ? xi_point(M, omega, v, py) : XiZ{};
      const XiZ& z = per_entry ? zloc : xz[idx];
      float beta_ij = (float)(-2.0 * temp_coef * (double)dot3_dev(z.xiz.x, z.xiz.y, z.xiz.z, dfx, dfy, dfz));
      float gamma_ij =
          (-temp_coef) * (z.normxiz2 + dot3_dev(2.0f * z.xi2z.x, 2.0f * z.xi2z.y, 2.0f * z.xi2z.z, dfx, dfy, dfz));
      float delta_ij = (float)(2.0 * temp_coef *
                               (double)(z.xiz_dot_xi2z + dot3_dev(-z.xi3z.x, -z.xi3z.y, -z.xi3z.z, dfx, dfy, dfz)));
      float epsil_ij =
          (-temp_coef) * (z.epsil_const + dot3_dev(2.0f * z.xi4z.x, 2.0f * z.xi4z.y, 2.0f * z.xi4z.z, dfx, dfy, dfz));
      float A_ij = mat[(size_t)i * K + j];
      double bi = (double)(A_ij * beta_ij);
      Bi += bi;
      double ci = (double)A_ij * ((double)gamma_ij + (double)(beta_ij * beta_ij) / 2.0);
      Ci += ci;
      double di = (double)A_ij * ((double)FMAF(beta_ij, gamma_ij, delta_ij) +
                                  (double)(beta_ij * beta_ij * beta_ij) / 6.0);
      Di += di;
      double ei = (double)A_ij * ((double)FMAF(beta_ij, delta_ij, epsil_ij) +
                                  1 / 2.0 * beta_ij * beta_ij * gamma_ij + 1 / 2.0 * gamma_ij * gamma_ij +
                                  1 / 24.0 * beta_ij * beta_ij * beta_ij * beta_ij);
      Ei += ei;
    }
    Bv[i] = Bi;
    Cv[i] = Ci;
    Dv[i] = Di;
    Ev[i] = Ei;
  }
  Coefs c{0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    c.B += Bv[i];
    c.C += Cv[i];
    c.D += Dv[i];
    c.E += Ev[i];
  }
  return c;
}

#define CUBIC_QUAL static
#define CUBIC_NAME cubic_roots_impl
#define CUBIC_FABS std::fabs
#define CUBIC_SQRT std::sqrt
#define CUBIC_ISFINITE std::isfinite
#define CUBIC_NAN std::numeric_limits<double>::quiet_NaN()
// Roots of p0 x^3 + p1 x^2 + p2 x + p3.  The reference takes the eigenvalues of the companion
// matrix with Eigen 3.3.9's EigenSolver (LieGroup.cpp:309-325; Eigen is not in the repository).
// Restated with a backward-stable scheme that uses only + - * / sqrt (so CPU and GPU agree
// bitwise): each real root is bracketed next to a critical point of the monic cubic (the outer
// brackets grow by doubling) and refined by safeguarded Newton (bisection fallback); a complex
// pair follows from Vieta.  Pinned against numpy.roots (LAPACK companion eigenvalues) in
// tests/test_oracle_math.py.
CUBIC_QUAL double cubic_solve_bracket(double a, double b, double c, double lo, double hi) {
  auto f = [&](double x) { return ((x + a) * x + b) * x + c; };
  auto df = [&](double x) { return (3.0 * x + 2.0 * a) * x + b; };
  double fl = f(lo), fh = f(hi);
  if (fl == 0.0) return lo;
  if (fh == 0.0) return hi;
  double xl, xh;
  if (fl < 0.0) {
    xl = lo;
    xh = hi;
  } else {
    xl = hi;
    xh = lo;
  }
  double x = 0.5 * (lo + hi);
  double dxold = CUBIC_FABS(hi - lo), dx = dxold;
  double fx = f(x), dfx = df(x);
  for (int it = 0; it < 200; it++) {
    if ((((x - xh) * dfx - fx) * ((x - xl) * dfx - fx) > 0.0) || (CUBIC_FABS(2.0 * fx) > CUBIC_FABS(dxold * dfx))) {
      dxold = dx;
      dx = 0.5 * (xh - xl);
      x = xl + dx;
      if (xl == x) return x;
    } else {
      dxold = dx;
      dx = fx / dfx;
      const double tmp = x;
      x -= dx;
      if (tmp == x) return x;
    }
    if (CUBIC_FABS(dx) <= 4.5e-16 * CUBIC_FABS(x)) return x;  // converged to ~2 ulp (Newton can ping-pong there)
    fx = f(x);
    dfx = df(x);
    if (fx == 0.0) return x;
    if (fx < 0.0)
      xl = x;
    else
      xh = x;
  }
  return x;
}

// Root of the monic cubic on the unbounded side of x0 (dir = +1: right of x0 where f(x0) <= 0 and f
// increases to +inf; dir = -1: left of x0 where f(x0) >= 0 and f decreases to -inf).  The bracket is
// grown by doubling so that Newton starts within a factor ~2 of the root.
CUBIC_QUAL double cubic_solve_outward(double a, double b, double c, double x0, double dir, double bound) {
  auto f = [&](double x) { return ((x + a) * x + b) * x + c; };
  double h = CUBIC_FABS(x0) * 0.5;
  if (h < 1e-3) h = 1e-3;
  double prev = x0;
  for (int it = 0; it < 1100; it++) {
    double x = x0 + dir * h;
    if (CUBIC_FABS(x) > bound) x = dir * bound;
    const double fx = f(x);
    if ((dir > 0.0) ? (fx >= 0.0) : (fx <= 0.0)) return dir > 0.0 ? cubic_solve_bracket(a, b, c, prev, x) : cubic_solve_bracket(a, b, c, x, prev);
    if (CUBIC_FABS(x) >= bound) return x;  // cannot happen for a finite cubic (Cauchy bound)
    prev = x;
    h *= 2.0;
  }
  return prev;
}

CUBIC_QUAL void CUBIC_NAME(const double coef[4], double re[3], double im[3]) {
  const double nan = CUBIC_NAN;
  const double a = coef[1] / coef[0], b = coef[2] / coef[0], c = coef[3] / coef[0];
  if (!CUBIC_ISFINITE(a) || !CUBIC_ISFINITE(b) || !CUBIC_ISFINITE(c)) {
    for (int i = 0; i < 3; i++) re[i] = im[i] = nan;
    return;
  }
  auto f = [&](double x) { return ((x + a) * x + b) * x + c; };
  double bound = CUBIC_FABS(a);
  if (CUBIC_FABS(b) > bound) bound = CUBIC_FABS(b);
  if (CUBIC_FABS(c) > bound) bound = CUBIC_FABS(c);
  bound = 1.0 + bound;  // Cauchy bound on |root|
  double r[3] = {0.0, 0.0, 0.0};
  int nr = 0;
  const double dq = a * a - 3.0 * b;
  if (!(dq > 0.0)) {  // monotone: one real root, on the side of the inflection point the sign says
    const double xi = -a / 3.0;
    const double fi = f(xi);
    r[nr++] = (fi == 0.0) ? xi : (fi < 0.0 ? cubic_solve_outward(a, b, c, xi, 1.0, bound)
                                           : cubic_solve_outward(a, b, c, xi, -1.0, bound));
  } else {
    const double s = CUBIC_SQRT(dq);
    const double t = (a >= 0.0) ? (-a - s) : (-a + s);
    const double xa = t / 3.0, xb = (t != 0.0) ? b / t : 0.0;
    const double x1 = xa < xb ? xa : xb, x2 = xa < xb ? xb : xa;
    const double f1 = f(x1), f2 = f(x2);
    if (f1 >= 0.0) r[nr++] = (f1 == 0.0) ? x1 : cubic_solve_outward(a, b, c, x1, -1.0, bound);
    if (f1 > 0.0 && f2 < 0.0) r[nr++] = cubic_solve_bracket(a, b, c, x1, x2);
    if (f2 <= 0.0) r[nr++] = (f2 == 0.0) ? x2 : cubic_solve_outward(a, b, c, x2, 1.0, bound);
  }
  if (nr == 3) {
    for (int i = 0; i < 3; i++) {
      re[i] = r[i];
      im[i] = 0.0;
    }
  } else if (nr == 2) {  // an exact double root at a critical point
    re[0] = r[0];
    re[1] = r[1];
    const double dbl = -a - r[0] - r[1];
    re[2] = dbl;
    im[0] = im[1] = im[2] = 0.0;
  } else {
    const double r0 = r[0];
    // complex pair z, conj(z) from Vieta: r0 + 2 Re z = -a, 2 r0 Re z + |z|^2 = b, r0 |z|^2 = -c;
    // use the pair of relations that does not cancel for the size of r0
    double rr, mod2;
    if (CUBIC_FABS(r0) >= 0.5 * CUBIC_FABS(a) && r0 != 0.0) {
      mod2 = -c / r0;
      rr = (b - mod2) / (2.0 * r0);
    } else {
      rr = 0.5 * (-a - r0);
      mod2 = b - 2.0 * r0 * rr;
    }
    const double ii = mod2 - rr * rr;
    re[0] = r0;
    im[0] = 0.0;
    re[1] = re[2] = rr;
    im[1] = ii > 0.0 ? CUBIC_SQRT(ii) : 0.0;
    im[2] = -im[1];
  }
}

// compute_step_size host half, CvoGPU.cu:1122-1158 (including the overwrite quirk: when no
// root is admissible temp_step stays DBL_MAX and the `> max_step` branch wins).
float select_step_impl(double B, double C, double D, double E, float min_step, float max_step) {
  double p_coef[4] = {4.0 * E, 3.0 * D, 2.0 * C, B};
  double re[3], im[3];
  cubic_roots_impl(p_coef, re, im);
  double temp_step = std::numeric_limits<double>::max();
  for (int i = 0; i < 3; i++)
    if (re[i] > 0 && re[i] < temp_step && std::fabs(im[i]) < 1e-5) temp_step = re[i];
  float step = temp_step == std::numeric_limits<double>::max() ? min_step : (float)temp_step;
  if (temp_step > max_step)
    step = max_step;
  else if (temp_step < min_step)
    step = min_step;
  else
    step = (float)temp_step;
  return step;
}

// Exp_SEK3(Matrix<float,6,1>, float dt), LieGroup.cpp:244-274 (host, float, unfused).
// out: 3x4 row-major [R | Jl*v].
void exp_sek3_impl(const float xi[6], float dt, float out[12]) {
  const float TOLERANCE = 1e-6f;  // LieGroup.cpp:9
  float w0 = xi[0], w1 = xi[1], w2 = xi[2];
  float theta = std::sqrt(SUM3(w0 * w0, w1 * w1, w2 * w2));
  float R[3][3], Jl[3][3];
  const float I[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (theta < TOLERANCE) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[i][j] = Jl[i][j] = I[i][j];
  } else {
    float A[3][3] = {{0, -w2, w1}, {w2, 0, -w0}, {-w1, w0, 0}};
    float theta2 = theta * theta;
    float stheta = std::sin(dt * theta);
    float ctheta = std::cos(dt * theta);
    float oneMinusCosTheta2 = (1 - ctheta) / (theta2);
    float A2[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) A2[i][j] = SUM3(A[i][0] * A[0][j], A[i][1] * A[1][j], A[i][2] * A[2][j]);
    float s1 = stheta / theta;
    float s3 = (dt * theta - stheta) / (theta2 * theta);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        R[i][j] = (I[i][j] + s1 * A[i][j]) + oneMinusCosTheta2 * A2[i][j];
        Jl[i][j] = (dt * I[i][j] + oneMinusCosTheta2 * A[i][j]) + s3 * A2[i][j];
      }
  }
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) out[4 * i + j] = R[i][j];
    out[4 * i + 3] = SUM3(Jl[i][0] * xi[3], Jl[i][1] * xi[4], Jl[i][2] * xi[5]);
  }
}

// ||Sophus::SE3d(dRT).log()||, CvoGPU.cu:1473-1476.  Sophus 1.0.0 is not in the repository;
// restated from its published algorithm: SO3(R) = Eigen::Quaterniond(R) (no normalisation),
// SO3::logAndTheta via 2*atan(n/w)/n, SE3::log via V^-1.  Pinned against scipy logm.
double se3_log_norm_impl(const double R[9], const double t[3]) {
  const double eps = 1e-10;  // Sophus::Constants<double>::epsilon()
  double q[4];               // x y z w
  auto m = [&](int i, int j) { return R[3 * i + j]; };
  double tr = m(0, 0) + m(1, 1) + m(2, 2);
  if (tr > 0) {
    double s = std::sqrt(tr + 1.0);
    q[3] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (m(2, 1) - m(1, 2)) * s;
    q[1] = (m(0, 2) - m(2, 0)) * s;
    q[2] = (m(1, 0) - m(0, 1)) * s;
  } else {
    int i = 0;
    if (m(1, 1) > m(0, 0)) i = 1;
    if (m(2, 2) > m(i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
    q[i] = 0.5 * s;
    s = 0.5 / s;
    q[3] = (m(k, j) - m(j, k)) * s;
    q[j] = (m(j, i) + m(i, j)) * s;
    q[k] = (m(k, i) + m(i, k)) * s;
  }
  double squared_n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  double n = std::sqrt(squared_n);
  double w = q[3];
  double two_atan_nbyw_by_n;
  if (n < eps) {
    double squared_w = w * w;
    two_atan_nbyw_by_n = 2.0 / w - 2.0 * squared_n / (w * squared_w);
  } else {
    if (std::fabs(w) < eps) {
      two_atan_nbyw_by_n = (w > 0 ? M_PI : -M_PI) / n;
    } else {
      two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
    }
  }
  double theta = two_atan_nbyw_by_n * n;
  double om[3] = {two_atan_nbyw_by_n * q[0], two_atan_nbyw_by_n * q[1], two_atan_nbyw_by_n * q[2]};
  double O[3][3] = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}};
  double O2[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) O2[i][j] = SUM3L(O[i][0] * O[0][j], O[i][1] * O[1][j], O[i][2] * O[2][j]);
  double coef;
  if (std::fabs(theta) < eps) {
    coef = 1.0 / 12.0;
  } else {
    double half_theta = 0.5 * theta;
    coef = (1.0 - theta * std::cos(half_theta) / (2.0 * std::sin(half_theta))) / (theta * theta);
  }
  double u[3];
  for (int i = 0; i < 3; i++) {
    double Vinv[3];
    for (int j = 0; j < 3; j++) Vinv[j] = (i == j ? 1.0 : 0.0) - 0.5 * O[i][j] + coef * O2[i][j];
    u[i] = SUM3L(Vinv[0] * t[0], Vinv[1] * t[1], Vinv[2] * t[2]);
  }
  double s = 0;
  for (int i = 0; i < 3; i++) s += u[i] * u[i];
  for (int i = 0; i < 3; i++) s += om[i] * om[i];
  return std::sqrt(s);
}

// A_sparsity_indicator_ell_update, CvoGPU.cu:1167-1285 (literal control flow: the three
// `if`s are sequential, not else-if).
struct Indicator {
  std::queue<float> start_q, end_q;
  float start_sum = 0, end_sum = 0;
  bool update(float indicator, int queue_len, float thr) {
    bool decrease = false;
    if ((int)start_q.size() < queue_len) {
      start_q.push(indicator);
      start_sum += indicator;
    }
    if ((int)start_q.size() >= queue_len && (int)end_q.size() < queue_len) {
      end_q.push(indicator);
      end_sum += indicator;
    }
    if ((int)start_q.size() >= queue_len && (int)end_q.size() >= queue_len) {
      if (end_sum / start_sum > 1 - thr && end_sum / start_sum < 1 + thr) {
        decrease = true;
        std::queue<float> e1, e2;
        std::swap(start_q, e1);
        std::swap(end_q, e2);
        start_sum = 0;
        end_sum = 0;
      } else {
        end_sum -= end_q.front();
        start_sum += end_q.front();
        start_q.push(end_q.front());
        end_q.pop();
        start_sum -= start_q.front();
        start_q.pop();
        end_q.push(indicator);
        end_sum += indicator;
      }
    }
    return decrease;
  }
};

struct IterResult {
  int status;  // 0 continue, 1 break (eps), 2 break (eps_2)
  int ret;
};

struct Workspace {
  std::vector<float> yt;  // transformed target xyz
  std::vector<float> mat;
  std::vector<int> ind;
  std::vector<unsigned> nonzeros;
  // literal_buffers: mat / ind are the reference's persistent rows x nearest_neighbors_max device buffers
  // (CvoState.cu:74-83), cleared over their first rows * num_neighbors entries at the top of every iteration
  // (reset_state_at_new_iter -> clear_SparseKernelMat(A, num_neighbors), CvoState.cu:143-146,
  // SparseKernelMat.cu:87-95) and written with row stride num_neighbors (CvoGPU.cu:577-578).  Only the association
  // export of align() can tell the difference (it may read them with another stride, see export_align_association).
  bool literal_buffers = false;
  int K_alloc = 0;
  PersistentGrid grid;  // "best-effort CPU" variant: a grid over the initial targets, kept across iterations
};

// One pass of the loop body of align_impl, CvoGPU.cu:1387-1531, up to (not including) the
// indicator / ell / K bookkeeping, which the caller does.
IterResult iterate(const OracleParams& P, const CloudView& X, const CloudView& Y0, float R[9], float T[3],
                   float ell, int K, Workspace& ws, OracleTrace* tr) {
  const int n = X.n, m = Y0.n;
  float Ri[9], Ti[3];
  update_tf_impl(R, T, Ri, Ti);  // CvoGPU.cu:1393
  ws.yt.resize(3 * (size_t)m);
#pragma omp parallel for schedule(static)
  for (int j = 0; j < m; j++) transform_point(Ri, Ti, Y0.p(j), &ws.yt[3 * (size_t)j]);  // 1404
  CloudView Y = Y0;
  Y.xyz = ws.yt.data();
  if (ws.literal_buffers) {
    if (ws.mat.size() != (size_t)n * ws.K_alloc) {  // cudaMalloc'ed, first cleared by iteration 0 (K == K_alloc there)
      ws.mat.assign((size_t)n * ws.K_alloc, 0.0f);
      ws.ind.assign((size_t)n * ws.K_alloc, -1);
    }
    std::fill(ws.mat.begin(), ws.mat.begin() + (size_t)n * K, 0.0f);
    std::fill(ws.ind.begin(), ws.ind.begin() + (size_t)n * K, -1);
  } else if (ws.mat.size() < (size_t)n * K) {
    ws.mat.resize((size_t)n * K);
    ws.ind.resize((size_t)n * K);
  }
  ws.nonzeros.resize(n);
  {
    const auto ts = std::chrono::steady_clock::now();
    se_kernel_impl(P, X, Y, K, ell, ws.mat.data(), ws.ind.data(), ws.nonzeros.data(), false, &ws.grid, &Y0, R, T);  // 1419
    g_scan_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count();
  }
  unsigned nnz = 0, mx = 0;  // compute_nonzeros SparseKernelMat.cu:37-46; max_element CvoGPU.cu:1518
  for (int i = 0; i < n; i++) {
    nnz += ws.nonzeros[i];
    mx = std::max(mx, ws.nonzeros[i]);
  }
  FlowOut f = compute_flow_impl(P, X, Y, K, ws.mat.data(), ws.ind.data());  // 1443
  Coefs c = poly_coeff_impl(P, X, Y, K, ell, ws.mat.data(), ws.ind.data(), f.omega, f.v);
  float step = select_step_impl(c.B, c.C, c.D, c.E, P.min_step, P.max_step);  // 1451
  if (tr) {
    tr->K = K;
    tr->ell = ell;
    tr->step = step;
    tr->nnz = nnz;
    tr->max_nnz = mx;
    for (int k = 0; k < 3; k++) {
      tr->omega[k] = f.omega[k];
      tr->v[k] = f.v[k];
    }
    tr->B = c.B;
    tr->C = c.C;
    tr->D = c.D;
    tr->E = c.E;
    tr->dist = 0;
  }
  auto norm3d = [](const float* a) {
    double x = a[0], y = a[1], z = a[2];
    return std::sqrt(SUM3(x * x, y * y, z * z));
  };
  IterResult res{0, 0};
  if (norm3d(f.omega) < P.eps && norm3d(f.v) < P.eps) {  // 1454-1458
    auto norm3f = [](const float* a) { return std::sqrt(SUM3(a[0] * a[0], a[1] * a[1], a[2] * a[2])); };
    if (norm3f(f.omega) < 1e-8 && norm3f(f.v) < 1e-8) res.ret = -1;
    res.status = 1;
    if (tr) {
      std::memcpy(tr->R, R, sizeof(float) * 9);
      std::memcpy(tr->T, T, sizeof(float) * 3);
    }
    return res;
  }
  float xi[6] = {f.omega[0], f.omega[1], f.omega[2], f.v[0], f.v[1], f.v[2]};
  float dtrans[12];
  exp_sek3_impl(xi, step, dtrans);  // 1462
  double dR[9], dT[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) dR[3 * i + j] = (double)dtrans[4 * i + j];
    dT[i] = (double)dtrans[4 * i + 3];
  }
  // T = (R.cast<double>() * dT + T.cast<double>()).cast<float>();  R = (R.cast<double>() * dR).cast<float>()
  float Tn[3], Rn[9];
  for (int i = 0; i < 3; i++) {
    double r0 = R[3 * i + 0], r1 = R[3 * i + 1], r2 = R[3 * i + 2];
    Tn[i] = (float)(SUM3(r0 * dT[0], r1 * dT[1], r2 * dT[2]) + (double)T[i]);
    for (int j = 0; j < 3; j++) Rn[3 * i + j] = (float)SUM3(r0 * dR[0 + j], r1 * dR[3 + j], r2 * dR[6 + j]);
  }
  std::memcpy(R, Rn, sizeof(Rn));
  std::memcpy(T, Tn, sizeof(Tn));
  double dist = se3_log_norm_impl(dR, dT);  // 1473-1476
  if (tr) {
    tr->dist = dist;
    std::memcpy(tr->R, R, sizeof(float) * 9);
    std::memcpy(tr->T, T, sizeof(float) * 3);
  }
  if (dist < P.eps_2) res.status = 2;  // 1505-1508 (checked by the caller after the indicator update)
  return res;
}

void mat4_from_RT_inverse(const float R[9], const float T[3], float out_colmajor[16]) {
  // final update_tf(R, T, &cvo_state, transform), CvoGPU.cu:1562
  float Ri[9], Ti[3];
  update_tf_impl(R, T, Ri, Ti);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) out_colmajor[4 * j + i] = Ri[3 * i + j];
    out_colmajor[12 + i] = Ti[i];
  }
  out_colmajor[3] = out_colmajor[7] = out_colmajor[11] = 0;
  out_colmajor[15] = 1;
}

}  // namespace

extern "C" {

void oracle_cubic_roots(const double coef[4], double re[3], double im[3]) { cubic_roots_impl(coef, re, im); }
float oracle_select_step(double B, double C, double D, double E, float min_step, float max_step) {
  return select_step_impl(B, C, D, E, min_step, max_step);
}
void oracle_exp_sek3(const float xi[6], float dt, float out[12]) { exp_sek3_impl(xi, dt, out); }
double oracle_se3_log_norm(const double dR[9], const double dT[3]) { return se3_log_norm_impl(dR, dT); }
void oracle_indicator_run(const float* indicators, int n, int window, float thr, unsigned char* decisions) {
  Indicator ind;
  for (int i = 0; i < n; i++) decisions[i] = ind.update(indicators[i], window, thr) ? 1 : 0;
}
void oracle_update_tf(const float R[9], const float T[3], float R_inv[9], float T_inv[3]) {
  update_tf_impl(R, T, R_inv, T_inv);
}
void oracle_transform(const float R_inv[9], const float T_inv[3], int m, const float* y0, float* yt) {
  for (int j = 0; j < m; j++) transform_point(R_inv, T_inv, y0 + 3 * j, yt + 3 * j);
}
// transform_point_pose_vec (CvoGPU_impl.cu:85-161), used by CvoFrameGPU::transform_pointcloud (CvoFrameGPU.cu:44-61)
// for the multi-frame edge kernel.  `Eigen::Vector3f trans = T * input` with T 3x4 row-major and input (x, y, z, 1):
// Eigen's unrolled 4-term inner product is (c0 + c1) + (c2 + c3); under nvcc -fmad=true the first product of each
// sum is assumed to be the fused one and T3 * 1.0f folds to T3.  (Assumption, like every device-side contraction
// here: no reference build exists to pin it.)
void oracle_transform_pose_vec(const float pose12[12], int n, const float* xyz_in, float* xyz_out) {
  const float* T = pose12;
  for (int i = 0; i < n; i++) {
    const float x = xyz_in[3 * i], y = xyz_in[3 * i + 1], z = xyz_in[3 * i + 2];
    for (int r = 0; r < 3; r++)
      xyz_out[3 * i + r] = FMAF(T[4 * r], x, T[4 * r + 1] * y) + FMAF(T[4 * r + 2], z, T[4 * r + 3]);
  }
}
void oracle_se_kernel(const OracleParams* p, const OracleCloud* x, const OracleCloud* y, int K, float ell,
                      float* mat, int* ind, unsigned int* nonzeros, int literal) {
  // reset_state_at_new_iter: mat = 0, ind = -1 over rows*K (CvoState.cu:143-157)
  std::fill(mat, mat + (size_t)x->n * K, 0.0f);
  std::fill(ind, ind + (size_t)x->n * K, -1);
  se_kernel_impl(*p, view(x), view(y), K, ell, mat, ind, nonzeros, literal != 0);
}

int oracle_iteration(const OracleParams* p, const OracleCloud* x, const OracleCloud* y, float R[9], float T[3],
                     float ell, int K, OracleTrace* out, int* ret_code, float* ell_mat_out, int* ell_ind_out,
                     unsigned int* nonzeros_out) {
  Workspace ws;
  IterResult r = iterate(*p, view(x), view(y), R, T, ell, K, ws, out);
  if (ret_code) *ret_code = r.ret;
  size_t nk = (size_t)x->n * K;
  if (ell_mat_out) {
    std::fill(ell_mat_out, ell_mat_out + nk, 0.0f);
    for (int i = 0; i < x->n; i++)
      for (unsigned s = 0; s < ws.nonzeros[i]; s++) ell_mat_out[(size_t)i * K + s] = ws.mat[(size_t)i * K + s];
  }
  if (ell_ind_out) {
    std::fill(ell_ind_out, ell_ind_out + nk, -1);
    for (int i = 0; i < x->n; i++)
      for (unsigned s = 0; s < ws.nonzeros[i]; s++) ell_ind_out[(size_t)i * K + s] = ws.ind[(size_t)i * K + s];
  }
  if (nonzeros_out) std::memcpy(nonzeros_out, ws.nonzeros.data(), sizeof(unsigned) * x->n);
  return r.status;
}

// CvoGPU::align + align_impl, CvoGPU.cu:1338-1632
static int align_core(const OracleParams* p, const OracleCloud* x, const OracleCloud* y, const float init[16],
                 float out[16], int* iterations, OracleTrace* trace, int max_trace, int trace_dense,
                 int trace_every, int* n_trace, double* seconds, int max_iter_override, OracleAssociation* assoc) {
  if (assoc) assoc->n_pairs = assoc->n_source_inliers = assoc->K_used = assoc->K_final = assoc->overflow = 0;
  if (n_trace) *n_trace = 0;
  if (iterations) *iterations = 0;
  if (x->n == 0 || y->n == 0) return 0;  // CvoGPU.cu:1614-1617: transform untouched
  const OracleParams& P = *p;
  CloudView X = view(x), Y0 = view(y);
  float R[9], T[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = init[4 * j + i];
    T[i] = init[12 + i];
  }
  auto t0 = std::chrono::steady_clock::now();
  int ret = 0;
  Indicator indicator;
  float ell = P.ell_init;  // CvoState ctor, CvoState.cu:30
  int num_neighbors = P.nearest_neighbors_max;
  int max_iter = max_iter_override > 0 ? std::min(max_iter_override, P.MAX_ITER) : P.MAX_ITER;
  Workspace ws;
  ws.literal_buffers = assoc != nullptr;
  ws.K_alloc = P.nearest_neighbors_max;
  int k = 0;
  int nt = 0;
  int K_used = num_neighbors;
  unsigned last_nnz = 0;
  for (; k < max_iter; k++) {
    OracleTrace tr;
    std::memset(&tr, 0, sizeof(tr));
    tr.k = k;
    K_used = num_neighbors;
    IterResult r = iterate(P, X, Y0, R, T, ell, num_neighbors, ws, &tr);
    last_nnz = tr.nnz;
    bool rec = trace && nt < max_trace && (k < trace_dense || (trace_every > 0 && k % trace_every == 0));
    if (rec) trace[nt++] = tr;
    if (r.status == 1) {
      ret = r.ret;
      break;
    }
    float ip_curr = (float)((double)tr.nnz / std::sqrt((double)X.n * (double)Y0.n));  // 1486
    bool need_decay_ell = indicator.update(ip_curr, P.indicator_window_size, P.indicator_stable_threshold);
    if (r.status == 2) break;  // 1505-1508
    if (k > P.ell_decay_start && need_decay_ell) {  // 1509-1513
      ell = ell * P.ell_decay_rate;
      if (ell < P.ell_min) ell = P.ell_min;
    }
    num_neighbors = std::min(P.nearest_neighbors_max, (int)(tr.max_nnz * 1.2));  // 1529
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  if (iterations) *iterations = k;
  if (n_trace) *n_trace = nt;
  // if (params.is_exporting_association && association_mat) gpu_association_to_cpu(A_host, ..., num_neighbors),
  // CvoGPU.cu:1552-1556 -> CvoGPU_impl.cu:366-427.  Literal, including its stride: `cols` is the value num_neighbors
  // has AFTER the loop.  When the loop ended through a `break` that is the stride the matrix was written with; when it
  // ran out of iterations, num_neighbors was already advanced for an iteration that never ran (CvoGPU.cu:1529), and the
  // row-major buffers are read with a stride they were not written with (whatever that yields is what upstream
  // returns; entries beyond rows * K_used are leftovers of earlier iterations).  source_inliers / target_inliers /
  // the pairs of a row are reported in row order (upstream fills them from an OpenMP loop, order unspecified).
  // (after ANY loop whose body ran once: a `break` in iteration 0 - flow vanished, or dist < eps_2 with a
  // warm start at the optimum and min_step < eps_2 - leaves k == 0 and still exports iteration 0's matrix)
  if (assoc && max_iter > 0) {
    assoc->K_used = K_used;
    assoc->K_final = num_neighbors;
    const int rows = X.n, cols = num_neighbors;
    if (last_nnz != 0) {
      for (int i = 0; i < rows; i++) {
        if (ws.nonzeros[i] == 0) continue;
        if (assoc->source_inliers && assoc->n_source_inliers < assoc->cap_rows) assoc->source_inliers[assoc->n_source_inliers] = i;
        assoc->n_source_inliers++;
        for (int j = 0; j < cols; j++) {
          const size_t f = (size_t)i * cols + j;
          if (ws.ind[f] == -1) break;
          if (assoc->n_pairs < assoc->cap_pairs) {
            assoc->row[assoc->n_pairs] = i;
            assoc->col[assoc->n_pairs] = ws.ind[f];
            assoc->val[assoc->n_pairs] = ws.mat[f];
          } else {
            assoc->overflow = 1;
          }
          assoc->n_pairs++;
        }
      }
    }
  }
  mat4_from_RT_inverse(R, T, out);
  return ret;
}

int oracle_align(const OracleParams* p, const OracleCloud* x, const OracleCloud* y, const float init[16],
                 float out[16], int* iterations, OracleTrace* trace, int max_trace, int trace_dense,
                 int trace_every, int* n_trace, double* seconds, int max_iter_override) {
  return align_core(p, x, y, init, out, iterations, trace, max_trace, trace_dense, trace_every, n_trace, seconds,
                    max_iter_override, nullptr);
}

int oracle_align_association(const OracleParams* p, const OracleCloud* x, const OracleCloud* y, const float init[16],
                             float out[16], int* iterations, int max_iter_override, OracleAssociation* assoc) {
  return align_core(p, x, y, init, out, iterations, nullptr, 0, 0, 0, nullptr, nullptr, max_iter_override, assoc);
}

// inner_product_impl + A_sum, CvoGPU.cu:1719-1778, SparseKernelMat.cu:62-68.  The float
// thrust::reduce over all rows*cols slots has unspecified order; restated as the float
// cast of the double sum of the stored entries.
static double inner_product_sum(const OracleParams& P, const CloudView& X, const CloudView& Y0,
                                const float Tm[16], float ell, std::vector<float>* mat_out,
                                std::vector<int>* ind_out, std::vector<unsigned>* nz_out) {
  float R[9], T[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Tm[4 * j + i];
    T[i] = Tm[12 + i];
  }
  float Ri[9], Ti[3];
  update_tf_impl(R, T, Ri, Ti);
  std::vector<float> yt(3 * (size_t)Y0.n);
  for (int j = 0; j < Y0.n; j++) transform_point(Ri, Ti, Y0.p(j), &yt[3 * (size_t)j]);
  CloudView Y = Y0;
  Y.xyz = yt.data();
  int K = P.nearest_neighbors_max;
  std::vector<float> mat((size_t)X.n * K, 0.0f);
  std::vector<int> ind((size_t)X.n * K, -1);
  std::vector<unsigned> nz(X.n);
  se_kernel_impl(P, X, Y, K, ell, mat.data(), ind.data(), nz.data(), false);
  double s = 0;
  for (int i = 0; i < X.n; i++)
    for (unsigned t = 0; t < nz[i]; t++) s += (double)mat[(size_t)i * K + t];
  if (mat_out) mat_out->swap(mat);
  if (ind_out) ind_out->swap(ind);
  if (nz_out) nz_out->swap(nz);
  return s;
}

float oracle_inner_product(const OracleParams* p, const OracleCloud* x, const OracleCloud* y, const float Tm[16],
                           float ell) {
  return (float)inner_product_sum(*p, view(x), view(y), Tm, ell, nullptr, nullptr, nullptr);
}

// function_angle, CvoGPU.cu:1814-1846 (is_gpu = true branch)
float oracle_function_angle(const OracleParams* p, const OracleCloud* x, const OracleCloud* y, const float Tm[16],
                            float ell, int is_approximate) {
  if (x->n == 0 || y->n == 0) return 0;
  const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float fxfz = oracle_inner_product(p, x, y, Tm, ell);
  float fx_norm, fz_norm;
  if (is_approximate) {
    fx_norm = (float)std::sqrt((double)x->n);
    fz_norm = (float)std::sqrt((double)y->n);
  } else {
    fx_norm = std::sqrt(oracle_inner_product(p, x, x, identity, ell));
    fz_norm = std::sqrt(oracle_inner_product(p, y, y, identity, ell));
  }
  return fxfz / (fx_norm * fz_norm);
}

// compute_association_gpu(float lengthscale) -> gpu_association_to_cpu, CvoGPU.cu:1876-1911,
// CvoGPU_impl.cu:366-427, as CSR (row-major sparse, columns in stored = ascending order).
int oracle_association(const OracleParams* p, const OracleCloud* x, const OracleCloud* y, const float Tm[16],
                       float ell, int* row_ptr, int* col, float* val) {
  std::vector<float> mat;
  std::vector<int> ind;
  std::vector<unsigned> nz;
  inner_product_sum(*p, view(x), view(y), Tm, ell, &mat, &ind, &nz);
  int K = p->nearest_neighbors_max, cnt = 0;
  for (int i = 0; i < x->n; i++) {
    row_ptr[i] = cnt;
    for (unsigned t = 0; t < nz[i]; t++) {
      col[cnt] = ind[(size_t)i * K + t];
      val[cnt] = mat[(size_t)i * K + t];
      cnt++;
    }
  }
  row_ptr[x->n] = cnt;
  return cnt;
}

// ---- non-isotropic kernel (CvoGPU.cu:152-171, 217-327, 1913-1995) ---------------------------------------------
// Eigen 3.3.9 Matrix3f::inverse() (Inverse.h, compute_inverse<..., 3>): cofactors, det = c00*m00 + (c10*m10 +
// c20*m20), result = cofactor^T * (1/det); host code, plain float arithmetic.  m and out are ROW-major here.
static void inverse3_eigen(const float m[9], float out[9]) {
  auto M = [&](int i, int j) { return m[3 * i + j]; };
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
  };
  const float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
  const float p0 = c0 * M(0, 0), p1 = c1 * M(1, 0), p2 = c2 * M(2, 0);
  const float det = SUM3(p0, p1, p2);
  const float invdet = 1.0f / det;
  out[0] = c0 * invdet;  // result.row(0) = cofactors_col0 * invdet
  out[1] = c1 * invdet;
  out[2] = c2 * invdet;
  out[3] = cof(0, 1) * invdet;
  out[4] = cof(1, 1) * invdet;
  out[5] = cof(2, 1) * invdet;
  out[6] = cof(0, 2) * invdet;
  out[7] = cof(1, 2) * invdet;
  out[8] = cof(2, 2) * invdet;
}

int oracle_association_non_isotropic(const OracleParams* p, const OracleCloud* x, const OracleCloud* y0,
                                     const float Tm[16], const float kernel_cm[9], int* row_ptr, int* col, float* val,
                                     float* kinv_out) {
  OracleParams P = *p;
  P.is_using_geometric_type = 0;  // CvoGPU.cu:1950-1951
  const CloudView X = view(x), Y0 = view(y0);
  float R[9], T[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[3 * i + j] = Tm[4 * j + i];
    T[i] = Tm[12 + i];
  }
  float Ri[9], Ti[3];
  update_tf_impl(R, T, Ri, Ti);
  std::vector<float> yt(3 * (size_t)Y0.n);
  for (int j = 0; j < Y0.n; j++) transform_point(Ri, Ti, Y0.p(j), &yt[3 * (size_t)j]);
  CloudView Y = Y0;
  Y.xyz = yt.data();
  float km[9], kinv[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) km[3 * i + j] = kernel_cm[3 * j + i];
  inverse3_eigen(km, kinv);
  if (kinv_out)
    for (int q = 0; q < 9; q++) kinv_out[q] = kinv[q];
  // prologue of fill_in_A_mat_gpu_dense_mat_kernel (CvoGPU.cu:233-252): squares kept in float, no geometric cut-off
  const float sigma_square = P.sigma * P.sigma, c_ell_square = P.c_ell * P.c_ell, s_ell_square = P.s_ell * P.s_ell;
  const float c_sigma_square = P.c_sigma * P.c_sigma, s_sigma_square = P.s_sigma * P.s_sigma;
  float d2_c_thres = 1, d2_s_thres = 1;
  if (P.is_using_intensity) d2_c_thres = (float)(-2.0 * c_ell_square * (double)std::log(P.sp_thres / c_sigma_square));
  if (P.is_using_semantics) d2_s_thres = (float)(-2.0 * s_ell_square * (double)std::log(P.sp_thres / s_sigma_square));
  const int K = P.nearest_neighbors_max;
  int cnt = 0;
  // fill_in_A_mat_gpu_dense_mat_kernel, CvoGPU.cu:217-327 (literal; rows sequential so the CSR is built in place)
  for (int i = 0; i < X.n; i++) {
    row_ptr[i] = cnt;
    unsigned num_inds = 0;
    for (int j = 0; j < Y.n; j++) {
      if (num_inds == (unsigned)K) break;
      float a = 1, sk = 1, ck = 1, k = 1, geo_sim = 1;
      if (P.is_using_geometry) {
        // mahananobis_distance (CvoGPU.cu:152-171): dist = a - b; (dist^T * kernel_inv) * dist, device code
        const float d0 = X.p(i)[0] - Y.p(j)[0], d1 = X.p(i)[1] - Y.p(j)[1], d2v = X.p(i)[2] - Y.p(j)[2];
        const float r0 = dot3_dev(d0, d1, d2v, kinv[0], kinv[3], kinv[6]);
        const float r1 = dot3_dev(d0, d1, d2v, kinv[1], kinv[4], kinv[7]);
        const float r2 = dot3_dev(d0, d1, d2v, kinv[2], kinv[5], kinv[8]);
        const float d2 = dot3_dev(r0, r1, r2, d0, d1, d2v);
        k = (float)((double)sigma_square * std::exp((double)(-d2) / 2.0));
      }
      if (P.is_using_intensity) {
        float d2_color = squared_dist_n(X.f(i), Y.f(j), FD);
        if (d2_color < d2_c_thres)
          ck = (float)((double)c_sigma_square * std::exp((double)(-d2_color) / (2.0 * c_ell_square)));
        else
          continue;
      }
      if (P.is_using_semantics) {
        float d2_semantic = squared_dist_n(X.l(i), Y.l(j), NC);
        if (d2_semantic < d2_s_thres)
          sk = (float)((double)s_sigma_square * std::exp((double)(-d2_semantic) / (2.0 * s_ell_square)));
        else
          continue;
      }
      a = ck * k * sk * geo_sim;
      if (a > P.sp_thres) {
        col[cnt] = j;
        val[cnt] = a;
        cnt++;
        num_inds++;
      }
    }
  }
  row_ptr[X.n] = cnt;
  return cnt;
}

void oracle_set_grid(int on) { g_use_grid = on != 0; }
double oracle_scan_seconds(int reset) {
  const double v = g_scan_seconds;
  if (reset) g_scan_seconds = 0;
  return v;
}
int oracle_get_grid(void) { return g_use_grid; }

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

}  // extern "C"
