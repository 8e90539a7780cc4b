// cvo_pair_math.h -- the exact per-pair arithmetic of fill_in_A_mat_gpu (CvoGPU.cu:477-593) and the per-row accumulators of the association kernels.
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_wave.h"

namespace cvo_dev {

// ------------------------------------------------------------------------------------------
// Exact per-pair arithmetic of fill_in_A_mat_gpu (CvoGPU.cu:528-573).
// ------------------------------------------------------------------------------------------
struct RowData {
  float x, y, z, l, d2_thres;
  int lid = 0;  // FEAT_HOT: the row's class (PairDesc::xlid), set by the caller that knows the row's feature index
  // denominator of the geometric kernel's exponent, 2.0 * l * l (CvoGPU.cu:552), and its refined reciprocal: the part of
  // the per-pair IEEE division that depends on the row only (rcp_refined / div_by, cvo_device.h)
  double den, rcp;
};
// per-row constants of fill_in_A_mat_gpu (CvoGPU.cu:504-510)
__device__ __forceinline__ RowData make_row(const DevParams& P, const float4 x, float ell) {
  const float a_to_sensor = sqrtf(__builtin_fmaf(x.z, x.z, __builtin_fmaf(x.y, x.y, x.x * x.x)));
  const float l = compute_range_ell(ell, a_to_sensor);
  float thr = 1.f;
  if (P.use_geo) thr = (float)(-2.0 * l * l * (double)P.log_geo);
  if (P.mode == 2) thr = P.d2_cull;  // non-isotropic kernel: no cut-off of its own, this one only steers the scan
  const double den = 2.0 * l * l;
  return RowData{x.x, x.y, x.z, l, thr, 0, den, rcp_refined(den)};
}
// 1 / (2.0 * ell * ell) narrowed to float (compute_step_size_poly_coeff, CvoGPU.cu:1060), the division in its hoisted form
__device__ __forceinline__ float coef_of_ell(float ell) {
  const double cden = 2.0 * ell * ell;
  return (float)div_by(1.0, cden, rcp_refined(cden));
}
// The colour and semantic kernels' exponent denominators (2.0 * c_ell^2, 2.0 * s_ell^2: the same for every pair of a
// call) with their refined reciprocals; evaluated once per thread, outside the row loops.
struct FeatDen {
  double c_den, c_rcp, s_den, s_rcp;
  float sk_same, sk_diff;  // FEAT_HOT: the semantic kernel for a squared class distance of 0 / of 2
  bool same_ok, diff_ok;   // ... and whether that distance passes the cut-off d2_s_thres at all
  ExpConsts ek;  // (rides along: every evaluation of a pair needs it)
};
__device__ __forceinline__ FeatDen make_feat_den(const DevParams& P) {
  FeatDen f;
  f.c_den = 2.0 * P.c2;
  f.c_rcp = rcp_refined(f.c_den);
  f.s_den = P.mode == 2 ? 2.0 * P.s_ell_sq : 2.0 * P.s_ell * P.s_ell;
  f.s_rcp = rcp_refined(f.s_den);
  f.ek = make_exp_consts();
  {  // (the general branch of eval_pair_yt, word for word, on the only two distances one-hot rows can have)
    const float ss = P.s_sigma * P.s_sigma;
    const float r0 = 0.f, r2 = 2.f;
    f.same_ok = r0 < P.d2_s_thres;
    f.diff_ok = r2 < P.d2_s_thres;
    f.sk_same = (float)((double)ss * exp_ocml<true>(div_by((double)(-r0), f.s_den, f.s_rcp), f.ek));
    f.sk_diff = (float)((double)ss * exp_ocml<true>(div_by((double)(-r2), f.s_den, f.s_rcp), f.ek));
  }
  return f;
}
struct Pose {  // the transform applied to the target cloud this iteration (update_tf, CvoGPU.cu:94-112)
  float Ri[9], Ti[3];
};
// What the row loops of one iteration need from the pair's state, by value: the two-kernel path fills it with scalar
// loads of the state an EARLIER launch wrote, the resident kernel with L1-bypassing loads of the state another block of
// the SAME launch wrote (a cached or compiler-hoisted copy would be stale there).
struct IterView {
  int K;
  float ell;
  int row_max;  // PairState::row_max
  Pose pose;
};
__device__ __forceinline__ Pose load_pose(const PairState* st) {
  Pose p;
#pragma unroll
  for (int q = 0; q < 9; q++) p.Ri[q] = st->Rinv[q];
#pragma unroll
  for (int q = 0; q < 3; q++) p.Ti[q] = st->Tinv[q];
  return p;
}
__device__ __forceinline__ IterView load_iter_view(const PairState* st) {
  IterView v;
  v.K = st->K;
  v.ell = st->ell;
  v.row_max = st->row_max;
  v.pose = load_pose(st);
  return v;
}

// FEAT selects what the instantiation carries (the host picks per call, launch_assoc):
//   FEAT_GEO  geometry only - no colour / semantic / geometric-type code at all: 1/3 fewer VGPRs, more waves per SIMD for the
//             latency-bound association kernel;
//   FEAT_COL  + colour and geometric types, no semantic code (config 3);
//   FEAT_HOT  + ONE-HOT semantics: every cloud of the call has exact one-hot class rows (checked at upload), so the 19-term
//             squared distance of CvoGPU.cu:563-569 is 0 (same class) or exactly 2 (different) and the semantic kernel one of
//             two constants - a 4-byte class id per candidate instead of two 80-byte rows (config 4);
//   FEAT_ALL  everything: soft class distributions, the non-isotropic kernel of mode 2.
constexpr int FEAT_GEO = 0, FEAT_ALL = 1, FEAT_COL = 2, FEAT_HOT = 3;
// i / j index the FEATURE arrays (colour, class distributions, geometric types), which clouds keep in spatial order:
// i = the row's sorted position, j = the target's sorted position.
// The pair arithmetic for an already transformed target yt (everything of CvoGPU.cu:528-573 but the transform).
template <int FEAT>
__device__ __forceinline__ bool eval_pair_yt(const DevParams& P, const PairDesc* __restrict__ D, const FeatDen& F, int i,
                                             const RowData& r, int j, const float4 yt, float& a_out) {
  constexpr bool GENERAL = FEAT != FEAT_GEO, SEM_ROWS = FEAT == FEAT_ALL, SEM_HOT = FEAT == FEAT_HOT, MODE2 = FEAT == FEAT_ALL;
  float sk = 1, ck = 1, k = 1, geo_sim = 1;
  if (GENERAL && P.use_geotype) {  // compute_geometric_type_ip, CvoGPU.cu:203-215
    const float2 ga = D->xgeo[i], gb = D->ygeo[j];
    const float n2a = __builtin_fmaf(ga.y, ga.y, ga.x * ga.x);
    const float n2b = __builtin_fmaf(gb.y, gb.y, gb.x * gb.x);
    const float dab = __builtin_fmaf(ga.y, gb.y, ga.x * gb.x);
    geo_sim = dab * dab / (n2a * n2b);
    if ((double)geo_sim < 0.01) return false;
  }
  if (MODE2 && P.use_geo && P.mode == 2) {  // (the host launches the FEAT_ALL instantiations for mode 2)
    // mahananobis_distance (CvoGPU.cu:152-171): dist = a - b, (dist^T * kernel_inv) * dist; no cut-off (236-238, 279-284)
    const float d0 = r.x - yt.x, d1 = r.y - yt.y, d2v = r.z - yt.z;
    const float r0 = dot3_dev(d0, d1, d2v, P.kinv[0], P.kinv[3], P.kinv[6]);
    const float r1 = dot3_dev(d0, d1, d2v, P.kinv[1], P.kinv[4], P.kinv[7]);
    const float r2 = dot3_dev(d0, d1, d2v, P.kinv[2], P.kinv[5], P.kinv[8]);
    const float d2 = dot3_dev(r0, r1, r2, d0, d1, d2v);
    k = (float)((double)P.sigma2 * exp_ocml<false>((double)(-d2) / 2.0, F.ek));  // (an indefinite kernel can make -d2 positive)
  } else if (P.use_geo) {
    const float dx = yt.x - r.x, dy = yt.y - r.y, dz = yt.z - r.z;
    const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    if (d2 < r.d2_thres)  // exp(-d2 / (2.0 * l * l)), CvoGPU.cu:552; d2 >= 0, so the exponent is <= 0
      k = (float)((double)P.sigma2 * exp_ocml<true>(div_by((double)(-d2), r.den, r.rcp), F.ek));
    else
      return false;
  }
  if (GENERAL && P.use_col) {
    const float4 a0 = D->xfeat[2 * i], a1 = D->xfeat[2 * i + 1];
    const float4 b0 = D->yfeat[2 * j], b1 = D->yfeat[2 * j + 1];
    float res = 0, tmp;
    tmp = a0.x - b0.x; res = __builtin_fmaf(tmp, tmp, res);
    tmp = a0.y - b0.y; res = __builtin_fmaf(tmp, tmp, res);
    tmp = a0.z - b0.z; res = __builtin_fmaf(tmp, tmp, res);
    tmp = a0.w - b0.w; res = __builtin_fmaf(tmp, tmp, res);
    tmp = a1.x - b1.x; res = __builtin_fmaf(tmp, tmp, res);
    if (res < P.d2_c_thres)  // (res is a sum of squares: exponent <= 0)
      ck = (float)((double)P.c_sigma2 * exp_ocml<true>(div_by((double)(-res), F.c_den, F.c_rcp), F.ek));
    else
      return false;
  }
  if (SEM_HOT && P.use_sem) {
    // one-hot rows: d2 is 0 or 2 exactly (FeatDen::sk_same / sk_diff are the general branch's own arithmetic on those two)
    const bool same = D->ylid[j] == r.lid;
    if (same ? F.same_ok : F.diff_ok)
      sk = same ? F.sk_same : F.sk_diff;
    else
      return false;
  }
  if (SEM_ROWS && P.use_sem) {
    float res = 0;
#pragma unroll
    for (int q = 0; q < NC_PAD / 4; q++) {
      const float4 a = D->xlabel[5 * i + q], b = D->ylabel[5 * j + q];
      float tmp;
      tmp = a.x - b.x; res = __builtin_fmaf(tmp, tmp, res);
      tmp = a.y - b.y; res = __builtin_fmaf(tmp, tmp, res);
      tmp = a.z - b.z; res = __builtin_fmaf(tmp, tmp, res);
      if (q < 4) {  // the 20th float is padding (0 - 0 adds exactly 0, skipped anyway)
        tmp = a.w - b.w; res = __builtin_fmaf(tmp, tmp, res);
      }
    }
    if (res < P.d2_s_thres)  // (F.s_den: 2.0 * s_ell^2 kept in float for mode 2, 2.0 * s_ell * s_ell otherwise)
      sk = (float)((double)(P.s_sigma * P.s_sigma) * exp_ocml<true>(div_by((double)(-res), F.s_den, F.s_rcp), F.ek));
    else
      return false;
  }
  a_out = ck * k * sk * geo_sim;
  return true;
}
// transform_point_R_T (CvoGPU_impl.cu:31-82) of the INITIAL target y0 = y4[j], recomputed where it is needed, then the
// pair arithmetic.  (The gates of a pair - geometric type, distance, colour, semantics - only ever reject: the order in
// which they are tested does not reach a result.)
template <int FEAT>
__device__ __forceinline__ bool eval_pair(const DevParams& P, const PairDesc* __restrict__ D, const FeatDen& F, const Pose& pose,
                                          int i, const RowData& r, int j, const float4 y0, float& a_out, float4& yt_out) {
  const V3 ytv = transform_point(pose.Ri, pose.Ti, y0.x, y0.y, y0.z);
  const float4 yt = make_float4(ytv.x, ytv.y, ytv.z, 0.f);
  yt_out = yt;
  return eval_pair_yt<FEAT>(P, D, F, i, r, j, yt, a_out);
}
// ------------------------------------------------------------------------------------------
// k_assoc: ordered association + flow, one thread per (sorted) source row.
// ------------------------------------------------------------------------------------------
// Rows per block of the two per-iteration kernels.  256 (four waves) against 128: half as many arrivals queue up on a
// pair's last-block counters and half as many partials are re-read by the serial tails (-1.9 % of the step); 512
// loses 10 % (the block reduction and its registers grow, a block waits for the slowest of eight waves).
#ifndef CVO_ASSOC_THREADS
#define CVO_ASSOC_THREADS 256
#endif
constexpr int ASSOC_THREADS = CVO_ASSOC_THREADS;
// candidates per row the sorted per-thread LDS list holds: 64 with 16-bit indices (M < 65536, 16.6 KB
// per block so ~9 blocks share a CU), 32 with 32-bit indices
constexpr int ASSOC_CAP16 = 64;
constexpr int ASSOC_CAP32 = 32;

struct RowAcc {
  float o0 = 0, o1 = 0, o2 = 0, v0 = 0, v1 = 0, v2 = 0;
  double asum = 0;
  unsigned nnz = 0;
  EllEntry* slot = nullptr;  // where the row's next nonzero goes: D->ell + nnz * N + pos, advanced by N per nonzero
  // this thread's column of the block's LDS staging area (AssocShared::stage), as an LDS-qualified pointer (a generic one
  // made hipcc 7.2 emit an illegal V_CMP against src_shared_base in one instantiation of k_assoc)
  CVO_LDS ell_vec_t* stage = nullptr;
};
// ELL entries a row parks in LDS before they are stored (see assoc_phase).  Six: 24.6 KB of LDS per block; 4 / 5 / 6 / 7 / 8
// slots measured 63.6 / 63.3 / 62.9 / 63.8 / 64.9 ms per step (the early iterations have ~8 nonzeros per row, the
// steady state 2-3; beyond 6 the LDS footprint costs more occupancy than the longer rows gain).
#ifndef CVO_ELL_STAGE_SLOTS
#define CVO_ELL_STAGE_SLOTS 6
#endif
constexpr int ELL_STAGE = CVO_ELL_STAGE_SLOTS;

// One pair (i, j) that passed the geometric cut-off, with its transformed target: the rest of CvoGPU.cu:528-589 (kernel
// values, a > sp_thres, ELL store) + the flow terms of 758-782.
template <int FEAT>
__device__ __forceinline__ void visit_pair_yt(const DevParams& P, const PairDesc* __restrict__ D, const FeatDen& F, int i, int pos,
                                              int N, const RowData& r, const V3& pxe, int j, const float4 yt, RowAcc& A) {
  float a;
  if (!eval_pair_yt<FEAT>(P, D, F, i, r, j, yt, a)) return;
  if (a > P.sp_thres) {
    // The row's first ELL_STAGE nonzeros are parked in the thread's own LDS column and leave after the loop as
    // write-through stores (assoc_phase); only rows longer than that store from inside the loop.
    if (A.nnz < (unsigned)ELL_STAGE)
      A.stage[A.nnz * ASSOC_THREADS] = ell_to_vec(make_ell(a, yt.x, yt.y, yt.z, j));
    else
      *A.slot = make_ell(a, yt.x, yt.y, yt.z, j);
    if (P.keep_columns) D->ell_j[(size_t)A.nnz * N + pos] = D->yorder[j];  // (list entries are sorted positions)
    A.slot += N;
    A.nnz++;
    const V3 pye{yt.x, yt.y, yt.z};
    const V3 cr = cross_dev(pxe, pye);
    const float dx = pye.x - pxe.x, dy = pye.y - pxe.y, dz = pye.z - pxe.z;
    A.o0 = __builtin_fmaf(cr.x, a, A.o0);
    A.o1 = __builtin_fmaf(cr.y, a, A.o1);
    A.o2 = __builtin_fmaf(cr.z, a, A.o2);
    A.v0 = __builtin_fmaf(dx, a, A.v0);
    A.v1 = __builtin_fmaf(dy, a, A.v1);
    A.v2 = __builtin_fmaf(dz, a, A.v2);
    if (P.mode != 0) A.asum += (double)a;  // A_sum (SparseKernelMat.cu:62-68): only the single evaluations read it
  }
}

}  // namespace cvo_dev
