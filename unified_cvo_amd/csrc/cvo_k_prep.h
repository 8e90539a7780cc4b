// cvo_k_prep.h -- k_prep: update_tf + transform + per-row cut-offs as cull operands and boxes (rebuild only).
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_k_scan.h"
#include "cvo_pair_math.h"

namespace cvo_dev {

// ------------------------------------------------------------------------------------------
// k_prep: everything the next iteration's kernels read.  Blocks [0, Mpad/512) handle the targets:
// transform_pointcloud_thrust (CvoGPU_impl.cu:164-173) from the INITIAL cloud (exact, original index),
// the cull form in sorted order and chunk / slice bounding boxes (one wave = one 64-target chunk).
// The remaining blocks handle the rows: per-row constants of fill_in_A_mat_gpu (CvoGPU.cu:504-510), the
// conservative cull operand (sorted order) and the bounding box of every group of ROWS_PER_GROUP rows,
// grown by the group's largest cut-off radius ("boxes disjoint" => no pair of the tile is a hit).
//
// Cull arithmetic: pair (i, j) is a candidate iff
//     |y~|^2 (1 - 4e-6) - 2 x~.y~  <  thr_i + 4e-6 |x~|^2 + 1e-5 thr_i - |x~|^2
// i.e. the exact test d2 < thr_i with a slack of 4e-6 (|x~|^2 + |y~|^2) + 1e-5 thr_i, > 5x the
// worst-case rounding of the expanded form plus the centring error.
// ------------------------------------------------------------------------------------------
constexpr int PREP_THREADS = 512;

__global__ __launch_bounds__(PREP_THREADS) void k_prep(const PairDesc* __restrict__ descs,
                                                        const DevParams* __restrict__ Pp,
                                                        const PairState* __restrict__ states) {
  const PairState* st = states + blockIdx.y;  // == D->st (see k_scan)
  {
    const int status_v = st->status, rebuild_v = st->rebuild;
    if (status_v != 0 || !rebuild_v) return;  // finished / the bitmap of an earlier iteration still covers this one
  }
  const PairDesc* __restrict__ D = descs + blockIdx.y;
  if (st->all_dense) {  // dense regime: no operands to prepare, only the overflow list to reset for k_list
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      D->st->n_ovf = 0;
      D->st->n_scan = 0;
      D->st->ncand_list = 0ull;
    }
    return;
  }
  const DevParams P = *Pp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float INF = __builtin_inff();
  const float cx = D->cx, cy = D->cy, cz = D->cz;
  const int ntb = D->Mpad / PREP_THREADS;
  if ((int)blockIdx.x < ntb) {
    __shared__ float s_box[PREP_THREADS / 64][6];
    float Ri[9], Ti[3];
#pragma unroll
    for (int q = 0; q < 9; q++) Ri[q] = st->Rinv[q];
#pragma unroll
    for (int q = 0; q < 3; q++) Ti[q] = st->Tinv[q];
    const int M = D->M;
    const int sidx = blockIdx.x * PREP_THREADS + tid;
    float ux = 0, uy = 0, uz = 0, nn = INF;
    float lox = INF, loy = INF, loz = INF, hix = -INF, hiy = -INF, hiz = -INF;
    if (sidx < M) {
      const float4 p = D->ys4[sidx];  // spatially ordered copy of the initial target cloud: pure streaming
      const V3 q = transform_point(Ri, Ti, p.x, p.y, p.z);
      ux = q.x - cx;
      uy = q.y - cy;
      uz = q.z - cz;
      nn = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
      nn = __builtin_fmaf(-4e-6f, nn, nn);
      lox = hix = ux;
      loy = hiy = uy;
      loz = hiz = uz;
    }
    D->ycull[sidx] = make_float4(ux, uy, uz, nn);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lox = fminf(lox, __shfl_xor(lox, o));
      loy = fminf(loy, __shfl_xor(loy, o));
      loz = fminf(loz, __shfl_xor(loz, o));
      hix = fmaxf(hix, __shfl_xor(hix, o));
      hiy = fmaxf(hiy, __shfl_xor(hiy, o));
      hiz = fmaxf(hiz, __shfl_xor(hiz, o));
    }
    if (lane == 0) {
      s_box[wave][0] = lox;
      s_box[wave][1] = loy;
      s_box[wave][2] = loz;
      s_box[wave][3] = hix;
      s_box[wave][4] = hiy;
      s_box[wave][5] = hiz;
    }
    __syncthreads();
    const int T = P.T;  // 1, 2, 4 or 8: slices never straddle a 512-target block
    if (tid < (PREP_THREADS / 64) / T) {
      float4 lo = make_float4(INF, INF, INF, 0.f), hi = make_float4(-INF, -INF, -INF, 0.f);
      for (int t = 0; t < T; t++) {
        const float* b = s_box[tid * T + t];
        lo.x = fminf(lo.x, b[0]);
        lo.y = fminf(lo.y, b[1]);
        lo.z = fminf(lo.z, b[2]);
        hi.x = fmaxf(hi.x, b[3]);
        hi.y = fmaxf(hi.y, b[4]);
        hi.z = fmaxf(hi.z, b[5]);
      }
      const int sl = blockIdx.x * ((PREP_THREADS / 64) / T) + tid;
      D->sbox[2 * (size_t)sl] = lo;
      D->sbox[2 * (size_t)sl + 1] = hi;
    }
    return;
  }
  // ---- rows
  const int N = D->N;
  const int rs = (blockIdx.x - ntb) * PREP_THREADS + tid;
  if (rs >= D->NGpad * ROWS_PER_GROUP) return;  // whole waves drop out together (NGpad*4 is a multiple of 256)
  const float ell = st->ell;  // == st->ell_build: a rebuild always uses the current lengthscale
  if (rs == 0) {  // k_list refills the overflow list of k_assoc_dense and the lists' candidate count
    D->st->n_ovf = 0;
    D->st->n_scan = 0;
    D->st->ncand_list = 0ull;
  }
  // per-row skin = skin_rot * rho_i + skin_tr (+ rounding slack), see PairState / update_body
  const float skin_rot = st->skin_rot, skin_tr = st->skin_tr;
  const float tb_norm = sqrtf(__builtin_fmaf(st->Tinv[2], st->Tinv[2], __builtin_fmaf(st->Tinv[1], st->Tinv[1], st->Tinv[0] * st->Tinv[0])));
  float ux = 0, uy = 0, uz = 0, cw = -INF, rad = 0;
  float lox = INF, loy = INF, loz = INF, hix = -INF, hiy = -INF, hiz = -INF;
  if (rs < N) {
    const float4 x = D->xs4[rs];
    const RowData r = make_row(P, x, ell);
    // cut-off of the scan: (sqrt(thr) + skin)^2, rounded up, so that the bitmap stays a superset of the
    // exact test while the targets move by less than `skin` (and ell does not grow)
    // rho_i bounds |y0| of every target that can enter the row's ball while the lists live: such a target sits at
    // y_t = Rinv y0 + Tinv with |y_t - x_i| < r_i and |Tinv - Tb| <= skin_tr, Rinv a (float) rotation
    const float r_i = sqrtf(fmaxf(r.d2_thres, 0.f));
    const float a_to_sensor = sqrtf(__builtin_fmaf(x.z, x.z, __builtin_fmaf(x.y, x.y, x.x * x.x)));
    const float rho = 1.001f * (a_to_sensor + r_i + tb_norm + skin_tr);
    const float skin = __builtin_fmaf(skin_rot, rho, skin_tr) + 2e-5f * (rho + 1.f);
    const float rs_ = __builtin_fmaf(r_i, 1.000001f, (skin_rot > 0.f || skin_tr > 0.f) ? skin : 0.f);
    const float thr = rs_ * rs_ * 1.000001f;
    ux = x.x - cx;
    uy = x.y - cy;
    uz = x.z - cz;
    const float nx = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
    const float margin = 4e-6f * nx + 1e-5f * fabsf(thr);
    cw = (thr + margin) - nx;
    rad = sqrtf(fmaxf(thr + margin, 0.f)) * 1.00001f + 1e-30f;
    if (!P.use_geo || !(thr == thr)) {  // no geometric cut-off (or NaN): every pair is a candidate
      cw = INF;
      rad = INF;
    }
    lox = hix = ux;
    loy = hiy = uy;
    loz = hiz = uz;
    // the bitmap is rebuilt from scratch: drop this row's slice bits and candidate count (k_scan runs after this kernel)
    D->row_cnt[rs] = 0;
    unsigned* rb = D->rowbits + (size_t)rs * D->rbw;
    for (int w0 = 0; w0 < D->rbw; w0 += 4) *reinterpret_cast<uint4*>(rb + w0) = make_uint4(0, 0, 0, 0);
  }
  if (rs < N + XCULL_PAD) D->xcull[rs] = make_float4(-2.f * ux, -2.f * uy, -2.f * uz, cw);
#pragma unroll
  for (int o = 1; o < ROWS_PER_GROUP; o <<= 1) {
    lox = fminf(lox, __shfl_xor(lox, o));
    loy = fminf(loy, __shfl_xor(loy, o));
    loz = fminf(loz, __shfl_xor(loz, o));
    hix = fmaxf(hix, __shfl_xor(hix, o));
    hiy = fmaxf(hiy, __shfl_xor(hiy, o));
    hiz = fmaxf(hiz, __shfl_xor(hiz, o));
    rad = fmaxf(rad, __shfl_xor(rad, o));
  }
  lox -= rad;
  loy -= rad;
  loz -= rad;
  hix += rad;
  hiy += rad;
  hiz += rad;
  if ((rs & (ROWS_PER_GROUP - 1)) == 0) {
    const int g = rs / ROWS_PER_GROUP;
    D->gbox[2 * (size_t)g] = make_float4(lox, loy, loz, 0.f);
    D->gbox[2 * (size_t)g + 1] = make_float4(hix, hiy, hiz, 0.f);
  }
  // level-1 boxes of k_scan: one per wave = cell of 64 sorted rows (16 groups)
#pragma unroll
  for (int o = ROWS_PER_GROUP; o < 64; o <<= 1) {
    lox = fminf(lox, __shfl_xor(lox, o));
    loy = fminf(loy, __shfl_xor(loy, o));
    loz = fminf(loz, __shfl_xor(loz, o));
    hix = fmaxf(hix, __shfl_xor(hix, o));
    hiy = fmaxf(hiy, __shfl_xor(hiy, o));
    hiz = fmaxf(hiz, __shfl_xor(hiz, o));
  }
  if (lane == 0) {
    const int c = rs >> 6;
    D->cellbox[2 * (size_t)c] = make_float4(lox, loy, loz, 0.f);
    D->cellbox[2 * (size_t)c + 1] = make_float4(hix, hiy, hiz, 0.f);
  }
}

}  // namespace cvo_dev
