// cvo_kernels.h -- the hand-written gfx950 kernels of the pairwise align() hot path.
//
// One optimiser iteration of every in-flight frame pair is four launches (blockIdx.z/y = pair):
//
//   k_scan   N x M candidate scan.  Replaces the O(N*M) part of fill_in_A_mat_gpu
//            (CvoGPU.cu:477-593).  Lanes hold targets (coalesced float4 loads of the SoA
//            `ycull`), source rows are wave-uniform (scalar loads of `xcull`): per 64 pairs it
//            issues 3 v_fma_f32 + 1 v_cmp_lt_f32 whose 64-bit lane mask IS the row-major
//            candidate bitmap word.  Non-empty words (a few %) are stored with a flag.
//   k_assoc  one thread per source row walks its flagged mask words in ascending j and runs the
//            reference's exact per-pair arithmetic (double exp, colour / semantic kernels,
//            a > sp_thres, first-K truncation), writes the ELL matrix and accumulates the
//            per-row flow (compute_flow_gpu_no_eigen, CvoGPU.cu:729-790).
//   k_coeff  reduces the flow partials to the normalised twist, then one thread per row
//            accumulates B,C,D,E (compute_step_size_xi + _poly_coeff, CvoGPU.cu:953-1082).
//   k_step   one block per pair: reduces B..E, runs the reference's host-side scalar code on
//            one thread (cubic, Exp, pose update, SE(3) log, indicator, ell decay, K update;
//            CvoGPU.cu:1122-1158, 1452-1531) and then prepares the next iteration
//            (update_tf + transform_pointcloud_thrust + per-row cut-offs + cull operands).
//
// No host round trip happens inside the loop; finished pairs early-exit on their status word.
#pragma once
#include "cvo_device.h"

namespace cvo_dev {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_down((int)v, o));
  return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o));
  return v;
}

// ------------------------------------------------------------------------------------------
// k_scan<T>: each wave owns T consecutive 64-target chunks (one "slice") and a range of rows.
// Test per pair (conservative, DESIGN.md "Cull arithmetic"):
//     |y~|^2 - 2 x~.y~  <  d2_thres_i + margin_i - |x~|^2
// evaluated as 3 FMAs with wave-uniform row operands.  The exact reference test is re-done in
// k_assoc for every flagged pair, so the scan only has to be a superset.
// ------------------------------------------------------------------------------------------
// Address-space qualified views: pointers read out of a PairDesc are generic ("flat") to the
// compiler; the scan's hot pointers are re-qualified so that the target tile uses global_load,
// and the wave-uniform row operands use s_load (constant address space => scalar cache; xcull is
// written by the previous kernel, k_step, so it is read-only for the lifetime of k_scan).
#define CVO_GLOBAL __attribute__((address_space(1)))
#define CVO_CONST __attribute__((address_space(4)))
typedef float f32x4 __attribute__((ext_vector_type(4)));  // plain vector: loadable from any address space

template <int T, int RU>
__global__ __launch_bounds__(256) void k_scan(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp,
                                              int force) {
  const PairDesc* __restrict__ D = descs + blockIdx.z;
  if (!force && D->st->status != 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int slice = blockIdx.x * 4 + wave;
  const int nslices = D->nslices;
  if (slice >= nslices) return;
  const int rpb = Pp->rows_per_block;  // a multiple of RU
  const int N = D->N;
  const int r0 = blockIdx.y * rpb;
  if (r0 >= N) return;
  const int r1 = min(r0 + rpb, N);

  const CVO_GLOBAL f32x4* yc = (const CVO_GLOBAL f32x4*)D->ycull;
  float y1[T], y2[T], y3[T], yy[T];
#pragma unroll
  for (int t = 0; t < T; t++) {
    const f32x4 q = yc[(size_t)(slice * T + t) * 64 + lane];
    y1[t] = q.x;
    y2[t] = q.y;
    y3[t] = q.z;
    yy[t] = q.w;
  }
  // xcull is padded with never-passing rows (w = -inf) up to N + XCULL_PAD, so neither the last
  // row group nor the prefetch of the group after it needs clamping.
  const CVO_CONST f32x4* xr = (const CVO_CONST f32x4*)D->xcull + r0;
  CVO_GLOBAL unsigned long long* masks = (CVO_GLOBAL unsigned long long*)D->masks;
  CVO_GLOBAL unsigned short* flags = (CVO_GLOBAL unsigned short*)D->flags;
  const int nchunks = D->nchunks;
  const int nsl_pad = D->nsl_pad;

  f32x4 cur[RU], nxt[RU];
#pragma unroll
  for (int u = 0; u < RU; u++) cur[u] = xr[u];
  for (int r = r0; r < r1; r += RU) {
    xr += RU;
#pragma unroll
    for (int u = 0; u < RU; u++) nxt[u] = xr[u];  // prefetch the next row group (scalar loads)
    float acc[RU][T];
    unsigned long long mu[RU];
    unsigned long long any = 0;
#pragma unroll
    for (int u = 0; u < RU; u++) {
#pragma unroll
      for (int t = 0; t < T; t++) {
        float a = __builtin_fmaf(y1[t], cur[u].x, yy[t]);
        a = __builtin_fmaf(y2[t], cur[u].y, a);
        acc[u][t] = __builtin_fmaf(y3[t], cur[u].z, a);
      }
      // one compare per row: min over the wave's T chunks (v_min3_f32) against the row threshold
      float mn = acc[u][0];
#pragma unroll
      for (int t = 1; t < T; t++) mn = __builtin_fminf(mn, acc[u][t]);
      mu[u] = __ballot(mn < cur[u].w);
      any |= mu[u];
    }
    if (any) {  // rare: some row of the group has a candidate among this wave's 64*T targets
#pragma unroll
      for (int u = 0; u < RU; u++) {
        if (mu[u]) {
          unsigned long long mine = 0;
          unsigned fl = 0;
#pragma unroll
          for (int t = 0; t < T; t++) {
            const unsigned long long m = __ballot(acc[u][t] < cur[u].w);
            if (lane == t) mine = m;
            fl |= (m != 0 ? 1u : 0u) << t;
          }
          if (lane < T && mine) masks[(size_t)(r + u) * nchunks + slice * T + lane] = mine;
          if (lane == 0) flags[(size_t)(r + u) * nsl_pad + slice] = (unsigned short)fl;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RU; u++) cur[u] = nxt[u];
  }
}

// ------------------------------------------------------------------------------------------
// Exact per-pair arithmetic of fill_in_A_mat_gpu (CvoGPU.cu:528-573).
// ------------------------------------------------------------------------------------------
struct RowData {
  float x, y, z, l, d2_thres;
};

__device__ __forceinline__ bool eval_pair(const DevParams& P, const PairDesc* __restrict__ D, int i,
                                          const RowData& r, int j, float& a_out, float4& yt_out) {
  float sk = 1, ck = 1, k = 1, geo_sim = 1;
  if (P.use_geotype) {  // compute_geometric_type_ip, CvoGPU.cu:203-215
    const float2 ga = D->xgeo[i], gb = D->ygeo[j];
    const float n2a = __builtin_fmaf(ga.y, ga.y, ga.x * ga.x);
    const float n2b = __builtin_fmaf(gb.y, gb.y, gb.x * gb.x);
    const float dab = __builtin_fmaf(ga.y, gb.y, ga.x * gb.x);
    geo_sim = dab * dab / (n2a * n2b);
    if ((double)geo_sim < 0.01) return false;
  }
  const float4 yt = D->yt4[j];
  yt_out = yt;
  if (P.use_geo) {
    const float dx = yt.x - r.x, dy = yt.y - r.y, dz = yt.z - r.z;
    const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    if (d2 < r.d2_thres)
      k = (float)((double)P.sigma2 * exp((double)(-d2) / (2.0 * r.l * r.l)));
    else
      return false;
  }
  if (P.use_col) {
    const float4 a0 = D->xfeat[2 * i], a1 = D->xfeat[2 * i + 1];
    const float4 b0 = D->yfeat[2 * j], b1 = D->yfeat[2 * j + 1];
    float res = 0, tmp;
    tmp = a0.x - b0.x; res = __builtin_fmaf(tmp, tmp, res);
    tmp = a0.y - b0.y; res = __builtin_fmaf(tmp, tmp, res);
    tmp = a0.z - b0.z; res = __builtin_fmaf(tmp, tmp, res);
    tmp = a0.w - b0.w; res = __builtin_fmaf(tmp, tmp, res);
    tmp = a1.x - b1.x; res = __builtin_fmaf(tmp, tmp, res);
    if (res < P.d2_c_thres)
      ck = (float)((double)P.c_sigma2 * exp((double)(-res) / (2.0 * P.c2)));
    else
      return false;
  }
  if (P.use_sem) {
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
    if (res < P.d2_s_thres)
      sk = (float)((double)(P.s_sigma * P.s_sigma) * exp((double)(-res) / (2.0 * P.s_ell * P.s_ell)));
    else
      return false;
  }
  a_out = ck * k * sk * geo_sim;
  return true;
}

// ------------------------------------------------------------------------------------------
// k_assoc: ordered association + flow, one thread per source row.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assoc(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp) {
  const PairDesc* __restrict__ D = descs + blockIdx.y;
  PairState* st = D->st;
  if (st->status != 0) return;
  const DevParams P = *Pp;
  const int N = D->N;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int K = st->K;
  const int T = P.T;
  float o0 = 0, o1 = 0, o2 = 0, v0 = 0, v1 = 0, v2 = 0;
  double asum = 0;
  unsigned nnz = 0;
  unsigned long long ncand = 0;
  if (i < N) {
    const float4 x = D->x4[i];
    const float2 rc = D->rowc[i];
    const RowData r{x.x, x.y, x.z, rc.x, rc.y};
    const V3 pxe{x.x, x.y, x.z};
    unsigned short* frow = D->flags + (size_t)i * D->nsl_pad;
    const unsigned long long* mrow = D->masks + (size_t)i * D->nchunks;
    const int nsl = D->nslices;
    for (int s0 = 0; s0 < nsl; s0 += 8) {
      uint4 w = *reinterpret_cast<const uint4*>(frow + s0);
      if ((w.x | w.y | w.z | w.w) == 0) continue;
      *reinterpret_cast<uint4*>(frow + s0) = make_uint4(0, 0, 0, 0);  // self-cleaning flags
      const unsigned ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int h = 0; h < 8; h++) {
        unsigned f = (ww[h >> 1] >> ((h & 1) * 16)) & 0xffffu;
        while (f) {
          const int t = __builtin_ctz(f);
          f &= f - 1;
          const int chunk = (s0 + h) * T + t;
          unsigned long long m = mrow[chunk];
          ncand += (unsigned long long)__builtin_popcountll(m);
          while (m && nnz < (unsigned)K) {  // `if (num_inds == num_neighbors) break;` CvoGPU.cu:526
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            const int j = chunk * 64 + b;
            float a;
            float4 yt;
            if (!eval_pair(P, D, i, r, j, a, yt)) continue;
            if (a > P.sp_thres) {  // CvoGPU.cu:576-589
              D->ell_a[(size_t)nnz * N + i] = a;
              D->ell_j[(size_t)nnz * N + i] = j;
              nnz++;
              // compute_flow_gpu_no_eigen, CvoGPU.cu:758-782 (float accumulation in j order)
              const V3 pye{yt.x, yt.y, yt.z};
              const V3 cr = cross_dev(pxe, pye);
              const float dx = pye.x - pxe.x, dy = pye.y - pxe.y, dz = pye.z - pxe.z;
              o0 = __builtin_fmaf(cr.x, a, o0);
              o1 = __builtin_fmaf(cr.y, a, o1);
              o2 = __builtin_fmaf(cr.z, a, o2);
              v0 = __builtin_fmaf(dx, a, v0);
              v1 = __builtin_fmaf(dy, a, v1);
              v2 = __builtin_fmaf(dz, a, v2);
              asum += (double)a;
            }
          }
        }
      }
    }
    D->nnz_row[i] = nnz;
  }
  // per-row (omega_i / c, v_i / d) cast to double, then reduced in double (CvoGPU.cu:784-787, 824-825)
  double red[7] = {(double)(o0 / P.c), (double)(o1 / P.c), (double)(o2 / P.c), (double)(v0 / P.d),
                   (double)(v1 / P.d), (double)(v2 / P.d), asum};
  __shared__ double s_red[4][8];
  __shared__ unsigned long long s_cnt[4][3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 7; c++) red[c] = wave_sum(red[c]);
  unsigned long long nn = wave_sum_u64(nnz);
  unsigned mx = wave_max_u32(nnz);
  unsigned long long nc = wave_sum_u64(ncand);
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 7; c++) s_red[wave][c] = red[c];
    s_cnt[wave][0] = nn;
    s_cnt[wave][1] = mx;
    s_cnt[wave][2] = nc;
  }
  __syncthreads();
  if (threadIdx.x < 7) {
    const int c = threadIdx.x;
    D->flow_part[(size_t)blockIdx.x * 8 + c] = ((s_red[0][c] + s_red[1][c]) + s_red[2][c]) + s_red[3][c];
  } else if (threadIdx.x == 8) {
    D->cnt_part[(size_t)blockIdx.x * 4 + 0] = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0];
    D->cnt_part[(size_t)blockIdx.x * 4 + 1] = max(max(s_cnt[0][1], s_cnt[1][1]), max(s_cnt[2][1], s_cnt[3][1]));
    D->cnt_part[(size_t)blockIdx.x * 4 + 2] = s_cnt[0][2] + s_cnt[1][2] + s_cnt[2][2] + s_cnt[3][2];
  }
}

// ------------------------------------------------------------------------------------------
// k_coeff: normalised twist (compute_flow host half, CvoGPU.cu:824-835) + B,C,D,E partials.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_coeff(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp) {
  const PairDesc* __restrict__ D = descs + blockIdx.y;
  PairState* st = D->st;
  if (st->status != 0) return;
  const DevParams P = *Pp;
  if (P.mode != 0) return;
  __shared__ double s_ov[6];
  __shared__ XiMats s_M;
  __shared__ double s_red[4][4];
  const int nblk = D->nblk;
  if (threadIdx.x < 6) {  // sequential double sum over the row blocks (fixed order)
    double s = 0;
    for (int b = 0; b < nblk; b++) s += D->flow_part[(size_t)b * 8 + threadIdx.x];
    s_ov[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float ov[6];
    for (int c = 0; c < 6; c++) ov[c] = (float)s_ov[c];
    float z = 0;  // Eigen normalize(): z = squaredNorm(); if (z > 0) *this /= sqrt(z)
    for (int c = 0; c < 6; c++) z = z + ov[c] * ov[c];
    if (z > 0) {
      const float s = sqrtf(z);
      for (int c = 0; c < 6; c++) ov[c] = ov[c] / s;
    }
    xi_mats(ov, ov + 3, s_M);
    if (blockIdx.x == 0) {
      for (int c = 0; c < 3; c++) {
        st->omega[c] = ov[c];
        st->v[c] = ov[3 + c];
      }
    }
  }
  __syncthreads();
  const int N = D->N;
  const int i = blockIdx.x * 256 + threadIdx.x;
  double Bi = 0, Ci = 0, Di = 0, Ei = 0;
  if (i < N) {
    const unsigned nnz = D->nnz_row[i];
    if (nnz) {
      const float4 x = D->x4[i];
      float temp_ell = st->ell;
      if (P.use_range_ell) {
        const float d2_sqrt = sqrtf(dot3_dev(x.x, x.y, x.z, x.x, x.y, x.z));
        temp_ell = compute_range_ell(temp_ell, d2_sqrt);
      }
      const float temp_coef = (float)(1 / (2.0 * temp_ell * temp_ell));
      const V3 w{s_M.omega[0], s_M.omega[1], s_M.omega[2]};
      for (unsigned s = 0; s < nnz; s++) {
        const int idx = D->ell_j[(size_t)s * N + i];
        const float A_ij = D->ell_a[(size_t)s * N + i];
        const float4 y = D->yt4[idx];
        const V3 yy{y.x, y.y, y.z};
        // compute_step_size_xi for target idx (CvoGPU.cu:974-986)
        const V3 c = cross_dev(w, yy);
        const V3 xiz{c.x + s_M.v[0], c.y + s_M.v[1], c.z + s_M.v[2]};
        V3 t = matvec_dev(s_M.m2, yy);
        const V3 xi2z{t.x + s_M.ohv.x, t.y + s_M.ohv.y, t.z + s_M.ohv.z};
        t = matvec_dev(s_M.m3, yy);
        const V3 xi3z{t.x + s_M.m2v.x, t.y + s_M.m2v.y, t.z + s_M.m2v.z};
        t = matvec_dev(s_M.m4, yy);
        const V3 xi4z{t.x + s_M.m3v.x, t.y + s_M.m3v.y, t.z + s_M.m3v.z};
        const float normxiz2 = dot3_dev(xiz.x, xiz.y, xiz.z, xiz.x, xiz.y, xiz.z);
        const float xiz_dot_xi2z = -dot3_dev(xiz.x, xiz.y, xiz.z, xi2z.x, xi2z.y, xi2z.z);
        const float epsil_const = __builtin_fmaf(2.0f, dot3_dev(xiz.x, xiz.y, xiz.z, xi3z.x, xi3z.y, xi3z.z),
                                                 dot3_dev(xi2z.x, xi2z.y, xi2z.z, xi2z.x, xi2z.y, xi2z.z));
        // compute_step_size_poly_coeff (CvoGPU.cu:1053-1078)
        const float dfx = x.x - y.x, dfy = x.y - y.y, dfz = x.z - y.z;
        const float beta_ij = (float)(-2.0 * temp_coef * (double)dot3_dev(xiz.x, xiz.y, xiz.z, dfx, dfy, dfz));
        const float gamma_ij =
            (-temp_coef) * (normxiz2 + dot3_dev(2.0f * xi2z.x, 2.0f * xi2z.y, 2.0f * xi2z.z, dfx, dfy, dfz));
        const float delta_ij =
            (float)(2.0 * temp_coef * (double)(xiz_dot_xi2z + dot3_dev(-xi3z.x, -xi3z.y, -xi3z.z, dfx, dfy, dfz)));
        const float epsil_ij =
            (-temp_coef) * (epsil_const + dot3_dev(2.0f * xi4z.x, 2.0f * xi4z.y, 2.0f * xi4z.z, dfx, dfy, dfz));
        Bi += (double)(A_ij * beta_ij);
        Ci += (double)A_ij * ((double)gamma_ij + (double)(beta_ij * beta_ij) / 2.0);
        Di += (double)A_ij * ((double)__builtin_fmaf(beta_ij, gamma_ij, delta_ij) +
                              (double)(beta_ij * beta_ij * beta_ij) / 6.0);
        Ei += (double)A_ij * ((double)__builtin_fmaf(beta_ij, delta_ij, epsil_ij) +
                              1 / 2.0 * beta_ij * beta_ij * gamma_ij + 1 / 2.0 * gamma_ij * gamma_ij +
                              1 / 24.0 * beta_ij * beta_ij * beta_ij * beta_ij);
      }
    }
  }
  double red[4] = {Bi, Ci, Di, Ei};
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 4; c++) red[c] = wave_sum(red[c]);
  if (lane == 0)
    for (int c = 0; c < 4; c++) s_red[wave][c] = red[c];
  __syncthreads();
  if (threadIdx.x < 4) {
    const int c = threadIdx.x;
    D->coef_part[(size_t)blockIdx.x * 4 + c] = ((s_red[0][c] + s_red[1][c]) + s_red[2][c]) + s_red[3][c];
  }
}

// ------------------------------------------------------------------------------------------
// k_step: per-pair scalar bookkeeping + preparation of the next iteration.  INIT = true is the
// launch before the first iteration (no bookkeeping, state comes from the host).
// ------------------------------------------------------------------------------------------
constexpr int STEP_THREADS = 1024;
constexpr int XCULL_PAD = 32;  // >= 2 * the largest RU of k_scan

template <bool INIT>
__global__ __launch_bounds__(STEP_THREADS) void k_step(const PairDesc* __restrict__ descs,
                                                       const DevParams* __restrict__ Pp) {
  const PairDesc* __restrict__ D = descs + blockIdx.x;
  PairState* st = D->st;
  __shared__ float s_Ri[9], s_Ti[3];
  __shared__ float s_ell;
  __shared__ int s_done;
  __shared__ double s_c[4];
  __shared__ unsigned long long s_n[3];
  __shared__ float s_wmax[STEP_THREADS / 64];
  const DevParams P = *Pp;
  const int tid = threadIdx.x;
  if (!INIT) {
    if (st->status != 0) return;
    const int nblk = D->nblk;
    if (tid < 4) {  // the four thrust::reduce of compute_step_size (CvoGPU.cu:1118-1121)
      double s = 0;
      const int slot = P.mode == 0 ? tid : 0;
      if (P.mode == 0)
        for (int b = 0; b < nblk; b++) s += D->coef_part[(size_t)b * 4 + slot];
      else if (tid == 0)
        for (int b = 0; b < nblk; b++) s += D->flow_part[(size_t)b * 8 + 6];
      s_c[tid] = s;
    } else if (tid >= 64 && tid < 67) {
      const int c = tid - 64;
      unsigned long long s = 0;
      for (int b = 0; b < nblk; b++) {
        const unsigned long long q = D->cnt_part[(size_t)b * 4 + c];
        s = (c == 1) ? max(s, q) : s + q;
      }
      s_n[c] = s;
    }
    __syncthreads();
  }
  if (tid == 0) {
    int done = 0;
    if (!INIT) {
      const unsigned nnz = (unsigned)s_n[0], max_nnz = (unsigned)s_n[1];
      st->nnz = nnz;
      st->max_nnz = max_nnz;
      st->ncand = s_n[2];
      if (P.mode != 0) {  // single evaluation: A_sum (SparseKernelMat.cu:62-68)
        st->asum = s_c[0];
        done = 1;
      } else {
        const double B = s_c[0], C = s_c[1], Dd = s_c[2], E = s_c[3];
        st->B = B;
        st->C = C;
        st->D = Dd;
        st->E = E;
        const float step = select_step(B, C, Dd, E, P.min_step, P.max_step);
        st->step = step;
        const int k = st->k;
        const int K_used = st->K;
        const float ell_used = st->ell;
        float R[9], T[3];
        for (int q = 0; q < 9; q++) R[q] = st->R[q];
        for (int q = 0; q < 3; q++) T[q] = st->T[q];
        const float* om = st->omega;
        const float* vv = st->v;
        double dist = 0;
        auto norm3d = [](const float* a) {
          const double x = a[0], y = a[1], z = a[2];
          return sqrt(x * x + (y * y + z * z));
        };
        if (norm3d(om) < (double)P.eps && norm3d(vv) < (double)P.eps) {  // CvoGPU.cu:1454-1458
          auto norm3f = [](const float* a) { return sqrtf(a[0] * a[0] + (a[1] * a[1] + a[2] * a[2])); };
          if ((double)norm3f(om) < 1e-8 && (double)norm3f(vv) < 1e-8) st->ret = -1;
          done = 1;
          st->iterations = k;
        } else {
          const float xi[6] = {om[0], om[1], om[2], vv[0], vv[1], vv[2]};
          float dtrans[12];
          exp_sek3(xi, step, dtrans);  // CvoGPU.cu:1462
          double dR[9], dT[3];
          for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) dR[3 * i + j] = (double)dtrans[4 * i + j];
            dT[i] = (double)dtrans[4 * i + 3];
          }
          float Tn[3], Rn[9];  // CvoGPU.cu:1463-1469
          for (int i = 0; i < 3; i++) {
            const double r0 = R[3 * i + 0], r1 = R[3 * i + 1], r2 = R[3 * i + 2];
            Tn[i] = (float)((r0 * dT[0] + (r1 * dT[1] + r2 * dT[2])) + (double)T[i]);
            for (int j = 0; j < 3; j++)
              Rn[3 * i + j] = (float)(r0 * dR[0 + j] + (r1 * dR[3 + j] + r2 * dR[6 + j]));
          }
          for (int q = 0; q < 9; q++) st->R[q] = R[q] = Rn[q];
          for (int q = 0; q < 3; q++) st->T[q] = T[q] = Tn[q];
          dist = se3_log_norm(dR, dT);  // CvoGPU.cu:1473-1476
          const float ip_curr = (float)((double)nnz / sqrt((double)D->N * (double)D->M));  // 1486
          const bool need_decay_ell = indicator_update(st, ip_curr, P.window, P.stable_thr);
          if (dist < (double)P.eps_2) {  // CvoGPU.cu:1505-1508
            done = 1;
            st->iterations = k;
          } else {
            if (k > P.ell_decay_start && need_decay_ell) {  // CvoGPU.cu:1509-1513
              float e = ell_used * P.ell_decay_rate;
              if (e < P.ell_min) e = P.ell_min;
              st->ell = e;
            }
            st->K = min(P.K_max, (int)((double)max_nnz * 1.2));  // CvoGPU.cu:1529
            st->k = k + 1;
            if (k + 1 >= P.max_iter) {
              done = 1;
              st->iterations = k + 1;
            }
          }
        }
        st->dist = dist;
        // optional per-iteration trace (the reference's is_logging history files, CvoGPU.cu:1495-1503)
        if (D->trace && st->n_trace < P.trace_capacity &&
            (k < P.trace_dense || (P.trace_every > 0 && k % P.trace_every == 0))) {
          cvo_trace_t* tr = D->trace + st->n_trace;
          tr->k = k;
          tr->K = K_used;
          tr->ell = ell_used;
          tr->step = step;
          tr->nnz = nnz;
          tr->max_nnz = max_nnz;
          for (int q = 0; q < 3; q++) {
            tr->omega[q] = om[q];
            tr->v[q] = vv[q];
          }
          tr->B = B;
          tr->C = C;
          tr->D = Dd;
          tr->E = E;
          tr->dist = dist;
          for (int q = 0; q < 9; q++) tr->R[q] = R[q];
          for (int q = 0; q < 3; q++) tr->T[q] = T[q];
          st->n_trace++;
        }
      }
    }
    // update_tf (CvoGPU.cu:94-112): the transform applied next, and the returned matrix when done
    float Ri[9], Ti[3];
    update_tf(st->R, st->T, Ri, Ti);
    for (int q = 0; q < 9; q++) st->Rinv[q] = s_Ri[q] = Ri[q];
    for (int q = 0; q < 3; q++) st->Tinv[q] = s_Ti[q] = Ti[q];
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) st->out_T[4 * j + i] = Ri[3 * i + j];
      st->out_T[12 + i] = Ti[i];
    }
    st->out_T[3] = st->out_T[7] = st->out_T[11] = 0;
    st->out_T[15] = 1;
    if (done) {
      st->status = 1;
      *D->status_out = 1;
    }
    s_done = done;
    s_ell = st->ell;
  }
  __syncthreads();
  if (s_done) return;

  // ---- prepare the next iteration -----------------------------------------------------------
  // transform_pointcloud_thrust (CvoGPU_impl.cu:164-173) from the INITIAL cloud, plus the cull form
  const int M = D->M, Mpad = D->Mpad, N = D->N;
  const float cx = D->cx, cy = D->cy, cz = D->cz;
  float lmax = 0;
  for (int j = tid; j < Mpad; j += STEP_THREADS) {
    if (j < M) {
      const float4 p = D->y4[j];
      const V3 q = transform_point(s_Ri, s_Ti, p.x, p.y, p.z);
      D->yt4[j] = make_float4(q.x, q.y, q.z, 0.f);
      const float ux = q.x - cx, uy = q.y - cy, uz = q.z - cz;
      const float nn = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
      D->ycull[j] = make_float4(ux, uy, uz, nn);
      lmax = fmaxf(lmax, nn);
    } else {
      D->ycull[j] = make_float4(0.f, 0.f, 0.f, __builtin_inff());
    }
  }
  lmax = wave_max_f32(lmax);
  if ((tid & 63) == 0) s_wmax[tid >> 6] = lmax;
  __syncthreads();
  float ymax2 = 0;
#pragma unroll
  for (int w = 0; w < STEP_THREADS / 64; w++) ymax2 = fmaxf(ymax2, s_wmax[w]);
  // per-row constants of fill_in_A_mat_gpu (CvoGPU.cu:504-510) and the conservative cull operand
  const float ell = s_ell;
  for (int i = tid; i < N; i += STEP_THREADS) {
    const float4 x = D->x4[i];
    const float a_to_sensor = sqrtf(__builtin_fmaf(x.z, x.z, __builtin_fmaf(x.y, x.y, x.x * x.x)));
    const float l = compute_range_ell(ell, a_to_sensor);
    float thr = 1.f;
    if (P.use_geo) thr = (float)(-2.0 * l * l * (double)P.log_geo);
    D->rowc[i] = make_float2(l, thr);
    const float ux = x.x - cx, uy = x.y - cy, uz = x.z - cz;
    const float nx = __builtin_fmaf(uz, uz, __builtin_fmaf(uy, uy, ux * ux));
    const float margin = 4e-6f * (nx + ymax2) + 1e-5f * fabsf(thr);
    float cw = (thr + margin) - nx;
    if (!P.use_geo) cw = __builtin_inff();
    D->xcull[i] = make_float4(-2.f * ux, -2.f * uy, -2.f * uz, cw);
  }
  // never-passing pad rows so k_scan can run whole row groups and prefetch past the end
  for (int i = N + tid; i < N + XCULL_PAD; i += STEP_THREADS) D->xcull[i] = make_float4(0.f, 0.f, 0.f, -__builtin_inff());
}

}  // namespace cvo_dev
