// cvo_k_coeff_dense.h -- k_coeff_dense: the coefficient pass of the rows beyond their cached lists, a wave per row.
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_k_coeff.h"

namespace cvo_dev {

// ------------------------------------------------------------------------------------------
// k_coeff_dense (graphs with the dense kernels, after k_assoc, before k_coeff).  The rows k_assoc_dense evaluated carry
// hundreds of nonzeros (a clustered cloud, rows on the K cap): one thread walking such a row - compute_step_size_xi +
// _poly_coeff per nonzero, CvoGPU.cu:953-1082 - was the longest kernel of a clustered pair's iteration, and spreading a
// row's slots over more blocks while rows were long made the order of a row's sum depend on that choice.  Here a wave
// owns the row: its 64 lanes evaluate 64 nonzeros' terms at a time, and the sums are then formed EXACTLY as a thread of
// k_coeff forms them - slice q of the pair's own coefficient split adds the terms of slots q, q + csplit, ... one after
// the other, in double - by one lane per (slice, component) reading the terms back from LDS.  The result goes to
// PairDesc::rowcoef; k_coeff picks it up at the row's position.  Whether a row is evaluated here or by a thread of
// k_coeff does not reach a bit of B, C, D, E: the row classes (PairState::row_max) are free to follow the launch.
// ------------------------------------------------------------------------------------------
// rows_per_wave: 1 - a wave per row, what a few pairs in flight want (the kernel lasts as long as its longest row) - or 8: a
// wave takes eight consecutive overflow rows at once, lane l the slots l / 8, l / 8 + 8, ... of row l % 8, so that a batch of
// 64 terms holds eight slots of every row and the eight rows' ordered sums advance side by side (32 chain lanes instead of
// 4, full term batches for rows of any length): the choice of a chip full of pairs, where these kernels are throughput.
// Pairs with a coefficient split (small clouds) always take the wave per row.  Neither reaches a bit: a row's terms are
// added in slot order either way.
template <int DENSE_WAVES>
__global__ __launch_bounds__(64 * DENSE_WAVES) void k_coeff_dense(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp,
                                                                  const PairState* __restrict__ states, int rows_per_wave) {
  // (the state through the kernel-argument array: wave-uniform SCALAR loads - through the descriptor's pointer the twist
  // matrices alone were 42 VGPRs)
  const PairState* __restrict__ st = states + blockIdx.y;  // == D->st
  if (st->status != 0) return;
  const PairDesc* __restrict__ D = descs + blockIdx.y;
  const DevParams P = *Pp;
  if (P.mode != 0) return;
  const int n_ovf = st->n_ovf;
  if (n_ovf == 0 || st->rebuild) return;  // (see k_assoc_dense)
  const int N = D->N, csplit = D->csplit;
  const bool all_dense = st->all_dense != 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  XiMats Mu;
  {
    float* mu = reinterpret_cast<float*>(&Mu);
#pragma unroll
    for (int q = 0; q < (int)(sizeof(XiMats) / sizeof(float)); q++) mu[q] = st->xi[q];
  }
  const float ell = st->ell, coef_ell = st->temp_coef;
  __shared__ double s_terms[DENSE_WAVES][64][4];
  if (rows_per_wave == 8 && csplit == 1) {
    constexpr int R = 8, SL = 64 / R;
    const int rsel = lane % R, jsel = lane / R;  // this lane's row of the group, its slot inside a batch
    const int rc = lane >> 2, comp = lane & 3;   // chain lanes (lane < 4 R): row and component
    for (int g = blockIdx.x * DENSE_WAVES + wave; g * R < n_ovf; g += (int)gridDim.x * DENSE_WAVES) {
      const int q = g * R + rsel;
      const bool have = q < n_ovf;
      const int pos = have ? (all_dense ? q : D->ovf_rows[q]) : 0;
      const unsigned nnz = have ? nnz_count(D->nnz_row[pos]) : 0u;
      const int off = have ? D->dense_off[pos] : -1;
      const float4 x = D->xp4[pos];
      float temp_coef = coef_ell;
      if (P.use_range_ell) {  // CvoGPU.cu:1035-1037
        const float d2_sqrt = sqrtf(dot3_dev(x.x, x.y, x.z, x.x, x.y, x.z));
        temp_coef = coef_of_ell(compute_range_ell(ell, d2_sqrt));
      }
      const unsigned max_nnz = wave_max_u32(nnz);
      // what the chain lane (rc, comp) needs of ITS row: lanes 0 .. R - 1 hold rows 0 .. R - 1
      const unsigned nnz_rc = (unsigned)__shfl((int)nnz, rc & (R - 1));
      const int pos_rc = __shfl(pos, rc & (R - 1));
      const bool have_rc = lane < 4 * R && g * R + rc < n_ovf;
      double acc = 0;
      for (unsigned b0 = 0; b0 < max_nnz; b0 += (unsigned)SL) {
        const unsigned s = b0 + (unsigned)jsel;
        double t[4] = {0, 0, 0, 0};
        if (s < nnz) {
          const EllEntry e = D->ell[ell_index(N, (int)s, pos, off)];
#ifdef CVO_ELL8
          const f32x4 y0 = ((const CVO_GLOBAL f32x4*)D->ys4)[e.p];
          const V3 yy = transform_point(st->Rinv, st->Tinv, y0.x, y0.y, y0.z);
#else
          const V3 yy{e.yx, e.yy, e.yz};
#endif
          coeff_terms(Mu, x, temp_coef, yy, e.a, t);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) s_terms[wave][lane][c] = t[c];
        __builtin_amdgcn_wave_barrier();
        if (have_rc) {
#pragma unroll
          for (int j = 0; j < SL; j++)
            if (b0 + (unsigned)j < nnz_rc) acc += s_terms[wave][j * R + rc][comp];  // slot b0 + j of row rc
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (have_rc) D->rowcoef[(size_t)pos_rc * 4 + comp] = acc;
    }
    return;
  }
  for (int q = blockIdx.x * DENSE_WAVES + wave; q < n_ovf; q += (int)gridDim.x * DENSE_WAVES) {
    const int pos = all_dense ? q : D->ovf_rows[q];
    const unsigned nnz = nnz_count(D->nnz_row[pos]);
    const int off = D->dense_off[pos];  // the row's row-major run (k_assoc_dense), -1: slot-major
    const float4 x = D->xp4[pos];
    float temp_coef = coef_ell;
    if (P.use_range_ell) {  // CvoGPU.cu:1035-1037
      const float d2_sqrt = sqrtf(dot3_dev(x.x, x.y, x.z, x.x, x.y, x.z));
      temp_coef = coef_of_ell(compute_range_ell(ell, d2_sqrt));
    }
    // chains: (slice, component) = lane / 4, lane % 4, and a second one 16 slices further for splits above 16
    double acc0 = 0, acc1 = 0;
    const int qs0 = lane >> 2, qs1 = 16 + (lane >> 2), comp = lane & 3;
    for (unsigned s0 = 0; s0 < nnz; s0 += 64u) {
      const unsigned s = s0 + (unsigned)lane;
      double t[4] = {0, 0, 0, 0};
      if (s < nnz) {
        const EllEntry e = D->ell[ell_index(N, (int)s, pos, off)];
#ifdef CVO_ELL8
        const f32x4 y0 = ((const CVO_GLOBAL f32x4*)D->ys4)[e.p];
        const V3 yy = transform_point(st->Rinv, st->Tinv, y0.x, y0.y, y0.z);
#else
        const V3 yy{e.yx, e.yy, e.yz};
#endif
        coeff_terms(Mu, x, temp_coef, yy, e.a, t);
      }
#pragma unroll
      for (int c = 0; c < 4; c++) s_terms[wave][lane][c] = t[c];
      __builtin_amdgcn_wave_barrier();  // (one wave: its LDS operations complete in order)
      const int n_here = (int)min(64u, nnz - s0);
      // slot s belongs to slice s % csplit; s0 is a multiple of 64 and csplit divides 64: local index l, slice l % csplit
      // (eight terms per round trip to the LDS: the sum is one dependent chain per (slice, component))
      auto chain = [&](int qs, double acc) {
        int l = qs;
        for (; l + 7 * csplit < n_here; l += 8 * csplit) {
          double v[8];
#pragma unroll
          for (int u = 0; u < 8; u++) v[u] = s_terms[wave][l + u * csplit][comp];
#pragma unroll
          for (int u = 0; u < 8; u++) acc += v[u];
        }
        for (; l < n_here; l += csplit) acc += s_terms[wave][l][comp];
        return acc;
      };
      if (qs0 < csplit) acc0 = chain(qs0, acc0);
      if (qs1 < csplit) acc1 = chain(qs1, acc1);
      __builtin_amdgcn_wave_barrier();
    }
    double* rc = D->rowcoef + (size_t)pos * csplit * 4;
    if (qs0 < csplit) rc[qs0 * 4 + comp] = acc0;
    if (qs1 < csplit) rc[qs1 * 4 + comp] = acc1;
  }
}

}  // namespace cvo_dev
