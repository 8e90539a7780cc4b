// cvo_k_assoc_dense.h -- k_assoc_dense: rows beyond the cached lists, a wave per row (long lists / literal ordered scan).
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_k_assoc.h"

namespace cvo_dev {

// ------------------------------------------------------------------------------------------
// k_assoc_dense: the rows k_assoc could not list (more than ASSOC_CAP candidates).  One wave per row at a time, 64
// candidates per lane step: lanes evaluate the exact pair arithmetic in parallel, a ballot + prefix count gives every
// hit its ELL slot in ascending j (so the first-K truncation and its early exit are exact), and the float flow
// accumulation of compute_flow_gpu_no_eigen is replayed serially in lane (= j) order.
//   * rows with at most LONG_CAP candidates walk a LONG LIST: the row's candidates from the bitmap, sorted by original
//     target index (the order of the reference's scan, CvoGPU.cu:522-590).  The wave that owns the row builds the list
//     the first time it meets the row after a rebuild (decode, bitonic sort of (j << 16 | position) keys in LDS) and
//     leaves it in HBM for the iterations that follow - a clustered cloud has thousands of rows with a few hundred
//     neighbours each, and scanning all M targets for each of them cost 60x the slab's iteration (profiles/r4/scene.txt);
//   * the others (and every row in the dense regime) run the literal ordered scan over ALL targets.
// ------------------------------------------------------------------------------------------
// The candidates of sorted row rr from the bitmap -> keys[0 .. cnt) = (original index << 16 | sorted position), ascending.
__device__ __forceinline__ int build_long_list(const PairDesc* __restrict__ D, const int T, const int rr, unsigned* keys,
                                               const int lane) {
  const int N = D->N;
  const int rbw = D->rbw;
  const unsigned* rb = D->rowbits + (size_t)rr * rbw;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int cnt = 0;
  const int wpr = 32 * T;  // mask words behind one word of slice bits
  for (int w = 0; w < rbw; w++) {
    const unsigned f = rb[w];  // (uniform)
    if (f == 0) continue;
    for (int h = 0; h < wpr; h += 64) {
      // lane l: mask word h + l of this group = slice w * 32 + (h + l) / T, word (h + l) % T
      const int l2 = h + lane;
      const int sl = w * 32 + l2 / T;
      unsigned long long m = 0;
      if (l2 < wpr && ((f >> (l2 / T)) & 1u)) m = D->masks[((size_t)sl * N + rr) * T + (l2 % T)];
      unsigned long long todo = __ballot(m != 0ull);
      while (todo) {
        const int l = __builtin_ctzll(todo);
        todo &= todo - 1;
        const unsigned long long mm = lane_u64(m, l);
        const int chunk = w * wpr + h + l;  // == sl * T + t of lane l
        if ((mm >> lane) & 1ull) {
          const int idx = cnt + __builtin_popcountll(mm & lt);
          if (idx < LONG_CAP) keys[idx] = (unsigned)(chunk * 64 + lane);
        }
        cnt += __builtin_popcountll(mm);
      }
    }
  }
  cnt = min(cnt, LONG_CAP);  // (the caller only comes here with a count that fits)
  __builtin_amdgcn_wave_barrier();
  int p2 = 64;
  while (p2 < cnt) p2 <<= 1;
  const int* yorder = D->yorder;
  for (int k = lane; k < p2; k += 64) {
    unsigned key = 0xffffffffu;
    if (k < cnt) {
      const unsigned p = keys[k];
      key = ((unsigned)yorder[p] << 16) | p;
    }
    keys[k] = key;
  }
  __builtin_amdgcn_wave_barrier();
  // bitonic sort, ascending (the LDS operations of one wave complete in order; the barriers only pin the compiler)
  for (int k = 2; k <= p2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < (p2 >> 1); t += 64) {
        const int i1 = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int i2 = i1 | j;
        const unsigned a = keys[i1], b = keys[i2];
        const bool up = (i1 & k) == 0;
        if ((a > b) == up) {
          keys[i1] = b;
          keys[i2] = a;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  return cnt;
}

template <int FEAT, int DENSE_WAVES>
__global__ __launch_bounds__(64 * DENSE_WAVES) void k_assoc_dense(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp,
                                                     const int* __restrict__ status) {
  if (status[blockIdx.y] != 0) return;
  const PairDesc* __restrict__ D = descs + blockIdx.y;
  const PairState* st = D->st;
  const DevParams P = *Pp;
  const int N = D->N, M = D->M;
  const int K = st->K;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_ovf = st->n_ovf;
  if (n_ovf == 0) return;   // no row of this pair is beyond its cached list
  if (st->rebuild) return;  // (lean graphs with this kernel: the pair waits for its rebuild opportunity, see k_assoc)
  const bool all_dense = st->all_dense != 0;
  __shared__ float2 s_hits[DENSE_WAVES][128][6];  // per wave: the hits of one step, compacted ({flow term, value} per component)
  __shared__ unsigned s_keys[DENSE_WAVES][LONG_CAP];  // per wave: the long list being built (sort keys)
  {
    const Pose pose = load_pose(st);
    const FeatDen F = make_feat_den(P);
    const bool long_lists = !all_dense && P.long_lists != 0 && D->long_j != nullptr;
    const unsigned long long gen = (D->call_serial << 24) | (unsigned long long)((unsigned)st->n_builds & 0xffffffu);
    for (int q = blockIdx.x * DENSE_WAVES + wave; q < n_ovf; q += (int)gridDim.x * DENSE_WAVES) {
      // a position of k_list's ordering (all per-row outputs are stored by position); dense regime: every row
      const int r_sorted = all_dense ? q : D->ovf_rows[q];
      const int i = D->ip[r_sorted];
      const float4 x = D->xp4[r_sorted];
      RowData r = make_row(P, x, st->ell);
      if (FEAT == FEAT_HOT) r.lid = D->xlid[i];
      const V3 pxe{x.x, x.y, x.z};
      // where this row's candidates come from: its long list (built now if it is not the current one) or all targets
      int n_cand = M;
      bool listed = false, fresh = false;
      const unsigned short* lj = nullptr;
      if (long_lists) {
        const int cnt = __float_as_int(x.w);  // (k_list keeps the row's candidate count next to its coordinates)
        if (cnt <= LONG_CAP) {
          listed = true;
          n_cand = cnt;
          lj = D->long_j + (size_t)q * LONG_CAP;
          if (D->long_stamp[q] != gen) {
            n_cand = build_long_list(D, P.T, D->rowperm[r_sorted], s_keys[wave], lane);
            fresh = true;
            unsigned short* out = D->long_j + (size_t)q * LONG_CAP;
            for (int k = lane; k < n_cand; k += 64) out[k] = (unsigned short)(s_keys[wave][k] & 0xffffu);
            if (lane == 0) D->long_stamp[q] = gen;
          }
        }
      }
      unsigned nnz = 0;
      // Two chunks of 64 candidates per step: their (independent) evaluations overlap in the pipeline; if the first one
      // already fills the row, the second was evaluated for nothing.  Hits are compacted into LDS in ascending j
      // (slot = rank inside the step), then lanes 0..5 replay the reference's ordered float accumulation, one
      // component each (one LDS read + one FMA per hit and lane; lane 6 carries the double sum of the values).
      float acc = 0.f;   // lanes 0..2: omega_i, lanes 3..5: v_i  (CvoGPU.cu:779-780)
      double asum = 0;   // lane 6
      for (int j0 = 0; j0 < n_cand && nnz < (unsigned)K; j0 += 128) {
        float a[2] = {0.f, 0.f};
        float4 yt[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
        bool ok[2] = {false, false};
        int col[2] = {0, 0}, psort[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int c = j0 + 64 * h + lane;
          if (c < n_cand) {
            if (listed) {  // list entries are sorted positions: coordinates and features from the spatially ordered arrays
              const int p = fresh ? (int)(s_keys[wave][c] & 0xffffu) : (int)lj[c];
              col[h] = p;
              psort[h] = p;
              ok[h] = eval_pair<FEAT>(P, D, F, pose, i, r, p, D->ys4[p], a[h], yt[h]) && (a[h] > P.sp_thres);
            } else {
              col[h] = c;
              psort[h] = (FEAT != FEAT_GEO || ELL8) ? D->yinv[c] : 0;
              ok[h] = eval_pair<FEAT>(P, D, F, pose, i, r, psort[h], D->y4[c], a[h], yt[h]) && (a[h] > P.sp_thres);
            }
          }
        }
        int nstaged = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const unsigned long long m = __ballot(ok[h]);
          const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          const unsigned rank = nnz + below;
          const bool keep = ok[h] && rank < (unsigned)K;  // `if (num_inds == num_neighbors) break;`
          if (keep) {
            D->ell[(size_t)rank * N + r_sorted] = make_ell(a[h], yt[h].x, yt[h].y, yt[h].z, psort[h]);
            if (P.keep_columns) D->ell_j[(size_t)rank * N + r_sorted] = listed ? D->yorder[col[h]] : col[h];
            // flow terms of this lane's pair (CvoGPU.cu:767-769)
            const V3 pye{yt[h].x, yt[h].y, yt[h].z};
            const V3 cr = cross_dev(pxe, pye);
            float2* slot = s_hits[wave][nstaged + (int)below];
            slot[0] = make_float2(cr.x, a[h]);
            slot[1] = make_float2(cr.y, a[h]);
            slot[2] = make_float2(cr.z, a[h]);
            slot[3] = make_float2(pye.x - pxe.x, a[h]);
            slot[4] = make_float2(pye.y - pxe.y, a[h]);
            slot[5] = make_float2(pye.z - pxe.z, a[h]);
          }
          const int nkeep = __builtin_popcountll(__ballot(keep));
          nnz += (unsigned)nkeep;
          nstaged += nkeep;
        }
        __builtin_amdgcn_wave_barrier();  // (same wave wrote the slots: LDS operations of a wave complete in order)
        const int c = lane < 6 ? lane : 0;
        int k = 0;
        for (; k + 4 <= nstaged; k += 4) {
          const float2 e0 = s_hits[wave][k][c], e1 = s_hits[wave][k + 1][c], e2 = s_hits[wave][k + 2][c],
                       e3 = s_hits[wave][k + 3][c];
          acc = __builtin_fmaf(e0.x, e0.y, acc);
          acc = __builtin_fmaf(e1.x, e1.y, acc);
          acc = __builtin_fmaf(e2.x, e2.y, acc);
          acc = __builtin_fmaf(e3.x, e3.y, acc);
          asum += (double)e0.y;
          asum += (double)e1.y;
          asum += (double)e2.y;
          asum += (double)e3.y;
        }
        for (; k < nstaged; k++) {
          const float2 e = s_hits[wave][k][c];
          acc = __builtin_fmaf(e.x, e.y, acc);
          asum += (double)e.y;
        }
        __builtin_amdgcn_wave_barrier();
      }
      const float o0 = __shfl(acc, 0), o1 = __shfl(acc, 1), o2 = __shfl(acc, 2);
      const float v0 = __shfl(acc, 3), v1 = __shfl(acc, 4), v2 = __shfl(acc, 5);
      // The row's result - nonzero count and its float flow sums, as compute_flow_gpu_no_eigen leaves them before the
      // division by c / d (CvoGPU.cu:779-787) - for the thread of k_assoc that owns the position: k_assoc runs AFTER
      // this kernel and reduces every row of the pair, whoever evaluated it, in one fixed order (no partial of this
      // kernel's own: a pair's sums do not depend on this launch's grid, i.e. on how many pairs are in flight).
      if (lane == 0) {
        D->nnz_row[r_sorted] = nnz;
        D->rowres[r_sorted] = RowRes{{o0, o1, o2}, {v0, v1, v2}, asum};
      }
    }
  }
}

}  // namespace cvo_dev
