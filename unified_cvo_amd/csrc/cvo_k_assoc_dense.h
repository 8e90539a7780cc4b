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

#ifndef CVO_DENSE_WAVES_PER_SIMD
#define CVO_DENSE_WAVES_PER_SIMD 5
#endif
// WIDE: the instantiation a small pair solved alone gets (a block per overflow row, see "wide rows" below); everybody else
// runs the one without that phase - 24 KB of LDS per block instead of 40, six blocks per CU instead of four: the dense
// kernels are most of a clustered BATCH's time, and there the waves in flight are throughput.
template <int FEAT, int DENSE_WAVES, bool WIDE>
__global__ __launch_bounds__(64 * DENSE_WAVES, (!WIDE && FEAT == FEAT_GEO) ? CVO_DENSE_WAVES_PER_SIMD : 1) void k_assoc_dense(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp,
                                                     const PairState* __restrict__ states) {
  const PairState* __restrict__ st = states + blockIdx.y;  // == D->st, as wave-uniform scalar loads (see k_coeff_dense)
  if (st->status != 0) return;
  const PairDesc* __restrict__ D = descs + blockIdx.y;
  const DevParams P = *Pp;
  const int N = D->N, M = D->M;
  const int K = st->K;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_ovf = st->n_ovf;
  if (n_ovf == 0) return;   // no row of this pair is beyond its cached list
  if (st->rebuild) return;  // (lean graphs with this kernel: the pair waits for its rebuild opportunity, see k_assoc)
  const bool all_dense = st->all_dense != 0;
  static_assert(DENSE_WAVES == 4, "the wide-row phase splits a row over four waves");
  // LDS.  Narrow rows (a wave per row), a region per wave: the hits of one step, compacted ({flow term, value} per
  // component) - and, before the row's first step, the sort keys of the long list being built (the list is written out and
  // read back from memory like a list of an earlier iteration: the keys are dead when the first hit is stored).  Wide rows
  // (the block per row, WIDE only) carve the same bytes again: wave 0's keys, then every wave's hits of its quarter of the
  // row ({value, transformed target}, column) and two 128-slot replay blocks.
  constexpr int WIDE_MIN = 256;   // candidates from which a row is worth the whole block
  constexpr int WIDE_CAP = 304;   // candidates (hence hits) a wave's quarter of a wide row can have: rows of up to 1216
  constexpr size_t WAVE_REGION = sizeof(float2) * 128 * 6;  // 6 KB >= the 4 KB of keys
  static_assert(WAVE_REGION >= sizeof(unsigned) * LONG_CAP, "a wave's keys fit its hit buffer");
  constexpr size_t WIDE_BYTES = sizeof(unsigned) * LONG_CAP + (sizeof(float4) + sizeof(int)) * DENSE_WAVES * (WIDE_CAP + 1) + 2 * sizeof(float2) * 128 * 6;
  constexpr size_t RAW_BYTES = WIDE && WIDE_BYTES > WAVE_REGION * DENSE_WAVES ? ((WIDE_BYTES + 255) / 256) * 256 : WAVE_REGION * DENSE_WAVES;
  __shared__ __attribute__((aligned(16))) char s_raw[RAW_BYTES];
  unsigned* const my_keys = reinterpret_cast<unsigned*>(s_raw + WAVE_REGION * wave);
  float2(*const my_hits)[6] = reinterpret_cast<float2(*)[6]>(s_raw + WAVE_REGION * wave);
  unsigned* const keys0 = reinterpret_cast<unsigned*>(s_raw);
  float4(*w_hit)[WIDE_CAP] = reinterpret_cast<float4(*)[WIDE_CAP]>(s_raw + sizeof(unsigned) * LONG_CAP);
  int(*w_col)[WIDE_CAP] = reinterpret_cast<int(*)[WIDE_CAP]>(s_raw + sizeof(unsigned) * LONG_CAP + sizeof(float4) * DENSE_WAVES * WIDE_CAP);
  float2(*w_rep)[128][6] = reinterpret_cast<float2(*)[128][6]>(s_raw + sizeof(unsigned) * LONG_CAP + (sizeof(float4) + sizeof(int)) * DENSE_WAVES * WIDE_CAP);
  int* s_wcnt = reinterpret_cast<int*>(s_raw + sizeof(unsigned) * LONG_CAP + (sizeof(float4) + sizeof(int)) * DENSE_WAVES * WIDE_CAP + 2 * sizeof(float2) * 128 * 6);
  // WIDE rows.  With a block per overflow row to spare (a small pair solved alone: the demo pair's 523 rows all scan 1080
  // targets, a wave at a time that is nine dependent steps and 19 of its 43 us per iteration) a row of more than WIDE_MIN
  // candidates whose quarters fit the buffers - its long list, or, rows beyond every list and the dense regime, all targets
  // of a small target cloud - is evaluated by the four waves of block q, a quarter each; see the second phase below.  With
  // more rows than blocks the waves are better spent on a row each (the replay is a serial chain either way).
  const bool wide_mode = WIDE && n_ovf <= (int)gridDim.x;
  auto wide_row = [&](int n_cand) { return wide_mode && n_cand > WIDE_MIN && n_cand <= DENSE_WAVES * WIDE_CAP; };
  {
    const Pose pose = load_pose(st);
    const FeatDen F = make_feat_den(P);
    const bool long_lists = !all_dense && P.long_lists != 0 && D->long_j != nullptr;
    const unsigned long long gen = (D->call_serial << 24) | (unsigned long long)((unsigned)st->n_builds & 0xffffffu);
    // The row's run in the row-major part of the ELL, laid out by k_list at the last rebuild (PairDesc::dense_rel /
    // word_base).  All rows of the pair or none: the runs' total has to fit the part (the demo pair, all of its 523 rows on a
    // cap of 256 of 256, stays slot-major); not in the dense regime, whose rows k_list does not list.
    // In the dense regime (every row is evaluated here, the thread-per-row kernels store nothing) the whole matrix is
    // row-major, K_max entries per row.
    const size_t upper_cap = ell_upper_capacity(N, P.K_max);
    const bool small = (size_t)N * (size_t)P.K_max < 0x7fffffffull;  // (run indices are ints)
    const bool use_runs = small && (all_dense || (size_t)D->word_base[(N + 63) >> 6] <= upper_cap);
    auto run_of = [&](int pos) {
      if (!use_runs) return -1;
      return all_dense ? pos * P.K_max : ELL_LOWER_SLOTS * N + D->word_base[pos >> 6] + D->dense_rel[pos];
    };
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
      bool listed = false;
      const unsigned short* lj = nullptr;
      if (long_lists) {
        const int cnt = __float_as_int(x.w);  // (k_list keeps the row's candidate count next to its coordinates)
        if (cnt <= LONG_CAP) {
          listed = true;
          n_cand = cnt;
          if (wide_row(n_cand)) continue;  // the whole block's, below
          lj = D->long_j + (size_t)q * LONG_CAP;
          if (D->long_stamp[q] != gen) {
            n_cand = build_long_list(D, P.T, D->rowperm[r_sorted], my_keys, lane);
            unsigned short* out = D->long_j + (size_t)q * LONG_CAP;
            for (int k = lane; k < n_cand; k += 64) out[k] = (unsigned short)(my_keys[k] & 0xffffu);
            if (lane == 0) D->long_stamp[q] = gen;
            // (the list is read back below by other lanes of this wave: the stores have reached the L2 first; no line of
            // this row's list can sit in this CU's L1 - nobody read it in this launch)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
      if (!listed && wide_row(n_cand)) continue;
      const int off = run_of(r_sorted);
      if (lane == 0) D->dense_off[r_sorted] = off;  // (what the readers of this iteration's matrix go by)
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
              const int p = (int)lj[c];
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
            D->ell[ell_index(N, (int)rank, r_sorted, off)] = make_ell(a[h], yt[h].x, yt[h].y, yt[h].z, psort[h]);
            if (P.keep_columns) D->ell_j[(size_t)rank * N + r_sorted] = listed ? D->yorder[col[h]] : col[h];
            // flow terms of this lane's pair (CvoGPU.cu:767-769)
            const V3 pye{yt[h].x, yt[h].y, yt[h].z};
            const V3 cr = cross_dev(pxe, pye);
            float2* slot = my_hits[nstaged + (int)below];
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
        // (sixteen hits per round trip to the LDS: the replay is one dependent chain per component, and with four reads in
        // flight it - not the evaluation - was most of a long row's time; the double sum of the values only where somebody
        // reads it: the single evaluations)
        const int c = lane < 6 ? lane : 0;
        const bool want_asum = P.mode != 0;
        int k = 0;
        for (; k + 16 <= nstaged; k += 16) {
          float2 e[16];
#pragma unroll
          for (int u = 0; u < 16; u++) e[u] = my_hits[k + u][c];
#pragma unroll
          for (int u = 0; u < 16; u++) acc = __builtin_fmaf(e[u].x, e[u].y, acc);
          if (want_asum) {
#pragma unroll
            for (int u = 0; u < 16; u++) asum += (double)e[u].y;
          }
        }
        for (; k + 4 <= nstaged; k += 4) {
          const float2 e0 = my_hits[k][c], e1 = my_hits[k + 1][c], e2 = my_hits[k + 2][c], e3 = my_hits[k + 3][c];
          acc = __builtin_fmaf(e0.x, e0.y, acc);
          acc = __builtin_fmaf(e1.x, e1.y, acc);
          acc = __builtin_fmaf(e2.x, e2.y, acc);
          acc = __builtin_fmaf(e3.x, e3.y, acc);
          if (want_asum) {
            asum += (double)e0.y;
            asum += (double)e1.y;
            asum += (double)e2.y;
            asum += (double)e3.y;
          }
        }
        for (; k < nstaged; k++) {
          const float2 e = my_hits[k][c];
          acc = __builtin_fmaf(e.x, e.y, acc);
          if (want_asum) asum += (double)e.y;
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
        D->nnz_row[r_sorted] = nnz | NNZ_DENSE_FLAG;
        D->rowres[r_sorted] = RowRes{{o0, o1, o2}, {v0, v1, v2}, asum};
      }
    }
    // ---- wide rows: block q evaluates row q.  Every wave evaluates a quarter of the row and compacts its hits in LDS; the
    // quarters' hit counts give every hit its slot (first-K exact: slots >= K are dropped, whichever quarter they come from),
    // the ELL entries leave in parallel, and the float flow sums are then replayed in ascending slot order by ONE wave - the
    // order, hence every bit, is that of the lone wave's scan.
    if (!wide_mode) return;
    __syncthreads();  // (the carve changes: every wave has left its narrow rows)
    const int q = (int)blockIdx.x;
    if (q >= n_ovf) return;
    const int r_sorted = all_dense ? q : D->ovf_rows[q];
    const float4 x = D->xp4[r_sorted];
    const int cnt = __float_as_int(x.w);
    const bool listed = long_lists && cnt <= LONG_CAP;
    const int n_cand = listed ? cnt : M;
    if (!wide_row(n_cand)) return;  // (block-uniform)
    const int i = D->ip[r_sorted];
    RowData r = make_row(P, x, st->ell);
    if (FEAT == FEAT_HOT) r.lid = D->xlid[i];
    const V3 pxe{x.x, x.y, x.z};
    const unsigned short* lj = D->long_j + (size_t)q * LONG_CAP;
    bool fresh = false;
    if (listed) {
      fresh = D->long_stamp[q] != gen;  // (read by everybody BEFORE wave 0 may move it)
      __syncthreads();
      if (fresh) {
        if (wave == 0) {
          const int nb = build_long_list(D, P.T, D->rowperm[r_sorted], keys0, lane);
          unsigned short* out = D->long_j + (size_t)q * LONG_CAP;
          for (int k = lane; k < nb; k += 64) out[k] = (unsigned short)(keys0[k] & 0xffffu);
          if (lane == 0) D->long_stamp[q] = gen;
        }
        __syncthreads();
      }
    }
    const int off = run_of(r_sorted);
    if (threadIdx.x == 0) D->dense_off[r_sorted] = off;
    // this wave's quarter: a multiple of 128 candidates, two chunks of 64 per step (independent evaluations in flight together)
    const int per = ((n_cand + 128 * DENSE_WAVES - 1) / (128 * DENSE_WAVES)) * 128;
    const int lo = wave * per, hi = min(n_cand, lo + per);
    int nh = 0;
    for (int c0 = lo; c0 < hi; c0 += 128) {
      float a[2] = {0.f, 0.f};
      float4 yt[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
      bool ok[2] = {false, false};
      int col[2] = {0, 0};
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int c = c0 + 64 * h + lane;
        if (c < hi) {
          if (listed) {
            col[h] = fresh ? (int)(keys0[c] & 0xffffu) : (int)lj[c];
            ok[h] = eval_pair<FEAT>(P, D, F, pose, i, r, col[h], D->ys4[col[h]], a[h], yt[h]) && (a[h] > P.sp_thres);
          } else {
            col[h] = c;
            ok[h] = eval_pair<FEAT>(P, D, F, pose, i, r, (FEAT != FEAT_GEO) ? D->yinv[c] : 0, D->y4[c], a[h], yt[h]) && (a[h] > P.sp_thres);
          }
        }
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const unsigned long long m = __ballot(ok[h]);
        const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (ok[h]) {
          w_hit[wave][nh + below] = make_float4(a[h], yt[h].x, yt[h].y, yt[h].z);
          w_col[wave][nh + below] = col[h];
        }
        nh += __builtin_popcountll(m);
      }
    }
    if (lane == 0) s_wcnt[wave] = nh;
    __syncthreads();
    // slots: quarter v holds slots [kb[v], kb[v + 1]) of the row's first K hits
    int kb[DENSE_WAVES + 1];
    kb[0] = 0;
#pragma unroll
    for (int v = 0; v < DENSE_WAVES; v++) kb[v + 1] = min(K, kb[v] + s_wcnt[v]);
    const int nnz = kb[DENSE_WAVES];
    const int base = kb[wave], keep = kb[wave + 1] - kb[wave];
    for (int idx = lane; idx < keep; idx += 64) {
      const float4 h = w_hit[wave][idx];
      const int col = w_col[wave][idx];
      const int psort = listed ? col : ((FEAT != FEAT_GEO || ELL8) ? D->yinv[col] : 0);
      D->ell[ell_index(N, base + idx, r_sorted, off)] = make_ell(h.x, h.y, h.z, h.w, psort);
      if (P.keep_columns) D->ell_j[(size_t)(base + idx) * N + r_sorted] = listed ? D->yorder[col] : col;
    }
    // ordered replay, 128 slots at a time: waves 2 and 3 lay the flow terms of block b + 1 out while wave 0 accumulates block b
    // (CvoGPU.cu:767-780) - the replay is one dependent chain per component and the only serial part of a wide row
    float acc = 0.f;
    double asum = 0;
    const bool want_asum = P.mode != 0;
    auto fill = [&](int s0, float2(*buf)[6]) {
      const int t = (int)threadIdx.x - 128;
      const int sl = s0 + t;
      if (t < 0 || sl >= nnz) return;
      int v = 0;
#pragma unroll
      for (int u = 1; u < DENSE_WAVES; u++) v += (sl >= kb[u]) ? 1 : 0;
      const float4 h = w_hit[v][sl - kb[v]];
      const V3 pye{h.y, h.z, h.w};
      const V3 cr = cross_dev(pxe, pye);
      float2* slot = buf[t];
      slot[0] = make_float2(cr.x, h.x);
      slot[1] = make_float2(cr.y, h.x);
      slot[2] = make_float2(cr.z, h.x);
      slot[3] = make_float2(pye.x - pxe.x, h.x);
      slot[4] = make_float2(pye.y - pxe.y, h.x);
      slot[5] = make_float2(pye.z - pxe.z, h.x);
    };
    fill(0, w_rep[0]);
    __syncthreads();
    for (int s0 = 0, b = 0; s0 < nnz; s0 += 128, b ^= 1) {
      if (s0 + 128 < nnz) fill(s0 + 128, w_rep[b ^ 1]);
      if (wave == 0) {
        const float2(*buf)[6] = w_rep[b];
        const int n_here = min(128, nnz - s0);
        const int c = lane < 6 ? lane : 0;
        int k = 0;
        for (; k + 16 <= n_here; k += 16) {
          float2 e[16];
#pragma unroll
          for (int u = 0; u < 16; u++) e[u] = buf[k + u][c];
#pragma unroll
          for (int u = 0; u < 16; u++) acc = __builtin_fmaf(e[u].x, e[u].y, acc);
          if (want_asum) {
#pragma unroll
            for (int u = 0; u < 16; u++) asum += (double)e[u].y;
          }
        }
        for (; k < n_here; k++) {
          const float2 e = buf[k][c];
          acc = __builtin_fmaf(e.x, e.y, acc);
          if (want_asum) asum += (double)e.y;
        }
      }
      __syncthreads();
    }
    if (wave == 0) {
      const float o0 = __shfl(acc, 0), o1 = __shfl(acc, 1), o2 = __shfl(acc, 2);
      const float v0 = __shfl(acc, 3), v1 = __shfl(acc, 4), v2 = __shfl(acc, 5);
      if (lane == 0) {
        D->nnz_row[r_sorted] = (unsigned)nnz | NNZ_DENSE_FLAG;
        D->rowres[r_sorted] = RowRes{{o0, o1, o2}, {v0, v1, v2}, asum};
      }
    }
  }
}

}  // namespace cvo_dev
