// cvo_k_list.h -- k_list: sorted per-row candidate lists from the bitmap, rows re-ordered by candidate count.
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_wave.h"
#include "cvo_pair_math.h"

namespace cvo_dev {

// ------------------------------------------------------------------------------------------
// k_list: runs only when the bitmap was rebuilt.  A block owns a window of LIST_THREADS consecutive (sorted)
// source rows.  It counts every row's candidates, then re-orders the rows of the window by that count: position
// p of the window holds the row with the p-th smallest count (stable).  Everything the per-iteration kernels touch
// is stored by POSITION (lists, counts, row coordinates, ELL), so their loads stay coalesced while the 64 lanes
// of a wave get rows with similar trip counts - the association and coefficient loops are thread-per-row and a
// wave runs as long as its longest row.  One thread per position then decodes its row's candidates from the
// bitmap, maps them to original target indices and sorts them ascending ([slot][position], coalesced); the list
// serves every iteration until the next rebuild.  Rows with more candidates than a list holds go to the
// overflow list of k_assoc_dense (also cached).
// ------------------------------------------------------------------------------------------
constexpr int LIST_THREADS = 256;
#ifndef CVO_LIST_RB
#define CVO_LIST_RB 8
#endif
constexpr int LIST_RB = CVO_LIST_RB;  // candidates ranked per sweep of a row's list in k_list (2 / 4 / 8 / 16: 59.7 / 59.5 / 59.1 / 59.5 ms per step)

// A row's candidates in ascending ORIGINAL target index, 16-bit positions: NN keys (original index << 16 | sorted position)
// in REGISTERS - every index below is a compile-time constant once the loops are unrolled -, sorted by a bitonic network of
// v_min_u32 / v_max_u32 pairs (NN = 8 / 16 / 32 / 64: 24 / 80 / 240 / 672 compare-exchanges) and written to their
// slots.  The rank sort this replaces cost cnt^2 compares and cnt^2 / LIST_RB LDS reads per row plus a second dependent
// gather (position -> index -> position) per candidate: the longest row of a wave decided k_list's time, and k_list a
// fifth of the first iterations of a batch.  The keys of a row are distinct: any correct sort leaves the same list.
template <int NN>
__device__ __forceinline__ void sort_network(unsigned (&v)[NN]) {
#pragma unroll
  for (int k = 2; k <= NN; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < NN; i++) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned a = v[i], b = v[l];
          const unsigned lo = min(a, b), hi = max(a, b);
          const bool asc = (i & k) == 0;
          v[i] = asc ? lo : hi;
          v[l] = asc ? hi : lo;
        }
      }
    }
  }
}
template <int NN>
__device__ __forceinline__ void sort_and_store_list(const unsigned short* list, const int cnt, const int* __restrict__ yorder,
                                                    unsigned short* __restrict__ out, const int N, const int pos) {
  unsigned v[NN];
#pragma unroll
  for (int k = 0; k < NN; k++) {  // all gathers of the row are in flight together
    const unsigned p = list[k];   // (slots >= cnt hold leftovers of earlier rows: valid LDS, masked below)
    v[k] = k < cnt ? (((unsigned)yorder[k < cnt ? p : 0u] << 16) | p) : 0xffffffffu;
  }
  sort_network<NN>(v);
#pragma unroll
  for (int k = 0; k < NN; k++)
    if (k < cnt) out[(size_t)k * N + pos] = (unsigned short)(v[k] & 0xffffu);
}

template <typename IdxT, int ASSOC_CAP>
__global__ __launch_bounds__(LIST_THREADS) void k_list(const PairDesc* __restrict__ descs,
                                                        const DevParams* __restrict__ Pp,
                                                        const PairState* __restrict__ states, int nblk, int n_pairs) {
  constexpr int ASSOC_STRIDE = ASSOC_CAP + 1;  // odd stride: conflict-free per-thread lists
  PairBlock pb;
  if (!pair_block(nblk, n_pairs, pb)) return;
  {
    const PairState* __restrict__ st0 = states + pb.pair;  // == D->st (see k_scan)
    const int status_v = st0->status, rebuild_v = st0->rebuild;
    if (status_v != 0 || !rebuild_v) return;
  }
  const PairDesc* __restrict__ D = descs + pb.pair;
  const int N = D->N;
  const int T = Pp->T;
  const int rbw = D->rbw;
  __shared__ IdxT s_list[LIST_THREADS * ASSOC_STRIDE];
  __shared__ int s_row[LIST_THREADS];
  const int row_max = min(D->st->row_max, ASSOC_CAP);  // rows with more candidates go to k_assoc_dense (PairState::row_max)
  const int tid = threadIdx.x;
  const int w0row = pb.bx * LIST_THREADS;
  // ---- candidates of row w0row + tid
  int ncand = ASSOC_CAP + 2;  // rows past N sort behind every real row
  if (w0row + tid < N) {  // accumulated by k_scan's emission; dense regime: every row takes the overflow path
    const int rc = D->row_cnt[w0row + tid];
    ncand = D->st->all_dense ? ASSOC_CAP + 1 : rc;
  }
  // (Round 3 tried super-windows of 1024 rows - every block ranking the 1024 rows around its own 256 positions: the sum
  // over the waves of their longest row drops by 20 %, 614 -> 481 at ell = 0.15, all tests green - and the 64-pair step
  // went from 69.0 to 74.2 ms: the long rows of 1024 rows then sit together in one block, whose four waves all run long,
  // and a sub-batch's chain waits for its slowest block; the rows of a wave are also spatial neighbours only at the
  // 1024-row scale, so their candidate gathers share fewer cache lines.  256-row windows stay.)
  // ---- stable rank by key = min(count, CAP + 1) (overflow rows last, pad rows behind them): a counting sort.  Every
  // wave finds, key by key among the keys it holds, how many of its lanes have that key and where a lane stands among
  // them (ballots); the per-wave counts meet in LDS, one wave turns them into the first position of every key.
  // (~200 wave instructions; counting the 256 keys that sort before one's own took ~1 100.)
  constexpr int NKEY = ASSOC_CAP + 3;
  constexpr int NWV = LIST_THREADS / 64;
  __shared__ int s_hist[NWV][NKEY];
  __shared__ int s_first[NKEY];
  const int key = (w0row + tid < N) ? min(ncand, ASSOC_CAP + 1) : ASSOC_CAP + 2;
  for (int q = tid; q < NWV * NKEY; q += LIST_THREADS) (&s_hist[0][0])[q] = 0;
  __syncthreads();
  int eq_lower = 0;
  {
    const int wv = tid >> 6;
    const unsigned lo = __builtin_amdgcn_mbcnt_lo(~0u, 0u);
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, lo);
    unsigned long long todo = __ballot(true);
    while (todo) {
      const int leader = __builtin_ctzll(todo);
      const int k0 = __builtin_amdgcn_readlane(key, leader);
      const unsigned long long m = __ballot(key == k0);
      if (key == k0) eq_lower = __builtin_popcountll(m & ((1ull << lane) - 1ull));
      if ((int)lane == leader) s_hist[wv][k0] = __builtin_popcountll(m);
      todo &= ~m;
    }
  }
  __syncthreads();
  if (tid < 64) {  // first position of every key: exclusive prefix of the keys' totals (NKEY <= 128: two per lane)
    int t0 = 0, t1 = 0;
#pragma unroll
    for (int w = 0; w < NWV; w++) {
      t0 += (2 * tid < NKEY) ? s_hist[w][2 * tid] : 0;
      t1 += (2 * tid + 1 < NKEY) ? s_hist[w][2 * tid + 1] : 0;
    }
    int incl = t0 + t1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (tid >= o) incl += v;
    }
    const int excl = incl - (t0 + t1);
    if (2 * tid < NKEY) s_first[2 * tid] = excl;
    if (2 * tid + 1 < NKEY) s_first[2 * tid + 1] = excl + t0;
  }
  __syncthreads();
  {
    int rank = s_first[key] + eq_lower;
    for (int w = 0; w < (tid >> 6); w++) rank += s_hist[w][key];
    s_row[rank] = tid | (ncand << 8);  // position `rank` of the window holds row tid (ncand <= ~M < 2^23)
  }
  __syncthreads();
  // ---- position w0row + tid: build the list of the row that was ranked there
  const int pos = w0row + tid;
  const int rr = w0row + (s_row[tid] & 0xff);
  const int cnt_all = s_row[tid] >> 8;
  IdxT* list = s_list + tid * ASSOC_STRIDE;
  int cnt = 0;  // candidates this thread lists (0: a pad position, or a row for k_assoc_dense)
  if (rr < N) {  // real rows occupy the positions below N
    D->cand_cnt[pos] = cnt_all;
    D->rowperm[pos] = rr;
    {  // the row's head for the per-iteration kernels: coordinates + its candidate count in one 16-byte record
      float4 xh = D->xs4[rr];
      xh.w = __int_as_float(cnt_all);
      D->xp4[pos] = xh;
    }
    D->ip[pos] = rr;  // the row's index into the (spatially ordered) feature arrays
    D->iorig[pos] = D->xorder[rr];
    if (cnt_all > row_max) {
      // more candidates than a list holds (dense regime, e.g. rows sitting on K_max): k_assoc_dense
      // evaluates these rows against all targets, 64 at a time.  Only flagged here (below, one bit per position): the
      // list itself is written in ascending position order by the block that finishes last, so that the order in
      // which k_assoc_dense's waves accumulate their rows never depends on the arrival order of atomics.
    } else {
      const unsigned* rb = D->rowbits + (size_t)rr * rbw;
      // sorted-space positions of the candidates (the mask words come from L1/L2 this time)
      // The slices the row has candidates in (its rowbits) and their mask words: up to WQ words are requested together and
      // decoded when they have arrived - one round trip per batch instead of one per slice on every thread's serial chain
      // (a row of the slab touches 1-4 slices, a clustered one a dozen).
      constexpr int WQ = 8;  // (T = 1, 2, 4 or 8 words per slice: a slice's words always fit the rest of a batch)
      int ch_q[WQ];
      int nq = 0;
      auto flush = [&]() {
        unsigned long long mq[WQ];
#pragma unroll
        for (int u = 0; u < WQ; u++) mq[u] = u < nq ? D->masks[((size_t)(ch_q[u] / T) * N + rr) * T + (ch_q[u] % T)] : 0ull;
#pragma unroll
        for (int u = 0; u < WQ; u++) {
          unsigned long long m = mq[u];  // (0 beyond nq)
          const int chunk = ch_q[u];
          while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            list[cnt++] = (IdxT)(chunk * 64 + b);
          }
        }
        nq = 0;
      };
      for (int w0 = 0; w0 < rbw; w0 += 4) {
        const uint4 bits4 = *reinterpret_cast<const uint4*>(rb + w0);
        if ((bits4.x | bits4.y | bits4.z | bits4.w) == 0) continue;
        const unsigned bw[4] = {bits4.x, bits4.y, bits4.z, bits4.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          unsigned f = bw[q];
          while (f) {
            const int sl = (w0 + q) * 32 + __builtin_ctz(f);
            f &= f - 1;
            if (nq + T > WQ) flush();
            for (int t = 0; t < T; t++) {
#pragma unroll
              for (int u = 0; u < WQ; u++)
                if (u == nq) ch_q[u] = sl * T + t;  // (no run-time index into a register array)
              nq++;
            }
          }
        }
      }
      if (nq > 0) flush();
    }
  }
  if constexpr (sizeof(IdxT) == 2) {
    // ---- ascending original j (the order of the reference's first-K truncation and float accumulation), the list entry
    // being the target's sorted position: a sorting network in registers, sized by the longest row of the wave (uniform)
    const unsigned wmax = wave_max_u32((unsigned)cnt);
    const int* __restrict__ yorder = D->yorder;
    unsigned short* __restrict__ out = reinterpret_cast<unsigned short*>(D->cand_j);
    const unsigned short* l16 = reinterpret_cast<const unsigned short*>(list);
    if (wmax == 0u) {
    } else if (wmax <= 8u) {
      sort_and_store_list<8>(l16, cnt, yorder, out, N, pos);
    } else if (wmax <= 16u) {
      sort_and_store_list<16>(l16, cnt, yorder, out, N, pos);
    } else if (wmax <= 32u) {
      sort_and_store_list<32>(l16, cnt, yorder, out, N, pos);
    } else {
      sort_and_store_list<64>(l16, cnt, yorder, out, N, pos);
    }
  } else if (cnt > 0) {
    {
      {
        const int* yorder = D->yorder;
      // 32-bit positions (M >= 65536): rank sort.  Original indices first: independent gathers, LIST_RB in flight
      for (int k0 = 0; k0 < cnt; k0 += LIST_RB) {
        int jj[LIST_RB];
#pragma unroll
        for (int u = 0; u < LIST_RB; u++) jj[u] = yorder[(int)list[min(k0 + u, cnt - 1)]];
#pragma unroll
        for (int u = 0; u < LIST_RB; u++)
          if (k0 + u < cnt) list[k0 + u] = (IdxT)jj[u];
      }
      // ascending original j (the order of the reference's first-K truncation and float accumulation): every entry
      // is written straight to its rank (the indices of a row are distinct); cnt^2 independent LDS reads instead of
      // an insertion sort's chain of dependent shifts
      // the list entry is the target's sorted position (gathered while the rank is counted), the ORDER is that of the
      // original indices
      // (LIST_RB candidates per round: their position gathers are in flight together, and one pass over the list ranks all
      // of them - a dependent global load and a list sweep per CANDIDATE sat on every thread's serial chain before.  Packing
      // (original index, position) into one LDS word instead removes the gather altogether and is 2 % faster for a lone
      // pair, but the doubled LDS - 69 KB per block, two blocks per CU - costs the 64-pair batch 3 %: measured, not kept.)
      IdxT* out = reinterpret_cast<IdxT*>(D->cand_j);
      const int* yinv = D->yinv;
      for (int k0 = 0; k0 < cnt; k0 += LIST_RB) {
        int j[LIST_RB], entry[LIST_RB], rank[LIST_RB];
#pragma unroll
        for (int u = 0; u < LIST_RB; u++) {
          j[u] = (int)list[min(k0 + u, cnt - 1)];
          rank[u] = 0;
        }
#pragma unroll
        for (int u = 0; u < LIST_RB; u++) entry[u] = yinv[j[u]];
        for (int m2 = 0; m2 < cnt; m2++) {
          const int v = (int)list[m2];
#pragma unroll
          for (int u = 0; u < LIST_RB; u++) rank[u] += (v < j[u]) ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < LIST_RB; u++)
          if (k0 + u < cnt) out[(size_t)rank[u] * N + pos] = (IdxT)entry[u];
      }
      }
    }
  }
  // The block that finishes last validates the list: every block has read `rebuild` by then, and the
  // kernels of the iteration (stream order) see rebuild == 0 <=> bitmap, lists and overflow list are current.
  // Which positions overflow: one 64-bit word per wave, the only thing of this block another block of the launch
  // reads (the last one, below) - a coherent (sc1) store the wave waits for, then the gate.  (An agent-scope fence in
  // front of the gate - write back the XCD's L2 with a block's freshly written lists in it - cost 4 ms of the 74 ms
  // step: 3.5 us and more per block, four blocks per CU.)
  {
    const bool ov = rr < N && cnt_all > row_max;
    const unsigned long long m = __ballot(ov);
    // ... and which of them are beyond a long list as well (k_assoc_dense scans all targets for those)
    const unsigned long long m_scan = (Pp->long_lists && !D->st->all_dense) ? __ballot(ov && cnt_all > LONG_CAP) : m;
    // statistic: candidate pairs the association evaluates per iteration while these lists live (one returnless atomic
    // per wave and rebuild instead of a wave reduction in every wave of every k_assoc launch)
    const unsigned wsum = wave_sum_u32(rr < N ? (unsigned)min(cnt_all, 0x3ffffff) : 0u);
    {
      // the row-major runs of the overflow rows (PairDesc::dense_rel): min(candidates, K_max) entries each, in position order
      const int seg = ov ? min(cnt_all, Pp->K_max) : 0;
      const int incl = wave_prefix_incl_i32(seg, tid & 63);
      if (ov) D->dense_rel[pos] = incl - seg;
      if ((tid & 63) == 63) st_x<true>(D->ovf_wsum + (pos >> 6), incl);
    }
    if ((tid & 63) == 0) {
      st_x<true>(D->ovf_bits + (pos >> 6), m);
      if (m) atomicAdd(&D->st->n_ovf, __builtin_popcountll(m));
      if (m_scan) atomicAdd(&D->st->n_scan, __builtin_popcountll(m_scan));
      if (wsum) (void)__hip_atomic_fetch_add(&D->st->ncand_list, (unsigned long long)wsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (every wave drains its own stores, see flow_gate)
  __shared__ int s_last_block;
  __syncthreads();
  if (tid == 0) {
    const int done = atomicAdd(D->gate, 1);
    s_last_block = (done == nblk - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last_block) return;
  // Overflow list in ascending position order (dense regime: every row overflows, the list is the identity and
  // k_assoc_dense does not read it), from the waves' bit words: thread t takes word t of a 256-word chunk.
  const int n_ovf = __hip_atomic_load(&D->st->n_ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n_ovf > 0 && !D->st->all_dense) {
    __shared__ int s_wave_cnt[LIST_THREADS / 64];
    const int nwords = (N + 63) >> 6;
    int base = 0;
    long long run_base = 0;  // entries of the row-major runs so far (PairDesc::word_base)
    __shared__ int s_wave_run[LIST_THREADS / 64];
    for (int w0 = 0; w0 < nwords; w0 += LIST_THREADS) {
      const int wi = w0 + tid;
      unsigned long long bits = wi < nwords ? ld_x<true>(D->ovf_bits + wi) : 0ull;
      const int run = (wi < nwords && bits) ? ld_x<true>(D->ovf_wsum + wi) : 0;
      const int run_incl = wave_prefix_incl_i32(run, tid & 63);
      if ((tid & 63) == 63) s_wave_run[tid >> 6] = run_incl;
      const int mine = __builtin_popcountll(bits);
      // exclusive prefix of the popcounts over the chunk: inside the wave by a DPP-free shuffle scan, across waves via LDS
      int incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if ((tid & 63) >= o) incl += v;
      }
      if ((tid & 63) == 63) s_wave_cnt[tid >> 6] = incl;
      __syncthreads();
      int off = base + incl - mine, tot = 0;
      long long rb = run_base + run_incl - run;
      int run_tot = 0;
#pragma unroll
      for (int w = 0; w < LIST_THREADS / 64; w++) {
        off += (w < (tid >> 6)) ? s_wave_cnt[w] : 0;
        tot += s_wave_cnt[w];
        rb += (w < (tid >> 6)) ? s_wave_run[w] : 0;
        run_tot += s_wave_run[w];
      }
      if (wi < nwords) D->word_base[wi] = (int)(rb > 0x7fffffffll ? 0x7fffffffll : rb);
      run_base += run_tot;
      while (bits) {
        const int b = __builtin_ctzll(bits);
        bits &= bits - 1;
        D->ovf_rows[off++] = (wi << 6) + b;
      }
      base += tot;
      __syncthreads();
    }
    if (tid == 0) D->word_base[nwords] = (int)(run_base > 0x7fffffffll ? 0x7fffffffll : run_base);  // the total: all rows or none
  }
  if (tid == 0) {
    *D->gate = 0;
    D->st->rebuild = 0;
  }
}

}  // namespace cvo_dev
