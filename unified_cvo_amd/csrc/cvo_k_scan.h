// cvo_k_scan.h -- k_scan: the O(N x M) candidate scan of a rebuild (bitmap of the conservative cut-off test).
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_wave.h"

namespace cvo_dev {

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
// written by the previous kernel, k_prep, so it is read-only for the lifetime of k_scan).
constexpr int XCULL_PAD = 32;  // rows k_scan may read past N (whole groups + prefetch)

// v_writelane_b32 with compile-time lanes: moves wave-uniform values (SGPRs: the halves of ballot
// masks) into consecutive lanes of two VGPRs.  (The clang builtin is not declared for hipcc's host
// pass, hence inline asm.)  An asm statement is opaque to the hazard recogniser, and a v_cmp that
// has just written the SGPR must not be followed directly by the v_writelane that reads it (measured:
// stale masks without the wait), so every statement opens with its own s_nop.
template <int T, int BASE>
__device__ __forceinline__ void scatter_row_masks(const unsigned long long (&m)[T], unsigned& lo, unsigned& hi) {
  static_assert(T == 1 || T == 2 || T == 4 || T == 8, "T");
  if constexpr (T == 1) {
    asm("s_nop 4\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4"
        : "+v"(lo), "+v"(hi) : "s"((unsigned)m[0]), "s"((unsigned)(m[0] >> 32)), "n"(BASE));
  } else if constexpr (T == 2) {
    asm("s_nop 4\n\tv_writelane_b32 %0, %2, %6\n\tv_writelane_b32 %1, %3, %6\n\t"
        "v_writelane_b32 %0, %4, %7\n\tv_writelane_b32 %1, %5, %7"
        : "+v"(lo), "+v"(hi)
        : "s"((unsigned)m[0]), "s"((unsigned)(m[0] >> 32)), "s"((unsigned)m[1]), "s"((unsigned)(m[1] >> 32)),
          "n"(BASE), "n"(BASE + 1));
  } else {
    unsigned long long a[T / 2], b[T / 2];
#pragma unroll
    for (int t = 0; t < T / 2; t++) {
      a[t] = m[t];
      b[t] = m[T / 2 + t];
    }
    scatter_row_masks<T / 2, BASE>(a, lo, hi);
    scatter_row_masks<T / 2, BASE + T / 2>(b, lo, hi);
  }
}
template <int T, int U, int RG>
struct ScatterTile {
  static __device__ __forceinline__ void run(const unsigned long long (&mm)[RG][T], unsigned& lo, unsigned& hi) {
    scatter_row_masks<T, U * T>(mm[U], lo, hi);
    ScatterTile<T, U + 1, RG>::run(mm, lo, hi);
  }
};
template <int T, int RG>
struct ScatterTile<T, RG, RG> {
  static __device__ __forceinline__ void run(const unsigned long long (&)[RG][T], unsigned&, unsigned&) {}
};

constexpr int SCAN_TILE_CAP = 128;  // (row group, slice) tiles a wave queues in LDS per round

template <int T>
__global__ __launch_bounds__(256) void k_scan(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp,
                                              const PairState* __restrict__ states, int force) {
  constexpr int RG = ROWS_PER_GROUP;
  // per-wave tile queue: the row operands of every overlapping group, fetched by the lane that found it
  __shared__ f32x4 s_rows[4][SCAN_TILE_CAP][RG];
  __shared__ int s_tile_g[4][SCAN_TILE_CAP];
  // (the three rebuild kernels run as rebuild OPPORTUNITIES - every lean_U iterations in the lean graphs - and mostly
  // find nothing to do: what they branch on comes from the kernel-argument state array in ONE round of scalar loads,
  // not through status[] -> descriptor -> state pointer -> flag)
  {
    const PairState* __restrict__ st0 = states + blockIdx.z;  // == D->st
    const int status_v = st0->status, rebuild_v = st0->rebuild, dense_v = st0->all_dense;
    if (!force && (status_v != 0 || !rebuild_v || dense_v)) return;  // finished / the bitmap is still a superset / dense regime
  }
  const PairDesc* __restrict__ D = descs + blockIdx.z;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int slice = blockIdx.x * 4 + wave;
  const int nslices = D->nslices;
  if (slice >= nslices) return;
  // The gridDim.y blocks of a slice share its rows cell by cell (cell = 16 groups = 64 sorted rows): block k
  // owns cells k, k + S, k + 2S, ...  Interleaving matters: the cells a slice overlaps are neighbours in
  // the k-d order, so contiguous row segments would leave all the fine work of a slice to one wave.
  const int NCr = (D->NG + 15) >> 4;  // cells with real rows
  const int S = gridDim.y, kseg = blockIdx.y;
  if (kseg >= NCr) return;
  const int c_end = (NCr - kseg + S - 1) / S;  // this block's cells: i * S + kseg, i < c_end

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
  // bounding box of this wave's 64*T targets (wave-uniform -> scalar loads)
  const CVO_CONST f32x4* sb = (const CVO_CONST f32x4*)D->sbox + 2 * slice;
  const f32x4 smin = sb[0], smax = sb[1];
  const CVO_GLOBAL f32x4* cellbox = (const CVO_GLOBAL f32x4*)D->cellbox;
  const CVO_GLOBAL f32x4* gbox = (const CVO_GLOBAL f32x4*)D->gbox;
  const CVO_GLOBAL f32x4* xc = (const CVO_GLOBAL f32x4*)D->xcull;
  CVO_GLOBAL unsigned long long* masks = (CVO_GLOBAL unsigned long long*)D->masks;
  CVO_GLOBAL unsigned* rowbits = (CVO_GLOBAL unsigned*)D->rowbits;
  const int rbw = D->rbw;
  const unsigned slice_bit = 1u << (slice & 31);
  const int N = D->N;
  f32x4(*rows)[RG] = s_rows[wave];
  int* tile_g = s_tile_g[wave];
  // emission addresses: masks are [slice][row][T], so the RG*T words of a tile are one contiguous run
  // (lane q = u*T+t <-> row u, chunk t) and rows that are neighbours in space share cache lines
  CVO_GLOBAL unsigned long long* mask_lane = masks + (size_t)slice * N * T + lane;
  CVO_GLOBAL unsigned* rowbits_lane = rowbits + (size_t)(lane / T) * rbw + (slice >> 5);
  CVO_GLOBAL int* rowcnt_lane = (CVO_GLOBAL int*)D->row_cnt + (lane / T);

  // Two-level cull.  Level 1: lane l tests the box of row cell c (64 rows that the k-d ordering made a
  // compact block) against the slice box -> m1.  Level 2: four overlapping cells at a time, lane l tests
  // group (l & 15) of cell (l >> 4); the lane that finds an overlap fetches that group's RG row operands
  // straight into the wave's LDS tile queue.  Boxes are already grown by the cut-off radius; pad groups
  // and pad cells carry empty boxes.
  unsigned long long m1 = 0;
  int cb = 0;  // cell of bit 0 of m1
  int next_cb = 0;
  unsigned tiles_done = 0;
  for (;;) {
    int ntiles = 0;
    while (ntiles + 64 <= SCAN_TILE_CAP) {
      if (m1 == 0) {
        if (next_cb >= c_end) break;
        cb = next_cb;
        next_cb += 64;
        const int c = cb + lane;
        const int cc = min(c, c_end - 1);
        const int cell = cc * S + kseg;
        const f32x4 bmin = cellbox[2 * (size_t)cell], bmax = cellbox[2 * (size_t)cell + 1];
        const bool ov = (bmin.x <= smax.x) & (bmax.x >= smin.x) & (bmin.y <= smax.y) & (bmax.y >= smin.y) &
                        (bmin.z <= smax.z) & (bmax.z >= smin.z) & (c < c_end);
        m1 = __ballot(ov);
        continue;
      }
      const int s0 = __builtin_ctzll(m1);
      m1 &= m1 - 1;
      const int s1 = m1 ? __builtin_ctzll(m1) : -1;
      m1 &= m1 - 1;
      const int s2 = m1 ? __builtin_ctzll(m1) : -1;
      m1 &= m1 - 1;
      const int s3 = m1 ? __builtin_ctzll(m1) : -1;
      m1 &= m1 - 1;
      const int q = lane >> 4;
      const int sel = q == 0 ? s0 : (q == 1 ? s1 : (q == 2 ? s2 : s3));
      const int g = (((cb + max(sel, 0)) * S + kseg) << 4) + (lane & 15);
      const f32x4 bmin = gbox[2 * (size_t)g], bmax = gbox[2 * (size_t)g + 1];
      const bool overlap = (bmin.x <= smax.x) & (bmax.x >= smin.x) & (bmin.y <= smax.y) & (bmax.y >= smin.y) &
                           (bmin.z <= smax.z) & (bmax.z >= smin.z) & (sel >= 0);
      const unsigned long long m = __ballot(overlap);
      if (overlap) {
        const int slot = ntiles + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        const CVO_GLOBAL f32x4* xr = xc + (size_t)g * RG;
#pragma unroll
        for (int u = 0; u < RG; u++) rows[slot][u] = xr[u];
        tile_g[slot] = g;
      }
      ntiles += __builtin_popcountll(m);
    }
    if (ntiles == 0) break;  // the segment is exhausted
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- fine level over the queued tiles: 3 FMA per pair, a v_min3 tree per row, one compare per row
    const int nproc = (force & 4) ? 0 : ntiles;  // timing variants of cvo_debug_time_scan
    // software pipeline: the LDS reads of tile ti + 1 are in flight while tile ti is evaluated
    f32x4 nxt[RG];
#pragma unroll
    for (int u = 0; u < RG; u++) nxt[u] = rows[0][u];  // wave-uniform address: LDS broadcast
    int tg_nxt = tile_g[0];
    for (int ti = 0; ti < nproc; ti++) {
      f32x4 cur[RG];
#pragma unroll
      for (int u = 0; u < RG; u++) cur[u] = nxt[u];
      const int tg = tg_nxt;
      {
        const int tn = min(ti + 1, nproc - 1);
#pragma unroll
        for (int u = 0; u < RG; u++) nxt[u] = rows[tn][u];
        tg_nxt = tile_g[tn];
      }
      float acc[RG][T];
      unsigned long long mu[RG];
      unsigned long long any = 0;
#pragma unroll
      for (int u = 0; u < RG; u++) {
#pragma unroll
        for (int t = 0; t < T; t++) {
          float a = __builtin_fmaf(y1[t], cur[u].x, yy[t]);
          a = __builtin_fmaf(y2[t], cur[u].y, a);
          acc[u][t] = __builtin_fmaf(y3[t], cur[u].z, a);
        }
        float mn = acc[u][0];
#pragma unroll
        for (int t = 1; t < T; t++) mn = __builtin_fminf(mn, acc[u][t]);
        mu[u] = __ballot(mn < cur[u].w);
        any |= mu[u];
      }
      if (any && !(force & 2)) {  // usual case once tiles are culled: the group has candidates among this wave's 64*T targets
        // Lane q = u*T+t receives the bitmap word of (row u, chunk t) with v_writelane; the T lanes of a row
        // the whole tile is emitted with one (contiguous) mask
        // store and one returnless atomic instruction.
        const int r = __builtin_amdgcn_readfirstlane(tg) * RG;
        unsigned long long mm[RG][T];
#pragma unroll
        for (int u = 0; u < RG; u++) {
#pragma unroll
          for (int t = 0; t < T; t++) mm[u][t] = __ballot(acc[u][t] < cur[u].w);
        }
        unsigned lo = 0, hi = 0;
        ScatterTile<T, 0, RG>::run(mm, lo, hi);
        // lanes u*T .. u*T+T-1 of every row u that has a candidate in this slice (wave-uniform mask: no
        // cross-lane traffic); all T words of such a row are stored
        unsigned rowsel = 0;
#pragma unroll
        for (int u = 0; u < RG; u++) rowsel |= mu[u] ? (((1u << T) - 1u) << (u * T)) : 0u;
        // candidates of each row in this slice: scalar popcounts of the ballot masks, handed to the row's first lane
        int row_pc = 0;
#pragma unroll
        for (int u = 0; u < RG; u++) {
          int c = 0;
#pragma unroll
          for (int t = 0; t < T; t++) c += __builtin_popcountll(mm[u][t]);
          row_pc = (lane == u * T) ? c : row_pc;
        }
        if (lane < RG * T && ((rowsel >> lane) & 1u)) {
          mask_lane[(size_t)r * T] = ((unsigned long long)hi << 32) | lo;
          if ((lane % T) == 0) {
            // tells k_list that this (row, slice) has valid mask words, and how many candidates they add to the row
            __hip_atomic_fetch_or(rowbits_lane + (size_t)r * rbw, slice_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(rowcnt_lane + r, row_pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    tiles_done += (unsigned)ntiles;
    __builtin_amdgcn_wave_barrier();  // the queue is reused by the next round
  }
  if (lane == 0 && tiles_done)  // statistics only (cvo_debug_scan_stats): one returnless atomic per wave
    __hip_atomic_fetch_add(D->tile_count, (unsigned long long)tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace cvo_dev
