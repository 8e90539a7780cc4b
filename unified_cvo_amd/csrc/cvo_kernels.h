// cvo_kernels.h -- the hand-written gfx950 kernels of the pairwise align() hot path.
//
// Every iteration of every in-flight frame pair (the pair is part of the grid):
//
//   k_assoc  one thread per source row walks the row's CACHED candidate list in ascending ORIGINAL j and runs the
//            reference's exact per-pair arithmetic (double exp, colour / semantic kernels, a > sp_thres, first-K
//            truncation; fill_in_A_mat_gpu, CvoGPU.cu:477-593), writes the ELL matrix and accumulates the
//            per-row flow (compute_flow_gpu_no_eigen, CvoGPU.cu:729-790).
//   k_coeff  reduces the flow partials to the normalised twist, then one thread per row accumulates B,C,D,E
//            (compute_step_size_xi + _poly_coeff, CvoGPU.cu:953-1082); the block of the pair that finishes last
//            runs the reference's host-side scalar code (cubic, Exp, pose update, SE(3) log, indicator, ell decay,
//            K update; CvoGPU.cu:1122-1158, 1452-1531) and decides whether the candidate lists are still valid.
//
// Only when a pair's lists have expired (the targets moved further than the skin the scan added to every cut-off):
//
//   k_prep   update_tf + transform_pointcloud_thrust + per-row cut-offs as cull operands and bounding boxes.
//   k_scan   N x M candidate scan in spatially sorted index space: the O(N*M) part of fill_in_A_mat_gpu.  Lanes
//            hold targets, row operands are broadcast from an LDS tile queue; a two-level bounding-box cull leaves
//            a few percent of the (4 rows x 128 targets) tiles; 3 FMAs per pair, and the lane mask of the
//            compare IS the candidate bitmap word.
//   k_list   per-row sorted candidate lists from the bitmap, rows re-ordered by candidate count inside 256-row
//            windows (load balance of the thread-per-row kernels).
//   k_assoc_dense  rows with more candidates than a list holds: the reference's literal ordered scan, a wave per
//            row (full graph only).
//
// No host round trip happens inside the loop; finished pairs early-exit on their status word.
#pragma once
#include "cvo_device.h"

// These kernels are written for gfx950 only: v_permlane16_swap / v_permlane32_swap (xor16_sum, xor32_sum), the
// XCC_ID hardware register, kernel-argument preloading, and - in k_assoc / k_coeff - waves that RETURN while the first
// wave of their block goes on to further __syncthreads(): on gfx9 an s_barrier counts only the waves of the workgroup
// that are still alive (a terminated wave is taken out of the barrier's count), which is outside what the HIP
// programming model promises.  The device pass refuses any other target instead of miscompiling quietly.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "unified_cvo_amd kernels target gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

// min. waves per SIMD the register allocator is asked to leave room for (__launch_bounds__ second argument)
#ifndef CVO_COEFF_WAVES
#define CVO_COEFF_WAVES 1
#endif
// k_assoc, geometry-only instantiation: 6 (68 VGPRs, no spills; the GENERAL one needs 121 and would spill: it stays at 1; with its 24.6 KB of LDS per block six blocks = six waves per SIMD fit a CU anyway).
// The batch is throughput-bound above ~64 pairs (scripts/scale_probe.py): 62.2 vs 63.0 ms per step at 64 pairs, 111.9
// vs 114.1 at 128.  k_coeff stays at 1: its register count comes from the update it carries in its last block (93), and
// asking for 6 / 8 waves spills 68 / 140 bytes there (64.0 / 67.2 ms).
#ifndef CVO_ASSOC_WAVES
#define CVO_ASSOC_WAVES 6
#endif

namespace cvo_dev {

// Wave-wide reductions on the DPP cross-lane paths (no LDS traffic, unlike ds_bpermute shuffles): a butterfly
// inside every row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), then the four row results
// through scalar registers.  Every lane takes part and every lane gets the result; the order of the
// additions is fixed.
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = dpp_i32<CTRL>(__double2loint(v)), hi = dpp_i32<CTRL>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// own + partner across rows of 16 lanes (lane ^ 16, lane ^ 32) on gfx950's v_permlane16_swap / v_permlane32_swap:
// with both operands = v the two results are {own, partner} in an order that depends on the lane's half - the sum
// does not (IEEE addition commutes), so this is bit for bit `v + __shfl_xor(v, 16 / 32)` without the two
// ds_bpermute round trips through the LDS (scripts/ubench/permlane_swap.hip prints what the instructions return).
__device__ __forceinline__ double xor16_sum(double v) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double xor32_sum(double v) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double lane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
  const unsigned lo = (unsigned)dpp_i32<CTRL>((int)(unsigned)v), hi = (unsigned)dpp_i32<CTRL>((int)(unsigned)(v >> 32));
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long lane_u64(unsigned long long v, int l) {
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32) |
         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<DPP_XOR1>(v);
  v += dpp_f64<DPP_XOR2>(v);
  v += dpp_f64<DPP_HALF_MIRROR>(v);
  v += dpp_f64<DPP_MIRROR>(v);
  return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {  // (the caller's sum fits 32 bits)
  v += (unsigned)dpp_i32<DPP_XOR1>((int)v);
  v += (unsigned)dpp_i32<DPP_XOR2>((int)v);
  v += (unsigned)dpp_i32<DPP_HALF_MIRROR>((int)v);
  v += (unsigned)dpp_i32<DPP_MIRROR>((int)v);
  return (unsigned)(__builtin_amdgcn_readlane((int)v, 0) + __builtin_amdgcn_readlane((int)v, 16) +
                    __builtin_amdgcn_readlane((int)v, 32) + __builtin_amdgcn_readlane((int)v, 48));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)dpp_i32<DPP_XOR1>((int)v));
  v = max(v, (unsigned)dpp_i32<DPP_XOR2>((int)v));
  v = max(v, (unsigned)dpp_i32<DPP_HALF_MIRROR>((int)v));
  v = max(v, (unsigned)dpp_i32<DPP_MIRROR>((int)v));
  return max(max((unsigned)__builtin_amdgcn_readlane((int)v, 0), (unsigned)__builtin_amdgcn_readlane((int)v, 16)),
             max((unsigned)__builtin_amdgcn_readlane((int)v, 32), (unsigned)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o));
  return v;
}

// XCD-aware placement of the row-block kernels (k_list, k_assoc, k_coeff): the dispatcher is observed to place
// workgroup b on XCD b % 8, so a 1-D grid is decoded as pair = (b / 8 / nblk) * 8 + b % 8, row block = (b / 8) % nblk:
// all blocks of a pair run on one XCD and its lists, targets and ELL rows stay in that XCD's 4 MB L2 across
// kernels and iterations (with the default mapping every XCD touches every pair: ~8x the L2 footprint).  Purely a
// speed choice: nothing depends on where a block actually runs.  Grid = nblk * round_up(n_pairs, 8).
struct PairBlock {
  int pair, bx;
};
__device__ __forceinline__ bool pair_block(int nblk, int n_pairs, PairBlock& pb) {
  const int b = (int)blockIdx.x;
  const int slot = b >> 3;
  const int grp = slot / nblk;
  pb.pair = grp * 8 + (b & 7);
  pb.bx = slot - grp * nblk;
  return pb.pair < n_pairs;
}

// Values exchanged between the blocks of one launch (k_coeff's partials -> its last block): on this multi-die part the L2 of an XCD is not
// coherent with the others inside a kernel, and agent-scope fences write back / invalidate whole caches.  Relaxed
// agent-scope atomics carry the coherence bits on the instruction itself, which is all a handful of partial
// sums needs.  COH = false: plain accesses (the producer is an earlier kernel).
// Address-space qualified views: pointers read out of a PairDesc are generic ("flat") to the compiler.  A flat load
// counts against vmcnt AND lgkmcnt and the compiler waits for both counters to reach zero before it uses one: a loop
// that prefetches (k_assoc's candidates, k_coeff's entries) or a tail that has several groups of loads in flight then
// serialises on every use.  Re-qualified as global, the same loads are global_load with exact vmcnt(n) waits.
#define CVO_GLOBAL __attribute__((address_space(1)))
#define CVO_CONST __attribute__((address_space(4)))
typedef float f32x4 __attribute__((ext_vector_type(4)));  // plain vector: loadable from any address space
// (float4 is a class type: its copy constructor only takes generic references)
__device__ __forceinline__ float4 ldg_f4(const CVO_GLOBAL f32x4* p) {
  const f32x4 v = *p;
  return make_float4(v.x, v.y, v.z, v.w);
}
// ... and only its xyz: a 16-byte load whose fourth register is dead gets that register handed to the next load the loop
// issues, which then has to wait for this one (k_assoc's candidate prefetch was serialised that way)
typedef float f32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ float4 ldg_xyz(const CVO_GLOBAL f32x4* p) {
  const f32x3 v = *reinterpret_cast<const CVO_GLOBAL f32x3*>(p);
  return make_float4(v.x, v.y, v.z, 0.f);
}
template <typename T>
__device__ __forceinline__ const CVO_GLOBAL T* as_global(const T* p) {
  return (const CVO_GLOBAL T*)p;
}
template <bool COH, typename T>
__device__ __forceinline__ T ld_g(const CVO_GLOBAL T* p) {
  if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}
template <bool COH, typename T>
__device__ __forceinline__ void st_x(T* p, T v) {
  if (COH)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    *p = v;
}
template <bool COH, typename T>
__device__ __forceinline__ T ld_x(const T* p) {
  if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
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

// ------------------------------------------------------------------------------------------
// Exact per-pair arithmetic of fill_in_A_mat_gpu (CvoGPU.cu:528-573).
// ------------------------------------------------------------------------------------------
struct RowData {
  float x, y, z, l, d2_thres;
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
  return RowData{x.x, x.y, x.z, l, thr, den, rcp_refined(den)};
}
// The colour and semantic kernels' exponent denominators (2.0 * c_ell^2, 2.0 * s_ell^2: the same for every pair of a
// call) with their refined reciprocals; evaluated once per thread, outside the row loops.
struct FeatDen {
  double c_den, c_rcp, s_den, s_rcp;
  ExpConsts ek;  // (rides along: every evaluation of a pair needs it)
};
__device__ __forceinline__ FeatDen make_feat_den(const DevParams& P) {
  FeatDen f;
  f.c_den = 2.0 * P.c2;
  f.c_rcp = rcp_refined(f.c_den);
  f.s_den = P.mode == 2 ? 2.0 * P.s_ell_sq : 2.0 * P.s_ell * P.s_ell;
  f.s_rcp = rcp_refined(f.s_den);
  f.ek = make_exp_consts();
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

// GENERAL = false is the geometry-only specialisation (no colour / semantic / geometric-type code at all:
// 1/3 fewer VGPRs, one more wave per SIMD for the latency-bound association kernel).
// i / j index the FEATURE arrays (colour, class distributions, geometric types), which clouds keep in spatial order:
// i = the row's sorted position, j = the target's sorted position.
// The pair arithmetic for an already transformed target yt (everything of CvoGPU.cu:528-573 but the transform).
template <bool GENERAL>
__device__ __forceinline__ bool eval_pair_yt(const DevParams& P, const PairDesc* __restrict__ D, const FeatDen& F, int i,
                                             const RowData& r, int j, const float4 yt, float& a_out) {
  float sk = 1, ck = 1, k = 1, geo_sim = 1;
  if (GENERAL && P.use_geotype) {  // compute_geometric_type_ip, CvoGPU.cu:203-215
    const float2 ga = D->xgeo[i], gb = D->ygeo[j];
    const float n2a = __builtin_fmaf(ga.y, ga.y, ga.x * ga.x);
    const float n2b = __builtin_fmaf(gb.y, gb.y, gb.x * gb.x);
    const float dab = __builtin_fmaf(ga.y, gb.y, ga.x * gb.x);
    geo_sim = dab * dab / (n2a * n2b);
    if ((double)geo_sim < 0.01) return false;
  }
  if (GENERAL && P.use_geo && P.mode == 2) {  // (the host launches the GENERAL instantiations for mode 2)
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
  if (GENERAL && P.use_sem) {
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
template <bool GENERAL>
__device__ __forceinline__ bool eval_pair(const DevParams& P, const PairDesc* __restrict__ D, const FeatDen& F, const Pose& pose,
                                          int i, const RowData& r, int j, const float4 y0, float& a_out, float4& yt_out) {
  const V3 ytv = transform_point(pose.Ri, pose.Ti, y0.x, y0.y, y0.z);
  const float4 yt = make_float4(ytv.x, ytv.y, ytv.z, 0.f);
  yt_out = yt;
  return eval_pair_yt<GENERAL>(P, D, F, i, r, j, yt, a_out);
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
  float4* stage = nullptr;   // this thread's column of the block's LDS staging area (AssocShared::stage)
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
template <bool GENERAL>
__device__ __forceinline__ void visit_pair_yt(const DevParams& P, const PairDesc* __restrict__ D, const FeatDen& F, int i, int pos,
                                              int N, const RowData& r, const V3& pxe, int j, const float4 yt, RowAcc& A) {
  float a;
  if (!eval_pair_yt<GENERAL>(P, D, F, i, r, j, yt, a)) return;
  if (a > P.sp_thres) {
    // The row's first ELL_STAGE nonzeros are parked in the thread's own LDS column and leave after the loop as
    // write-through stores (assoc_phase); only rows longer than that store from inside the loop.
    if (A.nnz < (unsigned)ELL_STAGE)
      A.stage[A.nnz * ASSOC_THREADS] = make_float4(a, yt.x, yt.y, yt.z);
    else
      *A.slot = EllEntry{a, yt.x, yt.y, yt.z};
#ifdef CVO_EXP_DOUBLE_ELL
    if (A.nnz < 64u) reinterpret_cast<EllEntry*>(D->ell_j)[(size_t)A.nnz * N + pos] = EllEntry{a, yt.x, yt.y, yt.z};
#endif
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
    A.asum += (double)a;
  }
}

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
  if (rr < N) {  // real rows occupy the positions below N
    const int* yorder = D->yorder;
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
      int cnt = 0;  // sorted-space positions of the candidates (the mask words come from L1/L2 this time)
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
            const unsigned long long* mw = D->masks + ((size_t)sl * N + rr) * T;
            for (int t = 0; t < T; t++) {
              unsigned long long m = mw[t];
              const int chunk = sl * T + t;
              while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                list[cnt++] = (IdxT)(chunk * 64 + b);
              }
            }
          }
        }
      }
      // original indices: independent gathers, four in flight
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
    for (int w0 = 0; w0 < nwords; w0 += LIST_THREADS) {
      const int wi = w0 + tid;
      unsigned long long bits = wi < nwords ? ld_x<true>(D->ovf_bits + wi) : 0ull;
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
#pragma unroll
      for (int w = 0; w < LIST_THREADS / 64; w++) {
        off += (w < (tid >> 6)) ? s_wave_cnt[w] : 0;
        tot += s_wave_cnt[w];
      }
      while (bits) {
        const int b = __builtin_ctzll(bits);
        bits &= bits - 1;
        D->ovf_rows[off++] = (wi << 6) + b;
      }
      base += tot;
      __syncthreads();
    }
  }
  if (tid == 0) {
    *D->gate = 0;
    D->st->rebuild = 0;
  }
}

// thrust::reduce of omega_gpu / v_gpu (CvoGPU.cu:824-825) from the association block partials, Eigen's
// normalize() and the matrices of compute_step_size_xi: once per pair and iteration, by one wave of the block of
// the association launch that stores its partial last (k_assoc in the lean graph, k_assoc_dense in the full one).
// Lane l owns component (l & 7) of blocks l>>3, l>>3 + 8, ... (independent loads, all in flight), the eight
// groups meet through DPP / ds_swizzle; the order of the additions is fixed.  k_coeff reads the 42 floats with
// scalar loads in its first burst (they used to be reduced again by every one of its blocks: ~3 us of
// dependent round trips in front of each row loop and a hot spot of 150 readers per cache line).
static_assert(sizeof(XiMats) <= 48 * sizeof(float), "PairState::xi holds an XiMats");
template <bool COH>
__device__ __forceinline__ double coeff_twist_load(const PairDesc* __restrict__ D, int nparts) {
  const int lane = threadIdx.x & 63;
  const double* __restrict__ src = D->flow_part + (lane & 7);
  double acc = 0;
  // sixteen loads per round, all issued before the first addition (79 row blocks = one round); slots past the
  // end re-read the last one and add zero
  for (int b = lane >> 3; b < nparts; b += 128) {
    double p[16];
#pragma unroll
    for (int u = 0; u < 16; u++) p[u] = ld_x<COH>(src + (size_t)min(b + 8 * u, nparts - 1) * 8);
#pragma unroll
    for (int u = 0; u < 16; u++) acc += (b + 8 * u < nparts) ? p[u] : 0.0;
  }
  return acc;
}
// GRAN: resident kernel - the 42 floats leave as data-tagged granules (ResidentSync::xi) instead of PairState::xi.
template <bool GRAN = false>
__device__ __forceinline__ void twist_finalize(const PairDesc* __restrict__ D, int nparts, unsigned long long* gran = nullptr,
                                               unsigned tag = 0u, float* twist_out = nullptr) {
  double acc = coeff_twist_load<true>(D, nparts);
  acc += dpp_f64<0x128>(acc);  // row_ror:8 : groups g and g ^ 1
  acc = xor16_sum(acc);
  acc = xor32_sum(acc);
  // every lane now holds the total of component (lane & 7): lane q converts / divides ITS component, so the six
  // IEEE divisions of the normalisation are one (this wave is the serial tail of the pair's iteration)
  float own = (float)acc;
  float ov[6];
#pragma unroll
  for (int q = 0; q < 6; q++) ov[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, own), q));
  float z = 0;  // Eigen normalize(): z = squaredNorm(); if (z > 0) *this /= sqrt(z)
#pragma unroll
  for (int q = 0; q < 6; q++) z = z + ov[q] * ov[q];
  if (z > 0) {
    const float sq = sqrtf(z);
    own = own / sq;
#pragma unroll
    for (int q = 0; q < 6; q++) ov[q] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, own), q));
  }
  XiMats M;
  xi_mats(ov, ov + 3, M);
  if (twist_out) {
#pragma unroll
    for (int q = 0; q < 6; q++) twist_out[q] = ov[q];
  }
  if ((threadIdx.x & 63) == 0) {
    const float* mv = reinterpret_cast<const float*>(&M);
    if (GRAN) {
#pragma unroll
      for (int q = 0; q < (int)(sizeof(XiMats) / sizeof(float)); q++)
        gran[q] = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(mv[q]);
    } else {
      float* dst = D->st->xi;
#pragma unroll
      for (int q = 0; q < (int)(sizeof(XiMats) / sizeof(float)); q++) dst[q] = mv[q];
    }
  }
}
// The flow partial of this block is stored; the block that finds it was the last one of its pair reduces them.
// Every thread of the block calls this.
__device__ __forceinline__ bool flow_gate(const PairDesc* __restrict__ D, int nblocks, int nparts) {
  __shared__ int s_flow_last;
  // this block's partial must have reached the L2 before its counter increment can be seen: the barrier alone only
  // orders LDS traffic (the compiler emits no vmcnt wait for it), and a store and an atomic of one wave to
  // different addresses are not ordered on their way to memory
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int done = __hip_atomic_fetch_add(D->gate_flow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_flow_last = (done == nblocks - 1) ? 1 : 0;
    if (done == nblocks - 1) __hip_atomic_store(D->gate_flow, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_flow_last && threadIdx.x < 64) twist_finalize(D, nparts);
  return s_flow_last != 0;
}

// ------------------------------------------------------------------------------------------
// Association phase: ordered association + flow, one thread per (sorted) source row, over the cached
// candidate list.
// ------------------------------------------------------------------------------------------
// CVO_PHASE_TICKS=1: thread 0 of every block of k_assoc [0] / k_coeff [1] leaves four s_memtime stamps (entry, row loop
// start, row loop end, exit; for k_coeff: entry, rows start, rows end, counter), and the updating block of pair p its
// entry / counter / exit at [1][4096 + p].  Only differences inside a block mean anything (the counters of
// different XCDs are not aligned).  Printed by cvo_debug_time_kernels.
__device__ unsigned long long g_phase_ticks[2][8192][4];
// ... and inside the update (the last pair to get there wins; meant for one pair in flight): entry, partials reduced, step
// chosen, pose / distance / indicator done (update_tf next), list bookkeeping done, state written back
__device__ unsigned long long g_upd_ticks[8];
#define CVO_UPD_STAMP(i) do { if (P.phase_ticks && threadIdx.x == 0) g_upd_ticks[i] = __builtin_readcyclecounter(); } while (0)

// CVO_KERNEL_CLOCK (see PairState::clk_*): the first block of pair p stamps its entry (blocks are dispatched in
// order, so it is the pair's earliest or close to it; an atomic minimum over all blocks would serialise 79 atomics per
// pair on one address), the block that finishes the pair's work in the launch (flow gate / update) closes the interval.
__device__ __forceinline__ void pair_clock_begin(bool on, PairState* st, int which) {
  if (on && threadIdx.x == 0) st_x<true>(&st->clk_start[which], (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
// t0: the stamp, read (coherently) by the caller before it waited for its last-block counter - the pair's first block
// is long past its entry by then, and the load stays off the serial tail
__device__ __forceinline__ unsigned long long pair_clock_peek(bool on, const PairState* st, int which) {
  return (on && threadIdx.x == 0) ? ld_x<true>(&st->clk_start[which]) : 0ull;
}
__device__ __forceinline__ unsigned pair_clock_ticks(unsigned long long t0) {
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  // (a stamp left over from an earlier launch - the first block of this one has not run yet - would show up as an
  // interval of many milliseconds: dropped)
  return (t0 != 0ull && t0 <= t1 && t1 - t0 < 400000ull) ? (unsigned)(t1 - t0) : 0u;
}
// What a row needs first, requested from kernel-argument addresses (see row_off_* in cvo_device.h) before the
// descriptor has arrived.
struct AssocRowHead {
  int cnt, ip, j1;
  float4 x;
};
// Block reduction of NC doubles per thread through LDS: every thread deposits its values (column-major: conflict-free
// 8-byte writes), then lane (c, g) of the first wave - eight lanes per component - adds the values of threads
// g, g + 8, g + 16, ... in that order and the eight partial sums meet through a 3-step DPP butterfly; lanes with g == 0
// return the total of component c = lane / 8 (valid for lane < 8 * NC).  ~20 wave-instructions per wave instead of ~27
// per COMPONENT for a DPP / readlane reduction of doubles (the epilogues were a third of k_assoc's instructions).
// The order of the additions is fixed.  Contains a __syncthreads().
template <int NC>
struct BlockRedShared {
  double v[NC][ASSOC_THREADS + 8];  // (+ 8: components land on different banks)
};
template <int NC>
__device__ __forceinline__ double block_reduce_lds(BlockRedShared<NC>& S, const double (&x)[NC]) {
  static_assert(NC <= 8, "eight lanes per component in one wave");
#pragma unroll
  for (int c = 0; c < NC; c++) S.v[c][threadIdx.x] = x[c];
  __syncthreads();
  double t = 0;
  if (threadIdx.x < 8 * NC) {
    const int c = threadIdx.x >> 3, g = threadIdx.x & 7;
    double p[ASSOC_THREADS / 8];
#pragma unroll
    for (int k = 0; k < ASSOC_THREADS / 8; k++) p[k] = S.v[c][8 * k + g];
#pragma unroll
    for (int k = 0; k < ASSOC_THREADS / 8; k++) t += p[k];
  }
  if (threadIdx.x < 64) {  // (whole wave: the DPP steps need their partner lanes active)
    t += dpp_f64<DPP_XOR1>(t);
    t += dpp_f64<DPP_XOR2>(t);
    t += dpp_f64<DPP_HALF_MIRROR>(t);
  }
  return t;
}

struct AssocShared {
  union {
    BlockRedShared<7> red;                   // after the row loop
    float4 stage[ELL_STAGE][ASSOC_THREADS];  // during it: the rows' first ELL entries, one column per thread
  };
  unsigned long long cnt[ASSOC_THREADS / 64][4];
};

// LOCAL: the block partials are read by a block of the SAME XCD (resident kernel): plain stores keep the line in that
// XCD's L2, where the reader's L1-bypassing loads find it; otherwise coherent (sc1, write-through) stores.
template <typename IdxT, int ASSOC_CAP, bool GENERAL, bool INSTR, bool LOCAL = false>
__device__ __forceinline__ void assoc_phase(const DevParams& P, const PairDesc* __restrict__ D, const IterView& iv,
                                            AssocShared& S, const int bx, const AssocRowHead& head) {
  const int N = D->N;
  const int pos = bx * ASSOC_THREADS + threadIdx.x;  // position in k_list's count-ordered row windows
  const int K = iv.K;
  RowAcc A;
  A.slot = D->ell + pos;
  A.stage = &S.stage[0][threadIdx.x];
  unsigned overflowed = 0;
  unsigned long long tt1 = 0, tt2 = 0;
  if (pos < N) {
    const int j1s = head.j1;  // (the first list slot exists whatever the count is)
    const int j2s = (int)(reinterpret_cast<const IdxT*>(D->cand_j) + pos)[N];
    const int cnt = head.cnt;
    overflowed = cnt > min(iv.row_max, ASSOC_CAP) ? 1u : 0u;
    if (!overflowed) {
      const int i = head.ip;
      const float4 x = head.x;
      const RowData r = make_row(P, x, iv.ell);
      const FeatDen F = make_feat_den(P);
      const V3 pxe{x.x, x.y, x.z};
      const Pose& pose = iv.pose;
      const CVO_GLOBAL IdxT* cj = as_global(reinterpret_cast<const IdxT*>(D->cand_j)) + pos;
      // list entries are sorted positions: coordinates (and features) come from the spatially ordered arrays of the
      // target cloud - the candidates of the 64 neighbouring rows of a wave fall into a few cache lines instead of 64
      const CVO_GLOBAL f32x4* ysrc = (const CVO_GLOBAL f32x4*)D->ys4;
      // exact evaluation in ascending original j; index and coordinates of the next candidates are in
      // flight while the current one is evaluated
      int j1 = cnt > 0 ? j1s : 0;
      int j2 = cnt > 1 ? j2s : 0;
      if (INSTR) tt1 = __builtin_readcyclecounter();
      float4 y1 = ldg_xyz(ysrc + j1);
      for (int k = 0; k < cnt && A.nnz < (unsigned)K; k++) {
        const int j = j1;
        const float4 ycur = y1;
        j1 = j2;
        if (k + 1 < cnt) y1 = ldg_xyz(ysrc + j1);
        if (k + 2 < cnt) j2 = (int)cj[(size_t)(k + 2) * N];
        // (Tried in round 3: a first pass that only transforms and tests the distance, parking what passes in LDS, and
        // the kernel values in a second pass over the parked entries - bit-identical, -8 % for a lone pair's resident
        // iteration, +5 % for the 64-pair batch: most waves hold rows of one to three candidates, where the exp already
        // runs once or twice per wave either way, and the second loop and its LDS traffic are pure overhead.)
        const V3 ytv = transform_point(pose.Ri, pose.Ti, ycur.x, ycur.y, ycur.z);
        visit_pair_yt<GENERAL>(P, D, F, i, pos, N, r, pxe, j, make_float4(ytv.x, ytv.y, ytv.z, 0.f), A);
      }
      D->nnz_row[pos] = A.nnz;
      {
        // The ELL entries leave now, back to back, as WRITE-THROUGH (sc1) 16-byte stores.  A dependent kernel boundary
        // costs its ~1.5 us plus (bytes the predecessor left dirty in the XCDs' L2s) / 6 TB/s (MI355X_MICROARCH.md): the
        // 5.7 MB of ELL entries a 16-pair launch used to leave behind as plain stores put ~0.9 us in front of every
        // k_coeff; written through they drain while the other waves still work, and the row loop's wait for its
        // prefetched loads (vmcnt(0): flat addresses) no longer includes a store acknowledgement.  Measured with
        // scripts/exp_time.py: 66.2 -> 63.6 ms per step (4 slots; 62.9 with 6); a plain second copy of every entry (twice the dirty bytes)
        // costs 14 ms, sc1 stores from inside the loop 2 ms (profiles/r4/ell_store_experiments.txt).
        const unsigned ns = min(A.nnz, (unsigned)ELL_STAGE);
        EllEntry* dst = D->ell + pos;
        for (unsigned q = 0; q < ns; q++) {
          const float4 e4 = A.stage[q * ASSOC_THREADS];  // (own LDS column: no barrier)
          const f32x4 ev = {e4.x, e4.y, e4.z, e4.w};
          asm volatile("flat_store_dwordx4 %0, %1 sc1" ::"v"(dst), "v"(ev) : "memory");
          dst += N;
        }
      }
      if (INSTR) tt2 = __builtin_readcyclecounter();
    }
  }
  if (INSTR && P.phase_ticks && threadIdx.x == 0) {
    g_phase_ticks[0][blockIdx.x & 8191][1] = tt1;
    g_phase_ticks[0][blockIdx.x & 8191][2] = tt2;
  }
  // per-row (omega_i / c, v_i / d) cast to double, then reduced in double (CvoGPU.cu:784-787, 824-825)
  // (six IEEE float divisions per row by two call-wide constants: 72 of a wave's ~500 VALU instructions as the compiler
  // expands them, 30 + 16 with the denominators' halves hoisted and the operand check that licenses it)
  float fq[6];
  {
    const float fn[6] = {A.o0, A.o1, A.o2, A.v0, A.v1, A.v2};
    const bool safe = P.fast_div_cd != 0 && fdiv_operands_safe(fn);
    if (__ballot(!safe) == 0ull) {
      const FDivU uc = fdiv_prepare(P.c), ud = fdiv_prepare(P.d);
#pragma unroll
      for (int q = 0; q < 3; q++) {
        fq[q] = fdiv_hoisted(fn[q], uc);
        fq[3 + q] = fdiv_hoisted(fn[3 + q], ud);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 3; q++) {
        fq[q] = fn[q] / P.c;
        fq[3 + q] = fn[3 + q] / P.d;
      }
    }
  }
  const double red[7] = {(double)fq[0], (double)fq[1], (double)fq[2], (double)fq[3], (double)fq[4], (double)fq[5], A.asum};
  constexpr int NW = ASSOC_THREADS / 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long nn = wave_sum_u32(A.nnz);  // 64 rows x K_max
  const unsigned mx = wave_max_u32(A.nnz);
  const unsigned long long nov = (unsigned long long)__builtin_popcountll(__ballot(overflowed != 0));
  if (lane == 0) {
    S.cnt[wave][0] = nn;
    S.cnt[wave][1] = mx;
    S.cnt[wave][2] = 0ull;  // (the candidate statistic is a property of the lists: k_list leaves it in PairState::ncand_list)
    S.cnt[wave][3] = nov;
  }
  __syncthreads();  // (the staging columns share their LDS with the reduction: every thread has drained its own)
  const double tot = block_reduce_lds<7>(S.red, red);  // (its barrier also covers S.cnt)
  if (threadIdx.x < 56 && (threadIdx.x & 7) == 0) {
    st_x<!LOCAL>(D->flow_part + (size_t)bx * 8 + (threadIdx.x >> 3), tot);  // read by another block of this launch (flow_gate)
  } else if (threadIdx.x == 57) {
    unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      a0 += S.cnt[w][0];
      a1 = max(a1, S.cnt[w][1]);
      a2 += S.cnt[w][2];
      a3 += S.cnt[w][3];
    }
    unsigned long long* cp = D->cnt_part + (size_t)bx * 4;
    cp[0] = a0;
    cp[1] = a1;
    cp[2] = a2;
    cp[3] = a3;
  }
}

// INSTR = true is the instrumented instantiation (CVO_KERNEL_CLOCK / CVO_PHASE_TICKS); the production one carries no
// time stamps at all.
template <typename IdxT, int ASSOC_CAP, bool GENERAL, bool INSTR>
__global__ __launch_bounds__(ASSOC_THREADS, GENERAL ? 1 : CVO_ASSOC_WAVES) void k_assoc(const PairDesc* __restrict__ descs,
                                                          const DevParams* __restrict__ Pp,
                                                          const PairState* __restrict__ states,
                                                          const char* __restrict__ arena, int lean_nblk_pairs,
                                                          unsigned stride256, int Npad) {
  const unsigned long long tt0 = INSTR ? __builtin_readcyclecounter() : 0ull;
  // one packed argument keeps everything inside the preloaded kernel-argument registers
  const int lean = lean_nblk_pairs & 0xf, nblk = (lean_nblk_pairs >> 4) & 0xffff, n_pairs = (int)((unsigned)lean_nblk_pairs >> 20);
  PairBlock pb;
  if (!pair_block(nblk, n_pairs, pb)) return;
  const PairDesc* __restrict__ D = descs + pb.pair;
  const PairState* __restrict__ st = states + pb.pair;  // == D->st, without the dependent pointer load
  AssocRowHead head;
  {
    const char* wb = arena + (size_t)pb.pair * ((size_t)stride256 << 8);
    const int pos = pb.bx * ASSOC_THREADS + threadIdx.x;  // < Npad; values of rows >= N are never used
    head.ip = GENERAL ? reinterpret_cast<const int*>(wb + row_off_ip(Npad))[pos] : 0;  // (only the feature lookups need it)
    head.j1 = (int)reinterpret_cast<const IdxT*>(wb + row_off_cand_j(Npad))[pos];
    head.x = reinterpret_cast<const float4*>(wb + row_off_xp4(Npad))[pos];
    head.cnt = __float_as_int(head.x.w);  // (k_list packs the row's candidate count next to its coordinates)
  }
  // everything the prologue branches on, requested in one burst of scalar loads (a chain of dependent ~0.5 us
  // round trips in front of every block is what this latency-bound kernel can least afford)
  const int status_v = st->status, rebuild_v = st->rebuild, n_ovf_v = st->n_ovf;
  const DevParams P = *Pp;
  {
    // ... including what the row loop needs first: the empty asm keeps these loads above the early exits, so they
    // are all in flight together instead of one round trip after each branch
    const int n = D->N, k = st->K;
    const int* a0 = D->cand_cnt;
    const void* a1 = D->cand_j;
    const float4* a2 = D->xp4;
    const float4* a3 = D->ys4;
    const EllEntry* a4 = D->ell;
    const float e = st->ell, r0 = st->Rinv[0], t0 = st->Tinv[0];
    // (c, d, log_geo and d2_c_thres share one 16-byte scalar load: with all four pinned none of its registers is dead, so
    // the allocator cannot hand one to another load of this burst - that reuse put a wait, one more round trip, in the
    // middle of it: +0.6 us per iteration for a lone pair, found in the ISA)
    asm volatile("" ::"s"(n), "s"(k), "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(e), "s"(r0), "s"(t0), "s"(P.sp_thres),
                 "s"(P.log_geo), "s"(P.c), "s"(P.d), "s"(P.d2_c_thres));
  }
  const bool replay = (lean & 2) != 0;  // cvo_debug_time_kernels: re-run on the state the last call left behind
  if (!replay && status_v != 0) return;
  // lean graph (no rebuild / dense kernels inside the iteration): a pair whose list has expired, or that has
  // rows for k_assoc_dense, does not advance; it waits for the next rebuild opportunity / for the host to
  // switch its group to the full graph (k_coeff skips it too and tells the host)
  if ((lean & 1) && (rebuild_v || (n_ovf_v > 0 && !(lean & 4)))) return;  // (bit 2: k_assoc_dense follows in this graph)
  pair_clock_begin(INSTR && P.kernel_clock && (lean & 3) == 1 && pb.bx == 0, const_cast<PairState*>(st), 0);
  __shared__ AssocShared S;
  assoc_phase<IdxT, ASSOC_CAP, GENERAL, INSTR>(P, D, load_iter_view(st), S, pb.bx, head);
  // Everything from here on - the block's partial is on its way, the last-block counter, possibly the twist - is the
  // first wave's business.  The other waves retire now instead of sitting on their registers through a store
  // acknowledgement and an atomic round trip (~2 us of a ~7 us wave life; with thousands of waves queued behind them
  // that wait was throughput, not just latency).  A barrier only counts the waves that are still alive.
  if (threadIdx.x >= 64) return;
  // lean graph, or a pair without overflow rows in the full one (k_assoc_dense then has nothing to add and leaves at
  // once): nothing else adds to the flow, the twist of the iteration can be finished here
  if ((n_ovf_v == 0 || ((lean & 3) && !(lean & 4))) && P.mode == 0) {  // (bit 1: the timing replay includes it)
    const unsigned long long clk0 = pair_clock_peek(INSTR && P.kernel_clock && (lean & 3) == 1, st, 0);
    const bool last = flow_gate(D, nblk, nblk);
    if (last && threadIdx.x == 0 && clk0) D->st->clk_last_assoc = pair_clock_ticks(clk0);  // added up by the update
  }
  if (INSTR && P.phase_ticks && threadIdx.x == 0) {
    g_phase_ticks[0][blockIdx.x & 8191][0] = tt0;
    g_phase_ticks[0][blockIdx.x & 8191][3] = __builtin_readcyclecounter();
  }
}

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

// flow partials k_assoc_dense leaves for a pair with n_ovf overflow rows (4 waves per block, one row per wave at a time)
__device__ __forceinline__ int dense_parts(int dense_blocks, int n_ovf) { return min(dense_blocks, (n_ovf + 3) >> 2); }

template <bool GENERAL, int DENSE_WAVES>
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
  if (n_ovf == 0 && P.mode == 0) return;  // nothing to add: k_assoc has finished the twist, the update skips these slots
  if (st->rebuild) return;  // (lean graphs with this kernel: the pair waits for its rebuild opportunity, see k_assoc)
  // one row per wave is the most there is to do: blocks beyond that leave no partial and stay out of the gate (the last
  // block's reduction and the update read nblk_assoc + dense_parts() slots - with the whole grid's 1024 that tail alone
  // was most of a launch that serves a few dozen rows)
  const int n_parts = P.mode == 0 ? dense_parts((int)gridDim.x, n_ovf) : (int)gridDim.x;
  if ((int)blockIdx.x >= n_parts) return;
  const bool all_dense = st->all_dense != 0;
  __shared__ float2 s_hits[DENSE_WAVES][128][6];  // per wave: the hits of one step, compacted ({flow term, value} per component)
  __shared__ unsigned s_keys[DENSE_WAVES][LONG_CAP];  // per wave: the long list being built (sort keys)
  double red[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long nnz_sum = 0;
  unsigned nnz_max = 0;
  if (n_ovf > 0) {
    const Pose pose = load_pose(st);
    const FeatDen F = make_feat_den(P);
    const bool long_lists = !all_dense && P.long_lists != 0 && D->long_j != nullptr;
    const unsigned long long gen = (P.call_serial << 24) | (unsigned long long)((unsigned)st->n_builds & 0xffffffu);
    for (int q = blockIdx.x * DENSE_WAVES + wave; q < n_ovf; q += (int)gridDim.x * DENSE_WAVES) {
      // a position of k_list's ordering (all per-row outputs are stored by position); dense regime: every row
      const int r_sorted = all_dense ? q : D->ovf_rows[q];
      const int i = D->ip[r_sorted];
      const float4 x = D->xp4[r_sorted];
      const RowData r = make_row(P, x, st->ell);
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
        int col[2] = {0, 0};
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int c = j0 + 64 * h + lane;
          if (c < n_cand) {
            if (listed) {  // list entries are sorted positions: coordinates and features from the spatially ordered arrays
              const int p = fresh ? (int)(s_keys[wave][c] & 0xffffu) : (int)lj[c];
              col[h] = p;
              ok[h] = eval_pair<GENERAL>(P, D, F, pose, i, r, p, D->ys4[p], a[h], yt[h]) && (a[h] > P.sp_thres);
            } else {
              col[h] = c;
              ok[h] = eval_pair<GENERAL>(P, D, F, pose, i, r, GENERAL ? D->yinv[c] : 0, D->y4[c], a[h], yt[h]) && (a[h] > P.sp_thres);
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
            D->ell[(size_t)rank * N + r_sorted] = EllEntry{a[h], yt[h].x, yt[h].y, yt[h].z};
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
      if (lane == 0) {
        D->nnz_row[r_sorted] = nnz;
        red[0] += (double)(o0 / P.c);
        red[1] += (double)(o1 / P.c);
        red[2] += (double)(o2 / P.c);
        red[3] += (double)(v0 / P.d);
        red[4] += (double)(v1 / P.d);
        red[5] += (double)(v2 / P.d);
        red[6] += asum;
        nnz_sum += nnz;
        nnz_max = max(nnz_max, nnz);
      }
    }
  }
  // block partials are always written (zeros when there was nothing to do): k_coeff / k_update sum them
  __shared__ double s_red[DENSE_WAVES][8];
  __shared__ unsigned long long s_cnt[DENSE_WAVES][2];
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 7; c++) s_red[wave][c] = red[c];
    s_cnt[wave][0] = nnz_sum;
    s_cnt[wave][1] = nnz_max;
  }
  __syncthreads();
  const size_t slot = (size_t)D->nblk_assoc + blockIdx.x;
  if (threadIdx.x < 7) {
    const int c = threadIdx.x;
    double t = s_red[0][c];
#pragma unroll
    for (int w = 1; w < DENSE_WAVES; w++) t += s_red[w][c];
    st_x<true>(D->flow_part + slot * 8 + c, t);
  } else if (threadIdx.x == 8) {
    unsigned long long* cp = D->cnt_part + slot * 4;
    unsigned long long c0 = 0, c1 = 0;
#pragma unroll
    for (int w = 0; w < DENSE_WAVES; w++) {
      c0 += s_cnt[w][0];
      c1 = max(c1, s_cnt[w][1]);
    }
    cp[0] = c0;
    cp[1] = c1;
    cp[2] = 0;
    cp[3] = 0;
  }
  // full graph: the twist of the iteration from the partials of k_assoc (an earlier launch) and of this kernel
  if (P.mode == 0) flow_gate(D, n_parts, D->nblk_assoc + n_parts);
}

// ------------------------------------------------------------------------------------------
// Coefficient phase: normalised twist (compute_flow host half, CvoGPU.cu:824-835) + B,C,D,E partials, one
// thread per row position, blocks of ASSOC_THREADS rows.
// ------------------------------------------------------------------------------------------
struct CoeffShared {
  BlockRedShared<4> red;
};

// one nonzero (i, j): compute_step_size_xi for target j (CvoGPU.cu:974-986) + compute_step_size_poly_coeff
// (CvoGPU.cu:1053-1078); yy is the transformed target
__device__ __forceinline__ void coeff_entry(const XiMats& M, const float4 x, float temp_coef, const V3 yy, float A_ij,
                                            double& Bi, double& Ci, double& Di, double& Ei) {
  const V3 w{M.omega[0], M.omega[1], M.omega[2]};
  const V3 c = cross_dev(w, yy);
  const V3 xiz{c.x + M.v[0], c.y + M.v[1], c.z + M.v[2]};
  V3 t = matvec_dev(M.m2, yy);
  const V3 xi2z{t.x + M.ohv.x, t.y + M.ohv.y, t.z + M.ohv.z};
  t = matvec_dev(M.m3, yy);
  const V3 xi3z{t.x + M.m2v.x, t.y + M.m2v.y, t.z + M.m2v.z};
  t = matvec_dev(M.m4, yy);
  const V3 xi4z{t.x + M.m3v.x, t.y + M.m3v.y, t.z + M.m3v.z};
  const float normxiz2 = dot3_dev(xiz.x, xiz.y, xiz.z, xiz.x, xiz.y, xiz.z);
  const float xiz_dot_xi2z = -dot3_dev(xiz.x, xiz.y, xiz.z, xi2z.x, xi2z.y, xi2z.z);
  const float epsil_const = __builtin_fmaf(2.0f, dot3_dev(xiz.x, xiz.y, xiz.z, xi3z.x, xi3z.y, xi3z.z),
                                           dot3_dev(xi2z.x, xi2z.y, xi2z.z, xi2z.x, xi2z.y, xi2z.z));
  const float dfx = x.x - yy.x, dfy = x.y - yy.y, dfz = x.z - yy.z;
  const float beta_ij = (float)(-2.0 * temp_coef * (double)dot3_dev(xiz.x, xiz.y, xiz.z, dfx, dfy, dfz));
  const float gamma_ij =
      (-temp_coef) * (normxiz2 + dot3_dev(2.0f * xi2z.x, 2.0f * xi2z.y, 2.0f * xi2z.z, dfx, dfy, dfz));
  const float delta_ij =
      (float)(2.0 * temp_coef * (double)(xiz_dot_xi2z + dot3_dev(-xi3z.x, -xi3z.y, -xi3z.z, dfx, dfy, dfz)));
  const float epsil_ij =
      (-temp_coef) * (epsil_const + dot3_dev(2.0f * xi4z.x, 2.0f * xi4z.y, 2.0f * xi4z.z, dfx, dfy, dfz));
  Bi += (double)(A_ij * beta_ij);
  Ci += (double)A_ij * ((double)gamma_ij + (double)(beta_ij * beta_ij) / 2.0);
  // beta^3 / 6.0 (CvoGPU.cu:1072): the IEEE division with its constant half folded (rcp_refined / div_by, cvo_device.h)
  Di += (double)A_ij * ((double)__builtin_fmaf(beta_ij, gamma_ij, delta_ij) +
                        div_by((double)(beta_ij * beta_ij * beta_ij), 6.0, rcp_refined(6.0)));
  Ei += (double)A_ij * ((double)__builtin_fmaf(beta_ij, delta_ij, epsil_ij) +
                        1 / 2.0 * beta_ij * beta_ij * gamma_ij + 1 / 2.0 * gamma_ij * gamma_ij +
                        1 / 24.0 * beta_ij * beta_ij * beta_ij * beta_ij);
}

// Rows of this block, in two steps so that the first loads of the row loop (count -> first ELL entry -> its target:
// three dependent round trips) are in flight while the twist is reduced.
struct CoeffRowHead {
  unsigned nnz;
  float4 x;
  EllEntry e_n;  // the row's first entry of this block's slice
};
// COH: the block partial is read by another block of the same launch.
template <bool COH>
__device__ __forceinline__ void coeff_rows(const DevParams& P, const PairDesc* __restrict__ D, const float ell,
                                           CoeffShared& S, const XiMats& Mu, const CoeffRowHead& h, const int bx,
                                           const int q, const int nsplit) {
  const int N = D->N;
  const int i = bx * ASSOC_THREADS + threadIdx.x;
  double Bi = 0, Ci = 0, Di = 0, Ei = 0;
  // this block's share of the row: slots q, q + nsplit, ...  (small clouds whose rows sit on K_max would
  // otherwise leave the chip to a handful of waves walking hundreds of entries each).  Software pipeline: the
  // next entry's index / value / target are in flight while the current one is evaluated.
  const unsigned nnz = h.nnz;
  if ((unsigned)q < nnz) {
    const float4 x = h.x;
    float temp_ell = ell;
    if (P.use_range_ell) {
      const float d2_sqrt = sqrtf(dot3_dev(x.x, x.y, x.z, x.x, x.y, x.z));
      temp_ell = compute_range_ell(temp_ell, d2_sqrt);
    }
    const double cden = 2.0 * temp_ell * temp_ell;
    const float temp_coef = (float)div_by(1.0, cden, rcp_refined(cden));  // 1 / (2.0 * ell * ell), CvoGPU.cu:1060
    EllEntry e_n = h.e_n;
    for (unsigned s = (unsigned)q; s < nnz; s += (unsigned)nsplit) {
      const EllEntry e = e_n;
      if (s + nsplit < nnz) e_n = D->ell[(size_t)(s + nsplit) * N + i];  // next entry in flight
      // (the transformed target k_assoc evaluated the pair with: transform_point of the same operands, stored)
      coeff_entry(Mu, x, temp_coef, V3{e.yx, e.yy, e.yz}, e.a, Bi, Ci, Di, Ei);
    }
  }
  const double red[4] = {Bi, Ci, Di, Ei};
  const double tot = block_reduce_lds<4>(S.red, red);
  if (threadIdx.x < 32 && (threadIdx.x & 7) == 0) st_x<COH>(D->coef_part + ((size_t)bx * nsplit + q) * 4 + (threadIdx.x >> 3), tot);
}

// ------------------------------------------------------------------------------------------
// k_update: per-pair scalar bookkeeping, one wave per pair.  INIT = true is the launch before the first
// iteration (no bookkeeping, state comes from the host).
// ------------------------------------------------------------------------------------------
constexpr int HOT_DWORDS = (int)(offsetof(PairState, sq) / 4);  // the scalar part of the state (the 4 KB of indicator FIFOs stay in HBM)
struct UpdateShared {
  double c[4];
  unsigned long long n[4];
  unsigned hot[HOT_DWORDS];
};

// What the update reads from the pair descriptor, requested in one burst of scalar loads (k_coeff issues it while
// the last-block counter is on its way): every field first touched in the middle of the serial tail would be
// another cold round trip there.
struct UpdDesc {
  PairState* st;
  const double* coef_part;
  const double* flow_part;
  const unsigned long long* cnt_part;
  cvo_trace_t* trace;
  int* status_out;
  int* want_out;
  int* status_host;
  int* want_host;
  int nblk_coeff, N, M;
  float ymax;
  double sqrt_nm;
};
__device__ __forceinline__ UpdDesc load_upd_desc(const PairDesc* __restrict__ D) {
  UpdDesc u;
  u.st = D->st;
  u.coef_part = D->coef_part;
  u.flow_part = D->flow_part;
  u.cnt_part = D->cnt_part;
  u.trace = D->trace;
  u.status_out = D->status_out;
  u.want_out = D->want_out;
  u.status_host = D->status_host;
  u.want_host = D->want_host;
  u.nblk_coeff = D->nblk_coeff;
  u.N = D->N;
  u.M = D->M;
  u.ymax = D->ymax;
  u.sqrt_nm = D->sqrt_nm;
  asm volatile("" ::"s"(u.st), "s"(u.coef_part), "s"(u.flow_part), "s"(u.cnt_part), "s"(u.trace), "s"(u.status_out),
               "s"(u.want_out), "s"(u.nblk_coeff), "s"(u.N), "s"(u.M), "s"(u.ymax));
  return u;
}

// Executed by the first wave of the calling block (the other threads only take part in the barriers).
// flags: bit 1 = the rebuild kernels run right after this iteration, bit 2 = called from k_coeff, bit 3 = replay for
// timing (nothing is written back), bits 8.. = how many
// iterations the list has to survive without another rebuild opportunity (0 in the full graph).  n_flow_parts: association partials to
// sum (the lean graph has no k_assoc_dense, so its slots are not read).
// RES: called from the resident kernel - counts, state and indicator FIFOs were (or may have been) written by OTHER
// blocks of the SAME launch, so they are read with L1-bypassing loads as well.
// PairState::want_full, the graph a pair asks the host for: a LEVEL - 2 = a rebuild opportunity in every iteration,
// 1 = every lean_U2 iterations (short lean graph), 0 = every lean_U (lean graph), -1 = calm, one per chunk - and whether
// k_assoc_dense has to run (overflow rows / the dense regime).  Encoded as: level without the dense kernel; 4 = level 2
// with it; 8 + (level + 1) = a leaner level with it.  (3 is the resident launch's time-out, see k_resident.)
__device__ __forceinline__ int want_level(int w) { return w == 4 ? 2 : (w >= 8 ? w - 9 : w); }
__device__ __forceinline__ int want_encode(int level, bool dense) { return !dense ? level : (level >= 2 ? 4 : 9 + level); }

struct NoEarlyPublish {
  __device__ __forceinline__ void operator()(int, int, float, const float*, const float*) const {}
};
// Early: called by thread 0 as soon as everything the NEXT iteration's row blocks need is known - stop word (1 done,
// 2 list expired, 3 overflow rows, 0 go on), K, ell, Rinv, Tinv - i.e. in front of the list bookkeeping, the skin
// arithmetic and the write-back of the state (the resident kernel publishes them there: the next association runs while
// this wave finishes its tail).
template <bool INIT, bool COH, bool RES = false, typename Early = NoEarlyPublish>
__device__ __forceinline__ void update_body(const UpdDesc& D, const DevParams& P, int flags,
                                            int n_flow_parts, UpdateShared& U, const float* twist,
                                            const unsigned* preloaded_hot, unsigned long long clk0 = 0ull,
                                            Early early = Early()) {
  PairState* const gst = D.st;
  const bool trio_follows = INIT || (flags & 2) != 0;
  const bool dry = (flags & 8) != 0;  // timing replay: compute everything, write nothing back
  const int horizon = flags >> 8;
  double* const s_c = U.c;
  unsigned long long* const s_n = U.n;
  unsigned* const s_hot = U.hot;
  const int tid = threadIdx.x;
  const bool act = tid < 64;
  CVO_UPD_STAMP(0);

  // the scalar part of the state is staged through LDS: one coalesced burst in, one out, instead of
  // dozens of dependent global accesses from a single lane
  static_assert(HOT_DWORDS <= 128, "two dwords per lane of the first wave cover the scalar state");
  if (act && preloaded_hot) {  // the caller read the state into registers while it waited for something else
    s_hot[tid] = preloaded_hot[0];
    if (tid + 64 < HOT_DWORDS) s_hot[tid + 64] = preloaded_hot[1];
  } else if (act) {
    for (int q = tid; q < HOT_DWORDS; q += 64) s_hot[q] = reinterpret_cast<const unsigned*>(gst)[q];
  }
  PairState* const st = reinterpret_cast<PairState*>(s_hot);
  float* const sq = gst->sq;
  float* const eq = gst->eq;
  if (!INIT && act) {
    // The four thrust::reduce of compute_step_size (CvoGPU.cu:1118-1121) and the nonzero / max counts
    // (SparseKernelMat.cu:37-46, CvoGPU.cu:1518): lane l owns component (l & 3) of blocks l>>2, l>>2 + 16, ...
    // so all loads are in flight at once; a fixed xor-shuffle tree finishes (deterministic order).
    // Row c of 16 lanes owns component c; lane l of the row takes blocks l, l + 16, ... (eight loads in flight), the
    // row meets through a DPP butterfly (no LDS) and every lane reads the four results with v_readlane.
    const int nba = n_flow_parts, nbc = D.nblk_coeff;
    const int c = tid >> 4, bl = tid & 15;
    double s = 0;
    // the counts were written by the association kernel(s), i.e. before this launch: plain loads, requested ahead of
    // the coherent ones so that the two round trips overlap
    // (every count of one iteration fits 32 bits - at most rows x K_max nonzeros, rows x targets candidates, checked
    // at set-up - and the block partials are read as such: the 64-bit DPP steps cost four times the instructions)
    // (global, not flat, addresses: the two groups of loads below are then really in flight together - a flat load makes
    // the compiler wait for vmcnt AND lgkmcnt to drain before anything that follows it)
    const CVO_GLOBAL unsigned* cnt32 = as_global(reinterpret_cast<const unsigned*>(D.cnt_part));
    const CVO_GLOBAL double* coef_part = as_global(D.coef_part);
    unsigned vq[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int b = bl + 16 * u;
      vq[u] = b < nba ? ld_g<RES>(cnt32 + ((size_t)b * 4 + c) * 2) : 0u;
    }
    if (P.mode == 0) {
      // eight (coherent) loads in flight per lane, summed in block order
      for (int b0 = bl; b0 < nbc; b0 += 128) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int b = b0 + 16 * u;
          v[u] = b < nbc ? ld_g<COH>(coef_part + (size_t)b * 4 + c) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
      }
    } else if (c == 0) {
      for (int b = bl; b < nba; b += 16) s += as_global(D.flow_part)[(size_t)b * 8 + 6];
    }
    // (component 2, the candidate statistic, can exceed 32 bits for very large clouds before the dense regime engages:
    // it saturates instead of wrapping; nnz / overflow rows are bounded by the set-up check)
    auto addsat = [](unsigned a, unsigned b) { const unsigned r = a + b; return r < a ? 0xffffffffu : r; };
    unsigned q = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) q = (c == 1) ? max(q, vq[u]) : addsat(q, vq[u]);
    for (int b0 = bl + 128; b0 < nba; b0 += 128) {
      unsigned v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int b = b0 + 16 * u;
        v[u] = b < nba ? ld_g<RES>(cnt32 + ((size_t)b * 4 + c) * 2) : 0u;
      }
#pragma unroll
      for (int u = 0; u < 8; u++) q = (c == 1) ? max(q, v[u]) : addsat(q, v[u]);
    }
    s += dpp_f64<DPP_XOR1>(s);
    s += dpp_f64<DPP_XOR2>(s);
    s += dpp_f64<DPP_HALF_MIRROR>(s);
    s += dpp_f64<DPP_MIRROR>(s);
    {
      auto meet = [&](unsigned o) { q = (c == 1) ? max(q, o) : addsat(q, o); };
      meet((unsigned)dpp_i32<DPP_XOR1>((int)q));
      meet((unsigned)dpp_i32<DPP_XOR2>((int)q));
      meet((unsigned)dpp_i32<DPP_HALF_MIRROR>((int)q));
      meet((unsigned)dpp_i32<DPP_MIRROR>((int)q));
    }
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
      const double sv = lane_f64(s, 16 * cc);
      const unsigned long long qv = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)q, 16 * cc);
      if (tid == 0) {
        s_c[cc] = sv;
        s_n[cc] = qv;
      }
    }
  }
  __syncthreads();
  // fronts of the indicator FIFOs (HBM), on their way while the step is computed
  float e_front = 0.f, s_front = 0.f;
  if (!INIT && tid == 0) {
    e_front = ld_x<RES>(eq + st->e_head);
    s_front = ld_x<RES>(sq + st->s_head);
  }
  CVO_UPD_STAMP(1);
  // the step of this iteration: the cubic's real roots are searched on three lanes side by side
  float step_w = 0.f;
  if (!INIT && act && P.mode == 0) step_w = select_step<true>(s_c[0], s_c[1], s_c[2], s_c[3], P.min_step, P.max_step);
  CVO_UPD_STAMP(2);
  if (tid == 0) {
    int done = 0;
    if (twist) {  // k_coeff: every block derived the same normalised twist
      for (int c = 0; c < 3; c++) {
        st->omega[c] = twist[c];
        st->v[c] = twist[3 + c];
      }
    }
    if (!INIT) {
      if (flags & 4) st->epoch++;  // generation of k_coeff's last-block counter
      const unsigned nnz = (unsigned)s_n[0], max_nnz = (unsigned)s_n[1];
      st->nnz = nnz;
      st->max_nnz = max_nnz;
      st->ncand = st->ncand_list;  // candidates of the current lists (k_list), evaluated exactly in this iteration
      st->ncand_total += st->ncand_list;
      st->noverflow = s_n[3];
      st->K_last = st->K;  // the stride upstream wrote this iteration's A matrix with (gpu_association_to_cpu)
      if (P.mode != 0) {  // single evaluation: A_sum (SparseKernelMat.cu:62-68)
        st->asum = s_c[0];
        done = 1;
      } else {
        const double B = s_c[0], C = s_c[1], Dd = s_c[2], E = s_c[3];
        st->B = B;
        st->C = C;
        st->D = Dd;
        st->E = E;
        const float step = step_w;
        st->step = step;
        const int k = st->k;
        const int K_used = st->K;
        const float ell_used = st->ell;
        const float* om = st->omega;
        const float* vv = st->v;
        double dist = 0;
        auto sqnorm3d = [](const float* a) {
          const double x = a[0], y = a[1], z = a[2];
          return x * x + (y * y + z * z);
        };
        // `omega.norm() < eps && v.norm() < eps` (double sqrt of the float-derived sums).  The twist is normalised, so
        // one of the two is ~1: sqrt is monotonic and correctly rounded, x > eps^2 (1 + 1e-12) decides sqrt(x) >= eps
        // without the ~60 dependent instructions of a double square root on the serial tail (exact shortcut).
        const double n2o = sqnorm3d(om), n2v = sqnorm3d(vv);
        const double eps2_hi = (double)P.eps * (double)P.eps * (1.0 + 1e-12);
        bool vanished = false;
        if (!(n2o > eps2_hi || n2v > eps2_hi)) vanished = sqrt(n2o) < (double)P.eps && sqrt(n2v) < (double)P.eps;
        if (vanished) {  // CvoGPU.cu:1454-1458
          auto norm3f = [](const float* a) { return sqrtf(a[0] * a[0] + (a[1] * a[1] + a[2] * a[2])); };
          if ((double)norm3f(om) < 1e-8 && (double)norm3f(vv) < 1e-8) st->ret = -1;
          done = 1;
          st->iterations = k;
        } else {
          const float xi[6] = {om[0], om[1], om[2], vv[0], vv[1], vv[2]};
          float dtrans[12];
          exp_sek3(xi, step, dtrans);  // CvoGPU.cu:1462
          // (the increment stays in its twelve floats; widened where it is used: kept as doubles it held 24 registers across
          // everything up to the - rarely taken - logarithm below, and the update's registers are what caps k_coeff's occupancy)
          auto dRd = [&](int q) { return (double)dtrans[4 * (q / 3) + (q % 3)]; };
          auto dTd = [&](int i) { return (double)dtrans[4 * i + 3]; };
          // (the running pose is fetched from the staged state row by row, only now: held in registers from the top of the
          // update it was live through Exp_SEK3, where the register count of the whole kernel peaks)
          float Rc[9], Tc[3];  // (requested together, one LDS round trip; the rows below are kept apart by scheduling
          for (int q = 0; q < 9; q++) Rc[q] = st->R[q];  // barriers: one row's double temporaries at a time)
          for (int q = 0; q < 3; q++) Tc[q] = st->T[q];
#pragma unroll
          for (int i = 0; i < 3; i++) {  // CvoGPU.cu:1463-1469
            const double r0 = Rc[3 * i + 0], r1 = Rc[3 * i + 1], r2 = Rc[3 * i + 2];
            const float tn = (float)((r0 * dTd(0) + (r1 * dTd(1) + r2 * dTd(2))) + (double)Tc[i]);
            float rn[3];
            for (int j = 0; j < 3; j++) rn[j] = (float)(r0 * dRd(0 + j) + (r1 * dRd(3 + j) + r2 * dRd(6 + j)));
            st->T[i] = tn;
            for (int j = 0; j < 3; j++) st->R[3 * i + j] = rn[j];
            __builtin_amdgcn_sched_barrier(0);
          }
          // dist = || log SE3(dR, dT) || (CvoGPU.cu:1473-1476) decides one thing: dist < eps_2.  dR / dT are the float
          // Exp_SEK3 of a unit twist times `step`, so in exact arithmetic dist = step * |xi|_6 = step; the float
          // rounding of dtrans (6e-8 per entry, entries <= 1) and of the normalisation move it by < 1e-6 + 1e-4 step.
          // When step clears eps_2 by that margin the comparison is decided and the ~300 dependent double-precision
          // instructions of the log (quaternion, atan, sin / cos) stay off the serial tail: exact shortcut, like the
          // min_step clamp of select_step.  Not taken when the value itself is recorded (trace) and in Exp_SEK3's
          // theta < 1e-6 branch (translation v instead of step * v: dist ~ 1 there).
          const bool want_trace = !dry && D.trace && st->n_trace < P.trace_capacity &&
                                  (k < P.trace_dense || (P.trace_every > 0 && k % P.trace_every == 0));
          const float theta_f = sqrtf(om[0] * om[0] + (om[1] * om[1] + om[2] * om[2]));
          if (!want_trace && theta_f >= 1e-6f && step * 0.9999f - 1e-6f > P.eps_2 && step <= 1.f)
            dist = (double)step;
          else
          {
            double dR[9], dT[3];
            for (int q = 0; q < 9; q++) dR[q] = dRd(q);
            for (int i = 0; i < 3; i++) dT[i] = dTd(i);
            dist = se3_log_norm(dR, dT);
          }
          const float ip_curr = (float)((double)nnz / D.sqrt_nm);  // 1486 (sqrt(N * M): IEEE, evaluated on the host)
          const bool need_decay_ell = dry ? false : indicator_update(st, sq, eq, ip_curr, P.window, P.stable_thr, e_front, s_front);
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
        if (!dry && D.trace && st->n_trace < P.trace_capacity &&
            (k < P.trace_dense || (P.trace_every > 0 && k % P.trace_every == 0))) {
          cvo_trace_t* tr = D.trace + st->n_trace;
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
          // (re-read from the staged state: eight + twelve values that would otherwise stay in registers across Exp_SEK3, the
          // pose update and the indicator just for this optional record)
          tr->B = st->B;
          tr->C = st->C;
          tr->D = st->D;
          tr->E = st->E;
          tr->dist = dist;
          for (int q = 0; q < 9; q++) tr->R[q] = st->R[q];
          for (int q = 0; q < 3; q++) tr->T[q] = st->T[q];
          st->n_trace++;
        }
      }
    }
    CVO_UPD_STAMP(3);
    // update_tf (CvoGPU.cu:94-112): the transform applied next, and the returned matrix when done
    float Ri[9], Ti[3];
    update_tf(st->R, st->T, Ri, Ti);
    {
      // Candidate-list reuse.  Target j moves by at most |Ri - Rb|_F * |y0_j| + |Ti - Tb| between the pose the
      // bitmap was built with and the one applied next.  The scan added skin_rot * rho_i + skin_tr to the cut-off
      // radius of row i, rho_i >= |y0_j| for every target that can come within the row's radius (k_prep); so as long
      // as |Ri - Rb|_F <= skin_rot and |Ti - Tb| <= skin_tr (and ell, hence every radius, has not grown) the bitmap
      // still contains every pair the exact test of k_assoc can accept.
      // (None of this reaches a result: the allowances only have to be what k_prep adds to the radii, and the motion
      // bounds carry a 0.1 % margin - hardware square roots and reciprocals, 1 ulp, instead of ~12 dependent
      // instructions per IEEE sqrtf / division on the serial tail.)
      auto fsqrt = [](float x) { return __builtin_amdgcn_sqrtf(x); };
      auto frcp = [](float x) { return __builtin_amdgcn_rcpf(x); };
      // how the last build classed the rows, the regime and the request in force: read here, once, so that the decisions
      // at the end of this block do not each start with a staging-area round trip of their own
      const int c_ovf = st->n_ovf, c_scan = st->n_scan, c_want = st->want_full;
      int c_dense = st->all_dense;
      const float ell_next = st->ell;
      const float radius = ell_next * fsqrt(fmaxf(-2.f * P.log_geo, 0.f));  // cut-off radius for l = ell
      float dr = 0, dt = 0, dr1 = 0, dt1 = 0;
      for (int q = 0; q < 9; q++) {
        const float a = Ri[q] - st->Rb[q], b = Ri[q] - st->Rinv[q];
        dr = __builtin_fmaf(a, a, dr);
        dr1 = __builtin_fmaf(b, b, dr1);
      }
      for (int q = 0; q < 3; q++) {
        const float a = Ti[q] - st->Tb[q], b = Ti[q] - st->Tinv[q];
        dt = __builtin_fmaf(a, a, dt);
        dt1 = __builtin_fmaf(b, b, dt1);
      }
      const float ymax = D.ymax;
      float rot_b = fsqrt(dr) * 1.001f, tr_b = fsqrt(dt) * 1.001f;   // since the build (the rounding slack of the two
                                                                      // transform evaluations is part of every row's skin)
      float rot_1 = fsqrt(dr1), tr_1 = fsqrt(dt1);                    // this iteration alone
      float step_move = rot_1 * ymax + tr_1;                          // what this iteration moved the farthest target
      if (P.debug_no_motion_bound) rot_b = tr_b = rot_1 = tr_1 = step_move = 0.f;  // (tests: a deliberately broken bound)
      // share of the allowances used up / used per iteration (inf when an allowance is zero and something moved)
      auto share = [&](float used, float allowance) { return used <= 0.f ? 0.f : (allowance > 0.f ? used * frcp(allowance) : __builtin_inff()); };
      const float used = fmaxf(share(rot_b, st->skin_rot), share(tr_b, st->skin_tr));
      const float rate = fmaxf(share(rot_1, st->skin_rot), share(tr_1, st->skin_tr));
      st->last_used = used;
      st->last_rate = rate;
      // the list is unusable for the coming iteration ...
      // (A list built for a larger ell stays a superset: rebuilding it after ell has shrunk only sheds candidates.  That
      // rebuild is optional, so it waits for a rebuild opportunity - flagged in the middle of a lean period it would
      // stall the pair until the next one - and, in a batch, for an iteration count that is a multiple of 64: the pairs
      // of a sub-batch decay in step, their shrink rebuilds then share one pass of the rebuild kernels instead of
      // putting real work into a different one each.)
      const bool shrink_due = ell_next < P.rebuild_shrink * st->ell_build;
      const bool shrink_now = shrink_due && trio_follows &&
                              ((st->k & P.shrink_align) == 0 || ell_next < 0.85f * P.rebuild_shrink * st->ell_build);
      bool rebuild = INIT || P.mode != 0 || !(used <= 1.f) || ell_next > st->ell_build || shrink_now;
      // ... or would expire before the next rebuild opportunity of the lean graph
      if (trio_follows && horizon > 0 && !(used + P.horizon_margin * (float)horizon * rate <= 1.f)) rebuild = true;
      // ... and, in a batch, at the common iteration counts of the optional rebuilds: a list that would not survive
      // until the next of them is renewed now, together with the other pairs' (the pass runs anyway), instead of
      // putting work into a pass of its own some opportunities later
      if (trio_follows && horizon > 0 && P.shrink_align > 0 && (st->k & P.shrink_align) == 0 &&
          !(used + P.horizon_margin * (float)(P.shrink_align + 1) * rate <= 1.f))
        rebuild = true;
      // Dense regime (rows sitting on K_max, e.g. the first iterations of an outdoor pair at a large ell): when most
      // rows overflow their lists anyway, lists are pointless - every row goes to k_assoc_dense (the reference's
      // literal ordered scan), nothing is rebuilt while that lasts, and the pair returns to lists once the rows have
      // thinned out (mean nonzeros per row below 12, far from the 32 / 64 a list holds).
      if (!INIT && P.mode == 0 && P.dense_regime) {
        const bool was = c_dense != 0;
        // (with long lists an overflow row costs what its candidates cost: the literal scan of everything only pays when
        // most rows are beyond even those, when the target cloud is small - 2048 targets are 32 lane steps, no bitmap, no
        // sort, no rebuilds: the demo pair on its K cap - or when the rows see a third of it anyway)
        const bool now = was ? (unsigned long long)st->nnz >= 12ull * (unsigned long long)D.N
                             : (2 * c_scan > D.N ||
                                (2 * c_ovf > D.N &&
                                 (D.M <= 2048 || 3ull * st->ncand_list > (unsigned long long)D.N * (unsigned long long)D.M)));
        if (now != was) {
          c_dense = now ? 1 : 0;
          st->all_dense = c_dense;
          rebuild = true;
        } else if (now) {
          rebuild = false;
        }
      }
      early(done ? 1 : (rebuild ? 2 : (c_ovf > 0 ? 3 : 0)), st->K, st->ell, Ri, Ti);
      if (rebuild) {
        for (int q = 0; q < 9; q++) st->Rb[q] = Ri[q];
        for (int q = 0; q < 3; q++) st->Tb[q] = Ti[q];
        st->ell_build = ell_next;
        // Skin: a longer-lived list costs (1 + s)^3 more candidates per iteration, a shorter-lived one more
        // rebuilds; s ~ 1.5 sqrt(step / radius) balances the two for this kernel set.  The lean graph only has a
        // rebuild opportunity every lean_U iterations, so it needs s >= ~1.3 lean_U step / radius; when that is
        // too much (fast motion) or rows overflow their lists, ask the host for the full graph.
        // 2 = a rebuild opportunity in every iteration; 4 = and k_assoc_dense (rows that overflowed the lists of the last
        // build, or the dense regime): the host has a full graph without the dense kernel for large clouds
        // (overflow rows as the LAST build left them: a pair that gains its first ones in a graph without the dense kernel
        // waits there and asks for it, see k_coeff)
        const bool dense_rows = c_ovf > 0 || c_dense != 0;
        // A wave of k_assoc runs as long as its longest row.  While a sixteenth of the rows overflow anyway (a clustered
        // cloud: k_assoc_dense runs in every iteration, its long lists cost what their candidates cost), rows of more
        // than 24 candidates join them - a wave per row, 64 candidates per step - instead of holding 63 neighbours back.
        st->row_max = (!INIT && P.long_lists && !c_dense && 16 * c_ovf > D.N) ? 24 : ASSOC_CAP16;
        int want_full = 2;
        float s = 0.f;
        // rows beyond every list fall back to the literal scan over all targets (k_assoc_dense): fine for a few
        // rows or a small cloud, ruinous if a generous skin pushes many rows of a large one over the edge - the skin
        // backs off by halves while the last build left such rows and recovers slowly afterwards
        if (INIT) st->skin_scale = 1.f;
        else if (c_scan > 0 && !c_dense) st->skin_scale = fmaxf(0.5f * st->skin_scale, 1.f / 64.f);
        else st->skin_scale = fminf(1.f, 1.1f * st->skin_scale);
        if (!INIT && P.mode == 0 && P.use_geo && radius > 0.f && P.skin_frac > 0.f && !c_dense) {
          const float rel = step_move * frcp(radius);
          s = st->skin_scale * P.skin_frac * fminf(fmaxf(1.5f * fsqrt(rel), P.skin_min), P.skin_max);
          const float s_lean = fmaxf(s, P.lean_skin * (float)P.lean_U * rel);
          const float s_lean2 = fmaxf(s, P.lean_skin * (float)P.lean_U2 * rel);
          // (rows that walk long lists cost what their candidates cost, whatever the skin; rows scanned literally do not)
          if (s_lean <= 0.5f && c_scan == 0) {
            s = s_lean;
            want_full = 0;
          } else if (P.lean_U2 > 0 && s_lean2 <= 0.5f && c_scan == 0) {
            s = s_lean2;  // too fast for lean_U iterations between rebuilds, slow enough for lean_U2
            want_full = 1;
          } else if (!(s >= 2.f * rel)) {
            s = 0.f;  // would not survive two iterations: plain scan every iteration
          }
        }
        if (!(s == s)) s = 0.f;
        // s * radius is what the FARTHEST target may move; split into a rotation and a translation allowance in the
        // proportion of the current motion (plus a blend of the pooled budget for either, so that a change of
        // direction does not expire the lists at once): every row's skin follows from its own distance (k_prep)
        {
          // (normalised so that the farthest row gets exactly s * radius)
          const float life = step_move > 0.f ? s * radius * frcp(step_move * (1.f + P.skin_blend)) : 0.f;  // iterations at the current speed
          const float bl = P.skin_blend;
          st->skin_rot = life * ((1.f - bl) * rot_1 + bl * step_move * frcp(fmaxf(ymax, 1e-20f)));
          st->skin_tr = life * ((1.f - bl) * tr_1 + bl * step_move);
          if (!(st->skin_rot == st->skin_rot) || !(st->skin_tr == st->skin_tr)) st->skin_rot = st->skin_tr = 0.f;
        }
        if (c_dense) want_full = -1;  // dense regime: nothing is rebuilt until the pair leaves it
        want_full = want_encode(want_full, dense_rows);
        st->want_full = want_full;
        if (!dry) {
          *D.want_out = want_full;
          *D.want_host = want_full;
        }
        st->n_builds = INIT ? 1 : st->n_builds + 1;
        st->rebuild = 1;  // cleared by k_list once bitmap and lists are current
      } else if (c_scan == 0 && !c_dense) {  // has the motion slowed down enough for a leaner graph?
        const float c = fminf(P.lean_skin, 1.3f);
        int want = want_level(c_want);
        if (used + c * (float)P.lean_U * rate <= 1.f)
          want = 0;
        else if (P.lean_U2 > 0 && used + c * (float)P.lean_U2 * rate <= 1.f)
          want = min(want, 1);
        // -1 = calm: at the current speed the list outlives P.calm_U more iterations - the host may run this pair on the
        // lean graph with ONE rebuild opportunity per chunk (the opportunities are three launches each, and in the end
        // game - the step clamped at min_step, rebuilds only when ell has decayed - nearly all of them find nothing to do)
        if (want == 0 && P.calm_U > 0 && used + c * (float)P.calm_U * rate <= 1.f) want = -1;
        want = want_encode(want, c_ovf > 0);
        if (want != c_want) {
          st->want_full = want;
          if (!dry) {
            *D.want_out = want;
            *D.want_host = want;
          }
        }
      }
    }
    CVO_UPD_STAMP(4);
    for (int q = 0; q < 9; q++) st->Rinv[q] = Ri[q];
    for (int q = 0; q < 3; q++) st->Tinv[q] = Ti[q];
    if (done || INIT || P.mode != 0) {  // the returned matrix (final update_tf, CvoGPU.cu:1562): only read once the pair is done
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) st->out_T[4 * j + i] = Ri[3 * i + j];
        st->out_T[12 + i] = Ti[i];
      }
      st->out_T[3] = st->out_T[7] = st->out_T[11] = 0;
      st->out_T[15] = 1;
    }
    if (clk0 && !dry) {  // CVO_KERNEL_CLOCK (k_coeff): this launch's interval and the association's, see PairState
      if (st->clk_last_assoc) {
        st->clk_sum[0] += st->clk_last_assoc;
        st->clk_n[0]++;
        st->clk_last_assoc = 0;
      }
      const unsigned dt_coeff = pair_clock_ticks(clk0);
      if (dt_coeff) {
        st->clk_sum[1] += dt_coeff;
        st->clk_n[1]++;
      }
    }
    if (done && !dry) {
      st->status = 1;
      *D.status_out = 1;
      *D.status_host = 1;
    }
  }
  CVO_UPD_STAMP(5);
  __syncthreads();
  if (act && !dry)
    for (int q = tid; q < HOT_DWORDS; q += 64) reinterpret_cast<unsigned*>(gst)[q] = s_hot[q];
  CVO_UPD_STAMP(6);
}

template <bool INIT>
__global__ __launch_bounds__(64) void k_update(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp,
                                               const int* __restrict__ status, int flags) {
  if (!INIT && status[blockIdx.x] != 0) return;
  const PairDesc* __restrict__ D = descs + blockIdx.x;
  if (!INIT && (flags & 1) && (D->st->rebuild || (D->st->n_ovf > 0 && !(flags & 32)))) return;  // lean graph: the pair is waiting (k_assoc)
  if (INIT && threadIdx.x == 0) {  // the pair's cross-block counters start at zero
    *D->gate = 0;
    *D->gate_flow = 0;
    *D->done = 0;
    *D->tile_count = 0ull;
  }
  if (INIT)  // the resident kernel's arrival counters and granules (tags of an earlier call must not match)
    for (int q = threadIdx.x; q < (int)(sizeof(ResidentSync) / 8); q += 64) reinterpret_cast<unsigned long long*>(D->rsync)[q] = 0ull;
  const DevParams P = *Pp;
  __shared__ UpdateShared U;
  update_body<INIT, false>(load_upd_desc(D), P, flags,
                           ((flags & 1) && !(flags & 32)) ? D->nblk_assoc
                                                          : D->nblk_assoc + (P.mode == 0 ? dense_parts(D->dense_blocks, D->st->n_ovf) : D->dense_blocks),
                           U, nullptr,
                           nullptr);
}

// ------------------------------------------------------------------------------------------
// k_coeff: coefficient phase + (align loop) the update.  The block of a pair that finishes last runs update_body:
// one launch less on the critical path of every iteration, and no block ever waits for another one.  Partials
// cross blocks inside the launch, hence the coherent stores / loads (st_x / ld_x).
// flags: bit 0 = lean graph, bit 5 = ... with k_assoc_dense in every iteration, the rest see update_body.
// ------------------------------------------------------------------------------------------
template <bool INSTR>
__global__ __launch_bounds__(ASSOC_THREADS, CVO_COEFF_WAVES) void k_coeff(const PairDesc* __restrict__ descs,
                                                         const DevParams* __restrict__ Pp, PairState* states,
                                                         const char* __restrict__ arena, int flags, int nblk_split_pairs,
                                                         unsigned stride256, int Npad) {
  const unsigned long long tt0 = INSTR ? __builtin_readcyclecounter() : 0ull;
  // grid: per pair nblk row blocks x launch_split slices of the ELL slots; a pair uses csplit <= launch_split of them
  const int nblk = nblk_split_pairs & 0x3fff, launch_split = (nblk_split_pairs >> 14) & 0x3f,
            n_pairs = (int)((unsigned)nblk_split_pairs >> 20);
  PairBlock pb;
  if (!pair_block(nblk * launch_split, n_pairs, pb)) return;
  const PairDesc* __restrict__ D = descs + pb.pair;
  const int cq = pb.bx % launch_split;
  pb.bx /= launch_split;
  // head of the row loop, from kernel-argument addresses (row_off_*): count, coordinates and the first ELL entry
  // of this block's slice - requested before the count is known, used only if it exists
  CoeffRowHead head;
  {
    const char* wb = arena + (size_t)pb.pair * ((size_t)stride256 << 8);
    const int pos = pb.bx * ASSOC_THREADS + threadIdx.x;  // < Npad; values of rows >= N are never used
    head.nnz = reinterpret_cast<const unsigned*>(wb + row_off_nnz(Npad))[pos];
    head.x = reinterpret_cast<const float4*>(wb + row_off_xp4(Npad))[pos];
    head.e_n = EllEntry{0.f, 0.f, 0.f, 0.f};
    if (cq == 0) head.e_n = reinterpret_cast<const EllEntry*>(wb + row_off_ell(Npad))[pos];
  }
  const int csplit_light = D->csplit, csplit_heavy = D->csplit_heavy;
  PairState* const st = states + pb.pair;  // == D->st, without the dependent pointer load
  // The state as this launch found it, through a read-only view so that the loads are scalar (only the block that
  // finishes last writes the state, after every block has read it); one burst together with what the row loop
  // needs first, see k_assoc.
  const PairState* __restrict__ st_in = states + pb.pair;
  const int status_v = st_in->status, rebuild_v = st_in->rebuild, ovf = st_in->n_ovf;
  const unsigned max_nnz_prev = st_in->max_nnz;  // longest row of the iteration before (this one's is not reduced yet)
  const DevParams P = *Pp;
  // the twist and its matrices (twist_finalize): wave-uniform scalar loads
  XiMats Mu;
  {
    float* mu = reinterpret_cast<float*>(&Mu);
#pragma unroll
    for (int q = 0; q < (int)(sizeof(XiMats) / sizeof(float)); q++) mu[q] = st_in->xi[q];
  }
  {
    // (everything the kernel will branch on or start its row loop with - the slice count, the parameters and the twist
    // matrices included - requested before the first wait: each dependent round of scalar loads is ~0.3-0.5 us here)
    const int n = D->N, nb = D->nblk_assoc, ep = st_in->epoch;
    const double* a0 = D->flow_part;
    const unsigned* a1 = D->nnz_row;
    const float4* a2 = D->xp4;
    const EllEntry* a3 = D->ell;
    const int a4 = D->M;
    const float e = st_in->ell;
    const int k_line = st_in->K;  // (rides in the 16-byte load of status / rebuild / n_ovf: pinned so that none of its
                                  // registers is dead and reused inside the burst, see k_assoc)
    asm volatile("" ::"s"(n), "s"(nb), "s"(ep), "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(e), "s"(csplit_light),
                 "s"(csplit_heavy), "s"(status_v), "s"(rebuild_v), "s"(ovf), "s"(k_line), "s"(max_nnz_prev), "s"(P.mode), "s"(P.sp_thres), "s"(P.use_range_ell),
                 "s"(Mu.omega[0]), "s"(Mu.m2.m[0][0]), "s"(Mu.m4.m[2][2]), "s"(Mu.v[2]));
    // (nothing computed from these values - the slice count below is the first - may be scheduled into the middle of
    // the burst, where it would need a wait of its own: one more round trip)
    __builtin_amdgcn_sched_barrier(0);
  }
  // rows with hundreds of nonzeros (a pair with overflow rows: clustered clouds, the K cap) are spread over more blocks -
  // while its rows really are that long (the longest row of the iteration before: the pair's own state, like n_ovf)
  // (both counts are requested in the burst above: a load that depends on the branch would be one more round trip)
  // (as many slices as keep a thread's share of the longest row at ~32 entries: every slice is four more partials for the
  // update to fetch, 128 per round trip)
  // ... and at least ~128 blocks per pair while rows are long
  int csplit = csplit_light;
  if (ovf > 0 && max_nnz_prev > 48u && (!(flags & 1) || (flags & 32)))
    while (csplit < csplit_heavy && ((unsigned)(csplit * 32) < max_nnz_prev || nblk * csplit < 128)) csplit <<= 1;
  if (cq >= csplit) return;
  const bool replay = (flags & 8) != 0;  // cvo_debug_time_kernels: same work, nothing written back
  if (!replay && status_v != 0) return;
  if (flags & 1) {
    if (rebuild_v || (ovf > 0 && !(flags & 32))) {  // waiting, see k_assoc; tell the host which graph this pair needs
      if (pb.bx == 0 && cq == 0 && threadIdx.x == 0) {
        st->n_stalls++;
        if (ovf > 0 && !(flags & 32)) {
          st->want_full = 4;
          *D->want_out = 4;
          *D->want_host = 4;
        }
      }
      return;
    }
  }
  if (P.mode != 0) return;
  pair_clock_begin(INSTR && P.kernel_clock && !replay && pb.bx == 0 && cq == 0, st, 1);
  const int epoch = st_in->epoch;  // launches of this kernel the pair has completed (bumped by the updating block)
  __shared__ union {
    CoeffShared c;
    UpdateShared u;
  } S;
  __shared__ int s_last;
  const int N_ = D->N, pos_ = pb.bx * ASSOC_THREADS + threadIdx.x;
  if (pos_ >= N_) head.nnz = 0;
  if (cq > 0 && (unsigned)cq < head.nnz) head.e_n = D->ell[(size_t)cq * N_ + pos_];  // (small clouds only: later slices)
  float twist[6];
  for (int c = 0; c < 3; c++) {
    twist[c] = Mu.omega[c];
    twist[3 + c] = Mu.v[c];
  }
  const unsigned long long tt1 = INSTR ? __builtin_readcyclecounter() : 0ull;
  coeff_rows<true>(P, D, st_in->ell, S.c, Mu, head, pb.bx, cq, csplit);
  const unsigned long long tt2 = INSTR ? __builtin_readcyclecounter() : 0ull;
  if (threadIdx.x >= 64) return;  // the counter and (in one block of the pair) the update are the first wave's, see k_assoc
  // the scalar state, for whichever block turns out to be the last one: in flight while the counter round trip runs
  unsigned hot_regs[2] = {0u, 0u};
  if (threadIdx.x < 64) {
    hot_regs[0] = reinterpret_cast<const unsigned*>(st)[threadIdx.x];
    if (threadIdx.x + 64 < HOT_DWORDS) hot_regs[1] = reinterpret_cast<const unsigned*>(st)[threadIdx.x + 64];
  }
  const unsigned long long clk0 = pair_clock_peek(INSTR && P.kernel_clock && !replay, st, 1);
  UpdDesc upd = load_upd_desc(D);
  upd.nblk_coeff = nblk * csplit;
  const int n_flow_upd = (((flags & 1) && !(flags & 32)) || ovf == 0) ? D->nblk_assoc : D->nblk_assoc + dense_parts(D->dense_blocks, ovf);  // (see k_assoc_dense)
  // (how many flow partials the update will sum: known now - left to the compiler, the two descriptor words behind it are
  // requested after the counter's round trip, one more dependent wait on the pair's serial tail)
  asm volatile("" ::"s"(n_flow_upd));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // partial stores before the counter, see flow_gate
  __syncthreads();
  if (threadIdx.x == 0) {
    // the counter advances by nblk * COEFF_SPLIT_MAX per iteration whatever the split of the iteration is (splits are
    // powers of two): each of the nblk * csplit blocks that store a partial adds its share
    const unsigned share = (unsigned)COEFF_SPLIT_MAX >> __builtin_ctz((unsigned)csplit), per_it = (unsigned)(nblk * COEFF_SPLIT_MAX);
    const unsigned done = (unsigned)__hip_atomic_fetch_add(D->done, (int)share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + share;
    s_last = replay ? (done % per_it == 0u) : (done == (unsigned)(epoch + 1) * per_it);
  }
  __syncthreads();
  const unsigned long long tt3 = INSTR ? __builtin_readcyclecounter() : 0ull;
  if (INSTR && P.phase_ticks && threadIdx.x == 0) {
    g_phase_ticks[1][blockIdx.x & 4095][0] = tt0;
    g_phase_ticks[1][blockIdx.x & 4095][1] = tt1;
    g_phase_ticks[1][blockIdx.x & 4095][2] = tt2;
    g_phase_ticks[1][blockIdx.x & 4095][3] = tt3;
  }
  if (!s_last || (flags & 16)) return;  // (bit 4: cost breakdown of cvo_debug_time_kernels, coefficient phase only)
  update_body<false, true>(upd, P, flags | 4, n_flow_upd, S.u, twist, hot_regs, clk0);
  if (INSTR && P.phase_ticks && threadIdx.x == 0) {
    g_phase_ticks[1][4096 + pb.pair][0] = tt0;
    g_phase_ticks[1][4096 + pb.pair][1] = tt3;
    g_phase_ticks[1][4096 + pb.pair][2] = __builtin_readcyclecounter();
  }
}

// ------------------------------------------------------------------------------------------
// k_resident (compiled only with -DCVO_WITH_RESIDENT; slower than the two launches on every configuration, kept as
// the harness of the in-launch experiments - ROUND_LOG.md round 3): up to U optimiser iterations of every pair of a
// (small) sub-batch in ONE launch - the lean iterations
// between two rebuild opportunities, which the two-kernel path runs as U x [k_assoc, k_coeff].  For calls with few
// pairs in flight (one frame pair at a time is the reference's own use: frame-to-frame tracking), where an iteration
// is a chain of latencies - two launch gaps, two cold prologues, two last-block elections - and not throughput.
//
// The two earlier in-launch attempts (ROUND_LOG.md: k_persist, round 2's k_resident) exchanged partials, twist and state
// with sc1 accesses - placement-independent, hence served behind the L2 - and lost to the two launches they replaced.
// Here co-location is CREATED instead of hoped for: a block reads the XCD it actually runs on (HW_REG_XCC_ID), draws its
// index among the blocks of that XCD from a per-XCD counter and takes the role (pair, block-of-pair) that index stands
// for; pairs are bound to XCDs (pair p <-> XCD p % 8), so all blocks of a pair share one L2 BY CONSTRUCTION and the
// exchange goes through it: plain stores (the line stays in that L2), relaxed atomics for arrival, L1-bypassing (sc1)
// loads.  scripts/ubench/xcd_exchange.hip: one hop 260 ns (342 with sc1 stores), reduce + broadcast among 32 blocks of
// an XCD 2.0 us against 3.7 us, and such a value is NEVER seen from another XCD.  No kernel boundary separates the
// iterations, so the L2 stays valid: row heads, candidate lists, targets and the ELL entries are L2 / L1 hits.
//
// Roles of a pair's NB + 1 blocks:
//   row blocks 0 .. NB-1  serve the row blocks role, role + NB, ... of BOTH passes (an ELL entry is read back by the thread
//                         that wrote it); partials land in the same per-row-block slots as in the two-kernel path.
//   tail block NB         waits for the row blocks' arrivals, reduces the partials with the same code in the same order
//                         as the two-kernel path (results are bit-identical to it) and runs the reference's host-side
//                         scalar code: twist_finalize after pass 1, update_body after pass 2.  It has no rows, so what
//                         the next iteration needs (stop word, K, ell, Rinv, Tinv) is PUBLISHED AS SOON AS IT IS KNOWN
//                         and the list bookkeeping, skins and the write-back of the state run while the row blocks are
//                         already in the next association.
// Broadcasts (head, twist matrices) are data-tagged 8-byte granules: one hop, no flag, no store drain.  Pairs advance
// independently; a pair whose list expires (or that finishes) leaves the launch.  Every wait is bounded: a timeout
// marks the pair (ResidentSync::abort, want = 3) and the host falls back to the two-kernel graphs.
//
// Residency: all blocks of a launch must be co-resident (they wait for each other): the host keeps the launches of all
// sub-batch streams together at or below 1.5 blocks of 256 threads per CU; __launch_bounds__(256, 2) guarantees 2 (the
// whole register file for two blocks: rows, twist and update code share one allocation without spilling).
// ------------------------------------------------------------------------------------------
#ifdef CVO_WITH_RESIDENT  // opt-in build (unified_cvo_amd/build.py: build_resident): not part of the default library
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

// CVO_PHASE_TICKS=1: where an iteration of the resident kernel goes, in ticks of the 100 MHz s_memrealtime counter, summed
// over the iterations of every pair's row block 0 (thread 0): [0] wait for the head, [1] pass-1 rows, [2] arrive A,
// [3] wait for the twist, [4] pass-2 rows, [5] arrive B, [8] iterations counted; by the tail blocks: [9] wait for
// arrivals A, [10] twist_finalize, [11] wait for arrivals B, [12] update until the head is published, [13] rest of the
// update, [14] iterations.  Read by cvo_debug_resident_ticks.
__device__ unsigned long long g_res_ticks[16];

struct ResidentShared {
  union {
    AssocShared a;
    CoeffShared c;
    UpdateShared u;
  };
  unsigned view[64];  // granule values / state dwords as the block's first wave received them
  int slot, go;
  unsigned xcc;
};

struct ResidentHead {
  int stop;  // 0 go on, 1 finished, 2 list expired, 3 overflow rows
  IterView iv;
};

// The tail block's early publication of the next iteration's head (see update_body).
struct ResidentPublish {
  unsigned long long* head;
  unsigned tag;
  unsigned long long* t_pub;  // CVO_PHASE_TICKS: when the head left
  __device__ __forceinline__ void put(int q, unsigned bits) const { head[q] = ((unsigned long long)tag << 32) | (unsigned long long)bits; }
  __device__ __forceinline__ void operator()(int stop, int K, float ell, const float* Ri, const float* Ti) const {
    put(RES_HEAD_STOP, (unsigned)stop);
    put(RES_HEAD_K, (unsigned)K);
    put(RES_HEAD_ELL, __float_as_uint(ell));
#pragma unroll
    for (int q = 0; q < 9; q++) put(RES_HEAD_RINV + q, __float_as_uint(Ri[q]));
#pragma unroll
    for (int q = 0; q < 3; q++) put(RES_HEAD_TINV + q, __float_as_uint(Ti[q]));
    if (t_pub) *t_pub = (unsigned long long)__builtin_amdgcn_s_memrealtime();
  }
};

template <typename IdxT, int ASSOC_CAP, bool GENERAL>
__global__ __launch_bounds__(ASSOC_THREADS, 2) void k_resident(const PairDesc* __restrict__ descs,
                                                               const DevParams* __restrict__ Pp, PairState* states,
                                                               const char* __restrict__ arena, ResidentTeams* teams,
                                                               int U_NB_pairs, int nblk_split, unsigned stride256, int Npad) {
  static_assert(ASSOC_THREADS == 256, "roles are blocks of four waves");
  __shared__ ResidentShared S;
  const int U = U_NB_pairs & 0xff, NB = (U_NB_pairs >> 8) & 0xfff, n_pairs = (int)((unsigned)U_NB_pairs >> 20);
  const int nblk = nblk_split & 0x3fff;  // (bits 14..19: the launch's coefficient split, 1 here)
  const int xoff = (nblk_split >> 20) & 7;  // pair q of this launch lives on XCD (q + xoff) % 8: sub-batches spread over the chip
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // ---- self-placement
  if (tid == 0) {
    const unsigned x = xcc_id();
    S.xcc = x;
    S.slot = (int)__hip_atomic_fetch_add(&teams->joined[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int xcc = (int)S.xcc, slot = S.slot;
  const int pair = (slot / (NB + 1)) * 8 + ((xcc - xoff) & 7), role = slot % (NB + 1);
  const bool is_tail = role == NB;
  if (pair < n_pairs) {
    // (descriptor and parameters through the kernel's __restrict__ arguments, the parameters by value: a laundered or
    // local pointer makes every field a may-alias of the ELL stores and puts scalar re-loads into the row loops)
    const PairDesc* __restrict__ D = descs + pair;
    const PairDesc* D0 = D;
    const DevParams P = *Pp;
    PairState* const st = states + pair;
    ResidentSync* const sy = D0->rsync;
    const char* wb = arena + (size_t)pair * ((size_t)stride256 << 8);
    bool aborted = false;
    const bool ticks = P.phase_ticks != 0 && tid == 0 && (role == 0 || is_tail);
    auto now = [] { return (unsigned long long)__builtin_amdgcn_s_memrealtime(); };
    auto tick = [&](int slot_, unsigned long long& t) {
      if (ticks) {
        const unsigned long long t1 = now();
        atomicAdd(&g_res_ticks[slot_], t1 - t);
        t = t1;
      }
    };
    // Waits until the first `n` granules of `g` carry `tag`, then leaves their values in S.view (every thread calls it;
    // false = the launch is aborted).  One 8-byte load per lane and poll; granules need no ordering among themselves.
    auto wait_granules = [&](const unsigned long long* g, int n, unsigned tag) {
      if (wave == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        for (;;) {
          const unsigned long long v = lane < n ? ld_x<true>(g + lane) : ((unsigned long long)tag << 32);
          if (__ballot((unsigned)(v >> 32) != tag) == 0ull) {
            if (lane < n) S.view[lane] = (unsigned)v;
            break;
          }
          if (ld_x<true>(&sy->abort) != 0u) {
            ok = 0;
            break;
          }
          if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)RESIDENT_TIMEOUT_TICKS) {
            if (lane == 0) __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) S.go = ok;
      }
      __syncthreads();
      const bool ok = S.go != 0;
      return ok;
    };
    // row blocks: this block's partials are stored (only the first wave stores partials; every wave's ELL entries are read
    // back by the thread that wrote them)
    auto arrive = [&](unsigned* ctr) {
      if (wave == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();  // (also: the reduction's LDS is free again)
      if (tid == 0) (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // nobody waits for the result
    };
    // tail block: all NB row blocks have arrived (the counter is reset for the next iteration)
    auto wait_arrivals = [&](unsigned* ctr) {
      if (tid == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        while (ld_x<true>(ctr) != (unsigned)NB) {
          if (ld_x<true>(&sy->abort) != 0u) {
            ok = 0;
            break;
          }
          if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)RESIDENT_TIMEOUT_TICKS) {
            __hip_atomic_store(&sy->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = 0;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (ok) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        S.go = ok;
      }
      __syncthreads();
      const bool ok = S.go != 0;
      __syncthreads();
      return ok;
    };
    if (ld_x<true>(&sy->abort) != 0u) aborted = true;  // (sticky: an aborted pair is left to the two-kernel path)
    // ---- the state as the previous launch left it: head of iteration k0
    int k0 = 0;
    ResidentHead h;
    {
      if (wave == 0) S.view[lane] = ld_x<true>(reinterpret_cast<const unsigned*>(st) + lane);
      __syncthreads();
      const unsigned* v = S.view;
      auto ui = [&](int q) { return __builtin_amdgcn_readfirstlane((int)v[q]); };
      auto uf = [&](int q) { return __int_as_float(__builtin_amdgcn_readfirstlane((int)v[q])); };
      static_assert(offsetof(PairState, status) == 0 && offsetof(PairState, rebuild) == 4 && offsetof(PairState, n_ovf) == 8 &&
                        offsetof(PairState, K) == 12 && offsetof(PairState, ell) == 16 && offsetof(PairState, k) == 24 &&
                        offsetof(PairState, Rinv) == 32 && offsetof(PairState, Tinv) == 68,
                    "head of PairState");
      h.stop = ui(0) != 0 ? 1 : (ui(1) != 0 ? 2 : (ui(2) > 0 ? 3 : 0));
      h.iv.K = ui(3);
      h.iv.ell = uf(4);
      h.iv.row_max = ld_x<true>(&st->row_max);  // (changes only with a rebuild: never inside this launch)
      k0 = ui(6);
#pragma unroll
      for (int q = 0; q < 9; q++) h.iv.pose.Ri[q] = uf(8 + q);
#pragma unroll
      for (int q = 0; q < 3; q++) h.iv.pose.Ti[q] = uf(17 + q);
      __syncthreads();
    }
    for (int u = 0; u < U && !aborted; u++) {
      unsigned long long tk = ticks ? now() : 0ull;
      const unsigned tag_head = 2u * (unsigned)(k0 + u), tag_xi = tag_head + 1u;
      if (u > 0) {  // the head the tail block published in the middle of its update
        if (!wait_granules(sy->head, RES_HEAD_WORDS, tag_head)) {
          aborted = true;
          break;
        }
        const unsigned* v = S.view;
        auto ui = [&](int q) { return __builtin_amdgcn_readfirstlane((int)v[q]); };
        auto uf = [&](int q) { return __int_as_float(__builtin_amdgcn_readfirstlane((int)v[q])); };
        h.stop = ui(RES_HEAD_STOP);
        h.iv.K = ui(RES_HEAD_K);
        h.iv.ell = uf(RES_HEAD_ELL);
#pragma unroll
        for (int q = 0; q < 9; q++) h.iv.pose.Ri[q] = uf(RES_HEAD_RINV + q);
#pragma unroll
        for (int q = 0; q < 3; q++) h.iv.pose.Ti[q] = uf(RES_HEAD_TINV + q);
        __syncthreads();  // (S.view is reused)
      }
      if (!is_tail) tick(0, tk);
      if (h.stop == 1) break;
      if (h.stop != 0) {  // waits for a rebuild opportunity / the full graph: tell the host (as lean k_coeff does)
        if (is_tail && tid == 0) {
          // (the update of the previous iteration, if it ran in this launch, is this block's own: program order)
          st->n_stalls += U - u;
          if (h.stop == 3) {
            st->want_full = 4;
            *D->want_out = 4;
            *D->want_host = 4;
          }
        }
        break;
      }
      if (!is_tail) {
        // ---- pass 1: association + flow over this block's row blocks
        for (int rb = role; rb < nblk; rb += NB) {
          AssocRowHead head;
          const int pos = rb * ASSOC_THREADS + tid;  // < Npad
          head.ip = GENERAL ? reinterpret_cast<const int*>(wb + row_off_ip(Npad))[pos] : 0;
          head.j1 = (int)reinterpret_cast<const IdxT*>(wb + row_off_cand_j(Npad))[pos];
          head.x = reinterpret_cast<const float4*>(wb + row_off_xp4(Npad))[pos];
          head.cnt = __float_as_int(head.x.w);
          if (rb != role) __syncthreads();  // the reduction's LDS is reused
          assoc_phase<IdxT, ASSOC_CAP, GENERAL, false, true>(P, D, h.iv, S.a, rb, head);
        }
        tick(1, tk);
        arrive(&sy->arrive_a);
        tick(2, tk);
        // ---- the twist and its matrices, from the tail block
        if (!wait_granules(sy->xi, (int)(sizeof(XiMats) / sizeof(float)), tag_xi)) {
          aborted = true;
          break;
        }
        XiMats Mu;
        {
          float* mu = reinterpret_cast<float*>(&Mu);
#pragma unroll
          for (int q = 0; q < (int)(sizeof(XiMats) / sizeof(float)); q++)
            mu[q] = __int_as_float(__builtin_amdgcn_readfirstlane((int)S.view[q]));
        }
        __syncthreads();
        tick(3, tk);
        // ---- pass 2: coefficients
        for (int rb = role; rb < nblk; rb += NB) {
          const int pos = rb * ASSOC_THREADS + tid;
          CoeffRowHead ch;
          ch.nnz = pos < D->N ? D->nnz_row[pos] : 0u;  // (the base pass 1 stored through: no second, restrict-qualified view)
          ch.x = reinterpret_cast<const float4*>(wb + row_off_xp4(Npad))[pos];
          ch.e_n = EllEntry{0.f, 0.f, 0.f, 0.f};
          if (ch.nnz > 0u) ch.e_n = D->ell[pos];
          if (rb != role) __syncthreads();
          coeff_rows<false>(P, D, h.iv.ell, S.c, Mu, ch, rb, 0, 1);
        }
        tick(4, tk);
        arrive(&sy->arrive_b);
        tick(5, tk);
        if (ticks) atomicAdd(&g_res_ticks[8], 1ull);
      } else {
        // ---- tail block: twist after pass 1 ...
        if (!wait_arrivals(&sy->arrive_a)) {
          aborted = true;
          break;
        }
        tick(9, tk);
        float twist[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // (only thread 0 of the update reads it)
        if (wave == 0) twist_finalize<true>(D, nblk, sy->xi, tag_xi, twist);  // partials: sc1 loads, served by this XCD's L2
        tick(10, tk);
        // ... the scalar state in flight while pass 2 runs (the previous update's write-back is this block's own)
        unsigned hot_regs[2] = {0u, 0u};
        if (tid < 64) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          hot_regs[0] = ld_x<true>(reinterpret_cast<const unsigned*>(st) + tid);
          if (tid + 64 < HOT_DWORDS) hot_regs[1] = ld_x<true>(reinterpret_cast<const unsigned*>(st) + tid + 64);
        }
        // ... then the update after pass 2
        if (!wait_arrivals(&sy->arrive_b)) {
          aborted = true;
          break;
        }
        tick(11, tk);
        const UpdDesc upd = load_upd_desc(D);
        const int flags = ((u == U - 1) ? 2 : 0) | (U << 8) | 1;
        __shared__ unsigned long long s_tpub;
        if (tid == 0) s_tpub = 0ull;
        ResidentPublish pub{sy->head, tag_head + 2u, ticks ? &s_tpub : nullptr};
        update_body<false, true, true, ResidentPublish>(upd, P, flags, nblk, S.u, twist, hot_regs, 0ull, pub);
        if (tid == 0) st->res_iters += 1u;
        if (ticks) {
          const unsigned long long t1 = now();
          const unsigned long long tp = s_tpub ? s_tpub : t1;
          atomicAdd(&g_res_ticks[12], tp - tk);
          atomicAdd(&g_res_ticks[13], t1 - tp);
          atomicAdd(&g_res_ticks[14], 1ull);
        }
        __syncthreads();
      }
    }
    // tell the host: this pair must be served by the two-kernel graphs.  Published by the tail block AND by the
    // pair's first row block: if the dispatcher did not deal this pair a tail block (self-placement assumes it hands
    // every XCD ppx * (NB + 1) blocks), the row blocks still time out, and one of them must say so.
    if (aborted && (is_tail || role == 0) && tid == 0) {
      if (is_tail) st->want_full = 3;
      *D0->want_out = 3;
      *D0->want_host = 3;
    }
  }
  // ---- leave: the last block of the launch resets the placement counters for the next launch of this stream
  __syncthreads();
  if (tid == 0) {
    const unsigned l = __hip_atomic_fetch_add(&teams->left, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (l == gridDim.x - 1) {
#pragma unroll
      for (int x = 0; x < 8; x++) __hip_atomic_store(&teams->joined[x], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&teams->left, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

#endif  // CVO_WITH_RESIDENT

// ------------------------------------------------------------------------------------------
// k_verify (CVO_VERIFY_LISTS=1): the self-check of the candidate-list reuse.  After the association of an iteration
// (k_assoc over the cached lists [+ k_assoc_dense]) one wave per row re-derives the row with the reference's literal
// ordered scan over ALL targets (CvoGPU.cu:522-590) at the pose / ell / K of that iteration and compares it with the
// row the lists produced: nonzero count, every column, every value bit for bit.  A list that had lost a pair - a skin
// too small for the motion since the build, a cull that was not conservative - shows up as a missing or shifted entry.
// The first mismatch of a pair is latched in its state (sticky) and turns the call's return code into CVO_E_VERIFY.
// Independent of the oracle and of the clouds' size: the tests run it at 10k x 10k over the fast-moving first iterations.
// ------------------------------------------------------------------------------------------
template <bool GENERAL>
__global__ __launch_bounds__(256) void k_verify(const PairDesc* __restrict__ descs, const DevParams* __restrict__ Pp,
                                                const int* __restrict__ status, int lean) {
  if (status[blockIdx.y] != 0) return;
  const PairDesc* __restrict__ D = descs + blockIdx.y;
  PairState* st = D->st;
  if ((lean & 1) && (st->rebuild || (st->n_ovf > 0 && !(lean & 4)))) return;  // the pair did not advance in this slot (see k_assoc)
  const DevParams P = *Pp;
  const int N = D->N, M = D->M, K = st->K;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const Pose pose = load_pose(st);
  const FeatDen F = make_feat_den(P);
  unsigned checked = 0;
  for (int pos = blockIdx.x * 4 + wave; pos < N; pos += gridDim.x * 4) {
    const int i = D->ip[pos];
    const float4 x = D->xp4[pos];
    const RowData r = make_row(P, x, st->ell);
    unsigned nnz = 0;
    int err = 0;
    for (int j0 = 0; j0 < M && nnz < (unsigned)K; j0 += 64) {
      const int j = j0 + lane;
      float a = 0.f;
      float4 yt;
      bool ok = false;
      if (j < M) ok = eval_pair<GENERAL>(P, D, F, pose, i, r, GENERAL ? D->yinv[j] : 0, D->y4[j], a, yt) && (a > P.sp_thres);
      const unsigned long long m = __ballot(ok);
      const unsigned rank = nnz + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      const bool keep = ok && rank < (unsigned)K;
      if (keep) {
        const EllEntry e = D->ell[(size_t)rank * N + pos];
        if (D->ell_j[(size_t)rank * N + pos] != j)
          err = 2;
        else if (__float_as_uint(e.a) != __float_as_uint(a) || __float_as_uint(e.yx) != __float_as_uint(yt.x) ||
                 __float_as_uint(e.yy) != __float_as_uint(yt.y) || __float_as_uint(e.yz) != __float_as_uint(yt.z))
          err = 3;
      }
      nnz += (unsigned)__builtin_popcountll(__ballot(keep));
    }
    if (D->nnz_row[pos] != nnz) err = 1;  // (also catches entries the list path has and the scan does not)
    if (__ballot(err != 0) != 0ull) {
      int e = err;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) e = max(e, __shfl_xor(e, o));
      if (lane == 0 && atomicCAS(&st->verify_err, 0, 1) == 0) {
        st->verify_k = st->k;
        st->verify_pos = pos;
        st->verify_what = (D->nnz_row[pos] != nnz) ? 1 : e;
      }
    }
    checked++;
  }
  if (lane == 0 && checked) atomicAdd(&st->verify_rows, (unsigned long long)checked);
}

// ------------------------------------------------------------------------------------------
// k_scalar_math (cvo_debug_scalar_math): the device's scalar restatements of the reference's host-side maths, run
// on caller-supplied inputs so that the tests can pin THE DEVICE CODE ITSELF against numpy / scipy (the oracle
// carries the same text for some of them, so "GPU == oracle" alone only shows that two compilers agree).
// One wave per item; item q reads in[16 q ..] and writes out[16 q ..].
//   op 0  cubic_roots            in: p0..p3                      out: re[3], im[3]
//   op 1  cubic_roots_wave       (the three-lane search used by the update)  same layout
//   op 2  select_step<false>     in: B, C, D, E, min_step, max_step          out: step
//   op 3  select_step<true>      same
//   op 4  exp_sek3               in: xi[6], dt                   out: 3x4 row-major
//   op 5  se3_log_norm           in: R[9] row-major, t[3]        out: norm
//   op 6  update_tf              in: R[9], T[3]                  out: Rinv[9], Tinv[3]
//   op 7  indicator windows      ONE item: in = {window, threshold, x_0 .. x_{n-1}}, out[k] = decision of sample k
//                                (indicator_update on a scratch PairState, exactly as the update calls it)
// The hoisted arithmetic of the row loops against the compiler's / the device library's own forms (eight operands per
// item, lane l takes operand l; out[2 l] = the plain form, out[2 l + 1] = the hoisted form - the tests compare the BITS):
//   op 8  n / d  vs  div_by(n, d, rcp_refined(d))            in: {n_l, d_l} pairs (in[2 l], in[2 l + 1])
//   op 9  x / 6.0  vs  div_by(x, 6.0, rcp_refined(6.0))      in: x_l (in[l])
//   op 10 exp(x)  vs  exp_ocml<false>(x)                     in: x_l
//   op 11 exp(x)  vs  exp_ocml<true>(x)   (x <= 0)           in: x_l
//   op 12 (float) n / d  vs  fdiv_hoisted(n, fdiv_prepare(d)) in: {n_l, d_l} pairs; out[2 l + 1] = NaN-boxed -1 (as a
//         double: -1.0) where fdiv_operands_safe refuses the operand (the kernel then divides the plain way)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_scalar_math(int op, int n, const double* __restrict__ in, double* __restrict__ out,
                                                    PairState* scratch) {
  const int lane = threadIdx.x;
  if (op == 7) {
    if (blockIdx.x != 0 || lane != 0) return;
    const int window = (int)in[0];
    const float thr = (float)in[1];
    for (int k = 0; k < n; k++) {
      const float e_front = scratch->eq[scratch->e_head], s_front = scratch->sq[scratch->s_head];
      out[k] = indicator_update(scratch, scratch->sq, scratch->eq, (float)in[2 + k], window, thr, e_front, s_front) ? 1.0 : 0.0;
    }
    return;
  }
  const double* a = in + 16 * (size_t)blockIdx.x;
  double* o = out + 16 * (size_t)blockIdx.x;
  if (op == 12) {
    if (lane >= 8) return;
    const float nn = (float)a[2 * lane], dd = (float)a[2 * lane + 1];
    const float six[6] = {nn, 0.f, 0.f, 0.f, 0.f, 0.f};
    o[2 * lane] = (double)(nn / dd);
    o[2 * lane + 1] = fdiv_operands_safe(six) ? (double)fdiv_hoisted(nn, fdiv_prepare(dd)) : -1.0;
    return;
  }
  if (op >= 8 && op <= 11) {
    if (lane >= 8) return;
    double plain, hoisted;
    if (op == 8) {
      const double nn = a[2 * lane], dd = a[2 * lane + 1];
      plain = nn / dd;
      hoisted = div_by(nn, dd, rcp_refined(dd));
    } else if (op == 9) {
      const double x = a[lane];
      plain = x / 6.0;
      hoisted = div_by(x, 6.0, rcp_refined(6.0));
    } else {
      const double x = a[lane];
      const ExpConsts ek = make_exp_consts();
      plain = exp(x);
      hoisted = op == 10 ? exp_ocml<false>(x, ek) : exp_ocml<true>(x, ek);
    }
    o[2 * lane] = plain;
    o[2 * lane + 1] = hoisted;
    return;
  }
  if (op == 0 || op == 1) {
    const double coef[4] = {a[0], a[1], a[2], a[3]};
    double re[3], im[3];
    if (op == 0)
      cubic_roots(coef, re, im);
    else
      cubic_roots_wave(coef, re, im);
    if (lane == 0)
      for (int q = 0; q < 3; q++) {
        o[q] = re[q];
        o[3 + q] = im[q];
      }
  } else if (op == 2 || op == 3) {
    const float st = op == 2 ? select_step<false>(a[0], a[1], a[2], a[3], (float)a[4], (float)a[5])
                             : select_step<true>(a[0], a[1], a[2], a[3], (float)a[4], (float)a[5]);
    if (lane == 0) o[0] = (double)st;
  } else if (op == 4) {
    float xi[6], dt = (float)a[6], res[12];
    for (int q = 0; q < 6; q++) xi[q] = (float)a[q];
    exp_sek3(xi, dt, res);
    if (lane == 0)
      for (int q = 0; q < 12; q++) o[q] = (double)res[q];
  } else if (op == 5) {
    double R[9], t[3];
    for (int q = 0; q < 9; q++) R[q] = a[q];
    for (int q = 0; q < 3; q++) t[q] = a[9 + q];
    const double v = se3_log_norm(R, t);
    if (lane == 0) o[0] = v;
  } else if (op == 6) {
    float R[9], T[3], Ri[9], Ti[3];
    for (int q = 0; q < 9; q++) R[q] = (float)a[q];
    for (int q = 0; q < 3; q++) T[q] = (float)a[9 + q];
    update_tf(R, T, Ri, Ti);
    if (lane == 0) {
      for (int q = 0; q < 9; q++) o[q] = (double)Ri[q];
      for (int q = 0; q < 3; q++) o[9 + q] = (double)Ti[q];
    }
  }
}

// Waits for `ticks` of the s_memrealtime counter (cvo_debug_kernel_clock calibrates the counter's rate with it).
__global__ void k_hold(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

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

// ------------------------------------------------------------------------------------------
// k_kd_order: the spatial (k-d) ordering of a cloud, on the device - what cvo_cloud_upload used to do on the calling
// thread with std::nth_element (1.2 ms of host CPU per 10k cloud: at 8 ranks on a 16-CPU box the upload pipeline of a
// 64-pair batch needed more cores than a rank has).  One block per cloud, any number of clouds per launch.
//
// The ordering is the one spatial_order() (cvo_hip.hip) defines: segments are halved recursively at a multiple of
// 512 / 64 / 4 points (so every aligned run of 512, 64 or 4 sorted points is a compact box), along the axis of the
// largest extent - here the extent of the ROOT box halved once per split along that axis, i.e. one axis per LEVEL
// (measured against per-segment boxes on the host: +0.3 % on the headline batch, nothing on the demo pair and config 3;
// no result depends on the ordering at all, tests/test_gpu_parity.py).  Level by level: every position p carries the
// 64-bit key (segment << 48 | ordered coordinate << 16 | point), one bitonic sort of the whole array in LDS puts every
// segment in coordinate order (a segment never leaves its range of positions: the segment number is the key's top),
// the split positions follow from the segment sizes alone, a block-wide prefix sum renumbers the segments.  Segments
// that are done (<= 4 points) keep their order (their key's coordinate field is the position).  12-14 levels for
// 16k points; up to KD_MAX_POINTS points per cloud (128 KB of keys in LDS), larger clouds are ordered on the host.
// Then the kernel writes order / inverse / the sorted coordinates and gathers the attribute arrays the caller
// supplied into spatial order (colour 5 -> 8 floats, classes 19 -> 20, geometric type 2).
// ------------------------------------------------------------------------------------------
constexpr int KD_THREADS = 1024;
constexpr int KD_MAX_POINTS = 16384;
constexpr int KD_MAX_SEGS = KD_MAX_POINTS / 2 + 2;
struct KdJob {
  int n, NP;                 // points, next power of two >= n (>= 2 * KD_THREADS / ... see launch)
  const float4* x4;          // coordinates, ORIGINAL order (already on the device)
  unsigned short* seg_of_pos;  // [NP] scratch
  unsigned short* seg_lo;      // [2][KD_MAX_SEGS] scratch: first position of every segment, + one sentinel
  int* order;                // out: sorted position -> original index
  int* inv;                  // out: original index -> sorted position
  float4* xs4;               // out: coordinates in spatial order
  const float* raw_feat;     // n x FD, original order, or null
  float4* feat;              // out: n x FD_PAD
  const float* raw_label;    // n x NC
  float4* label;             // out: n x NC_PAD
  const float* raw_geo;      // n x 2
  float2* geo;
};

__device__ __forceinline__ unsigned kd_ordered(float v) {
  const unsigned b = __float_as_uint(v);
  return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
}
// split of a segment of nn points (spatial_order / kd_split in cvo_hip.hip): 0 = the segment is done
__device__ __forceinline__ int kd_left(int nn) {
  if (nn <= 4) return 0;
  const int unit = nn > 512 ? 512 : (nn > 64 ? 64 : 4);
  int left = ((nn / 2 + unit - 1) / unit) * unit;
  if (left >= nn) left -= unit;
  return left > 0 ? left : 0;
}

__global__ __launch_bounds__(KD_THREADS) void k_kd_order(const KdJob* __restrict__ jobs) {
  extern __shared__ unsigned long long kd_key[];  // [NP]
  __shared__ float s_red[KD_THREADS / 64][6];
  __shared__ int s_scan[KD_THREADS / 64];
  __shared__ int s_total;
  const KdJob J = jobs[blockIdx.x];
  const int n = J.n, NP = J.NP, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- root box
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
  for (int p = tid; p < n; p += KD_THREADS) {
    const float4 x = J.x4[p];
    lo[0] = fminf(lo[0], x.x); hi[0] = fmaxf(hi[0], x.x);
    lo[1] = fminf(lo[1], x.y); hi[1] = fmaxf(hi[1], x.y);
    lo[2] = fminf(lo[2], x.z); hi[2] = fmaxf(hi[2], x.z);
  }
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[c] = fminf(lo[c], __shfl_xor(lo[c], o));
      hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o));
    }
  if (lane == 0)
    for (int c = 0; c < 3; c++) {
      s_red[wave][c] = lo[c];
      s_red[wave][3 + c] = hi[c];
    }
  __syncthreads();
  float ext[3];
  for (int c = 0; c < 3; c++) {
    float a = s_red[0][c], b = s_red[0][3 + c];
    for (int w = 1; w < KD_THREADS / 64; w++) {
      a = fminf(a, s_red[w][c]);
      b = fmaxf(b, s_red[w][3 + c]);
    }
    ext[c] = b - a;
  }
  // ---- one segment: all points in their original order
  const int per = NP / KD_THREADS;  // consecutive positions per thread in the renumbering pass (NP >= KD_THREADS)
  for (int p = tid; p < NP; p += KD_THREADS) {
    kd_key[p] = p < n ? (unsigned long long)p : ~0ull;
    J.seg_of_pos[p] = p < n ? (unsigned short)0 : (unsigned short)0xffff;
  }
  if (tid == 0) {
    J.seg_lo[0] = 0;
    J.seg_lo[1] = (unsigned short)n;  // (n <= 16384 < 65536)
  }
  __syncthreads();
  int cur = 0, nseg = 1;
  for (int level = 0; level < 24; level++) {
    int axis = 0;
    if (ext[1] > ext[axis]) axis = 1;
    if (ext[2] > ext[axis]) axis = 2;
    const unsigned short* slo = J.seg_lo + cur * KD_MAX_SEGS;
    // ---- keys of this level
    for (int p = tid; p < n; p += KD_THREADS) {
      const unsigned id = (unsigned)(kd_key[p] & 0xffffull);
      const unsigned s = J.seg_of_pos[p];
      const int l0 = slo[s], nn = (int)slo[s + 1] - l0;
      unsigned coord = (unsigned)p;  // a finished segment keeps its order
      if (kd_left(nn) > 0) {
        const float4 x = J.x4[id];
        coord = kd_ordered(axis == 0 ? x.x : (axis == 1 ? x.y : x.z));
      }
      kd_key[p] = ((unsigned long long)s << 48) | ((unsigned long long)coord << 16) | id;
    }
    __syncthreads();
    // ---- bitonic sort of the NP keys (pads are ~0: they stay at the end)
    for (int k = 2; k <= NP; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < NP / 2; t += KD_THREADS) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
          const unsigned long long a = kd_key[i], b = kd_key[l];
          const bool asc = (i & k) == 0;
          if ((a > b) == asc) {
            kd_key[i] = b;
            kd_key[l] = a;
          }
        }
        __syncthreads();
      }
    // ---- new segments: a position starts one if it is the first of its segment or the split position of it
    const int p0 = tid * per;
    int cnt = 0;
    unsigned flags = 0;  // per <= 16 positions per thread
    for (int q = 0; q < per; q++) {
      const int p = p0 + q;
      if (p < n) {
        const unsigned s = J.seg_of_pos[p];
        const int l0 = slo[s], nn = (int)slo[s + 1] - l0;
        const int left = kd_left(nn);
        if (p == l0 || (left > 0 && p == l0 + left)) {
          flags |= 1u << q;
          cnt++;
        }
      }
    }
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    int base = incl - cnt;
    for (int w = 0; w < wave; w++) base += s_scan[w];
    if (tid == KD_THREADS - 1) s_total = base + cnt;
    __syncthreads();
    const int total = s_total;
    unsigned short* nlo = J.seg_lo + (cur ^ 1) * KD_MAX_SEGS;
    int id_run = base - 1;
    for (int q = 0; q < per; q++) {
      const int p = p0 + q;
      if (p < n) {
        if (flags & (1u << q)) {
          id_run++;
          nlo[id_run] = (unsigned short)p;
        }
        J.seg_of_pos[p] = (unsigned short)id_run;
      }
    }
    if (tid == 0) nlo[total] = (unsigned short)n;
    __syncthreads();  // (the scratch arrays live in global memory: the barrier's workgroup-scope fence publishes them)
    if (total == nseg) break;  // nothing was split: the ordering is complete
    nseg = total;
    cur ^= 1;
    ext[axis] *= 0.5f;
  }
  // ---- outputs
  for (int p = tid; p < n; p += KD_THREADS) {
    const int id = (int)(kd_key[p] & 0xffffull);
    J.order[p] = id;
    J.inv[id] = p;
    J.xs4[p] = J.x4[id];
  }
  if (J.raw_feat)
    for (int q = tid; q < n * 2; q += KD_THREADS) {  // two float4 per point: 5 floats + 3 zeros
      const int r = q >> 1, h = q & 1;
      const float* src = J.raw_feat + (size_t)(kd_key[r] & 0xffffull) * FD;
      J.feat[q] = h == 0 ? make_float4(src[0], src[1], src[2], src[3]) : make_float4(src[4], 0.f, 0.f, 0.f);
    }
  if (J.raw_label)
    for (int q = tid; q < n * 5; q += KD_THREADS) {  // five float4 per point: 19 floats + 1 zero
      const int r = q / 5, h = q - 5 * r;
      const float* src = J.raw_label + (size_t)(kd_key[r] & 0xffffull) * NC + 4 * h;
      J.label[q] = make_float4(src[0], src[1], src[2], h < 4 ? src[3] : 0.f);
    }
  if (J.raw_geo)
    for (int r = tid; r < n; r += KD_THREADS) {
      const float* src = J.raw_geo + (size_t)(kd_key[r] & 0xffffull) * 2;
      J.geo[r] = make_float2(src[0], src[1]);
    }
}

// ------------------------------------------------------------------------------------------
// k_transform_pose: CvoFrameGPU::transform_pointcloud (CvoFrameGPU.cu:44-61) - the points of a frame under its
// 3x4 row-major pose, for the multi-frame edge kernel.  Both copies of the coordinates (original and spatial
// order) are rewritten; a rigid motion keeps the spatial order compact, so it is reused.
// ------------------------------------------------------------------------------------------
struct Pose12 {
  float T[12];
};
__global__ __launch_bounds__(256) void k_transform_pose(int n, Pose12 pose, const float4* __restrict__ in_x4,
                                                        const float4* __restrict__ in_xs4, float4* __restrict__ out_x4,
                                                        float4* __restrict__ out_xs4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float4 a = in_x4[i], b = in_xs4[i];
  const V3 ta = transform_point_pose_vec(pose.T, a.x, a.y, a.z);
  const V3 tb = transform_point_pose_vec(pose.T, b.x, b.y, b.z);
  out_x4[i] = make_float4(ta.x, ta.y, ta.z, 0.f);
  out_xs4[i] = make_float4(tb.x, tb.y, tb.z, 0.f);
}

}  // namespace cvo_dev
