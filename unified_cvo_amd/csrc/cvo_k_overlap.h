// cvo_k_overlap.h -- k_tile_spheres and k_overlap: inner_product_gpu / function_angle (CvoGPU.cu:1719-1778, 1814-1846) in ONE launch.
// Part of the kernel set of cvo_kernels.h; compiled only as part of cvo_hip.hip.
//
// A single evaluation of <f_X, f_Y> needs no candidate structure: nothing is reused.  The list chain (k_update<INIT> ->
// k_prep -> k_scan -> k_list -> k_assoc_dense -> k_assoc) spends 90 us on five dependent launches for 10k x 10k points, as
// long at 5k x 5k.  Here a block takes 64 consecutive source rows of the spatial order (a compact blob), its OV_WAVES waves
// share out the 64-target tiles of the spatially ordered target cloud whose bounding sphere - under the pose of the call -
// comes within reach of the rows' sphere, and every lane runs its row against a tile's 64 transformed targets (broadcast
// reads out of LDS): the reference's own distance expression as the filter.  What passes is compacted per tile, and the
// list is evaluated 64 pairs at a time - eval_pair_yt, the reference's exact per-pair arithmetic, on full waves - with
// every row adding its own pairs in ascending position.  Row sums and counts meet in LDS, block partials in a last-block
// gate; the sum lands in pinned host memory.
//
// What the chain has and this kernel has not: the first-K truncation of a row (`if (num_inds == num_neighbors) break;`,
// CvoGPU.cu:585) needs the hits of a row in ascending ORIGINAL j.  A row that finds more than K pairs raises a flag and the
// caller repeats the evaluation with the list chain; while no row does, the two sum the same values (in another order:
// double accumulation, the float the API returns is the same).
#pragma once
#include "cvo_pair_math.h"

namespace cvo_dev {

#ifndef CVO_OV_WAVES
#define CVO_OV_WAVES 8
#endif
constexpr int OV_WAVES = CVO_OV_WAVES;  // waves per block of k_overlap = shares of a row tile's target tiles
constexpr int OV_LIST_CAP = 1024;       // target tiles a block lists per round (65536 targets)
constexpr int OV_QCAP = 1024;           // pairs a wave's compacted list holds: 16 rows x 64 targets

struct OverlapJob {
  PairDesc D;              // N, M, xs4, ys4 and the attribute arrays in spatial order (nothing else is set)
  const float4* xtile;     // [2 per tile] bounding sphere {centre, radius} and box {half extents about that centre} of every
                           // 64 consecutive positions of the source cloud ...
  const float4* ytile;     // ... and of the target cloud (in its own frame: the centre moves with the pose)
  int n_xtiles, n_ytiles;
  float R[9], T[3];        // the transform of the call as PairState::R / T hold it (CvoGPU.cu:1363-1364)
  float ell;
  int K;
  float stretch;           // bound on |Rinv v| / |v| (1 + rounding for a rotation; a caller may pass any matrix)
  double* part;            // [n_xtiles] sums of the row tiles
  int* gate;               // [2] blocks that have stored their partial; rows that found more than K pairs
  double* sum_host;        // pinned host memory: the sum ...
  int* over_host;          // ... and whether it is void (some row found more than K pairs: first-K needs the list chain)
};

// Bounding spheres and boxes of the 64-point tiles of a spatially ordered cloud: centre of the bounding box, largest distance
// to it; half extents of the box.
__global__ __launch_bounds__(256) void k_tile_spheres(int n, const float4* __restrict__ xs4, float4* __restrict__ tile4) {
  const int tile = (int)(blockIdx.x * 4 + (threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
  if (tile * 64 >= n) return;
  const float4 p = xs4[min(tile * 64 + lane, n - 1)];
  float lo[3] = {p.x, p.y, p.z}, hi[3] = {p.x, p.y, p.z};
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      lo[c] = fminf(lo[c], __shfl_xor(lo[c], o));
      hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o));
    }
  const float cx = 0.5f * lo[0] + 0.5f * hi[0], cy = 0.5f * lo[1] + 0.5f * hi[1], cz = 0.5f * lo[2] + 0.5f * hi[2];
  const float dx = p.x - cx, dy = p.y - cy, dz = p.z - cz;
  float r = sqrtf(dx * dx + dy * dy + dz * dz);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) r = fmaxf(r, __shfl_xor(r, o));
  // (rounding of the three squares, their sum and the root: a few ulp of r, plus what the centre's own rounding moved)
  if (lane == 0) {
    const float slack = 1e-6f * (fabsf(cx) + fabsf(cy) + fabsf(cz)) + 1e-30f;
    tile4[2 * tile] = make_float4(cx, cy, cz, r * 1.00001f + slack);
    tile4[2 * tile + 1] = make_float4((hi[0] - lo[0]) * 0.500001f + slack, (hi[1] - lo[1]) * 0.500001f + slack,
                                      (hi[2] - lo[2]) * 0.500001f + slack, 0.f);
  }
}

#ifdef CVO_OV_STAMPS  // experiment builds only: where a wave of k_overlap spends its time (scripts/overlap_stamps.py)
__device__ unsigned long long g_ov_ticks[4096][8];
#define OV_T() __builtin_readcyclecounter()
#else
#define OV_T() 0ull
#endif
// (the body; the kernel proper - k_overlap_entry, cvo_hip.hip - hands it one of the jobs it takes as kernel ARGUMENTS: no
// descriptor upload precedes the launch)
template <int FEAT>
__device__ __forceinline__ void k_overlap(const OverlapJob& Jr, const DevParams& P) {
  const OverlapJob* __restrict__ J = &Jr;
  if ((int)blockIdx.x >= J->n_xtiles) return;  // (the jobs of a launch share the grid of the largest)
  const PairDesc* __restrict__ D = &J->D;
  const int N = D->N, M = D->M, n_ytiles = J->n_ytiles;
  // (wave-uniform values are said to be: the compiler keeps what it derives from threadIdx.x in vector registers, and the
  // tile loop, the queue length and every LDS base would be per-lane arithmetic)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
  [[maybe_unused]] const unsigned long long ov_t0 = OV_T();
  [[maybe_unused]] unsigned long long ov_cull = 0, ov_scan = 0, ov_flush = 0, ov_n_seen = 0, ov_n_scan = 0, ov_n_flush = 0;
  __shared__ f32x4 s_y[OV_WAVES][64];     // per wave: the transformed targets of the tile being scanned
  __shared__ double s_sum[OV_WAVES][64];  // per wave: its share of every row's sum ...
  __shared__ unsigned s_cnt[OV_WAVES][64];  // ... and of its number of pairs
  __shared__ int s_last;
  // update_tf (CvoGPU.cu:94-112), as k_update<INIT> runs it for the list chain: same operands, same floats
  Pose pose;
  update_tf(J->R, J->T, pose.Ri, pose.Ti);
  const int row = (int)blockIdx.x * 64 + lane;  // a sorted position of the source cloud = the index of its attributes
  const bool live = row < N;
  const float4 x = D->xs4[live ? row : N - 1];
  RowData r = make_row(P, x, J->ell);
  if (FEAT == FEAT_HOT) r.lid = D->xlid[live ? row : N - 1];
  const FeatDen F = make_feat_den(P);
  // how far a target can be from the rows' sphere / box and still pass some row's cut-off
  const float4 xs = J->xtile[2 * blockIdx.x], xh = J->xtile[2 * blockIdx.x + 1];
  const float r_cut_all = sqrtf(wave_minmax_f32<true>(live ? r.d2_thres : 0.f)) * 1.000001f;
  const float stretch = J->stretch;
  double asum = 0;
  unsigned cnt = 0;
  f32x4* const my_y = &s_y[wave][0];
  // The rows of the block as the evaluating lanes need them (a lane evaluates pairs of OTHER lanes' rows).
  __shared__ f32x4 s_rowx[64];   // x, y, z, d2_thres
  __shared__ double s_rowd[64][2];  // den, rcp
  __shared__ int s_rowlid[64];
  if (wave == 0) {
    s_rowx[lane] = f32x4{r.x, r.y, r.z, r.d2_thres};
    s_rowd[lane][0] = r.den;
    s_rowd[lane][1] = r.rcp;
    s_rowlid[lane] = r.lid;
  }
  // The pairs of the tile being scanned that passed the cut-off, compacted: {row | slot of the target in the tile << 6},
  // row-major, a row's pairs in ascending slot - every lane knows where its row's run starts - and the values they
  // evaluate to (-1: the pair was dropped), read back by the rows.  OV_QCAP entries: all a tile can have for 16 rows.
  __shared__ unsigned short s_qm[OV_WAVES][OV_QCAP];
  __shared__ float s_qa[OV_WAVES][OV_QCAP];
  unsigned short* const qm = &s_qm[wave][0];
  float* const qa = &s_qa[wave][0];
  const int row0 = (int)blockIdx.x * 64;
  [[maybe_unused]] const unsigned long long ov_t1 = OV_T();
  // The target tiles this block has to look at: those whose bounding sphere AND bounding box - moved by the pose - come
  // within reach of the rows'.  All waves test (a tile per thread), the survivors are listed in ascending tile order and
  // dealt out to the waves round robin: a wave's share depends on the data only (the sums stay reproducible) and no wave
  // gets more than its fair number of tiles (dealt by tile NUMBER the slowest wave had four while the average had 1.4).
  __shared__ unsigned short s_list[OV_LIST_CAP];
  __shared__ int s_wcount[OV_WAVES];
  for (int base = 0; base < n_ytiles; base += OV_LIST_CAP) {
    const int round_end = min(base + OV_LIST_CAP, n_ytiles);
    int n_list = 0;
    for (int t0 = base; t0 < round_end; t0 += 64 * OV_WAVES) {
      const int t = t0 + (int)threadIdx.x;
      bool visit = false;
      if (t < round_end) {
        const float4 s = J->ytile[2 * t], h = J->ytile[2 * t + 1];
        const V3 c = transform_point(pose.Ri, pose.Ti, s.x, s.y, s.z);
        const float dx = c.x - xs.x, dy = c.y - xs.y, dz = c.z - xs.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        // (slack: the rounding of the transformed centre and of the tile's transformed points - a few ulp of the TERMS of
        // Rinv y + Tinv, which a translation that cancels the rotated coordinates leaves far larger than the result)
        const float ym = fabsf(s.x) + fabsf(s.y) + fabsf(s.z) + s.w;
        const float rm = fabsf(pose.Ri[0]) + fabsf(pose.Ri[1]) + fabsf(pose.Ri[2]) + fabsf(pose.Ri[3]) + fabsf(pose.Ri[4]) +
                         fabsf(pose.Ri[5]) + fabsf(pose.Ri[6]) + fabsf(pose.Ri[7]) + fabsf(pose.Ri[8]);
        const float slack = 4e-6f * (rm * ym + fabsf(pose.Ti[0]) + fabsf(pose.Ti[1]) + fabsf(pose.Ti[2]) + fabsf(xs.x) + fabsf(xs.y) + fabsf(xs.z));
        const float reach = (xs.w + r_cut_all + s.w * stretch) * 1.0001f + slack;
        // the box of the moved tile: |Rinv| h about the moved centre
        const float bx = fabsf(pose.Ri[0]) * h.x + fabsf(pose.Ri[1]) * h.y + fabsf(pose.Ri[2]) * h.z;
        const float by = fabsf(pose.Ri[3]) * h.x + fabsf(pose.Ri[4]) * h.y + fabsf(pose.Ri[5]) * h.z;
        const float bz = fabsf(pose.Ri[6]) * h.x + fabsf(pose.Ri[7]) * h.y + fabsf(pose.Ri[8]) * h.z;
        const bool apart = fabsf(dx) > (bx + xh.x + r_cut_all) * 1.0001f + slack || fabsf(dy) > (by + xh.y + r_cut_all) * 1.0001f + slack ||
                           fabsf(dz) > (bz + xh.z + r_cut_all) * 1.0001f + slack;
        visit = !(d2 > reach * reach) && !apart;  // (NaNs visit)
      }
      const unsigned long long vm = __ballot(visit);
      if (lane == 0) s_wcount[wave] = __builtin_popcountll(vm);
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int w = 0; w < OV_WAVES; w++) {
        const int c = s_wcount[w];
        before += w < wave ? c : 0;
        total += c;
      }
      if (visit)
        s_list[n_list + before + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(vm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vm, 0u))] =
            (unsigned short)(t - base);
      n_list += __builtin_amdgcn_readfirstlane(total);
      __syncthreads();
    }
    // the next tile's targets are requested before the current tile is scanned
    int e = wave;
    int cur = -1;
    float4 y_cur = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fetch = [&](int& tile, float4& y0) {
      if (e >= n_list) {
        tile = -1;
        return;
      }
      tile = base + __builtin_amdgcn_readfirstlane((int)s_list[e]);
      e += OV_WAVES;
      const int j = tile * 64 + lane;
      y0 = ldg_f4(as_global(reinterpret_cast<const f32x4*>(D->ys4)) + min(j, M - 1));
    };
    fetch(cur, y_cur);
    while (cur >= 0) {
      int nxt;
      float4 y_nxt = make_float4(0.f, 0.f, 0.f, 0.f);
      fetch(nxt, y_nxt);
      [[maybe_unused]] const unsigned long long tc = OV_T();
      ov_n_seen++;
      const int jl = cur * 64 + lane;
      V3 yt = transform_point(pose.Ri, pose.Ti, y_cur.x, y_cur.y, y_cur.z);
      if (jl >= M) {  // positions beyond the cloud: a target no row can reach (inf - x is inf, and inf is below no cut-off)
        yt.x = yt.y = yt.z = __builtin_inff();
      }
      {
        // Second cull, against the tile as it really lies: the bounding box of its transformed targets and every row's own
        // distance to it (the spheres of two 64-point blobs overlap for half the tiles that hold no pair at all).
        const float inf = __builtin_inff();
        const bool in = jl < M;
        const float lx = wave_minmax_f32<false>(yt.x), ly = wave_minmax_f32<false>(yt.y), lz = wave_minmax_f32<false>(yt.z);
        const float hx = wave_minmax_f32<true>(in ? yt.x : -inf), hy = wave_minmax_f32<true>(in ? yt.y : -inf),
                    hz = wave_minmax_f32<true>(in ? yt.z : -inf);
        const float ex = fmaxf(fmaxf(lx - r.x, r.x - hx), 0.f), ey = fmaxf(fmaxf(ly - r.y, r.y - hy), 0.f),
                    ez = fmaxf(fmaxf(lz - r.z, r.z - hz), 0.f);
        const float e2 = ex * ex + ey * ey + ez * ez;  // <= the squared distance to any target of the tile (up to rounding)
        if (!__ballot(live && !(e2 * 0.9999f > r.d2_thres))) {
          cur = nxt;
          y_cur = y_nxt;
#ifdef CVO_OV_STAMPS
          ov_cull += OV_T() - tc;
#endif
          continue;
        }
      }
#ifdef CVO_OV_STAMPS
      ov_cull += OV_T() - tc;
      ov_n_scan++;
      const unsigned long long ts = OV_T();
      const unsigned long long fl0 = ov_flush;
#endif
      __builtin_amdgcn_wave_barrier();  // (the previous tile's slots have been read: LDS operations of a wave complete in order)
      my_y[lane] = f32x4{yt.x, yt.y, yt.z, 0.f};
      __builtin_amdgcn_wave_barrier();
      // Lanes = rows; the tile's targets reach them as broadcast LDS reads, eight at a time, and leave a bit per target in
      // the lane's hit mask.  A pair that passes the cut-off is not evaluated at once - the whole wave would run the double
      // exp for one or two lanes, at nearly every target of a near tile - and not queued one by one either (a ballot, a
      // rank and a store per lane and target cost as much as the scan): after the tile every lane knows its row's count,
      // a prefix sum over the wave gives every row its run in the compacted list, the runs are written, the list is
      // evaluated 64 pairs at a time on full waves, and every row adds its own run in ascending position.
      unsigned long long hits = 0;
      for (int k0 = 0; k0 < 64; k0 += 8) {
        unsigned bits = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) {
          // (one address for the wave: a broadcast; the eight reads of a group are in flight together.  v_readlane from the
          // lanes that hold the targets measured the same: 4.4k cycles per tile either way)
          const f32x4 tv = my_y[k0 + u];
          const float tx = tv.x, ty = tv.y, tz = tv.z;
          const float dx = tx - r.x, dy = ty - r.y, dz = tz - r.z;
          const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));  // (eval_pair_yt's expression)
          bits |= (d2 < r.d2_thres) ? (1u << u) : 0u;
        }
        hits |= (unsigned long long)bits << k0;
      }
      if (!live) hits = 0;
      if (__ballot(hits != 0) != 0ull) {
        [[maybe_unused]] const unsigned long long tf = OV_T();
        // runs: inclusive prefix of the counts inside every row of 16 lanes (DPP row_shr, zero fill), the four row totals
        // through scalar registers
        const int c = __builtin_popcountll(hits);
        int inc = c;
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
        inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
        const int t0 = __builtin_amdgcn_readlane(inc, 15), t1 = __builtin_amdgcn_readlane(inc, 31),
                  t2 = __builtin_amdgcn_readlane(inc, 47), t3 = __builtin_amdgcn_readlane(inc, 63);
        const int total = t0 + t1 + t2 + t3;
        const int quarter = lane >> 4;
        // all rows at once while the list holds them; a quarter of the rows (16 x 64 pairs at most) at a time otherwise
        const int n_groups = total <= OV_QCAP ? 1 : 4;
        for (int g = 0; g < n_groups; g++) {
          const bool mine = n_groups == 1 || quarter == g;
          const int before = n_groups == 1 ? (quarter > 0 ? t0 : 0) + (quarter > 1 ? t1 : 0) + (quarter > 2 ? t2 : 0) : 0;
          const int base = before + inc - c;
          const int n_here = n_groups == 1 ? total : (g == 0 ? t0 : g == 1 ? t1 : g == 2 ? t2 : t3);
          if (mine) {
            int q = base;
            for (unsigned long long m = hits; m; m &= m - 1) qm[q++] = (unsigned short)(lane | (__builtin_ctzll(m) << 6));
          }
          __builtin_amdgcn_wave_barrier();
          for (int b0 = 0; b0 < n_here; b0 += 64) {
            const int q = b0 + lane;
            if (q < n_here) {
              const unsigned m = qm[q];
              const int rho = (int)(m & 63u), k = (int)(m >> 6);
              const f32x4 y = my_y[k];
              const f32x4 rx = s_rowx[rho];
              RowData rr;
              rr.x = rx.x;
              rr.y = rx.y;
              rr.z = rx.z;
              rr.l = 0.f;  // (not read by the pair arithmetic)
              rr.d2_thres = rx.w;
              rr.lid = FEAT == FEAT_HOT ? s_rowlid[rho] : 0;
              rr.den = s_rowd[rho][0];
              rr.rcp = s_rowd[rho][1];
              float a;
              const bool kept = eval_pair_yt<FEAT>(P, D, F, row0 + rho, rr, cur * 64 + k, make_float4(y.x, y.y, y.z, 0.f), a) && a > P.sp_thres;
              qa[q] = kept ? a : -1.f;  // (kernel values are products of squares and exponentials: never negative)
            }
#ifdef CVO_OV_STAMPS
            ov_n_flush++;
#endif
          }
          __builtin_amdgcn_wave_barrier();
          if (mine) {
            for (int q = base; q < base + c; q++) {
              const float v = qa[q];
              if (v >= 0.f) {
                asum += (double)v;
                cnt++;
              }
            }
          }
          __builtin_amdgcn_wave_barrier();  // (the list is the next group's, the next tile's)
        }
#ifdef CVO_OV_STAMPS
        ov_flush += OV_T() - tf;
#endif
      }
#ifdef CVO_OV_STAMPS
      ov_scan += OV_T() - ts - (ov_flush - fl0);
#endif
      cur = nxt;
      y_cur = y_nxt;
    }
    __syncthreads();  // (the list is the next round's)
  }
  [[maybe_unused]] const unsigned long long ov_t2 = OV_T();
  s_sum[wave][lane] = asum;
  s_cnt[wave][lane] = cnt;
  __syncthreads();
  if (wave == 0) {
    double s = 0;
    unsigned c = 0;
#pragma unroll
    for (int w = 0; w < OV_WAVES; w++) {
      s += s_sum[w][lane];
      c += s_cnt[w][lane];
    }
    const bool over = live && c > (unsigned)J->K;
    if (!live) s = 0;
    s += dpp_f64<DPP_XOR1>(s);
    s += dpp_f64<DPP_XOR2>(s);
    s += dpp_f64<DPP_HALF_MIRROR>(s);
    s += dpp_f64<DPP_MIRROR>(s);
    const double tot = lane_f64(s, 0) + lane_f64(s, 16) + (lane_f64(s, 32) + lane_f64(s, 48));
    const bool any_over = __ballot(over) != 0ull;
    if (lane == 0) {
      st_x<true>(J->part + blockIdx.x, tot);
      if (any_over) __hip_atomic_fetch_add(J->gate + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int done = __hip_atomic_fetch_add(J->gate, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = done == J->n_xtiles - 1 ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
    if (s_last) {
      // the pair's last block: the row tiles' sums in tile order (lane l takes tiles l, l + 64, ...; a fixed tree finishes)
      double t = 0;
      for (int b = lane; b < J->n_xtiles; b += 64) t += ld_x<true>(J->part + b);
      t += dpp_f64<DPP_XOR1>(t);
      t += dpp_f64<DPP_XOR2>(t);
      t += dpp_f64<DPP_HALF_MIRROR>(t);
      t += dpp_f64<DPP_MIRROR>(t);
      const double all = lane_f64(t, 0) + lane_f64(t, 16) + (lane_f64(t, 32) + lane_f64(t, 48));
      if (lane == 0) {
        const int n_over = __hip_atomic_load(J->gate + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(J->gate, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (ready for the next call)
        __hip_atomic_store(J->gate + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *J->over_host = n_over;
        *J->sum_host = all;
      }
    }
  }
#ifdef CVO_OV_STAMPS
  if (lane == 0 && blockIdx.y == 0) {
    unsigned long long* o = g_ov_ticks[((int)blockIdx.x * OV_WAVES + wave) & 4095];
    o[0] = ov_t1 - ov_t0;                 // prologue
    o[1] = ov_cull;                       // tiles: transform + box test
    o[2] = ov_scan;                       // tiles: scan + parking
    o[3] = ov_flush;                      // evaluations
    o[4] = OV_T() - ov_t2;                // reduction + gate
    o[5] = ov_n_seen | (ov_n_scan << 16) | (ov_n_flush << 32);
    o[6] = OV_T() - ov_t0;                // everything
    o[7] = ov_t1 - ov_t0 + (ov_t2 - ov_t1);
  }
#endif
}

}  // namespace cvo_dev
