// cvo_k_assoc.h -- k_assoc: ordered association + flow, one thread per row over the cached candidate lists; the flow gate (twist of the iteration).
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_wave.h"
#include "cvo_pair_math.h"

namespace cvo_dev {

// thrust::reduce of omega_gpu / v_gpu (CvoGPU.cu:824-825) from the association block partials, Eigen's
// normalize() and the matrices of compute_step_size_xi: once per pair and iteration, by one wave of the block of
// k_assoc that stores its partial last.
// Lane l owns component (l & 7) of blocks l>>3, l>>3 + 8, ... (independent loads, all in flight), the eight
// groups meet through DPP / ds_swizzle; the order of the additions is fixed.  k_coeff reads the 42 floats with
// scalar loads in its first burst (they used to be reduced again by every one of its blocks: ~3 us of
// dependent round trips in front of each row loop and a hot spot of 150 readers per cache line).
static_assert(sizeof(XiMats) <= 46 * sizeof(float), "PairState::xi holds an XiMats and the two words of its stamp");
// (the partials are data-tagged granules, cvo_wave.h: a granule that has not landed yet carries an older tag and the
// round is read again - the elected block no longer waits for anybody's store acknowledgement)
__device__ __forceinline__ double coeff_twist_load(const PairDesc* __restrict__ D, int nparts, unsigned tag) {
  const int lane = threadIdx.x & 63;
  const CVO_GLOBAL unsigned long long* src = as_global(D->flow_part) + 2 * (lane & 7);
  double acc = 0;
  // eight blocks per lane and round (64 row blocks = one round), all sixteen granule loads issued before the first
  // addition; slots past the end re-read the last block and add zero
  for (int b = lane >> 3; b < nparts; b += 64) {
    TaggedF64 p[8];
    // (component 7 is padding nobody writes: its lanes read it - the sum is never used - without looking at tags.  The
    // poll is bounded: a partial that has not shown up after ~0.5 s never will - the pair is latched as failed,
    // PairState::sync_err, and the call returns CVO_E_HIP instead of hanging the device)
    const bool pad = (lane & 7) == 7;
    for (int polls = 0;; polls++) {
#pragma unroll
      for (int u = 0; u < 8; u++) p[u] = ld_tagged<true>(src + (size_t)min(b + 8 * u, nparts - 1) * FLOW_GRANULES);
      bool ok = true;
#pragma unroll
      for (int u = 0; u < 8; u++) ok = ok && (pad || p[u].carries(tag));
      if (__ballot(!ok) == 0ull) break;
      if (polls > PARTIAL_POLL_LIMIT) {
        if (lane == 0) D->st->sync_err = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) acc += (b + 8 * u < nparts) ? p[u].value() : 0.0;
  }
  return acc;
}
// stamp_epoch / stamp_k: the pair's launch generation and iteration count as this launch of k_assoc found them, left next to
// the matrices (xi[46], xi[47]): whoever reads the twist can tell which iteration it belongs to (update_speculate must - a
// speculative block that starts after its launch's update has run would otherwise combine the NEXT state with THIS twist)
__device__ __forceinline__ void twist_finalize(const PairDesc* __restrict__ D, int nparts, unsigned tag, int stamp_epoch, int stamp_k) {
  double acc = coeff_twist_load(D, nparts, tag);
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
  if ((threadIdx.x & 63) == 0) {
    const float* mv = reinterpret_cast<const float*>(&M);
    float* dst = D->st->xi;
#pragma unroll
    for (int q = 0; q < (int)(sizeof(XiMats) / sizeof(float)); q++) dst[q] = mv[q];
    dst[46] = __int_as_float(stamp_epoch);
    dst[47] = __int_as_float(stamp_k);
  }
}
// The flow partial of this block is stored; the block that finds it was the last one of its pair reduces them.
// Every thread of the block calls this.
__device__ __forceinline__ bool flow_gate(const PairDesc* __restrict__ D, int nblocks, int nparts, unsigned tag, int stamp_epoch, int stamp_k) {
  __shared__ int s_flow_last;
  // (no wait for this block's partial: a store and an atomic of one wave to different addresses are not ordered on their
  // way to memory, so the counter says who reduces, the granules' tags say when a partial has arrived - cvo_wave.h)
  __syncthreads();
  if (threadIdx.x == 0) {
    const int done = __hip_atomic_fetch_add(D->gate_flow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_flow_last = (done == nblocks - 1) ? 1 : 0;
    if (done == nblocks - 1) __hip_atomic_store(D->gate_flow, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (s_flow_last && threadIdx.x < 64) twist_finalize(D, nparts, tag, stamp_epoch, stamp_k);
  return s_flow_last != 0;
}

// A_sum of a single evaluation (SparseKernelMat.cu:62-68; inner_product_gpu, CvoGPU.cu:1719-1778): the block of k_assoc
// that stores its partial last adds the row blocks' sums of kernel values - in the order k_update's reduction uses (lane l
// of sixteen takes blocks l, l + 16, ...; a four-step butterfly), so the value is the one that path returns - and posts
// it to pinned host memory: the chain of an inner product ends with k_assoc.  Every thread of the first wave calls this.
__device__ __forceinline__ void asum_gate(const PairDesc* __restrict__ D, int nblk) {
  __shared__ int s_asum_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (see flow_gate)
  __syncthreads();
  if (threadIdx.x == 0) {
    const int done = __hip_atomic_fetch_add(D->gate_flow, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_asum_last = (done == nblk - 1) ? 1 : 0;
    if (done == nblk - 1) __hip_atomic_store(D->gate_flow, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_asum_last) return;
  double s = 0;
  if (threadIdx.x < 16)  // (this gate waits for the stores; the tags are not needed)
    for (int b = (int)threadIdx.x; b < nblk; b += 16) s += ld_tagged<true>(as_global(D->flow_part) + (size_t)b * FLOW_GRANULES + 12).value();
  s += dpp_f64<DPP_XOR1>(s);
  s += dpp_f64<DPP_XOR2>(s);
  s += dpp_f64<DPP_HALF_MIRROR>(s);
  s += dpp_f64<DPP_MIRROR>(s);
  if (threadIdx.x == 0) {
    D->st->asum = s;
    *D->asum_host = s;
  }
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
    EllEntry stage[ELL_STAGE][ASSOC_THREADS];  // during it: the rows' first ELL entries, one column per thread
  };
  unsigned long long cnt[ASSOC_THREADS / 64][4];
};

template <typename IdxT, int ASSOC_CAP, int FEAT, bool INSTR>
__device__ __forceinline__ void assoc_phase(const DevParams& P, const PairDesc* __restrict__ D, const IterView& iv,
                                            AssocShared& S, const int bx, const AssocRowHead& head, const unsigned tag) {
  const int N = D->N;
  const int pos = bx * ASSOC_THREADS + threadIdx.x;  // position in k_list's count-ordered row windows
  const int K = iv.K;
  RowAcc A;
  A.slot = D->ell + pos;
  A.stage = (CVO_LDS ell_vec_t*)&S.stage[0][threadIdx.x];
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
      // (Round 5 cached den = 2 l^2 and its refined reciprocal per row, 16 bytes next to the row head, rewritten only when
      // ell decays: 35 VALU instructions per row and iteration less - and the 64-pair step 3 % SLOWER, 60.9 against 59.1 ms
      // in scripts/exp_time.py: these kernels pay for bytes, not for arithmetic.  The constants are recomputed.)
      RowData r = make_row(P, x, iv.ell);
      if (FEAT == FEAT_HOT) r.lid = D->xlid[i];
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
        visit_pair_yt<FEAT>(P, D, F, i, pos, N, r, pxe, j, make_float4(ytv.x, ytv.y, ytv.z, 0.f), A);
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
          const ell_vec_t ev = A.stage[q * ASSOC_THREADS];  // (own LDS column: no barrier)
#ifdef CVO_ELL8
          asm volatile("flat_store_dwordx2 %0, %1 sc1" ::"v"(dst), "v"(ev) : "memory");
#else
          asm volatile("flat_store_dwordx4 %0, %1 sc1" ::"v"(dst), "v"(ev) : "memory");
#endif
          dst += N;
        }
      }
      if (INSTR) tt2 = __builtin_readcyclecounter();
    } else {
      // a row beyond its list: k_assoc_dense, which ran before this kernel, evaluated it (a wave per row) and left its
      // nonzero count and float flow sums; they join the block's reduction at the row's own position
      const RowRes rr = D->rowres[pos];
      A.o0 = rr.o[0]; A.o1 = rr.o[1]; A.o2 = rr.o[2];
      A.v0 = rr.v[0]; A.v1 = rr.v[1]; A.v2 = rr.v[2];
      A.asum = rr.asum;
      A.nnz = nnz_count(D->nnz_row[pos]);
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
  if (threadIdx.x < 56 && (threadIdx.x & 7) == 0 && !(P.debug_drop_partial == 1 && bx == 1)) {
    st_tagged(D->flow_part + (size_t)bx * FLOW_GRANULES + 2 * (threadIdx.x >> 3), tot, tag);  // read by another block of this launch (flow_gate)
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
template <typename IdxT, int ASSOC_CAP, int FEAT, bool INSTR>
__global__ __launch_bounds__(ASSOC_THREADS, FEAT != FEAT_GEO ? 1 : CVO_ASSOC_WAVES) void k_assoc(const PairDesc* __restrict__ descs,
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
    head.ip = FEAT != FEAT_GEO ? reinterpret_cast<const int*>(wb + row_off_ip(Npad))[pos] : 0;  // (only the feature lookups need it)
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
  // tag of this launch's partials (cvo_wave.h): the call's serial and the pair's iteration count
  const int stamp_epoch = st->epoch, stamp_k = st->k;
  const unsigned tag = partial_tag(D->call_serial, (unsigned)stamp_k);
  assoc_phase<IdxT, ASSOC_CAP, FEAT, INSTR>(P, D, load_iter_view(st), S, pb.bx, head, tag);
  // Everything from here on - the block's partial is on its way, the last-block counter, possibly the twist - is the
  // first wave's business.  The other waves retire now instead of sitting on their registers through a store
  // acknowledgement and an atomic round trip (~2 us of a ~7 us wave life; with thousands of waves queued behind them
  // that wait was throughput, not just latency).  A barrier only counts the waves that are still alive.
  if (threadIdx.x >= 64) return;
  // every row of the pair has been reduced (rows beyond their lists were evaluated by k_assoc_dense before this launch):
  // the block that stores its partial last finishes the twist of the iteration
  if (P.mode == 0) {
    const unsigned long long clk0 = pair_clock_peek(INSTR && P.kernel_clock && (lean & 3) == 1, st, 0);
    const bool last = flow_gate(D, nblk, nblk, tag, stamp_epoch, stamp_k);
    if (last && threadIdx.x == 0 && clk0) D->st->clk_last_assoc = pair_clock_ticks(clk0);  // added up by the update
  } else if (lean & 8) {  // single evaluation that only wants A_sum (inner_product_gpu)
    asum_gate(D, nblk);
  }
  if (INSTR && P.phase_ticks && threadIdx.x == 0) {
    g_phase_ticks[0][blockIdx.x & 8191][0] = tt0;
    g_phase_ticks[0][blockIdx.x & 8191][3] = __builtin_readcyclecounter();
  }
}

}  // namespace cvo_dev
