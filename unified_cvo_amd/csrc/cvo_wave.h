// cvo_wave.h -- wave-wide primitives of the gfx950 kernels: DPP / permlane reductions, the XCD-aware pair -> block mapping, address-space qualified loads and the coherent (sc1) accessors.
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_device.h"

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
// min / max of a float over the wave on the DPP paths (every lane gets the result; NaNs are dropped as fminf / fmaxf drop them)
template <bool MAX>
__device__ __forceinline__ float wave_minmax_f32(float v) {
  auto meet = [](float a, int ob) {
    const float o = __builtin_bit_cast(float, ob);
    return MAX ? fmaxf(a, o) : fminf(a, o);
  };
  v = meet(v, dpp_i32<DPP_XOR1>(__builtin_bit_cast(int, v)));
  v = meet(v, dpp_i32<DPP_XOR2>(__builtin_bit_cast(int, v)));
  v = meet(v, dpp_i32<DPP_HALF_MIRROR>(__builtin_bit_cast(int, v)));
  v = meet(v, dpp_i32<DPP_MIRROR>(__builtin_bit_cast(int, v)));
  const int iv = __builtin_bit_cast(int, v);
  float m = meet(__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), __builtin_amdgcn_readlane(iv, 16));
  m = meet(m, __builtin_amdgcn_readlane(iv, 32));
  return meet(m, __builtin_amdgcn_readlane(iv, 48));
}
// inclusive prefix sum of an int over the wave, in lane order, on the DPP paths: row_shr 1 / 2 / 4 / 8 with zero fill inside
// every row of 16 lanes, the three lower rows' totals through scalar registers
__device__ __forceinline__ int wave_prefix_incl_i32(int v, int lane) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
  const int quarter = lane >> 4;
  return v + (quarter > 0 ? t0 : 0) + (quarter > 1 ? t1 : 0) + (quarter > 2 ? t2 : 0);
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
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CVO_LDS __attribute__((address_space(3)))
// an ELL entry as a plain vector (storable through address-space qualified pointers, which class types are not)
#ifdef CVO_ELL8
typedef f32x2 ell_vec_t;
__device__ __forceinline__ ell_vec_t ell_to_vec(const EllEntry& e) { return ell_vec_t{e.a, __int_as_float(e.p)}; }
#else
typedef f32x4 ell_vec_t;
__device__ __forceinline__ ell_vec_t ell_to_vec(const EllEntry& e) { return ell_vec_t{e.a, e.yx, e.yy, e.yz}; }
#endif
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

// ---- data-tagged partials (round 6) ------------------------------------------------------------------------------
// A block partial that another block of the SAME launch reads (k_assoc's flow sums -> the flow gate, k_coeff's
// coefficient sums -> the update) used to be: coherent store, s_waitcnt vmcnt(0) for its acknowledgement from the memory
// side, then the arrival counter - a store and an atomic of one wave to different addresses are not ordered on their way
// to memory -, then the elected block's coherent loads: three dependent trips to the memory side per kernel on every
// pair's serial chain.  Now every double travels as two 8-byte granules {tag : 32 | half of the value : 32}, each
// written and read by ONE 64-bit relaxed agent-scope atomic (indivisible by construction), and the store is NOT waited
// for: store and counter are in flight together, the elected block reads the granules and checks that every one of them
// carries the tag of THIS launch of THIS pair of THIS call (partial_tag: a 32-bit mix of the call's serial - unique per
// call / queue submission - and the pair's iteration / launch count); a granule that has not landed yet still carries an
// older tag and is read again.  Two trips instead of three.  (The same form the XCD-resident experiment of round 3 used
// for its broadcasts.)  The VALUES and the order they are added in are what they were.
__device__ __forceinline__ unsigned partial_tag(unsigned long long serial, unsigned n) {
  unsigned h = (unsigned)serial * 0x9E3779B1u + (unsigned)(serial >> 32) * 0x7FEB352Du + (n + 1u) * 0x85EBCA6Bu;
  h ^= h >> 15;
  h *= 0x2C1B3C6Du;
  h ^= h >> 12;
  return h;
}
__device__ __forceinline__ void st_tagged(unsigned long long* g, double v, unsigned tag) {  // g: two granules
  const unsigned long long t = (unsigned long long)tag << 32;
  __hip_atomic_store(g, t | (unsigned long long)(unsigned)__double2loint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(g + 1, t | (unsigned long long)(unsigned)__double2hiint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct TaggedF64 {
  unsigned long long lo, hi;
  __device__ __forceinline__ bool carries(unsigned tag) const { return (unsigned)(lo >> 32) == tag && (unsigned)(hi >> 32) == tag; }
  __device__ __forceinline__ double value() const { return __hiloint2double((int)(unsigned)hi, (int)(unsigned)lo); }
};
// COH = false: the producer was an earlier kernel (plain loads; the tags are not looked at)
template <bool COH>
__device__ __forceinline__ TaggedF64 ld_tagged(const CVO_GLOBAL unsigned long long* g) {
  TaggedF64 t;
  t.lo = ld_g<COH>(g);
  t.hi = ld_g<COH>(g + 1);
  return t;
}
// granules per row block: k_assoc's partial (7 doubles, padded to one 128-byte line), k_coeff's (4 doubles)
constexpr int FLOW_GRANULES = 16, COEF_GRANULES = 8;
// polls (one memory-side round trip + s_sleep each, ~1-2 us) after which the elected block gives a missing partial up
constexpr int PARTIAL_POLL_LIMIT = 400000;

}  // namespace cvo_dev
