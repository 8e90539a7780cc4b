// cvo_k_cloud.h -- k_kd_order (spatial ordering of an uploaded cloud) and k_transform_pose (multi-frame edges).
// Part of the kernel set of cvo_kernels.h (which states the whole iteration); compiled only as part of cvo_hip.hip.
#pragma once
#include "cvo_wave.h"

namespace cvo_dev {

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
  const int* raw_lid;        // n class ids (one-hot clouds), original order, or null
  int* lid;                  // out: the ids in spatial order
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
  if (J.raw_lid)
    for (int r = tid; r < n; r += KD_THREADS) J.lid[r] = J.raw_lid[kd_key[r] & 0xffffull];
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
