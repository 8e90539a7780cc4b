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


#include "cvo_wave.h"
#include "cvo_pair_math.h"
#include "cvo_k_prep.h"
#include "cvo_k_scan.h"
#include "cvo_k_list.h"
#include "cvo_k_assoc.h"
#include "cvo_k_assoc_dense.h"
#include "cvo_update.h"
#include "cvo_k_coeff.h"
#include "cvo_k_overlap.h"
#include "cvo_k_coeff_dense.h"
#include "cvo_k_debug.h"
#include "cvo_k_cloud.h"
