/*
 * cvo_hip_debug.h -- test and profiling hooks of the MI355X backend.  NOT part of the drop-in boundary
 * (include/cvo_hip.h): nothing here replaces a reference interface; tests/, bench.py and scripts/ use these entry
 * points to look inside a context (last ELL matrix, kernel timings, candidate-list statistics, the device's scalar
 * routines on caller inputs).
 */
#ifndef CVO_HIP_DEBUG_H
#define CVO_HIP_DEBUG_H

#include "cvo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- test / profiling hooks ---------------------------------------------------------- */
/* Dumps the ELL kernel matrix of the LAST iteration executed by cvo_align_ex (row stride K):
 * mat/ind sized n_source*K, nonzeros sized n_source; K = the num_neighbors of that iteration. */
int cvo_debug_last_ell(cvo_ctx* ctx, int K, float* mat, int* ind, unsigned int* nonzeros);
/* A batch is enqueued as n_groups sub-batches (one stream each) of pairs_per_group pairs: these are the
 * launches a profiler sees. */
int cvo_debug_last_geometry(cvo_ctx* ctx, int* n_groups, int* pairs_per_group);
/* Re-issues the k_scan launches of one optimiser iteration (one per sub-batch) `reps` times on the final
 * state of the last call, timed with HIP events on the context's stream (bench.py's roofline leg).
 * *ms = average milliseconds per k_scan launch (of pairs_per_group pairs). */
int cvo_debug_time_scan(cvo_ctx* ctx, int reps, float* ms);
/* Tile-culling statistics of the last align call (all pairs, all iterations): number of fine tiles
 * the scan executed and the tile shape; executed pair tests = tiles * rows_per_tile * targets_per_tile. */
int cvo_debug_scan_stats(cvo_ctx* ctx, unsigned long long* tiles, int* rows_per_tile, int* targets_per_tile);
/* Re-issues the two per-iteration kernels (k_assoc; k_coeff including its update tail) one launch per sub-batch,
 * `reps` times, on the state the last call left behind and without writing anything back; timed with HIP events on
 * the context's stream.  *ms_* = average milliseconds per launch (of pairs_per_group pairs). */
int cvo_debug_time_kernels(cvo_ctx* ctx, int reps, float* ms_assoc, float* ms_coeff);
/* Kernel durations inside the optimiser loop itself.  With CVO_KERNEL_CLOCK set in the environment when the context is
 * created, the first block of a pair in k_assoc (lean graph) / k_coeff stamps its entry on the device's constant-rate counter
 * (s_memrealtime) and the block that finishes the pair's work in the launch (twist reduction / update) closes the
 * interval; the per-pair sums are part of the state.  Returns the averages of the last align call in ms - first
 * block in to last block out per pair and launch, the quantity rocprofv3 --kernel-trace --stats averages per launch -
 * and the number of k_coeff intervals behind them.  The counter's rate is calibrated against HIP events. */
int cvo_debug_kernel_clock(cvo_ctx* ctx, float* ms_assoc, float* ms_coeff, unsigned long long* launches);
/* Candidate-list reuse of the last align call, summed over the pairs: how many times the candidate bitmap was
 * (re)built by k_scan, the optimiser iterations run, and the candidate pairs k_assoc evaluated exactly. */
int cvo_debug_list_builds(cvo_ctx* ctx, unsigned long long* builds, unsigned long long* iterations,
                          unsigned long long* candidate_evaluations);
/* How the last list build of pair `pair` of the last align call classed its rows: rows beyond the 64-entry candidate
 * lists (served by k_assoc_dense), those of them beyond a long list as well (literal scan of all targets), and whether
 * the pair ended in the dense regime (no lists at all). */
int cvo_debug_row_classes(cvo_ctx* ctx, int pair, int* overflow_rows, int* scanned_rows, int* dense_regime);
/* Number of candidate pairs in the bitmap the last iteration used (superset of nnz). */
int cvo_debug_last_candidates(cvo_ctx* ctx, unsigned long long* out);
/* Runs the device's scalar restatements of the reference's host-side maths (cubic roots of poly_solver_order3,
 * the step selection of compute_step_size, Exp_SEK3, ||SE3 log||, update_tf, the indicator windows) on caller-supplied
 * inputs, so that the device code itself can be pinned against numpy / scipy.  ops and layouts: k_scalar_math in
 * unified_cvo_amd/csrc/cvo_kernels.h.  in / out: host arrays of 16 doubles per item (op 7: one item of 2 + n / n). */
int cvo_debug_scalar_math(cvo_ctx* ctx, int op, int n, const double* in, double* out);
/* CVO_VERIFY_LISTS=1 (environment, read when a call starts): rows k_verify re-derived with the literal scan during the
 * last align call, summed over pairs and iterations (0 when the check was off). */
int cvo_debug_verified_rows(cvo_ctx* ctx, unsigned long long* rows);
/* The spatial (k-d) ordering of a resident cloud: out[r] = original index of the point at sorted position r (n entries).
 * Computed on the device at upload (k_kd_order) for clouds of 8 .. 16384 finite points, on the host otherwise and under
 * CVO_ORDER=host / virtual / CVO_NO_SORT; no result depends on it. */
int cvo_debug_cloud_order(const cvo_cloud* cloud, int* out);
/* Free / total bytes of the context's device (hipMemGetInfo), for leak checks without a second HIP runtime in the process. */
int cvo_debug_device_memory(cvo_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
#ifdef __cplusplus
}
#endif
#endif
