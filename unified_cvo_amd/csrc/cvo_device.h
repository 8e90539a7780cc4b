// cvo_device.h -- device-side data structures and scalar maths for the gfx950 backend.
//
// The arithmetic conventions (which a*b+c patterns are fused, the association of 3-term
// sums, where float is promoted to double) are the ones fixed in DESIGN.md "Numerics":
// device-side code of the reference (nvcc, -fmad=true) uses explicit fmaf(); host-side code
// of the reference (LieGroup.cpp, align_impl) uses plain unfused arithmetic.  This file is
// compiled with -ffp-contract=off so nothing else fuses.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/cvo_hip.h"
#include "../../include/cvo_hip_debug.h"

namespace cvo_dev {

constexpr int FD = CVO_FEATURE_DIMENSIONS;  // 5 floats stored as 8 (two float4)
constexpr int NC = CVO_NUM_CLASSES;         // 19 floats stored as 20 (five float4)
constexpr int FD_PAD = 8;
constexpr int NC_PAD = 20;
constexpr int IND_CAP = 512;  // capacity of each indicator FIFO (window + 1 must fit)

// Per-call constants (passed by value as a kernel argument).
struct DevParams {
  float sp_thres, sigma2, c2, c_sigma2, s_ell, s_sigma, c, d;
  float log_geo;                 // logf(sp_thres / sigma2), evaluated on the host
  float d2_c_thres, d2_s_thres;  // row independent cut-offs (CvoGPU.cu:512-515)
  float ell_min, ell_decay_rate;
  int ell_decay_start;
  int max_iter;  // loop bound (MAX_ITER, or the caller's smaller max_iterations)
  float eps, eps_2, min_step, max_step;
  int K_max;
  int window;
  float stable_thr;
  int use_geo, use_col, use_sem, use_range_ell, use_geotype;
  int trace_dense, trace_every, trace_capacity;
  int mode;  // 0 = align loop, 1 = single evaluation (inner product / association), 2 = single evaluation with the
             // non-isotropic kernel (CvoGPU.cu:217-327)
  float kinv[9];   // mode 2: inverse of the 3x3 kernel matrix, row-major
  float d2_cull;   // mode 2: squared Euclidean radius beyond which no pair can reach sp_thres (cull only)
  float s_ell_sq;  // mode 2: s_ell * s_ell kept in float, as that kernel's prologue does
  int T;     // target chunks (of 64) per scan wave: 1, 2, 4 or 8
  int groups_per_block;  // row groups per k_scan block (a multiple of 64)
  // Candidate-list reuse (DESIGN.md "Reusing the candidate list"): the scan runs with the cut-off radius
  // grown by skin = skin_frac * ell * sqrt(-2 log_geo); its bitmap stays valid while the targets have moved
  // less than the skin and ell has not grown.  0 = scan every iteration.
  float skin_frac;
  float rebuild_shrink;  // rebuild when ell < rebuild_shrink * ell_build (the lists would be (1/shrink)^3 too long)
  int lean_U;            // iterations between two rebuild opportunities in the lean graph
  int lean_U2;           // ... and in the short lean graph (motion too fast for lean_U, slow enough for a list to last lean_U2)
  int calm_U;            // a pair whose list outlives this many more iterations at its current speed reports itself calm
                         // (want = -1): the host may give it one rebuild opportunity per chunk (0 = never)
  int shrink_align;      // optional (ell-shrink) rebuilds wait for an iteration count with (k & shrink_align) == 0:
                         // 0 = at the next opportunity (the default since round 6; 63 was round 4's choice for batches)
  float skin_min, skin_max;  // clamp of the skin (fractions of the cut-off radius) before skin_frac
  float skin_blend;          // share of the pooled motion budget both the rotation and the translation allowance get on top of their own
  int dense_regime;      // 0: never switch a pair to the all-rows-dense regime (CVO_NO_DENSE_REGIME)
  float lean_skin;       // skin of the lean graph in units of (lean_U x the motion of one iteration)
  float horizon_margin;  // a list is renewed at a rebuild opportunity if it would not outlive (this x iterations to the next one) at its current speed
  int kernel_clock;  // CVO_KERNEL_CLOCK: accumulate per-pair kernel durations in PairState::clk_*
  int phase_ticks;  // CVO_PHASE_TICKS: leave per-block phase timestamps (g_phase_ticks) for cvo_debug_time_kernels
  int verify_lists;  // CVO_VERIFY_LISTS: k_verify re-derives every row with the literal scan after each association
  int keep_columns;  // write ell_j (the column of every ELL entry): exports, traces, the self-check and the single
                     // evaluations need it, the optimiser loop itself never reads it (4 of 20 bytes per nonzero)
  int fast_div_cd;  // 2^-20 <= |c|, |d| <= 2^20: the per-row float divisions by c and d may take their hoisted form (fdiv_hoisted)
  int row_max_cap;  // rows with more candidates than this leave the thread-per-row kernel for k_assoc_dense (a wave per row,
                    // long lists) even though a cached list would hold them: the host lowers it when few pairs are in flight
                    // (an iteration is then a chain of latencies and a wave runs as long as its longest row); only with long
                    // lists, ASSOC_CAP16 = off.  No result depends on it (every row joins k_assoc's reduction at its position)
  int row_max_busy;  // ... and the limit while a sixteenth of the rows overflow anyway (k_assoc_dense runs in every iteration
                     // then): 24 in a batch, 8 when at most four pairs are in flight (scripts/rowmax_probe.py: a clustered
                     // 3000-point pair 42.8 -> 34.4 us per iteration, 10k 92.5 -> 85.5)
  int long_lists;  // overflow rows keep a cached sorted candidate list of up to LONG_CAP entries (PairDesc::long_j)
  int debug_drop_partial;     // CVO_DEBUG_DROP_PARTIAL (tests only): 1 / 2 = row block 1 of k_assoc / k_coeff never publishes its partial:
                              // the elected block's bounded poll must end the pair (PairState::sync_err), not hang the device
  int debug_no_motion_bound;  // CVO_DEBUG_NO_MOTION_BOUND (tests only): the update pretends no target ever moves, so
                              // lists outlive their validity - what CVO_VERIFY_LISTS exists to catch
};

// Running state of one frame pair; lives in HBM, only touched by one thread of k_step.
struct PairState {
  // ---- what every block of the per-iteration kernels reads before it does anything (one scalar-load burst) ----
  int status;      // 0 running, 1 finished
  int rebuild;     // the candidate lists are stale: k_prep / k_scan / k_list rebuild them, everybody else waits
  int n_ovf;       // rows on the overflow list of k_assoc_dense (filled by k_list, reset by k_prep)
  int K;           // num_neighbors
  float ell;
  int epoch;       // k_coeff launches this pair completed (generation of its last-block counter)
  int k;           // iteration counter
  int iterations;
  float Rinv[9], Tinv[3];  // transform applied to the target cloud this iteration
  // ---- the rest of the scalar state ----
  float R[9], T[3];        // running pose (row-major R), CvoGPU.cu:1363-1364
  int ret;
  float step;
  float omega[3], v[3];
  unsigned nnz, max_nnz;
  double B, C, D, E;
  double dist;
  double asum;  // sum of kernel values (mode 1)
  unsigned long long ncand;
  unsigned long long ncand_list;  // candidates held by the current lists, summed over the rows (k_prep zeroes, k_list adds)
  unsigned long long noverflow;  // rows that took k_assoc's literal path
  unsigned long long ncand_total;  // candidate pairs evaluated exactly, summed over the iterations (statistics)
  int n_trace;
  float out_T[16];  // column-major [R^T | -R^T T]
  // candidate-list reuse: pose / ell the current bitmap was built with and its skin
  float Rb[9], Tb[3];
  float ell_build;
  // Skin of the current lists, per row: skin_i = skin_rot * rho_i + skin_tr, rho_i >= |y0| of every target that can
  // come within the row's cut-off while the lists live (k_prep).  The lists stay supersets of the exact test while
  // |Rinv - Rb|_F <= skin_rot and |Tinv - Tb| <= skin_tr: rows near the sensor, whose targets a rotation moves little,
  // get a thin skin; only the farthest rows pay for the whole motion bound.
  float skin_rot, skin_tr;
  int n_builds;
  int n_adopted;  // iterations whose scalar tail was adopted from the speculative run (update_speculate; statistics)
  // sticky: a data-tagged partial of another block of the launch did not arrive within PARTIAL_POLL_LIMIT polls (1 = flow
  // gate, 2 = update; cvo_wave.h).  Never seen in practice - every block issues its partial before its arrival counter -:
  // it ends the pair and turns what would be a hung device into CVO_E_HIP.
  int sync_err;
  int want_full, n_stalls;  // host hint (want_level / want_encode, cvo_kernels.h): 2 = a rebuild opportunity in every iteration,
                            // 1 = the short lean graph (one every lean_U2 iterations), 0 = the lean graph, -1 = calm (one per
                            // chunk); 4 = 2 with k_assoc_dense, 8 / 9 / 10 = -1 / 0 / 1 with it
  int all_dense;
  float skin_scale;  // backs the skin off while rows fall back to the literal scan (see update_body)
  int row_max;       // candidates a row may have and still be served thread-per-row by k_assoc (<= the list capacity): set by
                     // the update when it orders a rebuild, read by k_list (who overflows) and k_assoc until the next one
  int n_scan;        // rows of the last build beyond every list (more than LONG_CAP candidates, or ASSOC_CAP without long
                     // lists): k_assoc_dense scans all targets for them (filled by k_list, reset by k_prep)
  // all_dense = dense regime: every row is served by k_assoc_dense, no lists (see update_body)
  // A_sparsity_indicator_ell_update FIFOs (CvoGPU.cu:1167-1285): bookkeeping here, storage below
  int s_head, s_size, e_head, e_size;
  float s_sum, e_sum;
  // CVO_KERNEL_CLOCK: ticks of the s_memrealtime counter between the entry of a pair's first block and the exit of the
  // block that finishes the pair's work in the launch, [0] k_assoc (lean graph; its last interval is left in
  // clk_last_assoc by the flow gate and added by the update) / [1] k_coeff, summed over clk_n iterations
  unsigned clk_last_assoc, clk_n[2];
  float last_used, last_rate;  // list reuse: share of the motion allowance used since the build / used per iteration (statistics)
  int K_last;  // num_neighbors of the last EXECUTED iteration: the row stride upstream wrote its A matrix with (0: none ran)
  float temp_coef;  // 1 / (2.0 * ell * ell) narrowed to float (CvoGPU.cu:1060) for the CURRENT ell: the same for every row
                    // unless is_using_range_ell; evaluated by the update when ell changes instead of by every row of k_coeff
  unsigned long long clk_sum[2];
  // ---- everything above is the "hot" prefix k_update stages through LDS ----
  float sq[IND_CAP], eq[IND_CAP];
  // the normalised twist of the iteration and the matrices of compute_step_size_xi (XiMats), written by the last
  // block of the association launch, read by every block of k_coeff
  float xi[48];
  unsigned long long clk_start[2];  // CVO_KERNEL_CLOCK: entry stamp of the pair's first block in the running launch
  // CVO_VERIFY_LISTS: sticky result of k_verify (0 = every row of every iteration matched the literal scan); on the
  // first mismatch: iteration, row position and what differed (1 = count, 2 = column, 3 = value).  Outside the hot
  // prefix: written by k_verify's blocks with atomics, never staged by the update.
  int verify_err, verify_k, verify_pos, verify_what;
  unsigned long long verify_rows;  // rows checked so far
};

// Everything a kernel needs to know about one frame pair.
//
// Index spaces: "original" = the caller's point order (what the reference's ordered truncation and
// float accumulation order are defined on); "sorted" = spatial (k-d) order computed once per cloud at
// upload (xorder / yorder map sorted position -> original index).  k_scan works entirely in sorted
// space so that 4-row groups and 64-target chunks are spatially compact and whole tiles can be
// rejected by a bounding-box test; k_assoc maps candidates back and restores ascending original j.
// One nonzero of the ELL kernel matrix, as k_assoc leaves it for k_coeff: the kernel value and the TRANSFORMED target
// the pair was evaluated with (so that the coefficient pass neither gathers the target again nor repeats the
// transform: one streaming 16-byte load per nonzero).  The column index j lives in a parallel array (ell_j) that only
// the exports and the self-check read.
#ifdef CVO_ELL8
// Experiment build (scripts/exp_time.py ell8): the 8-byte entry {value, the target's SORTED position}; k_coeff gathers the
// initial target again and repeats transform_point_R_T (same function, same operands: same floats).  Half the bytes
// k_assoc leaves behind and k_coeff streams, against a dependent gather + 9 FMAs per nonzero in the coefficient loop.
struct EllEntry {
  float a;
  int p;
};
static_assert(sizeof(EllEntry) == 8, "EllEntry");
constexpr bool ELL8 = true;
#else
struct EllEntry {
  float a, yx, yy, yz;
};
static_assert(sizeof(EllEntry) == 16, "EllEntry");
constexpr bool ELL8 = false;
#endif
// p: the target's sorted position (what a cached list entry is)
__device__ __forceinline__ EllEntry make_ell(float a, float yx, float yy, float yz, int p) {
#ifdef CVO_ELL8
  return EllEntry{a, p};
#else
  return EllEntry{a, yx, yy, yz};
#endif
}

// What k_assoc_dense leaves per overflow row for the thread of k_assoc that owns the row's position.
struct RowRes {
  float o[3], v[3];  // sum_j a_ij (x_i x y_j), sum_j a_ij (y_j - x_i): float, accumulated in ascending j (CvoGPU.cu:779-780)
  double asum;       // sum_j a_ij (mode 1: A_sum)
};
static_assert(sizeof(RowRes) == 32, "RowRes");

// The row arrays of the per-iteration kernels open every pair's workspace, at offsets that depend only on the
// launch-wide padded row count: a block computes their addresses from kernel arguments (arena of the launch's
// first pair, stride between pairs, padded rows) and requests its rows together with the descriptor and the
// state, instead of one cold round trip later (every kernel starts with an invalidated L2 on this multi-die part).
//   cand_cnt int[Np] | ip int[Np] | nnz_row u32[Np] | pad | xp4 float4[Np] | cand_j 128 B x Np | ell [K_max][N] | ell_j
constexpr int ROW_PAD = 256;
__host__ __device__ inline size_t row_off_cand_cnt(int) { return 0; }
__host__ __device__ inline size_t row_off_ip(int Np) { return (size_t)4 * Np; }
__host__ __device__ inline size_t row_off_nnz(int Np) { return (size_t)8 * Np; }
__host__ __device__ inline size_t row_off_xp4(int Np) { return (size_t)16 * Np; }
__host__ __device__ inline size_t row_off_cand_j(int Np) { return (size_t)32 * Np; }
__host__ __device__ inline size_t row_off_ell(int Np) { return (size_t)160 * Np; }

struct PairDesc {
  // ---- read by the per-iteration kernels (k_assoc, k_coeff): kept together at the front ----
  int N, M, nblk_assoc, nblk_coeff;  // nblk_coeff = nblk_assoc * csplit partials of the coefficient phase
  // k_list orders the rows of every 256-row window by candidate count; POSITION = index in that order
  int* cand_cnt;   // [N] candidates of the row at each position
  void* cand_j;    // [ASSOC_CAP][N] cached candidate lists by position, u16 or i32: the targets' SORTED positions, in
                   // ascending ORIGINAL target index (coordinates and features are gathered from the spatially ordered
                   // arrays, where the candidates of neighbouring rows share cache lines)
  float4* xp4;     // [N] source xyz of the row at each position
  int* ip;         // [N] sorted row at each position = the row's index into the (spatially ordered) feature arrays
  const float4* y4;   // target xyz (initial cloud), ORIGINAL index
  EllEntry* ell;              // ELL kernel matrix [K_max][N] by POSITION: value + transformed target, ascending ORIGINAL j in a row
  int* ell_j;                 // [K_max][N]: the column (ORIGINAL j) of every entry
  unsigned* nnz_row;          // nonzeros[N], by position
  double* rowcoef;            // [N x csplit][4] by position and slice: (B, C, D, E) of the rows k_coeff_dense evaluated - summed
                              // slot by slot in the order a thread of k_coeff would have - picked up by k_coeff
  RowRes* rowres;             // [N] by position: results of the rows k_assoc_dense evaluated, picked up by k_assoc
  unsigned long long* flow_part;  // [nblk_assoc][FLOW_GRANULES]: omega(3), v(3), sum a as data-tagged granules (cvo_wave.h), two per value - one partial per row block of k_assoc
  unsigned long long* cnt_part;  // [nblk_assoc][4]: nnz, max, candidates, overflow rows
  unsigned long long* shadow;     // [SHADOW_WORDS]: the state a speculative run of the update left for this launch (update_speculate), data-tagged granules
  unsigned long long* coef_part;  // [nblk_assoc * COEFF_SPLIT_MAX][COEF_GRANULES]: B C D E as data-tagged granules
  int* done;        // [1] k_coeff: blocks that stored their partials (monotonic; the last one runs the update)
  int csplit;                // k_coeff: blocks per row block; block q of a row block takes the ELL slots s = q (mod csplit).
                             // (small clouds; a function of the pair's own size, so that batch == solo)
  const float4* xfeat;
  const float4* yfeat;
  const float4* xlabel;
  const float4* ylabel;
  const float2* xgeo;
  const float2* ygeo;
  const int* xlid;  // class ids of one-hot clouds, spatial order (FEAT_HOT launches only: every cloud of the call has them)
  const int* ylid;
  // ---- rebuild kernels, update, exports ----
  int Mpad, nchunks, nslices, rbw;  // rbw: 32-bit words of slice bits per row
  int NG;     // row groups of ROWS_PER_GROUP sorted rows
  int NGpad;  // NG rounded up for the coarse test (pad groups have empty boxes)
  float cx, cy, cz;  // centre subtracted in the cull arithmetic only
  float ymax;        // largest |y0| of the target cloud (bounds how far a pose change moves any target)
  double sqrt_nm;    // sqrt(double(N) * double(M)), the denominator of the sparsity indicator (CvoGPU.cu:1486): correctly
                     // rounded on the host once instead of a double square root on the update's serial tail every iteration
  const float4* x4;   // source xyz, ORIGINAL index
  const float4* xs4;  // source xyz, SORTED order (coalesced row reads)
  const int* xorder;
  const float4* ys4;  // target xyz (initial cloud), SORTED order
  const int* yorder;
  const int* yinv;    // original target index -> sorted position (inverse of yorder)
  float4* ycull;  // SORTED: {y~x, y~y, y~z, |y~|^2}, y~ = yt - centre; pads are {0,0,0,+inf}
  float4* xcull;  // SORTED: {-2x~x, -2x~y, -2x~z, thres_i + margin_i - |x~|^2}
  int* rowperm;    // [N] position -> sorted row
  int* iorig;      // [N] position -> ORIGINAL row index (the exports; stays valid after the clouds are gone)
  float4* gbox;   // [NGpad][2]: AABB of each sorted row group, grown by the group's cut-off radius
  float4* cellbox;  // [NGpad/16][2]: AABB of each cell of 16 row groups (level 1 of the scan)
  float4* sbox;   // [nslices][2]: AABB of each scan slice (T chunks)
  unsigned long long* masks;  // [nslices][N sorted rows][T] candidate bit masks; a row's T words of a slice are
                              // valid iff that slice's bit is set in the row's rowbits (never memset)
  unsigned* rowbits;          // [N sorted rows][rbw]: bit s set <=> the row has candidates in scan slice s;
                              // set by k_scan (returnless atomic OR), cleared by k_prep on a rebuild
  int* row_cnt;               // [N sorted rows]: candidates of the row in the bitmap (k_prep zeroes, k_scan adds)
  unsigned long long* tile_count;  // [1]: fine tiles executed so far this call (statistics)
  int* ovf_rows;   // [N]: positions of rows with more candidates than a list holds (handled by k_assoc_dense)
  unsigned long long* ovf_bits;  // [ceil(N / 64)]: bit p % 64 of word p / 64 <=> position p overflows (k_list's blocks ->
                                 // its last block, coherent stores / loads)
  // Long lists: the candidates of an overflow row (more than ASSOC_CAP, at most LONG_CAP), as the targets' sorted
  // positions in ascending ORIGINAL index, [overflow index q][LONG_CAP]; built and consumed by k_assoc_dense, valid while
  // long_stamp[q] == (call_serial << 24 | n_builds), call_serial = a process-wide serial of the align call (of the
  // submission, in a batch queue) this occupant of the workspace slot came with.  Null when positions do not fit 16 bits or CVO_NO_LONG_LISTS is set.
  unsigned short* long_j;
  unsigned long long* long_stamp;  // [N]
  unsigned long long call_serial;
  int max_iter;  // this pair's loop bound: DevParams::max_iter, or the smaller bound its submission to a batch queue came with
  PairState* st;
  cvo_trace_t* trace;
  int* status_out;  // mirror of st->status: the kernels' own early-exit word (device memory)
  int* want_out;    // mirror of st->want_full
  int* status_host;  // the same two words in pinned HOST memory: the device writes them when they change (posted
  int* want_host;    // writes), the host reads them after a chunk's event - no copy kernel between two chunks
  double* asum_host;  // pinned HOST memory: the sum of the kernel values of a single evaluation (inner_product_gpu), written by
                      // the block of k_assoc that finishes last - no k_update launch, no copy back
  int* gate;        // [1] blocks of k_list that finished the current rebuild (the last one validates the list)
  int* gate_flow;   // [1] blocks of k_assoc that stored their flow partial (the last one reduces them)
  // Rows evaluated by the wave-per-row kernels keep their ELL entries ROW-major, a run of min(candidates, K) entries in the
  // part of the ELL the thread-per-row kernels never touch (their rows hold at most 64 entries: slots >= 64 of the
  // slot-major matrix, (K_max - 64) * N entries).  A 16-byte entry per 64-byte line - what a column walk of the slot-major
  // matrix moves - made those kernels bandwidth-bound in a batch of clustered pairs (k_coeff_dense fetched 3.8x its bytes).
  // The runs are laid out by k_list at every rebuild (no atomics in the loop: 8000 rows of a lone pair on one counter cost
  // 80 us per iteration): dense_rel = the row's offset inside its 64-row word, word_base = the word's, word_base[words] = the
  // total; all rows of the pair or none - the total has to fit, a slot-major row beyond slot 63 would write into the runs.
  int* dense_rel;         // [N] by position (overflow rows only)
  int* ovf_wsum;          // [words] entries the overflow rows of a 64-row word need
  int* word_base;         // [words + 1]
  int* dense_off;         // [N] by position: where k_assoc_dense put the row THIS iteration (index of the run's first entry),
                          // -1 = slot-major; valid for rows whose nnz_row carries NNZ_DENSE_FLAG.  In the dense regime every
                          // row is evaluated there and the whole matrix is row-major: row pos at pos * K_max
};

// nnz_row[pos] of a row the wave-per-row kernels evaluated carries this flag (its entries may live row-major, dense_off)
constexpr unsigned NNZ_DENSE_FLAG = 0x80000000u;
__host__ __device__ inline unsigned nnz_count(unsigned v) { return v & ~NNZ_DENSE_FLAG; }
constexpr int ELL_LOWER_SLOTS = 64;  // slots of the slot-major matrix the thread-per-row kernels can reach (ASSOC_CAP16)
// index of entry `s` of the row at position `pos`: in its row-major run (off >= 0: the run's first entry) or in the slot-major matrix
__host__ __device__ inline size_t ell_index(int N, int s, int pos, int off) {
  return off >= 0 ? (size_t)off + (size_t)s : (size_t)s * (size_t)N + (size_t)pos;
}
__host__ __device__ inline size_t ell_upper_capacity(int N, int K_max) {
  return K_max > ELL_LOWER_SLOTS ? (size_t)(K_max - ELL_LOWER_SLOTS) * (size_t)N : 0;
}
constexpr int COEFF_SPLIT_MAX = 32;
constexpr int ROWS_PER_GROUP = 4;
// k_assoc_dense blocks per pair (one overflow row per wave at a time; 4 waves per block).  A batch launches few per pair (its pairs fill the chip and
// most of them have no overflow rows at all); a pair solved alone gets enough waves to fill it by itself: clustered
// clouds put thousands of rows on this path (profiles/r4/scene.txt).
constexpr int LONG_CAP = 1024;  // candidates a cached long list holds (rows beyond it are scanned literally)
constexpr int DENSE_BLOCKS_MIN = 64;
constexpr int DENSE_BLOCKS_MAX = 2048;
inline int dense_waves_for(int) { return 4; }
inline int dense_blocks_for(int N, int pairs_in_launch) {
  const int waves_pair = 8192 / (pairs_in_launch < 1 ? 1 : pairs_in_launch);  // 1024 SIMDs x 8 wave slots
  int nb = waves_pair / dense_waves_for(N);
  // one row per wave is the most there is to do - or, a small pair solved alone, one row per BLOCK (k_assoc_dense's wide rows).
  // (A lone pair's 2048 blocks are twice what the chip holds at once: the blocks that wait take the place of those that
  // finish, which deals the rows out by how long they take - a clustered 10k pair's first 600 iterations 66.2 -> 63.3 us
  // per iteration against two rows per wave of 1024 blocks.)
  const int rows = (pairs_in_launch <= 1 && N <= DENSE_BLOCKS_MAX / 2) ? N : (N + dense_waves_for(N) - 1) / dense_waves_for(N);
  if (nb > rows) nb = rows;
  if (nb > DENSE_BLOCKS_MAX) nb = DENSE_BLOCKS_MAX;
  if (nb < DENSE_BLOCKS_MIN) nb = DENSE_BLOCKS_MIN;
  return nb;
}

// ---- arithmetic conventions (DESIGN.md "Numerics") ------------------------------------------
__device__ __forceinline__ float dot3_dev(float a0, float a1, float a2, float b0, float b1, float b2) {
  return __builtin_fmaf(a0, b0, __builtin_fmaf(a1, b1, a2 * b2));
}
__device__ __forceinline__ float dop_dev(float a, float b, float c, float d) {
  return __builtin_fmaf(a, b, -(c * d));
}
struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 cross_dev(V3 a, V3 b) {
  return {dop_dev(a.y, b.z, a.z, b.y), dop_dev(a.z, b.x, a.x, b.z), dop_dev(a.x, b.y, a.y, b.x)};
}
struct M3 {
  float m[3][3];
};
__device__ __forceinline__ M3 matmul_dev(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++)
      r.m[i][j] = dot3_dev(a.m[i][0], a.m[i][1], a.m[i][2], b.m[0][j], b.m[1][j], b.m[2][j]);
  return r;
}
__device__ __forceinline__ V3 matvec_dev(const M3& a, V3 v) {
  return {dot3_dev(a.m[0][0], a.m[0][1], a.m[0][2], v.x, v.y, v.z),
          dot3_dev(a.m[1][0], a.m[1][1], a.m[1][2], v.x, v.y, v.z),
          dot3_dev(a.m[2][0], a.m[2][1], a.m[2][2], v.x, v.y, v.z)};
}

// transform_point_R_T (CvoGPU_impl.cu:31-82): R*p + T, R row-major here.
__device__ __forceinline__ V3 transform_point(const float* Ri, const float* Ti, float x, float y, float z) {
  return {dot3_dev(Ri[0], Ri[1], Ri[2], x, y, z) + Ti[0], dot3_dev(Ri[3], Ri[4], Ri[5], x, y, z) + Ti[1],
          dot3_dev(Ri[6], Ri[7], Ri[8], x, y, z) + Ti[2]};
}

// transform_point_pose_vec (CvoGPU_impl.cu:85-161): a 3x4 ROW-major pose times (x, y, z, 1).  Eigen evaluates the
// fixed-size 4-term inner product as (c0 + c1) + (c2 + c3); with nvcc's fmad contraction that is assumed to lower to
// fma(T0, x, T1*y) + fma(T2, z, T3) (the first product of each sum is fused; T3 * 1.0f folds).  The oracle mirrors
// exactly this form; like everything else here it is unpinned against a real reference build.
__device__ __forceinline__ V3 transform_point_pose_vec(const float* T, float x, float y, float z) {
  return {__builtin_fmaf(T[0], x, T[1] * y) + __builtin_fmaf(T[2], z, T[3]),
          __builtin_fmaf(T[4], x, T[5] * y) + __builtin_fmaf(T[6], z, T[7]),
          __builtin_fmaf(T[8], x, T[9] * y) + __builtin_fmaf(T[10], z, T[11])};
}

// ---- double-precision division and exp() of the row loops, restated so that their loop-invariant parts can be hoisted ----
//
// The reference evaluates `exp(-d2 / (2.0 * l * l))` per pair (CvoGPU.cu:552: double division, double exp) and
// `beta^3 / 6.0` per nonzero (CvoGPU.cu:1072).  hipcc expands an IEEE double division into 12 VALU instructions -
// v_div_scale x2, v_rcp_f64, two Newton steps (4 FMAs), q = n r, e = fma(-d, q, n), v_div_fmas (= fma(e, r, q)),
// v_div_fixup - of which the first nine depend on the DENOMINATOR only whenever v_div_scale leaves its operands
// alone, i.e. away from zero / denormal / huge operands and quotients.  rcp_refined() is that denominator part,
// instruction for instruction; div_by() is the numerator part.  Together they return what `n / d` returns, bit for bit,
// for every pair of operands the scaling and fix-up instructions would pass through unchanged - which the row
// constants (l in [ell_min, 1.2 ell_0] (1 + range / 500), den = 2 l^2) and the numerators (squared distances below the
// cut-off, float cubes) are.  cvo_debug_scalar_math ops 8 / 9 compare the two forms bit for bit on caller-supplied
// operands (tests/test_gpu_surface.py: 10^7 random operands from the real ranges, plus the edges); a numerator of +-0
// gives +0 here and -0 there, which exp() maps to the same 1.0 and the coefficient sums absorb (x + (+-0) = x).
// Theory for the constant case: with r = RN(1/d), q = RN(n r) is within one ulp of n / d and RN(q + r * RN(n - d q)) is
// the correctly rounded quotient (Markstein's theorem; Muller et al., Handbook of Floating-Point Arithmetic, 4.7).
__device__ __forceinline__ double rcp_refined(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
__device__ __forceinline__ double div_by(double n, double d, double r) {
  const double q = n * r;
  const double e = __builtin_fma(-d, q, n);
  return __builtin_fma(e, r, q);
}

// The same for the IEEE FLOAT division by a wave-uniform denominator (omega_i / c, v_i / d per row, CvoGPU.cu:784-787):
// hipcc's 12-instruction sequence is v_div_scale x2, v_rcp_f32, ONE Newton step, q = n r, two residual corrections, the
// second one inside v_div_fmas, v_div_fixup.  fdiv_prepare() is the denominator's part, fdiv_hoisted() the numerator's
// (5 instructions); identical to `n / d` wherever the scale / fix-up instructions pass their operands through, which for
// 2^-20 <= |d| <= 2^20 means: n == 0 (up to the sign of the zero, see div_by), or 2^-100 <= |n| < 2^60.  The caller
// checks exactly that (fdiv_operands_safe) and takes the plain division otherwise.  cvo_debug_scalar_math op 12.
struct FDivU {
  float d, r;
};
__device__ __forceinline__ FDivU fdiv_prepare(float d) {
  float r = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  return FDivU{d, r};
}
__device__ __forceinline__ float fdiv_hoisted(float n, const FDivU& u) {
  float q = n * u.r;
  float e = __builtin_fmaf(-u.d, q, n);
  q = __builtin_fmaf(e, u.r, q);
  e = __builtin_fmaf(-u.d, q, n);
  return __builtin_fmaf(e, u.r, q);
}
// every one of six numerators is zero or has 2^-100 <= |n| < 2^60 (and none is inf / NaN): frexp's exponent is 0 for a
// zero and e with |n| in [2^(e-1), 2^e) otherwise, denormals included; the sum of the magnitudes bounds each of them
// from above and turns inf / NaN into a failed comparison
__device__ __forceinline__ bool fdiv_operands_safe(const float (&n)[6]) {
  const float s = ((__builtin_fabsf(n[0]) + __builtin_fabsf(n[1])) + (__builtin_fabsf(n[2]) + __builtin_fabsf(n[3]))) +
                  (__builtin_fabsf(n[4]) + __builtin_fabsf(n[5]));
  int emin = __builtin_amdgcn_frexp_expf(n[0]);
#pragma unroll
  for (int q = 1; q < 6; q++) emin = min(emin, __builtin_amdgcn_frexp_expf(n[q]));
  return s < 0x1p60f && emin >= -99;
}

// compute_range_ell (CvoGPU.cu:86-90): (dist / 500.0 + 1.0) * ell in double; the division with its constant half folded
// (dist is a float square root: >= 0, never denormal as a double)
__device__ __forceinline__ float compute_range_ell(float ell, float dist) {
  return (float)((div_by((double)dist, 500.0, rcp_refined(500.0)) + 1.0) * (double)ell);
}

// exp(double) of the ROCm device library (ocml expD_base, what `exp()` compiled to inside the row loops), restated
// operation for operation with the library's constants so that the kernel value stays bit-identical to the `exp()`
// the previous rounds shipped (and the oracle's glibc exp within the 1 ulp DESIGN.md section 4 states):
//   n = rint(x log2 e);  r = fma(-ln2_lo, n, fma(-ln2_hi, n, x));  p = degree-11 Horner in r, closed by two fma(r, p, 1);
//   z = ldexp(p, n);  x > 1024 -> inf, x < -1075 -> 0.
// The Horner steps come out as three-operand v_fma_f64 (addends in SGPR pairs, ExpConsts): the compiler's own choice was the
// two-address v_fmac_f64 with a v_mov_b64 of the constant in front of every step (9 extra instructions per call) and
// 18 VGPRs of constants.  NONPOS: the caller guarantees x <= 0 (or NaN), which makes the overflow clamp dead.
// cvo_debug_scalar_math op 10 compares both variants with exp() bit for bit.
// The nine Horner addends, pinned to SGPR pairs ONCE, in front of the row loop (make_exp_consts): a step `fma(r, p, c)`
// whose addend lives in scalar registers can only be the three-operand v_fma_f64 (the two-address v_fmac_f64 wants its
// addend in the destination VGPR), and nothing of the pinning is left inside the loop.
struct ExpConsts {
  double c[9];
  double c11, c10;  // the first step, fma(r, c11, c10): one of its two constants in a VGPR pair (one SGPR operand per VOP3)
};
__device__ __forceinline__ ExpConsts make_exp_consts() {
  ExpConsts k = {{0x1.71dee623fde64p-19, 0x1.a01997c89e6b0p-16, 0x1.a01a014761f6ep-13, 0x1.6c16c1852b7b0p-10,
                  0x1.1111111122322p-7, 0x1.55555555502a1p-5, 0x1.5555555555511p-3, 0x1.000000000000bp-1, 1.0},
                 0x1.ade156a5dcb37p-26, 0x1.28af3fca7ab0cp-22};
#pragma unroll
  for (int q = 0; q < 8; q++) asm("" : "+s"(k.c[q]));  // (1.0 is an inline constant)
  asm("" : "+v"(k.c11));
  asm("" : "+s"(k.c10));
  return k;
}
template <bool NONPOS>
__device__ __forceinline__ double exp_ocml(double x, const ExpConsts& k) {
  const double n = __builtin_rint(x * 0x1.71547652b82fep+0);
  double r = __builtin_fma(-0x1.62e42fefa39efp-1, n, x);
  r = __builtin_fma(-0x1.abc9e3b39803fp-56, n, r);
  double p = __builtin_fma(r, k.c11, k.c10);
#pragma unroll
  for (int q = 0; q < 9; q++) p = __builtin_fma(r, p, k.c[q]);
  p = __builtin_fma(r, p, 1.0);
  double z = __builtin_ldexp(p, (int)n);
  if (!NONPOS) z = x > 1024.0 ? __builtin_inf() : z;
  z = x < -1075.0 ? 0.0 : z;
  return z;
}

// The matrices of compute_step_size_xi (CvoGPU.cu:953-998), which depend only on (omega, v).
struct XiMats {
  M3 m2, m3, m4;
  V3 ohv, m2v, m3v;
  float omega[3], v[3];
};
__device__ inline void xi_mats(const float* omega, const float* v, XiMats& x) {
  M3 oh;
  oh.m[0][0] = 0;          oh.m[0][1] = -omega[2]; oh.m[0][2] = omega[1];
  oh.m[1][0] = omega[2];   oh.m[1][1] = 0;         oh.m[1][2] = -omega[0];
  oh.m[2][0] = -omega[1];  oh.m[2][1] = omega[0];  oh.m[2][2] = 0;
  V3 vv{v[0], v[1], v[2]};
  x.m2 = matmul_dev(oh, oh);
  x.m3 = matmul_dev(x.m2, oh);
  x.m4 = matmul_dev(x.m3, oh);
  x.ohv = matvec_dev(oh, vv);
  x.m2v = matvec_dev(x.m2, vv);
  x.m3v = matvec_dev(x.m3, vv);
  for (int i = 0; i < 3; i++) {
    x.omega[i] = omega[i];
    x.v[i] = v[i];
  }
}

// ---- scalar host-side maths of the reference, run by one device thread ---------------------

#define CUBIC_QUAL __device__ inline
#define CUBIC_NAME cubic_roots
#define CUBIC_FABS fabs
#define CUBIC_SQRT sqrt
#define CUBIC_ISFINITE isfinite
#define CUBIC_NAN __builtin_nan("")
// Roots of p0 x^3 + p1 x^2 + p2 x + p3.  The reference takes the eigenvalues of the companion
// matrix with Eigen 3.3.9's EigenSolver (LieGroup.cpp:309-325; Eigen is not in the repository).
// Restated with a backward-stable scheme that uses only + - * / sqrt (so CPU and GPU agree
// bitwise): each real root is bracketed next to a critical point of the monic cubic (the outer
// brackets grow by doubling) and refined by safeguarded Newton (bisection fallback); a complex
// pair follows from Vieta.  Pinned against numpy.roots (LAPACK companion eigenvalues) in
// tests/test_oracle_math.py.
CUBIC_QUAL double cubic_solve_bracket(double a, double b, double c, double lo, double hi) {
  auto f = [&](double x) { return ((x + a) * x + b) * x + c; };
  auto df = [&](double x) { return (3.0 * x + 2.0 * a) * x + b; };
  double fl = f(lo), fh = f(hi);
  if (fl == 0.0) return lo;
  if (fh == 0.0) return hi;
  double xl, xh;
  if (fl < 0.0) {
    xl = lo;
    xh = hi;
  } else {
    xl = hi;
    xh = lo;
  }
  double x = 0.5 * (lo + hi);
  double dxold = CUBIC_FABS(hi - lo), dx = dxold;
  double fx = f(x), dfx = df(x);
  for (int it = 0; it < 200; it++) {
    if ((((x - xh) * dfx - fx) * ((x - xl) * dfx - fx) > 0.0) || (CUBIC_FABS(2.0 * fx) > CUBIC_FABS(dxold * dfx))) {
      dxold = dx;
      dx = 0.5 * (xh - xl);
      x = xl + dx;
      if (xl == x) return x;
    } else {
      dxold = dx;
      dx = fx / dfx;
      const double tmp = x;
      x -= dx;
      if (tmp == x) return x;
    }
    if (CUBIC_FABS(dx) <= 4.5e-16 * CUBIC_FABS(x)) return x;  // converged to ~2 ulp (Newton can ping-pong there)
    fx = f(x);
    dfx = df(x);
    if (fx == 0.0) return x;
    if (fx < 0.0)
      xl = x;
    else
      xh = x;
  }
  return x;
}

// Root of the monic cubic on the unbounded side of x0 (dir = +1: right of x0 where f(x0) <= 0 and f
// increases to +inf; dir = -1: left of x0 where f(x0) >= 0 and f decreases to -inf).  The bracket is
// grown by doubling so that Newton starts within a factor ~2 of the root.
CUBIC_QUAL double cubic_solve_outward(double a, double b, double c, double x0, double dir, double bound) {
  auto f = [&](double x) { return ((x + a) * x + b) * x + c; };
  double h = CUBIC_FABS(x0) * 0.5;
  if (h < 1e-3) h = 1e-3;
  double prev = x0;
  for (int it = 0; it < 1100; it++) {
    double x = x0 + dir * h;
    if (CUBIC_FABS(x) > bound) x = dir * bound;
    const double fx = f(x);
    if ((dir > 0.0) ? (fx >= 0.0) : (fx <= 0.0)) return dir > 0.0 ? cubic_solve_bracket(a, b, c, prev, x) : cubic_solve_bracket(a, b, c, x, prev);
    if (CUBIC_FABS(x) >= bound) return x;  // cannot happen for a finite cubic (Cauchy bound)
    prev = x;
    h *= 2.0;
  }
  return prev;
}

CUBIC_QUAL void CUBIC_NAME(const double coef[4], double re[3], double im[3]) {
  const double nan = CUBIC_NAN;
  const double a = coef[1] / coef[0], b = coef[2] / coef[0], c = coef[3] / coef[0];
  if (!CUBIC_ISFINITE(a) || !CUBIC_ISFINITE(b) || !CUBIC_ISFINITE(c)) {
    for (int i = 0; i < 3; i++) re[i] = im[i] = nan;
    return;
  }
  auto f = [&](double x) { return ((x + a) * x + b) * x + c; };
  double bound = CUBIC_FABS(a);
  if (CUBIC_FABS(b) > bound) bound = CUBIC_FABS(b);
  if (CUBIC_FABS(c) > bound) bound = CUBIC_FABS(c);
  bound = 1.0 + bound;  // Cauchy bound on |root|
  double r[3] = {0.0, 0.0, 0.0};
  int nr = 0;
  const double dq = a * a - 3.0 * b;
  if (!(dq > 0.0)) {  // monotone: one real root, on the side of the inflection point the sign says
    const double xi = -a / 3.0;
    const double fi = f(xi);
    r[nr++] = (fi == 0.0) ? xi : (fi < 0.0 ? cubic_solve_outward(a, b, c, xi, 1.0, bound)
                                           : cubic_solve_outward(a, b, c, xi, -1.0, bound));
  } else {
    const double s = CUBIC_SQRT(dq);
    const double t = (a >= 0.0) ? (-a - s) : (-a + s);
    const double xa = t / 3.0, xb = (t != 0.0) ? b / t : 0.0;
    const double x1 = xa < xb ? xa : xb, x2 = xa < xb ? xb : xa;
    const double f1 = f(x1), f2 = f(x2);
    if (f1 >= 0.0) r[nr++] = (f1 == 0.0) ? x1 : cubic_solve_outward(a, b, c, x1, -1.0, bound);
    if (f1 > 0.0 && f2 < 0.0) r[nr++] = cubic_solve_bracket(a, b, c, x1, x2);
    if (f2 <= 0.0) r[nr++] = (f2 == 0.0) ? x2 : cubic_solve_outward(a, b, c, x2, 1.0, bound);
  }
  if (nr == 3) {
    for (int i = 0; i < 3; i++) {
      re[i] = r[i];
      im[i] = 0.0;
    }
  } else if (nr == 2) {  // an exact double root at a critical point
    re[0] = r[0];
    re[1] = r[1];
    const double dbl = -a - r[0] - r[1];
    re[2] = dbl;
    im[0] = im[1] = im[2] = 0.0;
  } else {
    const double r0 = r[0];
    // complex pair z, conj(z) from Vieta: r0 + 2 Re z = -a, 2 r0 Re z + |z|^2 = b, r0 |z|^2 = -c;
    // use the pair of relations that does not cancel for the size of r0
    double rr, mod2;
    if (CUBIC_FABS(r0) >= 0.5 * CUBIC_FABS(a) && r0 != 0.0) {
      mod2 = -c / r0;
      rr = (b - mod2) / (2.0 * r0);
    } else {
      rr = 0.5 * (-a - r0);
      mod2 = b - 2.0 * r0 * rr;
    }
    const double ii = mod2 - rr * rr;
    re[0] = r0;
    im[0] = 0.0;
    re[1] = re[2] = rr;
    im[1] = ii > 0.0 ? CUBIC_SQRT(ii) : 0.0;
    im[2] = -im[1];
  }
}

// ---- the same roots with the independent real-root searches spread over lanes 0..2 of the calling wave ----
// Doubling phase of cubic_solve_outward, statement for statement; returns true with the bracket [lo, hi] that
// cubic_solve_bracket has to refine, or false with the final value.
__device__ inline bool cubic_outward_bracket(double a, double b, double c, double x0, double dir, double bound, double& lo,
                                             double& hi, double& val) {
  auto f = [&](double x) { return ((x + a) * x + b) * x + c; };
  double h = fabs(x0) * 0.5;
  if (h < 1e-3) h = 1e-3;
  double prev = x0;
  for (int it = 0; it < 1100; it++) {
    double x = x0 + dir * h;
    if (fabs(x) > bound) x = dir * bound;
    const double fx = f(x);
    if ((dir > 0.0) ? (fx >= 0.0) : (fx <= 0.0)) {
      if (dir > 0.0) {
        lo = prev;
        hi = x;
      } else {
        lo = x;
        hi = prev;
      }
      return true;
    }
    if (fabs(x) >= bound) {
      val = x;
      return false;
    }
    prev = x;
    h *= 2.0;
  }
  val = prev;
  return false;
}

// cubic_roots for a whole wave (all 64 lanes call with the same coefficients and get the same result).  Every
// real root comes out of exactly the scalar code above - only who executes it changes: the up to three searches
// run side by side on lanes 0..2 (first their doubling phases, then their Newton refinements), which takes the
// time of the slowest one instead of the sum.
__device__ inline void cubic_roots_wave(const double coef[4], double re[3], double im[3]) {
  const int lane = (int)(threadIdx.x & 63);
  const double nan = __builtin_nan("");
  const double a = coef[1] / coef[0], b = coef[2] / coef[0], c = coef[3] / coef[0];
  if (!isfinite(a) || !isfinite(b) || !isfinite(c)) {
    for (int i = 0; i < 3; i++) re[i] = im[i] = nan;
    return;
  }
  auto f = [&](double x) { return ((x + a) * x + b) * x + c; };
  double bound = fabs(a);
  if (fabs(b) > bound) bound = fabs(b);
  if (fabs(c) > bound) bound = fabs(c);
  bound = 1.0 + bound;
  // The (up to) three searches cubic_roots runs one after the other, each on its own lane: kind 1 = the value p
  // itself, 2 = outward search from p in direction q, 3 = bracket [p, q], 0 = nothing.  Lane 0: the root left of the
  // first critical point (or the single root of a monotone cubic), lane 1: the root between the critical points,
  // lane 2: the root right of the second one; r[] lists the results that exist in that order, as cubic_roots does.
  // (No run-time indexed task list: that would live in scratch memory.)
  bool hasA = false, hasB = false, hasC = false;
  int kindA = 0, kindC = 0;
  double pA = 0, qA = 0, pB = 0, qB = 0, pC = 0, qC = 0;
  const double dq = a * a - 3.0 * b;
  if (!(dq > 0.0)) {
    const double xi = -a / 3.0;
    const double fi = f(xi);
    hasA = true;
    kindA = fi == 0.0 ? 1 : 2;
    pA = xi;
    qA = fi < 0.0 ? 1.0 : -1.0;
  } else {
    const double s = sqrt(dq);
    const double t = (a >= 0.0) ? (-a - s) : (-a + s);
    const double xa = t / 3.0, xb = (t != 0.0) ? b / t : 0.0;
    const double x1 = xa < xb ? xa : xb, x2 = xa < xb ? xb : xa;
    const double f1 = f(x1), f2 = f(x2);
    hasA = f1 >= 0.0;
    kindA = f1 == 0.0 ? 1 : 2;
    pA = x1;
    qA = -1.0;
    hasB = f1 > 0.0 && f2 < 0.0;
    pB = x1;
    qB = x2;
    hasC = f2 <= 0.0;
    kindC = f2 == 0.0 ? 1 : 2;
    pC = x2;
    qC = 1.0;
  }
  const int kind = lane == 0 ? (hasA ? kindA : 0) : (lane == 1 ? (hasB ? 3 : 0) : (lane == 2 ? (hasC ? kindC : 0) : 0));
  const double p = lane == 0 ? pA : (lane == 1 ? pB : pC), q = lane == 0 ? qA : (lane == 1 ? qB : qC);
  double val = 0.0, lo = 0.0, hi = 0.0;
  bool refine = false;
  if (kind == 1) {
    val = p;
  } else if (kind == 2) {
    refine = cubic_outward_bracket(a, b, c, p, q, bound, lo, hi, val);
  } else if (kind == 3) {
    lo = p;
    hi = q;
    refine = true;
  }
  if (refine) val = cubic_solve_bracket(a, b, c, lo, hi);
  const double vA = __shfl(val, 0), vB = __shfl(val, 1), vC = __shfl(val, 2);
  const int nr = (hasA ? 1 : 0) + (hasB ? 1 : 0) + (hasC ? 1 : 0);
  double r[3];
  r[0] = hasA ? vA : (hasB ? vB : vC);
  r[1] = hasA ? (hasB ? vB : vC) : vC;
  r[2] = vC;
  if (nr == 3) {
    for (int i = 0; i < 3; i++) {
      re[i] = r[i];
      im[i] = 0.0;
    }
  } else if (nr == 2) {
    re[0] = r[0];
    re[1] = r[1];
    re[2] = -a - r[0] - r[1];
    im[0] = im[1] = im[2] = 0.0;
  } else {
    const double r0 = r[0];
    double rr, mod2;
    if (fabs(r0) >= 0.5 * fabs(a) && r0 != 0.0) {
      mod2 = -c / r0;
      rr = (b - mod2) / (2.0 * r0);
    } else {
      rr = 0.5 * (-a - r0);
      mod2 = b - 2.0 * r0 * rr;
    }
    const double ii = mod2 - rr * rr;
    re[0] = r0;
    im[0] = 0.0;
    re[1] = re[2] = rr;
    im[1] = ii > 0.0 ? sqrt(ii) : 0.0;
    im[2] = -im[1];
  }
}

// Exact shortcut for the iterations whose step is NOT clamped at min_step (the first ~hundred of every BASELINE shape,
// every iteration of a warm-started tracking solve or of the demo pair): the bracketing solve below costs ~7 000 cycles
// of one wave on the pair's serial tail, and all that survives of it is ONE float - the smallest admissible root, clamped
// and rounded (CvoGPU.cu:1136-1158).  Here: g = sign(B) p, so g(0) > 0; Newton from 0 (an approximate reciprocal is
// enough: the iteration corrects itself) finds the first root r where g decreases through zero, and a CERTIFICATE in
// exact-arithmetic terms says that whatever cubic_roots returns selects the same float:
//   * g(r (1 - d)) > noise and g(r (1 + d)) < -noise with d = 2^-38 and noise = 1e-14 x the magnitude sum of the Horner
//     evaluation (45 eps): a simple root of the real polynomial lies in that interval, and every x where the MONIC
//     polynomial cubic_roots works with can evaluate to zero or change sign does too;
//   * g' < 0 on [0, r (1 + d)] - at both ends and, where the parabola g' opens downwards, at its vertex if that lies in
//     between - hence no other real root below r, and |g'| >= 4e-10 |g3| on it, hence no complex pair u +- iv with
//     0 < u < r and |v| < 1e-5 either (g'(u) = g3 v^2 for such a pair): the reference's `fabs(im) < 1e-5` rule (1142)
//     cannot admit anything smaller;
//   * both ends of the interval round to the same float and fall on the same side of min_step and max_step.
// "No root up to max_step" is certified the same way (g > noise at max_step (1 + d), g' < 0 up to there) and returns
// max_step, as does temp_step = DBL_MAX upstream.  Anything else - an increasing start, a sign the noise could flip, a root
// within 2^-38 of a float boundary (1e-4 of the iterations), leading coefficients cubic_roots would turn into NaN - returns
// false and the caller runs the full solve.  `path` (tests): 1 = root, 2 = max_step, 0 = not taken.
__device__ inline bool step_newton_certified(const double p[4], float min_step, float max_step, float* step_out, int* path) {
  if (path) *path = 0;
  const double a0 = fabs(p[0]), a1 = fabs(p[1]), a2 = fabs(p[2]), a3 = fabs(p[3]);
  if (!(min_step > 0.f && min_step <= max_step && max_step <= 100.f && a0 >= 1e-200 && a0 <= 1e100 && a1 <= 1e100 && a2 <= 1e100 &&
        a3 <= 1e100 && a3 > 0.0))
    return false;
  const double sg = p[3] > 0.0 ? 1.0 : -1.0;
  const double g3 = sg * p[0], g2 = sg * p[1], g1 = sg * p[2], g0 = a3;
  if (!(g1 < 0.0)) return false;  // (g has to leave 0 downwards)
  auto g = [&](double x) { return __builtin_fma(__builtin_fma(__builtin_fma(g3, x, g2), x, g1), x, g0); };
  auto dg = [&](double x) { return __builtin_fma(__builtin_fma(3.0 * g3, x, 2.0 * g2), x, g1); };
  auto noise = [&](double x) { return 1e-14 * __builtin_fma(__builtin_fma(__builtin_fma(a0, x, a1), x, a2), x, a3); };
  const double DELTA = 0x1p-38;
  const double dmin = 4e-10 * a0;  // |g'| below this could hide a complex pair with |imag| < 1e-5
  // g' < 0 on [0, x1], clear of dmin (see above)
  // (g' is a parabola: over an interval it is largest at an end or - when it opens downwards, g3 < 0 - at its vertex
  // -g2 / (3 g3) if that lies inside; g' is flat there, so the rounding of the vertex does not matter)
  auto decreasing_up_to = [&](double x1) {
    if (!(g1 < -dmin) || !(dg(x1) < -dmin)) return false;
    if (g3 < 0.0) {
      const double xv = -g2 / (3.0 * g3);
      if (xv > 0.0 && xv < x1 && !(dg(xv) < -dmin)) return false;
    }
    return true;
  };
  const double xcap = (double)max_step * (1.0 + 0x1p-30);
  double x = 0.0;
  bool conv = false;
  for (int it = 0; it < 40; it++) {
    const double f = g(x), d = dg(x);
    if (!(d < 0.0)) return false;
    const double dx = f * __builtin_amdgcn_rcp(d);
    double xn = x - dx;
    if (!(xn == xn) || !(xn > 0.0)) return false;
    if (xn > xcap) {
      const double gc = g(xcap);
      if (gc > noise(xcap)) {  // no root up to max_step?
        if (!decreasing_up_to(xcap)) return false;
        *step_out = max_step;
        if (path) *path = 2;
        return true;
      }
      if (x == xcap) return false;  // (already restarted from there once)
      x = xcap;                     // the root is inside: come back from the right end
      continue;
    }
    conv = fabs(dx) <= 0x1p-42 * xn;
    x = xn;
    if (conv) break;
  }
  if (!conv) return false;
  const double lo = x * (1.0 - DELTA), hi = x * (1.0 + DELTA);
  if (!(g(lo) > noise(lo)) || !(g(hi) < -noise(hi)) || !decreasing_up_to(hi)) return false;
  const float flo = (float)lo, fhi = (float)hi;
  if (flo != fhi) return false;
  const double mn = (double)min_step, mx = (double)max_step;
  float st;
  if (lo > mx)
    st = max_step;
  else if (hi < mn)
    st = min_step;
  else if (lo >= mn && hi <= mx)
    st = flo;
  else
    return false;
  *step_out = st;
  if (path) *path = 1;
  return true;
}

// compute_step_size host half (CvoGPU.cu:1122-1158), overwrite quirk included.  FAST = false: without the certified
// Newton shortcut (cvo_debug_scalar_math runs both and the tests compare the bits).
template <bool WAVE, bool FAST = true>
__device__ inline float select_step(double B, double C, double D, double E, float min_step, float max_step, int* path = nullptr) {
  const double DMAX = 1.7976931348623157e308;
  double p_coef[4] = {4.0 * E, 3.0 * D, 2.0 * C, B};
  {
    // Exact shortcut for the common end game (the loop spends most of its iterations clamped at min_step): if the cubic
    // changes sign between 0 and min_step it has a real root there, so the smallest admissible root - whichever it is -
    // lies below min_step and the clamp returns min_step; no root has to be computed.  Signs are only trusted clear of
    // the rounding noise of the Horner evaluation, and only where cubic_roots would not have given up (finite monic
    // coefficients); everything else takes the full solve below.
    const double ms = (double)min_step;
    const double a0 = fabs(p_coef[0]), a1 = fabs(p_coef[1]), a2 = fabs(p_coef[2]), a3 = fabs(p_coef[3]);
    if (min_step > 0.f && min_step <= max_step && a0 >= 1e-200 && a0 <= 1e100 && a1 <= 1e100 && a2 <= 1e100 && a3 <= 1e100) {
      const double fm = ((p_coef[0] * ms + p_coef[1]) * ms + p_coef[2]) * ms + p_coef[3];
      const double noise = 3.6e-15 * (((a0 * ms + a1) * ms + a2) * ms + a3);  // 16 eps times the magnitude sum
      if ((p_coef[3] > 0.0 && fm < -noise) || (p_coef[3] < 0.0 && fm > noise)) return min_step;
    }
  }
  if (FAST) {
    float st;
    if (step_newton_certified(p_coef, min_step, max_step, &st, path)) return st;
  }
  double re[3], im[3];
  if (WAVE)
    cubic_roots_wave(p_coef, re, im);  // all lanes of the wave are here
  else
    cubic_roots(p_coef, re, im);
  double temp_step = DMAX;
  for (int i = 0; i < 3; i++)
    if (re[i] > 0 && re[i] < temp_step && fabs(im[i]) < 1e-5) temp_step = re[i];
  float step;
  if (temp_step > (double)max_step)
    step = max_step;
  else if (temp_step < (double)min_step)
    step = min_step;
  else
    step = (float)temp_step;
  return step;
}

// Exp_SEK3 (LieGroup.cpp:244-274); out = 3x4 row-major [R | Jl v].
__device__ inline void exp_sek3(const float xi[6], float dt, float out[12]) {
  const float TOLERANCE = 1e-6f;
  float w0 = xi[0], w1 = xi[1], w2 = xi[2];
  float theta = sqrtf(w0 * w0 + (w1 * w1 + w2 * w2));
  float R[3][3], Jl[3][3];
  const float I[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (theta < TOLERANCE) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[i][j] = Jl[i][j] = I[i][j];
  } else {
    float A[3][3] = {{0, -w2, w1}, {w2, 0, -w0}, {-w1, w0, 0}};
    float theta2 = theta * theta;
    float stheta = sinf(dt * theta);
    float ctheta = cosf(dt * theta);
    float oneMinusCosTheta2 = (1 - ctheta) / (theta2);
    float A2[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) A2[i][j] = A[i][0] * A[0][j] + (A[i][1] * A[1][j] + A[i][2] * A[2][j]);
    float s1 = stheta / theta;
    float s3 = (dt * theta - stheta) / (theta2 * theta);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        R[i][j] = (I[i][j] + s1 * A[i][j]) + oneMinusCosTheta2 * A2[i][j];
        Jl[i][j] = (dt * I[i][j] + oneMinusCosTheta2 * A[i][j]) + s3 * A2[i][j];
      }
  }
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) out[4 * i + j] = R[i][j];
    out[4 * i + 3] = Jl[i][0] * xi[3] + (Jl[i][1] * xi[4] + Jl[i][2] * xi[5]);
  }
}

// ||Sophus::SE3d(dRT).log()|| (CvoGPU.cu:1473-1476), Sophus 1.0.0 algorithm.
__device__ inline double se3_log_norm(const double R[9], const double t[3]) {
  const double eps = 1e-10;
  const double PI = 3.14159265358979323846;
  double q[4];
  double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    double s = sqrt(tr + 1.0);
    q[3] = 0.5 * s;
    s = 0.5 / s;
    q[0] = (R[7] - R[5]) * s;
    q[1] = (R[2] - R[6]) * s;
    q[2] = (R[3] - R[1]) * s;
  } else {
    // Eigen's quaternion-from-matrix, branch tr <= 0: i = index of the largest diagonal entry, j = i + 1, k = j + 1
    // (mod 3).  Spelled out per case: run-time indices into q / R would put both arrays into scratch memory.
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > (i == 0 ? R[0] : R[4])) i = 2;
    if (i == 0) {
      double s = sqrt(R[0] - R[4] - R[8] + 1.0);
      q[0] = 0.5 * s;
      s = 0.5 / s;
      q[3] = (R[7] - R[5]) * s;
      q[1] = (R[3] + R[1]) * s;
      q[2] = (R[6] + R[2]) * s;
    } else if (i == 1) {
      double s = sqrt(R[4] - R[8] - R[0] + 1.0);
      q[1] = 0.5 * s;
      s = 0.5 / s;
      q[3] = (R[2] - R[6]) * s;
      q[2] = (R[7] + R[5]) * s;
      q[0] = (R[1] + R[3]) * s;
    } else {
      double s = sqrt(R[8] - R[0] - R[4] + 1.0);
      q[2] = 0.5 * s;
      s = 0.5 / s;
      q[3] = (R[3] - R[1]) * s;
      q[0] = (R[2] + R[6]) * s;
      q[1] = (R[5] + R[7]) * s;
    }
  }
  double squared_n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  double n = sqrt(squared_n);
  double w = q[3];
  double two_atan_nbyw_by_n;
  if (n < eps) {
    double squared_w = w * w;
    two_atan_nbyw_by_n = 2.0 / w - 2.0 * squared_n / (w * squared_w);
  } else {
    if (fabs(w) < eps) {
      two_atan_nbyw_by_n = (w > 0 ? PI : -PI) / n;
    } else {
      two_atan_nbyw_by_n = 2.0 * atan(n / w) / n;
    }
  }
  double theta = two_atan_nbyw_by_n * n;
  double om[3] = {two_atan_nbyw_by_n * q[0], two_atan_nbyw_by_n * q[1], two_atan_nbyw_by_n * q[2]};
  double O[3][3] = {{0, -om[2], om[1]}, {om[2], 0, -om[0]}, {-om[1], om[0], 0}};
  double O2[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) O2[i][j] = O[i][0] * O[0][j] + O[i][1] * O[1][j] + O[i][2] * O[2][j];
  double coef;
  if (fabs(theta) < eps) {
    coef = 1.0 / 12.0;
  } else {
    double half_theta = 0.5 * theta;
    coef = (1.0 - theta * cos(half_theta) / (2.0 * sin(half_theta))) / (theta * theta);
  }
  double s = 0;
  for (int i = 0; i < 3; i++) {
    double u = 0;
    for (int j = 0; j < 3; j++) {
      double Vinv = (i == j ? 1.0 : 0.0) - 0.5 * O[i][j] + coef * O2[i][j];
      u += Vinv * t[j];
    }
    s += u * u;
  }
  for (int i = 0; i < 3; i++) s += om[i] * om[i];
  return sqrt(s);
}

// A_sparsity_indicator_ell_update (CvoGPU.cu:1167-1285), FIFOs as ring buffers.
// e_front / s_front: eq[st->e_head] / sq[st->s_head] as they were on entry, read ahead by the caller (two dependent
// cold loads in the middle of the serial tail otherwise; the pushes below never touch the heads, window + 1 < IND_CAP)
__device__ inline bool indicator_update(PairState* st, float* sq, float* eq, float indicator, int queue_len,
                                        float thr, float e_front, float s_front) {
  bool decrease = false;
  auto s_push = [&](float x) {
    sq[(st->s_head + st->s_size) % IND_CAP] = x;
    st->s_size++;
  };
  auto e_push = [&](float x) {
    eq[(st->e_head + st->e_size) % IND_CAP] = x;
    st->e_size++;
  };
  if (st->s_size < queue_len) {
    s_push(indicator);
    st->s_sum += indicator;
  }
  if (st->s_size >= queue_len && st->e_size < queue_len) {
    e_push(indicator);
    st->e_sum += indicator;
  }
  if (st->s_size >= queue_len && st->e_size >= queue_len) {
    if (st->e_sum / st->s_sum > 1 - thr && st->e_sum / st->s_sum < 1 + thr) {
      decrease = true;
      st->s_head = st->s_size = st->e_head = st->e_size = 0;
      st->s_sum = 0;
      st->e_sum = 0;
    } else {
      float ef = e_front;
      st->e_sum -= ef;
      st->s_sum += ef;
      s_push(ef);
      st->e_head = (st->e_head + 1) % IND_CAP;
      st->e_size--;
      st->s_sum -= s_front;
      st->s_head = (st->s_head + 1) % IND_CAP;
      st->s_size--;
      e_push(indicator);
      st->e_sum += indicator;
    }
  }
  return decrease;
}

__device__ inline void update_tf(const float R[9], const float T[3], float Ri[9], float Ti[3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Ri[3 * i + j] = R[3 * j + i];
  for (int i = 0; i < 3; i++) {
    float a0 = (-Ri[3 * i + 0]) * T[0], a1 = (-Ri[3 * i + 1]) * T[1], a2 = (-Ri[3 * i + 2]) * T[2];
    Ti[i] = a0 + (a1 + a2);
  }
}

}  // namespace cvo_dev
