// cvo_launch.hip -- every kernel launch of the solver: scan geometry, the launch wrappers of the row-block kernels, LaunchGeom, one iteration (launch_core) and one chunk of iterations (launch_chunk) as the graphs capture them.
// A SECTION of the one translation unit cvo_hip.hip (which includes the sections in dependency order and says why it is one
// unit); not compiled on its own.  Shared declarations: cvo_internal.h.
#ifndef CVO_COEFF_DENSE_MULTI_FROM
#define CVO_COEFF_DENSE_MULTI_FROM 8  // pairs per launch from which k_coeff_dense takes eight rows per wave
#endif

namespace {

void choose_scan_config(const cvo_ctx* ctx, int n_pairs, int NG, int Mpad, int* T_out, int* gpb_out) {
  const int T = 2;  // (1 / 4 / 8 were swept in round 2: the k_scan<T> instantiations remain)
  // Measured on MI355X (64 x 10k x 10k, T = 2): one row segment per wave (5120 waves) beats 128-group
  // blocks by 1.4x; a single pair needs the row range split to fill the chip.  Rule: the fewest
  // segments that still give ~4096 waves.
  const long slices = Mpad / (64 * T);
  const int ngr = (int)align_up((size_t)NG, 64);
  int gpb = ngr;
  while (gpb > 64) {
    const long waves = slices * ((ngr + gpb - 1) / gpb) * n_pairs;
    if (waves >= 4096) break;
    gpb = (int)align_up((size_t)gpb / 2, 64);
  }
  *T_out = T;
  *gpb_out = gpb;
}

void launch_scan(hipStream_t s, int T, dim3 grid, const PairDesc* descs, const DevParams* dp, const PairState* st, int force) {
  switch (T) {
    case 1: hipLaunchKernelGGL(k_scan<1>, grid, dim3(256), 0, s, descs, dp, st, force); break;
    case 2: hipLaunchKernelGGL(k_scan<2>, grid, dim3(256), 0, s, descs, dp, st, force); break;
    case 4: hipLaunchKernelGGL(k_scan<4>, grid, dim3(256), 0, s, descs, dp, st, force); break;
    default: hipLaunchKernelGGL(k_scan<8>, grid, dim3(256), 0, s, descs, dp, st, force); break;
  }
}

// 1-D grid of the XCD-aware row-block kernels (see pair_block)
inline dim3 row_grid(int nblk, int n_pairs) { return dim3((unsigned)(nblk * ((n_pairs + 7) / 8 * 8))); }

void launch_list(hipStream_t s, bool idx16, int N, int n_pairs, const PairDesc* descs, const DevParams* dp,
                 const PairState* st) {
  const int nblk = (N + LIST_THREADS - 1) / LIST_THREADS;
  const dim3 blk(LIST_THREADS), grid = row_grid(nblk, n_pairs);
  if (idx16)
    hipLaunchKernelGGL((k_list<unsigned short, ASSOC_CAP16>), grid, blk, 0, s, descs, dp, st, nblk, n_pairs);
  else
    hipLaunchKernelGGL((k_list<int, ASSOC_CAP32>), grid, blk, 0, s, descs, dp, st, nblk, n_pairs);
}

// Where the workspaces of a launch's pairs are (kernel arguments of the row-block kernels, see row_off_*)
struct ArenaArg {
  const char* base;    // workspace of the launch's first pair
  unsigned stride256;  // bytes / 256 between consecutive pairs
  int Npad;
};

// instr: the instantiation with time stamps (CVO_KERNEL_CLOCK / CVO_PHASE_TICKS); the production kernels have none
template <typename IdxT, int CAP, int FEAT>
void launch_assoc_t(hipStream_t s, bool instr, dim3 grid, const PairDesc* descs, const DevParams* dp, const PairState* st,
                    const ArenaArg& A, int packed) {
  const dim3 blk(ASSOC_THREADS);
  if (instr)
    hipLaunchKernelGGL((k_assoc<IdxT, CAP, FEAT, true>), grid, blk, 0, s, descs, dp, st, A.base, packed, A.stride256, A.Npad);
  else
    hipLaunchKernelGGL((k_assoc<IdxT, CAP, FEAT, false>), grid, blk, 0, s, descs, dp, st, A.base, packed, A.stride256, A.Npad);
}

// feat: FEAT_GEO / FEAT_ALL / FEAT_COL / FEAT_HOT (cvo_pair_math.h), chosen per call by call_feat()
void launch_assoc(hipStream_t s, bool idx16, int feat, bool instr, int nblk, int n_pairs, const PairDesc* descs,
                  const DevParams* dp, const PairState* st, const ArenaArg& A, int lean) {
  const dim3 grid = row_grid(nblk, n_pairs);
  const int packed = (lean & 0xf) | (nblk << 4) | (int)((unsigned)n_pairs << 20);  // (ensure_workspace bounds both)
#define CVO_ASSOC_CASE(F)                                                                              \
  case F:                                                                                              \
    if (idx16)                                                                                         \
      launch_assoc_t<unsigned short, ASSOC_CAP16, F>(s, instr, grid, descs, dp, st, A, packed);        \
    else                                                                                               \
      launch_assoc_t<int, ASSOC_CAP32, F>(s, instr, grid, descs, dp, st, A, packed);                   \
    break;
  switch (feat) {
    CVO_ASSOC_CASE(FEAT_GEO)
    CVO_ASSOC_CASE(FEAT_COL)
    CVO_ASSOC_CASE(FEAT_HOT)
    default:
      CVO_ASSOC_CASE(FEAT_ALL)
  }
#undef CVO_ASSOC_CASE
}

void launch_coeff(hipStream_t s, bool instr, int nblk, int split, int n_pairs, const PairDesc* descs, const DevParams* dp,
                  PairState* st, const ArenaArg& A, int flags) {
  const int packed = nblk | (split << 14) | (int)((unsigned)n_pairs << 20);  // 14 + 6 + 12 bits
  if (instr)
    hipLaunchKernelGGL(k_coeff<true>, row_grid(nblk * split + 1, n_pairs), dim3(ASSOC_THREADS), 0, s, descs, dp, st, A.base, flags,
                       packed, A.stride256, A.Npad);
  else  // (+ 1 block per pair: the speculative run of the update, update_speculate)
    hipLaunchKernelGGL(k_coeff<false>, row_grid(nblk * split + 1, n_pairs), dim3(ASSOC_THREADS), 0, s, descs, dp, st, A.base, flags,
                       packed, A.stride256, A.Npad);
}

// CVO_VERIFY_LISTS: literal re-derivation of every row after the association of an iteration (k_verify)
void launch_verify(hipStream_t s, int feat, int N, int n_pairs, const PairDesc* descs, const DevParams* dp, const int* st,
                   int lean) {
  const dim3 grid((unsigned)std::min((N + 3) / 4, 2048), (unsigned)n_pairs);
  // (the self-check always takes the general form of the semantic kernel: one-hot rows through the row arithmetic)
  if (feat != FEAT_GEO)
    hipLaunchKernelGGL(k_verify<FEAT_ALL>, grid, dim3(256), 0, s, descs, dp, st, lean);
  else
    hipLaunchKernelGGL(k_verify<FEAT_GEO>, grid, dim3(256), 0, s, descs, dp, st, lean);
}

void launch_dense(hipStream_t s, int feat, int N, int n_pairs, int dense_blocks, const PairDesc* descs, const DevParams* dp,
                  const PairState* st) {
  const dim3 grid(dense_blocks, n_pairs);
  // a small pair solved alone has a block per overflow row (dense_blocks_for): the instantiation with the wide-row phase
  const bool wide = n_pairs <= 1 && N <= DENSE_BLOCKS_MAX / 2;
#define CVO_LAUNCH_DENSE(F)                                                                                  \
  do {                                                                                                       \
    if (wide)                                                                                                \
      hipLaunchKernelGGL((k_assoc_dense<F, 4, true>), grid, dim3(256), 0, s, descs, dp, st);                 \
    else                                                                                                     \
      hipLaunchKernelGGL((k_assoc_dense<F, 4, false>), grid, dim3(256), 0, s, descs, dp, st);                \
  } while (0)
  switch (feat) {  // (4 waves per block: dense_waves_for)
    case FEAT_GEO: CVO_LAUNCH_DENSE(FEAT_GEO); break;
    case FEAT_COL: CVO_LAUNCH_DENSE(FEAT_COL); break;
    case FEAT_HOT: CVO_LAUNCH_DENSE(FEAT_HOT); break;
    default: CVO_LAUNCH_DENSE(FEAT_ALL); break;
  }
#undef CVO_LAUNCH_DENSE
}

// which instantiation of the association kernels a call needs (FEAT_*, cvo_pair_math.h)
inline int call_feat(const DevParams& dp, bool all_one_hot) {
  if (dp.mode == 2) return FEAT_ALL;
  if (!(dp.use_col || dp.use_sem || dp.use_geotype)) return FEAT_GEO;
  if (!dp.use_sem) return FEAT_COL;
  return all_one_hot ? FEAT_HOT : FEAT_ALL;
}

struct LaunchGeom {
  int n_pairs, p0, T, gx, gy, nba, nbc, npb, N, csplit;
  int dense_blocks = DENSE_BLOCKS_MIN;  // k_assoc_dense grid x = PairDesc::dense_blocks of every pair of the launch
  int group = 0;        // sub-batch index (its stream)
  int horizon_cap = 1 << 20;  // the lean graph's period (DevParams::lean_U)
  bool idx16, instr, verify;
  int feat = FEAT_GEO;  // which instantiation of the association kernels the call needs (call_feat)
  hipStream_t stream;
  ArenaArg arena;  // of pair p0
};

void launch_init(cvo_ctx* c, const LaunchGeom& g) {
  hipLaunchKernelGGL(k_update<true>, dim3(g.n_pairs), dim3(64), 0, g.stream, c->d_descs + g.p0, c->d_params,
                     c->d_status + 2 * g.p0, 0);
}

// The rebuild kernels: no-ops (early exit) unless k_update flagged the pair's candidate list as expired.
void launch_rebuild(cvo_ctx* c, const LaunchGeom& g) {
  const PairDesc* descs = c->d_descs + g.p0;
  const PairState* states = c->d_states + g.p0;
  hipLaunchKernelGGL(k_prep, dim3(g.npb, g.n_pairs), dim3(PREP_THREADS), 0, g.stream, descs, c->d_params, states);
  launch_scan(g.stream, g.T, dim3(g.gx, g.gy, g.n_pairs), descs, c->d_params, states, 0);
  launch_list(g.stream, g.idx16, g.N, g.n_pairs, descs, c->d_params, states);
}

// One optimiser iteration over the current lists: association, [overflow rows], coefficients + update (the last
// block of k_coeff).  Lean: no k_assoc_dense, pairs with overflow rows or an expired list wait.  `flags` see
// update_body.
// `dense`: a lean graph that runs k_assoc_dense all the same (pairs with overflow rows / in the dense regime that need
// no rebuild opportunity in every iteration).
void launch_core(cvo_ctx* c, const LaunchGeom& g, bool lean, int flags, bool dense = false) {
  const PairDesc* descs = c->d_descs + g.p0;
  const int* st = c->d_status + 2 * g.p0;  // the sub-batch's status words (see setup_batch)
  const bool lean_dense = lean && dense;
  // rows beyond their cached lists first (a wave per row; per-row results), then every row's reduction in k_assoc
  if (!lean || dense) launch_dense(g.stream, g.feat, g.N, g.n_pairs, g.dense_blocks, descs, c->d_params, c->d_states + g.p0);
  launch_assoc(g.stream, g.idx16, g.feat, g.instr, g.nba, g.n_pairs, descs, c->d_params, c->d_states + g.p0, g.arena,
               (lean ? 1 : 0) | (lean_dense ? 4 : 0));
  if (g.verify) launch_verify(g.stream, g.feat, g.N, g.n_pairs, descs, c->d_params, st, (lean ? 1 : 0) | (lean_dense ? 4 : 0));
  // ... their coefficient sums likewise (k_coeff_dense leaves per-row sums, k_coeff picks them up)
  if (!lean || dense)
    // (7 waves per SIMD against k_assoc_dense's 4: twice the blocks, so that a lone pair's rows get a wave each - the kernel
    // then lasts as long as its longest row, not as two)
    hipLaunchKernelGGL((k_coeff_dense<4>), dim3(g.n_pairs <= 4 ? std::min(2 * g.dense_blocks, (int)DENSE_BLOCKS_MAX) : g.dense_blocks, g.n_pairs), dim3(256), 0, g.stream, descs,
                       c->d_params, c->d_states + g.p0, g.n_pairs >= CVO_COEFF_DENSE_MULTI_FROM ? 8 : 1);
  launch_coeff(g.stream, g.instr, g.nba, g.csplit, g.n_pairs, descs, c->d_params, c->d_states + g.p0, g.arena,
               flags | (lean ? 1 : 0) | (lean_dense ? 32 : 0) | (g.idx16 ? 0 : 64));
}

// A chunk of U iterations.  Full: every iteration can rebuild its candidate list and serve overflow rows.
// Lean: rebuild opportunities only every lean_U iterations; pairs that need more wait for a full chunk.
// lean_U == 0: the full chunk WITHOUT k_assoc_dense - a rebuild opportunity in every iteration with the full graph's
// rebuild rule (no horizon), but a pair whose rows overflow their lists waits (and asks for the dense kernel: want = 4).
// Large clouds run their fast first iterations here: the dense kernel, launched for nothing, is 5 us + a launch gap.
void launch_chunk(cvo_ctx* c, const LaunchGeom& g, int U, bool lean, int lean_U, bool dense = false) {
  if (lean && lean_U == 0) {
    for (int u = 0; u < U; u++) {
      launch_rebuild(c, g);
      launch_core(c, g, true, 2);
    }
    return;
  }
  if (!lean) {
    for (int u = 0; u < U; u++) {
      launch_rebuild(c, g);
      launch_core(c, g, false, 2);
    }
    return;
  }
  for (int u = 0; u < U; u++) {
    if (u % lean_U == 0) launch_rebuild(c, g);
    const bool last = (u % lean_U == lean_U - 1) || u == U - 1;
    // (horizon of the rebuild rule: the lean graph's period even in a calm chunk, whose one opportunity per chunk is a bet
    // on the list outliving the linear prediction - a pair that loses it waits for the next chunk)
    launch_core(c, g, true, (last ? 2 : 0) | (std::min(lean_U, g.horizon_cap) << 8), dense);
  }
}

}  // namespace
