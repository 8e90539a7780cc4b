// cvo_sched.hip -- the optimiser loop on the host: parameters -> DevParams, descriptors and initial states of a batch (setup_batch), the eight chunk graphs of a sub-batch and which one runs next, cvo_align_batch / cvo_align_ex / cvo_align.
// A SECTION of the one translation unit cvo_hip.hip (which includes the sections in dependency order and says why it is one
// unit); not compiled on its own.  Shared declarations: cvo_internal.h.
namespace {

DevParams make_dev_params(const cvo_ctx* ctx, const cvo_params_t& p) {
  DevParams d{};
  d.sp_thres = p.sp_thres;
  d.sigma2 = p.sigma * p.sigma;
  d.c2 = p.c_ell * p.c_ell;
  d.c_sigma2 = p.c_sigma * p.c_sigma;
  d.s_ell = p.s_ell;
  d.s_sigma = p.s_sigma;
  d.c = p.c;
  d.d = p.d;
  const float s_sigma2 = p.s_sigma * p.s_sigma;
  // log() on float arguments (CvoGPU.cu:509-515); evaluated with the host libm, once per call
  d.log_geo = std::log(p.sp_thres / d.sigma2);
  d.d2_c_thres = 1.f;
  d.d2_s_thres = 1.f;
  if (p.is_using_intensity) d.d2_c_thres = (float)(-2.0 * d.c2 * (double)std::log(p.sp_thres / d.c_sigma2));
  if (p.is_using_semantics)
    d.d2_s_thres = (float)(-2.0 * d.s_ell * d.s_ell * (double)std::log(p.sp_thres / s_sigma2));
  d.ell_min = p.ell_min;
  d.ell_decay_rate = p.ell_decay_rate;
  d.ell_decay_start = p.ell_decay_start;
  d.max_iter = p.MAX_ITER;
  d.eps = p.eps;
  d.eps_2 = p.eps_2;
  d.min_step = p.min_step;
  d.max_step = p.max_step;
  d.K_max = p.nearest_neighbors_max;
  d.window = p.indicator_window_size;
  d.stable_thr = p.indicator_stable_threshold;
  d.use_geo = p.is_using_geometry != 0;
  d.use_col = p.is_using_intensity != 0;
  d.use_sem = p.is_using_semantics != 0;
  d.use_range_ell = p.is_using_range_ell != 0;
  d.use_geotype = p.is_using_geometric_type != 0;
  {
    auto mid = [](float v) { const float a = std::fabs(v); return std::isfinite(v) && a >= 0x1p-20f && a <= 0x1p20f; };
    d.fast_div_cd = mid(p.c) && mid(p.d) ? 1 : 0;
  }
  // List-reuse knobs, re-tuned in round 4 (scripts/skin_sweep.py, profiles/r4/skin_sweep.txt): the linear "outlives the
  // next h iterations at the current speed" predictions are pessimistic once the pose jitters around its optimum (the
  // allowance used since a build stays at a few percent while every iteration moves ~10 % of it), so thinner skins and
  // a smaller margin win on every configuration: headline batch 62.05 -> 60.6 ms, config 3 single pair 21.5 -> 18.8 us
  // per iteration.  (Round-2 values: 2.0 / 1.3 / 1.25.)
  d.skin_frac = 1.0f;
  d.lean_skin = 0.5f;
  d.dense_regime = ctx_opt(ctx, "NO_DENSE_REGIME") ? 0 : 1;
  d.skin_blend = 0.25f;
  d.skin_min = 0.05f;
  d.skin_max = 0.35f;  // (0.25 until round 6; re-swept on the un-aligned shrink rebuilds: 0.25 / 0.3 / 0.35 / 0.4 / 0.5 -> 55.55 / 55.22 / 55.06 / 55.19 /
                       // 55.25 ms for the 64-pair step, 16 pairs -1.3 %, config 3 batch +0.3 %, single pairs unchanged: profiles/r6/shrink_align.txt)
  if (const char* e = ctx_opt(ctx, "SKIN_MAX")) d.skin_max = std::max(d.skin_min, (float)atof(e));
  if (const char* e = ctx_opt(ctx, "LEAN_SKIN")) d.lean_skin = std::max(0.1f, (float)atof(e));
  d.horizon_margin = 0.3f;
  if (const char* e = ctx_opt(ctx, "HORIZON_MARGIN")) d.horizon_margin = std::max(0.f, (float)atof(e));
#ifndef CVO_REBUILD_SHRINK
#define CVO_REBUILD_SHRINK 0.9f
#endif
  d.rebuild_shrink = CVO_REBUILD_SHRINK;
  if (const char* e = ctx_opt(ctx, "SKIN")) d.skin_frac = std::max(0.f, (float)atof(e));
  d.phase_ticks = ctx_opt(ctx, "PHASE_TICKS") ? 1 : 0;
  d.kernel_clock = ctx_opt_on(ctx, "KERNEL_CLOCK") ? 1 : 0;
  d.verify_lists = ctx_opt_on(ctx, "VERIFY_LISTS") ? 1 : 0;
  d.debug_no_motion_bound = ctx_opt(ctx, "DEBUG_NO_MOTION_BOUND") ? 1 : 0;
  d.debug_drop_partial = ctx_opt(ctx, "DEBUG_DROP_PARTIAL") ? atoi(ctx_opt(ctx, "DEBUG_DROP_PARTIAL")) : 0;
  return d;
}

// Scan geometry: T chunks of 64 sorted targets per wave (smaller slices cull better, larger ones
// amortise the row operands) and the number of row groups per block, chosen so that a launch has a

struct BatchSetup {
  int N, M, T, gpb, gx, gy, G;
  bool long_lists = false;
  Dims d;
  PairLayout L;
  LaunchGeom geom;
};

// Sizes of a batch queue (cvo_batch_open): the slots of the workspace are laid out for clouds of up to n_max / m_max points
// and the launches for source clouds of at least n_min (the coefficient split of a pair follows from its own size).
struct QueueDims {
  int n_max, m_max, n_min;
};

// Descriptor + initial state of the pair that occupies slot p of the workspace (host copies; the caller uploads them).
void fill_pair(cvo_ctx* ctx, const BatchSetup* S, const cvo_params_t* params, const cvo_align_opts_t* opts, int mode, float mode_ell,
               int n_slots, int p, const cvo_cloud* X, const cvo_cloud* Y, const float* Tm, unsigned long long serial, int max_iter) {
  const int trace_cap = (opts && opts->trace) ? opts->trace_capacity : 0;
  const int Kmax = params->nearest_neighbors_max;
  {
    char* base = ctx->arena + S->L.total * (size_t)p;
    PairDesc& D = ctx->h_descs[p];
    std::memset(&D, 0, sizeof(D));
    D.N = X->n;
    D.M = Y->n;
    D.call_serial = serial;
    D.max_iter = max_iter;
    // paddings are derived from the batch maxima so every pair shares one launch geometry
    D.Mpad = S->d.Mpad;
    D.nchunks = S->d.nchunks;
    D.nslices = S->d.Mpad / (64 * S->T);
    D.rbw = (int)align_up((size_t)(S->d.Mpad / (64 * S->T) + 31) / 32, 4);
    D.nblk_assoc = S->d.nblk_assoc;
    // coefficient phase: small clouds get several blocks per row block (see coeff_rows); a function of the pair's own
    // size only, so that a pair is reduced in the same order whether it is solved alone or inside a batch
    D.csplit = coeff_split(X->n);
    D.nblk_coeff = S->d.nblk_assoc * D.csplit;
    D.NG = (X->n + ROWS_PER_GROUP - 1) / ROWS_PER_GROUP;
    D.NGpad = S->d.NGpad;
    D.ymax = Y->rmax;
    D.sqrt_nm = std::sqrt((double)X->n * (double)Y->n);
    D.cx = X->cx;
    D.cy = X->cy;
    D.cz = X->cz;
    D.x4 = X->x4;
    D.xs4 = X->xs4;
    D.xfeat = X->feat;
    D.xlabel = X->label;
    D.xgeo = X->geo;
    D.xorder = X->order;
    D.y4 = Y->x4;
    D.ys4 = Y->xs4;
    D.yfeat = Y->feat;
    D.ylabel = Y->label;
    D.ygeo = Y->geo;
    D.xlid = X->lid;
    D.ylid = Y->lid;
    D.yorder = Y->order;
    D.yinv = Y->inv;
    D.ycull = (float4*)(base + S->L.ycull);
    D.xcull = (float4*)(base + S->L.xcull);
    D.gbox = (float4*)(base + S->L.gbox);
    D.cellbox = (float4*)(base + S->L.cellbox);
    D.sbox = (float4*)(base + S->L.sbox);
    D.masks = (unsigned long long*)(base + S->L.masks);
    D.rowbits = (unsigned*)(base + S->L.rowbits);
    D.row_cnt = (int*)(base + S->L.row_cnt);
    D.tile_count = (unsigned long long*)(base + S->L.tile_count);
    D.ovf_rows = (int*)(base + S->L.ovf_rows);
    D.ovf_bits = (unsigned long long*)(base + S->L.ovf_bits);
    D.cand_cnt = (int*)(base + S->L.cand_cnt);
    D.rowperm = (int*)(base + S->L.rowperm);
    D.xp4 = (float4*)(base + S->L.xp4);
    D.ip = (int*)(base + S->L.ip);
    D.iorig = (int*)(base + S->L.iorig);
    D.long_j = S->long_lists ? (unsigned short*)(base + S->L.long_j) : nullptr;
    D.long_stamp = (unsigned long long*)(base + S->L.long_stamp);
    D.cand_j = (void*)(base + S->L.cand_j);
    D.ell = (EllEntry*)(base + S->L.ell);
    D.ell_j = (int*)(base + S->L.ell_j);
    D.nnz_row = (unsigned*)(base + S->L.nnz_row);
    D.rowres = (RowRes*)(base + S->L.rowres);
    D.rowcoef = (double*)(base + S->L.rowcoef);
    D.flow_part = (unsigned long long*)(base + S->L.flow_part);
    D.cnt_part = (unsigned long long*)(base + S->L.cnt_part);
    D.coef_part = (unsigned long long*)(base + S->L.coef_part);
    D.shadow = (unsigned long long*)(base + S->L.shadow);
    D.st = ctx->d_states + p;
    D.trace = trace_cap > 0 ? (cvo_trace_t*)(base + S->L.trace) : nullptr;
    {
      // status / requested-graph mirrors the host polls: every sub-batch owns ONE contiguous block [status[n_g] | want[n_g]]
      // at 2 * p0(g), fetched with one copy per chunk (two copies per chunk and stream were two 5 us blits)
      int g = 0;
      while (g + 1 < S->G && (int)((long)n_slots * (g + 1) / S->G) <= p) g++;
      const int p0 = (int)((long)n_slots * g / S->G), p1 = (int)((long)n_slots * (g + 1) / S->G);
      D.status_out = ctx->d_status + 2 * p0 + (p - p0);
      D.want_out = ctx->d_status + 2 * p0 + (p1 - p0) + (p - p0);
      D.status_host = ctx->h_status[0] + 2 * p0 + (p - p0);
      D.want_host = ctx->h_status[0] + 2 * p0 + (p1 - p0) + (p - p0);
    }
    D.asum_host = reinterpret_cast<double*>(ctx->h_status[1]) + p;  // (2 ints per pair = one double)
    D.gate = (int*)(base + S->L.gate);
    D.gate_flow = (int*)(base + S->L.gate_flow);
    D.dense_off = (int*)(base + S->L.dense_off);
    D.dense_rel = (int*)(base + S->L.dense_rel);
    D.ovf_wsum = (int*)(base + S->L.ovf_wsum);
    D.word_base = (int*)(base + S->L.word_base);
    D.done = (int*)(base + S->L.done);

    PairState& st = ctx->h_states[p];
    std::memset(&st, 0, sizeof(st));
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) st.R[3 * i + j] = Tm[4 * j + i];  // CvoGPU.cu:1363-1364
      st.T[i] = Tm[12 + i];
    }
    st.ell = mode == 0 ? params->ell_init : mode_ell;  // CvoState.cu:30
    st.K = Kmax;                                        // CvoGPU.cu:1385
    if (mode == 0 && opts && opts->override_state) {  // (validated by the caller)
      st.ell = opts->ell0;
      st.K = opts->K0;
    }
    st.K_last = 0;  // set by the update of every EXECUTED iteration: > 0 <=> at least one association pass ran
    // (the pair's counters - gate, gate_flow, done, tile_count - are zeroed by k_update<INIT>; the slice bits of a row
    // are cleared by k_prep before every build, the first one included: five memsets per pair used to cost 10 us each call)
  }
}

unsigned long long next_call_serial() {
  static std::atomic<unsigned long long> g_call_serial{1};  // never repeats inside a process: see PairDesc::long_stamp
  return g_call_serial.fetch_add(1);
}

// Builds descriptors + initial states for a batch and uploads them.  qd != nullptr: plans the workspace of a batch queue
// (cvo_batch_open) for n_pairs SLOTS without occupants - every slot starts out finished, cvo_batch_submit fills them.
// What every entry point checks before it touches the device: the arguments of the call, the parameter and coordinate
// ranges the kernels' arithmetic is stated for, the attribute arrays the call's kernels will read.  N / M: the largest
// source / target cloud of the call.
int check_call(cvo_ctx* ctx, const cvo_params_t* params, int n_pairs, const cvo_cloud* const* sources, const cvo_cloud* const* targets,
               const cvo_align_opts_t* opts, int mode, float mode_ell, const QueueDims* qd, int* N_out, int* M_out) {
  if (!ctx) return CVO_E_INVALID;
  if (ctx->queue_open && !qd) return fail(ctx, CVO_E_INVALID, "a batch queue is open on this context (cvo_batch_close it first)");
  if (!params || n_pairs <= 0 || (!qd && (!sources || !targets))) return fail(ctx, CVO_E_INVALID, "null argument");
  if (params->is_using_kdtree)
    return fail(ctx, CVO_E_UNSUPPORTED, "is_using_kdtree=1 is out of scope (SURVEY.md section 2, row 11)");
  if (params->nearest_neighbors_max <= 0) return fail(ctx, CVO_E_INVALID, "nearest_neighbors_max must be > 0");
  if (params->indicator_window_size + 1 >= IND_CAP || params->indicator_window_size < 0)
    return fail(ctx, CVO_E_INVALID, "indicator_window_size out of range");
  {
    // The row loops evaluate their IEEE double divisions in a hoisted form (rcp_refined / div_by, cvo_device.h) that equals
    // the plain division wherever v_div_scale / v_div_fixup would pass the operands through: denominators 2 l^2, 2 c_ell^2,
    // 2 s_ell^2 far from zero, denormals and infinity.  Lengthscales outside [1e-30, 1e15] (and non-finite ones) are refused
    // here instead of silently leaving that domain; coordinates are bounded the same way below.
    auto ok_scale = [](float v) { return std::isfinite(v) && v >= 1e-30f && v <= 1e15f; };
    const float ell0 = mode == 0 ? ((opts && opts->override_state) ? opts->ell0 : params->ell_init) : mode_ell;
    if (!ok_scale(ell0) || (mode == 0 && !ok_scale(params->ell_min)))
      return fail(ctx, CVO_E_INVALID, "lengthscale outside [1e-30, 1e15] (ell_init / ell_min / the ell of the call)");
    if (params->is_using_intensity && !ok_scale(params->c_ell)) return fail(ctx, CVO_E_INVALID, "c_ell outside [1e-30, 1e15]");
    if (params->is_using_semantics && !ok_scale(params->s_ell)) return fail(ctx, CVO_E_INVALID, "s_ell outside [1e-30, 1e15]");
  }
  if (mode == 0 && opts && opts->override_state) {
    // the ELL holds nearest_neighbors_max slots per row and the kernels write slot nnz while nnz < K
    if (opts->K0 < 1 || opts->K0 > params->nearest_neighbors_max)
      return fail(ctx, CVO_E_INVALID, "cvo_align_opts_t.K0 must lie in [1, nearest_neighbors_max]");
    if (!(opts->ell0 > 0.f) || !std::isfinite(opts->ell0))
      return fail(ctx, CVO_E_INVALID, "cvo_align_opts_t.ell0 must be finite and > 0");
  }
  int N = qd ? qd->n_max : 0, M = qd ? qd->m_max : 0;
  if (qd && (qd->n_max <= 0 || qd->m_max <= 0 || qd->n_min <= 0 || qd->n_min > qd->n_max))
    return fail(ctx, CVO_E_INVALID, "cvo_batch_open: bad cloud sizes");
  for (int p = 0; p < n_pairs && !qd; p++) {
    if (!sources[p] || !targets[p]) return fail(ctx, CVO_E_INVALID, "null cloud");
    if (sources[p]->ctx != ctx || targets[p]->ctx != ctx)
      return fail(ctx, CVO_E_INVALID, "cloud belongs to another context");
    if (sources[p]->n <= 0 || targets[p]->n <= 0) return fail(ctx, CVO_E_INVALID, "empty cloud in batch");
    if (!(sources[p]->rmax <= 1e15f) || !(targets[p]->rmax <= 1e15f))  // (NaN sticks in rmax, see upload_host_cloud)
      return fail(ctx, CVO_E_INVALID, "cloud with non-finite or astronomically large coordinates (|p| > 1e15)");
    N = std::max(N, sources[p]->n);
    M = std::max(M, targets[p]->n);
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  (void)hipGetLastError();  // a stale error of an unrelated earlier call must not be attributed to this one
  {  // attribute arrays the kernels of this call read but a cloud was uploaded without: zeros, as upstream has them
    const bool nf = params->is_using_intensity != 0, nl = params->is_using_semantics != 0,
               ng = params->is_using_geometric_type != 0 && mode != 2;
    if ((nf || nl || ng) && !qd)
      for (int p = 0; p < n_pairs; p++) {
        int rc0 = ensure_attributes(ctx, sources[p], nf, nl, ng);
        if (rc0 == CVO_OK) rc0 = ensure_attributes(ctx, targets[p], nf, nl, ng);
        if (rc0 != CVO_OK) return rc0;
      }
  }
  *N_out = N;
  *M_out = M;
  return CVO_OK;
}

int setup_batch(cvo_ctx* ctx, const cvo_params_t* params, int n_pairs, const cvo_cloud* const* sources,
                const cvo_cloud* const* targets, const float* init_T, const cvo_align_opts_t* opts, int mode,
                float mode_ell, BatchSetup* S, DevParams* dp_out, const float* kernel_inv_and_cull = nullptr,
                const QueueDims* qd = nullptr) {
  int N = 0, M = 0;
  {
    const int rc0 = check_call(ctx, params, n_pairs, sources, targets, opts, mode, mode_ell, qd, &N, &M);
    if (rc0 != CVO_OK) return rc0;
  }
  const int trace_cap = (opts && opts->trace) ? opts->trace_capacity : 0;
  const int Kmax = params->nearest_neighbors_max;
  S->N = N;
  S->M = M;
  // k_list packs a row's candidate count next to an 8-bit row number; the candidate bitmap of a pair takes
  // N * M / 8 bytes (DESIGN.md "Data layout"), every pair of a batch sized by the batch maxima
  if (M >= (1 << 23)) return fail(ctx, CVO_E_INVALID, "target clouds are limited to 8388607 points");
  // (the update reduces an iteration's nonzero count in 32 bits: rows x nearest_neighbors_max must fit)
  if ((unsigned long long)N * (unsigned long long)std::max(params->nearest_neighbors_max, 1) >= (1ull << 32))
    return fail(ctx, CVO_E_INVALID, "source rows x nearest_neighbors_max must stay below 2^32");
  // overflow rows keep sorted candidate lists of their own (PairDesc::long_j) when sorted positions fit 16 bits
  S->long_lists = M <= 65535 && ctx_opt(ctx, "NO_LONG_LISTS") == nullptr;
  S->L = make_layout(N, M, Kmax, trace_cap, S->long_lists, &S->d);
  {
    size_t free_b = 0, total_b = 0;
    size_t need = S->L.total * (size_t)n_pairs;
    const bool tight = need > ctx->arena_bytes && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b + ctx->arena_bytes;
    if (tight && S->long_lists) {
      // the long lists of overflow rows (N x 2 KB per pair) are a speed feature: without them such rows are scanned
      // literally.  Give them up before giving up the call.
      S->long_lists = false;
      S->L = make_layout(N, M, Kmax, trace_cap, false, &S->d);
      need = S->L.total * (size_t)n_pairs;
      if (ctx_opt(ctx, "VERBOSE")) fprintf(stderr, "[cvo] workspace: long lists dropped to fit device memory\n");
    }
    if (need > ctx->arena_bytes && hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b + ctx->arena_bytes) {
      char msg[320];
      snprintf(msg, sizeof msg,
               "workspace of %d pair(s) of %d x %d points needs %.1f GiB (candidate bitmap N*M/8 = %.1f GiB per pair, ELL "
               "%.1f GiB per pair; the long lists of overflow rows have already been dropped) but %.1f GiB of device memory are "
               "free: split the batch or the clouds",
               n_pairs, N, M, need / 1073741824.0, (double)N * S->d.Mpad / 8.0 / 1073741824.0,
               (double)S->d.Npad * Kmax * 20.0 / 1073741824.0, (free_b + ctx->arena_bytes) / 1073741824.0);
      return fail(ctx, CVO_E_NOMEM, msg);
    }
  }
  int rc = ensure_workspace(ctx, n_pairs, S->L.total);
  if (rc != CVO_OK) return rc;
  // sub-batches on separate streams (see cvo_ctx): the scan geometry is chosen for one group's launch
  S->G = n_pairs >= 32 ? 4 : (n_pairs >= 8 ? 2 : 1);
  if (const char* e = ctx_opt(ctx, "STREAMS")) S->G = std::max(1, std::min(atoi(e), (int)cvo_ctx::MAX_GROUPS));
  S->G = std::min(S->G, n_pairs);
  if ((n_pairs + S->G - 1) / S->G > 4095 || S->d.nblk_assoc > 16383)
    return fail(ctx, CVO_E_INVALID, "batch too large for one call: at most 4095 pairs per stream and 2097024 source points per cloud");
  choose_scan_config(ctx, (n_pairs + S->G - 1) / S->G, S->d.NG, S->d.Mpad, &S->T, &S->gpb);

  DevParams dp = make_dev_params(ctx, *params);
  dp.mode = mode;
  if (mode == 2) {  // non-isotropic kernel: 9 floats of the inverse (row-major) + the squared cull radius
    for (int q = 0; q < 9; q++) dp.kinv[q] = kernel_inv_and_cull[q];
    dp.d2_cull = kernel_inv_and_cull[9];
    dp.s_ell_sq = params->s_ell * params->s_ell;
    dp.use_geotype = 0;  // CvoGPU.cu:1950-1951
    // that kernel's prologue keeps s_ell^2 in float (CvoGPU.cu:236, 252)
    if (params->is_using_semantics)
      dp.d2_s_thres = (float)(-2.0 * dp.s_ell_sq * (double)std::log(params->sp_thres / (params->s_sigma * params->s_sigma)));
  }
  dp.T = S->T;
  dp.groups_per_block = S->gpb;
  dp.long_lists = S->long_lists ? 1 : 0;
  dp.row_max_cap = ASSOC_CAP16;
  // (clustered 10k scenes, scripts/scene_batch.py: 8 for a lone pair; 24 against 64 wins 9 % at 8 pairs in flight, nothing at
  // 16, and LOSES 5 % at 32 and 10 % at 64 - a chip full of pairs wants its rows in the thread-per-row kernel, whose lanes
  // are all rows, not in steps of 128 candidate slots per row)
  dp.row_max_busy = n_pairs <= 4 ? 8 : (n_pairs <= 16 ? 24 : (int)ASSOC_CAP16);
  if (const char* e = ctx_opt(ctx, "ROW_MAX")) dp.row_max_cap = std::max(1, std::min(atoi(e), (int)ASSOC_CAP16));
  dp.lean_U = 8;
  if (const char* e = ctx_opt(ctx, "LEAN_U")) dp.lean_U = std::max(1, atoi(e));
  // Calm pairs (see PairState::want_full).  In the end game the pose jitters around its optimum: the motion PER ITERATION
  // stays at ~10 % of a list's allowance while the allowance used SINCE THE BUILD stays below 5 % for hundreds of
  // iterations (CVO_VERBOSE=2 prints both), so a linear "outlives the next 64 iterations" test never fires.  Four
  // iterations of linear margin it is: 62.3 -> 61.4 ms per headline step, single pairs -1.5 ... -2.5 %, no additional waits.
  dp.calm_U = 4;
  dp.lean_U2 = 2;
  if (dp.lean_U2 >= dp.lean_U) dp.lean_U2 = 0;
  // (Round 4 made the optional shrink rebuilds of a batch wait for iteration counts that are multiples of 64 so that the pairs
  // of a sub-batch share a pass of the rebuild kernels: -1.7 % then.  Re-measured in round 6, with cheaper rebuild kernels and
  // a shorter serial tail: the stale lists' extra candidates cost more than the shared passes save - 64 x 10k geometric 56.55 ->
  // 55.67 ms, 64 x config 3 163.5 -> 158.1, 64 clustered scenes 793 -> 779, 16 / 32 pairs 0 / -1.6 % (profiles/r6/shrink_align.txt).
  // The mask stays as a switch.)
  dp.shrink_align = 0;
  if (const char* e = ctx_opt(ctx, "SHRINK_ALIGN")) dp.shrink_align = std::max(0, atoi(e));
  if (opts && opts->max_iterations > 0) dp.max_iter = std::min(dp.max_iter, opts->max_iterations);
  if (opts && opts->kernel_clock) dp.kernel_clock = 1;
  dp.trace_capacity = trace_cap;
  // the columns of the ELL entries (ell_j) are only written when somebody can ask for them afterwards
  dp.keep_columns = (mode != 0 || trace_cap > 0 || dp.verify_lists || params->is_exporting_association ||
                     ctx_opt(ctx, "KEEP_COLUMNS")) ? 1 : 0;
  dp.trace_dense = opts ? opts->trace_dense : 0;
  dp.trace_every = opts ? opts->trace_every : 0;
  *dp_out = dp;

  ctx->h_descs.resize(n_pairs);
  ctx->h_states.resize(n_pairs);
  if (!qd) {
    const unsigned long long serial = next_call_serial();
    for (int p = 0; p < n_pairs; p++)
      fill_pair(ctx, S, params, opts, mode, mode_ell, n_pairs, p, sources[p], targets[p], init_T + 16 * (size_t)p, serial, dp.max_iter);
  } else {  // empty slots: finished pairs, which every kernel skips
    for (int p = 0; p < n_pairs; p++) {
      std::memset(&ctx->h_descs[p], 0, sizeof(PairDesc));
      std::memset(&ctx->h_states[p], 0, sizeof(PairState));
      ctx->h_states[p].status = 1;
    }
  }
  // the blocks of k_assoc beyond a smaller pair's N still write their (zero) partials, but the
  // partial arrays of pairs whose N is smaller than the batch maximum are fully covered by nblk.
  // one copy from the pinned staging block (no call is in flight on this context: every call ends synchronised)
  std::memcpy(ctx->h_ctl, &dp, sizeof(DevParams));
  std::memset(ctx->h_ctl + ctx->ctl_off_status, qd ? 1 : 0, sizeof(int) * 2 * (size_t)ctx->cap_pairs);  // (queue: any non-zero word = finished)
  std::memcpy(ctx->h_ctl + ctx->ctl_off_descs, ctx->h_descs.data(), sizeof(PairDesc) * (size_t)n_pairs);
  std::memcpy(ctx->h_ctl + ctx->ctl_off_states, ctx->h_states.data(), sizeof(PairState) * (size_t)n_pairs);
  std::memset(ctx->h_status[0], qd ? 1 : 0, sizeof(int) * 2 * (size_t)ctx->cap_pairs);
  {
    // (descriptors and states of at most n_pairs <= cap_pairs slots are used; the block is laid out for cap_pairs)
    const size_t upto = ctx->ctl_off_states + sizeof(PairState) * (size_t)n_pairs;
    if ((size_t)n_pairs * 2 >= (size_t)ctx->cap_pairs) {
      HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ctl, ctx->h_ctl, upto, hipMemcpyHostToDevice, ctx->stream));
    } else {  // a small call on a context sized for a large batch: skip the unused descriptors in between
      HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ctl, ctx->h_ctl, ctx->ctl_off_descs + sizeof(PairDesc) * (size_t)n_pairs, hipMemcpyHostToDevice, ctx->stream));
      HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ctl + ctx->ctl_off_states, ctx->h_ctl + ctx->ctl_off_states, sizeof(PairState) * (size_t)n_pairs,
                                  hipMemcpyHostToDevice, ctx->stream));
    }
  }
  S->gx = (S->d.Mpad / (64 * S->T) + 3) / 4;
  S->gy = ((int)align_up((size_t)S->d.NG, 64) + S->gpb - 1) / S->gpb;
  S->geom.n_pairs = n_pairs;
  S->geom.p0 = 0;
  S->geom.stream = ctx->stream;
  S->geom.T = S->T;
  S->geom.gx = S->gx;
  S->geom.gy = S->gy;
  S->geom.nba = S->d.nblk_assoc;
  S->geom.N = N;
  S->geom.dense_blocks = dense_blocks_for(N, n_pairs);
  S->geom.arena.base = ctx->arena;
  S->geom.arena.stride256 = (unsigned)(S->L.total >> 8);
  S->geom.arena.Npad = S->d.Npad;
  S->geom.csplit = qd ? coeff_split(qd->n_min) : 1;
  for (int p = 0; p < n_pairs && !qd; p++) S->geom.csplit = std::max(S->geom.csplit, coeff_split(sources[p]->n));
  S->geom.nbc = S->d.nblk_coeff;
  S->geom.npb = S->d.Mpad / PREP_THREADS + (S->d.NGpad * ROWS_PER_GROUP + PREP_THREADS - 1) / PREP_THREADS;
  S->geom.idx16 = M < 65536;
  // (the non-isotropic kernel of mode 2 lives in the GENERAL instantiations only: single evaluations, never the loop)
  {
    // every cloud of the call with exact one-hot class rows (ids made at upload): the semantic kernel by class id
    bool all_hot = !qd && ctx_opt(ctx, "NO_ONEHOT") == nullptr;
    for (int p = 0; p < n_pairs && all_hot; p++) all_hot = sources[p]->lid != nullptr && targets[p]->lid != nullptr;
    S->geom.feat = call_feat(dp, all_hot);
  }
  S->geom.instr = dp.kernel_clock || dp.phase_ticks;
  S->geom.verify = dp.verify_lists != 0;
  S->geom.horizon_cap = std::max(1, dp.lean_U);
  if (!qd) ctx->last_xorder = sources[0]->h_order;
  ctx->last_groups = S->G;
  ctx->last_feat = S->geom.feat;
  ctx->last_pairs = n_pairs;
  ctx->last_N = N;
  ctx->last_M = M;
  ctx->last_Kmax = Kmax;
  ctx->last_params = dp;
  ctx->last_csplit = S->geom.csplit;
  ctx->last_stride256 = S->geom.arena.stride256;
  ctx->last_Npad = S->geom.arena.Npad;
  ctx->last_gx = S->gx;
  ctx->last_gy = S->gy;
  ctx->last_layout = S->L;
  return CVO_OK;
}

}  // namespace

extern "C" {

// ---- the chunk graphs of a sub-batch (shared by cvo_align_batch and the batch queue) ----------------------------
// graphs: 0 full (rebuild opportunity + k_assoc_dense in every iteration), 1 lean, 2 short lean, 3 full without the
// dense kernel, 4 calm; 5 / 6 / 7 = lean / short lean / calm WITH the dense kernel (pairs with overflow rows, or in
// the dense regime, whose lists live long enough)
namespace {
struct LoopCfg {
  int U, U_late, lean_U, lean_U2;
  int v_instr;  // 8 when the instrumented kernels run (they have their own cached graphs)
};
inline int graph_lean_base(int v) { return v >= 5 ? (v == 7 ? 4 : v - 4) : v; }
inline int graph_lean_period(const LoopCfg& c, int v, int Uc) {
  const int b = graph_lean_base(v);
  return b == 4 ? Uc : (b == 3 ? 0 : (b == 2 ? c.lean_U2 : c.lean_U));
}
inline int graph_slot(const LoopCfg& c, int v, int Uc) { return v + c.v_instr + (Uc == c.U ? 0 : (Uc == c.U_late && c.U_late != c.U ? 16 : 32)); }

int ensure_graph(cvo_ctx* ctx, const BatchSetup& S, const LaunchGeom* geom, int G, const LoopCfg& cfg, int g, int v, int Uc) {
  const int vi = graph_slot(cfg, v, Uc);
  GraphKey key;
  key.n_pairs = geom[g].n_pairs;
  key.p0 = geom[g].p0;
  key.T = S.T;
  key.gx = S.gx;
  key.gy = S.gy;
  key.nba = S.d.nblk_assoc;
  key.nbc = S.d.nblk_coeff * 64 + S.geom.csplit;
  key.npb = (int)((unsigned)S.geom.npb + ((unsigned)S.geom.dense_blocks << 20));  // (npb < 2^20: Mpad / 256 + rows / 256)
  key.idx16 = S.geom.idx16 ? 1 : 0;
  key.general = S.geom.feat;
  key.U = Uc * 256 + graph_lean_period(cfg, v, Uc) + (v == 3 ? 128 : 0);
  key.flags = (S.geom.instr ? 1 : 0) | (S.geom.verify ? 2 : 0) | (v << 24);
  key.arena = geom[g].arena.base;
  key.stride256 = geom[g].arena.stride256;
  key.Npad = geom[g].arena.Npad;
  if (ctx->graph_exec[g][vi] && ctx->graph_key[g][vi] == key) return CVO_OK;
  if (ctx->graph_exec[g][vi]) {
    (void)hipGraphExecDestroy(ctx->graph_exec[g][vi]);
    ctx->graph_exec[g][vi] = nullptr;
  }
  hipGraph_t gr = nullptr;
  HIP_TRY(ctx, hipStreamBeginCapture(geom[g].stream, hipStreamCaptureModeThreadLocal));
  launch_chunk(ctx, geom[g], Uc, v != 0, graph_lean_period(cfg, v, Uc), v >= 5);
  // (the capture is always ended, whatever the launches reported: a stream left in capture mode would poison
  // every later call on this context)
  const hipError_t e_launch = hipGetLastError();
  hipError_t e = hipStreamEndCapture(geom[g].stream, &gr);
  if (e == hipSuccess && e_launch != hipSuccess) e = e_launch;
  if (e == hipSuccess) e = hipGraphInstantiate(&ctx->graph_exec[g][vi], gr, nullptr, nullptr, 0);
  if (gr) (void)hipGraphDestroy(gr);
  if (e != hipSuccess) {
    ctx->graph_exec[g][vi] = nullptr;
    for (int q = 0; q < G; q++) (void)hipStreamSynchronize(geom[q].stream);  // other groups may be in flight
    return fail(ctx, CVO_E_HIP, std::string("graph capture / instantiate: ") + hipGetErrorString(e));
  }
  ctx->graph_key[g][vi] = key;
  return CVO_OK;
}

// Which graph a sub-batch runs next: `want` = the level its most demanding unfinished pair asked for (2 = a rebuild
// opportunity in every iteration, 1 = short lean, 0 = lean, -1 = calm), `dense` = one of them needs k_assoc_dense.
inline int choose_graph(int want, bool dense, bool allow_lean, bool start_nodense, bool allow_calm, int lean_U2) {
  if (want == 1 && lean_U2 <= 0) want = 2;
  if (!allow_lean) return 0;
  if (want >= 2) return dense ? 0 : (start_nodense ? 3 : 0);
  if (want == 1) return dense ? 6 : 2;
  if (want == 0 || !allow_calm) return dense ? 5 : 1;
  return dense ? 7 : 4;
}
}  // namespace

int cvo_align_batch(cvo_ctx* ctx, const cvo_params_t* params, int n_pairs, const cvo_cloud* const* sources,
                    const cvo_cloud* const* targets, const float* init_T, float* out_T, cvo_align_info_t* infos,
                    const cvo_align_opts_t* opts) {
  if (!ctx) return CVO_E_INVALID;
  if (!init_T || !out_T) return fail(ctx, CVO_E_INVALID, "null transform pointer");
  BatchSetup S;
  DevParams dp;
  const auto t_host0 = std::chrono::steady_clock::now();
  int rc = setup_batch(ctx, params, n_pairs, sources, targets, init_T, opts, 0, 0.f, &S, &dp);
  if (rc != CVO_OK) return rc;
  const auto t_host1 = std::chrono::steady_clock::now();

  const int max_iter = dp.max_iter;
  // Iterations per chunk (= per host check).  A chunk boundary costs a stream ~10 us (graph launch, the event; the
  // status words reach the host by themselves), a longer chunk lets a finished or re-planned sub-batch run on for nothing: 16 iterations for
  // the first 256 (short warm-started solves end there, and the early requests change quickly), 32 afterwards.
  int U = (opts && opts->iters_per_launch > 0) ? opts->iters_per_launch : 16;
  U = std::max(1, std::min(U, std::max(1, max_iter)));
  const bool adaptive_chunks = !(opts && opts->iters_per_launch > 0) && !ctx_opt(ctx, "FIXED_CHUNKS") && max_iter >= 512;
  const int U_late = adaptive_chunks ? 2 * U : U;
  const int n_early_chunks = adaptive_chunks ? 256 / U : 0;
  // The first two chunks of a call are chosen blind (the host learns what a pair wants one chunk behind) and are full
  // graphs: short ones, so that a warm-started pair whose lists outlive dozens of iterations from the start is not held
  // on six launches per iteration for 32 of its few hundred iterations.
  int U_first = adaptive_chunks ? std::max(1, U / 4) : U;
  if (adaptive_chunks && ctx_opt(ctx, "FIRST_U")) U_first = std::max(1, std::min(atoi(ctx_opt(ctx, "FIRST_U")), U));
  int n_first_chunks = U_first != U ? 2 : 0;
  if (U_first != U && ctx_opt(ctx, "FIRST_CHUNKS")) n_first_chunks = std::max(0, atoi(ctx_opt(ctx, "FIRST_CHUNKS")));
  const int graph_mode = opts ? opts->use_graph : 0;
  const bool use_graph = graph_mode != 1;

  // sub-batches on separate streams (see cvo_ctx): contiguous blocks of pairs
  const int G = S.G;
  LaunchGeom geom[cvo_ctx::MAX_GROUPS];
  for (int g = 0; g < G; g++) {
    const int p0 = (int)((long)n_pairs * g / G), p1 = (int)((long)n_pairs * (g + 1) / G);
    geom[g] = S.geom;
    geom[g].group = g;
    geom[g].p0 = p0;
    geom[g].n_pairs = p1 - p0;
    geom[g].arena.base = S.geom.arena.base + S.L.total * (size_t)p0;
    geom[g].stream = ctx->gstream[g];
  }

  HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));  // the setup copies were enqueued on group 0's stream
  for (int g = 0; g < G; g++) {
    if (g) HIP_TRY(ctx, hipStreamWaitEvent(geom[g].stream, ctx->ev_fork, 0));
    launch_init(ctx, geom[g]);
  }
  HIP_TRY(ctx, hipGetLastError());

  if (max_iter > 0) {
    const int lean_U = std::max(1, std::min(dp.lean_U, U));
    const int lean_U2 = std::max(0, std::min(dp.lean_U2, U));
    const LoopCfg cfg{U, U_late, lean_U, lean_U2, S.geom.instr ? 8 : 0};
    auto lean_period = [&](int v, int Uc) { return graph_lean_period(cfg, v, Uc); };
    auto graph_index = [&](int v, int Uc) { return graph_slot(cfg, v, Uc); };
    auto get_graph = [&](int g, int v, int Uc) -> int { return ensure_graph(ctx, S, geom, G, cfg, g, v, Uc); };
    // Chunks are enqueued until every pair has finished.  A pair advances one iteration per slot unless it is
    // waiting in a lean chunk for a rebuild / dense kernel, so the bound below is only a safety net.
    const int n_chunks = (max_iter + U_first - 1) / U_first;
    const int chunk_cap = 4 * n_chunks + 16;
    const bool allow_lean = ctx_opt(ctx, "NO_LEAN") == nullptr;
    int graph_next[cvo_ctx::MAX_GROUPS];  // 0 = full, 1 = lean, 2 = short lean, 3 = full without the dense kernel
    // the first iterations move fast: full graph - for large clouds without the dense kernel (rows that overflow their
    // lists are a small-cloud / huge-lengthscale matter; a pair that has some waits two chunks for the real full graph)
    const bool start_nodense = allow_lean && S.N > 4096;
    const bool allow_calm = dp.calm_U > 0;
    for (int g = 0; g < G; g++) graph_next[g] = start_nodense ? 3 : 0;
    bool all_done = false;
    int ch = 0;
    int n_lean_launch = 0, n_full_launch = 0;
    double t_launch = 0, t_wait = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(now() - t).count(); };
    for (; ch < chunk_cap && !all_done; ch++) {
      const int slot = ch & 1;
      const int Uc = ch < n_first_chunks ? U_first : (ch < n_early_chunks + n_first_chunks ? U : U_late);
      if (ctx_opt(ctx, "VERBOSE") && atoi(ctx_opt(ctx, "VERBOSE")) >= 3) {
        fprintf(stderr, "[cvo] chunk %d (%d iterations): graphs", ch, Uc);
        for (int g = 0; g < G; g++)
          fprintf(stderr, " %s", kGraphNames[graph_next[g]]);
        fprintf(stderr, "\n");
      }
      for (int g = 0; g < G; g++) {
        const int v = graph_next[g];
        (v ? n_lean_launch : n_full_launch)++;
        if (use_graph) {
          rc = get_graph(g, v, Uc);
          if (rc != CVO_OK) return rc;
          const auto tl = now();
          HIP_TRY(ctx, hipGraphLaunch(ctx->graph_exec[g][graph_index(v, Uc)], geom[g].stream));
          t_launch += ms_since(tl);
        } else {
          launch_chunk(ctx, geom[g], Uc, v != 0, lean_period(v, Uc), v >= 5);
          HIP_TRY(ctx, hipGetLastError());
        }
        // (no status copy: the device keeps a mirror of every pair's two words in pinned host memory up to date, and the
        // event's system-scope release makes what the chunk wrote visible - a copy kernel and its two boundaries per
        // chunk and stream were 7 us of a chunk's ~300)
        HIP_TRY(ctx, hipEventRecord(ctx->ev_chk[slot][g], geom[g].stream));
      }
      // keep one chunk of speculation in flight: inspect the chunk before this one
      if (ch >= 1) {
        const int ws = (ch - 1) & 1;
        const auto tw = now();
        for (int g = 0; g < G; g++) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_chk[ws][g]));
        t_wait += ms_since(tw);
        all_done = true;
        for (int g = 0; g < G; g++) {
          const volatile int* hs = ctx->h_status[0] + 2 * geom[g].p0;  // [status[n_g] | want[n_g]], live (may be newer than chunk ch - 1)
          const int ng = geom[g].n_pairs;
          for (int q = 0; q < ng; q++) all_done = all_done && hs[q] != 0;
          // the most demanding unfinished pair of the group decides the level (2 = full, 1 = short lean, 0 = lean,
          // -1 = calm), any of them that needs k_assoc_dense gets it (want_level / want_encode, cvo_kernels.h)
          int want = -1;
          bool dense = false;
          for (int q = 0; q < ng; q++)
            if (hs[q] == 0) {
              const int w = hs[ng + q];
              dense = dense || w == 4 || w >= 8;
              want = std::max(want, w == 4 ? 2 : (w >= 8 ? w - 9 : w));
            }
          graph_next[g] = choose_graph(want, dense, allow_lean, start_nodense, allow_calm, lean_U2);
          if (ctx_opt(ctx, "VERBOSE") && atoi(ctx_opt(ctx, "VERBOSE")) >= 2 && ch < 12) {
            int nw = 0;
            for (int q = 0; q < ng; q++) nw += hs[ng + q] != 0;
            fprintf(stderr, "[cvo] after chunk %d group %d: %d of %d pairs ask for the full graph\n", ch - 1, g, nw, geom[g].n_pairs);
          }
        }
      }
    }
    ctx->last_chunks = ch;
    if (ctx_opt(ctx, "VERBOSE")) fprintf(stderr, "[cvo] host loop: %.2f ms in hipGraphLaunch, %.2f ms waiting for the device\n", t_launch, t_wait);
    ctx->last_lean_launches = n_lean_launch;
    ctx->last_full_launches = n_full_launch;
    if (!all_done) {  // the in-flight chunk may have finished the stragglers; otherwise report it
      for (int g = 0; g < G; g++) HIP_TRY(ctx, hipStreamSynchronize(geom[g].stream));
      bool fin = true;
      for (int g = 0; g < G; g++)
        for (int q = 0; q < geom[g].n_pairs; q++) fin = fin && ((volatile int*)ctx->h_status[0])[2 * geom[g].p0 + q] != 0;
      if (!fin && ch >= chunk_cap) return fail(ctx, CVO_E_HIP, "cvo_align_batch: optimiser loop did not terminate");
    }
  }
  for (int g = 1; g < G; g++) {  // join
    HIP_TRY(ctx, hipEventRecord(ctx->ev_join[g], geom[g].stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[g], 0));
  }
  HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_states.data(), ctx->d_states, sizeof(PairState) * (size_t)n_pairs,
                              hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const auto t_host2 = std::chrono::steady_clock::now();
  float ms = 0;
  HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
  if (ctx_opt(ctx, "VERBOSE"))
    fprintf(stderr, "[cvo] host: setup %.2f ms, enqueue + wait %.2f ms\n",
            std::chrono::duration<double, std::milli>(t_host1 - t_host0).count(),
            std::chrono::duration<double, std::milli>(t_host2 - t_host1).count());
  for (int p = 0; p < n_pairs; p++) {  // CVO_VERIFY_LISTS: a row of the list path differed from the literal scan
    const PairState& st = ctx->h_states[p];
    if (st.sync_err)
      return fail(ctx, CVO_E_HIP, st.sync_err == 1 ? "a block partial of k_assoc never arrived at the flow gate (tagged-partial poll limit)"
                                                    : "a block partial of k_coeff never arrived at the update (tagged-partial poll limit)");
    if (st.verify_err) {
      char msg[256];
      snprintf(msg, sizeof msg,
               "CVO_VERIFY_LISTS: pair %d, iteration %d, row position %d: the list-derived row differs from the literal "
               "scan (%s)", p, st.verify_k, st.verify_pos,
               st.verify_what == 1 ? "nonzero count" : (st.verify_what == 2 ? "column" : "value"));
      return fail(ctx, CVO_E_VERIFY, msg);
    }
  }
  if (ctx_opt(ctx, "VERBOSE")) {
    long builds = 0, stalls = 0, its = 0, adopted = 0;
    for (int p = 0; p < n_pairs; p++) {
      adopted += ctx->h_states[p].n_adopted;
      builds += ctx->h_states[p].n_builds;
      stalls += ctx->h_states[p].n_stalls;
      its += ctx->h_states[p].status ? ctx->h_states[p].iterations : ctx->h_states[p].k;
    }
    if (atoi(ctx_opt(ctx, "VERBOSE")) >= 2)
      for (int p = 0; p < std::min(n_pairs, 4); p++)
        fprintf(stderr, "[cvo]   pair %d: k %d, list allowance used %.3f, per iteration %.5f, want %d, builds %d, ell %.4f (built at %.4f)\n", p,
                ctx->h_states[p].k, ctx->h_states[p].last_used, ctx->h_states[p].last_rate, ctx->h_states[p].want_full,
                ctx->h_states[p].n_builds, ctx->h_states[p].ell, ctx->h_states[p].ell_build);
    fprintf(stderr, "[cvo] %d pairs, %d groups: %d chunks (%d full + %d lean group launches), iterations %ld (%ld with the speculative update adopted), list builds %ld, waits %ld, %.3f ms\n",
            n_pairs, G, ctx->last_chunks, ctx->last_full_launches, ctx->last_lean_launches, its, adopted, builds, stalls, ms);
  }
  for (int p = 0; p < n_pairs; p++) {
    const PairState& st = ctx->h_states[p];
    std::memcpy(out_T + 16 * (size_t)p, st.out_T, sizeof(float) * 16);
    if (infos) {
      infos[p].iterations = st.status ? st.iterations : st.k;
      infos[p].ret = st.ret;
      infos[p].final_ell = st.ell;
      infos[p].final_num_neighbors = st.K;
      infos[p].seconds = (double)ms * 1e-3;
    }
    if (opts && opts->trace && opts->trace_capacity > 0) {
      const int nt = std::min(st.n_trace, opts->trace_capacity);
      if (nt > 0)
        HIP_TRY(ctx, hipMemcpy(opts->trace + (size_t)p * opts->trace_capacity, ctx->h_descs[p].trace,
                               sizeof(cvo_trace_t) * (size_t)nt, hipMemcpyDeviceToHost));
      if (opts->n_trace) opts->n_trace[p] = nt;
    }
  }
  return CVO_OK;
}

int cvo_align_ex(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source, const cvo_cloud* target,
                 const float init_T[16], float out_T[16], cvo_align_info_t* info, const cvo_align_opts_t* opts) {
  if (!ctx) return CVO_E_INVALID;
  if (!source || !target) return fail(ctx, CVO_E_INVALID, "null cloud");
  if (info) std::memset(info, 0, sizeof(*info));
  // empty input: return 0 and leave `transform` untouched (CvoGPU.cu:1614-1617)
  if (source->n == 0 || target->n == 0) return 0;
  const cvo_cloud* src[1] = {source};
  const cvo_cloud* tgt[1] = {target};
  cvo_align_info_t local;
  int rc = cvo_align_batch(ctx, params, 1, src, tgt, init_T, out_T, &local, opts);
  if (rc != CVO_OK) return rc;
  if (info) *info = local;
  return local.ret;
}

int cvo_align(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source, const cvo_cloud* target,
              const float init_T[16], float out_T[16], cvo_align_info_t* info) {
  return cvo_align_ex(ctx, params, source, target, init_T, out_T, info, nullptr);
}

int cvo_batch_poses_to_device(cvo_ctx* ctx, void* dst_device, int n_pairs) {
  if (!ctx || !dst_device || n_pairs <= 0 || n_pairs > ctx->last_pairs)
    return fail(ctx, CVO_E_INVALID, "cvo_batch_poses_to_device: bad argument");
  std::vector<float> poses(16 * (size_t)n_pairs);
  for (int p = 0; p < n_pairs; p++) std::memcpy(&poses[16 * (size_t)p], ctx->h_states[p].out_T, sizeof(float) * 16);
  HIP_TRY(ctx, hipMemcpyAsync(dst_device, poses.data(), sizeof(float) * poses.size(), hipMemcpyHostToDevice,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CVO_OK;
}

}  // extern "C"
