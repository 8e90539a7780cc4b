// cvo_debug.hip -- test / profiling hooks of include/cvo_hip_debug.h: the device scalar maths on caller inputs, kernel replays and clocks, counters of the last call.
// A SECTION of the one translation unit cvo_hip.hip (which includes the sections in dependency order and says why it is one
// unit); not compiled on its own.  Shared declarations: cvo_internal.h.
extern "C" {

int cvo_debug_scalar_math(cvo_ctx* ctx, int op, int n, const double* in, double* out) {
  if (!ctx || !in || !out || n <= 0 || op < 0 || op > 13) return fail(ctx, CVO_E_INVALID, "cvo_debug_scalar_math: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t n_in = op == 7 ? (size_t)n + 2 : 16 * (size_t)n, n_out = op == 7 ? (size_t)n : 16 * (size_t)n;
  double *d_in = nullptr, *d_out = nullptr;
  PairState* d_st = nullptr;
  int rc = CVO_OK;
  auto cleanup = [&]() {
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (d_st) (void)hipFree(d_st);
  };
  if (hipMalloc(&d_in, sizeof(double) * n_in) != hipSuccess || hipMalloc(&d_out, sizeof(double) * n_out) != hipSuccess ||
      hipMalloc(&d_st, sizeof(PairState)) != hipSuccess) {
    cleanup();
    return fail(ctx, CVO_E_NOMEM, "cvo_debug_scalar_math: hipMalloc failed");
  }
  hipError_t e = hipMemcpyAsync(d_in, in, sizeof(double) * n_in, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, sizeof(double) * n_out, ctx->stream);
  if (e == hipSuccess) e = hipMemsetAsync(d_st, 0, sizeof(PairState), ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_scalar_math, dim3(op == 7 ? 1 : n), dim3(64), 0, ctx->stream, op, n, d_in, d_out, d_st);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(double) * n_out, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) rc = fail(ctx, CVO_E_HIP, std::string("cvo_debug_scalar_math: ") + hipGetErrorString(e));
  cleanup();
  return rc;
}

int cvo_debug_cloud_order(const cvo_cloud* c, int* out) {
  if (!c || !out) return CVO_E_INVALID;
  for (int r = 0; r < c->n; r++) out[r] = c->h_order[r];
  return CVO_OK;
}

int cvo_debug_device_memory(cvo_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
  if (!ctx || !free_bytes || !total_bytes) return fail(ctx, CVO_E_INVALID, "cvo_debug_device_memory: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemGetInfo(free_bytes, total_bytes));
  return CVO_OK;
}

int cvo_debug_verified_rows(cvo_ctx* ctx, unsigned long long* rows) {
  if (!ctx || !rows || ctx->last_pairs < 1) return fail(ctx, CVO_E_INVALID, "cvo_debug_verified_rows: bad argument");
  unsigned long long t = 0;
  for (int p = 0; p < ctx->last_pairs; p++) t += ctx->h_states[p].verify_rows;
  *rows = t;
  return CVO_OK;
}

int cvo_debug_last_candidates(cvo_ctx* ctx, unsigned long long* out) {
  if (!ctx || !out || ctx->last_pairs < 1) return fail(ctx, CVO_E_INVALID, "cvo_debug_last_candidates: bad argument");
  *out = ctx->h_states[0].ncand;
  return CVO_OK;
}

int cvo_debug_time_kernels(cvo_ctx* ctx, int reps, float* ms_assoc, float* ms_coeff) {
  if (!ctx || reps <= 0 || ctx->last_pairs < 1 || !ms_assoc || !ms_coeff)
    return fail(ctx, CVO_E_INVALID, "cvo_debug_time_kernels: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // the per-iteration launches of the optimiser loop (one per sub-batch), replayed on the state the last call
  // left behind: same lists, same rows, same arithmetic; k_coeff's last block runs the update without writing
  // anything back
  const int n_pairs = ctx->last_pairs, G = ctx->last_groups;
  const bool idx16 = ctx->last_M < 65536;
  const DevParams& dp = ctx->last_params;
  const int general = ctx->last_feat;
  const bool instr = dp.kernel_clock || dp.phase_ticks;
  const int nba = (ctx->last_N + ASSOC_THREADS - 1) / ASSOC_THREADS;
  float out[2] = {0.f, 0.f};
  for (int which = 0; which < 2; which++) {
    auto sweep = [&]() {
      for (int g = 0; g < G; g++) {
        const int p0 = (int)((long)n_pairs * g / G), p1 = (int)((long)n_pairs * (g + 1) / G);
        const ArenaArg A{ctx->arena + ((size_t)ctx->last_stride256 << 8) * (size_t)p0, ctx->last_stride256, ctx->last_Npad};
        if (which == 0)
          launch_assoc(ctx->stream, idx16, general, instr, nba, p1 - p0, ctx->d_descs + p0, ctx->d_params, ctx->d_states + p0, A,
                       2);
        else
          launch_coeff(ctx->stream, instr, nba, ctx->last_csplit, p1 - p0, ctx->d_descs + p0, ctx->d_params, ctx->d_states + p0,
                       A, 8 | 2);
      }
    };
    sweep();  // warm-up
    HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    for (int r = 0; r < reps; r++) sweep();
    HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipEventElapsedTime(&out[which], ctx->ev_start, ctx->ev_stop));
    out[which] /= (float)(reps * G);
  }
  if (dp.phase_ticks) {  // where the blocks of the last sub-batch's launches spent their time (see g_phase_ticks)
    static unsigned long long h[2][8192][4];
    HIP_TRY(ctx, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_ticks), sizeof(h)));
    const int np = n_pairs - (int)((long)n_pairs * (G - 1) / G);
    for (int which = 0; which < 2; which++) {
      const int nb = std::min(4096, 8 * ((np + 7) / 8) * nba * (which ? ctx->last_csplit : 1));
      double sum[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
      int cnt = 0;
      for (int b = 0; b < nb; b++) {
        if (!h[which][b][0] || !h[which][b][3]) continue;
        for (int q = 0; q < 3; q++) {
          const double d = (double)(long long)(h[which][b][q + 1] - h[which][b][q]);
          sum[q] += d;
          mx[q] = std::max(mx[q], d);
        }
        cnt++;
      }
      if (!cnt) continue;
      fprintf(stderr, "[cvo] %s: %d blocks; ticks (avg / max) %s %.0f / %.0f, row loop %.0f / %.0f, %s %.0f / %.0f\n",
              which ? "k_coeff" : "k_assoc", cnt, which ? "prologue + twist" : "prologue", sum[0] / cnt, mx[0], sum[1] / cnt,
              mx[1], which ? "reduction + counter" : "reduction + flow gate", sum[2] / cnt, mx[2]);
      if (which)
        for (int p = 0; p < std::min(np, 3); p++)
          fprintf(stderr, "[cvo]   pair %d, updating block: entry -> counter %.0f, update %.0f ticks\n", p,
                  (double)(long long)(h[1][4096 + p][1] - h[1][4096 + p][0]),
                  (double)(long long)(h[1][4096 + p][2] - h[1][4096 + p][1]));
      if (which) {
        unsigned long long u[8];
        HIP_TRY(ctx, hipMemcpyFromSymbol(u, HIP_SYMBOL(g_upd_ticks), sizeof(u)));
        fprintf(stderr, "[cvo]   inside the update (last pair to run it): reduce %lld, step %lld, pose + distance + indicator %lld, "
                        "update_tf + list bookkeeping %lld, rest %lld, write-back %lld ticks\n",
                (long long)(u[1] - u[0]), (long long)(u[2] - u[1]), (long long)(u[3] - u[2]), (long long)(u[4] - u[3]),
                (long long)(u[5] - u[4]), (long long)(u[6] - u[5]));
      }
    }
  }
  *ms_assoc = out[0];
  *ms_coeff = out[1];
  return CVO_OK;
}

int cvo_debug_kernel_clock(cvo_ctx* ctx, float* ms_assoc, float* ms_coeff, unsigned long long* launches) {
  if (!ctx || ctx->last_pairs < 1 || !ms_assoc || !ms_coeff)
    return fail(ctx, CVO_E_INVALID, "cvo_debug_kernel_clock: bad argument");
  if (!ctx->last_params.kernel_clock) return fail(ctx, CVO_E_INVALID, "cvo_debug_kernel_clock: the last call ran without CVO_KERNEL_CLOCK");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->clock_ms_per_tick <= 0.0) {  // the counter's rate, against HIP events around a kernel that waits 1e6 ticks
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    for (int rep = 0; rep < 2; rep++) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
      hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 0, ctx->stream, 1000000ull);
      HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
    }
    ctx->clock_ms_per_tick = (double)ms / 1e6;
  }
  double sum[2] = {0, 0}, n[2] = {0, 0};
  for (int p = 0; p < ctx->last_pairs; p++)
    for (int w = 0; w < 2; w++) {
      sum[w] += (double)ctx->h_states[p].clk_sum[w];
      n[w] += (double)ctx->h_states[p].clk_n[w];
    }
  *ms_assoc = n[0] > 0 ? (float)(sum[0] / n[0] * ctx->clock_ms_per_tick) : 0.f;
  *ms_coeff = n[1] > 0 ? (float)(sum[1] / n[1] * ctx->clock_ms_per_tick) : 0.f;
  if (launches) *launches = (unsigned long long)n[1];
  return CVO_OK;
}

int cvo_debug_list_builds(cvo_ctx* ctx, unsigned long long* builds, unsigned long long* iterations,
                          unsigned long long* candidate_evaluations) {
  if (!ctx || !builds || ctx->last_pairs < 1) return fail(ctx, CVO_E_INVALID, "cvo_debug_list_builds: bad argument");
  unsigned long long b = 0, it = 0, ce = 0;
  for (int p = 0; p < ctx->last_pairs; p++) {
    b += (unsigned long long)ctx->h_states[p].n_builds;
    it += (unsigned long long)ctx->h_states[p].iterations;
    ce += ctx->h_states[p].ncand_total;
  }
  *builds = b;
  if (iterations) *iterations = it;
  if (candidate_evaluations) *candidate_evaluations = ce;
  return CVO_OK;
}

int cvo_debug_row_classes(cvo_ctx* ctx, int pair, int* overflow_rows, int* scanned_rows, int* dense_regime) {
  if (!ctx || pair < 0 || pair >= ctx->last_pairs) return fail(ctx, CVO_E_INVALID, "cvo_debug_row_classes: bad argument");
  const PairState& st = ctx->h_states[pair];
  if (overflow_rows) *overflow_rows = st.n_ovf;
  if (scanned_rows) *scanned_rows = st.n_scan;
  if (dense_regime) *dense_regime = st.all_dense;
  return CVO_OK;
}

int cvo_debug_scan_stats(cvo_ctx* ctx, unsigned long long* tiles, int* rows_per_tile, int* targets_per_tile) {
  if (!ctx || !tiles || ctx->last_pairs < 1) return fail(ctx, CVO_E_INVALID, "cvo_debug_scan_stats: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  unsigned long long total = 0;
  for (int p = 0; p < ctx->last_pairs; p++) {
    unsigned long long v = 0;
    HIP_TRY(ctx, hipMemcpy(&v, ctx->h_descs[p].tile_count, sizeof(v), hipMemcpyDeviceToHost));
    total += v;
  }
  *tiles = total;
  if (rows_per_tile) *rows_per_tile = ROWS_PER_GROUP;
  if (targets_per_tile) *targets_per_tile = 64 * ctx->last_params.T;
  return CVO_OK;
}

int cvo_debug_last_geometry(cvo_ctx* ctx, int* n_groups, int* pairs_per_group) {
  if (!ctx || ctx->last_pairs < 1) return fail(ctx, CVO_E_INVALID, "cvo_debug_last_geometry: bad argument");
  if (n_groups) *n_groups = ctx->last_groups;
  if (pairs_per_group) *pairs_per_group = (ctx->last_pairs + ctx->last_groups - 1) / ctx->last_groups;
  return CVO_OK;
}

int cvo_debug_time_scan(cvo_ctx* ctx, int reps, float* ms) {
  if (!ctx || !ms || reps <= 0 || ctx->last_pairs < 1)
    return fail(ctx, CVO_E_INVALID, "cvo_debug_time_scan: bad argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // the same launches the optimiser loop issues: one k_scan per sub-batch, here back to back on one stream
  const int n_pairs = ctx->last_pairs, G = ctx->last_groups;
  const DevParams& dp = ctx->last_params;
  const int variant = 1;  // force the scan for pairs whose lists are current (k_scan's `force` bits: 2 = no emission, 4 = no fine tiles)
  auto sweep = [&]() {
    for (int g = 0; g < G; g++) {
      const int p0 = (int)((long)n_pairs * g / G), p1 = (int)((long)n_pairs * (g + 1) / G);
      launch_scan(ctx->stream, dp.T, dim3(ctx->last_gx, ctx->last_gy, p1 - p0), ctx->d_descs + p0, ctx->d_params,
                  ctx->d_states + p0, variant);
    }
  };
  sweep();  // warm-up
  HIP_TRY(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
  for (int r = 0; r < reps; r++) sweep();
  HIP_TRY(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
  HIP_TRY(ctx, hipGetLastError());
  // the extra scans leave slice bits behind; clean them so the workspace stays consistent
  for (int p = 0; p < n_pairs; p++) {
    const PairDesc& D = ctx->h_descs[p];
    HIP_TRY(ctx, hipMemsetAsync(D.rowbits, 0, sizeof(unsigned) * (size_t)(ctx->last_N + 4) * D.rbw, ctx->stream));
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  float t = 0;
  HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->ev_start, ctx->ev_stop));
  *ms = t / (reps * G);
  return CVO_OK;
}

}  // extern "C"
