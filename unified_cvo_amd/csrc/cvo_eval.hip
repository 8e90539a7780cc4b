// cvo_eval.hip -- single evaluations: inner_product_gpu / function_angle in one launch of k_overlap or through the list chain, run_single_eval for the association exports.
// A SECTION of the one translation unit cvo_hip.hip (which includes the sections in dependency order and says why it is one
// unit); not compiled on its own.  Shared declarations: cvo_internal.h.
namespace {

int run_single_eval(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source, const cvo_cloud* target,
                    const float Tm[16], float ell, BatchSetup* S, const float* kernel_inv_and_cull = nullptr) {
  DevParams dp;
  const cvo_cloud* src[1] = {source};
  const cvo_cloud* tgt[1] = {target};
  int rc = setup_batch(ctx, params, 1, src, tgt, Tm, nullptr, kernel_inv_and_cull ? 2 : 1, ell, S, &dp, kernel_inv_and_cull);
  if (rc != CVO_OK) return rc;
  launch_init(ctx, S->geom);
  launch_rebuild(ctx, S->geom);
  launch_core(ctx, S->geom, false, 2);  // mode 1: k_coeff is a no-op ...
  hipLaunchKernelGGL(k_update<false>, dim3(1), dim3(64), 0, ctx->stream, ctx->d_descs, ctx->d_params, ctx->d_status, 2);  // ... k_update collects the sums
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_states.data(), ctx->d_states, sizeof(PairState), hipMemcpyDeviceToHost,
                              ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CVO_OK;
}

// inner_product_gpu for n (<= 8) pairs in ONE chain: INIT, the rebuild trio, [k_assoc_dense], k_assoc whose last block
// posts A_sum to pinned host memory - one upload, one graph launch, one synchronisation.  The three inner products of the
// exact function_angle (CvoGPU.cu:1835-1837) are such a batch.  Every value is what the one-pair path returns.
// The inner products of a call in one launch of k_overlap (cvo_k_overlap.h).  *void_out: some row found more than
// nearest_neighbors_max pairs - its first-K truncation needs the hits in ascending original index, i.e. the list chain.
struct OverlapArgs {
  OverlapJob job[3];
  DevParams P;
};
static_assert(sizeof(OverlapArgs) <= 4096, "k_overlap takes its jobs as kernel arguments");
template <int FEAT>
__global__ __launch_bounds__(64 * OV_WAVES) void k_overlap_entry(const OverlapArgs A) {
  k_overlap<FEAT>(A.job[blockIdx.y], A.P);
}

int ensure_tiles(cvo_ctx* ctx, const cvo_cloud* c, hipStream_t s) {
  if (c->tile4) return CVO_OK;
  const int nt = (c->n + 63) / 64;
  float4* t = nullptr;
  HIP_TRY(ctx, hipMalloc(&t, sizeof(float4) * 2 * (size_t)nt));
  hipLaunchKernelGGL(k_tile_spheres, dim3((nt + 3) / 4), dim3(256), 0, s, c->n, c->xs4, t);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    (void)hipFree(t);
    return fail(ctx, CVO_E_HIP, std::string("k_tile_spheres: ") + hipGetErrorString(e));
  }
  c->tile4 = t;
  return CVO_OK;
}

int run_overlap_kernel(cvo_ctx* ctx, const cvo_params_t* params, int n, const cvo_cloud* const* src, const cvo_cloud* const* tgt,
                       const float* Tms, float ell, double* out, bool* void_out) {
  int N = 0, M = 0;
  int rc = check_call(ctx, params, n, src, tgt, nullptr, 1, ell, nullptr, &N, &M);
  if (rc != CVO_OK) return rc;
  if (n > 3) return fail(ctx, CVO_E_INVALID, "run_overlap_kernel: at most three pairs per launch");
  hipStream_t stream = ctx->stream;
  const int tiles_max = (N + 63) / 64;
  if (!ctx->h_ov) HIP_TRY(ctx, hipHostMalloc(&ctx->h_ov, 64, hipHostMallocMapped | hipHostMallocCoherent));
  if (tiles_max > ctx->ov_tiles_cap) {
    if (ctx->d_ov) (void)hipFree(ctx->d_ov);
    ctx->d_ov = nullptr;
    ctx->ov_tiles_cap = 0;
    const size_t bytes = 256 + 3 * sizeof(double) * (size_t)tiles_max;
    HIP_TRY(ctx, hipMalloc(&ctx->d_ov, bytes));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_ov, 0, 256, stream));  // (the gate words; the kernel leaves them at zero)
    ctx->ov_tiles_cap = tiles_max;
  }
  OverlapArgs A;
  std::memset(&A, 0, sizeof(A));
  A.P = make_dev_params(ctx, *params);
  A.P.mode = 1;
  bool all_hot = ctx_opt(ctx, "NO_ONEHOT") == nullptr;
  for (int p = 0; p < n; p++) {
    const cvo_cloud* X = src[p];
    const cvo_cloud* Y = tgt[p];
    if ((rc = ensure_tiles(ctx, X, stream)) != CVO_OK || (rc = ensure_tiles(ctx, Y, stream)) != CVO_OK) return rc;
    all_hot = all_hot && X->lid != nullptr && Y->lid != nullptr;
    OverlapJob& J = A.job[p];
    J.D.N = X->n;
    J.D.M = Y->n;
    J.D.xs4 = X->xs4;
    J.D.ys4 = Y->xs4;
    J.D.xfeat = X->feat;
    J.D.yfeat = Y->feat;
    J.D.xlabel = X->label;
    J.D.ylabel = Y->label;
    J.D.xgeo = X->geo;
    J.D.ygeo = Y->geo;
    J.D.xlid = X->lid;
    J.D.ylid = Y->lid;
    J.xtile = X->tile4;
    J.ytile = Y->tile4;
    J.n_xtiles = (X->n + 63) / 64;
    J.n_ytiles = (Y->n + 63) / 64;
    const float* Tm = Tms + 16 * (size_t)p;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) J.R[3 * i + j] = Tm[4 * j + i];  // CvoGPU.cu:1363-1364 (as fill_pair)
      J.T[i] = Tm[12 + i];
    }
    {
      // |R^T v| <= stretch |v|: 1 (+ rounding) for a rotation, the Frobenius norm for anything else a caller may pass
      double dev = 0, fro = 0;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
          double g = 0;
          for (int k = 0; k < 3; k++) g += (double)J.R[3 * k + i] * (double)J.R[3 * k + j];
          dev = std::max(dev, std::fabs(g - (i == j ? 1.0 : 0.0)));
          fro += (double)J.R[3 * i + j] * (double)J.R[3 * i + j];
        }
      J.stretch = (dev <= 1e-4) ? 1.001f : (float)(std::sqrt(fro) * 1.001);
      if (!std::isfinite(J.stretch)) J.stretch = __builtin_inff();  // (every tile is visited)
    }
    J.ell = ell;
    J.K = params->nearest_neighbors_max;
    J.part = reinterpret_cast<double*>(ctx->d_ov + 256) + (size_t)p * ctx->ov_tiles_cap;
    J.gate = reinterpret_cast<int*>(ctx->d_ov) + 2 * p;
    J.sum_host = reinterpret_cast<double*>(ctx->h_ov) + p;
    J.over_host = reinterpret_cast<int*>(ctx->h_ov + 32) + p;
  }
  const int feat = call_feat(A.P, all_hot);
  const dim3 grid(tiles_max, n), block(64 * OV_WAVES);
  switch (feat) {
    case FEAT_GEO: hipLaunchKernelGGL((k_overlap_entry<FEAT_GEO>), grid, block, 0, stream, A); break;
    case FEAT_COL: hipLaunchKernelGGL((k_overlap_entry<FEAT_COL>), grid, block, 0, stream, A); break;
    case FEAT_HOT: hipLaunchKernelGGL((k_overlap_entry<FEAT_HOT>), grid, block, 0, stream, A); break;
    default: hipLaunchKernelGGL((k_overlap_entry<FEAT_ALL>), grid, block, 0, stream, A); break;
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) {
    (void)hipMemset(ctx->d_ov, 0, 256);  // (a launch that died may have left the gate words behind)
    return fail(ctx, CVO_E_HIP, std::string("k_overlap: ") + hipGetErrorString(e));
  }
  *void_out = false;
  for (int p = 0; p < n; p++) {
    out[p] = reinterpret_cast<const volatile double*>(ctx->h_ov)[p];
    if (reinterpret_cast<const volatile int*>(ctx->h_ov + 32)[p] != 0) *void_out = true;
  }
  ctx->last_pairs = 0;  // (no workspace of the list chain belongs to this call: the debug getters have nothing to read)
  return CVO_OK;
}

int run_inner_products(cvo_ctx* ctx, const cvo_params_t* params, int n, const cvo_cloud* const* src, const cvo_cloud* const* tgt,
                       const float* Tms, float ell, double* out) {
  // One launch when the call has a geometric cut-off to cull by and nobody asked for the list chain (CVO_IP_CHAIN; the
  // instrumented / verifying runs are the chain's); the chain when a row overflows K (first-K needs the original order).
  if (ctx && params && params->is_using_geometry && !params->is_using_kdtree && n <= 3 && ctx_opt(ctx, "IP_CHAIN") == nullptr &&
      ctx_opt(ctx, "VERIFY_LISTS") == nullptr && ctx_opt(ctx, "KERNEL_CLOCK") == nullptr && ctx_opt(ctx, "PHASE_TICKS") == nullptr) {
    bool void_sum = false;
    const int rc = run_overlap_kernel(ctx, params, n, src, tgt, Tms, ell, out, &void_sum);
    if (rc != CVO_OK) return rc;
    if (!void_sum) return CVO_OK;
  }
  BatchSetup S;
  DevParams dp;
  int rc = setup_batch(ctx, params, n, src, tgt, Tms, nullptr, 1, ell, &S, &dp);
  if (rc != CVO_OK) return rc;
  if (S.G != 1) return fail(ctx, CVO_E_INVALID, "run_inner_products: too many pairs for one chain");
  const LaunchGeom& g = S.geom;
  constexpr int VI = cvo_ctx::GRAPH_VARIANTS - 1;
  GraphKey key;
  key.n_pairs = n;
  key.T = S.T;
  key.gx = S.gx;
  key.gy = S.gy;
  key.nba = S.d.nblk_assoc;
  key.npb = (int)((unsigned)g.npb + ((unsigned)g.dense_blocks << 20));  // (dense_blocks <= 2048: twelve bits)
  key.idx16 = g.idx16 ? 1 : 0;
  key.general = g.feat;
  key.flags = (g.instr ? 1 : 0) | (99 << 24);
  key.arena = g.arena.base;
  key.stride256 = g.arena.stride256;
  key.Npad = g.arena.Npad;
  if (!(ctx->graph_exec[0][VI] && ctx->graph_key[0][VI] == key)) {
    if (ctx->graph_exec[0][VI]) (void)hipGraphExecDestroy(ctx->graph_exec[0][VI]);
    ctx->graph_exec[0][VI] = nullptr;
    hipGraph_t gr = nullptr;
    HIP_TRY(ctx, hipStreamBeginCapture(g.stream, hipStreamCaptureModeThreadLocal));
    launch_init(ctx, g);
    launch_rebuild(ctx, g);
    launch_dense(g.stream, g.feat, g.N, g.n_pairs, g.dense_blocks, ctx->d_descs, ctx->d_params, ctx->d_states);
    launch_assoc(g.stream, g.idx16, g.feat, g.instr, g.nba, g.n_pairs, ctx->d_descs, ctx->d_params, ctx->d_states, g.arena, 8);
    const hipError_t e_launch = hipGetLastError();
    hipError_t e = hipStreamEndCapture(g.stream, &gr);
    if (e == hipSuccess && e_launch != hipSuccess) e = e_launch;
    if (e == hipSuccess) e = hipGraphInstantiate(&ctx->graph_exec[0][VI], gr, nullptr, nullptr, 0);
    if (gr) (void)hipGraphDestroy(gr);
    if (e != hipSuccess) {
      ctx->graph_exec[0][VI] = nullptr;
      return fail(ctx, CVO_E_HIP, std::string("inner product graph: ") + hipGetErrorString(e));
    }
    ctx->graph_key[0][VI] = key;
  }
  HIP_TRY(ctx, hipGraphLaunch(ctx->graph_exec[0][VI], g.stream));
  HIP_TRY(ctx, hipStreamSynchronize(g.stream));
  const volatile double* res = reinterpret_cast<const volatile double*>(ctx->h_status[1]);
  for (int p = 0; p < n; p++) {
    out[p] = res[p];
    ctx->h_states[p].asum = res[p];
  }
  return CVO_OK;
}

}  // namespace

extern "C" {

#ifdef CVO_OV_STAMPS
int cvo_debug_overlap_ticks(unsigned long long* out) {  // experiment builds only
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ov_ticks), sizeof(unsigned long long) * 4096 * 8) == hipSuccess ? 0 : -1;
}
#endif

int cvo_inner_product(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source, const cvo_cloud* target,
                      const float T[16], float ell, float* out) {
  if (!ctx || !out || !T) return fail(ctx, CVO_E_INVALID, "cvo_inner_product: bad argument");
  if (!source || !target) return fail(ctx, CVO_E_INVALID, "null cloud");
  if (source->n == 0 || target->n == 0) {
    *out = 0.f;
    return CVO_OK;
  }
  const cvo_cloud* src[1] = {source};
  const cvo_cloud* tgt[1] = {target};
  double v = 0;
  const int rc = run_inner_products(ctx, params, 1, src, tgt, T, ell, &v);
  if (rc != CVO_OK) return rc;
  *out = (float)v;
  return CVO_OK;
}

int cvo_function_angle(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source, const cvo_cloud* target,
                       const float T[16], float ell, int is_approximate, float* out) {
  // function_angle, CvoGPU.cu:1814-1846
  if (!ctx || !out || !T) return fail(ctx, CVO_E_INVALID, "cvo_function_angle: bad argument");
  if (!source || !target) return fail(ctx, CVO_E_INVALID, "null cloud");
  if (source->n == 0 || target->n == 0) {
    *out = 0.f;
    return CVO_OK;
  }
  const float identity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float fxfz = 0, fx_norm = 0, fz_norm = 0;
  if (is_approximate) {
    const int rc = cvo_inner_product(ctx, params, source, target, T, ell, &fxfz);
    if (rc != CVO_OK) return rc;
    fx_norm = (float)std::sqrt((double)source->n);
    fz_norm = (float)std::sqrt((double)target->n);
  } else {
    // the three inner products of CvoGPU.cu:1829-1837 - <fx, fz>, <fx, fx>, <fz, fz> - as one three-pair batch: one chain
    // of launches instead of three (each value is what its own call returns: a pair's sums do not depend on its company)
    const cvo_cloud* src[3] = {source, source, target};
    const cvo_cloud* tgt[3] = {target, source, target};
    float Ts[48];
    std::memcpy(Ts, T, sizeof(float) * 16);
    std::memcpy(Ts + 16, identity, sizeof(float) * 16);
    std::memcpy(Ts + 32, identity, sizeof(float) * 16);
    double v[3] = {0, 0, 0};
    const int rc = run_inner_products(ctx, params, 3, src, tgt, Ts, ell, v);
    if (rc != CVO_OK) return rc;
    fxfz = (float)v[0];
    fx_norm = std::sqrt((float)v[1]);
    fz_norm = std::sqrt((float)v[2]);
  }
  *out = fxfz / (fx_norm * fz_norm);
  return CVO_OK;
}

}  // extern "C"
