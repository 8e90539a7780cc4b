// cvo_ctx.hip -- contexts: the hardware-queue contract, the stream pool, options, the workspace of a call (arena, control block, layout of one pair), defaults of cvo_params_t.
// A SECTION of the one translation unit cvo_hip.hip (which includes the sections in dependency order and says why it is one
// unit); not compiled on its own.  Shared declarations: cvo_internal.h.
// ---- the hardware-queue contract ---------------------------------------------------------------------------------
// A batch runs on four sub-batch streams that must sit on four DIFFERENT hardware queues (two streams on one queue take
// turns kernel by kernel: 0.37 s instead of 0.25 s per step measured under torchrun, where RCCL brings streams of its
// own; see also the note in cvo_ctx_create).  HIP deals streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and
// reads that variable once, when the runtime initialises - i.e. at the process's first HIP call.  So:
//   * cvo_process_hint_hw_queues() puts GPU_MAX_HW_QUEUES=8 into the environment unless the variable is already set or
//     CVO_NO_HW_QUEUE_HINT is: an EXPLICIT call a host makes before its first HIP call and before it starts threads
//     (unified_cvo_amd/_capi.py does right after loading the library, cvo::CvoGPU's constructor before its context);
//     at load time only with CVO_HW_QUEUE_HINT_AT_LOAD=1;
//   * cvo_ctx_create checks what the variable says NOW and, below 8, leaves an advisory text in cvo_ctx_advice() and
//     prints it once per process (stderr) - the case of a host that initialised HIP first with the default, or
//     that set a smaller value on purpose.
namespace {
bool g_hw_queue_hint_set = false;
void hw_queue_hint() {
  if (std::getenv("CVO_NO_HW_QUEUE_HINT")) return;
  if (!std::getenv("GPU_MAX_HW_QUEUES")) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    g_hw_queue_hint_set = true;
  }
}
// At LOAD time only on request (CVO_HW_QUEUE_HINT_AT_LOAD=1): a library constructor that edits the environment changes HIP's
// queue allocation for the whole host process behind its back, and setenv is not safe against getenv in other threads.
// The hint is an explicit call - cvo_process_hint_hw_queues() - that a host makes where it controls the ordering: before
// its first HIP call, before it starts threads (the Python wrapper and the C++ veneer's CvoGPU constructor do).
__attribute__((constructor(101))) void cvo_hw_queue_hint_at_load() {
  if (std::getenv("CVO_HW_QUEUE_HINT_AT_LOAD")) hw_queue_hint();
}
}  // namespace

extern "C" int cvo_process_hint_hw_queues(void) {
  hw_queue_hint();
  const char* q = std::getenv("GPU_MAX_HW_QUEUES");
  return q ? atoi(q) : 4;
}

namespace {

struct Dims {
  int Mpad, nchunks, rbw_max, nblk_assoc, nblk_coeff, NG, NGpad, Npad;
};

PairLayout make_layout(int N, int M, int Kmax, int trace_capacity, bool long_lists, Dims* d) {
  const int Mpad = (int)align_up((size_t)M, 512);
  const int nchunks = Mpad / 64;
  const int rbw_max = (int)align_up((size_t)(nchunks + 31) / 32, 4);  // slice bits per row, enough for T = 1
  const int nba = (N + ASSOC_THREADS - 1) / ASSOC_THREADS;
  const int nbc = nba;  // the coefficient phase uses the association's row blocks
  const int NG = (N + ROWS_PER_GROUP - 1) / ROWS_PER_GROUP;
  const int NGpad = (int)align_up((size_t)NG, 64) + 64;
  PairLayout L{};
  // the row arrays of the per-iteration kernels first, at the fixed offsets of cvo_device.h (row_off_*)
  const int Npad = (int)align_up((size_t)N, ROW_PAD);
  L.cand_cnt = row_off_cand_cnt(Npad);
  L.ip = row_off_ip(Npad);
  L.nnz_row = row_off_nnz(Npad);
  L.xp4 = row_off_xp4(Npad);
  L.cand_j = row_off_cand_j(Npad);  // ASSOC_CAP16 x u16 == ASSOC_CAP32 x i32 == 128 bytes per row
  L.ell = row_off_ell(Npad);
  L.ell_j = align_up(L.ell + sizeof(EllEntry) * (size_t)Npad * Kmax, 256);
  size_t off = align_up(L.ell_j + sizeof(int) * (size_t)Npad * Kmax, 256);
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  L.ycull = take(sizeof(float4) * (size_t)Mpad);
  L.xcull = take(sizeof(float4) * (size_t)(N + XCULL_PAD));
  L.gbox = take(sizeof(float4) * 2 * (size_t)NGpad);
  L.cellbox = take(sizeof(float4) * 2 * (size_t)(NGpad / 16));
  L.sbox = take(sizeof(float4) * 2 * (size_t)nchunks);
  L.masks = take(sizeof(unsigned long long) * ((size_t)N + 8) * nchunks);
  L.rowbits = take(sizeof(unsigned) * (size_t)(N + 4) * rbw_max);
  L.row_cnt = take(sizeof(int) * (size_t)N);
  L.tile_count = take(sizeof(unsigned long long));
  L.ovf_rows = take(sizeof(int) * (size_t)N);
  L.ovf_bits = take(sizeof(unsigned long long) * (((size_t)N + 63) / 64 + 4));
  L.gate = take(sizeof(int));
  L.gate_flow = take(sizeof(int));
  L.dense_off = take(sizeof(int) * (size_t)N);
  L.dense_rel = take(sizeof(int) * (size_t)N);
  L.ovf_wsum = take(sizeof(int) * (((size_t)N + 63) / 64 + 4));
  L.word_base = take(sizeof(int) * (((size_t)N + 63) / 64 + 5));
  L.done = take(sizeof(int));
  L.rowperm = take(sizeof(int) * (size_t)N);
  L.iorig = take(sizeof(int) * (size_t)N);
  L.long_stamp = take(sizeof(unsigned long long) * (size_t)N);
  L.long_j = long_lists ? take(sizeof(unsigned short) * (size_t)N * LONG_CAP) : 0;
  L.rowres = take(sizeof(RowRes) * (size_t)N);
  // (rows x the pair's own coefficient split, coeff_split(): one slice above 4096 points, at most 32768 / rows below)
  L.rowcoef = take(sizeof(double) * 4 * (size_t)std::max(N, 32768));
  L.flow_part = take(sizeof(unsigned long long) * FLOW_GRANULES * (size_t)nba);
  L.cnt_part = take(sizeof(unsigned long long) * 4 * (size_t)nba);
  L.coef_part = take(sizeof(unsigned long long) * COEF_GRANULES * (size_t)nbc * COEFF_SPLIT_MAX);
  L.shadow = take(sizeof(unsigned long long) * SHADOW_WORDS);
  L.trace = take(sizeof(cvo_trace_t) * (size_t)std::max(trace_capacity, 0));
  L.total = off;
  d->Mpad = Mpad;
  d->nchunks = nchunks;
  d->rbw_max = rbw_max;
  d->nblk_assoc = nba;
  d->nblk_coeff = nbc;
  d->NG = NG;
  d->NGpad = NGpad;
  d->Npad = Npad;
  return L;
}

void free_workspace(cvo_ctx* c) {
  if (c->arena) (void)hipFree(c->arena);
  if (c->d_ov) (void)hipFree(c->d_ov);
  if (c->h_ov) (void)hipHostFree(c->h_ov);
  c->d_ov = c->h_ov = nullptr;
  c->ov_tiles_cap = 0;
  if (c->d_ctl) (void)hipFree(c->d_ctl);
  if (c->h_ctl) (void)hipHostFree(c->h_ctl);
  for (int i = 0; i < 2; i++)
    if (c->h_status[i]) (void)hipHostFree(c->h_status[i]);
  c->arena = nullptr;
  c->d_ctl = c->h_ctl = nullptr;
  c->d_params = nullptr;
  c->d_descs = nullptr;
  c->d_states = nullptr;
  c->d_status = nullptr;
  c->h_status[0] = c->h_status[1] = nullptr;
  c->arena_bytes = 0;
  c->cap_pairs = 0;
}

void drop_graphs(cvo_ctx* c) {
  for (int g = 0; g < cvo_ctx::MAX_GROUPS; g++)
    for (int v = 0; v < cvo_ctx::GRAPH_VARIANTS; v++)
      if (c->graph_exec[g][v]) {
        (void)hipGraphExecDestroy(c->graph_exec[g][v]);
        c->graph_exec[g][v] = nullptr;
      }
}

int ensure_workspace(cvo_ctx* c, int n_pairs, size_t bytes_per_pair) {
  if (n_pairs > c->cap_pairs) {
    if (c->d_ctl) (void)hipFree(c->d_ctl);
    if (c->h_ctl) (void)hipHostFree(c->h_ctl);
    for (int i = 0; i < 2; i++)
      if (c->h_status[i]) (void)hipHostFree(c->h_status[i]);
    c->d_ctl = c->h_ctl = nullptr;
    c->d_params = nullptr;
    c->d_descs = nullptr;
    c->d_states = nullptr;
    c->d_status = nullptr;
    c->cap_pairs = 0;
    // control block: [DevParams | status words: per sub-batch status[n_g], want[n_g] | PairDesc[n] | PairState[n]]
    c->ctl_off_status = align_up(sizeof(DevParams), 256);
    c->ctl_off_descs = align_up(c->ctl_off_status + sizeof(int) * 2 * (size_t)n_pairs, 256);
    c->ctl_off_states = align_up(c->ctl_off_descs + sizeof(PairDesc) * (size_t)n_pairs, 256);
    c->ctl_bytes = align_up(c->ctl_off_states + sizeof(PairState) * (size_t)n_pairs, 256);
    HIP_TRY(c, hipMalloc(&c->d_ctl, c->ctl_bytes));
    HIP_TRY(c, hipHostMalloc(&c->h_ctl, c->ctl_bytes, hipHostMallocDefault));
    c->d_params = (DevParams*)c->d_ctl;
    c->d_status = (int*)(c->d_ctl + c->ctl_off_status);
    c->d_descs = (PairDesc*)(c->d_ctl + c->ctl_off_descs);
    c->d_states = (PairState*)(c->d_ctl + c->ctl_off_states);
    for (int i = 0; i < 2; i++)  // fine-grained: what the device writes there needs no cache maintenance to be seen
      HIP_TRY(c, hipHostMalloc(&c->h_status[i], sizeof(int) * 2 * (size_t)n_pairs, hipHostMallocMapped | hipHostMallocCoherent));
    c->cap_pairs = n_pairs;
    drop_graphs(c);
  }
  const size_t need = bytes_per_pair * (size_t)n_pairs;
  if (need > c->arena_bytes) {
    if (c->arena) (void)hipFree(c->arena);
    c->arena = nullptr;
    c->arena_bytes = 0;
    hipError_t e = hipMalloc(&c->arena, need);
    if (e != hipSuccess) return fail(c, CVO_E_NOMEM, "workspace hipMalloc failed: " + std::string(hipGetErrorString(e)));
    c->arena_bytes = need;
    drop_graphs(c);
  }
  return CVO_OK;
}

int coeff_split(int n) {
  int s = 1;
  while (s < 8 && (long)n * (2 * s) <= 8192) s *= 2;
  // tiny clouds (the dense regime walks hundreds of entries per row): a few more slices as long as the launch stays small
  while (s >= 8 && s < COEFF_SPLIT_MAX && (long)n * (2 * s) <= 32768) s *= 2;
  return s;
}
}  // namespace

extern "C" {

const char* cvo_version(void) { return CVO_VERSION_STRING; }

void cvo_params_default(cvo_params_t* p) {
  // CvoParams::CvoParams(), CvoParams.hpp:75-126
  std::memset(p, 0, sizeof(*p));
  p->ell_init_first_frame = 0.5f;
  p->ell_init = 0.5f;
  p->ell_min = 0.05f;
  p->min_ell_iter_limit = 1;
  p->ell_max = 1.2f;
  p->dl = 0;
  p->dl_step = 0.3;
  p->sigma = 0.1f;
  p->sp_thres = 0.0006f;
  p->c = 7.0f;
  p->d = 7.0f;
  p->c_ell = 0.15f;
  p->c_sigma = 0.6f;
  p->s_ell = 0.1f;
  p->s_sigma = 0.8f;
  p->MAX_ITER = 10000;
  p->min_step = 2e-5f;
  p->eps = 0.00005f;
  p->eps_2 = 0.000012f;
  p->max_step = 0.8f;  // uninitialised upstream; see DESIGN.md
  p->step = 0.f;       // uninitialised upstream, unused by the path
  p->ell_decay_rate = 0.9f;
  p->ell_decay_rate_first_frame = 0.99f;
  p->ell_decay_start = 30;
  p->ell_decay_start_first_frame = 300;
  p->indicator_window_size = 15;
  p->indicator_stable_threshold = 0.2f;
  p->is_pcl_visualization_on = 0;
  p->is_using_least_square = 0;
  p->is_ell_adaptive = 0;
  p->is_full_ip_matrix = 0;
  p->is_using_geometry = 1;
  p->is_using_intensity = 0;
  p->is_using_semantics = 0;
  p->is_using_range_ell = 0;
  p->is_using_kdtree = 0;
  p->is_using_geometric_type = 0;
  p->is_exporting_association = 0;
  p->multiframe_using_cpu = 1;
  p->multiframe_max_iters = 200;
  p->nearest_neighbors_max = 512;
  p->multiframe_ell_init = 0.15f;
  p->multiframe_ell_min = 0.05f;
  p->multiframe_iter_per_ell = 10;
  p->multiframe_ell_decay_rate = 0.7f;
  p->multiframe_iterations_per_ell = 50;
  p->multiframe_iterations_per_solve = 8;
  p->multiframe_downsample_voxel_size = 0.5f;
  p->multiframe_expected_points = 1000;
  p->multiframe_num_neighbors = 128;
  p->multiframe_min_nonzeros = 300;
  p->multiframe_least_squares_num_threads = 24;
}

// The streams of a context - group 0 (= the context's stream), seven more sub-batch streams, the upload stream - are
// handed back to a per-device pool when the context is destroyed and reused, in the same roles, by the next context of
// that device.  HIP deals streams onto hardware queues as they are first used; a context created after another one had
// been DESTROYED found its four sub-batch streams sharing queues (214 ms instead of 64 ms per headline step,
// scripts/upload_probe.py) however carefully it ordered their creation.  Streams that are never destroyed keep the
// queues the first context's careful order gave them.  Contexts alive at the same time still get streams of their own.
namespace {
struct StreamSet {
  hipStream_t g[cvo_ctx::MAX_GROUPS] = {};
  hipStream_t upload = nullptr;
};
std::mutex g_stream_pool_mutex;
std::map<int, std::vector<StreamSet>> g_stream_pool;
}  // namespace

int cvo_ctx_create(int device, cvo_ctx** out) {
  if (!out) return CVO_E_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CVO_E_HIP;
  if (hipSetDevice(device) != hipSuccess) return CVO_E_HIP;
  cvo_ctx* c = new cvo_ctx();
  c->device = device;
  for (const char* name : kOptionNames)  // the ONLY place the library reads the environment
    if (const char* v = std::getenv((std::string("CVO_") + name).c_str())) c->opt[name] = v;
  bool pooled = false;
  {
    std::lock_guard<std::mutex> lk(g_stream_pool_mutex);
    auto& pool = g_stream_pool[device];
    if (!pool.empty()) {
      const StreamSet ss = pool.back();
      pool.pop_back();
      for (int g = 0; g < cvo_ctx::MAX_GROUPS; g++) c->gstream[g] = ss.g[g];
      c->stream = ss.g[0];
      c->upload_stream = ss.upload;
      pooled = true;
    }
  }
  bool ok = (pooled || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess) &&
            // (nothing here may run on the NULL stream - a synchronous hipMemset, say: its hardware queue would then be the
            // first one this process creates - see the note on the sub-batch streams below)
            hipEventCreate(&c->ev_start) == hipSuccess &&
            hipEventCreate(&c->ev_stop) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess;
  {  // the hardware-queue contract (top of this file)
    const char* q = std::getenv("GPU_MAX_HW_QUEUES");
    const int nq = q ? atoi(q) : 4;
    if (nq < 8) {
      char msg[400];
      snprintf(msg, sizeof msg,
               "GPU_MAX_HW_QUEUES is %s%s: batches run on four sub-batch streams next to the upload stream and whatever "
               "RCCL / the host application adds; with fewer than 8 hardware queues streams share a queue and take turns "
               "(measured: 0.37 s instead of 0.25 s per 64-pair step under torchrun).  Export GPU_MAX_HW_QUEUES=8 before "
               "the process's first HIP call",
               q ? q : "unset (HIP's default: 4)", q ? "" : ": cvo_process_hint_hw_queues() was not called before HIP initialised");
      c->advice = msg;
      static std::atomic<bool> said{false};
      if (!said.exchange(true) && !std::getenv("CVO_QUIET")) fprintf(stderr, "[cvo] advice: %s\n", msg);
    }
  }
  c->gstream[0] = c->stream;
  for (int g = 0; ok && g < cvo_ctx::MAX_GROUPS; g++) {
    if (g && !pooled) ok = ok && hipStreamCreateWithFlags(&c->gstream[g], hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->ev_join[g], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 2; i++) ok = ok && hipEventCreateWithFlags(&c->ev_chk[i][g], hipEventDisableTiming) == hipSuccess;
  }
  // HIP binds a stream to a hardware queue when the stream is first USED, in the order of first use, and the four
  // sub-batch streams of a batch must sit on four different compute pipes (two of them on one pipe take turns kernel by
  // kernel: 205 ms instead of 69 ms per step for the headline batch, measured when a process's first context uploaded
  // its clouds - a pool of temporary streams - before its first solve).  So the sub-batch streams are touched here, in
  // order, before any other stream of this context exists.
  for (int g = 0; ok && g < 4; g++) {
    hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 0, c->gstream[g], 0ull);
    ok = ok && hipGetLastError() == hipSuccess;
  }
  for (int g = 0; ok && g < 4; g++) ok = ok && hipStreamSynchronize(c->gstream[g]) == hipSuccess;
  if (!pooled) ok = ok && hipStreamCreateWithFlags(&c->upload_stream, hipStreamNonBlocking) == hipSuccess;
  // k_kd_order keeps the keys of a whole cloud in LDS: up to 128 KB of dynamic shared memory
  ok = ok && hipFuncSetAttribute((const void*)k_kd_order, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)(sizeof(unsigned long long) * KD_MAX_POINTS)) == hipSuccess;
  if (!ok) {
    cvo_ctx_destroy(c);
    return CVO_E_HIP;
  }
  *out = c;
  return CVO_OK;
}

static void queue_release(cvo_batch_queue* q);

void cvo_ctx_destroy(cvo_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  // an open batch queue goes first: its streams are drained, its pinned block freed and the handle orphaned - the host
  // object stays until its owner calls cvo_batch_close, every other call on it returns CVO_E_INVALID
  if (c->queue) queue_release(c->queue);
  // every stream of the set must be idle before the workspace goes (work queued by a call that returned early on an
  // error would otherwise run against freed memory) - and a set whose streams cannot be drained is not pooled
  bool drained = true;
  for (int g = 0; g < cvo_ctx::MAX_GROUPS; g++)
    if (c->gstream[g]) drained = (hipStreamSynchronize(c->gstream[g]) == hipSuccess) && drained;
  if (c->upload_stream) drained = (hipStreamSynchronize(c->upload_stream) == hipSuccess) && drained;
  drop_graphs(c);
  free_workspace(c);
  if (c->d_kd_jobs) (void)hipFree(c->d_kd_jobs);
  for (int g = 0; g < cvo_ctx::MAX_GROUPS; g++) {
    for (int i = 0; i < 2; i++)
      if (c->ev_chk[i][g]) (void)hipEventDestroy(c->ev_chk[i][g]);
    if (c->ev_join[g]) (void)hipEventDestroy(c->ev_join[g]);
  }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_start) (void)hipEventDestroy(c->ev_start);
  if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
  {  // a complete, drained set goes back to the device's pool (see StreamSet); anything else is destroyed
    bool complete = drained && c->upload_stream != nullptr;
    for (int g = 0; g < cvo_ctx::MAX_GROUPS; g++) complete = complete && c->gstream[g] != nullptr;
    if (complete) {
      StreamSet ss;
      for (int g = 0; g < cvo_ctx::MAX_GROUPS; g++) ss.g[g] = c->gstream[g];
      ss.upload = c->upload_stream;
      std::lock_guard<std::mutex> lk(g_stream_pool_mutex);
      g_stream_pool[c->device].push_back(ss);
    } else {
      for (int g = 1; g < cvo_ctx::MAX_GROUPS; g++)
        if (c->gstream[g]) (void)hipStreamDestroy(c->gstream[g]);
      if (c->stream) (void)hipStreamDestroy(c->stream);
      if (c->upload_stream) (void)hipStreamDestroy(c->upload_stream);
    }
  }
  delete c;
}

void cvo_shutdown(void) {
  // the pooled stream sets of destroyed contexts (contexts still alive keep theirs)
  std::lock_guard<std::mutex> lk(g_stream_pool_mutex);
  for (auto& kv : g_stream_pool) {
    (void)hipSetDevice(kv.first);
    for (StreamSet& ss : kv.second) {
      for (int g = 0; g < cvo_ctx::MAX_GROUPS; g++)
        if (ss.g[g]) (void)hipStreamDestroy(ss.g[g]);
      if (ss.upload) (void)hipStreamDestroy(ss.upload);
    }
    kv.second.clear();
  }
}

int cvo_ctx_set_option(cvo_ctx* ctx, const char* name, const char* value) {
  if (!ctx || !name) return CVO_E_INVALID;
  if (std::strncmp(name, "CVO_", 4) == 0) name += 4;
  bool known = false;
  for (const char* k : kOptionNames) known = known || std::strcmp(k, name) == 0;
  if (!known) return fail(ctx, CVO_E_INVALID, std::string("cvo_ctx_set_option: unknown option ") + name);
  // (an open queue has chunks in flight on graphs that bake the switches in, and re-captures from its own copy of them)
  if (ctx->queue_open) return fail(ctx, CVO_E_INVALID, "cvo_ctx_set_option: a batch queue is open on this context (cvo_batch_close it first)");
  if (value)
    ctx->opt[name] = value;
  else
    ctx->opt.erase(name);
  drop_graphs(ctx);  // cached graphs bake some of the switches in
  return CVO_OK;
}

const char* cvo_last_error(const cvo_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
const char* cvo_ctx_advice(const cvo_ctx* ctx) { return ctx ? ctx->advice.c_str() : ""; }
void* cvo_ctx_stream(cvo_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int cvo_ctx_synchronize(cvo_ctx* ctx) {
  if (!ctx) return CVO_E_INVALID;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return CVO_OK;
}

// Spatial permutation of a cloud (sorted position -> original index): a balanced k-d ordering whose
// splits fall on multiples of 512 / 64 / 4 points, so that every aligned run of 512, 64 (a k_scan

}  // extern "C"
