// cvo_queue.hip -- the batch queue (cvo_batch_open / _submit / _poll / _close): a stream of frame pairs through a fixed number of in-flight slots.
// A SECTION of the one translation unit cvo_hip.hip (which includes the sections in dependency order and says why it is one
// unit); not compiled on its own.  Shared declarations: cvo_internal.h.
extern "C" {

// ---- batch queue (new, not in the reference): a STREAM of frame pairs through a fixed number of in-flight slots --------
// The reference's real use is a frame stream with warm starts and very different iteration counts
// (main_cvo_gpu_align_raw_image.cpp:100-170, one align() per frame).  cvo_align_batch takes a fixed set, and a sub-batch
// runs at the pace of its most demanding pair until its last pair ends.  Here a pair that finishes hands its slice of
// the workspace to the next queued pair at the next chunk boundary: the slot's result is read out, descriptor and
// initial state of the newcomer are copied in and k_update<INIT> runs for that slot, all in stream order behind the
// chunk in flight - the sub-batch's graphs never change (their kernel arguments are the slots, not the occupants).
// Results are delivered in submission order.
struct cvo_batch_queue {
  cvo_ctx* ctx = nullptr;
  cvo_params_t params{};
  cvo_align_opts_t opts{};
  BatchSetup S{};
  DevParams dp{};
  LoopCfg cfg{};
  int slots = 0, G = 1;
  bool allow_lean = true, start_nodense = false, allow_calm = true;
  LaunchGeom geom[cvo_ctx::MAX_GROUPS];
  struct Job {
    long long ticket;
    const cvo_cloud* X;
    const cvo_cloud* Y;
    float T[16];
    int max_iter;
  };
  std::deque<Job> waiting;
  struct Slot {
    long long ticket = -1;  // occupant (-1 = free)
    int start_chunk = 0;    // first chunk of the group whose status words belong to this occupant
    std::chrono::steady_clock::time_point t0;
  };
  std::vector<Slot> slot;
  struct Readout {  // a finished pair whose state is on its way to h_out[slot]
    long long ticket;
    int slot, ready_chunk;  // complete once the group's chunk `ready_chunk` has been waited for
    double seconds;
  };
  std::vector<Readout> readouts[cvo_ctx::MAX_GROUPS];
  std::map<long long, cvo_batch_result_t> done;
  long long next_ticket = 0, next_deliver = 0;
  int launched[cvo_ctx::MAX_GROUPS] = {}, inspected[cvo_ctx::MAX_GROUPS] = {}, graph_next[cvo_ctx::MAX_GROUPS] = {};
  int running[cvo_ctx::MAX_GROUPS] = {};  // occupied slots per group
  char* pinned = nullptr;                 // [slots] x (PairState out | PairDesc stage | PairState stage)
  PairState* h_out = nullptr;
  PairDesc* h_desc_stage = nullptr;
  PairState* h_state_stage = nullptr;
  unsigned long long n_chunks = 0, n_full_chunks = 0, n_refills = 0;
};

namespace {

int queue_group_of(const cvo_batch_queue* q, int p) {
  int g = 0;
  while (g + 1 < q->G && (int)((long)q->slots * (g + 1) / q->G) <= p) g++;
  return g;
}

// Places `job` into free slot p: descriptor + initial state + k_update<INIT>, in stream order on the slot's sub-batch stream.
int queue_fill(cvo_batch_queue* q, int p, const cvo_batch_queue::Job& job) {
  cvo_ctx* ctx = q->ctx;
  const int g = queue_group_of(q, p);
  fill_pair(ctx, &q->S, &q->params, &q->opts, 0, 0.f, q->slots, p, job.X, job.Y, job.T, next_call_serial(), job.max_iter);
  q->h_desc_stage[p] = ctx->h_descs[p];
  q->h_state_stage[p] = ctx->h_states[p];
  hipStream_t st = q->geom[g].stream;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_descs + p, q->h_desc_stage + p, sizeof(PairDesc), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_states + p, q->h_state_stage + p, sizeof(PairState), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_update<true>, dim3(1), dim3(64), 0, st, ctx->d_descs + p, ctx->d_params, ctx->d_status, 0);
  HIP_TRY(ctx, hipGetLastError());
  q->slot[p].ticket = job.ticket;
  q->slot[p].start_chunk = q->launched[g];
  q->slot[p].t0 = std::chrono::steady_clock::now();
  q->running[g]++;
  q->graph_next[g] = q->start_nodense ? 3 : 0;  // a newcomer moves fast: a rebuild opportunity in every iteration
  q->n_refills++;
  return CVO_OK;
}

// One step of sub-batch g: wait for its older chunk in flight (if two are) and act on what it reports - collect
// read-outs, retire finished pairs, refill their slots - then enqueue the next chunk.  block = false: returns without
// waiting when the older chunk has not finished yet.
int queue_step(cvo_batch_queue* q, int g, bool block, bool* progressed) {
  cvo_ctx* ctx = q->ctx;
  const int p0 = q->geom[g].p0, ng = q->geom[g].n_pairs;
  hipStream_t st = q->geom[g].stream;
  // ---- inspect
  const bool idle_tail = q->running[g] == 0 && q->launched[g] > q->inspected[g];  // nothing left to launch for: drain what is in flight
  if (q->launched[g] - q->inspected[g] >= 2 || idle_tail) {
    const int c = q->inspected[g];
    hipEvent_t ev = ctx->ev_chk[c & 1][g];
    if (!block) {
      const hipError_t e = hipEventQuery(ev);
      if (e == hipErrorNotReady) return CVO_OK;
      if (e != hipSuccess) return fail(ctx, CVO_E_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
    } else {
      HIP_TRY(ctx, hipEventSynchronize(ev));
    }
    q->inspected[g] = c + 1;
    *progressed = true;
    // read-outs enqueued before chunk c was launched are complete
    auto& ro = q->readouts[g];
    for (size_t k = 0; k < ro.size();) {
      if (ro[k].ready_chunk <= c) {
        const PairState& ps = q->h_out[ro[k].slot];
        cvo_batch_result_t r{};
        r.ticket = ro[k].ticket;
        std::memcpy(r.transform, ps.out_T, sizeof(float) * 16);
        r.info.iterations = ps.status ? ps.iterations : ps.k;
        r.info.ret = ps.sync_err ? CVO_E_HIP : ps.ret;  // (a block partial never arrived: cvo_wave.h; never seen in practice)
        r.info.final_ell = ps.ell;
        r.info.final_num_neighbors = ps.K;
        r.info.seconds = ro[k].seconds;
        q->done[r.ticket] = r;
        ro[k] = ro.back();
        ro.pop_back();
      } else {
        k++;
      }
    }
    // finished pairs: their state is read out behind everything enqueued so far; the slot goes to the next waiting pair
    const volatile int* hs = ctx->h_status[0] + 2 * p0;  // [status[ng] | want[ng]]
    int want = -1;
    bool dense = false;
    for (int k = 0; k < ng; k++) {
      cvo_batch_queue::Slot& sl = q->slot[p0 + k];
      if (sl.ticket < 0 || c < sl.start_chunk) {
        if (sl.ticket >= 0) want = 2;  // (placed, not yet reported: still asks for the full graph)
        continue;
      }
      if (hs[k] != 0) {
        HIP_TRY(ctx, hipMemcpyAsync(q->h_out + p0 + k, ctx->d_states + p0 + k, offsetof(PairState, sq), hipMemcpyDeviceToHost, st));
        q->readouts[g].push_back({sl.ticket, p0 + k, q->launched[g],
                                  std::chrono::duration<double>(std::chrono::steady_clock::now() - sl.t0).count()});
        sl.ticket = -1;
        q->running[g]--;
      } else {
        const int w = hs[ng + k];
        dense = dense || w == 4 || w >= 8;
        want = std::max(want, w == 4 ? 2 : (w >= 8 ? w - 9 : w));
      }
    }
    q->graph_next[g] = choose_graph(want, dense, q->allow_lean, q->start_nodense, q->allow_calm, q->cfg.lean_U2);
    // Admission.  A newcomer moves fast: its lists last an iteration or two, so its sub-batch runs the full graph (six
    // launches per iteration, three of which find nothing to do for the settled pairs) until it has calmed down.  Free
    // slots are therefore refilled in cohorts: at once while the sub-batch runs a fast graph anyway or stands empty,
    // otherwise when a quarter of its slots have come free.
    if (!q->waiting.empty() && q->running[g] < ng) {
      const int v = q->graph_next[g];
      const bool fast = v == 0 || v == 3 || v == 2 || v == 6;
      int den = 4;  // (QUEUE_ADMIT: the share of free slots - 1 / den - at which a settled sub-batch takes newcomers)
      if (const char* e = ctx_opt(ctx, "QUEUE_ADMIT")) den = std::max(1, atoi(e));
      if (fast || q->running[g] == 0 || den * (ng - q->running[g]) >= ng)
        for (int k = 0; k < ng && !q->waiting.empty(); k++)
          if (q->slot[p0 + k].ticket < 0) {
            const cvo_batch_queue::Job job = q->waiting.front();
            q->waiting.pop_front();
            const int rc = queue_fill(q, p0 + k, job);
            if (rc != CVO_OK) return rc;
          }
    }
  }
  // ---- launch
  if (q->running[g] > 0 && q->launched[g] - q->inspected[g] < 2) {
    const int v = q->graph_next[g];
    const bool fast = v == 0 || v == 3 || v == 2 || v == 6;
    const int Uc = fast ? q->cfg.U : q->cfg.U_late;
    int rc = ensure_graph(ctx, q->S, q->geom, q->G, q->cfg, g, v, Uc);
    if (rc != CVO_OK) return rc;
    HIP_TRY(ctx, hipGraphLaunch(ctx->graph_exec[g][graph_slot(q->cfg, v, Uc)], st));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_chk[q->launched[g] & 1][g], st));
    q->launched[g]++;
    q->n_chunks++;
    if (v == 0 || v == 3) q->n_full_chunks++;
    *progressed = true;
  } else if (q->running[g] == 0 && !q->readouts[g].empty() && q->launched[g] == q->inspected[g]) {
    // read-outs behind the last chunk of a group that has gone idle: an event of their own
    HIP_TRY(ctx, hipEventRecord(ctx->ev_chk[q->launched[g] & 1][g], st));
    q->launched[g]++;
    *progressed = true;
  }
  return CVO_OK;
}

int queue_pending(const cvo_batch_queue* q) { return (int)(q->next_ticket - q->next_deliver); }

}  // namespace

int cvo_batch_open(cvo_ctx* ctx, const cvo_params_t* params, int slots, int max_source_points, int max_target_points,
                   int min_source_points, const cvo_align_opts_t* opts, cvo_batch_queue** out) {
  if (!ctx || !out) return CVO_E_INVALID;
  *out = nullptr;
  if (!params || slots <= 0) return fail(ctx, CVO_E_INVALID, "cvo_batch_open: bad argument");
  if (ctx->queue_open) return fail(ctx, CVO_E_INVALID, "cvo_batch_open: this context already has an open batch queue");
  if (opts && (opts->trace || opts->override_state))
    return fail(ctx, CVO_E_UNSUPPORTED, "cvo_batch_open: traces and state overrides are per-call features of cvo_align_ex / cvo_align_batch");
  cvo_batch_queue* q = new cvo_batch_queue();
  q->ctx = ctx;
  q->params = *params;
  if (opts) q->opts.max_iterations = opts->max_iterations;
  q->slots = slots;
  const QueueDims qd{max_source_points, max_target_points, min_source_points > 0 ? min_source_points : max_source_points};
  int rc = setup_batch(ctx, params, slots, nullptr, nullptr, nullptr, &q->opts, 0, 0.f, &q->S, &q->dp, nullptr, &qd);
  if (rc != CVO_OK) {
    delete q;
    return rc;
  }
  q->G = q->S.G;
  for (int g = 0; g < q->G; g++) {
    const int p0 = (int)((long)slots * g / q->G), p1 = (int)((long)slots * (g + 1) / q->G);
    q->geom[g] = q->S.geom;
    q->geom[g].group = g;
    q->geom[g].p0 = p0;
    q->geom[g].n_pairs = p1 - p0;
    q->geom[g].arena.base = q->S.geom.arena.base + q->S.L.total * (size_t)p0;
    q->geom[g].stream = ctx->gstream[g];
  }
  // Iterations per chunk: a finished pair idles until the chunk after next (the host learns of it one chunk behind), so a
  // queue of short solves wants short chunks; a boundary costs a stream ~10 us.  16 iterations for the fast graphs,
  // twice that for the lean ones, as cvo_align_batch.
  int U = 16;
  q->cfg = LoopCfg{U, 2 * U, std::max(1, std::min(q->dp.lean_U, U)), std::max(0, std::min(q->dp.lean_U2, U)), q->S.geom.instr ? 8 : 0};
  q->allow_lean = ctx_opt(ctx, "NO_LEAN") == nullptr;
  q->start_nodense = q->allow_lean && q->S.N > 4096;
  q->allow_calm = q->dp.calm_U > 0;
  q->slot.assign((size_t)slots, cvo_batch_queue::Slot());
  const size_t per = align_up(sizeof(PairState), 256) * 2 + align_up(sizeof(PairDesc), 256);
  hipError_t e = hipHostMalloc(&q->pinned, per * (size_t)slots, hipHostMallocDefault);
  if (e != hipSuccess) {
    delete q;
    return fail(ctx, CVO_E_NOMEM, std::string("cvo_batch_open: hipHostMalloc: ") + hipGetErrorString(e));
  }
  q->h_out = (PairState*)q->pinned;
  q->h_state_stage = (PairState*)(q->pinned + align_up(sizeof(PairState), 256) * (size_t)slots);
  q->h_desc_stage = (PairDesc*)(q->pinned + align_up(sizeof(PairState), 256) * 2 * (size_t)slots);
  // the set-up copies went to group 0's stream: the other sub-batch streams start behind them
  e = hipEventRecord(ctx->ev_fork, ctx->stream);
  for (int g = 1; g < q->G && e == hipSuccess; g++) e = hipStreamWaitEvent(q->geom[g].stream, ctx->ev_fork, 0);
  if (e != hipSuccess) {
    (void)hipHostFree(q->pinned);
    delete q;
    return fail(ctx, CVO_E_HIP, std::string("cvo_batch_open: ") + hipGetErrorString(e));
  }
  ctx->queue_open = true;
  ctx->queue = q;
  *out = q;
  return CVO_OK;
}

int cvo_batch_submit(cvo_batch_queue* q, const cvo_cloud* source, const cvo_cloud* target, const float init_T[16],
                     int max_iterations, long long* ticket) {
  if (!q || !q->ctx) return CVO_E_INVALID;  // (ctx == nullptr: the context was destroyed under the queue)
  cvo_ctx* ctx = q->ctx;
  if (!source || !target || !init_T) return fail(ctx, CVO_E_INVALID, "cvo_batch_submit: null argument");
  if (source->ctx != ctx || target->ctx != ctx) return fail(ctx, CVO_E_INVALID, "cloud belongs to another context");
  if (source->n <= 0 || target->n <= 0) return fail(ctx, CVO_E_INVALID, "cvo_batch_submit: empty cloud");
  if (!(source->rmax <= 1e15f) || !(target->rmax <= 1e15f))
    return fail(ctx, CVO_E_INVALID, "cloud with non-finite or astronomically large coordinates (|p| > 1e15)");
  if (source->n > q->S.N || target->n > q->S.M || coeff_split(source->n) > q->S.geom.csplit)
    return fail(ctx, CVO_E_INVALID, "cvo_batch_submit: cloud outside the sizes the queue was opened for");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  {
    const bool nf = q->params.is_using_intensity != 0, nl = q->params.is_using_semantics != 0, ng = q->params.is_using_geometric_type != 0;
    if (nf || nl || ng) {
      bool created = false;
      int rc0 = ensure_attributes(ctx, source, nf, nl, ng, &created);
      if (rc0 == CVO_OK) rc0 = ensure_attributes(ctx, target, nf, nl, ng, &created);
      if (rc0 != CVO_OK) return rc0;
      // a zero slab is filled on the context's stream (= sub-batch 0's) and read on every sub-batch stream: wait for the
      // fill - only when this call made one (the stream carries sub-batch 0's chunks: a wait per submission serialised the
      // host with the device)
      if (created) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
  }
  cvo_batch_queue::Job job;
  job.ticket = q->next_ticket++;
  job.X = source;
  job.Y = target;
  std::memcpy(job.T, init_T, sizeof(float) * 16);
  job.max_iter = max_iterations > 0 ? std::min(max_iterations, q->dp.max_iter) : q->dp.max_iter;
  if (ticket) *ticket = job.ticket;
  // a free slot, in the sub-batch with the fewest occupants (newcomers of one sub-batch share their fast first iterations)
  int best = -1, best_run = 1 << 30;
  if (q->waiting.empty())
    for (int g = 0; g < q->G; g++) {
      if (q->running[g] >= q->geom[g].n_pairs || q->running[g] >= best_run) continue;
      for (int k = 0; k < q->geom[g].n_pairs; k++)
        if (q->slot[q->geom[g].p0 + k].ticket < 0) {
          // (its previous occupant's read-out may still be in flight: stream order protects it)
          best = q->geom[g].p0 + k;
          best_run = q->running[g];
          break;
        }
    }
  if (best >= 0) return queue_fill(q, best, job);
  q->waiting.push_back(job);
  return CVO_OK;
}

int cvo_batch_poll(cvo_batch_queue* q, int wait, int capacity, cvo_batch_result_t* results, int* n_results) {
  if (!q || !n_results || (capacity > 0 && !results)) return CVO_E_INVALID;
  *n_results = 0;
  if (!q->ctx) return CVO_E_INVALID;
  cvo_ctx* ctx = q->ctx;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  auto deliverable = [&] { return q->done.count(q->next_deliver) != 0; };
  for (;;) {
    bool progressed = false;
    // (every sub-batch is stepped without blocking first; only when none of them moved does the call wait for one)
    for (int g = 0; g < q->G; g++) {
      const int rc = queue_step(q, g, false, &progressed);
      if (rc != CVO_OK) return rc;
    }
    if (wait == 0) break;
    if (wait == 1 && (deliverable() || queue_pending(q) == 0)) break;
    if (wait >= 2 && ((int)q->done.size() == queue_pending(q) || (capacity > 0 && (int)q->done.size() >= capacity && deliverable()))) break;
    if (!progressed) {
      int gw = -1;  // the sub-batch with the most chunks in flight
      for (int g = 0; g < q->G; g++)
        if (q->launched[g] > q->inspected[g] && (gw < 0 || q->launched[g] - q->inspected[g] > q->launched[gw] - q->inspected[gw])) gw = g;
      if (gw < 0) break;  // nothing in flight and nothing to launch
      const int rc = queue_step(q, gw, true, &progressed);
      if (rc != CVO_OK) return rc;
    }
  }
  while (*n_results < capacity && deliverable()) {
    results[*n_results] = q->done[q->next_deliver];
    q->done.erase(q->next_deliver);
    q->next_deliver++;
    (*n_results)++;
  }
  return CVO_OK;
}

int cvo_batch_pending(const cvo_batch_queue* q) { return (q && q->ctx) ? queue_pending(q) : 0; }

int cvo_batch_stats(const cvo_batch_queue* q, unsigned long long* chunks, unsigned long long* full_chunks, unsigned long long* refills) {
  if (!q) return CVO_E_INVALID;
  if (chunks) *chunks = q->n_chunks;
  if (full_chunks) *full_chunks = q->n_full_chunks;
  if (refills) *refills = q->n_refills;
  return CVO_OK;
}

// The device side of a queue: streams drained, pinned block freed, the context unlocked and the handle orphaned.
static void queue_release(cvo_batch_queue* q) {
  cvo_ctx* ctx = q->ctx;
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (int g = 0; g < q->G; g++) (void)hipStreamSynchronize(q->geom[g].stream);
  if (q->pinned) (void)hipHostFree(q->pinned);
  q->pinned = nullptr;
  q->h_out = nullptr;
  q->h_desc_stage = nullptr;
  q->h_state_stage = nullptr;
  ctx->queue_open = false;
  ctx->queue = nullptr;
  q->ctx = nullptr;
}

void cvo_batch_close(cvo_batch_queue* q) {
  if (!q) return;
  queue_release(q);  // (a no-op when cvo_ctx_destroy already ran it)
  delete q;
}

}  // extern "C"
