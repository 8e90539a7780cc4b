// cvo_upload.hip -- resident clouds: host-side spatial ordering (the twin of k_kd_order), staging, one allocation + one copy per cloud, cvo_cloud_upload / _aos192 / _many, cvo_cloud_transformed, zero slabs for attributes a cloud was uploaded without.
// A SECTION of the one translation unit cvo_hip.hip (which includes the sections in dependency order and says why it is one
// unit); not compiled on its own.  Shared declarations: cvo_internal.h.
namespace {

// *created (optional) is set when this call allocated the slab: its zero fill is in flight on ctx->stream.
int ensure_attributes(cvo_ctx* ctx, const cvo_cloud* c, bool need_feat, bool need_label, bool need_geo, bool* created = nullptr) {
  if ((!need_feat || c->feat) && (!need_label || c->label) && (!need_geo || c->geo)) return CVO_OK;
  cvo_cloud* m = const_cast<cvo_cloud*>(c);
  const size_t nn = (size_t)std::max(c->n, 1);
  const size_t o_feat = 0, o_label = align_up(sizeof(float4) * 2 * nn, 256), o_geo = o_label + align_up(sizeof(float4) * 5 * nn, 256);
  const size_t bytes = o_geo + align_up(sizeof(float2) * nn, 256);
  if (!m->zero_slab) {
    HIP_TRY(ctx, hipSetDevice(c->device));
    hipError_t e = hipMalloc(&m->zero_slab, bytes);
    if (e != hipSuccess) return fail(ctx, CVO_E_NOMEM, std::string("cloud hipMalloc: ") + hipGetErrorString(e));
    e = hipMemsetAsync(m->zero_slab, 0, bytes, ctx->stream);
    if (e != hipSuccess) {  // never hand the kernels an allocated-but-not-zeroed "zero" slab on a later call
      (void)hipFree(m->zero_slab);
      m->zero_slab = nullptr;
      return fail(ctx, CVO_E_HIP, std::string("cloud hipMemsetAsync: ") + hipGetErrorString(e));
    }
    if (created) *created = true;
  }
  if (!m->feat) m->feat = (float4*)(m->zero_slab + o_feat);
  if (!m->label) m->label = (float4*)(m->zero_slab + o_label);
  if (!m->geo) m->geo = (float2*)(m->zero_slab + o_geo);
  return CVO_OK;
}

}  // namespace

extern "C" {

// k_scan's tile culling depends on it, never a result (CVO_NO_SORT=1 keeps the identity order).
struct KdPoint {
  float c[3];
  int i;
};
// vext != nullptr: the split axis comes from the root box's extents, halved once per split along that axis - one axis
// per level, what k_kd_order does on the device (option ORDER=virtual: the host twin of the device ordering)
static void kd_split(KdPoint* pts, int lo, int hi, const float* vext = nullptr) {
  const int n = hi - lo;
  if (n <= 4) return;
  const int unit = n > 512 ? 512 : (n > 64 ? 64 : 4);
  int left = ((n / 2 + unit - 1) / unit) * unit;
  if (left >= n) left -= unit;
  if (left <= 0) return;
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (!vext)
    for (int k = lo; k < hi; k++)
      for (int c = 0; c < 3; c++) {
        const float v = pts[k].c[c];
        mn[c] = std::min(mn[c], v);
        mx[c] = std::max(mx[c], v);
      }
  else
    for (int c = 0; c < 3; c++) {
      mn[c] = 0.f;
      mx[c] = vext[c];
    }
  int axis = 0;
  for (int c = 1; c < 3; c++)
    if (mx[c] - mn[c] > mx[axis] - mn[axis]) axis = c;
  float vnext[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
  vnext[axis] *= 0.5f;
  // the records themselves are permuted (no index indirection in the comparator: ~4x faster at 10k points)
  std::nth_element(pts + lo, pts + lo + left, pts + hi, [axis](const KdPoint& a, const KdPoint& b) {
    return a.c[axis] < b.c[axis] || (a.c[axis] == b.c[axis] && a.i < b.i);
  });
  kd_split(pts, lo, lo + left, vext ? vnext : nullptr);
  kd_split(pts, lo + left, hi, vext ? vnext : nullptr);
}

static void spatial_order(const float* x4, int n, std::vector<int>& order, bool no_sort, bool level_axes) {
  order.resize(n);
  for (int i = 0; i < n; i++) order[i] = i;
  if (n < 8 || no_sort) return;

  std::vector<KdPoint> pts((size_t)n);
  for (int i = 0; i < n; i++) {
    for (int c = 0; c < 3; c++) {
      const float v = x4[4 * (size_t)i + c];
      if (!std::isfinite(v)) return;  // keep the identity order for odd inputs
      pts[i].c[c] = v;
    }
    pts[i].i = i;
  }
  if (level_axes) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < n; i++)
      for (int c = 0; c < 3; c++) {
        mn[c] = std::min(mn[c], pts[i].c[c]);
        mx[c] = std::max(mx[c], pts[i].c[c]);
      }
    const float ext[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
    kd_split(pts.data(), 0, n, ext);
  } else {
    kd_split(pts.data(), 0, n);
  }
  for (int r = 0; r < n; r++) order[r] = pts[r].i;
}

// One cloud: spatial order on the calling thread, ONE device allocation and ONE host-to-device copy (a hipMalloc / a
// synchronous copy cost ~100 us each) of exactly the arrays the caller supplied.  xyz: n x 3 (stride3) or n x 4
// records of `stride` bytes; feat / label / geo may be NULL.  Self-contained and thread-safe: it touches the context
// only to read its device ordinal, and copies on the stream it is given.
struct HostCloud {
  int n;
  const char* xyz;   size_t xyz_stride;    // 3 floats at xyz + i * xyz_stride
  const char* feat;  size_t feat_stride;   // FD floats, or NULL
  const char* label; size_t label_stride;  // NC floats, or NULL
  const char* geo;   size_t geo_stride;    // 2 floats, or NULL
};

// A cloud whose spatial ordering runs on the device (k_kd_order): staged and copied, not yet ordered.  The staging
// buffer lives until the caller has synchronised the stream the copy was enqueued on.
struct StagedCloud {
  cvo_cloud* c = nullptr;
  KdJob job{};          // job.n == 0: ordered on the host, nothing left to do
  std::vector<char> stage;
};

static int upload_host_cloud(cvo_ctx* ctx, const HostCloud& h, hipStream_t stream, StagedCloud* sc) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int n = h.n;
  cvo_cloud* c = new cvo_cloud();
  c->ctx = ctx;
  c->device = ctx->device;
  c->n = n;
  const size_t nn = (size_t)std::max(n, 1);
  // Where the ordering runs: on the device for clouds k_kd_order holds in LDS (the host then only stages, allocates and
  // copies: ~0.1 ms of CPU per 10k cloud instead of 1.2), on this thread otherwise (tiny, huge or non-finite clouds,
  // CVO_NO_SORT, CVO_ORDER=host).
  bool finite = true;
  for (int i = 0; i < n && finite; i++) {
    const float* p = reinterpret_cast<const float*>(h.xyz + (size_t)i * h.xyz_stride);
    finite = std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]);
  }
  const char* ord = ctx_opt(ctx, "ORDER");
  const bool device_order = n >= 8 && n <= KD_MAX_POINTS && finite && ctx_opt(ctx, "NO_SORT") == nullptr &&
                            !(ord && (std::strcmp(ord, "host") == 0 || std::strcmp(ord, "virtual") == 0));
  // one-hot class rows?  (exactly: the fast path replaces arithmetic on the rows by two constants)
  std::vector<int> lid_host;
  if (h.label && n > 0 && ctx_opt(ctx, "NO_ONEHOT") == nullptr) {
    lid_host.resize((size_t)n);
    bool onehot = true;
    for (int i = 0; i < n && onehot; i++) {
      const float* l = reinterpret_cast<const float*>(h.label + (size_t)i * h.label_stride);
      int hot = -1, ones = 0;
      for (int c = 0; c < NC; c++) {
        if (l[c] == 1.0f) {
          hot = c;
          ones++;
        } else if (!(l[c] == 0.0f)) {
          ones = 2;  // (neither 0 nor 1: a soft distribution)
        }
      }
      onehot = ones == 1;
      lid_host[i] = hot;
    }
    if (!onehot) lid_host.clear();
  }
  const bool has_lid = !lid_host.empty();
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  // (device ordering: the caller's arrays go up in ORIGINAL order - x4 stays, the raw attribute arrays are scratch - and
  // the kernel writes the spatially ordered ones; host ordering: everything is staged in its final form)
  const size_t o_x4 = take(sizeof(float4) * nn);
  const size_t o_rawf = device_order && h.feat ? take(sizeof(float) * FD * nn) : 0, o_rawl = device_order && h.label ? take(sizeof(float) * NC * nn) : 0,
               o_rawg = device_order && h.geo ? take(sizeof(float) * 2 * nn) : 0,
               o_rawlid = device_order && has_lid ? take(sizeof(int) * nn) : 0;
  const size_t up_bytes_device = off;
  const size_t o_xs4 = take(sizeof(float4) * nn), o_order = take(sizeof(int) * nn), o_inv = take(sizeof(int) * nn);
  const size_t o_feat = h.feat ? take(sizeof(float4) * 2 * nn) : 0, o_label = h.label ? take(sizeof(float4) * 5 * nn) : 0,
               o_geo = h.geo ? take(sizeof(float2) * nn) : 0, o_lid = has_lid ? take(sizeof(int) * nn) : 0;
  int NP = KD_THREADS;
  while (NP < n) NP *= 2;
  const size_t o_segpos = device_order ? take(sizeof(unsigned short) * (size_t)NP) : 0,
               o_seglo = device_order ? take(sizeof(unsigned short) * 2 * KD_MAX_SEGS) : 0;
  const size_t up_bytes = device_order ? up_bytes_device : off;
  std::vector<char>& stage = sc->stage;
  stage.assign(up_bytes, 0);  // (pageable: kept alive by the caller until the stream has been synchronised)
  float* x4 = reinterpret_cast<float*>(&stage[o_x4]);
  double sx = 0, sy = 0, sz = 0, r2max = 0;
  for (int i = 0; i < n; i++) {
    const float* p = reinterpret_cast<const float*>(h.xyz + (size_t)i * h.xyz_stride);
    x4[4 * (size_t)i] = p[0];
    x4[4 * (size_t)i + 1] = p[1];
    x4[4 * (size_t)i + 2] = p[2];
    const double px = p[0], py = p[1], pz = p[2];
    sx += px;
    sy += py;
    sz += pz;
    const double r2 = px * px + py * py + pz * pz;
    if (r2max == r2max && !(r2 <= r2max)) r2max = r2;  // a NaN sticks (an unbounded cloud is refused by the solvers)
  }
  c->rmax = (float)(std::sqrt(r2max) * 1.000001);
  if (n > 0) {
    c->cx = (float)(sx / n);
    c->cy = (float)(sy / n);
    c->cz = (float)(sz / n);
  }
  if (!std::isfinite(c->cx) || !std::isfinite(c->cy) || !std::isfinite(c->cz)) c->cx = c->cy = c->cz = 0.f;
  if (device_order) {
    if (h.feat) {
      float* f = reinterpret_cast<float*>(&stage[o_rawf]);
      for (int i = 0; i < n; i++) std::memcpy(&f[FD * (size_t)i], h.feat + (size_t)i * h.feat_stride, sizeof(float) * FD);
    }
    if (h.label) {
      float* l = reinterpret_cast<float*>(&stage[o_rawl]);
      for (int i = 0; i < n; i++) std::memcpy(&l[NC * (size_t)i], h.label + (size_t)i * h.label_stride, sizeof(float) * NC);
    }
    if (h.geo) {
      float* g = reinterpret_cast<float*>(&stage[o_rawg]);
      for (int i = 0; i < n; i++) std::memcpy(&g[2 * (size_t)i], h.geo + (size_t)i * h.geo_stride, sizeof(float) * 2);
    }
    if (has_lid) std::memcpy(&stage[o_rawlid], lid_host.data(), sizeof(int) * (size_t)n);
  } else {
    std::vector<int> order;
    spatial_order(x4, n, order, ctx_opt(ctx, "NO_SORT") != nullptr, ord && std::strcmp(ord, "virtual") == 0);
    // colour, class distributions and geometric types are kept in SPATIAL order only (position r holds the attributes of
    // point order[r]): the kernels index them by sorted position, like the coordinates they gather per candidate
    if (h.feat) {
      float* f8 = reinterpret_cast<float*>(&stage[o_feat]);
      for (int r = 0; r < n; r++) std::memcpy(&f8[FD_PAD * (size_t)r], h.feat + (size_t)order[r] * h.feat_stride, sizeof(float) * FD);
    }
    if (h.label) {
      float* l20 = reinterpret_cast<float*>(&stage[o_label]);
      for (int r = 0; r < n; r++) std::memcpy(&l20[NC_PAD * (size_t)r], h.label + (size_t)order[r] * h.label_stride, sizeof(float) * NC);
    }
    if (h.geo) {
      float* g2 = reinterpret_cast<float*>(&stage[o_geo]);
      for (int r = 0; r < n; r++) std::memcpy(&g2[2 * (size_t)r], h.geo + (size_t)order[r] * h.geo_stride, sizeof(float) * 2);
    }
    if (has_lid) {
      int* li = reinterpret_cast<int*>(&stage[o_lid]);
      for (int r = 0; r < n; r++) li[r] = lid_host[(size_t)order[r]];
    }
    float* xs = reinterpret_cast<float*>(&stage[o_xs4]);
    for (int r = 0; r < n; r++) std::memcpy(&xs[4 * (size_t)r], &x4[4 * (size_t)order[r]], 16);
    if (n > 0) std::memcpy(&stage[o_order], order.data(), sizeof(int) * (size_t)n);
    {
      int* inv = reinterpret_cast<int*>(&stage[o_inv]);
      for (int r = 0; r < n; r++) inv[order[r]] = r;
    }
    c->h_order = std::move(order);
  }
  hipError_t e = hipMalloc(&c->slab, off);
  c->slab_bytes = off;
  if (e != hipSuccess) {
    cvo_cloud_free(c);
    return fail(ctx, CVO_E_NOMEM, std::string("cloud hipMalloc: ") + hipGetErrorString(e));
  }
  c->x4 = (float4*)(c->slab + o_x4);
  c->xs4 = (float4*)(c->slab + o_xs4);
  c->order = (int*)(c->slab + o_order);
  c->inv = (int*)(c->slab + o_inv);
  c->feat = h.feat ? (float4*)(c->slab + o_feat) : nullptr;
  c->label = h.label ? (float4*)(c->slab + o_label) : nullptr;
  c->geo = h.geo ? (float2*)(c->slab + o_geo) : nullptr;
  c->lid = has_lid ? (int*)(c->slab + o_lid) : nullptr;
  if (n > 0) {
    e = hipMemcpyAsync(c->slab, stage.data(), up_bytes, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) {
      cvo_cloud_free(c);
      return fail(ctx, CVO_E_HIP, std::string("cloud upload: ") + hipGetErrorString(e));
    }
  }
  sc->c = c;
  sc->job = KdJob{};
  if (device_order) {
    KdJob& J = sc->job;
    J.n = n;
    J.NP = NP;
    J.x4 = c->x4;
    J.seg_of_pos = (unsigned short*)(c->slab + o_segpos);
    J.seg_lo = (unsigned short*)(c->slab + o_seglo);
    J.order = c->order;
    J.inv = c->inv;
    J.xs4 = c->xs4;
    J.raw_feat = h.feat ? (const float*)(c->slab + o_rawf) : nullptr;
    J.feat = c->feat;
    J.raw_label = h.label ? (const float*)(c->slab + o_rawl) : nullptr;
    J.label = c->label;
    J.raw_geo = h.geo ? (const float*)(c->slab + o_rawg) : nullptr;
    J.geo = c->geo;
    J.raw_lid = has_lid ? (const int*)(c->slab + o_rawlid) : nullptr;
    J.lid = c->lid;
    c->h_order.assign((size_t)n, 0);
  }
  return CVO_OK;
}

// Second half of an upload: the copies of `clouds` have been enqueued (and, for upload_many, completed) - order the
// clouds that asked for it with ONE launch of k_kd_order (a block per cloud) on the context's upload stream, bring the
// permutations back (exports map rows through them), synchronise.  On error every cloud of the list is released.
static int finish_uploads(cvo_ctx* ctx, std::vector<StagedCloud>& clouds) {
  std::vector<KdJob> jobs;
  int np_max = 0;
  for (auto& sc : clouds)
    if (sc.c && sc.job.n > 0) {
      jobs.push_back(sc.job);
      np_max = std::max(np_max, sc.job.NP);
    }
  hipError_t e = hipSuccess;
  std::lock_guard<std::mutex> lk(ctx->kd_mutex);
  if (!jobs.empty()) {
    if ((int)jobs.size() > ctx->kd_jobs_cap) {
      if (ctx->d_kd_jobs) (void)hipFree(ctx->d_kd_jobs);
      ctx->d_kd_jobs = nullptr;
      ctx->kd_jobs_cap = 0;
      e = hipMalloc(&ctx->d_kd_jobs, sizeof(KdJob) * jobs.size());
      if (e == hipSuccess) ctx->kd_jobs_cap = (int)jobs.size();
    }
    if (e == hipSuccess)
      e = hipMemcpyAsync(ctx->d_kd_jobs, jobs.data(), sizeof(KdJob) * jobs.size(), hipMemcpyHostToDevice, ctx->upload_stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_kd_order, dim3((unsigned)jobs.size()), dim3(KD_THREADS), sizeof(unsigned long long) * (size_t)np_max,
                         ctx->upload_stream, (const KdJob*)ctx->d_kd_jobs);
      e = hipGetLastError();
    }
    for (auto& sc : clouds)
      if (e == hipSuccess && sc.c && sc.job.n > 0)
        e = hipMemcpyAsync(sc.c->h_order.data(), sc.c->order, sizeof(int) * (size_t)sc.job.n, hipMemcpyDeviceToHost, ctx->upload_stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->upload_stream);
  if (e != hipSuccess) {
    for (auto& sc : clouds) {
      if (sc.c) cvo_cloud_free(sc.c);
      sc.c = nullptr;
    }
    return fail(ctx, CVO_E_HIP, std::string("cloud upload (ordering): ") + hipGetErrorString(e));
  }
  return CVO_OK;
}

// One cloud on the context's upload stream (cvo_cloud_upload, cvo_cloud_upload_aos192).
static int upload_one(cvo_ctx* ctx, const HostCloud& h, cvo_cloud** out) {
  std::lock_guard<std::mutex> lk(ctx->upload_mutex);
  std::vector<StagedCloud> one(1);
  int rc = upload_host_cloud(ctx, h, ctx->upload_stream, &one[0]);
  if (rc != CVO_OK) return rc;
  rc = finish_uploads(ctx, one);
  if (rc != CVO_OK) return rc;
  *out = one[0].c;
  return CVO_OK;
}

int cvo_cloud_upload(cvo_ctx* ctx, int n, const float* xyz, const float* feat, const float* label,
                     const float* geotype, cvo_cloud** out) {
  if (!ctx || !out || n < 0 || (n > 0 && !xyz)) return fail(ctx, CVO_E_INVALID, "cvo_cloud_upload: bad argument");
  const HostCloud h{n, (const char*)xyz, 12, (const char*)feat, sizeof(float) * FD, (const char*)label, sizeof(float) * NC,
                    (const char*)geotype, 8};
  return upload_one(ctx, h, out);
}

// n_clouds clouds from a pool of host threads (each cloud: spatial ordering on its thread, one allocation, one copy on
// that thread's own stream).  Arrays of per-cloud pointers; feat / label / geotype (the arrays or single entries) may be
// NULL.  On error every cloud of the call is released.
static int upload_many_impl(cvo_ctx* ctx, int n_clouds, const int* n, const float* const* xyz, const float* const* feat,
                            const float* const* label, const float* const* geotype, int threads, cvo_cloud** out) {
  if (!ctx || !out || n_clouds < 0 || (n_clouds > 0 && (!n || !xyz)))
    return fail(ctx, CVO_E_INVALID, "cvo_cloud_upload_many: bad argument");
  for (int q = 0; q < n_clouds; q++) {
    out[q] = nullptr;
    if (n[q] < 0 || (n[q] > 0 && !xyz[q])) return fail(ctx, CVO_E_INVALID, "cvo_cloud_upload_many: bad cloud");
  }
  if (n_clouds == 0) return CVO_OK;
  // (with the ordering on the device a cloud costs its thread ~0.08 ms - staging, one hipMalloc, one copy - and the
  // allocator serialises: 128 clouds take 10.0 / 8.0 / 7.3 / 8.2 ms of wall time with 1 / 2 / 4 / 16 threads)
  const int T = std::max(1, std::min(std::min(threads > 0 ? threads : 4, n_clouds), 64));
  std::vector<int> rcs(T, CVO_OK);
  std::vector<std::string> errs(T);
  std::atomic<int> next(0);
  std::vector<StagedCloud> staged((size_t)n_clouds);
  auto work_body = [&](int t) {
    // Every thread copies on the context's ONE upload stream (enqueueing from several threads is legal; a pageable
    // source makes each copy synchronous for its thread anyway).  No temporary streams: HIP deals streams onto hardware
    // queues in creation order, and streams created between two contexts used to push a later context's sub-batch
    // streams onto shared queues (3x slower batches, scripts/upload_probe.py).
    hipStream_t s = ctx->upload_stream;
    if (hipSetDevice(ctx->device) != hipSuccess) {
      rcs[t] = CVO_E_HIP;
      errs[t] = "cvo_cloud_upload_many: hipSetDevice failed";
      return;
    }
    cvo_ctx local;  // error text of this thread (the shared context's string is not thread-safe)
    local.device = ctx->device;
    local.opt = ctx->opt;
    for (;;) {
      const int q = next.fetch_add(1);
      if (q >= n_clouds || rcs[t] != CVO_OK) break;
      const HostCloud h{n[q], (const char*)xyz[q], 12, (const char*)(feat ? feat[q] : nullptr), sizeof(float) * FD,
                        (const char*)(label ? label[q] : nullptr), sizeof(float) * NC,
                        (const char*)(geotype ? geotype[q] : nullptr), 8};
      const int rc = upload_host_cloud(&local, h, s, &staged[q]);
      if (rc != CVO_OK) {
        rcs[t] = rc;
        errs[t] = local.err;
        break;
      }
      staged[q].c->ctx = ctx;
      out[q] = staged[q].c;
    }
    // (the ordering kernel of the call is launched on the same stream: it runs after every copy)
  };
  auto work = [&](int t) {  // (bad_alloc of a staging buffer etc. must not leave a worker or cross the C ABI)
    try {
      work_body(t);
    } catch (const std::exception& e) {
      rcs[t] = CVO_E_NOMEM;
      errs[t] = std::string("cvo_cloud_upload_many: ") + e.what();
    } catch (...) {
      rcs[t] = CVO_E_NOMEM;
      errs[t] = "cvo_cloud_upload_many: unknown exception";
    }
  };
  std::vector<std::thread> pool;
  try {
    for (int t = 1; t < T; t++) pool.emplace_back(work, t);
  } catch (const std::exception&) {  // thread limit: the threads already started share the work with this one
  }
  work(0);
  for (auto& th : pool) th.join();
  for (int t = 0; t < T; t++)
    if (rcs[t] != CVO_OK) {
      for (int q = 0; q < n_clouds; q++) {
        if (out[q]) cvo_cloud_free(out[q]);
        out[q] = nullptr;
      }
      return fail(ctx, rcs[t], errs[t]);
    }
  const int rc = finish_uploads(ctx, staged);  // the spatial ordering of all clouds: one kernel launch
  if (rc != CVO_OK)
    for (int q = 0; q < n_clouds; q++) out[q] = nullptr;
  return rc;
}

int cvo_cloud_upload_many(cvo_ctx* ctx, int n_clouds, const int* n, const float* const* xyz, const float* const* feat,
                          const float* const* label, const float* const* geotype, int threads, cvo_cloud** out) {
  try {
    return upload_many_impl(ctx, n_clouds, n, xyz, feat, label, geotype, threads, out);
  } catch (const std::exception& e) {  // (allocation of the pool's bookkeeping itself)
    if (out)
      for (int q = 0; q < n_clouds; q++) {
        if (out[q]) cvo_cloud_free(out[q]);
        out[q] = nullptr;
      }
    return fail(ctx, CVO_E_NOMEM, std::string("cvo_cloud_upload_many: ") + e.what());
  }
}

int cvo_cloud_upload_aos192(cvo_ctx* ctx, int n, const void* pts, cvo_cloud** out) {
  if (!ctx || !out || n < 0 || (n > 0 && !pts)) return fail(ctx, CVO_E_INVALID, "cvo_cloud_upload_aos192: bad argument");
  // PointSegmentedDistribution<5,19> byte offsets: xyz@0, features@20,
  // label_distribution@44, geometric_type@120, sizeof = 192: read in place, record by record.
  const char* b = (const char*)pts;
  const HostCloud h{n, b, 192, b + 20, 192, b + 44, 192, b + 120, 192};
  return upload_one(ctx, h, out);
}

// ---- multi-frame edge kernel ---------------------------------------------------------
int cvo_cloud_transformed(cvo_ctx* ctx, const cvo_cloud* in, const float pose12[12], cvo_cloud** out) {
  if (!ctx || !in || !pose12 || !out) return fail(ctx, CVO_E_INVALID, "cvo_cloud_transformed: bad argument");
  if (in->ctx != ctx) return fail(ctx, CVO_E_INVALID, "cloud belongs to another context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  cvo_cloud* c = new cvo_cloud();
  c->ctx = ctx;
  c->device = ctx->device;
  c->n = in->n;
  c->h_order = in->h_order;
  c->slab_bytes = in->slab_bytes;
  hipError_t e = hipMalloc(&c->slab, std::max<size_t>(in->slab_bytes, 256));
  if (e != hipSuccess) {
    delete c;
    return fail(ctx, CVO_E_NOMEM, std::string("cloud hipMalloc: ") + hipGetErrorString(e));
  }
  // same slab layout: features, labels, geometric types and the spatial order are copied, coordinates rewritten
  // (attributes the input was uploaded without - NULL or pointing into its zero slab - stay absent in the copy)
  auto rebase = [&](const void* p) -> char* {
    const char* q = (const char*)p;
    return (q && q >= in->slab && q < in->slab + in->slab_bytes) ? c->slab + (q - in->slab) : nullptr;
  };
  c->x4 = (float4*)rebase(in->x4);
  c->xs4 = (float4*)rebase(in->xs4);
  c->feat = (float4*)rebase(in->feat);
  c->label = (float4*)rebase(in->label);
  c->geo = (float2*)rebase(in->geo);
  c->lid = (int*)rebase(in->lid);
  c->order = (int*)rebase(in->order);
  c->inv = (int*)rebase(in->inv);
  Pose12 P;
  for (int q = 0; q < 12; q++) P.T[q] = pose12[q];
  if (in->n > 0) {
    e = hipMemcpyAsync(c->slab, in->slab, in->slab_bytes, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_transform_pose, dim3((in->n + 255) / 256), dim3(256), 0, ctx->stream, in->n, P, in->x4, in->xs4,
                         c->x4, c->xs4);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      cvo_cloud_free(c);
      return fail(ctx, CVO_E_HIP, std::string("cvo_cloud_transformed: ") + hipGetErrorString(e));
    }
  }
  // cull centre and motion bound of the moved cloud (neither influences a result)
  const float* T = pose12;
  c->cx = T[0] * in->cx + T[1] * in->cy + T[2] * in->cz + T[3];
  c->cy = T[4] * in->cx + T[5] * in->cy + T[6] * in->cz + T[7];
  c->cz = T[8] * in->cx + T[9] * in->cy + T[10] * in->cz + T[11];
  if (!std::isfinite(c->cx) || !std::isfinite(c->cy) || !std::isfinite(c->cz)) c->cx = c->cy = c->cz = 0.f;
  double fro = 0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) fro += (double)T[4 * i + j] * T[4 * i + j];
  c->rmax = (float)((std::sqrt(fro) * in->rmax + std::sqrt((double)T[3] * T[3] + (double)T[7] * T[7] + (double)T[11] * T[11])) * 1.000001);
  *out = c;
  return CVO_OK;
}

int cvo_cloud_size(const cvo_cloud* c) { return c ? c->n : 0; }

void cvo_cloud_free(cvo_cloud* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->slab) (void)hipFree(c->slab);
  if (c->zero_slab) (void)hipFree(c->zero_slab);
  if (c->tile4) (void)hipFree(c->tile4);
  delete c;
}

}  // extern "C"
