// cvo_hip.hip -- host side of the C-ABI declared in include/cvo_hip.h: contexts, HBM-resident
// clouds, workspace layout, and the enqueue-only optimiser loop (hipGraph replays of
// [k_scan, k_assoc, k_coeff, k_update, k_prep] with a device-side status word; no host round trip per
// iteration, unlike the ~15 blocking syncs per iteration of CvoGPU.cu:1387-1533).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "cvo_kernels.h"

using namespace cvo_dev;

#define CVO_VERSION_STRING "unified_cvo_amd 0.1 (gfx950)"

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

#ifndef CVO_COEFF_DENSE_MULTI_FROM
#define CVO_COEFF_DENSE_MULTI_FROM 8  // pairs per launch from which k_coeff_dense takes eight rows per wave
#endif

struct cvo_cloud {
  cvo_ctx* ctx = nullptr;  // identity check only: never dereferenced after upload (the context may be gone)
  int device = 0;
  int n = 0;
  char* slab = nullptr;     // the one device allocation behind the pointers below
  size_t slab_bytes = 0;
  float4* x4 = nullptr;
  float4* xs4 = nullptr;    // x4 permuted into the spatial order
  float4* feat = nullptr;   // 2 float4 per point      } in SPATIAL order (position r = point order[r]): the kernels
  float4* label = nullptr;  // 5 float4 per point      } index them by sorted position, like the coordinates they
  float2* geo = nullptr;    //                         } gather per candidate
  int* lid = nullptr;       // class id per point, spatial order: only when EVERY label row is an exact one-hot (a single
                            // 1.0f, the rest 0.0f) - the semantic kernel then needs 4 bytes per candidate, not 80
  // Attributes the caller did not supply are zeros (what the reference leaves in the default-constructed CvoPoint).
  // They are not uploaded: a zeroed slab is allocated the first time a call needs them (colour / semantic /
  // geometric-type kernels on a cloud without those arrays), see ensure_attributes.
  mutable char* zero_slab = nullptr;
  // bounding spheres of the 64-point tiles of xs4 (k_tile_spheres), made the first time k_overlap reads this cloud
  mutable float4* tile4 = nullptr;
  int* order = nullptr;        // spatial (k-d) order: sorted position -> original index
  int* inv = nullptr;          // its inverse: original index -> sorted position
  std::vector<int> h_order;  // host copy (the ELL is stored by sorted row; exports map it back)
  float cx = 0, cy = 0, cz = 0;  // centroid (used only as the cull centre)
  float rmax = 0;                // largest |p| (bounds the motion of any point under a pose change)
};

namespace {

struct PairLayout {  // byte offsets of one pair's workspace inside the arena
  size_t ycull, xcull, gbox, cellbox, sbox, masks, rowbits, row_cnt, tile_count, ovf_rows, ovf_bits, gate, gate_flow, dense_off, dense_rel, ovf_wsum, word_base, done, cand_cnt, rowperm, iorig, long_j, long_stamp, xp4, ip, cand_j, rowres, rowcoef, ell, ell_j, nnz_row, flow_part, cnt_part,
      coef_part, trace, total;
};

static const char* const kGraphNames[8] = {"full", "lean", "short", "full-nodense", "calm", "lean+dense", "short+dense", "calm+dense"};

struct GraphKey {
  int n_pairs = 0, p0 = 0, T = 0, gx = 0, gy = 0, nba = 0, nbc = 0, npb = 0, idx16 = 0, general = 0, U = 0, flags = 0;
  const void* arena = nullptr;  // kernel arguments of the row-block kernels (ArenaArg)
  unsigned stride256 = 0;
  int Npad = 0;
  bool operator==(const GraphKey& o) const {
    return n_pairs == o.n_pairs && p0 == o.p0 && T == o.T && gx == o.gx && gy == o.gy && nba == o.nba &&
           nbc == o.nbc && npb == o.npb && idx16 == o.idx16 && general == o.general && U == o.U && flags == o.flags && arena == o.arena &&
           stride256 == o.stride256 && Npad == o.Npad;
  }
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

// Tuning / diagnostic switches of a context (none changes a result).  Read from the environment ONCE, when the context
// is created (CVO_<NAME>), and settable afterwards with cvo_ctx_set_option: no library call reads the process
// environment while it runs.
static const char* const kOptionNames[] = {
    "SKIN", "SKIN_BLEND", "SKIN_MIN", "SKIN_MAX", "LEAN_SKIN", "SHRINK", "SHRINK_ALIGN", "LEAN_U", "LEAN_U2", "NO_LEAN",
    "NO_DENSE_REGIME", "STREAMS", "SCAN_T", "SCAN_GROUPS", "SCAN_DEBUG", "NO_SORT", "FIXED_CHUNKS", "KEEP_COLUMNS", "VERBOSE",
    "KERNEL_CLOCK", "PHASE_TICKS", "VERIFY_LISTS", "IP_CHAIN", "DEBUG_NO_MOTION_BOUND", "COEFF_NO_UPDATE", "ORDER", "NO_NODENSE", "CALM_U", "HORIZON_MARGIN", "NO_LONG_LISTS", "FIRST_U", "FIRST_CHUNKS", "ROW_MAX", "ROW_MAX_BUSY", "NO_ONEHOT", "QUEUE_U", "QUEUE_ADMIT"};

struct cvo_ctx {
  int device = 0;
  std::map<std::string, std::string> opt;  // see kOptionNames
  std::mutex upload_mutex;                 // cvo_cloud_upload / _aos192 share upload_stream and the error string
  std::mutex kd_mutex;                     // the ordering launches of concurrent uploads share upload_stream and d_kd_jobs
  KdJob* d_kd_jobs = nullptr;              // job descriptors of the running k_kd_order launch
  int kd_jobs_cap = 0;
  hipStream_t stream = nullptr;
  hipStream_t upload_stream = nullptr;  // cvo_cloud_upload copies here (never waits for, nor delays, the solver's streams)
  std::string err;
  std::string advice;  // performance-relevant observations about the process set-up (cvo_ctx_advice), "" = none
  // workspace
  char* arena = nullptr;
  size_t arena_bytes = 0;
  PairDesc* d_descs = nullptr;
  PairState* d_states = nullptr;
  int* d_status = nullptr;
  DevParams* d_params = nullptr;
  // descriptors, states, status words and the parameter block live in ONE device allocation with a pinned staging copy
  // of the same layout: a call uploads its control state with one copy (four copies cost every cvo_align ~10 us and
  // an inner product a third of its time)
  char* d_ctl = nullptr;
  char* h_ctl = nullptr;
  size_t ctl_bytes = 0, ctl_off_status = 0, ctl_off_descs = 0, ctl_off_states = 0;
  int cap_pairs = 0;
  std::vector<PairDesc> h_descs;
  std::vector<PairState> h_states;
  // k_overlap (one-launch inner products): row-tile partials + gate words of up to three jobs (device), results (pinned)
  char* d_ov = nullptr;
  int ov_tiles_cap = 0;
  char* h_ov = nullptr;
  int* h_status[2] = {nullptr, nullptr};  // pinned; [0]: the live host mirror of the status / want words the device writes
                                          // (PairDesc::status_host / want_host), [1]: unused slot kept for the layout
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  // A batch is split into up to MAX_GROUPS sub-batches, each enqueued on its own stream: the pairs are
  // independent, so one group's latency-bound kernels (k_update: one wave per pair) and launch tails
  // overlap the other groups' wide kernels.  Group 0 runs on `stream`.
  static constexpr int MAX_GROUPS = 8;
  hipStream_t gstream[MAX_GROUPS] = {};
  hipEvent_t ev_chk[2][MAX_GROUPS] = {};
  hipEvent_t ev_fork = nullptr, ev_join[MAX_GROUPS] = {};
  // graph cache (one per group)
  // [group][0 = full chunk, 1 = lean chunk, 2 = short lean chunk, 3 = full chunk without k_assoc_dense, 4 = calm chunk (lean, one rebuild opportunity); + 5 for the instrumented kernels (CVO_KERNEL_CLOCK /
  // CVO_PHASE_TICKS), cached side by side so that a caller can time single steps of a loop without re-capturing]
  static constexpr int GRAPH_VARIANTS = 49;  // 8 graphs (see cvo_align_batch) x instrumented or not x 3 chunk lengths + the inner-product chain
  hipGraphExec_t graph_exec[MAX_GROUPS][GRAPH_VARIANTS] = {};
  GraphKey graph_key[MAX_GROUPS][GRAPH_VARIANTS] = {};
  int last_chunks = 0, last_lean_launches = 0, last_full_launches = 0;
  // last call (debug hooks)
  int last_pairs = 0;
  int last_N = 0, last_M = 0, last_Kmax = 0;
  DevParams last_params{};
  int last_gx = 0, last_gy = 0, last_csplit = 1;
  bool queue_open = false;  // a cvo_batch_queue owns the workspace: the other align / evaluation calls are refused meanwhile
  cvo_batch_queue* queue = nullptr;  // ... that queue (cvo_ctx_destroy releases its device side, see queue_release)
  double clock_ms_per_tick = 0.0;  // s_memrealtime, calibrated on first use (cvo_debug_kernel_clock)
  unsigned last_stride256 = 0;
  int last_Npad = 0;
  std::vector<int> last_xorder;  // pair 0's source order: sorted row -> original row
  int last_groups = 1;           // sub-batches (streams) of the last call
  int last_feat = 0;             // FEAT_* of the last call's association kernels
  PairLayout last_layout{};
};

namespace {

// value of option NAME (without the CVO_ prefix) or nullptr when it is not set
const char* ctx_opt(const cvo_ctx* ctx, const char* name) {
  if (!ctx) return nullptr;
  auto it = ctx->opt.find(name);
  return it == ctx->opt.end() ? nullptr : it->second.c_str();
}
bool ctx_opt_on(const cvo_ctx* ctx, const char* name) {  // set, and not to "0"
  const char* v = ctx_opt(ctx, name);
  return v && atoi(v) != 0;
}

int fail(cvo_ctx* ctx, int code, const std::string& msg) {
  if (ctx) ctx->err = msg;
  return code;
}

#define HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t e__ = (expr);                                                                     \
    if (e__ != hipSuccess)                                                                       \
      return fail(ctx, CVO_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));           \
  } while (0)

// A cloud that lacks an attribute array the kernels of a call dereference gets a zeroed one (once).
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
  L.flow_part = take(sizeof(double) * 8 * (size_t)nba);
  L.cnt_part = take(sizeof(unsigned long long) * 4 * (size_t)nba);
  L.coef_part = take(sizeof(double) * 4 * (size_t)nbc * COEFF_SPLIT_MAX);
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
  if (const char* e = ctx_opt(ctx, "SKIN_BLEND")) d.skin_blend = std::min(1.f, std::max(0.f, (float)atof(e)));
  d.skin_min = 0.05f;
  d.skin_max = 0.25f;
  if (const char* e = ctx_opt(ctx, "SKIN_MIN")) d.skin_min = std::max(0.f, (float)atof(e));
  if (const char* e = ctx_opt(ctx, "SKIN_MAX")) d.skin_max = std::max(d.skin_min, (float)atof(e));
  if (const char* e = ctx_opt(ctx, "LEAN_SKIN")) d.lean_skin = std::max(0.1f, (float)atof(e));
  d.horizon_margin = 0.3f;
  if (const char* e = ctx_opt(ctx, "HORIZON_MARGIN")) d.horizon_margin = std::max(0.f, (float)atof(e));
  d.rebuild_shrink = 0.9f;
  if (const char* e = ctx_opt(ctx, "SKIN")) d.skin_frac = std::max(0.f, (float)atof(e));
  d.phase_ticks = ctx_opt(ctx, "PHASE_TICKS") ? 1 : 0;
  d.kernel_clock = ctx_opt_on(ctx, "KERNEL_CLOCK") ? 1 : 0;
  d.verify_lists = ctx_opt_on(ctx, "VERIFY_LISTS") ? 1 : 0;
  d.debug_no_motion_bound = ctx_opt(ctx, "DEBUG_NO_MOTION_BOUND") ? 1 : 0;
  if (const char* e = ctx_opt(ctx, "SHRINK")) d.rebuild_shrink = std::min(0.99f, std::max(0.f, (float)atof(e)));
  return d;
}

// Scan geometry: T chunks of 64 sorted targets per wave (smaller slices cull better, larger ones
// amortise the row operands) and the number of row groups per block, chosen so that a launch has a
// few thousand waves (256 CUs x 4 SIMDs want several waves each).
void choose_scan_config(const cvo_ctx* ctx, int n_pairs, int NG, int Mpad, int* T_out, int* gpb_out) {
  int T = 2;
  const char* eT = ctx_opt(ctx, "SCAN_T");
  if (eT) {
    int v = atoi(eT);
    if (v == 1 || v == 2 || v == 4 || v == 8) T = v;
  }
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
  const char* eG = ctx_opt(ctx, "SCAN_GROUPS");
  if (eG) {
    int v = atoi(eG);
    if (v >= 64 && v % 64 == 0) gpb = v;
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
    hipLaunchKernelGGL(k_coeff<true>, row_grid(nblk * split, n_pairs), dim3(ASSOC_THREADS), 0, s, descs, dp, st, A.base, flags,
                       packed, A.stride256, A.Npad);
  else
    hipLaunchKernelGGL(k_coeff<false>, row_grid(nblk * split, n_pairs), dim3(ASSOC_THREADS), 0, s, descs, dp, st, A.base, flags,
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
    D.flow_part = (double*)(base + S->L.flow_part);
    D.cnt_part = (unsigned long long*)(base + S->L.cnt_part);
    D.coef_part = (double*)(base + S->L.coef_part);
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
  if (const char* e = ctx_opt(ctx, "ROW_MAX_BUSY")) dp.row_max_busy = std::max(1, std::min(atoi(e), (int)ASSOC_CAP16));
  dp.lean_U = 8;
  if (const char* e = ctx_opt(ctx, "LEAN_U")) dp.lean_U = std::max(1, atoi(e));
  // Calm pairs (see PairState::want_full).  In the end game the pose jitters around its optimum: the motion PER ITERATION
  // stays at ~10 % of a list's allowance while the allowance used SINCE THE BUILD stays below 5 % for hundreds of
  // iterations (CVO_VERBOSE=2 prints both), so a linear "outlives the next 64 iterations" test never fires.  Four
  // iterations of linear margin it is: 62.3 -> 61.4 ms per headline step, single pairs -1.5 ... -2.5 %, no additional waits.
  dp.calm_U = 4;
  if (const char* e = ctx_opt(ctx, "CALM_U")) dp.calm_U = std::max(0, atoi(e));
  dp.lean_U2 = 2;
  if (const char* e = ctx_opt(ctx, "LEAN_U2")) dp.lean_U2 = std::max(0, atoi(e));  // 0 = no short lean graph
  if (dp.lean_U2 >= dp.lean_U) dp.lean_U2 = 0;
  dp.shrink_align = n_pairs >= 8 ? 63 : 0;
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
// chunk) or 4 (a k_scan row group) consecutive sorted points is a compact box.  Only the speed of
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
    const bool start_nodense = allow_lean && S.N > 4096 && ctx_opt(ctx, "NO_NODENSE") == nullptr;
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
    long builds = 0, stalls = 0, its = 0;
    for (int p = 0; p < n_pairs; p++) {
      builds += ctx->h_states[p].n_builds;
      stalls += ctx->h_states[p].n_stalls;
      its += ctx->h_states[p].status ? ctx->h_states[p].iterations : ctx->h_states[p].k;
    }
    if (atoi(ctx_opt(ctx, "VERBOSE")) >= 2)
      for (int p = 0; p < std::min(n_pairs, 4); p++)
        fprintf(stderr, "[cvo]   pair %d: k %d, list allowance used %.3f, per iteration %.5f, want %d, builds %d, ell %.4f (built at %.4f)\n", p,
                ctx->h_states[p].k, ctx->h_states[p].last_used, ctx->h_states[p].last_rate, ctx->h_states[p].want_full,
                ctx->h_states[p].n_builds, ctx->h_states[p].ell, ctx->h_states[p].ell_build);
    fprintf(stderr, "[cvo] %d pairs, %d groups: %d chunks (%d full + %d lean group launches), iterations %ld, list builds %ld, waits %ld, %.3f ms\n",
            n_pairs, G, ctx->last_chunks, ctx->last_full_launches, ctx->last_lean_launches, its, builds, stalls, ms);
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
        r.info.ret = ps.ret;
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
  // queue of short solves wants short chunks; a boundary costs a stream ~10 us.  QUEUE_U (default 16 for the fast graphs,
  // twice that for the lean ones, as cvo_align_batch).
  int U = 16;
  if (const char* e = ctx_opt(ctx, "QUEUE_U")) U = std::max(2, std::min(atoi(e), 64));
  q->cfg = LoopCfg{U, 2 * U, std::max(1, std::min(q->dp.lean_U, U)), std::max(0, std::min(q->dp.lean_U2, U)), q->S.geom.instr ? 8 : 0};
  q->allow_lean = ctx_opt(ctx, "NO_LEAN") == nullptr;
  q->start_nodense = q->allow_lean && q->S.N > 4096 && ctx_opt(ctx, "NO_NODENSE") == nullptr;
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

// Nonzero counts (by position) and the values of the last evaluation's matrix in slot-major form, [slot][position], whatever
// the layout on the device: rows the wave-per-row kernels evaluated may keep their entries row-major (PairDesc::dense_off).
static int fetch_ell_values(cvo_ctx* ctx, const PairDesc& D, std::vector<unsigned>& nzp, std::vector<float>& ap, unsigned* max_out) {
  const int N = D.N;
  nzp.assign(N, 0u);
  HIP_TRY(ctx, hipMemcpy(nzp.data(), D.nnz_row, sizeof(unsigned) * (size_t)N, hipMemcpyDeviceToHost));
  std::vector<char> dense(N, 0);
  bool any_dense = false;
  unsigned mx = 0;
  for (int q = 0; q < N; q++) {
    dense[q] = (nzp[q] & NNZ_DENSE_FLAG) ? 1 : 0;
    any_dense = any_dense || dense[q];
    nzp[q] = nnz_count(nzp[q]);
    mx = std::max(mx, nzp[q]);
  }
  *max_out = mx;
  ap.assign((size_t)mx * N, 0.f);
  if (!mx) return CVO_OK;
  std::vector<int> off;
  bool any_run = false;
  if (any_dense) {
    off.resize(N);
    HIP_TRY(ctx, hipMemcpy(off.data(), D.dense_off, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost));
    for (int q = 0; q < N; q++) any_run = any_run || (dense[q] && off[q] >= 0);
  }
  // the slot-major part up to the longest row; the whole matrix when some row lives in the row-major part
  const size_t n_ent = any_run ? (size_t)N * (size_t)std::max(ctx->last_params.K_max, (int)mx) : (size_t)mx * N;
  std::vector<EllEntry> ep(n_ent);
  HIP_TRY(ctx, hipMemcpy(ep.data(), D.ell, sizeof(EllEntry) * n_ent, hipMemcpyDeviceToHost));
  for (int q = 0; q < N; q++) {
    const int o = (any_run && dense[q]) ? off[q] : -1;
    for (unsigned sl = 0; sl < nzp[q]; sl++) {
      const size_t e = ell_index(N, (int)sl, q, o);
      if (e >= n_ent) return fail(ctx, CVO_E_HIP, "fetch_ell_values: corrupt row run");
      ap[(size_t)sl * N + q] = ep[e].a;
    }
  }
  return CVO_OK;
}

// The per-row outputs of the last evaluation, re-indexed from k_list's positions to SORTED rows.
static int fetch_ell(cvo_ctx* ctx, int pair, std::vector<unsigned>& nz, std::vector<float>& a, std::vector<int>& j,
                     unsigned* max_out) {
  const PairDesc& D = ctx->h_descs[pair];
  const int N = D.N;
  std::vector<unsigned> nzp;
  std::vector<int> perm(N);
  std::vector<float> ap;
  unsigned mx = 0;
  {
    const int rc = fetch_ell_values(ctx, D, nzp, ap, &mx);
    if (rc != CVO_OK) return rc;
  }
  HIP_TRY(ctx, hipMemcpy(perm.data(), D.rowperm, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost));
  std::vector<int> jp((size_t)mx * N);
  if (mx) HIP_TRY(ctx, hipMemcpy(jp.data(), D.ell_j, sizeof(int) * (size_t)mx * N, hipMemcpyDeviceToHost));
  nz.assign(N, 0);
  a.assign((size_t)mx * N, 0.f);
  j.assign((size_t)mx * N, -1);
  for (int pos = 0; pos < N; pos++) {
    const int r = perm[pos];
    if (r < 0 || r >= N) return fail(ctx, CVO_E_HIP, "fetch_ell: corrupt row permutation");
    nz[r] = nzp[pos];
    for (unsigned s = 0; s < nzp[pos]; s++) {
      a[(size_t)s * N + r] = ap[(size_t)s * N + pos];
      j[(size_t)s * N + r] = jp[(size_t)s * N + pos];
    }
  }
  *max_out = mx;
  return CVO_OK;
}

// CSR export of the last single evaluation (gpu_association_to_cpu, CvoGPU_impl.cu:366-427)
static int export_association(cvo_ctx* ctx, int N, int* row_ptr, int* col, float* val, size_t capacity, size_t* nnz_out) {
  std::vector<unsigned> nz;
  std::vector<float> a;
  std::vector<int> jj;
  unsigned mx = 0;
  int rc = fetch_ell(ctx, 0, nz, a, jj, &mx);
  if (rc != CVO_OK) return rc;
  std::vector<int> sorted_of(N);  // original row -> sorted row (fetch_ell returns sorted rows)
  for (int r = 0; r < N; r++) sorted_of[ctx->last_xorder[r]] = r;
  size_t cnt = 0;
  for (int i = 0; i < N; i++) {
    row_ptr[i] = (int)cnt;
    const int r = sorted_of[i];
    for (unsigned s = 0; s < nz[r]; s++) {
      if (cnt < capacity && col && val) {
        col[cnt] = jj[(size_t)s * N + r];
        val[cnt] = a[(size_t)s * N + r];
      }
      cnt++;
    }
  }
  row_ptr[N] = (int)cnt;
  if (nnz_out) *nnz_out = cnt;
  if (cnt > capacity) return fail(ctx, CVO_E_NOMEM, "association capacity too small");
  return CVO_OK;
}

int cvo_association(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source, const cvo_cloud* target,
                    const float T[16], float ell, int* row_ptr, int* col, float* val, size_t capacity,
                    size_t* nnz_out) {
  if (!ctx || !row_ptr || !T) return fail(ctx, CVO_E_INVALID, "cvo_association: bad argument");
  if (!source || !target) return fail(ctx, CVO_E_INVALID, "null cloud");
  if (nnz_out) *nnz_out = 0;
  if (source->n == 0 || target->n == 0) return CVO_OK;  // CvoGPU.cu:1884-1885
  BatchSetup S;
  int rc = run_single_eval(ctx, params, source, target, T, ell, &S);
  if (rc != CVO_OK) return rc;
  return export_association(ctx, source->n, row_ptr, col, val, capacity, nnz_out);
}

// Eigen 3.3.9 Matrix3f::inverse() (Inverse.h, compute_inverse<..., 3>), as called on the host at CvoGPU.cu:1947:
// cofactors, det = c00*m00 + (c10*m10 + c20*m20), result = cofactor^T * (1/det), plain float arithmetic.
// m and out are ROW-major.
static void inverse3_eigen(const float m[9], float out[9]) {
  auto M = [&](int i, int j) { return m[3 * i + j]; };
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return M(i1, j1) * M(i2, j2) - M(i1, j2) * M(i2, j1);
  };
  const float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
  const float p0 = c0 * M(0, 0), p1 = c1 * M(1, 0), p2 = c2 * M(2, 0);
  const float det = p0 + (p1 + p2);
  const float invdet = 1.0f / det;
  out[0] = c0 * invdet;
  out[1] = c1 * invdet;
  out[2] = c2 * invdet;
  out[3] = cof(0, 1) * invdet;
  out[4] = cof(1, 1) * invdet;
  out[5] = cof(2, 1) * invdet;
  out[6] = cof(0, 2) * invdet;
  out[7] = cof(1, 2) * invdet;
  out[8] = cof(2, 2) * invdet;
}

// smallest eigenvalue of the symmetric part of a 3x3 matrix (cyclic Jacobi, double)
static double min_eig_sym3(const float a[9]) {
  double S[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) S[i][j] = 0.5 * ((double)a[3 * i + j] + (double)a[3 * j + i]);
  for (int sweep = 0; sweep < 30; sweep++) {
    const double off = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
    if (!(off > 1e-30)) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (S[p][q] == 0.0) continue;
        const double th = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 3; k++) {  // columns
          const double kp = S[k][p], kq = S[k][q];
          S[k][p] = c * kp - sn * kq;
          S[k][q] = sn * kp + c * kq;
        }
        for (int k = 0; k < 3; k++) {  // rows
          const double pk = S[p][k], qk = S[q][k];
          S[p][k] = c * pk - sn * qk;
          S[q][k] = sn * pk + c * qk;
        }
      }
  }
  return std::min(S[0][0], std::min(S[1][1], S[2][2]));
}

int cvo_association_non_isotropic(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source,
                                  const cvo_cloud* target, const float T[16], const float kernel_colmajor[9],
                                  int* row_ptr, int* col, float* val, size_t capacity, size_t* nnz_out) {
  if (!ctx || !params || !row_ptr || !T || !kernel_colmajor)
    return fail(ctx, CVO_E_INVALID, "cvo_association_non_isotropic: bad argument");
  if (!source || !target) return fail(ctx, CVO_E_INVALID, "null cloud");
  if (nnz_out) *nnz_out = 0;
  if (source->n == 0 || target->n == 0) return CVO_OK;  // CvoGPU.cu:1975-1976
  float km[9], extra[10];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) km[3 * i + j] = kernel_colmajor[3 * j + i];
  inverse3_eigen(km, extra);
  // The kernel has no cut-off of its own (CvoGPU.cu:236-238), but a = ck*k*sk can only exceed sp_thres while
  // k = sigma^2 exp(-d2/2) > sp_thres / (max ck * max sk), i.e. d^T Kinv d < d2m; with lambda = the smallest eigenvalue
  // of Kinv's symmetric part that bounds |d|^2 < d2m / lambda, which steers the scan (the exact arithmetic then
  // runs on the survivors only).  No usable bound (indefinite kernel, NaN) => every pair is a candidate.
  const double sigma2 = (double)params->sigma * params->sigma;
  const double cmax = params->is_using_intensity ? (double)params->c_sigma * params->c_sigma : 1.0;
  const double smax = params->is_using_semantics ? (double)params->s_sigma * params->s_sigma : 1.0;
  const double d2m = -2.0 * std::log((double)params->sp_thres / (sigma2 * cmax * smax));
  const double lam = min_eig_sym3(extra);
  double cull = INFINITY;
  if (params->is_using_geometry && std::isfinite(d2m) && std::isfinite(lam) && lam > 0.0)
    cull = d2m > 0.0 ? d2m / lam * 1.01 + 1e-12 : 0.0;
  extra[9] = (float)cull;
  if (!(extra[9] == extra[9])) extra[9] = INFINITY;
  BatchSetup S;
  int rc = run_single_eval(ctx, params, source, target, T, 1.0f, &S, extra);
  if (rc != CVO_OK) return rc;
  return export_association(ctx, source->n, row_ptr, col, val, capacity, nnz_out);
}

int cvo_edge_kernel_matrix(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* frame1, const cvo_cloud* frame2,
                           float ell, int num_neighbors, float* mat, int* ind, unsigned int* nonzeros,
                           unsigned int* nonzero_sum) {
  if (!ctx || !params || num_neighbors <= 0)
    return fail(ctx, CVO_E_INVALID, "cvo_edge_kernel_matrix: bad argument");
  if (!frame1 || !frame2) return fail(ctx, CVO_E_INVALID, "null cloud");
  if (nonzero_sum) *nonzero_sum = 0;
  if (frame1->n == 0 || frame2->n == 0) return CVO_OK;
  // fill_in_A_mat_gpu on the two (already transformed) frames with the caller's K and ell: a single evaluation
  // at the identity pose (R = I, T = 0 reproduces every coordinate exactly)
  cvo_params_t p = *params;
  p.nearest_neighbors_max = num_neighbors;
  const float I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  BatchSetup S;
  int rc = run_single_eval(ctx, &p, frame1, frame2, I, ell, &S);
  if (rc != CVO_OK) return rc;
  std::vector<unsigned> nz;
  std::vector<float> a;
  std::vector<int> jj;
  unsigned mx = 0;
  rc = fetch_ell(ctx, 0, nz, a, jj, &mx);
  if (rc != CVO_OK) return rc;
  const int N = frame1->n, K = num_neighbors;
  unsigned long long sum = 0;
  for (int r = 0; r < N; r++) {
    const int i = ctx->last_xorder[r];  // sorted row r holds original row i
    if (nonzeros) nonzeros[i] = nz[r];
    sum += nz[r];
    for (int s2 = 0; s2 < K; s2++) {  // the reference's cleared layout: mat = 0, ind = -1 beyond the row's entries
      const bool ok = (unsigned)s2 < nz[r];
      if (mat) mat[(size_t)i * K + s2] = ok ? a[(size_t)s2 * N + r] : 0.f;
      if (ind) ind[(size_t)i * K + s2] = ok ? jj[(size_t)s2 * N + r] : -1;
    }
  }
  if (nonzero_sum) *nonzero_sum = (unsigned int)sum;
  return CVO_OK;
}

int cvo_debug_last_ell(cvo_ctx* ctx, int K, float* mat, int* ind, unsigned int* nonzeros) {
  if (!ctx || ctx->last_pairs < 1 || K <= 0) return fail(ctx, CVO_E_INVALID, "cvo_debug_last_ell: bad argument");
  if (!ctx->last_params.keep_columns)
    return fail(ctx, CVO_E_INVALID, "cvo_debug_last_ell: the last call kept no column indices (request a trace, "
                                    "is_exporting_association or CVO_KEEP_COLUMNS=1)");
  std::vector<unsigned> nz;
  std::vector<float> a;
  std::vector<int> jj;
  unsigned mx = 0;
  int rc = fetch_ell(ctx, 0, nz, a, jj, &mx);
  if (rc != CVO_OK) return rc;
  const int N = ctx->h_descs[0].N;
  for (int r = 0; r < N; r++) {
    const int i = ctx->last_xorder[r];  // sorted row r holds original row i
    if (nonzeros) nonzeros[i] = nz[r];
    for (int s = 0; s < K; s++) {
      const bool ok = (unsigned)s < nz[r];
      if (mat) mat[(size_t)i * K + s] = ok ? a[(size_t)s * N + r] : 0.f;
      if (ind) ind[(size_t)i * K + s] = ok ? jj[(size_t)s * N + r] : -1;
    }
  }
  return CVO_OK;
}

// gpu_association_to_cpu(A_host, ..., num_neighbors) at the end of align_impl (CvoGPU.cu:1552-1556, CvoGPU_impl.cu:366-427)
int cvo_align_association(cvo_ctx* ctx, int pair, int* row_ptr, int* col, float* val, size_t capacity, size_t* nnz_out,
                          int* stride_written, int* stride_read) {
  if (!ctx || !row_ptr || pair < 0 || pair >= ctx->last_pairs || ctx->last_params.mode != 0)
    return fail(ctx, CVO_E_INVALID, "cvo_align_association: no align call to export from");
  if (!ctx->last_params.keep_columns)
    return fail(ctx, CVO_E_INVALID, "cvo_align_association: the last align ran without params.is_exporting_association (the "
                                    "column indices of the kernel matrix were not kept)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const PairDesc& D = ctx->h_descs[pair];
  const PairState& st = ctx->h_states[pair];
  const int N = D.N;
  // at least one se_kernel ran (the update of an executed iteration records its stride): upstream exports after ANY
  // loop that ran once, also one that left through `dist < eps_2` in iteration 0 with `iterations == 0`
  // (CvoGPU.cu:1505-1508, 1552; reachable with min_step < eps_2 when warm-started at the optimum)
  const bool executed = st.K_last > 0;
  const int Kw = executed ? st.K_last : st.K, Kr = st.K;  // written with / read with
  if (stride_written) *stride_written = Kw;
  if (stride_read) *stride_read = Kr;
  if (nnz_out) *nnz_out = 0;
  for (int i = 0; i <= N; i++) row_ptr[i] = 0;
  if (!executed || st.nnz == 0) return CVO_OK;  // `if (association_gpu.nonzero_sum == 0) return;`
  // the last iteration's matrix by position: count, original row index, entries (slot-major)
  std::vector<unsigned> nzp;
  std::vector<int> ip(N);
  std::vector<float> ea;  // values, [slot][position]
  unsigned mx = 0;
  {
    const int rc = fetch_ell_values(ctx, D, nzp, ea, &mx);
    if (rc != CVO_OK) return rc;
  }
  HIP_TRY(ctx, hipMemcpy(ip.data(), D.iorig, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost));
  std::vector<int> ej((size_t)mx * N);
  if (mx) HIP_TRY(ctx, hipMemcpy(ej.data(), D.ell_j, sizeof(int) * (size_t)mx * N, hipMemcpyDeviceToHost));
  std::vector<int> pos_of(N, -1);  // original row -> position
  for (int q = 0; q < N; q++) {
    if (ip[q] < 0 || ip[q] >= N) return fail(ctx, CVO_E_HIP, "cvo_align_association: corrupt row index");
    pos_of[ip[q]] = q;
  }
  // the reference's row-major buffer entry at flat index f (row stride Kw), defined for f < N * Kw
  auto buf = [&](size_t f, int* j, float* a) {
    const size_t r = f / (size_t)Kw, sidx = f % (size_t)Kw;
    if (r >= (size_t)N) {  // beyond what the last iteration cleared and wrote: leftovers upstream, the row ends here
      *j = -1;
      *a = 0.f;
      return;
    }
    const int q = pos_of[r];
    if (sidx < nzp[q]) {
      *j = ej[sidx * (size_t)N + q];
      *a = ea[sidx * (size_t)N + q];
    } else {
      *j = -1;
      *a = 0.f;
    }
  };
  size_t cnt = 0;
  for (int i = 0; i < N; i++) {
    row_ptr[i] = (int)cnt;
    if (nzp[pos_of[i]] == 0) continue;  // `if (nonzeros[i] > 0)`
    for (int c = 0; c < Kr; c++) {
      int j;
      float a;
      buf((size_t)i * Kr + c, &j, &a);
      if (j == -1) break;
      if (cnt < capacity && col && val) {
        col[cnt] = j;
        val[cnt] = a;
      }
      cnt++;
    }
  }
  row_ptr[N] = (int)cnt;
  if (nnz_out) *nnz_out = cnt;
  if (cnt > capacity) return fail(ctx, CVO_E_NOMEM, "association capacity too small");
  return CVO_OK;
}

int cvo_debug_scalar_math(cvo_ctx* ctx, int op, int n, const double* in, double* out) {
  if (!ctx || !in || !out || n <= 0 || op < 0 || op > 12) return fail(ctx, CVO_E_INVALID, "cvo_debug_scalar_math: bad argument");
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
                       A, 8 | 2 | (ctx_opt(ctx, "COEFF_NO_UPDATE") ? 16 : 0));
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
  int variant = 1;  // CVO_SCAN_DEBUG: 1 = no emission, 2 = no fine tiles (cost breakdown only)
  if (const char* e = ctx_opt(ctx, "SCAN_DEBUG")) variant |= atoi(e) << 1;
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
