// cvo_internal.h -- what the sections of the host side (cvo_ctx.hip, cvo_upload.hip, cvo_launch.hip, cvo_sched.hip,
// cvo_queue.hip, cvo_eval.hip, cvo_export.hip, cvo_debug.hip) share: the context, a resident cloud, the workspace layout of a
// pair, graph keys, option lookup and the error helpers.  Included once, by cvo_hip.hip.
#pragma once
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
      coef_part, shadow, trace, total;
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
    // list reuse / graphs (scripts/skin_sweep.py, early_sweep.py, first_chunk_sweep.sh)
    "SKIN", "SKIN_MAX", "LEAN_SKIN", "HORIZON_MARGIN", "SHRINK_ALIGN", "LEAN_U", "NO_LEAN", "NO_DENSE_REGIME", "FIXED_CHUNKS", "FIRST_U",
    "FIRST_CHUNKS", "STREAMS", "QUEUE_ADMIT",
    // A/B switches of the tests: every one of them leaves the results bit-identical
    "NO_SORT", "ORDER", "NO_LONG_LISTS", "ROW_MAX", "NO_ONEHOT", "IP_CHAIN", "KEEP_COLUMNS",
    // diagnostics
    "VERBOSE", "KERNEL_CLOCK", "PHASE_TICKS", "VERIFY_LISTS", "DEBUG_NO_MOTION_BOUND", "DEBUG_DROP_PARTIAL"};

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

}  // namespace
