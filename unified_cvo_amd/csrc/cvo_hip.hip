// cvo_hip.hip -- host side of the C-ABI declared in include/cvo_hip.h: contexts, HBM-resident
// clouds, workspace layout, and the enqueue-only optimiser loop (hipGraph replays of
// [k_scan, k_assoc, k_coeff, k_update, k_prep] with a device-side status word; no host round trip per
// iteration, unlike the ~15 blocking syncs per iteration of CvoGPU.cu:1387-1533).
//
// ONE translation unit, in sections.  The kernels are templates in headers (cvo_kernels.h) that share __device__ globals
// (g_phase_ticks ...) and are launched from several sections; without -fgpu-rdc every translation unit would get its own
// copy of those globals and of every kernel instantiation it touches, so the sections below are compiled together, in
// dependency order.  Each of them reads on its own next to cvo_internal.h:
//   cvo_ctx.hip     contexts, stream pool, options, workspace          cvo_upload.hip  resident clouds
//   cvo_launch.hip  every kernel launch of the solver                  cvo_sched.hip   setup, chunk graphs, cvo_align_batch
//   cvo_queue.hip   the batch queue                                    cvo_eval.hip    inner products, single evaluations
//   cvo_export.hip  association / ELL exports                          cvo_debug.hip   test and profiling hooks
#include "cvo_internal.h"

#include "cvo_ctx.hip"
#include "cvo_launch.hip"
#include "cvo_upload.hip"
#include "cvo_sched.hip"
#include "cvo_queue.hip"
#include "cvo_eval.hip"
#include "cvo_export.hip"
#include "cvo_debug.hip"
