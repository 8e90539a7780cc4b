/*
 * cvo_hip.h -- C-ABI of the MI355X (gfx950) backend for unified_cvo's pairwise
 * CvoGPU::align() / inner_product_gpu() / function_angle() hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  The C++
 * classes in include/UnifiedCvo/ (cvo::CvoGPU, cvo::CvoPointCloud, cvo::CvoParams) are a
 * thin veneer over these entry points, and a maintainer of the reference binds them as
 * shown in INTEGRATION.md.  Each entry point cites the reference interface it replaces
 * (file:line relative to the upstream repository).
 *
 * Conventions
 *   * 4x4 transforms are 16 floats, COLUMN-major (the memory layout of Eigen::Matrix4f).
 *   * `init_T` is the reference's T_target_frame_to_source_frame argument; it is taken as
 *     the running (R, T) directly (CvoGPU.cu:1363-1364).  `out_T` is the returned
 *     `transform` = [R^T | -R^T T] (CvoGPU.cu:94-112, 1562).
 *   * Return codes: 0 = ok, -1 = "flow vanished" exactly as the reference (CvoGPU.cu:1454-1458),
 *     <= -2 = argument / HIP errors (CVO_E_*); never exit().  cvo_last_error() gives text.
 *   * A context is bound to one HIP device and owns one stream; it is not thread-safe,
 *     distinct contexts are independent.
 */
#ifndef CVO_HIP_H
#define CVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVO_FEATURE_DIMENSIONS 5 /* CMakeLists.txt:498 (cvo_gpu_img_lib) */
#define CVO_NUM_CLASSES 19       /* CMakeLists.txt:498 */

#define CVO_OK 0
#define CVO_RET_FLOW_VANISHED (-1)
#define CVO_E_INVALID (-2)
#define CVO_E_HIP (-3)
#define CVO_E_NOMEM (-4)
#define CVO_E_UNSUPPORTED (-5)
#define CVO_E_VERIFY (-6) /* CVO_VERIFY_LISTS=1: a row derived from the cached candidate lists differed from the literal scan */

/* Layout-identical to cvo::CvoParams (include/UnifiedCvo/cvo/CvoParams.hpp:12-73): same
 * members, same order, same types, so a reference build can pass &params unchanged. */
typedef struct cvo_params_t {
  float ell_init_first_frame;
  float ell_init;
  float ell_min;
  int min_ell_iter_limit;
  float ell_max;
  double dl;
  double dl_step;
  float sigma;
  float sp_thres;
  float c;
  float d;
  float c_ell;
  float c_sigma;
  float s_ell;
  float s_sigma;
  int MAX_ITER;
  float eps;
  float eps_2;
  float min_step;
  float max_step;
  float step;
  int nearest_neighbors_max;
  float ell_decay_rate;
  float ell_decay_rate_first_frame;
  int ell_decay_start;
  int ell_decay_start_first_frame;
  int indicator_window_size;
  float indicator_stable_threshold;
  int is_pcl_visualization_on;
  int is_using_least_square;
  int is_ell_adaptive;
  int is_full_ip_matrix;
  int is_using_geometry;
  int is_using_intensity;
  int is_using_semantics;
  int is_using_range_ell;
  int is_using_kdtree;
  int is_exporting_association;
  int is_using_geometric_type;
  int multiframe_using_cpu;
  int multiframe_max_iters;
  float multiframe_ell_init;
  float multiframe_ell_min;
  int multiframe_iter_per_ell;
  float multiframe_ell_decay_rate;
  int multiframe_iterations_per_ell;
  int multiframe_iterations_per_solve;
  int multiframe_expected_points;
  float multiframe_downsample_voxel_size;
  int multiframe_num_neighbors;
  int multiframe_least_squares_num_threads;
  int multiframe_min_nonzeros;
} cvo_params_t;

/* Fills *p with the defaults of CvoParams::CvoParams() (CvoParams.hpp:75-126).  max_step and
 * step, which the reference leaves uninitialised, are set to 0.8 and 0 (documented in DESIGN.md). */
void cvo_params_default(cvo_params_t* p);

typedef struct cvo_ctx cvo_ctx;     /* device + stream + cached scratch */
typedef struct cvo_cloud cvo_cloud; /* a point cloud resident in HBM */

/* One record per optimiser iteration, written by the device when tracing is requested. */
typedef struct cvo_trace_t {
  int k;
  int K;
  float ell;
  float step;
  unsigned int nnz;
  unsigned int max_nnz;
  float omega[3];
  float v[3];
  double B, C, D, E;
  double dist;
  float R[9]; /* running R after the update, row-major */
  float T[3];
} cvo_trace_t;

/* Per-call outputs beyond the transform. */
typedef struct cvo_align_info_t {
  int iterations;        /* value of k when the loop ended ("cvo # of iterations", CvoGPU.cu:1545) */
  int ret;               /* 0 or -1, as CvoGPU::align returns (batch queue results: CVO_E_HIP if the pair was ended by a device-side synchronisation time-out) */
  float final_ell;
  int final_num_neighbors;
  double seconds;        /* registration_seconds: hipEvent time of the loop (CvoGPU.cu:1534-1560) */
} cvo_align_info_t;

/* Optional controls for cvo_align_ex / cvo_align_batch_ex (all zero = reference behaviour). */
typedef struct cvo_align_opts_t {
  int max_iterations;    /* >0: stop after this many iterations (parity tests, benchmarks) */
  int override_state;    /* 1: start from ell0 / K0 below instead of ell_init / nearest_neighbors_max */
  float ell0;
  int K0;
  cvo_trace_t* trace;    /* host array, or NULL */
  int trace_capacity;    /* rows available per pair */
  int trace_dense;       /* record every iteration k < trace_dense ... */
  int trace_every;       /* ... and every k % trace_every == 0 (0 = none) */
  int* n_trace;          /* out: rows written (per pair) */
  int iters_per_launch;  /* iterations enqueued per host check (0 = default) */
  int use_graph;         /* 0 = default (on), 1 = force plain launches, 2 = force graph */
  int kernel_clock;      /* 1: this call runs the instrumented instantiation of the per-iteration kernels, which time
                            every launch on the device clock (read back with cvo_debug_kernel_clock; ~3 % slower);
                            0 = follow the CVO_KERNEL_CLOCK environment switch */
} cvo_align_opts_t;

/* ---- context ------------------------------------------------------------------------ */
int cvo_ctx_create(int device, cvo_ctx** out);
void cvo_ctx_destroy(cvo_ctx* ctx);
const char* cvo_last_error(const cvo_ctx* ctx);
/* Hardware queues.  A batch (cvo_align_batch, CvoGPUSharded) runs on four sub-batch HIP streams that must sit on four
 * different hardware queues; HIP deals streams onto GPU_MAX_HW_QUEUES queues (default 4, read once at the process's
 * first HIP call) and the upload stream, RCCL and the host application bring streams of their own.
 * cvo_process_hint_hw_queues() puts GPU_MAX_HW_QUEUES=8 into the environment unless the variable is already set
 * (CVO_NO_HW_QUEUE_HINT=1 disables it) and returns the value in force: a PROCESS-WIDE side effect, which is why it is
 * an explicit call - make it before the process's first HIP call and before other threads read the environment
 * (setenv is not thread-safe).  The Python wrapper and cvo::CvoGPU's constructor call it; CVO_HW_QUEUE_HINT_AT_LOAD=1
 * makes the library do it when it is loaded.  If HIP was initialised earlier, or the variable says less than 8,
 * cvo_ctx_create leaves an advisory text in cvo_ctx_advice() ("" = nothing to report) and prints it once per process
 * on stderr (CVO_QUIET=1 silences the print).  Nothing but speed depends on it. */
int cvo_process_hint_hw_queues(void);
const char* cvo_ctx_advice(const cvo_ctx* ctx);
/* Destroys the HIP streams pooled from destroyed contexts (they are kept across contexts so that a later context finds
 * its sub-batch streams on the hardware queues the first one was given).  Optional, e.g. before unloading the library. */
void cvo_shutdown(void);
/* Tuning / diagnostic switches of a context (none changes a result; the list is in unified_cvo_amd/csrc/cvo_internal.h,
 * kOptionNames, and INTEGRATION.md).  A context reads CVO_<NAME> from the environment ONCE, in cvo_ctx_create; afterwards
 * only this call changes them (value NULL = unset), so no library call depends on the process environment while it
 * runs.  `name` with or without the CVO_ prefix.  Unknown names: CVO_E_INVALID. */
int cvo_ctx_set_option(cvo_ctx* ctx, const char* name, const char* value);
/* HIP stream of the context as an opaque pointer (hipStream_t). */
void* cvo_ctx_stream(cvo_ctx* ctx);
int cvo_ctx_synchronize(cvo_ctx* ctx);

/* ---- clouds: replaces CvoPointCloud_to_gpu (CvoGPU_impl.cu:206-285) ------------------
 * xyz: n x 3.  feat: n x 5 row-major or NULL.  label: n x 19 row-major or NULL.
 * geotype: n x 2 or NULL.  Missing arrays read as zeros, as the reference leaves the
 * default-constructed CvoPoint fields (PointSegmentedDistribution.hpp:40-56); they are neither uploaded nor allocated
 * until a call needs them.  Uploads of one context serialise on its upload stream (cvo_cloud_upload_many is the
 * parallel form); they never wait for, nor delay, a solve in flight.  A cloud handed to an align / inner-product
 * call may get a zeroed attribute slab attached by that call (see above): do not share ONE cvo_cloud between
 * concurrent calls of different contexts' threads. */
int cvo_cloud_upload(cvo_ctx* ctx, int n, const float* xyz, const float* feat, const float* label,
                     const float* geotype, cvo_cloud** out);
/* Replaces pcl_PointCloud_to_gpu (CvoGPU_impl.cu:287-362): n records of the 192-byte AoS
 * CvoPoint = pcl::PointSegmentedDistribution<5,19> (PointSegmentedDistribution.hpp:17-99). */
int cvo_cloud_upload_aos192(cvo_ctx* ctx, int n, const void* cvo_points, cvo_cloud** out);
/* n_clouds clouds at once from a pool of `threads` host threads (0 = default): the per-cloud work of cvo_cloud_upload -
 * spatial ordering, one allocation, one copy on the thread's own stream - runs in parallel.  n / xyz: per-cloud sizes and
 * pointers; feat / label / geotype: arrays of per-cloud pointers, the arrays or single entries may be NULL.
 * out: n_clouds handles (all NULL on error). */
int cvo_cloud_upload_many(cvo_ctx* ctx, int n_clouds, const int* n, const float* const* xyz, const float* const* feat,
                          const float* const* label, const float* const* geotype, int threads, cvo_cloud** out);
int cvo_cloud_size(const cvo_cloud* c);
void cvo_cloud_free(cvo_cloud* c);

/* ---- CvoGPU::align (CvoGPU.cu:1574-1632) --------------------------------------------- */
int cvo_align(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source, const cvo_cloud* target,
              const float init_T[16], float out_T[16], cvo_align_info_t* info);
int cvo_align_ex(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source, const cvo_cloud* target,
                 const float init_T[16], float out_T[16], cvo_align_info_t* info,
                 const cvo_align_opts_t* opts);

/* New (not in the reference): n independent frame pairs solved concurrently on the
 * context's device.  init_T / out_T: n x 16.  infos: n entries or NULL.  Returns CVO_OK or
 * a CVO_E_* code; the per-pair 0 / -1 results are in infos[i].ret. */
int cvo_align_batch(cvo_ctx* ctx, const cvo_params_t* params, int n_pairs, const cvo_cloud* const* sources,
                    const cvo_cloud* const* targets, const float* init_T, float* out_T,
                    cvo_align_info_t* infos, const cvo_align_opts_t* opts);
/* Copies the n x 16 result transforms of the last cvo_align_batch to DEVICE memory `dst`
 * on the context's stream (so a caller can hand them to an RCCL all-gather without a host
 * round trip). */
int cvo_batch_poses_to_device(cvo_ctx* ctx, void* dst_device, int n_pairs);

/* ---- batch queue (new, not in the reference): a STREAM of frame pairs through a fixed number of in-flight slots --------
 * cvo_align_batch takes a fixed set of pairs.  The reference's real use is a frame stream (one align() per frame, warm
 * starts, very different iteration counts: main_cvo_gpu_align_raw_image.cpp:100-170): here a pair that finishes hands its
 * slice of the workspace to the next submitted pair at the next chunk boundary, so `slots` pairs stay in flight however
 * long each of them runs.  Every pose is bit-identical to a solo cvo_align of the same pair.
 *   cvo_batch_open    sizes the workspace for `slots` pairs of up to max_source_points x max_target_points
 *                     (min_source_points: the smallest source cloud that will be submitted, 0 = same as max - small
 *                     clouds split their coefficient pass over more blocks and the launches must provide for it).
 *                     opts: only max_iterations is honoured.  While a queue is open the context's other align /
 *                     evaluation calls are refused.
 *   cvo_batch_submit  queues one pair (the clouds must stay alive until its result has been delivered); *ticket = its
 *                     number in submission order, from 0.  max_iterations > 0: this pair's own iteration limit (a warm
 *                     start that needs a few hundred iterations among cold starts that need thousands), at most the
 *                     queue's.
 *   cvo_batch_poll    drives the queue and delivers finished pairs IN SUBMISSION ORDER (a result is held back until
 *                     every earlier ticket has been delivered).  wait = 0: make progress, do not wait for results;
 *                     1: until at least one result can be delivered (or nothing is pending); 2: until everything
 *                     submitted so far has finished (or `capacity` results are ready).
 *   cvo_batch_pending submitted pairs not yet delivered.  cvo_batch_close waits for the device and releases the queue
 *                     (undelivered results are dropped).
 *   While a queue is open the context refuses cvo_align* / inner products / cvo_ctx_set_option (CVO_E_INVALID).
 *   cvo_ctx_destroy on a context with an open queue drains the queue's streams, releases its device side and ORPHANS the
 *   handle: every later call on it returns CVO_E_INVALID, cvo_batch_close then only frees the host object. */
typedef struct cvo_batch_queue cvo_batch_queue;
typedef struct cvo_batch_result_t {
  long long ticket;
  float transform[16]; /* column-major 4x4, as out_T of cvo_align */
  cvo_align_info_t info; /* info.seconds: wall time from the pair's placement into a slot to its retirement */
} cvo_batch_result_t;
int cvo_batch_open(cvo_ctx* ctx, const cvo_params_t* params, int slots, int max_source_points, int max_target_points,
                   int min_source_points, const cvo_align_opts_t* opts, cvo_batch_queue** out);
int cvo_batch_submit(cvo_batch_queue* q, const cvo_cloud* source, const cvo_cloud* target, const float init_T[16],
                     int max_iterations, long long* ticket);
int cvo_batch_poll(cvo_batch_queue* q, int wait, int capacity, cvo_batch_result_t* results, int* n_results);
int cvo_batch_pending(const cvo_batch_queue* q);
/* chunks launched (all sub-batches), how many of them were full graphs, slots filled so far (statistics) */
int cvo_batch_stats(const cvo_batch_queue* q, unsigned long long* chunks, unsigned long long* full_chunks,
                    unsigned long long* refills);
void cvo_batch_close(cvo_batch_queue* q);

/* ---- the Association align() exports (CvoGPU.cu:1552-1556 -> gpu_association_to_cpu, CvoGPU_impl.cu:366-427) -------
 * What `align(..., Association*)` returns when params.is_exporting_association is set: the kernel matrix of the LAST
 * EXECUTED iteration of the loop (pose before that iteration's update, that iteration's ell and num_neighbors), of
 * pair `pair` of the last cvo_align / cvo_align_ex / cvo_align_batch call on this context.  CSR as cvo_association:
 * row_ptr n_source + 1 ints; col / val up to `capacity` entries (NULL / 0 to size: CVO_E_NOMEM with *nnz_out set).
 * Stride: upstream reads its row-major buffers with the value num_neighbors has AFTER the loop.  After a `break`
 * (eps / eps_2) that is the stride they were written with.  When the loop ran out of iterations num_neighbors had
 * already been advanced (CvoGPU.cu:1529) and, if it changed, the buffers are read with another stride than they were
 * written with; that re-striding is reproduced here wherever it stays inside the part of the buffer the last iteration
 * defined (new stride <= old stride); beyond it upstream returns leftovers of earlier iterations, here the row ends
 * (see DESIGN.md "Association export").  *stride_written / *stride_read (optional) report the two values. */
int cvo_align_association(cvo_ctx* ctx, int pair, int* row_ptr, int* col, float* val, size_t capacity, size_t* nnz_out,
                          int* stride_written, int* stride_read);

/* ---- inner_product_gpu / function_angle (CvoGPU.cu:1780-1873) ------------------------
 * A_sum of fill_in_A_mat_gpu's matrix (first nearest_neighbors_max pairs of a row in ascending target index, a > sp_thres),
 * accumulated in double.  With a geometric cut-off the evaluation is ONE kernel launch over the resident clouds (no candidate
 * structure; up to three pairs - exact function_angle - in the same launch); a call in which a row finds more than
 * nearest_neighbors_max pairs, a call without geometry, or a context with option "IP_CHAIN" set runs the loop's
 * candidate-list chain instead.  Same pairs, same values; the two orders of summation agree in all but the last bits of
 * the double sum. */
int cvo_inner_product(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source,
                      const cvo_cloud* target, const float T[16], float ell, float* out);
int cvo_function_angle(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source,
                       const cvo_cloud* target, const float T[16], float ell, int is_approximate,
                       float* out);

/* ---- compute_association_gpu(float lengthscale) (CvoGPU.cu:1876-1911) -----------------
 * CSR of Association::pairs (row = source index, col = target index, ascending).
 * row_ptr: n_source + 1 ints.  col/val: capacity entries.  *nnz_out = pairs found (may
 * exceed capacity, in which case only row_ptr is complete and CVO_E_NOMEM is returned). */
int cvo_association(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source,
                    const cvo_cloud* target, const float T[16], float ell, int* row_ptr, int* col,
                    float* val, size_t capacity, size_t* nnz_out);

/* ---- compute_association_gpu(..., const Eigen::Matrix3f& non_isotropic_kernel) (CvoGPU.cu:1913-1995) ---------
 * Association under a Mahalanobis distance d^T kernel^-1 d (fill_in_A_mat_gpu_dense_mat_kernel, CvoGPU.cu:217-327):
 * no geometric cut-off, geometric types off, K = nearest_neighbors_max.  kernel: 9 floats in Eigen::Matrix3f
 * (column-major) layout.  Output as cvo_association. */
int cvo_association_non_isotropic(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* source,
                                  const cvo_cloud* target, const float T[16], const float kernel_colmajor[9],
                                  int* row_ptr, int* col, float* val, size_t capacity, size_t* nnz_out);

/* ---- multi-frame edge kernel: BinaryStateGPU::update_inner_product (IRLS_State_GPU.cu:43-79) -------------
 * A frame is a resident cloud under a pose (3x4 ROW-major floats, CvoFrameGPU.cu:7-61).
 * cvo_cloud_transformed = CvoFrameGPU::transform_pointcloud (transform_point_pose_vec, CvoGPU_impl.cu:85-185):
 * a new resident cloud with every point moved by the pose (features / labels / geometric types copied).
 * cvo_edge_kernel_matrix = fill_in_A_mat_gpu(frame1, frame2, num_neighbors, ell) + compute_nonzeros +
 * copy_internal_SparseKernelMat_gpu_to_cpu: mat / ind row-major [n1 x num_neighbors] in the reference's cleared
 * layout (0 / -1 beyond a row's entries), nonzeros [n1], *nonzero_sum = their sum (what the caller hands to
 * Ceres).  Any of the output pointers may be NULL. */
int cvo_cloud_transformed(cvo_ctx* ctx, const cvo_cloud* in, const float pose_3x4_rowmajor[12], cvo_cloud** out);
int cvo_edge_kernel_matrix(cvo_ctx* ctx, const cvo_params_t* params, const cvo_cloud* frame1_transformed,
                           const cvo_cloud* frame2_transformed, float ell, int num_neighbors, float* mat, int* ind,
                           unsigned int* nonzeros, unsigned int* nonzero_sum);

const char* cvo_version(void);

#ifdef __cplusplus
}
#endif
#endif
