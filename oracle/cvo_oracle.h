/*
 * cvo_oracle.h -- C interface of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (unified_cvo_amd/, include/,
 * the C-ABI library) may include, link or call anything declared here.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and only as the
 * checker / the CPU timing baseline.
 *
 * PARITY UNPINNED: the reference (UMich-CURLY/unified_cvo) ships no golden vectors or
 * known-answer tests for this path and none of its translation units can be compiled in
 * this image (they need CUDA/thrust, Eigen, PCL, Sophus, TBB, yaml-cpp).  The oracle is
 * a statement-by-statement restatement of the reference's CUDA path; its own pins are
 * independent numpy/scipy re-derivations (tests/test_oracle_*.py).
 */
#ifndef CVO_ORACLE_H
#define CVO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_FEATURE_DIMENSIONS 5   /* CMakeLists.txt:498 -DFEATURE_DIMENSIONS=5 */
#define ORACLE_NUM_CLASSES 19         /* CMakeLists.txt:498 -DNUM_CLASSES=19 */

/* The subset of cvo::CvoParams (include/UnifiedCvo/cvo/CvoParams.hpp:12-128) read by the path. */
typedef struct OracleParams {
  float ell_init, ell_min;
  float sigma, sp_thres, c, d, c_ell, c_sigma, s_ell, s_sigma;
  int MAX_ITER;
  float eps, eps_2, min_step, max_step;
  int nearest_neighbors_max;
  float ell_decay_rate;
  int ell_decay_start;
  int indicator_window_size;
  float indicator_stable_threshold;
  int is_using_geometry, is_using_intensity, is_using_semantics;
  int is_using_range_ell, is_using_geometric_type;
} OracleParams;

/* A point cloud as the device AoS record of the reference sees it (CvoGPU_impl.cu:206-263),
 * split into dense row-major arrays.  feat/label/geo may be NULL (= all zero). */
typedef struct OracleCloud {
  int n;
  const float* xyz;   /* n x 3 */
  const float* feat;  /* n x 5  or NULL */
  const float* label; /* n x 19 or NULL */
  const float* geo;   /* n x 2  or NULL */
} OracleCloud;

/* One row per optimiser iteration (state AFTER the iteration's update). */
typedef struct OracleTrace {
  int k;                /* iteration index */
  int K;                /* num_neighbors used by this iteration */
  float ell;            /* lengthscale used by this iteration */
  float step;
  unsigned int nnz;     /* A_host.nonzero_sum */
  unsigned int max_nnz; /* max_i nonzeros[i] */
  float omega[3], v[3]; /* normalised twist */
  double B, C, D, E;
  double dist;          /* ||log(dRT)|| */
  float R[9];           /* row-major running R after the update */
  float T[3];
} OracleTrace;

/* ---- scalar building blocks (each pinned by tests/test_oracle_math.py) ---- */
void oracle_cubic_roots(const double coef[4], double re[3], double im[3]);
float oracle_select_step(double B, double C, double D, double E, float min_step, float max_step);
void oracle_exp_sek3(const float xi[6], float dt, float out_rowmajor_3x4[12]);
double oracle_se3_log_norm(const double dR_rowmajor[9], const double dT[3]);
/* runs A_sparsity_indicator_ell_update over a sequence; decisions[i] = returned bool */
void oracle_indicator_run(const float* indicators, int n, int window, float thr, unsigned char* decisions);

/* ---- kernel-level ---- */
void oracle_update_tf(const float R[9], const float T[3], float R_inv[9], float T_inv[3]);
void oracle_transform(const float R_inv[9], const float T_inv[3], int m, const float* y0, float* yt);
/* transform_point_pose_vec (CvoGPU_impl.cu:85-161) over a cloud: 3x4 ROW-major pose times (x, y, z, 1). */
void oracle_transform_pose_vec(const float pose12[12], int n, const float* xyz_in, float* xyz_out);
/* fill_in_A_mat_gpu: ELL output with row stride K. mat/ind sized n*K, nonzeros sized n. */
void oracle_se_kernel(const OracleParams* p, const OracleCloud* x, const OracleCloud* y_transformed,
                      int K, float ell, float* mat, int* ind, unsigned int* nonzeros, int literal);

/* One full iteration on a given state; mirrors the loop body of align_impl.
 * Returns 0 = continue, 1 = break by eps (ret in *ret_code), 2 = break by eps_2. */
int oracle_iteration(const OracleParams* p, const OracleCloud* x, const OracleCloud* y,
                     float R[9], float T[3], float ell, int K, OracleTrace* out, int* ret_code,
                     float* ell_mat_out, int* ell_ind_out, unsigned int* nonzeros_out);

/* CvoGPU::align (CvoPointCloud overload).  init/out are 4x4 COLUMN-major (Eigen::Matrix4f layout).
 * trace may be NULL; at most max_trace rows are written; *n_trace = rows written.
 * trace_every: record iterations k < trace_dense and every k % trace_every == 0. */
int oracle_align(const OracleParams* p, const OracleCloud* x, const OracleCloud* y,
                 const float init_colmajor[16], float out_colmajor[16], int* iterations,
                 OracleTrace* trace, int max_trace, int trace_dense, int trace_every, int* n_trace,
                 double* seconds, int max_iter_override);

/* align() with is_exporting_association: the Association the reference fills after the loop (CvoGPU.cu:1552-1556 ->
 * gpu_association_to_cpu, CvoGPU_impl.cu:366-427) from the kernel matrix of the LAST EXECUTED iteration, read with the
 * stride num_neighbors has after the loop.  Pairs as (row, col, val) triplets in row order; caller-owned arrays. */
typedef struct OracleAssociation {
  int cap_pairs, cap_rows;   /* capacities of row/col/val and of source_inliers */
  int* row;
  int* col;
  float* val;
  int* source_inliers;       /* may be NULL */
  int n_pairs, n_source_inliers;
  int K_used, K_final;       /* stride the matrix was written with / read with */
  int overflow;              /* 1: more pairs than cap_pairs (n_pairs still counts them) */
} OracleAssociation;
int oracle_align_association(const OracleParams* p, const OracleCloud* x, const OracleCloud* y,
                             const float init_colmajor[16], float out_colmajor[16], int* iterations,
                             int max_iter_override, OracleAssociation* assoc);

float oracle_inner_product(const OracleParams* p, const OracleCloud* x, const OracleCloud* y,
                           const float T_colmajor[16], float ell);
float oracle_function_angle(const OracleParams* p, const OracleCloud* x, const OracleCloud* y,
                            const float T_colmajor[16], float ell, int is_approximate);

/* Association export (gpu_association_to_cpu, CvoGPU_impl.cu:366-427) in CSR form.
 * Returns number of pairs; arrays sized by caller (n*K worst case). */
int oracle_association(const OracleParams* p, const OracleCloud* x, const OracleCloud* y,
                       const float T_colmajor[16], float ell, int* row_ptr /* n+1 */, int* col,
                       float* val);

/* Non-isotropic kernel: compute_association_gpu(..., const Eigen::Matrix3f& kernel) (CvoGPU.cu:217-327, 1913-1995):
 * Mahalanobis distance d^T kernel^-1 d, no geometric cut-off, geometric types off.  kernel_colmajor: 9 floats in
 * Eigen::Matrix3f layout.  kernel_inv_rowmajor_out (optional, 9 floats) receives the restated Eigen inverse. */
int oracle_association_non_isotropic(const OracleParams* p, const OracleCloud* x, const OracleCloud* y,
                                     const float T_colmajor[16], const float kernel_colmajor[9], int* row_ptr,
                                     int* col, float* val, float* kernel_inv_rowmajor_out);

/* 1: se_kernel generates each row's candidates from a uniform grid over the targets ("best-effort CPU" timing
 * variant, identical results); 0 (default): dense scan as the reference's GPU kernel does it. */
void oracle_set_grid(int on);
int oracle_get_grid(void);
/* Seconds spent in the association scan (se_kernel) by oracle_align / oracle_iteration since the last reset:
 * the K2 share of the CPU baseline. */
double oracle_scan_seconds(int reset);

int oracle_num_threads(void);
void oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
