// cvo_export.hip -- what leaves the device as a matrix: the ELL of the last evaluation re-indexed to rows, cvo_association (+ non-isotropic), cvo_edge_kernel_matrix, cvo_align_association, cvo_debug_last_ell.
// A SECTION of the one translation unit cvo_hip.hip (which includes the sections in dependency order and says why it is one
// unit); not compiled on its own.  Shared declarations: cvo_internal.h.
extern "C" {

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

}  // extern "C"
