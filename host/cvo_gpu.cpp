// cvo::CvoGPU over the C-ABI (see include/UnifiedCvo/cvo/CvoGPU.hpp).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "cvo/CvoGPU.hpp"

namespace cvo {
namespace {

struct DeviceCloud {
  cvo_cloud* h = nullptr;
  ~DeviceCloud() {
    if (h) cvo_cloud_free(h);
  }
};

void check(cvo_ctx* ctx, int rc, const char* what) {
  if (rc <= CVO_E_INVALID) throw std::runtime_error(std::string(what) + ": " + cvo_last_error(ctx));
}

// What CvoPointCloud_to_gpu builds per point (upstream CvoGPU_impl.cu:206-263).
void upload(cvo_ctx* ctx, const CvoPointCloud& pc, DeviceCloud& out) {
  const int n = pc.num_points();
  std::vector<float> xyz(3 * (size_t)n), feat, label, geo(2 * (size_t)n, 0.f);
  for (int i = 0; i < n; i++)
    for (int c = 0; c < 3; c++) xyz[3 * (size_t)i + c] = pc.positions()[i][c];
  const MatXf& F = pc.features();
  if (F.rows() == n && F.cols() > 0) {
    feat.assign(CVO_FEATURE_DIMENSIONS * (size_t)n, 0.f);
    for (int i = 0; i < n; i++)
      for (int c = 0; c < CVO_FEATURE_DIMENSIONS && c < F.cols(); c++) feat[CVO_FEATURE_DIMENSIONS * (size_t)i + c] = F(i, c);
  }
  const MatXf& L = pc.labels();
  if (pc.num_classes() > 0 && L.rows() == n) {
    label.assign(CVO_NUM_CLASSES * (size_t)n, 0.f);
    for (int i = 0; i < n; i++)
      for (int c = 0; c < pc.num_classes() && c < CVO_NUM_CLASSES; c++) label[CVO_NUM_CLASSES * (size_t)i + c] = L(i, c);
  }
  const std::vector<float>& g = pc.geometric_types();
  for (size_t i = 0; i < geo.size() && i < g.size(); i++) geo[i] = g[i];
  check(ctx, cvo_cloud_upload(ctx, n, xyz.data(), feat.empty() ? nullptr : feat.data(),
                              label.empty() ? nullptr : label.data(), geo.data(), &out.h),
        "cvo_cloud_upload");
}

void fill_association(cvo_ctx* ctx, const cvo_params_t& p, cvo_cloud* s, cvo_cloud* t, const Mat4f& T, float ell,
                      Association& a) {
  const int n = cvo_cloud_size(s);
  a.source_inliers.clear();
  a.target_inliers.clear();
  a.pairs.rows = n;
  a.pairs.cols = cvo_cloud_size(t);
  a.pairs.row_ptr.assign(n + 1, 0);
  size_t nnz = 0;
  int rc = cvo_association(ctx, &p, s, t, T.data(), ell, a.pairs.row_ptr.data(), nullptr, nullptr, 0, &nnz);
  if (rc != CVO_E_NOMEM) check(ctx, rc, "cvo_association");
  a.pairs.col.assign(nnz, 0);
  a.pairs.val.assign(nnz, 0.f);
  if (nnz)
    check(ctx, cvo_association(ctx, &p, s, t, T.data(), ell, a.pairs.row_ptr.data(), a.pairs.col.data(),
                               a.pairs.val.data(), nnz, &nnz),
          "cvo_association");
  for (int i = 0; i < n; i++)
    if (a.pairs.row_ptr[i + 1] > a.pairs.row_ptr[i]) a.source_inliers.push_back(i);
  a.target_inliers = a.pairs.col;
}

// The Association align() exports (upstream CvoGPU.cu:1552-1556 -> gpu_association_to_cpu, CvoGPU_impl.cu:366-427):
// the kernel matrix of the loop's LAST EXECUTED iteration, which is still resident in the context.
void fill_align_association(cvo_ctx* ctx, int n_source, int n_target, Association& a) {
  a.source_inliers.clear();
  a.target_inliers.clear();
  a.pairs.rows = n_source;
  a.pairs.cols = n_target;
  a.pairs.row_ptr.assign(n_source + 1, 0);
  size_t nnz = 0;
  int rc = cvo_align_association(ctx, 0, a.pairs.row_ptr.data(), nullptr, nullptr, 0, &nnz, nullptr, nullptr);
  if (rc != CVO_E_NOMEM) check(ctx, rc, "cvo_align_association");
  a.pairs.col.assign(nnz, 0);
  a.pairs.val.assign(nnz, 0.f);
  if (nnz)
    check(ctx, cvo_align_association(ctx, 0, a.pairs.row_ptr.data(), a.pairs.col.data(), a.pairs.val.data(), nnz, &nnz,
                                     nullptr, nullptr),
          "cvo_align_association");
  for (int i = 0; i < n_source; i++)
    if (a.pairs.row_ptr[i + 1] > a.pairs.row_ptr[i]) a.source_inliers.push_back(i);
  a.target_inliers = a.pairs.col;
}

}  // namespace

std::vector<CvoPoint> CvoPointCloud_to_cvo_points(const CvoPointCloud& pc) {
  const int n = pc.num_points();
  std::vector<CvoPoint> out((size_t)n);
  const MatXf& F = pc.features();
  const MatXf& L = pc.labels();
  const std::vector<float>& g = pc.geometric_types();
  for (int i = 0; i < n; i++) {
    CvoPoint p{};
    p.x = pc.positions()[i][0];
    p.y = pc.positions()[i][1];
    p.z = pc.positions()[i][2];
    p.data_w = 1.f;
    if (F.rows() == n)
      for (int c = 0; c < 5 && c < F.cols(); c++) p.features[c] = F(i, c);
    if (F.rows() == n && F.cols() >= 3) {  // r, g, b bytes (CvoGPU_impl.cu:229-234)
      const unsigned r = (unsigned)std::min(255.0, (double)F(i, 0) * 255.0), gg = (unsigned)std::min(255.0, (double)F(i, 1) * 255.0),
                     b = (unsigned)std::min(255.0, (double)F(i, 2) * 255.0);
      p.rgba = (r << 16) | (gg << 8) | b;
    }
    if (pc.num_classes() > 0 && L.rows() == n) {
      int best = 0;
      for (int c = 0; c < pc.num_classes() && c < 19; c++) {
        p.label_distribution[c] = L(i, c);
        if (L(i, c) > L(i, best)) best = c;
      }
      p.label = best;
    }
    if (g.size() >= 2 * (size_t)i + 2) {
      p.geometric_type[0] = g[2 * (size_t)i];
      p.geometric_type[1] = g[2 * (size_t)i + 1];
    }
    out[(size_t)i] = p;
  }
  return out;
}

CvoGPU::CvoGPU(const std::string& f, int device) {
  std::vector<std::string> warnings;
  read_CvoParams_yaml(f.c_str(), &params, &warnings);
  if (std::getenv("CVO_VERBOSE"))
    for (const std::string& w : warnings) std::fprintf(stderr, "[cvo] %s: %s\n", f.c_str(), w.c_str());
  // (process-wide, explicit: GPU_MAX_HW_QUEUES=8 unless the host chose a value - include/cvo_hip.h, hardware queues; it takes
  // effect if this is the process's first contact with HIP, otherwise cvo_ctx_advice says what to export)
  cvo_process_hint_hw_queues();
  int rc = cvo_ctx_create(device, &ctx);
  if (rc != CVO_OK) throw std::runtime_error("cvo_ctx_create failed: no usable HIP device " + std::to_string(device));
}

CvoGPU::~CvoGPU() {
  if (ctx) cvo_ctx_destroy(ctx);
}

void CvoGPU::write_params(const CvoParams* p_cpu) {
  if (p_cpu != &params) params = *p_cpu;
}

int CvoGPU::align(const CvoPointCloud& source_points, const CvoPointCloud& target_points, const Mat4f& init,
                  Mat4f& transform, Association* association, double* registration_seconds) const {
  std::lock_guard<std::mutex> lk(call_mutex);  // (see CvoGPU.hpp: the const entry points share one context)
  if (source_points.num_points() == 0 || target_points.num_points() == 0) return 0;
  DeviceCloud s, t;
  upload(ctx, source_points, s);
  upload(ctx, target_points, t);
  cvo_align_info_t info;
  Mat4f out = transform;
  int rc = cvo_align(ctx, &params, s.h, t.h, init.data(), out.data(), &info);
  check(ctx, rc, "cvo_align");
  transform = out;
  if (registration_seconds) *registration_seconds = info.seconds;
  if (params.is_exporting_association && association)  // upstream CvoGPU.cu:1552-1556
    fill_align_association(ctx, source_points.num_points(), target_points.num_points(), *association);
  return rc;
}

int CvoGPU::align(const CvoPoint* src_pts, int n_source, const CvoPoint* tgt_pts, int n_target, const Mat4f& init,
                  Mat4f& transform, Association* association, double* registration_seconds) const {
  std::lock_guard<std::mutex> lk(call_mutex);  // (see CvoGPU.hpp: the const entry points share one context)
  if (n_source == 0 || n_target == 0) return 0;
  DeviceCloud s, t;
  check(ctx, cvo_cloud_upload_aos192(ctx, n_source, src_pts, &s.h), "cvo_cloud_upload_aos192");
  check(ctx, cvo_cloud_upload_aos192(ctx, n_target, tgt_pts, &t.h), "cvo_cloud_upload_aos192");
  cvo_align_info_t info;
  Mat4f out = transform;
  int rc = cvo_align(ctx, &params, s.h, t.h, init.data(), out.data(), &info);
  check(ctx, rc, "cvo_align");
  transform = out;
  if (registration_seconds) *registration_seconds = info.seconds;
  if (params.is_exporting_association && association) fill_align_association(ctx, n_source, n_target, *association);
  return rc;
}

float CvoGPU::inner_product_gpu(const CvoPoint* src_pts, int n_source, const CvoPoint* tgt_pts, int n_target, const Mat4f& T,
                                float ell) const {
  std::lock_guard<std::mutex> lk(call_mutex);  // (see CvoGPU.hpp: the const entry points share one context)
  if (n_source == 0 || n_target == 0) return 0.f;
  DeviceCloud s, t;
  check(ctx, cvo_cloud_upload_aos192(ctx, n_source, src_pts, &s.h), "cvo_cloud_upload_aos192");
  check(ctx, cvo_cloud_upload_aos192(ctx, n_target, tgt_pts, &t.h), "cvo_cloud_upload_aos192");
  float v = 0;
  check(ctx, cvo_inner_product(ctx, &params, s.h, t.h, T.data(), ell, &v), "cvo_inner_product");
  return v;
}

float CvoGPU::function_angle(const CvoPoint* src_pts, int n_source, const CvoPoint* tgt_pts, int n_target, const Mat4f& T,
                             float ell, bool is_approximate) const {
  std::lock_guard<std::mutex> lk(call_mutex);  // (see CvoGPU.hpp: the const entry points share one context)
  if (n_source == 0 || n_target == 0) return 0.f;  // upstream CvoGPU.cu:1855-1857
  DeviceCloud s, t;
  check(ctx, cvo_cloud_upload_aos192(ctx, n_source, src_pts, &s.h), "cvo_cloud_upload_aos192");
  check(ctx, cvo_cloud_upload_aos192(ctx, n_target, tgt_pts, &t.h), "cvo_cloud_upload_aos192");
  float v = 0;
  check(ctx, cvo_function_angle(ctx, &params, s.h, t.h, T.data(), ell, is_approximate ? 1 : 0, &v), "cvo_function_angle");
  return v;
}

std::vector<int> CvoGPU::align_batch(const std::vector<const CvoPointCloud*>& sources,
                                     const std::vector<const CvoPointCloud*>& targets,
                                     const std::vector<Mat4f>& inits, std::vector<Mat4f>& transforms,
                                     double* seconds) const {
  std::lock_guard<std::mutex> lk(call_mutex);  // (see CvoGPU.hpp: the const entry points share one context)
  const int n = (int)sources.size();
  if ((int)targets.size() != n || (int)inits.size() != n) throw std::runtime_error("align_batch: size mismatch");
  std::vector<DeviceCloud> s(n), t(n);
  std::vector<const cvo_cloud*> sh(n), th(n);
  std::vector<float> init(16 * (size_t)n), out(16 * (size_t)n);
  for (int i = 0; i < n; i++) {
    upload(ctx, *sources[i], s[i]);
    upload(ctx, *targets[i], t[i]);
    sh[i] = s[i].h;
    th[i] = t[i].h;
    std::copy(inits[i].data(), inits[i].data() + 16, &init[16 * (size_t)i]);
  }
  std::vector<cvo_align_info_t> infos(n);
  check(ctx, cvo_align_batch(ctx, &params, n, sh.data(), th.data(), init.data(), out.data(), infos.data(), nullptr),
        "cvo_align_batch");
  transforms.resize(n);
  std::vector<int> rets(n);
  for (int i = 0; i < n; i++) {
    std::copy(&out[16 * (size_t)i], &out[16 * (size_t)i] + 16, transforms[i].data());
    rets[i] = infos[i].ret;
  }
  if (seconds && n) *seconds = infos[0].seconds;
  return rets;
}

CvoGPU::ResidentClouds::~ResidentClouds() {
  for (cvo_cloud* h : handles)
    if (h) cvo_cloud_free(h);
}

std::unique_ptr<CvoGPU::ResidentClouds> CvoGPU::upload_clouds(const std::vector<const CvoPointCloud*>& clouds,
                                                              int host_threads) const {
  std::lock_guard<std::mutex> lk(call_mutex);
  const int k = (int)clouds.size();
  // what CvoPointCloud_to_gpu builds per point (upstream CvoGPU_impl.cu:206-263), for all clouds, then ONE parallel upload
  std::vector<std::vector<float>> xyz(k), feat(k), label(k), geo(k);
  std::vector<int> n(k);
  std::vector<const float*> px(k), pf(k), pl(k), pg(k);
  for (int c = 0; c < k; c++) {
    const CvoPointCloud& pc = *clouds[c];
    n[c] = pc.num_points();
    xyz[c].resize(3 * (size_t)n[c]);
    for (int i = 0; i < n[c]; i++)
      for (int d = 0; d < 3; d++) xyz[c][3 * (size_t)i + d] = pc.positions()[i][d];
    const MatXf& F = pc.features();
    if (F.rows() == n[c] && F.cols() > 0) {
      feat[c].assign(CVO_FEATURE_DIMENSIONS * (size_t)n[c], 0.f);
      for (int i = 0; i < n[c]; i++)
        for (int d = 0; d < CVO_FEATURE_DIMENSIONS && d < F.cols(); d++) feat[c][CVO_FEATURE_DIMENSIONS * (size_t)i + d] = F(i, d);
    }
    const MatXf& L = pc.labels();
    if (pc.num_classes() > 0 && L.rows() == n[c]) {
      label[c].assign(CVO_NUM_CLASSES * (size_t)n[c], 0.f);
      for (int i = 0; i < n[c]; i++)
        for (int d = 0; d < pc.num_classes() && d < CVO_NUM_CLASSES; d++) label[c][CVO_NUM_CLASSES * (size_t)i + d] = L(i, d);
    }
    const std::vector<float>& g = pc.geometric_types();
    if (!g.empty()) {
      geo[c].assign(2 * (size_t)n[c], 0.f);
      for (size_t i = 0; i < geo[c].size() && i < g.size(); i++) geo[c][i] = g[i];
    }
    px[c] = xyz[c].data();
    pf[c] = feat[c].empty() ? nullptr : feat[c].data();
    pl[c] = label[c].empty() ? nullptr : label[c].data();
    pg[c] = geo[c].empty() ? nullptr : geo[c].data();
  }
  std::unique_ptr<ResidentClouds> out(new ResidentClouds());
  out->handles.assign(k, nullptr);
  if (k)
    check(ctx, cvo_cloud_upload_many(ctx, k, n.data(), px.data(), pf.data(), pl.data(), pg.data(), host_threads, out->handles.data()),
          "cvo_cloud_upload_many");
  return out;
}

std::vector<int> CvoGPU::align_batch(const ResidentClouds& sources, const ResidentClouds& targets, const std::vector<Mat4f>& inits,
                                     std::vector<Mat4f>& transforms, double* seconds) const {
  std::lock_guard<std::mutex> lk(call_mutex);
  const int n = sources.size();
  if (targets.size() != n || (int)inits.size() != n) throw std::runtime_error("align_batch: size mismatch");
  std::vector<float> init(16 * (size_t)n), out(16 * (size_t)n);
  for (int i = 0; i < n; i++) std::copy(inits[i].data(), inits[i].data() + 16, &init[16 * (size_t)i]);
  std::vector<cvo_align_info_t> infos(n);
  transforms.resize(n);
  std::vector<int> rets(n);
  if (n == 0) return rets;
  check(ctx, cvo_align_batch(ctx, &params, n, sources.handles.data(), targets.handles.data(), init.data(), out.data(), infos.data(),
                             nullptr),
        "cvo_align_batch");
  for (int i = 0; i < n; i++) {
    std::copy(&out[16 * (size_t)i], &out[16 * (size_t)i] + 16, transforms[i].data());
    rets[i] = infos[i].ret;
  }
  if (seconds) *seconds = infos[0].seconds;
  return rets;
}

std::vector<int> CvoGPU::align_stream(const ResidentClouds& sources, const ResidentClouds& targets,
                                      const std::vector<std::pair<int, int>>& pairs, const std::vector<Mat4f>& inits,
                                      std::vector<Mat4f>& transforms, int slots, const std::vector<int>* max_iterations,
                                      double* seconds) const {
  std::lock_guard<std::mutex> lk(call_mutex);
  const int n = (int)pairs.size();
  if ((int)inits.size() != n || (max_iterations && (int)max_iterations->size() != n)) throw std::runtime_error("align_stream: size mismatch");
  transforms.resize(n);
  std::vector<int> rets(n, 0);
  if (n == 0) return rets;
  int n_max = 0, n_min = 1 << 30, m_max = 0;
  for (const auto& pr : pairs) {
    if (pr.first < 0 || pr.first >= sources.size() || pr.second < 0 || pr.second >= targets.size())
      throw std::runtime_error("align_stream: pair index out of range");
    const int ns = cvo_cloud_size(sources.handles[pr.first]), nt = cvo_cloud_size(targets.handles[pr.second]);
    n_max = std::max(n_max, ns);
    n_min = std::min(n_min, ns);
    m_max = std::max(m_max, nt);
  }
  std::vector<cvo_batch_result_t> buf((size_t)n);  // (before the queue opens: nothing below may throw while it is open ...)
  struct QueueGuard {  // (... and if something does, the queue is closed on the way out: an open queue locks the context)
    cvo_batch_queue* q = nullptr;
    ~QueueGuard() {
      if (q) cvo_batch_close(q);
    }
  } guard;
  check(ctx, cvo_batch_open(ctx, &params, std::max(1, std::min(slots, n)), n_max, m_max, n_min, nullptr, &guard.q), "cvo_batch_open");
  cvo_batch_queue* const q = guard.q;
  const auto t0 = std::chrono::steady_clock::now();
  int got = 0, rc = CVO_OK;
  for (int k = 0; k < n && rc == CVO_OK; k++) {
    rc = cvo_batch_submit(q, sources.handles[pairs[k].first], targets.handles[pairs[k].second], inits[k].data(),
                          max_iterations ? (*max_iterations)[k] : 0, nullptr);
    int m = 0;
    if (rc == CVO_OK && (k & 15) == 15) rc = cvo_batch_poll(q, 0, n - got, buf.data() + got, &m);  // keep the device fed while submitting
    got += m;
  }
  while (rc == CVO_OK && got < n) {
    int m = 0;
    rc = cvo_batch_poll(q, 2, n - got, buf.data() + got, &m);
    got += m;
    if (rc == CVO_OK && m == 0 && cvo_batch_pending(q) == 0) break;
  }
  cvo_batch_close(q);
  guard.q = nullptr;
  check(ctx, rc, "cvo_batch_submit / cvo_batch_poll");
  if (got != n) throw std::runtime_error("align_stream: the queue delivered fewer results than pairs were submitted");
  for (int k = 0; k < n; k++) {  // (delivered in submission order: buf[k].ticket == k)
    std::copy(buf[k].transform, buf[k].transform + 16, transforms[k].data());
    rets[k] = buf[k].info.ret;
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rets;
}

std::string CvoGPU::advice() const { return cvo_ctx_advice(ctx); }

float CvoGPU::inner_product_gpu(const CvoPointCloud& a, const CvoPointCloud& b, const Mat4f& T, float ell) const {
  std::lock_guard<std::mutex> lk(call_mutex);  // (see CvoGPU.hpp: the const entry points share one context)
  if (a.num_points() == 0 || b.num_points() == 0) return 0.f;
  DeviceCloud s, t;
  upload(ctx, a, s);
  upload(ctx, b, t);
  float v = 0;
  check(ctx, cvo_inner_product(ctx, &params, s.h, t.h, T.data(), ell, &v), "cvo_inner_product");
  return v;
}

float CvoGPU::function_angle(const CvoPointCloud& a, const CvoPointCloud& b, const Mat4f& T, float ell,
                             bool is_approximate, bool is_gpu) const {
  std::lock_guard<std::mutex> lk(call_mutex);  // (see CvoGPU.hpp: the const entry points share one context)
  if (a.num_points() == 0 || b.num_points() == 0) return 0.f;
  if (!is_gpu) {  // upstream CvoGPU.cu:1827-1843: the host inner product in all three places
    const float fxfz = inner_product_cpu(a, b, T, ell);
    float fx_norm, fz_norm;
    if (is_approximate) {
      fx_norm = std::sqrt((float)a.num_points());
      fz_norm = std::sqrt((float)b.num_points());
    } else {
      const Mat4f I = Mat4f::Identity();
      fx_norm = std::sqrt(inner_product_cpu(a, a, I, ell));
      fz_norm = std::sqrt(inner_product_cpu(b, b, I, ell));
    }
    return fxfz / (fx_norm * fz_norm);
  }
  DeviceCloud s, t;
  upload(ctx, a, s);
  upload(ctx, b, t);
  float v = 0;
  check(ctx, cvo_function_angle(ctx, &params, s.h, t.h, T.data(), ell, is_approximate ? 1 : 0, &v),
        "cvo_function_angle");
  return v;
}

void CvoGPU::compute_association_gpu(const CvoPointCloud& a, const CvoPointCloud& b, const Mat4f& T, float ell,
                                     Association& association) const {
  std::lock_guard<std::mutex> lk(call_mutex);  // (see CvoGPU.hpp: the const entry points share one context)
  if (a.num_points() == 0 || b.num_points() == 0) return;
  DeviceCloud s, t;
  upload(ctx, a, s);
  upload(ctx, b, t);
  fill_association(ctx, params, s.h, t.h, T, ell, association);
}

void CvoGPU::compute_association_gpu(const CvoPointCloud& a, const CvoPointCloud& b, const Mat4f& T, const Mat3f& kernel,
                                     Association& association) const {
  std::lock_guard<std::mutex> lk(call_mutex);  // (see CvoGPU.hpp: the const entry points share one context)
  if (a.num_points() == 0 || b.num_points() == 0) return;
  DeviceCloud s, t;
  upload(ctx, a, s);
  upload(ctx, b, t);
  const int n = a.num_points();
  Association& as = association;
  as.source_inliers.clear();
  as.target_inliers.clear();
  as.pairs.rows = n;
  as.pairs.cols = b.num_points();
  as.pairs.row_ptr.assign(n + 1, 0);
  size_t nnz = 0;
  int rc = cvo_association_non_isotropic(ctx, &params, s.h, t.h, T.data(), kernel.data(), as.pairs.row_ptr.data(), nullptr,
                                         nullptr, 0, &nnz);
  if (rc != CVO_E_NOMEM) check(ctx, rc, "cvo_association_non_isotropic");
  as.pairs.col.assign(nnz, 0);
  as.pairs.val.assign(nnz, 0.f);
  if (nnz)
    check(ctx, cvo_association_non_isotropic(ctx, &params, s.h, t.h, T.data(), kernel.data(), as.pairs.row_ptr.data(),
                                             as.pairs.col.data(), as.pairs.val.data(), nnz, &nnz),
          "cvo_association_non_isotropic");
  for (int i = 0; i < n; i++)
    if (as.pairs.row_ptr[i + 1] > as.pairs.row_ptr[i]) as.source_inliers.push_back(i);
  as.target_inliers = as.pairs.col;
}

}  // namespace cvo
