// The multi-frame edge kernel over the C-ABI: cvo::CvoFrame, cvo::CvoFrameGPU, cvo::BinaryStateGPU and the host
// half of cvo::SparseKernelMat (see the headers for the upstream file:line each one mirrors).
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

#include "cvo/IRLS_State_GPU.hpp"

namespace cvo {
namespace {

// One context per device, shared by all frames (upstream frames use the current CUDA device implicitly).
cvo_ctx* frame_context(int device) {
  static std::mutex m;
  static std::map<int, cvo_ctx*> ctxs;
  std::lock_guard<std::mutex> lock(m);
  auto it = ctxs.find(device);
  if (it != ctxs.end()) return it->second;
  cvo_ctx* c = nullptr;
  if (cvo_ctx_create(device, &c) != CVO_OK)
    throw std::runtime_error("CvoFrameGPU: no usable HIP device " + std::to_string(device));
  ctxs[device] = c;
  return c;
}

void check(cvo_ctx* ctx, int rc, const char* what) {
  if (rc <= CVO_E_INVALID) throw std::runtime_error(std::string(what) + ": " + cvo_last_error(ctx));
}

}  // namespace

// ---- SparseKernelMat (host) ----------------------------------------------------------------------------------
void init_internal_SparseKernelMat_cpu(int rows, int cols, SparseKernelMat* A) {
  A->rows = rows;
  A->cols = cols;
  A->nonzero_sum = 0;
  A->mat = new float[(size_t)rows * cols]();
  A->ind_row2col = new int[(size_t)rows * cols];
  A->nonzeros = new unsigned int[rows]();
  std::fill(A->ind_row2col, A->ind_row2col + (size_t)rows * cols, -1);
}
void delete_internal_SparseKernelMat_cpu(SparseKernelMat* A) {
  delete[] A->mat;
  delete[] A->ind_row2col;
  delete[] A->nonzeros;
  A->mat = nullptr;
  A->ind_row2col = nullptr;
  A->nonzeros = nullptr;
}
void clear_SparseKernelMat_cpu(SparseKernelMat* A, int num_neighbors) {
  A->nonzero_sum = 0;
  std::fill(A->mat, A->mat + (size_t)A->rows * num_neighbors, 0.f);
  std::fill(A->ind_row2col, A->ind_row2col + (size_t)A->rows * num_neighbors, -1);
  std::fill(A->nonzeros, A->nonzeros + A->rows, 0u);
}
unsigned int nonzeros(SparseKernelMat* A) { return A->nonzero_sum; }
unsigned int max_neighbors(SparseKernelMat* A) {
  return A->rows > 0 ? *std::max_element(A->nonzeros, A->nonzeros + A->rows) : 0u;
}

// ---- CvoFrame ------------------------------------------------------------------------------------------------
CvoFrame::CvoFrame(const CvoPointCloud* pts, const double poses[12]) : points(pts) {
  std::memcpy(pose_vec, poses, sizeof(double) * 12);
}
const std::vector<Vec3f>& CvoFrame::points_transformed() { return points_transformed_; }
void CvoFrame::transform_pointcloud() {
  float T[12];
  for (int i = 0; i < 12; i++) T[i] = (float)pose_vec[i];
  const int n = points->num_points();
  points_transformed_.resize(n);
  for (int i = 0; i < n; i++) {
    const Vec3f& p = points->positions()[i];
    for (int r = 0; r < 3; r++) points_transformed_[i][r] = T[4 * r] * p[0] + T[4 * r + 1] * p[1] + T[4 * r + 2] * p[2] + T[4 * r + 3];
  }
}

// ---- CvoFrameGPU ---------------------------------------------------------------------------------------------
CvoFrameGPU::CvoFrameGPU(const CvoPointCloud* pts, const double poses[12], int device)
    : CvoFrame(pts, poses), ctx_(frame_context(device)) {
  const int n = pts->num_points();
  std::vector<float> xyz(3 * (size_t)n), feat, label, geo(2 * (size_t)n, 0.f);
  for (int i = 0; i < n; i++)
    for (int c = 0; c < 3; c++) xyz[3 * (size_t)i + c] = pts->positions()[i][c];
  const MatXf& F = pts->features();
  if (F.rows() == n && F.cols() > 0) {
    feat.assign(CVO_FEATURE_DIMENSIONS * (size_t)n, 0.f);
    for (int i = 0; i < n; i++)
      for (int c = 0; c < CVO_FEATURE_DIMENSIONS && c < F.cols(); c++) feat[CVO_FEATURE_DIMENSIONS * (size_t)i + c] = F(i, c);
  }
  const MatXf& L = pts->labels();
  if (pts->num_classes() > 0 && L.rows() == n) {
    label.assign(CVO_NUM_CLASSES * (size_t)n, 0.f);
    for (int i = 0; i < n; i++)
      for (int c = 0; c < pts->num_classes() && c < CVO_NUM_CLASSES; c++) label[CVO_NUM_CLASSES * (size_t)i + c] = L(i, c);
  }
  const std::vector<float>& g = pts->geometric_types();
  for (size_t i = 0; i < geo.size() && i < g.size(); i++) geo[i] = g[i];
  check(ctx_, cvo_cloud_upload(ctx_, n, xyz.data(), feat.empty() ? nullptr : feat.data(),
                               label.empty() ? nullptr : label.data(), geo.data(), &init_),
        "cvo_cloud_upload");
  transform_pointcloud();
}
CvoFrameGPU::~CvoFrameGPU() {
  if (transformed_) cvo_cloud_free(transformed_);
  if (init_) cvo_cloud_free(init_);
}
void CvoFrameGPU::transform_pointcloud() {
  float pose[12];
  for (int i = 0; i < 12; i++) pose[i] = static_cast<float>(pose_vec[i]);
  cvo_cloud* t = nullptr;
  check(ctx_, cvo_cloud_transformed(ctx_, init_, pose, &t), "cvo_cloud_transformed");
  if (transformed_) cvo_cloud_free(transformed_);
  transformed_ = t;
}

// ---- BinaryStateGPU ------------------------------------------------------------------------------------------
BinaryStateGPU::BinaryStateGPU(std::shared_ptr<CvoFrameGPU> pc1, std::shared_ptr<CvoFrameGPU> pc2,
                               const CvoParams* params_cpu, const CvoParams* /*params_gpu*/, unsigned int num_neighbor,
                               float init_ell)
    : frame1_(pc1), frame2_(pc2), params_cpu_(params_cpu), num_neighbors_(num_neighbor),
      init_num_neighbors_(num_neighbor), ell_(init_ell) {
  if (!pc1 || !pc2 || !params_cpu || num_neighbor == 0) throw std::invalid_argument("BinaryStateGPU: bad argument");
  if (pc1->context() != pc2->context()) throw std::invalid_argument("BinaryStateGPU: frames live on different devices");
  init_internal_SparseKernelMat_cpu(pc1->points->num_points(), (int)num_neighbor, &A_result_cpu_);
}
BinaryStateGPU::~BinaryStateGPU() { delete_internal_SparseKernelMat_cpu(&A_result_cpu_); }

int BinaryStateGPU::update_inner_product() {
  const unsigned int last = max_neighbors(&A_result_cpu_);
  if (last > 0) num_neighbors_ = std::min(init_num_neighbors_, (unsigned int)(last * 1.1));
  clear_SparseKernelMat_cpu(&A_result_cpu_, (int)num_neighbors_);
  cvo_params_t p;
  static_assert(sizeof(cvo_params_t) == sizeof(CvoParams), "cvo_params_t must stay layout-identical to cvo::CvoParams");
  std::memcpy(&p, params_cpu_, sizeof(p));
  unsigned int total = 0;
  check(frame1_->context(),
        cvo_edge_kernel_matrix(frame1_->context(), &p, frame1_->points_transformed_gpu(),
                               frame2_->points_transformed_gpu(), ell_, (int)num_neighbors_, A_result_cpu_.mat,
                               A_result_cpu_.ind_row2col, A_result_cpu_.nonzeros, &total),
        "cvo_edge_kernel_matrix");
  A_result_cpu_.nonzero_sum = total;
  iter_++;
  return (int)total;
}

void BinaryStateGPU::update_ell() {
  if (ell_ > params_cpu_->multiframe_ell_min) ell_ = ell_ * params_cpu_->multiframe_ell_decay_rate;
}

}  // namespace cvo
