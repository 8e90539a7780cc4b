// cvo::BinaryStateGPU: one edge of the multi-frame graph (upstream include/UnifiedCvo/cvo/IRLS_State_GPU.hpp:21-96,
// src/cvo/IRLS_State_GPU.cu:16-79, IRLS_State_GPU.cpp:54-57).  update_inner_product() recomputes the edge's kernel
// matrix A between the two frames under their current poses and leaves it on the host for the solver; the Ceres
// residual construction (add_residual_to_problem) is out of scope - get_inner_product_mat() exposes what it reads.
#pragma once
#include <memory>

#include "cvo/CvoFrame.hpp"
#include "cvo/CvoParams.hpp"
#include "cvo/SparseKernelMat.hpp"

namespace cvo {

class BinaryStateGPU {
 public:
  typedef std::shared_ptr<BinaryStateGPU> Ptr;

  // params_gpu is accepted for signature compatibility (upstream passes a device copy of the same struct).
  BinaryStateGPU(std::shared_ptr<CvoFrameGPU> pc1, std::shared_ptr<CvoFrameGPU> pc2, const CvoParams* params_cpu,
                 const CvoParams* params_gpu, unsigned int num_neighbor, float init_ell);
  ~BinaryStateGPU();
  BinaryStateGPU(const BinaryStateGPU&) = delete;
  BinaryStateGPU& operator=(const BinaryStateGPU&) = delete;

  // Returns A's number of nonzeros (IRLS_State_GPU.cu:43-70); the neighbour budget follows the previous result:
  // num_neighbors = min(init, 1.1 * max row count) once a result exists.
  int update_inner_product();
  void update_ell();  // ell *= multiframe_ell_decay_rate while above multiframe_ell_min

  const SparseKernelMat& get_inner_product_mat() const { return A_result_cpu_; }
  unsigned int num_neighbors() const { return num_neighbors_; }
  float ell() const { return ell_; }
  CvoFrame* frame1() { return frame1_.get(); }
  CvoFrame* frame2() { return frame2_.get(); }

 private:
  std::shared_ptr<CvoFrameGPU> frame1_, frame2_;
  const CvoParams* params_cpu_;
  unsigned int num_neighbors_;
  const unsigned int init_num_neighbors_;
  float ell_;
  int iter_ = 0;
  SparseKernelMat A_result_cpu_;
};

}  // namespace cvo
