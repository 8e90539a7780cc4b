// cvo::CvoFrame / cvo::CvoFrameGPU: a point cloud under a pose, the vertex type of the multi-frame graph
// (upstream include/UnifiedCvo/cvo/CvoFrame.hpp:13-38, CvoFrameGPU.hpp:14-36).  Only what the edge kernel
// (BinaryStateGPU::update_inner_product) needs; the Ceres side of the multi-frame solver is out of scope.
#pragma once
#include <memory>
#include <vector>

#include "cvo_hip.h"
#include "utils/CvoPointCloud.hpp"
#include "utils/data_type.hpp"

namespace cvo {

struct CvoFrame {
  typedef std::shared_ptr<CvoFrame> Ptr;
  CvoFrame(const CvoPointCloud* pts, const double poses[12]);
  virtual ~CvoFrame() {}

  const CvoPointCloud* points;  // no ownership
  double pose_vec[12];          // 3x4 row-major [R t]

  const std::vector<Vec3f>& points_transformed();
  virtual void transform_pointcloud();  // host copy: R * p + t in float, like upstream CvoFrame.cpp

 private:
  std::vector<Vec3f> points_transformed_;
};

struct CvoFrameGPU : public CvoFrame {
  // The frame's clouds live on `device` (upstream uses the current CUDA device).
  CvoFrameGPU(const CvoPointCloud* pts, const double poses[12], int device = 0);
  ~CvoFrameGPU();
  CvoFrameGPU(const CvoFrameGPU&) = delete;
  CvoFrameGPU& operator=(const CvoFrameGPU&) = delete;

  // Re-derives the transformed device cloud from pose_vec (cast to float): upstream CvoFrameGPU.cu:44-61.
  void transform_pointcloud() override;

  const cvo_cloud* points_transformed_gpu() const { return transformed_; }
  cvo_ctx* context() const { return ctx_; }

 private:
  cvo_ctx* ctx_ = nullptr;
  cvo_cloud* init_ = nullptr;
  cvo_cloud* transformed_ = nullptr;
};

}  // namespace cvo
