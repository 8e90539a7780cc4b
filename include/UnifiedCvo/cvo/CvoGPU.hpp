// cvo::CvoGPU for MI355X: the public API of upstream include/UnifiedCvo/cvo/CvoGPU.hpp:49-229 (pairwise
// overloads) over the C-ABI of cvo_hip.h.  Differences forced by the missing dependencies: Mat4f instead
// of Eigen::Matrix4f (same 16-float column-major layout; UnifiedCvo/eigen_interop.hpp converts where Eigen exists),
// an array of the 192-byte CvoPoint record instead of pcl::PointCloud<CvoPoint> (UnifiedCvo/pcl_interop.hpp forwards
// pcl clouds where PCL exists).  The multi-frame overloads (Ceres IRLS) are out of scope.
#pragma once
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "cvo/Association.hpp"
#include "cvo/CvoParams.hpp"
#include "utils/CvoPoint.hpp"
#include "utils/CvoPointCloud.hpp"
#include "utils/data_type.hpp"

namespace cvo {

class CvoGPU {
 public:
  explicit CvoGPU(const std::string& yaml_param_file, int device = 0);
  ~CvoGPU();
  CvoGPU(const CvoGPU&) = delete;
  CvoGPU& operator=(const CvoGPU&) = delete;

  CvoParams& get_params() { return params; }
  // Upstream returns the DEVICE copy of the parameter struct (CvoGPU.hpp:52; its only users are the multi-frame IRLS
  // drivers, which hand it to CvoFrameGPU / BinaryStateGPU).  This backend passes the parameters to its kernels by
  // value at every launch, so there is no resident device copy to point at: the block the backend reads - the host
  // struct - is returned.  Valid as an opaque handle for this library's own classes; not dereferenceable in user
  // device code.
  const CvoParams* get_params_gpu() const { return &params; }
  // Upstream re-uploads *p_cpu to the device copy only (CvoGPU.cu:73-77); here the backend reads the
  // host struct at every call, so this stores *p_cpu as the parameters the next calls use.
  void write_params(const CvoParams* p_cpu);

  // 0 = success, -1 = the flow vanished (CvoGPU.cu:1454-1458).  Empty input: returns 0 and leaves
  // `transform` untouched (CvoGPU.cu:1614-1617).  Backend failures throw std::runtime_error.
  int align(const CvoPointCloud& source_points, const CvoPointCloud& target_points,
            const Mat4f& T_target_frame_to_source_frame, Mat4f& transform, Association* association = nullptr,
            double* registration_seconds = nullptr) const;
  // pcl overload (upstream CvoGPU.hpp:91-99): n records of the 192-byte AoS CvoPoint (PointSegmentedDistribution<5,19>).
  int align(const CvoPoint* source_cvo_points, int n_source, const CvoPoint* target_cvo_points, int n_target,
            const Mat4f& T_target_frame_to_source_frame, Mat4f& transform, Association* association = nullptr,
            double* registration_seconds = nullptr) const;

  // New: independent frame pairs solved concurrently on this object's GPU.  Returns per-pair 0 / -1.
  std::vector<int> align_batch(const std::vector<const CvoPointCloud*>& sources,
                               const std::vector<const CvoPointCloud*>& targets, const std::vector<Mat4f>& inits,
                               std::vector<Mat4f>& transforms, double* seconds = nullptr) const;
  // New: clouds that stay on the device across calls.  Upstream converts and uploads both clouds inside every align()
  // (CvoPointCloud_to_gpu, CvoGPU_impl.cu:206-285); a frame pipeline that matches a frame against several partners, or
  // a host that prepares batch k + 1 while batch k is being solved, uploads once (in parallel, cvo_cloud_upload_many).
  class ResidentClouds {
   public:
    ~ResidentClouds();
    ResidentClouds(const ResidentClouds&) = delete;
    ResidentClouds& operator=(const ResidentClouds&) = delete;
    int size() const { return (int)handles.size(); }

   private:
    friend class CvoGPU;
    ResidentClouds() = default;
    std::vector<cvo_cloud*> handles;
  };
  std::unique_ptr<ResidentClouds> upload_clouds(const std::vector<const CvoPointCloud*>& clouds, int host_threads = 0) const;
  std::vector<int> align_batch(const ResidentClouds& sources, const ResidentClouds& targets, const std::vector<Mat4f>& inits,
                               std::vector<Mat4f>& transforms, double* seconds = nullptr) const;
  // New: a STREAM of resident pairs through `slots` in-flight slots (cvo_batch_open / _submit / _poll, include/cvo_hip.h):
  // a pair that finishes hands its slice of the workspace to the next one at the next chunk boundary, so the GPU stays
  // full however different the pairs' iteration counts are - upstream's own use is a frame stream with warm starts
  // (main_cvo_gpu_align_raw_image.cpp:100-170).  pairs[k] = {index into sources, index into targets}; max_iterations[k]
  // (optional) = that pair's own iteration limit.  Transforms / return values in submission order, every pose
  // bit-identical to a solo align().  Memory: the workspace is sized up front for min(slots, pairs) pairs of the LARGEST
  // source x target sizes of the call - ~139 MB per slot at 10k x 10k with nearest_neighbors_max = 512 (17 GB for the
  // default 128 slots), N * M / 8 bytes of candidate bitmap per slot beyond that; a request that does not fit fails with
  // a sized CVO_E_NOMEM message (cvo_last_error).  The queue is closed on every way out, exceptions included.
  std::vector<int> align_stream(const ResidentClouds& sources, const ResidentClouds& targets,
                                const std::vector<std::pair<int, int>>& pairs, const std::vector<Mat4f>& inits,
                                std::vector<Mat4f>& transforms, int slots = 128, const std::vector<int>* max_iterations = nullptr,
                                double* seconds = nullptr) const;
  // cvo_ctx_advice of this object's context ("" = nothing to report; see include/cvo_hip.h, hardware queues)
  std::string advice() const;

  float function_angle(const CvoPointCloud& source_points, const CvoPointCloud& target_points,
                       const Mat4f& T_target_frame_to_source_frame, float ell, bool is_approximate = true,
                       bool is_gpu = true) const;
  float inner_product_gpu(const CvoPointCloud& source_points, const CvoPointCloud& target_points,
                          const Mat4f& T_target_frame_to_source_frame, float ell) const;
  // pcl overloads (upstream CvoGPU.hpp:167-171, 220-224; CvoGPU.cu:1796-1809, 1848-1873)
  float function_angle(const CvoPoint* source_cvo_points, int n_source, const CvoPoint* target_cvo_points, int n_target,
                       const Mat4f& T_target_frame_to_source_frame, float ell, bool is_approximate = true) const;
  float inner_product_gpu(const CvoPoint* source_cvo_points, int n_source, const CvoPoint* target_cvo_points, int n_target,
                          const Mat4f& T_target_frame_to_source_frame, float ell) const;
  // The reference's HOST function of the same name (upstream CvoGPU.cpp:95-213): NOT the function the GPU path
  // computes - plain ell (no range factor), no neighbour cap, no colour / semantic cut-offs, radius search instead of
  // the ordered scan - and not a fallback for anything: function_angle(..., is_gpu = false) routes here as upstream's does.
  float inner_product_cpu(const CvoPointCloud& source_points, const CvoPointCloud& target_points,
                          const Mat4f& T_target_frame_to_source_frame, float ell) const;
  void compute_association_gpu(const CvoPointCloud& source_points, const CvoPointCloud& target_points,
                               const Mat4f& T_target_frame_to_source_frame, float lengthscale,
                               Association& association) const;
  // Non-isotropic (Mahalanobis) kernel d^T K^-1 d: no geometric cut-off, geometric types off (CvoGPU.cu:1967-1988).
  void compute_association_gpu(const CvoPointCloud& source_points, const CvoPointCloud& target_points,
                               const Mat4f& T_target_frame_to_source_frame, const Mat3f& non_isotropic_kernel,
                               Association& association) const;

 private:
  CvoParams params;
  cvo_ctx* ctx = nullptr;
  // Upstream's align() const is re-entrant (all state is per call, CvoGPU.cu:1605-1632); here the const entry points
  // share one context (streams, cached workspace, graphs), which is not thread-safe: they serialise on this mutex, so
  // concurrent callers of ONE object get upstream's behaviour (upstream serialises them on the default stream as
  // well).  For concurrency use one CvoGPU per host thread, or align_batch.
  mutable std::mutex call_mutex;
};

}  // namespace cvo
