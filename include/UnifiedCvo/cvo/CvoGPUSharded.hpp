// cvo::CvoGPUSharded: the batched multi-frame mode over the GPUs of one node (SURVEY.md 8(b) "New", 8(e)).
// Independent frame pairs are sharded over the devices in contiguous blocks (pair p -> device p / ceil(n / n_devices),
// 64 per GPU for BASELINE.json configs[4]); every device solves its block with cvo_align_batch on its own context
// (one host thread per device, no traffic between devices while solving), then ONE ncclAllGather (RCCL over xGMI) of
// the 4x4 poses - 16 floats per pair - and one of the return codes leave the full result on every device.  There is
// no collective inside the optimiser loop.  Single process, ncclCommInitAll: the layout north_star describes for
// C++ hosts; the Python harness (unified_cvo_amd/sharding.py, bench.py) does the same with one process per GPU.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "cvo/CvoGPU.hpp"

namespace cvo {

class CvoGPUSharded {
 public:
  // devices: HIP device ordinals (empty = all visible devices)
  CvoGPUSharded(const std::string& yaml_param_file, const std::vector<int>& devices = {});
  ~CvoGPUSharded();
  CvoGPUSharded(const CvoGPUSharded&) = delete;
  CvoGPUSharded& operator=(const CvoGPUSharded&) = delete;

  int num_devices() const;
  CvoParams& get_params();                    // of device 0; write_params() copies them to every device
  void write_params(const CvoParams* p_cpu);

  // Solves n independent pairs; transforms[p] / the returned 0 / -1 codes are read back from DEVICE `read_from`'s copy
  // of the gathered result (any device holds all of it).  seconds: wall time of solve + gather.
  std::vector<int> align_batch(const std::vector<const CvoPointCloud*>& sources,
                               const std::vector<const CvoPointCloud*>& targets, const std::vector<Mat4f>& inits,
                               std::vector<Mat4f>& transforms, double* seconds = nullptr, int read_from = 0);
  // The same with the clouds kept on their devices across calls (CvoGPU::ResidentClouds): upload_batch() shards and
  // uploads n pairs once, align_resident() solves them from `inits` as often as the caller likes - what a frame
  // pipeline does that prepares batch k + 1 while batch k is solved, and what `cvo_align_sharded --bench` times.
  void upload_batch(const std::vector<const CvoPointCloud*>& sources, const std::vector<const CvoPointCloud*>& targets,
                    int host_threads_per_device = 0);
  std::vector<int> align_resident(const std::vector<Mat4f>& inits, std::vector<Mat4f>& transforms, double* seconds = nullptr,
                                  int read_from = 0);
  // cvo_ctx_advice of device 0's context ("" = nothing to report)
  std::string advice() const;
  // which device pair p of an n-pair batch runs on
  int device_of(int p, int n) const;

 private:
  struct Impl;
  std::unique_ptr<Impl> impl;
};

}  // namespace cvo
