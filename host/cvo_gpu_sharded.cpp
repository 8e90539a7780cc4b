// cvo::CvoGPUSharded over the C-ABI + RCCL (see include/UnifiedCvo/cvo/CvoGPUSharded.hpp).
#include "cvo/CvoGPUSharded.hpp"
#include "cvo/ShardPlan.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <thread>

namespace cvo {

namespace {
void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void nccl_ok(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}
}  // namespace

struct CvoGPUSharded::Impl {
  std::vector<int> devices;
  std::vector<std::unique_ptr<CvoGPU>> gpus;  // one context per device
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;           // the collective runs on its own stream per device
  std::vector<float*> send, recv;             // device buffers: this device's poses / everybody's
  std::vector<int*> send_ret, recv_ret;
  int cap_per = 0;                            // pairs per device the buffers hold
  // upload_batch(): the resident shards
  std::vector<std::unique_ptr<CvoGPU::ResidentClouds>> res_src, res_tgt;
  int res_n = -1;
  std::vector<int> solve_and_gather(int n, const std::vector<const CvoPointCloud*>* sources,
                                    const std::vector<const CvoPointCloud*>* targets, const std::vector<Mat4f>& inits,
                                    std::vector<Mat4f>& transforms, double* seconds, int read_from);

  void reserve(int per) {
    if (per <= cap_per) return;
    const int D = (int)devices.size();
    for (int d = 0; d < D; d++) {
      hip_ok(hipSetDevice(devices[d]), "hipSetDevice");
      if (send[d]) (void)hipFree(send[d]);
      if (recv[d]) (void)hipFree(recv[d]);
      if (send_ret[d]) (void)hipFree(send_ret[d]);
      if (recv_ret[d]) (void)hipFree(recv_ret[d]);
      hip_ok(hipMalloc(&send[d], sizeof(float) * 16 * (size_t)per), "hipMalloc");
      hip_ok(hipMalloc(&recv[d], sizeof(float) * 16 * (size_t)per * D), "hipMalloc");
      hip_ok(hipMalloc(&send_ret[d], sizeof(int) * (size_t)per), "hipMalloc");
      hip_ok(hipMalloc(&recv_ret[d], sizeof(int) * (size_t)per * D), "hipMalloc");
    }
    cap_per = per;
  }
};

CvoGPUSharded::CvoGPUSharded(const std::string& yaml, const std::vector<int>& devices_in) : impl(new Impl) {
  std::vector<int> devs = devices_in;
  cvo_process_hint_hw_queues();  // (before this host's first HIP call: include/cvo_hip.h, hardware queues)
  if (devs.empty()) {
    int n = 0;
    hip_ok(hipGetDeviceCount(&n), "hipGetDeviceCount");
    for (int d = 0; d < n; d++) devs.push_back(d);
  }
  if (devs.empty()) throw std::runtime_error("CvoGPUSharded: no HIP device");
  impl->devices = devs;
  const int D = (int)devs.size();
  for (int d = 0; d < D; d++) impl->gpus.emplace_back(new CvoGPU(yaml, devs[d]));
  impl->comms.resize(D);
  nccl_ok(ncclCommInitAll(impl->comms.data(), D, devs.data()), "ncclCommInitAll");
  impl->streams.assign(D, nullptr);
  impl->send.assign(D, nullptr);
  impl->recv.assign(D, nullptr);
  impl->send_ret.assign(D, nullptr);
  impl->recv_ret.assign(D, nullptr);
  for (int d = 0; d < D; d++) {
    hip_ok(hipSetDevice(devs[d]), "hipSetDevice");
    hip_ok(hipStreamCreateWithFlags(&impl->streams[d], hipStreamNonBlocking), "hipStreamCreate");
  }
}

CvoGPUSharded::~CvoGPUSharded() {
  if (!impl) return;
  const int D = (int)impl->devices.size();
  for (int d = 0; d < D; d++) {
    (void)hipSetDevice(impl->devices[d]);
    if (impl->streams[d]) (void)hipStreamSynchronize(impl->streams[d]);
    if (impl->comms[d]) (void)ncclCommDestroy(impl->comms[d]);
    if (impl->streams[d]) (void)hipStreamDestroy(impl->streams[d]);
    if (impl->send[d]) (void)hipFree(impl->send[d]);
    if (impl->recv[d]) (void)hipFree(impl->recv[d]);
    if (impl->send_ret[d]) (void)hipFree(impl->send_ret[d]);
    if (impl->recv_ret[d]) (void)hipFree(impl->recv_ret[d]);
  }
}

int CvoGPUSharded::num_devices() const { return (int)impl->devices.size(); }
CvoParams& CvoGPUSharded::get_params() { return impl->gpus[0]->get_params(); }
void CvoGPUSharded::write_params(const CvoParams* p) {
  for (auto& g : impl->gpus) g->write_params(p);
}
int CvoGPUSharded::device_of(int p, int n) const { return ShardPlan(n, (int)impl->devices.size()).device_of(p); }

std::vector<int> CvoGPUSharded::align_batch(const std::vector<const CvoPointCloud*>& sources,
                                            const std::vector<const CvoPointCloud*>& targets,
                                            const std::vector<Mat4f>& inits, std::vector<Mat4f>& transforms, double* seconds,
                                            int read_from) {
  const int n = (int)sources.size();
  if ((int)targets.size() != n || (int)inits.size() != n) throw std::runtime_error("align_batch: size mismatch");
  return impl->solve_and_gather(n, &sources, &targets, inits, transforms, seconds, read_from);
}

void CvoGPUSharded::upload_batch(const std::vector<const CvoPointCloud*>& sources,
                                 const std::vector<const CvoPointCloud*>& targets, int host_threads_per_device) {
  const int n = (int)sources.size();
  if ((int)targets.size() != n) throw std::runtime_error("upload_batch: size mismatch");
  const int D = (int)impl->devices.size();
  const ShardPlan plan(n, D);
  impl->res_src.clear();
  impl->res_tgt.clear();
  impl->res_src.resize(D);
  impl->res_tgt.resize(D);
  for (int d = 0; d < D; d++) {
    hip_ok(hipSetDevice(impl->devices[d]), "hipSetDevice");
    const int lo = plan.lo(d), hi = plan.hi(d);
    std::vector<const CvoPointCloud*> s(sources.begin() + lo, sources.begin() + std::max(lo, hi)),
        t(targets.begin() + lo, targets.begin() + std::max(lo, hi));
    impl->res_src[d] = impl->gpus[d]->upload_clouds(s, host_threads_per_device);
    impl->res_tgt[d] = impl->gpus[d]->upload_clouds(t, host_threads_per_device);
  }
  impl->res_n = n;
}

std::vector<int> CvoGPUSharded::align_resident(const std::vector<Mat4f>& inits, std::vector<Mat4f>& transforms, double* seconds,
                                               int read_from) {
  if (impl->res_n < 0) throw std::runtime_error("align_resident: upload_batch first");
  if ((int)inits.size() != impl->res_n) throw std::runtime_error("align_resident: size mismatch");
  return impl->solve_and_gather(impl->res_n, nullptr, nullptr, inits, transforms, seconds, read_from);
}

std::string CvoGPUSharded::advice() const { return impl->gpus[0]->advice(); }

std::vector<int> CvoGPUSharded::Impl::solve_and_gather(int n, const std::vector<const CvoPointCloud*>* sources_p,
                                                       const std::vector<const CvoPointCloud*>* targets_p,
                                                       const std::vector<Mat4f>& inits, std::vector<Mat4f>& transforms,
                                                       double* seconds, int read_from) {
  Impl* const impl = this;
  const int D = (int)impl->devices.size();
  if (read_from < 0 || read_from >= D) throw std::runtime_error("align_batch: read_from out of range");
  transforms.assign(n, Mat4f::Identity());
  std::vector<int> rets(n, 0);
  if (n == 0) return rets;
  const ShardPlan plan(n, D);  // contiguous blocks; the last devices may hold fewer (padded with identity poses)
  const int per = plan.per;
  impl->reserve(per);
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::exception_ptr> errs(D);
  std::vector<std::thread> th;
  for (int d = 0; d < D; d++)
    th.emplace_back([&, d] {
      try {
        hip_ok(hipSetDevice(impl->devices[d]), "hipSetDevice");
        const int lo = plan.lo(d), hi = plan.hi(d);
        std::vector<float> poses(16 * (size_t)per, 0.f);
        std::vector<int> rr(per, 0);
        for (int q = 0; q < per; q++) std::memcpy(&poses[16 * (size_t)q], Mat4f::Identity().data(), sizeof(float) * 16);
        if (hi > lo) {
          std::vector<Mat4f> in(inits.begin() + lo, inits.begin() + hi), out;
          std::vector<int> r;
          if (sources_p) {
            std::vector<const CvoPointCloud*> s(sources_p->begin() + lo, sources_p->begin() + hi),
                t(targets_p->begin() + lo, targets_p->begin() + hi);
            r = impl->gpus[d]->align_batch(s, t, in, out, nullptr);
          } else {
            r = impl->gpus[d]->align_batch(*impl->res_src[d], *impl->res_tgt[d], in, out, nullptr);
          }
          for (int q = 0; q < hi - lo; q++) {
            std::memcpy(&poses[16 * (size_t)q], out[q].data(), sizeof(float) * 16);
            rr[q] = r[q];
          }
        }
        hip_ok(hipMemcpyAsync(impl->send[d], poses.data(), sizeof(float) * poses.size(), hipMemcpyHostToDevice, impl->streams[d]),
               "hipMemcpyAsync");
        hip_ok(hipMemcpyAsync(impl->send_ret[d], rr.data(), sizeof(int) * rr.size(), hipMemcpyHostToDevice, impl->streams[d]),
               "hipMemcpyAsync");
        hip_ok(hipStreamSynchronize(impl->streams[d]), "hipStreamSynchronize");  // (the host vectors go out of scope)
      } catch (...) {
        errs[d] = std::current_exception();
      }
    });
  for (auto& t : th) t.join();
  for (int d = 0; d < D; d++)
    if (errs[d]) std::rethrow_exception(errs[d]);
  // the only exchange step of the path: poses (16 floats per pair) and return codes, gathered onto every device
  nccl_ok(ncclGroupStart(), "ncclGroupStart");
  for (int d = 0; d < D; d++) {
    nccl_ok(ncclAllGather(impl->send[d], impl->recv[d], 16 * (size_t)per, ncclFloat, impl->comms[d], impl->streams[d]), "ncclAllGather");
    nccl_ok(ncclAllGather(impl->send_ret[d], impl->recv_ret[d], (size_t)per, ncclInt32, impl->comms[d], impl->streams[d]),
            "ncclAllGather");
  }
  nccl_ok(ncclGroupEnd(), "ncclGroupEnd");
  for (int d = 0; d < D; d++) {
    hip_ok(hipSetDevice(impl->devices[d]), "hipSetDevice");
    hip_ok(hipStreamSynchronize(impl->streams[d]), "hipStreamSynchronize");
  }
  hip_ok(hipSetDevice(impl->devices[read_from]), "hipSetDevice");
  std::vector<float> all(16 * (size_t)per * D);
  std::vector<int> all_r((size_t)per * D);
  hip_ok(hipMemcpy(all.data(), impl->recv[read_from], sizeof(float) * all.size(), hipMemcpyDeviceToHost), "hipMemcpy");
  hip_ok(hipMemcpy(all_r.data(), impl->recv_ret[read_from], sizeof(int) * all_r.size(), hipMemcpyDeviceToHost), "hipMemcpy");
  for (int p = 0; p < n; p++) {  // rank-major blocks of `per` slots
    const size_t slot = (size_t)plan.slot_of(p);
    std::memcpy(transforms[p].data(), &all[16 * slot], sizeof(float) * 16);
    rets[p] = all_r[slot];
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return rets;
}

}  // namespace cvo
