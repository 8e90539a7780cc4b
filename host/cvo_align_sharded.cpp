// Batched multi-frame driver over the GPUs of one node (cvo::CvoGPUSharded: one context + host thread per device,
// pair p -> device p / ceil(n / n_devices), one ncclAllGather of the poses).
// usage: cvo_align_sharded [--bench REPS] params.yaml max_iter n_devices src0.pcd tgt0.pcd [src1.pcd tgt1.pcd ...]
// prints one line per pair: "pair p device d ret r T <16 floats, column-major>".
// --bench REPS: the clouds are uploaded ONCE (upload_batch), one warm-up solve, then REPS timed solves of the resident
// batch - solve on every device + the RCCL all-gather of the poses, the communicator alive the whole time - and a line
// "bench devices D pairs n reps R ms_per_batch min <ms> median <ms> align_per_s <n / min>" after the pose lines (which
// then show the last timed solve).  With 64 pairs of 10k points per device and MAX_ITER = 2000 this is BASELINE.json's
// headline workload on the C++ host; tests/test_cpp_host.py holds it to the C-ABI's own time.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "cvo/CvoGPUSharded.hpp"

int main(int argc, char* argv[]) {
  int reps = 0;
  if (argc > 2 && std::strcmp(argv[1], "--bench") == 0) {
    reps = std::max(1, std::atoi(argv[2]));
    argv += 2;
    argc -= 2;
  }
  if (argc < 6 || (argc - 4) % 2) {
    std::fprintf(stderr, "usage: %s [--bench REPS] cvo_params.yaml max_iter n_devices src0.pcd tgt0.pcd [src1.pcd tgt1.pcd ...]\n", argv[0]);
    return 2;
  }
  const int n_dev = std::atoi(argv[3]);
  std::vector<int> devices;
  for (int d = 0; d < n_dev; d++) devices.push_back(d);
  cvo::CvoGPUSharded cvo(argv[1], devices);
  cvo::CvoParams p = cvo.get_params();
  if (std::atoi(argv[2]) > 0) p.MAX_ITER = std::atoi(argv[2]);
  cvo.write_params(&p);
  std::vector<std::unique_ptr<cvo::CvoPointCloud>> clouds;
  std::vector<const cvo::CvoPointCloud*> src, tgt;
  for (int a = 4; a + 1 < argc; a += 2) {
    clouds.emplace_back(new cvo::CvoPointCloud(argv[a]));
    src.push_back(clouds.back().get());
    clouds.emplace_back(new cvo::CvoPointCloud(argv[a + 1]));
    tgt.push_back(clouds.back().get());
  }
  const int n = (int)src.size();
  std::vector<cvo::Mat4f> inits(n, cvo::Mat4f::Identity()), out;
  double seconds = 0;
  std::vector<int> rets;
  std::vector<double> ms;
  if (reps > 0) {
    cvo.upload_batch(src, tgt);
    rets = cvo.align_resident(inits, out, &seconds, cvo.num_devices() - 1);  // warm-up: workspace, graphs
    for (int r = 0; r < reps; r++) {
      const auto t0 = std::chrono::steady_clock::now();
      rets = cvo.align_resident(inits, out, &seconds, cvo.num_devices() - 1);
      ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
  } else {
    rets = cvo.align_batch(src, tgt, inits, out, &seconds, cvo.num_devices() - 1);
  }
  for (int q = 0; q < n; q++) {
    std::printf("pair %d device %d ret %d T", q, cvo.device_of(q, n), rets[q]);
    for (int k = 0; k < 16; k++) std::printf(" %.9g", out[q].m[k]);
    std::printf("\n");
  }
  if (reps > 0) {
    std::sort(ms.begin(), ms.end());
    std::printf("bench devices %d pairs %d reps %d ms_per_batch min %.3f median %.3f align_per_s %.1f\n", cvo.num_devices(), n,
                reps, ms.front(), ms[ms.size() / 2], n / (ms.front() * 1e-3));
    const std::string adv = cvo.advice();
    if (!adv.empty()) std::printf("advice %s\n", adv.c_str());
  }
  std::printf("devices %d pairs %d seconds %.6f\n", cvo.num_devices(), n, seconds);
  return 0;
}
