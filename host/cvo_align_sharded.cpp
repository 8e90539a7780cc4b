// Batched multi-frame driver over the GPUs of one node (cvo::CvoGPUSharded: one context + host thread per device,
// pair p -> device p / ceil(n / n_devices), one ncclAllGather of the poses).
// usage: cvo_align_sharded params.yaml max_iter n_devices src0.pcd tgt0.pcd [src1.pcd tgt1.pcd ...]
// prints one line per pair: "pair p device d ret r T <16 floats, column-major>".
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "cvo/CvoGPUSharded.hpp"

int main(int argc, char* argv[]) {
  if (argc < 6 || (argc - 4) % 2) {
    std::fprintf(stderr, "usage: %s cvo_params.yaml max_iter n_devices src0.pcd tgt0.pcd [src1.pcd tgt1.pcd ...]\n", argv[0]);
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
  const std::vector<int> rets = cvo.align_batch(src, tgt, inits, out, &seconds, cvo.num_devices() - 1);
  for (int q = 0; q < n; q++) {
    std::printf("pair %d device %d ret %d T", q, cvo.device_of(q, n), rets[q]);
    for (int k = 0; k < 16; k++) std::printf(" %.9g", out[q].m[k]);
    std::printf("\n");
  }
  std::printf("devices %d pairs %d seconds %.6f\n", cvo.num_devices(), n, seconds);
  return 0;
}
