// The README demo driver of the reference (src/experiments/main_cvo_gpu_align_two_color_pcd.cpp) on the
// MI355X backend:  cvo_align_gpu_two_color_pcd source.pcd target.pcd params.yaml [ell] [geometric_only]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "cvo/CvoGPU.hpp"

static cvo::Vec3f get_pc_mean(const cvo::CvoPointCloud& pc) {
  cvo::Vec3f m{{0, 0, 0}};
  for (int k = 0; k < pc.num_points(); k++)
    for (int c = 0; c < 3; c++) m[c] = m[c] + pc.positions()[k][c];
  for (int c = 0; c < 3; c++) m[c] = m[c] / pc.num_points();
  return m;
}

int main(int argc, char* argv[]) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s source.pcd target.pcd cvo_params.yaml [ell_init] [geometric_only=0|1]\n", argv[0]);
    return 2;
  }
  cvo::CvoPointCloud source(argv[1]), target(argv[2]);
  const cvo::Vec3f sm = get_pc_mean(source), tm = get_pc_mean(target);
  const float dx = sm[0] - tm[0], dy = sm[1] - tm[1], dz = sm[2] - tm[2];
  const float dist = std::sqrt(dx * dx + dy * dy + dz * dz);
  cvo::CvoGPU cvo_align(argv[3]);
  cvo::CvoParams& p = cvo_align.get_params();
  p.ell_init = argc > 4 ? std::strtof(argv[4], nullptr) : dist;
  p.ell_decay_rate = p.ell_decay_rate_first_frame;
  p.ell_decay_start = p.ell_decay_start_first_frame;
  if (argc > 5 && std::atoi(argv[5])) p.is_using_intensity = 0;
  cvo_align.write_params(&p);
  std::printf("Start align... num_fixed is %d, num_moving is %d, ell_init %f\n", source.num_points(),
              target.num_points(), p.ell_init);
  cvo::Mat4f init = cvo::Mat4f::Identity(), result = cvo::Mat4f::Identity();
  double seconds = 0;
  const int ret = cvo_align.align(source, target, init, result, nullptr, &seconds);
  std::printf("ret %d\nTransform is\n", ret);
  for (int r = 0; r < 4; r++) std::printf("%12.8f %12.8f %12.8f %12.8f\n", result(r, 0), result(r, 1), result(r, 2), result(r, 3));
  cvo::CvoPointCloud old_pc(3, 19), new_pc(3, 19);
  cvo::CvoPointCloud::transform(init, target, old_pc);
  cvo::CvoPointCloud::transform(result, target, new_pc);
  (old_pc + source).write_to_color_pcd("before_align.pcd");
  (new_pc + source).write_to_color_pcd("after_align.pcd");
  std::printf("Average registration time is %f\n", seconds);
  return 0;
}
