// Exercises the parts of the cvo::CvoGPU surface the README demo does not touch, for tests/test_cpp_host.py:
//   * the pcl overloads (arrays of the 192-byte CvoPoint record) of align / inner_product_gpu / function_angle,
//   * align(..., Association*) under is_exporting_association,
//   * inner_product_cpu and function_angle(..., is_gpu = false).
// usage: cvo_api_surface source.pcd target.pcd params.yaml max_iter ell [ell_init]
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <utility>
#include <vector>

#include "cvo/CvoGPU.hpp"

int main(int argc, char* argv[]) {
  if (argc < 6) {
    std::fprintf(stderr, "usage: %s source.pcd target.pcd cvo_params.yaml max_iter ell\n", argv[0]);
    return 2;
  }
  cvo::CvoPointCloud source(argv[1]), target(argv[2]);
  cvo::CvoGPU cvo_align(argv[3]);
  cvo::CvoParams& p = cvo_align.get_params();
  p.MAX_ITER = std::atoi(argv[4]);
  p.is_exporting_association = 1;
  if (argc > 6) p.ell_init = std::strtof(argv[6], nullptr);
  cvo_align.write_params(&p);
  const float ell = std::strtof(argv[5], nullptr);
  const cvo::Mat4f init = cvo::Mat4f::Identity();

  const std::vector<cvo::CvoPoint> ps = cvo::CvoPointCloud_to_cvo_points(source), pt = cvo::CvoPointCloud_to_cvo_points(target);
  cvo::Mat4f T_soa = cvo::Mat4f::Identity(), T_aos = cvo::Mat4f::Identity();
  cvo::Association A_soa, A_aos;
  const int r_soa = cvo_align.align(source, target, init, T_soa, &A_soa, nullptr);
  const int r_aos = cvo_align.align(ps.data(), (int)ps.size(), pt.data(), (int)pt.size(), init, T_aos, &A_aos, nullptr);
  std::printf("ret %d %d\n", r_soa, r_aos);
  std::printf("aos_equals_soa %d\n", (int)(std::memcmp(T_soa.data(), T_aos.data(), sizeof(float) * 16) == 0 &&
                                          A_soa.pairs.col == A_aos.pairs.col && A_soa.pairs.val == A_aos.pairs.val &&
                                          A_soa.pairs.row_ptr == A_aos.pairs.row_ptr));
  std::printf("transform");
  for (int q = 0; q < 16; q++) std::printf(" %.9g", T_soa.m[q]);
  std::printf("\n");
  double vs = 0;
  long cs = 0;
  for (size_t q = 0; q < A_soa.pairs.val.size(); q++) {
    vs += A_soa.pairs.val[q];
    cs += (long)A_soa.pairs.col[q] * (long)(q % 97 + 1);
  }
  std::printf("association nnz %zu rows %d cols %d source_inliers %zu target_inliers %zu value_sum %.9g col_checksum %ld\n",
              A_soa.pairs.nonZeros(), A_soa.pairs.rows, A_soa.pairs.cols, A_soa.source_inliers.size(),
              A_soa.target_inliers.size(), vs, cs);

  std::printf("inner_product_gpu %.9g %.9g\n", cvo_align.inner_product_gpu(source, target, init, ell),
              cvo_align.inner_product_gpu(ps.data(), (int)ps.size(), pt.data(), (int)pt.size(), init, ell));
  std::printf("function_angle_gpu %.9g %.9g %.9g %.9g\n", cvo_align.function_angle(source, target, init, ell, true, true),
              cvo_align.function_angle(ps.data(), (int)ps.size(), pt.data(), (int)pt.size(), init, ell, true),
              cvo_align.function_angle(source, target, init, ell, false, true),
              cvo_align.function_angle(ps.data(), (int)ps.size(), pt.data(), (int)pt.size(), init, ell, false));
  {  // align_stream: five submissions of the pair through two in-flight slots, two of them cut short
    std::vector<const cvo::CvoPointCloud*> cs{&source}, ct{&target};
    auto rs = cvo_align.upload_clouds(cs), rt = cvo_align.upload_clouds(ct);
    const std::vector<std::pair<int, int>> pairs(5, {0, 0});
    const std::vector<cvo::Mat4f> inits(5, init);
    const std::vector<int> limits{0, 7, 0, 7, 0};
    std::vector<cvo::Mat4f> Ts;
    const std::vector<int> rets = cvo_align.align_stream(*rs, *rt, pairs, inits, Ts, 2, &limits);
    bool ok = rets.size() == 5;
    for (int k = 0; ok && k < 5; k += 2) ok = rets[k] == r_soa && std::memcmp(Ts[k].data(), T_soa.data(), sizeof(float) * 16) == 0;
    ok = ok && std::memcmp(Ts[1].data(), Ts[3].data(), sizeof(float) * 16) == 0 && std::memcmp(Ts[1].data(), T_soa.data(), sizeof(float) * 16) != 0;
    std::printf("stream_equals_align %d\n", (int)ok);
  }
  std::printf("inner_product_cpu %.9g %.9g\n", cvo_align.inner_product_cpu(source, target, init, ell),
              cvo_align.inner_product_cpu(source, target, T_soa.inverse_rigid(), ell));
  std::printf("function_angle_cpu %.9g %.9g\n", cvo_align.function_angle(source, target, init, ell, true, false),
              cvo_align.function_angle(source, target, init, ell, false, false));
  return 0;
}
