// One edge of the multi-frame graph on the MI355X backend:
//   cvo_multiframe_edge frame1.pcd frame2.pcd params.yaml num_neighbors ell p2_0 ... p2_11
// Frame 1 sits at the identity, frame 2 under the 3x4 row-major pose p2; prints the kernel matrix summary the
// test compares with the Python mirror, runs the neighbour adaptation once more and one ell decay.
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "cvo/CvoGPU.hpp"
#include "cvo/IRLS_State_GPU.hpp"

static void report(const char* tag, const cvo::BinaryStateGPU& e) {
  const cvo::SparseKernelMat& A = e.get_inner_product_mat();
  double sum = 0;
  long long isum = 0;
  const int K = (int)e.num_neighbors();
  for (int r = 0; r < A.rows; r++)
    for (int c = 0; c < K; c++) {
      const int j = A.ind_row2col[(size_t)r * K + c];
      if (j < 0) break;
      sum += A.mat[(size_t)r * K + c];
      isum += (long long)j * (c + 1);
    }
  std::printf("%s nonzero_sum %u num_neighbors %d value_sum %.9g index_checksum %lld ell %.9g\n", tag, A.nonzero_sum, K, sum,
              isum, e.ell());
}

int main(int argc, char* argv[]) {
  if (argc < 18) {
    std::fprintf(stderr, "usage: %s frame1.pcd frame2.pcd params.yaml num_neighbors ell p2_0 ... p2_11\n", argv[0]);
    return 2;
  }
  cvo::CvoPointCloud c1(argv[1]), c2(argv[2]);
  cvo::CvoParams params;
  cvo::read_CvoParams_yaml(argv[3], &params);
  const unsigned K = (unsigned)std::atoi(argv[4]);
  const float ell = std::strtof(argv[5], nullptr);
  double p1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, p2[12];
  for (int i = 0; i < 12; i++) p2[i] = std::strtod(argv[6 + i], nullptr);
  auto f1 = std::make_shared<cvo::CvoFrameGPU>(&c1, p1);
  auto f2 = std::make_shared<cvo::CvoFrameGPU>(&c2, p2);
  cvo::BinaryStateGPU edge(f1, f2, &params, &params, K, ell);
  const int n0 = edge.update_inner_product();
  report("first", edge);
  const int n1 = edge.update_inner_product();  // neighbour budget adapts, entries stay
  report("second", edge);
  edge.update_ell();
  f2->pose_vec[3] += 0.25;
  f2->transform_pointcloud();
  edge.update_inner_product();
  report("moved", edge);
  return n0 == n1 ? 0 : 1;
}
