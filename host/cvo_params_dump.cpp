// Prints every member of cvo::CvoParams after reading a yaml file (used by tests/test_cpp_host.py).
#include <cstdio>

#include "cvo/CvoParams.hpp"

int main(int argc, char** argv) {
  cvo::CvoParams p;
  std::vector<std::string> warnings;
  if (argc > 1) {
    try {
      cvo::read_CvoParams_yaml(argv[1], &p, &warnings);
    } catch (const std::exception& e) {
      std::fprintf(stderr, "error: %s\n", e.what());
      return 1;
    }
  }
#define PF(n) std::printf(#n "=%.9g\n", (double)p.n)
#define PI(n) std::printf(#n "=%d\n", p.n)
  PF(ell_init_first_frame); PF(ell_init); PF(ell_min); PI(min_ell_iter_limit); PF(ell_max); PF(dl); PF(dl_step);
  PF(sigma); PF(sp_thres); PF(c); PF(d); PF(c_ell); PF(c_sigma); PF(s_ell); PF(s_sigma); PI(MAX_ITER); PF(eps);
  PF(eps_2); PF(min_step); PF(max_step); PF(step); PI(nearest_neighbors_max); PF(ell_decay_rate);
  PF(ell_decay_rate_first_frame); PI(ell_decay_start); PI(ell_decay_start_first_frame); PI(indicator_window_size);
  PF(indicator_stable_threshold); PI(is_pcl_visualization_on); PI(is_using_least_square); PI(is_ell_adaptive);
  PI(is_full_ip_matrix); PI(is_using_geometry); PI(is_using_intensity); PI(is_using_semantics);
  PI(is_using_range_ell); PI(is_using_kdtree); PI(is_exporting_association); PI(is_using_geometric_type);
  PI(multiframe_using_cpu); PI(multiframe_max_iters); PF(multiframe_ell_init); PF(multiframe_ell_min);
  PI(multiframe_iter_per_ell); PF(multiframe_ell_decay_rate); PI(multiframe_iterations_per_ell);
  PI(multiframe_iterations_per_solve); PI(multiframe_expected_points); PF(multiframe_downsample_voxel_size);
  PI(multiframe_num_neighbors); PI(multiframe_least_squares_num_threads); PI(multiframe_min_nonzeros);
  for (const std::string& w : warnings) std::printf("warning=%s\n", w.c_str());
  return 0;
}
