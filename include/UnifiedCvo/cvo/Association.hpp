// Output of the data association (upstream include/UnifiedCvo/cvo/Association.hpp:6-10).
#pragma once
#include <vector>

#include "utils/data_type.hpp"

namespace cvo {

struct Association {
  std::vector<int> source_inliers;  // rows with at least one associated target
  std::vector<int> target_inliers;  // column index of every stored pair, row by row
  SparseRowMat pairs;               // a_ij, row-major sparse (source index x target index)
};

}  // namespace cvo
