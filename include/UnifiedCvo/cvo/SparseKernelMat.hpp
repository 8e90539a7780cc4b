// cvo::SparseKernelMat, host side: the ELL kernel matrix the multi-frame edge state hands to its solver
// (upstream include/UnifiedCvo/cvo/SparseKernelMat.hpp:11-19; only the *_cpu helpers exist here, the device copy
// lives behind the C-ABI).  mat / ind_row2col are row-major [rows x cols]; a row's unused slots hold 0 / -1.
#pragma once

namespace cvo {

struct SparseKernelMat {
  int rows = 0;
  int cols = 0;
  unsigned int nonzero_sum = 0;
  float* mat = nullptr;
  int* ind_row2col = nullptr;
  unsigned int* nonzeros = nullptr;
};

// upstream SparseKernelMat.cu: init_internal_SparseKernelMat_cpu / delete_internal_SparseKernelMat_cpu /
// clear_SparseKernelMat_cpu
void init_internal_SparseKernelMat_cpu(int rows, int cols, SparseKernelMat* A_cpu);
void delete_internal_SparseKernelMat_cpu(SparseKernelMat* A_cpu);
void clear_SparseKernelMat_cpu(SparseKernelMat* A_cpu, int num_neighbors);
unsigned int nonzeros(SparseKernelMat* A_host);
unsigned int max_neighbors(SparseKernelMat* A_host);

}  // namespace cvo
