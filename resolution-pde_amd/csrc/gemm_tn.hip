// A x-major, B x-major (weight gradients with split-K over the grid points, adjoint DFTs)
#include "gemm_kernel.h"
namespace rpde {
int launch_tn(const GemmK& g, int bm, int bn, int pro, bool vec, dim3 grid, hipStream_t st) {
  return launch_layout_impl<false, false, 0b101>(g, bm, bn, pro, vec, grid, st);
}
}  // namespace rpde
