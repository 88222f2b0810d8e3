// A k-major, B x-major (backward-data, channels-last DFTs, mix forward, channels-first conv1x1)
#include "gemm_kernel.h"
namespace rpde {
int launch_nn(const GemmK& g, int bm, int bn, int pro, bool vec, dim3 grid, hipStream_t st) {
  return launch_layout_impl<true, false, 0b101>(g, bm, bn, pro, vec, grid, st);
}
}  // namespace rpde
