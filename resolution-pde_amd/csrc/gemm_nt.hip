// A k-major, B k-major (FeedForward forward, mix backward-data, channels-first DFTs, conv1x1 dW)
#include "gemm_kernel.h"
namespace rpde {
int launch_nt(const GemmK& g, int bm, int bn, int pro, bool vec, dim3 grid, hipStream_t st) {
  return launch_layout_impl<true, true, 0b111>(g, bm, bn, pro, vec, grid, st);
}
}  // namespace rpde
