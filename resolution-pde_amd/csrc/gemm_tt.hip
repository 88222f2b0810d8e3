// A x-major, B k-major (rare: kept for completeness of the strided GEMM)
#include "gemm_kernel.h"
namespace rpde {
int launch_tt(const GemmK& g, int bm, int bn, int pro, bool vec, dim3 grid, hipStream_t st) {
  return launch_layout_impl<false, true, 0b001>(g, bm, bn, pro, vec, grid, st);
}
}  // namespace rpde
