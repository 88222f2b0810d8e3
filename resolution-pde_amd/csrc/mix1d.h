// mix1d.hip: the per-mode channel mixing on spectra with few rows (FSpectralConv1d), straight from W[i][o][k][2]
#pragma once
#include <hip/hip_runtime.h>

namespace rpde {
bool mix1d_ok(int rows, int C);
int mix1d(const float* in, const float* w, float* out, int rows, int C, int K, int keff, int kp, bool transpose, hipStream_t st);
int mix1d_wgrad(const float* spec, const float* gspec, float* gw, int rows, int C, int K, int keff, int kp, hipStream_t st);
}  // namespace rpde
