// Fused FSpectralConv2d kernels (fused_spectral.hip): entry points used by fspectral.hip and core.hip.
#pragma once
#include "rpde_internal.h"
#include "plan.h"

namespace rpde {

// true when the fused path covers the problem (C = 64, grid sides multiples of 32 up to 256, <= 24 padded modes,
// same mode count on both axes); everything else keeps the GEMM-per-step path of fspectral.hip
bool fused2d_ok(int M, int N, int C, int keff_y, int keff_x);
int h2_build_tables(rpde_plan* p, hipStream_t st);

// B-fragment image of `lines` spectra of R rows x 64 channels
size_t fused2d_img_bytes(long lines, int R);
// spec_y[(b,m)][R][64], spec_x[(b,n)][R][64] = table . lines of x;  adjoint: tables Fs^T instead of Fa
int fused2d_analysis(const float* x, float* spec_y, float* spec_x, const rpde_plan* py, const rpde_plan* px, int adjoint,
                     int B, int M, int N, hipStream_t st);
int fused2d_split(const float* spec, void* img, float* inv, long lines, int R, hipStream_t st);
// out = table_y . img_y[row] + table_x . img_x[col] (+ skip);  adjoint: tables Fa^T instead of Fs
int fused2d_synthesis(const void* imgy, const void* imgx, const float* invy, const float* invx, const rpde_plan* py,
                      const rpde_plan* px, int adjoint, float* out, const float* skip, int B, int M, int N, hipStream_t st);

}  // namespace rpde
