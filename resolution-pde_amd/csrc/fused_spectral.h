// Fused FSpectralConv2d kernels (fused_spectral.hip): entry points used by fspectral.hip and core.hip.
#pragma once
#include "rpde_internal.h"
#include "plan.h"

namespace rpde {

// true when the fused path covers the problem (C = 64, grid sides multiples of 32 up to 256, <= 24 padded modes,
// same mode count on both axes); everything else keeps the GEMM-per-step path of fspectral.hip
bool fused2d_ok(int M, int N, int C, int keff_y, int keff_x);
int h2_build_tables(rpde_plan* p, hipStream_t st);

// ---- operand blocks of the synthesis MFMAs (layout: fused_spectral.hip, "operand blocks") ----
// R = 32 K32 + 8 TG reduction rows -> K32 hi + K32 lo fragments + NP packed tail fragments, 1 KB each
__host__ __device__ constexpr int h2_np(int TG) { return (3 * TG + 3) / 4; }
__host__ __device__ constexpr int h2_block_bytes(int K32, int TG) { return (2 * K32 + h2_np(TG)) * 1024; }

// B-fragment image of `lines` spectra of R rows x 64 channels
size_t fused2d_img_bytes(long lines, int R);
// spec_y[(b,m)][R][64], spec_x[(b,n)][R][64] = table . lines of x;  adjoint: tables Fs^T instead of Fa
// amax_y / amax_x (may be null): max |spectrum| of every line, for the mode-mix kernel's scaling (fused_mix.hip)
int fused2d_analysis(const float* x, float* spec_y, float* spec_x, float* amax_y, float* amax_x, const rpde_plan* py,
                     const rpde_plan* px, int adjoint, int B, int M, int N, hipStream_t st);
int fused2d_split(const float* spec, void* img, float* inv, long lines, int R, hipStream_t st);
// out = table_y . img_y[row] + table_x . img_x[col] (+ skip);  adjoint: tables Fa^T instead of Fs
int fused2d_synthesis(const void* imgy, const void* imgx, const float* invy, const float* invx, const rpde_plan* py,
                      const rpde_plan* px, int adjoint, float* out, const float* skip, int B, int M, int N, hipStream_t st);

// ---- mode mix in h2 arithmetic (fused_mix.hip) ----
constexpr int MIX_W_BYTES_PER_MODE = 4 * 2 * 2 * 2 * 1024;      // [cb 4][Wr|Wi][ks 2][hi|lo] fragments of 1 KB
size_t mix_wimg_bytes(int kp);
// weights of both axes -> fragment image + wc[2][kp][2] = {1 / scale, norm}; conj_t: W^H (the adjoint's operand)
int mix_prep(const float* w_y, const float* w_x, int K, int keff, int kp, int conj_t, void* wimg, float* wc, hipStream_t st);
// weight gradients of the mix, both axes: two launches (slabs of lines, then a fixed-order fold); amax_*: line maxima of the
// saved spectra / of the gradient spectra; slabs: mix_wgrad_slab_floats(kp, S) floats of workspace
size_t mix_wgrad_slab_floats(int kp, int S);
int mix_wgrad_h2(const float* spec_y, const float* spec_x, const float* gspec_y, const float* gspec_x, const float* amax_sy,
                 const float* amax_sx, const float* amax_gy, const float* amax_gx, float* gw_y, float* gw_x, long lines_y,
                 long lines_x, int K, int keff, int kp, float* slabs, int S, hipStream_t st);
int mix_h2(const float* spec_y, const float* spec_x, const float* amax_y, const float* amax_x, void* img_y, void* img_x,
           float* inv_y, float* inv_x, long lines_y, long lines_x, int kp, const void* wimg, const float* wc, hipStream_t st);

}  // namespace rpde
