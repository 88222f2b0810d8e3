// DFT plan: small device tables owned by the library.
#pragma once
#include "rpde_internal.h"

struct rpde_plan {
  int device;
  int n, modes, norm;
  int planar;   // real plans: 0 rows/cols ordered 2k+ri, 1 ordered ri*kp+k
  int kind;     // PLAN_REAL or PLAN_CPLX
  int kp;       // real: modes rounded up to 4;  complex: R = top + bot row slots
  int ldn;      // leading dim of fa: real: n rounded up to 4;  complex: 2n
  float* fa;    // real: [2kp, ldn] analysis;     complex: [2R, 2n]
  float* fs;    // real: [n, 2kp] synthesis;      complex: [2n, 2R]
  float* fs_t;  // fs transposed: real planar plans [2kp, n] (read along n by conv_small.hip / conv_syn_h2.hip),
                // complex plans [2R, 2n] (read along the output index by spectral_cf.hip's column stage)
  // real, interleaved plans: the tables pre-split for the split-bf16 GEMM (gemm_bf16x3.hip), as A operands:
  // [IMG_FA] Fa (2kp x n), [IMG_FST] Fs^T (2kp x n), [IMG_FS] Fs (n x 2kp), [IMG_FAT] Fa^T (n x 2kp)
  void* img[4];
  // real, interleaved plans with n % 32 == 0, n <= 256 and <= 24 padded modes: the tables as ready f16 hi/lo MFMA
  // fragments for the fused kernels (fused_spectral.hip): [0] forward operand (Fa / Fs), [1] adjoint (Fs^T / Fa^T)
  void* h2_ana[2];
  void* h2_ana_p[2];   // the same with the reduction index permuted inside each 32-chunk (slot 8 g + j <-> point 4 j + g):
                       // k_dft_analysis_sq_h2 builds its B fragments straight from the loaded registers
  void* h2_syn[2];
  // real, planar plans (channels-first layers, resizers) that cf_dft.hip covers: the tables as B fragments
  void* cf_ana[2];
  void* cf_syn[2];
};

namespace rpde {
enum { PLAN_REAL = 0, PLAN_CPLX = 1 };
enum { IMG_FA = 0, IMG_FST = 1, IMG_FS = 2, IMG_FAT = 3 };
// cached per (device, n, modes, norm, planar, kind); never freed
// PLAN_CPLX: `modes` = rows kept from the top of the spectrum, `bot` = rows kept from the bottom (-1: same)
int get_plan(const rpde_plan** out, int n, int modes, int norm, int planar, int kind, hipStream_t st, int bot = -1);
}  // namespace rpde
