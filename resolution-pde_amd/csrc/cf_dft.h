// Streaming truncated DFT along the contiguous axis of channels-first tensors (cf_dft.hip)
#pragma once
#include "rpde_internal.h"
#include "plan.h"

namespace rpde {
// planar real plans with n % 128 == 0, 2 kp <= 32 and a table that fits 64 KB of LDS
bool cf_h2_eligible(int n, int R);
// ... and, for the synthesis kernel, n <= 512 (its table has n / 16 fragments)
bool cf_h2_syn_eligible(int n, int R);
int cf_build_tables(rpde_plan* p, hipStream_t st);
// spec[rows, 2kp] = alpha * x[rows, n] . T^T          adjoint: T = Fs^T (adjoint of the synthesis) instead of Fa
int cf_analysis_h2(const rpde_plan* pl, int adjoint, const float* x, float* spec, long rows, float alpha, hipStream_t st);
// out[rows, n] = alpha * spec[rows, 2kp] . S^T        adjoint: S = Fa^T (adjoint of the analysis) instead of Fs
int cf_synthesis_h2(const rpde_plan* pl, int adjoint, const float* spec, float* out, long rows, float alpha, hipStream_t st);
}  // namespace rpde
