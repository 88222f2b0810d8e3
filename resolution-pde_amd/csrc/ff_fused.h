// Fused three-layer FeedForward (ff_fused.hip)
#pragma once
#include "rpde_internal.h"

namespace rpde {
// true for the FFNO2D shape (dim 64, factor 4, three layers) and at least one tile of points
bool ff3_fused_ok(const rpde_ff_params* p, long P);
size_t ff3_fused_ws_floats();
// hs and ds with all four buffers: training, h1, d1, h2, d2 and z_last are stored for rpde_feedforward_bwd;
// hs only (ds null): training in recompute mode, hs receive u = dropout(z) of the hidden layers and the backward
// re-evaluates gelu / gelu' from them; neither: evaluation, only `out` is written
// prepared: ws already holds ff3_fused_prepare's output for these weights (evaluation with frozen weights)
int ff3_fused_prepare(const rpde_ff_params* p, void* ws, hipStream_t st);
int ff3_fused_fwd(const rpde_ff_params* p, const float* x, const float* residual, float* const* hs, float* const* ds,
                  float* z_last, float* out, long P, void* ws, hipStream_t st, bool prepared = false);
// backward: everything except the three weight-gradient GEMMs (feedforward.hip runs those on dz3 / du2 / du1)
constexpr int FF3_PART = 704;            // per workgroup: db1[256] db2[256] db3[64] dgamma[64] dbeta[64]
size_t ff3_fused_bwd_part_floats();
// ds: the saved derivative factors, or (recompute != 0) the saved u of forward mode 2
int ff3_fused_bwd_launch(const rpde_ff_params* p, const float* const* ds, int recompute, const float* z_last,
                         const float* grad_out, float* dz3, float* du2, float* du1, float* dx, float* part, int* grid_out,
                         long P, void* ws, hipStream_t st);
}  // namespace rpde
