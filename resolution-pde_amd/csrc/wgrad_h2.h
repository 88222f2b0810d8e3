// Streaming weight-gradient kernel of the headline FeedForward shapes (wgrad_h2.hip).
#pragma once
#include "rpde_internal.h"
#include "pointwise.h"

namespace rpde {

// shapes with a kernel instance: (out, in) in {(256,256), (256,64), (64,256)}, P >= 8192;
// RPDE_WGRAD_H2=0 turns the path off (the generic split-bf16 GEMM takes over)
bool wgrad_h2_ok(long P, int out_f, int in_f);
size_t wgrad_h2_slab_floats(long P, int out_f, int in_f);
// gw[out, in] = sum_p gy[p, out] * act_b(h[p, in]);  act_b: RPDE_ACT_IDENTITY or RPDE_ACT_GELU;  slabs: scratch
// defer: the slab fold is left to the caller's FoldJobs launch (slabs must then stay untouched until it has run)
int wgrad_h2(const float* gy, const float* h, float* gw, long P, int in_f, int out_f, int act_b, float* slabs, hipStream_t st,
             FoldJobs* defer = nullptr);

// the same with the layer's data gradient riding along (out 256, in 64: the first FeedForward layer):
// gw as above (act_b = identity) and gx[P, in] = gy[P, out] . w[out, in]; gy is read from HBM once for both
bool wgrad_h2_dgrad_ok(long P, int out_f, int in_f);
int wgrad_h2_dgrad(const float* gy, const float* h, const float* w, float* gw, float* gx, long P, int in_f, int out_f,
                   float* slabs, hipStream_t st, FoldJobs* defer = nullptr);

}  // namespace rpde
