// Streaming kernels for linear layers with at most four features on one side (thin_linear.hip).
#pragma once
#include "rpde_internal.h"

namespace rpde {

// thin in 1..4, wide a multiple of 4 up to 256 whose quarter divides 256; RPDE_THIN_LINEAR=0 turns the path off
bool thin_linear_ok(int thin, int wide);
// y[p][o] = bias[o] + sum_t a[p][t] * w[o * ws_o + t * ws_t]        a [P,T], y [P,O]
int thin_expand(const float* a, const float* w, long ws_o, long ws_t, const float* bias, float* y, long P, int T, int O,
                hipStream_t st);
// y[p][t] = bias[t] + sum_o a[p][o] * w[t * ws_t + o * ws_o]        a [P,O], y [P,T]
int thin_contract(const float* a, const float* w, long ws_t, long ws_o, const float* bias, float* y, long P, int T, int O,
                  hipStream_t st);
// G[t][o] = sum_p a[p][t] b[p][o] (stored [T][O], or [O][T] when g_transposed), cs[o] = sum_p b[p][o], ts[t] = sum_p a[p][t];
// null outputs are skipped; ws: thin_outer_ws_floats(P, T, O) floats
size_t thin_outer_ws_floats(long P, int T, int O);
int thin_outer(const float* a, const float* b, float* G, int g_transposed, float* cs, float* ts, long P, int T, int O, float* ws,
               hipStream_t st);

}  // namespace rpde
