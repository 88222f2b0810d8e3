// Streaming 1x1 convolution for channels-first tensors with at most 32 output channels (conv_small.hip).
#pragma once
#include "rpde_internal.h"

namespace rpde {
// Cout <= 32, S a multiple of 4, 16-byte aligned tensors; RPDE_CONV_SMALL=0 turns the path off
bool conv1x1_small_ok(const float* x, const float* out, int Cin, int Cout, long S);
int conv1x1_small(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, long S, int act_in,
                  int accumulate, int act_out, hipStream_t st);
// the same pass with the last stage of a channels-first spectral convolution folded in (evaluation-mode FNO block):
// t [B,Cout,M,R2] row spectra, fs_t [R2,N] transposed synthesis table
bool conv1x1_syn_ok(const float* x, const float* out, int Cin, int Cout, int M, int N, int R2);
int conv1x1_syn(const float* x, const float* w, const float* bias, const float* t, const float* fs_t, float* out, int B, int Cin,
                int Cout, int M, int N, int R2, int act_out, hipStream_t st);
// the same on the matrix pipe (conv_syn_h2.hip): N in {64,128,256,512}, Cin = 32, Cout <= 32, R2 a multiple of 8
// up to 32; RPDE_CONV_SYN_H2=0 keeps the fp32 multiply-add kernel
bool conv_syn_h2_ok(const float* x, const float* out, const float* t, int Cin, int Cout, int M, int N, int R2);
// lift_u != null: x is not read; the block's input is lifting(cat(u, gx, gy)) with u [B,1,M,N], lift_w [32,3], lift_b [32],
// gx [M], gy [N] (M <= 1024), formed on the fly
int conv_syn_h2(const float* x, const float* w, const float* bias, const float* t, const float* fs_t, float* out, int B, int Cin,
                int Cout, int M, int N, int R2, int act_out, hipStream_t st, const float* lift_u = nullptr,
                const float* lift_w = nullptr, const float* lift_b = nullptr, const float* gx = nullptr, const float* gy = nullptr);

// last block's tail + projection MLP in one pass (conv_proj_h2.hip): out [B,Cq,M,N]
bool conv_syn_proj_ok(int Cin, int Cout, int M, int N, int R2, int Cmid, int Cq);
int conv_syn_proj(const float* x, const float* wc, const float* bc, const float* t, const float* fs_t, const float* w1,
                  const float* b1, const float* w2, const float* b2, float* out, int B, int Cout, int M, int N, int R2, int Cmid,
                  int Cq, int act, hipStream_t st);

}  // namespace rpde
