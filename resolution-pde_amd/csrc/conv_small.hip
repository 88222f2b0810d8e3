// 1x1 convolution on channels-first tensors with few output channels (reference: nn.Conv1d/Conv2d(k=1) of the FNO
// family -- lifting models/fno.py:30,93, the blocks' bypass conv models/fno_blocks.py:29,67, the projection's second
// layer :39,77):   out[b][o][s] = act_out( (accumulate ? out[b][o][s] : 0) + bias[o] + sum_i W[o][i] act_in(x[b][i][s]) )
// with Cout <= 32.  As a GEMM this is an [Cout x Cin] . [Cin x S] product per sample whose operands total a few KB of
// weights against hundreds of MB of field: pure streaming.  A thread owns four consecutive grid points of one sample and
// all Cout outputs (4 * Cout accumulators); it walks the input channels once -- every load is 16 B per lane, 1 KB
// contiguous per wave -- with the weights read from LDS as broadcast 16-byte pieces.  The field is read once and the
// output written once, at the HBM rate, where the generic GEMM path (activation staged into its k-loop, fp32 MFMA)
// reaches about half of it.
#include "rpde_internal.h"
#include "conv_small.h"

#include <stdlib.h>

namespace rpde {

// the last stage of a channels-first spectral convolution folded into the same pass (evaluation-mode FNO block,
// reference models/fno_blocks.py:63-83: activation(spectral_conv(x) + bypass_conv(x))): instead of reading the spectral
// branch back from memory (accumulate) the kernel forms it -- out += sum_r Fs[n][r] t[b][o][m][r], the inverse real DFT
// along the last axis of the row spectra t [B,Cout,M,R2] -- with the same fp32 multiply-adds that do the channel mix: R2
// more "input channels" whose weights depend on the row.  The spectral branch never crosses HBM (2 x 33.5 MB per sample
// and block at 512^2, width 32).
struct ConvSyn {
  const float* t;      // row spectra [B][Cout][M][R2]
  const float* fs_t;   // synthesis table transposed [R2][N]
  int M, N, R2;
};

// acc (four points) += w * v as two packed fp32 multiply-adds (v_pk_fma_f32: two per lane and instruction -- this kernel
// has no matrix work that packed fp32 would get in the way of, and with the synthesis folded in it is bound by these)
typedef float f32x2v __attribute__((ext_vector_type(2)));
struct Acc4 { f32x2v lo, hi; };
__device__ __forceinline__ void fma4(Acc4& a, float w, const float4& v) {
  const f32x2v ww = {w, w};
  a.lo = __builtin_elementwise_fma(ww, (f32x2v){v.x, v.y}, a.lo);
  a.hi = __builtin_elementwise_fma(ww, (f32x2v){v.z, v.w}, a.hi);
}

template <int CO, bool SYN>      // outputs padded to CO in {4, 8, 16, 32}
__global__ __launch_bounds__(256) void k_conv1x1_small(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, int Cin,
                                                       int Cout, long S, int act_in, int accumulate, int act_out, ConvSyn Y) {
  extern __shared__ float wt[];                  // [Cin][CO]: for one input channel the CO weights are contiguous
  for (int e = threadIdx.x; e < Cin * CO; e += 256) {
    const int i = e / CO, o = e % CO;
    wt[e] = o < Cout ? w[o * Cin + i] : 0.f;
  }
  const int b = blockIdx.y;
  float* const ts = wt + Cin * CO;               // SYN: [rows of this block][R2][CO]
  if (SYN) {
    // a block covers 1024 consecutive points of a sample: 1024 / N whole rows (N <= 1024) or a piece of one row
    const int rows = Y.N >= 1024 ? 1 : 1024 / Y.N;
    const long m0 = ((long)blockIdx.x * 1024) / Y.N;
    for (int e = threadIdx.x; e < rows * Y.R2 * CO; e += 256) {
      const int o = e % CO, r = (e / CO) % Y.R2, row = e / (CO * Y.R2);
      ts[e] = (o < Cout && m0 + row < Y.M) ? Y.t[(((long)b * Cout + o) * Y.M + m0 + row) * Y.R2 + r] : 0.f;
    }
  }
  __syncthreads();
  const long s4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (s4 >= S) return;
  const float* __restrict__ xb = x + (long)b * Cin * S + s4;
  float* __restrict__ ob = out + (long)b * Cout * S + s4;
  Acc4 acc[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    const float bo = (bias && o < Cout) ? bias[o] : 0.f;
    acc[o].lo = (f32x2v){bo, bo}; acc[o].hi = (f32x2v){bo, bo};
    if (accumulate && o < Cout) {
      const float4 p = *reinterpret_cast<const float4*>(ob + (long)o * S);
      acc[o].lo += (f32x2v){p.x, p.y}; acc[o].hi += (f32x2v){p.z, p.w};
    }
  }
  // (unrolling this loop for more loads in flight per thread was measured slower: 256 VGPRs, one wave per SIMD fewer)
  for (int i = 0; i < Cin; ++i) {
    float4 v = *reinterpret_cast<const float4*>(xb + (long)i * S);
    if (act_in) { v.x = act_f(act_in, v.x); v.y = act_f(act_in, v.y); v.z = act_f(act_in, v.z); v.w = act_f(act_in, v.w); }
    const float4* __restrict__ wr = reinterpret_cast<const float4*>(wt + i * CO);
#pragma unroll
    for (int q = 0; q < CO / 4; ++q) {
      const float4 ww = wr[q];
      fma4(acc[4 * q], ww.x, v); fma4(acc[4 * q + 1], ww.y, v); fma4(acc[4 * q + 2], ww.z, v); fma4(acc[4 * q + 3], ww.w, v);
    }
  }
  if (SYN) {
    const int n4 = (int)(s4 % Y.N);
    const int row = Y.N >= 1024 ? 0 : (threadIdx.x * 4) / Y.N;
    for (int r = 0; r < Y.R2; ++r) {
      const float4 v = *reinterpret_cast<const float4*>(Y.fs_t + (long)r * Y.N + n4);
      const float4* __restrict__ wr = reinterpret_cast<const float4*>(ts + (row * Y.R2 + r) * CO);
#pragma unroll
      for (int q = 0; q < CO / 4; ++q) {
        const float4 ww = wr[q];
        fma4(acc[4 * q], ww.x, v); fma4(acc[4 * q + 1], ww.y, v); fma4(acc[4 * q + 2], ww.z, v); fma4(acc[4 * q + 3], ww.w, v);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    if (o < Cout) {
      float4 a = make_float4(acc[o].lo.x, acc[o].lo.y, acc[o].hi.x, acc[o].hi.y);
      if (act_out) { a.x = act_f(act_out, a.x); a.y = act_f(act_out, a.y); a.z = act_f(act_out, a.z); a.w = act_f(act_out, a.w); }
      *reinterpret_cast<float4*>(ob + (long)o * S) = a;
    }
  }
}

bool conv1x1_small_ok(const float* x, const float* out, int Cin, int Cout, long S) {
  const char* e = getenv("RPDE_CONV_SMALL");
  if (e && e[0] == '0') return false;
  return Cout >= 1 && Cout <= 32 && Cin >= 1 && Cin <= 512 && S % 4 == 0 &&
         ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
}

int conv1x1_small(const float* x, const float* w, const float* bias, float* out, int B, int Cin, int Cout, long S, int act_in,
                  int accumulate, int act_out, hipStream_t st) {
  const int CO = Cout <= 4 ? 4 : (Cout <= 8 ? 8 : (Cout <= 16 ? 16 : 32));
  const dim3 grid((unsigned)((S / 4 + 255) / 256), B), block(256);
  const size_t lds = sizeof(float) * (size_t)Cin * CO;
  ConvSyn Y{nullptr, nullptr, 0, 0, 0};
  switch (CO) {
    case 4: hipLaunchKernelGGL((k_conv1x1_small<4, false>), grid, block, lds, st, x, w, bias, out, Cin, Cout, S, act_in, accumulate, act_out, Y); break;
    case 8: hipLaunchKernelGGL((k_conv1x1_small<8, false>), grid, block, lds, st, x, w, bias, out, Cin, Cout, S, act_in, accumulate, act_out, Y); break;
    case 16: hipLaunchKernelGGL((k_conv1x1_small<16, false>), grid, block, lds, st, x, w, bias, out, Cin, Cout, S, act_in, accumulate, act_out, Y); break;
    default: hipLaunchKernelGGL((k_conv1x1_small<32, false>), grid, block, lds, st, x, w, bias, out, Cin, Cout, S, act_in, accumulate, act_out, Y); break;
  }
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// rows of N points: N a multiple of 4 that divides 1024 or is a multiple of it (a block's 1024 points are whole rows or
// lie inside one)
// and the block's weights + its rows' spectra fit the 64 KB of dynamic LDS a launch gets without opting in to more
// (narrow grids keep many rows per block: width 32, 12 modes at N = 32 would need 100 KB -- such shapes take the
// two-step path)
static size_t conv1x1_syn_lds(int Cin, int Cout, int N, int R2) {
  const int CO = Cout <= 4 ? 4 : (Cout <= 8 ? 8 : (Cout <= 16 ? 16 : 32));
  const int rows = N >= 1024 ? 1 : 1024 / N;
  return sizeof(float) * ((size_t)Cin * CO + (size_t)rows * R2 * CO);
}
bool conv1x1_syn_ok(const float* x, const float* out, int Cin, int Cout, int M, int N, int R2) {
  return conv1x1_small_ok(x, out, Cin, Cout, (long)M * N) && N % 4 == 0 && (1024 % N == 0 || N % 1024 == 0) &&
         conv1x1_syn_lds(Cin, Cout, N, R2) <= 64 * 1024;
}

// out[b][o][m][n] = act_out(bias[o] + sum_i W[o][i] x[b][i][m][n] + sum_r fs_t[r][n] t[b][o][m][r])
int conv1x1_syn(const float* x, const float* w, const float* bias, const float* t, const float* fs_t, float* out, int B, int Cin,
                int Cout, int M, int N, int R2, int act_out, hipStream_t st) {
  const long S = (long)M * N;
  const int CO = Cout <= 4 ? 4 : (Cout <= 8 ? 8 : (Cout <= 16 ? 16 : 32));
  const dim3 grid((unsigned)((S / 4 + 255) / 256), B), block(256);
  const size_t lds = conv1x1_syn_lds(Cin, Cout, N, R2);
  ConvSyn Y{t, fs_t, M, N, R2};
  switch (CO) {
    case 4: hipLaunchKernelGGL((k_conv1x1_small<4, true>), grid, block, lds, st, x, w, bias, out, Cin, Cout, S, 0, 0, act_out, Y); break;
    case 8: hipLaunchKernelGGL((k_conv1x1_small<8, true>), grid, block, lds, st, x, w, bias, out, Cin, Cout, S, 0, 0, act_out, Y); break;
    case 16: hipLaunchKernelGGL((k_conv1x1_small<16, true>), grid, block, lds, st, x, w, bias, out, Cin, Cout, S, 0, 0, act_out, Y); break;
    default: hipLaunchKernelGGL((k_conv1x1_small<32, true>), grid, block, lds, st, x, w, bias, out, Cin, Cout, S, 0, 0, act_out, Y); break;
  }
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // namespace rpde
