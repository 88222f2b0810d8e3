// Projection MLP of the FNO family in ONE pass, evaluation only (reference: models/fno_blocks.py:38-45, 76-83:
// mlp2(gelu(mlp1(x))) with 1x1 convolutions on channels-first tensors):
//   out[b][o][s] = b2[o] + sum_m W2[o][m] gelu( b1[m] + sum_i W1[m][i] act_in(x[b][i][s]) )
// As two convolutions the 128-wide hidden tensor crosses HBM twice (FNO2d at 512^2, B = 16: 2.1 GB written, 2.1 GB read
// back -- 1.03 ms of a 4.4 ms forward).  Here a wave takes 16 grid points at a time: their Cin inputs become the B
// operand of the first product (h2 arithmetic, W1 as resident A fragments: rows = hidden features), bias and GELU are
// applied to the accumulators, and the second product -- Cout <= 4 outputs -- is a per-lane multiply-add over the
// lane's hidden features followed by a sum over the four lane groups.  Nothing but x is read and nothing but out is
// written.  Covers Cin in {32, 64}, hidden <= 128 (Cin = 32) or <= 64 (Cin = 64), Cout <= 4, S a multiple of 16; the
// training path (which needs the hidden tensor for backward) and other shapes keep the two-convolution sequence.
#include "h2.h"

#include <stdlib.h>

namespace rpde {

template <int KS, int MT, int CO>
__global__ __launch_bounds__(256) void k_conv_mlp_h2(const float* __restrict__ x, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, float* __restrict__ out, int Cmid, int Cout,
                                                     long S, long tiles_per_sample, long tiles, int act_in) {
  constexpr int CIN = 32 * KS;
  __shared__ float red[4];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, g = l >> 4, li = l & 15;
  // ---- W1: one power-of-two scale for the matrix, resident A fragments: rows m = 16 mt + li, k = 32 ks + 8g + j ----
  float m = 0.f;
  for (int e = tid; e < Cmid * CIN; e += 256) m = fmaxf(m, fabsf(w1[e]));
  m = wave_max(m);
  if (l == 0) red[wv] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float wsc, winv;
  h2_scale(m, 0, wsc, winv);
  f16x8 wh[MT][KS], wl[MT][KS];
  float4 bias1[MT], wo[CO][MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = 16 * mt + li;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = row < Cmid ? w1[row * CIN + 32 * ks + 8 * g + j] * wsc : 0.f;
      union { f16x8 v; struct { uint2 a, b; } u; } H, L;
      h2_split4(v[0], v[1], v[2], v[3], H.u.a, L.u.a);
      h2_split4(v[4], v[5], v[6], v[7], H.u.b, L.u.b);
      wh[mt][ks] = H.v; wl[mt][ks] = L.v;
    }
    // accumulator rows of this lane: hidden features 16 mt + 4g + j
    float bb[4], ww[CO][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = 16 * mt + 4 * g + j;
      bb[j] = (b1 && f < Cmid) ? b1[f] : 0.f;
#pragma unroll
      for (int o = 0; o < CO; ++o) ww[o][j] = (o < Cout && f < Cmid) ? w2[o * Cmid + f] : 0.f;
    }
    bias1[mt] = make_float4(bb[0], bb[1], bb[2], bb[3]);
#pragma unroll
    for (int o = 0; o < CO; ++o) wo[o][mt] = make_float4(ww[o][0], ww[o][1], ww[o][2], ww[o][3]);
  }
  float bias2[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) bias2[o] = (b2 && o < Cout) ? b2[o] : 0.f;

  // ---- tiles of 16 points: lane (g, li) fetches channels 32 ks + 8g + j of point li, one tile ahead ----
  const long stride = (long)gridDim.x * 4;
  float xv[KS][8];
  auto fetch = [&](long t) {
    if (t >= tiles) return;
    const long b = t / tiles_per_sample, s = (t - b * tiles_per_sample) * 16 + li;
    const float* p = x + (b * CIN + 8 * g) * S + s;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[ks][j] = p[(long)(32 * ks + j) * S];
  };
  long t = (long)blockIdx.x * 4 + wv;
  fetch(t);
  for (; t < tiles; t += stride) {
    float v[KS][8];
    float mx = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[ks][j] = act_in ? act_f(act_in, xv[ks][j]) : xv[ks][j];
        mx = fmaxf(mx, fabsf(v[ks][j]));
      }
    fetch(t + stride);
    mx = wave_max(mx);
    float sc, iv;
    h2_scale(mx, 0, sc, iv);
    const float inv = iv * winv;
    f16x8 bh[KS], bl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      union { f16x8 v; struct { uint2 a, b; } u; } H, L;
      h2_split4(v[ks][0] * sc, v[ks][1] * sc, v[ks][2] * sc, v[ks][3] * sc, H.u.a, L.u.a);
      h2_split4(v[ks][4] * sc, v[ks][5] * sc, v[ks][6] * sc, v[ks][7] * sc, H.u.b, L.u.b);
      bh[ks] = H.v; bl[ks] = L.v;
    }
    float part[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) part[o] = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      f32x4v c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) c = h2_mfma32(wh[mt][ks], wl[mt][ks], bh[ks], bl[ks], c);
      const float bb[4] = {bias1[mt].x, bias1[mt].y, bias1[mt].z, bias1[mt].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float h = gelu_f(fmaf(c[j], inv, bb[j]));
#pragma unroll
        for (int o = 0; o < CO; ++o) {
          const float wv4[4] = {wo[o][mt].x, wo[o][mt].y, wo[o][mt].z, wo[o][mt].w};
          part[o] = fmaf(wv4[j], h, part[o]);
        }
      }
    }
    const long b = t / tiles_per_sample, s = (t - b * tiles_per_sample) * 16 + li;
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      float r = part[o];
      r += lane_xor16(r);
      r += lane_xor32(r);
      if (g == 0 && o < Cout) out[(b * Cout + o) * S + s] = r + bias2[o];
    }
  }
}

bool conv_mlp_ok(int Cin, int Cmid, int Cout, long S) {
  if (const char* e = getenv("RPDE_CONV_MLP")) if (e[0] == '0') return false;
  if (Cout < 1 || Cout > 4 || S % 16 != 0 || Cmid < 1) return false;
  if (Cin == 32) return Cmid <= 128;
  if (Cin == 64) return Cmid <= 64;
  return false;
}

template <int KS, int MT>
static void conv_mlp_launch(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* out, int B,
                            int Cmid, int Cout, long S, int act_in, int grid, hipStream_t st) {
  const long tps = S / 16, tiles = tps * B;
  if (Cout == 1)
    hipLaunchKernelGGL((k_conv_mlp_h2<KS, MT, 1>), dim3(grid), dim3(256), 0, st, x, w1, b1, w2, b2, out, Cmid, Cout, S, tps, tiles, act_in);
  else if (Cout == 2)
    hipLaunchKernelGGL((k_conv_mlp_h2<KS, MT, 2>), dim3(grid), dim3(256), 0, st, x, w1, b1, w2, b2, out, Cmid, Cout, S, tps, tiles, act_in);
  else
    hipLaunchKernelGGL((k_conv_mlp_h2<KS, MT, 4>), dim3(grid), dim3(256), 0, st, x, w1, b1, w2, b2, out, Cmid, Cout, S, tps, tiles, act_in);
}

}  // namespace rpde

using namespace rpde;

extern "C" {

int rpde_conv_mlp_ok(int Cin, int Cmid, int Cout, int64_t S) { return conv_mlp_ok(Cin, Cmid, Cout, (long)S) ? 1 : 0; }

int rpde_conv_mlp_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* out, int B,
                      int Cin, int Cmid, int Cout, int64_t S, int act_in, void* stream) {
  RPDE_CHECK_ARG(x && w1 && w2 && out && B > 0 && S > 0, "conv_mlp_fwd: bad arguments");
  RPDE_CHECK_ARG(conv_mlp_ok(Cin, Cmid, Cout, (long)S), "conv_mlp_fwd: unsupported shape %d -> %d -> %d on %ld points", Cin, Cmid,
                 Cout, (long)S);
  hipStream_t st = as_stream(stream);
  int dev = 0, cus = 256;
  RPDE_HIP(hipGetDevice(&dev));
  RPDE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const long tiles = (long)B * (S / 16);
  long grid = (tiles + 3) / 4;
  if (grid > 8L * cus) grid = 8L * cus;
  const int MT = (Cmid + 15) / 16;
  // (MT rounded up to the instantiated sizes; rows beyond Cmid are zero fragments with zero second-layer weights)
  if (Cin == 32) {
    if (MT <= 2) conv_mlp_launch<1, 2>(x, w1, b1, w2, b2, out, B, Cmid, Cout, S, act_in, (int)grid, st);
    else if (MT <= 4) conv_mlp_launch<1, 4>(x, w1, b1, w2, b2, out, B, Cmid, Cout, S, act_in, (int)grid, st);
    else conv_mlp_launch<1, 8>(x, w1, b1, w2, b2, out, B, Cmid, Cout, S, act_in, (int)grid, st);
  } else {
    if (MT <= 2) conv_mlp_launch<2, 2>(x, w1, b1, w2, b2, out, B, Cmid, Cout, S, act_in, (int)grid, st);
    else conv_mlp_launch<2, 4>(x, w1, b1, w2, b2, out, B, Cmid, Cout, S, act_in, (int)grid, st);
  }
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // extern "C"
