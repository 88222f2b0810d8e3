// Per-mode complex channel mixing of FSpectralConv1d (models/spectral_convolution.py:190-198: the einsum
// "bix,iox->box" on the retained modes) when the spectra have FEW rows -- the 1-D layer has one row per sample, so per
// mode the product is [B x 2C] . [2C x 2C]: the weights (C*C*K complex numbers, 8.4 MB at width 128 / 64 modes) are the
// whole traffic and every one of them is used B times.  The general path (fspectral.hip: repack the weights into real
// [k][2C][2C] block matrices, then a batched GEMM with B rows per batch entry) writes and re-reads 2x the weights to
// feed tiles that are 3/4 empty; these kernels read W[i][o][k][re|im] where it lies, once:
//
//   k_mix1d<JW, false>   out[r][k][:][o] = sum_i in[r][k][:][i] * W[i][o][k]           forward
//   k_mix1d<JW, true>    out[r][k][:][i] = sum_o in[r][k][:][o] * conj(W[i][o][k])     data gradient
//   k_mix1d_wgrad<JW>    gW[i][o][k]     = sum_r conj(x[r][k][:][i]) * g[r][k][:][o]   weight gradient (k >= keff: 0)
//
// Spectra are [row][k][re|im][c] (fspectral.hip).  A workgroup owns 8 modes x 8 own-channels (lane = 8 * channel + mode:
// eight lanes read 64 contiguous bytes of W) and 16 rows (32 accumulators per lane); its eight waves split the reduced
// channel index (JW = C / 8 each) and their partial sums are combined through LDS in fixed order.  The 16 x 8 spectrum
// lines of the tile (2C floats each) are staged in LDS once, padded by 4 floats per line so that the eight modes of a
// wave's 16-byte reads fall in different banks.  fp32 FMAs: 134 MFLOP per pass at B = 16, nothing for the chip.
#include "rpde_internal.h"
#include "mix1d.h"

#include <stdlib.h>

namespace rpde {

constexpr int MX_R = 16, MX_K = 8, MX_Q = 8, MX_WAVES = 8;

struct Mix1dP {
  const float* in;    // [rows][kp][2][C]
  const float* w;     // [C][C][K][2]
  float* out;         // [rows][kp][2][C]
  const float* g;     // weight gradient only: [rows][kp][2][C]
  float* gw;          // weight gradient only: [C][C][K][2]
  int rows, C, K, keff, kp;
};

// the staged lines; the forward / data-gradient kernels reuse the area for the waves' partial sums (64 KB)
static inline size_t mix1d_lds(int C) {
  const size_t lines = (size_t)MX_R * MX_K * (2 * C + 4) * sizeof(float), red = (size_t)MX_WAVES * 2 * MX_R * 64 * sizeof(float);
  return lines > red ? lines : red;
}

// the tile's spectrum lines -> LDS (zero where the row or the mode does not exist).  C = 8 JW: every thread moves JW
// 16-byte pieces, all requested before the first is stored (a rolled loop would pay one memory latency per piece)
template <int JW>
__device__ __forceinline__ void mx_stage(const Mix1dP& P, float* xs, int r0, int k0) {
  constexpr int C = 8 * JW, ls = 2 * C + 4, v4 = C / 2;               // float4s per line
  const long rs = 2L * P.kp * C;
  float4 val[JW];
#pragma unroll
  for (int it = 0; it < JW; ++it) {
    const int idx = threadIdx.x + it * 64 * MX_WAVES;
    const int line = idx / v4, v = idx % v4;
    const int r = line >> 3, kl = line & 7;
    val[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r0 + r < P.rows && k0 + kl < P.keff)
      val[it] = *reinterpret_cast<const float4*>(P.in + (long)(r0 + r) * rs + (long)(k0 + kl) * 2 * C + 4 * v);
  }
#pragma unroll
  for (int it = 0; it < JW; ++it) {
    const int idx = threadIdx.x + it * 64 * MX_WAVES;
    const int line = idx / v4, v = idx % v4;
    *reinterpret_cast<float4*>(xs + line * ls + 4 * v) = val[it];
  }
}

template <int JW, bool T>
__global__ __launch_bounds__(64 * MX_WAVES) void k_mix1d(const Mix1dP P) {
  extern __shared__ float xs[];
  constexpr int C = 8 * JW, ls = 2 * C + 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kl = lane & 7, ql = lane >> 3;
  const int k0 = blockIdx.x * MX_K, q0 = blockIdx.y * MX_Q, r0 = blockIdx.z * MX_R;
  const int k = min(k0 + kl, P.K - 1), q = q0 + ql, j0 = wv * JW;
  // this lane's weights for the wave's JW reduced channels: requested before the staging so that they arrive under it
  float2 wreg[JW];
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) {
    const long io = T ? ((long)q * C + (j0 + jj)) : ((long)(j0 + jj) * C + q);
    wreg[jj] = *reinterpret_cast<const float2*>(P.w + (io * P.K + k) * 2);
    if (T) wreg[jj].y = -wreg[jj].y;
  }
  mx_stage<JW>(P, xs, r0, k0);
  __syncthreads();
  float ar[MX_R], ai[MX_R];
#pragma unroll
  for (int r = 0; r < MX_R; ++r) { ar[r] = 0.f; ai[r] = 0.f; }
#pragma unroll
  for (int jq = 0; jq < JW; jq += 4) {
#pragma unroll
    for (int r = 0; r < MX_R; ++r) {
      const float* line = xs + (r * MX_K + kl) * ls + j0 + jq;
      const float4 xr = *reinterpret_cast<const float4*>(line), xi = *reinterpret_cast<const float4*>(line + C);
      const float xrv[4] = {xr.x, xr.y, xr.z, xr.w}, xiv[4] = {xi.x, xi.y, xi.z, xi.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ar[r] = fmaf(xrv[e], wreg[jq + e].x, ar[r]);
        ar[r] = fmaf(-xiv[e], wreg[jq + e].y, ar[r]);
        ai[r] = fmaf(xrv[e], wreg[jq + e].y, ai[r]);
        ai[r] = fmaf(xiv[e], wreg[jq + e].x, ai[r]);
      }
    }
  }
  // the eight waves' partial sums: red[wave][2r + ri][lane], summed in wave order
  __syncthreads();
  float* red = xs;
#pragma unroll
  for (int r = 0; r < MX_R; ++r) {
    red[(wv * 2 * MX_R + 2 * r) * 64 + lane] = ar[r];
    red[(wv * 2 * MX_R + 2 * r + 1) * 64 + lane] = ai[r];
  }
  __syncthreads();
  const long rs = 2L * P.kp * C;
  for (int idx = threadIdx.x; idx < 2 * MX_R * 64; idx += 64 * MX_WAVES) {
    const int a = idx >> 6, oq = idx & 7, ok = (idx >> 3) & 7;        // own channel fastest: 32-byte runs in memory
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < MX_WAVES; ++w) s += red[(w * 2 * MX_R + a) * 64 + oq * 8 + ok];
    const int r = r0 + (a >> 1), kk = k0 + ok;
    if (r < P.rows && kk < P.kp) P.out[(long)r * rs + (long)kk * 2 * C + (a & 1) * C + q0 + oq] = kk < P.keff ? s : 0.f;
  }
}

template <int JW>
__global__ __launch_bounds__(64 * MX_WAVES) void k_mix1d_wgrad(const Mix1dP P) {
  extern __shared__ float xs[];
  constexpr int C = 8 * JW, ls = 2 * C + 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kl = lane & 7, ql = lane >> 3;
  const int k0 = blockIdx.x * MX_K, o = blockIdx.y * MX_Q + ql, i0 = wv * JW;
  const int k = k0 + kl;
  const long rs = 2L * P.kp * C;
  float accr[JW], acci[JW];
#pragma unroll
  for (int jj = 0; jj < JW; ++jj) { accr[jj] = 0.f; acci[jj] = 0.f; }
  if (k0 < P.keff) {
    for (int r0 = 0; r0 < P.rows; r0 += MX_R) {
      float gr[MX_R], gi[MX_R];
#pragma unroll
      for (int r = 0; r < MX_R; ++r) {
        const bool ok = r0 + r < P.rows && k < P.keff;
        const float* p = P.g + (long)(ok ? r0 + r : 0) * rs + (long)(ok ? k : 0) * 2 * C + o;
        gr[r] = ok ? p[0] : 0.f;
        gi[r] = ok ? p[C] : 0.f;
      }
      if (r0) __syncthreads();
      mx_stage<JW>(P, xs, r0, k0);
      __syncthreads();
#pragma unroll
      for (int jq = 0; jq < JW; jq += 4) {
#pragma unroll
        for (int r = 0; r < MX_R; ++r) {
          const float* line = xs + (r * MX_K + kl) * ls + i0 + jq;
          const float4 xr = *reinterpret_cast<const float4*>(line), xi = *reinterpret_cast<const float4*>(line + C);
          const float xrv[4] = {xr.x, xr.y, xr.z, xr.w}, xiv[4] = {xi.x, xi.y, xi.z, xi.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            accr[jq + e] = fmaf(xrv[e], gr[r], accr[jq + e]);
            accr[jq + e] = fmaf(xiv[e], gi[r], accr[jq + e]);
            acci[jq + e] = fmaf(xrv[e], gi[r], acci[jq + e]);
            acci[jq + e] = fmaf(-xiv[e], gr[r], acci[jq + e]);
          }
        }
      }
    }
  }
  if (k < P.K) {
#pragma unroll
    for (int jj = 0; jj < JW; ++jj)
      *reinterpret_cast<float2*>(P.gw + (((long)(i0 + jj) * C + o) * P.K + k) * 2) = make_float2(accr[jj], acci[jj]);
  }
}

bool mix1d_ok(int rows, int C) {
  static const int on = [] { const char* e = getenv("RPDE_MIX1D"); return (e && e[0] == '0') ? 0 : 1; }();
  return on && rows >= 1 && rows <= 64 && (C == 32 || C == 64 || C == 128);
}

template <int JW>
static int mix1d_launch(const Mix1dP& P, int what, hipStream_t st) {
  const size_t lds = mix1d_lds(P.C);
  const dim3 block(64 * MX_WAVES);
  if (what == 2) {
    static bool attr = false;
    if (!attr) { RPDE_HIP(hipFuncSetAttribute((const void*)k_mix1d_wgrad<JW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix1d_lds(128))); attr = true; }
    hipLaunchKernelGGL((k_mix1d_wgrad<JW>), dim3((P.K + MX_K - 1) / MX_K, P.C / MX_Q), block, lds, st, P);
  } else {
    const dim3 grid((P.kp + MX_K - 1) / MX_K, P.C / MX_Q, (P.rows + MX_R - 1) / MX_R);
    if (what == 0) {
      static bool attr = false;
      if (!attr) { RPDE_HIP(hipFuncSetAttribute((const void*)k_mix1d<JW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix1d_lds(128))); attr = true; }
      hipLaunchKernelGGL((k_mix1d<JW, false>), grid, block, lds, st, P);
    } else {
      static bool attr = false;
      if (!attr) { RPDE_HIP(hipFuncSetAttribute((const void*)k_mix1d<JW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mix1d_lds(128))); attr = true; }
      hipLaunchKernelGGL((k_mix1d<JW, true>), grid, block, lds, st, P);
    }
  }
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

static int mix1d_dispatch(const Mix1dP& P, int what, hipStream_t st) {
  switch (P.C) {
    case 32: return mix1d_launch<4>(P, what, st);
    case 64: return mix1d_launch<8>(P, what, st);
    default: return mix1d_launch<16>(P, what, st);
  }
}

// out[rows][kp][2][C] (all kp modes written; the padding modes with zeros)
int mix1d(const float* in, const float* w, float* out, int rows, int C, int K, int keff, int kp, bool transpose, hipStream_t st) {
  Mix1dP P;
  P.in = in; P.w = w; P.out = out; P.g = nullptr; P.gw = nullptr;
  P.rows = rows; P.C = C; P.K = K; P.keff = keff; P.kp = kp;
  return mix1d_dispatch(P, transpose ? 1 : 0, st);
}

// gw[C][C][K][2] (all K modes written; k >= keff with zeros)
int mix1d_wgrad(const float* spec, const float* gspec, float* gw, int rows, int C, int K, int keff, int kp, hipStream_t st) {
  Mix1dP P;
  P.in = spec; P.w = nullptr; P.out = nullptr; P.g = gspec; P.gw = gw;
  P.rows = rows; P.C = C; P.K = K; P.keff = keff; P.kp = kp;
  return mix1d_dispatch(P, 2, st);
}

}  // namespace rpde
