// Last FNO block + projection MLP of the evaluation forward in ONE pass (reference: models/fno.py:143-150 with
// models/fno_blocks.py:63-83 and :76-83):
//
//   H[c]  = act( bc[c] + sum_i Wc[c][i] x[b][i][m][n] + sum_r t[b][c][m][r] Fs[r][n] )            (the block's tail)
//   out[q] = b2[q] + sum_h W2[q][h] gelu( b1[h] + sum_c W1[h][c] H[c] )                            (the projection)
//
// As two launches (k_conv_syn_h2, then k_conv_mlp_h2) the block's output crosses HBM once each way -- 2 x 537 MB at
// 512^2, B = 16, width 32: the tail moves 1.07 GB at the rate its CUs stream (228 us) and the projection reads 537 MB
// back to spend ~400 us on 537 M exact GELUs.  Fused, the 32-channel field between them exists only as the 8 values a
// lane holds for its point: x is read once, a ONE-channel field is written, and the vector work of the projection --
// which bounds the pair either way -- runs beside the loads instead of behind them.
//
// Everything is an h2 product (h2.h) in the transposed form of ff_fused.hip: a wave owns whole rows (b, m) and walks
// along n in tiles of 16 points; the 32 x 16 accumulator pair of the tail (lane (g, li): channels 16 mt + 4 g + j of point
// li) IS the B operand of the projection's first product once W1's fragments carry the matching permutation of the
// reduction index (ff_perm), so H never changes lanes.  Resident per wave: Wc and W1 as A fragments (registers), the
// row's spectra t_bm as A fragments (4 loads per row); shared by the workgroup in LDS: the synthesis table Fs as ready B
// fragments per 16-point tile, b1 and W2.  Plain loads one tile ahead; no hand-counted waits.
#include "conv_small.h"
#include "h2.h"

#include <stdlib.h>

namespace rpde {

__device__ __forceinline__ int cp_perm(int g, int j) { return 16 * (j >> 2) + 4 * g + (j & 3); }     // = ff_fused.hip's ff_perm

struct ConvProjP {
  const float* x; const float* wc; const float* bc; const float* t; const float* fs_t;
  const float* w1; const float* b1; const float* w2; const float* b2;
  float* out;
  int B, Cout, M, N, R2, Cmid, Cq, act;
};

// Cin = 32, Cout <= 32 (the block's width), Cmid <= 16 MT, Cq <= CO outputs; N a multiple of 64.
//
// Loads.  The first version let lane (g, li) fetch its eight channels of point li with 4-byte loads (64-byte runs per
// channel), as k_conv_mlp_h2 does: 500 us for 537 MB -- bound by the load pattern, not by the 670 M GELUs.  Now a lane
// fetches FOUR consecutive points of each of its eight channels (16 bytes; a wave instruction = four channels x 256
// contiguous bytes) and the four components become column li of four different MFMA tiles: tile q of a 64-point
// group lets column li stand for point 4 li + q.  The synthesis table's fragments in LDS are built with the same map,
// and the four results a lane ends up with are four consecutive points: one 16-byte store.  No transposition anywhere.
template <int MT, int CO, int ACT>
__global__ __launch_bounds__(512, 2) void k_conv_syn_proj_h2(const ConvProjP P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // table fragments | W1 fragments | b1, W2
  __shared__ float red[3][8];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, g = l >> 4, li = l & 15;
  const int N = P.N, M = P.M, R2 = P.R2, ngrp = N >> 6;             // groups of 64 points = 4 tiles
  const long S = (long)M * N;
  char* const w1img = smem + ngrp * 4 * 2048;                       // [MT][hi | lo][1 KB]
  float* const vecs = reinterpret_cast<float*>(w1img + MT * 2048);  // b1[16 MT] | w2[CO][16 MT]
  // ---- scales of the resident matrices (one power of two each) ----
  float mc = 0.f, m1 = 0.f, mf = 0.f;
  for (int e = tid; e < P.Cout * 32; e += 512) mc = fmaxf(mc, fabsf(P.wc[e]));
  for (int e = tid; e < P.Cmid * P.Cout; e += 512) m1 = fmaxf(m1, fabsf(P.w1[e]));
  for (int e = tid; e < R2 * N; e += 512) mf = fmaxf(mf, fabsf(P.fs_t[e]));
  mc = wave_max(mc); m1 = wave_max(m1); mf = wave_max(mf);
  if (l == 0) { red[0][wv] = mc; red[1][wv] = m1; red[2][wv] = mf; }
  __syncthreads();
  mc = m1 = mf = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { mc = fmaxf(mc, red[0][i]); m1 = fmaxf(m1, red[1][i]); mf = fmaxf(mf, red[2][i]); }
  float csc, cinv, s1c, s1inv, fsc, finv;
  h2_scale(mc, 0, csc, cinv);
  h2_scale(m1, 0, s1c, s1inv);
  h2_scale(mf, 0, fsc, finv);
  // ---- LDS: table fragments B[k = r][col]: tile q of group T, lane (g, li) holds r = 8 g + j of point 64 T + 4 li + q ----
  for (int it = tid; it < ngrp * 4 * 64; it += 512) {
    const int tq = it >> 6, ln = it & 63, gg = ln >> 4, ll = ln & 15;
    const int point = 64 * (tq >> 2) + 4 * ll + (tq & 3);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 8 * gg + j;
      v[j] = r < R2 ? P.fs_t[(long)r * N + point] * fsc : 0.f;
    }
    uint2 h0, l0, h1, l1;
    h2_split4(v[0], v[1], v[2], v[3], h0, l0);
    h2_split4(v[4], v[5], v[6], v[7], h1, l1);
    *reinterpret_cast<uint4*>(smem + tq * 2048 + ln * 16) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4*>(smem + tq * 2048 + 1024 + ln * 16) = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
  // W1 (rows = hidden, k = H channel in the slot order of the tail's accumulators: cp_perm)
  for (int it = tid; it < MT * 64; it += 512) {
    const int mt = it >> 6, ln = it & 63, gg = ln >> 4, ll = ln & 15;
    const int hrow = 16 * mt + ll;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cp_perm(gg, j);
      v[j] = (hrow < P.Cmid && c < P.Cout) ? P.w1[hrow * P.Cout + c] * s1c : 0.f;
    }
    uint2 h0, l0, h1, l1;
    h2_split4(v[0], v[1], v[2], v[3], h0, l0);
    h2_split4(v[4], v[5], v[6], v[7], h1, l1);
    *reinterpret_cast<uint4*>(w1img + mt * 2048 + ln * 16) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4*>(w1img + mt * 2048 + 1024 + ln * 16) = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
  for (int e = tid; e < 16 * MT; e += 512) {
    vecs[e] = (P.b1 && e < P.Cmid) ? P.b1[e] : 0.f;
#pragma unroll
    for (int q = 0; q < CO; ++q) vecs[16 * MT * (1 + q) + e] = (q < P.Cq && e < P.Cmid) ? P.w2[q * P.Cmid + e] : 0.f;
  }
  // ---- registers: Wc (rows = block channels, k = input channel 8 g + j) ----
  f16x8 wch[2], wcl[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int o = 16 * mt + li;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = o < P.Cout ? P.wc[o * 32 + 8 * g + j] * csc : 0.f;
    union { f16x8 v; struct { uint2 a, b; } u; } H, L;
    h2_split4(v[0], v[1], v[2], v[3], H.u.a, L.u.a);
    h2_split4(v[4], v[5], v[6], v[7], H.u.b, L.u.b);
    wch[mt] = H.v; wcl[mt] = L.v;
  }
  float bb[2][4], b2v[CO];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int o = 16 * mt + 4 * g + j;
      bb[mt][j] = (P.bc && o < P.Cout) ? P.bc[o] : 0.f;
    }
#pragma unroll
  for (int q = 0; q < CO; ++q) b2v[q] = (P.b2 && q < P.Cq) ? P.b2[q] : 0.f;
  __syncthreads();

  // ---- rows (b, m): wave id, id + waves, ..; per row 4 loads of spectra, then N / 64 groups of four tiles ----
  const long rows = (long)P.B * M, wstride = (long)gridDim.x * 8;
  float4 xv[8];
  auto fetch = [&](long row, int T) {                  // lane (g, li): input channels 8 g + j, points 64 T + 4 li .. + 3
    const long b = row / M, m = row - b * M;
    const float* p = P.x + (b * 32 + 8 * g) * S + m * N + 64 * T + 4 * li;
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = *reinterpret_cast<const float4*>(p + j * S);
  };
  float4 prev_r[CO];
  long prev_off = 0;
  bool have_prev = false;
  long row = (long)blockIdx.x * 8 + wv;
  if (row < rows) fetch(row, 0);
  for (; row < rows; row += wstride) {
    const long b = row / M, m = row - b * M;
    // the row's spectra as A fragments: lane (g, li) holds t[o = 16 mt + li][r = 8 g .. 8 g + 7]
    f16x8 th[2], tl[2];
    float tinv;
    {
      float tv[2][8], mt_ = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int o = 16 * mt + li;
        const bool ok = o < P.Cout && 8 * g < R2;
        const float* pt = P.t + ((b * P.Cout + (ok ? o : 0)) * M + m) * R2 + (ok ? 8 * g : 0);
        const float4 a = *reinterpret_cast<const float4*>(pt), c = *reinterpret_cast<const float4*>(pt + 4);
        const float v8[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) { tv[mt][j] = ok ? v8[j] : 0.f; mt_ = fmaxf(mt_, fabsf(tv[mt][j])); }
      }
      float tsc;
      h2_scale(wave_max(mt_), 0, tsc, tinv);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        union { f16x8 v; struct { uint2 a, b; } u; } H, L;
        h2_split4(tv[mt][0] * tsc, tv[mt][1] * tsc, tv[mt][2] * tsc, tv[mt][3] * tsc, H.u.a, L.u.a);
        h2_split4(tv[mt][4] * tsc, tv[mt][5] * tsc, tv[mt][6] * tsc, tv[mt][7] * tsc, H.u.b, L.u.b);
        th[mt] = H.v; tl[mt] = L.v;
      }
    }
    const float i2 = tinv * finv;
    for (int T = 0; T < ngrp; ++T) {
      float4 xc[8];
      float mx = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xc[j] = xv[j];
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(xc[j].x), fabsf(xc[j].y)), fmaxf(fabsf(xc[j].z), fabsf(xc[j].w))));
      }
      // the PREVIOUS group's results leave here, in front of the next group's loads: the compiler waits for vmcnt(0)
      // before the first use of a prefetched value, and a store issued just before that wait would put its whole round
      // trip there; stored here it is older than the loads the next wait is for and has a group's time to complete
      if (have_prev) {
#pragma unroll
        for (int q = 0; q < CO; ++q)
          if (g == 0 && q < P.Cq) *reinterpret_cast<float4*>(P.out + prev_off + (long)q * S) = prev_r[q];
      }
      if (T + 1 < ngrp) fetch(row, T + 1);
      else if (row + wstride < rows) fetch(row + wstride, 0);
      float xs, xi;
      h2_scale(wave_max(mx), 0, xs, xi);
      const float i1 = xi * cinv;
      float res[CO][4];
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {                  // tile q4: column li = point 64 T + 4 li + q4
        asm volatile("" ::: "memory");                  // (keeps the LDS reads of later tiles from being hoisted: registers)
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (q4 == 0 ? xc[j].x : (q4 == 1 ? xc[j].y : (q4 == 2 ? xc[j].z : xc[j].w))) * xs;
        union { f16x8 v; struct { uint2 a, b; } u; } XH, XL;
        h2_split4(v[0], v[1], v[2], v[3], XH.u.a, XL.u.a);
        h2_split4(v[4], v[5], v[6], v[7], XH.u.b, XL.u.b);
        const char* tf = smem + (T * 4 + q4) * 2048 + l * 16;
        const f16x8 fh = *reinterpret_cast<const f16x8*>(tf), fl = *reinterpret_cast<const f16x8*>(tf + 1024);
        // ---- the block's tail: H[c = 16 mt + 4 g + j][column li] ----
        float hv[8], hm = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          f32x4v a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
          a1 = h2_mfma32(wch[mt], wcl[mt], XH.v, XL.v, a1);
          a2 = h2_mfma32(th[mt], tl[mt], fh, fl, a2);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float u = fmaf(a1[j], i1, fmaf(a2[j], i2, bb[mt][j]));
            hv[4 * mt + j] = act_f(ACT, u);              // (ACT is a template constant: no per-element branches)
            hm = fmaxf(hm, fabsf(hv[4 * mt + j]));
          }
        }
        // ---- projection: the eight H values of a lane are the slots (g, j) of a B fragment whose slot -> channel map is
        // cp_perm, the map W1's fragments were built with ----
        float hs, hi_;
        h2_scale(wave_max(hm), 0, hs, hi_);
        union { f16x8 v; struct { uint2 a, b; } u; } PH, PL;
        h2_split4(hv[0] * hs, hv[1] * hs, hv[2] * hs, hv[3] * hs, PH.u.a, PL.u.a);
        h2_split4(hv[4] * hs, hv[5] * hs, hv[6] * hs, hv[7] * hs, PH.u.b, PL.u.b);
        const float i3 = hi_ * s1inv;
        float part[CO];
#pragma unroll
        for (int q = 0; q < CO; ++q) part[q] = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if ((mt & 1) == 0) asm volatile("" ::: "memory");
          const f16x8 w1h = *reinterpret_cast<const f16x8*>(w1img + mt * 2048 + l * 16);
          const f16x8 w1l = *reinterpret_cast<const f16x8*>(w1img + mt * 2048 + 1024 + l * 16);
          f32x4v c = {0.f, 0.f, 0.f, 0.f};
          c = h2_mfma32(w1h, w1l, PH.v, PL.v, c);
          const float4 b1v = *reinterpret_cast<const float4*>(vecs + 16 * mt + 4 * g);
          const float b1a[4] = {b1v.x, b1v.y, b1v.z, b1v.w};
          float hh[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) hh[j] = gelu_f(fmaf(c[j], i3, b1a[j]));
#pragma unroll
          for (int q = 0; q < CO; ++q) {
            const float4 w2v = *reinterpret_cast<const float4*>(vecs + 16 * MT * (1 + q) + 16 * mt + 4 * g);
            part[q] = fmaf(w2v.x, hh[0], fmaf(w2v.y, hh[1], fmaf(w2v.z, hh[2], fmaf(w2v.w, hh[3], part[q]))));
          }
        }
#pragma unroll
        for (int q = 0; q < CO; ++q) {
          float r = part[q];
          r += lane_xor16(r);
          r += lane_xor32(r);
          res[q][q4] = r + b2v[q];
        }
      }
#pragma unroll
      for (int q = 0; q < CO; ++q) prev_r[q] = make_float4(res[q][0], res[q][1], res[q][2], res[q][3]);
      prev_off = b * P.Cq * S + m * N + 64 * T + 4 * li;
      have_prev = true;
    }
  }
  if (have_prev) {
#pragma unroll
    for (int q = 0; q < CO; ++q)
      if (g == 0 && q < P.Cq) *reinterpret_cast<float4*>(P.out + prev_off + (long)q * S) = prev_r[q];
  }
}

bool conv_syn_proj_ok(int Cin, int Cout, int M, int N, int R2, int Cmid, int Cq) {
  if (const char* e = getenv("RPDE_CONV_PROJ")) if (e[0] == '0') return false;
  return Cin == 32 && Cout >= 1 && Cout <= 32 && Cmid >= 1 && Cmid <= 128 && Cq >= 1 && Cq <= 4 && N % 64 == 0 && N >= 64 &&
         N <= 1024 && R2 >= 8 && R2 <= 32 && R2 % 8 == 0 && M >= 1;
}

template <int MT, int CO>
static int conv_syn_proj_launch(const ConvProjP& P, int grid, size_t lds, hipStream_t st) {
  // (more than the default dynamic LDS limit at N >= 512: opt in -- a host-side attribute of the function, set per call:
  //  cheap, and right on every device)
#define RPDE_CP_GO(ACT_)                                                                                                   \
  do {                                                                                                                   \
    if (lds > 64 * 1024)                                                                                                 \
      RPDE_HIP(hipFuncSetAttribute((const void*)k_conv_syn_proj_h2<MT, CO, ACT_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   (int)lds));                                                                           \
    hipLaunchKernelGGL((k_conv_syn_proj_h2<MT, CO, ACT_>), dim3(grid), dim3(512), lds, st, P);                           \
  } while (0)
  if (P.act == RPDE_ACT_GELU) RPDE_CP_GO(RPDE_ACT_GELU);
  else if (P.act == RPDE_ACT_RELU) RPDE_CP_GO(RPDE_ACT_RELU);
  else RPDE_CP_GO(RPDE_ACT_IDENTITY);
#undef RPDE_CP_GO
  return RPDE_OK;
}

int conv_syn_proj(const float* x, const float* wc, const float* bc, const float* t, const float* fs_t, const float* w1,
                  const float* b1, const float* w2, const float* b2, float* out, int B, int Cout, int M, int N, int R2, int Cmid,
                  int Cq, int act, hipStream_t st) {
  ConvProjP P{x, wc, bc, t, fs_t, w1, b1, w2, b2, out, B, Cout, M, N, R2, Cmid, Cq, act};
  int dev = 0, cus = 256;
  RPDE_HIP(hipGetDevice(&dev));
  RPDE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const long rows = (long)B * M;
  long grid = (rows + 7) / 8;
  if (grid > cus) grid = cus;                           // one workgroup of eight waves per CU
  const int MT = (Cmid + 15) / 16;
  const int MTi = MT <= 2 ? 2 : (MT <= 4 ? 4 : 8);
  const size_t lds = (size_t)(N / 16) * 2048 + (size_t)MTi * 2048 + sizeof(float) * 16 * MTi * 5;
  // (Cq = 2, 3 run the four-output instance with zero weights for the missing outputs)
  if (MTi == 2) { if (Cq == 1) RPDE_TRY((conv_syn_proj_launch<2, 1>(P, (int)grid, lds, st))); else RPDE_TRY((conv_syn_proj_launch<2, 4>(P, (int)grid, lds, st))); }
  else if (MTi == 4) { if (Cq == 1) RPDE_TRY((conv_syn_proj_launch<4, 1>(P, (int)grid, lds, st))); else RPDE_TRY((conv_syn_proj_launch<4, 4>(P, (int)grid, lds, st))); }
  else { if (Cq == 1) RPDE_TRY((conv_syn_proj_launch<8, 1>(P, (int)grid, lds, st))); else RPDE_TRY((conv_syn_proj_launch<8, 4>(P, (int)grid, lds, st))); }
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // namespace rpde
