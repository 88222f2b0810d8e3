// Mode mixing of FSpectralConv2d.forward_fourier (reference: models/spectral_convolution.py:270-275, 294-299: the
// einsums "bixy,ioy->boxy" / "bixy,iox->boxy" on the retained bins) and of its adjoint, for the fused path: ONE launch
// takes the fp32 spectra of both axes, [line][k][re|im][64], multiplies every retained mode by its complex 64 x 64
// weight in h2 arithmetic (h2.h) and writes the result directly as the B-fragment image + per-line scale that the
// synthesis kernel consumes -- what used to be pack + GEMM + split per axis (7 launches, every spectrum across HBM
// three times) is now prep + mix (2 launches, once each way).
//
//   * complex structure: out_re = re.Wr - im.Wi, out_im = re.Wi + im.Wr -- only Wr and Wi are kept (not the 128 x 128
//     real block form), the minus sign rides on a negated copy of the im input fragments (exact).
//   * weight-stationary: a 4-wave workgroup owns (axis, four modes 4q..4q+3); wave cb owns output channels
//     16cb..16cb+15 and keeps its Wr / Wi fragments of the four modes in registers (32 fragments, 128 VGPRs) while it
//     walks over line tiles.  Four modes x (re, im) = the eight reduction rows 8q..8q+7 of a line's spectrum = exactly
//     one 16-byte piece of the synthesis operand per (line, channel): the accumulators leave as whole pieces.
//   * per line tile (16 lines): wave w converts mode 4q+w of the 16 lines (fp32, prefetched one tile ahead) into A
//     fragments in LDS (shared by the four waves); 96 MFMAs per wave; epilogue: scale, split, 16-byte stores.
//   * scaling without a reduction over the result: the analysis kernel leaves max|spectrum| per line (amax); the input
//     fragments use it, and the output uses the BOUND amax * max_k,o sum_i (|Wr| + |Wi|) -- a bound that is 2^k too
//     large costs k of the ~17 spare bits of the two-piece format (h2.h), nothing else.
#include "fused_spectral.h"
#include "h2.h"

#include <stdlib.h>

namespace rpde {

// ------------------------------------------------------------------------------------------------------------
// weight preparation: w [64][64][K][2] of both axes -> B fragments [axis][k < kp][cb][Wr|Wi][ks][hi|lo] (1 KB each) and
// per (axis, k) {1 / scale, norm}; conj_t: the adjoint (W^H: transposed, Wi negated)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mix_prep(const float* __restrict__ w_y, const float* __restrict__ w_x, int K, int keff,
                                                  int kp, int conj_t, char* __restrict__ img, float* __restrict__ wc) {
  __shared__ float ws[64 * 64 * 2];
  __shared__ float red[2][4];
  const int k = blockIdx.x, a = blockIdx.y, tid = threadIdx.x, l = tid & 63, wv = tid >> 6, g = l >> 4, li = l & 15;
  const float* __restrict__ w = a ? w_x : w_y;
  char* out = img + (size_t)(a * kp + k) * MIX_W_BYTES_PER_MODE;
  if (k >= keff) {                       // padding modes (kp = modes rounded up to 4): zero weight
    for (int i = tid; i < MIX_W_BYTES_PER_MODE / 16; i += 256) reinterpret_cast<uint4*>(out)[i] = make_uint4(0, 0, 0, 0);
    if (tid == 0) { wc[(a * kp + k) * 2] = 1.f; wc[(a * kp + k) * 2 + 1] = 0.f; }
    return;
  }
  float m = 0.f;
  float2 v[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float2*>(w + ((long)(tid + 256 * u) * K + k) * 2);   // e = i * 64 + o
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = tid + 256 * u;
    ws[e * 2] = v[u].x; ws[e * 2 + 1] = v[u].y;
    m = fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y)));
  }
  __syncthreads();
  // norm: forward max_o sum_i, adjoint max_i sum_o of |Wr| + |Wi|; four threads per sum (16 terms each)
  float s = 0.f;
  {
    const int c = tid & 63, part = tid >> 6;
    for (int r = 16 * part; r < 16 * part + 16; ++r) {
      // adjoint: thread c walks row c; the start is rotated by c so that the 64 threads hit different LDS banks
      const int e = conj_t ? c * 64 + ((r + c) & 63) : r * 64 + c;
      s += fabsf(ws[e * 2]) + fabsf(ws[e * 2 + 1]);
    }
  }
  __shared__ float psum[4][64];
  psum[tid >> 6][tid & 63] = s;
  __syncthreads();
  s = tid < 64 ? (psum[0][tid] + psum[1][tid]) + (psum[2][tid] + psum[3][tid]) : 0.f;
  m = wave_max(m);
  s = wave_max(s);
  if (l == 0) { red[0][wv] = m; red[1][wv] = s; }
  __syncthreads();
  m = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  s = red[1][0];
  float sc, iv;
  h2_scale(m, 0, sc, iv);
  if (tid == 0) { wc[(a * kp + k) * 2] = iv; wc[(a * kp + k) * 2 + 1] = s; }
  // wave wv = channel block cb: B[kred = 32 ks + 8 g + j][col = 16 cb + li]
  const int cb = wv;
#pragma unroll
  for (int part = 0; part < 2; ++part)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kred = 32 * ks + 8 * g + j, col = 16 * cb + li;
        const int e = conj_t ? col * 64 + kred : kred * 64 + col;
        float x = ws[e * 2 + part];
        if (conj_t && part) x = -x;
        v[j] = x * sc;
      }
      uint2 h0, l0, h1, l1;
      h2_split4(v[0], v[1], v[2], v[3], h0, l0);
      h2_split4(v[4], v[5], v[6], v[7], h1, l1);
      char* p = out + (((cb * 2 + part) * 2 + ks) * 2) * 1024 + l * 16;
      *reinterpret_cast<uint4*>(p) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4*>(p + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

size_t mix_wimg_bytes(int kp) { return (size_t)2 * kp * MIX_W_BYTES_PER_MODE; }

int mix_prep(const float* w_y, const float* w_x, int K, int keff, int kp, int conj_t, void* wimg, float* wc, hipStream_t st) {
  hipLaunchKernelGGL(k_mix_prep, dim3(kp, 2), dim3(256), 0, st, w_y, w_x, K, keff, kp, conj_t, (char*)wimg, wc);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// ------------------------------------------------------------------------------------------------------------
// mix
// ------------------------------------------------------------------------------------------------------------
struct MixP {
  const float* spec[2];      // [lines][R][64]
  char* img[2];              // operand blocks of the synthesis (fused_spectral.hip)
  const float* amax[2];      // max |spectrum| per line (from the analysis kernel)
  float* inv[2];             // out: 1 / (line scale * table scale)
  int tiles[2];              // line tiles (16 lines) per axis
  int tps[2];                // tiles per slice
  int nblk0;                 // workgroups of axis 0
  const char* wimg; const float* wc;
  int kp, R, nq;
};

__device__ __forceinline__ void lds_barrier_mix() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS: A fragments of one line tile: [mode 4][re, im, -im][ks 2][hi|lo][1 KB] = 48 KB
constexpr int MIX_LDS = 4 * 3 * 2 * 2 * 1024;

template <int K32, int TG>
__global__ __launch_bounds__(256, 2) void k_mix_h2(const MixP P) {
  constexpr int BB = h2_block_bytes(K32, TG);
  constexpr int NP = h2_np(TG);
  __shared__ __attribute__((aligned(16))) char smem[MIX_LDS];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, li = l & 15;
  int bid = blockIdx.x;
  const int a = bid >= P.nblk0 ? 1 : 0;
  if (a) bid -= P.nblk0;
  const int q = bid % P.nq, slice = bid / P.nq;
  const int t0 = slice * P.tps[a], t1 = min(t0 + P.tps[a], P.tiles[a]);
  if (t0 >= t1) return;
  const int cb = w;
  const float* __restrict__ spec = P.spec[a];
  const float* __restrict__ amax = P.amax[a];
  char* __restrict__ img = P.img[a];
  const long lstride = (long)P.R * 64;

  // ---- this wave's weights: modes 4q..4q+3, output channels 16cb.., [mode][Wr|Wi][ks][hi|lo] ----
  f16x8 wf[4][2][2][2];
  float winv[4];
  {
    const char* base = P.wimg + ((size_t)(a * P.kp + 4 * q) * 4 + cb) * (MIX_W_BYTES_PER_MODE / 4) + l * 16;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int f = 0; f < 8; ++f)
        wf[kk][f >> 2][(f >> 1) & 1][f & 1] = *reinterpret_cast<const f16x8*>(base + (size_t)kk * MIX_W_BYTES_PER_MODE + f * 1024);
      winv[kk] = P.wc[(a * P.kp + 4 * q + kk) * 2];
    }
  }
  float wnorm = 0.f;
  for (int k = l; k < P.kp; k += 64) wnorm = fmaxf(wnorm, P.wc[(a * P.kp + k) * 2 + 1]);
  wnorm = wave_max(wnorm);

  // raw spectrum of mode 4q + w for the 16 lines of a tile: lane (g, li) holds, of line li, the floats
  // [part 64 + 32 ks + 8 g .. + 7] for part = re, im and ks = 0, 1 -- as eight float4, fetched one tile ahead
  // (everything a tile reads from global memory is requested before the previous tile's stores: loads and stores
  //  share one in-order counter, and a load issued behind the stores would wait for every one of them)
  float4 raw[8];
  float am_next = 0.f;
  auto issue = [&](int t) {
    const float* p = spec + ((long)t * 16 + li) * lstride + (4 * q + w) * 128 + 8 * g;
#pragma unroll
    for (int n = 0; n < 8; ++n) raw[n] = *reinterpret_cast<const float4*>(p + (n >> 2) * 64 + ((n >> 1) & 1) * 32 + (n & 1) * 4);
    am_next = amax[(long)t * 16 + li];
  };
  issue(t0);
  // octet 8q..8q+7 of the reduction rows -> where its 16-byte pieces live inside a line's operand block
  const bool tail = q >= 4 * K32;
  for (int t = t0; t < t1; ++t) {
    const long line0 = (long)t * 16;
    // ---- convert: A fragments of mode 4q+w -> LDS (rows = lines; one scale per line = row) ----
    const float am_li = am_next;                         // max |spectrum| of line line0 + li (fetched a tile ahead)
    {
      const float am = am_li;
      float sin_, iin;
      h2_scale(am, 0, sin_, iin);
#pragma unroll
      for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const float4 v0 = raw[part * 4 + ks * 2], v1 = raw[part * 4 + ks * 2 + 1];
          uint2 h0, l0, h1, l1;
          h2_split4(v0.x * sin_, v0.y * sin_, v0.z * sin_, v0.w * sin_, h0, l0);
          h2_split4(v1.x * sin_, v1.y * sin_, v1.z * sin_, v1.w * sin_, h1, l1);
          const uint4 hi = make_uint4(h0.x, h0.y, h1.x, h1.y), lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
          char* d = smem + (((w * 3 + part) * 2 + ks) * 2) * 1024 + l * 16;
          *reinterpret_cast<uint4*>(d) = hi;
          *reinterpret_cast<uint4*>(d + 1024) = lo;
          if (part) {            // -im: sign bits of the eight f16 flipped
            const unsigned s = 0x80008000u;
            char* dn = smem + (((w * 3 + 2) * 2 + ks) * 2) * 1024 + l * 16;
            *reinterpret_cast<uint4*>(dn) = make_uint4(hi.x ^ s, hi.y ^ s, hi.z ^ s, hi.w ^ s);
            *reinterpret_cast<uint4*>(dn + 1024) = make_uint4(lo.x ^ s, lo.y ^ s, lo.z ^ s, lo.w ^ s);
          }
        }
    }
    if (t + 1 < t1) issue(t + 1);
    lds_barrier_mix();                                   // fragments of all four modes are in LDS
    f32x4v cre[4], cim[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      cre[kk] = (f32x4v){0.f, 0.f, 0.f, 0.f};
      cim[kk] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const char* fa = smem + ((kk * 3 * 2 + ks) * 2) * 1024 + l * 16;
        const f16x8 reh = *reinterpret_cast<const f16x8*>(fa), rel = *reinterpret_cast<const f16x8*>(fa + 1024);
        const f16x8 imh = *reinterpret_cast<const f16x8*>(fa + 4096), iml = *reinterpret_cast<const f16x8*>(fa + 4096 + 1024);
        const f16x8 nih = *reinterpret_cast<const f16x8*>(fa + 8192), nil = *reinterpret_cast<const f16x8*>(fa + 8192 + 1024);
        cre[kk] = h2_mfma32(reh, rel, wf[kk][0][ks][0], wf[kk][0][ks][1], cre[kk]);
        cre[kk] = h2_mfma32(nih, nil, wf[kk][1][ks][0], wf[kk][1][ks][1], cre[kk]);
        cim[kk] = h2_mfma32(reh, rel, wf[kk][1][ks][0], wf[kk][1][ks][1], cim[kk]);
        cim[kk] = h2_mfma32(imh, iml, wf[kk][0][ks][0], wf[kk][0][ks][1], cim[kk]);
      }
    }
    lds_barrier_mix();                                   // everyone has its fragments in registers: LDS is free again
    // ---- epilogue: lane (g, li) holds, for lines 4g+jj, channel 16cb+li, the eight rows 8q..8q+7 ----
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const long line = line0 + 4 * g + jj;
      // this line's maximum sits in the lanes li = 4g + jj of the wave
      const float am = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (4 * g + jj), __float_as_int(am_li)));
      float sin_, iin, sout, iout;
      h2_scale(am, 0, sin_, iin);
      h2_scale(am * wnorm, H2_TABLE_EXP, sout, iout);
      const float f = iin * sout;
      float v[8];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float fk = f * winv[kk];
        v[2 * kk] = cre[kk][jj] * fk;
        v[2 * kk + 1] = cim[kk][jj] * fk;
      }
      uint2 h0, l0, h1, l1;
      h2_split4(v[0], v[1], v[2], v[3], h0, l0);
      h2_split4(v[4], v[5], v[6], v[7], h1, l1);
      const uint4 hi = make_uint4(h0.x, h0.y, h1.x, h1.y), lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
      char* blk = img + (line * 4 + cb) * (long)BB;
      if (!tail) {
        const int s = q >> 2, gq = q & 3;
        *reinterpret_cast<uint4*>(blk + s * 1024 + (gq * 16 + li) * 16) = hi;
        *reinterpret_cast<uint4*>(blk + (K32 + s) * 1024 + (gq * 16 + li) * 16) = lo;
      } else if (TG > 0) {
        const int qt = q - 4 * K32;
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {               // spectrum side of the packed tail: (hi, lo, hi)
          const int slot = tt * TG + qt;
          *reinterpret_cast<uint4*>(blk + (2 * K32 + slot / 4) * 1024 + ((slot & 3) * 16 + li) * 16) = tt == 1 ? lo : hi;
        }
        if (qt == 0) {                                  // slots no group uses: zero (the table side holds zeros there too)
#pragma unroll
          for (int slot = 3 * TG; slot < 4 * NP; ++slot)
            *reinterpret_cast<uint4*>(blk + (2 * K32 + slot / 4) * 1024 + ((slot & 3) * 16 + li) * 16) = make_uint4(0, 0, 0, 0);
        }
      }
      if (q == 0 && cb == 0 && li == 0) P.inv[a][line] = iout;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// weight gradient of the mode mix:  gWr[i][o] = sum_lines (re_i gre_o + im_i gim_o),  gWi[i][o] = sum_lines (re_i gim_o
// - im_i gre_o)  per axis and mode, from the saved spectra and the gradient spectra (both [line][k][re|im][64] fp32).
// The reduction runs over the LINES, so both MFMA operands need the line index as k: a tile of 32 lines x (re | im) x 64
// channels of each tensor is staged in LDS as f16 pieces in memory order and read back with transposing reads
// (ds_read_b64_tr_b16) -- the same register image serves as A operand (rows = channels i) and as B operand (columns =
// channels o).  Wave w owns the output rows i = 16w .. 16w + 15 of both gWr and gWi (8 accumulator tiles); one
// workgroup per (axis, mode, slab of lines); slabs are folded in fixed order by k_mix_wgrad_fold.  One power-of-two
// scale per tensor and axis, from the line maxima the analysis kernels leave (a product's error scales with the OTHER
// operand's magnitude, so modes far below the maximum keep their relative accuracy).
// Replaces the 128 x 128 real-block GEMM + unpack per axis of round 2 (2 x (52 + 26) us at B = 32).
// ------------------------------------------------------------------------------------------------------------
struct MixWgP {
  const float* spec[2]; const float* gspec[2];   // [lines][R][64]
  const float* amax_s[2]; const float* amax_g[2];
  float* slabs;                                   // [S][axis][kp][64][64][2]
  long lines[2];
  int kp, keff, R, S;
};

__device__ __forceinline__ int mixw_stage_off(int k, int c8) {      // = stage_off of fused_spectral.hip
  return k * 128 + ((c8 ^ ((((k >> 1) & 1) << 2) | (((k >> 3) & 1) << 3))) << 3);
}

__global__ __launch_bounds__(256, 2) void k_mix_wgrad_h2(const MixWgP P) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 8192];      // pieces spec re, spec im, g re, g im: [hi 4 KB | lo 4 KB]
  __shared__ float red[2][4];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, li = l & 15;
  const int k = blockIdx.x, s = blockIdx.y, a = blockIdx.z;
  const long lines = P.lines[a];
  const long tiles = lines / 32, tps = (tiles + P.S - 1) / P.S;
  const long t0 = s * tps, t1 = min(t0 + tps, tiles);
  float* const slab = P.slabs + (((long)s * 2 + a) * P.kp + k) * (64 * 64 * 2);
  // ---- scales: max over the axis' line maxima ----
  float ms = 0.f, mg = 0.f;
  for (long i = tid; i < lines; i += 256) { ms = fmaxf(ms, P.amax_s[a][i]); mg = fmaxf(mg, P.amax_g[a][i]); }
  ms = wave_max(ms); mg = wave_max(mg);
  if (l == 0) { red[0][w] = ms; red[1][w] = mg; }
  __syncthreads();
  ms = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
  mg = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  float sc_s, iv_s, sc_g, iv_g;
  h2_scale(ms, 0, sc_s, iv_s);
  h2_scale(mg, 0, sc_g, iv_g);

  // staging: thread -> line tid >> 3 of the tile, 16 consecutive floats (tid & 7) of the line's 128 (re 64 | im 64)
  const int sl = tid >> 3, ch = tid & 7, part = ch >> 2, c80 = 4 * (ch & 3);
  const long lstride = (long)P.R * 64;
  float4 rs[4], rg[4];
  auto issue = [&](long t) {
    const long off = (t * 32 + sl) * lstride + k * 128 + ch * 16;
    const float4* ps = reinterpret_cast<const float4*>(P.spec[a] + off);
    const float4* pg = reinterpret_cast<const float4*>(P.gspec[a] + off);
#pragma unroll
    for (int i = 0; i < 4; ++i) { rs[i] = ps[i]; rg[i] = pg[i]; }
  };
  auto put = [&](const float4 (&r)[4], int piece, float sc) {
    char* base = smem + piece * 8192;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint2 hi, lo;
      h2_split4(r[i].x * sc, r[i].y * sc, r[i].z * sc, r[i].w * sc, hi, lo);
      const int off = mixw_stage_off(sl, c80 + i);
      *reinterpret_cast<uint2*>(base + off) = hi;
      *reinterpret_cast<uint2*>(base + 4096 + off) = lo;
    }
  };
  // transposing-read address of this lane inside a piece (k_dft_analysis_h2's): rows 8g + q / 8g + 4 + q, chunk pp
  typedef s16x4v __attribute__((address_space(3))) * lds_tr;
  const int q = li >> 2, pp = li & 3;
  const int tsw = ((q >> 1) & 1) | ((g & 1) << 1);
  const int trow = (8 * g + q) * 128 + pp * 8;
  auto frag = [&](int piece, int hl, int ctile) {           // 8 lines (k = 8g + j) of channel 16 ctile + li
    const char* t = smem + piece * 8192 + hl * 4096 + trow + ((ctile ^ tsw) << 5);
    union { struct { s16x4v a, b; } h; f16x8 v; } u;
    u.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(t));
    u.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(t + 512));
    return u.v;
  };
  f32x4v cr[4], ci[4];
#pragma unroll
  for (int ot = 0; ot < 4; ++ot) { cr[ot] = (f32x4v){0.f, 0.f, 0.f, 0.f}; ci[ot] = (f32x4v){0.f, 0.f, 0.f, 0.f}; }
  if (t0 < t1) issue(t0);
  for (long t = t0; t < t1; ++t) {
    put(rs, part, sc_s);              // pieces 0 / 1: spectra re / im
    put(rg, 2 + part, sc_g);          // pieces 2 / 3: gradient spectra re / im
    if (t + 1 < t1) issue(t + 1);
    lds_barrier_mix();
    // A operands: this wave's 16 channels i of the saved spectra
    const f16x8 reh = frag(0, 0, w), rel = frag(0, 1, w), imh = frag(1, 0, w), iml = frag(1, 1, w);
    f16x8 nih, nil;
    {
      typedef unsigned u4v __attribute__((ext_vector_type(4)));
      union { f16x8 v; u4v u; } x, y;
      x.v = imh; y.v = iml;
      x.u ^= 0x80008000u; y.u ^= 0x80008000u;
      nih = x.v; nil = y.v;
    }
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) {
      const f16x8 grh = frag(2, 0, ot), grl = frag(2, 1, ot), gih = frag(3, 0, ot), gil = frag(3, 1, ot);
      cr[ot] = h2_mfma32(reh, rel, grh, grl, cr[ot]);
      cr[ot] = h2_mfma32(imh, iml, gih, gil, cr[ot]);
      ci[ot] = h2_mfma32(reh, rel, gih, gil, ci[ot]);
      ci[ot] = h2_mfma32(nih, nil, grh, grl, ci[ot]);
    }
    lds_barrier_mix();
  }
  // rows i = 16 w + 4 g + jj, column o = 16 ot + li: (gWr, gWi) pairs
  const float inv = iv_s * iv_g;
#pragma unroll
  for (int ot = 0; ot < 4; ++ot)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int i = 16 * w + 4 * g + jj, o = 16 * ot + li;
      *reinterpret_cast<float2*>(slab + (i * 64 + o) * 2) = make_float2(cr[ot][jj] * inv, ci[ot][jj] * inv);
    }
}

// gw[i][o][k][2] = sum_s slabs[s][axis][k][i][o][2]; zero for k >= keff
__global__ __launch_bounds__(256) void k_mix_wgrad_fold(const float* __restrict__ slabs, float* __restrict__ gw_y,
                                                        float* __restrict__ gw_x, int K, int keff, int kp, int S) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;        // (i * 64 + o) * K + k
  const int a = blockIdx.y;
  if (idx >= 4096L * K) return;
  const int k = (int)(idx % K);
  const long io = idx / K;
  float2 acc = make_float2(0.f, 0.f);
  if (k < keff)
    for (int s = 0; s < S; ++s) {
      const float2 v = *reinterpret_cast<const float2*>(slabs + ((((long)s * 2 + a) * kp + k) * 4096 + io) * 2);
      acc.x += v.x; acc.y += v.y;
    }
  float* gw = a ? gw_x : gw_y;
  if (gw) *reinterpret_cast<float2*>(gw + idx * 2) = acc;
}

size_t mix_wgrad_slab_floats(int kp, int S) { return (size_t)S * 2 * kp * 64 * 64 * 2; }

int mix_wgrad_h2(const float* spec_y, const float* spec_x, const float* gspec_y, const float* gspec_x, const float* amax_sy,
                 const float* amax_sx, const float* amax_gy, const float* amax_gx, float* gw_y, float* gw_x, long lines_y,
                 long lines_x, int K, int keff, int kp, float* slabs, int S, hipStream_t st) {
  MixWgP P;
  P.spec[0] = spec_y; P.spec[1] = spec_x; P.gspec[0] = gspec_y; P.gspec[1] = gspec_x;
  P.amax_s[0] = amax_sy; P.amax_s[1] = amax_sx; P.amax_g[0] = amax_gy; P.amax_g[1] = amax_gx;
  P.slabs = slabs; P.lines[0] = lines_y; P.lines[1] = lines_x; P.kp = kp; P.keff = keff; P.R = 2 * kp; P.S = S;
  hipLaunchKernelGGL(k_mix_wgrad_h2, dim3(keff, S, 2), dim3(256), 0, st, P);
  RPDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_mix_wgrad_fold, dim3((unsigned)((4096L * K + 255) / 256), 2), dim3(256), 0, st, slabs, gw_y, gw_x, K, keff, kp, S);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

#define RPDE_MIX_DISPATCH(R, ...)                                                           \
  do {                                                                                          \
    switch ((R) / 8) {                                                                          \
      case 1: hipLaunchKernelGGL((k_mix_h2<0, 1>), __VA_ARGS__); break;                         \
      case 2: hipLaunchKernelGGL((k_mix_h2<0, 2>), __VA_ARGS__); break;                         \
      case 3: hipLaunchKernelGGL((k_mix_h2<0, 3>), __VA_ARGS__); break;                         \
      case 4: hipLaunchKernelGGL((k_mix_h2<1, 0>), __VA_ARGS__); break;                         \
      case 5: hipLaunchKernelGGL((k_mix_h2<1, 1>), __VA_ARGS__); break;                         \
      default: hipLaunchKernelGGL((k_mix_h2<1, 2>), __VA_ARGS__); break;                        \
    }                                                                                           \
  } while (0)

// spectra of both axes -> operand blocks + inverse line scales, one launch
int mix_h2(const float* spec_y, const float* spec_x, const float* amax_y, const float* amax_x, void* img_y, void* img_x,
           float* inv_y, float* inv_x, long lines_y, long lines_x, int kp, const void* wimg, const float* wc, hipStream_t st) {
  MixP P;
  P.spec[0] = spec_y; P.spec[1] = spec_x; P.img[0] = (char*)img_y; P.img[1] = (char*)img_x;
  P.amax[0] = amax_y; P.amax[1] = amax_x; P.inv[0] = inv_y; P.inv[1] = inv_x;
  P.wimg = (const char*)wimg; P.wc = wc; P.kp = kp; P.R = 2 * kp; P.nq = kp / 4;
  int dev = 0, cus = 256;
  RPDE_HIP(hipGetDevice(&dev));
  RPDE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  // two workgroups per CU in one round: slices per (axis, quad) so that 2 * nq * slices ~ 2 * CUs
  const long lines[2] = {lines_y, lines_x};
  int nblk[2];
  for (int a = 0; a < 2; ++a) {
    P.tiles[a] = (int)(lines[a] / 16);
    int slices = cus / P.nq;
    if (slices < 1) slices = 1;
    if (slices > P.tiles[a]) slices = P.tiles[a];
    P.tps[a] = (P.tiles[a] + slices - 1) / slices;
    slices = (P.tiles[a] + P.tps[a] - 1) / P.tps[a];
    nblk[a] = slices * P.nq;
  }
  P.nblk0 = nblk[0];
  RPDE_MIX_DISPATCH(P.R, dim3(nblk[0] + nblk[1]), dim3(256), 0, st, P);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // namespace rpde
