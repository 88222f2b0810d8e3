// FeedForward 64 -> 256 -> 256 -> 64 (models/custom_layer.py:49-68 with dim 64, factor 4, three layers: the FFNO2D
// configuration) as ONE weight-stationary kernel in h2 arithmetic (h2.h).
//
// Everything is computed transposed (rows = features, columns = grid points), because then the accumulator of one
// layer IS the B operand of the next: a 16x16 accumulator tile holds feature rows 4g..4g+3 of point column (l & 15)
// in lane l, and two such tiles side by side are exactly the eight reduction entries a lane supplies to
// v_mfma_f32_16x16x32_f16 -- with the reduction index permuted, which the pre-arranged weight fragments absorb.
// So the hidden activations never leave the CU and never change lanes:
//
//   * a persistent 8-wave workgroup keeps ALL weights in registers for the whole launch: wave w owns rows
//     32w..32w+31 of W1 and W2 (2 output tiles each) and the reduction slice 32w..32w+31 of W3 -- 48 fragments,
//     192 VGPRs.  No weight traffic after the first microsecond.
//   * per tile of 32 points: GEMM1 (B = the input tile, split once and shared through LDS) -> bias, dropout,
//     GELU and GELU' in registers -> the wave's own slice of h1 goes to LDS as a ready fragment (8 KB per wave) ->
//     GEMM2 reads all eight slices -> activation -> the wave's slice of h2 stays in registers and feeds its slice of
//     GEMM3 -> the eight partial outputs are summed through LDS by one wave per 16 points, which also applies the
//     last dropout, LayerNorm, post-activation and the residual.  Three workgroup barriers per tile.
//   * training additionally stores h1, d1, h2, d2 (what rpde_feedforward_bwd consumes) and z3; evaluation stores
//     nothing but the output: 2 x 256 B per point instead of 4.6 KB.
//
// Operand scaling (f16 has a 5-bit exponent): weights carry one power-of-two scale per matrix; the input tile uses
// its own maximum; h1 and h2 use BOUNDS derived from it (|h| <= |u| <= max row L1 norm of W * max|x| + max|b|), so no
// reduction over activations is ever needed.  A bound that is 2^k too large costs k of the ~15 spare bits of the
// two-piece format, nothing else.
#include "ff_fused.h"
#include "cw.h"
#include "h2.h"
#include "pointwise.h"

namespace rpde {

#ifdef RPDE_STAMPS
// debug build: lane 0 of waves 0, 3 and 6 of workgroup 100 records s_memtime around the phases of its first tiles
__device__ unsigned long long g_ffstamps[3 * 64];
#define FFSTAMP(i) do { if (stamp_on && stamp_t < 8) g_ffstamps[stamp_w * 64 + stamp_t * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ unsigned long long g_fbstamps[3 * 64];      // the same for the backward chain kernel
#define FBSTAMP(i) do { if (stamp_on && stamp_t >= 0 && stamp_t < 8) g_fbstamps[stamp_w * 64 + stamp_t * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FFSTAMP(i) do { } while (0)
#define FBSTAMP(i) do { } while (0)
#endif

constexpr int FF_WAVES = 8;
constexpr int FF_FRAGS = 24;                 // per wave: W1 4, W2 16, W3 4
constexpr int FF_IMG_BYTES = FF_WAVES * FF_FRAGS * 2048;
constexpr int FF_NCONST = 16;
constexpr int FF_PREP_BLOCKS = FF_WAVES * FF_FRAGS * 64 / 1024;      // 12: one fragment item per thread

// reduction-slot -> hidden feature inside a 32-wide slice, for operands built from two accumulator tiles
__device__ __forceinline__ int ff_perm(int g, int j) { return 16 * (j >> 2) + 4 * g + (j & 3); }

// ---- weight preparation: FF_PREP_BLOCKS blocks; each finds the maxima (all of the 98 K weights: cheap, and no grid
// ---- synchronisation needed), then builds its share of the fragments in per-wave register order ----
__global__ __launch_bounds__(1024) void k_ff3_prep(const float* __restrict__ w1, const float* __restrict__ w2,
                                                   const float* __restrict__ w3, const float* __restrict__ b1,
                                                   const float* __restrict__ b2, char* __restrict__ img,
                                                   float* __restrict__ consts) {
  __shared__ float red[6][16];
  __shared__ float fin[6];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  // [0..2] max |W1|, |W2|, |W3|; [3] max row L1 of W1; [4] of W2; [5] max(|b1|), [6] max(|b2|) folded into 3,4 slots below
  float m1 = 0.f, m2 = 0.f, m3 = 0.f, r1 = 0.f, r2 = 0.f, bm = 0.f;
  // (16-byte loads, everything of a thread in flight at once: this launch sits in front of every forward, and its
  //  32 us were load latency, not work)
  {
    const float4* w14 = reinterpret_cast<const float4*>(w1);
    const float4* w34 = reinterpret_cast<const float4*>(w3);
    const float4* w24 = reinterpret_cast<const float4*>(w2);
    float4 a[4], c[4], b[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = w14[tid + 1024 * i]; c[i] = w34[tid + 1024 * i]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = w24[tid + 1024 * i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(a[i].x), fabsf(a[i].y)), fmaxf(fabsf(a[i].z), fabsf(a[i].w))));
      m3 = fmaxf(m3, fmaxf(fmaxf(fabsf(c[i].x), fabsf(c[i].y)), fmaxf(fabsf(c[i].z), fabsf(c[i].w))));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) m2 = fmaxf(m2, fmaxf(fmaxf(fabsf(b[i].x), fabsf(b[i].y)), fmaxf(fabsf(b[i].z), fabsf(b[i].w))));
  }
#pragma unroll
  for (int row = wv; row < 256; row += 16) {          // row L1 norms, one wave per row (coalesced); all 16 rows' loads in flight
    float a = fabsf(w1[row * 64 + l]);
    float b = (fabsf(w2[row * 256 + l]) + fabsf(w2[row * 256 + 64 + l])) + (fabsf(w2[row * 256 + 128 + l]) + fabsf(w2[row * 256 + 192 + l]));
    r1 = fmaxf(r1, wave_sum_dpp(a));
    r2 = fmaxf(r2, wave_sum_dpp(b));
  }
  if (tid < 256) {
    bm = fabsf(b1 ? b1[tid] : 0.f);
  } else if (tid < 512) {
    bm = fabsf(b2 ? b2[tid - 256] : 0.f);          // kept apart below: threads 256..511 carry |b2|
  }
  float v[6] = {m1, m2, m3, r1, r2, 0.f};
#pragma unroll
  for (int k = 0; k < 5; ++k) { v[k] = wave_max(v[k]); if (l == 0) red[k][wv] = v[k]; }
  const float bmw = wave_max(bm);
  if (l == 0) red[5][wv] = bmw;
  __syncthreads();
  if (tid < 5) { float a = 0.f; for (int i = 0; i < 16; ++i) a = fmaxf(a, red[tid][i]); fin[tid] = a; }
  if (tid == 5) { float a = 0.f; for (int i = 0; i < 4; ++i) a = fmaxf(a, red[5][i]); fin[5] = a; }       // |b1|: waves 0..3
  __shared__ float fin_b2;
  if (tid == 6) { float a = 0.f; for (int i = 4; i < 8; ++i) a = fmaxf(a, red[5][i]); fin_b2 = a; }       // |b2|: waves 4..7
  __syncthreads();
  float sc[3], iv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) h2_scale(fin[k], 0, sc[k], iv[k]);
  if (tid == 0 && blockIdx.x == 0) {
    consts[0] = iv[0]; consts[1] = iv[1]; consts[2] = iv[2];
    consts[3] = fin[3]; consts[4] = fin[4]; consts[5] = fin[5]; consts[6] = fin_b2;
  }
  // fragments: (wave, frag, lane) -> eight entries, scaled and split
  for (int it = blockIdx.x * 1024 + tid; it < FF_WAVES * FF_FRAGS * 64; it += gridDim.x * 1024) {
    const int ln = it & 63, f = (it >> 6) % FF_FRAGS, w = it / (64 * FF_FRAGS);
    const int g = ln >> 4, li = ln & 15;
    float x[8];
    if (f < 4) {                 // W1, natural reduction order: tile 2w + (f>>1), k-step f&1
      const int row = 16 * (2 * w + (f >> 1)) + li, k0 = 32 * (f & 1) + 8 * g;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = w1[row * 64 + k0 + j] * sc[0];
    } else if (f < 20) {         // W2, permuted: tile 2w + ((f-4)>>3), slice kappa = (f-4)&7
      const int row = 16 * (2 * w + ((f - 4) >> 3)) + li, kap = (f - 4) & 7;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = w2[row * 256 + 32 * kap + ff_perm(g, j)] * sc[1];
    } else {                     // W3, permuted: output tile f-20, slice kappa = w
      const int row = 16 * (f - 20) + li;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = w3[row * 256 + 32 * w + ff_perm(g, j)] * sc[2];
    }
    uint2 h0, l0, h1, l1;
    h2_split4(x[0], x[1], x[2], x[3], h0, l0);
    h2_split4(x[4], x[5], x[6], x[7], h1, l1);
    char* p = img + ((long)(w * FF_FRAGS + f)) * 2048 + ln * 16;
    *reinterpret_cast<uint4*>(p) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4*>(p + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
}

struct FF3P {
  const float* x; const float* res; float* out; float* z3;
  float* h1; float* d1; float* h2; float* d2;          // null in evaluation
  const char* wimg; const float* consts;
  const float* b1; const float* b2; const float* b3; const float* gamma; const float* beta;
  long P; int layer_norm; float eps; int post_act;
  DropCfg drop[3];
  int ntiles;
};

// LDS map (bytes)
constexpr int FF_SBUF = 0;                        // [2 blk][2 ks][hi|lo][1 KB]: input tile as B fragments       =  8 KB
constexpr int FF_RAW = FF_SBUF + 8192;            // [2 blk][4 pieces][1 KB]: fp32 of the NEXT input tile (LDS-DMA) =  8 KB
constexpr int FF_HBUF = FF_RAW + 8192;            // [2 blk][8 kappa][hi|lo][1 KB]: h1 slices                     = 32 KB
constexpr int FF_H2BUF = FF_HBUF + 32768;         // the same for h2                                              = 32 KB
constexpr int FF_W3 = FF_H2BUF + 32768;           // [4 tile][8 kappa][hi|lo][1 KB]: all of W3                    = 64 KB
constexpr int FF_VEC = FF_W3 + 65536;             // b1[256] b2[256] b3[64] gamma[64] beta[64] floats             = 2816 B
constexpr int FF_STAT = FF_VEC + 2816;            // [2 blk][4 tile][16 points][mean, M2] floats                  =  1 KB
constexpr int FF_INFO = FF_STAT + 1024;           // [2 parity][2 blk][16 points] floats: maximum of each input point
constexpr int FF_LDS = FF_INFO + 256;

// workgroup barrier that orders LDS traffic only: __syncthreads() would also wait for every outstanding global
// store (the h / d tensors a training forward writes) several times per tile
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// bias, dropout, GELU on four consecutive features of one point; u = dropout(z), h = gelu(u) and
// d = gelu'(u) * dropscale out (whatever the caller does not use is dead code)
__device__ __forceinline__ void ff_act4(const f32x4v acc, float inv, const float4 b, const DropCfg& drop, uint64_t id,
                                        float (&h)[4], float (&d)[4], float (&u)[4]) {
  float s[4] = {1.f, 1.f, 1.f, 1.f};
  if (drop.on()) drop_scale4(drop, id, s);
  const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    u[r] = fmaf(acc[r], inv, bb[r]) * s[r];
    act_both(RPDE_ACT_GELU, u[r], h[r], d[r]);
    d[r] *= s[r];
  }
}
// the backward side of the recompute mode: d = gelu'(u) * dropscale from the stored u and the regenerated mask
__device__ __forceinline__ float4 ff_dact4(const float4 u, const DropCfg& drop, uint64_t id) {
  float s[4] = {1.f, 1.f, 1.f, 1.f};
  if (drop.on()) drop_scale4(drop, id, s);
  return make_float4(dgelu_f(u.x) * s[0], dgelu_f(u.y) * s[1], dgelu_f(u.z) * s[2], dgelu_f(u.w) * s[3]);
}

// eight scaled activations of a lane (two accumulator tiles) -> one B fragment (hi, lo) at dst / dst + 1 KB
__device__ __forceinline__ void ff_put_frag(char* dst, const float (&v)[8]) {
  uint2 h0, l0, h1, l1;
  h2_split4(v[0], v[1], v[2], v[3], h0, l0);
  h2_split4(v[4], v[5], v[6], v[7], h1, l1);
  *reinterpret_cast<uint4*>(dst) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// MODE 0: evaluation (only `out` is written); 1: training, h and d = gelu'(u) * dropscale of both hidden layers stored
// (the per-GEMM backward's contract); 2: training, only u = dropout(z) stored (in the h buffers) -- the backward
// kernel and the weight-gradient kernels re-evaluate gelu' / gelu while the data passes through them, which halves
// this kernel's stores and the saved-for-backward footprint
template <int MODE>
__global__ __launch_bounds__(64 * FF_WAVES, 2) void k_ff3_fwd_h2(const FF3P A0) {
  constexpr bool TRAIN = MODE != 0;
  FF3P A = A0;                                                  // (the masks' device-side counter folded into the seeds)
#pragma unroll
  for (int i = 0; i < 3; ++i) A.drop[i] = drop_resolve(A0.drop[i]);
  __shared__ __attribute__((aligned(16))) char smem[FF_LDS];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, li = l & 15;
  // ---- W1, W2 slices -> registers; W3, biases, LayerNorm vectors -> LDS; once ----
  f16x8 w1h[2][2], w1l[2][2], w2h[2][8], w2l[2][8];
  {
    const char* base = A.wimg + (long)w * FF_FRAGS * 2048 + l * 16;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      w1h[f >> 1][f & 1] = *reinterpret_cast<const f16x8*>(base + f * 2048);
      w1l[f >> 1][f & 1] = *reinterpret_cast<const f16x8*>(base + f * 2048 + 1024);
    }
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      w2h[f >> 3][f & 7] = *reinterpret_cast<const f16x8*>(base + (4 + f) * 2048);
      w2l[f >> 3][f & 7] = *reinterpret_cast<const f16x8*>(base + (4 + f) * 2048 + 1024);
    }
    // the image holds W3 as [wave = kappa][tile]; LDS wants [tile][kappa]
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const uint4 hi = *reinterpret_cast<const uint4*>(base + (20 + f) * 2048);
      const uint4 lo = *reinterpret_cast<const uint4*>(base + (20 + f) * 2048 + 1024);
      *reinterpret_cast<uint4*>(smem + FF_W3 + (f * 8 + w) * 2048 + l * 16) = hi;
      *reinterpret_cast<uint4*>(smem + FF_W3 + (f * 8 + w) * 2048 + 1024 + l * 16) = lo;
    }
    float* vec = reinterpret_cast<float*>(smem + FF_VEC);
    for (int i = tid; i < 704; i += 64 * FF_WAVES) {
      float v;
      if (i < 256) v = A.b1 ? A.b1[i] : 0.f;
      else if (i < 512) v = A.b2 ? A.b2[i - 256] : 0.f;
      else if (i < 576) v = A.b3 ? A.b3[i - 512] : 0.f;
      else if (i < 640) v = A.gamma ? A.gamma[i - 576] : 1.f;
      else v = A.beta ? A.beta[i - 640] : 0.f;
      vec[i] = v;
    }
  }
  const float* const vec = reinterpret_cast<const float*>(smem + FF_VEC);
  const float winv1 = A.consts[0], winv2 = A.consts[1], winv3 = A.consts[2];
  const float c1 = A.consts[3], c2 = A.consts[4], b1max = A.consts[5], b2max = A.consts[6];
  float* const info = reinterpret_cast<float*>(smem + FF_INFO);
  float* const stat = reinterpret_cast<float*>(smem + FF_STAT);

  // input tile: wave `blk` (0, 1) owns 16 points.  Their 4 KB are fetched by LDS-DMA one tile ahead (each lane's four
  // 16-byte pieces land at [piece][lane], no registers held meanwhile) and turned into scaled f16 pieces afterwards.
  auto s_issue = [&](int tile) {
    const long p = min((long)tile * 32 + 16 * w + li, A.P - 1);
    const float* q = A.x + p * 64 + 8 * g;
    char* dst = smem + FF_RAW + w * 4096;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* glb_ptr;
    __builtin_amdgcn_global_load_lds((glb_ptr)(q), (lds_ptr)(dst), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)(q + 4), (lds_ptr)(dst + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)(q + 32), (lds_ptr)(dst + 2048), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr)(q + 36), (lds_ptr)(dst + 3072), 16, 0, 0);
  };
  auto s_convert = [&](int par) {
    float4 sn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) sn[i] = *reinterpret_cast<const float4*>(smem + FF_RAW + w * 4096 + i * 1024 + l * 16);
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(sn[i].x), "v"(sn[i].y));
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(sn[i].z), "v"(sn[i].w));
    }
    // one scale per POINT (a column of the B operand): the four lanes that hold a point's 64 channels agree on its
    // maximum.  Points of very different magnitude inside a block then keep their own relative precision, which
    // the per-point LayerNorm at the end would otherwise expose
    m = fmaxf(m, lane_xor16(m));
    m = fmaxf(m, lane_xor32(m));
    float sc, iv;
    h2_scale(m, 0, sc, iv);
    char* dst = smem + FF_SBUF + (w * 2) * 2048 + l * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float v[8] = {sn[2 * ks].x * sc, sn[2 * ks].y * sc, sn[2 * ks].z * sc, sn[2 * ks].w * sc,
                          sn[2 * ks + 1].x * sc, sn[2 * ks + 1].y * sc, sn[2 * ks + 1].z * sc, sn[2 * ks + 1].w * sc};
      ff_put_frag(dst + ks * 2048, v);
    }
    if (g == 0) info[(par * 2 + w) * 16 + li] = m;
  };
  if (w < 2 && (int)blockIdx.x < A.ntiles) {
    s_issue(blockIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_convert(0);
  }
  // role in the last layer: output tile t3 (16 features) of point block b3
  const int t3 = w & 3, b3 = w >> 2;

#ifdef RPDE_STAMPS
  const bool stamp_on = blockIdx.x == 100 && l == 0 && (w == 0 || w == 3 || w == 6);
  const int stamp_w = w == 0 ? 0 : (w == 3 ? 1 : 2);
  int stamp_t = -4;                      // skip the first four tiles (cold caches)
#endif
  int par = 0;
  for (int tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x, par ^= 1) {
    FFSTAMP(0);
    lds_barrier();                         // B0: input fragments of this tile (and, first time, W3 / vectors) are in LDS
    FFSTAMP(1);
    const int next_tile = tile + gridDim.x;
    if (w < 2 && next_tile < A.ntiles) { s_issue(next_tile); cw_mark<0>(); }
    const long p0 = (long)tile * 32;

    // per point (= per lane column): scales of x (its maximum), of h1 and h2 (bounds derived from it), recomputed
    // where needed from the point's maximum rather than kept in registers
    auto scales = [&](int blk, float& inv1, float& sh1, float& inv2, float& sh2, float& inv3) {
      const float smax = info[(par * 2 + blk) * 16 + li];      // (the other parity is being written for the next tile)
      float sx, sxi, a, ai;
      h2_scale(smax, 0, sx, sxi);
      inv1 = sxi * winv1;
      const float bound1 = fmaf(c1, smax, b1max) * A.drop[0].scale;
      h2_scale(bound1, 0, a, ai);
      sh1 = a; inv2 = ai * winv2;
      const float bound2 = fmaf(c2, bound1, b2max) * A.drop[1].scale;
      h2_scale(bound2, 0, a, ai);
      sh2 = a; inv3 = ai * winv3;
    };

    // (tried and dropped, same-box A/B: s_setprio(2) around the layer-2 matrix chain (evaluation -1.3 .. -2 %, training
    //  -2.6 % on one box and +0.9 % on another: not a robust gain; around every chain of the forward, backward and
    //  weight-gradient kernels: worse); different block orders for the two waves of a SIMD in layer 2 -- matrix,
    //  activation, matrix, activation against matrix, matrix, activation, activation -- 1.81 vs 1.82 ms in evaluation,
    //  2.74 vs 2.62 ms in training (5 spills); the activation pair in packed fp32 (v_pk_fma_f32 chains for two elements: 1.86
    //  vs 1.85 ms in evaluation, 2.78 vs 2.62 ms in training -- packed fp32 issues no faster here); a run without
    //  dropout (2.60 vs 2.62 ms: the counter hash is not what costs); issuing the accumulator chains of a layer interleaved (no faster: the phases
    //  are bound by the vector work of the activation; the extra live fragments spilled), and a two-barrier software
    //  pipeline across tiles -- layer 3 of tile i beside layer 1 of tile i+1 -- (1.75 vs 1.76 ms in evaluation: the two
    //  waves of a SIMD still run the same instruction mix at the same time, so matrix and vector work do not overlap))
    // ---- layer 1: this wave's 32 hidden features of both point blocks -> h1 slice w ----
    // (issuing the four accumulator chains interleaved was tried: no faster -- the phase is bound by the vector work
    //  of the activation -- and its extra live fragments spilled)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const char* sb = smem + FF_SBUF + (blk * 2) * 2048 + l * 16;
      const f16x8 xh0 = *reinterpret_cast<const f16x8*>(sb), xl0 = *reinterpret_cast<const f16x8*>(sb + 1024);
      const f16x8 xh1 = *reinterpret_cast<const f16x8*>(sb + 2048), xl1 = *reinterpret_cast<const f16x8*>(sb + 3072);
      const long pt = p0 + 16 * blk + li;
      float inv1, sh1, u0, u1, u2;
      scales(blk, inv1, sh1, u0, u1, u2);
      float hv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        acc = h2_mfma32(w1h[t][0], w1l[t][0], xh0, xl0, acc);
        acc = h2_mfma32(w1h[t][1], w1l[t][1], xh1, xl1, acc);
        float h[4], d[4], u[4];
        const int hid = 16 * (2 * w + t) + 4 * g;
        ff_act4(acc, inv1, *reinterpret_cast<const float4*>(vec + hid), A.drop[0], (uint64_t)(pt * 256 + hid), h, d, u);
        if (MODE == 1 && pt < A.P) {
#ifdef RPDE_EXP_CONTIG      // timing experiment: each store instruction writes 1 KB contiguous (fragment-major saved tensors)
          *reinterpret_cast<float4*>(A.h1 + p0 * 256 + ((blk * 8 + w) * 2 + t) * 256 + l * 4) = make_float4(h[0], h[1], h[2], h[3]);
          *reinterpret_cast<float4*>(A.d1 + p0 * 256 + ((blk * 8 + w) * 2 + t) * 256 + l * 4) = make_float4(d[0], d[1], d[2], d[3]);
#else
          *reinterpret_cast<float4*>(A.h1 + pt * 256 + hid) = make_float4(h[0], h[1], h[2], h[3]);
          *reinterpret_cast<float4*>(A.d1 + pt * 256 + hid) = make_float4(d[0], d[1], d[2], d[3]);
#endif
        }
        if (MODE == 2 && pt < A.P) *reinterpret_cast<float4*>(A.h1 + pt * 256 + hid) = make_float4(u[0], u[1], u[2], u[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[4 * t + r] = h[r] * sh1;
      }
      ff_put_frag(smem + FF_HBUF + ((blk * 8 + w) * 2) * 1024 + l * 16, hv);
    }
    FFSTAMP(2);
    lds_barrier();                         // B1: all eight slices of h1 are in LDS
    FFSTAMP(3);

    // residual of this wave's output tile: needed after the last layer, fetched now
    const long pt3 = p0 + 16 * b3 + li;
    const long off3 = min(pt3, A.P - 1) * 64 + 16 * t3 + 4 * g;
    float4 resv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (A.res) resv = *reinterpret_cast<const float4*>(A.res + off3);

    // ---- layer 2 (reads every slice of h1) -> h2 slice w ----
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kap = 0; kap < 8; ++kap) {
        const char* hb = smem + FF_HBUF + ((blk * 8 + kap) * 2) * 1024 + l * 16;
        const f16x8 bh = *reinterpret_cast<const f16x8*>(hb), bl = *reinterpret_cast<const f16x8*>(hb + 1024);
        acc[0] = h2_mfma32(w2h[0][kap], w2l[0][kap], bh, bl, acc[0]);
        acc[1] = h2_mfma32(w2h[1][kap], w2l[1][kap], bh, bl, acc[1]);
      }
      const long pt = p0 + 16 * blk + li;
      float u0, u1, inv2, sh2, u2;
      scales(blk, u0, u1, inv2, sh2, u2);
      float hv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float h[4], d[4], u[4];
        const int hid = 16 * (2 * w + t) + 4 * g;
        ff_act4(acc[t], inv2, *reinterpret_cast<const float4*>(vec + 256 + hid), A.drop[1], (uint64_t)(pt * 256 + hid), h, d, u);
        if (MODE == 2 && pt < A.P) *reinterpret_cast<float4*>(A.h2 + pt * 256 + hid) = make_float4(u[0], u[1], u[2], u[3]);
        if (MODE == 1 && pt < A.P) {
#ifdef RPDE_EXP_CONTIG
          *reinterpret_cast<float4*>(A.h2 + p0 * 256 + ((blk * 8 + w) * 2 + t) * 256 + l * 4) = make_float4(h[0], h[1], h[2], h[3]);
          *reinterpret_cast<float4*>(A.d2 + p0 * 256 + ((blk * 8 + w) * 2 + t) * 256 + l * 4) = make_float4(d[0], d[1], d[2], d[3]);
#else
          *reinterpret_cast<float4*>(A.h2 + pt * 256 + hid) = make_float4(h[0], h[1], h[2], h[3]);
          *reinterpret_cast<float4*>(A.d2 + pt * 256 + hid) = make_float4(d[0], d[1], d[2], d[3]);
#endif
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[4 * t + r] = h[r] * sh2;
      }
      ff_put_frag(smem + FF_H2BUF + ((blk * 8 + w) * 2) * 1024 + l * 16, hv);
    }
    FFSTAMP(4);
    lds_barrier();                         // B2: all eight slices of h2 are in LDS
    FFSTAMP(5);

    // ---- layer 3: output tile t3 of point block b3, both operands from LDS ----
    f32x4v acc3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kap = 0; kap < 8; ++kap) {
      const char* wb = smem + FF_W3 + (t3 * 8 + kap) * 2048 + l * 16;
      const char* hb = smem + FF_H2BUF + ((b3 * 8 + kap) * 2) * 1024 + l * 16;
      acc3 = h2_mfma32(*reinterpret_cast<const f16x8*>(wb), *reinterpret_cast<const f16x8*>(wb + 1024),
                       *reinterpret_cast<const f16x8*>(hb), *reinterpret_cast<const f16x8*>(hb + 1024), acc3);
    }
    if (w < 2 && next_tile < A.ntiles) {
      // the DMA of the next input tile was issued before this tile's stores (16 in mode 1, 8 in mode 2): it is older
      // than the youngest half of them, so a partial wait covers it without waiting for every store to be acknowledged
      // (cw.h; the ISA test checks the counts under the assumption stated above: a tile with a successor is full, so
      //  every lane-predicated store of it has live lanes and is issued)
      if (MODE == 1) cw_wait<0, 8>();
      else if (MODE == 2) cw_wait<0, 4>();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      s_convert(par ^ 1);
    }
    float v[4];
    {
      const float4 bb = *reinterpret_cast<const float4*>(vec + 512 + 16 * t3 + 4 * g);
      float u0, u1, u2, u3, inv3;
      scales(b3, u0, u1, u2, u3, inv3);
      v[0] = fmaf(acc3[0], inv3, bb.x); v[1] = fmaf(acc3[1], inv3, bb.y);
      v[2] = fmaf(acc3[2], inv3, bb.z); v[3] = fmaf(acc3[3], inv3, bb.w);
      if (TRAIN && pt3 < A.P) *reinterpret_cast<float4*>(A.z3 + off3) = make_float4(v[0], v[1], v[2], v[3]);
      if (A.drop[2].on()) {
        float s4[4];
        drop_scale4(A.drop[2], (uint64_t)(pt3 * 64 + 16 * t3 + 4 * g), s4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= s4[r];
      }
    }
    if (A.layer_norm) {
      // LayerNorm over 64 features = 4 waves x 4 lane groups x 4 registers: each wave reduces its 16 features of a
      // point to (mean, M2), the four are combined exactly (pairwise update of Chan et al.): as accurate as the
      // reference's two-pass variance, with one exchange through LDS
      float sum = (v[0] + v[1]) + (v[2] + v[3]);
      sum += lane_xor16(sum);
      sum += lane_xor32(sum);
      const float mw = sum * (1.f / 16.f);
      float m2 = (v[0] - mw) * (v[0] - mw) + (v[1] - mw) * (v[1] - mw) + (v[2] - mw) * (v[2] - mw) + (v[3] - mw) * (v[3] - mw);
      m2 += lane_xor16(m2);
      m2 += lane_xor32(m2);
      if (g == 0) *reinterpret_cast<float2*>(stat + ((b3 * 4 + t3) * 16 + li) * 2) = make_float2(mw, m2);
    }
    lds_barrier();                         // B3: per-wave statistics are in LDS (also: everyone is done with h2 / stats of the previous tile)
    if (A.layer_norm) {
      float mws[4], m2s = 0.f, mean = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float2 s2 = *reinterpret_cast<const float2*>(stat + ((b3 * 4 + t) * 16 + li) * 2);
        mws[t] = s2.x; m2s += s2.y; mean += s2.x;
      }
      mean *= 0.25f;
#pragma unroll
      for (int t = 0; t < 4; ++t) m2s += 16.f * (mws[t] - mean) * (mws[t] - mean);
      const float rstd = rsqrtf(m2s * (1.f / 64.f) + A.eps);
      const float4 gm = *reinterpret_cast<const float4*>(vec + 576 + 16 * t3 + 4 * g);
      const float4 bt = *reinterpret_cast<const float4*>(vec + 640 + 16 * t3 + 4 * g);
      v[0] = (v[0] - mean) * rstd * gm.x + bt.x; v[1] = (v[1] - mean) * rstd * gm.y + bt.y;
      v[2] = (v[2] - mean) * rstd * gm.z + bt.z; v[3] = (v[3] - mean) * rstd * gm.w + bt.w;
    }
    if (A.post_act) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = act_f(A.post_act, v[r]);
    }
    if (pt3 < A.P) *reinterpret_cast<float4*>(A.out + off3) = make_float4(v[0] + resv.x, v[1] + resv.y, v[2] + resv.z, v[3] + resv.w);
    FFSTAMP(6);
#ifdef RPDE_STAMPS
    ++stamp_t;
#endif
  }
}

// ============================================================================================================
// backward: LayerNorm / dropout / post-activation adjoint -> dz3, then the data-gradient chain
//   du2 = (W3^T dz3) * d2,  du1 = (W2^T du2) * d1
// in the same weight-stationary, transposed form (wave w owns hidden rows 32w..32w+31 of W3^T and W2^T).  The kernel
// reads g, z3, d2, d1 and writes dz3, du2, du1 -- the operands of the three weight-gradient GEMMs and of
// dx = du1 . W1, which follow as GEMMs; bias, gamma and beta gradients are summed per workgroup in LDS by
// single-writer updates (bitwise reproducible) and reduced over workgroups afterwards.
// d2 and d1 (half of the kernel's reads) arrive by LDS-DMA a whole tile ahead, 4 KB per wave each, no registers held:
// with register prefetch 72 % of the wave cycles were spent waiting for them (and the dx product, whose W1^T took
// the 64 KB of LDS the DMA areas need, moved out to a GEMM of its own).
// ============================================================================================================
__global__ __launch_bounds__(1024) void k_ff3_prep_bwd(const float* __restrict__ w1, const float* __restrict__ w2,
                                                       const float* __restrict__ w3, char* __restrict__ img,
                                                       float* __restrict__ consts) {
  __shared__ float red[5][16];
  __shared__ float fin[5];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6;
  float m1 = 0.f, m2 = 0.f, m3 = 0.f, r3 = 0.f, r2 = 0.f;
  {
    const float4* w14 = reinterpret_cast<const float4*>(w1);
    const float4* w34 = reinterpret_cast<const float4*>(w3);
    const float4* w24 = reinterpret_cast<const float4*>(w2);
    float4 a[4], c[4], b[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = w14[tid + 1024 * i]; c[i] = w34[tid + 1024 * i]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) b[i] = w24[tid + 1024 * i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(a[i].x), fabsf(a[i].y)), fmaxf(fabsf(a[i].z), fabsf(a[i].w))));
      m3 = fmaxf(m3, fmaxf(fmaxf(fabsf(c[i].x), fabsf(c[i].y)), fmaxf(fabsf(c[i].z), fabsf(c[i].w))));
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) m2 = fmaxf(m2, fmaxf(fmaxf(fabsf(b[i].x), fabsf(b[i].y)), fmaxf(fabsf(b[i].z), fabsf(b[i].w))));
  }
  {
    // column L1 norms (what bounds a row of the transposed products): four threads per column, a quarter of the
    // rows each, combined through LDS in fixed order
    __shared__ float colp[2][4][256];
    const int col = tid & 255, part = tid >> 8;
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int f = 0; f < 16; ++f) a += fabsf(w3[(16 * part + f) * 256 + col]);
    // (all 64 loads of a thread in flight at once: this launch sits in front of every backward)
    float wv2[64];
#pragma unroll
    for (int k = 0; k < 64; ++k) wv2[k] = w2[(64 * part + k) * 256 + col];
#pragma unroll
    for (int k = 0; k < 64; ++k) b += fabsf(wv2[k]);
    colp[0][part][col] = a; colp[1][part][col] = b;
    __syncthreads();
    if (tid < 256) {
      r3 = (colp[0][0][tid] + colp[0][1][tid]) + (colp[0][2][tid] + colp[0][3][tid]);
      r2 = (colp[1][0][tid] + colp[1][1][tid]) + (colp[1][2][tid] + colp[1][3][tid]);
    }
  }
  float v[5] = {m1, m2, m3, r3, r2};
#pragma unroll
  for (int k = 0; k < 5; ++k) { v[k] = wave_max(v[k]); if (l == 0) red[k][wv] = v[k]; }
  __syncthreads();
  if (tid < 5) { float a = 0.f; for (int i = 0; i < 16; ++i) a = fmaxf(a, red[tid][i]); fin[tid] = a; }
  __syncthreads();
  float sc[3], iv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) h2_scale(fin[k], 0, sc[k], iv[k]);
  if (tid == 0 && blockIdx.x == 0) { consts[0] = iv[0]; consts[1] = iv[1]; consts[2] = iv[2]; consts[3] = fin[3]; consts[4] = fin[4]; }
  for (int it = blockIdx.x * 1024 + tid; it < FF_WAVES * FF_FRAGS * 64; it += gridDim.x * 1024) {
    const int ln = it & 63, f = (it >> 6) % FF_FRAGS, w = it / (64 * FF_FRAGS);
    const int g = ln >> 4, li = ln & 15;
    float x[8];
    if (f < 4) {                 // W3^T: hidden tile 2w + (f>>1), reduction over the 64 output features, half f&1
      const int hid = 16 * (2 * w + (f >> 1)) + li;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = w3[(32 * (f & 1) + ff_perm(g, j)) * 256 + hid] * sc[2];
    } else if (f < 20) {         // W2^T: hidden-1 tile 2w + ((f-4)>>3), reduction slice kappa of hidden-2
      const int hid1 = 16 * (2 * w + ((f - 4) >> 3)) + li, kap = (f - 4) & 7;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = w2[(32 * kap + ff_perm(g, j)) * 256 + hid1] * sc[1];
    } else {                     // W1^T: input-feature tile f-20, reduction slice kappa = w of hidden-1
      const int feat = 16 * (f - 20) + li;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = w1[(32 * w + ff_perm(g, j)) * 64 + feat] * sc[0];
    }
    uint2 h0, l0, h1, l1;
    h2_split4(x[0], x[1], x[2], x[3], h0, l0);
    h2_split4(x[4], x[5], x[6], x[7], h1, l1);
    char* p = img + ((long)(w * FF_FRAGS + f)) * 2048 + ln * 16;
    *reinterpret_cast<uint4*>(p) = make_uint4(h0.x, h0.y, h1.x, h1.y);
    *reinterpret_cast<uint4*>(p + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
  }
}

constexpr int FFB_PART = 704;          // per workgroup: db1[256] db2[256] db3[64] dgamma[64] dbeta[64]
struct FF3B {
  const float* g; const float* z3; const float* d1; const float* d2;
  float* dz3; float* du2; float* du1; float* dx;
  const char* wimg; const float* consts;
  const float* gamma; const float* beta;
  float* part;
  long P; int layer_norm; float eps; int post_act;
  DropCfg drop2;
  DropCfg drop0, drop1;  // RECOMP only: masks of the hidden layers
  float dmax;            // bound of |gelu'| * dropout scale
  int ntiles;
};

constexpr int FB_DZBUF = 0;                        // [2 blk][2 ks][hi|lo][1 KB]                              =  8 KB
constexpr int FB_DUBUF = FB_DZBUF + 8192;          // [2 blk][8 kappa][hi|lo][1 KB]: du2 slices               = 32 KB
constexpr int FB_W3L = FB_DUBUF + 32768;           // [8 wave][4 frag][1 KB]: low pieces of the W3^T fragments  = 32 KB
constexpr int FB_D2 = FB_W3L + 32768;              // [8 wave][4 pieces][1 KB]: d2 of the wave's slice, next tile (LDS-DMA) = 32 KB
constexpr int FB_D1 = FB_D2 + 32768;               // the same for d1                                              = 32 KB
constexpr int FB_VEC = FB_D1 + 32768;              // gamma[64] beta[64]
constexpr int FB_ACC = FB_VEC + 512;               // db1[256] db2[256] | db3, dgamma, dbeta: [3][8 wave][64]         = 8 KB
constexpr int FB_INFO = FB_ACC + 8192;             // [2 blk][16 points] bound of |dz3|
constexpr int FB_LDS = FB_INFO + 128;

// sum over the 16 lanes of a row (one point block), valid in lane 15 of the row
__device__ __forceinline__ float row_sum15(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));
  return v;
}

// LDS reads / global loads the compiler does not see as such.  An ordinary read of an LDS-DMA landing area makes it wait
// for vmcnt(0) -- stores included, and again after every store in between -- and an ordinary load that is one tile in
// flight is answered with vmcnt(0) at its use, i.e. with a wait for the DMA issued just before.  Results pass through
// the wait statements as tied operands (what orders their uses behind the wait); ONE statement per wait, and the
// in-flight registers are followed through the ISA by tests/test_isa_pending_loads_cpu.py.
__device__ __forceinline__ f32x4v fb_lds_read_b128(unsigned addr) {
  f32x4v r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr) : "memory");
  return r;
}
__device__ __forceinline__ void fb_lds_wait2(f32x4v& a, f32x4v& b) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
}
__device__ __forceinline__ void fb_lds_write_b128(unsigned addr, f32x4v v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void fb_lds_write_b64(unsigned addr, uint2 v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void fb_lds_write_b32(unsigned addr, float v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4v fb_gload(const float* p) {
  f32x4v v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// sum / maximum over the 16 lanes of a row, in every lane (quad butterflies, then the two mirrors)
template <bool MAX>
__device__ __forceinline__ float row_all16(float v) {
#define RPDE_ROW_STEP(CTRL)                                                                               \
  {                                                                                                       \
    const float o_ = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); \
    v = MAX ? fmaxf(v, o_) : v + o_;                                                                      \
  }
  RPDE_ROW_STEP(0xB1)      // quad_perm [1,0,3,2]
  RPDE_ROW_STEP(0x4E)      // quad_perm [2,3,0,1]
  RPDE_ROW_STEP(0x141)     // row_half_mirror
  RPDE_ROW_STEP(0x140)     // row_mirror
#undef RPDE_ROW_STEP
  return v;
}

// RECOMP: A.d2 / A.d1 hold u = dropout(z) of the hidden layers (forward mode 2); the derivative factors are
// re-evaluated here from u and the regenerated dropout masks
template <bool RECOMP>
__global__ __launch_bounds__(64 * FF_WAVES, 2) void k_ff3_bwd_h2(const FF3B A0) {
  __shared__ __attribute__((aligned(16))) char smem[FB_LDS];
  FF3B A = A0;                                                  // (the masks' device-side counter folded into the seeds)
  A.drop2 = drop_resolve(A0.drop2); A.drop0 = drop_resolve(A0.drop0); A.drop1 = drop_resolve(A0.drop1);
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, li = l & 15;
  f16x8 w3h[2][2], w2h[2][8], w2l[2][8];
  {
    const char* base = A.wimg + (long)w * FF_FRAGS * 2048 + l * 16;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      w3h[f >> 1][f & 1] = *reinterpret_cast<const f16x8*>(base + f * 2048);
      *reinterpret_cast<uint4*>(smem + FB_W3L + (w * 4 + f) * 1024 + l * 16) = *reinterpret_cast<const uint4*>(base + f * 2048 + 1024);
    }
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      w2h[f >> 3][f & 7] = *reinterpret_cast<const f16x8*>(base + (4 + f) * 2048);
      w2l[f >> 3][f & 7] = *reinterpret_cast<const f16x8*>(base + (4 + f) * 2048 + 1024);
    }
    float* vecw = reinterpret_cast<float*>(smem + FB_VEC);
    float* accw = reinterpret_cast<float*>(smem + FB_ACC);
    for (int i = tid; i < 128; i += 64 * FF_WAVES) vecw[i] = i < 64 ? (A.gamma ? A.gamma[i] : 1.f) : (A.beta ? A.beta[i - 64] : 0.f);
    for (int i = tid; i < 2048; i += 64 * FF_WAVES) accw[i] = 0.f;
  }
  const float* const vec = reinterpret_cast<const float*>(smem + FB_VEC);
  float* const acc_db1 = reinterpret_cast<float*>(smem + FB_ACC);
  float* const acc_db2 = acc_db1 + 256;
  float* const acc_db3 = acc_db1 + 512;        // [8 wave][64]
  float* const acc_dg = acc_db1 + 1024;        // [8 wave][64]
  float* const acc_dbt = acc_db1 + 1536;       // [8 wave][64]
  float* const info = reinterpret_cast<float*>(smem + FB_INFO);
  const float winv1 = A.consts[0], winv2 = A.consts[1], winv3 = A.consts[2], c3t = A.consts[3], c2t = A.consts[4];
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  // role in the last-layer adjoint: one POINT per 16-lane row (point 4 w + q of the tile), lane li takes features
  // 4 li .. 4 li + 3 -- the LayerNorm statistics are sums over a row (DPP), no exchange between waves
  const int q3 = l >> 4, feat = 4 * li;

  // this wave's float4 of g and of z3 per lane, fetched one tile ahead into registers
  f32x4v g4n = {0.f, 0.f, 0.f, 0.f}, z4n = g4n;
  auto in_issue = [&](int tile) {
    const long p = min((long)tile * 32 + 4 * w + q3, A.P - 1);
    g4n = fb_gload(A.g + p * 64 + feat);
    z4n = fb_gload(A.z3 + p * 64 + feat);
  };
  // d2 / d1 of this wave's hidden slice for one tile: four 1 KB pieces (block, row tile), each lane's float4 at [piece][lane]
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;
  auto d_issue = [&](const float* d, int area, int tile) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float* q = d + min((long)tile * 32 + 16 * blk + li, A.P - 1) * 256 + 16 * (2 * w + t) + 4 * g;
        __builtin_amdgcn_global_load_lds((glb_ptr)(q), (lds_ptr)(smem + area + (w * 4 + blk * 2 + t) * 1024), 16, 0, 0);
      }
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the weights: from here on the queue is counted by hand)
  if ((int)blockIdx.x < A.ntiles) {
    in_issue(blockIdx.x);
    d_issue(A.d2, FB_D2, blockIdx.x);
    d_issue(A.d1, FB_D1, blockIdx.x);
  }
  lds_barrier();                               // vectors and zeroed accumulators are in LDS

  bool first = true;
#ifdef RPDE_STAMPS
  const bool stamp_on = blockIdx.x == 100 && l == 0 && (w == 0 || w == 3 || w == 6);
  const int stamp_w = w == 0 ? 0 : (w == 3 ? 1 : 2);
  int stamp_t = -4;
#endif
  for (int tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
    FBSTAMP(0);
    const int next_tile = tile + gridDim.x;
    const bool has_next = next_tile < A.ntiles;
    const long p0 = (long)tile * 32;
    const int ptl = 4 * w + q3;                    // this row's point of the tile
    const long pt3 = p0 + ptl;
    const bool live3 = pt3 < A.P;
    const long off3 = min(pt3, A.P - 1) * 64 + feat;
    // ---- last-layer adjoint: dropout, LayerNorm, post-activation (per row: no barrier, no exchange) ----
    // g / z3 of this tile were requested a tile ago; behind them, in program order, are the four du1 stores of that
    // tile (a tile that has a successor is full: all four are issued) and the four pieces of the d1 DMA -- first tile:
    // the eight DMA pieces of the prologue
    asm volatile("s_waitcnt vmcnt(8) ; landed %0 %1" : "+v"(g4n), "+v"(z4n)::"memory");
    const f32x4v g4 = g4n, z4 = z4n;
    float s4[4] = {1.f, 1.f, 1.f, 1.f};
    if (A.drop2.on()) drop_scale4(A.drop2, (uint64_t)(pt3 * 64 + feat), s4);
    const float tz[4] = {z4.x * s4[0], z4.y * s4[1], z4.z * s4[2], z4.w * s4[3]};
    float gy[4] = {g4.x, g4.y, g4.z, g4.w};
    if (!live3) { gy[0] = gy[1] = gy[2] = gy[3] = 0.f; }
    float dz[4], dzb;
    float dgm[4] = {0.f, 0.f, 0.f, 0.f}, dbt[4] = {0.f, 0.f, 0.f, 0.f};
    if (A.layer_norm) {
      const float mean = row_all16<false>((tz[0] + tz[1]) + (tz[2] + tz[3])) * (1.f / 64.f);
      const float m2 = row_all16<false>((tz[0] - mean) * (tz[0] - mean) + (tz[1] - mean) * (tz[1] - mean) +
                                        (tz[2] - mean) * (tz[2] - mean) + (tz[3] - mean) * (tz[3] - mean));
      const float rstd = rsqrtf(m2 * (1.f / 64.f) + A.eps);
      f32x4v gm = fb_lds_read_b128(lds0 + FB_VEC + feat * 4);
      f32x4v bt = fb_lds_read_b128(lds0 + FB_VEC + 256 + feat * 4);
      fb_lds_wait2(gm, bt);
      const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
      float xh[4], dxh[4], s1 = 0.f, s2 = 0.f, am = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xh[k] = (tz[k] - mean) * rstd;
        float dy = gy[k];
        if (A.post_act) dy *= dact_f(A.post_act, xh[k] * gmv[k] + btv[k]);
        dgm[k] = dy * xh[k];
        dbt[k] = dy;
        dxh[k] = dy * gmv[k];
        s1 += dxh[k];
        s2 += dxh[k] * xh[k];
        am = fmaxf(am, fabsf(dxh[k]));
      }
      const float S1 = row_all16<false>(s1) * (1.f / 64.f), S2 = row_all16<false>(s2) * (1.f / 64.f);
      const float AM = row_all16<true>(am);
#pragma unroll
      for (int k = 0; k < 4; ++k) dz[k] = rstd * (dxh[k] - S1 - xh[k] * S2) * s4[k];
      // bound of this POINT's |dz3|: |xhat| <= sqrt(63) < 8
      dzb = rstd * (AM + fabsf(S1) + 8.f * fabsf(S2)) * A.drop2.scale;
    } else {
      float am = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { dz[k] = gy[k] * dact_f(A.post_act, tz[k]) * s4[k]; am = fmaxf(am, fabsf(dz[k])); }
      dzb = row_all16<true>(am);
    }
    if (live3) *reinterpret_cast<float4*>(A.dz3 + off3) = make_float4(dz[0], dz[1], dz[2], dz[3]);
    // bias / gamma / beta gradients: sum over the wave's four points (rows), one writer per (wave, feature); the
    // eight waves' sums are combined in fixed order at the end
    {
      float b[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        b[k] = live3 ? dz[k] : 0.f;
        b[k] += lane_xor16(b[k]); b[k] += lane_xor32(b[k]);
        if (A.layer_norm) {
          dgm[k] += lane_xor16(dgm[k]); dgm[k] += lane_xor32(dgm[k]);
          dbt[k] += lane_xor16(dbt[k]); dbt[k] += lane_xor32(dbt[k]);
        }
      }
      if (q3 == 0) {
        // (asm: an ordinary access here is preceded by vmcnt(0) -- the compiler cannot tell these addresses from the
        //  DMA's landing areas -- i.e. by a wait for the d1 pieces requested a moment ago)
        const unsigned a0 = lds0 + FB_ACC + (512 + w * 64 + feat) * 4;
        f32x4v v3 = fb_lds_read_b128(a0), vg = fb_lds_read_b128(a0 + 2048), vb = fb_lds_read_b128(a0 + 4096);
        fb_lds_wait2(v3, vg);
        fb_lds_wait2(vb, vb);
        v3.x += b[0]; v3.y += b[1]; v3.z += b[2]; v3.w += b[3];
        fb_lds_write_b128(a0, v3);
        if (A.layer_norm) {
          vg.x += dgm[0]; vg.y += dgm[1]; vg.z += dgm[2]; vg.w += dgm[3];
          vb.x += dbt[0]; vb.y += dbt[1]; vb.z += dbt[2]; vb.w += dbt[3];
          fb_lds_write_b128(a0 + 2048, vg);
          fb_lds_write_b128(a0 + 4096, vb);
        }
      }
    }
    // dz3 -> B fragments (one scale per point, its bound dzb): features 4 li .. 4 li + 3 of point ptl are reduction slots
    // 4 (li >> 2 & 1) .. + 3 of lane group li & 3 in the fragment (block ptl >> 4, half ks = li >> 3), point column ptl & 15
    {
      float sdz, sdzi;
      h2_scale(dzb, 0, sdz, sdzi);
      uint2 hi, lo;
      h2_split4(dz[0] * sdz, dz[1] * sdz, dz[2] * sdz, dz[3] * sdz, hi, lo);
      // (the 16-byte unit of fragment lane (c, p) = (li & 3, ptl & 15) sits at unit 16 c + (p ^ c ^ 4 ks) of its 1 KB piece:
      //  written plainly at 16 c + p, the sixteen lanes of a row -- same p, four c, two ks, 256 / 2048 bytes apart -- hit
      //  two banks (8-way conflicts: 29.4 M conflict cycles per launch in round 3's counters); permuted inside each
      //  256-byte row they cover all 32.  The reader un-permutes its address; its own banking only sees a row's units
      //  in another order.)
      const int ksz = li >> 3, cz = li & 3;
      const unsigned dst = lds0 + FB_DZBUF + (((ptl >> 4) * 2 + ksz) * 2) * 1024 + (cz * 16 + ((ptl & 15) ^ cz ^ (4 * ksz))) * 16 +
                           ((li >> 2) & 1) * 8;
      fb_lds_write_b64(dst, hi);
      fb_lds_write_b64(dst + 1024, lo);
      if (li == 0) fb_lds_write_b32(lds0 + FB_INFO + ptl * 4, dzb);
    }
    lds_barrier();                                                                    // C: dz3 fragments + bounds in LDS
    FBSTAMP(3);

    // per point: scale of dz3 (its bound), of du2 and du1 (bounds derived from it); recomputed at each use
    auto bscales = [&](int blk, float& inv_a, float& s_du2, float& inv_b, float& s_du1, float& inv_c) {
      const float bz = info[blk * 16 + li];
      float a, ai;
      h2_scale(bz, 0, a, ai);
      inv_a = ai * winv3;
      const float b2 = c3t * bz * A.dmax;
      h2_scale(b2, 0, a, ai);
      s_du2 = a; inv_b = ai * winv2;
      const float b1 = c2t * b2 * A.dmax;
      h2_scale(b1, 0, a, ai);
      s_du1 = a; inv_c = ai * winv1;
    };

    // ---- du2 = (W3^T dz3) * d2: this wave's 32 hidden rows ----
    // the d2 pieces of this tile were requested a tile ago.  vmcnt(N) = all but the N youngest operations are done;
    // younger than that DMA are, in program order, the next tile's g / z3 loads (2), the du1 stores (4), the d1 DMA
    // (4) and this tile's dz3 store: 10 or 11 (first tile: only the d1 DMA and the dz3 store)
    if (first) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    FBSTAMP(4);
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      f32x4v dq[2];                                // the wave's two pieces [t] of d2 for this block
      dq[0] = fb_lds_read_b128(lds0 + FB_D2 + (w * 4 + blk * 2) * 1024 + l * 16);
      dq[1] = fb_lds_read_b128(lds0 + FB_D2 + (w * 4 + blk * 2 + 1) * 1024 + l * 16);
      fb_lds_wait2(dq[0], dq[1]);
      const int zu = (l & 48) | ((l & 15) ^ (l >> 4));             // this lane's unit in the ks = 0 pieces; ks = 1: ^ 4
      const char* zb = smem + FB_DZBUF + (blk * 2) * 2048;
      const f16x8 zh0 = *reinterpret_cast<const f16x8*>(zb + zu * 16), zl0 = *reinterpret_cast<const f16x8*>(zb + 1024 + zu * 16);
      const f16x8 zh1 = *reinterpret_cast<const f16x8*>(zb + 2048 + (zu ^ 4) * 16), zl1 = *reinterpret_cast<const f16x8*>(zb + 3072 + (zu ^ 4) * 16);
      const long pt = p0 + 16 * blk + li;
      float inv_a, s_du2, q0, q1, q2;
      bscales(blk, inv_a, s_du2, q0, q1, q2);
      float hv[8];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4v acc = {0.f, 0.f, 0.f, 0.f};
        const f16x8 w3l0 = *reinterpret_cast<const f16x8*>(smem + FB_W3L + (w * 4 + 2 * t) * 1024 + l * 16);
        const f16x8 w3l1 = *reinterpret_cast<const f16x8*>(smem + FB_W3L + (w * 4 + 2 * t + 1) * 1024 + l * 16);
        acc = h2_mfma32(w3h[t][0], w3l0, zh0, zl0, acc);
        acc = h2_mfma32(w3h[t][1], w3l1, zh1, zl1, acc);
        const int hid = 16 * (2 * w + t) + 4 * g;
        float4 d = make_float4(dq[t].x, dq[t].y, dq[t].z, dq[t].w);
        if (RECOMP) d = ff_dact4(d, A.drop1, (uint64_t)(pt * 256 + hid));
        const float u[4] = {acc[0] * inv_a * d.x, acc[1] * inv_a * d.y, acc[2] * inv_a * d.z, acc[3] * inv_a * d.w};
        if (pt < A.P) {
          *reinterpret_cast<float4*>(A.du2 + pt * 256 + hid) = make_float4(u[0], u[1], u[2], u[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) bsum[4 * t + r] += u[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[4 * t + r] = u[r] * s_du2;
      }
      {                                            // (ff_put_frag through asm writes: see fb_lds_read_b128)
        uint2 h0, l0, h1, l1;
        h2_split4(hv[0], hv[1], hv[2], hv[3], h0, l0);
        h2_split4(hv[4], hv[5], hv[6], hv[7], h1, l1);
        const unsigned dst = lds0 + FB_DUBUF + ((blk * 8 + w) * 2) * 1024 + l * 16;
        f32x4v ph, pl;
        ph.x = __uint_as_float(h0.x); ph.y = __uint_as_float(h0.y); ph.z = __uint_as_float(h1.x); ph.w = __uint_as_float(h1.y);
        pl.x = __uint_as_float(l0.x); pl.y = __uint_as_float(l0.y); pl.z = __uint_as_float(l1.x); pl.w = __uint_as_float(l1.y);
        fb_lds_write_b128(dst, ph);
        fb_lds_write_b128(dst + 1024, pl);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) bsum[i] = row_sum15(bsum[i]);
    if (li == 15) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc_db2[16 * (2 * w + (i >> 2)) + 4 * g + (i & 3)] += bsum[i];
    }
    if (has_next) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the d2 pieces have been read: their area is free
      d_issue(A.d2, FB_D2, next_tile);
      in_issue(next_tile);
    }
    FBSTAMP(5);
    lds_barrier();                                                                    // D: du2 slices in LDS
    FBSTAMP(6);

    // ---- du1 = (W2^T du2) * d1 ----
#pragma unroll
    for (int i = 0; i < 8; ++i) bsum[i] = 0.f;
    {
      f32x4v acc[2][2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        acc[blk][0] = (f32x4v){0.f, 0.f, 0.f, 0.f}; acc[blk][1] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kap = 0; kap < 8; ++kap) {
          const char* hb = smem + FB_DUBUF + ((blk * 8 + kap) * 2) * 1024 + l * 16;
          const f16x8 bh = *reinterpret_cast<const f16x8*>(hb), bl = *reinterpret_cast<const f16x8*>(hb + 1024);
          acc[blk][0] = h2_mfma32(w2h[0][kap], w2l[0][kap], bh, bl, acc[blk][0]);
          acc[blk][1] = h2_mfma32(w2h[1][kap], w2l[1][kap], bh, bl, acc[blk][1]);
        }
      }
      // the d1 pieces of this tile: younger than their DMA are the dz3 store, the four du2 stores and -- when there is
      // a next tile -- its d2 DMA (4) and g / z3 loads (2): 11; in the last tile only the stores of live points are
      // certain (at least two du2 stores: block 0 has a live point)
      if (has_next) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      FBSTAMP(7);
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        f32x4v dq[2];
        dq[0] = fb_lds_read_b128(lds0 + FB_D1 + (w * 4 + blk * 2) * 1024 + l * 16);
        dq[1] = fb_lds_read_b128(lds0 + FB_D1 + (w * 4 + blk * 2 + 1) * 1024 + l * 16);
        fb_lds_wait2(dq[0], dq[1]);
        const long pt = p0 + 16 * blk + li;
        float q0, q1, inv_b, s_du1, q2;
        bscales(blk, q0, q1, inv_b, s_du1, q2);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int hid = 16 * (2 * w + t) + 4 * g;
          float4 d = make_float4(dq[t].x, dq[t].y, dq[t].z, dq[t].w);
          if (RECOMP) d = ff_dact4(d, A.drop0, (uint64_t)(pt * 256 + hid));
          const float u[4] = {acc[blk][t][0] * inv_b * d.x, acc[blk][t][1] * inv_b * d.y,
                              acc[blk][t][2] * inv_b * d.z, acc[blk][t][3] * inv_b * d.w};
          if (pt < A.P) {
            *reinterpret_cast<float4*>(A.du1 + pt * 256 + hid) = make_float4(u[0], u[1], u[2], u[3]);
#pragma unroll
            for (int r = 0; r < 4; ++r) bsum[4 * t + r] += u[r];
          }
        }
      }
    }
    if (has_next) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      d_issue(A.d1, FB_D1, next_tile);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) bsum[i] = row_sum15(bsum[i]);
    if (li == 15) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc_db1[16 * (2 * w + (i >> 2)) + 4 * g + (i & 3)] += bsum[i];
    }
    first = false;
#ifdef RPDE_STAMPS
    ++stamp_t;
#endif
  }
  // ---- per-workgroup sums out ----
  lds_barrier();
  {
    float* part = A.part + (long)blockIdx.x * FFB_PART;
    for (int i = tid; i < FFB_PART; i += 64 * FF_WAVES) {
      float v;
      if (i < 512) v = acc_db1[i];                                   // db1, db2
      else {                                                         // db3 | dgamma | dbeta: the eight waves' sums, in order
        const float* a8 = acc_db3 + ((i - 512) >> 6) * 512 + ((i - 512) & 63);
        v = 0.f;
#pragma unroll
        for (int ww = 0; ww < FF_WAVES; ++ww) v += a8[ww * 64];
      }
      part[i] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
bool ff3_fused_ok(const rpde_ff_params* p, long P) {
  if (const char* e = getenv("RPDE_FUSED_FF")) if (e[0] == '0') return false;
  // (the preparation kernels read the weight matrices 16 bytes at a time: a view with an odd storage offset takes the
  //  per-GEMM path)
  for (int l = 0; l < 3 && p->weights; ++l)
    if (reinterpret_cast<uintptr_t>(p->weights[l]) & 15) return false;
  return p->n_layers == 3 && p->dim == 64 && p->factor == 4 && P >= 32;
}
size_t ff3_fused_ws_floats() { return (FF_IMG_BYTES + FF_NCONST * 4 + 3) / 4; }

// the weight image + constants of the forward kernel (ff3_fused_ws_floats() floats at ws)
int ff3_fused_prepare(const rpde_ff_params* p, void* ws, hipStream_t st) {
  char* img = static_cast<char*>(ws);
  float* consts = reinterpret_cast<float*>(img + FF_IMG_BYTES);
  const float* b1 = p->biases ? p->biases[0] : nullptr;
  const float* b2 = p->biases ? p->biases[1] : nullptr;
  hipLaunchKernelGGL(k_ff3_prep, dim3(FF_PREP_BLOCKS), dim3(1024), 0, st, p->weights[0], p->weights[1], p->weights[2], b1, b2, img, consts);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// prepared: ws already holds what ff3_fused_prepare wrote for these weights (evaluation with frozen weights)
int ff3_fused_fwd(const rpde_ff_params* p, const float* x, const float* residual, float* const* hs, float* const* ds,
                  float* z_last, float* out, long P, void* ws, hipStream_t st, bool prepared) {
  char* img = static_cast<char*>(ws);
  float* consts = reinterpret_cast<float*>(img + FF_IMG_BYTES);
  const float* b1 = p->biases ? p->biases[0] : nullptr;
  const float* b2 = p->biases ? p->biases[1] : nullptr;
  const float* b3 = p->biases ? p->biases[2] : nullptr;
  if (!prepared) RPDE_TRY(ff3_fused_prepare(p, ws, st));
  FF3P A;
  memset(&A, 0, sizeof(A));
  const bool have_h = hs && hs[0] && hs[1], have_d = ds && ds[0] && ds[1];
  const int mode = have_h ? (have_d ? 1 : 2) : 0;
  A.x = x; A.res = residual; A.out = out; A.z3 = z_last;
  if (mode) { A.h1 = hs[0]; A.h2 = hs[1]; }
  if (mode == 1) { A.d1 = ds[0]; A.d2 = ds[1]; }
  A.wimg = img; A.consts = consts; A.b1 = b1; A.b2 = b2; A.b3 = b3; A.gamma = p->ln_gamma; A.beta = p->ln_beta;
  A.P = P; A.layer_norm = p->layer_norm; A.eps = p->ln_eps; A.post_act = p->post_act;
  for (int l = 0; l < 3; ++l) {
    // same (seed, layer) -> mask mapping as the per-GEMM path, so that rpde_feedforward_bwd regenerates the same masks
    uint64_t z = p->seed + 0x9E3779B97F4A7C15ull * (uint64_t)(l + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    A.drop[l] = make_drop(p->dropout_p, z ^ (z >> 31), p->seed_epoch);
  }
  A.ntiles = (int)((P + 31) / 32);
  int dev = 0, cus = 256;
  RPDE_HIP(hipGetDevice(&dev));
  RPDE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = A.ntiles < cus ? A.ntiles : cus;
  if (mode == 1) hipLaunchKernelGGL(k_ff3_fwd_h2<1>, dim3(grid), dim3(64 * FF_WAVES), 0, st, A);
  else if (mode == 2) hipLaunchKernelGGL(k_ff3_fwd_h2<2>, dim3(grid), dim3(64 * FF_WAVES), 0, st, A);
  else hipLaunchKernelGGL(k_ff3_fwd_h2<0>, dim3(grid), dim3(64 * FF_WAVES), 0, st, A);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

size_t ff3_fused_bwd_part_floats() { return (size_t)1024 * FFB_PART; }

// prep + the fused kernel: dz3 [P,64], du2, du1 [P,256], dx [P,64] (may be null), part [grid][704]; returns the grid
int ff3_fused_bwd_launch(const rpde_ff_params* p, const float* const* ds, int recompute, const float* z_last,
                         const float* grad_out, float* dz3, float* du2, float* du1, float* dx, float* part, int* grid_out,
                         long P, void* ws, hipStream_t st) {
  char* img = static_cast<char*>(ws);
  float* consts = reinterpret_cast<float*>(img + FF_IMG_BYTES);
  hipLaunchKernelGGL(k_ff3_prep_bwd, dim3(FF_PREP_BLOCKS), dim3(1024), 0, st, p->weights[0], p->weights[1], p->weights[2], img, consts);
  RPDE_LAUNCH_CHECK();
  FF3B A;
  memset(&A, 0, sizeof(A));
  A.g = grad_out; A.z3 = z_last; A.d1 = ds[0]; A.d2 = ds[1];
  A.dz3 = dz3; A.du2 = du2; A.du1 = du1; A.dx = dx;
  A.wimg = img; A.consts = consts; A.gamma = p->ln_gamma; A.beta = p->ln_beta; A.part = part;
  A.P = P; A.layer_norm = p->layer_norm; A.eps = p->ln_eps; A.post_act = p->post_act;
  DropCfg dc[3];
  for (int l = 0; l < 3; ++l) {            // the forward's (seed, layer) -> mask mapping
    uint64_t z = p->seed + 0x9E3779B97F4A7C15ull * (uint64_t)(l + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    dc[l] = make_drop(p->dropout_p, z ^ (z >> 31), p->seed_epoch);
  }
  A.drop0 = dc[0]; A.drop1 = dc[1]; A.drop2 = dc[2];
  A.dmax = 1.13f * A.drop2.scale;          // |gelu'| <= 1.129, times the dropout scale folded into d
  A.ntiles = (int)((P + 31) / 32);
  int dev = 0, cus = 256;
  RPDE_HIP(hipGetDevice(&dev));
  RPDE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  if (cus > 1024) cus = 1024;
  const int grid = A.ntiles < cus ? A.ntiles : cus;
  if (recompute) hipLaunchKernelGGL(k_ff3_bwd_h2<true>, dim3(grid), dim3(64 * FF_WAVES), 0, st, A);
  else hipLaunchKernelGGL(k_ff3_bwd_h2<false>, dim3(grid), dim3(64 * FF_WAVES), 0, st, A);
  RPDE_LAUNCH_CHECK();
  *grid_out = grid;
  return RPDE_OK;
}

}  // namespace rpde

#ifdef RPDE_STAMPS
extern "C" int rpde_debug_ffb_stamps(unsigned long long* host_out) {
  RPDE_HIP(hipDeviceSynchronize());
  RPDE_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(rpde::g_fbstamps), sizeof(unsigned long long) * 3 * 64));
  return RPDE_OK;
}
extern "C" int rpde_debug_ff_stamps(unsigned long long* host_out) {
  RPDE_HIP(hipDeviceSynchronize());
  RPDE_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(rpde::g_ffstamps), sizeof(unsigned long long) * 3 * 64));
  return RPDE_OK;
}
#endif
