// Weight gradient of a pointwise linear layer, gW[out, in] = sum_p gy[p, out] * h[p, in], for the FeedForward of the
// headline configuration (reference: autograd of models/custom_layer.py:58 nn.Linear inside FeedForward): a product
// whose reduction runs over the P = B*M*N grid points (2.1 M at B = 32, 256^2) and whose output is tiny.  It streams
// both operands exactly once, so it is HBM bound as soon as the matrix work fits under the stream -- which the
// generic six-term bf16 split (gemm_bf16x3.hip, 1.59 ms for 4.3 GB at 256x256) does not achieve and the three-term f16
// split of h2.h does.
//
// One workgroup of eight waves owns the whole [out x in] output (up to 256 x 256: 128 accumulator registers per lane)
// and one contiguous chunk of the points, so every operand byte is read from HBM exactly once; 256 workgroups fill
// the chip once, the per-chunk results go to slabs that reduce_slabs folds in fixed order (no float atomics:
// run-to-run identical).  (A first version gave 128 output rows to each of two workgroups per chunk: the second
// read of the other operand did not stay in L2 and the kernel ran at the HBM roof of 1.5x the bytes, 1.25 ms instead
// of 0.9 ms at 256 x 256.)
//
// Per step of 32 points: the operands arrive as [32 points][64 channels] fp32 panels (256-byte rows: coalesced), one
// panel per loader wave, which finds the panel's maximum (DPP wave reduction), scales by a power of two, splits into
// hi / lo f16 pieces and stores them in memory order; both MFMA operands are then fetched with the transposing LDS
// read (ds_read_b64_tr_b16: the reduction index, the point, becomes the fragment's k).  Double-buffered LDS stages,
// one barrier per step, global loads one step ahead in registers.
//
// Scaling.  f16 pieces need the data near 2^14; a sum over two million points is dominated by its largest terms, so
// precision relative to the largest magnitude seen so far is what the result needs.  Each panel column block keeps a
// RUNNING exponent (the maximum over the steps so far, owned by its loader wave and published per stage); operands
// are scaled by it, and a consumer wave whose panels' exponents moved rescales its accumulators once (a wave-uniform
// branch that is taken a handful of times per launch).  No pass over the data to find a global maximum, no
// per-step accumulator arithmetic, robust to isolated huge points.
#include <stdio.h>
#include "h2.h"
#include "pointwise.h"
#include "wgrad_h2.h"

#include <stdlib.h>

namespace rpde {

constexpr int WG_WAVES = 8;
constexpr int WG_PANEL = 8192;                 // bytes per panel stage: [hi | lo][32 points][64 channels] f16
constexpr int WG_BLOCKS = 256;                 // one workgroup per CU

// byte offset of 8-byte chunk c8 (channels 4 c8 .. 4 c8 + 3) of point-row k in one piece of a panel (128-byte rows);
// XOR swizzle so that the transposing reads (rows 8g+q and 8g+4+q per 16-lane group) cover all banks
__device__ __forceinline__ int wg_off(int k, int c8) {
  return k * 128 + ((c8 ^ ((((k >> 1) & 1) << 2) | (((k >> 3) & 1) << 3))) << 3);
}

// LDS traffic done + workgroup barrier, without waiting for outstanding global loads (as __syncthreads() would)
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct WgP {
  const float* a;      // gy  [P, M]
  const float* b;      // h   [P, N]
  float* slabs;        // [nchunk][M][N]
  long npts;           // P
  long steps;          // ceil(P / 32): rows past P read as zeros
  int M, N, nchunk;
  const float* w;      // DG only: the layer's weight [M, N]
  float* gx;           // DG only: gx[P, N] = gy[P, M] . w
  int early;           // bit w: wave w converts the next step's panel BEFORE this step's products
};

// 4 x 4 transpose inside every quad of lanes: afterwards register c of lane p (p = lane & 3) holds what register p of
// lane c held (two butterfly stages on DPP quad permutes; the synthesis kernel's output uses the same)
template <int CTRL>
__device__ __forceinline__ float wg_quad_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void wg_quad_transpose(float& v0, float& v1, float& v2, float& v3) {
  const bool b0 = threadIdx.x & 1, b1 = threadIdx.x & 2;
  { const float a = wg_quad_dpp<0xB1>(v0), c = wg_quad_dpp<0xB1>(v1); v1 = b0 ? v1 : a; v0 = b0 ? c : v0; }   // quad_perm [1,0,3,2]
  { const float a = wg_quad_dpp<0xB1>(v2), c = wg_quad_dpp<0xB1>(v3); v3 = b0 ? v3 : a; v2 = b0 ? c : v2; }
  { const float a = wg_quad_dpp<0x4E>(v0), c = wg_quad_dpp<0x4E>(v2); v2 = b1 ? v2 : a; v0 = b1 ? c : v0; }   // quad_perm [2,3,0,1]
  { const float a = wg_quad_dpp<0x4E>(v1), c = wg_quad_dpp<0x4E>(v3); v3 = b1 ? v3 : a; v1 = b1 ? c : v1; }
}

// PA / PB: 64-channel panels of the A block / of B; waves WM x WN, each TM x TN tiles of 16 x 16; ACT: B = gelu(b).
// DG: the layer's data gradient rides along -- gx[p, :] = gy[p, :] . W for the 32 points of every step, from the gy
// panels that are in LDS anyway (plain 16-byte reads: here the reduction index is the channel), so gy is read from
// HBM once for both gradients.  W (M x 64) sits in LDS as ready-made B fragments; each gy panel carries its own
// scale, so a wave forms the partial product of one panel at a time and folds it in with that panel's factor.
template <int PA, int PB, int WM, int WN, int TM, int TN, bool ACT, bool DG = false>
__global__ __launch_bounds__(64 * WG_WAVES, 2) void k_wgrad_h2(const WgP P) {
  static_assert(!DG || (PB == 1 && PA * 2 <= 8), "data gradient: N = 64, M <= 256");
  constexpr int WFRAG = DG ? PA * 2 * 4 * 2048 : 0;      // [k-step][n-tile][hi | lo][64 lanes][16 B]
  static_assert(WM * WN == WG_WAVES && PA + PB <= WG_WAVES, "eight waves, one panel per loader wave");
  static_assert(WM * TM * 16 == PA * 64 && WN * TN * 16 == PB * 64, "wave tiles cover the block");
  static_assert((TM * 16) % 64 == 0 || 64 % (TM * 16) == 0, "a wave tile stays inside one panel or covers whole panels");
  constexpr int STAGE = (PA + PB) * WG_PANEL;
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + WFRAG];
  __shared__ int einfo[2][WG_WAVES];
  __shared__ float wmax[WG_WAVES];
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6, g = l >> 4, li = l & 15;
  const int chunk = blockIdx.x, mblock = blockIdx.y;
  const long s0 = P.steps * chunk / P.nchunk, s1 = P.steps * (chunk + 1) / P.nchunk;

  // ---- loader role: wave w < PA + PB owns panel w ----------------------------------------------------------------
  const bool loader = wave < PA + PB, isA = wave < PA;
  const int ld = isA ? P.M : P.N;
  const float* __restrict__ src = isA ? P.a + mblock * (64 * PA) + 64 * wave : P.b + 64 * (wave - PA);
  src += (long)g * ld + 4 * li;                       // this lane: rows g, g+4, .., g+28 of a step, channels 4 li ..
  float4 buf[8];
  int e_run = 15;
  auto issue = [&](long s) {
    if (!loader || s >= s1) return;
    const float* __restrict__ q0 = src + s * 32 * ld;
    if ((s + 1) * 32 <= P.npts) {
#pragma unroll
      for (int i = 0; i < 8; ++i) buf[i] = *reinterpret_cast<const float4*>(q0 + (long)(4 * i) * ld);
    } else {                                          // the last, partial step
#pragma unroll
      for (int i = 0; i < 8; ++i)
        buf[i] = s * 32 + 4 * i + g < P.npts ? *reinterpret_cast<const float4*>(q0 + (long)(4 * i) * ld) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto convert = [&](int st) {
    if (!loader) return;
    if (ACT && !isA) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { buf[i].x = gelu_f(buf[i].x); buf[i].y = gelu_f(buf[i].y); buf[i].z = gelu_f(buf[i].z); buf[i].w = gelu_f(buf[i].w); }
    }
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) m = fmaxf(fmaxf(m, fmaxf(fabsf(buf[i].x), fabsf(buf[i].y))), fmaxf(fabsf(buf[i].z), fabsf(buf[i].w)));
    m = wave_max(m);
    e_run = max(e_run, (int)(__float_as_uint(m) >> 23) & 0xff);
    e_run = min(e_run, 254);
    const float scale = __uint_as_float((unsigned)(268 - e_run) << 23);        // running maximum -> [2^14, 2^15)
    char* const dst = smem + st * STAGE + wave * WG_PANEL;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint2 hi, lo;
      h2_split4(buf[i].x * scale, buf[i].y * scale, buf[i].z * scale, buf[i].w * scale, hi, lo);
      const int off = wg_off(4 * i + g, li);
      *reinterpret_cast<uint2*>(dst + off) = hi;
      *reinterpret_cast<uint2*>(dst + 4096 + off) = lo;
    }
    if (l == 0) einfo[st][wave] = e_run;
  };

  // ---- consumer role ---------------------------------------------------------------------------------------------
  const int wm = wave / WN, wn = wave % WN;
  const int m_off = wm * TM * 16, n_off = wn * TN * 16;            // inside the block
  const int pa = m_off >> 6, ta0 = (m_off & 63) >> 4, pb = PA + (n_off >> 6), tb0 = (n_off & 63) >> 4;
  // transposing-read address of this lane inside a piece: rows 8g+q (and +4), chunk pp of the tile
  const int q = li >> 2, pp = li & 3;
  const int tsw = ((q >> 1) & 1) | ((g & 1) << 1);
  const int trow = (8 * g + q) * 128 + pp * 8;
  typedef s16x4v __attribute__((address_space(3))) * lds_tr;
  auto frag = [&](const char* panel, int tile, f16x8& hi, f16x8& lo) {
    const char* t = panel + trow + ((tile ^ tsw) << 5);
    union { struct { s16x4v a, b; } h; f16x8 v; } uh, ul;
    uh.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(t));
    uh.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(t + 512));
    ul.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(t + 4096));
    ul.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(t + 4096 + 512));
    hi = uh.v; lo = ul.v;
  };
  f32x4v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  constexpr int NPA = (TM * 16 + 63) / 64;     // A panels under this wave's tile (each has its own running exponent)
  static_assert(TN * 16 <= 64, "one B panel per wave tile");
  int ea_used[NPA], eb_used = 15;              // exponents the accumulators are expressed in
#pragma unroll
  for (int k = 0; k < NPA; ++k) ea_used[k] = 15;

  // ---- DG: W as B fragments (k = output channel of the layer, n = input channel), one global power-of-two scale -----
  char* const wfrag = smem + 2 * STAGE;
  float w_inv = 1.f;
  if (DG && s0 < s1) {
    constexpr int NF = PA * 2 * 4 * 64 / (64 * WG_WAVES);     // (k-step, n-tile, lane) triples per thread
    float wv[NF][8];
    float m = 0.f;
#pragma unroll
    for (int r = 0; r < NF; ++r) {
      const int c = tid + r * 64 * WG_WAVES, fl = c & 63, nt = (c >> 6) & 3, ks = c >> 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        wv[r][j] = P.w[(long)(32 * ks + 8 * (fl >> 4) + j) * P.N + 16 * nt + (fl & 15)];
        m = fmaxf(m, fabsf(wv[r][j]));
      }
    }
    m = wave_max(m);
    if (l == 0) wmax[wave] = m;
    __syncthreads();
    m = 0.f;
#pragma unroll
    for (int i = 0; i < WG_WAVES; ++i) m = fmaxf(m, wmax[i]);
    float w_scale;
    h2_scale(m, 0, w_scale, w_inv);
#pragma unroll
    for (int r = 0; r < NF; ++r) {
      const int c = tid + r * 64 * WG_WAVES, fl = c & 63, nt = (c >> 6) & 3, ks = c >> 8;
      uint2 h0, l0, h1, l1;
      h2_split4(wv[r][0] * w_scale, wv[r][1] * w_scale, wv[r][2] * w_scale, wv[r][3] * w_scale, h0, l0);
      h2_split4(wv[r][4] * w_scale, wv[r][5] * w_scale, wv[r][6] * w_scale, wv[r][7] * w_scale, h1, l1);
      char* d = wfrag + (ks * 4 + nt) * 2048 + fl * 16;
      *reinterpret_cast<uint4*>(d) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4*>(d + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
  }
  // this wave's output tile of gx: points 16 (wave & 1) .., input channels 16 (wave >> 1) ..
  const int dg_mt = wave & 1, dg_nt = wave >> 1;

  const bool early_convert = (P.early >> __builtin_amdgcn_readfirstlane(wave)) & 1;
  if (s0 < s1) {
    issue(s0);
    convert(0);
    issue(s0 + 1);
    wg_barrier();                            // the stage (and the weight fragments) are written; global loads stay in flight
    for (long s = s0; s < s1; ++s) {
      const int cur = (int)(s - s0) & 1;
      const char* const stage = smem + cur * STAGE;
      // The loader's conversion of the NEXT step's panels (vector work, into the other stage) stands in front of this
      // step's products (matrix work) for waves 4-7 and behind them for waves 0-3 -- wave w and w + 4 share a SIMD, so
      // one of them converts while the other feeds the matrix pipe; with all eight in the same order the two kinds of
      // work simply added up (the accumulator chains of a wave are independent: one wave fills the pipe).
      if (early_convert && s + 1 < s1) {
        convert(cur ^ 1);                    // waits for the loads of step s + 1
        issue(s + 2);
      }
      const int eb = __builtin_amdgcn_readfirstlane(einfo[cur][pb]);
#pragma unroll
      for (int k = 0; k < NPA; ++k) {
        const int ea = __builtin_amdgcn_readfirstlane(einfo[cur][pa + k]);
        if (ea + eb != ea_used[k] + eb_used) {                               // rare: a new running maximum
          const int d = (ea_used[k] + eb_used) - (ea + eb);                  // < 0
          const float f = d > -126 ? __uint_as_float((unsigned)(127 + d) << 23) : 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i)
            if (((ta0 + i) >> 2) == k) {
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] *= f;
            }
        }
        ea_used[k] = ea;
      }
      eb_used = eb;
      f16x8 bh[TN], bl[TN];                   // B fragments stay, A fragments stream: TN <= 4 in every instance
#pragma unroll
      for (int j = 0; j < TN; ++j) frag(stage + (pb + ((tb0 + j) >> 2)) * WG_PANEL, (tb0 + j) & 3, bh[j], bl[j]);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        f16x8 ah, al;
        frag(stage + (pa + ((ta0 + i) >> 2)) * WG_PANEL, (ta0 + i) & 3, ah, al);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = h2_mfma32(ah, al, bh[j], bl[j], acc[i][j]);
      }
      if (DG) {
        f32x4v tot = (f32x4v){0.f, 0.f, 0.f, 0.f};
        const int row = 16 * dg_mt + li;                    // A fragment: lane = point, 8 consecutive channels
#pragma unroll
        for (int k = 0; k < PA; ++k) {
          f32x4v part = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const char* pa_ = stage + k * WG_PANEL + wg_off(row, 2 * (4 * h + g));
            const f16x8 ah = *reinterpret_cast<const f16x8*>(pa_), al = *reinterpret_cast<const f16x8*>(pa_ + 4096);
            const char* wb = wfrag + ((2 * k + h) * 4 + dg_nt) * 2048 + l * 16;
            const f16x8 bh = *reinterpret_cast<const f16x8*>(wb), bl = *reinterpret_cast<const f16x8*>(wb + 1024);
            part = h2_mfma32(ah, al, bh, bl, part);
          }
          const int ek = __builtin_amdgcn_readfirstlane(einfo[cur][k]);
          const float fk = __uint_as_float((unsigned)(ek - 14) << 23) * w_inv;
          tot += part * fk;
        }
        // lane (g, li) holds points 4 g + r (r = 0..3) of channel li; a 4 x 4 transpose inside every quad of lanes turns
        // that into four consecutive channels of ONE point per lane: one 16-byte store instead of four 4-byte ones
        float t0 = tot[0], t1 = tot[1], t2 = tot[2], t3 = tot[3];
        wg_quad_transpose(t0, t1, t2, t3);
        const long pq = s * 32 + 16 * dg_mt + 4 * g + (li & 3);
        if (pq < P.npts) *reinterpret_cast<float4*>(P.gx + pq * P.N + 16 * dg_nt + 4 * (li >> 2)) = make_float4(t0, t1, t2, t3);
      }
      // (tried: the data-gradient stores behind convert / issue, so that the loader's vmcnt(0) does not wait for their
      //  acknowledgement: 772 vs 684 us in a run where the other two instances were 3-5 % slower than their record -- worse)
      if (!early_convert && s + 1 < s1) {
        convert(cur ^ 1);                    // waits for the loads of step s + 1
        issue(s + 2);
      }
      wg_barrier();
    }
  }

  // ---- slab [M][N] of this chunk ------------------------------------------------------------------------------------
  const float fb = __uint_as_float((unsigned)(eb_used - 14) << 23);
  float* __restrict__ out = P.slabs + (long)chunk * P.M * P.N + (long)(mblock * (64 * PA) + m_off) * P.N + n_off;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const float fa = __uint_as_float((unsigned)(ea_used[(ta0 + i) >> 2] - 14) << 23);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long)(16 * i + 4 * g + r) * P.N + 16 * j + li] = acc[i][j][r] * fa * fb;
  }
}

bool wgrad_h2_ok(long P, int out_f, int in_f) {
  const char* e = getenv("RPDE_WGRAD_H2");
  if ((e && e[0] == '0') || P < 32L * WG_BLOCKS) return false;
  return (out_f == 256 && in_f == 256) || (out_f == 256 && in_f == 64) || (out_f == 64 && in_f == 256);
}

static int wg_chunks(int, int) { return WG_BLOCKS; }

// Which waves convert early (k_wgrad_h2), measured same-box with rocprofv3 over the masks (RPDE_WG_EARLY=<hex 256x256>,
// <hex 64x256>,<hex 256x64> overrides, for such sweeps): 256x256 (eight loaders) waves 4-7 early 820-834 us, all in one
// order 878-898, waves 0-3 early 942 (the lower wave of a SIMD pair seems to win the matrix pipe: it should be the one
// that goes there first); 64x256 (loaders 0-4) waves 1-3 early 490 vs 513-518 us; 256x64 with the data gradient: no
// mask beat the common order (753-762 us), most lost.
static int wg_early_mask(int out_f, int in_f) {
  int m[3] = {0xF0, 0x0E, 0x00};
  if (const char* e = getenv("RPDE_WG_EARLY")) sscanf(e, "%x,%x,%x", &m[0], &m[1], &m[2]);
  return out_f == 256 && in_f == 256 ? m[0] : (out_f == 64 ? m[1] : m[2]);
}

size_t wgrad_h2_slab_floats(long P, int out_f, int in_f) {
  return wgrad_h2_ok(P, out_f, in_f) ? (size_t)wg_chunks(out_f, in_f) * out_f * in_f : 0;
}

template <int PA, int PB, int WM, int WN, int TM, int TN>
static void wg_launch(const WgP& p, int mblocks, bool act, hipStream_t st) {
  const dim3 grid(p.nchunk, mblocks);
  if (act) hipLaunchKernelGGL((k_wgrad_h2<PA, PB, WM, WN, TM, TN, true>), grid, dim3(64 * WG_WAVES), 0, st, p);
  else hipLaunchKernelGGL((k_wgrad_h2<PA, PB, WM, WN, TM, TN, false>), grid, dim3(64 * WG_WAVES), 0, st, p);
}

bool wgrad_h2_dgrad_ok(long P, int out_f, int in_f) { return wgrad_h2_ok(P, out_f, in_f) && out_f == 256 && in_f == 64; }

int wgrad_h2_dgrad(const float* gy, const float* h, const float* w, float* gw, float* gx, long P, int in_f, int out_f,
                   float* slabs, hipStream_t st, FoldJobs* defer) {
  RPDE_CHECK_ARG(wgrad_h2_dgrad_ok(P, out_f, in_f) && slabs && w && gx, "wgrad_h2_dgrad: unsupported shape");
  WgP p;
  p.a = gy; p.b = h; p.slabs = slabs; p.npts = P; p.steps = (P + 31) / 32; p.M = out_f; p.N = in_f; p.nchunk = WG_BLOCKS;
  p.early = wg_early_mask(out_f, in_f);
  p.w = w; p.gx = gx;
  hipLaunchKernelGGL((k_wgrad_h2<4, 1, 4, 2, 4, 2, false, true>), dim3(p.nchunk, 1), dim3(64 * WG_WAVES), 0, st, p);
  RPDE_LAUNCH_CHECK();
  if (defer && defer->add(slabs, gw, out_f * in_f, p.nchunk, (long)out_f * in_f)) return RPDE_OK;
  return reduce_slabs(slabs, gw, (long)out_f * in_f, p.nchunk, (long)out_f * in_f, 1.f, 0, st);
}

int wgrad_h2(const float* gy, const float* h, float* gw, long P, int in_f, int out_f, int act_b, float* slabs, hipStream_t st,
             FoldJobs* defer) {
  RPDE_CHECK_ARG(wgrad_h2_ok(P, out_f, in_f) && slabs, "wgrad_h2: unsupported shape");
  RPDE_CHECK_ARG(act_b == RPDE_ACT_IDENTITY || act_b == RPDE_ACT_GELU, "wgrad_h2: activation %d", act_b);
  WgP p;
  p.a = gy; p.b = h; p.slabs = slabs; p.npts = P; p.steps = (P + 31) / 32; p.M = out_f; p.N = in_f; p.nchunk = wg_chunks(out_f, in_f);
  p.early = wg_early_mask(out_f, in_f);
  p.w = nullptr; p.gx = nullptr;
  const bool act = act_b == RPDE_ACT_GELU;
  if (out_f == 256 && in_f == 256) wg_launch<4, 4, 2, 4, 8, 4>(p, 1, act, st);
  else if (out_f == 256) wg_launch<4, 1, 4, 2, 4, 2>(p, 1, act, st);
  else wg_launch<1, 4, 1, 8, 4, 2>(p, 1, act, st);
  RPDE_LAUNCH_CHECK();
  if (defer && defer->add(slabs, gw, out_f * in_f, p.nchunk, (long)out_f * in_f)) return RPDE_OK;
  return reduce_slabs(slabs, gw, (long)out_f * in_f, p.nchunk, (long)out_f * in_f, 1.f, 0, st);
}

}  // namespace rpde
