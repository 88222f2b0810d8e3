// Fused kernels of FSpectralConv2d.forward_fourier and its adjoint for channels-last fields with C = 64
// channels (reference: models/spectral_convolution.py:256-318), in h2 arithmetic (h2.h).
//
//   analysis   one launch, both axes: every WAVE owns whole lines (one (b,m) row along y, one (b,n) column
//              along x), streams them in 32-point chunks HBM -> registers -> (scale, split into two f16 pieces) ->
//              its private 8 KB of LDS -> transposing reads -> MFMA against the DFT table (resident in LDS as ready
//              fragments), and writes the line's [2K, C] spectrum.  No workgroup barrier after the table is loaded.
//              Blocks are ordered sample-chunk by sample-chunk, y lines then x lines, so that the second read of a
//              sample is served by the Infinity Cache rather than HBM.
//   split      the (mixed) spectra, ~15 % of the field, are re-laid once as MFMA B fragments: [line][16 channels]
//              [hi|lo] 1 KB pieces that a wave loads with one coalesced 16-byte-per-lane read, plus one scale per line.
//   synthesis  ONE pass writes out = Fs_y . At_y[row] + Fs_x . At_x[col] (+ skip gradient): a wave owns a
//              16 x 16 x 16-channel tile, forms the x-axis part column by column (MFMA rows = m), turns it
//              through its private 16 KB of LDS into the layout of the y-axis part (MFMA rows = n), adds and stores.
//              The field is written once and never read back (the round-1 path wrote it twice and re-read it once).
#include "fused_spectral.h"
#include "h2.h"

#include <stdlib.h>

namespace rpde {

// RPDE_STAMPS (debug build, rpde/build.py --stamps): lane 0 of waves from the middle of a launch records s_memtime at
// phase boundaries into a buffer of its own; rpde_debug_fused_stamps() copies it out (profiles/fused_stamps.py)
#ifdef RPDE_STAMPS
__device__ unsigned long long g_fstamps[2][64 * 32];
#define FSTAMP(k, i) do { if (stamp_on) g_fstamps[k][stamp_slot * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FSTAMP(k, i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------------------
// table fragments (built once per plan)
// ------------------------------------------------------------------------------------------------------------
// analysis operand of T[r][y] = src[r*rs + y*cs] (r < R, y < n): fragment (s, mt, piece) holds for lane l the
// eight entries T[16 mt + (l & 15)][32 s + 8 (l >> 4) + j] * 2^12
__global__ __launch_bounds__(64) void k_h2_table_ana(const float* __restrict__ src, long rs, long cs, int R, int n, int MT,
                                                     char* __restrict__ out) {
  const int f = blockIdx.x, l = threadIdx.x;
  const int s = f / MT, mt = f % MT;
  const int r = 16 * mt + (l & 15), y0 = 32 * s + 8 * (l >> 4);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (r < R && y0 + j < n) ? src[r * rs + (y0 + j) * cs] * (float)(1 << H2_TABLE_EXP) : 0.f;
  uint2 h0, l0, h1, l1;
  h2_split4(v[0], v[1], v[2], v[3], h0, l0);
  h2_split4(v[4], v[5], v[6], v[7], h1, l1);
  char* p = out + (long)f * 2048 + l * 16;
  *reinterpret_cast<uint4*>(p) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4*>(p + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

// ---- operand blocks of the synthesis MFMAs ----
// One operand (a 16-row tile of a table, or 16 channels of a spectrum line) over R = 32 K32 + 8 TG reduction rows is
// stored as K32 "hi" fragments, K32 "lo" fragments (1 KB each: lane l holds rows 32 s + 8 (l >> 4) + j) and NP packed
// tail fragments.  The tail has only TG < 4 real lane groups per 32-deep fragment, so the three h2 terms
// (lo*hi, hi*lo, hi*hi) of each real group are laid side by side along the reduction axis: slot t*TG + q holds, on the
// table side, piece (lo, hi, hi)[t] of group q and on the spectrum side piece (hi, lo, hi)[t]; slot -> fragment
// slot / 4, lane group slot % 4.  One MFMA on a packed fragment then yields all three terms: R = 40 costs 4 MFMAs and
// 3 loads per line instead of 6 and 4.  (A 16-deep MFMA for the tail is not an option: mixing
// v_mfma_f32_16x16x16_f16 and v_mfma_f32_16x16x32_f16 in one dependent chain gave wrong accumulator registers on
// gfx950 with ROCm 7.2.)
__host__ __device__ constexpr int h2_np(int TG) { return (3 * TG + 3) / 4; }
__host__ __device__ constexpr int h2_block_bytes(int K32, int TG) { return (2 * K32 + h2_np(TG)) * 1024; }

// 8 values of group q (rows 32 K32 + 8 q ..) -> the 16-byte pieces of the three slots they occupy
__device__ __forceinline__ void h2_store_tail(char* __restrict__ blk, int K32, int TG, int q, int c, bool table, const float (&v)[8]) {
  uint2 h0, l0, h1, l1;
  h2_split4(v[0], v[1], v[2], v[3], h0, l0);
  h2_split4(v[4], v[5], v[6], v[7], h1, l1);
  const uint4 hi = make_uint4(h0.x, h0.y, h1.x, h1.y), lo = make_uint4(l0.x, l0.y, l1.x, l1.y);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int slot = t * TG + q;
    const bool use_lo = table ? (t == 0) : (t == 1);
    *reinterpret_cast<uint4*>(blk + (2 * K32 + slot / 4) * 1024 + ((slot & 3) * 16 + c) * 16) = use_lo ? lo : hi;
  }
}

// synthesis operand of S[y][r] = src[y*rs + r*cs] (y < n, r < R), times 2^12: one block per 16-row tile
__global__ __launch_bounds__(64) void k_h2_table_syn(const float* __restrict__ src, long rs, long cs, int n, int R, int K32,
                                                     int TG, char* __restrict__ out) {
  const int yt = blockIdx.x, l = threadIdx.x, g = l >> 4, li = l & 15;
  const int y = 16 * yt + li;
  char* blk = out + (long)yt * h2_block_bytes(K32, TG);
  for (int i = l; i < h2_np(TG) * 64; i += 64) *reinterpret_cast<uint4*>(blk + 2 * K32 * 1024 + i * 16) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  for (int s = 0; s <= K32; ++s) {
    if (s == K32 && g >= TG) break;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 32 * s + 8 * g + j;
      v[j] = (y < n && r < R) ? src[y * rs + r * cs] * (float)(1 << H2_TABLE_EXP) : 0.f;
    }
    if (s < K32) {
      uint2 h0, l0, h1, l1;
      h2_split4(v[0], v[1], v[2], v[3], h0, l0);
      h2_split4(v[4], v[5], v[6], v[7], h1, l1);
      *reinterpret_cast<uint4*>(blk + s * 1024 + l * 16) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4*>(blk + (K32 + s) * 1024 + l * 16) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else {
      h2_store_tail(blk, K32, TG, g, li, true, v);
    }
  }
}

size_t h2_ana_bytes(int n, int R) { return (size_t)(n / 32) * ((R + 15) / 16) * 2048; }
size_t h2_syn_bytes(int n, int R) { return (size_t)((n + 15) / 16) * h2_block_bytes(R / 32, (R % 32) / 8); }

int h2_build_tables(rpde_plan* p, hipStream_t st) {
  const int R = 2 * p->kp, n = p->n;
  const int MT = (R + 15) / 16, K32 = R / 32, TG = (R % 32) / 8;
  for (int i = 0; i < 2; ++i) {
    RPDE_HIP(hipMalloc(&p->h2_ana[i], h2_ana_bytes(n, R)));
    RPDE_HIP(hipMalloc(&p->h2_syn[i], h2_syn_bytes(n, R)));
  }
  const dim3 ga((n / 32) * MT), gs((n + 15) / 16);
  // [0]: forward operands (Fa analysis, Fs synthesis); [1]: adjoint operands (Fs^T analysis, Fa^T synthesis)
  hipLaunchKernelGGL(k_h2_table_ana, ga, dim3(64), 0, st, p->fa, (long)p->ldn, 1L, R, n, MT, (char*)p->h2_ana[0]);
  hipLaunchKernelGGL(k_h2_table_ana, ga, dim3(64), 0, st, p->fs, 1L, (long)R, R, n, MT, (char*)p->h2_ana[1]);
  hipLaunchKernelGGL(k_h2_table_syn, gs, dim3(64), 0, st, p->fs, (long)R, 1L, n, R, K32, TG, (char*)p->h2_syn[0]);
  hipLaunchKernelGGL(k_h2_table_syn, gs, dim3(64), 0, st, p->fa, 1L, (long)p->ldn, n, R, K32, TG, (char*)p->h2_syn[1]);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// ------------------------------------------------------------------------------------------------------------
// analysis
// ------------------------------------------------------------------------------------------------------------
constexpr int ANA_WAVES = 8;          // lines in flight per workgroup: one round
constexpr int ANA_MAXKS = 8;          // n <= 256

struct AnaAxis {
  const char* timg;   // analysis-type table fragments
  float* spec;        // [lines][R][64]
  int n, ks;          // axis length, n / 32
  int lps, rps;       // lines per sample, rounds (of ANA_WAVES lines) per sample
  int zdiv;           // line z -> field offset (z / zdiv) * s1 + (z % zdiv) * s2
  long s1, s2, ldk;   // ldk: stride between consecutive points of a line
};
struct AnaP {
  const float* x;
  AnaAxis ax[2];
  int naxes, B, chunk, R;
  int items;          // rounds in the launch, ordered chunk of samples by chunk: all y rounds, then all x rounds
};

// byte offset of the 8-byte chunk c8 (channels 4 c8 .. 4 c8 + 3) of point-row k inside one staged piece
// ([32 points][64 channels] f16, 128-byte rows): chunks are XOR-swizzled so that the transposing reads, which
// fetch rows 8g+q / 8g+4+q per 16-lane group, spread over all 64 banks
__device__ __forceinline__ int stage_off(int k, int c8) {
  return k * 128 + ((c8 ^ ((((k >> 1) & 1) << 2) | (((k >> 3) & 1) << 3))) << 3);
}

template <int MT>
__global__ __launch_bounds__(64 * ANA_WAVES, 2) void k_dft_analysis_h2(const AnaP P) {
  __shared__ __attribute__((aligned(16))) char smem[ANA_MAXKS * MT * 2048 + ANA_WAVES * 8192];
  const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6;
  char* const stage = smem + ANA_MAXKS * MT * 2048 + wave * 8192;
  const int g = l >> 4, li = l & 15;
  // transposing-read addresses of this lane (piece 0, channel tile 0): rows 8g+q and 8g+4+q, chunk p
  const int q = li >> 2, pp = li & 3;
  const int tsw = ((q >> 1) & 1) | ((g & 1) << 1);       // chunk-group XOR of those rows (same for both)
  const int trow = (8 * g + q) * 128 + pp * 8;
  typedef s16x4v __attribute__((address_space(3))) * lds_tr;

  // persistent workgroups: round w, w + gridDim.x, ...; every wave walks the same rounds and owns line `wave` of each
  // (the host guarantees whole rounds: lines per sample % 8 == 0).  The 32-point chunks of all its lines form one
  // stream, read two chunks ahead into two register sets, so that 16 KB per wave (128 KB per CU) are in flight.
  const int per_chunk = P.chunk * (P.ax[0].rps + (P.naxes > 1 ? P.ax[1].rps : 0));
  struct Pos { int w, a, s; long z; };              // a < 0: past the end
  auto first_valid = [&](int w) {
    Pos p; p.a = -1; p.s = 0; p.z = -1;
    for (; w < P.items; w += gridDim.x) {          // skips the holes of the last, partial chunk of samples
      const int ck = w / per_chunk;
      int r = w - ck * per_chunk;
      const int nb = min(P.chunk, P.B - ck * P.chunk);
      int aa = 0;
      if (r >= nb * P.ax[0].rps) { r -= nb * P.ax[0].rps; aa = 1; }
      if (aa >= P.naxes || r >= nb * P.ax[aa].rps) continue;
      p.a = aa;
      p.z = (long)(ck * P.chunk + r / P.ax[aa].rps) * P.ax[aa].lps + (r % P.ax[aa].rps) * ANA_WAVES + wave;
      break;
    }
    p.w = w;
    return p;
  };
  auto next = [&](Pos p) {
    if (p.a < 0) return p;
    if (p.s + 1 < P.ax[p.a].ks) { ++p.s; return p; }
    return first_valid(p.w + gridDim.x);
  };
  auto issue = [&](float4 (&buf)[8], const Pos& p) {
    if (p.a < 0) return;
    const AnaAxis& A = P.ax[p.a];
    const float* __restrict__ q0 = P.x + (p.z / A.zdiv) * A.s1 + (p.z % A.zdiv) * A.s2 + (long)(32 * p.s + g) * A.ldk + li * 4;
    const long st4 = 4 * A.ldk;
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[i] = *reinterpret_cast<const float4*>(q0 + i * st4);
  };
  int cur_axis = -1;
  int line_E = 0;              // exponent field the current line is scaled by (0: nothing seen yet)
  f32x4v tot[MT][4];
#ifdef RPDE_STAMPS
  const bool stamp_blk = blockIdx.x >= 96 && blockIdx.x < 160;
  const int stamp_slot = (int)blockIdx.x - 96;
  int stamp_n = 0;
#endif
  // one chunk: buf holds chunk p; once it is in LDS the registers are refilled with chunk pn
  auto process = [&](float4 (&buf)[8], const Pos& p, const Pos& pn) {
    const AnaAxis& A = P.ax[p.a];
    if (p.s == 0) {
      if (p.a != cur_axis) {
        // ---- table fragments -> LDS (the only workgroup-wide step; every wave starts a line of this round here) ----
        __syncthreads();
        const uint4* src = reinterpret_cast<const uint4*>(A.timg);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        const int nvec = A.ks * MT * 128;
        for (int i = tid; i < nvec; i += 64 * ANA_WAVES) dst[i] = src[i];
        __syncthreads();
        cur_axis = p.a;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) tot[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
      line_E = 0;
    }
#ifdef RPDE_STAMPS
    const bool stamp_on = stamp_blk && tid == 0 && stamp_n < 6;
#endif
    FSTAMP(0, stamp_n * 5 + 0);
    // chunk maximum -> the line's running power-of-two scale.  The accumulators stay in scaled units for the whole
    // line (the MFMAs accumulate in place; no per-chunk rescaling on the VALU): the scale exponent only ever grows,
    // and when a chunk exceeds it the accumulators are multiplied by the (exact) ratio once.
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(buf[i].x), "v"(buf[i].y));
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(buf[i].z), "v"(buf[i].w));
    }
    m = wave_max(m);
    {
      const int E = max((int)(__float_as_uint(m) >> 23) & 0xff, 15 + H2_TABLE_EXP);
      if (E > line_E) {                                  // wave-uniform
        if (line_E > 0) {
          const float f = __uint_as_float((unsigned)(127 + line_E - E) << 23);      // 2^(old - new) <= 1/2, >= 2^-126
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
              for (int j = 0; j < 4; ++j) tot[mt][nt][j] *= (E - line_E < 126 ? f : 0.f);
        }
        line_E = E;
      }
    }
    const float scale = __uint_as_float((unsigned)(268 - line_E) << 23);
    FSTAMP(0, stamp_n * 5 + 1);
    // split into two f16 pieces, keep the memory order [point][channel] in LDS
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      uint2 hi, lo;
      h2_split4(buf[i].x * scale, buf[i].y * scale, buf[i].z * scale, buf[i].w * scale, hi, lo);
      const int off = stage_off(4 * i + g, li);
      *reinterpret_cast<uint2*>(stage + off) = hi;
      *reinterpret_cast<uint2*>(stage + 4096 + off) = lo;
    }
    issue(buf, pn);
    wave_lds_fence();
    FSTAMP(0, stamp_n * 5 + 2);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const char* t = stage + trow + ((nt ^ tsw) << 5);
      union { struct { s16x4v a, b; } h; f16x8 v; } bh, bl;
      bh.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(t));
      bh.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(t + 512));
      bl.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(t + 4096));
      bl.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr)(t + 4096 + 512));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        // table fragments straight from LDS each time (the LDS pipe is nearly idle; keeping all of them in
        // registers would cost 16 VGPRs that the second load buffer needs)
        const char* ta = smem + (p.s * MT + mt) * 2048 + l * 16;
        const f16x8 ah = *reinterpret_cast<const f16x8*>(ta), al = *reinterpret_cast<const f16x8*>(ta + 1024);
        tot[mt][nt] = h2_mfma32(ah, al, bh.v, bl.v, tot[mt][nt]);
      }
    }
    wave_lds_fence();
    FSTAMP(0, stamp_n * 5 + 3);
#ifdef RPDE_STAMPS
    ++stamp_n;
#endif
    if (p.s == A.ks - 1) {
      // spectrum of the line: row 16 mt + 4 g + j, channel 16 nt + li
      float* __restrict__ sp = A.spec + p.z * (long)P.R * 64;
      const float inv = __uint_as_float((unsigned)(line_E - 14 - H2_TABLE_EXP) << 23);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = 16 * mt + 4 * g + j;
          if (row < P.R) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) sp[row * 64 + 16 * nt + li] = tot[mt][nt][j] * inv;
          }
        }
    }
  };
  float4 ba[8], bb[8];
  Pos pa = first_valid(blockIdx.x), pb = next(pa);
  issue(ba, pa);
  issue(bb, pb);
  while (pa.a >= 0) {
    const Pos pc = next(pb);
    process(ba, pa, pc);
    if (pb.a < 0) break;
    const Pos pd = next(pc);
    process(bb, pb, pd);
    pa = pc; pb = pd;
  }
}

// ------------------------------------------------------------------------------------------------------------
// split: fp32 spectra [line][R][64] -> operand blocks [line][16-channel block] (layout above) + 1/scale per line
// ------------------------------------------------------------------------------------------------------------
template <int K32, int TG>
__global__ __launch_bounds__(256) void k_spec_split_h2(const float* __restrict__ spec, char* __restrict__ img,
                                                       float* __restrict__ inv_out, int R) {
  constexpr int NF = K32 + (TG ? 1 : 0);
  __shared__ float wmax[4];
  const long line = blockIdx.x;
  const int tid = threadIdx.x, l = tid & 63, cb = tid >> 6, g = l >> 4, li = l & 15;
  const float* __restrict__ src = spec + line * (long)R * 64 + 16 * cb + li;
  float v[NF][8];
  float m = 0.f;
#pragma unroll
  for (int s = 0; s < NF; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * s + 8 * g + j;
      v[s][j] = k < R ? src[k * 64] : 0.f;
      m = fmaxf(m, fabsf(v[s][j]));
    }
  m = wave_max(m);
  if (l == 0) wmax[cb] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  float scale, inv;
  h2_scale(m, H2_TABLE_EXP, scale, inv);
  if (tid == 0) inv_out[line] = inv;
  char* blk = img + (line * 4 + cb) * (long)h2_block_bytes(K32, TG);
  // unused slots of the packed fragments are never loaded (the synthesis masks those lanes)
#pragma unroll
  for (int s = 0; s < NF; ++s) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[s][j] *= scale;
    if (s < K32) {
      uint2 h0, l0, h1, l1;
      h2_split4(v[s][0], v[s][1], v[s][2], v[s][3], h0, l0);
      h2_split4(v[s][4], v[s][5], v[s][6], v[s][7], h1, l1);
      *reinterpret_cast<uint4*>(blk + s * 1024 + l * 16) = make_uint4(h0.x, h0.y, h1.x, h1.y);
      *reinterpret_cast<uint4*>(blk + (K32 + s) * 1024 + l * 16) = make_uint4(l0.x, l0.y, l1.x, l1.y);
    } else if (g < TG) {
      h2_store_tail(blk, K32, TG, g, li, false, v[s]);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// synthesis of both axes
// ------------------------------------------------------------------------------------------------------------
struct SynP {
  const char* imgy; const char* imgx;     // B fragments of the y lines (b,m) and the x lines (b,n)
  const float* invy; const float* invx;   // 1 / (line scale * table scale)
  const char* taby; const char* tabx;     // synthesis-type table fragments: rows n (y axis, length N), rows m (x axis, length M)
  float* out; const float* skip;          // [B,M,N,64]; optional tensor added to the result
  int B, M, N;
  int sy, sx;                              // super-tile of sy x sx tiles (divide M/16, N/16): what one XCD has in flight
};

// one operand block in registers
template <int K32, int TG>
struct Frag {
  static constexpr int NP = h2_np(TG);
  f16x8 h[K32 > 0 ? K32 : 1], lo[K32 > 0 ? K32 : 1], pk[NP > 0 ? NP : 1];
  __device__ __forceinline__ void load(const char* __restrict__ p, int l) {
#pragma unroll
    for (int s = 0; s < K32; ++s) {
      h[s] = *reinterpret_cast<const f16x8*>(p + s * 1024 + l * 16);
      lo[s] = *reinterpret_cast<const f16x8*>(p + (K32 + s) * 1024 + l * 16);
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      constexpr int full = 3 * TG / 4;               // packed fragments with all four lane groups in use
      const int slots = q < full ? 4 : 3 * TG - 4 * full;
      const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      pk[q] = z;
      if ((l >> 4) < slots) pk[q] = *reinterpret_cast<const f16x8*>(p + (2 * K32 + q) * 1024 + l * 16);
    }
  }
};

// float index of element (m, n, c) of a wave's 16 x 16 x 16 turning buffer: the n slot is XOR-ed so that both the
// writes (lanes differ in c and m>>2) and the reads (lanes differ in c and n>>2) touch 32 different banks
__device__ __forceinline__ int turn_idx(int m, int n, int c) {
  return m * 256 + ((n ^ ((m >> 2) & 1) ^ ((n >> 2) & 1)) << 4) + c;
}

template <int K32, int TG>
__global__ __launch_bounds__(256, 2) void k_dft_synthesis2_h2(const SynP P) {
  constexpr int BB = h2_block_bytes(K32, TG);
  constexpr long LB = 4L * BB;                 // bytes per line: 4 channel blocks
  __shared__ __attribute__((aligned(16))) float turn[4 * 4096];
  const int tid = threadIdx.x, l = tid & 63, cb = tid >> 6, g = l >> 4, li = l & 15;
  // tile: groups of 8 samples, one per XCD (blocks b, b+8, .. share an L2)
  const int tn = P.N >> 4, tps = (P.M >> 4) * tn;
  const long group = 8L * tps;
  const int blk = (int)(blockIdx.x / group);
  const int r0 = (int)(blockIdx.x - blk * group);
  const int nb = min(8, P.B - 8 * blk);
  const int b = 8 * blk + r0 % nb, t = r0 / nb;
  if (t >= tps) return;
  // tiles of a sample go super-tile by super-tile (8 x 8 tiles = the 64 workgroups an XCD runs at once): the lines a
  // super-tile needs, (sy + sx) * 16, fit in the XCD's L2 and each is fetched from beyond it once per super-tile
  const int per_st = P.sy * P.sx, st = t / per_st, wi = t - st * per_st;
  const int stn = tn / P.sx;
  const int m0 = ((st / stn) * P.sy + wi / P.sx) << 4, n0 = ((st % stn) * P.sx + wi % P.sx) << 4;
  float* const buf = turn + cb * 4096;
#ifdef RPDE_STAMPS
  const long stamp_first = ((long)gridDim.x / 2) & ~63L;
  const bool stamp_on = l == 0 && cb == 0 && blockIdx.x >= stamp_first && blockIdx.x < stamp_first + 64;
  const int stamp_slot = (int)(blockIdx.x - stamp_first);
#endif
  FSTAMP(1, 0);

  typedef Frag<K32, TG> F;
  F tx, ty;
  tx.load(P.tabx + (long)(m0 >> 4) * BB, l);        // rows m0.. of the x-axis table
  ty.load(P.taby + (long)(n0 >> 4) * BB, l);        // rows n0.. of the y-axis table

  // 32 lines: 0..15 are the columns n0 + i of the x-axis part ([16 m][16 c] tiles, written to the turning buffer),
  // 16..31 the rows m0 + i of the y-axis part ([16 n][16 c] tiles, added to the turned x part in the buffer).
  const long zx = (long)b * P.N + n0, zy = (long)b * P.M + m0;
  const char* __restrict__ srcx = P.imgx + zx * LB + cb * BB;
  const char* __restrict__ srcy = P.imgy + zy * LB + cb * BB;
  // the 32 line scales in one vector load each (a scalar load per line would put its latency on every step)
  const float invx_l = P.invx[zx + li], invy_l = P.invy[zy + li];
  // lines go through in groups of four whose MFMA chains are interleaved; the next group's fragments are in flight
  // (three groups in flight -- 12 lines, 238 VGPRs -- measured the same 338 us inside the training step: the kernel
  //  is bound by the bandwidth between the L2 / Infinity Cache and the CUs, not by the latency of these loads)
  F f[8];
  auto line_src = [&](int i) { return i < 16 ? srcx + i * LB : srcy + (i - 16) * LB; };
#pragma unroll
  for (int u = 0; u < 4; ++u) f[u].load(line_src(u), l);
#pragma unroll
  for (int gi = 0; gi < 8; ++gi) {
    if (gi + 1 < 8) {
#pragma unroll
      for (int u = 0; u < 4; ++u) f[((gi + 1) * 4 + u) & 7].load(line_src((gi + 1) * 4 + u), l);
    }
    const F& tab = gi < 4 ? tx : ty;
    f32x4v c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < F::NP; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.pk[q], f[(gi * 4 + u) & 7].pk[q], c[u], 0, 0, 0);
#pragma unroll
    for (int s = K32 - 1; s >= 0; --s) {
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.lo[s], f[(gi * 4 + u) & 7].h[s], c[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.h[s], f[(gi * 4 + u) & 7].lo[s], c[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.h[s], f[(gi * 4 + u) & 7].h[s], c[u], 0, 0, 0);
    }
    FSTAMP(1, 1 + 2 * gi);
    if (gi == 4) wave_lds_fence();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = gi * 4 + u;
      if (gi < 4) {
        const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invx_l), i));
#pragma unroll
        for (int j = 0; j < 4; ++j) buf[turn_idx(4 * g + j, i, li)] = c[u][j] * sc;
      } else {
        const int r = i - 16;
        const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invy_l), r));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ix = turn_idx(r, 4 * g + j, li);
          buf[ix] = fmaf(c[u][j], sc, buf[ix]);
        }
      }
    }
    FSTAMP(1, 2 + 2 * gi);
  }
  wave_lds_fence();
  FSTAMP(1, 17);
  // the finished [16 m][16 n][16 c] tile leaves as 16-byte pieces: lane -> point n = l >> 2, channels 4 (l & 3) ..
  {
    const int n = l >> 2, c4 = (l & 3) << 2;
    const long o = (zy * P.N + n0 + n) * 64 + 16 * cb + c4;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      float4 v = *reinterpret_cast<const float4*>(buf + turn_idx(m, n, c4));
      const long oo = o + (long)m * P.N * 64;
      if (P.skip) {
        const float4 k = *reinterpret_cast<const float4*>(P.skip + oo);
        v.x += k.x; v.y += k.y; v.z += k.z; v.w += k.w;
      }
      *reinterpret_cast<float4*>(P.out + oo) = v;
    }
  }
  FSTAMP(1, 18);
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
bool fused2d_ok(int M, int N, int C, int keff_y, int keff_x) {
  const int Ry = 2 * ((keff_y + 3) / 4 * 4), Rx = 2 * ((keff_x + 3) / 4 * 4);
  if (const char* e = getenv("RPDE_FUSED_SPECTRAL")) if (e[0] == '0') return false;
  return C == 64 && M % 32 == 0 && N % 32 == 0 && M <= 32 * ANA_MAXKS && N <= 32 * ANA_MAXKS && Ry == Rx && Ry <= 48;
}

size_t fused2d_img_bytes(long lines, int R) { return (size_t)lines * 4 * h2_block_bytes(R / 32, (R % 32) / 8); }

// R = 8 .. 48 in steps of 8 -> (K32, TG) = (0,1) (0,2) (0,3) (1,0) (1,1) (1,2)
#define RPDE_H2_DISPATCH(KERNEL, R, ...)                                                        \
  do {                                                                                          \
    switch ((R) / 8) {                                                                          \
      case 1: hipLaunchKernelGGL((KERNEL<0, 1>), __VA_ARGS__); break;                           \
      case 2: hipLaunchKernelGGL((KERNEL<0, 2>), __VA_ARGS__); break;                           \
      case 3: hipLaunchKernelGGL((KERNEL<0, 3>), __VA_ARGS__); break;                           \
      case 4: hipLaunchKernelGGL((KERNEL<1, 0>), __VA_ARGS__); break;                           \
      case 5: hipLaunchKernelGGL((KERNEL<1, 1>), __VA_ARGS__); break;                           \
      default: hipLaunchKernelGGL((KERNEL<1, 2>), __VA_ARGS__); break;                          \
    }                                                                                           \
  } while (0)

int fused2d_analysis(const float* x, float* spec_y, float* spec_x, const rpde_plan* py, const rpde_plan* px, int adjoint,
                     int B, int M, int N, hipStream_t st) {
  AnaP P;
  memset(&P, 0, sizeof(P));
  P.x = x; P.naxes = 2; P.B = B; P.R = 2 * py->kp;
  // chunk of samples whose field (read twice) stays inside the 256 MB Infinity Cache beside everything else that
  // streams through it (measured at B = 32, 256^2: no chunking 0.73 ms per forward, 160 MB 0.69, 40-80 MB 0.66)
  const long sample_bytes = (long)M * N * 64 * 4;
  long ch = (64L << 20) / sample_bytes;
  P.chunk = (int)(ch < 1 ? 1 : (ch > B ? B : ch));
  AnaAxis& ay = P.ax[0];
  ay.timg = (const char*)py->h2_ana[adjoint]; ay.spec = spec_y; ay.n = N; ay.ks = N / 32; ay.lps = M;
  ay.rps = (M + ANA_WAVES - 1) / ANA_WAVES; ay.zdiv = 1; ay.s1 = (long)N * 64; ay.s2 = 0; ay.ldk = 64;
  AnaAxis& ax = P.ax[1];
  ax.timg = (const char*)px->h2_ana[adjoint]; ax.spec = spec_x; ax.n = M; ax.ks = M / 32; ax.lps = N;
  ax.rps = (N + ANA_WAVES - 1) / ANA_WAVES; ax.zdiv = N; ax.s1 = (long)M * N * 64; ax.s2 = 64; ax.ldk = (long)N * 64;
  P.items = ((B + P.chunk - 1) / P.chunk) * P.chunk * (ay.rps + ax.rps);
  // one persistent workgroup per CU (the table + 8 staging areas take 80-112 KB of LDS)
  int dev = 0, cus = 256;
  RPDE_HIP(hipGetDevice(&dev));
  RPDE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int grid = P.items < cus ? P.items : cus;
  const int MT = (P.R + 15) / 16;
  if (MT == 1) hipLaunchKernelGGL(k_dft_analysis_h2<1>, dim3(grid), dim3(64 * ANA_WAVES), 0, st, P);
  else if (MT == 2) hipLaunchKernelGGL(k_dft_analysis_h2<2>, dim3(grid), dim3(64 * ANA_WAVES), 0, st, P);
  else hipLaunchKernelGGL(k_dft_analysis_h2<3>, dim3(grid), dim3(64 * ANA_WAVES), 0, st, P);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

int fused2d_split(const float* spec, void* img, float* inv, long lines, int R, hipStream_t st) {
  const dim3 grid((unsigned)lines), blk(256);
  RPDE_H2_DISPATCH(k_spec_split_h2, R, grid, blk, 0, st, spec, (char*)img, inv, R);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

int fused2d_synthesis(const void* imgy, const void* imgx, const float* invy, const float* invx, const rpde_plan* py,
                      const rpde_plan* px, int adjoint, float* out, const float* skip, int B, int M, int N, hipStream_t st) {
  SynP P;
  P.imgy = (const char*)imgy; P.imgx = (const char*)imgx; P.invy = invy; P.invx = invx;
  P.taby = (const char*)py->h2_syn[adjoint]; P.tabx = (const char*)px->h2_syn[adjoint];
  P.out = out; P.skip = skip; P.B = B; P.M = M; P.N = N;
  // (super-tile side, measured at B = 32, 256^2, forward + backward: 2 -> 1.52 ms, 4 -> 1.44, 8 -> 1.45, 16 -> 1.53)
  auto side = [](int tiles) { int s = tiles < 8 ? tiles : 8; while (tiles % s) --s; return s; };
  P.sy = side(M / 16); P.sx = side(N / 16);
  const int R = 2 * py->kp;
  const long tps = (long)(M / 16) * (N / 16);
  const dim3 grid((unsigned)(((B + 7) / 8) * 8 * tps)), blk(256);
  RPDE_H2_DISPATCH(k_dft_synthesis2_h2, R, grid, blk, 0, st, P);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // namespace rpde

#ifdef RPDE_STAMPS
extern "C" int rpde_debug_fused_stamps(unsigned long long* host_out) {
  RPDE_HIP(hipDeviceSynchronize());
  RPDE_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(rpde::g_fstamps), sizeof(unsigned long long) * 2 * 64 * 32));
  return RPDE_OK;
}
#endif
