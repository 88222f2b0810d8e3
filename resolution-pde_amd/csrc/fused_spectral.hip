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
#include "cw.h"
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
// perm: slot j of lane group g holds reduction index 32 s + 4 j + g instead of 32 s + 8 g + j (the points a lane of
// k_dft_analysis_sq_h2 loads)
__global__ __launch_bounds__(64) void k_h2_table_ana(const float* __restrict__ src, long rs, long cs, int R, int n, int MT,
                                                     char* __restrict__ out, int perm) {
  const int f = blockIdx.x, l = threadIdx.x;
  const int s = f / MT, mt = f % MT;
  const int r = 16 * mt + (l & 15), g = l >> 4;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int y = 32 * s + (perm ? 4 * j + g : 8 * g + j);
    v[j] = (r < R && y < n) ? src[r * rs + y * cs] * (float)(1 << H2_TABLE_EXP) : 0.f;
  }
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
// 3 loads per line instead of 6 and 4.  (A 16-deep MFMA for the tail is not an option: a v_mfma_f32_16x16x16_f16 directly
// behind the v_mfma_f32_16x16x32_f16 it accumulates onto reads a partly written accumulator on gfx950 with ROCm 7.2 --
// h2.h, profiles/r04_mfma_mix.txt.)
// (h2_np / h2_block_bytes: fused_spectral.h)

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
    RPDE_HIP(hipMalloc(&p->h2_ana_p[i], h2_ana_bytes(n, R)));
    RPDE_HIP(hipMalloc(&p->h2_syn[i], h2_syn_bytes(n, R)));
  }
  const dim3 ga((n / 32) * MT), gs((n + 15) / 16);
  // [0]: forward operands (Fa analysis, Fs synthesis); [1]: adjoint operands (Fs^T analysis, Fa^T synthesis)
  hipLaunchKernelGGL(k_h2_table_ana, ga, dim3(64), 0, st, p->fa, (long)p->ldn, 1L, R, n, MT, (char*)p->h2_ana[0], 0);
  hipLaunchKernelGGL(k_h2_table_ana, ga, dim3(64), 0, st, p->fs, 1L, (long)R, R, n, MT, (char*)p->h2_ana[1], 0);
  hipLaunchKernelGGL(k_h2_table_ana, ga, dim3(64), 0, st, p->fa, (long)p->ldn, 1L, R, n, MT, (char*)p->h2_ana_p[0], 1);
  hipLaunchKernelGGL(k_h2_table_ana, ga, dim3(64), 0, st, p->fs, 1L, (long)R, R, n, MT, (char*)p->h2_ana_p[1], 1);
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
  float* amax;        // [lines] max |spectrum| (null: not wanted)
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
      if (A.amax) {
        // (rows beyond R belong to zero table rows: they do not disturb the maximum)
        float am = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) {
            asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(am) : "v"(tot[mt][nt][0]), "v"(tot[mt][nt][1]));
            asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(am) : "v"(tot[mt][nt][2]), "v"(tot[mt][nt][3]));
          }
        am = wave_max(am);
        if (l == 0) A.amax[p.z] = am * inv;
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
// analysis of square grids, both axes out of ONE pass over HBM (round 3)
// ------------------------------------------------------------------------------------------------------------
// k_dft_analysis_h2 reads the field twice -- once per axis -- and the second read comes from the Infinity Cache at
// best, which streams no faster than HBM (profiles/r03_mall_bandwidth.txt: 6.1 vs 5.4 TB/s).  An L2 does (12 TB/s),
// but it holds 4 MB per XCD.  So the two reads of every 32-row block (2 MB at 256^2) are brought together in time and
// in place: the 32 workgroups of one XCD group (blockIdx % 8; placement is a speed matter only) walk through a sample
// row block by row block, and in step t
//   * every wave owns one COLUMN of the sample for the whole sample (256 waves per group = 256 columns) and adds the
//     32 rows of block t to its x-axis accumulators (one chunk of 32 points, 256 B apart by the row length);
//   * every workgroup owns one ROW of the block: its eight waves take one 32-point chunk of the row each, and the eight
//     partial y-axis spectra are summed through LDS (two halves of the channels, 48 KB).
// Each step reads its 2 MB row block once by rows and once by columns, within microseconds of each other, on one
// XCD.  No inter-workgroup synchronisation: the workgroups of a group run the same schedule and drift only by
// scheduling noise.  Needs M = N (one table for both axes).
// Grids below 256: a row has n / 32 < 8 chunks, so a workgroup takes rpw = 8 / (n / 32) rows of the block (wave ->
// (row, chunk)) and a group is 32 / rpw workgroups -- 8 * 32 / rpw waves >= n columns -- so that at 128^2 and 64^2 every
// wave has a column and a chunk as well; the chip then holds 8 rpw groups, each walking through its own samples
// (with one row per workgroup whatever the grid, 128^2 ran at 81 us for a quarter of the bytes of 256^2's 212 us).
struct AnaSqP {
  const float* x; const char* timg;
  float* spec_y; float* spec_x; float* amax_y; float* amax_x;
  int B, n, ks, R, ng, rpw;
  int b1;          // 1: keep the (redundant) barrier in front of the partial-spectrum writes
};

template <int MT>
__global__ __launch_bounds__(64 * ANA_WAVES, 2) void k_dft_analysis_sq_h2(const AnaSqP P) {
  constexpr int TAB = ANA_MAXKS * MT * 2048, STG = ANA_WAVES * 8192, RED = ANA_WAVES * MT * 2 * 1024;
  __shared__ __attribute__((aligned(16))) char smem[TAB + STG + RED];         // 160 KB at MT = 3: all of it
  // (wave through readfirstlane: everything derived from it -- duties, wait counts -- is then uniform for the compiler
  //  too: scalar branches instead of exec masks, and the if / else chain of waits below is one the ISA test can follow)
  const int tid = threadIdx.x, l = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = l >> 4, li = l & 15;
  char* const stage = smem + TAB + wave * 8192;
  // per-wave partial maxima of a row's spectrum: the last 32 bytes of the reduction area (the landing zone of the
  // partial spectra ends 16 KB before it)
  float* const wmax = reinterpret_cast<float*>(smem + TAB + STG + RED - 256);     // [row of the workgroup][wave]
  const int q = li >> 2, pp = li & 3;
  const int tsw = ((q >> 1) & 1) | ((g & 1) << 1);
  const int trow = (8 * g + q) * 128 + pp * 8;
  typedef s16x4v __attribute__((address_space(3))) * lds_tr;

  const int n = P.n, steps = P.ks;                 // rows per sample = columns = n; row blocks = chunks per row = n / 32
  const int xg = blockIdx.x % P.ng, jw = blockIdx.x / P.ng, gw = jw * ANA_WAVES + wave;
  const int nsamp = (P.B - xg + P.ng - 1) / P.ng;
  const long units = (long)nsamp * steps;           // (sample, row block) pairs of this group, in order
  const int rpw = P.rpw, yr = wave / steps, yc = wave - yr * steps;      // this wave's row of the workgroup's rows, its chunk
  const bool has_x = gw < n, has_y = wave < rpw * steps;
  if (units <= 0) return;                           // (more groups than samples)
  {
    const uint4* src = reinterpret_cast<const uint4*>(P.timg);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = tid; i < steps * MT * 128; i += 64 * ANA_WAVES) dst[i] = src[i];
    if (tid < 64) wmax[tid] = 0.f;
    __syncthreads();
  }
  const long rowf = (long)n * 64;                   // floats per row of the field
  // The field comes through registers, one unit ahead -- by loads the compiler does not see (asm) and waits counted by
  // hand.  Left to the compiler every use of a prefetched chunk is preceded by s_waitcnt vmcnt(0..7): it takes the
  // chunk for the youngest thing in flight, so each phase also waits for the OTHER axis's chunk, requested only a
  // phase ago, and for the spectrum stores in between -- the prefetch distance shrinks from a unit to nothing (this
  // kernel ran at 219 us, latency-bound).  The queue is in order: behind the x chunk of unit u there are always the 8
  // loads of the y chunk of u (and some stores), behind the y chunk of u the 8 loads of the x chunk of u + 1; so
  // vmcnt(8) is enough in both places.  Every wave issues exactly 8 loads per phase (past the end: a
  // clamped re-read; without a duty: a neighbour's chunk), so the count holds for all.
  auto gload = [](const float* p) {
    f32x4v v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
  };
  // (a wave without a column or without a chunk -- grids that are no power of two -- loads a neighbour's: every wave
  //  issues the same 16 loads per unit and the counts below hold for all)
  const int gwc = has_x ? gw : n - 1, yrc = has_y ? yr : 0, ycc = has_y ? yc : 0;
  auto issue_x = [&](f32x4v (&buf)[8], long u) {    // column gw, rows of block t: lane (g, li) takes rows 4i + g
    u = u < units ? u : units - 1;
    const int sb = (int)(u / steps), t = (int)(u - (long)sb * steps);
    const float* q0 = P.x + (((long)(xg + P.ng * sb) * n + 32 * t + g) * n + gwc) * 64 + li * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[i] = gload(q0 + i * 4 * rowf);
  };
  auto issue_y = [&](f32x4v (&buf)[8], long u) {    // row 32 t + rpw jw + yr, points 32 yc ..: lane (g, li) takes points 4i + g
    u = u < units ? u : units - 1;
    const int sb = (int)(u / steps), t = (int)(u - (long)sb * steps);
    const float* q0 = P.x + (((long)(xg + P.ng * sb) * n + 32 * t + rpw * jw + yrc) * n + 32 * ycc + g) * 64 + li * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[i] = gload(q0 + i * 256);
  };
  // The chunk in buf has landed once at most N later vector-memory instructions are outstanding.  N counts what is
  // CERTAIN to have been issued behind it: the other axis's 8 loads and this wave's spectrum stores of the reduction in
  // between (`nst`, wave-uniform: the items of the reduction loop whose first lane group has a valid row, plus the line
  // maximum written by wave 0) -- with the stores counted in, a wave no longer waits for their acknowledgement before
  // it may touch data that arrived long ago (1.2-1.7 K of 13 K cycles per step).
  int nst_wave = 0;
  for (int e = 64 * wave; e < rpw * MT * 256; e += 64 * ANA_WAVES) {
    const int rem = e % (MT * 256);
    nst_wave += (16 * (rem >> 8) + ((rem >> 6) & 3)) < P.R ? 1 : 0;
  }
  if (wave == 0 && P.amax_y) ++nst_wave;
  nst_wave = __builtin_amdgcn_readfirstlane(nst_wave);
  // ONE statement per wait, the choice of the count inside it (scalar compare + branch on nst): the chunk's registers
  // pass through it as tied operands -- that is what orders their uses behind the wait -- and with one statement per
  // alternative count the register allocator gave the four statements' operands other registers than the loads' and
  // copied in front of the wait, i.e. read registers whose loads were in flight (seen in the MT = 1 instance;
  // tests/test_isa_pending_loads_cpu.py follows the `landed` comment, which names the registers).  The very first
  // chunks have no stores behind them: the loop waits for everything before its first unit instead.
  auto landed = [&](f32x4v (&buf)[8]) {
    asm volatile(
        "s_cmp_lt_u32 %8, 2\n\t"
        "s_cbranch_scc1 1f\n\t"
        "s_cmp_eq_u32 %8, 2\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_waitcnt vmcnt(11)\n\t"
        "s_branch 4f\n"
        "2:\n\t"
        "s_waitcnt vmcnt(10)\n\t"
        "s_branch 4f\n"
        "1:\n\t"
        "s_cmp_eq_u32 %8, 1\n\t"
        "s_cbranch_scc1 3f\n\t"
        "s_waitcnt vmcnt(8)\n\t"
        "s_branch 4f\n"
        "3:\n\t"
        "s_waitcnt vmcnt(9)\n"
        "4:\n\t"
        "s_nop 0 ; landed %0 %1 %2 %3 %4 %5 %6 %7"
        : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5]), "+v"(buf[6]), "+v"(buf[7])
        : "s"(nst_wave)
        : "memory", "scc");
  };
  // one 32-point chunk into accumulators that stay in scaled units for the whole line (k_dft_analysis_h2's scheme)
  auto process = [&](f32x4v (&buf)[8], int s, f32x4v (&tot)[MT][4], int& line_E) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(buf[i].x), fabsf(buf[i].y)), fmaxf(fabsf(buf[i].z), fabsf(buf[i].w))));
    m = wave_max(m);
    {
      const int E = max((int)(__float_as_uint(m) >> 23) & 0xff, 15 + H2_TABLE_EXP);
      if (E > line_E) {
        if (line_E > 0) {
          const float f = __uint_as_float((unsigned)(127 + line_E - E) << 23);
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
    // No staging: the eight points a lane loaded for channel 4 li + e (buf[0..7], component e) ARE the eight reduction
    // slots of its lane group in a B fragment whose column li stands for channel 4 li + e -- the table image carries the
    // matching permutation of the reduction index (h2_ana_p).  Four fragments (e = 0..3) instead of four channel tiles;
    // accumulator column li of tot[mt][e] is channel 4 li + e.
    const float scale = __uint_as_float((unsigned)(268 - line_E) << 23);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      union { f16x8 v; struct { uint2 a, b; } u; } H, L;
      h2_split4(buf[0][e] * scale, buf[1][e] * scale, buf[2][e] * scale, buf[3][e] * scale, H.u.a, L.u.a);
      h2_split4(buf[4][e] * scale, buf[5][e] * scale, buf[6][e] * scale, buf[7][e] * scale, H.u.b, L.u.b);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const char* ta = smem + (s * MT + mt) * 2048 + l * 16;
        const f16x8 ah = *reinterpret_cast<const f16x8*>(ta), al = *reinterpret_cast<const f16x8*>(ta + 1024);
        tot[mt][e] = h2_mfma32(ah, al, H.v, L.v, tot[mt][e]);
      }
    }
  };

  f32x4v totx[MT][4], toty[MT][4];
  int Ex = 0;
  f32x4v bx[8], by[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bx[i] = by[i] = (f32x4v){0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the table copy above: from here on the queue is counted by hand)
  issue_x(bx, 0);
  issue_y(by, 0);
#ifdef RPDE_STAMPS
  const bool stamp_wave = l == 0 && wave == 3 && blockIdx.x >= 96 && blockIdx.x < 160;
  const int stamp_slot = (int)blockIdx.x - 96;
#endif
  for (long u = 0; u < units; ++u) {
    const int sb = (int)(u / steps), t = (int)(u - (long)sb * steps);
    const int b = xg + P.ng * sb;
#ifdef RPDE_STAMPS
    // units 9, 10, 11 (second sample, mid-line): 10 stamps each
    const bool stamp_on = stamp_wave && u >= 9 && u < 12;
    const int sq0 = (int)(u - 9) * 10;
#endif
    FSTAMP(0, sq0 + 0);
    // (x first.  The other order -- whole 64 KB rows going to HBM, the x axis's 256-byte pieces finding them in L2 -- was
    //  measured: same time, but 604 instead of 553 MB fetched per launch)
    // ---- x axis: this wave's column, rows of block t ----
    // EVERY wave waits for its chunk, with or without a duty: once the asm statement has returned the compiler takes
    // the registers for filled, and where the values are not used it hands the registers to something else -- here the
    // 64-bit division of the next address computation -- which a load still in flight then overwrites (found as a
    // memory-aperture fault at 96^2 in training, never in the tests: a wave without a column skipped the wait)
    if (u == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (no stores behind the first chunks yet)
    landed(bx);
    if (has_x) {
      if (t == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) totx[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        Ex = 0;
      }
      FSTAMP(0, sq0 + 8);
      process(bx, t, totx, Ex);
    }
    issue_x(bx, u + 1);                               // (every wave, with or without a column: the counts rely on it)
    if (has_x) {
      if (t == steps - 1) {
        const long z = (long)b * n + gw;
        float* __restrict__ sp = P.spec_x + z * (long)P.R * 64;
        const float inv = __uint_as_float((unsigned)(Ex - 14 - H2_TABLE_EXP) << 23);
        float am = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = 16 * mt + 4 * g + j;
            if (row < P.R)
              *reinterpret_cast<float4*>(sp + row * 64 + 4 * li) =
                  make_float4(totx[mt][0][j] * inv, totx[mt][1][j] * inv, totx[mt][2][j] * inv, totx[mt][3][j] * inv);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) am = fmaxf(am, fabsf(totx[mt][nt][j]));
          }
        am = wave_max(am);
        if (l == 0 && P.amax_x) P.amax_x[z] = am * inv;
      }
    }
    // ---- y axis: row 32 t + jw of the block; this wave's chunk of it, then the sum over the chunks ----
    float invy = 0.f;
    landed(by);
    if (has_y) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) toty[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
      int Ey = 0;
      FSTAMP(0, sq0 + 1);
      process(by, yc, toty, Ey);
      invy = __uint_as_float((unsigned)(Ey - 14 - H2_TABLE_EXP) << 23);
    }
    issue_y(by, u + 1);
    const long zy = (long)b * n + 32 * t + rpw * jw;      // the first of this workgroup's rows
    // the eight partial spectra of the row -> one: every wave is done with its staging area until the next step, so the
    // staging areas + the reduction area together take all partials at once ([wave][mt][nt][lane] float4, 4 MT KB per
    // wave), one pass, two barriers
    char* const land = smem + TAB;                  // STG + RED = 112 KB >= 8 waves x 12 KB
    FSTAMP(0, sq0 + 2);
    // (round 3 had a barrier here, from the time the chunks went through the staging areas that are now part of the landing
    //  zone; since the chunks stay in registers nothing reads or writes the zone between the previous step's last barrier
    //  and these writes.  RPDE_ANA_B1=1 brings it back for an A/B.)
    if (P.b1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    FSTAMP(0, sq0 + 3);
    // a lane's four fragments e = 0..3 hold channels 4 li .. 4 li + 3 of its rows: the partials land as
    // [wave][mt][j][lane] float4 over those four channels, so the sum over the waves is a float4 per (row, li) and the
    // spectrum leaves in 16-byte pieces, 256 contiguous bytes per 16 lanes (4-byte scattered stores made this phase the
    // longest of a step: 3-3.8 K of 13.5 K cycles)
    if (has_y) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4*>(land + ((wave * MT + mt) * 4 + j) * 1024 + l * 16) =
              make_float4(toty[mt][0][j] * invy, toty[mt][1][j] * invy, toty[mt][2][j] * invy, toty[mt][3][j] * invy);
    }
    FSTAMP(0, sq0 + 4);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    FSTAMP(0, sq0 + 5);
    for (int e = tid; e < rpw * MT * 256; e += 64 * ANA_WAVES) {            // (a wave's 64 items of a pass share the row)
      const int rr = e / (MT * 256), rem = e - rr * (MT * 256);
      const int mt = rem >> 8, j = (rem >> 6) & 3, ln = rem & 63;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c2 = 0; c2 < ANA_WAVES; ++c2) {             // (all reads in flight: the row's chunks sit in consecutive waves)
        if (c2 < steps) {
          const float4 v = *reinterpret_cast<const float4*>(land + (((rr * steps + c2) * MT + mt) * 4 + j) * 1024 + ln * 16);
          a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
      }
      const int row = 16 * mt + 4 * (ln >> 4) + j;
      float am = 0.f;
      if (row < P.R) {
        *reinterpret_cast<float4*>(P.spec_y + (zy + rr) * (long)P.R * 64 + row * 64 + 4 * (ln & 15)) = a;
        am = fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w)));
      }
      am = wave_max(am);
      if (l == 0) wmax[rr * ANA_WAVES + wave] = fmaxf(wmax[rr * ANA_WAVES + wave], am);   // (its own 256 bytes behind the landing zone)
    }
    FSTAMP(0, sq0 + 6);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the landing zone has been read: staging may be reused
    FSTAMP(0, sq0 + 7);
    if (tid < rpw) {                                  // (the next step's maxima are written two barriers from here)
      float a8 = 0.f;
      for (int i = 0; i < ANA_WAVES; ++i) { a8 = fmaxf(a8, wmax[tid * ANA_WAVES + i]); wmax[tid * ANA_WAVES + i] = 0.f; }
      if (P.amax_y) P.amax_y[zy + tid] = a8;
    }
    FSTAMP(0, sq0 + 9);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the re-reads past the end)
}

// ------------------------------------------------------------------------------------------------------------
// analysis of square grids, one pass over HBM and NO cross-wave reduction: the round-robin schedule (round 4)
// ------------------------------------------------------------------------------------------------------------
// k_dft_analysis_sq_h2 reads the field once, but a row's eight 32-point chunks sit in eight waves and their partial
// spectra are summed through 96 KB of LDS behind two barriers per step -- in-kernel stamps put 5.8 K of a step's 12.3 K
// cycles there, against 4.3 K for the chunks themselves.  The sum exists because in a 32-row block every wave has one
// chunk of a column (32 rows) but only an eighth of a row.  It goes away if a wave meets its WHOLE row over the rounds of
// a sample the way it already meets its whole column: cut the sample into nb x nb tiles of 32 x 32 points (nb = n / 32)
// and let wave gw of a group own row gw AND column gw.  In round t the wave in band i = gw / 32 adds
//   * to its row accumulators    the 32 points of its row    that lie in tile (i, i + t),      and
//   * to its column accumulators the 32 points of its column that lie in tile (i - t, i)       (indices mod nb).
// Tile (a, b) is then read in round b - a by the waves of band a (as rows) and by the waves of band b (as columns): both
// in the SAME round -- a tournament schedule -- so the second read finds the tile in the L2 of the XCD that the group's
// workgroups share (placement is a speed matter only), exactly the 2 MB per round that the row-block walk keeps there.
// After nb rounds every wave holds one finished row spectrum and one finished column spectrum.  No landing zone, no
// partial sums, no barrier after the table is in LDS; the table (48 KB) is all the LDS the kernel uses; every wave has a
// row and a column on every grid (a group is exactly n waves = n / 8 workgroups), so the idle-wave cases of the
// row-block kernel (and the memory fault they once caused) do not exist here.
// Loads: inline asm one round ahead, as there; behind a chunk there are always the 8 loads of the other axis's chunk, so
// a fixed vmcnt(8) covers it (the 20-odd spectrum stores of a sample's last round only make that wait longer).
// Measured (profiles/r04_analysis_variants.txt; 256^2, B = 32, same box within a pair; FETCH = 2 x FETCH_SIZE):
//   row-block kernel k_dft_analysis_sq_h2 ............................ 208 us, 551 MB fetched
//   this kernel, every wave free-running ............................. 213 us, 847 MB  (waves of a workgroup drift apart:
//                                                                              the second reader of a tile comes too late)
//   + odd bands take their row chunk first (RPDE_RR_PARITY) .......... 226 us, 838 MB
//   + ONE s_barrier per round (RPDE_RR_BARRIER; no LDS traffic) ...... 196 us, 565 MB  <- shipped
// In-kernel stamps of the shipped form (profiles/fused_stamps.py): of a round's ~13 K cycles the two chunks take 2 x 1.6 K;
// ISSUING a chunk's eight loads takes 1.5-4 K -- the CU's vector-memory pipeline accepts a 1 KB wave-load every ~100
// cycles once its queues are full, i.e. ~10 bytes per cycle and CU (the HBM-bound streaming rate of
// MI355X_MICROARCH.md's cycle table) -- and the barrier absorbs the skew that leaves.  Every byte of the field passes
// that pipeline TWICE (once per axis; the second time from L2 at about twice the rate): 2.1 MB / 10 + 2.1 MB / 23 +
// 0.66 MB / 10 bytes per cycle = 0.37 M cycles = 185 us at 2 GHz.  The kernel is at the rate of the bytes its CUs load,
// not of the bytes HBM delivers; only loading each byte once per CU would change that.
#ifndef RPDE_RR_PARITY
#define RPDE_RR_PARITY 1
#endif
#ifndef RPDE_RR_BARRIER
#define RPDE_RR_BARRIER 1
#endif
struct AnaRrP {
  const float* x; const char* timg;
  float* spec_y; float* spec_x; float* amax_y; float* amax_x;
  int B, n, nb, R, ng;
};

template <int MT>
__global__ __launch_bounds__(64 * ANA_WAVES, 2) void k_dft_analysis_rr_h2(const AnaRrP P) {
  __shared__ __attribute__((aligned(16))) char smem[ANA_MAXKS * MT * 2048];
  const int tid = threadIdx.x, l = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = l >> 4, li = l & 15;
  const int n = P.n, nb = P.nb;
  const int xg = blockIdx.x % P.ng, jw = blockIdx.x / P.ng, gw = jw * ANA_WAVES + wave;     // gw < n: this wave's row and column
  const int band = gw >> 5;
  const int nsamp = (P.B - xg + P.ng - 1) / P.ng;
  const long units = (long)nsamp * nb;               // (sample, round) pairs of this group, in order
  if (units <= 0) return;                            // (more groups than samples)
  {
    const uint4* src = reinterpret_cast<const uint4*>(P.timg);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = tid; i < nb * MT * 128; i += 64 * ANA_WAVES) dst[i] = src[i];
    __syncthreads();
  }
  const long rowf = (long)n * 64;                    // floats per row of the field
  auto gload = [](const float* p) {
    f32x4v v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
  };
  // Which chunk a wave takes first.  Tile (a, a + t) is read by band a as a ROW chunk and by band a + t as a COLUMN chunk;
  // the second read finds the tile in L2 more surely when both are requested in the same half of the round than half a
  // round apart (with "column first" everywhere 58 % of the second reads missed: 847 MB fetched for a 537 MB field).
  // Odd bands therefore take their row chunk first, even bands their column chunk: every tile of an odd round (a and
  // a + t of different parity) is then requested by both its readers in the same half.  The order is fixed per wave --
  // the two accumulator sets are bound to "first" and "second", so nothing in the loop depends on it but addresses.
  const bool row_first = RPDE_RR_PARITY && (band & 1);
  // chunk `half` (0: first, 1: second) of round u into buf: a row chunk (points of band (band + t) mod nb, lane (g, li)
  // takes points 4 i + g, channels 4 li ..) or a column chunk (rows of band (band - t) mod nb).  Past the end: a clamped
  // re-read, so that every round issues the same 16 loads.  Returns the table chunk that goes with it.
  auto chunk_of = [&](long u, int half, const float*& q0, long& st) {
    u = u < units ? u : units - 1;
    const int sb = (int)(u / nb), t = (int)(u - (long)sb * nb);
    const bool is_row = row_first == (half == 0);
    int ra = band - t; ra += ra < 0 ? nb : 0;
    int cb = band + t; cb -= cb >= nb ? nb : 0;
    const long base = (long)(xg + P.ng * sb) * n;
    q0 = is_row ? P.x + ((base + gw) * n + 32 * cb + g) * 64 + li * 4 : P.x + ((base + 32 * ra + g) * n + gw) * 64 + li * 4;
    st = is_row ? 256 : 4 * rowf;
    return is_row ? cb : ra;
  };
  auto issue = [&](f32x4v (&buf)[8], long u, int half) {
    const float* q0; long st;
    chunk_of(u, half, q0, st);
#pragma unroll
    for (int i = 0; i < 8; ++i) buf[i] = gload(q0 + i * st);
  };
  // the chunk in buf has landed once at most 8 later vector-memory instructions are outstanding: the other axis's chunk
  // (tests/test_isa_pending_loads_cpu.py follows the `landed` comment, which names the registers)
  auto landed = [&](f32x4v (&buf)[8]) {
    asm volatile("s_waitcnt vmcnt(8)\n\ts_nop 0 ; landed %0 %1 %2 %3 %4 %5 %6 %7"
                 : "+v"(buf[0]), "+v"(buf[1]), "+v"(buf[2]), "+v"(buf[3]), "+v"(buf[4]), "+v"(buf[5]), "+v"(buf[6]), "+v"(buf[7])
                 :
                 : "memory");
  };
  // one 32-point chunk into accumulators that stay in scaled units for the whole line (k_dft_analysis_sq_h2's scheme:
  // running power-of-two scale, no staging -- the loaded registers ARE the B fragments under the permuted table image)
  auto process = [&](f32x4v (&buf)[8], int s, f32x4v (&tot)[MT][4], int& line_E) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      m = fmaxf(m, fmaxf(fmaxf(fabsf(buf[i].x), fabsf(buf[i].y)), fmaxf(fabsf(buf[i].z), fabsf(buf[i].w))));
    m = wave_max(m);
    {
      const int E = max((int)(__float_as_uint(m) >> 23) & 0xff, 15 + H2_TABLE_EXP);
      if (E > line_E) {
        if (line_E > 0) {
          const float f = __uint_as_float((unsigned)(127 + line_E - E) << 23);
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
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      union { f16x8 v; struct { uint2 a, b; } u; } H, L;
      h2_split4(buf[0][e] * scale, buf[1][e] * scale, buf[2][e] * scale, buf[3][e] * scale, H.u.a, L.u.a);
      h2_split4(buf[4][e] * scale, buf[5][e] * scale, buf[6][e] * scale, buf[7][e] * scale, H.u.b, L.u.b);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const char* ta = smem + (s * MT + mt) * 2048 + l * 16;
        const f16x8 ah = *reinterpret_cast<const f16x8*>(ta), al = *reinterpret_cast<const f16x8*>(ta + 1024);
        tot[mt][e] = h2_mfma32(ah, al, H.v, L.v, tot[mt][e]);
      }
    }
  };
  // a finished line: row 16 mt + 4 g + j of the spectrum, channels 4 li .. 4 li + 3 (accumulator column li of tot[mt][e] is
  // channel 4 li + e), 256 contiguous bytes per 16 lanes; and the line's maximum for the mode mix
  auto finish = [&](f32x4v (&tot)[MT][4], int line_E, float* __restrict__ spec, float* __restrict__ amax, long z) {
    float* __restrict__ sp = spec + z * (long)P.R * 64;
    const float inv = __uint_as_float((unsigned)(line_E - 14 - H2_TABLE_EXP) << 23);
    float am = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 16 * mt + 4 * g + j;
        if (row < P.R)
          *reinterpret_cast<float4*>(sp + row * 64 + 4 * li) =
              make_float4(tot[mt][0][j] * inv, tot[mt][1][j] * inv, tot[mt][2][j] * inv, tot[mt][3][j] * inv);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) am = fmaxf(am, fabsf(tot[mt][nt][j]));
      }
    am = wave_max(am);
    if (l == 0 && amax) amax[z] = am * inv;
  };

  f32x4v tota[MT][4], totb[MT][4];                         // the line of the first / the second chunk
  int Ea = 0, Eb = 0;
  f32x4v ba[8], bb[8];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the table copy above: from here on the queue is counted by hand)
  issue(ba, 0, 0);
  issue(bb, 0, 1);
  for (long u = 0; u < units; ++u) {
    const int sb = (int)(u / nb), t = (int)(u - (long)sb * nb);
    const long z = (long)(xg + P.ng * sb) * n + gw;         // this wave's line of the sample, as a row and as a column
    if (t == 0) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) tota[mt][nt] = totb[mt][nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
      Ea = Eb = 0;
    }
    const float* q_; long st_;
    const int sa = chunk_of(u, 0, q_, st_), sbk = chunk_of(u, 1, q_, st_);
#ifdef RPDE_STAMPS
    // rounds 9 .. 12 (second sample) of wave 3 of workgroups 96 .. 159: 7 stamps each
    const bool stamp_on = l == 0 && wave == 3 && blockIdx.x >= 96 && blockIdx.x < 160 && u >= 9 && u < 13;
    const int stamp_slot = (int)blockIdx.x - 96, sq0 = (int)(u - 9) * 7;
#endif
    FSTAMP(0, sq0 + 0);
    if (RPDE_RR_BARRIER) asm volatile("s_barrier" ::: "memory");      // (keeps the eight waves of a workgroup in one round)
    FSTAMP(0, sq0 + 1);
    landed(ba);
    FSTAMP(0, sq0 + 2);
    process(ba, sa, tota, Ea);
    FSTAMP(0, sq0 + 3);
    issue(ba, u + 1, 0);
    landed(bb);
    FSTAMP(0, sq0 + 4);
    process(bb, sbk, totb, Eb);
    FSTAMP(0, sq0 + 5);
    issue(bb, u + 1, 1);
    FSTAMP(0, sq0 + 6);
    if (t == nb - 1) {
      finish(tota, Ea, row_first ? P.spec_y : P.spec_x, row_first ? P.amax_y : P.amax_x, z);
      finish(totb, Eb, row_first ? P.spec_x : P.spec_y, row_first ? P.amax_x : P.amax_y, z);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the re-reads past the end)
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
  // slots of the packed fragments that no group uses: zero (k_dft_synthesis3_h2 feeds whole fragments to the MFMAs)
  if (TG > 0 && g == 0) {
#pragma unroll
    for (int slot = 3 * TG; slot < 4 * h2_np(TG); ++slot)
      *reinterpret_cast<uint4*>(blk + (2 * K32 + slot / 4) * 1024 + ((slot & 3) * 16 + li) * 16) = make_uint4(0, 0, 0, 0);
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
// synthesis of both axes, 64 x 64 point tiles (round 3)
// ------------------------------------------------------------------------------------------------------------
// k_dft_synthesis2_h2 gives every wave a 16 x 16 x 16-channel tile and lets it fetch its 32 spectrum lines itself:
// 5.5 bytes loaded from L2 per byte stored (3.2 GB per forward at 256^2, B = 32), and that -- not HBM -- bounded it
// (317 us).  Here an 8-wave workgroup owns 64 x 64 points x 16 channels, so a line's fragments are fetched from L2 once
// per FOUR 16-point tiles: 1.5 bytes per byte stored.
//   * the 64 x-lines (columns) and 64 y-lines (rows) of the tile stream through a 3-slot LDS ring in 8 stages of 16
//     lines, by LDS-DMA (global_load_lds: 1 KB fragment pieces land lane-linear, exactly as the MFMA wants them; no
//     registers held), two stages ahead; one s_barrier per stage.  The table blocks and the line scales travel the
//     same way (a 12 KB corner of LDS that holds the x-axis table during the x phase and the y-axis table after it):
//     no ordinary global load anywhere in the kernel, so the compiler never drains the DMA queue for one.
//   * wave (mt, p) owns the 16-row block mt and the two 16-column blocks p, p + 2: 128 accumulator registers.  Stages
//     0-3 form the x-axis part column by column (accumulator rows = m), stages 4-7 the y-axis part row by row
//     (accumulator rows = n).  Between them the x part is turned into the y layout INSIDE the register file: both
//     layouts keep the channel on lane & 15, so the turn is sixteen 4 x 4 transposes between the register index and
//     the lane-row index -- two v_permlane32_swap + two v_permlane16_swap each.  No LDS turning buffer (the old kernel
//     spent ~200 LDS instructions and 2.1 M bank-conflict cycles per launch there).
//   * persistent: one workgroup per CU walks over its share of the tiles, and the ring never runs dry -- the first two
//     stages (and the tables) of the NEXT tile are requested during the last stages of the current one.  In-kernel
//     stamps of the first, one-tile-per-workgroup version: 33 % of a workgroup's life was the cold start of its DMA
//     queue, 29 % the burst of stores at its end.
//   * forward: the stores of a tile are spread over eight stages, four 16-byte stores per wave and stage (a 4 x 4
//     transpose inside each quad of lanes gives every lane four consecutive channels of one point); column block 1 of
//     a tile leaves during the x stages of the next tile.  Adjoint with a skip gradient: loads + stores in four batches
//     after the last stage.
//   * what bounds it now (measured with the parts switched off one at a time, 256^2, B = 32): DMA skeleton alone 76 us,
//     + MFMAs 25 us, + stores 130 us; HBM traffic 0.30 GB read + 0.54 GB written in 215-230 us.  Burst or spread, in step
//     or staggered across XCDs, 4-byte or 16-byte stores: the store time adds to the rest (a CU's memory pipeline
//     takes a 16-segment store instruction in ~60-90 cycles and DMA pieces through the same queue).
//   * every line carries its own power-of-two scale (undone on the accumulators, as before).
constexpr int SYN3_WAVES = 8;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// a and b hold one value per lane; afterwards a's upper half (lane-rows 2, 3) and b's lower half (rows 0, 1) have changed
// places / a's odd lane-rows and b's even lane-rows have changed places
__device__ __forceinline__ void swap_halves(float& a, float& b) {
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x); b = __uint_as_float(r.y);
}
__device__ __forceinline__ void swap_rows(float& a, float& b) {
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r.x); b = __uint_as_float(r.y);
}

// 4 x 4 transpose inside every quad of lanes: afterwards register c of lane p (p = lane & 3) holds what register p of
// lane c held.  Two butterfly stages on DPP quad permutes (lane ^ 1, then lane ^ 2).
template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ void quad_transpose(float& v0, float& v1, float& v2, float& v3) {
  const bool b0 = threadIdx.x & 1, b1 = threadIdx.x & 2;
  { const float a = quad_dpp<0xB1>(v0), c = quad_dpp<0xB1>(v1); v1 = b0 ? v1 : a; v0 = b0 ? c : v0; }   // quad_perm [1,0,3,2]
  { const float a = quad_dpp<0xB1>(v2), c = quad_dpp<0xB1>(v3); v3 = b0 ? v3 : a; v2 = b0 ? c : v2; }
  { const float a = quad_dpp<0x4E>(v0), c = quad_dpp<0x4E>(v2); v2 = b1 ? v2 : a; v0 = b1 ? c : v0; }   // quad_perm [2,3,0,1]
  { const float a = quad_dpp<0x4E>(v1), c = quad_dpp<0x4E>(v3); v3 = b1 ? v3 : a; v1 = b1 ? c : v1; }
}

// makes the compiler produce x here, in program order relative to the other volatile asm statements (barriers, waits):
// without it the VALU work between the two phases (scaling, turn, the y phase's multiply-adds) is sunk to the end of
// the kernel and both phases' 128 accumulators are alive at once (80 spilled registers)
__device__ __forceinline__ void pin(float& x) { asm volatile("" : "+v"(x)); }

// one tile of one channel block of one sample
struct Syn3Item {
  const char* gx; const char* gy;       // fragment pieces of the tile's first column / first row (this channel block)
  const char* tx; const char* ty;       // table blocks of the tile's first 16 rows / first 16 columns
  const float* ivx; const float* ivy;   // line scales of the tile's columns / rows
  long o;                               // float offset of the tile's first point, first channel of the block
};

template <int K32, int TG, bool SKIP>
__global__ __launch_bounds__(64 * SYN3_WAVES, 2) void k_dft_synthesis3_h2(const SynP P) {
  constexpr int NP = h2_np(TG), NF = 2 * K32 + NP;
  constexpr int BB = NF * 1024;
  constexpr long LB = 4L * BB;
  constexpr int SLOT = 16 * BB;                 // one stage: 16 lines
  static_assert(NF <= 3, "ring of three 16-line slots + the table corner must fit 160 KB");
  constexpr int TAB = 3 * SLOT, INV = TAB + 4 * BB, LDS_BYTES = INV + 512;
  constexpr int NST = SKIP ? 0 : 8;             // stores a wave issues at the end of a y stage
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = l >> 4, li = l & 15;
  const int mt = w & 3, p = w >> 2;
  const unsigned lane16 = l * 16;
  typedef Frag<K32, TG> F;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;

  // ---- this workgroup's tiles.  Blocks b, b + 8, .. share an XCD (an L2): group x = blockIdx & 7 takes the samples
  // x, x + 8, ..; its workgroups take the items round-robin, so the ~32 in flight are neighbours that share spectrum
  // lines.  (Placement is a speed matter only.)
  const int tn = P.N >> 6, wps = (P.M >> 6) * tn * 4;
  const int ng = P.sy;            // groups: 8, or the batch size when that is smaller
  const int xg = blockIdx.x % ng, jw = blockIdx.x / ng, nj = gridDim.x / ng;
  const int nsamp = (P.B - xg + ng - 1) / ng;
  const long nitems = (long)nsamp * wps;
  auto item_at = [&](long q) {
    Syn3Item it;
    const int sb = (int)(q / wps), t = (int)(q - (long)sb * wps);
    // channel blocks in pairs: first every tile of the sample for blocks 0 and 1, then for 2 and 3.  The ~32 workgroups
    // of a group then work on ONE pair at a time -- the pair's fragments of a 256^2 sample (3.2 MB) stay in the 4 MB L2
    // while all sixteen tiles use them (with the four blocks of a tile side by side only half of a sample's tiles ran
    // together and the other half fetched the lines again: 319 MB for 201 MB of operands) -- and the two blocks of a
    // pair still complete each other's 128-byte lines in L2.
    const int half = wps >> 1, hb = t / half, r = t - hb * half;
    const int b = xg + ng * sb, cb = 2 * hb + (r & 1), tile = r >> 1;
    const int m0 = (tile / tn) << 6, n0 = (tile % tn) << 6;
    it.gx = P.imgx + ((long)b * P.N + n0) * LB + cb * BB;
    it.gy = P.imgy + ((long)b * P.M + m0) * LB + cb * BB;
    it.tx = P.tabx + (long)(m0 >> 4) * BB;
    it.ty = P.taby + (long)(n0 >> 4) * BB;
    it.ivx = P.invx + (long)b * P.N + n0;
    it.ivy = P.invy + (long)b * P.M + m0;
    it.o = (((long)b * P.M + m0) * P.N + n0) * 64 + 16 * cb;
    return it;
  };
  long q = jw;
  if (q >= nitems) return;
  Syn3Item cur = item_at(q);

  // ---- DMA ----
  // piece i (of 2 NF) of this wave for stage s of a tile: stage s < 4: columns 8 (s & 1) .. + 7 of the column blocks
  // (s >> 1) * 2 + {0, 1}; stage s >= 4: rows 4 (s - 4) .. + 3 of the four row blocks
  auto issue_piece = [&](const Syn3Item& it, int s, int i, int slot) {
    const int idx = w + 8 * i, ell = idx / NF, f = idx - ell * NF;     // (ell * NF + f == idx: pieces land in order)
    const char* src;
    if (s < 4) src = it.gx + (long)(16 * ((ell >> 3) + 2 * (s >> 1)) + 8 * (s & 1) + (ell & 7)) * LB + f * 1024;
    else src = it.gy + (long)(16 * (ell >> 2) + 4 * (s - 4) + (ell & 3)) * LB + f * 1024;
    __builtin_amdgcn_global_load_lds((glb_ptr)(src + lane16), (lds_ptr)(smem + slot * SLOT + idx * 1024), 16, 0, 0);
  };
  // the four 16-row blocks of a table (x axis: rows m0 .. m0 + 63; y axis: the column blocks p + 2a at [p * 2 + a])
  auto issue_table = [&](const char* tab, bool yaxis) {
#pragma unroll
    for (int i = 0; i < (4 * NF + 7) / 8; ++i) {
      const int idx = w + 8 * i;
      if (idx < 4 * NF) {
        const int blk = idx / NF, f = idx - blk * NF;
        const int src_blk = yaxis ? (blk >> 1) + 2 * (blk & 1) : blk;
        __builtin_amdgcn_global_load_lds((glb_ptr)(tab + (long)src_blk * BB + f * 1024 + lane16), (lds_ptr)(smem + TAB + idx * 1024), 16, 0, 0);
      }
    }
  };
  auto issue_inv = [&](const Syn3Item& it) {          // 64 + 64 floats, one 4-byte DMA each
    if (w == 6) __builtin_amdgcn_global_load_lds((glb_ptr)(it.ivx + l), (lds_ptr)(smem + INV), 4, 0, 0);
    if (w == 7) __builtin_amdgcn_global_load_lds((glb_ptr)(it.ivy + l), (lds_ptr)(smem + INV + 256), 4, 0, 0);
  };

  auto lds_frag = [&](F& f, const char* base) {
#pragma unroll
    for (int s2 = 0; s2 < K32; ++s2) {
      f.h[s2] = *reinterpret_cast<const f16x8*>(base + s2 * 1024);
      f.lo[s2] = *reinterpret_cast<const f16x8*>(base + (K32 + s2) * 1024);
    }
#pragma unroll
    for (int qq = 0; qq < NP; ++qq) f.pk[qq] = *reinterpret_cast<const f16x8*>(base + (2 * K32 + qq) * 1024);
  };
  auto chain = [&](const F& tab, const F& f) {
    f32x4v c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qq = 0; qq < NP; ++qq) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.pk[qq], f.pk[qq], c, 0, 0, 0);
#pragma unroll
    for (int s2 = K32 - 1; s2 >= 0; --s2) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.lo[s2], f.h[s2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.h[s2], f.lo[s2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.h[s2], f.h[s2], c, 0, 0, 0);
    }
    return c;
  };

  // ---- prologue: tables / scales of the first tile, then its first two stages ----
  issue_table(cur.tx, false);
  issue_inv(cur);
#pragma unroll
  for (int i = 0; i < 2 * NF; ++i) issue_piece(cur, 0, i, 0);
  cw_mark<3>();                 // (tags 3 / 4: the first tile's stages 0 / 1, waited for under `first`; see k_dft_synthesis4_h2)
#pragma unroll
  for (int i = 0; i < 2 * NF; ++i) issue_piece(cur, 1, i, 1);
  cw_mark<4>();
  int ring = 0;                 // slot of the current tile's stage 0
  bool first = true;
  const int qd = li >> 2, pl = li & 3;
  // acc[a][i][jj]: x phase -- column i of column block a, rows m = 4g + jj; after the turn, physical register
  // [a][4 (r >> 2) + j][r & 3] holds row r, column 4g + j.  Lives across tiles: column block 1 of a finished tile is
  // stored while the x stages of the next tile refill column block 0 (below).
  float acc[2][16][4];
  long o_prev = 0;

  while (true) {
    const long qn = q + nj;
    const bool has_next = qn < nitems;
    Syn3Item nxt = cur;
    if (has_next) nxt = item_at(qn);
    F tx, ty[2];
    // line fragments are read one line ahead of their MFMAs, no further (the compiler barrier keeps the LDS reads of
    // later lines from being hoisted on top of 128 live accumulators)
    F fq[2];
    float invx_l[2], invy_l;
    // output: after the quad transpose lane (g, 4 qd + pl) holds channels 4 qd .. + 3 of the point (row r, column
    // 16 (p + 2a) + 4g + pl)
    const long o = cur.o + ((long)(16 * mt) * P.N + 16 * p + 4 * g + pl) * 64 + 4 * qd;
    auto tile_out = [&](int r, int a) {
      float4 v = make_float4(acc[a][4 * (r >> 2) + 0][r & 3], acc[a][4 * (r >> 2) + 1][r & 3], acc[a][4 * (r >> 2) + 2][r & 3],
                             acc[a][4 * (r >> 2) + 3][r & 3]);
      quad_transpose(v.x, v.y, v.z, v.w);
      return v;
    };
    // forward: the stores of a tile are spread evenly over EIGHT stages, four per wave and stage -- a CU's store path
    // takes ~60-90 cycles per 16-segment store instruction, so the 256 KB of a tile need about as long as all its other
    // work, and issued in one burst (at the end of the tile, or over its four y stages) they simply added their time
    // to it (measured: 139 of 229 us).  Rows 4k .. 4k + 3 of column block 0 leave at the end of y stage k (they are
    // final then), rows 4k .. 4k + 3 of column block 1 at the START of x stage k of the NEXT tile: x stage s refills
    // the registers of rows 8 (s & 1) .. + 7 of column block s >> 1, so block 0 is free when the next tile begins and
    // every row group of block 1 has left before its registers are written again.
    auto store_rows = [&](long obase, int k, int a) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = 4 * k + rr;
        *reinterpret_cast<float4*>(P.out + obase + ((long)r * P.N + 32 * a) * 64) = tile_out(r, a);
      }
    };
#ifdef RPDE_STAMPS
    const bool stamp_on = l == 0 && w == 5 && blockIdx.x >= 96 && blockIdx.x < 160 && q == jw + 2 * (long)nj;
    const int stamp_slot = (int)blockIdx.x - 96;
#endif
    FSTAMP(1, 0);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      // ---- this wave's pieces of stage s have landed.  vmcnt(N): everything but the N youngest operations of this wave
      // is done; younger than the pieces of stage s are the pieces of stage s + 1 (2 NF) and the stores of the y stages
      // in between (NST each); a wait that also covers table / scale pieces issued in between only over-waits
      // (forward: + the stores in between: 4 at the end of every y stage, 4 at the start of every x stage but the first tile's)
      // (cw.h: the requests of stage s are marked with the tag s % 3 behind their last piece, and the ISA test checks
      //  these counts on the compiled code)
      constexpr int Q = SKIP ? 0 : 4;
      const int T = s % 3;
      // (every stage requests its pieces, the last tile's stages 6 / 7 a harmless re-read: no count depends on has_next.
      //  The forward instances' counts for stages 2 .. 5 still depend on `first` through the stores at the start of the x
      //  stages, which the path-insensitive ISA check cannot follow: it covers the SKIP instances -- the ones the default
      //  dispatch uses; the forward runs k_dft_synthesis4_h2.)
      if (SKIP && !first && s < 2) cw_wait_t<63>(T);                       // (behind the 64 loads / stores of the epilogue)
      else if (first && s < 2) { if (s == 0) cw_wait<3, 2 * NF>(); else cw_wait<4, 2 * NF>(); }
      else if (first && s < 5) cw_wait_t<2 * NF>(T);
      else if (s == 0 || s == 1 || s == 6) cw_wait_t<2 * NF + 2 * Q>(T);
      else if (s <= 5) cw_wait_t<2 * NF + Q>(T);
      else cw_wait_t<2 * NF + 2 * Q>(T);
      FSTAMP(1, 1 + 3 * s);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everyone's have; everyone is done with stage s - 1
      FSTAMP(1, 2 + 3 * s);
      const int slot_s = (ring + s) % 3, slot_n = (ring + s + 2) % 3;
      // what this stage requests (spread between its MFMA chains): stage s + 2 of this tile, or stage s - 6 of the next
      auto request = [&](int i) {
        if (i >= 2 * NF) return;
        if (s + 2 < 8) issue_piece(cur, s + 2, i, slot_n);
        else issue_piece(nxt, s - 6, i, slot_n);               // (last tile: nxt = cur, a re-read nobody uses)
      };
      // tables: the y-axis blocks replace the x-axis blocks once every wave has its x block in registers (stage 0); the
      // next tile's x blocks and scales replace them once every wave has its y blocks (stage 4)
      if (!SKIP && s < 4 && !first) store_rows(o_prev, s, 1);
      if (s == 1) issue_table(cur.ty, true);
      if (s == 5 && has_next) { issue_table(nxt.tx, false); issue_inv(nxt); }
      const char* slot = smem + slot_s * SLOT + l * 16;
      if (s < 4) {
        const int a = s >> 1, h = s & 1;
        if (s == 0) {
          lds_frag(tx, smem + TAB + mt * BB + l * 16);
          invx_l[0] = *reinterpret_cast<const float*>(smem + INV + (16 * p + li) * 4);
          invx_l[1] = *reinterpret_cast<const float*>(smem + INV + (16 * (p + 2) + li) * 4);
          invy_l = *reinterpret_cast<const float*>(smem + INV + 256 + (16 * mt + li) * 4);
        }
        lds_frag(fq[0], slot + (p * 8) * BB);
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
          if (c8 + 1 < 8) lds_frag(fq[(c8 + 1) & 1], slot + (p * 8 + c8 + 1) * BB);
          asm volatile("" ::: "memory");
          const f32x4v c = chain(tx, fq[c8 & 1]);
          request(c8);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[a][8 * h + c8][jj] = c[jj];
        }
        cw_mark_t((s + 2) % 3);                        // the pieces of stage s + 2 are all requested
        if (s == 3) {
          // ---- x part -> true units, then turned into the y layout ----
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invx_l[a2]), i));
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) acc[a2][i][jj] *= sc;
            }
#pragma unroll
            for (int jy = 0; jy < 4; ++jy)
#pragma unroll
              for (int jr = 0; jr < 4; ++jr) {
                swap_halves(acc[a2][jy][jr], acc[a2][8 + jy][jr]);
                swap_halves(acc[a2][4 + jy][jr], acc[a2][12 + jy][jr]);
                swap_rows(acc[a2][jy][jr], acc[a2][4 + jy][jr]);
                swap_rows(acc[a2][8 + jy][jr], acc[a2][12 + jy][jr]);
              }
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) pin(acc[a2][i][jj]);
          }
        }
      } else {
        const int sy = s - 4;
        if (sy == 0) {
          lds_frag(ty[0], smem + TAB + ((p * 2 + 0) * NF) * 1024 + l * 16);
          lds_frag(ty[1], smem + TAB + ((p * 2 + 1) * NF) * 1024 + l * 16);
        }
        lds_frag(fq[0], slot + (mt * 4) * BB);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = 4 * sy + rr;
          if (rr + 1 < 4) lds_frag(fq[(rr + 1) & 1], slot + (mt * 4 + rr + 1) * BB);
          asm volatile("" ::: "memory");
          const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invy_l), r));
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const f32x4v c = chain(ty[a], fq[rr & 1]);
            request(2 * rr + a);                                  // (2 NF <= 6 pieces: all out before the stage's stores)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[a][4 * (r >> 2) + j][r & 3] = fmaf(c[j], sc, acc[a][4 * (r >> 2) + j][r & 3]);
              pin(acc[a][4 * (r >> 2) + j][r & 3]);
            }
          }
        }
        cw_mark_t(s + 2 < 8 ? (s + 2) % 3 : (s - 6) % 3);      // the pieces of stage s + 2 (of the next tile: s - 6)
        if (!SKIP) store_rows(o, sy, 0);
      }
      FSTAMP(1, 3 + 3 * s);
    }
    if (SKIP) {
      // the tensor added to the result (the gradient that arrives through the skip connection) is fetched in four
      // batches of 8 float4, each batch a step ahead of the additions and stores it feeds: loads and stores share
      // one in-order counter, so load - add - store per element would wait for every store before the next load
      float4 sk[2][8];
      auto fetch = [&](float4 (&d)[8], int k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = *reinterpret_cast<const float4*>(P.skip + o + ((long)(4 * k + (e >> 1)) * P.N + 32 * (e & 1)) * 64);
      };
      fetch(sk[0], 0);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k + 1 < 4) fetch(sk[(k + 1) & 1], k + 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = 4 * k + (e >> 1), a = e & 1;
          float4 v = tile_out(r, a);
          const float4 q4 = sk[k & 1][e];
          v.x += q4.x; v.y += q4.y; v.z += q4.z; v.w += q4.w;
          *reinterpret_cast<float4*>(P.out + o + ((long)r * P.N + 32 * a) * 64) = v;
        }
      }
    }
    FSTAMP(1, 26);
    if (!has_next) {
      if (!SKIP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) store_rows(o, k, 1);        // the last tile's column block 1
      }
      wait_vmcnt<0>();          // (the re-reads of the last two stages must have landed before the LDS is given back)
      break;
    }
    o_prev = o;
    cur = nxt;
    q = qn;
    ring = (ring + 8) % 3;
    first = false;
  }
}

// ------------------------------------------------------------------------------------------------------------
// synthesis of both axes, 64 x 32 point tiles x 32 channels: whole 128-byte lines per store (round 4)
// ------------------------------------------------------------------------------------------------------------
// k_dft_synthesis3_h2 owns 16 channels of its points, so every store instruction writes sixteen 64-byte segments -- half
// cache lines -- and a CU's memory pipeline takes such an instruction at a quarter of the rate of one that writes eight
// whole 128-byte lines (profiles/ubench/storepat.hip: 37 vs 135 GB/s per CU); with the stores in, that kernel ran 225 us
// against 100 us without them.  Here a workgroup owns a PAIR of channel blocks (32 channels = one 128-byte line per
// point) of a 64-row x 32-column tile -- the same 128 accumulator registers per lane -- at the price of 2.25 instead of
// 1.5 bytes of fragments fetched from L2 per byte stored (the fragments of a channel pair of a sample, 3.1 MB at 256^2,
// still fit the XCD's L2 while all its tiles use them).
//   * wave (mt, p) owns rows 16 mt .., columns 16 p .. of the tile for BOTH channel blocks a = 0, 1: acc[a][i][jj].
//   * twelve stages of 16 (line, channel block) fragments through the same 3-slot ring: x stages 0-3 (channel block
//     s >> 1, columns 8 (s & 1) .. + 7 of both column blocks), y stages 4-11 (rows 2 (s - 4), + 1 of the four row blocks,
//     both channel blocks).  12 = 0 mod 3: the ring position is the same for every tile.
//   * a row is final after its y stage and leaves right there: four store instructions per wave and y stage, each
//     eight points x 128 bytes.  The 4 x 4 quad transpose that gives a lane four consecutive channels of one point
//     takes its four inputs from BOTH channel blocks -- (a, column) = (0, 0), (0, 1), (1, 0), (1, 1) -- so the lanes
//     of a quad end up with the two 64-byte halves of two points' lines: no exchange beyond the transpose itself.
//   * forward and adjoint without a skip gradient (with one: k_dft_synthesis3_h2<.., true>).
template <int S> struct StageC { static constexpr int value = S; };

// Tried on top and dropped (same-box, 256^2, B = 32; profiles/r04_synthesis_variants.txt): requesting ALL pieces of stage
// s + 2 right behind the barrier and issuing the stores of a stage's rows one stage later, behind the next requests, so
// that a store has three stages instead of two to be acknowledged before a wait on younger pieces covers it (the counter
// is in order) and no request queues behind a burst of stores: 225-230 us against 219 us.  Neither the store segment
// (this kernel against k_dft_synthesis3_h2: 219 vs 224-235 us), nor the fragment volume (2.25 vs 1.5 bytes per byte
// stored), nor the store window moves the time; the same store pattern alone streams at 6.2 TB/s = 87 us
// (profiles/ubench/storepat.hip, pattern C / D).  What is left is that ONE workgroup per CU runs all its waves through
// the same phase at the same time -- wait for pieces | fragments + matrix work | transposes + stores -- so the three
// add up instead of overlapping.
template <int K32, int TG>
__global__ __launch_bounds__(64 * SYN3_WAVES, 2) void k_dft_synthesis4_h2(const SynP P) {
  constexpr int NP = h2_np(TG), NF = 2 * K32 + NP;
  constexpr int BB = NF * 1024;
  constexpr long LB = 4L * BB;
  constexpr int SLOT = 16 * BB;                 // one stage: 16 fragments
  static_assert(NF <= 3, "ring of three 16-fragment slots + the table corner must fit 160 KB");
  constexpr int TAB = 3 * SLOT, INV = TAB + 4 * BB, LDS_BYTES = INV + 512;
  constexpr int NS = 12;                        // stages per tile
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  const int tid = threadIdx.x, l = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), g = l >> 4, li = l & 15;
  const int mt = w & 3, p = w >> 2;
  const unsigned lane16 = l * 16;
  typedef Frag<K32, TG> F;
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef const __attribute__((address_space(1))) void* glb_ptr;

  // ---- items: (sample, channel pair, tile); group x = blockIdx % ng takes the samples x, x + ng, ..; within a sample all
  // tiles of channel pair 0, then all of pair 1 (a pair's fragments stay in the group's L2 meanwhile)
  const int tn = P.N >> 5, tiles = (P.M >> 6) * tn, wps = tiles * 2;
  const int ng = P.sy;
  const int xg = blockIdx.x % ng, jw = blockIdx.x / ng, nj = gridDim.x / ng;
  const int nsamp = (P.B - xg + ng - 1) / ng;
  const long nitems = (long)nsamp * wps;
  auto item_at = [&](long q) {
    Syn3Item it;
    const int sb = (int)(q / wps), t = (int)(q - (long)sb * wps);
    const int hb = t / tiles, tile = t - hb * tiles;
    const int b = xg + ng * sb;
    const int m0 = (tile / tn) << 6, n0 = (tile % tn) << 5;
    it.gx = P.imgx + ((long)b * P.N + n0) * LB + (2 * hb) * BB;
    it.gy = P.imgy + ((long)b * P.M + m0) * LB + (2 * hb) * BB;
    it.tx = P.tabx + (long)(m0 >> 4) * BB;
    it.ty = P.taby + (long)(n0 >> 4) * BB;
    it.ivx = P.invx + (long)b * P.N + n0;
    it.ivy = P.invy + (long)b * P.M + m0;
    it.o = (((long)b * P.M + m0) * P.N + n0) * 64 + 32 * hb;
    return it;
  };
  long q = jw;
  if (q >= nitems) return;
  Syn3Item cur = item_at(q);

  // piece i (of 2 NF) of this wave for stage s of a tile
  auto issue_piece = [&](const Syn3Item& it, int s, int i, int slot) {
    const int idx = w + 8 * i, ell = idx / NF, f = idx - ell * NF;     // fragment ell of the stage, piece f of it
    const char* src;
    if (s < 4) src = it.gx + (long)(16 * (ell >> 3) + 8 * (s & 1) + (ell & 7)) * LB + (s >> 1) * BB + f * 1024;
    else src = it.gy + (long)(16 * (ell >> 2) + 2 * (s - 4) + ((ell >> 1) & 1)) * LB + (ell & 1) * BB + f * 1024;
    __builtin_amdgcn_global_load_lds((glb_ptr)(src + lane16), (lds_ptr)(smem + slot * SLOT + idx * 1024), 16, 0, 0);
  };
  // nblk consecutive 16-row blocks of a table (x axis: the tile's four row blocks; y axis: its two column blocks)
  auto issue_table = [&](const char* tab, int nblk) {
#pragma unroll
    for (int i = 0; i < (4 * NF + 7) / 8; ++i) {
      const int idx = w + 8 * i;
      if (idx < nblk * NF)
        __builtin_amdgcn_global_load_lds((glb_ptr)(tab + (long)idx * 1024 + lane16), (lds_ptr)(smem + TAB + idx * 1024), 16, 0, 0);
    }
  };
  auto issue_inv = [&](const Syn3Item& it) {          // 32 + 64 floats, one 4-byte DMA each
    if (w == 6 && l < 32) __builtin_amdgcn_global_load_lds((glb_ptr)(it.ivx + l), (lds_ptr)(smem + INV), 4, 0, 0);
    if (w == 7) __builtin_amdgcn_global_load_lds((glb_ptr)(it.ivy + l), (lds_ptr)(smem + INV + 256), 4, 0, 0);
  };
  auto lds_frag = [&](F& f, const char* base) {
#pragma unroll
    for (int s2 = 0; s2 < K32; ++s2) {
      f.h[s2] = *reinterpret_cast<const f16x8*>(base + s2 * 1024);
      f.lo[s2] = *reinterpret_cast<const f16x8*>(base + (K32 + s2) * 1024);
    }
#pragma unroll
    for (int qq = 0; qq < NP; ++qq) f.pk[qq] = *reinterpret_cast<const f16x8*>(base + (2 * K32 + qq) * 1024);
  };
  auto chain = [&](const F& tab, const F& f) {
    f32x4v c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int qq = 0; qq < NP; ++qq) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.pk[qq], f.pk[qq], c, 0, 0, 0);
#pragma unroll
    for (int s2 = K32 - 1; s2 >= 0; --s2) {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.lo[s2], f.h[s2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.h[s2], f.lo[s2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(tab.h[s2], f.h[s2], c, 0, 0, 0);
    }
    return c;
  };

  // ---- prologue: tables / scales of the first tile, then its first two stages ----
  issue_table(cur.tx, 4);
  issue_inv(cur);
#pragma unroll
  for (int i = 0; i < 2 * NF; ++i) issue_piece(cur, 0, i, 0);
  cw_mark<3>();                 // (tags 3 / 4: the first tile's stages 0 / 1, waited for under `first` -- a path-insensitive
#pragma unroll                  //  check cannot know that `first` and "came from the prologue" are the same thing)
  for (int i = 0; i < 2 * NF; ++i) issue_piece(cur, 1, i, 1);
  cw_mark<4>();
  bool first = true;
  const int qd = li >> 2, pl = li & 3;
  // acc[a][i][jj]: x phase -- channel block a, column i of the wave's column block, rows m = 4g + jj; after the turn,
  // physical register [a][4 (r >> 2) + j][r & 3] holds row r, column 4g + j
  float acc[2][16][4];

  while (true) {
    const long qn = q + nj;
    const bool has_next = qn < nitems;
    Syn3Item nxt = cur;
    if (has_next) nxt = item_at(qn);
    F tx, ty;
    F fq[2];
    float invx_l, invy_l;
    // a lane's share of a store instruction: the two points (row r, column 16 p + 4 g + 2 h + (pl & 1)), h = 0, 1; of each
    // the 16 bytes of channels 16 (pl >> 1) + 4 qd .. + 3 of the pair
    const long o = cur.o + ((long)(16 * mt) * P.N + 16 * p + 4 * g + (pl & 1)) * 64 + 16 * (pl >> 1) + 4 * qd;
    auto store_row = [&](long ob, int r) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 v = make_float4(acc[0][4 * (r >> 2) + 2 * h][r & 3], acc[0][4 * (r >> 2) + 2 * h + 1][r & 3],
                               acc[1][4 * (r >> 2) + 2 * h][r & 3], acc[1][4 * (r >> 2) + 2 * h + 1][r & 3]);
        quad_transpose(v.x, v.y, v.z, v.w);
        *reinterpret_cast<float4*>(P.out + ob + ((long)r * P.N + 2 * h) * 64) = v;
      }
    };
    // (one call per stage with the stage number as a type: left as a loop of twelve the optimizer gives up unrolling the
    //  instances with three fragment pieces, and the accumulators -- indexed by the stage -- go to scratch memory)
    auto stage = [&](auto stage_c) {
      constexpr int s = decltype(stage_c)::value;
      // ---- this wave's pieces of stage s have landed: all but the N youngest operations of the wave are done; younger
      // than the pieces of stage s are the stores of stage s - 2 (4 if it was a y stage), the pieces of stage s + 1
      // (2 NF) and the stores of stage s - 1; table / scale pieces issued in between only make the wait longer
      // (cw.h: tag = ring slot; tests/test_isa_counted_waits_cpu.py checks these counts on the compiled code)
      constexpr int T = s % 3;
      // (every stage requests its 2 NF pieces, the last tile's stages 10 / 11 a harmless re-read: the counts do not
      //  depend on has_next)
      if (s == 0) { if (first) cw_wait<3, 2 * NF>(); else cw_wait<T, 2 * NF + 8>(); }
      else if (s == 1) { if (first) cw_wait<4, 2 * NF>(); else cw_wait<T, 2 * NF + 4>(); }
      else if (s <= 4) cw_wait<T, 2 * NF>();
      else if (s == 5) cw_wait<T, 2 * NF + 4>();
      else cw_wait<T, 2 * NF + 8>();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // everyone's have; everyone is done with stage s - 1
      const int slot_s = s % 3, slot_n = (s + 2) % 3;
      auto request = [&](int i) {
        if (i >= 2 * NF) return;
        if (s + 2 < NS) issue_piece(cur, s + 2, i, slot_n);
        else issue_piece(nxt, s + 2 - NS, i, slot_n);          // (last tile: nxt = cur, a re-read nobody uses)
      };
      // tables: the y-axis blocks replace the x-axis blocks once every wave has its x block in registers (stage 0); the
      // next tile's x blocks and scales replace them once every wave has its y block (stage 4)
      if (s == 1) issue_table(cur.ty, 2);
      if (s == 5 && has_next) { issue_table(nxt.tx, 4); issue_inv(nxt); }
      const char* slot = smem + slot_s * SLOT + l * 16;
      if (s < 4) {
        const int a = s >> 1, h = s & 1;
        if (s == 0) {
          lds_frag(tx, smem + TAB + mt * BB + l * 16);
          invx_l = *reinterpret_cast<const float*>(smem + INV + (16 * p + li) * 4);
          invy_l = *reinterpret_cast<const float*>(smem + INV + 256 + (16 * mt + li) * 4);
        }
        lds_frag(fq[0], slot + (p * 8) * BB);
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
          if (c8 + 1 < 8) lds_frag(fq[(c8 + 1) & 1], slot + (p * 8 + c8 + 1) * BB);
          asm volatile("" ::: "memory");
          const f32x4v c = chain(tx, fq[c8 & 1]);
          request(c8);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[a][8 * h + c8][jj] = c[jj];
        }
        cw_mark<(s + 2) % 3>();                        // the pieces of stage s + 2 are all requested
        if (s == 3) {
          // ---- x part -> true units, then turned into the y layout (k_dft_synthesis3_h2's turn) ----
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invx_l), i));
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) acc[a2][i][jj] *= sc;
            }
#pragma unroll
            for (int jy = 0; jy < 4; ++jy)
#pragma unroll
              for (int jr = 0; jr < 4; ++jr) {
                swap_halves(acc[a2][jy][jr], acc[a2][8 + jy][jr]);
                swap_halves(acc[a2][4 + jy][jr], acc[a2][12 + jy][jr]);
                swap_rows(acc[a2][jy][jr], acc[a2][4 + jy][jr]);
                swap_rows(acc[a2][8 + jy][jr], acc[a2][12 + jy][jr]);
              }
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) pin(acc[a2][i][jj]);
          }
        }
      } else {
        const int sy = s - 4;
        if (sy == 0) lds_frag(ty, smem + TAB + p * BB + l * 16);
        lds_frag(fq[0], slot + (mt * 4) * BB);
#pragma unroll
        for (int e = 0; e < 4; ++e) {                   // fragment mt * 4 + e: row 2 sy + (e >> 1), channel block e & 1
          const int r = 2 * sy + (e >> 1), a = e & 1;
          if (e + 1 < 4) lds_frag(fq[(e + 1) & 1], slot + (mt * 4 + e + 1) * BB);
          asm volatile("" ::: "memory");
          const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(invy_l), r));
          const f32x4v c = chain(ty, fq[e & 1]);
          request(2 * e);
          request(2 * e + 1);                           // (2 NF <= 6 pieces: all out before the stage's stores)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[a][4 * (r >> 2) + j][r & 3] = fmaf(c[j], sc, acc[a][4 * (r >> 2) + j][r & 3]);
            pin(acc[a][4 * (r >> 2) + j][r & 3]);
          }
        }
        cw_mark<(s + 2) % 3>();                        // the pieces of stage s + 2 (of the next tile: s + 2 - 12)
        store_row(o, 2 * sy);
        store_row(o, 2 * sy + 1);
      }
    };
    stage(StageC<0>{}); stage(StageC<1>{}); stage(StageC<2>{}); stage(StageC<3>{});
    stage(StageC<4>{}); stage(StageC<5>{}); stage(StageC<6>{}); stage(StageC<7>{});
    stage(StageC<8>{}); stage(StageC<9>{}); stage(StageC<10>{}); stage(StageC<11>{});
    if (!has_next) {
      wait_vmcnt<0>();          // (the re-reads of the last two stages must have landed before the LDS is given back)
      break;
    }
    cur = nxt;
    q = qn;
    first = false;
  }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
bool fused2d_ok(int M, int N, int C, int keff_y, int keff_x) {
  const int Ry = 2 * ((keff_y + 3) / 4 * 4), Rx = 2 * ((keff_x + 3) / 4 * 4);
  if (const char* e = getenv("RPDE_FUSED_SPECTRAL")) if (e[0] == '0') return false;
  // keff_y == keff_x, not only equal padded counts: the h2 mode mix (k_mix_prep, k_mix_wgrad_fold) zeroes the weights and
  // gradients of modes >= keff with ONE keff for both axes, while the reference clamps each axis by its own length
  // (spectral_convolution.py:269-300) -- e.g. a [64, 32] grid with 20 modes keeps 17 along y and 20 along x
  return C == 64 && M % 32 == 0 && N % 32 == 0 && M <= 32 * ANA_MAXKS && N <= 32 * ANA_MAXKS && keff_y == keff_x && Ry == Rx &&
         Ry <= 48;
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

// RPDE_ANA_SQ=0: keep the two-read analysis kernel for every shape (A/B, tests).  Default: the one-pass kernel on every
// square grid (with a workgroup's rows matched to the grid it is level with the two-read kernel at 64^2 and 128^2 for
// B = 32 and ahead at B = 8: 2.03 vs 2.25 ms per training step at 64^2).
static bool ana_sq_ok(int M, int N, int cus) {
  const char* e = getenv("RPDE_ANA_SQ");
  if (e && e[0] == '0') return false;
  return M == N && cus >= 256;
}

// RPDE_ANA_RR=0: the row-block kernel k_dft_analysis_sq_h2 (cross-wave sums through LDS) instead of the round-robin one
static bool ana_rr_on() {
  const char* e = getenv("RPDE_ANA_RR");
  return !(e && e[0] == '0');
}

int fused2d_analysis(const float* x, float* spec_y, float* spec_x, float* amax_y, float* amax_x, const rpde_plan* py,
                     const rpde_plan* px, int adjoint, int B, int M, int N, hipStream_t st) {
  {
    int dev = 0, cus = 256;
    RPDE_HIP(hipGetDevice(&dev));
    RPDE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (ana_sq_ok(M, N, cus) && ana_rr_on() && py->h2_ana[adjoint] == px->h2_ana[adjoint]) {
      // the round-robin schedule: a group = N waves = N / 8 workgroups; as many groups as fit (never more than samples)
      AnaRrP Q;
      Q.x = x; Q.timg = (const char*)py->h2_ana_p[adjoint]; Q.spec_y = spec_y; Q.spec_x = spec_x; Q.amax_y = amax_y; Q.amax_x = amax_x;
      Q.B = B; Q.n = N; Q.nb = N / 32; Q.R = 2 * py->kp;
      const int wpg = N / ANA_WAVES;
      int ng = cus / wpg;
      if (ng < 1) ng = 1;
      Q.ng = B < ng ? B : ng;
      const dim3 grid(wpg * Q.ng), blk(64 * ANA_WAVES);
      const int MTq = (Q.R + 15) / 16;
      if (MTq == 1) hipLaunchKernelGGL(k_dft_analysis_rr_h2<1>, grid, blk, 0, st, Q);
      else if (MTq == 2) hipLaunchKernelGGL(k_dft_analysis_rr_h2<2>, grid, blk, 0, st, Q);
      else hipLaunchKernelGGL(k_dft_analysis_rr_h2<3>, grid, blk, 0, st, Q);
      RPDE_LAUNCH_CHECK();
      return RPDE_OK;
    }
    if (ana_sq_ok(M, N, cus) && py->h2_ana[adjoint] == px->h2_ana[adjoint]) {
      AnaSqP Q;
      Q.x = x; Q.timg = (const char*)py->h2_ana_p[adjoint]; Q.spec_y = spec_y; Q.spec_x = spec_x; Q.amax_y = amax_y; Q.amax_x = amax_x;
      Q.B = B; Q.n = N; Q.ks = N / 32; Q.R = 2 * py->kp;
      { const char* e = getenv("RPDE_ANA_B1"); Q.b1 = (e && e[0] == '1') ? 1 : 0; }
      Q.rpw = ANA_WAVES / Q.ks;                            // rows of a block per workgroup; 32 / rpw workgroups per group
      const int wpg = 32 / Q.rpw;
      Q.ng = B < 256 / wpg ? B : 256 / wpg;
      const dim3 grid(wpg * Q.ng), blk(64 * ANA_WAVES);
      const int MTq = (Q.R + 15) / 16;
      if (MTq == 1) hipLaunchKernelGGL(k_dft_analysis_sq_h2<1>, grid, blk, 0, st, Q);
      else if (MTq == 2) hipLaunchKernelGGL(k_dft_analysis_sq_h2<2>, grid, blk, 0, st, Q);
      else hipLaunchKernelGGL(k_dft_analysis_sq_h2<3>, grid, blk, 0, st, Q);
      RPDE_LAUNCH_CHECK();
      return RPDE_OK;
    }
  }
  AnaP P;
  memset(&P, 0, sizeof(P));
  P.x = x; P.naxes = 2; P.B = B; P.R = 2 * py->kp;
  // chunk of samples whose field (read twice) stays inside the 256 MB Infinity Cache beside everything else that
  // streams through it (measured at B = 32, 256^2: no chunking 0.73 ms per forward, 160 MB 0.69, 40-80 MB 0.66)
  const long sample_bytes = (long)M * N * 64 * 4;
  long ch = (64L << 20) / sample_bytes;
  P.chunk = (int)(ch < 1 ? 1 : (ch > B ? B : ch));
  AnaAxis& ay = P.ax[0];
  ay.timg = (const char*)py->h2_ana[adjoint]; ay.spec = spec_y; ay.amax = amax_y; ay.n = N; ay.ks = N / 32; ay.lps = M;
  ay.rps = (M + ANA_WAVES - 1) / ANA_WAVES; ay.zdiv = 1; ay.s1 = (long)N * 64; ay.s2 = 0; ay.ldk = 64;
  AnaAxis& ax = P.ax[1];
  ax.timg = (const char*)px->h2_ana[adjoint]; ax.spec = spec_x; ax.amax = amax_x; ax.n = M; ax.ks = M / 32; ax.lps = N;
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

// RPDE_SYN4=0: keep the 64 x 64 x 16-channel tile kernel where the 64 x 32 x 32-channel one would run (A/B, tests)
static bool syn4_on() {
  const char* e = getenv("RPDE_SYN4");
  return !(e && e[0] == '0');
}

// RPDE_SYN3=0: keep the 16 x 16 tile kernel for every shape (A/B, tests)
static bool syn3_ok(int M, int N) {
  if (const char* e = getenv("RPDE_SYN3")) if (e[0] == '0') return false;
  return M % 64 == 0 && N % 64 == 0;
}

int fused2d_synthesis(const void* imgy, const void* imgx, const float* invy, const float* invx, const rpde_plan* py,
                      const rpde_plan* px, int adjoint, float* out, const float* skip, int B, int M, int N, hipStream_t st) {
  SynP P;
  P.imgy = (const char*)imgy; P.imgx = (const char*)imgx; P.invy = invy; P.invx = invx;
  P.taby = (const char*)py->h2_syn[adjoint]; P.tabx = (const char*)px->h2_syn[adjoint];
  P.out = out; P.skip = skip; P.B = B; P.M = M; P.N = N;
  const int R3 = 2 * py->kp;
  const int NF3 = 2 * (R3 / 32) + (3 * ((R3 % 32) / 8) + 3) / 4;
  if (syn3_ok(M, N) && NF3 <= 3) {
    int dev = 0, cus = 256;
    RPDE_HIP(hipGetDevice(&dev));
    RPDE_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (!skip && syn4_on()) {
      // 64 x 32 x 32-channel tiles, whole 128-byte lines per store: (M / 64) (N / 32) 2 items per sample
      const int ng = B < 8 ? B : 8;
      P.sy = ng;
      P.sx = 1;
      const long wps = (long)(M / 64) * (N / 32) * 2;
      long per_group = cus / ng;
      const long most = (long)((B + ng - 1) / ng) * wps;
      if (per_group > most) per_group = most;
      if (per_group < 1) per_group = 1;
      const dim3 grid4((unsigned)(ng * per_group)), blk4(64 * SYN3_WAVES);
      switch (R3 / 8) {
        case 1: hipLaunchKernelGGL((k_dft_synthesis4_h2<0, 1>), grid4, blk4, 0, st, P); break;
        case 2: hipLaunchKernelGGL((k_dft_synthesis4_h2<0, 2>), grid4, blk4, 0, st, P); break;
        case 3: hipLaunchKernelGGL((k_dft_synthesis4_h2<0, 3>), grid4, blk4, 0, st, P); break;
        case 4: hipLaunchKernelGGL((k_dft_synthesis4_h2<1, 0>), grid4, blk4, 0, st, P); break;
        default: hipLaunchKernelGGL((k_dft_synthesis4_h2<1, 1>), grid4, blk4, 0, st, P); break;
      }
      RPDE_LAUNCH_CHECK();
      return RPDE_OK;
    }
    // persistent: one workgroup per CU, in 8 groups (one per XCD; fewer when the batch is smaller); never more per group
    // than the busiest group has tiles
    const int ng = B < 8 ? B : 8;
    P.sy = ng;
    P.sx = 1;
    const long wps = (long)(M / 64) * (N / 64) * 4;
    long per_group = cus / ng;
    const long most = (long)((B + ng - 1) / ng) * wps;
    if (per_group > most) per_group = most;
    if (per_group < 1) per_group = 1;
    const dim3 grid3((unsigned)(ng * per_group)), blk3(64 * SYN3_WAVES);
#define RPDE_SYN3_LAUNCH(K32_, TG_)                                                                                     \
    do {                                                                                                                \
      if (skip) hipLaunchKernelGGL((k_dft_synthesis3_h2<K32_, TG_, true>), grid3, blk3, 0, st, P);                      \
      else hipLaunchKernelGGL((k_dft_synthesis3_h2<K32_, TG_, false>), grid3, blk3, 0, st, P);                          \
    } while (0)
    switch (R3 / 8) {
      case 1: RPDE_SYN3_LAUNCH(0, 1); break;
      case 2: RPDE_SYN3_LAUNCH(0, 2); break;
      case 3: RPDE_SYN3_LAUNCH(0, 3); break;
      case 4: RPDE_SYN3_LAUNCH(1, 0); break;
      default: RPDE_SYN3_LAUNCH(1, 1); break;
    }
#undef RPDE_SYN3_LAUNCH
    RPDE_LAUNCH_CHECK();
    return RPDE_OK;
  }
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
