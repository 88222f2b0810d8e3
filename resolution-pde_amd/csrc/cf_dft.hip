// Truncated real DFT along the contiguous axis of channels-first tensors (SpectralConv1d / SpectralConv2d stage 1 and
// its inverse, models/spectral_convolution.py:38-55, 79-98; the spectral resizers), as two streaming kernels in h2
// arithmetic (h2.h):
//
//   analysis   spec[row][r] = sum_y x[row][y] T[r][y]      rows x n  ->  rows x R      (R = 2 kp <= 32)
//   synthesis  out[row][y]  = alpha sum_r spec[row][r] S[y][r]   rows x R  ->  rows x n
//
// Here the FIELD is the A operand (16 lines per MFMA tile, the reduction index contiguous in memory) and the small
// table is the B operand, resident in LDS as ready fragments for the whole (persistent) workgroup.  A wave owns 16
// lines: analysis streams them in 128-point chunks (scale from the chunk's maximum, split, 4 x NT x 3 MFMAs, fp32
// accumulation of the chunk results); synthesis forms 16 x 128 output blocks, turns them through 8 KB of LDS and
// stores whole 512-byte row pieces.  The field is read once / written once; everything else is ~5 % of it.
// Inside a 32-deep reduction step lane group g holds entries {4g..4g+3} and {16+4g..16+4g+3} (ff_perm order, the
// table fragments are built to match), so that a load instruction covers 64 contiguous bytes per line.
#include "cf_dft.h"
#include "h2.h"

namespace rpde {

constexpr int CF_WAVES = 8;
constexpr int CF_CHUNK = 4;            // reduction steps (of 32) per analysis chunk

__device__ __forceinline__ int cf_perm(int g, int j) { return 16 * (j >> 2) + 4 * g + (j & 3); }

// B fragments of a table for the analysis-type product: entry (k = field index y, col = r) = src[r*rs + y*cs];
// layout [ks][nt][hi|lo][1 KB], scaled by 2^12
__global__ __launch_bounds__(64) void k_cf_table_ana(const float* __restrict__ src, long rs, long cs, int R, int n, int NT,
                                                     char* __restrict__ out) {
  const int f = blockIdx.x, l = threadIdx.x, g = l >> 4, li = l & 15;
  const int ks = f / NT, nt = f % NT;
  const int r = 16 * nt + li;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int y = 32 * ks + cf_perm(g, j);
    v[j] = (r < R && y < n) ? src[r * rs + y * cs] * (float)(1 << H2_TABLE_EXP) : 0.f;
  }
  uint2 h0, l0, h1, l1;
  h2_split4(v[0], v[1], v[2], v[3], h0, l0);
  h2_split4(v[4], v[5], v[6], v[7], h1, l1);
  char* p = out + (long)f * 2048 + l * 16;
  *reinterpret_cast<uint4*>(p) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4*>(p + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
}
// B fragments for the synthesis-type product: entry (k = r, col = y) = src[y*rs + r*cs]; layout [yt][hi|lo][1 KB]
__global__ __launch_bounds__(64) void k_cf_table_syn(const float* __restrict__ src, long rs, long cs, int R, int n,
                                                     char* __restrict__ out) {
  const int yt = blockIdx.x, l = threadIdx.x, g = l >> 4, li = l & 15;
  const int y = 16 * yt + li;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = cf_perm(g, j);
    v[j] = (r < R && y < n) ? src[y * rs + r * cs] * (float)(1 << H2_TABLE_EXP) : 0.f;
  }
  uint2 h0, l0, h1, l1;
  h2_split4(v[0], v[1], v[2], v[3], h0, l0);
  h2_split4(v[4], v[5], v[6], v[7], h1, l1);
  char* p = out + (long)yt * 2048 + l * 16;
  *reinterpret_cast<uint4*>(p) = make_uint4(h0.x, h0.y, h1.x, h1.y);
  *reinterpret_cast<uint4*>(p + 1024) = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

bool cf_h2_eligible(int n, int R) {
  if (const char* e = getenv("RPDE_FUSED_CF")) if (e[0] == '0') return false;
  const int NT = (R + 15) / 16;
  return n % 128 == 0 && R <= 32 && (n / 32) * NT * 2048 <= 65536;
}

// the synthesis kernel keeps n / 16 table fragments of 2 KB in its 64 KB table area: n <= 512.  (cf_h2_eligible alone let
// n = 1024 with R <= 16 through -- its analysis table fits -- and the synthesis kernel then overran its LDS: NaNs.  Found
// by tests/test_gpu_kernels.py::test_fused_evaluation_fnoblock2d_equals_the_two_step_path.)
bool cf_h2_syn_eligible(int n, int R) { return cf_h2_eligible(n, R) && (n / 16) * 2048 <= 65536; }

int cf_build_tables(rpde_plan* p, hipStream_t st) {
  const int R = 2 * p->kp, n = p->n, NT = (R + 15) / 16;
  const size_t ab = (size_t)(n / 32) * NT * 2048, sb = (size_t)(n / 16) * 2048;
  for (int i = 0; i < 2; ++i) {
    RPDE_HIP(hipMalloc(&p->cf_ana[i], ab));
    RPDE_HIP(hipMalloc(&p->cf_syn[i], sb));
  }
  // [0]: forward operands (Fa for analysis, Fs for synthesis); [1]: adjoint operands (Fs^T, Fa^T)
  hipLaunchKernelGGL(k_cf_table_ana, dim3((n / 32) * NT), dim3(64), 0, st, p->fa, (long)p->ldn, 1L, R, n, NT, (char*)p->cf_ana[0]);
  hipLaunchKernelGGL(k_cf_table_ana, dim3((n / 32) * NT), dim3(64), 0, st, p->fs, 1L, (long)R, R, n, NT, (char*)p->cf_ana[1]);
  hipLaunchKernelGGL(k_cf_table_syn, dim3(n / 16), dim3(64), 0, st, p->fs, (long)R, 1L, R, n, (char*)p->cf_syn[0]);
  hipLaunchKernelGGL(k_cf_table_syn, dim3(n / 16), dim3(64), 0, st, p->fa, 1L, (long)p->ldn, R, n, (char*)p->cf_syn[1]);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

struct CfP {
  const float* in; float* out; const char* timg;
  long rows; int n, R, ldo;          // ldo: leading dimension of the spectrum (analysis: out, synthesis: in)
  float alpha;
};

// ---- analysis: rows x n -> rows x R ----
template <int NT>
__global__ __launch_bounds__(64 * CF_WAVES, 2) void k_cf_analysis_h2(const CfP P) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, li = l & 15;
  const int KS = P.n / 32;
  {
    const uint4* src = reinterpret_cast<const uint4*>(P.timg);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = tid; i < KS * NT * 128; i += 64 * CF_WAVES) dst[i] = src[i];
  }
  __syncthreads();
  const long ntiles = (P.rows + 15) / 16;
  const int nchunk = KS / CF_CHUNK;
  for (long tile = (long)blockIdx.x * CF_WAVES + w; tile < ntiles; tile += (long)gridDim.x * CF_WAVES) {
    const long row = min(tile * 16 + li, P.rows - 1);
    const float* __restrict__ base = P.in + row * P.n + 4 * g;
    f32x4v tot[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) tot[nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    float4 nxt[2 * CF_CHUNK];
#pragma unroll
    for (int i = 0; i < CF_CHUNK; ++i) {
      nxt[2 * i] = *reinterpret_cast<const float4*>(base + 32 * i);
      nxt[2 * i + 1] = *reinterpret_cast<const float4*>(base + 32 * i + 16);
    }
    for (int c = 0; c < nchunk; ++c) {
      float4 cur[2 * CF_CHUNK];
#pragma unroll
      for (int i = 0; i < 2 * CF_CHUNK; ++i) cur[i] = nxt[i];
      if (c + 1 < nchunk) {
        const float* __restrict__ nb = base + (long)(c + 1) * 32 * CF_CHUNK;
#pragma unroll
        for (int i = 0; i < CF_CHUNK; ++i) {
          nxt[2 * i] = *reinterpret_cast<const float4*>(nb + 32 * i);
          nxt[2 * i + 1] = *reinterpret_cast<const float4*>(nb + 32 * i + 16);
        }
      }
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * CF_CHUNK; ++i) {
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(cur[i].x), "v"(cur[i].y));
        asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(cur[i].z), "v"(cur[i].w));
      }
      m = wave_max(m);
      float sc, inv;
      h2_scale(m, H2_TABLE_EXP, sc, inv);
      f32x4v acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < CF_CHUNK; ++i) {
        uint2 h0, l0, h1, l1;
        h2_split4(cur[2 * i].x * sc, cur[2 * i].y * sc, cur[2 * i].z * sc, cur[2 * i].w * sc, h0, l0);
        h2_split4(cur[2 * i + 1].x * sc, cur[2 * i + 1].y * sc, cur[2 * i + 1].z * sc, cur[2 * i + 1].w * sc, h1, l1);
        union { uint4 u; f16x8 v; } ah, al;
        ah.u = make_uint4(h0.x, h0.y, h1.x, h1.y);
        al.u = make_uint4(l0.x, l0.y, l1.x, l1.y);
        const int ks = c * CF_CHUNK + i;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const char* t = smem + (ks * NT + nt) * 2048 + l * 16;
          acc[nt] = h2_mfma32(ah.v, al.v, *reinterpret_cast<const f16x8*>(t), *reinterpret_cast<const f16x8*>(t + 1024), acc[nt]);
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) tot[nt][r] = fmaf(acc[nt][r], inv, tot[nt][r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long orow = tile * 16 + 4 * g + r;
      if (orow < P.rows) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          if (16 * nt + li < P.R) P.out[orow * P.ldo + 16 * nt + li] = tot[nt][r] * P.alpha;
      }
    }
  }
}

// ---- synthesis: rows x R -> rows x n ----
constexpr int CFS_LD = 132;            // floats per staged row: 128 + 4 (the two lane groups of a write hit different banks)
__global__ __launch_bounds__(64 * CF_WAVES, 2) void k_cf_synthesis_h2(const CfP P) {
  __shared__ __attribute__((aligned(16))) char smem[65536 + CF_WAVES * 16 * CFS_LD * 4];
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, g = l >> 4, li = l & 15;
  const int YT = P.n / 16;
  {
    const uint4* src = reinterpret_cast<const uint4*>(P.timg);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    for (int i = tid; i < YT * 128; i += 64 * CF_WAVES) dst[i] = src[i];
  }
  __syncthreads();
  float* const stg = reinterpret_cast<float*>(smem + 65536) + w * 16 * CFS_LD;
  const long ntiles = (P.rows + 15) / 16;
  for (long tile = (long)blockIdx.x * CF_WAVES + w; tile < ntiles; tile += (long)gridDim.x * CF_WAVES) {
    const long row = min(tile * 16 + li, P.rows - 1);
    const float* __restrict__ sp = P.in + row * P.ldo;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const int r = cf_perm(g, j); v[j] = r < P.R ? sp[r] : 0.f; }
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
    m = wave_max(m);
    float sc, inv;
    h2_scale(m, H2_TABLE_EXP, sc, inv);
    inv *= P.alpha;
    uint2 h0, l0, h1, l1;
    h2_split4(v[0] * sc, v[1] * sc, v[2] * sc, v[3] * sc, h0, l0);
    h2_split4(v[4] * sc, v[5] * sc, v[6] * sc, v[7] * sc, h1, l1);
    union { uint4 u; f16x8 v; } ah, al;
    ah.u = make_uint4(h0.x, h0.y, h1.x, h1.y);
    al.u = make_uint4(l0.x, l0.y, l1.x, l1.y);
    for (int blk = 0; blk < YT / 8; ++blk) {                 // 128 output points at a time
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const char* t = smem + (blk * 8 + q) * 2048 + l * 16;
        const f32x4v c = h2_mfma32(ah.v, al.v, *reinterpret_cast<const f16x8*>(t), *reinterpret_cast<const f16x8*>(t + 1024),
                                   (f32x4v){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(4 * g + r) * CFS_LD + 16 * q + li] = c[r] * inv;
      }
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int f = i * 64 + l, rr = f >> 5, c4 = (f & 31) << 2;
        const long orow = tile * 16 + rr;
        const float4 o = *reinterpret_cast<const float4*>(stg + rr * CFS_LD + c4);
        if (orow < P.rows) *reinterpret_cast<float4*>(P.out + orow * P.n + blk * 128 + c4) = o;
      }
      wave_lds_fence();
    }
  }
}

static int cf_grid(long rows) {
  int dev = 0, cus = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long need = ((rows + 15) / 16 + CF_WAVES - 1) / CF_WAVES;
  return (int)(need < cus ? need : cus);
}

int cf_analysis_h2(const rpde_plan* pl, int adjoint, const float* x, float* spec, long rows, float alpha, hipStream_t st) {
  CfP P;
  P.in = x; P.out = spec; P.timg = (const char*)pl->cf_ana[adjoint]; P.rows = rows; P.n = pl->n; P.R = 2 * pl->kp;
  P.ldo = 2 * pl->kp; P.alpha = alpha;
  const dim3 grid(cf_grid(rows)), blk(64 * CF_WAVES);
  if (P.R <= 16) hipLaunchKernelGGL(k_cf_analysis_h2<1>, grid, blk, 0, st, P);
  else hipLaunchKernelGGL(k_cf_analysis_h2<2>, grid, blk, 0, st, P);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

int cf_synthesis_h2(const rpde_plan* pl, int adjoint, const float* spec, float* out, long rows, float alpha, hipStream_t st) {
  CfP P;
  P.in = spec; P.out = out; P.timg = (const char*)pl->cf_syn[adjoint]; P.rows = rows; P.n = pl->n; P.R = 2 * pl->kp;
  P.ldo = 2 * pl->kp; P.alpha = alpha;
  hipLaunchKernelGGL(k_cf_synthesis_h2, dim3(cf_grid(rows)), dim3(64 * CF_WAVES), 0, st, P);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // namespace rpde
