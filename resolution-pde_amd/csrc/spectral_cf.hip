// SpectralConv1d / SpectralConv2d (models/spectral_convolution.py:24-98) on
// channels-first tensors, norm='backward'.
//
//   1-D:  spec[b,i][re|im][k]   = act(x)[b,i,:] . Fa^T            (NT GEMM, rows = B*Cin)
//         mix over i per (b,o,k)                                   (VALU, 8*B*Cin*Cout*K flop)
//         out[b,o,:]            = ospec[b,o][re|im][k] . Fs^T
//   2-D:  stage 1 along N as above with rows = B*Cin*M (planar re|im),
//         stage 2 along M: row-restricted complex DFT to the R = 2*m1 retained
//         rows as a real [2R,2M] block matrix per (b,i), mix with weights1
//         (slots < m1) / weights2, inverse stage 2 ([2M,2R], columns of
//         overwritten slots zeroed: quirk Q6), inverse stage 1.
//
// irfft2 = complex inverse over M then C2R over N, so Im(DC)/Im(Nyquist) along N
// are ignored by the synthesis table exactly as torch does (quirk Q7).
#include "rpde_internal.h"
#include "plan.h"
#include "conv_small.h"
#include "pointwise.h"
#include "cf_dft.h"
#include "h2.h"

namespace rpde {

// complex position (r, ky) of a planar block [R][2][kp]: re at (2r)*kp+ky, im at (2r+1)*kp+ky
struct MixGeom { int B, Ci, Co, R, m1, m2, kp; };

__device__ __forceinline__ const float* wsel(const float* w1, const float* w2, const MixGeom& g, int i, int o, int r, int ky) {
  const float* w = r < g.m1 ? w1 : w2;
  const int rr = r < g.m1 ? r : r - g.m1;
  return w + ((((long)i * g.Co + o) * g.m1 + rr) * g.m2 + ky) * 2;
}

// ---- per-mode complex channel mixing (compl_mul1d / complex_mul2d of the reference, spectral_convolution.py:34-36,
// ---- 75-77) and its two adjoints, one template:
//   MODE 0  out[b,o] = sum_i s[b,i] W[i,o]            (reduction over i, 'a' = input spectra  [B,Ci], 'c' -> [B,Co])
//   MODE 1  ds[b,i]  = sum_o g[b,o] conj(W[i,o])      (reduction over o, 'a' = output grads   [B,Co], 'c' -> [B,Ci])
//   MODE 2  gW[i,o]  = sum_b conj(s[b,i]) g[b,o]      (reduction over b)
// per retained mode (r, ky).  A wave owns one (row r, kept channel) pair: lanes 0..15 of each 16-lane group are 16
// consecutive ky (the contiguous index of both the spectra and the weights: 64 / 128-byte coalesced segments), the
// four lane groups split the reduction index four ways and are combined with two lane-swap additions (v_permlane16_swap, v_permlane32_swap) at the end.  The
// eight waves of a workgroup share one row r of the spectra, staged in LDS in chunks of the reduction index, so the
// spectra are read from memory once per workgroup and every weight exactly once per launch.
constexpr int CMIX_WAVES = 8;
constexpr int CMIX_LDS_FLOATS = 8192;          // 32 KB of staged spectra per chunk

template <int MODE, int MAXB>
__global__ __launch_bounds__(64 * CMIX_WAVES) void k_cmix(const float* __restrict__ a, const float* __restrict__ a2,
                                                            const float* __restrict__ w1, const float* __restrict__ w2,
                                                            float* __restrict__ c, float* __restrict__ c2, MixGeom g) {
  __shared__ float stage[CMIX_LDS_FLOATS];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, grp = l >> 4, kl = l & 15;
  const int nwaves = blockDim.x >> 6;               // 8, or 4 when there are few (row, channel, slab) triples
  const int r = blockIdx.y;
  const long blk = 2L * g.R * g.kp;                     // floats per (b, channel) spectrum block
  if (MODE != 2) {
    // batch slab blockIdx.z: MAXB samples whose accumulators a lane keeps in registers
    const int b0 = blockIdx.z * MAXB;
    a += (long)b0 * (MODE == 0 ? g.Ci : g.Co) * blk;
    c += (long)b0 * (MODE == 0 ? g.Co : g.Ci) * blk;
    g.B = min(MAXB, g.B - b0);
  }
  // kept index of this wave (the one that is neither reduced nor ky): o (MODE 0), i (MODE 1), (i,o) pair (MODE 2)
  const int nkept = MODE == 0 ? g.Co : (MODE == 1 ? g.Ci : g.Ci * g.Co);
  const int nred = MODE == 0 ? g.Ci : (MODE == 1 ? g.Co : g.B);
  const int kept = blockIdx.x * nwaves + wv;
  const bool live = kept < nkept;
  for (int ky0 = 0; ky0 < g.m2; ky0 += 16) {
    const int ky = ky0 + kl;
    const bool kyok = ky < g.m2;
    float accr[MODE == 2 ? 1 : MAXB], acci[MODE == 2 ? 1 : MAXB];
#pragma unroll
    for (int b = 0; b < (MODE == 2 ? 1 : MAXB); ++b) { accr[b] = 0.f; acci[b] = 0.f; }
    if (MODE == 2) {
      // no staging: each wave needs one (i, o) pair of columns only
      if (live && kyok) {
        const int i = kept / g.Co, o = kept % g.Co;
        for (int b = grp; b < g.B; b += 4) {
          const float* sp = a + ((long)b * g.Ci + i) * blk + (2L * r) * g.kp + ky;
          const float* gp = a2 + ((long)b * g.Co + o) * blk + (2L * r) * g.kp + ky;
          const float xr = sp[0], xi = sp[g.kp], gr = gp[0], gi = gp[g.kp];
          accr[0] += xr * gr + xi * gi;
          acci[0] += xr * gi - xi * gr;
        }
      }
    } else {
      // chunks of the reduction channel index: stage a[b][ch][row r][re|im][16 ky] for all b
      const int per_ch = g.B * 32;                                   // floats per reduction channel in the stage
      const int chunk = max(4, min(nred, (CMIX_LDS_FLOATS / per_ch) & ~3));
      for (int c0 = 0; c0 < nred; c0 += chunk) {
        const int nc = min(chunk, nred - c0);
        __syncthreads();
        for (int e = tid; e < nc * per_ch; e += blockDim.x) {
          const int kk = e & 15, ri = (e >> 4) & 1, b = (e >> 5) % g.B, ch = e / per_ch;
          const int kyy = ky0 + kk;
          stage[e] = kyy < g.m2 ? a[((long)b * nred + c0 + ch) * blk + (2L * r + ri) * g.kp + kyy] : 0.f;
        }
        __syncthreads();
        if (live && kyok) {
          for (int ch = grp; ch < nc; ch += 4) {
            const int red = c0 + ch;
            const int i = MODE == 0 ? red : kept, o = MODE == 0 ? kept : red;
            const float* w = wsel(w1, w2, g, i, o, r, ky);
            const float wr = w[0], wi = MODE == 0 ? w[1] : -w[1];     // adjoint: conj(W)
            const float* sp = stage + ch * per_ch + kl;
#pragma unroll
            for (int b = 0; b < MAXB; ++b) {
              if (b < g.B) {
                const float xr = sp[b * 32], xi = sp[b * 32 + 16];
                accr[b] += xr * wr - xi * wi;
                acci[b] += xr * wi + xi * wr;
              }
            }
          }
        }
      }
    }
    // combine the four lane groups (reduction index mod 4)
#pragma unroll
    for (int b = 0; b < (MODE == 2 ? 1 : MAXB); ++b) {
      accr[b] += lane_xor16(accr[b]); accr[b] += lane_xor32(accr[b]);
      acci[b] += lane_xor16(acci[b]); acci[b] += lane_xor32(acci[b]);
    }
    if (live && kyok && grp == 0) {
      if (MODE == 2) {
        const int i = kept / g.Co, o = kept % g.Co;
        float* gw = r < g.m1 ? c : c2;
        const int rr = r < g.m1 ? r : r - g.m1;
        float* p = gw + ((((long)i * g.Co + o) * g.m1 + rr) * g.m2 + ky) * 2;
        p[0] = accr[0];
        p[1] = acci[0];
      } else {
#pragma unroll
        for (int b = 0; b < MAXB; ++b) {
          if (b < g.B) {
            float* op = c + ((long)b * nkept + kept) * blk + (2L * r) * g.kp + ky;
            op[0] = accr[b];
            op[g.kp] = acci[b];
          }
        }
      }
    }
  }
}

template <int MODE, int MAXB>
static void launch_cmix(const float* a, const float* w1, const float* w2, float* out, const MixGeom& g, int nout, int nw,
                        hipStream_t st) {
  const dim3 grid((nout + nw - 1) / nw, g.R, (g.B + MAXB - 1) / MAXB);
  hipLaunchKernelGGL((k_cmix<MODE, MAXB>), grid, dim3(64 * nw), 0, st, a, nullptr, w1, w2, out, nullptr, g);
}

static int launch_mix(int which, const float* a, const float* b, const float* w1, const float* w2, float* out, float* out2,
                      const MixGeom& g0, hipStream_t st) {
  if (which == 2) {
    const dim3 grid((g0.Ci * g0.Co + CMIX_WAVES - 1) / CMIX_WAVES, g0.R);
    hipLaunchKernelGGL((k_cmix<2, 1>), grid, dim3(64 * CMIX_WAVES), 0, st, a, b, w1, w2, out, out2, g0);
    RPDE_LAUNCH_CHECK();
    return RPDE_OK;
  }
  // MODE 0 / 1 keep one batch slab of accumulators in registers; slabs run along grid.z.  The slab is 16 samples
  // unless that leaves most of the chip idle (the 1-D layers: R = 1, a handful of channel groups): then 4 or 2, and
  // as a last step half-size workgroups -- every workgroup stages only its own slab, so smaller slabs cost no
  // extra spectrum traffic, only more reads of the (L2-resident) weights.
  const int nout = which == 0 ? g0.Co : g0.Ci;
  auto blocks = [&](int maxb, int nw) { return (long)((nout + nw - 1) / nw) * g0.R * ((g0.B + maxb - 1) / maxb); };
  int maxb = 16, nw = CMIX_WAVES;
  if (blocks(16, nw) < 64) maxb = blocks(4, nw) >= 64 ? 4 : 2;
  if (blocks(maxb, nw) < 64) nw = 4;
#define RPDE_CMIX(MB) (which == 0 ? launch_cmix<0, MB>(a, w1, w2, out, g0, nout, nw, st) : launch_cmix<1, MB>(a, w1, w2, out, g0, nout, nw, st))
  if (maxb == 16) RPDE_CMIX(16);
  else if (maxb == 4) RPDE_CMIX(4);
  else RPDE_CMIX(2);
#undef RPDE_CMIX
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// Thin products -- a long reduction onto few output tiles (the 1-D layers at small batch: [1024 x 1024] . [1024 x 32]
// is 8 workgroups) -- split the reduction over the grid into slabs and fold them in fixed order; `slabs` is the
// caller's scratch of thin_slab_floats() entries (nullptr: never split).
static int thin_ksplit(long rows, int ncols, int kred) {
  if (ncols > 32 || kred < 256) return 1;
  const long tiles = (rows + 127) / 128;
  if (tiles >= 64) return 1;
  int ks = 1;
  while (ks < 16 && tiles * ks < 64 && kred / (2 * ks) >= 64) ks *= 2;
  return ks;
}
static size_t thin_slab_floats(long rows, int ncols, int kred) {
  const int ks = thin_ksplit(rows, ncols, kred);
  return ks > 1 ? (size_t)ks * rows * ncols : 0;
}
static int thin_gemm(rpde_gemm_desc& d, float* slabs, hipStream_t st) {
  const int ks = slabs ? thin_ksplit(d.M, d.N, d.K) : 1;
  if (ks == 1 || d.ldc != d.N) return launch_gemm(d, st);
  float* out = d.C;
  const float alpha = d.alpha;
  d.C = slabs; d.ksplit = ks; d.sCk = (long)d.M * d.N; d.alpha = 1.f;
  RPDE_TRY(launch_gemm(d, st));
  return reduce_slabs(slabs, out, (long)d.M * d.N, ks, (long)d.M * d.N, alpha, 0, st);
}

// rows x n  ->  rows x 2kp   (forward real DFT along the contiguous axis), optional act on the input
static int cf_analysis(const rpde_plan* pl, const float* x, float* spec, long rows, int n, int act_in, hipStream_t st,
                       float* slabs = nullptr) {
  if (!act_in && pl->cf_ana[0] && rows >= 16 && cf_h2_eligible(n, 2 * pl->kp)) return cf_analysis_h2(pl, 0, x, spec, rows, 1.f, st);
  rpde_gemm_desc d = gemm_desc();
  d.A = x; d.a_kmajor = 1; d.lda = n; d.act_a = act_in;
  d.B = pl->fa; d.b_kmajor = 1; d.ldb = pl->ldn;
  d.C = spec; d.ldc = 2L * pl->kp;
  d.M = (int)rows; d.N = 2 * pl->kp; d.K = n;
  return thin_gemm(d, slabs, st);
}
// rows x 2kp -> rows x n  (C2R synthesis)
static int cf_synthesis(const rpde_plan* pl, const float* spec, float* out, long rows, int n, hipStream_t st,
                        float alpha = 1.f) {
  if (pl->cf_syn[0] && rows >= 16 && cf_h2_syn_eligible(n, 2 * pl->kp)) return cf_synthesis_h2(pl, 0, spec, out, rows, alpha, st);
  rpde_gemm_desc d = gemm_desc();
  d.alpha = alpha;
  d.A = spec; d.a_kmajor = 1; d.lda = 2L * pl->kp;
  d.B = pl->fs; d.b_kmajor = 1; d.ldb = 2L * pl->kp;
  d.C = out; d.ldc = n;
  d.M = (int)rows; d.N = n; d.K = 2 * pl->kp;
  return launch_gemm(d, st);
}
// adjoint of synthesis: g[rows,n] . Fs -> [rows, 2kp]
static int cf_synthesis_T(const rpde_plan* pl, const float* g, float* gspec, long rows, int n, hipStream_t st,
                          float* slabs = nullptr) {
  if (pl->cf_ana[1] && rows >= 16 && cf_h2_eligible(n, 2 * pl->kp)) return cf_analysis_h2(pl, 1, g, gspec, rows, 1.f, st);
  rpde_gemm_desc d = gemm_desc();
  d.A = g; d.a_kmajor = 1; d.lda = n;
  d.B = pl->fs; d.b_kmajor = 0; d.ldb = 2L * pl->kp;
  d.C = gspec; d.ldc = 2L * pl->kp;
  d.M = (int)rows; d.N = 2 * pl->kp; d.K = n;
  return thin_gemm(d, slabs, st);
}
// adjoint of analysis: dspec[rows,2kp] . Fa -> gx[rows,n], through act'(x) when act_in
static int cf_analysis_T(const rpde_plan* pl, const float* dspec, float* gx, long rows, int n, int act_in, const float* x,
                         hipStream_t st) {
  if (!act_in && pl->cf_syn[1] && rows >= 16 && cf_h2_syn_eligible(n, 2 * pl->kp)) return cf_synthesis_h2(pl, 1, dspec, gx, rows, 1.f, st);
  rpde_gemm_desc d = gemm_desc();
  d.A = dspec; d.a_kmajor = 1; d.lda = 2L * pl->kp;
  d.B = pl->fa; d.b_kmajor = 0; d.ldb = pl->ldn;
  d.C = gx; d.ldc = n;
  d.M = (int)rows; d.N = n; d.K = 2 * pl->kp;
  if (act_in) { d.epi_dact = act_in; d.aux = x; d.ldaux = n; }
  return launch_gemm(d, st);
}

// per (b,c) block GEMM with a shared [rowsA, colsA] table: out_z = T . in_z  (or T^T . in_z)
static int cf_rowdft(const float* table, long ld_table, bool transpose, int m_out, int k_red, const float* in, float* out,
                     int nblocks, int width, hipStream_t st) {
  rpde_gemm_desc d = gemm_desc();
  d.A = table; d.a_kmajor = transpose ? 0 : 1; d.lda = ld_table;
  d.B = in; d.b_kmajor = 0; d.ldb = width;
  d.C = out; d.ldc = width;
  d.M = m_out; d.N = width; d.K = k_red;
  d.batch = nblocks; d.sB1 = (long)k_red * width; d.sC1 = (long)m_out * width;
  return launch_gemm(d, st);
}

// ---- column stage of SpectralConv2d in two launches (evaluation paths) ------------------------------------------
// Between the two passes over the field the layer touches little data -- at 512^2, width 32, 12 x 12 modes: 1.5 MB of row
// spectra per sample in, 74 KB of retained modes, 1.5 MB out -- but as generic GEMMs with a 12-wide free dimension the
// three steps (column DFT, mode mix, inverse column DFT) took 62 + 28 + 39 us per block at B = 16: latency, not work.
// Plain fp32 multiply-adds, shaped to the data instead:
//   k_col_analysis       one workgroup per (b, c): the [2M, kp] block of row spectra is staged in LDS once; a wave owns
//                        rows rho, rho + 4, .. of the [2R, 2M] table, lanes run along the reduction index (coalesced table
//                        reads), kp accumulators per lane, one wave-wide sum per row.
//   k_col_mix_synthesis  one workgroup per (b, o): the mode mix of compl_mul2d (reference spectral_convolution.py:75-77)
//                        for this output channel into LDS (2R x kp values), then a thread per output row (m, re | im):
//                        its 2R table entries (contiguous) against the mixed block (LDS broadcasts), kp contiguous results.
constexpr int COL_THREADS = 256;
constexpr int COL_MIX_THREADS = 320;
constexpr int COL_ANA_THREADS = 768;      // up to 12 waves: one group of four table rows each (2R <= 48), so that a
                                          // workgroup's time is one sweep over the staged block, not three

// LIFT (first block of an FNO2d in evaluation, rpde_fno2d_lift_block_eval_fwd): the block's input is the lifting
// convolution of a one-channel field u and the grid coordinates, x0[c] = wl[c][0] u + wl[c][1] gx[m] + wl[c][2] gy[n] + bl[c].
// The row DFT is linear, so the row spectra of channel c are
//   wl[c][0] S_u[b][m] + (wl[c][1] gx[m] + bl[c]) S_1 + wl[c][2] S_gy
// with S_u the row spectra of u (B M rows instead of B C M), S_1 / S_gy those of the constant row and of gy: they are
// formed here while the block is staged and never exist in memory.
struct ColLift {
  const float* su;     // [B][M][2 kp]
  const float* sc;     // [2][2 kp]: row spectra of ones(N) and of gy
  const float* wl;     // [C][3]
  const float* bl;     // [C] or null
  const float* gx;     // [M]
  int C;
};

template <int KP, bool LIFT>
__global__ __launch_bounds__(COL_ANA_THREADS) void k_col_analysis(const float* __restrict__ fa, const float* __restrict__ s1,
                                                              float* __restrict__ s2, int M2, int R2, ColLift L) {
  extern __shared__ __attribute__((aligned(16))) float col_sm[];
  const int tid = threadIdx.x, l = tid & 63, wv = tid >> 6, nth = blockDim.x;
  const long bc = blockIdx.x;
  if (LIFT) {
    const int b = (int)(bc / L.C), c = (int)(bc % L.C);
    const float wu = L.wl[c * 3], wx = L.wl[c * 3 + 1], wy = L.wl[c * 3 + 2], bb = L.bl ? L.bl[c] : 0.f;
    const float4* src = reinterpret_cast<const float4*>(L.su + (long)b * M2 * KP);      // [M][2 KP] = [2M][KP]
    float4* dst = reinterpret_cast<float4*>(col_sm);
    for (int e = tid; e < M2 * KP / 4; e += nth) {
      const int mu = (4 * e) / KP, col = (mu & 1) * KP + (4 * e) % KP;     // (KP % 4 == 0: a float4 stays inside a row)
      const float4 one = *reinterpret_cast<const float4*>(L.sc + col), gy = *reinterpret_cast<const float4*>(L.sc + 2 * KP + col);
      const float a = fmaf(wx, L.gx[mu >> 1], bb);
      const float4 u = src[e];
      dst[e] = make_float4(fmaf(wu, u.x, fmaf(wy, gy.x, a * one.x)), fmaf(wu, u.y, fmaf(wy, gy.y, a * one.y)),
                           fmaf(wu, u.z, fmaf(wy, gy.z, a * one.z)), fmaf(wu, u.w, fmaf(wy, gy.w, a * one.w)));
    }
  } else {
    const float4* src = reinterpret_cast<const float4*>(s1 + bc * (long)M2 * KP);
    float4* dst = reinterpret_cast<float4*>(col_sm);
    for (int e = tid; e < M2 * KP / 4; e += nth) dst[e] = src[e];
  }
  __syncthreads();
  // four table rows at a time: a row of the staged block (kp values from LDS) then feeds 4 kp multiply-adds -- with one
  // row at a time the kernel was bound by the LDS return path (79 us)
  for (int rho0 = 4 * wv; rho0 < R2; rho0 += 4 * (nth >> 6)) {
    float acc[4][KP];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < KP; ++k) acc[j][k] = 0.f;
    const float* __restrict__ frow = fa + (long)rho0 * M2;
    // (R2 = 4 m1: always whole groups of four rows)
#pragma unroll 2
    for (int mu = l; mu < M2; mu += 64) {
      float f[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = frow[(long)j * M2 + mu];
      const float4* row = reinterpret_cast<const float4*>(col_sm + mu * KP);
#pragma unroll
      for (int q = 0; q < KP / 4; ++q) {
        const float4 v = row[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j][4 * q] = fmaf(f[j], v.x, acc[j][4 * q]); acc[j][4 * q + 1] = fmaf(f[j], v.y, acc[j][4 * q + 1]);
          acc[j][4 * q + 2] = fmaf(f[j], v.z, acc[j][4 * q + 2]); acc[j][4 * q + 3] = fmaf(f[j], v.w, acc[j][4 * q + 3]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float mine = 0.f;
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const float t = wave_sum_dpp(acc[j][k]);
        if (l == k) mine = t;
      }
      if (l < KP) s2[(bc * R2 + rho0 + j) * KP + l] = mine;
    }
  }
}

template <int KP>
__global__ __launch_bounds__(COL_MIX_THREADS) void k_col_mix_synthesis(const float* __restrict__ s2, const float* __restrict__ w1,
                                                                   const float* __restrict__ w2, const float* __restrict__ fs_t,
                                                                   float* __restrict__ t1, MixGeom g, int M2) {
  extern __shared__ __attribute__((aligned(16))) float col_sm[];       // mixed block [2R][KP]
  const int tid = threadIdx.x, nth = blockDim.x;      // (320 threads when R kp <= 320: the mix in one pass)
  const long bo = blockIdx.x;
  const int b = (int)(bo / g.Co), o = (int)(bo % g.Co), R2 = 2 * g.R;
  const long blk = (long)R2 * KP;
  for (int e = tid; e < g.R * KP; e += nth) {
    const int r = e / KP, ky = e % KP;
    float ar = 0.f, ai = 0.f;
    if (ky < g.m2) {
      const float* sp = s2 + (long)b * g.Ci * blk + (2L * r) * KP + ky;
#pragma unroll 16
      for (int i = 0; i < g.Ci; ++i) {
        const float* w = wsel(w1, w2, g, i, o, r, ky);
        const float wr = w[0], wi = w[1], xr = sp[i * blk], xi = sp[i * blk + KP];
        ar += xr * wr - xi * wi;
        ai += xr * wi + xi * wr;
      }
    }
    col_sm[(2 * r) * KP + ky] = ar;
    col_sm[(2 * r + 1) * KP + ky] = ai;
  }
  __syncthreads();
  // four CONSECUTIVE output rows per thread: a row of the mixed block (LDS broadcast) feeds 4 kp multiply-adds, the table
  // (transposed, [2R][2M]) is read 16 bytes per thread and row -- 48 loads per thread instead of 192 -- and the 4 kp
  // results leave as one contiguous run.  (M2 = 2M is a multiple of 4 whenever M is even; odd M takes the tail loop.)
  const int M4 = M2 & ~3;
  for (int mu0 = 4 * tid; mu0 < M4; mu0 += 4 * nth) {
    float acc[4][KP];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < KP; ++k) acc[j][k] = 0.f;
    const bool vec = ((M2 & 3) == 0);                  // rows of the transposed table are 16-byte aligned
#pragma unroll 12
    for (int rho = 0; rho < R2; ++rho) {
      float f[4];
      if (vec) {
        const float4 f4 = *reinterpret_cast<const float4*>(fs_t + (long)rho * M2 + mu0);
        f[0] = f4.x; f[1] = f4.y; f[2] = f4.z; f[3] = f4.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = fs_t[(long)rho * M2 + mu0 + j];
      }
      const float4* row = reinterpret_cast<const float4*>(col_sm + rho * KP);
#pragma unroll
      for (int q = 0; q < KP / 4; ++q) {
        const float4 v = row[q];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j][4 * q] = fmaf(f[j], v.x, acc[j][4 * q]); acc[j][4 * q + 1] = fmaf(f[j], v.y, acc[j][4 * q + 1]);
          acc[j][4 * q + 2] = fmaf(f[j], v.z, acc[j][4 * q + 2]); acc[j][4 * q + 3] = fmaf(f[j], v.w, acc[j][4 * q + 3]);
        }
      }
    }
    float4* dst = reinterpret_cast<float4*>(t1 + (bo * M2 + mu0) * KP);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < KP / 4; ++q)
        dst[j * (KP / 4) + q] = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
  }
  for (int mu = M4 + tid; mu < M2; mu += nth) {        // at most three rows (odd M)
    float acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = 0.f;
    for (int rho = 0; rho < R2; ++rho) {
      const float f = fs_t[(long)rho * M2 + mu];
#pragma unroll
      for (int k = 0; k < KP; ++k) acc[k] = fmaf(f, col_sm[rho * KP + k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < KP; ++k) t1[(bo * M2 + mu) * KP + k] = acc[k];
  }
}

// row spectra of the two constant rows the lifted block needs -- ones(N) and gy -- straight from the fp32 analysis
// table: sc[0][r] = sum_n Fa[r][n], sc[1][r] = sum_n Fa[r][n] gy[n]; one wave per r (as a 2-row GEMM this took 26 us
// of launch latency in every forward)
__global__ __launch_bounds__(256) void k_lift_const_spectra(const float* __restrict__ fa, int ldn, const float* __restrict__ gy,
                                                            float* __restrict__ sc, int N, int R2) {
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wv; r < R2; r += gridDim.x * 4) {
    float a = 0.f, b = 0.f;
    for (int n = l; n < N; n += 64) {
      const float f = fa[(long)r * ldn + n];
      a += f;
      b = fmaf(f, gy[n], b);
    }
    a = wave_sum_dpp(a);
    b = wave_sum_dpp(b);
    if (l == 0) { sc[r] = a; sc[R2 + r] = b; }
  }
}

// kp in {4, 8, 12, 16}, the staged block within 64 KB, 16-byte aligned workspaces; RPDE_COL_FUSED=0: the three GEMM-shaped steps
static bool col_stage_ok(int M, int m1, int kp, const float* s1, const float* t1) {
  if (const char* e = getenv("RPDE_COL_FUSED")) if (e[0] == '0') return false;
  return kp >= 4 && kp <= 16 && kp % 4 == 0 && (size_t)2 * M * kp * 4 <= 65536 && m1 >= 1 &&
         ((reinterpret_cast<uintptr_t>(s1) | reinterpret_cast<uintptr_t>(t1)) & 15) == 0;
}

// s1 [B*Ci][2M][kp] -> s2 [B*Ci][2R][kp] (kept for a backward) -> t1 [B*Co][2M][kp]
static int col_stage(const rpde_plan* pm, const float* s1, float* s2, const float* w1, const float* w2, float* t1,
                     const MixGeom& g, int M, hipStream_t st, const ColLift* lift = nullptr) {
  const int M2 = 2 * M, R2 = 2 * g.R;
  const size_t lds_a = sizeof(float) * (size_t)M2 * g.kp, lds_b = sizeof(float) * (size_t)R2 * g.kp;
  const dim3 ga((unsigned)(g.B * g.Ci)), gb((unsigned)(g.B * g.Co)), bk(COL_THREADS);
  const dim3 bkm(g.R * g.kp > COL_THREADS ? COL_MIX_THREADS : COL_THREADS);
  const int wa = (R2 / 4 + 0) < 4 ? 4 : (R2 / 4 > COL_ANA_THREADS / 64 ? COL_ANA_THREADS / 64 : R2 / 4);
  const dim3 bka(64 * wa);
  switch (g.kp) {
    case 4:
      if (lift) hipLaunchKernelGGL((k_col_analysis<4, true>), ga, bka, lds_a, st, pm->fa, s1, s2, M2, R2, *lift);
      else hipLaunchKernelGGL((k_col_analysis<4, false>), ga, bka, lds_a, st, pm->fa, s1, s2, M2, R2, ColLift{});
      hipLaunchKernelGGL((k_col_mix_synthesis<4>), gb, bkm, lds_b, st, s2, w1, w2, pm->fs_t, t1, g, M2);
      break;
    case 8:
      if (lift) hipLaunchKernelGGL((k_col_analysis<8, true>), ga, bka, lds_a, st, pm->fa, s1, s2, M2, R2, *lift);
      else hipLaunchKernelGGL((k_col_analysis<8, false>), ga, bka, lds_a, st, pm->fa, s1, s2, M2, R2, ColLift{});
      hipLaunchKernelGGL((k_col_mix_synthesis<8>), gb, bkm, lds_b, st, s2, w1, w2, pm->fs_t, t1, g, M2);
      break;
    case 12:
      if (lift) hipLaunchKernelGGL((k_col_analysis<12, true>), ga, bka, lds_a, st, pm->fa, s1, s2, M2, R2, *lift);
      else hipLaunchKernelGGL((k_col_analysis<12, false>), ga, bka, lds_a, st, pm->fa, s1, s2, M2, R2, ColLift{});
      hipLaunchKernelGGL((k_col_mix_synthesis<12>), gb, bkm, lds_b, st, s2, w1, w2, pm->fs_t, t1, g, M2);
      break;
    default:
      if (lift) hipLaunchKernelGGL((k_col_analysis<16, true>), ga, bka, lds_a, st, pm->fa, s1, s2, M2, R2, *lift);
      else hipLaunchKernelGGL((k_col_analysis<16, false>), ga, bka, lds_a, st, pm->fa, s1, s2, M2, R2, ColLift{});
      hipLaunchKernelGGL((k_col_mix_synthesis<16>), gb, bkm, lds_b, st, s2, w1, w2, pm->fs_t, t1, g, M2);
      break;
  }
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // namespace rpde

using namespace rpde;

extern "C" {

// --------------------------------- 1-D --------------------------------------
size_t rpde_spectral1d_ws_bytes(int B, int Cin, int Cout, int n, int K) {
  const int kp = (K + 3) / 4 * 4;
  const int cm = Cin > Cout ? Cin : Cout;
  const size_t a = thin_slab_floats((long)B * Cin, 2 * kp, n), b = thin_slab_floats((long)B * Cout, 2 * kp, n);
  return 2 * arena_bytes((size_t)B * cm * 2 * kp) + arena_bytes(a > b ? a : b);
}

int rpde_spectral1d_fwd(const float* x, const float* w, float* out, float* spec_in, int B, int Cin, int Cout, int n, int K,
                        int act_in, void* ws, size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(x && w && out && spec_in && B > 0 && Cin > 0 && Cout > 0 && n > 0 && K > 0, "spectral1d_fwd: bad arguments");
  if (K > n / 2 + 1) { set_error("SpectralConv1d: modes1=%d exceeds n//2+1=%d", K, n / 2 + 1); return RPDE_ERR_MODES; }
  hipStream_t st = as_stream(stream);
  const rpde_plan* pl;
  RPDE_TRY(get_plan(&pl, n, K, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  Arena ar(ws, ws_bytes);
  float* ospec = ar.take((size_t)B * Cout * 2 * pl->kp);
  const size_t nslab = thin_slab_floats((long)B * Cin, 2 * pl->kp, n);
  float* slabs = nslab ? ar.take(nslab) : nullptr;
  if (!ar.ok()) { set_error("spectral1d_fwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(cf_analysis(pl, x, spec_in, (long)B * Cin, n, act_in, st, slabs));
  if (pl->kp != K) RPDE_HIP(hipMemsetAsync(ospec, 0, sizeof(float) * (size_t)B * Cout * 2 * pl->kp, st));
  MixGeom g{B, Cin, Cout, 1, 1, K, pl->kp};
  RPDE_TRY(launch_mix(0, spec_in, nullptr, w, w, ospec, nullptr, g, st));
  return cf_synthesis(pl, ospec, out, (long)B * Cout, n, st);
}

int rpde_spectral1d_bwd(const float* grad_out, const float* spec_in, const float* w, const float* x, float* grad_x,
                        float* grad_w, int B, int Cin, int Cout, int n, int K, int act_in, void* ws, size_t ws_bytes,
                        void* stream) {
  RPDE_CHECK_ARG(grad_out && spec_in && w && B > 0 && Cin > 0 && Cout > 0 && n > 0 && K > 0, "spectral1d_bwd: bad arguments");
  RPDE_CHECK_ARG(!act_in || x, "spectral1d_bwd: act_in needs x");
  if (K > n / 2 + 1) { set_error("SpectralConv1d: modes1=%d exceeds n//2+1=%d", K, n / 2 + 1); return RPDE_ERR_MODES; }
  hipStream_t st = as_stream(stream);
  const rpde_plan* pl;
  RPDE_TRY(get_plan(&pl, n, K, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  Arena ar(ws, ws_bytes);
  float* gspec = ar.take((size_t)B * Cout * 2 * pl->kp);
  float* dspec = ar.take((size_t)B * Cin * 2 * pl->kp);
  const size_t nslab = thin_slab_floats((long)B * Cout, 2 * pl->kp, n);
  float* slabs = nslab ? ar.take(nslab) : nullptr;
  if (!ar.ok()) { set_error("spectral1d_bwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(cf_synthesis_T(pl, grad_out, gspec, (long)B * Cout, n, st, slabs));
  MixGeom g{B, Cin, Cout, 1, 1, K, pl->kp};
  if (grad_w) RPDE_TRY(launch_mix(2, spec_in, gspec, nullptr, nullptr, grad_w, grad_w, g, st));
  if (grad_x) {
    if (pl->kp != K) RPDE_HIP(hipMemsetAsync(dspec, 0, sizeof(float) * (size_t)B * Cin * 2 * pl->kp, st));
    RPDE_TRY(launch_mix(1, gspec, nullptr, w, w, dspec, nullptr, g, st));
    RPDE_TRY(cf_analysis_T(pl, dspec, grad_x, (long)B * Cin, n, act_in, x, st));
  }
  return RPDE_OK;
}

// --------------------------------- 2-D --------------------------------------
static inline int r4(int v) { return (v + 3) / 4 * 4; }

size_t rpde_spectral2d_spec_elems(int B, int Cin, int M, int N, int m1, int m2) {
  (void)M; (void)N;
  return (size_t)B * Cin * 2 * (2 * m1) * r4(m2);
}

size_t rpde_spectral2d_ws_bytes(int B, int Cin, int Cout, int M, int N, int m1, int m2) {
  (void)N;
  const int cm = Cin > Cout ? Cin : Cout;
  const size_t stage1 = arena_bytes((size_t)B * cm * M * 2 * r4(m2));
  const size_t small = arena_bytes((size_t)B * cm * 2 * (2 * m1) * r4(m2));
  return 2 * stage1 + 2 * small;
}

int rpde_spectral2d_fwd(const float* x, const float* w1, const float* w2, float* out, float* spec_in, int B, int Cin,
                        int Cout, int M, int N, int m1, int m2, int act_in, void* ws, size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(x && w1 && w2 && out && spec_in && B > 0 && Cin > 0 && Cout > 0 && M > 0 && N > 0 && m1 > 0 && m2 > 0,
                 "spectral2d_fwd: bad arguments");
  if (m2 > N / 2 + 1 || m1 > M) {
    set_error("SpectralConv2d: modes (%d,%d) exceed the spectrum (%d,%d)", m1, m2, M, N / 2 + 1);
    return RPDE_ERR_MODES;
  }
  hipStream_t st = as_stream(stream);
  const rpde_plan *pn, *pm;
  RPDE_TRY(get_plan(&pn, N, m2, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  RPDE_TRY(get_plan(&pm, M, m1, RPDE_NORM_BACKWARD, 0, PLAN_CPLX, st));
  const int kp = pn->kp, R = 2 * m1;
  Arena ar(ws, ws_bytes);
  float* s1 = ar.take((size_t)B * Cin * M * 2 * kp);
  float* t1 = ar.take((size_t)B * Cout * M * 2 * kp);
  float* o2 = ar.take((size_t)B * Cout * 2 * R * kp);
  if (!ar.ok()) { set_error("spectral2d_fwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(cf_analysis(pn, x, s1, (long)B * Cin * M, N, act_in, st));
  MixGeom g{B, Cin, Cout, R, m1, m2, kp};
  if (col_stage_ok(M, m1, kp, s1, t1)) {
    RPDE_TRY(col_stage(pm, s1, spec_in, w1, w2, t1, g, M, st));
  } else {
    RPDE_TRY(cf_rowdft(pm->fa, 2L * M, false, 2 * R, 2 * M, s1, spec_in, B * Cin, kp, st));
    if (kp != m2) RPDE_HIP(hipMemsetAsync(o2, 0, sizeof(float) * (size_t)B * Cout * 2 * R * kp, st));
    RPDE_TRY(launch_mix(0, spec_in, nullptr, w1, w2, o2, nullptr, g, st));
    RPDE_TRY(cf_rowdft(pm->fs, 2L * R, false, 2 * M, 2 * R, o2, t1, B * Cout, kp, st));
  }
  return cf_synthesis(pn, t1, out, (long)B * Cout * M, N, st);
}

// ---- evaluation-mode FNOBlock2d in the library: activation(SpectralConv2d(x) + bypass_conv(x)), reference
// models/fno_blocks.py:63-83.  The spectral branch stops at the row spectra; the streaming kernel of conv_small.hip forms
// the inverse DFT along the last axis, the 1x1 bypass convolution, bias and activation in one pass over x.
size_t rpde_fnoblock2d_eval_ws_bytes(int B, int Cin, int Cout, int M, int N, int m1, int m2) {
  return rpde_spectral2d_ws_bytes(B, Cin, Cout, M, N, m1, m2) + arena_bytes(rpde_spectral2d_spec_elems(B, Cin, M, N, m1, m2));
}

int rpde_fnoblock2d_eval_ok(int Cin, int Cout, int M, int N, int m2) {
  // (pointer alignment is checked at the call; torch allocations are 256-byte aligned)
  return conv1x1_syn_ok(nullptr, nullptr, Cin, Cout, M, N, 2 * ((m2 + 3) / 4 * 4)) ? 1 : 0;
}

int rpde_fnoblock2d_eval_fwd(const float* x, const float* w1, const float* w2, const float* wc, const float* bc, float* out, int B,
                             int Cin, int Cout, int M, int N, int m1, int m2, int act_out, void* ws, size_t ws_bytes,
                             void* stream) {
  RPDE_CHECK_ARG(x && w1 && w2 && wc && out && B > 0 && Cin > 0 && Cout > 0 && M > 0 && N > 0 && m1 > 0 && m2 > 0,
                 "fnoblock2d_eval_fwd: bad arguments");
  if (m2 > N / 2 + 1 || m1 > M) {
    set_error("SpectralConv2d: modes (%d,%d) exceed the spectrum (%d,%d)", m1, m2, M, N / 2 + 1);
    return RPDE_ERR_MODES;
  }
  RPDE_CHECK_ARG(conv1x1_syn_ok(x, out, Cin, Cout, M, N, 2 * ((m2 + 3) / 4 * 4)), "fnoblock2d_eval_fwd: shape not covered (%d -> %d on %d x %d)", Cin, Cout, M, N);
  hipStream_t st = as_stream(stream);
  const rpde_plan *pn, *pm;
  RPDE_TRY(get_plan(&pn, N, m2, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  RPDE_TRY(get_plan(&pm, M, m1, RPDE_NORM_BACKWARD, 0, PLAN_CPLX, st));
  const int kp = pn->kp, R = 2 * m1;
  Arena ar(ws, ws_bytes);
  float* s1 = ar.take((size_t)B * Cin * M * 2 * kp);
  float* t1 = ar.take((size_t)B * Cout * M * 2 * kp);
  float* o2 = ar.take((size_t)B * Cout * 2 * R * kp);
  float* s2 = ar.take((size_t)B * Cin * 2 * R * kp);
  if (!ar.ok()) { set_error("fnoblock2d_eval_fwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(cf_analysis(pn, x, s1, (long)B * Cin * M, N, 0, st));
  MixGeom g{B, Cin, Cout, R, m1, m2, kp};
  if (col_stage_ok(M, m1, kp, s1, t1)) {
    RPDE_TRY(col_stage(pm, s1, s2, w1, w2, t1, g, M, st));
  } else {
    RPDE_TRY(cf_rowdft(pm->fa, 2L * M, false, 2 * R, 2 * M, s1, s2, B * Cin, kp, st));
    if (kp != m2) RPDE_HIP(hipMemsetAsync(o2, 0, sizeof(float) * (size_t)B * Cout * 2 * R * kp, st));
    RPDE_TRY(launch_mix(0, s2, nullptr, w1, w2, o2, nullptr, g, st));
    RPDE_TRY(cf_rowdft(pm->fs, 2L * R, false, 2 * M, 2 * R, o2, t1, B * Cout, kp, st));
  }
  if (conv_syn_h2_ok(x, out, t1, Cin, Cout, M, N, 2 * kp))
    return conv_syn_h2(x, wc, bc, t1, pn->fs_t, out, B, Cin, Cout, M, N, 2 * kp, act_out, st);
  return conv1x1_syn(x, wc, bc, t1, pn->fs_t, out, B, Cin, Cout, M, N, 2 * kp, act_out, st);
}

// ---- the LAST block and the projection MLP in one entry point (evaluation): out = mlp2(gelu(mlp1(act(SpectralConv2d(x) +
// Conv2d_1x1(x))))), reference models/fno.py:143-150.  The spectral branch as in rpde_fnoblock2d_eval_fwd; its last
// transform, the bypass convolution, the activation and both projection layers are ONE pass over x (conv_proj_h2.hip):
// the block's output is never written.
int rpde_fnoblock2d_proj_eval_ok(int Cin, int Cout, int M, int N, int m1, int m2, int Cmid, int Cq) {
  const int kp = (m2 + 3) / 4 * 4;
  return m2 <= N / 2 + 1 && m1 <= M && conv_syn_proj_ok(Cin, Cout, M, N, 2 * kp, Cmid, Cq) ? 1 : 0;
}

int rpde_fnoblock2d_proj_eval_fwd(const float* x, const float* w1, const float* w2, const float* wc, const float* bc,
                                  const float* pw1, const float* pb1, const float* pw2, const float* pb2, float* out, int B,
                                  int Cin, int Cout, int M, int N, int m1, int m2, int act_out, int Cmid, int Cq, void* ws,
                                  size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(x && w1 && w2 && wc && pw1 && pw2 && out && B > 0, "fnoblock2d_proj_eval_fwd: bad arguments");
  RPDE_CHECK_ARG(rpde_fnoblock2d_proj_eval_ok(Cin, Cout, M, N, m1, m2, Cmid, Cq), "fnoblock2d_proj_eval_fwd: shape not covered");
  RPDE_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 15) == 0, "fnoblock2d_proj_eval_fwd: x must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const rpde_plan *pn, *pm;
  RPDE_TRY(get_plan(&pn, N, m2, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  RPDE_TRY(get_plan(&pm, M, m1, RPDE_NORM_BACKWARD, 0, PLAN_CPLX, st));
  const int kp = pn->kp, R = 2 * m1;
  Arena ar(ws, ws_bytes);
  float* s1 = ar.take((size_t)B * Cin * M * 2 * kp);
  float* t1 = ar.take((size_t)B * Cout * M * 2 * kp);
  float* o2 = ar.take((size_t)B * Cout * 2 * R * kp);
  float* s2 = ar.take((size_t)B * Cin * 2 * R * kp);
  if (!ar.ok()) { set_error("fnoblock2d_proj_eval_fwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(cf_analysis(pn, x, s1, (long)B * Cin * M, N, 0, st));
  MixGeom g{B, Cin, Cout, R, m1, m2, kp};
  if (col_stage_ok(M, m1, kp, s1, t1)) {
    RPDE_TRY(col_stage(pm, s1, s2, w1, w2, t1, g, M, st));
  } else {
    RPDE_TRY(cf_rowdft(pm->fa, 2L * M, false, 2 * R, 2 * M, s1, s2, B * Cin, kp, st));
    if (kp != m2) RPDE_HIP(hipMemsetAsync(o2, 0, sizeof(float) * (size_t)B * Cout * 2 * R * kp, st));
    RPDE_TRY(launch_mix(0, s2, nullptr, w1, w2, o2, nullptr, g, st));
    RPDE_TRY(cf_rowdft(pm->fs, 2L * R, false, 2 * M, 2 * R, o2, t1, B * Cout, kp, st));
  }
  return conv_syn_proj(x, wc, bc, t1, pn->fs_t, pw1, pb1, pw2, pb2, out, B, Cout, M, N, 2 * kp, Cmid, Cq, act_out, st);
}

// ---- evaluation-mode FNO2d: lifting + first block without the lifted field (reference models/fno.py:121-147:
// cat(x, gridx, gridy) -> lifting -> fno_blocks[0]); u [B,1,M,N], gx [M], gy [N], wl [C,3], bl [C] ----
size_t rpde_fno2d_lift_block_eval_ws_bytes(int B, int C, int Cout, int M, int N, int m1, int m2) {
  const size_t kp = (size_t)((m2 + 3) / 4 * 4), R = 2 * (size_t)m1;
  return arena_bytes((size_t)B * M * 2 * kp) + arena_bytes(4 * kp) +
         arena_bytes((size_t)B * C * 2 * R * kp) + arena_bytes((size_t)B * Cout * M * 2 * kp) + 4096;
}

int rpde_fno2d_lift_block_eval_ok(int Cu, int C, int Cout, int M, int N, int m1, int m2) {
  if (const char* e = getenv("RPDE_LIFT_FUSED")) if (e[0] == '0') return 0;
  const int kp = (m2 + 3) / 4 * 4;
  return Cu == 1 && M <= 1024 && m2 <= N / 2 + 1 && m1 <= M && conv_syn_h2_ok(nullptr, nullptr, nullptr, C, Cout, M, N, 2 * kp) &&
         col_stage_ok(M, m1, kp, nullptr, nullptr) ? 1 : 0;
}

int rpde_fno2d_lift_block_eval_fwd(const float* u, const float* gx, const float* gy, const float* wl, const float* bl,
                                   const float* w1, const float* w2, const float* wc, const float* bc, float* out, int B, int C,
                                   int Cout, int M, int N, int m1, int m2, int act_out, void* ws, size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(u && gx && gy && wl && w1 && w2 && wc && out && B > 0 && C > 0 && Cout > 0 && M > 0 && N > 0 && m1 > 0 && m2 > 0,
                 "fno2d_lift_block_eval_fwd: bad arguments");
  RPDE_CHECK_ARG(rpde_fno2d_lift_block_eval_ok(1, C, Cout, M, N, m1, m2), "fno2d_lift_block_eval_fwd: shape not covered");
  hipStream_t st = as_stream(stream);
  const rpde_plan *pn, *pm;
  RPDE_TRY(get_plan(&pn, N, m2, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  RPDE_TRY(get_plan(&pm, M, m1, RPDE_NORM_BACKWARD, 0, PLAN_CPLX, st));
  const int kp = pn->kp, R = 2 * m1;
  Arena ar(ws, ws_bytes);
  float* su = ar.take((size_t)B * M * 2 * kp);
  float* sc = ar.take(4 * (size_t)kp);
  float* s2 = ar.take((size_t)B * C * 2 * R * kp);
  float* t1 = ar.take((size_t)B * Cout * M * 2 * kp);
  if (!ar.ok()) { set_error("fno2d_lift_block_eval_fwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_CHECK_ARG(((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0,
                 "fno2d_lift_block_eval_fwd: u, gy and out must be 16-byte aligned");
  hipLaunchKernelGGL(k_lift_const_spectra, dim3((2 * kp + 3) / 4), dim3(256), 0, st, pn->fa, pn->ldn, gy, sc, N, 2 * kp);
  RPDE_TRY(cf_analysis(pn, u, su, (long)B * M, N, 0, st));
  MixGeom g{B, C, Cout, R, m1, m2, kp};
  ColLift L{su, sc, wl, bl, gx, C};
  RPDE_TRY(col_stage(pm, nullptr, s2, w1, w2, t1, g, M, st, &L));
  return conv_syn_h2(nullptr, wc, bc, t1, pn->fs_t, out, B, C, Cout, M, N, 2 * kp, act_out, st, u, wl, bl, gx, gy);
}

int rpde_spectral2d_bwd(const float* grad_out, const float* spec_in, const float* w1, const float* w2, const float* x,
                        float* grad_x, float* grad_w1, float* grad_w2, int B, int Cin, int Cout, int M, int N, int m1, int m2,
                        int act_in, void* ws, size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(grad_out && spec_in && w1 && w2 && B > 0 && Cin > 0 && Cout > 0 && M > 0 && N > 0 && m1 > 0 && m2 > 0,
                 "spectral2d_bwd: bad arguments");
  RPDE_CHECK_ARG(!act_in || x, "spectral2d_bwd: act_in needs x");
  if (m2 > N / 2 + 1 || m1 > M) { set_error("SpectralConv2d: modes exceed the spectrum"); return RPDE_ERR_MODES; }
  hipStream_t st = as_stream(stream);
  const rpde_plan *pn, *pm;
  RPDE_TRY(get_plan(&pn, N, m2, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  RPDE_TRY(get_plan(&pm, M, m1, RPDE_NORM_BACKWARD, 0, PLAN_CPLX, st));
  const int kp = pn->kp, R = 2 * m1;
  Arena ar(ws, ws_bytes);
  float* gt1 = ar.take((size_t)B * Cout * M * 2 * kp);
  float* ds1 = ar.take((size_t)B * Cin * M * 2 * kp);
  float* go2 = ar.take((size_t)B * Cout * 2 * R * kp);
  float* ds2 = ar.take((size_t)B * Cin * 2 * R * kp);
  if (!ar.ok()) { set_error("spectral2d_bwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(cf_synthesis_T(pn, grad_out, gt1, (long)B * Cout * M, N, st));
  RPDE_TRY(cf_rowdft(pm->fs, 2L * R, true, 2 * R, 2 * M, gt1, go2, B * Cout, kp, st));
  MixGeom g{B, Cin, Cout, R, m1, m2, kp};
  if (grad_w1 && grad_w2) RPDE_TRY(launch_mix(2, spec_in, go2, nullptr, nullptr, grad_w1, grad_w2, g, st));
  if (grad_x) {
    if (kp != m2) RPDE_HIP(hipMemsetAsync(ds2, 0, sizeof(float) * (size_t)B * Cin * 2 * R * kp, st));
    RPDE_TRY(launch_mix(1, go2, nullptr, w1, w2, ds2, nullptr, g, st));
    RPDE_TRY(cf_rowdft(pm->fa, 2L * M, true, 2 * M, 2 * R, ds2, ds1, B * Cin, kp, st));
    RPDE_TRY(cf_analysis_T(pn, ds1, grad_x, (long)B * Cin * M, N, act_in, x, st));
  }
  return RPDE_OK;
}


// ---- spectral resize (reference: utils/res_utils.py:29-50 `resize`, :93-125 `resize_1d`) ------------------
// rfft -> keep the bins both sizes share -> irfft at the new size, times out/in: the same truncated-DFT
// plans, analysis at the source size and synthesis at the target size.
size_t rpde_resize1d_ws_bytes(int64_t rows, int n_in, int n_out) {
  const int k = (n_in / 2 + 1) < (n_out / 2 + 1) ? (n_in / 2 + 1) : (n_out / 2 + 1);
  return arena_bytes((size_t)rows * 2 * r4(k));
}

int rpde_resize1d(const float* x, float* out, int64_t rows, int n_in, int n_out, void* ws, size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(x && out && rows > 0 && n_in > 0 && n_out > 0 && rows < (1L << 31), "resize1d: bad arguments");
  hipStream_t st = as_stream(stream);
  const int k = (n_in / 2 + 1) < (n_out / 2 + 1) ? (n_in / 2 + 1) : (n_out / 2 + 1);
  const rpde_plan *pa, *ps;
  RPDE_TRY(get_plan(&pa, n_in, k, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  RPDE_TRY(get_plan(&ps, n_out, k, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  Arena ar(ws, ws_bytes);
  float* spec = ar.take((size_t)rows * 2 * pa->kp);
  if (!ar.ok()) { set_error("resize1d: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(cf_analysis(pa, x, spec, rows, n_in, 0, st));
  return cf_synthesis(ps, spec, out, rows, n_out, st, (float)((double)n_out / (double)n_in));
}

size_t rpde_resize2d_ws_bytes(int64_t rows, int M, int N, int Mo, int No) {
  const int k2 = (N / 2 + 1) < (No / 2 + 1) ? (N / 2 + 1) : (No / 2 + 1);
  const int top = ((M + 1) / 2) < ((Mo + 1) / 2) ? (M + 1) / 2 : (Mo + 1) / 2;
  const int bot = (M / 2) < (Mo / 2) ? M / 2 : Mo / 2;
  const int mm = M > Mo ? M : Mo;
  return 2 * arena_bytes((size_t)rows * mm * 2 * r4(k2)) + arena_bytes((size_t)rows * 2 * (top + bot) * r4(k2));
}

// x [rows, M, N] -> out [rows, Mo, No]  (rows = batch * channels)
int rpde_resize2d(const float* x, float* out, int64_t rows, int M, int N, int Mo, int No, void* ws, size_t ws_bytes,
                  void* stream) {
  RPDE_CHECK_ARG(x && out && rows > 0 && M > 0 && N > 0 && Mo > 0 && No > 0 && rows * (long)(M > Mo ? M : Mo) < (1L << 31),
                 "resize2d: bad arguments");
  hipStream_t st = as_stream(stream);
  const int k2 = (N / 2 + 1) < (No / 2 + 1) ? (N / 2 + 1) : (No / 2 + 1);
  const int top = ((M + 1) / 2) < ((Mo + 1) / 2) ? (M + 1) / 2 : (Mo + 1) / 2;
  const int bot = (M / 2) < (Mo / 2) ? M / 2 : Mo / 2;
  const int R = top + bot;
  const rpde_plan *pn, *pno, *pm, *pmo;
  RPDE_TRY(get_plan(&pn, N, k2, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  RPDE_TRY(get_plan(&pno, No, k2, RPDE_NORM_BACKWARD, 1, PLAN_REAL, st));
  RPDE_TRY(get_plan(&pm, M, top, RPDE_NORM_BACKWARD, 0, PLAN_CPLX, st, bot));
  RPDE_TRY(get_plan(&pmo, Mo, top, RPDE_NORM_BACKWARD, 0, PLAN_CPLX, st, bot));
  const int kp = pn->kp;
  const int mm = M > Mo ? M : Mo;
  Arena ar(ws, ws_bytes);
  float* s1 = ar.take((size_t)rows * mm * 2 * kp);
  float* t1 = ar.take((size_t)rows * mm * 2 * kp);
  float* s2 = ar.take((size_t)rows * 2 * R * kp);
  if (!ar.ok()) { set_error("resize2d: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(cf_analysis(pn, x, s1, rows * M, N, 0, st));
  RPDE_TRY(cf_rowdft(pm->fa, 2L * M, false, 2 * R, 2 * M, s1, s2, (int)rows, kp, st));
  RPDE_TRY(cf_rowdft(pmo->fs, 2L * R, false, 2 * Mo, 2 * R, s2, t1, (int)rows, kp, st));
  const double scale = ((double)Mo / M) * ((double)No / N);
  return cf_synthesis(pno, t1, out, rows * Mo, No, st, (float)scale);
}

}  // extern "C"
