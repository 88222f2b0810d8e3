// Strided batched fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[z][m,n] (+)= alpha * sum_k actA(A[z][m,k]) * actB(B[z][k,n])  (+ bias, * act'(aux))
//
// Every heavy operation of the hot path is expressed through this kernel:
//   * the pointwise FeedForward / lifting / projection linears (forward NT,
//     backward-data NN, weight-gradient TN with split-K over the grid points),
//   * the truncated real DFTs (analysis [2K,n]·X, synthesis [n,2K]·A) for both
//     channels-last (FFNO) and channels-first (FNO) tensors,
//   * the per-mode complex channel mixing as real [rows,2C]·[2C,2C] blocks.
//
// Design (MI355X_MICROARCH / cdna_hip_programming guides):
//   * exact-fp32 MFMA 32x32x2: 64 cycles/SIMD per instruction, so LDS fragment
//     traffic is far from the limit; the kernel is MFMA-issue bound when the
//     reduction is long and HBM bound when it is short.
//   * 256 threads = 4 waves arranged WM x WN, each wave TM x TN tiles of 32x32.
//   * BK = 32 per stage, two LDS stages, register prefetch of the next stage's
//     global loads (issue early / write late), one barrier per stage.
//   * k order inside a stage is permuted (lane half h owns k = 16h .. 16h+15) so
//     a k-major operand is read with ds_read_b128; row stride BK+4 dwords makes
//     those reads bank-conflict free (36r mod 64 distinct multiples of 4).
//   * operands may be k-major or "x-major" (the non-reduction index
//     contiguous); both are loaded from HBM as coalesced 16-byte vectors along
//     their contiguous index and stored to LDS in the same orientation.
//   * activation (+ counter-hash dropout) is applied to an operand while it is
//     staged, so hidden activations are never written to HBM.
//   * tile ids are dealt so that the N-tiles of one M-tile land on one XCD
//     (blocks b and b+8 share an L2) and reuse the A panel from that L2.
#include "rpde_internal.h"

#include <mutex>

namespace rpde {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int NTHREADS = 256;

struct GemmK {
  const float* A; const float* B; float* C;
  int M, N, K;
  long lda, ldb, ldc;
  int zdiv, ztotal;
  long sA1, sA2, sB1, sB2, sC1, sC2;
  int ksplit, kchunk; long sCk;
  float alpha; int accumulate;
  const float* bias; int bias_mode;
  int act_a, act_b, epi_dact, write_act;
  const float* aux; long ldaux;
  DropCfg drop; long drop_ld; int drop_where;
  int a_vec, b_vec;
  int mtiles, ntiles;
};

template <int ROWS, bool KMAJOR>
struct Tile {
  static constexpr int NV = ROWS * BK / 4 / NTHREADS;          // float4 per thread
  static constexpr int LDK = BK + 4;                           // k-major row stride
  static constexpr int LDS_FLOATS = KMAJOR ? ROWS * LDK : BK * ROWS;
  static_assert(NV >= 1, "tile too small for 256 threads");

  __device__ __forceinline__ static void coords(int v, int& rr, int& kk) {
    if (KMAJOR) { rr = v >> 3; kk = (v & 7) << 2; }
    else { kk = v / (ROWS / 4); rr = (v % (ROWS / 4)) << 2; }
  }

  // HBM -> registers (zero fill outside [.., rmax) x [.., kend))
  __device__ __forceinline__ static void load(float4 (&r)[NV], const float* __restrict__ base, long ld,
                                              int r0, int rmax, int k0, int kend, int vec_ok, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int rr, kk;
      coords(tid + i * NTHREADS, rr, kk);
      const int gr = r0 + rr, gk = k0 + kk;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < rmax && gk < kend) {
        if (KMAJOR) {
          const float* p = base + (long)gr * ld + gk;
          if (vec_ok && gk + 3 < kend) {
            val = *reinterpret_cast<const float4*>(p);
          } else {
            val.x = p[0];
            if (gk + 1 < kend) val.y = p[1];
            if (gk + 2 < kend) val.z = p[2];
            if (gk + 3 < kend) val.w = p[3];
          }
        } else {
          const float* p = base + (long)gk * ld + gr;
          if (vec_ok && gr + 3 < rmax) {
            val = *reinterpret_cast<const float4*>(p);
          } else {
            val.x = p[0];
            if (gr + 1 < rmax) val.y = p[1];
            if (gr + 2 < rmax) val.z = p[2];
            if (gr + 3 < rmax) val.w = p[3];
          }
        }
      }
      r[i] = val;
    }
  }

  // registers -> LDS, applying h = act(dropout(z)) when requested
  __device__ __forceinline__ static void store(float* __restrict__ lds, const float4 (&r)[NV], int tid, int act,
                                               bool use_drop, const DropCfg& drop, long drop_ld, int r0, int k0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int rr, kk;
      coords(tid + i * NTHREADS, rr, kk);
      float4 v = r[i];
      if (act != RPDE_ACT_IDENTITY || use_drop) {
        float s[4] = {1.f, 1.f, 1.f, 1.f};
        if (use_drop) {
          const long slow = KMAJOR ? (long)(r0 + rr) : (long)(k0 + kk);
          const long fast = KMAJOR ? (long)(k0 + kk) : (long)(r0 + rr);
          const uint64_t id = (uint64_t)(slow * drop_ld + fast);
          if ((id & 3) == 0) {
            drop_scale4(drop, id, s);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = drop_scale1(drop, id + j);
          }
        }
        v.x = act_f(act, v.x * s[0]);
        v.y = act_f(act, v.y * s[1]);
        v.z = act_f(act, v.z * s[2]);
        v.w = act_f(act, v.w * s[3]);
      }
      float* dst = KMAJOR ? (lds + rr * LDK + kk) : (lds + kk * ROWS + rr);
      *reinterpret_cast<float4*>(dst) = v;
    }
  }

  // fragment of 4 consecutive k-steps for MFMA lane (i = l31, half = lh), chunk q
  __device__ __forceinline__ static void frag(const float* __restrict__ lds, int row, int lh, int q, float (&f)[4]) {
    if (KMAJOR) {
      const float4 t = *reinterpret_cast<const float4*>(lds + row * LDK + lh * (BK / 2) + q * 4);
      f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) f[j] = lds[(lh * (BK / 2) + q * 4 + j) * ROWS + row];
    }
  }
};

template <int WM, int WN, int TM, int TN, bool AK, bool BKM>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(const GemmK g) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static_assert(WM * WN == 4, "four waves per workgroup");
  using TA = Tile<BM, AK>;
  using TB = Tile<BN, BKM>;
  __shared__ __attribute__((aligned(16))) float smem[2 * (TA::LDS_FLOATS + TB::LDS_FLOATS)];
  float* const As0 = smem;
  float* const As1 = smem + TA::LDS_FLOATS;
  float* const Bs0 = smem + 2 * TA::LDS_FLOATS;
  float* const Bs1 = Bs0 + TB::LDS_FLOATS;

  const int tid = threadIdx.x;
  // ---- which tile, which batch entry, which K slice -------------------------
  const int L = blockIdx.x;
  const int mt = (L / (8 * g.ntiles)) * 8 + (L & 7);
  const int nt = (L >> 3) % g.ntiles;
  if (mt >= g.mtiles) return;
  const int zz = blockIdx.z * gridDim.y + blockIdx.y;
  if (zz >= g.ztotal) return;
  const int z = zz / g.ksplit, ks = zz - z * g.ksplit;
  const int z1 = z / g.zdiv, z2 = z - z1 * g.zdiv;
  const float* __restrict__ A = g.A + z1 * g.sA1 + z2 * g.sA2;
  const float* __restrict__ B = g.B + z1 * g.sB1 + z2 * g.sB2;
  const long coff = z1 * g.sC1 + z2 * g.sC2 + (long)ks * g.sCk;
  float* __restrict__ C = g.C + coff;
  const int m0 = mt * BM, n0 = nt * BN;
  const int kbeg = ks * g.kchunk;
  const int kend = min(g.K, kbeg + g.kchunk);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool drop_a = g.drop.on() && (g.drop_where & 1);
  const bool drop_b = g.drop.on() && (g.drop_where & 2);
  const bool drop_e = g.drop.on() && (g.drop_where & 4);
  float4 ra[TA::NV], rb[TB::NV];
  if (nkt > 0) {
    TA::load(ra, A, g.lda, m0, g.M, kbeg, kend, g.a_vec, tid);
    TB::load(rb, B, g.ldb, n0, g.N, kbeg, kend, g.b_vec, tid);
    TA::store(As0, ra, tid, g.act_a, drop_a, g.drop, g.drop_ld, m0, kbeg);
    TB::store(Bs0, rb, tid, g.act_b, drop_b, g.drop, g.drop_ld, n0, kbeg);
  }
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const float* as = (kt & 1) ? As1 : As0;
    const float* bs = (kt & 1) ? Bs1 : Bs0;
    const bool more = kt + 1 < nkt;
    const int knext = kbeg + (kt + 1) * BK;
    if (more) {  // issue the next stage's HBM loads before computing this one
      TA::load(ra, A, g.lda, m0, g.M, knext, kend, g.a_vec, tid);
      TB::load(rb, B, g.ldb, n0, g.N, knext, kend, g.b_vec, tid);
    }
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) TA::frag(as, (wm * TM + i) * 32 + l31, lh, q, af[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) TB::frag(bs, (wn * TN + j) * 32 + l31, lh, q, bf[j]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
    if (more) {
      TA::store((kt & 1) ? As0 : As1, ra, tid, g.act_a, drop_a, g.drop, g.drop_ld, m0, knext);
      TB::store((kt & 1) ? Bs0 : Bs1, rb, tid, g.act_b, drop_b, g.drop, g.drop_ld, n0, knext);
    }
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const float* __restrict__ aux = g.aux ? g.aux + coff : nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + l31;
      if (n >= g.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float v = acc[i][j][r] * g.alpha;
        if (g.bias_mode == 1) v += g.bias[n];
        else if (g.bias_mode == 2) v += g.bias[m];
        if (g.epi_dact) {
          float s = 1.f;
          if (drop_e) s = drop_scale1(g.drop, (uint64_t)((long)m * g.drop_ld + n));
          const float u = aux[(long)m * g.ldaux + n] * s;
          v = v * dact_f(g.epi_dact, u) * s;
        }
        float* cp = C + (long)m * g.ldc + n;
        if (g.accumulate) v += *cp;
        if (g.write_act) v = act_f(g.write_act, v);
        *cp = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN>
static int launch_cfg(const GemmK& g, bool ak, bool bk, dim3 grid, hipStream_t st) {
  if (ak && bk) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, true, true>), grid, dim3(NTHREADS), 0, st, g);
  else if (ak && !bk) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, true, false>), grid, dim3(NTHREADS), 0, st, g);
  else if (!ak && !bk) hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, false, false>), grid, dim3(NTHREADS), 0, st, g);
  else hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, false, true>), grid, dim3(NTHREADS), 0, st, g);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int launch_gemm(const rpde_gemm_desc& d, hipStream_t st) {
  RPDE_CHECK_ARG(d.A && d.B && d.C, "gemm: null operand");
  RPDE_CHECK_ARG(d.M > 0 && d.N > 0 && d.K >= 0, "gemm: bad dims %d %d %d", d.M, d.N, d.K);
  RPDE_CHECK_ARG(d.batch >= 1 && d.zdiv >= 1 && d.ksplit >= 1, "gemm: bad batch/zdiv/ksplit");
  RPDE_CHECK_ARG(!(d.ksplit > 1 && (d.bias_mode || d.epi_dact || d.accumulate || d.write_act)),
                 "gemm: split-K slabs take no epilogue");
  RPDE_CHECK_ARG(!d.epi_dact || d.aux, "gemm: epi_dact needs aux");
  GemmK g;
  g.A = d.A; g.B = d.B; g.C = d.C; g.M = d.M; g.N = d.N; g.K = d.K;
  g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
  g.zdiv = d.zdiv; g.ztotal = d.batch * d.ksplit;
  g.sA1 = d.sA1; g.sA2 = d.sA2; g.sB1 = d.sB1; g.sB2 = d.sB2; g.sC1 = d.sC1; g.sC2 = d.sC2;
  g.ksplit = d.ksplit; g.sCk = d.sCk;
  int kchunk = (d.K + d.ksplit - 1) / d.ksplit;
  kchunk = ((kchunk + BK - 1) / BK) * BK;
  g.kchunk = kchunk > 0 ? kchunk : BK;
  g.alpha = d.alpha; g.accumulate = d.accumulate;
  g.bias = d.bias; g.bias_mode = d.bias ? d.bias_mode : 0;
  g.act_a = d.act_a; g.act_b = d.act_b; g.epi_dact = d.epi_dact; g.write_act = d.write_act;
  g.aux = d.aux; g.ldaux = d.ldaux;
  g.drop = make_drop(d.drop_p, d.drop_seed); g.drop_ld = d.drop_ld; g.drop_where = d.drop_where;
  // 16-byte vector loads need aligned bases, strides and leading dims
  g.a_vec = al16(d.A) && (d.lda % 4 == 0) && (d.sA1 % 4 == 0) && (d.sA2 % 4 == 0);
  g.b_vec = al16(d.B) && (d.ldb % 4 == 0) && (d.sB1 % 4 == 0) && (d.sB2 % 4 == 0);

  const bool ak = d.a_kmajor != 0, bk = d.b_kmajor != 0;
  const int bm = d.M <= 32 ? 32 : (d.M <= 64 ? 64 : 128);
  int bn = d.N <= 32 ? 32 : (d.N <= 64 ? 64 : 128);
  int BMc, BNc;
  if (bm == 32) { BMc = 32; BNc = 128; }
  else if (bn == 32) { BMc = 128; BNc = 32; }
  else { BMc = bm; BNc = bn; }
  g.mtiles = (d.M + BMc - 1) / BMc;
  g.ntiles = (d.N + BNc - 1) / BNc;
  const long zt = (long)g.ztotal;
  const int gy = (int)(zt < 32768 ? zt : 32768);
  const int gz = (int)((zt + gy - 1) / gy);
  dim3 grid(((g.mtiles + 7) / 8) * 8 * g.ntiles, gy, gz);

  if (BMc == 128 && BNc == 128) return launch_cfg<2, 2, 2, 2>(g, ak, bk, grid, st);
  if (BMc == 128 && BNc == 64) return launch_cfg<4, 1, 1, 2>(g, ak, bk, grid, st);
  if (BMc == 64 && BNc == 128) return launch_cfg<1, 4, 2, 1>(g, ak, bk, grid, st);
  if (BMc == 64 && BNc == 64) return launch_cfg<2, 2, 1, 1>(g, ak, bk, grid, st);
  if (BMc == 128 && BNc == 32) return launch_cfg<4, 1, 1, 1>(g, ak, bk, grid, st);
  return launch_cfg<1, 4, 1, 1>(g, ak, bk, grid, st);   // 32 x 128
}

}  // namespace rpde

extern "C" int rpde_gemm_f32(const rpde_gemm_desc* d, void* stream) {
  if (!d) { rpde::set_error("gemm: null descriptor"); return RPDE_ERR_ARG; }
  return rpde::launch_gemm(*d, rpde::as_stream(stream));
}
