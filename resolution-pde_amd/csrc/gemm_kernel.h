// Strided batched fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   C[z][m,n] (+)= alpha * sum_k actA(A[z][m,k]) * actB(B[z][k,n])  (+ bias, * act'(aux))
//
// Every heavy operation of the hot path is expressed through this kernel:
//   * the pointwise FeedForward / lifting / projection linears (forward NT,
//     backward-data NN, weight-gradient TN with split-K over the grid points),
//   * the truncated real DFTs (analysis [2K,n].X, synthesis [n,2K].A) for both
//     channels-last (FFNO) and channels-first (FNO) tensors,
//   * the per-mode complex channel mixing as real [rows,2C].[2C,2C] blocks.
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
//   * activation (+ counter-hash dropout) is applied to ONE operand while it is
//     staged (template PRO), so hidden activations are never written to HBM.
//   * everything in the k-loop is resolved at compile time (VEC: 16-byte loads
//     only; PRO: which operand is activated): the first version kept those as
//     run-time branches and its loop body overflowed the instruction cache.
//     Odd shapes / unaligned operands take the VEC=false instantiation.
//   * with many M-tiles the N-tiles of one M-tile are dealt to one XCD (blocks b
//     and b+8 share an L2) and reuse the A panel from that L2; with few tiles
//     (batched / split-K launches) ids are dense so the batch index spreads
//     the work over all 8 XCDs.
#pragma once
#include "rpde_internal.h"

namespace rpde {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NTHREADS = 256;
constexpr int BK_MAX = 32;     // host-side K-slice rounding (every kernel BK divides it)

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
  int mtiles, ntiles, swz;
  int cvec;          // C / aux / bias allow 16-byte row accesses: LDS-staged epilogue
  float* colsum;     // optional [mtiles][N] per-M-tile column sums of the stored C
  float* aux_out;    // optional act'(u)*dropscale beside C = act(u)
  const char* Bimg;  // optional pre-split operand (split_weights): B, or A when a_img is set
  int npad;          // rows per image of Bimg
  int a_img;
  const float* acc_src;   // accumulate reads this tensor (laid out like C) instead of C itself
};

template <int ROWS, int BK, bool KMAJOR, bool VEC, bool PRO>
struct Tile {
  static constexpr int NV = ROWS * BK / 4 / NTHREADS;          // float4 per thread
  static constexpr int LDK = BK + 4;                           // k-major row stride
  static constexpr int LDS_FLOATS = KMAJOR ? ROWS * LDK : BK * ROWS;
  static_assert(NV >= 1, "tile too small for 256 threads");

  __device__ __forceinline__ static void coords(int v, int& rr, int& kk) {
    if (KMAJOR) { rr = v / (BK / 4); kk = (v % (BK / 4)) << 2; }
    else { kk = v / (ROWS / 4); rr = (v % (ROWS / 4)) << 2; }
  }

  // HBM -> registers (zero fill outside [.., rmax) x [.., kend))
  __device__ __forceinline__ static void load(float4 (&r)[NV], const float* __restrict__ base, long ld,
                                              int r0, int rmax, int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int rr, kk;
      coords(tid + i * NTHREADS, rr, kk);
      const int gr = r0 + rr, gk = k0 + kk;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (VEC) {
        // host guarantees: extent along the contiguous index is a multiple of 4 and 16-byte aligned.
        // Branch-free: out-of-range lanes read the (valid) base address and are zeroed by a select,
        // so the loads stay in one basic block and the compiler can use counted vmcnt waits.
        const bool ok = gr < rmax && gk < kend;
        const long off = KMAJOR ? (long)gr * ld + gk : (long)gk * ld + gr;
        const float4 t = *reinterpret_cast<const float4*>(base + (ok ? off : 0L));
        val.x = ok ? t.x : 0.f; val.y = ok ? t.y : 0.f; val.z = ok ? t.z : 0.f; val.w = ok ? t.w : 0.f;
      } else if (gr < rmax && gk < kend) {
        if (KMAJOR) {
          const float* p = base + (long)gr * ld + gk;
          val.x = p[0];
          if (gk + 1 < kend) val.y = p[1];
          if (gk + 2 < kend) val.z = p[2];
          if (gk + 3 < kend) val.w = p[3];
        } else {
          const float* p = base + (long)gk * ld + gr;
          val.x = p[0];
          if (gr + 1 < rmax) val.y = p[1];
          if (gr + 2 < rmax) val.z = p[2];
          if (gr + 3 < rmax) val.w = p[3];
        }
      }
      r[i] = val;
    }
  }

  // registers -> LDS, applying h = act(dropout(z)) when this operand is the activated one.
  // PART/NPARTS: only the vectors i with i % NPARTS == PART (the k-loop spreads the staging of the next
  // tile between the MFMA groups of the current one)
  template <int PART = 0, int NPARTS = 1>
  __device__ __forceinline__ static void store(float* __restrict__ lds, const float4 (&r)[NV], int tid, int act,
                                               bool use_drop, const DropCfg& drop, long drop_ld, int r0, int k0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i % NPARTS != PART) continue;
      int rr, kk;
      coords(tid + i * NTHREADS, rr, kk);
      float4 v = r[i];
      if (PRO) {
        if (use_drop) {
          const long slow = KMAJOR ? (long)(r0 + rr) : (long)(k0 + kk);
          const long fast = KMAJOR ? (long)(k0 + kk) : (long)(r0 + rr);
          const uint64_t id = (uint64_t)(slow * drop_ld + fast);
          float s[4];
          if (VEC) {
            drop_scale4(drop, id, s);            // id % 4 == 0: host checks drop_ld % 4 == 0
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] = drop_scale1(drop, id + j);
          }
          v.x *= s[0]; v.y *= s[1]; v.z *= s[2]; v.w *= s[3];
        }
        if (act == RPDE_ACT_GELU) {
          v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
        } else if (act == RPDE_ACT_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
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

// PRO: 0 no staged activation, 1 on A, 2 on B.  BK: k-extent of a stage; STAGES: LDS stages (2 = double
// buffered, one barrier per stage; 1 = single buffer, two barriers, half the LDS -> more resident waves)
// Which C tile and which (batch entry, K slab) this workgroup owns.  Workgroups go round-robin to the 8 XCDs
// in dispatch order and each XCD has its own L2, so tiles that re-read the same operand rows should share an
// XCD:  swz = 1 (many M-tiles): the N-tiles of one M-tile sit 8 apart in the linear id;
//       swz = 2 (few tiles, many K slabs / batch entries): the T tiles of one slab sit 8 apart, so each
//       slab's operands cross the fabric once instead of once per tile (measured with FETCH_SIZE: the 2x2-tile
//       weight gradient read its operands twice, 8.6 GB per launch instead of 4.3).
__device__ __forceinline__ bool tile_coords(const GemmK& g, int& mt, int& nt, int& zz) {
  const int L = blockIdx.x;
  zz = blockIdx.z * gridDim.y + blockIdx.y;
  if (g.swz == 1) {
    mt = (L / (8 * g.ntiles)) * 8 + (L & 7);
    nt = (L >> 3) % g.ntiles;
    if (mt >= g.mtiles) return false;
  } else if (g.swz == 2) {
    const int T = g.mtiles * g.ntiles;                 // = gridDim.x, gridDim.z == 1, ztotal % 8 == 0
    const long lin = (long)blockIdx.y * T + L;
    const int grp = (int)(lin / (8 * T)), r = (int)(lin - (long)grp * 8 * T);
    zz = grp * 8 + (r & 7);
    const int tile = r >> 3;
    mt = tile / g.ntiles;
    nt = tile - mt * g.ntiles;
  } else {
    mt = L / g.ntiles;
    nt = L - mt * g.ntiles;
  }
  return zz < g.ztotal;
}

template <int WM, int WN, int TM, int TN, bool AK, bool BKM, int PRO, bool VEC, int BK = 32, int STAGES = 2, bool PIPE = false,
          bool PREAUX = false>
__global__ __launch_bounds__(NTHREADS, PREAUX ? 3 : 1) void gemm_f32_kernel(const GemmK g) {
  const DropCfg gdrop = drop_resolve(g.drop);      // (device-side mask counter folded in: rpde_internal.h)
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  static_assert(WM * WN == 4, "four waves per workgroup");
  using TA = Tile<BM, BK, AK, VEC, PRO == 1>;
  using TB = Tile<BN, BK, BKM, VEC, PRO == 2>;
  constexpr int STAGE_FLOATS = TA::LDS_FLOATS + TB::LDS_FLOATS;
  constexpr int SMEM_FLOATS = STAGES * STAGE_FLOATS;
  // the epilogue stages the C tile through the same LDS in EP row-slabs
  constexpr int EP = (BM * BN + SMEM_FLOATS - 1) / SMEM_FLOATS;
  static_assert(EP == 1 || EP == 2 || EP == 4, "C tile needs too many epilogue passes");
  static_assert((BM / 32) % EP == 0, "epilogue slabs must be whole 32-row tiles");
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  float* const As0 = smem;
  float* const As1 = STAGES == 2 ? smem + TA::LDS_FLOATS : smem;
  float* const Bs0 = smem + STAGES * TA::LDS_FLOATS;
  float* const Bs1 = STAGES == 2 ? Bs0 + TB::LDS_FLOATS : Bs0;

  const int tid = threadIdx.x;
  // ---- which tile, which batch entry, which K slice -------------------------
  int mt, nt, zz;
  if (!tile_coords(g, mt, nt, zz)) return;
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

  // PREAUX (short reductions, HBM/latency bound): fetch the rows the epilogue will need (stored derivative
  // `aux` for RPDE_EPI_MULAUX, old C for accumulate) now, so they fly during the operand loads and MFMAs
  constexpr int PA_SMEM = STAGES * (Tile<BM, BK, AK, VEC, false>::LDS_FLOATS + Tile<BN, BK, BKM, VEC, false>::LDS_FLOATS);
  constexpr int PA_EP = (BM * BN + PA_SMEM - 1) / PA_SMEM;
  constexpr int PA_SLAB = BM / PA_EP, PA_VPR = BN / 4, PA_RSTEP = NTHREADS / PA_VPR;
  constexpr int PA_NV4 = PA_SLAB * BN / 4 / NTHREADS;
  float4 pre[PREAUX ? PA_EP * PA_NV4 : 1];
  if constexpr (PREAUX) {
    const float* __restrict__ src = g.accumulate ? (g.acc_src ? g.acc_src + coff : C) : g.aux + coff;
    const long lds_ = g.accumulate ? g.ldc : g.ldaux;
    const int c4p = (tid % PA_VPR) * 4, row0p = tid / PA_VPR;
    const int gnp = min(n0 + c4p, g.N - 4);
#pragma unroll
    for (int e = 0; e < PA_EP; ++e)
#pragma unroll
      for (int it = 0; it < PA_NV4; ++it) {
        const int gm = min(m0 + e * PA_SLAB + row0p + it * PA_RSTEP, g.M - 1);
        pre[e * PA_NV4 + it] = *reinterpret_cast<const float4*>(src + (long)gm * lds_ + gnp);
      }
  }

  const bool drop_a = gdrop.on() && (g.drop_where & 1);
  const bool drop_b = gdrop.on() && (g.drop_where & 2);
  const bool drop_e = gdrop.on() && (g.drop_where & 4);
  // ---- main loop (PIPE = false): register prefetch one tile ahead, stage after the MFMA block ----
  if constexpr (!PIPE) {
    float4 ra[TA::NV], rb[TB::NV];
    if (nkt > 0) {
      TA::load(ra, A, g.lda, m0, g.M, kbeg, kend, tid);
      TB::load(rb, B, g.ldb, n0, g.N, kbeg, kend, tid);
      TA::store(As0, ra, tid, g.act_a, drop_a, gdrop, g.drop_ld, m0, kbeg);
      TB::store(Bs0, rb, tid, g.act_b, drop_b, gdrop, g.drop_ld, n0, kbeg);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      const float* as = (kt & 1) ? As1 : As0;
      const float* bs = (kt & 1) ? Bs1 : Bs0;
      const bool more = kt + 1 < nkt;
      const int knext = kbeg + (kt + 1) * BK;
      if (more) {  // issue the next stage's HBM loads before computing this one
        TA::load(ra, A, g.lda, m0, g.M, knext, kend, tid);
        TB::load(rb, B, g.ldb, n0, g.N, knext, kend, tid);
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
      if (STAGES == 1) __syncthreads();
      if (more) {
        TA::store((kt & 1) ? As0 : As1, ra, tid, g.act_a, drop_a, gdrop, g.drop_ld, m0, knext);
        TB::store((kt & 1) ? Bs0 : Bs1, rb, tid, g.act_b, drop_b, gdrop, g.drop_ld, n0, knext);
      }
      __syncthreads();
    }
  } else {
  // ---- main loop, software-pipelined two tiles deep ---------------------------------------------
  // tile t lives in register set t&1 and LDS stage t&1.  Iteration kt: issue the HBM loads of tile kt+2
  // (into the set that held tile kt), run the MFMAs of tile kt from LDS, and between the MFMA groups
  // stage tile kt+1 (activation + LDS writes) from the other set -- its loads were issued a whole
  // iteration ago, and its VALU work hides under this tile's 64-cycle MFMAs.
  float4 ra0[TA::NV], rb0[TB::NV], ra1[TA::NV], rb1[TB::NV];
  constexpr int NQ = BK / 8;        // MFMA groups (4 k-steps each) per tile
  if (nkt > 0) {
    TA::load(ra0, A, g.lda, m0, g.M, kbeg, kend, tid);
    TB::load(rb0, B, g.ldb, n0, g.N, kbeg, kend, tid);
    TA::load(ra1, A, g.lda, m0, g.M, kbeg + BK, kend, tid);
    TB::load(rb1, B, g.ldb, n0, g.N, kbeg + BK, kend, tid);
    TA::store(As0, ra0, tid, g.act_a, drop_a, gdrop, g.drop_ld, m0, kbeg);
    TB::store(Bs0, rb0, tid, g.act_b, drop_b, gdrop, g.drop_ld, n0, kbeg);
  }
  __syncthreads();

#define RPDE_GEMM_STEP(RA_FREE, RB_FREE, RA_NEXT, RB_NEXT, AS_CUR, BS_CUR, AS_NEXT, BS_NEXT)                          \
  {                                                                                                                 \
    const bool more = kt + 1 < nkt;                                                                                 \
    const int knext = kbeg + (kt + 1) * BK;                                                                         \
    /* unconditional: past the end the loads are predicated to the base address (VEC) -> countable vmcnt */       \
    TA::load(RA_FREE, A, g.lda, m0, g.M, knext + BK, kend, tid);                                                    \
    TB::load(RB_FREE, B, g.ldb, n0, g.N, knext + BK, kend, tid);                                                    \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                                \
      float af[TM][4], bf[TN][4];                                                                                   \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) TA::frag(AS_CUR, (wm * TM + i) * 32 + l31, lh, q, af[i]);      \
      _Pragma("unroll") for (int j = 0; j < TN; ++j) TB::frag(BS_CUR, (wn * TN + j) * 32 + l31, lh, q, bf[j]);      \
      _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                              \
          _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                            \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);               \
      if (STAGES == 2 && more) {                                                                                    \
        if (q == 0) { TA::template store<0, NQ>(AS_NEXT, RA_NEXT, tid, g.act_a, drop_a, gdrop, g.drop_ld, m0, knext);   \
                      TB::template store<0, NQ>(BS_NEXT, RB_NEXT, tid, g.act_b, drop_b, gdrop, g.drop_ld, n0, knext); } \
        if (q == 1) { TA::template store<1, NQ>(AS_NEXT, RA_NEXT, tid, g.act_a, drop_a, gdrop, g.drop_ld, m0, knext);   \
                      TB::template store<1, NQ>(BS_NEXT, RB_NEXT, tid, g.act_b, drop_b, gdrop, g.drop_ld, n0, knext); } \
        if (NQ > 2 && q == 2) { TA::template store<2 % NQ, NQ>(AS_NEXT, RA_NEXT, tid, g.act_a, drop_a, gdrop, g.drop_ld, m0, knext);   \
                      TB::template store<2 % NQ, NQ>(BS_NEXT, RB_NEXT, tid, g.act_b, drop_b, gdrop, g.drop_ld, n0, knext); } \
        if (NQ > 3 && q == 3) { TA::template store<3 % NQ, NQ>(AS_NEXT, RA_NEXT, tid, g.act_a, drop_a, gdrop, g.drop_ld, m0, knext);   \
                      TB::template store<3 % NQ, NQ>(BS_NEXT, RB_NEXT, tid, g.act_b, drop_b, gdrop, g.drop_ld, n0, knext); } \
      }                                                                                                             \
    }                                                                                                               \
    if (STAGES == 1) {                                                                                              \
      __syncthreads();                                                                                              \
      if (more) {                                                                                                   \
        TA::store(AS_NEXT, RA_NEXT, tid, g.act_a, drop_a, gdrop, g.drop_ld, m0, knext);                            \
        TB::store(BS_NEXT, RB_NEXT, tid, g.act_b, drop_b, gdrop, g.drop_ld, n0, knext);                            \
      }                                                                                                             \
    }                                                                                                               \
    __syncthreads();                                                                                                \
  }

  for (int kt = 0; kt < nkt;) {
    RPDE_GEMM_STEP(ra0, rb0, ra1, rb1, As0, Bs0, As1, Bs1)      // even tile: compute stage 0, stage tile kt+1 from set 1
    if (++kt >= nkt) break;
    RPDE_GEMM_STEP(ra1, rb1, ra0, rb0, As1, Bs1, As0, Bs0)      // odd tile
    ++kt;
  }
#undef RPDE_GEMM_STEP

  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (VEC && g.cvec) {
    // Stage the tile through LDS (the operand buffers are free after the loop's
    // last barrier) so that HBM sees whole 16-byte vectors of contiguous rows:
    // 4x fewer store / aux-load instructions, one dropout hash per 4 elements,
    // and the per-tile column sums (bias gradients) come for free.
    float* __restrict__ cs = smem;
    constexpr int SLAB = BM / EP;                     // rows per epilogue pass
    constexpr int VPR = BN / 4;                       // vectors per tile row
    constexpr int NV4 = SLAB * BN / 4 / NTHREADS;     // vectors per thread per pass
    constexpr int RSTEP = NTHREADS / VPR;             // rows between a thread's vectors
    static_assert(NV4 >= 1, "epilogue slab too small");
    const int c4 = (tid % VPR) * 4, row0 = tid / VPR;
    const int gn = n0 + c4;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bn4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g.bias_mode == 1 && gn < g.N) bn4 = *reinterpret_cast<const float4*>(g.bias + gn);
    const float* __restrict__ aux = g.aux ? g.aux + coff : nullptr;
#pragma unroll
    for (int e = 0; e < EP; ++e) {
      if (e > 0) __syncthreads();                     // previous slab fully consumed
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int rt = (wm * TM + i) * 32;            // first row of this 32-row tile inside the block
        if (rt / SLAB != e) continue;                 // wave-uniform
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = (wn * TN + j) * 32 + l31;
          const int rb = rt - e * SLAB + 4 * lh;
#pragma unroll
          for (int r = 0; r < 16; ++r) cs[(rb + (r & 3) + 8 * (r >> 2)) * BN + col] = acc[i][j][r];
        }
      }
      __syncthreads();
      if (gn < g.N) {
#pragma unroll
        for (int it = 0; it < NV4; ++it) {
          const int row = row0 + it * RSTEP;
          const int gm = m0 + e * SLAB + row;
          if (gm >= g.M) break;
          float4 v = *reinterpret_cast<const float4*>(cs + row * BN + c4);
          v.x = fmaf(v.x, g.alpha, bn4.x); v.y = fmaf(v.y, g.alpha, bn4.y);
          v.z = fmaf(v.z, g.alpha, bn4.z); v.w = fmaf(v.w, g.alpha, bn4.w);
          if (g.bias_mode == 2) { const float bm = g.bias[gm]; v.x += bm; v.y += bm; v.z += bm; v.w += bm; }
          if (g.epi_dact == RPDE_EPI_MULAUX) {
            float4 a;
            if constexpr (PREAUX) a = pre[e * NV4 + it];
            else a = *reinterpret_cast<const float4*>(aux + (long)gm * g.ldaux + gn);
            v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w;
          } else if (g.epi_dact) {
            float s[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop_e) drop_scale4(gdrop, (uint64_t)((long)gm * g.drop_ld + gn), s);
            const float4 a = *reinterpret_cast<const float4*>(aux + (long)gm * g.ldaux + gn);
            v.x *= dact_f(g.epi_dact, a.x * s[0]) * s[0];
            v.y *= dact_f(g.epi_dact, a.y * s[1]) * s[1];
            v.z *= dact_f(g.epi_dact, a.z * s[2]) * s[2];
            v.w *= dact_f(g.epi_dact, a.w * s[3]) * s[3];
          }
          float4* cp = reinterpret_cast<float4*>(C + (long)gm * g.ldc + gn);
          if (g.accumulate) {
            float4 o;
            if constexpr (PREAUX) o = pre[e * NV4 + it];
            else o = g.acc_src ? *reinterpret_cast<const float4*>(g.acc_src + coff + (long)gm * g.ldc + gn) : *cp;
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          if (g.write_act) {
            float s[4] = {1.f, 1.f, 1.f, 1.f};
            if (drop_e && !g.epi_dact) drop_scale4(gdrop, (uint64_t)((long)gm * g.drop_ld + gn), s);
            v.x *= s[0]; v.y *= s[1]; v.z *= s[2]; v.w *= s[3];
            if (g.aux_out) {
              float4 dv;
              act_both(g.write_act, v.x, v.x, dv.x); act_both(g.write_act, v.y, v.y, dv.y);
              act_both(g.write_act, v.z, v.z, dv.z); act_both(g.write_act, v.w, v.w, dv.w);
              dv.x *= s[0]; dv.y *= s[1]; dv.z *= s[2]; dv.w *= s[3];
              *reinterpret_cast<float4*>(g.aux_out + coff + (long)gm * g.ldc + gn) = dv;
            } else {
              v.x = act_f(g.write_act, v.x); v.y = act_f(g.write_act, v.y);
              v.z = act_f(g.write_act, v.z); v.w = act_f(g.write_act, v.w);
            }
          }
          *cp = v;
          csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w;
        }
      }
    }
    if (g.colsum) {      // uniform
      __syncthreads();   // everyone is done reading cs
      *reinterpret_cast<float4*>(cs + row0 * BN + c4) = csum;
      __syncthreads();
      if (tid < BN && n0 + tid < g.N) {
        float t = 0.f;
#pragma unroll
        for (int rr = 0; rr < RSTEP; ++rr) t += cs[rr * BN + tid];
        g.colsum[(long)mt * g.N + n0 + tid] = t;
      }
    }
    return;
  }
  const bool plain = !g.epi_dact && !g.accumulate && !g.write_act;
  const long ldc = g.ldc;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + l31;
      if (n >= g.N) continue;
      const float bn = g.bias_mode == 1 ? g.bias[n] : 0.f;
      const int mb = m0 + (wm * TM + i) * 32 + 4 * lh;
      float* __restrict__ cb = C + (long)mb * ldc + n;
      const float* __restrict__ ab = g.aux ? g.aux + coff + (long)mb * g.ldaux + n : nullptr;
      const uint64_t idb = (uint64_t)((long)mb * g.drop_ld + n);
#pragma unroll 4
      for (int r = 0; r < 16; ++r) {
        const int dm = (r & 3) + 8 * (r >> 2);
        if (mb + dm >= g.M) continue;
        float v = fmaf(acc[i][j][r], g.alpha, bn);
        if (g.bias_mode == 2) v += g.bias[mb + dm];
        if (!plain) {
          if (g.epi_dact == RPDE_EPI_MULAUX) {
            v *= ab[dm * g.ldaux];
          } else if (g.epi_dact) {
            float s = 1.f;
            if (drop_e) s = drop_scale1(gdrop, idb + (uint64_t)(dm * g.drop_ld));
            const float u = ab[dm * g.ldaux] * s;
            v = v * dact_f(g.epi_dact, u) * s;
          }
          if (g.accumulate) v += g.acc_src ? g.acc_src[coff + (long)(mb + dm) * ldc + n] : cb[dm * ldc];
          if (g.write_act) {
            float s = 1.f;
            if (drop_e && !g.epi_dact) s = drop_scale1(gdrop, idb + (uint64_t)(dm * g.drop_ld));
            v *= s;
            if (g.aux_out) g.aux_out[coff + (long)(mb + dm) * ldc + n] = dact_f(g.write_act, v) * s;
            v = act_f(g.write_act, v);
          }
        }
        cb[dm * ldc] = v;
      }
    }
  }
}

// ---- host-side dispatch, instantiated once per operand layout (one translation
// unit each, so the instantiations compile in parallel).  PM: bit p set <=> the
// staged-activation variant PRO = p is built for this layout.
template <int WM, int WN, int TM, int TN, bool AK, bool BKM, int PM, bool VEC, int BK = 32, int STAGES = 2, bool PIPE = false,
          bool PREAUX = false>
inline void launch_pro(const GemmK& g, int pro, dim3 grid, hipStream_t st) {
  if (pro == 0) {
    if constexpr ((PM & 1) != 0)
      hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, AK, BKM, 0, VEC, BK, STAGES, PIPE, PREAUX>), grid, dim3(NTHREADS), 0, st, g);
  } else if (pro == 1) {
    if constexpr ((PM & 2) != 0)
      hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, AK, BKM, 1, VEC, BK, STAGES, PIPE, PREAUX>), grid, dim3(NTHREADS), 0, st, g);
  } else {
    if constexpr ((PM & 4) != 0)
      hipLaunchKernelGGL((gemm_f32_kernel<WM, WN, TM, TN, AK, BKM, 2, VEC, BK, STAGES, PIPE, PREAUX>), grid, dim3(NTHREADS), 0, st, g);
  }
}

int gemm_variant();   // experiment switch (env RPDE_GEMM_VARIANT) for the 128 x 128 tile

template <bool AK, bool BKM, int PM>
inline int launch_layout_impl(const GemmK& g, int bm, int bn, int pro, bool vec, dim3 grid, hipStream_t st) {
  if (!((PM >> pro) & 1)) {
    set_error("gemm: staged activation on operand %d is not built for layout a_kmajor=%d b_kmajor=%d", pro, (int)AK, (int)BKM);
    return RPDE_ERR_ARG;
  }
  // RPDE_GEMM_VARIANT (A/B experiments, same-box numbers in profiles/r01_c_gemm_variants.txt):
  //   0 default: two-deep software pipeline on every vector tile;  1: lean loop on the small tiles;
  //   2: lean loop everywhere;  3: no lean single-stage kernel for short reductions;  4: no epilogue prefetch.  (Smaller-LDS variants -- BK16 x 2, BK32 x 1 -- were within 2 %: the
  //   resident-wave count is not what limits this kernel.)
  const int v = gemm_variant();
  const bool pipe_big = v != 2, pipe_small = v != 1 && v != 2;
  if (!vec) launch_pro<2, 2, 1, 1, AK, BKM, PM, false>(g, pro, grid, st);            // 64 x 64, scalar loads
  else if (bm == 128 && bn == 128) {
    // reductions of <= 2 stages are epilogue / HBM-latency bound: lean single-stage kernel (~90 VGPRs,
    // 37 KB LDS -> 4 workgroups per CU) keeps more memory requests in flight  (variant 3 disables)
    const bool preaux = g.cvec && pro == 0 && v != 4 && !g.write_act &&
                        ((g.epi_dact == RPDE_EPI_MULAUX) != (g.accumulate != 0));   // exactly one of the two
    if (g.kchunk <= 64 && v != 3 && v != 2) {
      if constexpr ((PM & 1) != 0) {
        if (preaux) launch_pro<2, 2, 2, 2, AK, BKM, 1, true, 32, 1, false, true>(g, pro, grid, st);
        else launch_pro<2, 2, 2, 2, AK, BKM, PM, true, 32, 1, false>(g, pro, grid, st);
      }
    }
    else if (pipe_big) launch_pro<2, 2, 2, 2, AK, BKM, PM, true, 32, 2, true>(g, pro, grid, st);
    else launch_pro<2, 2, 2, 2, AK, BKM, PM, true>(g, pro, grid, st);
  } else if (bm == 128 && bn == 64) {
    if (pipe_small) launch_pro<4, 1, 1, 2, AK, BKM, PM, true, 32, 2, true>(g, pro, grid, st);
    else launch_pro<4, 1, 1, 2, AK, BKM, PM, true>(g, pro, grid, st);
  } else if (bm == 64 && bn == 128) {
    if (pipe_small) launch_pro<1, 4, 2, 1, AK, BKM, PM, true, 32, 2, true>(g, pro, grid, st);
    else launch_pro<1, 4, 2, 1, AK, BKM, PM, true>(g, pro, grid, st);
  } else if (bm == 64 && bn == 64) {
    if (pipe_small) launch_pro<2, 2, 1, 1, AK, BKM, PM, true, 32, 2, true>(g, pro, grid, st);
    else launch_pro<2, 2, 1, 1, AK, BKM, PM, true>(g, pro, grid, st);
  } else if (bm == 128 && bn == 32) launch_pro<4, 1, 1, 1, AK, BKM, PM, true>(g, pro, grid, st);
  else launch_pro<1, 4, 1, 1, AK, BKM, PM, true>(g, pro, grid, st);                   // 32 x 128
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// split-bf16 path (gemm_bf16x3.hip)
bool bf16x3_supports(int bm, int bn);
int split_npad(int N);
size_t split_bytes(int N, int K);
int split_weights(const float* w, int kmajor, long ld, int N, int K, void* out, hipStream_t st);
int launch_bf16x3(const GemmK& g, int bm, int bn, bool ak, bool bk, dim3 grid, hipStream_t st);

int launch_nt(const GemmK& g, int bm, int bn, int pro, bool vec, dim3 grid, hipStream_t st);  // A k-major, B k-major
int launch_nn(const GemmK& g, int bm, int bn, int pro, bool vec, dim3 grid, hipStream_t st);  // A k-major, B x-major
int launch_tn(const GemmK& g, int bm, int bn, int pro, bool vec, dim3 grid, hipStream_t st);  // A x-major, B x-major
int launch_tt(const GemmK& g, int bm, int bn, int pro, bool vec, dim3 grid, hipStream_t st);  // A x-major, B k-major

}  // namespace rpde
