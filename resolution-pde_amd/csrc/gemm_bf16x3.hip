// fp32 GEMM on the bf16 matrix pipe by exact-product splitting.
//
// On gfx950 v_mfma_f32_32x32x2_f32 runs at the vector fp32 rate and (measured) does
// not overlap with VALU work: it is the slow path for fp32 matrix products.  The bf16 MFMA pipe is
// separate and 16x faster per flop.  Every fp32 value is the exact sum of three bf16 values,
//      a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)      (3 x 8 = 24 mantissa bits),
// and a bf16 x bf16 product is exact in fp32, so
//      a*b = a1b1 + (a1b2 + a2b1) + (a1b3 + a3b1 + a2b2) + O(2^-24 |ab|)
// -- six v_mfma_f32_32x32x16_bf16 per fp32 tile product, accumulated in fp32.  Inputs, outputs and
// accumulation stay fp32; the dropped terms are below the fp32 rounding of the product itself, so the
// result has fp32-GEMM accuracy (tests: <= 2e-6 rel-L2 against float64, same bound as the fp32-MFMA kernel;
// every golden-vector parity test runs through this kernel).  Effective rate: 16/6 = 2.7x the fp32 MFMA
// peak, which turns the FeedForward GEMMs from ALU bound into HBM bound.
//
// Structure: 256 threads = 4 waves (WM x WN) x (TM x TN) tiles of 32x32, BK = 32 per stage, one LDS stage
// + register prefetch (two barriers per stage, 48 KB LDS -> 3 workgroups per CU).  The splitting happens
// while a tile moves registers -> LDS; LDS holds three bf16 images per operand as [row][32 k] = 64-byte
// rows whose 16-byte chunks are XOR-swizzled with (row>>2)&3, so the ds_read_b128 fragment reads
// (lane = row, 8 consecutive k) are bank-conflict free without padding.  x-major operands (reduction
// index slow in memory: weight gradients, channels-last fields) keep their memory order in LDS and are
// transposed by the read: ds_read_b64_tr_b16 (see XTile).
// The epilogue is the fp32 kernel's (LDS-staged 16-byte rows, bias / activation+derivative / multiply by
// stored derivative / accumulate / per-tile column sums).
#include "gemm_kernel.h"

#include <stdlib.h>

namespace rpde {

// RPDE_STAMPS (debug builds only: rpde/build.py --stamps): wave 0 of 64 mid-launch workgroups records
// s_memtime at its phase boundaries; rpde_debug_stamps() copies them out.  profiles/stamps.py prints them.
#ifdef RPDE_STAMPS
__device__ unsigned long long g_stamps[64 * 32];
#define STAMP(i) do { if (stamp_on) g_stamps[stamp_slot * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define STAMP_RT(i) do { if (stamp_on) g_stamps[stamp_slot * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP_RT(i) do { } while (0)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int XBK = 32;

// three bf16 pieces of two floats, packed pairwise (low half = first element).
// Scalar subtractions on purpose (and -fno-slp-vectorize in the build): packed fp32 VALU (v_pk_add_f32 ...)
// does not overlap with MFMAs of the other waves on the SIMD, ordinary VALU does
// (profiles/ubench/overlap.hip: v_fma_f32 beside an MFMA wave costs 30 % of the MFMA time, v_pk_fma_f32 100 %).
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  union { bf16x2 v; unsigned u; } p;
  p.v = __builtin_convertvector((f32x2_){a, b}, bf16x2);
  return p.u;
}
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  const float q0 = r0 - __uint_as_float(m << 16), q1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(q0, q1);
}

// byte offset of element (row, k) inside one [ROWS][32] bf16 image
__device__ __forceinline__ int img_off(int row, int k) {
  return row * 64 + (((k >> 3) ^ ((row >> 2) & 3)) << 4) + ((k & 7) << 1);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));

// One operand stage in LDS: three bf16 images of ROWS x 32(k).
//  k-major operand (k contiguous in memory): image rows are the operand's rows, [row][32 k] = 64-byte rows,
//    16-byte chunks XOR-swizzled with (row>>2)&3; fragments are ds_read_b128 (lane = row, 8 consecutive k).
//  x-major operand (k slow in memory: weight gradients, spectral fields): the image keeps the memory
//    order, [k][ROWS] bf16 rows of 2*ROWS bytes, and the fragments come out of gfx950's transposing read
//    ds_read_b64_tr_b16 (each group of 16 lanes fetches a 4(k) x 16(row) block and receives it column
//    major), so the store side stays a plain vector write.  8-byte chunks are XOR-swizzled so that the four
//    k-rows a half-wave reads land on different banks.
template <int ROWS, bool KMAJOR, int NT = NTHREADS>
struct XTile {
  static constexpr int NV = ROWS * XBK / 4 / NT;           // float4 per thread
  static constexpr int IMG_BYTES = ROWS * 64;              // one bf16 image
  static constexpr int CPR = ROWS / 4;                     // x-major: 8-byte chunks per k-row
  static_assert(NV >= 1 && (ROWS == 64 || ROWS == 128), "unsupported tile");

  __device__ __forceinline__ static int xswz(int k) { return ROWS == 128 ? ((k & 3) << 3) : (((k >> 1) & 1) << 3); }
  // byte offset of 8-byte chunk c8 (operand rows 4 c8 .. 4 c8 + 3) of k-row k inside an x-major image
  __device__ __forceinline__ static int xoff(int k, int c8) { return k * (2 * ROWS) + ((c8 ^ xswz(k)) << 3); }

  // vector i of this thread covers: k-major: row rr, k = kk..kk+3;  x-major: rows rr..rr+3 at k = kk
  __device__ __forceinline__ static void coords(int tid, int i, int& rr, int& kk) {
    const int v = tid + i * NT;
    if (KMAJOR) { rr = v >> 3; kk = (v & 7) << 2; }
    else { kk = v / CPR; rr = (v % CPR) << 2; }
  }

  // Loop-invariant 32-bit element offsets of this thread's vectors relative to the tile origin
  // (k-major: A + r0*ld + k0;  x-major: B + k0*ld + r0).  Rows beyond the matrix are clamped to the last
  // valid row: they only feed output rows / columns that are never stored, so the k-loop needs no
  // predication at all (the host dispatches here only when K % 32 == 0).
  __device__ __forceinline__ static void prep(int (&off)[NV], long ld, int r0, int rmax, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int rr, kk;
      coords(tid, i, rr, kk);
      if (KMAJOR) off[i] = min(rr, rmax - 1 - r0) * (int)ld + kk;
      else off[i] = kk * (int)ld + min(rr, rmax - 4 - r0);
    }
  }
  __device__ __forceinline__ static void load(float4 (&r)[NV], const float* __restrict__ tile, const int (&off)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) r[i] = *reinterpret_cast<const float4*>(tile + off[i]);
  }

  // x-major only: last stage of a K that is not a multiple of 32 -- k-rows past `valid` re-read row valid-1
  // (the other operand is a zero-padded pre-split image there, so they contribute exact zeros)
  __device__ __forceinline__ static void load_tail(float4 (&r)[NV], const float* __restrict__ tile, const int (&off)[NV],
                                                   int ld, int valid, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int rr, kk;
      coords(tid, i, rr, kk);
      r[i] = *reinterpret_cast<const float4*>(tile + off[i] - max(kk - (valid - 1), 0) * ld);
    }
  }

  // split into three bf16 images (hi, mid, lo at lds, lds + IMG, lds + 2*IMG)
  __device__ __forceinline__ static void store(char* __restrict__ lds, const float4 (&r)[NV], int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int rr, kk;
      coords(tid, i, rr, kk);
      unsigned h0, m0, l0, h1, m1, l1;
      split2(r[i].x, r[i].y, h0, m0, l0);
      split2(r[i].z, r[i].w, h1, m1, l1);
      char* p = lds + (KMAJOR ? img_off(rr, kk) : xoff(kk, rr >> 2));
      *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(p + IMG_BYTES) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(p + 2 * IMG_BYTES) = make_uint2(l0, l1);
    }
  }

  // per-lane byte offset (inside an image) of the fragment of 32-row tile `t` at k-step 0; fragments of
  // the other k-step / images sit at compile-time distances from it
  __device__ __forceinline__ static int frag_base(int t, int lane) {
    const int l31 = lane & 31, lh = lane >> 5;
    if (KMAJOR) return (t * 32 + l31) * 64;             // the chunk is picked per k-step in frag()
    // x-major: lane 4q+p of each 16-lane group addresses k-row q, operand rows 4p..4p+3 of the group's 16
    const int g = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3;
    return xoff(8 * lh + q, t * 8 + 4 * g + p);
  }
  // 8 consecutive k (k = 16 s + 8 lh + j) of the lane's row, image `which`
  __device__ __forceinline__ static bf16x8 frag(const char* __restrict__ lds, int base, int which, int row, int lh, int s) {
    if (KMAJOR) {
      const int chunk = (2 * s + lh) ^ ((row >> 2) & 3);
      return *reinterpret_cast<const bf16x8*>(lds + which * IMG_BYTES + base + (chunk << 4));
    }
    // k-rows 16 s + 8 lh + q (+4): same swizzle class as the base row, so plain byte distances
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const char* p = lds + which * IMG_BYTES + base + s * 16 * (2 * ROWS);
    union { struct { s16x4 a, b; } h; bf16x8 v; } u;
    u.h.a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
    u.h.b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * (2 * ROWS)));
    return u.v;
  }
};

// one stage of a pre-split operand: NBI 16-byte chunks per image per thread, straight copies
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int NBI, int NT>
__device__ __forceinline__ void bimg_load(u32x4 (&r)[3 * NBI], const char* __restrict__ p, long img_stride) {
#pragma unroll
  for (int i = 0; i < 3 * NBI; ++i)
    r[i] = *reinterpret_cast<const u32x4*>(p + (i / NBI) * img_stride + (i % NBI) * (NT * 16));
}
template <int NBI, int NT>
__device__ __forceinline__ void bimg_store(char* __restrict__ lds, const u32x4 (&r)[3 * NBI], int img_bytes) {
#pragma unroll
  for (int i = 0; i < 3 * NBI; ++i)
    *reinterpret_cast<u32x4*>(lds + (i / NBI) * img_bytes + (i % NBI) * (NT * 16)) = r[i];
}

// BIMG: the B operand arrives pre-split (rpde_split_weights: [K/32][3][Npad][32] bf16 images whose rows are
// already chunk-swizzled), so a B stage is a straight 16-byte copy into LDS and costs no VALU work.  Every
// workgroup needs the same weight tile, so splitting it once per call instead of once per workgroup
// removes half of the loop's vector instructions (MFMA and VALU issue serially on a SIMD, measured in
// profiles/ubench/overlap.hip).
// AIMG: the same for the A operand (the DFT tables of the spectral layers, split once per plan); with it
// K may have a tail (the images are zero-padded to a multiple of 32) as long as B is x-major.
template <int WM, int WN, int TM, int TN, bool AK, bool BKM, bool BIMG = false, bool AIMG = false>
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 8 ? 4 : 3) void gemm_bf16x3_kernel(const GemmK g) {
  const DropCfg gdrop = drop_resolve(g.drop);      // (device-side mask counter folded in: rpde_internal.h)
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int NT = 64 * WM * WN;          // 4 waves (3 workgroups per CU) or 8 waves (2 per CU = 4 waves per SIMD)
  static_assert(WM * WN == 4 || WM * WN == 8, "four or eight waves per workgroup");
  static_assert(!(AIMG && BIMG), "one pre-split operand per product");
  using TA = XTile<BM, AK || AIMG, NT>;
  using TB = XTile<BN, BKM, NT>;
  constexpr int SMEM_BYTES = 3 * (TA::IMG_BYTES + TB::IMG_BYTES);
  constexpr int SMEM_FLOATS = SMEM_BYTES / 4;
  constexpr int EP = (BM * BN + SMEM_FLOATS - 1) / SMEM_FLOATS;
  static_assert(EP == 1 || EP == 2 || EP == 4, "C tile needs too many epilogue passes");
  static_assert((BM / 32) % EP == 0, "epilogue slabs must be whole 32-row tiles");
  __shared__ __attribute__((aligned(16))) char smem_raw[SMEM_BYTES];
  char* const As = smem_raw;
  char* const Bs = smem_raw + 3 * TA::IMG_BYTES;

  const int tid = threadIdx.x;
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
  const int nkt = (kend - kbeg + XBK - 1) / XBK;

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lh = lane >> 5;
#ifdef RPDE_STAMPS
  const long stamp_lin = (long)blockIdx.y * gridDim.x + blockIdx.x;
  const long stamp_first = ((long)gridDim.x * gridDim.y / 2) & ~63L;
  const bool stamp_on = tid == 0 && stamp_lin >= stamp_first && stamp_lin < stamp_first + 64;
  const int stamp_slot = (int)(stamp_lin - stamp_first);
#endif
  STAMP(0);
  STAMP_RT(31);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int NBI = BIMG ? BN * 4 / NT : (AIMG ? BM * 4 / NT : 1);   // 16-byte chunks per image per thread
  static_assert(!BIMG || (BKM && NBI >= 1), "pre-split B is k-major");
  constexpr bool IMG = AIMG || BIMG;
  int oa[TA::NV], ob[TB::NV];
  if (!AIMG) TA::prep(oa, g.lda, m0, g.M, tid);
  if (!BIMG) TB::prep(ob, g.ldb, n0, g.N, tid);
  // the pre-split operand (either one): rows r0.. of every stage are one contiguous block per image
  const char* __restrict__ bimg = g.Bimg + ((long)(kbeg / XBK) * 3 * g.npad + (AIMG ? m0 : n0)) * 64 + tid * 16;
  const long bimg_step = 3L * g.npad * 64;
  u32x4 rbi[3 * NBI];
  // K tail (AIMG with an x-major B only): number of valid k-rows of the last stage, 0 = full
  const int tail = (AIMG && !BKM) ? ((kend - kbeg) & (XBK - 1)) : 0;
  // uniform tile origins, advanced by one stage per iteration
  const float* __restrict__ at = AK ? A + (long)m0 * g.lda + kbeg : A + (long)kbeg * g.lda + m0;
  const float* __restrict__ bt = BKM ? B + (long)n0 * g.ldb + kbeg : B + (long)kbeg * g.ldb + n0;
  const long astep = AK ? (long)XBK : (long)XBK * g.lda;
  const long bstep = BKM ? (long)XBK : (long)XBK * g.ldb;

  int fa[TM], fb[TN];      // per-lane fragment offsets
#pragma unroll
  for (int i = 0; i < TM; ++i) fa[i] = TA::frag_base(wm * TM + i, lane);
#pragma unroll
  for (int j = 0; j < TN; ++j) fb[j] = TB::frag_base(wn * TN + j, lane);

  float4 ra[TA::NV], rb[TB::NV];
  if (nkt > 0) {
    if (!AIMG) TA::load(ra, at, oa);
    if (IMG) bimg_load<NBI, NT>(rbi, bimg, (long)g.npad * 64);
    if (!BIMG) {
      if (tail && nkt == 1) TB::load_tail(rb, bt, ob, (int)g.ldb, tail, tid); else TB::load(rb, bt, ob);
    }
  }
  // (a second register set fetching two stages ahead was measured slower here: the extra 32 VGPRs cost
  //  a resident wave, and three workgroups per CU already cover the load latency)
  STAMP(1);
  for (int kt = 0; kt < nkt; ++kt) {
    if (AIMG) bimg_store<NBI, NT>(As + tid * 16, rbi, TA::IMG_BYTES); else TA::store(As, ra, tid);
    if (BIMG) bimg_store<NBI, NT>(Bs + tid * 16, rbi, TB::IMG_BYTES); else TB::store(Bs, rb, tid);
    if (kt < 8) STAMP(2 + 3 * kt);
    __syncthreads();
    if (kt < 8) STAMP(3 + 3 * kt);
    if (kt + 1 < nkt) {
      at += astep; bt += bstep; bimg += bimg_step;
      if (!AIMG) TA::load(ra, at, oa);
      if (IMG) bimg_load<NBI, NT>(rbi, bimg, (long)g.npad * 64);
      if (!BIMG) {
        if (tail && kt + 2 == nkt) TB::load_tail(rb, bt, ob, (int)g.ldb, tail, tid); else TB::load(rb, bt, ob);
      }
    }
#ifndef RPDE_NO_SETPRIO
    __builtin_amdgcn_s_setprio(1);       // MFMA phase first: the other waves' vector work fits in its issue gaps
#endif
#pragma unroll
    for (int s = 0; s < XBK / 16; ++s) {
      // two rounds so that only 8 fragments (32 VGPRs) are live at a time: (hi, mid) of both operands first,
      // then the low images take over the registers of the mid ones
      bf16x8 a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        a0[i] = TA::frag(As, fa[i], 0, (wm * TM + i) * 32 + l31, lh, s);
        a1[i] = TA::frag(As, fa[i], 1, (wm * TM + i) * 32 + l31, lh, s);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        b0[j] = TB::frag(Bs, fb[j], 0, (wn * TN + j) * 32 + l31, lh, s);
        b1[j] = TB::frag(Bs, fb[j], 1, (wn * TN + j) * 32 + l31, lh, s);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b1[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b1[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[i], b0[j], c, 0, 0, 0);
          acc[i][j] = c;
        }
      // (fencing the rounds pays only where the fragment reads are the transposing ones: weight gradient +3 %,
      //  k-major kernels -2 %; same-box A/B)
      if constexpr (!AK && !BKM && !AIMG) __builtin_amdgcn_sched_barrier(0);
      bf16x8 a2[TM], b2[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a2[i] = TA::frag(As, fa[i], 2, (wm * TM + i) * 32 + l31, lh, s);
#pragma unroll
      for (int j = 0; j < TN; ++j) b2[j] = TB::frag(Bs, fb[j], 2, (wn * TN + j) * 32 + l31, lh, s);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b2[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[i], b0[j], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], c, 0, 0, 0);
          acc[i][j] = c;
        }
      // (fencing the rounds pays only where the fragment reads are the transposing ones: weight gradient +3 %,
      //  k-major kernels -2 %; same-box A/B)
      if constexpr (!AK && !BKM && !AIMG) __builtin_amdgcn_sched_barrier(0);
    }
#ifndef RPDE_NO_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();
    if (kt < 8) STAMP(4 + 3 * kt);
  }
  STAMP(26);
  STAMP_RT(25);   // (overwrites k7's last stamp: constant-rate clock at loop end, for the shader frequency)

  // ---- epilogue (same as the fp32 kernel's vector path; the host only dispatches here when g.cvec) ----
  const bool drop_e = gdrop.on() && (g.drop_where & 4);
  float* __restrict__ cs = reinterpret_cast<float*>(smem_raw);
  constexpr int SLAB = BM / EP;
  constexpr int VPR = BN / 4;
  constexpr int NV4 = SLAB * BN / 4 / NT;
  constexpr int RSTEP = NT / VPR;
  static_assert(NV4 >= 1, "epilogue slab too small");
  const int c4 = (tid % VPR) * 4, row0 = tid / VPR;
  const int gn = n0 + c4;
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 bn4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g.bias_mode == 1 && gn < g.N) bn4 = *reinterpret_cast<const float4*>(g.bias + gn);
  const float* __restrict__ aux = g.aux ? g.aux + coff : nullptr;
  const bool plain = g.bias_mode != 2 && !g.accumulate;
  const bool fast_hd = plain && g.write_act == RPDE_ACT_GELU && g.aux_out && !g.epi_dact;
  const bool fast_mul = plain && g.epi_dact == RPDE_EPI_MULAUX && !g.write_act;
#pragma unroll
  for (int e = 0; e < EP; ++e) {
    if (e == 1) STAMP(29);
    if (e > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rt = (wm * TM + i) * 32;
      if (rt / SLAB != e) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 32 + l31;
        const int rb0 = rt - e * SLAB + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) cs[(rb0 + (r & 3) + 8 * (r >> 2)) * BN + col] = acc[i][j][r];
      }
    }
    __syncthreads();
    if (e == 0) STAMP(28); else STAMP(30);
    if (gn < g.N) {
      // the mode is uniform over the launch: pick the row loop once, not per element
      if (fast_hd) {
        // C = gelu(u), aux_out = gelu'(u) * dropscale, u = dropout(alpha*acc + bias)
        // running offsets instead of a 64-bit row*ld multiply per row; the column sums only when asked for
        const int gm0 = m0 + e * SLAB + row0;
        long off = (long)gm0 * g.ldc + gn;
        uint64_t did = (uint64_t)((long)gm0 * g.drop_ld + gn);
        const long off_step = (long)RSTEP * g.ldc;
        const uint64_t did_step = (uint64_t)((long)RSTEP * g.drop_ld);
        float* __restrict__ dout = g.aux_out + coff;
        const bool want_sum = g.colsum != nullptr;
#pragma unroll 4
        for (int it = 0; it < NV4; ++it, off += off_step, did += did_step) {
          if (gm0 + it * RSTEP >= g.M) break;
          const float4 t = *reinterpret_cast<const float4*>(cs + (row0 + it * RSTEP) * BN + c4);
          float s[4] = {1.f, 1.f, 1.f, 1.f};
          if (drop_e) drop_scale4(gdrop, did, s);
          const float u0 = fmaf(t.x, g.alpha, bn4.x) * s[0], u1 = fmaf(t.y, g.alpha, bn4.y) * s[1];
          const float u2 = fmaf(t.z, g.alpha, bn4.z) * s[2], u3 = fmaf(t.w, g.alpha, bn4.w) * s[3];
          // (scalar on purpose: a packed-fp32 version, 11 instead of 20 issues per element, measured 0.5 % slower
          //  end to end -- v_pk_* instructions do not overlap with the other waves' MFMAs)
          float4 hv, dv;
          act_both(RPDE_ACT_GELU, u0, hv.x, dv.x); act_both(RPDE_ACT_GELU, u1, hv.y, dv.y);
          act_both(RPDE_ACT_GELU, u2, hv.z, dv.z); act_both(RPDE_ACT_GELU, u3, hv.w, dv.w);
          dv.x *= s[0]; dv.y *= s[1]; dv.z *= s[2]; dv.w *= s[3];
          *reinterpret_cast<float4*>(dout + off) = dv;
          *reinterpret_cast<float4*>(C + off) = hv;
          if (want_sum) { csum.x += hv.x; csum.y += hv.y; csum.z += hv.z; csum.w += hv.w; }
        }
      } else if (fast_mul) {
        // C = (alpha*acc + bias) * aux: backward-data through the stored derivative
#pragma unroll 4
        for (int it = 0; it < NV4; ++it) {
          const int row = row0 + it * RSTEP;
          const int gm = m0 + e * SLAB + row;
          if (gm >= g.M) break;
          float4 v = *reinterpret_cast<const float4*>(cs + row * BN + c4);
          const float4 a = *reinterpret_cast<const float4*>(aux + (long)gm * g.ldaux + gn);
          v.x = fmaf(v.x, g.alpha, bn4.x) * a.x; v.y = fmaf(v.y, g.alpha, bn4.y) * a.y;
          v.z = fmaf(v.z, g.alpha, bn4.z) * a.z; v.w = fmaf(v.w, g.alpha, bn4.w) * a.w;
          *reinterpret_cast<float4*>(C + (long)gm * g.ldc + gn) = v;
          csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w;
        }
      } else {
#pragma unroll 4
        for (int it = 0; it < NV4; ++it) {
          const int row = row0 + it * RSTEP;
          const int gm = m0 + e * SLAB + row;
          if (gm >= g.M) break;
          float4 v = *reinterpret_cast<const float4*>(cs + row * BN + c4);
          v.x = fmaf(v.x, g.alpha, bn4.x); v.y = fmaf(v.y, g.alpha, bn4.y);
          v.z = fmaf(v.z, g.alpha, bn4.z); v.w = fmaf(v.w, g.alpha, bn4.w);
          if (g.bias_mode == 2) { const float bm = g.bias[gm]; v.x += bm; v.y += bm; v.z += bm; v.w += bm; }
          if (g.epi_dact == RPDE_EPI_MULAUX) {
            const float4 a = *reinterpret_cast<const float4*>(aux + (long)gm * g.ldaux + gn);
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
            const float4 o = g.acc_src ? *reinterpret_cast<const float4*>(g.acc_src + coff + (long)gm * g.ldc + gn) : *cp;
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
  }
  STAMP(27);
  if (g.colsum) {
    __syncthreads();
    *reinterpret_cast<float4*>(cs + row0 * BN + c4) = csum;
    __syncthreads();
    if (tid < BN && n0 + tid < g.N) {
      float t = 0.f;
#pragma unroll
      for (int rr = 0; rr < RSTEP; ++rr) t += cs[rr * BN + tid];
      g.colsum[(long)mt * g.N + n0 + tid] = t;
    }
  }
}

template <int WM, int WN, int TM, int TN>
static void launch_x3_layout(const GemmK& g, bool ak, bool bk, dim3 grid, hipStream_t st) {
  if (g.a_img && bk) hipLaunchKernelGGL((gemm_bf16x3_kernel<WM, WN, TM, TN, true, true, false, true>), grid, dim3(64 * WM * WN), 0, st, g);
  else if (g.a_img) hipLaunchKernelGGL((gemm_bf16x3_kernel<WM, WN, TM, TN, true, false, false, true>), grid, dim3(64 * WM * WN), 0, st, g);
  else if (ak && bk && g.Bimg) hipLaunchKernelGGL((gemm_bf16x3_kernel<WM, WN, TM, TN, true, true, true>), grid, dim3(64 * WM * WN), 0, st, g);
  else if (ak && bk) hipLaunchKernelGGL((gemm_bf16x3_kernel<WM, WN, TM, TN, true, true>), grid, dim3(64 * WM * WN), 0, st, g);
  else if (ak && !bk) hipLaunchKernelGGL((gemm_bf16x3_kernel<WM, WN, TM, TN, true, false>), grid, dim3(64 * WM * WN), 0, st, g);
  else if (!ak && !bk) hipLaunchKernelGGL((gemm_bf16x3_kernel<WM, WN, TM, TN, false, false>), grid, dim3(64 * WM * WN), 0, st, g);
  else hipLaunchKernelGGL((gemm_bf16x3_kernel<WM, WN, TM, TN, false, true>), grid, dim3(64 * WM * WN), 0, st, g);
}

// fp32 weights [N,K] (k-major, ld) or [K,N] (x-major) -> [K/32][3][Npad][32] bf16 images, 16-byte chunks of
// each 64-byte row XOR-swizzled with (row>>2)&3 (= the LDS image of a stage), pad rows zero
int split_npad(int N) { return ((N + 127) / 128) * 128; }

__device__ __forceinline__ void split_weights_thread(long t, const float* __restrict__ w, int kmajor, long ld, int N, int K,
                                                     int npad, char* __restrict__ out) {
  const int kchunks = ((K + XBK - 1) / XBK) * (XBK / 8);
  if (t >= (long)npad * kchunks) return;
  const int n = (int)(t % npad), kc = (int)(t / npad);     // consecutive threads: consecutive rows
  const int k0 = kc * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    v[j] = (n < N && k0 + j < K) ? (kmajor ? w[(long)n * ld + k0 + j] : w[(long)(k0 + j) * ld + n]) : 0.f;
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split2(v[2 * j], v[2 * j + 1], h[j], m[j], l[j]);
  const int kt = k0 / XBK, c = (k0 % XBK) / 8;
  char* p = out + ((long)kt * 3 * npad + n) * 64 + ((c ^ ((n >> 2) & 3)) << 4);
  *reinterpret_cast<uint4*>(p) = make_uint4(h[0], h[1], h[2], h[3]);
  *reinterpret_cast<uint4*>(p + (long)npad * 64) = make_uint4(m[0], m[1], m[2], m[3]);
  *reinterpret_cast<uint4*>(p + 2L * npad * 64) = make_uint4(l[0], l[1], l[2], l[3]);
}

__global__ __launch_bounds__(256) void k_split_weights(const float* __restrict__ w, int kmajor, long ld, int N, int K,
                                                       int npad, char* __restrict__ out) {
  split_weights_thread((long)blockIdx.x * 256 + threadIdx.x, w, kmajor, ld, N, K, npad, out);
}

// the images of several weights in one launch (all layers of a FeedForward: the small configurations are launch-bound)
__global__ __launch_bounds__(256) void k_split_weights_multi(const SplitJobs J) {
  int jn = 0;
#pragma unroll
  for (int q = 1; q < SplitJobs::MAX; ++q)
    if (q < J.n && (int)blockIdx.x >= J.j[q].blk0) jn = q;
  SplitJobs::Job jb = J.j[0];
#pragma unroll
  for (int q = 1; q < SplitJobs::MAX; ++q)
    if (q == jn) jb = J.j[q];
  split_weights_thread((long)((int)blockIdx.x - jb.blk0) * 256 + threadIdx.x, jb.w, jb.kmajor, jb.ld, jb.N, jb.K, ((jb.N + 127) / 128) * 128,
                       jb.out);
}

size_t split_bytes(int N, int K) { return (size_t)((K + XBK - 1) / XBK) * 3 * split_npad(N) * 64; }

int split_weights(const float* w, int kmajor, long ld, int N, int K, void* out, hipStream_t st) {
  RPDE_CHECK_ARG(w && out && N > 0 && K > 0, "split_weights: bad arguments (N=%d K=%d)", N, K);
  RPDE_CHECK_ARG((reinterpret_cast<uintptr_t>(out) & 15) == 0, "split_weights: output must be 16-byte aligned");
  const int npad = split_npad(N);
  const long threads = (long)npad * ((K + XBK - 1) / XBK) * (XBK / 8);
  hipLaunchKernelGGL(k_split_weights, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, w, kmajor, ld, N, K, npad,
                     static_cast<char*>(out));
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

int split_weights_multi(SplitJobs& J, hipStream_t st) {
  if (J.n == 0) return RPDE_OK;
  int blocks = 0;
  for (int q = 0; q < J.n; ++q) {
    RPDE_CHECK_ARG(J.j[q].w && J.j[q].out && (reinterpret_cast<uintptr_t>(J.j[q].out) & 15) == 0, "split_weights_multi: bad job %d", q);
    const long threads = (long)split_npad(J.j[q].N) * ((J.j[q].K + XBK - 1) / XBK) * (XBK / 8);
    J.j[q].blk0 = blocks;
    blocks += (int)((threads + 255) / 256);
  }
  hipLaunchKernelGGL(k_split_weights_multi, dim3((unsigned)blocks), dim3(256), 0, st, J);
  RPDE_LAUNCH_CHECK();
  J.n = 0;
  return RPDE_OK;
}

bool bf16x3_supports(int bm, int bn) { return (bm == 128 || bm == 64) && (bn == 128 || bn == 64); }

int launch_bf16x3(const GemmK& g, int bm, int bn, bool ak, bool bk, dim3 grid, hipStream_t st) {
  // (an eight-wave 128x128 variant -- gemm_bf16x3_kernel<2, 4, 2, 1, ...>, 98 VGPRs, 4 waves per SIMD -- is
  //  correct but measured 6 % slower on the FeedForward forward GEMM: the kernel is not latency bound)
  if (bm == 128 && bn == 128) launch_x3_layout<2, 2, 2, 2>(g, ak, bk, grid, st);
  else if (bm == 128 && bn == 64) launch_x3_layout<4, 1, 1, 2>(g, ak, bk, grid, st);
  else if (bm == 64 && bn == 128) launch_x3_layout<1, 4, 2, 1>(g, ak, bk, grid, st);
  else if (bm == 64 && bn == 64) launch_x3_layout<2, 2, 1, 1>(g, ak, bk, grid, st);
  else { set_error("bf16x3: unsupported tile %dx%d", bm, bn); return RPDE_ERR_ARG; }
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // namespace rpde

#ifdef RPDE_STAMPS
extern "C" int rpde_debug_stamps(unsigned long long* host_out) {
  RPDE_HIP(hipDeviceSynchronize());
  RPDE_HIP(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(rpde::g_stamps), sizeof(unsigned long long) * 64 * 32));
  return RPDE_OK;
}
#endif
extern "C" size_t rpde_split_weights_bytes(int N, int K) { return (N > 0 && K > 0) ? rpde::split_bytes(N, K) : 0; }
extern "C" int rpde_split_weights(const float* w, int kmajor, int64_t ld, int N, int K, void* out, void* stream) {
  return rpde::split_weights(w, kmajor, (long)ld, N, K, out, rpde::as_stream(stream));
}
