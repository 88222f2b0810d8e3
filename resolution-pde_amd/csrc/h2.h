// fp32 products on the f16 matrix pipe by two-piece splitting ("h2" arithmetic), shared by the fused kernels.
//
//   a * 2^ea = ah + al + O(2^-22 |a 2^ea|),   ah = f16(a 2^ea),  al = f16(a 2^ea - ah)         (11 + 11 significant bits)
//   a b = 2^-(ea+eb) (ah bh + ah bl + al bh) + O(2^-22 |a b|)        -- three v_mfma_f32_16x16x32_f16, fp32 accumulate
//
// f16 has a 5-bit exponent, so every operand block carries a power-of-two scale chosen from the block's own maximum
// (|max| -> [2^14, 2^15)): multiplying by it is exact, the low piece keeps 2^-24 absolute resolution in scaled units
// (2^-39 of the block maximum), and the product of the scales is undone on the fp32 accumulator.  Per product the
// error is <= 3 * 2^-22 (representation of a, of b, and the dropped al*bl), rms ~1.5e-7: fp32-GEMM class accuracy
// with half the matrix instructions of the three-piece bf16 split (gemm_bf16x3.hip), which stays the generic path.
#pragma once
#include "rpde_internal.h"

namespace rpde {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short s16x4v __attribute__((ext_vector_type(4)));

constexpr int H2_TABLE_EXP = 12;     // DFT tables (|entry| <= 2/sqrt(n) <= 1) are stored times 2^12

// power-of-two scale for a block whose largest magnitude is m (any finite float, 0 allowed):
// scale = 2^(141 - E) puts m into [2^14, 2^15); inv = 2^(E - 141 - extra) undoes it together with `extra` more
// binary places (the table scale).  E is clamped so that both stay normal numbers.
__device__ __forceinline__ void h2_scale(float m, int extra, float& scale, float& inv) {
  int E = (int)(__float_as_uint(m) >> 23) & 0xff;
  E = max(E, 15 + extra);
  scale = __uint_as_float((unsigned)(268 - E) << 23);
  inv = __uint_as_float((unsigned)(E - 14 - extra) << 23);
}

// four scaled floats -> hi / lo f16 pieces (round to nearest both times)
__device__ __forceinline__ void h2_split4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  union { f16x4 v; uint2 u; } h, l;
  h.v = __builtin_convertvector((f4){a, b, c, d}, f16x4);
  const f4 hf = __builtin_convertvector(h.v, f4);
  l.v = __builtin_convertvector((f4){a - hf.x, b - hf.y, c - hf.z, d - hf.w}, f16x4);
  hi = h.u; lo = l.u;
}

// maximum over the wave of non-negative values, returned in every lane.  Six DPP steps on the VALU (row shifts, then
// the two row broadcasts) and one v_readlane -- __shfl_xor would make six dependent round trips through the LDS crossbar
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_step(float v) {
  const int t = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return fmaxf(v, __int_as_float(t));
}
__device__ __forceinline__ float wave_max(float v) {
  v = dpp_max_step<0x111, 0xf>(v);     // row_shr:1
  v = dpp_max_step<0x112, 0xf>(v);     // row_shr:2
  v = dpp_max_step<0x114, 0xf>(v);     // row_shr:4
  v = dpp_max_step<0x118, 0xf>(v);     // row_shr:8   -> lane 15 of each row of 16 holds the row maximum
  v = dpp_max_step<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
  v = dpp_max_step<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave maximum
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// sum over the wave, returned in every lane: the same six DPP steps with additions (out-of-row sources read as zero)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_step(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = dpp_add_step<0x111, 0xf>(v);
  v = dpp_add_step<0x112, 0xf>(v);
  v = dpp_add_step<0x114, 0xf>(v);
  v = dpp_add_step<0x118, 0xf>(v);     // lane 15 of each row: the row's sum
  v = dpp_add_step<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3: lane 31 = rows 0+1, lane 63 = rows 2+3 (so far)
  v = dpp_add_step<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3: lane 63 = everything
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// the value held by lane l ^ 16 / l ^ 32, by gfx950's v_permlane16_swap / v_permlane32_swap (vector ALU: no trip through
// the LDS crossbar as __shfl_xor takes).  swap(v, v) returns {v with its odd rows (upper half) replaced by the even rows
// (lower half), v with its even rows (lower half) replaced by the odd rows (upper half)}: each lane picks the copy in
// which its own position was overwritten by its partner.
__device__ __forceinline__ float lane_xor16(float v) {
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float((threadIdx.x & 16) ? r.x : r.y);
}
__device__ __forceinline__ float lane_xor32(float v) {
  typedef unsigned u2v __attribute__((ext_vector_type(2)));
  const u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float((threadIdx.x & 32) ? r.x : r.y);
}

// three-term product of split operands, small terms first
__device__ __forceinline__ f32x4v h2_mfma32(f16x8 ah, f16x8 al, f16x8 bh, f16x8 bl, f32x4v c) {
#ifdef RPDE_EXP_H2_1TERM      // TIMING EXPERIMENT ONLY (wrong to 2^-11): what the two correction terms cost in time / energy
  (void)al; (void)bl;
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
#else
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
#endif
}
// (There is deliberately no 16-deep variant on v_mfma_f32_16x16x16_f16.  Round 2 saw wrong accumulators when the fused
//  synthesis kernel closed a 32-deep chain with a 16-deep tail; round 4 reduced it to two instructions
//  (profiles/ubench/mfma_mix.hip, output + disassembly in profiles/r04_mfma_mix.txt): ROCm 7.2 emits
//  `v_mfma_f32_16x16x32_f16 v[0:3], ..` and the dependent `v_mfma_f32_16x16x16_f16 v[0:3], .., v[0:3]` back to back and
//  half of the result is wrong; with s_nop padding between them, or in the order 16-deep -> 32-deep, it is exact.  The
//  shorter instruction reads its SrcC before the longer one has written all of it, and neither the hardware nor the
//  compiler's hazard recognizer waits for this opcode pair.  Same-shape chains are handled.  Short reduction tails are
//  therefore packed into 32-deep fragments -- fused_spectral.hip, h2_store_tail -- which also costs no padding.)

// LDS accesses of one wave to its private staging area: the hardware executes a wave's DS instructions in order;
// this only keeps the compiler from reordering a lane's read above another lane's write
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

}  // namespace rpde
