// Host entry of the strided batched fp32-MFMA GEMM: validation, tile choice,
// grid shape; the kernels live in gemm_kernel.h, instantiated per operand
// layout in gemm_{nt,nn,tn,tt}.hip.
#include "gemm_kernel.h"

#include <stdlib.h>

namespace rpde {

int gemm_variant() {
  static const int v = [] { const char* e = getenv("RPDE_GEMM_VARIANT"); return e ? atoi(e) : 0; }();
  return v;
}

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int launch_gemm(const rpde_gemm_desc& d, hipStream_t st) {
  RPDE_CHECK_ARG(d.A && d.B && d.C, "gemm: null operand");
  RPDE_CHECK_ARG(d.M > 0 && d.N > 0 && d.K >= 0, "gemm: bad dims %d %d %d", d.M, d.N, d.K);
  RPDE_CHECK_ARG(d.batch >= 1 && d.zdiv >= 1 && d.ksplit >= 1, "gemm: bad batch/zdiv/ksplit");
  RPDE_CHECK_ARG(!(d.ksplit > 1 && (d.bias_mode || d.epi_dact || d.accumulate || d.write_act)),
                 "gemm: split-K slabs take no epilogue");
  RPDE_CHECK_ARG(!d.epi_dact || d.aux, "gemm: epi_dact needs aux");
  RPDE_CHECK_ARG(!(d.act_a && d.act_b), "gemm: only one operand can carry a staged activation");
  GemmK g;
  g.A = d.A; g.B = d.B; g.C = d.C; g.M = d.M; g.N = d.N; g.K = d.K;
  g.lda = d.lda; g.ldb = d.ldb; g.ldc = d.ldc;
  g.zdiv = d.zdiv; g.ztotal = d.batch * d.ksplit;
  g.sA1 = d.sA1; g.sA2 = d.sA2; g.sB1 = d.sB1; g.sB2 = d.sB2; g.sC1 = d.sC1; g.sC2 = d.sC2;
  g.ksplit = d.ksplit; g.sCk = d.sCk;
  int kchunk = (d.K + d.ksplit - 1) / d.ksplit;
  kchunk = ((kchunk + BK_MAX - 1) / BK_MAX) * BK_MAX;
  g.kchunk = kchunk > 0 ? kchunk : BK_MAX;
  g.alpha = d.alpha; g.accumulate = d.accumulate;
  g.bias = d.bias; g.bias_mode = d.bias ? d.bias_mode : 0;
  g.act_a = d.act_a; g.act_b = d.act_b; g.epi_dact = d.epi_dact; g.write_act = d.write_act;
  g.aux = d.aux; g.ldaux = d.ldaux;
  g.drop = make_drop(d.drop_p, d.drop_seed, d.drop_epoch); g.drop_ld = d.drop_ld; g.drop_where = d.drop_where;
  const int pro = d.act_a ? 1 : ((d.act_b || (g.drop.on() && (d.drop_where & 2))) ? 2 : ((g.drop.on() && (d.drop_where & 1)) ? 1 : 0));

  // 16-byte vector path: aligned bases / strides / leading dims, and the extent
  // along each operand's contiguous index a multiple of 4 (whole vectors only)
  const bool ak = d.a_kmajor != 0, bk = d.b_kmajor != 0;
  bool vec = al16(d.A) && al16(d.B) && (d.lda % 4 == 0) && (d.ldb % 4 == 0) && (d.sA1 % 4 == 0) && (d.sA2 % 4 == 0) &&
             (d.sB1 % 4 == 0) && (d.sB2 % 4 == 0);
  vec = vec && ((ak ? d.K : d.M) % 4 == 0) && ((bk ? d.K : d.N) % 4 == 0);
  if (g.drop.on() && (d.drop_where & 3)) vec = vec && (d.drop_ld % 4 == 0);

  int BMc = 64, BNc = 64;
  if (vec) {
    const int bm = d.M <= 32 ? 32 : (d.M <= 64 ? 64 : 128);
    const int bn = d.N <= 32 ? 32 : (d.N <= 64 ? 64 : 128);
    if (bm == 32) { BMc = 32; BNc = 128; }
    else if (bn == 32) { BMc = 128; BNc = 32; }
    else { BMc = bm; BNc = bn; }
    // small problems (the 1-D configurations): 128x128 tiles would leave most of the 256 CUs idle, or give each just
    // one four-wave workgroup (FFNO1D at B = 16: 256 tiles; 64x64 tiles: 1.50 -> 1.44 ms per step under a HIP graph)
    static const bool small_tiles = [] { const char* e = getenv("RPDE_SMALL_TILES"); return !(e && e[0] == '0'); }();
    static const long small_below = [] { const char* e = getenv("RPDE_SMALL_TILES_BELOW"); return e ? atol(e) : 384L; }();
    if (small_tiles && BMc == 128 && BNc == 128 && !d.colsum &&
        (long)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.batch * d.ksplit < small_below) { BMc = 64; BNc = 64; }
  }
  g.mtiles = (d.M + BMc - 1) / BMc;
  g.ntiles = (d.N + BNc - 1) / BNc;
  g.swz = (g.ntiles > 1 && g.mtiles >= 64) ? 1 : 0;
  const long zt = (long)g.ztotal;
  const int gy = (int)(zt < 32768 ? zt : 32768);
  const int gz = (int)((zt + gy - 1) / gy);
  const int tiles = g.mtiles * g.ntiles;
  if (!g.swz && tiles > 1 && tiles <= 16 && gz == 1 && zt % 8 == 0) g.swz = 2;
  dim3 grid(g.swz == 1 ? ((g.mtiles + 7) / 8) * 8 * g.ntiles : g.mtiles * g.ntiles, gy, gz);

  // LDS-staged vector epilogue: C rows, aux rows, bias and the dropout ids must be 16-byte friendly
  g.cvec = vec && al16(d.C) && (d.ldc % 4 == 0) && (d.sC1 % 4 == 0) && (d.sC2 % 4 == 0) && (d.sCk % 4 == 0) &&
           (d.N % 4 == 0) && (!d.aux || (al16(d.aux) && d.ldaux % 4 == 0)) &&
           (!(d.bias && d.bias_mode == 1) || al16(d.bias)) && (!(g.drop.on() && (d.drop_where & 4)) || d.drop_ld % 4 == 0);
  g.colsum = d.colsum;
  g.aux_out = d.aux_out;
  g.Bimg = nullptr; g.npad = 0; g.a_img = 0;
  g.acc_src = d.accumulate ? d.acc_src : nullptr;
  if (g.acc_src) g.cvec = g.cvec && al16(g.acc_src);
  RPDE_CHECK_ARG(!d.aux_out || d.write_act, "gemm: aux_out needs write_act");
  if (d.aux_out) g.cvec = g.cvec && al16(d.aux_out);
  RPDE_CHECK_ARG(!d.colsum || (g.cvec && BMc == 128 && d.batch == 1 && d.ksplit == 1),
                 "gemm: colsum needs the vector epilogue, M > 64 and a single un-split problem");

  // exact-product split-bf16 path (fp32 in / out / accumulate, products via 3-way bf16 splitting on the
  // bf16 matrix pipe): every plain vector GEMM on a supported tile.  RPDE_SPLIT_BF16=0 forces the
  // native fp32 MFMA kernels.
  // RPDE_SPLIT_BF16: 0 = native fp32 MFMA kernels only, 1 = split path for k-major x k-major problems only,
  // default = every operand layout (x-major operands go through the transposing LDS read)
  static const int split_mode = [] { const char* e = getenv("RPDE_SPLIT_BF16"); return e ? atoi(e) : 2; }();
  const bool split_on = split_mode != 0, split_all = split_mode >= 2;
  // a pre-split operand (rpde_split_weights) is k-major whatever the layout of the fp32 original, and
  // zero-padded in k: with a pre-split A and an x-major B any K works
  const bool bimg = d.b_split && ak && d.sB1 == 0 && d.sB2 == 0 && g.kchunk % 32 == 0 && d.K % 32 == 0;
  const bool aimg = !bimg && d.a_split && d.sA1 == 0 && d.sA2 == 0 && g.kchunk % 32 == 0 && (d.K % 32 == 0 || (!bk && d.ksplit == 1));
  const bool k_ok = d.K >= 32 && d.K % 32 == 0;
  if (split_on && vec && g.cvec && pro == 0 && (k_ok || (aimg && d.K >= 1)) && d.lda < (1L << 24) && d.ldb < (1L << 24) &&
      (((ak || aimg) && (bk || bimg)) || split_all) && bf16x3_supports(BMc, BNc)) {
    if (bimg) {
      g.Bimg = static_cast<const char*>(d.b_split);
      g.npad = split_npad(d.N);
    } else if (aimg) {
      g.Bimg = static_cast<const char*>(d.a_split);
      g.npad = split_npad(d.M);
      g.a_img = 1;
    }
    return launch_bf16x3(g, BMc, BNc, ak || aimg, bk || bimg, grid, st);
  }

  if (ak && bk) return launch_nt(g, BMc, BNc, pro, vec, grid, st);
  if (ak && !bk) return launch_nn(g, BMc, BNc, pro, vec, grid, st);
  if (!ak && !bk) return launch_tn(g, BMc, BNc, pro, vec, grid, st);
  return launch_tt(g, BMc, BNc, pro, vec, grid, st);
}

}  // namespace rpde

extern "C" int rpde_gemm_f32(const rpde_gemm_desc* d, void* stream) {
  if (!d) { rpde::set_error("gemm: null descriptor"); return RPDE_ERR_ARG; }
  return rpde::launch_gemm(*d, rpde::as_stream(stream));
}
