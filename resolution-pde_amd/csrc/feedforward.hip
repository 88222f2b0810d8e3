// FeedForward (models/custom_layer.py:49-68) and the pointwise linears /
// 1x1 convolutions around the spectral layers, as fp32-MFMA GEMMs.
//
// Each hidden pre-activation z_l stays on chip: the producing GEMM's epilogue evaluates the
// activation ONCE per element, u = dropout(z_l), and stores h_l = gelu(u) (input of the next GEMM
// and of the weight-gradient GEMM) and, when training, d_l = gelu'(u) * dropscale (what the
// backward-data GEMM multiplies by).  Every consumer is then a plain GEMM.  (Round-1 history: the
// first design re-evaluated gelu while staging z_l into LDS -- 5 evaluations per element over a
// training step, costing 25-30 % of each fused GEMM; see profiles/r01_b_kernel_bench.txt.)
#include "rpde_internal.h"
#include "pointwise.h"
#include "gemm_kernel.h"
#include "ff_fused.h"
#include "wgrad_h2.h"
#include "thin_linear.h"
#include "conv_small.h"

namespace rpde {

static inline uint64_t layer_seed(uint64_t seed, int l) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(l + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// the thin-linear kernels move the wide operand as float4: a view with an odd storage offset takes the GEMM path
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static inline int ff_in(const rpde_ff_params* p, int l) { return l == 0 ? p->dim : p->dim * p->factor; }
static inline int ff_out(const rpde_ff_params* p, int l) { return l == p->n_layers - 1 ? p->dim : p->dim * p->factor; }

static inline int wgrad_split(long P, int out_f, int in_f) {
  const int tiles = ((out_f + 127) / 128) * ((in_f + 127) / 128);
  long s = 768 / tiles;          // 3 resident workgroups per CU (split-bf16 kernel), one wave of them
  const long cap = (P + 127) / 128;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return (int)s;
}

// y[P,out] = act_in(x)[P,in] . W[out,in]^T + b
// act_out != 0: y = act(dropout(z)) and, if dy, dy = act'(dropout(z)) * dropscale (mask of THIS layer's output)
// wimg: optional scratch of split_bytes(out_f, in_f): the weight is split into bf16 images once per call
// instead of once per workgroup (same bits out; see gemm_bf16x3.hip)
static inline bool worth_presplit(long P, int n, int k) { return P >= 1024 && k % 32 == 0 && n > 32; }
static int linear_fwd_impl(const float* x, const float* w, const float* b, float* y, long P, int in_f, int out_f,
                           int act_out, float* dy, float drop_p, uint64_t drop_seed, hipStream_t st,
                           void* wimg = nullptr, const uint64_t* drop_epoch = nullptr, bool wimg_ready = false) {
  if (act_out == RPDE_ACT_IDENTITY && !dy && drop_p == 0.f) {
    // lifting / projection shapes: streaming kernels instead of degenerate GEMMs (thin_linear.hip)
    if (in_f <= 4 && thin_linear_ok(in_f, out_f) && al16(y)) return thin_expand(x, w, in_f, 1, b, y, P, in_f, out_f, st);
    if (out_f <= 4 && thin_linear_ok(out_f, in_f) && al16(x)) return thin_contract(x, w, in_f, 1, b, y, P, out_f, in_f, st);
  }
  rpde_gemm_desc d = gemm_desc();
  d.A = x; d.a_kmajor = 1; d.lda = in_f;
  d.B = w; d.b_kmajor = 1; d.ldb = in_f;
  if (wimg && worth_presplit(P, out_f, in_f)) {
    if (!wimg_ready) RPDE_TRY(split_weights(w, 1, in_f, out_f, in_f, wimg, st));
    d.b_split = wimg;
  }
  d.C = y; d.ldc = out_f;
  d.M = (int)P; d.N = out_f; d.K = in_f;
  d.bias = b; d.bias_mode = b ? 1 : 0;
  d.write_act = act_out; d.aux_out = act_out ? dy : nullptr;
  if (act_out) { d.drop_p = drop_p; d.drop_seed = drop_seed; d.drop_ld = out_f; d.drop_where = 4; d.drop_epoch = drop_epoch; }
  return launch_gemm(d, st);
}

// gw[out,in] = gy[P,out]^T . act_in(x)[P,in]   (split over P, slabs reduced here);  gb = colsum(gy)
// act_x: the x operand is act_x(x) (recompute mode of the fused FeedForward: x holds u, the layer's input is gelu(u))
static int linear_wgrad_impl(const float* x, const float* gy, float* gw, float* gb, long P, int in_f, int out_f,
                             float* ws_slabs, float* ws_colsum, hipStream_t st, int act_x = RPDE_ACT_IDENTITY,
                             FoldJobs* defer = nullptr) {
  if (act_x == RPDE_ACT_IDENTITY && (gw || gb)) {
    // one pass over both operands gives the weight gradient and the bias gradient of a lifting / projection layer
    if (in_f <= 4 && thin_linear_ok(in_f, out_f) && al16(gy)) return thin_outer(x, gy, gw, 1, gb, nullptr, P, in_f, out_f, ws_slabs, st);
    if (out_f <= 4 && thin_linear_ok(out_f, in_f) && al16(x)) return thin_outer(gy, x, gw, 0, nullptr, gb, P, out_f, in_f, ws_slabs, st);
  }
  if (gw && wgrad_h2_ok(P, out_f, in_f)) {
    RPDE_TRY(wgrad_h2(gy, x, gw, P, in_f, out_f, act_x, ws_slabs, st, defer));
  } else if (gw) {
    const int S = wgrad_split(P, out_f, in_f);
    rpde_gemm_desc d = gemm_desc();
    d.A = gy; d.a_kmajor = 0; d.lda = out_f;
    d.B = x; d.b_kmajor = 0; d.ldb = in_f; d.act_b = act_x;
    d.M = out_f; d.N = in_f; d.K = (int)P;
    if (S > 1) {
      d.C = ws_slabs; d.ldc = in_f; d.ksplit = S; d.sCk = (long)out_f * in_f;
      RPDE_TRY(launch_gemm(d, st));
      // (defer: the caller folds every slab set of its backward pass in one launch; ws_slabs is then its own region)
      if (!(defer && defer->add(ws_slabs, gw, out_f * in_f, S, (long)out_f * in_f)))
        RPDE_TRY(reduce_slabs(ws_slabs, gw, (long)out_f * in_f, S, (long)out_f * in_f, 1.f, 0, st));
    } else {
      d.C = gw; d.ldc = in_f;
      RPDE_TRY(launch_gemm(d, st));
    }
  }
  if (gb) RPDE_TRY(colsum(gy, gb, P, out_f, out_f, ws_colsum, 0, st));
  return RPDE_OK;
}

// gx[P,in] = gy[P,out] . W[out,in]   (optionally through act': * act'(drop(z)) * dropscale)
static inline bool al16p(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline long colsum_tiles(long P) { return (P + 127) / 128; }
// the backward-data GEMM can hand back per-tile column sums of gx (= the bias gradient of the layer below)
static inline bool can_fuse_colsum(const float* gy, const float* w, const float* gx, const float* z, long P, int in_f,
                                   int out_f) {
  return P > 64 && in_f % 4 == 0 && out_f % 4 == 0 && al16p(gy) && al16p(w) && al16p(gx) && (!z || al16p(z));
}

// gx[P,in] = (gy[P,out] . W[out,in]) (* dstored[P,in] when given: the derivative saved by the forward epilogue)
// wimg: optional scratch of split_bytes(in_f, out_f): W (x-major here) is split into k-major bf16 images
// first, which is what lets this GEMM take the split-bf16 path at all
static int linear_dgrad_impl(const float* gy, const float* w, float* gx, long P, int in_f, int out_f,
                             const float* dstored, float* colsum_slab, hipStream_t st, void* wimg = nullptr,
                             bool wimg_ready = false) {
  if (!dstored && !colsum_slab) {
    if (out_f <= 4 && thin_linear_ok(out_f, in_f) && al16(gx)) return thin_expand(gy, w, 1, in_f, nullptr, gx, P, out_f, in_f, st);
    if (in_f <= 4 && thin_linear_ok(in_f, out_f) && al16(gy)) return thin_contract(gy, w, 1, in_f, nullptr, gx, P, in_f, out_f, st);
  }
  rpde_gemm_desc d = gemm_desc();
  d.A = gy; d.a_kmajor = 1; d.lda = out_f;
  d.B = w; d.b_kmajor = 0; d.ldb = in_f;
  if (wimg && worth_presplit(P, in_f, out_f)) {
    if (!wimg_ready) RPDE_TRY(split_weights(w, 0, in_f, in_f, out_f, wimg, st));
    d.b_split = wimg;
  }
  d.C = gx; d.ldc = in_f;
  d.M = (int)P; d.N = in_f; d.K = out_f;
  if (dstored) { d.epi_dact = RPDE_EPI_MULAUX; d.aux = dstored; d.ldaux = in_f; }
  d.colsum = colsum_slab;
  return launch_gemm(d, st);
}

// scratch for one layer's split weight images (floats), any layer of a FeedForward with hidden width hid
static size_t ff_wimg_floats(int hid) { return (split_bytes(hid, ((hid + 31) / 32) * 32) + 3) / 4; }

static size_t wgrad_ws_floats(long P, int in_f, int out_f) {
  const int S = wgrad_split(P, out_f, in_f);
  size_t a = (S > 1 ? (size_t)S * out_f * in_f : 0);
  const size_t b = wgrad_h2_slab_floats(P, out_f, in_f);
  if (b > a) a = b;
  if (in_f <= 4 && thin_linear_ok(in_f, out_f) && thin_outer_ws_floats(P, in_f, out_f) > a) a = thin_outer_ws_floats(P, in_f, out_f);
  if (out_f <= 4 && thin_linear_ok(out_f, in_f) && thin_outer_ws_floats(P, out_f, in_f) > a) a = thin_outer_ws_floats(P, out_f, in_f);
  return a;
}

}  // namespace rpde

using namespace rpde;

extern "C" {

// ------------------------------ FeedForward --------------------------------
// backward workspace of the GEMM path: every layer has its own weight-gradient slabs, column-sum slabs and weight
// images, so that the folds and the splits of the whole pass are one launch each (FoldJobs, SplitJobs)
size_t rpde_feedforward_ws_bytes(int64_t P, int dim, int factor, int n_layers) {
  const int hid = n_layers > 1 ? dim * factor : dim;
  size_t n = 2 * arena_bytes((size_t)P * hid);
  for (int l = 0; l < n_layers; ++l) {
    const int i = l == 0 ? dim : dim * factor, o = l == n_layers - 1 ? dim : dim * factor;
    n += arena_bytes(wgrad_ws_floats(P, i, o));
    n += arena_bytes((size_t)(colsum_tiles(P) + REDUCE_CHUNKS) * hid);
    n += arena_bytes(ff_wimg_floats(hid));
  }
  n += arena_bytes(colsum_ws_floats(P, hid)) + arena_bytes(ff_tail_bwd_ws_floats(P, dim));
  rpde_ff_params q;
  memset(&q, 0, sizeof(q));
  q.dim = dim; q.factor = factor; q.n_layers = n_layers;
  if (ff3_fused_ok(&q, (long)P))           // + dz3, the per-workgroup sums, the weight fragments of the fused kernel
    n += arena_bytes((size_t)P * dim) + arena_bytes(ff3_fused_bwd_part_floats()) + arena_bytes(ff3_fused_ws_floats());
  return n;
}

size_t rpde_feedforward_fwd_ws_bytes(int dim, int factor, int n_layers) {
  const size_t a = (size_t)n_layers * arena_bytes(ff_wimg_floats(n_layers > 1 ? dim * factor : dim)), b = arena_bytes(ff3_fused_ws_floats());
  return a > b ? a : b;
}

// 1 when rpde_feedforward_fwd runs this shape as one fused kernel that needs no hidden buffers in evaluation
int rpde_feedforward_is_fused(int dim, int factor, int n_layers, int64_t P) {
  rpde_ff_params q;
  memset(&q, 0, sizeof(q));
  q.dim = dim; q.factor = factor; q.n_layers = n_layers;
  return ff3_fused_ok(&q, (long)P) ? 1 : 0;
}

// ---- evaluation with frozen weights: the weight fragments of the fused kernel are built once (rpde_feedforward_prepare)
// and reused by every call until the weights change (rpde.ops.frozen_weights owns that promise) ----
int rpde_feedforward_prepare(const rpde_ff_params* p, void* prep, size_t prep_bytes, void* stream) {
  RPDE_CHECK_ARG(p && prep, "feedforward_prepare: bad arguments");
  RPDE_CHECK_ARG(ff3_fused_ok(p, 32) && prep_bytes >= arena_bytes(ff3_fused_ws_floats()), "feedforward_prepare: shape not fused or buffer too small");
  RPDE_CHECK_ARG(p->weights && p->weights[0] && p->weights[1] && p->weights[2], "feedforward_prepare: null weights");
  return ff3_fused_prepare(p, prep, as_stream(stream));
}

int rpde_feedforward_fwd_prepared(const rpde_ff_params* p, const float* x, const float* residual, float* out, int64_t P,
                                  const void* prep, size_t prep_bytes, void* stream) {
  RPDE_CHECK_ARG(p && x && out && prep && P > 0 && P < (1L << 31), "feedforward_fwd_prepared: bad arguments");
  RPDE_CHECK_ARG(ff3_fused_ok(p, P) && prep_bytes >= arena_bytes(ff3_fused_ws_floats()), "feedforward_fwd_prepared: shape not fused");
  RPDE_CHECK_ARG(!p->layer_norm || (p->ln_gamma && p->ln_beta), "feedforward_fwd_prepared: layer_norm needs gamma/beta");
  RPDE_CHECK_ARG(p->dropout_p == 0.f, "feedforward_fwd_prepared: evaluation only");
  return ff3_fused_fwd(p, x, residual, nullptr, nullptr, out /* z_last is not written in evaluation */, out, P,
                       const_cast<void*>(prep), as_stream(stream), true);
}

int rpde_feedforward_fwd(const rpde_ff_params* p, const float* x, const float* residual, float* const* hs,
                         float* const* ds, float* z_last, float* out, int64_t P, void* ws, size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(p && x && z_last && out && P > 0, "feedforward_fwd: bad arguments");
  RPDE_CHECK_ARG(p->n_layers >= 1 && p->dim > 0 && p->factor > 0, "feedforward_fwd: bad shape");
  RPDE_CHECK_ARG(P < (1L << 31), "feedforward_fwd: too many points for one call");
  RPDE_CHECK_ARG(p->dropout_p >= 0.f && p->dropout_p < 1.f, "feedforward_fwd: dropout %f", p->dropout_p);
  hipStream_t st = as_stream(stream);
  if (ff3_fused_ok(p, P) && ws && ws_bytes >= arena_bytes(ff3_fused_ws_floats())) {
    RPDE_CHECK_ARG(p->weights[0] && p->weights[1] && p->weights[2], "feedforward_fwd: null weights");
    RPDE_CHECK_ARG(!p->layer_norm || (p->ln_gamma && p->ln_beta), "feedforward_fwd: layer_norm needs gamma/beta");
    return ff3_fused_fwd(p, x, residual, hs, ds, z_last, out, P, ws, st);
  }
  RPDE_CHECK_ARG(p->n_layers == 1 || hs, "feedforward_fwd: hidden buffers missing");
  const int L = p->n_layers;
  // optional scratch (rpde_feedforward_fwd_ws_bytes): weights are pre-split once per call; without it the
  // GEMMs split them per workgroup (slower, same bits)
  // (one image per layer, all of them split by ONE launch: with a single image the layers split theirs one by one)
  Arena ar(ws, ws_bytes);
  const size_t img_floats = ff_wimg_floats(L > 1 ? p->dim * p->factor : p->dim);
  float* wimg[SplitJobs::MAX] = {nullptr, nullptr, nullptr, nullptr};
  bool all_imgs = ws != nullptr && L <= SplitJobs::MAX;
  for (int l = 0; l < (all_imgs ? L : 1) && ws; ++l) {
    float* t = ar.take(img_floats);
    if (!t) { all_imgs = false; break; }
    wimg[l] = t;
  }
  if (all_imgs) {
    SplitJobs sj;
    for (int l = 0; l < L; ++l) {
      RPDE_CHECK_ARG(p->weights[l], "feedforward_fwd: null layer %d weight", l);
      if (worth_presplit(P, ff_out(p, l), ff_in(p, l))) sj.add(p->weights[l], 1, ff_in(p, l), ff_out(p, l), ff_in(p, l), wimg[l]);
    }
    RPDE_TRY(split_weights_multi(sj, st));
  }
  for (int l = 0; l < L; ++l) {
    const bool last = l == L - 1;
    RPDE_CHECK_ARG(p->weights[l] && (last || hs[l]), "feedforward_fwd: null layer %d buffers", l);
    const float* in = l == 0 ? x : hs[l - 1];
    RPDE_TRY(linear_fwd_impl(in, p->weights[l], p->biases ? p->biases[l] : nullptr, last ? z_last : hs[l], P, ff_in(p, l),
                             ff_out(p, l), last ? RPDE_ACT_IDENTITY : RPDE_ACT_GELU, (last || !ds) ? nullptr : ds[l],
                             p->dropout_p, layer_seed(p->seed, l), st, all_imgs ? wimg[l] : wimg[0], p->seed_epoch, all_imgs));
  }
  return ff_tail_fwd(z_last, residual, out, P, p->dim, p->layer_norm, p->ln_eps, p->ln_gamma, p->ln_beta,
                     make_drop(p->dropout_p, layer_seed(p->seed, L - 1), p->seed_epoch), p->post_act, st);
}

int rpde_feedforward_bwd(const rpde_ff_params* p, const float* x, const float* const* hs, const float* const* ds,
                         const float* z_last, const float* grad_out, float* grad_x, float* const* grad_weights,
                         float* const* grad_biases, float* grad_gamma, float* grad_beta, int64_t P, void* ws,
                         size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(p && x && z_last && grad_out && P > 0, "feedforward_bwd: bad arguments");
  RPDE_CHECK_ARG(p->n_layers == 1 || (hs && ds), "feedforward_bwd: saved hidden tensors missing");
  RPDE_CHECK_ARG(P < (1L << 31), "feedforward_bwd: too many points for one call");
  hipStream_t st = as_stream(stream);
  const int L = p->n_layers;
  const int hid = L > 1 ? p->dim * p->factor : p->dim;
  RPDE_CHECK_ARG(L <= 8, "feedforward_bwd: %d layers", L);
  Arena ar(ws, ws_bytes);
  float* buf0 = ar.take((size_t)P * hid);
  float* buf1 = ar.take((size_t)P * hid);
  // per layer: weight-gradient slabs, column-sum slabs, weight image (rpde_feedforward_ws_bytes) -- the regions of layer
  // 0 are followed by the others', so the fused path below may use them as one region of the largest layer's size
  float* slabs_l[8]; float* csum_l[8]; float* wt_l[8];
  for (int l = 0; l < L; ++l) slabs_l[l] = ar.take(wgrad_ws_floats(P, ff_in(p, l), ff_out(p, l)));
  for (int l = 0; l < L; ++l) csum_l[l] = ar.take((size_t)(colsum_tiles(P) + REDUCE_CHUNKS) * hid);
  for (int l = 0; l < L; ++l) wt_l[l] = ar.take(ff_wimg_floats(hid));
  float* small = ar.take(colsum_ws_floats(P, hid));
  float* tailws = ar.take(ff_tail_bwd_ws_floats(P, p->dim));
  if (!ar.ok()) { set_error("feedforward_bwd: workspace too small (%zu bytes given)", ws_bytes); return RPDE_ERR_WORKSPACE; }
  void* wt = wt_l[0];

  if (ff3_fused_ok(p, P) && hs && hs[0] && hs[1]) {
    // ds given: they hold d = gelu'(u) * dropscale and hs hold h.  ds absent: recompute mode, hs hold u = dropout(z)
    const int recompute = !(ds && ds[0] && ds[1]);
    const int act_h = recompute ? RPDE_ACT_GELU : RPDE_ACT_IDENTITY;
    // one fused kernel for the LayerNorm / dropout adjoint and the whole data-gradient chain; then the three
    // weight-gradient GEMMs on what it stored
    float* dz3 = ar.take((size_t)P * p->dim);
    float* part = ar.take(ff3_fused_bwd_part_floats());
    void* fimg = ar.take(ff3_fused_ws_floats());
    if (!ar.ok()) { set_error("feedforward_bwd: workspace too small (%zu bytes given)", ws_bytes); return RPDE_ERR_WORKSPACE; }
    int grid = 0;
    RPDE_TRY(ff3_fused_bwd_launch(p, recompute ? hs : ds, recompute, z_last, grad_out, dz3, buf0, buf1, nullptr, part, &grid, P,
                                  fimg, st));
    // first layer: du1 feeds both the data gradient and the weight gradient -- one kernel reads it once for both
    const bool both = grad_x && grad_weights && wgrad_h2_dgrad_ok(P, hid, p->dim);
    if (grad_x && !both) RPDE_TRY(linear_dgrad_impl(buf1, p->weights[0], grad_x, P, p->dim, hid, nullptr, nullptr, st, wt));
    if (grad_weights) {
      // each weight gradient in its own slab region, the three folds in one launch
      FoldJobs folds;
      RPDE_TRY(linear_wgrad_impl(hs[1], dz3, grad_weights[2], nullptr, P, hid, p->dim, slabs_l[2], small, st, act_h, &folds));
      RPDE_TRY(linear_wgrad_impl(hs[0], buf0, grad_weights[1], nullptr, P, hid, hid, slabs_l[1], small, st, act_h, &folds));
      if (both) RPDE_TRY(wgrad_h2_dgrad(buf1, x, p->weights[0], grad_weights[0], grad_x, P, p->dim, hid, slabs_l[0], st, &folds));
      else RPDE_TRY(linear_wgrad_impl(x, buf1, grad_weights[0], nullptr, P, p->dim, hid, slabs_l[0], small, st, RPDE_ACT_IDENTITY, &folds));
      RPDE_TRY(fold_jobs(folds, st));
    }
    {   // per-workgroup partial sums -> the five small gradients, one launch
      ReduceSegs sg;
      memset(&sg, 0, sizeof(sg));
      sg.n = FF3_PART; sg.nseg = 5;
      const int off[5] = {0, 256, 512, 576, 640}, len[5] = {hid, hid, p->dim, p->dim, p->dim};
      float* dst[5] = {grad_biases ? grad_biases[0] : nullptr, grad_biases ? grad_biases[1] : nullptr,
                       grad_biases ? grad_biases[2] : nullptr, p->layer_norm ? grad_gamma : nullptr,
                       p->layer_norm ? grad_beta : nullptr};
      for (int k = 0; k < 5; ++k) { sg.off[k] = off[k]; sg.len[k] = len[k]; sg.dst[k] = dst[k]; }
      RPDE_TRY(reduce_slabs_seg(part, grid, FF3_PART, sg, st));
    }
    return RPDE_OK;
  }

  // tail: d(out) -> dz_{L-1}, d(gamma), d(beta) and, fused, the last layer's bias gradient
  // every fold of the pass (LayerNorm / bias sums of the tail, weight-gradient slabs, column sums) is collected and done
  // by one launch at the end; the transposed weight images of all layers are split by one launch at the start
  FoldJobs folds;
  float* dz = buf0;
  float* other = buf1;
  int bias_done = 0;      // grad_biases[l] already produced by the kernel that produced dz_l
  RPDE_TRY(ff_tail_bwd(z_last, grad_out, dz, P, p->dim, p->layer_norm, p->ln_eps, p->ln_gamma, p->ln_beta,
                       make_drop(p->dropout_p, layer_seed(p->seed, L - 1), p->seed_epoch), p->post_act, grad_gamma, grad_beta,
                       grad_biases ? grad_biases[L - 1] : nullptr, &bias_done, tailws, st, &folds));
  const bool all_imgs = L <= SplitJobs::MAX;
  if (all_imgs) {
    SplitJobs sj;
    for (int l = grad_x ? 0 : 1; l < L; ++l)
      if (worth_presplit(P, ff_in(p, l), ff_out(p, l))) sj.add(p->weights[l], 0, ff_in(p, l), ff_in(p, l), ff_out(p, l), wt_l[l]);
    RPDE_TRY(split_weights_multi(sj, st));
  }
  for (int l = L - 1; l >= 0; --l) {
    const int in_f = ff_in(p, l), out_f = ff_out(p, l);
    const float* in = l == 0 ? x : hs[l - 1];
    float* gb = (grad_biases && !bias_done) ? grad_biases[l] : nullptr;
    RPDE_TRY(linear_wgrad_impl(in, dz, grad_weights ? grad_weights[l] : nullptr, gb, P, in_f, out_f, slabs_l[l], small, st,
                               RPDE_ACT_IDENTITY, &folds));
    bias_done = 0;
    void* img = all_imgs ? wt_l[l] : wt;
    if (l > 0) {
      RPDE_CHECK_ARG(ds[l - 1], "feedforward_bwd: derivative of layer %d was not saved", l - 1);
      const bool fuse = grad_biases && can_fuse_colsum(dz, p->weights[l], other, ds[l - 1], P, in_f, out_f);
      RPDE_TRY(linear_dgrad_impl(dz, p->weights[l], other, P, in_f, out_f, ds[l - 1], fuse ? csum_l[l] : nullptr, st, img, all_imgs));
      if (fuse) {
        const int S = (int)colsum_tiles(P);
        if (!(S <= 4 * REDUCE_CHUNKS && folds.add(csum_l[l], grad_biases[l - 1], in_f, S, in_f)))
          RPDE_TRY(reduce_slabs_2pass(csum_l[l], grad_biases[l - 1], in_f, S, in_f, csum_l[l] + colsum_tiles(P) * in_f, st));
        bias_done = 1;
      }
      float* t = dz; dz = other; other = t;
    } else if (grad_x) {
      RPDE_TRY(linear_dgrad_impl(dz, p->weights[0], grad_x, P, in_f, out_f, nullptr, nullptr, st, img, all_imgs));
    }
  }
  return fold_jobs(folds, st);
}

// ------------------------------ nn.Linear ----------------------------------
size_t rpde_linear_ws_bytes(int64_t P, int in_f, int out_f) {
  return arena_bytes(wgrad_ws_floats(P, in_f, out_f)) + arena_bytes(colsum_ws_floats(P, out_f));
}

int rpde_linear_fwd(const float* x, const float* w, const float* b, float* out, int64_t P, int in_f, int out_f, void* stream) {
  RPDE_CHECK_ARG(x && w && out && P > 0 && in_f > 0 && out_f > 0 && P < (1L << 31), "linear_fwd: bad arguments");
  return linear_fwd_impl(x, w, b, out, P, in_f, out_f, RPDE_ACT_IDENTITY, nullptr, 0.f, 0, as_stream(stream));
}

int rpde_linear_bwd(const float* x, const float* w, const float* grad_out, float* grad_x, float* grad_w, float* grad_b,
                    int64_t P, int in_f, int out_f, void* ws, size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(x && w && grad_out && P > 0 && in_f > 0 && out_f > 0 && P < (1L << 31), "linear_bwd: bad arguments");
  hipStream_t st = as_stream(stream);
  Arena ar(ws, ws_bytes);
  float* slabs = ar.take(wgrad_ws_floats(P, in_f, out_f));
  float* small = ar.take(colsum_ws_floats(P, out_f));
  if (!ar.ok()) { set_error("linear_bwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(linear_wgrad_impl(x, grad_out, grad_w, grad_b, P, in_f, out_f, slabs, small, st));
  if (grad_x) RPDE_TRY(linear_dgrad_impl(grad_out, w, grad_x, P, in_f, out_f, nullptr, nullptr, st));
  return RPDE_OK;
}

// --------------------------- 1x1 conv, channels-first -----------------------
// out[b][Cout,S] (+)= W[Cout,Cin] . act_in(x[b])[Cin,S] + bias
// weight gradient: one slab per (batch entry, reduction split); the split only when the batch alone leaves the chip idle
static int conv1x1_ksplit(int B, int64_t S) {
  int ks = 1;
  while (ks < 8 && B * ks < 64 && S / (2 * ks) >= 128) ks *= 2;
  return ks;
}
size_t rpde_conv1x1_ws_bytes(int B, int Cin, int Cout, int64_t S) {
  return arena_bytes((size_t)B * conv1x1_ksplit(B, S) * Cout * Cin) + arena_bytes((size_t)B * Cout * 64);
}

int rpde_conv1x1_fwd(const float* x, const float* w, const float* b, float* out, int B, int Cin, int Cout, int64_t S,
                     int act_in, int accumulate, void* stream) {
  return rpde_conv1x1_act_fwd(x, w, b, out, B, Cin, Cout, S, act_in, accumulate, RPDE_ACT_IDENTITY, stream);
}

int rpde_conv1x1_act_fwd(const float* x, const float* w, const float* b, float* out, int B, int Cin, int Cout, int64_t S,
                         int act_in, int accumulate, int act_out, void* stream) {
  RPDE_CHECK_ARG(x && w && out && B > 0 && Cin > 0 && Cout > 0 && S > 0 && S < (1L << 31), "conv1x1_fwd: bad arguments");
  // (a thread walks all input channels for its four points: worth it once there are enough points to fill the chip --
  //  the small 1-D configurations measured slower than the GEMM path)
  if (conv1x1_small_ok(x, out, Cin, Cout, S) && B <= 65535 && (long)B * S >= (1L << 20))
    return conv1x1_small(x, w, b, out, B, Cin, Cout, S, act_in, accumulate, act_out, as_stream(stream));
  rpde_gemm_desc d = gemm_desc();
  d.write_act = act_out;
  d.A = w; d.a_kmajor = 1; d.lda = Cin;
  d.B = x; d.b_kmajor = 0; d.ldb = S;
  d.C = out; d.ldc = S;
  d.M = Cout; d.N = (int)S; d.K = Cin;
  d.batch = B; d.sB1 = (long)Cin * S; d.sC1 = (long)Cout * S;
  d.bias = b; d.bias_mode = b ? 2 : 0;
  d.act_b = act_in; d.accumulate = accumulate;
  return launch_gemm(d, as_stream(stream));
}

namespace rpde {
// gb[c] = sum_{b,s} g[b][c][s]: rows (b,c) of length S -> one sum per row (fixed-shape tree, so the result does
// not depend on scheduling), then fold over b
__global__ __launch_bounds__(256) void k_rowsum(const float* __restrict__ g, float* __restrict__ part, long S) {
  __shared__ float red[256];
  const float* r = g + (long)blockIdx.x * S;
  float acc = 0.f;
  if ((S & 3) == 0 && (reinterpret_cast<uintptr_t>(r) & 15) == 0) {
    const float4* r4 = reinterpret_cast<const float4*>(r);
    for (long i = threadIdx.x; i < (S >> 2); i += 256) { const float4 v = r4[i]; acc += (v.x + v.y) + (v.z + v.w); }
  } else {
    for (long i = threadIdx.x; i < S; i += 256) acc += r[i];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void k_fold_bias(const float* __restrict__ part, float* __restrict__ gb, int B, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += part[(long)b * C + c];
  gb[c] = acc;
}
}  // namespace rpde

int rpde_conv1x1_bwd(const float* x, const float* w, const float* grad_out, float* grad_x, float* grad_w, float* grad_b,
                     int B, int Cin, int Cout, int64_t S, int act_in, int accumulate_gx, void* ws, size_t ws_bytes,
                     void* stream) {
  RPDE_CHECK_ARG(x && w && grad_out && B > 0 && Cin > 0 && Cout > 0 && S > 0 && S < (1L << 31), "conv1x1_bwd: bad arguments");
  hipStream_t st = as_stream(stream);
  Arena ar(ws, ws_bytes);
  const int ks = conv1x1_ksplit(B, S);
  float* slabs = ar.take((size_t)B * ks * Cout * Cin);
  float* part = ar.take((size_t)B * Cout * 64);
  if (!ar.ok()) { set_error("conv1x1_bwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  if (grad_w) {
    // gw[o,i] = sum_b sum_s g[b][o][s] * act(x[b][i][s]): one slab per batch entry, reduced over b
    rpde_gemm_desc d = gemm_desc();
    d.A = grad_out; d.a_kmajor = 1; d.lda = S;
    d.B = x; d.b_kmajor = 1; d.ldb = S;
    d.C = slabs; d.ldc = Cin;
    d.M = Cout; d.N = Cin; d.K = (int)S;
    d.batch = B; d.sA1 = (long)Cout * S; d.sB1 = (long)Cin * S; d.sC1 = (long)ks * Cout * Cin;
    d.ksplit = ks; d.sCk = (long)Cout * Cin;
    d.act_b = act_in;
    RPDE_TRY(launch_gemm(d, st));
    RPDE_TRY(reduce_slabs(slabs, grad_w, (long)Cout * Cin, B * ks, (long)Cout * Cin, 1.f, 0, st));
  }
  if (grad_b) {
    hipLaunchKernelGGL(k_rowsum, dim3(B * Cout), dim3(256), 0, st, grad_out, part, (long)S);
    RPDE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fold_bias, dim3((Cout + 63) / 64), dim3(64), 0, st, part, grad_b, B, Cout);
    RPDE_LAUNCH_CHECK();
  }
  if (grad_x) {
    // gx[b][i,s] = sum_o W[o,i] * g[b][o,s]   (through act'(x) when act_in is set)
    rpde_gemm_desc d = gemm_desc();
    d.A = w; d.a_kmajor = 0; d.lda = Cin;
    d.B = grad_out; d.b_kmajor = 0; d.ldb = S;
    d.C = grad_x; d.ldc = S;
    d.M = Cin; d.N = (int)S; d.K = Cout;
    d.batch = B; d.sB1 = (long)Cout * S; d.sC1 = (long)Cin * S;
    if (act_in) { d.epi_dact = act_in; d.aux = x; d.ldaux = S; }
    d.accumulate = accumulate_gx;
    RPDE_TRY(launch_gemm(d, st));
  }
  return RPDE_OK;
}

}  // extern "C"
