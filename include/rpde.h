/*
 * rpde.h -- C ABI of librpde_hip.so: the MI355X (gfx950) implementation of the
 * spectral-convolution hot path of RohanVKashyap/resolution-pde.
 *
 * Boundary rules:
 *   - plain C, raw DEVICE pointers + sizes + a hipStream_t (passed as void*);
 *     no torch types, no exceptions; every entry point returns 0 on success or
 *     a negative rpde_status; rpde_last_error() gives the thread-local message.
 *   - the caller (PyTorch) owns every tensor, including workspaces and the
 *     tensors saved for backward; the library owns only DFT plans (small
 *     device tables) inside opaque handles / an internal mutex-guarded cache.
 *   - all tensors are fp32 and contiguous in the stated layout; complex64
 *     parameters are passed as their (re,im)-interleaved float storage.
 *   - entry points are re-entrant per device and asynchronous on `stream`.
 *
 * Each entry point cites the reference code it replaces (paths are relative to
 * the reference repository root).
 */
#ifndef RPDE_H
#define RPDE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  RPDE_OK = 0,
  RPDE_ERR_ARG = -1,        /* bad shape / null pointer / unsupported option    */
  RPDE_ERR_HIP = -2,        /* a HIP runtime call or launch failed              */
  RPDE_ERR_WORKSPACE = -3,  /* workspace smaller than the *_ws_bytes() answer   */
  RPDE_ERR_MODES = -4       /* modes exceed the spectrum (SpectralConv* quirk Q5) */
} rpde_status;

/* fft_norm as torch.fft spells it */
enum { RPDE_NORM_BACKWARD = 0, RPDE_NORM_ORTHO = 1, RPDE_NORM_FORWARD = 2 };
/* FSpectralConv* `mode` (models/spectral_convolution.py:187-196, 273-281) */
enum { RPDE_MODE_FULL = 0, RPDE_MODE_LOWPASS = 1 };
/* activations (models/spectral_convolution.py:104-106, fno_blocks.py:33,71) */
enum { RPDE_ACT_IDENTITY = 0, RPDE_ACT_GELU = 1, RPDE_ACT_RELU = 2 };
enum { RPDE_EPI_MULAUX = 100 };

const char* rpde_last_error(void);
int rpde_version(void);

/* ---- DFT plans ---------------------------------------------------------
 * Truncated real-DFT tables for (n, modes, norm): the R2C transform restricted
 * to bins [0,modes) and the C2R transform of a spectrum that is zero beyond
 * them (Im(DC)/Im(Nyquist) ignored, as torch.fft.irfft does).  Replaces the
 * torch.fft.rfft / irfft calls at models/spectral_convolution.py:41,54,165,198,
 * 265,284,289,308.  The op entry points below look plans up in an internal
 * cache; create/destroy are exported for callers that want to pre-plan
 * (multi-resolution batches, SURVEY hard part 6). */
typedef struct rpde_plan rpde_plan;
int rpde_plan_create(rpde_plan** plan, int n, int modes, int norm, void* stream);
int rpde_plan_destroy(rpde_plan* plan);
/* copies of the float tables for tests: analysis [2*kp, ldn], synthesis [n, 2*kp] */
int rpde_plan_info(const rpde_plan* plan, int* n, int* modes, int* kp, int* ldn);
/* number of plans the per-process cache holds (every spectral entry point below takes its plans from that cache and
 * builds a missing one on first use: hipMalloc + one stream synchronisation).  Constant across training steps once
 * rpde.ops.warm_plans() has seen the run's grid sizes. */
int rpde_plan_cache_count(void);
int rpde_plan_tables(const rpde_plan* plan, float* analysis_host, float* synthesis_host);

/* ---- generic strided batched GEMM (fp32 MFMA) -----------------------------
 * C[z][m,n] (+)= alpha * sum_k A[z][m,k] * B[z][k,n]  with fused prologue /
 * epilogue.  Everything heavy below is expressed through it; exported so the
 * parity tests and bench.py can time the kernel itself. */
typedef struct {
  const float* A; const float* B; float* C;
  int M, N, K;
  int a_kmajor;            /* 1: A[m*lda+k] (k contiguous)   0: A[k*lda+m]          */
  int b_kmajor;            /* 1: B^T stored, B[n*ldb+k]      0: B[k*ldb+n]          */
  int64_t lda, ldb, ldc;
  int batch, zdiv;         /* z in [0,batch): z1 = z / zdiv, z2 = z % zdiv          */
  int64_t sA1, sA2, sB1, sB2, sC1, sC2;
  int ksplit;              /* >1: split K, partial s goes to C + s*sCk (no epilogue) */
  int64_t sCk;
  float alpha;
  int accumulate;          /* C += result                                           */
  const float* bias; int bias_mode;   /* 0 none, 1 bias[n], 2 bias[m]                */
  /* activation applied to an operand while it is staged (h = act(drop(z))) */
  int act_a, act_b;        /* RPDE_ACT_*                                            */
  /* epilogue: C = acc * act'(drop(aux)) * dropscale   (backward through act), or with
   * RPDE_EPI_MULAUX: C = acc * aux (aux holds a stored derivative)                */
  int epi_dact;            /* RPDE_ACT_*, RPDE_EPI_MULAUX or 0                      */
  const float* aux; int64_t ldaux;
  /* dropout shared by prologue / epilogue: element id = point*drop_ld + feature;
   * drop_where: bit 0 mask the A operand, bit 1 the B operand, bit 2 the epilogue's aux */
  float drop_p; uint64_t drop_seed; int64_t drop_ld;
  int write_act;           /* epilogue applies act (RPDE_ACT_*) to the stored value */
  int drop_where;
  /* optional: per-M-tile column sums of the stored C, [ceil(M/128)][N] floats (bias gradients
   * for free); needs M > 64, batch = ksplit = 1 and 16-byte aligned rows */
  float* colsum;
  /* with write_act and a non-null aux_out (same layout as C): the epilogue evaluates the activation
   * once per element, u = dropout(acc+bias): C = act(u), aux_out = act'(u) * dropscale -- so every
   * consumer of the hidden activation and of its derivative is a plain GEMM (drop_where bit 2) */
  float* aux_out;
  /* optional accelerator for a B operand shared by the whole batch (a weight matrix): the images written by
   * rpde_split_weights for this B (same N, K).  Results are bit-identical with and without it; B must
   * still be valid (the native fp32 kernels use it when the split-bf16 path does not apply). */
  const void* b_split;
  /* the same for an A operand shared by the whole batch (e.g. a DFT table): rpde_split_weights(A, a_kmajor,
   * lda, M, K).  With it and an x-major B, K need not be a multiple of 32. */
  const void* a_split;
  /* with accumulate: add this tensor (same layout and strides as C) instead of the old contents of C,
   * i.e. C = alpha*A.B + acc_src -- sums a skip-connection gradient without a separate pass */
  const float* acc_src;
  /* optional device counter folded into drop_seed by the kernel (8 bytes, read once per launch): lets a captured
   * hipGraph draw new masks at every replay -- see rpde_ff_params.seed_epoch */
  const uint64_t* drop_epoch;
} rpde_gemm_desc;
int rpde_gemm_f32(const rpde_gemm_desc* d, void* stream);

/* Pre-split a weight operand for the split-bf16 GEMM path: w is [N,K] (kmajor=1, row stride ld) or [K,N]
 * (kmajor=0); out receives rpde_split_weights_bytes(N,K) bytes: three bf16 images hi/mid/lo with
 * w = hi+mid+lo exactly, laid out as the kernel's LDS stages (k zero-padded to a multiple of 32). */
size_t rpde_split_weights_bytes(int N, int K);
int rpde_split_weights(const float* w, int kmajor, int64_t ld, int N, int K, void* out, void* stream);

/* ---- FSpectralConv1d.forward_fourier  (models/spectral_convolution.py:158-204)
 * x,out [B,n,C] channels-last; w [C,C,K,2]; keff=min(K,n/2+1) is clamped here
 * (quirk Q5).  spec_in [B, 2*kp, C] (kp = keff rounded up to 4) receives the
 * truncated input spectrum and must be kept for the backward call. */
size_t rpde_fspectral1d_ws_bytes(int B, int n, int C, int K);
size_t rpde_fspectral1d_spec_elems(int B, int n, int C, int K);
int rpde_fspectral1d_fwd(const float* x, const float* w, float* out, float* spec_in,
                         int B, int n, int C, int K, int mode, int norm,
                         void* ws, size_t ws_bytes, void* stream);
/* grad_skip (nullable, shaped like x): added into grad_x by the last GEMM's epilogue -- the gradient
 * arriving through a skip connection around the layer (x + FF(spectral(x))) costs no extra pass */
int rpde_fspectral1d_bwd(const float* grad_out, const float* spec_in, const float* w,
                         float* grad_x, float* grad_w, const float* grad_skip,
                         int B, int n, int C, int K, int mode, int norm,
                         void* ws, size_t ws_bytes, void* stream);

/* ---- FSpectralConv2d.forward_fourier  (models/spectral_convolution.py:256-318)
 * x,out [B,M,N,C]; w_y,w_x [C,C,K,2]; norm 'ortho' (hard-coded in the
 * reference, quirk Q4).  spec_y [B*M, 2*kpy, C], spec_x [B*N, 2*kpx, C]. */
size_t rpde_fspectral2d_ws_bytes(int B, int M, int N, int C, int K);
size_t rpde_fspectral2d_spec_elems(int B, int M, int N, int C, int K, int axis /*0:y 1:x*/);
int rpde_fspectral2d_fwd(const float* x, const float* w_y, const float* w_x, float* out,
                         float* spec_y, float* spec_x,
                         int B, int M, int N, int C, int K, int mode,
                         void* ws, size_t ws_bytes, void* stream);
int rpde_fspectral2d_bwd(const float* grad_out, const float* spec_y, const float* spec_x,
                         const float* w_y, const float* w_x,
                         float* grad_x, float* grad_wy, float* grad_wx, const float* grad_skip,
                         int B, int M, int N, int C, int K, int mode,
                         void* ws, size_t ws_bytes, void* stream);
/* The same for FSpectralConv2d.forward_fourier in evaluation: rpde_fspectral2d_prep_bytes (0: nothing to prepare for
 * this shape, use rpde_fspectral2d_fwd) bytes hold the mode-mix weight fragments of both axes; the forward then needs
 * rpde_fspectral2d_eval_ws_bytes of workspace (the spectra live there: nothing is saved for a backward). */
size_t rpde_fspectral2d_prep_bytes(int M, int N, int C, int K);
size_t rpde_fspectral2d_eval_ws_bytes(int B, int M, int N, int C, int K);
int rpde_fspectral2d_prepare(const float* w_y, const float* w_x, int M, int N, int C, int K,
                             void* prep, size_t prep_bytes, void* stream);
int rpde_fspectral2d_fwd_prepared(const float* x, const void* prep, float* out, int B, int M, int N, int C, int K,
                                  void* ws, size_t ws_bytes, void* stream);

/* ---- SpectralConv1d.forward  (models/spectral_convolution.py:38-55)
 * x [B,Cin,n] channels-first, w [Cin,Cout,K] complex64 (interleaved floats),
 * out [B,Cout,n]; norm 'backward'.  K > n/2+1 -> RPDE_ERR_MODES (quirk Q5).
 * act_in: activation applied to x while it is read (the FNO block feeds
 * act(previous pre-activation), fno_blocks.py:33).  spec_in [B*Cin, 2*kp]. */
size_t rpde_spectral1d_ws_bytes(int B, int Cin, int Cout, int n, int K);
int rpde_spectral1d_fwd(const float* x, const float* w, float* out, float* spec_in,
                        int B, int Cin, int Cout, int n, int K, int act_in,
                        void* ws, size_t ws_bytes, void* stream);
int rpde_spectral1d_bwd(const float* grad_out, const float* spec_in, const float* w,
                        const float* x, float* grad_x, float* grad_w,
                        int B, int Cin, int Cout, int n, int K, int act_in,
                        void* ws, size_t ws_bytes, void* stream);

/* ---- SpectralConv2d.forward  (models/spectral_convolution.py:79-98)
 * x [B,Cin,M,N], w1,w2 [Cin,Cout,m1,m2] complex64, out [B,Cout,M,N]; rows
 * [0,m1) use w1, rows [M-m1,M) use w2 and win on overlap (quirk Q6).
 * spec_in [B*Cin, 2*R, m2p] planar (R = 2*m1 retained rows). */
size_t rpde_spectral2d_ws_bytes(int B, int Cin, int Cout, int M, int N, int m1, int m2);
size_t rpde_spectral2d_spec_elems(int B, int Cin, int M, int N, int m1, int m2);
int rpde_spectral2d_fwd(const float* x, const float* w1, const float* w2, float* out, float* spec_in,
                        int B, int Cin, int Cout, int M, int N, int m1, int m2, int act_in,
                        void* ws, size_t ws_bytes, void* stream);
int rpde_spectral2d_bwd(const float* grad_out, const float* spec_in, const float* w1, const float* w2,
                        const float* x, float* grad_x, float* grad_w1, float* grad_w2,
                        int B, int Cin, int Cout, int M, int N, int m1, int m2, int act_in,
                        void* ws, size_t ws_bytes, void* stream);

/* ---- FeedForward  (models/custom_layer.py:49-68) + the residual glue of
 * FFNO*.forward (models/ffno.py:118,230) and FSpectralConv1d's act (:154).
 * x [P,dim] channels-last points.  Hidden layer l < L-1: z_l = in_l @ W_l^T + b_l
 * never reaches HBM; the GEMM epilogue evaluates the activation once per element,
 * u = dropout(z_l): hs[l] = gelu(u) ([P,out_l], input of the next GEMM and of the
 * weight-gradient GEMM) and ds[l] = gelu'(u) * dropscale (what backward multiplies
 * by; pass ds = NULL or ds[l] = NULL when no gradient is needed).  The last layer
 * stores z_last ([P,dim]) and out = residual + post_act( LayerNorm( dropout(z_last) ) )
 * (LN optional, residual may be NULL).  Dropout masks come from a counter hash of
 * (seed, layer, element). */
typedef struct {
  int n_layers; int dim; int factor;
  int layer_norm; float ln_eps;
  float dropout_p; uint64_t seed;      /* p = 0 in eval mode                         */
  int post_act;                        /* RPDE_ACT_*                                 */
  const float* const* weights;         /* [n_layers] W_l [out_l, in_l]               */
  const float* const* biases;          /* [n_layers] b_l [out_l]                     */
  const float* ln_gamma; const float* ln_beta;
  /* optional (may be NULL): device address of a 64-bit counter that every kernel of the call mixes into the seed.
   * A hipGraph replays its launch arguments, so a host-drawn seed alone would repeat one mask for ever; the training
   * step advances this counter on the device once per step (between backward and the next forward), and forward and
   * backward of a step read the same value.  The reference draws its masks from torch's generator
   * (models/custom_layer.py:60, nn.Dropout): parity is statistical either way. */
  const uint64_t* seed_epoch;
} rpde_ff_params;
size_t rpde_feedforward_ws_bytes(int64_t P, int dim, int factor, int n_layers);      /* backward */
/* forward scratch (optional: ws may be NULL; with it each weight is pre-split once per call for the
 * split-bf16 GEMMs -- faster, bit-identical output) */
size_t rpde_feedforward_fwd_ws_bytes(int dim, int factor, int n_layers);
/* 1 when rpde_feedforward_fwd (given its scratch) runs this shape as ONE fused kernel (dim 64, factor 4, three
 * layers: hidden activations never reach HBM).  Pointer contract of the fused path, forward and backward alike:
 *   hs and ds NULL ............ evaluation: only `out` is written;
 *   hs[0..1] and ds[0..1] ..... training: h and d = gelu'(u) * dropscale of both hidden layers and z_last are
 *                               stored by the forward and consumed by rpde_feedforward_bwd;
 *   hs[0..1] only (ds NULL) ... training in recompute mode: hs receive u = dropout(z) and the backward kernels
 *                               re-evaluate gelu / gelu' from them (half the saved bytes; measured slower).
 * The weight gradients of these shapes come from the streaming kernel of csrc/wgrad_h2.hip (for the first layer
 * together with grad_x), bias / gamma / beta gradients from per-workgroup partial sums folded in fixed order. */
int rpde_feedforward_is_fused(int dim, int factor, int n_layers, int64_t P);
int rpde_feedforward_fwd(const rpde_ff_params* p, const float* x, const float* residual,
                         float* const* hs, float* const* ds, float* z_last, float* out, int64_t P,
                         void* ws, size_t ws_bytes, void* stream);
int rpde_feedforward_bwd(const rpde_ff_params* p, const float* x, const float* const* hs,
                         const float* const* ds, const float* z_last,
                         const float* grad_out, float* grad_x,
                         float* const* grad_weights, float* const* grad_biases,
                         float* grad_gamma, float* grad_beta, int64_t P,
                         void* ws, size_t ws_bytes, void* stream);
/* Evaluation with frozen weights (rollouts, all-resolution sweeps, validation): the fused kernel's weight fragments are
 * built once by rpde_feedforward_prepare into a caller-owned buffer of rpde_feedforward_fwd_ws_bytes bytes and reused
 * by rpde_feedforward_fwd_prepared until the caller knows the weights have changed.  Fused shapes only
 * (rpde_feedforward_is_fused); dropout_p must be 0. */
int rpde_feedforward_prepare(const rpde_ff_params* p, void* prep, size_t prep_bytes, void* stream);
int rpde_feedforward_fwd_prepared(const rpde_ff_params* p, const float* x, const float* residual, float* out, int64_t P,
                                  const void* prep, size_t prep_bytes, void* stream);

/* ---- pointwise linear, channels-last: nn.Linear / WNLinear applied to
 * [P,in] (models/ffno.py:113,121,225,233; custom_layer.py:70).  Weight-norm is
 * resolved by the caller into an effective weight. */
size_t rpde_linear_ws_bytes(int64_t P, int in_f, int out_f);
int rpde_linear_fwd(const float* x, const float* w, const float* b, float* out,
                    int64_t P, int in_f, int out_f, void* stream);
int rpde_linear_bwd(const float* x, const float* w, const float* grad_out,
                    float* grad_x, float* grad_w, float* grad_b,
                    int64_t P, int in_f, int out_f, void* ws, size_t ws_bytes, void* stream);

/* Weight normalisation of WNLinear (models/custom_layer.py:70-108: torch.nn.utils.weight_norm on dim 0, parameters
 * weight_g [out,1], weight_v [out,in]):  w = v * (g / |v|_row), and its adjoint -- one launch each where autograd runs
 * a dozen ATen kernels per projection.  grad_v or grad_g may be null. */
int rpde_weight_norm_fwd(const float* v, const float* g, float* w, int out_f, int in_f, void* stream);
int rpde_weight_norm_bwd(const float* v, const float* g, const float* grad_w, float* grad_v, float* grad_g,
                         int out_f, int in_f, void* stream);

/* ---- 1x1 convolution, channels-first: nn.Conv1d/Conv2d(k=1) lifting, bypass
 * and projection (models/fno.py:30,93; fno_blocks.py:29,38-39,67,76-77).
 * x [B,Cin,S], w [Cout,Cin], out [B,Cout,S] (S = flattened spatial size).
 * act_in is applied to x while staged; accumulate adds into out (the FNO
 * block's spectral_conv(x) + bypass_conv(x)). */
size_t rpde_conv1x1_ws_bytes(int B, int Cin, int Cout, int64_t S);
int rpde_conv1x1_fwd(const float* x, const float* w, const float* b, float* out,
                     int B, int Cin, int Cout, int64_t S, int act_in, int accumulate, void* stream);
/* the same with an activation applied to the result before it is stored (evaluation-mode FNO blocks:
 * out = act(spectral + W . x + b) written once, models/fno_blocks.py:25-45 of the reference) */
int rpde_conv1x1_act_fwd(const float* x, const float* w, const float* b, float* out,
                         int B, int Cin, int Cout, int64_t S, int act_in, int accumulate, int act_out, void* stream);
/* evaluation-mode FNOBlock2d in one entry point: out = act_out(SpectralConv2d(x; w1, w2) + Conv2d_1x1(x; wc, bc)),
 * reference models/fno_blocks.py:63-83 with models/spectral_convolution.py:79-98.  The inverse DFT along the last
 * axis, the bypass convolution, bias and activation run as ONE streaming pass over x: the spectral branch is never
 * written.  x [B,Cin,M,N], w1/w2 complex [Cin,Cout,m1,m2] (as float pairs), wc [Cout,Cin], out [B,Cout,M,N].
 * rpde_fnoblock2d_eval_ok: 1 when the shape is covered (Cout <= 32, N % 4 == 0, N divides 1024 or is a multiple of it, and the
 * block's weights plus its rows' 2 * ceil4(m2) spectrum entries fit 64 KB of LDS). */
size_t rpde_fnoblock2d_eval_ws_bytes(int B, int Cin, int Cout, int M, int N, int m1, int m2);
int rpde_fnoblock2d_eval_ok(int Cin, int Cout, int M, int N, int m2);
int rpde_fnoblock2d_eval_fwd(const float* x, const float* w1, const float* w2, const float* wc, const float* bc, float* out,
                             int B, int Cin, int Cout, int M, int N, int m1, int m2, int act_out,
                             void* ws, size_t ws_bytes, void* stream);
/* the LAST block of an FNO2d and its projection MLP in evaluation, one entry point (reference models/fno.py:143-150:
 * fno_blocks[-1] -> projection = mlp2(gelu(mlp1(.)))): out [B,Cq,M,N] = pw2 . gelu(pw1 . act_out(SpectralConv2d(x) +
 * Conv2d_1x1(x)) + pb1) + pb2 with pw1 [Cmid,Cout], pw2 [Cq,Cmid].  The block's output never reaches HBM.  Workspace:
 * rpde_fnoblock2d_eval_ws_bytes.  _ok: Cin = 32, Cout <= 32, Cmid <= 128, Cq <= 4, N % 64 == 0, N <= 1024, m2 <= 16. */
int rpde_fnoblock2d_proj_eval_ok(int Cin, int Cout, int M, int N, int m1, int m2, int Cmid, int Cq);
int rpde_fnoblock2d_proj_eval_fwd(const float* x, const float* w1, const float* w2, const float* wc, const float* bc,
                                  const float* pw1, const float* pb1, const float* pw2, const float* pb2, float* out, int B,
                                  int Cin, int Cout, int M, int N, int m1, int m2, int act_out, int Cmid, int Cq,
                                  void* ws, size_t ws_bytes, void* stream);
/* FNO2d.forward in evaluation, up to and including the first block (reference models/fno.py:121-147:
 * cat(x, gridx, gridy) -> lifting -> fno_blocks[0]), without the lifted field ever being written or read:
 * u [B,1,M,N], gx [M], gy [N] (the grid coordinates, device arrays), wl [C,3], bl [C] (lifting conv), then the block's
 * parameters as for rpde_fnoblock2d_eval_fwd.  out [B,Cout,M,N]. */
size_t rpde_fno2d_lift_block_eval_ws_bytes(int B, int C, int Cout, int M, int N, int m1, int m2);
int rpde_fno2d_lift_block_eval_ok(int Cu, int C, int Cout, int M, int N, int m1, int m2);
int rpde_fno2d_lift_block_eval_fwd(const float* u, const float* gx, const float* gy, const float* wl, const float* bl,
                                   const float* w1, const float* w2, const float* wc, const float* bc, float* out, int B, int C,
                                   int Cout, int M, int N, int m1, int m2, int act_out, void* ws, size_t ws_bytes, void* stream);
/* the projection MLP mlp2(gelu(mlp1(act_in(x)))) of models/fno_blocks.py:38-45,76-83 in one pass, EVALUATION only
 * (the hidden tensor [B,Cmid,S] is never written; training keeps the two convolutions, whose backward needs it):
 * x [B,Cin,S], w1 [Cmid,Cin], b1 [Cmid], w2 [Cout,Cmid], b2 [Cout], out [B,Cout,S].  rpde_conv_mlp_ok: 1 when the
 * shape is covered (Cin 32 with Cmid <= 128, or Cin 64 with Cmid <= 64; Cout <= 4; S % 16 == 0). */
int rpde_conv_mlp_ok(int Cin, int Cmid, int Cout, int64_t S);
int rpde_conv_mlp_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* out,
                      int B, int Cin, int Cmid, int Cout, int64_t S, int act_in, void* stream);
int rpde_conv1x1_bwd(const float* x, const float* w, const float* grad_out,
                     float* grad_x, float* grad_w, float* grad_b,
                     int B, int Cin, int Cout, int64_t S, int act_in, int accumulate_gx,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- grid channels + layout change at the model boundary
 * (models/ffno.py:92,201-222; fno.py:51,121-139): builds the lifted input
 * [.., Cin+G] (channels-last) or [B,Cin+G,S] (channels-first) with
 * endpoint-inclusive linspace coordinates (quirk Q10) generated on device. */
int rpde_concat_grid(const float* x, float* out, int B, int Cin, int M, int N /*1 for 1-D*/,
                     int grid_dims /*0,1,2*/, double lo, double hi, int channels_last,
                     const float* gridx, const float* gridy, void* stream);
/* [B,S,C] <-> [B,C,S] */
int rpde_transpose_cs(const float* in, float* out, int B, int64_t S, int C, int to_channels_first, void* stream);

/* ---- spectral resize: rfft -> shared bins -> irfft at the new size, x out/in
 * (reference: utils/res_utils.py:29-50 `resize`, :93-125 `resize_1d`; used by the
 * all-resolution evaluators, utils/naive_utils.py, utils/resize_utils.py).
 * x [rows, n_in] -> out [rows, n_out];  x [rows, M, N] -> out [rows, Mo, No]. */
size_t rpde_resize1d_ws_bytes(int64_t rows, int n_in, int n_out);
int rpde_resize1d(const float* x, float* out, int64_t rows, int n_in, int n_out,
                  void* ws, size_t ws_bytes, void* stream);
size_t rpde_resize2d_ws_bytes(int64_t rows, int M, int N, int Mo, int No);
int rpde_resize2d(const float* x, float* out, int64_t rows, int M, int N, int Mo, int No,
                  void* ws, size_t ws_bytes, void* stream);

/* ---- elementwise activation: out = act(x); backward dx = g * act'(x) */
int rpde_act_fwd(const float* x, float* out, int64_t n, int act, void* stream);
int rpde_act_bwd(const float* x, const float* g, float* dx, int64_t n, int act, void* stream);

/* ---- RelativeL2Loss.forward (utils/loss.py:31-59): rel[b] = |x-y|_2/(|y|_2+1e-8).
 * rel [B]; loss scalar = mean (size_average) or sum; pass loss=NULL for
 * reduction=False.  stats (rpde_rel_l2_stats_elems(B) floats: per-sample diff
 * norm and y norm, then scratch partials) is kept for backward. */
int64_t rpde_rel_l2_stats_elems(int B);
int rpde_rel_l2_fwd(const float* x, const float* y, float* rel, float* loss, float* stats,
                    int B, int64_t per, int size_average, void* stream);
/* grad_rel [B] if reduction=False else NULL with grad_loss a device scalar */
int rpde_rel_l2_bwd(const float* x, const float* y, const float* stats,
                    const float* grad_loss, const float* grad_rel, float* grad_x,
                    int B, int64_t per, int size_average, void* stream);

/* ---- optimizer step: torch.optim.AdamW as built at main_1d.py:144 / main_2d.py:173 (decoupled weight decay,
 * bias-corrected moments, no amsgrad), one streaming kernel over flat fp32 buffers of n (multiple of 4) elements.
 * The caller passes the step's scalars: 1 - lr*wd, 1 - b1, b2, 1 - b2, lr / (1 - b1^t), sqrt(1 - b2^t), eps. */
int rpde_adamw_step(float* p, const float* g, float* m, float* v, int64_t n,
                    float one_minus_lr_wd, float one_minus_b1, float b2, float one_minus_b2,
                    float step_size, float bc2_sqrt, float eps, void* stream);
/* the same with the step state on the device (step_dev: 8 floats -- [0] t, incremented by the call, [1] lr/(1-b1^t),
 * [2] sqrt(1-b2^t), [3] lr, [4] weight decay, [5] 1 - lr*wd): nothing step-dependent crosses the host, so the step can be
 * captured in a hipGraph.  Called eagerly it stores its lr / weight_decay arguments in step_dev first; while its stream
 * is being captured it does not, and every replay runs with what rpde_adamw_set_hyper_dev put there -- a captured step
 * follows a learning-rate schedule (main_1d.py:145-151, main_2d.py:174-180) without being captured again. */
int rpde_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n,
                        float lr, float b1, float b2, float eps, float weight_decay,
                        float* step_dev, void* stream);
int rpde_adamw_set_hyper_dev(float* step_dev, float lr, float weight_decay, void* stream);
/* the update of rpde_adamw_step_dev without advancing the counter (a second buffer of the same optimizer step) */
int rpde_adamw_apply_dev(float* p, const float* g, float* m, float* v, int64_t n,
                         float lr, float b1, float b2, float eps, float weight_decay,
                         const float* step_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RPDE_H */
