// FSpectralConv1d / FSpectralConv2d forward_fourier and its adjoint on
// channels-last tensors, as truncated-DFT GEMMs (never a full spectrum):
//
//   analysis   A[row][2kp][C]  = Fa[2kp, n] . x[row][n][C]        (row = b, or (b,m), or (b,n))
//   mix        At[row][k][2C]  = A[row][k][2C] . Wblk[k][2C][2C]  (per retained mode k)
//   synthesis  out[row][n][C] (+)= Fs[n, 2kp] . At[row][2kp][C]
//
// The spectrum layout [row][k][re|im][c] makes the (re,im,c) reduction index of
// the mix contiguous, so the complex channel mixing is one real GEMM per mode.
// Backward runs the transposed tables (same device tables read x-major) and the
// transposed / split-K weight-gradient GEMMs.  Only K of n/2+1 bins are ever
// computed or stored: the spectra are ~15 % of the field size at K=20, n=256.
//
// Reference: models/spectral_convolution.py:158-204 (1-D), :256-318 (2-D).
#include "rpde_internal.h"
#include "plan.h"
#include "pointwise.h"
#include "fused_spectral.h"
#include "mix1d.h"

#include <stdlib.h>

namespace rpde {

struct Axis {
  // geometry of one transformed axis of a channels-last tensor
  int n;            // axis length
  int keff, kp;     // retained modes, padded to 4
  int rows;         // number of independent lines = batch of the DFT GEMMs
  int zdiv;         // rows are indexed (z / zdiv, z % zdiv)
  long s1, s2;      // element offset of line z in the field: z1*s1 + z2*s2
  long ld;          // stride between consecutive points of a line (in floats)
  const rpde_plan* plan;
};

static inline int split_for(long k_total, int tiles) {
  long s = 512 / (tiles > 0 ? tiles : 1);
  const long cap = (k_total + 127) / 128;
  if (s > cap) s = cap;
  if (s < 1) s = 1;
  if (s > 512) s = 512;
  return (int)s;
}

// spec[row][2kp][C] = table . field-line      (table = Fa, or Fs^T for the adjoint of synthesis)
static int dft_analysis(const Axis& ax, const float* field, float* spec, int C, bool adjoint_of_synthesis, hipStream_t st) {
  rpde_gemm_desc d = gemm_desc();
  if (!adjoint_of_synthesis) { d.A = ax.plan->fa; d.a_kmajor = 1; d.lda = ax.plan->ldn; d.a_split = ax.plan->img[IMG_FA]; }
  else { d.A = ax.plan->fs; d.a_kmajor = 0; d.lda = 2L * ax.kp; d.a_split = ax.plan->img[IMG_FST]; }
  d.B = field; d.b_kmajor = 0; d.ldb = ax.ld;
  d.C = spec; d.ldc = C;
  d.M = 2 * ax.kp; d.N = C; d.K = ax.n;
  d.batch = ax.rows; d.zdiv = ax.zdiv;
  d.sB1 = ax.s1; d.sB2 = ax.s2;
  d.sC1 = (long)ax.zdiv * 2 * ax.kp * C; d.sC2 = 2L * ax.kp * C;
  return launch_gemm(d, st);
}

// field-line (+)= table . spec[row]           (table = Fs, or Fa^T for the adjoint of analysis)
// accumulate: 0 overwrite, 1 add to field, 2 field = result + acc_src (a tensor laid out like field)
static int dft_synthesis(const Axis& ax, const float* spec, float* field, int C, bool adjoint_of_analysis, int accumulate,
                         hipStream_t st, const float* acc_src = nullptr) {
  rpde_gemm_desc d = gemm_desc();
  if (!adjoint_of_analysis) { d.A = ax.plan->fs; d.a_kmajor = 1; d.lda = 2L * ax.kp; d.a_split = ax.plan->img[IMG_FS]; }
  else { d.A = ax.plan->fa; d.a_kmajor = 0; d.lda = ax.plan->ldn; d.a_split = ax.plan->img[IMG_FAT]; }
  d.B = spec; d.b_kmajor = 0; d.ldb = C;
  d.C = field; d.ldc = ax.ld;
  d.M = ax.n; d.N = C; d.K = 2 * ax.kp;
  d.batch = ax.rows; d.zdiv = ax.zdiv;
  d.sB1 = (long)ax.zdiv * 2 * ax.kp * C; d.sB2 = 2L * ax.kp * C;
  d.sC1 = ax.s1; d.sC2 = ax.s2;
  d.accumulate = accumulate ? 1 : 0;
  d.acc_src = accumulate == 2 ? acc_src : nullptr;
  return launch_gemm(d, st);
}

// out[row][k][2C] = in[row][k][2C] . Wblk[k]   (transpose: . Wblk[k]^T)
static int mode_mix(const Axis& ax, const float* in, const float* wblk, float* out, int C, bool transpose, hipStream_t st) {
  rpde_gemm_desc d = gemm_desc();
  d.A = in; d.a_kmajor = 1; d.lda = 2L * ax.kp * C;
  d.B = wblk; d.b_kmajor = transpose ? 1 : 0; d.ldb = 2L * C;
  d.C = out; d.ldc = 2L * ax.kp * C;
  d.M = ax.rows; d.N = 2 * C; d.K = 2 * C;
  d.batch = ax.keff;
  d.sA1 = 2L * C; d.sB1 = 4L * C * C; d.sC1 = 2L * C;
  return launch_gemm(d, st);
}

// slabs[s][k][2C][2C] = sum_rows spec[row][k][:]^T . gspec[row][k][:]
static int mode_mix_wgrad(const Axis& ax, const float* spec, const float* gspec, float* slabs, int C, int S, hipStream_t st) {
  rpde_gemm_desc d = gemm_desc();
  d.A = spec; d.a_kmajor = 0; d.lda = 2L * ax.kp * C;
  d.B = gspec; d.b_kmajor = 0; d.ldb = 2L * ax.kp * C;
  d.C = slabs; d.ldc = 2L * C;
  d.M = 2 * C; d.N = 2 * C; d.K = ax.rows;
  d.batch = ax.keff;
  d.sA1 = 2L * C; d.sB1 = 2L * C; d.sC1 = 4L * C * C;
  d.ksplit = S; d.sCk = (long)ax.keff * 4 * C * C;
  return launch_gemm(d, st);
}

static size_t spec_floats(const Axis& ax, int C) { return (size_t)ax.rows * 2 * ax.kp * C; }
static int wgrad_tiles(int C) { const int t = (2 * C + 127) / 128; return t * t; }

static int make_axis(Axis& ax, int n, int K, int norm, int rows, int zdiv, long s1, long s2, long ld, hipStream_t st) {
  ax.n = n;
  ax.keff = K < n / 2 + 1 ? K : n / 2 + 1;
  ax.kp = (ax.keff + 3) / 4 * 4;
  ax.rows = rows; ax.zdiv = zdiv; ax.s1 = s1; ax.s2 = s2; ax.ld = ld;
  return get_plan(&ax.plan, n, ax.keff, norm, 0, PLAN_REAL, st);
}

// geometry without touching the device (for the size queries)
static void axis_dims(Axis& ax, int n, int K, int rows) {
  ax.n = n; ax.keff = K < n / 2 + 1 ? K : n / 2 + 1; ax.kp = (ax.keff + 3) / 4 * 4; ax.rows = rows;
  ax.zdiv = 1; ax.s1 = ax.s2 = ax.ld = 0; ax.plan = nullptr;
}

static size_t axis_ws_bytes(const Axis& ax, int C) {
  // forward: Wblk + mixed spectrum;  backward: Wblk + g-spectrum + d-spectrum + weight-grad slabs
  const size_t wblk = arena_bytes((size_t)ax.keff * 4 * C * C);
  const size_t sp = arena_bytes(spec_floats(ax, C));
  const int S = split_for(ax.rows, wgrad_tiles(C) * ax.keff);
  const size_t slabs = arena_bytes((size_t)S * ax.keff * 4 * C * C);
  return wblk + 2 * sp + slabs;
}

// forward of one axis: field x -> out (+)=
static int axis_fwd(const Axis& ax, const float* x, const float* w, int K, float* out, float* spec_in, int C, int mode,
                    int accumulate, Arena& ar, hipStream_t st) {
  RPDE_TRY(dft_analysis(ax, x, spec_in, C, false, st));
  const float* syn_in = spec_in;
  if (mode == RPDE_MODE_FULL && mix1d_ok(ax.rows, C)) {
    // few rows (the 1-D layer): the mix reads the weights where they lie (mix1d.hip)
    float* mixed = ar.take(spec_floats(ax, C));
    if (!ar.ok()) { set_error("fspectral: workspace too small"); return RPDE_ERR_WORKSPACE; }
    RPDE_TRY(mix1d(spec_in, w, mixed, ax.rows, C, K, ax.keff, ax.kp, false, st));
    syn_in = mixed;
  } else if (mode == RPDE_MODE_FULL) {
    float* wblk = ar.take((size_t)ax.keff * 4 * C * C);
    float* mixed = ar.take(spec_floats(ax, C));
    if (!ar.ok()) { set_error("fspectral: workspace too small"); return RPDE_ERR_WORKSPACE; }
    RPDE_TRY(pack_mix_weights(w, wblk, C, C, K, ax.keff, st));
    if (ax.kp != ax.keff) RPDE_HIP(hipMemsetAsync(mixed, 0, spec_floats(ax, C) * sizeof(float), st));
    RPDE_TRY(mode_mix(ax, spec_in, wblk, mixed, C, false, st));
    syn_in = mixed;
  }
  return dft_synthesis(ax, syn_in, out, C, false, accumulate, st);
}

// backward of one axis: g -> gx (+)=, gw
static int axis_bwd(const Axis& ax, const float* g, const float* spec_in, const float* w, int K, float* gx, float* gw, int C,
                    int mode, int accumulate, Arena& ar, hipStream_t st, const float* skip = nullptr) {
  float* gspec = ar.take(spec_floats(ax, C));
  if (!ar.ok()) { set_error("fspectral: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(dft_analysis(ax, g, gspec, C, true, st));
  const float* dspec = gspec;
  if (mode == RPDE_MODE_FULL && mix1d_ok(ax.rows, C)) {
    float* dsp = ar.take(spec_floats(ax, C));
    if (!ar.ok()) { set_error("fspectral: workspace too small"); return RPDE_ERR_WORKSPACE; }
    if (gw) RPDE_TRY(mix1d_wgrad(spec_in, gspec, gw, ax.rows, C, K, ax.keff, ax.kp, st));
    if (gx) {
      RPDE_TRY(mix1d(gspec, w, dsp, ax.rows, C, K, ax.keff, ax.kp, true, st));
      dspec = dsp;
    }
  } else if (mode == RPDE_MODE_FULL) {
    float* wblk = ar.take((size_t)ax.keff * 4 * C * C);
    float* dsp = ar.take(spec_floats(ax, C));
    const int S = split_for(ax.rows, wgrad_tiles(C) * ax.keff);
    float* slabs = ar.take((size_t)S * ax.keff * 4 * C * C);
    if (!ar.ok()) { set_error("fspectral: workspace too small"); return RPDE_ERR_WORKSPACE; }
    if (gw) {
      RPDE_TRY(mode_mix_wgrad(ax, spec_in, gspec, slabs, C, S, st));
      RPDE_TRY(unpack_mix_grad(slabs, gw, C, C, K, ax.keff, S, (long)ax.keff * 4 * C * C, st));
    }
    if (gx) {
      RPDE_TRY(pack_mix_weights(w, wblk, C, C, K, ax.keff, st));
      if (ax.kp != ax.keff) RPDE_HIP(hipMemsetAsync(dsp, 0, spec_floats(ax, C) * sizeof(float), st));
      RPDE_TRY(mode_mix(ax, gspec, wblk, dsp, C, true, st));
      dspec = dsp;
    }
  } else if (gw) {
    RPDE_HIP(hipMemsetAsync(gw, 0, sizeof(float) * 2 * (size_t)C * C * K, st));
  }
  if (gx) RPDE_TRY(dft_synthesis(ax, dspec, gx, C, true, (!accumulate && skip) ? 2 : accumulate, st, skip));
  return RPDE_OK;
}

// ---- fused 2-D path (fused_spectral.hip): one analysis launch for both axes, mode mix on the small spectra,
// ---- one synthesis launch that writes the field once
// RPDE_FUSED_MIX=0: mode mix by pack + GEMM + split per axis (the round-2 sequence; A/B and tests)
static bool fused_mix_on() {
  const char* e = getenv("RPDE_FUSED_MIX");
  return !(e && e[0] == '0');
}

// slabs of lines for the h2 weight gradient of the mix: enough workgroups to fill the chip (keff x S x 2), at least four
// 32-line tiles each (more slabs do not help: S = 20 / 24 / 32 / 48 gave 108 / 105 / 110 / 109 us against 102 at 12,
// and the fold grows from 7.6 to 29 us)
static int mixw_slabs(const Axis& ay, const Axis& ax) {
  const long tiles = (ay.rows < ax.rows ? ay.rows : ax.rows) / 32;
  long S = tiles / 4;
  if (S > 12) S = 12;
  if (S < 1) S = 1;
  return (int)S;
}

static size_t fused2d_ws(const Axis& ay, const Axis& ax, int C, bool backward) {
  const size_t spy = arena_bytes(spec_floats(ay, C)), spx = arena_bytes(spec_floats(ax, C));
  const size_t wblk = arena_bytes((size_t)ay.keff * 4 * C * C);
  const size_t img = arena_bytes(fused2d_img_bytes(ay.rows, 2 * ay.kp) / 4) + arena_bytes(fused2d_img_bytes(ax.rows, 2 * ax.kp) / 4);
  const size_t inv = arena_bytes(ay.rows) + arena_bytes(ax.rows);
  // h2 mode mix: weight fragments + per-mode constants + max |spectrum| per line
  const size_t mix = arena_bytes(mix_wimg_bytes(ay.kp) / 4) + arena_bytes(4 * (size_t)ay.kp) + arena_bytes(ay.rows) + arena_bytes(ax.rows);
  size_t n = wblk + (spy > spx ? spy : spx) + img + inv + mix;
  if (backward) {
    const int Sy = split_for(ay.rows, wgrad_tiles(C) * ay.keff), Sx = split_for(ax.rows, wgrad_tiles(C) * ax.keff);
    const size_t sl = arena_bytes((size_t)(Sy > Sx ? Sy : Sx) * ay.keff * 4 * C * C);
    n += spy + spx + sl + arena_bytes(mix_wgrad_slab_floats(ay.kp, mixw_slabs(ay, ax)));
  }
  return n;
}

// prep (optional): weight fragments + constants written by rpde_fspectral2d_prepare for these weights
static int fused2d_fwd(const Axis& ay, const Axis& ax, const float* x, const float* w_y, const float* w_x, float* out,
                       float* spec_y, float* spec_x, int B, int M, int N, int C, int K, int mode, Arena& ar, hipStream_t st,
                       const void* prep = nullptr) {
  float* wblk = ar.take((size_t)ay.keff * 4 * C * C);
  const size_t sy = spec_floats(ay, C), sx = spec_floats(ax, C);
  float* mixed = ar.take(sy > sx ? sy : sx);
  void* imgy = ar.take(fused2d_img_bytes(ay.rows, 2 * ay.kp) / 4);
  void* imgx = ar.take(fused2d_img_bytes(ax.rows, 2 * ax.kp) / 4);
  float* invy = ar.take(ay.rows);
  float* invx = ar.take(ax.rows);
  void* wimg = ar.take(mix_wimg_bytes(ay.kp) / 4);
  float* wc = ar.take(4 * (size_t)ay.kp);
  // max |spectrum| per line: kept behind the saved spectra (rpde_fspectral2d_spec_elems), the backward needs them too
  float* amy = spec_y + spec_floats(ay, C);
  float* amx = spec_x + spec_floats(ax, C);
  if (!ar.ok()) { set_error("fspectral2d: workspace too small"); return RPDE_ERR_WORKSPACE; }
  if (mode == RPDE_MODE_FULL && fused_mix_on()) {
    // prep (weights -> fragments) | analysis (+ max per line) | mix (spectra -> operand blocks) | synthesis
    if (prep) {
      wimg = const_cast<void*>(prep);
      wc = reinterpret_cast<float*>(static_cast<char*>(wimg) + arena_bytes(mix_wimg_bytes(ay.kp) / 4));
    } else {
      RPDE_TRY(mix_prep(w_y, w_x, K, ay.keff, ay.kp, 0, wimg, wc, st));
    }
    RPDE_TRY(fused2d_analysis(x, spec_y, spec_x, amy, amx, ay.plan, ax.plan, 0, B, M, N, st));
    RPDE_TRY(mix_h2(spec_y, spec_x, amy, amx, imgy, imgx, invy, invx, ay.rows, ax.rows, ay.kp, wimg, wc, st));
    return fused2d_synthesis(imgy, imgx, invy, invx, ay.plan, ax.plan, 0, out, nullptr, B, M, N, st);
  }
  RPDE_TRY(fused2d_analysis(x, spec_y, spec_x, nullptr, nullptr, ay.plan, ax.plan, 0, B, M, N, st));
  const Axis* axes[2] = {&ay, &ax};
  const float* ws_[2] = {w_y, w_x};
  float* specs[2] = {spec_y, spec_x};
  void* imgs[2] = {imgy, imgx};
  float* invs[2] = {invy, invx};
  for (int a = 0; a < 2; ++a) {
    const Axis& A = *axes[a];
    const float* syn_in = specs[a];
    if (mode == RPDE_MODE_FULL) {
      RPDE_TRY(pack_mix_weights(ws_[a], wblk, C, C, K, A.keff, st));
      if (A.kp != A.keff) RPDE_HIP(hipMemsetAsync(mixed, 0, spec_floats(A, C) * sizeof(float), st));
      RPDE_TRY(mode_mix(A, specs[a], wblk, mixed, C, false, st));
      syn_in = mixed;
    }
    RPDE_TRY(fused2d_split(syn_in, imgs[a], invs[a], A.rows, 2 * A.kp, st));
  }
  return fused2d_synthesis(imgy, imgx, invy, invx, ay.plan, ax.plan, 0, out, nullptr, B, M, N, st);
}

static int fused2d_bwd(const Axis& ay, const Axis& ax, const float* g, const float* spec_y, const float* spec_x,
                       const float* w_y, const float* w_x, float* gx, float* gwy, float* gwx, const float* skip, int B, int M,
                       int N, int C, int K, int mode, Arena& ar, hipStream_t st) {
  float* gsy = ar.take(spec_floats(ay, C));
  float* gsx = ar.take(spec_floats(ax, C));
  float* wblk = ar.take((size_t)ay.keff * 4 * C * C);
  const size_t sy = spec_floats(ay, C), sx = spec_floats(ax, C);
  float* dsp = ar.take(sy > sx ? sy : sx);
  void* imgy = ar.take(fused2d_img_bytes(ay.rows, 2 * ay.kp) / 4);
  void* imgx = ar.take(fused2d_img_bytes(ax.rows, 2 * ax.kp) / 4);
  float* invy = ar.take(ay.rows);
  float* invx = ar.take(ax.rows);
  const int Sy = split_for(ay.rows, wgrad_tiles(C) * ay.keff), Sx = split_for(ax.rows, wgrad_tiles(C) * ax.keff);
  float* slabs = ar.take((size_t)(Sy > Sx ? Sy : Sx) * ay.keff * 4 * C * C);
  void* wimg = ar.take(mix_wimg_bytes(ay.kp) / 4);
  float* wc = ar.take(4 * (size_t)ay.kp);
  float* amy = ar.take(ay.rows);
  float* amx = ar.take(ax.rows);
  const int Sw = mixw_slabs(ay, ax);
  float* wslabs = ar.take(mix_wgrad_slab_floats(ay.kp, Sw));
  if (!ar.ok()) { set_error("fspectral2d: workspace too small"); return RPDE_ERR_WORKSPACE; }
  const bool h2mix = mode == RPDE_MODE_FULL && fused_mix_on();
  const bool hmix = h2mix && gx, hwg = h2mix && (gwy || gwx);
  if (hmix) RPDE_TRY(mix_prep(w_y, w_x, K, ay.keff, ay.kp, 1, wimg, wc, st));
  RPDE_TRY(fused2d_analysis(g, gsy, gsx, h2mix ? amy : nullptr, h2mix ? amx : nullptr, ay.plan, ax.plan, 1, B, M, N, st));
  if (hmix) {
    // d-spectra = g-spectra . W^H, straight into the operand blocks of the adjoint synthesis
    RPDE_TRY(mix_h2(gsy, gsx, amy, amx, imgy, imgx, invy, invx, ay.rows, ax.rows, ay.kp, wimg, wc, st));
  }
  if (hwg) {
    // weight gradients of both axes from the saved spectra (their line maxima sit behind them) and the g-spectra
    RPDE_TRY(mix_wgrad_h2(spec_y, spec_x, gsy, gsx, spec_y + spec_floats(ay, C), spec_x + spec_floats(ax, C), amy, amx, gwy, gwx,
                          ay.rows, ax.rows, K, ay.keff, ay.kp, wslabs, Sw, st));
  }
  const Axis* axes[2] = {&ay, &ax};
  const float* ws_[2] = {w_y, w_x};
  const float* specs[2] = {spec_y, spec_x};
  float* gspecs[2] = {gsy, gsx};
  float* gws[2] = {gwy, gwx};
  void* imgs[2] = {imgy, imgx};
  float* invs[2] = {invy, invx};
  const int Ss[2] = {Sy, Sx};
  for (int a = 0; a < 2; ++a) {
    const Axis& A = *axes[a];
    const float* dspec = gspecs[a];
    if (mode == RPDE_MODE_FULL) {
      if (gws[a] && !hwg) {
        RPDE_TRY(mode_mix_wgrad(A, specs[a], gspecs[a], slabs, C, Ss[a], st));
        RPDE_TRY(unpack_mix_grad(slabs, gws[a], C, C, K, A.keff, Ss[a], (long)A.keff * 4 * C * C, st));
      }
      if (gx && !hmix) {
        RPDE_TRY(pack_mix_weights(ws_[a], wblk, C, C, K, A.keff, st));
        if (A.kp != A.keff) RPDE_HIP(hipMemsetAsync(dsp, 0, spec_floats(A, C) * sizeof(float), st));
        RPDE_TRY(mode_mix(A, gspecs[a], wblk, dsp, C, true, st));
        dspec = dsp;
      }
    } else if (gws[a]) {
      RPDE_HIP(hipMemsetAsync(gws[a], 0, sizeof(float) * 2 * (size_t)C * C * K, st));
    }
    if (gx && !hmix) RPDE_TRY(fused2d_split(dspec, imgs[a], invs[a], A.rows, 2 * A.kp, st));
  }
  if (gx) RPDE_TRY(fused2d_synthesis(imgy, imgx, invy, invx, ay.plan, ax.plan, 1, gx, skip, B, M, N, st));
  return RPDE_OK;
}

}  // namespace rpde

using namespace rpde;

extern "C" {

// ------------------------------- 1-D ---------------------------------------
size_t rpde_fspectral1d_ws_bytes(int B, int n, int C, int K) {
  Axis ax; axis_dims(ax, n, K, B);
  return axis_ws_bytes(ax, C);
}
size_t rpde_fspectral1d_spec_elems(int B, int n, int C, int K) {
  Axis ax; axis_dims(ax, n, K, B);
  return spec_floats(ax, C);
}

int rpde_fspectral1d_fwd(const float* x, const float* w, float* out, float* spec_in, int B, int n, int C, int K, int mode,
                         int norm, void* ws, size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(x && out && spec_in && B > 0 && n > 0 && C > 0 && K > 0, "fspectral1d_fwd: bad arguments");
  RPDE_CHECK_ARG(mode == RPDE_MODE_LOWPASS || w, "fspectral1d_fwd: null weight");
  RPDE_CHECK_ARG(mode == RPDE_MODE_FULL || mode == RPDE_MODE_LOWPASS, "Mode %d not recognized", mode);
  hipStream_t st = as_stream(stream);
  Axis ax;
  RPDE_TRY(make_axis(ax, n, K, norm, B, 1, (long)n * C, 0, C, st));
  Arena ar(ws, ws_bytes);
  return axis_fwd(ax, x, w, K, out, spec_in, C, mode, 0, ar, st);
}

int rpde_fspectral1d_bwd(const float* grad_out, const float* spec_in, const float* w, float* grad_x, float* grad_w,
                         const float* grad_skip, int B, int n, int C, int K, int mode, int norm, void* ws, size_t ws_bytes,
                         void* stream) {
  RPDE_CHECK_ARG(grad_out && spec_in && B > 0 && n > 0 && C > 0 && K > 0, "fspectral1d_bwd: bad arguments");
  RPDE_CHECK_ARG(mode == RPDE_MODE_LOWPASS || w, "fspectral1d_bwd: null weight");
  hipStream_t st = as_stream(stream);
  Axis ax;
  RPDE_TRY(make_axis(ax, n, K, norm, B, 1, (long)n * C, 0, C, st));
  Arena ar(ws, ws_bytes);
  return axis_bwd(ax, grad_out, spec_in, w, K, grad_x, grad_w, C, mode, 0, ar, st, grad_skip);
}

// ------------------------------- 2-D ---------------------------------------
size_t rpde_fspectral2d_ws_bytes(int B, int M, int N, int C, int K) {
  Axis ay, ax;
  axis_dims(ay, N, K, B * M);
  axis_dims(ax, M, K, B * N);
  const size_t a = axis_ws_bytes(ay, C), b = axis_ws_bytes(ax, C);
  size_t n = a > b ? a : b;   // the two axes run back to back and reuse the arena
  if (fused2d_ok(M, N, C, ay.keff, ax.keff)) { const size_t f = fused2d_ws(ay, ax, C, true); if (f > n) n = f; }
  return n;
}
size_t rpde_fspectral2d_spec_elems(int B, int M, int N, int C, int K, int axis) {
  Axis a, o;
  if (axis == 0) { axis_dims(a, N, K, B * M); axis_dims(o, M, K, B * N); } else { axis_dims(a, M, K, B * N); axis_dims(o, N, K, B * M); }
  // fused path: the line maxima of the spectrum ride behind it (the backward's weight-gradient kernel scales by them)
  const bool fused = axis == 0 ? fused2d_ok(M, N, C, a.keff, o.keff) : fused2d_ok(M, N, C, o.keff, a.keff);
  return spec_floats(a, C) + (fused ? (size_t)a.rows : 0);
}

int rpde_fspectral2d_fwd(const float* x, const float* w_y, const float* w_x, float* out, float* spec_y, float* spec_x, int B,
                         int M, int N, int C, int K, int mode, void* ws, size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(x && out && spec_y && spec_x && B > 0 && M > 0 && N > 0 && C > 0 && K > 0, "fspectral2d_fwd: bad arguments");
  RPDE_CHECK_ARG(mode == RPDE_MODE_LOWPASS || (w_y && w_x), "fspectral2d_fwd: null weight");
  RPDE_CHECK_ARG(mode == RPDE_MODE_FULL || mode == RPDE_MODE_LOWPASS, "fspectral2d_fwd: mode %d", mode);
  hipStream_t st = as_stream(stream);
  Axis ay, ax;
  // lines along y (last spatial dim): one per (b,m), points C apart
  RPDE_TRY(make_axis(ay, N, K, RPDE_NORM_ORTHO, B * M, 1, (long)N * C, 0, C, st));
  // lines along x: one per (b,n), points N*C apart
  RPDE_TRY(make_axis(ax, M, K, RPDE_NORM_ORTHO, B * N, N, (long)M * N * C, C, (long)N * C, st));
  if (fused2d_ok(M, N, C, ay.keff, ax.keff)) {
    Arena ar(ws, ws_bytes);
    return fused2d_fwd(ay, ax, x, w_y, w_x, out, spec_y, spec_x, B, M, N, C, K, mode, ar, st);
  }
  {
    Arena ar(ws, ws_bytes);
    RPDE_TRY(axis_fwd(ax, x, w_x, K, out, spec_x, C, mode, 0, ar, st));
  }
  Arena ar(ws, ws_bytes);
  return axis_fwd(ay, x, w_y, K, out, spec_y, C, mode, 1, ar, st);
}

// ---- evaluation with frozen weights: rpde_fspectral2d_prepare builds the mode-mix weight fragments once,
// rpde_fspectral2d_fwd_prepared runs analysis | mix | synthesis with them; spectra live in the workspace ----
size_t rpde_fspectral2d_prep_bytes(int M, int N, int C, int K) {
  Axis ay, ax;
  axis_dims(ay, N, K, M);
  axis_dims(ax, M, K, N);
  if (!fused2d_ok(M, N, C, ay.keff, ax.keff) || !fused_mix_on()) return 0;      // 0: this shape has nothing to prepare
  return arena_bytes(mix_wimg_bytes(ay.kp) / 4) + arena_bytes(4 * (size_t)ay.kp);
}

size_t rpde_fspectral2d_eval_ws_bytes(int B, int M, int N, int C, int K) {
  return rpde_fspectral2d_ws_bytes(B, M, N, C, K) + arena_bytes(rpde_fspectral2d_spec_elems(B, M, N, C, K, 0)) +
         arena_bytes(rpde_fspectral2d_spec_elems(B, M, N, C, K, 1));
}

int rpde_fspectral2d_prepare(const float* w_y, const float* w_x, int M, int N, int C, int K, void* prep, size_t prep_bytes,
                             void* stream) {
  RPDE_CHECK_ARG(w_y && w_x && prep && M > 0 && N > 0 && K > 0, "fspectral2d_prepare: bad arguments");
  const size_t need = rpde_fspectral2d_prep_bytes(M, N, C, K);
  RPDE_CHECK_ARG(need > 0 && prep_bytes >= need, "fspectral2d_prepare: shape not on the fused path or buffer too small");
  Axis ay;
  axis_dims(ay, N, K, M);
  float* wc = reinterpret_cast<float*>(static_cast<char*>(prep) + arena_bytes(mix_wimg_bytes(ay.kp) / 4));
  return mix_prep(w_y, w_x, K, ay.keff, ay.kp, 0, prep, wc, as_stream(stream));
}

int rpde_fspectral2d_fwd_prepared(const float* x, const void* prep, float* out, int B, int M, int N, int C, int K, void* ws,
                                  size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(x && prep && out && ws && B > 0 && M > 0 && N > 0 && C > 0 && K > 0, "fspectral2d_fwd_prepared: bad arguments");
  RPDE_CHECK_ARG(rpde_fspectral2d_prep_bytes(M, N, C, K) > 0, "fspectral2d_fwd_prepared: shape not on the fused path");
  hipStream_t st = as_stream(stream);
  Axis ay, ax;
  RPDE_TRY(make_axis(ay, N, K, RPDE_NORM_ORTHO, B * M, 1, (long)N * C, 0, C, st));
  RPDE_TRY(make_axis(ax, M, K, RPDE_NORM_ORTHO, B * N, N, (long)M * N * C, C, (long)N * C, st));
  Arena ar(ws, ws_bytes);
  float* spec_y = ar.take(rpde_fspectral2d_spec_elems(B, M, N, C, K, 0));
  float* spec_x = ar.take(rpde_fspectral2d_spec_elems(B, M, N, C, K, 1));
  if (!ar.ok()) { set_error("fspectral2d_fwd_prepared: workspace too small"); return RPDE_ERR_WORKSPACE; }
  Arena rest(static_cast<char*>(ws) + ar.used, ws_bytes - ar.used);
  return fused2d_fwd(ay, ax, x, nullptr, nullptr, out, spec_y, spec_x, B, M, N, C, K, RPDE_MODE_FULL, rest, st, prep);
}

int rpde_fspectral2d_bwd(const float* grad_out, const float* spec_y, const float* spec_x, const float* w_y, const float* w_x,
                         float* grad_x, float* grad_wy, float* grad_wx, const float* grad_skip, int B, int M, int N, int C,
                         int K, int mode, void* ws, size_t ws_bytes, void* stream) {
  RPDE_CHECK_ARG(grad_out && spec_y && spec_x && B > 0 && M > 0 && N > 0 && C > 0 && K > 0, "fspectral2d_bwd: bad arguments");
  RPDE_CHECK_ARG(mode == RPDE_MODE_LOWPASS || (w_y && w_x), "fspectral2d_bwd: null weight");
  hipStream_t st = as_stream(stream);
  Axis ay, ax;
  RPDE_TRY(make_axis(ay, N, K, RPDE_NORM_ORTHO, B * M, 1, (long)N * C, 0, C, st));
  RPDE_TRY(make_axis(ax, M, K, RPDE_NORM_ORTHO, B * N, N, (long)M * N * C, C, (long)N * C, st));
  if (fused2d_ok(M, N, C, ay.keff, ax.keff)) {
    Arena ar(ws, ws_bytes);
    return fused2d_bwd(ay, ax, grad_out, spec_y, spec_x, w_y, w_x, grad_x, grad_wy, grad_wx, grad_skip, B, M, N, C, K, mode,
                       ar, st);
  }
  {
    Arena ar(ws, ws_bytes);
    RPDE_TRY(axis_bwd(ax, grad_out, spec_x, w_x, K, grad_x, grad_wx, C, mode, 0, ar, st, grad_skip));
  }
  Arena ar(ws, ws_bytes);
  return axis_bwd(ay, grad_out, spec_y, w_y, K, grad_x, grad_wy, C, mode, 1, ar, st);
}

}  // extern "C"
