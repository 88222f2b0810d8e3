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
#include "pointwise.h"
#include "cf_dft.h"

namespace rpde {

// complex position (r, ky) of a planar block [R][2][kp]: re at (2r)*kp+ky, im at (2r+1)*kp+ky
struct MixGeom { int B, Ci, Co, R, m1, m2, kp; };

__device__ __forceinline__ const float* wsel(const float* w1, const float* w2, const MixGeom& g, int i, int o, int r, int ky) {
  const float* w = r < g.m1 ? w1 : w2;
  const int rr = r < g.m1 ? r : r - g.m1;
  return w + ((((long)i * g.Co + o) * g.m1 + rr) * g.m2 + ky) * 2;
}

__global__ void k_cmix_fwd(const float* __restrict__ s, const float* __restrict__ w1, const float* __restrict__ w2,
                           float* __restrict__ out, MixGeom g) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tot = (long)g.B * g.Co * g.R * g.m2;
  if (idx >= tot) return;
  const int ky = (int)(idx % g.m2);
  const int r = (int)((idx / g.m2) % g.R);
  const int o = (int)((idx / ((long)g.m2 * g.R)) % g.Co);
  const int b = (int)(idx / ((long)g.m2 * g.R * g.Co));
  const long blk = 2L * g.R * g.kp;
  float ar = 0.f, ai = 0.f;
  for (int i = 0; i < g.Ci; ++i) {
    const float* sp = s + ((long)b * g.Ci + i) * blk + (2L * r) * g.kp + ky;
    const float xr = sp[0], xi = sp[g.kp];
    const float* w = wsel(w1, w2, g, i, o, r, ky);
    ar += xr * w[0] - xi * w[1];
    ai += xr * w[1] + xi * w[0];
  }
  float* op = out + ((long)b * g.Co + o) * blk + (2L * r) * g.kp + ky;
  op[0] = ar;
  op[g.kp] = ai;
}

// ds[b,i] = sum_o g[b,o] * conj(W[i,o])
__global__ void k_cmix_bwd_data(const float* __restrict__ gsp, const float* __restrict__ w1, const float* __restrict__ w2,
                                float* __restrict__ ds, MixGeom g) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tot = (long)g.B * g.Ci * g.R * g.m2;
  if (idx >= tot) return;
  const int ky = (int)(idx % g.m2);
  const int r = (int)((idx / g.m2) % g.R);
  const int i = (int)((idx / ((long)g.m2 * g.R)) % g.Ci);
  const int b = (int)(idx / ((long)g.m2 * g.R * g.Ci));
  const long blk = 2L * g.R * g.kp;
  float ar = 0.f, ai = 0.f;
  for (int o = 0; o < g.Co; ++o) {
    const float* gp = gsp + ((long)b * g.Co + o) * blk + (2L * r) * g.kp + ky;
    const float gr = gp[0], gi = gp[g.kp];
    const float* w = wsel(w1, w2, g, i, o, r, ky);
    ar += gr * w[0] + gi * w[1];
    ai += gi * w[0] - gr * w[1];
  }
  float* dp = ds + ((long)b * g.Ci + i) * blk + (2L * r) * g.kp + ky;
  dp[0] = ar;
  dp[g.kp] = ai;
}

// gW[i,o] = sum_b conj(s[b,i]) * g[b,o]
__global__ void k_cmix_bwd_w(const float* __restrict__ s, const float* __restrict__ gsp, float* __restrict__ gw1,
                             float* __restrict__ gw2, MixGeom g) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long tot = (long)g.Ci * g.Co * g.R * g.m2;
  if (idx >= tot) return;
  const int ky = (int)(idx % g.m2);
  const int r = (int)((idx / g.m2) % g.R);
  const int o = (int)((idx / ((long)g.m2 * g.R)) % g.Co);
  const int i = (int)(idx / ((long)g.m2 * g.R * g.Co));
  const long blk = 2L * g.R * g.kp;
  float ar = 0.f, ai = 0.f;
  for (int b = 0; b < g.B; ++b) {
    const float* sp = s + ((long)b * g.Ci + i) * blk + (2L * r) * g.kp + ky;
    const float* gp = gsp + ((long)b * g.Co + o) * blk + (2L * r) * g.kp + ky;
    const float xr = sp[0], xi = sp[g.kp], gr = gp[0], gi = gp[g.kp];
    ar += xr * gr + xi * gi;
    ai += xr * gi - xi * gr;
  }
  float* gw = r < g.m1 ? gw1 : gw2;
  const int rr = r < g.m1 ? r : r - g.m1;
  float* p = gw + ((((long)i * g.Co + o) * g.m1 + rr) * g.m2 + ky) * 2;
  p[0] = ar;
  p[1] = ai;
}

static int launch_mix(int which, const float* a, const float* b, const float* w1, const float* w2, float* out, float* out2,
                      const MixGeom& g, hipStream_t st) {
  long tot;
  if (which == 0) { tot = (long)g.B * g.Co * g.R * g.m2; hipLaunchKernelGGL(k_cmix_fwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, a, w1, w2, out, g); }
  else if (which == 1) { tot = (long)g.B * g.Ci * g.R * g.m2; hipLaunchKernelGGL(k_cmix_bwd_data, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, a, w1, w2, out, g); }
  else { tot = (long)g.Ci * g.Co * g.R * g.m2; hipLaunchKernelGGL(k_cmix_bwd_w, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, a, b, out, out2, g); }
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// rows x n  ->  rows x 2kp   (forward real DFT along the contiguous axis), optional act on the input
static int cf_analysis(const rpde_plan* pl, const float* x, float* spec, long rows, int n, int act_in, hipStream_t st) {
  if (!act_in && pl->cf_ana[0] && rows >= 16 && cf_h2_eligible(n, 2 * pl->kp)) return cf_analysis_h2(pl, 0, x, spec, rows, 1.f, st);
  rpde_gemm_desc d = gemm_desc();
  d.A = x; d.a_kmajor = 1; d.lda = n; d.act_a = act_in;
  d.B = pl->fa; d.b_kmajor = 1; d.ldb = pl->ldn;
  d.C = spec; d.ldc = 2L * pl->kp;
  d.M = (int)rows; d.N = 2 * pl->kp; d.K = n;
  return launch_gemm(d, st);
}
// rows x 2kp -> rows x n  (C2R synthesis)
static int cf_synthesis(const rpde_plan* pl, const float* spec, float* out, long rows, int n, hipStream_t st,
                        float alpha = 1.f) {
  if (pl->cf_syn[0] && rows >= 16 && cf_h2_eligible(n, 2 * pl->kp)) return cf_synthesis_h2(pl, 0, spec, out, rows, alpha, st);
  rpde_gemm_desc d = gemm_desc();
  d.alpha = alpha;
  d.A = spec; d.a_kmajor = 1; d.lda = 2L * pl->kp;
  d.B = pl->fs; d.b_kmajor = 1; d.ldb = 2L * pl->kp;
  d.C = out; d.ldc = n;
  d.M = (int)rows; d.N = n; d.K = 2 * pl->kp;
  return launch_gemm(d, st);
}
// adjoint of synthesis: g[rows,n] . Fs -> [rows, 2kp]
static int cf_synthesis_T(const rpde_plan* pl, const float* g, float* gspec, long rows, int n, hipStream_t st) {
  if (pl->cf_ana[1] && rows >= 16 && cf_h2_eligible(n, 2 * pl->kp)) return cf_analysis_h2(pl, 1, g, gspec, rows, 1.f, st);
  rpde_gemm_desc d = gemm_desc();
  d.A = g; d.a_kmajor = 1; d.lda = n;
  d.B = pl->fs; d.b_kmajor = 0; d.ldb = 2L * pl->kp;
  d.C = gspec; d.ldc = 2L * pl->kp;
  d.M = (int)rows; d.N = 2 * pl->kp; d.K = n;
  return launch_gemm(d, st);
}
// adjoint of analysis: dspec[rows,2kp] . Fa -> gx[rows,n], through act'(x) when act_in
static int cf_analysis_T(const rpde_plan* pl, const float* dspec, float* gx, long rows, int n, int act_in, const float* x,
                         hipStream_t st) {
  if (!act_in && pl->cf_syn[1] && rows >= 16 && cf_h2_eligible(n, 2 * pl->kp)) return cf_synthesis_h2(pl, 1, dspec, gx, rows, 1.f, st);
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

}  // namespace rpde

using namespace rpde;

extern "C" {

// --------------------------------- 1-D --------------------------------------
size_t rpde_spectral1d_ws_bytes(int B, int Cin, int Cout, int n, int K) {
  const int kp = (K + 3) / 4 * 4;
  const int cm = Cin > Cout ? Cin : Cout;
  (void)n;
  return 2 * arena_bytes((size_t)B * cm * 2 * kp);
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
  if (!ar.ok()) { set_error("spectral1d_fwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(cf_analysis(pl, x, spec_in, (long)B * Cin, n, act_in, st));
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
  if (!ar.ok()) { set_error("spectral1d_bwd: workspace too small"); return RPDE_ERR_WORKSPACE; }
  RPDE_TRY(cf_synthesis_T(pl, grad_out, gspec, (long)B * Cout, n, st));
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
  RPDE_TRY(cf_rowdft(pm->fa, 2L * M, false, 2 * R, 2 * M, s1, spec_in, B * Cin, kp, st));
  if (kp != m2) RPDE_HIP(hipMemsetAsync(o2, 0, sizeof(float) * (size_t)B * Cout * 2 * R * kp, st));
  MixGeom g{B, Cin, Cout, R, m1, m2, kp};
  RPDE_TRY(launch_mix(0, spec_in, nullptr, w1, w2, o2, nullptr, g, st));
  RPDE_TRY(cf_rowdft(pm->fs, 2L * R, false, 2 * M, 2 * R, o2, t1, B * Cout, kp, st));
  return cf_synthesis(pn, t1, out, (long)B * Cout * M, N, st);
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
