// Error reporting and DFT plans (truncated real / row-restricted complex DFT
// tables built on device in double precision, cached per device).
#include "rpde_internal.h"
#include "plan.h"
#include "fused_spectral.h"
#include "cf_dft.h"

#include <stdarg.h>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace rpde {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

// ---------------------------------------------------------------------------
// table kernels.  Angles use the exact integer (k*y) mod n and sincospi in
// double, so the float tables are correctly rounded cos/sin values.
// ---------------------------------------------------------------------------
__device__ inline void unit(int k, int y, int n, double& c, double& s) {
  const long r = ((long)k * (long)y) % n;
  sincospi(2.0 * (double)r / (double)n, &s, &c);
}

// analysis [2*kp, ldn]: row(k,ri) x col y.  planar=0: row = 2k+ri; planar=1: row = ri*kp+k
__global__ void k_real_analysis(float* fa, int n, int modes, int kp, int ldn, double sf, int planar) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= kp * ldn) return;
  const int k = idx / ldn, y = idx % ldn;
  float re = 0.f, im = 0.f;
  if (k < modes && y < n) {
    double c, s;
    unit(k, y, n, c, s);
    re = (float)(sf * c);
    im = (float)(-sf * s);
  }
  const int rre = planar ? k : 2 * k, rim = planar ? kp + k : 2 * k + 1;
  fa[(long)rre * ldn + y] = re;
  fa[(long)rim * ldn + y] = im;
}

// synthesis [n, 2*kp]: row y x col(k,ri), Hermitian weights c_k, Im(DC/Nyquist) vanish by sin = 0
__global__ void k_real_synthesis(float* fs, int n, int modes, int kp, double si, int planar) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * kp) return;
  const int y = idx / kp, k = idx % kp;
  float re = 0.f, im = 0.f;
  if (k < modes) {
    double c, s;
    unit(k, y, n, c, s);
    const double w = (k == 0 || (n % 2 == 0 && k == n / 2)) ? 1.0 : 2.0;
    re = (float)(si * w * c);
    im = (float)(-si * w * s);
  }
  const int cre = planar ? k : 2 * k, cim = planar ? kp + k : 2 * k + 1;
  fs[(long)y * 2 * kp + cre] = re;
  fs[(long)y * 2 * kp + cim] = im;
}

__global__ void k_transpose_small(const float* __restrict__ in, float* __restrict__ out, int rows, int cols) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const int r = idx / cols, c = idx % cols;
  out[(long)c * rows + r] = in[idx];
}

// complex forward DFT along an axis of length m restricted to R = top + bot row
// slots (slot r < top -> bin r, else bin m - bot + (r - top)) as a real [2R, 2m] block
// matrix; rows 2r+ri2, cols 2*mm+ri1.
__global__ void k_cplx_analysis(float* t, int m, int top, int bot, double sf) {
  const int R = top + bot;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * m) return;
  const int r = idx / m, mm = idx % m;
  const int bin = r < top ? r : m - bot + (r - top);
  double c, s;
  unit(bin, mm, m, c, s);
  const long ld = 2L * m;
  t[(2L * r) * ld + 2 * mm] = (float)(sf * c);
  t[(2L * r) * ld + 2 * mm + 1] = (float)(sf * s);
  t[(2L * r + 1) * ld + 2 * mm] = (float)(-sf * s);
  t[(2L * r + 1) * ld + 2 * mm + 1] = (float)(sf * c);
}

// inverse: [2m, 2R]; slots r < m1 whose bin is also covered by the upper block
// (2*m1 > m: the reference's second slice-assign overwrites them, quirk Q6)
// get zero columns.
__global__ void k_cplx_synthesis(float* t, int m, int top, int bot, double si) {
  const int R = top + bot;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= R * m) return;
  const int mm = idx / R, r = idx % R;
  const int bin = r < top ? r : m - bot + (r - top);
  double c, s;
  unit(bin, mm, m, c, s);
  const bool dead = (r < top) && (bin >= m - bot);
  const double a = dead ? 0.0 : si;
  const long ld = 2L * R;
  t[(2L * mm) * ld + 2 * r] = (float)(a * c);
  t[(2L * mm) * ld + 2 * r + 1] = (float)(-a * s);
  t[(2L * mm + 1) * ld + 2 * r] = (float)(a * s);
  t[(2L * mm + 1) * ld + 2 * r + 1] = (float)(a * c);
}

static void norm_scales(int n, int norm, double& sf, double& si) {
  if (norm == RPDE_NORM_ORTHO) { sf = si = 1.0 / sqrt((double)n); }
  else if (norm == RPDE_NORM_FORWARD) { sf = 1.0 / n; si = 1.0; }
  else { sf = 1.0; si = 1.0 / n; }
}

static int build_plan(rpde_plan** out, int n, int modes, int norm, int planar, int kind, hipStream_t st, int bot = -1) {
  RPDE_CHECK_ARG(n >= 1 && modes >= (kind == PLAN_REAL ? 1 : 0), "plan: bad n=%d modes=%d", n, modes);
  RPDE_CHECK_ARG(norm >= 0 && norm <= 2, "plan: bad norm %d", norm);
  rpde_plan* p = new rpde_plan();
  memset(p, 0, sizeof(*p));
  RPDE_HIP(hipGetDevice(&p->device));
  p->n = n; p->modes = modes; p->norm = norm; p->planar = planar; p->kind = kind;
  double sf, si;
  norm_scales(n, norm, sf, si);
  if (kind == PLAN_REAL) {
    RPDE_CHECK_ARG(modes <= n / 2 + 1, "plan: modes %d exceed n/2+1 = %d", modes, n / 2 + 1);
    p->kp = (modes + 3) / 4 * 4;
    p->ldn = (n + 3) / 4 * 4;
    RPDE_HIP(hipMalloc(&p->fa, sizeof(float) * 2 * p->kp * p->ldn));
    RPDE_HIP(hipMalloc(&p->fs, sizeof(float) * (size_t)n * 2 * p->kp));
    const int t1 = p->kp * p->ldn, t2 = n * p->kp;
    hipLaunchKernelGGL(k_real_analysis, dim3((t1 + 255) / 256), dim3(256), 0, st, p->fa, n, modes, p->kp, p->ldn, sf, planar);
    hipLaunchKernelGGL(k_real_synthesis, dim3((t2 + 255) / 256), dim3(256), 0, st, p->fs, n, modes, p->kp, si, planar);
    if (!planar && p->ldn == n) {
      const int m2 = 2 * p->kp;
      RPDE_HIP(hipMalloc(&p->img[IMG_FA], split_bytes(m2, n)));
      RPDE_HIP(hipMalloc(&p->img[IMG_FST], split_bytes(m2, n)));
      RPDE_HIP(hipMalloc(&p->img[IMG_FS], split_bytes(n, m2)));
      RPDE_HIP(hipMalloc(&p->img[IMG_FAT], split_bytes(n, m2)));
      RPDE_TRY(split_weights(p->fa, 1, p->ldn, m2, n, p->img[IMG_FA], st));
      RPDE_TRY(split_weights(p->fs, 0, m2, m2, n, p->img[IMG_FST], st));
      RPDE_TRY(split_weights(p->fs, 1, m2, n, m2, p->img[IMG_FS], st));
      RPDE_TRY(split_weights(p->fa, 0, p->ldn, n, m2, p->img[IMG_FAT], st));
      if (n % 32 == 0 && n <= 256 && m2 <= 48) RPDE_TRY(h2_build_tables(p, st));
    }
    if (planar && p->ldn == n && cf_h2_eligible(n, 2 * p->kp)) RPDE_TRY(cf_build_tables(p, st));
    if (planar) {
      RPDE_HIP(hipMalloc(&p->fs_t, sizeof(float) * (size_t)n * 2 * p->kp));
      const int tt = n * 2 * p->kp;
      hipLaunchKernelGGL(k_transpose_small, dim3((tt + 255) / 256), dim3(256), 0, st, p->fs, p->fs_t, n, 2 * p->kp);
    }
  } else {
    if (bot < 0) bot = modes;
    RPDE_CHECK_ARG(modes <= n && bot <= n && modes >= 0 && bot >= 0 && modes + bot >= 1, "plan: rows (%d,%d) vs M %d", modes, bot, n);
    const int R = modes + bot;
    p->kp = R; p->ldn = 2 * n;
    RPDE_HIP(hipMalloc(&p->fa, sizeof(float) * 2 * R * 2 * n));
    RPDE_HIP(hipMalloc(&p->fs, sizeof(float) * 2 * n * 2 * R));
    const int t = R * n;
    hipLaunchKernelGGL(k_cplx_analysis, dim3((t + 255) / 256), dim3(256), 0, st, p->fa, n, modes, bot, sf);
    hipLaunchKernelGGL(k_cplx_synthesis, dim3((t + 255) / 256), dim3(256), 0, st, p->fs, n, modes, bot, si);
    // fs transposed, [2R, 2n]: read along the output index by k_col_mix_synthesis (spectral_cf.hip)
    RPDE_HIP(hipMalloc(&p->fs_t, sizeof(float) * 2 * n * 2 * R));
    const int tt = 2 * n * 2 * R;
    hipLaunchKernelGGL(k_transpose_small, dim3((tt + 255) / 256), dim3(256), 0, st, p->fs, p->fs_t, 2 * n, 2 * R);
  }
  RPDE_LAUNCH_CHECK();
  *out = p;
  return RPDE_OK;
}

// ---- cache ------------------------------------------------------------------
static std::mutex g_mu;
static std::map<std::tuple<int, int, int, int, int, int, int>, rpde_plan*> g_cache;

int get_plan(const rpde_plan** out, int n, int modes, int norm, int planar, int kind, hipStream_t st, int bot) {
  int dev = 0;
  RPDE_HIP(hipGetDevice(&dev));
  const auto key = std::make_tuple(dev, n, modes, norm, planar, kind, bot);
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_cache.find(key);
  if (it != g_cache.end()) { *out = it->second; return RPDE_OK; }
  // tables are built on the calling stream: stream order makes them visible to
  // the first user; later users on other streams of this device are ordered by
  // this one-time synchronisation.
  rpde_plan* p = nullptr;
  RPDE_TRY(build_plan(&p, n, modes, norm, planar, kind, st, bot));
  RPDE_HIP(hipStreamSynchronize(st));
  g_cache[key] = p;
  *out = p;
  return RPDE_OK;
}

}  // namespace rpde

extern "C" {

const char* rpde_last_error(void) { return rpde::g_err.c_str(); }
int rpde_version(void) { return 100; }

int rpde_plan_cache_count(void) {
  std::lock_guard<std::mutex> lk(rpde::g_mu);
  return (int)rpde::g_cache.size();
}

int rpde_plan_create(rpde_plan** plan, int n, int modes, int norm, void* stream) {
  if (!plan) { rpde::set_error("plan_create: null out pointer"); return RPDE_ERR_ARG; }
  int keff = modes < n / 2 + 1 ? modes : n / 2 + 1;
  RPDE_TRY(rpde::build_plan(plan, n, keff, norm, 0, rpde::PLAN_REAL, rpde::as_stream(stream)));
  RPDE_HIP(hipStreamSynchronize(rpde::as_stream(stream)));
  return RPDE_OK;
}

int rpde_plan_destroy(rpde_plan* p) {
  if (!p) return RPDE_OK;
  if (p->fa) (void)hipFree(p->fa);
  if (p->fs) (void)hipFree(p->fs);
  if (p->fs_t) (void)hipFree(p->fs_t);
  for (int i = 0; i < 4; ++i) if (p->img[i]) (void)hipFree(p->img[i]);
  for (int i = 0; i < 2; ++i) {
    if (p->h2_ana[i]) (void)hipFree(p->h2_ana[i]);
    if (p->h2_ana_p[i]) (void)hipFree(p->h2_ana_p[i]);
    if (p->h2_syn[i]) (void)hipFree(p->h2_syn[i]);
    if (p->cf_ana[i]) (void)hipFree(p->cf_ana[i]);
    if (p->cf_syn[i]) (void)hipFree(p->cf_syn[i]);
  }
  delete p;
  return RPDE_OK;
}

int rpde_plan_info(const rpde_plan* p, int* n, int* modes, int* kp, int* ldn) {
  if (!p) { rpde::set_error("plan_info: null plan"); return RPDE_ERR_ARG; }
  if (n) *n = p->n;
  if (modes) *modes = p->modes;
  if (kp) *kp = p->kp;
  if (ldn) *ldn = p->ldn;
  return RPDE_OK;
}

int rpde_plan_tables(const rpde_plan* p, float* analysis_host, float* synthesis_host) {
  if (!p) { rpde::set_error("plan_tables: null plan"); return RPDE_ERR_ARG; }
  if (analysis_host)
    RPDE_HIP(hipMemcpy(analysis_host, p->fa, sizeof(float) * 2 * p->kp * p->ldn, hipMemcpyDeviceToHost));
  if (synthesis_host)
    RPDE_HIP(hipMemcpy(synthesis_host, p->fs, sizeof(float) * (size_t)p->n * 2 * p->kp, hipMemcpyDeviceToHost));
  return RPDE_OK;
}

}  // extern "C"
