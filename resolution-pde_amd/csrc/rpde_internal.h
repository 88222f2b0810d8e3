// Internal helpers shared by the HIP translation units of librpde_hip.so.
// gfx950 (CDNA4, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <string>

#include "rpde.h"

namespace rpde {

void set_error(const char* fmt, ...);

#define RPDE_CHECK_ARG(cond, ...)                       \
  do {                                                  \
    if (!(cond)) {                                      \
      ::rpde::set_error(__VA_ARGS__);                   \
      return RPDE_ERR_ARG;                              \
    }                                                   \
  } while (0)

#define RPDE_HIP(call)                                                              \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      ::rpde::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),      \
                        __FILE__, __LINE__);                                        \
      return RPDE_ERR_HIP;                                                          \
    }                                                                               \
  } while (0)

#define RPDE_LAUNCH_CHECK() RPDE_HIP(hipGetLastError())

#define RPDE_TRY(call)               \
  do {                               \
    int s_ = (call);                 \
    if (s_ != RPDE_OK) return s_;    \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- device math -----------------------------------------------------------
// Phi(x) and the normal density phi(x) from ONE exponential and one reciprocal (Abramowitz & Stegun 26.2.17, the
// normal-distribution form of 7.1.26):  1 - Phi(|x|) = phi(x) (b1 s + ... + b5 s^5),  s = 1 / (1 + 0.2316419 |x|).
// About 14 VALU instructions for Phi, 3 more for GELU and GELU' together, instead of the device library's erff
// (~100).  Measured in fp32 against scipy over [-9, 9]: |Phi error| <= 3.0e-7, |gelu error| <= 4.2e-7,
// |gelu' error| <= 3.1e-7 (parity budget of the hot path: 1e-5).
// gk = exp(-x^2/2) / sqrt(2 pi) = the normal density (the 1/sqrt(2 pi) rides in the exponent: exp2(-x^2 c + log2 k)),
// and the polynomial coefficients carry sqrt(2 pi) / 2 so that half = (p s) gk = 0.5 erfc(|x| / sqrt 2)
__device__ __forceinline__ void phi_parts(float x, float& cdf, float& gk) {
  const float s = __builtin_amdgcn_rcpf(fmaf(0.2316418882f, fabsf(x), 1.0f));
  gk = __builtin_amdgcn_exp2f(fmaf(x * x, -0.72134752044f, -1.32574806473f));
  float p = fmaf(1.330274429f, s, -1.821255978f);
  p = fmaf(p, s, 1.781477937f);
  p = fmaf(p, s, -0.356563782f);
  p = fmaf(p, s, 0.319381530f);
  const float half = (p * s) * gk;
  cdf = x < 0.f ? half : 1.0f - half;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float norm_cdf_f(float x) {
  float c, g;
  phi_parts(x, c, g);
  return c;
}
__device__ __forceinline__ float gelu_f(float u) { return u * norm_cdf_f(u); }
__device__ __forceinline__ float dgelu_f(float u) {
  float c, gk;
  phi_parts(u, c, gk);
  return fmaf(u, gk, c);
}
// h = act(u), d = act'(u) in one evaluation
__device__ __forceinline__ void act_both(int act, float u, float& h, float& d) {
#ifdef RPDE_EXP_NOACT      // timing experiment (profiles/ff_bench.py): what the fused kernels would cost with a free activation
  h = u; d = 1.f; return;
#endif
  if (act == RPDE_ACT_GELU) {
    float c, gk;
    phi_parts(u, c, gk);
    h = u * c;
    d = fmaf(u, gk, c);
  } else if (act == RPDE_ACT_RELU) {
    h = u > 0.f ? u : 0.f;
    d = u > 0.f ? 1.f : 0.f;
  } else {
    h = u;
    d = 1.f;
  }
}
__device__ __forceinline__ float act_f(int act, float u) {
  if (act == RPDE_ACT_GELU) return gelu_f(u);
  if (act == RPDE_ACT_RELU) return u > 0.f ? u : 0.f;
  return u;
}
__device__ __forceinline__ float dact_f(int act, float u) {
  if (act == RPDE_ACT_GELU) return dgelu_f(u);
  if (act == RPDE_ACT_RELU) return u > 0.f ? 1.f : 0.f;
  return 1.f;
}

// ---- counter-based dropout ---------------------------------------------------
// element id = point * ld + feature, so the forward staging, the backward
// epilogue and the weight-gradient staging regenerate the same mask without
// storing it.  Groups of four consecutive ids share one base word; each 32-bit
// avalanche hash (two multiplies) yields two 16-bit uniforms.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
struct DropCfg {
  uint64_t seed;
  uint32_t thresh;   // drop when 16-bit uniform < thresh
  float scale;       // 1/(1-p)
  // optional device counter mixed into the seed by the KERNEL (drop_resolve): a captured hipGraph replays its launch
  // arguments, so a seed drawn on the host would freeze the masks -- the training step advances this counter on the
  // device instead (rpde.ops.drop_epoch, rpde/graph.py); forward and backward of one step see the same value
  const uint64_t* epoch;
  __host__ __device__ bool on() const { return thresh != 0; }
};
inline DropCfg make_drop(float p, uint64_t seed, const uint64_t* epoch = nullptr) {
  DropCfg d;
  d.seed = seed;
  d.epoch = epoch;
  if (p <= 0.f) { d.thresh = 0; d.scale = 1.f; d.epoch = nullptr; return d; }
  double t = (double)p * 65536.0;
  d.thresh = (uint32_t)(t + 0.5);
  if (d.thresh > 65535u) d.thresh = 65535u;
  if (d.thresh == 0) d.thresh = 1;
  d.scale = (float)(1.0 / (1.0 - (double)d.thresh / 65536.0));
  return d;
}
// once per kernel, before the first mask: the configuration with the device counter folded into the seed
__device__ __forceinline__ DropCfg drop_resolve(DropCfg d) {
  if (d.epoch) {
    d.seed ^= (*d.epoch) * 0x9E3779B97F4A7C15ull;
    d.epoch = nullptr;
  }
  return d;
}
__device__ __forceinline__ uint32_t drop_base(const DropCfg& d, uint64_t group) {
  const uint32_t lo = (uint32_t)group, hi = (uint32_t)(group >> 32);
  return (lo ^ (uint32_t)d.seed) + (hi * 0x9E3779B9u ^ (uint32_t)(d.seed >> 32));
}
__device__ __forceinline__ float drop_scale1(const DropCfg& d, uint64_t id) {
  const uint32_t base = drop_base(d, id >> 2);
  const uint32_t h = mix32((id & 2) ? (base ^ 0x68E31DA4u) : base);
  const uint32_t u = (id & 1) ? (h >> 16) : (h & 0xFFFFu);
  return u < d.thresh ? 0.f : d.scale;
}
// id must be a multiple of 4
__device__ __forceinline__ void drop_scale4(const DropCfg& d, uint64_t id, float s[4]) {
  const uint32_t base = drop_base(d, id >> 2);
  const uint32_t h0 = mix32(base), h1 = mix32(base ^ 0x68E31DA4u);
  s[0] = (h0 & 0xFFFFu) < d.thresh ? 0.f : d.scale;
  s[1] = (h0 >> 16) < d.thresh ? 0.f : d.scale;
  s[2] = (h1 & 0xFFFFu) < d.thresh ? 0.f : d.scale;
  s[3] = (h1 >> 16) < d.thresh ? 0.f : d.scale;
}

// ---- wave / block reductions -----------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

int launch_gemm(const rpde_gemm_desc& d, hipStream_t stream);
// pre-split operands of the split-bf16 GEMM (gemm_bf16x3.hip)
int split_npad(int N);
size_t split_bytes(int N, int K);
int split_weights(const float* w, int kmajor, long ld, int N, int K, void* out, hipStream_t st);
struct SplitJobs {
  static constexpr int MAX = 4;
  int n = 0;
  struct Job { const float* w; char* out; long ld; int kmajor, N, K, blk0; } j[MAX];
  void add(const float* w, int kmajor, long ld, int N, int K, void* out) {
    j[n].w = w; j[n].out = static_cast<char*>(out); j[n].ld = ld; j[n].kmajor = kmajor; j[n].N = N; j[n].K = K; j[n].blk0 = 0;
    ++n;
  }
};
int split_weights_multi(SplitJobs& J, hipStream_t st);

inline rpde_gemm_desc gemm_desc() {
  rpde_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.batch = 1; d.zdiv = 1; d.ksplit = 1; d.alpha = 1.f;
  d.a_kmajor = 1; d.b_kmajor = 1;
  return d;
}

}  // namespace rpde
