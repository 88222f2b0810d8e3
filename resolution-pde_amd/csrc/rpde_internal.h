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
__device__ __forceinline__ float gelu_f(float u) {
  return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f));
}
__device__ __forceinline__ float dgelu_f(float u) {
  const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * u * u);
  return cdf + u * pdf;
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
// One splitmix64 round per group of four consecutive element ids yields four
// 16-bit uniforms; element id = point * ld + feature, so the forward staging,
// the backward epilogue and the weight-gradient staging regenerate the same
// mask without storing it.
__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t g) {
  uint64_t z = seed + (g + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct DropCfg {
  uint64_t seed;
  uint32_t thresh;   // drop when 16-bit uniform < thresh
  float scale;       // 1/(1-p)
  __host__ __device__ bool on() const { return thresh != 0; }
};
inline DropCfg make_drop(float p, uint64_t seed) {
  DropCfg d;
  d.seed = seed;
  if (p <= 0.f) { d.thresh = 0; d.scale = 1.f; return d; }
  double t = (double)p * 65536.0;
  d.thresh = (uint32_t)(t + 0.5);
  if (d.thresh > 65535u) d.thresh = 65535u;
  if (d.thresh == 0) d.thresh = 1;
  d.scale = (float)(1.0 / (1.0 - (double)d.thresh / 65536.0));
  return d;
}
__device__ __forceinline__ float drop_scale1(const DropCfg& d, uint64_t id) {
  const uint64_t z = mix64(d.seed, id >> 2);
  const uint32_t u = (uint32_t)(z >> (16 * (id & 3))) & 0xFFFFu;
  return u < d.thresh ? 0.f : d.scale;
}
// id must be a multiple of 4
__device__ __forceinline__ void drop_scale4(const DropCfg& d, uint64_t id, float s[4]) {
  const uint64_t z = mix64(d.seed, id >> 2);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t u = (uint32_t)(z >> (16 * j)) & 0xFFFFu;
    s[j] = u < d.thresh ? 0.f : d.scale;
  }
}

// ---- wave / block reductions -----------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

int launch_gemm(const rpde_gemm_desc& d, hipStream_t stream);

inline rpde_gemm_desc gemm_desc() {
  rpde_gemm_desc d;
  memset(&d, 0, sizeof(d));
  d.batch = 1; d.zdiv = 1; d.ksplit = 1; d.alpha = 1.f;
  d.a_kmajor = 1; d.b_kmajor = 1;
  return d;
}

}  // namespace rpde
