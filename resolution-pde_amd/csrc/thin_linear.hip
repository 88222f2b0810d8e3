// Linear layers with at most four features on one side: the lifting (1-3 input channels -> width) and the projection
// (width -> 1-2 output channels) of every model on the path (reference: models/ffno.py:113,121,225,233 in_proj /
// out_proj; models/fno.py fc0 / fc2 through LinearMLP1d).  As GEMMs these are degenerate -- K = 3 or N = 1 falls
// off the 16-byte vector path and runs at a tenth of the streaming rate -- while as streaming kernels they are
// trivially HBM bound: the wide side (P x O floats) is read or written exactly once, the thin side is 4-16 B/point.
//
//   expand    y[p][o] = bias[o] + sum_t a[p][t] W(o,t)         lifting forward; projection data gradient
//   contract  y[p][t] = bias[t] + sum_o a[p][o] W(t,o)         projection forward; lifting data gradient
//   outer     G[t][o] = sum_p a[p][t] b[p][o], cs[o] = sum_p b[p][o], ts[t] = sum_p a[p][t]
//                                                            both weight gradients and both bias gradients
// with T <= 4 thin features and O (a multiple of 4, at most 256) wide ones.  `outer` writes one partial slab per
// workgroup, folded by k_thin_fold in fixed order (no float atomics: bitwise reproducible).
#include "rpde_internal.h"
#include "pointwise.h"
#include "thin_linear.h"

#include <stdlib.h>

namespace rpde {

constexpr int TL_THREADS = 256;

// lanes of a point: O4 = O / 4 (each four consecutive wide features); 256 / O4 points per sweep of a workgroup
template <int T>
__global__ __launch_bounds__(TL_THREADS) void k_thin_expand(const float* __restrict__ a, const float* __restrict__ w, long ws_o,
                                                            long ws_t, const float* __restrict__ bias, float* __restrict__ y,
                                                            long P, int O) {
  const int O4 = O >> 2, og = threadIdx.x % O4, pl = threadIdx.x / O4, pps = TL_THREADS / O4;
  float wr[4][T], br[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    br[k] = bias ? bias[4 * og + k] : 0.f;
#pragma unroll
    for (int t = 0; t < T; ++t) wr[k][t] = w[(long)(4 * og + k) * ws_o + t * ws_t];
  }
  for (long p = (long)blockIdx.x * pps + pl; p < P; p += (long)gridDim.x * pps) {
    float av[T];
#pragma unroll
    for (int t = 0; t < T; ++t) av[t] = a[p * T + t];
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = br[k];
#pragma unroll
      for (int t = 0; t < T; ++t) s = fmaf(av[t], wr[k][t], s);
      o[k] = s;
    }
    *reinterpret_cast<float4*>(y + p * O + 4 * og) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// sum over the 16 lanes of a DPP row, valid in lane 15 of the row
__device__ __forceinline__ float tl_row_sum15(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));
  return v;
}

// 16 lanes per point, lane j owns features 4j + 64q (q < NQ = ceil(O / 64)); four points per wave instruction
template <int T, int NQ>
__global__ __launch_bounds__(TL_THREADS) void k_thin_contract(const float* __restrict__ a, const float* __restrict__ w, long ws_t,
                                                              long ws_o, const float* __restrict__ bias, float* __restrict__ y,
                                                              long P, int O) {
  const int li = threadIdx.x & 15, pl = threadIdx.x >> 4;          // 16 points per sweep of a workgroup
  float wr[T][NQ][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int o = 64 * q + 4 * li + k;
        wr[t][q][k] = o < O ? w[t * ws_t + (long)o * ws_o] : 0.f;
      }
  float br[T];
#pragma unroll
  for (int t = 0; t < T; ++t) br[t] = bias ? bias[t] : 0.f;
  const long sweeps = (P + 15) / 16;
  // (four sweeps per pass with their loads issued together was measured slower: 138 vs 126 us)
  for (long sw = blockIdx.x; sw < sweeps; sw += gridDim.x) {        // whole rows of 16 lanes stay together: DPP sums
    const long p = sw * 16 + pl;
    const bool live = p < P;
    float s[T];
#pragma unroll
    for (int t = 0; t < T; ++t) s[t] = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int o = 64 * q + 4 * li;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && o < O) v = *reinterpret_cast<const float4*>(a + p * O + o);
#pragma unroll
      for (int t = 0; t < T; ++t)
        s[t] += (v.x * wr[t][q][0] + v.y * wr[t][q][1]) + (v.z * wr[t][q][2] + v.w * wr[t][q][3]);
    }
#pragma unroll
    for (int t = 0; t < T; ++t) s[t] = tl_row_sum15(s[t]);
    if (live && li == 15) {
#pragma unroll
      for (int t = 0; t < T; ++t) y[p * T + t] = s[t] + br[t];
    }
  }
}

// per workgroup: slab[(T + 1) * O + 4] = G[T][O], cs[O], ts[T] (+ padding: 16-byte rows) over its points
template <int T>
__global__ __launch_bounds__(TL_THREADS) void k_thin_outer(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ slabs, long P, int O, long pts_per_block) {
  __shared__ float red[TL_THREADS * 4];
  const int O4 = O >> 2, og = threadIdx.x % O4, pl = threadIdx.x / O4, pps = TL_THREADS / O4;
  const long p0 = (long)blockIdx.x * pts_per_block, p1 = min(P, p0 + pts_per_block);
  float acc[T][4], cs[4] = {0.f, 0.f, 0.f, 0.f}, ts[T];
#pragma unroll
  for (int t = 0; t < T; ++t) { ts[t] = 0.f; acc[t][0] = acc[t][1] = acc[t][2] = acc[t][3] = 0.f; }
  // eight points per thread in flight (all loads first, then the multiply-adds in point order: the sums do not change).
  // With one load per thread and iteration the kernel ran at 2 TB/s -- 16 KB in flight per CU.
  constexpr int UN = 8;
  long p = p0 + pl;
  for (; p + (long)(UN - 1) * pps < p1; p += (long)UN * pps) {
    float4 v[UN];
    float at[UN][T];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      v[u] = *reinterpret_cast<const float4*>(b + (p + (long)u * pps) * O + 4 * og);
#pragma unroll
      for (int t = 0; t < T; ++t) at[u][t] = a[(p + (long)u * pps) * T + t];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      cs[0] += v[u].x; cs[1] += v[u].y; cs[2] += v[u].z; cs[3] += v[u].w;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        ts[t] += at[u][t];
        acc[t][0] = fmaf(at[u][t], v[u].x, acc[t][0]); acc[t][1] = fmaf(at[u][t], v[u].y, acc[t][1]);
        acc[t][2] = fmaf(at[u][t], v[u].z, acc[t][2]); acc[t][3] = fmaf(at[u][t], v[u].w, acc[t][3]);
      }
    }
  }
  for (; p < p1; p += pps) {
    const float4 v = *reinterpret_cast<const float4*>(b + p * O + 4 * og);
    cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const float at = a[p * T + t];
      ts[t] += at;
      acc[t][0] = fmaf(at, v.x, acc[t][0]); acc[t][1] = fmaf(at, v.y, acc[t][1]);
      acc[t][2] = fmaf(at, v.z, acc[t][2]); acc[t][3] = fmaf(at, v.w, acc[t][3]);
    }
  }
  float* const slab = slabs + (long)blockIdx.x * ((T + 1) * O + 4);
  // fold the pps point-lanes of each feature group through LDS, one quantity at a time, in fixed order
#pragma unroll
  for (int r = 0; r <= T + 1; ++r) {
    float4 v;
    if (r < T) v = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
    else if (r == T) v = make_float4(cs[0], cs[1], cs[2], cs[3]);
    else v = make_float4(T > 0 ? ts[0] : 0.f, T > 1 ? ts[T > 1 ? 1 : 0] : 0.f, T > 2 ? ts[T > 2 ? 2 : 0] : 0.f, T > 3 ? ts[T > 3 ? 3 : 0] : 0.f);
    __syncthreads();
    *reinterpret_cast<float4*>(red + threadIdx.x * 4) = v;
    __syncthreads();
    if (pl == 0 && (r <= T || og == 0)) {
      float4 s = v;
      for (int j = 1; j < pps; ++j) {
        const float4 u = *reinterpret_cast<const float4*>(red + (j * O4 + og) * 4);
        s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
      }
      if (r <= T) {
        *reinterpret_cast<float4*>(slab + r * O + 4 * og) = s;
      } else {
        // thin sums: every feature group carried the same values, group 0 stands for all (divide nothing: each point
        // lane saw each of its points once, and only group 0 is folded)
        const float tv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int t = 0; t < T; ++t) slab[(T + 1) * O + t] = tv[t];
      }
    }
  }
}

bool thin_linear_ok(int thin, int wide) {
  const char* e = getenv("RPDE_THIN_LINEAR");
  if (e && e[0] == '0') return false;
  return thin >= 1 && thin <= 4 && wide >= 4 && wide <= 256 && wide % 4 == 0 && TL_THREADS % (wide / 4) == 0;
}

// grid-stride kernels: one workgroup per sweep up to 2048 workgroups (8 per CU)
static int tl_grid(long sweeps) {
  long g = sweeps;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

int thin_expand(const float* a, const float* w, long ws_o, long ws_t, const float* bias, float* y, long P, int T, int O,
                hipStream_t st) {
  RPDE_CHECK_ARG(thin_linear_ok(T, O), "thin_expand: unsupported shape %d x %d", T, O);
  const int pps = TL_THREADS / (O / 4);
  const dim3 grid(tl_grid((P + pps - 1) / pps)), block(TL_THREADS);
  switch (T) {
    case 1: hipLaunchKernelGGL(k_thin_expand<1>, grid, block, 0, st, a, w, ws_o, ws_t, bias, y, P, O); break;
    case 2: hipLaunchKernelGGL(k_thin_expand<2>, grid, block, 0, st, a, w, ws_o, ws_t, bias, y, P, O); break;
    case 3: hipLaunchKernelGGL(k_thin_expand<3>, grid, block, 0, st, a, w, ws_o, ws_t, bias, y, P, O); break;
    default: hipLaunchKernelGGL(k_thin_expand<4>, grid, block, 0, st, a, w, ws_o, ws_t, bias, y, P, O); break;
  }
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

template <int T>
static void tl_contract_launch(const float* a, const float* w, long ws_t, long ws_o, const float* bias, float* y, long P, int O,
                               hipStream_t st) {
  const dim3 grid(tl_grid((P + 15) / 16)), block(TL_THREADS);
  const int nq = (O + 63) / 64;
  if (nq == 1) hipLaunchKernelGGL((k_thin_contract<T, 1>), grid, block, 0, st, a, w, ws_t, ws_o, bias, y, P, O);
  else if (nq == 2) hipLaunchKernelGGL((k_thin_contract<T, 2>), grid, block, 0, st, a, w, ws_t, ws_o, bias, y, P, O);
  else hipLaunchKernelGGL((k_thin_contract<T, 4>), grid, block, 0, st, a, w, ws_t, ws_o, bias, y, P, O);
}

int thin_contract(const float* a, const float* w, long ws_t, long ws_o, const float* bias, float* y, long P, int T, int O,
                  hipStream_t st) {
  RPDE_CHECK_ARG(thin_linear_ok(T, O), "thin_contract: unsupported shape %d x %d", T, O);
  switch (T) {
    case 1: tl_contract_launch<1>(a, w, ws_t, ws_o, bias, y, P, O, st); break;
    case 2: tl_contract_launch<2>(a, w, ws_t, ws_o, bias, y, P, O, st); break;
    case 3: tl_contract_launch<3>(a, w, ws_t, ws_o, bias, y, P, O, st); break;
    default: tl_contract_launch<4>(a, w, ws_t, ws_o, bias, y, P, O, st); break;
  }
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// at least 128 points per workgroup (a sweep of 256 threads covers 4 .. 64 of them), at most 1024 workgroups = four per
// CU: with one (four waves per CU, one 16-byte load in flight per thread) the kernel ran at 1.8 TB/s, bound by load
// latency.  (Round 2 capped this at 256 because the fold over 1024 slabs took 65 us: the fold is now spread over sixteen
// slab groups per output.)
static int tl_outer_blocks(long P) {
  long nb = (P + 127) / 128;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (int)nb;
}
size_t thin_outer_ws_floats(long P, int T, int O) { return (size_t)tl_outer_blocks(P) * ((T + 1) * O + 4); }

// slabs [nb][(T + 1) * O + 4] -> G (as [T][O], or transposed [O][T]), cs [O], ts [T]; 16 outputs per workgroup, sixteen
// groups of threads split the slabs and are combined through LDS in fixed order
__global__ __launch_bounds__(256) void k_thin_fold(const float* __restrict__ slabs, int nb, int T, int O, float* __restrict__ G,
                                                   int g_transposed, float* __restrict__ cs, float* __restrict__ ts) {
  __shared__ float red[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, n = (T + 1) * O + T, stride = (T + 1) * O + 4;
  const int e = blockIdx.x * 16 + tx;
  float acc = 0.f;
  if (e < n) {
    // eight loads in flight per thread, summed in slab order (64 slabs per thread at 1024 slabs: one load at a time was
    // 20 us of pure latency, twice per step)
    int s = ty;
    for (; s + 112 < nb; s += 128) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = slabs[(long)(s + 16 * j) * stride + e];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    }
    for (; s < nb; s += 16) acc += slabs[(long)s * stride + e];
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && e < n) {
    acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += red[j][tx];
    if (e < T * O) {
      if (G) { const int t = e / O, o = e % O; G[g_transposed ? o * T + t : e] = acc; }
    } else if (e < (T + 1) * O) {
      if (cs) cs[e - T * O] = acc;
    } else if (ts) {
      ts[e - (T + 1) * O] = acc;
    }
  }
}

// G [T][O] (g_transposed: [O][T]), cs [O] and ts [T]: any of the three outputs may be null
int thin_outer(const float* a, const float* b, float* G, int g_transposed, float* cs, float* ts, long P, int T, int O, float* ws,
               hipStream_t st) {
  RPDE_CHECK_ARG(thin_linear_ok(T, O) && ws, "thin_outer: unsupported shape %d x %d", T, O);
  const int nb = tl_outer_blocks(P);
  const long ppb = (P + nb - 1) / nb;
  const dim3 grid(nb), block(TL_THREADS);
  switch (T) {
    case 1: hipLaunchKernelGGL(k_thin_outer<1>, grid, block, 0, st, a, b, ws, P, O, ppb); break;
    case 2: hipLaunchKernelGGL(k_thin_outer<2>, grid, block, 0, st, a, b, ws, P, O, ppb); break;
    case 3: hipLaunchKernelGGL(k_thin_outer<3>, grid, block, 0, st, a, b, ws, P, O, ppb); break;
    default: hipLaunchKernelGGL(k_thin_outer<4>, grid, block, 0, st, a, b, ws, P, O, ppb); break;
  }
  RPDE_LAUNCH_CHECK();
  const int n = (T + 1) * O + T;
  hipLaunchKernelGGL(k_thin_fold, dim3((n + 15) / 16), dim3(256), 0, st, ws, nb, T, O, G, g_transposed, cs, ts);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // namespace rpde
