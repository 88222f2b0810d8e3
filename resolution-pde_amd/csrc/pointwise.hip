// HBM-bound pointwise / reduction kernels around the GEMMs.
// All are coalesced (16 B per lane where the layout allows), wave64 shuffles
// for row reductions, deterministic two-stage sums (per-block partial slabs,
// then reduce_slabs) -- no float atomics, so results are run-to-run identical.
#include "rpde_internal.h"
#include "pointwise.h"

namespace rpde {

// ---------------------------------------------------------------------------
// out[i] (+)= scale * sum_s slabs[s*stride + i]
// ---------------------------------------------------------------------------
// 64 consecutive outputs per block; the 4 waves split the slabs, LDS combines them
__device__ __forceinline__ float slab_partial(const float* __restrict__ slabs, long i, int S, long stride, int ty) {
  float acc;
  {
    // with few outputs and many slabs this loop is pure load latency (2 blocks, 256 slabs: 13 us with eight loads in
    // flight per lane): 32 in flight while there are that many, then 8, then the rest -- the summation order is fixed by
    // S alone, so results stay reproducible
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = ty;
    for (; s + 124 < S; s += 128) {
      float v[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) v[j] = slabs[(long)(s + 4 * j) * stride + i];
#pragma unroll
      for (int j = 0; j < 32; ++j) a[j & 7] += v[j];
    }
    for (; s + 28 < S; s += 32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += slabs[(long)(s + 4 * j) * stride + i];
    }
    for (; s < S; s += 4) a[0] += slabs[(long)s * stride + i];
    acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  return acc;
}

__global__ __launch_bounds__(256) void k_reduce_slabs(const float* __restrict__ slabs, float* __restrict__ out, long n,
                                                      int S, long stride, float scale, int accumulate) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + tx;
  float acc = i < n ? slab_partial(slabs, i, S, stride, ty) : 0.f;
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && i < n) {
    acc = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    acc *= scale;
    if (accumulate) acc += out[i];
    out[i] = acc;
  }
}

// several independent folds, one launch: block -> job by the jobs' first-block numbers
__global__ __launch_bounds__(256) void k_fold_jobs(const FoldJobs J) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  int jn = 0;
#pragma unroll
  for (int q = 1; q < FoldJobs::MAX; ++q)
    if (q < J.n && (int)blockIdx.x >= J.j[q].blk0) jn = q;
  const float* src = J.j[0].src; float* dst = J.j[0].dst; long stride = J.j[0].stride; int len = J.j[0].len, S = J.j[0].S, b0 = 0;
#pragma unroll
  for (int q = 1; q < FoldJobs::MAX; ++q)          // (selects instead of a dynamically indexed kernel argument)
    if (q == jn) { src = J.j[q].src; dst = J.j[q].dst; stride = J.j[q].stride; len = J.j[q].len; S = J.j[q].S; b0 = J.j[q].blk0; }
  const long i = (long)((int)blockIdx.x - b0) * 64 + tx;
  float acc = i < len ? slab_partial(src, i, S, stride, ty) : 0.f;
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && i < len) dst[i] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

int fold_jobs(FoldJobs& jobs, hipStream_t st) {
  if (jobs.n == 0) return RPDE_OK;
  int blocks = 0;
  for (int q = 0; q < jobs.n; ++q) { jobs.j[q].blk0 = blocks; blocks += (jobs.j[q].len + 63) / 64; }
  hipLaunchKernelGGL(k_fold_jobs, dim3((unsigned)blocks), dim3(256), 0, st, jobs);
  RPDE_LAUNCH_CHECK();
  jobs.n = 0;
  return RPDE_OK;
}

int reduce_slabs(const float* slabs, float* out, long n, int S, long stride, float scale, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce_slabs, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, slabs, out, n, S, stride, scale, accumulate);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// the same fold for a row of up to eight segments that go to different tensors (bias / gamma / beta gradients of the fused
// FeedForward backward: one launch instead of five)
__global__ __launch_bounds__(256) void k_reduce_slabs_seg(const float* __restrict__ slabs, int S, long stride, ReduceSegs sg) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  float acc = 0.f;
  if (i < sg.n) {
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = ty;
    for (; s + 28 < S; s += 32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] += slabs[(long)(s + 4 * j) * stride + i];
    }
    for (; s < S; s += 4) a[0] += slabs[(long)s * stride + i];
    acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && i < sg.n) {
    acc = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < sg.nseg && i >= sg.off[k] && i < sg.off[k] + sg.len[k] && sg.dst[k]) sg.dst[k][i - sg.off[k]] = acc;
  }
}

int reduce_slabs_seg(const float* slabs, int S, long stride, const ReduceSegs& sg, hipStream_t st) {
  hipLaunchKernelGGL(k_reduce_slabs_seg, dim3((unsigned)((sg.n + 63) / 64)), dim3(256), 0, st, slabs, S, stride, sg);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// many slabs, few outputs (per-tile column sums): first fold S slabs into REDUCE_CHUNKS
// partial rows (tmp [REDUCE_CHUNKS][n]) with one block per (64 outputs, chunk), then fold those
__global__ __launch_bounds__(256) void k_reduce_slabs_chunk(const float* __restrict__ slabs, float* __restrict__ tmp, long n,
                                                            int S, long stride, int per_chunk) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + tx;
  const int s0 = blockIdx.y * per_chunk, s1 = min(S, s0 + per_chunk);
  float acc = 0.f;
  if (i < n)
    for (int s = s0 + ty; s < s1; s += 4) acc += slabs[(long)s * stride + i];
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && i < n) tmp[(long)blockIdx.y * n + i] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

int reduce_slabs_2pass(const float* slabs, float* out, long n, int S, long stride, float* tmp, hipStream_t st) {
  if (S <= 4 * REDUCE_CHUNKS) return reduce_slabs(slabs, out, n, S, stride, 1.f, 0, st);
  const int per = (S + REDUCE_CHUNKS - 1) / REDUCE_CHUNKS;
  hipLaunchKernelGGL(k_reduce_slabs_chunk, dim3((unsigned)((n + 63) / 64), REDUCE_CHUNKS), dim3(256), 0, st, slabs, tmp, n, S, stride, per);
  RPDE_LAUNCH_CHECK();
  return reduce_slabs(tmp, out, n, REDUCE_CHUNKS, n, 1.f, 0, st);
}

// ---------------------------------------------------------------------------
// column sums of x [P, N] (row stride ld): slab[block][n] partials
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ x, float* __restrict__ slab, long P, int N, long ld,
                                                long rows_per_block) {
  __shared__ float red[256];
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(P, r0 + rows_per_block);
  const int t = threadIdx.x;
  if (N <= 256 && 256 % N == 0) {
    const int tr = t / N, c = t % N, step = 256 / N;
    float acc = 0.f;
    for (long r = r0 + tr; r < r1; r += step) acc += x[r * ld + c];
    red[t] = acc;
    __syncthreads();
    if (tr == 0) {
      for (int j = 1; j < step; ++j) acc += red[j * N + c];
      slab[(long)blockIdx.x * N + c] = acc;
    }
  } else {
    for (int c = t; c < N; c += 256) {
      float acc = 0.f;
      for (long r = r0; r < r1; ++r) acc += x[r * ld + c];
      slab[(long)blockIdx.x * N + c] = acc;
    }
  }
}

size_t colsum_ws_floats(long P, int N) {
  long nb = (P + 255) / 256;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return (size_t)nb * N;
}

int colsum(const float* x, float* out, long P, int N, long ld, float* ws, int accumulate, hipStream_t st) {
  long nb = (P + 255) / 256;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  const long rpb = (P + nb - 1) / nb;
  hipLaunchKernelGGL(k_colsum, dim3((unsigned)nb), dim3(256), 0, st, x, ws, P, N, ld, rpb);
  RPDE_LAUNCH_CHECK();
  return reduce_slabs(ws, out, N, (int)nb, N, 1.f, accumulate, st);
}

// ---------------------------------------------------------------------------
// FeedForward tail: out = residual + post_act( LN( dropout(z) ) )
// One row of C floats is owned by G = C/4 lanes (float4 each); a wave holds
// 64/G rows.  C must be a multiple of 4 with C/4 in {1,2,4,...,64}; other
// widths take the one-wave-per-row path below.
// ---------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int G>
__global__ __launch_bounds__(256) void k_ff_tail_fwd(const float* __restrict__ z, const float* __restrict__ res,
                                                     float* __restrict__ out, long P, int layer_norm, float eps,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     DropCfg drop, int post_act) {
  drop = drop_resolve(drop);
  constexpr int C = 4 * G;
  const int lane_in_row = threadIdx.x % G;
  const long rows_per_block = 256 / G;
  for (long row = (long)blockIdx.x * rows_per_block + threadIdx.x / G; row < P; row += (long)gridDim.x * rows_per_block) {
    const long off = row * C + lane_in_row * 4;
    float4 v = *reinterpret_cast<const float4*>(z + off);
    if (drop.on()) {
      float s[4];
      drop_scale4(drop, (uint64_t)off, s);
      v.x *= s[0]; v.y *= s[1]; v.z *= s[2]; v.w *= s[3];
    }
    if (layer_norm) {
      const float mean = group_sum<G>(v.x + v.y + v.z + v.w) * (1.f / C);
      const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
      const float var = group_sum<G>(dx * dx + dy * dy + dz * dz + dw * dw) * (1.f / C);
      const float rstd = rsqrtf(var + eps);
      const float4 gm = *reinterpret_cast<const float4*>(gamma + lane_in_row * 4);
      const float4 bt = *reinterpret_cast<const float4*>(beta + lane_in_row * 4);
      v.x = dx * rstd * gm.x + bt.x;
      v.y = dy * rstd * gm.y + bt.y;
      v.z = dz * rstd * gm.z + bt.z;
      v.w = dw * rstd * gm.w + bt.w;
    }
    if (post_act) {
      v.x = act_f(post_act, v.x); v.y = act_f(post_act, v.y);
      v.z = act_f(post_act, v.z); v.w = act_f(post_act, v.w);
    }
    if (res) {
      const float4 r = *reinterpret_cast<const float4*>(res + off);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    *reinterpret_cast<float4*>(out + off) = v;
  }
}

// generic width: one wave per row, up to 8 elements per lane (C <= 512)
__global__ __launch_bounds__(256) void k_ff_tail_fwd_any(const float* __restrict__ z, const float* __restrict__ res,
                                                         float* __restrict__ out, long P, int C, int layer_norm, float eps,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         DropCfg drop, int post_act) {
  drop = drop_resolve(drop);
  const int lane = threadIdx.x & 63;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < P; row += (long)gridDim.x * 4) {
    float v[8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane + 64 * j;
      v[j] = 0.f;
      if (c < C) {
        const long off = row * C + c;
        v[j] = z[off];
        if (drop.on()) v[j] *= drop_scale1(drop, (uint64_t)off);
        sum += v[j];
      }
    }
    float mean = 0.f, rstd = 1.f;
    if (layer_norm) {
      mean = wave_sum(sum) / C;
      float sq = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) if (lane + 64 * j < C) sq += (v[j] - mean) * (v[j] - mean);
      rstd = rsqrtf(wave_sum(sq) / C + eps);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane + 64 * j;
      if (c < C) {
        float y = v[j];
        if (layer_norm) y = (y - mean) * rstd * gamma[c] + beta[c];
        y = act_f(post_act, y);
        const long off = row * C + c;
        if (res) y += res[off];
        out[off] = y;
      }
    }
  }
}

int ff_tail_fwd(const float* z, const float* res, float* out, long P, int C, int layer_norm, float eps,
                const float* gamma, const float* beta, DropCfg drop, int post_act, hipStream_t st) {
  RPDE_CHECK_ARG(!layer_norm || (gamma && beta), "ff_tail: layer_norm needs gamma/beta");
  const int G = C / 4;
  const bool vec = (C % 4 == 0) && (G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64);
  if (!vec) {
    RPDE_CHECK_ARG(C <= 512, "ff_tail: width %d > 512 unsupported", C);
    long nb = (P + 3) / 4; if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_ff_tail_fwd_any, dim3((unsigned)nb), dim3(256), 0, st, z, res, out, P, C, layer_norm, eps, gamma, beta, drop, post_act);
    RPDE_LAUNCH_CHECK();
    return RPDE_OK;
  }
  const long rpb = 256 / G;
  long nb = (P + rpb - 1) / rpb; if (nb > 8192) nb = 8192;
#define LAUNCH_G(GG) hipLaunchKernelGGL((k_ff_tail_fwd<GG>), dim3((unsigned)nb), dim3(256), 0, st, z, res, out, P, layer_norm, eps, gamma, beta, drop, post_act)
  switch (G) {
    case 1: LAUNCH_G(1); break; case 2: LAUNCH_G(2); break; case 4: LAUNCH_G(4); break;
    case 8: LAUNCH_G(8); break; case 16: LAUNCH_G(16); break; case 32: LAUNCH_G(32); break;
    default: LAUNCH_G(64); break;
  }
#undef LAUNCH_G
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// ---------------------------------------------------------------------------
// backward of the tail.  g = d(out).  Produces dz (through post_act', LN and
// the dropout mask) and per-block partial sums for d(gamma), d(beta):
// slab[block][0:C] = sum dy*xhat, slab[block][C:2C] = sum dy.
// One wave per row (any C <= 512): simple and HBM bound.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ff_tail_bwd(const float* __restrict__ z, const float* __restrict__ g,
                                                     float* __restrict__ dz, float* __restrict__ slab, long P, int C,
                                                     int layer_norm, float eps, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, DropCfg drop, int post_act) {
  drop = drop_resolve(drop);
  __shared__ float red[4][2][512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float dg[8], db[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { dg[j] = 0.f; db[j] = 0.f; }
  for (long row = (long)blockIdx.x * 4 + w; row < P; row += (long)gridDim.x * 4) {
    float t[8], s[8], gy[8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane + 64 * j;
      t[j] = 0.f; s[j] = 1.f; gy[j] = 0.f;
      if (c < C) {
        const long off = row * C + c;
        if (drop.on()) s[j] = drop_scale1(drop, (uint64_t)off);
        t[j] = z[off] * s[j];
        gy[j] = g[off];
        sum += t[j];
      }
    }
    if (!layer_norm) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = lane + 64 * j;
        if (c < C) dz[row * C + c] = gy[j] * dact_f(post_act, t[j]) * s[j];
      }
      continue;
    }
    const float mean = wave_sum(sum) / C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) if (lane + 64 * j < C) sq += (t[j] - mean) * (t[j] - mean);
    const float rstd = rsqrtf(wave_sum(sq) / C + eps);
    float xh[8], dxh[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane + 64 * j;
      xh[j] = 0.f; dxh[j] = 0.f;
      if (c < C) {
        xh[j] = (t[j] - mean) * rstd;
        const float gm = gamma[c];
        float dy = gy[j];
        if (post_act) dy *= dact_f(post_act, xh[j] * gm + beta[c]);
        dg[j] += dy * xh[j];
        db[j] += dy;
        dxh[j] = dy * gm;
        s1 += dxh[j];
        s2 += dxh[j] * xh[j];
      }
    }
    s1 = wave_sum(s1) / C;
    s2 = wave_sum(s2) / C;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = lane + 64 * j;
      if (c < C) dz[row * C + c] = rstd * (dxh[j] - s1 - xh[j] * s2) * s[j];
    }
  }
  if (!layer_norm) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    red[w][0][lane + 64 * j] = dg[j];
    red[w][1][lane + 64 * j] = db[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    slab[(long)blockIdx.x * 2 * C + c] = red[0][0][c] + red[1][0][c] + red[2][0][c] + red[3][0][c];
    slab[(long)blockIdx.x * 2 * C + C + c] = red[0][1][c] + red[1][1][c] + red[2][1][c] + red[3][1][c];
  }
}

// vector form: a row of C = 4G floats is owned by G lanes (float4 each); every
// thread keeps the same 4 columns for all its rows, so d(gamma), d(beta) and the
// bias gradient sum(dz) accumulate in registers and meet once in LDS.
// slab[block][3][C] = (sum dy*xhat, sum dy, sum dz)
template <int G>
__global__ __launch_bounds__(256) void k_ff_tail_bwd_vec(const float* __restrict__ z, const float* __restrict__ g,
                                                         float* __restrict__ dz, float* __restrict__ slab, long P,
                                                         int layer_norm, float eps, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, DropCfg drop, int post_act) {
  drop = drop_resolve(drop);
  constexpr int C = 4 * G;
  constexpr int RPB = 256 / G;
  __shared__ float red[3][RPB][C];
  const int lir = threadIdx.x % G, rib = threadIdx.x / G;
  float4 gm = make_float4(1.f, 1.f, 1.f, 1.f), bt = make_float4(0.f, 0.f, 0.f, 0.f);
  if (layer_norm) {
    gm = *reinterpret_cast<const float4*>(gamma + lir * 4);
    bt = *reinterpret_cast<const float4*>(beta + lir * 4);
  }
  float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, ds[4] = {0.f, 0.f, 0.f, 0.f};
  const float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
  for (long row = (long)blockIdx.x * RPB + rib; row < P; row += (long)gridDim.x * RPB) {
    const long off = row * C + lir * 4;
    const float4 z4 = *reinterpret_cast<const float4*>(z + off);
    const float4 g4 = *reinterpret_cast<const float4*>(g + off);
    float s[4] = {1.f, 1.f, 1.f, 1.f};
    if (drop.on()) drop_scale4(drop, (uint64_t)off, s);
    const float t[4] = {z4.x * s[0], z4.y * s[1], z4.z * s[2], z4.w * s[3]};
    const float gy[4] = {g4.x, g4.y, g4.z, g4.w};
    float o[4];
    if (layer_norm) {
      const float mean = group_sum<G>(t[0] + t[1] + t[2] + t[3]) * (1.f / C);
      float d[4], sq = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { d[k] = t[k] - mean; sq += d[k] * d[k]; }
      const float rstd = rsqrtf(group_sum<G>(sq) * (1.f / C) + eps);
      float xh[4], dxh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xh[k] = d[k] * rstd;
        float dy = gy[k];
        if (post_act) dy *= dact_f(post_act, xh[k] * gmv[k] + btv[k]);
        dg[k] += dy * xh[k];
        db[k] += dy;
        dxh[k] = dy * gmv[k];
        s1 += dxh[k];
        s2 += dxh[k] * xh[k];
      }
      s1 = group_sum<G>(s1) * (1.f / C);
      s2 = group_sum<G>(s2) * (1.f / C);
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = rstd * (dxh[k] - s1 - xh[k] * s2) * s[k];
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = gy[k] * dact_f(post_act, t[k]) * s[k];
    }
    *reinterpret_cast<float4*>(dz + off) = make_float4(o[0], o[1], o[2], o[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) ds[k] += o[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    red[0][rib][lir * 4 + k] = dg[k];
    red[1][rib][lir * 4 + k] = db[k];
    red[2][rib][lir * 4 + k] = ds[k];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 3 * C; idx += 256) {
    const int w = idx / C, c = idx % C;
    float a = 0.f;
#pragma unroll 4
    for (int r = 0; r < RPB; ++r) a += red[w][r][c];
    slab[(long)blockIdx.x * 3 * C + idx] = a;
  }
}

static long tail_bwd_blocks(long P) {
  long nb = (P + 3) / 4;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  return nb;
}
static bool tail_vec_ok(int C) {
  const int G = C / 4;
  return (C % 4 == 0) && (G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64);
}
size_t ff_tail_bwd_ws_floats(long P, int C) { return (size_t)tail_bwd_blocks(P) * 3 * C; }

// grad_bias (optional) receives colsum(dz) when the vector kernel runs; returns whether it did
int ff_tail_bwd(const float* z, const float* g, float* dz, long P, int C, int layer_norm, float eps, const float* gamma,
                const float* beta, DropCfg drop, int post_act, float* grad_gamma, float* grad_beta, float* grad_bias,
                int* bias_done, float* ws, hipStream_t st, FoldJobs* defer) {
  *bias_done = 0;
  // the per-block partial sums are folded here, or (defer) by the caller's one fold launch -- same sums
  auto fold = [&](const float* src, float* dst, int S, long stride) -> int {
    if (defer && defer->add(src, dst, C, S, stride)) return RPDE_OK;
    return reduce_slabs(src, dst, C, S, stride, 1.f, 0, st);
  };
  if (tail_vec_ok(C)) {
    const int G = C / 4;
    const long rpb = 256 / G;
    long nb = (P + rpb - 1) / rpb; if (nb > 1024) nb = 1024; if (nb < 1) nb = 1;
#define LAUNCH_G(GG) hipLaunchKernelGGL((k_ff_tail_bwd_vec<GG>), dim3((unsigned)nb), dim3(256), 0, st, z, g, dz, ws, P, layer_norm, eps, gamma, beta, drop, post_act)
    switch (G) {
      case 1: LAUNCH_G(1); break; case 2: LAUNCH_G(2); break; case 4: LAUNCH_G(4); break;
      case 8: LAUNCH_G(8); break; case 16: LAUNCH_G(16); break; case 32: LAUNCH_G(32); break;
      default: LAUNCH_G(64); break;
    }
#undef LAUNCH_G
    RPDE_LAUNCH_CHECK();
    if (layer_norm) {
      if (grad_gamma) RPDE_TRY(fold(ws, grad_gamma, (int)nb, 3L * C));
      if (grad_beta) RPDE_TRY(fold(ws + C, grad_beta, (int)nb, 3L * C));
    }
    if (grad_bias) { RPDE_TRY(fold(ws + 2 * C, grad_bias, (int)nb, 3L * C)); *bias_done = 1; }
    return RPDE_OK;
  }
  RPDE_CHECK_ARG(C <= 512, "ff_tail_bwd: width %d > 512 unsupported", C);
  const long nb = tail_bwd_blocks(P);
  hipLaunchKernelGGL(k_ff_tail_bwd, dim3((unsigned)nb), dim3(256), 0, st, z, g, dz, ws, P, C, layer_norm, eps, gamma, beta, drop, post_act);
  RPDE_LAUNCH_CHECK();
  if (layer_norm) {
    if (grad_gamma) RPDE_TRY(fold(ws, grad_gamma, (int)nb, 2L * C));
    if (grad_beta) RPDE_TRY(fold(ws + C, grad_beta, (int)nb, 2L * C));
  }
  return RPDE_OK;
}

// ---------------------------------------------------------------------------
// per-mode complex weights [Ci,Co,K,2] <-> real block matrices [k][2Ci][2Co]
//   (re,i)->(re,o): Wr   (im,i)->(re,o): -Wi   (re,i)->(im,o): Wi   (im,i)->(im,o): Wr
// ---------------------------------------------------------------------------
__global__ void k_pack_mix(const float* __restrict__ w, float* __restrict__ blk, int Ci, int Co, int K, int keff) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per = 4L * Ci * Co;
  if (idx >= per * keff) return;
  const int k = (int)(idx / per);
  const long r = idx % per;
  const int row = (int)(r / (2 * Co)), col = (int)(r % (2 * Co));
  const int ri_in = row / Ci, i = row % Ci, ri_out = col / Co, o = col % Co;
  const float* p = w + (((long)i * Co + o) * K + k) * 2;
  float v;
  if (ri_in == ri_out) v = p[0];
  else v = ri_in ? -p[1] : p[1];
  blk[idx] = v;
}

int pack_mix_weights(const float* w, float* blk, int Ci, int Co, int K, int keff, hipStream_t st) {
  const long tot = 4L * Ci * Co * keff;
  hipLaunchKernelGGL(k_pack_mix, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, w, blk, Ci, Co, K, keff);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// gw[i,o,k,:] = fold( sum_s slab[s][k][2Ci][2Co] ), zero for k >= keff
__global__ void k_unpack_mix_grad(const float* __restrict__ slabs, float* __restrict__ gw, int Ci, int Co, int K, int keff,
                                  int S, long sstride) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Ci * Co * K) return;
  const int k = (int)(idx % K);
  const long io = idx / K;
  const int o = (int)(io % Co), i = (int)(io / Co);
  float gr = 0.f, gi = 0.f;
  if (k < keff) {
    const long per = 4L * Ci * Co, ld = 2L * Co;
    for (int s = 0; s < S; ++s) {
      const float* b = slabs + (long)s * sstride + (long)k * per;
      gr += b[(long)i * ld + o] + b[(long)(Ci + i) * ld + Co + o];
      gi += b[(long)i * ld + Co + o] - b[(long)(Ci + i) * ld + o];
    }
  }
  gw[idx * 2] = gr;
  gw[idx * 2 + 1] = gi;
}

int unpack_mix_grad(const float* slabs, float* gw, int Ci, int Co, int K, int keff, int S, long sstride, hipStream_t st) {
  const long tot = (long)Ci * Co * K;
  hipLaunchKernelGGL(k_unpack_mix_grad, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, slabs, gw, Ci, Co, K, keff, S, sstride);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// ---------------------------------------------------------------------------
// weight normalisation of WNLinear (models/custom_layer.py:70-108; torch.nn.utils.weight_norm, dim 0):
//   w[o,:] = v[o,:] * (g[o] / |v[o,:]|)
//   gg[o]  = <gw[o,:], v[o,:]> / |v[o,:]|
//   gv[o,:] = (g[o] / |v[o,:]|) * (gw[o,:] - v[o,:] * <gw[o,:], v[o,:]> / |v[o,:]|^2)
// one wave per row (the reference's graph is a dozen ATen kernels on [128,1] / [1,128] tensors per step)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_weight_norm_fwd(const float* __restrict__ v, const float* __restrict__ g,
                                                         float* __restrict__ w, int out_f, int in_f) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= out_f) return;
  const float* vr = v + (long)o * in_f;
  float ss = 0.f;
  for (int i = lane; i < in_f; i += 64) ss = fmaf(vr[i], vr[i], ss);
  const float scale = g[o] / sqrtf(wave_sum(ss));
  for (int i = lane; i < in_f; i += 64) w[(long)o * in_f + i] = vr[i] * scale;
}

__global__ __launch_bounds__(256) void k_weight_norm_bwd(const float* __restrict__ v, const float* __restrict__ g,
                                                         const float* __restrict__ gw, float* __restrict__ gv,
                                                         float* __restrict__ gg, int out_f, int in_f) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= out_f) return;
  const float* vr = v + (long)o * in_f;
  const float* gr = gw + (long)o * in_f;
  float ss = 0.f, dot = 0.f;
  for (int i = lane; i < in_f; i += 64) { ss = fmaf(vr[i], vr[i], ss); dot = fmaf(gr[i], vr[i], dot); }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float nrm = sqrtf(ss), scale = g[o] / nrm, back = dot / ss;
  if (gg && lane == 0) gg[o] = dot / nrm;
  if (gv)
    for (int i = lane; i < in_f; i += 64) gv[(long)o * in_f + i] = scale * (gr[i] - vr[i] * back);
}

// ---------------------------------------------------------------------------
// relative L2 loss
// ---------------------------------------------------------------------------
constexpr int L2_BLOCKS = 64;   // partial blocks per sample

__global__ __launch_bounds__(256) void k_rel_l2_partial(const float* __restrict__ x, const float* __restrict__ y,
                                                        float* __restrict__ part, long per) {
  __shared__ float red[2][4];
  const int b = blockIdx.y;
  const float* xb = x + (long)b * per;
  const float* yb = y + (long)b * per;
  float sd = 0.f, sy = 0.f;
  const bool vec = (per % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
  if (vec) {
    const long nv = per / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
      const float4 a = reinterpret_cast<const float4*>(xb)[i];
      const float4 c = reinterpret_cast<const float4*>(yb)[i];
      const float d0 = a.x - c.x, d1 = a.y - c.y, d2 = a.z - c.z, d3 = a.w - c.w;
      sd += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
      sy += c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
      const float d = xb[i] - yb[i];
      sd += d * d;
      sy += yb[i] * yb[i];
    }
  }
  sd = wave_sum(sd);
  sy = wave_sum(sy);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = sd; red[1][w] = sy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[((long)b * gridDim.x + blockIdx.x) * 2] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    part[((long)b * gridDim.x + blockIdx.x) * 2 + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

// one block: stats[b] = (|x-y|, |y|), rel[b], loss = mean/sum
__global__ void k_rel_l2_final(const float* __restrict__ part, int nblk, float* __restrict__ rel, float* __restrict__ loss,
                               float* __restrict__ stats, int B, int size_average) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float sd = 0.f, sy = 0.f;
    for (int j = 0; j < nblk; ++j) { sd += part[((long)b * nblk + j) * 2]; sy += part[((long)b * nblk + j) * 2 + 1]; }
    const float dn = sqrtf(sd), yn = sqrtf(sy);
    const float r = dn / (yn + 1e-8f);
    stats[2 * b] = dn; stats[2 * b + 1] = yn;
    if (rel) rel[b] = r;
    acc += r;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss) *loss = size_average ? red[0] / B : red[0];
}

__global__ __launch_bounds__(256) void k_rel_l2_bwd(const float* __restrict__ x, const float* __restrict__ y,
                                                    const float* __restrict__ stats, const float* __restrict__ grad_loss,
                                                    const float* __restrict__ grad_rel, float* __restrict__ gx, int B, long per,
                                                    int size_average) {
  const int b = blockIdx.y;
  const float dn = stats[2 * b], yn = stats[2 * b + 1];
  float gr = grad_rel ? grad_rel[b] : (size_average ? grad_loss[0] / B : grad_loss[0]);
  const float coef = dn > 0.f ? gr / (dn * (yn + 1e-8f)) : 0.f;
  const float* xb = x + (long)b * per;
  const float* yb = y + (long)b * per;
  float* gb = gx + (long)b * per;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) gb[i] = coef * (xb[i] - yb[i]);
}

}  // namespace rpde

using namespace rpde;

extern "C" {

int rpde_weight_norm_fwd(const float* v, const float* g, float* w, int out_f, int in_f, void* stream) {
  RPDE_CHECK_ARG(v && g && w && out_f > 0 && in_f > 0, "weight_norm_fwd: bad arguments");
  hipLaunchKernelGGL(k_weight_norm_fwd, dim3((out_f + 3) / 4), dim3(256), 0, as_stream(stream), v, g, w, out_f, in_f);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

int rpde_weight_norm_bwd(const float* v, const float* g, const float* grad_w, float* grad_v, float* grad_g, int out_f, int in_f,
                         void* stream) {
  RPDE_CHECK_ARG(v && g && grad_w && (grad_v || grad_g) && out_f > 0 && in_f > 0, "weight_norm_bwd: bad arguments");
  hipLaunchKernelGGL(k_weight_norm_bwd, dim3((out_f + 3) / 4), dim3(256), 0, as_stream(stream), v, g, grad_w, grad_v, grad_g, out_f, in_f);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}


int rpde_rel_l2_fwd(const float* x, const float* y, float* rel, float* loss, float* stats, int B, int64_t per,
                    int size_average, void* stream) {
  RPDE_CHECK_ARG(x && y && stats && B > 0 && per > 0, "rel_l2_fwd: bad arguments");
  hipStream_t st = as_stream(stream);
  // partial sums live behind the stats the caller keeps: stats has 2*B floats,
  // the partials need 2*B*L2_BLOCKS more -> caller allocates stats with
  // rpde_rel_l2_stats_elems(B) floats.
  float* part = stats + 2L * B;
  hipLaunchKernelGGL(k_rel_l2_partial, dim3(L2_BLOCKS, B), dim3(256), 0, st, x, y, part, (long)per);
  RPDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_rel_l2_final, dim3(1), dim3(256), 0, st, part, L2_BLOCKS, rel, loss, stats, B, size_average);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

int64_t rpde_rel_l2_stats_elems(int B) { return 2L * B * (1 + L2_BLOCKS); }

int rpde_rel_l2_bwd(const float* x, const float* y, const float* stats, const float* grad_loss, const float* grad_rel,
                    float* grad_x, int B, int64_t per, int size_average, void* stream) {
  RPDE_CHECK_ARG(x && y && stats && grad_x && (grad_loss || grad_rel), "rel_l2_bwd: bad arguments");
  long nb = (per + 1023) / 1024; if (nb > 256) nb = 256; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_rel_l2_bwd, dim3((unsigned)nb, B), dim3(256), 0, as_stream(stream), x, y, stats, grad_loss, grad_rel, grad_x, B, (long)per, size_average);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// layout helpers at the model boundary
// ---------------------------------------------------------------------------
namespace rpde {

// out (channels-last [B,S,Ct] or channels-first [B,Ct,S]) = cat(x [B,Cin,S], grid coords)
__global__ void k_concat_grid(const float* __restrict__ x, float* __restrict__ out, int B, int Cin, int M, int N, int G,
                              double lo, double hi, int channels_last, const float* __restrict__ gx,
                              const float* __restrict__ gy) {
  const long S = (long)M * N;
  const int Ct = Cin + G;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)B * S * Ct) return;
  int c; long s; int b;
  if (channels_last) { c = (int)(idx % Ct); s = (idx / Ct) % S; b = (int)(idx / (Ct * S)); }
  else { s = idx % S; c = (int)((idx / S) % Ct); b = (int)(idx / (S * Ct)); }
  float v;
  if (c < Cin) {
    v = x[((long)b * Cin + c) * S + s];
  } else {
    const int axis = c - Cin;                 // 0: first spatial dim, 1: second
    const int len = (G == 2 && axis == 1) ? N : M;
    const int pos = (G == 2) ? (axis == 0 ? (int)(s / N) : (int)(s % N)) : (int)s;
    const float* tab = axis == 0 ? gx : gy;
    if (tab) v = tab[pos];
    else {
      // numpy.linspace(lo, hi, len) in double, then cast (quirk Q10)
      const double step = len > 1 ? (hi - lo) / (double)(len - 1) : 0.0;
      v = (pos == len - 1 && len > 1) ? (float)hi : (float)(lo + step * pos);
    }
  }
  out[idx] = v;
}

// tiled transpose [B,S,C] <-> [B,C,S]
__global__ __launch_bounds__(256) void k_transpose(const float* __restrict__ in, float* __restrict__ out, long R, long Cc) {
  // in [B][R][Cc] -> out [B][Cc][R]
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* ib = in + (long)b * R * Cc;
  float* ob = out + (long)b * R * Cc;
  for (int j = ty; j < 32; j += 8) {
    const long r = r0 + j, c = c0 + tx;
    if (r < R && c < Cc) tile[j][tx] = ib[r * Cc + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const long c = c0 + j, r = r0 + tx;
    if (r < R && c < Cc) ob[c * R + r] = tile[tx][j];
  }
}

__global__ void k_act_fwd(const float* __restrict__ x, float* __restrict__ out, long n, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = act_f(act, x[i]);
}
__global__ void k_act_bwd(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dx, long n, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = g[i] * dact_f(act, x[i]);
}

}  // namespace rpde

extern "C" {

int rpde_concat_grid(const float* x, float* out, int B, int Cin, int M, int N, int grid_dims, double lo, double hi,
                     int channels_last, const float* gridx, const float* gridy, void* stream) {
  RPDE_CHECK_ARG(x && out && B > 0 && Cin > 0 && M > 0 && N > 0, "concat_grid: bad arguments");
  RPDE_CHECK_ARG(grid_dims >= 0 && grid_dims <= 2, "concat_grid: grid_dims %d", grid_dims);
  const long tot = (long)B * M * N * (Cin + grid_dims);
  hipLaunchKernelGGL(k_concat_grid, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, as_stream(stream), x, out, B, Cin, M, N, grid_dims, lo, hi, channels_last, gridx, gridy);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

int rpde_transpose_cs(const float* in, float* out, int B, int64_t S, int C, int to_channels_first, void* stream) {
  RPDE_CHECK_ARG(in && out && B > 0 && S > 0 && C > 0, "transpose_cs: bad arguments");
  const long R = to_channels_first ? S : C, Cc = to_channels_first ? C : S;
  dim3 grid((unsigned)((Cc + 31) / 32), (unsigned)((R + 31) / 32), B);
  RPDE_CHECK_ARG(grid.y <= 65535 && B <= 65535, "transpose_cs: tensor too large for one launch");
  hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, as_stream(stream), in, out, R, Cc);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

int rpde_act_fwd(const float* x, float* out, int64_t n, int act, void* stream) {
  RPDE_CHECK_ARG(x && out && n >= 0, "act_fwd: bad arguments");
  if (n == 0) return RPDE_OK;
  hipLaunchKernelGGL(k_act_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, out, (long)n, act);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

int rpde_act_bwd(const float* x, const float* g, float* dx, int64_t n, int act, void* stream) {
  RPDE_CHECK_ARG(x && g && dx && n >= 0, "act_bwd: bad arguments");
  if (n == 0) return RPDE_OK;
  hipLaunchKernelGGL(k_act_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, g, dx, (long)n, act);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // extern "C"
