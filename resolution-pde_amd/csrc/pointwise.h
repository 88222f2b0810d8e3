// Internal declarations of the pointwise / reduction helpers (pointwise.hip).
#pragma once
#include "rpde_internal.h"

namespace rpde {

int reduce_slabs(const float* slabs, float* out, long n, int S, long stride, float scale, int accumulate, hipStream_t st);
// out_k[j] = sum_s slabs[s*stride + off_k + j] for up to eight segments of one slab row, one launch; null dst: skipped
struct ReduceSegs { int n, nseg; int off[8]; int len[8]; float* dst[8]; };
int reduce_slabs_seg(const float* slabs, int S, long stride, const ReduceSegs& sg, hipStream_t st);
// folds collected over several kernels and done by ONE launch (the small configurations are launch-bound: a FeedForward
// backward on the GEMM path has eight of them).  Each job is out[i] = sum_s src[s*stride + i], i < len, summed exactly as
// reduce_slabs does it (same bits).
struct FoldJobs {
  static constexpr int MAX = 12;
  int n = 0;
  struct Job { const float* src; float* dst; long stride; int len, S, blk0; } j[MAX];
  bool add(const float* src, float* dst, int len, int S, long stride) {
    if (n >= MAX) return false;
    j[n].src = src; j[n].dst = dst; j[n].len = len; j[n].S = S; j[n].stride = stride; j[n].blk0 = 0;
    ++n;
    return true;
  }
};
int fold_jobs(FoldJobs& jobs, hipStream_t st);
constexpr int REDUCE_CHUNKS = 64;
// tmp: REDUCE_CHUNKS * n floats of scratch
int reduce_slabs_2pass(const float* slabs, float* out, long n, int S, long stride, float* tmp, hipStream_t st);

size_t colsum_ws_floats(long P, int N);
int colsum(const float* x, float* out, long P, int N, long ld, float* ws, int accumulate, hipStream_t st);

int ff_tail_fwd(const float* z, const float* res, float* out, long P, int C, int layer_norm, float eps,
                const float* gamma, const float* beta, DropCfg drop, int post_act, hipStream_t st);
size_t ff_tail_bwd_ws_floats(long P, int C);
int ff_tail_bwd(const float* z, const float* g, float* dz, long P, int C, int layer_norm, float eps, const float* gamma,
                const float* beta, DropCfg drop, int post_act, float* grad_gamma, float* grad_beta, float* grad_bias,
                int* bias_done, float* ws, hipStream_t st, FoldJobs* defer = nullptr);

int pack_mix_weights(const float* w, float* blk, int Ci, int Co, int K, int keff, hipStream_t st);
int unpack_mix_grad(const float* slabs, float* gw, int Ci, int Co, int K, int keff, int S, long sstride, hipStream_t st);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// bump allocator over the caller's workspace (256-byte aligned pieces)
struct Arena {
  char* base; size_t size; size_t used;
  Arena(void* p, size_t n) : base(static_cast<char*>(p)), size(n), used(0) {}
  float* take(size_t floats) {
    const size_t bytes = align_up(floats * sizeof(float), 256);
    if (!base || used + bytes > size) { used = size + 1; return nullptr; }
    float* r = reinterpret_cast<float*>(base + used);
    used += bytes;
    return r;
  }
  bool ok() const { return used <= size; }
};
inline size_t arena_bytes(size_t floats) { return align_up(floats * sizeof(float), 256); }

}  // namespace rpde
