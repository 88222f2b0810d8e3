// Does a dependent accumulator chain that mixes v_mfma_f32_16x16x16_f16 and v_mfma_f32_16x16x32_f16 return the right
// sums on gfx950?  (h2.h keeps to the 32-deep form because the round-2 synthesis kernel read wrong accumulators with a
// 16-deep tail; this is the minimal form of that chain, each order, with and without padding between the two.)
//   hipcc --offload-arch=gfx950 -O3 -o profiles/ubench/mfma_mix profiles/ubench/mfma_mix.hip && profiles/ubench/mfma_mix
//   /opt/rocm/lib/llvm/bin/llvm-objdump -d --offloading ... (the disassembly of k<..> is in profiles/r04_mfma_mix.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// C[16][16] = A16[16][16] B16[16][16] + A32[16][32] B32[32][16];  ORDER 0: 16-deep first, 1: 32-deep first; NOPS: s_nop between
template <int ORDER, int NOPS>
__global__ void k(const _Float16* A16, const _Float16* B16, const _Float16* A32, const _Float16* B32, float* C, int reps) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  h4 a16, b16; h8 a32, b32;
  for (int j = 0; j < 4; ++j) { a16[j] = A16[r * 16 + 4 * g + j]; b16[j] = B16[(4 * g + j) * 16 + r]; }
  for (int j = 0; j < 8; ++j) { a32[j] = A32[r * 32 + 8 * g + j]; b32[j] = B32[(8 * g + j) * 16 + r]; }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < reps; ++i) {            // reps > 1: the chain repeated on the same accumulator (result = reps x C)
    if (ORDER == 0) {
      c = __builtin_amdgcn_mfma_f32_16x16x16f16(a16, b16, c, 0, 0, 0);
      if (NOPS) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a32, b32, c, 0, 0, 0);
    } else {
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a32, b32, c, 0, 0, 0);
      if (NOPS) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      c = __builtin_amdgcn_mfma_f32_16x16x16f16(a16, b16, c, 0, 0, 0);
    }
  }
  for (int j = 0; j < 4; ++j) C[(4 * g + j) * 16 + r] = c[j];
}

int main() {
  _Float16 hA16[256], hB16[256], hA32[512], hB32[512];
  srand(1);
  auto rnd = [] { return (_Float16)((rand() % 2001 - 1000) / 500.0f); };
  for (auto& v : hA16) v = rnd(); for (auto& v : hB16) v = rnd(); for (auto& v : hA32) v = rnd(); for (auto& v : hB32) v = rnd();
  _Float16 *A16, *B16, *A32, *B32; float* C;
  hipMalloc(&A16, 512); hipMalloc(&B16, 512); hipMalloc(&A32, 1024); hipMalloc(&B32, 1024); hipMalloc(&C, 1024);
  hipMemcpy(A16, hA16, 512, hipMemcpyHostToDevice); hipMemcpy(B16, hB16, 512, hipMemcpyHostToDevice);
  hipMemcpy(A32, hA32, 1024, hipMemcpyHostToDevice); hipMemcpy(B32, hB32, 1024, hipMemcpyHostToDevice);
  for (int reps : {1, 3}) {
    double ref[256];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double s = 0;
      for (int kk = 0; kk < 16; ++kk) s += (double)hA16[i * 16 + kk] * (double)hB16[kk * 16 + j];
      for (int kk = 0; kk < 32; ++kk) s += (double)hA32[i * 32 + kk] * (double)hB32[kk * 16 + j];
      ref[i * 16 + j] = reps * s;
    }
    auto run = [&](const char* name, auto kern) {
      float hC[256];
      hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, A16, B16, A32, B32, C, reps);
      hipMemcpy(hC, C, 1024, hipMemcpyDeviceToHost);
      double e = 0; int bad = 0;
      for (int i = 0; i < 256; ++i) { const double d = fabs(hC[i] - ref[i]); e = fmax(e, d); bad += d > 1e-3 * (1 + fabs(ref[i])); }
      printf("reps %d  %-28s max |err| %.3e  wrong elements %d / 256\n", reps, name, e, bad);
    };
    run("16-deep then 32-deep", k<0, 0>); run("16-deep, s_nop, 32-deep", k<0, 1>);
    run("32-deep then 16-deep", k<1, 0>); run("32-deep, s_nop, 16-deep", k<1, 1>);
  }
  return 0;
}
