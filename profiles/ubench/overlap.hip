// Does VALU work of one wave overlap with bf16 / fp32 MFMA work of another wave on the same SIMD?
// Build: hipcc --offload-arch=gfx950 -O3 overlap.hip -o overlap ; run: ./overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int MODE>  // bit0: MFMA waves active, bit1: VALU waves active, bit2: fp32 mfma instead of bf16
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;  // 8 waves: waves 0-3 -> one per SIMD do MFMA, 4-7 do VALU
  f16v acc = {0};
  float v0 = threadIdx.x, v1 = 1.0f, v2 = 0.5f, v3 = 0.25f;
  if (wave < 4) {
    if (MODE & 1) {
      bf8 a, b;
      for (int i = 0; i < 8; i++) { a[i] = (__bf16)1.0f; b[i] = (__bf16)0.5f; }
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
          if (MODE & 4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, v2, acc, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
      }
    }
  } else {
    if (MODE & 2) {
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 32; j++) {  // 4 independent chains: 128 v_fma per iteration
          v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 0.9999f, 0.25f);
          v2 = __builtin_fmaf(v2, 1.0002f, 0.125f); v3 = __builtin_fmaf(v3, 0.9998f, 0.0625f);
        }
      }
    }
  }
  float s = v0 + v1 + v2 + v3;
  for (int i = 0; i < 16; i++) s += acc[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE> float run(float* d, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, 512>>>(d, 10);
  hipEventRecord(e0);
  k<MODE><<<256, 512>>>(d, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* d; hipMalloc(&d, 4096);
  const int it = 20000;
  printf("bf16 mfma only   %.3f ms (16 mfma x %d)\n", run<1>(d, it), it);
  printf("valu only        %.3f ms (128 fma x %d)\n", run<2>(d, it), it);
  printf("bf16 mfma + valu %.3f ms\n", run<3>(d, it));
  printf("fp32 mfma only   %.3f ms\n", run<5>(d, it));
  printf("fp32 mfma + valu %.3f ms\n", run<7>(d, it));
  return 0;
}
