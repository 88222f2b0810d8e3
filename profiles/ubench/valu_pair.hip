// Two questions behind the FeedForward kernels' per-tile budget:
//  (1) do the vector instructions of the TWO waves of a SIMD issue side by side (2 cycles per wave64 instruction for the
//      pair) or does the pair share one 4-cycle issue slot?   -> 4 vs 8 waves per workgroup running the same fma stream
//  (2) what does a wave pay for k independent vector instructions behind each v_mfma_f32_16x16x32_f16 of a dependent
//      chain, alone on its SIMD and with a partner running the same stream?   -> cycles per (MFMA + k VALU) group
//   hipcc --offload-arch=gfx950 -O3 -o profiles/ubench/valu_pair profiles/ubench/valu_pair.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int K>      // K independent v_fma_f32 behind each MFMA of one dependent accumulator chain; K < 0: no MFMA, -K fmas
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * i); b[i] = (_Float16)0.5f; }
  f4 c = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (K >= 0) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < (K >= 0 ? K : -K); ++q)
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(1.0001f), "v"(0.5f));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = c[0] + c[1] + c[2] + c[3];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
  if (blockIdx.x == 7 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K> void run(int waves, float* d, unsigned long long* c) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<K>, dim3(256), dim3(64 * waves), 0, 0, d, 10, c);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<K>, dim3(256), dim3(64 * waves), 0, 0, d, iters, c);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  const double groups = 16.0 * iters;
  printf("waves/SIMD %d  %s %2d fma per group: %8.3f ms  %7.1f shader cycles per group (wave 0 of block 7)\n", waves / 4,
         K >= 0 ? "MFMA +" : "no MFMA,", K >= 0 ? K : -K, ms, (double)h / groups);
}

int main() {
  float* d; unsigned long long* c;
  hipMalloc(&d, 4096); hipMalloc(&c, 64);
  for (int waves : {4, 8}) {
    run<-4>(waves, d, c); run<-8>(waves, d, c);
    run<0>(waves, d, c); run<1>(waves, d, c); run<2>(waves, d, c); run<3>(waves, d, c); run<4>(waves, d, c);
    run<6>(waves, d, c); run<8>(waves, d, c); run<12>(waves, d, c);
  }
  return 0;
}
