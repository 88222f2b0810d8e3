// What do s_memtime and s_memrealtime count on this part?  One long single-wave kernel, timed by HIP events.
// Build: hipcc --offload-arch=gfx950 -O3 clocks.hip -o clocks
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long* out, int iters, int heavy) {
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
  if (v == 1234.5f) out[2] = 1;
}
int main() {
  unsigned long long *d, h[3];
  (void)hipMalloc(&d, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int blocks : {1, 2048}) {
    spin<<<blocks, 256>>>(d, 1000, 0);
    (void)hipEventRecord(e0);
    spin<<<blocks, 256>>>(d, 4000000, 0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("blocks %4d: kernel %.3f ms  s_memtime %llu ticks (%.1f MHz)  s_memrealtime %llu ticks (%.1f MHz)\n", blocks, ms,
           h[0], h[0] / (ms * 1e3), h[1], h[1] / (ms * 1e3));
  }
  return 0;
}
