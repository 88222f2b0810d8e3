// Read bandwidth by working-set size: is a re-read that fits the 256 MB Infinity Cache faster than HBM?
// (decides whether chunking the spectral pipeline sample-chunk by sample-chunk can beat the 6.3 TB/s stream rate)
//   hipcc --offload-arch=gfx950 -O3 -o profiles/ubench/mallbw profiles/ubench/mallbw.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ p, size_t n4, float* __restrict__ sink) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
    a.x += v0.x + v1.x + v2.x + v3.x; a.y += v0.y + v1.y + v2.y + v3.y;
    a.z += v0.z + v1.z + v2.z + v3.z; a.w += v0.w + v1.w + v2.w + v3.w;
  }
  for (; i < n4; i += stride) { const float4 v = p[i]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
  if (a.x + a.y + a.z + a.w == 1.2345e30f) sink[0] = a.x;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ p, size_t n4, float v) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p[i] = make_float4(v, v, v, v);
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ s, float4* __restrict__ d, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) d[i] = s[i];
}

int main() {
  const size_t maxb = (size_t)4 << 30;
  float4 *buf, *buf2; float* sink;
  hipMalloc(&buf, maxb); hipMalloc(&buf2, maxb); hipMalloc(&sink, 64);
  hipMemset(buf, 0, maxb); hipMemset(buf2, 0, maxb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * 8;
  const size_t sizes[] = {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024, 2048, 4096};
  printf("# working set MB | re-read GB/s (same buffer, back to back) | write GB/s | write-then-read: read GB/s | copy GB/s (r+w)\n");
  for (size_t s : sizes) {
    const size_t bytes = s << 20, n4 = bytes / 16;
    const int reps = (int)(s <= 256 ? 40 : (s <= 1024 ? 12 : 5));
    float ms;
    k_read<<<grid, 256>>>(buf, n4, sink);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k_read<<<grid, 256>>>(buf, n4, sink);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double rd = bytes * (double)reps / ms / 1e6;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k_write<<<grid, 256>>>(buf, n4, (float)r);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double wr = bytes * (double)reps / ms / 1e6;
    // write then read the same bytes: time only the reads (events around each read)
    double t_rd = 0;
    for (int r = 0; r < reps; ++r) {
      k_write<<<grid, 256>>>(buf, n4, (float)r);
      hipEventRecord(e0);
      k_read<<<grid, 256>>>(buf, n4, sink);
      hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      t_rd += ms;
    }
    const double wrd = bytes * (double)reps / t_rd / 1e6;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k_copy<<<grid, 256>>>(buf, buf2, n4);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    const double cp = 2.0 * bytes * reps / ms / 1e6;
    printf("%6zu  %9.0f  %9.0f  %9.0f  %9.0f\n", s, rd, wr, wrd, cp);
    fflush(stdout);
  }
  return 0;
}
