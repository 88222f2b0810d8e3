// How fast can 537 MB of fp32 [P][64] be written, by store pattern?  (the synthesis kernel's question)
//  A  1 KB contiguous per wave instruction (float4 per lane)
//  B  per instruction 16 points x 64 B (one quarter of each point's 256 B); the four waves of a workgroup write the four
//     quarters of the same points back to back (the round-2 kernels' pattern)
//  C  as B, but a workgroup writes ONE quarter of a 64 x 64 point tile; the other quarters come from other workgroups of
//     the same XCD group (blockIdx & 7), whenever they get there (the 64 x 64 x 16-channel tile kernel's pattern)
//  D  per instruction 8 points x 128 B (half of each point: a full 128-byte line per 8 lanes), quarters-pairs by other WGs
//   hipcc --offload-arch=gfx950 -O3 -o profiles/ubench/storepat profiles/ubench/storepat.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr long P = 32L * 256 * 256;      // points
__global__ __launch_bounds__(256) void k_a(float4* __restrict__ o) {
  const long n4 = P * 16;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) o[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ __launch_bounds__(256) void k_b(float* __restrict__ o) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (long t = blockIdx.x; t < P / 16; t += gridDim.x) {
    const long pt = t * 16 + (l >> 2);
    *reinterpret_cast<float4*>(o + pt * 64 + 16 * w + 4 * (l & 3)) = make_float4(1.f, 2.f, 3.f, (float)t);
  }
}
// tile = 64 x 64 points of a 256 x 256 sample; item = tile * 4 + cb; 8 waves, wave (mt, p): rows 16 mt + r, cols 16 (p + 2a) + 4g + pl
template <int SEG>   // 64 or 128 bytes per point per workgroup
__global__ __launch_bounds__(512) void k_c(float* __restrict__ o) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6, mt = w & 3, p = w >> 2, g = l >> 4, li = l & 15;
  const int xg = blockIdx.x & 7, jw = blockIdx.x >> 3, nj = gridDim.x >> 3;
  constexpr int NCB = 256 / SEG;                  // channel blocks per point
  const long items = 4L * 16 * NCB;               // per group: 4 samples x 16 tiles x NCB
  for (long q = jw; q < items; q += nj) {
    const int sb = (int)(q / (16 * NCB)), t = (int)(q % (16 * NCB));
    const int b = xg + 8 * sb, cb = t % NCB, tile = t / NCB;
    const int m0 = (tile / 4) * 64, n0 = (tile % 4) * 64;
    for (int r = 0; r < 16; ++r)
      for (int a = 0; a < 2; ++a) {
        if (SEG == 64) {
          const int qd = li >> 2, pl = li & 3;
          const long off = ((((long)b * 256 + m0 + 16 * mt + r) * 256) + n0 + 16 * (p + 2 * a) + 4 * g + pl) * 64 + 16 * cb + 4 * qd;
          *reinterpret_cast<float4*>(o + off) = make_float4(1.f, 2.f, 3.f, (float)q);
        } else {
          // 128 B per point: 8 lanes per point, 2 points per lane-row, 2 stores cover 4 points
          for (int h = 0; h < 2; ++h) {
            const int oct = li >> 3, c8 = li & 7;
            const long off = ((((long)b * 256 + m0 + 16 * mt + r) * 256) + n0 + 16 * (p + 2 * a) + 4 * g + 2 * h + oct) * 64 + 32 * cb + 4 * c8;
            *reinterpret_cast<float4*>(o + off) = make_float4(1.f, 2.f, 3.f, (float)q);
          }
        }
      }
  }
}

int main(int argc, char** argv) {
  const int cus = argc > 1 ? atoi(argv[1]) : 256;      // workgroups for C / D (one per CU); A / B use 8x as many small ones
  float* buf;
  hipMalloc(&buf, P * 256);
  hipMemset(buf, 0, P * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %8.1f us  %7.0f GB/s\n", name, ms * 100, P * 256.0 * 10 / ms / 1e6);
  };
  timeit("A  1 KB contiguous per instruction", [&] { k_a<<<8 * cus, 256>>>((float4*)buf); });
  timeit("B  16 x 64 B, quarters by the 4 waves of a workgroup", [&] { k_b<<<8 * cus, 256>>>(buf); });
  timeit("C  16 x 64 B, quarters by different workgroups (64x64 tiles)", [&] { k_c<64><<<cus, 512>>>(buf); });
  timeit("D  8 x 128 B, halves by different workgroups (64x64 tiles)", [&] { k_c<128><<<cus, 512>>>(buf); });
  return 0;
}
