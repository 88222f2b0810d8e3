// AdamW over flat fp32 buffers: the optimizer step of the training loop (reference: torch.optim.AdamW built in
// main_1d.py:144 / main_2d.py:173) as ONE streaming kernel over parameters, gradients and both moments, instead
// of the eight elementwise passes of the multi-tensor implementation.  Same update rule and operation order:
//     p *= 1 - lr * wd;  m += (g - m)(1 - b1);  v = v b2 + (1 - b2) g g;  p -= step_size * m / (sqrt(v) / sqrt(bc2) + eps)
// Complex parameters are their interleaved (re, im) floats, as torch.view_as_real treats them.
#include "rpde_internal.h"

namespace rpde {

struct AdamwScalars { float omlw, omb1, b2, omb2, step_size, bc2_sqrt, eps; };

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamwScalars& s) {
  p *= s.omlw;
  m = fmaf(g - m, s.omb1, m);
  v = fmaf(s.omb2 * g, g, v * s.b2);
  const float denom = sqrtf(v) / s.bc2_sqrt + s.eps;
  p = fmaf(-s.step_size, m / denom, p);
}

template <bool DEV>
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, long n4, AdamwScalars s, const float* __restrict__ dev) {
  if (DEV) { s.step_size = dev[1]; s.bc2_sqrt = dev[2]; s.omlw = dev[5]; }
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    adamw_one(pp.x, gg.x, mm.x, vv.x, s); adamw_one(pp.y, gg.y, mm.y, vv.y, s);
    adamw_one(pp.z, gg.z, mm.z, vv.z, s); adamw_one(pp.w, gg.w, mm.w, vv.w, s);
    reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
  }
}

// device-side step state (hipGraph-capturable steps): dev[0] = step (incremented here), dev[1] = lr / (1 - b1^t),
// dev[2] = sqrt(1 - b2^t), dev[3] = lr, dev[4] = weight decay, dev[5] = 1 - lr * wd.  lr and wd live on the device so
// that a captured step follows a learning-rate schedule: a replay repeats its launch ARGUMENTS, but reads these words
__global__ void k_adamw_tick(float* dev, float b1, float b2) {
  const double t = (double)dev[0] + 1.0, lr = (double)dev[3];
  dev[0] = (float)t;
  dev[1] = (float)(lr / (1.0 - pow((double)b1, t)));
  dev[2] = (float)sqrt(1.0 - pow((double)b2, t));
  dev[5] = (float)(1.0 - lr * (double)dev[4]);
}
__global__ void k_adamw_set_hyper(float* dev, float lr, float wd) { dev[3] = lr; dev[4] = wd; }

static bool capturing(hipStream_t st) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &status) != hipSuccess) { (void)hipGetLastError(); return false; }
  return status != hipStreamCaptureStatusNone;
}

static int adamw_grid(long n4) {
  long g = (n4 + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace rpde

using namespace rpde;

extern "C" {

int rpde_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float one_minus_lr_wd, float one_minus_b1,
                    float b2, float one_minus_b2, float step_size, float bc2_sqrt, float eps, void* stream) {
  RPDE_CHECK_ARG(p && g && m && v && n > 0 && n % 4 == 0, "adamw_step: null buffer or length %ld not a multiple of 4", (long)n);
  RPDE_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                   reinterpret_cast<uintptr_t>(v)) & 15) == 0, "adamw_step: buffers must be 16-byte aligned");
  const AdamwScalars s{one_minus_lr_wd, one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps};
  hipLaunchKernelGGL(k_adamw<false>, dim3(adamw_grid(n / 4)), dim3(256), 0, as_stream(stream), p, g, m, v, (long)(n / 4), s, nullptr);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

int rpde_adamw_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                        float weight_decay, float* step_dev, void* stream) {
  RPDE_CHECK_ARG(p && g && m && v && step_dev && n > 0 && n % 4 == 0, "adamw_step_dev: bad arguments");
  RPDE_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                   reinterpret_cast<uintptr_t>(v)) & 15) == 0, "adamw_step_dev: buffers must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  // an eager call runs with the lr / weight decay it is given; a call that is being CAPTURED leaves the device words
  // alone, so that every replay uses what rpde_adamw_set_hyper_dev (or the last eager call) put there
  if (!capturing(st)) {
    hipLaunchKernelGGL(k_adamw_set_hyper, dim3(1), dim3(1), 0, st, step_dev, lr, weight_decay);
    RPDE_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_adamw_tick, dim3(1), dim3(1), 0, st, step_dev, b1, b2);
  RPDE_LAUNCH_CHECK();
  const AdamwScalars s{1.f, 1.f - b1, b2, 1.f - b2, 0.f, 1.f, eps};
  hipLaunchKernelGGL(k_adamw<true>, dim3(adamw_grid(n / 4)), dim3(256), 0, st, p, g, m, v, (long)(n / 4), s, step_dev);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// the update of rpde_adamw_step_dev without advancing the counter: further buffers of the same optimizer step
int rpde_adamw_apply_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                         float weight_decay, const float* step_dev, void* stream) {
  RPDE_CHECK_ARG(p && g && m && v && step_dev && n > 0 && n % 4 == 0, "adamw_apply_dev: bad arguments");
  RPDE_CHECK_ARG(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                   reinterpret_cast<uintptr_t>(v)) & 15) == 0, "adamw_apply_dev: buffers must be 16-byte aligned");
  (void)lr; (void)weight_decay;           // (the step's values are on the device since rpde_adamw_step_dev)
  const AdamwScalars s{1.f, 1.f - b1, b2, 1.f - b2, 0.f, 1.f, eps};
  hipLaunchKernelGGL(k_adamw<true>, dim3(adamw_grid(n / 4)), dim3(256), 0, as_stream(stream), p, g, m, v, (long)(n / 4), s, step_dev);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

// learning rate / weight decay of the captured steps that follow (a scheduler's new values): one tiny launch, outside
// any capture
int rpde_adamw_set_hyper_dev(float* step_dev, float lr, float weight_decay, void* stream) {
  RPDE_CHECK_ARG(step_dev, "adamw_set_hyper_dev: null state");
  hipLaunchKernelGGL(k_adamw_set_hyper, dim3(1), dim3(1), 0, as_stream(stream), step_dev, lr, weight_decay);
  RPDE_LAUNCH_CHECK();
  return RPDE_OK;
}

}  // extern "C"
