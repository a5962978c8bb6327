// Fused AdamW over one flat parameter buffer (replaces torch.optim.AdamW as configured by
// train/main.py:13-14 and train/configs/spectralv2.yaml:99-111: lr 1e-4, betas (0.9, 0.999),
// eps 1e-8, weight_decay 1e-2, no amsgrad).  One launch updates all 76k parameters; the step
// counter lives on the device so the launch can be replayed from a HIP graph.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    const long long* __restrict__ step, long n, float lr, float b1, float b2,
                                                    float eps, float wd, float grad_scale) {
  const double t = (double)(*step);
  const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  const float decay = (float)(1.0 - (double)lr * (double)wd);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * grad_scale;
    float pi = p[i] * decay;
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);
    const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ void step_inc_kernel(long long* step) { *step += 1; }

}  // namespace

// Increments *step (device int64) and applies one AdamW update.  grad_scale folds the 1/world_size of
// a summed data-parallel gradient.
NOISE_EXPORT int noise_adamw_step(float* p, const float* g, float* m, float* v, long long* step, long n, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream) {
  NOISE_REQUIRE(p && g && m && v && step && n > 0, "noise_adamw_step: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, st, step);
  NOISE_CHECK_LAUNCH("noise_adamw_step/inc");
  hipLaunchKernelGGL(adamw_kernel, dim3(noise_grid(n, 256, 1024)), dim3(256), 0, st, p, g, m, v, step, n, lr, beta1, beta2,
                     eps, weight_decay, grad_scale);
  NOISE_CHECK_LAUNCH("noise_adamw_step");
  return 0;
}
