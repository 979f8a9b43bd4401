// torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay) on ONE flat fp32 buffer
// (train_model_B_gradFTM.py:453,121): a single launch over all 282,705 parameters instead of 53
// per-tensor updates.  Same arithmetic order as torch's single-tensor path:
//   m = lerp(m, g, 1-b1); v = b2*v + (1-b2) g*g; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// grad_scale multiplies the gradient first (1/world_size after a sum all-reduce).
#include "edge_conv.h"

namespace {
__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, int n, float step_size, float beta1, float beta2,
                                 float bc2_sqrt, float eps, float weight_decay, float grad_scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i] * grad_scale;
  const float pi = p[i];
  if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
  const float mi = m[i] + (1.f - beta1) * (gi - m[i]);      // lerp
  const float vi = fmaf(1.f - beta2, gi * gi, beta2 * v[i]);
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = pi - step_size * (mi / denom);
}
// hipGraph-capturable form: the step count lives in device memory, so a captured step can be replayed.
//   prep : t = ++(*step);  coef[0] = lr / (1 - b1^t);  coef[1] = sqrt(1 - b2^t)     (float64, as on the host)
__global__ void adam_prep_kernel(long long* step, float* coef, float lr, float beta1, float beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const long long t = *step + 1;
    *step = t;
    coef[0] = (float)((double)lr / (1.0 - pow((double)beta1, (double)t)));
    coef[1] = (float)sqrt(1.0 - pow((double)beta2, (double)t));
  }
}
__global__ void adam_flat_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                     float* __restrict__ v, int n, const float* __restrict__ coef, float beta1, float beta2,
                                     float eps, float weight_decay, float grad_scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float step_size = coef[0], bc2_sqrt = coef[1];
  float gi = g[i] * grad_scale;
  const float pi = p[i];
  if (weight_decay != 0.f) gi = fmaf(weight_decay, pi, gi);
  const float mi = m[i] + (1.f - beta1) * (gi - m[i]);
  const float vi = fmaf(1.f - beta2, gi * gi, beta2 * v[i]);
  m[i] = mi;
  v[i] = vi;
  p[i] = pi - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}
}  // namespace

int launch_adam_flat_dev(float* p, const float* g, float* m, float* v, int n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, long long* step_dev, float* coef2, float grad_scale, hipStream_t s) {
  if (n < 1 || !step_dev || !coef2) return SIFSR_ERR_ARG;
  hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, s, step_dev, coef2, lr, beta1, beta2);
  hipLaunchKernelGGL(adam_flat_dev_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, g, m, v, n, coef2, beta1, beta2, eps,
                     weight_decay, grad_scale);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}

int launch_adam_flat(float* p, const float* g, float* m, float* v, int n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, float grad_scale, hipStream_t s) {
  if (step < 1 || n < 1) return SIFSR_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_flat_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, g, m, v, n, (float)((double)lr / bc1),
                     beta1, beta2, (float)sqrt(bc2), eps, weight_decay, grad_scale);
  SIFSR_LAUNCH_CHECK();
  return SIFSR_OK;
}
