// Fused Adam over one flat fp32 parameter buffer (torch.optim.Adam semantics, main.py:147-168:
// lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0 by default; L2-style weight decay when non-zero).
#include <math.h>
#include "common.h"

__global__ void adam_kernel(float *p, const float *g, float *m, float *v, int64_t n, float lr, float b1, float b2, float eps,
                            float wd, float bc1, float bc2_sqrt, float gscale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      f32x4 pp = *(f32x4 *)(p + i), gg = *(const f32x4 *)(g + i), mm = *(f32x4 *)(m + i), vv = *(f32x4 *)(v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gr = gg[e] * gscale + wd * pp[e];
        mm[e] = b1 * mm[e] + (1.f - b1) * gr;
        vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
        const float denom = sqrtf(vv[e]) / bc2_sqrt + eps;
        pp[e] -= (lr / bc1) * (mm[e] / denom);
      }
      *(f32x4 *)(p + i) = pp; *(f32x4 *)(m + i) = mm; *(f32x4 *)(v + i) = vv;
    } else {
      for (int64_t j = i; j < n; ++j) {
        float gr = g[j] * gscale + wd * p[j];
        m[j] = b1 * m[j] + (1.f - b1) * gr;
        v[j] = b2 * v[j] + (1.f - b2) * gr * gr;
        const float denom = sqrtf(v[j]) / bc2_sqrt + eps;
        p[j] -= (lr / bc1) * (m[j] / denom);
      }
    }
  }
}

extern "C" int bn_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                            void *stream) {
  BN_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad arguments");
  BN_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0,
             "adam_step: buffers must be 16-byte aligned");
  // bias corrections in double on the host, like torch.optim.Adam (1 - beta2^step loses half its digits in fp32 early on)
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  const int64_t blocks = ceil_div64(ceil_div64(n, 4), 256);
  BnProfScope prof_(BN_K_ADAM, (hipStream_t)stream);
  adam_kernel<<<dim3((unsigned)(blocks < 2048 ? blocks : 2048)), 256, 0, (hipStream_t)stream>>>(
      param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2), grad_scale);
  BN_LAUNCH_CHECK("adam_step");
  return 0;
}

// ------------------------------------------------------------------------------------------ non-finite counter (debug)
__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float *__restrict__ x, int64_t n, unsigned long long *counts) {
  unsigned nan = 0, inf = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    nan += v != v;
    inf += (v - v != 0.f) && (v == v);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { nan += __shfl_xor(nan, o); inf += __shfl_xor(inf, o); }
  if ((threadIdx.x & 63) == 0) {
    if (nan) atomicAdd(counts, (unsigned long long)nan);
    if (inf) atomicAdd(counts + 1, (unsigned long long)inf);
  }
}

extern "C" int bn_count_nonfinite(const float *x, int64_t n, unsigned long long *counts, void *stream) {
  BN_REQUIRE(x && counts && n > 0, "count_nonfinite: bad arguments");
  const int64_t blocks = ceil_div64(n, 256 * 8);
  count_nonfinite_kernel<<<dim3((unsigned)(blocks < 4096 ? blocks : 4096)), 256, 0, (hipStream_t)stream>>>(x, n, counts);
  BN_LAUNCH_CHECK("count_nonfinite");
  return 0;
}
