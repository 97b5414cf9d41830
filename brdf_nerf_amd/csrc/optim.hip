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

// ------------------------------------------------------------------------------------------ all groups, one launch
// The fused step's optimiser: every parameter group of the flat buffer in one launch, hyper-parameters that change from
// step to step read from the device-resident step state (include/brdfnerf_hip.h, bn_step_state) so that a captured HIP
// graph replays unchanged.  The last workgroup to finish advances the state: all other workgroups have read it by then.
struct AdamMultiArgs {
  float *p, *g, *m, *v;
  int n_groups;
  int64_t lo[BN_ADAM_MAX_GROUPS], hi[BN_ADAM_MAX_GROUPS];
  int active[BN_ADAM_MAX_GROUPS];
  float b1, b2, eps, wd, gscale;
  int zero_grad;
  char *state;
  const unsigned int *fault[2];   // the trunks' sticky fault words (nullable): set -> this step's gradient is invalid, no update
};
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamMultiArgs A) {
  float *lr_p = (float *)(A.state + 16);
  unsigned int *done = (unsigned int *)(A.state + 20);
  int *steps = (int *)(A.state + 24);
  double *pw1 = (double *)(A.state + BN_STATE_POW_OFF), *pw2 = pw1 + BN_ADAM_MAX_GROUPS;     // beta^step per group, kept in double
  const float lr = *lr_p;
  // a lost LDS hand-over in a trunk kernel of this (or an earlier) step: the gradient is invalid - parameters and moments stay as
  // they are (the host learns of it from the mirrored words / bn_device_faults; inside a replayed graph nothing else would stop
  // the training from walking on: ADVICE r4)
  const bool faulted = (A.fault[0] && *A.fault[0] != 0u) || (A.fault[1] && *A.fault[1] != 0u);
  for (int gi = 0; gi < A.n_groups; ++gi) {
    if (!A.active[gi]) continue;
    // bias corrections in double, like torch.optim.Adam (1 - beta2^step loses half its digits in fp32 early on); the powers are
    // running products in the step state (a double pow() per workgroup cost the launch 10 us)
    const float bc1 = (float)(1.0 - pw1[gi] * (double)A.b1), bc2_sqrt = sqrtf((float)(1.0 - pw2[gi] * (double)A.b2));
    const int64_t n = A.hi[gi];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = A.lo[gi] + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
      f32x4 pp = *(f32x4 *)(A.p + i), gg = *(const f32x4 *)(A.g + i), mm = *(f32x4 *)(A.m + i), vv = *(f32x4 *)(A.v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float gr = gg[e] * A.gscale + A.wd * pp[e];
        mm[e] = A.b1 * mm[e] + (1.f - A.b1) * gr;
        vv[e] = A.b2 * vv[e] + (1.f - A.b2) * gr * gr;
        const float denom = sqrtf(vv[e]) / bc2_sqrt + A.eps;
        pp[e] -= (lr / bc1) * (mm[e] / denom);
      }
      if (!faulted) { *(f32x4 *)(A.p + i) = pp; *(f32x4 *)(A.m + i) = mm; *(f32x4 *)(A.v + i) = vv; }
      if (A.zero_grad) *(f32x4 *)(A.g + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) s_last = atomicAdd(done, 1u) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (s_last && threadIdx.x < 64) {       // the last workgroup to finish: every other one has passed its reads of the state
    unsigned long long *rng_step = (unsigned long long *)(A.state + 8);
    const unsigned long long cur = *rng_step;
    float *part = (float *)(A.state + BN_STATE_PART_OFF);
    float sum = part[threadIdx.x];          // BN_STATE_LOSS_SLOTS == 64 partial sums of this step's loss (bn_lambert_tail)
    part[threadIdx.x] = 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (threadIdx.x == 0) {
      *done = 0u;
      for (int gi = 0; gi < A.n_groups; ++gi)
        if (A.active[gi]) { steps[gi] += 1; pw1[gi] *= (double)A.b1; pw2[gi] *= (double)A.b2; }
      ((float *)(A.state + BN_STATE_LOSS_OFF))[cur % BN_STATE_LOSS_SLOTS] = sum;
      *rng_step = cur + 1ull;
    }
  }
}

extern "C" int bn_adam_multi(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t n_groups, const int64_t *lo,
                             const int64_t *hi, const int32_t *active, float beta1, float beta2, float eps, float weight_decay,
                             float grad_scale, int32_t zero_grad, void *state, void *stream) {
  BN_REQUIRE(param && grad && exp_avg && exp_avg_sq && state && lo && hi && active && n_groups >= 1 && n_groups <= BN_ADAM_MAX_GROUPS,
             "adam_multi: bad arguments");
  BN_REQUIRE(((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)state) % 16 == 0,
             "adam_multi: buffers must be 16-byte aligned");
  AdamMultiArgs a;
  a.p = param; a.g = grad; a.m = exp_avg; a.v = exp_avg_sq; a.n_groups = n_groups;
  int64_t most = 0;
  for (int i = 0; i < BN_ADAM_MAX_GROUPS; ++i) { a.lo[i] = a.hi[i] = 0; a.active[i] = 0; }
  for (int i = 0; i < n_groups; ++i) {
    BN_REQUIRE(lo[i] >= 0 && hi[i] >= lo[i] && lo[i] % 4 == 0 && hi[i] % 4 == 0, "adam_multi: group %d is not a multiple-of-4 range", i);
    a.lo[i] = lo[i]; a.hi[i] = hi[i]; a.active[i] = active[i];
    if (active[i] && hi[i] - lo[i] > most) most = hi[i] - lo[i];
  }
  a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay; a.gscale = grad_scale; a.zero_grad = zero_grad; a.state = (char *)state;
  a.fault[0] = bn_fwd_fault_ptr(); a.fault[1] = bn_bwd_fault_ptr();
  // (every workgroup ends on one atomic to the same word, ~12 ns each back to back: 512 of them, not 2048)
  int64_t blocks = ceil_div64(ceil_div64(most, 4), 256);
  blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
  BnProfScope prof_(BN_K_ADAM, (hipStream_t)stream);
  adam_multi_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(a);
  BN_LAUNCH_CHECK("adam_multi");
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
