// Closed-form BRDFs, forward and backward, one thread per shading point (per ray, or per sample with
// MultiBRDF).  Replaces BRDF/basic_func.py:5-44, BRDF/RPV.py:6-63, BRDF/Hapke.py:6-200 and
// BRDF/microfacet.py:20-118 of the reference (and eval_* wrappers, models/spsbrdfnerf.py:9-30).
//
// The math is written once over a scalar type S.  S = float gives the forward kernels; S = Dual<N>
// (value + N partial derivatives, forward-mode) gives the exact Jacobian from which the backward kernels
// form J^T d_brdf.  Derivative conventions follow torch autograd: clamp passes gradient on [min, max]
// inclusive, where() passes it to the selected branch only, nan_to_num blocks it where it replaced.
// NaN replacement values are the reference's check_nan(val_rep=...) ones.
#include "common.h"
// no FMA contraction: dot products and angle differences must round like the reference's separate ATen ops,
// otherwise degenerate geometry (v == n: 0/0 -> NaN -> replacement value) takes a different branch.
#pragma clang fp contract(off)

#include "brdf_eval.h"

// ------------------------------------------------------------------ kernels
__device__ __forceinline__ V3<float> ld3(const float *p, int64_t i) {
  V3<float> r = {p[i * 3], p[i * 3 + 1], p[i * 3 + 2]};
  return r;
}
__global__ void rpv_fwd_kernel(const float *l, const float *v, const float *n, const float *w, const float *k, const float *th,
                               const float *rc, int64_t N, float *brdf, float *aux) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float ww[3], kk[3], tt[3], rr[3], out[3], ax[BN_BRDF_AUX] = {0};
  for (int c = 0; c < 3; ++c) {
    ww[c] = w[i * 3 + c];
    kk[c] = k ? k[i * 3 + c] : 0.f; tt[c] = th ? th[i * 3 + c] : 0.f; rr[c] = rc ? rc[i * 3 + c] : 0.f;
  }
  rpv_eval<float>(ld3(l, i), ld3(v, i), ld3(n, i), ww, k ? kk : nullptr, th ? tt : nullptr, rc ? rr : nullptr, out, ax);
  for (int c = 0; c < 3; ++c) brdf[i * 3 + c] = out[c];
  if (aux) for (int c = 0; c < BN_BRDF_AUX; ++c) aux[i * BN_BRDF_AUX + c] = ax[c];
}

__global__ void rpv_bwd_kernel(const float *l, const float *v, const float *n, const float *w, const float *k, const float *th,
                               const float *rc, const float *d_brdf, int64_t N, float *d_n, float *d_w, float *d_k,
                               float *d_th, float *d_rc) {
  typedef Dual<15> D;  // slots: n 0-2, w 3-5, k 6-8, theta 9-11, rhoc 12-14
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  V3<D> nn = {seed<15>(n[i * 3], 0), seed<15>(n[i * 3 + 1], 1), seed<15>(n[i * 3 + 2], 2)};
  D ww[3], kk[3], tt[3], rr[3], out[3];
  for (int c = 0; c < 3; ++c) {
    ww[c] = seed<15>(w[i * 3 + c], 3 + c);
    kk[c] = seed<15>(k ? k[i * 3 + c] : 0.f, 6 + c);
    tt[c] = seed<15>(th ? th[i * 3 + c] : 0.f, 9 + c);
    rr[c] = seed<15>(rc ? rc[i * 3 + c] : 0.f, 12 + c);
  }
  rpv_eval<D>(cst3<15>(ld3(l, i)), cst3<15>(ld3(v, i)), nn, ww, k ? kk : nullptr, th ? tt : nullptr, rc ? rr : nullptr, out,
              nullptr);
  const float db[3] = {d_brdf[i * 3], d_brdf[i * 3 + 1], d_brdf[i * 3 + 2]};
  for (int c = 0; c < 3; ++c) {
    if (d_n) d_n[i * 3 + c] = jt(out, db, c);
    if (d_w) d_w[i * 3 + c] = jt(out, db, 3 + c);
    if (d_k) d_k[i * 3 + c] = jt(out, db, 6 + c);
    if (d_th) d_th[i * 3 + c] = jt(out, db, 9 + c);
    if (d_rc) d_rc[i * 3 + c] = jt(out, db, 12 + c);
  }
}

__global__ void hapke_fwd_kernel(const float *l, const float *v, const float *n, const float *w, const float *b, const float *c,
                                 const float *theta, float hpk_scl, int shell, int64_t N, float *brdf, float *aux) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float ww[3], bb[3], cc[3], out[3], ax[BN_BRDF_AUX] = {0};
  for (int q = 0; q < 3; ++q) { ww[q] = w[i * 3 + q]; bb[q] = b ? b[i * 3 + q] : 0.f; cc[q] = c ? c[i * 3 + q] : 0.f; }
  float th = theta ? theta[i] : 0.f;
  hapke_eval<float>(ld3(l, i), ld3(v, i), ld3(n, i), ww, b ? bb : nullptr, c ? cc : nullptr, theta ? &th : nullptr, hpk_scl,
                    shell, out, ax);
  for (int q = 0; q < 3; ++q) brdf[i * 3 + q] = out[q];
  if (aux) for (int q = 0; q < BN_BRDF_AUX; ++q) aux[i * BN_BRDF_AUX + q] = ax[q];
}

__global__ void hapke_bwd_kernel(const float *l, const float *v, const float *n, const float *w, const float *b, const float *c,
                                 const float *theta, float hpk_scl, int shell, const float *d_brdf, int64_t N, float *d_n,
                                 float *d_w, float *d_b, float *d_c, float *d_theta) {
  typedef Dual<13> D;  // slots: n 0-2, w 3-5, b 6-8, c 9-11, theta 12
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  V3<D> nn = {seed<13>(n[i * 3], 0), seed<13>(n[i * 3 + 1], 1), seed<13>(n[i * 3 + 2], 2)};
  D ww[3], bb[3], cc[3], out[3];
  for (int q = 0; q < 3; ++q) {
    ww[q] = seed<13>(w[i * 3 + q], 3 + q);
    bb[q] = seed<13>(b ? b[i * 3 + q] : 0.f, 6 + q);
    cc[q] = seed<13>(c ? c[i * 3 + q] : 0.f, 9 + q);
  }
  D th = seed<13>(theta ? theta[i] : 0.f, 12);
  hapke_eval<D>(cst3<13>(ld3(l, i)), cst3<13>(ld3(v, i)), nn, ww, b ? bb : nullptr, c ? cc : nullptr, theta ? &th : nullptr,
                hpk_scl, shell, out, nullptr);
  const float db[3] = {d_brdf[i * 3], d_brdf[i * 3 + 1], d_brdf[i * 3 + 2]};
  for (int q = 0; q < 3; ++q) {
    if (d_n) d_n[i * 3 + q] = jt(out, db, q);
    if (d_w) d_w[i * 3 + q] = jt(out, db, 3 + q);
    if (d_b) d_b[i * 3 + q] = jt(out, db, 6 + q);
    if (d_c) d_c[i * 3 + q] = jt(out, db, 9 + q);
  }
  if (d_theta) d_theta[i] = jt(out, db, 12);
}

__global__ void microfacet_fwd_kernel(const float *l, const float *v, const float *n, const float *albedo, const float *rough,
                                      float f0, int64_t N, float *brdf, float *aux) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float al[3] = {albedo[i * 3], albedo[i * 3 + 1], albedo[i * 3 + 2]}, out[3], ax[BN_BRDF_AUX] = {0};
  microfacet_eval<float>(ld3(l, i), ld3(v, i), ld3(n, i), al, rough[i], f0, out, ax);
  for (int q = 0; q < 3; ++q) brdf[i * 3 + q] = out[q];
  if (aux) for (int q = 0; q < BN_BRDF_AUX; ++q) aux[i * BN_BRDF_AUX + q] = ax[q];
}

__global__ void microfacet_bwd_kernel(const float *l, const float *v, const float *n, const float *albedo, const float *rough,
                                      float f0, const float *d_brdf, int64_t N, float *d_n, float *d_albedo, float *d_rough) {
  typedef Dual<7> D;  // slots: n 0-2, albedo 3-5, rough 6
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  V3<D> nn = {seed<7>(n[i * 3], 0), seed<7>(n[i * 3 + 1], 1), seed<7>(n[i * 3 + 2], 2)};
  D al[3] = {seed<7>(albedo[i * 3], 3), seed<7>(albedo[i * 3 + 1], 4), seed<7>(albedo[i * 3 + 2], 5)}, out[3];
  const D rg = seed<7>(rough[i], 6);
  microfacet_eval<D>(cst3<7>(ld3(l, i)), cst3<7>(ld3(v, i)), nn, al, rg, f0, out, nullptr);
  const float db[3] = {d_brdf[i * 3], d_brdf[i * 3 + 1], d_brdf[i * 3 + 2]};
  for (int q = 0; q < 3; ++q) {
    if (d_n) d_n[i * 3 + q] = jt(out, db, q);
    if (d_albedo) d_albedo[i * 3 + q] = jt(out, db, 3 + q);
  }
  if (d_rough) d_rough[i] = jt(out, db, 6);
}

#define GRID1D(N) dim3((unsigned)ceil_div64(N, 128)), 128, 0, (hipStream_t)stream

extern "C" int bn_brdf_rpv_forward(const float *l, const float *v, const float *n, const float *w, const float *k,
                                   const float *theta, const float *rhoc, int64_t N, float *brdf, float *aux, void *stream) {
  BN_REQUIRE(l && v && n && w && brdf && N > 0, "brdf_rpv_forward: null argument");
  rpv_fwd_kernel<<<GRID1D(N)>>>(l, v, n, w, k, theta, rhoc, N, brdf, aux);
  BN_LAUNCH_CHECK("brdf_rpv_forward");
  return 0;
}
extern "C" int bn_brdf_rpv_backward(const float *l, const float *v, const float *n, const float *w, const float *k,
                                    const float *theta, const float *rhoc, const float *d_brdf, int64_t N, float *d_n,
                                    float *d_w, float *d_k, float *d_theta, float *d_rhoc, void *stream) {
  BN_REQUIRE(l && v && n && w && d_brdf && N > 0, "brdf_rpv_backward: null argument");
  rpv_bwd_kernel<<<GRID1D(N)>>>(l, v, n, w, k, theta, rhoc, d_brdf, N, d_n, d_w, d_k, d_theta, d_rhoc);
  BN_LAUNCH_CHECK("brdf_rpv_backward");
  return 0;
}
extern "C" int bn_brdf_hapke_forward(const float *l, const float *v, const float *n, const float *w, const float *b,
                                     const float *c, const float *theta, float hpk_scl, int32_t shell, int64_t N, float *brdf,
                                     float *aux, void *stream) {
  BN_REQUIRE(l && v && n && w && brdf && N > 0, "brdf_hapke_forward: null argument");
  BN_REQUIRE(b || (shell >= 1 && shell <= 3), "brdf_hapke_forward: b == NULL needs shell_hapke in {1,2,3}");
  hapke_fwd_kernel<<<GRID1D(N)>>>(l, v, n, w, b, c, theta, hpk_scl, shell, N, brdf, aux);
  BN_LAUNCH_CHECK("brdf_hapke_forward");
  return 0;
}
extern "C" int bn_brdf_hapke_backward(const float *l, const float *v, const float *n, const float *w, const float *b,
                                      const float *c, const float *theta, float hpk_scl, int32_t shell, const float *d_brdf,
                                      int64_t N, float *d_n, float *d_w, float *d_b, float *d_c, float *d_theta, void *stream) {
  BN_REQUIRE(l && v && n && w && d_brdf && N > 0, "brdf_hapke_backward: null argument");
  BN_REQUIRE(b || (shell >= 1 && shell <= 3), "brdf_hapke_backward: b == NULL needs shell_hapke in {1,2,3}");
  hapke_bwd_kernel<<<GRID1D(N)>>>(l, v, n, w, b, c, theta, hpk_scl, shell, d_brdf, N, d_n, d_w, d_b, d_c, d_theta);
  BN_LAUNCH_CHECK("brdf_hapke_backward");
  return 0;
}
extern "C" int bn_brdf_microfacet_forward(const float *l, const float *v, const float *n, const float *albedo, const float *rough,
                                          float f0, int64_t N, float *brdf, float *aux, void *stream) {
  BN_REQUIRE(l && v && n && albedo && rough && brdf && N > 0, "brdf_microfacet_forward: null argument");
  microfacet_fwd_kernel<<<GRID1D(N)>>>(l, v, n, albedo, rough, f0, N, brdf, aux);
  BN_LAUNCH_CHECK("brdf_microfacet_forward");
  return 0;
}
extern "C" int bn_brdf_microfacet_backward(const float *l, const float *v, const float *n, const float *albedo,
                                           const float *rough, float f0, const float *d_brdf, int64_t N, float *d_n,
                                           float *d_albedo, float *d_rough, void *stream) {
  BN_REQUIRE(l && v && n && albedo && rough && d_brdf && N > 0, "brdf_microfacet_backward: null argument");
  microfacet_bwd_kernel<<<GRID1D(N)>>>(l, v, n, albedo, rough, f0, d_brdf, N, d_n, d_albedo, d_rough);
  BN_LAUNCH_CHECK("brdf_microfacet_backward");
  return 0;
}
