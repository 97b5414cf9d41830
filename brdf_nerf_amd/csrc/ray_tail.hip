// Ray-level shading + losses of the fused training step, forward and backward in one launch, one thread per ray.
//
// The step composites the merged sample set into per-ray sums acc[c] = sum_s w_s out[s][c] (bn_merged_composite_forward);
// everything the loss reads after that is a function of those sums when every ray has ONE BRDF (MultiBRDF == 0) and no
// per-sample irradiance: the composited albedo, the normalised composited normal, the composited BRDF parameters, the BRDF
// itself, the clamp, SNerfLoss, DepthLoss and HardSurfaceLoss.  This kernel evaluates that chain with forward-mode duals
// seeded at the sums and returns d loss / d acc, d loss / d (sum w) and d loss / d depth for bn_merged_composite_backward.
// Replaces, for the training step: models/spsbrdfnerf.py:259-357 (ray-level shading), BRDF/*.py, metrics.py:39-61
// (SNerfLoss, lambda_sc = 0), metrics.py:82-161 (DepthLoss), metrics.py:263-290 (HardSurfaceLoss) and their autograd graphs.
#include "common.h"
#include "brdfnerf_hip.h"
#include "prof.h"
// (no FMA contraction, like brdf.hip: the degenerate-geometry branches must round like the reference's separate ATen ops)
#pragma clang fp contract(off)
#include "brdf_eval.h"

namespace {

struct ShadeArgs {
  bn_shade_desc d;
  const float *acc, *wsum, *depth, *var;
  const float *rays_d, *sun_d;           // sun_d nullptr: (1, 1, 1) (the non-satellite data sets, rendering.py:190)
  int64_t rd_stride, sd_stride;
  const float *rgbs, *valid, *tdepth, *tweight, *tstd;
  int64_t v_stride, td_stride, tw_stride, ts_stride;
  int64_t R;
  float *rgb, *ray_loss, *loss_acc;
  int loss_slots;
  float *d_acc, *d_wsum, *d_depth;
  const float *extra_loss;              // nullable: per-ray loss terms computed elsewhere (NormalRegLoss of the compositing kernel)
  unsigned long long *nonfinite;        // nullable: a ray whose loss term is not finite contributes nothing (loss 0, gradients 0) and is counted
};

// KIND: BN_SHADE_LAMBERT / RPV / HAPKE / MICROFACET.  Dual slots: composited normal 0-2, composited albedo 3-5, then the
// BRDF parameters (RPV: k 6-8, theta 9-11, rhoc 12-14; Hapke: b 6-8, c 9-11, theta 12; microfacet: roughness 6).
template <int KIND> struct Slots { static constexpr int N = KIND == BN_SHADE_RPV ? 15 : KIND == BN_SHADE_HAPKE ? 13 : 7; };

template <int KIND> __global__ __launch_bounds__(64) void ray_shade_loss_kernel(const ShadeArgs A) {
  constexpr int N = Slots<KIND>::N;
  typedef Dual<N> D;
  const int64_t ray = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (ray >= A.R) return;
  const bn_shade_desc &q = A.d;
  const int C = q.C;
  const float *acc = A.acc + ray * C;
  const float ws = A.wsum[ray], depth = A.depth[ray];
  const float pad = q.rgb_padding;
  const bool has_n = q.ch_normal >= 0;
  const float sun[3] = {A.sun_d ? A.sun_d[ray * A.sd_stride] : 1.f, A.sun_d ? A.sun_d[ray * A.sd_stride + 1] : 1.f,
                        A.sun_d ? A.sun_d[ray * A.sd_stride + 2] : 1.f};
  // upward normal: |sun_z| (spsbrdfnerf.py:260-264); else the sun pass's visibility of the ray's last sample (:354), else 1
  const float irr = (q.cos_irradiance && has_n) ? fabsf(sun[2]) : (q.irr ? q.irr[ray * q.irr_stride] : 1.f);
  // composited albedo sum_s w (albedo (1 + 2 pad) - pad)   (:270, :275)
  D w[3], out[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) w[c] = seed<N>(acc[c] * (1.f + 2.f * pad) - pad * ws, 3 + c);
  if (KIND == BN_SHADE_LAMBERT) {
#pragma unroll
    for (int c = 0; c < 3; ++c) out[c] = w[c];
  } else {
    const V3<float> lf = {sun[0], sun[1], sun[2]};
    const float *rd = A.rays_d + ray * A.rd_stride;
    const V3<float> vf = {-rd[0], -rd[1], -rd[2]};
    // l2_normalize (train_utils.py:28-33) of the composited normal, differentiated with the rest
    const float *an = acc + q.ch_normal;
    V3<D> nn = {seed<N>(an[0], 0), seed<N>(an[1], 1), seed<N>(an[2], 2)};
    const D nrm = sqrt_(clamp_min_(dot3(nn, nn), 1.1920928955078125e-07f));
    V3<D> ns = {nn.x / nrm, nn.y / nrm, nn.z / nrm};
    const V3<D> l = cst3<N>(lf), v = cst3<N>(vf);
    if (KIND == BN_SHADE_RPV) {
      D k[3], th[3], rc[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        k[c] = seed<N>(q.ch_p0 >= 0 ? acc[q.ch_p0 + c] : 0.f, 6 + c);
        th[c] = seed<N>(q.ch_p1 >= 0 ? acc[q.ch_p1 + c] : 0.f, 9 + c);
        rc[c] = q.rhoc_is_albedo ? w[c] : seed<N>(q.ch_p2 >= 0 ? acc[q.ch_p2 + c] : 0.f, 12 + c);   // funcH == 2 (:288-291)
      }
      rpv_eval<D>(l, v, ns, w, q.ch_p0 >= 0 ? k : nullptr, q.ch_p1 >= 0 ? th : nullptr,
                  (q.ch_p2 >= 0 || q.rhoc_is_albedo) ? rc : nullptr, out, nullptr);
    } else if (KIND == BN_SHADE_HAPKE) {
      D b[3], cc[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        b[c] = seed<N>(q.ch_p0 >= 0 ? acc[q.ch_p0 + c] : 0.f, 6 + c);
        cc[c] = seed<N>(q.ch_p1 >= 0 ? acc[q.ch_p1 + c] : 0.f, 9 + c);
      }
      const D th = seed<N>(q.ch_p2 >= 0 ? acc[q.ch_p2] : 0.f, 12);
      hapke_eval<D>(l, v, ns, w, q.ch_p0 >= 0 ? b : nullptr, q.ch_p1 >= 0 ? cc : nullptr, q.ch_p2 >= 0 ? &th : nullptr,
                    q.hpk_scl, q.shell, out, nullptr);
    } else {
      const D rg = seed<N>(acc[q.ch_p0], 6);
      microfacet_eval<D>(l, v, ns, w, rg, q.f0, out, nullptr);
    }
  }
  // rgb = clamp(irradiance * brdf, 0, 1); SNerfLoss = lambda_rgb * mean over (rays, 3) of (rgb - target)^2
  const float invn = 1.f / (3.f * (float)A.R);
  float loss = 0.f, db[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float x = irr * out[c].v;          // (irr == 1 without the cosine term: exact)
    const float y = isnan(x) ? x : fminf(fmaxf(x, 0.f), 1.f);
    const float e = y - A.rgbs[ray * 3 + c];
    loss += q.lambda_rgb * e * e * invn;
    const float dy = (x >= 0.f && x <= 1.f) ? q.lambda_rgb * 2.f * e * invn : 0.f;
    db[c] = dy * irr;
    if (A.rgb) A.rgb[ray * 3 + c] = y;
  }
  float dd = 0.f, dws = 0.f;
  if (A.tdepth && A.valid[ray * A.v_stride] > 0.f) {          // DepthLoss (metrics.py:82-161), as bn_lambert_tail
    const float td = A.tdepth[ray * A.td_stride], tw = A.tweight[ray * A.tw_stride], ts = A.tstd[ray * A.ts_stride];
    const float var = A.var ? A.var[ray] : 0.f;
    const bool apply = q.usealldepth || (fabsf(depth - td) - ts > 0.f) || (ts < sqrtf(var));
    if (apply) {
      const float k = q.lambda_ds / 3.f / (float)A.R;
      loss += k * tw * (depth - td) * (depth - td);
      dd = k * 2.f * tw * (depth - td);
    }
  }
  if (q.lambda_hs > 0.f) {
    // HardSurfaceLoss: lambda/R * sum_s w_s (z_s - depth)^2; the per-sample part of its gradient, lambda/R (z_s - depth)^2,
    // is added by the composite backward (hs_scale); d/d depth = -2 lambda/R (sum_s w_s z_s - depth sum_s w_s)
    const float k = q.lambda_hs / (float)A.R;
    loss += k * (A.var ? A.var[ray] : 0.f);
    dd += -2.f * k * (depth - depth * ws);
  }
  // A ray whose shaded value is NaN / Inf (a BRDF evaluated at a singular geometry that the reference's check_nan replacements
  // do not cover) makes the reference's loss NaN and its whole gradient with it.  With `nonfinite` (FusedTrainer.sanitize_grads)
  // such a ray is left out of the step - loss term 0, gradients 0 - and counted, like the non-finite gradient elements the
  // composite backward drops.
  if (A.extra_loss) loss += A.extra_loss[ray];
  const bool drop = A.nonfinite && !(fabsf(loss) <= 3.0e38f);
  if (drop) { atomicAdd(A.nonfinite + (isnan(loss) ? 0 : 1), 1ull); loss = 0.f; }
  if (A.ray_loss) A.ray_loss[ray] = loss;
  if (A.loss_acc) atomicAdd(A.loss_acc + (int)(ray % A.loss_slots), loss);
  // J^T: slots -> composited sums
  float *da = A.d_acc + ray * C;
  for (int c = 0; c < C; ++c) da[c] = 0.f;
  if (drop) { A.d_wsum[ray] = 0.f; A.d_depth[ray] = 0.f; return; }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float dw = jt(out, db, 3 + c);
    da[c] = dw * (1.f + 2.f * pad);
    dws -= dw * pad;
  }
  if (KIND != BN_SHADE_LAMBERT) {
#pragma unroll
    for (int c = 0; c < 3; ++c) da[q.ch_normal + c] = jt(out, db, c);
    if (KIND == BN_SHADE_RPV) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (q.ch_p0 >= 0) da[q.ch_p0 + c] = jt(out, db, 6 + c);
        if (q.ch_p1 >= 0) da[q.ch_p1 + c] = jt(out, db, 9 + c);
        if (q.ch_p2 >= 0 && !q.rhoc_is_albedo) da[q.ch_p2 + c] = jt(out, db, 12 + c);
      }
    } else if (KIND == BN_SHADE_HAPKE) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (q.ch_p0 >= 0) da[q.ch_p0 + c] = jt(out, db, 6 + c);
        if (q.ch_p1 >= 0) da[q.ch_p1 + c] = jt(out, db, 9 + c);
      }
      if (q.ch_p2 >= 0) da[q.ch_p2] = jt(out, db, 12);
    } else {
      da[q.ch_p0] = jt(out, db, 6);
    }
  }
  A.d_wsum[ray] = dws;
  A.d_depth[ray] = dd;
}

}  // namespace

extern "C" int bn_ray_shade_loss(const bn_shade_desc *desc, const float *acc, const float *wsum, const float *depth, const float *var,
                                 const float *rays_d, int64_t rd_stride, const float *sun_d, int64_t sd_stride, const float *rgbs,
                                 const float *valid_depth, int64_t v_stride, const float *target_depth, int64_t td_stride,
                                 const float *target_weight, int64_t tw_stride, const float *target_std, int64_t ts_stride, int64_t R,
                                 float *rgb, float *ray_loss, float *loss_acc, int32_t loss_slots, float *d_acc, float *d_wsum,
                                 float *d_depth, unsigned long long *nonfinite, const float *extra_loss, void *stream) {
  BN_REQUIRE(desc && acc && wsum && depth && rgbs && d_acc && d_wsum && d_depth && R > 0, "ray_shade_loss: null argument");
  const bn_shade_desc &q = *desc;
  BN_REQUIRE(q.C >= 4 && q.C <= BN_MAX_CH, "ray_shade_loss: C=%d unsupported", q.C);
  BN_REQUIRE(q.kind >= BN_SHADE_LAMBERT && q.kind <= BN_SHADE_MICROFACET, "ray_shade_loss: kind=%d", q.kind);
  auto in_range = [&](int ch, int n) { return ch < 0 || (ch >= 4 && ch + n <= q.C); };
  BN_REQUIRE(in_range(q.ch_normal, 3), "ray_shade_loss: normal channel %d outside [4, %d)", q.ch_normal, q.C);
  if (q.kind != BN_SHADE_LAMBERT) {
    BN_REQUIRE(q.ch_normal >= 4 && rays_d, "ray_shade_loss: BRDF shading needs a normal field and the ray directions");
    const int n2 = q.kind == BN_SHADE_HAPKE ? 1 : 3;
    const int n0 = q.kind == BN_SHADE_MICROFACET ? 1 : 3;
    BN_REQUIRE(in_range(q.ch_p0, n0) && in_range(q.ch_p1, 3) && in_range(q.ch_p2, n2), "ray_shade_loss: parameter channels (%d, %d, %d) outside [4, %d)",
               q.ch_p0, q.ch_p1, q.ch_p2, q.C);
    BN_REQUIRE(q.kind != BN_SHADE_MICROFACET || q.ch_p0 >= 4, "ray_shade_loss: microfacet needs the roughness channel");
    BN_REQUIRE(q.kind != BN_SHADE_HAPKE || q.ch_p0 >= 4 || (q.shell >= 1 && q.shell <= 3), "ray_shade_loss: Hapke without b needs shell_hapke in {1,2,3}");
  }
  BN_REQUIRE(!target_depth || (valid_depth && target_weight && target_std && var), "ray_shade_loss: incomplete depth prior");
  BN_REQUIRE(!(q.lambda_hs > 0.f) || var, "ray_shade_loss: lambda_hs needs the per-ray variance");
  ShadeArgs a;
  a.d = q; a.acc = acc; a.wsum = wsum; a.depth = depth; a.var = var; a.rays_d = rays_d; a.sun_d = sun_d; a.rd_stride = rd_stride;
  a.sd_stride = sd_stride; a.rgbs = rgbs; a.valid = valid_depth; a.tdepth = target_depth; a.tweight = target_weight; a.tstd = target_std;
  a.v_stride = v_stride; a.td_stride = td_stride; a.tw_stride = tw_stride; a.ts_stride = ts_stride; a.R = R; a.rgb = rgb;
  a.ray_loss = ray_loss; a.loss_acc = loss_acc; a.loss_slots = loss_slots > 0 ? loss_slots : 1; a.d_acc = d_acc; a.d_wsum = d_wsum;
  a.d_depth = d_depth; a.nonfinite = nonfinite; a.extra_loss = extra_loss;
  const dim3 grid((unsigned)ceil_div64(R, 64));
  hipStream_t st = (hipStream_t)stream;
  BnProfScope prof_(BN_K_BRDF, st);
  switch (q.kind) {
    case BN_SHADE_LAMBERT: ray_shade_loss_kernel<BN_SHADE_LAMBERT><<<grid, 64, 0, st>>>(a); break;
    case BN_SHADE_RPV: ray_shade_loss_kernel<BN_SHADE_RPV><<<grid, 64, 0, st>>>(a); break;
    case BN_SHADE_HAPKE: ray_shade_loss_kernel<BN_SHADE_HAPKE><<<grid, 64, 0, st>>>(a); break;
    default: ray_shade_loss_kernel<BN_SHADE_MICROFACET><<<grid, 64, 0, st>>>(a); break;
  }
  BN_LAUNCH_CHECK("ray_shade_loss");
  return 0;
}
