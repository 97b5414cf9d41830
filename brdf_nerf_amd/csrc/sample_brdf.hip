// Per-SAMPLE shading on the field-output rows as they are stored, forward and backward, one thread per row (round 5):
// --MultiBRDF (models/spsbrdfnerf.py:289-307, 350-352: every sample is shaded by its own BRDF and the shaded colours are
// composited), and the per-sample irradiance of the sun-visibility pass (:265-273, :350-354: with --sun_v analystic each sample's
// colour - its BRDF value, or its padded albedo when no BRDF shades - is weighted by the sun pass's transparency at its position).  The BRDF is a pointwise function of a row - its normal, albedo and BRDF-parameter channels - and of
// its ray's sun / view directions, so the training step evaluates it on the [pass-1 block | guided block] rows without
// materialising the merged set and composites the result through the sort index (bn_lambert_tail / bn_merged_composite_*).
// Rounds 3-4 did this through torch.autograd.Function wrappers of the per-point BRDF kernels (brdf.hip): ~20 glue launches
// (gathers of the per-ray directions, channel slices, the padding / irradiance arithmetic and their backward).  Here: ONE launch
// forward (the padded, irradiance-weighted colour as channels 0-2 of a 4-wide [bp, sigma] or full-width copy of the rows) and
// ONE backward (J^T of the same evaluation in forward-mode duals, like ray_tail.hip, added to the pass-through channels).
#include "common.h"
#include "brdfnerf_hip.h"
#include "prof.h"
// (no FMA contraction, like brdf.hip: the degenerate-geometry branches must round like the reference's separate ATen ops)
#pragma clang fp contract(off)
#include "brdf_eval.h"

namespace {

struct SampleArgs {
  bn_shade_desc d;       // kind, C, ch_normal, ch_p0..2, rhoc_is_albedo, shell, cos_irradiance, hpk_scl, f0, rgb_padding
  const float *X;        // [N][C] field-output rows
  const float *rays;     // [R][ray_stride]: view = -rays[3:6], sun = rays[sun_col : +3] (sun_col < 0: (1, 1, 1))
  int64_t ray_stride;
  int sun_col;
  int64_t N, n1;         // rows [0, n1) belong to ray row / S1, rows [n1, N) to ray (row - n1) / S2
  int S1, S2;
  float *B;              // forward: [N][b_stride]; b_stride == 4: [bp, sigma]; b_stride == C: the row with channels 0-2 replaced by bp
  int b_stride;
  const float *dB;       // backward: [N][b_stride]
  float *dX;             // backward: [N][C]
};

template <int KIND> struct SSlots { static constexpr int N = KIND == BN_SHADE_RPV ? 15 : KIND == BN_SHADE_HAPKE ? 13 : 7; };

// the row's BRDF over the scalar type S (float: values; Dual<N>: values + Jacobian).  Slots as in brdf.hip / ray_tail.hip:
// normal 0-2, albedo 3-5, then RPV k 6-8, theta 9-11, rhoc 12-14 | Hapke b 6-8, c 9-11, theta 12 | microfacet roughness 6.
template <int KIND, typename S, typename Seed>
__device__ __forceinline__ void row_brdf(const bn_shade_desc &q, const float *x, const float (&sun)[3], const float (&view)[3], Seed seed_,
                                         S (&out)[3]) {
  const V3<S> n = {seed_(x[q.ch_normal], 0), seed_(x[q.ch_normal + 1], 1), seed_(x[q.ch_normal + 2], 2)};
  S w[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) w[c] = seed_(x[c], 3 + c);
  const V3<S> l = {cst(w[0], sun[0]), cst(w[0], sun[1]), cst(w[0], sun[2])}, v = {cst(w[0], view[0]), cst(w[0], view[1]), cst(w[0], view[2])};
  if (KIND == BN_SHADE_RPV) {
    S k[3], th[3], rc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      k[c] = seed_(q.ch_p0 >= 0 ? x[q.ch_p0 + c] : 0.f, 6 + c);
      th[c] = seed_(q.ch_p1 >= 0 ? x[q.ch_p1 + c] : 0.f, 9 + c);
      rc[c] = q.rhoc_is_albedo ? w[c] : seed_(q.ch_p2 >= 0 ? x[q.ch_p2 + c] : 0.f, 12 + c);      // funcH == 2 (spsbrdfnerf.py:288-291)
    }
    rpv_eval<S>(l, v, n, w, q.ch_p0 >= 0 ? k : nullptr, q.ch_p1 >= 0 ? th : nullptr, (q.ch_p2 >= 0 || q.rhoc_is_albedo) ? rc : nullptr,
                out, nullptr);
  } else if (KIND == BN_SHADE_HAPKE) {
    S b[3], cc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      b[c] = seed_(q.ch_p0 >= 0 ? x[q.ch_p0 + c] : 0.f, 6 + c);
      cc[c] = seed_(q.ch_p1 >= 0 ? x[q.ch_p1 + c] : 0.f, 9 + c);
    }
    const S th = seed_(q.ch_p2 >= 0 ? x[q.ch_p2] : 0.f, 12);
    hapke_eval<S>(l, v, n, w, q.ch_p0 >= 0 ? b : nullptr, q.ch_p1 >= 0 ? cc : nullptr, q.ch_p2 >= 0 ? &th : nullptr, q.hpk_scl, q.shell, out,
                  nullptr);
  } else {
    const S rg = seed_(x[q.ch_p0], 6);
    microfacet_eval<S>(l, v, n, w, rg, q.f0, out, nullptr);
  }
}

template <int KIND, bool BWD> __global__ __launch_bounds__(128) void sample_brdf_kernel(const SampleArgs A) {
  const int64_t row = (int64_t)blockIdx.x * 128 + threadIdx.x;
  if (row >= A.N) return;
  const bn_shade_desc &q = A.d;
  const int C = q.C;
  const int64_t ray = row < A.n1 ? row / A.S1 : (row - A.n1) / A.S2;
  const float *rr = A.rays + ray * A.ray_stride;
  const float view[3] = {-rr[3], -rr[4], -rr[5]};
  const float sun[3] = {A.sun_col >= 0 ? rr[A.sun_col] : 1.f, A.sun_col >= 0 ? rr[A.sun_col + 1] : 1.f, A.sun_col >= 0 ? rr[A.sun_col + 2] : 1.f};
  const float pad = q.rgb_padding;
  // bp = brdf (1 + 2 pad) - pad, times |sun_z| with the cosine irradiance of an upward normal (spsbrdfnerf.py:260-264, 270-275:
  // only a model with a normal field has it), else times the row's sun visibility (:265-273), else as it is
  const float gain = (1.f + 2.f * pad);
  const float irr = (q.cos_irradiance && q.ch_normal >= 0) ? fabsf(sun[2]) : (q.irr ? q.irr[row * q.irr_stride] : 1.f);
  const float *x = A.X + row * C;
  if constexpr (KIND == BN_SHADE_LAMBERT) {
    // no BRDF: the padded albedo itself (the Lambertian rgb under a per-sample irradiance)
    if constexpr (!BWD) {
      float *b = A.B + row * A.b_stride;
      if (A.b_stride == C)
        for (int c = 4; c < C; ++c) b[c] = x[c];
#pragma unroll
      for (int c = 0; c < 3; ++c) b[c] = (x[c] * gain - pad) * irr;
      b[3] = x[3];
    } else {
      const float *db_ = A.dB + row * A.b_stride;
      float *dx = A.dX + row * C;
#pragma unroll
      for (int c = 0; c < 3; ++c) dx[c] = db_[c] * gain * irr;
      dx[3] = db_[3];
      for (int c = 4; c < C; ++c) dx[c] = A.b_stride == C ? db_[c] : 0.f;
    }
    return;
  }
  if constexpr (!BWD) {
    float out[3];
    row_brdf<KIND, float>(q, x, sun, view, [](float v_, int) { return v_; }, out);
    float *b = A.B + row * A.b_stride;
    if (A.b_stride == C)
      for (int c = 4; c < C; ++c) b[c] = x[c];
#pragma unroll
    for (int c = 0; c < 3; ++c) b[c] = (out[c] * gain - pad) * irr;
    b[3] = x[3];
  } else {
    constexpr int NS = SSlots<KIND>::N;
    typedef Dual<NS> D;
    D out[3];
    row_brdf<KIND, D>(q, x, sun, view, [](float v_, int slot) { return seed<NS>(v_, slot); }, out);
    const float *db_ = A.dB + row * A.b_stride;
    const float db[3] = {db_[0] * gain * irr, db_[1] * gain * irr, db_[2] * gain * irr};
    float *dx = A.dX + row * C;
    // pass-through channels: sigma always; everything behind it when the copy is full width (the regularisers' terms on the
    // per-sample normals arrive there); the colour channels of the copy are the BRDF's output, so the row's own albedo gets the
    // BRDF path only
    dx[3] = db_[3];
    for (int c = 4; c < C; ++c) dx[c] = A.b_stride == C ? db_[c] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dx[c] = jt(out, db, 3 + c);
      dx[q.ch_normal + c] += jt(out, db, c);
    }
    if (KIND == BN_SHADE_RPV) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (q.ch_p0 >= 0) dx[q.ch_p0 + c] += jt(out, db, 6 + c);
        if (q.ch_p1 >= 0) dx[q.ch_p1 + c] += jt(out, db, 9 + c);
        if (q.ch_p2 >= 0 && !q.rhoc_is_albedo) dx[q.ch_p2 + c] += jt(out, db, 12 + c);
      }
    } else if (KIND == BN_SHADE_HAPKE) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (q.ch_p0 >= 0) dx[q.ch_p0 + c] += jt(out, db, 6 + c);
        if (q.ch_p1 >= 0) dx[q.ch_p1 + c] += jt(out, db, 9 + c);
      }
      if (q.ch_p2 >= 0) dx[q.ch_p2] += jt(out, db, 12);
    } else {
      dx[q.ch_p0] += jt(out, db, 6);
    }
  }
}

int check_args(const bn_shade_desc *desc, const float *X, const float *rays, int64_t R, int64_t ray_stride, int32_t sun_col, int64_t N,
               int64_t n1, int32_t S1, int32_t S2, int32_t stride, const char *what) {
  BN_REQUIRE(desc && X && rays && N > 0, "%s: null argument", what);
  const bn_shade_desc &q = *desc;
  BN_REQUIRE(q.kind >= BN_SHADE_LAMBERT && q.kind <= BN_SHADE_MICROFACET, "%s: kind=%d", what, q.kind);
  BN_REQUIRE(q.C >= 4 && q.C <= BN_MAX_CH && (stride == 4 || stride == q.C), "%s: C=%d, row stride %d (4 or C)", what, q.C, stride);
  auto in_range = [&](int ch, int n) { return ch < 0 || (ch >= 4 && ch + n <= q.C); };
  const int n2 = q.kind == BN_SHADE_HAPKE ? 1 : 3, n0 = q.kind == BN_SHADE_MICROFACET ? 1 : 3;
  BN_REQUIRE(in_range(q.ch_normal, 3) && (q.kind == BN_SHADE_LAMBERT || q.ch_normal >= 4), "%s: normal channel %d outside [4, %d)", what,
             q.ch_normal, q.C);
  BN_REQUIRE(q.kind == BN_SHADE_LAMBERT || (in_range(q.ch_p0, n0) && in_range(q.ch_p1, 3) && in_range(q.ch_p2, n2)),
             "%s: parameter channels %d %d %d outside [4, %d)", what, q.ch_p0, q.ch_p1, q.ch_p2, q.C);
  BN_REQUIRE(!q.irr || q.irr_stride >= 0, "%s: irradiance stride %lld", what, (long long)q.irr_stride);
  BN_REQUIRE(q.kind != BN_SHADE_MICROFACET || q.ch_p0 >= 4, "%s: microfacet needs the roughness channel", what);
  BN_REQUIRE(q.kind != BN_SHADE_HAPKE || q.ch_p0 >= 4 || (q.shell >= 1 && q.shell <= 3), "%s: Hapke without b needs shell_hapke in {1,2,3}", what);
  BN_REQUIRE(ray_stride >= 6 && (sun_col < 0 || sun_col + 3 <= ray_stride), "%s: ray stride %lld, sun column %d", what, (long long)ray_stride, sun_col);
  // every row's ray exists: the blocks are R rays of S1 (and, behind row n1, R rays of S2) samples each
  BN_REQUIRE(R > 0 && S1 > 0 && n1 == R * (int64_t)S1 && (n1 == N || (S2 > 0 && N - n1 == R * (int64_t)S2)),
             "%s: %lld rows do not split into %lld rays of %d (+ %d) samples at row %lld", what, (long long)N, (long long)R, S1, S2, (long long)n1);
  return 0;
}

template <bool BWD> int launch(const SampleArgs &a, hipStream_t st) {
  const dim3 grid((unsigned)ceil_div64(a.N, 128));
  BnProfScope prof_(BN_K_BRDF, st);
  switch (a.d.kind) {
    case BN_SHADE_LAMBERT: sample_brdf_kernel<BN_SHADE_LAMBERT, BWD><<<grid, 128, 0, st>>>(a); break;
    case BN_SHADE_RPV: sample_brdf_kernel<BN_SHADE_RPV, BWD><<<grid, 128, 0, st>>>(a); break;
    case BN_SHADE_HAPKE: sample_brdf_kernel<BN_SHADE_HAPKE, BWD><<<grid, 128, 0, st>>>(a); break;
    default: sample_brdf_kernel<BN_SHADE_MICROFACET, BWD><<<grid, 128, 0, st>>>(a); break;
  }
  BN_LAUNCH_CHECK(BWD ? "sample_brdf_backward" : "sample_brdf_forward");
  return 0;
}

}  // namespace

extern "C" int bn_sample_brdf_forward(const bn_shade_desc *desc, const float *X, const float *rays, int64_t R, int64_t ray_stride,
                                      int32_t sun_col, int64_t N, int64_t n1, int32_t S1, int32_t S2, float *B, int32_t b_stride,
                                      void *stream) {
  if (int e = check_args(desc, X, rays, R, ray_stride, sun_col, N, n1, S1, S2, b_stride, "sample_brdf_forward")) return e;
  BN_REQUIRE(B, "sample_brdf_forward: null output");
  SampleArgs a;
  a.d = *desc; a.X = X; a.rays = rays; a.ray_stride = ray_stride; a.sun_col = sun_col; a.N = N; a.n1 = n1; a.S1 = S1; a.S2 = S2 > 0 ? S2 : 1;
  a.B = B; a.b_stride = b_stride; a.dB = nullptr; a.dX = nullptr;
  return launch<false>(a, (hipStream_t)stream);
}

extern "C" int bn_sample_brdf_backward(const bn_shade_desc *desc, const float *X, const float *rays, int64_t R, int64_t ray_stride,
                                       int32_t sun_col, int64_t N, int64_t n1, int32_t S1, int32_t S2, const float *dB, int32_t b_stride,
                                       float *dX, void *stream) {
  if (int e = check_args(desc, X, rays, R, ray_stride, sun_col, N, n1, S1, S2, b_stride, "sample_brdf_backward")) return e;
  BN_REQUIRE(dB && dX, "sample_brdf_backward: null argument");
  SampleArgs a;
  a.d = *desc; a.X = X; a.rays = rays; a.ray_stride = ray_stride; a.sun_col = sun_col; a.N = N; a.n1 = n1; a.S1 = S1; a.S2 = S2 > 0 ? S2 : 1;
  a.B = nullptr; a.b_stride = b_stride; a.dB = dB; a.dX = dX;
  return launch<true>(a, (hipStream_t)stream);
}
