// The closed-form BRDFs as device functions over a scalar type S (float: values; Dual<N>: value + N partial derivatives,
// forward mode).  Shared by the standalone BRDF kernels (brdf.hip) and the fused per-ray shading + loss kernel
// (ray_tail.hip).  Replaces BRDF/basic_func.py:5-44, BRDF/RPV.py:6-63, BRDF/Hapke.py:6-200, BRDF/microfacet.py:20-118.
// Include it only from a translation unit compiled with `#pragma clang fp contract(off)` (see brdf.hip).
#pragma once
#include "common.h"

#define PI_F 3.14159265358979323846f

template <int N> struct Dual {
  float v;
  float d[N];
};

// ------------------------------------------------------------------ float primitives
__device__ __forceinline__ float val(float a) { return a; }
__device__ __forceinline__ float cst(float, float c) { return c; }
__device__ __forceinline__ float sin_(float a) { return sinf(a); }
__device__ __forceinline__ float cos_(float a) { return cosf(a); }
__device__ __forceinline__ float tan_(float a) { return tanf(a); }
__device__ __forceinline__ float acos_(float a) { return acosf(a); }
__device__ __forceinline__ float exp_(float a) { return expf(a); }
__device__ __forceinline__ float log_(float a) { return logf(a); }
__device__ __forceinline__ float sqrt_(float a) { return sqrtf(a); }
__device__ __forceinline__ float abs_(float a) { return fabsf(a); }
__device__ __forceinline__ float pow_(float a, float b) { return powf(a, b); }
__device__ __forceinline__ float powc_(float a, float c) { return powf(a, c); }
// torch.clamp propagates NaN (C fmin/fmax would drop it and change which NaN-replacement branch fires)
__device__ __forceinline__ float clamp_(float a, float lo, float hi) { return isnan(a) ? a : fminf(fmaxf(a, lo), hi); }
__device__ __forceinline__ float clamp_min_(float a, float lo) { return isnan(a) ? a : fmaxf(a, lo); }
__device__ __forceinline__ float nan_to(float y, float rep) { return isnan(y) ? rep : y; }
__device__ __forceinline__ float nan_to_num_(float y) {
  return isnan(y) ? 0.f : (isinf(y) ? (y > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f) : y);
}
__device__ __forceinline__ float detach_(float a) { return a; }
__device__ __forceinline__ float sel_(bool c, float a, float b) { return c ? a : b; }

// ------------------------------------------------------------------ dual primitives
template <int N> __device__ __forceinline__ float val(const Dual<N> &a) { return a.v; }
template <int N> __device__ __forceinline__ Dual<N> cst(const Dual<N> &, float c) {
  Dual<N> r; r.v = c;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = 0.f;
  return r;
}
// A partial that is structurally zero (the input does not reach this value) must stay zero even when the local
// derivative is inf/NaN (acos' at 1, 1/0 ...): reverse-mode autograd never visits such a path, 0 * inf would.
__device__ __forceinline__ float mz(float d, float x) { return d == 0.f ? 0.f : d * x; }
template <int N> __device__ __forceinline__ Dual<N> chain(const Dual<N> &a, float v, float dv) {
  Dual<N> r; r.v = v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = mz(a.d[i], dv);
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N> &a, const Dual<N> &b) {
  Dual<N> r; r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N> &a, const Dual<N> &b) {
  Dual<N> r; r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N> &a) { return chain(a, -a.v, -1.f); }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N> &a, const Dual<N> &b) {
  Dual<N> r; r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = mz(a.d[i], b.v) + mz(b.d[i], a.v);
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N> &a, const Dual<N> &b) {
  Dual<N> r; r.v = a.v / b.v;
  const float ib = 1.f / b.v, q = r.v * ib;   // d(a/b) = da/b - a db / b^2
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = mz(a.d[i], ib) - mz(b.d[i], q);
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N> &a, float c) { Dual<N> r = a; r.v += c; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(float c, const Dual<N> &a) { return a + c; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N> &a, float c) { Dual<N> r = a; r.v -= c; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(float c, const Dual<N> &a) { return chain(a, c - a.v, -1.f); }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N> &a, float c) { return chain(a, a.v * c, c); }
template <int N> __device__ __forceinline__ Dual<N> operator*(float c, const Dual<N> &a) { return a * c; }
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N> &a, float c) { return chain(a, a.v / c, 1.f / c); }
template <int N> __device__ __forceinline__ Dual<N> operator/(float c, const Dual<N> &a) {
  const float q = c / a.v;
  return chain(a, q, -q / a.v);
}
template <int N> __device__ __forceinline__ Dual<N> sin_(const Dual<N> &a) { return chain(a, sinf(a.v), cosf(a.v)); }
template <int N> __device__ __forceinline__ Dual<N> cos_(const Dual<N> &a) { return chain(a, cosf(a.v), -sinf(a.v)); }
template <int N> __device__ __forceinline__ Dual<N> tan_(const Dual<N> &a) {
  const float t = tanf(a.v);
  return chain(a, t, 1.f + t * t);
}
template <int N> __device__ __forceinline__ Dual<N> acos_(const Dual<N> &a) {
  return chain(a, acosf(a.v), -1.f / sqrtf(1.f - a.v * a.v));
}
template <int N> __device__ __forceinline__ Dual<N> exp_(const Dual<N> &a) {
  const float e = expf(a.v);
  return chain(a, e, e);
}
template <int N> __device__ __forceinline__ Dual<N> log_(const Dual<N> &a) { return chain(a, logf(a.v), 1.f / a.v); }
template <int N> __device__ __forceinline__ Dual<N> sqrt_(const Dual<N> &a) {
  const float s = sqrtf(a.v);
  return chain(a, s, 0.5f / s);
}
template <int N> __device__ __forceinline__ Dual<N> abs_(const Dual<N> &a) {
  return chain(a, fabsf(a.v), a.v > 0.f ? 1.f : (a.v < 0.f ? -1.f : 0.f));
}
// pow with constant exponent: d = c * a^(c-1)   (torch pow_backward)
template <int N> __device__ __forceinline__ Dual<N> powc_(const Dual<N> &a, float c) {
  return chain(a, powf(a.v, c), c == 0.f ? 0.f : c * powf(a.v, c - 1.f));
}
// pow with tensor exponent: d/da = b a^(b-1), d/db = a^b log(a) (0 where a == 0 and b >= 0)
template <int N> __device__ __forceinline__ Dual<N> pow_(const Dual<N> &a, const Dual<N> &b) {
  Dual<N> r; r.v = powf(a.v, b.v);
  const float da = b.v == 0.f ? 0.f : b.v * powf(a.v, b.v - 1.f);
  const float db = (a.v == 0.f && b.v >= 0.f) ? 0.f : r.v * logf(a.v);
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = mz(a.d[i], da) + mz(b.d[i], db);
  return r;
}
template <int N> __device__ __forceinline__ Dual<N> clamp_(const Dual<N> &a, float lo, float hi) {
  return chain(a, clamp_(a.v, lo, hi), (a.v >= lo && a.v <= hi) ? 1.f : 0.f);
}
template <int N> __device__ __forceinline__ Dual<N> clamp_min_(const Dual<N> &a, float lo) {
  return chain(a, clamp_min_(a.v, lo), a.v >= lo ? 1.f : 0.f);
}
template <int N> __device__ __forceinline__ Dual<N> nan_to(const Dual<N> &y, const Dual<N> &rep) { return isnan(y.v) ? rep : y; }
template <int N> __device__ __forceinline__ Dual<N> nan_to_num_(const Dual<N> &y) {
  if (isnan(y.v) || isinf(y.v)) return cst(y, nan_to_num_(y.v));
  return y;
}
template <int N> __device__ __forceinline__ Dual<N> detach_(const Dual<N> &a) { return cst(a, a.v); }
template <int N> __device__ __forceinline__ Dual<N> sel_(bool c, const Dual<N> &a, const Dual<N> &b) { return c ? a : b; }

template <typename S> struct V3 { S x, y, z; };
template <typename S> __device__ __forceinline__ S dot3(const V3<S> &a, const V3<S> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// ------------------------------------------------------------------ shared angles (basic_func.py:5-31)
template <typename S> struct Angles { S ci, sza, si, cv, vza, sv, cg, g, phi; };
template <typename S> __device__ __forceinline__ Angles<S> calc_angles(const V3<S> &l, const V3<S> &v, const V3<S> &n) {
  Angles<S> a;
  a.ci = clamp_(dot3(l, n), 1e-5f, 1.f);
  a.sza = acos_(a.ci);
  a.si = sin_(a.sza);
  a.cv = clamp_(dot3(v, n), 1e-5f, 1.f);
  a.vza = acos_(a.cv);
  a.sv = sin_(a.vza);
  a.cg = clamp_(dot3(v, l), -1.f, 1.f);
  a.g = acos_(a.cg);
  a.phi = acos_(clamp_((a.cg - a.ci * a.cv) / a.si / a.sv, -1.f, 1.f));
  return a;
}
// Henyey-Greenstein (basic_func.py:33-44)
template <typename S> __device__ __forceinline__ S hg(const S &x, const S &th) {
  const S t2 = th * th;
  const S y = (1.f - t2) / (powc_(1.f + 2.f * th * x + t2, 1.5f) + 1e-6f);
  return nan_to(y, cst(y, 0.f));
}

// ------------------------------------------------------------------ RPV (RPV.py:6-63)
template <typename S>
__device__ __forceinline__ void rpv_eval(const V3<S> &l, const V3<S> &v, const V3<S> &n, const S *w, const S *k, const S *th,
                                         const S *rc, S *brdf, float *aux) {
  const Angles<S> a = calc_angles(l, v, n);
  S G = cst(a.ci, 1.f);
  if (rc) {
    const S ti = tan_(a.sza), tv = tan_(a.vza), cp = cos_(a.phi);
    G = sqrt_(ti * ti + tv * tv - 2.f * ti * tv * cp + 1e-5f);
    G = detach_(nan_to(G, cst(G, 0.f)));                    // G_.detach()  RPV.py:55
  }
  const S base = a.ci * a.cv * (a.ci + a.cv) + 1e-5f;
  const S cgx = a.cg;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    S M1 = cst(base, 1.f), Fh = cst(base, 1.f), H = cst(base, 1.f);
    if (k) { M1 = pow_(base, k[c] - 1.f); M1 = nan_to(M1, cst(M1, 0.f)); }
    if (th) Fh = hg(cgx, th[c]);
    if (rc) { H = 1.f + (1.f - rc[c]) / (1.f + G + 1e-5f); H = nan_to(H, cst(H, 0.f)); }
    brdf[c] = w[c] * M1 * Fh * H;
    if (aux) { aux[c] = val(M1); aux[4 + c] = val(H); }
  }
  if (aux) { aux[3] = val(G); aux[7] = val(a.ci); aux[8] = val(a.cv); }
}

// ------------------------------------------------------------------ Hapke (Hapke.py:6-200)
template <typename S> __device__ __forceinline__ S hk_E1(const S &x, const S &th) {
  const S y = exp_(-(2.f / PI_F) / tan_(th + 1e-5f) / tan_(x + 1e-5f));
  return nan_to(y, cst(y, 0.f));
}
template <typename S> __device__ __forceinline__ S hk_E2(const S &x, const S &th) {
  const S a = 1.f / tan_(th + 1e-5f), b = 1.f / tan_(x + 1e-5f);
  const S y = exp_(-(1.f / PI_F) * (a * a) * (b * b));
  return nan_to(y, cst(y, 0.f));
}
template <typename S> __device__ __forceinline__ S hk_f(const S &phi) {
  const S y = exp_(-2.f * tan_((phi + 1e-5f) / 2.f));
  return nan_to(y, cst(y, 0.f));
}
template <typename S> __device__ __forceinline__ S hk_chi(const S &x) {
  const S t = tan_(x + 1e-5f);
  const S y = 1.f / sqrt_(1.f + PI_F * (t * t));
  return nan_to(y, cst(y, 0.f));
}
template <typename S> __device__ __forceinline__ S hk_eta(const S &x, const S &th) {
  const S y = hk_chi(th) * (cos_(x) + sin_(x) * tan_(th + 1e-5f) * (hk_E2(x, th) / (2.f - hk_E1(x, th))));
  return nan_to(y, cst(y, 0.f));
}
template <typename S> __device__ __forceinline__ S hk_mu0(const S &i, const S &e, const S &phi, const S &th) {
  const S sp = sin_(phi / 2.f);
  S y;
  if (val(i) <= val(e)) {
    y = cos_(phi) * hk_E2(e, th) + sp * sp * hk_E2(i, th);
    y = y / (2.f - hk_E1(e, th) - phi / PI_F * hk_E1(i, th));
  } else {
    y = hk_E2(i, th) - sp * sp * hk_E2(e, th);
    y = y / (2.f - hk_E1(i, th) - phi / PI_F * hk_E1(e, th));
  }
  y = hk_chi(th) * (cos_(i) + sin_(i) * tan_(th) * y);
  return nan_to(y, cos_(i));
}
template <typename S> __device__ __forceinline__ S hk_mu(const S &i, const S &e, const S &phi, const S &th) {
  const S sp = sin_(phi / 2.f);
  S y;
  if (val(i) <= val(e)) {
    y = hk_E2(e, th) - sp * sp * hk_E2(i, th);
    y = y / (2.f - hk_E1(e, th) - phi / PI_F * hk_E1(i, th));
  } else {
    y = cos_(phi) * hk_E2(i, th) + sp * sp * hk_E2(e, th);
    y = y / (2.f - hk_E1(i, th) - phi / PI_F * hk_E1(e, th));
  }
  y = hk_chi(th) * (cos_(e) + sin_(e) * tan_(th) * y);
  return nan_to(y, cos_(e));
}
template <typename S> __device__ __forceinline__ S hk_shadow(const S &i, const S &e, const S &phi, const S &th) {
  const S ci = cos_(i), cv = cos_(e);
  const S mue = hk_mu(i, e, phi, th), etai = hk_eta(i, th), etae = hk_eta(e, th), chit = hk_chi(th), ff = hk_f(phi);
  const S temp = (mue / etae) * (ci / etai) * chit;
  S y;
  if (val(i) <= val(e)) y = temp / (1.f - ff + ff * chit * (ci / etai));
  else y = temp / (1.f - ff + ff * chit * (cv / etae));
  return nan_to(y, cst(y, 0.f));
}
template <typename S> __device__ __forceinline__ S hk_PF(const S &x, const S &b, const S &c) {
  const S b2 = b * b, bx = b * x;
  S y = c * (1.f - b2) / (powc_(1.f - 2.f * bx + b2, 1.5f) + 1e-6f);
  y = y + (1.f - c) * (1.f - b2) / (powc_(1.f + 2.f * bx + b2, 1.5f) + 1e-6f);
  return nan_to(y, cst(y, 0.f));
}
template <typename S> __device__ __forceinline__ S hk_HF(const S &x, const S &w) {
  const S gamma = sqrt_(1.f - w);
  const S ro = (1.f - gamma) / (1.f + gamma);
  const S lg = log_(abs_((1.f + x) / x));
  const S y = powc_(1.f - w * x * (ro + (1.f - 2.f * ro * x) / 2.f * lg), -1.f);
  return nan_to(y, cst(y, 1.f));
}
template <typename S>
__device__ __forceinline__ void hapke_eval(const V3<S> &l, const V3<S> &v, const V3<S> &n, const S *w, const S *b, const S *c,
                                           const S *theta, float hpk_scl, int shell, S *brdf, float *aux) {
  const Angles<S> a = calc_angles(l, v, n);
  S ci = a.ci, cv = a.cv, Sh = cst(a.ci, 1.f);
  if (theta) {
    ci = hk_mu0(a.sza, a.vza, a.phi, *theta);
    cv = hk_mu(a.sza, a.vza, a.phi, *theta);
    Sh = hk_shadow(a.sza, a.vza, a.phi, *theta);
  }
  const S t1 = ci / (ci + cv) / cos_(a.sza);
  const S scl = (ci + cv) * hpk_scl + 1e-6f;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    S P = cst(a.ci, 1.f);
    if (b) P = c ? hk_PF(a.cg, b[ch], c[ch]) : hg(a.cg, b[ch]);
    const S Hi = hk_HF(ci, w[ch]), Hv = hk_HF(cv, w[ch]);
    if (!b) {
      if (shell == 1) brdf[ch] = w[ch] / hpk_scl;
      else if (shell == 2) brdf[ch] = w[ch] / scl;
      else brdf[ch] = w[ch] * (Hi * Hv) / scl;
    } else {
      brdf[ch] = w[ch] / hpk_scl * t1 * (P + Hi * Hv - 1.f) * Sh;   // B == 1 (B0, h are None)
    }
    if (aux) { aux[ch] = val(P); aux[3 + ch] = val(Hi); aux[6 + ch] = val(Hv); }
  }
  if (aux) { aux[9] = val(Sh); aux[10] = val(ci); aux[11] = val(cv); }
}

// ------------------------------------------------------------------ GGX microfacet (microfacet.py:20-118)
template <typename S> __device__ __forceinline__ V3<S> safe_norm(const V3<S> &a) {
  const S nn = clamp_min_(sqrt_(dot3(a, a)), 1e-6f);
  V3<S> r = {a.x / nn, a.y / nn, a.z / nn};
  return r;
}
template <typename S>
__device__ __forceinline__ void microfacet_eval(const V3<S> &l0, const V3<S> &v0, const V3<S> &n0, const S *albedo, const S &rough,
                                                float f0, S *brdf, float *aux) {
  const V3<S> l = safe_norm(l0), v = safe_norm(v0), n = safe_norm(n0);
  V3<S> hs = {l.x + v.x, l.y + v.y, l.z + v.z};
  const V3<S> h = safe_norm(hs);
  const S alpha = rough * rough;
  const S a2 = alpha * alpha;
  const S cm = dot3(h, n);
  const float chi = val(cm) > 0.f ? 1.f : 0.f;
  const S cm2 = cm * cm;
  const S tan2 = nan_to_num_((1.f - cm2) / cm2);
  const S den = PI_F * (cm2 * cm2) * ((a2 + tan2) * (a2 + tan2));
  const S d = nan_to_num_(a2 * chi / den);
  const S ldn = clamp_min_(abs_(dot3(l, n)), 0.001f);
  const S vdn = clamp_min_(abs_(dot3(v, n)), 0.001f);
  const S glossy = nan_to_num_(0.04f * d / (4.f * ldn * vdn));
#pragma unroll
  for (int c = 0; c < 3; ++c) brdf[c] = albedo[c] + glossy;
  if (aux) {
    const float om = 1.f - val(dot3(l, h));
    const float f = f0 + (1.f - f0) * om * om * om * om * om;
    // _get_g(v, h, n): visualisation only
    const float cosv = val(dot3(n, v));
    const float div = nan_to_num_(val(dot3(h, v)) / cosv);
    const float cv2 = fminf(fmaxf(cosv * cosv, 0.f), 1.f);
    float tv2 = nan_to_num_((1.f - cv2) / cv2);
    tv2 = nan_to_num_(fmaxf(tv2, 0.f));
    const float a2f = val(a2);
    const float g = nan_to_num_((div > 0.f ? 2.f : 0.f) / (1.f + sqrtf(1.f + a2f * tv2)));
    aux[0] = val(glossy); aux[1] = f; aux[2] = g; aux[3] = val(d); aux[4] = val(ldn); aux[5] = val(vdn);
    aux[6] = val(h.x); aux[7] = val(h.y); aux[8] = val(h.z); aux[9] = val(cm);
  }
}

// ------------------------------------------------------------------ seeding / J^T products
template <int N> __device__ __forceinline__ Dual<N> seed(float v, int slot) {
  Dual<N> r; r.v = v;
#pragma unroll
  for (int i = 0; i < N; ++i) r.d[i] = i == slot ? 1.f : 0.f;
  return r;
}
template <int N> __device__ __forceinline__ V3<Dual<N>> cst3(const V3<float> &a) {
  Dual<N> z; z.v = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) z.d[i] = 0.f;
  V3<Dual<N>> r = {cst(z, a.x), cst(z, a.y), cst(z, a.z)};
  return r;
}
// d_in[slot] = sum_c d_brdf[c] * d brdf_c / d in[slot]
template <int N> __device__ __forceinline__ float jt(const Dual<N> *brdf, const float *db, int slot) {
  return db[0] * brdf[0].d[slot] + db[1] * brdf[1].d[slot] + db[2] * brdf[2].d[slot];
}
