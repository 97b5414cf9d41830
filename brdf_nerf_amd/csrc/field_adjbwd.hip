// Backward of the analytic-normal adjoint chain (the "double backward" of calc_normals with create_graph=True,
// models/spsbrdfnerf.py:648-660, in training with --normal analystic).
//
// Forward adjoint (field_adjoint.hip):  a_L = s' w_sigma ; delta_l = a_{l+1} (.) D_l ; [g_PE ; a_l] += W_l^T delta_l ;
//                                        g = J_PE(x)^T g_PE ; n = -g / |g|.
// The forward chain runs, and stashes, the s'-free quantities a' = a / s', delta' = delta / s' (s' = sigmoid(sigma_raw) is a
// per-point scalar).  Everything below is bilinear in (forward, backward) quantities, so this kernel runs on
// gbar' = s' gbar, abar' = s' abar, dbar' = s' dbar and every product it forms is the true one:
//   zbar_l = -w0^2 y_l dbar_l a_{l+1} = -w0^2 y_l dbar'_l a'_{l+1};  dW_l += delta_l^T [..] = delta'_l^T [gbar'_PE ; abar'_l];
//   dw_sigma += s'^T abar_L = 1^T abar'_L;  sbar = (w_sigma . abar_L) s'(1 - s') = (w_sigma . abar'_L)(1 - s').
// The seed s' gbar = -s' (dn - n (n . dn)) / |g| has |g| = s' |g'|: the tiny factor cancels there too.
// Given dL/dn this kernel walks the chain the other way, which has the shape of a FORWARD pass of the field:
//   gbar_PE = J_PE(x) gbar ;   for l = 0 .. L-1:  dbar_l = W_l [gbar_PE ; abar_l]      (same packed weights as the forward)
//                                                  abar_{l+1} = dbar_l (.) D_l
//                                                  zbar_l = dbar_l (.) a_{l+1} (.) dD_l/dz_l = -w0^2 y_l dbar_l a_{l+1}
//   sbar = (w_sigma . abar_L) s'(1 - s')
// and stashes what the parameter gradients need:  dW_l += delta_l^T [gbar_PE ; abar_l]  (weight-gradient GEMMs),
// dw_sigma += s'^T abar_L (skinny), zbar_l and sbar (added to the primal backward chain's pre-activation gradients).
#include "field_kernels.h"

struct AdjBwdArgs {
  FieldGeom g;
  bn_field_params p;
  PackedLayout pl;
  StashLayout sl;
  const void *packed;
  bn_points pts;
  const float *d_out;
  char *stash;
  int prescaled;   // forward packs carry w0/(2 pi) (16-bit Siren modes): undo it here
  float *amax;         // fp16 mode: amax[1] = max |gbar_PE| picks this chain's loss scale (common.h); amax[2] receives
                       // max |zbar_l| (true scale) for the primal chain's; else nullptr
};

template <typename T, int MT, int NT, int WAVES, bool D16>
__global__ __launch_bounds__(WAVES * 64, 2) void field_adjbwd_kernel(const AdjBwdArgs A) {
  typedef typename DKind<T, D16>::type DK;     // kind of the derivative stash (field_kernels.h)
  typedef typename Elem<T>::vec4 vec4;
  constexpr int BM = MT * 32;
  constexpr int PADE = Elem<T>::kPad;
  constexpr bool NATY = Elem<T>::kNativeY;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FieldGeom &g = A.g;
  const int F = g.F, LDA = F + PADE, KP = g.KP, LDP = KP + PADE, P = g.P;
  T *ACT = (T *)smem;
  T *PE = ACT + (size_t)BM * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int64_t tile = blockIdx.x, m0 = tile * BM, M = A.pts.n_points;
  const T *packed = (const T *)A.packed;
  constexpr int DP = BwdDepth<T>::value;
  // fp16 mode: the chain runs scaled by gs (a power of two): gbar_PE, abar_l, zbar_l carry it (the weight-gradient jobs
  // and the primal chain remove it); sbar, an fp32 scalar per point, is stored unscaled
  const float gs = grad_scale_from(A.amax ? A.amax + 1 : nullptr, BN_GS_TARGET_ADJ);
  constexpr bool TRACK = std::is_same<T, f16>::value;   // fp16 only: the largest |zbar| this workgroup hands to the primal chain
  float zmax = 0.f;

  // ---------------------------------------------------------------- dL/dn -> dL/dg -> dL/dg_PE
  if (tid < BM) {
    const int64_t gm = m0 + tid;
    T *row = PE + (size_t)tid * LDP;
    float gb[3] = {0.f, 0.f, 0.f}, x[3] = {0.f, 0.f, 0.f};
    if (gm < M) {
      const float *gx = (const float *)(A.stash + A.sl.gradx) + gm * 4;
      const float *dn = A.d_out + gm * g.C + g.ch_normal_an;
      const float n2 = gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2];
      const float eps = 1.1920928955078125e-07f;
      const float inv = 1.f / sqrtf(fmaxf(n2, eps));
      const float gd = gx[0] * dn[0] + gx[1] * dn[1] + gx[2] * dn[2];
      const float k = n2 > eps ? gd * inv * inv * inv : 0.f;     // n = -g * inv
      const float spm = ((const float *)(A.stash + A.sl.sprime))[gm];       // gbar' = s' gbar
#pragma unroll
      for (int c = 0; c < 3; ++c) gb[c] = -(dn[c] * inv - gx[c] * k) * spm * gs;
      if (A.pts.xyz) {
        x[0] = A.pts.xyz[gm * 3]; x[1] = A.pts.xyz[gm * 3 + 1]; x[2] = A.pts.xyz[gm * 3 + 2];
      } else {
        // (a set of two sample blocks of the same rays: bn_points.seg1_points)
        const bool second = A.pts.seg1_points > 0 && gm >= A.pts.seg1_points;
        const int64_t gl = second ? gm - A.pts.seg1_points : gm;
        const float *rr = A.pts.rays + (gl / (second ? A.pts.n_samples2 : A.pts.n_samples)) * A.pts.ray_stride;
        const float zz = second ? A.pts.z2[gl] : A.pts.z[gl];
        x[0] = rr[0] + rr[3] * zz; x[1] = rr[1] + rr[4] * zz; x[2] = rr[2] + rr[5] * zz;
      }
    }
    if (g.pe_freqs > 0) {
      for (int k = 0; k < g.pe_freqs; ++k) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float s, co;
          sincos_cw(f * x[c], s, co);
          row[6 * k + c] = (T)(f * co * gb[c]);          // g_c = sum f (cos g_PE[sin] - sin g_PE[cos])
          row[6 * k + 3 + c] = (T)(-f * s * gb[c]);
        }
      }
      for (int k = P; k < KP; ++k) row[k] = (T)0.f;
    } else {
      for (int k = 0; k < KP; ++k) row[k] = (T)(k < 3 ? gb[k] : 0.f);
    }
  }
  __syncthreads();
  tile_to_global<T>(PE, LDP, (T *)(A.stash + A.sl.gbar_pe) + (size_t)m0 * KP, KP, BM, KP);

  const int ncol0 = wave * 32 * NT;
  const bool wave_on = ncol0 < F;
  const int KSP = KP / 16, KSF = F / 16;
  f32x16 acc[NT][MT];
  const int n_on = F / (32 * NT) < WAVES ? F / (32 * NT) : WAVES;
  // the row-major stash copy of abar_l (the tile layer l's GEMM reads) rides inside that GEMM when the shape fits
  const bool ride = NT == 2 ? true : tile_copy_exact(F, n_on, WAVES);
  // 16-bit modes (round 4): the layer's derivative bytes leave HBM before its GEMM (32 registers, as in the primal backward
  // chain), and its a_{l+1} / y_l pieces in batches of ADJ_GRP point tiles - one exposed latency per batch instead of one per
  // piece (the loads used to sit beside their uses inside the epilogue loops).
  constexpr bool PRE = std::is_same<DK, DK8>::value;     // (DK16: 64 registers for a layer - fetched per piece, like the fp32 mode's)
#ifndef ADJ_GRP
#define ADJ_GRP 2      // point tiles per batch of a_{l+1} / y_l loads (4 = a whole n-tile: 212 B of scratch per lane)
#endif
  DPiece<DK> dpre[PRE ? NT : 1][PRE ? MT : 1];
  for (int l = 0; l < g.L; ++l) {
    zero_acc<MT, NT>(acc);
    T *adst = (T *)(A.stash + A.sl.adj_abar[l]) + (size_t)m0 * F;   // abar_l, l >= 1
    if (!ride && l > 0) tile_to_global<T>(ACT, LDA, adst, F, BM, F);
    if constexpr (PRE) {
      if (wave_on) {
        const char *Dp = A.stash + A.sl.D[l] + (size_t)tile * dtile_bytes<DK>(BM, F);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) dpre[nt][mt] = dpiece_load<DK>(Dp + dpiece_off<DK, MT, NT>(wave, nt, mt, lane));
      }
    }
    if (wave_on) {
      const size_t t0 = (size_t)(ncol0 / 32);
      const T *w_pe = packed + A.pl.fwd_trunk[l][0] + t0 * KSP * 512;
      const T *w_h = packed + A.pl.fwd_trunk[l][l == g.skip ? 1 : 0] + t0 * KSF * 512;
      if (l == 0 || l == g.skip) gemm_seg<T, MT, NT, DP>(acc, w_pe, KSP, PE, LDP, lane);
      if (l > 0) {
        if (ride) {
          TileCopyExact<T> acopy(ACT, LDA, adst, F, F, tid, WAVES * 64);
          // (as a straight-line stream - gemm_fixed - this product takes the kernel from 0 to 624 B of scratch per lane: looped form)
          gemm_seg<T, MT, NT, DP>(acc, w_h, KSF, ACT, LDA, lane, acopy);
        } else {
          gemm_seg<T, MT, NT, DP>(acc, w_h, KSF, ACT, LDA, lane);
        }
      }
    }
    __syncthreads();
    // fp32 mode: y_l is stashed row-major (for the weight-gradient GEMMs): stage the tile through ACT, which the GEMM has
    // finished reading, with coalesced 16-byte loads, instead of 8-byte reads scattered over 32 rows per wave instruction.
    // Each lane then reads the values at exactly the positions it overwrites with abar_{l+1}.  16-bit modes: y_l is stashed
    // in accumulator order and read straight into registers in the epilogue below - no staging, no extra barrier.
    if (!NATY) {
      constexpr int EPC = 16 / sizeof(T);
      const int cpr = F / EPC, rpp = (WAVES * 64) / cpr, row0 = tid / cpr, cc = (tid % cpr) * EPC;
      const int passes = BM / rpp;
      const T *Yg = (const T *)(A.stash + A.sl.Y[l]) + (size_t)m0 * F;
      if ((WAVES * 64) % cpr != 0) {     // F = 192: a row of chunks does not divide the workgroup - walk chunk ids instead
        for (int id = tid; id < BM * cpr; id += WAVES * 64) {
          const int row = id / cpr, c = (id % cpr) * EPC;
          *(u32x4 *)(ACT + (size_t)row * LDA + c) = stash_load((const u32x4 *)(Yg + (size_t)row * F + c));
        }
      } else if ((passes & 7) == 0) {    // F >= 256: eight loads in flight per thread, no branch around them
        for (int i0 = 0; i0 < passes; i0 += 8) {
          u32x4 v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = stash_load((const u32x4 *)(Yg + (size_t)(row0 + (i0 + i) * rpp) * F + cc));
#pragma unroll
          for (int i = 0; i < 8; ++i) *(u32x4 *)(ACT + (size_t)(row0 + (i0 + i) * rpp) * LDA + cc) = v[i];
        }
      } else {
        for (int i = 0; i < passes; ++i)
          *(u32x4 *)(ACT + (size_t)(row0 + i * rpp) * LDA + cc) = stash_load((const u32x4 *)(Yg + (size_t)(row0 + i * rpp) * F + cc));
      }
      __syncthreads();
    }
    if (wave_on) {
      const float w0 = (l == 0) ? 30.f : 1.f;
      const float unscale = A.prescaled ? 6.283185307179586f / w0 : 1.f;
      const float e2 = g.act == BN_ACT_SIN ? -w0 * w0 : 0.f;      // dD/dz = -w0^2 sin(w0 z) = -w0^2 y (0 for ReLU)
      const float dscale = (g.act == BN_ACT_SIN) ? w0 : 1.f;      // the 16-bit modes stash the unscaled derivative (DTile)
      const char *Ds = A.stash + A.sl.D[l] + (size_t)tile * dtile_bytes<DK>(BM, F);
      const T *As = (const T *)(A.stash + A.sl.adj_a[l]) + (size_t)tile * BM * F;
      const T *Yn = (const T *)(A.stash + A.sl.Y[l]) + (size_t)tile * BM * F;        // native-order y_l (16-bit modes)
      typename Elem<T>::wide *Zs = (typename Elem<T>::wide *)(A.stash + A.sl.adj_zbar[l]) + (size_t)tile * BM * F;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        u32x4 araw[PRE ? MT : 1][2], yraw[PRE ? MT : 1][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (PRE) {
            if (mt % ADJ_GRP == 0) {     // the pieces of the next ADJ_GRP point tiles together
#pragma unroll
              for (int q = mt; q < mt + ADJ_GRP && q < MT; ++q)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                  araw[q][gp] = stash_load((const u32x4 *)(As + native_off8<MT, NT>(wave, nt, q, gp, lane)));
                  yraw[q][gp] = stash_load((const u32x4 *)(Yn + native_off8<MT, NT>(wave, nt, q, gp, lane)));
                }
            }
          }
          const int m = mt * 32 + r;
          DPiece<DK> pc;
          if constexpr (PRE) pc = dpre[nt][mt];
          else pc = dpiece_load<DK>(Ds + dpiece_off<DK, MT, NT>(wave, nt, mt, lane));
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            const int n0 = ncol0 + nt * 32 + 16 * gp + 4 * h;
            float dv[8], av[8], zb[8], db[8], yv[8];
            dpiece_get<DK>(pc, gp, g.act, dscale, dv);
            if constexpr (PRE) {
              const auto aq = __builtin_bit_cast(typename Elem<T>::frag, araw[mt][gp]), yq = __builtin_bit_cast(typename Elem<T>::frag, yraw[mt][gp]);
#pragma unroll
              for (int e = 0; e < 8; ++e) { av[e] = (float)aq[e]; yv[e] = (float)yq[e]; }
            } else {
            ld8(As + native_off8<MT, NT>(wave, nt, mt, gp, lane), av);
            if (NATY) {
              ld8(Yn + native_off8<MT, NT>(wave, nt, mt, gp, lane), yv);
            } else {
              const vec4 ya = *(const vec4 *)(ACT + (size_t)m * LDA + n0), yb = *(const vec4 *)(ACT + (size_t)m * LDA + n0 + 8);
#pragma unroll
              for (int e = 0; e < 4; ++e) { yv[e] = (float)ya[e]; yv[4 + e] = (float)yb[e]; }
            }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) db[e] = acc[nt][mt][8 * gp + e] * unscale;
#pragma unroll
            for (int e = 0; e < 8; ++e) zb[e] = e2 * yv[e] * db[e] * av[e];
            if (TRACK) {
#pragma unroll
              for (int e = 0; e < 8; ++e) { const float az = fabsf(zb[e]); zmax = (az < 3.0e38f && az > zmax) ? az : zmax; }
            }
            st8(Zs + native_off8<MT, NT>(wave, nt, mt, gp, lane), zb);
            *(vec4 *)(ACT + (size_t)m * LDA + n0) = to_vec4(T(), db[0] * dv[0], db[1] * dv[1], db[2] * dv[2], db[3] * dv[3]);
            *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) = to_vec4(T(), db[4] * dv[4], db[5] * dv[5], db[6] * dv[6], db[7] * dv[7]);
          }
        }
      }
    }
    __syncthreads();
  }
  if (TRACK && A.amax) {   // ONE atomic per workgroup (non-negative floats order like their bit patterns); true scale = stored / gs
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o));
    float *red = (float *)PE;            // the encoding tile is free after the last layer (the loop ends on a barrier)
    if (lane == 0) red[wave] = zmax;
    __syncthreads();
    if (tid == 0) {
      float m = 0.f;
      for (int w = 0; w < WAVES; ++w) m = fmaxf(m, red[w]);
      if (m > 0.f) atomicMax((unsigned int *)A.amax + 2, __float_as_uint(m / gs));
    }
  }
  // abar_L (the tile the loop leaves in LDS) has no later GEMM to ride in
  tile_to_global<T>(ACT, LDA, (T *)(A.stash + A.sl.adj_abar[g.L]) + (size_t)m0 * F, F, BM, F);

  // ---------------------------------------------------------------- sbar = (w_sigma . abar_L) s'(1-s')
  {
    constexpr int TPR = (WAVES * 64) / BM;
    const int m = tid / TPR, q = tid % TPR;
    float ds = 0.f;
    const T *row = ACT + (size_t)m * LDA;
    for (int c8 = q; c8 < F / 8; c8 += TPR) {
      const typename Elem<T>::frag v = lds_frag<T>(row + c8 * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) ds += (float)v[j] * A.p.sigma_w[c8 * 8 + j];
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) ds += __shfl_xor(ds, o);
    if (q == 0) {
      const float sp = ((const float *)(A.stash + A.sl.sprime))[m0 + m];
      ((float *)(A.stash + A.sl.sbar))[m0 + m] = ds * (1.f - sp) / gs;      // ds = w_sigma . abar'_L
    }
  }
}

template <typename T, int MT, int NT, int WAVES, bool D16> static int launch_adjbwd_k(const AdjBwdArgs &a, int64_t tiles, hipStream_t st) {
  constexpr int BM = MT * 32;
  const size_t lds = ((size_t)BM * (a.g.F + Elem<T>::kPad) + (size_t)BM * (a.g.KP + Elem<T>::kPad)) * sizeof(T);
  if (int e = bn_configure_lds((const void *)field_adjbwd_kernel<T, MT, NT, WAVES, D16>, lds, "field_adjbwd")) return e;
  BnProfScope prof_(BN_K_ADJBWD, st);
  field_adjbwd_kernel<T, MT, NT, WAVES, D16><<<dim3((unsigned)tiles), WAVES * 64, lds, st>>>(a);
  BN_LAUNCH_CHECK("field_adjbwd");
  return 0;
}
template <typename T, int MT, int NT, int WAVES> static int launch_adjbwd(const AdjBwdArgs &a, int64_t tiles, hipStream_t st) {
  if constexpr (std::is_same<T, f16>::value) {
    if (a.g.dsz == 2) return launch_adjbwd_k<T, MT, NT, WAVES, true>(a, tiles, st);     // fp16 derivative stash (analytic normals)
  }
  return launch_adjbwd_k<T, MT, NT, WAVES, false>(a, tiles, st);
}

// called by bn_field_backward() before the primal backward chain when desc->normal_an is set
int bn_field_adjoint_backward(const bn_field_desc *desc, const bn_field_params *params, const void *packed, const bn_points *pts,
                              const float *d_out, void *stash, void *stream) {
  AdjBwdArgs a;
  if (int e = bn_make_geom(desc, &a.g)) return e;
  a.p = *params; a.packed = packed; a.pts = *pts; a.d_out = d_out; a.stash = (char *)stash;
  bn_make_packed_layout(a.g, &a.pl);
  a.prescaled = bn_half(desc->dtype) && desc->act == BN_ACT_SIN;
  const int BM = a.g.BM;
  bn_make_stash_layout(a.g, pts->n_points, BM, bn_esize(desc->dtype), &a.sl);
  a.amax = desc->dtype == BN_F16 ? (float *)((char *)stash + a.sl.gscale) : nullptr;
  const int64_t tiles = ceil_div64(pts->n_points, BM);
  hipStream_t st = (hipStream_t)stream;
  BN_DISPATCH_TILE(desc->dtype, a.g, launch_adjbwd, (a, tiles, st));
}
