// Analytic normals: normal_an = -l2_normalize(d sigma / d xyz)  (calc_normals, models/spsbrdfnerf.py:648-660, :713-716).
// The reference recomputes the trunk and runs autograd; here the gradient is the explicit adjoint chain (SURVEY.md
// section 8 row a8, validated against autograd in fp64 by the oracle tests):
//   a_L = sigmoid(sigma_raw) * w_sigma ;  for l = L-1 .. 0:  delta_l = a_{l+1} (.) D_l ,  [g_PE ; a_l] += W_l^T delta_l
//   d sigma/d x_c = sum_k f_k ( cos(f_k x_c) g_PE[sin,k,c] - sin(f_k x_c) g_PE[cos,k,c] ),  f_k = 2^k
// The chain is linear in s' = sigmoid(sigma_raw), a per-point scalar that reaches 1e-5 and below in empty space: it is
// carried OUTSIDE the 16-bit chain (the chain runs on a'_l = a_l / s', i.e. seeds with w_sigma alone; the fp32 epilogue
// multiplies the finished gradient by s').  Same function; the 16-bit operands keep their precision wherever the density
// is low (fp16 would otherwise work on subnormals there), and the backward of this chain (field_adjbwd.hip) scales its
// seeds by s' instead - which cancels the 1/|g| of the normalisation, so its operands lose the tiny factor as well.
// It reuses the backward-chain tiling: delta tiles live in LDS, W_l^T streams from L2 in packed fragment order, D_l
// comes from the forward's stash.  The two PE-part products (l = skip and l = 0; 64 x BM outputs) are spread over all
// 8 waves (one 32x32 tile each) and accumulated in an fp32 LDS image.
#include "field_kernels.h"

struct AdjArgs {
  FieldGeom g;
  bn_field_params p;
  PackedLayout pl;
  StashLayout sl;
  const void *packed;
  bn_points pts;
  float *out;       // [M][C]: channels ch_normal_an..+3 are written
  float *grad_x;    // optional [M][3]: raw d sigma / d xyz
  const char *stash;
  int keep;         // 1: training - also stash delta_l (row-major), a_{l+1} (native), sigmoid(s_raw), grad_x
};

template <typename T, int MT, int NT, int WAVES, bool KEEP, bool D16>
__global__ __launch_bounds__(WAVES * 64, 2) void field_adjoint_kernel(const AdjArgs A) {
  typedef typename Elem<T>::vec4 vec4;
  typedef typename DKind<T, D16>::type DK;     // kind of the derivative stash (field_kernels.h)
  constexpr int BM = MT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FieldGeom &g = A.g;
  const int F = g.F, LDA = F + Elem<T>::kPad, KSF = F / 16, P = g.P;
  T *ACT = (T *)smem;
  float *GP = (float *)(ACT + (size_t)BM * LDA);     // [BM][P] fp32: gradient w.r.t. the positional encoding
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int64_t tile = blockIdx.x, m0 = tile * BM, M = A.pts.n_points;
  const T *packed = (const T *)A.packed;
  const int ncol0 = wave * 32 * NT;
  const bool wave_on = ncol0 < F;
  const float *sraw = (const float *)(A.stash + A.sl.sraw);
  constexpr int DP = BwdDepth<T>::value;
  constexpr int NPRE = std::is_same<DK, DK8>::value ? NT : 1;    // D pieces fetched ahead of the layer's GEMM (field_bwd.hip)
  auto dscale = [&](int lo) { return (g.act == BN_ACT_SIN && lo == 0) ? 30.f : 1.f; };   // w0 of layer lo (16-bit modes: unscaled stash)

  for (int i = tid; i < BM * P; i += WAVES * 64) GP[i] = 0.f;
  char *wstash = const_cast<char *>(A.stash);
  constexpr bool keep = KEEP;   // training: stash delta_l / a_l for the backward of this chain
  if (keep && tid < BM) ((float *)(wstash + A.sl.sprime))[m0 + tid] = sigmoid_f(sraw[m0 + tid]);

  // delta_{L-1} = (sigmoid(s_raw) w_sigma) (.) D_{L-1}
  if (wave_on) {
    const char *Ds = A.stash + A.sl.D[g.L - 1] + (size_t)tile * dtile_bytes<DK>(BM, F);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const DPiece<DK> pc = dpiece_load<DK>(Ds + dpiece_off<DK, MT, NT>(wave, nt, mt, lane));
        const int m = mt * 32 + r;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int n0 = ncol0 + nt * 32 + 16 * gp + 4 * h;
          const f32x4 wa = *(const f32x4 *)(A.p.sigma_w + n0), wb = *(const f32x4 *)(A.p.sigma_w + n0 + 8);
          float dv[8];
          dpiece_get<DK>(pc, gp, g.act, dscale(g.L - 1), dv);
          float av[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { av[e] = wa[e]; av[4 + e] = wb[e]; }      // a'_L = w_sigma (s' is applied at the end)
          if (keep) st8((T *)(wstash + A.sl.adj_a[g.L - 1]) + (size_t)tile * BM * F + native_off8<MT, NT>(wave, nt, mt, gp, lane), av);
          *(vec4 *)(ACT + (size_t)m * LDA + n0) = to_vec4(T(), av[0] * dv[0], av[1] * dv[1], av[2] * dv[2], av[3] * dv[3]);
          *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) = to_vec4(T(), av[4] * dv[4], av[5] * dv[5], av[6] * dv[6], av[7] * dv[7]);
        }
      }
  }
  __syncthreads();

  f32x16 acc[NT][MT];
  const int n_on = F / (32 * NT) < WAVES ? F / (32 * NT) : WAVES;
  // the row-major stash copy of delta_l (the tile the trunk GEMM reads) rides inside that GEMM when the shape fits
  const bool ride = NT == 2 ? true : tile_copy_exact(F, n_on, WAVES);
  for (int l = g.L - 1; l >= 0; --l) {
    T *ddst = (T *)(wstash + A.sl.adj_delta[l]) + (size_t)m0 * F;
    if (keep && (!ride || l == 0)) tile_to_global<T>(ACT, LDA, ddst, F, BM, F);
    // ACT holds delta_l.  PE-part product (only where the layer reads the encoding): wave -> (p-tile, m-tile)
    if (l == 0 || l == g.skip) {
      const int ptile = wave & 1, mtile = wave >> 1;
      if (ptile * 32 < g.KP && mtile < MT) {
        f32x16 pacc[1][1];
        zero_acc<1, 1>(pacc);
        const size_t off = A.pl.bwd_pe[l == 0 ? 0 : 1] + (size_t)ptile * KSF * 512;
        gemm_seg<T, 1, 1, DP>(pacc, packed + off, KSF, ACT + (size_t)mtile * 32 * LDA, LDA, lane);
        const int m = mtile * 32 + r;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int p0 = ptile * 32 + 8 * gq + 4 * h;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (p0 + e < P) GP[m * P + p0 + e] += pacc[0][0][4 * gq + e];
        }
      }
    }
    if (l == 0) break;
    zero_acc<MT, NT>(acc);
    // D_{l-1} in accumulator order (DTile pieces), fetched ahead of the GEMM and of the stash stores riding in it
    const char *Ds = A.stash + A.sl.D[l - 1] + (size_t)tile * dtile_bytes<DK>(BM, F);
    DPiece<DK> dpre[NT][MT];
    if (wave_on) {
#pragma unroll
      for (int nt = 0; nt < NPRE; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dpre[nt][mt] = dpiece_load<DK>(Ds + dpiece_off<DK, MT, NT>(wave, nt, mt, lane));
      const T *w_t = packed + A.pl.bwd_trunk[l] + (size_t)(ncol0 / 32) * KSF * 512;
      if (keep && ride) {
        TileCopyExact<T> dcopy(ACT, LDA, ddst, F, F, tid, WAVES * 64);
        gemm_full32<T, MT, NT, DP, NT == 2>(acc, w_t, KSF, ACT, LDA, lane, dcopy);
      } else {
#ifndef BN_ADJ_LOOP
        gemm_full<T, MT, NT, DP, NT == 2>(acc, w_t, KSF, ACT, LDA, lane);      // (inference: no riding copy)
#else
        gemm_seg<T, MT, NT, DP>(acc, w_t, KSF, ACT, LDA, lane);
#endif
      }
#pragma unroll
      for (int nt = NPRE; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dpre[nt][mt] = dpiece_load<DK>(Ds + dpiece_off<DK, MT, NT>(wave, nt, mt, lane));
    }
    __syncthreads();
    if (wave_on) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m = mt * 32 + r;
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            const int n0 = ncol0 + nt * 32 + 16 * gp + 4 * h;
            float dv[8];
            dpiece_get<DK>(dpre[nt][mt], gp, g.act, dscale(l - 1), dv);
            if (keep) {
              float av[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) av[e] = acc[nt][mt][8 * gp + e];
              st8((T *)(wstash + A.sl.adj_a[l - 1]) + (size_t)tile * BM * F + native_off8<MT, NT>(wave, nt, mt, gp, lane), av);
            }
            *(vec4 *)(ACT + (size_t)m * LDA + n0) =
                to_vec4(T(), acc[nt][mt][8 * gp] * dv[0], acc[nt][mt][8 * gp + 1] * dv[1], acc[nt][mt][8 * gp + 2] * dv[2], acc[nt][mt][8 * gp + 3] * dv[3]);
            *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) =
                to_vec4(T(), acc[nt][mt][8 * gp + 4] * dv[4], acc[nt][mt][8 * gp + 5] * dv[5], acc[nt][mt][8 * gp + 6] * dv[6], acc[nt][mt][8 * gp + 7] * dv[7]);
          }
        }
    }
    __syncthreads();
  }
  __syncthreads();

  // chain through the positional encoding (fp32, accurate sincos in both precision modes) and normalise
  if (tid < BM) {
    const int64_t gm = m0 + tid;
    if (gm < M) {
      float x[3];
      if (A.pts.xyz) {
        x[0] = A.pts.xyz[gm * 3]; x[1] = A.pts.xyz[gm * 3 + 1]; x[2] = A.pts.xyz[gm * 3 + 2];
      } else {
        const float *rr = A.pts.rays + (gm / A.pts.n_samples) * A.pts.ray_stride;
        const float zz = A.pts.z[gm];
        x[0] = rr[0] + rr[3] * zz; x[1] = rr[1] + rr[4] * zz; x[2] = rr[2] + rr[5] * zz;
      }
      const float *gp = GP + tid * P;
      float gx[3] = {0.f, 0.f, 0.f};
      if (g.pe_freqs > 0) {
        for (int k = 0; k < g.pe_freqs; ++k) {
          const float f = (float)(1 << k);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            float s, co;
            sincos_cw(f * x[c], s, co);
            gx[c] += f * (co * gp[6 * k + c] - s * gp[6 * k + 3 + c]);
          }
        }
      } else {
        gx[0] = gp[0]; gx[1] = gp[1]; gx[2] = gp[2];
      }
      const float spm = sigmoid_f(sraw[gm]);      // the chain ran on a' = a / s': back to d sigma / d xyz
      gx[0] *= spm; gx[1] *= spm; gx[2] *= spm;
      if (A.grad_x) { A.grad_x[gm * 3] = gx[0]; A.grad_x[gm * 3 + 1] = gx[1]; A.grad_x[gm * 3 + 2] = gx[2]; }
      if (keep) { float *gs = (float *)(wstash + A.sl.gradx) + gm * 4; gs[0] = gx[0]; gs[1] = gx[1]; gs[2] = gx[2]; gs[3] = 0.f; }
      const float inv = -1.f / sqrtf(fmaxf(gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2], 1.1920928955078125e-07f));
      float *o = A.out + gm * g.C + g.ch_normal_an;
      o[0] = gx[0] * inv; o[1] = gx[1] * inv; o[2] = gx[2] * inv;
    }
  }
}

template <typename T, int MT, int NT, int WAVES, bool KEEP, bool D16> static int launch_adj_k(const AdjArgs &a, int64_t tiles, hipStream_t st);
template <typename T, int MT, int NT, int WAVES> static int launch_adj(const AdjArgs &a, int64_t tiles, hipStream_t st) {
  if constexpr (std::is_same<T, f16>::value) {      // fp16 mode: the derivative stash of a model with analytic normals is fp16 (DK16)
    if (a.g.dsz == 2)
      return a.keep ? launch_adj_k<T, MT, NT, WAVES, true, true>(a, tiles, st) : launch_adj_k<T, MT, NT, WAVES, false, true>(a, tiles, st);
  }
  return a.keep ? launch_adj_k<T, MT, NT, WAVES, true, false>(a, tiles, st) : launch_adj_k<T, MT, NT, WAVES, false, false>(a, tiles, st);
}
template <typename T, int MT, int NT, int WAVES, bool KEEP, bool D16> static int launch_adj_k(const AdjArgs &a, int64_t tiles, hipStream_t st) {
  constexpr int BM = MT * 32;
  const size_t lds = (size_t)BM * (a.g.F + Elem<T>::kPad) * sizeof(T) + (size_t)BM * a.g.P * sizeof(float);
  if (int e = bn_configure_lds((const void *)field_adjoint_kernel<T, MT, NT, WAVES, KEEP, D16>, lds, "field_normals")) return e;
  BnProfScope prof_(BN_K_ADJOINT, st);
  field_adjoint_kernel<T, MT, NT, WAVES, KEEP, D16><<<dim3((unsigned)tiles), WAVES * 64, lds, st>>>(a);
  BN_LAUNCH_CHECK("field_normals");
  return 0;
}

int bn_field_normals_impl(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                          const bn_points *pts, const void *stash, float *out, float *grad_x, int keep, void *stream) {
  AdjArgs a;
  if (int e = bn_make_geom(desc, &a.g)) return e;
  BN_REQUIRE(desc->normal_an && a.g.ch_normal_an >= 0, "field_normals: desc.normal_an is not set");
  BN_REQUIRE(desc->act == BN_ACT_SIN || desc->act == BN_ACT_RELU, "field_normals: bad activation");
  BN_REQUIRE(pts && pts->n_points > 0 && packed && stash && out, "field_normals: null argument");
  a.p = *params; a.packed = packed; a.pts = *pts; a.out = out; a.grad_x = grad_x; a.stash = (const char *)stash; a.keep = keep;
  bn_make_packed_layout(a.g, &a.pl);
  const int BM = a.g.BM;
  BN_REQUIRE(pts->point_offset >= 0 && pts->point_offset % BM == 0, "field_normals: point_offset must be a multiple of %d", BM);
  bn_stash_layout_at(a.g, pts, BM, bn_esize(desc->dtype), &a.sl);
  a.out = out + pts->point_offset * a.g.C;
  if (grad_x) a.grad_x = grad_x + pts->point_offset * 3;
  const int64_t tiles = ceil_div64(pts->n_points, BM);
  hipStream_t st = (hipStream_t)stream;
  BN_DISPATCH_TILE(desc->dtype, a.g, launch_adj, (a, tiles, st));
}

extern "C" int bn_field_normals(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                                const bn_points *pts, const void *stash, float *out, float *grad_x, int32_t keep_for_backward,
                                void *stream) {
  return bn_field_normals_impl(desc, params, packed, pts, stash, out, grad_x, keep_for_backward, stream);
}
