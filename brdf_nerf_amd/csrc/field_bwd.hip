// Fused field-MLP backward for gfx950 (parameter gradients).
//   1. field_bwd_kernel: the dX chain - one workgroup per tile of BM points walks the network in reverse with
//      the gradient tile resident in LDS (same tiling as the forward; weights = pre-packed W^T fragments),
//      writing each layer's pre-activation gradient dZ_l row-major for step 2.
//   2. wgrad_kernel: dW_l[n][k] += sum_m dZ_l[m][n] X_l[m][k] as MFMA GEMMs over the stashed activations
//      (bf16: LDS tiles + ds_read_b64_tr_b16 transposing reads; fp32: plain ds_read_b32), split over point
//      chunks with fp32 atomics; bias gradients as column sums in the same kernel.
//   3. skinny_wgrad_kernel: the <= 4-row matrices (sigma head, learned normal, second head layers).
// Autograd counterpart in the reference: loss.backward() through SpSBRDFNeRF.forward (models/spsbrdfnerf.py:662-757).
#include "field_kernels.h"

int bn_field_adjoint_backward(const bn_field_desc *desc, const bn_field_params *params, const void *packed, const bn_points *pts,
                              const float *d_out, void *stash, void *stream);

BN_PH_DEFINE_READER(bn_debug_phase_read_bwd)
BN_CLK_DEFINE(bn_debug_clock_read_bwd)

struct BwdArgs {
  FieldGeom g;
  bn_field_desc d;
  bn_field_params p;
  PackedLayout pl;
  StashLayout sl;
  const void *packed;
  int64_t M;
  const float *out, *d_out;
  char *stash;
  int an;   // analytic normals in the graph: add the adjoint chain's sbar / zbar_l (field_adjbwd.hip)
  const float *amax;   // fp16 mode: [0] max |d pre-activation|, [1] max |gbar_PE|, [2] max |zbar_l| (loss scaling, common.h); else nullptr
};

// sigmoid (BN_HEAD_BETA: softplus) output y and dL/dy of head `hd`, channel c, recovered from the forward's rescaled output.
__device__ __forceinline__ void head_y_dy(int kind, int nout, const float *o, const float *dgo, int c, float &y, float &dy) {
  if (kind == BN_HEAD_PLAIN || kind == BN_HEAD_BETA) { y = o[c]; dy = dgo[c]; }
  else if (kind == BN_HEAD_HAPKE_THETA) { y = o[0] * (1.f / 0.52359877559829887f); dy = dgo[0] * 0.52359877559829887f; }
  else {
    const float ov = nout == 1 ? o[0] : o[c];
    const float dv = nout == 1 ? dgo[0] + dgo[1] + dgo[2] : dgo[c];
    if (kind == BN_HEAD_RPV_K) { y = (ov - 1.f) * 0.5f + 0.5f; dy = 2.f * dv; }
    else if (kind == BN_HEAD_RPV_THETA) { y = ov * 0.5f + 0.5f; dy = 2.f * dv; }
    else { y = ov; dy = dv; }
  }
}

// d L / d (pre-activation) of the small outputs of point gm: the <= 3 pre-sigmoid values of every head, sigma_raw and
// the learned-normal vector (shared by the chain kernel's prologue and the fp16 loss-scale reduction).
__device__ __forceinline__ void bwd_dpre(const BwdArgs &A, int64_t gm, float (&dph)[BN_DPH], float (&dpt)[4]) {
  const FieldGeom &g = A.g;
#pragma unroll
  for (int i = 0; i < BN_DPH; ++i) dph[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) dpt[i] = 0.f;
  if (gm >= A.M) return;
  const float *o = A.out + gm * g.C, *dgo = A.d_out + gm * g.C;
  for (int hd = 0; hd < g.n_heads; ++hd) {
    const int nout = A.d.head_out[hd], kind = A.d.head_kind[hd];
    for (int c = 0; c < nout; ++c) {
      float y, dy;
      head_y_dy(kind, nout, o + g.head_col[hd], dgo + g.head_col[hd], c, y, dy);
      // d sigmoid = y (1 - y);  d softplus(x) = sigmoid(x) = 1 - exp(-softplus(x))
      dph[hd * 3 + c] = kind == BN_HEAD_BETA ? dy * -expm1f(-y) : dy * y * (1.f - y);
    }
  }
  const float sraw = ((const float *)(A.stash + A.sl.sraw))[gm];
  dpt[0] = dgo[3] * sigmoid_f(sraw);
  if (A.an) dpt[0] += ((const float *)(A.stash + A.sl.sbar))[gm];
  if (g.ch_normal_lr >= 0) {
    const float *v = (const float *)(A.stash + A.sl.nraw) + gm * 4;
    const float *dn = dgo + g.ch_normal_lr;
    const float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const float eps = 1.1920928955078125e-07f;
    const float inv = 1.f / sqrtf(fmaxf(n2, eps));
    // out = -v * inv ; inv depends on v only when n2 > eps (torch.maximum passes the gradient to the larger)
    const float vd = v[0] * dn[0] + v[1] * dn[1] + v[2] * dn[2];
    const float k = n2 > eps ? vd * inv * inv * inv : 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) dpt[1 + i] = -(dn[i] * inv - v[i] * k);
  }
}

// fp16 loss scaling: amax[which] = max over the launch of the finite |seed gradients| (float bits; non-negative floats order
// like their bit patterns).  which = 0: the primal chain's seeds (bwd_dpre); which = 1: the analytic-normal double
// backward's seeds gbar_PE = J_PE(x) gbar, bounded by 2^(pe_freqs-1) |gbar| (field_adjbwd.hip).
template <int WHICH> __global__ __launch_bounds__(256) void grad_amax_kernel(const BwdArgs A, float *amax) {
  const FieldGeom &g = A.g;
  float mx = 0.f;
  for (int64_t gm = (int64_t)blockIdx.x * 256 + threadIdx.x; gm < A.M; gm += (int64_t)gridDim.x * 256) {
    if (WHICH == 0) {
      float dph[BN_DPH], dpt[4];
      bwd_dpre(A, gm, dph, dpt);
#pragma unroll
      for (int i = 0; i < BN_DPH; ++i) { const float a = fabsf(dph[i]); mx = (a < 3.0e38f && a > mx) ? a : mx; }
#pragma unroll
      for (int i = 0; i < 4; ++i) { const float a = fabsf(dpt[i]); mx = (a < 3.0e38f && a > mx) ? a : mx; }
    } else {
      const float *gx = (const float *)(A.stash + A.sl.gradx) + gm * 4;
      const float *dn = A.d_out + gm * g.C + g.ch_normal_an;
      const float n2 = gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2];
      const float eps = 1.1920928955078125e-07f;
      const float inv = 1.f / sqrtf(fmaxf(n2, eps));
      const float gd = gx[0] * dn[0] + gx[1] * dn[1] + gx[2] * dn[2];
      const float k = n2 > eps ? gd * inv * inv * inv : 0.f;
      const float fmx = g.pe_freqs > 0 ? (float)(1 << (g.pe_freqs - 1)) : 1.f;
      const float spm = ((const float *)(A.stash + A.sl.sprime))[gm];       // the chain's seeds are s' gbar (field_adjbwd.hip)
#pragma unroll
      for (int c = 0; c < 3; ++c) { const float a = fabsf(dn[c] * inv - gx[c] * k) * spm * fmx; mx = (a < 3.0e38f && a > mx) ? a : mx; }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax((unsigned int *)amax + WHICH, __float_as_uint(mx));
}

// --beta: d loss / d t_embed of every point = dG_beta[m][:] W_t  (W_t = beta_from_xyz.0.weight[:, F:], [H2][TD] fp32).
// One wave per point over the H2 hidden columns of head 1 inside pass 0's dG rows; HBM-bound on M x H2 elements.
template <typename T>
__global__ __launch_bounds__(256) void head_xin_grad_kernel(const T *__restrict__ dG, int ldg, int col0, int H2, const float *__restrict__ Wt,
                                                            int64_t ldw, int TD, int64_t M, const float *amax, float *__restrict__ d_t) {
  __shared__ float W[256 * 16];
  for (int i = threadIdx.x; i < H2 * 16; i += 256) W[i] = (i & 15) < TD ? Wt[(int64_t)(i >> 4) * ldw + (i & 15)] : 0.f;
  __syncthreads();
  const float osc = 1.f / chain_scale(amax);           // fp16 loss scaling carried by dG (1 in the other modes)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t gm = (int64_t)blockIdx.x * 4 + wave; gm < M; gm += (int64_t)gridDim.x * 4) {
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
    const T *row = dG + gm * ldg + col0;
    for (int j = lane; j < H2; j += 64) {
      const float d = (float)row[j];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] += d * W[j * 16 + c];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) acc[c] += __shfl_xor(acc[c], o);
    }
    if (lane < TD) {
      float v = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) v = lane == c ? acc[c] : v;
      d_t[gm * TD + lane] = v * osc;
    }
  }
}

template <typename T, int MT, int NTW>
__device__ __forceinline__ void bwd_head_dG(const BwdArgs &A, int p, T *ACT, const float *DPH, int64_t /*m0*/, int64_t tile) {
  typedef typename Elem<T>::vec4 vec4;
  constexpr int BM = MT * 32;
  const FieldGeom &g = A.g;
  const int F = g.F, LDA = F + Elem<T>::kPad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int N = g.pass_N[p];
  const int pc0 = wave * 32 * NTW;
  if (pc0 >= N) return;
  const int hl = pc0 / g.H2, hd = 2 * p + hl;
  const int nout = A.d.head_out[hd];
  const float *w2 = A.p.head_w2[hd];
  const T *DGs = (const T *)(A.stash + A.sl.DG[p]) + (size_t)tile * BM * F;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int n0 = pc0 + nt * 32 + 16 * gp + 4 * h;
      const int nl = n0 - hl * g.H2;
      f32x4 wa[3], wb[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        wa[c] = c < nout ? *(const f32x4 *)(w2 + (size_t)c * g.H2 + nl) : f32x4{0, 0, 0, 0};
        wb[c] = c < nout ? *(const f32x4 *)(w2 + (size_t)c * g.H2 + nl + 8) : f32x4{0, 0, 0, 0};
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 32 + r;
        const float d0 = DPH[m * BN_DPH + hd * 3 + 0], d1 = DPH[m * BN_DPH + hd * 3 + 1], d2 = DPH[m * BN_DPH + hd * 3 + 2];
        float dg[8];
        ld8(DGs + native_off8<MT, NTW>(wave, nt, mt, gp, lane), dg);
        float va[4], vb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          va[e] = (d0 * wa[0][e] + d1 * wa[1][e] + d2 * wa[2][e]) * dg[e];
          vb[e] = (d0 * wb[0][e] + d1 * wb[1][e] + d2 * wb[2][e]) * dg[4 + e];
        }
        *(vec4 *)(ACT + (size_t)m * LDA + n0) = to_vec4(T(), va[0], va[1], va[2], va[3]);
        *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) = to_vec4(T(), vb[0], vb[1], vb[2], vb[3]);
      }
    }
}

template <typename T, int MT, int NT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void field_bwd_kernel(const BwdArgs A) {
  typedef typename Elem<T>::vec4 vec4;
  constexpr int BM = MT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FieldGeom &g = A.g;
  const int F = g.F, LDA = F + Elem<T>::kPad, KSF = F / 16;
  T *ACT = (T *)smem;
  float *DPH = (float *)(ACT + (size_t)BM * LDA);  // [BM][BN_DPH] head pre-sigmoid gradients
  float *DPT = DPH + BM * BN_DPH;                        // [BM][4]  (d sigma_raw, d normal_raw)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int64_t tile = blockIdx.x, m0 = tile * BM, M = A.M;
  const T *packed = (const T *)A.packed;
  BN_PH_DECL
  BN_CLK_BEGIN

  // ---------------------------------------------------------------- pre-activation gradients of the small outputs
  // fp16 mode: the chain runs on gradients scaled by S (a power of two, from the device-side maximum); the fp32 copies
  // kept for the skinny weight-gradient kernel stay unscaled, the 16-bit dZ_l / dG stashes carry S and the weight-gradient
  // kernel removes it from its fp32 sums.
  const float gs = chain_scale(A.amax);
  if (tid < BM) {
    const int m = tid;
    const int64_t gm = m0 + m;
    float dph[BN_DPH], dpt[4];
    bwd_dpre(A, gm, dph, dpt);
    float *sh = (float *)(A.stash + A.sl.dpre_head) + gm * BN_DPH;
    float *st = (float *)(A.stash + A.sl.dpre_trunk) + gm * 4;
#pragma unroll
    for (int i = 0; i < BN_DPH; ++i) { DPH[m * BN_DPH + i] = dph[i] * gs; sh[i] = dph[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { DPT[m * 4 + i] = dpt[i] * gs; st[i] = dpt[i]; }
  }
  __syncthreads();
  BN_PH(0)

  const int ncol0 = wave * 32 * NT;
  const bool wave_on = ncol0 < F;
  const int n_on = F / (32 * NT) < WAVES ? F / (32 * NT) : WAVES;   // waves that own output columns
  // stash copies ride inside the GEMMs when the shape fits (NT == 2 means F = 512: always, decided at compile time)
  const bool ride = NT == 2 ? true : tile_copy_exact(F, n_on, WAVES);
  f32x16 acc[NT][MT];
  zero_acc<MT, NT>(acc);
  const float zr = (A.an && A.amax) ? gs / grad_scale_from(A.amax + 1, BN_GS_TARGET_ADJ) : 1.f;

  // ---------------------------------------------------------------- heads: dG -> LDS, dFeats += W1^T dG
  for (int p = 0; p < g.n_pass; ++p) {
    if (g.pass_heads[p] == 2) bwd_head_dG<T, MT, NT>(A, p, ACT, DPH, m0, tile);
    else bwd_head_dG<T, MT, BN_SINGLE_HEAD_NTW(NT)>(A, p, ACT, DPH, m0, tile);
    BN_PH(1)
    __syncthreads();
    BN_PH(2)
    BN_PH(3)
    const int KSp = g.pass_N[p] / 16;
    tile_to_global<T>(ACT, LDA, (T *)(A.stash + A.sl.dG[p]) + (size_t)m0 * g.pass_N[p], g.pass_N[p], BM, g.pass_N[p]);
    if (wave_on) gemm_seg<T, MT, NT>(acc, packed + A.pl.bwd_head[p] + (size_t)(ncol0 / 32) * KSp * 512, KSp, ACT, LDA, lane);
    BN_PH(4)
    __syncthreads();
    BN_PH(5)
  }
  // dFeats -> LDS (+ stash below).  With fold_feats the head products above already are W_f^T W_1^T dG = dL/dY_{L-1}
  // (before the rank-1 terms and D): they stay in the accumulators for the top epilogue and the W_f^T GEMM is skipped.
  if (wave_on && !g.fold) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = ncol0 + nt * 32 + 8 * gq + 4 * h;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m = mt * 32 + r;
          *(vec4 *)(ACT + (size_t)m * LDA + n) =
              to_vec4(T(), acc[nt][mt][4 * gq], acc[nt][mt][4 * gq + 1], acc[nt][mt][4 * gq + 2], acc[nt][mt][4 * gq + 3]);
        }
      }
  }
  BN_PH(6)
  __syncthreads();
  BN_PH(7)
  BN_PH(8)

  // ---------------------------------------------------------------- trunk, top layer first
  for (int l = g.L; l >= 1; --l) {
    // l == L: dY_{L-1} = Wf^T dFeats + sigma/normal rank-1 terms; else dY_{l-1} = W_l^T dZ_l
    const bool folded_top = g.fold && l == g.L;   // the product is already in the accumulators
    if (!folded_top) zero_acc<MT, NT>(acc);
    const int lo = l - 1;  // layer whose pre-activation gradient is produced
    // D_lo = d act / d z of that layer, read back in accumulator order.  The loads are issued around the GEMM -
    // n-tile 0 before it (in flight while the MFMAs run), the others right after its last MFMA, when the weight and
    // activation fragment registers are free - so the epilogue does not sit on HBM latency.
    const T *Ds = (const T *)(A.stash + A.sl.D[lo]) + (size_t)tile * BM * F;
    typename Elem<T>::frag dpre[NT][2][MT];
    T *zdst = (T *)(A.stash + (l == g.L ? A.sl.dfeats : A.sl.dZ[l])) + (size_t)m0 * F;
    if (!ride && !folded_top) tile_to_global<T>(ACT, LDA, zdst, F, BM, F);
    if (wave_on) {
#pragma unroll
      for (int gp = 0; gp < 2; ++gp)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dpre[0][gp][mt] = stash_load((const typename Elem<T>::frag *)(Ds + native_off8<MT, NT>(wave, 0, mt, gp, lane)));
      const size_t off = (l == g.L ? A.pl.bwd_feats : A.pl.bwd_trunk[l]) + (size_t)(ncol0 / 32) * KSF * 512;
      // the row-major stash copy of the tile this GEMM reads (dFeats, then dZ_l) rides inside the GEMM
      if (folded_top) {
#ifdef BN_AB_BWD_NATIVE_DZ   // ablation (weight gradients wrong): no riding row-major dZ copy; dZ is stored in accumulator order
      } else if (true) {     // from the epilogue's registers instead - what a native-order dZ stash would cost the chain
        gemm_seg<T, MT, NT>(acc, packed + off, KSF, ACT, LDA, lane);
#endif
      } else if (ride) {
        TileCopyExact<T> zcopy(ACT, LDA, zdst, F, F, tid, WAVES * 64);
        gemm_seg<T, MT, NT>(acc, packed + off, KSF, ACT, LDA, lane, zcopy);
      } else {
        gemm_seg<T, MT, NT>(acc, packed + off, KSF, ACT, LDA, lane);
      }
#pragma unroll
      for (int nt = 1; nt < NT; ++nt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) dpre[nt][gp][mt] = stash_load((const typename Elem<T>::frag *)(Ds + native_off8<MT, NT>(wave, nt, mt, gp, lane)));
    }
    BN_PH(9)
    __syncthreads();
    BN_PH(10)
    if (wave_on) {
      const bool top = l == g.L, nlr = g.ch_normal_lr >= 0;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int n0 = ncol0 + nt * 32 + 16 * gp + 4 * h;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const int m = mt * 32 + r;
            float dv[8], v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { dv[e] = (float)dpre[nt][gp][mt][e]; v[e] = acc[nt][mt][8 * gp + e]; }
            if (top) {  // rank-1 terms of the sigma head and the learned-normal head
              const float ds = DPT[m * 4], a0 = DPT[m * 4 + 1], a1 = DPT[m * 4 + 2], a2 = DPT[m * 4 + 3];
#pragma unroll
              for (int half = 0; half < 2; ++half) {
                const int n = n0 + 8 * half;
                const f32x4 ws = *(const f32x4 *)(A.p.sigma_w + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * half + e] += ws[e] * ds;
                if (nlr) {
                  const f32x4 wn0 = *(const f32x4 *)(A.p.normal_w + n), wn1 = *(const f32x4 *)(A.p.normal_w + F + n),
                              wn2 = *(const f32x4 *)(A.p.normal_w + 2 * F + n);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[4 * half + e] += wn0[e] * a0 + wn1[e] * a1 + wn2[e] * a2;
                }
              }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= dv[e];
            if (A.an) {  // + dL/dz_l through D_l of the analytic-normal adjoint chain (stored at that chain's own scale)
              float zb[8];
              ld8((const typename Elem<T>::wide *)(A.stash + A.sl.adj_zbar[lo]) + (size_t)tile * BM * F + native_off8<MT, NT>(wave, nt, mt, gp, lane), zb);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += zb[e] * zr;
            }
            *(vec4 *)(ACT + (size_t)m * LDA + n0) = to_vec4(T(), v[0], v[1], v[2], v[3]);
            *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) = to_vec4(T(), v[4], v[5], v[6], v[7]);
#ifdef BN_AB_BWD_NATIVE_DZ
            st8((T *)(A.stash + A.sl.dZ[lo]) + (size_t)tile * BM * F + native_off8<MT, NT>(wave, nt, mt, gp, lane), v);
#endif
          }
        }
    }
    BN_PH(11)
    __syncthreads();
    BN_PH(12)
  }
#ifndef BN_AB_BWD_NATIVE_DZ
  tile_to_global<T>(ACT, LDA, (T *)(A.stash + A.sl.dZ[0]) + (size_t)m0 * F, F, BM, F);
#endif
  BN_PH(13)
#ifndef BN_PHASE_TIMING_WGRAD
  BN_PH_FLUSH
#endif
#ifndef BN_CLOCK_STAMP_WGRAD
  BN_CLK_END
#endif
}

// ------------------------------------------------------------------------------------------ weight gradients
// ---- deterministic accumulation (bn_set_deterministic(1)) ---------------------------------------------------------------------
// The weight-gradient kernels split the points over many workgroups that add their partial sums into the same fp32 output with
// atomics: the order of those additions - and with it the last bits of the gradient - changes from run to run.  In deterministic
// mode the workgroups that add into one output tile take TURNS in split order: a ticket per output tile (zeroed per call, in the
// stash) counts the splits that have added; split s waits for ticket == s, adds (the same atomics), fences, and passes the turn.
// Blocks are numbered split-major, and the hardware starts blocks in id order on every XCD, so the block a waiter depends on was
// always started before it: the smallest unfinished id never waits.  Jobs that add into the same matrix (the primal and the
// analytic-normal term of a trunk layer) go to separate, stream-ordered launches.  The spin is bounded; a timeout is reported
// through bn_device_faults() (bit 1) and the block proceeds.
__device__ unsigned int g_det_fault = 0u;
__device__ __forceinline__ void det_enter(unsigned int *ticket, unsigned int seq) {
  if (ticket == nullptr) return;
  if (threadIdx.x == 0) {
    unsigned int spins = 0;
    while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq) {
      __builtin_amdgcn_s_sleep(32);
      if (++spins > (1u << 25)) { g_det_fault = 1u; break; }   // ~ 30 s: never in a correct run
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void det_leave(unsigned int *ticket) {
  if (ticket == nullptr) return;
  __threadfence();       // this workgroup's additions are performed before the next one's turn
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
unsigned int bn_bwd_fault_read(hipStream_t st) {   // (bn_device_faults, field_fwd.hip)
  unsigned int v = 0u;
  if (hipStreamSynchronize(st) != hipSuccess) return 0x80000000u;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_det_fault), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) return 0x80000000u;
  return v;
}

struct WgradJob {
  const void *A;   // [Mpad][lda] T : gradient rows (dZ / dFeats / dG)
  const void *B;   // [Mpad][ldb] T : layer input rows (PE / Y / feats)
  float *C;        // [N][ldc] fp32, +=
  float *bias;     // [N] fp32, += column sums of A (nullable)
  int lda, ldb, ldc;
  int a_col0, b_col0;  // first column used in A / B
  int N, K;            // valid output extents (rows of C, cols of C)
  int scale_sel;       // fp16 loss scaling carried by A: 0 none, 1 the primal chain's (amax[0]), 2 the adjoint chain's (amax[1])
  int b_native;        // 16-bit modes: B is a layer-output stash in accumulator-native order (tiles of b_bm points, F columns:
  int b_bm, b_F;       // chunk (col/32, point/32 % (bm/32), (col%32)/16) = 64 lanes x 16 B, see native_off8); else row-major [Mpad][ldb]
  int b_bm_shift;      // log2(b_bm)
};
// 1 / (scale carried by the job's gradient operand): multiplies the fp32 sums before they are accumulated
__device__ __forceinline__ float wg_unscale(const float *amax, int sel) {
  if (amax == nullptr || sel == 0) return 1.f;
  return 1.f / (sel == 1 ? chain_scale(amax) : grad_scale_from(amax + 1, BN_GS_TARGET_ADJ));
}
#define BN_MAX_WGRAD_JOBS 44
struct WgradArgs {
  WgradJob job[BN_MAX_WGRAD_JOBS];
  int tile0[BN_MAX_WGRAD_JOBS + 1];  // prefix sum of 128x128 output tiles per job
  int n_jobs;
  int64_t Mpad;
  int m_per_block;                   // points per split (multiple of 32)
  const float *amax;                 // fp16 mode only (else nullptr): see wg_unscale
  unsigned int *tickets;             // deterministic mode: one turn counter per output tile of this launch (else nullptr)
};

#define WG_BK 32
template <typename T> struct WgTile;
template <> struct WgTile<float> { static constexpr int LD = 128 + 4; };

// fp32 parity path (the bf16 path is wgrad256_kernel below).
// 8-element MFMA fragment of the TRANSPOSED tile: element j <-> contraction index (point) m, fixed column `col`.
template <typename T> __device__ __forceinline__ typename Elem<T>::frag wg_frag(const T *tile, int mm, int col0, int lane);
template <> __device__ __forceinline__ f32x8 wg_frag<float>(const float *tile, int mm, int col0, int lane) {
  // fp32 MFMA j consumes element j of both operands with lane-half h as its k index: m = mm + 2j + h.
  constexpr int LD = WgTile<float>::LD;
  const int h = lane >> 5, r = lane & 31;
  f32x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = tile[(size_t)(mm + 2 * j + h) * LD + col0 + r];
  return f;
}

template <typename T> __global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs A) {
  constexpr int LD = WgTile<T>::LD;
  constexpr int EPC = 16 / sizeof(T);          // elements per 16-byte chunk
  constexpr int CPR = 128 / EPC;               // chunks per tile row
  constexpr int NCH = WG_BK * CPR / 256;       // chunks per thread per operand
  __shared__ __attribute__((aligned(16))) T sA[WG_BK * LD];
  __shared__ __attribute__((aligned(16))) T sB[WG_BK * LD];
  // which job / output tile
  int jb = 0;
  while (jb + 1 < A.n_jobs && (int)blockIdx.x >= A.tile0[jb + 1]) ++jb;
  const WgradJob &J = A.job[jb];
  const int t = blockIdx.x - A.tile0[jb];
  const int tiles_k = (J.K + 127) / 128;
  const int n0 = (t / tiles_k) * 128, k0 = (t % tiles_k) * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const int64_t mb = (int64_t)blockIdx.y * A.m_per_block;
  const int64_t me = mb + A.m_per_block < A.Mpad ? mb + A.m_per_block : A.Mpad;
  const T *gA = (const T *)J.A + J.a_col0 + n0;
  const T *gB = (const T *)J.B + J.b_col0 + k0;
  // columns beyond the valid extent are zero-filled (they lie inside the row for n, may not for k: PE has K=60<64)
  uint4 ra[NCH], rb[NCH];
  auto gload = [&](int64_t m) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = tid + c * 256, row = ch / CPR, cc = (ch % CPR) * EPC;
      ra[c] = (n0 + cc < J.N) ? *(const uint4 *)(gA + (m + row) * J.lda + cc) : uint4{0, 0, 0, 0};
      rb[c] = (k0 + cc < J.K) ? *(const uint4 *)(gB + (m + row) * J.ldb + cc) : uint4{0, 0, 0, 0};
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = tid + c * 256, row = ch / CPR, cc = (ch % CPR) * EPC;
      *(uint4 *)(sA + row * LD + cc) = ra[c];
      *(uint4 *)(sB + row * LD + cc) = rb[c];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  float bsum = 0.f;
  const bool do_bias = J.bias != nullptr && k0 == 0 && tid < 128;
  gload(mb);
  for (int64_t m = mb; m < me; m += WG_BK) {
    __syncthreads();
    sstore();
    __syncthreads();
    if (m + WG_BK < me) gload(m + WG_BK);
#pragma unroll
    for (int mm = 0; mm < WG_BK; mm += 16) {
      typename Elem<T>::frag fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = wg_frag<T>(sA, mm, wr * 64 + a * 32, lane);
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = wg_frag<T>(sB, mm, wc * 64 + b * 32, lane);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) mma32(acc[a][b], fa[a], fb[b]);
    }
    if (do_bias) {
#pragma unroll 8
      for (int row = 0; row < WG_BK; ++row) bsum += (float)sA[row * LD + tid];
    }
  }
  // C[n][k]: accumulator row index = n (A operand rows), column (lane&31) = k
  const int r = lane & 31, h = lane >> 5;
  unsigned int *ticket = A.tickets ? A.tickets + blockIdx.x : nullptr;
  det_enter(ticket, blockIdx.y);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int k = k0 + wc * 64 + b * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + wr * 64 + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (n < J.N && k < J.K) atomicAdd(J.C + (size_t)n * J.ldc + k, acc[a][b][i]);
      }
    }
  if (do_bias && n0 + tid < J.N) atomicAdd(J.bias + n0 + tid, bsum);
  det_leave(ticket);
}

// ---- bf16 throughput variant: 256(n) x 256(k) output tile per 8-wave workgroup, 32-point stages double-buffered in
// LDS (one barrier per stage; the next stage's global loads are in flight during the MFMAs), transposing
// ds_read_b64_tr_b16 fragment reads.  Each wave owns 64(n) x 128(k): 2 x 4 accumulator tiles.  Blocks that share an
// (job, point-split) - i.e. the same A rows - get consecutive ids on ONE XCD so the second read of a tile hits L2.
__device__ __attribute__((aligned(16))) unsigned short w2_zeros[8];   // zero-initialised (16 bytes of +0 in bf16 and fp16)
#define W2_LD (256 + 32)       // 576-byte rows: the 4 rows of a tr-read block fall on disjoint bank groups
#define W2_BK 64               // points per stage (one barrier per stage; 2 stages x 2 operands = 144 KB of LDS)
#define W2_STAGE (W2_BK * W2_LD)
// The stage tiles are [point row][column] with 576-byte rows; the 8-byte column slots of a row are XOR-swizzled by the row:
//   slot' = slot ^ w2_swz(row),  w2_swz(row) = 2 ((row >> 1) & 3)        (even: the two slots of a 16-byte piece stay together)
// so that (a) a transposing fragment read - 4 rows x 8 slots per 32 lanes - still covers 64 distinct banks (the XOR permutes
// slots inside an aligned block of 8, the rows' 64-byte bank offsets stay disjoint), and (b) a NATIVE-order chunk - the 8
// lanes of a ds_write_b128 group writing the same 16-byte piece of 8 consecutive rows - spreads over all 32 banks
// (row & 1 moves a row by 16 banks, the XOR by 4, 8 or 12) instead of two.  mm is a multiple of 16, so a lane's swizzle is a
// constant of the kernel.
__device__ __forceinline__ int w2_swz(int row) { return ((row >> 1) & 3) << 1; }
template <typename T> __device__ __forceinline__ typename Elem<T>::frag w2_frag(const T *tile, int mm, int col0, int lane) {
  const int h = lane >> 5, grp = (lane >> 4) & 1, i = lane & 15, q = i >> 2, p = i & 3;
  // lane part of the address (a constant of the kernel): row 8 h + q, swizzled slot 4 grp + p of the 32-column block; col0 is
  // a multiple of 32 columns = 8 slots and the XOR stays inside an aligned block of 8 slots, so the block offset just adds.
  // The second read takes row + 4: its swizzle differs in the slot's bit 2 only, i.e. +-4 slots from the first, lane constant.
  const int s_lo = (4 * grp + p) ^ w2_swz(8 * h + q);
  const int d_hi = 4 * W2_LD + ((((s_lo ^ 4) - s_lo)) << 2);
  const T *a = tile + (size_t)(mm + 8 * h + q) * W2_LD + (((col0 >> 2) + s_lo) << 2);
  typedef __attribute__((address_space(3))) s16x4 lds_v4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)a);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4 *)(a + d_hi));
  union { s16x4 s[2]; typename Elem<T>::frag b; } u;   // the transposing read moves 16-bit lanes: element type agnostic
  u.s[0] = lo; u.s[1] = hi;
  return u.b;
}

#if defined(BN_PHASE_TIMING) && defined(BN_PHASE_TIMING_WGRAD)
#define WG_PH_DECL BN_PH_DECL
#define WG_PH(i) BN_PH(i)
#define WG_PH_FLUSH BN_PH_FLUSH
#else
#define WG_PH_DECL
#define WG_PH(i)
#define WG_PH_FLUSH
#endif
// W2_WAVES = 8: wave tile 64(n) x 128(k), 2 waves per SIMD.  W2_WAVES = 4 (wave tile 128 x 128, accumulators in the
// AGPR half of the register file, a third fewer LDS fragment bytes per MFMA) compiles but spills in the k-loop and
// measured 5.7x slower (profiles/r01_ablation.txt): kept only as an experiment switch.
#ifndef W2_WAVES
#define W2_WAVES 8
#endif
#define W2_RA (256 / ((W2_WAVES / 2) * 32))   // 32-row accumulator tiles per wave along n
// One 256 x 256 output tile over the points [mb, me).  NBV = 32-column accumulator tiles this WAVE multiplies (4 for a
// full tile; the 60-column positional-encoding operand only has columns for two tiles of the wc = 0 waves - the other
// waves of such a block just take part in staging and barriers).
template <typename T, int NBV, bool BNAT, bool DET>
__device__ __forceinline__ void w2_body(const WgradJob &J, int n0, int k0, int64_t mb, int64_t me, T *sA, T *sB, float osc,
                                        unsigned int *ticket, unsigned int seq) {
  typedef typename Elem<T>::frag frag_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
  const T *gA = (const T *)J.A + J.a_col0 + n0;
  const T *gB = (const T *)J.B + J.b_col0 + k0;
  // W2_BK rows x 32 chunks (16 B) per operand: NC per thread
  constexpr int RPP = W2_WAVES * 2;                          // rows per pass of the workgroup
  constexpr int NC = W2_BK / RPP;
  const int row0 = tid >> 5, cc = (tid & 31) * 8;            // row-major operands: rows row0 + RPP c, columns cc .. cc+7
  const bool a_ok = n0 + cc < J.N, b_ok = k0 + cc < J.K;
  const int scol = (((cc >> 2) ^ w2_swz(row0)) << 2);        // swizzled column of the thread's 16-byte piece (w2_frag)
  // Native-order B (BNAT; layer-output stashes of the 16-bit modes): wave-instruction c of wave w moves chunk q = NC w + c of
  // the stage = 64 lanes x 16 B of consecutive bytes: 32-point block q & 1, column half (q >> 1) & 1, 32-column block q >> 2
  // of this workgroup's 256 columns; lane (r, h) holds point r, columns 4 h + {0..3} and 8 + 4 h + {0..3} of the half.
  // Before the LDS write the two lanes of a point trade one run (v_permlane32_swap: lanes 32-63 of the first operand with
  // lanes 0-31 of the second), so that lane (r, h) holds the 8 CONSECUTIVE columns 8 h .. 8 h + 7 = one 16-byte piece, written
  // with one ds_write_b128 like a row-major piece.
  const int nr = lane & 31, nh = lane >> 5, nswz = w2_swz(nr);
  static_assert(W2_BK / (W2_WAVES * 2) == 4, "native staging: 4 wave-instructions per wave and stage");
  // chunk q = 4 w + c: 32-point block c & 1, column half (c >> 1) & 1, 32-column block w (one per wave): everything but the
  // wave / lane part of the addresses is a compile-time constant of c
  const int mtn = BNAT ? J.b_bm / 32 : 1, ncb = BNAT ? J.b_F / 32 : 1;
  int cbg = (k0 >> 5) + wave;
  cbg = cbg < ncb ? cbg : ncb - 1;                           // beyond the operand: any valid block (those output columns are never stored)
  const int boff0 = (cbg * mtn * 2 * 64 + lane) * 8;         // + ((c & 1) * 2 + ((c >> 1) & 1)) * 512 elements
  const int lslot0 = 8 * wave + 2 * nh;                      // + 4 ((c >> 1) & 1); LDS row = 32 (c & 1) + nr
  const int64_t tile_elems = (int64_t)J.b_bm * J.b_F;
  // Stage pipeline with ONE register set: while stage s is multiplied, the registers (stage s+1, loaded during stage
  // s-1) are written to the other LDS buffer a chunk pair per 16-point step and re-filled at once with stage s+2 -
  // every global load has a whole stage of MFMAs to arrive, every LDS buffer one barrier between its last read and its
  // next write.
  u32x4 ra[NC], rb[NC];
  [[maybe_unused]] int64_t m_dbg = mb;   // diagnostic variants only
  // columns beyond a row-major operand's extent read one 16-byte block of zeros with row stride 0: the stage loop has no
  // branch (an exec-masked load per chunk split its basic block and cost 6 % of the kernel: profiles/r01_ablation.txt)
  const T *pa = a_ok ? gA + cc : (const T *)w2_zeros, *pb = b_ok ? gB + cc : (const T *)w2_zeros;
  const int64_t sa = a_ok ? J.lda : 0, sb = b_ok ? J.ldb : 0;
  auto gload1 = [&](int64_t m, int c) {
    m = m < me ? m : me - W2_BK;   // the two prefetches past the end re-read the last stage: no branch in the stage loop
    const int64_t row = m + row0 + RPP * c;
    ra[c] = *(const u32x4 *)(pa + row * sa);
    if (BNAT) {
      // tile = m >> log2(bm); first 32-point block of the stage inside its tile = (m mod bm) / 32; a tile image is bm x F elements
      const int64_t tile_off = (m >> J.b_bm_shift) * tile_elems;
      const int mt0 = ((int)m & (J.b_bm - 1)) >> 5;
      rb[c] = *(const u32x4 *)((const T *)J.B + tile_off + mt0 * 1024 + boff0 + ((c & 1) * 2 + ((c >> 1) & 1)) * 512);
    } else {
      rb[c] = *(const u32x4 *)(pb + row * sb);
    }
  };
  auto gload = [&](int64_t m) {
#ifdef W2_SKIP_GLOAD   // diagnostic variant (profiles/ab_bench.sh): compute side only
    if (m > mb + W2_BK) return;
#endif
#pragma unroll
    for (int c = 0; c < NC; ++c) gload1(m, c);
  };
  auto sstore1 = [&](int buf, int c) {
    *(u32x4 *)(sA + buf * W2_STAGE + (row0 + RPP * c) * W2_LD + scol) = ra[c];
    if (BNAT) {
      T *rowp = sB + buf * W2_STAGE + (32 * (c & 1) + nr) * W2_LD;
      const int sl0 = lslot0 + 4 * ((c >> 1) & 1);
      const auto s02 = __builtin_amdgcn_permlane32_swap(rb[c][0], rb[c][2], false, false);   // (run 0, run 1) dword 0
      const auto s13 = __builtin_amdgcn_permlane32_swap(rb[c][1], rb[c][3], false, false);   // dword 1
      *(u32x4 *)(rowp + ((sl0 ^ nswz) << 2)) = u32x4{s02[0], s13[0], s02[1], s13[1]};
    } else {
      *(u32x4 *)(sB + buf * W2_STAGE + (row0 + RPP * c) * W2_LD + scol) = rb[c];
    }
  };
  auto sstore = [&](int buf) {
#ifdef W2_SKIP_SSTORE
    if (m_dbg > mb) return;
#endif
#pragma unroll
    for (int c = 0; c < NC; ++c) sstore1(buf, c);
  };
  f32x16 acc[W2_RA][4];
#pragma unroll
  for (int a = 0; a < W2_RA; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  // bias gradient = column sums of A: the waves that own output columns k0 .. k0+127 of the first k-block add up the
  // A fragments they already hold (lane (r, h): row n = r, points 8h .. 8h+7 of the step)
  const bool do_bias = J.bias != nullptr && k0 == 0 && wc == 0;
  float bsum[W2_RA];
#pragma unroll
  for (int a = 0; a < W2_RA; ++a) bsum[a] = 0.f;
  WG_PH_DECL
  // fragments of 16-point step i+1 are read while the MFMAs of step i run (two fragment sets; sched_barrier keeps
  // hipcc from sinking the reads below the MFMAs)
  auto compute = [&](int buf, int64_t m_next2) {
    const T *cA = sA + buf * W2_STAGE, *cB = sB + buf * W2_STAGE;
    frag_t fa[2][W2_RA], fb[2][NBV > 0 ? NBV : 1];
    auto frags = [&](int set, int mm) {
#pragma unroll
      for (int a = 0; a < W2_RA; ++a) fa[set][a] = w2_frag<T>(cA, mm, wr * (W2_RA * 32) + a * 32, lane);
#pragma unroll
      for (int b = 0; b < NBV; ++b) fb[set][b] = w2_frag<T>(cB, mm, wc * 128 + b * 32, lane);
    };
    frags(0, 0);
#pragma unroll
    for (int i = 0; i < W2_BK / 16; ++i) {
      const int cur = i & 1;
#ifdef W2_SKIP_FRAGS
      if (m_dbg == mb)
#endif
      if (i + 1 < W2_BK / 16) frags(cur ^ 1, (i + 1) * 16);
      __builtin_amdgcn_sched_barrier(0);
#ifdef W2_SKIP_MFMA    // diagnostic variant: memory side only (one MFMA keeps the fragment reads alive)
      if (NBV == 4) mma32(acc[0][0], fa[cur][0] + fa[cur][1], fb[cur][0] + fb[cur][1] + fb[cur][2] + fb[cur][3]);
#else
#pragma unroll
      for (int a = 0; a < W2_RA; ++a)
#pragma unroll
        for (int b = 0; b < NBV; ++b) mma32(acc[a][b], fa[cur][a], fb[cur][b]);
#endif
      {   // every wave adds its A fragments up, only the owners of the bias columns store the sums: no branch in the k-loop
#pragma unroll
        for (int a = 0; a < W2_RA; ++a)
#pragma unroll
          for (int j = 0; j < 8; ++j) bsum[a] += (float)fa[cur][a][j];
      }
      // the next stage's tile goes to the other LDS buffer one chunk pair per 16-point step, under this step's MFMAs, and
      // each register is re-filled with the stage after that at once (instead of 8 writes + 8 loads before the MFMAs
      // start: 1.259 -> 1.211 ms); past the end the re-read last stage lands in the buffer nobody reads again
      {
        constexpr int CPS = NC / (W2_BK / 16) > 0 ? NC / (W2_BK / 16) : 1;
#pragma unroll
        for (int q = 0; q < CPS; ++q) {
          const int c = i * CPS + q;
          if (c < NC) { sstore1(buf ^ 1, c); gload1(m_next2, c); }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  gload(mb);
  sstore(0);
  gload(mb + W2_BK);
  WG_PH(14)
  int buf = 0;
  for (int64_t m = mb; m < me; m += W2_BK) {
    m_dbg = m;
#ifdef W2_SKIP_BARRIER
    if (m == mb)
#endif
    __syncthreads();
    WG_PH(3)
    compute(buf, m + 2 * W2_BK);
    WG_PH(0)
    buf ^= 1;
  }
  const int r = lane & 31, h = lane >> 5;
  if constexpr (DET) det_enter(ticket, seq);
#pragma unroll
  for (int a = 0; a < W2_RA; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int k = k0 + wc * 128 + b * 32 + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + wr * (W2_RA * 32) + a * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (n < J.N && k < J.K) atomicAdd(J.C + (size_t)n * J.ldc + k, acc[a][b][i] * osc);
      }
    }
  if (do_bias) {
#pragma unroll
    for (int a = 0; a < W2_RA; ++a) {
      const float v = (bsum[a] + __shfl_xor(bsum[a], 32)) * osc;
      const int n = n0 + wr * (W2_RA * 32) + a * 32 + r;
      if (h == 0 && n < J.N) atomicAdd(J.bias + n, v);
    }
  }
  if constexpr (DET) det_leave(ticket);
  WG_PH(4)
  WG_PH_FLUSH
}

// DET (deterministic mode) is a template parameter: as a run-time branch it cost the default kernel 3 % (profiles/r02_ablation.txt)
template <typename T, bool DET>
__global__ __launch_bounds__(W2_WAVES * 64, W2_WAVES == 8 ? 2 : 1) void wgrad256_kernel(const WgradArgs A, int n_split, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) char smem_w[];
  T *sA = (T *)smem_w;                 // [2][W2_BK][W2_LD]
  T *sB = sA + 2 * W2_STAGE;
  // XCD-aware id: hardware deals consecutive block ids round-robin over the 8 XCDs; give each XCD a contiguous range
  // (deterministic mode: ids in dispatch order, see det_enter - the tiles of a split then sit on different XCDs)
  const int per = n_blocks / 8;              // n_blocks is a multiple of 8
  const int lid = DET ? (int)blockIdx.x : (int)((blockIdx.x % 8) * per + blockIdx.x / 8);
  const int total_tiles = A.tile0[A.n_jobs];
  if (lid >= total_tiles * n_split) return;
  const int split = lid / total_tiles, tt = lid % total_tiles;
  int jb = 0;
  while (jb + 1 < A.n_jobs && tt >= A.tile0[jb + 1]) ++jb;
  const WgradJob &J = A.job[jb];
  const int t = tt - A.tile0[jb];
  const int tiles_k = (J.K + 255) / 256;
  const int n0 = (t / tiles_k) * 256, k0 = (t % tiles_k) * 256;
  const int64_t mb = (int64_t)split * A.m_per_block;
  const int64_t me = mb + A.m_per_block < A.Mpad ? mb + A.m_per_block : A.Mpad;
  if (mb >= me) return;
  const int wc = (threadIdx.x >> 6) & 1;
  const int cols = J.K - k0 - wc * 128;          // output columns this wave's tiles can reach
  const float osc = wg_unscale(A.amax, J.scale_sel);
  unsigned int *ticket = DET ? A.tickets + tt : nullptr;
#ifdef BN_CLOCK_STAMP_WGRAD
  BN_CLK_BEGIN
#endif
  if (J.b_native) {     // layer-output operand in native order: full-width column blocks only (F is a multiple of 64)
    if (cols >= 65) w2_body<T, 4, true, DET>(J, n0, k0, mb, me, sA, sB, osc, ticket, (unsigned int)split);
    else if (cols >= 33) w2_body<T, 2, true, DET>(J, n0, k0, mb, me, sA, sB, osc, ticket, (unsigned int)split);
    else w2_body<T, 0, true, DET>(J, n0, k0, mb, me, sA, sB, osc, ticket, (unsigned int)split);
  } else if (cols >= 65) w2_body<T, 4, false, DET>(J, n0, k0, mb, me, sA, sB, osc, ticket, (unsigned int)split);
  else if (cols >= 33) w2_body<T, 2, false, DET>(J, n0, k0, mb, me, sA, sB, osc, ticket, (unsigned int)split);
  else if (cols >= 1) w2_body<T, 1, false, DET>(J, n0, k0, mb, me, sA, sB, osc, ticket, (unsigned int)split);
  else w2_body<T, 0, false, DET>(J, n0, k0, mb, me, sA, sB, osc, ticket, (unsigned int)split);
#ifdef BN_CLOCK_STAMP_WGRAD
  BN_CLK_END
#endif
}

struct SkinnyJob {
  const void *X;       // [Mpad][ldx] T, or (native != 0) accumulator-order tile images: see native_off8
  const float *dpre;   // [Mpad][ldp] fp32
  int ldx, x_col0, K, ldp, p_col0, nc;
  int native;          // 0: row-major X.  else: tiles of `bm` points, `ntw` 32-column tiles per wave, tile stride `tstride`
  int bm, ntw, tstride;
  float *out[4];       // row c of the gradient: out[c][k], k < K
  float *bias[4];      // scalar bias gradient of row c (nullable)
  int scale_sel;       // fp16 loss scaling carried by X (see WgradJob.scale_sel; the fp32 dpre columns are never scaled)
  int unit_dpre;       // 1: dpre == 1 for every point (column sums of X)
};
#define BN_MAX_SKINNY_JOBS 10
struct SkinnyArgs {
  SkinnyJob job[BN_MAX_SKINNY_JOBS];
  int n_jobs;
  int64_t Mpad;
  int m_per_block;
  const float *amax;
  unsigned int *tickets;   // deterministic mode: one turn counter per job of this launch (else nullptr), see det_enter
};

template <typename T> __global__ __launch_bounds__(256) void skinny_wgrad_kernel(const SkinnyArgs A) {
  const SkinnyJob &J = A.job[blockIdx.y];
  const float osc = wg_unscale(A.amax, J.scale_sel);
  const int64_t mb = (int64_t)blockIdx.x * A.m_per_block;
  const int64_t me = mb + A.m_per_block < A.Mpad ? mb + A.m_per_block : A.Mpad;
  const int tid = threadIdx.x;
  __shared__ float red[4 * 512 + 4];
  for (int i = tid; i < 4 * 512 + 4; i += 256) red[i] = 0.f;
  __syncthreads();
  if (J.native) {
    // X in accumulator order: one wave instruction reads one 1 KB image block = 32 points x 16 columns (lane (r, h)
    // holds columns 16 gp + 4 h + {0..3} and + 8 of point r).  Wave w of the block takes the 32-column blocks
    // cb = w, w + 4, ... of the head; a lane accumulates its 8 columns x nc outputs over the points, then the 32 lanes
    // of a column set meet in LDS.
    const int lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
    const int mt_n = J.bm / 32, ncb = J.K / 32;
    const T *X = (const T *)J.X;
    float bs[4] = {0, 0, 0, 0}, bs0[4] = {0, 0, 0, 0};
    for (int cb = wv; cb < ncb; cb += 4) {
      const int cbp = J.x_col0 / 32 + cb, wave_n = cbp / J.ntw, nt = cbp % J.ntw;
      float s[2][4][8];
#pragma unroll
      for (int gp = 0; gp < 2; ++gp)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 8; ++e) s[gp][c][e] = 0.f;
      // no branch inside the point loop: all four dpre columns are read and summed (columns beyond nc hold the next head's
      // values or row padding; their sums are never stored), every lane keeps the bias sums, one lane set stores them
#pragma unroll 2
      for (int64_t m0 = mb; m0 < me; m0 += 32) {
        const int64_t tile = m0 / J.bm;
        const int mt = (int)(m0 % J.bm) / 32;
        const float *dp = J.dpre + (m0 + r) * J.ldp + J.p_col0;
        float d[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = dp[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) bs[c] += d[c];
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          float x[8];
          ld8(X + (size_t)tile * J.tstride + ((((size_t)(wave_n * J.ntw + nt) * mt_n + mt) * 2 + gp) * 64 + lane) * 8, x);
#pragma unroll
          for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[gp][c][e] += d[c] * x[e];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < J.nc) {
#pragma unroll
          for (int gp = 0; gp < 2; ++gp)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float *dst = &red[c * 512 + cb * 32 + 16 * gp + 4 * h + (e & 3) + 8 * (e >> 2)];
              if (A.tickets) {          // deterministic mode: a fixed butterfly over the 32 points instead of 32-way LDS atomics
                float v = s[gp][c][e];
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o);
                if (r == 0) *dst = v;   // (column blocks of different waves are disjoint)
              } else atomicAdd(dst, s[gp][c][e]);   // LDS, 32-way
            }
        }
      if (cb == 0) {   // the bias gradient is the dpre column sum: taken from the pass over the head's first column block
#pragma unroll
        for (int c = 0; c < 4; ++c) bs0[c] = bs[c];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) bs[c] = 0.f;
    }
    if (wv == 0 && h == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < J.nc) {
          if (A.tickets) {
            float v = bs0[c];
#pragma unroll
            for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            if (r == 0) red[4 * 512 + c] = v;
          } else atomicAdd(&red[4 * 512 + c], bs0[c]);
        }
    }
  } else {
  const T *X = (const T *)J.X + J.x_col0;
  // thread = (row group rg, 8-column group cg): every wave instruction reads whole 16-byte chunks of consecutive
  // rows (K <= 512 columns -> K/8 <= 64 column groups, 256/(K/8) rows in flight per block)
  const int ncg = J.K / 8, nrg = 256 / ncg;
  const int cg = tid % ncg, rg = tid / ncg;
  float s[4][8], bs[4] = {0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) s[c][e] = 0.f;
  if (rg < nrg) {
#pragma unroll 2
    for (int64_t m = mb + rg; m < me; m += nrg) {   // branch-free like the native form above
      float x[8];
      ld8(X + m * J.ldx + cg * 8, x);
      const float *dp = J.dpre + m * J.ldp + J.p_col0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float d = J.unit_dpre ? 1.f : dp[c];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[c][e] += d * x[e];
        bs[c] += d;
      }
    }
  }
  // row groups meet in LDS (nrg-way LDS atomics; deterministic mode: the row groups add one after the other)
  if (A.tickets) {
    for (int turn = 0; turn < nrg; ++turn) {
      if (rg == turn) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (c < J.nc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[c * 512 + cg * 8 + e] += s[c][e];
            if (cg == 0) red[4 * 512 + c] += bs[c];
          }
      }
      __syncthreads();
    }
  } else if (rg < nrg) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
      if (c < J.nc) {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(&red[c * 512 + cg * 8 + e], s[c][e]);
        if (cg == 0) atomicAdd(&red[4 * 512 + c], bs[c]);
      }
  }
  }
  // ONE global atomic per output element per block (same-address atomics from thousands of adders serialise at the
  // memory side)
  __syncthreads();
  unsigned int *ticket = A.tickets ? A.tickets + blockIdx.y : nullptr;
  det_enter(ticket, blockIdx.x);
  for (int i = tid; i < J.nc * J.K; i += 256) {
    const int c = i / J.K, k = i % J.K;
    atomicAdd(J.out[c] + k, red[c * 512 + k] * osc);
  }
  if (tid < J.nc && J.bias[tid]) atomicAdd(J.bias[tid], red[4 * 512 + tid]);   // bias sums come from the unscaled fp32 dpre
  det_leave(ticket);
}

template <typename T, int MT, int NT, int WAVES> static int launch_bwd(const BwdArgs &a, int64_t tiles, hipStream_t st) {
  constexpr int BM = MT * 32;
  const size_t lds = (size_t)BM * (a.g.F + Elem<T>::kPad) * sizeof(T) + (size_t)BM * (BN_DPH + 4) * sizeof(float);
  if (int e = bn_configure_lds((const void *)field_bwd_kernel<T, MT, NT, WAVES>, lds, "field_bwd")) return e;
  BnProfScope prof_(BN_K_BWD_CHAIN, st);
  field_bwd_kernel<T, MT, NT, WAVES><<<dim3((unsigned)tiles), WAVES * 64, lds, st>>>(a);
  BN_LAUNCH_CHECK("field_bwd");
  return 0;
}

extern "C" int bn_field_backward(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                                 const bn_points *pts, const float *out, const float *d_out, void *stash,
                                 const bn_field_grads *G, void *stream) {
  BwdArgs a;
  if (int e = bn_make_geom(desc, &a.g)) return e;
  a.an = desc->normal_an ? 1 : 0;
  BN_REQUIRE(pts && pts->n_points > 0 && packed && out && d_out && stash && G, "field_backward: null argument");
  const FieldGeom &g = a.g;
  a.d = *desc; a.p = *params; a.packed = packed; a.M = pts->n_points; a.out = out; a.d_out = d_out; a.stash = (char *)stash;
  bn_make_packed_layout(g, &a.pl);
  const bool bf = bn_half(desc->dtype);    // 16-bit throughput modes (bf16, fp16)
  const bool f16m = desc->dtype == BN_F16;
  const int BM = g.BM;
  const size_t esz = bn_esize(desc->dtype);
  bn_make_stash_layout(g, pts->n_points, BM, esz, &a.sl);
  const int64_t tiles = ceil_div64(pts->n_points, BM);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  // fp16: the seeds' maxima pick the loss scales of the two backward chains on the device (no host round trip)
  float *amax = f16m ? (float *)((char *)stash + a.sl.gscale) : nullptr;
  a.amax = amax;
  const unsigned amax_grid = (unsigned)(ceil_div64(pts->n_points, 256) < 1024 ? ceil_div64(pts->n_points, 256) : 1024);
  if (f16m) {
    if (hipMemsetAsync(amax, 0, 16, st) != hipSuccess) { bn_set_error("field_backward: memset failed"); return BN_ELAUNCH; }
    if (a.an) {
      grad_amax_kernel<1><<<amax_grid, 256, 0, st>>>(a, amax);
      BN_LAUNCH_CHECK("grad_amax<1>");
    }
  }
  if (a.an) {  // double backward of the normals: produces gbar_PE, abar_l, zbar_l, sbar in the stash
    rc = bn_field_adjoint_backward(desc, params, packed, pts, d_out, stash, stream);
    if (rc) return rc;
  }
  if (f16m) {   // after the adjoint backward: the primal seeds include its sbar
    grad_amax_kernel<0><<<amax_grid, 256, 0, st>>>(a, amax);
    BN_LAUNCH_CHECK("grad_amax<0>");
  }
  rc = [&]() -> int { BN_DISPATCH_TILE(desc->dtype, g, launch_bwd, (a, tiles, st)); }();
  if (rc) return rc;

  // ---- weight gradients
  const StashLayout &sl = a.sl;
  char *S = (char *)stash;
  const int F = g.F, P0 = g.P;
  WgradArgs w;
  w.n_jobs = 0; w.Mpad = sl.Mpad; w.tile0[0] = 0; w.amax = amax;
  int scale_sel = 1;   // gradient operand of the jobs added next: 1 = primal chain (dZ_l, dG), 2 = adjoint chain (gbar_PE, abar_l)
  int b_native = 0;    // B operand of the jobs added next: a native-order layer-output stash (16-bit modes) or a row-major array
  auto add = [&](const void *A_, int lda, int a0, const void *B_, int ldb, int b0, float *C, int ldc, float *bias, int N, int K) {
    if (!C) return;
    WgradJob &j = w.job[w.n_jobs];
    j.scale_sel = scale_sel;
    j.b_native = b_native; j.b_bm = BM; j.b_F = F; j.b_bm_shift = BM == 128 ? 7 : 6;
    j.A = A_; j.B = B_; j.C = C; j.bias = bias; j.lda = lda; j.ldb = ldb; j.ldc = ldc; j.a_col0 = a0; j.b_col0 = b0; j.N = N; j.K = K;
    w.tile0[w.n_jobs + 1] = w.tile0[w.n_jobs] + ((N + 127) / 128) * ((K + 127) / 128);
    ++w.n_jobs;
  };
  const int naty = bf ? 1 : 0;   // Elem<T>::kNativeY: the Y_l stashes of the 16-bit modes are in native order
  for (int l = 0; l < g.L; ++l) {
    const void *dZ = S + sl.dZ[l];
    b_native = 0;
    if (l == 0) add(dZ, F, 0, S + sl.pe, g.KP, 0, G->trunk_w[l], P0, G->trunk_b[l], F, P0);
    else if (l == g.skip) {
      add(dZ, F, 0, S + sl.pe, g.KP, 0, G->trunk_w[l], F + P0, G->trunk_b[l], F, P0);
      b_native = naty;
      add(dZ, F, 0, S + sl.Y[l - 1], F, 0, G->trunk_w[l] ? G->trunk_w[l] + P0 : nullptr, F + P0, nullptr, F, F);
    } else {
      b_native = naty;
      add(dZ, F, 0, S + sl.Y[l - 1], F, 0, G->trunk_w[l], F, G->trunk_b[l], F, F);
    }
  }
  b_native = naty;
  if (!g.fold) add(S + sl.dfeats, F, 0, S + sl.Y[g.L - 1], F, 0, G->feats_w, F, G->feats_b, F, F);
  b_native = 0;
  if (a.an) {  // dW_l += delta_l^T [gbar_PE ; abar_l]  (delta_l is a forward quantity: the scale rides on gbar_PE / abar_l)
    scale_sel = 2;
    for (int l = 0; l < g.L; ++l) {
      const void *dl = S + sl.adj_delta[l];
      if (l == 0) add(dl, F, 0, S + sl.gbar_pe, g.KP, 0, G->trunk_w[l], P0, nullptr, F, P0);
      else if (l == g.skip) {
        add(dl, F, 0, S + sl.gbar_pe, g.KP, 0, G->trunk_w[l], F + P0, nullptr, F, P0);
        add(dl, F, 0, S + sl.adj_abar[l], F, 0, G->trunk_w[l] ? G->trunk_w[l] + P0 : nullptr, F + P0, nullptr, F, F);
      } else add(dl, F, 0, S + sl.adj_abar[l], F, 0, G->trunk_w[l], F, nullptr, F, F);
    }
  }
  scale_sel = 1;
  for (int hd = 0; hd < g.n_heads; ++hd) {
    const int p = hd / 2, hl = hd % 2;
    // folded: the head's first layer reads Y_{L-1}; the gradient is that of the folded matrix (bn_field_desc.fold_feats)
    b_native = g.fold ? naty : 0;         // (unfolded: B = the row-major feats stash)
    add(S + sl.dG[p], g.pass_N[p], hl * g.H2, S + (g.fold ? sl.Y[g.L - 1] : sl.feats), F, 0, G->head_w1[hd], F, G->head_b1[hd], g.H2, F);
  }
  if (g.DD > 0 && G->head0_wdir) {   // d/d rgb_from_xyzdir.0.weight[:, F:] = dG_rgb^T [encoded view direction]
    b_native = 0; scale_sel = 1;
    add(S + sl.dG[0], g.pass_N[0], 0, S + sl.dirpe, g.KD, 0, G->head0_wdir, (int)G->head0_wdir_ld, nullptr, g.H2, g.DD);
  }
  if (g.TD > 0 && G->head1_wt) {     // d/d beta_from_xyz.0.weight[:, F:] = dG_beta^T [image embedding]
    b_native = 0; scale_sel = 1;
    add(S + sl.dG[0], g.pass_N[0], g.H2, S + sl.dirpe, g.KD, g.KT0, G->head1_wt, (int)G->head1_wt_ld, nullptr, g.H2, g.TD);
  }
  BN_REQUIRE(w.n_jobs <= BN_MAX_WGRAD_JOBS, "field_backward: too many wgrad jobs");
  // deterministic mode (det_enter): turn counters in the stash, zeroed per call
  const bool det = bn_deterministic() != 0;
  unsigned int *tickets = det ? (unsigned int *)(S + sl.tickets) : nullptr;
  unsigned int tk_used = 0;
  if (det) BN_HIP_CHECK(hipMemsetAsync(tickets, 0, BN_DET_TICKETS * sizeof(unsigned int), st), "field_backward: ticket memset");
  auto launch_wgrad = [&](WgradArgs &wv, unsigned int *tk) -> int {
    if (wv.n_jobs == 0) return 0;
    wv.tickets = tk;
    if (bf) {
      // 256 x 256 tiles, one 8-wave workgroup per CU: size the point splits for ~4 workgroups per CU in total
      for (int j = 0; j < wv.n_jobs; ++j)
        wv.tile0[j + 1] = wv.tile0[j] + ((wv.job[j].N + 255) / 256) * ((wv.job[j].K + 255) / 256);
      const int tiles = wv.tile0[wv.n_jobs];
#ifndef W2_BLOCKS
#define W2_BLOCKS 512   // tiles x point splits <= two rounds of the 256 CUs (one 144 KB workgroup per CU): 1024 -> 1.295 ms, 512 -> 1.253, 256 -> 1.290
#endif
      int64_t n_split = W2_BLOCKS / tiles;
      if (n_split < 1) n_split = 1;
      int64_t mpb2 = ceil_div64(ceil_div64(sl.Mpad, n_split), W2_BK) * W2_BK;
      if (mpb2 < 512) mpb2 = 512;
      n_split = ceil_div64(sl.Mpad, mpb2);
      wv.m_per_block = (int)mpb2;
      const int n_blocks = (int)ceil_div64((int64_t)tiles * n_split, 8) * 8;
      const size_t lds = (size_t)4 * W2_STAGE * 2;
      const void *kfn = tk ? (f16m ? (const void *)wgrad256_kernel<f16, true> : (const void *)wgrad256_kernel<bf16, true>)
                           : (f16m ? (const void *)wgrad256_kernel<f16, false> : (const void *)wgrad256_kernel<bf16, false>);
      if (int e = bn_configure_lds(kfn, lds, "wgrad256")) return e;
      BnProfScope prof_(BN_K_WGRAD, st);
      const dim3 grd((unsigned)n_blocks), blk(W2_WAVES * 64);
      if (tk) {
        if (f16m) wgrad256_kernel<f16, true><<<grd, blk, lds, st>>>(wv, (int)n_split, n_blocks);
        else wgrad256_kernel<bf16, true><<<grd, blk, lds, st>>>(wv, (int)n_split, n_blocks);
      } else {
        if (f16m) wgrad256_kernel<f16, false><<<grd, blk, lds, st>>>(wv, (int)n_split, n_blocks);
        else wgrad256_kernel<bf16, false><<<grd, blk, lds, st>>>(wv, (int)n_split, n_blocks);
      }
      BN_LAUNCH_CHECK("wgrad256");
      return 0;
    }
    // fp32 parity path: 128 x 128 tiles; split the points so that the grid has a few thousand workgroups
    for (int j = 0; j < wv.n_jobs; ++j)
      wv.tile0[j + 1] = wv.tile0[j] + ((wv.job[j].N + 127) / 128) * ((wv.job[j].K + 127) / 128);
    int64_t splits = 2048 / (wv.tile0[wv.n_jobs] > 0 ? wv.tile0[wv.n_jobs] : 1);
    if (splits < 1) splits = 1;
    int64_t mpb = ceil_div64(ceil_div64(sl.Mpad, splits), WG_BK) * WG_BK;
    if (mpb < 256) mpb = 256;
    wv.m_per_block = (int)mpb;
    dim3 grid((unsigned)wv.tile0[wv.n_jobs], (unsigned)ceil_div64(sl.Mpad, mpb));
    BnProfScope prof_(BN_K_WGRAD, st);
    wgrad_kernel<float><<<grid, 256, 0, st>>>(wv);
    BN_LAUNCH_CHECK("wgrad");
    return 0;
  };
  if (!det) {
    if (int e = launch_wgrad(w, nullptr)) return e;
  } else {
    // jobs that add into the same matrix (same C: the primal and the analytic-normal term of a trunk layer) take separate,
    // stream-ordered launches, each job list in its original order
    bool left[BN_MAX_WGRAD_JOBS];
    for (int j = 0; j < w.n_jobs; ++j) left[j] = true;
    for (int n_left = w.n_jobs; n_left > 0;) {
      WgradArgs gen = w;
      gen.n_jobs = 0; gen.tile0[0] = 0;
      for (int j = 0; j < w.n_jobs; ++j) {
        if (!left[j]) continue;
        bool clash = false;
        for (int q = 0; q < gen.n_jobs; ++q) clash = clash || gen.job[q].C == w.job[j].C;
        if (clash) continue;
        gen.job[gen.n_jobs++] = w.job[j];
        left[j] = false; --n_left;
      }
      if (int e = launch_wgrad(gen, tickets + tk_used)) return e;
      tk_used += (unsigned int)gen.tile0[gen.n_jobs];
      BN_REQUIRE(tk_used <= BN_DET_TICKETS / 2, "field_backward: too many output tiles for the deterministic mode");
    }
  }
  SkinnyArgs s;
  s.n_jobs = 0; s.Mpad = sl.Mpad; s.amax = amax;
  for (int i = 0; i < BN_MAX_SKINNY_JOBS; ++i) { s.job[i].scale_sel = 0; s.job[i].unit_dpre = 0; }   // X = forward activations unless noted
  {
    SkinnyJob &j = s.job[s.n_jobs];
    j.X = S + sl.Y[g.L - 1]; j.ldx = F; j.x_col0 = 0; j.K = F; j.dpre = (const float *)(S + sl.dpre_trunk); j.ldp = 4; j.p_col0 = 0;
    j.nc = g.ch_normal_lr >= 0 ? 4 : 1; j.native = 0;
    if (naty) { j.native = 1; j.bm = BM; j.ntw = g.NT; j.tstride = BM * F; }     // Y_{L-1} in accumulator order (16-bit modes)
    for (int c = 0; c < 4; ++c) { j.out[c] = nullptr; j.bias[c] = nullptr; }
    j.out[0] = G->sigma_w; j.bias[0] = G->sigma_b;
    if (g.ch_normal_lr >= 0) {
      BN_REQUIRE(G->normal_w && G->normal_b, "field_backward: normal grads missing");
      for (int c = 0; c < 3; ++c) { j.out[1 + c] = G->normal_w + (size_t)c * F; j.bias[1 + c] = G->normal_b + c; }
    }
    if (j.out[0]) ++s.n_jobs;
  }
  if (a.an && G->sigma_w) {  // dw_sigma += sum_m s'(m) abar_L[m][:] = sum_m abar'_L[m][:]  (the stash holds abar' = s' abar)
    SkinnyJob &j = s.job[s.n_jobs++];
    j.unit_dpre = 1;
    j.scale_sel = 2;   // abar_L carries the adjoint chain's loss scale
    j.X = S + sl.adj_abar[g.L]; j.ldx = F; j.x_col0 = 0; j.K = F; j.dpre = (const float *)(S + sl.sprime); j.ldp = 1; j.p_col0 = 0; j.nc = 1; j.native = 0;
    for (int c = 0; c < 4; ++c) { j.out[c] = nullptr; j.bias[c] = nullptr; }
    j.out[0] = G->sigma_w;
  }
  for (int hd = 0; hd < g.n_heads; ++hd) {
    if (!G->head_w2[hd]) continue;
    const int p = hd / 2, hl = hd % 2;
    SkinnyJob &j = s.job[s.n_jobs++];
    j.X = S + sl.G[p]; j.ldx = 0; j.x_col0 = hl * g.H2; j.K = g.H2;
    j.native = 1; j.bm = BM; j.ntw = g.pass_NTW[p]; j.tstride = BM * F;
    j.dpre = (const float *)(S + sl.dpre_head); j.ldp = BN_DPH; j.p_col0 = hd * 3; j.nc = desc->head_out[hd];
    for (int c = 0; c < 4; ++c) { j.out[c] = nullptr; j.bias[c] = nullptr; }
    for (int c = 0; c < j.nc; ++c) { j.out[c] = G->head_w2[hd] + (size_t)c * g.H2; j.bias[c] = G->head_b2[hd] ? G->head_b2[hd] + c : nullptr; }
  }
  if (s.n_jobs > 0) {
#ifndef SKINNY_SPLITS
#define SKINNY_SPLITS 256   // 512: 0.129 ms, 256: 0.102 ms, 128: 0.169 ms per launch (the per-block LDS + global atomics tail vs parallelism)
#endif
    // deterministic mode: the splits of a job add one after the other (~3.5 us a turn): 64 instead of 256 (0.94 -> see
    // profiles/r02_ablation.txt)
    const int skinny_splits = det ? 64 : SKINNY_SPLITS;
    int64_t smpb = ceil_div64(ceil_div64(sl.Mpad, skinny_splits), BM) * BM;   // whole tiles per block (native jobs walk tile images)
    s.m_per_block = (int)smpb;
    auto launch_skinny = [&](SkinnyArgs &sv, unsigned int *tk) -> int {
      if (sv.n_jobs == 0) return 0;
      sv.tickets = tk;
      dim3 grid((unsigned)ceil_div64(sl.Mpad, smpb), (unsigned)sv.n_jobs);
      BnProfScope prof_(BN_K_SKINNY, st);
      if (f16m) skinny_wgrad_kernel<f16><<<grid, 256, 0, st>>>(sv);
      else if (bf) skinny_wgrad_kernel<bf16><<<grid, 256, 0, st>>>(sv);
      else skinny_wgrad_kernel<float><<<grid, 256, 0, st>>>(sv);
      BN_LAUNCH_CHECK("skinny_wgrad");
      return 0;
    };
    if (!det) {
      if (int e = launch_skinny(s, nullptr)) return e;
    } else {     // same rule as above: two jobs that add into the same row (sigma_w: primal and analytic-normal term) never share a launch
      bool left[BN_MAX_SKINNY_JOBS];
      for (int j = 0; j < s.n_jobs; ++j) left[j] = true;
      unsigned int tk2 = BN_DET_TICKETS / 2;
      for (int n_left = s.n_jobs; n_left > 0;) {
        SkinnyArgs gen = s;
        gen.n_jobs = 0;
        for (int j = 0; j < s.n_jobs; ++j) {
          if (!left[j]) continue;
          bool clash = false;
          for (int q = 0; q < gen.n_jobs; ++q)
            for (int c = 0; c < 4; ++c)
              for (int c2 = 0; c2 < 4; ++c2) clash = clash || (s.job[j].out[c] && gen.job[q].out[c2] == s.job[j].out[c]);
          if (clash) continue;
          gen.job[gen.n_jobs++] = s.job[j];
          left[j] = false; --n_left;
        }
        if (int e = launch_skinny(gen, tickets + tk2)) return e;
        tk2 += (unsigned int)gen.n_jobs;
      }
    }
  }
  if (g.TD > 0 && G->d_t_embed) {   // gradient of the beta head's embedding input, per point
    BN_REQUIRE(params->head1_wt && params->head1_wt_ld >= g.TD && g.H2 <= 256, "field_backward: head1_wt missing (beta)");
    const int64_t M = pts->n_points;
    const unsigned blocks = (unsigned)(ceil_div64(M, 4) < 4096 ? ceil_div64(M, 4) : 4096);
    const int ldg = g.pass_N[0];
    if (f16m) head_xin_grad_kernel<f16><<<dim3(blocks), 256, 0, st>>>((const f16 *)(S + sl.dG[0]), ldg, g.H2, g.H2, params->head1_wt, params->head1_wt_ld, g.TD, M, amax, G->d_t_embed);
    else if (bf) head_xin_grad_kernel<bf16><<<dim3(blocks), 256, 0, st>>>((const bf16 *)(S + sl.dG[0]), ldg, g.H2, g.H2, params->head1_wt, params->head1_wt_ld, g.TD, M, amax, G->d_t_embed);
    else head_xin_grad_kernel<float><<<dim3(blocks), 256, 0, st>>>((const float *)(S + sl.dG[0]), ldg, g.H2, g.H2, params->head1_wt, params->head1_wt_ld, g.TD, M, amax, G->d_t_embed);
    BN_LAUNCH_CHECK("head_xin_grad");
  }
  return 0;
}
