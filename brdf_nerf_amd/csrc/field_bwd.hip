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
#include "field_wgrad.h"

int bn_field_adjoint_backward(const bn_field_desc *desc, const bn_field_params *params, const void *packed, const bn_points *pts,
                              const float *d_out, void *stash, void *stream);

// bit 1 of bn_device_faults: a hand-over of the barrier-free backward trunk never arrived (pp_wait, field_kernels.h)
__device__ unsigned int g_bwd_fault;
unsigned int bn_bwd_fault_read(hipStream_t) {
  unsigned int v = 0u;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_bwd_fault), sizeof(unsigned int), 0, hipMemcpyDeviceToHost) != hipSuccess) return 0x80000000u;
  return v;
}
// device address of the word on the current device (the forward's launch wrapper mirrors it to the host beside its own, and
// bn_adam_multi reads both: a faulted step skips its update)
const unsigned int *bn_bwd_fault_ptr() {
  void *p = nullptr;
  return hipGetSymbolAddress(&p, HIP_SYMBOL(g_bwd_fault)) == hipSuccess ? (const unsigned int *)p : nullptr;
}

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

__global__ void clear_words_kernel(unsigned int *p, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
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

template <typename T, int MT, int NTW, bool D16>
__device__ __forceinline__ void bwd_head_dG(const BwdArgs &A, int p, T *ACT, const float *DPH, int64_t /*m0*/, int64_t tile) {
  typedef typename DKind<T, D16>::type DK;     // kind of the derivative stash (field_kernels.h)
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
  const char *DGs = A.stash + A.sl.DG[p] + (size_t)tile * dtile_bytes<DK>(BM, F);
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    f32x4 wa[2][3], wb[2][3];
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
      const int nl = pc0 + nt * 32 + 16 * gp + 4 * h - hl * g.H2;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        wa[gp][c] = c < nout ? *(const f32x4 *)(w2 + (size_t)c * g.H2 + nl) : f32x4{0, 0, 0, 0};
        wb[gp][c] = c < nout ? *(const f32x4 *)(w2 + (size_t)c * g.H2 + nl + 8) : f32x4{0, 0, 0, 0};
      }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = mt * 32 + r;
      const float d0 = DPH[m * BN_DPH + hd * 3 + 0], d1 = DPH[m * BN_DPH + hd * 3 + 1], d2 = DPH[m * BN_DPH + hd * 3 + 2];
      const DPiece<DK> pc = dpiece_load<DK>(DGs + dpiece_off<DK, MT, NTW>(wave, nt, mt, lane));
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int n0 = pc0 + nt * 32 + 16 * gp + 4 * h;
        float dg[8];
        dpiece_get<DK>(pc, gp, g.act, 1.f, dg);
        float va[4], vb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          va[e] = (d0 * wa[gp][0][e] + d1 * wa[gp][1][e] + d2 * wa[gp][2][e]) * dg[e];
          vb[e] = (d0 * wb[gp][0][e] + d1 * wb[gp][1][e] + d2 * wb[gp][2][e]) * dg[4 + e];
        }
        *(vec4 *)(ACT + (size_t)m * LDA + n0) = to_vec4(T(), va[0], va[1], va[2], va[3]);
        *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) = to_vec4(T(), vb[0], vb[1], vb[2], vb[3]);
      }
    }
  }
}

template <typename T, int MT, int NT, int WAVES, bool D16>
__global__ __launch_bounds__(WAVES * 64, 2) void field_bwd_kernel(const BwdArgs A) {
  typedef typename DKind<T, D16>::type DK;     // kind of the derivative stash (field_kernels.h)
  typedef typename Elem<T>::vec4 vec4;
  constexpr int BM = MT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FieldGeom &g = A.g;
  const int F = g.F, LDA = F + Elem<T>::kPad, KSF = F / 16;
  T *ACT = (T *)smem;
  float *DPH = (float *)(ACT + (size_t)BM * LDA);  // [BM][BN_DPH] head pre-sigmoid gradients
  float *DPT = DPH + BM * BN_DPH;                        // [BM][4]  (d sigma_raw, d normal_raw)
  int *WR = (int *)(DPT + BM * 4), *RD = WR + 2;         // hand-over counters of the barrier-free trunk (16 ints)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int64_t tile = blockIdx.x, m0 = tile * BM, M = A.M;
  const T *packed = (const T *)A.packed;
  // weight fragments in flight per wave: the barrier-free trunk (F = 512, 16-bit) gains 1-2 % from the forward's depth
  // (profiles/r04_ablation.txt item 7); the shapes under barriers keep the depth tuned for them in round 2
  // (BN_BWD_PP_DEPTH = 6: diag.h)
  constexpr int DP = (NT == 2 && WAVES == 8 && Elem<T>::kFastMath) ? (BN_BWD_PP_DEPTH | BN_GEMM_AFFINE) : BwdDepth<T>::value;
  // D_lo pieces fetched before the layer's GEMM (the rest right after its last MFMA): all of them when a piece is 16 bytes
  constexpr int NPRE = std::is_same<DK, DK8>::value ? NT : 1;
  BN_PH_DECL
  BN_CLK_BEGIN

  // ---------------------------------------------------------------- pre-activation gradients of the small outputs
  // fp16 mode: the chain runs on gradients scaled by S (a power of two, from the device-side maximum); the fp32 copies
  // kept for the skinny weight-gradient kernel stay unscaled, the 16-bit dZ_l / dG stashes carry S and the weight-gradient
  // kernel removes it from its fp32 sums.
  const float gs = chain_scale(A.amax);
  if (tid < 16) WR[tid] = 0;
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
    if (g.pass_heads[p] == 2) bwd_head_dG<T, MT, NT, D16>(A, p, ACT, DPH, m0, tile);
    else bwd_head_dG<T, MT, BN_SINGLE_HEAD_NTW(NT), D16>(A, p, ACT, DPH, m0, tile);
    BN_PH(1)
    __syncthreads();
    BN_PH(2)
    BN_PH(3)
    const int KSp = g.pass_N[p] / 16;
    tile_to_global<T>(ACT, LDA, (T *)(A.stash + A.sl.dG[p]) + (size_t)m0 * g.pass_N[p], g.pass_N[p], BM, g.pass_N[p]);
    if (wave_on) gemm_seg<T, MT, NT, DP>(acc, packed + A.pl.bwd_head[p] + (size_t)(ncol0 / 32) * KSp * 512, KSp, ACT, LDA, lane);
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
  // 16-bit modes (round 4): dZ_l leaves in accumulator-native order straight from the epilogue registers that produce it (one
  // coalesced 1 KB buffer store per wave instruction, like Y_l in the forward) - no row-major copy of the LDS tile riding in the
  // next GEMM; the weight-gradient kernel stages native chunks for both operands (field_wgrad.hip, w2_body<.., ANAT>).
  // F = 512 (NT == 2, all eight waves own columns): the trunk below the top layer runs WITHOUT workgroup barriers, as the
  // forward's does (field_fwd.hip): waves 0-3 (group 0) own output columns 0-255, waves 4-7 (group 1) columns 256-511, one of
  // each per SIMD; every wave multiplies over input-column half 0, then half 1; group 1 is held half a GEMM behind group 0, so
  // a group's epilogue - the wait for its derivative bytes, the dZ stores - runs under the other group's MFMAs.  Counters in
  // LDS (they only grow, 4 per layer each): WR[g] waves of group g past their epilogue; RD[2 h + g] waves of group g done
  // reading column half h.
  constexpr bool NATZ = Elem<T>::kNativeY;
#ifndef BN_BWD_NO_PINGPONG
  constexpr bool PING = NT == 2 && WAVES == 8 && Elem<T>::kFastMath;
#else
  constexpr bool PING = false;
#endif
  const int grp = wave >> 2;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  DPiece<DK> dpre[NT][MT];
  // D_lo = d act / d z of layer lo, read back in accumulator order (DTile pieces).  The loads are issued BEFORE the layer's
  // GEMM - in flight while the MFMAs run - so the epilogue does not sit on HBM latency (fp32 mode: only n-tile 0 fits there,
  // the others follow the last MFMA).
  auto load_D = [&](int lo, int nt0, int nt1) {
    const char *Ds = A.stash + A.sl.D[lo] + (size_t)tile * dtile_bytes<DK>(BM, F);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if (nt < nt0 || nt >= nt1) continue;
#ifdef BN_PROBE_NO_D          // timing probe only (results wrong): no derivative loads
        if constexpr (std::is_same<DK, DK8>::value) { dpre[nt][mt].w = u32x4{0x80808080u + (unsigned)lane, 0x90909090u, 0xa0a0a0a0u, 0xb0b0b0b0u}; continue; }
#endif
        dpre[nt][mt] = dpiece_load<DK>(Ds + dpiece_off<DK, MT, NT>(wave, nt, mt, lane));
      }
  };
  // epilogue of the layer that produces dZ_lo: (+ the top layer's rank-1 terms) x D_lo (+ zbar_lo), -> LDS tile and stash
  auto epilogue = [&](int lo, auto top_tag) {
    constexpr bool top = decltype(top_tag)::value;
    const bool nlr = g.ch_normal_lr >= 0;
    const float dscale = (g.act == BN_ACT_SIN && lo == 0) ? 30.f : 1.f;    // w0 of layer lo (the stash holds the unscaled derivative)
    const auto Zr = stash_rsrc(A.stash + A.sl.dZ[lo] + (size_t)tile * BM * F * sizeof(T));
    const int voff = lane * 16;
    // analytic normals: all of the layer's zbar_lo pieces leave HBM together at the top of the epilogue (the GEMM's fragment
    // registers are free by now) - one exposed latency per layer instead of one per group of pieces
    typedef typename Elem<T>::wide WT;
    constexpr bool ZPRE = sizeof(WT) == 2;
    u32x4 zraw[ZPRE ? NT : 1][ZPRE ? MT : 1][2];
    if constexpr (ZPRE) {
      if (A.an) {
        const WT *Zb = (const WT *)(A.stash + A.sl.adj_zbar[lo]) + (size_t)tile * BM * F;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) zraw[nt][mt][gp] = stash_load((const u32x4 *)(Zb + native_off8<MT, NT>(wave, nt, mt, gp, lane)));
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mt * 32 + r;
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int n0 = ncol0 + nt * 32 + 16 * gp + 4 * h;
          float dv[8], v[8];
          dpiece_get<DK>(dpre[nt][mt], gp, g.act, dscale, dv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = acc[nt][mt][8 * gp + e];
          if constexpr (top) {  // rank-1 terms of the sigma head and the learned-normal head
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
            if constexpr (ZPRE) {
              const auto zq = __builtin_bit_cast(bf16x8, zraw[nt][mt][gp]);
#pragma unroll
              for (int e = 0; e < 8; ++e) zb[e] = (float)zq[e];
            } else {
              ld8((const WT *)(A.stash + A.sl.adj_zbar[lo]) + (size_t)tile * BM * F + native_off8<MT, NT>(wave, nt, mt, gp, lane), zb);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += zb[e] * zr;
          }
          if constexpr (NATZ) {   // one conversion serves the LDS tile (the next GEMM's operand) and the native dZ stash
            const typename Elem<T>::frag q = cvt8(T(), v);
            *(vec4 *)(ACT + (size_t)m * LDA + n0) = __builtin_shufflevector(q, q, 0, 1, 2, 3);
            *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) = __builtin_shufflevector(q, q, 4, 5, 6, 7);
            stash_store_buf(Zr, voff, ((((wave_u * NT + nt) * MT + mt) * 2 + gp) * 1024), q);
          } else {
            *(vec4 *)(ACT + (size_t)m * LDA + n0) = to_vec4(T(), v[0], v[1], v[2], v[3]);
            *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) = to_vec4(T(), v[4], v[5], v[6], v[7]);
          }
        }
      }
  };
  // ---- top layer (under barriers: its input comes from the head passes): dY_{L-1} = Wf^T dFeats + sigma/normal rank-1 terms.
  // With fold_feats the product is already in the accumulators.
  {
    if (!g.fold) {
      zero_acc<MT, NT>(acc);
      T *fdst = (T *)(A.stash + A.sl.dfeats) + (size_t)m0 * F;     // dFeats stays a row-major stash
      if (!ride) tile_to_global<T>(ACT, LDA, fdst, F, BM, F);
      if (wave_on) {
        load_D(g.L - 1, 0, NPRE);
        const size_t off = A.pl.bwd_feats + (size_t)(ncol0 / 32) * KSF * 512;
        if (ride) {
          TileCopyExact<T> zcopy(ACT, LDA, fdst, F, F, tid, WAVES * 64);
          gemm_seg<T, MT, NT, DP>(acc, packed + off, KSF, ACT, LDA, lane, zcopy);
        } else {
          gemm_seg<T, MT, NT, DP>(acc, packed + off, KSF, ACT, LDA, lane);
        }
        load_D(g.L - 1, NPRE, NT);
      }
    } else if (wave_on) {
      load_D(g.L - 1, 0, NT);
    }
    BN_PH(9)
    __syncthreads();
    BN_PH(10)
    if (wave_on) epilogue(g.L - 1, std::true_type());
    BN_PH(11)
    __syncthreads();
    BN_PH(12)
  }
  // ---- layers L-1 .. 1: dY_{l-1} = W_l^T dZ_l
  for (int l = g.L - 1; l >= 1; --l) {
    const int it = g.L - 1 - l;                   // barrier-free layers so far
    const int lo = l - 1;                         // layer whose pre-activation gradient is produced
    zero_acc<MT, NT>(acc);
    T *zdst = (T *)(A.stash + A.sl.dZ[l]) + (size_t)m0 * F;
    if (!NATZ && !ride) tile_to_global<T>(ACT, LDA, zdst, F, BM, F);
    // (BN_BWD_D_AT, diag.h: where the barrier-free trunk issues a layer's derivative loads: 0 before the GEMM, 1 between its halves, 2 behind it)
    if (wave_on) {
      if (!PING || BN_BWD_D_AT == 0) load_D(lo, 0, NPRE);
      const size_t off = A.pl.bwd_trunk[l] + (size_t)(ncol0 / 32) * KSF * 512;
      if constexpr (PING) {
        const int half = KSF / 2;
        NoSide none;
        pp_wait(WR + 0, 4 * it, &g_bwd_fault);                        // half 0 of dZ_l is written
        if (grp == 1) pp_wait(RD + 0, 4 * (it + 1), &g_bwd_fault);    // group 0 is done with its phase 1 of this layer: the lag
        BN_PH(12)
        __builtin_amdgcn_s_setprio(1);
#ifdef BN_PP_SPLIT      // A/B switch (results unchanged): two half-GEMMs, each with its own weight prologue (round 4)
        gemm_range<T, MT, NT, DP | BN_PP_NKS>(acc, packed + off, KSF, 0, half, ACT, LDA, lane, none);
        BN_PH(9)
        pp_signal(RD + 0 + grp, lane);
        pp_wait(WR + 1, 4 * it, &g_bwd_fault);                        // half 1
        BN_PH(12)
        if (BN_BWD_D_AT == 1) load_D(lo, 0, NPRE);
        gemm_range<T, MT, NT, DP | BN_PP_NKS>(acc, packed + off, KSF, half, half, ACT, LDA, lane, none);
#else
        if constexpr (sizeof(T) == 2) {        // (PING implies a 16-bit mode)
          (void)half; (void)none;
          gemm_trunk<T, MT, NT, (DP & (BN_GEMM_AFFINE - 1)), 32, 16>(acc, packed + off, KSF, ACT, LDA, lane, [&]() {
            pp_signal(RD + 0 + grp, lane);
            pp_wait(WR + 1, 4 * it, &g_bwd_fault);                    // half 1
            if (BN_BWD_D_AT == 1) load_D(lo, 0, NPRE);
          });
        }
#endif
        __builtin_amdgcn_s_setprio(0);
        pp_signal(RD + 2 + grp, lane);
        if (BN_BWD_D_AT == 2) load_D(lo, 0, NPRE);
      } else if (!NATZ && ride) {
        // fp32 mode: the row-major stash copy of the tile this GEMM reads (dZ_l) rides inside the GEMM
        TileCopyExact<T> zcopy(ACT, LDA, zdst, F, F, tid, WAVES * 64);
        gemm_seg<T, MT, NT, DP>(acc, packed + off, KSF, ACT, LDA, lane, zcopy);
      } else {
        gemm_seg<T, MT, NT, DP>(acc, packed + off, KSF, ACT, LDA, lane);
      }
      load_D(lo, NPRE, NT);
    }
    BN_PH(9)
    if constexpr (PING) {     // all eight waves have read this group's columns of dZ_l
      pp_wait(RD + 2 * grp + 0, 4 * (it + 1), &g_bwd_fault);
      pp_wait(RD + 2 * grp + 1, 4 * (it + 1), &g_bwd_fault);
    } else {
      __syncthreads();
    }
    BN_PH(10)
    if (wave_on) epilogue(lo, std::false_type());
    BN_PH(11)
    if constexpr (PING) pp_signal(WR + grp, lane);
    else __syncthreads();
    BN_PH(12)
  }
  if (PING) __syncthreads();   // (every wave signalled its last epilogue before arriving here)
  if (!NATZ) tile_to_global<T>(ACT, LDA, (T *)(A.stash + A.sl.dZ[0]) + (size_t)m0 * F, F, BM, F);
  BN_PH(13)
#ifndef BN_PHASE_TIMING_WGRAD
  BN_PH_FLUSH
#endif
#ifndef BN_CLOCK_STAMP_WGRAD
  BN_CLK_END
#endif
}

// ------------------------------------------------------------------------------------------ weight gradients
template <typename T, int MT, int NT, int WAVES, bool D16> static int launch_bwd_k(const BwdArgs &a, int64_t tiles, hipStream_t st) {
  constexpr int BM = MT * 32;
  const size_t lds = (size_t)BM * (a.g.F + Elem<T>::kPad) * sizeof(T) + (size_t)BM * (BN_DPH + 4) * sizeof(float) + 64;
  if (int e = bn_configure_lds((const void *)field_bwd_kernel<T, MT, NT, WAVES, D16>, lds, "field_bwd")) return e;
  BnProfScope prof_(BN_K_BWD_CHAIN, st);
  field_bwd_kernel<T, MT, NT, WAVES, D16><<<dim3((unsigned)tiles), WAVES * 64, lds, st>>>(a);
  BN_LAUNCH_CHECK("field_bwd");
  return 0;
}
template <typename T, int MT, int NT, int WAVES> static int launch_bwd(const BwdArgs &a, int64_t tiles, hipStream_t st) {
  if constexpr (std::is_same<T, f16>::value) {
    if (a.g.dsz == 2) return launch_bwd_k<T, MT, NT, WAVES, true>(a, tiles, st);     // fp16 + analytic normals: fp16 derivative stash
  }
  return launch_bwd_k<T, MT, NT, WAVES, false>(a, tiles, st);
}

extern "C" int bn_field_backward(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                                 const bn_points *pts, const float *out, const float *d_out, void *stash,
                                 const bn_field_grads *G, void *stream) {
  return bn_field_backward_parts(desc, params, packed, pts, out, d_out, stash, G, BN_BWD_ALL, stream);
}

extern "C" int bn_field_backward_parts(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                                       const bn_points *pts, const float *out, const float *d_out, void *stash,
                                       const bn_field_grads *G, int32_t parts, void *stream) {
  BwdArgs a;
  if (int e = bn_make_geom(desc, &a.g)) return e;
  if (const int f = bn_field_fault_seen()) {
    bn_set_error("field_backward: an earlier %s launch lost an LDS hand-over (pp_wait timed out): its results are invalid",
                 (f & 1) ? "forward" : "backward");
    return BN_ELAUNCH;
  }
  a.an = desc->normal_an ? 1 : 0;
  BN_REQUIRE(pts && pts->n_points > 0 && packed && out && d_out && stash && G, "field_backward: null argument");
  BN_REQUIRE(parts > 0 && (parts & ~BN_BWD_ALL) == 0, "field_backward: parts=%d", parts);
  BN_REQUIRE(pts->point_offset == 0 && (pts->total_points == 0 || pts->total_points == pts->n_points),
             "field_backward: the backward runs over a whole point set (point_offset = 0)");
  BN_REQUIRE(pts->seg1_points == 0 || (pts->rays && pts->z && pts->z2 && pts->n_samples2 > 0 && pts->seg1_points < pts->n_points &&
                                       pts->seg1_points % pts->n_samples == 0 && (pts->n_points - pts->seg1_points) % pts->n_samples2 == 0),
             "field_backward: bad two-block point set");
  BN_REQUIRE(pts->seg1_points == 0 || !(desc->t_dim > 0 && G->d_t_embed), "field_backward: d_t_embed is not served for a two-block point set");
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
  if (parts & BN_BWD_CHAIN) {
  if (f16m) {
    // (a kernel, not hipMemsetAsync: inside a captured HIP graph the memset became a memset NODE, and replays of such graphs
    // were seen to run the atomicMax kernels below against a stale / late-cleared word once in a few hundred steps - fp16 steps
    // with analytic normals ended with non-finite parameters, profiles/history/r03_ablation.txt; kernel nodes are ordered like launches)
    clear_words_kernel<<<1, 64, 0, st>>>((unsigned int *)amax, 4);
    BN_LAUNCH_CHECK("clear amax");
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
  }

  // ---- weight gradients
  const StashLayout &sl = a.sl;
  char *S = (char *)stash;
  const int F = g.F, P0 = g.P;
  WgradArgs w;
  w.n_jobs = 0; w.Mpad = sl.Mpad; w.tile0[0] = 0; w.amax = amax; w.part = nullptr; w.n_split = 1;
  int scale_sel = 1;   // gradient operand of the jobs added next: 1 = primal chain (dZ_l, dG), 2 = adjoint chain (gbar_PE, abar_l)
  int b_native = 0;    // B operand of the jobs added next: a native-order layer-output stash (16-bit modes) or a row-major array
  int a_native = 0;    // A operand of the jobs added next: a native-order dZ_l stash (16-bit modes: the trunk's) or a row-major array
  int part = BN_BWD_WGRAD_TRUNK;   // which part of the backward the jobs added next belong to (bn_field_backward_parts)
  auto add = [&](const void *A_, int lda, int a0, const void *B_, int ldb, int b0, float *C, int ldc, float *bias, int N, int K) {
    if (!C || !(parts & part)) return;
    WgradJob &j = w.job[w.n_jobs];
    j.scale_sel = scale_sel;
    j.b_native = (b_native ? WG_B_NATIVE : 0) | (a_native ? WG_A_NATIVE : 0); j.b_bm = BM; j.b_F = F; j.b_bm_shift = BM == 128 ? 7 : 6;
    j.A = A_; j.B = B_; j.C = C; j.bias = bias; j.lda = lda; j.ldb = ldb; j.ldc = ldc; j.a_col0 = a0; j.b_col0 = b0; j.N = N; j.K = K;
    w.tile0[w.n_jobs + 1] = w.tile0[w.n_jobs] + ((N + 127) / 128) * ((K + 127) / 128);
    ++w.n_jobs;
  };
  const int naty = bf ? 1 : 0;   // Elem<T>::kNativeY: the Y_l stashes of the 16-bit modes are in native order
  a_native = naty;     // Elem<T>::kNativeY: the trunk's dZ_l are native images in the 16-bit modes (field_bwd_kernel's epilogue)
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
  a_native = 0;
  b_native = naty;
  part = BN_BWD_WGRAD_HEADS;
  if (!g.fold) add(S + sl.dfeats, F, 0, S + sl.Y[g.L - 1], F, 0, G->feats_w, F, G->feats_b, F, F);
  part = BN_BWD_WGRAD_TRUNK;
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
  part = BN_BWD_WGRAD_HEADS;
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
  // the point splits' partial tiles go to slabs in the stash and are summed in fixed order (field_wgrad.hip): no atomics, a
  // bitwise reproducible gradient in every mode
  float *wgpart = (float *)(S + sl.wgpart);
  if (w.n_jobs > 0)
    if (int e = bn_launch_wgrad(w, bf, f16m, sl.Mpad, wgpart, sl.wgpart_bytes, st)) return e;
  if (!(parts & BN_BWD_SKINNY)) return 0;
  SkinnyArgs s;
  s.n_jobs = 0; s.Mpad = sl.Mpad; s.amax = amax; s.part = nullptr;
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
    // (SKINNY_SPLITS = 256, diag.h: 512: 0.129 ms, 256: 0.102 ms, 128: 0.169 ms per launch - the per-block tail vs parallelism)
    // small batches (strong scaling: 512 rays per GPU = 32,768 points per launch): a block walks its points in a latency-bound
    // loop, so fewer points per block is faster until the per-block tail takes over - at least 2 point tiles per split
    // (session 50, 512 rays: 32 splits 0.085 ms, 64 0.066, 128 0.060, 256 0.089 per launch)
    int skinny_splits = SKINNY_SPLITS;
    {
      int64_t most = sl.Mpad / (2 * (int64_t)BM) > 1 ? sl.Mpad / (2 * (int64_t)BM) : 1;
      if (sl.Mpad <= 131072 && most > 128) most = 128;      // up to 2048 rays x 64 samples: 128 blocks per job keep the tail short
      if (skinny_splits > most) skinny_splits = (int)most;
    }
    int64_t smpb = ceil_div64(ceil_div64(sl.Mpad, skinny_splits), BM) * BM;   // whole tiles per block (native jobs walk tile images)
    s.m_per_block = (int)smpb;
    if (int e = bn_launch_skinny(s, bf, f16m, sl.Mpad, smpb, wgpart, sl.wgpart_bytes, st)) return e;
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
