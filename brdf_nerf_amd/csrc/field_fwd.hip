// Fused field-MLP forward for gfx950: positional encoding -> trunk (Siren/ReLU, skip concat) -> sigma /
// learned-normal heads -> feats -> two-layer sigmoid heads, one workgroup per tile of BM points, all
// activations resident in LDS.  Replaces Mapping.forward (models/nerf.py:53-70), calc_features
// (models/spsbrdfnerf.py:636-646) and SpSBRDFNeRF.forward (:662-757) of the reference.
#include "field_kernels.h"

struct FwdArgs {
  FieldGeom g;
  bn_field_desc d;
  bn_field_params p;
  PackedLayout pl;
  StashLayout sl;
  const void *packed;
  bn_points pts;
  float *out;     // [M][C], or sigma [M] when sigma_only
  char *stash;    // nullptr: inference, nothing kept
  int sigma_only;
};

// Activation + derivative of one pre-activation.  ACT is compile-time (the epilogues branch once, outside their
// element loops).  In the bf16 throughput mode (FAST) the Siren layers' packed weights and biases are pre-scaled by
// w0/(2 pi) (bn_pack_field), so `z` already is the argument of v_sin_f32/v_cos_f32 in revolutions: one transcendental
// per output, no range-reduction multiplies.  The parity mode keeps z = W x + b and the accurate sincos.
#define BN_INV_2PI 0.15915494309189535f

// sticky fault word of the forward kernels (lost LDS hand-over in the barrier-free trunk, field_kernels.h pp_wait)
__device__ unsigned int g_fwd_fault;
static unsigned int *g_fault_host = nullptr;         // pinned mirror of [forward word, backward word], refreshed asynchronously every 64th launch
static unsigned int g_fwd_launches = 0;
const unsigned int *bn_fwd_fault_ptr() {
  void *p = nullptr;
  return hipGetSymbolAddress(&p, HIP_SYMBOL(g_fwd_fault)) == hipSuccess ? (const unsigned int *)p : nullptr;
}

BN_PH_DEFINE_READER(bn_debug_phase_read_fwd)
BN_CLK_DEFINE(bn_debug_clock_read_fwd)
BN_TL_DEFINE_READER(bn_debug_timeline_read_fwd)
// `c` is the UNSCALED derivative (cos(.) / the ReLU mask); d act / d z = w0 c for a Siren layer (DTile, field_kernels.h).
template <bool FAST, int ACT> __device__ __forceinline__ void act_eval(float z, float w0, float &y, float &c) {
  if (ACT == BN_ACT_SIN) {
    if (FAST) {
      y = __builtin_amdgcn_sinf(z);
      c = __builtin_amdgcn_cosf(z);
    } else {
      sincos_cw(w0 * z, y, c);
    }
  } else {
    y = z > 0.f ? z : 0.f; c = z > 0.f ? 1.f : 0.f;
  }
}
template <bool FAST> __device__ __forceinline__ float act_prescale(int act, float w0) {
  return (FAST && act == BN_ACT_SIN) ? w0 * BN_INV_2PI : 1.f;
}

// One row of the extra-input tile (FieldGeom.KD columns): mapping[1](view direction of the point) for --input_viewdir
// (spsbrdfnerf.py:689-692), then the image embedding for --beta, zero elsewhere.
template <typename T>
__device__ __forceinline__ void fill_dir_row(const FwdArgs &A, int64_t gm, T *row) {
  constexpr bool FAST = Elem<T>::kFastMath;
  const FieldGeom &g = A.g;
  float dv[3] = {0.f, 0.f, 0.f};
  if (gm < A.pts.n_points) {
    if (A.pts.xyz) {
      if (A.pts.dirs) { dv[0] = A.pts.dirs[gm * 3]; dv[1] = A.pts.dirs[gm * 3 + 1]; dv[2] = A.pts.dirs[gm * 3 + 2]; }
    } else {
      const float *rr = A.pts.rays + (gm / A.pts.n_samples) * A.pts.ray_stride;
      dv[0] = rr[3]; dv[1] = rr[4]; dv[2] = rr[5];
    }
  }
  if (g.DD > 0) {
    if (g.dir_freqs > 0) {
      for (int k = 0; k < g.dir_freqs; ++k) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float sn, co;
          sincos_t<FAST>(f * dv[c], sn, co);
          row[6 * k + c] = (T)sn;
          row[6 * k + 3 + c] = (T)co;
        }
      }
    } else {
      for (int k = 0; k < 3; ++k) row[k] = (T)dv[k];
    }
  }
  for (int k = g.DD; k < g.KT0; ++k) row[k] = (T)0.f;
  // --beta: the per-image embedding of the point's ray, raw (spsbrdfnerf.py:709)
  const float *te = nullptr;
  if (g.TD > 0 && gm < A.pts.n_points && A.pts.t_embed)
    te = A.pts.t_embed + (A.pts.xyz ? gm : gm / A.pts.n_samples) * g.TD;
  for (int k = g.KT0; k < g.KD; ++k) row[k] = (T)((te && k - g.KT0 < g.TD) ? te[k - g.KT0] : 0.f);
}

// One pass over up to two heads: hidden = act(W1 feats + b1) kept in registers, second layer (<= 3 outputs)
// as per-lane partial dots reduced through LDS.  NTW = 32-column tiles per wave in this pass.
// The pass's first-layer biases and second-layer weights are staged in LDS (PRM, the positional-encoding buffer,
// free by now) ahead of the GEMM: the epilogue then reads them without queueing behind its own stash stores.
// The hidden activations G are stashed in accumulator order like DG (one coalesced 16-byte store per lane).
template <typename T, int MT, int NTW, int WAVES, bool KEEP, bool DIR, bool HOT, bool D16>
__device__ __forceinline__ void head_pass(const FwdArgs &A, int p, const T *ACT, float *PRM, float *RED, int64_t m0, int64_t tile
#ifdef BN_PHASE_TIMING
                                          , unsigned long long (&ph_)[BN_PH_N], unsigned long long &pt_
#endif
) {
  constexpr int BM = MT * 32;
  constexpr bool FAST = Elem<T>::kFastMath;
  constexpr int DP = FwdDepth<T, KEEP>::value;
  typedef typename DKind<T, D16>::type DK;     // kind of the derivative stash (field_kernels.h)
  const FieldGeom &g = A.g;
  const int F = g.F, LDA = F + Elem<T>::kPad, KSF = F / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int64_t M = A.pts.n_points;
  constexpr bool keep = KEEP;   // compile-time: the inference variant drops every derivative (cos) computation
  const int N = g.pass_N[p];
  const int pc0 = wave * 32 * NTW;                // first column of this wave in the pass
  const bool on = pc0 < N;
  // PRM[0][n] = b1 (pre-scaled for the fast sine), PRM[1 + c][n] = w2[c] (0 beyond the head's outputs), n < N
  for (int n = tid; n < N; n += WAVES * 64) {
    const int hn = n / g.H2, nl = n - hn * g.H2, hdn = 2 * p + hn;
    const int nout = A.d.head_out[hdn];
    PRM[n] = A.p.head_b1[hdn][nl] * act_prescale<FAST>(g.act, 1.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) PRM[(1 + c) * N + n] = c < nout ? A.p.head_w2[hdn][(size_t)c * g.H2 + nl] : 0.f;
  }
  // --input_viewdir: the encoded view direction of every point, an extra K segment of the rgb head's first layer (like the
  // positional-encoding segment of the skip layer).  The tile lives behind the staged parameters in the (free) encoding buffer.
  // DIR (= the model has extra inputs, KD > 0) is a template parameter: compiled into the one kernel, the segment costs the hot
  // head pass registers (2 - 3 % of the forward with no extra input at all: profiles/history/r02_ablation.txt)
  const bool dir_on = DIR && p == 0;
  T *DIRT = (T *)((char *)PRM + 8192);
  const int LDD = g.KD + Elem<T>::kPad;
  if constexpr (DIR) {
    if (dir_on && tid < BM) fill_dir_row<T>(A, m0 + tid, DIRT + (size_t)tid * LDD);
  }
  f32x16 acc[NTW][MT];
  zero_acc<MT, NTW>(acc);
  if (on) gemm_full<T, MT, NTW, DP, HOT>(acc, (const T *)A.packed + A.pl.fwd_head[p] + (size_t)(pc0 / 32) * KSF * 512, KSF, ACT, LDA, lane);
  BN_PH(9)
  __syncthreads();   // PRM (and the direction tile) filled
  if constexpr (DIR) {
    if (dir_on) {
      const int KSD = g.KD / 16;
      if (keep) tile_to_global<T>(DIRT, LDD, (T *)(A.stash + A.sl.dirpe) + (size_t)m0 * g.KD, g.KD, BM, g.KD);
      if (on) gemm_seg<T, MT, NTW, DP>(acc, (const T *)A.packed + A.pl.fwd_dir + (size_t)(pc0 / 32) * KSD * 512, KSD, DIRT, LDD, lane);
    }
  }
  if (on) {
    T *Gs = keep ? (T *)(A.stash + A.sl.G[p]) + (size_t)tile * BM * F : nullptr;
    char *DGs = keep ? A.stash + A.sl.DG[p] + (size_t)tile * dtile_bytes<DK>(BM, F) : nullptr;
    float part[MT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) part[mt][0] = part[mt][1] = part[mt][2] = 0.f;
    auto epilogue = [&](auto act_tag) {
      constexpr int ACTK = decltype(act_tag)::value;
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        DHalf<DK> dh0[MT];     // derivative bytes of group gp = 0, stored with those of gp = 1 (one 16-byte piece per lane)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int n0 = pc0 + nt * 32 + 16 * gp + 4 * h;   // column in the pass of the first run (second: +8)
          const f32x4 ba = *(const f32x4 *)(PRM + n0), bb = *(const f32x4 *)(PRM + n0 + 8);
          f32x4 wa[3], wb[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            wa[c] = *(const f32x4 *)(PRM + (1 + c) * N + n0);
            wb[c] = *(const f32x4 *)(PRM + (1 + c) * N + n0 + 8);
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            float y[8], dd[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              act_eval<FAST, ACTK>(acc[nt][mt][8 * gp + e] + ba[e], 1.f, y[e], dd[e]);
              act_eval<FAST, ACTK>(acc[nt][mt][8 * gp + 4 + e] + bb[e], 1.f, y[4 + e], dd[4 + e]);
            }
            // the second layer sees the stored (rounded) hidden value: fwd and bwd stay consistent
            const typename Elem<T>::frag yq = cvt8(T(), y);
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (float)yq[e];
            if (keep) {
              st_frag(Gs + native_off8<MT, NTW>(wave, nt, mt, gp, lane), yq);
              if (gp == 0) dh0[mt] = dhalf_make<DK>(dd, 1.f);
              else dpiece_store<DK>(DGs + dpiece_off<DK, MT, NTW>(wave, nt, mt, lane), dh0[mt], dhalf_make<DK>(dd, 1.f));
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
              for (int e = 0; e < 4; ++e) part[mt][c] += y[e] * wa[c][e] + y[4 + e] * wb[c][e];
          }
        }
      }
    };
    if (g.act == BN_ACT_SIN) epilogue(std::integral_constant<int, BN_ACT_SIN>());
    else epilogue(std::integral_constant<int, BN_ACT_RELU>());
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = part[mt][c];
        v += __shfl_xor(v, 32);
        if (h == 0) RED[(wave * 3 + c) * BM + mt * 32 + r] = v;
      }
  }
  BN_PH(10)
  __syncthreads();
  if (tid < BM * g.pass_heads[p]) {
    const int m = tid % BM, hl2 = tid / BM, hd2 = 2 * p + hl2;
    const int64_t gm = m0 + m;
    const int wph = g.H2 / (32 * NTW);  // waves per head
    const int nout = A.d.head_out[hd2], kind = A.d.head_kind[hd2];
    float v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float s = 0.f;
      for (int w = hl2 * wph; w < (hl2 + 1) * wph; ++w) s += RED[(w * 3 + c) * BM + m];
      v[c] = c < nout ? (kind == BN_HEAD_BETA ? softplus_f(s + A.p.head_b2[hd2][c]) : sigmoid_f(s + A.p.head_b2[hd2][c])) : 0.f;
    }
    if (gm < M) {
      float *o = A.out + gm * g.C + g.head_col[hd2];
      if (kind == BN_HEAD_PLAIN || kind == BN_HEAD_BETA) {
        for (int c = 0; c < nout; ++c) o[c] = v[c];
      } else if (kind == BN_HEAD_HAPKE_THETA) {
        o[0] = v[0] * 0.52359877559829887f;  // pi*30/180
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float y = nout == 1 ? v[0] : v[c];
          if (kind == BN_HEAD_RPV_K) y = (y - 0.5f) * 2.f + 1.f;
          else if (kind == BN_HEAD_RPV_THETA) y = (y - 0.5f) * 2.f;
          o[c] = y;
        }
      }
    }
  }
  __syncthreads();
  BN_PH(11)
}

#ifdef BN_PHASE_TIMING
#define BN_PH_ARGS , ph_, pt_
#else
#define BN_PH_ARGS
#endif
template <typename T, int MT, int NT, int WAVES, bool KEEP, bool DIR, bool D16>
__global__ __launch_bounds__(WAVES * 64, 2) void field_fwd_kernel(const FwdArgs A) {
  typedef typename Elem<T>::vec4 vec4;
  typedef typename DKind<T, D16>::type DK;     // kind of the derivative stash (field_kernels.h)
  constexpr int BM = MT * 32;
  constexpr int PADE = Elem<T>::kPad;
  constexpr bool FAST = Elem<T>::kFastMath;
  constexpr bool NATY = Elem<T>::kNativeY;
  constexpr int DP = FwdDepth<T, KEEP>::value;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const FieldGeom &g = A.g;
  const int F = g.F, LDA = F + PADE, KP = g.KP, LDP = KP + PADE;
  T *ACT = (T *)smem;
  T *PE = ACT + (size_t)BM * LDA;
  float *RED = (float *)(PE + (size_t)BM * LDP);  // [WAVES][3][BM]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int64_t tile = blockIdx.x, m0 = tile * BM, M = A.pts.n_points;
  const T *packed = (const T *)A.packed;
  constexpr bool keep = KEEP;   // compile-time: the inference variant drops every derivative (cos) computation
  BN_PH_DECL
  BN_CLK_BEGIN

  // ---------------------------------------------------------------- points + positional encoding
  {
    constexpr int NP = (WAVES * 64) / BM;
    const int m = tid % BM, part = tid / BM;
    const int64_t gm = m0 + m;
    float x[3] = {0.f, 0.f, 0.f};
    if (gm < M) {
      if (A.pts.xyz) {
        x[0] = A.pts.xyz[gm * 3 + 0]; x[1] = A.pts.xyz[gm * 3 + 1]; x[2] = A.pts.xyz[gm * 3 + 2];
      } else {
        const int64_t ray = gm / A.pts.n_samples;
        const float *rr = A.pts.rays + ray * A.pts.ray_stride;
        const float zz = A.pts.z[gm];
        x[0] = rr[0] + rr[3] * zz; x[1] = rr[1] + rr[4] * zz; x[2] = rr[2] + rr[5] * zz;
      }
    }
    T *row = PE + (size_t)m * LDP;
    if (g.pe_freqs > 0) {
      for (int k = part; k < g.pe_freqs; k += NP) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float s, co;
          sincos_t<FAST>(f * x[c], s, co);
          row[6 * k + c] = (T)s;
          row[6 * k + 3 + c] = (T)co;
        }
      }
      if (part == 0)
        for (int k = g.P; k < KP; ++k) row[k] = (T)0.f;
    } else if (part == 0) {
      for (int k = 0; k < KP; ++k) row[k] = (T)(k < 3 ? x[k] : 0.f);
    }
  }
  if (tid < 8) ((int *)RED)[tid] = 0;   // ping-pong counters (RED is not used before the head passes)
  __syncthreads();
  if (keep) tile_to_global<T>(PE, LDP, (T *)(A.stash + A.sl.pe) + (size_t)m0 * KP, KP, BM, KP);
  BN_PH(0)

  const int ncol0 = wave * 32 * NT;        // first feature of this wave in F-wide phases
  const bool wave_on = ncol0 < F;
  const int KSP = KP / 16, KSF = F / 16;
  f32x16 acc[NT][MT];

  // ---------------------------------------------------------------- trunk
  const int n_on = F / (32 * NT) < WAVES ? F / (32 * NT) : WAVES;   // waves that own output columns
  // stash copies ride inside the GEMMs when the shape fits (NT == 2 means F = 512: always, decided at compile time)
  const bool ride = NT == 2 ? true : tile_copy_exact(F, n_on, WAVES);
  // Two-group ping-pong (F = 512: NT == 2, all eight waves own columns; 16-bit modes).  Waves 0-3 (group 0) own output
  // columns 0-255, waves 4-7 (group 1) columns 256-511; each SIMD hosts one wave of either group.  There is no
  // workgroup barrier inside the trunk.  Every wave multiplies over input-column half 0, then half 1; group 1 is
  // held one half-GEMM behind group 0, so group 0's (VALU) epilogue runs under group 1's second half-GEMM and group
  // 1's epilogue under group 0's first half-GEMM of the next layer.
  // Counters in LDS (they only grow, 4 per layer each):  WR[g] waves of group g past epilogue l;
  // RD[h][g] waves of group g done reading column half h in layer l.
  //   phase 1 (half 0) needs WR[0] (and, for group 1, RD[0][0]: the enforced lag); phase 2 needs WR[1];
  //   the epilogue of group g rewrites half g: needs RD[g][0] and RD[g][1].
  // Round 4 (profiles/r04_simd_timeline*.txt, r04_ab_fwd_*.txt): a wave raises its priority while it multiplies - its MFMAs
  // then win the SIMD's issue arbitration against the partner's epilogue (forward -3 %); the biases START the accumulators
  // (staged per wave in LDS) instead of being added by the epilogue, and the stash stores are buffer instructions with scalar
  // offsets (-3 %, sigma-only -6 %).  Measured and NOT adopted: group 1 a WHOLE GEMM behind with the epilogue split into a
  // pass S (sin, LDS tile, Y stash) and a pass C (cos, D stash) run in opposite order by the two groups, which makes the
  // in-place update legal at that lag - one wave of a SIMD in its GEMM 69 % of the trunk instead of 63 %, but a wave alone
  // multiplies at 61-72 cycles per MFMA inside this kernel (35 in a kernel of its own, profiles/r04_probe_gemm_rate.txt):
  // forward +5 % without the priority, equal with it.
#ifndef BN_NO_PINGPONG
  constexpr bool PING = NT == 2 && WAVES == 8 && FAST;   // bf16 throughput mode only: the fp32 parity mode keeps k in order
#else
  constexpr bool PING = false;
#endif
  int *WR = (int *)RED, *RD = WR + 2;   // RD[2 * h + g]
  const int grp = wave >> 2;
  BN_TL_DECL((char *)RED + 64 + 4096)
  // (BN_GEMM_PRIO, diag.h: priority of a wave while it multiplies; 0 = none, as in rounds 1-3)
#ifdef BN_PRIO_YOUNG      // A/B switch: static priority for the later-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
#endif
  // Anti-phase trunk: a wave's biases (its 64 columns) wait in LDS, [2][64] floats per wave behind the counters, and START the
  // accumulators - the epilogue has no bias add and no bias registers; layer l + 1's are fetched (one dword per lane) at the
  // top of layer l, ahead of every load and store of that layer, and parked after its GEMM.
  float *BIASL = (float *)((char *)RED + 64) + wave * 128;
  float bnext = 0.f;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  if constexpr (PING) BIASL[lane] = A.p.trunk_b[0][ncol0 + lane] * act_prescale<FAST>(g.act, 30.f);
  for (int l = 0; l < g.L; ++l) {
    BN_TL(0)
    if constexpr (PING) {
      if (l + 1 < g.L) bnext = A.p.trunk_b[l + 1][ncol0 + lane];
      const float *bl = BIASL + (l & 1) * 64 + 4 * h;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const f32x4 b0 = *(const f32x4 *)(bl + nt * 32 + 16 * gp), b1 = *(const f32x4 *)(bl + nt * 32 + 16 * gp + 8);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[nt][mt][8 * gp + e] = b0[e]; acc[nt][mt][8 * gp + 4 + e] = b1[e]; }
        }
    } else {
      zero_acc<MT, NT>(acc);
    }
    if (keep && !NATY && !ride && l > 0) tile_to_global<T>(ACT, LDA, (T *)(A.stash + A.sl.Y[l - 1]) + (size_t)m0 * F, F, BM, F);
    if (wave_on) {
      const size_t t0 = (size_t)(ncol0 / 32);
      const T *w_pe = packed + A.pl.fwd_trunk[l][0] + t0 * KSP * 512;
      const T *w_h = packed + A.pl.fwd_trunk[l][l == g.skip ? 1 : 0] + t0 * KSF * 512;
      if (l == 0 || l == g.skip) gemm_seg<T, MT, NT, DP>(acc, w_pe, KSP, PE, LDP, lane);
      if (l > 0) {
        if (PING) {
          const int half = KSF / 2;
          NoSide none;
          BN_PH(1)
          pp_wait(WR + 0, 4 * l, &g_fwd_fault);                       // half 0 of Y_{l-1} is written
          if (grp == 1) pp_wait(RD + 0, 4 * l, &g_fwd_fault);         // group 0 is done with its phase 1 of this layer: the lag
          BN_PH(12)
          BN_TL(1)
          __builtin_amdgcn_s_setprio(BN_GEMM_PRIO);
#ifdef BN_PP_SPLIT      // A/B switch (results unchanged): two half-GEMMs, each with its own weight prologue (rounds 1-4)
          gemm_range<T, MT, NT, DP | BN_PP_NKS>(acc, w_h, KSF, 0, half, ACT, LDA, lane, none);
          BN_PH(1)
          BN_TL(2)
          pp_signal(RD + 0 + grp, lane);
          pp_wait(WR + 1, 4 * l, &g_fwd_fault);                       // half 1
          BN_PH(13)
          BN_TL(3)
          gemm_range<T, MT, NT, DP | BN_PP_NKS>(acc, w_h, KSF, half, half, ACT, LDA, lane, none);
#else
          if constexpr (sizeof(T) == 2) {      // (PING implies a 16-bit mode; the fp32 instantiation never gets here)
            (void)half; (void)none;
            gemm_trunk<T, MT, NT, (DP & (BN_GEMM_AFFINE - 1)), 32, 16>(acc, w_h, KSF, ACT, LDA, lane, [&]() {
              BN_TL(2)
              pp_signal(RD + 0 + grp, lane);
              pp_wait(WR + 1, 4 * l, &g_fwd_fault);                   // half 1
              BN_TL(3)
            });
          }
#endif
          __builtin_amdgcn_s_setprio(0);
          BN_TL(4)
          pp_signal(RD + 2 + grp, lane);
        } else if (keep && !NATY && ride) {
          // the row-major stash copy of Y_{l-1} (the tile this GEMM reads) rides inside the GEMM when the shape fits
          TileCopyExact<T> ycopy(ACT, LDA, (T *)(A.stash + A.sl.Y[l - 1]) + (size_t)m0 * F, F, F, tid, WAVES * 64);
          gemm_seg<T, MT, NT, DP>(acc, w_h, KSF, ACT, LDA, lane, ycopy);
        } else {
          gemm_seg<T, MT, NT, DP>(acc, w_h, KSF, ACT, LDA, lane);
        }
      }
    }
    // this layer's biases: loaded ahead of the barrier, and ahead of the epilogue's stash stores (a load issued
    // after a store is not seen before that store has been acknowledged)
    const float w0 = (l == 0) ? 30.f : 1.f;
    f32x4 bia[NT][2][2];
    if constexpr (PING) {
      if (l + 1 < g.L) BIASL[((l + 1) & 1) * 64 + lane] = bnext * act_prescale<FAST>(g.act, 1.f);
    } else if (wave_on) {
      const float bscale = act_prescale<FAST>(g.act, w0);
      const float *bias = A.p.trunk_b[l];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          const int n0 = ncol0 + nt * 32 + 16 * gp + 4 * h;
          bia[nt][gp][0] = *(const f32x4 *)(bias + n0) * bscale;
          bia[nt][gp][1] = *(const f32x4 *)(bias + n0 + 8) * bscale;
        }
    }
    BN_PH(1)
    if constexpr (PING) {
      // Two-pass epilogue of the anti-phase trunk (see above); the accumulators hold z = W x + b.  Stash stores go through
      // buffer instructions: one lane-offset register, everything else in scalar registers.
      const auto Yr = stash_rsrc(keep ? A.stash + A.sl.Y[l] + (size_t)tile * BM * F * sizeof(T) : nullptr);
      const auto Dr = stash_rsrc(keep ? A.stash + A.sl.D[l] + (size_t)tile * dtile_bytes<DK>(BM, F) : nullptr);
      const int voff = lane * 16;
      auto passes = [&](auto act_tag) {
        typedef typename Elem<T>::frag frag;
        constexpr int ACTK = decltype(act_tag)::value;
        const float dsc = ACTK == BN_ACT_SIN ? w0 : 1.f;
        if (l > 0) {   // all eight waves have read this group's columns of Y_{l-1}
          pp_wait(RD + 2 * grp + 0, 4 * l, &g_fwd_fault);
          pp_wait(RD + 2 * grp + 1, 4 * l, &g_fwd_fault);
        }
        BN_PH(2)
        BN_TL(5)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          DHalf<DK> dh0[MT];     // derivative bytes of group gp = 0, stored with those of gp = 1 (one 16-byte piece per lane)
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            const int n0 = ncol0 + nt * 32 + 16 * gp + 4 * h;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              float y[8], c[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) act_eval<FAST, ACTK>(acc[nt][mt][8 * gp + e], w0, y[e], c[e]);
              const int m = mt * 32 + r;
              const frag yq = cvt8(T(), y);     // one conversion serves the LDS tile (next layer's operand) and the native Y stash
              *(vec4 *)(ACT + (size_t)m * LDA + n0) = __builtin_shufflevector(yq, yq, 0, 1, 2, 3);
              *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) = __builtin_shufflevector(yq, yq, 4, 5, 6, 7);
              if (keep) {
                stash_store_buf(Yr, voff, ((((wave_u * NT + nt) * MT + mt) * 2 + gp) * 1024), yq);
                if (gp == 0) dh0[mt] = dhalf_make<DK>(c, dsc);
                else dpiece_store_buf<DK>(Dr, voff, (wave_u * NT + nt) * MT + mt, dh0[mt], dhalf_make<DK>(c, dsc));
              }
            }
          }
        }
        BN_TL(6)
        pp_signal(WR + grp, lane);
        BN_PH(3)
      };
      if (g.act == BN_ACT_SIN) passes(std::integral_constant<int, BN_ACT_SIN>());
      else passes(std::integral_constant<int, BN_ACT_RELU>());
      BN_PH(4)
    } else {
    __syncthreads();  // every wave has finished reading ACT (in-place update below)
    BN_PH(2)
    if (wave_on) {
      char *Ds = keep ? A.stash + A.sl.D[l] + (size_t)tile * dtile_bytes<DK>(BM, F) : nullptr;
      T *Ys = keep ? (T *)(A.stash + A.sl.Y[l]) + (size_t)tile * BM * F : nullptr;   // native order (16-bit modes)
      auto epilogue = [&](auto act_tag) {
        constexpr int ACTK = decltype(act_tag)::value;
        const float dsc = ACTK == BN_ACT_SIN ? w0 : 1.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          DHalf<DK> dh0[MT];     // derivative bytes of group gp = 0, stored with those of gp = 1 (one 16-byte piece per lane)
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            const int n0 = ncol0 + nt * 32 + 16 * gp + 4 * h;
            const f32x4 ba = bia[nt][gp][0], bb = bia[nt][gp][1];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              float y[8], dd[8];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                act_eval<FAST, ACTK>(acc[nt][mt][8 * gp + e] + ba[e], w0, y[e], dd[e]);
                act_eval<FAST, ACTK>(acc[nt][mt][8 * gp + 4 + e] + bb[e], w0, y[4 + e], dd[4 + e]);
              }
              const int m = mt * 32 + r;
              if (NATY) {     // one conversion serves the LDS tile (next layer's operand) and the native Y stash
                const typename Elem<T>::frag yq = cvt8(T(), y);
                *(vec4 *)(ACT + (size_t)m * LDA + n0) = __builtin_shufflevector(yq, yq, 0, 1, 2, 3);
                *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) = __builtin_shufflevector(yq, yq, 4, 5, 6, 7);
                if (keep) st_frag(Ys + native_off8<MT, NT>(wave, nt, mt, gp, lane), yq);
              } else {
                *(vec4 *)(ACT + (size_t)m * LDA + n0) = to_vec4(T(), y[0], y[1], y[2], y[3]);
                *(vec4 *)(ACT + (size_t)m * LDA + n0 + 8) = to_vec4(T(), y[4], y[5], y[6], y[7]);
              }
              if (keep) {
                if (gp == 0) dh0[mt] = dhalf_make<DK>(dd, dsc);
                else dpiece_store<DK>(Ds + dpiece_off<DK, MT, NT>(wave, nt, mt, lane), dh0[mt], dhalf_make<DK>(dd, dsc));
              }
            }
          }
        }
      };
      if (g.act == BN_ACT_SIN) epilogue(std::integral_constant<int, BN_ACT_SIN>());
      else epilogue(std::integral_constant<int, BN_ACT_RELU>());
    }
    BN_PH(3)
    __syncthreads();
    BN_PH(4)
    }
  }
  if (PING) __syncthreads();   // both halves of the last layer are in LDS (every wave signalled before arriving here)
  BN_TL(9)
  BN_TL_DUMP

  // ---------------------------------------------------------------- sigma (+ learned normal) heads on the matrix pipe
  // The one-row (three-row) heads are 32-row MFMA tiles with zero rows, split over K: wave w multiplies k-steps
  // [w KSF/8, (w+1) KSF/8) for all BM points (4 k-steps x 4 point tiles = 16 MFMAs per wave at F = 512), the 8 partial sums
  // of rows 0..3 meet in LDS (the encoding tile is free after the trunk).  Replaces per-thread VALU dots over the LDS tile
  // (9.7 % of the inference kernel's wave cycles for 0.03 % of its FLOPs, round 1).  16-bit modes only: the fp32 parity mode
  // keeps the fp32 dot products below (same summation order as the validated round-1 kernel).
  if constexpr (FAST) {
    const bool nlr = g.ch_normal_lr >= 0 && !A.sigma_only;
    float *SRED = (float *)PE;                         // [WAVES][BM][4]: sigma_raw, normal_raw xyz partial sums
    // k-steps split over the waves: ceil(KSF / WAVES) each (F = 192: 12 steps, two per wave on six waves), the last active wave
    // takes what is left
    const int nks_w = (KSF + WAVES - 1) / WAVES, ks0 = wave * nks_w;
    const bool kon = ks0 < KSF;
    const int nks = kon ? (KSF - ks0 < nks_w ? KSF - ks0 : nks_w) : 0;
    f32x16 sacc[1][MT], nacc[1][MT];
    zero_acc<MT, 1>(sacc);
    zero_acc<MT, 1>(nacc);
    if (kon) {
      NoSide none;
#if !defined(BN_NO_FIXED_FULL) && !defined(BN_NO_FIXED_SIGMA)
      if constexpr (NT == 2 && WAVES == 8) {       // F = 512: four k-steps per wave, straight-line
        (void)none; (void)nks;
        gemm_fixed<T, MT, 1, 4, 4, 0>(sacc, packed + A.pl.fwd_sigma, KSF, ks0, ACT, LDA, lane, NoMid());
        if (nlr) gemm_fixed<T, MT, 1, 4, 4, 0>(nacc, packed + A.pl.fwd_nlr, KSF, ks0, ACT, LDA, lane, NoMid());
      } else
#endif
      {
        gemm_range<T, MT, 1, DP>(sacc, packed + A.pl.fwd_sigma, KSF, ks0, nks, ACT, LDA, lane, none);
        if (nlr) gemm_range<T, MT, 1, DP>(nacc, packed + A.pl.fwd_nlr, KSF, ks0, nks, ACT, LDA, lane, none);
      }
    }
    if (h == 0) {                                      // accumulator rows 0..3 live in registers 0..3 of lanes 0-31
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        *(f32x4 *)(SRED + ((size_t)wave * BM + mt * 32 + r) * 4) = f32x4{sacc[0][mt][0], nacc[0][mt][0], nacc[0][mt][1], nacc[0][mt][2]};
    }
    __syncthreads();
    if (tid < BM) {
      const int m = tid;
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      for (int w = 0; w < WAVES; ++w)
        if (w * nks_w < KSF) sum += *(const f32x4 *)(SRED + ((size_t)w * BM + m) * 4);
      const int64_t gm = m0 + m;
      const float sraw = sum[0] + A.p.sigma_b[0];
      const float sig = softplus_f(sraw);
      if (keep) ((float *)(A.stash + A.sl.sraw))[gm] = sraw;
      if (gm < M) {
        if (A.sigma_only) A.out[gm] = sig;
        else A.out[gm * g.C + 3] = sig;
      }
      if (nlr) {
        const float v0 = sum[1] + A.p.normal_b[0], v1 = sum[2] + A.p.normal_b[1], v2 = sum[3] + A.p.normal_b[2];
        if (keep) {
          float *nr = (float *)(A.stash + A.sl.nraw) + gm * 4;
          nr[0] = v0; nr[1] = v1; nr[2] = v2; nr[3] = 0.f;
        }
        if (gm < M) {
          const float inv = -1.f / sqrtf(fmaxf(v0 * v0 + v1 * v1 + v2 * v2, 1.1920928955078125e-07f));
          float *o = A.out + gm * g.C + g.ch_normal_lr;
          o[0] = v0 * inv; o[1] = v1 * inv; o[2] = v2 * inv;
        }
      }
    }
    __syncthreads();                                   // the head passes reuse the encoding tile for their parameters
  }
  // fp32 parity mode: VALU dots over h8
  else {
    constexpr int TPR = (WAVES * 64) / BM;  // threads per point
    const int m = tid / TPR, q = tid % TPR;
    const bool nlr = g.ch_normal_lr >= 0;
    float ds = 0.f, dn0 = 0.f, dn1 = 0.f, dn2 = 0.f;
    const T *row = ACT + (size_t)m * LDA;
    // 16-byte weight loads, all independent of each other (the compiler batches them ahead of the FMAs); the
    // learned-normal rows get their own loop so the sigma-only loop has no branch in it
    auto dot8 = [&](const float *w, const typename Elem<T>::frag &v) {
      const f32x4 wa = *(const f32x4 *)w, wb = *(const f32x4 *)(w + 4);
      return (float)v[0] * wa[0] + (float)v[1] * wa[1] + (float)v[2] * wa[2] + (float)v[3] * wa[3] + (float)v[4] * wb[0] +
             (float)v[5] * wb[1] + (float)v[6] * wb[2] + (float)v[7] * wb[3];
    };
#pragma unroll 4
    for (int c8 = q; c8 < F / 8; c8 += TPR) ds += dot8(A.p.sigma_w + c8 * 8, lds_frag<T>(row + c8 * 8));
    if (nlr) {
#pragma unroll 2
      for (int c8 = q; c8 < F / 8; c8 += TPR) {
        const typename Elem<T>::frag v = lds_frag<T>(row + c8 * 8);
        dn0 += dot8(A.p.normal_w + c8 * 8, v);
        dn1 += dot8(A.p.normal_w + F + c8 * 8, v);
        dn2 += dot8(A.p.normal_w + 2 * F + c8 * 8, v);
      }
    }
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) {
      ds += __shfl_xor(ds, o);
      if (nlr) { dn0 += __shfl_xor(dn0, o); dn1 += __shfl_xor(dn1, o); dn2 += __shfl_xor(dn2, o); }
    }
    const int64_t gm = m0 + m;
    if (q == 0) {
      const float sraw = ds + A.p.sigma_b[0];
      const float sig = softplus_f(sraw);
      if (keep) ((float *)(A.stash + A.sl.sraw))[gm] = sraw;
      if (gm < M) {
        if (A.sigma_only) A.out[gm] = sig;
        else A.out[gm * g.C + 3] = sig;
      }
      if (nlr && !A.sigma_only) {
        const float v0 = dn0 + A.p.normal_b[0], v1 = dn1 + A.p.normal_b[1], v2 = dn2 + A.p.normal_b[2];
        if (keep) {
          float *nr = (float *)(A.stash + A.sl.nraw) + gm * 4;
          nr[0] = v0; nr[1] = v1; nr[2] = v2; nr[3] = 0.f;
        }
        if (gm < M) {
          const float inv = -1.f / sqrtf(fmaxf(v0 * v0 + v1 * v1 + v2 * v2, 1.1920928955078125e-07f));
          float *o = A.out + gm * g.C + g.ch_normal_lr;
          o[0] = v0 * inv; o[1] = v1 * inv; o[2] = v2 * inv;
        }
      }
    }
  }
  BN_PH(6)
  if (A.sigma_only) { BN_PH_FLUSH BN_CLK_END return; }

  // ---------------------------------------------------------------- feats = Wf h8 + bf (linear)
  // With fold_feats the caller has multiplied the feats layer into the heads' first layers: the heads read h8 directly.
  if (g.fold) {
    if (keep && !NATY) tile_to_global<T>(ACT, LDA, (T *)(A.stash + A.sl.Y[g.L - 1]) + (size_t)m0 * F, F, BM, F);
  } else {
  zero_acc<MT, NT>(acc);
  if (keep && !NATY && !ride) tile_to_global<T>(ACT, LDA, (T *)(A.stash + A.sl.Y[g.L - 1]) + (size_t)m0 * F, F, BM, F);
  if (wave_on) {
    const T *w_f = packed + A.pl.fwd_feats + (size_t)(ncol0 / 32) * KSF * 512;
    if (keep && !NATY && ride) {
      TileCopyExact<T> ycopy(ACT, LDA, (T *)(A.stash + A.sl.Y[g.L - 1]) + (size_t)m0 * F, F, F, tid, WAVES * 64);
      gemm_seg<T, MT, NT, DP>(acc, w_f, KSF, ACT, LDA, lane, ycopy);
    } else {
      gemm_seg<T, MT, NT, DP>(acc, w_f, KSF, ACT, LDA, lane);
    }
  }
  BN_PH(7)
  __syncthreads();
  if (wave_on) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int n = ncol0 + nt * 32 + 8 * gq + 4 * h;
        const f32x4 b4 = *(const f32x4 *)(A.p.feats_b + n);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int m = mt * 32 + r;
          *(vec4 *)(ACT + (size_t)m * LDA + n) = to_vec4(T(), acc[nt][mt][4 * gq + 0] + b4[0], acc[nt][mt][4 * gq + 1] + b4[1],
                                                         acc[nt][mt][4 * gq + 2] + b4[2], acc[nt][mt][4 * gq + 3] + b4[3]);
        }
      }
  }
  __syncthreads();
  if (keep) tile_to_global<T>(ACT, LDA, (T *)(A.stash + A.sl.feats) + (size_t)m0 * F, F, BM, F);
  }
  BN_PH(8)

  // ---------------------------------------------------------------- two-layer sigmoid heads, up to 2 per pass
  for (int p = 0; p < g.n_pass; ++p) {
    if (g.pass_heads[p] == 2) head_pass<T, MT, NT, WAVES, KEEP, DIR, NT == 2, D16>(A, p, ACT, (float *)PE, RED, m0, tile BN_PH_ARGS);
    else head_pass<T, MT, BN_SINGLE_HEAD_NTW(NT), WAVES, KEEP, DIR, NT == 2, D16>(A, p, ACT, (float *)PE, RED, m0, tile BN_PH_ARGS);
  }
  BN_PH_FLUSH
  BN_CLK_END
}

// ------------------------------------------------------------------------------------------ weight packing
struct PackJob {
  const float *src;
  size_t dst;        // element offset in the packed buffer
  int ld;            // row length of src
  int rows, K;       // valid extents of the PACKED matrix (rows = A rows, K = contraction)
  int rows_pad, K_pad;
  int row_off, col_off;  // offsets into src (in the packed matrix's own row/col sense)
  int transposed;    // 1: packed[row][k] = src[k - k_lo + col_off][row + row_off]
  int k_lo;          // first packed k this job owns
  int masked;        // 1: touch only k in [k_lo, k_lo+K_touch) (several jobs fill one packed matrix)
  int K_touch;       // masked jobs: width of the k range this job owns (>= K; the part beyond K is zero-filled)
  int row_lo;        // first PACKED row that holds src data (rows below it are zero-filled like rows >= row_lo + rows)
  float scale;       // multiplies every element (w0/(2 pi) for the forward Siren matrices in bf16 mode)
};
#define BN_MAX_PACK_JOBS 48
struct PackArgs {
  PackJob job[BN_MAX_PACK_JOBS];
  int n_jobs;
  void *dst;
};

template <typename T> __global__ void pack_kernel(const PackArgs A) {
  const PackJob &j = A.job[blockIdx.y];
  T *dst = (T *)A.dst + j.dst;
  const size_t total = (size_t)j.rows_pad * j.K_pad;
  const int KS = j.K_pad / 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int e = i & 7, lane = (i >> 3) & 63;
    const size_t blk = i >> 9;
    const int ks = blk % KS, rt = blk / KS;
    const int row = rt * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + e;
    const int kk = k - j.k_lo;
    if (j.masked && (kk < 0 || kk >= j.K_touch)) continue;
    float v = 0.f;
    const int rr = row - j.row_lo;
    if (rr >= 0 && rr < j.rows && kk >= 0 && kk < j.K)
      v = j.transposed ? j.src[(size_t)(kk + j.col_off) * j.ld + rr + j.row_off]
                       : j.src[(size_t)(rr + j.row_off) * j.ld + kk + j.col_off];
    dst[i] = (T)(v * j.scale);
  }
}

static size_t esize(int dtype) { return bn_esize(dtype); }

extern "C" size_t bn_field_packed_bytes(const bn_field_desc *desc) {
  FieldGeom g;
  if (bn_make_geom(desc, &g)) return 0;
  PackedLayout pl;
  bn_make_packed_layout(g, &pl);
  return pl.total * esize(desc->dtype);
}

extern "C" size_t bn_field_stash_bytes(const bn_field_desc *desc, int64_t n_points) {
  FieldGeom g;
  if (bn_make_geom(desc, &g)) return 0;
  StashLayout sl;
  bn_make_stash_layout(g, n_points, g.BM, esize(desc->dtype), &sl);
  return sl.total;
}

extern "C" int bn_pack_field(const bn_field_desc *desc, const bn_field_params *P, void *packed, void *stream) {
  FieldGeom g;
  if (int e = bn_make_geom(desc, &g)) return e;
  PackedLayout pl;
  bn_make_packed_layout(g, &pl);
  PackArgs a;
  a.n_jobs = 0;
  a.dst = packed;
  auto add = [&](const float *src, size_t dst, int ld, int rows, int K, int row_off, int col_off, int tr) {
    PackJob &j = a.job[a.n_jobs++];
    j.src = src; j.dst = dst; j.ld = ld; j.rows = rows; j.K = K;
    j.rows_pad = (int)bn_pad(rows, 32); j.K_pad = (int)bn_pad(K, 16);
    j.row_off = row_off; j.col_off = col_off; j.transposed = tr; j.k_lo = 0; j.masked = 0; j.scale = 1.f;
    j.K_touch = K; j.row_lo = 0;
  };
  // bf16 Siren: forward matrices carry w0/(2 pi) so the epilogue feeds v_sin/v_cos directly (see act_eval)
  const bool prescale = bn_half(desc->dtype) && desc->act == BN_ACT_SIN;
  auto fwd_scale = [&](int n_last, float w0) {
    if (prescale)
      for (int q = 0; q < n_last; ++q) a.job[a.n_jobs - 1 - q].scale = w0 * BN_INV_2PI;
  };
  const int F = g.F, P0 = g.P;
  for (int l = 0; l < g.L; ++l) {
    BN_REQUIRE(P->trunk_w[l] && P->trunk_b[l], "pack: trunk layer %d missing", l);
    if (l == 0) {
      add(P->trunk_w[l], pl.fwd_trunk[l][0], P0, F, P0, 0, 0, 0);
      a.job[a.n_jobs - 1].K_pad = g.KP;                 // the PE operand is padded to 4 MFMA k-steps
      fwd_scale(1, 30.f);
    } else if (l == g.skip) {
      add(P->trunk_w[l], pl.fwd_trunk[l][0], F + P0, F, P0, 0, 0, 0);
      a.job[a.n_jobs - 1].K_pad = g.KP;
      add(P->trunk_w[l], pl.fwd_trunk[l][1], F + P0, F, F, 0, P0, 0);
      fwd_scale(2, 1.f);
    } else {
      add(P->trunk_w[l], pl.fwd_trunk[l][0], F, F, F, 0, 0, 0);
      fwd_scale(1, 1.f);
    }
    if (l >= 1) {  // W_l^T over the h inputs: packed[row j][k n] = W[n][j (+P0 at the skip layer)]
      const int ld = l == g.skip ? F + P0 : F;
      add(P->trunk_w[l], pl.bwd_trunk[l], ld, F, F, l == g.skip ? P0 : 0, 0, 1);
    }
  }
  if (!g.fold) {
    BN_REQUIRE(P->feats_w && P->feats_b, "pack: feats layer missing");
    add(P->feats_w, pl.fwd_feats, F, F, F, 0, 0, 0);
    add(P->feats_w, pl.bwd_feats, F, F, F, 0, 0, 1);
  }
  // the one-row sigma head and the three-row learned-normal head as 32-row MFMA tiles (rows beyond the head's are zero)
  BN_REQUIRE(P->sigma_w, "pack: sigma head missing");
  add(P->sigma_w, pl.fwd_sigma, F, 1, F, 0, 0, 0);
  if (g.ch_normal_lr >= 0) {
    BN_REQUIRE(P->normal_w, "pack: learned-normal head missing");
    add(P->normal_w, pl.fwd_nlr, F, 3, F, 0, 0, 0);
  }
  // (W_l[:, :P])^T for the analytic-normal adjoint: packed[row p][k n] = W_l[n][p]
  add(P->trunk_w[0], pl.bwd_pe[0], P0, P0, F, 0, 0, 1);
  a.job[a.n_jobs - 1].rows_pad = g.KP;
  if (g.skip > 0) {
    add(P->trunk_w[g.skip], pl.bwd_pe[1], F + P0, P0, F, 0, 0, 1);
    a.job[a.n_jobs - 1].rows_pad = g.KP;
  }
  for (int p = 0; p < g.n_pass; ++p)
    for (int hl = 0; hl < g.pass_heads[p]; ++hl) {
      const int hd = 2 * p + hl;
      BN_REQUIRE(P->head_w1[hd] && P->head_b1[hd] && P->head_w2[hd] && P->head_b2[hd], "pack: head %d missing", hd);
      // forward: rows of the pass = [head 2p rows | head 2p+1 rows]; row tiles of a head are contiguous in the pass
      add(P->head_w1[hd], pl.fwd_head[p] + (size_t)hl * g.H2 * F, F, g.H2, F, 0, 0, 0);
      fwd_scale(1, 1.f);
    }
  // extra-input columns of pass 0's first layers, one [pass_N[0]][KD] block: k in [0, KT0) belongs to the view direction (rows
  // of head 0 = the rgb head's direction columns, other rows zero), k in [KT0, KD) to the image embedding (rows of head 1 = the
  // beta head's embedding columns, other rows zero)
  auto xin = [&](const float *src, int64_t ld, int row_lo, int k_lo, int K, int K_touch) {
    add(src, pl.fwd_dir, (int)ld, g.H2, K, 0, 0, 0);
    PackJob &j = a.job[a.n_jobs - 1];
    j.rows_pad = g.pass_N[0]; j.K_pad = g.KD; j.masked = 1; j.k_lo = k_lo; j.K_touch = K_touch; j.row_lo = row_lo;
    if (!src) j.rows = 0;   // zero fill only
    fwd_scale(1, 1.f);
  };
  if (g.KD > 0) {
    BN_REQUIRE(g.DD == 0 || (P->head0_wdir && P->head0_wdir_ld >= g.DD), "pack: head0_wdir missing (input_viewdir)");
    BN_REQUIRE(g.TD == 0 || (P->head1_wt && P->head1_wt_ld >= g.TD), "pack: head1_wt missing (beta)");
    if (g.KT0 > 0) xin(P->head0_wdir, P->head0_wdir_ld, 0, 0, g.DD, g.KT0);
    xin(g.TD > 0 ? P->head1_wt : nullptr, P->head1_wt_ld, g.H2, g.KT0, g.TD, g.KD - g.KT0);
  }
  // transposed head-1 weights: packed[row j][k = column in pass] = W1_hd[k - hl*H2][j]; the heads of a pass
  // interleave along k, so each head fills its own k range of the shared packed matrix (masked job).
  for (int p = 0; p < g.n_pass; ++p)
    for (int hl = 0; hl < g.pass_heads[p]; ++hl) {
      PackJob &j = a.job[a.n_jobs++];
      j.src = P->head_w1[2 * p + hl]; j.dst = pl.bwd_head[p]; j.ld = F; j.rows = F; j.K = g.H2;
      j.rows_pad = F; j.K_pad = g.pass_N[p]; j.row_off = 0; j.col_off = 0; j.transposed = 1;
      j.k_lo = hl * g.H2; j.masked = 1; j.scale = 1.f; j.K_touch = j.K; j.row_lo = 0;
    }
  BN_REQUIRE(a.n_jobs <= BN_MAX_PACK_JOBS, "pack: too many jobs");
  dim3 grid(64, a.n_jobs);
  BnProfScope prof_(BN_K_PACK, (hipStream_t)stream);
  if (desc->dtype == BN_BF16) pack_kernel<bf16><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  else if (desc->dtype == BN_F16) pack_kernel<f16><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  else pack_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(a);
  BN_LAUNCH_CHECK("bn_pack_field");
  return 0;
}

template <typename T, int MT, int NT, int WAVES, bool KEEP, bool DIR, bool D16> static int launch_fwd_k(const FwdArgs &a, int64_t tiles, hipStream_t st);
template <typename T, int MT, int NT, int WAVES> static int launch_fwd(const FwdArgs &a, int64_t tiles, hipStream_t st) {
  // fp16 mode of a model with analytic normals: the derivative stash in fp16 (DK16, FieldGeom.dsz == 2); only a stash-writing
  // forward has a derivative stash at all
  if constexpr (std::is_same<T, f16>::value) {
    if (a.g.dsz == 2 && a.stash) {
      if (a.g.KD > 0 && !a.sigma_only) return launch_fwd_k<T, MT, NT, WAVES, true, true, true>(a, tiles, st);
      return launch_fwd_k<T, MT, NT, WAVES, true, false, true>(a, tiles, st);
    }
  }
  if (a.g.KD > 0 && !a.sigma_only)      // --input_viewdir / --beta: the variant with the extra-input segment in pass 0
    return a.stash ? launch_fwd_k<T, MT, NT, WAVES, true, true, false>(a, tiles, st) : launch_fwd_k<T, MT, NT, WAVES, false, true, false>(a, tiles, st);
  return a.stash ? launch_fwd_k<T, MT, NT, WAVES, true, false, false>(a, tiles, st) : launch_fwd_k<T, MT, NT, WAVES, false, false, false>(a, tiles, st);
}
template <typename T, int MT, int NT, int WAVES, bool KEEP, bool DIR, bool D16> static int launch_fwd_k(const FwdArgs &a, int64_t tiles, hipStream_t st) {
  constexpr int BM = MT * 32;
  const size_t lds = ((size_t)BM * (a.g.F + Elem<T>::kPad) + (size_t)BM * (a.g.KP + Elem<T>::kPad)) * sizeof(T) +
                     (size_t)WAVES * 3 * BM * sizeof(float);
  if (int e = bn_configure_lds((const void *)field_fwd_kernel<T, MT, NT, WAVES, KEEP, DIR, D16>, lds, "field_fwd")) return e;
  {
    BnProfScope prof_(a.sigma_only ? BN_K_FWD_SIGMA : BN_K_FWD_FULL, st);
    field_fwd_kernel<T, MT, NT, WAVES, KEEP, DIR, D16><<<dim3((unsigned)tiles), WAVES * 64, lds, st>>>(a);
    BN_LAUNCH_CHECK("field_fwd");
  }
  // mirror the device fault word to the host now and then: asynchronous, no synchronisation on the hot path
  // (never inside a stream capture: a graph would replay the copy with every launch)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
  if (cap == hipStreamCaptureStatusNone && (g_fwd_launches++ & 63u) == 0u) {
    if (!g_fault_host && hipHostMalloc((void **)&g_fault_host, 2 * sizeof(unsigned int), hipHostMallocDefault) == hipSuccess)
      g_fault_host[0] = g_fault_host[1] = 0u;
    if (g_fault_host) {
      (void)hipMemcpyFromSymbolAsync(g_fault_host, HIP_SYMBOL(g_fwd_fault), sizeof(unsigned int), 0, hipMemcpyDeviceToHost, st);
      // the backward trunk's word rides along (every step launches a forward): both fail the next bn_field_* call
      if (const unsigned int *b = bn_bwd_fault_ptr()) (void)hipMemcpyAsync(g_fault_host + 1, b, sizeof(unsigned int), hipMemcpyDeviceToHost, st);
    }
  }
  return 0;
}

// 0: no fault seen so far (asynchronous view: what the last mirrored copy showed); bit 0 forward trunk, bit 1 backward trunk
int bn_field_fault_seen() { return g_fault_host ? (g_fault_host[0] != 0u ? 1 : 0) | (g_fault_host[1] != 0u ? 2 : 0) : 0; }

unsigned int bn_bwd_fault_read(hipStream_t st);   // field_bwd.hip: non-zero = a hand-over of the barrier-free backward trunk was lost

extern "C" int bn_device_faults(unsigned int *faults, void *stream) {
  BN_REQUIRE(faults, "device_faults: null argument");
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess ||
      hipMemcpyFromSymbol(faults, HIP_SYMBOL(g_fwd_fault), sizeof(unsigned int), 0, hipMemcpyDeviceToHost) != hipSuccess) {
    bn_set_error("device_faults: cannot read the fault word");
    return BN_ELAUNCH;
  }
  const unsigned int b = bn_bwd_fault_read((hipStream_t)stream);
  if (b & 0x80000000u) { bn_set_error("device_faults: cannot read the backward fault word"); return BN_ELAUNCH; }
  *faults = (*faults != 0u ? 1u : 0u) | (b ? 2u : 0u);      // bit 0: forward trunk hand-over lost; bit 1: backward trunk hand-over lost
  return 0;
}

int bn_field_forward_impl(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                          const bn_points *pts, float *out, void *stash, int sigma_only, void *stream) {
  FwdArgs a;
  if (int e = bn_make_geom(desc, &a.g)) return e;
  if (const int f = bn_field_fault_seen()) {
    bn_set_error("field_forward: an earlier %s launch lost an LDS hand-over (pp_wait timed out): its results are invalid",
                 (f & 1) ? "forward" : "backward");
    return BN_ELAUNCH;
  }
  // desc->normal_an only reserves 3 output channels here; bn_field_normals() fills them from the stash
  BN_REQUIRE(!desc->normal_an || sigma_only || stash, "field_forward: analytic normals need the activation stash");
  BN_REQUIRE(pts && pts->n_points > 0 && (pts->xyz || (pts->rays && pts->z && pts->n_samples > 0)), "field: bad points");
  BN_REQUIRE(!desc->dir_dim || sigma_only || !pts->xyz || pts->dirs, "field: input_viewdir needs pts.dirs with the xyz point form");
  BN_REQUIRE(!desc->dir_dim || pts->xyz || pts->ray_stride >= 6, "field: input_viewdir needs rays with directions");
  BN_REQUIRE(!desc->t_dim || sigma_only || pts->t_embed, "field: the beta head needs pts.t_embed");
  BN_REQUIRE(packed && out, "field: null buffer");
  a.d = *desc; a.p = *params; a.packed = packed; a.pts = *pts; a.out = out; a.stash = (char *)stash;
  a.sigma_only = sigma_only;
  bn_make_packed_layout(a.g, &a.pl);
  const int BM = a.g.BM;
  BN_REQUIRE(pts->point_offset >= 0 && pts->point_offset % BM == 0 &&
                 (pts->total_points == 0 || pts->point_offset + pts->n_points <= pts->total_points) &&
                 (pts->point_offset == 0 || (pts->total_points > 0 && !sigma_only)),
             "field_forward: point_offset=%lld total_points=%lld (offset must be a multiple of %d inside the set)",
             (long long)pts->point_offset, (long long)pts->total_points, BM);
  bn_stash_layout_at(a.g, pts, BM, esize(desc->dtype), &a.sl);
  a.out = out + pts->point_offset * (sigma_only ? 1 : a.g.C);        // rows of this call in the set's output array
  const int64_t tiles = ceil_div64(pts->n_points, BM);
  hipStream_t st = (hipStream_t)stream;
  BN_DISPATCH_TILE(desc->dtype, a.g, launch_fwd, (a, tiles, st));
}

extern "C" int bn_field_sigma(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                              const bn_points *pts, float *sigma, void *stream) {
  return bn_field_forward_impl(desc, params, packed, pts, sigma, nullptr, 1, stream);
}
extern "C" int bn_field_forward(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                                const bn_points *pts, float *out, void *stash, void *stream) {
  return bn_field_forward_impl(desc, params, packed, pts, out, stash, 0, stream);
}
