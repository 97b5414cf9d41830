// PROBE (not part of libbrdfnerf_hip.so; built and timed by profiles/probe_regchain.py).
//
// sigma-only forward of the spsbrdf-nerf trunk (models/spsbrdfnerf.py:636-646 + sigma head :700-705) with the
// activations of a point kept in REGISTERS from the positional encoding to the sigma head:
//   * one wave owns 32 points for the whole chain.  D[out feature][point] = W * Y is the "swapped" GEMM of the product
//     kernels, so the accumulator of output tile n already holds, per lane, the 16 values of ITS point that the next
//     layer's B operand wants: the weights are packed with their k order permuted to the accumulator's row order
//     (feature = 32 n + (e & 3) + 8 (e >> 2) + 16 s + 4 h for element e of k-step 2n + s) and no activation ever
//     goes through LDS; there is no workgroup barrier on the data path between layers.
//   * LDS holds only a ring of weight chunks (4 x 32 KB), filled by LDS-DMA (global_load_lds_dwordx4) two chunks ahead
//     of their use, every wave loading a quarter of each chunk; one s_barrier per chunk (32 MFMAs per wave) publishes it.
//     The weight stream is packed in consumption order, so a chunk is 32 consecutive KB and the ring keeps flowing
//     across layers and across the tiles of a persistent workgroup.
//   * 4 waves (one per SIMD) x 32 points per workgroup, one workgroup per CU.
// F = 512, 8 layers, skip at 4, PE 10 frequencies, sine activations, bf16 operands: the BASELINE shapes only.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

namespace {
constexpr int F = 512, L = 8, SKIP = 4, NT = F / 32, KSH = F / 16, KSP = 4;
constexpr int CHUNK_FRAGS = 32, CHUNK_BYTES = CHUNK_FRAGS * 1024, NSLOT = 4;
constexpr int FRAGS_PER_TILE = NT * KSP + 6 * NT * KSH + NT * (KSP + KSH);   // 3712
constexpr int CHUNKS_PER_TILE = FRAGS_PER_TILE / CHUNK_FRAGS;                 // 116
static_assert(FRAGS_PER_TILE % CHUNK_FRAGS == 0 && CHUNKS_PER_TILE % NSLOT == 0, "stream geometry");
constexpr int LDS_RING = NSLOT * CHUNK_BYTES, LDS_BIAS = L * F * 4, LDS_SW = F * 4;
#ifndef PROBE_NO_LDSREAD
#define PROBE_NO_LDSREAD 0     // 1: ablation, no fragment reads after a layer's first DEPTH
#endif
#ifndef PROBE_DEPTH
#define PROBE_DEPTH 8
#endif
constexpr int DEPTH = PROBE_DEPTH;   // weight fragments in flight (LDS -> registers) per wave

struct Args {
  const float *xyz;      // [M][3]
  int64_t M;
  const bf16 *wstream;   // FRAGS_PER_TILE KB, consumption order, pre-scaled by w0 / (2 pi)
  const float *bias;     // [L][F], pre-scaled
  const float *sw;       // [2][256]: sigma weights in the order lane half h holds the last layer's features
  float sb;
  float *out;            // [M]
  int n_tiles;
  unsigned long long *dbg;   // PROBE_TIMING: [blocks][4 waves][8] cycle buckets
};

// four 1-KB LDS-DMA pieces, 1 KB apart in global memory AND in LDS (the instruction offset applies to both addresses)
__device__ __forceinline__ void glds16x4(const void *gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\t"
               "global_load_lds_dwordx4 %1, off offset:1024\n\t"
               "global_load_lds_dwordx4 %1, off offset:2048\n\t"
               "global_load_lds_dwordx4 %1, off offset:3072\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

struct Stream {
  const bf16 *wl;      // wstream + this wave's quarter + lane * 8
  uint32_t ring;       // LDS byte address of the ring + this wave's quarter
  uint32_t rd_lane;    // LDS byte address of the ring + lane * 16 (fragment reads)
  int cw;              // next chunk to ISSUE, wrapped to [0, CHUNKS_PER_TILE)
  int cg;              // next chunk to FENCE (its slot = cg & 3)
#ifdef PROBE_TIMING
  unsigned long long tf = 0;
#endif
  __device__ __forceinline__ void issue() {
    const bf16 *src = wl + (size_t)cw * (CHUNK_BYTES / 2);
    const uint32_t dst = ring + (uint32_t)((cg + 2) & (NSLOT - 1)) * CHUNK_BYTES;
    glds16x4(src, dst);
    glds16x4(src + 4 * 512, dst + 4 * 1024);
    cw = cw + 1 == CHUNKS_PER_TILE ? 0 : cw + 1;
  }
  // before the first read of chunk cg: it has landed for every wave, and everybody is past chunk cg - 2
  __device__ __forceinline__ uint32_t fence() {
#ifdef PROBE_TIMING
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
#ifndef PROBE_NO_DMA     // ablations (results wrong): no weight stream / no per-chunk barrier
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
#ifndef PROBE_NO_BARRIER
    asm volatile("s_barrier" ::: "memory");
#endif
#ifdef PROBE_TIMING
    tf += __builtin_amdgcn_s_memtime() - t0_;
#endif
#ifndef PROBE_NO_DMA
    issue();
#endif
    const uint32_t base = rd_lane + (uint32_t)(cg & (NSLOT - 1)) * CHUNK_BYTES;
    ++cg;
    return base;
  }
};

__device__ __forceinline__ bf16x8 lds_frag(uint32_t addr) {
  return *(const __attribute__((address_space(3))) bf16x8 *)(uintptr_t)addr;
}
__device__ __forceinline__ f32x4 lds_f4(uint32_t addr) {
  return *(const __attribute__((address_space(3))) f32x4 *)(uintptr_t)addr;
}

// One layer: KS_PE k-steps over the positional-encoding fragments, then KS_H over the hidden ones, per pair of output
// tiles.  Software pipeline, pinned with sched_barrier: while the matrix pipe works on pair np, the VALU turns the
// accumulators of the PREVIOUS pair into bf16 B fragments, a few values per k-step.  The previous pair of pair 0 is the
// last pair of the previous layer (PEND_IN; its fragments are cur[28..31], first read at hidden k-steps 28..31), and this
// layer's last pair is handed to the next layer the same way (pend).  Accumulators start from the (pre-scaled) bias.
struct Pair { f32x16 a0, a1; };

template <int KS_PE, int KS_H, bool PEND_IN>
__device__ __forceinline__ void layer(Stream &st, const bf16x8 (&pe)[KSP], bf16x8 (&cur)[KSH], bf16x8 (&nxt)[KSH], Pair &pend,
                                      uint32_t bias_l /* LDS address of this layer's biases + 16 h */) {
  constexpr int KS = KS_PE + KS_H, NF = NT * KS;
  static_assert(NF % CHUNK_FRAGS == 0, "a layer is a whole number of chunks");
  constexpr int SPREAD = KS < 16 ? KS : 16;      // k-steps over which the 32 values of the previous pair are converted
  constexpr int PER = 32 / SPREAD;
  bf16x8 a[DEPTH];
  uint32_t rbase = 0;
#pragma unroll
  for (int g = 0; g < DEPTH; ++g) {
    if (g % CHUNK_FRAGS == 0) rbase = st.fence();
    a[g % DEPTH] = lds_frag(rbase + (g % CHUNK_FRAGS) * 1024);
  }
  Pair prev = pend;
#pragma unroll
  for (int np = 0; np < NT / 2; ++np) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b0 = lds_f4(bias_l + ((2 * np) * 32 + 8 * q) * 4), b1 = lds_f4(bias_l + ((2 * np + 1) * 32 + 8 * q) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc0[4 * q + e] = b0[e]; acc1[4 * q + e] = b1[e]; }
    }
    bf16x8 tmp;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int g0 = (np * KS + ks) * 2;
      const bf16x8 b = ks < KS_PE ? pe[ks < KS_PE ? ks : 0] : cur[ks >= KS_PE ? ks - KS_PE : 0];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g0 % DEPTH], b, acc0, 0, 0, 0);
      if (g0 + DEPTH < NF) {
        const int g = g0 + DEPTH;
        if (g % CHUNK_FRAGS == 0) rbase = st.fence();
        if (!PROBE_NO_LDSREAD) a[g % DEPTH] = lds_frag(rbase + (g % CHUNK_FRAGS) * 1024);
      }
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(g0 + 1) % DEPTH], b, acc1, 0, 0, 0);
      if (g0 + 1 + DEPTH < NF) {
        const int g = g0 + 1 + DEPTH;
        if (g % CHUNK_FRAGS == 0) rbase = st.fence();
        if (!PROBE_NO_LDSREAD) a[g % DEPTH] = lds_frag(rbase + (g % CHUNK_FRAGS) * 1024);
      }
      if ((np > 0 || PEND_IN) && ks < SPREAD) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
          const int v = ks * PER + i, t = v >> 4, r = v & 15;
          tmp[r & 7] = (bf16)__builtin_amdgcn_sinf(t ? prev.a1[r] : prev.a0[r]);
          if ((r & 7) == 7) {
            if (np == 0) cur[28 + 2 * t + (r >> 3)] = tmp;
            else nxt[(2 * (np - 1) + t) * 2 + (r >> 3)] = tmp;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    prev.a0 = acc0; prev.a1 = acc1;
  }
  pend = prev;
}

// the last pair of the last layer: no further GEMM to hide under
__device__ __forceinline__ void drain(const Pair &pend, bf16x8 (&cur)[KSH]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (bf16)__builtin_amdgcn_sinf(t ? pend.a1[8 * s + e] : pend.a0[8 * s + e]);
      cur[28 + 2 * t + s] = v;
    }
}

__global__ __launch_bounds__(256, 1) void regchain_sigma_kernel(const Args A) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  float *bias_s = (float *)(smem + LDS_RING);
  float *sw_s = (float *)(smem + LDS_RING + LDS_BIAS);
  for (int i = tid; i < L * F; i += 256) bias_s[i] = A.bias[i];
  for (int i = tid; i < F; i += 256) sw_s[i] = A.sw[i];
  __syncthreads();

  Stream st;
  st.wl = A.wstream + (size_t)wave * 8 * 512 + lane * 8;
  st.ring = lds0 + wave * 8 * 1024;
  st.rd_lane = lds0 + lane * 16;
  st.cw = 0;
  st.cg = -2;
  st.issue();   // chunk 0 -> slot 0
  st.cg = -1;
  st.issue();   // chunk 1 -> slot 1
  st.cg = 0;
  const uint32_t bias0 = lds0 + LDS_RING + 16 * h, sw0 = lds0 + LDS_RING + LDS_BIAS + h * 1024;

#ifdef PROBE_TIMING
  unsigned long long tb[6] = {0, 0, 0, 0, 0, 0}, tl_ = __builtin_amdgcn_s_memtime();
#define TMARK(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tb[i] += n_ - tl_; tl_ = n_; }
#else
#define TMARK(i)
#endif
  // the coordinates of a tile are loaded one tile ahead: by the time they are needed the load is ~100 chunks old and
  // its wait costs no more than the two weight chunks in flight (vmcnt retires in order)
  auto load_xyz = [&](int tile, float (&x)[3]) {
    const int64_t gm = (int64_t)tile * 128 + wave * 32 + j;
    x[0] = x[1] = x[2] = 0.f;
    if (tile < A.n_tiles && gm < A.M) { x[0] = A.xyz[gm * 3]; x[1] = A.xyz[gm * 3 + 1]; x[2] = A.xyz[gm * 3 + 2]; }
  };
  float xn[3];
  load_xyz(blockIdx.x, xn);
  for (int tile = blockIdx.x; tile < A.n_tiles; tile += gridDim.x) {
    const int64_t gm = (int64_t)tile * 128 + wave * 32 + j;
    const float x[3] = {xn[0], xn[1], xn[2]};
    load_xyz(tile + gridDim.x, xn);
    bf16x8 pe[KSP], ya[KSH], yb[KSH];
    // PE feature p = 16 s + 8 h + e in the natural order [sin x3, cos x3] per frequency (models/nerf.py:53-70): both
    // candidates of a lane (h = 0 / 1) have compile-time frequency and component, the lane half selects
#pragma unroll
    for (int s = 0; s < KSP; ++s) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int p = 16 * s + 8 * hh + e, k = p / 6, c = p % 6;
          v[hh] = p < 60 ? __builtin_amdgcn_sinf(fmaf(x[c % 3], (float)(1 << k) * 0.15915494309189535f, c >= 3 ? 0.25f : 0.f)) : 0.f;
        }
        pe[s][e] = (bf16)(h ? v[1] : v[0]);
      }
    }
#pragma unroll
    for (int s = 0; s < KSH; ++s) yb[s] = pe[0];
    Pair pend;
    pend.a0 = pend.a1 = f32x16{0.f};
    TMARK(0)
    layer<KSP, 0, false>(st, pe, yb, ya, pend, bias0);
    TMARK(1)
#ifdef PROBE_ALT
    // layers 1..7 alternate the two register images (no copies): 1,3,5,7 read ya and write yb; 2,4,6 the other way
    for (int i = 0;; ++i) {
      layer<0, KSH, true>(st, pe, ya, yb, pend, bias0 + (2 * i + 1) * F * 4);
      if (i == 3) break;
      if (2 * i + 2 == SKIP) layer<KSP, KSH, true>(st, pe, yb, ya, pend, bias0 + (2 * i + 2) * F * 4);
      else layer<0, KSH, true>(st, pe, yb, ya, pend, bias0 + (2 * i + 2) * F * 4);
    }
#else
    for (int l = 1; l < L; ++l) {
      if (l == SKIP) layer<KSP, KSH, true>(st, pe, ya, yb, pend, bias0 + l * F * 4);
      else layer<0, KSH, true>(st, pe, ya, yb, pend, bias0 + l * F * 4);
#pragma unroll
      for (int s = 0; s < KSH - 4; ++s) ya[s] = yb[s];
    }
#pragma unroll
    for (int s = 0; s < KSH - 4; ++s) yb[s] = ya[s];
#endif
    TMARK(2)
    drain(pend, yb);
    // sigma head: per-lane dot over the 256 features this lane half holds, then the other half
    float ds = 0.f;
#pragma unroll
    for (int q = 0; q < KSH; ++q) {
      const f32x4 wa = lds_f4(sw0 + q * 32), wb = lds_f4(sw0 + q * 32 + 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) ds += (float)yb[q][e] * wa[e] + (float)yb[q][4 + e] * wb[e];
    }
    ds += __shfl_xor(ds, 32);
    if (h == 0 && gm < A.M) {
      const float sraw = ds + A.sb;
      A.out[gm] = sraw > 20.f ? sraw : log1pf(expf(sraw));
    }
    TMARK(3)
  }
#ifdef PROBE_TIMING
  if (lane == 0 && A.dbg) {
    unsigned long long *d = A.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int i = 0; i < 4; ++i) d[i] = tb[i];
    d[4] = st.tf; d[5] = (unsigned long long)st.cg;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA may outlive the workgroup
}
}  // namespace

extern "C" int bn_probe_regchain_sigma(const float *xyz, int64_t M, const void *wstream, const float *bias, const float *sw, float sb,
                                       float *out, int blocks, hipStream_t stream, unsigned long long *dbg) {
  static bool once = false;
  constexpr int lds = LDS_RING + LDS_BIAS + LDS_SW;
  if (!once) {
    if (hipFuncSetAttribute((const void *)regchain_sigma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return 1;
    once = true;
  }
  Args a{xyz, M, (const bf16 *)wstream, bias, sw, sb, out, (int)((M + 127) / 128), dbg};
  if (blocks <= 0) blocks = 256;
  if (blocks > a.n_tiles) blocks = a.n_tiles;
  hipLaunchKernelGGL(regchain_sigma_kernel, dim3(blocks), dim3(256), lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
