// Device building blocks of the fused field MLP: the activation-stationary tile GEMM and the
// accumulator-register <-> (point, feature) maps shared by the forward and backward chain kernels.
//
// Tiling (one workgroup of WAVES = 8 waves per CU, 2 waves per SIMD):
//   * a tile of BM points (128 for bf16, 64 for fp32) keeps its activations in LDS as ACT[BM][F+pad]
//     (row-major, k contiguous) for the whole network;
//   * products are computed "swapped": D[n][m] = sum_k W[n][k] * ACT[m][k], i.e. the WEIGHTS are the MFMA
//     A operand (rows n) streamed from L2 in pre-packed fragment order (one coalesced 1 KB load per
//     32x16 block, no LDS), and the ACTIVATIONS are the B operand read from LDS with ds_read_b128;
//   * wave w owns output features [w*32*NTW, (w+1)*32*NTW) for all BM points: acc[NTW][MT] 32x32 tiles.
//     In the accumulator a lane owns ONE point (m = mt*32 + (lane&31)) and 4 runs of 4 consecutive features
//     (n = 8g + 4h + e), so activations are written back to ACT[m][n..n+3] with one 8-byte store.
#pragma once
#include "field.h"

// Phase-cycle instrumentation, compiled only into the diagnostic library built by profiles/phase_timing.py
// (-DBN_PHASE_TIMING): per-wave shader-clock cycles spent between BN_PH() marks, summed over the grid.
#ifdef BN_PHASE_TIMING
#define BN_PH_N 16
static __device__ unsigned long long bn_phase_clk[BN_PH_N + 1];
#define BN_PH_DEFINE_READER(NAME)                                                                              \
  extern "C" int NAME(unsigned long long *out, int reset) {                                                    \
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(bn_phase_clk), sizeof(unsigned long long) * (BN_PH_N + 1)) != hipSuccess) return -1; \
    if (reset) {                                                                                               \
      unsigned long long z[BN_PH_N + 1] = {0};                                                                 \
      if (hipMemcpyToSymbol(HIP_SYMBOL(bn_phase_clk), z, sizeof(z)) != hipSuccess) return -1;                  \
    }                                                                                                          \
    return 0;                                                                                                  \
  }
#define BN_PH_DECL unsigned long long ph_[BN_PH_N] = {0}, pt_ = __builtin_readcyclecounter();
#define BN_PH(i) { const unsigned long long n_ = __builtin_readcyclecounter(); ph_[i] += n_ - pt_; pt_ = n_; }
#define BN_PH_FLUSH if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < BN_PH_N; ++i_) atomicAdd(&bn_phase_clk[i_], ph_[i_]); atomicAdd(&bn_phase_clk[BN_PH_N], 1ull); }
#else
#define BN_PH_DEFINE_READER(NAME)
#define BN_PH_DECL
#define BN_PH(i)
#define BN_PH_FLUSH
#endif

// In-kernel clock stamps, compiled only into the diagnostic library built by profiles/clock_probe.py (-DBN_CLOCK_STAMP):
// wave 0 of every workgroup stamps s_memtime (shader clock) and s_memrealtime (100 MHz) at entry and exit; the quotient of
// the two differences is the clock the chip holds under this kernel (MI355X_MICROARCH.md, DVFS give-back item 6).  The stamps
// go to a buffer nothing else reads; no output depends on them.
#ifdef BN_CLOCK_STAMP
#define BN_CLK_N 8192
#define BN_CLK_DEFINE(NAME)                                                                                    \
  static __device__ unsigned long long bn_clk_buf[BN_CLK_N][2];                                               \
  extern "C" int NAME(unsigned long long *out, int n) {                                                        \
    if (n > BN_CLK_N) n = BN_CLK_N;                                                                            \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bn_clk_buf), sizeof(unsigned long long) * 2 * n) == hipSuccess ? 0 : -1; \
  }
#define BN_CLK_BEGIN const unsigned long long clk0_ = __builtin_amdgcn_s_memtime(), clk1_ = __builtin_amdgcn_s_memrealtime();
#define BN_CLK_END                                                                                             \
  if (threadIdx.x == 0 && blockIdx.x < BN_CLK_N) {                                                             \
    bn_clk_buf[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - clk0_;                                          \
    bn_clk_buf[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - clk1_;                                      \
  }
#else
#define BN_CLK_DEFINE(NAME)
#define BN_CLK_BEGIN
#define BN_CLK_END
#endif

// Stash traffic is streaming (written once here, read once by a later kernel) and several times larger than the
// packed weights every workgroup re-reads from L2: non-temporal stores / loads keep it from evicting the weights.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <typename V> __device__ __forceinline__ void stash_store(V *p, const V &v) {
#ifdef BN_SKIP_STASH_STORES   // diagnostic variant (profiles/ab_bench.sh): what the stash writes cost (results are wrong)
  if (p == nullptr) *p = v;
#elif defined(BN_NO_NT_STASH)
  *p = v;
#else
  __builtin_nontemporal_store(v, p);
#endif
}
template <typename V> __device__ __forceinline__ V stash_load(const V *p) {
#ifdef BN_NO_NT_STASH
  return *p;
#else
  return __builtin_nontemporal_load(p);
#endif
}

// Instantiate-and-call FN<T, MT, NT, WAVES> ARGS for the tile configuration in geometry G (returns from the caller).
#define BN_DISPATCH_TILE(DTYPE, G, FN, ARGS)                                                     \
  do {                                                                                           \
    if ((DTYPE) == BN_BF16) {                                                                    \
      if ((G).NT == 2) return FN<bf16, 4, 2, 8> ARGS;                                            \
      return FN<bf16, 4, 1, 8> ARGS;                                                             \
    }                                                                                            \
    if ((DTYPE) == BN_F16) {                                                                     \
      if ((G).NT == 2) return FN<f16, 4, 2, 8> ARGS;                                             \
      return FN<f16, 4, 1, 8> ARGS;                                                              \
    }                                                                                            \
    if ((G).NT == 2) return FN<float, 2, 2, 8> ARGS;                                             \
    return FN<float, 2, 1, 8> ARGS;                                                              \
  } while (0)


template <typename T> __device__ __forceinline__ typename Elem<T>::frag lds_frag(const T *p);
template <> __device__ __forceinline__ bf16x8 lds_frag<bf16>(const bf16 *p) { return *(const bf16x8 *)p; }
template <> __device__ __forceinline__ f16x8 lds_frag<f16>(const f16 *p) { return *(const f16x8 *)p; }
template <> __device__ __forceinline__ f32x8 lds_frag<float>(const float *p) {
  f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
  f32x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return r;
}
template <typename T> __device__ __forceinline__ typename Elem<T>::frag gld_frag(const T *p) { return lds_frag<T>(p); }

// acc[nt][mt] += W_packed(this wave's tiles) x B(lds).  `wp` points at the packed block of the wave's first
// n-tile for this K segment; consecutive n-tiles are KS*512 elements apart.
//
// Software pipeline: the weight fragments of DEPTH consecutive k-steps are always in flight from L2 (a slot is
// re-filled right after its MFMAs have issued and is consumed DEPTH k-steps later), and the LDS fragments of k-step
// s+1 are read while the MFMAs of k-step s run.  sched_barrier pins that order (hipcc otherwise sinks the prefetch
// loads next to their use).  The weight stream out of L2 is what bounds these kernels: DEPTH
// sets the bytes in flight per CU (8 waves x DEPTH x NTW KB), and no fragment is fetched twice.
// Side job of a GEMM segment: at(parity) is called once per k-step after that step's MFMAs have issued; the parity of
// the k-step is a constant once the k-loop is unrolled (the pipeline depth is even).  A side job must be straight-line code - a branch inside the
// unrolled k-loop splits its basic block and the waits on the weight stream turn conservative.
struct NoSide {
  __device__ __forceinline__ void at(int) const {}
};

// Row-major stash copy of the workgroup's LDS tile [rows][width] riding inside the NEXT GEMM over the same tile (which
// only reads it): one 16-byte chunk per thread every second k-step, so the stores drain under the MFMAs and no load
// the kernel is about to wait for sits behind a burst of them (vmcnt retires loads and stores in issue order).
// Exact fit only: rows * width / (16 B) == nthr * KS / 2 and nthr % (chunks per row) == 0 (F = 256 or 512 with all
// eight waves in the GEMM) - other shapes use the stand-alone tile_to_global.
template <typename T> struct TileCopyExact {
  static constexpr int EPC = 16 / sizeof(T);
  const T *lp;
  T *gp;
  int lstep, gstep;
  __device__ __forceinline__ TileCopyExact(const T *lds, int ld, T *g, int gld, int width, int t, int nthr) {
    const int cpr = width / EPC, row = t / cpr, cc = t % cpr, dr = nthr / cpr;
    lp = lds + (size_t)row * ld + cc * EPC;
    gp = g + (size_t)row * gld + cc * EPC;
    lstep = dr * ld; gstep = dr * gld;
  }
  __device__ __forceinline__ void at(int parity) {
    if (parity == 0) {
      stash_store((u32x4 *)gp, *(const u32x4 *)lp);
      lp += lstep; gp += gstep;
    }
  }
};
__host__ __device__ __forceinline__ bool tile_copy_exact(int F, int n_on, int waves) { return n_on == waves && F % 256 == 0; }

// Ping-pong form (F = 512, 8 waves in two groups of four): a group copies ITS half (halfw columns from col0) of the
// tile, one chunk per thread at every k-step of its 16-step own-half segment (256 threads x 16 chunks = the half tile
// for both element types: 128 rows x 512 B in bf16, 64 rows x 1 KB in fp32).
template <typename T> struct TileCopyHalf {
  static constexpr int EPC = 16 / sizeof(T);
  const T *lp;
  T *gp;
  int lstep, gstep;
  __device__ __forceinline__ TileCopyHalf(const T *lds, int ld, T *g, int gld, int col0, int halfw, int tg) {
    const int cph = halfw / EPC, row = tg / cph, col = col0 + (tg % cph) * EPC, rstep = 256 / cph;
    lp = lds + (size_t)row * ld + col;
    gp = g + (size_t)row * gld + col;
    lstep = rstep * ld; gstep = rstep * gld;
  }
  __device__ __forceinline__ void at(int) {
    stash_store((u32x4 *)gp, *(const u32x4 *)lp);
    lp += lstep; gp += gstep;
  }
};

// Flags of the two-group ping-pong (LDS ints, zeroed before first use): counters only grow; a waiter spins with
// s_sleep until the count is reached.  LDS executes one wave's operations in issue order, so data written before a
// signal is visible to whoever sees the signal.  The spin is bounded (a count that is never reached in a correct run):
// every wave reaches its exit even if a signal were lost.
typedef __attribute__((address_space(3))) int lds_int;
// A hand-over that never arrives (impossible in a correct run) must not pass silently: the waiter gives up after ~2^24
// sleeps, records the fault in a device word (*fault, owned by the translation unit that launches the kernel) and carries on
// to its exit; the launch wrapper mirrors that word to pinned host memory and the next library call returns BN_ELAUNCH
// (bn_device_faults() reads it synchronously).
__device__ __forceinline__ void pp_wait(int *flag, int target, unsigned int *fault) {
  volatile lds_int *f = (volatile lds_int *)flag;
  int spin = 0;
  for (; *f < target && spin < (1 << 24); ++spin) __builtin_amdgcn_s_sleep(1);
  if (spin >= (1 << 24) && (threadIdx.x & 63) == 0) atomicOr(fault, 1u);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pp_signal(int *flag, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add((lds_int *)flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

#ifndef BN_DEPTH_BF16
#define BN_DEPTH_BF16 4
#endif
template <typename T> struct PipeDepth { static constexpr int value = 4; };
template <> struct PipeDepth<bf16> { static constexpr int value = BN_DEPTH_BF16; };
template <> struct PipeDepth<f16> { static constexpr int value = BN_DEPTH_BF16; };

template <typename T, int MT, int NTW, typename Side>
__device__ __forceinline__ void gemm_range(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, int ks0, int nks, const T *bsrc,
                                           int ldb, int lane, Side &side) {
  // k-steps [ks0, ks0 + nks) of a packed matrix whose n-tiles are KS k-steps apart
  typedef typename Elem<T>::frag frag;
  constexpr int DEPTH = PipeDepth<T>::value;
  static_assert(DEPTH % 2 == 0, "side jobs rely on an even pipeline depth");
  const int r = lane & 31, h = lane >> 5;
  const T *wl = wp + (size_t)lane * 8;
  const T *bl = bsrc + (size_t)r * ldb + 8 * h;
  const int kend = ks0 + nks;
  frag A[DEPTH][NTW], Bc[MT];
  auto loadA = [&](frag(&a)[NTW], int ks) {
#ifdef BN_SKIP_A       // diagnostic variant (profiles/ab_bench.sh): no weight stream after the prologue
    if (ks >= ks0 + DEPTH) return;
#endif
    ks = ks < kend ? ks : kend - 1;
    {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) a[nt] = gld_frag<T>(wl + ((size_t)nt * KS + ks) * 512);
    }
  };
  auto loadB = [&](frag(&B)[MT], int ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) B[mt] = lds_frag<T>(bl + (size_t)mt * 32 * ldb + ks * 16);
  };
#ifdef BN_AB_MFMA16
  // Ablation (results WRONG by construction; profiles/ab_bench.sh): what would v_mfma_f32_16x16x32_bf16 buy through the
  // clock the chip holds (MI355X_MICROARCH.md, DVFS give-back item 7)?  The same two operand registers go to two 16x16x32
  // MFMAs per 32x32x16 one - equal FLOPs, equal matrix-pipe cycles, equal operand and accumulator register traffic per
  // FLOP; every accumulator quad is written every second k-step - on accumulator quads held as separate 4-register values
  // inside the loop (copied in and out around it), before fragment / accumulator / stash layouts are re-plumbed for it.
  constexpr bool AB16 = std::is_same<T, bf16>::value;
  f32x4 q[NTW][MT][4];
  if (AB16) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) q[nt][mt][i] = f32x4{acc[nt][mt][4 * i], acc[nt][mt][4 * i + 1], acc[nt][mt][4 * i + 2], acc[nt][mt][4 * i + 3]};
  }
#endif
  auto step = [&](frag(&a)[NTW], int ks, int par) {   // consumes Bc (fragments of k-step ks), leaves those of ks+1 in Bc
    frag Bn[MT];
#ifdef BN_SKIP_B       // diagnostic variant: no LDS fragment reads after the first
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) Bn[mt] = Bc[mt];
#else
    loadB(Bn, ks + 1 < kend ? ks + 1 : ks0);
#endif
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#ifdef BN_AB_MFMA16
        if constexpr (AB16) {
          q[nt][mt][2 * par] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], Bc[mt], q[nt][mt][2 * par], 0, 0, 0);
          q[nt][mt][2 * par + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Bc[mt], a[nt], q[nt][mt][2 * par + 1], 0, 0, 0);
          continue;
        }
#endif
        mma32(acc[nt][mt], a[nt], Bc[mt]);
      }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) Bc[mt] = Bn[mt];
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) loadA(A[d], ks0 + d);
  loadB(Bc, ks0);
  __builtin_amdgcn_sched_barrier(0);
  int ks = ks0;
  for (; ks + DEPTH <= kend; ks += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      step(A[d], ks + d, d & 1);
      loadA(A[d], ks + d + DEPTH);
      side.at(d & 1);   // constant after unrolling
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (ks + d < kend) {
      step(A[d], ks + d, d & 1);
      side.at(d & 1);
    }
#ifdef BN_AB_MFMA16
  if (AB16) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][mt][i] = q[nt][mt][i >> 2][i & 3];
  }
#endif
}
template <typename T, int MT, int NTW, typename Side>
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, const T *bsrc, int ldb,
                                         int lane, Side &side) {
  gemm_range<T, MT, NTW, Side>(acc, wp, KS, 0, KS, bsrc, ldb, lane, side);
}
template <typename T, int MT, int NTW>
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, const T *bsrc, int ldb,
                                         int lane) {
  NoSide none;
  gemm_range<T, MT, NTW, NoSide>(acc, wp, KS, 0, KS, bsrc, ldb, lane, none);
}

template <int MT, int NTW> __device__ __forceinline__ void zero_acc(f32x16 (&acc)[NTW][MT]) {
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nt][mt][i] = 0.f;
}

// Offset (elements) of the 8-element chunk (nt, mt, gp) of this lane in a "native" stash image of one tile:
// chunk gp holds accumulator registers 8*gp .. 8*gp+7 (feature runs 16*gp + 4h + {0..3} and 16*gp + 8 + 4h + {0..3}).
// One wave instruction moves 64 lanes x 16 B (bf16) fully coalesced.
template <int MT, int NTW> __device__ __forceinline__ size_t native_off8(int wave, int nt, int mt, int gp, int lane) {
  return ((((size_t)(wave * NTW + nt) * MT + mt) * 2 + gp) * 64 + lane) * 8;
}
// 8 floats -> one 16-bit fragment (round to nearest even), and the store of a ready-made fragment
__device__ __forceinline__ bf16x8 cvt8(bf16, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
  return o;
}
__device__ __forceinline__ f16x8 cvt8(f16, const float (&v)[8]) {
  const f32x8 f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  return __builtin_convertvector(f, f16x8);
}
__device__ __forceinline__ f32x8 cvt8(float, const float (&v)[8]) {
  const f32x8 f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  return f;
}
__device__ __forceinline__ void st_frag(bf16 *p, const bf16x8 &o) { stash_store((bf16x8 *)p, o); }
__device__ __forceinline__ void st_frag(f16 *p, const f16x8 &o) { stash_store((f16x8 *)p, o); }
__device__ __forceinline__ void st_frag(float *p, const f32x8 &o) {
  stash_store((f32x4 *)p, f32x4{o[0], o[1], o[2], o[3]});
  stash_store((f32x4 *)(p + 4), f32x4{o[4], o[5], o[6], o[7]});
}
__device__ __forceinline__ void st8(bf16 *p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
  stash_store((bf16x8 *)p, o);
}
__device__ __forceinline__ void st8(f16 *p, const float (&v)[8]) {
  const f32x8 f = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  stash_store((f16x8 *)p, __builtin_convertvector(f, f16x8));
}
__device__ __forceinline__ void st8(float *p, const float (&v)[8]) {
  stash_store((f32x4 *)p, f32x4{v[0], v[1], v[2], v[3]});
  stash_store((f32x4 *)(p + 4), f32x4{v[4], v[5], v[6], v[7]});
}
__device__ __forceinline__ void ld8(const bf16 *p, float (&v)[8]) {
  const bf16x8 o = stash_load((const bf16x8 *)p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)o[i];
}
__device__ __forceinline__ void ld8(const f16 *p, float (&v)[8]) {
  const f16x8 o = stash_load((const f16x8 *)p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)o[i];
}
__device__ __forceinline__ void ld8(const float *p, float (&v)[8]) {
  const f32x4 a = stash_load((const f32x4 *)p), b = stash_load((const f32x4 *)(p + 4));
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}

// Copy the workgroup's LDS tile [rows][width] (row stride ld) to a row-major global array [rows][gld] in 16-byte
// chunks: every wave instruction writes 1 KB of consecutive bytes (full 128-B lines) instead of the 32 x 16 B
// fragments the accumulator layout would give.
template <typename T> __device__ __forceinline__ void tile_to_global(const T *lds, int ld, T *g, int gld, int rows, int width) {
  constexpr int EPC = 16 / sizeof(T);
  const int cpr = width / EPC, nthr = blockDim.x;
  // chunk c = row * cpr + cc, walked with stride nthr: one division per call, not one per chunk
  int row = threadIdx.x / cpr, cc = threadIdx.x % cpr;
  const int dr = nthr / cpr, dc = nthr % cpr;
  while (row < rows) {
    stash_store((u32x4 *)(g + (size_t)row * gld + cc * EPC), *(const u32x4 *)(lds + (size_t)row * ld + cc * EPC));
    row += dr; cc += dc;
    if (cc >= cpr) { cc -= cpr; ++row; }
  }
}
