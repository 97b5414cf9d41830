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

// Per-wave event timeline, compiled only into the diagnostic library built by profiles/simd_timeline.py (-DBN_TIMELINE):
// every wave of the first BN_TL_BLOCKS workgroups stamps s_memtime at the phase boundaries of the trunk into an LDS log
// (a region the trunk does not use) and dumps it to a buffer nothing else reads; word 0 of a wave's log is its HW_ID
// (which SIMD it sits on).  No output depends on the stamps; the product build executes none of this.
#ifdef BN_TIMELINE
#define BN_TL_BLOCKS 8
#define BN_TL_EVENTS 112
static __device__ unsigned long long bn_tl_buf[BN_TL_BLOCKS][8][BN_TL_EVENTS];
#define BN_TL_DEFINE_READER(NAME)                                                                              \
  extern "C" int NAME(unsigned long long *out) {                                                               \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(bn_tl_buf), sizeof(unsigned long long) * BN_TL_BLOCKS * 8 * BN_TL_EVENTS) == hipSuccess ? 0 : -1; \
  }
// log = LDS pointer to this wave's BN_TL_EVENTS slots; event code in the top byte
#define BN_TL_DECL(LDSBASE) unsigned long long *tl_log_ = (unsigned long long *)(LDSBASE) + (threadIdx.x >> 6) * BN_TL_EVENTS; int tl_n_ = 1; \
  if ((threadIdx.x & 63) == 0) tl_log_[0] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
#define BN_TL(ev) { if ((threadIdx.x & 63) == 0 && tl_n_ < BN_TL_EVENTS) tl_log_[tl_n_] = ((unsigned long long)(ev) << 56) | (__builtin_amdgcn_s_memtime() & 0x00ffffffffffffffull); ++tl_n_; }
#define BN_TL_DUMP { if (blockIdx.x < BN_TL_BLOCKS && (threadIdx.x & 63) == 0) { const int w_ = threadIdx.x >> 6; const int n_ = tl_n_ < BN_TL_EVENTS ? tl_n_ : BN_TL_EVENTS; \
    for (int i_ = 0; i_ < BN_TL_EVENTS; ++i_) bn_tl_buf[blockIdx.x][w_][i_] = i_ < n_ ? tl_log_[i_] : 0ull; } }
#else
#define BN_TL_DEFINE_READER(NAME)
#define BN_TL_DECL(LDSBASE)
#define BN_TL(ev)
#define BN_TL_DUMP
#endif

// Stash traffic is streaming (written once here, read once by a later kernel) and several times larger than the
// packed weights every workgroup re-reads from L2: non-temporal stores / loads keep it from evicting the weights.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
template <typename V> __device__ __forceinline__ void stash_store(V *p, const V &v) {
#ifdef BN_NO_NT_STASH      // A/B switch (results unchanged): plain stores / loads for the stash
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

// The same stores as buffer instructions: `base` wave-uniform (kernel arguments and blockIdx only), the lane's part in one VGPR
// (voff), everything else in the scalar offset - no 64-bit vector address arithmetic per store (16 stores of a layer's epilogue
// otherwise pin 32 address registers).
// Written as inline asm WITH its wait states: behind `__builtin_amdgcn_raw_buffer_store_b128(..., soffset = an SGPR, ...)` hipcc
// (ROCm 7.2) schedules a VALU write of the store's data registers directly after the store - LLVM's hazard recognizer knows no
// store-data hazard for MUBUF stores with a register soffset - and on gfx950 the first dword of the stored piece then came out
// clobbered now and then: a stash that differed from run to run, found by the fuzz as non-finite gradients
// (profiles/r04_ablation.txt item 5).  hipcc does not count an asm store in its vmcnt bookkeeping; that only makes its waits for
// later loads wait for more than they need.
// (the base goes through readfirstlane: the asm below wants the descriptor in scalar registers whatever the compiler's uniformity
// analysis made of the address arithmetic behind it)
__device__ __forceinline__ auto stash_rsrc(const void *base) {
  const unsigned long long b = (unsigned long long)base;
  const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)b), hi = __builtin_amdgcn_readfirstlane((unsigned int)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
}
template <typename R, typename V> __device__ __forceinline__ void stash_store_buf(R rsrc, int voff, int soff, const V &v) {
  static_assert(sizeof(V) == 16, "one 16-byte piece per lane");
  const u32x4 d = __builtin_bit_cast(u32x4, v);
#ifdef BN_NO_NT_STASH
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(d), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
#else
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(d), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
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
#ifndef BN_PROBE_NO_RIDE      // timing probe only (results wrong): the copy without its global stores
      stash_store((u32x4 *)gp, *(const u32x4 *)lp);
#endif
      lp += lstep; gp += gstep;
    }
  }
};
__host__ __device__ __forceinline__ bool tile_copy_exact(int F, int n_on, int waves) { return n_on == waves && F % 256 == 0; }

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

// Weight-fragment prefetch depth of the chain GEMM (k-steps in flight per wave), chosen per kernel instantiation
// (profiles/history/r02_ablation.txt session 42: the training forward gains 1-1.7 % with 6, the backward / adjoint chains 1-3 % with 2).
// (BN_FWD_DEPTH_TRAIN = 6, BN_BWD_DEPTH = 2: diag.h)
// BN_GEMM_AFFINE added to a depth: the k-loop keeps ONE LDS base address per point tile and block of DEPTH k-steps (4 vector adds
// per block instead of 4 per k-step; 4 more live registers).  Round 4, profiles/r04_ablation.txt item 18: training forward -1.4 %
// (config 3: -3.7 %), backward chain -1.3 % (-2.7 %) - and the inference forward +38 %, the adjoint chains +19 ... 25 % (they
// spill with it): a property of the kernel instantiation, hence carried by its depth constant.
#define BN_GEMM_AFFINE 64
// BN_PP_NKS or-ed into a depth: the number of k-steps is a compile-time constant (bits 8 and up; the caller guarantees nks ==
// that number) and the range is straight-line code - no clamps, no tail steps.  The barrier-free trunks' half-GEMMs (16 k-steps:
// NT == 2 means F = 512) use it: in the looped form their tail steps copied the accumulator set (~70 v_mov_b64 per half-GEMM,
// profiles/r04_isa_scan.txt).  Round 4 ablation item 20, adopted in round 5: training forward -3.8 %, sigma-only forward -8.6 %.
#ifdef BN_PP_LOOP_NKS      // A/B switch (results unchanged): the looped form of rounds 1-4
#define BN_PP_NKS 0
#else
#define BN_PP_NKS (16 << 8)
#endif
template <typename T, bool TRAIN_FWD> struct FwdDepth { static constexpr int value = 4; };
template <> struct FwdDepth<bf16, true> { static constexpr int value = BN_FWD_DEPTH_TRAIN | BN_GEMM_AFFINE; };
template <> struct FwdDepth<f16, true> { static constexpr int value = BN_FWD_DEPTH_TRAIN | BN_GEMM_AFFINE; };
template <typename T> struct BwdDepth { static constexpr int value = Elem<T>::kFastMath ? BN_BWD_DEPTH : 4; };

// LDS (B) fragments: ONE register set (`Bc = Bn` after the MFMAs; hipcc coalesces the two and issues the reads of k-step s + 1
// behind the last MFMA of k-step s that uses the registers).  Two named sets with the reads pinned AHEAD of the MFMAs
// (round 4) measured slower wherever tried: the GEMM alone in a kernel 42 vs 35 cycles per MFMA with one wave
// per SIMD and 78 vs 56 with two (profiles/r04_probe_gemm_rate.txt), the training forward +1 %, the backward chain +5 %
// (profiles/r04_ablation.txt).
template <typename T, int MT, int NTW, int DEPTH_, typename Side>
__device__ __forceinline__ void gemm_range(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, int ks0, int nks, const T *bsrc,
                                           int ldb, int lane, Side &side) {
  // k-steps [ks0, ks0 + nks) of a packed matrix whose n-tiles are KS k-steps apart
  constexpr int DEPTH = DEPTH_ & (BN_GEMM_AFFINE - 1);
  constexpr bool AFFINE = (DEPTH_ & BN_GEMM_AFFINE) != 0;
  constexpr int NKS = DEPTH_ >> 8;      // experiment: a compile-time number of k-steps (the caller guarantees nks == NKS): straight-line code, no tail
  typedef typename Elem<T>::frag frag;
  static_assert(DEPTH % 2 == 0, "side jobs rely on an even pipeline depth");
  const int r = lane & 31, h = lane >> 5;
  const T *wl = wp + (size_t)lane * 8;
  const T *bl = bsrc + (size_t)r * ldb + 8 * h;
  const int kend = ks0 + nks;
  frag A[DEPTH][NTW];
  // (BN_PROBE_NO_A / BN_PROBE_NO_B: profiles/probe_gemm_rate.py only - the loop without its weight / LDS fragment traffic)
  auto loadA = [&](frag(&a)[NTW], int ks) {
    ks = ks < kend ? ks : kend - 1;
    {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
#ifdef BN_PROBE_NO_A
        if (ks >= ks0 + DEPTH) { asm volatile("" : "+v"(a[nt])); continue; }
#endif
        a[nt] = gld_frag<T>(wl + ((size_t)nt * KS + ks) * 512);
      }
    }
  };
  auto loadB = [&](frag(&B)[MT], int ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#ifdef BN_PROBE_NO_B
      if (ks != ks0) { asm volatile("" : "+v"(B[mt])); continue; }
#endif
      B[mt] = lds_frag<T>(bl + (size_t)mt * 32 * ldb + ks * 16);
    }
  };
  frag Bc[MT];
  auto step = [&](frag(&a)[NTW], int ks) {   // consumes Bc (fragments of k-step ks), leaves those of ks+1 in Bc
    frag Bn[MT];
    loadB(Bn, ks + 1 < kend ? ks + 1 : ks0);
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) mma32(acc[nt][mt], a[nt], Bc[mt]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) Bc[mt] = Bn[mt];
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) loadA(A[d], ks0 + d);
  loadB(Bc, ks0);
  __builtin_amdgcn_sched_barrier(0);
  int ks = ks0;
  if constexpr (NKS > 0) {
    const T *bm[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bm[mt] = bl + (size_t)mt * 32 * ldb + (size_t)ks0 * 16;
#pragma unroll
    for (int j = 0; j < NKS; ++j) {
      frag Bn[MT];
      if (j + 1 < NKS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Bn[mt] = lds_frag<T>(bm[mt] + (j + 1) * 16);
      }
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mma32(acc[nt][mt], A[j % DEPTH][nt], Bc[mt]);
      if (j + 1 < NKS) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Bc[mt] = Bn[mt];
      }
      if (j + DEPTH < NKS) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) A[j % DEPTH][nt] = gld_frag<T>(wl + ((size_t)nt * KS + ks0 + j + DEPTH) * 512);
      }
      side.at(j & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  if constexpr (AFFINE) {
  // Whole blocks of DEPTH k-steps: the LDS fragment addresses of a block are ONE base per point tile (advanced once per block)
  // plus compile-time offsets - the look-ahead of a range's last step reads the 16 columns behind the range (row pad / the
  // next columns of the tile: inside the LDS allocation, never used) instead of wrapping to ks0, which made every address a
  // select and cost a vector add per read.
  {
    const T *bm[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) bm[mt] = bl + (size_t)mt * 32 * ldb + (size_t)ks0 * 16;
    for (; ks + DEPTH <= kend; ks += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        frag Bn[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Bn[mt] = lds_frag<T>(bm[mt] + (d + 1) * 16);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) mma32(acc[nt][mt], A[d][nt], Bc[mt]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) Bc[mt] = Bn[mt];
        loadA(A[d], ks + d + DEPTH);
        side.at(d & 1);   // constant after unrolling
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) bm[mt] += DEPTH * 16;
    }
  }
  } else {
  for (; ks + DEPTH <= kend; ks += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      step(A[d], ks + d);
      loadA(A[d], ks + d + DEPTH);
      side.at(d & 1);   // constant after unrolling
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  }
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (ks + d < kend) {
      step(A[d], ks + d);
      side.at(d & 1);
    }
}
// GEMM over a COMPILE-TIME number of k-steps as one straight-line weight stream (round 5): k-steps [ks0, ks0 + NKS) of a packed
// matrix whose n-tiles are KS k-steps apart.  NT == 2 means F = 512, so the full-width products of the hot instantiations have 32
// k-steps, their k-split sigma head 4 per wave.  No clamps, no tail steps (the looped form's tails copied the accumulator set:
// ~70 v_mov_b64 each, profiles/r04_isa_scan.txt).
// The weight fragments are BUFFER loads: descriptor = the wave's packed block (wave-uniform), one lane-offset register for the
// whole kernel, everything else in the scalar offset - the global-load form spent two 64-bit vector adds per k-step on its
// addresses and, unrolled, spilled scalar registers to vector lanes (26 + 42 v_readlane per layer in the ISA of the half-GEMM
// form).  Loads are the compiler's builtin (counted in its vmcnt bookkeeping), not inline asm: the
// store-data hazard of stash_store_buf is a store's.
// MID > 0: `mid()` runs between k-steps MID - 1 and MID - the hand-over of the barrier-free trunks' two column halves (signal
// "done reading half 0", wait for half 1).  The weight ring keeps running through it (rounds 1-4: two half-GEMMs, each with its
// own prologue of DEPTH exposed L2 round trips behind the hand-over wait); only the LDS look-ahead read of step MID waits.
// Session 1 of round 5 (profiles/r05_ab_trunk_stream_lambert.txt): training forward 1.062 -> 1.020 ms, backward chain 2.240 ->
// 2.051 ms, sigma-only forward 1.573 -> 1.530 ms against the half-GEMM form with global loads.
struct NoMid {
  __device__ __forceinline__ void operator()() const {}
};
template <typename T, int MT, int NTW, int DEPTH, int NKS, int MID, typename Mid, typename Side = NoSide>
__device__ __forceinline__ void gemm_fixed(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, int ks0, const T *bsrc, int ldb,
                                           int lane, Mid &&mid, Side &&side = Side()) {
  typedef typename Elem<T>::frag frag;
  static_assert(sizeof(frag) == 16, "16-bit modes only: one 16-byte fragment piece per lane");
  static_assert(MID >= 0 && MID < NKS && DEPTH <= NKS && (MID == 0 || DEPTH <= MID), "the ring must not wrap inside the prologue");
  const int r = lane & 31, h = lane >> 5;
  const T *bm[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) bm[mt] = bsrc + (size_t)(r + 32 * mt) * ldb + 8 * h + (size_t)ks0 * 16;
  const auto rs = stash_rsrc(wp);
  const int voff = lane * 16;
  auto ldA = [&](int nt, int ks) {
#ifdef BN_NO_BUFW      // A/B switch (results unchanged): global loads with 64-bit vector addresses, as in rounds 1-4
    return gld_frag<T>(wp + (size_t)lane * 8 + ((size_t)nt * KS + ks0 + ks) * 512);
#else
    return __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (nt * KS + ks0 + ks) * (512 * (int)sizeof(T)), 0));
#endif
  };
  frag A[DEPTH][NTW], Bc[MT];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) A[d][nt] = ldA(nt, d);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) Bc[mt] = lds_frag<T>(bm[mt]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < NKS; ++j) {
    const bool ahead = j + 1 < NKS && (MID == 0 || j + 1 != MID);      // (compile-time after unrolling)
    frag Bn[MT];
    if (ahead) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) Bn[mt] = lds_frag<T>(bm[mt] + (j + 1) * 16);
    }
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) mma32(acc[nt][mt], A[j % DEPTH][nt], Bc[mt]);
    if (ahead) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) Bc[mt] = Bn[mt];
    }
    if (j + DEPTH < NKS) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) A[j % DEPTH][nt] = ldA(nt, j + DEPTH);
    }
    side.at(j & 1);      // (side job of the segment, gemm_range: a row-major stash copy riding under the MFMAs)
    __builtin_amdgcn_sched_barrier(0);
    if (MID > 0 && j + 1 == MID) {
      mid();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) Bc[mt] = lds_frag<T>(bm[mt] + MID * 16);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
// the barrier-free trunks' layer GEMM: 32 k-steps, hand-over behind the 16th
template <typename T, int MT, int NTW, int DEPTH, int NKS, int MID, typename Mid>
__device__ __forceinline__ void gemm_trunk(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, const T *bsrc, int ldb, int lane,
                                           Mid &&mid) {
  gemm_fixed<T, MT, NTW, DEPTH, NKS, MID>(acc, wp, KS, 0, bsrc, ldb, lane, mid);
}
template <typename T, int MT, int NTW, int DEPTH, typename Side>
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, const T *bsrc, int ldb,
                                         int lane, Side &side) {
  gemm_range<T, MT, NTW, DEPTH, Side>(acc, wp, KS, 0, KS, bsrc, ldb, lane, side);
}
template <typename T, int MT, int NTW, int DEPTH>
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, const T *bsrc, int ldb,
                                         int lane) {
  NoSide none;
  gemm_range<T, MT, NTW, DEPTH, NoSide>(acc, wp, KS, 0, KS, bsrc, ldb, lane, none);
}

// A full-width product under barriers (head passes, the backward's top layer).  HOT = the F = 512 instantiation of a 16-bit mode
// (NT == 2): its k-step counts are 32 (F / 16) or 16 (a single head's 256 hidden columns) - straight-line streams; every other
// shape keeps the looped form.
template <typename T, int MT, int NTW, int DP, bool HOT>
__device__ __forceinline__ void gemm_full(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, const T *bsrc, int ldb, int lane) {
#ifndef BN_NO_FIXED_FULL      // A/B switch (results unchanged): the looped form everywhere outside the trunks (round 4)
  if constexpr (HOT && sizeof(T) == 2) {
    constexpr int D = (DP & (BN_GEMM_AFFINE - 1)) < 2 ? 2 : (DP & (BN_GEMM_AFFINE - 1));
    if (KS == 32) { gemm_fixed<T, MT, NTW, D, 32, 0>(acc, wp, KS, 0, bsrc, ldb, lane, NoMid()); return; }
    if (KS == 16) { gemm_fixed<T, MT, NTW, D, 16, 0>(acc, wp, KS, 0, bsrc, ldb, lane, NoMid()); return; }
  }
#endif
  gemm_seg<T, MT, NTW, DP>(acc, wp, KS, bsrc, ldb, lane);
}
// the F = 512 trunk product (32 k-steps) of the analytic-normal chains with a riding stash copy (field_adjoint / field_adjbwd)
template <typename T, int MT, int NTW, int DP, bool HOT, typename Side>
__device__ __forceinline__ void gemm_full32(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, const T *bsrc, int ldb, int lane,
                                            Side &side) {
#ifndef BN_ADJ_LOOP      // A/B switch (results unchanged): the looped form (round 4); straight-line: adjoint chain -8.6 %, r05 item 7
  if constexpr (HOT && sizeof(T) == 2) {
    constexpr int D = (DP & (BN_GEMM_AFFINE - 1)) < 2 ? 2 : (DP & (BN_GEMM_AFFINE - 1));
    if (KS == 32) { gemm_fixed<T, MT, NTW, D, 32, 0>(acc, wp, KS, 0, bsrc, ldb, lane, NoMid(), side); return; }
  }
#endif
  gemm_seg<T, MT, NTW, DP>(acc, wp, KS, bsrc, ldb, lane, side);
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

// ------------------------------------------------------------------ the derivative stash D_l = d act / d z  (and DG of the heads)
// The backward chains only ever multiply by it element-wise, so it does not have to be an MFMA operand type.  Three kinds
// (common.h DKind<T, D16>):
// DK8 (16-bit modes, default): 8-bit fixed point of the UNSCALED derivative c (cos(.) of a Siren layer, 0 / 1 of a ReLU
// layer): the high byte of the 16-bit signed-normalised value, u = (rne(32767 c) >> 8) + 128 in [0, 255]; the consumer
// multiplies by the layer's w0.  Siren layers decode the middle of the byte's interval, (256 (u - 128) + 128) / 32767: absolute
// error <= 1/256 of the largest derivative, rms 2.3e-3 - the size of the bf16 rounding of the gradient operand it multiplies;
// ReLU layers decode (u - 128) / 127: the mask is exact (1 -> 127, 0 -> 0).  Half the bytes of a 16-bit image: the forward
// writes 3 instead of 4 bytes per activation, the backward chains
// read 1 instead of 2, and a layer's whole image is 32 registers per lane - all of it is prefetched BEFORE the layer's GEMM,
// ahead of the stash stores riding in that GEMM (vmcnt retires in issue order: a load issued after them waits for them).
// One piece = this lane's derivatives of one 32x32 accumulator tile (nt, mt): both 16-feature groups gp = 0, 1, i.e.
// accumulator registers 0..15, 16 bytes per lane, one coalesced 1 KB access per wave instruction.
// DK16 (round 5; fp16 mode of a model with ANALYTIC NORMALS, FieldGeom.dsz == 2): the unscaled derivative in fp16 (|c| <= 1:
// absolute error <= 2.4e-4), 32 bytes per lane and tile as two 1 KB wave instructions (one per 16-feature group).  The adjoint
// chain of the analytic normal multiplies by D_l in every layer; with the 8-bit image the normals of a trained field sit a
// median 0.4 degrees from the fp32 mode's (fp32 arithmetic + 8-bit D alone: 0.38 degrees) and the GGX / Hapke derivatives
// w.r.t. the normal turn that into a whole-gradient cosine of 0.08 (microfacet) / 0.87 (Hapke + theta) where fp32 arithmetic on
// fp16-rounded weights keeps 0.998 (profiles/r05_ablation.txt item 4).  The fp16 mode is BASELINE config 5's mode: it pays the
// byte.  bf16 keeps DK8: its referee (bf16-rounded weights) is as far from fp32 as the 8-bit image makes it.
// DK32 (fp32 parity mode): the scaled derivative in fp32, 64 bytes per lane and tile.
template <typename DK> struct DPiece { u32x4 w; };
template <> struct DPiece<DK16> { u32x4 w[2]; };
template <> struct DPiece<DK32> { f32x4 v[4]; };
template <typename DK> __host__ __device__ constexpr size_t dk_bytes() { return std::is_same<DK, DK32>::value ? 4 : (std::is_same<DK, DK16>::value ? 2 : 1); }
template <typename DK> __host__ __device__ constexpr size_t dtile_bytes(int BM, int F) { return (size_t)BM * F * dk_bytes<DK>(); }
// byte offset of the lane's piece (DK16: of its first half; the second one is 1 KB behind it)
template <typename DK, int MT, int NTW> __device__ __forceinline__ size_t dpiece_off(int wave, int nt, int mt, int lane) {
  return ((size_t)(wave * NTW + nt) * MT + mt) * (1024 * dk_bytes<DK>()) + (size_t)lane * (std::is_same<DK, DK32>::value ? 64 : 16);
}
// 4 unscaled derivatives -> 4 bytes: two v_cvt_pknorm_i16_f32 (saturating; NaN -> 0: a value out of range cannot carry into the
// neighbouring bytes), one v_perm_b32 that keeps the four high bytes, one v_xor (two's complement -> offset binary, which
// v_cvt_f32_ubyteN decodes) - 4 vector instructions where round 3's rne(127 c + 128) by fma / shift / add took 9: the packing
// was a fifth of the forward epilogue's vector issue (profiles/r04_ablation.txt item 16).
__device__ __forceinline__ unsigned int d8_pack4(float c0, float c1, float c2, float c3) {
  const unsigned int lo = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pknorm_i16(c0, c1)),
                     hi = __builtin_bit_cast(unsigned int, __builtin_amdgcn_cvt_pknorm_i16(c2, c3));
  return __builtin_amdgcn_perm(hi, lo, 0x07050301u) ^ 0x80808080u;
}
// d = k u + b.  Decoding constants of an activation (act: BN_ACT_SIN / BN_ACT_RELU) whose derivative is scaled by `scale`.
__device__ __forceinline__ void d8_consts(int act, float scale, float &k, float &b) {
  if (act == BN_ACT_SIN) { k = scale * (256.f / 32767.f); b = -127.5f * k; }
  else { k = scale * (1.f / 127.f); b = -128.f * k; }
}
__device__ __forceinline__ void d8_unpack4(unsigned int w, float k, float b, float (&d)[4]) {
  d[0] = fmaf((float)(w & 0xffu), k, b);
  d[1] = fmaf((float)((w >> 8) & 0xffu), k, b);
  d[2] = fmaf((float)((w >> 16) & 0xffu), k, b);
  d[3] = fmaf((float)(w >> 24), k, b);
}
// Producer side: one 16-feature group (gp) at a time.  c[e] = unscaled derivative of accumulator register 8 gp + e; `scale`
// (w0) is applied here in the fp32 mode only.  DK8: the two halves of a piece are stored together (one 16-byte store per lane).
template <typename DK> struct DHalf { unsigned int w[2]; };
template <> struct DHalf<DK16> { u32x4 w; };
template <> struct DHalf<DK32> { f32x4 v[2]; };
template <typename DK> __device__ __forceinline__ DHalf<DK> dhalf_make(const float (&c)[8], float scale) {
  DHalf<DK> r;
  if constexpr (std::is_same<DK, DK8>::value) {
    r.w[0] = d8_pack4(c[0], c[1], c[2], c[3]);
    r.w[1] = d8_pack4(c[4], c[5], c[6], c[7]);
  } else if constexpr (std::is_same<DK, DK16>::value) {
    const f32x8 f = {c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]};
    r.w = __builtin_bit_cast(u32x4, __builtin_convertvector(f, f16x8));
  } else {
#ifdef BN_DIAG_D8_IN_F32      // diagnostic (profiles/diag_c5_rows.py --d8lib): the fp32 mode with its derivatives sent through the 8-bit codec
    float q[8];
    d8_unpack4(d8_pack4(c[0], c[1], c[2], c[3]), 256.f / 32767.f, -127.5f * (256.f / 32767.f), *(float (*)[4])&q[0]);
    d8_unpack4(d8_pack4(c[4], c[5], c[6], c[7]), 256.f / 32767.f, -127.5f * (256.f / 32767.f), *(float (*)[4])&q[4]);
    r.v[0] = f32x4{q[0] * scale, q[1] * scale, q[2] * scale, q[3] * scale};
    r.v[1] = f32x4{q[4] * scale, q[5] * scale, q[6] * scale, q[7] * scale};
#else
    r.v[0] = f32x4{c[0] * scale, c[1] * scale, c[2] * scale, c[3] * scale};
    r.v[1] = f32x4{c[4] * scale, c[5] * scale, c[6] * scale, c[7] * scale};
#endif
  }
  return r;
}
template <typename DK> __device__ __forceinline__ void dpiece_store(char *p, const DHalf<DK> &h0, const DHalf<DK> &h1) {
  if constexpr (std::is_same<DK, DK8>::value) {
    stash_store((u32x4 *)p, u32x4{h0.w[0], h0.w[1], h1.w[0], h1.w[1]});
  } else if constexpr (std::is_same<DK, DK16>::value) {
    stash_store((u32x4 *)p, h0.w);
    stash_store((u32x4 *)(p + 1024), h1.w);
  } else {
    stash_store((f32x4 *)p, h0.v[0]); stash_store((f32x4 *)p + 1, h0.v[1]);
    stash_store((f32x4 *)p + 2, h1.v[0]); stash_store((f32x4 *)p + 3, h1.v[1]);
  }
}
// the same through buffer stores (the barrier-free forward trunk): `idx` = the piece's number (wave NT + nt) MT + mt, a scalar
template <typename DK, typename R> __device__ __forceinline__ void dpiece_store_buf(R rsrc, int voff, int idx, const DHalf<DK> &h0, const DHalf<DK> &h1) {
  static_assert(!std::is_same<DK, DK32>::value, "16-bit modes only");
  if constexpr (std::is_same<DK, DK8>::value) {
    stash_store_buf(rsrc, voff, idx * 1024, u32x4{h0.w[0], h0.w[1], h1.w[0], h1.w[1]});
  } else {
    stash_store_buf(rsrc, voff, idx * 2048, h0.w);
    stash_store_buf(rsrc, voff, idx * 2048 + 1024, h1.w);
  }
}
template <typename DK> __device__ __forceinline__ DPiece<DK> dpiece_load(const char *p) {
  DPiece<DK> r;
  if constexpr (std::is_same<DK, DK8>::value) {
    r.w = stash_load((const u32x4 *)p);
  } else if constexpr (std::is_same<DK, DK16>::value) {
    r.w[0] = stash_load((const u32x4 *)p);
    r.w[1] = stash_load((const u32x4 *)(p + 1024));
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) r.v[q] = stash_load((const f32x4 *)p + q);
  }
  return r;
}
// scaled derivatives of group gp (accumulator registers 8 gp .. 8 gp + 7); act = the activation the image was written for;
// `scale` = the layer's w0 for the unscaled kinds (DK8, DK16), ignored by DK32 (scaled when it was written)
template <typename DK> __device__ __forceinline__ void dpiece_get(const DPiece<DK> &pc, int gp, int act, float scale, float (&d)[8]) {
  if constexpr (std::is_same<DK, DK8>::value) {
    float k, b;
    d8_consts(act, scale, k, b);
    float lo[4], hi[4];
    d8_unpack4(pc.w[2 * gp], k, b, lo);
    d8_unpack4(pc.w[2 * gp + 1], k, b, hi);
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[e] = lo[e]; d[4 + e] = hi[e]; }
  } else if constexpr (std::is_same<DK, DK16>::value) {
    const f16x8 v = __builtin_bit_cast(f16x8, pc.w[gp]);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = (float)v[e] * scale;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) { d[e] = pc.v[2 * gp][e]; d[4 + e] = pc.v[2 * gp + 1][e]; }
  }
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
