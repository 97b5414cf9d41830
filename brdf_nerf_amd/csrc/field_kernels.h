// Device building blocks of the fused field MLP: the activation-stationary tile GEMM and the
// accumulator-register <-> (point, feature) maps shared by the forward and backward chain kernels.
//
// Tiling (one 512-thread workgroup = 8 waves = 2 waves per SIMD):
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

#define BN_THREADS 512
#define BN_WAVES 8

template <typename T> __device__ __forceinline__ typename Elem<T>::frag lds_frag(const T *p);
template <> __device__ __forceinline__ bf16x8 lds_frag<bf16>(const bf16 *p) { return *(const bf16x8 *)p; }
template <> __device__ __forceinline__ f32x8 lds_frag<float>(const float *p) {
  f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
  f32x8 r = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return r;
}
template <typename T> __device__ __forceinline__ typename Elem<T>::frag gld_frag(const T *p) { return lds_frag<T>(p); }

// acc[nt][mt] += W_packed(this wave's tiles) x B(lds).  `wp` points at the packed block of the wave's first
// n-tile for this K segment; consecutive n-tiles are KS*512 elements apart.
template <typename T, int MT, int NTW>
__device__ __forceinline__ void gemm_seg(f32x16 (&acc)[NTW][MT], const T *__restrict__ wp, int KS, const T *bsrc, int ldb,
                                         int lane) {
  typedef typename Elem<T>::frag frag;
  constexpr int U = Elem<T>::kU;
  const int r = lane & 31, h = lane >> 5;
  const T *wl = wp + (size_t)lane * 8;
  const T *bl = bsrc + (size_t)r * ldb + 8 * h;
  frag A0[U][NTW], A1[U][NTW], Bc[MT];
  // KS is a multiple of U (every K extent is a multiple of 32).  Software pipeline: two weight blocks (U k-steps
  // each) are always in flight from L2 - a block is re-filled right after its last MFMA has issued and consumed one
  // whole block later - and the LDS fragments of k-step s+1 are read while the MFMAs of k-step s run.
  // sched_barrier pins that order (hipcc otherwise sinks the prefetch loads next to their use).
  const int nb = KS / U;
  auto loadA = [&](frag(&A)[U][NTW], int blk) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) A[u][nt] = gld_frag<T>(wl + ((size_t)nt * KS + blk * U + u) * 512);
  };
  auto loadB = [&](frag(&B)[MT], int ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) B[mt] = lds_frag<T>(bl + (size_t)mt * 32 * ldb + ks * 16);
  };
  auto compute = [&](frag(&A)[U][NTW], int blk) {   // consumes Bc (fragments of k-step blk*U), leaves the next ones in Bc
#pragma unroll
    for (int u = 0; u < U; ++u) {
      frag Bn[MT];
      const int nxt = blk * U + u + 1;
      loadB(Bn, nxt < KS ? nxt : 0);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mma32(acc[nt][mt], A[u][nt], Bc[mt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) Bc[mt] = Bn[mt];
    }
  };
  loadA(A0, 0);
  loadA(A1, nb > 1 ? 1 : 0);
  loadB(Bc, 0);
  __builtin_amdgcn_sched_barrier(0);
  int b = 0;
  for (; b + 2 <= nb; b += 2) {
    compute(A0, b);
    loadA(A0, b + 2 < nb ? b + 2 : nb - 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(A1, b + 1);
    loadA(A1, b + 3 < nb ? b + 3 : nb - 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (b < nb) compute(A0, b);
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
__device__ __forceinline__ void st8(bf16 *p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (bf16)v[i];
  *(bf16x8 *)p = o;
}
__device__ __forceinline__ void st8(float *p, const float (&v)[8]) {
  *(f32x4 *)p = f32x4{v[0], v[1], v[2], v[3]};
  *(f32x4 *)(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void ld8(const bf16 *p, float (&v)[8]) {
  const bf16x8 o = *(const bf16x8 *)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)o[i];
}
__device__ __forceinline__ void ld8(const float *p, float (&v)[8]) {
  const f32x4 a = *(const f32x4 *)p, b = *(const f32x4 *)(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}

// Copy the workgroup's LDS tile [rows][width] (row stride ld) to a row-major global array [rows][gld] in 16-byte
// chunks: every wave instruction writes 1 KB of consecutive bytes (full 128-B lines) instead of the 32 x 16 B
// fragments the accumulator layout would give.
template <typename T> __device__ __forceinline__ void tile_to_global(const T *lds, int ld, T *g, int gld, int rows, int width) {
  constexpr int EPC = 16 / sizeof(T);
  const int cpr = width / EPC;
  for (int c = threadIdx.x; c < rows * cpr; c += BN_THREADS) {
    const int row = c / cpr, cc = (c % cpr) * EPC;
    *(uint4 *)(g + (size_t)row * gld + cc) = *(const uint4 *)(lds + (size_t)row * ld + cc);
  }
}
