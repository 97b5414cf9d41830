// PROBE (not part of libbrdfnerf_hip.so; built and run by profiles/probe_gemm_rate.py).
// The chain kernels' tile GEMM (field_kernels.h gemm_range: weights streamed from L2 as the MFMA A operand, activations read
// from the LDS tile as the B operand, 64 output features x 128 points per wave) alone in a kernel: shader cycles per MFMA
// of one wave with 1 or 2 waves per SIMD, by weight-prefetch depth, with and without the operand traffic.
#include "field_kernels.h"

// ---- The chain GEMM fed by a CONTINUOUS weight stream (measured here, not adopted).  The ring of DEPTH weight-fragment slots `A` belongs to the
// caller and is never drained: while segment s runs its last DEPTH k-steps, the slots are re-filled with the FIRST k-steps of
// the segment that follows (another k-range of the same matrix, the next layer's matrix, ...), which depend on nothing the
// kernel computes.  The idea: (a) no pipeline fill per segment - a chain kernel
// starts 2-3 GEMM segments per layer, each of which exposed an L2 round trip; (b) a wave's loads retire in issue order behind
// its own stores (vmcnt): the first weight loads of a GEMM used to be issued right after the epilogue's burst of stash
// stores and waited for every one of them (the GEMM alone 35 / 56 cycles per MFMA at one / two waves per SIMD, 45 / 78 with a
// layer's 24 stores per wave in front of it); here they are issued BEFORE the epilogue.  Result (profiles/r04_probe_gemm_rate.txt):
// no gain - with the stores the probe is bound by HBM write bandwidth (4.9 TB/s), not by the order of the queue.  Segments are whole multiples of DEPTH k-steps (slot indices stay compile-time).
template <typename T, int NTW, int DEPTH>
__device__ __forceinline__ void wstream_start(typename Elem<T>::frag (&A)[DEPTH][NTW], const T *__restrict__ wp, int KS, int k0, int lane) {
  const T *wl = wp + (size_t)lane * 8;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) A[d][nt] = gld_frag<T>(wl + ((size_t)nt * KS + k0 + d) * 512);
}
// k-steps [k0, k0 + nks) of the packed matrix wp (n-tiles KS k-steps apart); the stream continues at k-step k0n of matrix wpn
// (n-tiles KSn apart).  The last segment of a stream passes its own matrix and k0n = k0 + nks - DEPTH (a harmless re-read).
template <typename T, int MT, int NTW, int DEPTH>
__device__ __forceinline__ void gemm_stream(f32x16 (&acc)[NTW][MT], typename Elem<T>::frag (&A)[DEPTH][NTW], const T *__restrict__ wp, int KS,
                                            int k0, int nks, const T *__restrict__ wpn, int KSn, int k0n, const T *bsrc, int ldb, int lane) {
  typedef typename Elem<T>::frag frag;
  const int r = lane & 31, h = lane >> 5;
  const T *wl = wp + (size_t)lane * 8, *wln = wpn + (size_t)lane * 8;
  const T *bl = bsrc + (size_t)r * ldb + 8 * h;
  const int kend = k0 + nks;
  frag Bc[MT];
  auto loadB = [&](frag(&B)[MT], int ks) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) B[mt] = lds_frag<T>(bl + (size_t)mt * 32 * ldb + ks * 16);
  };
  loadB(Bc, k0);
  __builtin_amdgcn_sched_barrier(0);
  for (int ks = k0; ks < kend; ks += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      frag Bn[MT];
      loadB(Bn, ks + d + 1 < kend ? ks + d + 1 : k0);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mma32(acc[nt][mt], A[d][nt], Bc[mt]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) Bc[mt] = Bn[mt];
      // re-fill the slot: DEPTH k-steps ahead in this segment, or the head of the next one
      const int kk = ks + d + DEPTH;
      const bool cur = kk < kend;
      const T *pw = cur ? wl : wln;
      const int K2 = cur ? KS : KSn, ki = cur ? kk : k0n + (kk - kend);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) A[d][nt] = gld_frag<T>(pw + ((size_t)nt * K2 + ki) * 512);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}


template <int WPS, int DEPTH>
__global__ __launch_bounds__(256 * WPS, WPS) void gemm_rate_kernel(const bf16 *packed, int layers, int reps, unsigned long long *cyc, float *sink,
                                                                   char *stream, size_t stream_bytes, int stores) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int F = 512, LDA = F + 8, KS = F / 16;
  bf16 *ACT = (bf16 *)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 128 * LDA; i += 256 * WPS) ACT[i] = (bf16)(((i * 2654435761u) >> 20 & 255) * (1.f / 256.f) - 0.5f);
  __syncthreads();
  f32x16 acc[2][4];
  zero_acc<4, 2>(acc);
  NoSide none;
#ifdef BN_PROBE_STREAM
  typename Elem<bf16>::frag AS[DEPTH][2];
#endif
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int rep = 0; rep < reps; ++rep)
    for (int l = 0; l < layers; ++l) {
      const bf16 *wp = packed + (size_t)l * F * F + (size_t)(wave * 2) * KS * 512;
#ifdef BN_PROBE_STREAM     // continuous weight stream: two half-GEMMs per layer (as the anti-phase trunk runs them), the ring carried across layers
      const int ln = l + 1 < layers ? l + 1 : 0;
      const bf16 *wpn = packed + (size_t)ln * F * F + (size_t)(wave * 2) * KS * 512;
      if (rep == 0 && l == 0) wstream_start<bf16, 2, DEPTH>(AS, wp, KS, 0, lane);
      gemm_stream<bf16, 4, 2, DEPTH>(acc, AS, wp, KS, 0, KS / 2, wp, KS, KS / 2, ACT, LDA, lane);
      gemm_stream<bf16, 4, 2, DEPTH>(acc, AS, wp, KS, KS / 2, KS / 2, wpn, KS, 0, ACT, LDA, lane);
#elif defined(BN_PROBE_HALVES)   // two gemm_range calls per layer (the product trunk of rounds 1-3)
      gemm_range<bf16, 4, 2, DEPTH>(acc, wp, KS, 0, KS / 2, ACT, LDA, lane, none);
      gemm_range<bf16, 4, 2, DEPTH>(acc, wp, KS, KS / 2, KS / 2, ACT, LDA, lane, none);
#else
      gemm_range<bf16, 4, 2, DEPTH>(acc, wp, KS, 0, KS, ACT, LDA, lane, none);
#endif
      if (stores > 0) {   // a layer's stash traffic of the training forward: `stores` x 1 KB per wave, streaming (never re-read)
        const size_t per = (size_t)stores * 1024 * 4 * WPS;
        size_t off = (((size_t)(rep * layers + l) * gridDim.x + blockIdx.x) * per) % (stream_bytes - per);
        off = (off & ~(size_t)1023) + (size_t)wave * stores * 1024 + lane * 16;
        for (int i = 0; i < stores; ++i)
          stash_store((u32x4 *)(stream + off + (size_t)i * 1024), u32x4{(unsigned)i, (unsigned)l, (unsigned)rep, (unsigned)lane});
      }
    }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[nt][mt][i];
  if (s == 123.456f) sink[0] = s;
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int WPS, int DEPTH> static int launch(const void *packed, int layers, int reps, unsigned long long *cyc, float *sink, int blocks, hipStream_t st,
                                                char *stream, size_t stream_bytes, int stores) {
  const size_t lds = 128 * 520 * 2;
  if (hipFuncSetAttribute((const void *)gemm_rate_kernel<WPS, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
  gemm_rate_kernel<WPS, DEPTH><<<blocks, 256 * WPS, lds, st>>>((const bf16 *)packed, layers, reps, cyc, sink, stream, stream_bytes, stores);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int bn_probe_gemm_rate(const void *packed, int layers, int reps, int wps, int depth, unsigned long long *cyc, float *sink, int blocks, hipStream_t st,
                                  char *stream, size_t stream_bytes, int stores) {
#define CASE(W, D) if (wps == W && depth == D) return launch<W, D>(packed, layers, reps, cyc, sink, blocks, st, stream, stream_bytes, stores);
  CASE(1, 2) CASE(1, 4) CASE(1, 6) CASE(1, 8) CASE(2, 2) CASE(2, 4) CASE(2, 6) CASE(2, 8)
  return 3;
}
