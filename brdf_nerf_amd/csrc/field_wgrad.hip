// Weight-gradient kernels of the fused field-MLP backward for gfx950 (steps 2 and 3 of csrc/field_bwd.hip's header):
//   wgrad_kernel / wgrad256_kernel: dW_l[n][k] += sum_m dZ_l[m][n] X_l[m][k] as MFMA GEMMs over the stashed activations
//      (16-bit modes: 256 x 256 tiles, LDS stages + ds_read_b64_tr_b16 transposing reads; fp32: 128 x 128, plain ds_read_b32),
//      split over point chunks with fp32 atomics (deterministic mode: in turn order); bias gradients as column sums;
//   skinny_wgrad_kernel: the <= 4-row matrices (sigma head, learned normal, second head layers).
// Autograd counterpart in the reference: loss.backward() through SpSBRDFNeRF.forward (models/spsbrdfnerf.py:662-757).
// Compiled with -mllvm -amdgpu-sched-strategy=max-ilp (build.py FILE_FLAGS): -13 % on the fp16 256-tile kernel, -16 % on the
// skinny kernel with analytic normals, -1.7 % bf16; the same strategy costs the chain kernels 1-6 % (profiles/r02_ablation.txt).
#include "field_kernels.h"
#include "field_wgrad.h"

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

// 1 / (scale carried by the job's gradient operand): multiplies the fp32 sums before they are accumulated
__device__ __forceinline__ float wg_unscale(const float *amax, int sel) {
  if (amax == nullptr || sel == 0) return 1.f;
  return 1.f / (sel == 1 ? chain_scale(amax) : grad_scale_from(amax + 1, BN_GS_TARGET_ADJ));
}

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
      if (i + 1 < W2_BK / 16) frags(cur ^ 1, (i + 1) * 16);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int a = 0; a < W2_RA; ++a)
#pragma unroll
        for (int b = 0; b < NBV; ++b) mma32(acc[a][b], fa[cur][a], fb[cur][b]);
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

// ------------------------------------------------------------------------------------------------ host launchers
int bn_launch_wgrad(WgradArgs &wv, unsigned int *tk, bool bf, bool f16m, int64_t Mpad, hipStream_t st) {
  if (wv.n_jobs == 0) { wv.tile0[0] = 0; return 0; }
  wv.tickets = tk;
  wv.tile0[0] = 0;
  if (bf) {
    // 256 x 256 tiles, one 8-wave workgroup per CU: size the point splits for ~4 workgroups per CU in total
    for (int j = 0; j < wv.n_jobs; ++j)
      wv.tile0[j + 1] = wv.tile0[j] + ((wv.job[j].N + 255) / 256) * ((wv.job[j].K + 255) / 256);
    const int tiles = wv.tile0[wv.n_jobs];
#ifndef W2_BLOCKS
#define W2_BLOCKS 512   // tiles x point splits <= two rounds of the 256 CUs (one 144 KB workgroup per CU): 1024 -> 1.295 ms, 512 -> 1.253, 256 -> 1.290
#endif
    // (small batches - up to 1024 rays x 64 samples per launch - run faster with one round of workgroups: 512 rays 0.232 ->
    // 0.197 ms, 1024 rays 0.367 -> 0.345, session 51)
    // (both passes of a 4096-ray step in one call - 524,288 points: four rounds, i.e. the points per workgroup of the tuned
    // two-round shape; with twice the points per workgroup the launch ran 3-7 % slower than two launches, profiles/r03_ablation.txt)
    int64_t n_split = (Mpad <= 65536 ? W2_BLOCKS / 2 : (Mpad <= 327680 ? W2_BLOCKS : 2 * W2_BLOCKS)) / tiles;
    if (n_split < 1) n_split = 1;
    int64_t mpb2 = ceil_div64(ceil_div64(Mpad, n_split), W2_BK) * W2_BK;
    if (mpb2 < 512) mpb2 = 512;
    n_split = ceil_div64(Mpad, mpb2);
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
  int64_t mpb = ceil_div64(ceil_div64(Mpad, splits), WG_BK) * WG_BK;
  if (mpb < 256) mpb = 256;
  wv.m_per_block = (int)mpb;
  dim3 grid((unsigned)wv.tile0[wv.n_jobs], (unsigned)ceil_div64(Mpad, mpb));
  BnProfScope prof_(BN_K_WGRAD, st);
  wgrad_kernel<float><<<grid, 256, 0, st>>>(wv);
  BN_LAUNCH_CHECK("wgrad");
  return 0;
}

int bn_launch_skinny(SkinnyArgs &sv, unsigned int *tk, bool bf, bool f16m, int64_t Mpad, int64_t m_per_block, hipStream_t st) {
  if (sv.n_jobs == 0) return 0;
  sv.tickets = tk;
  sv.m_per_block = (int)m_per_block;
  dim3 grid((unsigned)ceil_div64(Mpad, m_per_block), (unsigned)sv.n_jobs);
  BnProfScope prof_(BN_K_SKINNY, st);
  if (f16m) skinny_wgrad_kernel<f16><<<grid, 256, 0, st>>>(sv);
  else if (bf) skinny_wgrad_kernel<bf16><<<grid, 256, 0, st>>>(sv);
  else skinny_wgrad_kernel<float><<<grid, 256, 0, st>>>(sv);
  BN_LAUNCH_CHECK("skinny_wgrad");
  return 0;
}
