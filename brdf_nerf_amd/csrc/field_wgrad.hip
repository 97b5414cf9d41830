// Weight-gradient kernels of the fused field-MLP backward for gfx950 (steps 2 and 3 of csrc/field_bwd.hip's header):
//   wgrad_kernel / wgrad256_kernel: dW_l[n][k] += sum_m dZ_l[m][n] X_l[m][k] as MFMA GEMMs over the stashed activations
//      (16-bit modes: 256 x 256 tiles, LDS stages + ds_read_b64_tr_b16 transposing reads; fp32: 128 x 128, plain ds_read_b32),
//      split over point chunks with fp32 atomics (deterministic mode: in turn order); bias gradients as column sums;
//   skinny_wgrad_kernel: the <= 4-row matrices (sigma head, learned normal, second head layers).
// Autograd counterpart in the reference: loss.backward() through SpSBRDFNeRF.forward (models/spsbrdfnerf.py:662-757).
// Compiled with -mllvm -amdgpu-sched-strategy=max-ilp (build.py FILE_FLAGS): -13 % on the fp16 256-tile kernel, -16 % on the
// skinny kernel with analytic normals, -1.7 % bf16; the same strategy costs the chain kernels 1-6 % (profiles/history/r02_ablation.txt).
#include "field_kernels.h"
#include "field_wgrad.h"

// ---- accumulation across the point splits: slabs + a fixed-order sum (round 4) ------------------------------------------------
// The weight-gradient kernels split the points over many workgroups.  Rounds 1-3 added the workgroups' partial tiles into the
// gradient with fp32 atomics (261 MB of atomic traffic per step at the 1.3 TB/s the chip gives them, and a run-to-run order: a
// bitwise reproducible gradient needed a turn-taking mode that cost 13-16 % of a step).  Now every workgroup writes its partial
// tile with plain 16-byte stores into ITS OWN slab of a workspace in the stash ([output tile][point split], accumulator-register
// order: 1 KB per wave instruction), and wgrad_reduce_kernel / skinny_reduce_kernel add the slabs of an output element in split
// order - and the jobs that feed the same matrix (primal + analytic-normal term of a trunk layer) in job order - into the gradient
// with one read-modify-write per element.  The gradient is bitwise reproducible by construction (the reference trains with
// Trainer(deterministic=True), main.py:726), there is no inter-workgroup protocol left, and the stores run at HBM speed.

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
  // this workgroup's slab: accumulators in register order (C[n][k]: accumulator row index = n, column (lane & 31) = k), then the
  // 128 bias column sums
  float *P = A.part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * WG_SLAB128;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4 *)(P + (size_t)(((((wave * 2 + a) * 2 + b) * 4 + q) * 64 + lane) * 4)) =
            f32x4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
  if (do_bias) P[128 * 128 + tid] = bsum;
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

// Sum of a fragment's 8 elements added to an fp32 value: four v_dot2c_f32_{bf16,f16} against (1, 1) in the 16-bit modes (the bias
// gradient's column sums in wgrad256).
__device__ __forceinline__ float frag_sum8(float s, const bf16x8 &f) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  const bf16x2 one = {(__bf16)1.f, (__bf16)1.f};
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 0, 1), one, s, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 2, 3), one, s, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 4, 5), one, s, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 6, 7), one, s, false);
  return s;
}
__device__ __forceinline__ float frag_sum8(float s, const f16x8 &f) {
  typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
  const f16x2 one = {(_Float16)1.f, (_Float16)1.f};
  s = __builtin_amdgcn_fdot2(__builtin_shufflevector(f, f, 0, 1), one, s, false);
  s = __builtin_amdgcn_fdot2(__builtin_shufflevector(f, f, 2, 3), one, s, false);
  s = __builtin_amdgcn_fdot2(__builtin_shufflevector(f, f, 4, 5), one, s, false);
  s = __builtin_amdgcn_fdot2(__builtin_shufflevector(f, f, 6, 7), one, s, false);
  return s;
}
__device__ __forceinline__ float frag_sum8(float s, const f32x8 &f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) s += f[j];
  return s;
}

#if defined(BN_PHASE_TIMING) && defined(BN_PHASE_TIMING_WGRAD)
BN_PH_DEFINE_READER(bn_debug_phase_read_wgrad)
#define WG_PH_DECL BN_PH_DECL
#define WG_PH(i) BN_PH(i)
#define WG_PH_FLUSH BN_PH_FLUSH
#else
#define WG_PH_DECL
#define WG_PH(i)
#define WG_PH_FLUSH
#endif
// W2_WAVES = 8: wave tile 64(n) x 128(k), 2 waves per SIMD.  W2_WAVES = 4 (wave tile 128 x 128, accumulators in the
// AGPR half of the register file, a third fewer LDS fragment bytes per MFMA) spilled in the k-loop and measured 5.7x
// slower in round 1 (profiles/history/r01_ablation.txt); rebuilt in round 4 with the native staging generalised to two
// 32-column blocks per wave it allocated cleanly (206 VGPRs + 256 AGPRs) and still lost, 2.727 ms against 2.397: with one
// wave per SIMD nothing covers the stage barrier and the first fragment reads behind it (profiles/r04_ablation.txt item 13).
// The staging below is the 8-wave form (one 32-column block per wave): the switch does not build any more - the
// static_assert says so - and stays only as the name of that experiment.
#ifndef W2_WAVES
#define W2_WAVES 8
#endif
#define W2_RA (256 / ((W2_WAVES / 2) * 32))   // 32-row accumulator tiles per wave along n
// One 256 x 256 output tile over the points [mb, me).  NBV = 32-column accumulator tiles this WAVE multiplies (4 for a
// full tile; the 60-column positional-encoding operand only has columns for two tiles of the wc = 0 waves - the other
// waves of such a block just take part in staging and barriers).
template <typename T, int NBV, bool BNAT, bool ANAT>
__device__ __forceinline__ void w2_body(const WgradJob &J, int n0, int k0, int64_t mb, int64_t me, T *sA, T *sB, float *P) {
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
  // with one ds_write_b128 like a row-major piece (two ds_write_b64 of the lane's own runs instead: 2-way bank conflicts, the
  // kernel +14 %, profiles/r04_ablation.txt item 9).
  const int nr = lane & 31, nh = lane >> 5, nswz = w2_swz(nr);
  static_assert(W2_BK / (W2_WAVES * 2) == 4, "native staging: 4 wave-instructions per wave and stage");
  // chunk q = 4 w + c: 32-point block c & 1, column half (c >> 1) & 1, 32-column block w (one per wave): everything but the
  // wave / lane part of the addresses is a compile-time constant of c
  const int mtn = (BNAT || ANAT) ? J.b_bm / 32 : 1, ncb = (BNAT || ANAT) ? J.b_F / 32 : 1;
  int cbg = (k0 >> 5) + wave;
  cbg = cbg < ncb ? cbg : ncb - 1;                           // beyond the operand: any valid block (those output columns are never stored)
  const int boff0 = (cbg * mtn * 2 * 64 + lane) * 8;         // + ((c & 1) * 2 + ((c >> 1) & 1)) * 512 elements
  // Native-order A (ANAT; round 4: the dZ_l stashes of the 16-bit modes, written straight from the backward chain's epilogue
  // registers like Y_l from the forward's): the same image (tiles of b_bm points x b_F columns), the same staging, with the
  // 32-column block taken from the tile's first output row n0.
  int cag = (n0 >> 5) + wave;
  cag = cag < ncb ? cag : ncb - 1;                           // (rows beyond N are never stored by the reduce)
  const int aoff0 = (cag * mtn * 2 * 64 + lane) * 8;
  const int lslot0 = 8 * wave + 2 * nh;                      // + 4 ((c >> 1) & 1); LDS row = 32 (c & 1) + nr
  const int64_t tile_elems = (int64_t)J.b_bm * J.b_F;
  // Stage pipeline with ONE register set: while stage s is multiplied, the registers (stage s+1, loaded during stage
  // s-1) are written to the other LDS buffer a chunk pair per 16-point step and re-filled at once with stage s+2 -
  // every global load has a whole stage of MFMAs to arrive, every LDS buffer one barrier between its last read and its
  // next write.
  u32x4 ra[NC], rb[NC];
  // columns beyond a row-major operand's extent read one 16-byte block of zeros with row stride 0: the stage loop has no
  // branch (an exec-masked load per chunk split its basic block and cost 6 % of the kernel: profiles/history/r01_ablation.txt)
  const T *pa = a_ok ? gA + cc : (const T *)w2_zeros, *pb = b_ok ? gB + cc : (const T *)w2_zeros;
  const int64_t sa = a_ok ? J.lda : 0, sb = b_ok ? J.ldb : 0;
  auto gload1 = [&](int64_t m, int c) {
    m = m < me ? m : me - W2_BK;   // the two prefetches past the end re-read the last stage: no branch in the stage loop
    const int64_t row = m + row0 + RPP * c;
    // tile = m >> log2(bm); first 32-point block of the stage inside its tile = (m mod bm) / 32; a tile image is bm x F elements
    const int64_t tile_off = (m >> J.b_bm_shift) * tile_elems;
    const int mt0 = ((int)m & (J.b_bm - 1)) >> 5;
    const T *qa = ANAT ? (const T *)J.A + tile_off + mt0 * 1024 + aoff0 + ((c & 1) * 2 + ((c >> 1) & 1)) * 512 : pa + row * sa;
    const T *qb = BNAT ? (const T *)J.B + tile_off + mt0 * 1024 + boff0 + ((c & 1) * 2 + ((c >> 1) & 1)) * 512 : pb + row * sb;
    ra[c] = *(const u32x4 *)qa;
    rb[c] = *(const u32x4 *)qb;
  };
  auto gload = [&](int64_t m) {
#pragma unroll
    for (int c = 0; c < NC; ++c) gload1(m, c);
  };
  auto sstore1 = [&](int buf, int c) {
    if (ANAT) {
      T *rowp = sA + buf * W2_STAGE + (32 * (c & 1) + nr) * W2_LD;
      const int sl0 = lslot0 + 4 * ((c >> 1) & 1);
      const auto s02 = __builtin_amdgcn_permlane32_swap(ra[c][0], ra[c][2], false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(ra[c][1], ra[c][3], false, false);
      *(u32x4 *)(rowp + ((sl0 ^ nswz) << 2)) = u32x4{s02[0], s13[0], s02[1], s13[1]};
    } else {
      *(u32x4 *)(sA + buf * W2_STAGE + (row0 + RPP * c) * W2_LD + scol) = ra[c];
    }
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
  // bias gradient = column sums of A, from the A fragments the waves hold anyway (lane (r, h): row n = r, points 8h .. 8h+7 of
  // the step), stored by the workgroups of the first k-block.  The two waves of a row group (wc = 0, 1) hold the same A
  // fragments: each adds up HALF of the row group's 32-row tiles (BA of them, picked by wc with four selects per tile), by
  // v_dot2c against (1, 1): 8 vector instructions per 16-point step and wave, no branch in the k-loop.  Rounds 1-3 and the first
  // half of round 4: every wave added up ALL its A fragments by shifts / masks / adds, ~36 vector instructions per 8 MFMAs -
  // 10 % of the kernel (a probe without the sums: -15 %; profiles/r04_ablation.txt item 17, where the other forms tried are:
  // stage-loop instances with / without the sums spill inside the loop; the tiles numbered from the wave's own half - no
  // selects - measured 3 % slower than the selects).
  // Round 5 (profiles/r05_ablation.txt item 6): the sums as a property of the KERNEL instantiation - the grid launched twice, once
  // for the workgroups that store them and once, without any sums, for the others - measured 3.48 ms against 2.25: the tiles of a
  // row / column of a job share their operands through L2 only while they run TOGETHER.  One launch, the sums in every workgroup.
  constexpr int BA = W2_RA / 2;
  const bool do_bias = J.bias != nullptr && k0 == 0;
  float bsum[BA];
#pragma unroll
  for (int a = 0; a < BA; ++a) bsum[a] = 0.f;
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
      {   // every wave adds its half of the row group's A fragments up, the workgroups of the first k-block store the sums: no branch in the k-loop
#pragma unroll
        for (int a = 0; a < BA; ++a) {
          const u32x4 f0 = __builtin_bit_cast(u32x4, fa[cur][a]), f1 = __builtin_bit_cast(u32x4, fa[cur][BA + a]);
          bsum[a] = frag_sum8(bsum[a], __builtin_bit_cast(frag_t, wc ? f1 : f0));
        }
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
  // The prefetch of stage 1 in the order the stage loop re-fills the registers (chunk by chunk, A then B), pinned: hipcc hoisted
  // and REVERSED these loads, the chunk the loop stores first entered it as the youngest load in flight, and its wait-count pass
  // kept the merged "s_waitcnt vmcnt(0)" in front of every stage's first LDS store instead of vmcnt(6).  (Timing: equal - the
  // loads have landed by then either way; profiles/r04_ablation.txt item 9.)
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    __builtin_amdgcn_sched_barrier(0);
    gload1(mb + W2_BK, c);
  }
  __builtin_amdgcn_sched_barrier(0);
  WG_PH(14)
  int buf = 0;
  for (int64_t m = mb; m < me; m += W2_BK) {
    __syncthreads();
    WG_PH(3)
    compute(buf, m + 2 * W2_BK);
    WG_PH(0)
    buf ^= 1;
  }
  // the workgroup's slab P: accumulators in register order (tile (a, b) of wave w, registers 4 q .. 4 q + 3 of lane l at float
  // ((((w W2_RA + a) 4 + b) 4 + q) 64 + l) 4: one 16-byte store per lane, 1 KB per wave instruction), then the 256 bias column sums
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int a = 0; a < W2_RA; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *(f32x4 *)(P + (size_t)(((((wave * W2_RA + a) * 4 + b) * 4 + q) * 64 + lane) * 4)) =
            f32x4{acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
  if (do_bias) {
#pragma unroll
    for (int a = 0; a < BA; ++a) {
      const float v = bsum[a] + __shfl_xor(bsum[a], 32);
      if (h == 0) P[256 * 256 + wr * (W2_RA * 32) + (wc * BA + a) * 32 + r] = v;
    }
  }
  WG_PH(4)
  WG_PH_FLUSH
}

template <typename T>
__global__ __launch_bounds__(W2_WAVES * 64, W2_WAVES == 8 ? 2 : 1) void wgrad256_kernel(const WgradArgs A, int n_split, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) char smem_w[];
  T *sA = (T *)smem_w;                 // [2][W2_BK][W2_LD]
  T *sB = sA + 2 * W2_STAGE;
  // XCD-aware id: hardware deals consecutive block ids round-robin over the 8 XCDs; give each XCD a contiguous range
  const int per = n_blocks / 8;              // n_blocks is a multiple of 8
  const int lid = (int)((blockIdx.x % 8) * per + blockIdx.x / 8);
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
  float *P = A.part + ((size_t)tt * n_split + split) * WG_SLAB256;
#ifdef BN_CLOCK_STAMP_WGRAD
  BN_CLK_BEGIN
#endif
  if (J.b_native & WG_A_NATIVE) {     // gradient operand in native order (the trunk's dZ_l)
    if (J.b_native & WG_B_NATIVE) {   // layer-output operand in native order: full-width column blocks only (F is a multiple of 64)
      if (cols >= 65) w2_body<T, 4, true, true>(J, n0, k0, mb, me, sA, sB, P);
      else if (cols >= 33) w2_body<T, 2, true, true>(J, n0, k0, mb, me, sA, sB, P);
      else w2_body<T, 0, true, true>(J, n0, k0, mb, me, sA, sB, P);
    } else if (cols >= 65) w2_body<T, 4, false, true>(J, n0, k0, mb, me, sA, sB, P);
    else if (cols >= 33) w2_body<T, 2, false, true>(J, n0, k0, mb, me, sA, sB, P);
    else if (cols >= 1) w2_body<T, 1, false, true>(J, n0, k0, mb, me, sA, sB, P);
    else w2_body<T, 0, false, true>(J, n0, k0, mb, me, sA, sB, P);
  } else if (J.b_native & WG_B_NATIVE) {
    if (cols >= 65) w2_body<T, 4, true, false>(J, n0, k0, mb, me, sA, sB, P);
    else if (cols >= 33) w2_body<T, 2, true, false>(J, n0, k0, mb, me, sA, sB, P);
    else w2_body<T, 0, true, false>(J, n0, k0, mb, me, sA, sB, P);
  } else if (cols >= 65) w2_body<T, 4, false, false>(J, n0, k0, mb, me, sA, sB, P);
  else if (cols >= 33) w2_body<T, 2, false, false>(J, n0, k0, mb, me, sA, sB, P);
  else if (cols >= 1) w2_body<T, 1, false, false>(J, n0, k0, mb, me, sA, sB, P);
  else w2_body<T, 0, false, false>(J, n0, k0, mb, me, sA, sB, P);
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
              float v = s[gp][c][e];     // a fixed butterfly over the 32 points (no LDS atomics: the order of additions is part of the result)
#pragma unroll
              for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o);
              if (r == 0) *dst = v;      // (column blocks of different waves are disjoint)
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
          float v = bs0[c];
#pragma unroll
          for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o);
          if (r == 0) red[4 * 512 + c] = v;
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
  // row groups meet in LDS one after the other (fixed order)
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
  }
  // the block's sums go to ITS slab [job][split][4 x 512 + 4] (plain stores; skinny_reduce_kernel adds the splits up in order)
  __syncthreads();
  float *SP = A.part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * SK_SLAB;
  for (int i = tid; i < J.nc * J.K; i += 256) {
    const int c = i / J.K, k = i % J.K;
    SP[c * 512 + k] = red[c * 512 + k];
  }
  if (tid < 4) SP[4 * 512 + tid] = tid < J.nc ? red[4 * 512 + tid] : 0.f;
}

// ---- fixed-order sums of the slabs into the gradient ------------------------------------------------------------------------
// One thread per float4 of a HEAD job's slab image (256 per workgroup): element = sum over the job's splits in split order, then
// over the jobs chained to it (same matrix, same tiling), each times its own loss-scale factor; one += per gradient element.
// The loads of 8 splits are issued together (independent addresses), the additions stay in split order.
template <int TILE> __global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs A) {
  constexpr int NB = TILE / 64;                          // 32-column accumulator tiles per wave (4: 256-tile, 2: 128-tile)
  constexpr int SLAB = TILE * TILE + TILE;
  constexpr int SUBS = TILE * TILE / 4 / 256;
  const int tt = blockIdx.x / SUBS, sub = blockIdx.x % SUBS, tid = threadIdx.x;
  int jb = 0;
  while (jb + 1 < A.n_jobs && tt >= A.tile0[jb + 1]) ++jb;
  if (!A.head[jb]) return;
  const WgradJob &J = A.job[jb];
  const int t = tt - A.tile0[jb];
  const int tiles_k = (J.K + TILE - 1) / TILE;
  const int n0 = (t / tiles_k) * TILE, k0 = (t % tiles_k) * TILE;
  const int ns = A.n_split;
  {
    const int idx4 = sub * 256 + tid;
    const int lane = idx4 & 63, q = (idx4 >> 6) & 3, b = (idx4 >> 8) % NB, a = ((idx4 >> 8) / NB) & 1, wave = (idx4 >> 8) / (NB * 2);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int jj = jb; jj >= 0; jj = A.chain_next[jj]) {
      const float osc = wg_unscale(A.amax, A.job[jj].scale_sel);
      const float *P = A.part + (size_t)(A.tile0[jj] + t) * ns * SLAB + (size_t)idx4 * 4;
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      int sp = 0;
      for (; sp + 8 <= ns; sp += 8) {
        f32x4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = *(const f32x4 *)(P + (size_t)(sp + u) * SLAB);
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += x[u];
      }
      for (; sp < ns; ++sp) sum += *(const f32x4 *)(P + (size_t)sp * SLAB);
      v += sum * osc;
    }
    const int r = lane & 31, h = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int k = k0 + wc * (TILE / 2) + b * 32 + r, nb = n0 + wr * 64 + a * 32 + 8 * q + 4 * h;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (nb + e < J.N && k < J.K) J.C[(size_t)(nb + e) * J.ldc + k] += v[e];
  }
  if (sub == 0 && k0 == 0 && tid < TILE && n0 + tid < J.N) {      // bias gradients: column sums of the k = 0 tiles
    float v = 0.f;
    float *dst = nullptr;
    for (int jj = jb; jj >= 0; jj = A.chain_next[jj]) {
      if (!A.job[jj].bias) continue;
      dst = A.job[jj].bias;
      const float osc = wg_unscale(A.amax, A.job[jj].scale_sel);
      const float *P = A.part + (size_t)(A.tile0[jj] + t) * ns * SLAB + TILE * TILE + tid;
      float sum = 0.f;
      for (int sp = 0; sp < ns; ++sp) sum += P[(size_t)sp * SLAB];
      v += sum * osc;
    }
    if (dst) dst[n0 + tid] += v;
  }
}

// Skinny slabs: one workgroup per (job, row c, 32 columns): thread (kk, g) adds splits g, g + 8, g + 16, ... of column kk in
// order (8 independent loads at a time), the 8 partial sums meet in LDS and are added in order g = 0 .. 7; then the later jobs
// that write the same row (the primal and the analytic-normal term of the sigma head), in job order.  Fixed order throughout.
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const SkinnyArgs A, int n_split) {
  __shared__ float part[8][32], bpart[256];
  const int jb = blockIdx.x, c = blockIdx.y, k0 = blockIdx.z * 32, tid = threadIdx.x, kk = tid & 31, g = tid >> 5;
  const SkinnyJob &J = A.job[jb];
  if (c >= J.nc || !J.out[c] || k0 >= J.K) return;
  for (int j2 = 0; j2 < jb; ++j2)          // an earlier job owns this row
    for (int c2 = 0; c2 < A.job[j2].nc; ++c2)
      if (A.job[j2].out[c2] == J.out[c]) return;
  float v = 0.f, vb = 0.f;
  for (int j2 = jb; j2 < A.n_jobs; ++j2)
    for (int c2 = 0; c2 < A.job[j2].nc; ++c2) {
      if (A.job[j2].out[c2] != J.out[c]) continue;
      const float osc = wg_unscale(A.amax, A.job[j2].scale_sel);
      const float *P = A.part + (size_t)j2 * n_split * SK_SLAB;
      // columns k0 + kk of row c2; lanes kk = 0 .. 3 of group 0 also carry the bias sum of row c2 (slab word 4 * 512 + c2)
      const bool col = k0 + kk < A.job[j2].K;
      float sum = 0.f;
      int sp = g;
      for (; sp + 56 < n_split; sp += 64) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = col ? P[(size_t)(sp + 8 * u) * SK_SLAB + c2 * 512 + k0 + kk] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) sum += x[u];
      }
      for (; sp < n_split; sp += 8) sum += col ? P[(size_t)sp * SK_SLAB + c2 * 512 + k0 + kk] : 0.f;
      __syncthreads();
      part[g][kk] = sum;
      __syncthreads();
      if (g == 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += part[q][kk];
        v += t * osc;
      }
      if (k0 == 0 && A.job[j2].bias[c2] && J.bias[c]) {     // bias sums (of the unscaled fp32 dpre): one load per thread, added in split order
        __syncthreads();
        bpart[tid] = tid < n_split ? P[(size_t)tid * SK_SLAB + 4 * 512 + c2] : 0.f;
        __syncthreads();
        if (tid == 0) {
          float t = 0.f;
          for (int q = 0; q < n_split; ++q) t += bpart[q];
          vb += t;
        }
      }
    }
  if (g == 0 && k0 + kk < J.K) J.out[c][k0 + kk] += v;
  if (k0 == 0 && tid == 0 && J.bias[c]) *J.bias[c] += vb;
}

// ------------------------------------------------------------------------------------------------ host launchers
// Chains of jobs that add into the same matrix (same C, extents and row stride: the primal and the analytic-normal term of a
// trunk layer): the reduce kernel walks a chain from its head, in job order.
static void wg_chain(WgradArgs &wv) {
  for (int j = 0; j < wv.n_jobs; ++j) { wv.chain_next[j] = -1; wv.head[j] = 1; }
  for (int j = 0; j < wv.n_jobs; ++j) {
    if (!wv.head[j]) continue;
    int last = j;
    for (int q = j + 1; q < wv.n_jobs; ++q) {
      const WgradJob &a = wv.job[j], &b = wv.job[q];
      if (b.C == a.C && b.N == a.N && b.K == a.K && b.ldc == a.ldc) { wv.chain_next[last] = q; wv.head[q] = 0; last = q; }
    }
  }
}

int bn_launch_wgrad(WgradArgs &wv, bool bf, bool f16m, int64_t Mpad, float *part, size_t part_bytes, hipStream_t st) {
  if (wv.n_jobs == 0) { wv.tile0[0] = 0; return 0; }
  BN_REQUIRE(part, "wgrad: no slab workspace");
  wv.part = part;
  wv.tile0[0] = 0;
  wg_chain(wv);
  if (bf) {
    // 256 x 256 tiles, one 8-wave workgroup per CU: size the point splits for ~4 workgroups per CU in total
    for (int j = 0; j < wv.n_jobs; ++j)
      wv.tile0[j + 1] = wv.tile0[j] + ((wv.job[j].N + 255) / 256) * ((wv.job[j].K + 255) / 256);
    const int tiles = wv.tile0[wv.n_jobs];
    static_assert(W2_BK == 64, "bn_wgrad_splits (field.h) rounds the points per workgroup to W2_BK");
    int64_t mpb2 = 0;
    const int64_t n_split = bn_wgrad_splits(true, tiles, Mpad, &mpb2);      // (shared with the stash sizing: field.h)
    wv.m_per_block = (int)mpb2;
    wv.n_split = (int)n_split;
    BN_REQUIRE((size_t)tiles * n_split * WG_SLAB256 * sizeof(float) <= part_bytes, "wgrad: %d tiles x %d splits do not fit the slab workspace (%zu bytes)",
               tiles, (int)n_split, part_bytes);
    const int n_blocks = (int)ceil_div64((int64_t)tiles * n_split, 8) * 8;
    const size_t lds = (size_t)4 * W2_STAGE * 2;
    const void *kfn = f16m ? (const void *)wgrad256_kernel<f16> : (const void *)wgrad256_kernel<bf16>;
    if (int e = bn_configure_lds(kfn, lds, "wgrad256")) return e;
    {
      BnProfScope prof_(BN_K_WGRAD, st);
      const dim3 grd((unsigned)n_blocks), blk(W2_WAVES * 64);
      if (f16m) wgrad256_kernel<f16><<<grd, blk, lds, st>>>(wv, (int)n_split, n_blocks);
      else wgrad256_kernel<bf16><<<grd, blk, lds, st>>>(wv, (int)n_split, n_blocks);
      BN_LAUNCH_CHECK("wgrad256");
    }
    BnProfScope prof_(BN_K_WGRAD_REDUCE, st);
    wgrad_reduce_kernel<256><<<dim3((unsigned)tiles * 64), 256, 0, st>>>(wv);
    BN_LAUNCH_CHECK("wgrad_reduce");
    return 0;
  }
  // fp32 parity path: 128 x 128 tiles; split the points so that the grid has a few thousand workgroups
  for (int j = 0; j < wv.n_jobs; ++j)
    wv.tile0[j + 1] = wv.tile0[j] + ((wv.job[j].N + 127) / 128) * ((wv.job[j].K + 127) / 128);
  const int tiles = wv.tile0[wv.n_jobs];
  static_assert(WG_BK == 32, "bn_wgrad_splits (field.h) rounds the points per workgroup to WG_BK");
  int64_t mpb = 0;
  wv.n_split = (int)bn_wgrad_splits(false, tiles, Mpad, &mpb);
  wv.m_per_block = (int)mpb;
  BN_REQUIRE((size_t)tiles * wv.n_split * WG_SLAB128 * sizeof(float) <= part_bytes, "wgrad: %d tiles x %d splits do not fit the slab workspace (%zu bytes)",
             tiles, wv.n_split, part_bytes);
  dim3 grid((unsigned)tiles, (unsigned)wv.n_split);
  {
    BnProfScope prof_(BN_K_WGRAD, st);
    wgrad_kernel<float><<<grid, 256, 0, st>>>(wv);
    BN_LAUNCH_CHECK("wgrad");
  }
  BnProfScope prof_(BN_K_WGRAD_REDUCE, st);
  wgrad_reduce_kernel<128><<<dim3((unsigned)tiles * 16), 256, 0, st>>>(wv);
  BN_LAUNCH_CHECK("wgrad_reduce");
  return 0;
}

int bn_launch_skinny(SkinnyArgs &sv, bool bf, bool f16m, int64_t Mpad, int64_t m_per_block, float *part, size_t part_bytes, hipStream_t st) {
  if (sv.n_jobs == 0) return 0;
  sv.part = part;
  sv.m_per_block = (int)m_per_block;
  const int n_split = (int)ceil_div64(Mpad, m_per_block);
  BN_REQUIRE(n_split <= 256, "skinny_wgrad: %d splits (at most 256)", n_split);
  BN_REQUIRE(part && (size_t)sv.n_jobs * n_split * SK_SLAB * sizeof(float) <= part_bytes, "skinny_wgrad: %d jobs x %d splits do not fit the slab workspace",
             sv.n_jobs, n_split);
  dim3 grid((unsigned)n_split, (unsigned)sv.n_jobs);
  {
    BnProfScope prof_(BN_K_SKINNY, st);
    if (f16m) skinny_wgrad_kernel<f16><<<grid, 256, 0, st>>>(sv);
    else if (bf) skinny_wgrad_kernel<bf16><<<grid, 256, 0, st>>>(sv);
    else skinny_wgrad_kernel<float><<<grid, 256, 0, st>>>(sv);
    BN_LAUNCH_CHECK("skinny_wgrad");
  }
  BnProfScope prof_(BN_K_WGRAD_REDUCE, st);
  skinny_reduce_kernel<<<dim3((unsigned)sv.n_jobs, 4, 16), 256, 0, st>>>(sv, n_split);
  BN_LAUNCH_CHECK("skinny_reduce");
  return 0;
}
