// Weight-gradient kernels of the field backward (csrc/field_wgrad.hip): job descriptions built by bn_field_backward
// (csrc/field_bwd.hip) and the launchers it calls.  The kernels live in their own translation unit because they want another
// instruction-scheduling strategy than the chain kernels (brdf_nerf_amd/build.py FILE_FLAGS, profiles/history/r02_ablation.txt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct WgradJob {
  const void *A;   // [Mpad][lda] T : gradient rows (dZ / dFeats / dG)
  const void *B;   // [Mpad][ldb] T : layer input rows (PE / Y / feats)
  float *C;        // [N][ldc] fp32, +=
  float *bias;     // [N] fp32, += column sums of A (nullable)
  int lda, ldb, ldc;
  int a_col0, b_col0;  // first column used in A / B
  int N, K;            // valid output extents (rows of C, cols of C)
  int scale_sel;       // fp16 loss scaling carried by A: 0 none, 1 the primal chain's (amax[0]), 2 the adjoint chain's (amax[1])
  int b_native;        // 16-bit modes, bit 0: B is a layer-output stash in accumulator-native order (tiles of b_bm points, F columns:
  int b_bm, b_F;       // chunk (col/32, point/32 % (bm/32), (col%32)/16) = 64 lanes x 16 B, see native_off8); else row-major [Mpad][ldb]
  int b_bm_shift;      // log2(b_bm)
                       // b_native bit 1 (WG_A_NATIVE): A is a native-order image too (the trunk's dZ_l, written from the backward
                       // chain's epilogue registers; same b_bm / b_F geometry, a_col0 = 0); else row-major [Mpad][lda]
};
#define WG_B_NATIVE 1
#define WG_A_NATIVE 2
#define BN_MAX_WGRAD_JOBS 44
struct WgradArgs {
  WgradJob job[BN_MAX_WGRAD_JOBS];
  int tile0[BN_MAX_WGRAD_JOBS + 1];  // prefix sum of 128x128 output tiles per job
  int n_jobs;
  int64_t Mpad;
  int m_per_block;                   // points per split (multiple of 32)
  const float *amax;                 // fp16 mode only (else nullptr): see wg_unscale
  float *part;                       // slab workspace: [output tile of the launch][point split][WG_SLAB256 / WG_SLAB128 floats]
  int n_split;                       // point splits of this launch
  int chain_next[BN_MAX_WGRAD_JOBS]; // next job that adds into the same matrix (-1: none): summed by the head job's reduce blocks
  unsigned char head[BN_MAX_WGRAD_JOBS];
};
// floats per slab: the accumulators of one workgroup in register order + its bias column sums
#define WG_SLAB256 (256 * 256 + 256)
#define WG_SLAB128 (128 * 128 + 128)
#define SK_SLAB (4 * 512 + 4)

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
  float *part;             // slab workspace: [job][point split][SK_SLAB floats]
};

// Launch the weight-gradient jobs of `wv` (tile0 / m_per_block / n_split / chains are filled in here) and the fixed-order sum of
// their slabs into the gradient.  bf: 16-bit modes (256 x 256 tiles, wgrad256_kernel), f16m: fp16; part / part_bytes: the slab
// workspace (StashLayout.wgpart).
int bn_launch_wgrad(WgradArgs &wv, bool bf, bool f16m, int64_t Mpad, float *part, size_t part_bytes, hipStream_t st);
// Launch the skinny (<= 4 rows) jobs of `sv` with `m_per_block` points per split, and their reduce.
int bn_launch_skinny(SkinnyArgs &sv, bool bf, bool f16m, int64_t Mpad, int64_t m_per_block, float *part, size_t part_bytes, hipStream_t st);
// (bn_device_faults bit 1: a lost hand-over of the barrier-free backward trunk; rounds 1-3 reported their turn-taking mode there)
unsigned int bn_bwd_fault_read(hipStream_t st);
