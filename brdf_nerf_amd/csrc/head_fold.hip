// Folding of the linear feats layer into the heads' first layers, and the chain rule back - the small fp32 products of
// the training step that ran as eight ATen / hipBLASLt launches per step in round 2 (FieldSpec.fold / unfold_grads):
//   fold:    W'_h = W1_h[:, :F] Wf            b'_h = W1_h[:, :F] bf + b1_h               (and zero the folded-gradient buffers)
//   unfold:  dW1_h[:, :F] += M_h Wf^T + s_h bf^T    db1_h += s_h    dWf += sum_h W1_h^T M_h    dbf += sum_h W1_h^T s_h
// with M_h = dL/dW'_h, s_h = dL/db'_h accumulated by the weight-gradient kernels.  feats_from_xyz is linear and feeds only
// the heads' first (linear) layers (models/spsbrdfnerf.py:694-755): W1 (Wf y + bf) + b1 = (W1 Wf) y + (W1 bf + b1).
// Exact fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32 = an fmaf chain over a fixed permutation of k): one workgroup per 32x32 output tile, the
// contraction split over its four waves, partial tiles summed through LDS.  The operands (<= 1 MB each) sit in L2.
#include "common.h"

struct FoldArgs {
  bn_fold_desc d;
  int tiles_per_head;      // fold: (rows/32) * (F/32)
  int tiles_dw1, tiles_dwf;
};

// acc (32x32, this wave's share of k in [k0, k1), both multiples of 8) += sum_k A(m0 + i, k) B(k, n0 + j);  element strides (sam, sak),
// (sbk, sbn).  Lane (r, h) feeds the contraction indices k + 4 h + {0..3} of every group of 8 - the same permutation on both
// operands, so an operand that is contiguous along k (sak == 1 / sbk == 1: rows of M, W1, Wf^T) is read with ONE 16-byte load per
// lane and four MFMA steps instead of four strided 4-byte loads.
__device__ __forceinline__ f32x4 k4_load(const float *p, int64_t sk, bool vec) {
  if (vec) return *(const f32x4 *)p;
  return f32x4{p[0], p[sk], p[2 * sk], p[3 * sk]};
}
__device__ __forceinline__ void tile_mac(f32x16 &acc, const float *A, int64_t sam, int64_t sak, const float *B, int64_t sbk, int64_t sbn,
                                         int k0, int k1, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const bool a4 = sak == 1 && (sam & 3) == 0 && ((uintptr_t)A & 15) == 0;
  const bool b4 = sbk == 1 && (sbn & 3) == 0 && ((uintptr_t)B & 15) == 0;
  const float *a = A + r * sam + 4 * h * sak, *b = B + r * sbn + 4 * h * sbk;
  int k = k0;
  for (; k + 32 <= k1; k += 32) {
    f32x4 av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = k4_load(a + (int64_t)(k + 8 * u) * sak, sak, a4);
      bv[u] = k4_load(b + (int64_t)(k + 8 * u) * sbk, sbk, b4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][q], bv[u][q], acc, 0, 0, 0);
  }
  for (; k + 8 <= k1; k += 8) {
    const f32x4 av = k4_load(a + (int64_t)k * sak, sak, a4), bv = k4_load(b + (int64_t)k * sbk, sbk, b4);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
  }
}
// sum the four waves' partial tiles; the result lands in wave 0's accumulator
__device__ __forceinline__ void tile_reduce(f32x16 &acc, float *red, int wave, int lane) {
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) red[((wave - 1) * 16 + i) * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] += red[(w * 16 + i) * 64 + lane];
  }
}

__global__ __launch_bounds__(256) void fold_kernel(const FoldArgs A) {
  __shared__ float red[3 * 16 * 64];
  const bn_fold_desc &d = A.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int F = d.F, rows = d.rows;
  const int head = blockIdx.x / A.tiles_per_head, t = blockIdx.x % A.tiles_per_head;
  const float *w1 = d.w1[head];
  const int64_t ld1 = d.w1_ld[head];
  const int nt = F / 32, m0 = (t / nt) * 32, n0 = (t % nt) * 32;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int kq = F / 4;
  tile_mac(acc, w1 + (int64_t)m0 * ld1, ld1, 1, d.wf + n0, F, 1, wave * kq, (wave + 1) * kq, lane);
  tile_reduce(acc, red, wave, lane);
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) d.w_fold[head][(int64_t)(m0 + (i & 3) + 8 * (i >> 2) + 4 * h) * F + n0 + r] = acc[i];
  }
  // the folded weight-gradient accumulator M_h starts the step at zero: this block clears its own 32x32 tile
  if (d.m[head]) {
    const int rr = tid >> 3, cc = (tid & 7) * 4;
    *(f32x4 *)(d.m[head] + (int64_t)(m0 + rr) * F + n0 + cc) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // b'_h = W1 bf + b1 for the tile row's 32 rows (the tiles of the first column block): wave w takes rows m0 + 8 w .. + 7,
  // lanes stride over k (coalesced), eight independent sums in flight; the folded bias-gradient accumulator is cleared too
  if (n0 == 0) {
    float sum[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sum[i] = 0.f;
    const float *wr = w1 + (int64_t)(m0 + 8 * wave) * ld1;
    for (int k = lane; k < F; k += 64) {
      const float bk = d.bf[k];
#pragma unroll
      for (int i = 0; i < 8; ++i) sum[i] = fmaf(wr[(int64_t)i * ld1 + k], bk, sum[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sum[i] += __shfl_xor(sum[i], o);
    }
    if (lane < 8) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) v = lane == i ? sum[i] : v;
      const int n = m0 + 8 * wave + lane;
      d.b_fold[head][n] = v + d.b1[head][n];
      if (d.s[head]) d.s[head][n] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void unfold_kernel(const FoldArgs A) {
  __shared__ float red[3 * 16 * 64];
  const bn_fold_desc &d = A.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int F = d.F, rows = d.rows, nt = F / 32;
  int b = blockIdx.x;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (b < A.tiles_dw1) {
    // dW1_h[m0.., n0..] += M_h Wf^T + s_h bf^T      (Wf^T: B(k, n) = Wf[n][k])
    const int head = b / A.tiles_per_head, t = b % A.tiles_per_head;
    if (!d.m[head] || !d.d_w1[head]) return;
    const int m0 = (t / nt) * 32, n0 = (t % nt) * 32, kq = F / 4;
    tile_mac(acc, d.m[head] + (int64_t)m0 * F, F, 1, d.wf + (int64_t)n0 * F, 1, F, wave * kq, (wave + 1) * kq, lane);
    tile_reduce(acc, red, wave, lane);
    if (wave == 0) {
      const float bfn = d.bf[n0 + r];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
        float *p = d.d_w1[head] + (int64_t)m * d.d_w1_ld[head] + n0 + r;
        *p += acc[i] + d.s[head][m] * bfn;
      }
    }
    if (n0 == 0 && tid < 32 && d.d_b1[head]) d.d_b1[head][m0 + tid] += d.s[head][m0 + tid];      // db1_h += s_h
    return;
  }
  b -= A.tiles_dw1;
  if (b < A.tiles_dwf) {
    // dWf[m0.., n0..] += sum_h W1_h^T M_h          (A(m, k) = W1_h[k][m])
    if (!d.d_wf) return;
    const int m0 = (b / nt) * 32, n0 = (b % nt) * 32, kq = rows / 4;
    for (int head = 0; head < d.n_heads; ++head)
      if (d.m[head])
        tile_mac(acc, d.w1[head] + m0, 1, d.w1_ld[head], d.m[head] + n0, F, 1, wave * kq, (wave + 1) * kq, lane);
    tile_reduce(acc, red, wave, lane);
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) d.d_wf[(int64_t)(m0 + (i & 3) + 8 * (i >> 2) + 4 * h) * F + n0 + r] += acc[i];
    }
    // dbf[m0 ..] += sum_h W1_h[:, m0 ..]^T s_h  (the tiles of the first column block): 32 columns x 8 row groups, summed through LDS
    if (n0 == 0 && d.d_bf) {
      const int kk = tid & 31, grp = tid >> 5;
      float sum = 0.f;
      for (int head = 0; head < d.n_heads; ++head)
        if (d.s[head])
          for (int n = grp; n < rows; n += 8) sum = fmaf(d.w1[head][(int64_t)n * d.w1_ld[head] + m0 + kk], d.s[head][n], sum);
      __syncthreads();               // (wave 0 has read the partial tiles)
      red[grp * 32 + kk] = sum;
      __syncthreads();
      if (tid < 32) {
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) v += red[q * 32 + tid];
        d.d_bf[m0 + tid] += v;
      }
    }
    return;
  }
}

static int fold_check(const bn_fold_desc *d, const char *what) {
  BN_REQUIRE(d && d->n_heads >= 1 && d->n_heads <= BN_MAX_HEADS && d->F >= 64 && d->F % 64 == 0 && d->rows >= 32 && d->rows % 32 == 0 &&
                 d->wf && d->bf,
             "%s: bad descriptor (F=%d rows=%d heads=%d)", what, d ? d->F : 0, d ? d->rows : 0, d ? d->n_heads : 0);
  for (int i = 0; i < d->n_heads; ++i) BN_REQUIRE(d->w1[i] && d->w1_ld[i] >= d->F, "%s: head %d first-layer weight missing", what, i);
  return 0;
}

extern "C" int bn_fold_heads(const bn_fold_desc *d, void *stream) {
  if (int e = fold_check(d, "fold_heads")) return e;
  for (int i = 0; i < d->n_heads; ++i) BN_REQUIRE(d->b1[i] && d->w_fold[i] && d->b_fold[i], "fold_heads: head %d buffers missing", i);
  FoldArgs a;
  a.d = *d; a.tiles_per_head = (d->rows / 32) * (d->F / 32); a.tiles_dw1 = a.tiles_dwf = 0;
  BnProfScope prof_(BN_K_PACK, (hipStream_t)stream);
  fold_kernel<<<dim3((unsigned)(d->n_heads * a.tiles_per_head)), 256, 0, (hipStream_t)stream>>>(a);
  BN_LAUNCH_CHECK("fold_heads");
  return 0;
}

extern "C" int bn_unfold_heads(const bn_fold_desc *d, void *stream) {
  if (int e = fold_check(d, "unfold_heads")) return e;
  FoldArgs a;
  a.d = *d; a.tiles_per_head = (d->rows / 32) * (d->F / 32);
  a.tiles_dw1 = d->n_heads * a.tiles_per_head; a.tiles_dwf = (d->F / 32) * (d->F / 32);
  BnProfScope prof_(BN_K_PACK, (hipStream_t)stream);
  unfold_kernel<<<dim3((unsigned)(a.tiles_dw1 + a.tiles_dwf)), 256, 0, (hipStream_t)stream>>>(a);
  BN_LAUNCH_CHECK("unfold_heads");
  return 0;
}
