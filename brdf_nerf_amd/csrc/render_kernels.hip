// Per-ray kernels of the render path (HBM-bound, one 64-lane wavefront per ray):
//   stratified depths   get_z_vals                 rendering.py:149-166
//   compositing scan    cal_weight + weighted sums models/spsbrdfnerf.py:50-69, :198-338
//   guided resampling   GenerateGuidedSamples..sample_pdf + merge   rendering.py:13-91,116-147,263-272
// fp32 throughout; FMA contraction is disabled so that the depth/sample arithmetic rounds like the
// reference's separate ATen ops (needed for bit-exact sample indices).
#include "common.h"
#pragma clang fp contract(off)

#define WAVES_PER_BLOCK 4

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// torch.linspace(start, end, steps)[i] in fp32 (symmetric formula of ATen's RangeFactories).
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
  const float step = (end - start) / (float)(steps - 1);
  return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - i - 1);
}

// ------------------------------------------------------------------------------------------ in-kernel random draws
// Philox4x32-10 (Salmon et al., SC'11) keyed by a seed, counter = (element index, draw stream id, step counter).  The fused
// training step (strict_rng = False) draws its uniforms in the kernels that consume them instead of launching one ATen
// fill per draw; seed and step counter live in device memory (rng[0], rng[1]) so that a captured HIP graph replays with
// fresh numbers - the step's last kernel (bn_adam_multi) advances the counter.  The reference-replaying paths (render_rays,
// strict_rng = True) keep taking torch's draws as arrays.
__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (unsigned int)p1; c[3] = (unsigned int)p0; c[0] = n0; c[2] = n2;
}
__device__ __forceinline__ void philox4(unsigned long long seed, unsigned long long step, unsigned int stream, unsigned long long index,
                                        unsigned int (&out)[4]) {
  unsigned int c[4] = {(unsigned int)index, (unsigned int)(index >> 32), stream, (unsigned int)step};
  unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32) ^ (unsigned int)(step >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}
// uniform in [0, 1) with 24 random bits, like torch.rand for float32
__device__ __forceinline__ float philox_uniform(const unsigned long long *rng, unsigned int stream, unsigned long long index) {
  unsigned int o[4];
  philox4(rng[0], rng[1], stream, index >> 2, o);
  return (float)(o[index & 3] >> 8) * 5.9604644775390625e-08f;
}

// standard normal by Box-Muller on one pair of a Philox block (two normals per block: pair index & 1); u1 in (0, 1]
__device__ __forceinline__ float philox_normal(const unsigned long long *rng, unsigned int stream, unsigned long long index) {
  unsigned int o[4];
  philox4(rng[0], rng[1], stream, index >> 1, o);
  const int q = (int)(index & 1) * 2;
  const float u1 = (float)((o[q] >> 8) + 1u) * 5.9604644775390625e-08f, u2 = (float)(o[q + 1] >> 8) * 5.9604644775390625e-08f;
  return sqrtf(-2.f * logf(u1)) * cospif(2.f * u2);
}
// sigma + noise of sample position s of ray `ray` in a compositing over S positions (bn_noise; rng == nullptr: sigma as it is)
struct NoiseArgs {
  const unsigned long long *rng;
  float noise_std;
  unsigned int stream;
  int64_t ray_offset;
};
// (noise_std < 0, ABI 7: the value lives in the step state the draws come from - f32 at BN_STATE_NOISE_OFF - so a step whose
// noise decays every step (main.py:246) keeps ONE launch signature and stays inside its captured graph)
__device__ __forceinline__ float noised(const NoiseArgs &N, float sg, int64_t ray, int S, int s) {
  if (!N.rng) return sg;
  const float sd = N.noise_std < 0.f ? ((const float *)N.rng)[BN_STATE_NOISE_OFF / 4] : N.noise_std;
  return sg + philox_normal(N.rng, N.stream, (unsigned long long)((ray + N.ray_offset) * S + s)) * sd;
}
static NoiseArgs make_noise(const bn_noise *n) {
  NoiseArgs a = {nullptr, 0.f, 0u, 0};
  if (n && n->rng && n->noise_std != 0.f) { a.rng = n->rng; a.noise_std = n->noise_std; a.stream = n->rng_stream; a.ray_offset = n->ray_offset; }
  return a;
}

// ------------------------------------------------------------------------------------------ stratified z
__global__ void stratified_z_kernel(const float *near, const float *far, int64_t nf_stride, const float *u,
                                    const unsigned long long *rng, unsigned int rng_stream, int64_t draw_offset, int64_t R, int S,
                                    float *z) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R * S) return;
  const int64_t ray = i / S;
  const int s = (int)(i % S);
  const float n = near[ray * nf_stride], f = far[ray * nf_stride];
  auto zc = [&](int j) {
    const float t = linspace_at(0.f, 1.f, S, j);
    return n * (1.f - t) + f * t;
  };
  const float zi = zc(s);
  const float lower = s == 0 ? zi : 0.5f * (zc(s - 1) + zi);
  const float upper = s == S - 1 ? zi : 0.5f * (zi + zc(s + 1));
  z[i] = lower + (upper - lower) * (u ? u[i] : philox_uniform(rng, rng_stream, (unsigned long long)(i + draw_offset)));
}

extern "C" int bn_stratified_z(const float *near, const float *far, int64_t nf_stride, const float *u, int64_t R,
                               int32_t S, float *z, void *stream) {
  BN_REQUIRE(near && far && u && z && R > 0 && S >= 2, "stratified_z: bad arguments");
  const int64_t n = R * S;
  BnProfScope prof_(BN_K_STRATIFIED, (hipStream_t)stream);
  stratified_z_kernel<<<dim3((unsigned)ceil_div64(n, 256)), 256, 0, (hipStream_t)stream>>>(near, far, nf_stride, u, nullptr, 0u, 0, R, S, z);
  BN_LAUNCH_CHECK("stratified_z");
  return 0;
}

extern "C" int bn_stratified_z_rng(const float *near, const float *far, int64_t nf_stride, const unsigned long long *rng,
                                   uint32_t rng_stream, int64_t ray_offset, int64_t R, int32_t S, float *z, void *stream) {
  BN_REQUIRE(near && far && rng && z && R > 0 && S >= 2, "stratified_z_rng: bad arguments");
  const int64_t n = R * S;
  BnProfScope prof_(BN_K_STRATIFIED, (hipStream_t)stream);
  stratified_z_kernel<<<dim3((unsigned)ceil_div64(n, 256)), 256, 0, (hipStream_t)stream>>>(near, far, nf_stride, nullptr, rng, rng_stream, ray_offset * S, R, S, z);
  BN_LAUNCH_CHECK("stratified_z_rng");
  return 0;
}

// fill u[n] with the uniforms stream `rng_stream` would hand to element 0 .. n-1 (tests: the in-kernel draws as an array)
__global__ void rng_uniform_kernel(const unsigned long long *rng, unsigned int rng_stream, int64_t n, float *u) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) u[i] = philox_uniform(rng, rng_stream, (unsigned long long)i);
}
__global__ void rng_normal_kernel(const unsigned long long *rng, unsigned int rng_stream, int64_t n, float *x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = philox_normal(rng, rng_stream, (unsigned long long)i);
}
extern "C" int bn_rng_normal(const unsigned long long *rng, uint32_t rng_stream, int64_t n, float *x, void *stream) {
  BN_REQUIRE(rng && x && n > 0, "rng_normal: bad arguments");
  rng_normal_kernel<<<dim3((unsigned)ceil_div64(n, 256)), 256, 0, (hipStream_t)stream>>>(rng, rng_stream, n, x);
  BN_LAUNCH_CHECK("rng_normal");
  return 0;
}
extern "C" int bn_rng_uniform(const unsigned long long *rng, uint32_t rng_stream, int64_t n, float *u, void *stream) {
  BN_REQUIRE(rng && u && n > 0, "rng_uniform: bad arguments");
  rng_uniform_kernel<<<dim3((unsigned)ceil_div64(n, 256)), 256, 0, (hipStream_t)stream>>>(rng, rng_stream, n, u);
  BN_LAUNCH_CHECK("rng_uniform");
  return 0;
}

// ------------------------------------------------------------------------------------------ compositing
// Lane i of the ray's wave owns samples [i*c, (i+1)*c), c = ceil(S/64) <= BN_MAX_C: the exclusive prefix product
// of (1 - alpha + 1e-10) is a per-lane serial product + a 6-step wavefront shuffle scan.
#define BN_MAX_CPL 8   // samples per lane -> S <= 512

struct CompArgs {
  const float *z, *sigma, *noise, *chan;
  int64_t sigma_stride, chan_stride;
  float noise_std;
  int C, S;
  int64_t R;
  float *alphas, *trans, *weights, *depth, *acc;
  // backward
  const float *d_weights, *d_depth, *d_acc;
  float *d_sigma, *d_chan;
  int64_t d_sigma_stride, d_chan_stride;
  int flat_lg;     // >= 0: dense channel rows (stride == C): the channel phase walks the ray's [S][C] block in 4-channel groups,
                   // 2^flat_lg groups per sample, consecutive lanes on consecutive pieces - instead of one 4-byte load per
                   // (sample, channel) per lane.  -1: generic path (strided rows)
  int flat_vec;    // C % 4 == 0 and 16-byte aligned blocks: one float4 per group
  int fused_dsigma;  // backward, flat path: d_sigma aliases channel 3 of d_chan -> written with the d_chan rows
};

// Channel phase of the flat path.  The ray's [S][C] block is walked in 4-channel groups: lane l handles group q = l + 64 k,
// i.e. sample q >> LG, channels 4 cg .. 4 cg + 3 with cg = l & (2^LG - 1) (the same for every k; 2^LG = groups per sample
// = ceil(C / 4) rounded up to a power of two).  Consecutive lanes read consecutive 16-byte pieces: one wave instruction
// moves ~1 KB of consecutive bytes.  VEC (C % 4 == 0, 16-byte aligned rows): one float4 per group; otherwise the group's
// valid channels as scalars (the same cache lines, four instructions).  Per-sample weights come from the wave's LDS slice.
template <bool VEC> __device__ __forceinline__ f32x4 comp_ld4(const float *p, int nv) {
  if (VEC) return nv > 0 ? *(const f32x4 *)p : f32x4{0.f, 0.f, 0.f, 0.f};   // C = 12: the fourth group of a sample is empty
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (nv > 0) v[0] = p[0];
  if (nv > 1) v[1] = p[1];
  if (nv > 2) v[2] = p[2];
  if (nv > 3) v[3] = p[3];
  return v;
}
template <bool VEC> __device__ __forceinline__ void comp_st4(float *p, const f32x4 &v, int nv) {
  if (VEC) { if (nv > 0) *(f32x4 *)p = v; return; }
  if (nv > 0) p[0] = v[0];
  if (nv > 1) p[1] = v[1];
  if (nv > 2) p[2] = v[2];
  if (nv > 3) p[3] = v[3];
}
template <int LG, bool VEC> __device__ __forceinline__ void comp_flat_fwd(const CompArgs &A, int64_t ray, int lane, const float *wl) {
  constexpr int LPS = 1 << LG;
  const int nq = A.S << LG, c0 = 4 * (lane & (LPS - 1)), nv = A.C - c0;
  const float *src = A.chan + ray * A.S * A.C + c0;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int q = lane; q < nq; q += 64) {
    const int sm = q >> LG;
    a += wl[sm] * comp_ld4<VEC>(src + (int64_t)sm * A.C, nv);
  }
#pragma unroll
  for (int o = LPS; o < 64; o <<= 1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) a[e] += __shfl_xor(a[e], o);
  }
  if (lane < LPS) comp_st4<VEC>(A.acc + ray * A.C + c0, a, nv);
}
// backward, part A: gs[s] = sum_c d_acc[c] chan[s][c]
template <int LG, bool VEC> __device__ __forceinline__ void comp_flat_bwd_dot(const CompArgs &A, int64_t ray, int lane, float *gs) {
  constexpr int LPS = 1 << LG;
  const int nq = A.S << LG, c0 = 4 * (lane & (LPS - 1)), nv = A.C - c0;
  const float *src = A.chan + ray * A.S * A.C + c0;
  const f32x4 da = comp_ld4<VEC>(A.d_acc + ray * A.C + c0, nv);
  for (int q = lane; q < nq; q += 64) {
    const int sm = q >> LG;
    const f32x4 v = comp_ld4<VEC>(src + (int64_t)sm * A.C, nv);
    float p = da[0] * v[0] + da[1] * v[1] + da[2] * v[2] + da[3] * v[3];
#pragma unroll
    for (int o = 1; o < LPS; o <<= 1) p += __shfl_xor(p, o);
    if ((lane & (LPS - 1)) == 0) gs[sm] = p;
  }
}
// backward, part B: d_chan[s][c] = w_s d_acc[c], channel 3 replaced by d sigma_s when the two alias
template <int LG, bool VEC> __device__ __forceinline__ void comp_flat_bwd_store(const CompArgs &A, int64_t ray, int lane,
                                                                                const float *wl, const float *dsl) {
  constexpr int LPS = 1 << LG;
  const int nq = A.S << LG, c0 = 4 * (lane & (LPS - 1)), nv = A.C - c0;
  float *dst = A.d_chan + ray * A.S * A.C + c0;
  const f32x4 da = comp_ld4<VEC>(A.d_acc + ray * A.C + c0, nv);
  const bool sig = A.fused_dsigma && c0 == 0;
  for (int q = lane; q < nq; q += 64) {
    const int sm = q >> LG;
    f32x4 o = wl[sm] * da;
    if (sig) o[3] = dsl[sm];
    comp_st4<VEC>(dst + (int64_t)sm * A.C, o, nv);
  }
}
// dispatch on (groups per sample, vector width)
#define BN_COMP_FLAT(FN, ...)                                                     \
  do {                                                                            \
    if (A.flat_vec) {                                                             \
      if (A.flat_lg == 0) FN<0, true>(__VA_ARGS__);                               \
      else if (A.flat_lg == 1) FN<1, true>(__VA_ARGS__);                          \
      else if (A.flat_lg == 2) FN<2, true>(__VA_ARGS__);                          \
      else FN<3, true>(__VA_ARGS__);                                              \
    } else {                                                                      \
      if (A.flat_lg == 0) FN<0, false>(__VA_ARGS__);                              \
      else if (A.flat_lg == 1) FN<1, false>(__VA_ARGS__);                         \
      else if (A.flat_lg == 2) FN<2, false>(__VA_ARGS__);                         \
      else FN<3, false>(__VA_ARGS__);                                             \
    }                                                                             \
  } while (0)
__device__ __forceinline__ void lds_wave_sync() {   // LDS hand-over inside one wave: its DS operations execute in order
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float wave_excl_prod(float v, int lane) {
  // inclusive Hillis-Steele scan, then shift by one lane
  float p = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_up(p, o);
    if (lane >= o) p *= t;
  }
  const float e = __shfl_up(p, 1);
  return lane == 0 ? 1.f : e;
}
__device__ __forceinline__ float wave_excl_sum_rev(float v, int lane) {
  // exclusive suffix sum: sum of v over lanes > lane
  float p = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float t = __shfl_down(p, o);
    if (lane + o < 64) p += t;
  }
  const float e = __shfl_down(p, 1);
  return lane == 63 ? 0.f : e;
}

template <bool BWD> __global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void composite_kernel(const CompArgs A) {
  __shared__ float lds_w[WAVES_PER_BLOCK][64 * BN_MAX_CPL];                 // per-wave slices: weights,
  __shared__ float lds_g[BWD ? WAVES_PER_BLOCK : 1][BWD ? 64 * BN_MAX_CPL : 1];   // sum_c d_acc[c] chan[s][c],
  __shared__ float lds_d[BWD ? WAVES_PER_BLOCK : 1][BWD ? 64 * BN_MAX_CPL : 1];   // d sigma
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wv;
  if (ray >= A.R) return;
  float *wl = lds_w[wv], *gs = lds_g[BWD ? wv : 0], *dsl = lds_d[BWD ? wv : 0];
  const bool flat = A.flat_lg >= 0;
  const int S = A.S, cpl = (S + 63) / 64;
  const float *z = A.z + ray * S;
  float zv[BN_MAX_CPL], al[BN_MAX_CPL], u[BN_MAX_CPL], dad[BN_MAX_CPL];  // dad = d alpha / d sigma
  float lp = 1.f;
#pragma unroll
  for (int j = 0; j < BN_MAX_CPL; ++j) {
    zv[j] = 0.f; al[j] = 0.f; u[j] = 1.f; dad[j] = 0.f;
    const int s = lane * cpl + j;
    if (j < cpl && s < S) {
      zv[j] = z[s];
      const float delta = s == S - 1 ? 1e10f : z[s + 1] - zv[j];
      float sg = A.sigma[(ray * S + s) * A.sigma_stride];
      if (A.noise) sg = sg + A.noise[ray * S + s] * A.noise_std;
      const float rs = sg > 0.f ? sg : 0.f;
      const float e = expf(-delta * rs);
      al[j] = 1.f - e;
      u[j] = 1.f - al[j] + 1e-10f;
      dad[j] = sg > 0.f ? delta * e : 0.f;
      lp *= u[j];
    }
  }
  float T = wave_excl_prod(lp, lane);  // transparency before this lane's first sample
  float tr[BN_MAX_CPL], w[BN_MAX_CPL];
#pragma unroll
  for (int j = 0; j < BN_MAX_CPL; ++j) {
    tr[j] = T;
    w[j] = al[j] * T;
    T *= u[j];
  }
  if (!BWD) {
    float dsum = 0.f;
#pragma unroll
    for (int j = 0; j < BN_MAX_CPL; ++j) {
      const int s = lane * cpl + j;
      if (j < cpl && s < S) {
        const int64_t o = ray * S + s;
        if (A.alphas) A.alphas[o] = al[j];
        if (A.trans) A.trans[o] = tr[j];
        if (A.weights) A.weights[o] = w[j];
        dsum += w[j] * zv[j];
      }
    }
    dsum = wave_sum(dsum);
    if (lane == 0 && A.depth) A.depth[ray] = dsum;
    if (flat) {
#pragma unroll
      for (int j = 0; j < BN_MAX_CPL; ++j) {
        const int s = lane * cpl + j;
        if (j < cpl && s < S) wl[s] = w[j];
      }
      lds_wave_sync();
      BN_COMP_FLAT(comp_flat_fwd, A, ray, lane, wl);
    } else if (A.chan && A.acc) {
      for (int c = 0; c < A.C; ++c) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < BN_MAX_CPL; ++j) {
          const int s = lane * cpl + j;
          if (j < cpl && s < S) a += w[j] * A.chan[(ray * S + s) * A.chan_stride + c];
        }
        a = wave_sum(a);
        if (lane == 0) A.acc[ray * A.C + c] = a;
      }
    }
  } else {
    // g_s = dL/dw_s; dL/dalpha_s = g_s T_s - (1/u_s) sum_{k>s} g_k w_k   
    const float dd = A.d_depth ? A.d_depth[ray] : 0.f;
    if (flat) {
#pragma unroll
      for (int j = 0; j < BN_MAX_CPL; ++j) {
        const int s = lane * cpl + j;
        if (j < cpl && s < S) wl[s] = w[j];
      }
      BN_COMP_FLAT(comp_flat_bwd_dot, A, ray, lane, gs);
      lds_wave_sync();
    }
    float g[BN_MAX_CPL], gw = 0.f;
#pragma unroll
    for (int j = 0; j < BN_MAX_CPL; ++j) {
      g[j] = 0.f;
      const int s = lane * cpl + j;
      if (j < cpl && s < S) {
        float gg = dd * zv[j];
        if (A.d_weights) gg += A.d_weights[ray * S + s];
        if (flat) gg += gs[s];
        else if (A.chan && A.d_acc) {
          const float *ch = A.chan + (ray * S + s) * A.chan_stride;
          float *dch = A.d_chan ? A.d_chan + (ray * S + s) * A.d_chan_stride : nullptr;
          for (int c = 0; c < A.C; ++c) {          // (generic strided rows only: the dense layout takes the flat path)
            const float dc = A.d_acc[ray * A.C + c];
            gg += dc * ch[c];
            if (dch) dch[c] = w[j] * dc;
          }
        }
        g[j] = gg;
        gw += gg * w[j];
      }
    }
    float suffix = wave_excl_sum_rev(gw, lane);  // sum of g*w over later lanes
#pragma unroll
    for (int j = BN_MAX_CPL - 1; j >= 0; --j) {
      const int s = lane * cpl + j;
      if (j < cpl && s < S) {
        const float dalpha = g[j] * tr[j] - suffix / u[j];
        if (flat && A.fused_dsigma) dsl[s] = dalpha * dad[j];
        else A.d_sigma[(ray * S + s) * A.d_sigma_stride] = dalpha * dad[j];
        suffix += g[j] * w[j];
      }
    }
    if (flat && A.d_chan) {
      lds_wave_sync();
      BN_COMP_FLAT(comp_flat_bwd_store, A, ray, lane, wl, dsl);
    }
  }
}

// flat channel path: dense rows (any C <= 32); float4 groups when C % 4 == 0 and every block is 16-byte aligned
static void comp_flat_config(CompArgs &a, const float *chan, int64_t chan_stride, int C, const void *p1, const void *p2) {
  a.flat_lg = -1; a.flat_vec = 0;
  if (!chan || chan_stride != C || C < 1) return;
  a.flat_lg = C <= 4 ? 0 : (C <= 8 ? 1 : (C <= 16 ? 2 : 3));
  a.flat_vec = (C % 4 == 0 && ((uintptr_t)chan | (uintptr_t)p1 | (uintptr_t)p2) % 16 == 0) ? 1 : 0;
}

extern "C" int bn_composite_forward(const float *z, const float *sigma, int64_t sigma_stride, const float *noise,
                                    float noise_std, const float *chan, int64_t chan_stride, int32_t C, int64_t R,
                                    int32_t S, float *alphas, float *trans, float *weights, float *depth, float *acc,
                                    void *stream) {
  BN_REQUIRE(z && sigma && R > 0 && S >= 1 && S <= 64 * BN_MAX_CPL, "composite: bad arguments (S=%d)", S);
  BN_REQUIRE(C >= 0 && C <= BN_MAX_CH, "composite: C=%d > %d", C, BN_MAX_CH);
  CompArgs a = {};
  a.z = z; a.sigma = sigma; a.noise = noise; a.chan = chan; a.sigma_stride = sigma_stride; a.chan_stride = chan_stride;
  a.noise_std = noise_std; a.C = C; a.S = S; a.R = R;
  a.alphas = alphas; a.trans = trans; a.weights = weights; a.depth = depth; a.acc = acc;
  a.flat_lg = -1;
  if (acc) comp_flat_config(a, chan, chan_stride, C, acc, nullptr);
#ifdef BN_NO_FLAT_COMPOSITE      // A/B switch (profiles/ab_kernels.py): the per-(sample, channel) scalar path everywhere
  a.flat_lg = -1;
#endif
  BnProfScope prof_(BN_K_COMPOSITE_FWD, (hipStream_t)stream);
  composite_kernel<false><<<dim3((unsigned)ceil_div64(R, WAVES_PER_BLOCK)), 64 * WAVES_PER_BLOCK, 0, (hipStream_t)stream>>>(a);
  BN_LAUNCH_CHECK("composite_forward");
  return 0;
}

extern "C" int bn_composite_backward(const float *z, const float *sigma, int64_t sigma_stride, const float *noise,
                                     float noise_std, const float *chan, int64_t chan_stride, int32_t C, int64_t R,
                                     int32_t S, const float *d_weights, const float *d_depth, const float *d_acc,
                                     float *d_sigma, int64_t d_sigma_stride, float *d_chan, int64_t d_chan_stride,
                                     void *stream) {
  BN_REQUIRE(z && sigma && d_sigma && R > 0 && S >= 1 && S <= 64 * BN_MAX_CPL, "composite_backward: bad arguments");
  BN_REQUIRE(C >= 0 && C <= BN_MAX_CH, "composite_backward: C=%d > %d", C, BN_MAX_CH);
  CompArgs a = {};
  a.z = z; a.sigma = sigma; a.noise = noise; a.chan = chan; a.sigma_stride = sigma_stride; a.chan_stride = chan_stride;
  a.noise_std = noise_std; a.C = C; a.S = S; a.R = R;
  a.d_weights = d_weights; a.d_depth = d_depth; a.d_acc = d_acc; a.d_sigma = d_sigma; a.d_chan = d_chan;
  a.d_sigma_stride = d_sigma_stride; a.d_chan_stride = d_chan_stride;
  // flat path: needs d_acc (the dot products) and, when channel gradients are written, dense d_chan rows as well
  a.flat_lg = -1;
  if (d_acc && (!d_chan || d_chan_stride == C)) comp_flat_config(a, chan, chan_stride, C, d_acc, d_chan);
#ifdef BN_NO_FLAT_COMPOSITE
  a.flat_lg = -1;
#endif
  a.fused_dsigma = (a.flat_lg >= 0 && d_chan && d_sigma == d_chan + 3 && d_sigma_stride == d_chan_stride) ? 1 : 0;
  BnProfScope prof_(BN_K_COMPOSITE_BWD, (hipStream_t)stream);
  composite_kernel<true><<<dim3((unsigned)ceil_div64(R, WAVES_PER_BLOCK)), 64 * WAVES_PER_BLOCK, 0, (hipStream_t)stream>>>(a);
  BN_LAUNCH_CHECK("composite_backward");
  return 0;
}

// ------------------------------------------------------------------------------------------ Lambertian loss glue
// Ray-level tail of a Lambertian training step in one launch: shade (models/spsbrdfnerf.py:270-282: rgb =
// clamp(sum_s w (albedo (1+2p) - p), 0, 1)), SNerfLoss (metrics.py:39-61, lambda_sc = 0) and DepthLoss
// (metrics.py:82-161, subset rule as a mask), and their gradients w.r.t. the composited sums, the depth and the weights -
// what ~60 small ATen launches (forward + autograd) compute otherwise.  One wavefront per ray.
struct LossArgs {
  const float *acc, *weights, *z, *depth, *rgbs, *valid, *tdepth, *tweight, *tstd;
  float *ray_loss, *rgb, *d_acc, *d_depth, *d_weights;
  int64_t R;
  int C, S, usealldepth;
  float pad, lambda_rgb, lambda_ds;
};

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void lambert_loss_kernel(const LossArgs A) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (ray >= A.R) return;
  const float d = A.depth[ray];
  float wsum = 0.f, var = 0.f;
  for (int s = lane; s < A.S; s += 64) {
    const float w = A.weights[ray * A.S + s], dz = A.z[ray * A.S + s] - d;
    wsum += w;
    var += dz * dz * w;
  }
  wsum = wave_sum(wsum);
  var = wave_sum(var);
  const float invn = 1.f / (3.f * (float)A.R);
  float loss = 0.f, dws = 0.f, dacc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float x = A.acc[ray * A.C + c] * (1.f + 2.f * A.pad) - A.pad * wsum;
    const float y = fminf(fmaxf(x, 0.f), 1.f);
    const float e = y - A.rgbs[ray * 3 + c];
    loss += A.lambda_rgb * e * e * invn;
    const float dy = (x >= 0.f && x <= 1.f) ? A.lambda_rgb * 2.f * e * invn : 0.f;   // clamp passes the gradient on [0, 1]
    dacc[c] = dy * (1.f + 2.f * A.pad);
    dws -= dy * A.pad;
    if (lane == 0) A.rgb[ray * 3 + c] = y;
  }
  float dd = 0.f;
  if (A.tdepth && A.valid[ray] > 0.f) {
    const float td = A.tdepth[ray], tw = A.tweight[ray], ts = A.tstd[ray];
    const bool apply = A.usealldepth || (fabsf(d - td) - ts > 0.f) || (ts < sqrtf(var));
    if (apply) {
      const float k = A.lambda_ds / 3.f / (float)A.R;
      loss += k * tw * (d - td) * (d - td);
      dd = k * 2.f * tw * (d - td);
    }
  }
  if (lane == 0) {
    A.ray_loss[ray] = loss;
    A.d_depth[ray] = dd;
  }
  for (int c = lane; c < A.C; c += 64) A.d_acc[ray * A.C + c] = c < 3 ? dacc[c] : 0.f;
  for (int s = lane; s < A.S; s += 64) A.d_weights[ray * A.S + s] = dws;
}

extern "C" int bn_lambert_loss(const float *acc, int32_t C, const float *weights, const float *z, int32_t S, const float *depth,
                               const float *rgbs, const float *valid_depth, const float *target_depth,
                               const float *target_weight, const float *target_std, float rgb_padding, float lambda_rgb,
                               float lambda_ds, int32_t usealldepth, int64_t R, float *ray_loss, float *rgb, float *d_acc,
                               float *d_depth, float *d_weights, void *stream) {
  BN_REQUIRE(acc && weights && z && depth && rgbs && ray_loss && rgb && d_acc && d_depth && d_weights && R > 0 && S >= 1 && C >= 3,
             "lambert_loss: bad arguments");
  BN_REQUIRE(!target_depth || (valid_depth && target_weight && target_std), "lambert_loss: incomplete depth prior");
  LossArgs a;
  a.acc = acc; a.weights = weights; a.z = z; a.depth = depth; a.rgbs = rgbs; a.valid = valid_depth; a.tdepth = target_depth;
  a.tweight = target_weight; a.tstd = target_std; a.ray_loss = ray_loss; a.rgb = rgb; a.d_acc = d_acc; a.d_depth = d_depth;
  a.d_weights = d_weights; a.R = R; a.C = C; a.S = S; a.usealldepth = usealldepth; a.pad = rgb_padding;
  a.lambda_rgb = lambda_rgb; a.lambda_ds = lambda_ds;
  lambert_loss_kernel<<<dim3((unsigned)ceil_div64(R, WAVES_PER_BLOCK)), 64 * WAVES_PER_BLOCK, 0, (hipStream_t)stream>>>(a);
  BN_LAUNCH_CHECK("lambert_loss");
  return 0;
}

// ------------------------------------------------------------------------------------------ guided resampling
#define BN_MAX_G 256
#define BN_MAX_SG 512

struct GuidedArgs {
  const float *z, *weights, *depth, *u, *use_target, *target_depth, *target_std, *u_target;
  const int32_t *target_row;
  int64_t R;
  int S, G;
  float near0, far0, d_range;
  float *z2_sorted, *z_all;
  int64_t *sort_idx;
  const float *near_far;   // device [2] (near0, far0), e.g. &rays[0][6]; overrides the two scalars when set (no host read)
  // fused-step extensions (bn_composite_guided): element strides of the per-ray prior arrays (depths[:, 0] of an [R][2] table)
  // and in-kernel draws (u == nullptr: Philox streams rng_u / rng_ut of rng)
  int64_t td_stride, ts_stride, ut_stride;
  const unsigned long long *rng;
  unsigned int rng_u, rng_ut;
  int64_t ray_offset;      // in-kernel draws are indexed by ray_offset + ray: a sharded batch draws what the whole batch would
};

// in-LDS bitonic sort of n2 (power of two) (key, index) pairs by one wave; ties broken by index (= stable).
__device__ __forceinline__ void wave_bitonic(float *key, int *idx, int n2, int lane) {
  for (int k = 2; k <= n2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = lane; i < n2; i += 64) {
        const int p = i ^ j;
        if (p > i) {
          const float a = key[i], b = key[p];
          const int ia = idx[i], ib = idx[p];
          const bool up = (i & k) == 0;
          const bool gt = a > b || (a == b && ia > ib);
          if (gt == up) { key[i] = b; key[p] = a; idx[i] = ib; idx[p] = ia; }
        }
      }
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
    }
}

// One ray, one wave.  `wrow` = the ray's S pass-1 weights (global memory or the wave's LDS slice), `centre0` its pass-1 depth.
__device__ __forceinline__ void guided_ray(const GuidedArgs &A, int64_t ray, int lane, const float *z, const float *wrow,
                                           float centre0, float *edges, float *cdf, float *key, int *idx) {
  const int S = A.S, G = A.G;
  // 1. centre and spread (train_utils.py:35-39), or the ground-truth depth prior (rendering.py:135-145)
  float centre, std;
  const float *u = nullptr;
  unsigned int rstream = A.rng_u;
  int64_t urow = ray;
  if (A.use_target && A.use_target[ray * A.ut_stride] > 0.f) {
    centre = A.target_depth[ray * A.td_stride];
    std = A.target_std[ray * A.ts_stride];
    urow = A.target_row ? (int64_t)A.target_row[ray] : ray;      // no row table: u_target has one row per ray
    if (A.u_target) u = A.u_target + urow * G;
    rstream = A.rng_ut;
  } else {
    centre = centre0;
    float acc = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float dz = z[s] - centre;
      acc += dz * dz * wrow[s];
    }
    std = sqrtf(wave_sum(acc));
    if (A.u) u = A.u + ray * G;
  }
  // 2. symmetric 3-sigma window inside [near0, far0] (rendering.py:76-83)
  float lo = centre - A.d_range * std, hi = centre + A.d_range * std;
  const float near0 = A.near_far ? A.near_far[0] : A.near0, far0 = A.near_far ? A.near_far[1] : A.far0;
  lo = fminf(fmaxf(lo, near0), far0);
  hi = fminf(fmaxf(hi, near0), far0);
  const float rng = fminf(fabsf(hi - centre), fabsf(lo - centre));
  lo = centre - rng;
  hi = centre + rng;
  // 3. bin edges and Gaussian bin weights (rendering.py:63-69)
  const float step = (hi - lo) / (float)(G - 1);
  for (int j = lane; j < G; j += 64) {
    const float t = linspace_at(0.f, 1.f, G, j);
    edges[j] = lo * (1.f - t) + hi * t;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  float wsum = 0.f;
  for (int j = lane; j < G - 1; j += 64) {
    const float factor = (edges[j + 1] - edges[j]) / (step + 1e-5f);
    const float x = linspace_at(-A.d_range, A.d_range, G - 1, j);
    const float bw = factor * (0.3989422804014327f * expf(-0.5f * (x * x)));
    const float w = bw + 1e-5f;
    key[j] = w;
    wsum += w;
  }
  wsum = wave_sum(wsum);
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  // 4. cdf: sequential cumsum (rendering.py:27-29); the reference's CPU torch.cumsum accumulates fp32 inputs in
  //    double and rounds each prefix to fp32 - done the same way so that searchsorted indices agree.
  if (lane == 0) {
    double c = 0.0;
    cdf[0] = 0.f;
    for (int j = 0; j < G - 1; ++j) {
      c += (double)(key[j] / wsum);
      cdf[j + 1] = (float)c;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  // 5. inverse-CDF sampling, searchsorted(right=True) (rendering.py:39-51)
  const int n2g = G <= 64 ? 64 : (G <= 128 ? 128 : 256);
  for (int j = lane; j < n2g; j += 64) {
    float smp = INFINITY;
    if (j < G) {
      const float uu = u ? u[j] : philox_uniform(A.rng, rstream, (unsigned long long)((urow + A.ray_offset) * G + j));
      int lo_i = 0, hi_i = G;  // first index with cdf[i] > uu
      while (lo_i < hi_i) {
        const int mid = (lo_i + hi_i) >> 1;
        if (cdf[mid] > uu) hi_i = mid; else lo_i = mid + 1;
      }
      const int inds = lo_i;
      const int below = inds - 1 > 0 ? inds - 1 : 0;
      const int above = inds < G - 1 ? inds : G - 1;
      const float c0 = cdf[below], c1 = cdf[above];
      float denom = c1 - c0;
      if (denom < 1e-5f) denom = 1.f;
      smp = edges[below] + (uu - c0) / denom * (edges[above] - edges[below]);
    }
    key[j] = smp;
    idx[j] = j;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  wave_bitonic(key, idx, n2g, lane);
  float *z2 = A.z2_sorted + ray * G;
  for (int j = lane; j < G; j += 64) z2[j] = key[j];
  if (!A.z_all) return;
  // 6. merge with the coarse depths: stable sort of cat[z, z2] (rendering.py:271-272)
  const int N = S + G;
  int n2 = 64;
  while (n2 < N) n2 <<= 1;
  float my[BN_MAX_G / 64];
  for (int j = lane, q = 0; j < G; j += 64, ++q) my[q] = key[j];
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  for (int j = lane, q = 0; j < G; j += 64, ++q) key[S + j] = my[q];
  for (int j = lane; j < S; j += 64) key[j] = z[j];
  for (int j = lane; j < n2; j += 64) {
    idx[j] = j;
    if (j >= N) key[j] = INFINITY;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  wave_bitonic(key, idx, n2, lane);
  for (int j = lane; j < N; j += 64) {
    A.z_all[ray * N + j] = key[j];
    if (A.sort_idx) A.sort_idx[ray * N + j] = idx[j];
  }
}

__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void guided_kernel(const GuidedArgs A) {
  __shared__ float s_edges[WAVES_PER_BLOCK][BN_MAX_G];
  __shared__ float s_cdf[WAVES_PER_BLOCK][BN_MAX_G];
  __shared__ float s_key[WAVES_PER_BLOCK][BN_MAX_SG];
  __shared__ int s_idx[WAVES_PER_BLOCK][BN_MAX_SG];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wv;
  if (ray >= A.R) return;
  guided_ray(A, ray, lane, A.z + ray * A.S, A.weights + ray * A.S, A.depth[ray], s_edges[wv], s_cdf[wv], s_key[wv], s_idx[wv]);
}

// Pass-1 compositing (cal_weight, sigma = channel 3 of the pass-1 field output, no noise) + depth-guided resampling + merge in
// ONE launch, one wave per ray: the pass-1 weights never leave the CU (LDS slice of the wave).  Fused training step only.
struct CompGuidedArgs {
  GuidedArgs g;
  const float *sigma;        // [R][S] with element stride sigma_stride
  int64_t sigma_stride;
  float *weights, *depth;    // optional copies of the pass-1 weights / depth (nullable)
  NoiseArgs noise;           // --noise_std on the pass-1 compositing
};
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void composite_guided_kernel(const CompGuidedArgs A) {
  __shared__ float s_edges[WAVES_PER_BLOCK][BN_MAX_G];
  __shared__ float s_cdf[WAVES_PER_BLOCK][BN_MAX_G];
  __shared__ float s_key[WAVES_PER_BLOCK][BN_MAX_SG];
  __shared__ int s_idx[WAVES_PER_BLOCK][BN_MAX_SG];
  __shared__ float s_w[WAVES_PER_BLOCK][64 * BN_MAX_CPL];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wv;
  if (ray >= A.g.R) return;
  const int S = A.g.S, cpl = (S + 63) / 64;
  const float *z = A.g.z + ray * S;
  float zv[BN_MAX_CPL], al[BN_MAX_CPL], u[BN_MAX_CPL];
  float lp = 1.f;
#pragma unroll
  for (int j = 0; j < BN_MAX_CPL; ++j) {
    zv[j] = 0.f; al[j] = 0.f; u[j] = 1.f;
    const int s = lane * cpl + j;
    if (j < cpl && s < S) {
      zv[j] = z[s];
      const float delta = s == S - 1 ? 1e10f : z[s + 1] - zv[j];
      const float sg = noised(A.noise, A.sigma[(ray * S + s) * A.sigma_stride], ray, S, s);
      const float rs = sg > 0.f ? sg : 0.f;
      al[j] = 1.f - expf(-delta * rs);
      u[j] = 1.f - al[j] + 1e-10f;
      lp *= u[j];
    }
  }
  float T = wave_excl_prod(lp, lane);
  float dsum = 0.f;
  float *wl = s_w[wv];
#pragma unroll
  for (int j = 0; j < BN_MAX_CPL; ++j) {
    const int s = lane * cpl + j;
    if (j < cpl && s < S) {
      const float w = al[j] * T;
      wl[s] = w;
      if (A.weights) A.weights[ray * S + s] = w;
      dsum += w * zv[j];
    }
    T *= u[j];
  }
  dsum = wave_sum(dsum);
  if (lane == 0 && A.depth) A.depth[ray] = dsum;
  lds_wave_sync();
  guided_ray(A.g, ray, lane, z, wl, dsum, s_edges[wv], s_cdf[wv], s_key[wv], s_idx[wv]);
}

static int guided_samples_impl(const float *z, const float *weights, const float *depth, const float *u, int64_t R,
                               int32_t S, int32_t G, float near0, float far0, const float *near_far, float d_range,
                               const float *use_target, const float *target_depth, const float *target_std,
                               const float *u_target, const int32_t *target_row, float *z2_sorted, float *z_all,
                               int64_t *sort_idx, void *stream) {
  BN_REQUIRE(z && weights && depth && u && z2_sorted && R > 0, "guided_samples: null argument");
  BN_REQUIRE(G >= 3 && G <= BN_MAX_G && S >= 1 && S + G <= BN_MAX_SG, "guided_samples: S=%d G=%d unsupported", S, G);
  BN_REQUIRE(!use_target || (target_depth && target_std && u_target), "guided_samples: target arrays");
  GuidedArgs a = {z, weights, depth, u, use_target, target_depth, target_std, u_target, target_row, R, S, G,
                  near0, far0, d_range, z2_sorted, z_all, sort_idx, near_far, 1, 1, 1, nullptr, 0u, 0u, 0};
  BnProfScope prof_(BN_K_GUIDED, (hipStream_t)stream);
  guided_kernel<<<dim3((unsigned)ceil_div64(R, WAVES_PER_BLOCK)), 64 * WAVES_PER_BLOCK, 0, (hipStream_t)stream>>>(a);
  BN_LAUNCH_CHECK("guided_samples");
  return 0;
}

extern "C" int bn_guided_samples(const float *z, const float *weights, const float *depth, const float *u, int64_t R,
                                 int32_t S, int32_t G, float near0, float far0, float d_range, const float *use_target,
                                 const float *target_depth, const float *target_std, const float *u_target,
                                 const int32_t *target_row, float *z2_sorted, float *z_all, int64_t *sort_idx,
                                 void *stream) {
  return guided_samples_impl(z, weights, depth, u, R, S, G, near0, far0, nullptr, d_range, use_target, target_depth, target_std,
                             u_target, target_row, z2_sorted, z_all, sort_idx, stream);
}

extern "C" int bn_guided_samples_nf(const float *z, const float *weights, const float *depth, const float *u, int64_t R,
                                    int32_t S, int32_t G, const float *near_far, float d_range, const float *use_target,
                                    const float *target_depth, const float *target_std, const float *u_target,
                                    const int32_t *target_row, float *z2_sorted, float *z_all, int64_t *sort_idx,
                                    void *stream) {
  BN_REQUIRE(near_far, "guided_samples_nf: near_far is null");
  return guided_samples_impl(z, weights, depth, u, R, S, G, 0.f, 0.f, near_far, d_range, use_target, target_depth, target_std,
                             u_target, target_row, z2_sorted, z_all, sort_idx, stream);
}

extern "C" int bn_composite_guided(const float *z, const float *sigma, int64_t sigma_stride, int64_t R, int32_t S, int32_t G,
                                   const float *near_far, float d_range, const float *use_target, int64_t ut_stride,
                                   const float *target_depth, int64_t td_stride, const float *target_std, int64_t ts_stride,
                                   const float *u, const float *u_target, const unsigned long long *rng, uint32_t rng_u,
                                   uint32_t rng_ut, int64_t ray_offset, float *z2_sorted, float *z_all, int64_t *sort_idx, float *weights,
                                   float *depth, const bn_noise *noise, void *stream) {
  BN_REQUIRE(z && sigma && near_far && z2_sorted && R > 0, "composite_guided: null argument");
  BN_REQUIRE(G >= 3 && G <= BN_MAX_G && S >= 1 && S <= 64 * BN_MAX_CPL && S + G <= BN_MAX_SG, "composite_guided: S=%d G=%d unsupported", S, G);
  BN_REQUIRE(!use_target || (target_depth && target_std && (u_target || rng)), "composite_guided: target arrays");
  BN_REQUIRE(u || rng, "composite_guided: neither draws nor an rng state");
  CompGuidedArgs a;
  a.g = GuidedArgs{z, nullptr, nullptr, u, use_target, target_depth, target_std, u_target, nullptr, R, S, G,
                   0.f, 0.f, d_range, z2_sorted, z_all, sort_idx, near_far, td_stride, ts_stride, ut_stride, rng, rng_u, rng_ut, ray_offset};
  a.sigma = sigma; a.sigma_stride = sigma_stride; a.weights = weights; a.depth = depth; a.noise = make_noise(noise);
  BnProfScope prof_(BN_K_GUIDED, (hipStream_t)stream);
  composite_guided_kernel<<<dim3((unsigned)ceil_div64(R, WAVES_PER_BLOCK)), 64 * WAVES_PER_BLOCK, 0, (hipStream_t)stream>>>(a);
  BN_LAUNCH_CHECK("composite_guided");
  return 0;
}

// ------------------------------------------------------------------------------------------ merged-set compositing
// Pass 2 of the fused training step composites the depth-sorted union of the S1 coarse samples (field output out1, evaluated
// once in pass 1 and reused) and the G guided samples (out2): sample s of ray r is row idx[r][s] of cat[out1[r], out2[r]]
// (sort_idx of the merge, rendering.py:271-272).  These kernels read the rows through that index - no cat / gather copy of
// the field outputs - and the backward writes each sample's gradient row straight to d_out1 / d_out2 (no scatter / split).
// One wave per ray; lane l owns samples [l cpl, (l+1) cpl).  MODE 0: forward (alphas, transparency, weights, depth, acc).
// MODE 1: Lambertian tail - forward, shading + SNerfLoss + DepthLoss (bn_lambert_loss) and the backward of all of it in one
// pass over the ray.  MODE 2: backward from d_weights / d_depth / d_acc.
struct MergedArgs {
  const float *z;
  const int64_t *idx;           // nullptr: identity (a single source block out1 with S1 = S2)
  const float *out1, *out2;
  int S1, S2, C;
  int64_t R;
  float *alphas, *trans, *weights, *depth, *acc, *wsum, *var;
  // MODE 1
  const float *rgbs, *valid, *tdepth, *tweight, *tstd;
  int64_t v_stride, td_stride, tw_stride, ts_stride;
  float pad, lambda_rgb, lambda_ds;
  int usealldepth;
  float *ray_loss, *rgb, *loss_acc;
  int loss_slots;                  // the rays' loss terms are added to loss_acc[ray % loss_slots] (spread: same-address atomics serialise)
  // MODE 2
  const float *d_weights, *d_depth, *d_acc, *d_wsum, *depth_in;
  float hs_scale;
  // MODE 0, 2: NormalRegLoss (metrics.py:179-216) on the per-sample normals: lambda * sum_s w_s min(0, n_s . view)^2, view = -rays_d
  bn_normal_reg nreg;              // rays_d == nullptr: off
  float *reg_out;                  // MODE 0: the ray's regulariser term (both normal fields, lambdas applied)
  // MODE 1, 2
  float *d_out1, *d_out2;
  unsigned long long *nonfinite;   // nullable: zero non-finite gradient elements and count them ([0] NaN, [1] Inf)
  NoiseArgs noise;                 // --noise_std on the merged set (position = sorted position)
};

// C4 (C == 4 and 16-byte aligned blocks: the Lambertian model): a sample's row is ONE 16-byte load kept in registers for the
// forward sums and the backward dot products, and one 16-byte store of its gradient row.
template <int MODE, bool C4> __global__ __launch_bounds__(64 * WAVES_PER_BLOCK) void merged_composite_kernel(const MergedArgs A) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wv;
  if (ray >= A.R) return;
  const int S = A.S2, C = A.C, cpl = (S + 63) / 64, S1 = A.S1, Sg = S - S1;
  const float *z = A.z + ray * S;
  float zv[BN_MAX_CPL], al[BN_MAX_CPL], u[BN_MAX_CPL], dad[BN_MAX_CPL];
  int64_t roff[BN_MAX_CPL];      // element offset of the sample's row; >= 0: in out1, < 0: ~offset in out2
  f32x4 rv[C4 ? BN_MAX_CPL : 1];
  float lp = 1.f;
  auto rowp = [&](int64_t o) { return o >= 0 ? A.out1 + o : A.out2 + ~o; };
  auto chan = [&](int j, int c) { return C4 ? rv[C4 ? j : 0][c] : rowp(roff[j])[c]; };
#pragma unroll
  for (int j = 0; j < BN_MAX_CPL; ++j) {
    zv[j] = 0.f; al[j] = 0.f; u[j] = 1.f; dad[j] = 0.f; roff[j] = 0;
    const int s = lane * cpl + j;
    if (j < cpl && s < S) {
      const int64_t i = A.idx ? A.idx[ray * S + s] : (int64_t)s;
      roff[j] = i < S1 ? (ray * S1 + i) * C : ~((ray * Sg + (i - S1)) * C);
      zv[j] = z[s];
      const float delta = s == S - 1 ? 1e10f : z[s + 1] - zv[j];
      if (C4) rv[C4 ? j : 0] = *(const f32x4 *)rowp(roff[j]);
      const float sg = noised(A.noise, chan(j, 3), ray, S, s);
      const float rs = sg > 0.f ? sg : 0.f;
      const float e = expf(-delta * rs);
      al[j] = 1.f - e;
      u[j] = 1.f - al[j] + 1e-10f;
      dad[j] = sg > 0.f ? delta * e : 0.f;
      lp *= u[j];
    } else if (C4) {
      rv[C4 ? j : 0] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  float T = wave_excl_prod(lp, lane);
  float tr[BN_MAX_CPL], w[BN_MAX_CPL];
#pragma unroll
  for (int j = 0; j < BN_MAX_CPL; ++j) {
    tr[j] = T;
    w[j] = al[j] * T;
    T *= u[j];
  }
  float dd = 0.f, dws = 0.f;         // d loss / d depth, d loss / d (sum_s w_s)
  float dacc[BN_MAX_CH];             // MODE 1: channels 0..2 only
  if (MODE != 2) {
    float dsum = 0.f, ws = 0.f;
#pragma unroll
    for (int j = 0; j < BN_MAX_CPL; ++j) {
      const int s = lane * cpl + j;
      if (j < cpl && s < S) {
        const int64_t o = ray * S + s;
        if (A.alphas) A.alphas[o] = al[j];
        if (A.trans) A.trans[o] = tr[j];
        if (A.weights) A.weights[o] = w[j];
        dsum += w[j] * zv[j];
        ws += w[j];
      }
    }
    dsum = wave_sum(dsum);
    ws = wave_sum(ws);
    if (lane == 0 && A.depth) A.depth[ray] = dsum;
    if (lane == 0 && A.wsum) A.wsum[ray] = ws;
    const int nacc = MODE == 1 ? 3 : C;
    float a3[3] = {0.f, 0.f, 0.f};
    if (MODE == 1 || A.acc) {
      for (int c = 0; c < nacc; ++c) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < BN_MAX_CPL; ++j) {
          const int s = lane * cpl + j;
          if (j < cpl && s < S) a += w[j] * chan(j, c);
        }
        a = wave_sum(a);
        if (lane == 0 && A.acc) A.acc[ray * C + c] = a;
        if (MODE == 1 && c < 3) a3[c] = a;
      }
      if (MODE == 1 && A.acc && lane == 0) A.acc[ray * C + 3] = 0.f;
    }
    if (MODE == 0 && A.nreg.rays_d && A.reg_out) {
      const float *rd = A.nreg.rays_d + ray * A.nreg.rd_stride;
      const float vx = -rd[0], vy = -rd[1], vz = -rd[2];
      float reg = 0.f;
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int c0 = f == 0 ? A.nreg.ch_an : A.nreg.ch_lr;
        const float lam = f == 0 ? A.nreg.lambda_an : A.nreg.lambda_lr;
        if (c0 < 0 || !(lam > 0.f)) continue;
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < BN_MAX_CPL; ++j) {
          const int s = lane * cpl + j;
          if (j < cpl && s < S) {
            const float *row = rowp(roff[j]);
            const float nv = fminf(row[c0] * vx + row[c0 + 1] * vy + row[c0 + 2] * vz, 0.f);
            part += w[j] * nv * nv;
          }
        }
        reg += lam * part;
      }
      reg = wave_sum(reg);
      if (lane == 0) A.reg_out[ray] = reg;
    }
    if (MODE == 0 && A.nreg.lambda_spv != 0.f && A.nreg.spv_ray) {     // NormalLoss 'an_lr': this ray's (sum w, sum |n_an - n_lr|)
      const int ca = A.nreg.spv_ch_an, cl = A.nreg.spv_ch_lr;
      float sw = 0.f, sd = 0.f;
#pragma unroll
      for (int j = 0; j < BN_MAX_CPL; ++j) {
        const int s = lane * cpl + j;
        if (j < cpl && s < S) {
          const float *row = rowp(roff[j]);
          sw += w[j];
          sd += fabsf(row[ca] - row[cl]) + fabsf(row[ca + 1] - row[cl + 1]) + fabsf(row[ca + 2] - row[cl + 2]);
        }
      }
      sw = wave_sum(sw); sd = wave_sum(sd);
      if (lane == 0) { A.nreg.spv_ray[ray * 2] = sw; A.nreg.spv_ray[ray * 2 + 1] = sd; }
    }
    float var = 0.f;
    if (MODE == 1 || A.var) {
#pragma unroll
      for (int j = 0; j < BN_MAX_CPL; ++j) {
        const int s = lane * cpl + j;
        if (j < cpl && s < S) { const float dz = zv[j] - dsum; var += dz * dz * w[j]; }
      }
      var = wave_sum(var);
      if (lane == 0 && A.var) A.var[ray] = var;
    }
    if (MODE == 0) return;
    // ---- Lambertian shading + SNerfLoss + DepthLoss and their gradients (lambert_loss_kernel)
    const float invn = 1.f / (3.f * (float)A.R);
    float loss = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = a3[c] * (1.f + 2.f * A.pad) - A.pad * ws;
      const float y = isnan(x) ? x : fminf(fmaxf(x, 0.f), 1.f);     // a NaN colour stays a NaN (fminf / fmaxf alone would return 0)
      const float e = y - A.rgbs[ray * 3 + c];
      loss += A.lambda_rgb * e * e * invn;
      const float dy = (x >= 0.f && x <= 1.f) ? A.lambda_rgb * 2.f * e * invn : (isnan(x) ? x : 0.f);
      dacc[c] = dy * (1.f + 2.f * A.pad);
      dws -= dy * A.pad;
      if (lane == 0 && A.rgb) A.rgb[ray * 3 + c] = y;
    }
    if (A.tdepth && A.valid[ray * A.v_stride] > 0.f) {
      const float td = A.tdepth[ray * A.td_stride], tw = A.tweight[ray * A.tw_stride], ts = A.tstd[ray * A.ts_stride];
      const bool apply = A.usealldepth || (fabsf(dsum - td) - ts > 0.f) || (ts < sqrtf(var));
      if (apply) {
        const float k = A.lambda_ds / 3.f / (float)A.R;
        loss += k * tw * (dsum - td) * (dsum - td);
        dd = k * 2.f * tw * (dsum - td);
      }
    }
    // a ray whose loss term is not finite: with `nonfinite` (FusedTrainer.sanitize_grads) it is left out of the step - loss term 0,
    // gradients 0 - and counted, as bn_ray_shade_loss does; without it the NaN reaches the loss and the gradients as upstream
    if (A.nonfinite && !(fabsf(loss) <= 3.0e38f)) {
      if (lane == 0) atomicAdd(A.nonfinite + (isnan(loss) ? 0 : 1), 1ull);
      loss = 0.f; dd = 0.f; dws = 0.f;
      dacc[0] = dacc[1] = dacc[2] = 0.f;
    }
    if (lane == 0) {
      if (A.ray_loss) A.ray_loss[ray] = loss;
      if (A.loss_acc) atomicAdd(A.loss_acc + (int)(ray % A.loss_slots), loss);
    }
  } else {
    dd = A.d_depth ? A.d_depth[ray] : 0.f;
    dws = A.d_wsum ? A.d_wsum[ray] : 0.f;
    for (int c = 0; c < C; ++c) dacc[c] = (A.d_acc && c != 3) ? A.d_acc[ray * C + c] : 0.f;
  }
  // ---- backward: g_s = dL/dw_s; dL/dalpha_s = g_s T_s - (1/u_s) sum_{k>s} g_k w_k   
  const int ng = MODE == 1 ? 3 : C;
  const float hs = MODE == 2 ? A.hs_scale : 0.f, hs_depth = (MODE == 2 && A.hs_scale != 0.f) ? A.depth_in[ray] : 0.f;
  const bool nreg_on = MODE == 2 && A.nreg.rays_d != nullptr;
  float nvx = 0.f, nvy = 0.f, nvz = 0.f;
  if (nreg_on) { const float *rd = A.nreg.rays_d + ray * A.nreg.rd_stride; nvx = -rd[0]; nvy = -rd[1]; nvz = -rd[2]; }
  const bool spv_on = MODE == 2 && A.nreg.lambda_spv != 0.f && A.nreg.spv_tot != nullptr;
  const float spv_w = spv_on ? A.nreg.spv_tot[0] : 0.f, spv_n = spv_on ? A.nreg.spv_tot[1] : 0.f;
  float g[BN_MAX_CPL], gw = 0.f;
#pragma unroll
  for (int j = 0; j < BN_MAX_CPL; ++j) {
    g[j] = 0.f;
    const int s = lane * cpl + j;
    if (j < cpl && s < S) {
      float gg = dd * zv[j] + dws;
      if (MODE == 2 && A.d_weights) gg += A.d_weights[ray * S + s];
      if (MODE == 2 && hs != 0.f) { const float dz = zv[j] - hs_depth; gg += hs * (dz * dz); }
      if (MODE == 2 && spv_on) gg += spv_w;
      if (MODE == 2 && nreg_on) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int c0 = f == 0 ? A.nreg.ch_an : A.nreg.ch_lr;
          const float lam = f == 0 ? A.nreg.lambda_an : A.nreg.lambda_lr;
          if (c0 < 0 || !(lam > 0.f)) continue;
          const float nv = fminf(chan(j, c0) * nvx + chan(j, c0 + 1) * nvy + chan(j, c0 + 2) * nvz, 0.f);
          gg += lam * nv * nv;
        }
      }
      for (int c = 0; c < ng; ++c)
        if (c != 3) gg += dacc[c] * chan(j, c);
      g[j] = gg;
      gw += gg * w[j];
    }
  }
  float suffix = wave_excl_sum_rev(gw, lane);
  unsigned int n_nan = 0, n_inf = 0;
#pragma unroll
  for (int j = BN_MAX_CPL - 1; j >= 0; --j) {
    const int s = lane * cpl + j;
    if (j < cpl && s < S) {
      const float dalpha = g[j] * tr[j] - suffix / u[j];
      suffix += g[j] * w[j];
      float *drow = roff[j] >= 0 ? A.d_out1 + roff[j] : A.d_out2 + ~roff[j];
      if (C4) {
        f32x4 o = {w[j] * dacc[0], w[j] * dacc[1], w[j] * dacc[2], dalpha * dad[j]};
        if (A.nonfinite) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (isnan(o[c])) { ++n_nan; o[c] = 0.f; }
            else if (isinf(o[c])) { ++n_inf; o[c] = 0.f; }
          }
        }
        *(f32x4 *)drow = o;
      } else {
        float nr_g[2] = {0.f, 0.f};        // 2 lambda w_s min(0, n_s . view): times view[c - c0] on the normal's channels
        if (nreg_on) {
#pragma unroll
          for (int f = 0; f < 2; ++f) {
            const int c0 = f == 0 ? A.nreg.ch_an : A.nreg.ch_lr;
            const float lam = f == 0 ? A.nreg.lambda_an : A.nreg.lambda_lr;
            if (c0 < 0 || !(lam > 0.f)) continue;
            const float nv = fminf(chan(j, c0) * nvx + chan(j, c0 + 1) * nvy + chan(j, c0 + 2) * nvz, 0.f);
            nr_g[f] = 2.f * lam * w[j] * nv;
          }
        }
        for (int c = 0; c < C; ++c) {
          float v = c == 3 ? dalpha * dad[j] : (c < ng ? w[j] * dacc[c] : 0.f);
          if (nreg_on) {
            const int ca = c - A.nreg.ch_an, cl = c - A.nreg.ch_lr;
            if (A.nreg.ch_an >= 0 && ca >= 0 && ca < 3) v += nr_g[0] * (ca == 0 ? nvx : ca == 1 ? nvy : nvz);
            if (A.nreg.ch_lr >= 0 && cl >= 0 && cl < 3) v += nr_g[1] * (cl == 0 ? nvx : cl == 1 ? nvy : nvz);
          }
          if (spv_on) {     // d |n_an - n_lr| / d n: the sign of the difference (0 at 0, as torch's L1), opposite on the two fields
            const int ka = c - A.nreg.spv_ch_an, kl = c - A.nreg.spv_ch_lr;
            if (ka >= 0 && ka < 3) { const float d = chan(j, c) - chan(j, A.nreg.spv_ch_lr + ka); v += spv_n * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)); }
            if (kl >= 0 && kl < 3) { const float d = chan(j, A.nreg.spv_ch_an + kl) - chan(j, c); v -= spv_n * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)); }
          }
          if (A.nonfinite) {
            if (isnan(v)) { ++n_nan; v = 0.f; }
            else if (isinf(v)) { ++n_inf; v = 0.f; }
          }
          drow[c] = v;
        }
      }
    }
  }
  if (A.nonfinite) {
    n_nan = (unsigned int)wave_sum((float)n_nan);      // counts <= 64 * 8 * 32 per wave: exact in fp32
    n_inf = (unsigned int)wave_sum((float)n_inf);
    if (lane == 0 && n_nan) atomicAdd(A.nonfinite, (unsigned long long)n_nan);
    if (lane == 0 && n_inf) atomicAdd(A.nonfinite + 1, (unsigned long long)n_inf);
  }
}

// NormalLoss 'an_lr' (metrics.py:218-261): the two batch-wide means from the rays' sums, added up in a fixed order (one workgroup:
// thread t takes rays t, t + 1024, ... in order, then a binary tree over the threads) - bitwise reproducible like the rest of the step.
__global__ __launch_bounds__(1024) void normal_spv_reduce_kernel(const float *spv_ray, int64_t R, int S, float lambda, float *tot,
                                                                 float *ray_loss, float *loss_acc) {
  __shared__ float sw[1024], sd[1024];
  float a = 0.f, b = 0.f;
  for (int64_t r = threadIdx.x; r < R; r += 1024) { a += spv_ray[r * 2]; b += spv_ray[r * 2 + 1]; }
  sw[threadIdx.x] = a; sd[threadIdx.x] = b;
  __syncthreads();
  for (int o = 512; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) { sw[threadIdx.x] += sw[threadIdx.x + o]; sd[threadIdx.x] += sd[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float n = (float)R * (float)S;
    const float mean_w = sw[0] / n, mean_d = sd[0] / (3.f * n);
    const float term = lambda * mean_w * mean_d;
    tot[0] = lambda * mean_d / n;              // d loss / d w_s
    tot[1] = lambda * mean_w / (3.f * n);      // d loss / d n_an[c] per unit sign(n_an - n_lr)
    tot[2] = term; tot[3] = 0.f;
    if (ray_loss) ray_loss[0] += term;
    if (loss_acc) atomicAdd(loss_acc, term);
  }
}
extern "C" int bn_normal_spv_reduce(const float *spv_ray, int64_t R, int32_t S, float lambda_spv, float *spv_tot, float *ray_loss,
                                    float *loss_acc, void *stream) {
  BN_REQUIRE(spv_ray && spv_tot && R > 0 && S > 0, "normal_spv_reduce: bad arguments");
  normal_spv_reduce_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(spv_ray, R, S, lambda_spv, spv_tot, ray_loss, loss_acc);
  BN_LAUNCH_CHECK("normal_spv_reduce");
  return 0;
}

static int merged_check(const MergedArgs &a, const char *what) {
  BN_REQUIRE(a.z && a.out1 && a.R > 0 && a.S2 >= 1 && a.S2 <= 64 * BN_MAX_CPL && a.S1 >= 1 && a.S1 <= a.S2, "%s: bad arguments (S1=%d S2=%d)", what, a.S1, a.S2);
  BN_REQUIRE(a.C >= 4 && a.C <= BN_MAX_CH, "%s: C=%d unsupported", what, a.C);
  BN_REQUIRE(a.S1 == a.S2 || (a.idx && a.out2), "%s: a merged set needs sort_idx and the second block", what);
  return 0;
}
#define MERGED_GRID(R) dim3((unsigned)ceil_div64(R, WAVES_PER_BLOCK)), 64 * WAVES_PER_BLOCK, 0, (hipStream_t)stream
static bool merged_c4(const MergedArgs &a) {
  return a.C == 4 && ((uintptr_t)a.out1 | (uintptr_t)a.out2 | (uintptr_t)a.d_out1 | (uintptr_t)a.d_out2) % 16 == 0;
}

extern "C" int bn_merged_composite_forward(const float *z, const int64_t *sort_idx, const float *out1, const float *out2, int32_t S1,
                                           int32_t S2, int32_t C, int64_t R, float *alphas, float *trans, float *weights, float *depth,
                                           float *acc, float *wsum, float *var, const bn_normal_reg *nreg, float *reg_out,
                                           const bn_noise *noise, void *stream) {
  MergedArgs a = {};
  a.noise = make_noise(noise);
  a.z = z; a.idx = sort_idx; a.out1 = out1; a.out2 = out2; a.S1 = S1; a.S2 = S2; a.C = C; a.R = R;
  a.alphas = alphas; a.trans = trans; a.weights = weights; a.depth = depth; a.acc = acc; a.wsum = wsum; a.var = var;
  if (nreg && nreg->lambda_spv != 0.f)
    BN_REQUIRE(nreg->spv_ray && nreg->spv_ch_an >= 4 && nreg->spv_ch_an + 3 <= C && nreg->spv_ch_lr >= 4 && nreg->spv_ch_lr + 3 <= C,
               "merged_composite_forward: bad NormalLoss arguments (channels %d, %d)", nreg->spv_ch_an, nreg->spv_ch_lr);
  if (nreg && nreg->rays_d)
    BN_REQUIRE(reg_out && (nreg->ch_an < 0 || (nreg->ch_an >= 4 && nreg->ch_an + 3 <= C)) && (nreg->ch_lr < 0 || (nreg->ch_lr >= 4 && nreg->ch_lr + 3 <= C)),
               "merged_composite_forward: bad normal-regulariser channels (%d, %d)", nreg->ch_an, nreg->ch_lr);
  if (nreg && (nreg->rays_d || nreg->lambda_spv != 0.f)) { a.nreg = *nreg; a.reg_out = reg_out; }
  if (int e = merged_check(a, "merged_composite_forward")) return e;
  BnProfScope prof_(BN_K_COMPOSITE_FWD, (hipStream_t)stream);
  if (merged_c4(a)) merged_composite_kernel<0, true><<<MERGED_GRID(R)>>>(a);
  else merged_composite_kernel<0, false><<<MERGED_GRID(R)>>>(a);
  BN_LAUNCH_CHECK("merged_composite_forward");
  return 0;
}

extern "C" int bn_merged_composite_backward(const float *z, const int64_t *sort_idx, const float *out1, const float *out2, int32_t S1,
                                            int32_t S2, int32_t C, int64_t R, const float *d_weights, const float *d_depth,
                                            const float *d_acc, const float *d_wsum, float hs_scale, const float *depth,
                                            const bn_normal_reg *nreg, float *d_out1, float *d_out2, unsigned long long *nonfinite,
                                            const bn_noise *noise, void *stream) {
  MergedArgs a = {};
  a.noise = make_noise(noise);
  a.z = z; a.idx = sort_idx; a.out1 = out1; a.out2 = out2; a.S1 = S1; a.S2 = S2; a.C = C; a.R = R;
  a.d_weights = d_weights; a.d_depth = d_depth; a.d_acc = d_acc; a.d_wsum = d_wsum; a.d_out1 = d_out1; a.d_out2 = d_out2;
  a.nonfinite = nonfinite; a.hs_scale = hs_scale; a.depth_in = depth;
  if (nreg && nreg->lambda_spv != 0.f)
    BN_REQUIRE(nreg->spv_tot && nreg->spv_ch_an >= 4 && nreg->spv_ch_an + 3 <= C && nreg->spv_ch_lr >= 4 && nreg->spv_ch_lr + 3 <= C,
               "merged_composite_backward: bad NormalLoss arguments (channels %d, %d)", nreg->spv_ch_an, nreg->spv_ch_lr);
  if (nreg && nreg->rays_d)
    BN_REQUIRE((nreg->ch_an < 0 || (nreg->ch_an >= 4 && nreg->ch_an + 3 <= C)) && (nreg->ch_lr < 0 || (nreg->ch_lr >= 4 && nreg->ch_lr + 3 <= C)),
               "merged_composite_backward: bad normal-regulariser channels (%d, %d)", nreg->ch_an, nreg->ch_lr);
  if (nreg && (nreg->rays_d || nreg->lambda_spv != 0.f)) a.nreg = *nreg;
  if (int e = merged_check(a, "merged_composite_backward")) return e;
  BN_REQUIRE(hs_scale == 0.f || depth, "merged_composite_backward: hs_scale needs the forward's depth");
  BN_REQUIRE(d_out1 && (S1 == S2 || d_out2), "merged_composite_backward: null gradient buffer");
  BnProfScope prof_(BN_K_COMPOSITE_BWD, (hipStream_t)stream);
  if (merged_c4(a)) merged_composite_kernel<2, true><<<MERGED_GRID(R)>>>(a);
  else merged_composite_kernel<2, false><<<MERGED_GRID(R)>>>(a);
  BN_LAUNCH_CHECK("merged_composite_backward");
  return 0;
}

extern "C" int bn_lambert_tail(const float *z, const int64_t *sort_idx, const float *out1, const float *out2, int32_t S1, int32_t S2,
                               int32_t C, int64_t R, const float *rgbs, const float *valid_depth, int64_t v_stride,
                               const float *target_depth, int64_t td_stride, const float *target_weight, int64_t tw_stride,
                               const float *target_std, int64_t ts_stride, float rgb_padding, float lambda_rgb, float lambda_ds,
                               int32_t usealldepth, float *ray_loss, float *loss_acc, int32_t loss_slots, float *rgb,
                               float *weights, float *depth, float *d_out1, float *d_out2, unsigned long long *nonfinite,
                               const bn_noise *noise, void *stream) {
  MergedArgs a = {};
  a.noise = make_noise(noise);
  a.z = z; a.idx = sort_idx; a.out1 = out1; a.out2 = out2; a.S1 = S1; a.S2 = S2; a.C = C; a.R = R;
  a.rgbs = rgbs; a.valid = valid_depth; a.tdepth = target_depth; a.tweight = target_weight; a.tstd = target_std;
  a.v_stride = v_stride; a.td_stride = td_stride; a.tw_stride = tw_stride; a.ts_stride = ts_stride;
  a.pad = rgb_padding; a.lambda_rgb = lambda_rgb; a.lambda_ds = lambda_ds; a.usealldepth = usealldepth;
  a.ray_loss = ray_loss; a.loss_acc = loss_acc; a.loss_slots = loss_slots > 0 ? loss_slots : 1; a.rgb = rgb; a.weights = weights; a.depth = depth; a.d_out1 = d_out1; a.d_out2 = d_out2;
  a.nonfinite = nonfinite;
  if (int e = merged_check(a, "lambert_tail")) return e;
  BN_REQUIRE(rgbs && d_out1 && (S1 == S2 || d_out2), "lambert_tail: null argument");
  BN_REQUIRE(!target_depth || (valid_depth && target_weight && target_std), "lambert_tail: incomplete depth prior");
  BnProfScope prof_(BN_K_COMPOSITE_BWD, (hipStream_t)stream);
  if (merged_c4(a)) merged_composite_kernel<1, true><<<MERGED_GRID(R)>>>(a);
  else merged_composite_kernel<1, false><<<MERGED_GRID(R)>>>(a);
  BN_LAUNCH_CHECK("lambert_tail");
  return 0;
}
