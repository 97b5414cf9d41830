// Shared device/host helpers for the gfx950 kernels.  CDNA4 only (wave64, MFMA 32x32).
#pragma once
#include "diag.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <type_traits>
#include "brdfnerf_hip.h"
#include "prof.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

void bn_set_error(const char *fmt, ...);
// Sticky device fault words of the barrier-free trunks (field_fwd.hip / field_bwd.hip; bn_device_faults): device addresses on the
// current device, and the host's asynchronous view of them (bit 0 forward, bit 1 backward; refreshed every 64th forward launch
// outside stream captures).  bn_adam_multi reads both words: a step whose kernels lost a hand-over does not update the parameters.
const unsigned int *bn_fwd_fault_ptr();
const unsigned int *bn_bwd_fault_ptr();
int bn_field_fault_seen();
// Raise a kernel's dynamic-LDS limit to `lds` bytes on the CURRENT device (cached per (device, kernel); thread-safe).
int bn_configure_lds(const void *kernel, size_t lds, const char *what);

#define BN_REQUIRE(cond, ...)            \
  do {                                   \
    if (!(cond)) {                       \
      bn_set_error(__VA_ARGS__);         \
      return BN_EINVAL;                  \
    }                                    \
  } while (0)

#define BN_LAUNCH_CHECK(what)                                             \
  do {                                                                    \
    hipError_t e_ = hipGetLastError();                                    \
    if (e_ != hipSuccess) {                                               \
      bn_set_error("%s: launch failed: %s", what, hipGetErrorString(e_)); \
      return BN_ELAUNCH;                                                  \
    }                                                                     \
  } while (0)

#define BN_HIP_CHECK(call, what)                                          \
  do {                                                                    \
    hipError_t e_ = (call);                                               \
    if (e_ != hipSuccess) {                                               \
      bn_set_error("%s: %s", what, hipGetErrorString(e_));                \
      return BN_ELAUNCH;                                                  \
    }                                                                     \
  } while (0)

// bn_set_deterministic(): process-wide switch read by bn_field_backward (field_bwd.hip, det_enter)
int bn_deterministic();
#define BN_MAX_CH 32             // channels of a field output row: rgb3 + sigma + beta + two normals + three 3-wide BRDF heads = 20 at most today

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t bn_esize(int dtype) { return dtype == BN_F32 ? 4 : 2; }
static inline bool bn_half(int dtype) { return dtype != BN_F32; }   // 16-bit throughput modes (bf16, fp16)
// bytes per element of the D_l / DG stashes (field_kernels.h, "the derivative stash"): fp32 in the parity mode, fp16 in the fp16 mode of
// a model with analytic normals (round 5), 8-bit fixed point otherwise
static inline size_t bn_dsize(int dtype, int normal_an) { return dtype == BN_F32 ? 4 : ((dtype == BN_F16 && normal_an) ? 2 : 1); }
struct DK8 {};
struct DK16 {};
struct DK32 {};
template <typename T, bool D16> struct DKind { typedef DK8 type; };
template <bool D16> struct DKind<float, D16> { typedef DK32 type; };
template <> struct DKind<_Float16, true> { typedef DK16 type; };

// ---------------------------------------------------------------- MFMA element traits
// A "fragment" is 8 consecutive k-elements of one row (A) / column (B) held by lane (r = lane&31,
// h = lane>>5) for k = 8h + j.  bf16 / fp16: one v_mfma_f32_32x32x16_{bf16,f16}.  f32: eight
// v_mfma_f32_32x32x2_f32, MFMA j consuming element j of both fragments (k-permutation is
// consistent between A and B, so the sum over k is the same).
template <typename T> struct Elem;
template <> struct Elem<bf16> {
  typedef bf16x8 frag;
  typedef bf16x4 vec4;
  typedef bf16 wide;                // storage type of element-wise stashes that need fp32's exponent range
  // Y_l (the layer outputs the weight gradient multiplies by) is stashed in accumulator-native order, written straight from
  // the forward epilogue's registers: no row-major copy riding in the next GEMM.  The weight-gradient kernel stages native
  // chunks (field_bwd.hip, w2_body<.., BNAT>); the fp32 parity mode keeps the row-major stash.
  static constexpr bool kNativeY = true;
  // (D_l = d act / d z is stashed as 8-bit fixed point, DKind<T, D16> below / field_kernels.h: half the bytes of a 16-bit image)
  static constexpr int kBM = 128;   // points per workgroup tile
  static constexpr int kPad = 8;    // LDS row pad (elements) = 16 B
  static constexpr int kU = 2;      // k-steps per prefetch block
  static constexpr bool kFastMath = true;
};
// fp16 throughput mode (BN_F16): same tiling and MFMA rate as bf16, 3 more mantissa bits (Siren activations live in
// [-1, 1]); the backward chains run on loss-scaled gradients (field_bwd.hip, GradScale) because fp16 has bf16's
// precision problem the other way round: 5 exponent bits.
template <> struct Elem<f16> {
  typedef f16x8 frag;
  typedef f16x4 vec4;
  typedef bf16 wide;
  static constexpr bool kNativeY = true;
  static constexpr int kBM = 128;
  static constexpr int kPad = 8;
  static constexpr int kU = 2;
  static constexpr bool kFastMath = true;
};
template <> struct Elem<float> {
  typedef f32x8 frag;
  typedef f32x4 vec4;
  typedef float wide;
  static constexpr bool kNativeY = false;
  static constexpr int kBM = 64;
  static constexpr int kPad = 4;
  static constexpr int kU = 2;
  static constexpr bool kFastMath = false;
};

__device__ __forceinline__ void mma32(f32x16 &acc, const bf16x8 &a, const bf16x8 &b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16 &acc, const f16x8 &a, const f16x8 &b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16 &acc, const f32x8 &a, const f32x8 &b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}

__device__ __forceinline__ bf16x4 to_vec4(bf16, float a, float b, float c, float d) {
  bf16x4 v;
  v[0] = (bf16)a; v[1] = (bf16)b; v[2] = (bf16)c; v[3] = (bf16)d;
  return v;
}
__device__ __forceinline__ f16x4 to_vec4(f16, float a, float b, float c, float d) {
  const f32x4 v = {a, b, c, d};
  return __builtin_convertvector(v, f16x4);   // two v_cvt_pk_f16_f32 (round to nearest even)
}
__device__ __forceinline__ f32x4 to_vec4(float, float a, float b, float c, float d) {
  f32x4 v = {a, b, c, d};
  return v;
}

// sin/cos of the Siren / PE arguments.  Parity mode (fp32) uses the accurate libm forms; the bf16
// throughput mode uses the hardware v_sin/v_cos (abs. error ~1e-6 for |x| < 100, far below bf16).
// Compact accurate sincos for |x| < ~1e4 (PE arguments reach 2^9 * |xyz|): 3-term Cody-Waite reduction by pi/2
// with FMA (exact products), then the Cephes single-precision minimax polynomials on [-pi/4, pi/4]
// (max abs error 9.3e-8 over |x| < 1220, checked on the host).
__device__ __forceinline__ void sincos_cw(float x, float &s, float &c) {
  const float k = rintf(x * 0.63661977236758134f);
  float r = fmaf(-k, 1.57079637050628662109375f, x);
  r = fmaf(-k, -4.37113900018624283e-8f, r);
  r = fmaf(-k, -1.71512449591634e-15f, r);
  const float z = r * r;
  const float S = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float C = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                       fmaf(-0.5f, z, 1.0f));
  const int q = ((int)k) & 3;
  const float ss = (q & 1) ? C : S, cc = (q & 1) ? S : C;
  s = (q & 2) ? -ss : ss;
  c = ((q + 1) & 2) ? -cc : cc;
}

template <bool FAST> __device__ __forceinline__ void sincos_t(float x, float &s, float &c) {
  if (FAST) {
    s = __sinf(x);
    c = __cosf(x);
  } else {
    sincos_cw(x, s, c);
  }
}

// Loss scaling of the fp16 backward chains.  `amax` (device, nullable) holds max |seed gradient| as float bits
// (grad_amax_kernel, field_bwd.hip); the scale is the power of two that brings it into [target/2, target).  Powers of two
// commute exactly with every fp32 operation of the chains, so the unscaled fp32 gradients differ from an unscaled run
// only where fp16 would have under- or overflowed.  amax == nullptr (fp32 / bf16 modes): scale 1.
__device__ __forceinline__ float grad_scale_from(const float *amax, int log2_target) {
  if (amax == nullptr) return 1.f;
  const float a = *amax;
  if (!(a > 0.f) || !(a < 3.0e38f)) return 1.f;
  int e;
  (void)frexpf(a, &e);                       // a = m 2^e, m in [0.5, 1)
  int k = log2_target - e;
  k = k < -24 ? -24 : (k > 40 ? 40 : k);
  return ldexpf(1.f, k);
}
#define BN_GS_TARGET_CHAIN 8                 // primal backward chain: max |d pre-activation| -> [128, 256)
#define BN_GS_TARGET_ADJ 6                   // analytic-normal double backward: max |gbar_PE| -> [32, 64)
// amax slots (fp32 as bits, in the stash header): [0] primal seeds, [1] adjoint-chain seeds, [2] the zbar_l terms the
// adjoint backward adds to the primal chain at every layer (true scale; layer 0 carries w0^2 = 900).  The primal chain's
// scale must cover [0] and [2]; fmaxf ignores a NaN operand.
__device__ __forceinline__ float chain_scale(const float *amax) {
  if (amax == nullptr) return 1.f;
  const float m = fmaxf(amax[0], amax[2]);
  return grad_scale_from(&m, BN_GS_TARGET_CHAIN);
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// Register index i (0..15) of a 32x32 MFMA accumulator -> row offset inside the tile for lane half h:
// row = (i & 3) + 8 * (i >> 2) + 4 * h ;  column = lane & 31.
