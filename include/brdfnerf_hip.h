/*
 * brdfnerf_hip.h - C ABI of the MI355X (gfx950) implementation of BRDF-NeRF's ray-batched
 * volume-rendering hot path (spsbrdf-nerf).
 *
 * The reference (LulinZhang/BRDF-NeRF) has no FFI layer: its boundary is the Python signatures
 * of `SpSBRDFNeRF.forward`, `inference`, `cal_weight`, `render_rays`, the BRDF classes and the
 * guided-sampling helpers.  Each entry point below replaces the ATen op
 * sequence of one of those functions; the file:line it replaces is cited per function
 * (paths relative to the reference repo root).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator in
 *    the shipped host code); nothing is allocated, freed or synchronised inside a call;
 *  - `stream` is a hipStream_t passed as void*; all calls are asynchronous on that stream,
 *    re-entrant, and keep no global state (graph-capturable);
 *  - return value: 0 on success, a negative bn_status otherwise; bn_last_error() returns a
 *    thread-local description of the last failure.  Nothing throws across the boundary;
 *  - NaN handling follows the reference's check_nan(val_rep=...) replacement values
 *    (train_utils.py:61-78) in-kernel, without its prints and host syncs.
 *  - `dtype`: BN_F32 computes the MLP with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32; parity
 *    mode, 1e-4 relative vs the reference), BN_BF16 with bf16 MFMA (v_mfma_f32_32x32x16_bf16,
 *    fp32 accumulate; throughput mode), BN_F16 with fp16 MFMA (v_mfma_f32_32x32x16_f16, fp32
 *    accumulate; same rate as bf16, 3 more mantissa bits; the backward chains run on gradients
 *    scaled by a power of two chosen on the device from max|d_out|, removed again before the
 *    fp32 gradient accumulation - BASELINE config 5).  Everything outside the dense layers is fp32.
 */
#ifndef BRDFNERF_HIP_H
#define BRDFNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BN_ABI_VERSION 7
#define BN_MAX_LAYERS 12
#define BN_MAX_HEADS 6 /* rgb (+ beta) + up to 3 BRDF heads evaluated together, two per pass */

typedef enum { BN_OK = 0, BN_EINVAL = -1, BN_EUNSUPPORTED = -2, BN_ELAUNCH = -3 } bn_status;
typedef enum { BN_F32 = 0, BN_BF16 = 1, BN_F16 = 2 } bn_dtype;
typedef enum { BN_ACT_SIN = 0, BN_ACT_RELU = 1 } bn_act;
/* post-sigmoid rescale of a head (spsbrdfnerf.py:730,735,754) */
typedef enum {
  BN_HEAD_PLAIN = 0,       /* sigmoid, width = head_out (rgb, roughness)                        */
  BN_HEAD_RPV_K = 1,       /* (v-.5)*2+1, 1-wide tiled x3                                       */
  BN_HEAD_RPV_THETA = 2,   /* (v-.5)*2,   1-wide tiled x3                                       */
  BN_HEAD_HAPKE_THETA = 3, /* v*pi/6, width 1                                                   */
  BN_HEAD_TILE3 = 4,       /* sigmoid, 1-wide tiled x3 (rhoc, b, c)                             */
  BN_HEAD_BETA = 5         /* --beta (spsbrdfnerf.py:571-575,708-711): SOFTPLUS instead of sigmoid, width 1, written to
                              channel 4 (right after sigma, before the normals); only as head 1; its first layer also
                              reads the per-image embedding (desc.t_dim columns)                  */
} bn_head_kind;

int bn_abi_version(void);
const char *bn_last_error(void);
/* Diagnostic / A-B preprocessor switches this build of the library was compiled with, space-separated ("" for the release
 * build).  Timing builds under profiles/ change instruction streams or drop work; the test-suite asserts that the library
 * it validates reports none. */
const char *bn_build_flags(void);
/* First 16 hex digits of the sha256 over the sources the library was compiled from (csrc/, this header, the build recipe). */
const char *bn_source_hash(void);
/* The reference trains with Lightning's deterministic=True (main.py:726).  Since ABI 5 bn_field_backward's parameter gradients
 * are bitwise identical for identical inputs in EVERY mode: the weight-gradient kernels write the partial tile of every point
 * split to a slab of the stash and a reduce kernel adds the slabs in split order (no fp32 atomics, no turn counters).  The
 * switch remains for callers that keyed on it (the Python trainer reports a fixed-order loss sum with it); the library's kernels
 * no longer read it.  Process-wide; returns the old value. */
int bn_set_deterministic(int on);
/* The current setting (read-only: callers that only want to know must not toggle a process-wide switch to find out). */
int bn_get_deterministic(void);

/* ---------------------------------------------------------------------------------------------
 * Field MLP (SpSBRDFNeRF.forward, models/spsbrdfnerf.py:662-757; calc_features :636-646;
 * Mapping.forward models/nerf.py:53-70).
 *
 * Geometry: F = feat (multiple of 64, <= 512), `layers` trunk layers, skip-concat [PE, h] at
 * layer `skip` (-1: none), PE with `pe_freqs` octaves (0: raw xyz).  Heads: sigma (F->1,
 * softplus), feats (F->F, linear), then n_heads two-layer heads F -> F/2 (act) -> head_out[i]
 * (sigmoid), head 0 = rgb.  Optional learned normal (grad_from_xyz, F->3, -l2_normalize).
 *
 * Weights are consumed in a packed, MFMA-fragment-ordered copy produced by bn_pack_field();
 * biases and the small second-layer head matrices are read as fp32 from the caller's tensors.
 * ------------------------------------------------------------------------------------------- */
typedef struct bn_field_desc {
  int32_t feat, layers, skip, pe_freqs, act, dtype;
  int32_t n_heads;                     /* heads evaluated in this call (>=1, head 0 = rgb)    */
  int32_t head_out[BN_MAX_HEADS];      /* 1 or 3                                              */
  int32_t head_kind[BN_MAX_HEADS];     /* bn_head_kind                                        */
  int32_t normal_lr;                   /* 1: evaluate grad_from_xyz                            */
  int32_t normal_an;                   /* 1: analytic normal = -normalize(d sigma/d xyz)       */
  int32_t out_channels;                /* row length of `out`                                  */
  int32_t fold_feats;                  /* 1: the linear feats layer is folded into every head's first layer by the caller:
                                          params.head_w1[h] = <head>.0.weight * feats_from_xyz.weight  ([F/2][F]),
                                          params.head_b1[h] = <head>.0.weight * feats_from_xyz.bias + <head>.0.bias,
                                          feats_w / feats_b are ignored; bn_field_backward then returns dL/d(folded w1),
                                          dL/d(folded b1) in grads.head_w1/head_b1 and leaves grads.feats_* untouched
                                          (chain rule back to the three factors: brdf_nerf_amd/functions.py, unfold_grads).
                                          Same function, one F x F product less per point in forward, backward and
                                          weight gradient.                                           */
  int32_t dir_dim;                     /* --input_viewdir 1 (spsbrdfnerf.py:458,689-692): width of the (encoded) view direction the
                                          rgb head's first layer reads beside the features: 0 (off), 3 (raw, no --mapping) or
                                          6 * dir_freqs.  Needs fold_feats = 1: params.head_w1[0] is the folded [F/2][F] matrix and
                                          params.head0_wdir the direction columns of <rgb head>.0.weight                        */
  int32_t dir_freqs;                   /* octaves of the direction encoding (mapping_sizes[1] = 4), 0 = raw direction              */
  int32_t t_dim;                       /* --beta: width of the per-image embedding (--t_embbeding_tau, default 4) that head 1
                                          (kind BN_HEAD_BETA) reads beside the features; 0 without that head.  Needs fold_feats;
                                          params.head1_wt = beta_from_xyz.0.weight[:, F:].  Together with dir_dim:
                                          round_up(dir_dim, 8) + t_dim <= 32                                                      */
} bn_field_desc;

/* fp32 parameter tensors in PyTorch nn.Linear layout (weight [out][in], row-major). */
typedef struct bn_field_params {
  const float *trunk_w[BN_MAX_LAYERS]; /* fc_net.{2i}.weight                                   */
  const float *trunk_b[BN_MAX_LAYERS];
  const float *sigma_w, *sigma_b;      /* sigma_from_xyz.0                                     */
  const float *feats_w, *feats_b;      /* feats_from_xyz                                       */
  const float *head_w1[BN_MAX_HEADS], *head_b1[BN_MAX_HEADS]; /* <head>.0                      */
  const float *head_w2[BN_MAX_HEADS], *head_b2[BN_MAX_HEADS]; /* <head>.2                      */
  const float *normal_w, *normal_b;    /* grad_from_xyz (may be NULL)                          */
  const float *head0_wdir;             /* rgb_from_xyzdir.0.weight[:, F:]  ([F/2][dir_dim], row stride head0_wdir_ld); NULL when dir_dim == 0 */
  int64_t head0_wdir_ld;
  const float *head1_wt;               /* beta_from_xyz.0.weight[:, F:]  ([F/2][t_dim], row stride head1_wt_ld); NULL when t_dim == 0 */
  int64_t head1_wt_ld;
} bn_field_params;

/* Same shape as bn_field_params but writable: gradient accumulators (fp32, += semantics). */
typedef struct bn_field_grads {
  float *trunk_w[BN_MAX_LAYERS], *trunk_b[BN_MAX_LAYERS];
  float *sigma_w, *sigma_b, *feats_w, *feats_b;
  float *head_w1[BN_MAX_HEADS], *head_b1[BN_MAX_HEADS], *head_w2[BN_MAX_HEADS], *head_b2[BN_MAX_HEADS];
  float *normal_w, *normal_b;
  float *head0_wdir;                   /* += d/d rgb_from_xyzdir.0.weight[:, F:]  (row stride head0_wdir_ld) */
  int64_t head0_wdir_ld;
  float *head1_wt;                     /* += d/d beta_from_xyz.0.weight[:, F:]  (row stride head1_wt_ld) */
  int64_t head1_wt_ld;
  float *d_t_embed;                    /* nullable; = (overwritten) d/d t_embed per POINT, [n_points][t_dim], in both point forms
                                          (the rays form's per-ray gradient is its sum over the ray's samples)            */
} bn_field_grads;

/* Bytes of the packed weight buffer (forward + transposed copies) for `desc`. */
size_t bn_field_packed_bytes(const bn_field_desc *desc);
/* Re-pack after every optimizer step (a few MB; one launch per matrix). */
int bn_pack_field(const bn_field_desc *desc, const bn_field_params *params, void *packed, void *stream);

/* Bytes of the activation stash a training forward writes for `n_points` points. */
size_t bn_field_stash_bytes(const bn_field_desc *desc, int64_t n_points);

/* Points: either `xyz` [n_points][3], or rays (`rays` [n_rays][ray_stride] = o3,d3,near,far[,sun3];
 * `z` [n_rays][n_samples]; xyz = o + d*z, rendering.py:184) when xyz == NULL. */
typedef struct bn_points {
  const float *xyz;
  const float *rays;
  const float *z;
  int32_t ray_stride, n_samples;
  int64_t n_points;
  const float *dirs;                   /* xyz form with desc.dir_dim > 0: view direction per point [n_points][3] (the rays form reads
                                          rays[ray][3:6], as inference() repeats rays_d per sample, spsbrdfnerf.py:96,121)       */
  const float *t_embed;                /* desc.t_dim > 0: the per-image embedding models['t'](ts) (rendering.py:226-229) - xyz form:
                                          per point [n_points][t_dim]; rays form: per ray [n_rays][t_dim] (repeated per sample,
                                          spsbrdfnerf.py:98)                                                                       */
  /* One output array and ONE stash for several forward calls (round 3: pass 1 and pass 2 of the fused step are evaluated by
   * two launches - the second needs the first's result to place its samples - and back-propagated by ONE set of launches).
   * Forward / normals: this call's points are points [point_offset, point_offset + n_points) of a set of total_points points
   * (point_offset a multiple of the tile: 128 points in the 16-bit modes, 64 in fp32; `out` and `stash` are the SET's);
   * total_points == 0: a set of its own.  Backward over the whole set (rays form): points [0, seg1_points) are samples
   * z[r][0 .. n_samples) of ray r, points [seg1_points, n_points) are samples z2[r][0 .. n_samples2); seg1_points == 0: one block. */
  int64_t point_offset, total_points;
  const float *z2;
  int32_t n_samples2;
  int64_t seg1_points;
} bn_points;

/* sigma-only forward (forward(sigma_only=True), spsbrdfnerf.py:684): sigma[n_points]. */
int bn_field_sigma(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                   const bn_points *pts, float *sigma, void *stream);
/* full forward: out[n_points][desc->out_channels], channel order of spsbrdfnerf.py:694-755:
 * [rgb3, sigma, (beta1), (normal_an3), (normal_lr3), head outputs (1-wide RPV/Hapke b,c heads tiled x3)].
 * `stash` != NULL keeps what bn_field_backward needs. */
int bn_field_forward(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                     const bn_points *pts, float *out, void *stash, void *stream);
/* Analytic normals (calc_normals, models/spsbrdfnerf.py:648-660 and :713-716): fills channels [4,7) of `out` with
 * -l2_normalize(d sigma / d xyz) by the explicit adjoint chain over the stash of a preceding bn_field_forward() call on
 * the same points (desc->normal_an must be 1 in both).  grad_x (nullable) receives the raw gradient [n_points][3].
 * keep_for_backward = 1 also stashes what bn_field_backward needs to differentiate THROUGH the normals (the reference's
 * create_graph=True double backward). */
int bn_field_normals(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                     const bn_points *pts, const void *stash, float *out, float *grad_x, int32_t keep_for_backward,
                     void *stream);
/* Parameter gradients from d_out[n_points][out_channels] (autograd of forward, K9 in SURVEY.md).
 * `out` is the forward's output.  grads are accumulated (+=). */
int bn_field_backward(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                      const bn_points *pts, const float *out, const float *d_out, void *stash,
                      const bn_field_grads *grads, void *stream);

/* bn_field_backward in parts (a bit mask), for callers that overlap the gradient all-reduce with the rest of the backward
 * (the reference: Lightning DDP's bucketed all-reduce, main.py:720-731): BN_BWD_CHAIN = the dX chain(s) that fill the
 * stash with the pre-activation gradients; BN_BWD_WGRAD_TRUNK = the weight gradients of the trunk layers (the leading,
 * contiguous share of a flat parameter buffer in state_dict order); BN_BWD_WGRAD_HEADS = those of the (folded) head first
 * layers, the feats layer and the extra-input columns; BN_BWD_SKINNY = the one- to four-row matrices (sigma head, learned
 * normal, second head layers) and d_t_embed.  The parts of one backward are called in this order on one stream; the
 * deterministic mode takes BN_BWD_ALL only. */
enum { BN_BWD_CHAIN = 1, BN_BWD_WGRAD_TRUNK = 2, BN_BWD_WGRAD_HEADS = 4, BN_BWD_SKINNY = 8, BN_BWD_ALL = 15 };
int bn_field_backward_parts(const bn_field_desc *desc, const bn_field_params *params, const void *packed,
                            const bn_points *pts, const float *out, const float *d_out, void *stash,
                            const bn_field_grads *grads, int32_t parts, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Alpha compositing (cal_weight, models/spsbrdfnerf.py:50-69) fused with the per-ray weighted
 * sums of inference() (:198,:242,:271,:275,:292,:314-317,:326-338).
 * z, sigma: [R][S]; noise (nullable): [R][S] standard normals, used as sigma + noise*noise_std.
 * chan (nullable): [R][S][C] per-sample channels; acc: [R][C] = sum_s w*chan.
 * Outputs alphas, trans, weights [R][S] (nullable individually), depth [R].
 * ------------------------------------------------------------------------------------------- */
int bn_composite_forward(const float *z, const float *sigma, int64_t sigma_stride, const float *noise,
                         float noise_std, const float *chan, int64_t chan_stride, int32_t C, int64_t R, int32_t S,
                         float *alphas, float *trans, float *weights, float *depth, float *acc, void *stream);
/* Backward: given d_weights [R][S] (nullable), d_depth [R] (nullable), d_acc [R][C] (nullable),
 * writes d_sigma (strided like sigma) and d_chan (strided like chan, nullable). */
int bn_composite_backward(const float *z, const float *sigma, int64_t sigma_stride, const float *noise,
                          float noise_std, const float *chan, int64_t chan_stride, int32_t C, int64_t R, int32_t S,
                          const float *d_weights, const float *d_depth, const float *d_acc, float *d_sigma,
                          int64_t d_sigma_stride, float *d_chan, int64_t d_chan_stride, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Ray-level tail of a Lambertian training step in one launch: shading of the composited sums
 * (models/spsbrdfnerf.py:270-282, rgb = clamp(sum_s w (albedo (1+2p) - p), 0, 1)), SNerfLoss
 * (metrics.py:39-61, lambda_sc = 0) and DepthLoss (metrics.py:82-161, subset=True, GNLL=False; the
 * np.where row selection as a mask), with the gradients autograd would return for acc [R][C], depth [R] and
 * weights [R][S].  target_depth == NULL: no depth term.  ray_loss [R] sums to the step's loss.
 * ------------------------------------------------------------------------------------------- */
int bn_lambert_loss(const float *acc, int32_t C, const float *weights, const float *z, int32_t S, const float *depth,
                    const float *rgbs, const float *valid_depth, const float *target_depth, const float *target_weight,
                    const float *target_std, float rgb_padding, float lambda_rgb, float lambda_ds, int32_t usealldepth,
                    int64_t R, float *ray_loss, float *rgb, float *d_acc, float *d_depth, float *d_weights, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Stratified depths (get_z_vals, rendering.py:149-166, perturb=1): z[R][S] from near/far taken
 * from rays[:,6], rays[:,7] (or explicit near/far arrays) and uniforms u[R][S].
 * ------------------------------------------------------------------------------------------- */
int bn_stratified_z(const float *near, const float *far, int64_t nf_stride, const float *u, int64_t R, int32_t S,
                    float *z, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Depth-guided resampling + merge (GenerateGuidedSamples .. sample_pdf, rendering.py:13-91,
 * :116-147; merge :263-272).  Per ray: std = sqrt(sum w (z-depth)^2); 3-sigma window clamped to
 * [near0, far0] and symmetrised; G Gaussian-weighted bins; inverse-CDF with u[R][G]; sort;
 * then z_all[R][S+G] = sort(cat[z, z2]) with sort_idx int64 (bit-exact index semantics).
 * Rows with use_target[r] != 0 (train mode & valid depth) sample around target_depth/target_std
 * with u_target[target_row[r]][G] instead (rendering.py:135-145: the reference draws one row per VALID ray,
 * target_row[r] = rank of ray r among them; target_row == NULL: u_target has one row per ray, [R][G]).
 * ------------------------------------------------------------------------------------------- */
int bn_guided_samples(const float *z, const float *weights, const float *depth, const float *u, int64_t R,
                      int32_t S, int32_t G, float near0, float far0, float d_range, const float *use_target,
                      const float *target_depth, const float *target_std, const float *u_target,
                      const int32_t *target_row, float *z2_sorted, float *z_all, int64_t *sort_idx, void *stream);

/* The same with the clamp window read on the device: near_far -> two floats (near0, far0), e.g. &rays[0][6] for
 * the reference's `near[0,0]`, `far[0,0]` (rendering.py:133,144) - no device->host read in the caller. */
int bn_guided_samples_nf(const float *z, const float *weights, const float *depth, const float *u, int64_t R,
                         int32_t S, int32_t G, const float *near_far, float d_range, const float *use_target,
                         const float *target_depth, const float *target_std, const float *u_target,
                         const int32_t *target_row, float *z2_sorted, float *z_all, int64_t *sort_idx, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Per-ray BRDF shading (eval_RPV / eval_Hapke / eval_microfacet_brdf, models/spsbrdfnerf.py:9-30;
 * BRDF/RPV.py:40-63, BRDF/Hapke.py:139-200, BRDF/microfacet.py:20-72, BRDF/basic_func.py:5-44).
 * All inputs [N][3] unless noted; nullable inputs select the reference's "None" branches.
 * Forward writes brdf [N][3] and aux [N][BN_BRDF_AUX] (model specific, see csrc/brdf.hip);
 * backward writes gradients w.r.t. the non-null differentiable inputs.
 * ------------------------------------------------------------------------------------------- */
#define BN_BRDF_AUX 16
int bn_brdf_rpv_forward(const float *l, const float *v, const float *n, const float *w, const float *k,
                        const float *theta, const float *rhoc, int64_t N, float *brdf, float *aux, void *stream);
int bn_brdf_rpv_backward(const float *l, const float *v, const float *n, const float *w, const float *k,
                         const float *theta, const float *rhoc, const float *d_brdf, int64_t N, float *d_n,
                         float *d_w, float *d_k, float *d_theta, float *d_rhoc, void *stream);
int bn_brdf_hapke_forward(const float *l, const float *v, const float *n, const float *w, const float *b,
                          const float *c, const float *theta /*[N]*/, float hpk_scl, int32_t shell, int64_t N,
                          float *brdf, float *aux, void *stream);
int bn_brdf_hapke_backward(const float *l, const float *v, const float *n, const float *w, const float *b,
                           const float *c, const float *theta, float hpk_scl, int32_t shell, const float *d_brdf,
                           int64_t N, float *d_n, float *d_w, float *d_b, float *d_c, float *d_theta, void *stream);
int bn_brdf_microfacet_forward(const float *l, const float *v, const float *n, const float *albedo,
                               const float *rough /*[N]*/, float f0, int64_t N, float *brdf, float *aux,
                               void *stream);
int bn_brdf_microfacet_backward(const float *l, const float *v, const float *n, const float *albedo,
                                const float *rough, float f0, const float *d_brdf, int64_t N, float *d_n,
                                float *d_albedo, float *d_rough, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Training-step tail (main.py:147-168 Adam; metrics.py:39-61 MSE): fused Adam over one flat fp32
 * parameter buffer (the nn.Parameters are views into it).
 * ------------------------------------------------------------------------------------------- */
int bn_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                 float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_scale,
                 void *stream);

/* ---------------------------------------------------------------------------------------------
 * Launch-lean fused training step (round 3).  The reference's step is ~150 ATen launches of glue around the field
 * evaluations (rendering.py:168-291, models/spsbrdfnerf.py:198-416, metrics.py:39-161, main.py:147-168); these entry points
 * fold that glue into a handful of per-ray kernels so that a Lambertian step is <= 20 launches and can be captured in a
 * HIP graph (no argument changes from step to step: draws, step counters and the learning rate live in device memory).
 * ------------------------------------------------------------------------------------------- */

/* Device-resident state of the fused step (bn_step_state, 512 bytes, 16-byte aligned; the caller allocates and zero-fills
 * it, sets rng_seed / lr, and may rewrite lr between steps):
 *   [0]  u64 rng_seed      Philox key of the in-kernel draws
 *   [8]  u64 rng_step      Philox counter word, advanced by bn_adam_multi (one per training step)
 *   [16] f32 lr            learning rate read by bn_adam_multi
 *   [20] u32 done          internal (completion ticket of bn_adam_multi)
 *   [24] i32 adam_step[4]  optimiser step count per parameter group (torch.optim.Adam keeps one per parameter)
 *   [40] f32 noise_std     --noise_std of the running step, read by the compositing kernels when bn_noise.noise_std < 0 (ABI 7)
 *   [64] f32 loss_ring[64] loss of step t at slot t % 64 (written by bn_adam_multi from the partial sums)
 *   [320] f64 beta1_pow[4], [352] f64 beta2_pow[4]   beta^adam_step per group (1.0 at step 0): the bias corrections
 *   [512] f32 loss_part[64] partial sums of the running step's loss (bn_lambert_tail adds ray r's term to slot r % 64;
 *                           bn_adam_multi folds them into the ring and clears them) */
#define BN_STATE_BYTES 1024
#define BN_STATE_NOISE_OFF 40
#define BN_STATE_LOSS_OFF 64
#define BN_STATE_LOSS_SLOTS 64
#define BN_STATE_POW_OFF 320
#define BN_STATE_PART_OFF 512
#define BN_ADAM_MAX_GROUPS 4
/* in-kernel draw streams of one step */
enum { BN_RNG_COARSE = 1, BN_RNG_GUIDED = 2, BN_RNG_GUIDED_TARGET = 3, BN_RNG_NOISE_COARSE = 4, BN_RNG_NOISE_MERGED = 5, BN_RNG_SUN = 6 };

/* --noise_std (models/spsbrdfnerf.py:57-59: alphas = 1 - exp(-deltas * relu(sigmas + randn * noise_std)), main.py:246) with
 * in-kernel draws (ABI 6): a standard normal per (ray, sample position) from stream `rng_stream` of the state at `rng`
 * (Box-Muller on a Philox pair), element (ray_offset + r) * S + s of the stream, S = the sample count of the compositing it
 * perturbs (pass 1: the S coarse samples; the final compositing: the depth-sorted S + G set, position = sorted position).
 * NULL, or noise_std == 0: no noise.  noise_std < 0 (ABI 7): the kernels read the value from the step state at `rng` (f32 at byte
 * BN_STATE_NOISE_OFF): the reference multiplies noise_std by 0.9 after EVERY step (main.py:246), and a value baked into the launch
 * arguments would give every step a new signature - no graph replay while the noise decays.  The forward and the backward of a
 * step name the same stream and see the same draws. */
typedef struct {
  const unsigned long long *rng;
  float noise_std;
  uint32_t rng_stream;
  int64_t ray_offset;
} bn_noise;
/* The standard normals an in-kernel noise stream hands to elements 0..n-1, as an array (tests, replay). */
int bn_rng_normal(const unsigned long long *rng, uint32_t rng_stream, int64_t n, float *x, void *stream);

/* bn_stratified_z with in-kernel uniforms (stream `rng_stream` of the state at `rng` = &state[0]).  The draw of sample s of
 * ray r is element (ray_offset + r) * S + s of the stream: a rank that holds rays [ray_offset, ray_offset + R) of a global
 * batch draws exactly what a single process would draw for them. */
int bn_stratified_z_rng(const float *near, const float *far, int64_t nf_stride, const unsigned long long *rng,
                        uint32_t rng_stream, int64_t ray_offset, int64_t R, int32_t S, float *z, void *stream);
/* The uniforms an in-kernel stream hands to elements 0..n-1, as an array (tests, and callers that want to replay a step). */
int bn_rng_uniform(const unsigned long long *rng, uint32_t rng_stream, int64_t n, float *u, void *stream);

/* Pass-1 compositing (cal_weight on sigma [R][S], element stride sigma_stride; `noise` nullable, see bn_noise) +
 * bn_guided_samples_nf in one launch.  Per-ray prior arrays carry an element stride (depths[:, 0] of the [R][2] table, satellite_rgb_dep.py:339).
 * Draws: arrays u [R][G] / u_target [R][G] (one row per ray), or - when u == NULL - the in-kernel streams rng_u / rng_ut
 * of `rng`.  weights [R][S] / depth [R] (nullable) receive the pass-1 weights and depth. */
int bn_composite_guided(const float *z, const float *sigma, int64_t sigma_stride, int64_t R, int32_t S, int32_t G,
                        const float *near_far, float d_range, const float *use_target, int64_t ut_stride,
                        const float *target_depth, int64_t td_stride, const float *target_std, int64_t ts_stride,
                        const float *u, const float *u_target, const unsigned long long *rng, uint32_t rng_u,
                        uint32_t rng_ut, int64_t ray_offset, float *z2_sorted, float *z_all, int64_t *sort_idx, float *weights,
                        float *depth, const bn_noise *noise, void *stream);

/* Compositing of the depth-sorted union of two field outputs without materialising it: sample s of ray r is row
 * sort_idx[r][s] of cat[out1[r] ([S1][C]), out2[r] ([S2-S1][C])] (rendering.py:263-272; sigma = channel 3).  sort_idx ==
 * NULL: out1 alone (S1 == S2).  Forward: cal_weight + the weighted sums acc [R][C] (acc[:, 3] is not defined), wsum [R] =
 * sum_s w.  Backward: from d_weights [R][S2], d_depth [R], d_acc [R][C] (d_acc[:, 3] ignored), d_wsum [R] (all nullable)
 * to the gradient rows d_out1 / d_out2 in the SOURCE layouts (channel 3 = d sigma).  nonfinite (nullable): NaN / Inf
 * gradient elements are written as 0 and counted ([0] NaN, [1] Inf) - FusedTrainer.sanitize_grads without extra passes. */
/* NormalRegLoss (metrics.py:179-216) inside the merged-set compositing: lambda_an sum_s w_s min(0, n_an,s . view)^2 + the same
 * for the learned normals, view = -rays_d [R] rows with element stride rd_stride; channel < 0 or lambda <= 0: that field is off.
 * The forward writes the ray's term to reg_out [R] (an extra loss term for bn_ray_shade_loss), the backward adds its gradient
 * w.r.t. the weights and the per-sample normal channels.  NULL (or rays_d == NULL): off. */
typedef struct {
  const float *rays_d;
  int64_t rd_stride;
  int32_t ch_an, ch_lr;
  float lambda_an, lambda_lr;
  /* NormalLoss between the two per-sample normal fields (metrics.py:218-261, keyword 'an_lr' = --nr_spv_type 1, main.py:297-303):
   * lambda_spv * mean(weights) * mean|n_an - n_lr| over the whole batch - two batch-wide means, so the step makes three stops:
   * the forward writes each ray's (sum_s w_s, sum_s sum_c |n_an - n_lr|) to spv_ray [R][2]; bn_normal_spv_reduce adds them up in a
   * fixed order into spv_tot [4] = (d loss / d w_s, d loss / d n_an per component and unit sign, the loss term, 0) and adds the
   * term to the step's loss; the backward reads spv_tot.  lambda_spv == 0: off.  (ABI 6) */
  float lambda_spv;
  int32_t spv_ch_an, spv_ch_lr;
  float *spv_ray;
  float *spv_tot;
} bn_normal_reg;
int bn_normal_spv_reduce(const float *spv_ray, int64_t R, int32_t S, float lambda_spv, float *spv_tot, float *ray_loss,
                         float *loss_acc, void *stream);
int bn_merged_composite_forward(const float *z, const int64_t *sort_idx, const float *out1, const float *out2, int32_t S1,
                                int32_t S2, int32_t C, int64_t R, float *alphas, float *trans, float *weights, float *depth,
                                float *acc, float *wsum, float *var, const bn_normal_reg *nreg, float *reg_out,
                                const bn_noise *noise, void *stream);
/* hs_scale != 0 (with depth [R], the forward's result): adds hs_scale (z_s - depth)^2 to d loss / d w_s - the per-sample part
 * of HardSurfaceLoss's gradient (metrics.py:263-290), see bn_ray_shade_loss. */
int bn_merged_composite_backward(const float *z, const int64_t *sort_idx, const float *out1, const float *out2, int32_t S1,
                                 int32_t S2, int32_t C, int64_t R, const float *d_weights, const float *d_depth,
                                 const float *d_acc, const float *d_wsum, float hs_scale, const float *depth,
                                 const bn_normal_reg *nreg, float *d_out1, float *d_out2, unsigned long long *nonfinite,
                                 const bn_noise *noise, void *stream);

/* Ray-level shading + losses of a training step whose rays have ONE BRDF each (MultiBRDF == 0) and no per-sample
 * irradiance, forward AND backward in one launch (one thread per ray), between bn_merged_composite_forward and
 * bn_merged_composite_backward.  From the composited sums acc [R][C], wsum [R], depth [R] and var [R] = sum_s w (z - depth)^2:
 *   albedo_s = acc[0:3] (1 + 2 pad) - pad wsum ;  n = l2_normalize(acc[ch_normal : +3]) ;  parameters = acc[ch_p* : ...]
 *   rgb = clamp(irradiance * BRDF(sun_d, -rays_d, n, albedo_s, parameters), 0, 1)        (models/spsbrdfnerf.py:259-357)
 *   loss = SNerfLoss(rgb, rgbs) + DepthLoss + HardSurfaceLoss                           (metrics.py:39-61, 82-161, 263-290)
 * and d loss / d acc [R][C], d loss / d wsum [R], d loss / d depth [R] (the HardSurfaceLoss term's per-sample part is
 * bn_merged_composite_backward's hs_scale = lambda_hs / R).  kind: BN_SHADE_*; channels < 0: head absent (RPV: p0 k, p1
 * theta, p2 rhoc, or rhoc = albedo_s with rhoc_is_albedo (funcH == 2); Hapke: p0 b, p1 c, p2 theta (1 wide); microfacet: p0
 * roughness (1 wide)).  sun_d NULL: (1,1,1).  The prior arrays carry element strides and are nullable together.  nonfinite
 * (nullable, [0] NaN [1] Inf counters): a ray whose loss term is not finite is left out (loss 0, gradients 0) and counted -
 * without it the NaN reaches the loss and the gradients as it does upstream.  extra_loss [R] (nullable): per-ray terms computed
 * elsewhere that belong to the step's loss (bn_merged_composite_forward's reg_out), added to the ray's term. */
enum { BN_SHADE_LAMBERT = 0, BN_SHADE_RPV = 1, BN_SHADE_HAPKE = 2, BN_SHADE_MICROFACET = 3 };
typedef struct {
  int32_t kind, C, ch_normal, ch_p0, ch_p1, ch_p2;
  int32_t rhoc_is_albedo, shell, cos_irradiance, usealldepth;
  float hpk_scl, f0, rgb_padding, lambda_rgb, lambda_ds, lambda_hs;
  /* ABI 6: per-ray irradiance (nullable; element stride irr_stride) - the sun visibility of the ray's LAST sample from the
   * sun pass (--sun_v analystic, rendering.py:244-259, models/spsbrdfnerf.py:354), a constant of the step; the cosine term
   * (cos_irradiance) wins when both are given, as upstream (:260-266). */
  const float *irr;
  int64_t irr_stride;
} bn_shade_desc;
int bn_ray_shade_loss(const bn_shade_desc *desc, const float *acc, const float *wsum, const float *depth, const float *var,
                      const float *rays_d, int64_t rd_stride, const float *sun_d, int64_t sd_stride, const float *rgbs,
                      const float *valid_depth, int64_t v_stride, const float *target_depth, int64_t td_stride,
                      const float *target_weight, int64_t tw_stride, const float *target_std, int64_t ts_stride, int64_t R,
                      float *rgb, float *ray_loss, float *loss_acc, int32_t loss_slots, float *d_acc, float *d_wsum,
                      float *d_depth, unsigned long long *nonfinite, const float *extra_loss, void *stream);
/* Ray-level tail of a Lambertian step in ONE launch: bn_merged_composite_forward + bn_lambert_loss (shading, SNerfLoss,
 * DepthLoss; metrics.py:39-61,82-161) + bn_merged_composite_backward.  The prior arrays carry element strides.  ray_loss [R]
 * (nullable) and/or loss_acc (nullable): ray r's term is atomically added to loss_acc[r % loss_slots] - partial sums the
 * caller adds up (4096 atomics on ONE word serialise to ~50 us; the step state's 64 partials are folded into its loss ring
 * by bn_adam_multi).  nonfinite (nullable, [0] NaN [1] Inf counters), as in bn_ray_shade_loss / bn_merged_composite_backward:
 * a ray whose loss term is not finite is left out (loss 0, gradients 0) and counted, non-finite gradient elements are zeroed
 * and counted; without it a NaN colour stays a NaN in rgb, in the loss and in the gradients, as it does upstream. */
int bn_lambert_tail(const float *z, const int64_t *sort_idx, const float *out1, const float *out2, int32_t S1, int32_t S2,
                    int32_t C, int64_t R, const float *rgbs, const float *valid_depth, int64_t v_stride,
                    const float *target_depth, int64_t td_stride, const float *target_weight, int64_t tw_stride,
                    const float *target_std, int64_t ts_stride, float rgb_padding, float lambda_rgb, float lambda_ds,
                    int32_t usealldepth, float *ray_loss, float *loss_acc, int32_t loss_slots, float *rgb,
                    float *weights, float *depth, float *d_out1, float *d_out2, unsigned long long *nonfinite,
                    const bn_noise *noise, void *stream);

/* Per-SAMPLE shading on the field-output rows as they are stored, one launch forward and one backward: --MultiBRDF
 * (models/spsbrdfnerf.py:289-307, 350-352: every sample is shaded by its own BRDF, the shaded colours are composited) and the
 * per-sample irradiance of the sun pass (:265-273).  X [N][C]: rows [0, n1) belong to ray row / S1, rows [n1, N) to ray
 * (row - n1) / S2 (the [pass-1 | guided] blocks of a training step, n1 == R S1 and N - n1 == R S2 for the R rays; n1 == N: one block).  view = -rays[ray][3:6], sun =
 * rays[ray][sun_col : +3] (sun_col < 0: (1, 1, 1)); the normal, albedo (channels 0-2) and parameter channels are read per row as
 * bn_shade_desc names them (its loss fields unused).  kind LAMBERT: the "BRDF" is the albedo itself.  irradiance: |sun_z| with
 * cos_irradiance and a normal channel, else desc->irr[row * irr_stride] - here per ROW, not per ray - when given, else 1.
 *   forward:  B[row][0:3] = (BRDF (1 + 2 pad) - pad) * irradiance ;  B[row][3] = sigma ;
 *             b_stride == C: B[row][4:] = X[row][4:]   (b_stride is 4 or C)
 *   backward: dX[row] = J^T dB[row][0:3] on the normal / albedo / parameter channels + dB[row][3] on sigma
 *             (+ dB[row][4:] on the channels behind it when b_stride == C); every channel of dX is written.
 * The derivative conventions are those of the per-point BRDF backward entry points above: torch autograd's. */
int bn_sample_brdf_forward(const bn_shade_desc *desc, const float *X, const float *rays, int64_t R, int64_t ray_stride,
                           int32_t sun_col, int64_t N, int64_t n1, int32_t S1, int32_t S2, float *B, int32_t b_stride, void *stream);
int bn_sample_brdf_backward(const bn_shade_desc *desc, const float *X, const float *rays, int64_t R, int64_t ray_stride,
                            int32_t sun_col, int64_t N, int64_t n1, int32_t S1, int32_t S2, const float *dB, int32_t b_stride,
                            float *dX, void *stream);

/* Folding of the linear feats layer into the heads' first layers (bn_field_desc.fold_feats) and the chain rule back, as
 * two launches of exact-fp32 MFMA tiles:
 *   bn_fold_heads:    w_fold[h] = w1[h][:, :F] wf ;  b_fold[h] = w1[h][:, :F] bf + b1[h] ;  m[h] = 0, s[h] = 0
 *   bn_unfold_heads:  d_w1[h][:, :F] += m[h] wf^T + s[h] bf^T ;  d_b1[h] += s[h] ;  d_wf += sum_h w1[h]^T m[h] ;  d_bf += sum_h w1[h]^T s[h]
 * m[h] [rows][F] / s[h] [rows] are the gradients of the folded first layers that bn_field_backward accumulates. */
typedef struct {
  int32_t n_heads, F, rows;              /* rows = hidden width of the heads (F / 2) */
  const float *wf, *bf;                  /* feats_from_xyz [F][F], [F] */
  const float *w1[BN_MAX_HEADS];         /* first-layer weights [rows][w1_ld], columns [0, F) used */
  int64_t w1_ld[BN_MAX_HEADS];
  const float *b1[BN_MAX_HEADS];
  float *w_fold[BN_MAX_HEADS], *b_fold[BN_MAX_HEADS];   /* [rows][F], [rows] */
  float *m[BN_MAX_HEADS], *s[BN_MAX_HEADS];             /* folded-gradient accumulators (nullable: inference) */
  float *d_w1[BN_MAX_HEADS];
  int64_t d_w1_ld[BN_MAX_HEADS];
  float *d_b1[BN_MAX_HEADS];
  float *d_wf, *d_bf;
} bn_fold_desc;
int bn_fold_heads(const bn_fold_desc *d, void *stream);
int bn_unfold_heads(const bn_fold_desc *d, void *stream);

/* Adam over the parameter groups of one flat buffer in ONE launch (torch.optim.Adam semantics per group, main.py:147-168):
 * group g = elements [lo[g], hi[g]) (multiples of 4), stepped only when active[g] (a group that is not in this step's graph
 * has grad None upstream and is skipped).  Step counts, the learning rate and the draw counter live in `state`
 * (bn_step_state): the kernel uses adam_step[g] + 1 and, when its last workgroup finishes, increments the active groups'
 * counts and rng_step and folds the loss partials into the loss ring.  zero_grad != 0: the gradient elements it read are set to 0. */
int bn_adam_multi(float *param, float *grad, float *exp_avg, float *exp_avg_sq, int32_t n_groups, const int64_t *lo,
                  const int64_t *hi, const int32_t *active, float beta1, float beta2, float eps, float weight_decay,
                  float grad_scale, int32_t zero_grad, void *state, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Debug hook replacing the reference's check_nan / checknan / check_nan_parms (train_utils.py:14-78,
 * models/spsbrdfnerf.py:32-48,419-424), which copy a flag to the host and print after every call:
 * counts[0] += number of NaN elements of x[0..n), counts[1] += number of +-Inf elements, on the
 * stream, without a device->host synchronisation; the caller reads the two counters when it wants.
 * ------------------------------------------------------------------------------------------- */
int bn_count_nonfinite(const float *x, int64_t n, unsigned long long *counts, void *stream);

/* Device-side fault word of the fused kernels (bit 0: a wave of the barrier-free forward trunk gave up waiting for an LDS
 * hand-over - never in a correct run; the affected launch's results are invalid; bit 1: the same for the barrier-free trunk of
 * the backward chain (round 4; ABI 3-4 reported their turn-taking deterministic mode there).  The library mirrors both bits to
 * the host asynchronously (every 64th forward launch outside a stream capture) and fails the NEXT bn_field_forward /
 * bn_field_backward call with BN_ELAUNCH once one is set; bn_adam_multi reads both words on the device and leaves parameters and
 * moments untouched while one is set (ABI 7: a replayed graph cannot train on an invalid gradient); this call synchronises
 * `stream` and reads them directly. */
int bn_device_faults(unsigned int *faults, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement hooks (no reference counterpart; the reference only has Lightning's wall-clock
 * "simple" profiler, main.py:731).  When enabled, every kernel launch of this library is
 * bracketed by HIP events on the caller's stream; bn_prof_collect() synchronises them, adds the
 * durations per kernel id (order below) and clears the log.
 * ids: 0 pack, 1 field_fwd(sigma only), 2 field_fwd(full), 3 field_bwd chain, 4 wgrad, 5 skinny
 * wgrad, 6 composite fwd, 7 composite bwd, 8 guided samples, 9 stratified z, 10 adam, 11 brdf, 12 normals adjoint,
 * 13 normals adjoint backward.
 * ------------------------------------------------------------------------------------------- */
#define BN_PROF_IDS 14
int bn_prof_enable(int on);
int bn_prof_collect(double *ms_sum, int *count, int n_ids);

#ifdef __cplusplus
}
#endif
#endif /* BRDFNERF_HIP_H */
