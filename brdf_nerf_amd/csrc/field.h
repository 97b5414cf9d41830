// Geometry, packed-weight layout and activation-stash layout of the fused field MLP.
// Shared by the pack, forward, backward and weight-gradient kernels (host + device).
#pragma once
#include "common.h"

#define BN_MAX_PASS 3
#ifndef BN_DPH                    // (diag.h: an A/B build may pin 12 = the four-head layout of ABI <= 2, profiles/history/r02_ablation.txt)
#define BN_DPH (3 * BN_MAX_HEADS)   // pre-activation gradients of the heads' (<= 3) outputs kept per point
#endif
// 32-column tiles per wave in a single-head pass (N = F/2 columns): NT/2 spreads the pass over all eight waves; NT keeps
// half of them idle but halves the LDS fragment reads per MFMA (BN_HEAD_WIDE, A/B switch).
#ifdef BN_HEAD_WIDE
#define BN_SINGLE_HEAD_NTW(NT) (NT)
#else
#define BN_SINGLE_HEAD_NTW(NT) ((NT) > 1 ? (NT) / 2 : 1)
#endif

struct FieldGeom {
  int F, L, skip, pe_freqs, act;
  int P;        // valid trunk input width: 6*pe_freqs, or 3 without mapping
  int KP;       // P rounded up to 64 (60 -> 64, 3 -> 64): k-extent of the PE operand (4 MFMA k-steps)
  int NT;       // 32-column tiles per wave in the F-wide phases
  int BM, waves; // points per workgroup tile, waves per workgroup (tile configuration, bn_tile_config)
  int dsz;       // bytes per element of the derivative stashes D_l / DG: 4 (fp32), 2 (fp16 + analytic normals), 1 (common.h bn_dsize)
  int H2;       // head hidden width F/2
  int n_heads, n_pass;
  int pass_heads[BN_MAX_PASS];  // heads evaluated in pass p (2 or 1)
  int pass_N[BN_MAX_PASS];      // columns of pass p = pass_heads * H2
  int pass_NTW[BN_MAX_PASS];    // tiles per wave in pass p
  int head_col[BN_MAX_HEADS];   // first output channel of head i in `out`
  int C;                        // out channels
  int ch_normal_an, ch_normal_lr;  // channel index or -1
  int fold;                        // feats layer folded into the heads' first layers (bn_field_desc.fold_feats)
  // Extra inputs of pass 0's first layers, one LDS tile / stash row of KD columns (KD = 0: none, else 16 or 32):
  //   [0, DD)        the (encoded) view direction, read by head 0 (--input_viewdir; dir_freqs octaves, 0 = raw)
  //   [KT0, KT0+TD)  the per-image embedding, read by head 1 of kind BN_HEAD_BETA (--beta);  KT0 = round_up(DD, 8)
  int DD, KD, dir_freqs, TD, KT0;
};

// Tile configuration of the fused chain kernels: 8 waves per workgroup, one workgroup per CU (2 waves per SIMD);
// 64 points per tile in fp32 (parity mode), 128 in bf16.  A measured alternative - 64 points x 4 waves, two
// workgroups per CU running out of phase - lost 20 % (r01): each workgroup streams the full weight set from L2, so
// halving the tile doubles the L2->CU weight traffic per flop, which costs more than the phase overlap wins.
static inline void bn_tile_config(int dtype, int *BM, int *waves) {
  *BM = bn_half(dtype) ? 128 : 64;
  *waves = 8;
}

static inline int bn_make_geom(const bn_field_desc *d, FieldGeom *g) {
  BN_REQUIRE(d->feat >= 64 && d->feat <= 512 && d->feat % 64 == 0 && (d->feat == 512 || d->feat <= 256),
             "field: feat=%d unsupported (64,128,192,256,512)", d->feat);
  BN_REQUIRE(d->dtype == BN_F32 || d->dtype == BN_BF16 || d->dtype == BN_F16, "field: dtype=%d unknown", d->dtype);
  BN_REQUIRE(d->layers >= 2 && d->layers <= BN_MAX_LAYERS, "field: layers=%d unsupported", d->layers);
  BN_REQUIRE(d->skip < d->layers && d->skip != 0, "field: skip=%d invalid", d->skip);
  BN_REQUIRE(d->n_heads >= 1 && d->n_heads <= BN_MAX_HEADS, "field: n_heads=%d", d->n_heads);
  BN_REQUIRE(d->pe_freqs >= 0 && d->pe_freqs <= 10, "field: pe_freqs=%d", d->pe_freqs);
  g->F = d->feat; g->L = d->layers; g->skip = d->skip; g->pe_freqs = d->pe_freqs; g->act = d->act;
  g->P = d->pe_freqs > 0 ? 6 * d->pe_freqs : 3;
  g->KP = (g->P + 63) / 64 * 64;
  bn_tile_config(d->dtype, &g->BM, &g->waves);
  g->dsz = (int)bn_dsize(d->dtype, d->normal_an);
  {
    const int per = (d->feat + 32 * g->waves - 1) / (32 * g->waves);   // 32-column tiles each wave must cover
    g->NT = per <= 1 ? 1 : (per <= 2 ? 2 : 4);
  }
  g->H2 = d->feat / 2;
  g->fold = d->fold_feats != 0;
  g->DD = d->dir_dim; g->dir_freqs = d->dir_freqs; g->TD = d->t_dim;
  g->KT0 = (d->dir_dim + 7) / 8 * 8;
  g->KD = (g->KT0 + d->t_dim + 15) / 16 * 16;
  BN_REQUIRE(d->dir_dim == 0 || (d->dir_dim == (d->dir_freqs > 0 ? 6 * d->dir_freqs : 3) && d->dir_dim <= 32 && g->fold),
             "field: dir_dim=%d dir_freqs=%d fold=%d unsupported (input_viewdir needs fold_feats and at most 5 octaves)", d->dir_dim,
             d->dir_freqs, g->fold);
  {
    const bool beta = d->n_heads >= 2 && d->head_kind[1] == BN_HEAD_BETA;
    BN_REQUIRE(beta == (d->t_dim > 0) && d->t_dim >= 0 && d->t_dim <= 16 && (!beta || (g->fold && d->head_out[1] == 1)),
               "field: t_dim=%d needs exactly a 1-wide head 1 of kind BN_HEAD_BETA (and fold_feats)", d->t_dim);
    for (int i = 0; i < d->n_heads; ++i)
      BN_REQUIRE(i == 1 || d->head_kind[i] != BN_HEAD_BETA, "field: BN_HEAD_BETA is head 1 only (head %d)", i);
    BN_REQUIRE(g->KD <= 32, "field: dir_dim=%d + t_dim=%d do not fit the 32-column extra-input tile", d->dir_dim, d->t_dim);
  }
  g->n_heads = d->n_heads;
  g->n_pass = (d->n_heads + 1) / 2;
  int c = d->t_dim > 0 ? 5 : 4;          // [rgb3, sigma, (beta)]
  g->ch_normal_an = g->ch_normal_lr = -1;
  if (d->normal_an) { g->ch_normal_an = c; c += 3; }
  if (d->normal_lr) { g->ch_normal_lr = c; c += 3; }
  g->head_col[0] = 0;
  for (int i = 1; i < d->n_heads; ++i) {
    BN_REQUIRE(d->head_out[i] == 1 || d->head_out[i] == 3, "field: head_out[%d]=%d", i, d->head_out[i]);
    if (d->head_kind[i] == BN_HEAD_BETA) { g->head_col[i] = 4; continue; }
    g->head_col[i] = c;
    // 1-wide heads are tiled x3 in the output except roughness / Hapke theta (spsbrdfnerf.py:726,731,755)
    c += (d->head_kind[i] == BN_HEAD_PLAIN || d->head_kind[i] == BN_HEAD_HAPKE_THETA) ? d->head_out[i] : 3;
  }
  BN_REQUIRE(d->head_out[0] == 3, "field: head 0 must be rgb (3 outputs)");
  g->C = c;
  BN_REQUIRE(d->out_channels == c, "field: out_channels=%d but layout needs %d", d->out_channels, c);
  for (int p = 0; p < g->n_pass; ++p) {
    g->pass_heads[p] = (d->n_heads - 2 * p) >= 2 ? 2 : 1;
    g->pass_N[p] = g->pass_heads[p] * g->H2;
    g->pass_NTW[p] = g->pass_heads[p] == 2 ? g->NT : BN_SINGLE_HEAD_NTW(g->NT);
  }
  return 0;
}

// ------------------------------------------------------------------ packed weights (elements of T)
// A matrix with R rows and K contraction columns is stored as [R/32][K/16][64 lanes][8]: element j of
// lane (r = lane&31, h = lane>>5) of block (rt, ks) is M[rt*32 + r][ks*16 + 8h + j]  (one MFMA A operand
// = one fully coalesced 64-lane x 16 B (bf16) load).
struct PackedLayout {
  size_t fwd_trunk[BN_MAX_LAYERS][2];  // [l][0]: PE part (l==0, l==skip) or h part; [l][1]: h part of the skip layer
  size_t fwd_feats, fwd_head[BN_MAX_PASS];
  size_t fwd_sigma, fwd_nlr;           // one 32-row tile each over K = F: row 0 = w_sigma / rows 0..2 = grad_from_xyz (rest zero)
  size_t fwd_dir;                      // [pass_N[0] rows][KD]: the extra-input columns of pass 0's first layers (FieldGeom.KD)
  size_t bwd_trunk[BN_MAX_LAYERS];     // W_l^T restricted to the h inputs, l >= 1
  size_t bwd_feats, bwd_head[BN_MAX_PASS];
  size_t bwd_pe[2];                    // (W_l[:, :P])^T for l = 0 and l = skip: [KP rows][F k], analytic-normal adjoint only
  size_t total;
};

static inline size_t bn_pad(size_t v, size_t a) { return (v + a - 1) / a * a; }

static inline void bn_make_packed_layout(const FieldGeom &g, PackedLayout *pl) {
  size_t off = 0;
  auto take = [&](size_t rows, size_t k) { size_t o = off; off += bn_pad(rows, 32) * bn_pad(k, 16); return o; };
  for (int l = 0; l < g.L; ++l) {
    pl->fwd_trunk[l][0] = pl->fwd_trunk[l][1] = 0;
    if (l == 0) pl->fwd_trunk[l][0] = take(g.F, g.KP);
    else if (l == g.skip) { pl->fwd_trunk[l][0] = take(g.F, g.KP); pl->fwd_trunk[l][1] = take(g.F, g.F); }
    else pl->fwd_trunk[l][0] = take(g.F, g.F);
  }
  // The trunk and the sigma tile come first, at offsets that depend on the trunk geometry alone: a sigma-only evaluation (pass 1
  // with gsam_only, the sun-visibility pass) is described by a desc WITHOUT BRDF heads / normals / extra inputs and still reads
  // the packed buffer of the model's full desc (brdf_nerf_amd/rendering.py inference(_packed=...)).
  pl->fwd_sigma = take(32, g.F);
  pl->fwd_feats = g.fold ? 0 : take(g.F, g.F);
  for (int p = 0; p < g.n_pass; ++p) pl->fwd_head[p] = take(g.pass_N[p], g.F);
  pl->fwd_nlr = g.ch_normal_lr >= 0 ? take(32, g.F) : 0;
  pl->fwd_dir = g.KD > 0 ? take(g.pass_N[0], g.KD) : 0;
  for (int l = 0; l < g.L; ++l) pl->bwd_trunk[l] = l >= 1 ? take(g.F, g.F) : 0;
  pl->bwd_feats = g.fold ? 0 : take(g.F, g.F);
  for (int p = 0; p < g.n_pass; ++p) pl->bwd_head[p] = take(g.F, g.pass_N[p]);
  pl->bwd_pe[0] = take(g.KP, g.F);
  pl->bwd_pe[1] = g.skip > 0 ? take(g.KP, g.F) : 0;
  pl->total = off;
}

// ------------------------------------------------------------------ activation stash (byte offsets)
// Row-major arrays [Mpad][width] of T feed the weight-gradient GEMMs; "native" arrays hold one
// accumulator-register image per tile ([tile][wave][nt][mt][g][lane][4]) and are only re-read by
// the backward chain, which uses the same tiling.
struct StashLayout {
  size_t gscale;                  // fp32 [2] as bits: max |d pre-activation| (primal chain), max |gbar_PE| (adjoint chain): fp16 loss scaling
  size_t sraw;                    // fp32 [Mpad]  pre-softplus sigma
  size_t nraw;                    // fp32 [Mpad][4] learned-normal pre-normalisation vector
  size_t dpre_trunk;              // fp32 [Mpad][4]  (d sigma_raw, d normal_raw xyz)      (bwd-produced)
  size_t dpre_head;               // fp32 [Mpad][BN_DPH] (per head, 3 each)                   (bwd-produced)
  size_t pe;                      // T [Mpad][KP]
  size_t dirpe;                   // T [Mpad][KD]  extra-input tile rows: encoded view direction, image embedding (FieldGeom.KD)
  size_t Y[BN_MAX_LAYERS];        // T [Mpad][F]  output of trunk layer l
  size_t D[BN_MAX_LAYERS];        // DTile image  d act / d z of trunk layer l: 8-bit fixed point in the 16-bit modes, fp32 native in fp32
  size_t feats;                   // T [Mpad][F]
  size_t G[BN_MAX_PASS];          // T native     head hidden activations
  size_t DG[BN_MAX_PASS];         // DTile image (like D)
  size_t dZ[BN_MAX_LAYERS];       // T [Mpad][F]                                          (bwd-produced)
  size_t dfeats;                  // T [Mpad][F]                                          (bwd-produced)
  size_t dG[BN_MAX_PASS];         // T [Mpad][pass_N]                                     (bwd-produced)
  // analytic-normal training (double backward of the adjoint chain); allocated only when desc.normal_an
  size_t gradx;                   // fp32 [Mpad][4]  raw d sigma / d xyz
  size_t sbar;                    // fp32 [Mpad]     extra d L / d sigma_raw from the adjoint's sigmoid seed  (bwd-produced)
  size_t sprime;                  // fp32 [Mpad]     sigmoid(sigma_raw)
  size_t gbar_pe;                 // T [Mpad][KP]    d L / d g_PE                                           (bwd-produced)
  size_t adj_delta[BN_MAX_LAYERS];  // T [Mpad][F]   delta_l = a_{l+1} (.) D_l
  size_t adj_a[BN_MAX_LAYERS];      // T native      a_{l+1}
  size_t adj_abar[BN_MAX_LAYERS + 1];  // T [Mpad][F] abar_l, l = 1..L                                       (bwd-produced)
  size_t adj_zbar[BN_MAX_LAYERS];   // Elem<T>::wide native  extra d L / d z_l through D_l (bf16 in the fp16 mode)    (bwd-produced)
  size_t wgpart, wgpart_bytes;    // fp32 slabs of the weight-gradient kernels: one per (output tile, point split) (field_wgrad.hip)
  size_t total;
  int64_t Mpad;
};

// Point splits of one weight-gradient launch over `tiles` output tiles and Mpad points - THE place both the launcher
// (field_wgrad.hip bn_launch_wgrad) and the stash sizing below take them from (two formulas that could drift).
//   16-bit modes (256 x 256 tiles, one 144 KB workgroup per CU): one round of the 256 CUs up to 327,680 points, two beyond (both
//   passes of a 4096-ray step in one call: 524,288 points).  With the fp32 atomics of rounds 1-3 four rounds were fastest; with
//   slabs every split costs a 257 KB slab written and read once more: 1020 -> 510 workgroups took the launch from 2.372 to
//   2.344 ms and its reduce from 0.089 to 0.059 ms (profiles/r04_ab_slab_lambert.txt).  At least 512 points per workgroup.
//   fp32 parity mode (128 x 128 tiles): a grid of ~2048 workgroups, at least 256 points each.
// (BN_W2_BLOCKS = 256, diag.h: tiles x point splits per round of the 256 CUs)
static inline int64_t bn_wgrad_splits(bool half, int64_t tiles, int64_t Mpad, int64_t *m_per_block) {
  const int64_t stage = half ? 64 : 32;          // W2_BK / WG_BK: points per LDS stage
  const int64_t min_mpb = half ? 512 : 256;
  int64_t n_split = (half ? (Mpad <= 327680 ? BN_W2_BLOCKS : 2 * BN_W2_BLOCKS) : 2048) / (tiles > 0 ? tiles : 1);
  if (n_split < 1) n_split = 1;
  int64_t mpb = ceil_div64(ceil_div64(Mpad, n_split), stage) * stage;
  if (mpb < min_mpb) mpb = min_mpb;
  if (m_per_block) *m_per_block = mpb;
  return ceil_div64(Mpad, mpb);                  // <= the first n_split: tiles x splits <= max(tiles, the grid bound above)
}
// Bytes of the weight-gradient slab workspace for a point set of Mpad rows: tiles x splits of the largest job list a model can
// have ((2 L + 12) matrices of ceil(F / tile)^2 tiles each: trunk + analytic-normal jobs + heads), and the skinny jobs'
// [job][split] slabs.  bn_launch_wgrad checks its actual product against this (BN_REQUIRE: an error, never an overrun).
static inline size_t bn_wgpart_bytes(const FieldGeom &g, int64_t Mpad, size_t esz) {
  const bool half = esz != 4;
  const int64_t tl = half ? 256 : 128, ft = (g.F + tl - 1) / tl;
  const int64_t tiles_bound = (2 * g.L + 12) * ft * ft;
  // tiles x splits is largest either for the full job list or where the split count peaks (few tiles): both are bounded by
  // max(tiles, grid bound), and by tiles x the split count of a ONE-tile launch
  const int64_t grid_bound = half ? 2 * BN_W2_BLOCKS : 2048;
  int64_t slabs = tiles_bound > grid_bound ? tiles_bound : grid_bound;
  const int64_t by_points = tiles_bound * bn_wgrad_splits(half, 1, Mpad, nullptr);
  if (by_points < slabs) slabs = by_points;
  const size_t wg = (size_t)slabs * (size_t)(tl * tl + tl) * 4;
  int64_t sk_splits = ceil_div64(Mpad, g.BM);
  if (sk_splits > 256) sk_splits = 256;
  const size_t sk = (size_t)10 * (size_t)sk_splits * (4 * 512 + 4) * 4;
  return wg > sk ? wg : sk;
}

static inline void bn_make_stash_layout(const FieldGeom &g, int64_t n_points, int BM, size_t esz, StashLayout *s) {
  const size_t dsz = (size_t)g.dsz;      // DK32 / DK16 / DK8 (field_kernels.h): 4, 2 or 1 byte per derivative
  int64_t Mpad = ceil_div64(n_points, BM) * BM;
  s->Mpad = Mpad;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += bn_pad(bytes, 256); return o; };
  s->gscale = take(256);
  s->sraw = take((size_t)Mpad * 4);
  s->nraw = take((size_t)Mpad * 16);
  s->dpre_trunk = take((size_t)Mpad * 16);
  s->dpre_head = take((size_t)Mpad * BN_DPH * 4);
  s->pe = take((size_t)Mpad * g.KP * esz);
  s->dirpe = g.KD > 0 ? take((size_t)Mpad * g.KD * esz) : 0;
  for (int l = 0; l < g.L; ++l) s->Y[l] = take((size_t)Mpad * g.F * esz);
  for (int l = 0; l < g.L; ++l) s->D[l] = take((size_t)Mpad * g.F * dsz);
  s->feats = take((size_t)Mpad * g.F * esz);
  for (int p = 0; p < BN_MAX_PASS; ++p) {
    s->G[p] = s->DG[p] = s->dG[p] = 0;
    if (p < g.n_pass) {
      s->G[p] = take((size_t)Mpad * g.F * esz);   // native images sized for a full-width phase
      s->DG[p] = take((size_t)Mpad * g.F * dsz);
    }
  }
  for (int l = 0; l < g.L; ++l) s->dZ[l] = take((size_t)Mpad * g.F * esz);
  s->dfeats = take((size_t)Mpad * g.F * esz);
  for (int p = 0; p < g.n_pass; ++p) s->dG[p] = take((size_t)Mpad * g.pass_N[p] * esz);
  s->gradx = s->sbar = s->sprime = s->gbar_pe = 0;
  for (int l = 0; l < BN_MAX_LAYERS; ++l) s->adj_delta[l] = s->adj_a[l] = s->adj_abar[l] = s->adj_zbar[l] = 0;
  s->adj_abar[BN_MAX_LAYERS] = 0;
  if (g.ch_normal_an >= 0) {
    s->gradx = take((size_t)Mpad * 16);
    s->sbar = take((size_t)Mpad * 4);
    s->sprime = take((size_t)Mpad * 4);
    s->gbar_pe = take((size_t)Mpad * g.KP * esz);
    for (int l = 0; l < g.L; ++l) {
      s->adj_delta[l] = take((size_t)Mpad * g.F * esz);
      s->adj_a[l] = take((size_t)Mpad * g.F * esz);
      s->adj_abar[l + 1] = take((size_t)Mpad * g.F * esz);
      s->adj_zbar[l] = take((size_t)Mpad * g.F * esz);
    }
  }
  s->wgpart_bytes = bn_wgpart_bytes(g, Mpad, esz);
  s->wgpart = take(s->wgpart_bytes);
  s->total = off;
}

// Layout of a forward / normals call that writes points [off, off + n) of a larger set's stash: the set's layout with every
// per-point array advanced by `off` points (off a multiple of the tile: the native tile images advance by whole tiles).
static inline void bn_stash_layout_at(const FieldGeom &g, const bn_points *pts, int BM, size_t esz, StashLayout *s) {
  const int64_t total = pts->total_points > 0 ? pts->total_points : pts->n_points;
  bn_make_stash_layout(g, total, BM, esz, s);
  const size_t off = (size_t)pts->point_offset;
  if (off == 0) return;
  const size_t dsz = (size_t)g.dsz, F = (size_t)g.F;
  s->sraw += off * 4; s->nraw += off * 16; s->dpre_trunk += off * 16; s->dpre_head += off * BN_DPH * 4;
  s->pe += off * g.KP * esz;
  if (g.KD > 0) s->dirpe += off * g.KD * esz;
  for (int l = 0; l < g.L; ++l) { s->Y[l] += off * F * esz; s->D[l] += off * F * dsz; s->dZ[l] += off * F * esz; }
  s->feats += off * F * esz; s->dfeats += off * F * esz;
  for (int p = 0; p < g.n_pass; ++p) { s->G[p] += off * F * esz; s->DG[p] += off * F * dsz; s->dG[p] += off * g.pass_N[p] * esz; }
  if (g.ch_normal_an >= 0) {
    s->gradx += off * 16; s->sbar += off * 4; s->sprime += off * 4; s->gbar_pe += off * g.KP * esz;
    for (int l = 0; l < g.L; ++l) {
      s->adj_delta[l] += off * F * esz; s->adj_a[l] += off * F * esz; s->adj_abar[l + 1] += off * F * esz; s->adj_zbar[l] += off * F * esz;
    }
  }
  s->Mpad = ceil_div64(pts->n_points, BM) * BM;     // rows this call walks
}
