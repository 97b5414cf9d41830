"""Differential fuzz of the field kernels against the CPU oracle over RANDOM network shapes and flag sets (fixed seeds).

Every case draws a geometry load_model() can build - width, depth (the skip layer sits at 4: deeper networks have it, shallower
ones do not), with / without positional encoding, Siren / ReLU,
the BRDF family and its heads, normal mode, --input_viewdir, --beta, dim_RPV - and a ragged point count, then compares the HIP
fp32 path (forward, every parameter gradient, the gradient w.r.t. the --beta embedding input) with the oracle's autograd
(oracle/field.py, pinned by the reference's goldens).  The fixed-configuration tests in test_gpu_parity.py cover the reference's
own shapes; this covers the combinations between them."""
import os

import numpy as np
import pytest
import torch

from conftest import tparams
from oracle.config import FieldConfig
from oracle import field as OF

pytestmark = pytest.mark.gpu
DEV = "cuda"
# BN_FUZZ_SCALE=k runs k times as many seeds per test (a one-off hunt; the committed default is 1)
_K = int(os.environ.get("BN_FUZZ_SCALE", "1"))


def draw_config(rng):
    feat = int(rng.choice([64, 128, 192, 256, 512]))
    layers = int(rng.integers(2, 9))
    family = rng.choice(["lambert", "rpv", "hapke", "microfacet"])
    kw = dict(feat=feat, layers=layers, siren=bool(rng.random() < 0.7), mapping=bool(rng.random() < 0.8),
              normal=str(rng.choice(["none", "learned", "analystic", "analystic_learned"])),
              input_viewdir=int(rng.random() < 0.35), beta=bool(rng.random() < 0.35), n_samples=16, guided_samples=16)
    if kw["beta"]:
        kw["t_dim"] = int(rng.choice([2, 4, 8]))
    if family == "rpv":
        kw.update(funcM=int(rng.random() < 0.7), funcF=int(rng.random() < 0.7), funcH=int(rng.random() < 0.7), dim_RPV=int(rng.choice([1, 3])))
        if not (kw["funcM"] or kw["funcF"] or kw["funcH"]):
            kw["funcM"] = 1
        # funcH == 2 (rhoc := albedo, no rhoc head; spsbrdfnerf.py:306,317) for a third of the draws without a rhoc head - chosen
        # from values already drawn, so the random stream of the other cases is the one of round 2
        if kw["funcH"] == 0 and (feat // 64 + layers) % 3 == 0:
            kw["funcH"] = 2
            if not (kw["funcM"] or kw["funcF"]):
                kw["funcM"] = 1
    elif family == "hapke":
        kw.update(b=1, c=int(rng.random() < 0.7), theta=int(rng.random() < 0.5))
    elif family == "microfacet":
        kw.update(roughness=True)
    if family != "lambert" and kw["normal"] == "none":
        kw["normal"] = "learned"                      # the BRDFs need a normal
    return FieldConfig(**kw)


@pytest.mark.parametrize("seed", list(range(64 * _K)))
def test_random_field_configuration_against_oracle(seed):
    from test_gpu_parity import build_model, diag
    rng = np.random.default_rng(1000 + seed)
    cfg = draw_config(rng)
    B = int(rng.integers(1, 400))
    nr_an = cfg.normal in ("analystic", "analystic_learned")
    flags = dict(apply_brdf=bool(rng.random() < 0.8), apply_theta=bool(rng.random() < 0.7), nr_an_on=nr_an,
                 nr_lr_on=cfg.normal in ("learned", "analystic_learned"))
    model = build_model(cfg, 40 + seed)
    p = tparams(cfg, 40 + seed)
    for v in p.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, 3, generator=g) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1) if cfg.dir_dim else None
    t_ref = torch.randn(B, cfg.t_dim, generator=g).requires_grad_(True) if cfg.beta else None
    ref = OF.field_forward(p, cfg, xyz, dirs=dirs, t_embed=t_ref, **flags)
    coef = torch.randn(ref.shape, generator=g)
    t_gpu = t_ref.detach().to(DEV).requires_grad_(True) if cfg.beta else None
    out = model(xyz.to(DEV), input_dir=None if dirs is None else dirs.to(DEV), input_t=t_gpu, **flags)
    tag = (f"fuzz {seed}: F={cfg.feat} L={cfg.layers} siren={int(cfg.siren)} pe={int(cfg.mapping)} normal={cfg.normal} "
           f"viewdir={cfg.input_viewdir} beta={int(cfg.beta)} heads={cfg.brdf_head_names(flags['apply_brdf'], flags['apply_theta'])} B={B}")
    assert out.shape == ref.shape, tag
    dif = (out.detach().cpu() - ref.detach()).abs()
    err = float(dif.max())
    diag(f"{tag}: out max|err| {err:.2e}")
    kink = nr_an and not cfg.siren      # ReLU + analytic normal: d sigma/dx is piecewise constant - a pre-activation within rounding of
    if kink:                            # 0 takes the other branch in one of the two evaluations and moves that point's normal
        c0 = 5 if cfg.beta else 4
        other = torch.ones(out.shape[1], dtype=torch.bool)
        other[c0:c0 + 3] = False
        assert float(dif[:, other].max()) <= 2e-4 * float(ref.detach().abs().max()) + 2e-5, tag
        at_kink = dif[:, c0:c0 + 3].amax(-1) > 2e-4
        assert int(at_kink.sum()) <= max(1, B // 50), f"{tag}: analytic normals off at too many points"
        # ONE point on the other side of a ReLU moves a bias gradient by several percent (seed 847 of a 15 x hunt: 4.4 %, the fp32
        # oracle itself landing on either side depending on the host's BLAS threading): those points - the two evaluations took
        # different branches there, both legitimate - are left out of the backward comparison on BOTH sides, everything else is
        # held to the ordinary analytic-normal tolerance
        coef[at_kink] = 0.0
        if int(at_kink.sum()):
            diag(f"{tag}: {int(at_kink.sum())} point(s) at a ReLU kink left out of the backward comparison")
    else:
        assert err <= 2e-4 * float(ref.detach().abs().max()) + 2e-5, tag
    (ref * coef).sum().backward()
    (out * coef.to(DEV)).sum().backward()
    tol = 1e-3 if nr_an else 2e-4
    if cfg.beta:
        scale = float(t_ref.grad.abs().max())
        assert float((t_gpu.grad.cpu() - t_ref.grad).abs().max()) <= tol * scale + 1e-7, tag + " d_t_embed"
    truth = {}

    def fp64_grads():      # only when a comparison fails: is the fp32 ORACLE the one that is off (an ill-conditioned case)?
        if not truth:
            p64 = {k_: v_.detach().double().requires_grad_(True) for k_, v_ in p.items()}
            o64 = OF.field_forward(p64, cfg, xyz.double(), dirs=None if dirs is None else dirs.double(),
                                   t_embed=None if t_ref is None else t_ref.detach().double(), **flags)
            (o64 * coef.double()).sum().backward()
            truth.update({k_: v_.grad for k_, v_ in p64.items() if v_.grad is not None})
        return truth

    for k, v in model.named_parameters():
        want = p[k].grad
        got = v.grad
        if want is None:
            assert got is None or float(got.abs().max()) == 0.0, f"{tag} {k}: unused parameter got a gradient"
            continue
        scale = float(want.abs().max())
        e = float((got.cpu() - want).abs().max()) if got is not None else scale
        if e > tol * scale + 1e-7 and got is not None:
            t64 = fp64_grads()[k]
            e_hip, e_ref = float((got.cpu().double() - t64).abs().max()), float((want.double() - t64).abs().max())
            assert e_hip <= 3 * e_ref + tol * float(t64.abs().max()) + 1e-7, \
                f"{tag} {k}: err vs fp64 {e_hip:.3e} (the fp32 oracle's own: {e_ref:.3e}) scale {scale:.3e}"
            diag(f"{tag} {k}: fp64 referee used (HIP {e_hip:.2e}, fp32 oracle {e_ref:.2e}, scale {scale:.2e})")
            continue
        assert e <= tol * scale + 1e-7, f"{tag} {k}: err {e:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("seed", list(range(40 * _K)))
def test_random_field_configuration_half_modes_track_fp32(seed):
    """The same random configurations in the 16-bit throughput modes against the HIP fp32 mode (itself held to the oracle
    above): outputs within the stated half bounds, every sizeable parameter gradient pointing the same way.  Catches
    shape-specific faults of the 16-bit kernels (256-tile weight gradient at widths that are not multiples of 256,
    native-order stashes with idle waves, loss scaling) that the fixed F = 512 / 64 tests cannot see."""
    from test_gpu_parity import build_model, diag, HALF_BOUNDS
    rng = np.random.default_rng(5000 + seed)
    cfg = draw_config(rng)
    dtype = "bf16" if seed % 2 == 0 else "fp16"
    B = int(rng.integers(1, 700))
    nr_an = cfg.normal in ("analystic", "analystic_learned")
    flags = dict(apply_brdf=bool(rng.random() < 0.8), apply_theta=bool(rng.random() < 0.7), nr_an_on=nr_an,
                 nr_lr_on=cfg.normal in ("learned", "analystic_learned"))
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.rand(B, 3, generator=g) * 2 - 1).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1).to(DEV) if cfg.dir_dim else None
    t0 = torch.randn(B, cfg.t_dim, generator=g).to(DEV) if cfg.beta else None
    outs, grads, tg = {}, {}, {}
    coef = None
    for dt in ("fp32", dtype):
        model = build_model(cfg, 70 + seed, dt)
        t_in = t0.clone().requires_grad_(True) if cfg.beta else None
        out = model(xyz, input_dir=dirs, input_t=t_in, **flags)
        if coef is None:
            coef = torch.randn(out.shape, generator=g).to(DEV)
            if nr_an:                               # the analytic normal of a random network can be ill-conditioned: weigh it down
                coef[:, 5 if cfg.beta else 4:(8 if cfg.beta else 7)] *= 0.1
        (out * coef).sum().backward()
        outs[dt] = out.detach()
        grads[dt] = {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}
        tg[dt] = t_in.grad.clone() if cfg.beta else None
        # the inference variants of the forward kernel (no stash; sigma only) must give what the training forward gave
        with torch.no_grad():
            inf = model(xyz, input_dir=dirs, input_t=t0, **dict(flags, nr_an_on=False))
            sg = model(xyz, sigma_only=True)
        keep = [c for c in range(out.shape[1]) if not (nr_an and (5 if cfg.beta else 4) <= c < (8 if cfg.beta else 7))]
        ref_cols = out.detach()[:, keep]
        assert inf.shape[1] == len(keep), f"{dt}: inference forward has {inf.shape[1]} channels, expected {len(keep)}"
        tol_inf = 1e-5 if dt == "fp32" else 2e-2
        assert float((inf - ref_cols).abs().max()) <= tol_inf * max(1.0, float(ref_cols.abs().max())), f"{dt}: inference forward differs from the training forward"
        assert float((sg[:, 0] - out.detach()[:, 3]).abs().max()) <= tol_inf * max(1.0, float(out.detach()[:, 3].abs().max())), f"{dt}: sigma-only forward differs"
    tag = (f"fuzz-half {seed} {dtype}: F={cfg.feat} L={cfg.layers} siren={int(cfg.siren)} pe={int(cfg.mapping)} normal={cfg.normal} "
           f"viewdir={cfg.input_viewdir} beta={int(cfg.beta)} heads={cfg.brdf_head_names(flags['apply_brdf'], flags['apply_theta'])} B={B}")
    b = HALF_BOUNDS[dtype]
    assert bool(torch.isfinite(outs[dtype]).all()), tag
    e_rgb = float((outs[dtype][:, :3] - outs["fp32"][:, :3]).abs().max())
    sig = outs["fp32"][:, 3]
    e_sig = float(((outs[dtype][:, 3] - sig).abs() / (sig.abs() + 1.0)).max())
    worst = (1.0, "")
    for k, g32 in grads["fp32"].items():
        g16 = grads[dtype].get(k)
        assert g16 is not None and bool(torch.isfinite(g16).all()), f"{tag} {k}: non-finite gradient"
        if float(g32.norm()) < 1e-6 * max(1.0, float(g32.numel()) ** 0.5) or g32.numel() < 16:
            continue                               # (a scalar / 3-vector bias gradient is a near-cancelling sum: its direction is noise)
        c = float(torch.nn.functional.cosine_similarity(g16.flatten().double(), g32.flatten().double(), dim=0))
        worst = min(worst, (c, k))
    diag(f"{tag}: rgb err {e_rgb:.2e} sigma rel err {e_sig:.2e} worst gradient cosine {worst[0]:.4f} ({worst[1]})")
    # deeper / narrower random networks than the reference's are less forgiving than F = 512: twice the stated F = 512 bounds
    assert e_rgb <= 2 * b["rgb"] and e_sig <= 2 * b["sig"], tag
    assert worst[0] >= (0.90 if nr_an else (0.95 if dtype == "bf16" else 0.97)), f"{tag}: gradient cosine {worst[0]:.4f} at {worst[1]}"
    if cfg.beta:
        c = float(torch.nn.functional.cosine_similarity(tg[dtype].flatten().double(), tg["fp32"].flatten().double(), dim=0))
        assert c >= 0.97, f"{tag}: d_t_embed cosine {c:.4f}"


def _sat_rays(R, g):
    """Satellite-shaped rays: origins on a plane above the unit cube, near-nadir view directions, one sun direction per image."""
    o = torch.cat([torch.rand(R, 2, generator=g) * 1.6 - 0.8, 1.0 + 0.02 * torch.rand(R, 1, generator=g)], -1)
    d = torch.nn.functional.normalize(torch.cat([0.2 * torch.randn(R, 2, generator=g), -torch.ones(R, 1)], -1), dim=-1)
    sun = torch.nn.functional.normalize(torch.tensor([0.3, -0.4, 0.85]) + 0.05 * torch.randn(3, generator=g), dim=0).expand(R, 3)
    return torch.cat([o, d, torch.zeros(R, 1), 2.0 * torch.ones(R, 1), sun], -1).contiguous()


@pytest.mark.parametrize("seed", list(range(40 * _K)))
def test_random_render_rays_against_oracle(seed):
    """render_rays end to end (both passes, guided sampling, merge, compositing, shading) on random configurations against
    the oracle's render_rays with the SAME random draws (recorded from the oracle, replayed into the HIP path in the
    reference's order): ragged ray / sample counts, train and test mode with depth priors, gsam_only, the sun-visibility
    pass, MultiBRDF, density noise, --beta, --input_viewdir.  Ray-level results against an fp64 evaluation of the same
    algorithm: error <= 2e-4 relative + 4 x the fp32 oracle's own error on that ray + its worst ray's; at most 2 % of the rays may differ more
    (a guided sample that lands on the other side of a bin edge by one ulp changes that ray, not the rest)."""
    from test_gpu_parity import build_model, make_args, Replay, diag
    from brdf_nerf_amd import render_rays
    from oracle import render as ORD
    rng = np.random.default_rng(9000 + seed)
    cfg = draw_config(rng)
    S, G = int(rng.choice([8, 16, 24, 40])), int(rng.choice([8, 16, 24]))
    kw = dict(vars(cfg))
    kw.update(feat=int(rng.choice([64, 128, 192])), n_samples=S, guided_samples=G, noise_std=float(rng.choice([0.0, 0.0, 0.3])))
    brdf = bool(cfg.roughness or cfg.RPV or cfg.b)
    gsam_only = bool(rng.random() < 0.3)
    if brdf and gsam_only and rng.random() < 0.5:
        kw["sun_v"] = "analystic"
    if cfg.RPV and rng.random() < 0.3:
        kw["MultiBRDF"] = True
    cfg = FieldConfig(**kw)
    R = int(rng.integers(1, 90))
    mode = "train" if rng.random() < 0.5 else "test"
    flags = dict(apply_brdf=brdf and bool(rng.random() < 0.85), apply_theta=bool(rng.random() < 0.7), cos_irra_on=bool(rng.random() < 0.6),
                 gsam_only=gsam_only)
    if cfg.sun_v == "analystic" and not flags["apply_brdf"]:
        flags["apply_brdf"] = True
    g = torch.Generator().manual_seed(seed)
    rays = _sat_rays(R, g)
    dk = {}
    if mode == "train" and rng.random() < 0.6:
        dk = dict(valid_depth=(torch.rand(R, generator=g) < 0.6).float(),
                  target_depths=torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1),
                  target_std=0.02 + 0.05 * torch.rand(R, generator=g))
    p = tparams(cfg, 90 + seed)
    emb = torch.randn(6, cfg.t_dim, generator=g) if cfg.beta else None
    ts = torch.randint(0, 6, (R,), generator=g) if cfg.beta else None
    rnd = ORD.Randoms(generator=torch.Generator().manual_seed(100 + seed))
    with torch.no_grad():
        ref, bt_ref = ORD.render_rays(p, cfg, rays, rnd, mode=mode, rays_t=emb[ts] if cfg.beta else None, **flags, **dk)
        # the same algorithm and draws in fp64: random networks make some rays ill-conditioned (a normal from a tiny gradient,
        # a BRDF near grazing incidence) - there the fp32 REFERENCE is itself off by 1e-3, and the HIP path is judged by its
        # error against this truth relative to the reference's own
        truth, _ = ORD.render_rays({k: v.double() for k, v in p.items()}, cfg, rays.double(), ORD.Randoms(replay=[t.double() for t in rnd.log]),
                                   mode=mode, rays_t=emb[ts].double() if cfg.beta else None, **flags,
                                   **{k: v.double() for k, v in dk.items()})
    model = build_model(cfg, 90 + seed)
    models = {"coarse": model}
    if cfg.beta:
        models["t"] = torch.nn.Embedding(6, cfg.t_dim).to(DEV)
        with torch.no_grad():
            models["t"].weight.copy_(emb)
    with Replay(rnd.log) as rp, torch.no_grad():
        res, bt = render_rays(models, make_args(cfg), rays.to(DEV), None if ts is None else ts.to(DEV), mode=mode,
                              **flags, **{k: v.to(DEV) for k, v in dk.items()})
        assert rp.draws == [], "consumed a different number of random draws than the oracle"
    tag = (f"fuzz-render {seed}: F={cfg.feat} L={cfg.layers} siren={int(cfg.siren)} normal={cfg.normal} sun_v={cfg.sun_v} multi={int(cfg.MultiBRDF)} "
           f"beta={int(cfg.beta)} viewdir={cfg.input_viewdir} noise={cfg.noise_std} R={R} S={S} G={G} {mode} {flags} prior={bool(dk)}")
    assert bt == bt_ref, tag
    worst = 0.0
    for k in ("rgb_coarse", "depth_coarse", "weights_coarse", "z_vals_coarse", "albedo_accu_coarse", "transparency_coarse"):
        a, r32, r64 = res[k].detach().cpu().double().reshape(R, -1), ref[k].double().reshape(R, -1), truth[k].reshape(R, -1)
        e_ref = (r32 - r64).abs().amax(-1)                                                     # the fp32 reference's own error, per ray
        # (two fp32 evaluations of an ill-conditioned ray err by the same SCALE, not the same amount: the worst ray's error counts too)
        err = ((a - r64).abs() - 2e-4 * r64.abs()).amax(-1) - 4.0 * e_ref - e_ref.max()        # per ray
        bad = int((err > 5e-5).sum())
        worst = max(worst, float(err.clamp_min(0).max()))
        assert bad <= max(1, R // 50), f"{tag}: {k}: {bad} of {R} rays off (worst excess {float(err.max()):.2e})"
    diag(f"{tag}: worst excess over (2e-4 rel + 4 x reference error) {worst:.2e}")
    for k, v in res.items():
        if v.dtype.is_floating_point and k != "hpk_scl_coarse":
            assert bool(torch.isfinite(v).all()) == bool(torch.isfinite(ref[k]).all()), f"{tag}: {k} finiteness differs"
    assert {k for k in ref if not k.startswith("_")} == set(res), tag


class _Record:
    """Pass-through recorder of torch.rand / rand_like / randn (device draws of the HIP path), in call order."""

    def __init__(self):
        self.log = []

    def __enter__(self):
        self._o = (torch.rand, torch.rand_like, torch.randn)
        o_rand, o_like, o_randn = self._o

        def rand(*a, **k):
            t = o_rand(*a, **k); self.log.append(t.clone()); return t

        def rand_like(x, **k):
            t = o_like(x, **k); self.log.append(t.clone()); return t

        def randn(*a, **k):
            t = o_randn(*a, **k); self.log.append(t.clone()); return t

        torch.rand, torch.rand_like, torch.randn = rand, rand_like, randn
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like, torch.randn = self._o


@pytest.mark.parametrize("seed", list(range(32 * _K)))
def test_random_fused_step_against_autograd_path(seed):
    """FusedTrainer.step (the path bench.py times: stash forward, composite, loss glue, explicit backward kernels, flat
    gradient, fused Adam) on random configurations against render_rays + losses + loss.backward() + torch.optim.Adam of the
    HIP autograd path (itself held to the oracle above) on the SAME draws: loss, every gradient, parameters after the step.
    Random width / depth / heads / normals, regulariser lambdas, depth priors (target_std = 0: reference quirk 7), gsam_only,
    sun visibility, MultiBRDF, --beta (left out by the fused step), --input_viewdir; fp32 mode."""
    from test_gpu_parity import build_model, make_args, Replay, diag
    from brdf_nerf_amd import render_rays, losses
    from brdf_nerf_amd.trainer import FusedTrainer
    rng = np.random.default_rng(13000 + seed)
    cfg = draw_config(rng)
    S, G = int(rng.choice([8, 16, 24])), int(rng.choice([8, 16]))
    kw = dict(vars(cfg))
    kw.update(feat=int(rng.choice([64, 128, 192])), n_samples=S, guided_samples=G)
    brdf = bool(cfg.roughness or cfg.RPV or cfg.b)
    gsam_only = bool(rng.random() < 0.3)
    if brdf and gsam_only and rng.random() < 0.5:
        kw["sun_v"] = "analystic"
    if cfg.RPV and rng.random() < 0.3:
        kw["MultiBRDF"] = True
    cfg = FieldConfig(**kw)
    args = make_args(cfg)
    R = int(rng.integers(4, 80))
    flags = dict(apply_brdf=brdf and bool(rng.random() < 0.85), apply_theta=bool(rng.random() < 0.7), cos_irra_on=bool(rng.random() < 0.6))
    if cfg.sun_v == "analystic":
        flags["apply_brdf"] = True
    lam = dict(nr_reg_an_lambda=float(rng.choice([0.0, 0.2])), nr_reg_lr_lambda=float(rng.choice([0.0, 0.1])),
               hs_lambda=float(rng.choice([0.0, 0.3])), nr_spv_lambda=float(rng.choice([0.0, 0.05])))
    with_depth = bool(rng.random() < 0.5)
    g = torch.Generator().manual_seed(seed)
    rays = _sat_rays(R, g).to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV)
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV)
    dstd = torch.zeros(R, device=DEV)
    dk = dict(valid_depth=valid, target_depths=depths, target_std=dstd) if with_depth else {}
    tag = (f"fuzz-step {seed}: F={cfg.feat} L={cfg.layers} siren={int(cfg.siren)} normal={cfg.normal} sun_v={cfg.sun_v} multi={int(cfg.MultiBRDF)} "
           f"beta={int(cfg.beta)} viewdir={cfg.input_viewdir} R={R} S={S} G={G} gsam_only={gsam_only} depth={with_depth} {flags} {lam}")

    ma = build_model(cfg, 60 + seed)
    opt = torch.optim.Adam(ma.parameters(), lr=5e-4)
    models_a, ts = {"coarse": ma}, None
    if cfg.beta:
        models_a["t"] = torch.nn.Embedding(5, cfg.t_dim).to(DEV)
        ts = torch.randint(0, 5, (R,), generator=g).to(DEV)
    torch.manual_seed(7)
    with _Record() as rec:
        res, _ = render_rays(models_a, args, rays, ts, mode="train", gsam_only=gsam_only, **flags, **dk)
    loss_a = losses.snerf_loss(res["rgb_coarse"], rgbs)
    w, z, d = res["weights_coarse"], res["z_vals_coarse"], res["depth_coarse"]
    if with_depth:
        loss_a = loss_a + losses.depth_loss(z, d, w, depths[:, 0], depths[:, 1], valid, dstd, 10.0)
    view = -rays[:, 3:6]
    if lam["nr_reg_an_lambda"] > 0 and "normal_an_coarse" in res:
        loss_a = loss_a + losses.normal_reg_loss(res["normal_an_coarse"], w, view, lam["nr_reg_an_lambda"])[0]
    if lam["nr_reg_lr_lambda"] > 0 and "normal_lr_coarse" in res:
        loss_a = loss_a + losses.normal_reg_loss(res["normal_lr_coarse"], w, view, lam["nr_reg_lr_lambda"])[0]
    if lam["hs_lambda"] > 0:
        loss_a = loss_a + losses.hard_surface_loss(z, d, w, lam["hs_lambda"])
    if lam["nr_spv_lambda"] > 0 and "normal_an_coarse" in res and "normal_lr_coarse" in res:
        loss_a = loss_a + losses.normal_loss(w, res["normal_an_coarse"], res["normal_lr_coarse"], lam["nr_spv_lambda"])
    loss_a.backward()
    grads_a = {k: v.grad.clone() for k, v in ma.named_parameters() if v.grad is not None}
    opt.step()

    # the fused step draws (R, G) uniforms for the depth-prior rows where render_rays draws (n_valid, G); with target_std = 0 the
    # guided samples of those rows do not depend on them
    n_valid = int(valid.sum())
    draws = [torch.rand(R, G, device=DEV) if (with_depth and tuple(t.shape) == (n_valid, G) and i >= 3) else t for i, t in enumerate(rec.log)]
    mb = build_model(cfg, 60 + seed)
    tr = FusedTrainer(mb, args, lr=5e-4, ds_lambda=10.0 if with_depth else 0.0, **lam)
    tr.sanitize_grads = False
    with Replay(draws) as rp:
        loss_b, _ = tr.step(rays, rgbs, valid_depth=valid if with_depth else None, depths=depths if with_depth else None,
                            depth_std=dstd if with_depth else None, gsam_only=gsam_only, **flags)
        assert rp.draws == [], f"{tag}: the fused step consumed a different number of draws"
    la, lb = float(loss_a.detach()), float(loss_b)
    if not np.isfinite(la):
        pytest.skip(f"{tag}: the autograd path's loss is not finite for this random model")
    assert abs(la - lb) <= 2e-5 * abs(la) + 1e-7, f"{tag}: loss {lb} vs {la}"
    worst = 0.0
    for k, ga in grads_a.items():
        if not bool(torch.isfinite(ga).all()):
            continue                                   # (singular BRDF point of a random model: autograd itself gives NaN / inf)
        gb = tr.grad_views[k]
        scale = float(ga.abs().max())
        e = float((gb - ga).abs().max())
        worst = max(worst, e / max(scale, 1e-30))
        assert e <= 5e-4 * scale + 1e-9, f"{tag}: grad {k}: err {e:.3e} scale {scale:.3e}"
    diag(f"{tag}: loss {la:.6f}, worst relative gradient error {worst:.2e}")
    for k in tr.grad_views:
        if k not in grads_a:
            assert float(tr.grad_views[k].abs().max()) == 0.0, f"{tag}: {k} has no gradient in the autograd path"
    if all(bool(torch.isfinite(ga).all()) for ga in grads_a.values()):
        for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
            assert float((pa - pb).detach().abs().max()) <= 2.1 * 5e-4, f"{tag}: param {k} after Adam"


@pytest.mark.parametrize("seed", list(range(40 * _K)))
def test_random_fused_step_half_and_deterministic(seed):
    """The fused step on random configurations in a 16-bit mode: (1) deterministic mode twice - bitwise identical flat
    gradients (random widths give tile / split / launch-generation counts the fixed tests do not); (2) against the fp32
    fused step on the same draws - loss within 2 %, flat gradient pointing the same way."""
    import brdf_nerf_amd
    from test_gpu_parity import build_model, make_args, Replay, diag
    from brdf_nerf_amd.trainer import FusedTrainer
    rng = np.random.default_rng(17000 + seed)
    cfg = draw_config(rng)
    S, G = int(rng.choice([8, 16, 32])), int(rng.choice([8, 16, 32]))
    kw = dict(vars(cfg))
    kw.update(n_samples=S, guided_samples=G)
    brdf = bool(cfg.roughness or cfg.RPV or cfg.b)
    gsam_only = bool(rng.random() < 0.35)
    if brdf and gsam_only and rng.random() < 0.5:
        kw["sun_v"] = "analystic"
    if cfg.RPV and rng.random() < 0.3:
        kw["MultiBRDF"] = True
    cfg = FieldConfig(**kw)
    dtype = "bf16" if seed % 2 == 0 else "fp16"
    R = int(rng.integers(16, 200))
    flags = dict(apply_brdf=brdf and bool(rng.random() < 0.85), apply_theta=bool(rng.random() < 0.7), cos_irra_on=bool(rng.random() < 0.6),
                 gsam_only=gsam_only)
    if cfg.sun_v == "analystic":
        flags["apply_brdf"] = True
    g = torch.Generator().manual_seed(seed)
    rays = _sat_rays(R, g).to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    if rng.random() < 0.5:          # depth priors (target_std = 0: the guided rows of valid rays do not depend on the uniforms)
        flags.update(valid_depth=(torch.rand(R, generator=g) < 0.6).float().to(DEV),
                     depths=torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV),
                     depth_std=torch.zeros(R, device=DEV))
    tag = (f"fuzz-step16 {seed} {dtype}: F={cfg.feat} L={cfg.layers} siren={int(cfg.siren)} normal={cfg.normal} sun_v={cfg.sun_v} "
           f"multi={int(cfg.MultiBRDF)} R={R} S={S} G={G} gsam_only={gsam_only} prior={'depths' in flags} "
           f"{ {k: v for k, v in flags.items() if isinstance(v, bool)} }")

    def run(dt, det, draws=None):
        prev = brdf_nerf_amd.set_deterministic(det)
        try:
            model = build_model(cfg, 80 + seed, dt)
            tr = FusedTrainer(model, make_args(cfg, dt), lr=5e-4, hs_lambda=0.2, ds_lambda=10.0 if "depths" in flags else 0.0)
            if draws is None:
                torch.manual_seed(5)
                with _Record() as rec:
                    loss, _ = tr.step(rays, rgbs, **flags)
                draws = rec.log
            else:
                with Replay(list(draws)) as rp:
                    loss, _ = tr.step(rays, rgbs, **flags)
                    assert rp.draws == []
            torch.cuda.synchronize()
            return float(loss), tr.flat_grad.clone(), draws
        finally:
            brdf_nerf_amd.set_deterministic(prev)

    l32, g32, draws = run("fp32", False)
    l16a, g16a, _ = run(dtype, True, draws)
    l16b, g16b, _ = run(dtype, True, draws)
    assert torch.equal(g16a, g16b) and l16a == l16b, f"{tag}: deterministic mode is not reproducible (max diff {float((g16a - g16b).abs().max()):.3e})"
    if not (np.isfinite(l32) and bool(torch.isfinite(g32).all())):
        pytest.skip(f"{tag}: the fp32 step is not finite for this random model")
    assert bool(torch.isfinite(g16a).all()), tag
    cos = float(torch.nn.functional.cosine_similarity(g16a.double(), g32.double(), dim=0))
    diag(f"{tag}: loss fp32 {l32:.5f} {dtype} {l16a:.5f}, flat-gradient cosine {cos:.5f}")
    nr_an = cfg.normal in ("analystic", "analystic_learned")
    # Tracking is asserted where the problem is well conditioned.  With a BRDF on a RANDOM (untrained) model the loss gradient is
    # a near-cancelling sum over rays at grazing angles / GGX peaks: fp32 itself is dominated by a handful of rays there, and the
    # autograd path in the same 16-bit mode reproduces the fused step exactly (profiles/history/r02_ablation.txt, session 41 note) - those
    # cases only have to stay finite and keep the loss.  Without a BRDF: fp16 (11 bits) tightly - it runs the same templates as
    # bf16, so a kernel fault shows there - bf16 (8 bits) coarsely; its acceptance criterion is the PSNR gate.
    assert abs(l16a - l32) <= 0.05 * abs(l32) + 1e-4, f"{tag}: loss {l16a} vs {l32}"
    if not flags["apply_brdf"]:
        if dtype == "fp16":
            assert abs(l16a - l32) <= 0.02 * abs(l32) + 1e-4, f"{tag}: loss {l16a} vs {l32}"
            assert cos >= (0.90 if nr_an else 0.97), f"{tag}: cosine {cos:.4f}"
        else:
            assert cos >= 0.8, f"{tag}: cosine {cos:.4f}"


@pytest.mark.parametrize("seed", list(range(24 * _K)))
def test_random_composite_and_guided_shapes_against_oracle(seed):
    """The per-ray kernels at random (ragged) sizes: S in [1, 512], C in [1, 32], G in [1, 256], rays not a multiple of the
    workgroup's, density noise on / off, depth priors on / off.  Compositing forward + backward against the oracle's autograd;
    guided sampling + merge against the oracle with the same uniforms: depths to 2e-5, sort indices exact away from near-ties, the merged depths sorted and a permutation of the inputs."""
    from brdf_nerf_amd import functions as Fn
    from oracle import render as ORD
    rng = np.random.default_rng(21000 + seed)
    S, C, R = int(rng.integers(1, 513)), int(rng.integers(4, 33)), int(rng.integers(1, 150))
    if seed % 3 == 0:
        S = int(rng.integers(1, 70))
    noise_std = float(rng.choice([0.0, 0.4]))
    g = torch.Generator().manual_seed(seed)
    z = torch.sort(torch.rand(R, S, generator=g) * 2, -1)[0]
    out = torch.randn(R, S, C, generator=g)
    out[..., 3] = torch.relu(out[..., 3]) * 8 * (torch.rand(R, S, generator=g) < 0.3)
    noise = torch.randn(R, S, generator=g)
    cw, cd, ca = torch.rand(R, S, generator=g), torch.rand(R, generator=g), torch.rand(R, C, generator=g)
    ca[:, 3] = 0
    o_ref = out.clone().requires_grad_(True)
    a, T, w, d = ORD.composite(z, o_ref[..., 3], noise, noise_std)
    acc = (w.unsqueeze(-1) * o_ref).sum(-2)
    ((w * cw).sum() + (d * cd).sum() + (acc * ca).sum()).backward()
    o_gpu = out.clone().to(DEV).requires_grad_(True)
    a2, T2, w2, d2, acc2 = Fn.composite(z.to(DEV), o_gpu, noise.to(DEV) if noise_std else None, noise_std)
    ((w2 * cw.to(DEV)).sum() + (d2 * cd.to(DEV)).sum() + (acc2 * ca.to(DEV)).sum()).backward()
    tag = f"fuzz-ray {seed}: R={R} S={S} C={C} noise={noise_std}"
    for name, got, want, tol in (("alphas", a2, a, 1e-6), ("trans", T2, T, 1e-6), ("weights", w2, w, 1e-6), ("depth", d2, d, 2e-6), ("acc", acc2, acc, 2e-5)):
        e = float((got.detach().cpu() - want.detach()).abs().max())
        assert e <= 1e-5 * float(want.detach().abs().max()) + tol, f"{tag}: {name} err {e:.2e}"
    scale = float(o_ref.grad.abs().max())
    assert float((o_gpu.grad.cpu() - o_ref.grad).abs().max()) <= 1e-4 * scale + 1e-7, f"{tag}: d_out"

    # guided sampling around the composited depth (S >= 2 samples to resample from)
    S2 = max(S, 2) if S <= 256 else 256
    G = int(rng.integers(3, 257))           # (G = 1 fails inside the reference's sample_pdf, G = 2 is its single-sample special case)
    z = torch.sort(torch.rand(R, S2, generator=g) * 2, -1)[0]
    sig = torch.relu(torch.randn(R, S2, generator=g)) * 20 * (torch.rand(R, S2, generator=g) < 0.25)
    _, _, w, d = ORD.composite(z, sig)
    train = bool(rng.random() < 0.5)
    valid = (torch.rand(R, generator=g) < 0.5).float()
    tdep = torch.stack([0.5 + torch.rand(R, generator=g), torch.rand(R, generator=g)], -1)
    tstd = 0.02 + 0.05 * torch.rand(R, generator=g)
    rnd = ORD.Randoms(generator=torch.Generator().manual_seed(seed))
    z2_ref, _, _ = ORD.guided_samples(d, w, z, G, torch.tensor(0.0), torch.tensor(2.0), rnd, 3.0, "train" if train else "test",
                                      valid if train else None, tdep if train else None, tstd if train else None)
    z2_ref = torch.sort(z2_ref, -1)[0]
    z_all_ref, idx_ref = torch.sort(torch.cat([z, z2_ref], -1), -1)
    kw = {}
    if train:
        v = valid > 0
        u_t = rnd.log[1] if len(rnd.log) > 1 else torch.zeros(0, G)
        if int(v.sum()) > 0:
            kw = dict(use_target=valid.to(DEV), target_depth=tdep[:, 0].contiguous().to(DEV), target_std=tstd.to(DEV), u_target=u_t.to(DEV),
                      target_row=(torch.cumsum(v.int(), 0) - 1).clamp_min(0).int().to(DEV))
    z2, z_all, idx = Fn.guided_samples(z.to(DEV), w.to(DEV), d.to(DEV), rnd.log[0].to(DEV), 0.0, 2.0, 3.0, **kw)
    tag = f"fuzz-ray {seed}: guided R={R} S={S2} G={G} train={train}"
    # (2e-6 at the reference's G = 64; the inverse-CDF interpolation over up to 256 bins sums in another order than torch.cumsum)
    assert float((z2.cpu() - z2_ref).abs().max()) <= 2e-5, f"{tag}: z2 {float((z2.cpu() - z2_ref).abs().max()):.2e}"
    assert float((z_all.cpu() - z_all_ref).abs().max()) <= 2e-5, tag
    assert bool((z_all[:, 1:] >= z_all[:, :-1]).all()), tag
    assert torch.equal(torch.sort(idx, -1)[0].cpu(), torch.arange(S2 + G).expand(R, -1)), tag
    assert torch.equal(torch.gather(torch.cat([z.to(DEV), z2], -1), 1, idx), z_all), tag
    near = (z_all_ref[:, 1:] - z_all_ref[:, :-1]).abs() <= 4e-5                 # depths closer than the tolerance may swap places
    loose = torch.zeros_like(z_all_ref, dtype=torch.bool)
    loose[:, 1:] |= near
    loose[:, :-1] |= near
    mism = (idx.cpu() != idx_ref) & ~loose
    assert int(mism.sum()) == 0, f"{tag}: {int(mism.sum())} sort indices differ away from near-ties"


@pytest.mark.parametrize("seed", list(range(18 * _K)))
def test_random_brdf_inputs_against_oracle(seed):
    """The three BRDF kernels (forward and the forward-mode-Jacobian backward) on random geometries - ragged element counts,
    sun / view / normal directions over the upper hemisphere INCLUDING grazing and back-facing ones, parameters over their
    whole ranges - against the oracle (BRDF/RPV.py, Hapke.py, microfacet.py restated, oracle/brdf.py) and its autograd.
    Where the oracle's own gradient is NaN / inf (the models' singular points) any value is accepted, as in the golden tests."""
    from brdf_nerf_amd import functions as Fn
    from oracle import brdf as OB
    rng = np.random.default_rng(25000 + seed)
    N = int(rng.integers(1, 3000))
    g = torch.Generator().manual_seed(seed)

    def hemi(spread):
        v = torch.cat([spread * torch.randn(N, 2, generator=g), torch.ones(N, 1)], -1)
        return torch.nn.functional.normalize(v, dim=-1)

    l, v, n = hemi(0.6), hemi(0.8), hemi(float(rng.choice([0.2, 1.0, 3.0])))
    w = torch.rand(N, 3, generator=g)
    coef = torch.randn(N, 3, generator=g)
    family = ["rpv", "hapke", "microfacet"][seed % 3]
    tag = f"fuzz-brdf {seed}: {family} N={N}"

    def leaf(t):
        return t.clone().requires_grad_(True)

    def check(pairs, brdf_got, brdf_ref, rtol):
        ok = torch.isfinite(brdf_ref).all(-1)
        if bool(ok.any()):     # elementwise: the GGX lobe spans five decades over these inputs (alpha down to 0.0025)
            dv = (brdf_got.detach().cpu()[ok] - brdf_ref.detach()[ok]).abs() - 5 * rtol * brdf_ref.detach()[ok].abs()
            assert float(dv.max()) <= 1e-4, f"{tag}: brdf err excess {float(dv.max()):.2e}"
        for name, got, want in pairs:
            if want is None:                       # the oracle's output does not depend on it (RPV with the HG factor alone: no normal)
                assert got is None or float(got.abs().max()) == 0.0, f"{tag}: d{name} should be empty"
                continue
            fin = torch.isfinite(want)
            if not bool(fin.any()):
                continue
            # gradients near the models' singular points are huge and ill-conditioned: compare where the reference's is moderate
            mod = fin & (want.abs() <= 1e3)
            scale = float(want[mod].abs().max()) if bool(mod.any()) else 0.0
            err = (got.detach().cpu() - want)[mod].abs()
            tol = 5e-3 * want[mod].abs() + 2e-3 * max(scale, 1e-6) * 1e-2 + 1e-5
            bad = int((err > tol).sum())
            assert bad <= max(1, int(mod.sum()) // 200), f"{tag}: d{name}: {bad} of {int(mod.sum())} entries off (max err {float(err.max()):.2e})"

    if family == "rpv":
        use = [bool(rng.random() < 0.7) for _ in range(3)]
        if not any(use):
            use[0] = True
        k = 2 * torch.rand(N, 3, generator=g)
        th = 2 * torch.rand(N, 3, generator=g) - 1
        rc = torch.rand(N, 3, generator=g)
        rn, rw, rk, rt, rr = leaf(n), leaf(w), leaf(k), leaf(th), leaf(rc)
        ref = OB.rpv(l, v, rn, rw, rk if use[0] else None, rt if use[1] else None, rr if use[2] else None)[0]
        (ref * coef).sum().backward()
        dn, dw, dk, dt, dr = [leaf(t.to(DEV)) for t in (n, w, k, th, rc)]
        got, _ = Fn.RPVFunction.apply(l.to(DEV), v.to(DEV), dn, dw, dk if use[0] else None, dt if use[1] else None, dr if use[2] else None)
        (got * coef.to(DEV)).sum().backward()
        pairs = [("n", dn.grad, rn.grad), ("w", dw.grad, rw.grad)]
        pairs += [(nm, a.grad, b.grad) for nm, a, b, u in (("k", dk, rk, use[0]), ("theta", dt, rt, use[1]), ("rhoc", dr, rr, use[2])) if u]
        check(pairs, got, ref, 2e-4)
    elif family == "hapke":
        use_c, use_t = bool(rng.random() < 0.6), bool(rng.random() < 0.5)
        b = torch.rand(N, 3, generator=g)
        c = torch.rand(N, 3, generator=g)
        th = torch.rand(N, generator=g) * 0.5
        rn, rw, rb, rc_, rt = leaf(n), leaf(w), leaf(b), leaf(c), leaf(th)
        ref = OB.hapke(l, v, rn, rw, rb, rc_ if use_c else None, rt if use_t else None, 4.0, 0)[0]
        (ref * coef).sum().backward()
        dn, dw, db, dc, dt = [leaf(t.to(DEV)) for t in (n, w, b, c, th)]
        got, _ = Fn.HapkeFunction.apply(l.to(DEV), v.to(DEV), dn, dw, db, dc if use_c else None, dt if use_t else None, 4.0, 0)
        (got * coef.to(DEV)).sum().backward()
        pairs = [("w", dw.grad, rw.grad), ("b", db.grad, rb.grad)]
        if use_c:
            pairs.append(("c", dc.grad, rc_.grad))
        check(pairs, got, ref, 5e-4)
    else:
        rough = 0.05 + 0.95 * torch.rand(N, 1, generator=g)
        rn, rw, rr = leaf(n), leaf(w), leaf(rough)
        ref = OB.microfacet(l, v, rn, rw, rr, 0.04)[1]
        (ref * coef).sum().backward()
        dn, dw, dr = [leaf(t.to(DEV)) for t in (n, w, rough)]
        got, _ = Fn.MicrofacetFunction.apply(l.to(DEV), v.to(DEV), dn, dw, dr, 0.04)
        (got * coef.to(DEV)).sum().backward()
        check([("w", dw.grad, rw.grad), ("rough", dr.grad, rr.grad), ("n", dn.grad, rn.grad)], got, ref, 2e-4)


@pytest.mark.parametrize("seed", list(range(24 * _K)))
def test_random_render_rays_half_modes_track_fp32(seed):
    """render_rays (test mode: the evaluation path - inference kernel variants, sigma-only pass, analytic normals without a
    backward stash) on random configurations in a 16-bit mode against the HIP fp32 mode on the same draws: finite everywhere
    the fp32 result is, depths and weights close, pixel values within the half bounds on all but a few rays."""
    from test_gpu_parity import build_model, make_args, Replay, diag, HALF_BOUNDS
    from brdf_nerf_amd import render_rays
    rng = np.random.default_rng(29000 + seed)
    cfg = draw_config(rng)
    S, G = int(rng.choice([8, 16, 32, 64])), int(rng.choice([8, 16, 64]))
    kw = dict(vars(cfg))
    kw.update(n_samples=S, guided_samples=G)
    brdf = bool(cfg.roughness or cfg.RPV or cfg.b)
    gsam_only = bool(rng.random() < 0.3)
    if brdf and gsam_only and rng.random() < 0.5:
        kw["sun_v"] = "analystic"
    cfg = FieldConfig(**kw)
    dtype = "bf16" if seed % 2 == 0 else "fp16"
    R = int(rng.integers(1, 300))
    flags = dict(apply_brdf=brdf and bool(rng.random() < 0.85), apply_theta=bool(rng.random() < 0.7), cos_irra_on=bool(rng.random() < 0.6),
                 gsam_only=gsam_only)
    if cfg.sun_v == "analystic":
        flags["apply_brdf"] = True
    g = torch.Generator().manual_seed(seed)
    rays = _sat_rays(R, g).to(DEV)
    ts = torch.randint(0, 6, (R,), generator=g).to(DEV) if cfg.beta else None
    emb = torch.randn(6, cfg.t_dim, generator=g) if cfg.beta else None
    res, draws = {}, None
    for dt in ("fp32", dtype):
        models = {"coarse": build_model(cfg, 95 + seed, dt)}
        if cfg.beta:
            models["t"] = torch.nn.Embedding(6, cfg.t_dim).to(DEV)
            with torch.no_grad():
                models["t"].weight.copy_(emb)
        with torch.no_grad():
            if draws is None:
                torch.manual_seed(9)
                with _Record() as rec:
                    res[dt], _ = render_rays(models, make_args(cfg, dt), rays, ts, mode="test", **flags)
                draws = rec.log
            else:
                with Replay(list(draws)) as rp:
                    res[dt], _ = render_rays(models, make_args(cfg, dt), rays, ts, mode="test", **flags)
                    assert rp.draws == []
    tag = (f"fuzz-render16 {seed} {dtype}: F={cfg.feat} L={cfg.layers} siren={int(cfg.siren)} normal={cfg.normal} sun_v={cfg.sun_v} "
           f"beta={int(cfg.beta)} viewdir={cfg.input_viewdir} R={R} S={S} G={G} {flags}")
    a, b = res[dtype], res["fp32"]
    assert set(a) == set(b), tag
    for k, v in b.items():
        if v.dtype.is_floating_point and k != "hpk_scl_coarse":
            assert bool(torch.isfinite(a[k][torch.isfinite(v)]).all()), f"{tag}: {k} not finite"
    bd = HALF_BOUNDS[dtype]
    e_rgb = (a["rgb_coarse"] - b["rgb_coarse"]).abs().amax(-1)
    e_dep = (a["depth_coarse"] - b["depth_coarse"]).abs()
    diag(f"{tag}: rgb max {float(e_rgb.max()):.2e} median {float(e_rgb.median()):.2e}; depth max {float(e_dep.max()):.2e}")
    # random (untrained) networks put some rays at ill-conditioned spots (guided samples moved by a rounding difference in pass 1,
    # BRDFs at grazing angles): the bulk must track, a tenth of the rays may stray
    lim_rgb, lim_dep = 4 * bd["rgb"], 0.05
    assert int((e_rgb > lim_rgb).sum()) <= max(1, R // 10), f"{tag}: {int((e_rgb > lim_rgb).sum())} of {R} pixels off by more than {lim_rgb}"
    assert int((e_dep > lim_dep).sum()) <= max(1, R // 10), f"{tag}: {int((e_dep > lim_dep).sum())} of {R} depths off by more than {lim_dep}"


@pytest.mark.parametrize("seed", list(range(24 * _K)))
def test_random_lean_step_against_general_step(seed):
    """Round 3: the launch-lean step (in-kernel draws, merged-set compositing through the sort index, ray-level shading + loss
    kernel, fold / unfold / multi-group Adam kernels; the third step replayed from its HIP graph) on random configurations
    against the general step (held to the autograd path by test_random_fused_step_against_autograd_path) fed with the Philox
    streams' draws as arrays: loss, rgb and the flat gradient, three steps resynchronised one by one.  Random width / depth /
    Siren or ReLU / heads / normals / funcH / shell_hapke, depth priors, hard-surface and normal-regulariser lambdas, ragged ray and sample counts."""
    import brdf_nerf_amd
    from test_gpu_parity import build_model, make_args, Replay, diag
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.trainer import FusedTrainer
    rng = np.random.default_rng(23000 + seed)
    cfg = draw_config(rng)
    S, G = int(rng.choice([8, 16, 24, 32])), int(rng.choice([8, 16, 32]))
    kw = dict(vars(cfg))
    kw.update(feat=int(rng.choice([64, 128, 192])), n_samples=S, guided_samples=G, beta=False)
    brdf = bool(cfg.roughness or cfg.RPV or cfg.b)
    if not brdf and cfg.normal != "none" and rng.random() < 0.5:
        kw["shell_hapke"] = int(rng.integers(1, 4))          # Hapke shell without BRDF heads (spsbrdfnerf.py:320,348,383)
    cfg = FieldConfig(**kw)
    args = make_args(cfg)
    R = int(rng.integers(8, 120))
    flags = dict(apply_brdf=brdf and bool(rng.random() < 0.85), apply_theta=bool(rng.random() < 0.7), cos_irra_on=bool(rng.random() < 0.6))
    hs = float(rng.choice([0.0, 0.3]))
    lam = dict(hs_lambda=hs, nr_reg_an_lambda=float(rng.choice([0.0, 0.2])), nr_reg_lr_lambda=float(rng.choice([0.0, 0.1])))
    g = torch.Generator().manual_seed(seed)
    rays = _sat_rays(R, g).to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    prior = {}
    if rng.random() < 0.6:
        prior = dict(valid_depth=(torch.rand(R, generator=g) < 0.6).float().to(DEV),
                     depths=torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV),
                     depth_std=(0.03 * torch.rand(R, generator=g)).to(DEV))
    if rng.random() < 0.3:                                   # round 4: the gsam_only stage is a lean step too
        flags["gsam_only"] = True
    # round 4: --noise_std on the lean step (in-kernel normal draws); a generator of its own keeps the other draws of a seed as they were
    rng4 = np.random.default_rng(91000 + seed)
    noise_std = 0.4 if rng4.random() < 0.3 else 0.0
    args.noise_std = noise_std
    if cfg.normal == "analystic_learned" and rng4.random() < 0.6:      # NormalLoss between the two normal fields (nr_spv_type 1)
        lam["nr_spv_lambda"] = 0.3
    sun_pass = bool(flags.get("gsam_only") and flags["apply_brdf"] and noise_std == 0.0 and rng4.random() < 0.5)
    # one BRDF per sample: a lean step too - with the regularisers and with the sun pass's per-sample irradiance since round 5
    multi = bool(flags["apply_brdf"] and rng4.random() < 0.6)
    if sun_pass or multi:                                               # the sun-visibility pass of the gsam_only stage
        cfg = FieldConfig(**dict(vars(cfg), sun_v="analystic" if sun_pass else cfg.sun_v, MultiBRDF=multi))
        args = make_args(cfg)
        args.noise_std = noise_std
    tag = (f"fuzz-lean {seed}: noise={noise_std} sun={int(sun_pass)} multi={int(cfg.MultiBRDF)} F={cfg.feat} L={cfg.layers} siren={int(cfg.siren)} pe={int(cfg.mapping)} normal={cfg.normal} viewdir={cfg.input_viewdir} "
           f"heads={cfg.brdf_head_names(flags['apply_brdf'], flags['apply_theta'])} funcH={cfg.funcH} shell={cfg.shell_hapke} R={R} S={S} G={G} "
           f"prior={bool(prior)} {lam} {flags}")
    prev = brdf_nerf_amd.set_deterministic(True)
    try:
        torch.manual_seed(11)
        ta = FusedTrainer(build_model(cfg, 70 + seed), args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
        tb = FusedTrainer(build_model(cfg, 70 + seed), args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
        ta.lean = False
        tb.graph_after = 1
        tb.keep_grads = True
        nr_an = cfg.normal in ("analystic", "analystic_learned")
        kink = nr_an and not cfg.siren
        worst = 0.0
        for step in range(3):
            tb.flat_param.copy_(ta.flat_param)
            tb.exp_avg.copy_(ta.exp_avg)
            tb.exp_avg_sq.copy_(ta.exp_avg_sq)
            n_t = R * G
            draws = [Fn.rng_uniform(tb.state, 1, R * S).view(R, S)]
            if noise_std:
                draws.append(Fn.rng_normal(tb.state, 4, R * S).view(R, S))
            if sun_pass:                # the general step's order: the sun pass's depths and its (unused) normals come before u
                draws += [Fn.rng_uniform(tb.state, 6, R * G).view(R, G), torch.zeros(R, G, device=DEV)]
            draws.append(Fn.rng_uniform(tb.state, 2, R * G).view(R, G))
            if prior:
                draws.append(Fn.rng_uniform(tb.state, 3, n_t).view(R, G))
            if noise_std:
                S2 = G if flags.get("gsam_only") else S + G
                draws.append(Fn.rng_normal(tb.state, 5, R * S2).view(R, S2))
            with Replay(draws) as rp:
                la, rgb_a = ta.step(rays, rgbs, **prior, **flags)
                assert rp.draws == [], tag
            lb, rgb_b = tb.step(rays, rgbs, **prior, **flags)
            la, lb = float(la), float(lb)
            if not (la == la):                      # a NaN ray: the general step reports the reference's NaN loss, the lean step leaves the ray out
                continue
            assert abs(la - lb) <= 5e-5 * abs(la) + 1e-7, (tag, step, la, lb)
            ga, gb = ta.flat_grad, tb.flat_grad
            scale = float(ga.abs().max())
            e = float((ga - gb).abs().max()) / max(scale, 1e-30)
            worst = max(worst, e)
            # (analytic normals / GGX amplify the 1e-7 differences of the two ray-level evaluations: test_gpu_lean.py; a ReLU
            # network's analytic normal is piecewise constant - a pre-activation within rounding of 0 may take the other branch)
            # (noise: an fma against mul + add; Hapke's shadow-hiding / roughness terms amplify the 1e-7 differences of the two
            # ray-level evaluations: 1.8e-4 at seed 34 of the doubled fuzz)
            tol = 2e-2 if kink else (1e-3 if (nr_an or cfg.roughness) else (5e-4 if (noise_std or cfg.b) else 1e-4))
            assert e <= tol, (tag, step, e)
        assert len(tb._graphs) == 1, tag
        diag(f"{tag}: worst flat-gradient difference {worst:.2e} of the largest entry")
    finally:
        brdf_nerf_amd.set_deterministic(prev)
