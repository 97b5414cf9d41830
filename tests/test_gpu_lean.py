"""GPU tests of the launch-lean fused step (round 3): the in-kernel draws, the fused per-ray kernels against the validated
unfused ones (which are held to the reference's goldens in test_gpu_parity.py) and against the reference's render golden,
the fold / unfold and multi-group Adam kernels against torch, and the lean FusedTrainer step - eager and replayed from a HIP
graph - against the general path fed with the same draws."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle.config import FieldConfig  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sat_rays(R, g):
    o = torch.cat([torch.rand(R, 2, generator=g) * 2 - 1, 1.0 + 0.02 * torch.rand(R, 1, generator=g)], -1)
    d = torch.tensor([0.15, 0.2, -0.96]).expand(R, 3)
    sun = torch.tensor([-0.37, 0.44, 0.82]).expand(R, 3)
    return torch.cat([o, d, torch.zeros(R, 1), torch.full((R, 1), 2.0), sun], -1).contiguous()


def _field_like(R, S, C, g):
    """Random per-sample field outputs with a realistic density channel (many zeros, a few large values)."""
    out = torch.rand(R, S, C, generator=g)
    sig = torch.rand(R, S, generator=g)
    out[..., 3] = torch.where(sig < 0.5, torch.zeros_like(sig), 40.0 * (sig - 0.5) ** 2)
    return out.to(DEV)


# ------------------------------------------------------------------------------------------------ draws
def test_in_kernel_draws_are_uniform_and_step_dependent():
    from brdf_nerf_amd import functions as Fn
    st = Fn.new_step_state(DEV, 1234, 5e-4)
    n = 1 << 20
    u = Fn.rng_uniform(st, 1, n)
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0
    assert abs(float(u.mean()) - 0.5) < 2e-3 and abs(float(u.var()) - 1 / 12) < 1e-3
    hist = torch.histc(u, bins=64, min=0, max=1)
    assert float((hist / (n / 64) - 1).abs().max()) < 0.03
    # lag-1 correlation, stream and step independence
    assert abs(float(((u[1:] - 0.5) * (u[:-1] - 0.5)).mean()) * 12) < 5e-3
    u2 = Fn.rng_uniform(st, 2, n)
    assert abs(float(((u - 0.5) * (u2 - 0.5)).mean()) * 12) < 5e-3
    st[1] += 1
    u3 = Fn.rng_uniform(st, 1, n)
    assert abs(float(((u - 0.5) * (u3 - 0.5)).mean()) * 12) < 5e-3 and not torch.equal(u, u3)
    st[1] -= 1
    assert torch.equal(Fn.rng_uniform(st, 1, n), u)                       # a pure function of (seed, step, stream, index)


def test_in_kernel_normal_draws_are_standard_normal():
    """bn_rng_normal (the --noise_std draws of the launch-lean step): Box-Muller on the Philox pairs of a stream."""
    from brdf_nerf_amd import functions as Fn
    st = Fn.new_step_state(DEV, 4321, 5e-4)
    n = 1 << 20
    x = Fn.rng_normal(st, 4, n).double()
    assert torch.isfinite(x).all()
    assert abs(float(x.mean())) < 4e-3 and abs(float(x.var()) - 1.0) < 6e-3
    assert abs(float((x ** 3).mean())) < 2e-2 and abs(float((x ** 4).mean()) - 3.0) < 6e-2
    # tails: P(|x| > 2) = 0.0455, P(|x| > 3) = 0.0027
    assert abs(float((x.abs() > 2).double().mean()) - 0.0455) < 1.5e-3 and abs(float((x.abs() > 3).double().mean()) - 0.0027) < 4e-4
    assert abs(float((x[1:] * x[:-1]).mean())) < 5e-3                      # the two normals of a Philox block, and neighbours
    y = Fn.rng_normal(st, 5, n).double()
    assert abs(float((x * y).mean())) < 5e-3                               # streams
    u = Fn.rng_uniform(st, 4, n).double()
    assert abs(float((x * (u - 0.5)).mean())) < 5e-3
    assert torch.equal(Fn.rng_normal(st, 4, n).double(), x)                # a pure function of (seed, step, stream, index)


def test_stratified_z_rng_is_stratified_z_of_the_streams_draws():
    from brdf_nerf_amd import functions as Fn
    g = torch.Generator().manual_seed(0)
    R, S = 97, 24
    rays = _sat_rays(R, g).to(DEV)
    st = Fn.new_step_state(DEV, 77, 5e-4)
    z = Fn.stratified_z_rng(rays, S, st)
    u = Fn.rng_uniform(st, 1, R * S).view(R, S)
    assert torch.equal(z, Fn.stratified_z(rays[:, 6:7], rays[:, 7:8], u))


# ------------------------------------------------------------------------------------------------ fused per-ray kernels
@pytest.mark.parametrize("R,S,G,prior", [(37, 16, 16, True), (64, 64, 64, True), (5, 40, 24, False), (130, 128, 64, True)])
def test_composite_guided_is_composite_then_guided(R, S, G, prior):
    """bn_composite_guided (one launch, pass-1 weights in LDS, in-kernel draws) against bn_composite_forward +
    bn_guided_samples_nf on the stream's draws as arrays: identical depths and sort indices."""
    from brdf_nerf_amd import functions as Fn
    g = torch.Generator().manual_seed(R + S)
    rays = _sat_rays(R, g).to(DEV)
    st = Fn.new_step_state(DEV, 5, 5e-4)
    z = Fn.stratified_z_rng(rays, S, st)
    out1 = _field_like(R, S, 7, g)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV) if prior else None
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV) if prior else None
    dstd = (0.02 * torch.rand(R, generator=g)).to(DEV) if prior else None
    z2, z_all, idx, w1, d1 = Fn.composite_guided(z, out1, G, rays[0, 6:8], 3.0, valid, None if depths is None else depths[:, 0], dstd,
                                                 state=st, want_pass1=True)
    _, _, w_ref, d_ref, _ = Fn.composite_forward_raw(z, out1)
    assert torch.equal(w1, w_ref) and torch.equal(d1, d_ref)
    u = Fn.rng_uniform(st, 2, R * G).view(R, G)
    u_t = Fn.rng_uniform(st, 3, R * G).view(R, G) if prior else None
    z2r, z_allr, idxr = Fn.guided_samples(z, w_ref, d_ref, u, rays[0, 6:8], None, 3.0, valid,
                                          None if depths is None else depths[:, 0].contiguous(), dstd, u_t, None)
    assert torch.equal(z2, z2r) and torch.equal(z_all, z_allr) and torch.equal(idx, idxr)
    # the same with the draws handed over as arrays
    z2b, z_allb, idxb = Fn.composite_guided(z, out1, G, rays[0, 6:8], 3.0, valid, None if depths is None else depths[:, 0], dstd,
                                            u=u, u_target=u_t)
    assert torch.equal(z2, z2b) and torch.equal(idx, idxb) and torch.equal(z_all, z_allb)


def _merged_reference(z_all, idx, out1, out2):
    C = out1.shape[-1]
    return torch.cat([out1, out2], 1).gather(1, idx.unsqueeze(-1).expand(-1, -1, C)).contiguous()


@pytest.mark.parametrize("R,S,G,C", [(33, 16, 16, 4), (64, 64, 64, 4), (21, 24, 8, 7), (50, 64, 64, 13), (9, 128, 64, 20)])
def test_merged_composite_is_gather_then_composite(R, S, G, C):
    """The merged-set compositing through the sort index (forward and backward, gradient rows in the source layouts) against
    cat + gather + bn_composite_forward / backward + scatter."""
    from brdf_nerf_amd import functions as Fn
    g = torch.Generator().manual_seed(R * 3 + C)
    out1, out2 = _field_like(R, S, C, g), _field_like(R, G, C, g)
    z = torch.sort(torch.rand(R, S, generator=g) * 2, -1)[0]
    z2 = torch.sort(torch.rand(R, G, generator=g) * 2, -1)[0]
    z_all, idx = torch.sort(torch.cat([z, z2], -1), dim=-1, stable=True)
    z_all, idx = z_all.to(DEV).contiguous(), idx.to(DEV).contiguous()
    merged = _merged_reference(z_all, idx, out1, out2)
    a, t, w, d, acc = Fn.composite_forward_raw(z_all, merged)
    o = Fn.merged_composite_forward(z_all, idx, out1, out2, want=("alphas", "trans", "weights", "depth", "acc", "wsum"))
    for k, ref in (("alphas", a), ("trans", t), ("weights", w)):
        assert torch.equal(o[k], ref), k
    assert float((o["depth"] - d).abs().max()) <= 2e-6 and float((o["wsum"] - w.sum(-1)).abs().max()) <= 2e-6
    keep = [c for c in range(C) if c != 3]
    assert float((o["acc"][:, keep] - acc[:, keep]).abs().max()) <= 2e-6
    d_w = torch.randn(R, S + G, generator=g).to(DEV)
    d_d = torch.randn(R, generator=g).to(DEV)
    d_acc = torch.randn(R, C, generator=g).to(DEV)
    d_acc[:, 3] = 0
    ref = Fn.composite_backward_raw(z_all, merged, d_w, d_d, d_acc)
    d_cat = torch.zeros(R, S + G, C, device=DEV).scatter_(1, idx.unsqueeze(-1).expand(-1, -1, C), ref)
    d1, d2 = torch.empty(R, S, C, device=DEV), torch.empty(R, G, C, device=DEV)
    cnt = torch.zeros(2, dtype=torch.int64, device=DEV)
    Fn.merged_composite_backward(z_all, idx, out1, out2, d_w, d_d, d_acc, d1, d2, nonfinite=cnt)
    scale = float(ref.abs().max())
    assert float((d1 - d_cat[:, :S]).abs().max()) <= 2e-6 * scale and float((d2 - d_cat[:, S:]).abs().max()) <= 2e-6 * scale
    assert cnt.tolist() == [0, 0]
    # a per-ray gradient of sum_s w is a constant added to every d_weights entry
    d_ws = torch.randn(R, generator=g).to(DEV)
    d1b, d2b = torch.empty_like(d1), torch.empty_like(d2)
    Fn.merged_composite_backward(z_all, idx, out1, out2, d_w, d_d, d_acc, d1b, d2b, d_wsum=d_ws)
    ref2 = Fn.composite_backward_raw(z_all, merged, d_w + d_ws[:, None], d_d, d_acc)
    d_cat2 = torch.zeros(R, S + G, C, device=DEV).scatter_(1, idx.unsqueeze(-1).expand(-1, -1, C), ref2)
    assert float((d1b - d_cat2[:, :S]).abs().max()) <= 2e-6 * scale and float((d2b - d_cat2[:, S:]).abs().max()) <= 2e-6 * scale
    # non-finite gradient elements are dropped and counted
    d_w_bad = d_w.clone()
    d_w_bad[0, 0] = float("nan")
    Fn.merged_composite_backward(z_all, idx, out1, out2, d_w_bad, d_d, d_acc, d1b, d2b, nonfinite=cnt)
    assert bool(torch.isfinite(d1b).all()) and bool(torch.isfinite(d2b).all()) and int(cnt.sum()) > 0
    # identity index: one source block
    o1 = Fn.merged_composite_forward(z_all, None, merged, None, want=("weights", "depth"))
    assert torch.equal(o1["weights"], w)


@pytest.mark.parametrize("R,S,G,prior", [(33, 16, 16, True), (64, 64, 64, True), (40, 24, 8, False)])
def test_lambert_tail_is_composite_loss_backward(R, S, G, prior):
    """bn_lambert_tail (one launch) against bn_composite_forward + bn_lambert_loss (held to the reference's SNerfLoss / DepthLoss
    golden in test_gpu_parity.py) + bn_composite_backward on the gathered merged set."""
    from brdf_nerf_amd import functions as Fn
    g = torch.Generator().manual_seed(R + 11)
    C = 4
    out1, out2 = _field_like(R, S, C, g), _field_like(R, G, C, g)
    z = torch.sort(torch.rand(R, S, generator=g) * 2, -1)[0]
    z2 = torch.sort(torch.rand(R, G, generator=g) * 2, -1)[0]
    z_all, idx = torch.sort(torch.cat([z, z2], -1), dim=-1, stable=True)
    z_all, idx = z_all.to(DEV).contiguous(), idx.to(DEV).contiguous()
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV) if prior else None
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV) if prior else None
    dstd = (0.05 * torch.rand(R, generator=g)).to(DEV) if prior else None
    merged = _merged_reference(z_all, idx, out1, out2)
    _, _, w, d, acc = Fn.composite_forward_raw(z_all, merged)
    kw = dict(valid_depth=valid, target_depth=depths[:, 0], target_weight=depths[:, 1], target_std=dstd, lambda_ds=10.0) if prior else {}
    loss, rgb, d_acc, d_depth, d_w = Fn.lambert_loss(acc, w, z_all, d, rgbs, 0.001, 1.0, **kw)
    d_acc = d_acc.contiguous()
    d_acc[:, 3] = 0
    ref = Fn.composite_backward_raw(z_all, merged, d_w, d_depth, d_acc)
    d_cat = torch.zeros(R, S + G, C, device=DEV).scatter_(1, idx.unsqueeze(-1).expand(-1, -1, C), ref)
    d1, d2 = torch.empty(R, S, C, device=DEV), torch.empty(R, G, C, device=DEV)
    ray_loss, lacc, rgb2 = torch.empty(R, device=DEV), torch.zeros(16, device=DEV), torch.empty(R, 3, device=DEV)
    w2, dep2 = torch.empty(R, S + G, device=DEV), torch.empty(R, device=DEV)
    Fn.lambert_tail(z_all, idx, out1, out2, rgbs, 0.001, 1.0, d1, d2, ray_loss=ray_loss, loss_acc=lacc, rgb=rgb2, weights=w2, depth=dep2,
                    **kw)
    assert torch.equal(w2, w) and float((dep2 - d).abs().max()) <= 2e-6
    assert float((rgb2 - rgb).abs().max()) <= 2e-6
    assert abs(float(ray_loss.sum()) - float(loss)) <= 1e-6 * abs(float(loss)) + 1e-9
    # ray r's term went to partial sum r % 16
    want_part = torch.zeros(16, device=DEV).index_add_(0, torch.arange(R, device=DEV) % 16, ray_loss)
    assert float((lacc - want_part).abs().max()) <= 2e-6 * abs(float(loss)) + 1e-9
    scale = float(ref.abs().max())
    assert float((d1 - d_cat[:, :S]).abs().max()) <= 5e-6 * scale and float((d2 - d_cat[:, S:]).abs().max()) <= 5e-6 * scale


def test_lambert_tail_forward_against_reference_render_golden():
    """The forward half of bn_lambert_tail on the REFERENCE's own per-sample outputs (tests/golden/render_lambert_train.npz:
    sigmas, albedo, z_vals of the merged S+G set, produced by rendering.py:168-291 + models/spsbrdfnerf.py:198-282): its
    weights, depth and shaded rgb are the reference's."""
    from test_gpu_parity import load_golden
    from brdf_nerf_amd import functions as Fn
    g = load_golden("render_lambert_train")
    t = lambda k: torch.from_numpy(g["out/" + k]).to(DEV)
    z_all, sig, alb = t("z_vals_coarse"), t("sigmas_coarse"), t("albedo_coarse")
    R, S2 = z_all.shape
    out = torch.cat([alb, sig.reshape(R, S2, 1)], -1).contiguous()
    rgbs = torch.from_numpy(g["tgt/rgbs"]).to(DEV)
    d1 = torch.empty(R, S2, 4, device=DEV)
    w, dep, rgb = torch.empty(R, S2, device=DEV), torch.empty(R, device=DEV), torch.empty(R, 3, device=DEV)
    Fn.lambert_tail(z_all.contiguous(), None, out, None, rgbs, 0.001, 1.0, d1, None, rgb=rgb, weights=w, depth=dep)
    assert float((w - t("weights_coarse")).abs().max()) <= 1e-5
    assert float((dep - t("depth_coarse")).abs().max()) <= 1e-5 * 2
    assert float((rgb - t("rgb_coarse")).abs().max()) <= 2e-5


def test_merged_composite_normal_regulariser_against_autograd():
    """NormalRegLoss (metrics.py:179-216; losses.normal_reg_loss is held to the reference's regulariser golden) inside the
    merged-set compositing: the forward's per-ray terms and the backward's gradient rows against torch autograd through
    bn_composite_forward on the gathered set."""
    from brdf_nerf_amd import functions as Fn, losses
    g = torch.Generator().manual_seed(4)
    R, S, G, C = 37, 16, 8, 13
    out1, out2 = _field_like(R, S, C, g), _field_like(R, G, C, g)
    for o_ in (out1, out2):
        o_[..., 4:10] = torch.randn(o_[..., 4:10].shape, generator=g).to(DEV)          # two normal fields, both signs of n . view
    z = torch.sort(torch.rand(R, S, generator=g) * 2, -1)[0]
    z2 = torch.sort(torch.rand(R, G, generator=g) * 2, -1)[0]
    z_all, idx = torch.sort(torch.cat([z, z2], -1), dim=-1, stable=True)
    z_all, idx = z_all.to(DEV).contiguous(), idx.to(DEV).contiguous()
    rays = _sat_rays(R, g).to(DEV)
    lam_an, lam_lr = 0.2, 0.1
    nreg = Fn.normal_reg(rays[:, 3:6], 4, 7, lam_an, lam_lr)
    o = Fn.merged_composite_forward(z_all, idx, out1, out2, want=("weights", "depth", "acc"), nreg=nreg)
    merged = _merged_reference(z_all, idx, out1, out2).requires_grad_(True)
    _, _, w, _, _ = Fn.composite(z_all, merged, None, 0.0)
    view = -rays[:, 3:6]
    per_ray = lam_an * (w * torch.clamp_max((merged[..., 4:7] * view[:, None, :]).sum(-1), 0.0) ** 2).sum(-1) + \
        lam_lr * (w * torch.clamp_max((merged[..., 7:10] * view[:, None, :]).sum(-1), 0.0) ** 2).sum(-1)
    total = losses.normal_reg_loss(merged[..., 4:7], w, view, lam_an)[0] + losses.normal_reg_loss(merged[..., 7:10], w, view, lam_lr)[0]
    assert abs(float(per_ray.sum()) - float(total)) <= 1e-5 * abs(float(total))
    assert float((o["reg"] - per_ray.detach()).abs().max()) <= 2e-6 * float(per_ray.abs().max())
    (ref,) = torch.autograd.grad(total, merged)
    d_cat = torch.zeros(R, S + G, C, device=DEV).scatter_(1, idx.unsqueeze(-1).expand(-1, -1, C), ref)
    d1, d2 = torch.empty(R, S, C, device=DEV), torch.empty(R, G, C, device=DEV)
    zero_acc = torch.zeros(R, C, device=DEV)
    Fn.merged_composite_backward(z_all, idx, out1, out2, None, None, zero_acc, d1, d2, nreg=nreg)
    scale = float(ref.abs().max())
    assert float((d1 - d_cat[:, :S]).abs().max()) <= 5e-6 * scale and float((d2 - d_cat[:, S:]).abs().max()) <= 5e-6 * scale


# ------------------------------------------------------------------------------------------------ ray-level shading + losses
_SHADE_CFGS = {
    "rpv111_nlr": (dict(funcM=1, funcF=1, funcH=1, normal="learned"), True),
    "rpv_m1f1h2_nan": (dict(funcM=1, funcF=1, funcH=2, normal="analystic"), True),
    "rpv_f1_nanlr": (dict(funcF=1, normal="analystic_learned"), True),
    "hapke_bct": (dict(b=1, c=1, theta=1, normal="learned"), True),
    "hapke_b": (dict(b=1, normal="analystic"), True),
    "hapke_shell3_nobrdf": (dict(shell_hapke=3, normal="learned"), False),
    "microfacet": (dict(roughness=True, normal="analystic"), True),
    "normal_only": (dict(normal="learned"), False),
}


@pytest.mark.parametrize("prior", [True, False])
@pytest.mark.parametrize("name", list(_SHADE_CFGS))
def test_ray_shade_loss_kernel_against_autograd_of_the_torch_statement(name, prior):
    """bn_ray_shade_loss against rendering.shade_ray (the torch-level ray shading, held to the reference's render goldens through
    shade() in test_gpu_parity.py) + losses.snerf_loss / depth_loss / hard_surface_loss (held to metrics.py goldens), differentiated
    with autograd: rgb, loss, d acc, d wsum, d depth - and bn_merged_composite_backward's HardSurfaceLoss term against autograd of
    the per-sample weights."""
    from test_gpu_parity import build_model, make_args
    from brdf_nerf_amd import functions as Fn, losses
    from brdf_nerf_amd.rendering import shade_ray, shade_desc
    kw, brdf = _SHADE_CFGS[name]
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16, **kw)
    args = make_args(cfg)
    model = build_model(cfg, 5)
    nr_lr, nr_an = cfg.normal in ("learned", "analystic_learned"), cfg.normal in ("analystic", "analystic_learned")
    spec = model.spec(brdf, brdf, nr_lr, nr_an, beta=False)
    C = spec.out_channels
    R, S = 301, 24
    g = torch.Generator().manual_seed(len(name) + 7 * prior)
    rays = _sat_rays(R, g).to(DEV)
    rays_d, sun_d = rays[:, 3:6], rays[:, 8:11]
    # a plausible merged set: weights of a real compositing, channel rows in the heads' ranges
    out = _field_like(R, S, C, g)
    z = torch.sort(0.5 + torch.rand(R, S, generator=g), -1)[0].to(DEV).contiguous()
    o = Fn.merged_composite_forward(z, None, out, None, want=("weights", "depth", "acc", "wsum", "var"))
    if spec.normal_an or spec.normal_lr:       # composited normals roughly facing the camera, a few degenerate (zero) ones
        c0 = spec.ch_normal_lr if spec.normal_lr else spec.ch_normal_an
        nrm = -rays_d + 0.5 * torch.randn(R, 3, generator=g).to(DEV)
        nrm[::37] = 0.0
        o["acc"][:, c0:c0 + 3] = nrm * o["wsum"][:, None]
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV)
    depths = torch.stack([o["depth"].cpu() + 0.1 * torch.randn(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV)
    dstd = (0.05 * torch.rand(R, generator=g)).to(DEV)
    lam_rgb, lam_ds, lam_hs = 0.7, 10.0 if prior else 0.0, 0.3
    cos_on = brdf or name == "normal_only"
    # ---- torch statement
    acc_l, depth_l, w_l = o["acc"].clone().requires_grad_(True), o["depth"].clone().requires_grad_(True), o["weights"].clone().requires_grad_(True)
    res, _ = shade_ray(model, args, spec, z, w_l, depth_l, acc_l, rays_d, sun_d, brdf, cos_on)
    loss = losses.snerf_loss(res["rgb"], rgbs, lam_rgb)
    if prior:
        loss = loss + losses.depth_loss(z, depth_l, w_l, depths[:, 0], depths[:, 1], valid, dstd, lam_ds, False)
    loss = loss + losses.hard_surface_loss(z, depth_l, w_l, lam_hs)
    d_acc, d_depth, d_w = torch.autograd.grad(loss, [acc_l, depth_l, w_l])
    # ---- kernel
    desc = shade_desc(model, args, spec, brdf, cos_on, lam_rgb, lam_ds, lam_hs, False)
    ray_loss, lacc = torch.empty(R, device=DEV), torch.zeros(8, device=DEV)
    k = Fn.ray_shade_loss(desc, o["acc"], o["wsum"], o["depth"], o["var"], rays_d, sun_d, rgbs, None,
                          valid if prior else None, depths[:, 0] if prior else None, depths[:, 1] if prior else None,
                          dstd if prior else None, ray_loss=ray_loss, loss_acc=lacc)
    tol_rgb = 1e-4 if name == "microfacet" else 2e-6
    assert float((k["rgb"] - res["rgb"]).abs().max()) <= tol_rgb, name
    assert abs(float(ray_loss.sum()) - float(loss)) <= 2e-5 * abs(float(loss)), (name, float(ray_loss.sum()), float(loss))
    assert abs(float(lacc.sum()) - float(loss)) <= 2e-5 * abs(float(loss))
    # d loss / d w_s = (d wsum + hs (z - depth)^2) + [shading terms through acc: not part of w's own gradient here]: the torch
    # statement reads weights in wsum (albedo padding), the depth-loss gate (no gradient) and HardSurfaceLoss
    want_w = k["d_wsum"][:, None] + (lam_hs / R) * (z - o["depth"][:, None]) ** 2
    assert float((want_w - d_w).abs().max()) <= 2e-5 * float(d_w.abs().max()) + 1e-12, name
    d_acc = d_acc.clone()
    d_acc[:, 3] = 0
    bad_k, bad_t = ~torch.isfinite(k["d_acc"]).all(-1), ~torch.isfinite(d_acc).all(-1)
    assert torch.equal(bad_k, bad_t), (name, bad_k.nonzero().flatten().tolist(), bad_t.nonzero().flatten().tolist())
    # (Hapke with theta: a degenerate normal gives NaN gradients on BOTH sides, in the same rays - the step's sanitize_grads
    # zeroes and counts them)
    assert int(bad_k.sum()) <= R // 20
    ok = ~bad_k
    sc = float(d_acc[ok].abs().max())
    err = float((k["d_acc"][ok] - d_acc[ok]).abs().max()) / sc
    # (the chain through a degenerate normal / a GGX lobe amplifies last-bit differences of the forward)
    assert err <= (2e-3 if name == "microfacet" else 2e-4), (name, err)
    assert float((k["d_depth"] - d_depth).abs().max()) <= 2e-5 * float(d_depth.abs().max()) + 1e-12, name
    # ---- the composite backward adds the per-sample HardSurfaceLoss term
    d1a, d1b = torch.empty(R, S, C, device=DEV), torch.empty(R, S, C, device=DEV)
    fin = {key: torch.nan_to_num(v, 0.0, 0.0, 0.0) for key, v in k.items()}
    Fn.merged_composite_backward(z, None, out, None, None, fin["d_depth"], fin["d_acc"], d1a, None, d_wsum=fin["d_wsum"],
                                 hs_scale=lam_hs / R, depth=o["depth"])
    want_w = fin["d_wsum"][:, None] + (lam_hs / R) * (z - o["depth"][:, None]) ** 2
    Fn.merged_composite_backward(z, None, out, None, want_w.contiguous(), fin["d_depth"], fin["d_acc"], d1b, None)
    assert float((d1a - d1b).abs().max()) <= 2e-6 * float(d1b.abs().max())


def test_ray_shade_loss_leaves_non_finite_rays_out_when_asked():
    """A ray whose shaded value is NaN (here: NaN composited sums) poisons loss and gradients as upstream - unless the caller
    passes the non-finite counters (FusedTrainer.sanitize_grads): then the ray contributes nothing and is counted."""
    from test_gpu_parity import build_model, make_args
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.rendering import shade_desc
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16, b=1, c=1, theta=1, normal="learned")
    args, model = make_args(cfg), build_model(cfg, 5)
    spec = model.spec(True, True, True, False, beta=False)
    R, S, C = 130, 16, spec.out_channels
    g = torch.Generator().manual_seed(0)
    rays = _sat_rays(R, g).to(DEV)
    z = torch.sort(0.5 + torch.rand(R, S, generator=g), -1)[0].to(DEV).contiguous()
    o = Fn.merged_composite_forward(z, None, _field_like(R, S, C, g), None, want=("weights", "depth", "acc", "wsum", "var"))
    o["acc"][:, spec.ch_normal_lr:spec.ch_normal_lr + 3] = -rays[:, 3:6] * o["wsum"][:, None]
    clean = o["acc"].clone()
    bad = [3, 64, 129]
    o["acc"][bad, 0] = float("nan")
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    desc = shade_desc(model, args, spec, True, True, 1.0, 0.0, 0.0, False)
    ray_loss = torch.empty(R, device=DEV)
    k0 = Fn.ray_shade_loss(desc, o["acc"], o["wsum"], o["depth"], o["var"], rays[:, 3:6], rays[:, 8:11], rgbs, ray_loss=ray_loss)
    assert not bool(torch.isfinite(ray_loss.sum())) and not bool(torch.isfinite(k0["d_acc"][bad]).all())
    cnt = torch.zeros(2, dtype=torch.int64, device=DEV)
    ray_loss2 = torch.empty(R, device=DEV)
    k1 = Fn.ray_shade_loss(desc, o["acc"], o["wsum"], o["depth"], o["var"], rays[:, 3:6], rays[:, 8:11], rgbs, ray_loss=ray_loss2,
                           nonfinite=cnt)
    assert cnt.tolist() == [3, 0]
    assert bool(torch.isfinite(ray_loss2).all()) and float(ray_loss2[bad].abs().max()) == 0.0
    assert float(k1["d_acc"][bad].abs().max()) == 0.0 and float(k1["d_wsum"][bad].abs().max()) == 0.0
    ref = torch.empty(R, device=DEV)
    k2 = Fn.ray_shade_loss(desc, clean, o["wsum"], o["depth"], o["var"], rays[:, 3:6], rays[:, 8:11], rgbs, ray_loss=ref)
    ok = torch.ones(R, dtype=torch.bool, device=DEV)
    ok[bad] = False
    assert torch.equal(ray_loss2[ok], ref[ok]) and torch.equal(k1["d_acc"][ok], k2["d_acc"][ok])


_REF_SHADE = {   # golden -> (model flags, step flags): the reference's own per-sample outputs of its render fixtures
    "render_rpv111_nlr_train": (dict(funcM=1, funcF=1, funcH=1, normal="learned"), True, True),
    "render_rpv111_nan_train": (dict(funcM=1, funcF=1, funcH=1, normal="analystic"), True, True),
    "render_hapke_bct_train": (dict(b=1, c=1, theta=1, normal="learned"), True, True),
    "render_hapke_bc_train": (dict(b=1, c=1, normal="analystic"), True, True),
    "render_microfacet_train": (dict(roughness=True, normal="learned"), True, True),
    "render_rpv_m1f1h2_test": (dict(funcM=1, funcF=1, funcH=2, normal="learned"), True, True),
    "render_rpv_m1h2_test": (dict(funcM=1, funcH=2, normal="learned"), True, True),
    "render_shell1_nobrdf_test": (dict(shell_hapke=1, normal="learned"), False, False),
    "render_shell2_nobrdf_test": (dict(shell_hapke=2, normal="learned"), False, True),
    "render_shell3_nobrdf_test": (dict(shell_hapke=3, normal="learned"), False, True),
    "render_shell3_brdf_test": (dict(shell_hapke=3, normal="learned"), True, True),
}
_HEAD_KEYS = {"k_from_xyz": "rpv_k", "theta_rpv_from_xyz": "rpv_theta", "rhoc_from_xyz": "rpv_rhoc", "b_from_xyz": "hpk_b",
              "c_from_xyz": "hpk_c", "theta_from_xyz": "hpk_theta", "roughness_from_xyz": "roughness"}


@pytest.mark.parametrize("name", list(_REF_SHADE))
def test_ray_shade_loss_on_the_references_per_sample_outputs(name):
    """the new glue kernel against the reference's render goldens.  The fixtures hold the REFERENCE's
    per-sample field outputs on the merged S+G set (albedo, sigmas, normals, BRDF parameters, z_vals; produced by rendering.py:
    168-291 + models/spsbrdfnerf.py:198-357): composited by bn_merged_composite_forward and shaded by bn_ray_shade_loss they must
    give the reference's rgb_coarse (and, for the training fixtures, its loss = mean((rgb - target)^2) + 0.01 mean(depth))."""
    from test_gpu_parity import build_model, make_args, load_golden, mini
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.rendering import shade_desc
    kw, brdf, cos_on = _REF_SHADE[name]
    g = load_golden(name)
    cfg = mini(**kw)
    args = make_args(cfg)
    model = build_model(cfg, 11)
    spec = model.spec(brdf, brdf, cfg.normal in ("learned", "analystic_learned"), cfg.normal in ("analystic", "analystic_learned"),
                      beta=False)
    t = lambda k: torch.from_numpy(g["out/" + k + "_coarse"]).to(DEV)
    z = t("z_vals").contiguous()
    R, S = z.shape
    out = torch.zeros(R, S, spec.out_channels, device=DEV)
    out[..., :3], out[..., 3] = t("albedo"), t("sigmas").reshape(R, S)
    if spec.normal_an:
        out[..., spec.ch_normal_an:spec.ch_normal_an + 3] = t("normal_an")
    if spec.normal_lr:
        out[..., spec.ch_normal_lr:spec.ch_normal_lr + 3] = t("normal_lr")
    for (hname, _, _), (c0, wdt) in zip(spec.heads[1:], spec.head_cols[1:]):
        out[..., c0:c0 + wdt] = t(_HEAD_KEYS[hname]).reshape(R, S, wdt)
    o = Fn.merged_composite_forward(z, None, out, None, want=("weights", "depth", "acc", "wsum", "var"))
    assert float((o["weights"] - t("weights")).abs().max()) <= 1e-5 and float((o["depth"] - t("depth")).abs().max()) <= 2e-5
    rays = torch.from_numpy(g["rays"]).to(DEV)
    rgbs = torch.from_numpy(g["tgt/rgbs"]).to(DEV) if "tgt/rgbs" in g else torch.zeros(R, 3, device=DEV)
    desc = shade_desc(model, args, spec, brdf, cos_on, 1.0, 0.0, 0.0, False)
    ray_loss = torch.empty(R, device=DEV)
    k = Fn.ray_shade_loss(desc, o["acc"], o["wsum"], o["depth"], o["var"], rays[:, 3:6], rays[:, 8:11], rgbs, ray_loss=ray_loss)
    # (same ray-level bounds as test_render_rays_golden_fp32: the GGX lobe and Hapke's opposition terms amplify 1e-7)
    tol = 1e-4 if "microfacet" in name else (4e-5 if ("hapke" in name or "shell" in name) else 2e-5)
    err = float((k["rgb"] - t("rgb")).abs().max())
    assert err <= tol, (name, err)
    if "loss" in g and "tgt/rgbs" in g:
        loss = float(ray_loss.sum()) + 0.01 * float(o["depth"].mean())
        assert abs(loss - float(g["loss"])) <= 1e-4 * abs(float(g["loss"])) + 1e-7, (name, loss, float(g["loss"]))


# ------------------------------------------------------------------------------------------------ fold / unfold / Adam
@pytest.mark.parametrize("F,heads,dir_dim", [(512, 1, 0), (64, 3, 0), (192, 2, 24)])
def test_fold_and_unfold_kernels_against_torch(F, heads, dir_dim):
    from brdf_nerf_amd import _lib as L
    from brdf_nerf_amd import functions as Fn
    g = torch.Generator().manual_seed(F + heads)
    H2 = F // 2
    names = ["rgb_from_xyzdir", "k_from_xyz", "rhoc_from_xyz"][:heads]
    hl = [(n, 3 if i == 0 else 1, L.BN_HEAD_PLAIN if i == 0 else L.BN_HEAD_TILE3) for i, n in enumerate(names)]
    spec = Fn.FieldSpec(F, 4, 2, 10, L.BN_ACT_SIN, L.BN_F32, hl, False, dir_dim=dir_dim, dir_freqs=4 if dir_dim else 0)
    named = {"feats_from_xyz.weight": torch.randn(F, F, generator=g).to(DEV) / F ** 0.5, "feats_from_xyz.bias": torch.randn(F, generator=g).to(DEV)}
    for i, n in enumerate(names):
        named[f"{n}.0.weight"] = (torch.randn(H2, F + (dir_dim if i == 0 else 0), generator=g) / F ** 0.5).to(DEV)
        named[f"{n}.0.bias"] = torch.randn(H2, generator=g).to(DEV)
    spec.fold(named)
    wf, bf = named["feats_from_xyz.weight"].double(), named["feats_from_xyz.bias"].double()
    for n in names:
        w1, b1 = named[f"{n}.0.weight"][:, :F].double(), named[f"{n}.0.bias"].double()
        assert float((spec.folded[n][0].double() - w1 @ wf).abs().max()) <= 2e-6 * float((w1 @ wf).abs().max())
        assert float((spec.folded[n][1].double() - (w1 @ bf + b1)).abs().max()) <= 2e-6 * float((w1 @ bf + b1).abs().max())
    # unfold: accumulates into the gradient buffers
    grads = {k: torch.randn(v.shape, generator=g).to(DEV) for k, v in named.items()}
    before = {k: v.clone() for k, v in grads.items()}
    for n in names:
        spec.fold_grads[n] = (torch.randn(H2, F, generator=g).to(DEV), torch.randn(H2, generator=g).to(DEV))
    ms = {n: (m.clone().double(), s.clone().double()) for n, (m, s) in spec.fold_grads.items()}
    spec.unfold_grads(named, grads, zero=True)
    dwf, dbf = before["feats_from_xyz.weight"].double(), before["feats_from_xyz.bias"].double()
    for n in names:
        m, sv = ms[n]
        w1 = named[f"{n}.0.weight"][:, :F].double()
        want = before[f"{n}.0.weight"].double()
        want[:, :F] += m @ wf.t() + torch.outer(sv, bf)
        assert float((grads[f"{n}.0.weight"].double() - want).abs().max()) <= 3e-6 * float(want.abs().max())
        assert float((grads[f"{n}.0.bias"].double() - (before[f"{n}.0.bias"].double() + sv)).abs().max()) <= 1e-6
        dwf = dwf + w1.t() @ m
        dbf = dbf + w1.t() @ sv
        assert float(spec.fold_grads[n][0].abs().max()) == 0.0 and float(spec.fold_grads[n][1].abs().max()) == 0.0
    assert float((grads["feats_from_xyz.weight"].double() - dwf).abs().max()) <= 3e-6 * float(dwf.abs().max())
    assert float((grads["feats_from_xyz.bias"].double() - dbf).abs().max()) <= 3e-6 * float(dbf.abs().max())
    # fold clears the accumulators
    for n in names:
        spec.fold_grads[n][0].fill_(1.0)
        spec.fold_grads[n][1].fill_(1.0)
    spec.fold(named)
    for n in names:
        assert float(spec.fold_grads[n][0].abs().max()) == 0.0 and float(spec.fold_grads[n][1].abs().max()) == 0.0


def test_adam_multi_matches_torch_per_group():
    """One launch over three groups with their own step counts (a group that joins late starts its bias corrections at 1, an
    inactive group is skipped), learning rate and counters from the device state; the gradients it read are cleared."""
    from brdf_nerf_amd import functions as Fn
    g = torch.Generator().manual_seed(0)
    sizes = [100004, 4096, 260]
    groups, lo = [], 0
    for n in sizes:
        groups.append((lo, lo + n))
        lo += n
    p0 = torch.randn(lo, generator=g)
    ref = [p0[a:b].clone().requires_grad_(True) for a, b in groups]
    opts = [torch.optim.Adam([r], lr=5e-4) for r in ref]
    p, m, v = p0.clone().to(DEV), torch.zeros(lo, device=DEV), torch.zeros(lo, device=DEV)
    st = Fn.new_step_state(DEV, 1, 5e-4)
    sched = [(True, False, False), (True, True, False), (True, True, True), (True, False, True)]
    for step, active in enumerate(sched):
        gr = torch.randn(lo, generator=g)
        if step == 2:                                        # learning-rate decay between steps
            Fn.state_views(st)[1].fill_(2.5e-4)
            for o in opts:
                o.param_groups[0]["lr"] = 2.5e-4
        for r, o, (a, b), on in zip(ref, opts, groups, active):
            if on:
                r.grad = gr[a:b].clone()
                o.step()
        gd = gr.to(DEV)
        Fn.adam_multi(p, gd, m, v, groups, active, st)
        for (a, b), on in zip(groups, active):
            assert (float(gd[a:b].abs().max()) == 0.0) == on          # read gradients are cleared, skipped groups untouched
        rng_step, _, steps, ring = Fn.state_views(st)
        assert int(rng_step) == step + 1
    assert steps.tolist()[:3] == [4, 2, 2]
    for r, (a, b) in zip(ref, groups):
        assert float((p[a:b].cpu() - r.detach()).abs().max()) <= 2e-6


# ------------------------------------------------------------------------------------------------ the lean step
def _lean_cfgs():
    return {
        "lambert": dict(),
        "rpv111_nlr": dict(funcM=1, funcF=1, funcH=1, normal="learned"),
        "rpv111_nan": dict(funcM=1, funcF=1, funcH=1, normal="analystic"),
        "hapke_bct": dict(b=1, c=1, theta=1, normal="learned"),
        "microfacet": dict(roughness=True, normal="analystic"),
    }


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("name", list(_lean_cfgs()))
def test_lean_step_matches_general_step_on_the_same_draws(name, dtype):
    """FusedTrainer's launch-lean step (in-kernel draws, fused per-ray kernels, fold / unfold / Adam kernels; eager, then
    replayed from a HIP graph) against the general step (ATen glue; held to the autograd path and the
    oracle by test_gpu_fuzz.py) fed with the Philox streams' draws as arrays: loss, rgb and parameters over five steps."""
    import brdf_nerf_amd
    from test_gpu_parity import build_model, make_args, Replay, diag
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.trainer import FusedTrainer
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16, **_lean_cfgs()[name])
    args = make_args(cfg, dtype)
    R, S, G = 96, 16, 16
    g = torch.Generator().manual_seed(3)
    rays = _sat_rays(R, g).to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV)
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV)
    dstd = (0.03 * torch.rand(R, generator=g)).to(DEV)
    brdf = name != "lambert"
    flags = dict(apply_brdf=brdf, apply_theta=brdf, cos_irra_on=brdf)
    prev = brdf_nerf_amd.set_deterministic(True)          # bitwise reproducible weight-gradient sums on both sides
    try:
        torch.manual_seed(11)
        ma, mb = build_model(cfg, 21, dtype), build_model(cfg, 21, dtype)
        # (hard-surface and normal regularisers on for the models with normals: both are part of the lean step)
        lam = dict(hs_lambda=0.1, nr_reg_an_lambda=0.2, nr_reg_lr_lambda=0.1) if brdf else {}
        ta = FusedTrainer(ma, args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
        tb = FusedTrainer(mb, args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
        ta.lean = False
        tb.graph_after = 2
        tb.keep_grads = True                                 # test hook: the Adam launch leaves the gradient it read in place
        worst_g = 0.0
        for step in range(6):
            if step == 4:
                ta.lr = tb.lr = 2.5e-4                        # learning-rate decay reaches the device state (and the replayed graph)
            # every step starts from the SAME parameters and moments: Adam's first steps turn a 1e-7 difference in a near-zero
            # gradient entry into +-lr, and a 16-bit mode rounds a 1e-7 input difference into a 4e-3 one - the steps are compared
            # one by one, not as two trajectories
            tb.flat_param.copy_(ta.flat_param)
            tb.exp_avg.copy_(ta.exp_avg)
            tb.exp_avg_sq.copy_(ta.exp_avg_sq)
            m_before = ta.exp_avg.clone()
            draws = [Fn.rng_uniform(tb.state, 1, R * S).view(R, S), Fn.rng_uniform(tb.state, 2, R * G).view(R, G),
                     Fn.rng_uniform(tb.state, 3, R * G).view(R, G)]
            with Replay(draws) as rp:
                la, rgb_a = ta.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
                assert rp.draws == []
            lb, rgb_b = tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
            la, lb = float(la), float(lb)
            assert abs(la - lb) <= 2e-5 * abs(la) + 1e-7, (name, dtype, step, la, lb)
            # (the GGX lobe turns a 1e-7 difference in the accumulated normal into 1e-5 .. 1e-4 on rgb: test_gpu_parity's bound)
            assert float((rgb_a - rgb_b).abs().max()) <= (1e-4 if name == "microfacet" else 2e-5), (name, dtype, step)
            ga, gb = ta.flat_grad, tb.flat_grad
            scale = float(ga.abs().max())
            e = float((ga - gb).abs().max()) / scale
            cos = float(torch.nn.functional.cosine_similarity(ga.double(), gb.double(), dim=0))
            worst_g = max(worst_g, e)
            # (analytic normals - rpv111_nan, microfacet: the two ray-level evaluations differ by fp32 rounding (1e-7) in the composited
            # normal, which the normalisation of a small density gradient and the GGX lobe amplify; measured 2e-5 .. 1.7e-4 depending
            # on the last bits of the folded weights, learned normals 3e-6)
            sensitive = name in ("microfacet", "rpv111_nan")
            if dtype == "fp32":
                assert e <= (5e-4 if sensitive else 5e-5), (name, dtype, step, e)
            else:       # a 1e-7 difference of a gradient seed can flip its 16-bit rounding
                assert cos >= 0.99995 and e <= 2e-2, (name, dtype, step, cos, e)
            # the optimiser: first moments are linear in the gradient, parameters move by at most lr per step
            dm = float(((ta.exp_avg - m_before) - (tb.exp_avg - m_before)).abs().max()) / (0.1 * scale)
            assert dm <= ((5e-4 if sensitive else 5e-5) if dtype == "fp32" else 2e-2), (name, dtype, step, dm)
            assert float((ta.flat_param - tb.flat_param).abs().max()) <= 2.1 * ta.lr
            assert float((ta.flat_param - tb.flat_param).abs().mean()) <= (2e-7 if dtype == "fp32" else 2e-5), (name, dtype, step)
        assert ta.adam_steps == tb.adam_steps
        assert len(tb._graphs) == 1, "the lean step was not captured into a HIP graph"
        diag(f"lean vs general step {name} {dtype}: worst flat-gradient difference over 6 steps {worst_g:.2e} of the largest entry; "
             f"graphs {len(tb._graphs)}")
    finally:
        brdf_nerf_amd.set_deterministic(prev)


@pytest.mark.parametrize("name,gsam", [("lambert", False), ("rpv111_nlr", False), ("lambert", True), ("rpv111_nan", True)])
def test_lean_step_with_noise_std_matches_general_step(name, gsam):
    """--noise_std > 0 (models/spsbrdfnerf.py:57-59) on the launch-lean step: in-kernel normal draws in the pass-1 compositing
    and in the compositing of the merged set (forward and backward see the same draws), against the general step fed with the
    streams' normals as arrays (torch.randn at the same two places); eager, then replayed from a HIP graph."""
    import brdf_nerf_amd
    from test_gpu_parity import build_model, make_args, Replay, diag
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.trainer import FusedTrainer
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16, **_lean_cfgs()[name])
    args = make_args(cfg, "fp32")
    args.noise_std = 0.7
    R, S, G = 96, 16, 16
    g = torch.Generator().manual_seed(5)
    rays = _sat_rays(R, g).to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV)
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV)
    dstd = (0.03 * torch.rand(R, generator=g)).to(DEV)
    brdf = name != "lambert"
    flags = dict(apply_brdf=brdf, apply_theta=brdf, cos_irra_on=brdf, gsam_only=gsam)
    prev = brdf_nerf_amd.set_deterministic(True)
    try:
        torch.manual_seed(13)
        ma, mb = build_model(cfg, 23, "fp32"), build_model(cfg, 23, "fp32")
        ta = FusedTrainer(ma, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
        tb = FusedTrainer(mb, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
        ta.lean = False
        tb.graph_after = 1
        tb.keep_grads = True
        worst = 0.0
        for step in range(4):
            tb.flat_param.copy_(ta.flat_param)
            tb.exp_avg.copy_(ta.exp_avg)
            tb.exp_avg_sq.copy_(ta.exp_avg_sq)
            S2 = G if gsam else S + G
            draws = [Fn.rng_uniform(tb.state, 1, R * S).view(R, S), Fn.rng_normal(tb.state, 4, R * S).view(R, S),
                     Fn.rng_uniform(tb.state, 2, R * G).view(R, G), Fn.rng_uniform(tb.state, 3, R * G).view(R, G),
                     Fn.rng_normal(tb.state, 5, R * S2).view(R, S2)]
            with Replay(draws) as rp:
                la, rgb_a = ta.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
                assert rp.draws == []
            lb, rgb_b = tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
            la, lb = float(la), float(lb)
            assert abs(la - lb) <= 2e-5 * abs(la) + 1e-7, (name, gsam, step, la, lb)
            assert float((rgb_a - rgb_b).abs().max()) <= 2e-5, (name, gsam, step)
            ga, gb = ta.flat_grad, tb.flat_grad
            e = float((ga - gb).abs().max()) / float(ga.abs().max())
            worst = max(worst, e)
            # (sigma + noise * noise_std is one fma in the kernels, a product and a sum in the general step's torch statement: a
            # last-bit difference in front of the exponential; measured 7e-5 with the BRDF heads, 5e-6 without)
            assert e <= (5e-4 if brdf else 5e-5), (name, gsam, step, e)
        # the noise matters: the same step without it differs
        args.noise_std = 0.0
        l0, _ = tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
        args.noise_std = 0.7
        assert abs(float(l0) - lb) > 1e-4 * abs(lb)
        assert len(tb._graphs) >= 1, "the lean step was not captured into a HIP graph"
        diag(f"lean step with noise_std {name} gsam_only={gsam}: worst flat-gradient difference over 4 steps {worst:.2e} of the largest entry")
    finally:
        brdf_nerf_amd.set_deterministic(prev)


@pytest.mark.parametrize("gsam", [False, True])
@pytest.mark.parametrize("with_reg", [False, True])
def test_lean_step_with_normal_loss_matches_general_step(with_reg, gsam):
    """NormalLoss between the two per-sample normal fields (--nr_spv_lambda, nr_spv_type 1: metrics.py:218-261, main.py:297-303)
    on the launch-lean step: the rays' sums from the forward compositing, bn_normal_spv_reduce (the two batch-wide means, fixed
    order), the gradient in the backward compositing - against the general step's torch statement; eager, then replayed."""
    import brdf_nerf_amd
    from test_gpu_parity import build_model, make_args, Replay, diag
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.trainer import FusedTrainer
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16, funcM=1, funcF=1, funcH=1, normal="analystic_learned")
    args = make_args(cfg, "fp32")
    R, S, G = 96, 16, 16
    g = torch.Generator().manual_seed(7)
    rays = _sat_rays(R, g).to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV)
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV)
    dstd = (0.03 * torch.rand(R, generator=g)).to(DEV)
    flags = dict(apply_brdf=True, apply_theta=True, cos_irra_on=True, gsam_only=gsam)
    lam = dict(nr_spv_lambda=0.5, **(dict(hs_lambda=0.1, nr_reg_an_lambda=0.2, nr_reg_lr_lambda=0.1) if with_reg else {}))
    prev = brdf_nerf_amd.set_deterministic(True)
    try:
        torch.manual_seed(17)
        ma, mb = build_model(cfg, 29, "fp32"), build_model(cfg, 29, "fp32")
        ta = FusedTrainer(ma, args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
        tb = FusedTrainer(mb, args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
        ta.lean = False
        tb.graph_after = 1
        tb.keep_grads = True
        worst = 0.0
        for step in range(4):
            tb.flat_param.copy_(ta.flat_param)
            tb.exp_avg.copy_(ta.exp_avg)
            tb.exp_avg_sq.copy_(ta.exp_avg_sq)
            draws = [Fn.rng_uniform(tb.state, 1, R * S).view(R, S), Fn.rng_uniform(tb.state, 2, R * G).view(R, G),
                     Fn.rng_uniform(tb.state, 3, R * G).view(R, G)]
            with Replay(draws) as rp:
                la, rgb_a = ta.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
                assert rp.draws == []
            lb, rgb_b = tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
            la, lb = float(la), float(lb)
            assert abs(la - lb) <= 2e-5 * abs(la) + 1e-7, (with_reg, gsam, step, la, lb)
            ga, gb = ta.flat_grad, tb.flat_grad
            e = float((ga - gb).abs().max()) / float(ga.abs().max())
            worst = max(worst, e)
            assert e <= 5e-4, (with_reg, gsam, step, e)          # (analytic normals: test_lean_step_matches_general_step's bound)
        # the term is there: the same step without it has another loss
        tb.reg["nr_spv"] = 0.0
        l0, _ = tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
        tb.reg["nr_spv"] = 0.5
        assert abs(float(l0) - lb) > 1e-4 * abs(lb)
        assert len(tb._graphs) >= 1, "the lean step was not captured into a HIP graph"
        diag(f"lean step with NormalLoss an_lr reg={with_reg} gsam_only={gsam}: worst flat-gradient difference over 4 steps {worst:.2e} of the largest entry")
    finally:
        brdf_nerf_amd.set_deterministic(prev)


@pytest.mark.parametrize("name,gsam,cosi,with_reg", [("rpv111_nlr", False, True, False), ("hapke_bct", False, False, False),
                                                     ("microfacet", True, True, False), ("rpv111_nan", True, False, False),
                                                     ("rpv111_nlr", False, True, True), ("microfacet", False, False, True),
                                                     ("hapke_bct", True, True, True)])
def test_lean_step_multibrdf_matches_general_step(name, gsam, cosi, with_reg):
    """--MultiBRDF (one BRDF per sample, models/spsbrdfnerf.py:289-307,350-352) on the launch-lean step: the BRDF evaluated on the
    stored rows by one launch (csrc/sample_brdf.hip), its padded value as the colour channels of a 4-channel copy that the
    Lambertian tail kernel composites, one launch for the chain rule back to the field outputs - against the general step (autograd
    through the per-point BRDF functions) on the same draws; eager, then replayed.
    with_reg (round 5): together with the hard-surface and normal regularisers - the lean step then composites a FULL-width copy of
    the rows (colour channels = the padded BRDF value) through the generic compositing / ray-loss kernels, which carry them."""
    import brdf_nerf_amd
    from test_gpu_parity import build_model, make_args, Replay, diag
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.trainer import FusedTrainer
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16, MultiBRDF=True, **_lean_cfgs()[name])
    args = make_args(cfg, "fp32")
    R, S, G = 96, 16, 16
    g = torch.Generator().manual_seed(21)
    rays = _sat_rays(R, g).to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV)
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV)
    dstd = (0.03 * torch.rand(R, generator=g)).to(DEV)
    flags = dict(apply_brdf=True, apply_theta=True, cos_irra_on=cosi, gsam_only=gsam)
    prev = brdf_nerf_amd.set_deterministic(True)
    try:
        torch.manual_seed(23)
        ma, mb = build_model(cfg, 37, "fp32"), build_model(cfg, 37, "fp32")
        lam = dict(hs_lambda=0.1, nr_reg_an_lambda=0.2, nr_reg_lr_lambda=0.1) if with_reg else {}
        ta = FusedTrainer(ma, args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
        tb = FusedTrainer(mb, args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
        ta.lean = False
        tb.graph_after = 1
        tb.keep_grads = True
        worst = 0.0
        for step in range(4):
            tb.flat_param.copy_(ta.flat_param)
            tb.exp_avg.copy_(ta.exp_avg)
            tb.exp_avg_sq.copy_(ta.exp_avg_sq)
            draws = [Fn.rng_uniform(tb.state, 1, R * S).view(R, S), Fn.rng_uniform(tb.state, 2, R * G).view(R, G),
                     Fn.rng_uniform(tb.state, 3, R * G).view(R, G)]
            with Replay(draws) as rp:
                la, rgb_a = ta.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
                assert rp.draws == []
            lb, rgb_b = tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
            la, lb = float(la), float(lb)
            if not (la == la):
                continue
            assert abs(la - lb) <= 2e-5 * abs(la) + 1e-7, (name, step, la, lb)
            assert float((rgb_a - rgb_b).abs().max()) <= (1e-4 if name == "microfacet" else 2e-5), (name, step)
            ga, gb = ta.flat_grad, tb.flat_grad
            e = float((ga - gb).abs().max()) / float(ga.abs().max())
            worst = max(worst, e)
            assert e <= 5e-4, (name, step, e)
        assert len(tb._graphs) >= 1, "the lean step was not captured into a HIP graph"
        assert any(k[0] == "sample" for k in tb._kind_cache), "the step did not take the per-sample BRDF branch of the lean path"
        diag(f"lean step with MultiBRDF {name} gsam_only={gsam} regularisers={with_reg}: worst flat-gradient difference over 4 steps {worst:.2e} of the largest entry")
    finally:
        brdf_nerf_amd.set_deterministic(prev)


def test_sample_brdf_kernel_lambert_kind_and_row_irradiance():
    """kind LAMBERT of the per-sample shading launch (the padded albedo under a per-ROW irradiance: the sun pass over a Lambertian
    rgb, models/spsbrdfnerf.py:265-273) and the per-row irradiance on a BRDF kind, against the same arithmetic in torch; the cosine
    term wins over the rows' irradiance exactly when the model has a normal channel (:260-266)."""
    from brdf_nerf_amd import _lib as L, functions as Fn
    g = torch.Generator().manual_seed(8)
    R, S1, pad = 29, 6, 0.001
    N = R * S1
    rays = torch.zeros(R, 11)
    rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) - torch.tensor([0.0, 0.0, 2.0]), dim=-1)
    rays[:, 8:11] = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    rays = rays.to(DEV)
    irr = torch.rand(N, generator=g).to(DEV)
    for C, ch_n, cosi in ((4, -1, False), (4, -1, True), (7, 4, False), (7, 4, True)):
        X = torch.rand(N, C, generator=g).to(DEV)
        d = L.ShadeDesc()
        d.kind, d.C, d.ch_normal = L.BN_SHADE_LAMBERT, C, ch_n
        d.ch_p0 = d.ch_p1 = d.ch_p2 = -1
        d.rhoc_is_albedo = d.shell = d.usealldepth = 0
        d.cos_irradiance = int(cosi)
        d.hpk_scl, d.f0, d.rgb_padding = 1.0, 0.04, pad
        d.lambda_rgb = d.lambda_ds = d.lambda_hs = 0.0
        d.irr, d.irr_stride = irr.data_ptr(), 1
        for Cb in sorted({4, C}):
            B = torch.full((N, Cb), float("nan"), device=DEV)
            Fn.sample_brdf(d, X, rays, N, S1, 0, B)
            w = rays[:, 10].abs().repeat_interleave(S1) if (cosi and ch_n >= 0) else irr
            ref = torch.cat([(X[:, :3] * (1 + 2 * pad) - pad) * w[:, None], X[:, 3:Cb]], 1)
            assert torch.equal(B, ref), (C, cosi, Cb)
            dB = torch.randn(N, Cb, generator=g).to(DEV)
            dX = torch.full((N, C), float("nan"), device=DEV)
            Fn.sample_brdf(d, X, rays, N, S1, 0, dX, backward_of=dB)
            dref = torch.zeros(N, C, device=DEV)
            dref[:, :3] = dB[:, :3] * (1 + 2 * pad) * w[:, None]
            dref[:, 3:Cb] = dB[:, 3:]
            assert torch.equal(dX, dref), (C, cosi, Cb)


@pytest.mark.parametrize("kind,heads,rhoc_is_albedo,shell", [("RPV", "k t r", False, 0), ("RPV", "k", False, 0), ("RPV", "k t", True, 0),
                                                             ("Hapke", "b c t", False, 0), ("Hapke", "c", False, 2), ("Hapke", "b t", False, 0),
                                                             ("Microfacet", "r", False, 0)])
@pytest.mark.parametrize("full,cosi", [(False, True), (True, False)])
def test_sample_brdf_kernel_matches_per_point_functions(kind, heads, rhoc_is_albedo, shell, full, cosi):
    """bn_sample_brdf_forward / _backward (the per-sample BRDF of --MultiBRDF on stored rows, one launch each way) against the
    per-point BRDF functions under torch autograd (the per-point entry points, themselves pinned by the oracle and the
    golden vectors) with the gathers, the padding and the irradiance as torch ops: every combination of present heads the model
    builder allows, a two-block row set with different samples per ray, rows with degenerate geometry (view == normal, sun below
    the horizon: the NaN-replacement branches) - values and gradients to the last bits, the pass-through channels exactly."""
    from types import SimpleNamespace
    from brdf_nerf_amd import _lib as L, functions as Fn
    from brdf_nerf_amd.rendering import _per_ray_brdf
    g = torch.Generator().manual_seed(5)
    R, S1, S2, pad = 37, 5, 3, 0.001
    names = heads.split()
    C, cols = 4 + 3 + 2, {}                       # [albedo 3, sigma, 2 spare channels, normal 3, heads ...]: the normal at 6
    ch_n = 6
    for nm in names:
        w = 1 if (kind == "Microfacet" or (kind == "Hapke" and nm == "t")) else 3
        cols[nm] = (C, w)
        C += w
    N, n1 = R * (S1 + S2), R * S1
    X = torch.rand(N, C, generator=g)
    nrm = torch.nn.functional.normalize(torch.randn(N, 3, generator=g) + torch.tensor([0.0, 0.0, 1.5]), dim=-1)
    X[:, ch_n:ch_n + 3] = nrm
    rays = torch.zeros(R, 11)
    rays[:, :3] = torch.randn(R, 3, generator=g)
    rays[:, 3:6] = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) - torch.tensor([0.0, 0.0, 2.0]), dim=-1)
    rays[:, 8:11] = torch.nn.functional.normalize(torch.randn(R, 3, generator=g) + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    rays[3, 8:11] = torch.tensor([0.6, 0.0, -0.8])           # a sun below the horizon
    X[7 * S1, ch_n:ch_n + 3] = -rays[7, 3:6]                  # view == normal
    X[n1 + 2 * S2 + 1, ch_n:ch_n + 3] = rays[2, 8:11]         # sun == normal
    if kind == "RPV" and "k" in cols:
        X[:, cols["k"][0]:cols["k"][0] + 3] += 0.3
    X, rays = X.to(DEV), rays.to(DEV)
    d = L.ShadeDesc()
    d.kind = {"RPV": L.BN_SHADE_RPV, "Hapke": L.BN_SHADE_HAPKE, "Microfacet": L.BN_SHADE_MICROFACET}[kind]
    d.C, d.ch_normal = C, ch_n
    order = {"RPV": ("k", "t", "r"), "Hapke": ("b", "c", "t"), "Microfacet": ("r",)}[kind]
    p = [cols[nm][0] if nm in cols else -1 for nm in order] + [-1, -1]
    d.ch_p0, d.ch_p1, d.ch_p2 = p[0], p[1], p[2]
    d.rhoc_is_albedo, d.shell, d.cos_irradiance, d.usealldepth = int(rhoc_is_albedo), shell, int(cosi), 0
    d.hpk_scl, d.f0, d.rgb_padding = 1.3, 0.04, pad
    d.lambda_rgb = d.lambda_ds = d.lambda_hs = 0.0
    Cb = C if full else 4
    B = torch.full((N, Cb), float("nan"), device=DEV)
    Fn.sample_brdf(d, X, rays, n1, S1, S2, B)
    dB = torch.randn(N, Cb, generator=g).to(DEV)
    dX = torch.full((N, C), float("nan"), device=DEV)
    Fn.sample_brdf(d, X, rays, n1, S1, S2, dX, backward_of=dB)
    # the same through autograd over the per-point functions
    ar = torch.arange(R, device=DEV)
    row_ray = torch.cat([ar.repeat_interleave(S1), ar.repeat_interleave(S2)])
    Xl = X.clone().requires_grad_(True)
    hn = {"RPV": dict(k="k_from_xyz", t="theta_rpv_from_xyz", r="rhoc_from_xyz"), "Hapke": dict(b="b_from_xyz", c="c_from_xyz", t="theta_from_xyz"),
          "Microfacet": dict(r="roughness_from_xyz")}[kind]
    hd = {hn[nm]: Xl[:, c0:c0 + w] for nm, (c0, w) in cols.items()}
    args = SimpleNamespace(funcH=2 if rhoc_is_albedo else 1, fresnel_f0=0.04, hpk_scl=1.3, shell_hapke=shell)
    brdf, _ = _per_ray_brdf(None, args, kind, rays[:, 8:11][row_ray], (-rays[:, 3:6])[row_ray], Xl[:, ch_n:ch_n + 3], Xl[:, :3], hd)
    bp = brdf * (1 + 2 * pad) - pad
    if cosi:
        bp = bp * rays[:, 10:11].abs()[row_ray]
    Bref = torch.cat([bp, Xl[:, 3:]] if full else [bp, Xl[:, 3:4]], 1)
    Bref.backward(dB)
    assert torch.equal(torch.isnan(B), torch.isnan(Bref))
    tol = lambda ref: 2e-6 * float(ref.detach().abs().nan_to_num(0.0).max()) + 1e-9
    assert float((B - Bref.detach()).abs().nan_to_num(0.0).max()) <= tol(Bref)
    assert torch.equal(B[:, 3:], Bref.detach()[:, 3:])
    gref = Xl.grad
    assert torch.equal(torch.isnan(dX), torch.isnan(gref))
    err = float((dX - gref).abs().nan_to_num(0.0).max())
    assert err <= 5e-6 * float(gref.abs().nan_to_num(0.0).max()) + 1e-9, err
    if full:
        spare = [4, 5]
        assert torch.equal(dX[:, spare], dB[:, spare]) and torch.equal(dX[:, 3], dB[:, 3])
    else:
        assert float(dX[:, [4, 5]].abs().max()) == 0.0 and torch.equal(dX[:, 3], dB[:, 3])
    # ragged arguments are refused, not launched: by the wrapper, and by the C entry point itself (error text names the argument)
    with pytest.raises(AssertionError):
        Fn.sample_brdf(d, X, rays, n1 + 1, S1, S2, B)
    import ctypes as C_
    lib, vp = L.lib(), lambda t: C_.c_void_p(t.data_ptr())
    call = lambda dd, n1_, s1_, stride: lib.bn_sample_brdf_forward(C_.byref(dd), vp(X), vp(rays), R, rays.stride(0), 8, N, n1_, s1_, S2, vp(B), stride, None)
    assert call(d, n1 + 1, S1, Cb) != 0 and b"do not split" in lib.bn_last_error()
    assert call(d, n1, S1, 5) != 0 and b"row stride" in lib.bn_last_error()
    bad = L.ShadeDesc.from_buffer_copy(d)
    bad.ch_normal = C - 1
    assert call(bad, n1, S1, Cb) != 0 and b"normal channel" in lib.bn_last_error()
    bad = L.ShadeDesc.from_buffer_copy(d)
    bad.kind = 9
    assert call(bad, n1, S1, Cb) != 0 and b"kind" in lib.bn_last_error()
    assert lib.bn_sample_brdf_backward(C_.byref(d), vp(X), vp(rays), R, rays.stride(0), 8, N, n1, S1, S2, None, Cb, vp(dX), None) != 0


@pytest.mark.parametrize("name,multi", [("rpv111_nan", False), ("hapke_bct", False), ("rpv111_nan", True), ("microfacet", True),
                                        ("lambert", False), ("normal_only", False), ("normal_only", True)])
def test_lean_step_sun_visibility_pass_matches_general_step(name, multi):
    """--sun_v analystic (rendering.py:244-259; the reference runs it in the gsam_only stage) on the launch-lean step: the
    sigma-only pass along the sun direction with in-kernel draws (stream BN_RNG_SUN), its transparency in front of the last sample as
    the ray's irradiance in bn_ray_shade_loss (spsbrdfnerf.py:354) - against the general step fed with the streams' draws.
    multi / lambert / normal_only (round 5): the PER-SAMPLE irradiance - with --MultiBRDF each sample's BRDF value, without a BRDF
    each sample's padded albedo, is weighted by the sun ray's transparency at the sample's position (:265-273) - applied by the
    per-sample shading launch (csrc/sample_brdf.hip) ahead of the compositing."""
    import brdf_nerf_amd
    from test_gpu_parity import build_model, make_args, Replay, diag
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.trainer import FusedTrainer
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16, sun_v="analystic", MultiBRDF=multi,
                      **dict(_lean_cfgs(), normal_only=dict(normal="learned"))[name])
    args = make_args(cfg, "fp32")
    R, S, G = 96, 16, 16
    g = torch.Generator().manual_seed(9)
    rays = _sat_rays(R, g).to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV)
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV)
    dstd = (0.03 * torch.rand(R, generator=g)).to(DEV)
    flags = dict(apply_brdf=True, apply_theta=True, cos_irra_on=False, gsam_only=True)
    prev = brdf_nerf_amd.set_deterministic(True)
    try:
        torch.manual_seed(19)
        ma, mb = build_model(cfg, 31, "fp32"), build_model(cfg, 31, "fp32")
        ta = FusedTrainer(ma, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
        tb = FusedTrainer(mb, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
        ta.lean = False
        tb.graph_after = 1
        tb.keep_grads = True
        worst = 0.0
        for step in range(4):
            tb.flat_param.copy_(ta.flat_param)
            tb.exp_avg.copy_(ta.exp_avg)
            tb.exp_avg_sq.copy_(ta.exp_avg_sq)
            # the general step's order: z, the sun pass's depths and its (unused: noise_std = 0) normals, u, u_target
            draws = [Fn.rng_uniform(tb.state, 1, R * S).view(R, S), Fn.rng_uniform(tb.state, 6, R * G).view(R, G),
                     torch.zeros(R, G, device=DEV), Fn.rng_uniform(tb.state, 2, R * G).view(R, G), Fn.rng_uniform(tb.state, 3, R * G).view(R, G)]
            with Replay(draws) as rp:
                la, rgb_a = ta.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
                assert rp.draws == []
            lb, rgb_b = tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
            la, lb = float(la), float(lb)
            assert abs(la - lb) <= 2e-5 * abs(la) + 1e-7, (name, step, la, lb)
            assert float((rgb_a - rgb_b).abs().max()) <= 2e-5, (name, step)
            ga, gb = ta.flat_grad, tb.flat_grad
            e = float((ga - gb).abs().max()) / float(ga.abs().max())
            worst = max(worst, e)
            assert e <= 5e-4, (name, step, e)
        assert len(tb._graphs) >= 1, "the lean step was not captured into a HIP graph"
        # the sun pass matters, and switching it off on the SAME trainer must not replay the graph captured with it (sun_v is a model
        # attribute: it has to be part of the step's signature): the step without the sun pass against the general step without it
        n_graphs = len(tb._graphs)
        ma.sun_v = mb.sun_v = "none"
        for again in range(2):
            tb.flat_param.copy_(ta.flat_param)
            tb.exp_avg.copy_(ta.exp_avg)
            tb.exp_avg_sq.copy_(ta.exp_avg_sq)
            draws = [Fn.rng_uniform(tb.state, 1, R * S).view(R, S), Fn.rng_uniform(tb.state, 2, R * G).view(R, G),
                     Fn.rng_uniform(tb.state, 3, R * G).view(R, G)]
            with Replay(draws) as rp:
                _, rgb_a0 = ta.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
                assert rp.draws == []
            # (the same draws WITH the sun pass, for the "it matters" half: the sun stream's draws are in-kernel on this side)
            _, rgb_b0 = tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
            assert float((rgb_a0 - rgb_b0).abs().max()) <= 2e-5, (name, "a stale graph with the sun pass was replayed", again)
        assert len(tb._graphs) == n_graphs + 1
        ma.sun_v = mb.sun_v = "analystic"
        tb.flat_param.copy_(ta.flat_param)
        tb.exp_avg.copy_(ta.exp_avg)
        tb.exp_avg_sq.copy_(ta.exp_avg_sq)
        draws_sun = [Fn.rng_uniform(tb.state, 1, R * S).view(R, S), Fn.rng_uniform(tb.state, 2, R * G).view(R, G),
                     Fn.rng_uniform(tb.state, 3, R * G).view(R, G)]
        ma.sun_v = "none"
        with Replay(draws_sun) as rp:
            _, rgb_nosun = ta.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
        ma.sun_v = "analystic"
        _, rgb_sun = tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
        assert float((rgb_nosun - rgb_sun).abs().max()) > 1e-3      # same draws, same parameters: only the sun pass differs
        with pytest.raises(NotImplementedError):
            tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **dict(flags, gsam_only=False))
        diag(f"lean step with the sun pass {name} MultiBRDF={int(multi)}: worst flat-gradient difference over 4 steps {worst:.2e} of the largest entry")
    finally:
        brdf_nerf_amd.set_deterministic(prev)


@pytest.mark.parametrize("name", ["lambert", "rpv111_nan", "hapke_bct"])
def test_lean_step_gsam_only_matches_general_step(name):
    """The gsam_only stage (main.py:201-203: pass 1 only places the guided samples, the step renders and back-propagates through
    the G guided samples alone) on the launch-lean path - sigma-only pass 1, compositing + resampling from its densities, the
    per-ray tail over ONE source block - against the general step on the same draws: loss, rgb, flat gradient, and the step is
    captured into a graph like the others."""
    from test_gpu_parity import build_model, make_args, Replay, diag
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.trainer import FusedTrainer
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16, **_lean_cfgs()[name])
    args = make_args(cfg, "fp32")
    R, S, G = 96, 16, 16
    g = torch.Generator().manual_seed(4)
    rays = _sat_rays(R, g).to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV)
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV)
    dstd = (0.03 * torch.rand(R, generator=g)).to(DEV)
    brdf = name != "lambert"
    flags = dict(apply_brdf=brdf, apply_theta=brdf, cos_irra_on=brdf, gsam_only=True)
    torch.manual_seed(12)
    ma, mb = build_model(cfg, 22, "fp32"), build_model(cfg, 22, "fp32")
    lam = dict(hs_lambda=0.1, nr_reg_an_lambda=0.2, nr_reg_lr_lambda=0.1) if brdf else {}
    ta = FusedTrainer(ma, args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
    tb = FusedTrainer(mb, args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
    ta.lean = False
    tb.graph_after, tb.keep_grads = 2, True
    worst = 0.0
    for step in range(5):
        tb.flat_param.copy_(ta.flat_param)
        tb.exp_avg.copy_(ta.exp_avg)
        tb.exp_avg_sq.copy_(ta.exp_avg_sq)
        draws = [Fn.rng_uniform(tb.state, 1, R * S).view(R, S), Fn.rng_uniform(tb.state, 2, R * G).view(R, G),
                 Fn.rng_uniform(tb.state, 3, R * G).view(R, G)]
        with Replay(draws) as rp:
            la, rgb_a = ta.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
            assert rp.draws == []
        n_before = tb._rng_step
        lb, rgb_b = tb.step(rays, rgbs, valid_depth=valid, depths=depths, depth_std=dstd, **flags)
        assert tb._rng_step == n_before + 1, "the gsam_only step did not take the launch-lean path"
        la, lb = float(la), float(lb)
        assert abs(la - lb) <= 2e-5 * abs(la) + 1e-7, (name, step, la, lb)
        assert float((rgb_a - rgb_b).abs().max()) <= 2e-5, (name, step)
        ga, gb = ta.flat_grad, tb.flat_grad
        e = float((ga - gb).abs().max()) / float(ga.abs().max())
        worst = max(worst, e)
        assert e <= (5e-4 if name == "rpv111_nan" else 5e-5), (name, step, e)
    assert len(tb._graphs) == 1, "the gsam_only lean step was not captured into a HIP graph"
    diag(f"lean vs general gsam_only step {name}: worst flat-gradient difference over 5 steps {worst:.2e} of the largest entry")


@pytest.mark.parametrize("name", ["c2_lambert", "c3_rpv_nan"])
def test_lean_step_at_full_width_against_oracle_autograd(name):
    """ONE hop from the production step to the oracle at the reference's width: FusedTrainer's launch-lean
    step - the step bench.py times - at F = 512, 8 layers, 1024 rays x (64 + 64) samples in the fp32 parity mode, for BASELINE
    config 2 (Lambertian + depth supervision) and config 3 (RPV funcM / F / H = 1 + analytic normals: the double backward),
    against the CPU oracle (oracle/render.py render_rays + oracle/losses.py + torch autograd) on the SAME draws - the step's
    Philox streams materialised with bn_rng_uniform - evaluated in fp32 AND fp64: loss and every parameter gradient no further
    from the fp64 evaluation than 3x the fp32 oracle is; config 2 additionally to 1e-5 (loss) and 5e-4 of the largest entry
    (gradient).  (Row r of the target-guided stream belongs to ray r: the oracle, which draws one row per VALID ray like the
    reference, is fed the valid rays' rows.)"""
    from test_gpu_parity import build_model, make_args, diag, ORD
    from oracle import losses as OL
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.trainer import FusedTrainer
    kw = dict() if name == "c2_lambert" else dict(funcM=1, funcF=1, funcH=1, normal="analystic")
    cfg = FieldConfig(n_samples=64, guided_samples=64, **kw)
    assert cfg.feat == 512 and cfg.layers == 8
    args = make_args(cfg, "fp32")
    R, S, G = 1024, 64, 64
    g = torch.Generator().manual_seed(17)
    rays = _sat_rays(R, g)
    rgbs = torch.rand(R, 3, generator=g)
    valid = (torch.rand(R, generator=g) < 0.6).float()
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1)
    dstd = 0.03 * torch.rand(R, generator=g)
    brdf = name != "c2_lambert"
    flags = dict(apply_brdf=brdf, apply_theta=brdf, cos_irra_on=brdf)
    torch.manual_seed(5)
    model = build_model(cfg, 31, "fp32")
    tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
    tr.use_graph, tr.keep_grads = False, True
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    uz, ug, ut = [Fn.rng_uniform(tr.state, sid, R * n).view(R, n).cpu() for sid, n in ((1, S), (2, G), (3, G))]
    loss, _ = tr.step(rays.to(DEV), rgbs.to(DEV), valid_depth=valid.to(DEV), depths=depths.to(DEV), depth_std=dstd.to(DEV), **flags)
    assert tr._rng_step == 1, "the step did not take the launch-lean path"
    got = {k: v.detach().cpu().clone() for k, v in tr.grad_views.items()}
    n_valid = int((valid > 0).sum())
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))

    def oracle(dt):
        p = {k: v.to(dt).clone().requires_grad_(True) for k, v in state.items()}
        c = lambda t: t.to(dt)
        draws = [c(uz), torch.zeros(R, S, dtype=dt), c(ug), c(ut[valid > 0]), torch.zeros(R, S + G, dtype=dt)]   # (randn draws act only through noise_std = 0)
        res, _ = ORD.render_rays(p, cfg, c(rays), ORD.Randoms(replay=draws), mode="train", valid_depth=c(valid), target_depths=c(depths),
                                 target_std=c(dstd), gsam_only=False, **flags)
        ol = OL.snerf_loss(res, c(rgbs)) + OL.depth_loss(res, c(depths[:, 0]), c(depths[:, 1]), c(valid), c(dstd), 10.0)
        ol.backward()
        return float(ol.detach()), {k: (v.grad.double() if v.grad is not None else None) for k, v in p.items()}

    # The guided depths of pass 2 depend on pass 1's densities: an fp32 rounding difference there (1e-6 of a depth) reaches
    # the 2^9 octave of the encoding and comes back as 1e-4 of the loss - in ANY fp32 evaluation.  The criterion is therefore the
    # one of the render tests (test_gpu_parity.py:604): against an fp64 evaluation of the same algorithm on the same draws the
    # step must be no further off than 3x the fp32 oracle is.
    l64, g64 = oracle(torch.float64)
    l32, g32 = oracle(torch.float32)
    lg = float(loss)
    scale = max(float(v.abs().max()) for v in g64.values() if v is not None)
    e_mine, e_ref = (0.0, ""), (0.0, "")
    for k, v in g64.items():
        if v is None:
            assert k not in got or float(got[k].abs().max()) == 0.0, k
            continue
        e_mine = max(e_mine, (float((got[k].double() - v).abs().max()) / scale, k))
        e_ref = max(e_ref, (float((g32[k] - v).abs().max()) / scale, k))
    diag(f"lean step vs oracle autograd {name} F=512 R={R}: loss {lg:.7f}, fp32 oracle {l32:.7f}, fp64 oracle {l64:.7f}; worst gradient "
         f"difference to fp64 (of its largest entry {scale:.3e}): step {e_mine[0]:.2e} ({e_mine[1]}), fp32 oracle {e_ref[0]:.2e} ({e_ref[1]})")
    assert abs(lg - l64) <= 3 * abs(l32 - l64) + 2e-6 * abs(l64), (lg, l32, l64)
    assert e_mine[0] <= 3 * e_ref[0] + 2e-5, (e_mine, e_ref)
    # and in absolute terms: config 2 to 1e-5 / 5e-4 (measured 1e-7 / 3e-6); config 3's analytic normals put ANY fp32 evaluation
    # 2e-2 from fp64 in the first layer's gradient (measured: step 1.90e-2, fp32 oracle 1.96e-2)
    if name == "c2_lambert":
        assert abs(lg - l64) <= 1e-5 * abs(l64) and e_mine[0] <= 5e-4, (lg, l64, e_mine)
    else:
        assert abs(lg - l64) <= 1e-4 * abs(l64) and e_mine[0] <= 6e-2, (lg, l64, e_mine)


def test_lean_step_mixes_with_the_general_path():
    """A schedule that leaves the lean path (a stage taken on the general path) and comes back: gradients are cleared where they have to be and
    the Adam step counts stay in step on host and device."""
    from test_gpu_parity import build_model, make_args
    from brdf_nerf_amd import functions as Fn
    from brdf_nerf_amd.trainer import FusedTrainer
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16)
    args = make_args(cfg)
    g = torch.Generator().manual_seed(5)
    R = 48
    rays, rgbs = _sat_rays(R, g).to(DEV), torch.rand(R, 3, generator=g).to(DEV)
    torch.manual_seed(2)
    tr = FusedTrainer(build_model(cfg, 4), args, lr=5e-4, strict_rng=False)
    tr.use_graph = False
    seq = [False, False, True, False, True, True, False]
    for gs in seq:       # (round 4: the gsam_only stage runs on the lean path too: the general path is forced for it here)
        tr.lean = not gs
        loss, _ = tr.step(rays, rgbs, gsam_only=gs)
        assert np.isfinite(float(loss))
    assert tr.adam_steps["base"] == len(seq)
    assert Fn.state_views(tr.state)[2].tolist()[0] == len(seq) or tr._state_adam[0] == len(seq)
    assert int(Fn.state_views(tr.state)[0]) == seq.count(False)


@pytest.mark.parametrize("name", ["lambert", "rpv111_nan"])
def test_graph_replay_survives_batches_of_another_shape(name):
    """a captured step bakes the addresses of the trainer's scratch into its graph.  An epoch ends in a SHORT
    batch (raytable.py); the full batches of the next epoch replay the graph captured before it.  The short step must not free
    (or re-use) what the graph writes: full, short, full batches with use_graph give the parameters of the eager run, bit for bit
    (deterministic gradients; the draws are keyed by (seed, step))."""
    from test_gpu_parity import build_model, make_args
    from brdf_nerf_amd.trainer import FusedTrainer
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16, **_lean_cfgs()[name])
    brdf = name != "lambert"
    flags = dict(apply_brdf=brdf, apply_theta=brdf, cos_irra_on=brdf)
    args = make_args(cfg)
    g = torch.Generator().manual_seed(11)
    R, Rs = 64, 24
    rays, rgbs = _sat_rays(R, g).to(DEV), torch.rand(R, 3, generator=g).to(DEV)
    rays_s, rgbs_s = rays[:Rs].contiguous(), rgbs[:Rs].contiguous()          # own storage: other addresses, another shape
    seq = ["full"] * 6 + ["short"] * 2 + ["full"] * 3 + ["short"] + ["full"] * 2

    def run(graph):
        torch.manual_seed(3)
        tr = FusedTrainer(build_model(cfg, 4), args, lr=5e-4, strict_rng=False)
        tr.use_graph, tr.graph_after = graph, 2
        held = []
        for i, kind in enumerate(seq):
            a, b = (rays, rgbs) if kind == "full" else (rays_s, rgbs_s)
            loss, _ = tr.step(a, b, near_far=(0.0, 2.0), **flags)
            assert np.isfinite(float(loss)), (i, kind)
            if kind == "short":
                # what a caching allocator hands out next is what the short step's re-allocation would have freed: occupy it
                held.append(torch.full((1 << 20,), float("nan"), device=DEV))
        return tr.flat_param.clone(), len(tr._graphs)

    p_eager, _ = run(False)
    p_graph, n_graphs = run(True)
    assert n_graphs >= 2, n_graphs                                           # the full AND the short signature were captured
    assert torch.equal(p_eager, p_graph), float((p_eager - p_graph).abs().max())


def test_decaying_noise_std_keeps_the_step_on_its_graph():
    """--noise_std is multiplied by 0.9 after every step (main.py:246, schedule.py).  The kernels read it from the
    device step state (ABI 7: bn_noise.noise_std < 0), so the step keeps ONE signature while the noise decays - it is captured
    once and replayed - and the replayed steps equal the eager ones bit for bit."""
    from test_gpu_parity import build_model, make_args
    from brdf_nerf_amd.trainer import FusedTrainer
    cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16)
    g = torch.Generator().manual_seed(5)
    R = 48
    rays, rgbs = _sat_rays(R, g).to(DEV), torch.rand(R, 3, generator=g).to(DEV)

    def run(graph):
        args = make_args(cfg)
        args.noise_std = 1.0
        torch.manual_seed(2)
        tr = FusedTrainer(build_model(cfg, 4), args, lr=5e-4, strict_rng=False)
        tr.use_graph, tr.graph_after = graph, 2
        for _ in range(10):
            tr.step(rays, rgbs, near_far=(0.0, 2.0))
            args.noise_std *= 0.9
        return tr.flat_param.clone(), len(tr._graphs), len(tr._sig_seen)

    p_eager, _, _ = run(False)
    p_graph, n_graphs, n_sigs = run(True)
    assert n_graphs == 1 and n_sigs == 1, (n_graphs, n_sigs)
    assert torch.equal(p_eager, p_graph), float((p_eager - p_graph).abs().max())
    # and the noise acts: the same run without it ends elsewhere
    args0 = make_args(cfg)
    torch.manual_seed(2)
    tr0 = FusedTrainer(build_model(cfg, 4), args0, lr=5e-4, strict_rng=False)
    for _ in range(10):
        tr0.step(rays, rgbs, near_far=(0.0, 2.0))
    assert not torch.equal(tr0.flat_param, p_graph)
