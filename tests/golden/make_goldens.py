#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation on CPU.

Run only in the build container (needs /root/reference, read-only):
    python tests/golden/make_goldens.py
Writes small .npz fixtures next to this file.  Nothing of the reference is copied: the script
imports it, feeds seeded inputs and records inputs/outputs/gradients.

Model parameters are NOT stored: they are regenerated from `FieldConfig.make_params(seed)`
(numpy PCG64, deterministic) both here and in the tests; a float64 checksum guards drift.
Random draws made by the reference (torch.rand / rand_like / randn) are recorded in call order
and stored, so the build can replay them (RNG protocol).
"""
import argparse
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.config import FieldConfig  # noqa: E402

REF = "/root/reference"


def import_reference():
    for name in ["torchvision", "torchvision.transforms", "cv2", "rasterio", "kornia", "kornia.losses"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["cv2"].COLORMAP_RAINBOW = 4
    sys.modules["kornia.losses"].ssim = lambda *a, **k: None
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["kornia"].losses = sys.modules["kornia.losses"]
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import rendering, metrics, train_utils  # noqa
        import models as ref_models
        import BRDF.RPV as rpv, BRDF.Hapke as hpk, BRDF.microfacet as mcf  # noqa
    return dict(rendering=rendering, metrics=metrics, models=ref_models, rpv=rpv, hpk=hpk, mcf=mcf,
                train_utils=train_utils)


def ref_args(cfg: FieldConfig):
    return argparse.Namespace(
        model="spsbrdf-nerf", fc_layers=cfg.layers, fc_feat=cfg.feat, mapping=cfg.mapping, siren=int(cfg.siren),
        t_embbeding_tau=getattr(cfg, "t_dim", 4), beta=bool(getattr(cfg, "beta", False)), roughness=cfg.roughness, normal=cfg.normal, indirect_light=False,
        glossy_scale=1.0, sun_v=cfg.sun_v, MultiBRDF=int(cfg.MultiBRDF), dim_RPV=cfg.dim_RPV,
        input_viewdir=int(getattr(cfg, "input_viewdir", 0)),
        funcM=cfg.funcM, funcF=cfg.funcF, funcH=cfg.funcH, b=cfg.b, c=cfg.c, theta=cfg.theta, B0=0, h=0,
        shell_hapke=cfg.shell_hapke, hpk_scl=cfg.hpk_scl, guided_samples=cfg.guided_samples,
        n_samples=cfg.n_samples, n_importance=0, std_range=cfg.std_range, data=cfg.data, sc_lambda=0.0,
        chunk=5120, noise_std=cfg.noise_std, margin=0.0001, stdscale=1, fresnel_f0=cfg.fresnel_f0)


def build_ref_model(ref, cfg, seed, dtype=torch.float32):
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref["models"].load_model(ref_args(cfg))
    params = cfg.make_params(seed)
    sd = {k: torch.from_numpy(v) for k, v in params.items()}
    assert set(sd) == set(model.state_dict()), (sorted(sd), sorted(model.state_dict()))
    model.load_state_dict(sd)
    model = model.to(dtype)
    csum = float(sum(v.astype(np.float64).sum() for v in params.values()))
    return model, csum


class RecordRandoms:
    """Record every torch.rand / rand_like / randn the reference draws, in order."""

    def __init__(self, gen):
        self.gen, self.log = gen, []

    def __enter__(self):
        self._o = (torch.rand, torch.rand_like, torch.randn)
        gen, log = self.gen, self.log

        def rand(*size, **kw):
            size = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
            t = self._o[0](tuple(size), generator=gen, dtype=kw.get("dtype", torch.float32))
            log.append(t.clone())
            return t

        def rand_like(x, **kw):
            t = self._o[0](tuple(x.shape), generator=gen, dtype=x.dtype)
            log.append(t.clone())
            return t

        def randn(*size, **kw):
            size = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
            t = self._o[2](tuple(size), generator=gen, dtype=kw.get("dtype", torch.float32))
            log.append(t.clone())
            return t

        torch.rand, torch.rand_like, torch.randn = rand, rand_like, randn
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like, torch.randn = self._o


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def npify(d):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            out[k] = v.detach().cpu().numpy()
        elif v is not None:
            out[k] = np.asarray(v)
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **npify(arrays))
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KB")


# ------------------------------------------------------------------------------ configs
def mini(**kw):
    base = dict(feat=64, n_samples=16, guided_samples=16)
    base.update(kw)
    return FieldConfig(**base)


CONFIGS = {
    "lambert": dict(),
    "rpv111_nan": dict(funcM=1, funcF=1, funcH=1, normal="analystic"),
    "rpv111_nlr": dict(funcM=1, funcF=1, funcH=1, normal="learned"),
    "hapke_bc": dict(b=1, c=1, normal="analystic"),
    "hapke_bct": dict(b=1, c=1, theta=1, normal="learned"),
    "microfacet": dict(roughness=True, normal="learned"),
}


def sat_rays(R, seed, dtype=torch.float32):
    """Satellite-shaped synthetic rays: constant near/far/sun per batch."""
    g = torch.Generator().manual_seed(seed)
    o = torch.cat([torch.rand(R, 2, generator=g) * 2 - 1, 1.0 + 0.02 * torch.rand(R, 1, generator=g)], -1)
    el = np.deg2rad(75.0)
    d = torch.tensor([np.cos(el) * 0.6, np.cos(el) * 0.8, -np.sin(el)]).float().expand(R, 3)
    near = torch.zeros(R, 1)
    far = torch.full((R, 1), 2.0)
    se, sa = np.deg2rad(55.0), np.deg2rad(130.0)
    sun = torch.tensor([np.cos(se) * np.cos(sa), np.cos(se) * np.sin(sa), np.sin(se)]).float().expand(R, 3)
    return torch.cat([o, d, near, far, sun], -1).to(dtype).contiguous()


def gen_field(ref):
    for name, kw in CONFIGS.items():
        cfg = mini(**kw)
        model, csum = build_ref_model(ref, cfg, seed=11)
        g = torch.Generator().manual_seed(5)
        xyz = torch.rand(257, 3, generator=g) * 2 - 1
        nr_an = cfg.normal in ("analystic", "analystic_learned")
        nr_lr = cfg.normal in ("learned", "analystic_learned")
        out = quiet(model, xyz.clone(), apply_brdf=True, apply_theta=True, nr_an_on=nr_an, nr_lr_on=nr_lr)
        out0 = quiet(model, xyz.clone(), apply_brdf=False, nr_an_on=nr_an, nr_lr_on=nr_lr)
        sig = quiet(model, xyz.clone(), sigma_only=True)
        save(f"field_{name}_F64", xyz=xyz, out_brdf=out, out_nobrdf=out0, sigma=sig, param_checksum=csum,
             param_seed=11)
    cfg = FieldConfig(**CONFIGS["rpv111_nan"])
    model, csum = build_ref_model(ref, cfg, seed=12)
    g = torch.Generator().manual_seed(6)
    xyz = torch.rand(33, 3, generator=g) * 2 - 1
    out = quiet(model, xyz.clone(), apply_brdf=True, nr_an_on=True)
    save("field_rpv111_nan_F512", xyz=xyz, out_brdf=out, param_checksum=csum, param_seed=12)
    # fp64 adjoint-chain check value for d sigma / d xyz
    model64, _ = build_ref_model(ref, mini(**CONFIGS["rpv111_nan"]), seed=11, dtype=torch.float64)
    x64 = (torch.rand(17, 3, generator=torch.Generator().manual_seed(7)) * 2 - 1).double()
    grad = quiet(model64.calc_normals, x64.clone(), graph=False)
    save("field_sigma_grad_F64_fp64", xyz=x64, grad=grad, param_seed=11)


def gen_field_variants(ref):
    """--siren 0 (ReLU trunk and heads, default init), no --mapping (raw xyz into the trunk) and --dim_RPV 3 (three-channel
    RPV heads), forward values and parameter gradients of a random linear functional."""
    for tag, kw in (("relu", dict(siren=False, funcM=1, funcF=1, funcH=1, normal="learned")),
                    ("nomap", dict(mapping=False, funcM=1, funcF=1, funcH=1, normal="learned")),
                    ("rpv333", dict(dim_RPV=3, funcM=1, funcF=1, funcH=1, normal="learned"))):
        cfg = mini(**kw)
        model, csum = build_ref_model(ref, cfg, seed=13)
        g = torch.Generator().manual_seed(9)
        xyz = torch.rand(201, 3, generator=g) * 2 - 1
        out = quiet(model, xyz.clone(), apply_brdf=True, apply_theta=True, nr_an_on=False, nr_lr_on=True)
        coef = torch.randn(out.shape, generator=g)
        (out * coef).sum().backward()
        grads = {f"grad/{k}": (p_.grad if p_.grad is not None else torch.zeros_like(p_)) for k, p_ in model.named_parameters()}
        save(f"field_{tag}_F64", xyz=xyz, out_brdf=out, coef=coef, param_checksum=csum, param_seed=13, **grads)


def gen_composite(ref):
    cal_weight = sys.modules["models.spsbrdfnerf"].cal_weight
    for S in (16, 128):
        g = torch.Generator().manual_seed(S)
        R = 8
        z = torch.sort(torch.rand(R, S, generator=g) * 2.0, -1)[0]
        sigma = (torch.randn(R, S, generator=g) * 3.0).requires_grad_(True)     # includes negatives (relu)
        with RecordRandoms(torch.Generator().manual_seed(1)):
            a, T, w, d = quiet(cal_weight, z, sigma, argparse.Namespace(noise_std=0.0))
        cw = torch.rand(R, S, generator=g)
        cd = torch.rand(R, generator=g)
        ((w * cw).sum() + (d * cd).sum()).backward()
        save(f"composite_S{S}", z=z, sigma=sigma, alphas=a, transparency=T, weights=w, depth=d, cw=cw, cd=cd,
             dsigma=sigma.grad)


def gen_guided(ref):
    rd = ref["rendering"]
    R, S, G = 50, 64, 64
    g = torch.Generator().manual_seed(3)
    near = torch.zeros(R, 1)
    far = torch.full((R, 1), 2.0)
    z = rd.get_z_vals(S, "cpu", near, far, perturb=0.0)
    z = z + (torch.rand(R, S, generator=g) - 0.5) * (2.0 / 63) * 0.9
    sigma = torch.relu(torch.randn(R, S, generator=g)) * 20 * (torch.rand(R, S, generator=g) < 0.15)
    sigma[0] = 0.0                                # empty ray: weight all on the last sample
    sigma[1] = 50.0                               # opaque at the first sample (std -> ~0)
    cal_weight = sys.modules["models.spsbrdfnerf"].cal_weight
    with RecordRandoms(torch.Generator().manual_seed(1)):
        a, T, w, d = quiet(cal_weight, z, sigma, argparse.Namespace(noise_std=0.0))
    res = {"depth": d, "weights": w}
    valid = (torch.rand(R, generator=g) < 0.6).float()
    tdep = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1)
    tstd = 0.01 + 0.05 * torch.rand(R, generator=g)
    for mode in ("test", "train"):
        with RecordRandoms(torch.Generator().manual_seed(9)) as rec:
            kw = dict(mode=mode, device="cpu", dRange=3.0)
            if mode == "train":
                kw.update(valid_depth=valid, target_depths=tdep, target_std=tstd)
            z2 = quiet(rd.GenerateGuidedSamples, res, z, G, 1.0, near, far, **kw)
        z2s = torch.sort(z2, -1)[0]
        z_unsort = torch.cat([z, z2s], -1)
        z_all, idx = torch.sort(z_unsort, -1)
        extra = {f"rand{i}": t for i, t in enumerate(rec.log)}
        save(f"guided_{mode}", z=z, depth=d, weights=w, valid_depth=valid, target_depths=tdep, target_std=tstd,
             z2=z2, z2_sorted=z2s, z_all=z_all, sort_idx=idx, **extra)


def brdf_inputs(N, seed):
    g = torch.Generator().manual_seed(seed)
    def unit(x):
        return x / x.norm(dim=-1, keepdim=True)
    n = unit(torch.randn(N, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.0, 1.0]))
    l = unit(torch.randn(N, 3, generator=g) * 0.5 + torch.tensor([0.3, 0.2, 0.8]))
    v = unit(torch.randn(N, 3, generator=g) * 0.5 + torch.tensor([-0.2, 0.1, 0.9]))
    # grazing / back-facing rows (clamp to 1e-5), coincident l==v, l==n
    n[0] = torch.tensor([1.0, 0.0, 0.0]); l[0] = torch.tensor([0.0, 0.0, 1.0])
    v[1] = -n[1]
    l[2] = v[2]
    l[3] = n[3]
    v[4] = n[4]
    w = torch.rand(N, 3, generator=g)
    return l, v, n, w, g


def gen_brdf(ref):
    N = 64
    l, v, n, w, g = brdf_inputs(N, 21)
    coef = torch.rand(N, 3, generator=g)
    # RPV
    k = torch.rand(N, 3, generator=g) * 2
    th = torch.rand(N, 3, generator=g) * 2 - 1
    rc = torch.rand(N, 3, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (n, w, k, th, rc)]
    brdf, M1, G, H, ci, cv = quiet(ref["rpv"].RPV(), l.unsqueeze(1), v, leaves[0], *leaves[1:])
    (brdf * coef).sum().backward()
    save("brdf_rpv", l=l, v=v, n=n, w=w, k=k, theta=th, rhoc=rc, coef=coef, brdf=brdf, M1=M1, G=G, H=H, ci=ci, cv=cv,
         dn=leaves[0].grad, dw=leaves[1].grad, dk=leaves[2].grad, dtheta=leaves[3].grad, drhoc=leaves[4].grad)
    # Hapke variants
    b = torch.rand(N, 3, generator=g)
    c = torch.rand(N, 3, generator=g)
    tht = torch.rand(N, generator=g) * (np.pi * 30 / 180)
    for tag, use_c, use_t, shell in (("hapke_b", 0, 0, 0), ("hapke_bc", 1, 0, 0), ("hapke_bct", 1, 1, 0),
                                     ("hapke_shell1", 0, 0, 1), ("hapke_shell2", 0, 0, 2), ("hapke_shell3", 0, 0, 3)):
        args = argparse.Namespace(hpk_scl=4.0, shell_hapke=shell)
        ln, lw, lb, lc, lt = [t.clone().requires_grad_(True) for t in (n, w, b, c, tht)]
        o = quiet(ref["hpk"].Hapke(args=args), l.unsqueeze(1), v, ln, lw, None if shell else lb,
                  lc if use_c else None, lt if use_t else None, None, None)
        brdf, P, Bf, Hi, Hv, Sh, ci, cv = o
        (brdf * coef).sum().backward()
        z = torch.zeros(1)
        save(f"brdf_{tag}", l=l, v=v, n=n, w=w, b=b, c=c, theta=tht, coef=coef, brdf=brdf, P=P, Hi=Hi, Hv=Hv, S=Sh,
             ci=ci, cv=cv, dn=ln.grad, dw=lw.grad, db=lb.grad if lb.grad is not None else z,
             dc=lc.grad if lc.grad is not None else z, dtheta=lt.grad if lt.grad is not None else z)
    # microfacet
    rough = torch.rand(N, 1, generator=g) * 0.9 + 0.05
    ln, lw, lr = [t.clone().requires_grad_(True) for t in (n, w, rough)]
    gl, brdf, f, gg, d, ldn, vdn, h, n_h = quiet(ref["mcf"].Microfacet(f0=0.04, lvis=False), l.unsqueeze(1), v, ln,
                                                 albedo=lw, rough=lr)
    (brdf.reshape(N, 3) * coef).sum().backward()
    save("brdf_microfacet", l=l, v=v, n=n, w=w, rough=rough, coef=coef, glossy=gl, brdf=brdf.reshape(N, 3), f=f, g=gg,
         d=d, l_dot_n=ldn, v_dot_n=vdn, h=h.reshape(N, 3), n_h=n_h, dn=ln.grad, dw=lw.grad, drough=lr.grad)


def run_render(ref, cfg, model, rays, mode, flags, targets=None, seed=2):
    kw = dict(flags)
    if mode == "train" and targets is not None:
        kw.update(valid_depth=targets["valid_depth"], target_depths=targets["depths"], target_std=targets["depth_std"])
    with RecordRandoms(torch.Generator().manual_seed(seed)) as rec:
        res, brdf_type = quiet(ref["rendering"].render_rays, {"coarse": model}, ref_args(cfg), rays, None, mode=mode,
                               **kw)
    return res, brdf_type, rec.log


def gen_render(ref):
    R = 64
    rays = sat_rays(R, 1)
    g = torch.Generator().manual_seed(4)
    targets = dict(rgbs=torch.rand(R, 3, generator=g), valid_depth=(torch.rand(R, generator=g) < 0.7).float(),
                   depths=torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1),
                   depth_std=0.02 + 0.05 * torch.rand(R, generator=g))
    for name, kw in CONFIGS.items():
        cfg = mini(**kw)
        flags = dict(apply_brdf=name != "lambert", apply_theta=True, cos_irra_on=name != "lambert")
        for mode in ("train", "test"):
            model, csum = build_ref_model(ref, cfg, seed=11)
            res, brdf_type, rlog = run_render(ref, cfg, model, rays, mode, flags, targets)
            arrays = {f"out/{k}": v for k, v in res.items()}
            arrays.update({f"rand{i}": t for i, t in enumerate(rlog)})
            if mode == "train":
                loss = torch.mean((res["rgb_coarse"] - targets["rgbs"]) ** 2) + 0.01 * torch.mean(res["depth_coarse"])
                loss.backward()
                arrays["loss"] = loss
                for k, p in model.named_parameters():
                    arrays[f"grad/{k}"] = p.grad if p.grad is not None else torch.zeros_like(p)
            save(f"render_{name}_{mode}", rays=rays, brdf_type=np.array(brdf_type), param_checksum=csum, param_seed=11,
                 **{f"tgt/{k}": v for k, v in targets.items()}, **arrays)
    # per-sample BRDF (MultiBRDF=1) and guided-samples-only (gsam_only) variants, test mode
    # (Hapke + theta with MultiBRDF raises IndexError inside the reference's mu0_eff: not a usable configuration)
    for tag, extra, gs in (("rpv111_nlr_multibrdf", dict(MultiBRDF=True), False), ("rpv111_nlr_gsamonly", dict(), True)):
        base = "rpv111_nlr" if tag.startswith("rpv") else "hapke_bct"
        cfg = mini(**dict(CONFIGS[base], **extra))
        model, csum = build_ref_model(ref, cfg, seed=11)
        res, brdf_type, rlog = run_render(ref, cfg, model, rays, "test", dict(apply_brdf=True, apply_theta=True, cos_irra_on=True,
                                                                               gsam_only=gs))
        arrays = {f"out/{k}": v for k, v in res.items()}
        arrays.update({f"rand{i}": t for i, t in enumerate(rlog)})
        save(f"render_{tag}_test", rays=rays, brdf_type=np.array(brdf_type), param_checksum=csum, param_seed=11, **arrays)
    gen_render_sunv(ref, rays, targets)
    # full-size network, forward only
    cfg = FieldConfig(**CONFIGS["rpv111_nan"])
    model, csum = build_ref_model(ref, cfg, seed=12)
    rays8 = sat_rays(8, 8)
    res, brdf_type, rlog = run_render(ref, cfg, model, rays8, "test", dict(apply_brdf=True, apply_theta=True,
                                                                             cos_irra_on=True))
    keep = ("rgb_coarse", "depth_coarse", "weights_coarse", "z_vals_coarse", "sort_idx_coarse", "sigmas_coarse",
            "albedo_coarse", "normal_an_coarse", "rpv_k_coarse", "rpv_theta_coarse", "rpv_rhoc_coarse")
    arrays = {f"out/{k}": res[k] for k in keep}
    arrays.update({f"rand{i}": t for i, t in enumerate(rlog)})
    save("render_rpv111_nan_F512", rays=rays8, brdf_type=np.array(brdf_type), param_checksum=csum, param_seed=12,
         **arrays)
    # blender-shaped rays (R,8): sun_d = ones (rendering.py:189), BASELINE config 1 plumbing case
    cfg = mini(data="blender")
    model, csum = build_ref_model(ref, cfg, seed=11)
    gb = torch.Generator().manual_seed(14)
    dirs = torch.randn(32, 3, generator=gb)
    dirs = dirs / dirs.norm(dim=-1, keepdim=True)
    raysb = torch.cat([torch.randn(32, 3, generator=gb) * 0.1 + torch.tensor([0.0, 0.0, 4.0]), dirs,
                       torch.full((32, 1), 2.0), torch.full((32, 1), 6.0)], -1)
    res, brdf_type, rlog = run_render(ref, cfg, model, raysb, "test", dict())
    arrays = {f"out/{k}": v for k, v in res.items()}
    arrays.update({f"rand{i}": t for i, t in enumerate(rlog)})
    save("render_lambert_blender", rays=raysb, brdf_type=np.array(brdf_type), param_checksum=csum, param_seed=11,
         **arrays)


BRANCHES = {     # round 3: branches of inference() that no fixture reached ("unpinned branches")
    # funcH == 2: rhoc := albedo, no rhoc head (models/spsbrdfnerf.py:306,317), per ray and per sample
    "rpv_m1f1h2": (dict(funcM=1, funcF=1, funcH=2, normal="learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
    "rpv_m1f1h2_multibrdf": (dict(funcM=1, funcF=1, funcH=2, normal="learned", MultiBRDF=True),
                             dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
    "rpv_m1h2": (dict(funcM=1, funcH=2, normal="learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
    # shell_hapke > 0 shades with the Hapke shell even with apply_brdf=False (:320,348,383)
    "shell1_nobrdf": (dict(shell_hapke=1, normal="learned"), dict(apply_brdf=False, apply_theta=False, cos_irra_on=False)),
    "shell2_nobrdf": (dict(shell_hapke=2, normal="learned"), dict(apply_brdf=False, apply_theta=False, cos_irra_on=True)),
    "shell3_nobrdf": (dict(shell_hapke=3, normal="learned"), dict(apply_brdf=False, apply_theta=False, cos_irra_on=True)),
    "shell3_brdf": (dict(shell_hapke=3, normal="learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
}


def gen_branches(ref):
    R = 48
    rays = sat_rays(R, 3)
    for name, (kw, flags) in BRANCHES.items():
        cfg = mini(**kw)
        model, csum = build_ref_model(ref, cfg, seed=11)
        res, brdf_type, rlog = run_render(ref, cfg, model, rays, "test", flags)
        arrays = {f"out/{k}": v for k, v in res.items()}
        arrays.update({f"rand{i}": t for i, t in enumerate(rlog)})
        save(f"render_{name}_test", rays=rays, brdf_type=np.array(brdf_type), param_checksum=csum, param_seed=11, **arrays)
    # rows / cols -> ref_sphere (:404-412; visualisation of the view sphere in validation, main.py:451-555)
    g = torch.Generator().manual_seed(6)
    rows, cols = (torch.rand(R, 1, generator=g) * 1.2 - 0.6), (torch.rand(R, 1, generator=g) * 1.2 - 0.6)
    cfg = mini(funcM=1, funcF=1, funcH=1, normal="learned")
    model, csum = build_ref_model(ref, cfg, seed=11)
    res, brdf_type, rlog = run_render(ref, cfg, model, rays, "test", dict(apply_brdf=True, apply_theta=True, cos_irra_on=True,
                                                                            rows=rows, cols=cols))
    arrays = {f"out/{k}": v for k, v in res.items()}
    arrays.update({f"rand{i}": t for i, t in enumerate(rlog)})
    save("render_rpv111_nlr_refsphere_test", rays=rays, rows=rows, cols=cols, brdf_type=np.array(brdf_type), param_checksum=csum,
         param_seed=11, **arrays)


def gen_render_sunv(ref, rays=None, targets=None):
    """Sun-visibility pass (--sun_v analystic, rendering.py:244-259) with gsam_only=True - the only combination the
    reference's pass 2 accepts - and cos_irra_on=False so the visibility IS the irradiance."""
    R = 64
    if rays is None:
        rays = sat_rays(R, 1)
        g = torch.Generator().manual_seed(4)
        targets = dict(rgbs=torch.rand(R, 3, generator=g), valid_depth=(torch.rand(R, generator=g) < 0.7).float(),
                       depths=torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1),
                       depth_std=0.02 + 0.05 * torch.rand(R, generator=g))
    for base in ("rpv111_nlr", "lambert"):
        cfg = mini(**dict(CONFIGS[base], sun_v="analystic"))
        for mode in ("train", "test"):
            model, csum = build_ref_model(ref, cfg, seed=11)
            res, brdf_type, rlog = run_render(ref, cfg, model, rays, mode,
                                              dict(apply_brdf=True, apply_theta=True, cos_irra_on=False, gsam_only=True), targets)
            arrays = {f"out/{k}": v for k, v in res.items()}
            arrays.update({f"rand{i}": t for i, t in enumerate(rlog)})
            if mode == "train":
                loss = ((res["rgb_coarse"] - targets["rgbs"]) ** 2).mean()
                loss.backward()
                for k, p_ in model.named_parameters():
                    arrays[f"grad/{k}"] = p_.grad if p_.grad is not None else torch.zeros_like(p_)
            save(f"render_{base}_sunv_{mode}", rays=rays, brdf_type=np.array(brdf_type), param_checksum=csum, param_seed=11,
                 **{f"tgt/{k}": v for k, v in targets.items()}, **arrays)


def gen_loss(ref):
    R, S = 64, 32
    g = torch.Generator().manual_seed(31)
    z = torch.sort(torch.rand(R, S, generator=g) * 2, -1)[0]
    w = torch.softmax(torch.randn(R, S, generator=g) * 2, -1).requires_grad_(True)
    depth = (w * z).sum(-1).detach().clone().requires_grad_(True)
    rgb = torch.rand(R, 3, generator=g).requires_grad_(True)
    tgt = torch.rand(R, 3, generator=g)
    valid = (torch.rand(R, generator=g) < 0.7).float()
    tdep = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1)
    tstd = 0.02 + 0.3 * torch.rand(R, generator=g)
    inputs = {"rgb_coarse": rgb, "depth_coarse": depth, "weights_coarse": w, "z_vals_coarse": z}
    m = ref["metrics"]
    l_rgb, _ = quiet(m.SNerfLoss(lambda_sc=0.0), inputs, tgt)
    dl = quiet(m.DepthLoss, lambda_ds=10.0, GNLL=False, usealldepth=False, margin=0.0001, stdscale=1, subset=True)
    l_ds, _ = quiet(dl, inputs, tdep[:, 0], tdep[:, 1], target_valid_depth=valid, target_std=tstd)
    (l_rgb + l_ds).backward()
    psnr = quiet(m.psnr, rgb.detach(), tgt)
    save("loss_snerf_depth", z=z, weights=w, depth=depth, rgb=rgb, tgt=tgt, valid_depth=valid, target_depths=tdep,
         target_std=tstd, loss_rgb=l_rgb, loss_ds=l_ds, drgb=rgb.grad, ddepth=depth.grad,
         dweights=w.grad if w.grad is not None else torch.zeros_like(w), psnr=torch.as_tensor(psnr[0]))


def gen_regularisers(ref):
    """NormalRegLoss, HardSurfaceLoss, NormalLoss (metrics.py:179-290) on seeded per-sample tensors, with gradients."""
    R, S = 48, 24
    g = torch.Generator().manual_seed(77)
    z = torch.sort(torch.rand(R, S, generator=g) * 2, -1)[0]
    w = torch.softmax(torch.randn(R, S, generator=g) * 2, -1).requires_grad_(True)
    depth = (w * z).sum(-1).detach().clone().requires_grad_(True)
    unit = lambda t: t / t.norm(dim=-1, keepdim=True)
    n_an = unit(torch.randn(R, S, 3, generator=g)).requires_grad_(True)
    n_lr = unit(torch.randn(R, S, 3, generator=g)).requires_grad_(True)
    view = unit(torch.randn(R, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.0, 1.0]))
    n_gt = unit(torch.randn(R, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.0, 1.0]))
    valid = (torch.rand(R, generator=g) < 0.7).float()
    tw = torch.rand(R, generator=g)
    inputs = {"normal_an_coarse": n_an, "normal_lr_coarse": n_lr, "weights_coarse": w, "rays_d_coarse": view,
              "z_vals_coarse": z, "depth_coarse": depth}
    m = ref["metrics"]
    l_an, _, perc_an = quiet(m.NormalRegLoss(lambda_nr_reg=0.1, keyword="normal_an"), inputs)
    l_lr, _, perc_lr = quiet(m.NormalRegLoss(lambda_nr_reg=0.05, keyword="normal_lr"), inputs)
    l_hs, _ = quiet(m.HardSurfaceLoss(lambda_hs=0.5), inputs)
    nl = m.NormalLoss(lambda_nr_spv=0.01)
    l_n1, _ = quiet(nl, w, n_an, n_lr, keyword="an_lr")
    l_n3, _ = quiet(nl, w, n_gt, n_an, target_weight=tw, target_valid_depth=valid, keyword="an")
    total = l_an + l_lr + l_hs + l_n1 + l_n3
    total.backward()
    save("loss_regularisers", z=z, weights=w, depth=depth, normal_an=n_an, normal_lr=n_lr, view=view, normal_gt=n_gt,
         valid_depth=valid, target_weight=tw, l_nr_an=l_an, l_nr_lr=l_lr, perc_an=torch.as_tensor(float(perc_an)),
         perc_lr=torch.as_tensor(float(perc_lr)), l_hs=l_hs, l_n1=l_n1, l_n3=l_n3, d_weights=w.grad, d_depth=depth.grad,
         d_normal_an=n_an.grad, d_normal_lr=n_lr.grad)


def gen_viewdir(ref):
    """--input_viewdir 1 (spsbrdfnerf.py:458,689-692): the rgb head reads cat([xyz_features, mapping[1](view dir)]).
    Field forward with per-point directions + parameter gradients of a random linear functional (with and without --mapping),
    and the full render_rays dict (train mode, RPV + learned normals)."""
    for tag, kw in (("viewdir", dict(input_viewdir=1, funcM=1, funcF=1, funcH=1, normal="learned")),
                    ("viewdir_nomap", dict(input_viewdir=1, mapping=False))):
        cfg = mini(**kw)
        model, csum = build_ref_model(ref, cfg, seed=14)
        g = torch.Generator().manual_seed(10)
        xyz = torch.rand(203, 3, generator=g) * 2 - 1
        dirs = torch.nn.functional.normalize(torch.randn(203, 3, generator=g), dim=-1)
        nlr = cfg.normal in ("learned", "analystic_learned")
        out = quiet(model, xyz.clone(), input_dir=dirs, apply_brdf=True, apply_theta=True, nr_an_on=False, nr_lr_on=nlr)
        coef = torch.randn(out.shape, generator=g)
        (out * coef).sum().backward()
        grads = {f"grad/{k}": (p_.grad if p_.grad is not None else torch.zeros_like(p_)) for k, p_ in model.named_parameters()}
        save(f"field_{tag}_F64", xyz=xyz, dirs=dirs, out_brdf=out, coef=coef, param_checksum=csum, param_seed=14, **grads)
    cfg = mini(input_viewdir=1, funcM=1, funcF=1, funcH=1, normal="learned")
    model, csum = build_ref_model(ref, cfg, seed=11)
    rays = sat_rays(64, 21)
    rays[32:, 3:6] = torch.nn.functional.normalize(torch.tensor([-0.2, 0.1, -0.95]), dim=0)      # a second view direction
    res, bt, rlog = run_render(ref, cfg, model, rays, "train", dict(apply_brdf=True, apply_theta=True, cos_irra_on=True))
    tgt = torch.rand(64, 3, generator=torch.Generator().manual_seed(5))
    loss = ((res["rgb_coarse"] - tgt) ** 2).mean() + 0.01 * res["depth_coarse"].mean()
    loss.backward()
    arrays = {f"out/{k}": v for k, v in res.items()}
    arrays.update({f"rand{i}": t for i, t in enumerate(rlog)})
    arrays.update({f"grad/{k}": (p_.grad if p_.grad is not None else torch.zeros_like(p_)) for k, p_ in model.named_parameters()})
    save("render_viewdir_train", rays=rays, targets=tgt, loss=loss, brdf_type=np.array(bt), param_checksum=csum, param_seed=11,
         **arrays)


def gen_beta(ref):
    """--beta (spsbrdfnerf.py:571-575,708-711; rendering.py:226-229): one more output channel after sigma,
    beta = Softplus(Linear(Siren(Linear(cat([xyz_features, t embedding]))))).  Field forward with per-point embeddings and the
    gradients of a random linear functional w.r.t. every parameter AND the embedding input; the full render_rays dict in train
    mode with models['t'] = an Embedding looked up by ts (RPV + learned normals), a loss that reads beta through
    uncertainty_aware_loss (metrics.py:24-28), gradients of the parameters and of the embedding table."""
    for tag, kw in (("beta", dict(beta=True, funcM=1, funcF=1, funcH=1, normal="learned")),
                    ("beta_viewdir_relu", dict(beta=True, input_viewdir=1, siren=False, t_dim=6))):
        cfg = mini(**kw)
        model, csum = build_ref_model(ref, cfg, seed=15)
        g = torch.Generator().manual_seed(12)
        xyz = torch.rand(150, 3, generator=g) * 2 - 1
        dirs = torch.nn.functional.normalize(torch.randn(150, 3, generator=g), dim=-1)
        t_in = torch.randn(150, cfg.t_dim, generator=g).requires_grad_(True)
        nlr = cfg.normal in ("learned", "analystic_learned")
        out = quiet(model, xyz.clone(), input_dir=dirs, input_t=t_in, apply_brdf=True, apply_theta=True, nr_an_on=False,
                    nr_lr_on=nlr)
        coef = torch.randn(out.shape, generator=g)
        (out * coef).sum().backward()
        grads = {f"grad/{k}": (p_.grad if p_.grad is not None else torch.zeros_like(p_)) for k, p_ in model.named_parameters()}
        save(f"field_{tag}_F64", xyz=xyz, dirs=dirs, t_in=t_in.detach(), out_brdf=out, coef=coef, d_t_in=t_in.grad,
             param_checksum=csum, param_seed=15, **grads)
    cfg = mini(beta=True, funcM=1, funcF=1, funcH=1, normal="learned")
    model, csum = build_ref_model(ref, cfg, seed=11)
    rays = sat_rays(64, 22)
    emb = torch.nn.Embedding(5, cfg.t_dim)
    with torch.no_grad():
        emb.weight.copy_(torch.randn(5, cfg.t_dim, generator=torch.Generator().manual_seed(8)))
    ts = torch.randint(0, 5, (64,), generator=torch.Generator().manual_seed(9))
    with RecordRandoms(torch.Generator().manual_seed(2)) as rec:
        res, bt = quiet(ref["rendering"].render_rays, {"coarse": model, "t": emb}, ref_args(cfg), rays, ts, mode="train",
                        apply_brdf=True, apply_theta=True, cos_irra_on=True)
    tgt = torch.rand(64, 3, generator=torch.Generator().manual_seed(5))
    ld = ref["metrics"].uncertainty_aware_loss({}, res, tgt, "coarse")
    loss = ld["coarse_color"] + ld["coarse_logbeta"] + 0.01 * res["depth_coarse"].mean()
    loss.backward()
    arrays = {f"out/{k}": v for k, v in res.items()}
    arrays.update({f"rand{i}": t for i, t in enumerate(rec.log)})
    arrays.update({f"grad/{k}": (p_.grad if p_.grad is not None else torch.zeros_like(p_)) for k, p_ in model.named_parameters()})
    save("render_beta_train", rays=rays, ts=ts, emb=emb.weight.detach(), d_emb=emb.weight.grad, targets=tgt, loss=loss,
         loss_color=ld["coarse_color"], loss_logbeta=ld["coarse_logbeta"], brdf_type=np.array(bt), param_checksum=csum,
         param_seed=11, **arrays)


INIT_CONFIGS = {
    "lambert": dict(),
    "rpv111_anlr": dict(funcM=1, funcF=1, funcH=1, normal="analystic_learned"),
    "hapke_bct_nlr": dict(b=1, c=1, theta=1, normal="learned"),
    "microfacet_nlr": dict(roughness=True, normal="learned"),
    "relu_rpvM": dict(siren=False, funcM=1),
    "nomap": dict(mapping=False),
    "viewdir": dict(input_viewdir=1),
    "beta_nlr": dict(beta=True, normal="learned", funcM=1),
}


def gen_init(ref):
    """state_dict of the reference's load_model(args) under torch.manual_seed(7) (models/__init__.py:6-17; inits
    models/spsbrdfnerf.py:537-539, models/nerf.py:9-21): the FULL state_dict for the RPV + both-normals model (every layer
    type), and per-tensor fingerprints (float64 sum, sum of |.|, first / last 8 elements) for the other head sets.  Pins
    the drop-in module's "same seed -> same weights" contract: layer construction order and init calls consume the RNG
    stream exactly like upstream."""
    out = {}
    for name, kw in INIT_CONFIGS.items():
        cfg = mini(**kw)
        torch.manual_seed(7)
        model = quiet(ref["models"].load_model, ref_args(cfg))
        sd = model.state_dict()
        out[f"{name}/keys"] = np.array("\n".join(sd.keys()))
        for k, v in sd.items():
            f = v.detach().double().flatten()
            if name == "rpv111_anlr":
                out[f"{name}/full/{k}"] = v.detach().numpy()
            out[f"{name}/fp/{k}"] = np.concatenate([[float(f.sum()), float(f.abs().sum()), float(f.numel())],
                                                    f[:8].numpy(), f[-8:].numpy()])
    save("init_state_dicts", seed=7, **out)


if __name__ == "__main__":
    torch.set_num_threads(4)
    ref = import_reference()
    if "--only-init" in sys.argv:
        gen_init(ref)
        sys.exit(0)
    if "--only-viewdir" in sys.argv:
        gen_viewdir(ref)
        sys.exit(0)
    if "--only-field-variants" in sys.argv:
        gen_field_variants(ref)
        sys.exit(0)
    if "--only-beta" in sys.argv:
        gen_beta(ref)
        sys.exit(0)
    if "--only-sunv" in sys.argv:
        gen_render_sunv(ref)
        sys.exit(0)
    if "--only-regularisers" in sys.argv:
        gen_regularisers(ref)
        sys.exit(0)
    if "--only-render" in sys.argv:
        gen_render(ref)
        sys.exit(0)
    if "--only-branches" in sys.argv:
        gen_branches(ref)
        sys.exit(0)
    gen_field(ref)
    gen_field_variants(ref)
    gen_composite(ref)
    gen_guided(ref)
    gen_brdf(ref)
    gen_render(ref)
    gen_loss(ref)
    gen_regularisers(ref)
    gen_init(ref)
    gen_viewdir(ref)
    gen_beta(ref)
    gen_branches(ref)
