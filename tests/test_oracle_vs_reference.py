"""The oracle against the REFERENCE ITSELF on random configurations (build container only: needs /root/reference, which is
imported read-only exactly as tests/golden/make_goldens.py does; skipped everywhere else - nothing here runs on the GPU box).

The golden fixtures pin the oracle on the reference's own shapes; tests/test_gpu_fuzz.py then uses the oracle as the checker on
RANDOM shapes and flag sets.  This closes the loop: on the same family of random configurations the oracle must reproduce the
reference - field forward and every parameter gradient, and the full render_rays dictionary with the reference's own random
draws recorded and replayed."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

from conftest import tparams
from oracle.config import FieldConfig
from oracle import field as OF
from oracle import render as ORD

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is only present in the build container")


@pytest.fixture(scope="module")
def mg():
    spec = importlib.util.spec_from_file_location("make_goldens", os.path.join(os.path.dirname(__file__), "golden", "make_goldens.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.ref = m.import_reference()
    return m


def draw_config(rng):
    family = rng.choice(["lambert", "rpv", "hapke", "microfacet"])
    kw = dict(feat=int(rng.choice([32, 64])), layers=int(rng.integers(2, 9)), siren=bool(rng.random() < 0.7), mapping=bool(rng.random() < 0.8),
              normal=str(rng.choice(["none", "learned", "analystic", "analystic_learned"])),
              input_viewdir=int(rng.random() < 0.35), beta=bool(rng.random() < 0.35), n_samples=int(rng.choice([8, 16])),
              guided_samples=int(rng.choice([8, 16])))
    if family == "rpv":
        kw.update(funcM=int(rng.random() < 0.7), funcF=int(rng.random() < 0.7), funcH=int(rng.random() < 0.7), dim_RPV=int(rng.choice([1, 3])))
        if not (kw["funcM"] or kw["funcF"] or kw["funcH"]):
            kw["funcM"] = 1
    elif family == "hapke":
        kw.update(b=1, c=int(rng.random() < 0.7), theta=int(rng.random() < 0.5))
    elif family == "microfacet":
        kw.update(roughness=True)
    if family != "lambert" and kw["normal"] == "none":
        kw["normal"] = "learned"
    return FieldConfig(**kw)


@pytest.mark.parametrize("seed", list(range(16)))
def test_oracle_field_matches_reference_on_random_configurations(mg, seed):
    rng = np.random.default_rng(31000 + seed)
    cfg = draw_config(rng)
    model, _ = mg.build_ref_model(mg.ref, cfg, seed=20 + seed)
    B = int(rng.integers(1, 120))
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, 3, generator=g) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    t_in = torch.randn(B, cfg.t_dim, generator=g)
    flags = dict(apply_brdf=bool(rng.random() < 0.8), apply_theta=bool(rng.random() < 0.7),
                 nr_an_on=cfg.normal in ("analystic", "analystic_learned"), nr_lr_on=cfg.normal in ("learned", "analystic_learned"))
    t_ref = t_in.clone().requires_grad_(True)
    out_ref = mg.quiet(model, xyz.clone(), input_dir=dirs, input_t=t_ref if cfg.beta else None, **flags)
    coef = torch.randn(out_ref.shape, generator=g)
    (out_ref * coef).sum().backward()
    p = tparams(cfg, 20 + seed)
    for v in p.values():
        v.requires_grad_(True)
    t_or = t_in.clone().requires_grad_(True)
    out = OF.field_forward(p, cfg, xyz, dirs=dirs if cfg.dir_dim else None, t_embed=t_or if cfg.beta else None, **flags)
    tag = f"seed {seed}: {cfg}"
    assert out.shape == out_ref.shape, tag
    assert float((out - out_ref).detach().abs().max()) <= 2e-5 * float(out_ref.detach().abs().max()) + 2e-6, tag
    (out * coef).sum().backward()
    if cfg.beta:
        assert float((t_or.grad - t_ref.grad).abs().max()) <= 1e-4 * float(t_ref.grad.abs().max()) + 1e-9, tag
    ref_grads = dict(model.named_parameters())
    for k, v in p.items():
        gr = ref_grads[k].grad
        if gr is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, f"{tag}: {k}"
            continue
        scale = float(gr.abs().max())
        assert float((v.grad - gr).abs().max()) <= 2e-4 * scale + 1e-9, f"{tag}: {k}"


@pytest.mark.parametrize("seed", list(range(12)))
def test_oracle_render_rays_matches_reference_on_random_configurations(mg, seed):
    rng = np.random.default_rng(33000 + seed)
    cfg = draw_config(rng)
    kw = dict(vars(cfg))
    brdf = bool(cfg.roughness or cfg.RPV or cfg.b)
    gsam_only = bool(rng.random() < 0.3)
    if brdf and gsam_only and rng.random() < 0.5:
        kw["sun_v"] = "analystic"
    if cfg.RPV and rng.random() < 0.3:
        kw["MultiBRDF"] = True
    kw["noise_std"] = float(rng.choice([0.0, 0.3]))
    cfg = FieldConfig(**kw)
    R = int(rng.integers(2, 40))
    mode = "train" if rng.random() < 0.5 else "test"
    flags = dict(apply_brdf=brdf and bool(rng.random() < 0.85), apply_theta=bool(rng.random() < 0.7), cos_irra_on=bool(rng.random() < 0.6),
                 gsam_only=gsam_only)
    if cfg.sun_v == "analystic":
        flags["apply_brdf"] = True
    model, _ = mg.build_ref_model(mg.ref, cfg, seed=30 + seed)
    rays = mg.sat_rays(R, 40 + seed)
    g = torch.Generator().manual_seed(seed)
    dk = {}
    if mode == "train" and rng.random() < 0.6:
        dk = dict(valid_depth=(torch.rand(R, generator=g) < 0.6).float(),
                  target_depths=torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1),
                  target_std=0.02 + 0.05 * torch.rand(R, generator=g))
    models = {"coarse": model}
    ts = emb = None
    if cfg.beta:
        emb = torch.nn.Embedding(5, cfg.t_dim)
        ts = torch.randint(0, 5, (R,), generator=g)
        models["t"] = emb
    with mg.RecordRandoms(torch.Generator().manual_seed(2)) as rec, torch.no_grad():
        res, bt = mg.quiet(mg.ref["rendering"].render_rays, models, mg.ref_args(cfg), rays, ts, mode=mode, **flags, **dk)
    p = tparams(cfg, 30 + seed)
    with torch.no_grad():
        got, bt2 = ORD.render_rays(p, cfg, rays, ORD.Randoms(replay=rec.log), mode=mode, rays_t=emb(ts) if cfg.beta else None, **flags, **dk)
    tag = f"seed {seed}: {cfg} {mode} {flags} prior={bool(dk)}"
    assert bt == bt2, tag
    assert set(res) == {k for k in got if not k.startswith("_")}, (tag, sorted(set(res) ^ {k for k in got if not k.startswith('_')}))
    for k, v in res.items():
        a = got[k]
        if k == "sort_idx_coarse":
            assert torch.equal(a, v), f"{tag}: {k}"
            continue
        if not torch.is_tensor(v):
            v = torch.as_tensor(v)
        fin = torch.isfinite(v)
        assert torch.equal(fin, torch.isfinite(a)), f"{tag}: {k} finiteness"
        if k == "hpk_scl_coarse":
            a, v = 1.0 / a, 1.0 / v
        e = float((a[fin] - v[fin]).abs().max()) if bool(fin.any()) else 0.0
        assert e <= 1e-4 * float(v[fin].abs().max() if bool(fin.any()) else 0.0) + 2e-5, f"{tag}: {k} err {e:.2e}"
