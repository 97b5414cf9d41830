"""Differential fuzz of the field kernels against the CPU oracle over RANDOM network shapes and flag sets (fixed seeds).

Every case draws a geometry load_model() can build - width, depth (the skip layer sits at 4: deeper networks have it, shallower
ones do not), with / without positional encoding, Siren / ReLU,
the BRDF family and its heads, normal mode, --input_viewdir, --beta, dim_RPV - and a ragged point count, then compares the HIP
fp32 path (forward, every parameter gradient, the gradient w.r.t. the --beta embedding input) with the oracle's autograd
(oracle/field.py, pinned by the reference's goldens).  The fixed-configuration tests in test_gpu_parity.py cover the reference's
own shapes; this covers the combinations between them."""
import numpy as np
import pytest
import torch

from conftest import tparams
from oracle.config import FieldConfig
from oracle import field as OF

pytestmark = pytest.mark.gpu
DEV = "cuda"


def draw_config(rng):
    feat = int(rng.choice([64, 128, 192, 256]))
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
    elif family == "hapke":
        kw.update(b=1, c=int(rng.random() < 0.7), theta=int(rng.random() < 0.5))
    elif family == "microfacet":
        kw.update(roughness=True)
    if family != "lambert" and kw["normal"] == "none":
        kw["normal"] = "learned"                      # the BRDFs need a normal
    return FieldConfig(**kw)


@pytest.mark.parametrize("seed", list(range(64)))
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
    (ref * coef).sum().backward()
    t_gpu = t_ref.detach().to(DEV).requires_grad_(True) if cfg.beta else None
    out = model(xyz.to(DEV), input_dir=None if dirs is None else dirs.to(DEV), input_t=t_gpu, **flags)
    tag = (f"fuzz {seed}: F={cfg.feat} L={cfg.layers} siren={int(cfg.siren)} pe={int(cfg.mapping)} normal={cfg.normal} "
           f"viewdir={cfg.input_viewdir} beta={int(cfg.beta)} heads={cfg.brdf_head_names(flags['apply_brdf'], flags['apply_theta'])} B={B}")
    assert out.shape == ref.shape, tag
    err = float((out.detach().cpu() - ref.detach()).abs().max())
    diag(f"{tag}: out max|err| {err:.2e}")
    assert err <= 2e-4 * float(ref.detach().abs().max()) + 2e-5, tag
    (out * coef.to(DEV)).sum().backward()
    tol = 1e-3 if nr_an else 2e-4
    if cfg.beta:
        scale = float(t_ref.grad.abs().max())
        assert float((t_gpu.grad.cpu() - t_ref.grad).abs().max()) <= tol * scale + 1e-7, tag + " d_t_embed"
    for k, v in model.named_parameters():
        want = p[k].grad
        got = v.grad
        if want is None:
            assert got is None or float(got.abs().max()) == 0.0, f"{tag} {k}: unused parameter got a gradient"
            continue
        scale = float(want.abs().max())
        e = float((got.cpu() - want).abs().max()) if got is not None else scale
        assert e <= tol * scale + 1e-7, f"{tag} {k}: err {e:.3e} scale {scale:.3e}"
