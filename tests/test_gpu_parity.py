"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed golden vectors.
Run on the MI355X box with `pytest -m gpu`.  Tolerances: fp32 mode 1e-4 relative (north_star); bf16 mode is
checked against looser, stated bounds (its acceptance criterion is PSNR)."""
import argparse

import numpy as np
import os
import pytest
import torch

from conftest import load_golden, tparams, replay_list, assert_close
from oracle.config import FieldConfig
from oracle import field as OF, render as ORD

pytestmark = pytest.mark.gpu

CONFIGS = {
    "lambert": dict(),
    "rpv111_nlr": dict(funcM=1, funcF=1, funcH=1, normal="learned"),
    "hapke_bct": dict(b=1, c=1, theta=1, normal="learned"),
    "microfacet": dict(roughness=True, normal="learned"),
}
DEV = "cuda:0"


def mini(**kw):
    base = dict(feat=64, n_samples=16, guided_samples=16)
    base.update(kw)
    return FieldConfig(**base)


def make_args(cfg, compute_dtype="fp32"):
    return argparse.Namespace(
        model="spsbrdf-nerf", fc_layers=cfg.layers, fc_feat=cfg.feat, mapping=cfg.mapping, siren=int(cfg.siren),
        t_embbeding_tau=cfg.t_dim, beta=bool(cfg.beta), roughness=cfg.roughness, normal=cfg.normal, indirect_light=False, glossy_scale=1.0,
        sun_v=getattr(cfg, "sun_v", "none"), MultiBRDF=int(cfg.MultiBRDF), dim_RPV=cfg.dim_RPV, input_viewdir=int(cfg.input_viewdir), funcM=cfg.funcM,
        funcF=cfg.funcF, funcH=cfg.funcH, b=cfg.b, c=cfg.c, theta=cfg.theta, shell_hapke=cfg.shell_hapke, hpk_scl=cfg.hpk_scl,
        guided_samples=cfg.guided_samples, n_samples=cfg.n_samples, n_importance=0, std_range=cfg.std_range, data=cfg.data,
        sc_lambda=0.0, chunk=5120, noise_std=cfg.noise_std, margin=0.0001, stdscale=1, fresnel_f0=cfg.fresnel_f0,
        compute_dtype=compute_dtype)


def build_model(cfg, seed, compute_dtype="fp32"):
    from brdf_nerf_amd import load_model
    model = load_model(make_args(cfg, compute_dtype))
    sd = {k: torch.from_numpy(v) for k, v in cfg.make_params(seed).items()}
    assert set(sd) == set(model.state_dict()), sorted(set(sd) ^ set(model.state_dict()))
    model.load_state_dict(sd)
    return model.to(DEV)


class Replay:
    """Feed recorded random draws (in the reference's order) to torch.rand / rand_like / randn on the device."""

    def __init__(self, draws):
        self.draws = list(draws)

    def __enter__(self):
        self._o = (torch.rand, torch.rand_like, torch.randn)

        def nxt(shape):
            t = self.draws.pop(0)
            assert tuple(t.shape) == tuple(shape), (tuple(t.shape), tuple(shape))
            return t.to(DEV)

        def rand(*size, **kw):
            size = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
            return nxt(size)

        torch.rand = rand
        torch.rand_like = lambda x, **kw: nxt(x.shape)
        torch.randn = rand
        return self

    def __exit__(self, *a):
        torch.rand, torch.rand_like, torch.randn = self._o


def diag(line):
    import os
    path = os.environ.get("BN_DIAG")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


# Per-sample field outputs at the END of the two-pass pipeline inherit the positional encoding's conditioning:
# the top octave multiplies a 1-ulp difference in a sample depth (torch.linspace / torch.sum low bits, see
# test_stratified_z) by 2^9, so they are held to 1e-3; with identical inputs the field itself is held to 1e-4
# (test_field_forward_*).  Ray-level results (rgb, depth, weights, ...) are held to the north_star's 1e-4.
PER_SAMPLE = ("sigmas", "albedo", "normal_lr", "beta", "rpv_k", "rpv_theta", "rpv_rhoc", "hpk_b", "hpk_c", "hpk_theta", "roughness")
# the north_star's 1e-4 quantities: rendered pixel values, depths and the compositing weights they are built from
RAY_HEADLINE = ("rgb", "depth", "albedo_accu", "weights", "alphas", "transparency", "z_vals", "z_vals_unsort", "irradiance",
                "rays_d", "sun_d")


def compare_render(res, g, tag, ray_tol=(1e-4, 2e-5), other_tol=(1e-3, 2e-4)):
    bad = []
    for k in sorted(k[4:] for k in g if k.startswith("out/")):
        ref = g["out/" + k]
        if k == "sort_idx_coarse":
            mism = int((res[k].cpu().numpy() != ref).sum())
            diag(f"{tag} {k}: {mism} index mismatches")
            if mism:
                bad.append(f"{k}: {mism} index mismatches")
            continue
        got = res[k].detach().cpu().double().numpy()
        refd = ref.astype(np.float64)
        if k == "hpk_scl_coarse":
            # 1 / (hpk_scl (n.v + n.l)) (spsbrdfnerf.py:248): a diagnostic that is unbounded where the two cosines cancel, so it
            # is compared through its reciprocal - the sum of two ray-level dot products, each held to ~1e-4 elsewhere
            got, refd = 1.0 / got, 1.0 / refd
        both = np.isnan(got) & np.isnan(refd)
        err = np.where(both, 0.0, np.abs(got - refd))
        base = k[:-7]
        if base in PER_SAMPLE:
            rtol, atol = 5e-3, 5e-3
        elif base in RAY_HEADLINE:
            rtol, atol = ray_tol
        elif base == "hpk_scl":
            rtol, atol = 5e-3, 1e-3
        else:
            rtol, atol = other_tol
        viol = float(np.nanmax(err - (atol + rtol * np.abs(np.where(both, 0.0, refd)))))
        diag(f"{tag} {k}: max|err| {np.nanmax(err):.3e} scale {np.nanmax(np.abs(refd)):.3e} viol {viol:.3e}")
        if not viol <= 0:
            bad.append(f"{k}: max|err| {np.nanmax(err):.3e}")
    assert not bad, f"{tag}: " + "; ".join(bad)


# ------------------------------------------------------------------------------------------------ C ABI smoke
def test_library_loads_on_gpu():
    from brdf_nerf_amd import _lib
    assert _lib.lib().bn_abi_version() == _lib.BN_ABI_VERSION == 7


def test_device_fault_word_stays_clear():
    """The barrier-free bf16 forward trunk hands column halves over through LDS counters with bounded waits; a wait that
    gave up would set the library's device fault word (and fail the next call with BN_ELAUNCH).  After real launches at
    F = 512 in both 16-bit modes it must read 0."""
    import ctypes as C
    from brdf_nerf_amd import _lib
    cfg = FieldConfig(funcM=1, funcF=1, funcH=1, normal="learned")
    xyz = (torch.rand(40000, 3, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    for dtype in ("bf16", "fp16"):
        model = build_model(cfg, 2, dtype)
        with torch.no_grad():
            model(xyz, apply_brdf=True, nr_lr_on=True)
        model(xyz, apply_brdf=True, nr_lr_on=True).sum().backward()
    faults = C.c_uint(123)
    _lib.check(_lib.lib().bn_device_faults(C.byref(faults), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "bn_device_faults")
    assert faults.value == 0


# ------------------------------------------------------------------------------------------------ per-ray kernels
def test_stratified_z():
    from brdf_nerf_amd import functions as Fn
    g = torch.Generator().manual_seed(0)
    R, S = 300, 64
    near, far = torch.rand(R, 1, generator=g) * 0.5, 1.5 + torch.rand(R, 1, generator=g)
    u = torch.rand(R, S, generator=g)
    ref = ORD.get_z_vals(S, near, far, u)
    got = Fn.stratified_z(near.to(DEV), far.to(DEV), u.to(DEV)).cpu()
    # torch.linspace on the CPU is vectorised (base + lane*step per SIMD vector): its low bits depend on the host's
    # vector width, so the kernel (elementwise ATen formula, as on CUDA) is held to 1 ulp of the largest depth.
    assert float((got - ref).abs().max()) <= 2.4e-7


@pytest.mark.parametrize("S", [16, 128])
def test_composite_golden(S):
    from brdf_nerf_amd import functions as Fn
    g = load_golden(f"composite_S{S}")
    z = torch.from_numpy(g["z"]).to(DEV)
    sigma = torch.from_numpy(g["sigma"]).to(DEV).requires_grad_(True)
    a, T, w, d = Fn.composite(z, sigma)
    for got, key in ((a, "alphas"), (T, "transparency"), (w, "weights"), (d, "depth")):
        assert_close(got, g[key], 1e-5, 1e-7, key)
    ((w * torch.from_numpy(g["cw"]).to(DEV)).sum() + (d * torch.from_numpy(g["cd"]).to(DEV)).sum()).backward()
    assert_close(sigma.grad, g["dsigma"], 1e-4, 1e-6, "dsigma")


def test_composite_channels_and_sizes():
    """Weighted channel sums + backward against the oracle, S from 1..192 incl. ragged (S not a multiple of 64)."""
    from brdf_nerf_amd import functions as Fn
    # dense rows take the flat 4-channel-group path (render_kernels.hip): float4 groups when C % 4 == 0; C in (16, 32] (beta /
    # both normals + three RPV heads) walks 8 groups per sample
    for S, C, R in ((1, 4, 7), (63, 7, 33), (64, 4, 128), (130, 13, 65), (192, 16, 9), (128, 16, 65), (128, 8, 9), (17, 8, 5),
                    (512, 16, 3), (3, 16, 2), (128, 12, 4), (128, 17, 9), (80, 20, 33), (128, 32, 5), (65, 19, 3)):
        g = torch.Generator().manual_seed(S)
        z = torch.sort(torch.rand(R, S, generator=g) * 2, -1)[0]
        out = torch.randn(R, S, C, generator=g)
        out[..., 3] *= 5
        noise = torch.randn(R, S, generator=g)
        cw, cd, ca = torch.rand(R, S, generator=g), torch.rand(R, generator=g), torch.rand(R, C, generator=g)
        ca[:, 3] = 0
        o_ref = out.clone().requires_grad_(True)
        a, T, w, d = ORD.composite(z, o_ref[..., 3], noise, 0.3)
        acc = (w.unsqueeze(-1) * o_ref).sum(-2)
        ((w * cw).sum() + (d * cd).sum() + (acc * ca).sum()).backward()
        o_gpu = out.clone().to(DEV).requires_grad_(True)
        a2, T2, w2, d2, acc2 = Fn.composite(z.to(DEV), o_gpu, noise.to(DEV), 0.3)
        ((w2 * cw.to(DEV)).sum() + (d2 * cd.to(DEV)).sum() + (acc2 * ca.to(DEV)).sum()).backward()
        assert_close(w2, w, 1e-5, 1e-7, f"w S={S}")
        assert_close(d2, d, 1e-5, 1e-6, f"depth S={S}")
        assert_close(acc2, acc, 1e-4, 1e-5, f"acc S={S}")
        scale = float(o_ref.grad.abs().max())
        assert float((o_gpu.grad.cpu() - o_ref.grad).abs().max()) <= 1e-4 * scale + 1e-7, f"d_out S={S}"


@pytest.mark.parametrize("mode", ["test", "train"])
def test_guided_samples_golden(mode):
    from brdf_nerf_amd import functions as Fn
    g = load_golden(f"guided_{mode}")
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    kw = {}
    u = t["rand0"]
    if mode == "train":
        valid = t["valid_depth"] > 0
        kw = dict(use_target=valid.float().to(DEV), target_depth=t["target_depths"][:, 0].contiguous().to(DEV),
                  target_std=t["target_std"].to(DEV), u_target=t["rand1"].to(DEV),
                  target_row=(torch.cumsum(valid.int(), 0) - 1).clamp_min(0).int().to(DEV))
    z2, z_all, idx = Fn.guided_samples(t["z"].to(DEV), t["weights"].to(DEV), t["depth"].to(DEV), u.to(DEV), 0.0, 2.0, 3.0, **kw)
    assert_close(z2, g["z2_sorted"], 2e-6, 2e-6, "z2_sorted")
    assert_close(z_all, g["z_all"], 2e-6, 2e-6, "z_all")
    # indices must be bit-exact wherever the sorted value is unique; inside a run of exactly tied depths
    # (a ray whose guided samples collapse onto one depth) any order is a valid torch.sort result, so there
    # the index set of the run must match instead.
    ref_idx, ref_z = t["sort_idx"], t["z_all"]
    got_idx = idx.cpu()
    tie = torch.zeros_like(ref_z, dtype=torch.bool)
    eq = ref_z[:, 1:] == ref_z[:, :-1]
    tie[:, 1:] |= eq
    tie[:, :-1] |= eq
    mism = (got_idx != ref_idx) & ~tie
    assert int(mism.sum()) == 0, f"{int(mism.sum())} sort indices differ outside tie runs"
    assert torch.equal(torch.sort(got_idx, -1)[0], torch.sort(ref_idx, -1)[0])
    assert torch.equal(torch.gather(torch.cat([t["z"], z2.cpu()], -1), 1, got_idx), z_all.cpu())
    assert torch.all(z_all[:, 1:] >= z_all[:, :-1])


def test_guided_samples_properties_full_size():
    """Size-independent properties at the BASELINE shape (4096 rays, S=G=64): sortedness, permutation, window."""
    from brdf_nerf_amd import functions as Fn
    g = torch.Generator().manual_seed(1)
    R, S, G = 4096, 64, 64
    near, far = torch.zeros(R, 1), torch.full((R, 1), 2.0)
    z = ORD.get_z_vals(S, near, far, torch.rand(R, S, generator=g)).to(DEV)
    sig = (torch.relu(torch.randn(R, S, generator=g)) * 20 * (torch.rand(R, S, generator=g) < 0.2)).to(DEV)
    a, T, w, d = Fn.composite(z, sig)
    z2, z_all, idx = Fn.guided_samples(z, w, d, torch.rand(R, G, generator=g).to(DEV), 0.0, 2.0, 3.0)
    assert torch.all(z_all[:, 1:] >= z_all[:, :-1]) and torch.all(z2[:, 1:] >= z2[:, :-1])
    assert torch.equal(torch.sort(idx, -1)[0], torch.arange(S + G, device=DEV).expand(R, -1))
    assert torch.equal(torch.gather(torch.cat([z, z2], -1), 1, idx), z_all)
    assert float(z2.min()) >= 0.0 and float(z2.max()) <= 2.0 + 1e-6


# ------------------------------------------------------------------------------------------------ BRDFs
def test_brdf_rpv_golden():
    from brdf_nerf_amd import functions as Fn
    g = load_golden("brdf_rpv")
    t = {k: torch.from_numpy(v).to(DEV) for k, v in g.items()}
    n, w, k, th, rc = [t[x].clone().requires_grad_(True) for x in ("n", "w", "k", "theta", "rhoc")]
    brdf, aux = Fn.RPVFunction.apply(t["l"], t["v"], n, w, k, th, rc)
    assert_close(brdf, g["brdf"], 1e-4, 1e-6, "brdf")
    assert_close(aux[:, 0:3], g["M1"], 1e-4, 1e-6, "M1")
    assert_close(aux[:, 3:4], g["G"], 1e-4, 1e-5, "G")
    assert_close(aux[:, 4:7], g["H"], 1e-4, 1e-6, "H")
    (brdf * t["coef"]).sum().backward()
    for got, key in ((n, "dn"), (w, "dw"), (k, "dk"), (th, "dtheta"), (rc, "drhoc")):
        assert_close(got.grad, g[key], 2e-3, 1e-4, key)


@pytest.mark.parametrize("tag,use_c,use_t,shell", [("hapke_b", 0, 0, 0), ("hapke_bc", 1, 0, 0), ("hapke_bct", 1, 1, 0),
                                                   ("hapke_shell1", 0, 0, 1), ("hapke_shell2", 0, 0, 2),
                                                   ("hapke_shell3", 0, 0, 3)])
def test_brdf_hapke_golden(tag, use_c, use_t, shell):
    from brdf_nerf_amd import functions as Fn
    g = load_golden(f"brdf_{tag}")
    t = {k: torch.from_numpy(v).to(DEV) for k, v in g.items()}
    n, w, b, c, th = [t[x].clone().requires_grad_(True) for x in ("n", "w", "b", "c", "theta")]
    brdf, aux = Fn.HapkeFunction.apply(t["l"], t["v"], n, w, None if shell else b, c if use_c else None,
                                       th if use_t else None, 4.0, shell)
    assert_close(brdf, g["brdf"], 2e-4, 2e-6, "brdf")
    assert_close(aux[:, 0:3], g["P"], 2e-4, 2e-6, "P")
    assert_close(aux[:, 3:6], g["Hi"], 2e-4, 2e-6, "Hi")
    assert_close(aux[:, 6:9], g["Hv"], 2e-4, 2e-6, "Hv")
    assert_close(aux[:, 9:10], g["S"], 2e-4, 2e-6, "S")
    (brdf * t["coef"]).sum().backward()
    assert_close(w.grad, g["dw"], 2e-3, 1e-5, "dw")
    if "dn" in g:
        assert_close(n.grad, g["dn"], 5e-3, 1e-3, "dn", ignore_ref_nan=True)
    if not shell:
        assert_close(b.grad, g["db"], 2e-3, 1e-5, "db")
    if use_c:
        assert_close(c.grad, g["dc"], 2e-3, 1e-5, "dc")
    if use_t:
        assert_close(th.grad, g["dtheta"], 5e-3, 1e-3, "dtheta", ignore_ref_nan=True)


def test_brdf_microfacet_golden():
    from brdf_nerf_amd import functions as Fn
    g = load_golden("brdf_microfacet")
    t = {k: torch.from_numpy(v).to(DEV) for k, v in g.items()}
    n, w, r = [t[x].clone().requires_grad_(True) for x in ("n", "w", "rough")]
    brdf, aux = Fn.MicrofacetFunction.apply(t["l"], t["v"], n, w, r, 0.04)
    assert_close(brdf, g["brdf"], 1e-4, 1e-6, "brdf")
    for col, key in ((0, "glossy"), (1, "f"), (2, "g"), (3, "d"), (4, "l_dot_n"), (9, "n_h")):
        assert_close(aux[:, col:col + 1], g[key].reshape(-1, 1), 1e-4, 1e-6, key)
    (brdf * t["coef"]).sum().backward()
    assert_close(w.grad, g["dw"], 1e-4, 1e-6, "dw")
    assert_close(n.grad, g["dn"], 2e-3, 1e-4, "dn")
    assert_close(r.grad, g["drough"], 2e-3, 1e-4, "drough")


# ------------------------------------------------------------------------------------------------ field MLP
@pytest.mark.parametrize("name", list(CONFIGS))
def test_field_forward_golden_fp32(name):
    g = load_golden(f"field_{name}_F64")
    cfg = mini(**CONFIGS[name])
    model = build_model(cfg, 11)
    xyz = torch.from_numpy(g["xyz"]).to(DEV)
    lr = cfg.normal == "learned"
    with torch.no_grad():
        out = model(xyz, apply_brdf=True, apply_theta=True, nr_lr_on=lr)
        out0 = model(xyz, apply_brdf=False, nr_lr_on=lr)
        sig = model(xyz, sigma_only=True)
    assert_close(out, g["out_brdf"], 1e-4, 1e-5, "out_brdf")
    assert_close(out0, g["out_nobrdf"], 1e-4, 1e-5, "out_nobrdf")
    assert_close(sig, g["sigma"], 1e-4, 1e-5, "sigma")


@pytest.mark.parametrize("feat,B", [(512, 1000), (256, 300), (128, 129)])
def test_field_forward_oracle_fp32(feat, B):
    cfg = FieldConfig(feat=feat, funcM=1, funcF=1, funcH=1, normal="learned")
    model = build_model(cfg, 3)
    p = tparams(cfg, 3)
    xyz = torch.rand(B, 3, generator=torch.Generator().manual_seed(B)) * 2 - 1
    ref = OF.field_forward(p, cfg, xyz, apply_brdf=True, nr_lr_on=True)
    with torch.no_grad():
        got = model(xyz.to(DEV), apply_brdf=True, nr_lr_on=True)
    assert_close(got, ref, 1e-4, 2e-5, f"F={feat}")


# stated bounds of the 16-bit throughput modes against the fp32 oracle at F=512 (their acceptance criterion is the held-out
# PSNR gate further down): bf16 keeps 8 significant bits, fp16 11 - its bounds are 6x tighter.
# ncos = worst analytic-normal cosine at F = 512.  Round 3 stashes d act / d z in 8-bit fixed point in both 16-bit modes
# (csrc/field_kernels.h DPiece; absolute error 1/254 of the largest derivative): measured bf16 0.9942 (0.988 with round 2's bf16
# image, whose error is relative), fp16 0.99915 (0.99977 with its fp16 image); gradient cosines 0.99983 / 0.99996.
HALF_BOUNDS = {"bf16": dict(rgb=3e-2, sig=0.15, cos=0.98, ncos=0.98), "fp16": dict(rgb=5e-3, sig=0.025, cos=0.999, ncos=0.999)}


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_field_forward_half_close(dtype):
    """16-bit throughput modes: per-point outputs within the stated absolute bound of the fp32 oracle at F=512."""
    cfg = FieldConfig(funcM=1, funcF=1, funcH=1, normal="learned")
    model = build_model(cfg, 3, dtype)
    p = tparams(cfg, 3)
    xyz = torch.rand(2000, 3, generator=torch.Generator().manual_seed(5)) * 2 - 1
    ref = OF.field_forward(p, cfg, xyz, apply_brdf=True, nr_lr_on=True)
    with torch.no_grad():
        got = model(xyz.to(DEV), apply_brdf=True, nr_lr_on=True).cpu()
    err = (got - ref).abs()
    sig_rel = (err[:, 3] / (ref[:, 3].abs() + 1e-2)).max()
    diag(f"{dtype} field forward F512: max abs err rgb {float(err[:, :3].max()):.3e}, sigma rel {float(sig_rel):.3e}, "
         f"heads {float(err[:, 7:].max()):.3e}")
    b = HALF_BOUNDS[dtype]
    assert float(err[:, :3].max()) < b["rgb"] and float(sig_rel) < b["sig"]


def _field_grads(cfg, seed, compute_dtype, B, heads_flags):
    model = build_model(cfg, seed, compute_dtype)
    p = tparams(cfg, seed)
    for v in p.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(B)
    xyz = torch.rand(B, 3, generator=g) * 2 - 1
    ref = OF.field_forward(p, cfg, xyz, **heads_flags)
    coef = torch.randn(ref.shape, generator=g)
    (ref * coef).sum().backward()
    out = model(xyz.to(DEV), **heads_flags)
    (out * coef.to(DEV)).sum().backward()
    return model, p, out, ref


@pytest.mark.parametrize("name", list(CONFIGS))
def test_field_backward_oracle_fp32(name):
    cfg = mini(**CONFIGS[name])
    flags = dict(apply_brdf=True, apply_theta=True, nr_lr_on=cfg.normal == "learned")
    model, p, out, ref = _field_grads(cfg, 11, "fp32", 257, flags)
    assert_close(out, ref, 1e-4, 1e-5, "out")
    for k, v in model.named_parameters():
        want = p[k].grad
        if want is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
            continue
        scale = float(want.abs().max())
        err = float((v.grad.cpu() - want).abs().max())
        assert err <= 2e-4 * scale + 1e-7, f"{k}: err {err:.3e} scale {scale:.3e}"


def test_field_backward_oracle_fp32_F512():
    cfg = FieldConfig(funcM=1, funcF=1, funcH=1, normal="learned")
    flags = dict(apply_brdf=True, nr_lr_on=True)
    model, p, out, ref = _field_grads(cfg, 4, "fp32", 333, flags)
    for k, v in model.named_parameters():
        want = p[k].grad
        scale = float(want.abs().max())
        err = float((v.grad.cpu() - want).abs().max())
        assert err <= 5e-4 * scale + 1e-7, f"{k}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("feat", [256, 128])
def test_field_backward_oracle_fp32_other_widths(feat):
    """F = 256 takes the riding stash copies with one column tile per wave, F = 128 the stand-alone copies with half the
    waves idle in the trunk: both against the oracle's autograd."""
    cfg = FieldConfig(feat=feat, funcM=1, funcF=1, funcH=1, normal="learned")
    flags = dict(apply_brdf=True, nr_lr_on=True)
    model, p, out, ref = _field_grads(cfg, 6, "fp32", 300, flags)
    assert_close(out, ref, 1e-4, 2e-5, "out")
    for k, v in model.named_parameters():
        want = p[k].grad
        scale = float(want.abs().max())
        err = float((v.grad.cpu() - want).abs().max())
        assert err <= 5e-4 * scale + 1e-7, f"{k}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("tag,kw", [("relu", dict(siren=False)), ("nomap", dict(mapping=False)), ("rpv333", dict(dim_RPV=3))])
def test_field_relu_and_no_mapping_golden_fp32(tag, kw):
    """--siren 0 (ReLU epilogues) and no --mapping (raw xyz padded into the 64-wide first operand) against reference goldens:
    forward values and every parameter gradient."""
    g = load_golden(f"field_{tag}_F64")
    cfg = mini(funcM=1, funcF=1, funcH=1, normal="learned", **kw)
    model = build_model(cfg, 13)
    out = model(torch.from_numpy(g["xyz"]).to(DEV), apply_brdf=True, apply_theta=True, nr_lr_on=True)
    assert_close(out, g["out_brdf"], 1e-4, 1e-5, "out")
    (out * torch.from_numpy(g["coef"]).to(DEV)).sum().backward()
    for k, v in model.named_parameters():
        ref = g[f"grad/{k}"]
        scale = max(float(np.abs(ref).max()), 1e-12)
        got = v.grad.cpu() if v.grad is not None else torch.zeros_like(v).cpu()
        assert float((got - torch.from_numpy(ref)).abs().max()) <= 3e-4 * scale + 1e-8, k


def test_field_both_normals_fp32():
    """--normal analystic_learned: analytic (channels 4-6) and learned (7-9) normals together, forward and the gradients
    through both (the analytic one needs the double backward) against the oracle's autograd."""
    cfg = mini(b=1, c=1, normal="analystic_learned")
    flags = dict(apply_brdf=True, apply_theta=True, nr_lr_on=True, nr_an_on=True)
    model, p, out, ref = _field_grads(cfg, 21, "fp32", 150, flags)
    assert out.shape[1] == 4 + 3 + 3 + 6
    assert_close(out, ref, 2e-4, 2e-5, "out")
    for k, v in model.named_parameters():
        want = p[k].grad
        if want is None:
            continue
        scale = float(want.abs().max())
        err = float((v.grad.cpu() - want).abs().max())
        assert err <= 1e-3 * scale + 1e-7, f"{k}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("layers", [4, 6])
def test_field_other_depths_fp32(layers):
    """--fc_layers 4 (the skip at layer 4 is never reached) and 6 (skip in the middle): forward and gradients vs the oracle."""
    cfg = FieldConfig(feat=128, layers=layers, funcM=1, funcF=1, funcH=1, normal="learned")
    flags = dict(apply_brdf=True, nr_lr_on=True)
    model, p, out, ref = _field_grads(cfg, 8, "fp32", 200, flags)
    assert_close(out, ref, 1e-4, 2e-5, "out")
    for k, v in model.named_parameters():
        want = p[k].grad
        scale = float(want.abs().max())
        err = float((v.grad.cpu() - want).abs().max())
        assert err <= 5e-4 * scale + 1e-7, f"{k}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("feat", [512, 256])
def test_field_backward_half_direction(feat, dtype):
    """16-bit gradients: cosine similarity with the fp32 oracle gradient per weight matrix above the stated bound."""
    cfg = FieldConfig(feat=feat)
    model, p, out, ref = _field_grads(cfg, 4, dtype, 1024, {})
    worst = 1.0
    for k, v in model.named_parameters():
        want = p[k].grad.flatten()
        got = v.grad.cpu().flatten()
        assert bool(torch.isfinite(got).all()), k
        cos = float((want * got).sum() / (want.norm() * got.norm() + 1e-30))
        worst = min(worst, cos)
        assert cos > HALF_BOUNDS[dtype]["cos"], f"{k}: cosine {cos}"
    diag(f"{dtype} field backward F{feat}: worst gradient cosine vs fp32 oracle {worst:.6f}")


@pytest.mark.parametrize("name", ["lambert", "rpv111_nan"])
def test_fp16_loss_scaling_is_scale_free(name):
    """fp16 has 5 exponent bits: the backward chains run on gradients scaled by a power of two chosen on the device from
    max |d_out| (grad_amax_kernel).  Upstream gradients 1e-7 times smaller (deep in fp16's subnormals without the
    scaling) or 1e+4 times larger (beyond its maximum) must give the same parameter gradients up to that factor -
    through the primal chain and through the analytic-normal double backward."""
    cfg = FieldConfig(**(dict(funcM=1, funcF=1, funcH=1, normal="analystic") if name == "rpv111_nan" else {}))
    flags = dict(apply_brdf=True, nr_an_on=True) if name == "rpv111_nan" else {}
    model = build_model(cfg, 6, "fp16")
    g = torch.Generator().manual_seed(3)
    xyz = (torch.rand(1500, 3, generator=g) * 2 - 1).to(DEV)
    coef = None
    grads = {}
    for scale in (1.0, 1e-7, 1e4):
        model.zero_grad()
        out = model(xyz, **flags)
        if coef is None:
            coef = torch.randn(out.shape, generator=g).to(DEV)
        (out * coef * scale).sum().backward()
        grads[scale] = {k: v.grad.clone() / scale for k, v in model.named_parameters() if v.grad is not None}
    for scale in (1e-7, 1e4):
        for k, g0 in grads[1.0].items():
            g1 = grads[scale][k]
            assert bool(torch.isfinite(g1).all()), (k, scale)
            ref_mag = float(g0.abs().max())
            err = float((g1 - g0).abs().max())
            # (1e-7 and 1e4 are not powers of two: the two runs round every fp16 operand independently; measured <= 5.1e-3)
            assert err <= 6e-3 * ref_mag + 1e-12, f"{k} at upstream scale {scale:g}: err {err:.3e} of {ref_mag:.3e}"


@pytest.mark.parametrize("name", ["lambert", "rpv111_nlr"])
def test_feats_folding_is_the_same_function(name):
    """fold_feats (the linear feats layer multiplied into the heads' first layers, gradients unfolded by the chain rule)
    against the layer-by-layer evaluation: outputs and every parameter gradient, fp32, F=512."""
    import os
    from brdf_nerf_amd import load_model
    cfg = FieldConfig(**CONFIGS[name])
    res = {}
    for fold in ("1", "0"):
        os.environ["BRDFNERF_FOLD_FEATS"] = fold
        try:
            model = build_model(cfg, 7)
            assert model.spec(True, True, cfg.normal == "learned").fold_feats == (fold == "1")
            xyz = (torch.rand(700, 3, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(DEV)
            coef = torch.randn(700, model.spec(True, True, cfg.normal == "learned").out_channels,
                               generator=torch.Generator().manual_seed(4)).to(DEV)
            out = model(xyz, apply_brdf=True, apply_theta=True, nr_lr_on=cfg.normal == "learned")
            (out * coef).sum().backward()
            res[fold] = (out.detach(), {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None})
        finally:
            os.environ.pop("BRDFNERF_FOLD_FEATS", None)
    assert_close(res["1"][0], res["0"][0], 1e-4, 1e-5, "out")
    assert set(res["1"][1]) == set(res["0"][1])
    for k, g0 in res["0"][1].items():
        scale = float(g0.abs().max())
        assert float((res["1"][1][k] - g0).abs().max()) <= 2e-4 * scale + 1e-8, k


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_half_forward_variants_agree_and_repeat(dtype):
    """The 16-bit trunk runs without workgroup barriers (two wave groups hand columns over through LDS counters): the
    repeated launches must give the same bits, and the inference and training (stash-keeping) variants - two
    instantiations whose fp32 head sums the compiler may contract differently - the same values to fp32 rounding."""
    cfg = FieldConfig(funcM=1, funcF=1, funcH=1, normal="learned")
    model = build_model(cfg, 9, dtype)
    xyz = (torch.rand(5000, 3, generator=torch.Generator().manual_seed(2)) * 2 - 1).to(DEV)
    with torch.no_grad():
        a = model(xyz, apply_brdf=True, nr_lr_on=True)
        b = model(xyz, apply_brdf=True, nr_lr_on=True)
    c = model(xyz, apply_brdf=True, nr_lr_on=True)          # grad mode: stash-keeping kernel
    assert torch.equal(a, b)
    c2 = model(xyz, apply_brdf=True, nr_lr_on=True)
    assert torch.equal(c.detach(), c2.detach())
    diff = float((a - c.detach()).abs().max())
    print("inference vs training variant: max |diff|", diff)
    assert diff <= 2e-6
    (c.sum()).backward()
    g1 = {k: v.grad.clone() for k, v in model.named_parameters()}
    model.zero_grad()
    c2 = model(xyz, apply_brdf=True, nr_lr_on=True)
    (c2.sum()).backward()
    for k, v in model.named_parameters():                   # fp32 atomics: order-dependent rounding only
        assert_close(v.grad, g1[k], 1e-3, 1e-3 * float(g1[k].abs().max()), f"repeat grad {k}")


# ------------------------------------------------------------------------------------------------ full render
@pytest.mark.parametrize("mode", ["train", "test"])
@pytest.mark.parametrize("name", list(CONFIGS))
def test_render_rays_golden_fp32(name, mode):
    from brdf_nerf_amd import render_rays
    g = load_golden(f"render_{name}_{mode}")
    cfg = mini(**CONFIGS[name])
    model = build_model(cfg, 11)
    args = make_args(cfg)
    kw = {}
    if mode == "train":
        kw = dict(valid_depth=torch.from_numpy(g["tgt/valid_depth"]).to(DEV), target_depths=torch.from_numpy(g["tgt/depths"]).to(DEV),
                  target_std=torch.from_numpy(g["tgt/depth_std"]).to(DEV))
    with Replay(replay_list(g)) as rp:
        res, brdf_type = render_rays({"coarse": model}, args, torch.from_numpy(g["rays"]).to(DEV), None, mode=mode,
                                     apply_brdf=name != "lambert", apply_theta=True, cos_irra_on=name != "lambert", **kw)
        assert rp.draws == [], "consumed a different number of random draws than the reference"
    assert brdf_type == str(g["brdf_type"])
    ref_keys = {k[4:] for k in g if k.startswith("out/")}
    assert ref_keys == set(res), sorted(ref_keys ^ set(res))
    # the GGX lobe turns a 1e-4 difference in the accumulated normal into ~1e-4 ABSOLUTE on rgb; Hapke's opposition / shadowing
    # terms amplify likewise: the REFERENCE's own fp32 rgb is 4.1e-5 away from the fp64 evaluation on the hapke_bct fixture,
    # this build 6-7e-5 from the reference depending on the compiler's FMA contraction (the criterion below bounds it by 3x)
    ray_tol = (1e-4, 1e-4) if name == "microfacet" else ((1e-4, 4e-5) if name.startswith("hapke") else (1e-4, 2e-5))
    compare_render(res, g, f"render_{name}_{mode}", ray_tol=ray_tol)
    # Independent accuracy criterion: against an fp64 evaluation of the same algorithm (oracle, same random draws) the
    # HIP path's rgb/depth error must be no worse than 3x the reference's own fp32 error.
    p64 = tparams(cfg, 11, torch.float64)
    kw64 = {k: v.cpu().double() for k, v in kw.items()}
    truth, _ = ORD.render_rays(p64, cfg, torch.from_numpy(g["rays"]).double(), ORD.Randoms(replay=replay_list(g)), mode=mode,
                               apply_brdf=name != "lambert", apply_theta=True, cos_irra_on=name != "lambert", **kw64)
    for k in ("rgb_coarse", "depth_coarse"):
        e_mine = float((res[k].detach().cpu().double() - truth[k]).abs().max())
        e_ref = float((torch.from_numpy(g["out/" + k]).double() - truth[k]).abs().max())
        diag(f"render_{name}_{mode} {k}: err vs fp64 truth: hip {e_mine:.3e} reference-fp32 {e_ref:.3e}")
        assert e_mine <= max(3 * e_ref, 5e-6), f"{k}: hip err {e_mine:.3e} vs reference fp32 err {e_ref:.3e}"
    if mode == "train":
        tgt = torch.from_numpy(g["tgt/rgbs"]).to(DEV)
        loss = torch.mean((res["rgb_coarse"] - tgt) ** 2) + 0.01 * torch.mean(res["depth_coarse"])
        assert_close(loss, g["loss"], 1e-4, 1e-7, "loss")
        loss.backward()
        for k, v in model.named_parameters():
            ref = g[f"grad/{k}"]
            got = v.grad.cpu().numpy() if v.grad is not None else np.zeros_like(ref)
            scale = max(float(np.abs(ref).max()), 1e-12)
            err = float(np.abs(got - ref).max())
            diag(f"render_{name}_train grad {k}: err {err:.3e} scale {scale:.3e}")
            # end-to-end (two passes + resampling): 5e-3 of the largest entry; the field backward alone, with identical
            # inputs, is held to 2e-4 (test_field_backward_oracle_fp32)
            assert err <= 5e-3 * scale + 1e-9, f"{k}: err {err:.3e} scale {scale:.3e}"


BRANCHES = {   # round 3: branches of inference() that no fixture reached before (tests/golden/make_goldens.py gen_branches)
    "rpv_m1f1h2": (dict(funcM=1, funcF=1, funcH=2, normal="learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
    "rpv_m1f1h2_multibrdf": (dict(funcM=1, funcF=1, funcH=2, normal="learned", MultiBRDF=True),
                             dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
    "rpv_m1h2": (dict(funcM=1, funcH=2, normal="learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
    "shell1_nobrdf": (dict(shell_hapke=1, normal="learned"), dict(apply_brdf=False, apply_theta=False, cos_irra_on=False)),
    "shell2_nobrdf": (dict(shell_hapke=2, normal="learned"), dict(apply_brdf=False, apply_theta=False, cos_irra_on=True)),
    "shell3_nobrdf": (dict(shell_hapke=3, normal="learned"), dict(apply_brdf=False, apply_theta=False, cos_irra_on=True)),
    "shell3_brdf": (dict(shell_hapke=3, normal="learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
}


@pytest.mark.parametrize("name", list(BRANCHES))
def test_render_rays_unpinned_branches_golden_fp32(name):
    """funcH == 2 (rhoc := albedo; models/spsbrdfnerf.py:306,317) per ray and per sample, and shell_hapke in {1, 2, 3} with
    apply_brdf=False (:320,348,383): render_rays against the reference's outputs, key set included."""
    from brdf_nerf_amd import render_rays
    g = load_golden(f"render_{name}_test")
    kw, flags = BRANCHES[name]
    cfg = mini(**kw)
    model = build_model(cfg, 11)
    with torch.no_grad(), Replay(replay_list(g)) as rp:
        res, brdf_type = render_rays({"coarse": model}, make_args(cfg), torch.from_numpy(g["rays"]).to(DEV), None, mode="test", **flags)
        assert rp.draws == []
    assert brdf_type == str(g["brdf_type"])
    assert {k[4:] for k in g if k.startswith("out/")} == set(res)
    compare_render(res, g, f"render_{name}_test", ray_tol=(1e-4, 4e-5) if name.startswith("shell") else (1e-4, 2e-5))


def test_render_rays_ref_sphere_golden():
    """rows / cols -> ref_sphere (models/spsbrdfnerf.py:404-412), bit for bit, with the reference's tiling order."""
    from brdf_nerf_amd import render_rays
    g = load_golden("render_rpv111_nlr_refsphere_test")
    cfg = mini(**CONFIGS["rpv111_nlr"])
    model = build_model(cfg, 11)
    with torch.no_grad(), Replay(replay_list(g)):
        res, _ = render_rays({"coarse": model}, make_args(cfg), torch.from_numpy(g["rays"]).to(DEV), None, mode="test", apply_brdf=True,
                             apply_theta=True, cos_irra_on=True, rows=torch.from_numpy(g["rows"]).to(DEV), cols=torch.from_numpy(g["cols"]).to(DEV))
    assert {k[4:] for k in g if k.startswith("out/")} == set(res)
    assert np.array_equal(res["ref_sphere_coarse"].cpu().numpy(), g["out/ref_sphere_coarse"])
    compare_render(res, g, "render_refsphere_test")


def test_render_blender_rays_golden():
    from brdf_nerf_amd import render_rays
    g = load_golden("render_lambert_blender")
    cfg = mini(data="blender")
    model = build_model(cfg, 11)
    with Replay(replay_list(g)):
        res, _ = render_rays({"coarse": model}, make_args(cfg), torch.from_numpy(g["rays"]).to(DEV), None)
    # blender rays reach |xyz| ~ 6: the top PE octave's argument (2^9 * 6 rad) has an fp32 ulp of 2.4e-4, so a 1-ulp
    # difference in a sample depth moves per-sample outputs by ~1e-3 and ray-level ones by ~1e-4: held to 3e-4.
    compare_render(res, g, "render_lambert_blender", ray_tol=(3e-4, 1e-4))


def test_adam_matches_torch():
    from brdf_nerf_amd import functions as Fn
    g = torch.Generator().manual_seed(0)
    n = 100003 // 4 * 4
    p0 = torch.randn(n, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=5e-4)
    p, m, v = p0.clone().to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        gr = torch.randn(n, generator=g)
        p_ref.grad = gr.clone()
        opt.step()
        Fn.adam_step(p, gr.to(DEV), m, v, step, 5e-4)
    assert_close(p, p_ref.detach(), 1e-5, 1e-6, "adam")


# ------------------------------------------------------------------------------------------------ analytic normals
CONFIGS_AN = {
    "rpv111_nan": dict(funcM=1, funcF=1, funcH=1, normal="analystic"),
    "hapke_bc": dict(b=1, c=1, normal="analystic"),
}


@pytest.mark.parametrize("name", list(CONFIGS_AN))
def test_field_forward_analytic_normal_golden_fp32(name):
    """normal_an = -normalize(d sigma/d xyz): explicit adjoint chain vs the reference's autograd (golden)."""
    g = load_golden(f"field_{name}_F64")
    cfg = mini(**CONFIGS_AN[name])
    model = build_model(cfg, 11)
    xyz = torch.from_numpy(g["xyz"]).to(DEV)
    with torch.no_grad():
        out = model(xyz, apply_brdf=True, apply_theta=True, nr_an_on=True)
        out0 = model(xyz, apply_brdf=False, nr_an_on=True)
    assert_close(out, g["out_brdf"], 2e-4, 2e-5, "out_brdf")
    assert_close(out0, g["out_nobrdf"], 2e-4, 2e-5, "out_nobrdf")


def test_field_forward_analytic_normal_F512_fp32_and_half():
    g = load_golden("field_rpv111_nan_F512")
    cfg = FieldConfig(**CONFIGS_AN["rpv111_nan"])
    xyz = torch.from_numpy(g["xyz"]).to(DEV)
    with torch.no_grad():
        out = build_model(cfg, 12)(xyz, apply_brdf=True, nr_an_on=True)
    assert_close(out, g["out_brdf"], 5e-4, 5e-5, "F512 out")
    ref_n = torch.from_numpy(g["out_brdf"][:, 4:7])
    for dtype in ("bf16", "fp16"):
        with torch.no_grad():
            out16 = build_model(cfg, 12, dtype)(xyz, apply_brdf=True, nr_an_on=True).cpu()
        cos = (out16[:, 4:7] * ref_n).sum(-1)
        diag(f"{dtype} analytic normal F512: min cosine vs reference {float(cos.min()):.6f}")
        assert float(cos.min()) > HALF_BOUNDS[dtype]["ncos"]      # bf16: within ~11 degrees worst case; fp16: ~1.8 degrees


def test_sigma_grad_matches_oracle_many_points():
    """Raw d sigma/d xyz through the public module path at a ragged size (not a tile multiple), F=256."""
    cfg = FieldConfig(feat=256, normal="analystic")
    model = build_model(cfg, 5)
    p = tparams(cfg, 5, torch.float64)
    xyz = torch.rand(777, 3, generator=torch.Generator().manual_seed(9)) * 2 - 1
    ref = -OF.l2_normalize(OF.sigma_grad_closed_form(p, cfg, xyz.double()))
    with torch.no_grad():
        out = model(xyz.to(DEV), nr_an_on=True)
    cos = (out[:, 4:7].cpu().double() * ref).sum(-1)
    assert float(cos.min()) > 1 - 1e-5, float(cos.min())


@pytest.mark.parametrize("name", list(CONFIGS_AN))
def test_render_rays_analytic_normal_golden_fp32(name):
    from brdf_nerf_amd import render_rays
    g = load_golden(f"render_{name}_test")
    cfg = mini(**CONFIGS_AN[name])
    model = build_model(cfg, 11)
    with torch.no_grad(), Replay(replay_list(g)) as rp:
        res, brdf_type = render_rays({"coarse": model}, make_args(cfg), torch.from_numpy(g["rays"]).to(DEV), None, mode="test",
                                     apply_brdf=True, apply_theta=True, cos_irra_on=True)
        assert rp.draws == []
    assert brdf_type == str(g["brdf_type"])
    assert {k[4:] for k in g if k.startswith("out/")} == set(res)
    PER = PER_SAMPLE + ("normal_an",)
    # fp64 evaluation of the same algorithm: with random-init analytic normals the per-ray BRDF reaches 1e4 (grazing
    # clamp at 1e-5) and rgb is hypersensitive to the normal, so rgb is judged against the reference's own fp32 error.
    p64 = tparams(cfg, 11, torch.float64)
    truth, _ = ORD.render_rays(p64, cfg, torch.from_numpy(g["rays"]).double(), ORD.Randoms(replay=replay_list(g)), mode="test",
                               apply_brdf=True, apply_theta=True, cos_irra_on=True)
    bad = []
    for k in sorted(res):
        ref = g["out/" + k]
        if k == "sort_idx_coarse":
            assert np.array_equal(res[k].cpu().numpy(), ref)
            continue
        base = k[:-7]
        got = res[k].cpu().double().numpy()
        err = np.abs(got - ref)
        diag(f"render_{name}_test(an) {k}: max|err| {err.max():.3e} scale {np.abs(ref).max():.3e}")
        if base in ("rgb", "brdf"):
            t = truth[k].detach().numpy()
            e_mine, e_ref = np.abs(got - t).max(), np.abs(ref - t).max()
            diag(f"render_{name}_test(an) {k}: err vs fp64 truth: hip {e_mine:.3e} reference-fp32 {e_ref:.3e}")
            if not e_mine <= max(3 * e_ref, 1e-5 * max(1.0, np.abs(t).max())):
                bad.append(f"{k}: hip {e_mine:.3e} vs reference {e_ref:.3e} (fp64 truth)")
            continue
        rtol, atol = (1e-2, 1e-2) if base in PER else ((1e-4, 1e-4) if base in RAY_HEADLINE else (2e-3, 5e-4))
        if base == "hpk_scl":
            rtol = 1e-2
        if not np.all(err <= atol + rtol * np.abs(ref)):
            bad.append(f"{k}: {err.max():.3e}")
    assert not bad, bad


@pytest.mark.parametrize("name", list(CONFIGS_AN))
def test_field_backward_through_analytic_normals_fp32(name):
    """Double backward: d(loss)/d(params) when the loss depends on normal_an = -normalize(d sigma/d xyz).
    Oracle: autograd with create_graph=True (what the reference does, spsbrdfnerf.py:648-660)."""
    cfg = mini(**CONFIGS_AN[name])
    flags = dict(apply_brdf=True, apply_theta=True, nr_an_on=True)
    model, p, out, ref = _field_grads(cfg, 11, "fp32", 200, flags)
    assert_close(out, ref, 2e-4, 2e-5, "out")
    for k, v in model.named_parameters():
        want = p[k].grad
        scale = float(want.abs().max())
        err = float((v.grad.cpu() - want).abs().max())
        diag(f"field_bwd_an_{name} {k}: err {err:.3e} scale {scale:.3e}")
        assert err <= 1e-3 * scale + 1e-7, f"{k}: err {err:.3e} scale {scale:.3e}"


def test_field_backward_through_analytic_normals_only_normal_loss_F512():
    """Loss that depends ONLY on the normal channels (isolates the adjoint-chain backward), F=512, ragged point count."""
    cfg = FieldConfig(normal="analystic")
    model = build_model(cfg, 6)
    p = tparams(cfg, 6)
    for v in p.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(77)
    xyz = torch.rand(150, 3, generator=g) * 2 - 1
    coef = torch.randn(150, 3, generator=g)
    ref = OF.field_forward(p, cfg, xyz, nr_an_on=True)
    (ref[:, 4:7] * coef).sum().backward()
    out = model(xyz.to(DEV), nr_an_on=True)
    (out[:, 4:7] * coef.to(DEV)).sum().backward()
    gscale = max(float(v.grad.abs().max()) for v in p.values() if v.grad is not None)
    for k, v in model.named_parameters():
        want = p[k].grad
        if want is None:
            continue
        scale = float(want.abs().max())
        err = float((v.grad.cpu() - want).abs().max())
        diag(f"field_bwd_an_only_F512 {k}: err {err:.3e} scale {scale:.3e}")
        # sigma bias: a sum of cancelling per-point terms (true value ~1e-7 of the largest gradient): absolute floor
        assert err <= 2e-3 * scale + 1e-6 * gscale, f"{k}: err {err:.3e} scale {scale:.3e} (global {gscale:.3e})"


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("which", ["all_channels", "normals_only"])
def test_field_backward_through_analytic_normals_half(which, dtype):
    """The double backward in the 16-bit modes at F = 512 against the fp32 oracle's autograd: cosine per parameter tensor.
    'normals_only' isolates the adjoint-chain backward (in fp16: its own loss scale, the zbar hand-over to the primal chain
    at a third scale); 'all_channels' mixes it with the primal seeds."""
    cfg = FieldConfig(funcM=1, funcF=1, funcH=1, normal="analystic")
    flags = dict(apply_brdf=True, apply_theta=True, nr_an_on=True)
    model = build_model(cfg, 6, dtype)
    p = tparams(cfg, 6)
    for v in p.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(78)
    B = 1500
    xyz = torch.rand(B, 3, generator=g) * 2 - 1
    ref = OF.field_forward(p, cfg, xyz, **flags)
    coef = torch.randn(ref.shape, generator=g)
    if which == "normals_only":
        coef[:, :4] = 0
        coef[:, 7:] = 0
    (ref * coef).sum().backward()
    out = model(xyz.to(DEV), **flags)
    (out * coef.to(DEV)).sum().backward()
    worst, bound = 1.0, (0.97 if dtype == "bf16" else 0.995)
    for k, v in model.named_parameters():
        want = p[k].grad
        if want is None or float(want.abs().max()) == 0.0:
            continue
        got = v.grad.cpu().flatten()
        assert bool(torch.isfinite(got).all()), k
        want = want.flatten()
        if float(want.norm()) < 1e-6 * max(float(x.grad.norm()) for x in p.values() if x.grad is not None):
            continue                                   # sums of cancelling per-point terms (sigma bias): direction undefined
        cos = float((want * got).sum() / (want.norm() * got.norm() + 1e-30))
        ratio = float(got.norm() / (want.norm() + 1e-30))
        diag(f"{dtype} field backward through analytic normals ({which}) {k}: cosine {cos:.6f}, norm ratio {ratio:.4f}")
        worst = min(worst, cos)
        assert cos > bound and 0.9 < ratio < 1.1, f"{k}: cosine {cos}, norm ratio {ratio}"


@pytest.mark.parametrize("name", list(CONFIGS_AN))
def test_render_rays_train_analytic_normal_golden_fp32(name):
    """End-to-end training step gradients with --normal analystic (BASELINE config 3 shape of graph) vs the reference."""
    from brdf_nerf_amd import render_rays
    g = load_golden(f"render_{name}_train")
    cfg = mini(**CONFIGS_AN[name])
    model = build_model(cfg, 11)
    kw = dict(valid_depth=torch.from_numpy(g["tgt/valid_depth"]).to(DEV), target_depths=torch.from_numpy(g["tgt/depths"]).to(DEV),
              target_std=torch.from_numpy(g["tgt/depth_std"]).to(DEV))
    with Replay(replay_list(g)):
        res, _ = render_rays({"coarse": model}, make_args(cfg), torch.from_numpy(g["rays"]).to(DEV), None, mode="train",
                             apply_brdf=True, apply_theta=True, cos_irra_on=True, **kw)
    tgt = torch.from_numpy(g["tgt/rgbs"]).to(DEV)
    loss = torch.mean((res["rgb_coarse"] - tgt) ** 2) + 0.01 * torch.mean(res["depth_coarse"])
    assert_close(loss, g["loss"], 2e-3, 1e-6, "loss")
    loss.backward()
    for k, v in model.named_parameters():
        ref = g[f"grad/{k}"]
        got = v.grad.cpu().numpy() if v.grad is not None else np.zeros_like(ref)
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(got - ref).max())
        diag(f"render_{name}_train(an) grad {k}: err {err:.3e} scale {scale:.3e}")
        assert err <= 2e-2 * scale + 1e-9, f"{k}: err {err:.3e} scale {scale:.3e}"


# ------------------------------------------------------------------------------------------------ --input_viewdir
VIEWDIR = {"viewdir": dict(input_viewdir=1, funcM=1, funcF=1, funcH=1, normal="learned"),
           "viewdir_nomap": dict(input_viewdir=1, mapping=False)}


@pytest.mark.parametrize("tag", list(VIEWDIR))
def test_field_input_viewdir_golden_fp32(tag):
    """--input_viewdir 1 (spsbrdfnerf.py:458,506-510,689-692): the rgb head's first layer reads cat([feats, mapping(dir)]).
    Here the direction is one more K segment of that layer's MFMA product; outputs and parameter gradients (including the
    direction columns of rgb_from_xyzdir.0.weight, written in place at column offset F) against the reference."""
    g = load_golden(f"field_{tag}_F64")
    cfg = mini(**VIEWDIR[tag])
    model = build_model(cfg, 14)
    assert model.rgb_from_xyzdir[0].weight.shape[1] == cfg.feat + cfg.dir_dim
    xyz, dirs = torch.from_numpy(g["xyz"]).to(DEV), torch.from_numpy(g["dirs"]).to(DEV)
    out = model(xyz, input_dir=dirs, apply_brdf=True, apply_theta=True, nr_lr_on=cfg.normal == "learned")
    assert_close(out, g["out_brdf"], 1e-4, 1e-5, "out")
    (out * torch.from_numpy(g["coef"]).to(DEV)).sum().backward()
    for k, v in model.named_parameters():
        ref = g[f"grad/{k}"]
        got = v.grad.cpu().numpy() if v.grad is not None else np.zeros_like(ref)
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(got - ref).max())
        diag(f"field_{tag} grad {k}: err {err:.3e} scale {scale:.3e}")
        assert err <= 2e-4 * scale + 1e-9, f"{k}: err {err:.3e} scale {scale:.3e}"
    with pytest.raises(ValueError):
        model(xyz, apply_brdf=True)                       # no direction given
    with torch.no_grad():                                 # sigma does not read the direction
        sig = model(xyz, sigma_only=True)
    assert_close(sig[:, 0], out[:, 3].detach(), 1e-6, 1e-6, "sigma")


def test_render_rays_input_viewdir_golden_fp32():
    """render_rays with --input_viewdir 1, train mode, two view directions in the batch: the kernels read rays_d of the
    sample's ray (rendering.py:96,121 repeat_interleave) - every key of the reference's dict, loss and gradients."""
    from brdf_nerf_amd import render_rays
    g = load_golden("render_viewdir_train")
    cfg = mini(**VIEWDIR["viewdir"])
    model = build_model(cfg, 11)
    with Replay(replay_list(g)) as rp:
        res, brdf_type = render_rays({"coarse": model}, make_args(cfg), torch.from_numpy(g["rays"]).to(DEV), None, mode="train",
                                     apply_brdf=True, apply_theta=True, cos_irra_on=True)
        assert rp.draws == []
    assert brdf_type == str(g["brdf_type"])
    assert {k[4:] for k in g if k.startswith("out/")} == set(res)
    compare_render(res, g, "render_viewdir_train", ray_tol=(1e-4, 2e-5))
    loss = torch.mean((res["rgb_coarse"] - torch.from_numpy(g["targets"]).to(DEV)) ** 2) + 0.01 * torch.mean(res["depth_coarse"])
    assert_close(loss, g["loss"], 1e-4, 1e-7, "loss")
    loss.backward()
    for k, v in model.named_parameters():
        ref = g[f"grad/{k}"]
        got = v.grad.cpu().numpy() if v.grad is not None else np.zeros_like(ref)
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(got - ref).max())
        diag(f"render_viewdir_train grad {k}: err {err:.3e} scale {scale:.3e}")
        assert err <= 5e-3 * scale + 1e-9, f"{k}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_field_input_viewdir_half_tracks_fp32(dtype):
    """The 16-bit modes with the direction segment at F=512: outputs within the stated half bounds of the fp32 mode, and the
    gradient of the direction columns points the same way."""
    cfg = FieldConfig(feat=512, input_viewdir=1, funcM=1, funcF=1, funcH=1, normal="learned")
    g = torch.Generator().manual_seed(2)
    xyz = (torch.rand(1500, 3, generator=g) * 2 - 1).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(1500, 3, generator=g), dim=-1).to(DEV)
    coef = torch.randn(1500, 13, generator=g).to(DEV)
    outs, grads = {}, {}
    for dt in ("fp32", dtype):
        model = build_model(cfg, 5, dt)
        out = model(xyz, input_dir=dirs, apply_brdf=True, nr_lr_on=True)
        (out[:, :coef.shape[1]] * coef[:, :out.shape[1]]).sum().backward()
        outs[dt], grads[dt] = out.detach(), model.rgb_from_xyzdir[0].weight.grad[:, cfg.feat:].clone()
    b = HALF_BOUNDS[dtype]
    assert float((outs[dtype][:, :3] - outs["fp32"][:, :3]).abs().max()) <= b["rgb"]
    cos = torch.nn.functional.cosine_similarity(grads[dtype].flatten(), grads["fp32"].flatten(), dim=0)
    diag(f"viewdir {dtype}: direction-column gradient cosine {float(cos):.5f}")
    assert float(cos) >= b["cos"]



# ------------------------------------------------------------------------------------------------ --beta
BETA = {"beta": dict(beta=True, funcM=1, funcF=1, funcH=1, normal="learned"),          # rgb + beta + 3 RPV heads: three head passes
        "beta_viewdir_relu": dict(beta=True, input_viewdir=1, siren=False, t_dim=6)}  # both extra-input segments in one tile


@pytest.mark.parametrize("tag", list(BETA))
def test_field_beta_golden_fp32(tag):
    """--beta (spsbrdfnerf.py:571-575,708-711): channel 4 = Softplus(Linear(nl(Linear(cat([xyz_features, t embedding]))))).
    The embedding is one more K segment of head 1's first layer; outputs, parameter gradients (the embedding columns of
    beta_from_xyz.0.weight in place at column offset F) and the gradient w.r.t. the embedding input against the reference."""
    g = load_golden(f"field_{tag}_F64")
    cfg = mini(**BETA[tag])
    model = build_model(cfg, 15)
    xyz, dirs = torch.from_numpy(g["xyz"]).to(DEV), torch.from_numpy(g["dirs"]).to(DEV)
    t_in = torch.from_numpy(g["t_in"]).to(DEV).requires_grad_(True)
    out = model(xyz, input_dir=dirs, input_t=t_in, apply_brdf=True, apply_theta=True, nr_lr_on=cfg.normal == "learned")
    assert_close(out, g["out_brdf"], 1e-4, 1e-5, "out")
    (out * torch.from_numpy(g["coef"]).to(DEV)).sum().backward()
    ref = g["d_t_in"]
    assert float((t_in.grad.cpu() - torch.from_numpy(ref)).abs().max()) <= 2e-4 * float(np.abs(ref).max()), "d_t_in"
    for k, v in model.named_parameters():
        ref = g[f"grad/{k}"]
        got = v.grad.cpu().numpy() if v.grad is not None else np.zeros_like(ref)
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(got - ref).max())
        diag(f"field_{tag} grad {k}: err {err:.3e} scale {scale:.3e}")
        assert err <= 2e-4 * scale + 1e-9, f"{k}: err {err:.3e} scale {scale:.3e}"
    with pytest.raises(ValueError):
        model(xyz, input_dir=dirs, apply_brdf=True)       # no embedding given


def test_render_rays_beta_golden_fp32():
    """render_rays with --beta, models['t'](ts) (rendering.py:226-229), train mode: every key of the reference's dict
    (beta_coarse among them), the uncertainty-aware loss (metrics.py:24-28), gradients of the parameters and of the
    embedding table (the per-ray embedding gradient is the kernel's per-sample one summed over both sample sets)."""
    from brdf_nerf_amd import render_rays, losses
    g = load_golden("render_beta_train")
    cfg = mini(**BETA["beta"])
    model = build_model(cfg, 11)
    emb = torch.nn.Embedding(*g["emb"].shape).to(DEV)
    with torch.no_grad():
        emb.weight.copy_(torch.from_numpy(g["emb"]))
    ts = torch.from_numpy(g["ts"]).to(DEV)
    with Replay(replay_list(g)) as rp:
        res, brdf_type = render_rays({"coarse": model, "t": emb}, make_args(cfg), torch.from_numpy(g["rays"]).to(DEV), ts,
                                     mode="train", apply_brdf=True, apply_theta=True, cos_irra_on=True)
        assert rp.draws == []
    assert brdf_type == str(g["brdf_type"])
    assert {k[4:] for k in g if k.startswith("out/")} == set(res)
    compare_render(res, g, "render_beta_train", ray_tol=(1e-4, 2e-5))
    tgt = torch.from_numpy(g["targets"]).to(DEV)
    l_color, l_logbeta = losses.uncertainty_aware_loss(res["rgb_coarse"], res["weights_coarse"], res["beta_coarse"], tgt)
    assert_close(l_color, g["loss_color"], 1e-4, 1e-7, "loss_color")
    assert_close(l_logbeta, g["loss_logbeta"], 1e-4, 1e-7, "loss_logbeta")
    loss = l_color + l_logbeta + 0.01 * torch.mean(res["depth_coarse"])
    loss.backward()
    ref = g["d_emb"]
    err = float((emb.weight.grad.cpu() - torch.from_numpy(ref)).abs().max())
    diag(f"render_beta_train d_emb: err {err:.3e} scale {float(np.abs(ref).max()):.3e}")
    assert err <= 5e-3 * float(np.abs(ref).max()) + 1e-9
    for k, v in model.named_parameters():
        ref = g[f"grad/{k}"]
        got = v.grad.cpu().numpy() if v.grad is not None else np.zeros_like(ref)
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(got - ref).max())
        diag(f"render_beta_train grad {k}: err {err:.3e} scale {scale:.3e}")
        assert err <= 5e-3 * scale + 1e-9, f"{k}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_field_beta_half_tracks_fp32(dtype):
    """The 16-bit modes with the embedding segment and five heads at F=512: beta within the rgb bound (relative), the
    embedding-input gradient and the embedding columns' gradient point the same way as in the fp32 mode."""
    cfg = FieldConfig(feat=512, beta=True, funcM=1, funcF=1, funcH=1, normal="learned")
    g = torch.Generator().manual_seed(4)
    xyz = (torch.rand(1500, 3, generator=g) * 2 - 1).to(DEV)
    t0 = torch.randn(1500, cfg.t_dim, generator=g).to(DEV)
    outs, gw, gt = {}, {}, {}
    for dt in ("fp32", dtype):
        model = build_model(cfg, 5, dt)
        t_in = t0.clone().requires_grad_(True)
        out = model(xyz, input_t=t_in, apply_brdf=True, nr_lr_on=True)
        coef = torch.randn(out.shape, generator=torch.Generator().manual_seed(6)).to(DEV)
        (out * coef).sum().backward()
        outs[dt], gw[dt], gt[dt] = out.detach(), model.beta_from_xyz[0].weight.grad[:, cfg.feat:].clone(), t_in.grad.clone()
    b = HALF_BOUNDS[dtype]
    rel = float(((outs[dtype][:, 4] - outs["fp32"][:, 4]).abs() / (outs["fp32"][:, 4].abs() + 0.1)).max())
    cw = float(torch.nn.functional.cosine_similarity(gw[dtype].flatten(), gw["fp32"].flatten(), dim=0))
    ct = float(torch.nn.functional.cosine_similarity(gt[dtype].flatten(), gt["fp32"].flatten(), dim=0))
    diag(f"beta {dtype}: max rel err of beta {rel:.3e}, cos(d W_t) {cw:.5f}, cos(d t_embed) {ct:.5f}")
    assert rel <= 3 * b["rgb"] and cw >= b["cos"] and ct >= b["cos"]



def test_field_everything_on_against_oracle():
    """The widest head set: --beta + --input_viewdir + RPV funcM/F/H + analytic AND learned normals (20 output channels,
    five heads in three passes, both extra-input segments, the double backward through the analytic normals), F=128, ragged
    point count, against the oracle (autograd with create_graph=True): outputs, every parameter gradient, d/d t_embed."""
    cfg = mini(feat=128, beta=True, input_viewdir=1, funcM=1, funcF=1, funcH=1, normal="analystic_learned")
    flags = dict(apply_brdf=True, apply_theta=True, nr_an_on=True, nr_lr_on=True)
    model = build_model(cfg, 21)
    p = tparams(cfg, 21)
    for v in p.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(77)
    B = 333
    xyz = torch.rand(B, 3, generator=g) * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=-1)
    t_ref = torch.randn(B, cfg.t_dim, generator=g).requires_grad_(True)
    ref = OF.field_forward(p, cfg, xyz, dirs=dirs, t_embed=t_ref, **flags)
    assert ref.shape[1] == 20
    coef = torch.randn(ref.shape, generator=g)
    (ref * coef).sum().backward()
    t_gpu = t_ref.detach().to(DEV).requires_grad_(True)
    out = model(xyz.to(DEV), input_dir=dirs.to(DEV), input_t=t_gpu, **flags)
    assert_close(out, ref, 2e-4, 2e-5, "out")
    (out * coef.to(DEV)).sum().backward()
    scale = float(t_ref.grad.abs().max())
    assert float((t_gpu.grad.cpu() - t_ref.grad).abs().max()) <= 2e-4 * scale, "d_t_embed"
    for k, v in model.named_parameters():
        want = p[k].grad
        scale = float(want.abs().max())
        err = float((v.grad.cpu() - want).abs().max())
        diag(f"field_everything_on {k}: err {err:.3e} scale {scale:.3e}")
        assert err <= 1e-3 * scale + 1e-7, f"{k}: err {err:.3e} scale {scale:.3e}"



# ------------------------------------------------------------------------------------------------ fused trainer
@pytest.mark.parametrize("name,with_depth", [("lambert", False), ("lambert", True), ("rpv111_nlr", True), ("rpv111_nan", False),
                                             ("rpv111_nlr_multibrdf", False), ("hapke_bct_multibrdf", True),
                                             ("rpv111_nlr_viewdir", False), ("hapke_bct_beta", True)])
def test_fused_trainer_matches_autograd_path(name, with_depth):
    """FusedTrainer.step (the path bench.py times) == render_rays + losses + loss.backward() + torch.optim.Adam, same draws.
    With depth priors the reference's quirk 7 (target_std == 0) is used, so the ground-truth-guided rows do not depend on
    their uniform draws (the trainer draws (R,G) instead of (n_valid,G) to avoid a host sync)."""
    from brdf_nerf_amd import render_rays, losses
    from brdf_nerf_amd.trainer import FusedTrainer
    # --MultiBRDF: one BRDF per SAMPLE (spsbrdfnerf.py:289-307,350-352): the loss reads per-sample field outputs directly
    allc = dict(CONFIGS, **CONFIGS_AN, rpv111_nlr_multibrdf=dict(CONFIGS["rpv111_nlr"], MultiBRDF=True),
                hapke_bct_multibrdf=dict(CONFIGS["hapke_bct"], MultiBRDF=True),
                rpv111_nlr_viewdir=dict(CONFIGS["rpv111_nlr"], input_viewdir=1),
                hapke_bct_beta=dict(CONFIGS["hapke_bct"], beta=True))
    cfg = mini(**allc[name])
    args = make_args(cfg)
    g = torch.Generator().manual_seed(3)
    R, S, G = 96, cfg.n_samples, cfg.guided_samples
    gold = load_golden("render_lambert_train")
    rays = torch.from_numpy(gold["rays"])[:R // 2].repeat(2, 1).contiguous().to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.6).float().to(DEV)
    depths = torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1).to(DEV)
    dstd = torch.zeros(R, device=DEV)
    n_valid = int(valid.sum())
    draws = [torch.rand(R, S, generator=g), torch.randn(R, S, generator=g), torch.rand(R, G, generator=g)]
    u_t = torch.rand(R, G, generator=g)
    draws_ref = list(draws) + ([u_t[:n_valid]] if with_depth else []) + [torch.randn(R, S + G, generator=g)]
    draws_tr = list(draws) + ([u_t] if with_depth else []) + [draws_ref[-1]]
    flags = dict(apply_brdf=name != "lambert", apply_theta=True, cos_irra_on=name != "lambert")
    dk = dict(valid_depth=valid, target_depths=depths, target_std=dstd) if with_depth else {}

    ma = build_model(cfg, 11)
    opt = torch.optim.Adam(ma.parameters(), lr=5e-4)
    models_a, ts = {"coarse": ma}, None
    if cfg.beta:      # --beta: the autograd path evaluates the beta head (models['t'](ts)); SNerfLoss never reads it, so it gets
        models_a["t"] = torch.nn.Embedding(5, cfg.t_dim).to(DEV)      # no gradient - the fused step leaves the head out
        ts = torch.randint(0, 5, (R,), generator=g).to(DEV)
    with Replay(draws_ref):
        res, _ = render_rays(models_a, args, rays, ts, mode="train", **flags, **dk)
    assert ("beta_coarse" in res) == bool(cfg.beta)
    loss_a = losses.snerf_loss(res["rgb_coarse"], rgbs)
    if with_depth:
        loss_a = loss_a + losses.depth_loss(res["z_vals_coarse"], res["depth_coarse"], res["weights_coarse"], depths[:, 0],
                                            depths[:, 1], valid, dstd, 10.0)
    loss_a.backward()
    grads_a = {k: v.grad.clone() for k, v in ma.named_parameters() if v.grad is not None}
    opt.step()

    mb = build_model(cfg, 11)
    tr = FusedTrainer(mb, args, lr=5e-4, ds_lambda=10.0 if with_depth else 0.0)
    with Replay(draws_tr):
        loss_b, _ = tr.step(rays, rgbs, valid_depth=valid if with_depth else None, depths=depths if with_depth else None,
                            depth_std=dstd if with_depth else None, **flags)
    assert_close(loss_b, loss_a.detach(), 1e-5, 1e-7, "loss")
    if cfg.beta:
        assert models_a["t"].weight.grad is None or float(models_a["t"].weight.grad.abs().max()) == 0.0
        for k in tr.grad_views:
            if k.startswith("beta_from_xyz."):
                assert k not in grads_a or float(grads_a[k].abs().max()) == 0.0
                assert float(tr.grad_views[k].abs().max()) == 0.0, k
    for k, ga in grads_a.items():
        gb = tr.grad_views[k]
        scale = float(ga.abs().max())
        assert float((gb - ga).abs().max()) <= 1e-4 * scale + 1e-9, f"grad {k}"
    # Adam's first step is lr * g / (|g| + eps): where |g| is not far above the agreed gradient tolerance the two updates
    # may differ by a sizeable fraction of lr; entries with a well-determined gradient must agree to fp32 rounding
    for (k, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        diff = (pa - pb).detach().abs()
        assert float(diff.max()) <= 2.1 * 5e-4, f"param {k} after Adam: {float(diff.max()):.3e}"
        if k in grads_a:
            firm = grads_a[k].abs() > 1e-2 * float(grads_a[k].abs().max())
            if bool(firm.any()):
                assert float(diff[firm].max()) <= 6e-6, f"param {k} after Adam (firm gradients): {float(diff[firm].max()):.3e}"
    sd = mb.state_dict()                       # flat-buffer views keep the reference's checkpoint contract
    assert list(sd) == [k for k, _, _ in cfg.param_shapes()]


@pytest.mark.parametrize("name", ["rpv111_nlr", "rpv111_nan"])
def test_fused_trainer_regularisers_match_autograd_path(name):
    """NormalRegLoss + HardSurfaceLoss (metrics.py:179-290) inside the fused step: the per-sample normals become extra
    leaves of the loss glue and their gradients join d_out; compared with autograd through render_rays."""
    from brdf_nerf_amd import render_rays, losses
    from brdf_nerf_amd.trainer import FusedTrainer
    allc = dict(CONFIGS, **CONFIGS_AN)
    cfg = mini(**allc[name])
    args = make_args(cfg)
    g = torch.Generator().manual_seed(8)
    R, S, G = 64, cfg.n_samples, cfg.guided_samples
    gold = load_golden("render_lambert_train")
    rays = torch.from_numpy(gold["rays"])[:R // 2].repeat(2, 1).contiguous().to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    draws = [torch.rand(R, S, generator=g), torch.randn(R, S, generator=g), torch.rand(R, G, generator=g),
             torch.randn(R, S + G, generator=g)]
    flags = dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)
    key = "normal_an" if cfg.normal == "analystic" else "normal_lr"
    lam = dict(nr_reg_an_lambda=0.3) if key == "normal_an" else dict(nr_reg_lr_lambda=0.3)

    ma = build_model(cfg, 5)
    with Replay(list(draws)):
        res, _ = render_rays({"coarse": ma}, args, rays, None, mode="train", **flags)
    loss_a = losses.snerf_loss(res["rgb_coarse"], rgbs)
    loss_a = loss_a + losses.normal_reg_loss(res[f"{key}_coarse"], res["weights_coarse"], res["rays_d_coarse"].squeeze(1), 0.3)[0]
    loss_a = loss_a + losses.hard_surface_loss(res["z_vals_coarse"], res["depth_coarse"], res["weights_coarse"], 0.5)
    loss_a.backward()

    mb = build_model(cfg, 5)
    tr = FusedTrainer(mb, args, lr=5e-4, hs_lambda=0.5, **lam)
    with Replay(list(draws)):
        loss_b, _ = tr.step(rays, rgbs, **flags)
    assert_close(loss_b, loss_a.detach(), 1e-5, 1e-7, "loss")
    for k, v in ma.named_parameters():
        if v.grad is None:
            continue
        gb = tr.grad_views[k]
        scale = float(v.grad.abs().max())
        assert float((gb - v.grad).abs().max()) <= 2e-4 * scale + 1e-9, f"grad {k}"


@pytest.mark.parametrize("name", ["lambert", "rpv111_nlr"])
def test_fused_trainer_gsam_only_matches_autograd_path(name):
    """gsam_only training (main.py:201-203): the step renders and back-propagates through the guided samples alone;
    compared with autograd through render_rays(gsam_only=True) on the same draws."""
    from brdf_nerf_amd import render_rays, losses
    from brdf_nerf_amd.trainer import FusedTrainer
    cfg = mini(**CONFIGS[name])
    args = make_args(cfg)
    g = torch.Generator().manual_seed(12)
    R, S, G = 64, cfg.n_samples, cfg.guided_samples
    rays = torch.from_numpy(load_golden("render_lambert_train")["rays"])[:R // 2].repeat(2, 1).contiguous().to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    draws = [torch.rand(R, S, generator=g), torch.randn(R, S, generator=g), torch.rand(R, G, generator=g),
             torch.randn(R, G, generator=g)]
    flags = dict(apply_brdf=name != "lambert", apply_theta=True, cos_irra_on=name != "lambert")
    ma = build_model(cfg, 5)
    with Replay(list(draws)):
        res, _ = render_rays({"coarse": ma}, args, rays, None, mode="train", gsam_only=True, **flags)
    loss_a = losses.snerf_loss(res["rgb_coarse"], rgbs)
    loss_a.backward()
    mb = build_model(cfg, 5)
    tr = FusedTrainer(mb, args, lr=5e-4)
    with Replay(list(draws)):
        loss_b, _ = tr.step(rays, rgbs, gsam_only=True, **flags)
    assert_close(loss_b, loss_a.detach(), 1e-5, 1e-7, "loss")
    for k, v in ma.named_parameters():
        if v.grad is None:
            continue
        scale = float(v.grad.abs().max())
        assert float((tr.grad_views[k] - v.grad).abs().max()) <= 2e-4 * scale + 1e-9, f"grad {k}"


@pytest.mark.parametrize("name", ["rpv111_nlr", "hapke_bct"])
def test_fused_trainer_sun_visibility_matches_autograd_path(name):
    """--sun_v analystic inside the fused step (sun pass rendering.py:244-259 + per-sample irradiance spsbrdfnerf.py:259-273),
    gsam_only=True, against autograd through render_rays on the same draws."""
    from brdf_nerf_amd import render_rays, losses
    from brdf_nerf_amd.trainer import FusedTrainer
    if name not in CONFIGS:
        pytest.skip(f"no config {name}")
    cfg = mini(**dict(CONFIGS[name], sun_v="analystic"))
    args = make_args(cfg)
    g = torch.Generator().manual_seed(13)
    R, S, G = 64, cfg.n_samples, cfg.guided_samples
    rays = torch.from_numpy(load_golden("render_lambert_train")["rays"])[:R // 2].repeat(2, 1).contiguous().to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    draws = [torch.rand(R, S, generator=g), torch.randn(R, S, generator=g), torch.rand(R, G, generator=g),
             torch.randn(R, G, generator=g), torch.rand(R, G, generator=g), torch.randn(R, G, generator=g)]
    flags = dict(apply_brdf=True, apply_theta=True, cos_irra_on=False)
    ma = build_model(cfg, 5)
    with Replay(list(draws)) as rp:
        res, _ = render_rays({"coarse": ma}, args, rays, None, mode="train", gsam_only=True, **flags)
        assert rp.draws == []
    loss_a = losses.snerf_loss(res["rgb_coarse"], rgbs)
    loss_a.backward()
    mb = build_model(cfg, 5)
    tr = FusedTrainer(mb, args, lr=5e-4)
    with Replay(list(draws)) as rp:
        loss_b, _ = tr.step(rays, rgbs, gsam_only=True, **flags)
        assert rp.draws == []
    with pytest.raises(NotImplementedError):
        tr.step(rays, rgbs, gsam_only=False, **flags)
    assert_close(loss_b, loss_a.detach(), 1e-5, 1e-7, "loss")
    for k, v in ma.named_parameters():
        if v.grad is None:
            continue
        scale = float(v.grad.abs().max())
        assert float((tr.grad_views[k] - v.grad).abs().max()) <= 2e-4 * scale + 1e-9, f"grad {k}"


@pytest.mark.parametrize("name", ["lambert", "rpv111_nan"])
def test_trainer_coarse_reuse_equals_full_reevaluation(name):
    """reuse_coarse=True (pass-1 evaluation kept, pass 2 only on the guided samples) must give the reference pipeline's
    result (pass 2 re-evaluates all S+G samples): same loss and gradients, step after step."""
    from brdf_nerf_amd.trainer import FusedTrainer
    allc = dict(CONFIGS, **CONFIGS_AN)
    cfg = mini(**allc[name])
    args = make_args(cfg)
    g = torch.Generator().manual_seed(5)
    R, S, G = 160, cfg.n_samples, cfg.guided_samples
    rays = torch.from_numpy(load_golden("render_lambert_train")["rays"])[:32].repeat(5, 1).contiguous().to(DEV)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    flags = dict(apply_brdf=name != "lambert", apply_theta=True, cos_irra_on=name != "lambert")
    steps = [[torch.rand(R, S, generator=g), torch.randn(R, S, generator=g), torch.rand(R, G, generator=g),
              torch.randn(R, S + G, generator=g)] for _ in range(2)]
    out = {}
    for reuse in (True, False):
        tr = FusedTrainer(build_model(cfg, 11), args, lr=5e-4, reuse_coarse=reuse)
        rec = []
        for draws in steps:
            with Replay(list(draws)):
                loss, rgb = tr.step(rays, rgbs, near_far=(0.0, 2.0), **flags)
            rec.append((float(loss), tr.flat_grad.clone()))
        out[reuse] = rec
    for (la, ga), (lb, gb) in zip(out[True], out[False]):
        assert abs(la - lb) <= 2e-4 * abs(lb) + 1e-7, (la, lb)
    ga, gb = out[True][0][1], out[False][0][1]
    assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max()) + 1e-9


# ------------------------------------------------------------------------------------------------ variants
@pytest.mark.parametrize("tag,extra,gs", [("rpv111_nlr_multibrdf", dict(MultiBRDF=True), False), ("rpv111_nlr_gsamonly", dict(), True)])
def test_render_variants_multibrdf_gsamonly_golden(tag, extra, gs):
    """MultiBRDF=1 (one BRDF per sample, models/spsbrdfnerf.py:289-307,350-352) and gsam_only (pass 2 on the guided samples
    only, rendering.py:266-269) against reference goldens."""
    from brdf_nerf_amd import render_rays
    g = load_golden(f"render_{tag}_test")
    cfg = mini(**dict(CONFIGS["rpv111_nlr"], **extra))
    model = build_model(cfg, 11)
    with torch.no_grad(), Replay(replay_list(g)) as rp:
        res, brdf_type = render_rays({"coarse": model}, make_args(cfg), torch.from_numpy(g["rays"]).to(DEV), None, mode="test",
                                     apply_brdf=True, apply_theta=True, cos_irra_on=True, gsam_only=gs)
        assert rp.draws == []
    assert brdf_type == str(g["brdf_type"])
    # per-sample shading / 16 tightly clustered samples: the normal's 1e-3 per-sample differences reach the pixel directly
    compare_render(res, g, f"render_{tag}_test", ray_tol=(5e-4, 2e-4), other_tol=(2e-3, 1e-3))


@pytest.mark.parametrize("mode", ["train", "test"])
@pytest.mark.parametrize("base", ["rpv111_nlr", "lambert"])
def test_render_sun_visibility_pass_golden(base, mode):
    """--sun_v analystic (rendering.py:244-259): transparency along the sun direction from the
    pass-1 surface point as per-sample irradiance, gsam_only=True (the combination the reference accepts), against
    reference goldens incl. the training gradient."""
    from brdf_nerf_amd import render_rays
    g = load_golden(f"render_{base}_sunv_{mode}")
    cfg = mini(**dict(CONFIGS[base], sun_v="analystic"))
    model = build_model(cfg, 11)
    args = make_args(cfg)
    kw = {}
    if mode == "train":
        kw = dict(valid_depth=torch.from_numpy(g["tgt/valid_depth"]).to(DEV), target_depths=torch.from_numpy(g["tgt/depths"]).to(DEV),
                  target_std=torch.from_numpy(g["tgt/depth_std"]).to(DEV))
    with Replay(replay_list(g)) as rp:
        res, brdf_type = render_rays({"coarse": model}, args, torch.from_numpy(g["rays"]).to(DEV), None, mode=mode,
                                     apply_brdf=True, apply_theta=True, cos_irra_on=False, gsam_only=True, **kw)
        assert rp.draws == []
    assert brdf_type == str(g["brdf_type"])
    assert {k[4:] for k in g if k.startswith("out/")} == set(res)
    g = dict(g)
    if "out/hpk_scl_coarse" in g:   # 1 / (4 (n.v + n.sun)) is unbounded where the sum passes through 0: compare its inverse
        assert_close(1.0 / res["hpk_scl_coarse"], 1.0 / g.pop("out/hpk_scl_coarse"), 2e-3, 2e-3, "1/hpk_scl")
    compare_render(res, g, f"render_{base}_sunv_{mode}", ray_tol=(5e-4, 2e-4), other_tol=(2e-3, 1e-3))
    if mode == "train":
        loss = torch.mean((res["rgb_coarse"] - torch.from_numpy(g["tgt/rgbs"]).to(DEV)) ** 2)
        loss.backward()
        for k, v in model.named_parameters():
            ref = g[f"grad/{k}"]
            got = v.grad.cpu() if v.grad is not None else torch.zeros_like(v).cpu()
            scale = max(float(np.abs(ref).max()), 1e-12)
            assert float((got - torch.from_numpy(ref)).abs().max()) <= 5e-3 * scale + 1e-9, k


def test_batched_inference_chunks_concatenate():
    """eval.batched_inference semantics (eval.py:56-76): chunked no-grad render, per-chunk dicts concatenated."""
    from brdf_nerf_amd import render_rays
    from brdf_nerf_amd.evaluate import batched_inference
    cfg = mini()
    model = build_model(cfg, 11)
    args = make_args(cfg)
    args.chunk = 40
    rays = torch.from_numpy(load_golden("render_lambert_test")["rays"]).to(DEV)
    torch.manual_seed(0)
    res = batched_inference({"coarse": model}, rays, None, args)
    assert res["rgb_coarse"].shape == (64, 3) and res["weights_coarse"].shape == (64, 32)
    torch.manual_seed(0)
    with torch.no_grad():
        first, _ = render_rays({"coarse": model}, args, rays[:40], None)
    assert torch.equal(res["rgb_coarse"][:40], first["rgb_coarse"])
    assert not res["rgb_coarse"].requires_grad


@pytest.mark.parametrize("with_depth", [False, True])
def test_lambert_loss_kernel_matches_autograd(with_depth):
    """bn_lambert_loss (shading + SNerfLoss + DepthLoss + gradients in one launch) against the torch glue it replaces;
    inputs straddle both clamp edges and both branches of the depth-subset rule."""
    from brdf_nerf_amd import functions as Fn, losses
    g = torch.Generator().manual_seed(21)
    R, S, C, pad = 300, 96, 4, 0.001
    z = torch.sort(torch.rand(R, S, generator=g) * 2, -1)[0].to(DEV)
    w = torch.softmax(torch.randn(R, S, generator=g) * 2, -1).to(DEV)
    acc = (torch.rand(R, C, generator=g) * 1.4 - 0.2).to(DEV)          # some composited albedos outside [0, 1]
    depth = (w * z).sum(-1)
    rgbs = torch.rand(R, 3, generator=g).to(DEV)
    valid = (torch.rand(R, generator=g) < 0.7).float().to(DEV)
    tdep = (0.8 + 0.4 * torch.rand(R, generator=g)).to(DEV)
    tw = torch.rand(R, generator=g).to(DEV)
    tstd = (0.3 * torch.rand(R, generator=g)).to(DEV)
    a_l, d_l, w_l = acc.clone().requires_grad_(True), depth.clone().requires_grad_(True), w.clone().requires_grad_(True)
    rgb = (a_l[:, :3] * (1 + 2 * pad) - pad * w_l.sum(-1, keepdim=True)).clamp(0.0, 1.0)
    loss = losses.snerf_loss(rgb, rgbs, 1.0)
    if with_depth:
        loss = loss + losses.depth_loss(z, d_l, w_l, tdep, tw, valid, tstd, 10.0)
    ga, gd, gw = torch.autograd.grad(loss, [a_l, d_l, w_l], allow_unused=True)
    kw = dict(valid_depth=valid, target_depth=tdep, target_weight=tw, target_std=tstd, lambda_ds=10.0) if with_depth else {}
    l2, rgb2, da, dd, dw = Fn.lambert_loss(acc, w, z, depth, rgbs, pad, 1.0, **kw)
    assert_close(l2, loss.detach(), 1e-5, 1e-8, "loss")
    assert_close(rgb2, rgb.detach(), 1e-6, 1e-7, "rgb")
    assert_close(da, ga, 1e-5, 1e-10, "d_acc")
    assert_close(dw, gw, 1e-5, 1e-10, "d_weights")
    assert_close(dd, gd if gd is not None else torch.zeros_like(dd), 1e-5, 1e-10, "d_depth")


def test_lambert_loss_kernel_against_reference_golden():
    """bn_lambert_loss directly against the REFERENCE's SNerfLoss + DepthLoss values and gradients
    (tests/golden/loss_snerf_depth.npz, generated by metrics.py:39-61,82-161).  The kernel shades first
    (rgb = acc (1+2p) - p sum_s w, spsbrdfnerf.py:270-272): it is fed acc = (rgb + p sum w) / (1+2p) so that its rgb is the
    fixture's, and its gradients map back by the chain rule (d rgb = d acc / (1+2p); the weights pick up -p sum_c d rgb_c)."""
    from brdf_nerf_amd import functions as Fn
    g = load_golden("loss_snerf_depth")
    t = {k: torch.from_numpy(v).to(DEV) for k, v in g.items()}
    pad = 0.001
    w, z, depth, rgb = t["weights"], t["z"], t["depth"], t["rgb"]
    acc = torch.zeros(rgb.shape[0], 4, device=DEV)
    acc[:, :3] = (rgb + pad * w.sum(-1, keepdim=True)) / (1 + 2 * pad)
    loss, rgb2, d_acc, d_depth, d_w = Fn.lambert_loss(acc, w, z, depth, t["tgt"], pad, 1.0, valid_depth=t["valid_depth"],
                                                      target_depth=t["target_depths"][:, 0], target_weight=t["target_depths"][:, 1],
                                                      target_std=t["target_std"], lambda_ds=10.0)
    assert_close(rgb2, g["rgb"], 1e-5, 1e-6, "rgb")
    assert_close(loss, g["loss_rgb"] + g["loss_ds"], 1e-5, 1e-8, "loss")
    drgb = d_acc[:, :3] / (1 + 2 * pad)
    assert_close(drgb, g["drgb"], 1e-5, 1e-9, "drgb")
    assert_close(d_depth, g["ddepth"], 1e-5, 1e-9, "ddepth")
    want_dw = torch.from_numpy(g["dweights"]).to(DEV) - pad * drgb.sum(-1, keepdim=True)
    assert_close(d_w, want_dw, 1e-5, 1e-9, "dweights")


def test_regulariser_losses_on_device_against_reference_golden():
    """The regulariser glue of the fused step (NormalRegLoss / HardSurfaceLoss / NormalLoss in mask form, losses.py) run ON
    THE DEVICE against the reference's values and gradients (tests/golden/loss_regularisers.npz, metrics.py:179-290)."""
    from brdf_nerf_amd import losses
    g = load_golden("loss_regularisers")
    t = {k: torch.from_numpy(v).to(DEV) for k, v in g.items()}
    w, depth, n_an, n_lr = [t[k].clone().requires_grad_(True) for k in ("weights", "depth", "normal_an", "normal_lr")]
    l_an, perc_an = losses.normal_reg_loss(n_an, w, t["view"], 0.1)
    l_lr, perc_lr = losses.normal_reg_loss(n_lr, w, t["view"], 0.05)
    l_hs = losses.hard_surface_loss(t["z"], depth, w, 0.5)
    l_n1 = losses.normal_loss(w, n_an, n_lr, 0.01, "an_lr")
    l_n3 = losses.normal_loss(w, t["normal_gt"], n_an, 0.01, "an", target_weight=t["target_weight"], valid_depth=t["valid_depth"])
    for name, got in (("l_nr_an", l_an), ("l_nr_lr", l_lr), ("l_hs", l_hs), ("l_n1", l_n1), ("l_n3", l_n3)):
        assert_close(got, g[name], 1e-5, 1e-9, name)
    (l_an + l_lr + l_hs + l_n1 + l_n3).backward()
    assert_close(w.grad, g["d_weights"], 1e-5, 1e-9, "d_weights")
    assert_close(depth.grad, g["d_depth"], 1e-5, 1e-9, "d_depth")
    assert_close(n_an.grad, g["d_normal_an"], 1e-5, 1e-10, "d_normal_an")
    assert_close(n_lr.grad, g["d_normal_lr"], 1e-5, 1e-10, "d_normal_lr")


def test_render_image_against_oracle():
    """evaluate.render_image (chunked full-image render, only the requested keys, PSNR) against the CPU oracle's
    render_rays on the same chunks with the same injected draws: rgb, depth and PSNR."""
    from brdf_nerf_amd.evaluate import render_image
    from oracle import losses as OL
    cfg = mini(**CONFIGS["rpv111_nlr"])
    model = build_model(cfg, 11)
    args = make_args(cfg)
    R, chunk = 70, 24                                  # three chunks, the last one ragged
    S, G = cfg.n_samples, cfg.guided_samples
    gen = torch.Generator().manual_seed(4)
    rays = torch.from_numpy(load_golden("render_lambert_test")["rays"])
    rays = torch.cat([rays, rays.flip(0)], 0)[:R].contiguous()
    tgt = torch.rand(R, 3, generator=gen)
    flags = dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)
    p = tparams(cfg, 11)
    draws, want_rgb, want_depth = [], [], []
    for i in range(0, R, chunk):
        n = min(chunk, R - i)
        d = [torch.rand(n, S, generator=gen), torch.randn(n, S, generator=gen), torch.rand(n, G, generator=gen),
             torch.randn(n, S + G, generator=gen)]
        ref, _ = ORD.render_rays(p, cfg, rays[i:i + n], ORD.Randoms(replay=d), mode="test", **flags)
        want_rgb.append(ref["rgb_coarse"]); want_depth.append(ref["depth_coarse"])
        draws += d
    with Replay(draws):
        img = render_image({"coarse": model}, args, rays.to(DEV), tgt.to(DEV), keys=("rgb", "depth"), chunk=chunk, **flags)
    assert set(img) == {"rgb", "depth", "psnr"}
    want_rgb, want_depth = torch.cat(want_rgb), torch.cat(want_depth)
    diag(f"render_image vs oracle: max |rgb err| {float((img['rgb'].cpu() - want_rgb).abs().max()):.3e}, "
         f"max |depth err| {float((img['depth'].cpu() - want_depth).abs().max()):.3e}")
    # RPV shading of a two-pass render: pass-1 differences of 1e-6 in depth reach the guided samples through the 2^9 PE
    # octave (same bound as the end-to-end golden tests of the BRDF variants)
    assert_close(img["rgb"], want_rgb, 5e-4, 5e-5, "rgb")
    assert_close(img["depth"], want_depth, 1e-4, 2e-5, "depth")
    assert_close(img["psnr"], OL.psnr(want_rgb, tgt), 1e-4, 1e-4, "psnr")


def test_render_image_keeps_requested_keys_and_psnr():
    """evaluate.render_image: full-image chunked render returning only rgb/depth (+ on-demand entries) and the PSNR."""
    from brdf_nerf_amd.evaluate import batched_inference, render_image
    from brdf_nerf_amd import losses
    cfg = mini(**CONFIGS["rpv111_nlr"])
    model = build_model(cfg, 11)
    args = make_args(cfg)
    args.chunk = 24
    rays = torch.from_numpy(load_golden("render_lambert_test")["rays"]).to(DEV)
    tgt = torch.rand(rays.shape[0], 3, generator=torch.Generator().manual_seed(1)).to(DEV)
    torch.manual_seed(0)
    full = batched_inference({"coarse": model}, rays, None, args, apply_brdf=True, apply_theta=True, cos_irra_on=True)
    torch.manual_seed(0)
    img = render_image({"coarse": model}, args, rays, tgt, keys=("rgb", "depth", "normal_lr"), apply_brdf=True,
                       apply_theta=True, cos_irra_on=True)
    assert set(img) == {"rgb", "depth", "normal_lr", "psnr"}
    assert torch.equal(img["rgb"], full["rgb_coarse"]) and torch.equal(img["depth"], full["depth_coarse"])
    assert torch.equal(img["normal_lr"], full["normal_lr_coarse"])
    assert_close(img["psnr"], losses.psnr(full["rgb_coarse"], tgt), 1e-6, 1e-6, "psnr")


def test_train_loop_runs_saves_and_resumes(tmp_path):
    """TrainLoop (schedule + on-device ray table + fused step + checkpoints): the loss goes down on a small synthetic
    table, the BRDF stage switches on at its threshold, and a resumed loop continues where the saved one does."""
    import argparse
    from brdf_nerf_amd.raytable import synthetic_table
    from brdf_nerf_amd.train import TrainLoop
    cfg = mini(**CONFIGS["rpv111_nlr"])

    def fresh():
        a = make_args(cfg)
        for k, v in dict(batch_size=64, lr=5e-4, max_train_steps=40, brdf_on=0.25, cos_irra_on=0.25, nrrg_on=0.0, ds_drop=0.5,
                         ds_lambda=10.0, gsam_only_on=1.0, nr_reg_lr_lambda=0.01, hs_lambda=0.0, in_ckpts="none").items():
            setattr(a, k, v)
        torch.manual_seed(0)
        return TrainLoop(a, synthetic_table(640, device=DEV, seed=4), compute_dtype="fp32", near_far=(0.0, 2.0))

    loop = fresh()
    torch.manual_seed(1)
    hist = [loop.step() for _ in range(12)]
    assert not hist[9]["apply_brdf"] and hist[10]["apply_brdf"] and hist[10]["cos_irra_on"]
    assert hist[8]["epoch"] == 0 and hist[9]["epoch"] == 1 and hist[9]["lr"] == 5e-4 and abs(hist[10]["lr"] - 4.5e-4) < 1e-12
    assert float(hist[9]["loss"]) < float(hist[0]["loss"])
    path = loop.save(str(tmp_path / "ckpts"), str(tmp_path / "logs"))
    assert os.path.exists(tmp_path / "logs" / "opts.json")
    ck = torch.load(path, weights_only=False)
    assert all(k.startswith("nerf_coarse.") for k in ck["state_dict"]) and ck["global_step"] == 12
    torch.manual_seed(2)
    cont = [float(loop.step()["loss"]) for _ in range(3)]
    loop2 = fresh()
    loop2.resume(path)
    torch.manual_seed(2)
    cont2 = [float(loop2.step()["loss"]) for _ in range(3)]
    assert cont[0] == cont2[0]                                   # same state, same draws
    assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(cont, cont2))   # later steps: fp32-atomic summation order only


def test_train_loop_with_beta_keeps_the_embedding(tmp_path):
    """--beta through the harness (main.py:113-118): TrainLoop owns the image embedding, checkpoints carry it under the
    reference's key, a resumed loop gets it back, the fused step leaves the beta head and the table untouched (the loss of
    this model never reads beta_coarse, metrics.py:172-173), and render_rays on `loop.models` returns beta_coarse."""
    from brdf_nerf_amd import render_rays
    from brdf_nerf_amd.raytable import synthetic_table
    from brdf_nerf_amd.train import TrainLoop
    cfg = mini(beta=True, b=1, c=1, normal="learned")

    def fresh(seed):
        a = make_args(cfg)
        for k, v in dict(batch_size=64, lr=5e-4, max_train_steps=40, brdf_on=0.0, cos_irra_on=0.0, nrrg_on=0.0, ds_drop=0.5,
                         ds_lambda=10.0, gsam_only_on=1.0, nr_reg_lr_lambda=0.0, hs_lambda=0.0, in_ckpts="none",
                         t_embbeding_vocab=7).items():
            setattr(a, k, v)
        torch.manual_seed(seed)
        return TrainLoop(a, synthetic_table(640, device=DEV, seed=4), compute_dtype="fp32", near_far=(0.0, 2.0))

    # --beta with --lambda_rgb != 1: the reference scores epochs 0 and 1 with SNerfLoss(lambda_sc), whose lambda_rgb is
    # 1, and only from epoch 2 with SNerfLoss(lambda_rgb = args.lambda_rgb) (main.py:82-86, 237-238); 10 steps = one epoch here
    probe = fresh(0)
    probe.args.lambda_rgb = 0.5
    probe._lambda_rgb = 0.5
    seen = []
    for _ in range(22):
        out = probe.step()
        seen.append((out["epoch"], probe.trainer.lambda_rgb))
    assert all(lam == (1.0 if ep < 2 else 0.5) for ep, lam in seen) and {ep for ep, _ in seen} == {0, 1, 2}, seen
    loop = fresh(0)
    assert loop.embedding_t.weight.shape == (7, cfg.t_dim)
    emb0 = loop.embedding_t.weight.detach().clone()
    beta0 = {k: v.detach().clone() for k, v in loop.model.state_dict().items() if k.startswith("beta_from_xyz.")}
    losses_ = [float(loop.step()["loss"]) for _ in range(6)]
    assert losses_[-1] < losses_[0]
    # batches are gathered into fixed staging buffers: the production loop replays captured steps, not only bench.py
    assert len(loop.trainer._graphs) >= 1, "TrainLoop's steps were not captured into a HIP graph"
    assert torch.equal(loop.embedding_t.weight, emb0)
    assert all(torch.equal(loop.model.state_dict()[k], v) for k, v in beta0.items())
    path = loop.save(str(tmp_path / "ckpts"))
    ck = torch.load(path, weights_only=False)
    assert "embedding_t.weight" in ck["state_dict"] and "nerf_coarse.beta_from_xyz.0.weight" in ck["state_dict"]
    loop2 = fresh(5)                                     # other initial embedding
    assert not torch.equal(loop2.embedding_t.weight, emb0)
    loop2.resume(path)
    assert torch.equal(loop2.embedding_t.weight, emb0)
    rays = loop.table.data["rays"][:32].contiguous()
    ts = torch.arange(32, device=DEV) % 7
    with torch.no_grad():
        res, _ = render_rays(loop2.models, loop2.args, rays, ts, mode="test", apply_brdf=True, apply_theta=True, cos_irra_on=True)
    assert res["beta_coarse"].shape == (32, cfg.n_samples + cfg.guided_samples, 1) and bool((res["beta_coarse"] > 0).all())


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_train_loop_every_stage_in_half_modes(dtype):
    """TrainLoop through EVERY stage of the reference's schedule in a 16-bit mode: Lambertian start, BRDF heads on (brdf_on),
    cosine irradiance, Hapke theta head (2 x brdf_on), depth supervision dropped (ds_drop), guided-samples-only rendering with
    the sun-visibility pass (gsam_only_on, --sun_v analystic), regularisers.  In deterministic mode, twice: the two loss
    trajectories are identical to the bit; against the fp32 loop on the same batches and draws the first steps' losses agree to 2 %,
    the trajectory on average to 10 % (fp16 drifts like bf16: it is the training that amplifies, not the mantissa)."""
    import brdf_nerf_amd
    from brdf_nerf_amd.raytable import synthetic_table
    from brdf_nerf_amd.train import TrainLoop
    cfg = mini(b=1, c=1, theta=1, normal="learned", sun_v="analystic")

    def run(dt):
        a = make_args(cfg, dt)
        for k, v in dict(batch_size=64, lr=5e-4, max_train_steps=40, brdf_on=0.2, cos_irra_on=0.3, nrrg_on=0.1, ds_drop=0.6,
                         ds_lambda=10.0, gsam_only_on=0.15, nr_reg_lr_lambda=0.01, hs_lambda=0.05, in_ckpts="none").items():
            setattr(a, k, v)      # (gsam_only before the BRDF stage: with --sun_v analystic the reference needs it whenever the BRDF is on)
        torch.manual_seed(0)
        loop = TrainLoop(a, synthetic_table(640, device=DEV, seed=4), compute_dtype=dt, near_far=(0.0, 2.0))
        torch.manual_seed(1)
        hist = [loop.step() for _ in range(36)]
        return [float(h["loss"]) for h in hist], [(h["apply_brdf"], h["apply_theta"], h["cos_irra_on"], h["gsam_only"], h["depth_loss_on"]) for h in hist]

    prev = brdf_nerf_amd.set_deterministic(True)
    try:
        l32, f32 = run("fp32")
        l16a, f16 = run(dtype)
        l16b, _ = run(dtype)
    finally:
        brdf_nerf_amd.set_deterministic(prev)
    assert f32 == f16
    stages = set(f16)
    assert any(s[0] and s[1] and s[3] for s in stages) and any(not s[0] and not s[3] for s in stages) and any(not s[4] for s in stages), stages
    assert l16a == l16b, "the 16-bit loop is not reproducible in deterministic mode"
    assert all(np.isfinite(l16a))
    rels = [abs(a - b) / max(abs(b), 1e-6) for a, b in zip(l16a, l32)]
    diag(f"train loop through every stage, {dtype} vs fp32: relative loss difference per step " + " ".join(f"{r:.3f}" for r in rels))
    # the two runs drift apart as training goes (64-ray batches of an untrained BRDF model: one ray at a grazing angle moves a
    # step's loss by 10 %): early steps tightly, the whole trajectory on average
    assert max(rels[:4]) <= 0.02, rels[:4]
    # (measured: mean 0.03-0.11, summed losses within 0.01-0.06 - two chaotic trajectories, not a bias: the held-out PSNR gates
    # below are where the 16-bit modes are held to the fp32 result)
    assert sum(rels) / len(rels) <= 0.15 and abs(sum(l16a) - sum(l32)) <= 0.10 * sum(l32), rels


@pytest.mark.parametrize("name", ["lambert", "rpv111_nlr"])
def test_train_loop_trajectory_against_oracle(name):
    """TrainLoop (stage schedule + ray table + fused step + Adam + StepLR) for 8 optimisation steps against the CPU oracle
    driven by the same batches, the same stage flags and learning rates, and the same random draws: render_rays(mode
    'train') + SNerfLoss + DepthLoss + autograd + torch.optim.Adam.  Loss per step and the parameters at the end."""
    from brdf_nerf_amd.raytable import synthetic_table
    from brdf_nerf_amd.train import TrainLoop
    from oracle import losses as OL
    cfg = mini(**CONFIGS[name])
    a = make_args(cfg)
    for k, v in dict(batch_size=64, lr=5e-4, max_train_steps=8, brdf_on=0.5, cos_irra_on=0.5, nrrg_on=0.0, ds_drop=0.75,
                     ds_lambda=10.0, gsam_only_on=1.0, in_ckpts="none").items():
        setattr(a, k, v)
    torch.manual_seed(0)
    loop = TrainLoop(a, synthetic_table(192, device=DEV, seed=4), compute_dtype="fp32", near_far=None)
    R, S, G, K = 64, cfg.n_samples, cfg.guided_samples, 8
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in loop.model.state_dict().items()}
    opt = torch.optim.Adam(list(p.values()), lr=a.lr)
    batches = []
    nb = loop.table.next_batch
    loop.table.next_batch = lambda *x, **kw: (batches.append(nb(*x, **kw)) or batches[-1])
    gen = torch.Generator().manual_seed(9)
    got, want = [], []
    from brdf_nerf_amd import functions as Fn
    for i in range(K):
        # the launch-lean step draws in its kernels: the same Philox streams, as arrays, for the oracle (a step that leaves the
        # lean path would take them through torch.rand instead)
        st = loop.trainer.state
        uz, ug, ut = [Fn.rng_uniform(st, sid, R * n).view(R, n).cpu() for sid, n in ((1, S), (2, G), (3, G))]
        with Replay([uz, ug, ut]):
            out = loop.step()
        got.append(float(out["loss"]))
        b = {k: v.cpu() for k, v in batches[-1].items()}
        n_valid = int((b["valid_depth"] > 0).sum())
        flags = dict(apply_brdf=out["apply_brdf"], apply_theta=out["apply_theta"], cos_irra_on=out["cos_irra_on"],
                     gsam_only=out["gsam_only"])
        draws = [uz, torch.zeros(R, S), ug, ut[b["valid_depth"] > 0], torch.zeros(R, S + G)]     # (row r of the stream is ray r's; the randn draws act only through noise_std = 0)
        for grp in opt.param_groups:
            grp["lr"] = out["lr"]
        opt.zero_grad(set_to_none=True)
        res, _ = ORD.render_rays(p, cfg, b["rays"], ORD.Randoms(replay=draws), mode="train", valid_depth=b["valid_depth"],
                                 target_depths=b["depths"], target_std=b["depth_std"], **flags)
        loss = OL.snerf_loss(res, b["rgbs"])
        if out["depth_loss_on"]:
            loss = loss + OL.depth_loss(res, b["depths"][:, 0], b["depths"][:, 1], b["valid_depth"], b["depth_std"], a.ds_lambda)
        loss.backward()
        opt.step()
        want.append(float(loss.detach()))
    diag(f"train loop trajectory {name}: fused {' '.join(f'{x:.6f}' for x in got)} | oracle {' '.join(f'{x:.6f}' for x in want)}")
    assert any(h for h in [out["apply_brdf"]]) == (name != "lambert") or name == "lambert"
    for i, (x, y) in enumerate(zip(got, want)):
        assert abs(x - y) <= 1e-3 * abs(y) + 1e-6, f"step {i}: fused {x} oracle {y}"
    worst = max(float((v.detach().cpu() - p[k].detach()).abs().max()) for k, v in loop.model.state_dict().items())
    diag(f"train loop trajectory {name}: max |parameter difference| after {K} steps {worst:.3e}")
    assert worst <= 2e-4, worst


FULL_SIZE = {   # BASELINE.json configs 2-5 at their per-GPU shapes (F=512, 8 layers, PE10), in the dtype BASELINE.json names
    "c2_lambert": (dict(), 4096, 64, 64, dict(apply_brdf=False, apply_theta=False, cos_irra_on=False), "bf16"),
    "c3_rpv_nan": (dict(funcM=1, funcF=1, funcH=1, normal="analystic"), 4096, 64, 64, dict(apply_brdf=True, apply_theta=True, cos_irra_on=True), "bf16"),
    # config 4 = config 3's model (RPV + analytic normals, SURVEY 8d "C4 as C3") at 8192 rays x 128 samples over 8 GPUs: the
    # per-rank shape is 1024 rays x (128 + 64) samples; the learned-normal variant stays as a second shape check
    "c4_rpv_nan_s128": (dict(funcM=1, funcF=1, funcH=1, normal="analystic"), 1024, 128, 64, dict(apply_brdf=True, apply_theta=True, cos_irra_on=True), "bf16"),
    "c4_rpv_nlr_s128": (dict(funcM=1, funcF=1, funcH=1, normal="learned"), 1024, 128, 64, dict(apply_brdf=True, apply_theta=True, cos_irra_on=True), "bf16"),
    # config 5 (Hapke + microfacet, ds_lambda = 10, fp16): the reference's model classes are mutually exclusive (SURVEY
    # quirk 10), so "mix" = both kernels exercised, each in fp16 as BASELINE.json states; the bf16 runs stay beside them
    "c5_hapke_fp16": (dict(b=1, c=1, normal="analystic"), 1024, 64, 64, dict(apply_brdf=True, apply_theta=True, cos_irra_on=True), "fp16"),
    # Hapke with macroscopic roughness: its azimuth term has an infinite derivative at phi = 0; this seed reaches a ray
    # whose fp32 cos(phi) rounds to exactly 1 (the oracle's autograd is finite only because the CPU rounds it just below).
    # The fused step drops that ray's non-finite gradient (FusedTrainer.sanitize_grads).
    "c5_hapke_theta_fp16": (dict(b=1, c=1, theta=1, normal="analystic"), 1024, 64, 64, dict(apply_brdf=True, apply_theta=True, cos_irra_on=True), "fp16"),
    "c5_microfacet_fp16": (dict(roughness=True, normal="analystic"), 1024, 64, 64, dict(apply_brdf=True, apply_theta=True, cos_irra_on=True), "fp16"),
    "c5_hapke_theta_bf16": (dict(b=1, c=1, theta=1, normal="analystic"), 1024, 64, 64, dict(apply_brdf=True, apply_theta=True, cos_irra_on=True), "bf16"),
    "c5_microfacet_bf16": (dict(roughness=True, normal="analystic"), 1024, 64, 64, dict(apply_brdf=True, apply_theta=True, cos_irra_on=True), "bf16"),
    "c2_lambert_fp16": (dict(), 4096, 64, 64, dict(apply_brdf=False, apply_theta=False, cos_irra_on=False), "fp16"),
}


@pytest.mark.parametrize("name", list(FULL_SIZE))
def test_full_size_render_and_train_step_properties(name):
    """BASELINE configurations at full width and batch in their stated dtype, through size-independent properties: sorted
    depths, a permutation as sort index, weights in [0,1] summing to <= 1, pixels in [0,1], finite outputs and gradients,
    and a few fused training steps (depth supervision, ds_lambda=10) that stay finite and settle below the initial losses."""
    import bench
    from brdf_nerf_amd import render_rays
    from brdf_nerf_amd.trainer import FusedTrainer
    kw, R, S, G, flags, dtype = FULL_SIZE[name]
    cfg = FieldConfig(n_samples=S, guided_samples=G, **kw)
    args = make_args(cfg, dtype)
    torch.manual_seed(0)
    from brdf_nerf_amd import load_model
    model = load_model(args).to(DEV)
    b = bench.synthetic_batch(R, 5, torch.device(DEV))
    with torch.no_grad():
        res, _ = render_rays({"coarse": model}, args, b["rays"], None, mode="test", **flags)
    z, w, idx = res["z_vals_coarse"], res["weights_coarse"], res["sort_idx_coarse"]
    assert z.shape == (R, S + G) and bool((z[:, 1:] >= z[:, :-1]).all())
    assert bool((torch.sort(idx, -1)[0] == torch.arange(S + G, device=DEV)).all())
    assert float(w.min()) >= 0.0 and float(w.sum(-1).max()) <= 1.0 + 1e-4
    assert float(res["rgb_coarse"].min()) >= 0.0 and float(res["rgb_coarse"].max()) <= 1.0
    for k, v in res.items():
        if v.dtype.is_floating_point and k != "hpk_scl_coarse":
            assert bool(torch.isfinite(v).all()), k
    # ---- the training step at full size (round 3): (1) the 16-bit flat gradient against the fp32 HIP mode on the same batch and
    # the same in-kernel draws, per parameter matrix; (2) 24 steps on a LEARNABLE table (one consistent scene: colours and depth
    # priors of a height field) - the loss has to go down
    tb = _learnable_table(R, 21)
    lb = {k: tb.data[k] for k in ("rays", "rgbs", "valid_depth", "depths", "depth_std")}
    step_kw = dict(valid_depth=lb["valid_depth"], depths=lb["depths"], depth_std=lb["depth_std"], near_far=(0.0, 2.0), **flags)
    grads = {}
    t16 = torch.bfloat16 if dtype == "bf16" else torch.float16
    lambert = name.startswith("c2_")
    start = None
    import brdf_nerf_amd
    prev_det = brdf_nerf_amd.set_deterministic(True)     # reproducible sums: the pretrained state and the cosines below do not vary run to run
    if not lambert:
        # The BRDF models are compared at a TRAINED geometry: 150 Lambertian-stage steps on the learnable table first (the
        # reference's stage 1, README.md:100-116) - at a random initialisation the analytic normals of an untrained density and
        # the rays at grazing angles make the BRDF-stage gradient a near-cancelling sum that fp32 itself does not reproduce
        # under a 2^-9 perturbation of the weights
        torch.manual_seed(0)
        m0 = load_model(args).to(DEV)
        torch.manual_seed(5)
        t0 = FusedTrainer(m0, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
        pre_kw = dict(step_kw, apply_brdf=False, apply_theta=False, cos_irra_on=False)
        for _ in range(150):
            t0.step(lb["rays"], lb["rgbs"], **pre_kw)
        start = {k: v.detach().clone() for k, v in m0.state_dict().items()}
        del t0, m0
    seeds = {}
    for tag, dt, round_w in (("fp32", "fp32", False), (dtype, dtype, False), ("fp32_w16", "fp32", True), (dtype + "_seeded", dtype, False)):
        if lambert and tag not in ("fp32", dtype):
            continue
        torch.manual_seed(0)
        m = load_model(make_args(cfg, dt)).to(DEV)
        if start is not None:
            m.load_state_dict(start)
        if round_w:                                            # the referee: fp32 arithmetic on weights rounded to the 16-bit type
            with torch.no_grad():
                for p_ in m.parameters():
                    p_.copy_(p_.to(t16).float())
        torch.manual_seed(7)                                   # seeds the step state's draw key: the same draws in every run
        t = FusedTrainer(m, make_args(cfg, dt), lr=5e-4, ds_lambda=10.0, strict_rng=False)
        t.keep_grads = True
        if tag == "fp32":                                      # keep the guided depths and the gradient rows the fp32 field backward starts from ...
            t.seed_hook = lambda k_, d: seeds.update({k_: d.clone()})
        if tag.endswith("_seeded"):                            # ... and run the 16-bit field kernels at the SAME points from the SAME rows
            t.seed_hook = lambda k_, d: d.copy_(seeds[k_])
        t.step(lb["rays"], lb["rgbs"], **step_kw)
        assert bool(torch.isfinite(t.flat_grad).all()), f"{tag}: non-finite gradient"
        grads[tag] = {k: v.clone() for k, v in t.grad_views.items()}
        dropped = t.dropped_samples
        del t, m

    def worst_cos(a, b, matrices_only=False):
        worst, worst_k = 1.0, ""
        for k, g32 in grads[a].items():
            if g32.numel() < 256 or float(g32.abs().max()) == 0.0:
                continue
            if matrices_only and (g32.dim() < 2 or g32.numel() < 4096):
                continue
            c = float(torch.nn.functional.cosine_similarity(g32.flatten().double(), grads[b][k].flatten().double(), dim=0))
            if c < worst:
                worst, worst_k = c, k
        return worst, worst_k

    worst, worst_k = worst_cos("fp32", dtype)
    if lambert:
        # Lambertian model: every matrix of the 16-bit gradient points where the fp32 HIP mode's does
        diag(f"full size {name} ({dtype}): worst per-matrix gradient cosine vs the fp32 HIP mode {worst:.5f} ({worst_k}), dropped {dropped}")
        assert worst >= 0.99, (worst, worst_k)
    else:
        # BRDF on a RANDOM model: the loss gradient is a near-cancelling sum over rays at grazing angles / GGX peaks and over
        # normals of an untrained density - ill-conditioned in the WEIGHTS already.  The referee is the fp32 HIP mode evaluated on
        # weights rounded to the 16-bit type (fp32 arithmetic throughout): the 16-bit mode must track fp32 about as well as that
        # does - what is left when the conditioning of the problem is taken out
        ref, ref_k = worst_cos("fp32", "fp32_w16")
        diag(f"full size {name} ({dtype}): worst per-matrix gradient cosine vs the fp32 HIP mode {worst:.5f} ({worst_k}); the fp32 mode on "
             f"{dtype}-rounded weights: {ref:.5f} ({ref_k}); dropped {dropped}")
        # (measured, profiles/history/r03_parity_errors.txt: the referee itself scatters between -0.99 and 0.997 over these configurations -
        # a sign flip of the roughness head's bias gradient included - i.e. the BRDF-stage gradient at this state is not
        # reproducible to 2^-9 in the weights; where it is well conditioned, hapke + theta, the 16-bit modes reach 0.94-0.95)
        # -> the end-to-end cosines are REPORTED.  ASSERTED is what the 16-bit kernels are responsible for: every mode runs the
        # ray-level shading and losses in fp32, and the ill-conditioning sits there (the BRDF Jacobian amplifies the 1e-3 difference
        # of the composited sums) - so the 16-bit field backward (chain, analytic-normal double backward, weight gradients, at full
        # width and batch) is started from the SAME per-sample gradient rows as the fp32 mode's and must reproduce its gradient
        # (weight MATRICES: a bias vector's gradient is a signed sum over all points of the batch that nearly cancels - its cosine
        # scatters between 0.98 and 0.9999 from run to run in either 16-bit mode and says little; it is reported with the rest)
        seeded, seeded_k = worst_cos("fp32", dtype + "_seeded", matrices_only=True)
        seeded_all, seeded_all_k = worst_cos("fp32", dtype + "_seeded")
        flat = lambda tag_: torch.cat([v.flatten().double() for v in grads[tag_].values()])
        whole = float(torch.nn.functional.cosine_similarity(flat("fp32"), flat(dtype + "_seeded"), dim=0))
        # the END-TO-END (unseeded) whole flat gradient: its cosine weighs every matrix by the gradient mass it carries, so it says
        # whether the matrices with a poor cosine matter; the referee's beside it
        e2e = float(torch.nn.functional.cosine_similarity(flat("fp32"), flat(dtype), dim=0))
        e2e_ref = float(torch.nn.functional.cosine_similarity(flat("fp32"), flat("fp32_w16"), dim=0))
        mass = {k: float(v.double().norm() ** 2) for k, v in grads["fp32"].items()}
        tot = sum(mass.values())
        bad = [(k, float(torch.nn.functional.cosine_similarity(v.flatten().double(), grads[dtype][k].flatten().double(), dim=0)), mass[k] / tot)
               for k, v in grads["fp32"].items() if v.numel() >= 256 and float(v.abs().max()) > 0]
        bad.sort(key=lambda x: x[1])
        diag(f"full size {name} ({dtype}): END-TO-END whole flat gradient cosine vs the fp32 HIP mode {e2e:.5f} (referee {e2e_ref:.5f}); "
             f"the three worst matrices and their share of |g|^2: " + ", ".join(f"{k} {c:.3f} ({m:.1e})" for k, c, m in bad[:3]))
        diag(f"full size {name} ({dtype}): {dtype} field backward started from the fp32 mode's gradient rows - worst weight-matrix gradient "
             f"cosine {seeded:.5f} ({seeded_k}), worst of all parameters {seeded_all:.5f} ({seeded_all_k}), whole flat gradient {whole:.6f}")
        # Round 5: where the problem is well conditioned - the referee, fp32 arithmetic on 16-bit-rounded
        # weights, keeps a whole-gradient cosine >= 0.9 - the UNSEEDED 16-bit step must stay within 0.1 of it.  Round 4's fp16
        # steps with the 8-bit derivative stash did not (microfacet 0.077 / Hapke + theta 0.875 against 0.998 / 0.997: the analytic
        # normal's adjoint chain multiplies by D_l in every layer, profiles/r05_ablation.txt item 4); with the fp16 derivative stash
        # of the fp16 mode (DK16): Hapke 0.987 / Hapke + theta 0.9935 against 0.990 / 0.9995.  The margin is wide because the GGX
        # lobe's derivative w.r.t. the normal turns 0.1 degrees into several per cent of the cosine.
        if e2e_ref >= 0.9:
            assert e2e >= e2e_ref - 0.1, (name, e2e, e2e_ref)
        bf16_an = dtype == "bf16" and kw.get("normal") in ("analystic", "analystic_learned")
        assert whole >= (0.98 if bf16_an else 0.995), whole         # (bf16 + analytic normals: measured 0.993 .. 0.9992, see below)
        # (bf16 through the analytic-normal double backward: 8 significant bits over two chained passes, and the first layer's
        # gradient carries the w0^2 = 900 hand-over of that chain - measured 0.980 .. 0.999 for fc_net.0.weight between runs whose
        # 150 bf16 pretraining steps ended in different states (fp32 atomics); every other combination 0.997 .. 0.9997)
        floor = 0.96 if bf16_an else 0.99
        assert seeded >= floor, (seeded, seeded_k)
    brdf_nerf_amd.set_deterministic(prev_det)
    torch.manual_seed(3)
    tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
    losses_ = [float(tr.step(lb["rays"], lb["rgbs"], **step_kw)[0]) for _ in range(24)]
    assert all(l == l for l in losses_), losses_
    diag(f"full size {name} ({dtype}): learnable-table losses {losses_[0]:.4f} {losses_[1]:.4f} .. {losses_[-2]:.4f} {losses_[-1]:.4f}, "
         f"dropped samples {int(tr.dropped_samples)}")
    # (Adam's first steps move every weight by lr whatever the gradient: the first losses may spike; the last ones are below them)
    assert max(losses_[-3:]) < min(losses_[:3]), losses_


def _learnable_table(n_rays, seed):
    """Djibouti-shaped synthetic table of ONE consistent scene (something to learn, geometry included): a smooth height field
    z = h(x, y) seen from the table's near-nadir cameras.  Colours are a smooth function of the point the ray hits, the
    depth priors are the (first-order) hit depths with weight 1 on every ray - unlike raytable.synthetic_table's random
    colours and random depths, which supervise no geometry at all and leave the density field, and with it the analytic
    normals, to chance.  Training rays and held-out rays are different draws (seeds) of the same scene."""
    from brdf_nerf_amd.raytable import synthetic_table
    t = synthetic_table(n_rays, device=DEV, seed=seed)
    rays = t.data["rays"]
    o, d = rays[:, :3], rays[:, 3:6]
    height = lambda x, y: 0.12 * torch.sin(2.0 * x) * torch.cos(2.0 * y)
    depth = (o[:, 2] - height(o[:, 0], o[:, 1])) / (-d[:, 2])
    hit = o + d * depth.unsqueeze(-1)
    depth = depth + (hit[:, 2] - height(hit[:, 0], hit[:, 1])) / (-d[:, 2])          # one fixed-point refinement
    hit = o + d * depth.unsqueeze(-1)
    t.data["rgbs"] = torch.stack([0.5 + 0.4 * torch.sin(3 * hit[:, 0]), 0.5 + 0.4 * torch.cos(2 * hit[:, 1]),
                                  0.5 + 0.3 * torch.sin(2 * hit[:, 0] + hit[:, 1])], -1).contiguous()
    t.data["depths"] = torch.stack([depth, torch.ones_like(depth)], -1).contiguous()
    t.data["valid_depth"] = torch.ones_like(depth)
    t.data["depth_std"] = torch.zeros_like(depth)
    return t


RPV_NAN = dict(funcM=1, funcF=1, funcH=1, normal="analystic")


def _psnr_run(cfg, dtype, n_pre, n_brdf, train, held, draw_seed, init_state=None, lr0=5e-4, adam=None, keep_adam=None):
    """One training run of the PSNR gates: each stage (Lambertian pretraining, then the BRDF stage: the reference trains them
    as two runs, README.md:100-132) decays its learning rate from 5e-4 to 0 on a cosine, so the end state does not ride on
    the slope of a still-climbing curve; depth supervision during the pretraining only (--ds_drop, main.py:264).
    Returns (held-out PSNR, first-step training PSNR, final state_dict)."""
    import math
    from brdf_nerf_amd import load_model, losses
    from brdf_nerf_amd.evaluate import render_image
    from brdf_nerf_amd.trainer import FusedTrainer
    args = make_args(cfg, dtype)
    torch.manual_seed(0)
    model = load_model(args).to(DEV)
    if init_state is not None:
        model.load_state_dict(init_state)
    tr = FusedTrainer(model, args, lr=lr0, ds_lambda=10.0, strict_rng=False)
    tr.seed_draws(draw_seed)      # (the launch-lean step's in-kernel draws; torch.manual_seed below covers the general path)
    if adam is not None:          # continue a run: the optimiser's moments and step counts come along (same flat layout in every mode)
        tr.exp_avg.copy_(adam["exp_avg"]); tr.exp_avg_sq.copy_(adam["exp_avg_sq"]); tr.adam_steps.update(adam["adam_steps"])
    train.load_state_dict({"gen": torch.Generator(device=DEV).manual_seed(5).get_state(), "perm": None, "cursor": 0, "epoch": 0})
    torch.manual_seed(draw_seed)
    first = None
    for i in range(n_pre + n_brdf):
        on = i >= n_pre
        j, n = (i - n_pre, n_brdf) if on else (i, n_pre)
        tr.lr = lr0 * math.cos(0.5 * math.pi * j / n) ** 2
        b = train.next_batch(1024)
        loss, rgb = tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                            near_far=(0.0, 2.2), apply_brdf=on, apply_theta=on, cos_irra_on=on, depth_loss_on=not on)
        if i == 0:
            first = float(losses.psnr(rgb, b["rgbs"]))
    torch.manual_seed(2)
    on = n_brdf > 0
    res = render_image({"coarse": model}, args, held.data["rays"], held.data["rgbs"], keys=("rgb",), chunk=2048,
                       apply_brdf=on, apply_theta=on, cos_irra_on=on)
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters()), dtype
    if keep_adam is not None:
        keep_adam.update(exp_avg=tr.exp_avg.clone(), exp_avg_sq=tr.exp_avg_sq.clone(), adam_steps=dict(tr.adam_steps))
    return float(res["psnr"]), first, {k: v.detach().clone() for k, v in model.state_dict().items()}


def test_reduced_precision_heldout_psnr_tracks_fp32_lambert():
    """north_star: PSNR within 0.05 dB of the reference.  The fp32 mode is held to the reference by the golden tests; here
    the bf16 and fp16 throughput modes are trained beside it on a LEARNABLE scene (BASELINE config 2's model: Lambertian
    pretraining) - same initialisation, same batches, same random draws, 400 fused steps of 1024 rays x (64 + 64) samples
    at F = 512 - and the PSNR of 8192 HELD-OUT rays (never trained on) must agree with fp32's within 0.05 dB.  Training
    amplifies rounding differences (the fp32 atomics of the 16-bit pipelines are order dependent: the same binary differs
    from itself run to run), so every mode is run BN_PSNR_REPEATS times (default 3, different sampling draws) and the
    MEANS are compared; the runs are reported."""
    cfg = FieldConfig(n_samples=64, guided_samples=64)
    train, held = _learnable_table(1024 * 64, 3), _learnable_table(8192, 11)
    reps = int(os.environ.get("BN_PSNR_REPEATS", "3"))
    psnr, first = {}, None
    for dtype in ("fp32", "bf16", "fp16"):
        psnr[dtype] = []
        for r in range(reps):
            p, f, _ = _psnr_run(cfg, dtype, 400, 0, train, held, draw_seed=1 + r)
            psnr[dtype].append(p)
            first = f if first is None else first
    mean = {k: sum(v) / len(v) for k, v in psnr.items()}
    diag(f"held-out PSNR lambert after 400 steps (first-step train PSNR {first:.2f} dB), {reps} runs per mode: "
         + ", ".join(f"{k} {mean[k]:.4f} dB (runs {' '.join(f'{x:.4f}' for x in v)})" for k, v in psnr.items())
         + f"; |bf16-fp32| {abs(mean['bf16'] - mean['fp32']):.4f}, |fp16-fp32| {abs(mean['fp16'] - mean['fp32']):.4f}")
    assert mean["fp32"] > first + 3.0, (psnr, first)          # the scene was learned, the gate is not vacuous
    assert abs(mean["bf16"] - mean["fp32"]) <= 0.05, psnr
    assert abs(mean["fp16"] - mean["fp32"]) <= 0.05, psnr


def test_reduced_precision_heldout_psnr_tracks_fp32_rpv_analytic_normals():
    """The same gate for BASELINE config 3's model (RPV funcM/F/H = 1 + analytic normals, double backward active).  All modes
    start from ONE Lambertian pretraining (fp32, 400 steps), then the BRDF stage runs in fp32 / bf16 / fp16 with identical
    batches and draws.  Two measurements:
      (a) the BRDF stage is first trained in fp32 (600 steps), then CONTINUED for 150 steps (lr 1e-4 -> 0) in every mode from
          that shared model AND its optimiser state (Adam moments and step counts carried over), BN_PSNR_REPEATS draw seeds
          per mode, differences paired by seed.  This isolates what the arithmetic does to training: trajectories that start
          together stay together (measured over four starting states: |difference| <= 0.018 dB, profiles/
          r02_psnr_state_study.txt) - gated at the north_star's 0.05 dB on the mean paired difference.  (With a FRESH Adam
          state the first updates are +-lr per parameter whatever the gradient: fp32 itself then loses 0.2-0.3 dB and the modes
          scatter by +-0.1 dB with either sign - a property of the restart, not of the arithmetic; same study.)
      (b) 600 BRDF steps from the warm start, paired by draw seed: the stage restarts three heads from their initialisation
          with a fresh optimiser state; its first 50 steps throw the held-out PSNR anywhere between 16.6 and 19.1 dB in EVERY
          mode and the end states scatter by +-0.15 dB with either sign (fp32 against itself included:
          profiles/history/r02_psnr_transient_study.txt).  Three seeds cannot resolve 0.05 dB: the default run only catches a broken
          mode (every mode learned the scene, paired means within 0.5 dB); the statistical statement - mean paired difference
          and its 95 % interval over >= 64 seeds in deterministic mode - is profiles/psnr_paired_study.py's, kept in
          profiles/history/r03_psnr_paired_rpv.txt, and this test applies the same gate when run with BN_PSNR_PAIRED_SEEDS >= 16."""
    import statistics
    cfg = FieldConfig(n_samples=64, guided_samples=64, **RPV_NAN)
    train, held = _learnable_table(1024 * 64, 3), _learnable_table(8192, 11)
    _, first, warm = _psnr_run(cfg, "fp32", 400, 0, train, held, draw_seed=1)
    adam = {}
    p_trained, _, trained = _psnr_run(cfg, "fp32", 0, 600, train, held, draw_seed=3, init_state=warm, keep_adam=adam)
    reps = int(os.environ.get("BN_PSNR_REPEATS", "3"))
    short = {dtype: [_psnr_run(cfg, dtype, 0, 150, train, held, draw_seed=7 + r, init_state=trained, lr0=1e-4, adam=adam)[0]
                     for r in range(reps)] for dtype in ("fp32", "bf16", "fp16")}
    smean = {k: sum(v) / len(v) for k, v in short.items()}
    pair = {k: [a - b for a, b in zip(short[k], short["fp32"])] for k in ("bf16", "fp16")}
    again = _psnr_run(cfg, "fp32", 0, 150, train, held, draw_seed=7, init_state=trained, lr0=1e-4, adam=adam)[0]
    diag(f"held-out PSNR rpv_nan, 150 more BRDF steps (lr 1e-4 -> 0, Adam state carried) from a shared fp32 model ({p_trained:.4f} dB "
         f"after 400 + 600 steps), {reps} draw seeds: "
         + ", ".join(f"{k} {smean[k]:.4f} dB (runs {' '.join(f'{x:.4f}' for x in v)})" for k, v in short.items())
         + "; paired differences to fp32: "
         + ", ".join(f"{k} {sum(v) / reps:+.4f} dB (per seed {' '.join(f'{x:+.4f}' for x in v)})" for k, v in pair.items())
         + f"; fp32 repeated on the first seed's draws: {again - short['fp32'][0]:+.4f} dB")
    for k, v in pair.items():
        assert abs(sum(v) / reps) <= 0.05, (k, short)
    # (b) paired by draw seed (round 3): per seed the three modes see the same batches and the same in-kernel draws; with
    # BN_PSNR_PAIRED_SEEDS >= 16 (profiles/psnr_paired_study.py runs the same protocol in deterministic mode and keeps its table
    # in profiles/history/r03_psnr_paired_rpv.txt: the paired differences have a standard deviation of ~0.19 dB, so the 95 % interval of
    # their mean closes to +-0.05 dB at ~64 seeds) the mean paired difference and its interval are gated; the default three seeds
    # only catch a broken mode
    from scipy import stats
    reps = int(os.environ.get("BN_PSNR_PAIRED_SEEDS", os.environ.get("BN_PSNR_REPEATS", "3")))
    n_long = int(os.environ.get("BN_PSNR_BRDF_STEPS", "600"))
    long_ = {dtype: [] for dtype in ("fp32", "bf16", "fp16")}
    for r in range(reps):
        for dtype in long_:
            long_[dtype].append(_psnr_run(cfg, dtype, 0, n_long, train, held, draw_seed=11 + r, init_state=warm)[0])
    mean = {k: sum(v) / len(v) for k, v in long_.items()}
    msg = []
    for k in ("bf16", "fp16"):
        d = [a - b for a, b in zip(long_[k], long_["fp32"])]
        m, sd = statistics.mean(d), statistics.stdev(d)
        hw = float(stats.t.ppf(0.975, reps - 1)) * sd / reps ** 0.5
        msg.append(f"{k} - fp32 paired: mean {m:+.4f} dB, sd {sd:.4f}, 95 % half-width {hw:.4f}")
        if reps >= 16:
            assert abs(m) <= 0.05, (k, m, hw, long_)
            assert hw <= 0.05 or reps < 64, (k, m, hw, "more seeds needed for a +-0.05 dB interval")
        assert abs(m) <= 0.5, (k, long_)                          # gross-error gate of the short default run
    diag(f"held-out PSNR rpv_nan, {n_long} BRDF steps, {reps} draw seeds per mode: "
         + ", ".join(f"{k} {mean[k]:.4f} dB (runs {' '.join(f'{x:.4f}' for x in v)})" for k, v in long_.items()) + "; " + "; ".join(msg))
    assert all(m > first + 3.0 for m in mean.values()), (long_, first)


C5_MODELS = {    # BASELINE config 5's BRDF models (FULL_SIZE above), analytic normals as in the configuration
    "hapke_bc": dict(b=1, c=1, normal="analystic"),
    "hapke_bct": dict(b=1, c=1, theta=1, normal="analystic"),
    "microfacet": dict(roughness=True, normal="analystic"),
}


@pytest.mark.parametrize("name", list(C5_MODELS))
def test_reduced_precision_heldout_psnr_tracks_fp32_config5(name):
    """the PSNR gate of config 3 for BASELINE config 5's models (Hapke (b, c), Hapke (b, c, theta), microfacet;
    fp16 is the dtype BASELINE.json names, bf16 runs beside it): gate (a) of
    test_reduced_precision_heldout_psnr_tracks_fp32_rpv_analytic_normals - the BRDF stage trained in fp32 (400 Lambertian + 800
    BRDF steps), then CONTINUED for 150 steps (lr 1e-4 -> 0, Adam state carried) in every mode, BN_PSNR_REPEATS (default 2) draw seeds,
    paired by seed: the mean paired difference within the north_star's 0.05 dB.
    No restart gate (b) here: on this synthetic scene a Hapke stage restarted with fresh heads ends anywhere between 8 and 13 dB
    in fp32 ITSELF depending on the draw seed (profiles/r05_c5_gate_stage*.txt: the scene's colours are not a Hapke surface's; the
    stage is still climbing after 800 steps and (b, c, theta) does not get off the ground) - a paired difference of restarts says
    nothing about the arithmetic there.  Where the curve still climbs the continuation difference is proportional to its slope
    (Hapke (b, c): -0.058 / -0.043 dB for bf16 / fp16 after a 400-step stage, -0.024 / -0.011 after 800: same file), hence the
    800 steps.  The statistical statement over 64 paired seeds is profiles/psnr_paired_study.py --protocol=continue
    --config=<name>, kept in profiles/r05_psnr_paired_<name>.txt (and the restart protocol beside it for microfacet)."""
    cfg = FieldConfig(n_samples=64, guided_samples=64, **C5_MODELS[name])
    train, held = _learnable_table(1024 * 64, 3), _learnable_table(8192, 11)
    _, first, warm = _psnr_run(cfg, "fp32", 400, 0, train, held, draw_seed=1)
    adam = {}
    n_stage = int(os.environ.get("BN_C5_BRDF_STEPS", "800"))
    p_trained, _, trained = _psnr_run(cfg, "fp32", 0, n_stage, train, held, draw_seed=3, init_state=warm, keep_adam=adam)
    reps = int(os.environ.get("BN_PSNR_REPEATS", "2"))      # (two seeds: the three models add ~2 minutes to the suite as it is)
    short = {dtype: [_psnr_run(cfg, dtype, 0, 150, train, held, draw_seed=7 + r, init_state=trained, lr0=1e-4, adam=adam)[0]
                     for r in range(reps)] for dtype in ("fp32", "bf16", "fp16")}
    pair = {k: [a - b for a, b in zip(short[k], short["fp32"])] for k in ("bf16", "fp16")}
    diag(f"held-out PSNR config 5 {name}, 150 more BRDF steps (lr 1e-4 -> 0, Adam state carried) from a shared fp32 model "
         f"({p_trained:.4f} dB after 400 + {n_stage} steps; first-step training PSNR {first:.2f} dB), {reps} draw seeds: "
         + ", ".join(f"{k} {sum(v) / reps:.4f} dB" for k, v in short.items()) + "; paired differences to fp32: "
         + ", ".join(f"{k} {sum(v) / reps:+.4f} dB (per seed {' '.join(f'{x:+.4f}' for x in v)})" for k, v in pair.items()))
    for k, v in pair.items():
        assert abs(sum(v) / reps) <= 0.05, (name, k, short)
    assert all(x == x for v in short.values() for x in v), short


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])


@pytest.mark.parametrize("name,lean", [("lambert", "1"), ("rpv_nan", "1"), ("lambert", "0")])
def test_two_rank_step_matches_one_rank(name, lean):
    """ray-batch data parallelism.  Two ranks (both on cuda:0, gloo - the one-GPU box has no second device for
    RCCL) each run FusedTrainer.step on half of a batch; the all-reduced flat gradient / 2 and the parameters after Adam
    must equal a one-rank step on the concatenated batch (tests/dist_step_worker.py).  lean = 1: the launch-lean step, whose
    in-kernel draws are indexed by the GLOBAL ray (FusedTrainer.ray_offset); lean = 0: the general step fed with slices of the
    whole batch's draws."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_step_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="2", BN_DIST_CONFIG=name, BN_DIST_LEAN=lean,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, worker], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT")
    for o in outs:
        for line in o.splitlines():
            if line.startswith("RESULT"):
                diag(line)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_rccl_initialises_and_reduces_the_flat_gradient():
    """RCCL through this code on the one-GPU box: a world of ONE rank on backend nccl (RCCL refuses two ranks on one device)
    runs the data-parallel FusedTrainer.step and `allreduce_sum_` on the flat gradient buffer; same worker, same checks as
    the two-rank gloo case (the N > 1 RCCL runs are the driver's, bench.py --gpus N)."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_step_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(), WORLD_SIZE="1", RANK="0", BN_DIST_CONFIG="lambert",
               BN_DIST_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    for line in p.stdout.splitlines():
        if line.startswith("RESULT"):
            diag("rccl world 1: " + line)
    assert p.returncode == 0, p.stdout


def test_ray_table_on_device_feeds_the_fused_step():
    """the ray table lives in HBM, batches are gathers by a device permutation (no host round trip):
    every row exactly once per epoch, rank shards partition each global batch, the generator state resumes the stream, and
    a batch goes straight into the fused step."""
    from brdf_nerf_amd import load_model
    from brdf_nerf_amd.raytable import synthetic_table
    from brdf_nerf_amd.trainer import FusedTrainer
    t = synthetic_table(1000, device=DEV, seed=3)
    assert all(v.is_cuda for v in t.data.values()) and t.data["rays"].shape == (1000, 11)
    tag = torch.arange(1000.0, device=DEV)
    t.data["rgbs"][:, 0] = tag
    seen = torch.cat([t.next_batch(96)["rgbs"][:, 0] for _ in range(11)])          # 10 x 96 + 40
    assert seen.numel() == 1000 and bool((torch.sort(seen)[0] == tag).all()) and t.epoch == 0
    state = t.state_dict()
    nxt = t.next_batch(96)["rgbs"][:, 0].clone()
    assert t.epoch == 1
    shards = [synthetic_table(1000, device=DEV, seed=3) for _ in range(3)]
    for s_ in shards:
        s_.data["rgbs"][:, 0] = tag
        s_.load_state_dict(state)
    parts = [s_.next_batch(96, rank=r, world=3)["rgbs"][:, 0] for r, s_ in enumerate(shards)]
    assert torch.equal(torch.cat(parts), nxt)                                         # shards partition the global batch, in order
    cfg = mini()
    args = make_args(cfg)
    torch.manual_seed(0)
    tr = FusedTrainer(load_model(args).to(DEV), args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
    t.data["rgbs"][:, 0] = 0.5
    b = t.next_batch(64)
    loss, rgb = tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"])
    assert rgb.shape == (64, 3) and bool(torch.isfinite(loss)) and bool(torch.isfinite(tr.flat_grad).all())


@pytest.mark.parametrize("name,dtype,feat,R", [("lambert", "bf16", 512, 1024), ("rpv_nan", "bf16", 512, 512), ("rpv_nan", "fp16", 256, 512),
                                               ("hapke_nlr", "fp32", 64, 256)])
def test_gradients_are_bitwise_reproducible_by_default(name, dtype, feat, R):
    """The reference trains with deterministic=True (main.py:726).  Since round 4 the weight-gradient kernels write per-split
    slabs that a reduce kernel adds in fixed order (field_wgrad.hip: no atomics, no turn counters), so WITHOUT any switch two
    fused training steps from the same state, batch and draws produce BITWISE identical flat gradients and parameters, and
    set_deterministic(True) changes neither.  Covers many point splits per tile (F=512), the chained analytic-normal jobs (two
    jobs into one matrix), the native and row-major skinny jobs and the fp32 kernels.  Then the same for REPLAYED steps: eight
    steps (the last four replayed from the captured HIP graph) twice from the same state - identical parameters, bit for bit."""
    import brdf_nerf_amd
    from brdf_nerf_amd import _lib
    from brdf_nerf_amd.trainer import FusedTrainer
    import ctypes as C
    kw = {"lambert": dict(), "rpv_nan": dict(funcM=1, funcF=1, funcH=1, normal="analystic"),
          "hapke_nlr": dict(b=1, c=1, theta=1, normal="learned")}[name]
    cfg = FieldConfig(feat=feat, n_samples=32, guided_samples=32, **kw)
    args = make_args(cfg, dtype)
    flags = dict() if name == "lambert" else dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)
    import bench
    b = bench.synthetic_batch(R, 5, torch.device(DEV))
    rays, rgbs = b["rays"], b["rgbs"]

    losses = []          # per run: the step losses as the trainer reports them

    def run(det, steps=1, graph=False):
        prev = brdf_nerf_amd.set_deterministic(det)
        try:
            torch.manual_seed(3)
            model = build_model(cfg, 11, dtype)
            tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
            tr.use_graph, tr.keep_grads = graph, steps == 1
            losses.append([])
            for _ in range(steps):
                l_, _ = tr.step(rays, rgbs, valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"], near_far=(0.0, 2.0),
                                **flags)
                losses[-1].append(l_.clone())          # (a replayed graph rewrites its output tensor: keep the value)
            torch.cuda.synchronize()
            return tr.flat_grad.clone(), tr.flat_param.clone(), len(tr._graphs)
        finally:
            brdf_nerf_amd.set_deterministic(prev)

    g1, p1, _ = run(False)
    g2, p2, _ = run(False)
    assert float(g1.abs().max()) > 0
    assert torch.equal(g1, g2) and torch.equal(p1, p2), f"max diff {float((g1 - g2).abs().max()):.3e}"
    g3, p3, _ = run(True)
    assert torch.equal(g1, g3) and torch.equal(p1, p3), "set_deterministic changed the gradients"
    _, q1, n1 = run(False, steps=8, graph=True)
    _, q2, n2 = run(False, steps=8, graph=True)
    assert n1 >= 1 and n2 >= 1, "the step was not captured"
    assert torch.equal(q1, q2), f"replayed steps differ: max {float((q1 - q2).abs().max()):.3e}"
    _, q3, _ = run(False, steps=8, graph=False)
    assert torch.equal(q1, q3), f"replayed and eager steps differ: max {float((q1 - q3).abs().max()):.3e}"
    # round 5: the LOGGED losses repeat bit for bit too (a fixed-order sum of the per-ray terms, FusedTrainer.repeatable_loss;
    # rounds 3-4 added them up with float atomics outside the deterministic mode) - run to run, eager or replayed
    as_bits = lambda ls: [float(x) for x in ls]
    assert as_bits(losses[0]) == as_bits(losses[1]) == as_bits(losses[2]), "the logged loss of one step differs between runs"
    assert as_bits(losses[3]) == as_bits(losses[4]) == as_bits(losses[5]), "the logged losses of eight steps differ between runs"
    faults = C.c_uint(0)
    _lib.check(_lib.lib().bn_device_faults(C.byref(faults), None), "bn_device_faults")
    assert faults.value == 0


def test_c_abi_from_a_plain_host_program(tmp_path):
    """The drop-in boundary is a C ABI: examples/abi_smoke.cpp - no Python, no torch - links the library, allocates with the
    HIP runtime, calls bn_stratified_z / bn_composite_forward on its own stream and checks them against a scalar restatement
    of get_z_vals / cal_weight; a bad argument must come back as a status with a message."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(root, "brdf_nerf_amd")
    subprocess.run([hipcc, "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "abi_smoke.cpp"), "-o", exe,
                    "-L" + libdir, "-lbrdfnerf_hip", "-Wl,-rpath," + libdir], check=True, timeout=600)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    for line in p.stdout.splitlines():
        diag("abi_smoke: " + line)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout


def test_d8_derivative_stash_round_trip(tmp_path):
    """The 8-bit derivative stash of the 16-bit modes (csrc/field_kernels.h d8_pack4 / d8_unpack4: the activation derivative the
    backward chains of spsbrdfnerf.py:636-646 multiply by) encoded and decoded on the device: a Siren layer's cosine comes back
    within half a step (128.5 / 32767) without bias, a ReLU mask exactly, and a NaN or out-of-range value stays inside its own
    byte (tests/d8_roundtrip.hip)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "d8_roundtrip")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result", "-Wno-pass-failed",
                    "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "brdf_nerf_amd", "csrc"),
                    os.path.join(root, "tests", "d8_roundtrip.hip"), "-o", exe], check=True, timeout=600)
    p = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    for line in p.stdout.splitlines():
        diag("d8_roundtrip: " + line)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout


def test_count_nonfinite_hook():
    """Sync-free replacement of check_nan (train_utils.py:14-25): NaN and Inf counters accumulate on the device."""
    from brdf_nerf_amd import functions as Fn
    x = torch.randn(100003, device=DEV)
    x[5] = float("nan"); x[77777] = float("nan"); x[9] = float("inf"); x[100002] = -float("inf"); x[50000] = float("inf")
    c = Fn.count_nonfinite(x)
    c = Fn.count_nonfinite(torch.ones(7, device=DEV), c)
    assert c.tolist() == [2, 3]
    assert Fn.count_nonfinite(torch.zeros(0, device=DEV)).tolist() == [0, 0]


def test_bad_arguments_come_back_as_errors_not_faults():
    """The C ABI never launches on arguments it cannot serve (SURVEY.md section 8b error convention): empty inputs, sample
    counts beyond the kernels' limits and missing buffers return a status with a message; the largest supported ray
    (S = 512 samples) still composites correctly."""
    from brdf_nerf_amd import functions as Fn
    near, far = torch.zeros(0, 1, device=DEV), torch.ones(0, 1, device=DEV)
    with pytest.raises(RuntimeError, match="stratified_z"):
        Fn.stratified_z(near, far, torch.rand(0, 16, device=DEV))                      # no rays
    with pytest.raises(RuntimeError, match="stratified_z"):
        Fn.stratified_z(torch.zeros(4, 1, device=DEV), torch.ones(4, 1, device=DEV), torch.rand(4, 1, device=DEV))   # S = 1
    z = torch.sort(torch.rand(3, 513, device=DEV), -1)[0]
    with pytest.raises(RuntimeError, match="composite"):
        Fn.composite_forward_raw(z, torch.rand(3, 513, 4, device=DEV))                # S beyond 512
    with pytest.raises(RuntimeError, match="composite"):
        Fn.composite_forward_raw(z[:, :8].contiguous(), torch.rand(3, 8, 33, device=DEV))   # more channels than supported
    z16 = torch.sort(torch.rand(5, 16, device=DEV), -1)[0]
    w = torch.rand(5, 16, device=DEV)
    with pytest.raises(RuntimeError, match="guided_samples"):
        Fn.guided_samples(z16, w, (w * z16).sum(-1), torch.rand(5, 2, device=DEV), 0.0, 1.0, 3.0)      # G < 3
    with pytest.raises(RuntimeError, match="guided_samples"):
        Fn.guided_samples(z16, w, (w * z16).sum(-1), torch.rand(5, 300, device=DEV), 0.0, 1.0, 3.0)    # G beyond 256
    cfg = mini()
    model = build_model(cfg, 3)
    spec = model.spec(False, False, False)
    with pytest.raises(RuntimeError, match="field"):
        Fn.field_sigma(spec, model.named(), model.repack(spec), xyz=torch.zeros(0, 3, device=DEV))     # no points
    # the largest ray the compositing kernels take
    g = torch.Generator().manual_seed(5)
    zc = torch.sort(torch.rand(6, 512, generator=g) * 2, -1)[0]
    out = torch.randn(6, 512, 4, generator=g)
    a, T, wr, d = ORD.composite(zc, out[..., 3], None, 0.0)
    _, _, w2, d2, acc2 = Fn.composite_forward_raw(zc.to(DEV), out.to(DEV))
    assert_close(w2, wr, 1e-5, 1e-7, "w S=512")
    assert_close(d2, d, 1e-5, 1e-6, "depth S=512")
    assert_close(acc2, (wr.unsqueeze(-1) * out).sum(-2), 1e-4, 1e-5, "acc S=512")
