"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, tparams, replay_list, assert_close
from oracle.config import FieldConfig
from oracle import field as F, render as RD, brdf as B, losses as L

CONFIGS = {
    "lambert": dict(),
    "rpv111_nan": dict(funcM=1, funcF=1, funcH=1, normal="analystic"),
    "rpv111_nlr": dict(funcM=1, funcF=1, funcH=1, normal="learned"),
    "hapke_bc": dict(b=1, c=1, normal="analystic"),
    "hapke_bct": dict(b=1, c=1, theta=1, normal="learned"),
    "microfacet": dict(roughness=True, normal="learned"),
}


def mini(**kw):
    base = dict(feat=64, n_samples=16, guided_samples=16)
    base.update(kw)
    return FieldConfig(**base)


def checksum(cfg, seed):
    return float(sum(v.astype(np.float64).sum() for v in cfg.make_params(seed).values()))


@pytest.mark.parametrize("name", list(CONFIGS))
def test_field_forward_matches_reference(name):
    g = load_golden(f"field_{name}_F64")
    cfg = mini(**CONFIGS[name])
    assert abs(checksum(cfg, 11) - float(g["param_checksum"])) < 1e-9
    p = tparams(cfg, 11)
    xyz = torch.from_numpy(g["xyz"])
    an = cfg.normal in ("analystic", "analystic_learned")
    lr = cfg.normal in ("learned", "analystic_learned")
    out = F.field_forward(p, cfg, xyz, apply_brdf=True, apply_theta=True, nr_an_on=an, nr_lr_on=lr)
    assert out.shape[1] == cfg.out_channels(True, True)
    assert_close(out, g["out_brdf"], 1e-5, 1e-6, "out_brdf")
    out0 = F.field_forward(p, cfg, xyz, apply_brdf=False, nr_an_on=an, nr_lr_on=lr)
    assert_close(out0, g["out_nobrdf"], 1e-5, 1e-6, "out_nobrdf")
    assert_close(F.field_forward(p, cfg, xyz, sigma_only=True), g["sigma"], 1e-5, 1e-6, "sigma")


def test_field_forward_F512():
    g = load_golden("field_rpv111_nan_F512")
    cfg = FieldConfig(**CONFIGS["rpv111_nan"])
    assert cfg.n_params() == 2690567          # SURVEY.md section 8 row a4 [probed]
    assert FieldConfig().n_params() == 2295812
    p = tparams(cfg, 12)
    out = F.field_forward(p, cfg, torch.from_numpy(g["xyz"]), apply_brdf=True, nr_an_on=True)
    assert out.shape[1] == 16
    assert_close(out, g["out_brdf"], 2e-4, 2e-5, "F512 out")


def test_sigma_grad_closed_form_fp64():
    g = load_golden("field_sigma_grad_F64_fp64")
    cfg = mini(**CONFIGS["rpv111_nan"])
    p = tparams(cfg, 11, torch.float64)
    x = torch.from_numpy(g["xyz"])
    assert_close(F.sigma_grad(p, cfg, x, create_graph=False), g["grad"], 1e-10, 1e-12, "autograd")
    assert_close(F.sigma_grad_closed_form(p, cfg, x), g["grad"], 1e-10, 1e-12, "closed form")


@pytest.mark.parametrize("S", [16, 128])
def test_composite(S):
    g = load_golden(f"composite_S{S}")
    z = torch.from_numpy(g["z"])
    sigma = torch.from_numpy(g["sigma"]).requires_grad_(True)
    a, T, w, d = RD.composite(z, sigma)
    for got, key in ((a, "alphas"), (T, "transparency"), (w, "weights"), (d, "depth")):
        assert np.array_equal(got.detach().numpy(), g[key]), key        # identical op sequence -> bit-exact
    ((w * torch.from_numpy(g["cw"])).sum() + (d * torch.from_numpy(g["cd"])).sum()).backward()
    assert_close(sigma.grad, g["dsigma"], 1e-6, 1e-9, "dsigma")


@pytest.mark.parametrize("mode", ["test", "train"])
def test_guided_samples(mode):
    g = load_golden(f"guided_{mode}")
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    rnd = RD.Randoms(replay=replay_list(g))
    kw = {}
    if mode == "train":
        kw = dict(valid_depth=t["valid_depth"], target_depths=t["target_depths"], target_std=t["target_std"])
    near0, far0 = torch.tensor(0.0), torch.tensor(2.0)
    z2, inds, inds_gt = RD.guided_samples(t["depth"], t["weights"], t["z"], 64, near0, far0, rnd, 3.0, mode, **kw)
    assert np.array_equal(z2.numpy(), g["z2"])
    z2s = torch.sort(z2, -1)[0]
    z_all, idx = torch.sort(torch.cat([t["z"], z2s], -1), -1)
    assert np.array_equal(z_all.numpy(), g["z_all"])
    assert np.array_equal(idx.numpy(), g["sort_idx"])
    assert inds.dtype == torch.int64 and int(inds.min()) >= 1 and int(inds.max()) <= 63


def test_brdf_rpv():
    g = load_golden("brdf_rpv")
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    n, w, k, th, rc = [t[x].clone().requires_grad_(True) for x in ("n", "w", "k", "theta", "rhoc")]
    brdf, M1, G, H, ci, cv = B.rpv(t["l"], t["v"], n, w, k, th, rc)
    for got, key in ((brdf, "brdf"), (M1, "M1"), (G, "G"), (H, "H"), (ci, "ci"), (cv, "cv")):
        assert_close(got, g[key], 1e-6, 1e-7, key)
    (brdf * t["coef"]).sum().backward()
    for got, key in ((n, "dn"), (w, "dw"), (k, "dk"), (th, "dtheta"), (rc, "drhoc")):
        assert_close(got.grad, g[key], 1e-5, 1e-6, key)


@pytest.mark.parametrize("tag,use_c,use_t,shell", [("hapke_b", 0, 0, 0), ("hapke_bc", 1, 0, 0), ("hapke_bct", 1, 1, 0),
                                                   ("hapke_shell1", 0, 0, 1), ("hapke_shell2", 0, 0, 2),
                                                   ("hapke_shell3", 0, 0, 3)])
def test_brdf_hapke(tag, use_c, use_t, shell):
    g = load_golden(f"brdf_{tag}")
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    n, w, b, c, th = [t[x].clone().requires_grad_(True) for x in ("n", "w", "b", "c", "theta")]
    o = B.hapke(t["l"], t["v"], n, w, None if shell else b, c if use_c else None, th if use_t else None, 4.0, shell)
    brdf, P, Bf, Hi, Hv, S, ci, cv = o
    for got, key in ((brdf, "brdf"), (P, "P"), (Hi, "Hi"), (Hv, "Hv"), (S, "S"), (ci, "ci"), (cv, "cv")):
        assert_close(got, g[key], 1e-5, 1e-6, key)
    (brdf * t["coef"]).sum().backward()
    assert_close(w.grad, g["dw"], 1e-4, 1e-6, "dw")
    if "dn" in g:        # shell_hapke==1 does not depend on the normal (reference grad is None)
        assert_close(n.grad, g["dn"], 1e-4, 1e-5, "dn")
    if not shell:
        assert_close(b.grad, g["db"], 1e-4, 1e-6, "db")
    if use_c:
        assert_close(c.grad, g["dc"], 1e-4, 1e-6, "dc")
    if use_t:
        assert_close(th.grad, g["dtheta"], 1e-4, 1e-5, "dtheta")


def test_brdf_microfacet():
    g = load_golden("brdf_microfacet")
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    n, w, r = [t[x].clone().requires_grad_(True) for x in ("n", "w", "rough")]
    gl, brdf, f, gg, d, ldn, vdn, h, n_h = B.microfacet(t["l"], t["v"], n, w, r, 0.04)
    for got, key in ((gl, "glossy"), (brdf, "brdf"), (f, "f"), (gg, "g"), (d, "d"), (ldn, "l_dot_n"),
                     (vdn, "v_dot_n"), (h, "h"), (n_h, "n_h")):
        assert_close(got, g[key], 1e-5, 1e-6, key)
    (brdf * t["coef"]).sum().backward()
    assert_close(w.grad, g["dw"], 1e-5, 1e-6, "dw")
    assert_close(n.grad, g["dn"], 1e-4, 1e-5, "dn")
    assert_close(r.grad, g["drough"], 1e-4, 1e-5, "drough")


def _render(name, mode, cfg=None, seed=11):
    g = load_golden(f"render_{name}_{mode}")
    cfg = cfg or mini(**CONFIGS[name])
    p = tparams(cfg, seed)
    for v in p.values():
        v.requires_grad_(True)
    rnd = RD.Randoms(replay=replay_list(g))
    kw = {}
    if mode == "train":
        kw = dict(valid_depth=torch.from_numpy(g["tgt/valid_depth"]), target_depths=torch.from_numpy(g["tgt/depths"]),
                  target_std=torch.from_numpy(g["tgt/depth_std"]))
    res, brdf_type = RD.render_rays(p, cfg, torch.from_numpy(g["rays"]), rnd, mode=mode,
                                    apply_brdf=name != "lambert", apply_theta=True, cos_irra_on=name != "lambert", **kw)
    assert rnd.replay == [], "oracle consumed a different number of random draws than the reference"
    return g, p, res, brdf_type


@pytest.mark.parametrize("mode", ["train", "test"])
@pytest.mark.parametrize("name", list(CONFIGS))
def test_render_rays_matches_reference(name, mode):
    g, p, res, brdf_type = _render(name, mode)
    assert brdf_type == str(g["brdf_type"])
    ref_keys = {k[4:] for k in g if k.startswith("out/")}
    got_keys = {k for k in res if not k.startswith("_")}
    assert ref_keys == got_keys, (sorted(ref_keys ^ got_keys))
    for k in sorted(ref_keys):
        if k == "sort_idx_coarse":
            assert np.array_equal(res[k].numpy(), g["out/" + k]), k
        else:
            assert_close(res[k], g["out/" + k], 2e-4, 2e-5, k)
    if mode == "train":
        tgt = torch.from_numpy(g["tgt/rgbs"])
        loss = torch.mean((res["rgb_coarse"] - tgt) ** 2) + 0.01 * torch.mean(res["depth_coarse"])
        assert_close(loss, g["loss"], 1e-5, 1e-7, "loss")
        loss.backward()
        for k, v in p.items():
            ref = g[f"grad/{k}"]
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            scale = max(float(np.abs(ref).max()), 1e-12)
            assert float((got - torch.from_numpy(ref)).abs().max()) <= 2e-3 * scale + 1e-9, k


def test_render_F512_and_blender():
    g = load_golden("render_rpv111_nan_F512")
    cfg = FieldConfig(**CONFIGS["rpv111_nan"])
    p = tparams(cfg, 12)
    res, bt = RD.render_rays(p, cfg, torch.from_numpy(g["rays"]), RD.Randoms(replay=replay_list(g)), mode="test",
                             apply_brdf=True, apply_theta=True, cos_irra_on=True)
    assert bt == "RPV"
    for k in [k[4:] for k in g if k.startswith("out/")]:
        if k == "sort_idx_coarse":
            assert np.array_equal(res[k].numpy(), g["out/" + k])
        else:
            assert_close(res[k], g["out/" + k], 1e-3, 1e-4, k)
    g = load_golden("render_lambert_blender")
    cfg = mini(data="blender")
    res, bt = RD.render_rays(tparams(cfg, 11), cfg, torch.from_numpy(g["rays"]), RD.Randoms(replay=replay_list(g)))
    for k in [k[4:] for k in g if k.startswith("out/")]:
        if k == "sort_idx_coarse":
            assert np.array_equal(res[k].numpy(), g["out/" + k])
        else:
            assert_close(res[k], g["out/" + k], 2e-4, 2e-5, k)


def test_losses():
    g = load_golden("loss_snerf_depth")
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    rgb, depth, w = [t[k].clone().requires_grad_(True) for k in ("rgb", "depth", "weights")]
    res = {"rgb_coarse": rgb, "depth_coarse": depth, "weights_coarse": w, "z_vals_coarse": t["z"]}
    l_rgb = L.snerf_loss(res, t["tgt"])
    l_ds = L.depth_loss(res, t["target_depths"][:, 0], t["target_depths"][:, 1], t["valid_depth"], t["target_std"], 10.0)
    assert_close(l_rgb, g["loss_rgb"], 1e-6, 1e-8, "loss_rgb")
    assert_close(l_ds, g["loss_ds"], 1e-6, 1e-8, "loss_ds")
    (l_rgb + l_ds).backward()
    assert_close(rgb.grad, g["drgb"], 1e-6, 1e-9, "drgb")
    assert_close(depth.grad, g["ddepth"], 1e-6, 1e-9, "ddepth")
    assert_close(L.psnr(t["rgb"], t["tgt"]), g["psnr"], 1e-6, 1e-6, "psnr")


@pytest.mark.parametrize("tag,extra,gs", [("rpv111_nlr_multibrdf", dict(MultiBRDF=True), False), ("rpv111_nlr_gsamonly", dict(), True)])
def test_render_variants_multibrdf_gsamonly(tag, extra, gs):
    g = load_golden(f"render_{tag}_test")
    cfg = mini(**dict(CONFIGS["rpv111_nlr"], **extra))
    res, bt = RD.render_rays(tparams(cfg, 11), cfg, torch.from_numpy(g["rays"]), RD.Randoms(replay=replay_list(g)), mode="test",
                             apply_brdf=True, apply_theta=True, cos_irra_on=True, gsam_only=gs)
    assert bt == str(g["brdf_type"])
    ref_keys = {k[4:] for k in g if k.startswith("out/")}
    assert ref_keys == {k for k in res if not k.startswith("_")}
    for k in sorted(ref_keys):
        if k == "sort_idx_coarse":
            assert np.array_equal(res[k].numpy(), g["out/" + k])
        else:
            assert_close(res[k], g["out/" + k], 2e-4, 2e-5, k)


@pytest.mark.parametrize("mode", ["train", "test"])
@pytest.mark.parametrize("base", ["rpv111_nlr", "lambert"])
def test_render_sun_visibility_pass(base, mode):
    """--sun_v analystic (rendering.py:244-259) with gsam_only=True, the combination the reference's pass 2 accepts."""
    g = load_golden(f"render_{base}_sunv_{mode}")
    cfg = mini(**dict(CONFIGS[base], sun_v="analystic"))
    p = tparams(cfg, 11)
    for v in p.values():
        v.requires_grad_(True)
    rnd = RD.Randoms(replay=replay_list(g))
    kw = {}
    if mode == "train":
        kw = dict(valid_depth=torch.from_numpy(g["tgt/valid_depth"]), target_depths=torch.from_numpy(g["tgt/depths"]),
                  target_std=torch.from_numpy(g["tgt/depth_std"]))
    res, bt = RD.render_rays(p, cfg, torch.from_numpy(g["rays"]), rnd, mode=mode, apply_brdf=True, apply_theta=True,
                             cos_irra_on=False, gsam_only=True, **kw)
    assert rnd.replay == [], "oracle consumed a different number of random draws than the reference"
    assert bt == str(g["brdf_type"])
    ref_keys = {k[4:] for k in g if k.startswith("out/")}
    assert ref_keys == {k for k in res if not k.startswith("_")}, sorted(ref_keys ^ {k for k in res if not k.startswith("_")})
    for k in sorted(ref_keys):
        assert_close(res[k], g["out/" + k], 2e-4, 2e-5, k)
    if mode == "train":
        loss = torch.mean((res["rgb_coarse"] - torch.from_numpy(g["tgt/rgbs"])) ** 2)
        loss.backward()
        for k, v in p.items():
            ref = g[f"grad/{k}"]
            got = v.grad if v.grad is not None else torch.zeros_like(v)
            scale = max(float(np.abs(ref).max()), 1e-12)
            assert float((got - torch.from_numpy(ref)).abs().max()) <= 2e-3 * scale + 1e-9, k


@pytest.mark.parametrize("tag,kw", [("relu", dict(siren=False)), ("nomap", dict(mapping=False)), ("rpv333", dict(dim_RPV=3))])
def test_field_relu_and_no_mapping(tag, kw):
    """--siren 0 and no --mapping: forward and parameter gradients against the reference."""
    g = load_golden(f"field_{tag}_F64")
    cfg = mini(funcM=1, funcF=1, funcH=1, normal="learned", **kw)
    p = tparams(cfg, 13)
    for v in p.values():
        v.requires_grad_(True)
    out = F.field_forward(p, cfg, torch.from_numpy(g["xyz"]), apply_brdf=True, apply_theta=True, nr_lr_on=True)
    assert_close(out, g["out_brdf"], 1e-5, 1e-6, "out")
    (out * torch.from_numpy(g["coef"])).sum().backward()
    for k, v in p.items():
        ref = g[f"grad/{k}"]
        scale = max(float(np.abs(ref).max()), 1e-12)
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert float((got - torch.from_numpy(ref)).abs().max()) <= 1e-4 * scale + 1e-9, k


@pytest.mark.parametrize("tag,kw", [("viewdir", dict(input_viewdir=1, funcM=1, funcF=1, funcH=1, normal="learned")),
                                    ("viewdir_nomap", dict(input_viewdir=1, mapping=False))])
def test_field_input_viewdir(tag, kw):
    """--input_viewdir 1 (spsbrdfnerf.py:458,689-692): the rgb head reads the encoded view direction; forward with per-point
    directions and parameter gradients against the reference."""
    g = load_golden(f"field_{tag}_F64")
    cfg = mini(**kw)
    p = tparams(cfg, 14)
    for v in p.values():
        v.requires_grad_(True)
    out = F.field_forward(p, cfg, torch.from_numpy(g["xyz"]), apply_brdf=True, apply_theta=True, nr_lr_on=cfg.normal == "learned",
                          dirs=torch.from_numpy(g["dirs"]))
    assert_close(out, g["out_brdf"], 1e-5, 1e-6, "out")
    (out * torch.from_numpy(g["coef"])).sum().backward()
    for k, v in p.items():
        ref = g[f"grad/{k}"]
        scale = max(float(np.abs(ref).max()), 1e-12)
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert float((got - torch.from_numpy(ref)).abs().max()) <= 1e-4 * scale + 1e-9, k


def test_render_rays_input_viewdir():
    """render_rays with --input_viewdir 1 (two view directions in the batch), train mode, RPV + learned normals."""
    g = load_golden("render_viewdir_train")
    cfg = mini(input_viewdir=1, funcM=1, funcF=1, funcH=1, normal="learned")
    p = tparams(cfg, 11)
    for v in p.values():
        v.requires_grad_(True)
    res, bt = RD.render_rays(p, cfg, torch.from_numpy(g["rays"]), RD.Randoms(replay=replay_list(g)), mode="train",
                             apply_brdf=True, apply_theta=True, cos_irra_on=True)
    assert bt == str(g["brdf_type"])
    for k in sorted(k[4:] for k in g if k.startswith("out/")):
        if k == "sort_idx_coarse":
            assert (res[k].numpy() == g["out/" + k]).all()
        else:
            assert_close(res[k], g["out/" + k], 2e-5, 2e-6, k)
    loss = ((res["rgb_coarse"] - torch.from_numpy(g["targets"])) ** 2).mean() + 0.01 * res["depth_coarse"].mean()
    assert_close(loss, g["loss"], 1e-5, 1e-7, "loss")
    loss.backward()
    for k, v in p.items():
        ref = g[f"grad/{k}"]
        scale = max(float(np.abs(ref).max()), 1e-12)
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert float((got - torch.from_numpy(ref)).abs().max()) <= 2e-4 * scale + 1e-9, k


@pytest.mark.parametrize("tag,kw", [("beta", dict(beta=True, funcM=1, funcF=1, funcH=1, normal="learned")),
                                    ("beta_viewdir_relu", dict(beta=True, input_viewdir=1, siren=False, t_dim=6))])
def test_field_beta(tag, kw):
    """--beta (spsbrdfnerf.py:571-575,708-711): the transient-uncertainty channel after sigma; forward, parameter gradients
    and the gradient w.r.t. the per-point embedding input against the reference."""
    g = load_golden(f"field_{tag}_F64")
    cfg = mini(**kw)
    p = tparams(cfg, 15)
    for v in p.values():
        v.requires_grad_(True)
    t_in = torch.from_numpy(g["t_in"]).requires_grad_(True)
    out = F.field_forward(p, cfg, torch.from_numpy(g["xyz"]), apply_brdf=True, apply_theta=True, nr_lr_on=cfg.normal == "learned",
                          dirs=torch.from_numpy(g["dirs"]), t_embed=t_in)
    assert_close(out, g["out_brdf"], 1e-5, 1e-6, "out")
    (out * torch.from_numpy(g["coef"])).sum().backward()
    assert_close(t_in.grad, g["d_t_in"], 1e-4, 1e-7, "d_t_in")
    for k, v in p.items():
        ref = g[f"grad/{k}"]
        scale = max(float(np.abs(ref).max()), 1e-12)
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert float((got - torch.from_numpy(ref)).abs().max()) <= 1e-4 * scale + 1e-9, k


def test_render_rays_beta():
    """render_rays with --beta and models['t'](ts) (rendering.py:226-229), train mode; uncertainty_aware_loss
    (metrics.py:24-28) reads beta_coarse; gradients of the parameters and of the embedding table."""
    g = load_golden("render_beta_train")
    cfg = mini(beta=True, funcM=1, funcF=1, funcH=1, normal="learned")
    p = tparams(cfg, 11)
    for v in p.values():
        v.requires_grad_(True)
    emb = torch.from_numpy(g["emb"]).requires_grad_(True)
    rays_t = emb[torch.from_numpy(g["ts"])]
    res, bt = RD.render_rays(p, cfg, torch.from_numpy(g["rays"]), RD.Randoms(replay=replay_list(g)), mode="train",
                             apply_brdf=True, apply_theta=True, cos_irra_on=True, rays_t=rays_t)
    assert bt == str(g["brdf_type"])
    assert {k[4:] for k in g if k.startswith("out/")} == {k for k in res if not k.startswith("_")}
    for k in sorted(k[4:] for k in g if k.startswith("out/")):
        if k == "sort_idx_coarse":
            assert (res[k].numpy() == g["out/" + k]).all()
        else:
            assert_close(res[k], g["out/" + k], 2e-5, 2e-6, k)
    l_color, l_logbeta = L.uncertainty_aware_loss(res["rgb_coarse"], res["weights_coarse"], res["beta_coarse"],
                                                  torch.from_numpy(g["targets"]))
    assert_close(l_color, g["loss_color"], 1e-5, 1e-7, "loss_color")
    assert_close(l_logbeta, g["loss_logbeta"], 1e-5, 1e-7, "loss_logbeta")
    loss = l_color + l_logbeta + 0.01 * res["depth_coarse"].mean()
    assert_close(loss, g["loss"], 1e-5, 1e-7, "loss")
    loss.backward()
    assert_close(emb.grad, g["d_emb"], 2e-4, 1e-7, "d_emb")
    for k, v in p.items():
        ref = g[f"grad/{k}"]
        scale = max(float(np.abs(ref).max()), 1e-12)
        got = v.grad if v.grad is not None else torch.zeros_like(v)
        assert float((got - torch.from_numpy(ref)).abs().max()) <= 2e-4 * scale + 1e-9, k


def test_regulariser_losses():
    g = load_golden("loss_regularisers")
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    w, depth, n_an, n_lr = [t[k].clone().requires_grad_(True) for k in ("weights", "depth", "normal_an", "normal_lr")]
    res = {"normal_an_coarse": n_an, "normal_lr_coarse": n_lr, "weights_coarse": w, "rays_d_coarse": t["view"],
           "z_vals_coarse": t["z"], "depth_coarse": depth}
    l_an, perc_an = L.normal_reg_loss(res, "normal_an", 0.1)
    l_lr, perc_lr = L.normal_reg_loss(res, "normal_lr", 0.05)
    l_hs = L.hard_surface_loss(res, 0.5)
    l_n1 = L.normal_loss(w, n_an, n_lr, 0.01, "an_lr")
    l_n3 = L.normal_loss(w, t["normal_gt"], n_an, 0.01, "an", target_weight=t["target_weight"], valid_depth=t["valid_depth"])
    for name, got in (("l_nr_an", l_an), ("l_nr_lr", l_lr), ("l_hs", l_hs), ("l_n1", l_n1), ("l_n3", l_n3)):
        assert_close(got, g[name], 1e-5, 1e-9, name)
    assert abs(perc_an - float(g["perc_an"])) < 1e-3 and abs(perc_lr - float(g["perc_lr"])) < 1e-3
    (l_an + l_lr + l_hs + l_n1 + l_n3).backward()
    assert_close(w.grad, g["d_weights"], 1e-5, 1e-9, "d_weights")
    assert_close(depth.grad, g["d_depth"], 1e-5, 1e-9, "d_depth")
    assert_close(n_an.grad, g["d_normal_an"], 1e-5, 1e-10, "d_normal_an")
    assert_close(n_lr.grad, g["d_normal_lr"], 1e-5, 1e-10, "d_normal_lr")


# ---- round 3: branches of inference() no fixture reached before (funcH == 2, shell_hapke with apply_brdf=False, ref_sphere)
BRANCHES = {
    "rpv_m1f1h2": (dict(funcM=1, funcF=1, funcH=2, normal="learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
    "rpv_m1f1h2_multibrdf": (dict(funcM=1, funcF=1, funcH=2, normal="learned", MultiBRDF=True),
                             dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
    "rpv_m1h2": (dict(funcM=1, funcH=2, normal="learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
    "shell1_nobrdf": (dict(shell_hapke=1, normal="learned"), dict(apply_brdf=False, apply_theta=False, cos_irra_on=False)),
    "shell2_nobrdf": (dict(shell_hapke=2, normal="learned"), dict(apply_brdf=False, apply_theta=False, cos_irra_on=True)),
    "shell3_nobrdf": (dict(shell_hapke=3, normal="learned"), dict(apply_brdf=False, apply_theta=False, cos_irra_on=True)),
    "shell3_brdf": (dict(shell_hapke=3, normal="learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
}


def _check_render(res, g):
    ref_keys = {k[4:] for k in g if k.startswith("out/")}
    got_keys = {k for k in res if not k.startswith("_")}
    assert ref_keys == got_keys, sorted(ref_keys ^ got_keys)
    for k in sorted(ref_keys):
        if k == "sort_idx_coarse":
            assert np.array_equal(res[k].numpy(), g["out/" + k]), k
        else:
            assert_close(res[k], g["out/" + k], 2e-4, 2e-5, k)


@pytest.mark.parametrize("name", list(BRANCHES))
def test_render_rays_unpinned_branches(name):
    """models/spsbrdfnerf.py:306,317 (funcH == 2: rhoc := albedo, per ray and per sample), :320,348,383 (shell_hapke > 0 shades
    with the Hapke shell even with apply_brdf=False)."""
    g = load_golden(f"render_{name}_test")
    kw, flags = BRANCHES[name]
    cfg = mini(**kw)
    assert abs(checksum(cfg, 11) - float(g["param_checksum"])) < 1e-9
    res, bt = RD.render_rays(tparams(cfg, 11), cfg, torch.from_numpy(g["rays"]), RD.Randoms(replay=replay_list(g)), mode="test", **flags)
    assert bt == str(g["brdf_type"])
    _check_render(res, g)


def test_render_rays_ref_sphere():
    """rows / cols -> ref_sphere (models/spsbrdfnerf.py:404-412), with the reference's tiling order."""
    g = load_golden("render_rpv111_nlr_refsphere_test")
    cfg = mini(**CONFIGS["rpv111_nlr"])
    res, bt = RD.render_rays(tparams(cfg, 11), cfg, torch.from_numpy(g["rays"]), RD.Randoms(replay=replay_list(g)), mode="test",
                             apply_brdf=True, apply_theta=True, cos_irra_on=True, rows=torch.from_numpy(g["rows"]),
                             cols=torch.from_numpy(g["cols"]))
    assert "ref_sphere_coarse" in res
    _check_render(res, g)
    assert np.array_equal(res["ref_sphere_coarse"].numpy(), g["out/ref_sphere_coarse"])
