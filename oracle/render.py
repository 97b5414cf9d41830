"""Oracle: ray sampling, alpha compositing, depth-guided resampling and the two-pass
`render_rays` of the spsbrdf-nerf variant (PyTorch CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.

Follows (path:line under /root/reference):
  sample_pdf / sample_3sigma / sample_3sigma_asym     rendering.py:13-91
  compute_samples_around_depth / GenerateGuidedSamples rendering.py:116-147
  get_z_vals                                           rendering.py:149-166
  render_rays (spsbrdf branch)                         rendering.py:168-291
  cal_weight                                           models/spsbrdfnerf.py:50-69
  inference                                            models/spsbrdfnerf.py:71-416
  calc_depth_std                                       train_utils.py:35-39

Random numbers are supplied by a `Randoms` object so that goldens can replay the draws the
reference made, in its order (SURVEY.md section 8c "RNG protocol").
"""
import math
import torch

from . import brdf as B
from .field import field_forward, l2_normalize


class Randoms:
    """Source of the path's random draws.  `replay` = list of tensors in draw order."""

    def __init__(self, replay=None, generator=None):
        self.replay = list(replay) if replay is not None else None
        self.generator = generator
        self.log = []

    def _next(self, kind, shape, dtype):
        if self.replay is not None:
            t = self.replay.pop(0)
            assert tuple(t.shape) == tuple(shape), (kind, tuple(t.shape), tuple(shape))
            t = t.to(dtype)
        elif kind == "u":
            t = torch.rand(shape, generator=self.generator, dtype=dtype)
        else:
            t = torch.randn(shape, generator=self.generator, dtype=dtype)
        self.log.append(t)
        return t

    def rand(self, shape, dtype=torch.float32):
        return self._next("u", shape, dtype)

    def randn(self, shape, dtype=torch.float32):
        return self._next("n", shape, dtype)


def get_z_vals(n_samples, near, far, u):
    """rendering.py:149-166 with perturb=1 (hard-coded at :175).  near,far: (R,1); u: (R,S) in [0,1)."""
    t = torch.linspace(0, 1, n_samples, dtype=near.dtype)
    z = near * (1 - t) + far * t
    mid = 0.5 * (z[:, :-1] + z[:, 1:])
    upper = torch.cat([mid, z[:, -1:]], -1)
    lower = torch.cat([z[:, :1], mid], -1)
    return lower + (upper - lower) * u


def composite(z, sigma, noise=None, noise_std=0.0):
    """cal_weight (spsbrdfnerf.py:50-69) -> alphas, transparency, weights, depth."""
    deltas = torch.cat([z[:, 1:] - z[:, :-1], 1e10 * torch.ones_like(z[:, :1])], -1)
    s = sigma if noise is None else sigma + noise * noise_std
    alphas = 1 - torch.exp(-deltas * torch.relu(s))
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    T = torch.cumprod(shifted, -1)[:, :-1]
    w = alphas * T
    return alphas, T, w, (w * z).sum(-1)


def depth_std(z, depth, w):
    """train_utils.py:35-39."""
    return (((z - depth.unsqueeze(-1)) ** 2) * w).sum(-1).sqrt()


def sample_pdf(bins, weights, u, eps=1e-5):
    """rendering.py:13-52 with det=False; u: (R,N) uniform.  Returns samples, inds (int64)."""
    n = weights.shape[1]
    w = weights + eps
    pdf = w / w.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp_min(0)
    above = inds.clamp_max(n)
    cdf0, cdf1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf1 - cdf0
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    return b0 + (u - cdf0) / denom * (b1 - b0), inds


def sample_3sigma_asym(depth, lo, hi, n, near0, far0, u, d_range=3.0, eps=1e-5):
    """rendering.py:54-91: symmetric clamp around depth, Gaussian-weighted bins, inverse-CDF, sort."""
    lo = lo.clamp(near0, far0)
    hi = hi.clamp(near0, far0)
    rng = torch.minimum((hi - depth).abs(), (lo - depth).abs())
    lo, hi = depth - rng, depth + rng
    t = torch.linspace(0.0, 1.0, n, dtype=depth.dtype)
    step = (hi - lo) / (n - 1)
    edges = lo.unsqueeze(-1) * (1.0 - t) + hi.unsqueeze(-1) * t
    factor = (edges[:, 1:] - edges[:, :-1]) / (step.unsqueeze(-1) + eps)
    x = torch.linspace(-d_range, d_range, n - 1, dtype=depth.dtype)
    bw = factor * (1.0 / math.sqrt(2 * math.pi) * torch.exp(-0.5 * x ** 2)).unsqueeze(0)
    s, inds = sample_pdf(edges, bw, u)
    return torch.sort(s, -1)[0], inds


def guided_samples(depth, weights, z, n_guided, near0, far0, rnd, d_range=3.0, mode="test",
                   valid_depth=None, target_depths=None, target_std=None):
    """GenerateGuidedSamples (rendering.py:132-147).  Returns z2 (R,G) and searchsorted indices."""
    std = depth_std(z, depth, weights)
    z2, inds = sample_3sigma_asym(depth, depth - d_range * std, depth + d_range * std, n_guided,
                                  near0, far0, rnd.rand((depth.shape[0], n_guided), depth.dtype), d_range)
    inds_gt = None
    if mode == "train" and valid_depth is not None:
        sel = valid_depth > 0
        td = target_depths[:, 0][sel].flatten()
        ts = target_std[sel].flatten()
        gt, inds_gt = sample_3sigma_asym(td, td - d_range * ts, td + d_range * ts, n_guided, near0, far0,
                                         rnd.rand((int(sel.sum()), n_guided), depth.dtype), d_range)
        z2 = z2.clone()
        z2[sel] = gt
    return z2, inds, inds_gt


def ref_sphere(rows, cols, R, S, like):
    """spsbrdfnerf.py:404-412.  rows / cols (R, 1) in [-1, 1] are tiled S times along the ray axis and read back as (R, S)[:, 0]:
    ray r gets element (r * S) mod R of the input, not element r - restated as written upstream."""
    sel = lambda t: t.reshape(-1).repeat(S).reshape(R, S)[:, 0]
    out = torch.ones(R, 1, 3, dtype=like.dtype)
    out[:, 0, 0] = sel(cols)
    out[:, 0, 1] = -sel(rows)
    out[:, 0, 2] = sel(torch.sqrt(torch.abs(1 - rows * rows - cols * cols)))
    return out


def inference(params, cfg, xyz, z, rays_d, sun_d, rnd, sigma_only=False, apply_brdf=False,
              apply_theta=False, cos_irra_on=False, sort_idx=None, z_unsort=None, bTestNormal=False, sun_res=None,
              rays_t=None, rows=None, cols=None):
    """inference (spsbrdfnerf.py:71-416) for sun_v in ('none', 'analystic').  Returns (dict, brdf_type).
    rays_t (R, t_dim): per-ray image embedding for --beta (repeated per sample, :98)."""
    R, S = z.shape
    pts = xyz.reshape(-1, 3)
    nr_an = cfg.normal in ("analystic", "analystic_learned") or bTestNormal
    nr_lr = cfg.normal in ("learned", "analystic_learned")
    dirs = torch.repeat_interleave(rays_d, S, dim=0) if cfg.dir_dim else None      # spsbrdfnerf.py:96,121
    out = field_forward(params, cfg, pts, sigma_only=sigma_only, apply_brdf=apply_brdf,
                        apply_theta=apply_theta, nr_an_on=nr_an, nr_lr_on=nr_lr, dirs=dirs,
                        t_embed=torch.repeat_interleave(rays_t, S, dim=0) if (cfg.beta and not sigma_only) else None)
    noise = rnd.randn((R, S), z.dtype)
    if sigma_only:
        sig = out.view(R, S)
        a, T, w, d = composite(z, sig, noise, cfg.noise_std)
        return {"sigmas": sig.unsqueeze(-1), "depth": d, "alphas": a, "weights": w,
                "transparency": T, "z_vals": z}, "Lambertian"
    C = out.shape[1]
    out = out.view(R, S, C)
    albedo, sig = out[..., :3], out[..., 3]
    idx = 4
    normal = None
    res = {}
    if cfg.beta:                                 # :156-158, stored as result['beta'] (:225-226)
        res["beta"] = out[..., idx:idx + 1]
        idx += 1
    if nr_an:
        res["normal_an"] = normal = out[..., idx:idx + 3]
        idx += 3
    if nr_lr:
        res["normal_lr"] = normal = out[..., idx:idx + 3]      # learned wins when both (:237-239)
        idx += 3
    heads = {}
    for name in cfg.brdf_head_names(apply_brdf, apply_theta):
        n = 1 if name in ("roughness_from_xyz", "theta_from_xyz") else 3
        heads[name] = out[..., idx:idx + n]
        idx += n
    a, T, w, depth = composite(z, sig, noise, cfg.noise_std)
    wx = w.unsqueeze(-1)
    result = {"sigmas": sig.unsqueeze(-1), "albedo": albedo,
              "albedo_accu": (wx * albedo).sum(-2).clamp(0.0, 1.0), "depth": depth, "alphas": a,
              "weights": w, "transparency": T, "z_vals": z}
    sun_v = None
    if sun_res and "sun" in sun_res:            # :148-151, :211-219 (both branches store the same two entries)
        result["sun"], result["weights_sc"] = sun_res["sun"], sun_res["weights_sc"]
        if cfg.sun_v == "analystic":
            sun_v = sun_res["sun"]
    if sort_idx is not None:
        result["sort_idx"] = sort_idx
    if z_unsort is not None:
        result["z_vals_unsort"] = z_unsort
    result.update(res)
    view = -rays_d
    if normal is not None:
        normal_s = l2_normalize((wx * normal).sum(-2))
        result["nr_vw"] = (normal_s * view).sum(-1).reshape(R, 1, 1)
        result["nr_sun"] = (normal_s * sun_d).sum(-1).reshape(R, 1, 1)
        result["hpk_scl"] = 1.0 / (cfg.hpk_scl * (result["nr_vw"] + result["nr_sun"]))
    irr = torch.ones_like(albedo)
    if cos_irra_on and normal is not None:
        irr = irr * sun_d[:, None, 2:3].abs()                   # upward normal (0,0,1): :260-264
    elif sun_v is not None:
        irr = sun_v.expand(-1, -1, 3)                           # per-sample sun visibility (:265-266)
    pad = cfg.rgb_padding
    albedo_p = albedo * (1 + 2 * pad) - pad
    result["rgb"] = (wx * albedo_p * irr).sum(-2).clamp(0.0, 1.0)
    albedo_s = (wx * albedo_p).sum(-2)
    if idx == 4:
        return result, "Lambertian"
    brdf_type = "Lambertian"
    rgb = result["rgb"]
    extra = {}
    if cfg.roughness and apply_brdf:
        brdf_type = "Microfacet"
        if cfg.MultiBRDF:
            rep = lambda t: t.repeat_interleave(S, 0)
            gl, brdf, f, g, d, ldn, vdn, h, n_h = B.microfacet(rep(sun_d), rep(view), normal.reshape(-1, 3),
                                                                albedo.reshape(-1, 3),
                                                                heads["roughness_from_xyz"].reshape(-1, 1),
                                                                cfg.fresnel_f0)
        else:
            rough_s = (w * heads["roughness_from_xyz"].reshape(R, S)).sum(-1, keepdim=True)
            gl, brdf, f, g, d, ldn, vdn, h, n_h = B.microfacet(sun_d, view, normal_s, albedo_s, rough_s,
                                                                cfg.fresnel_f0)
        nb = S if cfg.MultiBRDF else 1
        extra = {"roughness": heads["roughness_from_xyz"], "glossy": gl.reshape(R, nb, 1),
                 "brdf": brdf.reshape(R, nb, 3), "f": f.reshape(R, nb, 1), "g": g.reshape(R, nb, 1),
                 "d": d.reshape(R, nb, 1), "l_dot_n": ldn.reshape(R, nb, 1), "v_dot_n": vdn.reshape(R, nb, 1),
                 "halfvec": h.reshape(R, nb, 3), "n_h": n_h.reshape(R, nb, 1)}
    elif cfg.RPV and apply_brdf:
        brdf_type = "RPV"
        k_, t_, r_ = heads.get("k_from_xyz"), heads.get("theta_rpv_from_xyz"), heads.get("rhoc_from_xyz")
        if cfg.MultiBRDF:
            rep = lambda t: t.repeat_interleave(S, 0)
            fl = lambda t: None if t is None else t.reshape(-1, 3)
            rh = albedo.reshape(-1, 3) if cfg.funcH == 2 else fl(r_)
            brdf = B.rpv(rep(sun_d), rep(view), normal.reshape(-1, 3), albedo.reshape(-1, 3), fl(k_), fl(t_), rh)[0]
        else:
            ws = lambda t: None if t is None else (wx * t).sum(-2)
            rh = albedo_s if cfg.funcH == 2 else ws(r_)
            brdf = B.rpv(sun_d, view, normal_s, albedo_s, ws(k_), ws(t_), rh)[0]
        for key, t in (("rpv_k", k_), ("rpv_theta", t_), ("rpv_rhoc", r_)):
            if t is not None:
                extra[key] = t
    elif (apply_brdf and cfg.b == 1) or cfg.shell_hapke > 0:
        brdf_type = "Hapke"
        hb = heads.get("b_from_xyz") if apply_brdf else None
        hc = heads.get("c_from_xyz") if apply_brdf else None
        ht = heads.get("theta_from_xyz")
        if cfg.MultiBRDF:
            rep = lambda t: t.repeat_interleave(S, 0)
            fl = lambda t: None if t is None else t.reshape(-1, 3)
            o = B.hapke(rep(sun_d), rep(view), normal.reshape(-1, 3), albedo.reshape(-1, 3), fl(hb), fl(hc),
                        None if ht is None else ht.reshape(-1), cfg.hpk_scl, cfg.shell_hapke)
        else:
            ws = lambda t: None if t is None else (wx * t).sum(-2)
            th_s = None if ht is None else (w * ht.reshape(R, S)).sum(-1)
            o = B.hapke(sun_d, view, normal_s, albedo_s, ws(hb), ws(hc), th_s, cfg.hpk_scl, cfg.shell_hapke)
        brdf, P, Bf, Hi, Hv, Sh, ci, cv = o
        nb = S if cfg.MultiBRDF else 1
        if apply_brdf:
            extra = {"brdf": brdf.reshape(R, nb, 3), "hpk_P": P.reshape(R, nb, 3), "hpk_Hi": Hi.reshape(R, nb, 3),
                     "hpk_Hv": Hi.reshape(R, nb, 3),              # reference quirk 6: filled from Hi (:387)
                     "hpk_ci": ci.reshape(R, nb, 1), "hpk_cv": cv.reshape(R, nb, 1),
                     "hpk_ShadFunc": Sh.reshape(R, nb, 1)}
            if hb is not None:
                extra["hpk_b"] = hb
            if hc is not None:
                extra["hpk_c"] = hc
            if ht is not None:
                extra["hpk_theta"] = ht
    if apply_brdf or cfg.shell_hapke > 0:
        if cfg.MultiBRDF:
            bp = brdf.reshape(R, S, 3) * (1 + 2 * pad) - pad
            rgb = (wx * bp * irr).sum(-2)
        else:
            rgb = irr[:, -1, :] * brdf.reshape(R, 3)             # irradiance of the LAST sample (:354)
    result["rgb"] = rgb.clamp(0.0, 1.0)
    result["irradiance"] = irr
    if apply_brdf:
        result.update(extra)
    result["rays_d"] = view.reshape(R, 1, 3)
    result["sun_d"] = sun_d.reshape(R, 1, 3)
    if rows is not None and cols is not None:                    # (:404-412; not reached by the Lambertian early return)
        result["ref_sphere"] = ref_sphere(rows, cols, R, S, rays_d)
    return result, brdf_type


def render_rays(params, cfg, rays, rnd, mode="test", valid_depth=None, target_depths=None, target_std=None,
                apply_brdf=False, apply_theta=False, cos_irra_on=False, gsam_only=False, bTestNormal=False,
                bTestSun_v=False, rays_t=None, rows=None, cols=None):
    """render_rays, spsbrdf-nerf branch (rendering.py:168-291), guided_samples>0, sun_v in ('none', 'analystic').
    rays_t = models['t'](ts) (rendering.py:226-229), needed with cfg.beta."""
    o, d, near, far = rays[:, 0:3], rays[:, 3:6], rays[:, 6:7], rays[:, 7:8]
    S, G = cfg.n_samples, cfg.guided_samples
    assert G > 0, "guided_samples<=0 returns an un-suffixed dict in the reference (SURVEY quirk 1)"
    z = get_z_vals(S, near, far, rnd.rand(tuple(near.expand(-1, S).shape), rays.dtype))
    sun_d = rays[:, 8:11] if cfg.data == "sat" else torch.ones_like(o)
    xyz = o.unsqueeze(1) + d.unsqueeze(1) * z.unsqueeze(2)
    with torch.no_grad():   # pass 1 feeds only detached consumers
        res1, _ = inference(params, cfg, xyz, z, d, sun_d, rnd, sigma_only=True)
    d_range, g_r = cfg.std_range, G
    if G == 2:
        d_range, g_r = 0.0001, 1
    sun_res = {}
    if (cfg.sun_v == "analystic" and apply_brdf) or bTestSun_v:
        # sun-visibility pass (rendering.py:244-259): transparency along the sun direction from the pass-1 surface point.
        # The reference's pass 2 only accepts it with gsam_only; far_sun uses ROW 0's directions.
        assert gsam_only, "sun_v analystic with gsam_only=False raises in the reference (SURVEY quirk 2)"
        pt_surf = o + d * res1["depth"].unsqueeze(-1)
        far_sun = res1["depth"].clone().unsqueeze(-1)
        if abs(float(sun_d[0, 2])) > 0.00001:
            far_sun = torch.abs(d[0, 2] / sun_d[0, 2]) * far_sun
        n1 = g_r
        z_sun = get_z_vals(n1, far_sun * 0.01, far_sun, rnd.rand((rays.shape[0], n1), rays.dtype))
        xyz_sun = pt_surf.unsqueeze(1) + sun_d.unsqueeze(1) * z_sun.unsqueeze(2)
        with torch.no_grad():
            rs, _ = inference(params, cfg, xyz_sun, z_sun, sun_d, None, rnd, sigma_only=True)
        sun_res = {"sun": rs["transparency"].unsqueeze(-1).detach(), "weights_sc": rs["weights"].detach()}
    z2, inds, inds_gt = guided_samples(res1["depth"], res1["weights"], z, G, near[0, 0], far[0, 0], rnd, d_range,
                                       mode, valid_depth, target_depths, target_std)
    z2 = torch.sort(z2.detach(), -1)[0]
    if g_r == 1:
        z2 = z2.mean(1, keepdim=True)
    if gsam_only:
        z_unsort, z_all, idx = z2, z2, None
    else:
        z_unsort = torch.cat([z, z2], -1)
        z_all, idx = torch.sort(z_unsort, -1)
    xyz = o.unsqueeze(1) + d.unsqueeze(1) * z_all.unsqueeze(2)
    res, brdf_type = inference(params, cfg, xyz, z_all, d, sun_d, rnd, apply_brdf=apply_brdf,
                               apply_theta=apply_theta, cos_irra_on=cos_irra_on, sort_idx=idx,
                               z_unsort=z_unsort, bTestNormal=bTestNormal, sun_res=sun_res, rays_t=rays_t, rows=rows, cols=cols)
    out = {f"{k}_coarse": v for k, v in res.items()}
    out["_pass1"] = res1
    out["_guided_inds"] = inds
    out["_guided_inds_gt"] = inds_gt
    return out, brdf_type
