"""Drop-in `render_rays` / `inference` / `get_z_vals` for the spsbrdf-nerf variant, running on the HIP
library (reference: rendering.py:149-291, models/spsbrdfnerf.py:50-416).

Same signatures, same result-dict keys, same RNG draw order
(torch.rand_like (R,S) -> torch.randn (R,S) -> torch.rand (R,G) [-> torch.rand (n_valid,G)] ->
torch.randn (R,S+G)), so a harness that monkeypatches torch.rand/rand_like/randn replays the
reference's draws.  Documented differences: the reference's print/`check_nan` host syncs are not
reproduced; pass 1 runs without autograd (its result is detached upstream, rendering.py:262).
"""
import torch

from . import functions as Fn

FP32_EPS = float(torch.finfo(torch.float32).eps)


def l2_normalize(x):
    """train_utils.py:28-33."""
    return x / torch.sqrt(torch.clamp_min((x * x).sum(-1, keepdim=True), FP32_EPS))


def sun_far(depth, rays_d, sun_d):
    """far bound of the sun-visibility pass (rendering.py:246-248): depth scaled by |d_z / sun_z| of ROW 0 (fp32 tensor
    arithmetic like upstream, without its host read)."""
    far_sun = depth.clone().unsqueeze(-1)
    s0, r0 = sun_d[0, 2], rays_d[0, 2]
    ratio = torch.where(s0.abs() > 0.00001, (r0 / s0).abs(), torch.ones_like(s0))
    return ratio * far_sun


def get_z_vals(N_samples, device, near, far, use_disp=False, perturb=1.0):
    if use_disp or perturb != 1.0:
        raise NotImplementedError("render path always uses linear depth with perturb=1 (rendering.py:175)")
    u = torch.rand_like(near.expand(-1, N_samples).contiguous())
    return Fn.stratified_z(near, far, u)


def cal_weight(z_vals, sigmas, args):
    """models/spsbrdfnerf.py:50-69 (standalone form; `inference` uses the fused variant)."""
    noise = torch.randn(sigmas.shape, device=sigmas.device)
    a, T, w, d = Fn.composite(z_vals, sigmas.contiguous(), noise, args.noise_std)
    return a, T, w, d


def _per_ray_brdf(model, args, brdf_kind, sun_d, view, normal_s, albedo_s, s):
    """One BRDF per ray (MultiBRDF == False).  `s` maps head name -> weighted per-ray sum."""
    if brdf_kind == "Microfacet":
        brdf, aux = Fn.MicrofacetFunction.apply(sun_d, view, normal_s, albedo_s, s["roughness_from_xyz"], args.fresnel_f0)
    elif brdf_kind == "RPV":
        rh = albedo_s if args.funcH == 2 else s.get("rhoc_from_xyz")
        brdf, aux = Fn.RPVFunction.apply(sun_d, view, normal_s, albedo_s, s.get("k_from_xyz"), s.get("theta_rpv_from_xyz"), rh)
    else:
        th = s.get("theta_from_xyz")
        brdf, aux = Fn.HapkeFunction.apply(sun_d, view, normal_s, albedo_s, s.get("b_from_xyz"), s.get("c_from_xyz"),
                                           None if th is None else th.reshape(-1), args.hpk_scl, args.shell_hapke)
    return brdf, aux


def inference(model, args, rays_xyz, z_vals, rays_d=None, sun_d=None, rays_t=None, z_vals_unsort=None, apply_brdf=False,
              print_debuginfo=False, bTestNormal=False, sun_res=[], sort_idx=None, rows=None, cols=None, percent=0,
              mode="train", apply_theta=False, sigma_only=False, cos_irra_on=False, _rays=None, _packed=None):
    """inference (models/spsbrdfnerf.py:71-416).  `rays_xyz` (R,S,3) may be None when `_rays` (R,>=8) is given: the
    kernel then forms xyz = o + d*z itself (what render_rays does)."""
    R, S = z_vals.shape
    z_vals = z_vals.contiguous()
    nr_lr = model.normal in ("analystic_learned", "learned")
    nr_an = model.normal in ("analystic_learned", "analystic") or bTestNormal
    spec = model.spec(apply_brdf and not sigma_only, apply_theta, nr_lr and not sigma_only, nr_an and not sigma_only)
    packed = _packed if _packed is not None else model.repack(spec)
    xyz = None if _rays is not None else rays_xyz.reshape(-1, 3).detach().float().contiguous()
    noise = torch.randn(R, S, device=z_vals.device)   # drawn even when noise_std == 0, like the reference (:58)
    noise_arg = noise if args.noise_std != 0 else None
    if sigma_only:
        with torch.no_grad():
            sig = Fn.field_sigma(spec, model.named(), packed, xyz=xyz, rays=_rays, z=z_vals).view(R, S)
            a, T, w, d = Fn.composite(z_vals, sig, noise_arg, args.noise_std)
        return {"sigmas": sig.unsqueeze(-1), "depth": d, "alphas": a, "weights": w, "transparency": T,
                "z_vals": z_vals}, "Lambertian"

    dirs = t_embed = None
    if _rays is None and getattr(model, "dir_dim", 0):       # --input_viewdir with explicit points: rays_d per sample (:96,121)
        dirs = torch.repeat_interleave(rays_d.float(), S, dim=0).contiguous()
    if model.beta:                                           # --beta: the image embedding of the ray, repeated per sample (:98)
        if rays_t is None:
            raise ValueError("--beta: inference() needs rays_t = models['t'](ts)")
        t_embed = rays_t if _rays is not None else torch.repeat_interleave(rays_t, S, dim=0)
    out = model.evaluate(spec, packed, xyz=xyz, rays=_rays, z=None if _rays is None else z_vals, dirs=dirs,
                         t_embed=t_embed).view(R, S, spec.out_channels)
    if S == 1:
        raise NotImplementedError("single-sample pass 2 is undefined in the reference (SURVEY quirk 4)")
    alphas, transparency, weights, depth, acc = Fn.composite(z_vals, out, noise_arg, args.noise_std)
    return shade(model, args, spec, out, z_vals, alphas, transparency, weights, depth, acc, rays_d, sun_d, apply_brdf,
                 cos_irra_on, sort_idx, z_vals_unsort, sun_res, rows, cols)


def ref_sphere(rows, cols, R, S, like):
    """models/spsbrdfnerf.py:404-412 (validation visualisation of the view sphere).  The reference tiles rows / cols (R, 1) S
    times along the RAY axis and reads them back as (R, S)[:, 0]: ray r gets element (r * S) mod R of the input - reproduced as is."""
    sel = lambda t: t.reshape(-1).repeat(S).reshape(R, S)[:, 0]
    out = torch.ones(R, 1, 3, dtype=like.dtype, device=like.device)
    out[:, 0, 0] = sel(cols)
    out[:, 0, 1] = -sel(rows)
    out[:, 0, 2] = sel(torch.sqrt(torch.abs(1 - rows * rows - cols * cols)))
    return out


def shade(model, args, spec, out, z_vals, alphas, transparency, weights, depth, acc, rays_d, sun_d, apply_brdf,
          cos_irra_on, sort_idx=None, z_vals_unsort=None, sun_res=None, rows=None, cols=None):
    """Ray-level part of inference() (models/spsbrdfnerf.py:198-416) from the composited sums `acc` = sum_s w * out."""
    R, S = z_vals.shape
    nr_lr = spec.normal_lr
    albedo, sigmas = out[..., :3], out[..., 3]
    result = {"sigmas": sigmas.unsqueeze(-1), "albedo": albedo, "albedo_accu": acc[:, :3].clamp(0.0, 1.0), "depth": depth,
              "alphas": alphas, "weights": weights, "transparency": transparency, "z_vals": z_vals}
    sun_v = None                                 # per-sample sun visibility (R,S,1) of the sun pass (:148-151, :211-219)
    if sun_res and "sun" in sun_res:
        result["sun"], result["weights_sc"] = sun_res["sun"], sun_res["weights_sc"]
        if model.sun_v == "analystic":
            sun_v = sun_res["sun"]
    if sort_idx is not None:
        result["sort_idx"] = sort_idx
    if z_vals_unsort is not None:
        result["z_vals_unsort"] = z_vals_unsort
    if spec.ch_beta >= 0:                       # transient uncertainty per sample (:156-158, :225-226)
        result["beta"] = out[..., spec.ch_beta:spec.ch_beta + 1]
    normal = normal_s = None
    if spec.normal_an:
        c0 = spec.ch_normal_an
        result["normal_an"] = normal = out[..., c0:c0 + 3]
    if nr_lr:                                   # learned wins when both are present (spsbrdfnerf.py:234-239)
        c0 = spec.ch_normal_lr
        result["normal_lr"] = normal = out[..., c0:c0 + 3]
    if normal is not None:
        normal_s = l2_normalize(acc[:, c0:c0 + 3])
        view = -rays_d
        result["nr_vw"] = (normal_s * view).sum(-1).reshape(R, 1, 1)
        result["nr_sun"] = (normal_s * sun_d).sum(-1).reshape(R, 1, 1)
        result["hpk_scl"] = 1.0 / (args.hpk_scl * (result["nr_vw"] + result["nr_sun"]))
    pad = model.rgb_padding
    wsum = weights.sum(-1, keepdim=True)
    albedo_s = acc[:, :3] * (1 + 2 * pad) - pad * wsum          # sum_s w (albedo (1+2p) - p)   (:270,:275)
    irr_ray = None
    if cos_irra_on and normal is not None:
        irr_ray = sun_d[:, 2:3].abs()                            # upward normal (0,0,1): irradiance = |sun_z| (:260-264)
        sun_v = None                                             # the cosine branch wins (:260-266)
    rgb = albedo_s if irr_ray is None else albedo_s * irr_ray
    if sun_v is not None:                                        # per-sample irradiance: no composited shortcut (:265-273)
        rgb = (weights.unsqueeze(-1) * (albedo * (1 + 2 * pad) - pad) * sun_v).sum(-2)
    result["rgb"] = rgb.clamp(0.0, 1.0)
    heads = {}
    for (name, n_out, kind), (c0, wdt) in zip(spec.heads[1:], spec.head_cols[1:]):
        heads[name] = out[..., c0:c0 + wdt]
    if normal is None and not heads:
        return result, "Lambertian"
    if sun_v is not None:
        irr_ray = sun_v[:, -1, :]                                # per-ray BRDF: irradiance of the LAST sample (:354)

    brdf_type = "Lambertian"
    extra = {}
    view = -rays_d
    irr = torch.ones_like(albedo) if irr_ray is None else irr_ray[:, None, :].expand(R, S, 3)
    if sun_v is not None:
        irr = sun_v.expand(R, S, 3)
    shell = getattr(args, "shell_hapke", 0)
    kind = None
    if model.roughness and apply_brdf:
        kind = "Microfacet"
    elif model.RPV and apply_brdf:
        kind = "RPV"
    elif (apply_brdf and args.b == True) or shell > 0:  # noqa: E712
        kind = "Hapke"
    if kind is not None:
        if normal is None:
            raise RuntimeError("BRDF shading needs a normal field (--normal learned | analystic | analystic_learned)")
        brdf_type = kind
        if model.MultiBRDF:
            rep = lambda t: t.repeat_interleave(S, 0)
            flat = {k: v.reshape(R * S, -1) for k, v in heads.items()}
            brdf, aux = _per_ray_brdf(model, args, kind, rep(sun_d), rep(view), normal.reshape(-1, 3), albedo.reshape(-1, 3), flat)
            nb = S
        else:
            sums = {name: acc[:, c0:c0 + wdt] for (name, _, _), (c0, wdt) in zip(spec.heads[1:], spec.head_cols[1:])}
            brdf, aux = _per_ray_brdf(model, args, kind, sun_d, view, normal_s, albedo_s, sums)
            nb = 1
        if model.MultiBRDF:
            bp = brdf.reshape(R, S, 3) * (1 + 2 * pad) - pad
            rgb = (weights.unsqueeze(-1) * bp * irr).sum(-2)
        else:
            rgb = brdf if irr_ray is None else irr_ray * brdf        # irradiance of the last sample (:354)
        if apply_brdf:
            if kind == "Microfacet":
                extra = {"roughness": heads["roughness_from_xyz"], "glossy": aux[:, 0].reshape(R, nb, 1),
                         "brdf": brdf.reshape(R, nb, 3), "f": aux[:, 1].reshape(R, nb, 1), "g": aux[:, 2].reshape(R, nb, 1),
                         "d": aux[:, 3].reshape(R, nb, 1), "l_dot_n": aux[:, 4].reshape(R, nb, 1),
                         "v_dot_n": aux[:, 5].reshape(R, nb, 1), "halfvec": aux[:, 6:9].reshape(R, nb, 3),
                         "n_h": aux[:, 9].reshape(R, nb, 1)}
            elif kind == "RPV":
                for key, name in (("rpv_k", "k_from_xyz"), ("rpv_theta", "theta_rpv_from_xyz"), ("rpv_rhoc", "rhoc_from_xyz")):
                    if name in heads:
                        extra[key] = heads[name]
            else:
                extra = {"brdf": brdf.reshape(R, nb, 3), "hpk_P": aux[:, 0:3].reshape(R, nb, 3),
                         "hpk_Hi": aux[:, 3:6].reshape(R, nb, 3), "hpk_Hv": aux[:, 3:6].reshape(R, nb, 3),  # quirk 6
                         "hpk_ci": aux[:, 10].reshape(R, nb, 1), "hpk_cv": aux[:, 11].reshape(R, nb, 1),
                         "hpk_ShadFunc": aux[:, 9].reshape(R, nb, 1)}
                for key, name in (("hpk_b", "b_from_xyz"), ("hpk_c", "c_from_xyz"), ("hpk_theta", "theta_from_xyz")):
                    if name in heads:
                        extra[key] = heads[name]
    result["rgb"] = rgb.clamp(0.0, 1.0)
    result["irradiance"] = irr
    result.update(extra)
    if rays_d is not None:
        result["rays_d"] = view.reshape(R, 1, 3)
    if sun_d is not None:
        result["sun_d"] = sun_d.reshape(R, 1, 3)
    if rows is not None and cols is not None:
        result["ref_sphere"] = ref_sphere(rows.to(rays_d), cols.to(rays_d), R, S, rays_d)
    return result, brdf_type


def shade_ray(model, args, spec, z_vals, weights, depth, acc, rays_d, sun_d, apply_brdf, cos_irra_on):
    """The part of shade() a training loss reads when every ray has ONE BRDF and no per-sample irradiance (MultiBRDF == 0, no
    sun-visibility pass): rgb from the composited sums alone (models/spsbrdfnerf.py:259-357), without the per-sample entries
    of the result dict.  The torch-level statement of what bn_ray_shade_loss (shade_desc) computes in the launch-lean fused
    step, whose merged sample set is never materialised; tests differentiate it with autograd against that kernel."""
    normal_c0 = None
    if spec.normal_an:
        normal_c0 = spec.ch_normal_an
    if spec.normal_lr:                              # learned wins when both are present (spsbrdfnerf.py:234-239)
        normal_c0 = spec.ch_normal_lr
    pad = model.rgb_padding
    wsum = weights.sum(-1, keepdim=True)
    albedo_s = acc[:, :3] * (1 + 2 * pad) - pad * wsum
    irr_ray = sun_d[:, 2:3].abs() if (cos_irra_on and normal_c0 is not None) else None
    rgb = albedo_s if irr_ray is None else albedo_s * irr_ray
    heads = spec.heads[1:]
    if normal_c0 is None and not heads:
        return {"rgb": rgb.clamp(0.0, 1.0)}, "Lambertian"
    shell = getattr(args, "shell_hapke", 0)
    kind = None
    if model.roughness and apply_brdf:
        kind = "Microfacet"
    elif model.RPV and apply_brdf:
        kind = "RPV"
    elif (apply_brdf and args.b == True) or shell > 0:  # noqa: E712
        kind = "Hapke"
    if kind is None:
        return {"rgb": rgb.clamp(0.0, 1.0)}, "Lambertian"
    if normal_c0 is None:
        raise RuntimeError("BRDF shading needs a normal field (--normal learned | analystic | analystic_learned)")
    normal_s = l2_normalize(acc[:, normal_c0:normal_c0 + 3])
    sums = {name: acc[:, c0:c0 + wdt] for (name, _, _), (c0, wdt) in zip(spec.heads[1:], spec.head_cols[1:])}
    brdf, _ = _per_ray_brdf(model, args, kind, sun_d, -rays_d, normal_s, albedo_s, sums)
    rgb = brdf if irr_ray is None else irr_ray * brdf
    return {"rgb": rgb.clamp(0.0, 1.0)}, kind


def shade_desc(model, args, spec, apply_brdf, cos_irra_on, lambda_rgb=1.0, lambda_ds=0.0, lambda_hs=0.0, usealldepth=False, irr=None):
    """bn_shade_desc of shade_ray() for this model / spec: which BRDF (the same selection as shade(), models/spsbrdfnerf.py:
    277-357) and where its inputs sit among the composited channels."""
    from . import _lib as L
    cols = {name: c0 for (name, _, _), (c0, _) in zip(spec.heads[1:], spec.head_cols[1:])}
    ch_n = -1
    if spec.normal_an:
        ch_n = spec.ch_normal_an
    if spec.normal_lr:
        ch_n = spec.ch_normal_lr
    shell = int(getattr(args, "shell_hapke", 0))
    kind, p = L.BN_SHADE_LAMBERT, (-1, -1, -1)
    if ch_n >= 0 or cols:
        if model.roughness and apply_brdf:
            kind, p = L.BN_SHADE_MICROFACET, (cols["roughness_from_xyz"], -1, -1)
        elif model.RPV and apply_brdf:
            kind = L.BN_SHADE_RPV
            p = (cols.get("k_from_xyz", -1), cols.get("theta_rpv_from_xyz", -1),
                 -1 if args.funcH == 2 else cols.get("rhoc_from_xyz", -1))
        elif (apply_brdf and args.b == True) or shell > 0:  # noqa: E712
            kind = L.BN_SHADE_HAPKE
            p = (cols.get("b_from_xyz", -1), cols.get("c_from_xyz", -1), cols.get("theta_from_xyz", -1))
    if kind != L.BN_SHADE_LAMBERT and ch_n < 0:
        raise RuntimeError("BRDF shading needs a normal field (--normal learned | analystic | analystic_learned)")
    d = L.ShadeDesc()
    d.kind, d.C, d.ch_normal, (d.ch_p0, d.ch_p1, d.ch_p2) = kind, spec.out_channels, ch_n, p
    d.rhoc_is_albedo = int(kind == L.BN_SHADE_RPV and args.funcH == 2)
    d.shell, d.cos_irradiance, d.usealldepth = shell, int(bool(cos_irra_on)), int(bool(usealldepth))
    d.hpk_scl, d.f0 = float(getattr(args, "hpk_scl", 1.0)), float(getattr(args, "fresnel_f0", 0.04))
    d.rgb_padding, d.lambda_rgb, d.lambda_ds, d.lambda_hs = float(model.rgb_padding), float(lambda_rgb), float(lambda_ds), float(lambda_hs)
    if irr is not None:          # per-ray irradiance of the sun pass (1-d float32 view)
        assert irr.is_cuda and irr.dtype == torch.float32 and irr.dim() == 1
        d.irr, d.irr_stride, d._keep = irr.data_ptr(), (irr.stride(0) if irr.shape[0] > 1 else 1), irr
    return d


def render_rays(models, args, rays, ts, mode="test", valid_depth=None, target_depths=None, target_std=None,
                apply_brdf=False, print_debuginfo=False, bTestNormal=False, bTestSun_v=False, gsam_only=False, rows=None,
                cols=None, percent=0, apply_theta=False, cos_irra_on=False):
    """render_rays, spsbrdf-nerf branch (rendering.py:168-291)."""
    if args.model != "spsbrdf-nerf":
        raise ValueError("brdf_nerf_amd.render_rays serves --model spsbrdf-nerf only")
    if args.n_importance > 0:
        # rendering.py:294-332 calls `inference(models['fine'], ...)` through its generic `else:` branch for this model and then
        # iterates `result.keys()` - but spsbrdfnerf.inference returns a (dict, brdf_type) TUPLE: the reference raises
        # AttributeError("'tuple' object has no attribute 'keys'") for every spsbrdf-nerf run with --n_importance > 0
        # (probed in the build container).  There is no upstream behaviour to reproduce.
        raise NotImplementedError("n_importance > 0: the reference itself raises AttributeError in its fine pass for --model "
                                  "spsbrdf-nerf (rendering.py:327-330: spsbrdfnerf.inference returns a tuple); nothing to mirror")
    model = models["coarse"]
    G, S = args.guided_samples, args.n_samples
    if G <= 0:
        raise NotImplementedError("guided_samples <= 0 returns an un-suffixed dict upstream (SURVEY quirk 1)")
    if G == 2:
        raise NotImplementedError("guided_samples == 2 (single mean sample) hits SURVEY quirk 4 upstream")
    rays = rays.float().contiguous()
    R = rays.shape[0]
    near, far = rays[:, 6:7], rays[:, 7:8]
    rays_d = rays[:, 3:6]
    sun_d = rays[:, 8:11] if args.data == "sat" else torch.ones_like(rays[:, 0:3])

    nr_lr = model.normal in ("analystic_learned", "learned")
    nr_an = model.normal in ("analystic_learned", "analystic") or bTestNormal
    spec = model.spec(apply_brdf, apply_theta, nr_lr, nr_an)
    packed = model.repack(spec)
    rays_t = None
    if model.beta:                              # rendering.py:226-229
        if ts is None or "t" not in models:
            raise ValueError("--beta: render_rays needs ts and models['t'] (the image embedding, main.py:113-118)")
        rays_t = models["t"](ts)

    z_vals = get_z_vals(S, rays.device, near, far)
    C = spec.out_channels
    noise_on = args.noise_std != 0
    if gsam_only:
        # only the guided samples are rendered: pass 1 just guides them (sigma only, no autograd)
        res1, _ = inference(model, args, None, z_vals, rays_d=rays_d, sun_d=sun_d, mode=mode, sigma_only=True, _rays=rays,
                            _packed=packed)
        w1, d1, out1 = res1["weights"], res1["depth"], None
    else:
        # The field is a pointwise function of xyz: the S coarse samples are evaluated ONCE, in full, and reused in the
        # merged pass-2 set (the reference evaluates them twice - sigma only, then again among the S+G samples - with
        # the same values).  Pass-1 compositing stays detached, as upstream (rendering.py:262).
        noise1 = torch.randn(R, S, device=rays.device)
        out1 = model.evaluate(spec, packed, rays=rays, z=z_vals, t_embed=rays_t).view(R, S, C)
        with torch.no_grad():
            _, _, w1, d1, _ = Fn.composite_forward_raw(z_vals, out1.detach(), noise1 if noise_on else None, args.noise_std)
    sun_res = {}
    if (model.sun_v == "analystic" and apply_brdf) or bTestSun_v:
        # Sun-visibility pass (rendering.py:244-259): transparency along the sun direction from the pass-1 surface
        # point, sigma only, detached.  far_sun is scaled with ROW 0's directions, as upstream (:247-248).
        if not gsam_only:
            raise NotImplementedError("--sun_v analystic needs gsam_only=True: with the merged S+G sample set the reference "
                                      "raises a shape error in pass 2 (SURVEY quirk 2)")
        with torch.no_grad():
            far_sun = sun_far(d1, rays_d, sun_d)
            z_sun = get_z_vals(G, rays.device, far_sun * 0.01, far_sun)
            sun_rays = torch.cat([rays[:, 0:3] + rays_d * d1.unsqueeze(-1), sun_d], -1).contiguous()
            rs, _ = inference(model, args, None, z_sun, rays_d=sun_d, mode=mode, sigma_only=True, _rays=sun_rays, _packed=packed)
        sun_res = {"sun": rs["transparency"].unsqueeze(-1).detach(), "weights_sc": rs["weights"].detach()}
    # guided samples around the pass-1 depth (or the ground-truth depth prior in training)
    u = torch.rand(R, G, device=rays.device)
    use_t = tdep = tstd = u_t = trow = None
    if mode == "train" and valid_depth is not None:
        valid = (valid_depth > 0)
        # the reference draws rand(n_valid, G): the SHAPE of that draw is data dependent, so replaying its random stream
        # needs n_valid on the host (one sync; the reference does three via np.where(...cpu())).  The fused training step
        # draws (R, G) and indexes it by the valid-row rank instead: same distribution, no sync (trainer.py).
        n_valid = int(valid.sum())
        u_t = torch.rand(n_valid, G, device=rays.device)        # drawn (an empty tensor) even without a valid row, as upstream (:76-91)
        if n_valid > 0:
            use_t = valid.float().contiguous()
            tdep = target_depths[:, 0].float().contiguous()
            tstd = target_std.float().reshape(-1).contiguous()
            trow = (torch.cumsum(valid.int(), 0) - 1).clamp_min(0).int().contiguous()
    with torch.no_grad():
        # the clamp window is the FIRST ray's (near, far) (rendering.py:133,144): read by the kernel from rays[0, 6:8]
        z2, z_all, idx = Fn.guided_samples(z_vals, w1, d1, u, rays[0, 6:8], None, args.std_range, use_t, tdep,
                                           tstd, u_t, trow, merge=not gsam_only)
    if gsam_only:
        result, brdf_type = inference(model, args, None, z2, rays_d=rays_d, sun_d=sun_d, rays_t=rays_t, z_vals_unsort=z2, apply_brdf=apply_brdf,
                                      bTestNormal=bTestNormal, sun_res=sun_res, sort_idx=None, mode=mode, apply_theta=apply_theta,
                                      cos_irra_on=cos_irra_on, _rays=rays, _packed=packed, rows=rows, cols=cols)
        return {f"{k}_coarse": v for k, v in result.items()}, brdf_type
    z_unsort = torch.cat([z_vals, z2], -1)
    out2 = model.evaluate(spec, packed, rays=rays, z=z2, t_embed=rays_t).view(R, G, C)
    out = torch.cat([out1, out2], 1).gather(1, idx.unsqueeze(-1).expand(-1, -1, C))      # depth-sorted order
    noise2 = torch.randn(R, S + G, device=rays.device)
    alphas, transparency, weights, depth, acc = Fn.composite(z_all, out, noise2 if noise_on else None, args.noise_std)
    result, brdf_type = shade(model, args, spec, out, z_all, alphas, transparency, weights, depth, acc, rays_d, sun_d, apply_brdf,
                              cos_irra_on, idx, z_unsort, rows=rows, cols=cols)
    return {f"{k}_coarse": v for k, v in result.items()}, brdf_type
