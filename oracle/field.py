"""Oracle: the SpS-BRDF-NeRF field MLP, restated functionally (PyTorch CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.  `params` is a dict of tensors
keyed by the reference's state_dict names.

Follows (path:line under /root/reference):
  Mapping.forward            models/nerf.py:53-70
  Siren                      models/nerf.py:23-33
  calc_features              models/spsbrdfnerf.py:636-646
  calc_normals               models/spsbrdfnerf.py:648-660
  SpSBRDFNeRF.forward        models/spsbrdfnerf.py:662-757
  l2_normalize               train_utils.py:28-33
"""
import math
import torch

FP32_EPS = float(torch.finfo(torch.float32).eps)


def l2_normalize(x):
    """train_utils.py:28-33: x / sqrt(max(sum x^2, fp32 eps))."""
    n2 = (x * x).sum(-1, keepdim=True)
    return x / torch.sqrt(torch.clamp_min(n2, FP32_EPS))


def positional_encoding(x, n_freqs):
    """nerf.py:53-70: [sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]; no raw x."""
    parts = []
    for k in range(n_freqs):
        f = float(2 ** k)
        parts.append(torch.sin(f * x))
        parts.append(torch.cos(f * x))
    return torch.cat(parts, -1)


def _act(cfg, z, w0):
    return torch.sin(w0 * z) if cfg.siren else torch.relu(z)


def trunk(params, cfg, xyz):
    """calc_features (spsbrdfnerf.py:636-646): PE + `layers` dense layers, skip concat [PE, h]."""
    pe = positional_encoding(xyz, cfg.pe_freqs) if cfg.mapping else xyz
    h = pe
    for i in range(cfg.layers):
        if i in cfg.skips:
            h = torch.cat([pe, h], -1)
        z = torch.nn.functional.linear(h, params[f"fc_net.{2*i}.weight"], params[f"fc_net.{2*i}.bias"])
        h = _act(cfg, z, 30.0 if i == 0 else 1.0)
    return h


def sigma_of(params, cfg, xyz):
    h = trunk(params, cfg, xyz)
    return torch.nn.functional.softplus(
        torch.nn.functional.linear(h, params["sigma_from_xyz.0.weight"], params["sigma_from_xyz.0.bias"]))


def sigma_grad(params, cfg, xyz, create_graph=True):
    """calc_normals (spsbrdfnerf.py:648-660): d sigma / d xyz by autograd."""
    with torch.enable_grad():
        x = xyz if xyz.requires_grad else xyz.detach().requires_grad_(True)
        s = sigma_of(params, cfg, x)
        (g,) = torch.autograd.grad(s, x, torch.ones_like(s), create_graph=create_graph,
                                   retain_graph=create_graph)
    return g


def _head(params, cfg, name, feats):
    g = _act(cfg, torch.nn.functional.linear(feats, params[f"{name}.0.weight"], params[f"{name}.0.bias"]), 1.0)
    return torch.sigmoid(torch.nn.functional.linear(g, params[f"{name}.2.weight"], params[f"{name}.2.bias"]))


def _tile3(v):
    return v.repeat(1, 3) if v.shape[1] == 1 else v


def field_forward(params, cfg, xyz, sigma_only=False, apply_brdf=False, apply_theta=False,
                  nr_an_on=False, nr_lr_on=False, dirs=None, t_embed=None):
    """SpSBRDFNeRF.forward (spsbrdfnerf.py:662-757) for sun_v='none', indirect_light=False.
    `t_embed` (B, cfg.t_dim): per-point image embedding, used only with cfg.beta (spsbrdfnerf.py:708-711: one more output
    channel beta = beta_from_xyz(cat([xyz_features, input_t])) right after sigma).
    `dirs` (B,3): per-point view directions, used only with cfg.input_viewdir == 1 (spsbrdfnerf.py:689-692: the rgb head
    reads cat([xyz_features, mapping[1](input_dir)])).

    Channel order: [rgb3, sigma1, (beta1), (normal_an3), (normal_lr3), (rough1 | k3,theta3,rhoc3 | b3,c3,theta1)].
    """
    h = trunk(params, cfg, xyz)
    sigma = torch.nn.functional.softplus(
        torch.nn.functional.linear(h, params["sigma_from_xyz.0.weight"], params["sigma_from_xyz.0.bias"]))
    if sigma_only:
        return sigma
    feats = torch.nn.functional.linear(h, params["feats_from_xyz.weight"], params["feats_from_xyz.bias"])
    rgb_in = feats
    if cfg.dir_dim:
        rgb_in = torch.cat([feats, positional_encoding(dirs, cfg.dir_freqs) if cfg.mapping else dirs], -1)
    rgb = _head(params, cfg, "rgb_from_xyzdir", rgb_in)
    out = [rgb, sigma]
    if cfg.beta:
        g = _act(cfg, torch.nn.functional.linear(torch.cat([feats, t_embed], -1), params["beta_from_xyz.0.weight"],
                                                 params["beta_from_xyz.0.bias"]), 1.0)
        out.append(torch.nn.functional.softplus(
            torch.nn.functional.linear(g, params["beta_from_xyz.2.weight"], params["beta_from_xyz.2.bias"])))
    if nr_an_on:
        out.append(-l2_normalize(sigma_grad(params, cfg, xyz, create_graph=True)))
    if nr_lr_on:
        g = torch.nn.functional.linear(h, params["grad_from_xyz.weight"], params["grad_from_xyz.bias"])
        out.append(-l2_normalize(g))
    for name in cfg.brdf_head_names(apply_brdf, apply_theta):
        v = _head(params, cfg, name, feats)
        if name == "k_from_xyz":
            v = _tile3((v - 0.5) * 2 + 1)            # [0, 2]      :730
        elif name == "theta_rpv_from_xyz":
            v = _tile3((v - 0.5) * 2)                # [-1, 1]     :735
        elif name in ("rhoc_from_xyz", "b_from_xyz", "c_from_xyz"):
            v = _tile3(v)
        elif name == "theta_from_xyz":
            v = v * (math.pi * 30.0 / 180.0)         # [0, 30 deg] :754
        out.append(v)
    return torch.cat(out, 1)


def sigma_grad_closed_form(params, cfg, xyz):
    """Closed-form adjoint chain for d sigma/d xyz; used to
    cross-check the autograd version and as the spec for the HIP adjoint kernel.  Siren + mapping only."""
    assert cfg.siren and cfg.mapping
    pe = positional_encoding(xyz, cfg.pe_freqs)
    hs, zs = [], []
    h = pe
    for i in range(cfg.layers):
        if i in cfg.skips:
            h = torch.cat([pe, h], -1)
        hs.append(h)
        z = torch.nn.functional.linear(h, params[f"fc_net.{2*i}.weight"], params[f"fc_net.{2*i}.bias"])
        zs.append(z)
        h = torch.sin((30.0 if i == 0 else 1.0) * z)
    s_raw = torch.nn.functional.linear(h, params["sigma_from_xyz.0.weight"], params["sigma_from_xyz.0.bias"])
    a = torch.sigmoid(s_raw) * params["sigma_from_xyz.0.weight"]            # d softplus = sigmoid
    P = pe.shape[1]
    g_pe = torch.zeros_like(pe)
    for i in reversed(range(cfg.layers)):
        w0 = 30.0 if i == 0 else 1.0
        delta = a * (w0 * torch.cos(w0 * zs[i]))
        back = delta @ params[f"fc_net.{2*i}.weight"]
        if i in cfg.skips:
            g_pe = g_pe + back[:, :P]
            a = back[:, P:]
        elif i == 0:
            g_pe = g_pe + back
        else:
            a = back
    grad = torch.zeros_like(xyz)
    for k in range(cfg.pe_freqs):
        f = float(2 ** k)
        gs = g_pe[:, 6 * k:6 * k + 3]
        gc = g_pe[:, 6 * k + 3:6 * k + 6]
        grad = grad + f * torch.cos(f * xyz) * gs - f * torch.sin(f * xyz) * gc
    return grad
