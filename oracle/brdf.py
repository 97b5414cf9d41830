"""Oracle: closed-form BRDFs (RPV, Hapke, GGX microfacet), PyTorch CPU.

TEST INFRASTRUCTURE - see oracle/__init__.py.

Follows (path:line under /root/reference):
  calc_angles, Henyey_Greenstein   BRDF/basic_func.py:5-44
  func_M1, func_G, func_H, calc_rpv BRDF/RPV.py:6-63
  E1,E2,f,chi,eta,mu0_eff,mu_eff,S,PF,HF,hapkeHG_6var  BRDF/Hapke.py:6-200
  Microfacet.forward,_get_g,_get_d,_get_f  BRDF/microfacet.py:20-118
NaN handling: the reference's check_nan(val_rep=...) (train_utils.py:61-78) replaces NaNs
elementwise; restated here as where(isnan, rep, y) without the prints/syncs.
"""
import math
import torch


def _nan_to(y, rep):
    return torch.where(torch.isnan(y), rep, y)


def calc_angles(l, v, n, eps=1e-5):
    """l,v,n: (N,3).  Returns ci, sza, si, cv, vza, sv, cg, g, phi (each (N,))."""
    ci = (l * n).sum(-1).clamp(eps, 1.0)
    sza = torch.acos(ci)
    si = torch.sin(sza)
    cv = (v * n).sum(-1).clamp(eps, 1.0)
    vza = torch.acos(cv)
    sv = torch.sin(vza)
    cg = (v * l).sum(-1).clamp(-1.0, 1.0)
    g = torch.acos(cg)
    cp = ((cg - ci * cv) / si / sv).clamp(-1.0, 1.0)
    phi = torch.acos(cp)
    return ci, sza, si, cv, vza, sv, cg, g, phi


def henyey_greenstein(x, theta, eps=1e-6):
    """x: (N,1) cos(phase), theta: (N,3)."""
    t2 = theta * theta
    y = (1 - t2) / (torch.pow(1 + 2 * theta * x + t2, 1.5) + eps)
    return _nan_to(y, torch.zeros_like(y))


# ----------------------------------------------------------------------------- RPV
def rpv(l, v, n, w, k=None, theta=None, rhoc=None, eps=1e-5):
    """RPV.calc_rpv.  l,v,n,w,(k,theta,rhoc): (N,3).  Returns brdf (N,3), M1, G, H, ci, cv."""
    ci, sza, si, cv, vza, sv, cg, g, phi = calc_angles(l, v, n)
    if k is not None:
        base = (ci * cv * (ci + cv) + eps).unsqueeze(-1)
        M1 = torch.pow(base, k - 1)
        M1 = _nan_to(M1, torch.zeros_like(M1))
    else:
        M1 = torch.ones_like(ci).unsqueeze(-1)
    F = henyey_greenstein(cg.unsqueeze(-1), theta) if theta is not None else torch.ones_like(cg).unsqueeze(-1)
    if rhoc is not None:
        ti, tv, cp = torch.tan(sza), torch.tan(vza), torch.cos(phi)
        G = torch.sqrt(ti ** 2 + tv ** 2 - 2 * ti * tv * cp + eps)
        G = _nan_to(G, torch.zeros_like(G)).unsqueeze(-1)
        H = 1 + (1 - rhoc) / (1 + G.detach() + eps)          # G detached: RPV.py:55
        H = _nan_to(H, torch.zeros_like(H))
    else:
        G = torch.ones_like(sza).unsqueeze(-1)
        H = torch.ones_like(sza).unsqueeze(-1)
    return w * M1 * F * H, M1, G, H, ci, cv


# --------------------------------------------------------------------------- Hapke
def _E1(x, th, eps=1e-5):
    y = torch.exp(-(2.0 / math.pi) / torch.tan(th + eps) / torch.tan(x + eps))
    return _nan_to(y, torch.zeros_like(y))


def _E2(x, th, eps=1e-5):
    y = torch.exp(-(1.0 / math.pi) * (1.0 / torch.tan(th + eps)) ** 2 * (1.0 / torch.tan(x + eps)) ** 2)
    return _nan_to(y, torch.zeros_like(y))


def _f(phi, eps=1e-5):
    y = torch.exp(-2.0 * torch.tan((phi + eps) / 2))
    return _nan_to(y, torch.zeros_like(y))


def _chi(x, eps=1e-5):
    y = 1.0 / torch.sqrt(1.0 + math.pi * torch.tan(x + eps) ** 2)
    return _nan_to(y, torch.zeros_like(y))


def _eta(x, th, eps=1e-5):
    y = _chi(th) * (torch.cos(x) + torch.sin(x) * torch.tan(th + eps) * (_E2(x, th) / (2 - _E1(x, th))))
    return _nan_to(y, torch.zeros_like(y))


def _branch(i, e, fn_le, fn_gt):
    """Evaluate fn_le on rows with i<=e and fn_gt on the others (Hapke.py:35-44 index lists)."""
    y = torch.zeros_like(e)
    m1 = i <= e
    m2 = ~m1
    if m1.any():
        y = y.masked_scatter(m1, fn_le(m1))
    if m2.any():
        y = y.masked_scatter(m2, fn_gt(m2))
    return y


def _mu0_eff(i, e, phi, th):
    def le(m):
        ii, ee, pp, tt = i[m], e[m], phi[m], th[m]
        y = torch.cos(pp) * _E2(ee, tt) + torch.sin(pp / 2) ** 2 * _E2(ii, tt)
        y = y / (2 - _E1(ee, tt) - pp / math.pi * _E1(ii, tt))
        return _chi(tt) * (torch.cos(ii) + torch.sin(ii) * torch.tan(tt) * y)

    def gt(m):
        ii, ee, pp, tt = i[m], e[m], phi[m], th[m]
        y = _E2(ii, tt) - torch.sin(pp / 2) ** 2 * _E2(ee, tt)
        y = y / (2 - _E1(ii, tt) - pp / math.pi * _E1(ee, tt))
        return _chi(tt) * (torch.cos(ii) + torch.sin(ii) * torch.tan(tt) * y)

    return _nan_to(_branch(i, e, le, gt), torch.cos(i))


def _mu_eff(i, e, phi, th):
    def le(m):
        ii, ee, pp, tt = i[m], e[m], phi[m], th[m]
        y = _E2(ee, tt) - torch.sin(pp / 2) ** 2 * _E2(ii, tt)
        y = y / (2 - _E1(ee, tt) - pp / math.pi * _E1(ii, tt))
        return _chi(tt) * (torch.cos(ee) + torch.sin(ee) * torch.tan(tt) * y)

    def gt(m):
        ii, ee, pp, tt = i[m], e[m], phi[m], th[m]
        y = torch.cos(pp) * _E2(ii, tt) + torch.sin(pp / 2) ** 2 * _E2(ee, tt)
        y = y / (2 - _E1(ii, tt) - pp / math.pi * _E1(ee, tt))
        return _chi(tt) * (torch.cos(ee) + torch.sin(ee) * torch.tan(tt) * y)

    return _nan_to(_branch(i, e, le, gt), torch.cos(e))


def _shadow(i, e, phi, th):
    ci, cv = torch.cos(i), torch.cos(e)
    mue = _mu_eff(i, e, phi, th)
    etai, etae, chit, ff = _eta(i, th), _eta(e, th), _chi(th), _f(phi)
    temp = (mue / etae) * (ci / etai) * chit
    y = _branch(i, e,
                lambda m: temp[m] / (1 - ff[m] + ff[m] * chit[m] * (ci[m] / etai[m])),
                lambda m: temp[m] / (1 - ff[m] + ff[m] * chit[m] * (cv[m] / etae[m])))
    return _nan_to(y, torch.zeros_like(y))


def _PF(x, b, c):
    b2, bx = b * b, b * x
    y = c * (1 - b2) / (torch.pow(1 - 2 * bx + b2, 1.5) + 1e-6)
    y = y + (1 - c) * (1 - b2) / (torch.pow(1 + 2 * bx + b2, 1.5) + 1e-6)
    return _nan_to(y, torch.zeros_like(y))


def _HF(x, w):
    gamma = torch.sqrt(1 - w)
    ro = (1 - gamma) / (1 + gamma)
    lg = torch.log(torch.abs((1 + x) / x))
    y = torch.pow(1 - w * x * (ro + (1 - 2 * ro * x) / 2 * lg), -1)
    return _nan_to(y, torch.ones_like(y))


def hapke(l, v, n, w, b=None, c=None, theta=None, hpk_scl=4.0, shell_hapke=0):
    """Hapke.hapkeHG_6var with B0=h=None.  l,v,n,w,(b,c): (N,3); theta: (N,).
    Returns brdf (N,3), P, B, Hi, Hv, S (N,1), ci, cv (N,)."""
    ci, sza, si, cv, vza, sv, cg, g, phi = calc_angles(l, v, n)
    if b is None:
        P = torch.ones_like(cg).unsqueeze(-1).repeat(1, 3)
    elif c is None:
        P = henyey_greenstein(cg.unsqueeze(-1), b)
    else:
        P = _PF(cg.unsqueeze(-1), b, c)
    B = torch.ones_like(g).unsqueeze(-1)
    if theta is not None:
        ci = _mu0_eff(sza, vza, phi, theta)
        cv = _mu_eff(sza, vza, phi, theta)
        S = _shadow(sza, vza, phi, theta).unsqueeze(-1)
    else:
        S = torch.ones_like(sza).unsqueeze(-1)
    Hi = _HF(ci.unsqueeze(-1), w)
    Hv = _HF(cv.unsqueeze(-1), w)
    if b is None:
        if shell_hapke == 1:
            brdf = w / hpk_scl
        elif shell_hapke == 2:
            brdf = w / ((ci + cv) * hpk_scl + 1e-6).unsqueeze(-1)
        elif shell_hapke == 3:
            brdf = w * (Hi * Hv) / ((ci + cv) * hpk_scl + 1e-6).unsqueeze(-1)
        else:
            raise ValueError("hapke: b is None needs shell_hapke in {1,2,3}")
    else:
        t1 = (ci / (ci + cv) / torch.cos(sza)).unsqueeze(-1)
        brdf = w / hpk_scl * t1 * (P * B + Hi * Hv - 1) * S
    return brdf, P, B, Hi, Hv, S, ci, cv


# ---------------------------------------------------------------------- microfacet
def _safe_norm(x, eps=1e-6):
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


def microfacet(l, v, n, albedo, rough, f0=0.04):
    """Microfacet.forward with L=1, lvis=False.  l,v,n,albedo: (N,3); rough: (N,1).
    Returns glossy (N,1), brdf (N,3), f, g, d, l_dot_n (N,1), v_dot_n (N,), h (N,3), n_h (N,1)."""
    l, v, n = _safe_norm(l), _safe_norm(v), _safe_norm(n)
    h = _safe_norm(l + v)
    f = f0 + (1 - f0) * (1 - (l * h).sum(-1, keepdim=True)) ** 5
    alpha = rough ** 2
    # _get_d
    cos_m = (h * n).sum(-1, keepdim=True)
    chi = (cos_m > 0).to(cos_m.dtype)
    cm2 = cos_m ** 2
    tan2 = torch.nan_to_num((1 - cm2) / cm2)
    d = torch.nan_to_num(alpha ** 2 * chi / (math.pi * cm2 ** 2 * (alpha ** 2 + tan2) ** 2))
    # _get_g (returned for visualisation only)
    cos_v = (n * v).sum(-1)
    div = torch.nan_to_num((h * v).sum(-1, keepdim=True) / cos_v.unsqueeze(1))
    chig = (div > 0).to(div.dtype)
    cv2 = (cos_v ** 2).clamp(0.0, 1.0)
    tv2 = torch.nan_to_num(torch.nan_to_num((1 - cv2) / cv2).clamp(0.0, math.inf))
    g = torch.nan_to_num(chig * 2 / (1 + torch.sqrt(1 + alpha ** 2 * tv2.unsqueeze(1))))
    l_dot_n = (l * n).sum(-1, keepdim=True).abs().clamp_min(0.001)
    v_dot_n = (v * n).sum(-1).abs().clamp_min(0.001)
    glossy = torch.nan_to_num(0.04 * d / (4 * l_dot_n * v_dot_n.unsqueeze(1)))
    brdf = albedo + glossy
    return glossy, brdf, f, g, d, l_dot_n, v_dot_n, h, cos_m
