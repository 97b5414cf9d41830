"""Field configuration + deterministic parameter generation for the oracle.

Parameter names/shapes follow the reference's state_dict contract
(models/spsbrdfnerf.py:514-613; SURVEY.md section 5 "Checkpoint / resume").
Init ranges follow models/nerf.py:9-21 (Siren inits) and torch.nn.Linear's
default (U(+-1/sqrt(fan_in)) for weight and bias).
"""
from dataclasses import dataclass
import math
import numpy as np


@dataclass
class FieldConfig:
    feat: int = 512
    layers: int = 8
    skips: tuple = (4,)
    siren: bool = True
    mapping: bool = True
    pe_freqs: int = 10                 # mapping_sizes[0], spsbrdfnerf.py:445
    normal: str = "none"               # none | analystic | learned | analystic_learned
    roughness: bool = False            # microfacet head
    funcM: int = 0                     # RPV k head
    funcF: int = 0                     # RPV theta head
    funcH: int = 0                     # RPV rhoc head (2 = use albedo, no head)
    dim_RPV: int = 1
    b: int = 0                         # Hapke b head
    c: int = 0                         # Hapke c head
    theta: int = 0                     # Hapke theta head
    shell_hapke: int = 0
    hpk_scl: float = 4.0
    fresnel_f0: float = 0.04
    MultiBRDF: bool = False
    rgb_padding: float = 0.001         # spsbrdfnerf.py:459
    # render-level knobs (opt.py defaults)
    n_samples: int = 64
    guided_samples: int = 64
    std_range: float = 3.0
    noise_std: float = 0.0
    data: str = "sat"
    sun_v: str = "none"                # none | analystic (sun-visibility pass, rendering.py:244-259; no parameters)
    input_viewdir: int = 0             # 1: the encoded view direction joins the rgb head's input (spsbrdfnerf.py:458,689-692)
    dir_freqs: int = 4                 # mapping_sizes[1], spsbrdfnerf.py:445
    beta: bool = False                 # --beta: transient-uncertainty head on cat([xyz_features, t embedding]) (spsbrdfnerf.py:571-575,708-711)
    t_dim: int = 4                     # --t_embbeding_tau (opt.py:199), width of the per-image embedding

    @property
    def RPV(self):
        return bool(self.funcM == 1 or self.funcF == 1 or self.funcH == 1)

    @property
    def in_dim(self):
        return 2 * self.pe_freqs * 3 if self.mapping else 3

    @property
    def dir_dim(self):
        """in_size[1] (spsbrdfnerf.py:506-510): 0 without --input_viewdir, else the (encoded) direction width."""
        if self.input_viewdir != 1:
            return 0
        return 2 * self.dir_freqs * 3 if self.mapping else 3

    def param_shapes(self):
        """Ordered [(state_dict key, shape, init kind)] - registration order of the reference."""
        F, P = self.feat, self.in_dim
        out = []
        for i in range(self.layers):
            k = P if i == 0 else (F + P if i in self.skips else F)
            kind = ("siren0" if i == 0 else "siren") if self.siren else "linear"
            out.append((f"fc_net.{2*i}.weight", (F, k), kind))
            out.append((f"fc_net.{2*i}.bias", (F,), "bias%d" % k))
        out.append(("sigma_from_xyz.0.weight", (1, F), "linear"))
        out.append(("sigma_from_xyz.0.bias", (1,), "bias%d" % F))
        out.append(("feats_from_xyz.weight", (F, F), "linear"))
        out.append(("feats_from_xyz.bias", (F,), "bias%d" % F))
        rgb = self._head("rgb_from_xyzdir", 3)
        if self.dir_dim:      # Linear(feat + in_size[1], feat // 2), spsbrdfnerf.py:534
            rgb[0] = ("rgb_from_xyzdir.0.weight", (F // 2, F + self.dir_dim), "linear")
            rgb[1] = ("rgb_from_xyzdir.0.bias", (F // 2,), "bias%d" % (F + self.dir_dim))
        out += rgb
        if self.beta:         # Linear(t_embedding_dims + feat, feat // 2), nl, Linear(feat // 2, 1), Softplus  (spsbrdfnerf.py:571-574)
            H2, k = F // 2, self.t_dim + F
            out += [("beta_from_xyz.0.weight", (H2, k), "linear"), ("beta_from_xyz.0.bias", (H2,), "bias%d" % k),
                    ("beta_from_xyz.2.weight", (1, H2), "linear"), ("beta_from_xyz.2.bias", (1,), "bias%d" % H2)]
        if self.normal in ("learned", "analystic_learned"):
            out.append(("grad_from_xyz.weight", (3, F), "linear"))
            out.append(("grad_from_xyz.bias", (3,), "bias%d" % F))
        if self.roughness:
            out += self._head("roughness_from_xyz", 1)
        if self.funcM == 1:
            out += self._head("k_from_xyz", self.dim_RPV)
        if self.funcF == 1:
            out += self._head("theta_rpv_from_xyz", self.dim_RPV)
        if self.funcH == 1:
            out += self._head("rhoc_from_xyz", self.dim_RPV)
        if self.b == 1:
            out += self._head("b_from_xyz", 1)
        if self.c == 1:
            out += self._head("c_from_xyz", 1)
        if self.theta == 1:
            out += self._head("theta_from_xyz", 1)
        return out

    def _head(self, name, n_out):
        F = self.feat
        return [(f"{name}.0.weight", (F // 2, F), "linear"), (f"{name}.0.bias", (F // 2,), "bias%d" % F),
                (f"{name}.2.weight", (n_out, F // 2), "linear"), (f"{name}.2.bias", (n_out,), "bias%d" % (F // 2))]

    def brdf_head_names(self, apply_brdf, apply_theta):
        """Heads evaluated by forward() in channel order (spsbrdfnerf.py:722-755)."""
        if not apply_brdf:
            return []
        if self.roughness:
            return ["roughness_from_xyz"]
        if self.RPV:
            names = []
            if self.funcM == 1:
                names.append("k_from_xyz")
            if self.funcF == 1:
                names.append("theta_rpv_from_xyz")
            if self.funcH == 1:
                names.append("rhoc_from_xyz")
            return names
        names = []
        if self.b == 1:
            names.append("b_from_xyz")
        if self.c == 1:
            names.append("c_from_xyz")
        if apply_theta and self.theta == 1:
            names.append("theta_from_xyz")
        return names

    def out_channels(self, apply_brdf=False, apply_theta=False, bTestNormal=False):
        """Channel count of forward() (inference(): spsbrdfnerf.py:104-115)."""
        c = 5 if self.beta else 4
        if self.normal in ("analystic", "analystic_learned") or bTestNormal:
            c += 3
        if self.normal in ("learned", "analystic_learned"):
            c += 3
        for h in self.brdf_head_names(apply_brdf, apply_theta):
            c += 1 if h in ("roughness_from_xyz", "theta_from_xyz") else 3
        return c

    def make_params(self, seed=0, dtype=np.float32):
        """Deterministic parameters (numpy PCG64) with the reference's init ranges."""
        rng = np.random.default_rng(seed)
        params = {}
        for name, shape, kind in self.param_shapes():
            if kind == "siren0":
                lim = 1.0 / shape[1]
            elif kind == "siren":
                lim = math.sqrt(6.0 / shape[1])
            elif kind == "linear":
                lim = 1.0 / math.sqrt(shape[1])
            else:  # bias<fan_in>
                lim = 1.0 / math.sqrt(int(kind[4:]))
            params[name] = rng.uniform(-lim, lim, size=shape).astype(dtype)
        return params

    def n_params(self):
        return sum(int(np.prod(s)) for _, s, _ in self.param_shapes())
