"""brdf_nerf_amd - MI355X (gfx950) native implementation of BRDF-NeRF's ray-batched volume-rendering
hot path (spsbrdf-nerf): fused field MLP (forward/backward), alpha compositing, depth-guided
resampling and RPV / Hapke / GGX BRDF shading as hand-written HIP kernels behind a C ABI
(include/brdfnerf_hip.h), exposed through the reference's own Python surface.

    from brdf_nerf_amd import load_model, render_rays      # replaces `models.load_model`, `rendering.render_rays`

The HIP library is mandatory: importing the compute entry points without it raises.
"""
from .field import SpSBRDFNeRF, load_model  # noqa: F401
from .rendering import render_rays, inference, get_z_vals, cal_weight  # noqa: F401
from . import functions  # noqa: F401
from ._lib import set_deterministic  # noqa: F401

__all__ = ["SpSBRDFNeRF", "load_model", "render_rays", "inference", "get_z_vals", "cal_weight", "functions", "set_deterministic"]
