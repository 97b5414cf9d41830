"""Oracle: the two losses that consume the path's outputs in training (PyTorch CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.
Follows /root/reference metrics.py:39-61 (SNerfLoss, lambda_sc=0) and metrics.py:82-161
(DepthLoss, subset=True, GNLL=False) as configured by main.py:60-72 for spsbrdf-nerf.
"""
import torch
from .render import depth_std


def snerf_loss(results, rgbs, lambda_rgb=1.0):
    return lambda_rgb * torch.mean((results["rgb_coarse"] - rgbs) ** 2)


def depth_loss(results, target_depth, target_weight, valid_depth, target_std, lambda_ds, usealldepth=False):
    """ComputeSubsetDepthLoss: rows with valid depth, optionally only those outside the expected
    distribution; loss = lambda_ds/3 * mean(n_sel/n_rays * w * (d - d*)^2)."""
    sel = valid_depth > 0
    z, d, w = results["z_vals_coarse"][sel], results["depth_coarse"][sel], results["weights_coarse"][sel]
    if d.shape[0] == 0:
        return torch.zeros((), dtype=target_depth.dtype)
    std = depth_std(z, d, w)
    tw, td, ts = target_weight[sel], target_depth[sel], target_std[sel]
    if usealldepth:
        apply = torch.ones_like(td, dtype=torch.bool)
    else:
        apply = (((d - td).abs() - ts) > 0) | (ts < std)
    d, td, tw = d[apply], td[apply], tw[apply]
    if d.shape[0] == 0:
        return torch.zeros((), dtype=target_depth.dtype)
    ratio = float(d.shape[0]) / float(valid_depth.shape[0])
    return (lambda_ds / 3.0) * torch.mean(ratio * tw * (d - td) ** 2)
