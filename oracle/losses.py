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


# ---- regularisers (metrics.py:179-290), as main.py:271-327 applies them to render_rays' result dict
def normal_reg_loss(results, keyword, lambda_nr_reg):
    """NormalRegLoss (metrics.py:179-216): normals facing away from the camera, weighted by the compositing weights.
    The reference SUMS w * min(0, n.v)^2 over every sample of the batch and multiplies by lambda (its torch.mean acts
    on that scalar).  Returns (loss, percentage of back-facing normals)."""
    normal = results[f"{keyword}_coarse"].reshape(-1, 3)
    weights = results["weights_coarse"].reshape(-1)
    view = results["rays_d_coarse"].reshape(-1, 3)
    rep = normal.shape[0] // view.shape[0]
    n_dot_v = (normal * torch.repeat_interleave(view, rep, dim=0)).sum(-1)
    perc = 100.0 * float((n_dot_v < 0).sum()) / n_dot_v.numel()
    return lambda_nr_reg * (weights * torch.minimum(torch.zeros_like(n_dot_v), n_dot_v) ** 2).sum(), perc


def hard_surface_loss(results, lambda_hs):
    """HardSurfaceLoss (metrics.py:263-290): mean over rays of sum_s w (z - depth)^2 (train_utils.py:38-39)."""
    z, d, w = results["z_vals_coarse"], results["depth_coarse"], results["weights_coarse"]
    return lambda_hs * torch.mean(((z - d.unsqueeze(-1)) ** 2 * w).sum(-1))


def normal_loss(weights, normal_gt, normal_pred, lambda_nr_spv, keyword="an_lr", target_weight=None, valid_depth=None):
    """NormalLoss (metrics.py:218-261).  'an_lr': mean(weights) * L1(normal_gt, normal_pred) over per-sample normals;
    otherwise the composited normal of the rows with a valid depth prior against per-ray normals, L1 with weights."""
    if keyword == "an_lr":
        return lambda_nr_spv * torch.mean(weights.reshape(-1) * torch.mean((normal_gt - normal_pred).abs()))
    pred = (weights.unsqueeze(-1) * normal_pred).sum(-2)
    sel = valid_depth > 0
    tw = target_weight[sel].unsqueeze(-1)
    return lambda_nr_spv * torch.mean((tw * normal_gt[sel] - tw * pred[sel]).abs())


def psnr(image_pred, image_gt):
    """metrics.py:292-325 (mse / psnr_ / psnr with valid_mask=None, reduction='mean', scl=False): the squared error is
    normalised by max(image_gt)^2 before the mean; returns the first element of the reference's (psnr, psnr_scl) pair."""
    value = (image_pred - image_gt) ** 2 / (torch.max(image_gt) ** 2)
    return -10.0 * torch.log10(torch.mean(value))


def uncertainty_aware_loss(rgb, weights, beta_samples, targets, beta_min=0.05):
    """uncertainty_aware_loss (metrics.py:24-28): beta = sum_s w beta_s + beta_min; colour term mean((rgb - gt)^2 / (2 beta^2)),
    log term (3 + mean(log beta)) / 2.  Returns (colour, logbeta).  For spsbrdf-nerf the reference's training never calls it
    (load_loss gives SNerfLoss, metrics.py:172-173): kept for users who add it, and to pin the beta channel end to end."""
    beta = torch.sum(weights.unsqueeze(-1) * beta_samples, -2) + beta_min
    return ((rgb - targets) ** 2 / (2 * beta ** 2)).mean(), (3 + torch.log(beta).mean()) / 2
