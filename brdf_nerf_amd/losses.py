"""The losses that consume the path's outputs in training, without the reference's host round-trips.

  SNerfLoss  metrics.py:39-61  (lambda_sc = 0: MSE on rgb)
  DepthLoss  metrics.py:82-161 (subset=True, GNLL=False): rows with a valid depth prior, optionally only those
             outside the expected distribution; lambda_ds/3 * mean(n_sel/n_rays * w * (d - d*)^2).
The reference selects rows with np.where(valid.cpu() > 0) (three device->host syncs per step); here the same
subset is expressed with masks on the device (identical value and gradient, no sync).
  NormalRegLoss / HardSurfaceLoss / NormalLoss  metrics.py:179-290 (optional regularisers, all lambdas default 0).
"""
import torch


def snerf_loss(rgb, target, lambda_rgb=1.0):
    return lambda_rgb * torch.mean((rgb - target) ** 2)


def uncertainty_aware_loss(rgb, weights, beta_samples, target, beta_min=0.05):
    """uncertainty_aware_loss (metrics.py:24-28) on the --beta channel: beta = sum_s w beta_s + beta_min; returns
    (mean((rgb - gt)^2 / (2 beta^2)), (3 + mean(log beta)) / 2).  The reference's own training of spsbrdf-nerf never calls it
    (load_loss returns SNerfLoss for this model, metrics.py:172-173); offered for callers who want the SatNeRF-style term."""
    beta = torch.sum(weights.unsqueeze(-1) * beta_samples, -2) + beta_min
    return ((rgb - target) ** 2 / (2 * beta ** 2)).mean(), (3 + torch.log(beta).mean()) / 2


def depth_loss(z_vals, depth, weights, target_depth, target_weight, valid_depth, target_std, lambda_ds,
               usealldepth=False):
    sel = valid_depth > 0
    std = (((z_vals - depth.unsqueeze(-1)) ** 2) * weights).sum(-1).sqrt()
    if usealldepth:
        apply = sel
    else:
        apply = sel & ((((depth - target_depth).abs() - target_std) > 0) | (target_std < std))
    # mean over the n_apply selected rows of (n_apply / n_rays) * w * se  ==  sum_apply(w * se) / n_rays
    se = torch.where(apply, target_weight * (depth - target_depth) ** 2, torch.zeros_like(depth))
    return (lambda_ds / 3.0) * se.sum() / float(valid_depth.shape[0])


def psnr(rgb, target):
    """metrics.py:292-325: the reference normalises the squared error by max(target)^2."""
    return -10.0 * torch.log10(torch.mean((rgb - target) ** 2 / (torch.max(target) ** 2)))


def normal_reg_loss(normal, weights, view_dir, lambda_nr_reg):
    """NormalRegLoss (metrics.py:179-216) for one normal field: normal (R,S,3), weights (R,S), view_dir (R,3) pointing
    toward the camera.  The reference sums w * min(0, n.v)^2 over every sample of the batch (its mean acts on that
    scalar).  Returns (loss, fraction of back-facing normals as a 0-d tensor: no host sync)."""
    n_dot_v = (normal * view_dir.unsqueeze(1)).sum(-1)
    loss = lambda_nr_reg * (weights * torch.clamp_max(n_dot_v, 0.0) ** 2).sum()
    return loss, (n_dot_v < 0).float().mean() * 100.0


def hard_surface_loss(z_vals, depth, weights, lambda_hs):
    """HardSurfaceLoss (metrics.py:263-290): mean over rays of sum_s w (z - depth)^2."""
    return lambda_hs * torch.mean(((z_vals - depth.unsqueeze(-1)) ** 2 * weights).sum(-1))


def normal_loss(weights, normal_gt, normal_pred, lambda_nr_spv, keyword="an_lr", target_weight=None, valid_depth=None):
    """NormalLoss (metrics.py:218-261).  keyword 'an_lr' (nr_spv_type 1): mean(weights) * L1 between the two per-sample
    normal fields.  Otherwise (types 2/3): the composited normal of the rows with a valid depth prior against per-ray
    target normals, weighted L1 - rows selected with a mask instead of np.where(valid.cpu() > 0)."""
    if keyword == "an_lr":
        return lambda_nr_spv * torch.mean(weights) * torch.mean((normal_gt - normal_pred).abs())
    pred = (weights.unsqueeze(-1) * normal_pred).sum(-2)
    sel = (valid_depth > 0).float().unsqueeze(-1)
    tw = target_weight.unsqueeze(-1)
    n_sel = sel.sum().clamp_min(1.0) * 3.0
    return lambda_nr_spv * ((tw * normal_gt - tw * pred).abs() * sel).sum() / n_sel
