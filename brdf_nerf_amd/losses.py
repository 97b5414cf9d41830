"""The two losses that consume the path's outputs in training, without the reference's host round-trips.

  SNerfLoss  metrics.py:39-61  (lambda_sc = 0: MSE on rgb)
  DepthLoss  metrics.py:82-161 (subset=True, GNLL=False): rows with a valid depth prior, optionally only those
             outside the expected distribution; lambda_ds/3 * mean(n_sel/n_rays * w * (d - d*)^2).
The reference selects rows with np.where(valid.cpu() > 0) (three device->host syncs per step); here the same
subset is expressed with masks on the device (identical value and gradient, no sync).
"""
import torch


def snerf_loss(rgb, target, lambda_rgb=1.0):
    return lambda_rgb * torch.mean((rgb - target) ** 2)


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
