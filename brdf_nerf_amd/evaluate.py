"""Evaluation-side callers of the path (reference: eval.py:26-105).

  batched_inference  eval.py:56-76   chunked no-grad render_rays, per-chunk dicts concatenated
  load_ckpt / extract_model_state_dict  eval.py:26-54  Lightning checkpoint -> module state_dict (prefix stripping)
"""
from collections import defaultdict

import torch

from .rendering import render_rays


@torch.no_grad()
def batched_inference(models, rays, ts, args, mode="test", apply_brdf=False, apply_theta=False, cos_irra_on=False, **kw):
    chunk = args.chunk
    results = defaultdict(list)
    for i in range(0, rays.shape[0], chunk):
        out, _ = render_rays(models, args, rays[i:i + chunk], None if ts is None else ts[i:i + chunk], mode=mode,
                             apply_brdf=apply_brdf, apply_theta=apply_theta, cos_irra_on=cos_irra_on, **kw)
        for k, v in out.items():
            results[k].append(v)
    return {k: torch.cat(v, 0) for k, v in results.items()}


def extract_model_state_dict(ckpt_path, model_name="model", prefixes_to_ignore=(), drop_len=-1, trusted=False):
    """eval.py:26-47: keep the entries whose key starts with `model_name` and drop the first drop_len+1 characters
    (drop_len = len(model_name) by default; main.py:97-104 passes 'nerf_coarse.<sub>' with drop_len=11 for partial warm starts).
    Checkpoints written by the reference (Lightning 1.3) pickle callback classes and optimizer state next to the weights,
    which torch's safe unpickler (the default of torch.load since 2.6) refuses: pass trusted=True for files you trust."""
    try:
        checkpoint = torch.load(ckpt_path, map_location="cpu", weights_only=not trusted)
    except Exception as e:      # pickle.UnpicklingError from the safe unpickler
        if trusted:
            raise
        raise RuntimeError(f"{ckpt_path}: not loadable with weights_only=True ({type(e).__name__}: {str(e)[:200]}). A Lightning "
                           f"checkpoint of the reference carries pickled callback / optimizer objects: pass trusted=True "
                           f"(load_ckpt(..., trusted=True)) if you trust the file.") from e
    sd = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
    if drop_len < 0:
        drop_len = len(model_name)
    out = {}
    for k, v in sd.items():
        if not k.startswith(model_name):
            continue
        k2 = k[drop_len + 1:]
        if any(k2.startswith(p) for p in prefixes_to_ignore):
            continue
        out[k2] = v
    return out


def load_ckpt(model, ckpt_path, model_name="model", prefixes_to_ignore=(), drop_len=-1, trusted=False):
    """eval.py:49-54: partial load (strict=False semantics of updating the model's own state_dict)."""
    sd = model.state_dict()
    sd.update(extract_model_state_dict(ckpt_path, model_name, prefixes_to_ignore, drop_len, trusted))
    model.load_state_dict(sd)
    return model


@torch.no_grad()
def render_image(models, args, rays, rgbs=None, keys=("rgb", "depth"), chunk=None, apply_brdf=False, apply_theta=False,
                 cos_irra_on=False, group=None, **kw):
    """Full-image evaluation (eval.py:56-76, 379-507; metrics.py:292-325): the H*W rays of one image rendered in chunks,
    keeping ONLY the requested ray-level / per-sample entries of each chunk (an image at S+G = 128 samples would
    otherwise hold ~40 per-sample tensors), and the PSNR against `rgbs` when given.  Under data parallelism every rank
    renders its contiguous share of the rays and the rows are all-gathered (the only exchange).
    Returns a dict with `keys` (suffix-free) and `psnr`."""
    from .distributed import gather_rows, shard_bounds, world_info
    from .losses import psnr
    rank, world = world_info(group)
    lo, hi = shard_bounds(rays.shape[0], rank, world)
    mine = rays[lo:hi]
    chunk = chunk or args.chunk
    parts = {k: [] for k in keys}
    for i in range(0, mine.shape[0], chunk):
        out, _ = render_rays(models, args, mine[i:i + chunk], None, mode="test", apply_brdf=apply_brdf, apply_theta=apply_theta,
                             cos_irra_on=cos_irra_on, **kw)
        for k in keys:
            parts[k].append(out[f"{k}_coarse"])
    res = {}
    for k in keys:
        v = torch.cat(parts[k], 0) if parts[k] else rays.new_zeros((0,))
        res[k] = gather_rows(v, group) if world > 1 else v
    if rgbs is not None and "rgb" in res:
        res["psnr"] = psnr(res["rgb"], rgbs)
    return res
