"""Uninitialised-memory hunt: torch.empty() hands out whatever the caching allocator holds.  Run the evaluation render and one
deterministic training step twice - once after filling the allocator's free blocks with NaN, once after filling them with
zeros (and once with 1e30) - from the same state and draws: any difference (or any non-finite result) means some kernel
consumed memory nobody wrote."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import brdf_nerf_amd
import bench
import test_gpu_parity as T
from oracle.config import FieldConfig
from brdf_nerf_amd import load_model
from brdf_nerf_amd.evaluate import render_image
from brdf_nerf_amd.trainer import FusedTrainer

DEV = "cuda"


def poison(value):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    big = [torch.full((1 << 28,), value, device=DEV) for _ in range(24)]           # 24 GiB of large blocks
    small = [torch.full((s,), value, device=DEV) for s in (64, 256, 1024, 4096, 16384, 65536, 200000) for _ in range(64)]
    torch.cuda.synchronize()
    del big, small


held = T._learnable_table(4096, 11)
b = bench.synthetic_batch(1024, 5, torch.device(DEV))
brdf_nerf_amd.set_deterministic(True)
CASES = (("rpv_nan", T.RPV_NAN, dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)), ("lambert", {}, {}),
         ("hapke_nlr_beta_viewdir", dict(b=1, c=1, theta=1, normal="learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)),
         ("rpv_anlr", dict(funcM=1, funcF=1, funcH=1, normal="analystic_learned"), dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)))
for name, kw, flags in CASES:
    for feat in (512, 192):
        cfg = FieldConfig(feat=feat, n_samples=64, guided_samples=64, **kw)
        for dt in ("fp32", "bf16", "fp16"):
            args = T.make_args(cfg, dt)
            outs = {}
            for tag, val in (("nan", float("nan")), ("zero", 0.0), ("big", 1e30)):
                torch.manual_seed(0)
                model = load_model(args).to(DEV)
                poison(val)
                torch.manual_seed(2)
                with torch.no_grad():
                    res = render_image({"coarse": model}, args, held.data["rays"], held.data["rgbs"], keys=("rgb", "depth"), chunk=2048, **flags)
                ev = torch.cat([res["rgb"].flatten(), res["depth"].flatten()]).clone()
                tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
                poison(val)
                torch.manual_seed(3)
                loss, _ = tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                                  near_far=(0.0, 2.0), **flags)
                outs[tag] = (ev, tr.flat_grad.clone(), float(loss))
            e0, g0, l0 = outs["zero"]
            msg = []
            for tag in ("nan", "big"):
                e, g, l = outs[tag]
                fin = bool(torch.isfinite(e).all()) and bool(torch.isfinite(g).all())
                same_e, same_g = torch.equal(e, e0), torch.equal(g, g0)
                msg.append(f"{tag}: finite {fin} eval-identical {same_e} grad-identical {same_g}" + ("" if same_g else f" (max diff {float((g - g0).abs().max()):.3e})"))
            print(f"{name} F={feat} {dt}: " + "; ".join(msg), flush=True)
