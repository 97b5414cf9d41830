"""Diagnostic: when a Hapke(+theta) training step yields non-finite gradients, dump the per-ray BRDF inputs of the rows
that produce them (gpurun_out/hapke_nan_rows.pt) so the oracle's autograd can be run on the same rows."""
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from oracle.config import FieldConfig  # noqa: E402
from test_gpu_parity import make_args  # noqa: E402
from brdf_nerf_amd import load_model  # noqa: E402
from brdf_nerf_amd import functions as Fn  # noqa: E402
from brdf_nerf_amd.trainer import FusedTrainer  # noqa: E402

DEV = "cuda:0"
rec = {}
orig = Fn.HapkeFunction.apply


def spy(*a):
    rec["in"] = [t.detach().clone() if torch.is_tensor(t) else t for t in a]
    return orig(*a)


Fn.HapkeFunction.apply = spy
cfg = FieldConfig(n_samples=64, guided_samples=64, b=1, c=1, theta=1, normal="analystic")
args = make_args(cfg, "bf16")
b = bench.synthetic_batch(1024, 5, torch.device(DEV))


def dump():
    ins = [t.clone().requires_grad_(True) if torch.is_tensor(t) and t.dtype.is_floating_point else t for t in rec["in"]]
    brdf, aux = orig(*ins)
    brdf.sum().backward()
    bad = ~torch.isfinite(brdf).all(-1)
    for t in ins:
        if torch.is_tensor(t) and t.grad is not None:
            bad |= ~torch.isfinite(t.grad.reshape(t.shape[0], -1)).all(-1)
    rows = torch.nonzero(bad).flatten()
    print("rows with non-finite BRDF value/gradient:", rows.tolist()[:20], "of", brdf.shape[0])
    os.makedirs("gpurun_out", exist_ok=True)
    torch.save({"rows": rows.cpu(), "inputs": [t.detach()[rows].cpu() if torch.is_tensor(t) else t for t in rec["in"]],
                "brdf": brdf.detach()[rows].cpu(),
                "grads": [t.grad[rows].cpu() if torch.is_tensor(t) and t.grad is not None else None for t in ins]},
               "gpurun_out/hapke_nan_rows.pt")


def search():
    from brdf_nerf_amd import render_rays
    for seed in range(8):
        torch.manual_seed(0)
        model = load_model(args).to(DEV)
        if seed == 0:      # the sequence of tests/test_gpu_parity.py::test_full_size_render_and_train_step_properties
            with torch.no_grad():
                render_rays({"coarse": model}, args, b["rays"], None, mode="test", apply_brdf=True, apply_theta=True, cos_irra_on=True)
        else:
            torch.manual_seed(100 + seed)
        tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
        for i in range(10):
            loss, rgb = tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                                near_far=(0.0, 2.0), apply_brdf=True, apply_theta=True, cos_irra_on=True)
            nbad = int((~torch.isfinite(tr.flat_grad)).sum())
            print("seed", seed, "step", i, "loss", float(loss), "non-finite grads", nbad, flush=True)
            if nbad:
                dump()
                return
    print("no non-finite gradient in 8 x 10 steps")


search()
