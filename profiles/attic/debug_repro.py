"""Hunt for rare non-reproducibility: (1) the evaluation render of one fixed model, repeated - no atomics are involved, so every
repetition must be bitwise identical; (2) one deterministic-mode training step from one fixed state, repeated likewise."""
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
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
held = T._learnable_table(8192, 11)
b = bench.synthetic_batch(1024, 5, torch.device(DEV))
for name, kw, flags in (("rpv_nan", T.RPV_NAN, dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)), ("lambert", {}, {})):
    cfg = FieldConfig(n_samples=64, guided_samples=64, **kw)
    for dt in ("fp32", "bf16", "fp16"):
        args = T.make_args(cfg, dt)
        torch.manual_seed(0)
        model = load_model(args).to(DEV)
        ref = None
        bad = 0
        for i in range(N):
            torch.manual_seed(2)
            res = render_image({"coarse": model}, args, held.data["rays"], held.data["rgbs"], keys=("rgb", "depth"), chunk=2048, **flags)
            cur = torch.cat([res["rgb"].flatten(), res["depth"].flatten()])
            if ref is None:
                ref = cur.clone()
            elif not torch.equal(cur, ref):
                bad += 1
                d = (cur - ref).abs()
                print(f"  eval {name} {dt} rep {i}: {int((d > 0).sum())} values differ, max {float(d.max()):.3e}", flush=True)
        print(f"eval {name} {dt}: {bad} of {N - 1} repetitions differ", flush=True)
        brdf_nerf_amd.set_deterministic(True)
        ref = None
        bad = 0
        for i in range(N):
            torch.manual_seed(0)
            m2 = load_model(args).to(DEV)
            tr = FusedTrainer(m2, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
            torch.manual_seed(3)
            tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"], near_far=(0.0, 2.0), **flags)
            cur = tr.flat_grad.clone()
            if ref is None:
                ref = cur
            elif not torch.equal(cur, ref):
                bad += 1
                d = (cur - ref).abs()
                print(f"  step {name} {dt} rep {i}: {int((d > 0).sum())} values differ, max {float(d.max()):.3e} of {float(ref.abs().max()):.3e}", flush=True)
        brdf_nerf_amd.set_deterministic(False)
        print(f"step {name} {dt}: {bad} of {N - 1} repetitions differ", flush=True)
