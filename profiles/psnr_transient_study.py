"""Where do the 16-bit modes fall behind in the BRDF stage of config 3 (protocol (b) of the PSNR gate: 600 steps from a Lambertian
warm start, fresh heads, fresh Adam state)?  Held-out PSNR at checkpoints along the stage, dropped (non-finite) per-sample
gradient elements, per mode, several sampling seeds; deterministic mode.
    python profiles/psnr_transient_study.py [n_seeds] [variant]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import brdf_nerf_amd  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from oracle.config import FieldConfig  # noqa: E402
from brdf_nerf_amd import load_model  # noqa: E402
from brdf_nerf_amd.evaluate import render_image  # noqa: E402
from brdf_nerf_amd.trainer import FusedTrainer  # noqa: E402

DEV = "cuda"


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    variant = sys.argv[2] if len(sys.argv) > 2 else "default"
    brdf_nerf_amd.set_deterministic(True)
    cfg = FieldConfig(n_samples=64, guided_samples=64, **T.RPV_NAN)
    train, held = T._learnable_table(1024 * 64, 3), T._learnable_table(8192, 11)
    _, first, warm = T._psnr_run(cfg, "fp32", 400, 0, train, held, draw_seed=1)
    marks = (50, 100, 200, 300, 450, 600)
    for seed in range(n_seeds):
        for dt in ("fp32", "bf16", "fp16"):
            args = T.make_args(cfg, dt)
            torch.manual_seed(0)
            model = load_model(args).to(DEV)
            model.load_state_dict(warm)
            tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
            if variant == "nosanitize":
                tr.sanitize_grads = False
            train.load_state_dict({"gen": torch.Generator(device=DEV).manual_seed(5).get_state(), "perm": None, "cursor": 0, "epoch": 0})
            torch.manual_seed(11 + seed)
            out = []
            for i in range(600):
                tr.lr = 5e-4 * math.cos(0.5 * math.pi * i / 600) ** 2
                b = train.next_batch(1024)
                tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"], near_far=(0.0, 2.2),
                        apply_brdf=True, apply_theta=True, cos_irra_on=True, depth_loss_on=False)
                if i + 1 in marks:
                    st = torch.cuda.get_rng_state()
                    torch.manual_seed(2)
                    res = render_image({"coarse": model}, args, held.data["rays"], held.data["rgbs"], keys=("rgb",), chunk=2048, apply_brdf=True,
                                       apply_theta=True, cos_irra_on=True)
                    torch.cuda.set_rng_state(st)
                    out.append(float(res["psnr"]))
            print(f"seed {seed} {dt}: " + " ".join(f"{m}:{p:.3f}" for m, p in zip(marks, out)) + f"  dropped {tr.dropped_grad_elems()}", flush=True)


if __name__ == "__main__":
    main()
