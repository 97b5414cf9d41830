"""is the CPU oracle (bench.py's cpu_baseline, kind "port") a fair stand-in for the
reference's own CPU path?  Times ONE training step (render_rays + SNerfLoss + DepthLoss + backward + Adam) of the imported
reference and of the oracle on the same rays / parameters, alternating, in this container (8 host cores).  Build-container
only: needs /root/reference (read-only; imported with the six I/O-only modules stubbed, exactly like
tests/golden/make_goldens.py).  Output -> profiles/history/r02_cpu_cross_timing.txt.

    python profiles/cpu_cross_timing.py [rays=1024] [reps=2]
"""
import contextlib
import io
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
import make_goldens as MG  # noqa: E402
from oracle.config import FieldConfig  # noqa: E402
from oracle import render as ORD, losses as OL  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    cfg = FieldConfig(n_samples=64, guided_samples=64)           # F = 512, 8 layers, PE10, Lambertian (BASELINE config 2)
    b = bench.synthetic_batch(R, 123, "cpu")
    ref = MG.import_reference()
    model, _ = MG.build_ref_model(ref, cfg, seed=0)
    args = MG.ref_args(cfg)
    opt_r = torch.optim.Adam(model.parameters(), lr=5e-4)
    with contextlib.redirect_stdout(io.StringIO()):
        snerf = ref["metrics"].SNerfLoss(lambda_sc=0.0)
        dloss = ref["metrics"].DepthLoss(lambda_ds=10.0, GNLL=False, usealldepth=False, margin=0.0001, stdscale=1, subset=True)

    def step_ref():
        opt_r.zero_grad(set_to_none=True)
        with contextlib.redirect_stdout(io.StringIO()):
            res, _ = ref["rendering"].render_rays({"coarse": model}, args, b["rays"], None, mode="train", valid_depth=b["valid_depth"],
                                                  target_depths=b["depths"], target_std=b["depth_std"])
            loss, _ = snerf(res, b["rgbs"])
            l2, _ = dloss(res, b["depths"][:, 0], b["depths"][:, 1], target_valid_depth=b["valid_depth"], target_std=b["depth_std"])
            loss = loss + l2
        loss.backward()
        opt_r.step()
        return float(loss)

    params = {k: torch.from_numpy(v).requires_grad_(True) for k, v in cfg.make_params(0).items()}
    opt_o = torch.optim.Adam(list(params.values()), lr=5e-4)

    def step_orc():
        opt_o.zero_grad(set_to_none=True)
        res, _ = ORD.render_rays(params, cfg, b["rays"], ORD.Randoms(), mode="train", valid_depth=b["valid_depth"],
                                 target_depths=b["depths"], target_std=b["depth_std"])
        loss = OL.snerf_loss(res, b["rgbs"]) + OL.depth_loss(res, b["depths"][:, 0], b["depths"][:, 1], b["valid_depth"],
                                                              b["depth_std"], 10.0)
        loss.backward()
        opt_o.step()
        return float(loss)

    t = {"reference": [], "oracle": []}
    step_ref(); step_orc()                                       # warm-up
    for _ in range(reps):
        for name, fn in (("reference", step_ref), ("oracle", step_orc)):
            t0 = time.perf_counter()
            fn()
            t[name].append(time.perf_counter() - t0)
    lines = [f"CPU cross-timing, build container ({threads} host threads, torch {torch.__version__} CPU backend, fp32), one training step of",
             f"{R} rays x 64+64 samples, F=512, 8 Siren layers, PE10, SNerfLoss + DepthLoss(ds_lambda=10) + backward + Adam; best of {reps} after warm-up:"]
    for name in ("reference", "oracle"):
        best = min(t[name])
        lines.append(f"  {name:10s} {best:7.3f} s/step = {R / best:7.1f} rays/s   (all: {', '.join(f'{x:.3f}' for x in t[name])})")
    ratio = min(t["oracle"]) / min(t["reference"])
    lines.append(f"  oracle / reference time = {ratio:.3f}  -> bench.py's cpu_baseline (the oracle, kind 'port') "
                 f"{'understates' if ratio > 1 else 'overstates'} the reference's CPU rate by {abs(1 - 1 / ratio) * 100:.1f} %")
    lines.append("  (the reference's render_rays also pays its check_nan host syncs and prints; the oracle restates the same ATen op sequence "
                 "without them, and skips autograd bookkeeping in pass 1, which the reference leaves enabled - SURVEY quirk 11)")
    out = "\n".join(lines)
    print(out)
    open(os.path.join(ROOT, "profiles", "r02_cpu_cross_timing.txt"), "w").write(out + "\n")


if __name__ == "__main__":
    main()
