"""A/B of variant builds of the library IN ONE PROCESS, alternating (cdna_hip_programming.md rule 24): per-kernel HIP-event
times of the fused training step and of the sigma-only inference forward, N rounds, median and min per variant.

    python -m brdf_nerf_amd.build -DFLAG ...          # builds brdf_nerf_amd/build/<tag>/libbrdfnerf_hip.so
    python profiles/ab_kernels.py default FLAG [...]  [--config lambert|rpv_nan] [--dtype bf16] [--rounds 5]

Every variant gets its own model / trainer (same seed) and its own dlopen'ed library; `_lib.use()` switches between them.
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd import _lib as L  # noqa: E402
from brdf_nerf_amd import functions as Fn  # noqa: E402
from brdf_nerf_amd import load_model  # noqa: E402
from brdf_nerf_amd.trainer import FusedTrainer  # noqa: E402


def main():
    tags = [a for a in sys.argv[1:] if not a.startswith("--")]
    opt = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    config, dtype, rounds = opt.get("config", "lambert"), opt.get("dtype", "bf16"), int(opt.get("rounds", 5))
    n_rays = int(opt.get("rays", 4096))
    dev = torch.device("cuda", 0)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    variants = {}
    for tag in tags:
        # "<build>[:attr=value,...]": a library build (default = the in-tree one) and FusedTrainer attributes to set on its trainer
        build, _, attrs = tag.partition(":")
        path = os.path.join(here, "brdf_nerf_amd", "libbrdfnerf_hip.so") if build == "default" else \
            os.path.join(here, "brdf_nerf_amd", "build", build, "libbrdfnerf_hip.so")
        h = L.load(path, baseline=build.startswith("r0"))     # "r0N...": an earlier round's library (profiles/build_baseline.py)
        L.use(h)
        args = bench.make_args(n_rays, 64, 64, dtype, **bench.CONFIG_FLAGS[config][0])
        torch.manual_seed(0)
        model = load_model(args).to(dev)
        tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
        tr.use_graph = False              # the per-kernel events live in the library's launch sites: eager steps
        for kv in filter(None, attrs.split(",")):
            k, v = kv.split("=")
            if k == "zero_stash":      # timing probes whose forward does not write its stash: the backward must not read NaN bit patterns
                spec0 = model.spec(*[bool(x) for x in (bench.CONFIG_FLAGS[config][1]["apply_brdf"], bench.CONFIG_FLAGS[config][1]["apply_theta"])],
                                   tr.nr_lr, tr.nr_an, beta=False)
                tr._buf("stash_all", (Fn.field_stash_bytes(spec0, n_rays * 128),), torch.uint8).zero_()
                continue
            setattr(tr, k, {"True": True, "False": False}.get(v, int(v) if v.lstrip("-").isdigit() else v))
        variants[tag] = (h, args, model, tr)
    b = bench.synthetic_batch(n_rays, 1, dev)
    flags = bench.CONFIG_FLAGS[config][1]
    z = torch.sort(torch.rand(n_rays, 128, device=dev) * 2, -1)[0]
    times = {tag: {} for tag in tags}
    counts = {tag: {} for tag in tags}       # launches per step (field_fwd_sigma: the separate sigma-only inference forward, 1 per call)
    for rnd in range(rounds + 1):                       # round 0 = warm-up
        for tag in tags:
            h, args, model, tr = variants[tag]
            L.use(h)
            L.prof_enable(True)
            for _ in range(3):
                tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                        near_far=(0.0, 2.0), **flags)
            spec = model.spec(False, False, False)
            packed = model.repack(spec)
            for _ in range(4):
                Fn.field_sigma(spec, model.named(), packed, rays=b["rays"], z=z)
            torch.cuda.synchronize()
            prof = L.prof_collect()
            L.prof_enable(False)
            if rnd == 0:
                continue
            for k, (ms, n) in prof.items():
                times[tag].setdefault(k, []).append(ms / n)
                counts[tag][k] = n / (4 if k == "field_fwd_sigma" else 3)
    # whole steps, graph replay where the trainer captures one: wall time of 30 steps per variant, alternating
    import time
    step_ms = {tag: [] for tag in tags}
    for rnd in range(rounds + 1):
        for tag in tags:
            h, args, model, tr = variants[tag]
            L.use(h)
            tr.use_graph = True
            for _ in range(6):
                tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                        near_far=(0.0, 2.0), **flags)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30):
                tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                        near_far=(0.0, 2.0), **flags)
            torch.cuda.synchronize()
            if rnd:
                step_ms[tag].append((time.perf_counter() - t0) / 30 * 1e3)
    keys = ["pack", "field_fwd_sigma", "field_fwd_full", "field_bwd_chain", "wgrad", "skinny_wgrad", "wgrad_reduce", "field_adjoint", "field_adjoint_bwd",
            "composite_fwd", "composite_bwd", "brdf", "guided_samples", "stratified_z", "adam"]
    print(f"config {config} dtype {dtype} rays {n_rays}: launches per step x ms per launch, median (min) over {rounds} alternating rounds")
    print(f"{'kernel':>18} " + " ".join(f"{t[:26]:>26}" for t in tags))
    for k in keys:
        if not any(k in times[t] for t in tags):
            continue
        row = []
        for t in tags:
            v = times[t].get(k)
            row.append(f"{counts[t][k]:g} x {statistics.median(v):.4f} ({min(v):.4f})" if v else "-")
        print(f"{k:>18} " + " ".join(f"{c:>26}" for c in row))
    print(f"{'step (wall, ms)':>18} " + " ".join(f"{statistics.median(step_ms[t]):.4f} ({min(step_ms[t]):.4f})".rjust(26) for t in tags))


if __name__ == "__main__":
    main()
