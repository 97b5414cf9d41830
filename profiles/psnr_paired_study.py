"""north_star "PSNR within 0.05 dB": paired study of the BRDF stage (BASELINE config 3's model, RPV + analytic normals; round 5:
BASELINE config 5's models too - Hapke (b, c), Hapke (b, c, theta), microfacet, all with analytic normals).

All modes start from ONE fp32 Lambertian pretraining (400 steps); then the BRDF stage (default 600 steps, lr 5e-4 -> 0) runs
in fp32 / bf16 / fp16 with the same batches and, per seed, the same in-kernel draws, in deterministic mode (bitwise
reproducible sums: what differs between the modes of a pair is the arithmetic, nothing else).  Reports the held-out PSNR per
run, the paired differences to fp32, their mean, standard deviation and the 95 % interval of the mean (Student t).

    python profiles/psnr_paired_study.py [--seeds=16] [--first-seed=101] [--steps=600] [--config=rpv_nan|lambert|hapke_bc|hapke_bct|microfacet]
    python profiles/psnr_paired_study.py --combine=a.txt,b.txt,...     # statistics over the "seed N: ..." lines of earlier runs

--protocol=restart (default; rounds 3-4): every seed restarts the BRDF stage from the warm start with fresh heads and a fresh
optimiser state.  --protocol=continue (round 5): the BRDF stage is trained ONCE in fp32 (--stage-steps, default 1500), then every
seed continues it for --cont-steps (default 150; lr 1e-4 -> 0) in each mode from that shared model AND its optimiser state - the
in-suite gate (a) of tests/test_gpu_parity.py at study size.  The restart protocol measures the arithmetic THROUGH the stage's
chaotic first steps (three fresh heads, Adam's first +-lr updates): for RPV and microfacet the paired differences have a standard
deviation of 0.15-0.2 dB; for the Hapke models on this synthetic scene fp32 ITSELF ends between 8 and 13 dB depending on the draw
seed (profiles/r05_c5_gate_stage*.txt) - no number of seeds resolves 0.05 dB there.  The continue protocol starts the modes
together and asks what the arithmetic alone does to training.
"""
import os
import statistics
import sys
import time

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, here)
sys.path.insert(0, os.path.join(here, "tests"))
import torch  # noqa: E402
from scipy import stats  # noqa: E402


def main():
    opt = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    n_seeds, steps, config = int(opt.get("seeds", 16)), int(opt.get("steps", 600)), opt.get("config", "rpv_nan")
    seed0 = int(opt.get("first-seed", 101))
    if "combine" in opt:
        import re
        runs, seen = {m: [] for m in ("fp32", "bf16", "fp16")}, set()
        for path in opt["combine"].split(","):
            for line in open(path):
                m = re.match(r"seed (\d+): fp32 ([\d.]+)  bf16 ([\d.]+)  fp16 ([\d.]+)", line)
                if m and int(m.group(1)) not in seen:
                    seen.add(int(m.group(1)))
                    for k, v in zip(runs, m.groups()[1:]):
                        runs[k].append(float(v))
        print(f"combined {len(seen)} seeds ({min(seen)}..{max(seen)}) of {opt['combine']}")
        return summary(runs, len(seen))
    import brdf_nerf_amd
    import test_gpu_parity as T
    from oracle.config import FieldConfig
    brdf_nerf_amd.set_deterministic(True)
    brdf = config != "lambert"
    models = dict(T.C5_MODELS, rpv_nan=T.RPV_NAN, lambert={})
    cfg = FieldConfig(n_samples=64, guided_samples=64, **models[config])
    from brdf_nerf_amd import build as B
    import hashlib
    per_file = " ".join(f"{f}:{hashlib.sha256(open(os.path.join(B.CSRC, f), 'rb').read()).hexdigest()[:10]}"
                        for f in sorted(os.listdir(B.CSRC)) if f.endswith((".hip", ".h", ".cpp")))
    print(f"kernel sources {B.source_hash()} (library {B.library_hash()}); per file: {per_file}", flush=True)
    train, held = T._learnable_table(1024 * 64, 3), T._learnable_table(8192, 11)
    warm = first = None
    protocol = opt.get("protocol", "restart")
    stage_steps, cont_steps = int(opt.get("stage-steps", 1500)), int(opt.get("cont-steps", 150))
    adam, trained, p_trained = {}, None, None
    if brdf:
        _, first, warm = T._psnr_run(cfg, "fp32", 400, 0, train, held, draw_seed=1)
        if protocol == "continue":
            p_trained, _, trained = T._psnr_run(cfg, "fp32", 0, stage_steps, train, held, draw_seed=3, init_state=warm, keep_adam=adam)
            print(f"shared fp32 model: held-out PSNR {p_trained:.4f} dB after 400 Lambertian + {stage_steps} BRDF steps", flush=True)
    runs = {m: [] for m in ("fp32", "bf16", "fp16")}
    t0 = time.time()
    for s in range(n_seeds):
        for m in runs:
            if brdf and protocol == "continue":
                p = T._psnr_run(cfg, m, 0, cont_steps, train, held, draw_seed=seed0 + s, init_state=trained, lr0=1e-4, adam=adam)[0]
            elif brdf:
                p = T._psnr_run(cfg, m, 0, steps, train, held, draw_seed=seed0 + s, init_state=warm)[0]
            else:
                p, first, _ = T._psnr_run(cfg, m, steps, 0, train, held, draw_seed=seed0 + s)
            runs[m].append(p)
        print(f"seed {seed0 + s}: " + "  ".join(f"{m} {runs[m][-1]:.4f}" for m in runs) +
              f"   (bf16-fp32 {runs['bf16'][-1] - runs['fp32'][-1]:+.4f}, fp16-fp32 {runs['fp16'][-1] - runs['fp32'][-1]:+.4f})"
              f"   [{time.time() - t0:.0f} s]", flush=True)
    what = (f"{cont_steps} BRDF steps (lr 1e-4 -> 0, Adam state carried) continuing a shared fp32 model ({p_trained:.4f} dB after 400 + {stage_steps} steps)"
            if (brdf and protocol == "continue") else f"{steps} {'BRDF' if brdf else 'Lambertian'} steps ({'restart from the warm start' if brdf else 'from the initialisation'})")
    print(f"config {config}, protocol {protocol}: {what} of 1024 rays x (64 + 64) samples, F = 512, {n_seeds} draw seeds, "
          f"deterministic mode; first-step training PSNR {first:.2f} dB; held-out PSNR of 8192 rays")
    summary(runs, n_seeds)


def summary(runs, n_seeds):
    for m in runs:
        print(f"  {m}: mean {statistics.mean(runs[m]):.4f} dB, sd over seeds {statistics.stdev(runs[m]):.4f}")
    tq = float(stats.t.ppf(0.975, n_seeds - 1))
    for m in ("bf16", "fp16"):
        d = [a - b for a, b in zip(runs[m], runs["fp32"])]
        mean, sd = statistics.mean(d), statistics.stdev(d)
        hw = tq * sd / n_seeds ** 0.5
        print(f"  {m} - fp32, paired by seed: mean {mean:+.4f} dB, sd {sd:.4f}, 95 % interval [{mean - hw:+.4f}, {mean + hw:+.4f}] "
              f"(half-width {hw:.4f}; gate |mean| <= 0.05 and half-width <= 0.05: {'PASS' if abs(mean) <= 0.05 and hw <= 0.05 else 'FAIL'})")


if __name__ == "__main__":
    main()
