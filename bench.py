#!/usr/bin/env python3
"""Headline benchmark: spsbrdf-nerf TRAIN rays/s on MI355X (BASELINE.json metric).

One "step" = one full training step on a batch of R rays per GPU:
  render_rays (pass 1 on S stratified samples, depth-guided resampling, pass 2 on the merged S+G samples)
  + SNerfLoss + DepthLoss (ds_lambda=10, README stage 1) + backward + [RCCL grad all-reduce] + Adam.
Workload at N=1: BASELINE config 2 - Lambertian pretrain, 4096 rays x 64 samples (+64 guided), F=512, 8 Siren layers,
PE(10), bf16 MFMA, synthetic satellite-shaped rays, random-init weights (seed 0).
The step is FusedTrainer's launch-lean one (own in-kernel draws, as brdf_nerf_amd.train.TrainLoop runs it): 14 library launches + a fixed-order loss sum,
no host synchronisation, replayed from a captured HIP graph at N = 1 once its inputs have kept their addresses for 3 steps.
N>1: one process per GPU, gradients all-reduced over RCCL in two buckets overlapped with the rest of the backward; `value` is the
weak-scaling rate (`--scaling weak`, default: 4096 rays per GPU) and the line carries a `strong` sub-record - ONE 4096-ray batch
split N ways (north_star's "4096-ray/64-sample batches at 8 GPUs") next to the same shard timed on one rank alone, both in this
invocation (`speedup_vs_n1_ms`); `--scaling strong` makes the strong shape the headline instead.

    python bench.py --gpus N --steps K --warmup W        # N > 1: this process spawns the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  Timing protocol: W warm-up steps, two steady steps for an estimate, then untimed "settling"
steps until the chip has run STEADY-STATE steps for >= 1.5 s (DVFS settles over seconds), then EXACTLY K steps between barrier +
synchronize pairs with no instrumentation inside, then `sustained` (200 more steps, outside `value`); the per-kernel HIP-event
times and `launches_per_step` come from a SEPARATE eager pass after the timed one; `box_calibration_before` / `box_calibration`
(N = 1) = ~1 s of vendor bf16 GEMMs before the warm-up steps and after all of that: a box-speed indicator (boxes of the pool differ
by up to 10 % with unchanged code).  `roofline`
prices the kernel with the largest share of the step, as a fraction of the dense bf16/fp16 MFMA peak both by the
reference network's algorithmic FLOPs and by the FLOPs the build executes (the linear feats layer is folded into the
heads); `traffic` / `mfma_busy` come from the committed rocprofv3 PMC pass of the SAME kernel sources
(profiles/r05_pmc.json, keyed by a hash of csrc/; `traffic_live` false says so), null when the sources have changed since.  `cpu_baseline` times the
CPU oracle (a port of the reference's PyTorch path) on a bounded sample of the same workload.
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA16_TFLOPS = 2500.0   # MI355X dense bf16 / fp16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3       # fp32-input MFMA
PEAK_HBM_GBS = 8000.0         # HBM3E spec (6.29 TB/s measured copy rate, MI355X_MICROARCH.md)
METRIC = "train rays/sec (+ MFMA% of roofline), spsbrdf-nerf 64 samples/ray, 1/2/4/8 MI355X"

# name -> (model flags, step flags, BASELINE config it belongs to)
_BRDF_ON = dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)
CONFIG_FLAGS = {
    "lambert": (dict(), dict(apply_brdf=False, apply_theta=False, cos_irra_on=False), 2),
    "rpv_nan": (dict(funcM=1, funcF=1, funcH=1, normal="analystic"), _BRDF_ON, 3),
    "rpv_nlr": (dict(funcM=1, funcF=1, funcH=1, normal="learned"), _BRDF_ON, 3),
    "hapke": (dict(b=1, c=1, theta=1, normal="analystic"), _BRDF_ON, 5),
    "microfacet": (dict(roughness=True, normal="analystic"), _BRDF_ON, 5),
}


def flops_per_point(F=512, P=60, L=8, n_heads=1, executed=False):
    """FLOPs (2*MAC) per sample point, by kernel.  executed=False: the reference network's algorithmic count (DESIGN.md
    section 3).  executed=True: what the build runs - the linear F x F feats layer is folded into
    the heads' first layers, so one F x F product less in forward, backward chain and weight gradient."""
    H2 = F // 2
    fold = 0 if not executed else 2 * F * F
    trunk = 2 * (P * F + (L - 2) * F * F + (F + P) * F)
    heads1 = n_heads * 2 * F * H2
    heads2 = 2 * 3 * H2 + (n_heads - 1) * 2 * H2          # rgb (3 outputs) + 1-wide BRDF heads
    fwd_sigma = trunk + 2 * F
    fwd_full = trunk + 2 * F + 2 * F * F + heads1 + heads2 - fold
    bwd_chain = heads1 + heads2 + 2 * F * F + 2 * F + (L - 1) * 2 * F * F - fold     # dX products (h inputs only)
    wgrad = trunk + 2 * F * F + heads1 - fold
    skinny = 2 * F + heads2
    return dict(field_fwd_sigma=fwd_sigma, field_fwd_full=fwd_full, field_bwd_chain=bwd_chain, wgrad=wgrad, skinny_wgrad=skinny)


def make_args(batch, n_samples, guided, dtype, **over):
    a = argparse.Namespace(
        model="spsbrdf-nerf", fc_layers=8, fc_feat=512, mapping=True, siren=1, t_embbeding_tau=4, beta=False, roughness=False,
        normal="none", indirect_light=False, glossy_scale=1.0, sun_v="none", MultiBRDF=0, dim_RPV=1, input_viewdir=0, funcM=0,
        funcF=0, funcH=0, b=0, c=0, theta=0, shell_hapke=0, hpk_scl=4.0, guided_samples=guided, n_samples=n_samples,
        n_importance=0, std_range=3.0, data="sat", sc_lambda=0.0, chunk=5120, noise_std=0.0, margin=1e-4, stdscale=1,
        fresnel_f0=0.04, compute_dtype=dtype, batch_size=batch, lr=5e-4, ds_lambda=10.0)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def synthetic_batch(R, seed, device):
    """Satellite-shaped rays: normalised scene cube, near-nadir view, constant near/far/sun."""
    g = torch.Generator().manual_seed(seed)
    o = torch.cat([torch.rand(R, 2, generator=g) * 2 - 1, 1.0 + 0.02 * torch.rand(R, 1, generator=g)], -1)
    el = torch.deg2rad(torch.tensor(75.0))
    d = torch.stack([torch.cos(el) * 0.6, torch.cos(el) * 0.8, -torch.sin(el)]).expand(R, 3)
    se, sa = torch.deg2rad(torch.tensor(55.0)), torch.deg2rad(torch.tensor(130.0))
    sun = torch.stack([torch.cos(se) * torch.cos(sa), torch.cos(se) * torch.sin(sa), torch.sin(se)]).expand(R, 3)
    rays = torch.cat([o, d, torch.zeros(R, 1), torch.full((R, 1), 2.0), sun], -1).float().contiguous()
    batch = dict(rays=rays, rgbs=torch.rand(R, 3, generator=g), valid_depth=(torch.rand(R, generator=g) < 0.7).float(),
                 depths=torch.stack([0.8 + 0.4 * torch.rand(R, generator=g), torch.rand(R, generator=g)], -1),
                 depth_std=torch.zeros(R))                       # reference quirk 7: target_std == 0 in training
    return {k: v.to(device) for k, v in batch.items()}


def box_calibration(dev, n=8192, seconds=1.0, when="after the bench"):
    """How fast is THIS box?  The boxes of the pool run the same binary up to 10 % apart (round 4, DESIGN.md section 7:
    MI355X_MICROARCH.md DVFS give-back item 5), so a bench line is read beside a workload that does not depend on this
    repository: ~1 s of back-to-back vendor-library bf16 GEMMs (n x n x n, random operands), after everything that is timed.
    Not part of `value`; never used by the product path."""
    try:
        x = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        y = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        for _ in range(5):
            torch.matmul(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.matmul(x, y)
        torch.cuda.synchronize()
        reps = max(10, int(seconds / max(time.perf_counter() - t0, 1e-4)))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            torch.matmul(x, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        return {"kind": f"torch.matmul bf16 {n}x{n}x{n} (vendor GEMM library), {reps} back-to-back calls {when}", "ms": ms,
                "tflops": 2.0 * n ** 3 / (ms * 1e-3) / 1e12,
                "note": "box-speed indicator: compare bench lines of different boxes through it, not a property of this repository"}
    except Exception as e:      # never let the indicator break the bench line
        return {"error": repr(e)}


def cpu_baseline(args, rays=512, seconds_budget=30.0):
    """Time the CPU oracle (port of the reference's PyTorch path) on a bounded sample: same network, same S/G, `rays` rays
    per step.  profiles/history/r02_cpu_cross_timing.txt holds the oracle-vs-imported-reference timing of the same step taken in
    the build container: the stand-in is within a few per cent of the reference itself.
    Bounded (round 4's sweep spent ~150 s in a 256-thread leg): 8 / 16 / min(allowed, 64) threads, each leg
    one warm-up step (itself timed, and taken as the leg's result when it alone exceeds the leg's budget) and at most two steps
    inside seconds_budget / 3."""
    from oracle.config import FieldConfig
    from oracle import render as ORD, losses as OL
    allowed = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))        # the cores this process may use (the box's allotment for one GPU)
    except Exception:
        pass
    cfg = FieldConfig(feat=args.fc_feat, layers=args.fc_layers, n_samples=args.n_samples, guided_samples=args.guided_samples)
    params = {k: torch.from_numpy(v).requires_grad_(True) for k, v in cfg.make_params(0).items()}
    opt = torch.optim.Adam(list(params.values()), lr=args.lr)
    b = synthetic_batch(rays, 123, "cpu")

    def step():
        opt.zero_grad(set_to_none=True)
        res, _ = ORD.render_rays(params, cfg, b["rays"], ORD.Randoms(), mode="train", valid_depth=b["valid_depth"],
                                 target_depths=b["depths"], target_std=b["depth_std"])
        loss = OL.snerf_loss(res, b["rgbs"]) + OL.depth_loss(res, b["depths"][:, 0], b["depths"][:, 1], b["valid_depth"],
                                                              b["depth_std"], args.ds_lambda)
        loss.backward()
        opt.step()

    leg = seconds_budget / 3
    sweep, t_all = {}, time.perf_counter()
    for th in sorted({min(8, allowed), min(16, allowed), min(64, allowed)}):
        torch.set_num_threads(th)
        t0 = time.perf_counter()
        step()                                                   # warm-up (allocator, thread pool)
        warm = time.perf_counter() - t0
        if warm > leg:                                           # a leg this slow is not the best one: its warm-up step is its sample
            sweep[th] = (rays / warm, 1, True)
            continue
        t0, n = time.perf_counter(), 0
        while n < 2 and (n == 0 or (time.perf_counter() - t0) * (n + 1) / n + warm < leg):
            step()
            n += 1
        sweep[th] = (rays * n / (time.perf_counter() - t0), n, False)
    cores = max(sweep, key=lambda k: sweep[k][0])
    rate, n, warm_only = sweep[cores]
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.lower().startswith("model name")), "unknown")
    except OSError:
        pass
    return dict(value=rate, unit="rays/s", cores=cores, kind="port", cpu_model=cpu_model, threads=cores,
                host_logical_cpus=os.cpu_count(), allowed_cpus=allowed,
                sweep={str(k): round(v[0], 1) for k, v in sweep.items()}, seconds=round(time.perf_counter() - t_all, 1),
                sample=f"{n} training step(s) of {rays} rays x {args.n_samples}+{args.guided_samples} samples (same network, fp32, "
                       f"torch CPU oracle, {cores} threads)" + (" = the leg's first step" if warm_only else " after 1 warm-up step"),
                cross_timing="profiles/history/r02_cpu_cross_timing.txt (oracle vs the imported reference, build container)")


from brdf_nerf_amd.build import source_hash  # noqa: E402  (key of the committed PMC passes)


def pmc_record(config, dtype):
    """Per-kernel counters of the committed rocprofv3 PMC passes (profiles/pmc_collect.sh -> profiles/r05_pmc.json), or
    ({}, reason) when there is none for this workload or the kernel sources have changed since it was taken."""
    # the newest committed record whose key matches the sources wins (one per round)
    cands = [os.path.join(ROOT, "profiles", "r05_pmc.json"), os.path.join(ROOT, "profiles", "r04_pmc.json")]
    cands = [c for c in cands if os.path.exists(c)]
    if not cands:
        return {}, "no PMC pass committed"
    recs = [json.load(open(c)) for c in cands]
    rec = next((r for r in recs if r.get("source_hash") == source_hash()), recs[0])
    if rec.get("source_hash") != source_hash():
        return {}, f"PMC pass is of other kernel sources ({rec.get('source_hash')}): re-run profiles/pmc_collect.sh"
    runs = rec.get("workloads", {})
    key = f"{config}_{dtype}"
    if key not in runs:
        return {}, f"no PMC pass for workload {key}"
    return runs[key], "rocprofv3 --pmc, one counter group per pass over profiles/prof_step.py (profiles/pmc_collect.sh); " \
                      "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE per launch (gfx950 correction, MI355X_MICROARCH.md section HBM)"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU per step (weak) / per global batch (strong)")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--guided", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--config", default="lambert", choices=list(CONFIG_FLAGS),
                    help="lambert = BASELINE config 2 (headline); rpv_nan = config 3 (RPV + analytic normals; with --samples 128 "
                         "--rays 1024: config 4's per-GPU shape); hapke / microfacet = the two halves of config 5 (use --dtype fp16); "
                         "rpv_nlr = RPV with learned normals")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--settle-seconds", type=float, default=1.5, help="minimum time of STEADY-STATE steps before the timed region")
    ap.add_argument("--sustained-steps", type=int, default=0, help="steps of the `sustained` run after the timed region (0: as many as "
                                                                  "--sustained-seconds take)")
    ap.add_argument("--sustained-seconds", type=float, default=10.0, help="length of the `sustained` run when --sustained-steps is 0 "
                                                                          "(0: none); long enough for an external GPU-busy sampler")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as fresh child processes (this parent never touches
    the GPU, and nothing is re-exec'ed), rank 0 inherits stdout for the JSON line."""
    have = torch.cuda.device_count()           # (may bring the HIP runtime up in this parent; the ranks are fresh child processes)
    share = os.environ.get("BN_BENCH_SHARE_GPU") == "1"
    if have < a.gpus and not share:
        sys.exit(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + 3000
    while procs and time.time() < deadline:
        for p in list(procs):
            c = p.poll()
            if c is None:
                continue
            procs.remove(p)
            if c != 0:
                rc = rc or c
                for q in procs:            # a rank died: the others would wait in a collective forever
                    q.terminate()
        time.sleep(0.2)
    for p in procs:
        p.kill()
        rc = rc or 1
    sys.exit(rc)


def main():
    a = parse_args()
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if a.gpus > 1 and not under_launcher:
        spawn_ranks(a)
        return
    # stdout carries ONE JSON line and nothing else: libraries that chat on fd 1 (gloo's "[Gloo] Rank 0 is connected ...",
    # RCCL with NCCL_DEBUG set) are sent to stderr for the whole run; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # Rehearsal hooks for a one-GPU box (the N > 1 path end to end on the real kernels): BN_BENCH_SHARE_GPU=1 puts every
    # rank on cuda:0, BN_BENCH_BACKEND=gloo replaces RCCL (which wants one GPU per rank).  Never set by the driver.
    if os.environ.get("BN_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("BN_BENCH_BACKEND", "nccl")
    dist = torch.distributed
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
        assert dist.get_backend() == backend, dist.get_backend()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from brdf_nerf_amd import load_model, _lib
    from brdf_nerf_amd.trainer import FusedTrainer

    over, flags, base_cfg = CONFIG_FLAGS[a.config]
    rays_gpu = a.rays if a.scaling == "weak" else a.rays // world
    assert rays_gpu >= 1 and (a.scaling == "weak" or rays_gpu * world == a.rays), "strong scaling: --rays must divide by --gpus"
    args = make_args(rays_gpu, a.samples, a.guided, a.dtype, **over)
    torch.manual_seed(0)
    model = load_model(args).to(dev)
    # own draws (strict_rng=False), as brdf_nerf_amd.train.TrainLoop runs it: the launch-lean step with in-kernel draws, replayed
    # from a HIP graph once its inputs have kept their addresses for a few steps
    trainer = FusedTrainer(model, args, lr=args.lr, ds_lambda=args.ds_lambda, strict_rng=False)
    if a.scaling == "weak":
        batches = [synthetic_batch(rays_gpu, 1000 * rank + i + 1, dev) for i in range(4)]
    else:       # strong: the SAME global batches on every world size, each rank takes its contiguous share
        from brdf_nerf_amd.distributed import shard_bounds
        lo, hi = shard_bounds(a.rays, rank, world)
        batches = [{k: v[lo:hi].contiguous() for k, v in synthetic_batch(a.rays, i + 1, dev).items()} for i in range(4)]

    def run(i):
        b = batches[i % len(batches)]
        return trainer.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                            near_far=(0.0, 2.0), **flags)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def agree_max(x):
        if world == 1:
            return x
        t = torch.tensor([float(x)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(fn, n):
        """n steps between barrier + synchronize pairs, nothing else inside; max over ranks."""
        barrier()
        t0 = time.perf_counter()
        out = None
        for i in range(n):
            out = fn(i)
        barrier()
        return agree_max(time.perf_counter() - t0), out

    def settle_then_time(fn, warmup, steps):
        """W warm-up steps (lazy initialisation, allocator, graph capture), then the per-step time from two STEADY-STATE steps,
        then settling steps until the chip has run >= settle_seconds of steady-state steps (its clock settles over seconds),
        then exactly `steps` timed steps."""
        barrier()
        for i in range(warmup):
            fn(i)
        est, _ = timed(fn, 2)
        est /= 2
        n_settle = min(int(math.ceil(a.settle_seconds / max(est, 1e-5))), 20000) if a.settle_seconds > 0 else 0
        for i in range(n_settle):
            fn(i)
        dt_, out = timed(fn, steps)
        return dt_, out, n_settle

    # box-speed indicator BEFORE anything is timed (and again after everything, below): both are printed
    calib_before = box_calibration(dev, when="before the warm-up steps") if world == 1 else None
    dt, (loss, _), settle = settle_then_time(run, a.warmup, a.steps)
    # ---- sustained: a longer run after the timed region, outside `value` (does the step hold its time?)
    sustained = None
    n_sus = a.sustained_steps if a.sustained_steps > 0 else int(math.ceil(a.sustained_seconds / max(dt / a.steps, 1e-5)))
    if n_sus > 0:
        ds, _ = timed(run, n_sus)
        sustained = {"steps": n_sus, "seconds": ds, "ms_per_step": ds / n_sus * 1e3,
                     "value": world * rays_gpu * n_sus / ds, "unit": "rays/s"}
    # ---- N > 1, weak scaling asked: the strong-scaling answer in the same invocation - ONE 4096-ray batch split N ways, and
    # the same batch on ONE rank's shape (every rank runs it alone, no collective) for the speed-up, both timed in this run
    strong = None
    if world > 1 and a.scaling == "weak" and a.rays % world == 0:
        from brdf_nerf_amd.distributed import shard_bounds
        lo, hi = shard_bounds(a.rays, rank, world)
        args_s = make_args(hi - lo, a.samples, a.guided, a.dtype, **over)
        torch.manual_seed(0)
        tr_s = FusedTrainer(load_model(args_s).to(dev), args_s, lr=args_s.lr, ds_lambda=args_s.ds_lambda, strict_rng=False)
        tr_s.ray_offset = lo
        bs = [{k: v[lo:hi].contiguous() for k, v in synthetic_batch(a.rays, i + 1, dev).items()} for i in range(4)]
        run_s = lambda i: tr_s.step(bs[i % 4]["rays"], bs[i % 4]["rgbs"], valid_depth=bs[i % 4]["valid_depth"], depths=bs[i % 4]["depths"],
                                    depth_std=bs[i % 4]["depth_std"], near_far=(0.0, 2.0), **flags)
        dt_s, _, _ = settle_then_time(run_s, a.warmup, a.steps)
        torch.manual_seed(0)
        tr_1 = FusedTrainer(load_model(args).to(dev), args, lr=args.lr, ds_lambda=args.ds_lambda, strict_rng=False, data_parallel=False)
        b1 = [synthetic_batch(a.rays, i + 1, dev) for i in range(4)]
        run_1 = lambda i: tr_1.step(b1[i % 4]["rays"], b1[i % 4]["rgbs"], valid_depth=b1[i % 4]["valid_depth"], depths=b1[i % 4]["depths"],
                                    depth_std=b1[i % 4]["depth_std"], near_far=(0.0, 2.0), **flags)
        dt_1, _, _ = settle_then_time(run_1, a.warmup, a.steps)
        strong = {"rays_per_global_step": a.rays, "rays_per_gpu": hi - lo, "ms_per_step": dt_s / a.steps * 1e3,
                  "value": a.rays * a.steps / dt_s, "unit": "rays/s",
                  "n1_shape_ms_per_step": dt_1 / a.steps * 1e3, "speedup_vs_n1_ms": dt_1 / dt_s,
                  "note": "one 4096-ray batch split over the ranks (gradient all-reduce in two overlapped buckets) against the same "
                          "batch on one rank's shape, every rank running it alone, both timed in this invocation"}
        del tr_s, tr_1
    # ---- separate pass: per-kernel durations from HIP events around every launch (on the launch stream)
    prof_steps = max(10, min(a.steps, 20))
    trainer.use_graph = False            # a graph replay does not pass through the library's launch sites: this pass runs eagerly
    _lib.prof_enable(True)
    for i in range(prof_steps):
        run(i)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    trainer.use_graph = True
    prof = _lib.prof_collect()
    dropped = trainer.dropped_grad_elems()
    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    n_heads = len(model.head_list(flags["apply_brdf"], flags["apply_theta"]))
    F, P, Lh = 512, 60, 8
    fpp = {False: flops_per_point(n_heads=n_heads), True: flops_per_point(n_heads=n_heads, executed=True)}
    if model.normal in ("analystic", "analystic_learned"):   # adjoint chain + its backward (transposed / forward trunk products)
        for f in fpp.values():
            f["field_adjoint"] = 2 * ((Lh - 1) * F * F + 2 * P * F)
            f["field_adjoint_bwd"] = 2 * (P * F + (Lh - 2) * F * F + (F + P) * F)
            f["wgrad"] += 2 * (P * F + (Lh - 2) * F * F + (F + P) * F)
    M1, M2 = rays_gpu * a.samples, rays_gpu * (a.samples + a.guided)
    pts = dict(field_fwd_sigma=M1, field_fwd_full=M2, field_bwd_chain=M2, wgrad=M2, skinny_wgrad=M2, field_adjoint=M2,
               field_adjoint_bwd=M2)
    peak = PEAK_F32_TFLOPS if a.dtype == "fp32" else PEAK_MFMA16_TFLOPS
    pmc, pmc_note = pmc_record(a.config, a.dtype) if (rays_gpu, a.samples, a.guided) == (4096, 64, 64) else ({}, "PMC passes are of the 4096 x 64+64 shape")
    # algorithmic bytes per launch of the HBM-bound per-ray kernels
    C_out = model.spec(flags["apply_brdf"], flags["apply_theta"], trainer.nr_lr, trainer.nr_an).out_channels
    S2 = a.samples + a.guided
    hbm_bytes = {
        "composite_fwd": rays_gpu * (2 * S2 * 4 + S2 * C_out * 4 + 3 * S2 * 4 + 4 + C_out * 4),   # z, sigma, chan in; alpha, T, w, depth, acc out
        "composite_bwd": rays_gpu * (2 * S2 * 4 + S2 * C_out * 4 + S2 * 4 + 4 + C_out * 4 + S2 * C_out * 4),
        "guided_samples": rays_gpu * (2 * a.samples * 4 + 4 + a.guided * 4 + a.guided * 4 + S2 * 4 + S2 * 8),
        "adam": trainer.flat_param.numel() * 4 * 7,                                               # p, g, m, v in; p, m, v out
    }
    kernels = {}
    for name, (ms, cnt) in prof.items():
        k = dict(ms_per_launch=ms / cnt, launches_per_step=cnt / prof_steps)
        if name in fpp[False]:      # pts[name] = points per STEP through this kernel (however many launches they are split over)
            k["tflops"] = fpp[False][name] * pts[name] * prof_steps / (ms * 1e-3) / 1e12
            k["tflops_executed"] = fpp[True][name] * pts[name] * prof_steps / (ms * 1e-3) / 1e12
            k["frac_of_peak"] = k["tflops"] / peak
            k["frac_of_peak_executed"] = k["tflops_executed"] / peak
        if name in hbm_bytes:       # per-ray kernels: algorithmic bytes per launch / duration (first launch shape of the step)
            k["algorithmic_GBs"] = hbm_bytes[name] / (ms / cnt * 1e-3) / 1e9
            k["frac_of_hbm_peak"] = k["algorithmic_GBs"] / PEAK_HBM_GBS
        if name in pmc:
            k["pmc"] = pmc[name]
        kernels[name] = k
    half_step = lambda n: kernels[n]["ms_per_launch"] * kernels[n]["launches_per_step"] / 2      # ms per M2 / 2 points
    mfma = {n: k for n, k in kernels.items() if "tflops" in k and n != "skinny_wgrad"}
    dom = max(mfma, key=lambda n: mfma[n]["ms_per_launch"] * mfma[n]["launches_per_step"])
    dpm = pmc.get(dom, {})
    flops_step = {e: sum(fpp[e][n] * pts[n] for n in fpp[e] if n in kernels) for e in (False, True)}
    # the reference pipeline's algorithmic work (pass 1 sigma-only + pass 2 on all S+G samples);
    # the fused trainer evaluates each sample once (pass 1 is kept and reused), so it executes less than this
    flops_ref = sum(fpp[False][n] * pts[n] for n in fpp[False])
    step_s = dt / a.steps
    rays_step = world * rays_gpu
    # SURVEY.md section 8(d), per train ray of the reference-default pipeline: S sigma-only forwards + (S + G) x 3 x (full forward
    # [+ the analytic-normal adjoint]): 2.002 GFLOP (Lambertian), 3.761 GFLOP (RPV111 + analytic normals, fused accounting)
    per_ray_ref = a.samples * fpp[False]["field_fwd_sigma"] + (a.samples + a.guided) * 3 * (
        fpp[False]["field_fwd_full"] + fpp[False].get("field_adjoint", 0))
    line = {
        "metric": METRIC, "value": rays_step * a.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
        "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"BASELINE config {base_cfg}: Djibouti-shaped synthetic rays, spsbrdf-nerf {a.config} train step "
                               f"(pass1 {a.samples} + guided {a.guided} samples/ray, F=512, 8 Siren layers, PE10, ds_lambda=10), "
                               f"{rays_gpu} rays/GPU/step ({rays_step} rays per global step, {a.scaling} scaling)",
                   "rays_per_gpu": rays_gpu, "n_samples": a.samples, "guided_samples": a.guided, "parallelism": f"dp{world}",
                   "backend": (backend if world > 1 else None), "settle_steps": settle},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": mfma[dom]["tflops"], "peak": peak, "unit": "TFLOP/s",
                     "frac": mfma[dom]["tflops"] / peak, "frac_algorithmic": mfma[dom]["tflops"] / peak,
                     # the OTHER accounting, never to be confused with the kernel fraction: rays/s x SURVEY section 8(d)'s FLOP per
                     # train ray of the REFERENCE pipeline (pass 1 sigma-only + pass 2 on all S + G samples, x 3 for the backward:
                     # 2.002 GFLOP per Lambertian ray) / peak - the build evaluates every sample once, so it executes less
                     "frac_per_ray_accounting": per_ray_ref * rays_step / step_s / 1e12 / peak,
                     "flop_per_ray_reference_accounting": per_ray_ref,
                     "achieved_executed": mfma[dom]["tflops_executed"], "frac_executed": mfma[dom]["tflops_executed"] / peak,
                     "mfma_busy": dpm.get("mfma_busy"), "traffic": dpm.get("hbm_bytes"),
                     "traffic_unit": "bytes/launch", "traffic_live": False, "traffic_source": pmc_note,
                     "note": "achieved = algorithmic FLOPs per launch (reference network, DESIGN.md section 3) / mean launch "
                             "duration from HIP events on the launch stream, taken in a pass of its own after the timed region; "
                             "_executed excludes the folded feats layer, which the build does not run"},
        # SURVEY.md section 8d kernel-level figure: the fused MLP on M = rays x samples rows (one launch of each kernel)
        "mlp_microbench": {
            "rows": M2 // 2,
            "fwd_ms": half_step("field_fwd_full"),
            "fwd_tflops": fpp[False]["field_fwd_full"] * (M2 // 2) / (half_step("field_fwd_full") * 1e-3) / 1e12,
            "fwd_bwd_ms": sum(half_step(n) for n in ("field_fwd_full", "field_bwd_chain", "wgrad", "skinny_wgrad") if n in kernels),
            "fwd_bwd_tflops": 3.0 * fpp[False]["field_fwd_full"] * (M2 // 2) /
                              (sum(half_step(n) for n in ("field_fwd_full", "field_bwd_chain", "wgrad", "skinny_wgrad")
                                   if n in kernels) * 1e-3) / 1e12,
            "note": "per M2 / 2 rows = half of a step's points (the backward kernels run once over both passes: half of their launch); "
                    "fwd = full forward with activation stash; fwd_bwd = forward + backward chain + weight gradients, 3x the "
                    "forward's algorithmic FLOPs (dX + dW), analytic-normal kernels not included",
        } if "field_fwd_full" in kernels else None,
        "step_tflops": flops_step[False] / step_s / 1e12, "step_frac_of_peak": flops_step[False] / step_s / 1e12 / peak,
        "step_tflops_executed": flops_step[True] / step_s / 1e12, "step_frac_of_peak_executed": flops_step[True] / step_s / 1e12 / peak,
        "step_tflops_reference_accounting": flops_ref / step_s / 1e12,
        "kernels": kernels, "final_loss": float(loss), "dropped_nonfinite_grad_elems": list(dropped),
        "sustained": sustained, "strong": strong,
        "launches_per_step": sum(k["launches_per_step"] for k in kernels.values()),
    }
    if world == 1:
        line["box_calibration"] = box_calibration(dev, when="after everything that is timed")
        line["box_calibration_before"] = calib_before
    if world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(line) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
