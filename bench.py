#!/usr/bin/env python3
"""Headline benchmark: spsbrdf-nerf TRAIN rays/s on MI355X (BASELINE.json metric).

One "step" = one full training step on a batch of R rays per GPU:
  render_rays (pass 1 sigma-only on S samples, depth-guided resampling, pass 2 on S+G samples)
  + SNerfLoss + DepthLoss (ds_lambda=10, README stage 1) + backward + [RCCL grad all-reduce] + Adam.
Workload at N=1: BASELINE config 2 - Lambertian pretrain, 4096 rays x 64 samples (+64 guided), F=512, 8 Siren layers,
PE(10), bf16 MFMA, synthetic satellite-shaped rays, random-init weights (seed 0).
N>1: weak scaling, 4096 rays per GPU, one process per GPU, gradients all-reduced over RCCL.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel (HIP-event timed inside the timed region);
`cpu_baseline` times the CPU oracle (a port of the reference's PyTorch path) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3       # fp32-input MFMA
# HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_traffic.txt); valid for the
# default workload only (lambert, 4096 rays, 64+64 samples, bf16), otherwise `traffic` is null
# HBM-side bytes per launch of the default workload (262,144 points per launch, two launches per step), from rocprofv3
# PMC passes over profiles/prof_step.py: 2 x FETCH_SIZE + WRITE_SIZE (profiles/r01_pmc_traffic.txt)
PMC_TRAFFIC_BYTES = {"field_fwd_full": 4.87e9, "field_bwd_chain": 4.75e9, "wgrad": 6.63e9}


def flops_per_point(F=512, P=60, L=8, n_heads=1):
    """Algorithmic FLOPs (2*MAC) per sample point, by kernel."""
    H2 = F // 2
    trunk = 2 * (P * F + (L - 2) * F * F + (F + P) * F)
    heads1 = n_heads * 2 * F * H2
    heads2 = 2 * 3 * H2 + (n_heads - 1) * 2 * H2          # rgb (3 outputs) + 1-wide BRDF heads
    fwd_sigma = trunk + 2 * F
    fwd_full = trunk + 2 * F + 2 * F * F + heads1 + heads2
    bwd_chain = heads1 + heads2 + 2 * F * F + 2 * F + (L - 1) * 2 * F * F     # dX products (h inputs only)
    wgrad = trunk + 2 * F * F + heads1
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


def cpu_baseline(args, seconds_budget=25.0):
    """Time the CPU oracle (port of the reference's PyTorch path) on a bounded sample: same network, same S/G, fewer rays."""
    from oracle.config import FieldConfig
    from oracle import render as ORD, losses as OL
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)          # a 1-GPU box share is 16 host cores; more threads only oversubscribe
    torch.set_num_threads(cores)
    cfg = FieldConfig(feat=args.fc_feat, layers=args.fc_layers, n_samples=args.n_samples, guided_samples=args.guided_samples)
    params = {k: torch.from_numpy(v).requires_grad_(True) for k, v in cfg.make_params(0).items()}
    opt = torch.optim.Adam(list(params.values()), lr=args.lr)
    R = 512
    b = synthetic_batch(R, 123, "cpu")

    def step():
        opt.zero_grad(set_to_none=True)
        res, _ = ORD.render_rays(params, cfg, b["rays"], ORD.Randoms(), mode="train", valid_depth=b["valid_depth"],
                                 target_depths=b["depths"], target_std=b["depth_std"])
        loss = OL.snerf_loss(res, b["rgbs"]) + OL.depth_loss(res, b["depths"][:, 0], b["depths"][:, 1], b["valid_depth"],
                                                              b["depth_std"], args.ds_lambda)
        loss.backward()
        opt.step()

    step()                                                       # warm-up (allocator, thread pool)
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 > seconds_budget * 0.6 or n >= 8:
            break
    dt = time.perf_counter() - t0
    return dict(value=R * n / dt, unit="rays/s", cores=cores, kind="port",
                sample=f"{n} training steps of {R} rays x {args.n_samples}+{args.guided_samples} samples (same network, fp32, "
                       f"torch CPU oracle, {cores} threads) after 1 warm-up step")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096, help="rays per GPU per step")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--guided", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--config", default="lambert", choices=["lambert", "rpv_nlr", "rpv_nan"],
                    help="lambert = BASELINE config 2 (headline); rpv_nan = config 3 (RPV + analytic normals); rpv_nlr = learned normals")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal hooks for a one-GPU box (the N > 1 path end to end on the real kernels): BN_BENCH_SHARE_GPU=1 puts every
    # rank on cuda:0, BN_BENCH_BACKEND=gloo replaces RCCL (which wants one GPU per rank).  Never set by the driver.
    if os.environ.get("BN_BENCH_SHARE_GPU") == "1":
        local = 0
    backend = os.environ.get("BN_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            torch.distributed.init_process_group(backend)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from brdf_nerf_amd import load_model, _lib
    from brdf_nerf_amd.trainer import FusedTrainer

    over = {}
    flags = dict(apply_brdf=False, apply_theta=False, cos_irra_on=False)
    if a.config in ("rpv_nlr", "rpv_nan"):
        over = dict(funcM=1, funcF=1, funcH=1, normal="learned" if a.config == "rpv_nlr" else "analystic")
        flags = dict(apply_brdf=True, apply_theta=True, cos_irra_on=True)
    args = make_args(a.rays, a.samples, a.guided, a.dtype, **over)
    torch.manual_seed(0)
    model = load_model(args).to(dev)
    trainer = FusedTrainer(model, args, lr=args.lr, ds_lambda=args.ds_lambda)
    batches = [synthetic_batch(a.rays, 1000 * rank + i + 1, dev) for i in range(4)]

    def run(i):
        b = batches[i % len(batches)]
        return trainer.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                            near_far=(0.0, 2.0), **flags)

    for i in range(a.warmup):
        run(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    _lib.prof_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss, _ = run(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    _lib.prof_enable(False)
    prof = _lib.prof_collect()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    n_heads = 1 + (3 if a.config != "lambert" else 0)
    fpp = flops_per_point(n_heads=n_heads)
    if a.config == "rpv_nan":       # analytic normals: adjoint chain + its backward (transposed / forward trunk products)
        F, P, Lh = 512, 60, 8
        fpp["field_adjoint"] = 2 * ((Lh - 1) * F * F + 2 * P * F)
        fpp["field_adjoint_bwd"] = 2 * (P * F + (Lh - 2) * F * F + (F + P) * F)
        fpp["wgrad"] += 2 * (P * F + (Lh - 2) * F * F + (F + P) * F)
    M1, M2 = a.rays * a.samples, a.rays * (a.samples + a.guided)
    pts = dict(field_fwd_sigma=M1, field_fwd_full=M2, field_bwd_chain=M2, wgrad=M2, skinny_wgrad=M2, field_adjoint=M2,
               field_adjoint_bwd=M2)
    peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
    kernels = {}
    for name, (ms, cnt) in prof.items():
        k = dict(ms_per_launch=ms / cnt, launches_per_step=cnt / a.steps)
        if name in fpp:      # pts[name] = points per STEP through this kernel (however many launches they are split over)
            k["tflops"] = fpp[name] * pts[name] * a.steps / (ms * 1e-3) / 1e12
            k["frac_of_peak"] = k["tflops"] / peak
        kernels[name] = k
    mfma = {n: k for n, k in kernels.items() if "tflops" in k and n != "skinny_wgrad"}
    dom = max(mfma, key=lambda n: mfma[n]["ms_per_launch"] * mfma[n]["launches_per_step"])
    # algorithmic FLOPs of the launched kernels (the network as the reference defines it: the linear feats layer is counted
    # although the build folds it into the heads and does not execute it)
    flops_step = sum(fpp[n] * pts[n] for n in fpp if n in kernels)
    # the reference pipeline's algorithmic work (pass 1 sigma-only + pass 2 on all S+G samples);
    # the fused trainer evaluates each sample once (pass 1 is kept and reused), so it executes less than this
    flops_ref = sum(fpp[n] * pts[n] for n in fpp)
    line = {
        "metric": "train rays/sec (+ MFMA% of roofline), spsbrdf-nerf 64 samples/ray, 1/2/4/8 MI355X",
        "value": world * a.rays * a.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
        "data": "synthetic",
        "config": {"workload": f"BASELINE config {3 if a.config == 'rpv_nan' else 2}: Djibouti-shaped synthetic rays, spsbrdf-nerf {a.config} train step "
                               f"(pass1 {a.samples} + guided {a.guided} samples/ray, F=512, 8 Siren layers, PE10, ds_lambda=10), "
                               f"{a.rays} rays/GPU/step", "rays_per_gpu": a.rays, "n_samples": a.samples,
                   "guided_samples": a.guided, "parallelism": f"dp{world}"},
        "roofline": {"bound": "mfma", "kernel": dom, "achieved": mfma[dom]["tflops"], "peak": peak, "unit": "TFLOP/s",
                     "frac": mfma[dom]["tflops"] / peak,
                     "traffic": PMC_TRAFFIC_BYTES.get(dom) if (a.config, a.rays, a.samples, a.guided, a.dtype) ==
                     ("lambert", 4096, 64, 64, "bf16") else None,
                     "traffic_unit": "bytes/launch (rocprofv3 PMC, offline pass: profiles/r01_pmc_traffic.txt)"},
        # SURVEY.md section 8d kernel-level figure: the fused MLP on M = rays x samples rows (one launch of each kernel)
        "mlp_microbench": {
            "rows": M2 // 2,
            "fwd_ms": kernels["field_fwd_full"]["ms_per_launch"],
            "fwd_tflops": fpp["field_fwd_full"] * (M2 // 2) / (kernels["field_fwd_full"]["ms_per_launch"] * 1e-3) / 1e12,
            "fwd_bwd_ms": sum(kernels[n]["ms_per_launch"] for n in ("field_fwd_full", "field_bwd_chain", "wgrad", "skinny_wgrad")
                              if n in kernels),
            "fwd_bwd_tflops": 3.0 * fpp["field_fwd_full"] * (M2 // 2) /
                              (sum(kernels[n]["ms_per_launch"] for n in ("field_fwd_full", "field_bwd_chain", "wgrad", "skinny_wgrad")
                                   if n in kernels) * 1e-3) / 1e12,
            "note": "fwd = full forward with activation stash; fwd_bwd = forward + backward chain + weight gradients, 3x the "
                    "forward's algorithmic FLOPs (dX + dW), analytic-normal kernels not included",
        } if "field_fwd_full" in kernels else None,
        "step_tflops": flops_step / (dt / a.steps) / 1e12, "step_frac_of_peak": flops_step / (dt / a.steps) / 1e12 / peak,
        "step_tflops_reference_accounting": flops_ref / (dt / a.steps) / 1e12,
        "kernels": kernels, "final_loss": float(loss),
    }
    if world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
