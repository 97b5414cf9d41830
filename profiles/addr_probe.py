"""Does the PLACEMENT of the activation stash change the fused kernels' times?  (round 4, session 41: in profiles/r04_ab_fwd_tail_probes.txt
the identical backward / weight-gradient kernels of five variant libraries differed by up to 5 % in one process.)

One library, K trainers that differ only in where their stash lives: allocation order, a spacer allocated in front, the stash
tensor shifted inside a larger allocation.  Per-kernel HIP-event times over alternating rounds, like profiles/ab_kernels.py.

    python profiles/addr_probe.py [--rounds=3] [--config=lambert]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd import _lib as L  # noqa: E402
from brdf_nerf_amd import load_model  # noqa: E402
from brdf_nerf_amd.trainer import FusedTrainer  # noqa: E402

MAX_SHIFT = 64 << 20


def main():
    opt = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    rounds, config = int(opt.get("rounds", 3)), opt.get("config", "lambert")
    n_rays = int(opt.get("rays", 4096))
    dev = torch.device("cuda", 0)
    # (name, spacer bytes allocated (and kept) before the trainer's buffers, shift of every stash tensor inside its allocation)
    plan = [("plain_a", 0, 0), ("plain_b", 0, 0), ("spacer_1GiB+68KiB", (1 << 30) + (68 << 10), 0), ("shift_256B", 0, 256),
            ("shift_4KiB", 0, 4096), ("shift_68KiB", 0, 68 << 10), ("shift_1MiB", 0, 1 << 20), ("shift_2MiB+4KiB", 0, (2 << 20) + 4096),
            ("plain_c", 0, 0)]
    if "only" in opt:
        plan = [p for p in plan if p[0] in opt["only"].split(",")]
    keep, variants = [], {}
    for name, spacer, shift in plan:
        if spacer:
            keep.append(torch.empty(spacer, dtype=torch.uint8, device=dev))
        args = bench.make_args(n_rays, 64, 64, "bf16", **bench.CONFIG_FLAGS[config][0])
        torch.manual_seed(0)
        model = load_model(args).to(dev)
        tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
        tr.use_graph = False
        plain_buf = tr._buf

        def shifted(key, shape, dtype=torch.float32, _tr=tr, _plain=plain_buf, _shift=shift):
            if not key.startswith("stash") or _shift == 0:
                return _plain(key, shape, dtype)
            n = int(shape[0])
            k = (key, (n,), dtype)              # FusedTrainer._bufs is keyed by (name, shape, dtype)
            b = _tr._bufs.get(k)
            if b is None:
                big = torch.empty(n + MAX_SHIFT, dtype=torch.uint8, device=dev)
                keep.append(big)
                b = big[_shift:_shift + n]
                _tr._bufs[k] = b
            return b

        tr._buf = shifted
        variants[name] = tr
    b = bench.synthetic_batch(n_rays, 1, dev)
    flags = bench.CONFIG_FLAGS[config][1]
    times = {n: {} for n in variants}
    for rnd in range(rounds + 1):
        for name, tr in variants.items():
            L.prof_enable(True)
            for _ in range(3):
                tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                        near_far=(0.0, 2.0), **flags)
            torch.cuda.synchronize()
            prof = L.prof_collect()
            L.prof_enable(False)
            if rnd:
                for k, (ms, n) in prof.items():
                    times[name].setdefault(k, []).append(ms / n)
    keys = ["field_fwd_full", "field_bwd_chain", "wgrad", "skinny_wgrad", "field_adjoint", "field_adjoint_bwd"]
    print(f"config {config} rays {n_rays}: ms per launch, median (min) over {rounds} alternating rounds; stash address of each trainer")
    for name, tr in variants.items():
        st = [(k[0], v.data_ptr(), v.numel()) for k, v in tr._bufs.items() if k[0].startswith("stash")]
        addr = " ".join(f"{k}@0x{p:x} (+{p % (2 << 20)} mod 2MiB, {n / 2**30:.2f} GiB)" for k, p, n in st)
        row = " ".join(f"{k} {statistics.median(times[name][k]):.4f} ({min(times[name][k]):.4f})" for k in keys if k in times[name])
        print(f"{name:>20}: {row} | {addr}")


if __name__ == "__main__":
    main()
