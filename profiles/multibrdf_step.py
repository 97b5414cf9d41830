"""Time of a --MultiBRDF training step (one BRDF per sample) at the BASELINE shape (4096 rays x (64 + 64) samples), by model, with
the per-kernel split of the launch-lean step.  Run from two trees to compare builds (the script imports the tree it lies in).
    python profiles/multibrdf_step.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd import load_model, build  # noqa: E402
from brdf_nerf_amd.trainer import FusedTrainer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    dev = torch.device("cuda", 0)
    print(f"tree {ROOT}, sources {build.source_hash()}")
    for config, dtype, reg in (("rpv_nan", "bf16", False), ("hapke", "fp16", False), ("microfacet", "fp16", False), ("rpv_nan", "bf16", True)):
        over, flags, _ = bench.CONFIG_FLAGS[config]
        args = bench.make_args(4096, 64, 64, dtype, **dict(over, MultiBRDF=1))
        torch.manual_seed(0)
        model = load_model(args).to(dev)
        lam = dict(hs_lambda=0.1, nr_reg_an_lambda=0.2) if reg else {}
        tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False, **lam)
        batches = [bench.synthetic_batch(4096, s + 1, dev) for s in range(2)]
        run = lambda i: tr.step(batches[i % 2]["rays"], batches[i % 2]["rgbs"], valid_depth=batches[i % 2]["valid_depth"],
                                depths=batches[i % 2]["depths"], depth_std=batches[i % 2]["depth_std"], near_far=(0.0, 2.0), **flags)
        for i in range(40):
            run(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            loss, _ = run(i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        print(f"MultiBRDF {config} {dtype} regularisers={int(reg)}: {ms:.3f} ms per step ({4096 / ms:.1f} k rays/s), graphs {len(tr._graphs)}, "
              f"loss {float(loss):.5f}", flush=True)
        del tr, model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
