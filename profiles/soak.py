"""Soak: thousands of fused training steps per (config, dtype) at the BASELINE shapes - finite losses and gradients throughout, the
device fault word clear, no growth of the allocator's footprint, steady step time.
    python profiles/soak.py [steps] [only]      only: a substring of "config dtype variant" (e.g. multibrdf)"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd import _lib, load_model  # noqa: E402
from brdf_nerf_amd.trainer import FusedTrainer  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    dev = torch.device("cuda", 0)
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    runs = [("lambert", "bf16", ""), ("lambert", "fp16", ""), ("rpv_nan", "bf16", ""), ("hapke", "fp16", ""), ("microfacet", "fp16", ""),
            # round 5: one BRDF per sample (the per-sample shading launch, csrc/sample_brdf.hip), alone and with the sun pass's
            # per-sample irradiance in the gsam_only stage
            ("rpv_nan", "bf16", "multibrdf"), ("microfacet", "fp16", "multibrdf"), ("hapke", "fp16", "multibrdf_sun_gsam")]
    for config, dtype, variant in runs:
        if only and only not in f"{config} {dtype} {variant}":
            continue
        over = dict(bench.CONFIG_FLAGS[config][0])
        if "multibrdf" in variant:
            over["MultiBRDF"] = 1
        if "sun" in variant:
            over["sun_v"] = "analystic"
        args = bench.make_args(4096, 64, 64, dtype, **over)
        torch.manual_seed(0)
        model = load_model(args).to(dev)
        tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
        flags = dict(bench.CONFIG_FLAGS[config][1], **({"gsam_only": True} if "gsam" in variant else {}))
        batches = [bench.synthetic_batch(4096, s, dev) for s in range(4)]
        n = steps if config == "lambert" else steps // 3
        torch.cuda.synchronize()
        mem0 = None
        t0 = time.time()
        worst = 0.0
        marks = []
        for i in range(n):
            b = batches[i % 4]
            loss, _ = tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"], near_far=(0.0, 2.0), **flags)
            if i == 20:
                torch.cuda.synchronize()
                mem0 = torch.cuda.memory_reserved()
            if i % 100 == 99:
                torch.cuda.synchronize()
                marks.append(time.time())
                lv = float(loss)
                assert lv == lv and abs(lv) < 1e6, (config, dtype, variant, i, lv)
                assert bool(torch.isfinite(tr.flat_grad).all()) and bool(torch.isfinite(tr.flat_param).all()), (config, dtype, variant, i)
        torch.cuda.synchronize()
        faults = C.c_uint(0)
        _lib.check(_lib.lib().bn_device_faults(C.byref(faults), None), "bn_device_faults")
        per = [(b_ - a_) / 100 * 1e3 for a_, b_ in zip(marks[:-1], marks[1:])]
        print(f"{config} {dtype}{' ' + variant if variant else ''}: {n} steps, final loss {float(loss):.5f}, faults {faults.value}, dropped {tr.dropped_grad_elems()}, reserved "
              f"{mem0 / 2**30:.2f} -> {torch.cuda.memory_reserved() / 2**30:.2f} GiB, ms/step per 100-step window min {min(per):.3f} max {max(per):.3f} "
              f"(window {per.index(max(per)) + 1} of {len(per)}; the first four: {' '.join(f'{x:.3f}' for x in per[:4])}; graphs {len(tr._graphs)})", flush=True)
        assert faults.value == 0 and torch.cuda.memory_reserved() <= mem0 * 1.05
        del tr, model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
