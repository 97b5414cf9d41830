"""In-kernel shader clock of the chain kernels under load (MI355X_MICROARCH.md, DVFS give-back item 6): diagnostic build
with -DBN_CLOCK_STAMP (fwd + bwd chain kernels; add -DBN_CLOCK_STAMP_WGRAD for the weight-gradient kernel instead of the
backward chain), >= 2 s of back-to-back launches on random data, then clock = d s_memtime / d s_memrealtime x 100 MHz per
workgroup (median over the workgroups of the LAST launch of each kernel).

    python -m brdf_nerf_amd.build -DBN_CLOCK_STAMP
    BRDFNERF_HIP_LIB=brdf_nerf_amd/build/BN_CLOCK_STAMP/libbrdfnerf_hip.so python profiles/clock_probe.py [--config=lambert] [--dtype=bf16]
"""
import ctypes as C
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd import _lib as L  # noqa: E402
from brdf_nerf_amd import functions as Fn  # noqa: E402
from brdf_nerf_amd import load_model  # noqa: E402
from brdf_nerf_amd.trainer import FusedTrainer  # noqa: E402


def read_clock(fn_name, n=2048):
    fn = getattr(L.lib(), fn_name, None)
    if fn is None:
        return None
    buf = (C.c_ulonglong * (2 * n))()
    fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.c_int]
    if fn(buf, n) != 0:
        return None
    ghz = [buf[2 * i] / buf[2 * i + 1] * 0.1 for i in range(n) if buf[2 * i + 1] > 0]
    return (statistics.median(ghz), min(ghz), max(ghz), len(ghz)) if ghz else None


def main():
    opt = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    config, dtype = opt.get("config", "lambert"), opt.get("dtype", "bf16")
    dev = torch.device("cuda", 0)
    args = bench.make_args(4096, 64, 64, dtype, **bench.CONFIG_FLAGS[config][0])
    torch.manual_seed(0)
    model = load_model(args).to(dev)
    b = bench.synthetic_batch(4096, 1, dev)
    flags = bench.CONFIG_FLAGS[config][1]
    # (1) sigma-only inference forward, 2 s back to back
    spec = model.spec(False, False, False)
    packed = model.repack(spec)
    z = torch.sort(torch.rand(4096, 128, device=dev) * 2, -1)[0]
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.5:
        for _ in range(50):
            Fn.field_sigma(spec, model.named(), packed, rays=b["rays"], z=z)
        torch.cuda.synchronize()
        n += 50
    print(f"sigma-only forward ({n} launches): in-kernel clock GHz median/min/max/workgroups", read_clock("bn_debug_clock_read_fwd"))
    # (2) training steps, 2.5 s back to back
    tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.5:
        for _ in range(20):
            tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                    near_far=(0.0, 2.0), **flags)
        torch.cuda.synchronize()
        n += 20
    print(f"training step x{n} ({config}, {dtype}): forward+stash kernel", read_clock("bn_debug_clock_read_fwd"))
    print("   backward chain (or weight-gradient kernel with -DBN_CLOCK_STAMP_WGRAD)", read_clock("bn_debug_clock_read_bwd"))


if __name__ == "__main__":
    main()
