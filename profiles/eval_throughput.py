"""Evaluation (serving) throughput: render_image over a 512 x 512 image worth of rays (262,144), rgb + depth only, chunked no-grad
render_rays in test mode - the path eval.py / create_dsm.py drive.  Rays per second per (config, dtype)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd import load_model  # noqa: E402
from brdf_nerf_amd.evaluate import render_image  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    N = 512 * 512
    b = bench.synthetic_batch(N, 3, dev)
    for config in ("lambert", "rpv_nan", "hapke"):
        for dtype in ("fp32", "bf16", "fp16"):
            args = bench.make_args(4096, 64, 64, dtype, **bench.CONFIG_FLAGS[config][0])
            torch.manual_seed(0)
            model = load_model(args).to(dev)
            flags = {k: v for k, v in bench.CONFIG_FLAGS[config][1].items()}
            for chunk in (16384,):
                with torch.no_grad():
                    render_image({"coarse": model}, args, b["rays"][:chunk], None, keys=("rgb", "depth"), chunk=chunk, **flags)
                    torch.cuda.synchronize()
                    t0 = time.time()
                    out = render_image({"coarse": model}, args, b["rays"], None, keys=("rgb", "depth"), chunk=chunk, **flags)
                    torch.cuda.synchronize()
                    dt = time.time() - t0
                print(f"{config} {dtype} chunk {chunk}: {N / dt / 1e3:.0f} k rays/s ({dt * 1e3:.1f} ms per 512 x 512 image), rgb finite {bool(torch.isfinite(out['rgb']).all())}", flush=True)


if __name__ == "__main__":
    main()
