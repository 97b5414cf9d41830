"""Throughput of the two-line drop-in: the reference's own training step - render_rays(mode="train") +
SNerfLoss + DepthLoss + loss.backward() + torch.optim.Adam - with `load_model` / `render_rays` imported from brdf_nerf_amd, beside
FusedTrainer.step on the same batch.  Rays per second per (config, dtype)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd import load_model, render_rays, losses  # noqa: E402
from brdf_nerf_amd.trainer import FusedTrainer  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    R = 4096
    b = bench.synthetic_batch(R, 1, dev)
    for config in ("lambert", "rpv_nan"):
        for dtype in ("bf16", "fp32"):
            args = bench.make_args(R, 64, 64, dtype, **bench.CONFIG_FLAGS[config][0])
            flags = bench.CONFIG_FLAGS[config][1]
            torch.manual_seed(0)
            model = load_model(args).to(dev)
            opt = torch.optim.Adam(model.parameters(), lr=5e-4)

            def step():
                res, _ = render_rays({"coarse": model}, args, b["rays"], None, mode="train", valid_depth=b["valid_depth"],
                                     target_depths=b["depths"], target_std=b["depth_std"], **flags)
                loss = losses.snerf_loss(res["rgb_coarse"], b["rgbs"])
                loss = loss + losses.depth_loss(res["z_vals_coarse"], res["depth_coarse"], res["weights_coarse"], b["depths"][:, 0],
                                                b["depths"][:, 1], b["valid_depth"], b["depth_std"], 10.0)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                return loss
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            n = 30 if dtype == "bf16" else 8
            t0 = time.time()
            for _ in range(n):
                loss = step()
            torch.cuda.synchronize()
            dt = (time.time() - t0) / n
            torch.manual_seed(0)
            tr = FusedTrainer(load_model(args).to(dev), args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
            for _ in range(8):
                tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"], **flags)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(n):
                tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"], **flags)
            torch.cuda.synchronize()
            dtf = (time.time() - t0) / n
            print(f"{config} {dtype}: drop-in render_rays + torch losses + autograd + torch.optim.Adam {R / dt / 1e3:.1f} k rays/s ({dt * 1e3:.2f} ms/step, "
                  f"loss {float(loss):.4f}); FusedTrainer.step {R / dtf / 1e3:.1f} k rays/s ({dtf * 1e3:.2f} ms/step)", flush=True)
            del tr, model, opt
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
