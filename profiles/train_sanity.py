"""Sanity run of the Lightning-free harness at BASELINE shapes (bf16, F=512): a 3-image synthetic table whose colours are a
smooth function of the ray origin (so there is something to learn), 300 steps, loss / PSNR / lr every 50.
    python profiles/train_sanity.py [lambert | rpv_nan | hapke]
lambert = config 2 (Lambertian throughout); rpv_nan = config 3 (Lambertian pretrain, RPV funcM/F/H + analytic normals from 40 %
of the steps, normal regulariser from 60 %); hapke = config 5's Hapke half (b, c, theta heads, learned normals, BRDF stage from 40 %)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd.raytable import synthetic_table  # noqa: E402
from brdf_nerf_amd.train import TrainLoop  # noqa: E402

dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "lambert"
extra = {"lambert": dict(brdf_on=1.0, cos_irra_on=1.0, nrrg_on=0.0),
         "rpv_nan": dict(brdf_on=0.4, cos_irra_on=0.4, nrrg_on=0.6, funcM=1, funcF=1, funcH=1, normal="analystic"),
         "hapke": dict(brdf_on=0.4, cos_irra_on=0.4, nrrg_on=0.6, b=1, c=1, theta=1, normal="learned")}[which]
args = bench.make_args(4096, 64, 64, "bf16", max_train_steps=300, ds_drop=0.5, gsam_only_on=1.0, in_ckpts="none", **extra)
print("config:", which)
table = synthetic_table(4096 * 20, device=dev, seed=3)
o = table.data["rays"][:, :3]
table.data["rgbs"] = torch.stack([0.5 + 0.4 * torch.sin(3 * o[:, 0]), 0.5 + 0.4 * torch.cos(2 * o[:, 1]),
                                  0.5 + 0.3 * torch.sin(2 * o[:, 0] + o[:, 1])], -1).contiguous()
torch.manual_seed(0)
loop = TrainLoop(args, table, near_far=(0.0, 2.2))
t0 = time.perf_counter()
for i in range(300):
    out = loop.step()
    if (i + 1) % 50 == 0:
        torch.cuda.synchronize()
        print(f"step {i + 1:4d} epoch {out['epoch']} brdf {out.get('apply_brdf', '-')} loss {float(out['loss']):.5f} psnr {float(out['psnr']):.2f} dB "
              f"lr {out['lr']:.2e} depth_loss_on {out['depth_loss_on']}  {(time.perf_counter() - t0) / (i + 1) * 1e3:.2f} ms/step",
              flush=True)
finite = all(bool(torch.isfinite(p).all()) for p in loop.model.parameters())
print("parameters finite:", finite)
