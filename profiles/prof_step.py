"""Small driver for rocprofv3: a few fused training steps of the bench workload (no CPU baseline, no event hooks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from brdf_nerf_amd import load_model
from brdf_nerf_amd.trainer import FusedTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
args = bench.make_args(4096, 64, 64, "bf16")
torch.manual_seed(0)
model = load_model(args).to(dev)
tr = FusedTrainer(model, args, lr=args.lr, ds_lambda=args.ds_lambda)
b = bench.synthetic_batch(4096, 1, dev)
for i in range(steps):
    tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"], near_far=(0.0, 2.0))
torch.cuda.synchronize()
print("done")
