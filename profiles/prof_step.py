"""Small driver for rocprofv3: a few fused training steps of a bench workload (no CPU baseline, no event hooks).
    python3 profiles/prof_step.py [steps] [config] [dtype]        (defaults: 5 lambert bf16; configs: bench.CONFIG_FLAGS)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from brdf_nerf_amd import load_model
from brdf_nerf_amd.trainer import FusedTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
config = sys.argv[2] if len(sys.argv) > 2 else "lambert"
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
dev = torch.device("cuda", 0)
over, flags, _ = bench.CONFIG_FLAGS[config]
args = bench.make_args(4096, 64, 64, dtype, **over)
torch.manual_seed(0)
model = load_model(args).to(dev)
tr = FusedTrainer(model, args, lr=args.lr, ds_lambda=args.ds_lambda, strict_rng=False)     # the launch-lean step, as bench.py runs it
tr.use_graph = False      # eager launches: one profiler record per kernel launch
b = bench.synthetic_batch(4096, 1, dev)
for i in range(steps):
    tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"], near_far=(0.0, 2.0), **flags)
torch.cuda.synchronize()
print("done")
