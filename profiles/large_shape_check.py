"""BASELINE config 4's GLOBAL batch on one GPU (8192 rays x 128 + 64 samples = 1.57 M points per step, RPV + learned
normals, bf16): two fused steps, finite loss/gradients, peak memory."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd import load_model  # noqa: E402
from brdf_nerf_amd.trainer import FusedTrainer  # noqa: E402

dev = torch.device("cuda", 0)
args = bench.make_args(8192, 128, 64, "bf16", funcM=1, funcF=1, funcH=1, normal="learned")
torch.manual_seed(0)
model = load_model(args).to(dev)
tr = FusedTrainer(model, args, lr=args.lr, ds_lambda=10.0, strict_rng=False)
b = bench.synthetic_batch(8192, 3, dev)
for i in range(3):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    loss, _ = tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"],
                      near_far=(0.0, 2.0), apply_brdf=True, apply_theta=True, cos_irra_on=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"step {i}: loss {float(loss):.5f} finite grads {bool(torch.isfinite(tr.flat_grad).all())} "
          f"{e0.elapsed_time(e1):.1f} ms  peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
