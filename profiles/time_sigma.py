"""Times the sigma-only fused forward (inference kernel) on the bench shape: ms per launch and executed TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from brdf_nerf_amd import load_model
from brdf_nerf_amd import functions as Fn
dev = torch.device("cuda", 0)
args = bench.make_args(4096, 64, 64, "bf16")
torch.manual_seed(0)
model = load_model(args).to(dev)
spec = model.spec(False, False, False)
packed = model.repack(spec)
b = bench.synthetic_batch(4096, 1, dev)
z = torch.sort(torch.rand(4096, 128, device=dev) * 2, -1)[0]
run = lambda: Fn.field_sigma(spec, model.named(), packed, rays=b["rays"], z=z)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
F, P = 512, 64
flops = 2.0 * z.numel() * (P * F + 6 * F * F + (P + F) * F + F)
print(f"sigma_only ms {ms:.4f} TFLOPs {flops / ms / 1e9:.1f}")
