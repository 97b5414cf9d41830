"""rocprofv3 driver: sigma-only fused forward (inference kernel) on the bench shape, a few launches."""
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
for i in range(30):
    Fn.field_sigma(spec, model.named(), packed, rays=b["rays"], z=z)
torch.cuda.synchronize()
print("done")
