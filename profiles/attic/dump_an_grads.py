"""Diagnostic: parameter gradients of one bf16 forward+backward through the analytic-normal path at the bench shape, dumped to
gpurun_out/an_grads_<tag>.pt (compare two library builds: BRDFNERF_HIP_LIB=... python profiles/dump_an_grads.py <tag>)."""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd import load_model  # noqa: E402

tag = sys.argv[1]
dev = torch.device("cuda", 0)
args = bench.make_args(4096, 64, 64, "bf16", funcM=1, funcF=1, funcH=1, normal="analystic")
torch.manual_seed(0)
model = load_model(args).to(dev)
spec = model.spec(True, False, False, True)
packed = model.repack(spec)
g = torch.Generator().manual_seed(1)
xyz = (torch.rand(4096 * 16, 3, generator=g) * 2 - 1).to(dev)
out = model.evaluate(spec, packed, xyz=xyz)
w = torch.randn(out.shape, generator=g).to(dev)
(out * w).sum().backward()
os.makedirs("gpurun_out", exist_ok=True)
torch.save({k: v.grad.cpu() for k, v in model.named_parameters() if v.grad is not None}, f"gpurun_out/an_grads_{tag}.pt")
print("saved", tag)
