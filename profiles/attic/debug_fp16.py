"""Debug aid: one fused step of a full-size config in fp16, reporting where non-finite gradients appear."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
from oracle.config import FieldConfig
from test_gpu_parity import FULL_SIZE, make_args
from brdf_nerf_amd import load_model
from brdf_nerf_amd.trainer import FusedTrainer

name = sys.argv[1] if len(sys.argv) > 1 else "c5_hapke_fp16"
dtype_over = sys.argv[2] if len(sys.argv) > 2 else None
DEV = "cuda:0"
kw, R, S, G, flags, dtype = FULL_SIZE[name]
dtype = dtype_over or dtype
cfg = FieldConfig(n_samples=S, guided_samples=G, **kw)
args = make_args(cfg, dtype)
torch.manual_seed(0)
model = load_model(args).to(DEV)
b = bench.synthetic_batch(R, 5, torch.device(DEV))
tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
for i in range(3):
    loss, _ = tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"], near_far=(0.0, 2.0), **flags)
    print(f"step {i} loss {float(loss):.5f} dropped {tr.dropped_grad_elems()}")
    for key in ("stash1", "stash2"):
        st = tr._bufs.get(key)
        if st is not None:
            print("  ", key, "amax (chain, adjoint):", st[:8].view(torch.float32).tolist())
    d = tr._bufs.get("d_out")
    if d is not None:
        fin = torch.isfinite(d)
        print("   d_out finite:", bool(fin.all()), "max |d_out| per channel:", [f"{float(x):.3e}" for x in torch.where(fin, d, torch.zeros_like(d)).abs().amax((0, 1))])
    bad = {k: int((~torch.isfinite(v)).sum()) for k, v in tr.grad_views.items()}
    print("   non-finite gradient elements:", {k: v for k, v in bad.items() if v} or "none",
          " max |grad|:", f"{float(torch.nan_to_num(tr.flat_grad, 0, 0, 0).abs().max()):.3e}")
