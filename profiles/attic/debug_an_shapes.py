"""Which (F, L) break the analytic-normal double backward in fp32 mode?  Per-parameter gradient errors against the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import test_gpu_parity as T  # noqa: E402
from conftest import tparams  # noqa: E402
from oracle.config import FieldConfig  # noqa: E402
from oracle import field as OF  # noqa: E402

for feat, layers, pe in ((192, 4, True), (192, 8, True), (128, 4, True), (192, 4, False)):
    cfg = FieldConfig(feat=feat, layers=layers, mapping=pe, normal="analystic", n_samples=16, guided_samples=16)
    flags = dict(nr_an_on=True)
    model = T.build_model(cfg, 11)
    p = tparams(cfg, 11)
    for v in p.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(3)
    B = 160
    xyz = torch.rand(B, 3, generator=g) * 2 - 1
    ref = OF.field_forward(p, cfg, xyz, **flags)
    coef = torch.randn(ref.shape, generator=g)
    (ref * coef).sum().backward()
    out = model(xyz.to("cuda"), **flags)
    (out * coef.to("cuda")).sum().backward()
    errs = []
    for k, v in model.named_parameters():
        want = p[k].grad
        if want is None:
            continue
        errs.append((float((v.grad.cpu() - want).abs().max()) / max(float(want.abs().max()), 1e-20), k))
    worst = sorted(errs, reverse=True)[:3]
    print(f"F={feat} L={layers} pe={int(pe)}: out err {float((out.detach().cpu() - ref.detach()).abs().max()):.2e}; worst rel grad errs: "
          + ", ".join(f"{k} {e:.2e}" for e, k in worst), flush=True)
