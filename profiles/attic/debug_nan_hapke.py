"""Hunt for the first non-finite quantity of a soak configuration: fused steps with every buffer checked after each step.
    python profiles/debug_nan_hapke.py [config] [dtype] [steps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from brdf_nerf_amd import load_model  # noqa: E402
from brdf_nerf_amd.trainer import FusedTrainer  # noqa: E402


def main():
    sys.argv = [a for a in sys.argv]
    pos = [a for a in sys.argv[1:] if not a.startswith('--')]
    sys_pos = pos
    config = pos[0] if len(pos) > 0 else "hapke"
    dtype = pos[1] if len(pos) > 1 else "fp16"
    steps = int(pos[2]) if len(pos) > 2 else 500
    dev = torch.device("cuda", 0)
    args = bench.make_args(4096, 64, 64, dtype, **bench.CONFIG_FLAGS[config][0])
    torch.manual_seed(0)
    model = load_model(args).to(dev)
    tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False)
    tr.use_graph = "--graph" in sys.argv
    tr.keep_grads = "--keep" in sys.argv
    print("use_graph", tr.use_graph, "keep_grads", tr.keep_grads, flush=True)
    flags = bench.CONFIG_FLAGS[config][1]
    batches = [bench.synthetic_batch(4096, s, dev) for s in range(4)]
    fin = lambda t: bool(torch.isfinite(t).all())
    for i in range(steps):
        b = batches[i % 4]
        loss, rgb = tr.step(b["rays"], b["rgbs"], valid_depth=b["valid_depth"], depths=b["depths"], depth_std=b["depth_std"], near_far=(0.0, 2.0), **flags)
        g, p = tr.flat_grad, tr.flat_param
        d_all, out_all = tr._bufs["d_all"], tr._bufs["out_all"]
        ok = fin(g) and fin(p) and fin(loss) and fin(tr.exp_avg) and fin(tr.exp_avg_sq)
        if i % 20 == 0 or not ok:
            print(f"step {i}: loss {float(loss):.5f} |g|max {float(g.abs().max()):.3e} |d_all|max {float(d_all.abs().max()):.3e} "
                  f"|out|max {float(out_all.abs().max()):.3e} dropped {tr.dropped_grad_elems()} finite g {fin(g)} p {fin(p)} d_all {fin(d_all)} out {fin(out_all)} rgb {fin(rgb)}", flush=True)
        if not ok:
            print("   finite: exp_avg", fin(tr.exp_avg), "exp_avg_sq", fin(tr.exp_avg_sq), "state", tr.state[:4].tolist())
            for name, view in tr.named_views.items() if hasattr(tr, "named_views") else []:
                if not fin(view):
                    print(f"   non-finite parameter: {name} {tuple(view.shape)} {int((~torch.isfinite(view)).sum())} elements")
            for name, view in tr.grad_views.items():
                if not fin(view):
                    bad = ~torch.isfinite(view)
                    print(f"   non-finite gradient: {name} {tuple(view.shape)} {int(bad.sum())} elements (nan {int(torch.isnan(view).sum())})")
            for k in ("m_acc", "m_depth", "m_wsum", "m_var", "s_d_acc", "s_d_wsum", "s_d_depth", "s_rgb"):
                if k in tr._bufs:
                    t = tr._bufs[k]
                    print(f"   {k}: finite {fin(t)} max {float(torch.nan_to_num(t, 0.0, 0.0, 0.0).abs().max()):.3e}")
            top = d_all.abs().amax(-1).flatten().topk(5)
            print("   largest seed rows", top.values.tolist(), top.indices.tolist())
            break
    print("done", i)


if __name__ == "__main__":
    main()
