"""WHERE do the 16-bit modes of BASELINE config 5's models leave the fp32 mode?

`tests/test_gpu_parity.py::test_full_size_render_and_train_step_properties` reports the whole-gradient cosine of one training step
against the fp32 HIP mode at a trained state: microfacet fp16 0.077 where the referee (fp32 arithmetic on fp16-rounded weights)
keeps 0.998, while the 16-bit field backward STARTED FROM THE fp32 ROWS reproduces the fp32 gradient (0.9998).  So the divergence
is upstream of the field backward.  This script runs that step at the same state in fp32 and in the 16-bit mode (same points:
the guided depths of the fp32 run are handed to the 16-bit run) and compares, stage by stage,

  per-sample field outputs (albedo, sigma, analytic normal, BRDF parameters)  ->  composited sums  ->  d loss / d sums
  ->  the gradient rows d loss / d (per-sample outputs) the field backward starts from,

then swaps ONE channel group of the fp32 run's per-sample outputs for the 16-bit run's and reports what that alone does to the
gradient rows: the quantity that carries the divergence.

    python profiles/diag_c5_rows.py [--name=c5_microfacet_fp16] [--pre=150] [--d8lib=brdf_nerf_amd/build/BN_DIAG_D8_IN_F32/libbrdfnerf_hip.so]

--d8lib: a library built with -DBN_DIAG_D8_IN_F32 (the fp32 mode with its activation derivatives sent through the 8-bit codec of the
16-bit modes): one more run, fp32 arithmetic + 8-bit D, which separates the 8-bit derivative stash from the 16-bit activations.
"""
import os
import sys

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, here)
sys.path.insert(0, os.path.join(here, "tests"))
import torch  # noqa: E402


def cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float(torch.nn.functional.cosine_similarity(a, b, dim=0))


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def main():
    opt = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    name, n_pre = opt.get("name", "c5_microfacet_fp16"), int(opt.get("pre", 150))
    import brdf_nerf_amd
    import test_gpu_parity as T
    from oracle.config import FieldConfig
    from brdf_nerf_amd import load_model
    from brdf_nerf_amd.trainer import FusedTrainer
    DEV = T.DEV
    kw, R, S, G, flags, dtype = T.FULL_SIZE[name]
    cfg = FieldConfig(n_samples=S, guided_samples=G, **kw)
    brdf_nerf_amd.set_deterministic(True)
    tb = T._learnable_table(R, 21)
    lb = {k: tb.data[k] for k in ("rays", "rgbs", "valid_depth", "depths", "depth_std")}
    step_kw = dict(valid_depth=lb["valid_depth"], depths=lb["depths"], depth_std=lb["depth_std"], near_far=(0.0, 2.0), **flags)
    # the trained state of the test: n_pre Lambertian-stage steps in the 16-bit mode
    args16 = T.make_args(cfg, dtype)
    torch.manual_seed(0)
    m0 = load_model(args16).to(DEV)
    torch.manual_seed(5)
    t0 = FusedTrainer(m0, args16, lr=5e-4, ds_lambda=10.0, strict_rng=False)
    pre_kw = dict(step_kw, apply_brdf=False, apply_theta=False, cos_irra_on=False)
    for _ in range(n_pre):
        t0.step(lb["rays"], lb["rgbs"], **pre_kw)
    start = {k: v.detach().clone() for k, v in m0.state_dict().items()}
    del t0, m0
    t16 = torch.bfloat16 if dtype == "bf16" else torch.float16

    def run(dt, hooks, round_w=False):
        torch.manual_seed(0)
        m = load_model(T.make_args(cfg, dt)).to(DEV)
        m.load_state_dict(start)
        if round_w:
            with torch.no_grad():
                for p_ in m.parameters():
                    p_.copy_(p_.to(t16).float())
        torch.manual_seed(7)
        t = FusedTrainer(m, T.make_args(cfg, dt), lr=5e-4, ds_lambda=10.0, strict_rng=False)
        t.keep_grads = True
        seen = {}

        def hook(k, d):
            if k in hooks:
                hooks[k](d)
            seen[k] = d.clone()
        t.seed_hook = hook
        t.step(lb["rays"], lb["rgbs"], **step_kw)
        bufs = {k[0]: v.clone() for k, v in t._bufs.items() if k[0] in ("m_acc", "m_depth", "m_wsum", "m_var", "s_d_acc", "s_d_depth", "s_d_wsum", "s_rgb")}
        spec = m.spec(True, True, t.nr_lr, t.nr_an, beta=False)
        return dict(seen=seen, bufs=bufs, grad=t.flat_grad.clone(), spec=spec)

    ref = run("fp32", {})
    spec = ref["spec"]
    C = spec.out_channels
    groups = {"albedo": (0, 3), "sigma": (3, 4)}
    if spec.normal_an:
        groups["normal_an"] = (spec.ch_normal_an, spec.ch_normal_an + 3)
    if spec.normal_lr:
        groups["normal_lr"] = (spec.ch_normal_lr, spec.ch_normal_lr + 3)
    for (hname, _, _), (h0, w) in zip(spec.heads[1:], spec.head_cols[1:]):
        groups[hname] = (h0, h0 + w)
    print(f"{name}: {R} rays x ({S} + {G}), C = {C}, channel groups {groups}")
    z2_ref = ref["seen"]["z2"]
    low = run(dtype, {"z2": lambda d: d.copy_(z2_ref)})
    w16 = run("fp32", {"z2": lambda d: d.copy_(z2_ref)}, round_w=True)
    print(f"whole flat gradient cosine vs fp32: {dtype} {cos(low['grad'], ref['grad']):.5f}, fp32 on {dtype}-rounded weights (referee) {cos(w16['grad'], ref['grad']):.5f}")
    others = [(dtype, low), ("referee", w16)]
    if "d8lib" in opt:
        from brdf_nerf_amd import _lib as L
        hv = L.load(os.path.abspath(opt["d8lib"]))
        assert b"BN_DIAG_D8_IN_F32" in hv.bn_build_flags(), hv.bn_build_flags()
        hv.bn_set_deterministic(1)
        h0 = L.use(hv)
        d8 = run("fp32", {"z2": lambda d: d.copy_(z2_ref)})
        L.use(h0)
        print(f"whole flat gradient cosine vs fp32: fp32 arithmetic with the 8-bit D codec {cos(d8['grad'], ref['grad']):.5f}")
        others.append(("fp32 + 8-bit D", d8))
    for tag, other in others:
        print(f"--- {tag} vs fp32, same points")
        oa, ob = other["seen"]["out_all"].view(-1, C), ref["seen"]["out_all"].view(-1, C)
        for g, (a, b) in groups.items():
            d = (oa[:, a:b] - ob[:, a:b]).abs()
            print(f"  per-sample {g:>10}: rel L2 {rel(oa[:, a:b], ob[:, a:b]):.3e}, max |diff| {float(d.max()):.3e}, rms ref {float(ob[:, a:b].double().pow(2).mean().sqrt()):.3e}")
        if spec.normal_an:
            a = spec.ch_normal_an
            na, nb = oa[:, a:a + 3].double(), ob[:, a:a + 3].double()
            ang = torch.rad2deg(torch.acos((torch.nn.functional.normalize(na, dim=-1) * torch.nn.functional.normalize(nb, dim=-1)).sum(-1).clamp(-1, 1)))
            sig = ob[:, 3].double()
            wgt = sig / sig.sum()
            print(f"  analytic-normal angle to fp32: median {float(ang.median()):.3f} deg, p90 {float(ang.quantile(0.9)):.3f}, p99 {float(ang.quantile(0.99)):.3f}, "
                  f"sigma-weighted mean {float((ang * wgt).sum()):.3f} deg")
        for k in ("m_acc", "m_depth", "m_wsum", "s_rgb", "s_d_acc", "s_d_depth", "s_d_wsum"):
            if k in other["bufs"] and k in ref["bufs"]:
                x, y = other["bufs"][k], ref["bufs"][k]
                if x.dim() == 2 and x.shape[1] == C:
                    per = "  ".join(f"{g} {cos(x[:, a:b], y[:, a:b]):.4f}" for g, (a, b) in groups.items() if float(y[:, a:b].abs().max()) > 0)
                    print(f"  ray-level {k:>9}: cosine {cos(x, y):.5f}  rel L2 {rel(x, y):.3e}   per group: {per}")
                else:
                    print(f"  ray-level {k:>9}: cosine {cos(x, y):.5f}  rel L2 {rel(x, y):.3e}")
        da, db = other["seen"]["d_all"].view(-1, C), ref["seen"]["d_all"].view(-1, C)
        per = "  ".join(f"{g} {cos(da[:, a:b], db[:, a:b]):.4f} (|g|^2 share {float(db[:, a:b].double().pow(2).sum() / db.double().pow(2).sum()):.2e})" for g, (a, b) in groups.items())
        print(f"  gradient rows d_all: cosine {cos(da, db):.5f}; per group: {per}")
    # ---- swap ONE channel group of the fp32 run for the 16-bit run's values: what does that alone do to the rows and the gradient?
    print(f"--- fp32 run with ONE channel group of its per-sample outputs replaced by the {dtype} run's")
    o16 = low["seen"]["out_all"].view(-1, C)
    for g, (a, b) in groups.items():
        def swap(d, a=a, b=b):
            d.view(-1, C)[:, a:b] = o16[:, a:b]
        r = run("fp32", {"z2": lambda d: d.copy_(z2_ref), "out_all": swap})
        print(f"  {g:>10} swapped: gradient rows cosine {cos(r['seen']['d_all'], ref['seen']['d_all']):.5f}, whole flat gradient cosine {cos(r['grad'], ref['grad']):.5f}")


if __name__ == "__main__":
    main()
