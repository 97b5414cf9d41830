"""Worker of test_two_rank_step_matches_one_rank (tests/test_gpu_parity.py): rank r of 2, both on cuda:0, gloo.

Each rank runs ONE FusedTrainer.step on its contiguous half of a global batch; the all-reduced flat gradient / world must
equal the gradient of a purely local trainer stepping the whole batch (what Lightning DDP's gradient averaging gives the
reference, main.py:196,718-731).  The random draws of the step are served from full-batch tensors (seeded on the host),
each rank taking its rows, so both runs see the same numbers.
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    backend = os.environ.get("BN_DIST_BACKEND", "gloo")      # "nccl" = RCCL: world 1 on the one-GPU box (one rank per device)
    dist.init_process_group(backend, rank=rank, world_size=world)
    import bench
    from oracle.config import FieldConfig
    from test_gpu_parity import make_args
    from brdf_nerf_amd import load_model
    from brdf_nerf_amd.distributed import shard_bounds
    from brdf_nerf_amd.trainer import FusedTrainer

    name = os.environ.get("BN_DIST_CONFIG", "lambert")
    kw = dict(funcM=1, funcF=1, funcH=1, normal="analystic") if name == "rpv_nan" else {}
    flags = dict(apply_brdf=True, apply_theta=True, cos_irra_on=True) if name == "rpv_nan" else {}
    R, S, G = 256, 16, 16
    cfg = FieldConfig(feat=64, n_samples=S, guided_samples=G, **kw)
    args = make_args(cfg, "fp32")
    b = bench.synthetic_batch(R, 7, dev)
    g = torch.Generator().manual_seed(11)
    u_t_row = torch.rand(1, G, generator=g)
    draws_full = [torch.rand(R, S, generator=g), torch.rand(R, G, generator=g), u_t_row.expand(R, G).contiguous()]

    lean = os.environ.get("BN_DIST_LEAN", "1") == "1"

    def run(lo, hi, data_parallel):
        torch.manual_seed(0)
        model = load_model(args).to(dev)
        torch.manual_seed(5)               # the step state's draw key: the same on every rank
        tr = FusedTrainer(model, args, lr=5e-4, ds_lambda=10.0, strict_rng=False, data_parallel=data_parallel)
        tr.lean = lean
        tr.keep_grads = True
        tr.ray_offset = lo                 # lean step: in-kernel draws per GLOBAL ray - a shard draws what the whole batch would
        feed = [d[lo:hi].to(dev) for d in draws_full] if not lean else []
        orig = torch.rand

        def rand(*size, **k):
            t = feed.pop(0)
            assert tuple(t.shape) == tuple(size), (tuple(t.shape), size)
            return t

        torch.rand = rand
        try:
            tr.step(b["rays"][lo:hi].contiguous(), b["rgbs"][lo:hi].contiguous(), valid_depth=b["valid_depth"][lo:hi].contiguous(),
                    depths=b["depths"][lo:hi].contiguous(), depth_std=b["depth_std"][lo:hi].contiguous(), near_far=(0.0, 2.0), **flags)
        finally:
            torch.rand = orig
        assert not feed
        return tr

    lo, hi = shard_bounds(R, rank, world)
    tr2 = run(lo, hi, True)
    assert tr2.world == world
    if backend == "nccl":       # the product's collective on the product's buffer, through RCCL (sum over one rank: unchanged)
        before = tr2.flat_grad.clone()      # (allreduce_sum_ skips the call at world 1: issue the same collective directly)
        dist.all_reduce(tr2.flat_grad, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        assert dist.get_backend() == "nccl" and torch.equal(before, tr2.flat_grad)
    got = tr2.flat_grad / world                      # what Adam's grad_scale = 1 / world applied
    tr1 = run(0, R, False)
    assert tr1.world == 1
    want = tr1.flat_grad
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    perr = float((tr2.flat_param - tr1.flat_param).abs().max())
    pmean = float((tr2.flat_param - tr1.flat_param).abs().mean())
    # (the first Adam step moves every parameter by lr g / (|g| + eps): where |g| is of the order of eps = 1e-8, a 1e-8 difference
    # of the summed gradient - the order two ranks add their halves in - changes the step by a fraction of lr; everywhere else
    # the parameters agree to rounding: bounded by lr in the maximum, held to 1e-7 in the mean)
    ok = scale > 0 and err <= 2e-5 * scale and perr <= 1.05 * tr1.lr and pmean <= 1e-7
    print(f"RESULT rank {rank} {name}: grad err {err:.3e} of {scale:.3e}, param err after Adam max {perr:.3e} mean {pmean:.3e} -> "
          f"{'ok' if ok else 'FAIL'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
