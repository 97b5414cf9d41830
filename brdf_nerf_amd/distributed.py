"""Ray-batch data parallelism: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  Rays are independent; the only exchange is ONE all-reduce of the flat gradient buffer per
step (reference: Lightning DDP's bucketed all-reduce, main.py:720-731)."""
import torch
import torch.distributed as dist


def world_info(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def shard_bounds(n_rays, rank, world):
    """Contiguous, balanced split of n_rays rows: rank r owns [lo, hi)."""
    base, rem = divmod(n_rays, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_sum_(flat_grad, group=None):
    """In-place SUM all-reduce of the single flat gradient buffer (the DDP mean's 1/world is folded into Adam's
    grad_scale so no extra pass over the buffer is needed)."""
    _, world = world_info(group)
    if world > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


def gather_rows(local, group=None):
    """Eval: all-gather per-ray results (R/W, C) -> (R, C) (rows in rank order)."""
    _, world = world_info(group)
    if world == 1:
        return local
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    mx = int(max(int(s) for s in sizes))
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[:int(s)] for o, s in zip(out, sizes)], 0)
