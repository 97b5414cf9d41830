"""On-device ray table and batch sampler (replaces DataLoader(shuffle=True, num_workers=4), main.py:170-184).

The training set of the reference is one concatenated table of rows `[o3, d3, near, far, sun3]` with rgb (N,3) and,
for the depth-supervised variants, depths (N,2), valid_depth (N), depth_std (N) (satellite_rgb_dep.py:311-394, 708-716).
Here the table stays in HBM and a batch is a gather by a slice of a per-epoch permutation drawn on the device: no
worker processes, no pinned-memory copies, no host sync.  Under data parallelism every rank draws the same permutation
(same seed) and takes its own contiguous share of each global batch (distributed.shard_bounds).
"""
import math

import torch

from .distributed import shard_bounds

_KEYS = ("rays", "rgbs", "depths", "valid_depth", "depth_std")


class RayTable:
    def __init__(self, rays, rgbs, depths=None, valid_depth=None, depth_std=None, device=None, seed=0):
        dev = torch.device(device) if device is not None else rays.device
        n = rays.shape[0]
        self.data = {"rays": rays.to(dev).float().contiguous(), "rgbs": rgbs.to(dev).float().contiguous()}
        for k, v in (("depths", depths), ("valid_depth", valid_depth), ("depth_std", depth_std)):
            if v is not None:
                assert v.shape[0] == n, k
                self.data[k] = v.to(dev).float().contiguous()
        assert self.data["rgbs"].shape[0] == n
        self.n, self.device = n, dev
        self.gen = torch.Generator(device=dev)
        self.gen.manual_seed(seed)
        self.perm = None
        self.cursor = 0
        self.epoch = 0
        self.last_offset = 0

    def __len__(self):
        return self.n

    def _reshuffle(self):
        self.perm = torch.randperm(self.n, device=self.device, generator=self.gen)
        self.cursor = 0

    def next_batch(self, batch_size, rank=0, world=1, out=None):
        """The next `batch_size` rows of the epoch's permutation (the last batch of an epoch is short, like
        DataLoader(drop_last=False)); rank r of `world` gets rows shard_bounds(len, r, world) of that global batch
        (`last_offset` = index of its first row in the global batch).  out: a dict of staging buffers from staging() - the rows
        are gathered INTO them, so successive batches live at the same addresses (the fused step then replays its HIP graph)."""
        if self.perm is None or self.cursor >= self.n:
            if self.perm is not None:
                self.epoch += 1
            self._reshuffle()
        idx = self.perm[self.cursor:self.cursor + batch_size]
        self.cursor += batch_size
        self.last_offset = 0
        if world > 1:
            lo, hi = shard_bounds(idx.shape[0], rank, world)
            idx = idx[lo:hi]
            self.last_offset = lo
        if out is None:
            return {k: v.index_select(0, idx) for k, v in self.data.items()}
        n = idx.shape[0]
        return {k: torch.index_select(v, 0, idx, out=out[k][:n]) for k, v in self.data.items()}

    def staging(self, rows):
        """Buffers for next_batch(out=...): one per table column, `rows` rows each."""
        return {k: torch.empty((rows,) + tuple(v.shape[1:]), dtype=v.dtype, device=self.device) for k, v in self.data.items()}

    def state_dict(self):
        return {"gen": self.gen.get_state(), "perm": self.perm, "cursor": self.cursor, "epoch": self.epoch}

    def load_state_dict(self, sd):
        self.gen.set_state(sd["gen"].cpu())
        self.perm = None if sd["perm"] is None else sd["perm"].to(self.device)
        self.cursor, self.epoch = int(sd["cursor"]), int(sd["epoch"])


def synthetic_table(n_rays, n_images=3, device="cpu", seed=1, with_depth=True):
    """Djibouti-shaped synthetic table: normalised scene cube, near-nadir views, one sun direction
    per image, near = 0, far constant per image."""
    g = torch.Generator().manual_seed(seed)
    per = (n_rays + n_images - 1) // n_images
    rows = []
    for _ in range(n_images):
        o = torch.cat([torch.rand(per, 2, generator=g) * 2 - 1, 1 + 0.02 * torch.randn(per, 1, generator=g)], -1)
        el = math.radians(60 + 30 * float(torch.rand(1, generator=g)))
        az = 6.283185307179586 * float(torch.rand(1, generator=g))
        d = torch.tensor([math.cos(el) * math.cos(az), math.cos(el) * math.sin(az), -math.sin(el)]).expand(per, 3)
        sel = math.radians(40 + 30 * float(torch.rand(1, generator=g)))
        saz = 6.283185307179586 * float(torch.rand(1, generator=g))
        sun = torch.tensor([math.cos(sel) * math.cos(saz), math.cos(sel) * math.sin(saz), math.sin(sel)]).expand(per, 3)
        far = torch.full((per, 1), 1.8 + 0.4 * float(torch.rand(1, generator=g)))
        rows.append(torch.cat([o, d, torch.zeros(per, 1), far, sun], -1))
    rays = torch.cat(rows, 0)[:n_rays]
    rgbs = torch.rand(n_rays, 3, generator=g)
    kw = {}
    if with_depth:
        kw = dict(depths=torch.stack([0.8 + 0.4 * torch.rand(n_rays, generator=g), torch.rand(n_rays, generator=g)], -1),
                  valid_depth=(torch.rand(n_rays, generator=g) < 0.7).float(), depth_std=torch.zeros(n_rays))
    return RayTable(rays, rgbs, device=device, seed=seed, **kw)

