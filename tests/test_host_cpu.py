"""CPU-side tests: C-ABI library loads and exports every declared symbol, host-side layout logic, the drop-in module's
state_dict contract, product-side losses vs the golden loss fixture, and the data-parallel gradient sync (gloo, world 2)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden, assert_close, ROOT
from oracle.config import FieldConfig

CONFIGS = {
    "lambert": dict(),
    "rpv111_nlr": dict(funcM=1, funcF=1, funcH=1, normal="learned"),
    "rpv333_nlr": dict(funcM=1, funcF=1, funcH=1, dim_RPV=3, normal="learned"),
    "hapke_bct": dict(b=1, c=1, theta=1, normal="learned"),
    "microfacet": dict(roughness=True, normal="learned"),
    "rpv111_nan": dict(funcM=1, funcF=1, funcH=1, normal="analystic"),
    "hapke_bc_both": dict(b=1, c=1, normal="analystic_learned"),
}


def make_args(cfg, **over):
    import argparse
    a = argparse.Namespace(
        model="spsbrdf-nerf", fc_layers=cfg.layers, fc_feat=cfg.feat, mapping=cfg.mapping, siren=int(cfg.siren),
        t_embbeding_tau=cfg.t_dim, beta=bool(cfg.beta), roughness=cfg.roughness, normal=cfg.normal, indirect_light=False, glossy_scale=1.0,
        sun_v="none", MultiBRDF=0, dim_RPV=cfg.dim_RPV, input_viewdir=int(cfg.input_viewdir), funcM=cfg.funcM, funcF=cfg.funcF, funcH=cfg.funcH,
        b=cfg.b, c=cfg.c, theta=cfg.theta, shell_hapke=0, hpk_scl=4.0, guided_samples=64, n_samples=64, n_importance=0,
        std_range=3.0, data="sat", sc_lambda=0.0, chunk=5120, noise_std=0.0, margin=1e-4, stdscale=1, fresnel_f0=0.04)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_library_exports_every_declared_symbol():
    from brdf_nerf_amd import _lib
    header = open(os.path.join(ROOT, "include", "brdfnerf_hip.h")).read()
    declared = set(re.findall(r"\b(bn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    L = _lib.lib()                       # dlopen; resolves every symbol or raises
    assert L.bn_abi_version() == _lib.BN_ABI_VERSION == 7
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name


def test_release_library_has_no_diagnostic_switches():
    """bn_build_flags(): the in-tree library is built without any A/B or timing define."""
    from brdf_nerf_amd import _lib
    assert _lib.lib().bn_build_flags() == b""


def test_library_carries_the_hash_of_the_sources_beside_it(tmp_path, monkeypatch):
    """build() decides by content, not mtimes - the library holds the sha256 prefix of the sources it was
    compiled from; the loader refuses a library built from other sources."""
    from brdf_nerf_amd import _lib, build
    want = build.source_hash()
    assert _lib.lib().bn_source_hash().decode() == want
    assert build.library_hash() == want and not build.needs_build()
    # another tree state: the same library is stale, and the loader says so
    monkeypatch.setattr(build, "source_hash", lambda: "0123456789abcdef")
    assert build.needs_build()
    with pytest.raises(_lib.LibraryMissing, match="other sources"):
        _lib.load(_lib.LIB_PATH)
    monkeypatch.setenv("BRDFNERF_ALLOW_STALE_LIB", "1")
    _lib.load(_lib.LIB_PATH)


def test_device_test_programs_compile_for_gfx950(tmp_path):
    """tests/d8_roundtrip.hip (run by the GPU suite against csrc/field_kernels.h's 8-bit derivative stash) cross-compiles here:
    a change of the header that breaks the program shows up in the CPU suite, not first on the GPU box."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-value", "-Wno-unused-result", "-Wno-pass-failed",
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "brdf_nerf_amd", "csrc"),
                    os.path.join(ROOT, "tests", "d8_roundtrip.hip"), "-o", str(tmp_path / "d8_roundtrip")], check=True, timeout=600)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from brdf_nerf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.LibraryMissing):
        _lib.lib()


def test_product_path_does_not_import_oracle():
    code = "import sys; import brdf_nerf_amd, brdf_nerf_amd.trainer; print(any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules))"
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, text=True).strip()
    assert out == "False"


@pytest.mark.parametrize("name", list(CONFIGS))
def test_module_contract_and_channel_layout(name):
    from brdf_nerf_amd import load_model
    cfg = FieldConfig(feat=64, **CONFIGS[name])
    model = load_model(make_args(cfg))
    want = {k: tuple(s) for k, s, _ in cfg.param_shapes()}
    got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert got == want                                   # reference state_dict keys and shapes
    assert list(got) == list(want)                       # and registration order
    for apply_brdf in (False, True):
        spec = model.spec(apply_brdf, True, cfg.normal in ("learned", "analystic_learned"),
                          cfg.normal in ("analystic", "analystic_learned"))
        assert spec.out_channels == cfg.out_channels(apply_brdf, True)
        assert spec.packed_bytes > 0
    assert model.number_of_outputs == 4
    assert model.RPV == cfg.RPV


def test_init_ranges_follow_reference():
    from brdf_nerf_amd import load_model
    torch.manual_seed(0)
    model = load_model(make_args(FieldConfig(feat=128)))
    w0 = model.fc_net[0].weight
    assert float(w0.abs().max()) <= 1 / 60 + 1e-7                       # first_layer_sine_init
    w1 = model.fc_net[2].weight
    assert float(w1.abs().max()) <= np.sqrt(6 / 128) + 1e-6 and float(w1.abs().max()) > 0.9 * np.sqrt(6 / 128)
    w4 = model.fc_net[8].weight
    assert w4.shape == (128, 188)
    assert float(model.sigma_from_xyz[0].weight.abs().max()) <= 1 / np.sqrt(128) + 1e-6   # nn.Linear default


def test_init_is_rng_order_identical_to_reference():
    """tests/golden/init_state_dicts.npz: state_dict of the REFERENCE's load_model(args) under torch.manual_seed(7) (rows
    a4 / a21 of SURVEY section 8).  The drop-in module built under the same seed must hold the same bits: same keys in
    the same order, every tensor of the RPV + both-normals model equal, and the fingerprints (sum, sum |.|, size, first /
    last 8 elements) of every tensor of the other head sets."""
    from brdf_nerf_amd import load_model
    g = load_golden("init_state_dicts")
    seed = int(g["seed"])
    kinds = {"lambert": dict(), "rpv111_anlr": dict(funcM=1, funcF=1, funcH=1, normal="analystic_learned"),
             "hapke_bct_nlr": dict(b=1, c=1, theta=1, normal="learned"), "microfacet_nlr": dict(roughness=True, normal="learned"),
             "relu_rpvM": dict(siren=False, funcM=1), "nomap": dict(mapping=False),
             "viewdir": dict(input_viewdir=1), "beta_nlr": dict(beta=True, normal="learned", funcM=1)}
    for name, kw in kinds.items():
        cfg = FieldConfig(feat=64, n_samples=16, guided_samples=16, **kw)
        torch.manual_seed(seed)
        sd = load_model(make_args(cfg)).state_dict()
        assert list(sd) == str(g[f"{name}/keys"]).split("\n"), name
        for k, v in sd.items():
            f = v.detach().double().flatten()
            fp = np.concatenate([[float(f.sum()), float(f.abs().sum()), float(f.numel())], f[:8].numpy(), f[-8:].numpy()])
            assert np.array_equal(fp, g[f"{name}/fp/{k}"]), (name, k)
            if name == "rpv111_anlr":
                assert np.array_equal(v.numpy(), g[f"{name}/full/{k}"]), (name, k)


def test_unsupported_flags_raise():
    from brdf_nerf_amd import load_model
    for over in (dict(sun_v="learned"), dict(indirect_light=True)):
        with pytest.raises(NotImplementedError):
            load_model(make_args(FieldConfig(feat=64), **over))
    with pytest.raises(ValueError):
        load_model(make_args(FieldConfig(feat=64), model="sat-nerf"))
    assert load_model(make_args(FieldConfig(feat=64), sun_v="analystic")).sun_v == "analystic"   # no parameters: sun pass in render_rays


def test_packed_and_stash_sizes():
    from brdf_nerf_amd import load_model, _lib
    from brdf_nerf_amd import functions as Fn
    model = load_model(make_args(FieldConfig()))
    spec = model.spec(False, False, False)
    F, P = 512, 64
    fold = spec.fold_feats                                    # feats layer folded into the heads: no F x F feats blocks
    fwd = F * P + 6 * F * F + (F * P + F * F) + (0 if fold else F * F) + 256 * F + 32 * F     # + the sigma head's 32-row tile
    bwd = 7 * F * F + (0 if fold else F * F) + F * 256 + 2 * P * F   # + (W_0[:, :P])^T and (W_skip[:, :P])^T for the normals adjoint
    assert spec.packed_bytes == (fwd + bwd) * 4                      # fp32 parity mode
    n = Fn.field_stash_bytes(spec, 1000)
    assert n > 1024 * F * 4 * 16 and n % 256 == 0
    model.compute_dtype = "bf16"
    assert model.spec(False, False, False).packed_bytes == (fwd + bwd) * 2
    vd = load_model(make_args(FieldConfig(input_viewdir=1), compute_dtype="bf16"))         # + [256 rows][32] direction columns
    assert vd.spec(False, False, False).packed_bytes == (fwd + bwd + 256 * 32) * 2


def test_losses_match_golden():
    from brdf_nerf_amd import losses
    g = load_golden("loss_snerf_depth")
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    rgb, depth, w = [t[k].clone().requires_grad_(True) for k in ("rgb", "depth", "weights")]
    l_rgb = losses.snerf_loss(rgb, t["tgt"])
    l_ds = losses.depth_loss(t["z"], depth, w, t["target_depths"][:, 0], t["target_depths"][:, 1], t["valid_depth"],
                             t["target_std"], 10.0)
    assert_close(l_rgb, g["loss_rgb"], 1e-6, 1e-8, "loss_rgb")
    assert_close(l_ds, g["loss_ds"], 1e-6, 1e-8, "loss_ds")
    (l_rgb + l_ds).backward()
    assert_close(rgb.grad, g["drgb"], 1e-6, 1e-9, "drgb")
    assert_close(depth.grad, g["ddepth"], 1e-6, 1e-9, "ddepth")
    assert_close(losses.psnr(t["rgb"], t["tgt"]), g["psnr"], 1e-6, 1e-6, "psnr")


def test_regulariser_losses_match_golden():
    """NormalRegLoss / HardSurfaceLoss / NormalLoss in their mask (sync-free) form against the reference's values and
    gradients (tests/golden/loss_regularisers.npz)."""
    from brdf_nerf_amd import losses
    g = load_golden("loss_regularisers")
    t = {k: torch.from_numpy(v) for k, v in g.items()}
    w, depth, n_an, n_lr = [t[k].clone().requires_grad_(True) for k in ("weights", "depth", "normal_an", "normal_lr")]
    l_an, perc_an = losses.normal_reg_loss(n_an, w, t["view"], 0.1)
    l_lr, perc_lr = losses.normal_reg_loss(n_lr, w, t["view"], 0.05)
    l_hs = losses.hard_surface_loss(t["z"], depth, w, 0.5)
    l_n1 = losses.normal_loss(w, n_an, n_lr, 0.01, "an_lr")
    l_n3 = losses.normal_loss(w, t["normal_gt"], n_an, 0.01, "an", target_weight=t["target_weight"], valid_depth=t["valid_depth"])
    for name, got in (("l_nr_an", l_an), ("l_nr_lr", l_lr), ("l_hs", l_hs), ("l_n1", l_n1), ("l_n3", l_n3)):
        assert_close(got, g[name], 1e-5, 1e-9, name)
    assert abs(float(perc_an) - float(g["perc_an"])) < 1e-3 and abs(float(perc_lr) - float(g["perc_lr"])) < 1e-3
    (l_an + l_lr + l_hs + l_n1 + l_n3).backward()
    assert_close(w.grad, g["d_weights"], 1e-5, 1e-9, "d_weights")
    assert_close(depth.grad, g["d_depth"], 1e-5, 1e-9, "d_depth")
    assert_close(n_an.grad, g["d_normal_an"], 1e-5, 1e-10, "d_normal_an")
    assert_close(n_lr.grad, g["d_normal_lr"], 1e-5, 1e-10, "d_normal_lr")


def test_stage_schedule_follows_reference_thresholds():
    """main.py:56-76,194-246: thresholds = round(fraction * max_train_steps), `>` switches a stage on, train_steps advance
    by the GPU count, epoch = train_steps // (n_rays // batch_size), StepLR(0.9) per loader pass, noise_std *= 0.9 per step."""
    import argparse
    from brdf_nerf_amd.schedule import StageSchedule
    a = argparse.Namespace(max_train_steps=100, brdf_on=0.2, nrrg_on=0.1, gsam_only_on=1.0, cos_irra_on=0.25, ds_drop=0.5,
                           ds_lambda=10.0, batch_size=8, lr=5e-4, noise_std=1.0)
    s = StageSchedule(a, n_train_rays=83, world=1)          # 10 steps per epoch
    assert (s.brdf_on, s.nrrg_on, s.cos_irra_on, s.ds_drop, s.steps_per_epoch, s.max_steps) == (20, 10, 25, 50, 10, 100)
    seen = {}
    for i in range(1, 61):
        f = s.begin_step()
        s.end_step()
        seen[i] = f
    assert not seen[20]["apply_brdf"] and seen[21]["apply_brdf"]
    assert not seen[40]["apply_theta"] and seen[41]["apply_theta"]
    assert not seen[25]["cos_irra_on"] and seen[26]["cos_irra_on"]
    assert seen[49]["depth_loss_on"] and not seen[50]["depth_loss_on"]
    assert not seen[10]["nr_reg_on"] and seen[11]["nr_reg_on"]
    assert seen[29]["epoch"] == 2 and not seen[29]["hs_on"] and seen[30]["epoch"] == 3 and seen[30]["hs_on"]
    assert not any(f["gsam_only"] for f in seen.values())
    assert abs(a.noise_std - 0.9 ** 60) < 1e-12
    # StepLR(0.9) ticks per pass over the rank's loader: len(DataLoader(83 rays, batch 8, drop_last=False)) = 11 steps
    assert s.lr_steps_per_epoch == 11
    assert abs(s.lr(0) - 5e-4) < 1e-15 and abs(s.lr(10) - 5e-4) < 1e-15 and abs(s.lr(11) - 4.5e-4) < 1e-12
    assert abs(s.lr(25) - 5e-4 * 0.81) < 1e-12
    s4 = StageSchedule(argparse.Namespace(**dict(vars(a), noise_std=0.0)), 83, world=4)
    f = [s4.begin_step() for _ in range(6)]
    assert s4.train_steps == 24 and s4.max_steps == 25 and not f[4]["apply_brdf"] and f[5]["apply_brdf"]
    # 4 GPUs: DistributedSampler gives each rank ceil(83/4) = 21 rays = 3 batches per epoch -> the rate decays 4x faster
    # per optimiser step than on one GPU (Lightning DDP), and RayTable serves 3 global batches of 32 per permutation
    assert s4.lr_steps_per_epoch == 3 and abs(s4.lr(2) - 5e-4) < 1e-15 and abs(s4.lr(3) - 4.5e-4) < 1e-12
    from brdf_nerf_amd.raytable import synthetic_table
    t = synthetic_table(83, device="cpu", seed=0)
    for _ in range(3):
        t.next_batch(8 * 4, rank=1, world=4)
    assert t.epoch == 0
    t.next_batch(8 * 4, rank=1, world=4)
    assert t.epoch == 1


def test_ray_table_epochs_and_shards():
    from brdf_nerf_amd.raytable import synthetic_table
    t = synthetic_table(50, device="cpu", seed=3)
    assert t.data["rays"].shape == (50, 11) and abs(float(t.data["rays"][:, 3:6].norm(dim=-1).mean()) - 1) < 1e-5
    tag = torch.arange(50.0)
    t.data["rgbs"][:, 0] = tag                                   # tag the rows to follow them through the sampler
    got = [t.next_batch(16) for _ in range(4)]
    assert [b["rays"].shape[0] for b in got] == [16, 16, 16, 2]  # last batch of the epoch is short (drop_last=False)
    ids = torch.cat([b["rgbs"][:, 0] for b in got]).long()
    assert sorted(ids.tolist()) == list(range(50))               # every row exactly once per epoch
    b5 = t.next_batch(16)
    assert t.epoch == 1 and b5["rays"].shape[0] == 16
    # data parallel: same seed on every rank, contiguous shares of each global batch
    ta, tb = synthetic_table(50, device="cpu", seed=3), synthetic_table(50, device="cpu", seed=3)
    for tt in (ta, tb):
        tt.data["rgbs"][:, 0] = tag
    a0, b1 = ta.next_batch(16, 0, 2), tb.next_batch(16, 1, 2)
    assert a0["rays"].shape[0] == 8 and b1["rays"].shape[0] == 8
    assert torch.equal(torch.cat([a0["rgbs"][:, 0], b1["rgbs"][:, 0]]), got[0]["rgbs"][:, 0])
    sd = ta.state_dict()
    nxt = ta.next_batch(16)["rgbs"][:, 0].clone()
    ta.load_state_dict(sd)
    assert torch.equal(ta.next_batch(16)["rgbs"][:, 0], nxt)


def test_shard_bounds_cover_rows():
    from brdf_nerf_amd.distributed import shard_bounds
    for n, w in ((4096, 8), (1000, 3), (7, 8), (0, 2)):
        edges = [shard_bounds(n, r, w) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        assert max(hi - lo for lo, hi in edges) - min(hi - lo for lo, hi in edges) <= 1


_DDP_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BN_ROOT"])
from brdf_nerf_amd.distributed import allreduce_sum_, shard_bounds, gather_rows
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(0)
W = torch.randn(16, 8, requires_grad=True)            # replicated weights
x = torch.randn(64, 8); y = torch.randn(64, 16)       # the global ray batch (same on every rank)
lo, hi = shard_bounds(64, rank, world)
loss = ((x[lo:hi] @ W.t() - y[lo:hi]) ** 2).mean()    # per-rank mean, like the reference under DDP
loss.backward()
flat = W.grad.flatten().clone()
allreduce_sum_(flat)
flat /= world                                         # what Adam's grad_scale = 1/world applies
full = ((x @ W.detach().t() - y) ** 2).mean()
Wf = W.detach().clone().requires_grad_(True)
((x @ Wf.t() - y) ** 2).mean().backward()
err = float((flat - Wf.grad.flatten()).abs().max())
rows = gather_rows(x[lo:hi])
ok = err < 1e-6 and torch.equal(rows, x)
print("RESULT", rank, ok, err)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def test_gradient_sync_world2_gloo(tmp_path):
    script = tmp_path / "ddp_worker.py"
    script.write_text(_DDP_WORKER)
    env = dict(os.environ, BN_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_checkpoint_prefix_round_trip(tmp_path):
    """Lightning-style checkpoint -> module (eval.py:26-54) and the stage-2 partial warm start (main.py:97-104)."""
    from brdf_nerf_amd import load_model
    from brdf_nerf_amd.evaluate import load_ckpt
    cfg1 = FieldConfig(feat=64)
    m1 = load_model(make_args(cfg1))
    ckpt = {"state_dict": {f"nerf_coarse.{k}": v.clone() for k, v in m1.state_dict().items()}, "epoch": 9}
    path = tmp_path / "epoch=9.ckpt"
    torch.save(ckpt, path)
    m2 = load_model(make_args(cfg1))
    load_ckpt(m2, str(path), model_name="nerf_coarse")
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # stage 2: RPV heads are new, the Lambertian sub-modules are warm-started by prefix with drop_len=11
    cfg2 = FieldConfig(feat=64, funcM=1, funcF=1, funcH=1, normal="learned")
    m3 = load_model(make_args(cfg2))
    before = {k: v.clone() for k, v in m3.state_dict().items()}
    for sub in ("fc_net", "sigma_from_xyz", "feats_from_xyz", "rgb_from_xyzdir"):
        load_ckpt(m3, str(path), model_name=f"nerf_coarse.{sub}", drop_len=11)
    for k, v in m3.state_dict().items():
        if k in m1.state_dict():
            assert torch.equal(v, m1.state_dict()[k]), k
        else:
            assert torch.equal(v, before[k]), k


@pytest.mark.parametrize("N,B,W", [(1000, 64, 1), (1000, 64, 2), (1003, 32, 4), (517, 100, 8), (4096, 512, 8), (70, 64, 4)])
def test_ray_table_epoch_semantics_against_dataloader_and_distributed_sampler(N, B, W):
    """SURVEY 8(f) row 3 against the loaders the reference runs (main.py:170-184: DataLoader(shuffle=True, batch_size=B);
    under `Trainer(gpus=W)` Lightning swaps its sampler for DistributedSampler(shuffle=True), one loader per rank).
    Same: optimiser steps per epoch per rank (the unit StepLR ticks in), every batch of B rows except the epoch's last, every
    row of the table once per epoch, a fresh permutation per epoch, `train_steps += W` per step.  Deliberately different (and
    documented in DESIGN.md): DistributedSampler PADS its permutation to a multiple of W by repeating rows (up to W - 1 rows are
    seen twice per epoch) and deals rows to ranks with stride W; the ray table serves each row exactly once and gives rank r a
    contiguous share of every global batch - the last global batch of an epoch is split as evenly as it divides, so ranks may
    differ by one row there (never by more), where the reference's ranks are equal."""
    import argparse
    import torch
    from torch.utils.data import DataLoader, DistributedSampler, TensorDataset
    from brdf_nerf_amd.raytable import RayTable
    from brdf_nerf_amd.schedule import StageSchedule
    ids = torch.arange(N, dtype=torch.float32)
    rays = ids[:, None].expand(N, 11).contiguous()            # row id in every column: a batch reveals which rows it holds
    ds = TensorDataset(ids)
    ref_steps, ref_rows = [], []
    for r in range(W):
        sampler = DistributedSampler(ds, num_replicas=W, rank=r, shuffle=True, seed=0) if W > 1 else None
        dl = DataLoader(ds, batch_size=B, shuffle=(sampler is None), sampler=sampler)
        ref_steps.append(len(dl))
        sizes = [b[0].shape[0] for b in dl]
        assert all(s == B for s in sizes[:-1]) and 0 < sizes[-1] <= B
        ref_rows.append(sum(sizes))
    assert len(set(ref_steps)) == 1 and len(set(ref_rows)) == 1          # equal across ranks upstream (padding)
    steps = ref_steps[0]
    assert ref_rows[0] == -(-N // W)
    args = argparse.Namespace(batch_size=B, max_train_steps=10 * steps * W, lr=5e-4, noise_std=0.0, ds_lambda=0.0)
    sch = StageSchedule(args, N, W)
    assert sch.lr_steps_per_epoch == steps                               # StepLR ticks after the same number of optimiser steps
    assert sch.max_steps == (args.max_train_steps if W == 1 else args.max_train_steps // W)
    tables = [RayTable(rays, torch.zeros(N, 3), seed=5) for _ in range(W)]      # one per rank, same seed: same permutations
    perms = []
    for epoch in range(3):
        seen = [[] for _ in range(W)]
        for step in range(steps):
            got = [t.next_batch(B * W, r, W)["rays"][:, 0].long() for r, t in enumerate(tables)]
            n_global = sum(g.shape[0] for g in got)
            if step < steps - 1:
                assert all(g.shape[0] == B for g in got)                 # full batches: B rows per rank, like upstream
            else:
                assert n_global == N - (steps - 1) * B * W and max(g.shape[0] for g in got) - min(g.shape[0] for g in got) <= 1
            for r in range(W):
                seen[r].append(got[r])
            assert all(t.epoch == epoch for t in tables)
        flat = torch.cat([torch.cat(s) for s in seen])
        assert flat.shape[0] == N and torch.equal(torch.sort(flat)[0], torch.arange(N))   # every row exactly once per epoch
        per_rank = [int(sum(x.shape[0] for x in s)) for s in seen]
        assert all(N // W <= n <= -(-N // W) for n in per_rank)          # upstream: ceil(N / W) each, with repeats
        perms.append(torch.cat([torch.cat([seen[r][k] for r in range(W)]) for k in range(steps)]))
    assert not torch.equal(perms[0], perms[1]) and not torch.equal(perms[1], perms[2])    # reshuffled every epoch
    for k in range(2 * steps + 1):                                        # train_steps += W per step; lr decays per epoch
        flags = sch.begin_step()
        assert sch.train_steps == (k + 1) * W
        assert abs(sch.lr(k) - 5e-4 * 0.9 ** (k // steps)) < 1e-15
        assert flags["epoch"] == ((k + 1) * W) // max(1, N // B)          # get_current_epoch (train_utils.py:117-118)


def test_ray_table_staging_buffers_keep_their_addresses():
    """next_batch(out=staging): the same rows as without, gathered into the same buffers every batch (what lets TrainLoop's fused
    step replay a HIP graph), short last batch included; last_offset = the rank's first row in the global batch."""
    from brdf_nerf_amd.raytable import synthetic_table
    a, b = synthetic_table(1000, seed=3), synthetic_table(1000, seed=3)
    st = b.staging(96)
    ptr = {k: v.data_ptr() for k, v in st.items()}
    for i in range(12):
        x, y = a.next_batch(96), b.next_batch(96, out=st)
        assert set(x) == set(y)
        for k in x:
            assert torch.equal(x[k], y[k]) and y[k].data_ptr() == ptr[k], (i, k)
    assert b.epoch == 1 and y["rays"].shape[0] == 96
    c = synthetic_table(1000, seed=3)
    parts, offs = [], []
    for r in range(3):
        c.load_state_dict(a.state_dict())
        parts.append(c.next_batch(100, rank=r, world=3, out=c.staging(34))["rgbs"].clone())
        offs.append(c.last_offset)
    assert offs == [0, 34, 67] and torch.equal(torch.cat(parts), a.next_batch(100)["rgbs"])
