"""Training loop around the fused step, without Lightning (reference main.py:31-118,
194-353, 694-736): stage schedule, on-device ray table, StepLR(0.9 / epoch), checkpoints in the reference's layout
(`{"state_dict": {"nerf_coarse.<name>": tensor}, "epoch", "global_step"}` so eval.py:26-54 / main.py:97-104 can load
them, plus the optimiser and schedule state this loop needs to resume), `opts.json` beside them.

    loop = TrainLoop(args, table)            # args: the reference's argparse namespace
    loop.run(n_steps)                        # or loop.step() one at a time
"""
import json
import os

import torch

from .distributed import world_info
from .evaluate import load_ckpt
from .field import load_model
from .losses import psnr
from .schedule import StageSchedule
from .trainer import FusedTrainer


class TrainLoop:
    def __init__(self, args, table, device=None, compute_dtype=None, process_group=None, near_far=None, trusted_ckpts=False):
        """near_far: the (near, far) pair the guided-sampling clamp uses (rendering.py:133 reads row 0 of the batch;
        satellite batches share one pair per image) - pass it to keep the step free of device->host reads."""
        self.args, self.table, self.near_far = args, table, near_far
        self._staging = None
        # trusted_ckpts: --in_ckpts points at a checkpoint written by the reference (Lightning pickles callback objects
        # beside the weights; torch's safe loader refuses those unless the caller vouches for the file)
        self.rank, self.world = world_info()
        dev = torch.device(device) if device is not None else table.device
        self.model = load_model(args, compute_dtype).to(dev)
        # --beta: the per-image embedding of main.py:113-118, looked up by render_rays (rendering.py:226-229).  The loss of this
        # model never reads beta_coarse, so like upstream the table (and the beta head) receives no gradient in training; it is
        # created under the same RNG stream position, kept in the checkpoints under the reference's key and used by rendering.
        self.embedding_t = None
        if getattr(args, "beta", False):
            self.embedding_t = torch.nn.Embedding(getattr(args, "t_embbeding_vocab", 30), args.t_embbeding_tau).to(dev)
        in_ckpts = getattr(args, "in_ckpts", "none")
        if in_ckpts != "none":                              # stage-2 warm start of the shared sub-modules (main.py:97-104)
            subs = ["fc_net", "sigma_from_xyz", "feats_from_xyz"] + ([] if args.b == True else ["rgb_from_xyzdir"])  # noqa: E712
            for sub in subs:
                load_ckpt(self.model, in_ckpts, model_name=f"nerf_coarse.{sub}", drop_len=11, trusted=trusted_ckpts)
            if self.embedding_t is not None:                # main.py:116-117
                load_ckpt(self.embedding_t, in_ckpts, model_name="embedding_t", trusted=trusted_ckpts)
        g = lambda k, d=0.0: getattr(args, k, d)
        self.trainer = FusedTrainer(self.model, args, lr=args.lr, lambda_rgb=g("lambda_rgb", 1.0), ds_lambda=g("ds_lambda"),
                                    usealldepth=bool(g("usealldepth", False)), process_group=process_group, strict_rng=False,
                                    nr_reg_an_lambda=g("nr_reg_an_lambda"), nr_reg_lr_lambda=g("nr_reg_lr_lambda"),
                                    hs_lambda=g("hs_lambda"), nr_spv_lambda=g("nr_spv_lambda") if g("nr_spv_type", 1) == 1 else 0.0)
        self.schedule = StageSchedule(args, len(table), self.world)
        self._lambda_rgb = float(g("lambda_rgb", 1.0))
        self._trusted_ckpts = bool(trusted_ckpts)
        self.global_step = 0
        self.last = {}

    # ------------------------------------------------------------------ one optimisation step
    @property
    def models(self):
        """The dict render_rays / render_image take (main.py:92-118): {'coarse': field[, 't': image embedding]}."""
        m = {"coarse": self.model}
        if self.embedding_t is not None:
            m["t"] = self.embedding_t
        return m

    def step(self):
        a, tr, sch = self.args, self.trainer, self.schedule
        flags = sch.begin_step()
        tr.lr = sch.lr(self.global_step)
        # --beta: for the first two epochs the reference scores with SNerfLoss(lambda_sc) - lambda_rgb = 1 - and only then with
        # SNerfLoss(lambda_rgb=args.lambda_rgb) (main.py:82-86, 237-238: 'beta_coarse' in results and epoch < 2)
        tr.lambda_rgb = 1.0 if (getattr(a, "beta", False) and flags["epoch"] < 2) else self._lambda_rgb
        # batches are gathered into the same staging buffers every step: unchanged addresses let the fused step replay its HIP graph
        if self._staging is None:
            self._staging = self.table.staging(a.batch_size)      # a rank's share of a global batch: at most batch_size rows
        b = self.table.next_batch(a.batch_size * self.world, self.rank, self.world, out=self._staging)
        tr.ray_offset = self.table.last_offset        # in-kernel draws are taken per GLOBAL ray: the same whatever the sharding
        has_depth = "depths" in b
        depths = b.get("depths")
        if has_depth and getattr(a, "ds_noweights", False):
            depths[:, 1] = 1.0                        # (the staging copy, not the table)
        reg = dict(tr.reg)
        if not flags["nr_reg_on"]:
            reg["nr_an"] = reg["nr_lr"] = 0.0
        if not flags["hs_on"]:
            reg["hs"] = 0.0
        saved, tr.reg = tr.reg, reg
        try:
            loss, rgb = tr.step(b["rays"], b["rgbs"], valid_depth=b.get("valid_depth"), depths=depths,
                                depth_std=b.get("depth_std"), apply_brdf=flags["apply_brdf"], apply_theta=flags["apply_theta"],
                                cos_irra_on=flags["cos_irra_on"], depth_loss_on=flags["depth_loss_on"], near_far=self.near_far,
                                gsam_only=flags["gsam_only"])
        finally:
            tr.reg = saved
        sch.end_step()
        self.global_step += 1
        self.last = {"loss": loss, "psnr": psnr(rgb, b["rgbs"]), "lr": tr.lr, **flags}
        return self.last

    def run(self, n_steps=None, log_every=0):
        n = self.schedule.max_steps - self.global_step if n_steps is None else n_steps
        for i in range(n):
            out = self.step()
            if log_every and (i + 1) % log_every == 0:
                self.check_device_faults()
            if log_every and (i + 1) % log_every == 0 and self.rank == 0:
                nan, inf = self.trainer.dropped_grad_elems()
                print(f"step {self.global_step} epoch {out['epoch']} loss {float(out['loss']):.5f} "
                      f"psnr {float(out['psnr']):.2f} lr {out['lr']:.2e} dropped non-finite gradient elements {nan}+{inf}", flush=True)
        return self.last

    def check_device_faults(self):
        """The library's sticky device fault word, read synchronously (log / checkpoint time): bit 0 = a forward launch lost an
        LDS hand-over, bit 1 = a backward-chain launch did (barrier-free trunks: field_fwd.hip / field_bwd.hip); either way the
        launch's results are invalid: raise."""
        import ctypes
        from . import _lib as L
        from .functions import _stream
        w = ctypes.c_uint(0)
        L.check(L.lib().bn_device_faults(ctypes.byref(w), _stream()), "bn_device_faults")
        if w.value & 3:
            raise RuntimeError(f"brdf_nerf_amd: a fused {'forward' if w.value & 1 else 'backward'} launch lost an LDS hand-over "
                               f"(device fault word {w.value}); its results are invalid")
        return w.value

    # ------------------------------------------------------------------ checkpoints
    def save(self, ckpts_dir, logs_dir=None):
        """`<ckpts_dir>/epoch=<e>.ckpt` in the reference's layout + resume state; `opts.json` in logs_dir (opt.py)."""
        self.check_device_faults()
        os.makedirs(ckpts_dir, exist_ok=True)
        tr = self.trainer
        sd = {f"nerf_coarse.{k}": v.detach().clone() for k, v in self.model.state_dict().items()}
        if self.embedding_t is not None:
            sd.update({f"embedding_t.{k}": v.detach().clone() for k, v in self.embedding_t.state_dict().items()})
        ckpt = {"state_dict": sd,
                "epoch": self.schedule.epoch, "global_step": self.global_step,
                "fused_trainer": {"exp_avg": tr.exp_avg.clone(), "exp_avg_sq": tr.exp_avg_sq.clone(),
                                  "adam_steps": dict(tr.adam_steps), "rng": tr.state[:2].clone()},   # (draw seed, draw counter)
                "schedule": self.schedule.state_dict(), "ray_table": self.table.state_dict()}
        path = os.path.join(ckpts_dir, f"epoch={self.schedule.epoch}.ckpt")
        torch.save(ckpt, path)
        if logs_dir is not None:
            os.makedirs(logs_dir, exist_ok=True)
            with open(os.path.join(logs_dir, "opts.json"), "w") as f:
                json.dump({k: v for k, v in sorted(vars(self.args).items()) if isinstance(v, (int, float, str, bool, type(None)))},
                          f, indent=2)
        return path

    def resume(self, path, trusted=None):
        """Checkpoints written by save() hold tensors, numbers and dicts only: the safe unpickler loads them (weights_only=True);
        trusted=True (or the constructor's trusted_ckpts) is the explicit opt-in for files with pickled objects."""
        trusted = self._trusted_ckpts if trusted is None else bool(trusted)
        ckpt = torch.load(path, map_location=self.trainer.flat_param.device, weights_only=not trusted)
        pick = lambda prefix: {k[len(prefix) + 1:]: v for k, v in ckpt["state_dict"].items() if k.startswith(prefix + ".")}
        sd = self.model.state_dict()
        sd.update(pick("nerf_coarse"))
        self.model.load_state_dict(sd)
        if self.embedding_t is not None:
            sd = self.embedding_t.state_dict()
            sd.update(pick("embedding_t"))
            self.embedding_t.load_state_dict(sd)
        tr = self.trainer
        ft = ckpt["fused_trainer"]
        tr.exp_avg.copy_(ft["exp_avg"]); tr.exp_avg_sq.copy_(ft["exp_avg_sq"])
        if "adam_steps" in ft:
            tr.adam_steps.update({k: int(v) for k, v in ft["adam_steps"].items() if k in tr.adam_steps})
        else:                                        # round-1 checkpoints: two counters (base / everything else)
            tr.adam_steps = {k: int(ft["steps_a"] if k == "base" else ft["steps_b"]) for k in tr.adam_steps}
        if "rng" in ft:                               # the in-kernel draws resume where they stopped
            tr.state[:2].copy_(ft["rng"])
            tr._rng_step = int(ft["rng"][1])
        self.schedule.load_state_dict(ckpt["schedule"])
        self.table.load_state_dict(ckpt["ray_table"])
        self.global_step = int(ckpt["global_step"])
        return ckpt
