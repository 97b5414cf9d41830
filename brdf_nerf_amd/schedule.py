"""Stage schedule of a training run, restated from the reference's LightningModule (main.py:56-76, 194-246, 265-293;
train_utils.py:117-118, 144-159) so the fused step can be driven without Lightning.

All thresholds are fractions of --max_train_steps, rounded as the reference rounds them; `train_steps` advances by the
number of GPUs per optimisation step (main.py:196, `self.train_steps += self.args.gpu_id`).  Comparisons are the
reference's (strict `>` for switching stages on, `<` for dropping the depth loss).
"""


def _round_half_even(x):
    return float(round(x))          # np.round and Python's round both round half to even


class StageSchedule:
    def __init__(self, args, n_train_rays, world=1):
        self.args = args
        self.world = max(1, int(world))
        m = args.max_train_steps
        self.brdf_on = _round_half_even(getattr(args, "brdf_on", 1.0) * m)
        self.nrrg_on = _round_half_even(getattr(args, "nrrg_on", 0.0) * m)
        self.gsam_only_on = _round_half_even(getattr(args, "gsam_only_on", 1.0) * m)
        self.cos_irra_on = _round_half_even(getattr(args, "cos_irra_on", 1.0) * m)
        self.depth = getattr(args, "ds_lambda", 0.0) > 0
        self.ds_drop = _round_half_even(getattr(args, "ds_drop", 1.0) * m) if self.depth else 0.0
        self.steps_per_epoch = max(1, n_train_rays // args.batch_size)      # get_current_epoch's divisor (train_utils.py:117-118)
        # optimiser steps per pass over the data: Lightning gives every rank a DistributedSampler share of ceil(N / world)
        # rays, i.e. ceil(N / (B world)) batches per epoch - the unit StepLR(interval='epoch') ticks in (main.py:161-167),
        # and the number of global batches RayTable serves per permutation
        self.lr_steps_per_epoch = max(1, -(-n_train_rays // (args.batch_size * self.world)))
        self.train_steps = 0
        self.lr0 = args.lr
        # Lightning stops after max_steps optimiser steps (main.py:718)
        self.max_steps = m if self.world <= 1 else int(m / self.world)

    def epoch_of(self, train_step):
        return int(train_step // self.steps_per_epoch)

    @property
    def epoch(self):
        return self.epoch_of(self.train_steps)

    def lr(self, optimiser_steps_done):
        """StepLR(step_size=1, gamma=0.9), interval 'epoch': the scheduler ticks once per finished pass over a rank's
        loader, i.e. per `lr_steps_per_epoch` OPTIMISER steps - with W GPUs an epoch is W times shorter, so the rate
        decays W times faster per step, as upstream under DDP."""
        return self.lr0 * 0.9 ** (optimiser_steps_done // self.lr_steps_per_epoch)

    def begin_step(self):
        """Advance the counters for one optimisation step and return the flags the step runs with."""
        self.train_steps += self.world
        t = self.train_steps
        flags = dict(
            gsam_only=t > self.gsam_only_on,
            apply_brdf=t > self.brdf_on,
            apply_theta=t > self.brdf_on * 2,
            cos_irra_on=t > self.cos_irra_on,
            depth_loss_on=self.depth and t < self.ds_drop,
            nr_reg_on=t > self.nrrg_on,
            hs_on=self.epoch > 2,
            epoch=self.epoch,
        )
        return flags

    def end_step(self):
        """main.py:246: the density noise decays by 0.9 after every step (on the shared args, like the reference)."""
        self.args.noise_std *= 0.9

    def state_dict(self):
        return {"train_steps": self.train_steps, "noise_std": self.args.noise_std}

    def load_state_dict(self, sd):
        self.train_steps = int(sd["train_steps"])
        self.args.noise_std = float(sd["noise_std"])
