"""Fused training step for spsbrdf-nerf on one GPU per process (reference: NeRF_pl.training_step, main.py:194-353,
+ Adam, main.py:147-168, + Lightning DDP's gradient all-reduce, main.py:720-731).

One step = render_rays (pass 1 sigma-only, depth-guided resampling, pass 2) + SNerfLoss [+ DepthLoss] + backward +
[RCCL all-reduce of ONE flat fp32 gradient buffer] + Adam.  Everything numerical runs in the HIP library; torch
autograd is used only for the per-ray loss glue ((R,3)/(R,) tensors).  Parameters live in one flat buffer (the
nn.Parameters of the drop-in module are views into it, state_dict keys unchanged), so the optimizer and the
collective are single launches over ~10 MB.
"""
from collections import OrderedDict

import torch

from . import functions as Fn
from . import _lib as L
from . import losses
from .distributed import allreduce_sum_
from .rendering import shade, shade_desc, get_z_vals, inference, sun_far


class FusedTrainer:
    def __init__(self, model, args, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, lambda_rgb=1.0, ds_lambda=0.0,
                 usealldepth=False, process_group=None, strict_rng=True, reuse_coarse=True, nr_reg_an_lambda=0.0,
                 nr_reg_lr_lambda=0.0, hs_lambda=0.0, nr_spv_lambda=0.0, data_parallel=True):
        """The regulariser lambdas are the reference's --nr_reg_an_lambda / --nr_reg_lr_lambda / --hs_lambda /
        --nr_spv_lambda (opt.py:232-246, all 0 by default); WHEN they act (train_steps > nrrg_on, epoch > 2, main.py:271-327)
        is the caller's schedule: pass regularisers=False to step() until then."""
        self.model, self.args = model, args
        self.reg = dict(nr_an=nr_reg_an_lambda, nr_lr=nr_reg_lr_lambda, hs=hs_lambda, nr_spv=nr_spv_lambda)
        self.fused_glue = True      # Lambertian steps: one launch for shading + losses + their gradients
        # The BRDF models have genuine singularities (Hapke's azimuth phi = acos(.) at phi -> 0, grazing angles): autograd -
        # and the kernels, which follow it - return inf/NaN there, and ONE such ray would poison every weight through Adam.
        # The fused step drops non-finite per-sample gradients instead (a deviation only where the reference's own
        # gradient is non-finite); set False for strict autograd semantics.
        self.sanitize_grads = True
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.lambda_rgb, self.ds_lambda, self.usealldepth = lambda_rgb, ds_lambda, usealldepth
        self.pg, self.strict_rng, self.reuse_coarse = process_group, strict_rng, reuse_coarse
        # data_parallel=False: a purely local trainer even inside an initialised process group (reference runs in tests)
        self.world = 1
        if data_parallel and (process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized())):
            self.world = torch.distributed.get_world_size(process_group)
        self.nr_lr = model.normal in ("analystic_learned", "learned")
        self.nr_an = model.normal in ("analystic_learned", "analystic")
        self._flatten()
        self._bufs = OrderedDict()      # (name, shape, dtype) -> scratch tensor, see _buf
        self._touched = None            # list collecting the scratch a step body uses while it is being captured
        self.max_buf_variants = 3
        # Launch-lean step (round 3; own draws only): in-kernel draws, pass-1 compositing fused with the resampling, the merged
        # sample set composited through its sort index, one-launch Lambertian tail, fold / unfold kernels, one Adam launch for
        # all groups - and, for a step whose inputs keep their addresses, replayed from a captured HIP graph.
        self.lean = True
        self.merge_passes = True        # one output array / stash / backward for both passes (False: two backward launch sets)
        self.overlap_allreduce = True   # world > 1: all-reduce the trunk's gradient while the rest of the backward still runs
        self._kind_cache = {}
        self.ray_offset = 0             # index of this rank's first ray in the global batch: in-kernel draws are taken per GLOBAL ray
        self.keep_grads = False         # True: the optimiser launch leaves the step's gradient in flat_grad (tests, diagnostics)
        self.seed_hook = None           # callable(name, tensor): sees / may overwrite "z2" (guided depths), "out_all" (the per-sample field
                                        # outputs of both passes, merged step) and "d_all" (the gradient rows the field backward starts
                                        # from) of a launch-lean step (tests, profiles/diag_c5_rows.py; such steps are not captured)
        # The LOGGED loss as a fixed-order sum of the per-ray terms (one more small launch per step) instead of float atomics into
        # 64 slots of the step state: a run's printed losses repeat bit for bit like its gradients do (rounds 3-4
        # did this in deterministic mode only).  False: the atomics (the loss ring of the step state), one launch fewer.
        self.repeatable_loss = True
        self.use_graph = True
        self.graph_after = 3            # eager steps with an unchanged signature before the step is captured
        self.max_graphs = 16            # captured steps kept (least recently used evicted; each owns a private memory pool)
        self._graphs, self._sig_seen, self._nf_dev, self._no_graph = OrderedDict(), OrderedDict(), {}, set()
        self._graph_cap_warned = False
        self.state = Fn.new_step_state(self.flat_param.device, torch.initial_seed(), lr)
        self._rng_step, self._state_lr, self._grads_clean, self._state_noise = 0, float(lr), True, 0.0
        self._state_adam = [0, 0, 0, 0]
        # sanitize_grads bookkeeping: NaN / Inf elements of d(loss)/d(per-sample outputs) zeroed so far, counted on the
        # device (bn_count_nonfinite, no host round trip); read it with dropped_grad_elems()
        self._nonfinite = torch.zeros(2, dtype=torch.int64, device=self.flat_param.device)

    def seed_draws(self, seed):
        """Key of the in-kernel draws of the launch-lean step (default: torch.initial_seed() at construction)."""
        self.state[0] = int(seed) & 0x7FFFFFFFFFFFFFFF

    # ------------------------------------------------------------------ flat parameter / gradient storage
    def _flatten(self):
        """One flat fp32 buffer, in Adam GROUPS: [base | BRDF heads | theta head].  torch.optim.Adam (the reference's
        optimiser) skips a parameter whose grad is None and keeps a step counter per parameter, so a head that joins the
        graph later (BRDF heads at brdf_on, Hapke's theta head at 2 * brdf_on, main.py:207-210) starts its bias
        corrections at step 1 then: every group has its own counter and is stepped only while it is in the graph."""
        model = self.model
        named = dict(model.named_parameters())
        base = set(model.spec(False, False, self.nr_lr, beta=False).used_param_names())
        theta = {n for n in named if n.startswith("theta_from_xyz.")}
        groups = [("base", [n for n in named if n in base]),
                  ("brdf", [n for n in named if n not in base and n not in theta]),
                  ("theta", [n for n in named if n in theta])]
        offs, tot = {}, 0
        self.groups = []                               # (name, lo, hi) element ranges of the flat buffer
        for gname, names in groups:
            lo = tot
            for n in names:
                offs[n] = tot
                tot += (named[n].numel() + 3) // 4 * 4
            if tot > lo:
                self.groups.append((gname, lo, tot))
        self.n_base = self.groups[0][2]
        # the trunk's parameters (fc_net.*) as a leading, contiguous range of the flat buffer: the first all-reduce bucket
        trunk = [n for n in named if n.startswith("fc_net.")]
        ends = [offs[n] + (named[n].numel() + 3) // 4 * 4 for n in trunk]
        self.n_trunk = max(ends) if trunk and max(ends) <= min([offs[n] for n in named if n not in trunk] + [tot]) else 0
        self.adam_steps = {gname: 0 for gname, _, _ in self.groups}
        dev = next(model.parameters()).device
        self.flat_param = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros_like(self.flat_param)
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        self.grad_views = {}
        with torch.no_grad():
            for _, names in groups:
                for n in names:
                    p, o = named[n], offs[n]
                    view = self.flat_param[o:o + p.numel()].view(p.shape)
                    view.copy_(p.data)
                    p.data = view                     # the module's parameters now alias the flat buffer
                    self.grad_views[n] = self.flat_grad[o:o + p.numel()].view(p.shape)

    # kept for checkpoints written by round 1 (two counters: base / everything else)
    @property
    def steps_a(self):
        return self.adam_steps["base"]

    @property
    def steps_b(self):
        return self.adam_steps.get("brdf", self.adam_steps.get("theta", 0))

    def dropped_grad_elems(self):
        """(NaN, Inf) elements of the per-sample output gradients zeroed by sanitize_grads since construction (host read)."""
        return tuple(int(v) for v in self._nonfinite.tolist())

    @property
    def dropped_samples(self):
        return sum(self.dropped_grad_elems())

    def _buf(self, key, shape, dtype=torch.float32):
        """Scratch buffer `key` of this shape and dtype.  A captured step bakes the ADDRESSES of its scratch into the graph, so
        a buffer is never dropped because a step of another shape (the short last batch of an epoch, raytable.py) asked for the
        same name: buffers are kept per (name, shape, dtype), and a capture pins every buffer its body touched (`_touched`,
        stored in the graph entry).  At most `max_buf_variants` shapes per name stay in the table (least recently used first
        out); a variant a live graph still references survives its eviction from the table through that reference."""
        k = (key, tuple(shape), dtype)
        b = self._bufs.get(k)
        if b is None:
            b = torch.empty(shape, dtype=dtype, device=self.flat_param.device)
            same = [q for q in self._bufs if q[0] == key]
            for q in same[:max(0, len(same) + 1 - self.max_buf_variants)]:
                del self._bufs[q]
            self._bufs[k] = b
        else:
            self._bufs.move_to_end(k)
        if self._touched is not None:
            self._touched.append(b)
        return b

    # ------------------------------------------------------------------ one step
    def step(self, rays, rgbs, valid_depth=None, depths=None, depth_std=None, apply_brdf=False, apply_theta=False,
             cos_irra_on=False, depth_loss_on=True, near_far=None, regularisers=True, gsam_only=False):
        """gsam_only (main.py:201-203, rendering.py:266-269): pass 1 only guides the sampling; the step renders, and
        back-propagates through, the G guided samples alone."""
        model, args = self.model, self.args
        S, G = args.n_samples, args.guided_samples
        R = rays.shape[0]
        dev = rays.device
        spec = model.spec(apply_brdf, apply_theta, self.nr_lr, self.nr_an, beta=False)   # (field.py spec(): the loss never reads beta)
        reg0 = self.reg if regularisers else {}
        # --MultiBRDF (one BRDF per sample, spsbrdfnerf.py:289-307,350-352) is a lean step (_lean_body: one per-sample shading launch
        # each way, csrc/sample_brdf.hip); the regularisers ride in the compositing kernels as they do for one BRDF per ray.
        # The sun-visibility pass (rendering.py:244-259) is a lean step where the reference runs it (gsam_only): a per-ray BRDF reads
        # it through ONE per-ray factor (spsbrdfnerf.py:354), a per-sample BRDF or a Lambertian rgb through a per-sample one
        # (:265-273) that the same launch applies; the sun pass's own noise draws keep the general path
        sun_on = getattr(model, "sun_v", "none") == "analystic" and apply_brdf
        sun_lean = sun_on and gsam_only and args.noise_std == 0 and args.data == "sat"
        if (self.lean and not self.strict_rng and self.reuse_coarse
                and (not sun_on or sun_lean)
                and rays.dtype == torch.float32 and rays.is_contiguous() and rays.shape[1] >= 8 and S + G <= 512):
            return self._step_lean(spec, rays, rgbs, valid_depth, depths, depth_std, apply_brdf, apply_theta, cos_irra_on,
                                   depth_loss_on, near_far, regularisers, gsam_only)
        self._grads_clean = False
        named = model.named()
        packed = model.repack(spec)
        near, far = rays[:, 6:7], rays[:, 7:8]
        rays_d = rays[:, 3:6]
        sun_d = rays[:, 8:11] if args.data == "sat" else torch.ones_like(rays[:, 0:3])
        need_noise = self.strict_rng or args.noise_std != 0
        C = spec.out_channels
        S2 = G if gsam_only else S + G
        reuse = self.reuse_coarse and not gsam_only
        with torch.no_grad():
            z = Fn.stratified_z(near, far, torch.rand(R, S, device=dev))
            noise1 = torch.randn(R, S, device=dev) if need_noise else None
            noise1 = noise1 if args.noise_std != 0 else None
            if reuse:
                # pass 1 = FULL forward on the S coarse samples, kept for the backward: the field is a pointwise function
                # of xyz, so pass 2 only has to evaluate the G new samples (the reference re-evaluates all S+G: same values)
                out1 = self._buf("out1", (R * S, C))
                stash1 = self._buf("stash1", (Fn.field_stash_bytes(spec, R * S),), torch.uint8)
                Fn.field_forward_raw(spec, named, packed, out1, stash1, rays=rays, z=z)
                _, _, w1, d1, _ = Fn.composite_forward_raw(z, out1.view(R, S, C), noise1, args.noise_std)
            else:
                sig = Fn.field_sigma(spec, named, packed, rays=rays, z=z).view(R, S)
                _, _, w1, d1 = Fn.composite(z, sig, noise1, args.noise_std)
            sun_res = None
            if getattr(model, "sun_v", "none") == "analystic" and apply_brdf:
                # sun-visibility pass (rendering.py:244-259), detached upstream: sigma only along the sun direction from
                # the pass-1 surface point; same draws, in the same order, as render_rays
                if not gsam_only:
                    raise NotImplementedError("--sun_v analystic needs gsam_only=True (SURVEY quirk 2: the reference raises a "
                                              "shape error with the merged S+G sample set)")
                far_sun = sun_far(d1, rays_d, sun_d)
                z_sun = get_z_vals(G, dev, far_sun * 0.01, far_sun)
                sun_rays = torch.cat([rays[:, 0:3] + rays_d * d1.unsqueeze(-1), sun_d], -1).contiguous()
                rs, _ = inference(model, args, None, z_sun, rays_d=sun_d, mode="train", sigma_only=True, _rays=sun_rays,
                                  _packed=packed)
                sun_res = {"sun": rs["transparency"].unsqueeze(-1), "weights_sc": rs["weights"]}
            # depth-guided resampling + merge
            u = torch.rand(R, G, device=dev)
            use_t = tdep = tstd = u_t = trow = None
            if valid_depth is not None:
                valid = (valid_depth > 0) if self.strict_rng else None
                # the reference draws rand(n_valid, G); a (R, G) draw indexed by the valid-row rank is the same
                # distribution and needs no host sync
                u_t = torch.rand(R, G, device=dev)
                tdep = depths[:, 0].float().contiguous()
                tstd = depth_std.float().reshape(-1).contiguous()
                if self.strict_rng:     # replaying the reference's stream: row k of its (n_valid, G) draw belongs to the k-th valid ray
                    use_t = valid.float().contiguous()
                    trow = (torch.cumsum(valid.int(), 0) - 1).clamp_min(0).int().contiguous()
                else:                   # own draws: row r for ray r (no row table, no conversions: the kernel tests use_target > 0)
                    use_t = valid_depth if (valid_depth.dtype == torch.float32 and valid_depth.is_contiguous()) \
                        else (valid_depth > 0).float().contiguous()
            # the clamp window is the FIRST ray's (near, far) (rendering.py:133): read on the device from rays[0, 6:8]
            # unless the caller passes the pair
            near0, far0 = near_far if near_far is not None else (rays[0, 6:8], None)
            z2, z_all, idx = Fn.guided_samples(z, w1, d1, u, near0, far0, args.std_range, use_t, tdep, tstd, u_t, trow,
                                               merge=not gsam_only)
            if gsam_only:
                z_all = z2
            noise2 = torch.randn(R, S2, device=dev) if need_noise else None
            noise2 = noise2 if args.noise_std != 0 else None
            if reuse:
                out2 = self._buf("out2", (R * G, C))
                stash2 = self._buf("stash2", (Fn.field_stash_bytes(spec, R * G),), torch.uint8)
                Fn.field_forward_raw(spec, named, packed, out2, stash2, rays=rays, z=z2)
                idx_c = idx.unsqueeze(-1).expand(-1, -1, C)
                out3 = torch.cat([out1.view(R, S, C), out2.view(R, G, C)], 1).gather(1, idx_c)   # depth-sorted order
            else:
                # pass 2: full field on all S+G samples with activation stash
                out = self._buf("out", (R * S2, C))
                stash = self._buf("stash", (Fn.field_stash_bytes(spec, R * S2),), torch.uint8)
                Fn.field_forward_raw(spec, named, packed, out, stash, rays=rays, z=z_all)
                out3 = out.view(R, S2, C)
            alphas, trans, weights, depth, acc = Fn.composite_forward_raw(z_all, out3, noise2, args.noise_std)
        reg = self.reg if regularisers else {}
        lambertian = (len(spec.heads) == 1 and not spec.normal_an and not spec.normal_lr and reg.get("hs", 0) <= 0
                      and sun_res is None)
        n_leaf = {}                                   # channel offset -> per-sample normal leaf
        grads = ()
        d_out3 = None
        if lambertian and self.fused_glue:
            # Lambertian step: shading + SNerfLoss + DepthLoss + their gradients in ONE launch (bn_lambert_loss)
            use_ds = self.ds_lambda > 0 and depth_loss_on and valid_depth is not None
            with torch.no_grad():
                loss, rgb_out, d_acc, d_depth, d_weights = Fn.lambert_loss(
                    acc, weights, z_all, depth, rgbs, model.rgb_padding, self.lambda_rgb,
                    valid_depth if use_ds else None, depths[:, 0] if use_ds else None, depths[:, 1] if use_ds else None,
                    depth_std if use_ds else None, self.ds_lambda if use_ds else 0.0, self.usealldepth)
            res = {"rgb": rgb_out}
        else:
            # ray-level loss glue under autograd (leaves: acc, depth, weights; the per-sample normals too when a regulariser
            # reads them)
            acc_l, depth_l, weights_l = acc.requires_grad_(True), depth.requires_grad_(True), weights.requires_grad_(True)
            # shade() reads PER-SAMPLE channels of the field output when every sample has its own BRDF (--MultiBRDF,
            # spsbrdfnerf.py:289-307,350-352) or its own sun visibility (:265-273): the loss then depends on out3 directly,
            # not only through the composited sums, so out3 is a leaf as well and its gradient joins d_out below
            per_sample = (bool(model.MultiBRDF) and apply_brdf) or sun_res is not None
            out3_l = out3.detach().requires_grad_(True) if per_sample else out3
            res, _ = shade(model, args, spec, out3_l, z_all, alphas, trans, weights_l, depth_l, acc_l, rays_d, sun_d, apply_brdf,
                           cos_irra_on, sun_res=sun_res)
            loss = losses.snerf_loss(res["rgb"], rgbs, self.lambda_rgb)
            if self.ds_lambda > 0 and depth_loss_on and valid_depth is not None:
                loss = loss + losses.depth_loss(z_all, depth_l, weights_l, depths[:, 0], depths[:, 1], valid_depth, depth_std,
                                                self.ds_lambda, self.usealldepth)
            def normal_leaf(key):
                c0 = spec.ch_normal_an if key == "normal_an" else spec.ch_normal_lr
                if c0 not in n_leaf:
                    n_leaf[c0] = out3[..., c0:c0 + 3].detach().clone().requires_grad_(True)
                return n_leaf[c0]
            if reg.get("nr_an", 0) > 0 and spec.normal_an:
                loss = loss + losses.normal_reg_loss(normal_leaf("normal_an"), weights_l, -rays_d, reg["nr_an"])[0]
            if reg.get("nr_lr", 0) > 0 and spec.normal_lr:
                loss = loss + losses.normal_reg_loss(normal_leaf("normal_lr"), weights_l, -rays_d, reg["nr_lr"])[0]
            if reg.get("hs", 0) > 0:
                loss = loss + losses.hard_surface_loss(z_all, depth_l, weights_l, reg["hs"])
            if abs(reg.get("nr_spv", 0)) > 1e-5 and spec.normal_an and spec.normal_lr:   # nr_spv_type 1 (main.py:297-303)
                loss = loss + losses.normal_loss(weights_l, normal_leaf("normal_an"), normal_leaf("normal_lr"), reg["nr_spv"])
            leaves = [acc_l, depth_l, weights_l] + list(n_leaf.values()) + ([out3_l] if per_sample else [])
            grads = torch.autograd.grad(loss, leaves, allow_unused=True)
            d_acc, d_depth, d_weights = grads[:3]
            d_out3 = grads[-1] if per_sample else None
            grads = grads[:3 + len(n_leaf)]
        with torch.no_grad():
            if d_acc is None:      # nothing read the composited sums: the kernel must still write (zero) channel gradients
                d_acc = torch.zeros(R, C, dtype=torch.float32, device=dev)
            else:
                d_acc = d_acc.contiguous()
                d_acc[:, 3] = 0
            d_out = Fn.composite_backward_raw(z_all, out3, None if d_weights is None else d_weights.contiguous(),
                                              None if d_depth is None else d_depth.contiguous(), d_acc, noise2, args.noise_std,
                                              self._buf("d_out", (R, S2, C)))
            for c0, dn in zip(n_leaf.keys(), grads[3:]):
                if dn is not None:
                    d_out[..., c0:c0 + 3] += dn          # regulariser gradients on the per-sample normals
            if not (lambertian and self.fused_glue) and d_out3 is not None:
                d_out3 = d_out3.clone()
                d_out3[..., 3] = 0                       # sigma acts through the compositing only (already in d_out)
                d_out += d_out3                          # per-sample shading terms (MultiBRDF / per-sample sun visibility)
            if self.sanitize_grads and not (lambertian and self.fused_glue):
                Fn.count_nonfinite(d_out, self._nonfinite)
                torch.nan_to_num_(d_out, nan=0.0, posinf=0.0, neginf=0.0)
            self.flat_grad.zero_()
            if reuse:
                d_cat = self._buf("d_cat", (R, S2, C)).scatter_(1, idx_c, d_out)             # back to [coarse | guided] order
                d1o, d2o = d_cat[:, :S].contiguous().view(R * S, C), d_cat[:, S:].contiguous().view(R * G, C)
                Fn.field_backward_raw(spec, named, self.grad_views, packed, out1, d1o, stash1, rays=rays, z=z, unfold=False)
                Fn.field_backward_raw(spec, named, self.grad_views, packed, out2, d2o, stash2, rays=rays, z=z2)
            else:
                Fn.field_backward_raw(spec, named, self.grad_views, packed, out, d_out.view(R * S2, C), stash, rays=rays, z=z_all)
            if self.world > 1:
                allreduce_sum_(self.flat_grad, self.pg)       # RCCL over xGMI: ONE collective over the ~10 MB flat buffer
            self._adam(apply_brdf, apply_theta)
        return loss.detach(), res["rgb"].detach()

    # ------------------------------------------------------------------ launch-lean step
    def _sample_desc(self, spec, apply_brdf, cos_irra_on):
        """bn_shade_desc of the per-sample BRDF launches of a MultiBRDF lean step (cached: its fields are launch arguments)."""
        key = ("sample", spec.key(), bool(apply_brdf), bool(cos_irra_on))
        d = self._kind_cache.get(key)
        if d is None:
            d = self._kind_cache[key] = shade_desc(self.model, self.args, spec, apply_brdf, cos_irra_on)
        return d

    def _shade_kind(self, spec, apply_brdf, cos_irra_on):
        """BN_SHADE_* kind of the ray-level shading for this head set (cached: step() asks on every call)."""
        key = (spec.key(), bool(apply_brdf), bool(cos_irra_on))
        k = self._kind_cache.get(key)
        if k is None:
            k = self._kind_cache[key] = shade_desc(self.model, self.args, spec, apply_brdf, cos_irra_on).kind
        return k

    def _near_far(self, rays, near_far):
        if near_far is None:
            return rays[0, 6:8]
        key = (float(near_far[0]), float(near_far[1]))
        t = self._nf_dev.get(key)
        if t is None:
            t = self._nf_dev[key] = torch.tensor(key, dtype=torch.float32, device=rays.device)
        return t

    def _sync_state(self, on):
        """Host-side hyper-parameters -> device step state, only when they changed (learning-rate decay, a step taken on the
        other path)."""
        _, lr_v, steps_v, _ = Fn.state_views(self.state)
        if float(self.lr) != self._state_lr:
            lr_v.fill_(float(self.lr))
            self._state_lr = float(self.lr)
        # --noise_std decays after EVERY step (schedule.py:66, as main.py:246): the kernels read it from the device state, so the
        # launch signature only knows noise on / off (a value in the launch arguments kept such steps eager)
        nz = self._noise_f32()
        if nz != self._state_noise:
            Fn.set_state_noise(self.state, nz)
            self._state_noise = nz
        want = [self.adam_steps[g] for g, _, _ in self.groups] + [0] * (4 - len(self.groups))
        if want != self._state_adam:
            Fn.set_adam_steps(self.state, want, self.betas)
            self._state_adam = list(want)

    def _noise_f32(self):
        """--noise_std as the kernels see it (float32): a Python float that underflows float32 is OFF."""
        import numpy as np
        return float(np.float32(self.args.noise_std))

    def _step_lean(self, spec, rays, rgbs, valid_depth, depths, depth_std, apply_brdf, apply_theta, cos_irra_on, depth_loss_on,
                   near_far, regularisers, gsam_only=False):
        model, args = self.model, self.args
        reg = self.reg if regularisers else {}
        lambertian = (len(spec.heads) == 1 and not spec.normal_an and not spec.normal_lr and reg.get("hs", 0) <= 0)
        if not self._grads_clean:          # the general path leaves its gradient in the flat buffer
            self.flat_grad.zero_()
            self._grads_clean = True
        # fp32 / contiguous views of the inputs with STABLE addresses: a tensor that already is one is used as it is; anything
        # else is copied into a trainer-owned staging buffer (a fresh temporary per step would give every step a new graph
        # signature: up to max_graphs captures, each pinning its temporary, then eager forever)
        def stage(key, t, flat=False):
            if t is None:
                return None
            if t.dtype == torch.float32 and t.is_contiguous():
                return t.reshape(-1) if flat else t
            buf = self._buf("in_" + key, (t.numel(),) if flat else tuple(t.shape))
            buf.copy_(t.reshape(-1) if flat else t)
            return buf
        rgbs = stage("rgbs", rgbs)
        use_ds = self.ds_lambda > 0 and depth_loss_on and valid_depth is not None
        valid_depth, depths = stage("valid_depth", valid_depth), stage("depths", depths)
        depth_std = stage("depth_std", depth_std, flat=True)
        nf = self._near_far(rays, near_far)
        on = {"base": True, "brdf": bool(apply_brdf), "theta": bool(apply_brdf and apply_theta)}
        active = [on[g] for g, _, _ in self.groups]
        self._sync_state(on)
        slot = self._rng_step % 64
        body = lambda: self._lean_body(spec, rays, rgbs, valid_depth, depths, depth_std, apply_brdf, cos_irra_on, use_ds, nf, reg,
                                       lambertian, active, gsam_only)
        res = None
        if self.use_graph and self.world == 1 and self.seed_hook is None:
            sig = (spec.key(), rays.shape, rays.data_ptr(), rgbs.data_ptr(), None if valid_depth is None else valid_depth.data_ptr(),
                   None if depths is None else depths.data_ptr(), None if depth_std is None else depth_std.data_ptr(),
                   nf.data_ptr(), use_ds, tuple(active), float(self.ds_lambda), float(self.lambda_rgb), bool(self.usealldepth),
                   L.deterministic(), int(self.ray_offset), bool(self.keep_grads), bool(self.merge_passes), bool(apply_brdf),
                   bool(cos_irra_on), float(reg.get("hs", 0)), bool(self.sanitize_grads), float(reg.get("nr_an", 0)),
                   float(reg.get("nr_lr", 0)), bool(gsam_only), self._noise_f32() != 0.0, float(reg.get("nr_spv", 0)),
                   bool(self.repeatable_loss),
                   # model switches that change WHICH launches the step consists of (they are attributes, not part of the head set)
                   str(getattr(model, "sun_v", "none")), bool(model.MultiBRDF), str(args.data),
                   # and the host-side constants baked into a captured launch's arguments
                   int(args.n_samples), int(args.guided_samples), float(args.std_range), float(model.rgb_padding),
                   int(getattr(args, "funcH", 1)), float(getattr(args, "hpk_scl", 1.0)), float(getattr(args, "fresnel_f0", 0.04)),
                   int(getattr(args, "shell_hapke", 0)), tuple(float(b) for b in self.betas), float(self.eps), float(self.wd))
            ent = self._graphs.get(sig)
            if ent is not None:
                self._graphs.move_to_end(sig)
                ent[0].replay()
                res = ent[1]
            else:
                n = self._sig_seen.pop(sig, 0) + 1
                self._sig_seen[sig] = n
                while len(self._sig_seen) > 256:           # bounded: signatures that never recur (short last batches, changing flags)
                    self._sig_seen.popitem(last=False)
                if n > self.graph_after and sig not in self._no_graph:
                    if len(self._graphs) >= self.max_graphs:     # evict the least recently replayed capture (and its inputs / pool)
                        self._graphs.popitem(last=False)
                        if not self._graph_cap_warned:
                            import warnings
                            warnings.warn(f"brdf_nerf_amd: more than {self.max_graphs} distinct training-step signatures were captured; the "
                                          f"least recently used graph is dropped (inputs whose addresses change every step defeat the replay: "
                                          f"keep batches in fixed buffers, e.g. RayTable.next_batch(out=...))")
                            self._graph_cap_warned = True
                    # keep the inputs AND the scratch alive with the graph: their addresses are baked into it (a step of
                    # another shape must not free what this graph writes on replay: ADVICE r4)
                    keep = [rays, rgbs, valid_depth, depths, depth_std, nf]
                    g = torch.cuda.CUDAGraph()
                    torch.cuda.synchronize()
                    self._touched = []
                    try:
                        with torch.cuda.graph(g):
                            out = body()
                    except Exception as e:      # a runtime that cannot capture this step: it stays eager (the capture launched nothing)
                        import warnings
                        warnings.warn(f"brdf_nerf_amd: HIP graph capture of the training step failed ({type(e).__name__}: {e}); "
                                      f"this step signature runs eagerly")
                        self._no_graph.add(sig)
                        self._touched = None
                        torch.cuda.synchronize()
                    else:
                        keep += self._touched
                        self._touched = None
                        self._graphs[sig] = (g, out, tuple(keep))
                        g.replay()            # the capture itself does not execute: this is the step
                        res = out
        if res is None:
            res = body()
        self._grads_clean = not self.keep_grads
        self._rng_step += 1
        for (gname, _, _), a in zip(self.groups, active):
            if a:
                self.adam_steps[gname] += 1
        self._state_adam = [self.adam_steps[g] for g, _, _ in self.groups] + [0] * (4 - len(self.groups))
        loss, rgb = res
        if loss is None:
            loss = Fn.state_views(self.state)[3][slot]
        return loss.detach(), rgb.detach()

    def _lean_body(self, spec, rays, rgbs, valid_depth, depths, depth_std, apply_brdf, cos_irra_on, use_ds, nf, reg, lambertian,
                   active, gsam_only=False):
        model, args = self.model, self.args
        S, G = args.n_samples, args.guided_samples
        R, C = rays.shape[0], spec.out_channels
        st = self.state
        named = model.named()
        with torch.no_grad():
            packed = model.repack(spec)                                   # fold + pack: 2 launches
            z = Fn.stratified_z_rng(rays, S, st, self._buf("z", (R, S)), ray_offset=self.ray_offset)
            # ONE output array, one stash and one gradient array for both passes: pass 2 is a second forward launch (it needs
            # pass 1's result to place its samples) into rows [R S, R (S + G)) of the same set, and the backward - chain,
            # weight gradients, skinny jobs - runs ONCE over all R (S + G) points (two launch sets before round 3's merge)
            tile = 64 if spec.dtype == L.BN_F32 else 128
            merged = self.merge_passes and (R * S) % tile == 0 and not gsam_only
            n_all = R * (S + G)
            has_t = valid_depth is not None
            # --noise_std (models/spsbrdfnerf.py:57-59, multiplied by 0.9 after every step, main.py:246): in-kernel normal draws, one stream for
            # the pass-1 compositing and one for the final compositing of the merged set (the general step draws randn(R, S) and
            # randn(R, S + G) at the same two places)
            nz1 = Fn.noise_arg(st, self._noise_f32(), L.BN_RNG_NOISE_COARSE, self.ray_offset, from_state=True)
            nz2 = Fn.noise_arg(st, self._noise_f32(), L.BN_RNG_NOISE_MERGED, self.ray_offset, from_state=True)
            bufs = {"z2": self._buf("z2", (R, G)), "z_all": self._buf("z_all", (R, S + G)),
                    "idx": self._buf("idx", (R, S + G), torch.int64)}
            if gsam_only:
                # gsam_only stage (main.py:201-203, rendering.py:266-269): pass 1 is a sigma-only forward that only places the
                # guided samples; the step renders, and back-propagates through, the G guided samples alone
                sig1 = Fn.field_sigma(spec, named, packed, rays=rays, z=z, out=self._buf("sig1", (R * S,)))
                sun_on = getattr(model, "sun_v", "none") == "analystic" and apply_brdf
                if sun_on:
                    bufs = dict(bufs, w1=self._buf("w1", (R, S)), d1=self._buf("d1", (R,)))
                cg = Fn.composite_guided(z, None, G, nf, args.std_range, valid_depth if has_t else None,
                                         depths[:, 0] if has_t else None, depth_std if has_t else None, state=st, bufs=bufs,
                                         ray_offset=self.ray_offset, sigma=sig1.view(R, S), noise=nz1, want_pass1=sun_on)
                z2 = cg[0]
                if sun_on:
                    # sun-visibility pass (rendering.py:244-259), detached upstream: sigma only along the sun direction from the
                    # pass-1 surface point, G stratified samples in [0.01 far_sun, far_sun] (in-kernel draws, stream BN_RNG_SUN);
                    # the BRDF-shaded rgb reads the transparency in front of its LAST sample (spsbrdfnerf.py:354)
                    d1 = cg[4]
                    rays_d, sun_d = rays[:, 3:6], rays[:, 8:11]
                    far_sun = sun_far(d1, rays_d, sun_d)
                    z_sun = Fn.stratified_z_rng(None, G, st, self._buf("z_sun", (R, G)), ray_offset=self.ray_offset,
                                                near_far=torch.cat([far_sun * 0.01, far_sun], -1).contiguous(), stream_id=L.BN_RNG_SUN)
                    sun_rays = torch.cat([rays[:, 0:3] + rays_d * d1.unsqueeze(-1), sun_d], -1).contiguous()
                    sig_sun = Fn.field_sigma(spec, named, packed, rays=sun_rays, z=z_sun)
                    sun_T = Fn.composite(z_sun, sig_sun.view(R, G), None, 0.0)[1]       # transparency in front of each sample
                    sun_irr = sun_T[:, -1]
                if self.seed_hook is not None:
                    self.seed_hook("z2", z2)
                out2 = self._buf("out2", (R * G, C))
                stash2 = self._buf("stash2", (Fn.field_stash_bytes(spec, R * G),), torch.uint8)
                Fn.field_forward_raw(spec, named, packed, out2, stash2, rays=rays, z=z2)
                # the "merged set" is the guided block alone: no sort index, one source block
                z_all, idx, out1v, out2v, S = z2, None, out2.view(R, G, C), None, G
                d_all = self._buf("d_all", (R * G, C))
                d1o, d2o = d_all.view(R, G, C), None
            elif merged:
                out_all = self._buf("out_all", (n_all, C))
                stash_all = self._buf("stash_all", (Fn.field_stash_bytes(spec, n_all),), torch.uint8)
                out1, out2 = out_all[:R * S], out_all[R * S:]
                Fn.field_forward_raw(spec, named, packed, out_all, stash_all, rays=rays, z=z, point_offset=0, total_points=n_all)
            else:
                out1 = self._buf("out1", (R * S, C))
                stash1 = self._buf("stash1", (Fn.field_stash_bytes(spec, R * S),), torch.uint8)
                Fn.field_forward_raw(spec, named, packed, out1, stash1, rays=rays, z=z)
            if not gsam_only:
                out1v = out1.view(R, S, C)
                z2, z_all, idx = Fn.composite_guided(z, out1v, G, nf, args.std_range, valid_depth if has_t else None,
                                                     depths[:, 0] if has_t else None, depth_std if has_t else None, state=st, bufs=bufs,
                                                     ray_offset=self.ray_offset, noise=nz1)
                if self.seed_hook is not None:   # (and the guided depths pass 2 is evaluated at)
                    self.seed_hook("z2", z2)
                if merged:
                    Fn.field_forward_raw(spec, named, packed, out_all, stash_all, rays=rays, z=z2, point_offset=R * S, total_points=n_all)
                    if self.seed_hook is not None:   # (and the per-sample field outputs [R S + R G][C] of both passes: diagnostics)
                        self.seed_hook("out_all", out_all)
                else:
                    out2 = self._buf("out2", (R * G, C))
                    stash2 = self._buf("stash2", (Fn.field_stash_bytes(spec, R * G),), torch.uint8)
                    Fn.field_forward_raw(spec, named, packed, out2, stash2, rays=rays, z=z2)
                out2v = out2.view(R, G, C)
                d_all = self._buf("d_all", (n_all, C))
                d1o, d2o = d_all[:R * S].view(R, S, C), d_all[R * S:].view(R, G, C)
        loss = None
        kind_id = self._shade_kind(spec, apply_brdf, cos_irra_on)
        multi = bool(model.MultiBRDF) and apply_brdf
        sun_rows = gsam_only and getattr(model, "sun_v", "none") == "analystic" and apply_brdf and (multi or kind_id == L.BN_SHADE_LAMBERT)
        if (multi and kind_id != L.BN_SHADE_LAMBERT) or sun_rows:
            # One BRDF per sample (or a per-sample irradiance): the BRDF is a pointwise function of a sample's field outputs, so it
            # is evaluated on the rows as they are STORED (pass-1 block, guided block: the merged set is never materialised) by ONE
            # launch (csrc/sample_brdf.hip) that writes a copy of the rows with the padded, irradiance-weighted BRDF value in the
            # place of the albedo; the compositing + loss + composite-backward kernels run on that copy (padding 0: it is inside bp),
            # and ONE launch turns the copy's gradient rows into those of the field outputs (J^T in forward-mode duals, like
            # ray_tail.hip).  Rounds 3-4 went through autograd over the per-point BRDF kernels: ~20 glue launches.
            rgb = self._buf("rgb", (R, 3))
            det = self.repeatable_loss
            ray_loss = self._buf("ray_loss", (R,)) if det else None
            n1 = R * S                                   # rows of the first block (gsam_only: all of them, S = G here)
            n_rows = n1 if gsam_only else n_all
            # (with the sun pass: each row's irradiance is the sun ray's transparency at the row's position in its ray, :265-273 -
            # a constant of the step; the descriptor then carries this step's array)
            bdesc = (shade_desc(model, args, spec, apply_brdf, cos_irra_on, irr=sun_T.reshape(-1)) if sun_rows
                     else self._sample_desc(spec, apply_brdf, cos_irra_on))
            sun_col = 8 if args.data == "sat" else -1
            hs = float(reg.get("hs", 0))
            spv = float(reg.get("nr_spv", 0)) if (spec.normal_an and spec.normal_lr and abs(reg.get("nr_spv", 0)) > 1e-5) else 0.0
            with_reg = hs > 0 or spv != 0.0 or float(reg.get("nr_an", 0)) > 0 or float(reg.get("nr_lr", 0)) > 0
            # no regulariser: a 4-channel copy [bp, sigma] for the one-launch Lambertian tail; with one: a FULL-width copy for the
            # generic compositing kernels, which carry NormalRegLoss / NormalLoss / HardSurfaceLoss on the per-sample normals and weights
            Cb = C if with_reg else 4
            # the row blocks as (rows, samples per ray of the first / second part, first row): one launch over [pass 1 | guided]
            # when they share an array, one per pass otherwise
            if gsam_only:
                blocks = [(out2, n1, S, 0, 0)]
            elif merged:
                blocks = [(out_all, n1, S, G, 0)]
            else:
                blocks = [(out1, n1, S, 0, 0), (out2, R * G, G, 0, n1)]
            with torch.no_grad():
                Bd = self._buf("B_full" if with_reg else "B_4", (n_rows, Cb))
                d_B = self._buf("d_Bfull" if with_reg else "d_B", (n_rows, Cb))
                for Xb, nb1, s1, s2, r0 in blocks:
                    Fn.sample_brdf(bdesc, Xb, rays, nb1, s1, s2, Bd[r0:r0 + Xb.shape[0]], sun_col=sun_col)
                B1 = Bd[:n1].view(R, S, Cb)
                B2 = None if gsam_only else Bd[n1:].view(R, G, Cb)
                dB1 = d_B[:n1].view(R, S, Cb)
                dB2 = None if gsam_only else d_B[n1:].view(R, G, Cb)
                if with_reg:
                    # identity shading of the composited colour: a Lambertian descriptor with no padding and no irradiance (both
                    # are inside bp already), the losses and regularisers as for one BRDF per ray
                    from . import _lib as L_
                    desc = L_.ShadeDesc()
                    desc.kind, desc.C, desc.ch_normal = L.BN_SHADE_LAMBERT, C, -1
                    desc.ch_p0 = desc.ch_p1 = desc.ch_p2 = -1
                    desc.rhoc_is_albedo = desc.shell = desc.cos_irradiance = 0
                    desc.usealldepth = int(bool(self.usealldepth))
                    desc.hpk_scl, desc.f0, desc.rgb_padding = 1.0, 0.04, 0.0
                    desc.lambda_rgb, desc.lambda_ds, desc.lambda_hs = float(self.lambda_rgb), float(self.ds_lambda if use_ds else 0.0), hs
                    nreg = Fn.normal_reg(rays[:, 3:6], spec.ch_normal_an if spec.normal_an else -1, spec.ch_normal_lr if spec.normal_lr else -1,
                                         float(reg.get("nr_an", 0)), float(reg.get("nr_lr", 0)), lambda_spv=spv,
                                         spv_ray=self._buf("spv_ray", (R, 2)) if spv else None, spv_tot=self._buf("spv_tot", (4,)) if spv else None)
                    o = Fn.merged_composite_forward(z_all, idx, B1, B2,
                                                    {k: self._buf("m_" + k, sh) for k, sh in (("depth", (R,)), ("acc", (R, C)),
                                                                                              ("wsum", (R,)), ("var", (R,)), ("reg", (R,)))},
                                                    want=("depth", "acc", "wsum", "var"), nreg=nreg, noise=nz2)
                    sb = {k: self._buf("s_" + k, sh) for k, sh in (("rgb", (R, 3)), ("d_acc", (R, C)), ("d_wsum", (R,)), ("d_depth", (R,)))}
                    gq = Fn.ray_shade_loss(desc, o["acc"], o["wsum"], o["depth"], o["var"], rays[:, 3:6], None, rgbs, sb,
                                           valid_depth if use_ds else None, depths[:, 0] if use_ds else None,
                                           depths[:, 1] if use_ds else None, depth_std if use_ds else None, ray_loss=ray_loss,
                                           loss_acc=None if det else Fn.state_loss_partials(st),
                                           nonfinite=self._nonfinite if self.sanitize_grads else None, extra_loss=o.get("reg"))
                    rgb = gq["rgb"]
                    if spv:
                        Fn.normal_spv_reduce(nreg, R, z_all.shape[1], ray_loss=ray_loss, loss_acc=None if det else Fn.state_loss_partials(st))
                    Fn.merged_composite_backward(z_all, idx, B1, B2, None, gq["d_depth"], gq["d_acc"], dB1, dB2, d_wsum=gq["d_wsum"],
                                                 nonfinite=self._nonfinite if self.sanitize_grads else None,
                                                 hs_scale=hs / R if hs > 0 else 0.0, depth=o["depth"], nreg=nreg, noise=nz2)
                else:
                    Fn.lambert_tail(z_all, idx, B1, B2, rgbs, 0.0, self.lambda_rgb, dB1, dB2,
                                    valid_depth if use_ds else None, depths[:, 0] if use_ds else None, depths[:, 1] if use_ds else None,
                                    depth_std if use_ds else None, self.ds_lambda if use_ds else 0.0, self.usealldepth,
                                    ray_loss=ray_loss, loss_acc=None if det else Fn.state_loss_partials(st), rgb=rgb,
                                    nonfinite=self._nonfinite if self.sanitize_grads else None, noise=nz2)
                if det:
                    loss = ray_loss.sum()
                for Xb, nb1, s1, s2, r0 in blocks:
                    Fn.sample_brdf(bdesc, Xb, rays, nb1, s1, s2, d_all[r0:r0 + Xb.shape[0]], backward_of=d_B[r0:r0 + Xb.shape[0]], sun_col=sun_col)
                if self.sanitize_grads:
                    Fn.count_nonfinite(d_all, self._nonfinite)
                    torch.nan_to_num_(d_all, nan=0.0, posinf=0.0, neginf=0.0)
        elif lambertian:
            rgb = self._buf("rgb", (R, 3))
            det = self.repeatable_loss        # the reported loss too: a fixed-order sum of the per-ray terms instead of atomics
            ray_loss = self._buf("ray_loss", (R,)) if det else None
            with torch.no_grad():
                Fn.lambert_tail(z_all, idx, out1v, out2v, rgbs, model.rgb_padding, self.lambda_rgb, d1o, d2o,
                                valid_depth if use_ds else None, depths[:, 0] if use_ds else None, depths[:, 1] if use_ds else None,
                                depth_std if use_ds else None, self.ds_lambda if use_ds else 0.0, self.usealldepth,
                                ray_loss=ray_loss, loss_acc=None if det else Fn.state_loss_partials(st), rgb=rgb,
                                nonfinite=self._nonfinite if self.sanitize_grads else None, noise=nz2)
                if det:
                    loss = ray_loss.sum()
        else:
            # BRDF / normal models: three launches - composited sums of the merged set, the ray-level shading + losses with
            # their gradients w.r.t. those sums (bn_ray_shade_loss), and the composite backward
            hs = float(reg.get("hs", 0))
            desc = shade_desc(model, args, spec, apply_brdf, cos_irra_on, self.lambda_rgb, self.ds_lambda if use_ds else 0.0,
                              hs, self.usealldepth,
                              irr=sun_irr if (gsam_only and getattr(model, "sun_v", "none") == "analystic" and apply_brdf) else None)
            det = self.repeatable_loss
            ray_loss = self._buf("ray_loss", (R,)) if det else None
            # NormalRegLoss on the per-sample normals (metrics.py:179-216): its value and gradient come from the compositing kernels
            # NormalLoss between the two normal fields (nr_spv_type 1, main.py:297-303): two batch-wide means - the rays' sums come
            # from the forward compositing, a one-workgroup reduce adds them up in fixed order, the backward compositing reads them
            spv = float(reg.get("nr_spv", 0)) if (spec.normal_an and spec.normal_lr and abs(reg.get("nr_spv", 0)) > 1e-5) else 0.0
            nreg = Fn.normal_reg(rays[:, 3:6], spec.ch_normal_an if spec.normal_an else -1, spec.ch_normal_lr if spec.normal_lr else -1,
                                 float(reg.get("nr_an", 0)), float(reg.get("nr_lr", 0)), lambda_spv=spv,
                                 spv_ray=self._buf("spv_ray", (R, 2)) if spv else None, spv_tot=self._buf("spv_tot", (4,)) if spv else None)
            with torch.no_grad():
                o = Fn.merged_composite_forward(z_all, idx, out1v, out2v,
                                                {k: self._buf("m_" + k, sh) for k, sh in (("depth", (R,)), ("acc", (R, C)),
                                                                                          ("wsum", (R,)), ("var", (R,)), ("reg", (R,)))},
                                                want=("depth", "acc", "wsum", "var"), nreg=nreg, noise=nz2)
                sun_d = rays[:, 8:11] if args.data == "sat" else None
                sb = {k: self._buf("s_" + k, sh) for k, sh in (("rgb", (R, 3)), ("d_acc", (R, C)), ("d_wsum", (R,)),
                                                               ("d_depth", (R,)))}
                g = Fn.ray_shade_loss(desc, o["acc"], o["wsum"], o["depth"], o["var"], rays[:, 3:6], sun_d, rgbs, sb,
                                      valid_depth if use_ds else None, depths[:, 0] if use_ds else None,
                                      depths[:, 1] if use_ds else None, depth_std if use_ds else None, ray_loss=ray_loss,
                                      loss_acc=None if det else Fn.state_loss_partials(st),
                                      nonfinite=self._nonfinite if self.sanitize_grads else None, extra_loss=o.get("reg"))
                rgb = g["rgb"]
                if spv:
                    Fn.normal_spv_reduce(nreg, R, z_all.shape[1], ray_loss=ray_loss, loss_acc=None if det else Fn.state_loss_partials(st))
                Fn.merged_composite_backward(z_all, idx, out1v, out2v, None, g["d_depth"], g["d_acc"], d1o, d2o, d_wsum=g["d_wsum"],
                                             nonfinite=self._nonfinite if self.sanitize_grads else None, hs_scale=hs / R if hs > 0 else 0.0,
                                             depth=o["depth"], nreg=nreg, noise=nz2)
                if det:
                    loss = ray_loss.sum()
        if self.seed_hook is not None:       # test hook: sees (and may overwrite) the gradient rows [R (S + G)][C] the field backward starts from
            self.seed_hook("d_all", d_all)
        with torch.no_grad():
            def backward(parts, last):
                """bn_field_backward over the merged set, or over the two passes one after the other; `last`: unfold afterwards."""
                if gsam_only:
                    Fn.field_backward_raw(spec, named, self.grad_views, packed, out2, d_all, stash2, rays=rays, z=z2,
                                          unfold=last, zero_folded=False, parts=parts)
                elif merged:
                    Fn.field_backward_raw(spec, named, self.grad_views, packed, out_all, d_all, stash_all, rays=rays, z=z, z2=z2,
                                          unfold=last, zero_folded=False, parts=parts)
                else:
                    Fn.field_backward_raw(spec, named, self.grad_views, packed, out1, d1o.reshape(R * S, C), stash1, rays=rays, z=z,
                                          unfold=False, parts=parts)
                    Fn.field_backward_raw(spec, named, self.grad_views, packed, out2, d2o.reshape(R * G, C), stash2, rays=rays, z=z2,
                                          unfold=last, zero_folded=False, parts=parts)       # (bn_fold_heads clears the folded accumulators)
            if self.world > 1 and self.overlap_allreduce and 0 < self.n_trunk < self.flat_grad.numel():
                # Two buckets (the reference: DDP's bucketed all-reduce overlapped with backward, main.py:720-731): the trunk's
                # gradient is final after the trunk weight-gradient launch - its all-reduce (RCCL's own stream, ordered behind
                # this one) runs under the head / skinny weight gradients and the unfold launch
                import torch.distributed as dist
                backward(L.BN_BWD_CHAIN | L.BN_BWD_WGRAD_TRUNK, False)
                wa = dist.all_reduce(self.flat_grad[:self.n_trunk], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                backward(L.BN_BWD_WGRAD_HEADS | L.BN_BWD_SKINNY, True)
                wb = dist.all_reduce(self.flat_grad[self.n_trunk:], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
                wa.wait()
                wb.wait()
            else:
                backward(L.BN_BWD_ALL, True)
                if self.world > 1:
                    allreduce_sum_(self.flat_grad, self.pg)
            keep = self.keep_grads       # test hook: leave the gradient in the flat buffer
            Fn.adam_multi(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, [(lo, hi) for _, lo, hi in self.groups],
                          active, st, self.betas, self.eps, self.wd, 1.0 / self.world, zero_grad=not keep)
        return loss, rgb

    def _adam(self, apply_brdf, apply_theta=False):
        scale = 1.0 / self.world          # DDP averages gradients
        on = {"base": True, "brdf": bool(apply_brdf), "theta": bool(apply_brdf and apply_theta)}
        for gname, lo, hi in self.groups:
            if not on[gname]:
                continue                  # not in this step's graph: torch.optim.Adam would see grad None and skip it
            self.adam_steps[gname] += 1
            Fn.adam_step(self.flat_param[lo:hi], self.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                         self.adam_steps[gname], self.lr, self.betas, self.eps, self.wd, scale)
