"""torch.autograd.Function wrappers over the C ABI (include/brdfnerf_hip.h).

PyTorch is plumbing here: it owns device memory and the stream and stitches the autograd graph;
every numerical step of the hot path runs in the HIP library.  No CPU fallback exists.
"""
import ctypes as C
import os

import torch

from . import _lib as L


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "C ABI needs contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def _f32(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


# ----------------------------------------------------------------------------------------- field MLP
class FieldSpec:
    """Host-side description of one evaluation of the field: which heads, which dtype."""

    def __init__(self, feat, layers, skip, pe_freqs, act, dtype, heads, normal_lr, normal_an=False, fold_feats=None,
                 dir_dim=0, dir_freqs=0, t_dim=0):
        # heads: list of (name, n_out, kind); heads[0] must be ("rgb_from_xyzdir", 3, PLAIN)
        self.feat, self.layers, self.skip, self.pe_freqs, self.act, self.dtype = feat, layers, skip, pe_freqs, act, dtype
        self.heads, self.normal_lr, self.normal_an = list(heads), bool(normal_lr), bool(normal_an)
        # feats_from_xyz is linear and feeds only the heads' first (linear) layers: W1 (Wf y + bf) + b1 = (W1 Wf) y + (W1 bf + b1).
        # Folding the two (one small GEMM per head per weight update, fold()) removes an F x F product per point from the
        # forward, the backward chain and the weight-gradient kernels; unfold_grads() takes the folded gradients back to
        # W1, b1, Wf, bf by the chain rule.  BRDFNERF_FOLD_FEATS=0 keeps the layer-by-layer evaluation.
        if fold_feats is None:
            fold_feats = os.environ.get("BRDFNERF_FOLD_FEATS", "1") != "0"
        self.fold_feats = bool(fold_feats)
        # --input_viewdir: the rgb head's first layer also reads the (encoded) view direction, dir_dim extra input columns
        self.dir_dim, self.dir_freqs = int(dir_dim), int(dir_freqs)
        if self.dir_dim and not self.fold_feats:
            raise NotImplementedError("--input_viewdir needs the folded feats layer (BRDFNERF_FOLD_FEATS=1, the default)")
        # --beta: head 1 (kind BN_HEAD_BETA) also reads the per-image embedding, t_dim extra input columns
        self.t_dim = int(t_dim)
        if (self.t_dim > 0) != (len(self.heads) > 1 and self.heads[1][2] == L.BN_HEAD_BETA):
            raise ValueError("t_dim goes with a beta head at index 1")
        if self.t_dim and not self.fold_feats:
            raise NotImplementedError("--beta needs the folded feats layer (BRDFNERF_FOLD_FEATS=1, the default)")
        self.folded, self.fold_grads = {}, {}
        d = L.FieldDesc()
        d.feat, d.layers, d.skip, d.pe_freqs, d.act, d.dtype = feat, layers, skip, pe_freqs, act, dtype
        d.fold_feats = int(self.fold_feats)
        d.dir_dim, d.dir_freqs, d.t_dim = self.dir_dim, self.dir_freqs, self.t_dim
        d.n_heads = len(self.heads)
        c0 = 5 if self.t_dim else 4                     # [rgb3, sigma, (beta)]
        c = c0 + (3 if normal_an else 0) + (3 if normal_lr else 0)
        self.head_cols = []
        for i, (_, n_out, kind) in enumerate(self.heads):
            d.head_out[i], d.head_kind[i] = n_out, kind
            if i == 0:
                self.head_cols.append((0, 3))
            elif kind == L.BN_HEAD_BETA:
                self.head_cols.append((4, 1))
            else:
                w = n_out if kind in (L.BN_HEAD_PLAIN, L.BN_HEAD_HAPKE_THETA) else 3
                self.head_cols.append((c, w))
                c += w
        d.normal_lr, d.normal_an, d.out_channels = int(normal_lr), int(normal_an), c
        self.desc, self.out_channels = d, c
        self.ch_beta = 4 if self.t_dim else -1
        self.ch_normal_an = c0 if normal_an else -1
        self.ch_normal_lr = (c0 + 3 if normal_an else c0) if normal_lr else -1
        self.packed_bytes = L.lib().bn_field_packed_bytes(C.byref(d))
        if self.packed_bytes == 0:
            raise RuntimeError("bn_field_packed_bytes: " + L.lib().bn_last_error().decode())

    def key(self):
        return (self.feat, self.layers, self.skip, self.pe_freqs, self.act, self.dtype, tuple(self.heads), self.normal_lr,
                self.normal_an, self.fold_feats, self.dir_dim, self.dir_freqs, self.t_dim)

    def params_struct(self, named, grads=False):
        """named: dict state_dict-key -> tensor (parameters, or same-shaped gradient buffers)."""
        s = L.FieldGrads() if grads else L.FieldParams()
        for l in range(self.layers):
            s.trunk_w[l] = named[f"fc_net.{2*l}.weight"].data_ptr()
            s.trunk_b[l] = named[f"fc_net.{2*l}.bias"].data_ptr()
        s.sigma_w, s.sigma_b = named["sigma_from_xyz.0.weight"].data_ptr(), named["sigma_from_xyz.0.bias"].data_ptr()
        s.feats_w, s.feats_b = named["feats_from_xyz.weight"].data_ptr(), named["feats_from_xyz.bias"].data_ptr()
        for i, (name, _, _) in enumerate(self.heads):
            s.head_w1[i], s.head_b1[i] = named[f"{name}.0.weight"].data_ptr(), named[f"{name}.0.bias"].data_ptr()
            s.head_w2[i], s.head_b2[i] = named[f"{name}.2.weight"].data_ptr(), named[f"{name}.2.bias"].data_ptr()
        if self.fold_feats:
            if grads:        # the library accumulates d/d(folded w1, b1) here; feats_* are produced by unfold_grads()
                s.feats_w = s.feats_b = None
                for i, (name, _, _) in enumerate(self.heads):
                    w1 = named[f"{name}.0.weight"]
                    ent = self.fold_grads.get(name)
                    if ent is None or ent[0].device != w1.device:
                        ent = (torch.zeros(w1.shape[0], self.feat, dtype=torch.float32, device=w1.device),
                               torch.zeros(w1.shape[0], dtype=torch.float32, device=w1.device))
                        self.fold_grads[name] = ent
                    s.head_w1[i], s.head_b1[i] = ent[0].data_ptr(), ent[1].data_ptr()
            else:
                if any(name not in self.folded for name, _, _ in self.heads):
                    self.fold(named)     # a spec evaluated with another spec's packed weights (sigma-only passes)
                for i, (name, _, _) in enumerate(self.heads):
                    s.head_w1[i], s.head_b1[i] = self.folded[name][0].data_ptr(), self.folded[name][1].data_ptr()
        if self.normal_lr:
            s.normal_w, s.normal_b = named["grad_from_xyz.weight"].data_ptr(), named["grad_from_xyz.bias"].data_ptr()
        if self.dir_dim:        # direction columns of the rgb head's first layer (parameters, or their gradient buffer)
            w = named[f"{self.heads[0][0]}.0.weight"]
            assert w.shape[1] == self.feat + self.dir_dim and w.is_contiguous()
            s.head0_wdir, s.head0_wdir_ld = w.data_ptr() + 4 * self.feat, w.shape[1]
        if self.t_dim:          # embedding columns of the beta head's first layer
            w = named[f"{self.heads[1][0]}.0.weight"]
            assert w.shape[1] == self.feat + self.t_dim and w.is_contiguous()
            s.head1_wt, s.head1_wt_ld = w.data_ptr() + 4 * self.feat, w.shape[1]
        return s

    def _fold_desc(self, named, named_grads=None, with_accumulators=True):
        """bn_fold_desc over this spec's heads (functions.fold / unfold_grads)."""
        d = L.FoldDesc()
        wf, bf = named["feats_from_xyz.weight"], named["feats_from_xyz.bias"]
        assert wf.is_contiguous() and wf.dtype == torch.float32
        d.n_heads, d.F = len(self.heads), self.feat
        d.wf, d.bf = wf.data_ptr(), bf.data_ptr()
        keep = [wf, bf]
        for i, (name, _, _) in enumerate(self.heads):
            w1, b1 = named[f"{name}.0.weight"], named[f"{name}.0.bias"]
            assert w1.is_contiguous() and w1.dtype == torch.float32
            d.rows = w1.shape[0]
            d.w1[i], d.w1_ld[i], d.b1[i] = w1.data_ptr(), w1.shape[1], b1.data_ptr()
            ent = self.folded.get(name)
            if ent is None or ent[0].device != w1.device:
                ent = (torch.empty(w1.shape[0], self.feat, dtype=torch.float32, device=w1.device), torch.empty_like(b1))
                self.folded[name] = ent
            d.w_fold[i], d.b_fold[i] = ent[0].data_ptr(), ent[1].data_ptr()
            acc = self.fold_grads.get(name) if with_accumulators else None
            d.m[i], d.s[i] = (acc[0].data_ptr(), acc[1].data_ptr()) if acc is not None else (None, None)
            if named_grads is not None:
                dw1, db1 = named_grads[f"{name}.0.weight"], named_grads[f"{name}.0.bias"]
                assert dw1.stride(-1) == 1
                d.d_w1[i], d.d_w1_ld[i], d.d_b1[i] = dw1.data_ptr(), dw1.stride(0), db1.data_ptr()
        if named_grads is not None:
            d.d_wf, d.d_bf = named_grads["feats_from_xyz.weight"].data_ptr(), named_grads["feats_from_xyz.bias"].data_ptr()
        return d, keep

    @torch.no_grad()
    def fold(self, named):
        """(W1 Wf, W1 bf + b1) per head, refreshed whenever the weights are re-packed (bn_fold_heads: one launch; it also
        clears the folded-gradient accumulators, which the weight-gradient kernels add to)."""
        named = {k: v.detach() for k, v in named.items()}
        d, _keep = self._fold_desc(named)
        L.check(L.lib().bn_fold_heads(C.byref(d), _stream()), "bn_fold_heads")

    @torch.no_grad()
    def unfold_grads(self, named, named_grads, zero=True):
        """Chain rule from the folded first layers back to the three factors (accumulating, like the library):
        dW1 += M Wf^T + s bf^T, db1 += s, dWf += W1^T M, dbf += W1^T s, with M = dL/d(W1 Wf), s = dL/d(W1 bf + b1)
        (bn_unfold_heads: one launch).  zero: clear M, s afterwards (callers that re-fold before the next backward - the fused
        step - skip it: bn_fold_heads clears them)."""
        if not self.fold_grads:
            return
        named = {k: v.detach() for k, v in named.items()}
        d, _keep = self._fold_desc(named, named_grads)
        L.check(L.lib().bn_unfold_heads(C.byref(d), _stream()), "bn_unfold_heads")
        if zero:
            for m, sv in self.fold_grads.values():
                m.zero_()
                sv.zero_()

    def used_param_names(self):
        names = []
        for l in range(self.layers):
            names += [f"fc_net.{2*l}.weight", f"fc_net.{2*l}.bias"]
        names += ["sigma_from_xyz.0.weight", "sigma_from_xyz.0.bias", "feats_from_xyz.weight", "feats_from_xyz.bias"]
        for name, _, _ in self.heads:
            names += [f"{name}.0.weight", f"{name}.0.bias", f"{name}.2.weight", f"{name}.2.bias"]
        if self.normal_lr:
            names += ["grad_from_xyz.weight", "grad_from_xyz.bias"]
        return names


def make_points(xyz=None, rays=None, z=None, dirs=None, t_embed=None, point_offset=0, total_points=0, z2=None):
    """point_offset / total_points: this call's points inside a larger set that shares one output array and one stash;
    z2 (R, G): the backward over a set of TWO sample blocks of the same rays (bn_points)."""
    pts = L.Points()
    pts.dirs = pts.t_embed = pts.z2 = None
    pts.point_offset, pts.total_points, pts.n_samples2, pts.seg1_points = int(point_offset), int(total_points), 0, 0
    if t_embed is not None:      # per point with xyz, per ray with rays
        rows = xyz.shape[0] if xyz is not None else rays.shape[0]
        assert t_embed.shape[0] == rows and t_embed.is_contiguous() and t_embed.dtype == torch.float32
        pts.t_embed = t_embed.data_ptr()
    if xyz is not None:
        pts.xyz, pts.rays, pts.z = xyz.data_ptr(), None, None
        pts.ray_stride, pts.n_samples, pts.n_points = 0, 0, xyz.shape[0]
        if dirs is not None:
            assert dirs.shape == xyz.shape and dirs.is_contiguous() and dirs.dtype == torch.float32
            pts.dirs = dirs.data_ptr()
    else:
        pts.xyz, pts.rays, pts.z = None, rays.data_ptr(), z.data_ptr()
        pts.ray_stride, pts.n_samples, pts.n_points = rays.shape[1], z.shape[1], z.shape[0] * z.shape[1]
        if z2 is not None:
            assert z2.shape[0] == z.shape[0] and z2.is_contiguous() and z2.dtype == torch.float32
            pts.z2, pts.n_samples2, pts.seg1_points = z2.data_ptr(), z2.shape[1], pts.n_points
            pts.n_points = pts.n_points + z2.shape[0] * z2.shape[1]
    return pts


def pack_field(spec, named_params, packed=None):
    dev = named_params["fc_net.0.weight"].device
    if packed is None:
        packed = torch.empty(spec.packed_bytes, dtype=torch.uint8, device=dev)
    if spec.fold_feats:
        spec.fold(named_params)
    ps = spec.params_struct(named_params)
    L.check(L.lib().bn_pack_field(C.byref(spec.desc), C.byref(ps), _p(packed), _stream()), "bn_pack_field")
    return packed


def field_sigma(spec, named_params, packed, xyz=None, rays=None, z=None, out=None):
    """sigma-only forward, no autograd (pass 1 / sun pass).  out: a caller-owned [n_points] buffer (captured steps)."""
    pts = make_points(xyz, rays, z)
    ref = xyz if xyz is not None else z
    sigma = out if out is not None else torch.empty(pts.n_points, dtype=torch.float32, device=ref.device)
    ps = spec.params_struct(named_params)
    L.check(L.lib().bn_field_sigma(C.byref(spec.desc), C.byref(ps), _p(packed), C.byref(pts), _p(sigma), _stream()),
            "bn_field_sigma")
    return sigma


class FieldFunction(torch.autograd.Function):
    """out[n_points][C] = field(points; params).  Differentiable w.r.t. the parameters and the --beta embedding input
    `t_embed` only (the reference never needs d/d xyz outside the analytic-normal path: z_vals are detached,
    rendering.py:262)."""

    @staticmethod
    def forward(ctx, spec, packed, xyz, rays, z, t_embed, names_grad, *params):
        names, grad_enabled, dirs = names_grad      # grad mode is always off inside Function.forward: passed in
        named = dict(zip(names, params))
        if t_embed is not None:
            t_embed = t_embed.detach().float().contiguous()
        pts = make_points(xyz, rays, z, dirs, t_embed)
        ref = xyz if xyz is not None else z
        out = torch.empty(pts.n_points, spec.out_channels, dtype=torch.float32, device=ref.device)
        ctx.t_grad = bool(grad_enabled and t_embed is not None and ctx.needs_input_grad[5])
        need_grad = grad_enabled and (any(p.requires_grad for p in params) or ctx.t_grad)
        stash = None
        if need_grad or spec.normal_an:
            nbytes = L.lib().bn_field_stash_bytes(C.byref(spec.desc), pts.n_points)
            stash = torch.empty(nbytes, dtype=torch.uint8, device=ref.device)
        ps = spec.params_struct(named)
        L.check(L.lib().bn_field_forward(C.byref(spec.desc), C.byref(ps), _p(packed), C.byref(pts), _p(out), _p(stash),
                                         _stream()), "bn_field_forward")
        if spec.normal_an:
            L.check(L.lib().bn_field_normals(C.byref(spec.desc), C.byref(ps), _p(packed), C.byref(pts), _p(stash), _p(out), None,
                                             1 if need_grad else 0, _stream()), "bn_field_normals")
            if not need_grad:
                stash = None
        ctx.spec, ctx.packed, ctx.names, ctx.stash = spec, packed, names, stash
        ctx.pts_t = (xyz, rays, z, dirs, t_embed)
        ctx.save_for_backward(out, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        out, *params = ctx.saved_tensors
        spec, names = ctx.spec, ctx.names
        if ctx.stash is None:
            raise RuntimeError("FieldFunction.backward without a stash (forward ran under no_grad)")
        named = dict(zip(names, params))
        sizes = [p.numel() for p in params]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 3) // 4 * 4
        flat = torch.zeros(tot, dtype=torch.float32, device=out.device)
        grads = [flat[o:o + n].view(p.shape) for o, n, p in zip(offs, sizes, params)]
        named_grads = dict(zip(names, grads))
        gs = spec.params_struct(named_grads, grads=True)
        ps = spec.params_struct(named)
        xyz, rays, z, dirs, t_embed = ctx.pts_t
        pts = make_points(xyz, rays, z, dirs, t_embed)
        d_t = None
        if ctx.t_grad:          # per point; the rays form's per-ray gradient is the sum over the ray's samples
            d_t = torch.empty(pts.n_points, spec.t_dim, dtype=torch.float32, device=out.device)
            gs.d_t_embed = d_t.data_ptr()
            if rays is not None:
                d_t = d_t.view(rays.shape[0], -1, spec.t_dim)
        d_out = _f32(d_out)
        L.check(L.lib().bn_field_backward(C.byref(spec.desc), C.byref(ps), _p(ctx.packed), C.byref(pts), _p(out),
                                          _p(d_out), _p(ctx.stash), C.byref(gs), _stream()), "bn_field_backward")
        if spec.fold_feats:
            spec.unfold_grads(named, named_grads)
        ctx.stash = None
        if d_t is not None and d_t.dim() == 3:
            d_t = d_t.sum(1)
        return (None, None, None, None, None, d_t, None) + tuple(g if p.requires_grad else None
                                                                 for g, p in zip(grads, params))


# ----------------------------------------------------------------------------------------- compositing
class CompositeFunction(torch.autograd.Function):
    """(z, out[R,S,C] with sigma in channel 3, noise) -> alphas, transparency, weights, depth, acc[R,C]."""

    @staticmethod
    def forward(ctx, z, out, noise, noise_std):
        R, S = z.shape
        Cc = out.shape[-1] if out.dim() == 3 else 1
        dev = z.device
        alphas = torch.empty(R, S, dtype=torch.float32, device=dev)
        trans = torch.empty_like(alphas)
        weights = torch.empty_like(alphas)
        depth = torch.empty(R, dtype=torch.float32, device=dev)
        sigma_only = out.dim() == 2
        acc = None if sigma_only else torch.empty(R, Cc, dtype=torch.float32, device=dev)
        base = out.data_ptr()
        sig_ptr = C.c_void_p(base if sigma_only else base + 3 * 4)
        L.check(L.lib().bn_composite_forward(_p(z), sig_ptr, 1 if sigma_only else Cc, _p(noise), float(noise_std),
                                             None if sigma_only else _p(out), Cc, 0 if sigma_only else Cc, R, S,
                                             _p(alphas), _p(trans), _p(weights), _p(depth), _p(acc), _stream()),
                "bn_composite_forward")
        ctx.save_for_backward(z, out, noise)
        ctx.noise_std, ctx.sigma_only = float(noise_std), sigma_only
        ctx.mark_non_differentiable(alphas, trans)
        if sigma_only:
            return alphas, trans, weights, depth
        return alphas, trans, weights, depth, acc

    @staticmethod
    def backward(ctx, d_alphas, d_trans, d_weights, d_depth, d_acc=None):
        z, out, noise = ctx.saved_tensors
        R, S = z.shape
        sigma_only = ctx.sigma_only
        Cc = 1 if sigma_only else out.shape[-1]
        d_out = torch.zeros_like(out)
        base = out.data_ptr()
        dbase = d_out.data_ptr()
        sig_ptr = C.c_void_p(base if sigma_only else base + 12)
        dsig_ptr = C.c_void_p(dbase if sigma_only else dbase + 12)
        dw = None if d_weights is None else _f32(d_weights)
        dd = None if d_depth is None else _f32(d_depth)
        da = None if (d_acc is None or sigma_only) else _f32(d_acc)
        L.check(L.lib().bn_composite_backward(_p(z), sig_ptr, 1 if sigma_only else Cc, _p(noise), ctx.noise_std,
                                              None if sigma_only else _p(out), Cc, 0 if sigma_only else Cc, R, S,
                                              _p(dw), _p(dd), _p(da), dsig_ptr, 1 if sigma_only else Cc,
                                              None if sigma_only else _p(d_out), Cc, _stream()),
                "bn_composite_backward")
        return None, d_out, None, None


def composite(z, out, noise=None, noise_std=0.0):
    return CompositeFunction.apply(z.contiguous(), out.contiguous(), noise, noise_std)


def composite_forward_raw(z, out, noise=None, noise_std=0.0, bufs=None):
    """No-autograd compositing of out[R,S,C] (sigma = channel 3): -> alphas, trans, weights, depth, acc.
    `bufs` (optional dict) supplies preallocated outputs so a training loop does not hit the allocator."""
    R, S = z.shape
    Cc = out.shape[-1]
    dev = z.device
    b = bufs if bufs is not None else {}
    mk = lambda k, shape: b.get(k) if k in b else torch.empty(shape, dtype=torch.float32, device=dev)
    alphas, trans, weights = mk("alphas", (R, S)), mk("trans", (R, S)), mk("weights", (R, S))
    depth, acc = mk("depth", (R,)), mk("acc", (R, Cc))
    L.check(L.lib().bn_composite_forward(_p(z), C.c_void_p(out.data_ptr() + 12), Cc, _p(noise), float(noise_std), _p(out), Cc,
                                         Cc, R, S, _p(alphas), _p(trans), _p(weights), _p(depth), _p(acc), _stream()),
            "bn_composite_forward")
    return alphas, trans, weights, depth, acc


def composite_backward_raw(z, out, d_weights, d_depth, d_acc, noise=None, noise_std=0.0, d_out=None):
    """-> d_out[R,S,C].  Channel 3 of d_out receives d sigma; d_acc[:, 3] must be zero (acc[:, 3] is not an output)."""
    R, S = z.shape
    Cc = out.shape[-1]
    if d_out is None:
        d_out = torch.empty_like(out)
    L.check(L.lib().bn_composite_backward(_p(z), C.c_void_p(out.data_ptr() + 12), Cc, _p(noise), float(noise_std), _p(out), Cc,
                                          Cc, R, S, _p(d_weights), _p(d_depth), _p(d_acc),
                                          C.c_void_p(d_out.data_ptr() + 12), Cc, _p(d_out), Cc, _stream()),
            "bn_composite_backward")
    return d_out


def field_forward_raw(spec, named_params, packed, out, stash, xyz=None, rays=None, z=None, point_offset=0, total_points=0):
    """out / stash are the whole SET's when total_points > 0 (this call fills rows [point_offset, point_offset + n_points))."""
    pts = make_points(xyz, rays, z, point_offset=point_offset, total_points=total_points)
    ps = spec.params_struct(named_params)
    L.check(L.lib().bn_field_forward(C.byref(spec.desc), C.byref(ps), _p(packed), C.byref(pts), _p(out), _p(stash), _stream()),
            "bn_field_forward")
    if spec.normal_an:
        L.check(L.lib().bn_field_normals(C.byref(spec.desc), C.byref(ps), _p(packed), C.byref(pts), _p(stash), _p(out), None, 1,
                                         _stream()), "bn_field_normals")
    return out


def field_backward_raw(spec, named_params, named_grads, packed, out, d_out, stash, xyz=None, rays=None, z=None, unfold=True,
                       zero_folded=True, parts=L.BN_BWD_ALL, z2=None):
    """unfold=False leaves the folded first-layer gradients in spec.fold_grads (they keep accumulating): a caller that
    back-propagates several batches before the optimizer step unfolds once, on the last call.  parts: bn_field_backward_parts
    (a caller that overlaps the all-reduce of the trunk's gradient with the rest of the backward).  z2: the points are the two
    sample blocks [z | z2] of the same rays evaluated into one output / stash (field_forward_raw(point_offset=...))."""
    pts = make_points(xyz, rays, z, z2=z2)
    ps, gs = spec.params_struct(named_params), spec.params_struct(named_grads, grads=True)
    L.check(L.lib().bn_field_backward_parts(C.byref(spec.desc), C.byref(ps), _p(packed), C.byref(pts), _p(out), _p(d_out), _p(stash),
                                            C.byref(gs), int(parts), _stream()), "bn_field_backward")
    if spec.fold_feats and unfold:
        spec.unfold_grads(named_params, named_grads, zero=zero_folded)


def field_stash_bytes(spec, n_points):
    return L.lib().bn_field_stash_bytes(C.byref(spec.desc), int(n_points))


def lambert_loss(acc, weights, z, depth, rgbs, rgb_padding, lambda_rgb, valid_depth=None, target_depth=None,
                 target_weight=None, target_std=None, lambda_ds=0.0, usealldepth=False):
    """Shading + SNerfLoss + DepthLoss of a Lambertian step and their gradients in one launch (bn_lambert_loss).
    Returns (loss 0-d, rgb (R,3), d_acc (R,C), d_depth (R), d_weights (R,S))."""
    R, S = weights.shape
    Cc = acc.shape[1]
    dev = acc.device
    ray_loss = torch.empty(R, dtype=torch.float32, device=dev)
    rgb = torch.empty(R, 3, dtype=torch.float32, device=dev)
    d_acc = torch.empty(R, Cc, dtype=torch.float32, device=dev)
    d_depth = torch.empty(R, dtype=torch.float32, device=dev)
    d_weights = torch.empty(R, S, dtype=torch.float32, device=dev)
    use = target_depth is not None and lambda_ds > 0
    f = lambda t: _f32(t).contiguous() if (use and t is not None) else None
    # every converted / compacted input stays referenced until the launch has been issued (a temporary freed earlier could
    # be handed to the next conversion by the caching allocator)
    acc, weights, z, depth, rgbs = _f32(acc), _f32(weights), _f32(z), _f32(depth), _f32(rgbs)
    vd, td, tw, ts = f(valid_depth), f(target_depth), f(target_weight), f(target_std)
    L.check(L.lib().bn_lambert_loss(_p(acc), Cc, _p(weights), _p(z), S, _p(depth), _p(rgbs), _p(vd), _p(td), _p(tw), _p(ts),
                                    float(rgb_padding), float(lambda_rgb), float(lambda_ds), int(bool(usealldepth)), R,
                                    _p(ray_loss), _p(rgb), _p(d_acc), _p(d_depth), _p(d_weights), _stream()), "bn_lambert_loss")
    return ray_loss.sum(), rgb, d_acc, d_depth, d_weights


# ----------------------------------------------------------------------------------------- sampling
def stratified_z(near, far, u):
    """near, far: (R,1) (any stride along R), u: (R,S) -> z (R,S)."""
    R, S = u.shape
    near, far, u = _f32(near).reshape(R), _f32(far).reshape(R), _f32(u)
    z = torch.empty(R, S, dtype=torch.float32, device=u.device)
    L.check(L.lib().bn_stratified_z(_p(near), _p(far), 1, _p(u), R, S, _p(z), _stream()), "bn_stratified_z")
    return z


def guided_samples(z, weights, depth, u, near0, far0, d_range, use_target=None, target_depth=None, target_std=None,
                   u_target=None, target_row=None, merge=True):
    """near0 may be a device tensor holding (near0, far0) as two consecutive floats - e.g. rays[0, 6:8] - with far0 = None:
    the kernel then reads the clamp window itself (no device->host read)."""
    R, S = z.shape
    G = u.shape[1]
    dev = z.device
    z2 = torch.empty(R, G, dtype=torch.float32, device=dev)
    z_all = torch.empty(R, S + G, dtype=torch.float32, device=dev) if merge else None
    idx = torch.empty(R, S + G, dtype=torch.int64, device=dev) if merge else None
    if torch.is_tensor(near0):
        assert far0 is None and near0.is_cuda and near0.dtype == torch.float32 and near0.numel() == 2 and near0.is_contiguous()
        L.check(L.lib().bn_guided_samples_nf(_p(z), _p(weights), _p(depth), _p(u), R, S, G, C.c_void_p(near0.data_ptr()),
                                             float(d_range), _p(use_target), _p(target_depth), _p(target_std), _p(u_target),
                                             _p(target_row), _p(z2), _p(z_all), _p(idx), _stream()), "bn_guided_samples_nf")
        return z2, z_all, idx
    L.check(L.lib().bn_guided_samples(_p(z), _p(weights), _p(depth), _p(u), R, S, G, float(near0), float(far0),
                                      float(d_range), _p(use_target), _p(target_depth), _p(target_std), _p(u_target),
                                      _p(target_row), _p(z2), _p(z_all), _p(idx), _stream()), "bn_guided_samples")
    return z2, z_all, idx


# ----------------------------------------------------------------------------------------- launch-lean fused step
def new_step_state(device, seed, lr):
    """Device-resident bn_step_state (include/brdfnerf_hip.h): draw seed / counter, learning rate, Adam step counts, loss ring."""
    st = torch.zeros(L.BN_STATE_BYTES // 8, dtype=torch.int64, device=device)
    st[0] = int(seed) & 0x7FFFFFFFFFFFFFFF
    st.view(torch.float32)[4] = float(lr)
    st.view(torch.float64)[L.BN_STATE_POW_OFF // 8:L.BN_STATE_POW_OFF // 8 + 8] = 1.0          # beta^0
    return st


def set_adam_steps(state, steps, betas=(0.9, 0.999)):
    """Write the per-group optimiser step counts (and the beta powers that go with them) into a step state."""
    steps = list(steps) + [0] * (4 - len(steps))
    state.view(torch.int32)[6:10] = torch.tensor(steps, dtype=torch.int32)
    pw = [float(betas[0]) ** s for s in steps] + [float(betas[1]) ** s for s in steps]
    state.view(torch.float64)[L.BN_STATE_POW_OFF // 8:L.BN_STATE_POW_OFF // 8 + 8] = torch.tensor(pw, dtype=torch.float64)


def state_views(state):
    """(rng_step 0-d int64, lr 0-d float32, adam_step int32[4], loss_ring float32[64]) views of a step state."""
    f, i = state.view(torch.float32), state.view(torch.int32)
    return state[1], f[4], i[6:10], f[L.BN_STATE_LOSS_OFF // 4:L.BN_STATE_LOSS_OFF // 4 + L.BN_STATE_LOSS_SLOTS]


def set_state_noise(state, noise_std):
    """--noise_std of the running step into the device step state (read by the compositing kernels when bn_noise.noise_std < 0)."""
    state.view(torch.float32)[L.BN_STATE_NOISE_OFF // 4].fill_(float(noise_std))


def state_loss_partials(state):
    """The 64 partial sums bn_lambert_tail adds the running step's loss terms to (bn_adam_multi folds them into the ring)."""
    return state.view(torch.float32)[L.BN_STATE_PART_OFF // 4:L.BN_STATE_PART_OFF // 4 + L.BN_STATE_LOSS_SLOTS]


def stratified_z_rng(rays, S, state, z=None, ray_offset=0, near_far=None, stream_id=None):
    """Stratified depths from rays[:, 6], rays[:, 7] - or from near_far [R][2] - with in-kernel uniforms (stream BN_RNG_COARSE of
    `state`, or stream_id)."""
    src, col = (rays, 6) if near_far is None else (near_far, 0)
    R = src.shape[0]
    assert src.is_contiguous() and src.dtype == torch.float32 and src.shape[1] >= col + 2
    if z is None:
        z = torch.empty(R, S, dtype=torch.float32, device=src.device)
    base = src.data_ptr() + 4 * col
    L.check(L.lib().bn_stratified_z_rng(C.c_void_p(base), C.c_void_p(base + 4), src.shape[1], _p(state),
                                        L.BN_RNG_COARSE if stream_id is None else int(stream_id),
                                        int(ray_offset), R, S, _p(z), _stream()), "bn_stratified_z_rng")
    return z


def rng_uniform(state, stream_id, n):
    u = torch.empty(n, dtype=torch.float32, device=state.device)
    L.check(L.lib().bn_rng_uniform(_p(state), int(stream_id), n, _p(u), _stream()), "bn_rng_uniform")
    return u


def rng_normal(state, stream_id, n):
    x = torch.empty(n, dtype=torch.float32, device=state.device)
    L.check(L.lib().bn_rng_normal(_p(state), int(stream_id), n, _p(x), _stream()), "bn_rng_normal")
    return x


def noise_arg(state, noise_std, stream_id, ray_offset=0, from_state=False):
    """bn_noise: --noise_std with in-kernel draws (stream `stream_id` of the step state); None when noise_std == 0.
    from_state: the kernels read the value from the step state (set_state_noise) instead of the launch arguments."""
    if not noise_std:
        return None
    n = L.Noise()
    n.rng, n.noise_std, n.rng_stream, n.ray_offset = state.data_ptr(), -1.0 if from_state else float(noise_std), int(stream_id), int(ray_offset)
    n._keep = state
    return n


def _nz(noise):
    return None if noise is None else C.byref(noise)


def _strided(t):
    """(pointer, element stride) of a 1-d float32 view (e.g. depths[:, 0]); None -> (None, 1)."""
    if t is None:
        return None, 1
    assert t.dtype == torch.float32 and t.dim() == 1 and t.is_cuda
    return C.c_void_p(t.data_ptr()), t.stride(0) if t.shape[0] > 1 else 1


def composite_guided(z, out1, G, near_far, d_range, use_target=None, target_depth=None, target_std=None, u=None, u_target=None,
                     state=None, bufs=None, want_pass1=False, ray_offset=0, sigma=None, noise=None):
    """Pass-1 compositing of out1 [R][S][C] (sigma = channel 3; or `sigma` [R][S] from a sigma-only pass 1, out1 = None) +
    depth-guided resampling + merge in one launch.
    Draws: arrays u / u_target [R][G], or the in-kernel streams of `state`.  -> z2 [R][G], z_all [R][S+G], sort_idx [R][S+G]
    (+ pass-1 weights, depth when want_pass1)."""
    R, S = z.shape
    Cc = out1.shape[-1] if sigma is None else 1
    sig_ptr = C.c_void_p(out1.data_ptr() + 12) if sigma is None else _p(sigma)
    dev = z.device
    b = bufs if bufs is not None else {}
    mk = lambda k, shape, dt=torch.float32: b[k] if k in b else torch.empty(shape, dtype=dt, device=dev)
    z2, z_all, idx = mk("z2", (R, G)), mk("z_all", (R, S + G)), mk("idx", (R, S + G), torch.int64)
    w1 = mk("w1", (R, S)) if want_pass1 else None
    d1 = mk("d1", (R,)) if want_pass1 else None
    ut_p, ut_s = _strided(use_target)
    td_p, td_s = _strided(target_depth)
    ts_p, ts_s = _strided(target_std)
    L.check(L.lib().bn_composite_guided(_p(z), sig_ptr, Cc, R, S, G, C.c_void_p(near_far.data_ptr()),
                                        float(d_range), ut_p, ut_s, td_p, td_s, ts_p, ts_s, _p(u), _p(u_target), _p(state),
                                        L.BN_RNG_GUIDED, L.BN_RNG_GUIDED_TARGET, int(ray_offset), _p(z2), _p(z_all), _p(idx), _p(w1), _p(d1),
                                        _nz(noise), _stream()), "bn_composite_guided")
    return (z2, z_all, idx, w1, d1) if want_pass1 else (z2, z_all, idx)


def normal_reg(rays_d, ch_an, ch_lr, lambda_an, lambda_lr, lambda_spv=0.0, spv_ray=None, spv_tot=None):
    """bn_normal_reg: NormalRegLoss (metrics.py:179-216) inside the merged-set compositing; rays_d (R,3) view with unit inner
    stride - and / or NormalLoss 'an_lr' (metrics.py:218-261) between the two normal fields at channels ch_an / ch_lr (lambda_spv
    != 0, with spv_ray [R][2] and spv_tot [4]: see normal_spv_reduce).  None when neither is on."""
    on_an, on_lr = ch_an >= 0 and lambda_an > 0, ch_lr >= 0 and lambda_lr > 0
    on_spv = bool(lambda_spv) and ch_an >= 0 and ch_lr >= 0
    if not (on_an or on_lr or on_spv):
        return None
    nr = L.NormalReg()
    if on_an or on_lr:
        assert rays_d.is_cuda and rays_d.dtype == torch.float32 and rays_d.dim() == 2 and rays_d.stride(1) == 1
        nr.rays_d, nr.rd_stride = rays_d.data_ptr(), rays_d.stride(0)
    nr.ch_an, nr.ch_lr = (ch_an if on_an else -1), (ch_lr if on_lr else -1)
    nr.lambda_an, nr.lambda_lr = float(lambda_an), float(lambda_lr)
    if on_spv:
        nr.lambda_spv, nr.spv_ch_an, nr.spv_ch_lr = float(lambda_spv), int(ch_an), int(ch_lr)
        nr.spv_ray, nr.spv_tot = spv_ray.data_ptr(), spv_tot.data_ptr()
    nr._keep = (rays_d, spv_ray, spv_tot)
    return nr


def normal_spv_reduce(nreg, R, S, ray_loss=None, loss_acc=None):
    """The two batch-wide means of NormalLoss 'an_lr' from the rays' sums (fixed order), the loss term added to the step's loss."""
    _, spv_ray, spv_tot = nreg._keep
    L.check(L.lib().bn_normal_spv_reduce(_p(spv_ray), int(R), int(S), float(nreg.lambda_spv), _p(spv_tot), _p(ray_loss), _p(loss_acc),
                                         _stream()), "bn_normal_spv_reduce")


def merged_composite_forward(z_all, idx, out1, out2, bufs=None, want=("weights", "depth", "acc"), nreg=None, noise=None):
    """Compositing of the depth-sorted union of out1 [R][S1][C] and out2 [R][G][C] read through sort_idx (no cat / gather).
    -> dict with the requested entries of alphas, trans, weights, depth, acc, wsum, var (+ reg: the rays' NormalRegLoss terms,
    with nreg = normal_reg(...))."""
    R, S2 = z_all.shape
    S1, Cc = out1.shape[1], out1.shape[2]
    dev = z_all.device
    b = bufs if bufs is not None else {}
    shapes = dict(alphas=(R, S2), trans=(R, S2), weights=(R, S2), depth=(R,), acc=(R, Cc), wsum=(R,), var=(R,), reg=(R,))
    want = tuple(want) + (("reg",) if (nreg is not None and nreg.rays_d and "reg" not in want) else ())
    o = {k: (b[k] if k in b else torch.empty(shapes[k], dtype=torch.float32, device=dev)) for k in want}
    g = lambda k: _p(o.get(k))
    L.check(L.lib().bn_merged_composite_forward(_p(z_all), _p(idx), _p(out1), _p(out2), S1, S2, Cc, R, g("alphas"), g("trans"),
                                                g("weights"), g("depth"), g("acc"), g("wsum"), g("var"),
                                                None if nreg is None else C.byref(nreg), g("reg") if (nreg is not None and nreg.rays_d) else None,
                                                _nz(noise), _stream()),
            "bn_merged_composite_forward")
    return o


def merged_composite_backward(z_all, idx, out1, out2, d_weights, d_depth, d_acc, d_out1, d_out2, d_wsum=None, nonfinite=None,
                              hs_scale=0.0, depth=None, nreg=None, noise=None):
    """Gradient rows in the SOURCE layouts (d_out1 [R][S1][C], d_out2 [R][G][C]); channel 3 receives d sigma.  hs_scale (with
    the forward's depth): + hs_scale (z - depth)^2 on d loss / d w (HardSurfaceLoss, see ray_shade_loss)."""
    R, S2 = z_all.shape
    S1, Cc = out1.shape[1], out1.shape[2]
    L.check(L.lib().bn_merged_composite_backward(_p(z_all), _p(idx), _p(out1), _p(out2), S1, S2, Cc, R, _p(d_weights), _p(d_depth),
                                                 _p(d_acc), _p(d_wsum), float(hs_scale), _p(depth if hs_scale else None),
                                                 None if nreg is None else C.byref(nreg), _p(d_out1), _p(d_out2), _p(nonfinite),
                                                 _nz(noise), _stream()),
            "bn_merged_composite_backward")


def ray_shade_loss(desc, acc, wsum, depth, var, rays_d, sun_d, rgbs, bufs=None, valid_depth=None, target_depth=None,
                   target_weight=None, target_std=None, ray_loss=None, loss_acc=None, nonfinite=None, extra_loss=None):
    """Ray-level shading + SNerfLoss + DepthLoss + HardSurfaceLoss and their gradients w.r.t. the composited sums in one
    launch (bn_ray_shade_loss).  desc: L.ShadeDesc (rendering.shade_desc).  rays_d / sun_d: (R,3) views with unit inner
    stride (sun_d None: ones).  nonfinite (int64[2]): rays with a non-finite loss term are left out of the step and counted.
    -> dict rgb (R,3), d_acc (R,C), d_wsum (R,), d_depth (R,)."""
    R, Cc = acc.shape
    b = bufs if bufs is not None else {}
    shapes = dict(rgb=(R, 3), d_acc=(R, Cc), d_wsum=(R,), d_depth=(R,))
    o = {k: (b[k] if k in b else torch.empty(sh, dtype=torch.float32, device=acc.device)) for k, sh in shapes.items()}
    use = target_depth is not None and desc.lambda_ds > 0
    vp, vs = _strided(valid_depth if use else None)
    tdp, tds = _strided(target_depth if use else None)
    twp, tws = _strided(target_weight if use else None)
    tsp, tss = _strided(target_std if use else None)

    def rows3(t):
        if t is None:
            return None, 0
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 3
        if t.stride(1) != 1:
            t = t.contiguous()
        return C.c_void_p(t.data_ptr()), t.stride(0)
    rdp, rds = rows3(rays_d)
    sdp, sds = rows3(sun_d)
    L.check(L.lib().bn_ray_shade_loss(C.byref(desc), _p(acc), _p(wsum), _p(depth), _p(var), rdp, rds, sdp, sds, _p(rgbs), vp, vs, tdp, tds,
                                      twp, tws, tsp, tss, R, _p(o["rgb"]), _p(ray_loss), _p(loss_acc),
                                      0 if loss_acc is None else loss_acc.numel(), _p(o["d_acc"]), _p(o["d_wsum"]), _p(o["d_depth"]),
                                      _p(nonfinite), _p(extra_loss), _stream()), "bn_ray_shade_loss")
    return o


def sample_brdf(desc, X, rays, n1, S1, S2, out, backward_of=None, sun_col=8):
    """Per-sample BRDF of --MultiBRDF on stored field-output rows (bn_sample_brdf_forward / _backward).  X (N, C) rows of which the
    first n1 are S1 per ray and the rest S2 per ray; rays (R, >= 6) fp32 rows (sun at sun_col, < 0: ones).  Forward: `out` (N, 4) or
    (N, C) receives [bp, sigma(, the channels behind)].  backward_of = dB (N, 4 | C): `out` (N, C) receives d loss / d X."""
    N, Cc = X.shape
    assert X.is_cuda and X.dtype == torch.float32 and X.is_contiguous() and rays.dtype == torch.float32 and rays.stride(1) == 1
    assert out.is_contiguous() and out.dtype == torch.float32 and out.shape[0] == N
    R = rays.shape[0]
    assert n1 == R * S1 and (n1 == N or N - n1 == R * S2), (N, n1, R, S1, S2)
    if backward_of is None:
        L.check(L.lib().bn_sample_brdf_forward(C.byref(desc), _p(X), _p(rays), R, rays.stride(0), sun_col, N, n1, S1, S2, _p(out),
                                               out.shape[1], _stream()), "bn_sample_brdf_forward")
    else:
        dB = backward_of
        assert dB.is_contiguous() and dB.dtype == torch.float32 and dB.shape[0] == N and out.shape[1] == Cc
        L.check(L.lib().bn_sample_brdf_backward(C.byref(desc), _p(X), _p(rays), R, rays.stride(0), sun_col, N, n1, S1, S2, _p(dB),
                                                dB.shape[1], _p(out), _stream()), "bn_sample_brdf_backward")
    return out


def lambert_tail(z_all, idx, out1, out2, rgbs, rgb_padding, lambda_rgb, d_out1, d_out2, valid_depth=None, target_depth=None,
                 target_weight=None, target_std=None, lambda_ds=0.0, usealldepth=False, ray_loss=None, loss_acc=None, rgb=None,
                 weights=None, depth=None, nonfinite=None, noise=None):
    """Ray-level tail of a Lambertian step in one launch (bn_lambert_tail): merged compositing + shading + SNerfLoss +
    DepthLoss + the backward of all of it, gradient rows written to d_out1 / d_out2.  nonfinite ([2] int64 counters): rays with a
    non-finite loss term are left out and counted, non-finite gradient elements zeroed and counted."""
    R, S2 = z_all.shape
    S1, Cc = out1.shape[1], out1.shape[2]
    use = target_depth is not None and lambda_ds > 0
    vp, vs = _strided(valid_depth if use else None)
    tdp, tds = _strided(target_depth if use else None)
    twp, tws = _strided(target_weight if use else None)
    tsp, tss = _strided(target_std if use else None)
    L.check(L.lib().bn_lambert_tail(_p(z_all), _p(idx), _p(out1), _p(out2), S1, S2, Cc, R, _p(rgbs), vp, vs, tdp, tds, twp, tws,
                                    tsp, tss, float(rgb_padding), float(lambda_rgb), float(lambda_ds if use else 0.0),
                                    int(bool(usealldepth)), _p(ray_loss), _p(loss_acc), 0 if loss_acc is None else loss_acc.numel(),
                                    _p(rgb), _p(weights), _p(depth),
                                    _p(d_out1), _p(d_out2), _p(nonfinite), _nz(noise), _stream()), "bn_lambert_tail")


def adam_multi(param, grad, exp_avg, exp_avg_sq, groups, active, state, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
               grad_scale=1.0, zero_grad=True):
    """Adam over the (lo, hi) groups of one flat buffer in one launch; step counts / lr / draw counter from `state`."""
    n = len(groups)
    lo = (C.c_int64 * n)(*[g[0] for g in groups])
    hi = (C.c_int64 * n)(*[g[1] for g in groups])
    act = (C.c_int32 * n)(*[int(bool(a)) for a in active])
    L.check(L.lib().bn_adam_multi(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), n, lo, hi, act, float(betas[0]), float(betas[1]),
                                  float(eps), float(weight_decay), float(grad_scale), int(bool(zero_grad)), _p(state), _stream()),
            "bn_adam_multi")


# ----------------------------------------------------------------------------------------- BRDFs
def _opt(t):
    return None if t is None else _f32(t)


class RPVFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, l, v, n, w, k, theta, rhoc):
        l, v, n, w, k, theta, rhoc = [_opt(t) for t in (l, v, n, w, k, theta, rhoc)]
        N = n.shape[0]
        brdf = torch.empty(N, 3, dtype=torch.float32, device=n.device)
        aux = torch.empty(N, L.BN_BRDF_AUX, dtype=torch.float32, device=n.device)
        L.check(L.lib().bn_brdf_rpv_forward(_p(l), _p(v), _p(n), _p(w), _p(k), _p(theta), _p(rhoc), N, _p(brdf), _p(aux),
                                            _stream()), "bn_brdf_rpv_forward")
        ctx.save_for_backward(*[t if t is not None else torch.empty(0, device=n.device) for t in (l, v, n, w, k, theta, rhoc)])
        ctx.has = (k is not None, theta is not None, rhoc is not None)
        ctx.mark_non_differentiable(aux)
        return brdf, aux

    @staticmethod
    def backward(ctx, d_brdf, _d_aux):
        l, v, n, w, k, theta, rhoc = ctx.saved_tensors
        hk, ht, hr = ctx.has
        k, theta, rhoc = (k if hk else None), (theta if ht else None), (rhoc if hr else None)
        N = n.shape[0]
        mk = lambda on: torch.empty(N, 3, dtype=torch.float32, device=n.device) if on else None
        d_n, d_w, d_k, d_t, d_r = mk(True), mk(True), mk(hk), mk(ht), mk(hr)
        d_brdf = _f32(d_brdf)      # referenced until the launch is issued
        L.check(L.lib().bn_brdf_rpv_backward(_p(l), _p(v), _p(n), _p(w), _p(k), _p(theta), _p(rhoc), _p(d_brdf), N,
                                             _p(d_n), _p(d_w), _p(d_k), _p(d_t), _p(d_r), _stream()), "bn_brdf_rpv_backward")
        return None, None, d_n, d_w, d_k, d_t, d_r


class HapkeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, l, v, n, w, b, c, theta, hpk_scl, shell):
        l, v, n, w, b, c, theta = [_opt(t) for t in (l, v, n, w, b, c, theta)]
        N = n.shape[0]
        brdf = torch.empty(N, 3, dtype=torch.float32, device=n.device)
        aux = torch.empty(N, L.BN_BRDF_AUX, dtype=torch.float32, device=n.device)
        L.check(L.lib().bn_brdf_hapke_forward(_p(l), _p(v), _p(n), _p(w), _p(b), _p(c), _p(theta), float(hpk_scl), int(shell),
                                              N, _p(brdf), _p(aux), _stream()), "bn_brdf_hapke_forward")
        ctx.save_for_backward(*[t if t is not None else torch.empty(0, device=n.device) for t in (l, v, n, w, b, c, theta)])
        ctx.has = (b is not None, c is not None, theta is not None)
        ctx.hpk_scl, ctx.shell = float(hpk_scl), int(shell)
        ctx.mark_non_differentiable(aux)
        return brdf, aux

    @staticmethod
    def backward(ctx, d_brdf, _d_aux):
        l, v, n, w, b, c, theta = ctx.saved_tensors
        hb, hc, ht = ctx.has
        b, c, theta = (b if hb else None), (c if hc else None), (theta if ht else None)
        N = n.shape[0]
        mk = lambda on: torch.empty(N, 3, dtype=torch.float32, device=n.device) if on else None
        d_n, d_w, d_b, d_c = mk(True), mk(True), mk(hb), mk(hc)
        d_t = torch.empty(N, dtype=torch.float32, device=n.device) if ht else None
        d_brdf = _f32(d_brdf)      # referenced until the launch is issued
        L.check(L.lib().bn_brdf_hapke_backward(_p(l), _p(v), _p(n), _p(w), _p(b), _p(c), _p(theta), ctx.hpk_scl, ctx.shell,
                                               _p(d_brdf), N, _p(d_n), _p(d_w), _p(d_b), _p(d_c), _p(d_t), _stream()),
                "bn_brdf_hapke_backward")
        return None, None, d_n, d_w, d_b, d_c, d_t, None, None


class MicrofacetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, l, v, n, albedo, rough, f0):
        ctx.rough_shape = rough.shape
        l, v, n, albedo, rough = [_f32(t) for t in (l, v, n, albedo, rough.reshape(-1))]
        N = n.shape[0]
        brdf = torch.empty(N, 3, dtype=torch.float32, device=n.device)
        aux = torch.empty(N, L.BN_BRDF_AUX, dtype=torch.float32, device=n.device)
        L.check(L.lib().bn_brdf_microfacet_forward(_p(l), _p(v), _p(n), _p(albedo), _p(rough), float(f0), N, _p(brdf), _p(aux),
                                                   _stream()), "bn_brdf_microfacet_forward")
        ctx.save_for_backward(l, v, n, albedo, rough)
        ctx.f0 = float(f0)
        ctx.mark_non_differentiable(aux)
        return brdf, aux

    @staticmethod
    def backward(ctx, d_brdf, _d_aux):
        l, v, n, albedo, rough = ctx.saved_tensors
        N = n.shape[0]
        d_n = torch.empty(N, 3, dtype=torch.float32, device=n.device)
        d_a = torch.empty_like(d_n)
        d_r = torch.empty(N, dtype=torch.float32, device=n.device)
        d_brdf = _f32(d_brdf)      # referenced until the launch is issued
        L.check(L.lib().bn_brdf_microfacet_backward(_p(l), _p(v), _p(n), _p(albedo), _p(rough), ctx.f0, _p(d_brdf), N,
                                                    _p(d_n), _p(d_a), _p(d_r), _stream()), "bn_brdf_microfacet_backward")
        return None, None, d_n, d_a, d_r.reshape(ctx.rough_shape), None


# ----------------------------------------------------------------------------------------- optimizer
def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
    L.check(L.lib().bn_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), float(lr), float(betas[0]),
                                 float(betas[1]), float(eps), float(weight_decay), int(step), float(grad_scale), _stream()),
            "bn_adam_step")


# ----------------------------------------------------------------------------------------- debug
def count_nonfinite(x, counts=None):
    """check_nan without the host round trip (train_utils.py:14-25): adds the number of NaN / Inf elements of `x` to
    counts[0] / counts[1] (int64 device tensor, created zeroed when None) on the current stream and returns it."""
    x = _f32(x)
    if counts is None:
        counts = torch.zeros(2, dtype=torch.int64, device=x.device)
    if x.numel():
        L.check(L.lib().bn_count_nonfinite(_p(x), x.numel(), _p(counts), _stream()), "bn_count_nonfinite")
    return counts
