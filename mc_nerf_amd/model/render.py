"""Autograd glue between the reference-shaped Python API and the HIP kernels.

``RenderTrainFn`` is one differentiable op for the whole of NeRF_Model.render_rays_train
(reference: model/mc_nerf.py:598-646): coarse MLP -> composite -> device-side selection -> fine MLP ->
composite, with a hand-written backward (composite bwd -> dX chain -> dW) instead of autograd
through ~900 ATen ops.  ``RaygenFn`` does the same for MC_Model.get_rays on selected pixels.

Every random draw of the reference (jitter, the three N(0,1) tensors of sigma2weights, the cap
permutation) is an explicit tensor argument: generated with torch's device RNG by the caller in
normal operation, passed in verbatim by the parity tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .. import ops


@dataclass
class RenderSettings:
    """The renderer constants the kernels need (reference: model/mc_nerf.py:548-571)."""
    samples_c: int
    scale: int
    weight_thresh: float
    sigma_default: float
    white_back: bool
    max_fine_per_ray: int = 128      # model/mc_nerf.py:630
    # "f32": exact-fp32 MFMA; "f16x3": split-f16 MFMA (fp32-grade: the 1e-4 parity modes); "f16" / "bf16": single-pass
    # 16-bit MFMA with 2-byte workspaces (throughput modes, accuracy of the operand rounding: csrc/mcnerf_16.h)
    precision: str = "f32"

    @property
    def samples_f(self):
        return self.samples_c * self.scale


class WorkspacePool:
    """The MLP kernels' saved-operand and gradient workspaces (up to ~46 GB per step at 32768 rays in the split-f16 mode), sized
    ONCE per (net, row capacity, precision) and handed from step to step instead of going through the allocator every forward /
    backward: RenderTrainFn.forward takes a set, .backward gives it back after the weight-gradient kernel has been enqueued
    (stream order makes the re-use safe: the next forward's stores are behind that kernel).  A forward whose backward never
    runs simply keeps its set (garbage-collected with the graph); a second forward before the first backward gets a fresh one.
    `NeRF_Model.reserve_workspaces(n_rays)` fills the pool before the first step (multi-GPU: no first-step allocation in any
    rank between two barriers)."""
    MAX_KEYS = 6

    def __init__(self):
        self.free = {}

    def _key(self, kind, net, capacity, device, precision):
        return (kind, net.triple, int(capacity), str(device), precision)

    def _take(self, key, make):
        lst = self.free.get(key)
        return lst.pop() if lst else make()

    def _give(self, key, ws):
        if key not in self.free and len(self.free) >= self.MAX_KEYS:
            self.free.pop(next(iter(self.free)))            # (a changing shape -- test paths -- must not pin memory for ever)
        self.free.setdefault(key, []).append(ws)

    def take_save(self, net, capacity, device, precision):
        return self._take(self._key("save", net, max(int(capacity), 1), device, precision), lambda: ops.alloc_save(net, capacity, device, precision=precision))

    def give_save(self, net, save, precision):
        self._give(self._key("save", net, save.capacity, save.act.device, precision), save)

    def take_grad(self, net, save, precision):
        return self._take(self._key("grad", net, save.capacity, save.act.device, precision), lambda: ops.alloc_grad_ws(net, save, precision))

    def give_grad(self, net, save, precision, ws):
        self._give(self._key("grad", net, save.capacity, save.act.device, precision), ws)


def _pool(owner) -> WorkspacePool:
    if getattr(owner, "ws_pool", None) is None:
        owner.ws_pool = WorkspacePool()
    return owner.ws_pool


def _cap_needed(st: RenderSettings) -> bool:
    """The cap of model/mc_nerf.py:630-632 can only bind when a ray can select more than 128 samples."""
    return st.samples_f > st.max_fine_per_ray


def select_and_cap(st: RenderSettings, w_sel, wmax, n_rays, cap_perm, train: bool):
    """Device-side selection; applies the random cap in training.  Returns idx, count, out_f, max_rows.
    No host synchronisation: the cap's random subset is drawn on the device (ops.cap_random) with a seed word taken
    from torch's device generator.  Only a caller-supplied permutation (`cap_perm`, the parity-test input that replays
    the reference's CPU randperm, :631) goes through the host: its length IS the host-side count."""
    idx, count, out_f = ops.select_fine(w_sel, wmax, st.weight_thresh, st.scale, st.sigma_default)
    max_rows = n_rays * st.samples_f
    if train and _cap_needed(st):
        keep = n_rays * st.max_fine_per_ray
        if cap_perm is not None:
            k = int(count.item())                  # (test path only)
            if k > keep:
                perm = cap_perm[:keep].to(device=idx.device, dtype=torch.int64).contiguous()
                idx, count = ops.cap_gather(idx, perm, keep)
                max_rows = keep
            else:
                max_rows = min(max_rows, max(k, 1))
        else:
            seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int32, device=idx.device)
            idx, count = ops.cap_random(idx, count, max_rows, keep, seed)
            max_rows = keep
    return idx, count, out_f, max_rows


class RenderTrainFn(torch.autograd.Function):
    """rgb_c, rgb_f, depth_c = f(rays_d, rays_o, *coarse_params, *fine_params) with explicit draws."""

    @staticmethod
    def forward(ctx, owner, model_c, model_f, step_r, only_coarse, jitter, eps_c, eps_sel, eps_f, cap_perm,
                rays_d, rays_o, *params):
        st: RenderSettings = owner.settings
        dev = rays_d.device
        N = rays_d.shape[0]
        rays_d = rays_d.contiguous()
        rays_o = rays_o.contiguous()
        need_grad = any(ctx.needs_input_grad)
        barf_w = owner.emmbedding_xyz.barf_weights_on(step_r, dev, pad=10)
        jit = jitter.reshape(-1).contiguous()

        # ---- coarse pass (dense [N,Sc] grid)
        net_c = model_c.net
        flat_c = model_c.flat_params()
        prec = st.precision
        packed_c = ops.pack_weights(net_c, flat_c, precision=prec, range_flags=model_c.range_flags(prec, dev))
        out_c = torch.empty(N, st.samples_c, 4, dtype=torch.float32, device=dev)
        save_c = _pool(owner).take_save(net_c, N * st.samples_c, dev, prec) if need_grad else None
        ops.mlp_fwd(net_c, flat_c, packed_c, rays_o, rays_d, owner.z_vals_c, jit, barf_w, out_c, save=save_c, precision=prec)
        rgb_c, depth_c, _, w_sel, wmax = ops.composite_fwd(out_c, rays_d, owner.z_vals_c, jit, eps_c,
                                                          None if only_coarse else eps_sel, st.white_back,
                                                          want_depth=only_coarse)
        ctx.only_coarse = only_coarse
        ctx.owner, ctx.model_c, ctx.model_f = owner, model_c, model_f
        ctx.n_c = len(model_c.ordered_parameters())
        if only_coarse:
            if need_grad:
                ctx.save_for_backward(rays_d, rays_o, jit, eps_c, barf_w, out_c, flat_c, packed_c)
                ctx.save_c = save_c
            ctx.mark_non_differentiable(depth_c)
            return rgb_c, None, depth_c

        # ---- selection (no host sync) and fine pass on the compacted (ray, sample) list
        idx, count, out_f, max_rows = select_and_cap(st, w_sel, wmax, N, cap_perm, train=True)
        net_f = model_f.net
        flat_f = model_f.flat_params()
        packed_f = ops.pack_weights(net_f, flat_f, precision=prec, range_flags=model_f.range_flags(prec, dev))
        save_f = _pool(owner).take_save(net_f, max_rows, dev, prec) if need_grad else None
        ops.mlp_fwd(net_f, flat_f, packed_f, rays_o, rays_d, owner.z_vals_f, jit, barf_w, out_f,
                    idx=idx, count=count, max_rows=max_rows, save=save_f, precision=prec)
        rgb_f, _, _, _, _ = ops.composite_fwd(out_f, rays_d, owner.z_vals_f, jit, eps_f, None, st.white_back)
        if need_grad:
            ctx.save_for_backward(rays_d, rays_o, jit, eps_c, barf_w, out_c, flat_c, packed_c,
                                  eps_f, out_f, flat_f, packed_f, idx, count)
            ctx.save_c, ctx.save_f, ctx.max_rows = save_c, save_f, max_rows
        owner.last_selection = (idx, count)
        return rgb_c, rgb_f, None

    @staticmethod
    def backward(ctx, d_rgb_c, d_rgb_f, _d_depth):
        owner, model_c, model_f = ctx.owner, ctx.model_c, ctx.model_f
        st: RenderSettings = owner.settings
        if getattr(ctx, "workspaces_given_back", False):
            # retain_graph=True + a second backward: the saved-operand set went back to the pool after the first one and the next
            # forward may already have overwritten it -- refuse instead of returning gradients of clobbered operands
            raise ops._lib.McnerfError("RenderTrainFn.backward ran twice on one forward (retain_graph): its saved-operand workspaces were "
                                  "returned to the pool by the first backward; run the forward again")
        saved = ctx.saved_tensors
        rays_d, rays_o, jit, eps_c, barf_w, out_c, flat_c, packed_c = saved[:8]
        dev = rays_d.device
        N = rays_d.shape[0]
        want_rays = ctx.needs_input_grad[10] or ctx.needs_input_grad[11]
        d_od = torch.zeros(2, N, 3, dtype=torch.float32, device=dev) if want_rays else None      # (one fill for both)
        d_o, d_d = (d_od[0], d_od[1]) if want_rays else (None, None)
        # gradient buffers: slices of the step-level arena when a FlatGradSync provided one (so the whole
        # step's gradient is one contiguous all-reduce message), otherwise fresh zeroed buffers
        arena = getattr(owner, "grad_arena", None)
        n_c, n_f = flat_c.numel(), model_f.flat_params().numel()
        if arena is not None:
            g_c = arena[:n_c]
            owner.grad_arena_used = True
        else:
            g_c = torch.zeros_like(flat_c)
        g_f = None

        def net_backward(model, flat, packed, zgrid, eps, out, d_rgb, save, grads, idx=None, count=None, max_rows=0):
            """One net's composite backward, dX chain and weight gradients.  The saved-operand set goes back to the pool on EVERY way
            out -- no gradient wanted, or an error (a range overflow raised by a kernel, out of memory in take_grad) --: a retried
            step must find the pool as the forward left it, not one set short."""
            net = model.net
            pool = _pool(owner)
            try:
                if d_rgb is None:
                    return
                d_out, gmax = ops.composite_bwd(out, zgrid, jit, eps, d_rgb.contiguous(), st.white_back, want_gmax=True)
                dy, dsh = pool.take_grad(net, save, st.precision)
                try:
                    ops.mlp_bwd(net, flat, packed, rays_o, rays_d, zgrid, jit, barf_w, out, d_out, save, dy, dsh,
                                d_o, d_d, idx=idx, count=count, max_rows=max_rows, precision=st.precision, gmax=gmax)
                    rows = max_rows if idx is not None else N * zgrid.numel()
                    ops.mlp_dw(net, save, dy, dsh, grads, rows, count=count, precision=st.precision, gmax=gmax)
                finally:
                    pool.give_grad(net, save, st.precision, (dy, dsh))
            finally:
                pool.give_save(net, save, st.precision)       # (everything that reads the set is enqueued: the next step may overwrite it)

        # from here on the saved-operand sets leave this context, whatever happens: a second backward on a retained graph is refused
        save_c, save_f = ctx.save_c, (None if ctx.only_coarse else ctx.save_f)
        ctx.save_c = ctx.save_f = None
        ctx.workspaces_given_back = True
        if not ctx.only_coarse:
            eps_f, out_f, flat_f, packed_f, idx, count = saved[8:]
            g_f = arena[n_c:n_c + n_f] if arena is not None else torch.zeros_like(flat_f)
            try:
                net_backward(model_f, flat_f, packed_f, owner.z_vals_f, eps_f, out_f, d_rgb_f, save_f, g_f,
                             idx=idx, count=count, max_rows=ctx.max_rows)
            except BaseException:
                _pool(owner).give_save(model_c.net, save_c, st.precision)      # (the coarse net's set never reaches its own backward)
                raise
        net_backward(model_c, flat_c, packed_c, owner.z_vals_c, eps_c, out_c, d_rgb_c, save_c, g_c)
        grads_c = model_c.grad_views(g_c)
        grads_f = model_f.grad_views(g_f) if g_f is not None else [None] * len(model_f.ordered_parameters())
        owner.last_flat_grads = (g_c, g_f)
        return (None,) * 10 + (d_d if ctx.needs_input_grad[10] else None,
                               d_o if ctx.needs_input_grad[11] else None) + tuple(grads_c) + tuple(grads_f)


def render_test(owner, model_c, model_f, rays_d, rays_o, eps_c, eps_sel, eps_f, prepared=None):
    """NeRF_Model.render_rays_test (model/mc_nerf.py:648-680): no jitter, step_r = 1, no cap, no grad.
    `prepared` = (packed_c, packed_f, barf_w) from an enclosing chunk loop (the weights do not change inside it)."""
    st: RenderSettings = owner.settings
    dev = rays_d.device
    N = rays_d.shape[0]
    rays_d = rays_d.contiguous()
    rays_o = rays_o.contiguous()
    net_c, net_f = model_c.net, model_f.net
    flat_c, flat_f = model_c.flat_params(), model_f.flat_params()
    prec = st.precision
    if prepared is None:
        prepared = (ops.pack_weights(net_c, flat_c, precision=prec, range_flags=model_c.range_flags(prec, dev)),
                    ops.pack_weights(net_f, flat_f, precision=prec, range_flags=model_f.range_flags(prec, dev)),
                    owner.emmbedding_xyz.barf_weights_on(1, dev, pad=10))
    packed_c, packed_f, barf_w = prepared
    out_c = torch.empty(N, st.samples_c, 4, dtype=torch.float32, device=dev)
    ops.mlp_fwd(net_c, flat_c, packed_c, rays_o, rays_d, owner.z_vals_c, None, barf_w, out_c, precision=prec)
    _, _, _, w_sel, wmax = ops.composite_fwd(out_c, rays_d, owner.z_vals_c, None, eps_c, eps_sel, st.white_back)
    idx, count, out_f, max_rows = select_and_cap(st, w_sel, wmax, N, None, train=False)
    ops.mlp_fwd(net_f, flat_f, packed_f, rays_o, rays_d, owner.z_vals_f, None, barf_w, out_f,
                idx=idx, count=count, max_rows=max_rows, precision=prec)
    rgb, depth, opacity, _, _ = ops.composite_fwd(out_f, rays_d, owner.z_vals_f, None, eps_f, None, st.white_back,
                                                  want_depth=True)
    owner.last_selection = (idx, count)
    return rgb, depth, opacity


class RaygenFn(torch.autograd.Function):
    """rays_d, rays_o = f(pose[3,4], kinv[3,3]) for the given pixel ids of one camera."""

    @staticmethod
    def forward(ctx, pose, kinv, pix, W):
        pose = pose.contiguous().float()
        kinv = kinv.contiguous().float()
        pix = pix.contiguous()
        d, o = ops.raygen_fwd(pose, kinv, pix, W)
        ctx.save_for_backward(pose, kinv, pix)
        ctx.W = W
        return d, o

    @staticmethod
    def backward(ctx, g_d, g_o):
        pose, kinv, pix = ctx.saved_tensors
        d_pose, d_kinv = ops.raygen_bwd(pose, kinv, pix, ctx.W, g_d.contiguous(), g_o.contiguous())
        return d_pose, d_kinv, None, None


class CameraFn(torch.autograd.Function):
    """K, Kinv, pose, calib_pose, pix_intr, pix_extr = f(weights_pose, weights_pose_intr, weights_fx, weights_fy, weights_ux,
    weights_uy; calibration world points) for all cameras in one fused kernel each way (reference: model/mc_nerf.py:171-210,
    269-316 and the reprojection branch :147-152, 236-267).  wpts_intr / wpts_extr [C,P,3] may be None (pixels not wanted)."""

    @staticmethod
    def forward(ctx, wpose, wpose_intr, wfx, wfy, wux, wuy, H, W, wpts_intr=None, wpts_extr=None):
        args = [t.contiguous().float() for t in (wpose, wpose_intr, wfx, wfy, wux, wuy)]
        pts = [None if t is None else t.contiguous().float() for t in (wpts_intr, wpts_extr)]
        K, Kinv, pose, calib, pi, pe = ops.camera_fwd(*args, H, W, pts[0], pts[1])
        ctx.save_for_backward(*args)
        ctx.pts = pts
        ctx.hw = (H, W)
        ctx.set_materialize_grads(False)
        return K, Kinv, pose, calib, pi, pe

    @staticmethod
    def backward(ctx, dK, dKinv, dpose, dcalib, dpi, dpe):
        args = ctx.saved_tensors
        grads = ops.camera_bwd(*args, ctx.hw[0], ctx.hw[1], dK, dKinv, dpose, dcalib, ctx.pts[0], ctx.pts[1], dpi, dpe)
        return tuple(g if need else None for g, need in zip(grads, ctx.needs_input_grad[:6])) + (None, None, None, None)


class TrainLossFn(torch.autograd.Function):
    """MC_NeRF_Loss.forward (model/loss.py:13-31) for the NeRF stages' keys {"intr", "rgb"} as ONE launch: value and every
    gradient; backward scales the saved gradients by the upstream scalar (one launch)."""

    @staticmethod
    def forward(ctx, pd, pt_gt, rgb_c, rgb_f, gt, H, W, normalise):
        cf = lambda t: None if t is None else t.contiguous().float()
        pd, pt_gt, rgb_c, rgb_f, gt = cf(pd), cf(pt_gt), cf(rgb_c), cf(rgb_f), cf(gt)
        out, d_pd, d_c, d_f = ops.train_loss(pd, pt_gt, H, W, normalise, rgb_c, rgb_f, gt)
        ctx.grads = (d_pd, d_c, d_f)
        ctx.parts = out
        return out[0]

    @staticmethod
    def backward(ctx, g):
        d_pd, d_c, d_f = ctx.grads
        ops.scale3_(d_pd, d_c, d_f, g.contiguous().float().reshape(1))
        ctx.grads = None
        return d_pd, None, d_c, d_f, None, None, None, None


class ReprojLossFn(torch.autograd.Function):
    """MC_NeRF_Loss.get_reproject_loss (model/loss.py:45-58) as one kernel each way."""

    @staticmethod
    def forward(ctx, pd, gt, H, W):
        pd, gt = pd.contiguous().float(), gt.contiguous().float()
        ctx.save_for_backward(pd, gt)
        ctx.hw = (H, W)
        return ops.reproj_loss_fwd(pd, gt, H, W)

    @staticmethod
    def backward(ctx, dloss):
        pd, gt = ctx.saved_tensors
        return ops.reproj_loss_bwd(pd, gt, ctx.hw[0], ctx.hw[1], dloss.contiguous().float()), None, None, None
