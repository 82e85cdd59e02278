"""MI355X-native counterparts of the reference's ``model/mc_nerf.py`` classes.

``MC_Model`` / ``NeRF_Model`` keep the reference's constructor (`sys_param` dict), call signatures,
attribute names, parameter names and checkpoint format, so the reference's
``main.py`` train / demo loop can drive them unchanged; the volumetric rendering underneath is the
HIP path of ``libmcnerf.so`` (no eager ATen ops, no host syncs on the train path for configs where
the 128-per-ray cap cannot bind).

Reference line numbers below are into the reference's ``model/mc_nerf.py`` unless stated otherwise.
"""
from __future__ import annotations

import logging
import os
import time
from pathlib import Path
from typing import Optional

import torch
import torch.nn as nn
import torch.distributed as dist

from .. import ops
from ..data import DeviceImageSet
from .net_block import CorseFine_NeRF, SinCosEmbedding
from .render import CameraFn, RaygenFn, RenderSettings, RenderTrainFn, render_test


def _rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


# ============================================================================ renderer
class NeRF_Model(nn.Module):
    """Renderer (reference :543-867).  forward(rays_d, rays_o, epoch, step_r) -> (rgb_c, rgb_f) in
    training mode, forward(rays_d, rays_o) -> (rgb, depth, opacity) otherwise (:586-596)."""

    def __init__(self, sys_param):
        logging.info("Creating NeRF Model (HIP / gfx950)...")
        super().__init__()
        self.sys_param = sys_param
        self.mode = sys_param["mode"]
        self.device = sys_param["device_type"]
        self.near = sys_param["near"]
        self.far = sys_param["far"]
        self.samples_c = sys_param["samples"]
        self.sample_scale = sys_param["scale"]
        self.samples_f = self.samples_c * self.sample_scale
        self.dim_sh = 3 * (sys_param["MLP_deg"] + 1) ** 2
        self.white_back = sys_param["white_back"]
        self.weights_pth = sys_param.get("root_weight", "./weights")
        self.train_img_pth = sys_param.get("demo_render_pth", "./results")
        self.batch_test = sys_param["batch"]
        self.sigma_default = sys_param["sigma_default"]
        self.weight_thresh = sys_param["sample_weight_thresh"]
        self.render_h = sys_param.get("res_h", 800)
        self.render_w = sys_param.get("res_w", 800)
        self.data_name = sys_param.get("data_name", "scene")
        # identical to the reference's grids (:570-571): fp32 torch.linspace on the target device
        self.z_vals_c = torch.linspace(self.near, self.far, self.samples_c, device=self.device)
        self.z_vals_f = torch.linspace(self.near, self.far, self.samples_f, device=self.device)
        self.global_step = 0
        self.emmbedding_xyz = SinCosEmbedding(sys_param)
        self.nerf_coarse = CorseFine_NeRF(sys_param, type="coarse")
        self.nerf_fine = CorseFine_NeRF(sys_param, type="fine")
        # "precision" is this build's own sys_param key (absent in the reference's config): "f32" or "f16x3"
        self.precision = sys_param.get("precision", "f32")
        if self.precision not in ops.PRECISIONS:
            raise ValueError(f"precision must be one of {ops.PRECISIONS}")
        if self.precision != "f32" and (self.nerf_coarse.net.fp32_only or self.nerf_fine.net.fp32_only):
            raise ValueError("a net with more than one skip layer or SH degree 3 runs in precision 'f32' only "
                             f"(the register-chain kernels of '{self.precision}' take one skip layer and MLP_deg <= 2)")
        self.settings = RenderSettings(self.samples_c, self.sample_scale, float(self.weight_thresh),
                                       float(self.sigma_default), bool(self.white_back), precision=self.precision)
        self.last_selection = None
        self.last_flat_grads = None
        self.grad_arena = None            # set per step by distributed.FlatGradSync.prepare()
        self.grad_arena_used = False
        self.ws_pool = None               # render.WorkspacePool: the kernels' saved-operand / gradient workspaces, re-used from step to step
        if self.mode != 0:
            self.nerf_ckpt_name = sys_param["demo_ckpt"]
            ckpt = torch.load(Path(self.nerf_ckpt_name), map_location=self.device)
            self.nerf_coarse.load_state_dict(self.rewrite_nerf_ckpt(ckpt, coarse=True))
            self.nerf_fine.load_state_dict(self.rewrite_nerf_ckpt(ckpt))
            logging.info("Loading weights:{}".format(self.nerf_ckpt_name))

    # ------------------------------------------------------------------ API (:586-596)
    def forward(self, *args):
        self.global_step += 1
        if self.mode == 0:
            rays_d, rays_o, cur_epoch, step_r = args
            rgb_c, rgb_f = self.render_rays_train(rays_d, rays_o, cur_epoch, step_r, only_coarse=False)
            return rgb_c, rgb_f
        rays_d, rays_o = args
        return self.render_rays_test(rays_d, rays_o, self.nerf_coarse, self.nerf_fine)

    def _dev(self, t):
        return t.to(device=self.z_vals_c.device, dtype=torch.float32)

    def render_rays_train(self, rays_d, rays_o, cur_epoch, step_r, only_coarse=False, *,
                          jitter=None, eps_c=None, eps_sel=None, eps_f=None, cap_perm=None):
        """Reference :598-646.  The keyword-only tensors are the reference's random draws
        (U(0,(far-near)/Sc) per ray; three N(0,1) tensors; the cap permutation); when omitted they are
        drawn from torch's device generator in the reference's order."""
        N, dev = rays_d.shape[0], rays_d.device
        if N == 0:                                  # empty batch: empty results (the reference's tensor ops do the same)
            e3, e1 = rays_d.new_zeros(0, 3), rays_d.new_zeros(0, 1)
            return (e3, None, e1) if only_coarse else (e3, e3.clone())
        if jitter is None:
            jitter = torch.empty(N, 1, device=dev).uniform_(0.0, (self.far - self.near) / self.samples_c)
        if eps_c is None:
            eps_c = torch.randn(N, self.samples_c, device=dev)
        if not only_coarse:
            if eps_sel is None:
                eps_sel = torch.randn(N, self.samples_c, device=dev)
            if eps_f is None:
                eps_f = torch.randn(N, self.samples_f, device=dev)
        params = self.nerf_coarse.ordered_parameters() + self.nerf_fine.ordered_parameters()
        self.nerf_coarse.flat_params()
        self.nerf_fine.flat_params()
        rgb_c, rgb_f, depth_c = RenderTrainFn.apply(self, self.nerf_coarse, self.nerf_fine, step_r, only_coarse,
                                                    self._dev(jitter), self._dev(eps_c).contiguous(),
                                                    None if eps_sel is None else self._dev(eps_sel).contiguous(),
                                                    None if eps_f is None else self._dev(eps_f).contiguous(),
                                                    cap_perm, rays_d, rays_o, *params)
        if only_coarse:
            return rgb_c, None, depth_c
        return rgb_c, rgb_f

    @torch.no_grad()
    def render_rays_test(self, rays_d, rays_o, model_coarse, model_fine, *, eps_c=None, eps_sel=None, eps_f=None, _prepared=None):
        """Reference :648-680 (the nets are arguments because valid_train passes freshly loaded ones)."""
        N, dev = rays_d.shape[0], rays_d.device
        if N == 0:
            return rays_d.new_zeros(0, 3), rays_d.new_zeros(0, 1), rays_d.new_zeros(0, 1)
        if eps_c is None:
            eps_c = torch.randn(N, self.samples_c, device=dev)
        if eps_sel is None:
            eps_sel = torch.randn(N, self.samples_c, device=dev)
        if eps_f is None:
            eps_f = torch.randn(N, self.samples_f, device=dev)
        return render_test(self, model_coarse, model_fine, rays_d.float(), rays_o.float(),
                           self._dev(eps_c).contiguous(), self._dev(eps_sel).contiguous(), self._dev(eps_f).contiguous(), _prepared)

    def reserve_workspaces(self, n_rays: int):
        """Sizes the training workspaces of both nets for `n_rays`-ray steps now (they are re-used from step to step afterwards:
        render.WorkspacePool), so that the first timed / synchronised step of a multi-GPU run allocates nothing."""
        from .render import _cap_needed, _pool
        dev, st, pool = self.z_vals_c.device, self.settings, _pool(self)
        rows_f = n_rays * (st.max_fine_per_ray if _cap_needed(st) else st.samples_f)
        for net, rows in ((self.nerf_coarse.net, n_rays * st.samples_c), (self.nerf_fine.net, rows_f)):
            save = pool.take_save(net, rows, dev, st.precision)
            pool.give_grad(net, save, st.precision, pool.take_grad(net, save, st.precision))
            pool.give_save(net, save, st.precision)

    # ------------------------------------------------------------------ per-pass API of the reference (:682-736)
    def inference(self, model, embedding_xyz, step_r, xyz, rays_d, z_vals, idx_render=None, coarse=True, *, eps=None):
        """Reference :682-727.  Differentiable like the reference's: when autograd is recording and any input or parameter
        requires a gradient the pass runs on the stand-alone differentiable kernels (`_inference_general`: EncodeFn, MlpApplyFn,
        composites in tensor ops); otherwise on the fused forward-only path (`_inference`).

        DISPATCH, explicitly: the differentiable path is taken whenever grad mode is on and ANY of `xyz`, `rays_d`, `z_vals` or
        the net's parameters requires a gradient -- i.e. always for a training-mode model called outside `torch.no_grad()`.
        That path computes in exact fp32 whatever `precision` the model was built with, ignores `coarse`, and keeps
        (depth + 2) * rows * width floats of saved activations: it exists for API parity (a caller differentiating through
        `inference` directly), not for rendering.  Render under `torch.no_grad()` (as `render_rays_test`, `valid_train` and the
        demo path do) to get the fused kernels in the configured precision; training goes through `render_rays_train`."""
        if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad
                                           for t in (xyz, rays_d, z_vals, *model.parameters())):
            return self._inference_general(model, embedding_xyz, step_r, xyz, rays_d, z_vals.float(), idx_render, eps)
        with torch.no_grad():
            return self._inference(model, embedding_xyz, step_r, xyz, rays_d, z_vals, idx_render, coarse, eps=eps)

    def _inference(self, model, embedding_xyz, step_r, xyz, rays_d, z_vals, idx_render=None, coarse=True, *, eps=None):
        """One pass of one net on given samples -> (rgb [N,3], sigmas [N,S], xyz, depth [N,1], opacity [N,1]), the
        reference's `inference` (:682-727) for callers that use it directly.

        The fused HIP kernels generate the positions themselves from (origin, direction, z grid + per-ray jitter).  When
        `z_vals` has the reference's structure (its own call sites always do): `z_vals[n, j] = grid[j] + jitter[n]` with
        `grid` this model's coarse or fine grid, that path is taken and the origins are recovered from `xyz[:, 0]`.  Any
        other `xyz` / `z_vals` goes through the stand-alone kernels on the given positions (`_inference_general`).
        Forward only (`inference` routes gradient requests to `_inference_general`).  `eps` is the N(0,1) draw of
        `sigma2weights` (drawn from the device generator when not given)."""
        N, S_ = z_vals.shape
        dev = rays_d.device
        z_vals = z_vals.float()
        structured = S_ in (self.samples_c, self.samples_f)
        if structured:
            grid = self.z_vals_c if S_ == self.samples_c else self.z_vals_f
            jitter = (z_vals[:, 0] - grid[0]).contiguous()
            structured = bool(torch.allclose(z_vals, grid.unsqueeze(0) + jitter.unsqueeze(1), atol=1e-5, rtol=0))
        if not structured:
            return self._inference_general(model, embedding_xyz, step_r, xyz, rays_d, z_vals, idx_render, eps)
        rays_d = rays_d.float().contiguous()
        rays_o = (xyz.reshape(N, S_, 3)[:, 0].float() - rays_d * z_vals[:, :1]).contiguous()
        st = self.settings
        flat = model.flat_params()
        packed = ops.pack_weights(model.net, flat, precision=st.precision)
        barf_w = embedding_xyz.barf_weights_on(step_r, dev, pad=10)
        if idx_render is None:
            out = torch.empty(N, S_, 4, dtype=torch.float32, device=dev)
            ops.mlp_fwd(model.net, flat, packed, rays_o, rays_d, grid, jitter, barf_w, out, precision=st.precision)
        else:                                                   # scatter into the defaults (:692-694, 701)
            out = torch.ones(N, S_, 4, dtype=torch.float32, device=dev)
            out[..., 0] = st.sigma_default
            idx = idx_render.to(device=dev, dtype=torch.int32).contiguous()
            count = torch.tensor([idx.shape[0]], dtype=torch.int32, device=dev)
            ops.mlp_fwd(model.net, flat, packed, rays_o, rays_d, grid, jitter, barf_w, out, idx=idx, count=count,
                        max_rows=idx.shape[0], precision=st.precision)
        if eps is None:
            eps = torch.randn(N, S_, device=dev)
        rgb, depth, opacity, _, _ = ops.composite_fwd(out, rays_d, grid, jitter, self._dev(eps).float().contiguous(), None,
                                                      st.white_back, want_depth=True)
        return rgb, out[..., 0], xyz, depth, opacity

    def _inference_general(self, model, embedding_xyz, step_r, xyz, rays_d, z_vals, idx_render, eps):
        """`inference` on arbitrary sample positions (reference :682-727 literally): encode the given `xyz` and run the net
        with the stand-alone exact-fp32 kernels through the modules' own forwards (`SinCosEmbedding.forward`,
        `CorseFine_NeRF.forward`: differentiable when a gradient is requested), scatter into the defaults (sigma_default,
        white) when an index list is given, then both composites in tensor ops (:705-725)."""
        st = self.settings
        N, S_ = z_vals.shape
        dev = rays_d.device
        rays_d = rays_d.float()
        xyz3 = xyz.reshape(N, S_, 3).float()
        if idx_render is None:
            dirs = rays_d.unsqueeze(1).expand(-1, S_, -1).reshape(-1, 3)
            out = model(embedding_xyz(xyz3.reshape(-1, 3), step_r), dirs).reshape(N, S_, 4)
        else:
            r, j = idx_render[:, 0].long().to(dev), idx_render[:, 1].long().to(dev)
            out = torch.ones(N, S_, 4, dtype=torch.float32, device=dev)
            out[..., 0] = st.sigma_default
            if r.numel():
                raw = model(embedding_xyz(xyz3[r, j], step_r), rays_d[r])
                out = out.index_put((r, j), raw)
        if eps is None:
            eps = torch.randn(N, S_, device=dev)
        sig, rgbs = out[..., 0], out[..., 1:]
        deltas = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full_like(z_vals[:, :1], 1e10)], -1)
        sd = torch.nn.functional.softplus(sig) * (deltas * rays_d.norm(dim=-1, keepdim=True))
        alpha = 1.0 - torch.exp(-sd)
        T = torch.exp(-torch.cat([torch.zeros_like(sd[:, :1]), sd[:, :-1]], 1).cumsum(1))
        prob = T * alpha
        opacity, depth = prob.sum(1, keepdim=True), (z_vals * prob).sum(1, keepdim=True)
        w = self.sigma2weights(deltas, sig, self._dev(eps).float())
        rgb = (w.unsqueeze(-1) * rgbs).sum(1)
        if st.white_back:
            rgb = rgb + 1.0 - w.sum(1, keepdim=True)
        return rgb, sig, xyz, depth, opacity

    @staticmethod
    def sigma2weights(deltas, sigmas, eps=None):
        """Reference :729-736 on arbitrary `deltas` (compatibility helper in plain tensor ops; the render path computes the
        same weights inside the fused composite kernels)."""
        if eps is None:
            eps = torch.randn_like(sigmas)
        alphas = 1.0 - torch.exp(-deltas * torch.nn.functional.softplus(sigmas + eps))
        shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1.0 - alphas + 1e-10], -1)
        return alphas * torch.cumprod(shifted, -1)[:, :-1]

    # ------------------------------------------------------------------ checkpoints (:738-752, 815-837)
    def save_model(self, model, epoch):
        save_path = os.path.join(Path(self.weights_pth), Path("train"))
        os.makedirs(save_path, exist_ok=True)
        stamp = time.strftime("%Y-%m-%d-%H-%M-%S", time.localtime())
        self.model_name = "{}-EPOCH-{}-{}.ckpt".format(self.data_name, epoch, stamp)        # reference attributes (:743-744)
        self.file_path = os.path.join(save_path, self.model_name)
        self.ckpt_path = self.file_path
        if _rank() == 0:
            # `model` is what main.py passes: the (possibly DDP-wrapped) MC_Model; its state_dict() keys are the
            # checkpoint's (DDP's "module." prefix included, exactly as the reference writes them)
            torch.save({"model_nerf": model.state_dict()}, self.file_path)
            logging.info("Save model:{}".format(self.model_name))
        return self.file_path

    @staticmethod
    def rewrite_nerf_ckpt(ckpt, coarse=False):
        """{'model_nerf': MC_Model.state_dict()} -> state dict of one net: the key suffix after the `nerf_coarse` /
        `nerf_fine` path component (reference :815-837; also strips DDP's `module.` prefix that way)."""
        name = "nerf_coarse" if coarse else "nerf_fine"
        out = {}
        for k, v in ckpt["model_nerf"].items():
            parts = k.split(".")
            if name in parts:
                out[".".join(parts[parts.index(name) + 1:])] = v
        return out

    @staticmethod
    def cal_psnr(pred, gt):
        return -10.0 * torch.log10(torch.mean((pred - gt) ** 2))

    psnr_score = cal_psnr          # reference name (:839-848)

    @torch.no_grad()
    def render_chunked(self, rays_d, rays_o, model_coarse, model_fine, chunk=None):
        """render_rays_test over all rays in `chunk`-ray pieces (the demo / validation hot loop, :106-122, 778-786);
        everything stays on the device.  -> rgb [M,3], depth [M,1], opacity [M,1]"""
        chunk = chunk or self.batch_test
        M = rays_d.shape[0]
        rgb = torch.empty(M, 3, dtype=torch.float32, device=rays_d.device)
        depth = torch.empty(M, 1, dtype=torch.float32, device=rays_d.device)
        opac = torch.empty(M, 1, dtype=torch.float32, device=rays_d.device)
        # the weights are constant inside the loop: re-layout them (and upload the BARF weights) once per image, not per chunk
        prec = self.settings.precision
        prepared = None
        if M > 0:
            prepared = (ops.pack_weights(model_coarse.net, model_coarse.flat_params(), precision=prec),
                        ops.pack_weights(model_fine.net, model_fine.flat_params(), precision=prec),
                        self.emmbedding_xyz.barf_weights_on(1, rays_d.device, pad=10))
        for i in range(0, M, chunk):
            r, d, o = self.render_rays_test(rays_d[i:i + chunk], rays_o[i:i + chunk], model_coarse, model_fine, _prepared=prepared)
            rgb[i:i + chunk], depth[i:i + chunk], opac[i:i + chunk] = r, d, o
        return rgb, depth, opac

    @torch.no_grad()
    def valid_train(self, epoch, val_data, epoch_type):
        """Reference :754-813: on rank 0 re-load the checkpoint just written by save_model into fresh nets, render the
        validation view in `batch` chunks, write the image / ground truth / depth PNGs and log the PSNR (computed on the
        device; SSIM / LPIPS are third-party metrics outside the hot path); the other ranks wait at the barrier.
        Returns 0 in the camera-only stage, None otherwise (as the reference); the render is kept in
        `self.last_validation` = dict(rgb, depth, psnr) for callers that want it."""
        if epoch_type in ["CAM_PARAM_EPOCH"]:
            return 0
        if _rank() == 0:
            rays_d, rays_o, gt_rgbs = val_data
            ckpt = torch.load(self.file_path, map_location=self.device)
            logging.info("Loading model: {}".format(self.model_name))
            val_c = CorseFine_NeRF(self.sys_param, type="coarse").to(self.device)
            val_f = CorseFine_NeRF(self.sys_param, type="fine").to(self.device)
            val_c.load_state_dict(self.rewrite_nerf_ckpt(ckpt, coarse=True))
            val_f.load_state_dict(self.rewrite_nerf_ckpt(ckpt, coarse=False))
            val_c.eval(), val_f.eval()
            rgb, depth, _ = self.render_chunked(rays_d, rays_o, val_c, val_f)
            gt = gt_rgbs.reshape(-1, 3).to(rgb.device)
            psnr = self.cal_psnr(rgb, gt)
            self.last_validation = {"rgb": rgb, "depth": depth, "psnr": psnr}
            logging.info("PSNR:{}".format(float(psnr)))
            if rgb.shape[0] == self.render_h * self.render_w:
                self._save_validation_images(epoch, rgb, gt, depth)
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
        return None

    def _save_validation_images(self, epoch, rgb, gt, depth):
        try:
            from PIL import Image
        except ImportError:                                     # image files are a convenience, not part of the path
            return
        base = os.path.join(Path(self.train_img_pth), Path(self.data_name))
        os.makedirs(base, exist_ok=True)

        def u8(t, ch):
            return (t.reshape(self.render_h, self.render_w, ch).clamp(0, 1) * 255).round().to(torch.uint8).cpu().numpy()
        Image.fromarray(u8(rgb, 3), "RGB").save(os.path.join(base, "epoch_{}.png".format(epoch)))
        Image.fromarray(u8(gt, 3), "RGB").save(os.path.join(base, "epoch_{}_gt.png".format(epoch)))
        Image.fromarray(u8(depth, 1)[..., 0], "L").save(os.path.join(base, "epoch_{}_depth.png".format(epoch)))


# ============================================================================ multi-camera model
class MC_Model(nn.Module):
    """Camera parametrisation + ray generation + renderer (reference :24-540).

    Train call:  model(data, epoch, epoch_type, cur_ratio) -> (loss_dict, intr_show, pose_show, rays_valid)
    Demo call:   model(img_idx) -> (rgbs, depth, opacity) on the CPU, chunked by ``batch`` (:106-122).
    """

    def __init__(self, sys_param):
        logging.info("Creating MC-NeRF Model (HIP / gfx950)...")
        super().__init__()
        self.sys_param = sys_param
        self.mode = sys_param["mode"]
        self.device = sys_param["device_type"]
        self.batch = sys_param["batch"]
        self.intr = sys_param["intr_mat"]
        self.intr_inv = sys_param["intr_mat_inv"]
        # the constant intrinsics live on the device from the start (a per-step .to(device) of a host tensor is a
        # synchronous copy: it would stall the host behind every queued kernel)
        self.intr_train, self.intr_test, self.intr_val = [t.to(self.device) for t in self.intr]
        self.intr_train_inv, self.intr_test_inv, self.intr_val_inv = [t.to(self.device) for t in self.intr_inv]
        self.gt_pose = sys_param["gt_pose"].to(self.device)
        self.test_pose = sys_param["test_pose"].to(self.device)
        self.valid_pose = sys_param["valid_pose"].to(self.device)
        self.valid_rgbs = sys_param["valid_rgbs"].to(self.device)
        self.img_h = sys_param["data_img_h"]
        self.img_w = sys_param["data_img_w"]
        self.data_name = sys_param.get("data_name", "scene")
        self.data_numb = sys_param["data_numb"]
        self.train_numb, self.test_numb, self.val_numb = self.data_numb
        self.register_parameters()
        self.nerf = NeRF_Model(sys_param).to(self.device)
        self.count_rays = 0
        self.intr_inv_adj = None
        self.opt_idx = 0
        self.last_epoch_type = 0

    # ------------------------------------------------------------------ parameters (:347-371)
    def register_parameters(self):
        C, dev = self.train_numb, self.device
        for name, shape in (("weights_pose", (C, 6)), ("weights_pose_intr", (C, 6)), ("weights_ux", (C,)),
                            ("weights_uy", (C,)), ("weights_fx", (C,)), ("weights_fy", (C,))):
            self.register_parameter(name, nn.Parameter(torch.ones(shape, device=dev), requires_grad=True))

    # ------------------------------------------------------------------ forward (:58-122)
    def forward(self, *args):
        if self.sys_param["mode"] == 0:
            return self._forward_train(*args)
        return self._forward_demo(args[0] if len(args) == 1 else args)

    def _forward_train(self, data, epoch, epoch_type, cur_ratio):
        cam = int(data[1].reshape(-1)[0])          # read the camera id on the host side, before the H2D copy
        images = data[0] if isinstance(data[0], DeviceImageSet) else None      # device-resident uint8 images (row f3)
        # (the image index stays on the host: it is only ever used as a python index)
        gt_rgbs, _, intr_wpts, intr_pts, extr_wpts, extr_pts = [
            t if isinstance(t, DeviceImageSet) or i == 1 else t.to(self.device) for i, t in enumerate(data)]
        loss_dict = {}
        emb = self.nerf.emmbedding_xyz
        if epoch_type == "CAM_PARAM_EPOCH":
            emb.barf_mode = False
            self.intr_adj, self.pose_adj, self.calib_pose_adj = self.add_weights2param(True, True, True, intr_wpts, extr_wpts)
            loss_dict["intr"] = [self._reproject(intr_wpts, self.intr_adj, self.calib_pose_adj, 0), intr_pts]
            loss_dict["extr"] = [self._reproject(extr_wpts, self.intr_adj, self.pose_adj, 1), extr_pts]
            self.opt_idx = 0
        else:
            joint = epoch_type == "GLOBAL_OPTIM_EPOCH"
            emb.barf_mode = joint                                   # :74 / :86
            self.intr_adj, self.pose_adj, self.calib_pose_adj = self.add_weights2param(True, joint, True, intr_wpts, None)
            loss_dict["intr"] = [self._reproject(intr_wpts, self.intr_adj, self.calib_pose_adj, 0), intr_pts]
            kinv = self.intr_inv_adj[cam] if self.intr_inv_adj is not None else \
                self.inverse_intrinsic(self.intr_adj[cam:cam + 1])[0]
            # pixel subset first (same device randperm as :329), rays only for those pixels
            rand_idx = self.sample_pixels(self.img_h * self.img_w)
            rays_d, rays_o = RaygenFn.apply(self.pose_adj[cam], kinv, rand_idx, self.img_w)
            rgbs_c, rgbs_f = self.nerf(rays_d, rays_o, epoch, cur_ratio if joint else 1)
            gt = images.gather(cam, rand_idx) if images is not None else gt_rgbs.reshape(-1, 3)[rand_idx]
            loss_dict["rgb"] = [rgbs_c, rgbs_f, gt]
            self.opt_idx = 1 if joint else 2
        # validation rays of the same index, every step, as the reference (:97-99)
        with torch.no_grad():
            rays_dv, rays_ov = self.get_rays(self.valid_pose, cam, self.intr_val_inv)
            rays_valid = [rays_dv, rays_ov, self.valid_rgbs[cam:cam + 1].detach()]
        intr_show = [self.intr_train.detach(), self.intr_adj.detach()]
        pose_show = [self.gt_pose.detach(), self.pose_adj.detach()]
        self.last_epoch_type = epoch_type
        return loss_dict, intr_show, pose_show, rays_valid

    def sample_pixels(self, npix):
        """The step's pixel subset, randperm(npix)[:batch] of the reference (:329): on the GPU one kernel
        (``ops.sample_perm``, keyed from torch's device generator); parity tests replace this method to inject the
        reference's draw."""
        if self.weights_pose.is_cuda:
            return ops.sample_perm(npix, min(self.batch, npix), self.weights_pose.device)
        return torch.randperm(npix, device=self.device)[: self.batch]

    @torch.no_grad()
    def render_image_device(self, img_id):
        """The demo render of one test camera (:106-122) with the result left on the device."""
        rays_d, rays_o = self.get_rays(self.test_pose, img_id, self.intr_test_inv)
        return self.nerf.render_chunked(rays_d, rays_o, self.nerf.nerf_coarse, self.nerf.nerf_fine, chunk=self.batch)

    @torch.no_grad()
    def _forward_demo(self, img_id):
        rgb, depth, opacity = self.render_image_device(img_id)
        return rgb.cpu(), depth.cpu(), opacity.cpu()

    # ------------------------------------------------------------------ rays (:124-145, 327-345)
    def get_rays(self, pose, img_id, intr_inv):
        """All H*W rays of camera ``img_id`` (row-major pixel centres), differentiable wrt the
        selected pose and inverse intrinsics."""
        cam = int(torch.as_tensor(img_id).reshape(-1)[0])
        if getattr(self, "_all_pix", None) is None:
            self._all_pix = torch.arange(self.img_h * self.img_w, device=self.device)
        return RaygenFn.apply(pose[cam].to(self.device), intr_inv[cam].to(self.device), self._all_pix, self.img_w)

    def generate_rand_rays(self, rays_d, rays_o, rand=True):
        if rand:
            idx = torch.randperm(rays_d.shape[0], device=rays_d.device)[: self.batch]
        else:
            start = (self.count_rays * self.batch) % rays_d.shape[0]
            idx = torch.arange(start, min(start + self.batch, rays_d.shape[0]), device=rays_d.device)
        self.count_rays += 1
        return rays_d[idx], rays_o[idx], idx

    # ------------------------------------------------------------------ camera parametrisation (:155-210, 269-316)
    # On the GPU all six parameter tensors go through ONE fused kernel each way (CameraFn, csrc/camera.hip;
    # SURVEY.md 8f row f1).  The per-piece torch methods below restate the same maths; they serve host tensors
    # (the CPU plumbing tests of the camera-only stage) and API compatibility.
    def add_weights2param(self, intr=True, extr=True, calib_extr=False, intr_wpts=None, extr_wpts=None):
        """K, pose, calib pose of all cameras.  On the GPU the fused kernel also projects the calibration points
        (`intr_wpts` through (K, calib pose), `extr_wpts` through (K, pose); [1,C,P,3] as the loader delivers them, :374-386):
        the pixels are kept in `self._reproj` for `_reproject`, so the whole camera preamble is one launch each way."""
        self.intr_inv_adj = None
        self._reproj = [None, None]
        if self.weights_pose.is_cuda:
            det = lambda t, on: t if on else t.detach()
            pts = [None if t is None else t.reshape(-1, t.shape[-2], 3) for t in (intr_wpts, extr_wpts)]
            K, Kinv, pose, calib, pi, pe = CameraFn.apply(det(self.weights_pose, extr), det(self.weights_pose_intr, calib_extr),
                                                          det(self.weights_fx, intr), det(self.weights_fy, intr),
                                                          det(self.weights_ux, intr), det(self.weights_uy, intr),
                                                          self.img_h, self.img_w, pts[0], pts[1])
            self.intr_inv_adj = Kinv
            self._reproj = [None if pi is None else pi.reshape(intr_wpts.shape[:-1] + (2,)),
                            None if pe is None else pe.reshape(extr_wpts.shape[:-1] + (2,))]
            return K, pose, calib
        return (self.add_weights2intr(self.img_h, self.img_w, adj=intr), self.add_weights2pose(adj=extr),
                self.add_weights2calib_pose(adj=calib_extr))

    def add_weights2intr(self, img_h, img_w, adj=True):
        w = [self.weights_fx, self.weights_fy, self.weights_ux, self.weights_uy]
        if not adj:
            w = [t.detach() for t in w]
        C = self.train_numb
        K = torch.zeros(C, 3, 3, device=self.device)
        # note: the reference scales fy by the image WIDTH as well (:172-173)
        K[:, 0, 0] = torch.abs(float(img_w) * w[0])
        K[:, 1, 1] = torch.abs(float(img_w) * w[1])
        K[:, 0, 2] = torch.abs(float(img_w) / 2 * w[2])
        K[:, 1, 2] = torch.abs(float(img_h) / 2 * w[3])
        K[:, 2, 2] = 1.0
        return K

    def add_weights2pose(self, adj=True):
        return self.se3_to_SE3(self.weights_pose if adj else self.weights_pose.detach())

    def add_weights2calib_pose(self, adj=True):
        return self.se3_to_SE3(self.weights_pose_intr if adj else self.weights_pose_intr.detach())

    def inverse_intrinsic(self, intr_mats):
        return torch.linalg.inv(intr_mats)

    @staticmethod
    def _series(theta, first_factor, nth=10):
        """sum_i (-1)^i theta^(2i) / denom_i with denom_0 = first_factor and
        denom_i = denom_{i-1} * (2i+a)(2i+a+1): the Taylor series A, B, C of :291-316."""
        a = {1.0: 0, 2.0: 1, 6.0: 2}[first_factor]
        ans = torch.zeros_like(theta)
        denom = 1.0
        for i in range(nth + 1):
            if a == 0:
                if i > 0:
                    denom *= (2 * i) * (2 * i + 1)
            else:
                denom *= (2 * i + a) * (2 * i + a + 1)
            ans = ans + (-1) ** i * theta ** (2 * i) / denom
        return ans

    def se3_to_SE3(self, wu):
        w, u = wu[..., :3], wu[..., 3:]
        O = torch.zeros_like(w[..., 0])
        wx = torch.stack([torch.stack([O, -w[..., 2], w[..., 1]], -1),
                          torch.stack([w[..., 2], O, -w[..., 0]], -1),
                          torch.stack([-w[..., 1], w[..., 0], O], -1)], -2)
        theta = w.norm(dim=-1)[..., None, None]
        I = torch.eye(3, device=wu.device)
        A, B, Cc = self._series(theta, 1.0), self._series(theta, 2.0), self._series(theta, 6.0)
        R = I + A * wx + B * wx @ wx
        V = I + B * wx + Cc * wx @ wx
        return torch.cat([R, V @ u[..., None]], dim=-1)

    # ------------------------------------------------------------------ reprojection branch (:147-152, 236-267)
    def _reproject(self, tag_wpts, intr_adj, pose_adj, which):
        """The pixels the fused camera kernel already produced (GPU), else the tensor-op restatement."""
        got = getattr(self, "_reproj", [None, None])[which]
        return got if got is not None else self.get_reproject_pixels(tag_wpts, intr_adj, pose_adj)

    def get_reproject_pixels(self, tag_wpts, intr_adj, pose_adj):
        """Projects calibration points [B,C,P,3] through [R|t] and K -> pixel coords [B,C,P,2]."""
        ones = torch.ones_like(tag_wpts[..., :1])
        wh = torch.cat([tag_wpts, ones], dim=-1)                       # [B,C,P,4]
        cam = wh @ pose_adj.unsqueeze(0).transpose(-2, -1)             # [B,C,P,3]
        pix = cam @ intr_adj.unsqueeze(0).transpose(-2, -1)
        return pix[..., :2] / pix[..., 2:]

    # ------------------------------------------------------------------ reporting hooks used by main.py
    def show_estimate_param(self, intr_show, pose_show, epoch, epoch_type):
        intr_err = (intr_show[0] - intr_show[1]).abs()
        pose_err = (pose_show[0] - pose_show[1]).abs()
        logging.info("EPOCH {} |K err| fx {:.4f} fy {:.4f} ux {:.4f} uy {:.4f} | R {:.5f} T {:.5f}".format(
            epoch, float(intr_err[:, 0, 0].mean()), float(intr_err[:, 1, 1].mean()), float(intr_err[:, 0, 2].mean()),
            float(intr_err[:, 1, 2].mean()), float(pose_err[..., :3].mean()), float(pose_err[..., 3].mean())))

    def show_RT_est_results(self, epoch, epoch_type=None, mode="epoch", show_info=True):
        """Reference signature (:388); main.py:94 calls show_RT_est_results(epoch, epoch_type, mode='epoch').  The
        reference aligns the estimated poses to the ground truth and plots the camera frusta; here the alignment-free
        pose error is logged (plotting is outside the hot path)."""
        if show_info and hasattr(self, "pose_adj"):
            err = (self.gt_pose - self.pose_adj.detach()).abs()
            logging.info("EPOCH {} [{}] pose |R err| {:.5f} |t err| {:.5f}".format(epoch, epoch_type, float(err[..., :3].mean()),
                                                                                   float(err[..., 3].mean())))
