"""Network blocks of the MI355X-native MC-NeRF path.

Mirrors the reference's ``model/net_block.py`` API (``SinCosEmbedding``, ``CorseFine_NeRF``): same
constructor arguments, same sub-module / parameter names (so state dicts interchange),
same default ``nn.Linear`` initialisation order.  Unlike the reference these modules do not run
eager ATen ops: the positional encoding, the MLP, the SH colour and the sigmoid all execute inside
the fused HIP kernels (``mcnerf_mlp_fwd`` / ``mcnerf_mlp_bwd`` / ``mcnerf_mlp_dw``); the modules are
parameter containers that keep every tensor of a net inside ONE flat fp32 buffer (coalesced weight
packing, one RCCL all-reduce per step).
"""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn as nn

from .. import _lib, ops


class EncodeFn(torch.autograd.Function):
    """SinCosEmbedding.forward (model/net_block.py:20-35) as a differentiable stand-alone call: mcnerf_encode / mcnerf_encode_bwd."""

    @staticmethod
    def forward(ctx, x, barf_w):
        flat = x.reshape(-1, 3).float().contiguous()
        ctx.save_for_backward(flat, barf_w)
        ctx.shape = x.shape
        return ops.encode(flat, barf_w).reshape(*x.shape[:-1], 3 + 6 * barf_w.numel())

    @staticmethod
    def backward(ctx, d_out):
        flat, barf_w = ctx.saved_tensors
        d_x = ops.encode_bwd(flat, barf_w, d_out.reshape(-1, 3 + 6 * barf_w.numel()).float().contiguous())
        return d_x.reshape(ctx.shape), None


class MlpApplyFn(torch.autograd.Function):
    """CorseFine_NeRF.forward (model/net_block.py:67-78) as a differentiable stand-alone call on caller-supplied encodings:
    the exact-fp32 fused forward with saved operands, its backward chain (d x, d dirs) and the weight-gradient kernel.  The
    parameters enter as ONE flat buffer (`module.flat_params()`); its gradient comes back flat and autograd slices it into the
    per-parameter views."""

    @staticmethod
    def forward(ctx, module, x, dirs, flat):
        net = module.net
        packed = ops.pack_weights(net, flat, precision="f32")
        xe, dd = x.reshape(-1, net.n_enc).float().contiguous(), dirs.reshape(-1, 3).float().contiguous()
        out, save = ops.mlp_apply_save(net, flat, packed, xe, dd)
        ctx.net, ctx.save, ctx.packed = net, save, packed
        ctx.save_for_backward(flat, dd, out)
        ctx.shapes = (x.shape, dirs.shape)
        return out.reshape(*x.shape[:-1], 4)

    @staticmethod
    def backward(ctx, d_out):
        flat, dd, out = ctx.saved_tensors
        grads = torch.zeros_like(flat) if ctx.needs_input_grad[3] else None
        # (ctx.save stays until autograd frees the node: a second backward through a retained graph works like the reference's)
        d_x, d_dirs = ops.mlp_apply_bwd(ctx.net, flat, ctx.packed, dd, out, d_out.reshape(-1, 4).float().contiguous(), ctx.save, grads)
        return (None, d_x.reshape(ctx.shapes[0]) if ctx.needs_input_grad[1] else None,
                d_dirs.reshape(ctx.shapes[1]) if ctx.needs_input_grad[2] else None, grads)


class SinCosEmbedding(nn.Module):
    """Frequency encoding description (reference: model/net_block.py:6-35).

    Holds ``n_freqs`` and the BARF schedule; the encoding itself is computed in the prologue of the
    fused MLP kernels.  ``barf_mode`` is toggled per training stage by ``MC_Model.forward`` exactly as
    in the reference (model/mc_nerf.py:65, 74, 86).
    """

    def __init__(self, sys_params):
        super().__init__()
        self.sys_param = sys_params
        self.device = sys_params["device_type"]
        self.n_freqs = sys_params["emb_freqs_xyz"]
        if not 1 <= self.n_freqs <= 10:
            raise ValueError("1 .. 10 encoding frequencies are built (emb_freqs_xyz; 64-column encoded-input tiles)")
        self.barf_mode = sys_params["barf_mask"]
        self.barf_start = sys_params["barf_start"]
        self.barf_end = sys_params["barf_end"]
        self.in_channels = 3
        self.out_channels = self.in_channels * (2 * self.n_freqs + 1)

    def barf_weights(self, step_r) -> torch.Tensor:
        """Per-frequency mask w_k = (1 - cos(pi * clamp(alpha - k, 0, 1))) / 2 (model/net_block.py:26-29),
        evaluated in fp32 on the host (10 values); all ones when BARF is off."""
        L = self.n_freqs
        if not self.barf_mode:
            return torch.ones(L, dtype=torch.float32)
        alpha = (step_r - self.barf_start) / (self.barf_end - self.barf_start) * L
        k = torch.arange(L, dtype=torch.float32)
        return (1.0 - torch.cos(torch.clamp(alpha - k, 0.0, 1.0) * math.pi)) / 2.0

    def barf_weights_on(self, step_r, device, pad: int = 0) -> torch.Tensor:
        """``barf_weights`` as a device tensor; on a GPU the ten host values are passed as kernel arguments
        (``ops.upload_f32``) so the host never blocks on the stream.  ``pad``: zero-extended to that many values (the fused MLP
        kernels of the register-chain modes read ten, whatever ``emb_freqs_xyz`` is: a net with fewer frequencies has zero
        weights on the channels of the others)."""
        w = self.barf_weights(step_r)
        if pad > w.numel():
            w = torch.cat([w, torch.zeros(pad - w.numel())])
        return ops.upload_f32(w, device) if torch.device(device).type == "cuda" else w.to(device)

    def forward(self, x, step_r):
        """Reference :20-35: [..., 3] -> [..., 63].  Stand-alone call of the HIP encoding kernel (the render path computes the
        encoding inside the fused MLP kernels and differentiates it there); differentiable with respect to x like the
        reference's (EncodeFn)."""
        bw = self.barf_weights_on(step_r, x.device)
        if torch.is_grad_enabled() and x.requires_grad:
            return EncodeFn.apply(x, bw)
        flat = x.detach().reshape(-1, 3).float().contiguous()
        return ops.encode(flat, bw).reshape(*x.shape[:-1], self.out_channels)


class CorseFine_NeRF(nn.Module):
    """Parameter container of one NeRF MLP (reference: model/net_block.py:37-78).

    Sub-modules are created exactly like the reference's (``xyz_encoding_{i}`` = Linear+ReLU,
    ``sigma`` / ``sh`` = Linear-ReLU-Linear) so parameter names, shapes and the default init stream
    are identical; afterwards every parameter's storage is re-pointed into ``flat`` (16-byte aligned
    slots in the reference's state-dict order, layout from ``mcnerf_param_offsets``).
    """

    def __init__(self, sys_params, type="coarse"):
        super().__init__()
        self.in_channels_xyz = 3 * (2 * sys_params["emb_freqs_xyz"] + 1)
        self.deg = sys_params["MLP_deg"]
        self.n_freqs = int(sys_params["emb_freqs_xyz"])
        if not 1 <= self.n_freqs <= 10:
            raise ValueError("1 .. 10 encoding frequencies are built (emb_freqs_xyz; 64-column encoded-input tiles)")
        if not 0 <= self.deg <= 3:
            raise ValueError("SH degrees 0 .. 3 are built (MLP_deg)")
        key = "coarse" if type == "coarse" else "fine"
        self.depth = sys_params[f"{key}_MLP_depth"]
        self.width = sys_params[f"{key}_MLP_width"]
        self.skips = list(sys_params[f"{key}_MLP_skip"])
        # any `skips` list (reference :45, 55-58), SH degree 0 .. 3 (:43, 75-76) and 1 .. 10 encoding frequencies (:11-18); more than one
        # skip layer or degree 3 run on the exact-fp32 kernel family only (NeRF_Model refuses the register-chain precision modes
        # for such a net)
        self.net = ops.Net(self.depth, self.width, ops.skip_code(self.skips, self.depth, self.deg, self.n_freqs))
        for i in range(self.depth):
            fan_in = self.net.in_features(i)
            setattr(self, f"xyz_encoding_{i+1}", nn.Sequential(nn.Linear(fan_in, self.width), nn.ReLU(True)))
        self.sigma = nn.Sequential(nn.Linear(self.width, self.width), nn.ReLU(True), nn.Linear(self.width, 1))
        self.sh = nn.Sequential(nn.Linear(self.width, self.width), nn.ReLU(True),
                                nn.Linear(self.width, 3 * (self.deg + 1) ** 2))
        self._flat = None
        self._offsets = None
        self._range_flags = {}
        self.net_type = key

    def weight_names(self) -> List[str]:
        """The weight tensors that go through the 16-bit packing, in stream order (csrc/mcnerf_16.h: mcn16_fwd_stream)."""
        return [f"xyz_encoding_{i + 1}.0.weight" for i in range(self.depth)] + ["sigma.0.weight", "sh.0.weight", "sh.2.weight"]

    def range_flags(self, precision: str, device):
        """Sticky per-tensor "weight outside the operand range of `precision`" words, filled by the packing kernel (None in
        f32, which has no range to exceed): what RAdam.raise_on_overflow() names when the overflow guard has refused steps."""
        if not ops.is16(precision):
            return None
        key = (precision, str(device))
        if key not in self._range_flags:
            f = torch.zeros(self.depth + 3, dtype=torch.int32, device=device)
            self._range_flags[key] = f
            ops.range_watch_register(f, f"{self.net_type} net", self.weight_names(), precision)
        return self._range_flags[key]

    # ------------------------------------------------------------------ flat storage
    def ordered_parameters(self) -> List[nn.Parameter]:
        """Parameters in the reference's state-dict order (== registration order)."""
        return list(self.parameters())

    def _aliased(self) -> bool:
        if self._flat is None:
            return False
        base = self._flat.data_ptr()
        for p, off in zip(self.ordered_parameters(), self._offsets):
            if p.device != self._flat.device or p.data_ptr() != base + 4 * off:
                return False
        return True

    def flat_params(self) -> torch.Tensor:
        """The flat fp32 buffer all parameters live in (rebuilt if ``.to()`` / ``load`` moved them)."""
        if not self._aliased():
            params = self.ordered_parameters()
            dev = params[0].device
            if self._offsets is None:
                self._offsets = ops.param_offsets(self.net)
            flat = torch.zeros(ops.param_count(self.net), dtype=torch.float32, device=dev)
            for p, off in zip(params, self._offsets):
                n = p.numel()
                flat[off:off + n].copy_(p.data.reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
            self._flat = flat
        return self._flat

    def grad_views(self, flat_grad: torch.Tensor):
        """Per-parameter views of a flat gradient buffer (same layout as ``flat_params``)."""
        return [flat_grad[off:off + p.numel()].view(p.shape)
                for p, off in zip(self.ordered_parameters(), self._offsets)]

    def flat_params_autograd(self) -> torch.Tensor:
        """The flat parameter buffer as a node of the autograd graph: differentiating through it delivers the flat gradient to
        every parameter's .grad (per-parameter views), so a stand-alone differentiable forward trains like the reference's."""
        flat = self.flat_params()
        params = self.ordered_parameters()
        return _FlatOf.apply(self, flat, *params)

    def forward(self, x, dirs):
        """Reference :67-78: encoded positions x [M,63] and view directions dirs [M,3] -> [M,4] = (sigma_raw, rgb).
        Stand-alone call of the exact-fp32 fused kernel on caller-supplied encodings (training normally differentiates
        through the fused NeRF_Model.render_rays_train).  Differentiable like the reference's with respect to x, dirs and
        every parameter (MlpApplyFn); without a gradient request it is the forward-only kernel."""
        needs = torch.is_grad_enabled() and (x.requires_grad or dirs.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs:
            return MlpApplyFn.apply(self, x, dirs, self.flat_params_autograd())
        flat = self.flat_params()
        packed = ops.pack_weights(self.net, flat, precision="f32")
        return ops.mlp_apply(self.net, flat, packed, x.detach().reshape(-1, self.net.n_enc).float().contiguous(),
                             dirs.detach().reshape(-1, 3).float().contiguous()).reshape(*x.shape[:-1], 4)


class _FlatOf(torch.autograd.Function):
    """Identity on the flat buffer whose backward hands each parameter its slice of the flat gradient."""

    @staticmethod
    def forward(ctx, module, flat, *params):
        ctx.module = module
        return flat.detach().view_as(flat)

    @staticmethod
    def backward(ctx, g):
        views = ctx.module.grad_views(g)
        params = ctx.module.ordered_parameters()
        return (None, None, *[v.clone() if p.requires_grad else None for v, p in zip(views, params)])
