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


def refuse_autograd(what, tensors):
    """The stand-alone per-module forwards run HIP kernels that have no backward of their own (the differentiable path is
    NeRF_Model.render_rays_train / MC_Model.forward, one fused autograd.Function).  A caller that would differentiate
    through them (model/net_block.py:20-35, 67-78 are differentiable in the reference) gets an error, not a tensor
    without a graph."""
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise _lib.McnerfError(f"{what} is forward-only on the HIP path and was called with autograd recording on tensors that "
                               "require a gradient: wrap the call in torch.no_grad(), or train through NeRF_Model.render_rays_train / "
                               "MC_Model.forward (the fused differentiable render)")


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
        if self.n_freqs != 10:
            raise ValueError("the HIP kernels are built for emb_freqs_xyz = 10 (63 encoded channels)")
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

    def barf_weights_on(self, step_r, device) -> torch.Tensor:
        """``barf_weights`` as a device tensor; on a GPU the ten host values are passed as kernel arguments
        (``ops.upload_f32``) so the host never blocks on the stream."""
        w = self.barf_weights(step_r)
        return ops.upload_f32(w, device) if torch.device(device).type == "cuda" else w.to(device)

    def forward(self, x, step_r):
        """Reference :20-35: [..., 3] -> [..., 63].  Stand-alone, FORWARD-ONLY call of the HIP encoding kernel (the render
        path computes the encoding inside the fused MLP kernels and differentiates it there): asked for a gradient it
        raises instead of returning a silently detached tensor."""
        refuse_autograd("SinCosEmbedding.forward", [x])
        with torch.no_grad():
            return self._forward(x, step_r)

    def _forward(self, x, step_r):
        flat = x.reshape(-1, 3).float().contiguous()
        out = ops.encode(flat, self.barf_weights_on(step_r, flat.device))
        return out.reshape(*x.shape[:-1], self.out_channels)


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
        if self.deg != 2 or self.in_channels_xyz != 63:
            raise ValueError("the HIP kernels are built for MLP_deg = 2 and 63 encoded channels")
        key = "coarse" if type == "coarse" else "fine"
        self.depth = sys_params[f"{key}_MLP_depth"]
        self.width = sys_params[f"{key}_MLP_width"]
        self.skips = list(sys_params[f"{key}_MLP_skip"])
        skips = [s for s in self.skips if 0 < s < self.depth]
        if len(skips) > 1:
            raise ValueError("at most one skip layer is supported by the HIP kernels")
        self.net = ops.Net(self.depth, self.width, skips[0] if skips else -1)
        for i in range(self.depth):
            fan_in = self.net.in_features(i)
            setattr(self, f"xyz_encoding_{i+1}", nn.Sequential(nn.Linear(fan_in, self.width), nn.ReLU(True)))
        self.sigma = nn.Sequential(nn.Linear(self.width, self.width), nn.ReLU(True), nn.Linear(self.width, 1))
        self.sh = nn.Sequential(nn.Linear(self.width, self.width), nn.ReLU(True),
                                nn.Linear(self.width, 3 * (self.deg + 1) ** 2))
        self._flat = None
        self._offsets = None

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

    def forward(self, x, dirs):
        """Reference :67-78: encoded positions x [M,63] and view directions dirs [M,3] -> [M,4] = (sigma_raw, rgb).
        Stand-alone, FORWARD-ONLY call of the exact-fp32 fused kernel on caller-supplied encodings (training
        differentiates through the fused NeRF_Model.render_rays_train): with autograd recording and a parameter or input
        that requires a gradient it raises (wrap the call in torch.no_grad() for inference) instead of silently returning a
        detached tensor that would train nothing."""
        refuse_autograd("CorseFine_NeRF.forward", [x, dirs, *self.parameters()])
        with torch.no_grad():
            return self._forward(x, dirs)

    def _forward(self, x, dirs):
        flat = self.flat_params()
        packed = ops.pack_weights(self.net, flat, precision="f32")
        return ops.mlp_apply(self.net, flat, packed, x.float().contiguous(), dirs.float().contiguous())
