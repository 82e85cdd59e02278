"""Tensor-level wrappers over the C ABI (include/mcnerf.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every function below
checks its tensors (device, dtype, contiguity), passes raw pointers to libmcnerf.so and returns
tensors.  All launches go to ``torch.cuda.current_stream()``.  There is no CPU path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib

Tensor = torch.Tensor


SKIP_MASK = 256          # MCN_SKIP_MASK: skip >= 256 = (bit mask of skip layers) << 8 (include/mcnerf.h)


def skip_code(skips, depth: int, deg: int = 2, n_freqs: int = 10) -> int:
    """The reference's `skips` list, `MLP_deg` and `emb_freqs_xyz` (model/net_block.py:11-18, 43-45) -> the C ABI's `skip` argument:
    -1 (no skip layer), the layer index (one), or the mask form -- (bit mask of skip layers | 1) << 8, with 0x80 | deg << 4 in the low
    byte when the SH degree is not 2 and n_freqs + 1 in its low nibble when that is not 10 (several skip layers, another degree or
    frequency count: exact-fp32 kernels only)."""
    ks = sorted({int(k) for k in skips if 0 < int(k) < depth})
    if not 0 <= deg <= 3:
        raise ValueError("SH degrees 0 .. 3 are built")
    if not 1 <= n_freqs <= 10:
        raise ValueError("1 .. 10 encoding frequencies are built (64-column encoded-input tiles)")
    if deg == 2 and n_freqs == 10 and not ks:
        return -1
    if deg == 2 and n_freqs == 10 and len(ks) == 1:
        return ks[0]
    return ((sum(1 << k for k in ks) | 1) << 8) | ((0x80 | (deg << 4)) if deg != 2 else 0) | ((n_freqs + 1) if n_freqs != 10 else 0)


@dataclass(frozen=True)
class Net:
    """(depth, width, skip) of one CorseFine_NeRF (model/net_block.py:40-49 in the reference); `skip` in the C ABI's encoding
    (`skip_code`)."""
    depth: int
    width: int
    skip: int

    @property
    def triple(self):
        return (self.depth, self.width, self.skip)

    @property
    def skips(self):
        if self.skip >= SKIP_MASK:
            return [l for l in range(1, self.depth) if (self.skip >> 8) >> l & 1]
        return [self.skip] if 0 < self.skip < self.depth else []

    @property
    def multi_skip(self) -> bool:
        return len(self.skips) > 1

    @property
    def deg(self) -> int:
        return (self.skip >> 4) & 7 if self.skip >= SKIP_MASK and self.skip & 0x80 else 2

    @property
    def n_sh(self) -> int:
        return 3 * (self.deg + 1) ** 2

    @property
    def n_shp(self) -> int:
        return 32 if self.n_sh < 32 else 64

    @property
    def n_freqs(self) -> int:
        return (self.skip & 15) - 1 if self.skip >= SKIP_MASK and self.skip & 15 else 10

    @property
    def n_enc(self) -> int:
        return 3 + 6 * self.n_freqs

    @property
    def fp32_only(self) -> bool:
        """Topologies only the exact-fp32 kernel family takes: more than one skip layer, SH degree 3.  (The register-chain modes have
        the geometry of one skip layer, degree 2 and 10 frequencies; a net with fewer frequencies or a lower degree is scattered
        into it when its weights are packed, csrc/mcnerf_common.h.)"""
        return self.multi_skip or self.deg > 2

    def in_features(self, i: int) -> int:
        if i == 0:
            return self.n_enc
        return self.width + self.n_enc if i in self.skips else self.width

    def shapes(self):
        """Tensor shapes in the reference's state-dict order."""
        s = []
        for i in range(self.depth):
            s += [(self.width, self.in_features(i)), (self.width,)]
        s += [(self.width, self.width), (self.width,), (1, self.width), (1,),
              (self.width, self.width), (self.width,), (self.n_sh, self.width), (self.n_sh,)]
        return s

    def names(self):
        n = []
        for i in range(self.depth):
            n += [f"xyz_encoding_{i+1}.0.weight", f"xyz_encoding_{i+1}.0.bias"]
        n += ["sigma.0.weight", "sigma.0.bias", "sigma.2.weight", "sigma.2.bias",
              "sh.0.weight", "sh.0.bias", "sh.2.weight", "sh.2.bias"]
        return n


def param_count(net: Net) -> int:
    return int(_lib.lib().mcnerf_param_count(*net.triple))


def packed_count(net: Net) -> int:
    return int(_lib.lib().mcnerf_packed_count(*net.triple))


def param_offsets(net: Net):
    import ctypes
    n = 2 * net.depth + 8
    arr = (ctypes.c_longlong * n)()
    _lib.call("mcnerf_param_offsets", *net.triple, ctypes.cast(arr, ctypes.c_void_p))
    return [int(v) for v in arr]


def tile_rows(width: int) -> int:
    return int(_lib.lib().mcnerf_tile_rows(width))


# --------------------------------------------------------------------------- helpers
def _p(t: Optional[Tensor], dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.McnerfError("mc_nerf_amd ops need CUDA/HIP tensors (no CPU fallback)")
    if t.dtype != dtype:
        raise _lib.McnerfError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.McnerfError("tensor must be contiguous")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def flatten_params(net: Net, tensors, device) -> Tensor:
    """Copies per-tensor parameters (reference order) into a fresh flat buffer."""
    flat = torch.zeros(param_count(net), dtype=torch.float32, device=device)
    for off, shp, t in zip(param_offsets(net), net.shapes(), tensors):
        n = 1
        for d in shp:
            n *= d
        flat[off:off + n].copy_(t.reshape(-1))
    return flat


# --------------------------------------------------------------------------- ops
PRECISIONS = ("f32", "f16x3", "f16x3h", "f16", "bf16")
# register-chain modes: single-pass 16-bit MFMA (csrc/mcnerf_16.h) and split-f16 "f16x3" (csrc/mcnerf_x3.h): name -> dtype
# code of the C ABI
# "f16x3h": the split-f16 forward / backward chains (colours and dX as in "f16x3") saving only the hi plane of every operand; the
# weight gradient is the single-pass f16 kernel on those planes (dW operands rounded to 11 bits; 2-byte saved operands)
DTYPE16 = {"f16": 0, "bf16": 1, "f16x3": 2, "f16x3h": 3}


def is16(precision: str) -> bool:
    return precision in DTYPE16


def packed16_split(net: Net, packed: Tensor, precision: str):
    """The register-chain modes keep both fragment streams of a net in ONE byte tensor: (forward stream, backward stream) views."""
    nf = int(_lib.lib().mcnerf_packed_bytes_16(*net.triple, DTYPE16[precision], 0))
    return packed[:nf], packed[nf:]


def pack_weights(net: Net, params: Tensor, packed=None, precision: str = "f32", range_flags: Optional[Tensor] = None):
    """Packed weights for the given precision mode: one fp32-sized buffer (f32: fp32 fragments, f16x3: split-f16
    fragments) or, in the 16-bit modes, one byte tensor holding the forward and the backward fragment stream.
    `range_flags` (int32 [depth + 3], 16-bit modes): sticky per-tensor "a weight is outside the mode's operand range" words
    (include/mcnerf.h: mcnerf_pack_weights_16)."""
    assert precision in PRECISIONS
    if is16(precision):
        if packed is None:
            l = _lib.lib()
            dt = DTYPE16[precision]
            packed = torch.empty(int(l.mcnerf_packed_bytes_16(*net.triple, dt, 0)) + int(l.mcnerf_packed_bytes_16(*net.triple, dt, 1)),
                                 dtype=torch.uint8, device=params.device)
        pf, pb = packed16_split(net, packed, precision)
        _lib.call("mcnerf_pack_weights_16", *net.triple, _p(params), _p(pf, torch.uint8), _p(pb, torch.uint8), DTYPE16[precision],
                  _p(range_flags, torch.int32), _stream())
        return packed
    if packed is None:
        packed = torch.empty(packed_count(net), dtype=torch.float32, device=params.device)
    _lib.call("mcnerf_pack_weights", *net.triple, _p(params), _p(packed), _stream())
    return packed


# ---- range watch of the reduced-precision modes: (flags tensor, owner label, tensor names, precision) per net that has packed weights
# with `range_flags`; read (one synchronisation) only when somebody asks why optimiser steps were refused
_RANGE_WATCH = []
RANGE_TEXT = {"f16": "|w| <= 65504", "bf16": "finite weights", "f16x3": "|w| <= 255.9 (the weights are scaled by 2^8 into f16)",
              "f16x3h": "|w| <= 255.9 (the weights are scaled by 2^8 into f16)"}


def range_watch_register(flags: Tensor, label: str, names, precision: str):
    import weakref
    _RANGE_WATCH[:] = [w for w in _RANGE_WATCH if w[0]() is not None]         # (models come and go: drop the words of dead ones)
    _RANGE_WATCH.append((weakref.ref(flags), label, list(names), precision))


def range_report():
    """[(label, tensor name, precision)] of every watched weight tensor that has held a value outside its mode's operand range."""
    out = []
    for ref, label, names, precision in list(_RANGE_WATCH):
        flags = ref()
        if flags is None:
            continue
        for i in torch.nonzero(flags.cpu()).reshape(-1).tolist():
            out.append((label, names[i], precision))
    return out


def raygen_fwd(pose: Tensor, kinv: Tensor, pix: Tensor, W: int) -> Tuple[Tensor, Tensor]:
    n = pix.numel()
    d = torch.empty(n, 3, dtype=torch.float32, device=pix.device)
    o = torch.empty_like(d)
    _lib.call("mcnerf_raygen_fwd", _p(pose), _p(kinv), _p(pix, torch.int64), n, W, _p(d), _p(o), _stream())
    return d, o


def raygen_bwd(pose: Tensor, kinv: Tensor, pix: Tensor, W: int, d_d: Tensor, d_o: Tensor) -> Tuple[Tensor, Tensor]:
    z = torch.zeros(24, dtype=torch.float32, device=pix.device)          # (one fill for both accumulation targets)
    d_pose, d_kinv = z[:12].view(3, 4), z[12:21].view(3, 3)
    _lib.call("mcnerf_raygen_bwd", _p(pose), _p(kinv), _p(pix, torch.int64), pix.numel(), W,
              _p(d_d), _p(d_o), _p(d_pose), _p(d_kinv), _stream())
    return d_pose, d_kinv


@dataclass
class MlpSave:
    """Workspaces written by mlp_fwd(save=True) and consumed by mlp_bwd / mlp_dw.  In the 16-bit modes `act`, `enc`
    `sh` and `mask` are byte / word buffers in the fragment-major layouts of csrc/mcnerf_16.h."""
    capacity: int
    act: Tensor
    enc: Tensor
    sh: Optional[Tensor]
    mask: Tensor


def ws_bytes_16(net: Net, capacity: int, which: int, precision: str = "f16") -> int:
    return int(_lib.lib().mcnerf_ws_bytes_16(net.depth, net.width, DTYPE16[precision], int(capacity), which))


def alloc_save(net: Net, capacity: int, device, precision: str = "f32") -> MlpSave:
    capacity = max(int(capacity), 1)
    if is16(precision):
        return MlpSave(capacity,
                       torch.empty(ws_bytes_16(net, capacity, 0, precision), dtype=torch.uint8, device=device),
                       torch.empty(ws_bytes_16(net, capacity, 1, precision), dtype=torch.uint8, device=device),
                       torch.empty(ws_bytes_16(net, capacity, 4, precision), dtype=torch.uint8, device=device),
                       torch.empty(ws_bytes_16(net, capacity, 2, precision) // 4, dtype=torch.int32, device=device))
    return MlpSave(capacity,
                   torch.empty((net.depth + 2) * capacity * net.width, dtype=torch.float32, device=device),
                   torch.empty(capacity * 64, dtype=torch.float32, device=device),
                   torch.empty(capacity * net.n_shp, dtype=torch.float32, device=device),
                   torch.empty((net.depth + 2) * capacity * (net.width // 32), dtype=torch.int32, device=device))


def mlp_fwd(net: Net, params: Tensor, packed: Tensor, rays_o: Tensor, rays_d: Tensor, zgrid: Tensor,
            jitter: Optional[Tensor], barf_w: Tensor, out: Tensor, idx: Optional[Tensor] = None,
            count: Optional[Tensor] = None, max_rows: int = 0, save: Optional[MlpSave] = None,
            precision: str = "f32") -> None:
    n_rays, S = rays_d.shape[0], zgrid.numel()
    assert out.numel() == n_rays * S * 4
    if is16(precision):
        _lib.call("mcnerf_mlp_fwd_16", *net.triple, DTYPE16[precision], _p(params), _p(packed16_split(net, packed, precision)[0], torch.uint8), _p(rays_o), _p(rays_d),
                  _p(zgrid), _p(jitter), _p(barf_w), _p(idx, torch.int32), _p(count, torch.int32), int(max_rows), n_rays, S,
                  _p(out), _p(save.act, torch.uint8) if save else None, save.capacity if save else 0,
                  _p(save.enc, torch.uint8) if save else None, _p(save.mask, torch.int32) if save else None,
                  _p(save.sh, torch.uint8) if save else None, _stream())
        return
    _lib.call("mcnerf_mlp_fwd", *net.triple, _p(params), _p(packed), _p(rays_o), _p(rays_d), _p(zgrid),
              _p(jitter), _p(barf_w), _p(idx, torch.int32), _p(count, torch.int32), int(max_rows), n_rays, S,
              _p(out), _p(save.act) if save else None, save.capacity if save else 0,
              _p(save.enc) if save else None, _p(save.sh) if save else None,
              _p(save.mask, torch.int32) if save else None, _stream())


def sample_perm(n: int, batch: int, device, seed: Optional[Tensor] = None) -> Tensor:
    """randperm(n)[:batch] (model/mc_nerf.py:329) in one kernel: `batch` distinct uniformly random ids of [0, n) in random
    order (int64); the key is a device word drawn from torch's device generator, so torch.manual_seed reproduces it."""
    if seed is None:
        seed = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int32, device=device)
    out = torch.empty(batch, dtype=torch.int64, device=device)
    _lib.call("mcnerf_sample_perm", _p(out, torch.int64), int(n), int(batch), _p(seed, torch.int32), _stream())
    return out


def upload_f32(host_vals: Tensor, device) -> Tensor:
    """A small fp32 host tensor (<= 16 values) as a fresh device tensor, stream-ordered and without a host-device copy
    (the values travel as kernel arguments): the host never waits for the kernels already queued."""
    hv = host_vals.detach().reshape(-1).float().contiguous()
    assert hv.device.type == "cpu" and hv.numel() <= 16
    out = torch.empty(hv.numel(), dtype=torch.float32, device=device)
    _lib.call("mcnerf_upload_f32", _p(out), hv.data_ptr(), hv.numel(), _stream())
    return out


def encode(x: Tensor, barf_w: Tensor) -> Tensor:
    """SinCosEmbedding.forward (model/net_block.py:20-35): [n,3] -> [n, 3 + 6 F] with F = len(barf_w) frequencies (10: 63)."""
    n, F = x.shape[0], barf_w.numel()
    out = torch.empty(n, 3 + 6 * F, dtype=torch.float32, device=x.device)
    _lib.call("mcnerf_encode", _p(x), _p(barf_w), n, F, _p(out), _stream())
    return out


def mlp_apply(net: Net, params: Tensor, packed: Tensor, x_enc: Tensor, dirs: Tensor) -> Tensor:
    """CorseFine_NeRF.forward (model/net_block.py:67-78) on given encodings: [n,63], [n,3] -> [n,4] (exact-fp32 kernel)."""
    n = x_enc.shape[0]
    out = torch.empty(n, 4, dtype=torch.float32, device=x_enc.device)
    _lib.call("mcnerf_mlp_apply", *net.triple, _p(params), _p(packed), _p(x_enc), _p(dirs), n, _p(out), _stream())
    return out


def sync_finish(arena: Tensor, n_grad: int, world: int, local_flags: Tensor, asym: Tensor) -> None:
    """arena[:n_grad] /= world; asym += any(arena[n_grad:] != world * local_flags) -- the finish of FlatGradSync's one all-reduce
    in one launch."""
    _lib.call("mcnerf_sync_finish", _p(arena), int(n_grad), int(arena.numel() - n_grad), int(world), _p(local_flags),
              _p(asym, torch.int32), _stream())


def encode_bwd(x: Tensor, barf_w: Tensor, d_out: Tensor) -> Tensor:
    """Backward of `encode`: d_out [n, 3 + 6 F] -> d_x [n,3]."""
    n = x.shape[0]
    d_x = torch.empty(n, 3, dtype=torch.float32, device=x.device)
    _lib.call("mcnerf_encode_bwd", _p(x), _p(barf_w), n, barf_w.numel(), _p(d_out), _p(d_x), _stream())
    return d_x


def mlp_apply_save(net: Net, params: Tensor, packed: Tensor, x_enc: Tensor, dirs: Tensor) -> Tuple[Tensor, MlpSave]:
    """`mlp_apply` that keeps the operands of its backward (exact-fp32 workspaces)."""
    n = x_enc.shape[0]
    out = torch.empty(n, 4, dtype=torch.float32, device=x_enc.device)
    save = alloc_save(net, n, x_enc.device, "f32")
    _lib.call("mcnerf_mlp_apply_save", *net.triple, _p(params), _p(packed), _p(x_enc), _p(dirs), n, _p(out),
              _p(save.act), save.capacity, _p(save.enc), _p(save.sh), _p(save.mask, torch.int32), _stream())
    return out, save


def mlp_apply_bwd(net: Net, params: Tensor, packed: Tensor, dirs: Tensor, out: Tensor, d_out: Tensor, save: MlpSave,
                  grads: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """Backward of `mlp_apply_save`: -> (d_x_enc [n,63], d_dirs [n,3]); the parameter gradients are ACCUMULATED into the flat
    `grads` (layout of the parameter buffer) unless it is None."""
    n = dirs.shape[0]
    dev = dirs.device
    dy, dsh = torch.empty_like(save.act), torch.empty_like(save.sh)
    d_x = torch.empty(n, net.n_enc, dtype=torch.float32, device=dev)
    z = torch.zeros(n * 3 + 1, dtype=torch.float32, device=dev)          # (d_dirs accumulator + the zero sample depth, one fill)
    d_dirs = z[:n * 3].view(n, 3)
    _lib.call("mcnerf_mlp_apply_bwd", *net.triple, _p(params), _p(packed), _p(dirs), _p(z[n * 3:]), n, _p(out), _p(d_out),
              _p(save.mask, torch.int32), save.capacity, _p(save.enc), _p(save.sh), _p(dy), _p(dsh), _p(d_x), _p(d_dirs), _stream())
    if grads is not None:
        mlp_dw(net, save, dy, dsh, grads, n)
    return d_x, d_dirs


def mlp_bwd(net: Net, params: Tensor, packed: Tensor, rays_o: Tensor, rays_d: Tensor, zgrid: Tensor,
            jitter: Optional[Tensor], barf_w: Tensor, out: Tensor, d_out: Tensor, save: MlpSave,
            dy: Tensor, dsh: Tensor, d_rays_o: Optional[Tensor], d_rays_d: Optional[Tensor],
            idx: Optional[Tensor] = None, count: Optional[Tensor] = None, max_rows: int = 0,
            precision: str = "f32", gmax: Optional[Tensor] = None) -> None:
    n_rays, S = rays_d.shape[0], zgrid.numel()
    if is16(precision):
        _lib.call("mcnerf_mlp_bwd_16", *net.triple, DTYPE16[precision], _p(params), _p(packed16_split(net, packed, precision)[1], torch.uint8), _p(rays_o), _p(rays_d),
                  _p(zgrid), _p(jitter), _p(barf_w), _p(idx, torch.int32), _p(count, torch.int32), int(max_rows), n_rays, S,
                  _p(out), _p(d_out), _p(save.mask, torch.int32), save.capacity, _p(save.enc, torch.uint8), _p(save.sh, torch.uint8),
                  _p(dy, torch.uint8), _p(dsh, torch.uint8), _p(d_rays_o), _p(d_rays_d), _p(gmax, torch.int32), _stream())
        return
    args = [*net.triple, _p(params), _p(packed), _p(rays_o), _p(rays_d), _p(zgrid),
            _p(jitter), _p(barf_w), _p(idx, torch.int32), _p(count, torch.int32), int(max_rows), n_rays, S,
            _p(out), _p(d_out), _p(save.mask, torch.int32), save.capacity, _p(save.enc), _p(save.sh),
            _p(dy), _p(dsh), _p(d_rays_o), _p(d_rays_d)]
    _lib.call("mcnerf_mlp_bwd", *args, _stream())


def mlp_dw(net: Net, save: MlpSave, dy: Tensor, dsh: Tensor, grads: Tensor, rows: int,
           count: Optional[Tensor] = None, precision: str = "f32", gmax: Optional[Tensor] = None) -> None:
    if is16(precision):
        _lib.call("mcnerf_mlp_dw_16", *net.triple, DTYPE16[precision], _p(count, torch.int32), int(rows), _p(save.act, torch.uint8),
                  _p(save.enc, torch.uint8), _p(dy, torch.uint8), _p(dsh, torch.uint8), save.capacity, _p(grads),
                  _p(gmax, torch.int32), _stream())
        return
    args = [*net.triple, _p(count, torch.int32), int(rows), _p(save.act), _p(save.enc), _p(dy), _p(dsh), save.capacity, _p(grads)]
    _lib.call("mcnerf_mlp_dw", *args, _stream())


def composite_fwd(sig_rgb: Tensor, rays_d: Tensor, zgrid: Tensor, jitter: Optional[Tensor], eps: Tensor,
                  eps_sel: Optional[Tensor] = None, white_back: bool = True, want_depth: bool = False):
    """-> rgb [N,3], depth [N,1] | None, opacity [N,1] | None, w_sel [N,S] | None, wmax_bits | None"""
    N, S = rays_d.shape[0], zgrid.numel()
    dev = rays_d.device
    rgb = torch.empty(N, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(N, 1, dtype=torch.float32, device=dev) if want_depth else None
    opac = torch.empty(N, 1, dtype=torch.float32, device=dev) if want_depth else None
    w_sel = torch.empty(N, S, dtype=torch.float32, device=dev) if eps_sel is not None else None
    wmax = torch.zeros(1, dtype=torch.int32, device=dev) if eps_sel is not None else None
    _lib.call("mcnerf_composite_fwd", _p(sig_rgb), _p(rays_d), _p(zgrid), _p(jitter), _p(eps), _p(eps_sel), N, S,
              int(bool(white_back)), _p(rgb), _p(depth), _p(opac), _p(w_sel), _p(wmax, torch.int32), _stream())
    return rgb, depth, opac, w_sel, wmax


def composite_bwd(sig_rgb: Tensor, zgrid: Tensor, jitter: Optional[Tensor], eps: Tensor, d_rgb: Tensor,
                  white_back: bool = True, want_gmax: bool = False):
    """-> d_sig_rgb [N,S,4] (and, with want_gmax, the int32 word holding max|d_sig_rgb| as float bits)"""
    N, S = d_rgb.shape[0], zgrid.numel()
    d = torch.empty(N, S, 4, dtype=torch.float32, device=d_rgb.device)
    gmax = torch.zeros(1, dtype=torch.int32, device=d_rgb.device) if want_gmax else None
    _lib.call("mcnerf_composite_bwd", _p(sig_rgb), _p(zgrid), _p(jitter), _p(eps), _p(d_rgb), N, S,
              int(bool(white_back)), _p(d), _p(gmax, torch.int32), _stream())
    return (d, gmax) if want_gmax else d


def select_fine(w_sel: Tensor, wmax: Tensor, thresh: float, scale: int, sigma_default: float,
                prefill: bool = True):
    """-> idx [N*Sc*scale,2] int32 (first *count rows valid), count [1] int32, out_f [N,Sf,4] | None"""
    N, Sc = w_sel.shape
    dev = w_sel.device
    idx = torch.empty(N * Sc * scale, 2, dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    rc = torch.empty(N, dtype=torch.int32, device=dev)
    ro = torch.empty(N, dtype=torch.int32, device=dev)
    out_f = torch.empty(N, Sc * scale, 4, dtype=torch.float32, device=dev) if prefill else None
    _lib.call("mcnerf_select_fine", _p(w_sel), _p(wmax, torch.int32), float(thresh), N, Sc, scale,
              float(sigma_default), _p(rc, torch.int32), _p(ro, torch.int32), _p(idx, torch.int32),
              _p(count, torch.int32), _p(out_f), _stream())
    return idx, count, out_f


def cap_gather(idx: Tensor, perm: Tensor, keep: int):
    idx2 = torch.empty(keep, 2, dtype=torch.int32, device=idx.device)
    count = torch.empty(1, dtype=torch.int32, device=idx.device)
    _lib.call("mcnerf_cap_gather", _p(idx, torch.int32), _p(perm, torch.int64), int(keep), _p(idx2, torch.int32),
              _p(count, torch.int32), _stream())
    return idx2, count


def cap_random(idx: Tensor, count: Tensor, max_rows: int, keep: int, seed: Tensor):
    """Device-side random cap (model/mc_nerf.py:630-632 without the host sync): -> idx2 [keep,2], count2 = min(count, keep)."""
    dev = idx.device
    idx2 = torch.empty(keep, 2, dtype=torch.int32, device=dev)
    count2 = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(_lib.lib().mcnerf_cap_ws_words()), dtype=torch.int32, device=dev)
    _lib.call("mcnerf_cap_random", _p(idx, torch.int32), _p(count, torch.int32), int(max_rows), int(keep), _p(seed, torch.int32),
              _p(ws, torch.int32), _p(idx2, torch.int32), _p(count2, torch.int32), _stream())
    return idx2, count2


def camera_fwd(wpose: Tensor, wpose_intr: Tensor, wfx: Tensor, wfy: Tensor, wux: Tensor, wuy: Tensor, H: int, W: int,
               wpts_intr: Optional[Tensor] = None, wpts_extr: Optional[Tensor] = None):
    """-> K [C,3,3], Kinv [C,3,3], pose [C,3,4], calib_pose [C,3,4], pix_intr [C,P,2] | None, pix_extr [C,P,2] | None
    (pixels of the calibration points wpts_* [C,P,3] through (K, calib) / (K, pose))."""
    C, dev = wpose.shape[0], wpose.device
    K = torch.empty(C, 3, 3, dtype=torch.float32, device=dev)
    Kinv = torch.empty_like(K)
    pose = torch.empty(C, 3, 4, dtype=torch.float32, device=dev)
    calib = torch.empty_like(pose)
    P = 0
    for w in (wpts_intr, wpts_extr):
        if w is not None:
            P = w.shape[1]
    pi = torch.empty(C, P, 2, dtype=torch.float32, device=dev) if wpts_intr is not None else None
    pe = torch.empty(C, P, 2, dtype=torch.float32, device=dev) if wpts_extr is not None else None
    _lib.call("mcnerf_camera_fwd", _p(wpose), _p(wpose_intr), _p(wfx), _p(wfy), _p(wux), _p(wuy), C, int(H), int(W),
              _p(K), _p(Kinv), _p(pose), _p(calib), _p(wpts_intr), _p(wpts_extr), int(P), _p(pi), _p(pe), _stream())
    return K, Kinv, pose, calib, pi, pe


def camera_bwd(wpose, wpose_intr, wfx, wfy, wux, wuy, H: int, W: int, dK, dKinv, dpose, dcalib,
               wpts_intr=None, wpts_extr=None, dpix_intr=None, dpix_extr=None):
    C = wpose.shape[0]
    outs = [torch.empty_like(t) for t in (wpose, wpose_intr, wfx, wfy, wux, wuy)]
    c = lambda t: None if t is None else t.contiguous()
    dK, dKinv, dpose, dcalib, dpix_intr, dpix_extr = c(dK), c(dKinv), c(dpose), c(dcalib), c(dpix_intr), c(dpix_extr)
    P = 0
    for w in (wpts_intr, wpts_extr):
        if w is not None:
            P = w.shape[1]
    _lib.call("mcnerf_camera_bwd", _p(wpose), _p(wpose_intr), _p(wfx), _p(wfy), _p(wux), _p(wuy), C, int(H), int(W),
              _p(dK), _p(dKinv), _p(dpose), _p(dcalib), _p(wpts_intr), _p(wpts_extr), int(P), _p(dpix_intr), _p(dpix_extr),
              *[_p(o) for o in outs], _stream())
    return outs


def reproj_loss_fwd(pd: Tensor, gt: Tensor, H: int, W: int) -> Tensor:
    loss = torch.empty((), dtype=torch.float32, device=pd.device)
    _lib.call("mcnerf_reproj_loss_fwd", _p(pd), _p(gt), pd.numel() // 2, int(H), int(W), _p(loss), _stream())
    return loss


def reproj_loss_bwd(pd: Tensor, gt: Tensor, H: int, W: int, dloss: Tensor) -> Tensor:
    d = torch.empty_like(pd)
    _lib.call("mcnerf_reproj_loss_bwd", _p(pd), _p(gt), pd.numel() // 2, int(H), int(W), _p(dloss), _p(d), _stream())
    return d


def train_loss(pd: Optional[Tensor], pt_gt: Optional[Tensor], H: int, W: int, normalise: bool, rgb_c: Tensor, rgb_f: Optional[Tensor], gt: Tensor):
    """MC_NeRF_Loss.forward for the keys {"intr", "rgb"} in one launch: -> out [3] = (total, L_intr, rgb term), d_pd | None, d_c, d_f | None."""
    dev = rgb_c.device
    out = torch.zeros(TRAIN_LOSS_OUT, dtype=torch.float32, device=dev)      # [3] = the kernel's arrival counter: zero on entry (include/mcnerf.h)
    np_ = pd.numel() // 2 if pd is not None else 0
    d_pd = torch.empty_like(pd) if pd is not None else None
    d_c = torch.empty_like(rgb_c)
    d_f = torch.empty_like(rgb_f) if rgb_f is not None else None
    _lib.call("mcnerf_train_loss", _p(pd), _p(pt_gt), np_, int(H), int(W), int(bool(normalise)), _p(rgb_c), _p(rgb_f), _p(gt), rgb_c.numel(),
              _p(out), _p(d_pd), _p(d_c), _p(d_f), _stream())
    return out[:3], d_pd, d_c, d_f


TRAIN_LOSS_OUT = 132        # MCNERF_TRAIN_LOSS_OUT


def scale3_(a: Optional[Tensor], b: Tensor, c: Optional[Tensor], g: Tensor):
    """a, b, c *= g (device scalar) in place, one launch."""
    _lib.call("mcnerf_scale3", _p(a), a.numel() if a is not None else 0, _p(b), b.numel(), _p(c), c.numel() if c is not None else 0, _p(g), _stream())


def gather_gt(image_u8: Tensor, pix: Tensor) -> Tensor:
    """image_u8 [H*W, 3|4] uint8 on the device, pix [n] int64 -> [n,3] fp32 (RGBA blended on white)."""
    n = pix.numel()
    out = torch.empty(n, 3, dtype=torch.float32, device=pix.device)
    _lib.call("mcnerf_gather_gt", _p(image_u8, torch.uint8), int(image_u8.shape[-1]), _p(pix, torch.int64), n, _p(out), _stream())
    return out


def alloc_grad_ws(net: Net, save: MlpSave, precision: str):
    """dy / dsh workspaces of mlp_bwd for the precision mode."""
    if is16(precision):
        return (torch.empty_like(save.act),
                torch.empty(ws_bytes_16(net, save.capacity, 3, precision), dtype=torch.uint8, device=save.act.device))
    return torch.empty_like(save.act), torch.empty_like(save.sh)


def decode_frags_16(buf: Tensor, n_slots: int, width: int, rows: int, precision: str = "f16") -> Tensor:
    """Fragment-major workspace of the register-chain modes (csrc/mcnerf_16.h, mcnerf_x3.h) -> [n_slots, rows, width] fp32
    (f16x3: hi + lo, still carrying the kernel's power-of-two scale: SPLIT_SCALE_X for activations).  Debug / test helper."""
    ks = width // 16
    if precision == "f16x3":                                    # [slot][tile][part hi / lo][k-step][h][m][j], value = hi + lo
        x2 = buf.view(torch.float16).view(n_slots, -1, 2, ks, 2, 32, 8).float()
        x = x2[:, :, 0] + x2[:, :, 1]
    else:
        dt = torch.bfloat16 if precision == "bf16" else torch.float16        # ("f16x3h": the hi planes, an f16 workspace)
        x = buf.view(dt).view(n_slots, -1, ks, 2, 32, 8)        # [slot][tile][k-step][h][m][j]
    tiles = x.shape[1]
    s_, h_, j_ = torch.meshgrid(torch.arange(ks), torch.arange(2), torch.arange(8), indexing="ij")
    chan = (16 * s_ + 8 * (j_ // 4) + 4 * h_ + (j_ % 4)).reshape(-1).to(buf.device)       # channel of (s, h, j)
    y = x.permute(0, 1, 4, 2, 3, 5).reshape(n_slots, tiles * 32, ks * 16).float()          # [slot][row][(s,h,j)]
    out = torch.empty_like(y)
    out[:, :, chan] = y
    return out[:, :rows]


def decode_masks_16(mask: Tensor, n_slots: int, width: int, rows: int) -> Tensor:
    """Lane-local ReLU bit masks of the 16-bit modes (csrc/mcnerf_16.h, mlp16_fwd.hip) -> bool [n_slots, rows, width].
    Debug / test helper.  Word iw of lane (m, h) covers output tiles 2 iw, 2 iw + 1; inside a tile's 16 bits, packed word
    i (registers 2 i, 2 i + 1) sits at bit 7 - i (+ 16 for the odd register), register r = channel 32 t + 8 (r >> 2) + 4 h + (r & 3)."""
    mw = max(1, width // 64)
    mk = mask.view(n_slots, -1, 2, 32, mw).cpu()                # [slot][tile][h][m][word]
    tiles = mk.shape[1]
    out = torch.zeros(n_slots, tiles, 32, width, dtype=torch.bool)
    for t in range(width // 32):
        for r in range(16):
            i, odd = r >> 1, r & 1
            bitpos = 8 * (t & 1) + 7 - i + 16 * odd
            bit = ((mk[..., t >> 1] >> bitpos) & 1).bool()       # [slot][tile][h][m]
            for h in range(2):
                out[:, :, :, 32 * t + 8 * (r >> 2) + 4 * h + (r & 3)] = bit[:, :, h, :]
    return out.reshape(n_slots, tiles * 32, width)[:, :rows]


SPLIT_SCALE_X = 8.0      # MCNX3_SX of csrc/mcnerf_x3.h


def decode_sh_x3(buf: Tensor, rows: int) -> Tensor:
    """sh.2 outputs saved by the f16x3 forward (the fp32 accumulator tile: [tile][q][lane = 32 h + m][4]) -> [rows, 32]:
    register 4 q + e of lane (m, h) is SH row 8 q + 4 h + e.  Debug / test helper."""
    x = buf.view(torch.float32).view(-1, 4, 2, 32, 4)            # [tile][q][h][m][e]
    return x.permute(0, 3, 1, 2, 4).reshape(-1, 32)[:rows]       # [tile, m][q, h, e] -> row 8 q + 4 h + e
