"""CPU oracle for the MC-NeRF volumetric-rendering hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline.  The product path
(``mc_nerf_amd``) never routes through this file.

It is a plain-PyTorch (fp32, CPU) functional restatement of the reference's
algorithm with every random draw turned into an explicit input, so that the
HIP kernels can be compared on identical rays + jitter + noise.  Every
function cites the reference ``file:line`` it follows (paths relative to the
reference checkout, which does not travel to the GPU box).

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the actual
reference (``/root/reference``) in the development container, captures its RNG
draws and outputs, and commits them as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this file against those vectors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class NetCfg:
    """Topology of one CorseFine_NeRF (model/net_block.py:37-65)."""
    depth: int
    width: int
    skips: Tuple[int, ...]


@dataclass
class RenderCfg:
    """Renderer constants (model/mc_nerf.py:548-571, config/config.yaml:58-82)."""
    near: float = 1.0
    far: float = 8.0
    samples: int = 64
    scale: int = 2
    weight_thresh: float = 1e-3
    sigma_default: float = -20.0
    white_back: bool = True
    n_freqs: int = 10
    barf_mode: bool = False
    barf_start: float = 0.0
    barf_end: float = 1.0
    deg: int = 2
    coarse: NetCfg = field(default_factory=lambda: NetCfg(4, 128, (2,)))
    fine: NetCfg = field(default_factory=lambda: NetCfg(8, 256, (4,)))
    max_fine_per_ray: int = 128  # model/mc_nerf.py:630

    @property
    def samples_f(self) -> int:
        return self.samples * self.scale


# --------------------------------------------------------------------------- encoding
def embed(x: Tensor, step_r: float, cfg: RenderCfg) -> Tensor:
    """SinCosEmbedding.forward (model/net_block.py:20-35).

    Output channel order: [x,y,z, then per coordinate c: sin(2^0 c..2^9 c), cos(2^0 c..2^9 c)].
    BARF weights w_k multiply both the sin and the cos of frequency k (:26-32).
    """
    L = cfg.n_freqs
    freqs = 2.0 ** torch.linspace(0, L - 1, L, dtype=torch.float32)
    arg = x.unsqueeze(-1) * freqs                      # [M,3,L]
    enc = torch.cat([arg.sin(), arg.cos()], dim=-1)    # [M,3,2L]: sin block then cos block
    if cfg.barf_mode:
        alpha = (step_r - cfg.barf_start) / (cfg.barf_end - cfg.barf_start) * L
        k = torch.arange(L, dtype=torch.float32)
        w = (1.0 - torch.cos(torch.clamp(alpha - k, 0.0, 1.0) * math.pi)) / 2.0
        enc = enc * torch.cat([w, w])
    return torch.cat([x, enc.reshape(x.shape[0], -1)], dim=-1)


def barf_weights(step_r: float, cfg: RenderCfg) -> Tensor:
    """The per-frequency mask of model/net_block.py:26-29 (all ones when barf is off)."""
    L = cfg.n_freqs
    if not cfg.barf_mode:
        return torch.ones(L, dtype=torch.float32)
    alpha = (step_r - cfg.barf_start) / (cfg.barf_end - cfg.barf_start) * L
    k = torch.arange(L, dtype=torch.float32)
    return (1.0 - torch.cos(torch.clamp(alpha - k, 0.0, 1.0) * math.pi)) / 2.0


# --------------------------------------------------------------------------- SH colour
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)


def sh_basis_deg2(dirs: Tensor) -> Tensor:
    """The nine signed basis factors of eval_sh at deg 2 (model/net_utils.py:154-169)."""
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    return torch.stack([
        torch.full_like(x, SH_C0),
        -SH_C1 * y, SH_C1 * z, -SH_C1 * x,
        SH_C2[0] * (x * y), SH_C2[1] * (y * z),
        SH_C2[2] * (2.0 * z * z - x * x - y * y),
        SH_C2[3] * (x * z), SH_C2[4] * (x * x - y * y)], dim=-1)


def eval_sh_deg2(sh: Tensor, dirs: Tensor) -> Tensor:
    """eval_sh(deg=2) (model/net_utils.py:103-191): sh [M,3,9], dirs [M,3] -> [M,3]."""
    return (sh * sh_basis_deg2(dirs).unsqueeze(-2)).sum(-1)


SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


def eval_sh(deg: int, sh: Tensor, dirs: Tensor) -> Tensor:
    """eval_sh for degrees 0 .. 3 (model/net_utils.py:103-179), term by term in the reference's order:
    sh [M,3,(deg+1)^2], dirs [M,3] -> [M,3]."""
    assert 0 <= deg <= 3 and sh.shape[-1] == (deg + 1) ** 2
    result = SH_C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        result = (result - SH_C1 * y * sh[..., 1] + SH_C1 * z * sh[..., 2] - SH_C1 * x * sh[..., 3])
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            result = (result + SH_C2[0] * xy * sh[..., 4] + SH_C2[1] * yz * sh[..., 5] + SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] +
                      SH_C2[3] * xz * sh[..., 7] + SH_C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                result = (result + SH_C3[0] * y * (3 * xx - yy) * sh[..., 9] + SH_C3[1] * xy * z * sh[..., 10] +
                          SH_C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12] +
                          SH_C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + SH_C3[5] * z * (xx - yy) * sh[..., 14] +
                          SH_C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return result


# --------------------------------------------------------------------------- MLP
def mlp_forward(p: Dict[str, Tensor], net: NetCfg, x_enc: Tensor, dirs: Tensor,
                return_hidden: bool = False):
    """CorseFine_NeRF.forward (model/net_block.py:67-78).

    ``p`` uses the reference state-dict keys of one net:
    ``xyz_encoding_{i}.0.{weight,bias}``, ``sigma.{0,2}.*``, ``sh.{0,2}.*``;
    Linear weights are [out,in]; the skip concatenation order is [x_enc, h] (:71).
    """
    h = x_enc
    hidden = []
    for i in range(net.depth):
        if i in net.skips:
            h = torch.cat([x_enc, h], dim=-1)
        h = F.relu(F.linear(h, p[f"xyz_encoding_{i+1}.0.weight"], p[f"xyz_encoding_{i+1}.0.bias"]))
        hidden.append(h)
    hs = F.relu(F.linear(h, p["sigma.0.weight"], p["sigma.0.bias"]))
    sigma = F.linear(hs, p["sigma.2.weight"], p["sigma.2.bias"])
    hc = F.relu(F.linear(h, p["sh.0.weight"], p["sh.0.bias"]))
    sh = F.linear(hc, p["sh.2.weight"], p["sh.2.bias"])
    nb = sh.shape[-1] // 3                                  # (deg + 1)^2, from the sh.2 layer's width (MLP_deg, model/net_block.py:43)
    rgb = torch.sigmoid(eval_sh_deg2(sh.reshape(-1, 3, 9), dirs) if nb == 9 else eval_sh(int(round(nb ** 0.5)) - 1, sh.reshape(-1, 3, nb), dirs))
    out = torch.cat([sigma, rgb], dim=-1)
    if return_hidden:
        return out, hidden + [hs, hc], sh
    return out


# --------------------------------------------------------------------------- compositing
def deltas_of(z_vals: Tensor) -> Tensor:
    """Forward differences with the 1e10 tail (model/mc_nerf.py:615-618, 708-710)."""
    d = z_vals[:, 1:] - z_vals[:, :-1]
    return torch.cat([d, torch.full_like(d[:, :1], 1e10)], dim=-1)


def sigma2weights(deltas: Tensor, sigmas: Tensor, eps: Tensor) -> Tensor:
    """NeRF_Model.sigma2weights with the N(0,1) draw passed in (model/mc_nerf.py:729-736)."""
    alphas = 1.0 - torch.exp(-deltas * F.softplus(sigmas + eps))
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1.0 - alphas + 1e-10], dim=-1)
    return alphas * torch.cumprod(shifted, dim=-1)[:, :-1]


def composite(sig_rgb: Tensor, rays_d: Tensor, z_vals: Tensor, eps: Tensor, white_back: bool):
    """Both composites of NeRF_Model.inference (model/mc_nerf.py:705-727).

    sig_rgb [N,S,4] -> rgb [N,3], depth [N,1], opacity [N,1], weights [N,S].
    Depth/opacity use the noise-free cumsum form with delta scaled by |d| (:707-718);
    rgb uses sigma2weights (delta NOT scaled by |d|) (:719-725).
    """
    sigmas, rgbs = sig_rgb[..., 0], sig_rgb[..., 1:]
    deltas = deltas_of(z_vals)
    sd = F.softplus(sigmas) * (deltas * rays_d.norm(dim=-1, keepdim=True))
    alpha = 1.0 - torch.exp(-sd)
    T = torch.exp(-torch.cat([torch.zeros_like(sd[:, :1]), sd[:, :-1]], dim=1).cumsum(dim=1))
    prob = T * alpha
    opacity = prob.sum(dim=1, keepdim=True)
    depth = (z_vals * prob).sum(dim=1, keepdim=True)
    w = sigma2weights(deltas, sigmas, eps)
    rgb = (w.unsqueeze(-1) * rgbs).sum(dim=1)
    if white_back:
        rgb = rgb + 1.0 - w.sum(dim=1, keepdim=True)
    return rgb, depth, opacity, w


# --------------------------------------------------------------------------- selection
def select_fine(weights: Tensor, cfg: RenderCfg) -> Tensor:
    """Weight-threshold refinement (model/mc_nerf.py:623-629): [N,Sc] -> idx [K,2] int64.

    Rows are in torch.nonzero order (ray-major, then coarse sample, then sub-sample r).
    """
    thr = min(cfg.weight_thresh, float(weights.max()))
    idx = torch.nonzero(weights >= thr)                       # [X,2]
    r = torch.arange(cfg.scale, dtype=idx.dtype)
    rays = idx[:, :1].expand(-1, cfg.scale)
    fine = idx[:, 1:] * cfg.scale + r.unsqueeze(0)
    return torch.stack([rays, fine], dim=-1).reshape(-1, 2)


# --------------------------------------------------------------------------- inference
def inference(p: Dict[str, Tensor], net: NetCfg, cfg: RenderCfg, step_r: float,
              rays_o: Optional[Tensor], rays_d: Tensor, z_vals: Tensor, eps: Tensor,
              idx_render: Optional[Tensor] = None, xyz: Optional[Tensor] = None):
    """NeRF_Model.inference (model/mc_nerf.py:682-727) for one net.  The reference takes the sample positions `xyz`
    [N,S,3] as an argument (its callers compute o + d z, :602 / :637); here they are built from `rays_o` unless given.

    Returns rgb [N,3], sigmas [N,S], depth [N,1], opacity [N,1], sig_rgb [N,S,4].
    """
    N, S = z_vals.shape
    if xyz is None:
        xyz = rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z_vals.unsqueeze(2)
    if idx_render is None:
        dirs = rays_d.unsqueeze(1).expand(-1, S, -1).reshape(-1, 3)
        out = mlp_forward(p, net, embed(xyz.reshape(-1, 3), step_r, cfg), dirs).reshape(N, S, 4)
    else:
        r, j = idx_render[:, 0], idx_render[:, 1]
        out_k = mlp_forward(p, net, embed(xyz[r, j], step_r, cfg), rays_d[r])
        sig0 = torch.full((N, S, 1), cfg.sigma_default)
        out = torch.cat([sig0, torch.ones(N, S, 3)], dim=-1)
        out = out.index_put((r, j), out_k)
    rgb, depth, opacity, _ = composite(out, rays_d, z_vals, eps, cfg.white_back)
    return rgb, out[..., 0], depth, opacity, out


def _grids(cfg: RenderCfg):
    """model/mc_nerf.py:570-571."""
    return (torch.linspace(cfg.near, cfg.far, cfg.samples),
            torch.linspace(cfg.near, cfg.far, cfg.samples_f))


def render_rays_train(pc: Dict[str, Tensor], pf: Dict[str, Tensor], cfg: RenderCfg,
                      rays_d: Tensor, rays_o: Tensor, step_r: float,
                      jitter: Tensor, eps_c: Tensor, eps_sel: Tensor, eps_f: Tensor,
                      cap_perm: Optional[Tensor] = None, only_coarse: bool = False,
                      idx_override: Optional[Tensor] = None):
    """NeRF_Model.render_rays_train (model/mc_nerf.py:598-646) with explicit draws.

    jitter [N,1] ~ U(0,(far-near)/Sc) (:600); eps_c, eps_sel [N,Sc] and eps_f [N,Sf] are the
    three N(0,1) draws of sigma2weights (:719 coarse, :619 selection, :719 fine);
    cap_perm is the CPU randperm of :631 (only used when K > N*128).
    idx_override [K,2], when given, IS the kept (ray, fine sample) list of :625-632 -- selection and cap already applied by
    the caller (the HIP path's device-side list restricted to a ray subset): the fine pass renders exactly those samples.
    Returns dict(rgb_c, rgb_f, depth_c, sig_c, w_sel, idx_f, out_c, out_f).
    """
    N = rays_d.shape[0]
    zc, zf = _grids(cfg)
    z_c = zc.unsqueeze(0).expand(N, -1) + jitter
    rgb_c, sig_c, depth_c, _, out_c = inference(pc, cfg.coarse, cfg, step_r, rays_o, rays_d, z_c, eps_c)
    res = dict(rgb_c=rgb_c, depth_c=depth_c, sig_c=sig_c, out_c=out_c)
    if only_coarse:
        return res
    with torch.no_grad():
        w_sel = sigma2weights(deltas_of(z_c), sig_c.detach(), eps_sel)
        idx_f = select_fine(w_sel, cfg) if idx_override is None else idx_override
        if idx_override is None and idx_f.shape[0] > N * cfg.max_fine_per_ray:
            idx_f = idx_f[cap_perm[: N * cfg.max_fine_per_ray]]
    z_f = zf.unsqueeze(0).expand(N, -1) + jitter
    rgb_f, _, _, _, out_f = inference(pf, cfg.fine, cfg, step_r, rays_o, rays_d, z_f, eps_f, idx_render=idx_f)
    res.update(rgb_f=rgb_f, w_sel=w_sel, idx_f=idx_f, out_f=out_f)
    return res


def render_rays_test(pc: Dict[str, Tensor], pf: Dict[str, Tensor], cfg: RenderCfg,
                     rays_d: Tensor, rays_o: Tensor,
                     eps_c: Tensor, eps_sel: Tensor, eps_f: Tensor,
                     idx_override: Optional[Tensor] = None):
    """NeRF_Model.render_rays_test (model/mc_nerf.py:648-680): no jitter, step_r=1, no cap."""
    N = rays_d.shape[0]
    zc, zf = _grids(cfg)
    z_c = zc.unsqueeze(0).expand(N, -1)
    rgb_c, sig_c, depth_c, op_c, out_c = inference(pc, cfg.coarse, cfg, 1, rays_o, rays_d, z_c, eps_c)
    w_sel = sigma2weights(deltas_of(z_c), sig_c, eps_sel)
    idx_f = select_fine(w_sel, cfg) if idx_override is None else idx_override
    z_f = zf.unsqueeze(0).expand(N, -1)
    rgb_f, _, depth_f, op_f, out_f = inference(pf, cfg.fine, cfg, 1, rays_o, rays_d, z_f, eps_f, idx_render=idx_f)
    return dict(rgb=rgb_f, depth=depth_f, opacity=op_f, rgb_c=rgb_c, sig_c=sig_c, w_sel=w_sel,
                idx_f=idx_f, out_c=out_c, out_f=out_f)


def rgb_loss(rgb_c: Tensor, rgb_f: Optional[Tensor], gt: Tensor) -> Tensor:
    """MC_NeRF_Loss.get_rgb_loss (model/loss.py:33-43)."""
    loss = F.mse_loss(rgb_c, gt)
    if rgb_f is not None:
        loss = loss + F.mse_loss(rgb_f, gt)
    return loss


def permute_hidden_units(p: Dict[str, Tensor], net: NetCfg, seed: int):
    """The same network function (model/net_block.py:67-78) with every hidden layer's units enumerated in another order (rows of
    the producing layer, matching input columns of its consumers).  fp32 arithmetic is not invariant under it -- sums run in
    another order, a pre-activation within a rounding of zero takes the other side of its ReLU -- so evaluating the path on
    both is a measurement of the reference arithmetic's own noise (the tolerance floor of at-size gradient comparisons).
    -> (state dict, un-permute function for a dict of gradients keyed like the state dict)"""
    gen = torch.Generator().manual_seed(seed)
    q = {k: v.clone() for k, v in p.items()}
    D, W = net.depth, net.width
    plan = []                                              # (producer keys, [(consumer key, column offset)], perm)
    for l in range(1, D + 1):
        cons = [f"xyz_encoding_{l + 1}.0.weight"] if l < D else ["sigma.0.weight", "sh.0.weight"]
        plan.append(([f"xyz_encoding_{l}.0.weight", f"xyz_encoding_{l}.0.bias"], cons, torch.randperm(W, generator=gen)))
    for head in ("sigma", "sh"):
        plan.append(([f"{head}.0.weight", f"{head}.0.bias"], [f"{head}.2.weight"], torch.randperm(W, generator=gen)))

    def apply(d, inverse):
        out = {k: v.clone() for k, v in d.items()}
        for prod, cons, pm in plan:
            ix = torch.argsort(pm) if inverse else pm
            for k in prod:
                out[k] = out[k][ix]
            for c in cons:
                off = out[c].shape[1] - W                  # (the skip layer's input is [x_enc(63), h])
                w = out[c].clone()
                w[:, off:] = out[c][:, off:][:, ix]
                out[c] = w
        return out
    return apply(q, False), (lambda grads: apply(grads, True))


# --------------------------------------------------------------------------- cameras / rays
def get_rays(pose: Tensor, intr_inv: Tensor, H: int, W: int):
    """MC_Model.get_rays for ONE camera (model/mc_nerf.py:124-145, 213-256).

    pose [3,4] is world->cam [R|t]; intr_inv [3,3].  Follows the reference's op order
    (pix @ K^-T, homogeneous lift, @ pose_inv^T, subtract origin, normalise).
    """
    ys = torch.arange(H, dtype=torch.float32) + 0.5
    xs = torch.arange(W, dtype=torch.float32) + 0.5
    Y, X = torch.meshgrid(ys, xs, indexing="ij")
    pix = torch.stack([X, Y, torch.ones_like(X)], dim=-1).reshape(-1, 3)
    cam = pix @ intr_inv.transpose(-2, -1)
    R_inv = pose[:, :3].transpose(-2, -1)
    pose_inv = torch.cat([R_inv, -R_inv @ pose[:, 3:]], dim=-1)          # [3,4]
    cam_h = torch.cat([cam, torch.ones_like(cam[:, :1])], dim=-1)
    org_h = torch.cat([torch.zeros_like(cam), torch.ones_like(cam[:, :1])], dim=-1)
    world = cam_h @ pose_inv.transpose(-2, -1)
    rays_o = org_h @ pose_inv.transpose(-2, -1)
    rays_d = world - rays_o
    rays_d = rays_d / rays_d.norm(dim=-1, keepdim=True)
    return rays_d, rays_o


def get_rays_at(pose: Tensor, intr_inv: Tensor, pix_idx: Tensor, W: int):
    """Closed form of get_rays restricted to pixel ids (row-major v*W+u):
    d = normalize(R^T K^-1 [u+.5, v+.5, 1]^T), o = -R^T t."""
    u = (pix_idx % W).to(torch.float32) + 0.5
    v = torch.div(pix_idx, W, rounding_mode="floor").to(torch.float32) + 0.5
    pix = torch.stack([u, v, torch.ones_like(u)], dim=-1)
    cam = pix @ intr_inv.transpose(-2, -1)
    Rt = pose[:, :3].transpose(-2, -1)
    d = cam @ Rt.transpose(-2, -1)
    d = d / d.norm(dim=-1, keepdim=True)
    o = (-Rt @ pose[:, 3:]).reshape(1, 3).expand_as(d)
    return d, o


def _taylor(theta: Tensor, kind: str, nth: int = 10) -> Tensor:
    """taylor_A/B/C (model/mc_nerf.py:291-316)."""
    ans = torch.zeros_like(theta)
    denom = 1.0
    for i in range(nth + 1):
        if kind == "A":
            if i > 0:
                denom *= (2 * i) * (2 * i + 1)
        elif kind == "B":
            denom *= (2 * i + 1) * (2 * i + 2)
        else:
            denom *= (2 * i + 2) * (2 * i + 3)
        ans = ans + (-1) ** i * theta ** (2 * i) / denom
    return ans


def se3_to_SE3(wu: Tensor) -> Tensor:
    """MC_Model.se3_to_SE3 (model/mc_nerf.py:269-289): [C,6] -> [C,3,4]."""
    w, u = wu[..., :3], wu[..., 3:]
    O = torch.zeros_like(w[..., 0])
    wx = torch.stack([torch.stack([O, -w[..., 2], w[..., 1]], -1),
                      torch.stack([w[..., 2], O, -w[..., 0]], -1),
                      torch.stack([-w[..., 1], w[..., 0], O], -1)], -2)
    theta = w.norm(dim=-1)[..., None, None]
    I = torch.eye(3)
    A, B, C = _taylor(theta, "A"), _taylor(theta, "B"), _taylor(theta, "C")
    R = I + A * wx + B * wx @ wx
    V = I + B * wx + C * wx @ wx
    return torch.cat([R, V @ u[..., None]], dim=-1)


def intrinsics_from_weights(H: int, W: int, wfx: Tensor, wfy: Tensor, wux: Tensor, wuy: Tensor) -> Tensor:
    """MC_Model.add_weights2intr (model/mc_nerf.py:171-186). Note W scales fy as well."""
    C = wfx.shape[0]
    K = torch.zeros(C, 3, 3)
    K[:, 0, 0] = torch.abs(float(W) * wfx)
    K[:, 1, 1] = torch.abs(float(W) * wfy)
    K[:, 0, 2] = torch.abs(float(W) / 2 * wux)
    K[:, 1, 2] = torch.abs(float(H) / 2 * wuy)
    K[:, 2, 2] = 1.0
    return K


def inverse_intrinsic(K: Tensor) -> Tensor:
    """MC_Model.inverse_intrinsic (model/mc_nerf.py:204-210)."""
    return torch.stack([k.inverse() for k in K], 0)


# --------------------------------------------------------------------------- utilities
def init_params(net: NetCfg, seed: int, in_ch: int = 63, n_sh: int = 27) -> Dict[str, Tensor]:
    """Deterministic U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weights from numpy's frozen
    RandomState stream (same bound as nn.Linear's default init); used by tests, bench and the
    golden generator so that weights never have to be stored in a fixture."""
    import numpy as np
    rs = np.random.RandomState(seed)
    p: Dict[str, Tensor] = {}

    def lin(name, fo, fi):
        b = 1.0 / math.sqrt(fi)
        p[name + ".weight"] = torch.from_numpy(rs.uniform(-b, b, (fo, fi)).astype(np.float32))
        p[name + ".bias"] = torch.from_numpy(rs.uniform(-b, b, (fo,)).astype(np.float32))

    for i in range(net.depth):
        fi = in_ch if i == 0 else (net.width + in_ch if i in net.skips else net.width)
        lin(f"xyz_encoding_{i+1}.0", net.width, fi)
    lin("sigma.0", net.width, net.width)
    lin("sigma.2", 1, net.width)
    lin("sh.0", net.width, net.width)
    lin("sh.2", n_sh, net.width)
    return p
