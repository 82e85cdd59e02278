"""GPU tests of the split-f16 ("f16x3") register-chain kernels (csrc/mcnerf_x3.h, mlp_x3_*.hip) against the fp32 CPU
oracle: the fp32-GRADE mode (22-bit operands, three f16 MFMAs per product, fp32 accumulate), held to the same per-op
bounds as the exact-fp32 kernels (tests/test_a_ops_gpu.py) - only summation order and the 2^-21 product rounding differ.
"""
import math

import numpy as np
import pytest
import torch

from oracle import mcnerf_oracle as O

pytestmark = pytest.mark.gpu

NETS = {32: O.NetCfg(4, 32, (2,)), 64: O.NetCfg(8, 64, (4,)), 128: O.NetCfg(4, 128, (2,)), 256: O.NetCfg(8, 256, (4,))}
P = "f16x3"


def _ops():
    from mc_nerf_amd import ops
    return ops


def make_rays(n, seed, radius=3.0):
    g = torch.Generator().manual_seed(seed)
    o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * radius
    tgt = (torch.rand(n, 3, generator=g) - 0.5) * 1.5
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    return d.contiguous(), o.contiguous()


def net_of(nc):
    return _ops().Net(nc.depth, nc.width, nc.skips[0])


def flat_params(nc, p, dev):
    ops = _ops()
    net = net_of(nc)
    return ops.flatten_params(net, [p[k].to(dev) for k in net.names()], dev)


def maxerr(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


@pytest.mark.parametrize("width", [32, 64, 128, 256])
@pytest.mark.parametrize("barf", [False, True])
def test_x3_fwd_dense(gpu_device, width, barf):
    """Dense grid, ragged row count; encodings, every hidden layer, the sh.2 outputs and the output against the oracle."""
    ops = _ops()
    dev = gpu_device
    nc = NETS[width]
    net = net_of(nc)
    S, N = 48, 37
    cfg = O.RenderCfg(samples=S, scale=2, coarse=nc, fine=nc, barf_mode=barf, barf_start=0.3846, barf_end=0.6923)
    step_r = 0.5
    p = O.init_params(nc, 100 + width)
    d, o = make_rays(N, 5 + width)
    g = torch.Generator().manual_seed(1)
    jitter = torch.rand(N, 1, generator=g) * (cfg.far - cfg.near) / S
    zg = torch.linspace(cfg.near, cfg.far, S)
    z = zg.unsqueeze(0) + jitter
    xyz = (o.unsqueeze(1) + d.unsqueeze(1) * z.unsqueeze(2)).reshape(-1, 3)
    dirs = d.unsqueeze(1).expand(-1, S, -1).reshape(-1, 3)
    ref, hidden, sh = O.mlp_forward(p, nc, O.embed(xyz, step_r, cfg), dirs, return_hidden=True)

    flat = flat_params(nc, p, dev)
    packed = ops.pack_weights(net, flat, precision=P)
    out = torch.full((N, S, 4), float("nan"), device=dev)
    save = ops.alloc_save(net, N * S, dev, precision=P)
    bw = O.barf_weights(step_r, cfg).to(dev)
    ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous(), bw,
                out, save=save, precision=P)
    torch.cuda.synchronize()
    enc = ops.decode_frags_16(save.enc, 1, 64, N * S, P)[0][:, :63] / ops.SPLIT_SCALE_X
    assert maxerr(enc, O.embed(xyz, step_r, cfg)) < 2e-6
    act = ops.decode_frags_16(save.act, nc.depth + 2, width, N * S, P) / ops.SPLIT_SCALE_X
    for l, h in enumerate(hidden):
        e = maxerr(act[l], h)
        assert e < 2e-5, f"layer {l}: {e}"
    assert maxerr(ops.decode_sh_x3(save.sh, N * S)[:, :27], sh) < 2e-5
    e_out = maxerr(out.view(-1, 4), ref)
    print(f"[f16x3 W={width} barf={barf}] out {e_out:.1e}")
    assert e_out < 2e-5
    # the ReLU bits are those of the saved activations
    masks = ops.decode_masks_16(save.mask, nc.depth + 2, width, N * S)
    assert torch.equal(masks, act.cpu() > 0)
    # the no-save instantiation gives the same output bit for bit
    out2 = torch.empty_like(out)
    ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous(), bw, out2,
                precision=P)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("width", [128, 256])
def test_x3_fwd_indexed_multi_pass(gpu_device, width):
    """Compacted (ray, sample) list with a device-side count, several passes per workgroup (the ring keeps streaming
    across passes), entries outside the list left untouched."""
    ops = _ops()
    dev = gpu_device
    N, S = 1400, 64
    nc = NETS[width]
    cfg = O.RenderCfg(samples=S, scale=2, coarse=nc, fine=nc)
    net = net_of(nc)
    p = O.init_params(nc, 103 + width)
    d, o = make_rays(N, 8 + width)
    g = torch.Generator().manual_seed(7)
    jitter = torch.rand(N, 1, generator=g) * (cfg.far - cfg.near) / S
    zg = torch.linspace(cfg.near, cfg.far, S)
    keep = torch.rand(N, S, generator=g) < 0.57
    idx = torch.nonzero(keep).to(torch.int32)
    K = idx.shape[0]
    cap = N * S
    idx_pad = torch.zeros(cap, 2, dtype=torch.int32)
    idx_pad[:K] = idx
    flat = flat_params(nc, p, dev)
    packed = ops.pack_weights(net, flat, precision=P)
    out = torch.full((N, S, 4), -7.0, device=dev)
    ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous(),
                O.barf_weights(1.0, cfg).to(dev), out, idx=idx_pad.to(dev), count=torch.tensor([K], dtype=torch.int32, device=dev),
                max_rows=cap, precision=P)
    torch.cuda.synchronize()
    out = out.cpu()
    assert torch.all(out[~keep] == -7.0)
    # oracle on a subset of the rows (the whole list takes a while on the CPU)
    sub = torch.randperm(K, generator=g)[:4000]
    r, j = idx[sub, 0].long(), idx[sub, 1].long()
    z = zg.unsqueeze(0) + jitter
    xyz = o[r] + d[r] * z[r, j].unsqueeze(-1)
    ref = O.mlp_forward(p, nc, O.embed(xyz, 1.0, cfg), d[r])
    e = maxerr(out[r, j], ref)
    print(f"[f16x3 W={width}] indexed, {K} rows: out {e:.1e}")
    assert e < 2e-5


@pytest.mark.parametrize("width", [32, 64, 128, 256])
def test_x3_fwd_bwd_indexed(gpu_device, width):
    """Fine-pass mode: (ray, sample) list + device count; forward, dX chain (ray gradients), dW against autograd of the
    oracle, at the exact-fp32 kernels' tolerance."""
    ops = _ops()
    dev = gpu_device
    nc = NETS[width]
    net = net_of(nc)
    S, N = 40, 29
    cfg = O.RenderCfg(samples=20, scale=2, coarse=nc, fine=nc, barf_mode=True, barf_start=0.2, barf_end=0.9)
    step_r = 0.6
    p = {k: v.requires_grad_(True) for k, v in O.init_params(nc, 200 + width).items()}
    d, o = make_rays(N, 9 + width)
    d.requires_grad_(True)
    o.requires_grad_(True)
    g = torch.Generator().manual_seed(2)
    jitter = torch.rand(N, 1, generator=g) * 0.2
    zg = torch.linspace(cfg.near, cfg.far, S)
    sel = torch.rand(N, S, generator=g) < 0.6
    idx = torch.nonzero(sel)
    K = idx.shape[0]
    z = zg.unsqueeze(0) + jitter
    r, j = idx[:, 0], idx[:, 1]
    xyz = o[r] + d[r] * z[r, j].unsqueeze(-1)
    ref = O.mlp_forward(p, nc, O.embed(xyz, step_r, cfg), d[r])
    gout = torch.randn(K, 4, generator=g)
    (ref * gout).sum().backward()

    flat = flat_params(nc, {k: v.detach() for k, v in p.items()}, dev)
    packed = ops.pack_weights(net, flat, precision=P)
    cap = K + 17
    idx_d = torch.zeros(cap, 2, dtype=torch.int32, device=dev)
    idx_d[:K] = idx.to(torch.int32).to(dev)
    count = torch.tensor([K], dtype=torch.int32, device=dev)
    out = torch.full((N, S, 4), 7.0, device=dev)
    save = ops.alloc_save(net, cap, dev, precision=P)
    bw = O.barf_weights(step_r, cfg).to(dev)
    od, dd, zd, jd = o.detach().to(dev), d.detach().to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous()
    ops.mlp_fwd(net, flat, packed, od, dd, zd, jd, bw, out, idx=idx_d, count=count, max_rows=cap, save=save, precision=P)
    got = out[r.to(dev), j.to(dev)]
    assert maxerr(got, ref) < 2e-5
    assert torch.all(out[~sel.to(dev)] == 7.0)          # untouched elsewhere

    d_out = torch.zeros(N, S, 4, device=dev)
    d_out[r.to(dev), j.to(dev)] = gout.to(dev)
    grads = torch.zeros_like(flat)
    dy, dsh = ops.alloc_grad_ws(net, save, P)
    d_o = torch.zeros(N, 3, device=dev)
    d_d = torch.zeros(N, 3, device=dev)
    gmax = d_out.abs().max().reshape(1).view(torch.int32)        # what composite_bwd hands to the backward
    ops.mlp_bwd(net, flat, packed, od, dd, zd, jd, bw, out, d_out, save, dy, dsh, d_o, d_d,
                idx=idx_d, count=count, max_rows=cap, precision=P, gmax=gmax)
    ops.mlp_dw(net, save, dy, dsh, grads, cap, count=count, precision=P, gmax=gmax)
    torch.cuda.synchronize()
    e_o, e_d = maxerr(d_o, o.grad), maxerr(d_d, d.grad)
    assert e_o < 2e-5 * max(1.0, float(o.grad.abs().max())), (e_o, float(o.grad.abs().max()))
    assert e_d < 2e-5 * max(1.0, float(d.grad.abs().max())), (e_d, float(d.grad.abs().max()))
    worst = 0.0
    for off, shp, name in zip(ops.param_offsets(net), net.shapes(), net.names()):
        n = int(np.prod(shp))
        gg = grads[off:off + n].view(shp)
        ref_g = p[name].grad
        tol = 2e-5 * max(1.0, float(ref_g.abs().max()))
        e = maxerr(gg, ref_g)
        worst = max(worst, e / max(1.0, float(ref_g.abs().max())))
        assert e < tol, f"{name}: {e} (tol {tol})"
    print(f"[f16x3 W={width}] d_o {e_o:.1e} d_d {e_d:.1e} worst dW (rel to max(1, max|ref|)) {worst:.1e}")


@pytest.mark.parametrize("width", [32, 64, 128, 256])
def test_x3_dw_at_scale(gpu_device, width):
    """The persistent dW kernel over ~0.4 M rows (thousands of tiles per workgroup, ring / DMA / transposed reads busy)
    against a torch fp64 GEMM of the very operands it reads (decoded hi + lo planes)."""
    ops = _ops()
    dev = gpu_device
    nc = NETS[width]
    net = net_of(nc)
    D, W, skip = nc.depth, nc.width, nc.skips[0]
    N, S = 3001, 128
    rows = N * S
    p = O.init_params(nc, 300 + width)
    flat = flat_params(nc, p, dev)
    packed = ops.pack_weights(net, flat, precision=P)
    d, o = make_rays(N, 77)
    od, dd, zd = o.to(dev), d.to(dev), torch.linspace(1, 8, S, device=dev)
    bw = torch.ones(10, device=dev)
    out = torch.empty(N, S, 4, device=dev)
    save = ops.alloc_save(net, rows, dev, precision=P)
    ops.mlp_fwd(net, flat, packed, od, dd, zd, None, bw, out, save=save, precision=P)
    d_out = torch.randn(N, S, 4, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 1e-3
    gmax = d_out.abs().max().reshape(1).view(torch.int32)
    dy, dsh = ops.alloc_grad_ws(net, save, P)
    d_o, d_d = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
    ops.mlp_bwd(net, flat, packed, od, dd, zd, None, bw, out, d_out, save, dy, dsh, d_o, d_d, precision=P, gmax=gmax)
    grads = torch.zeros_like(flat)
    ops.mlp_dw(net, save, dy, dsh, grads, rows, precision=P, gmax=gmax)
    torch.cuda.synchronize()
    sg = 2.0 ** (4 - math.ceil(math.log2(float(d_out.abs().max()))))
    sx = ops.SPLIT_SCALE_X
    ref = {}

    def gemm(a, b):          # a^T b on the GPU in fp64 (the workspaces are GBs: keep them there)
        return (a.double().t() @ b.double()).cpu()
    enc = ops.decode_frags_16(save.enc, 1, 64, rows, P)[0][:, :63] / sx
    dshv = ops.decode_frags_16(dsh, 1, 32, rows, P)[0] / sg
    act = lambda l: ops.decode_frags_16(save.act[l * (save.act.numel() // (D + 2)):(l + 1) * (save.act.numel() // (D + 2))], 1, W, rows, P)[0] / sx
    dyl = lambda l: ops.decode_frags_16(dy[l * (dy.numel() // (D + 2)):(l + 1) * (dy.numel() // (D + 2))], 1, W, rows, P)[0] / sg
    for l in range(D):
        x = enc if l == 0 else (torch.cat([enc, act(l - 1)], 1) if l == skip else act(l - 1))
        dyv = dyl(l)
        ref[f"xyz_encoding_{l + 1}.0.weight"] = gemm(dyv, x)
        ref[f"xyz_encoding_{l + 1}.0.bias"] = dyv.double().sum(0).cpu()
    ref["sigma.0.weight"], ref["sigma.0.bias"] = gemm(dyl(D), act(D - 1)), dyl(D).double().sum(0).cpu()
    ref["sh.0.weight"], ref["sh.0.bias"] = gemm(dyl(D + 1), act(D - 1)), dyl(D + 1).double().sum(0).cpu()
    ref["sh.2.weight"], ref["sh.2.bias"] = gemm(dshv[:, :27], act(D + 1)), dshv[:, :27].double().sum(0).cpu()
    ref["sigma.2.weight"], ref["sigma.2.bias"] = gemm(dshv[:, 27:28], act(D)), dshv[:, 27:28].double().sum(0).cpu()
    for off, shp, name in zip(ops.param_offsets(net), net.shapes(), net.names()):
        n = int(np.prod(shp))
        got = grads[off:off + n].view(shp).double().cpu()
        want = ref[name].view(shp)
        scale = float(want.abs().max())
        err = float((got - want).abs().max())
        # fp32 accumulation of 22-bit products without the lo x lo term: summation-order noise + 2^-21 per product
        assert math.isfinite(err) and err <= 2e-5 * max(scale, 1e-12) + 1e-9, f"{name}: err {err:.3e} scale {scale:.3e}"
