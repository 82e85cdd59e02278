"""GPU tests of the single-pass 16-bit MFMA mode ("f16" / "bf16", csrc/mcnerf_16.h) against the fp32 CPU oracle.

This mode does NOT claim the 1e-4 parity bar (that is the f32 / f16x3 modes): its operands are rounded to 11 (f16) or
8 (bf16) significant bits.  The tolerances below are this mode's own stated accuracy, and every test prints the error
it measured (run with -s to see them; tests/golden comparison in test_model_gpu.py).
"""
import numpy as np
import pytest
import torch

from oracle import mcnerf_oracle as O

pytestmark = pytest.mark.gpu

NETS = {32: O.NetCfg(4, 32, (2,)), 64: O.NetCfg(8, 64, (4,)), 128: O.NetCfg(4, 128, (2,)), 256: O.NetCfg(8, 256, (4,))}
# stated accuracy of the mode, relative to max(1, max|reference|): forward activations / outputs
TOL_FWD = {"f16": 4e-3, "bf16": 4e-2}


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


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(1.0, float(b.abs().max())))


def fwd_case(width, barf, N, S, seed=0):
    nc = NETS[width]
    cfg = O.RenderCfg(samples=S, scale=2, coarse=nc, fine=nc, barf_mode=barf, barf_start=0.3846, barf_end=0.6923)
    step_r = 0.5
    p = O.init_params(nc, 100 + width + seed)
    d, o = make_rays(N, 5 + width + seed)
    g = torch.Generator().manual_seed(1 + seed)
    jitter = torch.rand(N, 1, generator=g) * (cfg.far - cfg.near) / S
    zg = torch.linspace(cfg.near, cfg.far, S)
    z = zg.unsqueeze(0) + jitter
    xyz = (o.unsqueeze(1) + d.unsqueeze(1) * z.unsqueeze(2)).reshape(-1, 3)
    dirs = d.unsqueeze(1).expand(-1, S, -1).reshape(-1, 3)
    return nc, cfg, step_r, p, d, o, jitter, zg, xyz, dirs


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("width", [32, 64, 128, 256])
@pytest.mark.parametrize("barf", [False, True])
def test_mlp16_fwd_dense(gpu_device, width, barf, precision):
    ops = _ops()
    dev = gpu_device
    N, S = 37, 48           # ragged: N*S = 1776 is not a multiple of the 256-row pass
    nc, cfg, step_r, p, d, o, jitter, zg, xyz, dirs = fwd_case(width, barf, N, S)
    net = net_of(nc)
    ref, hidden, sh = O.mlp_forward(p, nc, O.embed(xyz, step_r, cfg), dirs, return_hidden=True)

    flat = flat_params(nc, p, dev)
    packed = ops.pack_weights(net, flat, precision=precision)
    out = torch.full((N, S, 4), float("nan"), device=dev)
    save = ops.alloc_save(net, N * S, dev, precision=precision)
    bw = O.barf_weights(step_r, cfg).to(dev)
    ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous(), bw,
                out, save=save, precision=precision)
    torch.cuda.synchronize()
    tol = TOL_FWD[precision]
    enc = ops.decode_frags_16(save.enc, 1, 64, N * S, precision)[0][:, :63]
    e_enc = relerr(enc, O.embed(xyz, step_r, cfg))
    assert e_enc < tol / 2, f"encoding: {e_enc}"
    act = ops.decode_frags_16(save.act, nc.depth + 2, width, N * S, precision)
    errs = []
    for l, h in enumerate(hidden):
        e = relerr(act[l], h)
        errs.append(e)
        assert e < tol, f"layer {l}: {e}"
    e_out = relerr(out.view(-1, 4), ref)
    print(f"[{precision} W={width} barf={barf}] enc {e_enc:.1e}  hidden max {max(errs):.1e}  out {e_out:.1e}")
    assert e_out < tol
    # the no-save instantiation gives the same output bit for bit
    out2 = torch.empty_like(out)
    ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous(), bw, out2,
                precision=precision)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("precision", ["f16", "bf16"])
def test_mlp16_fwd_indexed_multi_pass(gpu_device, precision):
    """Compacted (ray, sample) list with a device-side count, several passes per workgroup, entries outside the list
    left untouched."""
    ops = _ops()
    dev = gpu_device
    width, N, S = 128, 700, 64
    nc, cfg, step_r, p, d, o, jitter, zg, xyz, dirs = fwd_case(width, False, N, S, seed=3)
    net = net_of(nc)
    ref = O.mlp_forward(p, nc, O.embed(xyz, step_r, cfg), dirs).view(N, S, 4)
    g = torch.Generator().manual_seed(7)
    keep = torch.rand(N, S, generator=g) < 0.37
    idx = torch.nonzero(keep).to(torch.int32)                 # row-major (ray, sample) pairs
    K = idx.shape[0]
    cap = N * S
    idx_pad = torch.zeros(cap, 2, dtype=torch.int32)
    idx_pad[:K] = idx
    flat = flat_params(nc, p, dev)
    packed = ops.pack_weights(net, flat, precision=precision)
    out = torch.full((N, S, 4), -7.0, device=dev)
    ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous(),
                O.barf_weights(step_r, cfg).to(dev), out, idx=idx_pad.to(dev), count=torch.tensor([K], dtype=torch.int32, device=dev),
                max_rows=cap, precision=precision)
    torch.cuda.synchronize()
    out = out.cpu()
    assert torch.all(out[~keep] == -7.0)
    e = relerr(out[keep], ref[keep])
    print(f"[{precision}] indexed, {K} rows: out {e:.1e}")
    assert e < TOL_FWD[precision]
