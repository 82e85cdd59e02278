"""GPU parity tests of every C-ABI entry point against the CPU oracle on the same seeded inputs.

Tolerances: the north star asks for <= 1e-4 abs fp32 on rendered rgb/depth; the per-op checks below
are tighter (the kernels are exact-fp32 MFMA, only summation order differs from the CPU GEMMs).
"""
import math

import numpy as np
import pytest
import torch

from oracle import mcnerf_oracle as O

pytestmark = pytest.mark.gpu

NETS = {32: O.NetCfg(4, 32, (2,)), 64: O.NetCfg(8, 64, (4,)), 128: O.NetCfg(4, 128, (2,)), 256: O.NetCfg(8, 256, (4,))}


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


def test_layout_matches_header(gpu_device):
    ops = _ops()
    for nc in NETS.values():
        net = net_of(nc)
        offs = ops.param_offsets(net)
        assert all(o % 4 == 0 for o in offs)
        last = offs[-1] + 27
        assert ops.param_count(net) >= last
        assert ops.packed_count(net) > 0


@pytest.mark.parametrize("precision", ["f32"])       # (the f16x3 mode: tests/test_mlpx3_gpu.py)
@pytest.mark.parametrize("width", [32, 64, 128, 256])
@pytest.mark.parametrize("barf", [False, True])
def test_mlp_fwd_dense(gpu_device, width, barf, precision):
    ops = _ops()
    dev = gpu_device
    nc = NETS[width]
    net = net_of(nc)
    S, N = 48, 37          # ragged: N*S is not a multiple of the tile
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
    packed = ops.pack_weights(net, flat, precision=precision)
    out = torch.full((N, S, 4), float("nan"), device=dev)
    save = ops.alloc_save(net, N * S, dev)
    bw = O.barf_weights(step_r, cfg).to(dev)
    ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous(), bw,
                out, save=save, precision=precision)
    torch.cuda.synchronize()
    # layer by layer first (localises a failure), then the output
    dec = lambda t_: t_
    act = dec(save.act).view(nc.depth + 2, save.capacity, width)
    enc = dec(save.enc).view(save.capacity, 64)[:, :63]
    assert maxerr(enc, O.embed(xyz, step_r, cfg)) < 2e-6
    for l, h in enumerate(hidden):
        e = maxerr(act[l], h)
        assert e < 2e-5, f"layer {l}: {e}"
    assert maxerr(save.sh.view(-1, 32)[:, :27], sh) < 2e-5
    assert maxerr(out.view(-1, 4), ref) < 2e-5
    # the no-save instantiation gives the same output
    out2 = torch.empty_like(out)
    ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous(), bw, out2,
                precision=precision)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("S", [32, 64, 128, 160, 640])
def test_composite_fwd_bwd(gpu_device, S):
    ops = _ops()
    dev = gpu_device
    N = 53
    g = torch.Generator().manual_seed(S)
    sig_rgb = torch.cat([torch.randn(N, S, 1, generator=g) * 3.0, torch.rand(N, S, 3, generator=g)], -1)
    sig_rgb[:4, :, 0] = -20.0          # empty rays
    sig_rgb[4:8, : S // 2, 0] = 12.0    # opaque early
    d, _ = make_rays(N, 3)
    d = d * (0.5 + torch.rand(N, 1, generator=g))           # non-unit: depth path scales by |d|
    jitter = torch.rand(N, 1, generator=g) * 0.1
    zg = torch.linspace(1, 8, S)
    z = zg.unsqueeze(0) + jitter
    eps, eps2 = torch.randn(N, S, generator=g), torch.randn(N, S, generator=g)
    sr = sig_rgb.clone().requires_grad_(True)
    rgb, depth, opac, w = O.composite(sr, d, z, eps, True)
    w2 = O.sigma2weights(O.deltas_of(z), sig_rgb[..., 0], eps2)
    jd = jitter.reshape(-1).to(dev).contiguous()
    r, dp, op, ws, wmax = ops.composite_fwd(sig_rgb.to(dev), d.to(dev), zg.to(dev), jd, eps.to(dev), eps2.to(dev), True, True)
    assert maxerr(r, rgb) < 2e-6
    assert maxerr(dp, depth) < 2e-5
    assert maxerr(op, opac) < 2e-6
    assert maxerr(ws, w2) < 2e-6
    assert abs(wmax.view(torch.float32).item() - float(w2.max())) < 1e-6
    grgb = torch.randn(N, 3, generator=g)
    (rgb * grgb).sum().backward()
    dsr = ops.composite_bwd(sig_rgb.to(dev), zg.to(dev), jd, eps.to(dev), grgb.to(dev), True)
    assert maxerr(dsr, sr.grad) < 5e-6


def test_select_fine(gpu_device):
    ops = _ops()
    dev = gpu_device
    # (the scan gives every thread a run of a multiple of four rays: sizes around the run boundaries, a ragged tail, one ray,
    #  and more than 65536 rays -- the scalar path)
    for N, Sc, scale, seed in [(70, 64, 2, 0), (33, 32, 5, 1), (5, 128, 5, 2), (1, 16, 3, 3), (4097, 16, 2, 4), (7000, 64, 2, 5), (32768, 16, 1, 6), (70001, 8, 2, 7)]:
        g = torch.Generator().manual_seed(seed)
        w = torch.rand(N, Sc, generator=g) * 4e-3
        w[N // 2] = 0.0 if N > 1 else w[0]
        cfg = O.RenderCfg(samples=Sc, scale=scale)
        ref = O.select_fine(w, cfg)
        wmax = w.max().reshape(1).view(torch.int32).to(dev)
        idx, count, out_f = ops.select_fine(w.to(dev), wmax, cfg.weight_thresh, scale, cfg.sigma_default)
        k = int(count.item())
        assert k == ref.shape[0]
        assert torch.equal(idx[:k].cpu().long(), ref)
        assert torch.equal(out_f.cpu(), torch.tensor([cfg.sigma_default, 1.0, 1.0, 1.0]).expand(N, Sc * scale, 4))
    # max below the threshold: the threshold drops to the max (model/mc_nerf.py:623)
    w = torch.full((6, 32), 1e-5)
    w[2, 7] = 5e-4
    cfg = O.RenderCfg(samples=32, scale=2)
    ref = O.select_fine(w, cfg)
    idx, count, _ = ops.select_fine(w.to(dev), w.max().reshape(1).view(torch.int32).to(dev), 1e-3, 2, -20.0)
    assert int(count.item()) == 2 and torch.equal(idx[:2].cpu().long(), ref)
    # cap gather
    perm = torch.randperm(ref.shape[0])
    idx2, c2 = ops.cap_gather(idx, perm.to(dev), 1)
    assert int(c2.item()) == 1 and torch.equal(idx2.cpu().long(), ref[perm[:1]])


@pytest.mark.parametrize("precision", ["f32"])
@pytest.mark.parametrize("width", [32, 64, 128, 256])
def test_mlp_fwd_bwd_indexed(gpu_device, width, precision):
    """Fine-pass mode: (ray, sample) list + device count; forward, dX chain, dW against autograd."""
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
    packed = ops.pack_weights(net, flat, precision=precision)
    cap = K + 17
    idx_d = torch.zeros(cap, 2, dtype=torch.int32, device=dev)
    idx_d[:K] = idx.to(torch.int32).to(dev)
    count = torch.tensor([K], dtype=torch.int32, device=dev)
    out = torch.full((N, S, 4), 7.0, device=dev)
    save = ops.alloc_save(net, cap, dev)
    bw = O.barf_weights(step_r, cfg).to(dev)
    od, dd, zd, jd = o.detach().to(dev), d.detach().to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous()
    ops.mlp_fwd(net, flat, packed, od, dd, zd, jd, bw, out, idx=idx_d, count=count, max_rows=cap, save=save, precision=precision)
    got = out[r.to(dev), j.to(dev)]
    assert maxerr(got, ref) < 2e-5
    assert torch.all(out[~sel.to(dev)] == 7.0)          # untouched elsewhere

    d_out = torch.zeros(N, S, 4, device=dev)
    d_out[r.to(dev), j.to(dev)] = gout.to(dev)
    grads = torch.zeros_like(flat)
    dy = torch.empty_like(save.act)
    dsh = torch.empty_like(save.sh)
    d_o = torch.zeros(N, 3, device=dev)
    d_d = torch.zeros(N, 3, device=dev)
    gmax = d_out.abs().max().reshape(1).view(torch.int32)        # what composite_bwd hands to the split-f16 backward
    ops.mlp_bwd(net, flat, packed, od, dd, zd, jd, bw, out, d_out, save, dy, dsh, d_o, d_d,
                idx=idx_d, count=count, max_rows=cap, precision=precision, gmax=gmax)
    ops.mlp_dw(net, save, dy, dsh, grads, cap, count=count, precision=precision, gmax=gmax)
    torch.cuda.synchronize()
    scale_o = float(o.grad.abs().max())
    assert maxerr(d_o, o.grad) < 2e-5 * max(1.0, scale_o), (maxerr(d_o, o.grad), scale_o)
    assert maxerr(d_d, d.grad) < 2e-5 * max(1.0, float(d.grad.abs().max()))
    for off, shp, name in zip(ops.param_offsets(net), net.shapes(), net.names()):
        n = int(np.prod(shp))
        gg = grads[off:off + n].view(shp)
        ref_g = p[name].grad
        tol = 2e-5 * max(1.0, float(ref_g.abs().max()))
        e = maxerr(gg, ref_g)
        assert e < tol, f"{name}: {e} (tol {tol})"


@pytest.mark.parametrize("precision", ["f32"])
@pytest.mark.parametrize("width", [32, 64, 128, 256])
def test_mlp_dw_at_scale(gpu_device, width, precision):
    """The persistent dW kernels over ~0.4 M rows (thousands of slabs per workgroup, all ring/DMA paths busy) against
    a torch fp64 GEMM of the very operands they read (the saved activations and dY written by the HIP fwd/bwd)."""
    ops = _ops()
    dev = gpu_device
    nc = NETS[width]
    net = net_of(nc)
    D, W, skip = nc.depth, nc.width, nc.skips[0]
    N, S = 3001, 128                     # ragged row count
    rows = N * S
    p = O.init_params(nc, 300 + width)
    flat = flat_params(nc, p, dev)
    packed = ops.pack_weights(net, flat, precision=precision)
    d, o = make_rays(N, 77)
    od, dd, zd = o.to(dev), d.to(dev), torch.linspace(1, 8, S, device=dev)
    bw = torch.ones(10, device=dev)
    out = torch.empty(N, S, 4, device=dev)
    save = ops.alloc_save(net, rows, dev)
    ops.mlp_fwd(net, flat, packed, od, dd, zd, None, bw, out, save=save, precision=precision)
    d_out = torch.randn(N, S, 4, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 1e-3
    gmax = d_out.abs().max().reshape(1).view(torch.int32)
    dy, dsh = torch.empty_like(save.act), torch.empty_like(save.sh)
    d_o, d_d = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
    ops.mlp_bwd(net, flat, packed, od, dd, zd, None, bw, out, d_out, save, dy, dsh, d_o, d_d, precision=precision, gmax=gmax)
    grads = torch.zeros_like(flat)
    ops.mlp_dw(net, save, dy, dsh, grads, rows, precision=precision, gmax=gmax)
    torch.cuda.synchronize()
    act, enc, dyv, dshv = save.act, save.enc, dy, dsh
    act = act.view(D + 2, rows, W).double()
    dyv = dyv.view(D + 2, rows, W).double()
    enc = enc.view(rows, 64)[:, :63].double()
    dshv = dshv.view(rows, 32).double()
    ref = {}
    for l in range(D):
        x = enc if l == 0 else (torch.cat([enc, act[l - 1]], 1) if l == skip else act[l - 1])
        ref[f"xyz_encoding_{l + 1}.0.weight"] = dyv[l].t() @ x
        ref[f"xyz_encoding_{l + 1}.0.bias"] = dyv[l].sum(0)
    ref["sigma.0.weight"], ref["sigma.0.bias"] = dyv[D].t() @ act[D - 1], dyv[D].sum(0)
    ref["sh.0.weight"], ref["sh.0.bias"] = dyv[D + 1].t() @ act[D - 1], dyv[D + 1].sum(0)
    ref["sh.2.weight"], ref["sh.2.bias"] = dshv[:, :27].t() @ act[D + 1], dshv[:, :27].sum(0)
    ref["sigma.2.weight"], ref["sigma.2.bias"] = dshv[:, 27:28].t() @ act[D], dshv[:, 27:28].sum(0)
    for off, shp, name in zip(ops.param_offsets(net), net.shapes(), net.names()):
        n = int(np.prod(shp))
        got = grads[off:off + n].view(shp).double()
        want = ref[name].view(shp)
        scale = float(want.abs().max())
        err = float((got - want).abs().max())
        assert math.isfinite(err) and err <= 2e-5 * max(scale, 1e-12) + 1e-9, f"{name}: err {err:.3e} scale {scale:.3e}"


def test_raygen(gpu_device):
    ops = _ops()
    dev = gpu_device
    H, W = 30, 40
    g = torch.Generator().manual_seed(5)
    wu = torch.randn(1, 6, generator=g) * 0.6
    pose = O.se3_to_SE3(wu)[0].clone().requires_grad_(True)
    K = torch.tensor([[55.0, 0, 20.3], [0, 52.0, 14.1], [0, 0, 1]])
    kinv = K.inverse().contiguous().clone().requires_grad_(True)
    pix = torch.randperm(H * W, generator=g)[:257]
    d_ref, o_ref = O.get_rays(pose, kinv, H, W)
    d_ref, o_ref = d_ref[pix], o_ref[pix]
    d, o = ops.raygen_fwd(pose.detach().to(dev), kinv.detach().to(dev), pix.to(dev), W)
    assert maxerr(d, d_ref) < 5e-7
    assert maxerr(o, o_ref) < 5e-7
    gd, go = torch.randn(257, 3, generator=g), torch.randn(257, 3, generator=g)
    ((d_ref * gd).sum() + (o_ref * go).sum()).backward()
    d_pose, d_kinv = ops.raygen_bwd(pose.detach().to(dev), kinv.detach().to(dev), pix.to(dev), W, gd.to(dev), go.to(dev))
    assert maxerr(d_pose, pose.grad) < 1e-4 * max(1.0, float(pose.grad.abs().max()))
    assert maxerr(d_kinv, kinv.grad) < 1e-4 * max(1.0, float(kinv.grad.abs().max()))


def test_fused_radam_matches_host_arithmetic(gpu_device):
    """mcnerf_radam_step (one launch, many tensors) against the per-tensor torch implementation of the same
    update rule, through the SGD-like warm-up (N_sma < 5) and the rectified regime, with weight decay."""
    from mc_nerf_amd.model import RAdam
    dev = gpu_device
    g = torch.Generator().manual_seed(0)
    shapes = [(256, 319), (256,), (1, 256), (1,), (27, 256), (110, 6), (5000,)]
    cpu = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    gpu = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in cpu]
    oc = RAdam(cpu, lr=3e-3, weight_decay=4e-4)
    og = RAdam(gpu, lr=3e-3, weight_decay=4e-4)
    for step in range(9):
        for pc, pg in zip(cpu, gpu):
            gr = torch.randn(pc.shape, generator=g)
            pc.grad, pg.grad = gr, gr.to(dev)
        if step == 4:                      # a tensor without a gradient keeps its own step count
            cpu[2].grad = None
            gpu[2].grad = None
        oc.step()
        og.step()
    for pc, pg in zip(cpu, gpu):
        assert maxerr(pg, pc) < 1e-6
    assert og.state[gpu[2]]["step"] == 8 and og.state[gpu[0]]["step"] == 9


def test_fused_radam_overflow_guard(gpu_device):
    """A non-finite gradient (operand range of a reduced-precision mode exceeded) must not reach the parameters: the fused
    step is skipped as a whole (parameters and moments untouched), counted, and reported loudly on request."""
    from mc_nerf_amd.model import RAdam
    from mc_nerf_amd import _lib
    dev = gpu_device
    g = torch.Generator().manual_seed(3)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).to(dev)) for s in ((300, 7), (5000,), (3,))]
    opt = RAdam(ps, lr=1e-2, weight_decay=1e-3)
    for p in ps:
        p.grad = torch.randn_like(p)
    opt.step()
    before = [p.detach().clone() for p in ps]
    m_before = [opt.state[p]["exp_avg"].clone() for p in ps]
    for p in ps:
        p.grad = torch.randn_like(p)
    ps[1].grad[4321] = float("nan")
    opt.step()                                        # skipped
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, ps))
    assert all(torch.equal(a, opt.state[p]["exp_avg"]) for a, p in zip(m_before, ps))
    assert opt.skipped_steps() == 1
    ps[1].grad[4321] = float("inf")
    opt.step()
    assert opt.skipped_steps() == 2
    with pytest.raises(_lib.McnerfError):
        opt.raise_on_overflow()
    for p in ps:
        p.grad = torch.randn_like(p)
    opt.step()                                        # finite again: the update goes through
    assert not torch.equal(before[0], ps[0].detach()) and opt.skipped_steps() == 2
    assert all(bool(torch.isfinite(p).all()) for p in ps)


def test_camera_parametrisation_fwd_bwd(gpu_device):
    """Fused camera kernel vs the oracle (itself pinned to the reference by golden G9), incl. abs() on a negative
    multiplier, a zero rotation (theta = 0) and a near-pi rotation."""
    ops = _ops()
    dev = gpu_device
    C, H, W = 9, 600, 800
    g = torch.Generator().manual_seed(4)
    wp = (torch.randn(C, 6, generator=g) * 0.8)
    wp[1, :3] = 0.0
    wp[2, :3] = torch.tensor([0.0, 3.1, 0.2])
    wpi = torch.randn(C, 6, generator=g) * 0.5
    ws = [1.0 + 0.3 * torch.randn(C, generator=g) for _ in range(4)]
    ws[0][3] = -0.8
    leaves = [t.clone().requires_grad_(True) for t in [wp, wpi] + ws]
    K = O.intrinsics_from_weights(H, W, *leaves[2:])
    pose, calib = O.se3_to_SE3(leaves[0]), O.se3_to_SE3(leaves[1])
    Kinv = O.inverse_intrinsic(K)
    gK, gKi = torch.randn(C, 3, 3, generator=g), torch.randn(C, 3, 3, generator=g) * 100.0
    gp, gc = torch.randn(C, 3, 4, generator=g), torch.randn(C, 3, 4, generator=g)
    ((K * gK).sum() + (Kinv * gKi).sum() + (pose * gp).sum() + (calib * gc).sum()).backward()
    dl = [t.detach().to(dev) for t in leaves]
    K2, Ki2, p2, c2, _, _ = ops.camera_fwd(*dl, H, W)
    assert maxerr(K2, K) < 1e-4 and maxerr(p2, pose) < 2e-6 and maxerr(c2, calib) < 2e-6
    assert maxerr(Ki2, Kinv) < 1e-7 + 1e-6 * float(Kinv.abs().max())
    grads = ops.camera_bwd(*dl, H, W, gK.to(dev), gKi.to(dev), gp.to(dev), gc.to(dev))
    for got, leaf in zip(grads, leaves):
        ref = leaf.grad
        assert maxerr(got, ref) < 2e-5 * max(1.0, float(ref.abs().max())), (maxerr(got, ref), float(ref.abs().max()))
    # null upstreams are allowed
    g2 = ops.camera_bwd(*dl, H, W, None, None, gp.to(dev), None)
    assert float(g2[1].abs().max()) == 0.0 and float(g2[2].abs().max()) == 0.0 and maxerr(g2[0], leaves[0].grad) > 0


def test_camera_reprojection_branch_and_loss(gpu_device):
    """SURVEY 8f row f1, second half: the calibration reprojection (get_reproject_pixels, model/mc_nerf.py:147-152, 236-267)
    inside the fused camera kernels and the reprojection loss kernel (model/loss.py:45-58), forward and backward, against
    autograd through the oracle's camera parametrisation + the reference formulas in tensor ops."""
    ops = _ops()
    dev = gpu_device
    C, P, H, W = 11, 5, 600, 800
    g = torch.Generator().manual_seed(8)
    wp = torch.randn(C, 6, generator=g) * 0.3
    wp[:, 5] += 4.0                                             # points end up in front of the cameras
    wpi = torch.randn(C, 6, generator=g) * 0.3
    wpi[:, 5] += 4.0
    ws = [1.0 + 0.2 * torch.randn(C, generator=g) for _ in range(4)]
    leaves = [t.clone().requires_grad_(True) for t in [wp, wpi] + ws]
    wi = torch.randn(C, P, 3, generator=g) * 0.5
    we = torch.randn(C, P, 3, generator=g) * 0.5
    gt_i, gt_e = torch.rand(1, C, P, 2, generator=g) * 800, torch.rand(1, C, P, 2, generator=g) * 600

    def project(wpts, K, pose):                                 # the reference's tensor ops (:236-267)
        wh = torch.cat([wpts, torch.ones_like(wpts[..., :1])], -1)
        cam = wh @ pose.transpose(-2, -1)
        pix = cam @ K.transpose(-2, -1)
        return pix[..., :2] / pix[..., 2:]

    def loss_ref(pd, gt):
        return ((pd[..., 0] - gt[..., 0]) / W).pow(2).mean() + ((pd[..., 1] - gt[..., 1]) / H).pow(2).mean()
    K = O.intrinsics_from_weights(H, W, *leaves[2:])
    pose, calib = O.se3_to_SE3(leaves[0]), O.se3_to_SE3(leaves[1])
    pi_ref, pe_ref = project(wi, K, calib), project(we, K, pose)
    l_ref = loss_ref(pi_ref.unsqueeze(0), gt_i) + 3.0 * loss_ref(pe_ref.unsqueeze(0), gt_e)
    l_ref.backward()

    from mc_nerf_amd.model.render import CameraFn, ReprojLossFn
    dl = [t.detach().to(dev).requires_grad_(True) for t in leaves]
    _, _, _, _, pi, pe = CameraFn.apply(*dl, H, W, wi.to(dev), we.to(dev))
    assert maxerr(pi, pi_ref) < 2e-3 and maxerr(pe, pe_ref) < 2e-3       # pixels (values of several hundred): ~1e-6 relative
    l = ReprojLossFn.apply(pi.unsqueeze(0), gt_i.to(dev), H, W) + 3.0 * ReprojLossFn.apply(pe.unsqueeze(0), gt_e.to(dev), H, W)
    assert abs(float(l) - float(l_ref)) < 1e-5 * max(1.0, abs(float(l_ref)))
    l.backward()
    for got, leaf in zip(dl, leaves):
        ref = leaf.grad
        assert maxerr(got.grad, ref) < 5e-5 * max(1.0, float(ref.abs().max())), (maxerr(got.grad, ref), float(ref.abs().max()))
    # only one branch requested
    _, _, _, _, pi2, pe2 = CameraFn.apply(*[t.detach() for t in dl], H, W, wi.to(dev), None)
    assert pe2 is None and torch.equal(pi2, pi.detach())


def test_gather_gt_from_device_resident_images(gpu_device):
    """uint8 RGBA / RGB images in HBM -> fp32 GT of selected pixels, against the reference loader's arithmetic
    (ToTensor's /255 then rgb*a + (1-a), data/data_read.py:130-137)."""
    from mc_nerf_amd.data import DeviceImageSet
    dev = gpu_device
    g = torch.Generator().manual_seed(0)
    H, W = 20, 30
    for ch in (4, 3):
        u8 = torch.randint(0, 256, (3, H * W, ch), dtype=torch.uint8, generator=g)
        ds = DeviceImageSet(u8.to(dev), H, W)
        pix = torch.randperm(H * W, generator=g)[:257]
        f = u8[1].float() / 255.0
        ref = f[:, :3] * f[:, 3:] + (1 - f[:, 3:]) if ch == 4 else f
        got = ds.gather(1, pix.to(dev))
        assert maxerr(got, ref[pix]) == 0.0
        assert maxerr(ds.full_image(2), (u8[2].float() / 255.0)[:, :3] * ((u8[2].float() / 255.0)[:, 3:] if ch == 4 else 1)
                      + ((1 - u8[2].float()[:, 3:] / 255.0) if ch == 4 else 0)) < 1e-7


def test_cap_random_device_side(gpu_device):
    """The device-side random cap (model/mc_nerf.py:630-632 without the host sync): exactly `keep` distinct entries of the
    first *count, every entry equally likely, everything copied when the cap does not bind."""
    ops = _ops()
    dev = gpu_device
    K, cap, keep = 50000, 70000, 20000
    idx = torch.stack([torch.arange(cap, dtype=torch.int32) // 97, torch.arange(cap, dtype=torch.int32) % 97], 1).to(dev)
    count = torch.tensor([K], dtype=torch.int32, device=dev)
    hits = torch.zeros(K, device=dev)
    trials = 64
    for t in range(trials):
        seed = torch.tensor([1234567 * t + 11], dtype=torch.int32, device=dev)
        idx2, c2 = ops.cap_random(idx, count, cap, keep, seed)
        assert int(c2.item()) == keep
        flat = idx2[:, 0].long() * 97 + idx2[:, 1].long()
        assert int(flat.max()) < K and torch.unique(flat).numel() == keep            # a subset of the valid entries, no duplicates
        assert bool((flat[1:] > flat[:-1]).all())                                     # ... in the order of the list
        hits[flat] += 1
    # every entry is kept with probability keep / K = 0.4: the per-entry hit count is Binomial(64, 0.4)
    mean = float(hits.mean()) / trials
    assert abs(mean - keep / K) < 1e-6
    assert float(hits.max()) <= 48 and float(hits.min()) >= 6                        # ~ +-5.7 sigma
    halves = hits[: K // 2].mean() / hits[K // 2:].mean()
    assert abs(float(halves) - 1.0) < 0.01                                           # no positional bias
    # same seed -> the same list, entry for entry (no atomic decides a position)
    seed = torch.tensor([42], dtype=torch.int32, device=dev)
    a, _ = ops.cap_random(idx, count, cap, keep, seed)
    for _ in range(3):
        b, _ = ops.cap_random(idx, count, cap, keep, seed)
        assert torch.equal(a, b)
    # cap not binding: the first *count entries as they stand (the reference's order, model/mc_nerf.py:625-632)
    for n_small in (1500, 1, keep):
        small = torch.tensor([n_small], dtype=torch.int32, device=dev)
        idx3, c3 = ops.cap_random(idx, small, cap, keep, seed)
        assert int(c3.item()) == n_small and torch.equal(idx3[:n_small], idx[:n_small])
    # one more than fits: exactly one entry dropped, the rest in order
    over = torch.tensor([keep + 1], dtype=torch.int32, device=dev)
    idx4, c4 = ops.cap_random(idx, over, cap, keep, seed)
    f4 = idx4[:, 0].long() * 97 + idx4[:, 1].long()
    assert int(c4.item()) == keep and int(f4.max()) <= keep and bool((f4[1:] > f4[:-1]).all())


def test_sample_perm_is_a_random_subset_without_replacement(gpu_device):
    """mcnerf_sample_perm = randperm(n)[:batch] of model/mc_nerf.py:329 in one kernel: distinct ids in range, every pixel
    equally likely, random order, reproducible from the seed word; batch == n gives a full permutation."""
    ops = _ops()
    dev = gpu_device
    n, batch = 800 * 800, 7000
    hits = torch.zeros(n, device=dev)
    first = torch.zeros(n, device=dev)
    trials = 400
    for t in range(trials):
        seed = torch.tensor([7919 * t + 3], dtype=torch.int32, device=dev)
        idx = ops.sample_perm(n, batch, dev, seed)
        assert idx.dtype == torch.int64 and idx.shape == (batch,)
        assert int(idx.min()) >= 0 and int(idx.max()) < n and torch.unique(idx).numel() == batch
        hits[idx] += 1
        first[idx[:100]] += 1
    # pixel hit counts ~ Binomial(400, 7000 / 640000): mean 4.375, sd 2.08
    assert abs(float(hits.mean()) - trials * batch / n) < 1e-9
    assert float(hits.max()) <= 22
    # no positional structure: quarters of the image and even/odd pixels are hit equally (sd of a quarter's share ~ 0.0006)
    q = hits.view(4, -1).sum(1) / hits.sum()
    assert float((q - 0.25).abs().max()) < 0.004
    assert abs(float(hits[0::2].sum() / hits.sum()) - 0.5) < 0.004
    # the ORDER is random too: the first 100 ids of a draw are spread like the whole draw
    qf = first.view(4, -1).sum(1) / first.sum()
    assert float((qf - 0.25).abs().max()) < 0.02
    # seeded: same word -> same draw, another word -> another draw
    s1 = torch.tensor([42], dtype=torch.int32, device=dev)
    a, b = ops.sample_perm(n, batch, dev, s1), ops.sample_perm(n, batch, dev, s1)
    c = ops.sample_perm(n, batch, dev, torch.tensor([43], dtype=torch.int32, device=dev))
    assert torch.equal(a, b) and not torch.equal(a, c)
    # a full permutation, also for a non-power-of-two n and n = 1
    for m in (1, 2, 1000, 12345):
        full = ops.sample_perm(m, m, dev, s1)
        assert torch.equal(torch.sort(full).values, torch.arange(m, device=dev))
    # torch.manual_seed drives the default seed word
    torch.manual_seed(5); x = ops.sample_perm(n, 64, dev)
    torch.manual_seed(5); y = ops.sample_perm(n, 64, dev)
    assert torch.equal(x, y)


def test_upload_f32_passes_host_values_bit_exactly(gpu_device):
    """mcnerf_upload_f32: host floats travel as kernel arguments (no host-device copy) and arrive bit-exactly."""
    ops = _ops()
    for n in (1, 10, 16):
        hv = torch.randn(n) * 10 ** torch.randint(-20, 20, (n,)).float()
        out = ops.upload_f32(hv, gpu_device)
        assert out.dtype == torch.float32 and out.is_cuda
        assert torch.equal(out.cpu().view(torch.int32), hv.view(torch.int32))
    emb_w = torch.tensor([1.0, 1.0, 0.75, 0.25, 0, 0, 0, 0, 0, 0])
    assert torch.equal(ops.upload_f32(emb_w, gpu_device).cpu(), emb_w)


@pytest.mark.parametrize("normalise", [True, False])
@pytest.mark.parametrize("with_fine", [True, False])
def test_fused_train_loss_matches_eager_formulation(gpu_device, normalise, with_fine):
    """mcnerf_train_loss / TrainLossFn = MC_NeRF_Loss.forward for the keys {"intr", "rgb"} (model/loss.py:13-31): value and the
    gradients wrt the reprojected pixels and both renders against the eager torch formulation, incl. an upstream factor."""
    from mc_nerf_amd.model.render import TrainLossFn
    dev = gpu_device
    g = torch.Generator().manual_seed(3)
    H, W, n = 800, 600, 7001
    pd = (torch.rand(110, 5, 2, generator=g) * 700).to(dev).requires_grad_(True)
    ptg = (torch.rand(110, 5, 2, generator=g) * 700).to(dev)
    rc = torch.rand(n, 3, generator=g).to(dev).requires_grad_(True)
    rf = torch.rand(n, 3, generator=g).to(dev).requires_grad_(True) if with_fine else None
    gt = torch.rand(n, 3, generator=g).to(dev)
    total = TrainLossFn.apply(pd, ptg, rc, rf, gt, H, W, normalise)
    (total * 0.37).backward()
    got = [pd.grad.clone(), rc.grad.clone(), rf.grad.clone() if with_fine else None]
    for t in (pd, rc) + ((rf,) if with_fine else ()):
        t.grad = None
    F = torch.nn.functional
    li = F.mse_loss(pd[..., 0] / W, ptg[..., 0] / W) + F.mse_loss(pd[..., 1] / H, ptg[..., 1] / H)
    ref = (li / (li.detach() + 1e-8) if normalise else li) + F.mse_loss(rc, gt) + (F.mse_loss(rf, gt) if with_fine else 0.0)
    (ref * 0.37).backward()
    assert abs(float(total) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    for a, b in zip(got, [pd.grad, rc.grad, rf.grad if with_fine else None]):
        if b is not None:
            assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-12


def test_fused_radam_guard_refuses_the_whole_step(gpu_device):
    """An optimiser step that spans several fused launches -- two param groups (the NeRF nets and the cameras), tensors with
    different step counts, more than 64 tensors -- is refused AS A WHOLE: a non-finite gradient in the LAST launch's tensors leaves
    the first launch's parameters and moments untouched too, and the refused step is counted once (not once per launch)."""
    from mc_nerf_amd.model import RAdam
    dev = gpu_device
    g = torch.Generator().manual_seed(5)
    many = [torch.nn.Parameter(torch.randn(37, generator=g).to(dev)) for _ in range(70)]        # > 64 tensors: two chunks of one call
    cams = [torch.nn.Parameter(torch.randn(110, 6, generator=g).to(dev)), torch.nn.Parameter(torch.randn(110, generator=g).to(dev))]
    opt = RAdam([{"params": many}, {"params": cams, "lr": 1e-3}], lr=1e-2, weight_decay=1e-3)

    def grads(skip=None):
        for i, p in enumerate(many + cams):
            p.grad = None if i == skip else torch.randn_like(p)
    grads(skip=3)
    opt.step()                                        # tensor 3 sits out one step: its step count differs from now on (its own launch)
    grads()
    opt.step()
    before = [p.detach().clone() for p in many + cams]
    m_before = [opt.state[p]["exp_avg"].clone() for p in many + cams]
    grads()
    cams[1].grad[17] = float("nan")                   # the last tensor of the last group
    opt.step()
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, many + cams))
    assert all(torch.equal(a, opt.state[p]["exp_avg"]) for a, p in zip(m_before, many + cams))
    assert opt.skipped_steps() == 1
    grads()
    many[68].grad[0] = float("inf")                   # in the second chunk of the first group: the first chunk must not have moved
    opt.step()
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, many + cams)) and opt.skipped_steps() == 2
    grads()
    opt.step()                                        # finite again
    assert all(not torch.equal(a, p.detach()) for a, p in zip(before, many + cams)) and opt.skipped_steps() == 2
