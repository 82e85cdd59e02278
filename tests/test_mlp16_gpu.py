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


# stated accuracy of the gradients (16-bit operands, fp32 accumulation): max|err| relative to the tensor's max|reference|
# when the reference uses the same ReLU decisions, and relative L2 against the plain fp32 oracle (a pre-activation
# within the operand rounding of zero can fall on either side: its gradient is discontinuous there)
TOL_GRAD = {"f16": 4e-3, "bf16": 4e-2}
TOL_GRAD_L2 = {"f16": 1e-1, "bf16": 3e-1}       # few hundred rows: a handful of flipped ReLUs dominate


def masked_forward(p, nc, x_enc, dirs, masks):
    """CorseFine_NeRF.forward (model/net_block.py:67-78) with the ReLU decisions supplied (masks[slot] bool [rows, W]):
    identical to the oracle's forward wherever the decisions agree."""
    F = torch.nn.functional
    h = x_enc
    for i in range(nc.depth):
        if i in nc.skips:
            h = torch.cat([x_enc, h], dim=-1)
        h = F.linear(h, p[f"xyz_encoding_{i+1}.0.weight"], p[f"xyz_encoding_{i+1}.0.bias"]) * masks[i]
    hs = F.linear(h, p["sigma.0.weight"], p["sigma.0.bias"]) * masks[nc.depth]
    sigma = F.linear(hs, p["sigma.2.weight"], p["sigma.2.bias"])
    hc = F.linear(h, p["sh.0.weight"], p["sh.0.bias"]) * masks[nc.depth + 1]
    sh = F.linear(hc, p["sh.2.weight"], p["sh.2.bias"])
    rgb = torch.sigmoid(O.eval_sh_deg2(sh.reshape(-1, 3, 9), dirs))
    return torch.cat([sigma, rgb], dim=-1)


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("width", [32, 64, 128, 256])
def test_mlp16_fwd_bwd_indexed(gpu_device, width, precision):
    """Fine-pass mode: (ray, sample) list + device count; forward, dX chain (ray gradients), dW against autograd of the
    oracle -- (a) with the kernel's own ReLU decisions (tight), (b) plain oracle (relative L2)."""
    ops = _ops()
    dev = gpu_device
    nc = NETS[width]
    net = net_of(nc)
    S, N = 40, 29
    cfg = O.RenderCfg(samples=20, scale=2, coarse=nc, fine=nc, barf_mode=True, barf_start=0.2, barf_end=0.9)
    step_r = 0.6
    p0 = O.init_params(nc, 200 + width)
    d0, o0 = make_rays(N, 9 + width)
    g = torch.Generator().manual_seed(2)
    jitter = torch.rand(N, 1, generator=g) * 0.2
    zg = torch.linspace(cfg.near, cfg.far, S)
    sel = torch.rand(N, S, generator=g) < 0.6
    idx = torch.nonzero(sel)
    K = idx.shape[0]
    z = zg.unsqueeze(0) + jitter
    r, j = idx[:, 0], idx[:, 1]
    gout = torch.randn(K, 4, generator=g) * 1e-4            # gradient magnitudes of a mean-reduced loss

    def reference(masks):
        p = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
        d, o = d0.clone().requires_grad_(True), o0.clone().requires_grad_(True)
        xyz = o[r] + d[r] * z[r, j].unsqueeze(-1)
        enc = O.embed(xyz, step_r, cfg)
        ref = O.mlp_forward(p, nc, enc, d[r]) if masks is None else masked_forward(p, nc, enc, d[r], masks)
        (ref * gout).sum().backward()
        return ref.detach(), p, d.grad, o.grad

    flat = flat_params(nc, p0, dev)
    packed = ops.pack_weights(net, flat, precision=precision)
    cap = K + 17
    idx_d = torch.zeros(cap, 2, dtype=torch.int32, device=dev)
    idx_d[:K] = idx.to(torch.int32).to(dev)
    count = torch.tensor([K], dtype=torch.int32, device=dev)
    out = torch.full((N, S, 4), 7.0, device=dev)
    save = ops.alloc_save(net, cap, dev, precision=precision)
    bw = O.barf_weights(step_r, cfg).to(dev)
    od, dd, zd, jd = o0.to(dev), d0.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous()
    ops.mlp_fwd(net, flat, packed, od, dd, zd, jd, bw, out, idx=idx_d, count=count, max_rows=cap, save=save, precision=precision)
    ref, _, _, _ = reference(None)
    got = out[r.to(dev), j.to(dev)]
    assert relerr(got, ref) < TOL_FWD[precision]
    assert torch.all(out[~sel.to(dev)] == 7.0)

    d_out = torch.zeros(N, S, 4, device=dev)
    d_out[r.to(dev), j.to(dev)] = gout.to(dev)
    grads = torch.zeros_like(flat)
    dy, dsh = ops.alloc_grad_ws(net, save, precision)
    d_o = torch.zeros(N, 3, device=dev)
    d_d = torch.zeros(N, 3, device=dev)
    gmax = d_out.abs().max().reshape(1).view(torch.int32)
    ops.mlp_bwd(net, flat, packed, od, dd, zd, jd, bw, out, d_out, save, dy, dsh, d_o, d_d,
                idx=idx_d, count=count, max_rows=cap, precision=precision, gmax=gmax)
    ops.mlp_dw(net, save, dy, dsh, grads, cap, count=count, precision=precision, gmax=gmax)
    torch.cuda.synchronize()

    masks = ops.decode_masks_16(save.mask, nc.depth + 2, width, K)
    _, hidden, _ = O.mlp_forward(p0, nc, O.embed(o0[r] + d0[r] * z[r, j].unsqueeze(-1), step_r, cfg), d0[r], return_hidden=True)
    flips = sum(int((masks[l] != (h > 0)).sum()) for l, h in enumerate(hidden))
    total = sum(h.numel() for h in hidden)

    def rel(a, b):
        return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max() / max(1e-30, float(b.abs().max())))

    def l2(a, b):
        a, b = a.detach().cpu().double(), b.detach().cpu().double()
        return float((a - b).norm() / max(1e-30, float(b.norm())))
    report = []
    for tag, mk, metric, tol in (("same ReLU decisions", [m.float() for m in masks], rel, TOL_GRAD[precision]),
                                 ("plain oracle, rel L2", None, l2, TOL_GRAD_L2[precision])):
        _, p, dg, og = reference(mk)
        e_o, e_d = metric(d_o, og), metric(d_d, dg)
        worst, worst_name = 0.0, ""
        for off, shp, name in zip(ops.param_offsets(net), net.shapes(), net.names()):
            n = int(np.prod(shp))
            e = metric(grads[off:off + n].view(shp), p[name].grad)
            if e > worst:
                worst, worst_name = e, name
        report.append(f"{tag}: d_o {e_o:.1e} d_d {e_d:.1e} worst dW {worst:.1e} ({worst_name})")
        assert e_o < tol and e_d < tol and worst < tol, (tag, e_o, e_d, worst_name, worst)
    print(f"[{precision} W={width}] ReLU flips {flips}/{total} | " + " | ".join(report))


@pytest.mark.parametrize("precision", ["f16"])
@pytest.mark.parametrize("width", [128, 256])
def test_mlp16_dw_at_scale(gpu_device, width, precision):
    """The persistent dW kernel over ~0.4 M rows (thousands of tiles per workgroup, ring / DMA / transposed reads busy)
    against a torch fp64 GEMM of the very operands it reads (decoded fragment-major workspaces)."""
    import math
    ops = _ops()
    dev = gpu_device
    nc = NETS[width]
    net = net_of(nc)
    D, W, skip = nc.depth, nc.width, nc.skips[0]
    N, S = 3001, 128
    rows = N * S
    p = O.init_params(nc, 300 + width)
    flat = flat_params(nc, p, dev)
    packed = ops.pack_weights(net, flat, precision=precision)
    d, o = make_rays(N, 77)
    od, dd, zd = o.to(dev), d.to(dev), torch.linspace(1, 8, S, device=dev)
    bw = torch.ones(10, device=dev)
    out = torch.empty(N, S, 4, device=dev)
    save = ops.alloc_save(net, rows, dev, precision=precision)
    ops.mlp_fwd(net, flat, packed, od, dd, zd, None, bw, out, save=save, precision=precision)
    d_out = torch.randn(N, S, 4, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 1e-3
    gmax = d_out.abs().max().reshape(1).view(torch.int32)
    dy, dsh = ops.alloc_grad_ws(net, save, precision)
    d_o, d_d = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
    ops.mlp_bwd(net, flat, packed, od, dd, zd, None, bw, out, d_out, save, dy, dsh, d_o, d_d, precision=precision, gmax=gmax)
    grads = torch.zeros_like(flat)
    ops.mlp_dw(net, save, dy, dsh, grads, rows, precision=precision, gmax=gmax)
    torch.cuda.synchronize()
    sg = 2.0 ** (4 - math.ceil(math.log2(float(d_out.abs().max()))))
    act = ops.decode_frags_16(save.act, D + 2, W, rows, precision).double()
    dyv = ops.decode_frags_16(dy, D + 2, W, rows, precision).double() / sg
    enc = ops.decode_frags_16(save.enc, 1, 64, rows, precision)[0][:, :63].double()
    dshv = ops.decode_frags_16(dsh, 1, 32, rows, precision)[0].double() / sg
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
        # fp32 accumulation of exact 16-bit products: only summation-order noise
        assert math.isfinite(err) and err <= 2e-5 * max(scale, 1e-12) + 1e-9, f"{name}: err {err:.3e} scale {scale:.3e}"


# ---------------------------------------------------------------------------------------------------------------------
# The 16-bit modes against the REFERENCE's golden renders (tests/golden, generated from /root/reference by
# tests/golden/make_golden.py).  Stated accuracy of the modes on these fixtures = about 3x what they measure on MI355X:
#   f16 : rgb / opacity <= 1.2e-5, depth <= 6.0e-6, gradients <= 4.0e-2 relative L2 (worst of the 40 tensors and of the ray gradients)
#   bf16: rgb / opacity <= 1.2e-4, depth <= 3.5e-5, gradients <= 9.6e-2
# (the fixtures are 48-96 rays on 32- / 64-wide nets, so a single ReLU decision that falls the other way under the operand
# rounding is visible in a gradient tensor; at the bench size the whole-gradient error is 2e-3 / 6e-3, test_full_size_16bit_modes_against_f32).
TOL_GOLDEN = {"f16": (5e-5, 5e-5, 0.08), "bf16": (4e-4, 2e-4, 0.2)}      # (rgb / opacity, depth, gradient relative L2)


def _golden_model(g, dev, precision):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import cfg_from_golden, nets_from_golden, make_sys_param
    from mc_nerf_amd.model import NeRF_Model
    cfg = cfg_from_golden(g)
    pc, pf = nets_from_golden(g, cfg)
    m = NeRF_Model(make_sys_param(cfg, device=str(dev), mode=0, precision=precision)).to(dev)
    m.nerf_coarse.load_state_dict(pc)
    m.nerf_fine.load_state_dict(pf)
    return m, cfg


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("name", ["g7_train_s64x2_small", "g7_train_s32x5_small_barf", "g7_train_s32x5_cap", "g7_train_s64x2_full"])
def test_render_train_16bit_vs_reference_golden(gpu_device, name, precision):
    from conftest import load_golden, t
    from mc_nerf_amd.model import MC_NeRF_Loss
    g = load_golden(name)
    dev = gpu_device
    m, cfg = _golden_model(g, dev, precision)
    d = t(g["rays_d"]).to(dev).requires_grad_(True)
    o = t(g["rays_o"]).to(dev).requires_grad_(True)
    kw = {"cap_perm": t(g["cap_perm"])} if "cap_perm" in g else {}       # the reference's captured cap permutation (:631)
    rgb_c, rgb_f = m.render_rays_train(d, o, 0, float(g["step_r"]), jitter=t(g["jitter"]).to(dev), eps_c=t(g["eps_c"]).to(dev),
                                       eps_sel=t(g["eps_sel"]).to(dev), eps_f=t(g["eps_f"]).to(dev), **kw)
    loss = MC_NeRF_Loss(dict(data_img_h=800, data_img_w=800)).get_rgb_loss([rgb_c, rgb_f, t(g["gt"]).to(dev)])
    loss.backward()
    e_c = float((rgb_c.detach().cpu() - t(g["rgb_c"])).abs().max())
    e_f = float((rgb_f.detach().cpu() - t(g["rgb_f"])).abs().max())

    def l2(a, b):
        b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
        return float((a.detach().cpu().double() - b).norm() / max(1e-30, float(b.norm())))
    worst, wname, n = 0.0, "", 0
    for tag, net in (("c", m.nerf_coarse), ("f", m.nerf_fine)):
        for k_, p in net.named_parameters():
            if f"g{tag}.{k_}" in g:
                e = l2(p.grad, g[f"g{tag}.{k_}"])
                n += 1
                if e > worst:
                    worst, wname = e, f"{tag}.{k_}"
    e_d, e_o = l2(d.grad, g["d_rays_d"]), l2(o.grad, g["d_rays_o"])
    k_ref = int(np.asarray(g["idx_f"]).shape[0]) if "idx_f" in g else -1
    print(f"[{precision} {name}] max|rgb_c-ref| {e_c:.1e}  max|rgb_f-ref| {e_f:.1e}  loss err {abs(float(loss) - float(g['loss'])):.1e}  "
          f"grad rel-L2: worst of {n} tensors {worst:.1e} ({wname}), d_rays_d {e_d:.1e}, d_rays_o {e_o:.1e}  "
          f"fine samples {int(m.last_selection[1].item())} (reference {k_ref})")
    tol_rgb, _, tol_g = TOL_GOLDEN[precision]
    assert e_c < tol_rgb and e_f < tol_rgb
    if n:
        assert worst < tol_g and e_d < tol_g and e_o < tol_g


@pytest.mark.parametrize("precision", ["f16", "bf16"])
@pytest.mark.parametrize("name", ["g8_test_s64x2_small", "g8_test_s64x2_full", "g8_test_s128x5_small"])
def test_render_test_16bit_vs_reference_golden(gpu_device, name, precision):
    from conftest import load_golden, t
    g = load_golden(name)
    dev = gpu_device
    m, cfg = _golden_model(g, dev, precision)
    rgb, depth, opacity = m.render_rays_test(t(g["rays_d"]).to(dev), t(g["rays_o"]).to(dev), m.nerf_coarse, m.nerf_fine,
                                             eps_c=t(g["eps_c"]).to(dev), eps_sel=t(g["eps_sel"]).to(dev), eps_f=t(g["eps_f"]).to(dev))
    e_rgb = float((rgb.cpu() - t(g["rgb"])).abs().max())
    e_dep = float((depth.cpu() - t(g["depth"])).abs().max())
    e_op = float((opacity.cpu() - t(g["opacity"])).abs().max())
    print(f"[{precision} {name}] max|rgb-ref| {e_rgb:.1e}  max|depth-ref| {e_dep:.1e}  max|opacity-ref| {e_op:.1e}")
    tol_rgb, tol_depth, _ = TOL_GOLDEN[precision]
    assert e_rgb < tol_rgb and e_op < tol_rgb and e_dep < tol_depth
