import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from oracle import mcnerf_oracle as O
from mc_nerf_amd import ops
import test_mlp16_gpu as T
dev = torch.device("cuda:0")
precision = sys.argv[1] if len(sys.argv) > 1 else "f16"
for width in (32, 256):
    nc = T.NETS[width]; net = T.net_of(nc)
    S, N = 40, 29
    cfg = O.RenderCfg(samples=20, scale=2, coarse=nc, fine=nc, barf_mode=True, barf_start=0.2, barf_end=0.9)
    step_r = 0.6
    p = {k: v.requires_grad_(True) for k, v in O.init_params(nc, 200 + width).items()}
    d, o = T.make_rays(N, 9 + width); d.requires_grad_(True); o.requires_grad_(True)
    g = torch.Generator().manual_seed(2)
    jitter = torch.rand(N, 1, generator=g) * 0.2
    zg = torch.linspace(cfg.near, cfg.far, S)
    sel = torch.rand(N, S, generator=g) < 0.6
    idx = torch.nonzero(sel); K = idx.shape[0]
    z = zg.unsqueeze(0) + jitter
    r, j = idx[:, 0], idx[:, 1]
    xyz = (o[r] + d[r] * z[r, j].unsqueeze(-1)); xyz.retain_grad()
    enc_in = O.embed(xyz, step_r, cfg); enc_in.retain_grad()
    ref, hidden, sh = O.mlp_forward(p, nc, enc_in, d[r], return_hidden=True)
    for hh in hidden: hh.retain_grad()
    gout = torch.randn(K, 4, generator=g) * 1e-4
    (ref * gout).sum().backward()
    flat = T.flat_params(nc, {k: v.detach() for k, v in p.items()}, dev)
    packed = ops.pack_weights(net, flat, precision=precision)
    cap = K + 17
    idx_d = torch.zeros(cap, 2, dtype=torch.int32, device=dev); idx_d[:K] = idx.to(torch.int32).to(dev)
    count = torch.tensor([K], dtype=torch.int32, device=dev)
    out = torch.full((N, S, 4), 7.0, device=dev)
    save = ops.alloc_save(net, cap, dev, precision=precision)
    bw = O.barf_weights(step_r, cfg).to(dev)
    od, dd, zd, jd = o.detach().to(dev), d.detach().to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous()
    ops.mlp_fwd(net, flat, packed, od, dd, zd, jd, bw, out, idx=idx_d, count=count, max_rows=cap, save=save, precision=precision)
    d_out = torch.zeros(N, S, 4, device=dev); d_out[r.to(dev), j.to(dev)] = gout.to(dev)
    grads = torch.zeros_like(flat)
    dy, dsh = ops.alloc_grad_ws(net, save, precision)
    d_o = torch.zeros(N, 3, device=dev); d_d = torch.zeros(N, 3, device=dev)
    gmax = d_out.abs().max().reshape(1).view(torch.int32)
    ops.mlp_bwd(net, flat, packed, od, dd, zd, jd, bw, out, d_out, save, dy, dsh, d_o, d_d, idx=idx_d, count=count, max_rows=cap, precision=precision, gmax=gmax)
    ops.mlp_dw(net, save, dy, dsh, grads, cap, count=count, precision=precision, gmax=gmax)
    torch.cuda.synchronize()
    import math
    sg = 2.0 ** (4 - math.ceil(math.log2(float(d_out.abs().max()))))
    rel = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max() / max(1e-30, float(b.abs().max())))
    print(f"== W={width} {precision}: d_o {rel(d_o, o.grad):.2e} d_d {rel(d_d, d.grad):.2e}")
    dyv = ops.decode_frags_16(dy, nc.depth + 2, width, K, precision) / sg
    # reference pre-activation grads: dY_l = hidden[l].grad * (hidden[l] > 0)
    for l, hh in enumerate(hidden):
        refdy = hh.grad * (hh > 0)
        print(f"   dY slot {l}: {rel(dyv[l], refdy):.2e}  (max {float(refdy.abs().max()):.2e})")
    for off, shp, name in zip(ops.param_offsets(net), net.shapes(), net.names()):
        n = int(np.prod(shp))
        print(f"   {name:28s} {rel(grads[off:off + n].view(shp), p[name].grad):.2e}")
    dx_ref = xyz.grad
    print("   d_o rows err:", (d_o.cpu() - o.grad).abs().max(1).values[:8].tolist(), " ref:", o.grad.abs().max(1).values[:8].tolist())
    # ---- masks: decode mask_ws and compare with (hidden > 0)
    MW = max(1, width // 64)
    mk = save.mask.view(nc.depth + 2, -1, 64, MW).cpu()
    tiles = mk.shape[1]
    dec = torch.zeros(nc.depth + 2, tiles * 32, width, dtype=torch.bool)
    for iw in range(MW):
        for b in range(16):
            for half in range(2):
                t = 2 * iw + (b >> 3)
                if t >= width // 32: continue
                i = 7 - (b & 7)
                r_ = 2 * i + half
                bit = (mk[:, :, :, iw] >> (b + 16 * half)) & 1          # [slot, tile, lane]
                for h_ in range(2):
                    n = 32 * t + 8 * (r_ >> 2) + 4 * h_ + (r_ & 3)
                    dec[:, :, n] = bit[:, :, 32 * h_:32 * h_ + 32].reshape(nc.depth + 2, -1).bool() if False else dec[:, :, n]
                    dec.view(nc.depth + 2, tiles, 32, width)[:, :, :, n] = bit[:, :, 32 * h_:32 * h_ + 32].bool()
    for l, hh in enumerate(hidden):
        want = (hh > 0)
        got = dec[l, :K]
        bad = (want != got)
        print(f"   mask slot {l}: mismatches {int(bad.sum())} of {bad.numel()}  rows with mismatch: {bad.any(1).nonzero().flatten()[:10].tolist()}")
    l = nc.depth
    refdy = hidden[l].grad * (hidden[l] > 0)
    er = (dyv[l].cpu() - refdy).abs().max(1).values
    print("   dYs worst rows:", er.topk(6).indices.tolist(), er.topk(6).values.tolist())
    for name in ("sh.2.bias", "sigma.2.bias", "sigma.0.bias"):
        i = net.names().index(name); off = ops.param_offsets(net)[i]; n = int(np.prod(net.shapes()[i]))
        print("   ", name, "got", grads[off:off + min(n, 6)].cpu().tolist(), "ref", p[name].grad.reshape(-1)[:6].tolist())
    dshv = ops.decode_frags_16(dsh, 1, 32, K, precision)[0] / sg
    print("    colsum of decoded dsh[:, :6]", dshv[:, :6].sum(0).tolist(), " col27:", float(dshv[:, 27].sum()), "ref dsig sum", float(gout[:, 0].sum()))
