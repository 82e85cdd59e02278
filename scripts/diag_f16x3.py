"""Diagnostic: accuracy and speed of the split-f16 forward vs exact fp32 and the CPU oracle (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mcnerf_oracle as O
from mc_nerf_amd import ops
dev = torch.device("cuda:0")
for width, nc in ((32, O.NetCfg(4, 32, (2,))), (64, O.NetCfg(8, 64, (4,))), (128, O.NetCfg(4, 128, (2,))), (256, O.NetCfg(8, 256, (4,)))):
    net = ops.Net(nc.depth, nc.width, nc.skips[0])
    S, N = 48, 37
    cfg = O.RenderCfg(samples=S, scale=2, coarse=nc, fine=nc)
    p = O.init_params(nc, 100 + width)
    g = torch.Generator().manual_seed(1)
    o = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1) * 3
    d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(N, 3, generator=g), dim=-1)
    jit = torch.rand(N, 1, generator=g) * 7 / S
    zg = torch.linspace(1, 8, S)
    xyz = (o.unsqueeze(1) + d.unsqueeze(1) * (zg.unsqueeze(0) + jit).unsqueeze(2)).reshape(-1, 3)
    dirs = d.unsqueeze(1).expand(-1, S, -1).reshape(-1, 3)
    ref, hidden, sh = O.mlp_forward(p, nc, O.embed(xyz, 1.0, cfg), dirs, return_hidden=True)
    flat = ops.flatten_params(net, [p[k].to(dev) for k in net.names()], dev)
    res = {}
    for prec in ops.PRECISIONS:
        packed = ops.pack_weights(net, flat, precision=prec)
        out = torch.full((N, S, 4), float("nan"), device=dev)
        save = ops.alloc_save(net, N * S, dev)
        ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jit.reshape(-1).to(dev).contiguous(),
                    torch.ones(10, device=dev), out, save=save, precision=prec)
        torch.cuda.synchronize()
        dec = (lambda t_: t_) if prec == 'f32' else ops.decode_split_words
        act = dec(save.act).view(nc.depth + 2, save.capacity, width).cpu()
        errs = [float((act[l] - h).abs().max()) for l, h in enumerate(hidden)]
        res[prec] = (float((out.view(-1, 4).cpu() - ref).abs().max()), max(errs), float((dec(save.enc).view(-1, 64)[:, :63].cpu() - O.embed(xyz, 1.0, cfg)).abs().max()))
    print(f"W={width}: out/hidden/enc max err  f32 {res['f32'][0]:.2e}/{res['f32'][1]:.2e}/{res['f32'][2]:.1e}   f16x3 {res['f16x3'][0]:.2e}/{res['f16x3'][1]:.2e}/{res['f16x3'][2]:.1e}")
# speed at scale (fine net, dense mode, save)
nc = O.NetCfg(8, 256, (4,)); net = ops.Net(8, 256, 4)
p = O.init_params(nc, 7)
flat = ops.flatten_params(net, [p[k].to(dev) for k in net.names()], dev)
N, S = 16384, 128
o = torch.randn(N, 3, device=dev); d = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
zg = torch.linspace(1, 8, S, device=dev)
out = torch.empty(N, S, 4, device=dev)
save = ops.alloc_save(net, N * S, dev)
for prec in ops.PRECISIONS:
    packed = ops.pack_weights(net, flat, precision=prec)
    for use_save in (None, save):
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.mlp_fwd(net, flat, packed, o, d, zg, None, torch.ones(10, device=dev), out, save=use_save, precision=prec)
            b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        print(f"{prec:6s} save={use_save is not None}: {ms:7.2f} ms for {N*S/1e6:.2f} M samples -> {N*S*1.258e6/ms/1e9:.1f} algorithmic TFLOP/s")
