import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mcnerf_oracle as O
from mc_nerf_amd import ops
dev = torch.device("cuda:0")
nc = O.NetCfg(8, 256, (4,)); net = ops.Net(8, 256, 4)
p = O.init_params(nc, 7)
flat = ops.flatten_params(net, [p[k].to(dev) for k in net.names()], dev)
N, S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 128
g = torch.Generator(device=dev).manual_seed(0)
o = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1) * 3
d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(N, 3, device=dev, generator=g), dim=-1)
zg = torch.linspace(1, 8, S, device=dev)
gscale = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-6
res = {}
for prec in ("f32", "f16x3"):
    packed = ops.pack_weights(net, flat, precision=prec)
    out = torch.empty(N, S, 4, device=dev)
    save = ops.alloc_save(net, N * S, dev)
    ops.mlp_fwd(net, flat, packed, o, d, zg, None, torch.ones(10, device=dev), out, save=save, precision=prec)
    d_out = torch.randn(N, S, 4, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * gscale
    gmax = d_out.abs().max().reshape(1).view(torch.int32)
    dy = torch.empty_like(save.act); dsh = torch.empty_like(save.sh)
    grads = torch.zeros_like(flat)
    d_o = torch.zeros(N, 3, device=dev); d_d = torch.zeros(N, 3, device=dev)
    ops.mlp_bwd(net, flat, packed, o, d, zg, None, torch.ones(10, device=dev), out, d_out, save, dy, dsh, d_o, d_d, precision=prec, gmax=gmax)
    ops.mlp_dw(net, save, dy, dsh, grads, N * S, precision=prec, gmax=gmax)
    torch.cuda.synchronize()
    res[prec] = grads.clone()
    if prec == "f16x3":
        dyv = dy.view(10, N * S, 256)
        for l in range(10):
            w = dyv[l].view(torch.int32)
            hi = (w & 0xFFFF).to(torch.int16).view(torch.float16)
            lo = ((w >> 16) & 0xFFFF).to(torch.int16).view(torch.float16)
            print(f"dy slot {l}: hi finite {bool(torch.isfinite(hi).all())} max|hi| {float(hi.float().abs().max()):.3e}  lo finite {bool(torch.isfinite(lo).all())}")
        ev = ops.decode_split_words(save.enc)
        print("enc finite", bool(torch.isfinite(ev).all()), float(ev.abs().max()))
    print(prec, "grads finite:", bool(torch.isfinite(grads).all()), "d_o finite", bool(torch.isfinite(d_o).all()))
offs = ops.param_offsets(net)
for off, shp, name in zip(offs, net.shapes(), net.names()):
    n = 1
    for x in shp: n *= x
    a, b = res["f32"][off:off+n], res["f16x3"][off:off+n]
    print(f"{name:28s} max|f32| {float(a.abs().max()):.3e}  max diff {float((a-b).abs().max()):.3e}  finite {bool(torch.isfinite(b).all())}")
