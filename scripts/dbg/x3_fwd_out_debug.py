"""Debug aid: per-column / per-row error pattern of the split-f16 forward's output against the oracle (dense grid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import mcnerf_oracle as O
from mc_nerf_amd import ops
width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NETS = {32: O.NetCfg(4, 32, (2,)), 64: O.NetCfg(8, 64, (4,)), 128: O.NetCfg(4, 128, (2,)), 256: O.NetCfg(8, 256, (4,))}
dev = torch.device("cuda:0")
nc = NETS[width]
net = ops.Net(nc.depth, nc.width, nc.skips[0])
S, N = 48, 37
cfg = O.RenderCfg(samples=S, scale=2, coarse=nc, fine=nc, barf_mode=False)
p = O.init_params(nc, 100 + width)
g = torch.Generator().manual_seed(5 + width)
o = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1) * 3.0
d = torch.nn.functional.normalize((torch.rand(N, 3, generator=g) - 0.5) * 1.5 - o, dim=-1)
zg = torch.linspace(cfg.near, cfg.far, S)
xyz = (o.unsqueeze(1) + d.unsqueeze(1) * zg.view(1, S, 1)).reshape(-1, 3)
dirs = d.unsqueeze(1).expand(-1, S, -1).reshape(-1, 3)
ref, hidden, sh = O.mlp_forward(p, nc, O.embed(xyz, 1.0, cfg), dirs, return_hidden=True)
flat = ops.flatten_params(net, [p[k].to(dev) for k in net.names()], dev)
for prec in ("f16x3",):
    packed = ops.pack_weights(net, flat, precision=prec)
    out = torch.full((N, S, 4), float("nan"), device=dev)
    ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), None, torch.ones(10, device=dev), out, precision=prec)
    torch.cuda.synchronize()
    e = (out.view(-1, 4).cpu() - ref).abs()
    print(prec, "W", width, "per-column max err", e.max(0).values.tolist())
    bad = (e.max(1).values > 1e-4).nonzero().flatten()
    print("bad rows", bad.numel(), "of", e.shape[0], "; m = row % 32 histogram:", torch.bincount(bad % 32, minlength=32).tolist())
    for r in bad[:6].tolist():
        print(r, "m", r % 32, "got", out.view(-1, 4)[r].tolist(), "ref", ref[r].tolist())
