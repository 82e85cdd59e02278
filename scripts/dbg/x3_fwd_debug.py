import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import mcnerf_oracle as O
from mc_nerf_amd import ops
width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NETS = {32: O.NetCfg(4, 32, (2,)), 64: O.NetCfg(8, 64, (4,)), 128: O.NetCfg(4, 128, (2,)), 256: O.NetCfg(8, 256, (4,))}
nc = NETS[width]
net = ops.Net(nc.depth, nc.width, nc.skips[0])
dev = torch.device("cuda:0")
S, N = 48, 37
cfg = O.RenderCfg(samples=S, scale=2, coarse=nc, fine=nc)
p = O.init_params(nc, 100 + width)
g = torch.Generator().manual_seed(5)
o = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1) * 3
d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(N, 3, generator=g), dim=-1)
zg = torch.linspace(cfg.near, cfg.far, S)
xyz = (o.unsqueeze(1) + d.unsqueeze(1) * zg.view(1, S, 1)).reshape(-1, 3)
dirs = d.unsqueeze(1).expand(-1, S, -1).reshape(-1, 3)
ref, hidden, sh = O.mlp_forward(p, nc, O.embed(xyz, 1.0, cfg), dirs, return_hidden=True)
flat = ops.flatten_params(net, [p[k].to(dev) for k in net.names()], dev)
P = "f16x3"
packed = ops.pack_weights(net, flat, precision=P)
out = torch.full((N, S, 4), float("nan"), device=dev)
save = ops.alloc_save(net, N * S, dev, precision=P)
ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), None, torch.ones(10, device=dev), out, save=save, precision=P)
torch.cuda.synchronize()
act = (ops.decode_frags_16(save.act, nc.depth + 2, width, N * S, P) / ops.SPLIT_SCALE_X).cpu()
for l, h in enumerate(hidden):
    e = (act[l] - h).abs()
    print("layer", l, "maxerr", float(e.max()), "rows bad", int((e.max(1).values > 1e-4).sum()), "cols bad", int((e.max(0).values > 1e-4).sum()))
torch.set_printoptions(precision=4, linewidth=200)
print("ref  l0 row0", hidden[0][0, :16])
print("got  l0 row0", act[0][0, :16])
print("ref  l0 row1", hidden[0][1, :16])
print("got  l0 row1", act[0][1, :16])
# the packed forward stream: first fragment (layer 0, tile 0, k-step 0): lane (i, h) element j = W0[i][chan(0,h,j)] * 256
pf = ops.packed16_split(net, packed, P)[0].view(torch.float16)
fr = pf[:1024].float().view(2, 2, 32, 8)         # [part][h][i][j]
W0 = p["xyz_encoding_1.0.weight"]
chan = lambda s, h, j: 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)
want = torch.tensor([[[W0[i, chan(0, h, j)] * 256 for j in range(8)] for i in range(32)] for h in range(2)])
print("pack err", float(((fr[0] + fr[1]) - want).abs().max()))
print("out err", float((out.view(-1, 4).cpu() - ref).abs().max()))
