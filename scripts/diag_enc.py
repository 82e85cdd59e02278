"""Diagnostic: where does the fused kernel's encoding differ from CPU torch? (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import mcnerf_oracle as O
from mc_nerf_amd import ops
dev = torch.device("cuda:0")
nc = O.NetCfg(4, 32, (2,)); net = ops.Net(4, 32, 2)
S, N = 48, 37
cfg = O.RenderCfg(samples=S, scale=2, coarse=nc, fine=nc)
g = torch.Generator().manual_seed(1)
o = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1) * 3
d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=-1)
jitter = torch.rand(N, 1, generator=g) * 7 / S
zg = torch.linspace(1, 8, S)
z = zg.unsqueeze(0) + jitter
xyz = (o.unsqueeze(1) + d.unsqueeze(1) * z.unsqueeze(2)).reshape(-1, 3)
p = O.init_params(nc, 1)
flat = ops.flatten_params(net, [p[k].to(dev) for k in net.names()], dev)
packed = ops.pack_weights(net, flat)
out = torch.empty(N, S, 4, device=dev)
save = ops.alloc_save(net, N * S, dev)
ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous(),
            torch.ones(10, device=dev), out, save=save)
enc = save.enc.view(-1, 64).cpu()
ref32 = O.embed(xyz, 1.0, cfg)
x64 = xyz.double()
fr = 2.0 ** torch.arange(10, dtype=torch.float64)
arg = x64.unsqueeze(-1) * fr
ref64 = torch.cat([x64, torch.cat([arg.sin(), arg.cos()], -1).reshape(-1, 60)], -1)
print("xyz equal bits:", torch.equal(enc[:, :3], xyz), "max|dx|", (enc[:, :3] - xyz).abs().max().item())
e_gpu = (enc[:, :63].double() - ref64).abs()
e_cpu = (ref32.double() - ref64).abs()
print("kernel vs f64:", e_gpu.max().item(), " torch-cpu-f32 vs f64:", e_cpu.max().item())
col = e_gpu.max(0).values
print("per-frequency max err kernel :", [f"{col[3+f].item():.1e}" for f in range(10)])
colc = e_cpu.max(0).values
print("per-frequency max err cpu f32:", [f"{colc[3+f].item():.1e}" for f in range(10)])
tg = torch.sin(xyz.to(dev)[:, :1] * 512.0).cpu().double()
print("torch-gpu sin vs f64:", (tg - arg[:, 0, 9:10].sin()).abs().max().item())
