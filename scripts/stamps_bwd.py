"""Phase timings of the split-f16 backward chain from a -DMCN_STAMPS build:  MCNERF_LIB=.../libmcnerf_stamps.so python scripts/stamps_bwd.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import mcnerf_oracle as O
from mc_nerf_amd import ops, _lib
dev = torch.device("cuda:0")
nc = O.NetCfg(8, 256, (4,)); net = ops.Net(8, 256, 4)
p = O.init_params(nc, 7)
flat = ops.flatten_params(net, [p[k].to(dev) for k in net.names()], dev)
N, S = 25600, 128
g = torch.Generator(device=dev).manual_seed(0)
o = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1) * 3
d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(N, 3, device=dev, generator=g), dim=-1)
zg = torch.linspace(1, 8, S, device=dev); bw = torch.ones(10, device=dev)
packed = ops.pack_weights(net, flat, precision="f16x3")
out = torch.empty(N, S, 4, device=dev)
save = ops.alloc_save(net, N * S, dev)
ops.mlp_fwd(net, flat, packed, o, d, zg, None, bw, out, save=save, precision="f16x3")
d_out = torch.randn(N, S, 4, device=dev, generator=g) * 1e-4
gmax = d_out.abs().max().reshape(1).view(torch.int32)
dy, dsh = torch.empty_like(save.act), torch.empty_like(save.sh)
d_o, d_d = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
for _ in range(2):
    ops.mlp_bwd(net, flat, packed, o, d, zg, None, bw, out, d_out, save, dy, dsh, d_o, d_d, precision="f16x3", gmax=gmax)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 4 * 8))()
l = _lib.lib(); l.mcnerf_debug_stamps_bwd.restype = ctypes.c_int
assert l.mcnerf_debug_stamps_bwd(buf) == 0
T = np.frombuffer(buf, dtype=np.uint64).reshape(64, 4, 8).astype(np.int64)
ok = T[:, :, 5] > 0
names = ["per-sample prologue", "heads (3 GEMMs)", "trunk (8 layers)", "encoded-gradient to LDS", "encoding bwd + ray atomics"]
for i, n in enumerate(names):
    v = (T[:, :, i + 1] - T[:, :, i])[ok]
    print(f"  {n:28s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles")
print(f"  whole tile {(T[:, :, 5] - T[:, :, 0])[ok].mean():.0f}")
