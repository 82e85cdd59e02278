import ctypes, os, sys
sys.path.insert(0, "/root/repo") if os.path.exists("/root/repo") else None
sys.path.insert(0, os.getcwd())
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
save = ops.alloc_save(net, N * S, dev) if sys.argv[1] == "save" else None
for _ in range(2):
    ops.mlp_fwd(net, flat, packed, o, d, zg, None, bw, out, save=save, precision="f16x3")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 8 * 8))()
l = _lib.lib(); l.mcnerf_debug_stamps.restype = ctypes.c_int
assert l.mcnerf_debug_stamps(buf) == 0
T = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8, 8).astype(np.int64)[:, :4]
ok = T[:, :, 5] > 0
seq = [5, 6, 7, 0, 1, 2, 3, 4]
names = ["setup", "encoding", "layer 0", "layer 1", "layer 2", "layer 3", "layer 4 (skip)"]
for a, b, n in zip(seq[:-1], seq[1:], names):
    v = (T[:, :, b] - T[:, :, a])[ok]
    print(f"{sys.argv[1]:7s} {n:16s} mean {v.mean():8.0f}  p10 {np.percentile(v,10):8.0f} p90 {np.percentile(v,90):8.0f}")
