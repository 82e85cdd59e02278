"""Reads the in-kernel cycle stamps of a -DMCN_STAMPS build (trunk layer 3 of 64 steady-state workgroups of the
split-f16 forward):  MCNERF_LIB=.../libmcnerf_stamps.so python scripts/stamps.py [save|nosave]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import mcnerf_oracle as O
from mc_nerf_amd import ops, _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "save"
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
save = ops.alloc_save(net, N * S, dev) if mode == "save" else None
for _ in range(2):
    ops.mlp_fwd(net, flat, packed, o, d, zg, None, bw, out, save=save, precision="f16x3")
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 8 * 8))()
l = _lib.lib()
l.mcnerf_debug_stamps.restype = ctypes.c_int
assert l.mcnerf_debug_stamps(buf) == 0
T = np.frombuffer(buf, dtype=np.uint64).reshape(64, 8, 8).astype(np.int64)
for title, t, names in (("MFMA waves", T[:, :4], ["GEMM (incl. copy issue)", "wait at barrier 1", "epilogue", "wait at barrier 2"]),
                        ("store waves", T[:, 4:], ["tile copy", "masks", "wait at barrier 1", "wait at barrier 2"])):
    ok = (t[:, :, 0] > 0)
    if not ok.any():
        continue
    dt = np.diff(t[:, :, :5], axis=2)
    print(mode, title, "waves with stamps:", int(ok.sum()))
    for i, n in enumerate(names):
        v = dt[:, :, i][ok]
        print(f"  {n:26s} mean {v.mean():8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f} cycles")
    print("  layer total mean", (t[:, :, 4] - t[:, :, 0])[ok].mean())
    if title == "MFMA waves" and (t[:, :, 7] > 0).any():
        o2 = ok & (t[:, :, 7] > 0)
        print(f"  tile: prologue+layers0-2 {(t[:, :, 0] - t[:, :, 5])[o2].mean():.0f}  layers 3..D-1 {(t[:, :, 6] - t[:, :, 0])[o2].mean():.0f}"
              f"  heads+final {(t[:, :, 7] - t[:, :, 6])[o2].mean():.0f}  whole tile {(t[:, :, 7] - t[:, :, 5])[o2].mean():.0f}")
