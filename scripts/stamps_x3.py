"""Phase timings of the f16x3 forward chain from a -DMCNX3_STAMPS build (in-kernel s_memtime stamps of one pass per workgroup):
    python -m mc_nerf_amd.build --tag=stx3 -DMCNX3_STAMPS;  MCNERF_LIB=mc_nerf_amd/libmcnerf_stx3.so python scripts/stamps_x3.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mc_nerf_amd import ops, _lib
from _nets import make_net
prec = "f16x3"
dev = torch.device("cuda:0")
WIDTH = int(sys.argv[1]) if len(sys.argv) > 1 else 256       # (a -DMCNX3_STAMP_W=128 build stamps the 128-wide kernel)
net, flat = make_net(WIDTH, dev)
N, S = 25600, 128
g = torch.Generator(device=dev).manual_seed(0)
o = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1) * 3
d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(N, 3, device=dev, generator=g), dim=-1)
zg = torch.linspace(1, 8, S, device=dev); bw = torch.ones(10, device=dev)
packed = ops.pack_weights(net, flat, precision=prec)
out = torch.empty(N, S, 4, device=dev)
save = ops.alloc_save(net, N * S, dev, precision=prec)
for _ in range(3):
    ops.mlp_fwd(net, flat, packed, o, d, zg, None, bw, out, precision=prec)
for _ in range(3):
    ops.mlp_fwd(net, flat, packed, o, d, zg, None, bw, out, save=save, precision=prec)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 64 * 16))()
l = _lib.lib(); l.mcnerf_debug_stamps_x3_fwd.restype = ctypes.c_int
assert l.mcnerf_debug_stamps_x3_fwd(buf) == 0
T = np.frombuffer(buf, dtype=np.uint64).reshape(2, 64, 16).astype(np.int64)
names = ["prologue: inputs from LDS, index DMA (the encoding's arithmetic is scheduled into the next interval)", "encoding (30 sin / cos) + layer 0 (96 MFMAs)"] + \
        [f"trunk layer {l} ({'480' if l == 4 else '384'} MFMAs = {15360 if l == 4 else 12288} cycles of pipe)" for l in range(1, 8)] + \
        ["sigma head (384 MFMAs)", "next-pass index read + gather DMAs + SH head (384 MFMAs)", "sh.2 (48 MFMAs, no epilogue overlap)",
         "per-sample epilogue (SH colour, sigmoid) + output store"]
if WIDTH != 256:      # depth-4 net: stamps 0 (top), 1 (after the encoding), 2 (layer 0), 3..5 (layers 1..3), 10 (sigma head), 11 (SH head), 12 (sh.2), 13 (end)
    names = ["prologue", "encoding + layer 0", "layer 1", "layer 2 (skip)", "layer 3", "sigma head", "SH head (+ next-pass gathers)", "sh.2", "per-sample epilogue + output store"]
    idx = [0, 1, 2, 3, 4, 5, 10, 11, 12, 13]
else:
    idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13]
for v, tag in ((0, "no save (render)"), (1, "save (training)")):
    t = T[v]
    ok = t[:, 13] > t[:, 0]
    print(f"--- f16x3 forward, width {WIDTH}, {tag}: one pass = 128 rows ({'3687 MFMAs per wave = 117 984' if WIDTH == 256 else '600 MFMAs per wave = 19 200'} cycles of matrix pipe)")
    for i, n in enumerate(names):
        d_ = (t[:, idx[i + 1]] - t[:, idx[i]])[ok]
        print(f"  {n:64s} mean {d_.mean():9.0f}  p10 {np.percentile(d_, 10):9.0f}  p90 {np.percentile(d_, 90):9.0f}")
    print(f"  whole pass {(t[:, 13] - t[:, 0])[ok].mean():.0f} shader cycles ({ok.sum()} workgroups)")
