"""Phase timings of the 16-bit backward chain from a -DMCN16_STAMPS build:
    python -m mc_nerf_amd.build --tag=stamps -DMCN16_STAMPS;  MCNERF_LIB=.../libmcnerf_stamps.so python scripts/stamps16_bwd.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mc_nerf_amd import ops, _lib
from _nets import make_net
prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
dev = torch.device("cuda:0")
net, flat = make_net(256, dev)
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
d_out = torch.randn(N, S, 4, device=dev, generator=g) * 1e-4
gmax = d_out.abs().max().reshape(1).view(torch.int32)
dy, dsh = ops.alloc_grad_ws(net, save, prec)
d_o, d_d = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
for _ in range(3):
    ops.mlp_bwd(net, flat, packed, o, d, zg, None, bw, out, d_out, save, dy, dsh, d_o, d_d, precision=prec, gmax=gmax)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (64 * 16))()
l = _lib.lib(); l.mcnerf_debug_stamps16_bwd.restype = ctypes.c_int
assert l.mcnerf_debug_stamps16_bwd(buf) == 0
T = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
names = ["prologue (loads, sh/sigmoid bwd, dsh store)", "sigma.0 dY (outer product, store)", "sigma.0^T (partial)", "sh.2^T -> dY sh.0",
         "sh.0^T + partial -> dY_7"] + [f"trunk layer {l}^T -> dY_{l-1}" for l in range(7, 0, -1)] + ["layer 0^T (encoded cols)", "encoding bwd + ray atomics"]
ok = T[:, 14] > T[:, 0]
for i, n in enumerate(names):
    v = (T[:, i + 1] - T[:, i])[ok]
    print(f"  {n:46s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f}")
print(f"  whole pass {(T[:, 14] - T[:, 0])[ok].mean():.0f} shader cycles (s_memtime)   ({ok.sum()} workgroups)")

fbuf = (ctypes.c_ulonglong * (2 * 64 * 16))()
l.mcnerf_debug_stamps16_fwd.restype = ctypes.c_int
assert l.mcnerf_debug_stamps16_fwd(fbuf) == 0
F = np.frombuffer(fbuf, dtype=np.uint64).reshape(2, 64, 16).astype(np.int64)
fnames = ["setup (loads, encoding, enc store)", "layer 0"] + [f"trunk layer {l}" for l in range(1, 8)] + ["sigma.0 (+ sigma.2 dot)", "sh.0", "sh.2", "epilogue (sh save, SH colour, sigmoid, out)"]
for v_, tag in ((0, "forward, no save"), (1, "forward, saving")):
    Tf = F[v_]; okf = Tf[:, 13] > Tf[:, 0]
    print(tag)
    for i, n in enumerate(fnames):
        v = (Tf[:, i + 1] - Tf[:, i])[okf]
        print(f"  {n:46s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f}")
    print(f"  whole pass {(Tf[:, 13] - Tf[:, 0])[okf].mean():.0f} shader cycles   ({okf.sum()} workgroups)")
