"""Premise check for running the HBM-bound weight-gradient kernel beside the MFMA-bound backward chain on disjoint CU sets:
times dW / bwd alone under grid caps (MCNERF_EXP_DW_GRID / MCNERF_EXP_CHAIN_GRID, read by the launchers at every launch) and
both concurrently on two streams.  Needs the experiment build of the library:
    python -m mc_nerf_amd.build --tag=exp -DMCNERF_EXPERIMENTS
    MCNERF_LIB=$PWD/mc_nerf_amd/libmcnerf_exp.so python scripts/experiments/overlap_probe.py [rays] [width]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mc_nerf_amd import ops
from _nets import make_net
prec = "f16x3"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25600
width = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S = 128
dev = torch.device("cuda:0")
net, flat = make_net(width, dev)
g = torch.Generator(device=dev).manual_seed(0)
o = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1) * 3
d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(N, 3, device=dev, generator=g), dim=-1)
zg = torch.linspace(1, 8, S, device=dev)
bw = torch.ones(10, device=dev)
packed = ops.pack_weights(net, flat, precision=prec)
out = torch.empty(N, S, 4, device=dev)
save = ops.alloc_save(net, N * S, dev, precision=prec)
d_out = torch.randn(N, S, 4, device=dev, generator=g) * 1e-4
gmax = d_out.abs().max().reshape(1).view(torch.int32)
dy, dsh = ops.alloc_grad_ws(net, save, prec)
dy2, dsh2 = ops.alloc_grad_ws(net, save, prec)
grads = torch.zeros_like(flat)
d_o, d_d = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
ops.mlp_fwd(net, flat, packed, o, d, zg, None, bw, out, save=save, precision=prec)
bwd = lambda: ops.mlp_bwd(net, flat, packed, o, d, zg, None, bw, out, d_out, save, dy, dsh, d_o, d_d, precision=prec, gmax=gmax)
bwd2 = lambda: ops.mlp_bwd(net, flat, packed, o, d, zg, None, bw, out, d_out, save, dy2, dsh2, d_o, d_d, precision=prec, gmax=gmax)
dw = lambda: ops.mlp_dw(net, save, dy2, dsh2, grads, N * S, precision=prec, gmax=gmax)
bwd(); bwd2(); dw(); torch.cuda.synchronize()

def timed(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def setcap(dwg, chg):
    for k, v in (("MCNERF_EXP_DW_GRID", dwg), ("MCNERF_EXP_CHAIN_GRID", chg)):
        if v: os.environ[k] = str(v)
        else: os.environ.pop(k, None)

for cap in (0, 192, 128, 96, 64, 48, 32):
    setcap(cap, 0)
    print(f"dW alone grid {cap or 256}: {timed(dw):.2f} ms", flush=True)
for cap in (0, 224, 192, 160, 128):
    setcap(0, cap)
    print(f"bwd alone grid {cap or 256}: {timed(bwd):.2f} ms", flush=True)
side = torch.cuda.Stream()
for dwg, chg in ((64, 192), (48, 208), (96, 160), (32, 224), (128, 128)):
    setcap(dwg, chg)
    def both():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dw()
        bwd()
        torch.cuda.current_stream().wait_stream(side)
    print(f"dW grid {dwg} || bwd grid {chg}: {timed(both):.2f} ms   (alone: dW {0:.0f})", flush=True)
setcap(0, 0)
seq = lambda: (bwd(), dw())
print(f"sequential bwd + dW: {timed(seq):.2f} ms")
