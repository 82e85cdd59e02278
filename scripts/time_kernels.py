"""Times the fine-net forward / backward chain / weight-gradient calls alone at bench scale (dense rays x 128).
    python scripts/time_kernels.py [precision] [rays] [width] [which,...]     (MCNERF_LIB=... selects an ablation build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mc_nerf_amd import ops
from _nets import make_net
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25600
width = int(sys.argv[3]) if len(sys.argv) > 3 else 256
S = 128
dev = torch.device("cuda:0")
net, flat = make_net(width, dev)
if os.environ.get('WSCALE'):          # (power experiments: the weights scaled by a factor, 0 = all-zero operands)
    flat.mul_(float(os.environ['WSCALE']))
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
grads = torch.zeros_like(flat)
d_o, d_d = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
fns = {
    "fwd": lambda: ops.mlp_fwd(net, flat, packed, o, d, zg, None, bw, out, save=save, precision=prec),
    "fwd_nosave": lambda: ops.mlp_fwd(net, flat, packed, o, d, zg, None, bw, out, precision=prec),
    "bwd": lambda: ops.mlp_bwd(net, flat, packed, o, d, zg, None, bw, out, d_out, save, dy, dsh, d_o, d_d, precision=prec, gmax=gmax),
    "dw": lambda: ops.mlp_dw(net, save, dy, dsh, grads, N * S, precision=prec, gmax=gmax),
}
# optional shader-clock probe (scripts/dbg/libclockprobe.so): a one-wave kernel on a side stream samples s_memtime against the
# 100 MHz counter every 50 us while the timed kernel runs -> effective MHz under that kernel's load
probe = None
if os.environ.get("CLOCKPROBE"):
    import ctypes, subprocess
    _dbg = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg")
    if not os.path.exists(os.path.join(_dbg, "libclockprobe.so")):      # debug-only probe: built on demand, not by __graft_entry__.build()
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O2", "-shared", "-fPIC",
                        "-o", os.path.join(_dbg, "libclockprobe.so"), os.path.join(_dbg, "clockprobe.hip")], check=True)
    probe = ctypes.CDLL(os.path.join(_dbg, "libclockprobe.so"))
    probe.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_void_p]
    side = torch.cuda.Stream()
res = []
which = sys.argv[4].split(",") if len(sys.argv) > 4 else list(fns)
flop = 2.0 * sum(a * b for (a, b) in [sh for sh in net.shapes() if len(sh) == 2]) * N * S
for name, fn in fns.items():
    if name not in which:
        continue
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 4
    if probe is not None:
        ns = 200
        buf = torch.zeros(2 * ns, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        probe.clock_probe(buf.data_ptr(), ns, 5000, side.cuda_stream)      # 200 samples x 50 us = 10 ms
        reps = 8
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    clk = ""
    if probe is not None:
        b = buf.cpu().view(ns, 2).double()
        mhz = b[:, 0] / b[:, 1] * 100.0
        clk = f" clk {mhz[20:].mean():.0f} MHz (min {mhz[20:].min():.0f} max {mhz[20:].max():.0f})"
    res.append(f"{name} {ms:7.2f} ms ({flop / ms * 1e-9:6.1f} TF){clk}")
print(os.environ.get("MCNERF_LIB", "default"), prec, f"rows={N * S}", " | ".join(res), "finite", bool(torch.isfinite(out).all()))
