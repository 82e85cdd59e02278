"""Does the weight-gradient kernel's box-to-box spread (11.3 ... 12.7 ms per fine-net call in f16x3) come from WHERE its two streams
lie?  X planes (saved activations) and dY planes carved out of one arena at chosen relative offsets; the kernel timed at each.
    python scripts/experiments/dw_address_probe.py [precision] [rays]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mc_nerf_amd import ops
from _nets import make_net
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 25000
S = 128
dev = torch.device("cuda:0")
net, flat = make_net(256, dev)
ref = ops.alloc_save(net, N * S, dev, precision=prec)
na = ref.act.numel()
dy0, dsh = ops.alloc_grad_ws(net, ref, prec)
del dy0
MiB = 1 << 20
arena = torch.empty(2 * na + 64 * MiB, dtype=torch.uint8, device=dev)
if os.environ.get("PROBE_DATA", "zeros") == "random":          # operand data: zeros, or random f16 planes (finite, |x| ~ 1)
    arena[: arena.numel() // 2 * 2].view(torch.float16).normal_()
else:
    arena.zero_()
base = (-arena.data_ptr()) % (2 * MiB)                       # align the carve-outs to 2 MiB
print(f"{prec}: {N * S} rows, act / dY planes {na / 2**30:.2f} GiB each, slot stride {na // (net.depth + 2)} B (mod 2 MiB: {(na // (net.depth + 2)) % (2 * MiB)})")
grads = torch.zeros_like(flat)
gmax = torch.tensor([1e-4], device=dev).view(torch.int32)
na_r = (na + 2 * MiB - 1) // (2 * MiB) * (2 * MiB)
for rep in range(2):
    for delta in ((0, 65536, MiB + 4096) if os.environ.get("PROBE_DATA") else (0, 256, 4096, 65536, 256 * 1024, MiB, MiB + 4096, 3 * MiB // 2, 2 * MiB + 128 * 1024, 5 * MiB + 12288)):
        act = arena[base: base + na]
        dy = arena[base + na_r + delta: base + na_r + delta + na]
        save = ops.MlpSave(ref.capacity, act, ref.enc, ref.sh, ref.mask)
        fn = lambda: ops.mlp_dw(net, save, dy, dsh, grads, N * S, precision=prec, gmax=gmax)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"  rep {rep} dY base - X base = {na_r + delta:>12d} (+{delta:>8d}): {e0.elapsed_time(e1) / 4:.3f} ms", flush=True)
