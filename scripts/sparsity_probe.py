"""How sparse are the operands the weight-gradient kernel streams?  ("store/stream only live 8-channel groups
... or zero-skip whole 32x16 fragments whose mask word is 0 and measure the hit rate first".)
Runs the f16 fine-net forward + backward on bench-shaped rows (random-init weights, as the bench) and reports, for the saved X
(post-ReLU activations) and dY (masked pre-activation gradients): the fraction of exact zeros, of all-zero 8-channel groups (one
lane's 16-byte piece of a fragment) and of all-zero 32 x 16 fragments (1 KiB).  Usage (GPU box): python scripts/sparsity_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mc_nerf_amd import ops
from _nets import make_net
dev = torch.device("cuda:0")
N, S, W = 2048, 128, 256
net, flat = make_net(W, dev)
g = torch.Generator(device=dev).manual_seed(0)
o = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1) * 3
d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(N, 3, device=dev, generator=g), dim=-1)
zg, bw = torch.linspace(1, 8, S, device=dev), torch.ones(10, device=dev)
prec = "f16"
packed = ops.pack_weights(net, flat, precision=prec)
out = torch.empty(N, S, 4, device=dev)
save = ops.alloc_save(net, N * S, dev, precision=prec)
ops.mlp_fwd(net, flat, packed, o, d, zg, None, bw, out, save=save, precision=prec)
d_out = torch.randn(N, S, 4, device=dev, generator=g) * 1e-4
gmax = d_out.abs().max().reshape(1).view(torch.int32)
dy, dsh = ops.alloc_grad_ws(net, save, prec)
ops.mlp_bwd(net, flat, packed, o, d, zg, None, bw, out, d_out, save, dy, dsh, torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev), precision=prec, gmax=gmax)
torch.cuda.synchronize()
for name, buf in (("X  (saved activations)", save.act), ("dY (pre-activation gradients)", dy)):
    x = buf.view(torch.int16).view(net.depth + 2, -1, W // 16, 64, 8)       # [slot][tile][k-step][lane][8 halves]
    x = x[:, : N * S // 32]
    z = (x == 0) | (x == -32768)                                            # +0 / -0
    print(f"{name}: zeros {float(z.float().mean()):.3f}, all-zero 8-channel lane pieces {float(z.all(-1).float().mean()):.5f}, "
          f"all-zero 32x16 fragments {float(z.all(-1).all(-1).float().mean()):.6f}  (per slot zeros: "
          + " ".join(f"{float(z[s_].float().mean()):.2f}" for s_ in range(net.depth + 2)) + ")")
