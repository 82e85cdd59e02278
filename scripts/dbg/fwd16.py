import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import mcnerf_oracle as O
from mc_nerf_amd import ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_mlp16_gpu as T
dev = torch.device("cuda:0")
for width in (32, 256):
    N, S = 37, 48
    nc, cfg, step_r, p, d, o, jitter, zg, xyz, dirs = T.fwd_case(width, False, N, S)
    net = T.net_of(nc)
    ref, hidden, sh = O.mlp_forward(p, nc, O.embed(xyz, step_r, cfg), dirs, return_hidden=True)
    flat = T.flat_params(nc, p, dev)
    packed = ops.pack_weights(net, flat, precision="f16")
    out = torch.full((N, S, 4), float("nan"), device=dev)
    ops.mlp_fwd(net, flat, packed, o.to(dev), d.to(dev), zg.to(dev), jitter.reshape(-1).to(dev).contiguous(), O.barf_weights(step_r, cfg).to(dev), out, precision="f16")
    torch.cuda.synchronize()
    e = (out.view(-1, 4).cpu() - ref).abs()
    print(width, "per-column max err", e.max(0).values.tolist(), "ref absmax", ref.abs().max(0).values.tolist())
    print(" worst rows", e.max(1).values.topk(5).indices.tolist())
    r = int(e[:, 0].argmax()); print(" row", r, "out", out.view(-1, 4)[r].tolist(), "ref", ref[r].tolist())
