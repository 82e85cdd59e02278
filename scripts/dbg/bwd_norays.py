import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scripts")
import torch
from mc_nerf_amd import ops
from _nets import make_net
prec=os.environ.get("PREC","f16x3h"); dev=torch.device("cuda:0")
for width,N in ((128,32768),(256,25600)):
    S=128 if width==256 else 64
    net, flat = make_net(width, dev)
    g = torch.Generator(device=dev).manual_seed(0)
    o = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1) * 3
    d = torch.nn.functional.normalize(-o + 0.5 * torch.randn(N, 3, device=dev, generator=g), dim=-1)
    zg = torch.linspace(1, 8, S, device=dev); bw = torch.ones(10, device=dev)
    packed = ops.pack_weights(net, flat, precision=prec)
    out = torch.empty(N, S, 4, device=dev)
    save = ops.alloc_save(net, N * S, dev, precision=prec)
    ops.mlp_fwd(net, flat, packed, o, d, zg, None, bw, out, save=save, precision=prec)
    d_out = torch.randn(N, S, 4, device=dev, generator=g) * 1e-4
    gmax = d_out.abs().max().reshape(1).view(torch.int32)
    dy, dsh = ops.alloc_grad_ws(net, save, prec)
    d_o, d_d = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
    for name,(a,b) in (("with rays",(d_o,d_d)),("no rays",(None,None))):
        fn=lambda: ops.mlp_bwd(net, flat, packed, o, d, zg, None, bw, out, d_out, save, dy, dsh, a, b, precision=prec, gmax=gmax)
        fn(); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        print(width, N*S, name, round(e0.elapsed_time(e1)/5,3),"ms")
