"""End-to-end render_rays_train in a 16-bit mode against the CPU oracle (same draws): measured errors."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mc_nerf_amd.model import NeRF_Model, MC_NeRF_Loss
from mc_nerf_amd import synthetic as S
from oracle import mcnerf_oracle as O
dev = torch.device("cuda:0")
cfg = O.RenderCfg(samples=64, scale=2)
pc, pf = O.init_params(cfg.coarse, 1), O.init_params(cfg.fine, 2)
g = torch.Generator().manual_seed(0)
n = 256
o = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1) * 3.0
d = torch.nn.functional.normalize(-o + 0.4 * torch.randn(n, 3, generator=g), dim=-1)
jit = torch.rand(n, 1, generator=g) * 7.0 / 64
e = [torch.randn(n, s, generator=g) for s in (64, 64, 128)]
gt = torch.rand(n, 3, generator=g)
for p in list(pc.values()) + list(pf.values()):
    p.requires_grad_(True)
r = O.render_rays_train(pc, pf, cfg, d, o, 1.0, jit, e[0], e[1], e[2])
O.rgb_loss(r["rgb_c"], r["rgb_f"], gt).backward()
for precision in sys.argv[1:] or ["f16", "bf16"]:
    sp = S.make_sys_param(dev, samples=64, scale=2, batch=256, H=64, W=64, precision=precision)
    m = NeRF_Model(sp).to(dev)
    m.nerf_coarse.load_state_dict({k: v.detach() for k, v in pc.items()})
    m.nerf_fine.load_state_dict({k: v.detach() for k, v in pf.items()})
    rgb_c, rgb_f = m.render_rays_train(d.to(dev), o.to(dev), 0, 1.0, jitter=jit.to(dev), eps_c=e[0].to(dev), eps_sel=e[1].to(dev), eps_f=e[2].to(dev))
    loss = MC_NeRF_Loss(sp).get_rgb_loss([rgb_c, rgb_f, gt.to(dev)])
    loss.backward()
    ec = float((rgb_c.detach().cpu() - r["rgb_c"].detach()).abs().max())
    ef = float((rgb_f.detach().cpu() - r["rgb_f"].detach()).abs().max())
    def l2(a, b): return float((a.cpu().double() - b.double()).norm() / b.double().norm())
    gc = max(l2(p.grad, pc[k].grad) for k, p in m.nerf_coarse.state_dict(keep_vars=True).items())
    gf = max(l2(p.grad, pf[k].grad) for k, p in m.nerf_fine.state_dict(keep_vars=True).items())
    print(f"[{precision}] max|rgb_c - oracle| {ec:.2e}  max|rgb_f - oracle| {ef:.2e}  worst rel-L2 grad coarse {gc:.2e} fine {gf:.2e}  loss {float(loss):.6f}")
