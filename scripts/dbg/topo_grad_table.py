"""Per-tensor gradient errors of the general-topology goldens in every precision mode (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from conftest import load_golden, t
from test_model_gpu import build_model, err
from mc_nerf_amd.model import MC_NeRF_Loss
dev = torch.device("cuda:0")
names = sys.argv[1:] or ["g7_train_s32x2_deg0", "g7_train_s32x2_deg1", "g7_train_s32x2_freq6_barf", "g7_train_s64x2_small"]
for name in names:
    g = load_golden(name)
    for precision in ("f32", "f16x3", "f16", "bf16"):
        m, cfg, pc, pf = build_model(g, dev, precision=precision)
        d = t(g["rays_d"]).to(dev).requires_grad_(True); o = t(g["rays_o"]).to(dev).requires_grad_(True)
        rgb_c, rgb_f = m.render_rays_train(d, o, 0, float(g["step_r"]), jitter=t(g["jitter"]).to(dev), eps_c=t(g["eps_c"]).to(dev),
                                           eps_sel=t(g["eps_sel"]).to(dev), eps_f=t(g["eps_f"]).to(dev))
        MC_NeRF_Loss(dict(data_img_h=800, data_img_w=800)).get_rgb_loss([rgb_c, rgb_f, t(g["gt"]).to(dev)]).backward()
        rows = []
        for tag, net in (("c", m.nerf_coarse), ("f", m.nerf_fine)):
            for k_, p in net.named_parameters():
                ref = g[f"g{tag}.{k_}"]
                rows.append((err(p.grad, ref) / max(float(np.abs(ref).max()), 1e-30), f"{tag}.{k_}", float(np.abs(ref).max())))
        rows.sort(reverse=True)
        print(f"{name} {precision}: rgb {err(rgb_c, g['rgb_c']):.1e}/{err(rgb_f, g['rgb_f']):.1e} rays {err(d.grad, g['d_rays_d']) / float(np.abs(g['d_rays_d']).max()):.1e}  worst: "
              + ", ".join(f"{k} {e:.1e} (|g| {s:.1e})" for e, k, s in rows[:4]))
