"""Per-tensor gradient errors of the 2048-ray full-size golden (tests/golden/g7_train_s64x2_full2048.npz) in every precision mode:
error relative to the tensor's max |reference| and relative to its RMS.   python scripts/dbg/grad_golden_table.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden, t
from test_model_gpu import build_model
from mc_nerf_amd.model import MC_NeRF_Loss
# DY_HI_ONLY=1 / X_HI_ONLY=1: what the f16x3 weight gradients would be with the lo plane of dY / of X left out of the weight-gradient
# kernel (2 MFMAs per product, 3/4 of its bytes): the plane is zeroed between the backward chain and the weight-gradient call
if os.environ.get("DY_HI_ONLY") or os.environ.get("X_HI_ONLY"):
    from mc_nerf_amd import ops as _ops
    _dw = _ops.mlp_dw
    def _dw_dropped(net, save, dy, dsh, grads, rows, count=None, precision="f32", gmax=None):
        if precision == "f16x3":
            ks = net.width // 16
            if os.environ.get("DY_HI_ONLY"):
                dy.view(torch.float16).view(net.depth + 2, -1, 2, ks, 64, 8)[:, :, 1].zero_()
                dsh.view(torch.float16).view(-1, 2, 2, 64, 8)[:, 1].zero_()
            if os.environ.get("X_HI_ONLY"):
                save.act.view(torch.float16).view(net.depth + 2, -1, 2, ks, 64, 8)[:, :, 1].zero_()
                save.enc.view(torch.float16).view(-1, 2, 4, 64, 8)[:, 1].zero_()
        return _dw(net, save, dy, dsh, grads, rows, count=count, precision=precision, gmax=gmax)
    _ops.mlp_dw = _dw_dropped
g = load_golden("g7_train_s64x2_full2048")
dev = torch.device("cuda:0")
res = {}
for precision in ("f32", "f16x3", "f16"):
    m, cfg, pc, pf = build_model(g, dev, precision=precision)
    d = t(g["rays_d"]).to(dev).requires_grad_(True); o = t(g["rays_o"]).to(dev).requires_grad_(True)
    rgb_c, rgb_f = m.render_rays_train(d, o, 0, float(g["step_r"]), jitter=t(g["jitter"]).to(dev), eps_c=t(g["eps_c"]).to(dev),
                                       eps_sel=t(g["eps_sel"]).to(dev), eps_f=t(g["eps_f"]).to(dev))
    MC_NeRF_Loss(dict(data_img_h=800, data_img_w=800)).get_rgb_loss([rgb_c, rgb_f, t(g["gt"]).to(dev)]).backward()
    out = {"d_rays_d": d.grad.cpu().numpy(), "d_rays_o": o.grad.cpu().numpy()}
    for tag, net in (("c", m.nerf_coarse), ("f", m.nerf_fine)):
        for k_, p in net.named_parameters():
            out[f"{tag}.{k_}"] = p.grad.cpu().numpy()
    res[precision] = out
print(os.environ.get("MCNERF_LIB", "default library"))
print(f"{'tensor':34s} {'max|ref|':>10s} {'rms ref':>10s} | err/max: f32   f16x3     f16 | f16x3 vs f32 (err/max) | reference's own reorder noise")
for k in res["f32"]:
    tag, name = (k.split(".", 1) + [""])[:2] if k[0] in "cf" and k[1] == "." else ("", k)
    full = res["f32"][k]
    if k in g: ref, sel = g[k], (lambda a: a)
    elif f"g{tag}.{name}" in g: ref, sel = g[f"g{tag}.{name}"], (lambda a: a)
    else: ref, sel = g[f"gsamp{tag}.{name}"], (lambda a: a.reshape(-1)[::97])
    mx, rms = float(np.abs(ref).max()), float(np.sqrt((ref.astype(np.float64) ** 2).mean()))
    errs = [float(np.abs(sel(res[p][k]).astype(np.float64) - ref).max()) / mx for p in ("f32", "f16x3", "f16")]
    x = float(np.abs(res["f16x3"][k].astype(np.float64) - res["f32"][k]).max()) / mx
    print(f"{k:34s} {mx:10.3e} {rms:10.3e} | {errs[0]:9.1e} {errs[1]:9.1e} {errs[2]:9.1e} | {x:9.1e} | {float(g['noise_max.' + k]):9.1e}")
