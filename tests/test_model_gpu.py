"""End-to-end GPU parity of the reference-shaped API (NeRF_Model) against the golden vectors that were
captured from the actual reference, on identical rays + jitter + noise.

The bar (BASELINE.json north_star): rendered rgb / depth within 1e-4 abs fp32.
"""
import numpy as np
import pytest
import torch

from conftest import cfg_from_golden, load_golden, make_sys_param, nets_from_golden, t
from oracle import mcnerf_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def build_model(g, dev, mode=0):
    from mc_nerf_amd.model import NeRF_Model
    cfg = cfg_from_golden(g)
    sp = make_sys_param(cfg, device=str(dev), mode=0)
    m = NeRF_Model(sp).to(dev)
    pc, pf = nets_from_golden(g, cfg)
    m.nerf_coarse.load_state_dict(pc)
    m.nerf_fine.load_state_dict(pf)
    m.emmbedding_xyz.barf_mode = cfg.barf_mode
    return m, cfg, pc, pf


def err(a, b):
    return float(np.abs(a.detach().cpu().numpy().astype(np.float64) - np.asarray(b, np.float64)).max())


TRAIN = ["g7_train_s64x2_small", "g7_train_s32x5_small_barf", "g7_train_s32x5_cap", "g7_train_s64x2_full"]


@pytest.mark.parametrize("name", TRAIN)
def test_render_rays_train_matches_reference(gpu_device, name):
    g = load_golden(name)
    dev = gpu_device
    m, cfg, pc, pf = build_model(g, dev)
    d = t(g["rays_d"]).to(dev).requires_grad_(True)
    o = t(g["rays_o"]).to(dev).requires_grad_(True)
    kw = dict(jitter=t(g["jitter"]).to(dev), eps_c=t(g["eps_c"]).to(dev), eps_sel=t(g["eps_sel"]).to(dev),
              eps_f=t(g["eps_f"]).to(dev))
    if "cap_perm" in g:
        kw["cap_perm"] = t(g["cap_perm"])
    rgb_c, rgb_f = m.render_rays_train(d, o, 0, float(g["step_r"]), **kw)
    # selection must agree with the oracle's (a flipped sample would swap an MLP output for the default)
    with torch.no_grad():
        r = O.render_rays_train(pc, pf, cfg, t(g["rays_d"]), t(g["rays_o"]), float(g["step_r"]), t(g["jitter"]),
                                t(g["eps_c"]), t(g["eps_sel"]), t(g["eps_f"]),
                                cap_perm=t(g["cap_perm"]) if "cap_perm" in g else None)
    idx, count = m.last_selection
    k = int(count.item())
    assert k == r["idx_f"].shape[0]
    assert torch.equal(idx[:k].cpu().long(), r["idx_f"])
    assert err(rgb_c, g["rgb_c"]) < TOL
    assert err(rgb_f, g["rgb_f"]) < TOL
    from mc_nerf_amd.model import MC_NeRF_Loss
    loss = MC_NeRF_Loss(dict(data_img_h=800, data_img_w=800)).get_rgb_loss([rgb_c, rgb_f, t(g["gt"]).to(dev)])
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    gs = max(1.0, float(np.abs(g["d_rays_d"]).max()))
    assert err(d.grad, g["d_rays_d"]) < 1e-4 * gs
    assert err(o.grad, g["d_rays_o"]) < 1e-4 * max(1.0, float(np.abs(g["d_rays_o"]).max()))
    checked = 0
    for tag, net in (("c", m.nerf_coarse), ("f", m.nerf_fine)):
        for k_, p in net.named_parameters():
            key = f"g{tag}.{k_}"
            if key in g:
                assert p.grad is not None, key
                assert err(p.grad, g[key]) < 1e-4 * max(1.0, float(np.abs(g[key]).max())), key
                checked += 1
            elif f"gsamp{tag}.{k_}" in g:
                ref = g[f"gsamp{tag}.{k_}"]
                assert err(p.grad.reshape(-1)[::97], ref) < 1e-4 * max(1.0, float(np.abs(ref).max())), key
                assert abs(float(p.grad.norm()) - float(g[f"gnorm{tag}.{k_}"])) < 1e-3 * max(1.0, float(g[f"gnorm{tag}.{k_}"]))
                checked += 1
    assert checked == len(list(m.nerf_coarse.parameters())) + len(list(m.nerf_fine.parameters()))


def test_render_coarse_only(gpu_device):
    g = load_golden("g7_train_s32_coarse_only")
    m, cfg, pc, pf = build_model(g, gpu_device)
    dev = gpu_device
    rgb_c, none, depth_c = m.render_rays_train(t(g["rays_d"]).to(dev), t(g["rays_o"]).to(dev), 0, float(g["step_r"]),
                                               only_coarse=True, jitter=t(g["jitter"]).to(dev), eps_c=t(g["eps_c"]).to(dev))
    assert none is None
    assert err(rgb_c, g["rgb_c"]) < TOL
    assert err(depth_c, g["depth_c"]) < TOL


@pytest.mark.parametrize("name", ["g8_test_s64x2_small", "g8_test_s64x2_full", "g8_test_s128x5_small"])
def test_render_rays_test_matches_reference(gpu_device, name):
    g = load_golden(name)
    dev = gpu_device
    m, cfg, pc, pf = build_model(g, dev)
    rgb, depth, opacity = m.render_rays_test(t(g["rays_d"]).to(dev), t(g["rays_o"]).to(dev), m.nerf_coarse, m.nerf_fine,
                                             eps_c=t(g["eps_c"]).to(dev), eps_sel=t(g["eps_sel"]).to(dev),
                                             eps_f=t(g["eps_f"]).to(dev))
    assert rgb.shape == (g["rgb"].shape[0], 3) and depth.shape == (g["rgb"].shape[0], 1)
    assert err(rgb, g["rgb"]) < TOL
    assert err(depth, g["depth"]) < TOL
    assert err(opacity, g["opacity"]) < TOL


def test_default_draws_and_reference_style_call(gpu_device):
    """forward(rays_d, rays_o, epoch, step_r) with on-device RNG: seeded runs repeat, different seeds differ
    (the reference's noise is part of its semantics)."""
    g = load_golden("g7_train_s64x2_small")
    m, cfg, pc, pf = build_model(g, gpu_device)
    d, o = t(g["rays_d"]).to(gpu_device), t(g["rays_o"]).to(gpu_device)
    torch.manual_seed(5)
    a_c, a_f = m(d, o, 0, 1.0)
    torch.manual_seed(5)
    b_c, b_f = m(d, o, 0, 1.0)
    torch.manual_seed(6)
    c_c, c_f = m(d, o, 0, 1.0)
    assert torch.equal(a_c, b_c) and torch.equal(a_f, b_f)
    assert not torch.equal(a_f, c_f)
    assert a_c.shape == (d.shape[0], 3) and bool(torch.isfinite(a_f).all())


def test_params_survive_optimizer_and_state_dict(gpu_device):
    """Parameters are views of the flat buffer: an optimiser step and load_state_dict must both be seen by the
    kernels (no stale packed weights)."""
    from mc_nerf_amd.model import RAdam
    g = load_golden("g7_train_s64x2_small")
    m, cfg, pc, pf = build_model(g, gpu_device)
    dev = gpu_device
    d, o = t(g["rays_d"]).to(dev), t(g["rays_o"]).to(dev)
    kw = dict(jitter=t(g["jitter"]).to(dev), eps_c=t(g["eps_c"]).to(dev), eps_sel=t(g["eps_sel"]).to(dev),
              eps_f=t(g["eps_f"]).to(dev))
    opt = RAdam(m.parameters(), lr=1e-2)
    rgb_c, rgb_f = m.render_rays_train(d, o, 0, 1.0, **kw)
    ((rgb_c - 0.3) ** 2).mean().add(((rgb_f - 0.3) ** 2).mean()).backward()
    before = rgb_c.detach().clone()
    opt.step()
    rgb_c2, _ = m.render_rays_train(d, o, 0, 1.0, **kw)
    assert not torch.equal(before, rgb_c2.detach())
    m.nerf_coarse.load_state_dict(pc)
    m.nerf_fine.load_state_dict(pf)
    rgb_c3, _ = m.render_rays_train(d, o, 0, 1.0, **kw)
    assert err(rgb_c3, g["rgb_c"]) < TOL
