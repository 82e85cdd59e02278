import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def t(a):
    """numpy -> torch (fp32 stays fp32, ints stay int64)."""
    return torch.from_numpy(np.ascontiguousarray(a))


def cfg_from_golden(g):
    """Rebuild the oracle RenderCfg from a fixture's packed ``cfg`` / ``barf`` arrays."""
    from oracle import mcnerf_oracle as O
    c = [int(v) for v in g["cfg"]]
    sk = lambda key, first: tuple(int(v) for v in g[key]) if key in g else (first,)      # (fixtures of multi-skip nets carry the whole lists)
    kw = dict(samples=c[0], scale=c[1], coarse=O.NetCfg(c[2], c[3], sk("skips_c", c[4])), fine=O.NetCfg(c[5], c[6], sk("skips_f", c[7])),
              barf_mode=bool(c[8]))
    if "barf" in g:
        kw.update(barf_start=float(g["barf"][0]), barf_end=float(g["barf"][1]))
    if "deg" in g:
        kw.update(deg=int(g["deg"]))
    if "n_freqs" in g:
        kw.update(n_freqs=int(g["n_freqs"]))
    return O.RenderCfg(**kw)


def nets_from_golden(g, cfg):
    from oracle import mcnerf_oracle as O
    n_sh, in_ch = 3 * (cfg.deg + 1) ** 2, 3 + 6 * cfg.n_freqs
    pc = O.init_params(cfg.coarse, int(g["seed_c"]), in_ch=in_ch, n_sh=n_sh)
    pf = O.init_params(cfg.fine, int(g["seed_f"]), in_ch=in_ch, n_sh=n_sh)
    s = float(g["sigma_shift"])
    if s:
        pc["sigma.2.bias"] = pc["sigma.2.bias"] + s
        pf["sigma.2.bias"] = pf["sigma.2.bias"] + s
    return pc, pf


@pytest.fixture(autouse=True, scope="module")
def _give_back_gpu_cache_between_modules():
    """A test module's big workspaces (full-size passes: tens of GB) go back to the driver when the module is done, so that
    neither a later module nor a child process (tests/test_z_multirank_gpu.py) finds the GPU full of this process's cache."""
    yield
    if torch.cuda.is_available():
        import gc
        gc.collect()
        torch.cuda.empty_cache()


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def make_sys_param(cfg, device="cpu", mode=0, batch=128, **extra):
    """The flat `sys_param` dict the reference builds from config.yaml (config/config_read.py:42-74),
    restricted to the keys the hot path reads."""
    p = dict(mode=mode, device_type=device, near=cfg.near, far=cfg.far, samples=cfg.samples, scale=cfg.scale,
             MLP_deg=cfg.deg, white_back=cfg.white_back, root_weight="/tmp/mcnerf_w", demo_render_pth="/tmp/mcnerf_r",
             batch=batch, boader_min=-3.5, boader_max=3.5, grid_nerf=384, sigma_init=30.0,
             sigma_default=cfg.sigma_default, warmup_epoch=100, sample_weight_thresh=cfg.weight_thresh,
             res_h=800, res_w=800, data_name="lego", emb_freqs_xyz=cfg.n_freqs, barf_mask=cfg.barf_mode,
             barf_start=cfg.barf_start, barf_end=cfg.barf_end, coarse_MLP_depth=cfg.coarse.depth,
             coarse_MLP_width=cfg.coarse.width, coarse_MLP_skip=list(cfg.coarse.skips),
             fine_MLP_depth=cfg.fine.depth, fine_MLP_width=cfg.fine.width, fine_MLP_skip=list(cfg.fine.skips),
             distributed=False)
    p.update(extra)
    return p
