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
    kw = dict(samples=c[0], scale=c[1], coarse=O.NetCfg(c[2], c[3], (c[4],)), fine=O.NetCfg(c[5], c[6], (c[7],)),
              barf_mode=bool(c[8]))
    if "barf" in g:
        kw.update(barf_start=float(g["barf"][0]), barf_end=float(g["barf"][1]))
    return O.RenderCfg(**kw)


def nets_from_golden(g, cfg):
    from oracle import mcnerf_oracle as O
    pc = O.init_params(cfg.coarse, int(g["seed_c"]))
    pf = O.init_params(cfg.fine, int(g["seed_f"]))
    s = float(g["sigma_shift"])
    if s:
        pc["sigma.2.bias"] = pc["sigma.2.bias"] + s
        pf["sigma.2.bias"] = pf["sigma.2.bias"] + s
    return pc, pf


@pytest.fixture(scope="session")
def gpu_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
