"""CPU checks of the C-ABI boundary: the library builds, loads, and exports exactly what
include/mcnerf.h declares (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "mcnerf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcnerf_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built_lib():
    from mc_nerf_amd import build
    return build.build(verbose=False)


def test_header_declares_expected_surface():
    fns = header_functions()
    for must in ["mcnerf_mlp_fwd", "mcnerf_mlp_bwd", "mcnerf_mlp_dw", "mcnerf_composite_fwd", "mcnerf_composite_bwd",
                 "mcnerf_select_fine", "mcnerf_raygen_fwd", "mcnerf_raygen_bwd", "mcnerf_pack_weights"]:
        assert must in fns


def test_library_exports_every_declared_symbol(built_lib):
    l = ctypes.CDLL(built_lib)
    for fn in header_functions():
        assert hasattr(l, fn), f"{fn} declared in include/mcnerf.h but not exported"


def test_ctypes_signatures_cover_header(built_lib):
    from mc_nerf_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()
    l = _lib.lib()
    assert l.mcnerf_abi_version() == _lib.ABI_VERSION


def test_layout_queries_need_no_gpu(built_lib):
    from mc_nerf_amd import ops
    fine, coarse = ops.Net(8, 256, 4), ops.Net(4, 128, 2)
    # reference parameter counts (631 836 / 102 428) plus 16-byte alignment padding
    assert 631836 <= ops.param_count(fine) <= 631836 + 4 * 26
    assert 102428 <= ops.param_count(coarse) <= 102428 + 4 * 18
    offs = ops.param_offsets(fine)
    assert offs[0] == 0 and all(b > a for a, b in zip(offs, offs[1:])) and all(o % 4 == 0 for o in offs)
    assert ops.tile_rows(256) == 64
    assert ops._lib.lib().mcnerf_param_count(3, 100, 1) == -1   # unsupported width


def test_ops_refuse_cpu_tensors(built_lib):
    import torch
    from mc_nerf_amd import ops, _lib
    net = ops.Net(4, 32, 2)
    with pytest.raises(_lib.McnerfError):
        ops.pack_weights(net, torch.zeros(ops.param_count(net)), torch.zeros(ops.packed_count(net)))
