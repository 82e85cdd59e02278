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


def test_general_topology_codes_and_layouts(built_lib):
    """The `skip` argument's mask form (include/mcnerf.h): any `skips` list, SH degree 0 .. 3, 1 .. 10 encoding frequencies.  The flat
    parameter layout of such a net is the reference's state dict (model/net_block.py:51-65) in order, every tensor 16-byte aligned;
    the default topology keeps its legacy codes; the register-chain entry points refuse what they do not take."""
    import numpy as np
    from mc_nerf_amd import ops
    l = ops._lib.lib()
    assert ops.skip_code([4], 8) == 4 and ops.skip_code([], 4) == -1 and ops.skip_code([0, 9], 8) == -1        # (0 and >= depth are not skip layers)
    code = ops.skip_code([2, 4, 6], 8)
    net = ops.Net(8, 64, code)
    assert code >= ops.SKIP_MASK and net.skips == [2, 4, 6] and net.multi_skip and net.deg == 2 and net.n_freqs == 10
    assert [net.in_features(i) for i in range(8)] == [63, 64, 127, 64, 127, 64, 127, 64]
    g = ops.Net(4, 32, ops.skip_code([2], 4, deg=3, n_freqs=6))
    assert (g.skips, g.deg, g.n_sh, g.n_shp, g.n_freqs, g.n_enc) == ([2], 3, 48, 64, 6, 39)
    assert g.fp32_only and net.fp32_only and not ops.Net(8, 256, 4).fp32_only
    for n_ in (net, g, ops.Net(4, 128, ops.skip_code([1, 3], 4, deg=0)), ops.Net(8, 256, 4)):
        offs, shapes = ops.param_offsets(n_), n_.shapes()
        assert len(offs) == len(shapes) == 2 * n_.depth + 8
        end = 0
        for o, shp in zip(offs, shapes):
            assert o % 4 == 0 and end <= o < end + 4          # next 16-byte boundary after the previous tensor
            end = o + int(np.prod(shp))
        assert end <= ops.param_count(n_) < end + 4
        assert ops.packed_count(n_) > 0
    assert ops.param_count(ops.Net(8, 256, ops.skip_code([4], 8))) == ops.param_count(ops.Net(8, 256, 4))
    # refused: degree 4, 11 frequencies, a mask bit at or above the depth; the register-chain sizes of a net they do not take
    assert l.mcnerf_param_count(4, 32, (1 << 8) | 0x80 | (4 << 4)) == -1
    assert l.mcnerf_param_count(4, 32, (1 << 8) | 12) == -1
    assert l.mcnerf_param_count(4, 32, (1 << 8) | (1 << 13)) == -1
    assert l.mcnerf_packed_bytes_16(*net.triple, 2, 0) == -1 and l.mcnerf_packed_bytes_16(*g.triple, 0, 0) == -1
    assert l.mcnerf_packed_bytes_16(8, 256, 4, 2, 0) > 0
    with pytest.raises(ValueError):
        ops.skip_code([2], 4, deg=4)
    with pytest.raises(ValueError):
        ops.skip_code([2], 4, n_freqs=11)


def test_register_chain_dtypes_and_workspace_sizes(built_lib):
    """`dtype` of the `_16` entry points (include/mcnerf.h): 0 = f16, 1 = bf16, 2 = f16x3 (hi + lo planes), 3 = f16x3h (f16x3's chains,
    hi planes only): dtype 3 packs its weights like dtype 2, keeps dtype 2's fp32 sh.2-output tile, and sizes every other workspace
    like dtype 0; anything else is refused."""
    from mc_nerf_amd import ops
    l = ops._lib.lib()
    assert ops.DTYPE16 == {"f16": 0, "bf16": 1, "f16x3": 2, "f16x3h": 3} and set(ops.PRECISIONS) == {"f32", *ops.DTYPE16}
    for depth, width, skip in ((8, 256, 4), (4, 128, 2), (8, 64, 4)):
        for bwd in (0, 1):
            assert l.mcnerf_packed_bytes_16(depth, width, skip, 3, bwd) == l.mcnerf_packed_bytes_16(depth, width, skip, 2, bwd) >= l.mcnerf_packed_bytes_16(depth, width, skip, 0, bwd) > 0      # (slab padding differs between the streams)
        for cap in (1, 4096, 3_200_000):
            for which in (0, 1, 2, 3):
                assert l.mcnerf_ws_bytes_16(depth, width, 3, cap, which) == l.mcnerf_ws_bytes_16(depth, width, 0, cap, which) > 0
                if which != 2:
                    assert l.mcnerf_ws_bytes_16(depth, width, 2, cap, which) == 2 * l.mcnerf_ws_bytes_16(depth, width, 0, cap, which)
            assert l.mcnerf_ws_bytes_16(depth, width, 3, cap, 4) == l.mcnerf_ws_bytes_16(depth, width, 2, cap, 4) == 2 * l.mcnerf_ws_bytes_16(depth, width, 0, cap, 4)
    assert l.mcnerf_packed_bytes_16(8, 256, 4, 4, 0) == -1 and l.mcnerf_ws_bytes_16(8, 256, -1, 4096, 0) == -1 and l.mcnerf_ws_bytes_16(8, 256, 4, 4096, 0) == -1
