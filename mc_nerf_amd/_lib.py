"""ctypes binding of libmcnerf.so (the C ABI declared in include/mcnerf.h).

There is deliberately NO fallback: if the HIP library is missing or does not load, importing the
symbols raises, and every op in ``mc_nerf_amd.ops`` fails loudly.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_longlong, c_void_p

# torch ships its own HIP runtime (torch/lib/libamdhip64.so).  It MUST be in the process before
# libmcnerf.so is loaded, so that the library's libamdhip64 dependency resolves to the same runtime
# instance that owns torch's streams and allocations (two runtimes in one process do not share devices).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCNERF_LIB selects another build of the SAME library (kernel ablation / tuning variants, scripts/ablate.sh)
LIB_PATH = os.environ.get("MCNERF_LIB") or os.path.join(_HERE, "libmcnerf.so")
ABI_VERSION = 6

_P = c_void_p
_I = c_int
_L = c_longlong

# name -> (restype, argtypes); mirrors include/mcnerf.h line by line
SIGNATURES = {
    "mcnerf_abi_version": (_I, []),
    "mcnerf_last_error": (c_char_p, []),
    "mcnerf_param_count": (_L, [_I, _I, _I]),
    "mcnerf_packed_count": (_L, [_I, _I, _I]),
    "mcnerf_tile_rows": (_I, [_I]),
    "mcnerf_param_offsets": (_I, [_I, _I, _I, _P]),
    "mcnerf_pack_weights": (_I, [_I, _I, _I, _P, _P, _P]),
    "mcnerf_raygen_fwd": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "mcnerf_raygen_bwd": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "mcnerf_mlp_fwd": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _L, _P, _P, _P, _P]),
    "mcnerf_encode": (_I, [_P, _P, _I, _I, _P, _P]),
    "mcnerf_upload_f32": (_I, [_P, _P, _I, _P]),
    "mcnerf_train_loss": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P]),
    "mcnerf_scale3": (_I, [_P, _I, _P, _I, _P, _I, _P, _P]),
    "mcnerf_sample_perm": (_I, [_P, ctypes.c_longlong, _I, _P, _P]),
    "mcnerf_mlp_apply": (_I, [_I, _I, _I, _P, _P, _P, _P, _I, _P, _P]),
    "mcnerf_encode_bwd": (_I, [_P, _P, _I, _I, _P, _P, _P]),
    "mcnerf_sync_finish": (_I, [_P, _L, _I, _I, _P, _P, _P]),
    "mcnerf_mlp_apply_save": (_I, [_I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _L, _P, _P, _P, _P]),
    "mcnerf_mlp_apply_bwd": (_I, [_I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P]),
    "mcnerf_mlp_bwd": (_I, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _L, _P, _P,
                            _P, _P, _P, _P, _P]),
    "mcnerf_mlp_dw": (_I, [_I, _I, _I, _P, _I, _P, _P, _P, _P, _L, _P, _P]),
    "mcnerf_packed_bytes_16": (_L, [_I, _I, _I, _I, _I]),
    "mcnerf_pack_weights_16": (_I, [_I, _I, _I, _P, _P, _P, _I, _P, _P]),
    "mcnerf_ws_bytes_16": (_L, [_I, _I, _I, _L, _I]),
    "mcnerf_mlp_fwd_16": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _L, _P, _P, _P, _P]),
    "mcnerf_mlp_bwd_16": (_I, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _L, _P, _P,
                               _P, _P, _P, _P, _P, _P]),
    "mcnerf_mlp_dw_16": (_I, [_I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _L, _P, _P, _P]),
    "mcnerf_composite_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "mcnerf_composite_bwd": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "mcnerf_select_fine": (_I, [_P, _P, c_float, _I, _I, _I, c_float, _P, _P, _P, _P, _P, _P]),
    "mcnerf_cap_gather": (_I, [_P, _P, _I, _P, _P, _P]),
    "mcnerf_cap_ws_words": (_L, []),
    "mcnerf_cap_random": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "mcnerf_gather_gt": (_I, [_P, _I, _P, _I, _P, _P]),
    "mcnerf_camera_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "mcnerf_camera_bwd": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mcnerf_reproj_loss_fwd": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "mcnerf_reproj_loss_bwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "mcnerf_radam_step": (_I, [_I, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, c_float, c_float, _I, _P, _I, _P]),
}

_lib = None


class McnerfError(RuntimeError):
    pass


def lib():
    """Returns the loaded library; raises McnerfError if it is missing (build it with
    ``python -m mc_nerf_amd.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise McnerfError(f"{LIB_PATH} not found: the HIP extension is required (python -m mc_nerf_amd.build); "
                          "there is no CPU fallback")
    try:
        l = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise McnerfError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            f = getattr(l, name)
        except AttributeError as e:
            raise McnerfError(f"{LIB_PATH} does not export {name}") from e
        f.restype, f.argtypes = res, args
    if l.mcnerf_abi_version() != ABI_VERSION:
        raise McnerfError("libmcnerf.so ABI version mismatch; rebuild")
    _lib = l
    return l


def call(name, *args):
    """Calls a status-returning entry point and raises with the library's message on failure."""
    l = lib()
    rc = getattr(l, name)(*args)
    if rc != 0:
        raise McnerfError(f"{name} failed ({rc}): {l.mcnerf_last_error().decode()}")
