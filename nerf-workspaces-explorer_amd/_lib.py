"""ctypes binding of libnwe_hip.so (include/nwe.h).  There is no fallback: if the library is missing
or a symbol is absent, importing the render path fails loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NWE_LIB") or os.path.join(_HERE, "libnwe_hip.so")   # NWE_LIB: timing experiments with variant builds

NWE_OK, NWE_ERR_INVALID, NWE_ERR_UNSUPPORTED, NWE_ERR_HIP, NWE_ERR_STATE = 0, 1, 2, 3, 4
NET_COARSE, NET_FINE = 0, 1
PREC_F16X3, PREC_F16X1, PREC_F32 = 0, 1, 2
PRECISIONS = {"f16x3": PREC_F16X3, "f16x1": PREC_F16X1, "f32": PREC_F32}

OUTPUT_FIELDS = ("rgb", "depth", "acc", "disp", "z_std", "rgb_coarse", "depth_coarse", "acc_coarse", "disp_coarse",
                 "raw_coarse", "raw_fine", "z_fine", "weights_coarse", "sample_cond", "sample_amp", "sample_switch", "feat_map", "flags")

# every symbol include/nwe.h declares (tests/test_abi.py checks the library exports all of them)
SYMBOLS = ("nwe_create", "nwe_destroy", "nwe_last_error", "nwe_set_network", "nwe_set_sampling", "nwe_render", "nwe_render_tiled",
           "nwe_create_rays", "nwe_render_rays", "nwe_to8b", "nwe_flops_per_eval", "nwe_last_kernel_ms", "nwe_packed_bytes",
           "nwe_packed_copy", "nwe_packed_bias_count", "nwe_packed_bias_copy", "nwe_packed_scale",
           "nwe_debug_set_fine_depths", "nwe_debug_set_raw", "nwe_debug_set_coarse_weights", "nwe_debug_set_fold", "nwe_set_train_tables", "nwe_set_white_background", "nwe_debug_set_decomposition", "nwe_debug_last_plan", "nwe_debug_set_stamps", "nwe_selftest",
           "nwe_last_warning", "nwe_debug_peer_access", "nwe_set_network_no_view_dirs", "nwe_last_launch_parts")


class Outputs(C.Structure):
    _fields_ = [("struct_bytes", C.c_uint64)] + [(name, C.c_void_p) for name in OUTPUT_FIELDS]

    def __init__(self, **kw):
        super().__init__(struct_bytes=C.sizeof(Outputs), **kw)


_lib = None


def build_hint() -> str:
    return ("build it with `make -C nerf-workspaces-explorer_amd/csrc` (or `python -c 'import __graft_entry__ as g; "
            "g.build()'`); the render path has no CPU or torch fallback")


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"HIP extension {LIB_PATH} is missing: {build_hint()}")
    lib = C.CDLL(LIB_PATH)
    P, I, F, I64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
    sig = {
        "nwe_create": (I, [C.POINTER(P), I]),
        "nwe_destroy": (None, [P]),
        "nwe_last_error": (C.c_char_p, [P]),
        "nwe_set_network": (I, [P, I, I, I, I, I, I, C.POINTER(P), C.POINTER(P)]),
        "nwe_set_network_no_view_dirs": (I, [P, I, I, I, I, I, I, C.POINTER(P), C.POINTER(P)]),
        "nwe_set_sampling": (I, [P, P, P, I, P, I]),
        "nwe_render": (I, [P, P, I, I, I, F, F, F, F, F, F, I, I, I, C.POINTER(Outputs), P]),
        "nwe_render_tiled": (I, [C.POINTER(P), I, P, I, I, I, F, F, F, F, F, F, I, P, P, P, P, P]),
        "nwe_create_rays": (I, [P, P, I, I, I, F, F, F, F, F, F, I, I, P, P]),
        "nwe_render_rays": (I, [P, P, I64, I, C.POINTER(Outputs), P]),
        "nwe_to8b": (I, [P, P, P, I64, P]),
        "nwe_flops_per_eval": (I64, [P, I]),
        "nwe_last_kernel_ms": (F, [P]),
        "nwe_last_launch_parts": (I, [P, C.POINTER(F), C.POINTER(I64)]),
        "nwe_packed_bytes": (I64, [P, I]),
        "nwe_packed_copy": (I, [P, I, P, I64]),
        "nwe_packed_bias_count": (I64, [P, I]),
        "nwe_packed_bias_copy": (I, [P, I, P, I64]),
        "nwe_packed_scale": (F, [P, I]),
        "nwe_debug_set_fine_depths": (I, [P, P]),
        "nwe_debug_set_raw": (I, [P, P, P]),
        "nwe_debug_set_coarse_weights": (I, [P, P]),
        "nwe_debug_set_fold": (I, [P, I]),
        "nwe_set_train_tables": (I, [P, P, P, P, P]),
        "nwe_set_white_background": (I, [P, I]),
        "nwe_debug_set_decomposition": (I, [P, I]),
        "nwe_debug_last_plan": (I, [P]),
        "nwe_debug_set_stamps": (I, [P, P]),
        "nwe_selftest": (I, [P, C.POINTER(C.c_int32)]),
        "nwe_last_warning": (C.c_char_p, [P]),
        "nwe_debug_peer_access": (I, [P, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is missing: loud by design
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib
