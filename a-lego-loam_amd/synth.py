"""Synthetic scene S0 / trajectory T0 generator (csrc/synth.cpp) — data, not hot path."""
import ctypes as C
import os

import numpy as np

from .params import AlegoParams

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libalego_synth.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing — run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = C.CDLL(path)
        _lib.alego_synth_scan.restype = C.c_int
        _lib.alego_synth_scan.argtypes = [C.POINTER(AlegoParams), C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_int]
        _lib.alego_synth_pose.argtypes = [C.c_int, C.c_long, C.POINTER(C.c_double)]
        _lib.alego_synth_default_params.argtypes = [C.POINTER(AlegoParams), C.c_int, C.c_int]
        _lib.alego_synth_params_sizeof.restype = C.c_int
    return _lib


def default_params(n_scan=16, horizon=1800) -> AlegoParams:
    """Reference defaults (include/alego_params.h) for an n_scan x horizon sensor."""
    p = AlegoParams()
    lib().alego_synth_default_params(C.byref(p), n_scan, horizon)
    return p


def scan(params: AlegoParams, scan_index: int, stream: int = 0, flags: int = 0) -> np.ndarray:
    """One scan as an (n,4) float32 array (x,y,z,intensity), ring-major order."""
    cap = params.n_scan * params.horizon_scan
    out = np.empty((cap, 4), dtype=np.float32)
    n = lib().alego_synth_scan(C.byref(params), stream, scan_index, flags, out.ctypes.data, cap)
    return out[:n].copy()


def pose(scan_index: int, stream: int = 0) -> np.ndarray:
    """Ground-truth sensor pose (x, y, z, yaw) in the world frame."""
    buf = (C.c_double * 4)()
    lib().alego_synth_pose(stream, scan_index, buf)
    return np.array(buf[:], dtype=np.float64)
