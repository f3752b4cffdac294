"""ctypes binding of libfactorizer_hip.so (C ABI: include/factorizer_hip.h).

The native library is the product path for device tensors.  There is no silent fallback: if a
device tensor reaches an op and the library cannot be loaded, `lib()` raises.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfactorizer_hip.so")

FZ_OK = 0
FZ_E_UNSUPPORTED = -2
SOLVER_ID = {"mu": 0, "hals": 1}

_lock = threading.Lock()
_lib = None

_c = ctypes
_vp, _i, _i64, _f = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float

_SIGS = {
    "fz_version": ([], _i),
    "fz_last_error_string": ([], _c.c_char_p),
    "fz_launch_count": ([], _i64),
    "fz_swm_fwd": ([_vp, _vp] + [_i] * 10 + [_c.POINTER(_i), _i, _i, _i, _vp], _i),
    "fz_swm_inv": ([_vp, _vp] + [_i] * 10 + [_c.POINTER(_i), _i, _vp, _vp], _i),
    "fz_nmf_fwd": ([_vp] * 6 + [_i64] + [_i] * 5 + [_f, _vp], _i),
    "fz_nmf_bwd": ([_vp] * 7 + [_i64] + [_i] * 6 + [_f, _vp], _i),
    "fz_nmf_supported": ([_i] * 5, _i),
}


class NativeError(RuntimeError):
    pass


def declared_symbols():
    return sorted(_SIGS)


def lib():
    """Load (once) and return the native library; raise loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build it with `python -m factorizer_amd.build` "
                "(hipcc --offload-arch=gfx950). Device tensors have no fallback path.")
        h = ctypes.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(h, name)  # AttributeError if the .so does not export it
            fn.argtypes = args
            fn.restype = res
        _lib = h
    return _lib


def launch_count() -> int:
    return int(lib().fz_launch_count())


def check(rc: int, what: str):
    if rc != FZ_OK:
        msg = lib().fz_last_error_string().decode()
        raise NativeError(f"{what} failed (code {rc}): {msg}")


def stream_ptr(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


def shifts_array(shifts):
    flat = [int(v) for s in shifts for v in s]
    return (_i * len(flat))(*flat)
