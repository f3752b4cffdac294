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
# FZ_LIB_PATH: diagnostics only (A/B of two builds of this library on one box, tools/probes/build_alt.py)
LIB_PATH = os.environ.get("FZ_LIB_PATH") or os.path.join(_HERE, "libfactorizer_hip.so")

FZ_OK = 0
FZ_E_UNSUPPORTED = -2
ABI_VERSION = 6   # include/factorizer_hip.h: FZ_ABI_VERSION this binding's argument lists / descriptor layouts are written against
SOLVER_ID = {"mu": 0, "hals": 1}
STORE_F32, STORE_BF16 = 0, 1   # include/factorizer_hip.h: FZ_STORE_*
PRODUCTS_DEFAULT, PRODUCTS_SPLIT_BF16, PRODUCTS_FP32_MFMA = 0, 1, 2   # FZ_PRODUCTS_*: the `products` field of the descriptors
_products = threading.local()   # per thread: two models driven from two threads may use different pipes


def products() -> int:
    """the FZ_PRODUCTS_* value this package's layers put in their descriptors (default: the library's own default)"""
    return getattr(_products, "value", PRODUCTS_DEFAULT)


class use_products:
    """with use_products(PRODUCTS_FP32_MFMA): ...  — fp32 products of every dense layer whose FORWARD runs inside the block on
    the named pipe, through the descriptors' `products` field.  The setting is per thread; every autograd node records it in
    its ctx at forward time and re-establishes it for its backward (`capture_products` / `with_products` below), so a backward
    pass that runs after the block has exited, or on one of autograd's worker threads, multiplies on the same pipe as its
    forward did.  The C ABI has no switch to flip for this: a caller of the library sets the field per call."""

    def __init__(self, value):
        self.value = int(value)

    def __enter__(self):
        self.prev = products()
        _products.value = self.value

    def __exit__(self, *exc):
        _products.value = self.prev


# Walking order of the fused core's window launches (fz_set_tile_order, include/factorizer_hip.h): window w walks the patch
# tiles ascending for even w, descending for odd w — consecutive windows read the SAME tensors, so a window then starts where
# the previous one ended, in the part the 256 MiB Infinity Cache still holds.  FZ_TILE_ORDER=0 (read once): always ascending.
TILE_ORDER = os.environ.get("FZ_TILE_ORDER", "1") != "0"


def set_tile_order(descending: int):
    if TILE_ORDER:
        lib().fz_set_tile_order(int(descending) & 1)


def capture_products(fwd):
    """decorator of an autograd Function's forward(ctx, ...): remember the thread's products setting in the ctx"""
    import functools

    @functools.wraps(fwd)
    def wrapper(ctx, *a, **kw):
        ctx._fz_products = products()
        return fwd(ctx, *a, **kw)
    return wrapper


def with_products(bwd):
    """decorator of the matching backward(ctx, ...): run it under the setting its forward saw"""
    import functools

    @functools.wraps(bwd)
    def wrapper(ctx, *a, **kw):
        with use_products(getattr(ctx, "_fz_products", PRODUCTS_DEFAULT)):
            return bwd(ctx, *a, **kw)
    return wrapper


def act_dtype(t: torch.Tensor) -> int:
    """FZ_STORE_* code of an activation tensor (fp32, or bf16 storage with fp32 arithmetic)."""
    if t.dtype == torch.float32:
        return STORE_F32
    if t.dtype == torch.bfloat16:
        return STORE_BF16
    raise TypeError(f"native kernels take float32 or bfloat16 activations, got {t.dtype} "
                    "(float16 is refused on purpose: the NMF eps 1e-16 underflows in it, SURVEY.md §5)")

_lock = threading.Lock()
_lib = None

_c = ctypes
_vp, _i, _i64, _f = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float

_SIGS = {
    "fz_version": ([], _i),
    "fz_abi_version": ([], _i),
    "fz_set_tile_order": ([_i], _i),
    "fz_last_error_string": ([], _c.c_char_p),
    "fz_launch_count": ([], _i64),
    "fz_swm_fwd": ([_vp, _vp] + [_i] * 10 + [_c.POINTER(_i), _i, _i, _i, _vp], _i),
    "fz_swm_inv": ([_vp, _vp] + [_i] * 10 + [_c.POINTER(_i), _i, _vp, _i, _vp], _i),
    "fz_nmf_fwd": ([_vp] * 6 + [_i64] + [_i] * 5 + [_f, _i, _vp], _i),
    "fz_nmf_bwd": ([_vp] * 7 + [_i64] + [_i] * 6 + [_f, _i, _vp], _i),
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
        try:
            h.fz_abi_version.restype = ctypes.c_int
            abi = int(h.fz_abi_version())
        except AttributeError:
            abi = None
        if abi != ABI_VERSION:
            raise NativeError(f"{LIB_PATH} has ABI revision {abi}, this binding is written against {ABI_VERSION} "
                              "(include/factorizer_hip.h: FZ_ABI_VERSION): rebuild with `python -m factorizer_amd.build`")
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


# ---- descriptor structs of the GEMM family (include/factorizer_hip.h) -----------------------
_fp = _c.POINTER(_c.c_float)


class GemmDesc(_c.Structure):
    _fields_ = [("x", _vp * 4), ("nsrc", _i), ("src_mode", _i), ("c0", _i), ("Cin", _i), ("Vin", _i64),
                ("Di", _i), ("Hi", _i), ("Wi", _i), ("w", _vp), ("w_t", _i), ("ldw", _i), ("M", _i), ("K", _i),
                ("bias", _vp), ("ln", _i), ("ln_g", _vp), ("ln_b", _vp), ("ln_eps", _f), ("stats_out", _vp),
                ("bact", _i), ("bmul", _vp), ("bmul_kind", _i), ("eact", _i), ("res", _vp), ("emul", _vp), ("emul_kind", _i), ("y", _vp),
                ("Ncol", _i64), ("Ho", _i), ("Wo", _i), ("B", _i), ("loader", _i), ("epilogue", _i), ("lnb_x", _vp), ("lnb_stats", _vp), ("lnb_g", _vp),
                ("lnb_gadd", _vp), ("lnb_part", _vp), ("act_dtype", _i), ("products", _i), ("tune", _i)]


class MlpDesc(_c.Structure):
    _fields_ = [("mode", _i), ("inp", _vp), ("w1", _vp), ("w2", _vp), ("b1", _vp), ("b2", _vp), ("ln_g", _vp),
                ("ln_b", _vp), ("ln_eps", _f), ("stats", _vp), ("z1", _vp), ("gz1", _vp), ("x1", _vp), ("out", _vp),
                ("part", _vp), ("B", _i), ("C", _i), ("H", _i), ("V", _i64), ("act_dtype", _i), ("wpart", _vp),
                ("gw1", _vp), ("gb1", _vp), ("gw2", _vp), ("gb2", _vp), ("gln", _vp), ("glp", _vp), ("products", _i),
                ("pre_in", _vp), ("pre_w", _vp), ("pre_b", _vp), ("pre_res", _vp), ("pre_out", _vp),
                ("post_w", _vp), ("post_b", _vp), ("post_out", _vp), ("post_m", _i)]


class BlockPrologue(_c.Structure):
    """fz_block_prologue (include/factorizer_hip.h)"""
    _fields_ = [("ln_g", _vp), ("ln_b", _vp), ("ln_eps", _f), ("w", _vp), ("t", _vp), ("stats", _vp)]


class GemmDwDesc(_c.Structure):
    _fields_ = [("g", _vp), ("q", _vp), ("w", _vp), ("ln", _i), ("stats", _vp), ("ln_g", _vp), ("ln_b", _vp), ("gadd", _vp),
                ("y", _vp), ("gln", _vp), ("wpart", _vp), ("gw", _vp), ("gb", _vp), ("B", _i), ("C", _i), ("V", _i64),
                ("act_dtype", _i), ("ldgw", _i), ("ldw", _i)]


class WgradDesc(_c.Structure):
    _fields_ = [("p", _vp), ("M", _i), ("pmul", _vp), ("pmul_kind", _i), ("q", _vp * 4), ("nsrc", _i),
                ("src_mode", _i), ("c0", _i), ("Cin", _i), ("K", _i), ("Vq", _i64), ("D", _i), ("H", _i),
                ("W", _i), ("N", _i64), ("Ho", _i), ("Wo", _i), ("stats", _vp), ("qact", _i), ("ln_g", _vp),
                ("ln_b", _vp), ("gw", _vp), ("gbias", _vp), ("accumulate", _i), ("B", _i), ("loader", _i),
                ("act_dtype", _i), ("products", _i)]


_SIGS.update({
    "fz_gcorr_supported": ([_i] * 5, _i),
    "fz_gcorr": ([_vp] * 5 + [_i] * 11 + [_f, _vp], _i),
    "fz_gcorr_wgrad_workspace_bytes": ([_i] * 10, _i64),
    "fz_gcorr_wgrad": ([_vp] * 4 + [_i] * 11 + [_vp], _i),
    "fz_gnmf_supported": ([_i, _i64, _i, _i, _i], _i),
    "fz_gnmf_workspace_bytes": ([_i64, _i, _i64, _i, _i, _i], _i64),
    "fz_gnmf_launches": ([_i, _i, _i], _i),
    "fz_gnmf_fwd": ([_vp] * 6 + [_i64, _i, _i64, _i, _i, _i, _f, _vp, _vp], _i),
    "fz_gnmf_bwd": ([_vp] * 7 + [_i64, _i, _i64, _i, _i, _i, _i, _f, _vp, _vp], _i),
    "fz_nmf_cf_supported": ([_i] * 11, _i),
    "fz_nmf_cf_fwd": ([_vp] * 4 + [_i] * 5 + [_c.POINTER(_i)] + [_i] * 5 + [_f, _i, _vp], _i),
    "fz_nmf_cf_bwd": ([_vp] * 5 + [_i] * 5 + [_c.POINTER(_i)] + [_i] * 7 + [_f, _i, _vp], _i),
    "fz_nmf_pcf_supported": ([_i] * 11, _i),
    "fz_nmf_pcf_fwd": ([_vp] * 4 + [_i] * 8 + [_c.POINTER(_i)] + [_i] * 5 + [_f, _i, _vp], _i),
    "fz_nmf_pcf_bwd": ([_vp] * 5 + [_i] * 8 + [_c.POINTER(_i)] + [_i] * 7 + [_f, _i, _vp], _i),
    "fz_nmf_pcf_bwd_prefers_separate": ([_i] * 4, _i),
    "fz_act_add": ([_vp, _vp, _i64, _i, _vp], _i),
    "fz_mlp_pre_supported": ([_i, _i, _i64, _i], _i),
    "fz_conv3_prologue_supported": ([_i] * 4, _i),
    "fz_conv3_fwd2": ([_vp] * 4 + [_i] * 8 + [_c.POINTER(BlockPrologue), _vp], _i),
    "fz_upcat2": ([_vp, _vp, _vp, _i, _vp, _vp, _vp] + [_i] * 7 + [_c.POINTER(BlockPrologue), _vp], _i),
    "fz_gemm": ([_c.POINTER(GemmDesc), _vp], _i),
    "fz_gemm_bx_enable": ([_i], _i),
    "fz_gemm_lnbwd_partials": ([_c.POINTER(GemmDesc)], _i64),
    "fz_reduce_rows": ([_vp, _i64, _i, _vp, _vp, _vp], _i),
    "fz_adamw_step": ([_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _f, _vp], _i),
    "fz_sw_gather": ([_vp, _vp] + [_i] * 10 + [_vp], _i),
    "fz_sw_accumulate": ([_vp] * 6 + [_f] + [_i] * 10 + [_vp], _i),
    "fz_sw_finalize": ([_vp, _vp, _i, _i64, _vp], _i),
    "fz_mlp_supported": ([_i, _i, _i64], _i),
    "fz_mlp_partials": ([_i, _i64], _i64),
    "fz_mlp_wgrad_rows": ([_i, _i64], _i),
    "fz_gemm_dw_rows": ([_i, _i64], _i),
    "fz_gemm_dw_workspace_bytes": ([_i, _i64], _i64),
    "fz_gemm_dw": ([_c.POINTER(GemmDwDesc), _vp], _i),
    "fz_mlp_wgrad_workspace_bytes": ([_i, _i64], _i64),
    "fz_mlp_chain": ([_c.POINTER(MlpDesc), _vp], _i),
    "fz_head_bwd_rows": ([], _i),
    "fz_head_bwd_workspace_bytes": ([], _i64),
    "fz_head_bwd": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i, _vp], _i),
    "fz_head_fwd": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i, _vp], _i),
    "fz_upcat_supported": ([_i] * 5, _i),
    "fz_upcat_compose": ([_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "fz_upcat_wgrads": ([_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp], _i),
    "fz_upcat": ([_vp, _vp, _vp, _i, _vp, _vp, _vp] + [_i] * 7 + [_vp], _i),
    "fz_wgrad": ([_c.POINTER(WgradDesc), _vp, _vp], _i),
    "fz_wgrad_group": ([_c.POINTER(_c.POINTER(WgradDesc)), _c.POINTER(_vp), _i, _vp], _i),
    "fz_wgrad_workspace_bytes": ([_c.POINTER(WgradDesc)], _i64),
    "fz_conv3_fwd": ([_vp] * 4 + [_i] * 8 + [_vp], _i),
    "fz_conv3_wgrad_chunks": ([_i] * 4, _i),
    "fz_conv3_wgrad_partials": ([_vp] * 4 + [_i] * 8 + [_vp], _i),
    "fz_chunk_reduce": ([_vp, _i, _i64, _vp, _i, _vp], _i),
    "fz_chunk_reduce_ld": ([_vp, _i, _i64, _i64, _vp, _i, _vp], _i),
    "fz_finish_defer": ([_i], _i),
    "fz_finish_pending": ([], _i),
    "fz_finish_flush": ([_vp], _i),
    "fz_finish_flush_all": ([_vp], _i),
    "fz_dice_bce_chunks": ([_i64], _i),
    "fz_dice_bce_sums": ([_vp, _vp, _vp, _i, _i64, _vp], _i),
    "fz_dice_bce_grad": ([_vp, _vp, _vp, _vp, _i, _i64, _f, _f, _vp, _vp], _i),
    "fz_dice_ce_sums": ([_vp, _vp, _vp, _i, _i, _i64, _vp], _i),
    "fz_dice_ce_grad": ([_vp, _vp, _vp, _vp, _i, _i, _i64, _f, _f, _vp, _vp], _i),
    "fz_dice_ce_finish": ([_vp, _i, _i, _i64, _f, _vp, _vp, _vp], _i),
    "fz_rowsum_chunks": ([_i64], _i),
    "fz_rowsum": ([_vp, _vp, _vp, _i, _i, _i64, _i, _vp], _i),
    "fz_ln_fwd": ([_vp] * 5 + [_i, _i, _i64, _f, _i, _vp], _i),
    "fz_ln_bwd": ([_vp] * 8 + [_i, _i, _i64, _i, _vp], _i),
    "fz_ln_bwd_workspace_bytes": ([_i], _i64),
    "fz_ln_bwd_workspace_bytes2": ([_i, _i, _i64], _i64),
})
