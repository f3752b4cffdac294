"""The C-ABI shared library builds for gfx950, loads, and exports every symbol that
include/factorizer_hip.h declares (no compute calls — there is no GPU in this container)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from factorizer_amd import build
    return build.build(verbose=False)


def header_functions():
    src = open(os.path.join(ROOT, "include", "factorizer_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fz_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    names = header_functions()
    assert {"fz_swm_fwd", "fz_swm_inv", "fz_nmf_fwd", "fz_nmf_bwd", "fz_version"} <= set(names)
    lib = ctypes.CDLL(built_lib)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/factorizer_hip.h but not exported"


def test_integration_md_names_every_entry_point():
    """INTEGRATION.md maps each entry point to the reference call site it replaces: none may be missing from it."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in header_functions() if n not in text]
    assert not missing, missing


def test_python_binding_matches_header(built_lib):
    from factorizer_amd import _native
    assert set(_native.declared_symbols()) <= set(header_functions())
    lib = _native.lib()
    assert lib.fz_version() >= 100
    assert lib.fz_abi_version() == _native.ABI_VERSION   # the loader refuses a library of another revision
    assert lib.fz_last_error_string() is not None


def test_host_side_argument_checks(built_lib):
    """Entry points validate shapes before touching the device (no GPU needed)."""
    from factorizer_amd import _native
    lib = _native.lib()
    sh = _native.shifts_array([(0, 0, 0)])
    rc = lib.fz_swm_fwd(None, None, 1, 30, 8, 8, 8, 8, 4, 4, 4, 1, sh, 4, 0, 1, None)
    assert rc == -1 and b"head_dim" in lib.fz_last_error_string()
    rc = lib.fz_swm_fwd(None, None, 1, 32, 10, 12, 10, 8, 8, 8, 8, 1, sh, 4, 0, 1, None)
    assert rc == -1 and b"patch" in lib.fz_last_error_string()
    assert lib.fz_nmf_supported(8, 512, 1, 5, 5) == 1
    assert lib.fz_nmf_supported(8, 512, 2, 10, 10) == 1
    assert lib.fz_nmf_supported(16, 262144, 1, 5, 5) == 0
    assert lib.fz_nmf_supported(8, 512, 5, 5, 5) == 0
    rc = lib.fz_nmf_fwd(None, None, None, None, None, None, 4, 8, 512, 9, 5, 1, 1e-16, 0, None)
    assert rc == -2
    # storage type of the activations (FZ_STORE_F32 / FZ_STORE_BF16): anything else is refused by the host code
    d = _native.GemmDesc()
    d.act_dtype = 7
    assert lib.fz_gemm(ctypes.byref(d), None) == -4 and b"act_dtype" in lib.fz_last_error_string()
    w = _native.WgradDesc()
    w.act_dtype = 7
    assert lib.fz_wgrad(ctypes.byref(w), ctypes.c_void_p(8), None) == -4
    rc = lib.fz_swm_inv(ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 8, 8, 8, 8, 8, 8, 8, 8, 1, sh, 1, None, 5, None)
    assert rc == -4 and b"act_dtype" in lib.fz_last_error_string()


def test_fused_chain_and_finish_queue_argument_checks(built_lib):
    """Round-5 entry points: which shapes take the out-projection in front of the MLP chain, what the descriptor must carry, and
    the finish queue's host-side state — all decided before anything touches the device."""
    from factorizer_amd import _native
    lib = _native.lib()
    S = _native.PRODUCTS_SPLIT_BF16
    assert lib.fz_mlp_pre_supported(32, 64, 128 ** 3, S) == 1 and lib.fz_mlp_pre_supported(32, 64, 120, S) == 1
    assert lib.fz_mlp_pre_supported(64, 128, 64 ** 3, S) == 1          # 1024 tiles of 256 voxels per sample: even
    assert lib.fz_mlp_pre_supported(64, 128, 120, S) == 0              # one tile: the 512-thread form walks tiles in pairs
    assert lib.fz_mlp_pre_supported(64, 128, 64 ** 3, _native.PRODUCTS_FP32_MFMA) == 0
    assert lib.fz_mlp_pre_supported(32, 128, 128 ** 3, S) == 0 and lib.fz_mlp_pre_supported(128, 256, 32 ** 3, S) == 0
    p8 = 8   # (a non-null pointer value the host code never dereferences)
    d = _native.MlpDesc()
    d.mode, d.B, d.C, d.H, d.V, d.act_dtype, d.products = 0, 1, 64, 128, 64 ** 3, _native.STORE_F32, S
    d.w1 = d.w2 = d.out = d.z1 = d.stats = d.ln_g = d.ln_b = p8
    d.pre_in = p8                                                          # ... without pre_w / pre_res / pre_out
    assert lib.fz_mlp_chain(ctypes.byref(d), None) == -4 and b"pre_in" in lib.fz_last_error_string()
    d.pre_w = d.pre_res = d.pre_out = p8
    d.post_out, d.post_w, d.post_m = p8, p8, 3                             # the head rides only in the C = 32 chain
    assert lib.fz_mlp_chain(ctypes.byref(d), None) == -4 and b"post_out" in lib.fz_last_error_string()
    d.post_out = None
    d.V = 120
    assert lib.fz_mlp_chain(ctypes.byref(d), None) == -2                   # FZ_E_UNSUPPORTED (fz_mlp_pre_supported says no)
    # finish queue: state only
    assert lib.fz_finish_defer(-1) == 0 and lib.fz_finish_pending() == 0 and lib.fz_finish_flush(None) == 0
    assert lib.fz_finish_defer(1) == 0 and lib.fz_finish_defer(-1) == 1 and lib.fz_finish_defer(0) == 1
    assert lib.fz_chunk_reduce_ld(ctypes.c_void_p(8), 4, 96, 64, ctypes.c_void_p(8), 0, None) == -4   # ld < n


def test_missing_library_fails_loudly(monkeypatch):
    from factorizer_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libfactorizer_hip.so")
    with pytest.raises(_native.NativeError):
        _native.lib()
