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


def test_missing_library_fails_loudly(monkeypatch):
    from factorizer_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libfactorizer_hip.so")
    with pytest.raises(_native.NativeError):
        _native.lib()
