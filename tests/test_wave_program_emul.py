"""CPU check of the per-wave NMF program (factorizer_amd/csrc/nmf_core.h) through its host
lock-step emulation build (tests/emul/emul.cpp): the same source that runs on gfx950, with
F = 64-lane vector.  Compared with the reference goldens and with the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import cpu_ref as O

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emul", "emul.cpp")
LIB = os.path.join(HERE, "emul", "_fz_emul.so")
CORE = os.path.join(os.path.dirname(HERE), "factorizer_amd", "csrc", "nmf_core.h")


@pytest.fixture(scope="module")
def emu():
    newest = max(os.path.getmtime(SRC), os.path.getmtime(CORE), os.path.getmtime(CORE.replace("nmf_core.h", "nmf_gram.h")))
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC])
    lib = ctypes.CDLL(LIB)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.emu_nmf_fwd.argtypes = [fp, fp, fp, fp, fp, fp, ctypes.c_int64] + [ctypes.c_int] * 5 + [ctypes.c_float]
    lib.emu_nmf_bwd.argtypes = [fp] * 7 + [ctypes.c_int64] + [ctypes.c_int] * 6 + [ctypes.c_float]
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def emu_fwd(lib, x, u0, v0, T, solver):
    M, N = x.shape[-2:]
    R = u0.shape[1]
    xn = np.ascontiguousarray(x.reshape(-1, M, N).numpy())
    nmat = xn.shape[0]
    y = np.empty_like(xn)
    u = np.empty((nmat, M, R), np.float32)
    v = np.empty((nmat, N, R), np.float32)
    rc = lib.emu_nmf_fwd(_p(xn), _p(u0.numpy()), _p(v0.numpy()), _p(y), _p(u), _p(v), nmat, M, N, R, T,
                         {"mu": 0, "hals": 1}[solver], 1e-16)
    assert rc == 0
    lead = x.shape[:-2]
    return (torch.from_numpy(y).reshape(x.shape), torch.from_numpy(u).reshape(*lead, M, R),
            torch.from_numpy(v).reshape(*lead, N, R))


def emu_bwd(lib, x, u0, v0, gy, T, G, solver):
    M, N = x.shape[-2:]
    R = u0.shape[1]
    xn = np.ascontiguousarray(x.reshape(-1, M, N).numpy())
    gyn = np.ascontiguousarray(gy.reshape(-1, M, N).numpy())
    gx = np.empty_like(xn)
    rc = lib.emu_nmf_bwd(_p(xn), _p(u0.numpy()), _p(v0.numpy()), _p(gyn), None, None, _p(gx), xn.shape[0],
                         M, N, R, T, G, {"mu": 0, "hals": 1}[solver], 1e-16)
    assert rc == 0
    return torch.from_numpy(gx).reshape(x.shape)


from test_oracle_golden import NMF_CASES  # noqa: E402


@pytest.mark.parametrize("name", sorted(NMF_CASES))
def test_emul_vs_reference_goldens(emu, golden, name):
    g = golden("g2_nmf").case(name)
    kw = NMF_CASES[name]
    T = kw["num_iters"]
    G = kw.get("num_grad_steps") or T
    y, u, v = emu_fwd(emu, g["x"], g["u0"], g["v0"], T, kw["solver"])
    tol = dict(rtol=1e-4, atol=1e-5)
    assert torch.allclose(u, g["u"], **tol)
    assert torch.allclose(v, g["v"], **tol)
    assert torch.allclose(y, g["y"], **tol)
    gx = emu_bwd(emu, g["x"], g["u0"], g["v0"], g["gy"], T, G, kw["solver"])
    scale = g["gx"].abs().max().item()
    assert (gx - g["gx"]).abs().max().item() <= 1e-4 * scale + 1e-5


@pytest.mark.parametrize("M,N", [(8, 512), (8, 200), (8, 150), (8, 100), (4, 64), (16, 256), (16, 64), (32, 128), (32, 64), (5, 100)])
@pytest.mark.parametrize("solver", ["mu", "hals"])
def test_emul_vs_oracle_shapes(emu, M, N, solver):
    torch.manual_seed(M * 1000 + N)
    for R in (1, 2, 4):
        x = torch.rand(3, M, N)
        x[1, :, : N // 2] = 0
        u0, v0 = torch.rand(M, R), torch.rand(N, R)
        gy = torch.rand_like(x)
        y, u, v = emu_fwd(emu, x, u0, v0, 4, solver)
        yo = O.nmf_forward(x, u0, v0, 4, solver)
        assert torch.allclose(y, yo, rtol=2e-4, atol=1e-5), (R,)
        for G in (4, 2):
            gx = emu_bwd(emu, x, u0, v0, gy, 4, G, solver)
            gxo = O.nmf_backward(x, u0, v0, gy, 4, solver, G)
            # conditioning guard: HALS with R >= M can sit on a ReLU kink where even the
            # oracle in fp32 and fp64 disagree; scale the tolerance by that disagreement
            gx64 = O.nmf_backward(x.double(), u0.double(), v0.double(), gy.double(), 4, solver, G).float()
            kink = (gxo - gx64).abs().max().item()
            s = gxo.abs().max().item()
            assert (gx - gx64).abs().max().item() <= 2e-4 * s + 1e-5 + 30 * kink, (R, G)


# ---- the row-space (Gram) reverse mode of csrc/nmf_gram.h: HALS rank 1 on non-negative matrices -------------------------
def emu_gram_bwd(lib, x, v0, gy, T, G, gscale=1.0):
    M, N = x.shape[-2:]
    xn = np.ascontiguousarray(x.reshape(-1, M, N).numpy())
    gyn = np.ascontiguousarray(gy.reshape(-1, M, N).numpy())
    gx = np.empty_like(xn)
    lib.emu_gram_bwd.argtypes = [ctypes.POINTER(ctypes.c_float)] * 4 + [ctypes.c_int64] + [ctypes.c_int] * 4 + [ctypes.c_float] * 2
    rc = lib.emu_gram_bwd(_p(xn), _p(np.ascontiguousarray(v0.numpy())), _p(gyn), _p(gx), xn.shape[0], M, N, T, G, 1e-16, gscale)
    assert rc == 0
    return torch.from_numpy(gx).reshape(x.shape)


@pytest.mark.parametrize("name", ["cfg2_hals_r1_t5", "hals_r1_t5_g1", "hals_r1_t5", "hals_r1_t10"])
def test_gram_emul_vs_reference_goldens(emu, golden, name):
    """The row-space backward reproduces the input gradients the REFERENCE's autograd gave for its own rank-1 HALS
    iteration (matrix_factorization.py:210-229,506-533) — goldens generated by importing the reference; they include an
    all-zero matrix, where every ε of the iteration matters."""
    g = golden("g2_nmf").case(name)
    kw = NMF_CASES[name]
    assert (g["x"] >= 0).all() and g["u0"].shape[1] == 1
    T = kw["num_iters"]
    G = kw.get("num_grad_steps") or T
    gx = emu_gram_bwd(emu, g["x"], g["v0"], g["gy"], T, G)
    scale = g["gx"].abs().max().item()
    assert (gx - g["gx"]).abs().max().item() <= 1e-4 * scale + 1e-5
    # per matrix, against the float64 oracle: the row-space form is at least as accurate as the reference's own fp32 arithmetic
    x, gy = g["x"].reshape(-1, *g["x"].shape[-2:]), g["gy"].reshape(-1, *g["x"].shape[-2:])
    gx64 = O.nmf_backward(x.double(), g["u0"].double(), g["v0"].double(), gy.double(), T, "hals", G)
    s = gx64.abs().amax(dim=(1, 2)).clamp_min(1e-30)
    e = (gx.reshape(x.shape).double() - gx64).abs().amax(dim=(1, 2)) / s
    assert e.max().item() <= 5e-6, e


@pytest.mark.parametrize("M,N", [(8, 512), (8, 200), (8, 150), (8, 64), (5, 100)])
@pytest.mark.parametrize("T,G", [(5, 5), (4, 3), (4, 1), (1, 1), (10, 10)])
def test_gram_emul_vs_oracle(emu, M, N, T, G):
    """Shapes with masked rows / columns, every relation of graded to total iterations, an all-zero matrix, zero rows and
    zero column blocks, the 1 / windows scale of the fused core — against the float64 oracle, per matrix."""
    torch.manual_seed(M * 1000 + N + T)
    x = torch.rand(5, M, N)
    x[1] = 0
    x[2, :, : N // 2] = 0
    x[3, 2] = 0
    x[4] = x[4] * (torch.rand(M, N) > 0.7)
    u0, v0 = torch.rand(M, 1), torch.rand(N, 1)
    gy = torch.rand_like(x) - 0.3
    gx = emu_gram_bwd(emu, x, v0, gy, T, G, gscale=0.5)
    gx64 = O.nmf_backward(x.double(), u0.double(), v0.double(), 0.5 * gy.double(), T, "hals", G)
    s = gx64.abs().amax(dim=(1, 2)).clamp_min(1e-30)
    e = (gx.double() - gx64).abs().amax(dim=(1, 2)) / s
    assert e.max().item() <= 5e-6, e


def test_emulation_under_asan(tmp_path):
    """The wave programs under AddressSanitizer + UBSan on the host (tests/emul/asan_driver.cpp): GPU ASan is not available on the
    pool, so the shared headers are sanitized where they compile for the CPU — ragged shapes, ranks 1-4, both solvers, the
    decompose-gradient form, an all-zero matrix, the row-space backward; exact-size buffers."""
    exe = str(tmp_path / "emul_asan")
    drv = os.path.join(HERE, "emul", "asan_driver.cpp")
    # (-O0: the sanitized template instantiations take 7 minutes to compile at -O1 -g, 33 s at -O0; the run is 0.2 s)
    r = subprocess.run(["g++", "-O0", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe, drv, SRC],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "0 problem(s)" in r.stdout
