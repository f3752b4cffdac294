"""-m gpu: BOTH shift windows of the fused FactMixer core in ONE slab-major launch (csrc/nmf_cf.hip: fz_nmf_cf_fwd2 /
fz_nmf_cf_bwd2) against the one-launch-per-window entry points it replaces (fz_nmf_cf_fwd / fz_nmf_cf_bwd — themselves
pinned to the reference's goldens and the oracle in tests/test_gpu_parity.py).  The reference's semantics:
SWMatricize.forward -> NMF -> SWMatricize.inverse_forward, ((0 + z0) + z1) / 2 (operations.py:417-434).

The two-window launch contains an in-launch producer / consumer hand-off (window 0's tiles -> window 1's read-modify-write),
so beyond equal values the tests check that the result does not depend on the schedule: every slices-per-group / lag /
workgroup-count setting, fewer workgroups than a bundle (heavy queueing behind the ticket counter) and far more than are
resident, repeated launches, and the time-out word of the bounded spin staying zero."""
import ctypes

import pytest
import torch

from factorizer_amd import _native as N
from factorizer_amd import functional as Fn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _arrs(shifts):
    return [(N._i * 3)(*s) for s in shifts], (N._i * 6)(*shifts[0], *shifts[1])


def _two_launches_fwd(t, u0, v0, B, C, S, shifts, solver):
    out = torch.empty_like(t)
    one, _ = _arrs(shifts)
    for w in range(2):
        N.check(N.lib().fz_nmf_cf_fwd(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), out.data_ptr(), B, C, *S, one[w], int(w > 0),
                                      2 if w == 1 else 1, 1, 5, N.SOLVER_ID[solver], 1e-16, N.act_dtype(t), N.stream_ptr(t)), "fwd")
    return out


def _two_launches_bwd(t, u0, v0, ga, B, C, S, shifts, solver, gate):
    gt = torch.empty_like(t)
    one, _ = _arrs(shifts)
    for w in range(2):
        N.check(N.lib().fz_nmf_cf_bwd(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), ga.data_ptr(), gt.data_ptr(), B, C, *S, one[w],
                                      int(w > 0), 2, int(gate), 1, 5, 5, N.SOLVER_ID[solver], 1e-16, N.act_dtype(t),
                                      N.stream_ptr(t)), "bwd")
    return gt


def _ws(B, C, D):
    n = int(N.lib().fz_nmf_cf2_workspace_bytes(B, C, D))
    assert n > 0 and n % 16 == 0
    return torch.full((n // 4,), 0x5a5a5a5a, dtype=torch.int32, device=DEV)   # poisoned: the call must zero it itself


def _one_launch(t, u0, v0, ga, B, C, S, shifts, solver, gate, tune):
    _, both = _arrs(shifts)
    tn = (N._i * 3)(*tune) if tune is not None else None
    out, gt = torch.empty_like(t), torch.empty_like(t)
    ws1, ws2 = _ws(B, C, S[0]), _ws(B, C, S[0])
    N.check(N.lib().fz_nmf_cf_fwd2(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), out.data_ptr(), B, C, *S, both, 1, 5,
                                   N.SOLVER_ID[solver], 1e-16, N.act_dtype(t), ws1.data_ptr(), tn, N.stream_ptr(t)), "fwd2")
    N.check(N.lib().fz_nmf_cf_bwd2(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), ga.data_ptr(), gt.data_ptr(), B, C, *S, both,
                                   int(gate), 1, 5, 5, N.SOLVER_ID[solver], 1e-16, N.act_dtype(t), ws2.data_ptr(), tn,
                                   N.stream_ptr(t)), "bwd2")
    torch.cuda.synchronize()
    assert int(ws1[1]) == 0 and int(ws2[1]) == 0, "a bounded spin of the hand-off gave up"
    assert int(ws1[0]) >= 0
    return out, gt


def _inputs(B, C, S, dt, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = torch.rand(B, C, *S, generator=g).mul_(torch.rand(B, C, *S, generator=g) > 0.2)   # ReLU output: exact zeros inside
    ga = torch.randn(B, C, *S, generator=g)
    u0, v0 = torch.rand(8, 1, generator=g), torch.rand(512, 1, generator=g)
    return t.to(DEV).to(dt), ga.to(DEV).to(dt), u0.to(DEV), v0.to(DEV)


CASES = [
    # B, C, spatial, window-1 shift
    (2, 16, (64, 64, 64), (4, 4, 4)),        # the default half-patch shift
    (1, 16, (32, 64, 64), (4, 4, 4)),        # fewer slices than XCDs
    (3, 8, (16, 64, 128), (2, 6, 8)),        # G0 = 2 (planes 0 and 1 need each other cyclically), odd slice count
    (1, 8, (64, 16, 64), (7, 2, 60)),        # D shift 7, W shift wraps
]


@pytest.mark.parametrize("B,C,S,s1", CASES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("solver", ["hals", "mu"])
def test_one_launch_equals_two_launches(B, C, S, s1, dt, solver):
    shifts = [(0, 0, 0), s1]
    _, both = _arrs(shifts)
    assert N.lib().fz_nmf_cf2_supported(B, C, *S, both, 2, 1, 5, 5, N.STORE_BF16 if dt == torch.bfloat16 else N.STORE_F32)
    t, ga, u0, v0 = _inputs(B, C, S, dt)
    ref_o = _two_launches_fwd(t, u0, v0, B, C, S, shifts, solver)
    ref_g = _two_launches_bwd(t, u0, v0, ga, B, C, S, shifts, solver, True)
    nslice = B * C // 8
    tunes = [None, (1, 0, 0), (nslice, 0, 0), (nslice, 1, 0), (1, S[0] // 8 - 1, 0), (nslice, 0, 3), (1, 0, 7), (nslice, 0, 4096)]
    for tune in tunes:
        out, gt = _one_launch(t, u0, v0, ga, B, C, S, shifts, solver, True, tune)
        assert torch.equal(out, ref_o), ("forward", tune, float((out.float() - ref_o.float()).abs().max()))
        assert torch.equal(gt, ref_g), ("backward", tune, float((gt.float() - ref_g.float()).abs().max()))


def test_stage0_size_replays_and_module_path_uses_it(monkeypatch):
    """README stage 0 (B = 2, C = 32, 128^3): 10 launches bit for bit equal to the two-launch result, and FactCoreFn — the
    module path — takes the one-launch form for this geometry when it is switched on (FZ_CF2=1; off by default: measured
    slower, profiles/r04_cf2_sweep.json)."""
    B, C, S, shifts = 2, 32, (128, 128, 128), [(0, 0, 0), (4, 4, 4)]
    t, ga, u0, v0 = _inputs(B, C, S, torch.float32, seed=1)
    ref_o = _two_launches_fwd(t, u0, v0, B, C, S, shifts, "hals")
    ref_g = _two_launches_bwd(t, u0, v0, ga, B, C, S, shifts, "hals", True)
    for rep in range(10):
        out, gt = _one_launch(t, u0, v0, ga, B, C, S, shifts, "hals", True, None)
        assert torch.equal(out, ref_o) and torch.equal(gt, ref_g), rep
    geo = Fn.Geometry(C, S, 8, (8, 8, 8), shifts)
    monkeypatch.setattr(Fn, "_CF2", True)
    assert Fn.nmf_cf2_plan(geo, B, 1, 5, 5, N.act_dtype(t)) is not None
    tt = t.clone().requires_grad_(True)
    n0 = N.launch_count()
    a = Fn.FactCoreFn.apply(tt, u0, v0, geo, 5, 5, "hals", 1e-16, True)
    (g,) = torch.autograd.grad(a, tt, ga)
    assert N.launch_count() - n0 == 2, "one launch forward, one backward"
    assert torch.equal(a, ref_o) and torch.equal(g, ref_g)


def test_unsupported_geometries_are_refused_not_miscomputed():
    lib = N.lib()
    ok = lambda B, C, S, sh, R=1: lib.fz_nmf_cf2_supported(B, C, *S, (N._i * 6)(*sh[0], *sh[1]), 2, R, 5, 5, 0)  # noqa: E731
    assert ok(2, 32, (128, 128, 128), [(0, 0, 0), (4, 4, 4)])
    assert not ok(2, 32, (128, 128, 128), [(0, 0, 0), (4, 4, 4)], R=2)          # rank 2: one launch per window
    assert not ok(2, 32, (128, 128, 128), [(4, 4, 4), (0, 0, 0)])               # window 0 must be the unshifted one
    assert not ok(2, 32, (128, 128, 128), [(0, 0, 0), (0, 4, 4)])               # no D shift: a different dependency
    assert not ok(2, 32, (128, 128, 128), [(0, 0, 0), (8, 4, 4)])
    assert not ok(2, 32, (128, 128, 128), [(0, 0, 0), (4, 4, 2)])               # W shift 2 (mod 4): half-chunk kernels only
    assert not ok(2, 32, (32, 32, 32), [(0, 0, 0), (4, 4, 4)])                  # W / 8 = 4: no 8-patch tile
    t = torch.zeros(1, 8, 64, 64, 64, device=DEV)
    u0, v0 = torch.rand(8, 1, device=DEV), torch.rand(512, 1, device=DEV)
    ws = _ws(1, 8, 64)
    rc = lib.fz_nmf_cf_fwd2(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), t.data_ptr(), 1, 8, 64, 64, 64,
                            (N._i * 6)(0, 0, 0, 0, 4, 4), 1, 5, 1, 1e-16, 0, ws.data_ptr(), None, N.stream_ptr(t))
    assert rc != 0 and b"two-window" in lib.fz_last_error_string()
