"""-m gpu: the row-space backward of the fused FactMixer core (csrc/nmf_cf_gram.hip, csrc/nmf_gram.h) through the C ABI
(fz_nmf_cf_bwd with relu_gate = 1, HALS rank 1) against the CPU oracle's restatement of the reference chain
SWMatricize.forward -> ReLU'd input -> NMF(rank 1, "hals") -> SWMatricize.inverse_forward
(factorizer.py:41-50; operations.py:417-434; matrix_factorization.py:210-229,506-533) evaluated in FLOAT64, per matrix."""
import pytest
import torch

import factorizer_amd as ft
from factorizer_amd import _native as N
from factorizer_amd import functional as Fn
from oracle import cpu_ref as O
import parity as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_grad(t, u0, v0, ga, shifts, T, G):
    """d/dt of sum(ga * inverse(NMF(matricize(t)))) in float64 with the oracle's hand-derived NMF backward, then the
    ReLU gate [t > 0] (t = relu(z): the gradient with respect to z)."""
    B, C = t.shape[:2]
    S = tuple(t.shape[2:])
    td, gad = t.double(), ga.double()
    x = O.swm_forward(td, head_dim=8, patch_size=8, shifts=shifts)
    # inverse_forward is linear with adjoint = forward / nshift
    gy = O.swm_forward(gad, head_dim=8, patch_size=8, shifts=shifts) / len(shifts)
    gx = O.nmf_backward(x, u0.double(), v0.double(), gy, T, "hals", G)
    gt = O.swm_inverse(gx, C, S, head_dim=8, patch_size=8, shifts=shifts) * len(shifts)   # adjoint of forward: scatter-add
    return gt * (td > 0)


@pytest.mark.parametrize("S,shifts", [((8, 8, 32), [None, 4]), ((16, 8, 64), [None, (4, 4, 4)]), ((8, 16, 32), [None, 2, 4, 6]),
                                      ((8, 8, 24), [None, (4, 0, 2)]), ((8, 8, 32), [None])])
@pytest.mark.parametrize("T,G", [(5, 5), (4, 2), (1, 1)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_row_space_backward_vs_float64_oracle(S, shifts, T, G, dt):
    """W = 32 / 64: four patches per workgroup; W-axis shifts 2, 6: 8-byte halves; W = 24 with a shift of 2: one patch per
    workgroup; bf16 storage: raw register pairs.  An all-zero patch, zero channels, sparse patches."""
    torch.manual_seed(5)
    B, C = 2, 16
    t = torch.relu(torch.randn(B, C, *S))
    t[0, :8, :8, :8, :8] = 0
    t[1, 3] = 0
    m = ft.SWMatricize((None, C, *S), head_dim=8, patch_size=8, shifts=shifts)
    nmf = ft.NMF(size=(8, 512), rank=1, num_iters=T, num_grad_steps=G, init="uniform", solver="hals")
    u0, v0 = nmf.init.u0.clone(), nmf.init.v0.clone()
    assert Fn.nmf_cf_supported(m.geometry, 1, T, G)
    td = t.to(DEV).to(dt)
    t32 = td.float().cpu()
    ga = torch.randn(B, C, *S)
    gad = ga.to(DEV).to(dt)
    tr = td.clone().requires_grad_(True)
    n0 = N.launch_count()
    a = Fn.FactCoreFn.apply(tr, u0.to(DEV), v0.to(DEV), m.geometry, T, G, "hals", 1e-16, True)
    (g,) = torch.autograd.grad(a, tr, gad)
    torch.cuda.synchronize()
    assert N.launch_count() > n0
    sh = [tuple(s) for s in m.geometry.shifts]
    ref = _oracle_grad(t32, u0, v0, gad.float().cpu(), sh, T, G)
    if dt == torch.float32:
        P.close("gt row-space vs float64 oracle", g, ref.float())
    else:
        P.close("gt row-space (bf16 storage) vs float64 oracle", g.float(), ref.float(), rel=len(sh) * 2.0 ** -8,
                why="bf16 storage: one rounding of the stored running sum per window")
    # the general wave program on the same input (relu_gate = 0: no promise about the sign of t), gated by hand
    tr2 = td.clone().requires_grad_(True)
    a2 = Fn.FactCoreFn.apply(tr2, u0.to(DEV), v0.to(DEV), m.geometry, T, G, "hals", 1e-16, False)
    (g2,) = torch.autograd.grad(a2, tr2, gad)
    if dt == torch.float32:
        P.close("gt row-space vs general wave program", g, g2 * (td > 0), rel=1e-5)
    assert torch.equal(a, a2)


def test_row_space_backward_replays_bitwise_at_stage_size():
    torch.manual_seed(6)
    B, C, S = 1, 32, (64, 64, 64)
    m = ft.SWMatricize((None, C, *S), head_dim=8, patch_size=8)
    nmf = ft.NMF(size=(8, 512), rank=1, num_iters=5, init="uniform", solver="hals").to(DEV)
    t = torch.relu(torch.randn(B, C, *S, device=DEV))
    ga = torch.randn(B, C, *S, device=DEV)
    outs = []
    for _ in range(3):
        tr = t.clone().requires_grad_(True)
        a = Fn.FactCoreFn.apply(tr, nmf.init.u0, nmf.init.v0, m.geometry, 5, 5, "hals", 1e-16, True)
        (g,) = torch.autograd.grad(a, tr, ga)
        outs.append(g)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.isfinite(outs[0]).all()
