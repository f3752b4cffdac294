"""-m gpu: the gfx950 kernels (through the C ABI) against the reference goldens and the CPU
oracle.  Bit-exact for SWMatricize (pure data movement); 1e-4 (fp32) for NMF / blocks."""
import hashlib

import numpy as np
import pytest
import torch
from torch import nn

import factorizer_amd as ft
from factorizer_amd import _native
from oracle import cpu_ref as O
import parity as P
from test_oracle_golden import G1_CASES, NMF_CASES

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Launches:
    """Asserts that the native library actually launched kernels inside the block."""

    def __enter__(self):
        self.n0 = _native.launch_count()
        return self

    def __exit__(self, *a):
        torch.cuda.synchronize()
        assert _native.launch_count() > self.n0, "native kernels were not launched"


# ---------------------------------------------------------------- SWMatricize ---------
@pytest.mark.parametrize("name", sorted(G1_CASES))
def test_swm_goldens_bit_exact(golden, name):
    g = golden("g1_swmatricize").case(name)
    shape, kw = G1_CASES[name]
    x = torch.arange(int(np.prod(shape)), dtype=torch.float32).reshape(shape)
    m = ft.SWMatricize((None, *shape[1:]), **kw)
    with Launches():
        y = m(x.to(DEV))
        z = m.inverse_forward(y)
    assert torch.equal(y.cpu().to(torch.int32), g["y"])
    nw = len(kw.get("shifts", [0, 1]))
    if nw in (1, 2, 4):
        assert torch.equal(z.cpu(), g["z"])
    else:
        assert torch.allclose(z.cpu(), g["z"], rtol=2e-7, atol=0)
    if "yr" in g:
        zr = m.inverse_forward(g["yr"].to(DEV)).cpu()
        if nw in (1, 2, 4):
            assert torch.equal(zr, g["zr"])
        else:
            assert torch.allclose(zr, g["zr"], rtol=3e-7, atol=1e-7)


def test_swm_cfg2_sha256_and_roundtrip(golden):
    """BASELINE cfg 2 size: (1,32,128^3) -> (8,4096,8,512); hash of the reference's output."""
    g = golden("g1_swmatricize")
    torch.manual_seed(0)
    x = torch.rand(1, 32, 128, 128, 128)
    m = ft.SWMatricize((None, 32, 128, 128, 128), head_dim=8, patch_size=8)
    xd = x.to(DEV)
    with Launches():
        y = m(xd)
    assert list(y.shape) == g["cfg2:shape_y"].tolist()
    digest = hashlib.sha256(y.cpu().numpy().tobytes()).digest()
    assert np.frombuffer(digest, dtype=np.uint8).tolist() == g["cfg2:sha256_y"].tolist()
    z = m.inverse_forward(y)
    assert torch.equal(z, xd)  # encode -> decode round trip, size-independent property


SWM_RANDOM = [
    ((2, 16, 8, 16, 32), dict(head_dim=8, patch_size=(2, 4, 8))),
    ((1, 8, 12, 10, 6), dict(head_dim=4, patch_size=(3, 5, 2))),               # non power-of-two, VE=2
    ((2, 8, 8, 8, 8), dict(num_heads=2, patch_size=4, shifts=[None, 1, 2, 3])),  # odd shifts, VE=1
    ((1, 16, 16, 16, 16), dict(head_dim=8, patch_size=8, shifts=[None, 2, 4, 6])),  # production windows
    ((3, 8, 4, 4, 4), dict(head_dim=8, grid_size=1)),                           # global patch
    ((2, 8, 16, 16), dict(head_dim=4, patch_size=4)),                           # 2-D
    ((1, 8, 10, 12, 10), dict(head_dim=8, patch_size=(5, 6, 5))),               # cfg-5 bottleneck shape
]


@pytest.mark.parametrize("shape,kw", SWM_RANDOM)
def test_swm_vs_oracle_random(shape, kw):
    torch.manual_seed(1)
    x = torch.randn(shape)
    m = ft.SWMatricize((None, *shape[1:]), **kw)
    spatial = shape[2:]
    okw = dict(kw)
    if len(spatial) == 3:
        yo = O.swm_forward(x, **okw)
    else:  # oracle is 3-D: lift with a unit leading axis
        yo = None
    xd = x.to(DEV).requires_grad_(True)
    with Launches():
        y = m(xd)
    ycpu = ft.SWMatricize((None, *shape[1:]), **kw)(x)  # composed path, bit-exact too
    assert torch.equal(y.cpu(), ycpu)
    if yo is not None:
        assert torch.equal(y.cpu(), yo)
    yr = torch.randn_like(ycpu)
    z = m.inverse_forward(yr.to(DEV))
    zc = m.inverse_forward(yr)
    nw = m.geometry.nshift
    if nw in (1, 2, 4):
        assert torch.equal(z.cpu(), zc)
    else:
        assert torch.allclose(z.cpu(), zc, rtol=3e-7, atol=1e-7)
    # backward of forward = un-averaged inverse; backward of inverse = forward / nshift
    gy = torch.randn_like(ycpu)
    (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
    xc = x.clone().requires_grad_(True)
    (gxc,) = torch.autograd.grad(m(xc), xc, gy)
    assert torch.allclose(gx.cpu(), gxc, rtol=1e-6, atol=1e-6)
    yd = yr.to(DEV).requires_grad_(True)
    gz = torch.randn(shape)
    (gyd,) = torch.autograd.grad(m.inverse_forward(yd), yd, gz.to(DEV))
    ycl = yr.clone().requires_grad_(True)
    (gyc,) = torch.autograd.grad(m.inverse_forward(ycl), ycl, gz)
    assert torch.allclose(gyd.cpu(), gyc, rtol=1e-6, atol=1e-7)


def test_swm_bf16_moves_bits():
    torch.manual_seed(2)
    x = torch.randn(2, 16, 8, 8, 16).to(torch.bfloat16)
    m = ft.SWMatricize((None, 16, 8, 8, 16), head_dim=8, patch_size=(4, 4, 8))
    y = m(x.to(DEV))
    assert y.dtype == torch.bfloat16
    assert torch.equal(y.cpu(), m(x))


# ---------------------------------------------------------------- NMF -------------------
@pytest.mark.parametrize("name", sorted(NMF_CASES))
def test_nmf_goldens(golden, name):
    g = golden("g2_nmf").case(name)
    kw = dict(NMF_CASES[name])
    M, N = g["x"].shape[-2:]
    R = g["u0"].shape[1]
    nmf = ft.NMF(size=(M, N), rank=R, init="uniform", **kw)
    nmf.load_state_dict({"init.u0": g["u0"], "init.v0": g["v0"]})
    nmf = nmf.to(DEV)
    x = g["x"].to(DEV).requires_grad_(True)
    with Launches():
        u, v = nmf.decompose(x)
        y = nmf(x)
        (gx,) = torch.autograd.grad(y, x, g["gy"].to(DEV))
    P.close("u", u, g["u"])
    P.close("v", v, g["v"])
    P.close("y", y, g["y"])
    P.close("gx", gx, g["gx"])
    assert (u >= 0).all() and (v >= 0).all()


@pytest.mark.parametrize("solver", ["mu", "hals"])
@pytest.mark.parametrize("R", [1, 2, 3, 4])
def test_nmf_8x512_vs_oracle(solver, R):
    torch.manual_seed(10 + R)
    x = torch.rand(37, 3, 8, 512)           # ragged count: not a multiple of waves per block
    x[0, 0].zero_()                          # all-zero matrix (eps path)
    x[1, 1, :, :300] = 0
    u0, v0 = torch.rand(8, R), torch.rand(512, R)
    gy = torch.rand_like(x)
    nmf = ft.NMF(size=(8, 512), rank=R, num_iters=5, init="uniform", solver=solver)
    nmf.load_state_dict({"init.u0": u0, "init.v0": v0})
    nmf = nmf.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    with Launches():
        y = nmf(xd)
        (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
    yo = O.nmf_forward(x, u0, v0, 5, solver)
    gxo = O.nmf_backward(x, u0, v0, gy, 5, solver)
    gx64 = O.nmf_backward(x.double(), u0.double(), v0.double(), gy.double(), 5, solver).float()
    P.close("y", y, yo)
    kink = (gxo - gx64).abs().amax(dim=(-1, -2))          # fp32-vs-fp64 disagreement of the oracle
    # HALS gates its gradient by [w > 0]: matrices with a ReLU pre-activation within fp32
    # rounding of 0 have no well-defined fp32 gradient (rounding order flips the gate)
    well = torch.ones_like(kink, dtype=torch.bool)
    if solver == "hals":
        well = O.hals_gate_margin(x, u0, v0, 5) > 2e-6
        excluded = 1.0 - well.float().mean().item()
        P.note("hals_gate_excluded_fraction", value=excluded, R=R, matrices=well.numel())
        if R <= 2:  # the BASELINE ranks (cfg 2-4: R = 1, cfg 5: R = 2)
            assert excluded < 0.10, f"{excluded:.3f} of the matrices sit on a ReLU kink"
        assert excluded < 0.15
    gdev = gx.cpu()
    # per-matrix comparison against the fp64 oracle, each matrix normalised by its own gradient scale (an
    # all-zero matrix divides by eps = 1e-16: its gradient is ~1e7 times larger than its neighbours' and
    # would otherwise set the scale for all of them)
    err = (gdev - gx64).abs().amax(dim=(-1, -2))
    scale = gx64.abs().amax(dim=(-1, -2)) + 1e-30
    # a matrix is well-posed in fp32 when the fp32 ORACLE itself is within 5e-5 of fp64 (it depends on the
    # host's BLAS: the fraction is reported, and bounded for the BASELINE ranks): where it is not
    # (eps-dominated ratios, near-kink gates), no fp32 evaluation order can be asked to hit 1e-4 — those are
    # held to twice the oracle's own fp32 error instead
    posed = well & (kink <= 5e-5 * scale)
    frac_ill = 1.0 - posed.float().mean().item()
    P.note("nmf_grad_matrices_ill_posed_in_fp32", solver=solver, R=R, fraction=frac_ill,
           worst_err_over_oracle_fp32_err=float((err[~posed & well] / (kink[~posed & well] + 1e-30)).max())
           if (~posed & well).any() else 0.0)
    assert frac_ill < (0.10 if R <= 2 else 0.30)
    w = posed.reshape(*posed.shape, 1, 1)
    sc = scale.reshape(*scale.shape, 1, 1)
    P.close("gx / per-matrix max|gx| (matrices well-posed in fp32, vs fp64 oracle)",
            torch.where(w, gdev / sc, torch.zeros_like(gdev)), torch.where(w, gx64 / sc, torch.zeros_like(gx64)),
            floor=1e-7)
    ill = ~posed & well
    assert (err[ill] <= 1e-4 * scale[ill] + 2.0 * kink[ill]).all()


@pytest.mark.parametrize("M,N", [(8, 150), (4, 64), (16, 256), (16, 64), (32, 128), (5, 100), (8, 64)])
def test_nmf_masked_families_vs_oracle(M, N):
    torch.manual_seed(M * 1000 + N)
    for solver in ("mu", "hals"):
        for R in (1, 2):
            x = torch.rand(11, M, N)
            u0, v0 = torch.rand(M, R), torch.rand(N, R)
            gy = torch.rand_like(x)
            nmf = ft.NMF(size=(M, N), rank=R, num_iters=4, num_grad_steps=3, init="uniform", solver=solver)
            nmf.load_state_dict({"init.u0": u0, "init.v0": v0})
            nmf = nmf.to(DEV)
            xd = x.to(DEV).requires_grad_(True)
            with Launches():
                y = nmf(xd)
                (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
            yo = O.nmf_forward(x, u0, v0, 4, solver)
            gx64 = O.nmf_backward(x.double(), u0.double(), v0.double(), gy.double(), 4, solver, 3).float()
            gxo = O.nmf_backward(x, u0, v0, gy, 4, solver, 3)
            P.close(f"y {solver} R{R}", y, yo)
            kink = (gxo - gx64).abs().max().item()   # the fp32 oracle's own distance to fp64
            P.close(f"gx {solver} R{R} (vs fp64 oracle)", gx, gx64, extra=kink)


def test_nmf_decompose_backward():
    torch.manual_seed(5)
    x = torch.rand(9, 8, 512)
    for solver in ("mu", "hals"):
        nmf = ft.NMF(size=(8, 512), rank=2, num_iters=3, init="uniform", solver=solver)
        xc = x.clone().requires_grad_(True)
        u, v = nmf.decompose(xc)
        gu, gv = torch.rand_like(u), torch.rand_like(v)
        (gxc,) = torch.autograd.grad([u, v], xc, [gu, gv])
        nd = nmf.to(DEV)
        xd = x.to(DEV).requires_grad_(True)
        with Launches():
            ud, vd = nd.decompose(xd)
            (gxd,) = torch.autograd.grad([ud, vd], xd, [gu.to(DEV), gv.to(DEV)])
        P.close(f"u {solver}", ud, u)
        P.close(f"v {solver}", vd, v)
        P.close(f"gx {solver}", gxd, gxc)


def test_nmf_full_size_properties():
    """Stage-0 batch of BASELINE cfg 3/4 (65 536 matrices of 8x512): size-independent
    properties — batch-position independence (bit-exact), non-negativity, exact recovery of
    rank-1 non-negative matrices, linearity of the backward in gy."""
    torch.manual_seed(0)
    nmat = 65536
    nmf = ft.NMF(size=(8, 512), rank=1, num_iters=5, init="uniform", solver="hals").to(DEV)
    x = torch.rand(nmat, 8, 512, device=DEV)
    with Launches():
        y = nmf(x)
    assert torch.isfinite(y).all() and (y >= 0).all()
    perm = torch.randperm(nmat, device=DEV)
    assert torch.equal(nmf(x[perm]), y[perm])
    a = torch.rand(1024, 8, 1, device=DEV) + 0.1
    b = torch.rand(1024, 1, 512, device=DEV) + 0.1
    r1 = a * b
    P.close("rank-1 recovery", nmf(r1), r1)
    xs = x[:4096].clone().requires_grad_(True)
    ys = nmf(xs)
    g1, g2 = torch.rand_like(ys), torch.rand_like(ys)
    (ga,) = torch.autograd.grad(ys, xs, g1, retain_graph=True)
    (gb,) = torch.autograd.grad(ys, xs, g2, retain_graph=True)
    (gab,) = torch.autograd.grad(ys, xs, g1 + 2 * g2)
    P.close("backward linearity", gab, ga + 2 * gb)


def test_nmf_unsupported_shape_uses_composed_path_with_warning():
    # rank 5 is outside both native families (wave-resident and split-N: R <= 4)
    nmf = ft.NMF(size=(8, 1200), rank=5, num_iters=2, init="uniform", solver="hals").to(DEV)
    x = torch.rand(3, 8, 1200, device=DEV)
    with pytest.warns(RuntimeWarning):
        y = nmf(x)
    yo = O.nmf_forward(x.cpu(), nmf.init.u0.cpu(), nmf.init.v0.cpu(), 2, "hals")
    P.close("y (composed path)", y, yo)


# ---------------------------------------------------------------- blocks / model ---------
BLOCK_KW = {"hals_r1": dict(rank=1, num_iters=5, solver="hals"), "mu_r2": dict(rank=2, num_iters=3, solver="mu")}


@pytest.mark.parametrize("name", sorted(BLOCK_KW))
def test_block_goldens(golden, name):
    g = golden("g5_block").case(name)
    blk = ft.FactorizerBlock(channels=16, spatial_size=(8, 8, 8), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}), act=nn.ReLU,
                             factorize=ft.NMF, init="uniform", mlp_ratio=2, dropout=0.0, **BLOCK_KW[name])
    blk.load_state_dict({k[3:]: v for k, v in g.items() if k.startswith("sd:")})
    blk = blk.to(DEV)
    x = g["x"].to(DEV).requires_grad_(True)
    with Launches():
        y = blk(x)
        names = [k for k, _ in blk.named_parameters()]
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), g["gy"].to(DEV))
    P.close("y", y, g["y"])
    P.close("gx", grads[0], g["gx"])
    for k, gr in zip(names, grads[1:]):
        P.close("grad:" + k, gr, g["grad:" + k])


def test_model_goldens(golden):
    from test_modules_cpu import _tiny_model
    g = golden("g6_model")
    model = _tiny_model().eval()
    model.load_state_dict(g.case("sd"))
    model = model.to(DEV)
    x = g["x"].to(DEV).requires_grad_(True)
    with Launches():
        y = model(x)
        names = [k for k, _ in model.named_parameters()]
        grads = torch.autograd.grad(y, [x] + list(model.parameters()), g["gy"].to(DEV))
    P.close("y", y, g["y"])
    P.close("gx", grads[0], g["gx"])
    for k, gr in zip(names, grads[1:]):
        P.close("grad:" + k, gr, g["grad:" + k])


def test_block_cfg2_shape_vs_oracle_reduced():
    """BASELINE cfg 2 block (C=32, d=8, p=8, HALS R=1 T=5) on a 32^3 volume vs the oracle."""
    torch.manual_seed(0)
    blk = ft.FactorizerBlock(channels=32, spatial_size=(32, 32, 32), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                             factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals",
                             mlp_ratio=2, dropout=0.0)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    x = torch.rand(1, 32, 32, 32, 32)
    gy = torch.rand_like(x)
    xo = x.clone().requires_grad_(True)
    cfg = dict(reshape=dict(head_dim=8, patch_size=8), num_iters=5, solver="hals")
    yo = O.factorizer_block(xo, sd, "", cfg)
    (gxo,) = torch.autograd.grad(yo, xo, gy)
    blk = blk.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    with Launches():
        yd = blk(xd)
        (gxd,) = torch.autograd.grad(yd, xd, gy.to(DEV))
    P.close("y", yd, yo)
    P.close("gx", gxd, gxo)


def test_block_cfg2_full_size_runs():
    """BASELINE cfg 2 at full size (1,32,128^3): finite outputs and gradients."""
    torch.manual_seed(0)
    blk = ft.FactorizerBlock(channels=32, spatial_size=(128, 128, 128), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                             factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals",
                             mlp_ratio=2, dropout=0.0).to(DEV)
    x = torch.rand(1, 32, 128, 128, 128, device=DEV, requires_grad=True)
    with Launches():
        y = blk(x)
        (gx,) = torch.autograd.grad(y, x, torch.rand_like(y))
    assert y.shape == x.shape and torch.isfinite(y).all() and torch.isfinite(gx).all()


# ---------------------------------------------------------------- fused FactMixer core -----
@pytest.mark.parametrize("S", [(16, 8, 24), (8, 16, 64), (8, 8, 32)])
@pytest.mark.parametrize("shifts", [None, [None, 4, (4, 0, 4), (0, 4, 0)], [None], [None, 2, 4, 6], [(2, 6, 2), (5, 3, 6)]])
@pytest.mark.parametrize("solver,R", [("hals", 1), ("mu", 2), ("hals", 2)])
def test_fact_core_fused_vs_modular(S, shifts, solver, R):
    """csrc/nmf_cf.hip (gather → NMF → scatter/average per window) against the modular chain
    SWMatricize → NMF → inverse (itself checked against the reference goldens).  W = 24: direct
    gather kernels (3 patches along W); W = 64 / 32: line-coalesced kernels with 8 / 4 patches per
    workgroup; W-axis shifts 2 and 6 (the BraTS bundle's windows, train.yaml:50-54) move as 8-byte
    halves and, at W = 32, 24 use the 4- / 1-patch variants."""
    from factorizer_amd import functional as Fn
    torch.manual_seed(11)
    C = 16
    m = ft.SWMatricize((None, C, *S), head_dim=8, patch_size=8, shifts=shifts)
    nmf = ft.NMF(size=(8, 512), rank=R, num_iters=4, num_grad_steps=3, init="uniform", solver=solver).to(DEV)
    t = torch.rand(2, C, *S, device=DEV)
    t[0, :8, :8, :8, :8] = 0
    G = 3
    assert Fn.nmf_cf_supported(m.geometry, R, 4, G)
    t1 = t.clone().requires_grad_(True)
    t2 = t.clone().requires_grad_(True)
    with Launches():
        a1 = Fn.FactCoreFn.apply(t1, nmf.init.u0, nmf.init.v0, m.geometry, 4, G, solver, 1e-16, False)
    a2 = m.inverse_forward(nmf(m(t2)))
    P.close("a fused vs modular", a1, a2, rel=1e-5)
    ga = torch.rand_like(a1)
    (g1,) = torch.autograd.grad(a1, t1, ga)
    (g2,) = torch.autograd.grad(a2, t2, ga)
    P.close("gt fused vs modular", g1, g2)
    # relu_gate = True gates the gradient by [t > 0]
    t3 = t.clone().requires_grad_(True)
    a3 = Fn.FactCoreFn.apply(t3, nmf.init.u0, nmf.init.v0, m.geometry, 4, G, solver, 1e-16, True)
    (g3,) = torch.autograd.grad(a3, t3, ga)
    if solver == "hals" and R == 1:
        # relu_gate = True is the promise t >= 0: HALS rank 1 then runs its backward in the row space (csrc/nmf_gram.h) — the
        # same function evaluated in another order (tests/test_gpu_gram.py holds it to the float64 oracle)
        P.close("gt row-space vs general wave program", g3, g1 * (t > 0), rel=1e-5)
    else:
        assert torch.allclose(g3, g1 * (t > 0), rtol=0, atol=0)


@pytest.mark.parametrize("S,patch", [((10, 12, 20), (5, 6, 5)), ((8, 8, 16), (4, 4, 4)), ((6, 10, 14), (3, 5, 7)),
                                     ((4, 8, 16), (2, 4, 8)), ((8, 8, 8), (8, 8, 4))])
@pytest.mark.parametrize("shifts", [None, [None, (1, 2, 3), (2, 0, 1)], [None]])
@pytest.mark.parametrize("solver,R", [("hals", 1), ("mu", 2), ("hals", 2)])
def test_fact_core_any_patch_vs_modular(S, patch, shifts, solver, R):
    """csrc/nmf_pcf.hip — the fused gather → NMF → scatter/average core for ANY patch of <= 256 voxels (patch (5,6,5) of
    BASELINE configs[4], p = 4, anisotropic patches, 1 - 4 columns per lane) and any shift parity (odd W-axis shifts
    are outside the 8x8x8 kernels) — against the modular chain SWMatricize → NMF → inverse, values and gradients,
    with and without the ReLU gate."""
    from factorizer_amd import functional as Fn
    torch.manual_seed(13)
    C = 16
    m = ft.SWMatricize((None, C, *S), head_dim=8, patch_size=patch, shifts=shifts)
    geo = m.geometry
    nmf = ft.NMF(size=(8, geo.P), rank=R, num_iters=4, num_grad_steps=3, init="uniform", solver=solver).to(DEV)
    t = torch.rand(2, C, *S, device=DEV)
    t[0, :8, :patch[0], :patch[1], :patch[2]] = 0      # an all-zero matrix in window 0
    G = 3
    assert Fn.nmf_pcf_supported(geo, R, 4, G) and not Fn.nmf_cf_supported(geo, R, 4, G)
    t1 = t.clone().requires_grad_(True)
    t2 = t.clone().requires_grad_(True)
    with Launches():
        a1 = Fn.FactCoreFn.apply(t1, nmf.init.u0, nmf.init.v0, geo, 4, G, solver, 1e-16, False)
    a2 = m.inverse_forward(nmf(m(t2)))
    P.close("a fused vs modular", a1, a2, rel=1e-5)
    ga = torch.rand_like(a1)
    (g1,) = torch.autograd.grad(a1, t1, ga)
    (g2,) = torch.autograd.grad(a2, t2, ga)
    P.close("gt fused vs modular", g1, g2)
    t3 = t.clone().requires_grad_(True)
    a3 = Fn.FactCoreFn.apply(t3, nmf.init.u0, nmf.init.v0, geo, 4, G, solver, 1e-16, True)
    (g3,) = torch.autograd.grad(a3, t3, ga)
    assert torch.allclose(g3, g1 * (t > 0), rtol=0, atol=0)


@pytest.mark.parametrize("solver,R", [("hals", 1), ("mu", 2), ("hals", 2)])
@pytest.mark.parametrize("S,B,C", [((15, 6, 5), 1, 8), ((5, 18, 25), 1, 8), ((10, 12, 10), 3, 24)])
def test_fact_core_two_matrices_per_wave(S, B, C, solver, R):
    """The 8 x 150 forward of nmf_pcf.hip runs TWO matrices per wave (32 lanes x 5 columns each; BASELINE configs[4], patch
    (5, 6, 5)).  Matrix counts 3 and 15 (odd: the last wave's upper half is idle and must store nothing) and 72, two windows,
    against the modular chain SWMatricize -> NMF -> inverse AND against the CPU oracle's NMF on the device-matricized input."""
    from factorizer_amd import functional as Fn
    torch.manual_seed(29)
    m = ft.SWMatricize((None, C, *S), head_dim=8, patch_size=(5, 6, 5), shifts=[None, (2, 3, 1)])
    geo = m.geometry
    assert geo.P == 150
    nmf = ft.NMF(size=(8, geo.P), rank=R, num_iters=6, num_grad_steps=2, init="uniform", solver=solver).to(DEV)
    t = torch.rand(B, C, *S, device=DEV)
    t1 = t.clone().requires_grad_(True)
    t2 = t.clone().requires_grad_(True)
    sentinel = torch.full((4096,), 7.0, device=DEV)        # allocated right behind: a stray store of the idle half would show
    with Launches():
        a1 = Fn.FactCoreFn.apply(t1, nmf.init.u0, nmf.init.v0, geo, 6, 2, solver, 1e-16, False)
    a2 = m.inverse_forward(nmf(m(t2)))
    P.close("a two-per-wave vs modular", a1, a2, rel=1e-5)
    assert torch.equal(sentinel, torch.full_like(sentinel, 7.0))
    xm = m(t).cpu()                                          # (W·B·heads·patches, 8, 150)
    yo = O.nmf_forward(xm, nmf.init.u0.cpu(), nmf.init.v0.cpu(), 6, solver, None)
    P.close("a two-per-wave vs oracle", a1, m.inverse_forward(yo.to(DEV)), rel=1e-4)
    ga = torch.rand_like(a1)
    (g1,) = torch.autograd.grad(a1, t1, ga)
    (g2,) = torch.autograd.grad(a2, t2, ga)
    P.close("gt two-per-wave vs modular", g1, g2)
    b1 = Fn.FactCoreFn.apply(t.clone(), nmf.init.u0, nmf.init.v0, geo, 6, 2, solver, 1e-16, False)
    assert torch.equal(a1, b1)                               # replay


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_act_add_and_separate_window_backward(dt):
    """fz_act_add (dst += src, 16 bytes per lane, ragged tail) against torch, and the backward of a two-window 8 x 150 core —
    window 1 into its own buffer + fz_act_add where fz_nmf_pcf_bwd_prefers_separate() says so — against the modular chain."""
    from factorizer_amd import functional as Fn
    from factorizer_amd import _native as N
    torch.manual_seed(5)
    for n in (8 * 1000 + 3, 16, 5):
        a = torch.randn(n + 8, device=DEV).to(dt)[:n]        # (a view at offset 0 of a 16-byte aligned allocation)
        b = torch.randn(n + 8, device=DEV).to(dt)[:n]
        ref = (a.float() + b.float()).to(dt)
        N.check(N.lib().fz_act_add(a.data_ptr(), b.data_ptr(), n, N.act_dtype(a), N.stream_ptr(a)), "fz_act_add")
        assert torch.equal(a, ref), n
    bad = torch.zeros(64, device=DEV).to(dt)
    assert N.lib().fz_act_add(bad.data_ptr() + bad.element_size(), bad.data_ptr(), 8, N.act_dtype(bad), N.stream_ptr(bad)) != 0
    assert N.lib().fz_nmf_pcf_bwd_prefers_separate(5, 6, 5, N.STORE_BF16) == 1 and N.lib().fz_nmf_pcf_bwd_prefers_separate(4, 4, 4, N.STORE_BF16) == 0
    assert N.lib().fz_nmf_pcf_bwd_prefers_separate(5, 6, 5, N.STORE_F32) == 0
    S, C = (10, 12, 10), 16
    m = ft.SWMatricize((None, C, *S), head_dim=8, patch_size=(5, 6, 5), shifts=[None, (2, 3, 1), (1, 0, 2)])
    nmf = ft.NMF(size=(8, 150), rank=2, num_iters=5, num_grad_steps=3, init="uniform", solver="hals").to(DEV)
    t = torch.rand(2, C, *S, device=DEV).to(dt)          # bf16: windows 1, 2 go to their own buffers; fp32: read-modify-write
    t1, t2 = t.clone().requires_grad_(True), t.float().requires_grad_(True)
    a1 = Fn.FactCoreFn.apply(t1, nmf.init.u0, nmf.init.v0, m.geometry, 5, 3, "hals", 1e-16, True)
    a2 = m.inverse_forward(nmf(m(t2)))
    ga = torch.randn_like(a2).to(dt).float()
    (g1,) = torch.autograd.grad(a1, t1, ga.to(dt))
    (g2,) = torch.autograd.grad(a2, t2, ga)
    if dt == torch.float32:
        P.close("gt three windows", g1, g2 * (t > 0))
    else:   # three stored bf16 terms and their bf16 sums: 2^-8 of the largest element per rounding
        P.close("gt three windows, own buffers (bf16 storage)", g1.float(), g2 * (t > 0), rel=2e-2, why="bf16 storage of the three window terms")


# ---------------------------------------------------------------- other BASELINE / §8f configs ------
def _block_vs_oracle(C, S, reshape_kw, nmf_kw, mlp_ratio=2, B=1, tol=1e-4, why=None):
    torch.manual_seed(0)
    blk = ft.FactorizerBlock(channels=C, spatial_size=S, norm=ft.LayerNorm, reshape=(ft.SWMatricize, reshape_kw),
                             act=nn.ReLU, factorize=ft.NMF, init="uniform", mlp_ratio=mlp_ratio, dropout=0.0,
                             **nmf_kw)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    x = torch.rand(B, C, *S)
    gy = torch.rand_like(x)
    xo = x.clone().requires_grad_(True)
    cfg = dict(reshape=reshape_kw, num_iters=nmf_kw.get("num_iters", 5), solver=nmf_kw.get("solver", "hals"))
    yo = O.factorizer_block(xo, sd, "", cfg)
    (gxo,) = torch.autograd.grad(yo, xo, gy)
    blk = blk.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    with Launches():
        yd = blk(xd)
        (gxd,) = torch.autograd.grad(yd, xd, gy.to(DEV))
    P.close("y", yd, yo, rel=tol, why=why)
    P.close("gx", gxd, gxo, rel=tol, why=why)


def test_block_production_four_shift_windows():
    """the BraTS bundle's windows [None, 2, 4, 6] and mlp_ratio 4
    (model_zoo/factorizer_brats23/configs/train.yaml:50-54,62)."""
    _block_vs_oracle(32, (16, 16, 16), dict(head_dim=8, patch_size=8, shifts=[None, 2, 4, 6]),
                     dict(rank=1, num_iters=5, solver="hals"), mlp_ratio=4)


def test_block_cfg5_shape_rank2_t10():
    """BASELINE cfg 5 stress shape at reduced extent: anisotropic patch (5,6,5) (p = 8 is invalid for
    160x192x160, SURVEY headline 5), rank 2, 10 iterations — the masked 8x150 NMF family."""
    _block_vs_oracle(16, (10, 12, 20), dict(head_dim=8, patch_size=(5, 6, 5)),
                     dict(rank=2, num_iters=10, solver="hals"))


def test_block_mu_rank2_fused_core():
    _block_vs_oracle(16, (16, 16, 16), dict(head_dim=8, patch_size=8), dict(rank=2, num_iters=3, solver="mu"))


def test_block_training_dropout_runs():
    """README block with dropout 0.1 in training mode (factorizer.py:69,72): modular path, finite."""
    torch.manual_seed(0)
    blk = ft.FactorizerBlock(channels=16, spatial_size=(16, 16, 16), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                             factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2,
                             dropout=0.1).to(DEV).train()
    x = torch.rand(2, 16, 16, 16, 16, device=DEV, requires_grad=True)
    with Launches():
        y = blk(x)
        y.sum().backward()
    assert torch.isfinite(y).all() and torch.isfinite(x.grad).all()
    blk.eval()
    y1, y2 = blk(x), blk(x)
    assert torch.equal(y1, y2)


def test_readme_model_full_size_train_step():
    """BASELINE cfg 3/4: README Swin Factorizer (in 4, out 3, 128^3, widths 32-512) forward, backward
    and one AdamW step (the flat one-kernel optimizer of bench.py) at the per-GPU batch of cfg 4, B = 2;
    outputs / gradients finite, every parameter moved, replicas deterministic."""
    torch.manual_seed(0)
    model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(128, 128, 128), norm=ft.LayerNorm,
                          reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                          factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2,
                          dropout=0.1).to(DEV).eval()
    x = torch.rand(2, 4, 128, 128, 128, device=DEV)
    t = (torch.rand(2, 3, 128, 128, 128, device=DEV) > 0.5).float()
    opt = ft.FlatAdamW(model, lr=1e-4, weight_decay=1e-5)      # train.yaml:72-76
    with Launches():
        y = model(x)
        loss = ft.dice_ce_loss(y, t)
        loss.backward()
    assert y.shape == (2, 3, 128, 128, 128) and torch.isfinite(y).all() and torch.isfinite(loss)
    for n, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    with torch.no_grad():
        y2 = model(x)
    assert torch.equal(y, y2)  # deterministic kernels (no float atomics)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    opt.step()
    for n, p in model.named_parameters():
        assert torch.isfinite(p).all(), n
        if before[n].abs().max() > 0 and p.grad.abs().max() > 0:
            assert not torch.equal(p.detach(), before[n]), n
        # first AdamW step: |Δp| <= lr·(1 + wd·|p|) elementwise
        assert ((p.detach() - before[n]).abs() <= 1.001e-4 * (1 + 1e-5 * before[n].abs()) + 1e-12).all(), n
    with torch.no_grad():
        y3 = model(x)
    assert torch.isfinite(y3).all() and not torch.equal(y3, y)


# ---------------------------------------------------------------- sliding-window inference (§8 f-1) ----
@pytest.mark.parametrize("size,roi,ov", [((20, 24, 27), (16, 16, 16), 0.5), ((32, 16, 40), (16, 16, 8), 0.25),
                                         ((16, 16, 16), (16, 16, 16), 0.5)])
def test_sliding_window_inference_native(size, roi, ov):
    """Window gather / Gaussian-weighted accumulate / divide kernels (csrc/sw_infer.hip) against the
    dense-map restatement in the oracle; W = 27 and the pulled-back last window (x0 = 11) take the
    unaligned path, W = 40 the 16-byte path."""
    torch.manual_seed(0)
    conv = torch.nn.Conv3d(2, 3, 3, padding=1)
    net = lambda x: torch.tanh(conv(x))  # noqa: E731
    x = torch.randn(2, 2, *size)
    with torch.no_grad():
        yo = O.sliding_window_oracle(x, roi, 2, net, overlap=ov, mode="gaussian")
        conv.to(DEV)
        with Launches():
            y = ft.sliding_window_inference(x.to(DEV), roi, 2, net, overlap=ov, mode="gaussian")
    assert torch.allclose(y.cpu(), yo, rtol=1e-5, atol=1e-5)


def test_sliding_window_brats_volume_with_factorizer():
    """The bundle's inference geometry end to end: a 240 x 240 x 155 volume through the README Swin
    Factorizer with roi 128^3, sw_batch 2, overlap 0.5, Gaussian blending (inference.yaml:96-102)."""
    torch.manual_seed(0)
    model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(128, 128, 128), norm=ft.LayerNorm,
                          reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                          factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2,
                          dropout=0.1).to(DEV).eval()
    x = torch.rand(1, 4, 240, 240, 155, device=DEV)
    inf = ft.SlidingWindowInfererAdapt(roi_size=(128, 128, 128), sw_batch_size=2, overlap=0.5, mode="gaussian")
    with torch.no_grad(), Launches():
        y = inf(x, model)
    assert y.shape == (1, 3, 240, 240, 155) and torch.isfinite(y).all()
    # same windows, same (deterministic) network outputs, stitched with framework ops
    from factorizer_amd.inference import sliding_window_inference
    with torch.no_grad():
        y2 = sliding_window_inference(x, (128,) * 3, 2, model, overlap=0.5, mode="gaussian", _composed=True)
    assert torch.allclose(y, y2, rtol=1e-5, atol=1e-5)
    # partition of unity: a constant network comes back exactly constant
    ones = inf(x, lambda w: torch.ones(w.shape[0], 1, *w.shape[2:], device=w.device))
    assert torch.allclose(ones, torch.ones_like(ones), rtol=0, atol=1e-6)


# ---------------------------------------------------------------- edge cases: empty / single inputs ------
def test_empty_batch_through_modules():
    """Batch 0 takes the zero-work framework path on device (nothing to launch) and keeps shapes."""
    m = ft.SWMatricize((None, 16, 8, 8, 8), head_dim=8, patch_size=4)
    x = torch.rand(0, 16, 8, 8, 8, device=DEV)
    y = m(x)
    assert y.shape == (0, 8, 8, 64) and m.inverse_forward(y).shape == x.shape
    nmf = ft.NMF(size=(8, 64), rank=2, num_iters=3, init="uniform", solver="hals").to(DEV)
    assert nmf(y).shape == y.shape
    u, v = nmf.decompose(y)
    assert u.shape == (0, 8, 8, 2) and v.shape == (0, 8, 64, 2)
    blk = ft.FactorizerBlock(channels=32, spatial_size=(8, 8, 8), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                             factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2,
                             dropout=0.0).to(DEV)
    xb = torch.rand(0, 32, 8, 8, 8, device=DEV, requires_grad=True)
    yb = blk(xb)
    assert yb.shape == xb.shape
    yb.sum().backward()
    assert xb.grad.shape == xb.shape


def test_single_matrix_and_single_patch():
    """One matrix per launch (BASELINE cfg 1 on device) and a volume that is exactly one patch."""
    torch.manual_seed(0)
    nmf = ft.NMF(size=(8, 512), rank=2, num_iters=5, init="uniform", solver="mu")
    x = torch.rand(1, 8, 512)
    ref = nmf(x)
    with Launches():
        out = nmf.to(DEV)(x.to(DEV))
    P.close("single matrix", out, ref)
    torch.manual_seed(0)
    blk = ft.FactorizerBlock(channels=8, spatial_size=(8, 8, 8), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                             factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2,
                             dropout=0.0)
    xs = torch.rand(1, 8, 8, 8, 8)
    ref = blk(xs)
    with Launches():
        out = blk.to(DEV)(xs.to(DEV))
    P.close("single patch block", out, ref)


def _small_unet():
    torch.manual_seed(0)
    return ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(32, 32, 32), encoder_depth=(1, 1, 1),
                         encoder_width=(32, 64, 128), strides=(1, 2, 2), decoder_depth=(1, 1), norm=ft.LayerNorm,
                         reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU, factorize=ft.NMF,
                         rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0).to(DEV)


def test_late_wgrad_join_matches_per_block_join(monkeypatch):
    """(opt-in second stream, FZ_SIDE_WGRAD: off by default since round 3, profiles/r03_two_stream_interaction.md)
    Awaiting the side-stream weight gradients once per step (pointwise._LateJoin) instead of once per
    block changes when the streams meet, not the values: every gradient is bit-identical (fp32)."""
    from factorizer_amd import pointwise as PW
    monkeypatch.setenv("FZ_SIDE_WGRAD", str(2 * 64 ** 3))
    from factorizer_amd.parallel import FlatGradSync
    model = _small_unet()
    x = torch.rand(2, 4, 32, 32, 32, device=DEV)
    t = (torch.rand(2, 3, 32, 32, 32, device=DEV) > 0.5).float()

    def grads(late):
        sync = FlatGradSync(model, num_buckets=2, late_wgrad_join=late)
        sync.zero_grad()
        n0 = PW._LateJoin.verified
        ft.dice_ce_loss(model(x), t).backward()
        if late:  # joined (and ownership verified) by the end-of-backward callback
            assert PW._LateJoin.verified > n0, "no block put its weight gradients on the side stream"
        sync.finish()
        assert not PW._LateJoin.owed and not PW._LateJoin.keep
        torch.cuda.synchronize()
        out = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
        PW.late_wgrad_join(False)
        return out

    a, b = grads(False), grads(True)
    for n in a:
        assert torch.equal(a[n], b[n]), n


def test_late_wgrad_join_refuses_copied_gradients(monkeypatch):
    """(opt-in second stream) If autograd had to copy a returned weight gradient (here: accumulation into an existing .grad),
    the copy was taken before the side stream wrote it: the join raises instead of handing it to the optimizer."""
    from factorizer_amd import pointwise as PW
    monkeypatch.setenv("FZ_SIDE_WGRAD", str(2 * 64 ** 3))
    model = _small_unet()
    x = torch.rand(2, 4, 32, 32, 32, device=DEV)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    PW.late_wgrad_join(True)
    try:
        with pytest.raises(RuntimeError, match="late_wgrad_join"):
            model(x).sum().backward()      # raised by the end-of-backward join
    finally:
        PW._LateJoin.owed.clear()
        PW.late_wgrad_join(False)
        torch.cuda.synchronize()


@pytest.mark.parametrize("name,kw", [("fmu", dict(solver="fmu", init="uniform")), ("mu_0", dict(solver="mu-0", init="uniform"))])
def test_solver_keys_that_are_the_mu_update_run_native(golden, name, kw):
    """`fmu` (matrix_factorization.py:250-274) is the multiplicative update written as three-operand contractions: the same
    function, different rounding — on device it runs the native MU kernels and must reproduce the REFERENCE's own fmu outputs
    (goldens g8: u, v, y, dL/dx).  `mu-0` (U half-steps only) is outside the native kernels' U-then-V alternation: composed, warned."""
    g = golden("g8_solvers").case(name)
    torch.manual_seed(0)
    mf = ft.MatrixFactorization(size=(8, 24), rank=2, num_iters=3, **kw).to(DEV)
    x = g["x"].to(DEV).requires_grad_(True)
    if name == "fmu":
        with Launches():
            u, v = mf.decompose(x)
            y = mf(x)
            (gx,) = torch.autograd.grad(y, x, g["gy"].to(DEV))
    else:
        n0 = _native.launch_count()
        u, v = mf.decompose(x)
        y = mf(x)
        (gx,) = torch.autograd.grad(y, x, g["gy"].to(DEV))
        assert _native.launch_count() == n0      # composed ATen on device (the native kernels alternate U then V)
    P.close("u", u, g["u"])
    P.close("v", v, g["v"])
    P.close("y", y, g["y"])
    P.close("gx", gx, g["gx"])


def test_wmu_without_weights_runs_native_and_equals_mu():
    """wmu with no weights passed is the mu update (matrix_factorization.py:277-316 with w = 1): native kernels, same values as
    solver="mu" bit for bit; with weights it stays composed (goldens g8 "wmu" on CPU)."""
    torch.manual_seed(0)
    a = ft.NMF(size=(8, 64), rank=2, num_iters=4, init="uniform", solver="wmu").to(DEV)
    b = ft.NMF(size=(8, 64), rank=2, num_iters=4, init="uniform", solver="mu").to(DEV)
    b.load_state_dict(a.state_dict())
    x = torch.rand(5, 8, 64, device=DEV)
    with Launches():
        ya = a(x)
    assert torch.equal(ya, b(x))
    w = torch.rand(5, 8, 64, device=DEV)
    u, v = a.decompose(x, w)          # weights: the composed path (same ops on CPU give the reference's values: goldens g8 "wmu")
    ac = ft.NMF(size=(8, 64), rank=2, num_iters=4, init="uniform", solver="wmu")
    ac.load_state_dict({k: t.cpu() for k, t in a.state_dict().items()})
    uc, vc = ac.decompose(x.cpu(), w.cpu())
    P.close("u (weighted, composed on device)", u, uc)
    P.close("v (weighted, composed on device)", v, vc)
