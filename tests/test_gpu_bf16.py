"""-m gpu: the mixed-precision leg of BASELINE configs[4] — activations STORED as bf16 in HBM, parameters /
statistics / accumulation / the whole NMF iteration in fp32 (include/factorizer_hip.h FZ_STORE_BF16;
SURVEY.md §5: eps = 1e-16 of matrix_factorization.py:200,236 underflows in fp16, survives bf16, so the
factors and Gram matrices never leave fp32).

Reference for every comparison: the fp32 CPU oracle / ATen evaluated on the SAME bf16-rounded inputs.
Tolerances are stated in units of the bf16 unit roundoff u = 2^-8 (8 significand bits, round to nearest
even: the relative error of one store): a kernel that only rounds its OUTPUT must be within 1 u of max|ref|
(plus the fp32 noise floor); a chain of n stored tensors is held to n·u — the worst case of errors adding
coherently through maps of unit gain — with n counted in each test.  Where the map between two stored
tensors is NOT of unit gain (the gradient through T = 10 rank-2 HALS iterations amplifies a perturbation of
its input ~35x), the reference is the oracle with the same storage roundings inserted
(oracle.cpu_ref.factorizer_block_bf16_storage), and the distance of THAT to the plain fp32 oracle is
recorded next to the device's."""
import warnings

import pytest
import torch
import torch.nn.functional as F
from torch import nn

import factorizer_amd as ft
import parity as P
from factorizer_amd import _native
from factorizer_amd import functional as Fn
from factorizer_amd import pointwise as PW
from oracle import cpu_ref as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
U = 2.0 ** -8   # bf16 unit roundoff (8 significand bits, RNE)


def rb(t):
    """round to bf16 and come back: the value a bf16 tensor in HBM actually holds"""
    return t.to(BF).float()


def close(what, got, ref, n_stores, extra=0.0):
    """max|got − ref| ≤ n_stores · u · max|ref|"""
    assert got.dtype in (BF, torch.float32)
    return P.close(what, got.float(), ref, rel=n_stores * U, floor=1e-6, extra=extra,
                   why=f"bf16 storage: {n_stores} stored tensor(s) x u = 2^-8 between input and this result")


def _lin_cpu(x, w, b=None):
    return F.conv1d(x.flatten(2), w, b).reshape(x.shape[0], w.shape[0], *x.shape[2:])


# ---------------------------------------------------------------- single kernels ------------------------
@pytest.mark.parametrize("B,Cin,Cout,S", [(2, 32, 32, (8, 8, 8)), (1, 64, 32, (6, 4, 4)), (1, 32, 64, (8, 8, 16)),
                                          (1, 128, 128, (4, 4, 8)), (1, 512, 256, (4, 4, 4)), (1, 32, 3, (8, 8, 8))])
def test_linear_bf16_storage(B, Cin, Cout, S):
    """Resident / streaming GEMM with bf16 loads and a bf16 epilogue, fp32 MFMA accumulation: forward, input
    gradient and (bf16-MFMA, fp32-accumulated) weight gradient against ATen fp32 on the rounded operands."""
    torch.manual_seed(0)
    x = rb(torch.randn(B, Cin, *S))
    w = torch.randn(Cout, Cin, 1) / Cin ** 0.5
    b = torch.randn(Cout)
    gy = rb(torch.randn(B, Cout, *S))
    xc, wc, bc = (t.clone().requires_grad_(True) for t in (x, w, b))
    yc = _lin_cpu(xc, wc, bc)
    gxc, gwc, gbc = torch.autograd.grad(yc, [xc, wc, bc], gy)
    xd = x.to(DEV, BF).requires_grad_(True)
    wd, bd = w.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    n0 = _native.launch_count()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        yd = PW.linear_cf(xd, wd, bd)
        gxd, gwd, gbd = torch.autograd.grad(yd, [xd, wd, bd], gy.to(DEV, BF))
    assert _native.launch_count() > n0 and yd.dtype == BF and gxd.dtype == BF
    assert gwd.dtype == torch.float32 and gbd.dtype == torch.float32      # parameters' gradients stay fp32
    close("y", yd, yc, 1)
    close("gx", gxd, gxc, 1)
    close("gw (operands exact in bf16: fp32-accumulated products)", gwd, gwc, 1)
    close("gb", gbd, gbc, 1)


@pytest.mark.parametrize("C,M,S", [(32, 32, (8, 8, 8)), (64, 128, (4, 4, 8)), (256, 256, (4, 4, 4))])
def test_ln_linear_relu_bf16_storage(C, M, S):
    torch.manual_seed(1)
    x = rb(torch.randn(2, C, *S) * 2 + 0.5)
    g, bt = torch.rand(C) + 0.5, torch.randn(C)
    w = torch.randn(M, C, 1) / C ** 0.5
    gy = rb(torch.randn(2, M, *S))

    def cpu(x, g, bt, w):
        return torch.relu(_lin_cpu(F.layer_norm(x.movedim(1, -1), (C,), g, bt, 1e-5).movedim(-1, 1), w))
    cs = [t.clone().requires_grad_(True) for t in (x, g, bt, w)]
    yc = cpu(*cs)
    gc = torch.autograd.grad(yc, cs, gy)
    ds = [x.to(DEV, BF).requires_grad_(True)] + [t.to(DEV).requires_grad_(True) for t in (g, bt, w)]
    yd = PW.ln_linear(ds[0], ds[1], ds[2], 1e-5, ds[3], None, "relu")
    gd = torch.autograd.grad(yd, ds, gy.to(DEV, BF))
    close("y", yd, yc, 1)
    # backward: gl (stored bf16) -> LayerNorm backward -> gx ; weight gradient sees LN(x) rounded to bf16
    close("gx", gd[0], gc[0], 3)
    for k, a, b in zip(("dgamma", "dbeta", "gw"), gd[1:], gc[1:]):
        close(k, a, b, 2)


@pytest.mark.parametrize("Hd", [64, 128])
def test_mlp_chain_bf16_storage(Hd):
    """fz_mlp_chain with bf16 activations: x2 = x1 + fc2(gelu(fc1(LN(x1)))) and its backward chain."""
    torch.manual_seed(3)
    C, B, S = 32, 2, (8, 8, 12)
    x = rb(torch.randn(B, C, *S))
    lw, lb = torch.rand(C) + 0.5, torch.randn(C) * 0.1
    w1, b1 = torch.randn(Hd, C) / C ** 0.5, torch.randn(Hd) * 0.1
    w2, b2 = torch.randn(C, Hd) / Hd ** 0.5, torch.randn(C) * 0.1
    g2 = rb(torch.randn(B, C, *S))
    xc = x.clone().requires_grad_(True)
    z1c = _lin_cpu(F.layer_norm(xc.movedim(1, -1), (C,), lw, lb, 1e-5).movedim(-1, 1), w1[:, :, None], b1)
    x2c = xc + _lin_cpu(F.gelu(z1c), w2[:, :, None], b2)
    (gxc,) = torch.autograd.grad(x2c, xc, g2)
    xd = x.to(DEV, BF)
    x2, z1, st = PW._mlp_fwd_chain(xd, lw.to(DEV), lb.to(DEV), 1e-5, w1.to(DEV), b1.to(DEV), w2.to(DEV), b2.to(DEV))
    assert x2.dtype == BF and z1.dtype == BF and st.dtype == torch.float32
    close("x2", x2, x2c, 1)
    close("z1", z1, z1c, 1)
    gz1, gx1, gg, gb = PW._mlp_bwd_chain(g2.to(DEV, BF), z1, w1.to(DEV), w2.to(DEV), xd, st, lw.to(DEV))
    assert gx1.dtype == BF and gg.dtype == torch.float32
    close("gx1 (z1 was stored in bf16)", gx1, gxc, 3)


def test_convs_bf16_storage():
    """k2s2 down-conv (space-to-depth loader), k2s2 transposed conv (depth-to-space epilogue), k3 stem, k1 head."""
    torch.manual_seed(4)
    cases = [
        ("conv_k2s2", ft.Conv3d(16, 32, 2, stride=2), (2, 16, 8, 8, 8), lambda m, x: F.conv3d(x, m.weight, m.bias, stride=2)),
        ("tconv_k2s2", ft.ConvTranspose3d(32, 16, 2, stride=2), (2, 32, 4, 4, 8), lambda m, x: F.conv_transpose3d(x, m.weight, m.bias, stride=2)),
        ("stem_k3", ft.Conv3d(4, 32, 3, padding=1, bias=False), (1, 4, 8, 8, 32), lambda m, x: F.conv3d(x, m.weight, None, padding=1)),
        ("head_k1", ft.Conv3d(32, 3, 1), (1, 32, 8, 8, 8), lambda m, x: F.conv3d(x, m.weight, m.bias)),
    ]
    for name, mod, shape, ref in cases:
        x = rb(torch.randn(shape))
        xc = x.clone().requires_grad_(True)
        yc = ref(mod, xc)
        gy = rb(torch.randn_like(yc))
        pc = list(mod.parameters())
        gc = torch.autograd.grad(yc, [xc] + pc, gy)
        md = type(mod)(mod.in_channels, mod.out_channels, mod.kernel_size, stride=mod.stride, padding=mod.padding,
                       bias=mod.bias is not None).to(DEV)
        md.load_state_dict(mod.state_dict())
        stem = name == "stem_k3"   # the stem's input is data: its input gradient is not part of the training step
        xd = x.to(DEV, BF).requires_grad_(not stem)
        n0 = _native.launch_count()
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)
            yd = md(xd)
            gd = torch.autograd.grad(yd, ([] if stem else [xd]) + list(md.parameters()), gy.to(DEV, BF))
        assert _native.launch_count() > n0 and yd.dtype == BF, name
        close(name + " y", yd, yc, 1)
        if not stem:
            close(name + " gx", gd[0], gc[0], 1)
        for i, (a, b) in enumerate(zip(gd[0 if stem else 1:], gc[1:])):
            assert a.dtype == torch.float32
            close(f"{name} gparam{i}", a, b, 1)


def test_swm_and_nmf_bf16_storage():
    """Modular chain of the cfg-5 shape: matricize (bit-exact move) → masked 8x150 NMF with bf16 loads/stores and
    fp32 U / V / Gram / eps → inverse (fp32 window sum, one rounding); and the hot 8x512 family."""
    torch.manual_seed(5)
    for (S, p, R, T, solver) in (((10, 12, 10), (5, 6, 5), 2, 10, "hals"), ((8, 8, 16), 8, 1, 5, "hals"), ((8, 8, 8), 8, 2, 4, "mu")):
        C = 16
        m = ft.SWMatricize((None, C, *S), head_dim=8, patch_size=p)
        N = m.output_size[3]
        nmf = ft.NMF(size=(8, N), rank=R, num_iters=T, init="uniform", solver=solver)
        x = rb(torch.rand(2, C, *S))
        x[0, :8, :5, :6, :5] = 0                          # a patch of zeros: the eps path (NaN in fp16)
        xc = x.clone().requires_grad_(True)
        mc = m(xc)
        u0, v0 = nmf.init.u0.clone(), nmf.init.v0.clone()
        yc = m.inverse_forward(O.nmf_forward(mc, u0, v0, T, solver))
        ga = rb(torch.rand_like(yc))
        (gxc,) = torch.autograd.grad(yc, xc, ga)
        nd = nmf.to(DEV)
        assert nd.init.u0.dtype == torch.float32
        xd = x.to(DEV, BF).requires_grad_(True)
        n0 = _native.launch_count()
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)
            md = m(xd)
            assert md.dtype == BF and torch.equal(md.float().cpu(), mc.detach())     # pure data movement
            yd = m.inverse_forward(nd(md))
            (gxd,) = torch.autograd.grad(yd, xd, ga.to(DEV, BF))
        assert _native.launch_count() > n0 and yd.dtype == BF and torch.isfinite(yd.float()).all()
        close(f"{S} y: NMF output + window average", yd, yc, 2)
        # backward: ga/W (bf16) -> NMF backward (fp32 inside; the forward recomputed from the SAME bf16 x) ->
        # gm (bf16) -> window sum (bf16): 3 stores; HALS gates sit on the same values in both evaluations
        close(f"{S} gx", gxd, gxc, 3)
        u, v = nd.decompose(md)
        assert u.dtype == torch.float32 and v.dtype == torch.float32                 # factors never leave fp32
        uo, vo = O.nmf_decompose(mc.detach(), u0, v0, T, solver)
        P.close(f"{S} u (fp32 internals)", u, uo)
        P.close(f"{S} v (fp32 internals)", v, vo)


def test_fused_core_bf16_matches_modular_bf16():
    """csrc/nmf_cf.hip with bf16 loads/stores against the fp32 oracle on the rounded input (window 0 is stored in
    bf16 before window 1 is added: 2 roundings forward)."""
    torch.manual_seed(6)
    C, S = 16, (8, 16, 64)
    for shifts, solver, R in ((None, "hals", 1), ([None, 2, 4, 6], "hals", 1), (None, "mu", 2)):
        m = ft.SWMatricize((None, C, *S), head_dim=8, patch_size=8, shifts=shifts)
        nmf = ft.NMF(size=(8, 512), rank=R, num_iters=4, init="uniform", solver=solver)
        t = rb(torch.rand(2, C, *S))
        tc = t.clone().requires_grad_(True)
        ac = m.inverse_forward(O.nmf_forward(m(tc), nmf.init.u0.clone(), nmf.init.v0.clone(), 4, solver))
        ga = rb(torch.rand_like(ac))
        (gtc,) = torch.autograd.grad(ac, tc, ga)
        W = m.geometry.nshift
        td = t.to(DEV, BF).requires_grad_(True)
        u0, v0 = nmf.init.u0.to(DEV), nmf.init.v0.to(DEV)
        n0 = _native.launch_count()
        ad = Fn.FactCoreFn.apply(td, u0, v0, m.geometry, 4, 4, solver, 1e-16, False)
        (gtd,) = torch.autograd.grad(ad, td, ga.to(DEV, BF))
        assert _native.launch_count() > n0 and ad.dtype == BF and gtd.dtype == BF
        close(f"a W={W} {solver}", ad, ac, W)
        close(f"gt W={W} {solver}", gtd, gtc, W + 1)


# ---------------------------------------------------------------- BASELINE configs[4] ---------------------
def _block_bf16_vs_fp32_oracle(C, S, reshape_kw, nmf_kw, mlp_ratio=2, B=2, grad_vs="fp32"):
    """Device (bf16 storage) against (a) the fp32 oracle and (b) the oracle with the same storage roundings
    inserted, all on the same bf16-rounded input.  grad_vs: which of the two the GRADIENTS are held to."""
    torch.manual_seed(0)
    blk = ft.FactorizerBlock(channels=C, spatial_size=S, norm=ft.LayerNorm, reshape=(ft.SWMatricize, reshape_kw),
                             act=nn.ReLU, factorize=ft.NMF, init="uniform", mlp_ratio=mlp_ratio, dropout=0.0, **nmf_kw)
    sd = {k: v.clone() for k, v in blk.state_dict().items()}
    x = rb(torch.rand(B, C, *S))
    gy = rb(torch.rand_like(x))
    cfg = dict(reshape=reshape_kw, num_iters=nmf_kw["num_iters"], solver=nmf_kw["solver"])

    def oracle(fn):
        params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if not k.endswith(("u0", "v0"))}
        full = dict(sd)
        full.update(params)
        xo = x.clone().requires_grad_(True)
        yo = fn(xo, full, "", cfg)
        return yo.detach(), torch.autograd.grad(yo, [xo] + list(params.values()), gy), list(params.keys())

    yo, go, pnames = oracle(O.factorizer_block)
    ye, ge, _ = oracle(O.factorizer_block_bf16_storage)
    blk = blk.to(DEV)
    xd = x.to(DEV, BF).requires_grad_(True)
    names = [k for k, _ in blk.named_parameters()]
    assert names == pnames
    n0 = _native.launch_count()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)       # no composed fallback on this configuration
        yd = blk(xd)
        gd = torch.autograd.grad(yd, [xd] + list(blk.parameters()), gy.to(DEV, BF))
    torch.cuda.synchronize()
    assert _native.launch_count() > n0 and yd.dtype == BF and gd[0].dtype == BF
    assert torch.isfinite(yd.float()).all()
    # forward: t, ym, a, x1, z1, x2 are stored in bf16 between kernels (6 stores + the W-window accumulate)
    # (measured: 1.2 u against either oracle — the roundings mostly do not add coherently)
    close("y vs fp32 oracle", yd, yo, 4)
    close("y vs storage-emulating oracle", yd, ye, 4)

    def rel(a, b):
        return ((a.float().cpu() - b).abs().max() / (b.abs().max() + 1e-30)).item()
    P.note("storage_effect_on_gx (emulating oracle vs fp32 oracle) / device vs fp32 / device vs emulating",
           emul_vs_fp32=rel(ge[0], go[0]), device_vs_fp32=rel(gd[0], go[0]), device_vs_emul=rel(gd[0], ge[0]),
           rank=nmf_kw["rank"], num_iters=nmf_kw["num_iters"])
    gref = go if grad_vs == "fp32" else ge
    # backward: 7 more stored gradients (gz1, gx1, ga, gym, gm, gt, gx) on top of the rounded forward tensors;
    # measured 1-2 u, held to 4 u
    close(f"gx vs {grad_vs} oracle", gd[0], gref[0], 4)
    for k, a, b in zip(names, gd[1:], gref[1:]):
        assert a.dtype == torch.float32, k
        close(f"grad:{k} vs {grad_vs} oracle", a, b, 4)
    return yd


def test_block_cfg5_bf16_rank2_t10():
    """BASELINE configs[4] block: anisotropic patch (5,6,5) (p = 8 does not divide 160x192x160, SURVEY headline 5),
    HALS rank 2, 10 iterations, bf16 activations / fp32 NMF internals, at reduced extent, forward and backward
    with every parameter gradient, against the fp32 oracle on the same (bf16-rounded) input."""
    # gradients against the storage-emulating oracle: bf16 storage of the NMF input alone moves gx by ~13 % of
    # its maximum at R = 2, T = 10 (measured on the CPU oracle: emul_vs_fp32 in the recorded note; 4 % at
    # T = 5, 0.3 % at R = 1) — a property of the configuration, not of the kernels
    _block_bf16_vs_fp32_oracle(32, (10, 12, 20), dict(head_dim=8, patch_size=(5, 6, 5)),
                               dict(rank=2, num_iters=10, solver="hals"), grad_vs="storage-emulating")


def test_block_cfg2_bf16_fused_core():
    """The README block (p = 8, HALS R1 T5: fused core + MLP chain kernels) in the same mixed-precision mode."""
    _block_bf16_vs_fp32_oracle(32, (16, 16, 32), dict(head_dim=8, patch_size=8), dict(rank=1, num_iters=5, solver="hals"))


def test_block_production_windows_bf16():
    _block_bf16_vs_fp32_oracle(32, (16, 16, 16), dict(head_dim=8, patch_size=8, shifts=[None, 2, 4, 6]),
                               dict(rank=1, num_iters=5, solver="hals"), mlp_ratio=4)


def test_model_autocast_bf16_training_step():
    """Whole U-shape under torch.autocast(bfloat16): every activation is bf16, every parameter / gradient fp32, one
    AdamW step moves the weights; gradients agree in direction with the fp32 run (cosine > 0.99 per tensor with
    a non-negligible gradient); float16 autocast is refused (eps underflow)."""
    torch.manual_seed(0)
    kw = dict(in_channels=4, out_channels=3, spatial_size=(32, 32, 32), encoder_depth=(1, 1, 1),
              encoder_width=(32, 64, 128), strides=(1, 2, 2), decoder_depth=(1, 1), norm=ft.LayerNorm,
              reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU, factorize=ft.NMF,
              rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)
    model = ft.Factorizer(**kw).to(DEV)
    x = torch.rand(2, 4, 32, 32, 32, device=DEV)
    t = (torch.rand(2, 3, 32, 32, 32, device=DEV) > 0.5).float()
    ft.dice_ce_loss(model(x), t).backward()
    g32 = {n: p.grad.clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    seen = []
    hooks = [m.register_forward_hook(lambda mod, i, o: seen.append(o.dtype) if torch.is_tensor(o) else None)
             for m in model.modules() if isinstance(m, ft.FactorizerBlock)]
    n0 = _native.launch_count()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        with torch.autocast("cuda", dtype=BF):
            y = model(x)
            loss = ft.dice_ce_loss(y, t)
        loss.backward()
    for h in hooks:
        h.remove()
    assert _native.launch_count() > n0
    assert y.dtype == BF and seen and all(d == BF for d in seen)
    assert torch.isfinite(loss) and loss.dtype == torch.float32
    worst = 1.0
    for n, p in model.named_parameters():
        assert p.dtype == torch.float32 and p.grad is not None and p.grad.dtype == torch.float32, n
        assert torch.isfinite(p.grad).all(), n
        if g32[n].norm() > 1e-6 * max(1.0, g32[n].numel() ** 0.5):
            cos = F.cosine_similarity(p.grad.flatten(), g32[n].flatten(), dim=0).item()
            worst = min(worst, cos)
            assert cos > 0.99, (n, cos)
    P.note("autocast_bf16_vs_fp32_gradient_cosine_min", value=worst)
    opt = ft.FlatAdamW(model, lr=1e-4, weight_decay=1e-5)
    before = model.stem.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, model.stem.weight.detach())
    with pytest.raises(RuntimeError, match="float16"):
        with torch.autocast("cuda", dtype=torch.float16):
            model(x)
    with pytest.raises(RuntimeError, match="float16"):
        model(x.half())
