"""-m gpu: the split-bf16 MFMA GEMM family (csrc/gemm_bx.hip) — fp32 layers with a reduction length >= 64 run as six exact
bf16 products per fp32 product on the bf16 matrix cores.  The claim to verify is "fp32 accuracy": for every loader /
prologue / epilogue of the family the error against an fp64 evaluation on the CPU must not exceed that of the
fp32-MFMA kernels (v_mfma_f32_32x32x2_f32, `fz_gemm_bx_enable(0)`) on the same inputs by more than a rounding's worth,
and both must sit at fp32 rounding level.  Layers: linear.py:53-58, norm.py:29-34, mlp.py:54-63, unet.py:53,123,128."""
import pytest
import torch
import torch.nn.functional as F

from factorizer_amd import _native
from factorizer_amd import pointwise as PW
import parity as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both(fn):
    """fn() -> tensor, evaluated with the split-bf16 family on and off"""
    n0 = _native.launch_count()
    with _native.use_products(_native.PRODUCTS_SPLIT_BF16):     # the descriptors' own field: no process-wide switch is touched
        y_bx = fn().double().cpu()
    assert _native.launch_count() > n0
    with _native.use_products(_native.PRODUCTS_FP32_MFMA):
        y_f32 = fn().double().cpu()
    return y_bx, y_f32


def _check(what, y_bx, y_f32, ref, mag, slack=1.0):
    """Errors in units of the standard dot-product scale: |y - ref| / mag, mag[m, n] = Σ_k |a_mk| |b_kn| (+ |bias|, |res|)
    evaluated in fp64 — an fp32 FMA chain of length K errs by (sqrt(K)..K)·2^-24 of it depending on the summation order,
    which differs between the two kernels (the fp32 family splits K over waves for narrow problems).  The split-bf16
    kernel must be within 1.5x of the fp32-MFMA kernel or within 4 roundings (4·2^-24), whichever is larger."""
    e_bx = float(((y_bx - ref).abs() / mag).max())
    e_f32 = float(((y_f32 - ref).abs() / mag).max())
    P.note(what, err_bx_over_sum_abs=e_bx, err_f32_mfma_over_sum_abs=e_f32)
    assert e_bx <= max(1.5 * e_f32, slack * 4 * 2.0 ** -24), (what, e_bx, e_f32)


def _lin_mag(x, w, b=None, res=None):
    """Σ_k |w_mk| |x_kn| (+ |b_m| + |res|) in fp64, shaped like the layer output"""
    B, M = x.shape[0], w.shape[0]
    m = F.conv1d(x.double().abs().flatten(2), w.double().abs().reshape(M, -1, 1), None if b is None else b.double().abs())
    m = m.reshape(B, M, *x.shape[2:])
    return m if res is None else m + res.double().abs()


LIN = [(2, 64, 64, (8, 8, 8)), (1, 128, 64, (8, 8, 4)), (2, 256, 512, (4, 4, 4)), (1, 1024, 512, (4, 4, 4)),
       (1, 80, 96, (6, 4, 4)),      # K, M not multiples of 16 / 32, partial tiles
       (2, 64, 32, (16, 16, 8)),    # wide tiles (NACC 4)
       (1, 512, 1024, (8, 8, 8)),
       (1, 96, 64, (8, 8, 8))]      # K % 64 == 32: the trailing half chunk runs on zero weights


@pytest.mark.parametrize("B,Cin,Cout,S", LIN)
def test_linear_fp64(B, Cin, Cout, S):
    torch.manual_seed(0)
    x = torch.randn(B, Cin, *S)
    w = torch.randn(Cout, Cin, 1) / Cin ** 0.5
    b = torch.randn(Cout)
    ref = F.conv1d(x.double().flatten(2), w.double(), b.double()).reshape(B, Cout, *S)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y_bx, y_f32 = _both(lambda: PW.linear_cf(xd, wd, bd))
    _check(f"linear {Cin}->{Cout}", y_bx, y_f32, ref, _lin_mag(x, w, b))


@pytest.mark.parametrize("B,Cin,Cout,S", LIN[:5])
def test_linear_backward_fp64(B, Cin, Cout, S):
    """input gradient (transposed weights, w_t path) of the plain layer and of the GELU / ReLU-gated forms"""
    torch.manual_seed(1)
    x = torch.randn(B, Cin, *S)
    w = torch.randn(Cout, Cin, 1) / Cin ** 0.5
    gy = torch.randn(B, Cout, *S)
    ref = torch.einsum("oc,bov->bcv", w[:, :, 0].double(), gy.double().flatten(2)).reshape(B, Cin, *S)
    xd, wd, gd = x.to(DEV).requires_grad_(True), w.to(DEV), gy.to(DEV)

    def run():
        (g,) = torch.autograd.grad(PW.linear_cf(xd, wd, None), xd, gd)
        return g
    y_bx, y_f32 = _both(run)
    mag = torch.einsum("oc,bov->bcv", w[:, :, 0].double().abs(), gy.double().abs().flatten(2)).reshape(B, Cin, *S)
    _check(f"linear dgrad {Cout}->{Cin}", y_bx, y_f32, ref, mag)


@pytest.mark.parametrize("B,C,M,S", [(2, 64, 64, (8, 8, 8)), (1, 128, 256, (8, 4, 4)), (1, 512, 512, (4, 4, 4)),
                                     (2, 96, 64, (8, 8, 8))])   # (K % 64 == 32: LayerNorm sums must skip the clamped re-read)
@pytest.mark.parametrize("act", ["relu", "none"])
def test_ln_linear_fp64(B, C, M, S, act):
    torch.manual_seed(2)
    x = torch.randn(B, C, *S) * 2 + 0.5
    g, bt = torch.rand(C) + 0.5, torch.randn(C) * 0.3
    w = torch.randn(M, C, 1) / C ** 0.5
    b = torch.randn(M)
    xn = F.layer_norm(x.double().movedim(1, -1), (C,), g.double(), bt.double(), 1e-5).movedim(-1, 1)
    ref = F.conv1d(xn.flatten(2), w.double(), b.double()).reshape(B, M, *S)
    if act == "relu":
        ref = torch.relu(ref)
    t = [v.to(DEV) for v in (x, g, bt, w, b)]
    y_bx, y_f32 = _both(lambda: PW.ln_linear(t[0], t[1], t[2], 1e-5, t[3], t[4], act))
    # the LayerNorm prologue evaluates rstd*(W·γ(x - pivot) - mean·s) + t: the rounding scales with rstd·|Wγ|·|x - pivot|
    rstd = (x.double().var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    mag = _lin_mag((x.double() - x.double()[:, :1]).abs() * rstd, w.double() * g.double().view(1, -1, 1), b) + _lin_mag(torch.ones_like(x), w.double() * bt.double().view(1, -1, 1))
    _check(f"ln_linear {C}->{M} {act}", y_bx, y_f32, ref, mag, slack=2.0)


@pytest.mark.parametrize("B,C,M,S", [(2, 128, 64, (8, 8, 8)), (1, 512, 256, (4, 4, 4)), (1, 1024, 512, (4, 4, 4))])
def test_gelu_linear_res_fp64(B, C, M, S):
    torch.manual_seed(3)
    z = torch.randn(B, C, *S)
    w = torch.randn(M, C, 1) / C ** 0.5
    b = torch.randn(M)
    res = torch.randn(B, M, *S)
    ref = res.double() + F.conv1d(F.gelu(z.double()).flatten(2), w.double(), b.double()).reshape(B, M, *S)
    t = [v.to(DEV) for v in (z, w, b, res)]
    y_bx, y_f32 = _both(lambda: PW.act_linear_res(t[0], t[1], t[2], t[3], "gelu"))
    # fast_erf carries 1.5e-7 absolute per element of the operand in BOTH kernels: part of the scale
    mag = _lin_mag(F.gelu(z.double()).abs() + 1.0, w, b, res)
    _check(f"gelu_linear_res {C}->{M}", y_bx, y_f32, ref, mag)


@pytest.mark.parametrize("B,C1,C2,M,S", [(2, 64, 64, 64, (8, 8, 8)), (1, 256, 256, 256, (4, 4, 4)), (1, 32, 96, 64, (8, 4, 4))])
def test_cat_linear_fp64(B, C1, C2, M, S):
    torch.manual_seed(4)
    x1, x2 = torch.randn(B, C1, *S), torch.randn(B, C2, *S)
    w = torch.randn(M, C1 + C2, 1) / (C1 + C2) ** 0.5
    ref = F.conv1d(torch.cat([x1, x2], 1).double().flatten(2), w.double()).reshape(B, M, *S)
    t = [v.to(DEV) for v in (x1, x2, w)]
    y_bx, y_f32 = _both(lambda: PW.cat_linear(t[0], t[1], t[2]))
    _check(f"cat_linear {C1}+{C2}->{M}", y_bx, y_f32, ref, _lin_mag(torch.cat([x1, x2], 1), w))


@pytest.mark.parametrize("B,C,O,S", [(2, 32, 64, (16, 16, 16)), (1, 64, 128, (8, 8, 8)), (1, 256, 512, (4, 4, 4)), (1, 32, 48, (4, 8, 12))])
def test_conv_k2s2_fp64(B, C, O, S):
    """Conv3d(k2, s2) forward (space-to-depth loader) and its input gradient (depth-to-space epilogue + skip gradient)"""
    torch.manual_seed(5)
    x = torch.randn(B, C, *S)
    w = torch.randn(O, C, 2, 2, 2) / (8 * C) ** 0.5
    b = torch.randn(O)
    ref = F.conv3d(x.double(), w.double(), b.double(), stride=2)
    xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV), b.to(DEV)
    y_bx, y_f32 = _both(lambda: PW.ConvK2S2Fn.apply(xd, wd, bd))
    _check(f"conv_k2s2 {C}->{O}", y_bx, y_f32, ref, F.conv3d(x.double().abs(), w.double().abs(), b.double().abs(), stride=2))
    gy = torch.randn_like(ref, dtype=torch.float32)
    gref = torch.nn.grad.conv3d_input(x.shape, w.double(), gy.double(), stride=2)
    gd = gy.to(DEV)

    def run():
        (g,) = torch.autograd.grad(PW.ConvK2S2Fn.apply(xd, wd, bd), xd, gd)
        return g
    g_bx, g_f32 = _both(run)
    _check(f"conv_k2s2 dgrad {O}->{C}", g_bx, g_f32, gref, torch.nn.grad.conv3d_input(x.shape, w.double().abs(), gy.double().abs(), stride=2))


@pytest.mark.parametrize("B,C,O,S", [(2, 64, 32, (8, 8, 8)), (1, 128, 64, (4, 4, 4)), (1, 512, 256, (2, 2, 4)), (1, 64, 24, (2, 4, 6))])
def test_tconv_k2s2_fp64(B, C, O, S):
    """ConvTranspose3d(k2, s2) forward (depth-to-space epilogue) and its input gradient (space-to-depth loader)"""
    torch.manual_seed(6)
    x = torch.randn(B, C, *S)
    w = torch.randn(C, O, 2, 2, 2) / C ** 0.5
    b = torch.randn(O)
    ref = F.conv_transpose3d(x.double(), w.double(), b.double(), stride=2)
    xd, wd, bd = x.to(DEV).requires_grad_(True), w.to(DEV), b.to(DEV)
    y_bx, y_f32 = _both(lambda: PW.TConvK2S2Fn.apply(xd, wd, bd))
    _check(f"tconv_k2s2 {C}->{O}", y_bx, y_f32, ref, F.conv_transpose3d(x.double().abs(), w.double().abs(), b.double().abs(), stride=2))
    gy = torch.randn_like(ref, dtype=torch.float32)
    gref = F.conv3d(gy.double(), w.double(), stride=2)
    gd = gy.to(DEV)

    def run():
        (g,) = torch.autograd.grad(PW.TConvK2S2Fn.apply(xd, wd, bd), xd, gd)
        return g
    g_bx, g_f32 = _both(run)
    _check(f"tconv_k2s2 dgrad {O}->{C}", g_bx, g_f32, gref, F.conv3d(gy.double().abs(), w.double().abs(), stride=2))


def test_bx_is_the_default_path_and_bitwise_deterministic():
    torch.manual_seed(7)
    x = torch.randn(2, 128, 8, 8, 8, device=DEV)
    w = torch.randn(128, 128, 1, device=DEV) / 128 ** 0.5
    assert _native.lib().fz_gemm_bx_enable(-1) == 1
    a = PW.linear_cf(x, w, None)
    b = PW.linear_cf(x, w, None)
    assert torch.equal(a, b)


@pytest.mark.parametrize("B,S", [(2, (16, 16, 32)), (1, (8, 12, 64))])
def test_stem_conv_k3_fp64(B, S):
    """The stem Conv3d(4 -> 32, k3, p1) forward and its weight gradient (csrc/conv3.hip, split-bf16 forms) against float64,
    next to the fp32-MFMA kernels."""
    torch.manual_seed(8)
    x = torch.randn(B, 4, *S)
    w = torch.randn(32, 4, 3, 3, 3) / 108 ** 0.5
    ref = F.conv3d(x.double(), w.double(), padding=1)
    mag = F.conv3d(x.double().abs(), w.double().abs(), padding=1)
    xd, wd = x.to(DEV), w.to(DEV).requires_grad_(True)
    y_bx, y_f32 = _both(lambda: PW.ConvK3Fn.apply(xd, wd, None))
    _check("conv_k3 4->32", y_bx, y_f32, ref, mag)
    gy = torch.randn(B, 32, *S)
    gref = torch.nn.grad.conv3d_weight(x.double(), w.shape, gy.double(), padding=1)
    gmag = torch.nn.grad.conv3d_weight(x.double().abs(), w.shape, gy.double().abs(), padding=1)
    gd = gy.to(DEV)

    def run():
        (g,) = torch.autograd.grad(PW.ConvK3Fn.apply(xd, wd, None), wd, gd)
        return g
    g_bx, g_f32 = _both(run)
    _check("conv_k3 wgrad 32x108", g_bx, g_f32, gref, gmag, slack=4.0)   # sum over B·V = 8-16 K voxels in tile / chunk order


@pytest.mark.parametrize("case", ["x*2^60", "x*2^-60", "w*2^60, x*2^-60", "w*2^-40, x*2^-40", "mixed 1e-20..1e20"])
def test_split_bf16_dynamic_range_fp64(case):
    """Range of the three-level split (VERDICT r3 next-4 iii).  bf16 has fp32's exponent range, so scaling an operand by
    2^±60 must leave the RELATIVE error (in units of Σ|a||b|) where it was: the lower levels are 2^-8 / 2^-16 of the value
    and stay normal numbers down to |x| ~ 2^-110.  A tensor whose elements span 1e-20 .. 1e20 must err by fp32 roundings of
    its LARGEST products (the split is per element, not per tensor: a small element next to a large one keeps its own three
    levels).  Against float64, next to the fp32-MFMA kernel on the same inputs; plain linear, its input gradient and the
    LayerNorm-prologue form (whose statistics see the same range)."""
    torch.manual_seed(11)
    B, Cin, Cout, S = 2, 128, 64, (8, 8, 8)
    x = torch.randn(B, Cin, *S)
    w = torch.randn(Cout, Cin, 1) / Cin ** 0.5
    if case == "x*2^60":
        x = x * 2.0 ** 60
    elif case == "x*2^-60":
        x = x * 2.0 ** -60
    elif case == "w*2^60, x*2^-60":
        w, x = w * 2.0 ** 60, x * 2.0 ** -60
    elif case == "w*2^-40, x*2^-40":
        w, x = w * 2.0 ** -40, x * 2.0 ** -40       # products ~2^-80, third-level products ~2^-96: still normal
    else:
        x = x.sign() * 10.0 ** (torch.rand_like(x) * 40 - 20)
    ref = F.conv1d(x.double().flatten(2), w.double()).reshape(B, Cout, *S)
    xd, wd = x.to(DEV), w.to(DEV)
    y_bx, y_f32 = _both(lambda: PW.linear_cf(xd, wd, None))
    assert torch.isfinite(y_bx).all() and torch.isfinite(y_f32).all()
    _check(f"range[{case}] linear", y_bx, y_f32, ref, _lin_mag(x, w))
    # input gradient with the incoming gradient carrying the same range
    gy = torch.randn(B, Cout, *S) * float(x.abs().median())
    gref = torch.einsum("oc,bov->bcv", w[:, :, 0].double(), gy.double().flatten(2)).reshape(B, Cin, *S)
    xg, gd = xd.clone().requires_grad_(True), gy.to(DEV)

    def run():
        (g,) = torch.autograd.grad(PW.linear_cf(xg, wd, None), xg, gd)
        return g
    g_bx, g_f32 = _both(run)
    mag = torch.einsum("oc,bov->bcv", w[:, :, 0].double().abs(), gy.double().abs().flatten(2)).reshape(B, Cin, *S)
    _check(f"range[{case}] linear dgrad", g_bx, g_f32, gref, mag)
