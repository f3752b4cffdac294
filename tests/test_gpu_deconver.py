"""-m gpu: the Deconver family on device against the reference goldens (g9): the 1x1 projections,
LayerNorm, MLP, stem / k2s2 convolutions run the native GEMM-family kernels; the grouped correlations of the
multiplicative updates, their input gradients, filter gradients and the lag correlations of the filter update run
csrc/deconv.hip — no framework fallback (a RuntimeWarning would fail the tests)."""
import pytest
import torch

import factorizer_amd as ft
import parity as P
from factorizer_amd import _native, composed
from test_deconver_cpu import DECONV, MODELS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("tag", sorted(MODELS))
def test_deconver_model_on_device(golden, tag):
    import warnings
    g = golden("g9_deconver")
    model = ft.Deconver(in_channels=4, out_channels=3, **MODELS[tag]).eval()
    model.load_state_dict(g.case(f"{tag}:sd"))
    model = model.to(DEV)
    x = g[f"{tag}:x"].to(DEV).requires_grad_(True)
    composed._warned.clear()
    n0 = _native.launch_count()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)     # forward and backward are native end to end
        y = model(x)
        with torch.no_grad():
            y_inf = model(x)                               # inference: update + division fused into the kernel
        names = [k for k, _ in model.named_parameters()]
        gy = g[f"{tag}:gy"].to(DEV)
        grads = torch.autograd.grad(y, [x] + list(model.parameters()), gy, allow_unused=True)
    torch.cuda.synchronize()
    assert _native.launch_count() > n0
    P.close("y", y, g[f"{tag}:y"])
    P.close("y (no_grad: fused update epilogue)", y_inf, g[f"{tag}:y"])
    # Round 2 held the model gradients to 2e-4 "because the reference's CPU evaluation orders agree to 2e-4 only".  An fp64
    # evaluation of the same model (the composed CPU path in float64) says otherwise: the reference's fp32 gradients (the
    # goldens) are within 3e-6 of it, and so is the device.  Every gradient is held to the 1e-4 bound against the goldens,
    # and the distances to the fp64 evaluation are recorded next to each other.
    m64 = ft.Deconver(in_channels=4, out_channels=3, **MODELS[tag]).eval().double()
    m64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in g.case(f"{tag}:sd").items()})
    x64 = g[f"{tag}:x"].double().requires_grad_(True)
    g64 = torch.autograd.grad(m64(x64), [x64] + list(m64.parameters()), g[f"{tag}:gy"].double(), allow_unused=True)
    P.close("gx", grads[0], g[f"{tag}:gx"])
    worst_gold = worst_dev = 0.0
    for k, gr, r64 in zip(["x"] + names, grads, g64):
        key = f"{tag}:gx" if k == "x" else f"{tag}:grad:{k}"
        if key not in g.z:
            continue
        if k != "x":
            P.close("grad:" + k, gr, g[key])
        scale = r64.abs().max().item() + 1e-30
        worst_gold = max(worst_gold, (g[key].double() - r64).abs().max().item() / scale)
        worst_dev = max(worst_dev, (gr.double().cpu() - r64).abs().max().item() / scale)
    P.note(f"deconver {tag}: gradient distance to the fp64 evaluation", reference_fp32_goldens=worst_gold, device=worst_dev)
    assert worst_dev <= max(1e-5, 4 * worst_gold), (worst_dev, worst_gold)


@pytest.mark.parametrize("nd,k,G,Ci,Co,S,batched", [
    (3, 3, 8, 16, 4, (6, 9, 70), False), (3, 3, 2, 3, 5, (4, 4, 64), True), (3, 5, 1, 2, 16, (5, 6, 20), False),
    (3, 7, 3, 1, 1, (9, 5, 33), False), (2, 3, 5, 8, 4, (12, 12), False), (2, 5, 2, 4, 2, (10, 70), True),
    (2, 7, 4, 2, 8, (9, 31), False)])
def test_grouped_correlation_kernel(nd, k, G, Ci, Co, S, batched):
    """csrc/deconv.hip against F.conv{2,3}d: forward, the fused multiplicative-update epilogue, the input gradient
    (native, through the adjoint filters) and the filter gradient; ragged tiles, per-sample filters."""
    import torch.nn.functional as F
    from factorizer_amd import functional as Fn
    torch.manual_seed(nd * 100 + k)
    B = 2
    x = torch.rand(B, G * Ci, *S)
    w = torch.rand(B if batched else 1, G, Co, Ci, *([k] * nd)) / (Ci * k ** nd) ** 0.5
    conv = F.conv3d if nd == 3 else F.conv2d
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    if batched:
        yc = conv(xc.reshape(1, B * G * Ci, *S), wc.reshape(B * G * Co, Ci, *([k] * nd)), padding=k // 2, groups=B * G)
        yc = yc.reshape(B, G * Co, *S)
    else:
        yc = conv(xc, wc.reshape(G * Co, Ci, *([k] * nd)), padding=k // 2, groups=G)
    gy = torch.rand_like(yc)
    gxc, gwc = torch.autograd.grad(yc, [xc, wc], gy)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    assert Fn.gcorr_supported(xd, wd)
    n0 = _native.launch_count()
    yd = Fn.gcorr(xd, wd, 0.0)
    gxd, gwd = torch.autograd.grad(yd, [xd, wd], gy.to(DEV))
    assert _native.launch_count() - n0 == 4          # forward, input gradient, filter gradient (sweep + finish)
    P.close("y", yd, yc)
    P.close("gx", gxd, gxc)
    P.close("gw", gwd, gwc)
    a, b = torch.rand_like(yc), torch.rand_like(yc)
    fused = Fn.gcorr_mu_update(a.to(DEV), b.to(DEV), xd.detach(), wd.detach(), 1e-3)
    P.close("fused a*b/(corr+eps)", fused, a * b / (yc.detach() + 1e-3))


@pytest.mark.parametrize("ks,G,Ci,Co,S,batched", [
    ((5, 3, 3), 1, 8, 8, (6, 6, 8), False),        # the reference test suite's anisotropic kernel (tests/test_deconver.py)
    ((3, 5, 7), 2, 3, 5, (5, 9, 70), True),        # every extent different, ragged tiles, per-sample filters
    ((3, 3, 3), 2, 4, 20, (4, 6, 36), False),      # more than 16 output channels per group: three channel blocks of eight
    ((1, 3, 5), 3, 24, 2, (1, 10, 66), False),     # > 16 INPUT channels of the adjoint (its Co), depth-1 volume
    ((7, 1), 2, 2, 9, (12, 40), False)])           # 2-D anisotropic
def test_grouped_correlation_any_kernel_extent_and_channel_count(ks, G, Ci, Co, S, batched):
    """The run-time-extent kernels of csrc/deconv.hip (gcorr_any_kernel / gcorr_wgrad_any_kernel: any odd extent <= 7 per axis,
    any channel count per group) against F.conv{2,3}d: forward, fused update epilogue, input gradient, filter gradient —
    shapes the compile-time instantiations do not cover and that took framework convolutions on device until round 6."""
    import torch.nn.functional as F
    from factorizer_amd import functional as Fn
    nd = len(ks)
    torch.manual_seed(sum(ks) * 7 + Co)
    B = 2
    x = torch.rand(B, G * Ci, *S)
    w = torch.rand(B if batched else 1, G, Co, Ci, *ks) / (Ci * float(torch.tensor(ks).prod())) ** 0.5
    conv = F.conv3d if nd == 3 else F.conv2d
    pad = tuple(k // 2 for k in ks)
    xc, wc = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    if batched:
        yc = conv(xc.reshape(1, B * G * Ci, *S), wc.reshape(B * G * Co, Ci, *ks), padding=pad, groups=B * G).reshape(B, G * Co, *S)
    else:
        yc = conv(xc, wc.reshape(G * Co, Ci, *ks), padding=pad, groups=G)
    gy = torch.rand_like(yc)
    gxc, gwc = torch.autograd.grad(yc, [xc, wc], gy)
    xd, wd = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    assert Fn.gcorr_supported(xd, wd)
    n0 = _native.launch_count()
    yd = Fn.gcorr(xd, wd, 0.0)
    gxd, gwd = torch.autograd.grad(yd, [xd, wd], gy.to(DEV))
    assert _native.launch_count() - n0 == 4
    P.close("y", yd, yc)
    P.close("gx", gxd, gxc)
    P.close("gw", gwd, gwc)
    gwd2 = torch.autograd.grad(Fn.gcorr(xd, wd, 0.0), wd, gy.to(DEV))[0]
    assert torch.equal(gwd, gwd2)                    # fixed-order reduction: bit-identical when repeated
    a, b = torch.rand_like(yc), torch.rand_like(yc)
    fused = Fn.gcorr_mu_update(a.to(DEV), b.to(DEV), xd.detach(), wd.detach(), 1e-3)
    P.close("fused a*b/(corr+eps)", fused, a * b / (yc.detach() + 1e-3))


@pytest.mark.parametrize("tag", sorted(DECONV))
def test_deconv_layer_on_device(golden, tag):
    import warnings
    g = golden("g9_deconver")
    m = ft.Deconv(**DECONV[tag])
    m.load_state_dict(g.case(f"{tag}:sd"))
    m = m.to(DEV)
    x = g[f"{tag}:x"].to(DEV).requires_grad_(True)
    composed._warned.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)     # every case of g9 — the anisotropic (5, 3, 3) one included — is native
        y = m(x)
    (gx,) = torch.autograd.grad(y, x, g[f"{tag}:gy"].to(DEV))
    P.close("y", y, g[f"{tag}:y"])
    P.close("gx", gx, g[f"{tag}:gx"])
    with torch.no_grad():
        s, h = m.fit(x)
        P.close("reconstruct", m.reconstruct(s, h), g[f"{tag}:recon"])


@pytest.mark.parametrize("nd,k,G,K,C,S", [(3, 3, 2, 3, 4, (5, 6, 70)), (3, 5, 1, 2, 3, (4, 9, 20)), (2, 3, 4, 2, 2, (12, 66)),
                                          (2, 7, 1, 1, 5, (9, 31)), (3, 7, 2, 2, 1, (3, 5, 9))])
def test_lag_correlation_kernel(nd, k, G, K, C, S):
    """Fn.lag_corr (fz_gcorr_wgrad as a forward op + two grouped correlations as its gradients) against the framework
    restatement of deconvolution.py:43-50,150-156 on CPU, values and both gradients."""
    from factorizer_amd import functional as Fn
    from factorizer_amd.deconver import _lag_corr
    torch.manual_seed(7 * nd + k)
    B = 2
    s, x = torch.rand(B, G * K, *S), torch.rand(B, G * C, *S)
    sc, xc = s.clone().requires_grad_(True), x.clone().requires_grad_(True)
    Lc = _lag_corr(sc, xc, G, (k // 2,) * nd)
    gL = torch.rand_like(Lc)
    gsc, gxc = torch.autograd.grad(Lc, [sc, xc], gL)
    sd, xd = s.to(DEV).requires_grad_(True), x.to(DEV).requires_grad_(True)
    assert Fn.lag_corr_supported(sd, xd, G, (k,) * nd)
    Ld = Fn.lag_corr(sd, xd, G, (k,) * nd)
    gsd, gxd = torch.autograd.grad(Ld, [sd, xd], gL.to(DEV))
    P.close("L", Ld, Lc)
    P.close("gs", gsd, gsc)
    P.close("gx", gxd, gxc)


def test_gcorr_wgrad_is_deterministic_and_chunked():
    """several voxel tiles per workgroup chunk (the per-chunk register sums) and run-to-run bit equality"""
    from factorizer_amd import functional as Fn
    torch.manual_seed(3)
    B, G, Ci, Co, S = 2, 8, 16, 2, (8, 8, 130)            # 256 (sample, group, channel) triples -> 4 chunks of 3 tiles
    x, gy = torch.rand(B, G * Ci, *S), torch.rand(B, G * Co, *S)
    ref = torch.nn.grad.conv3d_weight(x.double(), (G * Co, Ci, 3, 3, 3), gy.double(), padding=1, groups=G).reshape(1, G, Co, Ci, 3, 3, 3)
    a = Fn._gcorr_wgrad_raw(x.to(DEV), gy.to(DEV), (1, G, Co, Ci, 3, 3, 3))
    b = Fn._gcorr_wgrad_raw(x.to(DEV), gy.to(DEV), (1, G, Co, Ci, 3, 3, 3))
    assert torch.equal(a, b)
    P.close("gw, 16640 voxels per filter tap", a, ref.float())
