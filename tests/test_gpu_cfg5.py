"""-m gpu: BASELINE configs[4] as a WHOLE model — the Swin Factorizer on BraTS-shaped volumes, 160 x 192 x 160, HALS rank 2,
10 iterations, fp32 and bf16 mixed precision (`patch_size=8` does not divide the 10 x 12 x 10 bottleneck — SURVEY.md
headline 5 — so the per-axis patch (5, 6, 5) is used at every stage: 8 x 150 matrices, the generic-patch fused core).
The U-shape bookkeeping under test is unet.py:80-83,149-152 (spatial size threaded through the strides, anisotropic
patches at every stage); the arithmetic is pinned against the CPU oracle on a reduced extent (40 x 48 x 40, three
stages), the full size is checked through size-independent properties."""
import warnings

import pytest
import torch
import torch.nn.functional as F
from torch import nn

import factorizer_amd as ft
from factorizer_amd import _native
from oracle import cpu_ref as O
import parity as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _model(spatial, widths, strides):
    return ft.Factorizer(in_channels=4, out_channels=3, spatial_size=spatial, encoder_depth=(1,) * len(widths),
                         encoder_width=widths, strides=strides, decoder_depth=(1,) * (len(widths) - 1), norm=ft.LayerNorm,
                         reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}), act=nn.ReLU, factorize=ft.NMF,
                         rank=2, num_iters=10, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)


def test_cfg5_whole_model_full_size_fp32_and_bf16():
    """160 x 192 x 160, B = 1: forward + DiceCE + backward in fp32 and under bf16 autocast.  No composed-ATen branch
    (RuntimeWarning -> error), finite, bitwise deterministic on replay, every parameter gradient fp32; the bf16 run's
    gradients point the same way as the fp32 run's."""
    torch.manual_seed(0)
    S = (160, 192, 160)
    model = _model(S, (32, 64, 128, 256, 512), (1, 2, 2, 2, 2)).to(DEV)
    x = torch.rand(1, 4, *S, device=DEV)
    t = (torch.rand(1, 3, *S, device=DEV) > 0.5).float()

    def run(amp):
        model.zero_grad(set_to_none=True)
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)
            if amp:
                with torch.autocast("cuda", dtype=BF):
                    y = model(x)
                    loss = ft.dice_ce_loss(y, t)
            else:
                y = model(x)
                loss = ft.dice_ce_loss(y, t)
            loss.backward()
        return y, loss, {n: p.grad.clone() for n, p in model.named_parameters()}

    n0 = _native.launch_count()
    y, loss, g32 = run(False)
    assert _native.launch_count() > n0
    assert y.shape == (1, 3, *S) and y.dtype == torch.float32 and torch.isfinite(y).all() and torch.isfinite(loss)
    for n, g in g32.items():
        assert g.dtype == torch.float32 and torch.isfinite(g).all(), n
    y2, loss2, g32b = run(False)
    assert torch.equal(y, y2) and torch.equal(loss, loss2)
    for n in g32:
        assert torch.equal(g32[n], g32b[n]), n          # no float atomics anywhere: bitwise replay
    del y2, g32b
    yb, lossb, g16 = run(True)
    assert yb.dtype == BF and torch.isfinite(yb.float()).all() and torch.isfinite(lossb) and lossb.dtype == torch.float32
    yb2, _, g16b = run(True)
    assert torch.equal(yb, yb2)
    worst, worst_name = 1.0, ""
    differs = {n: float((g - g16b[n]).abs().max() / (g.abs().max() + 1e-30)) for n, g in g16.items() if not torch.equal(g, g16b[n])}
    assert not differs, f"bf16 replay is not bitwise reproducible for {len(differs)} tensors: {dict(list(differs.items())[:8])}"
    total = torch.cat([g32[n].flatten() for n in g32]).norm().item()
    small_worst, small_name = 1.0, ""
    for n, g in g16.items():
        assert g.dtype == torch.float32 and torch.isfinite(g).all(), n
        if g32[n].norm() > 1e-6 * max(1.0, g32[n].numel() ** 0.5):
            cos = F.cosine_similarity(g.flatten(), g32[n].flatten(), dim=0).item()
            if g32[n].norm().item() >= 1e-3 * total:
                if cos < worst:
                    worst, worst_name = cos, n
            elif cos < small_worst:
                small_worst, small_name = cos, n
    whole = F.cosine_similarity(torch.cat([g16[n].flatten() for n in g32]), torch.cat([g32[n].flatten() for n in g32]), dim=0).item()
    P.note("cfg5_full_size_bf16_vs_fp32_gradient_cosine", whole_gradient=whole, per_tensor_min=worst, tensor=worst_name,
           per_tensor_min_among_small_tensors=small_worst, small_tensor=small_name, small_means="below 1e-3 of the gradient norm",
           loss_fp32=float(loss.detach()), loss_bf16=float(lossb.detach()))
    assert abs(float(lossb.detach()) - float(loss.detach())) <= 2e-2 * abs(float(loss.detach()))
    # The whole gradient must point the same way.  Per tensor the bar is lower: the gradients behind ten rank-2 HALS sweeps
    # are ill-conditioned (the fp32 arithmetic itself sits 1e-3 .. 3e-2 from float64, test below), and the worst tensor's
    # cosine moves between 0.97 and 0.995 with ANY change of fp32 rounding order (measured: 0.972 with every GEMM on the
    # fp32 MFMA, 0.984 with the split-bf16 products) — it measures the conditioning, not the bf16 path.
    # [r5] The per-tensor bound applies to tensors that carry at least 1e-3 of the gradient's norm.  The deepest levels
    # (decoder.blocks.0, encoder.blocks.3 / 4 at this extent: 10 x 12 x 10 voxels and below) have gradients orders of magnitude
    # smaller, and there the bf16 storage roundings are the signal: two equally accurate bf16 evaluations of the SAME network
    # (round 5's forward fusions on / off: every stage output a few bf16 ulps apart and equally far from the fp32 run, every
    # fp32 gradient cosine 1.0000 between the two forms — tools/probes/cfg5_bf16_fwd_ab.py, cfg5_bf16_cosine.py) give deep
    # gradients with cosine 0.6 - 0.8 to EACH OTHER; against fp32 such a tensor sat at 0.999 in one form and 0.84 in the other.
    # Their minimum is recorded, not asserted.
    assert whole >= 0.99, whole
    assert worst >= 0.95, (worst_name, worst)


def test_cfg5_reduced_extent_whole_model_vs_oracle():
    """40 x 48 x 40, three stages (32, 64, 128), patch (5, 6, 5) at 40x48x40 / 20x24x20 / 10x12x10, HALS R = 2, T = 10:
    output and every parameter gradient against the CPU oracle's whole-model restatement (oracle/cpu_ref.py:
    factorizer_forward, pinned to the reference by the goldens g5 / g6).

    The output is held to 1e-4 of the fp32 oracle.  The GRADIENTS of this configuration are not reproducible to 1e-4 in
    fp32 by anyone: ten rank-2 HALS sweeps (ReLU-gated Gauss-Seidel updates, matrix_factorization.py:214-223) per block
    amplify rounding, and the reference's own fp32 arithmetic (the oracle run in float32) sits 1e-3 .. 3e-2 of max|g| away
    from its float64 evaluation, tensor by tensor.  So each gradient is compared with the FLOAT64 oracle and must be as
    close to it as the fp32 oracle is: bound = max(1e-4, 2 x the fp32 oracle's own distance); both distances recorded."""
    torch.manual_seed(1)
    S = (40, 48, 40)
    widths, strides = (32, 64, 128), (1, 2, 2)
    model = _model(S, widths, strides)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = dict(widths=widths, strides=strides, reshape=dict(head_dim=8, patch_size=(5, 6, 5)), num_iters=10, solver="hals")
    x = torch.rand(1, 4, *S)
    gy = torch.randn(1, 3, *S)

    def oracle(dt):
        prm = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items()
               if v.is_floating_point() and not k.endswith(("u0", "v0"))}
        full = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
        full.update(prm)
        yo = O.factorizer_forward(x.to(dt), full, cfg)
        return yo, dict(zip(prm.keys(), torch.autograd.grad(yo, list(prm.values()), gy.to(dt))))

    y32, g32 = oracle(torch.float32)
    y64, g64 = oracle(torch.float64)
    model = model.to(DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        yd = model(x.to(DEV))
        yd.backward(gy.to(DEV))
    P.close("cfg5 reduced model y", yd, y32)
    worst = 0.0
    for n, p in model.named_parameters():
        scale = g64[n].abs().max().item() + 1e-30
        e32 = (g32[n].double() - g64[n]).abs().max().item() / scale
        ed = (p.grad.double().cpu() - g64[n]).abs().max().item() / scale
        worst = max(worst, ed / max(e32, 5e-5))
        P.close(f"cfg5 reduced model grad {n} (vs fp64 oracle; fp32 oracle is {e32:.1e} away)", p.grad, g64[n].float(),
                rel=max(1e-4, 2.0 * e32),
                why="ten rank-2 HALS sweeps per block amplify fp32 rounding: the reference's own fp32 arithmetic is this far "
                    "from its float64 evaluation; the device must be as close to float64 as the fp32 oracle (x2)")
    P.note("cfg5_reduced_model_worst_device_over_fp32oracle_distance_to_fp64", value=worst)


def test_cfg5_per_gpu_batch_of_four_bf16_step():
    """BASELINE configs[4] at ITS per-GPU batch: B = 4, 160 x 192 x 160, bf16 autocast (29.9 GB peak): one training step —
    forward, DiceCE, backward — with no composed-ATen branch, finite loss and gradients (all fp32), replayed bit for bit.
    (The B = 1 test above pins the arithmetic; this one pins that the launch geometries of the full batch — 4x the matrices
    per launch: 2 359 296 at stage 0, grid sizes beyond 2^21 threads — run and reproduce.)"""
    torch.manual_seed(0)
    S = (160, 192, 160)
    model = _model(S, (32, 64, 128, 256, 512), (1, 2, 2, 2, 2)).to(DEV)
    x = torch.rand(4, 4, *S, device=DEV)
    t = (torch.rand(4, 3, *S, device=DEV) > 0.5).float()

    def run():
        model.zero_grad(set_to_none=True)
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)
            with torch.autocast("cuda", dtype=BF):
                y = model(x)
                loss = ft.dice_ce_loss(y, t)
            loss.backward()
        return loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}

    torch.cuda.reset_peak_memory_stats()
    n0 = _native.launch_count()
    loss, g = run()
    assert _native.launch_count() > n0 and torch.isfinite(loss)
    for n, v in g.items():
        assert v.dtype == torch.float32 and torch.isfinite(v).all(), n
    loss2, g2 = run()
    assert torch.equal(loss, loss2)
    differs = [n for n in g if not torch.equal(g[n], g2[n])]
    assert not differs, differs[:8]
    P.note("cfg5_batch4_bf16_step", loss=float(loss), peak_mem_GB=round(torch.cuda.max_memory_allocated() / 1e9, 1))
