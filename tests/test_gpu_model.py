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


def _device_relu_gates(model, x):
    """Forward of the device model with a pre-hook on every FactorizerBlock: the block input is captured and the block's FIRST
    launch — t = relu(in_proj(LayerNorm1(x))), the very kernel and arguments the fused node uses — is repeated on it, so the
    returned masks [t > 0] are bit for bit the ReLU gates the device run had (execution order: encoder stages, then decoder)."""
    from factorizer_amd.blocks import FactorizerBlock
    inputs, hooks = [], []
    for m in model.modules():
        if isinstance(m, FactorizerBlock):
            hooks.append(m.register_forward_pre_hook(lambda mod, args: inputs.append((mod, args[0].detach()))))
    y = model(x)
    for h in hooks:
        h.remove()
    gates = []
    with torch.no_grad():
        for mod, xin in inputs:
            n1, lin = mod.norm1.norm, mod.fact.in_proj.linear
            t = PW.ln_linear(xin, n1.weight, n1.bias, n1.eps, lin.weight, None, "relu")
            gates.append((t > 0).cpu())
    return y, gates


@pytest.mark.parametrize("S,patch,B", [((64, 64, 64), 4, 2), ((32, 32, 32), 2, 1)])
def test_five_stage_model_vs_oracle_every_gradient(S, patch, B):
    """widths (32, 64, 128, 256, 512) at 64^3 / patch 4 (B = 2) and 32^3 / patch 2 (B = 1): output against the fp32 oracle, every
    parameter gradient against the FLOAT64 oracle, at 1e-4.

    One thing has to be handled, and round 4 found it the hard way (tools/probes/block_model_grad.py): the network has ~2 M ReLU
    pre-activations z = in_proj(LN(x)) (factorizer.py:38-44), and in every draw a handful lie within fp32 rounding of zero
    (|z| < 1e-7 of max|z|; seed 3 at 32^3: z = +8.6e-8 in float64, -5.9e-8 in ATen's fp32).  Two correct fp32 evaluations can put
    such an element on different sides; the gate of ONE element moves the input gradient of its voxel by 1e-3 relative and every
    voxel-sum gradient upstream by ~1e-4 — a discontinuity of the function, not an error of either evaluation.  So the float64
    oracle is run with the ReLU gates the DEVICE had ([t > 0] of the device's own t, obtained bit for bit by repeating the block's
    first launch on the hooked block input): where the device's gate differs from float64's own sign the element must be such a
    tie (|z64| <= 1e-6 max|z64|, asserted, counted), everywhere else nothing changes.  (The VALUE stays float64's own relu(z): a
    negative z let through an open gate would leave NMF's domain, and a row of X whose only non-zero is that element then trips
    the solver's own relu((Xv + eps) / (v'v + eps)) — measured: 2.6e-4 on in_proj.weight from that alone.)  With the ties resolved
    one way, the device must match float64 to 1e-4 with no further allowance."""
    torch.manual_seed(3)
    model = _model(S, patch)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    cfg = dict(widths=WIDTHS, strides=STRIDES, reshape=dict(head_dim=8, patch_size=patch), num_iters=5, solver="hals")
    x = torch.rand(B, 4, *S)
    gy = torch.randn(B, 3, *S)

    model = model.to(DEV)
    n0 = _native.launch_count()
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)     # no composed-ATen branch anywhere in the model
        yd, gates = _device_relu_gates(model, x.to(DEV))
        yd.backward(gy.to(DEV))
    assert _native.launch_count() > n0 and len(gates) == 9

    with torch.no_grad():
        y32 = O.factorizer_forward(x, sd, cfg)
    P.close(f"five-stage model {S} y", yd, y32)

    # float64 oracle with the device's gates (oracle/cpu_ref.py:fact_mixer restated with the gate injected)
    ties, it = [], iter(gates)

    def fact_mixer_gated(xx, sdd, prefix, c):
        C, spatial = xx.shape[1], tuple(xx.shape[2:])
        z = O.linear_cf(xx, sdd[prefix + "in_proj.linear.weight"])
        gate = next(it)
        own = z.detach() > 0
        diff = own != gate
        if diff.any():
            zt = z.detach()[diff].abs()
            assert float(zt.max()) <= 1e-6 * float(z.detach().abs().max()), (prefix, float(zt.max()))   # a tie, not a disagreement
        ties.append(int(diff.sum()))
        t = torch.relu(z).detach() + gate.to(z.dtype) * (z - z.detach())     # value relu(z) (float64's own), derivative = the device's gate
        m = O.swm_forward(t, **c["reshape"])
        m = O.nmf_forward(m, sdd[prefix + "factorize.init.u0"], sdd[prefix + "factorize.init.v0"], c.get("num_iters", 5),
                          c.get("solver", "hals"), c.get("num_grad_steps"))
        a = O.swm_inverse(m, C, spatial, **c["reshape"])
        return O.linear_cf(a, sdd[prefix + "out_proj.linear.weight"], sdd[prefix + "out_proj.linear.bias"])

    prm = {k: v.clone().double().requires_grad_(True) for k, v in sd.items()
           if v.is_floating_point() and not k.endswith(("u0", "v0"))}
    full = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    full.update(prm)
    orig = O.fact_mixer
    O.fact_mixer = fact_mixer_gated
    try:
        y64 = O.factorizer_forward(x.double(), full, cfg)
        g64 = dict(zip(prm.keys(), torch.autograd.grad(y64, list(prm.values()), gy.double())))
    finally:
        O.fact_mixer = orig
    assert len(ties) == 9 and sum(ties) <= 64, ties
    # ... and the PLAIN comparison, no gates injected: float64's own ReLU everywhere.  Where a tie element was gated the other
    # way this differs by the discontinuity described above (measured in round 4: 1.5e-4 .. 3.5e-4 on the four tensors
    # upstream of encoder block 0's ReLU at 32^3, nothing at 64^3), so its bound is 5e-4 — five times looser than the gated
    # comparison below, which stays at 1e-4, but no longer absent: a regression that moves a gradient by 1e-3 through any
    # other mechanism than the gate of a tie element fails here whether or not a tie is involved.
    y64p = O.factorizer_forward(x.double(), full, cfg)
    g64p = dict(zip(prm.keys(), torch.autograd.grad(y64p, list(prm.values()), gy.double())))
    worst_plain = 0.0
    for n, p in model.named_parameters():
        worst_plain = max(worst_plain, P.close(f"five-stage model {S} grad {n} (float64 oracle, its own gates)", p.grad, g64p[n].float(),
                                               rel=5e-4, why="ReLU tie elements gated differently by two correct evaluations (1-2 per run): "
                                                             "measured 1.5e-4 .. 3.5e-4 upstream of the tie") / (g64p[n].abs().max().item() + 1e-30))
    P.note("five_stage_model_gradients_plain", spatial=list(S), worst_distance_to_fp64_own_gates=worst_plain)
    P.note("five_stage_model_relu_ties", spatial=list(S), ties_per_block=ties,
           meaning="ReLU pre-activations within 1e-6 of zero that the device and float64 put on different sides")
    P.close(f"five-stage model {S} y (float64, device gates)", yd, y64.float())
    names = [n for n, _ in model.named_parameters()]
    assert set(names) == set(g64.keys())
    worst = 0.0
    for n, p in model.named_parameters():
        worst = max(worst, P.close(f"five-stage model {S} grad {n} (float64 oracle, device gates)", p.grad, g64[n].float())
                    / (g64[n].abs().max().item() + 1e-30))
    P.note("five_stage_model_gradients", spatial=list(S), tensors=len(names), worst_distance_to_fp64=worst)


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
