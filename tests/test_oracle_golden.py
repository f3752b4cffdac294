"""Pin the CPU oracle (oracle/cpu_ref.py) against vectors produced by the reference itself
(tools/make_goldens.py -> tests/golden/*.npz).  CPU only."""
import pytest
import torch

from oracle import cpu_ref as O

G1_CASES = {
    "a": ((1, 16, 16, 16, 16), dict(head_dim=8, patch_size=8)),
    "b": ((1, 8, 8, 8, 8), dict(head_dim=4, patch_size=4, shifts=[None, 1, 2, 3])),
    "c": ((1, 8, 8, 16, 4), dict(head_dim=8, patch_size=(4, 8, 2))),
    "d": ((2, 16, 8, 8, 8), dict(num_heads=8, patch_size=4)),
    "e": ((1, 8, 8, 8, 8), dict(head_dim=8, patch_size=4, shifts=[None, (1, 2, 3), 2])),
}


@pytest.mark.parametrize("name", sorted(G1_CASES))
def test_swm_bit_exact(golden, name):
    g = golden("g1_swmatricize").case(name)
    shape, kw = G1_CASES[name]
    n = 1
    for s in shape:
        n *= s
    x = torch.arange(n, dtype=torch.float32).reshape(shape)
    y = O.swm_forward(x, **kw)
    assert torch.equal(y.to(torch.int32), g["y"])
    assert list(y.shape[1:]) == g["output_size"][1:].tolist()
    z = O.swm_inverse(y, shape[1], shape[2:], **kw)
    nw = len(kw.get("shifts", [0, 1]))
    if nw in (1, 2, 4):
        assert torch.equal(z, g["z"])
    else:
        assert torch.allclose(z, g["z"], rtol=2e-7, atol=0)
    if "yr" in g:
        zr = O.swm_inverse(g["yr"], shape[1], shape[2:], **kw)
        if nw in (1, 2, 4):
            assert torch.equal(zr, g["zr"])
        else:
            assert torch.allclose(zr, g["zr"], rtol=3e-7, atol=1e-7)


NMF_CASES = {
    "cfg1_mu_r2_t5": dict(num_iters=5, solver="mu"),
    "cfg2_hals_r1_t5": dict(num_iters=5, solver="hals"),
    "hals_r2_t10_8x512": dict(num_iters=10, solver="hals"),
    "hals_r2_t5_g1": dict(num_iters=5, solver="hals", num_grad_steps=1),
    "mu_r2_t5_g2": dict(num_iters=5, solver="mu", num_grad_steps=2),
    "hals_r1_t5_g1": dict(num_iters=5, solver="hals", num_grad_steps=1),
    "test_nmf_shape": dict(num_iters=5, solver="hals"),
    "rank_auto": dict(num_iters=5, solver="hals"),
    "heads8_m4_n64": dict(num_iters=5, solver="hals"),
}
for _R in (1, 2, 3):
    for _T in (5, 10):
        NMF_CASES[f"hals_r{_R}_t{_T}"] = dict(num_iters=_T, solver="hals")
        NMF_CASES[f"mu_r{_R}_t{_T}"] = dict(num_iters=_T, solver="mu")


@pytest.mark.parametrize("name", sorted(NMF_CASES))
def test_nmf_forward_backward(golden, name):
    g = golden("g2_nmf").case(name)
    kw = NMF_CASES[name]
    x = g["x"].clone().requires_grad_(True)
    u, v = O.nmf_decompose(x, g["u0"], g["v0"], **kw)
    y = O.nmf_forward(x, g["u0"], g["v0"], **kw)
    tol = dict(rtol=1e-5, atol=1e-6)
    assert torch.allclose(u, g["u"], **tol)
    assert torch.allclose(v, g["v"], **tol)
    assert torch.allclose(y, g["y"], **tol)
    (gx,) = torch.autograd.grad(y, x, g["gy"])
    scale = g["gx"].abs().max().item() + 1e-12
    assert (gx - g["gx"]).abs().max().item() <= 2e-5 * scale + 1e-6
    # hand-derived reverse sweep (what the HIP backward implements)
    gxm = O.nmf_backward(g["x"], g["u0"], g["v0"], g["gy"], **kw)
    assert (gxm - g["gx"]).abs().max().item() <= 5e-5 * scale + 1e-6
    assert torch.allclose(O.relative_error(g["x"], u @ v.mT), g["loss"], rtol=1e-5, atol=1e-7)


def test_nmf_backward_float64_exact():
    """Appendix-A sweep equals autograd to ~1e-14 in float64 (all solvers / ranks)."""
    torch.manual_seed(3)
    for solver in ("mu", "hals"):
        for R in (1, 2, 3, 4):
            x = torch.rand(3, 5, 8, 64, dtype=torch.float64)
            x[0, 0].zero_()
            u0 = torch.rand(8, R, dtype=torch.float64)
            v0 = torch.rand(64, R, dtype=torch.float64)
            gy = torch.rand_like(x)
            for G in (None, 2):
                xr = x.clone().requires_grad_(True)
                y = O.nmf_forward(xr, u0, v0, 5, solver, G)
                (ga,) = torch.autograd.grad(y, xr, gy)
                gm = O.nmf_backward(x, u0, v0, gy, 5, solver, G)
                assert (ga - gm).abs().max().item() < 1e-12 * (1 + ga.abs().max().item()), (solver, R, G)


BLOCK_CFG = {
    "hals_r1": dict(reshape=dict(head_dim=8, patch_size=4), num_iters=5, solver="hals"),
    "mu_r2": dict(reshape=dict(head_dim=8, patch_size=4), num_iters=3, solver="mu"),
}


@pytest.mark.parametrize("name", sorted(BLOCK_CFG))
def test_block(golden, name):
    g = golden("g5_block").case(name)
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd:")}
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if not k.endswith(("u0", "v0"))}
    full = dict(sd)
    full.update(params)
    x = g["x"].clone().requires_grad_(True)
    y = O.factorizer_block(x, full, "", BLOCK_CFG[name])
    assert torch.allclose(y, g["y"], rtol=1e-4, atol=1e-5)
    names = sorted(params)
    grads = torch.autograd.grad(y, [x] + [params[k] for k in names], g["gy"])
    assert torch.allclose(grads[0], g["gx"], rtol=1e-3, atol=2e-4)
    for k, gr in zip(names, grads[1:]):
        ref = g["grad:" + k]
        assert (gr - ref).abs().max().item() <= 1e-3 * (ref.abs().max().item() + 1e-6), k


MODEL_CFG = dict(widths=(8, 16, 32), strides=(1, 2, 2), reshape=dict(head_dim=8, patch_size=4),
                 num_iters=5, solver="hals")


def test_model(golden):
    g = golden("g6_model")
    sd = g.case("sd")
    x = g["x"].clone().requires_grad_(True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if not k.endswith(("u0", "v0"))}
    full = dict(sd)
    full.update(params)
    y = O.factorizer_forward(x, full, MODEL_CFG)
    assert torch.allclose(y, g["y"], rtol=1e-4, atol=1e-5)
    names = sorted(params)
    grads = torch.autograd.grad(y, [x] + [params[k] for k in names], g["gy"])
    assert torch.allclose(grads[0], g["gx"], rtol=1e-3, atol=1e-4)
    for k, gr in zip(names, grads[1:]):
        ref = g["grad:" + k]
        assert (gr - ref).abs().max().item() <= 2e-3 * (ref.abs().max().item() + 1e-6), k


LAYER_FUNCS = {
    "conv_k2s2": lambda x, sd: O.conv_k2s2(x, sd["weight"], sd["bias"]),
    "tconv_k2s2": lambda x, sd: O.tconv_k2s2(x, sd["weight"], sd["bias"]),
    "conv_k3": lambda x, sd: O.conv_k3(x, sd["weight"]),
    "conv_k1": lambda x, sd: O.conv_k1(x, sd["weight"], sd["bias"]),
    "linear": lambda x, sd: O.linear_cf(x, sd["linear.weight"], sd["linear.bias"]),
    "linear_nobias": lambda x, sd: O.linear_cf(x, sd["linear.weight"]),
    "layernorm": lambda x, sd: O.layernorm_cf(x, sd["norm.weight"], sd["norm.bias"]),
    "mlp": lambda x, sd: O.mlp_cf(x, sd["block.0.linear.weight"], sd["block.0.linear.bias"],
                                 sd["block.3.linear.weight"], sd["block.3.linear.bias"]),
    "posembed": lambda x, sd: x + sd["pos"],
}


@pytest.mark.parametrize("name", sorted(LAYER_FUNCS))
def test_layers(golden, name):
    g = golden("g7_layers").case(name)
    sd = {k[3:]: v.clone().requires_grad_(True) for k, v in g.items() if k.startswith("sd:")}
    x = g["x"].clone().requires_grad_(True)
    y = LAYER_FUNCS[name](x, sd)
    assert torch.allclose(y, g["y"], rtol=1e-5, atol=1e-5)
    names = sorted(sd)
    grads = torch.autograd.grad(y, [x] + [sd[k] for k in names], g["gy"])
    assert torch.allclose(grads[0], g["gx"], rtol=1e-4, atol=1e-5)
    for k, gr in zip(names, grads[1:]):
        assert torch.allclose(gr, g["grad:" + k], rtol=1e-4, atol=1e-4), k
