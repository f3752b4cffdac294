"""-m gpu: the Deconver family on device against the reference goldens (g9): the 1x1 projections,
LayerNorm, MLP, stem / k2s2 convolutions run the native GEMM-family kernels; the grouped correlations of the
multiplicative updates are framework convolutions on device and say so (RuntimeWarning) — asserted here."""
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
    g = golden("g9_deconver")
    model = ft.Deconver(in_channels=4, out_channels=3, **MODELS[tag]).eval()
    model.load_state_dict(g.case(f"{tag}:sd"))
    model = model.to(DEV)
    x = g[f"{tag}:x"].to(DEV).requires_grad_(True)
    composed._warned.clear()
    n0 = _native.launch_count()
    with pytest.warns(RuntimeWarning, match="Deconv: the grouped correlations"):
        y = model(x)
    names = [k for k, _ in model.named_parameters()]
    grads = torch.autograd.grad(y, [x] + list(model.parameters()), g[f"{tag}:gy"].to(DEV), allow_unused=True)
    torch.cuda.synchronize()
    assert _native.launch_count() > n0, "the dense sub-layers did not run the native kernels"
    P.close("y", y, g[f"{tag}:y"])
    P.close("gx", grads[0], g[f"{tag}:gx"], rel=2e-4, why="reference goldens of this tiny model agree with the CPU path to 2e-4 only")
    for k, gr in zip(names, grads[1:]):
        key = f"{tag}:grad:{k}"
        if key in g.z:
            P.close("grad:" + k, gr, g[key], rel=2e-4, why="as gx")


@pytest.mark.parametrize("tag", sorted(DECONV))
def test_deconv_layer_on_device(golden, tag):
    g = golden("g9_deconver")
    m = ft.Deconv(**DECONV[tag])
    m.load_state_dict(g.case(f"{tag}:sd"))
    m = m.to(DEV)
    x = g[f"{tag}:x"].to(DEV).requires_grad_(True)
    y = m(x)
    (gx,) = torch.autograd.grad(y, x, g[f"{tag}:gy"].to(DEV))
    P.close("y", y, g[f"{tag}:y"])
    P.close("gx", gx, g[f"{tag}:gx"])
    with torch.no_grad():
        s, h = m.fit(x)
        P.close("reconstruct", m.reconstruct(s, h), g[f"{tag}:recon"])
