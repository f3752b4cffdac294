"""Per-tensor distances of the five-stage model's gradients: device vs fp64 oracle next to fp32 oracle vs fp64 oracle
(max-abs and RMS, each relative to max|g64|).  Diagnostic for tests/test_gpu_model.py."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from torch import nn
import factorizer_amd as ft
from oracle import cpu_ref as O

S, patch, B = (64, 64, 64), 4, 2
if len(sys.argv) > 1:
    S = (int(sys.argv[1]),) * 3; patch = int(sys.argv[2]); B = int(sys.argv[3])
WIDTHS, STRIDES = (32, 64, 128, 256, 512), (1, 2, 2, 2, 2)
torch.manual_seed(3)
model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=S, encoder_depth=(1,) * 5, encoder_width=WIDTHS, strides=STRIDES,
                      decoder_depth=(1,) * 4, norm=ft.LayerNorm, reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": patch}),
                      act=nn.ReLU, factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)
sd = {k: v.clone() for k, v in model.state_dict().items()}
cfg = dict(widths=WIDTHS, strides=STRIDES, reshape=dict(head_dim=8, patch_size=patch), num_iters=5, solver="hals")
x = torch.rand(B, 4, *S); gy = torch.randn(B, 3, *S)
def oracle(dt):
    prm = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith(("u0", "v0"))}
    full = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}; full.update(prm)
    yo = O.factorizer_forward(x.to(dt), full, cfg)
    return yo.detach(), dict(zip(prm.keys(), torch.autograd.grad(yo, list(prm.values()), gy.to(dt))))
y32, g32 = oracle(torch.float32); y64, g64 = oracle(torch.float64)
model = model.cuda()
from factorizer_amd import _native as _N
_prod = {"split": _N.PRODUCTS_SPLIT_BF16, "fp32": _N.PRODUCTS_FP32_MFMA}.get(os.environ.get("FZ_TABLE_PRODUCTS", ""), _N.PRODUCTS_DEFAULT)
print("products:", _prod)
with _N.use_products(_prod):
    yd = model(x.cuda()); yd.backward(gy.cuda())
print("y: dev-64 %.2e  32-64 %.2e" % ((yd.double().cpu() - y64).abs().max() / y64.abs().max(), (y32.double() - y64).abs().max() / y64.abs().max()))
rows = []
for n, p in model.named_parameters():
    sc = g64[n].abs().max().item() + 1e-30
    d = (p.grad.double().cpu() - g64[n]); e = (g32[n].double() - g64[n])
    rows.append((n, d.abs().max().item() / sc, e.abs().max().item() / sc, d.pow(2).mean().sqrt().item() / sc, e.pow(2).mean().sqrt().item() / sc, p.numel()))
rows.sort(key=lambda r: -r[1] / max(r[2], 1e-7))
print("%-70s %9s %9s %9s %9s %8s" % ("tensor", "dev max", "f32 max", "dev rms", "f32 rms", "numel"))
for r in rows: print("%-70s %9.2e %9.2e %9.2e %9.2e %8d" % r)
