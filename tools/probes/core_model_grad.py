"""The fused core of encoder block 0 with the tensors it REALLY sees in the five-stage model at 32^3 / patch 2: its input t and the
gradient g_a that arrives at its output (both taken from the float64 oracle run of the whole model, rounded to fp32).  Device vs
fp64 oracle next to fp32 oracle vs fp64 oracle."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from torch import nn
import factorizer_amd as ft
from factorizer_amd import functional as Fn
from oracle import cpu_ref as O

S, patch, B = (32, 32, 32), 2, 1
if len(sys.argv) > 2: S, patch = (int(sys.argv[1]),) * 3, int(sys.argv[2])
W = (32, 64, 128, 256, 512)
torch.manual_seed(3)
model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=S, encoder_depth=(1,) * 5, encoder_width=W, strides=(1, 2, 2, 2, 2),
                      decoder_depth=(1,) * 4, norm=ft.LayerNorm, reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": patch}),
                      act=nn.ReLU, factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)
sd = {k: (v.double() if v.is_floating_point() else v) for k, v in model.state_dict().items()}
cfg = dict(widths=W, strides=(1, 2, 2, 2, 2), reshape=dict(head_dim=8, patch_size=patch), num_iters=5, solver="hals")
x = torch.rand(B, 4, *S).double(); gy = torch.randn(B, 3, *S).double()
cap = {}
orig_fwd, orig_inv = O.swm_forward, O.swm_inverse


def fwd(t, **kw):
    if "t" not in cap: cap["t"] = t
    return orig_fwd(t, **kw)


def inv(m, C, spatial, **kw):
    a = orig_inv(m, C, spatial, **kw)
    if "a" not in cap: a.retain_grad(); cap["a"] = a
    return a


O.swm_forward, O.swm_inverse = fwd, inv
prm = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith(("u0", "v0"))}
full = dict(sd); full.update(prm)
yo = O.factorizer_forward(x, full, cfg)
yo.backward(gy)
O.swm_forward, O.swm_inverse = orig_fwd, orig_inv
t = cap["t"].detach().float(); ga = cap["a"].grad.float()
print("t %s zeros %.3f | g_a max %.3e rms %.3e" % (tuple(t.shape), (t == 0).float().mean(), ga.abs().max(), ga.pow(2).mean().sqrt()))
p = "encoder.blocks.0.block.blocks.0."
u0, v0 = sd[p + "fact.factorize.init.u0"].float(), sd[p + "fact.factorize.init.v0"].float()
rc = dict(head_dim=8, patch_size=patch)
res = {}
for dt in (torch.float32, torch.float64):
    tt = t.to(dt).requires_grad_(True)
    a = O.swm_inverse(O.nmf_forward(O.swm_forward(torch.relu(tt), **rc), u0.to(dt), v0.to(dt), 5, "hals", None), 32, S, **rc)
    (g,) = torch.autograd.grad(a, tt, ga.to(dt))
    res[dt] = g
geo = Fn.Geometry(32, S, 8, (patch,) * 3, [(0, 0, 0), (patch // 2,) * 3])
td = torch.relu(t).cuda().requires_grad_(True)
ad = Fn.FactCoreFn.apply(td, u0.cuda(), v0.cuda(), geo, 5, 5, "hals", 1e-16, False)
(gd,) = torch.autograd.grad(ad, td, ga.cuda())
r64 = res[torch.float64]; sc = r64.abs().max().item()
gate = (t > 0).double()
d = ((gd.double().cpu() - r64) * gate).abs(); e = ((res[torch.float32].double() - r64) * gate).abs()
print("core g_t: dev max %.2e rms %.2e | f32 max %.2e rms %.2e | max|g_t| %.3e vs max|g_a| %.3e" % (d.max() / sc, d.pow(2).mean().sqrt() / sc, e.max() / sc, e.pow(2).mean().sqrt() / sc, sc, ga.abs().max()))
