"""The fused core on the tensor it sees INSIDE the five-stage model at 32^3 / patch 2 (t = relu(in_proj(LN(stem(x)))) of encoder
block 0), random incoming gradient: device vs fp64 oracle next to fp32 oracle vs fp64 oracle, overall and for the worst patches."""
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
sd = {k: v.double() for k, v in model.state_dict().items()}
x = torch.rand(B, 4, *S).double()
p = "encoder.blocks.0.block.blocks.0."
h = O.conv_k3(x, sd["stem.weight"])
t = torch.relu(O.linear_cf(O.layernorm_cf(h, sd[p + "norm1.norm.weight"], sd[p + "norm1.norm.bias"]), sd[p + "fact.in_proj.linear.weight"]))
t = t.float()          # the fp32 tensor every evaluation starts from
print("t: zeros %.3f  min nonzero %.2e  max %.2e" % ((t == 0).float().mean(), t[t > 0].min(), t.max()))
ga = torch.randn_like(t)
u0, v0 = sd[p + "fact.factorize.init.u0"].float(), sd[p + "fact.factorize.init.v0"].float()
cfg = dict(head_dim=8, patch_size=patch)
res = {}
for dt in (torch.float32, torch.float64):
    tt = t.to(dt).requires_grad_(True)
    m = O.nmf_forward(O.swm_forward(tt, **cfg), u0.to(dt), v0.to(dt), 5, "hals", None)
    a = O.swm_inverse(m, 32, S, **cfg)
    (g,) = torch.autograd.grad(a, tt, ga.to(dt))
    res[dt] = (a.detach(), g)
geo = Fn.Geometry(32, S, 8, (patch,) * 3, [(0, 0, 0), (patch // 2,) * 3])
td = t.cuda().requires_grad_(True)
ad = Fn.FactCoreFn.apply(td, u0.cuda(), v0.cuda(), geo, 5, 5, "hals", 1e-16, False)
(gd,) = torch.autograd.grad(ad, td, ga.cuda())
for name, dv, k in (("core out", ad, 0), ("core g_t", gd, 1)):
    r64 = res[torch.float64][k]; sc = r64.abs().max().item()
    d = (dv.double().cpu() - r64).abs(); e = (res[torch.float32][k].double() - r64).abs()
    print("%-10s dev max %.2e rms %.2e | f32 max %.2e rms %.2e | max|ref| %.2e" % (name, d.max() / sc, d.pow(2).mean().sqrt() / sc, e.max() / sc, e.pow(2).mean().sqrt() / sc, sc))
    if k == 1:
        idx = d.flatten().topk(5).indices
        for i in idx.tolist():
            c = torch.unravel_index(torch.tensor(i), d.shape)
            c = tuple(int(v) for v in c)
            print("   worst at", c, "dev %.6e f32 %.6e f64 %.6e  t=%.3e" % (dv.cpu()[c], res[torch.float32][1][c], r64[c], t[c]))
