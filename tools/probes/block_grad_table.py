"""One FactorizerBlock: device vs float64 oracle next to fp32 oracle vs float64 oracle, per gradient (max and RMS, relative to
max|g64|), and the same for the fused core alone (FactCoreFn: g_t given g_a).  usage: block_grad_table.py C S patch [B]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from torch import nn
import factorizer_amd as ft
from factorizer_amd import functional as Fn
from oracle import cpu_ref as O

C, S, patch = int(sys.argv[1]), (int(sys.argv[2]),) * 3, int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
torch.manual_seed(0)
blk = ft.FactorizerBlock(channels=C, spatial_size=S, norm=ft.LayerNorm, reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": patch}),
                         act=nn.ReLU, factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)
sd = {k: v.clone() for k, v in blk.state_dict().items()}
x = torch.randn(B, C, *S); gy = torch.randn(B, C, *S)
if len(sys.argv) > 5 and sys.argv[5] == "stem":   # the block input the model produces: a 3x3x3 convolution of uniform noise
    torch.manual_seed(7)
    x = torch.nn.functional.conv3d(torch.rand(B, 4, *S), torch.randn(C, 4, 3, 3, 3) * (1.0 / 108 ** 0.5), padding=1)
    print("block input = conv3(uniform noise): per-voxel channel std min %.2e median %.2e" % (x.std(1).min(), x.std(1).median()))
cfg = dict(reshape=dict(head_dim=8, patch_size=patch), num_iters=5, solver="hals")


def oracle(dt):
    prm = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items() if not k.endswith(("u0", "v0"))}
    full = {k: v.to(dt) for k, v in sd.items()}; full.update(prm)
    xo = x.to(dt).requires_grad_(True)
    yo = O.factorizer_block(xo, full, "", cfg)
    gs = torch.autograd.grad(yo, [xo] + list(prm.values()), gy.to(dt))
    return yo.detach(), dict(zip(["x"] + list(prm.keys()), gs))


y32, g32 = oracle(torch.float32); y64, g64 = oracle(torch.float64)
blk = blk.cuda()
xd = x.cuda().requires_grad_(True)
names = [n for n, _ in blk.named_parameters()]
gs = torch.autograd.grad(blk(xd), [xd] + [p for _, p in blk.named_parameters()], gy.cuda())
gd = dict(zip(["x"] + names, gs))
print(f"block C={C} S={S} patch={patch} B={B}")
print("%-34s %9s %9s %9s %9s" % ("tensor", "dev max", "f32 max", "dev rms", "f32 rms"))
for n in g64:
    sc = g64[n].abs().max().item() + 1e-30
    d = gd[n].double().cpu() - g64[n]; e = g32[n].double() - g64[n]
    print("%-34s %9.2e %9.2e %9.2e %9.2e" % (n, d.abs().max() / sc, e.abs().max() / sc, d.pow(2).mean().sqrt() / sc, e.pow(2).mean().sqrt() / sc))

# the fused core alone on a ReLU'd input
t = torch.relu(torch.randn(B, C, *S)); ga = torch.randn(B, C, *S)
u0, v0 = sd["fact.factorize.init.u0"], sd["fact.factorize.init.v0"]
geo = Fn.Geometry(C, S, 8, (patch,) * 3, [(0, 0, 0), (patch // 2,) * 3])


def core_oracle(dt):
    tt = t.to(dt).requires_grad_(True)
    m = O.swm_forward(tt, **cfg["reshape"])
    m = O.nmf_forward(m, u0.to(dt), v0.to(dt), 5, "hals", None)
    return tt, O.swm_inverse(m, C, S, **cfg["reshape"])


if True:
    res = {}
    for dt in (torch.float32, torch.float64):
        tt, a = core_oracle(dt)
        (g,) = torch.autograd.grad(a, tt, ga.to(dt))
        res[dt] = (a.detach(), g)
    td = t.cuda().requires_grad_(True)
    ad = Fn.FactCoreFn.apply(td, u0.cuda(), v0.cuda(), geo, 5, 5, "hals", 1e-16, False)
    (gdv,) = torch.autograd.grad(ad, td, ga.cuda())
    for name, dv, k in (("core out", ad, 0), ("core g_t", gdv, 1)):
        r64 = res[torch.float64][k]; sc = r64.abs().max().item()
        d = dv.double().cpu() - r64; e = res[torch.float32][k].double() - r64
        print("%-34s %9.2e %9.2e %9.2e %9.2e" % (name, d.abs().max() / sc, e.abs().max() / sc, d.pow(2).mean().sqrt() / sc, e.pow(2).mean().sqrt() / sc))
else:
    print("(oracle has no fact_core entry point)")
