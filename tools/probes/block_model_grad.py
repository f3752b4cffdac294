"""Encoder block 0 with the tensors it REALLY sees in the five-stage model at 32^3 / patch 2: its input x (stem output) and the
gradient gy arriving at its output, both from the float64 oracle run of the whole model (rounded to fp32).  Device block vs
fp64 oracle block next to fp32 oracle block, per gradient."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from torch import nn
import factorizer_amd as ft
from oracle import cpu_ref as O

S, patch, B = (32, 32, 32), 2, 1
W = (32, 64, 128, 256, 512)
torch.manual_seed(3)
model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=S, encoder_depth=(1,) * 5, encoder_width=W, strides=(1, 2, 2, 2, 2),
                      decoder_depth=(1,) * 4, norm=ft.LayerNorm, reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": patch}),
                      act=nn.ReLU, factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in model.state_dict().items()}
cfg = dict(widths=W, strides=(1, 2, 2, 2, 2), reshape=dict(head_dim=8, patch_size=patch), num_iters=5, solver="hals")
x = torch.rand(B, 4, *S).double(); gy = torch.randn(B, 3, *S).double()
cap = {}
orig = O.factorizer_block


def blk(xx, sd, prefix, c):
    y = orig(xx, sd, prefix, c)
    if "x" not in cap:
        cap["x"] = xx; y.retain_grad(); cap["y"] = y; cap["prefix"] = prefix
    return y


O.factorizer_block = blk
prm = {k: v.clone().requires_grad_(True) for k, v in sd64.items() if v.is_floating_point() and not k.endswith(("u0", "v0"))}
full = dict(sd64); full.update(prm)
O.factorizer_forward(x, full, cfg).backward(gy)
O.factorizer_block = orig
xb = cap["x"].detach().float(); gyb = cap["y"].grad.float(); pre = cap["prefix"]
print("block", pre, "x", tuple(xb.shape), "gy max %.3e rms %.3e" % (gyb.abs().max(), gyb.pow(2).mean().sqrt()))
bsd = {k[len(pre):]: v for k, v in model.state_dict().items() if k.startswith(pre)}
bcfg = dict(reshape=dict(head_dim=8, patch_size=patch), num_iters=5, solver="hals")


def oracle(dt):
    p = {k: v.clone().to(dt).requires_grad_(True) for k, v in bsd.items() if not k.endswith(("u0", "v0"))}
    f = {k: v.to(dt) for k, v in bsd.items()}; f.update(p)
    xo = xb.to(dt).requires_grad_(True)
    yo = O.factorizer_block(xo, f, "", bcfg)
    return dict(zip(["x"] + list(p.keys()), torch.autograd.grad(yo, [xo] + list(p.values()), gyb.to(dt))))


g32, g64 = oracle(torch.float32), oracle(torch.float64)
dblk = model.encoder.blocks[0].block.blocks[0].cuda()
xd = xb.cuda().requires_grad_(True)
names = [n for n, _ in dblk.named_parameters()]
gd = dict(zip(["x"] + names, torch.autograd.grad(dblk(xd), [xd] + [p for _, p in dblk.named_parameters()], gyb.cuda())))
print("%-34s %9s %9s %9s %9s" % ("tensor", "dev max", "f32 max", "dev rms", "f32 rms"))
for n in g64:
    sc = g64[n].abs().max().item() + 1e-30
    d = gd[n].double().cpu() - g64[n]; e = g32[n].double() - g64[n]
    print("%-34s %9.2e %9.2e %9.2e %9.2e" % (n, d.abs().max() / sc, e.abs().max() / sc, d.pow(2).mean().sqrt() / sc, e.pow(2).mean().sqrt() / sc))

# ---- where are the outliers of the input gradient?  (window-0 patch of the worst voxels: how many entries of t are positive) ----
d = (gd["x"].double().cpu() - g64["x"]).abs()
sc = g64["x"].abs().max().item()
print("x-gradient: elements with error > 1e-5 of max|g|: %d of %d" % (int((d > 1e-5 * sc).sum()), d.numel()))
f64 = {k: v.double() for k, v in bsd.items()}
t = torch.relu(O.linear_cf(O.layernorm_cf(xb.double(), f64["norm1.norm.weight"], f64["norm1.norm.bias"]), f64["fact.in_proj.linear.weight"]))
idx = d.flatten().topk(6).indices
for i in idx.tolist():
    c = tuple(int(v) for v in torch.unravel_index(torch.tensor(i), d.shape))
    b_, ch, z0, z1, z2 = c
    hd = ch // 8
    for wname, s in (("w0", 0), ("w1", patch // 2)):
        # window with shift s: rolled[z] = t[z - s]; patch g covers rolled voxels [g*p, g*p+p) = original [g*p - s, ...)
        g0, g1, g2 = ((z0 + s) % S[0]) // patch, ((z1 + s) % S[1]) // patch, ((z2 + s) % S[2]) // patch
        zs = [[(g * patch + k - s) % n for k in range(patch)] for g, n in ((g0, S[0]), (g1, S[1]), (g2, S[2]))]
        pt = t[b_, hd * 8:(hd + 1) * 8][:, zs[0]][:, :, zs[1]][:, :, :, zs[2]]
        pos = pt[pt > 0]
        print("  err %.2e at %s (head %d) %s patch: %d/%d positive, max %.3e, min+ %.3e" % (d[c] / sc, c, hd, wname, pos.numel(), pt.numel(), pt.max(), pos.min() if pos.numel() else 0.0))

# ---- the outlier voxel: LayerNorm statistics and the three evaluations of dL/dx there ----
b_, ch, z0, z1, z2 = tuple(int(v) for v in torch.unravel_index(torch.tensor(idx[0].item()), d.shape))
xv = xb[b_, :, z0, z1, z2].double()
print("voxel (%d,%d,%d): mean %.6f  std over channels %.3e  rstd %.2f   (all voxels: min std %.3e, median %.3e)" % (
    z0, z1, z2, xv.mean(), xv.var(unbiased=False).sqrt(), (xv.var(unbiased=False) + 1e-5).rsqrt(),
    xb.double().var(1, unbiased=False).sqrt().min(), xb.double().var(1, unbiased=False).sqrt().median()))
for nme, gg in (("dev", gd["x"].double().cpu()), ("f32", g32["x"].double()), ("f64", g64["x"])):
    print("  %s gx[:4] at the voxel:" % nme, [float("%.6e" % v) for v in gg[b_, :4, z0, z1, z2]])
# is it the statistics?  device forward stats are not exported; compare an fp32 two-pass evaluation with fp64
x32 = xb[b_, :, z0, z1, z2]
m32 = x32.mean(); v32 = ((x32 - m32) ** 2).mean()
print("  fp32 two-pass mean %.9f var %.6e | fp64 mean %.9f var %.6e | fp32 E[x^2]-E[x]^2 var %.6e" % (
    m32, v32, xv.mean(), xv.var(unbiased=False), (x32 * x32).mean() - m32 * m32))

# ---- is it a ReLU gate within rounding of zero?  z = in_proj(LN1(x)) before the ReLU, float64 and float32 ----
z64 = O.linear_cf(O.layernorm_cf(xb.double(), f64["norm1.norm.weight"], f64["norm1.norm.bias"]), f64["fact.in_proj.linear.weight"])
f32 = {k: v.float() for k, v in bsd.items()}
z32 = O.linear_cf(O.layernorm_cf(xb, f32["norm1.norm.weight"], f32["norm1.norm.bias"]), f32["fact.in_proj.linear.weight"])
zv = z64[b_, :, z0, z1, z2]
k = int(zv.abs().argmin())
print("pre-activations at the voxel: min |z| = %.3e (channel %d; fp32 oracle %.3e) of max|z| %.3e; whole tensor: %d elements with |z| < 1e-6, smallest %.3e" % (
    zv.abs().min(), k, z32[b_, k, z0, z1, z2], z64.abs().max(), int((z64.abs() < 1e-6).sum()), z64.abs().min()))
tdev = torch.relu(dblk.fact.in_proj(dblk.norm1(xb.cuda()))).cpu() if False else None
