"""the generic-patch fused core alone at the cfg-5 stage-0 shape (160x192x160, patch (5,6,5), HALS R2 T10) — for counter passes"""
import sys, torch
sys.path.insert(0, '.')
import factorizer_amd as ft
from factorizer_amd import functional as Fn
DEV = 'cuda:0'
S, patch, C = (160, 192, 160), (5, 6, 5), 32
m = ft.SWMatricize((None, C, *S), head_dim=8, patch_size=patch)
nmf = ft.NMF(size=(8, 150), rank=2, num_iters=10, init="uniform", solver="hals").to(DEV)
t = torch.rand(1, C, *S, device=DEV, requires_grad=True)
for _ in range(3):
    a = Fn.FactCoreFn.apply(t, nmf.init.u0, nmf.init.v0, m.geometry, 10, 10, "hals", 1e-16, True)
    (g,) = torch.autograd.grad(a, t, torch.ones_like(a))
torch.cuda.synchronize()
