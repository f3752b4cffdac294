import os, sys, torch
sys.path.insert(0, '.')
import factorizer_amd as ft
from factorizer_amd import functional as Fn
DEV='cuda:0'
m = ft.SWMatricize((None, 32, 64, 64, 64), head_dim=8, patch_size=8)
geo = m.geometry
torch.manual_seed(0)
t = torch.rand(2, 32, 64, 64, 64, device=DEV); u0 = torch.rand(8, 1, device=DEV); v0 = torch.rand(512, 1, device=DEV)
outs={}
for tile in ("0","8"):
    os.environ["FZ_CF_TILE"]=tile
    ctx = type('X', (), {'save_for_backward': lambda self, *a: None})()
    outs[tile]=Fn.FactCoreFn.forward(ctx, t, u0, v0, geo, 5, 5, 'hals', 1e-16, True).clone()
d=(outs["0"]-outs["8"]).abs()
print("max abs diff", d.max().item(), "mean", d.mean().item(), "ref mean", outs["0"].abs().mean().item())
