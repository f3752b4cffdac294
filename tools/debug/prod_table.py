"""Per-kernel table of one training step of the BraTS-bundle model (4 shift windows, mlp_ratio 4)."""
import sys, torch
from torch import nn
sys.path.insert(0, '.')
import factorizer_amd as ft
from factorizer_amd import functional as Fn
DEV = 'cuda:0'
torch.manual_seed(0)
kw = dict(in_channels=4, out_channels=3, spatial_size=(128, 128, 128), norm=ft.LayerNorm,
          reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8, "shifts": [None, 2, 4, 6]}), act=nn.ReLU,
          factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=4, dropout=0.1)
model = ft.Factorizer(**kw).to(DEV).train()
x = torch.rand(2, 4, 128, 128, 128, device=DEV)
t = (torch.rand(2, 3, 128, 128, 128, device=DEV) > 0.5).float()
def fb():
    for p in model.parameters(): p.grad = None
    ft.dice_bce_loss(model(x), t).backward()
for _ in range(3): fb()
torch.cuda.synchronize()
tm = Fn.KernelTimer(); Fn.set_timer(tm)
for _ in range(2): fb()
Fn.set_timer(None)
agg = tm.summary()
tot = sum(v["ms"] for v in agg.values()) / 2
print(f"sum of native kernel time {tot:.2f} ms/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:28]:
    print(f"{k:34s} {v['ms']/2:7.3f} ms/step  {v['calls']//2:3d}x  {v['bytes']/max(v['ms'],1e-9)/1e6:7.0f} GB/s")
