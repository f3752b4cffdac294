"""Per-kernel table (HIP events) of one cfg-5 training fwd+bwd at B=1, fp32 and bf16."""
import os, sys, json, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch import nn
import factorizer_amd as ft
from factorizer_amd import functional as Fn
DEV = "cuda:0"
torch.manual_seed(0)
model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(160, 192, 160), norm=ft.LayerNorm,
                      reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}), act=nn.ReLU, factorize=ft.NMF,
                      rank=2, num_iters=10, init="uniform", solver="hals", mlp_ratio=2, dropout=0.1).to(DEV).train()
x = torch.rand(1, 4, 160, 192, 160, device=DEV)
t = (torch.rand(1, 3, 160, 192, 160, device=DEV) > 0.5).float()
for dt in ("f32", "bf16"):
    ctx = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if dt == "bf16" else contextlib.nullcontext
    def fb():
        for p in model.parameters():
            p.grad = None
        with ctx():
            loss = ft.dice_ce_loss(model(x), t)
        loss.backward()
    fb(); fb()
    tm = Fn.KernelTimer(); Fn.set_timer(tm); fb(); Fn.set_timer(None)
    agg = tm.summary()
    tot = sum(a["ms"] for a in agg.values())
    print(dt, "kernel ms total", round(tot, 1))
    fam = {}
    for k, a in agg.items():
        f = k.split("_")[0] + "_" + k.split("_")[1] if k.startswith(("nmf", "swm", "wgrad", "mlp", "ln", "act", "linear", "dgrad")) else k.split("_")[0]
        fam[f] = fam.get(f, 0) + a["ms"]
    print(sorted([(round(v, 2), k) for k, v in fam.items()], reverse=True)[:12])
    print(sorted([(round(a["ms"], 2), k, a["calls"], round(a["bytes"] / max(a["ms"], 1e-9) / 1e6)) for k, a in agg.items()], reverse=True)[:14])
