"""per-kernel table of one cfg5 (160x192x160, HALS R2 T10, patch (5,6,5)) training step: python tools/probes/cfg5_table.py [bf16|f32] [B]"""
import contextlib
import json
import os
import sys

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import factorizer_amd as ft  # noqa: E402
from factorizer_amd import functional as Fn  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
DEV = "cuda:0"
torch.manual_seed(0)
model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(160, 192, 160), norm=ft.LayerNorm,
                      reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}), act=nn.ReLU,
                      factorize=ft.NMF, rank=2, num_iters=10, init="uniform", solver="hals", mlp_ratio=2,
                      dropout=0.1).to(DEV).train()
x = torch.rand(B, 4, 160, 192, 160, device=DEV)
t = (torch.rand(B, 3, 160, 192, 160, device=DEV) > 0.5).float()
ctx = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if dt == "bf16" else contextlib.nullcontext


def fb():
    for p in model.parameters():
        p.grad = None
    with ctx():
        loss = ft.dice_ce_loss(model(x), t)
    loss.backward()


fb()
fb()
tm = Fn.KernelTimer()
Fn.set_timer(tm)
fb()
fb()
Fn.set_timer(None)
torch.cuda.synchronize()
agg = tm.summary()
rows = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in agg.values()) / 2
print(json.dumps({"dtype": dt, "B": B, "kernel_ms_per_step": round(tot, 2)}))
for n, v in rows[:40]:
    print(f"{v['ms'] / 2:8.3f} ms  {v['calls'] // 2:4d} x  {v['bytes'] / max(v['ms'], 1e-9) / 1e6:7.0f} GB/s  {n}")
