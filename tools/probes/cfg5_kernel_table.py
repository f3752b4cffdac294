"""Per-kernel table (HIP events around every native launch) of one BASELINE configs[4] training step, B = 4, bf16: the NMF rows."""
import os, sys, json, contextlib
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_)
import torch
from torch import nn
import factorizer_amd as ft
from factorizer_amd import functional as Fn
DEV = "cuda:0"
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(160, 192, 160), norm=ft.LayerNorm,
                      reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}), act=nn.ReLU,
                      factorize=ft.NMF, rank=2, num_iters=10, init="uniform", solver="hals", mlp_ratio=2, dropout=0.1).to(DEV).train()
x = torch.rand(B, 4, 160, 192, 160, device=DEV)
t = (torch.rand(B, 3, 160, 192, 160, device=DEV) > 0.5).float()
def fb():
    for p in model.parameters(): p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = ft.dice_ce_loss(model(x), t)
    loss.backward()
fb(); fb()
tm = Fn.KernelTimer(); Fn.set_timer(tm); fb(); Fn.set_timer(None)
agg = tm.summary()
tot = sum(a["ms"] for a in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])
print(json.dumps({"FZ_PCF_HALF": os.environ.get("FZ_PCF_HALF", "1"), "kernel_ms_total": round(tot, 2),
                  "nmf": {k: [a["calls"], round(a["ms"], 3)] for k, a in rows if k.startswith("nmf_")}}))
if len(sys.argv) > 2:
    for k, a in rows[:40]:
        print("%-44s %3d calls %8.3f ms  %7.1f GB/s" % (k, a["calls"], a["ms"], a["bytes"] / (a["ms"] * 1e-3) / 1e9 if a["ms"] > 0 else 0))
