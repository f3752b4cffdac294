"""BASELINE configs[4], one GPU's share: 160x192x160 volumes, HALS rank 2, 10 iterations, patch (5,6,5), bf16 autocast, per-GPU batch 4 —
N training steps (forward + DiceCE + backward), for rocprofv3 (`-- python3 tools/cfg5_step.py 3`) and for tools/profile_cfg5.sh."""
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
import torch  # noqa: E402
from torch import nn  # noqa: E402

import factorizer_amd as ft  # noqa: E402

DEV = "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.manual_seed(0)
model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(160, 192, 160), norm=ft.LayerNorm,
                      reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}), act=nn.ReLU,
                      factorize=ft.NMF, rank=2, num_iters=10, init="uniform", solver="hals", mlp_ratio=2, dropout=0.1).to(DEV).train()
x = torch.rand(B, 4, 160, 192, 160, device=DEV)
t = (torch.rand(B, 3, 160, 192, 160, device=DEV) > 0.5).float()
for _ in range(steps):
    for p in model.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = ft.dice_ce_loss(model(x), t)
    loss.backward()
torch.cuda.synchronize()
print("loss", float(loss))
