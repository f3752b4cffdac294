"""Dump standalone ft.NMF forward / backward results for an A/B of two library builds (FZ_LIB_PATH):
python tools/probes/nmf_ab.py out.pt"""
import sys
import torch
sys.path.insert(0, ".")
import factorizer_amd as ft  # noqa: E402
out = {}
for solver in ("mu", "hals"):
    for R in (1, 2, 3, 4):
        torch.manual_seed(R)
        nmf = ft.NMF(size=(8, 512), rank=R, num_iters=5, init="uniform", solver=solver).cuda()
        x = torch.rand(64, 1, 8, 512, device="cuda", requires_grad=True)
        gy = torch.randn(64, 1, 8, 512, device="cuda")
        y = nmf(x)
        (gx,) = torch.autograd.grad(y, x, gy)
        out[f"{solver}{R}_y"] = y.detach().cpu()
        out[f"{solver}{R}_gx"] = gx.cpu()
torch.save(out, sys.argv[1])
