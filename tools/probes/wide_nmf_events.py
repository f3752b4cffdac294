"""Per-call device time of fz_gnmf_fwd / fz_gnmf_bwd bracketed by events (KernelTimer), call by call."""
import json
import sys

import torch

sys.path.insert(0, ".")
import factorizer_amd as ft  # noqa: E402
from factorizer_amd import functional as Fn  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
nmf = ft.NMF(size=(16, 64 ** 3), rank=1, num_iters=5, init="uniform", solver="mu").to(dev)
td = torch.rand(1, 1, 16, 64 ** 3, device=dev, requires_grad=True)
gm = torch.rand_like(td)
for _ in range(3):
    torch.autograd.grad(nmf(td), td, gm)
torch.cuda.synchronize()
timer = Fn.KernelTimer()
Fn.set_timer(timer)
for i in range(12):
    torch.autograd.grad(nmf(td), td, gm)
    if i == 5:
        torch.cuda.synchronize()
Fn.set_timer(None)
torch.cuda.synchronize()
out = {}
for name, nbytes, s, e, cols, flops in timer.records:
    out.setdefault(name, []).append(round(s.elapsed_time(e), 4))
print(json.dumps(out))
# the same with a synchronize before every call (no queueing ahead)
timer = Fn.KernelTimer()
Fn.set_timer(timer)
for i in range(6):
    torch.cuda.synchronize()
    torch.autograd.grad(nmf(td), td, gm)
Fn.set_timer(None)
torch.cuda.synchronize()
out = {}
for name, nbytes, s, e, cols, flops in timer.records:
    out.setdefault(name + "_synced", []).append(round(s.elapsed_time(e), 4))
print(json.dumps(out))
