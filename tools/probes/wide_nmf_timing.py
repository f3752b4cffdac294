"""Split-N NMF at the reference's test shape (16 x 262144, MU R1 T5): forward and forward+backward timing,
per call, to find why the HEAD parity run of round 2 recorded fwd_bwd_ms 4.84 against 0.28 ms earlier."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import factorizer_amd as ft  # noqa: E402
from factorizer_amd import functional as Fn  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
nmf = ft.NMF(size=(16, 64 ** 3), rank=1, num_iters=5, init="uniform", solver="mu").to(dev)
td = torch.rand(1, 1, 16, 64 ** 3, device=dev, requires_grad=True)
gm = torch.rand_like(td)


def per_call(fn, n=30):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        out.append((round(s.elapsed_time(e), 4), round((time.perf_counter() - t0) * 1e3, 4)))
    return out


def batch(fn, n=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


res = {}
with torch.no_grad():
    res["fwd_batch_ms"] = batch(lambda: nmf(td))
res["fwd_bwd_batch_ms"] = batch(lambda: torch.autograd.grad(nmf(td), td, gm))
res["fwd_bwd_per_call_gpu_host_ms"] = per_call(lambda: torch.autograd.grad(nmf(td), td, gm))
# backward alone through the raw entry points (no autograd graph), workspace reused
u0, v0 = nmf.init.u0.contiguous(), nmf.init.v0.contiguous()
x = td.detach().contiguous()
res["raw_fwd_ms"] = batch(lambda: Fn._gnmf_fwd_raw(x, u0, v0, 5, "mu", 1e-16))
res["raw_bwd_ms"] = batch(lambda: Fn._gnmf_bwd_raw(x, u0, v0, gm, None, None, 5, 5, "mu", 1e-16))
# with the allocator under pressure (many live blocks): does an allocation in the timed lambda sync?
junk = [torch.empty(64 * 1024 * 1024, device=dev) for _ in range(40)]
res["fwd_bwd_batch_ms_with_10GB_live"] = batch(lambda: torch.autograd.grad(nmf(td), td, gm))
del junk
torch.cuda.empty_cache()
res["fwd_bwd_batch_ms_after_empty_cache"] = batch(lambda: torch.autograd.grad(nmf(td), td, gm))
print(json.dumps(res, indent=1))
