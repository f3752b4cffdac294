"""Hypothesis for the 0.25 ms -> 4.6 ms fwd+bwd of the split-N NMF inside the test suite: the CPU oracle that the test
runs first leaves the intra-op thread pool (one thread per host CPU) spinning, and the backward — which torch hands to
its autograd device thread — waits for a time slice on an oversubscribed host.  Time the GPU path (a) alone, (b) right
after CPU autograd work like the test's oracle, (c) while another thread keeps every CPU busy, (d) with KernelTimer
(device time of the launches only)."""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, ".")
import factorizer_amd as ft  # noqa: E402
from factorizer_amd import functional as Fn  # noqa: E402
from oracle import cpu_ref as O  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
nmf = ft.NMF(size=(16, 64 ** 3), rank=1, num_iters=5, init="uniform", solver="mu").to(dev)
td = torch.rand(1, 1, 16, 64 ** 3, device=dev, requires_grad=True)
gm = torch.rand_like(td)


def fb(n=20):
    def fn():
        return torch.autograd.grad(nmf(td), td, gm)
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / n, 4)


def cpu_oracle_like():
    x = torch.rand(1, 1, 16, 64 ** 3, requires_grad=True)
    u0, v0 = torch.rand(16, 1), torch.rand(64 ** 3, 1)
    y = O.nmf_forward(x, u0, v0, 5, "mu")
    torch.autograd.grad(y, x, torch.rand_like(y))


res = {"host_cpus": os.cpu_count(), "torch_threads": torch.get_num_threads(), "alone": fb()}
cpu_oracle_like()
res["right_after_cpu_oracle"] = fb()
time.sleep(1.0)
res["1s_after_cpu_oracle"] = fb()
stop = False


def burn():
    a = torch.rand(2048, 2048)
    while not stop:
        a = (a @ a).clamp_(0, 1)


th = threading.Thread(target=burn)
th.start()
time.sleep(0.3)
res["while_all_cpus_busy"] = fb()
stop = True
th.join()
timer = Fn.KernelTimer()
Fn.set_timer(timer)
for _ in range(10):
    torch.autograd.grad(nmf(td), td, gm)
Fn.set_timer(None)
agg = timer.summary()
res["device_time_per_call_ms"] = {k: round(v["ms"] / v["calls"], 4) for k, v in agg.items()}
print(json.dumps(res, indent=1))
