"""Is the training step host-bound?  (1) wall time per step (synchronised) against the time the host needs to ENQUEUE a
step (no synchronisation: returns as soon as Python has issued every launch); (2) the same step captured once in a
HIP graph (torch.cuda.CUDAGraph) and replayed — no Python, ctypes or allocator work per launch."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as BN  # noqa: E402
import factorizer_amd as ft  # noqa: E402
from factorizer_amd.parallel import FlatGradSync  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ft.Factorizer(**BN.MODEL_KW).to(dev).train()
sync = FlatGradSync(model, num_buckets=2, overlap=True, late_wgrad_join=True)
opt = ft.FlatAdamW(model, lr=1e-4, weight_decay=1e-5, flat_grad=sync.flat, grad_views=sync.views)
x = torch.rand(2, 4, 128, 128, 128, device=dev)
target = (torch.rand(2, 3, 128, 128, 128, device=dev) > 0.5).float()


def step(with_opt=True):
    sync.zero_grad()
    loss = ft.dice_ce_loss(model(x), target)
    loss.backward()
    scale = sync.finish(average=False)
    if with_opt:
        opt.step(grad_scale=scale)
    return loss


res = {}
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
res["eager_ms_per_step"] = (time.perf_counter() - t0) / 10 * 1e3
# host enqueue time: run one step at a time, measure until step() returns, then drain
enq = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    enq.append((time.perf_counter() - t0) * 1e3)
    torch.cuda.synchronize()
res["host_enqueue_ms_per_step"] = sorted(enq)[len(enq) // 2]
# single step latency from idle (sync before and after)
lat = []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - t0) * 1e3)
res["single_step_from_idle_ms"] = sorted(lat)[len(lat) // 2]
print(json.dumps(res), flush=True)

# ---- graph capture of the whole step ----
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        loss = step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    res["graph_replay_ms_per_step"] = (time.perf_counter() - t0) / 10 * 1e3
    res["graph_loss"] = float(loss)
except Exception as ex:  # noqa: BLE001
    res["graph_error"] = repr(ex)[:600]
print(json.dumps(res), flush=True)
