"""Time the stage-0 MLP chain kernels (forward; backward with both weight gradients) under FZ_CHAIN_STAGGER (read once per
process: run once per value).  usage: FZ_CHAIN_STAGGER=n python tools/probes/chain_stagger.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from factorizer_amd import pointwise as PW

B, C, Hd, S = 2, 32, 64, (128, 128, 128)
dev = "cuda:0"
torch.manual_seed(0)
x1 = torch.randn(B, C, *S, device=dev)
g2 = torch.randn(B, C, *S, device=dev)
n2w, n2b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
w1, b1 = torch.randn(Hd, C, device=dev) / C ** 0.5, torch.randn(Hd, device=dev) * 0.1
w2, b2 = torch.randn(C, Hd, device=dev) / Hd ** 0.5, torch.randn(C, device=dev) * 0.1


def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


x2, z1, st2 = PW._mlp_fwd_chain(x1, n2w, n2b, 1e-5, w1, b1, w2, b2)
tf = timeit(lambda: PW._mlp_fwd_chain(x1, n2w, n2b, 1e-5, w1, b1, w2, b2))
tb = timeit(lambda: PW._mlp_bwd_chain_wgrad(g2, z1, w1, w2, x1, st2, n2w, n2b))
print(json.dumps({"stagger": int(os.environ.get("FZ_CHAIN_STAGGER", "0")), "fwd_ms": round(tf, 4), "bwd_wgrad_ms": round(tb, 4)}))
