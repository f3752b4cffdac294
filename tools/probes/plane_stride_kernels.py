"""The library's own pointwise kernels on (2, 32, N) activations with N = 128^3 (plane stride 8 MiB exactly) and N = 128^3 + pad:
the same launches, 0.05 % more columns, only the distance between channel planes differs (tools/probes/plane_stride.hip)."""
import os, sys, json
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_)
import torch
from factorizer_amd import pointwise as PW
dev = "cuda:0"
torch.manual_seed(0)
C = 32
g = torch.rand(C, device=dev) + 0.5; b = torch.randn(C, device=dev) * 0.1
w = torch.randn(C, C, 1, device=dev) / C ** 0.5; bias = torch.randn(C, device=dev) * 0.1
w1 = torch.randn(64, C, 1, device=dev) / C ** 0.5; b1 = torch.randn(64, device=dev) * .1
w2 = torch.randn(C, 64, 1, device=dev) / 8; b2 = torch.randn(C, device=dev) * .1
wh = torch.randn(3, C, 1, device=dev) * .3; bh = torch.randn(3, device=dev) * .1


def timed(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for pad in (0, 1024, 4096 + 1024):
    N = 128 ** 3 + pad
    x = torch.randn(2, C, 1, 1, N, device=dev)
    r = torch.randn(2, C, 1, 1, N, device=dev)
    out = {"pad_columns": pad, "plane_stride_bytes": N * 4}
    with torch.no_grad():
        out["ln_linear_32->32 us"] = round(timed(lambda: PW.ln_linear(x, g, b, 1e-5, w, None, "relu")), 1)
        out["linear_res_32->32 us"] = round(timed(lambda: PW.act_linear_res(x, w, bias, r, "none")), 1)
        out["mlp 32->64->32 (two launches) us"] = round(timed(lambda: PW.mlp_cf(x, w1, b1, w2, b2)), 1)
        out["head 32->3 us"] = round(timed(lambda: PW.linear_cf(x, wh, bh)), 1)
    print(json.dumps(out))
