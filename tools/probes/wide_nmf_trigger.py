"""Find what makes the split-N NMF backward 15x slower inside the test suite than alone: time it, run a candidate
trigger, time it again (one process)."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import factorizer_amd as ft  # noqa: E402
from factorizer_amd import _native, pointwise as PW  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
nmf = ft.NMF(size=(16, 64 ** 3), rank=1, num_iters=5, init="uniform", solver="mu").to(dev)
td = torch.rand(1, 1, 16, 64 ** 3, device=dev, requires_grad=True)
gm = torch.rand_like(td)


def fb():
    def fn():
        return torch.autograd.grad(nmf(td), td, gm)
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.record()
    for _ in range(20):
        fn()
    e.record()
    host = (time.perf_counter() - t0) / 20 * 1e3
    torch.cuda.synchronize()
    return round(s.elapsed_time(e) / 20, 4), round(host, 4)


res = {"alone": fb()}
x = torch.randn(2, 128, 8, 8, 8, device=dev)
w = torch.randn(128, 128, 1, device=dev) / 128 ** 0.5
PW.linear_cf(x, w, None)
res["after_linear_cf"] = fb()
_native.lib().fz_gemm_bx_enable(-1)
res["after_bx_query"] = fb()
import parity  # noqa: E402,F401
res["after_import_parity"] = fb()
import pytest  # noqa: E402,F401
res["after_import_pytest"] = fb()
import warnings  # noqa: E402
with warnings.catch_warnings():
    warnings.simplefilter("error", RuntimeWarning)
    res["inside_warnings_error"] = fb()
torch.manual_seed(1)
big = [torch.empty(256 * 1024 * 1024, device=dev) for _ in range(8)]
del big
res["after_8GB_alloc_free"] = fb()
print(json.dumps(res, indent=1))
