"""Timings of the other BASELINE.json configs on one MI355X; bench.py is cfg 3/4.

  cfg 1  ft.NMF(size=(8,512), rank=2, num_iters=5, solver='mu') on CPU, x=(1,8,512)      [µs]
  cfg 2  FactorizerBlock(C=32, 128^3, d=8, p=8, HALS R1 T5) fwd / bwd, B in {1,2}          [ms]
  cfg 3  README Swin Factorizer, eval forward, B=2                                          [volumes/s]
Prints one JSON line per measurement.
"""
import json
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import factorizer_amd as ft  # noqa: E402

DEV = "cuda:0"


def gpu_time(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def cfg1():
    torch.set_num_threads(min(8, os.cpu_count() or 1))  # tiny op: more threads only add overhead
    torch.manual_seed(0)
    nmf = ft.NMF(size=(8, 512), rank=2, num_iters=5, init="uniform", solver="mu")
    x = torch.rand(1, 8, 512)
    with torch.no_grad():
        for _ in range(20):
            nmf(x)
        t0 = time.perf_counter()
        for _ in range(200):
            nmf(x)
        us = (time.perf_counter() - t0) / 200 * 1e6
    print(json.dumps({"config": "cfg1 NMF(8x512,R2,T5,mu) CPU forward, batch 1", "us": round(us, 1)}))
    xd = x.to(DEV)
    nd = nmf.to(DEV)
    with torch.no_grad():
        ms = gpu_time(lambda: nd(xd), 200, 20)
    print(json.dumps({"config": "cfg1 same on MI355X (one wave; launch-bound)", "us": round(ms * 1e3, 1)}))


def cfg2():
    torch.manual_seed(0)
    blk = ft.FactorizerBlock(channels=32, spatial_size=(128, 128, 128), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                             factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals",
                             mlp_ratio=2, dropout=0.0).to(DEV)
    for B in (1, 2):
        torch.manual_seed(0)
        x = torch.rand(B, 32, 128, 128, 128, device=DEV, requires_grad=True)
        torch.manual_seed(1)
        g = torch.rand(B, 32, 128, 128, 128, device=DEV)
        with torch.no_grad():
            fwd = gpu_time(lambda: blk(x), 50, 10)

        def fb():
            y = blk(x)
            torch.autograd.grad(y, [x] + list(blk.parameters()), g)
        both = gpu_time(fb, 50, 10)
        print(json.dumps({"config": f"cfg2 FactorizerBlock C=32 128^3 B={B}", "fwd_ms": round(fwd, 3),
                          "fwd_bwd_ms": round(both, 3), "volumes_per_s_fwd_bwd": round(B / both * 1e3, 1)}))


def cfg3():
    torch.manual_seed(0)
    model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(128, 128, 128), norm=ft.LayerNorm,
                          reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                          factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2,
                          dropout=0.1).to(DEV).eval()
    x = torch.rand(2, 4, 128, 128, 128, device=DEV)
    with torch.no_grad():
        ms = gpu_time(lambda: model(x), 30, 5)
    print(json.dumps({"config": "cfg3 README Swin Factorizer eval forward B=2", "ms": round(ms, 3),
                      "volumes_per_s": round(2 / ms * 1e3, 1)}))


def prod():
    """the BraTS bundle's model (model_zoo/factorizer_brats23/configs/train.yaml:44-66):
    shift windows [None, 2, 4, 6], mlp_ratio 4 — training step (fwd + bwd) and eval forward, B = 2."""
    torch.manual_seed(0)
    kw = dict(in_channels=4, out_channels=3, spatial_size=(128, 128, 128), norm=ft.LayerNorm,
              reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8, "shifts": [None, 2, 4, 6]}), act=nn.ReLU,
              factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=4, dropout=0.1)
    model = ft.Factorizer(**kw).to(DEV).train()
    x = torch.rand(2, 4, 128, 128, 128, device=DEV)
    t = (torch.rand(2, 3, 128, 128, 128, device=DEV) > 0.5).float()

    def fb():
        for p in model.parameters():
            p.grad = None
        ft.dice_ce_loss(model(x), t).backward()
    ms = gpu_time(fb, 10, 3)
    print(json.dumps({"config": "f-1 production Swin Factorizer (4 shift windows, mlp_ratio 4) fwd+bwd B=2",
                      "ms": round(ms, 3), "volumes_per_s": round(2 / ms * 1e3, 1)}))
    model.eval()
    with torch.no_grad():
        ms = gpu_time(lambda: model(x), 20, 3)
    eval_ms = ms
    print(json.dumps({"config": "f-1 production Swin Factorizer eval forward B=2", "ms": round(ms, 3),
                      "volumes_per_s": round(2 / ms * 1e3, 1)}))
    # the bundle's inference: one BraTS volume through SlidingWindowInfererAdapt (inference.yaml:96-102)
    vol = torch.rand(1, 4, 240, 240, 155, device=DEV)
    inf = ft.SlidingWindowInfererAdapt(roi_size=(128, 128, 128), sw_batch_size=2, overlap=0.5, mode="gaussian")
    with torch.no_grad():
        ms = gpu_time(lambda: inf(vol, model), 5, 2)
    net_only = eval_ms * 9  # 18 windows = 9 batches of 2, at the steady eval-forward time measured above
    print(json.dumps({"config": "f-1 sliding-window inference, 240x240x155 volume, roi 128^3, overlap 0.5, gaussian",
                      "ms_per_volume": round(ms, 2), "of_which_network_ms": round(net_only, 2),
                      "stitching_ms": round(ms - net_only, 2), "windows": 18}))


def cfg5(batches=(1, 2, 4), dtypes=("bf16", "f32")):
    """BASELINE configs[4] (SURVEY "cfg 5"), one GPU's share: 160x192x160 volumes, rank 2, 10 iterations,
    bf16 mixed precision (bf16 activation storage under torch.autocast, fp32 parameters / statistics /
    NMF internals) — and the same in fp32 for comparison.
    p = 8 does not divide the deeper stages of this shape (SURVEY headline 5), so the patch is (5, 6, 5)
    (N = 150: the masked 8x<=512 NMF family through the modular matricize -> NMF -> inverse kernels)."""
    import contextlib
    torch.manual_seed(0)
    model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(160, 192, 160), norm=ft.LayerNorm,
                          reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}), act=nn.ReLU,
                          factorize=ft.NMF, rank=2, num_iters=10, init="uniform", solver="hals", mlp_ratio=2,
                          dropout=0.1).to(DEV).train()
    for dt in dtypes:
      for B in batches:
        x = torch.rand(B, 4, 160, 192, 160, device=DEV)
        t = (torch.rand(B, 3, 160, 192, 160, device=DEV) > 0.5).float()
        ctx = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if dt == "bf16" else contextlib.nullcontext

        def fb():
            for p in model.parameters():
                p.grad = None
            with ctx():
                loss = ft.dice_ce_loss(model(x), t)
            loss.backward()
            return loss
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        loss = fb()
        assert torch.isfinite(loss).item()
        ms = gpu_time(fb, 3, 1)
        print(json.dumps({"config": f"cfg5 stress shape 160x192x160, HALS R2 T10, patch (5,6,5), fwd+bwd B={B}",
                          "dtype": dt + (" activations, fp32 parameters / statistics / NMF internals" if dt == "bf16" else ""),
                          "ms": round(ms, 2), "volumes_per_s": round(B / ms * 1e3, 2),
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
                          "loss": round(float(loss), 5)}), flush=True)
        del x, t


def wide():
    """The reference test config's global matricize (tests/test_factorizer.py: Matricize(num_heads=1, grid_size=1)) at
    16 channels x 64^3 voxels: ONE 16 x 262144 matrix, MU rank 1, 5 iterations — the split-N kernels (csrc/gnmf.hip).
    Per-call medians (one event pair per call) of forward and forward + backward."""
    torch.manual_seed(0)
    nmf = ft.NMF(size=(16, 64 ** 3), rank=1, num_iters=5, init="uniform", solver="mu").to(DEV)
    td = torch.rand(1, 1, 16, 64 ** 3, device=DEV, requires_grad=True)
    gm = torch.rand_like(td)

    def med(fn, n=30):
        for _ in range(5):
            fn()
        ev = []
        for _ in range(n):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            ev.append((s, e))
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in ev)
        return ts[len(ts) // 2], ts[-1]
    with torch.no_grad():
        f, fmax = med(lambda: nmf(td))
    fb, fbmax = med(lambda: torch.autograd.grad(nmf(td), td, gm))
    nbytes = td.numel() * 4
    print(json.dumps({"config": "wide NMF 16 x 262144 (global matricize, MU R1 T5), one matrix", "fwd_ms": round(f, 3),
                      "fwd_bwd_ms": round(fb, 3), "fwd_max_ms": round(fmax, 3), "fwd_bwd_max_ms": round(fbmax, 3),
                      "fwd_kernel_traffic_GBps": round((2 * 5 + 2) * nbytes / (f * 1e-3) / 1e9, 1)}))


if __name__ == "__main__":
    which = sys.argv[1:] or ["cfg1", "cfg2", "cfg3", "prod", "wide"]
    for name in which:
        {"cfg1": cfg1, "cfg2": cfg2, "cfg3": cfg3, "prod": prod, "cfg5": cfg5, "wide": wide}[name]()
