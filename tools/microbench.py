"""Per-kernel timing of the native hot-path kernels at BASELINE sizes (HIP events, GB/s of
ALGORITHMIC bytes: SURVEY.md §8d).  Usage: python tools/microbench.py [--quick]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import factorizer_amd as ft  # noqa: E402
from factorizer_amd import functional as Fn  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, warm=3):
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


def main():
    res = {}
    B = 2
    x = torch.rand(B, 32, 128, 128, 128, device=DEV)
    nbx = x.numel() * 4
    # plain copy reference (device memcpy-like)
    y0 = torch.empty_like(x)
    ms = timeit(lambda: y0.copy_(x))
    res["copy_537MB"] = {"ms": ms, "GBs": 2 * nbx / ms / 1e6}
    for shifts, tag in ((None, "w2"), ([None, 2, 4, 6], "w4")):
        m = ft.SWMatricize((None, 32, 128, 128, 128), head_dim=8, patch_size=8, shifts=shifts)
        geo = m.geometry
        W = geo.nshift
        y = Fn._swm_fwd_raw(x, geo)
        ms = timeit(lambda: Fn._swm_fwd_raw(x, geo))
        res[f"swm_fwd_{tag}"] = {"ms": ms, "GBs": (1 + W) * nbx / ms / 1e6}
        ms = timeit(lambda: Fn._swm_inv_raw(y, geo))
        res[f"swm_inv_{tag}"] = {"ms": ms, "GBs": (1 + W) * nbx / ms / 1e6}
        del y
    for solver, R, T in (("hals", 1, 5), ("mu", 2, 5), ("hals", 2, 10)):
        nmat = 65536
        xm = torch.rand(nmat, 8, 512, device=DEV)
        gy = torch.rand_like(xm)
        u0, v0 = torch.rand(8, R, device=DEV), torch.rand(512, R, device=DEV)
        nb = xm.numel() * 4
        ms = timeit(lambda: Fn._nmf_fwd_raw(xm, u0, v0, T, solver, 1e-16))
        flops = {"hals": 4 * 8 * 512 * R + 2 * 520 * R * R + 2 * 520 * R * (R - 1),
                 "mu": 4 * 8 * 512 * R + 4 * 520 * R * R}[solver] * T + 2 * 8 * 512 * R
        res[f"nmf_fwd_{solver}_r{R}_t{T}"] = {"ms": ms, "GBs": 2 * nb / ms / 1e6,
                                              "TFLOPs": flops * nmat / ms / 1e9, "Mmat_s": nmat / ms / 1e3}
        ms = timeit(lambda: Fn._nmf_bwd_raw(xm, u0, v0, gy, None, None, T, T, solver, 1e-16), iters=10)
        res[f"nmf_bwd_{solver}_r{R}_t{T}"] = {"ms": ms, "GBs": 3 * nb / ms / 1e6, "Mmat_s": nmat / ms / 1e3}
    for k, v in res.items():
        print(k, json.dumps({a: round(b, 3) for a, b in v.items()}))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/microbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
