"""Isolated timing of fz_gemm at the stage 1-4 shapes of the README model: split-bf16 family (per tile config,
fz_gemm_desc.tune) against the fp32-MFMA family.  Prints one JSON line per (shape, variant): us per launch, TB/s of
algorithmic bytes, TFLOP/s (fp32-equivalent).
usage: python tools/probes/gemm_bx_bench.py [out.jsonl] [--cfgs 42,41,22,21,12,11]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from factorizer_amd import _native as N  # noqa: E402
from factorizer_amd import pointwise as PW  # noqa: E402

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
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else None
    cfgs = ["", "n42", "n41", "n22", "n21", "n11", "k22", "k21", "k12", "k11"]
    if "--cfgs" in sys.argv:
        cfgs = [""] + sys.argv[sys.argv.index("--cfgs") + 1].split(",")
    B = 2
    # (name, Cin, M, spatial edge, kind)
    shapes = [("lin", 64, 64, 64), ("lin", 64, 128, 64), ("lin", 128, 64, 64),
              ("lin", 128, 128, 32), ("lin", 128, 256, 32), ("lin", 256, 128, 32),
              ("lin", 256, 256, 16), ("lin", 256, 512, 16), ("lin", 512, 256, 16),
              ("lin", 512, 512, 8), ("lin", 512, 1024, 8), ("lin", 1024, 512, 8),
              ("ln", 64, 64, 64), ("ln", 128, 128, 32), ("ln", 256, 256, 16), ("ln", 512, 512, 8),
              ("gelu", 128, 64, 64), ("gelu", 256, 128, 32), ("gelu", 512, 256, 16), ("gelu", 1024, 512, 8),
              ("conv", 32, 64, 128), ("conv", 64, 128, 64), ("conv", 128, 256, 32), ("conv", 256, 512, 16),
              ("tconv", 64, 32, 64), ("tconv", 128, 64, 32), ("tconv", 256, 128, 16), ("tconv", 512, 256, 8)]
    lib = N.lib()
    if "--only" in sys.argv:
        pat = sys.argv[sys.argv.index("--only") + 1]
        shapes = [s for s in shapes if pat in f"{s[0]}_{s[1]}->{s[2]}@{s[3]}"]
    for kind, Cin, M, E in shapes:
        S = (E, E, E)
        V = E ** 3
        x = torch.randn(B, Cin, *S, device=DEV)
        if kind in ("lin", "ln", "gelu"):
            w = torch.randn(M, Cin, 1, device=DEV) / Cin ** 0.5
            b = torch.randn(M, device=DEV)
            y = torch.empty(B, M, *S, device=DEV)
            res = torch.randn(B, M, *S, device=DEV)
            g, bt = torch.rand(Cin, device=DEV) + 0.5, torch.randn(Cin, device=DEV)
            st = torch.empty(B, 2, V, device=DEV)
            w2 = w.reshape(M, Cin)
            if kind == "lin":
                fn = lambda t: PW._gemm([x], w2, y, B=B, Cin=Cin, Vin=V, M=M, K=Cin, Ncol=V, bias=b, tune=t)  # noqa: E731
                nbytes = 4 * (x.numel() + y.numel())
            elif kind == "ln":
                fn = lambda t: PW._gemm([x], w2, y, B=B, Cin=Cin, Vin=V, M=M, K=Cin, Ncol=V, bias=b, ln=(g, bt, 1e-5), stats_out=st, eact=1, tune=t)  # noqa: E731
                nbytes = 4 * (x.numel() + y.numel())
            else:
                fn = lambda t: PW._gemm([x], w2, y, B=B, Cin=Cin, Vin=V, M=M, K=Cin, Ncol=V, bias=b, bact=2, res=res, tune=t)  # noqa: E731
                nbytes = 4 * (x.numel() + 2 * y.numel())
            flops = 2.0 * B * V * Cin * M
        elif kind == "conv":
            w = torch.randn(M, Cin, 2, 2, 2, device=DEV) / (8 * Cin) ** 0.5
            b = torch.randn(M, device=DEV)
            Eo = E // 2
            y = torch.empty(B, M, Eo, Eo, Eo, device=DEV)
            fn = lambda t: PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=M, K=8 * Cin, Ncol=Eo ** 3, bias=b, loader=PW.LOAD_S2D,  # noqa: E731
                                    Di=E, Hi=E, Wi=E, Ho=Eo, Wo=Eo, tune=t)
            nbytes = 4 * (x.numel() + y.numel())
            flops = 2.0 * B * Eo ** 3 * 8 * Cin * M
        else:
            w = torch.randn(Cin, M, 2, 2, 2, device=DEV) / Cin ** 0.5
            b = torch.randn(M, device=DEV)
            y = torch.empty(B, M, 2 * E, 2 * E, 2 * E, device=DEV)
            fn = lambda t: PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=8 * M, K=Cin, Ncol=V, w_t=True, ldw=8 * M, bias=b,  # noqa: E731
                                    epilogue=PW.EPI_D2S, Ho=E, Wo=E, tune=t)
            nbytes = 4 * (x.numel() + y.numel())
            flops = 2.0 * B * V * Cin * 8 * M
        for variant in ["f32mfma"] + cfgs:
            # per-call settings only: the descriptor's `products` and `tune` fields (no environment, no process-wide switch)
            prod, tune = N.PRODUCTS_SPLIT_BF16, 0
            if variant == "f32mfma":
                prod = N.PRODUCTS_FP32_MFMA
            elif variant.startswith("n"):      # streaming form, tile <nacc><mb>
                if kind == "conv":
                    continue
                tune = 100 + int(variant[1:])
            elif variant.startswith("k"):      # K-split form, tile <nacc><mb>
                if kind == "conv" and variant[1] != "2":
                    continue
                tune = 200 + int(variant[1:])
            run = (lambda f, t: (lambda: f(t)))(fn, tune)
            try:
                with N.use_products(prod):
                    us = timeit(run)
            except Exception as ex:  # unsupported tile for this shape
                print(f"# {kind} {Cin}->{M}@{E} {variant}: {ex}", file=sys.stderr)
                continue
            rec = {"shape": f"{kind}_{Cin}->{M}@{E}^3", "variant": "bx_" + (variant or "auto") if variant != "f32mfma" else variant,
                   "us": round(us, 1), "TBps": round(nbytes / us / 1e6, 2), "TFLOPs": round(flops / us / 1e6, 1)}
            line = json.dumps(rec)
            print(line, flush=True)
            if out:
                out.write(line + "\n")


if __name__ == "__main__":
    main()
