"""Replay / coverage check of fz_gemm in bf16 and fp32 storage: the output buffer is pre-filled with different junk in two
runs; any element that differs was not written (or was written nondeterministically)."""
import sys

import torch

sys.path.insert(0, ".")
from factorizer_amd import pointwise as PW  # noqa: E402

dev = "cuda:0"
for (B, Cin, M, V, w_t) in ((1, 64, 64, 76800, True), (1, 64, 64, 76800, False), (1, 128, 64, 76800, False), (2, 64, 64, 262144, True),
                            (1, 64, 64, 9600, True), (1, 128, 128, 9600, True)):
    for dt in (torch.bfloat16, torch.float32):
        torch.manual_seed(0)
        x = torch.randn(B, Cin, V, 1, 1, device=dev).to(dt)
        w = torch.randn(Cin, M, device=dev) / Cin ** 0.5 if w_t else torch.randn(M, Cin, device=dev) / Cin ** 0.5
        outs = []
        for rep in range(2):
            y = torch.full((B, M, V, 1, 1), float(100 + rep), device=dev).to(dt)
            if w_t:
                PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=M, K=Cin, Ncol=V, w_t=True, ldw=M)
            else:
                PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=M, K=Cin, Ncol=V)
            torch.cuda.synchronize()
            outs.append(y.float().clone())
        ref = torch.einsum("km,bkv->bmv" if w_t else "mk,bkv->bmv", w.double(), x.double().flatten(2)).reshape(B, M, V, 1, 1)
        nd = int((outs[0] != outs[1]).sum())
        err = ((outs[0].double() - ref).abs().max() / ref.abs().max()).item()
        print(f"B={B} {Cin}->{M} V={V} w_t={w_t} {dt}: differing {nd}, rel err {err:.2e}")
        if nd:
            idx = (outs[0] != outs[1]).nonzero()
            print("   rows:", sorted(set(idx[:, 1].tolist()))[:20], " cols min/max:", int(idx[:, 2].min()), int(idx[:, 2].max()), " vals", outs[0][tuple(idx[0])].item(), outs[1][tuple(idx[0])].item())
