"""Replay determinism of the weight-gradient and dgrad+LayerNorm-backward kernels of a C = 64 block in bf16 / fp32."""
import sys

import torch

sys.path.insert(0, ".")
from factorizer_amd import pointwise as PW  # noqa: E402

dev = "cuda:0"
B, C, V = 1, 64, 76800
S = (40, 48, 40)
for dt in (torch.bfloat16, torch.float32):
    torch.manual_seed(0)
    gt = torch.randn(B, C, *S, device=dev).to(dt)
    x = (torch.randn(B, C, *S, device=dev) * 2 + 0.3).to(dt)
    gadd = torch.randn(B, C, *S, device=dev).to(dt)
    w = torch.randn(C, C, device=dev) / 8
    g, bt = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    xf = x.float()
    mean = xf.mean(1, keepdim=True)
    rstd = (xf.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    st = torch.cat([mean, rstd], 1).reshape(B, 2, V).contiguous()
    res = {}
    for rep in range(3):
        junk = [torch.full((B, C, *S), float(rep), device=dev) for _ in range(3)]
        del junk
        gw = torch.empty(C, C, device=dev)
        PW._wgrad(gt, [x], gw, B=B, M=C, Cin=C, K=C, Vq=V, Ncols=V, stats=st, ln=(g, bt), name="wgrad_ln_linear")
        gw2 = torch.empty(C, C, device=dev)
        gb2 = torch.empty(C, device=dev)
        PW._wgrad(gt, [x], gw2, B=B, M=C, Cin=C, K=C, Vq=V, Ncols=V, gbias=gb2, name="wgrad_linear")
        gx, gg, gb = PW._dgrad_lnbwd(gt, w, x, st, g, gadd)
        torch.cuda.synchronize()
        for k, v in (("wgrad_ln", gw), ("wgrad_plain", gw2), ("gbias", gb2), ("lnbwd_gx", gx.float()), ("lnbwd_ggamma", gg), ("lnbwd_gbeta", gb)):
            res.setdefault(k, []).append(v.clone())
    for k, v in res.items():
        d = max((v[0] - v[1]).abs().max().item(), (v[0] - v[2]).abs().max().item())
        print(f"{dt} {k:14s} max|run0 - run_i| = {d:.3e}  (max|v| {v[0].abs().max().item():.3e})")
