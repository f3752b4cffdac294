"""Plane-stride aliasing: the stage-0 kernels on V = 2^21 voxels vs V = 2^21 + 512 (same work +0.02 %)."""
import os, sys, torch
sys.path.insert(0, '.')
from factorizer_amd import pointwise as PW
DEV = 'cuda:0'
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B, C, Hd = 2, 32, 64
for V in (2097152, 2097152 + 512, 2097152 + 2048 + 64):
    x = torch.randn(B, C, V, device=DEV); gy = torch.randn(B, C, V, device=DEV); z = torch.randn(B, C, V, device=DEV)
    w = torch.randn(C, C, device=DEV) * 0.2; b = torch.randn(C, device=DEV)
    y = torch.empty_like(x)
    lw, lb = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
    w1, b1 = torch.randn(Hd, C, device=DEV) * 0.2, torch.randn(Hd, device=DEV) * 0.1
    w2, b2 = torch.randn(C, Hd, device=DEV) * 0.2, torch.randn(C, device=DEV) * 0.1
    st = torch.empty(B, 2, V, device=DEV)
    t_plain = timeit(lambda: PW._gemm([x], w, y, B=B, Cin=C, Vin=V, M=C, K=C, Ncol=V, bias=b))
    t_res = timeit(lambda: PW._gemm([x], w, y, B=B, Cin=C, Vin=V, M=C, K=C, Ncol=V, bias=b, res=z))
    t_ln = timeit(lambda: PW._gemm([x], w, y, B=B, Cin=C, Vin=V, M=C, K=C, Ncol=V, ln=(lw, lb, 1e-5), stats_out=st, eact=1))
    x2, z1, st2 = PW._mlp_fwd_chain(x, lw, lb, 1e-5, w1, b1, w2, b2)
    t_cf = timeit(lambda: PW._mlp_fwd_chain(x, lw, lb, 1e-5, w1, b1, w2, b2))
    t_cb = timeit(lambda: PW._mlp_bwd_chain(gy, z1, w1, w2, x, st2, lw))
    gw = torch.empty(C, C, device=DEV); gb = torch.empty(C, device=DEV)
    t_wg = timeit(lambda: PW._wgrad(gy, [x], gw, B=B, M=C, Cin=C, K=C, Vq=V, Ncols=V, gbias=gb))
    gw2 = torch.empty(C, Hd, device=DEV)
    t_wg2 = timeit(lambda: PW._wgrad(gy, [z1], gw2, B=B, M=C, Cin=Hd, K=Hd, Vq=V, Ncols=V, gbias=gb, qact=2))
    t_dl = timeit(lambda: PW._dgrad_lnbwd(gy, w, x, st2, lw, gy))
    print(f"V={V}: plain {t_plain:.3f} res {t_res:.3f} ln {t_ln:.3f} chain_fwd {t_cf:.3f} chain_bwd {t_cb:.3f} wgrad32x32 {t_wg:.3f} wgrad32x64 {t_wg2:.3f} dgrad_lnbwd {t_dl:.3f}")
