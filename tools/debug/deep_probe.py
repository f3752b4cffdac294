"""Deep-stage (few columns, wide channels) layers: per-launch time of every kernel family."""
import os, sys, torch
sys.path.insert(0, '.')
from factorizer_amd import pointwise as PW
DEV = 'cuda:0'
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B = 2
for (Cin, Cout, S) in ((128, 128, 32), (128, 256, 32), (256, 256, 16), (256, 512, 16), (512, 512, 8), (512, 1024, 8), (1024, 512, 8)):
    V = S ** 3
    x = torch.randn(B, Cin, V, device=DEV); w = torch.randn(Cout, Cin, device=DEV) * 0.05; b = torch.randn(Cout, device=DEV)
    y = torch.empty(B, Cout, V, device=DEV); z = torch.randn(B, Cout, V, device=DEV)
    g = torch.rand(Cin, device=DEV); bt = torch.rand(Cin, device=DEV); st = torch.empty(B, 2, V, device=DEV)
    gy = torch.randn(B, Cout, V, device=DEV); gx = torch.empty_like(x); gw = torch.empty_like(w); gb = torch.empty(Cout, device=DEV)
    t_plain = timeit(lambda: PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=Cout, K=Cin, Ncol=V, bias=b))
    t_ln = timeit(lambda: PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=Cout, K=Cin, Ncol=V, bias=b, ln=(g, bt, 1e-5), stats_out=st))
    t_gelu = timeit(lambda: PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=Cout, K=Cin, Ncol=V, bias=b, bact=2, res=z))
    t_dg = timeit(lambda: PW._gemm([gy], w, gx, B=B, Cin=Cout, Vin=V, M=Cin, K=Cout, Ncol=V, w_t=True, ldw=Cin))
    t_wg = timeit(lambda: PW._wgrad(gy, [x], gw, B=B, M=Cout, Cin=Cin, K=Cin, Vq=V, Ncols=V, gbias=gb))
    t_lnb = timeit(lambda: PW._ln_backward(gx, x, st, g))
    fl = 2 * Cin * Cout * V * B
    print(f"{Cin:5d}->{Cout:5d} {S:3d}^3: plain {t_plain:6.1f} us ({fl/t_plain/1e6:5.1f} TF)  ln {t_ln:6.1f}  gelu+res {t_gelu:6.1f}  dgrad {t_dg:6.1f}  wgrad {t_wg:6.1f}  ln_bwd {t_lnb:6.1f}")
# k2s2 convs
import factorizer_amd as ft
for (Cin, Cout, S) in ((128, 256, 32), (256, 512, 16)):
    conv = ft.convs.Conv3d(Cin, Cout, 2, 2).to(DEV) if hasattr(ft, "convs") else None
    x = torch.randn(B, Cin, S, S, S, device=DEV, requires_grad=True)
    from factorizer_amd.convs import Conv3d, ConvTranspose3d
    c = Conv3d(Cin, Cout, kernel_size=2, stride=2).to(DEV)
    tc = ConvTranspose3d(Cout, Cin, kernel_size=2, stride=2).to(DEV)
    y = c(x); gy = torch.randn_like(y)
    t_f = timeit(lambda: c(x))
    def fb():
        yy = c(x); yy.backward(gy)
    t_fb = timeit(fb)
    xx = torch.randn(B, Cout, S // 2, S // 2, S // 2, device=DEV, requires_grad=True)
    t_tf = timeit(lambda: tc(xx))
    print(f"k2s2 {Cin}->{Cout} {S}^3: conv fwd {t_f:6.1f} us  fwd+bwd {t_fb:6.1f}  tconv fwd {t_tf:6.1f}")
