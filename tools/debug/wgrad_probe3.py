"""Weight gradients of the k2s2 (transposed) convolutions: chunking sweep (FZ_WGRAD_UNITS)."""
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
B = 2
for (C, O, S) in ((32, 64, 128), (64, 128, 64), (128, 256, 32), (256, 512, 16)):
    D = H = W = S
    Vc = (S // 2) ** 3
    x = torch.randn(B, C, D, H, W, device=DEV); gy = torch.randn(B, O, S // 2, S // 2, S // 2, device=DEV)
    gw = torch.empty(O, C, 2, 2, 2, device=DEV); gb = torch.empty(O, device=DEV)
    gwt = torch.empty(O, C, 2, 2, 2, device=DEV)
    res = []
    for u in ("512", "1024", "2048", "4096"):
        os.environ["FZ_WGRAD_UNITS"] = u
        t0 = timeit(lambda: PW._wgrad(gy, [x], gw, B=B, M=O, Cin=C, K=8 * C, Vq=D * H * W, Ncols=Vc, gbias=gb, loader=PW.LOAD_S2D,
                                      D=D, H=H, W=W, Ho=S // 2, Wo=S // 2, name="wgrad_conv_k2s2"))
        # transposed conv: roles swapped (x coarse with O channels is P; gy fine with C channels is Q)
        t1 = timeit(lambda: PW._wgrad(gy, [x], gwt, B=B, M=O, Cin=C, K=8 * C, Vq=D * H * W, Ncols=Vc, loader=PW.LOAD_S2D,
                                      D=D, H=H, W=W, Ho=S // 2, Wo=S // 2, name="wgrad_tconv_k2s2"))
        res.append(f"u{u}: {t0*1e3:.0f}/{t1*1e3:.0f}")
    print(f"conv {C}->{O} fine {S}^3 [us conv/tconv-like]: " + "  ".join(res))
