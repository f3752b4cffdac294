"""stem Conv3d(4->32, k3) forward / weight gradient at 128^3, B = 2"""
import sys, torch
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
x = torch.randn(2, 4, 128, 128, 128, device=DEV, requires_grad=True)
w = (torch.randn(32, 4, 3, 3, 3, device=DEV) * 0.1).requires_grad_(True)
y = PW.ConvK3Fn.apply(x.detach(), w, None)
gy = torch.randn_like(y)
print(f"fwd {timeit(lambda: PW.ConvK3Fn.apply(x.detach(), w, None)):.3f} ms")
def fb():
    yy = PW.ConvK3Fn.apply(x.detach(), w, None)
    torch.autograd.grad(yy, w, gy)
t = timeit(fb)
print(f"fwd+wgrad {t:.3f} ms")
