"""Asymptotic MFMA rate of the streaming GEMM: large K / M / N."""
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
for (Cin, Cout, V) in ((512, 512, 131072), (256, 256, 262144), (128, 128, 524288), (64, 64, 1048576), (128, 512, 262144), (1024, 512, 65536)):
    B = 1
    x = torch.randn(B, Cin, V, device=DEV); w = torch.randn(Cout, Cin, device=DEV) * 0.05
    y = torch.empty(B, Cout, V, device=DEV)
    ms = timeit(lambda: PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=Cout, K=Cin, Ncol=V))
    fl = 2 * Cin * Cout * V
    print(f"{Cin}->{Cout} V={V}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s  {(Cin+Cout)*V*4/ms/1e6:.0f} GB/s")
