"""Deep-stage GEMM shapes of the README model (B = 2): time, TFLOP/s, and how they compare with the fp32 MFMA peak."""
import os, sys, torch
sys.path.insert(0, '.')
from factorizer_amd import pointwise as PW
DEV = 'cuda:0'
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B = 2
for (Cin, Cout, S) in ((128, 128, 32), (128, 256, 32), (256, 128, 32), (256, 256, 16), (256, 512, 16), (512, 256, 16),
                       (512, 512, 8), (512, 1024, 8), (1024, 512, 8)):
    V = S ** 3
    x = torch.randn(B, Cin, V, device=DEV); w = torch.randn(Cout, Cin, device=DEV) * 0.05
    y = torch.empty(B, Cout, V, device=DEV)
    row = []
    for cfg in (None, "41", "21", "11"):
        for ks in (None, "4"):
            if cfg: os.environ["FZ_GEMM_CFG"] = cfg
            else: os.environ.pop("FZ_GEMM_CFG", None)
            if ks: os.environ["FZ_GEMM_KS"] = ks
            else: os.environ.pop("FZ_GEMM_KS", None)
            if (cfg is None) != (ks is None): continue
            ms = timeit(lambda: PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=Cout, K=Cin, Ncol=V))
            tf = 2.0 * Cin * Cout * V * B / 1e12
            row.append(f"{(cfg or 'dflt')}{'k4' if ks else ''}: {ms*1e3:6.1f} us {tf/ms*1e3:5.1f} TF")
    ideal = 2.0 * Cin * Cout * V * B / 157e12 * 1e6
    print(f"{Cin:>4}->{Cout:<4} {S}^3 (ideal {ideal:5.1f} us) | " + " | ".join(row), flush=True)
