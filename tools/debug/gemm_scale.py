"""Stage-1 GEMM shapes (64^3 voxels per sample) at growing batch: does the time per sample fall when the grid runs
in several rounds (lock-step phases of a single resident round) or not?"""
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
    return s.elapsed_time(e) / iters
V = 64 ** 3
for (Cin, Cout) in ((64, 64), (64, 128), (128, 64), (128, 128)):
    for B in (1, 2, 4, 8, 16):
        x = torch.randn(B, Cin, V, device=DEV); w = torch.randn(Cout, Cin, device=DEV) * 0.05
        y = torch.empty(B, Cout, V, device=DEV)
        row = []
        for cfg in (None, "41", "21", "11"):
            if cfg: os.environ["FZ_GEMM_CFG"] = cfg
            else: os.environ.pop("FZ_GEMM_CFG", None)
            ms = timeit(lambda: PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=Cout, K=Cin, Ncol=V))
            gb = (Cin + Cout) * 4 * V * B / 1e9
            tf = 2.0 * Cin * Cout * V * B / 1e12
            row.append(f"{cfg or 'dflt'}: {ms*1e3/B:6.1f} us/sample {gb/ms:5.2f} TB/s {tf/ms*1e3:5.1f} TF")
        print(f"{Cin:>3}->{Cout:<3} B={B:<2} | " + " | ".join(row), flush=True)
        del x, y
