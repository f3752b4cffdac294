"""wgrad kernels at stage-0/1 shapes: generic vs fast path."""
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
for (M, K, S) in ((32, 32, 128), (32, 64, 128), (64, 32, 128), (64, 64, 64), (64, 128, 64), (128, 128, 32)):
    V = S ** 3
    p = torch.randn(B, M, V, device=DEV); q = torch.randn(B, K, V, device=DEV)
    st = torch.rand(B, 2, V, device=DEV) + 0.5
    g, bt = torch.rand(K, device=DEV), torch.rand(K, device=DEV)
    gw = torch.empty(M, K, device=DEV); gb = torch.empty(M, device=DEV)
    nb = (p.numel() + q.numel()) * 4
    res = []
    for fast in ("0", "1"):
        os.environ["FZ_WGRAD_FAST"] = fast
        t0 = timeit(lambda: PW._wgrad(p, [q], gw, B=B, M=M, Cin=K, K=K, Vq=V, Ncols=V, gbias=gb))
        ref = gw.clone()
        t1 = timeit(lambda: PW._wgrad(p, [q], gw, B=B, M=M, Cin=K, K=K, Vq=V, Ncols=V, gbias=gb, stats=st, ln=(g, bt)))
        t2 = timeit(lambda: PW._wgrad(p, [q], gw, B=B, M=M, Cin=K, K=K, Vq=V, Ncols=V, gbias=gb, qact=2))
        res.append((t0, t1, t2, ref))
    d = (res[0][3] - res[1][3]).abs().max().item() / res[0][3].abs().max().item()
    print(f"{M:4d}x{K:4d} {S}^3: generic plain {res[0][0]:.3f} ln {res[0][1]:.3f} gelu {res[0][2]:.3f} | fast plain {res[1][0]:.3f} ({nb/res[1][0]/1e6:.0f} GB/s) ln {res[1][1]:.3f} gelu {res[1][2]:.3f}  reldiff {d:.1e}")
