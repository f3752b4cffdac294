"""fz_mlp_chain forward / backward at the stage-0 shape vs the unfused layers."""
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
B, C, S = 2, 32, 128
Hd = int(os.environ.get('HD', '64'))
V = S ** 3
x = torch.randn(B, C, S, S, S, device=DEV); g2 = torch.randn_like(x)
lw, lb = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
w1, b1 = torch.randn(Hd, C, device=DEV) * 0.2, torch.randn(Hd, device=DEV) * 0.1
w2, b2 = torch.randn(C, Hd, device=DEV) * 0.2, torch.randn(C, device=DEV) * 0.1
P = x.numel() * 4
for nacc in ("2",):
    for wgs in ("512", "768", "1024", "100000"):
        os.environ["FZ_MLP_NACC"] = nacc; os.environ["FZ_MLP_WGS"] = wgs
        x2, z1, st = PW._mlp_fwd_chain(x, lw, lb, 1e-5, w1, b1, w2, b2)
        ms = timeit(lambda: PW._mlp_fwd_chain(x, lw, lb, 1e-5, w1, b1, w2, b2))
        msb = timeit(lambda: PW._mlp_bwd_chain(g2, z1, w1, w2, x, st, lw))
        print(f"nacc={nacc} wgs={wgs:>6}: fwd {ms:.3f} ms ({4*P/ms/1e6:.0f} GB/s of 4P)  bwd {msb:.3f} ms ({(3+2*Hd//32)*P/msb/1e6:.0f} GB/s)")
