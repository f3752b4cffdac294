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
# fused input-gradient chain + both weight gradients (fz_mlp_chain mode 2) against the three launches it replaces
if Hd == 64:
    os.environ.pop("FZ_MLP_WGS", None)
    lbt = torch.randn(C, device=DEV) * 0.1
    x2, z1, st = PW._mlp_fwd_chain(x, lw, lbt, 1e-5, w1, b1, w2, b2)
    def unfused():
        gz1, gx1, gg, gb = PW._mlp_bwd_chain(g2, z1, w1, w2, x, st, lw)
        gw2 = torch.empty_like(w2); gb2 = torch.empty(C, device=DEV)
        PW._wgrad(g2, [z1], gw2, B=B, M=C, Cin=Hd, K=Hd, Vq=V, Ncols=V, gbias=gb2, qact=PW.ACT["gelu"], name="wgrad_linear")
        gw1 = torch.empty_like(w1); gb1 = torch.empty(Hd, device=DEV)
        PW._wgrad(gz1, [x], gw1, B=B, M=Hd, Cin=C, K=C, Vq=V, Ncols=V, gbias=gb1, stats=st, ln=(lw, lbt), name="wgrad_ln_linear")
    msu = timeit(unfused)
    for wgs in ("512", "384", "768"):
        os.environ["FZ_MLP_WG_WGS"] = wgs
        msf = timeit(lambda: PW._mlp_bwd_chain_wgrad(g2, z1, w1, w2, x, st, lw, lbt))
        print(f"MLP backward: chain + 2 wgrad launches {msu:.3f} ms ; fused (wgs={wgs}) {msf:.3f} ms ({5*P/msf/1e6:.0f} GB/s of 5P)")
