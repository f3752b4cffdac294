"""the fused MLP backward (fz_mlp_chain mode 2) alone at the stage-0 shape — for counter passes"""
import sys, torch
sys.path.insert(0, '.')
from factorizer_amd import pointwise as PW
DEV = 'cuda:0'
B, C, S, Hd = 2, 32, 128, 64
x = torch.randn(B, C, S, S, S, device=DEV); g2 = torch.randn_like(x)
lw, lb = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.1
w1, b1 = torch.randn(Hd, C, device=DEV) * 0.2, torch.randn(Hd, device=DEV) * 0.1
w2, b2 = torch.randn(C, Hd, device=DEV) * 0.2, torch.randn(C, device=DEV) * 0.1
x2, z1, st = PW._mlp_fwd_chain(x, lw, lb, 1e-5, w1, b1, w2, b2)
for _ in range(4):
    PW._mlp_bwd_chain_wgrad(g2, z1, w1, w2, x, st, lw, lb)
    PW._mlp_bwd_chain(g2, z1, w1, w2, x, st, lw)
torch.cuda.synchronize()
