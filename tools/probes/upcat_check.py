"""upcat forward at README size vs the two-launch path, and run-to-run determinism"""
import os, sys, torch
sys.path.insert(0, ".")
from factorizer_amd import pointwise as PW
torch.manual_seed(0)
B, C, Cd = 2, 32, 64
S = (64, 64, 64)
skip = torch.randn(B, C, 128, 128, 128, device="cuda")
deep = torch.randn(B, Cd, *S, device="cuda")
w_t = torch.randn(Cd, C, 2, 2, 2, device="cuda") * 0.2
b_t = torch.randn(C, device="cuda") * 0.1
w_ad = torch.randn(C, 2 * C, 1, device="cuda") * 0.2
ys = []
for i in range(3):
    ys.append(PW.up_cat_linear(skip, deep, w_t, b_t, w_ad, None).clone())
torch.cuda.synchronize()
print("replay equal:", torch.equal(ys[0], ys[1]), torch.equal(ys[0], ys[2]))
PW._UPCAT_FWD = False
ref = PW.up_cat_linear(skip, deep, w_t, b_t, w_ad, None)
d = (ys[0] - ref).abs()
print("max abs diff vs two launches:", d.max().item(), "rel", (d.max() / ref.abs().max()).item())
bad = (d > 1e-4 * ref.abs().max()).nonzero()
print("bad elements:", bad.shape[0], bad[:5].tolist())
d01 = (ys[0] - ys[1]).abs()
print("replay diff elements:", int((d01 > 0).sum()), (d01 > 0).nonzero()[:5].tolist())
