"""upcat forward launch time at README size (events over 20 launches)"""
import sys, torch
sys.path.insert(0, ".")
from factorizer_amd import pointwise as PW
torch.manual_seed(0)
B, C, Cd = 2, 32, 64
skip = torch.randn(B, C, 128, 128, 128, device="cuda")
deep = torch.randn(B, Cd, 64, 64, 64, device="cuda")
w_t = torch.randn(Cd, C, 2, 2, 2, device="cuda") * 0.2
b_t = torch.randn(C, device="cuda") * 0.1
w_ad = torch.randn(C, 2 * C, 1, device="cuda") * 0.2
for _ in range(3):
    PW.up_cat_linear(skip, deep, w_t, b_t, w_ad, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    PW.up_cat_linear(skip, deep, w_t, b_t, w_ad, None)
e1.record()
torch.cuda.synchronize()
print("ms per call (compose + upcat):", e0.elapsed_time(e1) / 20)
