import sys, torch
sys.path.insert(0, '.')
from factorizer_amd import pointwise as PW
DEV = 'cuda:0'
Cin, Cout, V, B = 256, 256, 262144, 1
x = torch.randn(B, Cin, V, device=DEV); w = torch.randn(Cout, Cin, device=DEV) * 0.05
y = torch.empty(B, Cout, V, device=DEV)
for _ in range(5):
    PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=Cout, K=Cin, Ncol=V)
torch.cuda.synchronize()
