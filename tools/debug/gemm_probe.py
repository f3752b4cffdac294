import os, sys, torch
sys.path.insert(0,'.')
from factorizer_amd import pointwise as PW
import factorizer_amd
DEV='cuda:0'
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters
B,V=2,128**3
for (Cin,Cout) in ((32,32),(32,64),(64,32)):
    x=torch.randn(B,Cin,128,128,128,device=DEV); w=torch.randn(Cout,Cin,1,device=DEV); b=torch.randn(Cout,device=DEV)
    y=torch.empty(B,Cout,128,128,128,device=DEV)
    g=torch.rand(Cin,device=DEV); bt=torch.rand(Cin,device=DEV); st=torch.empty(B,2,V,device=DEV)
    nb=(x.numel()+y.numel())*4
    ms=timeit(lambda: PW._gemm([x],w.reshape(Cout,Cin),y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b))
    print(f"plain {Cin}->{Cout}: {ms:.3f} ms {nb/ms/1e6:.0f} GB/s")
    ms=timeit(lambda: PW._gemm([x],w.reshape(Cout,Cin),y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b,ln=(g,bt,1e-5),stats_out=st))
    print(f"ln    {Cin}->{Cout}: {ms:.3f} ms {nb/ms/1e6:.0f} GB/s")
    ms=timeit(lambda: PW._gemm([x],w.reshape(Cout,Cin),y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b,bact=2))
    print(f"gelu  {Cin}->{Cout}: {ms:.3f} ms {nb/ms/1e6:.0f} GB/s")
x=torch.randn(B,32,128,128,128,device=DEV); y=torch.empty_like(x)
ms=timeit(lambda: y.copy_(x)); print(f"copy: {ms:.3f} ms {2*x.numel()*4/ms/1e6:.0f} GB/s")
