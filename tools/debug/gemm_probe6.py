"""Does the streaming GEMM reach its large-grid rate on the mid stages?  Same shapes, growing batch."""
import os, sys, torch
sys.path.insert(0,'.')
from factorizer_amd import pointwise as PW
DEV='cuda:0'
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters
for (Cin,Cout,S) in ((64,64,64),(64,128,64),(128,64,64),(128,128,32),(128,256,32),(256,128,32),(256,256,16)):
    for B in (2, 8, 32):
        V=S**3
        x=torch.randn(B,Cin,V,device=DEV); w=torch.randn(Cout,Cin,device=DEV); b=torch.randn(Cout,device=DEV)
        y=torch.empty(B,Cout,V,device=DEV); z=torch.randn(B,Cout,V,device=DEV)
        g=torch.rand(Cin,device=DEV); bt=torch.rand(Cin,device=DEV); st=torch.empty(B,2,V,device=DEV)
        nb=(x.numel()+y.numel())*4
        ms=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b))
        ms2=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b,ln=(g,bt,1e-5),stats_out=st))
        fl=2.0*Cin*Cout*V*B
        print(f"{Cin}->{Cout} {S}^3 B={B}: plain {ms*1e3:.1f} us ({nb/ms/1e6:.0f} GB/s, {fl/ms/1e9:.1f} TF) ln {ms2*1e3:.1f} us ({nb/ms2/1e6:.0f} GB/s)")
