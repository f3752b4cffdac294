"""Streaming GEMM: does the launcher's tile choice match the best of a sweep (FZ_GEMM_CFG = <nacc><mb>, FZ_GEMM_KS)?"""
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
B=2
shapes=[(64,64,64),(64,128,64),(128,64,64),(128,128,32),(128,256,32),(256,128,32),(256,256,16),(256,512,16),(512,256,16),(512,512,8),(512,1024,8),(1024,512,8)]
for (Cin,Cout,S) in shapes:
    V=S**3
    x=torch.randn(B,Cin,V,device=DEV); w=torch.randn(Cout,Cin,device=DEV); b=torch.randn(Cout,device=DEV)
    y=torch.empty(B,Cout,V,device=DEV)
    g=torch.rand(Cin,device=DEV); bt=torch.rand(Cin,device=DEV); st=torch.empty(B,2,V,device=DEV)
    res=[]
    for cfg,ks in ((None,None),("42","1"),("41","1"),("21","1"),("11","1"),("21","4"),("11","4")):
        for k in ("FZ_GEMM_CFG","FZ_GEMM_KS"): os.environ.pop(k,None)
        if cfg: os.environ["FZ_GEMM_CFG"]=cfg; os.environ["FZ_GEMM_KS"]=ks
        ms=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b))
        ms2=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b,ln=(g,bt,1e-5),stats_out=st))
        res.append(f"{cfg or 'auto'}/k{ks or '-'} {ms*1e3:.0f}/{ms2*1e3:.0f}")
    print(f"{Cin}->{Cout} {S}^3 [us plain/ln]: " + " | ".join(res))
