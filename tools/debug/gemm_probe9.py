"""Stage-0 GEMMs (K = 32): register-resident kernel vs the streaming ring (FZ_GEMM_RESMAXK), plain / LN+ReLU / residual."""
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
B=2; S=128; V=S**3
for (Cin,Cout) in ((32,32),(32,64),(32,3)):
    x=torch.randn(B,Cin,V,device=DEV); w=torch.randn(Cout,Cin,device=DEV); b=torch.randn(Cout,device=DEV)
    y=torch.empty(B,Cout,V,device=DEV); z=torch.randn(B,Cout,V,device=DEV)
    g=torch.rand(Cin,device=DEV); bt=torch.rand(Cin,device=DEV); st=torch.empty(B,2,V,device=DEV)
    for mk in ("32","16"):
        os.environ["FZ_GEMM_RESMAXK"]=mk
        ms=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b))
        ms2=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,ln=(g,bt,1e-5),stats_out=st,eact=1))
        ms3=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b,res=z))
        print(f"{Cin}->{Cout} 128^3 resmaxk={mk}: plain {ms*1e3:.0f} us  ln+relu {ms2*1e3:.0f} us  bias+res {ms3*1e3:.0f} us")
