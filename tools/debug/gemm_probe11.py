"""Stage-1 MLP GEMMs of the BraTS-bundle model (mlp_ratio 4: hidden 256 at C = 64), isolated."""
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
for (Cin,Cout,S) in ((64,256,64),(256,64,64),(128,512,32),(512,128,32)):
    V=S**3
    x=torch.randn(B,Cin,V,device=DEV); w=torch.randn(Cout,Cin,device=DEV); b=torch.randn(Cout,device=DEV)
    y=torch.empty(B,Cout,V,device=DEV); z=torch.randn(B,Cout,V,device=DEV)
    g=torch.rand(Cin,device=DEV); bt=torch.rand(Cin,device=DEV); st=torch.empty(B,2,V,device=DEV)
    nb=(x.numel()+y.numel())*4; fl=2.0*Cin*Cout*V*B
    res=[]
    for cfg in (None,"42","41"):
        os.environ.pop("FZ_GEMM_CFG",None)
        if cfg: os.environ["FZ_GEMM_CFG"]=cfg
        ms=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b))
        ms2=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b,ln=(g,bt,1e-5),stats_out=st))
        ms3=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b,bact=2,res=z))
        res.append(f"{cfg or 'auto'}: {ms*1e3:.0f}/{ms2*1e3:.0f}/{ms3*1e3:.0f}")
    print(f"{Cin}->{Cout} {S}^3 (floor {nb/5.1e6:.0f} us traffic, {fl/150e6:.0f} us MFMA) [us plain/ln/gelu+res]: " + " | ".join(res))
