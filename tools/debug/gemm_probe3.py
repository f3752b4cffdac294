import os, sys, torch
sys.path.insert(0,'.')
from factorizer_amd import pointwise as PW
DEV='cuda:0'
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters
B=2
for (Cin,Cout,V) in ((64,64,64**3),(64,128,64**3),(128,128,32**3),(128,256,32**3),(256,256,16**3),(256,512,16**3),(512,512,8**3),(512,1024,8**3),(1024,512,8**3)):
    x=torch.randn(B,Cin,V,device=DEV); w=torch.randn(Cout,Cin,device=DEV); b=torch.randn(Cout,device=DEV)
    y=torch.empty(B,Cout,V,device=DEV)
    nb=(x.numel()+y.numel())*4; fl=2*Cin*Cout*V*B
    ms=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b))
    gy=torch.randn(B,Cout,V,device=DEV); gw=torch.empty(Cout,Cin,device=DEV); gb=torch.empty(Cout,device=DEV)
    ms2=timeit(lambda: PW._wgrad(gy,[x],gw,B=B,M=Cout,Cin=Cin,K=Cin,Vq=V,Ncols=V,gbias=gb))
    print(f"{Cin}->{Cout} V={V}: gemm {ms*1e3:.0f} us {nb/ms/1e6:.0f} GB/s {fl/ms/1e9:.1f} TF | wgrad {ms2*1e3:.0f} us {fl/ms2/1e9:.1f} TF")
