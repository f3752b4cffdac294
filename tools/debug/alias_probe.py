import os, sys, torch
sys.path.insert(0,'.')
from factorizer_amd import pointwise as PW
from factorizer_amd import _native as N
DEV='cuda:0'
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters
B=2
for V in (2097152, 2097152+512):
    x=torch.randn(B,32,V,device=DEV); gy=torch.randn(B,32,V,device=DEV); gz=torch.randn(B,64,V,device=DEV)
    gw=torch.empty(32,32,device=DEV); gb=torch.empty(32,device=DEV); gw2=torch.empty(64,32,device=DEV); gb2=torch.empty(64,device=DEV)
    ms=timeit(lambda: PW._wgrad(gy,[x],gw,B=B,M=32,Cin=32,K=32,Vq=V,Ncols=V,gbias=gb))
    ms2=timeit(lambda: PW._wgrad(gz,[x],gw2,B=B,M=64,Cin=32,K=32,Vq=V,Ncols=V,gbias=gb2))
    # ln_bwd
    st=torch.rand(B,2,V,device=DEV)+0.5; g=torch.rand(32,device=DEV); gx=torch.empty_like(x)
    ms3=timeit(lambda: PW._ln_backward(gy,x,st,g,gadd=gy))
    # dgrad_lnbwd 32->32
    w=torch.randn(32,32,device=DEV)
    ms4=timeit(lambda: PW._dgrad_lnbwd(gy,w,x,st,g,gy))
    ms5=timeit(lambda: PW._dgrad_lnbwd(gz,torch.randn(64,32,device=DEV),x,st,g,gy))
    print(f"V={V}: wgrad32x32 {ms:.3f} ({2*x.numel()*4/ms/1e6:.0f} GB/s) wgrad64x32 {ms2:.3f} ({3*x.numel()*4/ms2/1e6:.0f}) ln_bwd32 {ms3:.3f} dgrad_lnbwd32 {ms4:.3f} ({4*x.numel()*4/ms4/1e6:.0f}) dgrad_lnbwd64 {ms5:.3f} ({5*x.numel()*4/ms5/1e6:.0f})")
