"""What bounds the streaming GEMM: operand traffic or the fp32 MFMA pipe?  Same launches with the
diagnostics builds that issue half / none of the MFMAs (tools/debug/bin/libfz_probe_*.so)."""
import os, sys, torch
sys.path.insert(0,'.')
import factorizer_amd._native as N
if len(sys.argv) > 1 and sys.argv[1] != "full":
    N.LIB_PATH = os.path.abspath(f"tools/debug/bin/libfz_probe_{sys.argv[1]}.so")
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
out = []
for (Cin,Cout,S,B) in ((64,64,64,2),(64,64,64,16),(64,128,64,2),(128,64,64,2),(128,128,32,2),(128,128,32,16),(256,256,16,2),(256,256,16,16)):
    V=S**3
    x=torch.randn(B,Cin,V,device=DEV); w=torch.randn(Cout,Cin,device=DEV); b=torch.randn(Cout,device=DEV)
    y=torch.empty(B,Cout,V,device=DEV)
    nb=(x.numel()+y.numel())*4
    ms=timeit(lambda: PW._gemm([x],w,y,B=B,Cin=Cin,Vin=V,M=Cout,K=Cin,Ncol=V,bias=b))
    out.append(f"{Cin}->{Cout} {S}^3 B={B}: {ms*1e3:.1f} us ({nb/ms/1e6:.0f} GB/s)")
print(sys.argv[1:], " | ".join(out))
