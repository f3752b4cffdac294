import os, sys, torch
sys.path.insert(0,'.')
import factorizer_amd as ft
from factorizer_amd import functional as Fn
DEV='cuda:0'
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters
m = ft.SWMatricize((None,32,128,128,128), head_dim=8, patch_size=8)
geo=m.geometry
t=torch.rand(2,32,128,128,128,device=DEV); u0=torch.rand(8,1,device=DEV); v0=torch.rand(512,1,device=DEV)
ga=torch.rand_like(t)
U=t.numel()*4
for T in (0,1,5):
    def fwd():
        ctx=type('X',(),{'save_for_backward':lambda self,*a:None})()
        return Fn.FactCoreFn.forward(ctx,t,u0,v0,geo,T,T,'hals',1e-16,True)
    ms=timeit(fwd); print(f"T={T} cf fwd (2 windows): {ms:.3f} ms  {5*U/ms/1e6:.0f} GB/s")
    xm=torch.rand(65536,8,512,device=DEV)
    ms=timeit(lambda: Fn._nmf_fwd_raw(xm,u0,v0,T,'hals',1e-16)); print(f"T={T} standalone fwd 65536 mats: {ms:.3f} ms {2*xm.numel()*4/ms/1e6:.0f} GB/s")
