import sys, torch, ctypes
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle import cpu_ref as O
import factorizer_amd as ft
R=4; solver='hals'
torch.manual_seed(10 + R)
x = torch.rand(37, 3, 8, 512); x[0,0].zero_(); x[1,1,:,:300]=0
u0, v0 = torch.rand(8, R), torch.rand(512, R)
gy = torch.rand_like(x)
nmf = ft.NMF(size=(8, 512), rank=R, num_iters=5, init="uniform", solver=solver)
nmf.load_state_dict({"init.u0": u0, "init.v0": v0}); nmf=nmf.to('cuda')
xd = x.cuda().requires_grad_(True)
y = nmf(xd); (gx,) = torch.autograd.grad(y, xd, gy.cuda()); gx=gx.cpu()
gxo = O.nmf_backward(x, u0, v0, gy, 5, solver)
gx64 = O.nmf_backward(x.double(), u0.double(), v0.double(), gy.double(), 5, solver).float()
kink = (gxo - gx64).abs().amax(dim=(-1, -2)); err = (gx - gx64).abs().amax(dim=(-1, -2)); scale = gx64.abs().amax(dim=(-1, -2))
bad = ~(err <= 1e-4 * scale + 1e-5 + 30 * kink)
for i,j in bad.nonzero().tolist(): print(i,j, 'err',err[i,j].item(), 'scale',scale[i,j].item(), 'kink',kink[i,j].item(), 'fwd err', (y.cpu()[i,j]-O.nmf_forward(x[i,j],u0,v0,5,solver)).abs().max().item())
print('nan?', torch.isnan(gx).any().item(), 'n bad', int(bad.sum()))
