import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import factorizer_amd as ft
from oracle import cpu_ref as O
DEV = "cuda:0"

def run(M, N, R, T, solver, lead, relu_input=False, G=None, seed=0):
    torch.manual_seed(seed)
    x = torch.randn(*lead, M, N).relu() if relu_input else torch.rand(*lead, M, N)
    u0, v0 = torch.rand(M, R), torch.rand(N, R)
    gy = torch.rand_like(x)
    nmf = ft.NMF(size=(M, N), rank=R, num_iters=T, num_grad_steps=G, init="uniform", solver=solver)
    nmf.load_state_dict({"init.u0": u0, "init.v0": v0})
    nd = nmf.to(DEV)
    xd = x.to(DEV).requires_grad_(True)
    y = nd(xd)
    (gx,) = torch.autograd.grad(y, xd, gy.to(DEV))
    gx64 = O.nmf_backward(x.double(), u0.double(), v0.double(), gy.double(), T, solver, G).float()
    gxo = O.nmf_backward(x, u0, v0, gy, T, solver, G)
    nm = x.numel() // (M * N)
    e_dev = (gx.cpu() - gx64).reshape(nm, -1).abs().amax(1) / gx64.reshape(nm, -1).abs().amax(1)
    e_orc = (gxo - gx64).reshape(nm, -1).abs().amax(1) / gx64.reshape(nm, -1).abs().amax(1)
    mar = O.hals_gate_margin(x, u0, v0, T).reshape(-1) if solver == "hals" else torch.zeros(nm)
    print(f"{M}x{N} R{R} T{T} G{G} {solver} lead{lead} relu_in={relu_input}: dev {[f'{v:.1e}' for v in e_dev.tolist()]} "
          f"oracle32 {[f'{v:.1e}' for v in e_orc.tolist()]} margin {[f'{v:.1e}' for v in mar.tolist()]}")

run(16, 4096, 1, 5, "hals", (4,))
run(16, 4096, 1, 5, "hals", (4,), relu_input=True)
run(16, 4096, 1, 5, "hals", (2, 2, 1))
run(16, 4096, 1, 4, "hals", (2,), relu_input=True)
run(16, 4096, 1, 5, "mu", (4,), relu_input=True)
for s in range(3):
    run(8, 1200, 2, 4, "hals", (3,), seed=s)
run(8, 1200, 2, 4, "hals", (3,), G=1)
run(8, 1200, 2, 4, "hals", (3,), G=2)
run(8, 1200, 2, 4, "hals", (3,), G=3)
run(8, 1200, 2, 1, "hals", (3,))
run(8, 1200, 2, 2, "hals", (3,))
