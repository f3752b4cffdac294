"""Does the generic-patch core backward give run-to-run identical results while OTHER kernels run on a second stream?
Side kernels tried: none, a torch elementwise loop, a torch matmul loop, this library's weight-gradient kernel."""
import sys

import torch

sys.path.insert(0, ".")
from factorizer_amd import functional as Fn  # noqa: E402
from factorizer_amd import pointwise as PW  # noqa: E402


class Ctx:
    saved_tensors = ()


dev = "cuda:0"
S, C = (40, 48, 40), 128
dt = torch.bfloat16 if "--fp32" not in sys.argv else torch.float32
torch.manual_seed(0)
geo = Fn.Geometry(C, S, 8, (5, 6, 5), [(0, 0, 0), (2, 3, 2)])
t = torch.rand(1, C, *S, device=dev).to(dt)
ga = torch.randn(1, C, *S, device=dev).to(dt)
u0, v0 = torch.rand(8, 2, device=dev), torch.rand(150, 2, device=dev)
side = torch.cuda.Stream()
V = S[0] * S[1] * S[2]
p = torch.randn(1, 128, *S, device=dev).to(dt)
q = torch.randn(1, 256, *S, device=dev).to(dt)
big = torch.randn(4096, 4096, device=dev)
ew = torch.randn(64 * 1024 * 1024, device=dev)


def side_work(kind):
    with torch.cuda.stream(side):
        for _ in range(6):
            if kind == "elementwise":
                ew.mul_(1.0001)
            elif kind == "matmul":
                torch.mm(big, big)
            elif kind == "wgrad":
                gw = torch.empty(128, 256, device=dev)
                gb = torch.empty(128, device=dev)
                PW._wgrad(p, [q], gw, B=1, M=128, Cin=256, K=256, Vq=V, Ncols=V, gbias=gb, qact=2)


def core():
    c = Ctx()
    c.saved_tensors = (t, u0, v0)
    c.cfg = (geo, 10, 10, "hals", 1e-16, True)
    return Fn.FactCoreFn.backward(c, ga)[0]


ref = core().float().clone()
torch.cuda.synchronize()
for kind in ("none", "elementwise", "matmul", "wgrad"):
    bad = 0
    worst = 0.0
    for rep in range(12):
        torch.cuda.synchronize()
        if kind != "none":
            side_work(kind)
        out = core().float()
        torch.cuda.synchronize()
        nd = int((out != ref).sum())
        if nd:
            bad += 1
            worst = max(worst, ((out - ref).abs().max() / ref.abs().max()).item())
    print(f"{dt} side={kind:12s}: {bad}/12 runs differ from the solo result (worst rel {worst:.2e})")

# ---- is it the core kernel, or are its INPUTS being overwritten by the side kernel? ----
t0, ga0 = t.clone(), ga.clone()
torch.cuda.synchronize()
for rep in range(4):
    side_work("wgrad")
    torch.cuda.synchronize()
    print(f"after side wgrad #{rep}: t changed {int((t != t0).sum())} elements, ga changed {int((ga != ga0).sum())}, "
          f"p changed?, u0/v0 finite {bool(torch.isfinite(u0).all() and torch.isfinite(v0).all())}")
# the weight-gradient kernel alone, twice: deterministic?
outs = []
for rep in range(3):
    gw = torch.empty(128, 256, device=dev)
    gb = torch.empty(128, device=dev)
    PW._wgrad(p, [q], gw, B=1, M=128, Cin=256, K=256, Vq=V, Ncols=V, gbias=gb, qact=2)
    torch.cuda.synchronize()
    outs.append(gw.clone())
print("wgrad alone reproducible:", torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]))
# weight gradient on the side stream while the core runs on the main stream: which of the two outputs varies?
gws, cores = [], []
for rep in range(4):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        gw = torch.empty(128, 256, device=dev)
        gb = torch.empty(128, device=dev)
        PW._wgrad(p, [q], gw, B=1, M=128, Cin=256, K=256, Vq=V, Ncols=V, gbias=gb, qact=2)
    out = core().float()
    torch.cuda.synchronize()
    gws.append(gw.clone())
    cores.append(out.clone())
print("concurrent: wgrad output stable", all(torch.equal(gws[0], g) for g in gws), "| core output stable", all(torch.equal(cores[0], c) for c in cores),
      "| core == solo", torch.equal(cores[0], ref))
