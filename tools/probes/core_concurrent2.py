"""Mixed storage types: which of the two kernels must be the bf16 instantiation for the core backward to go wrong beside
a busy second stream?  Also: side stream running the library's OTHER LDS-heavy kernels."""
import sys

import torch

sys.path.insert(0, ".")
from factorizer_amd import functional as Fn  # noqa: E402
from factorizer_amd import pointwise as PW  # noqa: E402


class Ctx:
    saved_tensors = ()


dev = "cuda:0"
S, C = (40, 48, 40), 128
V = S[0] * S[1] * S[2]
side = torch.cuda.Stream()
torch.manual_seed(0)
geo = Fn.Geometry(C, S, 8, (5, 6, 5), [(0, 0, 0), (2, 3, 2)])
u0, v0 = torch.rand(8, 2, device=dev), torch.rand(150, 2, device=dev)
base_t = torch.rand(1, C, *S, device=dev)
base_ga = torch.randn(1, C, *S, device=dev)
base_p = torch.randn(1, 128, *S, device=dev)
base_q = torch.randn(1, 256, *S, device=dev)


def run(core_dt, side_dt, side_kind):
    t, ga = base_t.to(core_dt), base_ga.to(core_dt)
    p, q = base_p.to(side_dt), base_q.to(side_dt)
    w = torch.randn(128, 256, device=dev) / 16
    y = torch.empty(1, 128, *S, device=dev, dtype=side_dt)

    def core():
        c = Ctx()
        c.saved_tensors = (t, u0, v0)
        c.cfg = (geo, 10, 10, "hals", 1e-16, True)
        return Fn.FactCoreFn.backward(c, ga)[0]

    ref = core().float().clone()
    torch.cuda.synchronize()
    bad = 0
    for rep in range(8):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(6):
                if side_kind == "wgrad":
                    gw = torch.empty(128, 256, device=dev)
                    gb = torch.empty(128, device=dev)
                    PW._wgrad(p, [q], gw, B=1, M=128, Cin=256, K=256, Vq=V, Ncols=V, gbias=gb, qact=2)
                elif side_kind == "gemm":
                    PW._gemm([q], w, y, B=1, Cin=256, Vin=V, M=128, K=256, Ncol=V)
        out = core().float()
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
    print(f"core {str(core_dt):15s} side {side_kind:6s} {str(side_dt):15s}: {bad}/8 differ")


for cdt in (torch.bfloat16, torch.float32):
    for sdt in (torch.bfloat16, torch.float32):
        run(cdt, sdt, "wgrad")
run(torch.bfloat16, torch.bfloat16, "gemm")
run(torch.bfloat16, torch.float32, "gemm")
