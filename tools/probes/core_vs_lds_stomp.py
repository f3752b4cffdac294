"""The generic-patch NMF backward (history in 73.6 KB of wave-private LDS per workgroup) on the main stream, a kernel that
only owns and rewrites X KB of LDS per workgroup on a second stream: does the backward's result change with X?"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from factorizer_amd import functional as Fn  # noqa: E402


class Ctx:
    saved_tensors = ()


lib = ctypes.CDLL("tools/probes/bin/liblds_stomp.so")
lib.lds_stomp.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dev = "cuda:0"
S, C = (40, 48, 40), 128
torch.manual_seed(0)
geo = Fn.Geometry(C, S, 8, (5, 6, 5), [(0, 0, 0), (2, 3, 2)])
u0, v0 = torch.rand(8, 2, device=dev), torch.rand(150, 2, device=dev)
t = torch.rand(1, C, *S, device=dev)
ga = torch.randn(1, C, *S, device=dev)
sink = torch.zeros(1 << 16, device=dev)
side = torch.cuda.Stream()


def core():
    c = Ctx()
    c.saved_tensors = (t, u0, v0)
    c.cfg = (geo, 10, 10, "hals", 1e-16, True)
    return Fn.FactCoreFn.backward(c, ga)[0]


ref = core().clone()
torch.cuda.synchronize()
for kb in (0, 8, 16, 32, 48, 64, 72, 80, 96, 128, 160):
    bad = 0
    nonfinite = 0
    for rep in range(6):
        torch.cuda.synchronize()
        if kb:
            rc = lib.lds_stomp(kb * 1024, 4096, 40, sink.data_ptr(), side.cuda_stream)
            assert rc == 0, rc
        out = core()
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
        nonfinite += int(not torch.isfinite(out).all())
    print(f"side kernel owning {kb:3d} KB of LDS per workgroup: core result differs in {bad}/6 runs (non-finite in {nonfinite})")
