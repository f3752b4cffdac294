"""Replay determinism of the fused core backward (generic patch, bf16 / fp32) called twice on the same inputs."""
import sys

import torch

sys.path.insert(0, ".")
from factorizer_amd import functional as Fn  # noqa: E402


class Ctx:
    saved_tensors = ()

    def save_for_backward(self, *ts):
        self.saved_tensors = ts


dev = "cuda:0"
for S, C in (((20, 24, 20), 64), ((40, 48, 40), 64), ((80, 96, 80), 32)):
    for dt in (torch.bfloat16, torch.float32):
        torch.manual_seed(0)
        geo = Fn.Geometry(C, S, 8, (5, 6, 5), [(0, 0, 0), (2, 3, 2)])
        t = torch.rand(1, C, *S, device=dev).to(dt)
        ga = torch.randn(1, C, *S, device=dev).to(dt)
        u0, v0 = torch.rand(8, 2, device=dev), torch.rand(150, 2, device=dev)
        outs = []
        for rep in range(3):
            c = Ctx()
            c.saved_tensors = (t, u0, v0)
            c.cfg = (geo, 10, 10, "hals", 1e-16, True)
            # poison the allocator's next block so that stale memory differs between calls
            junk = torch.full_like(t, float(rep + 1))
            del junk
            gt = Fn.FactCoreFn.backward(c, ga)[0]
            torch.cuda.synchronize()
            outs.append(gt.float().clone())
        d01 = (outs[0] - outs[1]).abs().max().item()
        d02 = (outs[0] - outs[2]).abs().max().item()
        nz = int((outs[0] != outs[1]).sum())
        print(f"S={S} C={C} {dt}: max|d| {d01:.3e} {d02:.3e}  differing elements {nz} of {outs[0].numel()}  finite {bool(torch.isfinite(outs[0]).all())}")
        if nz:
            idx = (outs[0] != outs[1]).nonzero()[:6].tolist()
            print("   first differing indices (b, c, d, h, w):", idx)
