"""Does any kernel of the library read a vector register or an LDS word it never wrote?  Every native launch of a whole
training step (forward + DiceCE + backward) is preceded, on the same stream, by tools/probes/poison.hip: all 512 vector
registers per lane and all 160 KB of LDS of every CU are left holding a bit pattern.  The step is run with three patterns
(quiet NaN, zero, FLT_MAX) and the loss / output / every parameter gradient are compared bit for bit.
usage: python tools/probes/poison_step.py [readme|cfg5] [f32|bf16]"""
import contextlib
import ctypes
import sys

import torch
from torch import nn

sys.path.insert(0, ".")
import factorizer_amd as ft  # noqa: E402
from factorizer_amd import functional as Fn  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "readme"
dt = sys.argv[2] if len(sys.argv) > 2 else "f32"
lib = ctypes.CDLL("tools/probes/bin/libpoison.so")
lib.poison.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
DEV = "cuda:0"
torch.manual_seed(0)
if which == "cfg5":
    S = (80, 96, 80)
    model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=S, encoder_depth=(1, 1, 1, 1), encoder_width=(32, 64, 128, 256),
                          strides=(1, 2, 2, 2), decoder_depth=(1, 1, 1), norm=ft.LayerNorm,
                          reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}), act=nn.ReLU, factorize=ft.NMF,
                          rank=2, num_iters=10, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0).to(DEV)
else:
    S = (64, 64, 64)
    model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=S, encoder_depth=(1, 1, 1, 1), encoder_width=(32, 64, 128, 256),
                          strides=(1, 2, 2, 2), decoder_depth=(1, 1, 1), norm=ft.LayerNorm,
                          reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU, factorize=ft.NMF,
                          rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0).to(DEV)
x = torch.rand(2, 4, *S, device=DEV)
t = (torch.rand(2, 3, *S, device=DEV) > 0.5).float()
sink = torch.zeros(4096, dtype=torch.int32, device=DEV)
pattern = [None]
count = [0]
orig = Fn._timed


def timed(name, nbytes, fn, cols=0, flops=0):
    if pattern[0] is not None:
        rc = lib.poison(pattern[0], 1024, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        count[0] += 1
    return orig(name, nbytes, fn, cols=cols, flops=flops)


Fn._timed = timed
import factorizer_amd.pointwise as PW  # noqa: E402
import factorizer_amd.losses as LS  # noqa: E402
ctx = (lambda: torch.autocast("cuda", dtype=torch.bfloat16)) if dt == "bf16" else contextlib.nullcontext


def step(pat):
    pattern[0] = pat
    count[0] = 0
    model.zero_grad(set_to_none=True)
    with ctx():
        y = model(x)
        loss = ft.dice_ce_loss(y, t)
    loss.backward()
    torch.cuda.synchronize()
    return y.detach().clone(), loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()}


ref = step(None)
names = {0x7FC00000: "quiet NaN", 0x00000000: "zero", 0x7F7FFFFF: "FLT_MAX", 0xFFFFFFFF: "all ones"}
for pat, label in names.items():
    y, loss, g = step(pat)
    bad = [n for n in g if not torch.equal(g[n], ref[2][n])]
    nonfinite = [n for n in g if not torch.isfinite(g[n]).all()]
    print(f"{which} {dt} pattern {label:9s} ({count[0]} poisoned launches): output equal {torch.equal(y, ref[0])}, loss equal "
          f"{torch.equal(loss, ref[1])}, gradients differing {len(bad)} / {len(g)}, non-finite {len(nonfinite)}"
          + (f"  first: {bad[:4]}" if bad else ""), flush=True)
