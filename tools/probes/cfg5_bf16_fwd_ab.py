"""BASELINE configs[4] model under bf16 autocast: forward activations of every stage with the round-5 forward fusions on vs off
(out-projection + MLP in one launch; block prologue inside the producing launch) — how far apart are two CORRECT bf16 evaluations
of the same network, stage by stage?  (tools/probes/cfg5_bf16_cosine.py: the deep-stage gradients of the two differ a lot.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch import nn
import factorizer_amd as ft
from factorizer_amd import pointwise as PW
DEV = "cuda:0"
torch.manual_seed(0)
S = (160, 192, 160)
model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=S, encoder_depth=(1,) * 5, encoder_width=(32, 64, 128, 256, 512),
                      strides=(1, 2, 2, 2, 2), decoder_depth=(1,) * 4, norm=ft.LayerNorm,
                      reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}), act=nn.ReLU, factorize=ft.NMF, rank=2,
                      num_iters=10, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0).to(DEV).eval()
x = torch.rand(1, 4, *S, device=DEV)
acts = {}
def hook(name):
    def f(m, i, o):
        acts[name] = (o[1] if isinstance(o, tuple) else o).detach().float().clone()
    return f
for i, b in enumerate(model.encoder.blocks):
    b.register_forward_hook(hook(f"enc{i}"))
for i, b in enumerate(model.decoder.blocks):
    b.register_forward_hook(hook(f"dec{i}"))
def run(on, amp):
    PW._OUTPROJ_MLP = on; PW._PRODUCER_PROLOGUE = on
    acts.clear()
    with torch.no_grad():
        if amp:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = model(x)
        else:
            y = model(x)
    acts["out"] = y.float().clone()
    return dict(acts)
f32 = run(False, False)
for amp in (True, False):
    a, b = run(True, amp), run(False, amp)
    for k in a:
        ref = f32[k]
        s = ref.abs().max().item()
        print("bf16" if amp else "fp32", k, "on-vs-off max %.2e" % ((a[k] - b[k]).abs().max().item() / s),
              "| on-vs-fp32 rms %.2e  off-vs-fp32 rms %.2e" % (((a[k] - ref).pow(2).mean().sqrt().item()) / s, ((b[k] - ref).pow(2).mean().sqrt().item()) / s))
