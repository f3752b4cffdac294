"""Which parameter gradients differ between two identical bf16-autocast runs of the cfg-5 model (and at which size)?"""
import sys

import torch
from torch import nn

sys.path.insert(0, ".")
import factorizer_amd as ft  # noqa: E402

dev = "cuda:0"


def run(S, widths, strides, amp):
    torch.manual_seed(0)
    model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=S, encoder_depth=(1,) * len(widths), encoder_width=widths,
                          strides=strides, decoder_depth=(1,) * (len(widths) - 1), norm=ft.LayerNorm,
                          reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}), act=nn.ReLU, factorize=ft.NMF,
                          rank=2, num_iters=10, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0).to(dev)
    x = torch.rand(1, 4, *S, device=dev)
    t = (torch.rand(1, 3, *S, device=dev) > 0.5).float()
    outs = []
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        if amp:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = ft.dice_ce_loss(model(x), t)
        else:
            loss = ft.dice_ce_loss(model(x), t)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((loss.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()}))
    (l1, g1), (l2, g2) = outs
    same = [n for n in g1 if torch.equal(g1[n], g2[n])]
    diff = [n for n in g1 if not torch.equal(g1[n], g2[n])]
    print(f"S={S} amp={amp} loss_equal={torch.equal(l1, l2)} same={len(same)} differ={len(diff)}")
    if diff:
        for n in reversed(list(g1)):
            if "decoder" in n or "head" in n:
                d = (g1[n] - g2[n]).abs().max().item() / (g1[n].abs().max().item() + 1e-30)
                print(f"   {'same  ' if torch.equal(g1[n], g2[n]) else 'DIFFER'} {d:.1e} {n}")


for S, widths, strides in (((80, 96, 80), (32, 64, 128, 256), (1, 2, 2, 2)),):
    run(S, widths, strides, True)
