"""Segmentation loss of the training step: soft Dice (sigmoid, squared denominators) + BCE with
logits — the form of the reference bundle's ``DiceCELoss(sigmoid=True, squared_pred=True)``
(model_zoo/factorizer_brats23/configs/train.yaml:67-70; MONAI itself is not importable here, so
the reduction conventions below are this build's and the loss is used for timing only).

Device tensors: one fused reduction pass + one fused gradient pass (csrc/loss.hip); CPU tensors:
composed ATen ops.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _native as N
from . import functional as Fn


def dice_bce_loss_composed(logits, target, smooth: float = 1e-5):
    p = torch.sigmoid(logits)
    dims = tuple(range(2, logits.ndim))
    inter = (p * target).sum(dims)
    den = (p * p).sum(dims) + (target * target).sum(dims)
    dice = 1.0 - (2.0 * inter + smooth) / (den + smooth)
    return dice.mean() + F.binary_cross_entropy_with_logits(logits, target)


class DiceBCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, smooth):
        z = logits.contiguous()
        t = target.contiguous()
        planes = z.shape[0] * z.shape[1]
        V = z.numel() // planes
        lib = N.lib()
        nchunk = lib.fz_dice_bce_chunks(V)
        part = torch.empty((planes, nchunk, 4), dtype=z.dtype, device=z.device)
        with torch.cuda.device(z.device):
            rc = Fn._timed("dice_bce_sums", 8 * z.numel(), lambda: lib.fz_dice_bce_sums(
                z.data_ptr(), t.data_ptr(), part.data_ptr(), planes, V, N.stream_ptr(z)))
        N.check(rc, "fz_dice_bce_sums")
        s = part.sum(dim=1)  # (planes, 4) — tiny
        num = 2.0 * s[:, 0] + smooth
        den = s[:, 1] + s[:, 2] + smooth
        loss = (1.0 - num / den).mean() + s[:, 3].sum() / (planes * V)
        ctx.save_for_backward(z, t, torch.stack([num, den], dim=1).contiguous())
        ctx.dims = (planes, V)
        return loss

    @staticmethod
    def backward(ctx, g):
        z, t, coef = ctx.saved_tensors
        planes, V = ctx.dims
        gz = torch.empty_like(z)
        gs = g.reshape(1).to(z.dtype).contiguous()
        with torch.cuda.device(z.device):
            rc = Fn._timed("dice_bce_grad", 12 * z.numel(), lambda: N.lib().fz_dice_bce_grad(
                z.data_ptr(), t.data_ptr(), coef.data_ptr(), gz.data_ptr(), planes, V, 1.0 / planes,
                1.0 / (planes * V), gs.data_ptr(), N.stream_ptr(z)))
        N.check(rc, "fz_dice_bce_grad")
        return gz, None, None


def dice_bce_loss(logits, target, smooth: float = 1e-5):
    planes = logits.shape[0] * logits.shape[1]
    if logits.is_cuda and logits.numel() and logits.dtype == torch.float32 and target.dtype == torch.float32 \
            and (logits.numel() // planes) % 4 == 0:
        return DiceBCEFn.apply(logits, target, smooth)
    return dice_bce_loss_composed(logits, target, smooth)
