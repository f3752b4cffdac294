"""Segmentation loss of the training step: the reference bundle's
``DiceCELoss(sigmoid=True, squared_pred=True)`` (model_zoo/factorizer_brats23/configs/train.yaml:67-70).
MONAI (pinned 1.4.0, docs/requirements.txt:11) is third-party and absent here; its published
evaluation is restated:

* Dice term: ``DiceLoss(sigmoid=True, squared_pred=True)`` — per (b, c) plane
  ``1 − (2 Σ p t + 1e-5) / (Σ p² + Σ t² + 1e-5)`` with ``p = sigmoid(z)``, mean over (b, c);
* cross-entropy term: for **one** prediction channel ``BCEWithLogitsLoss``; for C > 1 channels (the
  3-channel BraTS head) ``nn.CrossEntropyLoss`` over the channel softmax with the float multi-label
  target as class probabilities: ``mean_{b,v} −Σ_c t_c · log_softmax(z)_c`` — `dice_ce_loss`.

`dice_bce_loss` (sigmoid Dice + per-channel BCE for any C) is NOT the recipe's objective for C > 1; it
is kept as a separate, documented loss.

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
                z.data_ptr(), t.data_ptr(), part.data_ptr(), planes, V, N.stream_ptr(z)), cols=z.shape[0] * V)
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
                1.0 / (planes * V), gs.data_ptr(), N.stream_ptr(z)), cols=z.shape[0] * V)
        N.check(rc, "fz_dice_bce_grad")
        return gz, None, None


def _fp32_logits(logits, target):
    """The loss is evaluated in fp32 whatever the storage type of the network's activations (the head of a
    mixed-precision run emits bf16 logits: one cast of a (B, out_channels, V) tensor)."""
    if logits.dtype in (torch.bfloat16, torch.float16):
        logits = logits.float()
    if target.dtype != logits.dtype:
        target = target.to(logits.dtype)
    return logits, target


def dice_bce_loss(logits, target, smooth: float = 1e-5):
    logits, target = _fp32_logits(logits, target)
    planes = logits.shape[0] * logits.shape[1]
    if logits.is_cuda and logits.numel() and logits.dtype == torch.float32 and target.dtype == torch.float32 \
            and (logits.numel() // planes) % 4 == 0:
        return DiceBCEFn.apply(logits, target, smooth)
    return dice_bce_loss_composed(logits, target, smooth)


# ---- the recipe's loss: DiceCELoss(sigmoid=True, squared_pred=True), MONAI 1.4 semantics -------------------
def _dice_term(logits, target, smooth):
    p = torch.sigmoid(logits)
    dims = tuple(range(2, logits.ndim))
    inter = (p * target).sum(dims)
    den = (p * p).sum(dims) + (target * target).sum(dims)
    return (1.0 - (2.0 * inter + smooth) / (den + smooth)).mean()


def dice_ce_loss_composed(logits, target, smooth: float = 1e-5):
    if logits.shape[1] == 1:
        return dice_bce_loss_composed(logits, target, smooth)
    target = target.to(logits.dtype)
    return _dice_term(logits, target, smooth) + F.cross_entropy(logits, target)


class DiceCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, smooth):
        z = logits.contiguous()
        t = target.contiguous()
        B, C = z.shape[:2]
        V = z.numel() // (B * C)
        lib = N.lib()
        nchunk = lib.fz_dice_bce_chunks(V)
        part = torch.empty((B, nchunk, 3 * C + 1), dtype=z.dtype, device=z.device)
        with torch.cuda.device(z.device):
            rc = Fn._timed("dice_ce_sums", 8 * z.numel(), lambda: lib.fz_dice_ce_sums(
                z.data_ptr(), t.data_ptr(), part.data_ptr(), B, C, V, N.stream_ptr(z)), cols=B * V)
        N.check(rc, "fz_dice_ce_sums")
        if B <= 8:   # one finish launch instead of a dozen framework kernels (chunk sum, num, den, means)
            loss = torch.empty((), dtype=z.dtype, device=z.device)
            coef = torch.empty((B * C, 2), dtype=z.dtype, device=z.device)
            with torch.cuda.device(z.device):
                rc = lib.fz_dice_ce_finish(part.data_ptr(), B, C, V, float(smooth), loss.data_ptr(), coef.data_ptr(), N.stream_ptr(z))
            N.check(rc, "fz_dice_ce_finish")
        else:
            s = part.sum(dim=1)  # (B, 3C+1) — tiny
            d = s[:, :3 * C].reshape(B, C, 3)
            num = 2.0 * d[..., 0] + smooth
            den = d[..., 1] + d[..., 2] + smooth
            loss = (1.0 - num / den).mean() + s[:, 3 * C].sum() / (B * V)
            coef = torch.stack([num, den], dim=-1).reshape(B * C, 2).contiguous()
        ctx.save_for_backward(z, t, coef)
        ctx.dims = (B, C, V)
        return loss

    @staticmethod
    def backward(ctx, g):
        z, t, coef = ctx.saved_tensors
        B, C, V = ctx.dims
        gz = torch.empty_like(z)
        gs = g.reshape(1).to(z.dtype).contiguous()
        with torch.cuda.device(z.device):
            rc = Fn._timed("dice_ce_grad", 12 * z.numel(), lambda: N.lib().fz_dice_ce_grad(
                z.data_ptr(), t.data_ptr(), coef.data_ptr(), gz.data_ptr(), B, C, V, 1.0 / (B * C),
                1.0 / (B * V), gs.data_ptr(), N.stream_ptr(z)), cols=B * V)
        N.check(rc, "fz_dice_ce_grad")
        return gz, None, None


def dice_ce_loss(logits, target, smooth: float = 1e-5):
    """``DiceCELoss(sigmoid=True, squared_pred=True)(logits, target)`` of the training recipe."""
    B, C = logits.shape[:2]
    if C == 1:
        return dice_bce_loss(logits, target, smooth)
    logits, target = _fp32_logits(logits, target)
    if logits.is_cuda and logits.numel() and logits.dtype == torch.float32 and 2 <= C <= 8 \
            and (logits.numel() // (B * C)) % 4 == 0:
        return DiceCEFn.apply(logits, target, smooth)
    if logits.is_cuda and logits.numel():
        from .composed import warn_once
        warn_once("dice_ce_loss", f"no native kernel for shape {tuple(logits.shape)} / {logits.dtype}: composed framework ops")
    return dice_ce_loss_composed(logits, target, smooth)


class DiceCELoss(torch.nn.Module):
    """Drop-in for the recipe's ``DiceCELoss(sigmoid=True, squared_pred=True)`` (train.yaml:67-70)."""

    def __init__(self, sigmoid: bool = True, squared_pred: bool = True, smooth_nr: float = 1e-5,
                 smooth_dr: float = 1e-5):
        super().__init__()
        if not (sigmoid and squared_pred) or smooth_nr != smooth_dr:
            raise NotImplementedError("DiceCELoss: only the recipe's form (sigmoid=True, squared_pred=True, "
                                      "smooth_nr == smooth_dr) is implemented")
        self.smooth = float(smooth_nr)

    def forward(self, input, target):
        return dice_ce_loss(input, target, self.smooth)
