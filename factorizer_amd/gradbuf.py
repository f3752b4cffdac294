"""Destinations for parameter gradients inside a flat gradient buffer.

The reference's multi-GPU recipe wraps the network in ``DistributedDataParallel`` (model_zoo/factorizer_brats23/configs/
train_multigpu.yaml:3-6), whose reducer copies every parameter gradient into a bucket buffer before the all-reduce
(``gradient_as_bucket_view=False``, the default).  Here the weight-gradient kernels write where the collective and the
optimizer read: `FlatAdamW` / `FlatGradSync` register, for every parameter, its slice of the flat buffer; the backward
of a layer asks `out_like(weight)` for the output of its weight-gradient launch and returns a fresh view of that slice,
which autograd's AccumulateGrad adopts as ``p.grad`` without a copy (nothing else references the view object).

A slice is handed out ONCE per `release()` (called by the owners' ``zero_grad``): a weight used twice in one backward,
or a second backward without ``zero_grad`` (gradient accumulation), gets an ordinary fresh tensor, which autograd then
ADDS to the slice in place — the accumulate semantics of torch are kept.  Gradients obtained through
``torch.autograd.grad`` while destinations are registered alias the buffer too: they are overwritten by the next
backward after ``zero_grad`` (clone them to keep them)."""
from __future__ import annotations

import weakref

import torch

_DST = {}   # data_ptr of the parameter -> _Entry


class _Entry:
    __slots__ = ("param", "_flat", "off", "n", "claimed")

    def __init__(self, param, flat, off):
        # both references are weak: the table must not keep the buffers of a dead model / optimizer allocated
        self.param, self._flat, self.off, self.n, self.claimed = weakref.ref(param), weakref.ref(flat), int(off), param.numel(), False

    @property
    def flat(self):
        return self._flat()


def register(flat: torch.Tensor, views: dict) -> None:
    """views: {parameter: its view inside `flat`}.  Re-registering a parameter replaces its entry; entries whose
    parameter has died or moved (``p.data`` re-pointed) are dropped."""
    _sweep()
    base = flat.storage_offset()
    for p, v in views.items():
        if p.dtype == flat.dtype and v.is_contiguous() and v.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr():
            _DST[p.data_ptr()] = _Entry(p, flat, v.storage_offset() - base)


def _sweep() -> None:
    for k in [k for k, e in _DST.items() if e.param() is None or e.flat is None or e.param().data_ptr() != k]:
        del _DST[k]


def release(flat: torch.Tensor = None) -> None:
    for e in _DST.values():
        if flat is None or e.flat is flat:
            e.claimed = False


def unregister(flat: torch.Tensor) -> None:
    for k in [k for k, e in _DST.items() if e.flat is flat]:
        del _DST[k]


def out_like(w: torch.Tensor, shape=None, dtype=None) -> torch.Tensor:
    """Output tensor for the gradient of the weight `w` (a parameter or a same-storage reshape of one): its slice of the
    registered flat buffer when one is free, else a fresh tensor."""
    shape = tuple(w.shape) if shape is None else tuple(shape)
    dtype = w.dtype if dtype is None else dtype
    e = _DST.get(w.data_ptr())
    flat = e.flat if e is not None else None
    if flat is None and e is not None:
        del _DST[w.data_ptr()]       # the buffer's owner is gone
    if flat is not None and not e.claimed and e.n == w.numel() and flat.device == w.device and flat.dtype == dtype:
        p = e.param()
        if p is not None and p.data_ptr() == w.data_ptr() and p.grad is None:
            e.claimed = True
            return flat[e.off:e.off + e.n].view(shape)
    return torch.empty(shape, dtype=dtype, device=w.device)
