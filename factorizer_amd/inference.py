"""Sliding-window inference: the caller on the inference side of the hot path.

The BraTS bundle wraps the network in MONAI's `SlidingWindowInfererAdapt(roi_size=128^3,
sw_batch_size=2, overlap=0.5, mode="gaussian")` (model_zoo/factorizer_brats23/configs/
inference.yaml:96-102, train.yaml:206-212) because real volumes (240 x 240 x 155) are larger than
the training crop.  MONAI is a third-party dependency that is not under the reference tree; the
published algorithm of `monai.inferers.utils.sliding_window_inference` /
`compute_importance_map` is restated here with the same constructor arguments.  On device the
three data movements (window gather, weighted accumulate, final divide) are kernels of
libfactorizer_hip (csrc/sw_infer.hip); CPU tensors use the composed path.
"""
from __future__ import annotations

import itertools
import math
from typing import Callable, Sequence

import torch
import torch.nn.functional as F

from . import _native as N


def _tuple3(v, n):
    if isinstance(v, (int, float)):
        return (v,) * n
    v = tuple(v)
    if len(v) != n:
        raise ValueError(f"expected {n} values, got {v}")
    return v


def scan_interval(image_size, roi_size, overlap):
    """interval[i] = roi if roi == image else max(int(roi * (1 - overlap)), 1)"""
    out = []
    for im, r, o in zip(image_size, roi_size, overlap):
        out.append(int(r) if r == im else max(int(r * (1 - o)), 1))
    return tuple(out)


def window_starts(image_size, roi_size, interval):
    """Dense window origins, first axis slowest; the last window of an axis is pulled back so that
    it ends at the image border."""
    per_axis = []
    for im, r, it in zip(image_size, roi_size, interval):
        n = int(math.ceil(float(im - r) / it)) + 1 if it > 0 else 1
        starts = []
        for i in range(n):
            s = i * it
            s -= max(s + r - im, 0)
            starts.append(s)
        per_axis.append(starts)
    return list(itertools.product(*per_axis))


def gaussian_factors(roi_size, sigma_scale=0.125, dtype=torch.float32, device="cpu"):
    """The 1-D factors of the Gaussian importance map and the floor applied to their product:
    g_i(x) = exp(-x^2 / (2 (sigma_scale_i * roi_i)^2)), x centred on the patch;
    floor = max(min over the map of the product, 1e-3)."""
    sig = _tuple3(sigma_scale, len(roi_size))
    fac = []
    for r, s in zip(roi_size, sig):
        x = torch.arange(-(r - 1) / 2.0, (r - 1) / 2.0 + 1, dtype=torch.float32)[:r]
        fac.append(torch.exp(x ** 2 / (-2 * (r * s) ** 2)))
    mn = 1.0
    for f in fac:
        mn *= float(f.min())
    floor = max(mn, 1e-3)
    return [f.to(device=device, dtype=dtype) for f in fac], floor


class SlidingWindowInferer:
    """`inferer(inputs, network)` with MONAI's argument names.  `inputs` (B, C, D, H, W); the network
    maps (n, C, *roi) -> (n, C_out, *roi).  mode "gaussian" or "constant"."""

    def __init__(self, roi_size, sw_batch_size: int = 1, overlap=0.25, mode: str = "constant",
                 sigma_scale=0.125, padding_mode: str = "constant", cval: float = 0.0, **_ignored):
        if mode not in ("gaussian", "constant"):
            raise ValueError(f"unsupported blend mode {mode!r}")
        self.roi_size, self.sw_batch_size, self.overlap = roi_size, int(sw_batch_size), overlap
        self.mode, self.sigma_scale, self.padding_mode, self.cval = mode, sigma_scale, padding_mode, cval

    def __call__(self, inputs: torch.Tensor, network: Callable[[torch.Tensor], torch.Tensor]) -> torch.Tensor:
        return sliding_window_inference(inputs, self.roi_size, self.sw_batch_size, network, self.overlap, self.mode,
                                        self.sigma_scale, self.padding_mode, self.cval)


SlidingWindowInfererAdapt = SlidingWindowInferer  # the bundle's name; no OOM fallback is needed here


def sliding_window_inference(inputs, roi_size, sw_batch_size, predictor, overlap=0.25, mode="constant",
                             sigma_scale=0.125, padding_mode="constant", cval=0.0, _composed=False):
    nd = inputs.dim() - 2
    if nd != 3:
        raise ValueError("sliding_window_inference: 3-D volumes (B, C, D, H, W)")
    B = inputs.shape[0]
    orig = tuple(inputs.shape[2:])
    roi = tuple(int(r) if r and r > 0 else int(o) for r, o in zip(_tuple3(roi_size, nd), orig))
    ov = _tuple3(overlap, nd)
    # volumes smaller than the roi are padded symmetrically (remainder at the end), then cropped back
    pad = []
    for k in range(nd - 1, -1, -1):
        diff = max(roi[k] - orig[k], 0)
        half = diff // 2
        pad.extend([half, diff - half])
    if any(pad):
        inputs = F.pad(inputs, pad, mode=padding_mode, value=cval) if padding_mode == "constant" \
            else F.pad(inputs, pad, mode=padding_mode)
    size = tuple(inputs.shape[2:])
    starts = window_starts(size, roi, scan_interval(size, roi, ov))
    if mode == "gaussian":
        fac, floor = gaussian_factors(roi, sigma_scale, inputs.dtype, inputs.device)
    else:
        fac, floor = [torch.ones(r, dtype=inputs.dtype, device=inputs.device) for r in roi], 1.0
    # `_composed` (tests only): stitch a device tensor with framework ops instead of the kernels
    native = inputs.is_cuda and inputs.dtype == torch.float32 and roi[2] % 4 == 0 and not _composed
    inputs = inputs.contiguous()
    out = cnt = None
    jobs = [(b, s) for b in range(B) for s in starts]
    for j0 in range(0, len(jobs), sw_batch_size):
        chunk = jobs[j0:j0 + sw_batch_size]
        if native:
            win = torch.empty((len(chunk), inputs.shape[1], *roi), dtype=inputs.dtype, device=inputs.device)
            for i, (b, s) in enumerate(chunk):
                N.check(N.lib().fz_sw_gather(inputs[b].data_ptr(), win[i].data_ptr(), inputs.shape[1], *size, *roi, *s,
                                             N.stream_ptr(inputs)), "fz_sw_gather")
        else:
            win = torch.stack([inputs[b, :, s[0]:s[0] + roi[0], s[1]:s[1] + roi[1], s[2]:s[2] + roi[2]]
                               for b, s in chunk])
        prob = predictor(win)
        if isinstance(prob, (tuple, list)):
            prob = prob[0]
        if tuple(prob.shape[2:]) != roi:
            raise ValueError("sliding_window_inference: the network must keep the window size")
        prob = prob.contiguous()
        if out is None:
            out = torch.zeros((B, prob.shape[1], *size), dtype=prob.dtype, device=prob.device)
            cnt = torch.zeros((B, *size), dtype=prob.dtype, device=prob.device)
        for i, (b, s) in enumerate(chunk):
            if native and prob.is_cuda and prob.dtype == torch.float32:
                N.check(N.lib().fz_sw_accumulate(prob[i].data_ptr(), out[b].data_ptr(), cnt[b].data_ptr(),
                                                 fac[0].data_ptr(), fac[1].data_ptr(), fac[2].data_ptr(), float(floor),
                                                 prob.shape[1], *size, *roi, *s, N.stream_ptr(prob)),
                        "fz_sw_accumulate")
            else:
                w = (fac[0][:, None, None] * fac[1][None, :, None] * fac[2][None, None, :]).clamp_min(floor)
                sl = (slice(s[0], s[0] + roi[0]), slice(s[1], s[1] + roi[1]), slice(s[2], s[2] + roi[2]))
                out[(b, slice(None)) + sl] += w * prob[i]
                cnt[(b,) + sl] += w
    V = size[0] * size[1] * size[2]
    if native and out.is_cuda and out.dtype == torch.float32:
        for b in range(B):
            N.check(N.lib().fz_sw_finalize(out[b].data_ptr(), cnt[b].data_ptr(), out.shape[1], V, N.stream_ptr(out)),
                    "fz_sw_finalize")
    else:
        out = out / cnt[:, None]
    if any(pad):
        crop = []
        for k in range(nd):
            lo = pad[2 * (nd - 1 - k)]
            crop.append(slice(lo, lo + orig[k]))
        out = out[(slice(None), slice(None)) + tuple(crop)]
    return out
