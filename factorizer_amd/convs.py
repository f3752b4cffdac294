"""Conv3d / ConvTranspose3d parameter containers whose forward runs the native GEMM-family
kernels for the shapes the Factorizer U-shape uses (unet.py:53,123,231,253):
kernel 2 / stride 2 (space-to-depth GEMM), transposed kernel 2 / stride 2 (GEMM +
depth-to-space), kernel 3 / padding 1 (stem) and kernel 1 (head).  They subclass the torch
modules, so initialisation and state_dict keys are those of the reference's nn.Conv3d."""
from __future__ import annotations

import torch
from torch import nn

from . import composed
from . import pointwise as PW


def _all(v, n):
    return tuple(v) == (n,) * len(v)


class Conv3d(nn.Conv3d):
    def _k2s2_native(self, x) -> bool:
        return (x.is_cuda and x.numel() and x.dtype in (torch.float32, torch.bfloat16) and self.weight.dtype == torch.float32 and x.dim() == 5 and self.groups == 1
                and _all(self.dilation, 1) and self.padding_mode == "zeros"
                and _all(self.kernel_size, 2) and _all(self.stride, 2) and _all(self.padding, 0)
                and x.shape[-1] % 4 == 0 and x.shape[-2] % 2 == 0 and x.shape[-3] % 2 == 0
                and (x.shape[2] * x.shape[3] * x.shape[4] // 8) % 4 == 0 and self.in_channels % 2 == 0)

    def forward_fork(self, x):
        """``(x, self(x))`` for an ``x`` that is also kept as a skip connection: on the native path both
        come out of one autograd node whose backward adds the skip gradient inside the input-gradient
        kernel (PW.SkipConvK2S2Fn) instead of leaving the sum to a separate accumulation pass."""
        if self._k2s2_native(x) and x.requires_grad and torch.is_grad_enabled():
            return PW.SkipConvK2S2Fn.apply(x, self.weight, self.bias)
        return x, self(x)

    def forward(self, x):
        ok = (x.is_cuda and x.numel() and x.dtype in (torch.float32, torch.bfloat16) and self.weight.dtype == torch.float32 and x.dim() == 5 and self.groups == 1
              and _all(self.dilation, 1) and self.padding_mode == "zeros")
        if ok:
            C = self.in_channels
            if C % 2 and _all(self.kernel_size, 3) and _all(self.stride, 1) and _all(self.padding, 1) \
                    and x.shape[-1] % 4 == 0:
                # odd C_in (1- or 3-modality stems: ISLES / BraTS variants): the kernels consume channel
                # pairs, so run them on one extra all-zero channel (input 1/C larger, weights padded to match)
                xp = torch.nn.functional.pad(x, (0, 0, 0, 0, 0, 0, 0, 1))
                wp = torch.nn.functional.pad(self.weight, (0, 0, 0, 0, 0, 0, 0, 1))
                return PW.ConvK3Fn.apply(xp, wp, self.bias, None)
            if self._k2s2_native(x):
                return PW.ConvK2S2Fn.apply(x, self.weight, self.bias)
            if _all(self.kernel_size, 3) and _all(self.stride, 1) and _all(self.padding, 1) \
                    and x.shape[-1] % 4 == 0 and C % 2 == 0:
                return PW.ConvK3Fn.apply(x, self.weight, self.bias, None)
            if _all(self.kernel_size, 1) and _all(self.stride, 1) and _all(self.padding, 0) and C % 2 == 0 \
                    and (x.shape[2] * x.shape[3] * x.shape[4]) % 4 == 0:
                return PW.LinearFn.apply(x, self.weight.reshape(self.out_channels, C, 1), self.bias)
            composed.warn_once(f"conv3d{self.kernel_size}{self.stride}{tuple(x.shape[2:])}",
                               f"Conv3d kernel={self.kernel_size} stride={self.stride} on {tuple(x.shape)} is "
                               "outside the native kernel set; using ATen on device")
        return super().forward(x)


class ConvTranspose3d(nn.ConvTranspose3d):
    def native_ok(self, x, output_size=None):
        return bool(x.is_cuda and x.numel() and x.dtype in (torch.float32, torch.bfloat16) and self.weight.dtype == torch.float32 and x.dim() == 5 and self.groups == 1 and output_size is None
                    and _all(self.dilation, 1) and _all(self.kernel_size, 2) and _all(self.stride, 2)
                    and _all(self.padding, 0) and _all(self.output_padding, 0) and self.in_channels % 2 == 0
                    and (x.shape[2] * x.shape[3] * x.shape[4]) % 4 == 0)

    def forward(self, x, output_size=None):
        ok = self.native_ok(x, output_size)
        if ok:
            return PW.TConvK2S2Fn.apply(x, self.weight, self.bias)
        if x.is_cuda and x.numel():
            composed.warn_once(f"tconv3d{self.kernel_size}{self.stride}{tuple(x.shape[2:])}",
                               f"ConvTranspose3d kernel={self.kernel_size} stride={self.stride} on "
                               f"{tuple(x.shape)} is outside the native kernel set; using ATen on device")
        return super().forward(x, output_size)
