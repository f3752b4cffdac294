"""Deconver — the reference's second model family: blind non-negative deconvolution by
multiplicative updates as the token mixer of a U-shaped network.

Mirrors, name for name and state_dict key for key:
  ``Deconv``            factorizer/factorization/deconvolution.py:88-260 (+ its ``Initializer`` :60-85)
  ``DeconvMixer``, ``DeconverBlock``, ``DeconverStage``, ``Stem``, ``Deconver``      factorizer/deconver.py:9-230

Model: per sample and channel group g,  x_g ≈ H_g s_g  with  (H s)[c] = Σ_k s[k] ⋆ h[c, k]  (cross-correlation,
"same" padding), x_g (C/G channels), sources s_g (K channels) ≥ 0, filters h_g (C/G, K, *kernel) ≥ 0.
One iteration (deconvolution.py:136-156):

    s ← s ∘ (Hᵀx + ε) / (HᵀH s + ε)                     update_source
    h ← h ∘ (corr(s, x) + ε) / (corr(s, H s) + ε)       update_filter   (lags within the kernel support)

where Hᵀ is the exact adjoint of H (correlation with the channel-transposed, spatially flipped filters).

Evaluation here: every operator is a grouped ``F.conv{1,2,3}d``.  While the filters are shared by the batch
(``update_filter=False``, the default: h = relu(h0) is a parameter) the G groups are the convolution's groups;
once they depend on the sample the batch is folded into the groups.  Hᵀx + ε does not depend on s, so it is
computed once per call, not once per iteration.  The 1×1 projections, LayerNorm and MLP around the mixer run
the native GEMM-family kernels on device (factorizer_amd/pointwise.py); H, Hᵀ and their input gradients run
the native grouped-correlation kernel (csrc/deconv.hip: fp32, ≤ 16 channels per group, cubic or square 3/5/7
kernels; iterations without gradient fuse the update and the division into its epilogue); the filter gradient and
`update_filter`'s lag correlations — the same reduction over all voxels — run `fz_gcorr_wgrad` (deterministic
two-stage sums).  Shapes outside that set use framework convolutions on device, announced once by a RuntimeWarning.
"""
from __future__ import annotations

import math
from contextlib import nullcontext
from typing import Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor, nn

from . import composed, convs
from . import functional as Fn
from .layers import MLP, LayerNorm, Linear
from .nmf import relative_error
from .ushape import UNet
from .utils import partialize

_CONV = {1: F.conv1d, 2: F.conv2d, 3: F.conv3d}


def _adjoint_filters(h: Tensor) -> Tensor:
    """(Bw, G, C, K, *k) → (Bw, G, K, C, *k) with every spatial axis reversed: correlation with it is Hᵀ."""
    return torch.flip(h.transpose(2, 3), dims=tuple(range(4, h.ndim)))


def _gcorr(inp: Tensor, w: Tensor, padding) -> Tensor:
    """Grouped "same" cross-correlation.  inp (B, G·Cin, *S); w (Bw, G, Cout, Cin, *k) with Bw ∈ {1, B}
    → (B, G·Cout, *S):  out[b, g·Cout + o] = Σ_i inp[b, g·Cin + i] ⋆ w[b or 0, g, o, i]."""
    B = inp.shape[0]
    Bw, G, Cout, Cin = w.shape[:4]
    nd = inp.ndim - 2
    if Bw == 1:
        return _CONV[nd](inp, w.reshape(G * Cout, Cin, *w.shape[4:]), padding=padding, groups=G)
    out = _CONV[nd](inp.reshape(1, B * G * Cin, *inp.shape[2:]), w.reshape(B * G * Cout, Cin, *w.shape[4:]),
                    padding=padding, groups=B * G)
    return out.reshape(B, G * Cout, *out.shape[2:])


def _lag_corr(s: Tensor, x: Tensor, G: int, padding) -> Tensor:
    """corr(s, x)[b, g, c, k, τ] = Σ_v s[b, g·K + k, v + τ − p] · x[b, g·C + c, v]  for the lags τ of the kernel
    support (deconvolution.py:43-50 `sconv` + the transpose of `update_h`) → (B, G, C, K, *kernel)."""
    B = s.shape[0]
    K, C = s.shape[1] // G, x.shape[1] // G
    nd = s.ndim - 2
    sp = s.shape[2:]
    # sources as a batch of K images over B·G channels, x as B·G·C single-channel filters of full spatial size
    inp = s.reshape(B * G, K, *sp).transpose(0, 1)
    w = x.reshape(B * G * C, 1, *sp)
    out = _CONV[nd](inp, w, padding=padding, groups=B * G)           # (K, B·G·C, *kernel)
    out = out.reshape(K, B, G, C, *out.shape[2:])
    return out.permute(1, 2, 3, 0, *range(4, out.ndim))


class DeconvInitializer(nn.Module):
    """s0 = relu(Linear(x)), h0 = relu(parameter) broadcast over the batch (deconvolution.py:60-85)."""

    def __init__(self, channels: int, source_channels: int, kernel_size: Sequence[int], groups: int) -> None:
        super().__init__()
        groups = channels if groups is None else groups
        if channels % groups:
            raise ValueError("`channels` must be divisible by groups")
        h0 = torch.empty(channels, source_channels, *kernel_size)
        nn.init.kaiming_uniform_(h0, a=math.sqrt(5))
        self.h0 = nn.Parameter(h0)
        self.linear = Linear(channels, groups * source_channels)

    def forward(self, x: Tensor):
        s = F.relu(self.linear(x))
        h = F.relu(self.h0).expand(x.shape[0], *self.h0.shape)
        return s, h


class Deconv(nn.Module):
    """Blind deconvolution layer: returns the sources after `num_iters` multiplicative updates."""

    def __init__(self, channels: int, kernel_size: Sequence[int] = (3, 3, 3), source_channels: Optional[int] = None,
                 ratio: float = 4, groups: int = 8, update_source=True, update_filter=False, eps: float = 1e-16,
                 num_iters: int = 2, num_grad_iters: Optional[int] = None, verbose: bool = False, **kwargs) -> None:
        super().__init__()
        self.channels = channels
        self.groups = channels if groups == -1 else groups
        if self.channels % self.groups:
            raise ValueError("`channels` must be divisible by groups")
        self.source_channels = round(channels * ratio / self.groups if source_channels is None else source_channels)
        if self.source_channels < 1:
            raise ValueError(f"ratio {ratio} with {self.groups} groups of {channels} channels leaves no source channel")
        self.kernel_size = tuple(kernel_size)
        self.init = DeconvInitializer(self.channels, self.source_channels, self.kernel_size, self.groups)
        self.update_source = update_source
        self.update_filter = update_filter
        self.num_iters = num_iters
        self.num_grad_iters = num_iters if num_grad_iters is None else num_grad_iters
        self.eps = eps
        self.verbose = verbose
        self.padding = tuple(k // 2 for k in self.kernel_size)
        self.conv_kwargs = dict(kwargs)
        if kwargs:
            raise TypeError(f"unexpected arguments {sorted(kwargs)}")

    # ---- grouped views ------------------------------------------------------------------------------
    def _filters(self, h: Tensor) -> Tensor:
        """(B or 1, C, K, *k) → (Bw, G, C/G, K, *k)"""
        return h.reshape(h.shape[0], self.groups, self.channels // self.groups, *h.shape[2:])

    def split_channels(self, t: Tensor) -> Tensor:
        """'b (g c) ... -> (b g) c ...' (deconvolution.py:125)"""
        return t.reshape(t.shape[0] * self.groups, t.shape[1] // self.groups, *t.shape[2:])

    def merge_channels(self, t: Tensor) -> Tensor:
        return t.reshape(t.shape[0] // self.groups, t.shape[1] * self.groups, *t.shape[2:])

    def normalize_h(self, h: Tensor) -> Tensor:
        return (h + self.eps) / (h.sum([d for d in range(h.ndim) if d not in (0, 2)], keepdim=True) + self.eps)

    # ---- operators ------------------------------------------------------------------------------------
    def _corr(self, inp: Tensor, wg: Tensor, add_eps: float = 0.0) -> Tensor:
        """grouped correlation (+ eps): the native gfx950 kernel (csrc/deconv.hip) on device where it applies"""
        if Fn.gcorr_supported(inp, wg):
            return Fn.gcorr(inp, wg, add_eps)
        if inp.is_cuda and inp.numel():
            composed.warn_once(f"deconv{tuple(inp.shape[1:])}{tuple(wg.shape)}",
                               f"Deconv: grouped correlation {tuple(wg.shape)} on {tuple(inp.shape)} ({inp.dtype}) is outside "
                               "the native kernel set (fp32, odd kernel extents <= 7); "
                               "using framework convolutions on device")
        out = _gcorr(inp, wg, self.padding)
        return out + add_eps if add_eps else out

    def _lags(self, s: Tensor, x: Tensor) -> Tensor:
        """corr(s, x) over the lags of the kernel support → (B, G, C/G, K, *kernel): fz_gcorr_wgrad on device"""
        if Fn.lag_corr_supported(s, x, self.groups, self.kernel_size):
            return Fn.lag_corr(s, x, self.groups, self.kernel_size)
        if x.is_cuda and x.numel():
            composed.warn_once("deconv_update_h", "Deconv(update_filter=True): lag correlations outside the native kernel "
                               "set (fp32, odd kernel extents <= 7); using framework "
                               "convolutions on device")
        return _lag_corr(s, x, self.groups, self.padding)

    def _H(self, s: Tensor, hg: Tensor) -> Tensor:
        return self._corr(s, hg)

    def _Ht(self, r: Tensor, hg: Tensor, add_eps: float = 0.0) -> Tensor:
        return self._corr(r, _adjoint_filters(hg), add_eps)

    def context(self, it: int):
        return torch.no_grad() if it < self.num_iters - self.num_grad_iters + 1 else nullcontext()

    def _iterate(self, x: Tensor, s: Tensor, h: Tensor):
        """x (B, C, *S), s (B, G·K, *S), h (Bw, C, K, *k) → (s, h) after num_iters updates."""
        shared_num = {}   # Hᵀx + ε is the same in every iteration while h is fixed: one per autograd mode
        for it in range(1, self.num_iters + 1):
            with self.context(it):
                hg = self._filters(h)
                if self.verbose:
                    print(f"iter {it}: loss = {relative_error(x, self._H(s, hg))}")
                if self.update_source:
                    key = torch.is_grad_enabled()
                    if self.update_filter or key not in shared_num:
                        num = self._Ht(x, hg, self.eps)
                        shared_num[key] = num
                    else:
                        num = shared_num[key]
                    r = self._H(s, hg)
                    hT = _adjoint_filters(hg)
                    no_grad_needed = not (torch.is_grad_enabled() and (s.requires_grad or num.requires_grad
                                                                       or r.requires_grad or hT.requires_grad))
                    if no_grad_needed and Fn.gcorr_supported(r, hT):
                        s = Fn.gcorr_mu_update(s, num, r, hT, self.eps)      # update + division fused in the kernel
                    else:
                        s = s * num / self._corr(r, hT, self.eps)
                if self.update_filter:
                    num_h = self._lags(s, x) + self.eps
                    den_h = self._lags(s, self._H(s, hg)) + self.eps
                    ratio = (num_h / den_h).reshape(x.shape[0], self.channels, self.source_channels, *self.kernel_size)
                    h = h * ratio
        return s, h

    def _start(self, x: Tensor):
        s, h = self.init(x)
        if not self.update_filter:
            h = h[:1]          # shared by the batch
        return s, h

    # ---- API (deconvolution.py:176-260) ---------------------------------------------------------------
    def fit(self, x: Tensor):
        s, h = self._start(x)
        s, h = self._iterate(x, s, h)
        return s, h.expand(x.shape[0], *h.shape[1:])

    def reconstruct(self, s: Tensor, h: Tensor) -> Tensor:
        return self._H(s, self._filters(h))

    def loss(self, x: Tensor, s: Tensor, h: Tensor) -> Tensor:
        """relative error per sample of the (already group-split, as in the reference) operands"""
        Bg = x.shape[0]
        hg = h.reshape(Bg, 1, *h.shape[1:])
        return relative_error(x, _gcorr(s, hg, self.padding))

    def forward(self, x: Tensor) -> Tensor:
        s, h = self._start(x)
        s, _ = self._iterate(x, s, h)
        return s


class DeconvMixer(nn.Module):
    """in_proj → act → Deconv → out_proj → dropout (deconver.py:9-47)."""

    def __init__(self, in_channels, out_channels, act=nn.ReLU, dropout=0.0, **kwargs):
        super().__init__()
        self.in_proj = Linear(in_channels, out_channels, bias=False)
        self.deconv = Deconv(out_channels, **kwargs)
        self.act = partialize(act)()
        self.out_proj = Linear(self.deconv.groups * self.deconv.source_channels, out_channels)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        return self.dropout(self.out_proj(self.deconv(self.act(self.in_proj(x)))))


class DeconverBlock(nn.Module):
    """x += dcm(norm1(x)); x += mlp(norm2(x)) (deconver.py:50-66)."""

    def __init__(self, channels, norm=LayerNorm, dropout=0.0, mlp_ratio=4, **kwargs):
        super().__init__()
        self.norm1 = partialize(norm)(channels)
        self.dcm = DeconvMixer(channels, channels, **kwargs)
        self.norm2 = partialize(norm)(channels)
        self.mlp = MLP(channels, ratio=mlp_ratio, dropout=dropout)

    def forward(self, x):
        x = x + self.dcm(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class DeconverStage(nn.Module):
    """[adapter if in ≠ out] → depth × DeconverBlock (deconver.py:91-126)."""

    def __init__(self, in_channels, out_channels, spatial_size=None, depth=1, adapter=(Linear, {"bias": False}), **kwargs):
        super().__init__()
        if in_channels != out_channels:
            self.adapter = partialize(adapter)(in_channels, out_channels)
        self.blocks = nn.ModuleList(DeconverBlock(out_channels, **kwargs) for _ in range(depth))

    def forward(self, x):
        out = self.adapter(x) if hasattr(self, "adapter") else x
        for blk in self.blocks:
            out = blk(out)
        return out


class Stem(nn.Sequential):
    """patchifying convolution + norm (deconver.py:129-138)."""

    def __init__(self, in_channels, out_channels, patch_size=(4, 4), norm=LayerNorm):
        nd = len(patch_size)
        super().__init__(getattr(nn, f"Conv{nd}d")(in_channels, out_channels, patch_size, stride=patch_size),
                         partialize(norm)(out_channels))


class Deconver(UNet):
    """U-shaped segmentation network whose every stage is a DeconverStage (deconver.py:141-190)."""

    def __init__(self, in_channels, out_channels, spatial_dims=3, encoder_depth=(1, 1, 1, 1, 1),
                 encoder_width=(32, 64, 128, 256, 512), strides=(1, 2, 2, 2, 2), decoder_depth=(1, 1, 1, 1),
                 stem=None, downsample=None, upsample=None, head=None, num_deep_supr=False, **kwargs):
        stages = (len(encoder_depth) + len(decoder_depth)) * [DeconverStage]
        if stem is None:
            conv = convs.Conv3d if spatial_dims == 3 else getattr(nn, f"Conv{spatial_dims}d")
            stem = (conv, {"kernel_size": 3, "padding": 1, "bias": False})
        super().__init__(in_channels, out_channels, spatial_dims=spatial_dims, encoder_depth=encoder_depth,
                         encoder_width=encoder_width, strides=strides, decoder_depth=decoder_depth, stem=stem,
                         downsample=downsample, block=stages, upsample=upsample, head=head,
                         num_deep_supr=num_deep_supr, **kwargs)
