"""Channels-first layers of the Factorizer block: ``Linear`` (1×1 "conv"), ``LayerNorm`` over
the channel dim, ``MLP`` and the learnable ``PositionalEmbedding`` — host mirrors of the
reference's layers/linear.py:7-58, layers/norm.py:5-34, layers/mlp.py:10-63 and
layers/pos_embed.py:72-93 (same constructor arguments, same parameter names/shapes so
reference checkpoints load: 1×1 weights are Conv1d-shaped (out, in, 1))."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.modules.utils import _pair

from . import pointwise as PW


class Linear(nn.Module):
    """y[b, :, v] = W x[b, :, v] + bias for every voxel v of a channels-first tensor."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True, device=None, dtype=None):
        super().__init__()
        self.flatten = nn.Flatten(start_dim=2)
        # nn.Conv1d is only the parameter container (identical init / state_dict keys)
        self.linear = nn.Conv1d(in_channels, out_channels, kernel_size=1, bias=bias, device=device,
                                dtype=dtype)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return PW.linear_cf(x, self.linear.weight, self.linear.bias)


class LayerNorm(nn.Module):
    """nn.LayerNorm(C) applied across the channel dim of (B, C, S1, ..., Sp)."""

    def __init__(self, dim: int, **kwargs):
        super().__init__()
        self.norm = nn.LayerNorm(dim, **kwargs)

    def forward(self, x):
        return PW.layernorm_cf(x, self.norm.weight, self.norm.bias, self.norm.eps)


class MLP(nn.Module):
    """Linear → GELU(erf) → Dropout → Linear → Dropout (mlp.py:40-63)."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None,
                 hidden_channels: Optional[int] = None, ratio: float = 3.0, dropout=0.0, **kwargs):
        super().__init__()
        out_channels = out_channels or in_channels
        hidden_channels = hidden_channels or int(ratio * in_channels)
        dropout = _pair(dropout)
        self.block = nn.Sequential(
            Linear(in_channels, hidden_channels, **kwargs),
            nn.GELU(),
            nn.Dropout(dropout[0]),
            Linear(hidden_channels, out_channels, **kwargs),
            nn.Dropout(dropout[1]),
        )

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        fc1, act, drop1, fc2, drop2 = self.block
        fusable = (isinstance(act, nn.GELU) and act.approximate == "none"
                   and not (self.training and (drop1.p > 0 or drop2.p > 0)))
        if fusable:
            return PW.mlp_cf(x, fc1.linear.weight, fc1.linear.bias, fc2.linear.weight, fc2.linear.bias)
        return self.block(x)


class PositionalEmbedding(nn.Module):
    """Learnable additive embedding of shape (1, C, *spatial), N(0,1) init."""

    def __init__(self, channels: int, spatial_size) -> None:
        super().__init__()
        self.pos = nn.Parameter(torch.empty(1, channels, *spatial_size))
        nn.init.normal_(self.pos, std=1.0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dtype != self.pos.dtype and x.dtype == torch.bfloat16:
            return x + self.pos.to(x.dtype)   # bf16 activations stay bf16 (no promotion by the fp32 parameter)
        return x + self.pos


PosEmbed = PositionalEmbedding
