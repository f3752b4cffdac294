"""FactMixer / FactorizerBlock / FactorizerStage — host mirrors of the reference's
factorizer.py:9-122 (same constructor arguments, sub-module names and construction order, so
seeds and state_dicts line up)."""
from __future__ import annotations

from torch import nn

from .layers import MLP, LayerNorm, Linear
from .matricize import Matricize
from .nmf import NMF
from .utils import partialize


class FactMixer(nn.Module):
    """in_proj → reshape (matricize) → act → factorize → reshape⁻¹ → out_proj → dropout
    (factorizer.py:34-57).  ``reshape`` and ``factorize`` are duck-typed slots
    (SURVEY.md §8b): reshape(input_size) with .output_size/.forward/.inverse_forward,
    factorize(size, **kwargs) with .forward."""

    def __init__(self, in_channels, out_channels, spatial_size,
                 reshape=(Matricize, {"num_heads": 1, "grid_size": 1}), act=nn.ReLU, factorize=NMF,
                 dropout=0.0, **kwargs):
        super().__init__()
        self.in_proj = Linear(in_channels, out_channels, bias=False)
        self.reshape = partialize(reshape)((None, out_channels, *spatial_size))
        self.act = partialize(act)()
        self.reshaped_size = self.reshape.output_size[2:]
        self.factorize = partialize(factorize)(self.reshaped_size, **kwargs)
        # third positional argument of Linear is `bias` (factorizer.py:31): out_proj has a bias
        self.out_proj = Linear(in_channels, out_channels, True)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        out = self.in_proj(x)
        out = self.reshape(out)
        out = self.act(out)
        out = self.factorize(out)
        out = self.reshape.inverse_forward(out)
        out = self.out_proj(out)
        if self.dropout.p > 0:
            out = self.dropout(out)
        return out


class FactorizerBlock(nn.Module):
    """x += fact(norm1(x)); x += mlp(norm2(x))  (factorizer.py:60-77)."""

    def __init__(self, channels, spatial_size, norm=LayerNorm, dropout=0.0, mlp_ratio=2, **kwargs):
        super().__init__()
        self.norm1 = partialize(norm)(channels)
        self.fact = FactMixer(channels, channels, spatial_size, dropout=dropout, **kwargs)
        self.norm2 = partialize(norm)(channels)
        self.mlp = MLP(channels, ratio=mlp_ratio, dropout=dropout)

    def forward(self, x):
        x = x + self.fact(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x


class FactorizerStage(nn.Module):
    """[adapter if in≠out] → [pos_embed (+pos_drop)] → depth × FactorizerBlock
    (factorizer.py:80-122).  Note: `dropout` is consumed here and reaches only `pos_drop`."""

    def __init__(self, in_channels, out_channels, spatial_size, depth=1,
                 adapter=(Linear, {"bias": False}), pos_embed=nn.Identity, dropout=0.0, **subblocks):
        super().__init__()
        if in_channels != out_channels:
            self.adapter = partialize(adapter)(in_channels, out_channels)
        self.pos_embed = partialize(pos_embed)(out_channels, spatial_size)
        if len(list(self.pos_embed.parameters())) > 0:
            self.pos_drop = nn.Dropout(dropout)
        self.blocks = nn.ModuleList()
        for _ in range(depth):
            self.blocks.append(FactorizerBlock(out_channels, spatial_size, **subblocks))

    def forward(self, x):
        out = self.adapter(x) if hasattr(self, "adapter") else x
        if not isinstance(self.pos_embed, nn.Identity):
            out = self.pos_embed(out)
        if hasattr(self, "pos_drop") and self.pos_drop.p > 0:
            out = self.pos_drop(out)
        for blk in self.blocks:
            out = blk(out)
        return out
