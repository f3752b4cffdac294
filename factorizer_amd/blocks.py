"""FactMixer / FactorizerBlock / FactorizerStage — host mirrors of the reference's
factorizer.py:9-122 (same constructor arguments, sub-module names and construction order, so
seeds and state_dicts line up)."""
from __future__ import annotations

import os

import torch
from torch import nn

from . import functional as Fn
from . import pointwise as PW
from .layers import MLP, LayerNorm, Linear
from .matricize import Matricize, SWMatricize
from .nmf import NMF, MatrixFactorization, RandomInit
from .utils import partialize


def _env_int(name, default):
    """diagnostic knob, read ONCE at import and validated (a malformed value must not surface in the middle of training)"""
    v = os.environ.get(name)
    if v is None:
        return default
    try:
        return int(v, 0)
    except ValueError:
        raise ValueError(f"{name}={v!r}: expected an integer") from None


_UP_FUSED = _env_int("FZ_UP_FUSED", 1) != 0            # 0 = decoder levels as separate autograd nodes
_UP_FUSED_MIN = _env_int("FZ_UP_FUSED_MIN", 1 << 18)   # voxels x batch from which the one-node decoder level pays


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

    def _fusable(self, x) -> bool:
        """Standard Swin block on a device tensor (fp32, or bf16 activations with fp32 parameters): LayerNorm / ReLU / exact GELU, no live
        dropout — then LayerNorm, bias, ReLU, GELU and both residual adds are fused into the GEMM
        kernels (csrc/gemm.hip) instead of running as separate full-tensor passes."""
        f, m = self.fact, self.mlp
        if not (x.is_cuda and x.numel() and x.dtype in (torch.float32, torch.bfloat16) and PW._vox(x) % 4 == 0
                and x.shape[1] % 2 == 0):
            return False
        if any(p.dtype != torch.float32 for p in self.parameters()):  # bf16 activations keep fp32 parameters
            return False
        if not (isinstance(self.norm1, LayerNorm) and isinstance(self.norm2, LayerNorm)
                and self.norm1.norm.elementwise_affine and self.norm2.norm.elementwise_affine
                and self.norm1.norm.bias is not None and self.norm2.norm.bias is not None):
            return False
        if not (isinstance(f.act, nn.ReLU) and isinstance(f.in_proj, Linear) and isinstance(f.out_proj, Linear)):
            return False
        blk = m.block
        if not (len(blk) == 5 and isinstance(blk[1], nn.GELU) and blk[1].approximate == "none"
                and blk[0].linear.weight.shape[0] % 2 == 0):
            return False
        live = self.training and (f.dropout.p > 0 or blk[2].p > 0 or blk[4].p > 0)
        return not live

    def _modular_native_cfg(self, x) -> bool:
        """matricize / NMF / inverse all covered by the native modular kernels (any patch size)."""
        f = self.fact
        mf = f.factorize
        if not (isinstance(f.reshape, SWMatricize) and isinstance(mf, MatrixFactorization)):
            return False
        t_like = x.new_empty((1, *f.reshape.output_size[1:]))
        # (wave-resident NMF family only: the block's hand-chained backward calls those kernels directly)
        return mf._native_solver(t_like) is not None and not mf._wide and len(f.reshape.geometry.spatial) <= 3

    def _core_cfg(self):
        """(grad steps, solver id) if matricize→NMF→inverse can run as the fused channels-first
        kernels (SWMatricize head_dim 8, patch 8³ — csrc/nmf_cf.hip — or any patch of ≤ 256 voxels — csrc/nmf_pcf.hip;
        native MU/HALS, rank ≤ 2), else None."""
        f = self.fact
        mf = f.factorize
        if not (isinstance(f.reshape, SWMatricize) and isinstance(mf, MatrixFactorization)):
            return None
        sid = getattr(mf.solver, "native_id", None)
        if sid is None or not isinstance(mf.init, RandomInit) or mf.solver.factor != (0, 1) or mf.verbose:
            return None
        G = min(max(mf.num_grad_steps, 0), mf.num_iters)
        geo = f.reshape.geometry
        if tuple(mf.size) != (8, geo.P) or not Fn.nmf_core_supported(geo, mf.rank, mf.num_iters, G):
            return None
        return G, sid

    def prologue_params(self, like):
        """what a producer of this block's input needs to apply the block's first layer itself (pointwise.BlockPrologue), or
        None when this block would not take the one-node native path with the fused core for a tensor like `like`"""
        f, blk = self.fact, self.mlp.block
        n1 = self.norm1.norm
        if not (like.is_cuda and like.shape[1] == 32 and self._fusable(like) and self._core_cfg() is not None
                and f.in_proj.linear.bias is None and blk[0].linear.bias is not None and blk[3].linear.bias is not None
                and f.out_proj.linear.bias is not None and f.in_proj.linear.weight.dtype == torch.float32
                and n1.weight is not None and n1.bias is not None
                and not (self.norm1._forward_hooks or f.in_proj._forward_hooks or f._forward_hooks)):
            return None
        return n1.weight, n1.bias, n1.eps, f.in_proj.linear.weight

    def forward(self, x):
        if self._fusable(x):
            f, blk = self.fact, self.mlp.block
            n1, n2 = self.norm1.norm, self.norm2.norm
            core = self._core_cfg()
            mf = f.factorize
            nat = core is not None or self._modular_native_cfg(x)
            if nat and f.in_proj.linear.bias is None and blk[0].linear.bias is not None \
                    and blk[3].linear.bias is not None and f.out_proj.linear.bias is not None:
                if core is not None:
                    G, sid = core
                else:
                    G, sid = min(max(mf.num_grad_steps, 0), mf.num_iters), mf.solver.native_id
                cfg = dict(geo=f.reshape.geometry, T=mf.num_iters, G=G, solver=sid, nmf_eps=mf.solver.eps,
                           eps1=n1.eps, eps2=n2.eps, core=core is not None)
                args = (x, n1.weight, n1.bias, f.in_proj.linear.weight, mf.init.u0, mf.init.v0,
                        f.out_proj.linear.weight, f.out_proj.linear.bias, n2.weight, n2.bias,
                        blk[0].linear.weight, blk[0].linear.bias, blk[3].linear.weight, blk[3].linear.bias, cfg)
                pre = PW.BlockPrologue.take(x) if core is not None else None   # t, statistics already formed by x's producer
                t_pre, st_pre = pre if pre is not None else (None, None)
                slot = PW.head_fusion_slot()
                if slot is not None and slot.block is self and core is not None:
                    # this block's output feeds ONLY the network's head (ushape.UNet.forward): the head runs inside the
                    # block's last launch, HeadOfBlockFn carries its gradient
                    hw, hb = slot.head_params
                    C, Hd = x.shape[1], blk[0].linear.weight.shape[0]
                    if PW.block_head_fusable(C, Hd, x[0, 0].numel(), hw, x):
                        y, raw = PW.FactorizerBlockFn.apply(*args, hw, hb, t_pre, st_pre)
                        slot.logits = PW.HeadOfBlockFn.apply(y, hw, hb, raw)
                        return y
                return PW.FactorizerBlockFn.apply(*args, None, None, t_pre, st_pre)
            # per-layer fused path (composed NMF, unusual bias layout, ...)
            t = PW.ln_linear(x, n1.weight, n1.bias, n1.eps, f.in_proj.linear.weight, f.in_proj.linear.bias, "relu")
            a = f.reshape.inverse_forward(f.factorize(f.reshape(t)))  # ReLU already applied (commutes)
            x = PW.act_linear_res(a, f.out_proj.linear.weight, f.out_proj.linear.bias, x, "none")
            z = PW.ln_linear(x, n2.weight, n2.bias, n2.eps, blk[0].linear.weight, blk[0].linear.bias, "none")
            return PW.act_linear_res(z, blk[3].linear.weight, blk[3].linear.bias, x, "gelu")
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

    def forward_pair(self, skip, up):
        """forward(torch.cat([skip, up], 1)) without materialising the concatenation
        (unet.py:128): the adapter GEMM reads its two channel groups from two pointers."""
        if hasattr(self, "adapter") and isinstance(self.adapter, Linear):
            out = PW.cat_linear(skip, up, self.adapter.linear.weight, self.adapter.linear.bias)
            return self._after_adapter(out)
        return self.forward(torch.cat([skip, up], dim=1))

    def forward_up_pair(self, skip, deep, upsample):
        """forward(torch.cat([skip, upsample(deep)], 1)) with the transposed convolution and the adapter as ONE autograd
        node (pointwise.UpCatLinearFn: gradients from the composed weights, the up-sampled tensor is not kept), or None
        when this stage / up-sampler / tensor is outside that node's scope."""
        from . import convs as _convs
        ok = (hasattr(self, "adapter") and isinstance(self.adapter, Linear) and type(upsample) is _convs.ConvTranspose3d
              and upsample.native_ok(deep) and skip.is_cuda and skip.dtype == deep.dtype and skip.dim() == 5
              and skip.shape[1] % 16 == 0 and upsample.out_channels % 16 == 0 and deep.shape[1] % 16 == 0
              and self.adapter.linear.weight.dtype == torch.float32
              and tuple(skip.shape[2:]) == tuple(2 * d for d in deep.shape[2:])
              and self.adapter.linear.weight.shape[1] == skip.shape[1] + upsample.out_channels
              # (the node saves 4 full-resolution tensor passes of the level and costs ~10 tiny launches for the weight
              # compositions: it pays from 2^18 voxels x batch on — the two shallow levels of the README model)
              and _UP_FUSED and skip.shape[0] * skip[0, 0].numel() >= _UP_FUSED_MIN
              # forward hooks on the two modules the node replaces must keep firing: leave such stages to the module path
              and not (upsample._forward_hooks or upsample._forward_pre_hooks
                       or self.adapter._forward_hooks or self.adapter._forward_pre_hooks
                       or self.adapter.linear._forward_hooks or self.adapter.linear._forward_pre_hooks))
        if not ok:
            return None
        pro = self._first_block_prologue(skip) if PW.upcat_prologue_ok(skip, deep, upsample.weight, self.adapter.linear.weight) else None
        if pro is not None:
            # the first block's LayerNorm 1 + in_proj + ReLU in the launch that produces its input (csrc/upcat.hip PRO)
            out, t, st = PW.up_cat_linear(skip, deep, upsample.weight, upsample.bias, self.adapter.linear.weight, self.adapter.linear.bias, pro)
            with PW.BlockPrologue(out, t, st):
                return self._after_adapter(out)
        out = PW.up_cat_linear(skip, deep, upsample.weight, upsample.bias, self.adapter.linear.weight, self.adapter.linear.bias)
        return self._after_adapter(out)

    def _first_block_prologue(self, like):
        """(ln1 weight, ln1 bias, eps, in_proj weight) of the first block when the stage hands its adapter output straight to a
        FactorizerBlock of 32 channels that runs as the one-node native block with the fused core (no position embedding or
        dropout in between, no hooks that expect the separate launch), else None."""
        if not isinstance(self.pos_embed, nn.Identity) or (hasattr(self, "pos_drop") and self.pos_drop.p > 0) or len(self.blocks) == 0:
            return None
        blk = self.blocks[0]
        if type(blk).__name__ != "FactorizerBlock" or blk._forward_hooks or blk._forward_pre_hooks:
            return None
        return blk.prologue_params(like)

    def forward(self, x):
        out = self.adapter(x) if hasattr(self, "adapter") else x
        return self._after_adapter(out)

    def _after_adapter(self, out):
        if not isinstance(self.pos_embed, nn.Identity):
            out = self.pos_embed(out)
        if hasattr(self, "pos_drop") and self.pos_drop.p > 0:
            out = self.pos_drop(out)
        for blk in self.blocks:
            out = blk(out)
        return out
