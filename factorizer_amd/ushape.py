"""U-shaped encoder/decoder shell and the ``Factorizer`` model.

Host mirror of the reference's generic ``UNet`` family (factorizer/unet.py:11-276) and of
``Factorizer(UNet)`` (factorizer/factorizer.py:125-171): same constructor arguments, sub-module
names (``stem``, ``encoder.blocks.i.{downsample,block}``, ``decoder.blocks.j.{upsample,block}``,
``head`` / ``heads``) and parameter-creation order, so seeds and checkpoints line up.  Unlike the
reference, which threads ``spatial_size`` through a mutated kwargs dict, the per-stage geometry is
resolved once, up front (`_plan`), and handed to each stage explicitly.
"""
from __future__ import annotations

import math
from collections.abc import Sequence

import torch
from torch import nn

from . import convs
from .blocks import FactorizerStage
from .layers import PositionalEmbedding
from .utils import as_tuple, partialize


def _scale(size, factor, down: bool):
    """Spatial size after one stage: ``d // s`` going down, ``d * s`` going up; non-sequences
    (``spatial_size=None``) pass through untouched."""
    if not isinstance(size, Sequence):
        return size
    return tuple(d // factor for d in size) if down else tuple(d * factor for d in size)


def _plan(spatial_size, strides, n_dec):
    """Per-stage spatial sizes: encoder stage i sees the input divided by strides[0..i]; decoder
    stage j multiplies back by the reversed strides (unet.py:80-83, 149-152)."""
    enc, cur = [], spatial_size
    for s in strides:
        cur = _scale(cur, s, True)
        enc.append(cur)
    dec = []
    for s in list(strides)[::-1][:n_dec]:
        cur = _scale(cur, s, False)
        dec.append(cur)
    return enc, dec


class Same:
    """``Same(spec)[i]`` is ``spec`` for every i — one block spec shared by all stages."""

    def __init__(self, block):
        self.block = block

    def __getitem__(self, index):
        return self.block


class UNetStage(nn.Module):
    """`depth` copies of a block; the first one changes the channel count."""

    def __init__(self, in_channels, out_channels, depth=1, block=None, **kwargs):
        super().__init__()
        if block is None:
            raise ValueError("UNetStage needs a `block` (the reference's CNN DoubleConv default is outside "
                             "this build's scope)")
        make = partialize(block)
        layers = [make(in_channels, out_channels, **kwargs)]
        layers += [make(out_channels, out_channels, **kwargs) for _ in range(depth - 1)]
        self.blocks = nn.Sequential(*layers)

    def forward(self, x):
        return self.blocks(x)


class UNetEncoderBlock(nn.Module):
    """Optional kernel-2 / stride-2 down-convolution, then the stage."""

    def __init__(self, in_channels, out_channels, depth=1, stride=2,
                 downsample=(convs.Conv3d, {"kernel_size": 2}), block=UNetStage, **kwargs):
        super().__init__()
        keep_resolution = math.prod(as_tuple(stride)) == 1
        down = nn.Identity if keep_resolution else partialize(downsample)
        # the reference always builds the down-conv with stride 2 (unet.py:53)
        self.downsample = down(in_channels, out_channels, stride=2)
        self.block = partialize(block)(out_channels, out_channels, depth=depth, **kwargs)

    def forward(self, x):
        return self.block(self.downsample(x))

    def forward_fork(self, x):
        """``(x_for_the_skip, self(x))``: lets a native down-convolution take over the addition of the
        two gradients of ``x`` (skip + down path), see convs.Conv3d.forward_fork."""
        fork = getattr(self.downsample, "forward_fork", None)
        if fork is None:
            return x, self(x)
        skip, y = fork(x)
        return skip, self.block(y)


class UNetEncoder(nn.Module):
    def __init__(self, in_channels, out_channels=(32, 64, 128, 256, 512), depth=(1, 1, 1, 1, 1),
                 strides=(1, 2, 2, 2, 2), downsample=None, block=None, spatial_size=None, **kwargs):
        super().__init__()
        sizes, _ = _plan(spatial_size, strides[:len(out_channels)], 0)
        widths = (in_channels, *out_channels)
        self.in_spatial_size = spatial_size
        self.out_spatial_size = sizes[-1] if sizes else spatial_size
        self.blocks = nn.ModuleList(
            UNetEncoderBlock(widths[i], widths[i + 1], depth[i], strides[i], downsample, block[i],
                             spatial_size=sizes[i], **kwargs)
            for i in range(len(out_channels)))

    def forward(self, x):
        feats = []
        for i, stage in enumerate(self.blocks):
            if i == 0:
                x = stage(x)
            else:
                feats[-1], x = stage.forward_fork(x)  # x is both a skip connection and this stage's input
            feats.append(x)
        return feats


class UNetDecoderBlock(nn.Module):
    """Transposed kernel-2 up-convolution, skip connection, stage (unet.py:107-130).  When the stage
    can read the two halves of the concatenation through separate pointers (FactorizerStage) the
    ``torch.cat`` is never materialised."""

    def __init__(self, in_channels, out_channels, depth=1, stride=2,
                 upsample=(convs.ConvTranspose3d, {"kernel_size": 2}), block=UNetStage, **kwargs):
        super().__init__()
        self.upsample = partialize(upsample)(in_channels, out_channels, stride=stride)
        self.block = partialize(block)(2 * out_channels, out_channels, depth=depth, **kwargs)

    def forward(self, deep, skip):
        fused = getattr(self.block, "forward_up_pair", None)
        if fused is not None:
            out = fused(skip, deep, self.upsample)
            if out is not None:
                return out
        up = self.upsample(deep)
        pair = getattr(self.block, "forward_pair", None)
        if pair is not None:
            return pair(skip, up)
        return self.block(torch.cat([skip, up], dim=1))


class UNetDecoder(nn.Module):
    def __init__(self, in_channels=(512, 256, 128, 64, 32), depth=(1, 1, 1, 1), strides=(2, 2, 2, 2),
                 upsample=None, block=None, spatial_size=None, **kwargs):
        super().__init__()
        n = len(in_channels) - 1
        sizes, cur = [], spatial_size
        for j in range(n):
            cur = _scale(cur, strides[j], False)
            sizes.append(cur)
        self.in_spatial_size = spatial_size
        self.out_spatial_size = sizes[-1] if sizes else spatial_size
        self.blocks = nn.ModuleList(
            UNetDecoderBlock(in_channels[j], in_channels[j + 1], depth[j], strides[j], upsample, block[j],
                             spatial_size=sizes[j], **kwargs)
            for j in range(n))

    def forward(self, feats):
        """feats: encoder outputs, shallow → deep.  Returns the list with every level below the
        bottleneck replaced by its decoded version (index 0 = full resolution)."""
        feats = list(feats)
        level = len(feats) - 1
        for stage in self.blocks:
            feats[level - 1] = stage(feats[level], feats[level - 1])
            level -= 1
        return feats


class UNet(nn.Module):
    """stem → encoder → decoder → head(s)."""

    def __init__(self, in_channels, out_channels, spatial_dims=3, spatial_size=None,
                 encoder_depth=(1, 1, 1, 1, 1), encoder_width=(32, 64, 128, 256, 512),
                 strides=(1, 2, 2, 2, 2), decoder_depth=(1, 1, 1, 1), stem=None, downsample=None,
                 block=None, upsample=None, head=None, num_deep_supr=False, **kwargs):
        super().__init__()
        if block is None:
            raise ValueError("UNet needs a `block` spec per stage (Factorizer supplies FactorizerStage)")
        self.spatial_dims = spatial_dims
        self.spatial_size = spatial_size
        n_enc, n_dec = len(encoder_depth), len(decoder_depth)
        if spatial_dims == 3:
            conv, tconv = convs.Conv3d, convs.ConvTranspose3d
        else:
            conv, tconv = getattr(nn, f"Conv{spatial_dims}d"), getattr(nn, f"ConvTranspose{spatial_dims}d")
        downsample = downsample or (conv, {"kernel_size": 2})
        upsample = upsample or (tconv, {"kernel_size": 2})
        head = head or (conv, {"kernel_size": 1})
        has_stem = stem not in (None, nn.Identity)
        stem_width = encoder_width[0] if has_stem else in_channels

        # creation order = reference order: stem, encoder, decoder, head(s)
        self.stem = partialize(stem)(in_channels, stem_width) if has_stem else nn.Identity()
        self.encoder = UNetEncoder(stem_width, encoder_width, encoder_depth, strides, downsample,
                                   [block[i] for i in range(n_enc)], spatial_size=spatial_size, **kwargs)
        self.decoder = UNetDecoder(encoder_width[::-1], decoder_depth, strides[::-1][:n_dec], upsample,
                                   [block[n_enc + j] for j in range(n_dec)],
                                   spatial_size=self.encoder.out_spatial_size, **kwargs)
        make_head = partialize(head)
        if num_deep_supr in (False, None):
            self.num_deep_supr = False
            self.head = make_head(encoder_width[0], out_channels)
        else:
            # reference quirk kept: True → attribute 3, but range(True) builds ONE head (unet.py:255-258)
            self.num_deep_supr = 3 if num_deep_supr is True else num_deep_supr
            self.heads = nn.ModuleList(make_head(encoder_width[j], out_channels) for j in range(num_deep_supr))

    @staticmethod
    def _mixed_precision_input(x):
        """Mixed precision (BASELINE configs[4]): under ``torch.autocast("cuda", dtype=torch.bfloat16)`` — or when
        handed a bfloat16 input — every activation of the network is STORED as bf16 while parameters,
        statistics, accumulation and the whole NMF iteration stay fp32 (include/factorizer_hip.h,
        FZ_STORE_BF16).  float16 is refused: eps = 1e-16 of the NMF ratios underflows in it and all-zero
        patches turn into NaN (matrix_factorization.py:200,236)."""
        if x.is_cuda and torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            if dt == torch.float16:
                raise RuntimeError("Factorizer under float16 autocast: the NMF eps (1e-16) underflows in fp16; "
                                   "use torch.autocast('cuda', dtype=torch.bfloat16)")
            if dt == torch.bfloat16 and x.dtype == torch.float32:
                x = x.to(torch.bfloat16)
        if x.dtype == torch.float16:
            raise RuntimeError("Factorizer on a float16 input: use bfloat16 (or float32) activations")
        return x

    def _stem_prologue(self, x):
        """(ln1 weight, ln1 bias, eps, in_proj weight) when the stem is the native k3 convolution to 32 channels and its output
        goes straight into a FactorizerBlock (stage 0: Identity down-sampling, no adapter / position embedding in front of the
        first block): the stem's launch then also forms that block's first layer (pointwise.BlockPrologue), else None."""
        from . import convs as _convs
        from . import pointwise as _PW
        st = self.stem
        one = lambda t, v: all(int(e) == v for e in t)  # noqa: E731
        if type(st) is not _convs.Conv3d or not (one(st.kernel_size, 3) and one(st.stride, 1) and one(st.padding, 1) and one(st.dilation, 1)
                                                 and st.groups == 1 and st.padding_mode == "zeros" and st.out_channels == 32
                                                 and st.in_channels % 2 == 0 and st.weight.dtype == torch.float32):
            return None
        if st._forward_hooks or st._forward_pre_hooks or x.dim() != 5 or x.shape[-1] % 4 or not _PW.conv3_prologue_ok(x, st.weight):
            return None
        e0 = self.encoder.blocks[0] if len(self.encoder.blocks) else None
        stage = getattr(e0, "block", None)
        if e0 is None or not isinstance(getattr(e0, "downsample", None), nn.Identity) or hasattr(stage, "adapter") or e0._forward_hooks:
            return None
        fn = getattr(stage, "_first_block_prologue", None)
        if fn is None or stage._forward_hooks:
            return None
        from .utils import TensorSpec   # (the stem's output as the block will see it: real batch, no allocation)
        return fn(TensorSpec((x.shape[0], 32, *x.shape[2:]), x.dtype, x.device))

    def forward_features(self, x):
        x = self._mixed_precision_input(x)
        pro = self._stem_prologue(x) if x.is_cuda else None
        if pro is not None:
            from . import pointwise as _PW
            y, t, st = _PW.ConvK3Fn.apply(x.contiguous(), self.stem.weight, self.stem.bias, pro)
            with _PW.BlockPrologue(y, t, st):
                return self.decoder(self.encoder(y))
        return self.decoder(self.encoder(self.stem(x)))

    def _head_fusion_target(self):
        """(block module, (head weight, head bias)) when the full-resolution decoder output feeds ONLY a k1 head of <= 4
        channels and is produced by a FactorizerBlock: that block can apply the head inside its last launch
        (pointwise.HeadFusion; unet.py:253,274), else None."""
        from . import convs as _convs
        from . import pointwise as _PW
        h = getattr(self, "head", None)
        if self.num_deep_supr or type(h) is not _convs.Conv3d or len(self.decoder.blocks) == 0:
            return None
        one = lambda t, v: all(int(e) == v for e in t)  # noqa: E731
        if not (one(h.kernel_size, 1) and one(h.stride, 1) and one(h.padding, 0) and h.groups == 1 and h.out_channels <= 4
                and h.weight.dtype == torch.float32) or h._forward_hooks or h._forward_pre_hooks:
            return None
        stage = getattr(self.decoder.blocks[-1], "block", None)
        blocks = getattr(stage, "blocks", None)
        if not blocks:
            return None
        blk = blocks[-1]
        if type(blk).__name__ != "FactorizerBlock" or blk._forward_hooks or blk._forward_pre_hooks or stage._forward_hooks:
            return None
        return blk, (h.weight, h.bias), _PW

    def forward(self, x):
        tgt = self._head_fusion_target() if x.is_cuda else None
        if tgt is not None:
            blk, params, _PW = tgt
            with _PW.HeadFusion(blk, params) as slot:
                feats = self.forward_features(x)
            if slot.logits is not None:
                return slot.logits
            return self.head(feats[0])
        feats = self.forward_features(x)
        if self.num_deep_supr:
            return [h(feats[j]) for j, h in enumerate(self.heads)]
        return self.head(feats[0])


class Factorizer(UNet):
    """U-shaped segmentation network whose every stage is a FactorizerStage; only the bottleneck
    stage carries the positional embedding (factorizer.py:125-171)."""

    def __init__(self, in_channels, out_channels, spatial_size, encoder_depth=(1, 1, 1, 1, 1),
                 encoder_width=(32, 64, 128, 256, 512), strides=(1, 2, 2, 2, 2),
                 decoder_depth=(1, 1, 1, 1), stem=None, downsample=None, upsample=None, head=None,
                 pos_embed=PositionalEmbedding, num_deep_supr=False, **kwargs):
        nd = len(spatial_size)
        if stem is None:
            conv = convs.Conv3d if nd == 3 else getattr(nn, f"Conv{nd}d")
            stem = (conv, {"kernel_size": 3, "padding": 1, "bias": False})
        plain = (FactorizerStage, kwargs)
        bottleneck = (FactorizerStage, {"pos_embed": pos_embed, **kwargs})
        stages = [plain] * (len(encoder_depth) - 1) + [bottleneck] + [plain] * len(decoder_depth)
        super().__init__(in_channels, out_channels, spatial_dims=nd, spatial_size=spatial_size,
                         encoder_depth=encoder_depth, encoder_width=encoder_width, strides=strides,
                         decoder_depth=decoder_depth, stem=stem, downsample=downsample, block=stages,
                         upsample=upsample, head=head, num_deep_supr=num_deep_supr)
