"""U-shaped encoder/decoder shell and the ``Factorizer`` model — host mirrors of the
reference's unet.py:11-276 and factorizer.py:125-171 (same constructor arguments, module
names and construction order; state_dict-compatible)."""
from __future__ import annotations

import math
from collections.abc import Sequence

import torch
from torch import nn

from . import convs
from .blocks import FactorizerStage
from .layers import PositionalEmbedding
from .utils import as_tuple, partialize


class Same:
    """Indexable that returns the same block spec for every stage (unet.py:11-17)."""

    def __init__(self, block):
        self.block = block

    def __getitem__(self, *args, **kwargs):
        return self.block


class UNetStage(nn.Module):
    def __init__(self, in_channels, out_channels, depth=1, block=None, **kwargs):
        super().__init__()
        if block is None:
            raise ValueError("UNetStage needs a `block` (the CNN DoubleConv default of the reference "
                             "is outside this build's scope)")
        block = partialize(block)
        self.blocks = nn.Sequential(block(in_channels, out_channels, **kwargs))
        for _ in range(1, depth):
            self.blocks.append(block(out_channels, out_channels, **kwargs))

    def forward(self, x):
        return self.blocks(x)


class UNetEncoderBlock(nn.Module):
    """[stride-2 k2 conv unless the stage stride is 1] → stage block (unet.py:36-59)."""

    def __init__(self, in_channels, out_channels, depth=1, stride=2,
                 downsample=(nn.Conv3d, {"kernel_size": 2}), block=UNetStage, **kwargs):
        super().__init__()
        block = partialize(block)
        downsample = nn.Identity if math.prod(as_tuple(stride)) == 1 else downsample
        # the reference hard-codes stride=2 here regardless of `stride` (unet.py:53)
        self.downsample = partialize(downsample)(in_channels, out_channels, stride=2)
        self.block = block(out_channels, out_channels, depth=depth, **kwargs)

    def forward(self, x):
        return self.block(self.downsample(x))


class UNetEncoder(nn.Module):
    def __init__(self, in_channels, out_channels=(32, 64, 128, 256, 512), depth=(1, 1, 1, 1, 1),
                 strides=(1, 2, 2, 2, 2), downsample=None, block=None, **kwargs):
        super().__init__()
        channels = [in_channels, *out_channels]
        self.in_spatial_size = kwargs.get("spatial_size")
        self.blocks = nn.ModuleList()
        for i in range(len(out_channels)):
            if isinstance(kwargs.get("spatial_size"), Sequence):
                kwargs["spatial_size"] = tuple(d // strides[i] for d in kwargs["spatial_size"])
            self.blocks.append(UNetEncoderBlock(channels[i], channels[i + 1], depth[i], strides[i],
                                                downsample, block[i], **kwargs))
        self.out_spatial_size = kwargs.get("spatial_size")

    def forward(self, x):
        out = [self.blocks[0](x)]
        for blk in self.blocks[1:]:
            out.append(blk(out[-1]))
        return out


class UNetDecoderBlock(nn.Module):
    """transposed k2 conv → cat([skip, up]) → stage block (unet.py:107-130)."""

    def __init__(self, in_channels, out_channels, depth=1, stride=2,
                 upsample=(nn.ConvTranspose3d, {"kernel_size": 2}), block=UNetStage, **kwargs):
        super().__init__()
        upsample = partialize(upsample)
        block = partialize(block)
        self.upsample = upsample(in_channels, out_channels, stride=stride)
        self.block = block(2 * out_channels, out_channels, depth=depth, **kwargs)

    def forward(self, x1, x2):
        x1 = self.upsample(x1)
        if hasattr(self.block, "forward_pair"):
            return self.block.forward_pair(x2, x1)
        return self.block(torch.cat([x2, x1], dim=1))


class UNetDecoder(nn.Module):
    def __init__(self, in_channels=(512, 256, 128, 64, 32), depth=(1, 1, 1, 1), strides=(2, 2, 2, 2),
                 upsample=None, block=None, **kwargs):
        super().__init__()
        self.in_spatial_size = kwargs.get("spatial_size")
        self.blocks = nn.ModuleList()
        for i in range(len(in_channels) - 1):
            if isinstance(kwargs.get("spatial_size"), Sequence):
                kwargs["spatial_size"] = tuple(d * strides[i] for d in kwargs["spatial_size"])
            self.blocks.append(UNetDecoderBlock(in_channels[i], in_channels[i + 1], depth[i], strides[i],
                                                upsample, block[i], **kwargs))
        self.out_spatial_size = kwargs.get("spatial_size")

    def forward(self, x):
        out = list(x)
        for i, blk in enumerate(self.blocks):
            out[-2 - i] = blk(out[-1 - i], out[-2 - i])
        return out


class UNet(nn.Module):
    """stem → encoder → decoder → head(s) (unet.py:177-276)."""

    def __init__(self, in_channels, out_channels, spatial_dims=3, spatial_size=None,
                 encoder_depth=(1, 1, 1, 1, 1), encoder_width=(32, 64, 128, 256, 512),
                 strides=(1, 2, 2, 2, 2), decoder_depth=(1, 1, 1, 1), stem=None, downsample=None,
                 block=None, upsample=None, head=None, num_deep_supr=False, **kwargs):
        super().__init__()
        self.spatial_dims = spatial_dims
        self.spatial_size = spatial_size
        for s in strides:
            if math.prod(as_tuple(s)) not in (1, 2 ** spatial_dims) and as_tuple(s) != (2,):
                raise ValueError("only strides of 1 or 2 are coherent in this U-shape (unet.py:53,123)")
        conv = convs.Conv3d if spatial_dims == 3 else getattr(nn, f"Conv{spatial_dims}d")
        tconv = convs.ConvTranspose3d if spatial_dims == 3 else getattr(nn, f"ConvTranspose{spatial_dims}d")
        if stem in (None, nn.Identity):
            stem = nn.Identity
            stem_width = in_channels
        else:
            stem_width = encoder_width[0]
        if downsample is None:
            downsample = (conv, {"kernel_size": 2})
        if block is None:
            raise ValueError("UNet needs a `block` spec per stage (Factorizer supplies FactorizerStage)")
        if upsample is None:
            upsample = (tconv, {"kernel_size": 2})
        if head is None:
            head = (conv, {"kernel_size": 1})
        stem = partialize(stem)
        head = partialize(head)
        self.stem = stem(in_channels, stem_width)
        self.encoder = UNetEncoder(stem_width, encoder_width, encoder_depth, strides, downsample,
                                   [block[i] for i in range(len(encoder_depth))],
                                   spatial_size=spatial_size, **kwargs)
        self.decoder = UNetDecoder(encoder_width[::-1], decoder_depth, strides[::-1][:len(decoder_depth)],
                                   upsample,
                                   [block[i + len(encoder_depth)] for i in range(len(decoder_depth))],
                                   spatial_size=self.encoder.out_spatial_size, **kwargs)
        if num_deep_supr in (False, None):
            self.num_deep_supr = False
            self.head = head(encoder_width[0], out_channels)
        else:
            # mirrors the reference, including `range(True)` == one head for num_deep_supr=True
            self.num_deep_supr = 3 if num_deep_supr is True else num_deep_supr
            self.heads = nn.ModuleList()
            for j in range(num_deep_supr):
                self.heads.append(head(encoder_width[j], out_channels))

    def forward_features(self, x):
        return self.decoder(self.encoder(self.stem(x)))

    def forward(self, x):
        y = self.forward_features(x)
        if self.num_deep_supr:
            return [head(y[j]) for j, head in enumerate(self.heads)]
        return self.head(y[0])


class Factorizer(UNet):
    """U-shaped segmentation network whose every stage is a FactorizerStage
    (factorizer.py:125-171); the bottleneck stage gets the positional embedding."""

    def __init__(self, in_channels, out_channels, spatial_size, encoder_depth=(1, 1, 1, 1, 1),
                 encoder_width=(32, 64, 128, 256, 512), strides=(1, 2, 2, 2, 2),
                 decoder_depth=(1, 1, 1, 1), stem=None, downsample=None, upsample=None, head=None,
                 pos_embed=PositionalEmbedding, num_deep_supr=False, **kwargs):
        nd = len(spatial_size)
        if stem is None:
            stem = (convs.Conv3d if nd == 3 else getattr(nn, f"Conv{nd}d"),
                    {"kernel_size": 3, "padding": 1, "bias": False})
        n_enc, n_dec = len(encoder_depth), len(decoder_depth)
        block = ((n_enc - 1) * [(FactorizerStage, kwargs)]
                 + [(FactorizerStage, {"pos_embed": pos_embed, **kwargs})]
                 + n_dec * [(FactorizerStage, kwargs)])
        super().__init__(in_channels, out_channels, spatial_dims=nd, spatial_size=spatial_size,
                         encoder_depth=encoder_depth, encoder_width=encoder_width, strides=strides,
                         decoder_depth=decoder_depth, stem=stem, downsample=downsample, block=block,
                         upsample=upsample, head=head, num_deep_supr=num_deep_supr)
