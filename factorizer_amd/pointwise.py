"""Per-voxel ops on channels-first tensors (LayerNorm over C, 1×1 GEMMs, MLP).

Round-1 state: these dispatch to composed PyTorch ops (rocBLAS/ATen on device); the fused
gfx950 kernels (LN→GEMM, GEMM→GELU→GEMM→residual) replace them behind the same functions.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def linear_cf(x, weight, bias=None):
    B, C = x.shape[:2]
    y = torch.matmul(weight.reshape(weight.shape[0], weight.shape[1]), x.reshape(B, C, -1))
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    return y.reshape(B, weight.shape[0], *x.shape[2:])


def layernorm_cf(x, weight, bias, eps):
    y = F.layer_norm(x.movedim(1, -1), (x.shape[1],), weight, bias, eps)
    return y.movedim(-1, 1).contiguous()


def mlp_cf(x, w1, b1, w2, b2):
    return linear_cf(F.gelu(linear_cf(x, w1, b1)), w2, b2)
