"""Matricize / SWMatricize / Reshape — the reshape slot of FactMixer.

Host-side mirror of the reference's factorization/operations.py:147-434 (same constructor
arguments, attributes ``input_size`` / ``output_size``, methods ``forward`` /
``inverse_forward``).  Device tensors go through the single-pass gfx950 kernels
(csrc/swm.hip via functional.py); the reference issues roll + einops + cat copies instead.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
from torch import nn

from . import composed, functional as Fn


def _ntuple(v, n):
    if isinstance(v, (tuple, list)):
        if len(v) != n:
            raise ValueError(f"expected {n} values, got {v}")
        return tuple(v)
    return (v,) * n


def _resolve_geometry(input_size, num_heads, head_dim, grid_size, patch_size):
    """(h, d, grid, patch) from the partially specified groups, the way
    Reshape.infer_dims does (operations.py:199-236) — but indivisible sizes raise here, at
    construction, instead of failing later inside einops."""
    assert (num_heads, head_dim) != (None, None), "'num_heads' or 'head_dim' must be specified."
    assert (grid_size, patch_size) != (None, None), "'grid_size' or 'kernel_size' must be specified."
    C = input_size[1]
    spatial = tuple(input_size[2:])
    nd = len(spatial)
    if C is None or any(s is None for s in spatial):
        raise ValueError("Matricize needs static channel and spatial sizes: (None, C, *spatial)")
    if head_dim is not None:
        d = max(int(head_dim), 1)
        h = max(int(num_heads), 1) if num_heads is not None else C // d
    else:
        h = max(int(num_heads), 1)
        d = C // h
    if h * d != C:
        raise ValueError(f"channels={C} cannot be split into num_heads={h} x head_dim={d}")
    grid, patch = [], []
    for s, g, p in zip(spatial, _ntuple(grid_size, nd), _ntuple(patch_size, nd)):
        if p is not None:
            p = max(int(p), 1)
            g = max(int(g), 1) if g is not None else s // p
        elif g is not None:
            g = max(int(g), 1)
            p = s // g
        else:
            raise AssertionError("'grid_size' or 'kernel_size' must be specified.")
        if g * p != s:
            raise ValueError(f"spatial size {spatial} is not divisible into grid x patch ({g} x {p})")
        grid.append(g)
        patch.append(p)
    return h, d, tuple(grid), tuple(patch)


def _norm_shift(s, nd):
    if s is None:
        return (0,) * nd
    return tuple(int(v) for v in _ntuple(s, nd))


class _WindowedMatricize(nn.Module):
    """Shared machinery: a list of cyclic-shift windows over one static geometry."""

    def _setup(self, input_size, num_heads, head_dim, grid_size, patch_size, shift_list):
        self.input_size = tuple(input_size)
        h, d, grid, patch = _resolve_geometry(input_size, num_heads, head_dim, grid_size, patch_size)
        nd = len(patch)
        shifts = [_norm_shift(s, nd) for s in shift_list]
        self.geometry = Fn.Geometry(input_size[1], input_size[2:], d, patch, shifts)
        self.num_heads, self.head_dim = h, d
        self.grid_size, self.patch_size = grid, patch
        # '(b h) (g..) d (p..)': the batch·head group is unknown until forward
        self.output_size = (None, self.geometry.G, d, self.geometry.P)

    def _check_input(self, x):
        if tuple(x.shape[1:]) != tuple(self.input_size[1:]):
            raise ValueError(f"expected input of shape (B, {', '.join(map(str, self.input_size[1:]))}), "
                             f"got {tuple(x.shape)}")

    def _is_view(self) -> bool:
        """One unshifted window over a 1-patch grid — the reference's default FactMixer reshape
        Matricize(num_heads=1, grid_size=1) (factorizer.py:17) and every `grid_size=1` form: the rearrangement
        'b (h d) (g p).. -> (b h) (g..) d (p..)' moves nothing, so it is a view, not a kernel."""
        geo = self.geometry
        return geo.nshift == 1 and geo.G == 1 and not any(geo.shifts[0])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        self._check_input(x)
        x = x.as_subclass(torch.Tensor)
        if self._is_view() and x.is_contiguous():
            geo = self.geometry
            return x.view(x.shape[0] * geo.h, 1, geo.d, geo.P)
        if x.is_cuda and x.numel():
            return Fn.swm_forward(x, self.geometry)
        return composed.swm_forward(x, self.geometry)

    def inverse_forward(self, y: torch.Tensor) -> torch.Tensor:
        y = y.as_subclass(torch.Tensor)
        geo = self.geometry
        if y.shape[0] % (geo.nshift * geo.h) or tuple(y.shape[1:]) != (geo.G, geo.d, geo.P):
            raise ValueError(f"expected (num_shifts*B*{geo.h}, {geo.G}, {geo.d}, {geo.P}), got {tuple(y.shape)}")
        if self._is_view() and y.is_contiguous():
            # ((0.0 + z_0)) / 1 of operations.py:426-433 is the identity on the values
            return y.view(y.shape[0] // geo.h, geo.C, *geo.spatial)
        if not y.numel():
            return composed.swm_inverse(y, geo)
        if y.is_cuda and y.dtype in (torch.float32, torch.bfloat16):
            return Fn.swm_inverse(y, geo)
        if y.is_cuda:
            return Fn.swm_inverse(y.float(), geo).to(y.dtype)
        return composed.swm_inverse(y, geo)


class Matricize(_WindowedMatricize):
    """Tensor -> batch of (head_dim x patch) matrices, optional cyclic shift
    (operations.py:283-355).  ``(B, h·d, g0·p0, ...) -> (B·h, g0·g1·.., d, p0·p1·..)``."""

    def __init__(self, input_size: Sequence[int], num_heads: Optional[int] = None,
                 head_dim: Optional[int] = None, grid_size=None, patch_size=None, shifts=None, **kwargs):
        super().__init__()
        if kwargs:
            raise TypeError(f"unexpected arguments {sorted(kwargs)}")
        self._setup(input_size, num_heads, head_dim, grid_size, patch_size, [shifts])
        if shifts is not None:
            self.shifts = self.geometry.shifts[0]
            self.shifts_inv = tuple(-s for s in self.shifts)
            self.dims = tuple(range(2, 2 + len(self.shifts)))


class SWMatricize(_WindowedMatricize):
    """Shifted-window matricize (operations.py:358-434): every window's matricization
    concatenated on dim 0; ``inverse_forward`` averages the windows back.  Default windows:
    ``[None, patch_size // 2]`` (operations.py:397-398)."""

    def __init__(self, input_size: Sequence[int], num_heads: Optional[int] = None,
                 head_dim: Optional[int] = None, grid_size=None, patch_size=None, shifts=None, **kwargs):
        super().__init__()
        if kwargs:
            raise TypeError(f"unexpected arguments {sorted(kwargs)}")
        nd = len(input_size) - 2
        if shifts is None:
            _, _, _, patch = _resolve_geometry(input_size, num_heads, head_dim, grid_size, patch_size)
            shifts = [None, tuple(p // 2 for p in patch)]
        shifts = list(shifts)
        if len(shifts) == 0:
            raise ValueError("at least one window is required")
        self._setup(input_size, num_heads, head_dim, grid_size, patch_size, shifts)
        # per-window views, for parity with the reference's `shifted_windows` ModuleList
        self.shifted_windows = nn.ModuleList(
            Matricize(input_size, num_heads=self.num_heads, head_dim=self.head_dim,
                      patch_size=self.patch_size, shifts=(None if s is None else s))
            for s in shifts)


class Reshape(nn.Module):
    """Generic einops-pattern reshape with optional cyclic shift and an inverse
    (operations.py:147-280).  Not on the hot path: composed from einops + torch.roll."""

    def __init__(self, input_size, equation: Optional[str] = None, shifts=None, dims=None, **kwargs):
        super().__init__()
        self.input_size = input_size
        self.equation = equation
        self.axes_lengths = dict(kwargs)
        if equation is None:
            self.output_size = input_size
        else:
            left, right = (s.strip() for s in equation.split("->"))
            self.left, self.right = left, right
            self.equation_inv = f"{right} -> {left}"
            self.output_size = None  # resolved lazily from the first forward
        if shifts is not None:
            self.shifts = tuple(shifts)
            self.shifts_inv = tuple(-s for s in self.shifts)
            self.dims = tuple(dims)
        self._known = None

    def forward(self, x):
        import einops
        if hasattr(self, "shifts"):
            x = torch.roll(x, self.shifts, self.dims)
        if self.equation is None:
            return x
        out = einops.rearrange(x, self.equation, **self.axes_lengths)
        if self._known is None:
            self._known = einops.parse_shape(x, self.left.replace("(", " ").replace(")", " ")) \
                if "(" not in self.left else dict(self.axes_lengths)
            self._in_shape = tuple(x.shape)
            self.output_size = (None, *out.shape[1:])
        return out

    def inverse_forward(self, y):
        import einops
        if self.equation is not None:
            # all axis lengths except the leading batch can be recovered from the static input size
            lengths = dict(self.axes_lengths)
            y = einops.rearrange(y, self.equation_inv, **self._inverse_lengths(lengths))
        if hasattr(self, "shifts"):
            y = torch.roll(y, self.shifts_inv, self.dims)
        return y

    def _inverse_lengths(self, lengths):
        import re
        groups = re.findall(r"\(([^)]+)\)|(\w+)", self.left)
        out = dict(lengths)
        for (grp, single), size in zip(groups, self.input_size):
            names = grp.split() if grp else [single]
            if size is None:
                continue
            unknown = [n for n in names if n not in out]
            if len(unknown) == 1:
                prod = 1
                for n in names:
                    if n in out:
                        prod *= out[n]
                out[unknown[0]] = size // prod
        # einops only accepts lengths of axes that appear inside a composition on the input side
        comp = set()
        for grp, _ in re.findall(r"\(([^)]+)\)|(\w+)", self.right):
            if grp:
                comp.update(grp.split())
        return {k: v for k, v in out.items() if k in comp}
