"""Composed-PyTorch implementations of the hot-path ops (differentiable by autograd).

Used (1) for CPU tensors — BASELINE config 0 runs ``ft.NMF`` on CPU — and (2) on device only
for shapes/options outside the native kernel families, with a one-time
warning.  Hot-path device shapes never come here (tests assert the native launch counter).
"""
from __future__ import annotations

import warnings

import torch

_warned = set()


def warn_once(key: str, msg: str):
    if key not in _warned:
        _warned.add(key)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


# ---- matricize as roll + view + permute (operations.py:266-280, 321-325) -------------------
def matricize_window(x, geo, w):
    """One window: cyclic roll by +s then 'b (h d) (g p).. -> (b h) (g..) d (p..)'."""
    nd = len(geo.spatial)
    s = geo.shifts[w]
    if any(s):
        x = torch.roll(x, s, tuple(range(2, 2 + nd)))
    B = x.shape[0]
    shape = [B, geo.h, geo.d]
    for g, p in zip(geo.grid, geo.patch):
        shape += [g, p]
    x = x.reshape(shape)
    gdims = [3 + 2 * i for i in range(nd)]
    pdims = [4 + 2 * i for i in range(nd)]
    x = x.permute(0, 1, *gdims, 2, *pdims)
    return x.reshape(B * geo.h, geo.G, geo.d, geo.P)


def unmatricize_window(y, geo, w):
    nd = len(geo.spatial)
    B = y.shape[0] // geo.h
    y = y.reshape(B, geo.h, *geo.grid, geo.d, *geo.patch)
    # dims: 0 b, 1 h, 2..2+nd-1 g_i, 2+nd d, 3+nd.. p_i  ->  b h d g0 p0 g1 p1 ...
    order = [0, 1, 2 + nd]
    for i in range(nd):
        order += [2 + i, 3 + nd + i]
    x = y.permute(order).reshape(B, geo.C, *geo.spatial)
    s = geo.shifts[w]
    if any(s):
        x = torch.roll(x, tuple(-v for v in s), tuple(range(2, 2 + nd)))
    return x


def swm_forward(x, geo):
    return torch.cat([matricize_window(x, geo, w) for w in range(geo.nshift)], dim=0)


def swm_inverse(y, geo):
    """out = (((0.0 + z0) + z1) + ...) / num_shifts (operations.py:423-434)."""
    chunk = y.shape[0] // geo.nshift
    out = 0.0
    for w in range(geo.nshift):
        out = out + unmatricize_window(y[w * chunk:(w + 1) * chunk], geo, w)
    return out / geo.nshift


# ---- NMF half-steps (matrix_factorization.py:210-229, 241-247) -----------------------------
def mu_update(z, w, s, eps):
    a = z @ s
    b = s.mT @ s
    return (w * a + eps) / (w @ b + eps)


def cd_update(z, w, s, eps, project):
    a = z @ s
    b = s.mT @ s
    R = w.shape[-1]
    if R == 1:
        return project((a + eps) / (b + eps))
    cols = list(w.unbind(-1))
    for r in range(R):
        others = [j for j in range(R) if j != r]
        acc = torch.stack([cols[j] for j in others], dim=-1) @ b[..., others, r:r + 1]
        num = a[..., r:r + 1] - acc + eps
        den = b[..., r:r + 1, r:r + 1] + eps
        cols[r] = project(num / den).squeeze(-1)
    return torch.stack(cols, dim=-1)
