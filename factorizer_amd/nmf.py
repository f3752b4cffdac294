"""Differentiable matrix-factorization layers: ``MatrixFactorization`` / ``NMF`` with the
reference's constructor surface, initialisers and solver string table
(factorization/matrix_factorization.py:19-58, 108-136, 194-247, 454-685).

On device the whole unrolled iteration loop (init → T×[update U, update V] → u vᵀ) is ONE
gfx950 kernel that keeps the matrix in registers (csrc/nmf_core.h), and the backward is a
second kernel that recomputes the iterations instead of saving autograd intermediates.
"""
from __future__ import annotations

import math
from collections.abc import Sequence
from contextlib import nullcontext
from typing import Optional

import torch
from torch import Tensor, nn

from . import composed, functional as Fn
from .utils import as_tuple, is_partializable, partialize


# ---- initialisers ------------------------------------------------------------------------
class Initializer(nn.Module):
    def forward(self, x: Tensor):
        raise NotImplementedError(f"Subclass {self.__class__.__name__} must implement this method.")


class RandomInit(Initializer):
    """u0 (M,R) then v0 (N,R) drawn once at construction from the global RNG and kept as
    BUFFERS (part of the state_dict); forward broadcasts them over the batch
    (matrix_factorization.py:28-58)."""

    def __init__(self, rank: int, size, method="uniform"):
        super().__init__()
        method = as_tuple(method)
        if len(method) == 1:
            method = (method[0], method[0])
        elif len(method) != 2:
            raise ValueError("`method` not valid.")
        self.method = method
        self.register_buffer("u0", torch.empty(size[0], rank))
        getattr(nn.init, f"{method[0]}_")(self.u0)
        self.register_buffer("v0", torch.empty(size[1], rank))
        getattr(nn.init, f"{method[1]}_")(self.v0)

    def forward(self, x: Tensor):
        lead = x.shape[:-2]
        return self.u0.expand(*lead, *self.u0.shape), self.v0.expand(*lead, *self.v0.shape)


# ---- solvers (composed form; the device kernels implement MU and HALS natively) ------------
class BCDSolver(nn.Module):
    """Block-coordinate alternation: factor order (0,1) = U then V, V sees the new U
    (matrix_factorization.py:108-136)."""

    def __init__(self, factor=(0, 1), *args, **kwargs):
        super().__init__()
        self.factor = as_tuple(factor)
        assert set(self.factor).issubset({0, 1}), "`factor` elements must be 0 or 1."

    def update_u(self, x, u, v):
        raise NotImplementedError(f"Subclass {self.__class__.__name__} must implement this method.")

    def update_v(self, x, u, v):
        return self.update_u(x.mT, v, u)

    def forward(self, x, factor_matrices):
        u, v = factor_matrices
        for j in self.factor:
            if j == 0:
                u = self.update_u(x, u, v)
            else:
                v = self.update_v(x, u, v)
        return u, v


class MultiplicativeUpdate(BCDSolver):
    native_id = "mu"

    def __init__(self, factor=(0, 1), eps: float = 1e-16, **kwargs):
        super().__init__(factor=factor)
        self.eps = eps

    def update_u(self, x, u, v):
        return composed.mu_update(x, u, v, self.eps)


class CoordinateDescent(BCDSolver):
    def __init__(self, factor=(0, 1), eps: float = 1e-16, project=None, **kwargs):
        super().__init__(factor=factor)
        self.eps = eps
        self.project = partialize(nn.Identity if project is None else project)()

    @property
    def native_id(self):
        return "hals" if isinstance(self.project, nn.ReLU) else None

    def update_u(self, x, u, v):
        return composed.cd_update(x, u, v, self.eps, self.project)


class Compose(nn.Module):
    """Apply several solvers in sequence per iteration (matrix_factorization.py:386-400)."""

    def __init__(self, solvers, **kwargs):
        super().__init__()
        self.solvers = nn.ModuleList(partialize(s)(**kwargs) for s in solvers)

    def forward(self, x, factor_matrices):
        for s in self.solvers:
            factor_matrices = s(x, factor_matrices)
        return factor_matrices


class _Unavailable:
    """Placeholder for solver/init keys of the reference that this build does not implement
    (svd/nndsvd init, fmu/wmu/smu/ls/nnls solvers)."""

    def __init__(self, key):
        self.key = key

    def __call__(self, *a, **k):
        raise NotImplementedError(
            f"'{self.key}' is outside the accelerated Factorizer hot path (uniform/normal init, "
            "mu / hals / cd solvers); see DESIGN.md 'out of scope'.")


INIT_DISPATCH_MAP = {
    "uniform": (RandomInit, {"method": "uniform"}),
    "normal": (RandomInit, {"method": "normal"}),
    "normal-uniform": (RandomInit, {"method": ("normal", "uniform")}),
    "uniform-normal": (RandomInit, {"method": ("uniform", "normal")}),
    "svd": _Unavailable("svd"),
    "nndsvd": _Unavailable("nndsvd"),
}

SOLVER_DISPATCH_MAP = {
    "mu": MultiplicativeUpdate,
    "mu-0": (MultiplicativeUpdate, {"factor": 0}),
    "mu-1": (MultiplicativeUpdate, {"factor": 1}),
    "cd": CoordinateDescent,
    "cd-0": (CoordinateDescent, {"factor": 0}),
    "cd-1": (CoordinateDescent, {"factor": 1}),
    "nncd": (CoordinateDescent, {"project": nn.ReLU}),
    "nncd-0": (CoordinateDescent, {"factor": 0, "project": nn.ReLU}),
    "nncd-1": (CoordinateDescent, {"factor": 1, "project": nn.ReLU}),
    "hals": (CoordinateDescent, {"project": nn.ReLU}),
    "hals-0": (CoordinateDescent, {"factor": 0, "project": nn.ReLU}),
    "hals-1": (CoordinateDescent, {"factor": 1, "project": nn.ReLU}),
    # the reference maps "wmu-0"/"wmu-1" to the plain multiplicative update (:598-599)
    "wmu-0": (MultiplicativeUpdate, {"factor": 0}),
    "wmu-1": (MultiplicativeUpdate, {"factor": 1}),
}
for _k in ("fmu", "fmu-0", "fmu-1", "wmu", "smu", "smu-0", "smu-1", "ls", "ls-0", "ls-1", "nnls",
           "nnls-0", "nnls-1"):
    SOLVER_DISPATCH_MAP[_k] = _Unavailable(_k)


def _parse_init(obj):
    return INIT_DISPATCH_MAP.get(obj, obj) if isinstance(obj, str) else obj


def _parse_solver(obj):
    """str | partial spec | sequence of those -> partial spec (matrix_factorization.py:634-685)."""
    if isinstance(obj, str):
        if obj not in SOLVER_DISPATCH_MAP:
            raise ValueError(f"unknown solver '{obj}'")
        return SOLVER_DISPATCH_MAP[obj]
    if is_partializable(obj):
        return obj
    if isinstance(obj, Sequence):
        out = []
        for item in obj:
            if isinstance(item, str):
                out.append(_parse_solver(item))
            elif is_partializable(item):
                out.append(item)
            else:
                raise ValueError
        return (Compose, {"solvers": out})
    raise ValueError


def relative_error(x, y, w=None, eps: float = 1e-16):
    """Per-sample relative L2 error (operations.py:99-122, norm2 :34-51)."""
    def norm2(t):
        t = t.flatten(1).square()
        if w is not None:
            t = t * w.flatten(1)
        return torch.sqrt(t.sum(1))
    return (norm2(x - y) + eps) / (norm2(x) + eps)


# ---- the layer -----------------------------------------------------------------------------
class MatrixFactorization(nn.Module):
    """X ≈ U Vᵀ by `num_iters` unrolled solver iterations (matrix_factorization.py:454-546)."""

    def __init__(self, size, rank: Optional[int] = None, compression: float = 10, init="normal",
                 solver="cd", num_iters: int = 5, num_grad_steps: Optional[int] = None,
                 verbose: bool = False, **kwargs):
        super().__init__()
        self.size = M, N = tuple(size)
        self.num_iters = num_iters
        self.num_grad_steps = num_iters if num_grad_steps is None else num_grad_steps
        assert (rank, compression) != (None, None), "'rank' or 'compression' must be specified."
        if rank is None:
            rank = max(math.ceil(M * N / (compression * (M + N))), 1)
        self.rank = rank
        self.compression = M * N / (rank * (M + N))
        self.init = partialize(_parse_init(init))(size=self.size, rank=rank)
        self.solver = partialize(_parse_solver(solver))(size=self.size, rank=rank)
        self.verbose = verbose

    # -- which path -------------------------------------------------------------------
    def _native_solver(self, x: Tensor):
        """Solver id if this call is covered by the gfx950 kernels, else None."""
        if not x.is_cuda or self.verbose:
            return None
        sid = getattr(self.solver, "native_id", None)
        if sid is None or not isinstance(self.init, RandomInit) or self.solver.factor != (0, 1):
            return None
        if x.dtype != torch.float32 or tuple(x.shape[-2:]) != self.size:
            return None
        G = min(max(self.num_grad_steps, 0), self.num_iters)
        if not Fn.nmf_supported(self.size[0], self.size[1], self.rank, self.num_iters, G):
            composed.warn_once(
                f"nmf{self.size}{self.rank}",
                f"NMF size={self.size} rank={self.rank} is outside the native kernel families; "
                "using the composed PyTorch path on device")
            return None
        return sid

    def _grad_steps(self, x):
        G = min(max(self.num_grad_steps, 0), self.num_iters)
        return G if (torch.is_grad_enabled() and x.requires_grad) else 0

    def context(self, it: int):
        return torch.no_grad() if it < self.num_iters - self.num_grad_steps + 1 else nullcontext()

    # -- API --------------------------------------------------------------------------
    def decompose(self, x: Tensor, *args, **kwargs):
        x = x.as_subclass(Tensor)
        sid = self._native_solver(x)
        if sid is not None:
            return Fn.nmf_decompose(x, self.init.u0, self.init.v0, self.num_iters,
                                    min(max(self.num_grad_steps, 0), self.num_iters), sid, self.solver.eps)
        with self.context(0):
            u, v = self.init(x)
        for it in range(1, self.num_iters + 1):
            with self.context(it):
                if self.verbose:
                    print(f"iter {it}, loss = {self.loss(x, u, v)}")
                u, v = self.solver(x, [u, v], *args, **kwargs)
        return u, v

    def reconstruct(self, u: Tensor, v: Tensor) -> Tensor:
        return u @ v.mT

    def loss(self, x, u, v, w=None):
        return relative_error(x, self.reconstruct(u, v), w)

    def forward(self, x: Tensor) -> Tensor:
        x = x.as_subclass(Tensor)
        sid = self._native_solver(x)
        if sid is not None:
            return Fn.nmf(x, self.init.u0, self.init.v0, self.num_iters,
                          min(max(self.num_grad_steps, 0), self.num_iters), sid, self.solver.eps)
        u, v = self.decompose(x)
        return self.reconstruct(u, v)


class NMF(MatrixFactorization):
    """Non-negative matrix factorization, X, U, V ≥ 0 (matrix_factorization.py:549-578)."""

    def __init__(self, size, rank: Optional[int] = None, compression: float = 10, num_iters: int = 5,
                 num_grad_steps: Optional[int] = None, init="uniform", solver="hals",
                 verbose: bool = False, **kwargs):
        super().__init__(size, rank=rank, compression=compression, num_iters=num_iters,
                         num_grad_steps=num_grad_steps, init=init, solver=solver, verbose=verbose,
                         **kwargs)
