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


def _gram(x, v):
    """a = X V (M x R), b = V^T V (R x R): the two products every least-squares-type half-step needs"""
    return x @ v, v.mT @ v


def _pair_dot(p, q):
    """<p, q> over the last two dims, kept as a trailing singleton (operations.py:13-29)"""
    return (p * q).sum(dim=(-2, -1)).unsqueeze(-1)


class LeastSquares(BCDSolver):
    """Exact minimiser of ||X - U V^T|| over U, optionally projected ("ls", "nnls" with ReLU)
    (matrix_factorization.py:139-165).  Tall X uses the pseudo-inverse of V, wide X the normal
    equations — the reference's two branches, kept because they round differently."""

    def __init__(self, factor=(0, 1), eps: float = 1e-16, project=None, **kwargs):
        super().__init__(factor=factor)
        self.eps = eps
        self.project = partialize(nn.Identity if project is None else project)()

    def update_u(self, x, u, v):
        M, N = x.shape[-2:]
        if M >= N:
            sol = x @ torch.linalg.pinv(v).mT
        else:
            a, b = _gram(x, v)
            sol = torch.linalg.solve(b, a.mT).mT
        return self.project(sol)


class ProjectedGradient(BCDSolver):
    """One exact-line-search gradient step on ||X - U V^T||^2 followed by the projection
    (matrix_factorization.py:168-191): g = XV - U V^T V, step = <g,g> / <g, g V^T V>."""

    def __init__(self, factor=(0, 1), project=None, eps: float = 1e-16, **kwargs):
        super().__init__(factor=factor)
        self.eps = eps
        self.project = partialize(nn.Identity if project is None else project)()

    def update_u(self, x, u, v):
        a, b = _gram(x, v)
        g = a - u @ b
        step = (_pair_dot(g, g) + self.eps) / (_pair_dot(g, g @ b) + self.eps)
        return self.project(u + step.unsqueeze(-1) * g)


class FastMultiplicativeUpdate(BCDSolver):
    """The multiplicative update written as three-operand contractions, with an explicit V half-step
    (matrix_factorization.py:250-274).  Algebraically the "mu" update — num = u ∘ (X V) + eps, den = u (VᵀV) + eps, and the
    same with the roles swapped — it differs in rounding only: on device it runs the native MU kernels (pinned against the
    reference's own fmu outputs, goldens g8, at the 1e-4 bound)."""
    native_id = "mu"

    def __init__(self, factor=(0, 1), eps: float = 1e-16, **kwargs):
        super().__init__(factor=factor)
        self.eps = eps

    def update_u(self, x, u, v):
        num = torch.einsum("...ij,...ir,...jr->...ir", x, u, v) + self.eps
        den = torch.einsum("...is,...js,...jr->...ir", u, v, v) + self.eps
        return num / den

    def update_v(self, x, u, v):
        num = torch.einsum("...ij,...ir,...jr->...jr", x, u, v) + self.eps
        den = torch.einsum("...ir,...is,...js->...jr", u, u, v) + self.eps
        return num / den


class WeightedMultiplicativeUpdate(BCDSolver):
    """Multiplicative update of min ||W ∘ (X - U V^T)||^2, U, V >= 0 (matrix_factorization.py:277-316);
    `forward(x, (u, v), w=None)` — unit weights when `w` is omitted, and then it IS the "mu" update (num = u ∘ (X V) + eps,
    den = (U Vᵀ) V + eps): `MatrixFactorization` runs the native MU kernels for it when no weights are passed."""
    native_id = "mu"

    def __init__(self, factor=(0, 1), eps: float = 1e-16, **kwargs):
        super().__init__(factor=factor)
        self.eps = eps

    def update_u(self, x, u, v, w):
        num = u * ((w * x) @ v) + self.eps
        den = (w * (u @ v.mT)) @ v + self.eps
        return num / den

    def update_v(self, x, u, v, w):
        return self.update_u(x.mT, v, u, w.mT)

    def forward(self, x, factor_matrices, w=None):
        u, v = factor_matrices
        w = torch.ones_like(x) if w is None else w
        for j in self.factor:
            if j == 0:
                u = self.update_u(x, u, v, w)
            else:
                v = self.update_v(x, u, v, w)
        return u, v


class SemiMultiplicativeUpdate(BCDSolver):
    """Semi-NMF update (X of any sign, the updated factor >= 0), matrix_factorization.py:319-341:
    U <- U ∘ sqrt((a+ + U b-) / (a- + U b+)) with a = XV, b = V^T V split into positive/negative parts."""

    def __init__(self, factor=(0, 1), eps: float = 1e-16, **kwargs):
        super().__init__(factor=factor)
        self.eps = eps

    def update_u(self, x, u, v):
        a, b = _gram(x, v)
        num = torch.relu(a) + u @ torch.relu(-b) + self.eps
        den = torch.relu(-a) + u @ torch.relu(b) + self.eps
        return u * torch.sqrt(num / den)


class Compose(BCDSolver):
    """Apply several solvers in sequence per iteration (matrix_factorization.py:344-378); indexable like
    the reference's (`compose[i]`, `len(compose)`), `factor` = the member solvers' factor tuples."""

    def __init__(self, solvers=None, **kwargs):
        super().__init__()
        built = [partialize(s)(**kwargs) for s in as_tuple([] if solvers is None else solvers)]
        self.solvers = nn.ModuleList(built)
        self.factor = [getattr(m, "factor") for m in built]
        self.size, self.rank = kwargs.get("size"), kwargs.get("rank")

    def forward(self, x, factor_matrices):
        u, v = factor_matrices
        for m in self.solvers:
            u, v = m(x, (u, v))
        return u, v

    def __setitem__(self, idx, solver):
        self.solvers[idx] = solver

    def __getitem__(self, idx):
        return self.solvers[idx]

    def __len__(self):
        return len(self.solvers)


class SVD(nn.Module):
    """Truncated SVD layer (matrix_factorization.py:386-451): `torch.svd_lowrank` of rank R.  Like the
    reference it re-seeds the GLOBAL torch RNG with 42 before every decomposition (:434) — a side
    effect callers can observe, kept for drop-in behaviour."""

    def __init__(self, size, rank: Optional[int] = None, compression: float = 10, no_grad: bool = False,
                 verbose: bool = False):
        super().__init__()
        self.size = M, N = tuple(size)
        self.no_grad = no_grad
        assert (rank, compression) != (None, None), "'rank' or 'compression' must be specified."
        if rank is None:
            rank = max(math.ceil(M * N / (compression * (M + N))), 1)
        self.rank = rank
        self.compression = M * N / (rank * (M + N))
        self.verbose = verbose

    def context(self):
        return torch.no_grad() if self.no_grad else nullcontext()

    def decompose(self, x: Tensor):
        with self.context():
            torch.manual_seed(42)
            u, s, v = torch.svd_lowrank(x, self.rank)
            if self.verbose:
                print(f"loss = {self.loss(x, u, s, v)}")
        return u, s, v

    def reconstruct(self, u, s, v):
        return (u * s.unsqueeze(-2)) @ v.mT

    def loss(self, x, u, s, v):
        return relative_error(x, self.reconstruct(u, s, v))

    def forward(self, x: Tensor) -> Tensor:
        return self.reconstruct(*self.decompose(x))


class SVDInit(Initializer):
    """U0 = U sqrt(S), V0 = V sqrt(S) from the truncated SVD of the input (matrix_factorization.py:61-71)"""

    def __init__(self, size, rank: Optional[int] = None):
        super().__init__()
        self.svd = SVD(size=size, rank=rank)

    def forward(self, x: Tensor):
        u, s, v = self.svd.decompose(x)
        r = torch.sqrt(s).unsqueeze(-2)
        return u * r, v * r


class NNDSVDInit(Initializer):
    """Non-negative double SVD start (matrix_factorization.py:74-100), input (B, M, N): per component keep
    the sign pattern — positive parts (a+, b+) or negative parts (a-, b-) — with the larger
    ||a±||·||b±||, chosen per batch element."""

    def __init__(self, size, rank: Optional[int] = None):
        super().__init__()
        self.svd = SVD(size, rank)

    def forward(self, x: Tensor):
        u, s, v = self.svd.decompose(x)
        r = torch.sqrt(s).unsqueeze(-2)
        u, v = u * r, v * r
        up, un, vp, vn = torch.relu(u), torch.relu(-u), torch.relu(v), torch.relu(-v)
        # column norms over the row axis: (B, R)
        pos = up.square().sum(-2).sqrt() * vp.square().sum(-2).sqrt()
        neg = un.square().sum(-2).sqrt() * vn.square().sum(-2).sqrt()
        keep_pos = (pos >= neg).unsqueeze(-2)
        return torch.where(keep_pos, up, un), torch.where(keep_pos, vp, vn)


INIT_DISPATCH_MAP = {
    "uniform": (RandomInit, {"method": "uniform"}),
    "normal": (RandomInit, {"method": "normal"}),
    "normal-uniform": (RandomInit, {"method": ("normal", "uniform")}),
    "uniform-normal": (RandomInit, {"method": ("uniform", "normal")}),
    "svd": SVDInit,
    "nndsvd": NNDSVDInit,
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
    "fmu": FastMultiplicativeUpdate,
    "fmu-0": (FastMultiplicativeUpdate, {"factor": 0}),
    "fmu-1": (FastMultiplicativeUpdate, {"factor": 1}),
    "wmu": WeightedMultiplicativeUpdate,
    # the reference maps "wmu-0"/"wmu-1" to the plain multiplicative update (:598-599)
    "wmu-0": (MultiplicativeUpdate, {"factor": 0}),
    "wmu-1": (MultiplicativeUpdate, {"factor": 1}),
    "smu": SemiMultiplicativeUpdate,
    "smu-0": (SemiMultiplicativeUpdate, {"factor": 0}),
    "smu-1": (SemiMultiplicativeUpdate, {"factor": 1}),
    "ls": LeastSquares,
    "ls-0": (LeastSquares, {"factor": 0}),
    "ls-1": (LeastSquares, {"factor": 1}),
    "nnls": (LeastSquares, {"project": nn.ReLU}),
    "nnls-0": (LeastSquares, {"factor": 0, "project": nn.ReLU}),
    "nnls-1": (LeastSquares, {"factor": 1, "project": nn.ReLU}),
}


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
        self._wide = False

    # -- which path -------------------------------------------------------------------
    def _native_solver(self, x: Tensor):
        """Solver id if this call is covered by the gfx950 kernels, else None; `self._wide` then says which
        family: the wave-resident kernels (False) or the split-N kernels for wide matrices (True)."""
        self._wide = False
        if not x.is_cuda or self.verbose or not x.numel():
            return None
        sid = getattr(self.solver, "native_id", None)
        if sid is None or not isinstance(self.init, RandomInit) or self.solver.factor != (0, 1):
            return None
        # float32, or bfloat16 STORAGE: the kernels then load/store bf16 and factorise in fp32
        # (u0 / v0 stay the fp32 buffers; SURVEY.md §5, eps at matrix_factorization.py:200,236)
        if x.dtype not in (torch.float32, torch.bfloat16) or tuple(x.shape[-2:]) != self.size:
            return None
        if self.init.u0.dtype != torch.float32:
            return None
        G = min(max(self.num_grad_steps, 0), self.num_iters)
        if not Fn.nmf_supported(self.size[0], self.size[1], self.rank, self.num_iters, G):
            # wider than one wavefront can hold (global Matricize, num_heads= forms): the split-N kernels
            if x.dtype == torch.float32 and Fn.gnmf_supported(self.size[0], self.size[1], self.rank, self.num_iters, G) \
                    and x.numel() // (self.size[0] * self.size[1]) <= 65535:
                self._wide = True
                return sid
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
        sid = self._native_solver(x) if not args and not kwargs else None   # (extra solver arguments — wmu's weights — : composed)
        if sid is not None:
            fn = Fn.gnmf_decompose if self._wide else Fn.nmf_decompose
            return fn(x, self.init.u0, self.init.v0, self.num_iters,
                      min(max(self.num_grad_steps, 0), self.num_iters), sid, self.solver.eps)
        if x.dtype in (torch.bfloat16, torch.float16):
            # composed path in reduced precision: factorise in fp32 (eps = 1e-16 vanishes in fp16 and the
            # Gram matrices lose their low bits in bf16), factors are returned in fp32
            x = x.float()
        with self.context(0):
            u, v = self.init(x)
            u, v = u.to(x.dtype), v.to(x.dtype)
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
            fn = Fn.gnmf if self._wide else Fn.nmf
            return fn(x, self.init.u0, self.init.v0, self.num_iters,
                      min(max(self.num_grad_steps, 0), self.num_iters), sid, self.solver.eps)
        u, v = self.decompose(x)
        return self.reconstruct(u, v).to(x.dtype)


class NMF(MatrixFactorization):
    """Non-negative matrix factorization, X, U, V ≥ 0 (matrix_factorization.py:549-578)."""

    def __init__(self, size, rank: Optional[int] = None, compression: float = 10, num_iters: int = 5,
                 num_grad_steps: Optional[int] = None, init="uniform", solver="hals",
                 verbose: bool = False, **kwargs):
        super().__init__(size, rank=rank, compression=compression, num_iters=num_iters,
                         num_grad_steps=num_grad_steps, init=init, solver=solver, verbose=verbose,
                         **kwargs)
