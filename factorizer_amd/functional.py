"""Device ops of the hot path as autograd Functions over the C ABI (include/factorizer_hip.h).

Every Function here runs ONLY the native gfx950 kernels; CPU tensors never get here (modules
route them to `composed.py`).  Backward passes are hand-written kernels as well
(SURVEY.md Appendix A for NMF).
"""
from __future__ import annotations

import math
import os

import torch

from . import _native as N


def _dev_guard(t: torch.Tensor):
    return torch.cuda.device(t.device)


class KernelTimer:
    """Optional per-launch timing of the native kernels with HIP events recorded on the
    stream the kernels are launched on (torch's current stream).  bench.py installs one over
    its timed region to get the dominant kernel's average launch duration and algorithmic
    bytes for the roofline line.  Every record also carries the launch's voxel-column count
    (`cols`: batch x voxels of the finest tensor it touches — which stage of the U-shape it belongs to) and its
    matrix-core flops (`flops`: 2·M·K per output column of the GEMM-shaped layers; 0 for the byte-moving kernels)."""

    def __init__(self, only=None):
        self.records = []  # (name, algorithmic_bytes, start_event, end_event, cols, flops)
        self.only = None if only is None else set(only)  # time these keys only (events cost ~1 us of stream each)

    def launch(self, name, nbytes, fn, cols=0, flops=0):
        if self.only is not None and name not in self.only:
            return fn()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.records.append((name, nbytes, s, e, cols, flops))
        return out

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, nbytes, s, e, cols, flops in self.records:
            a = agg.setdefault(name, {"calls": 0, "ms": 0.0, "bytes": 0, "flops": 0, "cols": cols})
            a["calls"] += 1
            a["ms"] += s.elapsed_time(e)
            a["bytes"] += nbytes
            a["flops"] += flops
            a["cols"] = max(a["cols"], cols)
        return agg


_timer = None


def set_timer(t):
    global _timer
    _timer = t


def _timed(name, nbytes, fn, cols=0, flops=0):
    if _timer is None:
        return fn()
    return _timer.launch(name, nbytes, fn, cols, flops)


class Geometry:
    """Static shape contract of a shifted-window matricize (operations.py:299-355, 381-415):
    channels C = h·d, spatial S_i = G_i·p_i, windows with cyclic shifts s_w."""

    def __init__(self, channels, spatial, head_dim, patch, shifts):
        self.C = int(channels)
        self.spatial = tuple(int(s) for s in spatial)
        self.d = int(head_dim)
        self.patch = tuple(int(p) for p in patch)
        self.shifts = [tuple(int(v) for v in s) for s in shifts]
        if self.C % self.d:
            raise ValueError(f"channels {self.C} not divisible by head_dim {self.d}")
        for s, p in zip(self.spatial, self.patch):
            if s % p:
                raise ValueError(f"spatial size {self.spatial} not divisible by patch size {self.patch}")
        self.h = self.C // self.d
        self.grid = tuple(s // p for s, p in zip(self.spatial, self.patch))
        self.G = math.prod(self.grid)
        self.P = math.prod(self.patch)
        self.nshift = len(self.shifts)
        # the native kernels are 3-D; lower-D problems are padded with leading unit axes
        nd = len(self.spatial)
        if nd > 3:
            raise ValueError("at most 3 spatial dims")
        pad = 3 - nd
        self.s3 = (1,) * pad + self.spatial
        self.p3 = (1,) * pad + self.patch
        self.shifts3 = [(0,) * pad + s for s in self.shifts]
        self._carr = N.shifts_array(self.shifts3)

    def y_shape(self, B):
        return (self.nshift * B * self.h, self.G, self.d, self.P)


def _swm_fwd_raw(x, geo: Geometry, relu=False, div=1):
    B = x.shape[0]
    y = torch.empty(geo.y_shape(B), dtype=x.dtype, device=x.device)
    if x.dtype == torch.float32:
        es = 4
    elif x.dtype in (torch.bfloat16, torch.float16):
        es = 2
        if x.dtype == torch.float16 and (relu or div > 1):
            raise TypeError("SWMatricize with fused arithmetic: float32 or bfloat16 (float16 only moves bits)")
    else:
        raise TypeError(f"SWMatricize: unsupported dtype {x.dtype}")
    with _dev_guard(x):
        rc = _timed("swm_fwd", (1 + geo.nshift) * x.numel() * es, lambda: N.lib().fz_swm_fwd(
            x.data_ptr(), y.data_ptr(), B, geo.C, *geo.s3, geo.d, *geo.p3, geo.nshift, geo._carr, es,
            int(relu), int(div), N.stream_ptr(x)))
    N.check(rc, "fz_swm_fwd")
    return y


def _swm_inv_raw(y, geo: Geometry, average=True, gate=None):
    ad = N.act_dtype(y)   # float32, or bfloat16 storage (window sum / average in fp32, rounded once)
    if gate is not None and gate.dtype != y.dtype:
        raise TypeError("SWMatricize.inverse_forward: gate and y must have the same dtype")
    B = y.shape[0] // (geo.nshift * geo.h)
    x = torch.empty((B, geo.C, *geo.spatial), dtype=y.dtype, device=y.device)
    with _dev_guard(y):
        rc = _timed("swm_inv", (1 + geo.nshift) * x.numel() * y.element_size(), lambda: N.lib().fz_swm_inv(
            y.data_ptr(), x.data_ptr(), B, geo.C, *geo.s3, geo.d, *geo.p3, geo.nshift, geo._carr,
            int(average), N.ptr(gate), ad, N.stream_ptr(y)))
    N.check(rc, "fz_swm_inv")
    return x


class SWMForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, geo):
        ctx.geo = geo
        return _swm_fwd_raw(x.contiguous(), geo)

    @staticmethod
    def backward(ctx, gy):
        if gy.dtype not in (torch.float32, torch.bfloat16):   # fp16: sum the windows in fp32
            return _swm_inv_raw(gy.float().contiguous(), ctx.geo, average=False).to(gy.dtype), None
        return _swm_inv_raw(gy.contiguous(), ctx.geo, average=False), None


class SWMInverseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, geo):
        ctx.geo = geo
        return _swm_inv_raw(y.contiguous(), geo, average=True)

    @staticmethod
    def backward(ctx, gx):
        if gx.dtype == torch.float16:
            return _swm_fwd_raw(gx.float().contiguous(), ctx.geo, div=ctx.geo.nshift).to(gx.dtype), None
        return _swm_fwd_raw(gx.contiguous(), ctx.geo, div=ctx.geo.nshift), None


def swm_forward(x, geo):
    return SWMForwardFn.apply(x, geo)


def swm_inverse(y, geo):
    return SWMInverseFn.apply(y, geo)


# ---- batched NMF --------------------------------------------------------------------------
def nmf_supported(M, N_, R, T, G) -> bool:
    return bool(N.lib().fz_nmf_supported(int(M), int(N_), int(R), int(T), int(G)))


def _join_side_streams(t):
    """The standalone NMF kernels (csrc/nmf_r*.hip) are the only ones of the library that use scratch; a kernel with a
    private segment must not run beside the weight-gradient kernels of the second stream (results varied from run to run,
    tests/test_no_spills.py): the current stream waits for the side streams first."""
    from . import pointwise as _PW
    if _PW._SIDE:
        _PW.wait_wgrad_streams_all(t.device)


def _nmf_fwd_raw(x, u0, v0, T, solver, eps, want_uv=False):
    M, Nn = x.shape[-2:]
    R = u0.shape[1]
    nmat = x.numel() // (M * Nn)
    y = torch.empty_like(x)
    u = v = None
    if want_uv:
        # the factors are fp32 whatever the storage type of x (fp32 U/V/Gram/eps)
        u = torch.empty((*x.shape[:-2], M, R), dtype=torch.float32, device=x.device)
        v = torch.empty((*x.shape[:-2], Nn, R), dtype=torch.float32, device=x.device)
    ad = N.act_dtype(x)
    with _dev_guard(x):
        _join_side_streams(x)
        rc = _timed(f"nmf_fwd_{M}x{Nn}", 2 * x.numel() * x.element_size(), lambda: N.lib().fz_nmf_fwd(
            x.data_ptr(), u0.data_ptr(), v0.data_ptr(), y.data_ptr(), N.ptr(u), N.ptr(v), nmat, M, Nn, R, T,
            N.SOLVER_ID[solver], eps, ad, N.stream_ptr(x)))
    N.check(rc, "fz_nmf_fwd")
    return y, u, v


def _nmf_bwd_raw(x, u0, v0, gy, gu, gv, T, G, solver, eps):
    M, Nn = x.shape[-2:]
    R = u0.shape[1]
    nmat = x.numel() // (M * Nn)
    gx = torch.empty_like(x)
    ad = N.act_dtype(x)
    if gy is not None and gy.dtype != x.dtype:
        gy = gy.to(x.dtype)
    if gu is not None:
        gu, gv = gu.float().contiguous(), gv.float().contiguous()
    with _dev_guard(x):
        _join_side_streams(x)
        rc = _timed(f"nmf_bwd_{M}x{Nn}", 3 * x.numel() * x.element_size(), lambda: N.lib().fz_nmf_bwd(
            x.data_ptr(), u0.data_ptr(), v0.data_ptr(), N.ptr(gy), N.ptr(gu), N.ptr(gv), gx.data_ptr(), nmat,
            M, Nn, R, T, G, N.SOLVER_ID[solver], eps, ad, N.stream_ptr(x)))
    N.check(rc, "fz_nmf_bwd")
    return gx


class NMFFn(torch.autograd.Function):
    """y = u_T v_Tᵀ after T unrolled iterations (matrix_factorization.py:514-546)."""

    @staticmethod
    def forward(ctx, x, u0, v0, T, G, solver, eps):
        x = x.contiguous()
        u0 = u0.contiguous()
        v0 = v0.contiguous()
        y, _, _ = _nmf_fwd_raw(x, u0, v0, T, solver, eps)
        ctx.save_for_backward(x, u0, v0)
        ctx.cfg = (T, G, solver, eps)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, u0, v0 = ctx.saved_tensors
        T, G, solver, eps = ctx.cfg
        if G <= 0:
            return torch.zeros_like(x), None, None, None, None, None, None
        gx = _nmf_bwd_raw(x, u0, v0, gy.contiguous(), None, None, T, G, solver, eps)
        return gx, None, None, None, None, None, None


class NMFDecomposeFn(torch.autograd.Function):
    """(u_T, v_T) of the same iterations — NMF.decompose (matrix_factorization.py:514-530)."""

    @staticmethod
    def forward(ctx, x, u0, v0, T, G, solver, eps):
        x = x.contiguous()
        u0 = u0.contiguous()
        v0 = v0.contiguous()
        _, u, v = _nmf_fwd_raw(x, u0, v0, T, solver, eps, want_uv=True)
        ctx.save_for_backward(x, u0, v0)
        ctx.cfg = (T, G, solver, eps)
        return u, v

    @staticmethod
    def backward(ctx, gu, gv):
        x, u0, v0 = ctx.saved_tensors
        T, G, solver, eps = ctx.cfg
        if G <= 0:
            return torch.zeros_like(x), None, None, None, None, None, None
        gx = _nmf_bwd_raw(x, u0, v0, None, gu.contiguous(), gv.contiguous(), T, G, solver, eps)
        return gx, None, None, None, None, None, None


def nmf(x, u0, v0, T, G, solver, eps=1e-16):
    return NMFFn.apply(x, u0, v0, T, G, solver, eps)


def nmf_decompose(x, u0, v0, T, G, solver, eps=1e-16):
    return NMFDecomposeFn.apply(x, u0, v0, T, G, solver, eps)


# ---- split-N NMF: matrices too wide for one wavefront (csrc/nmf_global.hip) -------
def gnmf_supported(M, N_, R, T, G) -> bool:
    return bool(N.lib().fz_gnmf_supported(int(M), int(N_), int(R), int(T), int(G)))


def _gnmf_ws(x, nmat, M, Nn, R, T, backward):
    nb = N.lib().fz_gnmf_workspace_bytes(nmat, M, Nn, R, T, int(backward))
    if nb < 0:
        raise N.NativeError("fz_gnmf_workspace_bytes failed")
    return torch.empty(max(nb // 4, 1), dtype=torch.float32, device=x.device)


def _gnmf_fwd_raw(x, u0, v0, T, solver, eps, want_uv=False):
    M, Nn = x.shape[-2:]
    R = u0.shape[1]
    nmat = x.numel() // (M * Nn)
    y = torch.empty_like(x)
    u = v = None
    if want_uv:
        u = torch.empty((*x.shape[:-2], M, R), dtype=x.dtype, device=x.device)
        v = torch.empty((*x.shape[:-2], Nn, R), dtype=x.dtype, device=x.device)
    ws = _gnmf_ws(x, nmat, M, Nn, R, T, False)
    with _dev_guard(x):
        rc = _timed(f"gnmf_fwd_{M}x{Nn}", 2 * x.numel() * 4, lambda: N.lib().fz_gnmf_fwd(
            x.data_ptr(), u0.data_ptr(), v0.data_ptr(), y.data_ptr(), N.ptr(u), N.ptr(v), nmat, M, Nn, R, T,
            N.SOLVER_ID[solver], eps, ws.data_ptr(), N.stream_ptr(x)))
    N.check(rc, "fz_gnmf_fwd")
    return y, u, v


def _gnmf_bwd_raw(x, u0, v0, gy, gu, gv, T, G, solver, eps):
    M, Nn = x.shape[-2:]
    R = u0.shape[1]
    nmat = x.numel() // (M * Nn)
    gx = torch.empty_like(x)
    ws = _gnmf_ws(x, nmat, M, Nn, R, T, True)
    with _dev_guard(x):
        rc = _timed(f"gnmf_bwd_{M}x{Nn}", 3 * x.numel() * 4, lambda: N.lib().fz_gnmf_bwd(
            x.data_ptr(), u0.data_ptr(), v0.data_ptr(), N.ptr(gy), N.ptr(gu), N.ptr(gv), gx.data_ptr(), nmat, M, Nn,
            R, T, G, N.SOLVER_ID[solver], eps, ws.data_ptr(), N.stream_ptr(x)))
    N.check(rc, "fz_gnmf_bwd")
    return gx


class GNMFFn(torch.autograd.Function):
    """y = u_T v_Tᵀ for matrices whose columns are split over workgroups."""

    @staticmethod
    def forward(ctx, x, u0, v0, T, G, solver, eps):
        x, u0, v0 = x.contiguous(), u0.contiguous(), v0.contiguous()
        y, _, _ = _gnmf_fwd_raw(x, u0, v0, T, solver, eps)
        ctx.save_for_backward(x, u0, v0)
        ctx.cfg = (T, G, solver, eps)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, u0, v0 = ctx.saved_tensors
        T, G, solver, eps = ctx.cfg
        if G <= 0:
            return torch.zeros_like(x), None, None, None, None, None, None
        return _gnmf_bwd_raw(x, u0, v0, gy.contiguous(), None, None, T, G, solver, eps), None, None, None, None, None, None


class GNMFDecomposeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, u0, v0, T, G, solver, eps):
        x, u0, v0 = x.contiguous(), u0.contiguous(), v0.contiguous()
        _, u, v = _gnmf_fwd_raw(x, u0, v0, T, solver, eps, want_uv=True)
        ctx.save_for_backward(x, u0, v0)
        ctx.cfg = (T, G, solver, eps)
        return u, v

    @staticmethod
    def backward(ctx, gu, gv):
        x, u0, v0 = ctx.saved_tensors
        T, G, solver, eps = ctx.cfg
        if G <= 0:
            return torch.zeros_like(x), None, None, None, None, None, None
        gx = _gnmf_bwd_raw(x, u0, v0, None, gu.contiguous(), gv.contiguous(), T, G, solver, eps)
        return gx, None, None, None, None, None, None


def gnmf(x, u0, v0, T, G, solver, eps=1e-16):
    return GNMFFn.apply(x, u0, v0, T, G, solver, eps)


def gnmf_decompose(x, u0, v0, T, G, solver, eps=1e-16):
    return GNMFDecomposeFn.apply(x, u0, v0, T, G, solver, eps)


# ---- fused FactMixer core on channels-first tensors ------------------------------------------
def nmf_cf_supported(geo: Geometry, R, T, G) -> bool:
    """the 8x8x8 hot-shape kernels (csrc/nmf_cf.hip)"""
    if len(geo.spatial) != 3 or any(s[2] % 2 for s in geo.shifts):  # odd W-axis shifts, 1-D / 2-D: the generic-patch kernels
        return False
    return bool(N.lib().fz_nmf_cf_supported(geo.C, *geo.spatial, geo.d, *geo.patch, int(R), int(T), int(G)))


def nmf_pcf_supported(geo: Geometry, R, T, G) -> bool:
    """the generic-patch fused core (csrc/nmf_pcf.hip): head_dim 8, <= 256 voxels per patch, any shift.  1-D and 2-D tensors
    (operations.py:318-325 is N-D generic; the reference's own test models are 2-D) run as depth-1 (and height-1) volumes:
    the matricized order '(b h) (g..) d (p..)' of a (1, H, W) volume with (1, ph, pw) patches is that of the 2-D tensor."""
    if os.environ.get("FZ_NMF_PCF", "1") == "0":
        return False
    return bool(N.lib().fz_nmf_pcf_supported(geo.C, *geo.s3, geo.d, *geo.p3, int(R), int(T), int(G)))


def nmf_core_supported(geo: Geometry, R, T, G) -> bool:
    return nmf_cf_supported(geo, R, T, G) or nmf_pcf_supported(geo, R, T, G)


class FactCoreFn(torch.autograd.Function):
    """a = SWMatricize⁻¹(NMF(SWMatricize(t))) for t >= 0 already activated (factorizer.py:41-50)
    without materialising the matricized tensors (csrc/nmf_cf.hip).  `relu_gate`: the caller's
    t is relu(z) (so t >= 0 — the library relies on it: HALS rank 1 then runs its backward in the row space,
    csrc/nmf_gram.h); the backward returns the gradient w.r.t. z (gated by [t > 0])."""

    @staticmethod
    def forward(ctx, t, u0, v0, geo, T, G, solver, eps, relu_gate):
        t = t.contiguous()
        u0, v0 = u0.contiguous(), v0.contiguous()
        B = t.shape[0]
        out = torch.empty_like(t)
        R = u0.shape[1]
        es = t.element_size()
        nb = 2 * es * t.numel()
        ad = N.act_dtype(t)
        hot = nmf_cf_supported(geo, R, T, G)      # 8x8x8 patches: csrc/nmf_cf.hip; any other patch: csrc/nmf_pcf.hip
        with _dev_guard(t):
            for w, s in enumerate(geo.shifts3):
                arr = (N._i * 3)(*s)
                last = geo.nshift if w == geo.nshift - 1 else 1
                N.set_tile_order(w & 1)       # odd windows walk the tiles backwards (_native.py: set_tile_order)
                if hot:
                    rc = _timed(f"nmf_cf_fwd_{geo.C}x" + "x".join(str(v) for v in geo.spatial), nb + (es * t.numel() if w else 0), cols=t.numel() // geo.C, fn=lambda: N.lib().fz_nmf_cf_fwd(
                        t.data_ptr(), u0.data_ptr(), v0.data_ptr(), out.data_ptr(), B, geo.C, *geo.spatial, arr,
                        int(w > 0), last, R, T, N.SOLVER_ID[solver], eps, ad, N.stream_ptr(t)))
                else:
                    rc = _timed(f"nmf_pcf_fwd_{geo.C}x" + "x".join(str(v) for v in geo.spatial), nb + (es * t.numel() if w else 0), cols=t.numel() // geo.C, fn=lambda: N.lib().fz_nmf_pcf_fwd(
                        t.data_ptr(), u0.data_ptr(), v0.data_ptr(), out.data_ptr(), B, geo.C, *geo.s3, *geo.p3, arr,
                        int(w > 0), last, R, T, N.SOLVER_ID[solver], eps, ad, N.stream_ptr(t)))
                N.check(rc, "fz_nmf_cf_fwd" if hot else "fz_nmf_pcf_fwd")
            N.set_tile_order(0)
        ctx.save_for_backward(t, u0, v0)
        ctx.cfg = (geo, T, G, solver, eps, relu_gate)
        return out

    @staticmethod
    def backward(ctx, ga):
        t, u0, v0 = ctx.saved_tensors
        geo, T, G, solver, eps, relu_gate = ctx.cfg
        if G <= 0:
            return (torch.zeros_like(t),) + (None,) * 8
        ga = ga.contiguous()
        if ga.dtype != t.dtype:
            ga = ga.to(t.dtype)
        gt = torch.empty_like(t)
        B = t.shape[0]
        R = u0.shape[1]
        es = t.element_size()
        nb = 3 * es * t.numel()
        ad = N.act_dtype(t)
        hot = nmf_cf_supported(geo, R, T, G)
        with _dev_guard(t):
            for w, s in enumerate(geo.shifts3):
                arr = (N._i * 3)(*s)
                N.set_tile_order(w & 1)
                if hot:
                    rc = _timed(f"nmf_cf_bwd_{geo.C}x" + "x".join(str(v) for v in geo.spatial), nb + (es * t.numel() if w else 0), cols=t.numel() // geo.C, fn=lambda: N.lib().fz_nmf_cf_bwd(
                        t.data_ptr(), u0.data_ptr(), v0.data_ptr(), ga.data_ptr(), gt.data_ptr(), B, geo.C,
                        *geo.spatial, arr, int(w > 0), geo.nshift, int(relu_gate), R, T, G, N.SOLVER_ID[solver], eps,
                        ad, N.stream_ptr(t)))
                elif w > 0 and N.lib().fz_nmf_pcf_bwd_prefers_separate(*geo.p3, ad):
                    # the window's gradient into its own buffer, then one coalesced add (include/factorizer_hip.h)
                    tmp = torch.empty_like(gt)
                    rc = _timed(f"nmf_pcf_bwd_{geo.C}x" + "x".join(str(v) for v in geo.spatial), nb, cols=t.numel() // geo.C, fn=lambda: N.lib().fz_nmf_pcf_bwd(
                        t.data_ptr(), u0.data_ptr(), v0.data_ptr(), ga.data_ptr(), tmp.data_ptr(), B, geo.C,
                        *geo.s3, *geo.p3, arr, 0, geo.nshift, int(relu_gate), R, T, G, N.SOLVER_ID[solver], eps,
                        ad, N.stream_ptr(t)))
                    N.check(rc, "fz_nmf_pcf_bwd")
                    rc = _timed(f"window_add_{geo.C}x" + "x".join(str(v) for v in geo.spatial), nb, cols=t.numel() // geo.C, fn=lambda: N.lib().fz_act_add(
                        gt.data_ptr(), tmp.data_ptr(), gt.numel(), ad, N.stream_ptr(t)))
                    N.check(rc, "fz_act_add")
                    del tmp
                else:
                    rc = _timed(f"nmf_pcf_bwd_{geo.C}x" + "x".join(str(v) for v in geo.spatial), nb + (es * t.numel() if w else 0), cols=t.numel() // geo.C, fn=lambda: N.lib().fz_nmf_pcf_bwd(
                        t.data_ptr(), u0.data_ptr(), v0.data_ptr(), ga.data_ptr(), gt.data_ptr(), B, geo.C,
                        *geo.s3, *geo.p3, arr, int(w > 0), geo.nshift, int(relu_gate), R, T, G, N.SOLVER_ID[solver], eps,
                        ad, N.stream_ptr(t)))
                N.check(rc, "fz_nmf_cf_bwd" if hot else "fz_nmf_pcf_bwd")
            N.set_tile_order(0)
        return (gt,) + (None,) * 8


# ---- grouped "same" cross-correlation of the Deconver family (csrc/deconv.hip) --------------
def _k3(w):
    """kernel dims of a (Bw, G, Co, Ci, *k) filter bank as (kd, kh, kw) with 2-D layers as depth 1"""
    k = tuple(w.shape[4:])
    return (1,) * (3 - len(k)) + k


def gcorr_supported(inp, w) -> bool:
    if not (inp.is_cuda and inp.numel() and inp.dtype == torch.float32 and w.dtype == torch.float32):
        return False
    if inp.dim() not in (4, 5) or w.dim() != inp.dim() + 2 or w.shape[0] not in (1, inp.shape[0]):
        return False
    if not N.lib().fz_gcorr_supported(int(w.shape[3]), int(w.shape[2]), *_k3(w)):
        return False
    if torch.is_grad_enabled() and inp.requires_grad:      # the input gradient is the adjoint correlation: Co ↔ Ci
        return bool(N.lib().fz_gcorr_supported(int(w.shape[2]), int(w.shape[3]), *_k3(w)))
    return True


def _gcorr_raw(inp, w, add_eps=0.0, mul_a=None, mul_b=None):
    B = inp.shape[0]
    Bw, G, Co, Ci = w.shape[:4]
    sp = tuple(inp.shape[2:])
    D, H, W = (1,) * (3 - len(sp)) + sp
    out = torch.empty((B, G * Co, *sp), dtype=inp.dtype, device=inp.device)
    kd, kh, kw = _k3(w)
    with _dev_guard(inp):
        rc = _timed(f"gcorr_{Ci}->{Co}_k{kd}{kh}{kw}", 4 * (inp.numel() + out.numel() * (3 if mul_a is not None else 1)),
                    lambda: N.lib().fz_gcorr(inp.data_ptr(), w.data_ptr(), out.data_ptr(), N.ptr(mul_a), N.ptr(mul_b), B, G,
                                             Ci, Co, D, H, W, kd, kh, kw, int(Bw != 1), float(add_eps),
                                             N.stream_ptr(inp)))
    N.check(rc, "fz_gcorr")
    return out


def adjoint_filters(w):
    """(Bw, G, Co, Ci, *k) → (Bw, G, Ci, Co, *k), spatially flipped: correlation with it is the adjoint operator"""
    return torch.flip(w.transpose(2, 3), dims=tuple(range(4, w.ndim))).contiguous()


def _gcorr_wgrad_raw(inp, gout, w_shape):
    """gw (Bw, G, Co, Ci, *k) of out = corr(inp, w): fz_gcorr_wgrad (deterministic two-stage reduction)"""
    B = inp.shape[0]
    Bw, G, Co, Ci = w_shape[:4]
    sp = tuple(inp.shape[2:])
    D, H, W = (1,) * (3 - len(sp)) + sp
    ks = tuple(w_shape[4:])
    kd, kh, kw = (1,) * (3 - len(ks)) + ks
    gw = torch.empty(tuple(w_shape), dtype=torch.float32, device=inp.device)
    nb = N.lib().fz_gcorr_wgrad_workspace_bytes(B, G, Ci, Co, D, H, W, kd, kh, kw)
    ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=inp.device)
    with _dev_guard(inp):
        rc = _timed(f"gcorr_wgrad_{Ci}->{Co}_k{kd}{kh}{kw}", 4 * (inp.numel() + gout.numel()),
                    lambda: N.lib().fz_gcorr_wgrad(inp.data_ptr(), gout.data_ptr(), gw.data_ptr(), ws.data_ptr(), B, G, Ci, Co,
                                                   D, H, W, kd, kh, kw, int(Bw != 1), N.stream_ptr(inp)))
    N.check(rc, "fz_gcorr_wgrad")
    return gw


class GCorrFn(torch.autograd.Function):
    """out = corr(inp, w) + add_eps: native forward, input gradient (the adjoint correlation) and filter gradient
    (the lag-correlation of inp with the output gradient, fz_gcorr_wgrad)."""

    @staticmethod
    def forward(ctx, inp, w, add_eps):
        inp, w = inp.contiguous(), w.contiguous()
        ctx.save_for_backward(inp, w)
        return _gcorr_raw(inp, w, add_eps)

    @staticmethod
    def backward(ctx, gout):
        inp, w = ctx.saved_tensors
        gout = gout.contiguous()
        ginp = gw = None
        if ctx.needs_input_grad[0]:
            ginp = _gcorr_raw(gout, adjoint_filters(w))
        if ctx.needs_input_grad[1]:
            gw = _gcorr_wgrad_raw(inp, gout, w.shape)
        return ginp, gw, None


class LagCorrFn(torch.autograd.Function):
    """L[b, g, c, k, τ] = Σ_v x[b, g·C + c, v] · s[b, g·K + k, v + τ − p] — the lag correlations of the filter update
    (deconvolution.py:43-50 `sconv`, :150-156 `update_h`).  The same reduction as the filter gradient of the grouped
    correlation (fz_gcorr_wgrad); its own gradients are two grouped correlations with gL as per-sample filters."""

    @staticmethod
    def forward(ctx, s, x, G, ksize):
        s, x = s.contiguous(), x.contiguous()
        ctx.save_for_backward(s, x)
        B = s.shape[0]
        return _gcorr_wgrad_raw(s, x, (B, G, x.shape[1] // G, s.shape[1] // G, *ksize))

    @staticmethod
    def backward(ctx, gL):
        s, x = ctx.saved_tensors
        gL = gL.contiguous()
        gs = gx = None
        if ctx.needs_input_grad[0]:
            gs = _gcorr_raw(x, adjoint_filters(gL))
        if ctx.needs_input_grad[1]:
            gx = _gcorr_raw(s, gL)
        return gs, gx, None, None


def lag_corr_supported(s, x, G, ksize) -> bool:
    if not (s.is_cuda and s.numel() and s.dtype == torch.float32 and x.dtype == torch.float32 and s.dim() in (4, 5)):
        return False
    K, C = s.shape[1] // G, x.shape[1] // G
    k3 = (1,) * (3 - len(ksize)) + tuple(int(k) for k in ksize)
    return bool(N.lib().fz_gcorr_supported(K, C, *k3)) and bool(N.lib().fz_gcorr_supported(C, K, *k3))


def lag_corr(s, x, G, ksize):
    return LagCorrFn.apply(s, x, G, tuple(int(k) for k in ksize))


def gcorr(inp, w, add_eps=0.0):
    return GCorrFn.apply(inp, w, add_eps)


def gcorr_mu_update(s, num, r, wT, eps):
    """s ∘ num / (corr(r, wT) + eps) in one launch — for iterations that carry no gradient (inference, or the
    leading `num_iters − num_grad_iters` iterations of deconvolution.py:158-166)."""
    return _gcorr_raw(r.contiguous(), wT.contiguous(), eps, s.contiguous(), num.contiguous())
