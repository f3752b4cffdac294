"""Dense per-voxel layers on channels-first tensors: LayerNorm over C, 1×1 GEMMs ("Linear"),
the MLP, k2s2 (transposed) convolutions and the k3 stem — device paths over the fp32-MFMA GEMM
family of libfactorizer_hip (csrc/gemm.hip, wgrad.hip, ln.hip), CPU paths composed from ATen.

Each autograd Function below is one fused layer of the reference block:
  LNLinearFn      LayerNorm → Linear (+bias) → [ReLU]     norm.py:29-34 + linear.py:53-58 (+ factorizer.py:44)
  ActLinearResFn  res + Linear([GELU](z)) + bias           mlp.py:54-60 / factorizer.py:53,75-76
  LinearFn        plain Linear                              linear.py:53-58
  CatLinearFn     Linear(cat([x1, x2], 1)) without the cat  unet.py:128 + factorizer.py:116
  ConvK2S2Fn / TConvK2S2Fn / ConvK3Fn / ConvK1            unet.py:53,123,231,253
"""
from __future__ import annotations

import ctypes
import os
import threading as _threading
import weakref

import torch
import torch.nn.functional as F

from . import _native as N
from . import gradbuf as _GB
from . import functional as Fn

ACT = {"none": 0, "relu": 1, "gelu": 2}
LOAD_PLAIN, LOAD_S2D, LOAD_K3 = 0, 1, 2
EPI_PLAIN, EPI_D2S = 0, 1


def _p(t):
    return None if t is None else t.data_ptr()


def _vox(x):
    v = 1
    for s in x.shape[2:]:
        v *= s
    return v


# ---- raw launches ---------------------------------------------------------------------------
def _gemm(xs, w, y, *, B, Cin, Vin, M, K, Ncol, w_t=False, ldw=None, bias=None, ln=None, stats_out=None,
          bact=0, bmul=None, bmul_kind=0, eact=0, res=None, emul=None, emul_kind=0, src_mode=0, c0=0,
          loader=LOAD_PLAIN, epilogue=EPI_PLAIN, Di=0, Hi=0, Wi=0, Ho=0, Wo=0, name="gemm", tune=0):
    d = N.GemmDesc()
    d.products = N.products()
    d.tune = tune
    for i in range(4):
        d.x[i] = _p(xs[i]) if i < len(xs) else None
    d.nsrc, d.src_mode, d.c0, d.Cin, d.Vin = len(xs), src_mode, c0, Cin, Vin
    d.Di, d.Hi, d.Wi = Di, Hi, Wi
    d.w, d.w_t, d.ldw, d.M, d.K = _p(w), int(w_t), (ldw if ldw is not None else K), M, K
    d.bias = _p(bias)
    if ln is not None:
        d.ln, d.ln_g, d.ln_b, d.ln_eps = 1, _p(ln[0]), _p(ln[1]), float(ln[2])
    d.stats_out = _p(stats_out)
    d.bact, d.bmul, d.bmul_kind = bact, _p(bmul), bmul_kind
    d.eact, d.res, d.emul, d.emul_kind = eact, _p(res), _p(emul), emul_kind
    d.y, d.Ncol, d.Ho, d.Wo, d.B = _p(y), Ncol, Ho, Wo, B
    d.loader, d.epilogue = loader, epilogue
    x0 = xs[0]
    d.act_dtype = N.act_dtype(x0)
    nbytes = x0.element_size() * (sum(t.numel() for t in xs) + y.numel() + (res.numel() if res is not None else 0))
    with torch.cuda.device(x0.device):
        rc = Fn._timed(f"{name}_{Cin}->{M}", nbytes,
                       lambda: N.lib().fz_gemm(ctypes.byref(d), N.stream_ptr(x0)),
                       cols=B * max(Vin, Ncol * (8 if epilogue == EPI_D2S else 1)), flops=2 * B * Ncol * M * K)
    N.check(rc, "fz_gemm")
    return y


def _wgrad_desc(p, qs, gw, *, B, M, Cin, K, Vq, Ncols, gbias=None, pmul=None, pmul_kind=0, src_mode=0, c0=0,
                stats=None, qact=0, ln=None, loader=0, D=0, H=0, W=0, Ho=0, Wo=0, accumulate=False, name="wgrad"):
    """(descriptor, workspace, timer key, algorithmic bytes, columns, flops) of one weight-gradient problem"""
    d = N.WgradDesc()
    d.products = N.products()
    d.p, d.M, d.pmul, d.pmul_kind = _p(p), M, _p(pmul), pmul_kind
    for i in range(4):
        d.q[i] = _p(qs[i]) if i < len(qs) else None
    d.nsrc, d.src_mode, d.c0, d.Cin, d.K, d.Vq = len(qs), src_mode, c0, Cin, K, Vq
    d.D, d.H, d.W, d.N, d.Ho, d.Wo = D, H, W, Ncols, Ho, Wo
    d.stats, d.qact = _p(stats), qact
    if ln is not None:
        d.ln_g, d.ln_b = _p(ln[0]), _p(ln[1])
    d.gw, d.gbias, d.accumulate, d.B, d.loader = _p(gw), _p(gbias), int(accumulate), B, loader
    d.act_dtype = N.act_dtype(p)
    nb = N.lib().fz_wgrad_workspace_bytes(ctypes.byref(d))
    if nb < 0:
        raise N.NativeError("fz_wgrad_workspace_bytes failed")
    ws = torch.empty(max(nb // 4, 1), dtype=torch.float32, device=p.device)
    nbytes = p.element_size() * (p.numel() + sum(t.numel() for t in qs))
    return d, ws, f"{name}_{M}x{K}", nbytes, B * max(Vq, Ncols), 2 * B * Ncols * M * K


def _wgrad(p, qs, gw, **kw):
    d, ws, key, nbytes, cols, flops = _wgrad_desc(p, qs, gw, **kw)
    with torch.cuda.device(p.device), _finish_scope(ws, gw, kw.get("gbias")):
        rc = Fn._timed(key, nbytes, lambda: N.lib().fz_wgrad(ctypes.byref(d), ws.data_ptr(), N.stream_ptr(p)),
                       cols=cols, flops=flops)
    N.check(rc, "fz_wgrad")
    return gw


_WGRAD_GROUP = os.environ.get("FZ_WGRAD_GROUP", "1") != "0"   # diagnostics: 0 = one launch per weight gradient


def _wgrad_group(problems, name):
    """`problems`: up to four (p, qs, gw, kwargs) weight-gradient problems of one layer block, launched as ONE partial-sum
    grid (fz_wgrad_group; the results are those of the single launches)."""
    plans = [_wgrad_desc(p, qs, gw, **kw) for p, qs, gw, kw in problems]
    n = len(plans)
    descs = (ctypes.POINTER(N.WgradDesc) * n)(*[ctypes.pointer(pl[0]) for pl in plans])
    wss = (ctypes.c_void_p * n)(*[pl[1].data_ptr() for pl in plans])
    t0 = problems[0][0]
    held = [pl[1] for pl in plans] + [pr[2] for pr in problems] + [pr[3].get("gbias") for pr in problems]
    with torch.cuda.device(t0.device), _finish_scope(*held):
        rc = Fn._timed(name, sum(pl[3] for pl in plans),
                       lambda: N.lib().fz_wgrad_group(descs, wss, n, N.stream_ptr(t0)),
                       cols=max(pl[4] for pl in plans), flops=sum(pl[5] for pl in plans))
    N.check(rc, "fz_wgrad_group")


def _ln_backward(gl, x, stats, ln_w, gadd=None):
    """LayerNorm backward: (gx, gγ, gβ) in one pass (csrc/ln.hip; the affine gradients are
    accumulated per workgroup and reduced in a fixed order)."""
    B, C = x.shape[:2]
    V = _vox(x)
    gx = torch.empty_like(x)
    gpar = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    ws = torch.empty(max(N.lib().fz_ln_bwd_workspace_bytes2(B, C, V) // 4, 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _finish_scope(ws, gpar):
        rc = Fn._timed(f"ln_bwd_{C}", 3 * x.element_size() * x.numel(), lambda: N.lib().fz_ln_bwd(
            gl.data_ptr(), x.data_ptr(), stats.data_ptr(), ln_w.data_ptr(), _p(gadd), gx.data_ptr(), _p(gpar), _p(ws),
            B, C, V, N.act_dtype(x), N.stream_ptr(x)), cols=B * V)
    N.check(rc, "fz_ln_bwd")
    return gx, gpar[:C], gpar[C:]


def _dgrad_lnbwd(gz, w2, x, stats, ln_w, gadd):
    """gx = LayerNormBackward(W2ᵀ·gz; x, stats, γ) + gadd and (gγ, gβ), in ONE kernel when the
    LayerNorm width is 32, or 64 with a 64-channel gradient (csrc/gemm.hip EPI_LNBWD: gl never leaves the accumulators)."""
    B, C = x.shape[:2]
    V = _vox(x)
    Mz = gz.shape[1]
    fused64 = C == 64 and Mz == 64 and gadd is not None and V % 4 == 0 and os.environ.get("FZ_LNBWD64", "1") != "0"
    if not fused64 and (C != 32 or Mz > 64 or Mz % 2):
        gl = torch.empty_like(x)
        _gemm([gz], w2, gl, B=B, Cin=Mz, Vin=V, M=C, K=Mz, Ncol=V, w_t=True, ldw=C, name="linear_dgrad")
        return _ln_backward(gl, x, stats, ln_w, gadd=gadd)
    gx = torch.empty_like(x)
    d = N.GemmDesc()
    d.products = N.products()
    d.x[0] = gz.data_ptr()
    d.nsrc, d.src_mode, d.c0, d.Cin, d.Vin = 1, 0, 0, Mz, V
    d.w, d.w_t, d.ldw, d.M, d.K = w2.data_ptr(), 1, C, C, Mz
    d.y, d.Ncol, d.B = gx.data_ptr(), V, B
    d.loader, d.epilogue = LOAD_PLAIN, 2
    d.lnb_x, d.lnb_stats, d.lnb_g, d.lnb_gadd = x.data_ptr(), stats.data_ptr(), ln_w.data_ptr(), _p(gadd)
    d.act_dtype = N.act_dtype(x)
    rows = N.lib().fz_gemm_lnbwd_partials(ctypes.byref(d))
    part = torch.empty((rows, 2 * C), dtype=torch.float32, device=x.device)
    d.lnb_part = part.data_ptr()
    gpar = torch.empty(2 * C, dtype=torch.float32, device=x.device)
    tmp = torch.empty((64, 2 * C), dtype=torch.float32, device=x.device)
    nbytes = x.element_size() * (gz.numel() + 2 * x.numel() + (gadd.numel() if gadd is not None else 0))
    with torch.cuda.device(x.device):
        rc = Fn._timed(f"dgrad_lnbwd_{Mz}->{C}", nbytes, lambda: N.lib().fz_gemm(ctypes.byref(d), N.stream_ptr(x)),
                       cols=B * V, flops=2 * B * V * C * Mz)
        N.check(rc, "fz_gemm")
        with _finish_scope(part, gpar, tmp):
            rc = N.lib().fz_reduce_rows(part.data_ptr(), rows, 2 * C, gpar.data_ptr(), tmp.data_ptr(), N.stream_ptr(x))
        N.check(rc, "fz_reduce_rows")
    return gx, gpar[:C], gpar[C:]


def _dw_fused_ok(C, V):
    """input gradient + weight gradient of a 32 -> 32 layer in one pass (csrc/gemm.hip gemm_dw_kernel)"""
    return C == 32 and V % 4 == 0 and V <= (1 << 27) and os.environ.get("FZ_DW_FUSED", "1") != "0"


def _gemm_dw(g, w2, q, ln=None, stats=None, gadd=None, want_bias=False, name="dgrad_wgrad", gw_out=None, gb_out=None):
    """y = Wᵀ g [→ LayerNorm backward + gadd], gw = Σ_v g ⊗ in, gb = Σ_v g — fz_gemm_dw.  ln = (γ, β) or None.
    Returns (y, gw, gb or None, gγ, gβ) (gγ, gβ None without ln)."""
    B, C = q.shape[:2]
    V = _vox(q)
    dev = q.device
    y = torch.empty_like(q)
    # gw_out: a (C, C) column block of a wider weight gradient (row stride = its width), written in place
    gw = _GB.out_like(w2, (C, C), torch.float32) if gw_out is None else gw_out
    gb = (gb_out if gb_out is not None else torch.empty(C, dtype=torch.float32, device=dev)) if want_bias else None
    wpart = torch.empty(N.lib().fz_gemm_dw_workspace_bytes(B, V) // 4, dtype=torch.float32, device=dev)
    d = N.GemmDwDesc()
    d.g, d.q, d.w, d.y = g.data_ptr(), q.data_ptr(), w2.data_ptr(), y.data_ptr()
    d.wpart, d.gw, d.gb = wpart.data_ptr(), gw.data_ptr(), _p(gb)
    d.B, d.C, d.V, d.act_dtype = B, C, V, N.act_dtype(q)
    d.ldgw = gw.stride(0)
    d.ldw = w2.stride(0)          # (a 32-column block of a wider weight is read in place)
    gpar = None
    if ln is not None:
        gpar = torch.empty(64, dtype=torch.float32, device=dev)
        d.ln, d.stats, d.ln_g, d.ln_b, d.gadd, d.gln = 1, stats.data_ptr(), ln[0].data_ptr(), ln[1].data_ptr(), _p(gadd), gpar.data_ptr()
    nbytes = q.element_size() * (3 * q.numel() + (gadd.numel() if gadd is not None else 0))
    with torch.cuda.device(dev), _finish_scope(wpart, gw, gb, gpar):
        rc = Fn._timed(f"{name}_{C}", nbytes, lambda: N.lib().fz_gemm_dw(ctypes.byref(d), N.stream_ptr(q)),
                       cols=B * V, flops=4 * B * V * C * C)
        N.check(rc, "fz_gemm_dw")
    if ln is not None:
        return y, gw, gb, gpar[:32], gpar[32:]
    return y, gw, gb, None, None


def _mlp_chain_ok(C, Hd, V):
    return os.environ.get("FZ_MLP_CHAIN", "1") != "0" and bool(N.lib().fz_mlp_supported(C, Hd, V))


def _mlp_fwd_chain(x1, ln_w, ln_b, eps, w12, b1, w22, b2):
    """x2 = x1 + fc2(gelu(fc1(LN(x1)))) in ONE kernel (csrc/gemm.hip gemm_chain_kernel): the hidden
    tensor goes from the accumulators of the first GEMM into the second; z1 (pre-activation) and
    the LayerNorm statistics are written once for the backward."""
    B, C = x1.shape[:2]
    V = _vox(x1)
    Hd = w12.shape[0]
    z1 = torch.empty((B, Hd, *x1.shape[2:]), dtype=x1.dtype, device=x1.device)
    st = torch.empty((B, 2, V), dtype=torch.float32, device=x1.device)
    x2 = torch.empty_like(x1)
    d = N.MlpDesc()
    d.products = N.products()
    d.mode, d.inp, d.w1, d.w2, d.b1, d.b2 = 0, x1.data_ptr(), w12.data_ptr(), w22.data_ptr(), _p(b1), _p(b2)
    d.ln_g, d.ln_b, d.ln_eps = ln_w.data_ptr(), ln_b.data_ptr(), float(eps)
    d.stats, d.z1, d.out = st.data_ptr(), z1.data_ptr(), x2.data_ptr()
    d.B, d.C, d.H, d.V = B, C, Hd, V
    d.act_dtype = N.act_dtype(x1)
    with torch.cuda.device(x1.device):
        rc = Fn._timed(f"mlp_chain_fwd_{C}", x1.element_size() * (2 * x1.numel() + z1.numel()),
                       lambda: N.lib().fz_mlp_chain(ctypes.byref(d), N.stream_ptr(x1)), cols=B * V, flops=4 * B * V * C * Hd)
    N.check(rc, "fz_mlp_chain")
    return x2, z1, st


_OUTPROJ_MLP = os.environ.get("FZ_OUTPROJ_MLP", "1") != "0"   # diagnostics: 0 = out_proj and the MLP chain as two launches


def _outproj_mlp_ok(C, Hd, V):
    return _OUTPROJ_MLP and _mlp_chain_ok(C, Hd, V) and bool(N.lib().fz_mlp_pre_supported(C, Hd, V, N.products()))


def _outproj_mlp_fwd_chain(a, wout2, bout, x, ln_w, ln_b, eps, w12, b1, w22, b2, head=None):
    """x1 = x + out_proj(a) (factorizer.py:53,75) and x2 = x1 + fc2(gelu(fc1(LN(x1)))) (factorizer.py:76; mlp.py:54-63) in ONE
    launch: the out-projection runs on the accumulators in front of the chained MLP GEMMs, x1 is written once (the backward
    needs it) and never read back — 6 instead of 7 tensor passes.  Returns (x1, x2, z1, stats), plus the logits when `head` =
    (weight (M, C), bias or None) of the network's head Linear(C -> M <= 4) is given: it is applied to x2 in the same launch."""
    B, C = a.shape[:2]
    V = _vox(a)
    Hd = w12.shape[0]
    z1 = torch.empty((B, Hd, *a.shape[2:]), dtype=a.dtype, device=a.device)
    st = torch.empty((B, 2, V), dtype=torch.float32, device=a.device)
    x1 = torch.empty_like(a)
    x2 = torch.empty_like(a)
    d = N.MlpDesc()
    d.products = N.products()
    d.mode, d.inp, d.w1, d.w2, d.b1, d.b2 = 0, None, w12.data_ptr(), w22.data_ptr(), _p(b1), _p(b2)
    d.ln_g, d.ln_b, d.ln_eps = ln_w.data_ptr(), ln_b.data_ptr(), float(eps)
    d.stats, d.z1, d.out = st.data_ptr(), z1.data_ptr(), x2.data_ptr()
    d.B, d.C, d.H, d.V = B, C, Hd, V
    d.act_dtype = N.act_dtype(a)
    d.pre_in, d.pre_w, d.pre_b, d.pre_res, d.pre_out = a.data_ptr(), wout2.data_ptr(), _p(bout), x.data_ptr(), x1.data_ptr()
    logits = None
    if head is not None:
        hw, hb = head
        Mh = hw.shape[0]
        logits = torch.empty((B, Mh, *a.shape[2:]), dtype=a.dtype, device=a.device)
        d.post_w, d.post_b, d.post_out, d.post_m = hw.data_ptr(), _p(hb), logits.data_ptr(), Mh
    with torch.cuda.device(a.device):
        rc = Fn._timed(f"outproj_mlp_chain_fwd_{C}", a.element_size() * (4 * a.numel() + z1.numel() + (logits.numel() if head is not None else 0)),
                       lambda: N.lib().fz_mlp_chain(ctypes.byref(d), N.stream_ptr(a)), cols=B * V, flops=4 * B * V * C * Hd + 2 * B * V * C * C)
    N.check(rc, "fz_mlp_chain")
    if head is not None:
        return x1, x2, z1, st, logits
    return x1, x2, z1, st


def _mlp_bwd_chain(g2, z1, w12, w22, x1, st, ln_w):
    """(gz1, gx1, gγ, gβ): gz1 = (W2ᵀ g2) ∘ gelu'(z1); gx1 = LNbwd(W1ᵀ gz1) + g2, one kernel."""
    B, C = x1.shape[:2]
    V = _vox(x1)
    Hd = w12.shape[0]
    gz1 = torch.empty_like(z1)
    gx1 = torch.empty_like(x1)
    rows = N.lib().fz_mlp_partials(B, V)
    part = torch.empty((rows, 2 * C), dtype=torch.float32, device=x1.device)
    gpar = torch.empty(2 * C, dtype=torch.float32, device=x1.device)
    tmp = torch.empty((64, 2 * C), dtype=torch.float32, device=x1.device)
    d = N.MlpDesc()
    d.products = N.products()
    d.mode, d.inp, d.w1, d.w2 = 1, g2.data_ptr(), w12.data_ptr(), w22.data_ptr()
    d.ln_g, d.stats, d.z1, d.gz1, d.x1 = ln_w.data_ptr(), st.data_ptr(), z1.data_ptr(), gz1.data_ptr(), x1.data_ptr()
    d.out, d.part = gx1.data_ptr(), part.data_ptr()
    d.B, d.C, d.H, d.V = B, C, Hd, V
    d.act_dtype = N.act_dtype(x1)
    with torch.cuda.device(x1.device):
        rc = Fn._timed(f"mlp_chain_bwd_{C}", x1.element_size() * (3 * x1.numel() + 2 * z1.numel()),
                       lambda: N.lib().fz_mlp_chain(ctypes.byref(d), N.stream_ptr(x1)), cols=B * V, flops=4 * B * V * C * Hd)
        N.check(rc, "fz_mlp_chain")
        with _finish_scope(part, gpar, tmp):
            rc = N.lib().fz_reduce_rows(part.data_ptr(), rows, 2 * C, gpar.data_ptr(), tmp.data_ptr(), N.stream_ptr(x1))
        N.check(rc, "fz_reduce_rows")
    return gz1, gx1, gpar[:C], gpar[C:]


def _mlp_wgrad_fused_ok(C, Hd, V):
    """the chain backward that also forms dW1, db1, dW2, db2 (csrc/gemm.hip gemm_chain_bwd_wg_kernel): C = 32, hidden 64
    (one launch) or 128 (one launch per 64-row half of the hidden tensor)"""
    return _mlp_chain_ok(C, Hd, V) and C == 32 and Hd in (64, 128) and os.environ.get("FZ_MLP_FUSED_WGRAD", "1") != "0"


def _mlp_bwd_chain_wgrad(g2, z1, w12, w22, x1, st, ln_w, ln_b):
    """(gx1, gγ, gβ, gw1, gb1, gw2, gb2) in ONE pass over (g2, z1, x1): gz1 never reaches HBM."""
    B, C = x1.shape[:2]
    V = _vox(x1)
    Hd = w12.shape[0]
    dev = x1.device
    gx1 = torch.empty_like(x1)
    gpar = torch.empty(64, dtype=torch.float32, device=dev)
    wpart = torch.empty(N.lib().fz_mlp_wgrad_workspace_bytes(B, V) // 4, dtype=torch.float32, device=dev)
    gw1 = _GB.out_like(w12, (Hd, C), torch.float32)
    gw2 = _GB.out_like(w22, (C, Hd), torch.float32)
    gb1 = torch.empty(Hd, dtype=torch.float32, device=dev)
    gb2 = torch.empty(C, dtype=torch.float32, device=dev)
    d = N.MlpDesc()
    d.products = N.products()
    d.mode, d.inp, d.w1, d.w2 = 2, g2.data_ptr(), w12.data_ptr(), w22.data_ptr()
    d.ln_g, d.ln_b, d.stats, d.z1, d.x1 = ln_w.data_ptr(), ln_b.data_ptr(), st.data_ptr(), z1.data_ptr(), x1.data_ptr()
    d.out, d.gln, d.wpart = gx1.data_ptr(), gpar.data_ptr(), wpart.data_ptr()
    d.gw1, d.gb1, d.gw2, d.gb2 = gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr()
    glp = torch.empty(x1.shape, dtype=torch.float32, device=dev) if Hd == 128 else None
    d.glp = _p(glp)
    d.B, d.C, d.H, d.V = B, C, Hd, V
    d.act_dtype = N.act_dtype(x1)
    # algorithmic bytes: H = 64: g2, z1, x1 in, gx1 out; H = 128: both halves read g2 and x1, + the fp32 partial out and in
    nbytes = x1.element_size() * (3 * x1.numel() + z1.numel()) + (0 if Hd == 64 else x1.element_size() * 2 * x1.numel() + 8 * x1.numel())
    with torch.cuda.device(dev), _finish_scope(wpart, gpar, gw1, gb1, gw2, gb2):
        rc = Fn._timed(f"mlp_chain_bwd_wgrad_{C}" + ("" if Hd == 64 else f"x{Hd}"), nbytes,
                       lambda: N.lib().fz_mlp_chain(ctypes.byref(d), N.stream_ptr(x1)), cols=B * V, flops=8 * B * V * C * Hd)
        N.check(rc, "fz_mlp_chain")
    return gx1, gpar[:32], gpar[32:], gw1, gb1, gw2, gb2


_SIDE = {}


def _side_stream(dev, ncols):
    """Second HIP stream for the weight-gradient kernels of the small, deep stages — OFF by default since round 3.
    Two of this library's kernels co-resident on the chip from different queues gave run-to-run different results in
    one NMF matrix of a bf16 step (every kernel alone is bitwise reproducible, no buffer is overrun, LDS is not shared:
    profiles/r03_two_stream_interaction.md has the whole hunt); the overlap was worth 0.2-0.3 ms of a 19.3 ms step.
    FZ_SIDE_WGRAD=<columns> re-enables it for launches of at most that many voxel columns (experiments only)."""
    lim = int(os.environ.get("FZ_SIDE_WGRAD", "0"))
    if lim <= 0 or ncols > lim:
        return None
    key = (dev.type, dev.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


class _LateJoin:
    """Opt-in (`late_wgrad_join`): the weight-gradient kernels that a block backward puts on the second
    stream are NOT awaited at the end of that block (the current stream then idles 30-120 us per block
    for the last of them) but once, before anything reads the gradients: `join_wgrad_streams()`, called
    by `FlatGradSync` (before its collectives) and `FlatAdamW.step()`.  Until then `p.grad` holds the
    tensor the side stream is still writing, so this is only for callers that own the step: autograd
    must take the returned tensors over without copying them (it does when `p.grad` is None), which
    the join VERIFIES — a gradient that was copied early raises instead of training on garbage."""
    enabled = False
    keep = []      # tensors read by in-flight side-stream kernels
    owed = []      # (weakref(param), data_ptr of the gradient tensor written on the side stream)
    devices = set()
    queued = False  # an end-of-backward engine callback is pending
    verified = 0    # gradients whose ownership has been checked so far (tests)


def _end_of_backward():
    _LateJoin.queued = False
    join_wgrad_streams()


def reset_late_join():
    """Start of a new step (the owners' zero_grad): meet the side streams and drop whatever a FAILED backward left
    behind — an engine callback that was queued but never ran (the engine drops it when backward() raises) would leave
    `queued` set and no later backward would re-queue the end-of-backward join; stale `owed` entries would raise a
    spurious ownership error at the next join."""
    d = _LateJoin
    if d.devices:
        wait_wgrad_streams()
    d.owed.clear()
    d.keep.clear()
    d.queued = False


def late_wgrad_join(flag: bool = True):
    if not flag:
        join_wgrad_streams()
    _LateJoin.enabled = bool(flag) and os.environ.get("FZ_LATE_JOIN", "1") != "0"


def wait_wgrad_streams():
    """Current stream waits for the side-stream weight gradients issued so far (no ownership check:
    usable from inside backward, e.g. a bucket hook, while later gradients are still to be assigned)."""
    d = _LateJoin
    for key in list(d.devices):
        dev = torch.device(*key)
        torch.cuda.current_stream(dev).wait_stream(_SIDE[key])
    d.devices.clear()
    d.keep.clear()


def wait_wgrad_streams_all(dev):
    """Current stream of `dev` waits for that device's side stream whether or not a deferred join is pending (used in
    front of the few kernels that must not run beside any other kernel of the library, functional._join_side_streams)."""
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    st = _SIDE.get(key)
    if st is not None:
        torch.cuda.current_stream(dev).wait_stream(st)


def join_wgrad_streams():
    """After backward: `wait_wgrad_streams()` + check that every parameter still holds the very tensor
    its gradient was written into."""
    d = _LateJoin
    wait_wgrad_streams()
    owed, d.owed = d.owed, []
    d.verified += len(owed)
    for ref, ptr in owed:
        prm = ref()
        if prm is not None and prm.is_leaf and prm.requires_grad and (prm.grad is None or prm.grad.data_ptr() != ptr):
            raise RuntimeError("late_wgrad_join: a weight gradient was copied or replaced before the kernel "
                               "writing it had finished (accumulating into an existing .grad?); call "
                               "factorizer_amd.pointwise.late_wgrad_join(False) for this training loop")


class _Defer:
    """Deferred finishes (csrc/finish.h; include/factorizer_hip.h: fz_finish_defer / fz_finish_flush_all).  Every weight-gradient
    launch ends with a tiny fixed-order reduction of per-workgroup partial rows whose output is a PARAMETER gradient; a README
    training step has 54 of them, each a 5-10 us serial slot of the stream.  A caller that owns the step — `FlatAdamW`,
    `FlatGradSync`: gradients are None at zero_grad, nothing reads them before the owner does — lets the library queue them and
    run them as one grid at the end of the backward (and in front of every gradient bucket's collective).
    Armed by the owner's zero_grad FOR THE OWNER'S PARAMETERS, disarmed by the end-of-backward flush: a second backward
    without zero_grad (gradient accumulation: autograd then ADDS to .grad while the backward runs) is never deferred.
    A returned gradient buffer stays unwritten until the flush, so a node defers only when nothing can read its parameter
    gradients before the backward ends — `_param_sinks_ok` decides that per node from the autograd graph (each one must go,
    through views at most, straight to the AccumulateGrad of an owned leaf that has no gradient yet, no tensor hooks, and is
    used once in this backward; no double backward).  Everything else — a weight that is computed from a parameter (the
    padded stem weight of odd channel counts), a parameter used twice, `p.register_hook`, `create_graph=True`, a model the
    owner does not cover — runs its finishes at once, as do sites whose results feed a torch op inside the backward
    (`_no_defer`).  Post-accumulate hooks other than the owner's must call `flush_finishes()` before they read `.grad`.
    The arithmetic is the same either way (bitwise)."""
    enabled = False
    armed = False
    suppress = 0
    owned = frozenset()   # id() of the parameters the arming owner covers
    seen = set()          # id() of the parameters whose gradient a node of this backward has already produced
    keep = {}             # device index -> workspaces and gradient outputs of queued finishes: alive until the flush
    lock = _threading.Lock()
    queued = False  # an end-of-backward engine callback is pending
    flushed = 0     # finishes run through flushes so far (tests)
    flushes = 0
    refused = 0     # backward nodes that ran their finishes at once because `_param_sinks_ok` said no (tests)


def defer_finishes(flag: bool = True):
    """Owner-level switch (FlatAdamW / FlatGradSync constructors); FZ_DEFER_FINISH=0 in the environment keeps it off."""
    if not flag:
        flush_finishes(disarm=True)
    _Defer.enabled = bool(flag) and os.environ.get("FZ_DEFER_FINISH", "1") != "0"


def arm_deferred_finishes(params=()):
    """The owner's zero_grad: the gradients of `params` are None from here to the end of the next backward."""
    flush_finishes(disarm=True)
    d = _Defer
    d.queued = False
    d.owned = params if isinstance(params, frozenset) else frozenset(id(p) for p in params)
    d.seen = set()
    d.armed = d.enabled and bool(d.owned)


def flush_finishes(disarm: bool = False):
    """Run what is queued — every queue on the stream its jobs were issued on; the current stream of each device waits for
    them (a forward under `torch.cuda.stream(s)` with `backward()` called outside it queues on `s`) — and release the buffers
    held for it."""
    d = _Defer
    if disarm:
        d.armed = False
    with d.lock:
        devs, keep, d.keep = list(d.keep), d.keep, {}
    for idx in devs:
        with torch.cuda.device(idx):
            n = N.lib().fz_finish_pending()
            if n:
                rc = Fn._timed("finish_batch", 0, lambda: N.lib().fz_finish_flush_all(torch.cuda.current_stream(idx).cuda_stream), cols=0)
                if rc < 0:
                    N.check(rc, "fz_finish_flush_all")
                d.flushed += rc
                d.flushes += 1
    keep.clear()


def _end_of_backward_finishes():
    _Defer.queued = False
    _Defer.seen = set()
    flush_finishes(disarm=True)


class _finish_scope:
    """with _finish_scope(ws, gw, gb, ...): the finish launches of the calls inside may be deferred; the tensors named —
    EVERY buffer those finishes read or write that autograd does not own — are held until the flush."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.on = False

    def __enter__(self):
        d = _Defer
        if d.armed and d.suppress == 0:
            t0 = next((t for t in self.tensors if t is not None), None)
            if t0 is not None and t0.is_cuda:
                if not d.queued:
                    try:   # only while the autograd engine runs a backward: otherwise the finishes stay immediate
                        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward_finishes)
                        d.queued = True
                    except RuntimeError:
                        return self
                # (aliases, not the tensors themselves: autograd adopts a returned gradient as p.grad without a copy only while
                # nothing else references the tensor object — holding the object would make it CLONE the not-yet-written buffer)
                held = [t.detach() for t in self.tensors if t is not None]
                with d.lock:
                    d.keep.setdefault(t0.device.index, []).extend(held)
                self.was = N.lib().fz_finish_defer(1)   # (per thread: the launches inside this scope are issued by this thread)
                self.on = True
        return self

    def __exit__(self, *exc):
        if self.on:
            N.lib().fz_finish_defer(self.was)
        return False


class _no_defer:
    def __enter__(self):
        _Defer.suppress += 1

    def __exit__(self, *exc):
        _Defer.suppress -= 1
        return False


# autograd nodes a gradient passes through WITHOUT being read (and without losing the dense layout AccumulateGrad adopts as is)
_VIEW_NODES = frozenset(("ViewBackward0", "ReshapeAliasBackward0", "UnsafeViewBackward0", "UnsqueezeBackward0", "SqueezeBackward0",
                         "SqueezeBackward1", "AliasBackward0", "DetachBackward0"))


def _param_sinks_ok(ctx) -> bool:
    """May this backward node leave its parameter gradients unwritten until the end of the backward?  Inputs listed in the
    Function's `_fz_acts` (default: input 0) are activations — their gradients are complete when the node returns; every other
    input that needs a gradient gets it from a finish job and must lead, through views at most, to the AccumulateGrad of an
    owned, gradient-less, hook-less leaf that no earlier node of this backward has served."""
    d = _Defer
    if torch.is_grad_enabled():          # create_graph=True: the gradients are graph inputs of a double backward
        return False
    acts = getattr(getattr(ctx, "_forward_cls", None), "_fz_acts", (0,))   # (a backward ctx is not an instance of the Function class)
    for i, need in enumerate(ctx.needs_input_grad):
        if not need or i in acts:
            continue
        fn = ctx.next_functions[i][0]
        while fn is not None and type(fn).__name__ in _VIEW_NODES:
            fn = fn.next_functions[0][0]
        if fn is None or type(fn).__name__ != "AccumulateGrad":
            if os.environ.get("FZ_DEFER_DEBUG"):
                print(f"[defer] {type(ctx).__name__}: input {i} sink {type(fn).__name__}", flush=True)
            return False
        v = fn.variable
        vid = id(v)
        if vid not in d.owned or v.grad is not None or v._backward_hooks:
            if os.environ.get("FZ_DEFER_DEBUG"):
                print(f"[defer] {type(ctx).__name__}: input {i} {tuple(v.shape)} owned={vid in d.owned} grad={v.grad is not None} hooks={bool(v._backward_hooks)}", flush=True)
            return False
        if vid in d.seen:
            if os.environ.get("FZ_DEFER_DEBUG"):
                print(f"[defer] {type(ctx).__name__}: input {i} {tuple(v.shape)} already served in this backward", flush=True)
            # second use of a parameter in one graph: the engine is about to ADD this node's gradient to the first one's —
            # which must therefore be complete now
            flush_finishes()
            return False
        d.seen.add(vid)
    return True


def _bwd(fn):
    """decorator of every backward below: N.with_products + the deferral decision above (made once per node)"""
    import functools
    inner = N.with_products(fn)

    @functools.wraps(fn)
    def wrapper(ctx, *a, **kw):
        d = _Defer
        if d.armed and d.suppress == 0 and not getattr(ctx, "_fz_defer_checked", False):
            ctx._fz_defer_checked = True
            if not _param_sinks_ok(ctx):
                d.refused += 1
                with _no_defer():
                    return inner(ctx, *a, **kw)
        return inner(ctx, *a, **kw)
    return wrapper


def _native_ok(*ts):
    """Device tensors the GEMM family takes: activations (5-D, first) fp32 or — mixed precision — bf16, all of
    one type; parameters (<= 3-D) fp32; voxel count divisible by 4."""
    t0 = ts[0]
    if not (t0.is_cuda and t0.numel() > 0 and t0.dtype in (torch.float32, torch.bfloat16) and _vox(t0) % 4 == 0):
        return False
    for t in ts[1:]:
        if t is None:
            continue
        if t.dim() >= 4:
            if t.dtype != t0.dtype:
                return False
        elif t.dtype != torch.float32:
            return False
    return True


def _head_rows_reduce(part, rows, w2, M, like):
    """The head backward's partial rows (132 floats: gW[m][c] at m*32 + c, gb[m] at 128 + m) -> (gw, gb): two column blocks of the
    rows, each reduced straight into its gradient tensor (the weight's slice of a flat gradient buffer when one is attached)."""
    lib = N.lib()
    gw = _GB.out_like(w2, (M, 32), torch.float32)
    gb = torch.empty(M, dtype=torch.float32, device=like.device)
    st = N.stream_ptr(like)
    with _finish_scope(part, gw, gb):
        N.check(lib.fz_chunk_reduce_ld(part.data_ptr(), rows, M * 32, 132, gw.data_ptr(), 0, st), "fz_chunk_reduce_ld")
        N.check(lib.fz_chunk_reduce_ld(part.data_ptr() + 128 * 4, rows, M, 132, gb.data_ptr(), 0, st), "fz_chunk_reduce_ld")
    return gw, gb


# ---- LayerNorm → Linear → [ReLU] -----------------------------------------------------------
class LNLinearFn(torch.autograd.Function):
    @staticmethod
    @N.capture_products
    def forward(ctx, x, ln_w, ln_b, eps, w, b, act):
        x = x.contiguous()
        B, C = x.shape[:2]
        V = _vox(x)
        M = w.shape[0]
        w2 = w.reshape(M, C)
        y = torch.empty((B, M, *x.shape[2:]), dtype=x.dtype, device=x.device)
        stats = torch.empty((B, 2, V), dtype=torch.float32, device=x.device)
        _gemm([x], w2, y, B=B, Cin=C, Vin=V, M=M, K=C, Ncol=V, bias=b, ln=(ln_w, ln_b, eps), stats_out=stats,
              eact=ACT["relu" if act == "relu_out" else act], name="ln_linear")
        ctx.save_for_backward(x, stats, ln_w, ln_b, w2, y if act == "relu" else None)
        ctx.act, ctx.has_bias, ctx.wshape = act, b is not None, w.shape
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, stats, ln_w, ln_b, w2, y = ctx.saved_tensors
        gy = gy.contiguous()
        B, C = x.shape[:2]
        V = _vox(x)
        M = w2.shape[0]
        gate = y if ctx.act == "relu" else None
        # gl = Wᵀ (gy ∘ relu'(y))
        gl = torch.empty_like(x)
        _gemm([gy], w2, gl, B=B, Cin=M, Vin=V, M=C, K=M, Ncol=V, w_t=True, ldw=C, bmul=gate,
              bmul_kind=ACT["relu"], name="linear_dgrad")
        gx, ggamma, gbeta = _ln_backward(gl, x, stats, ln_w)
        # weight / bias grads: GW = (gy∘gate) · LN(x)ᵀ with the affine folded in the reduce step
        gw = _GB.out_like(w2)
        gb = torch.empty(M, dtype=torch.float32, device=x.device)
        _wgrad(gy, [x], gw, B=B, M=M, Cin=C, K=C, Vq=V, Ncols=V, gbias=gb, pmul=gate, pmul_kind=ACT["relu"],
               stats=stats, ln=(ln_w, ln_b), name="wgrad_ln_linear")
        return gx, ggamma, gbeta, None, gw.reshape(ctx.wshape), (gb if ctx.has_bias else None), None


# ---- res + Linear(act(z)) + bias -------------------------------------------------------------
class ActLinearResFn(torch.autograd.Function):
    _fz_acts = (0, 3)   # inputs that are activations (see _param_sinks_ok); every other input is a parameter

    @staticmethod
    @N.capture_products
    def forward(ctx, z, w, b, res, bact):
        z = z.contiguous()
        B, C = z.shape[:2]
        V = _vox(z)
        M = w.shape[0]
        w2 = w.reshape(M, C)
        if res is not None:
            res = res.contiguous()
        y = torch.empty((B, M, *z.shape[2:]), dtype=z.dtype, device=z.device)
        _gemm([z], w2, y, B=B, Cin=C, Vin=V, M=M, K=C, Ncol=V, bias=b, bact=ACT[bact], res=res,
              name="act_linear_res")
        ctx.save_for_backward(z, w2)
        ctx.bact, ctx.has_bias, ctx.has_res, ctx.wshape = bact, b is not None, res is not None, w.shape
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        z, w2 = ctx.saved_tensors
        gy = gy.contiguous()
        B, C = z.shape[:2]
        V = _vox(z)
        M = w2.shape[0]
        gz = torch.empty_like(z)
        if M <= 4 and C == 32 and ctx.bact == "none" and V % 4 == 0 and _HEAD_BWD:
            # the head (32 -> out_channels): input, weight and bias gradient in one bandwidth-bound pass (csrc/headbwd.hip)
            lib = N.lib()
            rows = lib.fz_head_bwd_rows()
            part = torch.empty(lib.fz_head_bwd_workspace_bytes() // 4, dtype=torch.float32, device=z.device)
            es = z.element_size()
            with torch.cuda.device(z.device):
                rc = Fn._timed(f"head_bwd_{C}->{M}", es * (gy.numel() + 2 * z.numel()),
                               lambda: lib.fz_head_bwd(gy.data_ptr(), z.data_ptr(), w2.data_ptr(), gz.data_ptr(), part.data_ptr(),
                                                       B, M, C, V, N.act_dtype(z), N.stream_ptr(z)), cols=B * V)
                N.check(rc, "fz_head_bwd")
                gw, gb = _head_rows_reduce(part, rows, w2, M, z)
            return gz, gw.reshape(ctx.wshape), (gb if ctx.has_bias else None), (gy if ctx.has_res else None), None
        _gemm([gy], w2, gz, B=B, Cin=M, Vin=V, M=C, K=M, Ncol=V, w_t=True, ldw=C,
              emul=(z if ctx.bact != "none" else None), emul_kind=ACT[ctx.bact], name="linear_dgrad")
        gw = _GB.out_like(w2)
        gb = torch.empty(M, dtype=torch.float32, device=z.device)
        _wgrad(gy, [z], gw, B=B, M=M, Cin=C, K=C, Vq=V, Ncols=V, gbias=gb, qact=ACT[ctx.bact], name="wgrad_linear")
        return gz, gw.reshape(ctx.wshape), (gb if ctx.has_bias else None), (gy if ctx.has_res else None), None


# ---- Linear over a virtual channel concat -------------------------------------------------------
class CatLinearFn(torch.autograd.Function):
    _fz_acts = (0, 1)   # inputs that are activations (see _param_sinks_ok); every other input is a parameter

    @staticmethod
    @N.capture_products
    def forward(ctx, x1, x2, w, b):
        x1, x2 = x1.contiguous(), x2.contiguous()
        B, C1 = x1.shape[:2]
        C2 = x2.shape[1]
        V = _vox(x1)
        M = w.shape[0]
        w2 = w.reshape(M, C1 + C2)
        y = torch.empty((B, M, *x1.shape[2:]), dtype=x1.dtype, device=x1.device)
        _gemm([x1, x2], w2, y, B=B, Cin=C1 + C2, Vin=V, M=M, K=C1 + C2, Ncol=V, bias=b, c0=C1, name="cat_linear")
        ctx.save_for_backward(x1, x2, w2)
        ctx.has_bias, ctx.wshape = b is not None, w.shape
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x1, x2, w2 = ctx.saved_tensors
        gy = gy.contiguous()
        B, C1 = x1.shape[:2]
        C2 = x2.shape[1]
        V = _vox(x1)
        M = w2.shape[0]
        C = C1 + C2
        if M == 32 and C1 == 32 and C2 == 32 and _dw_fused_ok(M, V):
            # full-resolution adapter (64 -> 32): each half is a 32 -> 32 layer — input gradient and weight gradient of a
            # half in one pass (gy is read twice instead of three times, x1 / x2 once instead of twice)
            gw = _GB.out_like(w2)
            g1, _, gb, _, _ = _gemm_dw(gy, w2[:, :C1], x1, want_bias=ctx.has_bias, name="dgrad_wgrad", gw_out=gw[:, :C1])
            g2, _, _, _, _ = _gemm_dw(gy, w2[:, C1:], x2, name="dgrad_wgrad", gw_out=gw[:, C1:])
            return g1, g2, gw.reshape(ctx.wshape), gb
        g1 = torch.empty_like(x1)
        g2 = torch.empty_like(x2)
        _gemm([gy], w2, g1, B=B, Cin=M, Vin=V, M=C1, K=M, Ncol=V, w_t=True, ldw=C, name="linear_dgrad")
        _gemm([gy], w2[:, C1:], g2, B=B, Cin=M, Vin=V, M=C2, K=M, Ncol=V, w_t=True, ldw=C, name="linear_dgrad")
        gw = _GB.out_like(w2)
        gb = torch.empty(M, dtype=torch.float32, device=gy.device)
        _wgrad(gy, [x1, x2], gw, B=B, M=M, Cin=C, K=C, Vq=V, Ncols=V, gbias=gb, c0=C1, name="wgrad_cat_linear")
        return g1, g2, gw.reshape(ctx.wshape), (gb if ctx.has_bias else None)


class LinearFn(torch.autograd.Function):
    @staticmethod
    @N.capture_products
    def forward(ctx, x, w, b):
        return ActLinearResFn.forward(ctx, x, w, b, None, "none")

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        return ActLinearResFn.backward(ctx, gy)[:3]


# ---- standalone LayerNorm -------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    @N.capture_products
    def forward(ctx, x, w, b, eps):
        x = x.contiguous()
        B, C = x.shape[:2]
        V = _vox(x)
        y = torch.empty_like(x)
        stats = torch.empty((B, 2, V), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = Fn._timed(f"ln_fwd_{C}", 2 * x.element_size() * x.numel(), lambda: N.lib().fz_ln_fwd(
                x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), stats.data_ptr(), B, C, V, eps,
                N.act_dtype(x), N.stream_ptr(x)), cols=B * V)
        N.check(rc, "fz_ln_fwd")
        ctx.save_for_backward(x, stats, w)
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, stats, w = ctx.saved_tensors
        gy = gy.contiguous()
        B, C = x.shape[:2]
        V = _vox(x)
        gx, ggamma, gbeta = _ln_backward(gy, x, stats, w)
        return gx, ggamma, gbeta, None


# ---- strided convolutions of the U-shape ------------------------------------------------------------
class ConvK2S2Fn(torch.autograd.Function):
    """Conv3d(kernel 2, stride 2): space-to-depth gather fused into the GEMM's operand loads."""

    @staticmethod
    @N.capture_products
    def forward(ctx, x, w, b):
        x = x.contiguous()
        B, C, D, H, W = x.shape
        O = w.shape[0]
        Do, Ho, Wo = D // 2, H // 2, W // 2
        y = torch.empty((B, O, Do, Ho, Wo), dtype=x.dtype, device=x.device)
        _gemm([x], w, y, B=B, Cin=C, Vin=D * H * W, M=O, K=8 * C, Ncol=Do * Ho * Wo, bias=b, loader=LOAD_S2D,
              Di=D, Hi=H, Wi=W, Ho=Ho, Wo=Wo, name="conv_k2s2")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy, g_skip=None):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        B, C, D, H, W = x.shape
        O = w.shape[0]
        Do, Ho, Wo = D // 2, H // 2, W // 2
        Vc = Do * Ho * Wo
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            if g_skip is not None:
                g_skip = g_skip.contiguous()
            # rows (ci, tap), reduction over o:  A[m][k] = w[k*(8C) + m]
            _gemm([gy], w, gx, B=B, Cin=O, Vin=Vc, M=8 * C, K=O, Ncol=Vc, w_t=True, ldw=8 * C,
                  epilogue=EPI_D2S, Ho=Ho, Wo=Wo, res=g_skip, name="conv_k2s2_dgrad")
        gw = _GB.out_like(w)
        gb = torch.empty(O, dtype=torch.float32, device=x.device)
        _wgrad(gy, [x], gw, B=B, M=O, Cin=C, K=8 * C, Vq=D * H * W, Ncols=Vc, gbias=gb, loader=LOAD_S2D,
               D=D, H=H, W=W, Ho=Ho, Wo=Wo, name="wgrad_conv_k2s2")
        return gx, gw, (gb if ctx.has_bias else None)


class SkipConvK2S2Fn(torch.autograd.Function):
    """(x, Conv3d(k=2, s=2)(x)) as ONE node: the encoder output feeds both the skip connection and
    the next stage's down-convolution (unet.py:95-99), so its gradient is the sum of two terms that
    autograd would add in a separate full-tensor pass; here the skip gradient is the residual of the
    depth-to-space epilogue of the convolution's input-gradient kernel."""

    @staticmethod
    @N.capture_products
    def forward(ctx, x, w, b):
        y = ConvK2S2Fn.forward(ctx, x, w, b)
        return x.view_as(x), y

    @staticmethod
    @_bwd
    def backward(ctx, g_skip, gy):
        if gy is None:  # the down path took no part in the loss
            return g_skip, None, None
        return ConvK2S2Fn.backward(ctx, gy, g_skip=g_skip)


class TConvK2S2Fn(torch.autograd.Function):
    """ConvTranspose3d(kernel 2, stride 2): GEMM with (o, tap) rows + depth-to-space epilogue."""

    @staticmethod
    @N.capture_products
    def forward(ctx, x, w, b):
        x = x.contiguous()
        B, C, D, H, W = x.shape
        O = w.shape[1]
        y = torch.empty((B, O, 2 * D, 2 * H, 2 * W), dtype=x.dtype, device=x.device)
        V = D * H * W
        _gemm([x], w, y, B=B, Cin=C, Vin=V, M=8 * O, K=C, Ncol=V, w_t=True, ldw=8 * O, bias=b,
              epilogue=EPI_D2S, Ho=H, Wo=W, name="tconv_k2s2")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        B, C, D, H, W = x.shape
        O = w.shape[1]
        V = D * H * W
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            # GX[ci, coarse] = Σ_{o,tap} w[ci, o, tap] · GY[o, fine]: a k2s2 "conv" of gy with w as [C][8O]
            _gemm([gy], w, gx, B=B, Cin=O, Vin=8 * V, M=C, K=8 * O, Ncol=V, loader=LOAD_S2D,
                  Di=2 * D, Hi=2 * H, Wi=2 * W, Ho=H, Wo=W, name="tconv_k2s2_dgrad")
        gw = _GB.out_like(w)
        # GW[ci][(o,tap)] = Σ X[ci][n] · GY[o][fine(n,tap)]
        _wgrad(x, [gy], gw, B=B, M=C, Cin=O, K=8 * O, Vq=8 * V, Ncols=V, loader=LOAD_S2D, D=2 * D, H=2 * H,
               W=2 * W, Ho=H, Wo=W, name="wgrad_tconv_k2s2")
        gb = None
        if ctx.has_bias:
            Vf = 8 * V
            gb = torch.empty(O, dtype=torch.float32, device=x.device)
            part = torch.empty(B * N.lib().fz_rowsum_chunks(Vf) * O, dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device), _finish_scope(part, gb):
                rc = N.lib().fz_rowsum(gy.data_ptr(), part.data_ptr(), gb.data_ptr(), B, O, Vf, N.act_dtype(gy),
                                       N.stream_ptr(x))
            N.check(rc, "fz_rowsum")
        return gx, gw, gb


class ConvK3Fn(torch.autograd.Function):
    """Conv3d(kernel 3, padding 1) — the stem (C_in = 4): direct implicit-GEMM kernels
    (csrc/conv3.hip); shapes outside them use the generic tap loaders of the GEMM family."""

    _fz_acts = (0, 3)   # inputs that are activations (see _param_sinks_ok); every other input is a parameter

    @staticmethod
    @N.capture_products
    def forward(ctx, x, w, b, pro=None):
        # pro: as UpCatLinearFn.forward — the consuming block's (ln1 weight, ln1 bias, eps, in_proj weight); the node then returns
        # (y, t, statistics), the last two formed in the same launch and not differentiable here
        x = x.contiguous()
        B, C, D, H, W = x.shape
        O = w.shape[0]
        V = D * H * W
        y = torch.empty((B, O, D, H, W), dtype=x.dtype, device=x.device)
        ctx.npro = 0 if pro is None else 1
        if C * 27 * 64 * 4 <= 65536:
            bp = None
            if pro is not None:
                ln_w, ln_b, eps, w_in = pro
                t_pro = torch.empty_like(y)
                st_pro = torch.empty((B, 2, V), dtype=torch.float32, device=x.device)
                bp = N.BlockPrologue(ln_w.data_ptr(), ln_b.data_ptr(), float(eps), w_in.data_ptr(), t_pro.data_ptr(), st_pro.data_ptr())
            with torch.cuda.device(x.device):
                rc = Fn._timed(f"conv_k3_{C}->{O}" + ("_ln_linear" if bp is not None else ""),
                               x.element_size() * (x.numel() + y.numel() * (2 if bp is not None else 1)), lambda: N.lib().fz_conv3_fwd2(
                    x.data_ptr(), w.data_ptr(), _p(b), y.data_ptr(), B, C, O, D, H, W, N.act_dtype(x), N.products(),
                    ctypes.byref(bp) if bp is not None else None, N.stream_ptr(x)),
                    cols=B * V, flops=2 * B * V * 27 * C * O)
            N.check(rc, "fz_conv3_fwd")
        elif pro is not None:
            raise RuntimeError("ConvK3Fn: a block prologue needs the stem kernel (callers check conv3_prologue_ok)")
        else:
            _gemm([x], w, y, B=B, Cin=C, Vin=V, M=O, K=27 * C, Ncol=V, bias=b, loader=LOAD_K3, Di=D, Hi=H, Wi=W,
                  name="conv_k3")
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        if pro is not None:
            ctx.mark_non_differentiable(t_pro, st_pro)
            ctx.set_materialize_grads(False)
            return y, t_pro, st_pro
        return y

    @staticmethod
    @_bwd
    def backward(ctx, gy, _gt=None, _gst=None):
        x, w = ctx.saved_tensors
        if gy is None:
            gy = torch.zeros((x.shape[0], w.shape[0], *x.shape[2:]), dtype=x.dtype, device=x.device)
        gy = gy.contiguous()
        B, C, D, H, W = x.shape
        O = w.shape[0]
        V = D * H * W
        gx = None
        if ctx.needs_input_grad[0]:
            # not on the training path (the stem's input is data): the adjoint is the same k3 correlation with the
            # channel-transposed, spatially flipped filters — the generic tap loader of the GEMM family
            if O % 2 == 0 and W % 4 == 0:
                wt = torch.flip(w, dims=(2, 3, 4)).transpose(0, 1).contiguous()
                gx = torch.empty_like(x)
                _gemm([gy], wt, gx, B=B, Cin=O, Vin=V, M=C, K=27 * O, Ncol=V, loader=LOAD_K3, Di=D, Hi=H, Wi=W,
                      name="conv_k3_dgrad")
            else:
                _warn_composed("Conv3d(k=3) input gradient", x)
                gx = torch.nn.grad.conv3d_input(x.shape, w, gy, padding=1)
        gw = _GB.out_like(w)
        gb = torch.empty(O, dtype=torch.float32, device=x.device)
        if W % 32 == 0 and 27 * C <= 128:
            lib = N.lib()
            nchunk = lib.fz_conv3_wgrad_chunks(B, D, H, W)
            K = 27 * C
            part = torch.empty(nchunk * O * K, dtype=torch.float32, device=x.device)
            pbias = torch.empty(nchunk * O, dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device), _finish_scope(part, pbias, gw, gb):
                st = N.stream_ptr(x)
                prod = N.products()

                def run():
                    rc = lib.fz_conv3_wgrad_partials(gy.data_ptr(), x.data_ptr(), part.data_ptr(), pbias.data_ptr(),
                                                     B, C, O, D, H, W, N.act_dtype(x), prod, st)
                    if rc == 0:
                        rc = lib.fz_chunk_reduce(part.data_ptr(), nchunk, O * K, gw.data_ptr(), 0, st)
                    if rc == 0:
                        rc = lib.fz_chunk_reduce(pbias.data_ptr(), nchunk, O, gb.data_ptr(), 0, st)
                    return rc
                rc = Fn._timed(f"wgrad_conv_k3_{O}x{K}", x.element_size() * (x.numel() + gy.numel()), run,
                               cols=B * V, flops=2 * B * V * K * O)
            N.check(rc, "fz_conv3_wgrad")
        else:
            _wgrad(gy, [x], gw, B=B, M=O, Cin=C, K=27 * C, Vq=V, Ncols=V, gbias=gb, loader=LOAD_K3, D=D, H=H, W=W,
                   name="wgrad_conv_k3")
        return (gx, gw, (gb if ctx.has_bias else None), None)[:len(ctx.needs_input_grad)]   # (one per argument actually passed)


# ---- public dispatchers (device → native, CPU → composed ATen) ---------------------------------------
def _warn_composed(op, x, *more):
    """A device tensor is about to take the composed-ATen path: say so, once per (op, shape, dtype)."""
    if x.is_cuda and x.numel():
        from .composed import warn_once
        dts = sorted({str(t.dtype) for t in (x, *more) if t is not None})
        warn_once(f"{op}:{tuple(x.shape)}:{dts}",
                  f"{op} on {tuple(x.shape)} ({', '.join(dts)}) is outside the native kernel set (needs fp32 or "
                  "bf16 activations with fp32 parameters, an even channel count and a voxel count divisible by 4); "
                  "using composed framework ops on device")


def linear_cf(x, weight, bias=None):
    if _native_ok(x, weight, bias) and weight.shape[1] % 2 == 0:
        return LinearFn.apply(x, weight, bias)
    _warn_composed("Linear", x, weight, bias)
    B, C = x.shape[:2]
    y = torch.matmul(weight.reshape(weight.shape[0], weight.shape[1]), x.reshape(B, C, _vox(x)))
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    return y.reshape(B, weight.shape[0], *x.shape[2:])


def layernorm_cf(x, weight, bias, eps):
    if weight is not None and bias is not None and _native_ok(x, weight, bias):
        return LayerNormFn.apply(x, weight, bias, eps)
    _warn_composed("LayerNorm", x, weight, bias)
    y = F.layer_norm(x.movedim(1, -1), (x.shape[1],), weight, bias, eps)
    return y.movedim(-1, 1).contiguous()


def mlp_cf(x, w1, b1, w2, b2):
    if _native_ok(x, w1, b1, w2, b2) and w1.shape[1] % 2 == 0 and w2.shape[1] % 2 == 0:
        z = LinearFn.apply(x, w1, b1)
        return ActLinearResFn.apply(z, w2, b2, None, "gelu")
    _warn_composed("MLP", x, w1, b1, w2, b2)
    return linear_cf(F.gelu(linear_cf(x, w1, b1)), w2, b2)


def ln_linear(x, ln_w, ln_b, eps, w, b, act="none"):
    """act(Linear(LayerNorm(x))).  act: "none" | "relu" | "relu_out" (ReLU applied, but the
    incoming gradient is already gated by the consumer — FactCoreFn — so backward skips it)."""
    if _native_ok(x, ln_w, ln_b, w, b) and w.shape[1] % 2 == 0:
        return LNLinearFn.apply(x, ln_w, ln_b, eps, w, b, act)
    _warn_composed("LayerNorm+Linear", x, ln_w, ln_b, w, b)
    y = linear_cf(layernorm_cf(x, ln_w, ln_b, eps), w, b)
    return torch.relu(y) if act != "none" else y


def act_linear_res(z, w, b, res, act="none"):
    """res + Linear(act(z))."""
    if _native_ok(z, w, b, res) and w.shape[1] % 2 == 0:
        return ActLinearResFn.apply(z, w, b, res, act)
    _warn_composed("Linear+residual", z, w, b, res)
    y = linear_cf(F.gelu(z) if act == "gelu" else z, w, b)
    return y if res is None else res + y


# ---- ConvTranspose3d(k2, s2) + virtual concat + Linear as ONE autograd node ---------------------------------------
class UpCatLinearFn(torch.autograd.Function):
    """out = adapter(cat([skip, upsample(deep)], 1)) — the first three lines of a decoder level (unet.py:125-128 with
    the FactorizerStage adapter, factorizer.py:116).  Forward: the two existing launches (the up-sampled tensor is a
    temporary).  Backward: the adapter's up half W_b and the transposed convolution W_t are both linear with nothing in
    between, so every gradient is formed from the COMPOSED weights  Wc[k][m][tap] = Σ_c W_t[k][c][tap]·W_b[m][c]:
      g_deep = k2s2-correlation of g with Wc          (reads g once — not W_bᵀg written, then read back)
      Gt[k][m][tap] = Σ_n deep[k][n]·g[m][fine(n,tap)]  (the transposed-convolution weight-gradient launch, on g itself)
      dW_b = Σ_{k,tap} Gt·W_t + (Σ_v g)⊗b_t ,  dW_t = Σ_m W_b·Gt ,  db_t = W_bᵀ·Σ_v g          (tiny contractions)
    and the up-sampled tensor, its gradient and the row-sum pass over it never exist in the backward: 6 U of traffic per
    level instead of 10 U (U = one full-resolution activation tensor of the level)."""

    _fz_acts = (0, 1, 6)   # inputs that are activations (see _param_sinks_ok); every other input is a parameter

    @staticmethod
    @N.capture_products
    def forward(ctx, skip, deep, w_t, b_t, w_ad, b_ad, pro=None):
        # pro = (ln1 weight, ln1 bias, eps, in_proj weight) of the FactorizerBlock that consumes the output: the node then also
        # returns t = relu(in_proj(LN1(out))) and the LayerNorm statistics, formed in the same launch (not differentiable here:
        # the block's own backward carries the gradient through them)
        skip, deep = skip.contiguous(), deep.contiguous()
        B, Cd, D, H, W = deep.shape
        O = w_t.shape[1]
        V = D * H * W
        C1 = skip.shape[1]
        M = w_ad.shape[0]
        w2 = w_ad.reshape(M, C1 + O)
        y = torch.empty((B, M, *skip.shape[2:]), dtype=skip.dtype, device=skip.device)
        if (M == 32 and C1 == 32 and O == 32 and _UPCAT_FWD and N.lib().fz_upcat_supported(C1, Cd, D, H, W)):
            # one pass over (skip, deep) -> out with the composed weights (csrc/upcat.hip): no up-sampled tensor at all
            w_b = w2[:, C1:]
            wbt = torch.empty((8, M, Cd), dtype=torch.float32, device=skip.device)
            bias = torch.empty(M, dtype=torch.float32, device=skip.device) if (b_t is not None or b_ad is not None) else None
            with torch.cuda.device(skip.device):
                N.check(N.lib().fz_upcat_compose(w_t.data_ptr(), w_b.data_ptr(), C1 + O, _p(b_t), _p(b_ad), None, wbt.data_ptr(),
                                                 _p(bias), Cd, O, M, N.stream_ptr(skip)), "fz_upcat_compose")
            es = skip.element_size()
            bp = None
            if pro is not None:
                ln_w, ln_b, eps, w_in = pro
                t_pro = torch.empty_like(y)
                st_pro = torch.empty((B, 2, 8 * V), dtype=torch.float32, device=skip.device)
                bp = N.BlockPrologue(ln_w.data_ptr(), ln_b.data_ptr(), float(eps), w_in.data_ptr(), t_pro.data_ptr(), st_pro.data_ptr())
            with torch.cuda.device(skip.device):
                rc = Fn._timed(f"upcat_{Cd}->{M}" + ("_ln_linear" if bp is not None else ""),
                               es * (skip.numel() + deep.numel() + y.numel() * (2 if bp is not None else 1)),
                               lambda: N.lib().fz_upcat2(skip.data_ptr(), deep.data_ptr(), w2.data_ptr(), C1 + O, wbt.data_ptr(),
                                                         _p(bias), y.data_ptr(), B, C1, Cd, D, H, W, N.act_dtype(skip),
                                                         ctypes.byref(bp) if bp is not None else None, N.stream_ptr(skip)),
                               cols=B * 8 * V, flops=2 * B * 8 * V * M * (C1 + Cd))
            N.check(rc, "fz_upcat")
        elif pro is not None:
            raise RuntimeError("UpCatLinearFn: a block prologue needs the one-pass forward (callers check upcat_prologue_ok)")
        else:
            up = torch.empty((B, O, 2 * D, 2 * H, 2 * W), dtype=deep.dtype, device=deep.device)
            _gemm([deep], w_t, up, B=B, Cin=Cd, Vin=V, M=8 * O, K=Cd, Ncol=V, w_t=True, ldw=8 * O, bias=b_t,
                  epilogue=EPI_D2S, Ho=H, Wo=W, name="tconv_k2s2")
            _gemm([skip, up], w2, y, B=B, Cin=C1 + O, Vin=8 * V, M=M, K=C1 + O, Ncol=8 * V, bias=b_ad, c0=C1, name="cat_linear")
        ctx.save_for_backward(skip, deep, w_t, w2, b_t)
        ctx.has_bt, ctx.has_bad, ctx.wshape = b_t is not None, b_ad is not None, w_ad.shape
        ctx.npro = 0 if pro is None else 1
        if pro is not None:
            ctx.mark_non_differentiable(t_pro, st_pro)
            ctx.set_materialize_grads(False)   # (else autograd fills a full-size zero "gradient" for t: 537 MB, 67 us per step)
            return y, t_pro, st_pro
        return y

    @staticmethod
    @_bwd
    def backward(ctx, g, _gt=None, _gst=None):
        skip, deep, w_t, w2, b_t = ctx.saved_tensors
        if g is None:   # (only with materialize_grads off and an unused output: a zero gradient, as autograd would have made)
            g = torch.zeros((skip.shape[0], w2.shape[0], *skip.shape[2:]), dtype=skip.dtype, device=skip.device)
        g = g.contiguous()
        B, Cd, D, H, W = deep.shape
        O = w_t.shape[1]
        V = D * H * W
        Vf = 8 * V
        C1 = skip.shape[1]
        M = w2.shape[0]
        dev = g.device
        gw_ad = _GB.out_like(w2)
        gb_ad = torch.empty(M, dtype=torch.float32, device=dev)
        # --- skip half: input gradient + W_a weight gradient (+ Σ_v g) ---
        if M == 32 and C1 == 32 and _dw_fused_ok(M, Vf):
            g_skip, _, _, _, _ = _gemm_dw(g, w2[:, :C1], skip, want_bias=True, name="dgrad_wgrad",
                                          gw_out=gw_ad[:, :C1], gb_out=gb_ad)
        else:
            g_skip = torch.empty_like(skip)
            _gemm([g], w2, g_skip, B=B, Cin=M, Vin=Vf, M=C1, K=M, Ncol=Vf, w_t=True, ldw=C1 + O, name="linear_dgrad")
            gwa = torch.empty((M, C1), dtype=torch.float32, device=dev)
            with _no_defer():   # (the copy below reads the result inside this backward)
                _wgrad(g, [skip], gwa, B=B, M=M, Cin=C1, K=C1, Vq=Vf, Ncols=Vf, gbias=gb_ad, name="wgrad_linear")
            gw_ad[:, :C1].copy_(gwa)
        w_b = w2[:, C1:]                                       # (M, O), row stride C1 + O
        wc = torch.empty((Cd, M, 2, 2, 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            N.check(N.lib().fz_upcat_compose(w_t.data_ptr(), w_b.data_ptr(), C1 + O, None, None, wc.data_ptr(), None, None,
                                             Cd, O, M, N.stream_ptr(g)), "fz_upcat_compose")
        # --- deep half: input gradient and the (deep x g) correlation, both on g itself ---
        g_deep = None
        if ctx.needs_input_grad[1]:
            g_deep = torch.empty_like(deep)
            _gemm([g], wc, g_deep, B=B, Cin=M, Vin=Vf, M=Cd, K=8 * M, Ncol=V, loader=LOAD_S2D,
                  Di=2 * D, Hi=2 * H, Wi=2 * W, Ho=H, Wo=W, name="tconv_k2s2_dgrad")
        gt = torch.empty((Cd, M, 8), dtype=torch.float32, device=dev)
        _wgrad(deep, [g], gt, B=B, M=Cd, Cin=M, K=8 * M, Vq=Vf, Ncols=V, loader=LOAD_S2D, D=2 * D, H=2 * H,
               W=2 * W, Ho=H, Wo=W, name="wgrad_tconv_k2s2")
        gw_t = _GB.out_like(w_t)       # (its slice of the flat gradient buffer when one is attached)
        gb_t = torch.empty(O, dtype=torch.float32, device=dev) if ctx.has_bt else None
        with torch.cuda.device(dev), _finish_scope(gt, gb_ad, gw_t, gw_ad, gb_t):   # up = T(deep) + b_t: the constant part of up meets Σ_v g in dW_b
            N.check(N.lib().fz_upcat_wgrads(gt.data_ptr(), w_t.data_ptr(), w_b.data_ptr(), C1 + O, gb_ad.data_ptr(), _p(b_t),
                                            gw_t.data_ptr(), gw_ad[:, C1:].data_ptr(), C1 + O, _p(gb_t), Cd, O, M,
                                            N.stream_ptr(g)), "fz_upcat_wgrads")
        return (g_skip, g_deep, gw_t, gb_t, gw_ad.reshape(ctx.wshape), gb_ad if ctx.has_bad else None, None)[:len(ctx.needs_input_grad)]


_UPCAT_FWD = os.environ.get("FZ_UPCAT_FWD", "1") != "0"   # diagnostics: 0 = the two forward launches
_HEAD_BWD = os.environ.get("FZ_HEAD_BWD", "1") != "0"     # diagnostics: 0 = separate input- and weight-gradient launches


def up_cat_linear(skip, deep, w_t, b_t, w_ad, b_ad=None, pro=None):
    """adapter(cat([skip, ConvTranspose3d(k2, s2)(deep)], 1)) as one autograd node (UpCatLinearFn); with `pro` (see
    UpCatLinearFn.forward) returns (out, t, stats)."""
    return UpCatLinearFn.apply(skip, deep, w_t, b_t, w_ad, b_ad, pro)


_PRODUCER_PROLOGUE = os.environ.get("FZ_PRODUCER_PROLOGUE", "1") != "0"   # diagnostics: 0 = LayerNorm 1 + in_proj as its own launch


def conv3_prologue_ok(x, w):
    """would ConvK3Fn run the stem kernel in the form that can apply a block prologue?"""
    return (_PRODUCER_PROLOGUE and x.is_cuda and x.dim() == 5 and x.shape[1] * 27 * 64 * 4 <= 65536 and x.dtype in (torch.float32, torch.bfloat16)
            and bool(N.lib().fz_conv3_prologue_supported(int(x.shape[1]), int(w.shape[0]), int(x.shape[-1]), N.products())))


def upcat_prologue_ok(skip, deep, w_t, w_ad):
    """would up_cat_linear take its one-pass forward (the form that can apply a block prologue)?"""
    B, Cd, D, H, W = deep.shape
    O, C1, M = w_t.shape[1], skip.shape[1], w_ad.shape[0]
    return (_PRODUCER_PROLOGUE and M == 32 and C1 == 32 and O == 32 and _UPCAT_FWD and bool(N.lib().fz_upcat_supported(C1, Cd, D, H, W)))


class BlockPrologue:
    """with BlockPrologue(x, t, st): the FactorizerBlock whose input IS x takes t = relu(in_proj(LN1(x))) and the statistics
    from here instead of launching its first layer (the producer of x already formed them)."""
    _tls = _threading.local()

    def __init__(self, x, t, st):
        self.x, self.t, self.st = x, t, st

    def __enter__(self):
        tls = BlockPrologue._tls
        self.prev = getattr(tls, "cur", None)
        tls.cur = self
        return self

    def __exit__(self, *exc):
        BlockPrologue._tls.cur = self.prev

    @staticmethod
    def take(x):
        cur = getattr(BlockPrologue._tls, "cur", None)
        if cur is not None and cur.x is x:
            return cur.t, cur.st
        return None


def cat_linear(x1, x2, w, b=None):
    """Linear(cat([x1, x2], dim=1)) without materialising the concatenation."""
    if _native_ok(x1, x2, w, b) and x1.shape[1] % 2 == 0 and x2.shape[1] % 2 == 0:
        return CatLinearFn.apply(x1, x2, w, b)
    _warn_composed("cat+Linear", x1, x2, w, b)
    return linear_cf(torch.cat([x1, x2], dim=1), w, b)


# ---- whole FactorizerBlock as ONE autograd node ------------------------------------------------------
class FactorizerBlockFn(torch.autograd.Function):
    """x → x + out_proj(core(relu(in_proj(LN1(x))))) → (+ MLP(LN2(·)))  (factorizer.py:74-77) with a
    hand-chained backward: both residual-gradient additions are fused into the LayerNorm-backward
    kernels (`gadd`), nothing but the tensors listed in `save_for_backward` survives the forward.

    core = matricize → NMF → inverse: the fused channels-first kernels (csrc/nmf_cf.hip) when
    `cfg["core"]`, else the native modular chain swm_fwd → nmf → swm_inv."""

    _fz_acts = (0, 4, 5, 14, 15, 16, 17, 18)   # inputs that are activations (see _param_sinks_ok); every other input is a parameter

    @staticmethod
    @N.capture_products
    def forward(ctx, x, n1w, n1b, win, u0, v0, wout, bout, n2w, n2b, w1, b1, w2, b2, cfg, head_w=None, head_b=None,
                t_pre=None, st_pre=None):
        # t_pre / st_pre: t = relu(in_proj(LN1(x))) and the LayerNorm statistics already formed by the launch that produced x
        # (BlockPrologue): step 1 is skipped; constants here — the backward below differentiates through them anyway
        # head_w (M <= 4, 32[, 1...]) / head_b: the network's head, applied to the block output inside the block's last launch;
        # the node then returns (x2, logits) with the logits NOT differentiable here — HeadOfBlockFn carries their graph
        x = x.contiguous()
        B, C = x.shape[:2]
        V = _vox(x)
        sp = x.shape[2:]
        geo, T, G, solver, neps = cfg["geo"], cfg["T"], cfg["G"], cfg["solver"], cfg["nmf_eps"]
        win2, wout2 = win.reshape(C, C), wout.reshape(C, C)
        Hd = w1.shape[0]
        w12, w22 = w1.reshape(Hd, C), w2.reshape(C, Hd)
        u0c, v0c = u0.contiguous(), v0.contiguous()
        new = lambda ch: torch.empty((B, ch, *sp), dtype=x.dtype, device=x.device)  # noqa: E731
        # 1. t = relu(in_proj(LN1(x)))
        if t_pre is not None:
            t, st1 = t_pre, st_pre
        else:
            t = new(C)
            st1 = torch.empty((B, 2, V), dtype=torch.float32, device=x.device)
            _gemm([x], win2, t, B=B, Cin=C, Vin=V, M=C, K=C, Ncol=V, ln=(n1w, n1b, cfg["eps1"]), stats_out=st1,
                  eact=ACT["relu"], name="ln_linear")
        # 2. a = inverse(NMF(matricize(t)))
        m = None
        if cfg["core"]:
            a = Fn.FactCoreFn.forward(_Ctx(), t, u0c, v0c, geo, T, G, solver, neps, True)
        else:
            m = Fn._swm_fwd_raw(t, geo)
            ym, _, _ = Fn._nmf_fwd_raw(m, u0c, v0c, T, solver, neps)
            a = Fn._swm_inv_raw(ym, geo, average=True)
            del ym
        # 3. x1 = x + out_proj(a)   4. z1 = fc1(LN2(x1)) ; x2 = x1 + fc2(gelu(z1))
        logits = None
        if _outproj_mlp_ok(C, Hd, V) and a.is_contiguous() and x.is_contiguous():
            if head_w is not None:
                x1, x2, z1, st2, logits = _outproj_mlp_fwd_chain(a, wout2, bout, x, n2w, n2b, cfg["eps2"], w12, b1, w22, b2,
                                                                 head=(head_w.reshape(head_w.shape[0], C), head_b))
            else:
                x1, x2, z1, st2 = _outproj_mlp_fwd_chain(a, wout2, bout, x, n2w, n2b, cfg["eps2"], w12, b1, w22, b2)   # both in one launch
        else:
            x1 = new(C)
            _gemm([a], wout2, x1, B=B, Cin=C, Vin=V, M=C, K=C, Ncol=V, bias=bout, res=x, name="act_linear_res")
            if _mlp_chain_ok(C, Hd, V):
                x2, z1, st2 = _mlp_fwd_chain(x1, n2w, n2b, cfg["eps2"], w12, b1, w22, b2)
            else:
                z1 = new(Hd)
                st2 = torch.empty((B, 2, V), dtype=torch.float32, device=x.device)
                _gemm([x1], w12, z1, B=B, Cin=C, Vin=V, M=Hd, K=C, Ncol=V, bias=b1, ln=(n2w, n2b, cfg["eps2"]),
                      stats_out=st2, name="ln_linear")
                x2 = new(C)
                _gemm([z1], w22, x2, B=B, Cin=Hd, Vin=V, M=C, K=Hd, Ncol=V, bias=b2, bact=ACT["gelu"], res=x1,
                      name="act_linear_res")
        ctx.save_for_backward(x, st1, t, a, x1, st2, z1, m, n1w, n1b, win2, u0c, v0c, wout2, n2w, n2b, w12, w22)
        ctx.cfg = cfg
        ctx.shapes = (win.shape, wout.shape, w1.shape, w2.shape)
        ctx.prm = tuple(weakref.ref(t) for t in (win, wout, bout, w1, b1, w2, b2))
        if head_w is not None:
            if logits is None:
                raise RuntimeError("FactorizerBlockFn: head fusion asked for a configuration outside fz_mlp_pre_supported "
                                   "(callers check block_head_fusable first)")
            ctx.mark_non_differentiable(logits)
            ctx.set_materialize_grads(False)   # (no zero-filled "gradient" of the logits)
            return x2, logits
        return x2

    @staticmethod
    @_bwd
    def backward(ctx, g2, _g_logits=None):
        x, st1, t, a, x1, st2, z1, m, n1w, n1b, win2, u0c, v0c, wout2, n2w, n2b, w12, w22 = ctx.saved_tensors
        cfg = ctx.cfg
        if g2 is None:
            g2 = torch.zeros_like(x)
        g2 = g2.contiguous()
        B, C = x.shape[:2]
        V = _vox(x)
        Hd = w12.shape[0]
        dev, dt = x.device, x.dtype
        geo, T, G, solver, neps = cfg["geo"], cfg["T"], cfg["G"], cfg["solver"], cfg["nmf_eps"]
        cur = torch.cuda.current_stream(dev)
        side = _side_stream(dev, B * V)
        keep = []  # tensors read on the side stream stay referenced until the streams are joined

        pending = []   # single stream: the block's weight-gradient problems go out as ONE grid at the end (fz_wgrad_group)

        def wgrad(*args, **kw):
            if _WGRAD_GROUP:   # (with a side stream too [r6]: the ONE grouped launch goes there, at the end of the block)
                kw.pop("name", None)
                pending.append((args[0], args[1], args[2], kw))   # (keeps gz1 alive until the launch)
                return args[2]
            if side is None:
                return _wgrad(*args, **kw)
            side.wait_stream(cur)
            keep.extend(t for t in (args[0], *args[1], kw.get("stats")) if t is not None)
            with torch.cuda.stream(side):   # (deferred finishes queue per stream since round 6: csrc/finish.hip)
                return _wgrad(*args, **kw)

        # --- MLP ---
        chain = _mlp_chain_ok(C, Hd, V)
        if _mlp_wgrad_fused_ok(C, Hd, V):
            # input-gradient chain + both weight gradients in one pass over (g2, z1, x1)
            gx1, gg2, gbt2, gw1, gb1, gw2, gb2 = _mlp_bwd_chain_wgrad(g2, z1, w12, w22, x1, st2, n2w, n2b)
        else:
            if chain:
                gz1, gx1, gg2, gbt2 = _mlp_bwd_chain(g2, z1, w12, w22, x1, st2, n2w)   # + residual path of the MLP
            else:
                gz1 = torch.empty_like(z1)
                _gemm([g2], w22, gz1, B=B, Cin=C, Vin=V, M=Hd, K=C, Ncol=V, w_t=True, ldw=Hd, emul=z1,
                      emul_kind=ACT["gelu"], name="linear_dgrad")
            gw2 = _GB.out_like(w22)
            gb2 = torch.empty(C, dtype=torch.float32, device=dev)
            wgrad(g2, [z1], gw2, B=B, M=C, Cin=Hd, K=Hd, Vq=V, Ncols=V, gbias=gb2, qact=ACT["gelu"], name="wgrad_linear")
            if not chain:
                gx1, gg2, gbt2 = _dgrad_lnbwd(gz1, w12, x1, st2, n2w, g2)      # + residual path of the MLP
            gw1 = _GB.out_like(w12)
            gb1 = torch.empty(Hd, dtype=torch.float32, device=dev)
            wgrad(gz1, [x1], gw1, B=B, M=Hd, Cin=C, K=C, Vq=V, Ncols=V, gbias=gb1, stats=st2, ln=(n2w, n2b),
                   name="wgrad_ln_linear")
            del gz1
        # --- out_proj ---
        dw = _dw_fused_ok(C, V)
        if dw:
            ga, gwo, gbo, _, _ = _gemm_dw(gx1, wout2, a, want_bias=True, name="dgrad_wgrad")
        else:
            ga = torch.empty_like(a)
            _gemm([gx1], wout2, ga, B=B, Cin=C, Vin=V, M=C, K=C, Ncol=V, w_t=True, ldw=C, name="linear_dgrad")
            gwo = _GB.out_like(wout2)
            gbo = torch.empty(C, dtype=torch.float32, device=dev)
            wgrad(gx1, [a], gwo, B=B, M=C, Cin=C, K=C, Vq=V, Ncols=V, gbias=gbo, name="wgrad_linear")
        # --- core (gradient arrives gated by [t > 0]) ---
        if G <= 0:
            gt = torch.zeros_like(t)
        elif cfg["core"]:
            c2 = _Ctx()
            c2.saved_tensors = (t, u0c, v0c)
            c2.cfg = (geo, T, G, solver, neps, True)
            gt = Fn.FactCoreFn.backward(c2, ga)[0]
        else:
            gym = Fn._swm_fwd_raw(ga, geo, div=geo.nshift)
            gm = Fn._nmf_bwd_raw(m, u0c, v0c, gym, None, None, T, G, solver, neps)
            del gym
            gt = Fn._swm_inv_raw(gm, geo, average=False, gate=m)
            del gm
        del ga
        # --- in_proj + LN1 ---
        if dw:
            gx, gwi, _, gg1, gbt1 = _gemm_dw(gt, win2, x, ln=(n1w, n1b), stats=st1, gadd=gx1, name="dgrad_lnbwd_wgrad")
        else:
            gx, gg1, gbt1 = _dgrad_lnbwd(gt, win2, x, st1, n1w, gx1)        # + residual path of the mixer
            gwi = _GB.out_like(win2)
            wgrad(gt, [x], gwi, B=B, M=C, Cin=C, K=C, Vq=V, Ncols=V, stats=st1, ln=(n1w, n1b), name="wgrad_ln_linear")
        if pending:
            if side is None:
                _wgrad_group(pending, f"wgrad_block_{C}x{Hd}")
            else:
                # the block's four weight gradients as one grid on the second stream: nothing in the backward reads them, and a
                # launch of 64-512 workgroups of a deep stage leaves most of the chip to the next block's kernels on the main stream
                side.wait_stream(cur)
                for pr in pending:
                    keep.extend(t for t in (pr[0], *pr[1], pr[3].get("stats"), pr[3].get("pmul")) if t is not None)
                with torch.cuda.stream(side):
                    _wgrad_group(pending, f"wgrad_block_{C}x{Hd}")
            pending.clear()
        if side is not None:
            if _LateJoin.enabled:
                _LateJoin.keep.extend(keep)
                _LateJoin.devices.add((dev.type, dev.index))
                _LateJoin.owed.extend(zip(ctx.prm, (t.data_ptr() for t in (gwi, gwo, gbo, gw1, gb1, gw2, gb2))))
                if not _LateJoin.queued:
                    # the streams always meet (and ownership is verified) before backward() returns: any
                    # reader of p.grad after backward — a stock optimizer, gradient clipping — is safe
                    _LateJoin.queued = True
                    torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
            else:
                cur.wait_stream(side)
            keep.clear()
        s_in, s_out, s_1, s_2 = ctx.shapes
        return (gx, gg1, gbt1, gwi.reshape(s_in), None, None, gwo.reshape(s_out), gbo, gg2, gbt2,
                gw1.reshape(s_1), gb1, gw2.reshape(s_2), gb2, None, None, None, None, None)[:len(ctx.needs_input_grad)]


_head_fusion = _threading.local()


class HeadFusion:
    """with HeadFusion(block, (w, b)) as slot: run the network; `block` (the module whose output feeds only the head) leaves
    the logits in slot.logits if it applied the head itself, else None (the caller then runs the head module)."""

    def __init__(self, block, head_params):
        self.block, self.head_params, self.logits = block, head_params, None

    def __enter__(self):
        self.prev = getattr(_head_fusion, "slot", None)
        _head_fusion.slot = self
        return self

    def __exit__(self, *exc):
        _head_fusion.slot = self.prev


def head_fusion_slot():
    return getattr(_head_fusion, "slot", None)


def block_head_fusable(C, Hd, V, head_w, x):
    """may FactorizerBlockFn apply the head Linear(C -> M <= 4) inside its last launch? (csrc/gemm.hip: fz_mlp_chain post_*)"""
    return (C == 32 and _outproj_mlp_ok(C, Hd, V) and head_w is not None and head_w.dtype == torch.float32 and head_w.numel() == head_w.shape[0] * C
            and 1 <= head_w.shape[0] <= 4 and x.is_cuda and _HEAD_BWD)


class HeadOfBlockFn(torch.autograd.Function):
    """logits = head(y) (unet.py:253,274) where the logits were ALREADY produced by the launch that produced y
    (FactorizerBlockFn with head_w): forward hands them out, backward is the head's own one-pass gradient kernel
    (csrc/headbwd.hip: input, weight and bias gradient)."""

    _fz_acts = (0, 3)   # inputs that are activations (see _param_sinks_ok); every other input is a parameter

    @staticmethod
    @N.capture_products
    def forward(ctx, y, w, b, logits):
        ctx.save_for_backward(y, w)
        ctx.has_bias, ctx.wshape = b is not None, w.shape
        return logits.view_as(logits)

    @staticmethod
    @_bwd
    def backward(ctx, gl):
        y, w = ctx.saved_tensors
        B, C = y.shape[:2]
        V = _vox(y)
        M = w.shape[0]
        w2 = w.reshape(M, C)
        gl = gl.contiguous()
        if gl.dtype != y.dtype:
            gl = gl.to(y.dtype)
        gy = torch.empty_like(y)
        lib = N.lib()
        rows = lib.fz_head_bwd_rows()
        part = torch.empty(lib.fz_head_bwd_workspace_bytes() // 4, dtype=torch.float32, device=y.device)
        es = y.element_size()
        with torch.cuda.device(y.device):
            rc = Fn._timed(f"head_bwd_{C}->{M}", es * (gl.numel() + 2 * y.numel()),
                           lambda: lib.fz_head_bwd(gl.data_ptr(), y.data_ptr(), w2.data_ptr(), gy.data_ptr(), part.data_ptr(),
                                                   B, M, C, V, N.act_dtype(y), N.stream_ptr(y)), cols=B * V)
            N.check(rc, "fz_head_bwd")
            gw, gb = _head_rows_reduce(part, rows, w2, M, y)
        return gy, gw.reshape(ctx.wshape), (gb if ctx.has_bias else None), None


class _Ctx:
    """Minimal stand-in for an autograd ctx when a Function's forward/backward is called directly."""

    saved_tensors = ()

    def save_for_backward(self, *ts):
        self.saved_tensors = ts
