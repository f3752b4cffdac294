"""-m gpu: deferred finishes (csrc/finish.h, include/factorizer_hip.h: fz_finish_defer / fz_finish_pending / fz_finish_flush;
factorizer_amd/pointwise.py: _Defer).  The fixed-order reductions that end every weight-gradient launch are queued during a
backward that an owner of the step armed (FlatAdamW / FlatGradSync zero_grad) and run as one or two grids at its end.  The
sums and their order are those of the immediate launches: every gradient must be BIT-identical, deferred or not; the second
backward without zero_grad (autograd then adds to .grad while the backward runs) must not be deferred at all."""
import ctypes

import pytest
import torch
from torch import nn

import factorizer_amd as ft
from factorizer_amd import _native as N
from factorizer_amd import pointwise as PW
from factorizer_amd.training import FlatAdamW

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(S, widths, strides, patch):
    return ft.Factorizer(in_channels=4, out_channels=3, spatial_size=S, encoder_depth=(1,) * len(widths), encoder_width=widths,
                         strides=strides, decoder_depth=(1,) * (len(widths) - 1), norm=ft.LayerNorm,
                         reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": patch}), act=nn.ReLU, factorize=ft.NMF, rank=1,
                         num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)


def _grads(model, opt, x, t, backwards=1, amp=False):
    opt.zero_grad()
    for _ in range(backwards):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            loss = ft.dice_ce_loss(model(x), t)
        loss.backward()
    torch.cuda.synchronize()
    return [p.grad.detach().clone() for p in model.parameters()]


@pytest.mark.parametrize("S,widths,strides,patch,B,amp", [
    ((64, 64, 64), (32, 64, 128, 256), (1, 2, 2, 2), 8, 2, False),     # fused C = 32 forms, C = 64 chain, deep generic blocks
    ((64, 64, 64), (32, 64, 128, 256), (1, 2, 2, 2), 8, 2, True),      # ... bf16 storage
    ((32, 32, 32), (32, 64, 128), (1, 2, 2), 4, 2, False)])            # generic-patch forms
def test_deferred_finishes_are_bit_identical_and_all_run(S, widths, strides, patch, B, amp):
    torch.manual_seed(0)
    model = _model(S, widths, strides, patch).to(DEV)
    x = torch.rand(B, 4, *S, device=DEV)
    t = (torch.rand(B, 3, *S, device=DEV) > 0.5).float()
    opt = FlatAdamW(model, lr=1e-3, deferred_finishes=True)
    try:
        assert PW._Defer.enabled
        f0, n0 = PW._Defer.flushed, PW._Defer.flushes
        l0 = N.lib().fz_launch_count()
        g_def = _grads(model, opt, x, t, amp=amp)
        l_def = N.lib().fz_launch_count() - l0
        queued, flushes = PW._Defer.flushed - f0, PW._Defer.flushes - n0
        assert queued >= 20 and flushes == 1, (queued, flushes)          # one flush at the end of the backward
        assert N.lib().fz_finish_pending() == 0 and not PW._Defer.armed and not PW._Defer.keep
        PW.defer_finishes(False)
        l0 = N.lib().fz_launch_count()
        g_imm = _grads(model, opt, x, t, amp=amp)
        l_imm = N.lib().fz_launch_count() - l0
        assert PW._Defer.flushed - f0 == queued                            # nothing was queued this time
        assert l_imm - l_def >= 20, (l_imm, l_def, queued)                 # launches fewer (grouped finishes were one launch for up to 4 jobs)
        for a, b in zip(g_def, g_imm):
            assert torch.equal(a, b)
        # gradient accumulation: the second backward of a step is never deferred, and the sums are those of two immediate ones
        PW.defer_finishes(True)
        f1 = PW._Defer.flushed
        g2_def = _grads(model, opt, x, t, backwards=2, amp=amp)
        assert 0 < PW._Defer.flushed - f1 <= queued
        PW.defer_finishes(False)
        g2_imm = _grads(model, opt, x, t, backwards=2, amp=amp)
        for a, b in zip(g2_def, g2_imm):
            assert torch.equal(a, b)
    finally:
        PW.defer_finishes(False)


def test_queue_semantics_through_the_c_abi():
    lib = N.lib()
    torch.manual_seed(1)
    rows, n = 700, 96                     # > 512 rows: the two-stage reduction (a phase-0 and a phase-1 job)
    part = torch.randn(rows, n, device=DEV)
    tmp = torch.empty(64, n, device=DEV)
    ref = torch.empty(n, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    N.check(lib.fz_reduce_rows(part.data_ptr(), rows, n, ref.data_ptr(), tmp.data_ptr(), st), "fz_reduce_rows")
    torch.cuda.synchronize()
    assert torch.allclose(ref, part.sum(0), rtol=1e-4, atol=1e-4)
    out = torch.full((n,), -7.0, device=DEV)
    other = torch.cuda.Stream()
    assert lib.fz_finish_defer(1) == 0
    try:
        N.check(lib.fz_reduce_rows(part.data_ptr(), rows, n, out.data_ptr(), tmp.data_ptr(), st), "fz_reduce_rows")
        assert lib.fz_finish_pending() == 2 and lib.fz_finish_defer(-1) == 1
        torch.cuda.synchronize()
        assert bool((out == -7.0).all())                                   # nothing ran yet
        assert lib.fz_finish_flush(other.cuda_stream) == 0                 # nothing is queued for THAT stream
        assert lib.fz_finish_pending() == 2
        # a job that accumulates drains the queue first and runs at once
        acc = torch.ones(n, device=DEV)
        N.check(lib.fz_chunk_reduce(part.data_ptr(), rows, n, acc.data_ptr(), 1, st), "fz_chunk_reduce")
        assert lib.fz_finish_pending() == 0
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        assert torch.allclose(acc, 1.0 + part.sum(0), rtol=1e-4, atol=1e-4)
        out.fill_(-7.0)
        N.check(lib.fz_reduce_rows(part.data_ptr(), rows, n, out.data_ptr(), tmp.data_ptr(), st), "fz_reduce_rows")
        assert lib.fz_finish_flush(st) == 2 and lib.fz_finish_pending() == 0 and lib.fz_finish_flush(st) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        # one queue per stream; fz_finish_flush_all runs each on its own stream and orders the waiter behind them
        out.fill_(-7.0)
        out2 = torch.full((n,), -7.0, device=DEV)
        tmp2 = torch.empty(64, n, device=DEV)
        other.wait_stream(torch.cuda.current_stream())
        N.check(lib.fz_reduce_rows(part.data_ptr(), rows, n, out.data_ptr(), tmp.data_ptr(), st), "fz_reduce_rows")
        N.check(lib.fz_reduce_rows(part.data_ptr(), rows, n, out2.data_ptr(), tmp2.data_ptr(), other.cuda_stream), "fz_reduce_rows")
        assert lib.fz_finish_pending() == 4
        assert lib.fz_finish_flush_all(st) == 4 and lib.fz_finish_pending() == 0
        both = out + out2            # on the current stream: ordered behind `other` by the flush itself
        torch.cuda.synchronize()
        assert torch.equal(both, ref + ref)
        # the flag is per thread: another thread's calls stay immediate while this one defers
        import threading
        seen = {}

        def worker():
            seen["flag"] = lib.fz_finish_defer(-1)
            o3 = torch.full((n,), -7.0, device=DEV)
            t3 = torch.empty(64, n, device=DEV)
            N.check(lib.fz_reduce_rows(part.data_ptr(), rows, n, o3.data_ptr(), t3.data_ptr(), torch.cuda.current_stream().cuda_stream), "fz_reduce_rows")
            seen["pending"] = lib.fz_finish_pending()
            torch.cuda.synchronize()
            seen["equal"] = torch.equal(o3, ref)
        th = threading.Thread(target=worker)
        th.start()
        th.join()
        assert seen == {"flag": 0, "pending": 0, "equal": True}, seen
    finally:
        assert lib.fz_finish_defer(0) == 1
    # column blocks of wider rows (the head's 132-float partial rows)
    wide = torch.randn(40, 132, device=DEV)
    gw, gb = torch.empty(96, device=DEV), torch.empty(3, device=DEV)
    N.check(lib.fz_chunk_reduce_ld(wide.data_ptr(), 40, 96, 132, gw.data_ptr(), 0, st), "fz_chunk_reduce_ld")
    N.check(lib.fz_chunk_reduce_ld(wide.data_ptr() + 128 * 4, 40, 3, 132, gb.data_ptr(), 0, st), "fz_chunk_reduce_ld")
    whole = torch.empty(132, device=DEV)
    N.check(lib.fz_chunk_reduce(wide.data_ptr(), 40, 132, whole.data_ptr(), 0, st), "fz_chunk_reduce")
    torch.cuda.synchronize()
    assert torch.equal(gw, whole[:96]) and torch.equal(gb, whole[128:131])
    assert lib.fz_chunk_reduce_ld(wide.data_ptr(), 40, 96, 64, gw.data_ptr(), 0, st) == -4
    assert ctypes.c_int(lib.fz_finish_pending()).value == 0


def _tiny(in_channels=4):
    return ft.Factorizer(in_channels=in_channels, out_channels=3, spatial_size=(32, 32, 32), encoder_depth=(1, 1), encoder_width=(32, 64),
                         strides=(1, 2), decoder_depth=(1,), norm=ft.LayerNorm,
                         reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}), act=nn.ReLU, factorize=ft.NMF, rank=1,
                         num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)


def _deferred_vs_immediate(model, opt, step):
    """gradients of `step()` with deferral armed by opt.zero_grad() vs with deferral switched off: (deferred, immediate, #queued, #refused)"""
    PW.defer_finishes(True)
    f0, r0 = PW._Defer.flushed, PW._Defer.refused
    opt.zero_grad()
    step()
    torch.cuda.synchronize()
    g_def = [None if p.grad is None else p.grad.detach().clone() for p in model.parameters()]
    queued, refused = PW._Defer.flushed - f0, PW._Defer.refused - r0
    PW.defer_finishes(False)
    opt.zero_grad()
    step()
    torch.cuda.synchronize()
    g_imm = [None if p.grad is None else p.grad.detach().clone() for p in model.parameters()]
    return g_def, g_imm, queued, refused


@pytest.mark.parametrize("cin", [1, 3])
def test_odd_input_channels_stem_weight_is_not_a_leaf(cin):
    """Conv3d with an odd C_in hands ConvK3Fn a PADDED weight — a non-leaf whose gradient autograd slices
    (and clones) inside the backward.  That node must run its finishes at once; every gradient equals the immediate form."""
    torch.manual_seed(0)
    model = _tiny(cin).to(DEV)
    x = torch.rand(2, cin, 32, 32, 32, device=DEV)
    t = (torch.rand(2, 3, 32, 32, 32, device=DEV) > 0.5).float()
    opt = FlatAdamW(model, lr=1e-3, deferred_finishes=True)
    try:
        g_def, g_imm, queued, refused = _deferred_vs_immediate(model, opt, lambda: ft.dice_ce_loss(model(x), t).backward())
        assert queued > 0 and refused >= 1, (queued, refused)      # the rest of the network still defers; the stem does not
        for (name, _), a, b in zip(model.named_parameters(), g_def, g_imm):
            assert torch.equal(a, b), name
        # and the values are right, not merely equal: against ATen's own gradient of the same convolution
        stem = model.stem if hasattr(model, "stem") else next(m for m in model.modules() if isinstance(m, nn.Conv3d) and m.kernel_size == (3, 3, 3))
        xs = torch.rand(2, cin, 32, 32, 32, device=DEV)
        gy = torch.rand(2, stem.out_channels, 32, 32, 32, device=DEV)
        opt.zero_grad()
        PW.defer_finishes(True)
        opt.zero_grad()
        stem(xs).backward(gy)
        torch.cuda.synchronize()
        ref_w = torch.nn.grad.conv3d_weight(xs, stem.weight.shape, gy, padding=1)
        assert (stem.weight.grad - ref_w).abs().max() <= 1e-4 * ref_w.abs().max()
    finally:
        PW.defer_finishes(False)


def test_deferral_is_refused_where_autograd_reads_the_gradient_early():
    """a parameter used twice in one graph, a tensor hook on a parameter, a model the arming owner does not
    cover, create_graph=True — each must fall back to immediate finishes for the nodes concerned and give the immediate
    gradients bit for bit."""
    torch.manual_seed(0)
    model = _tiny().to(DEV)
    x = torch.rand(2, 4, 32, 32, 32, device=DEV)
    t = (torch.rand(2, 3, 32, 32, 32, device=DEV) > 0.5).float()
    opt = FlatAdamW(model, lr=1e-3, deferred_finishes=True)
    try:
        # (1) every parameter used twice: two forwards in one graph
        g_def, g_imm, queued, refused = _deferred_vs_immediate(
            model, opt, lambda: (ft.dice_ce_loss(model(x), t) + ft.dice_ce_loss(model(x * 0.5), t)).backward())
        assert refused >= 1
        for (name, _), a, b in zip(model.named_parameters(), g_def, g_imm):
            assert torch.equal(a, b), name
        # (2) a tensor hook that reads the gradient inside the backward
        p0 = next(p for n, p in model.named_parameters() if n.endswith("out_proj.linear.weight"))
        got = []
        h = p0.register_hook(lambda g: got.append(g.detach().clone()))
        g_def, g_imm, queued, refused = _deferred_vs_immediate(model, opt, lambda: ft.dice_ce_loss(model(x), t).backward())
        h.remove()
        assert refused >= 1 and len(got) == 2 and torch.equal(got[0], got[1]) and torch.equal(got[0], p0.grad)
        for (name, _), a, b in zip(model.named_parameters(), g_def, g_imm):
            assert torch.equal(a, b), name
        # (3) a second model in the process that the arming owner does not cover: never deferred
        other = _tiny().to(DEV)
        PW.defer_finishes(True)
        opt.zero_grad()                      # arms deferral — for `model`'s parameters
        f0 = PW._Defer.flushed
        ft.dice_ce_loss(other(x), t).backward()
        torch.cuda.synchronize()
        assert PW._Defer.flushed == f0 and N.lib().fz_finish_pending() == 0
        g_other = [p.grad.detach().clone() for p in other.parameters()]
        PW.defer_finishes(False)
        for p in other.parameters():
            p.grad = None
        ft.dice_ce_loss(other(x), t).backward()
        torch.cuda.synchronize()
        for a, p in zip(g_other, other.parameters()):
            assert torch.equal(a, p.grad)
    finally:
        PW.defer_finishes(False)


def test_flush_follows_the_stream_the_backward_ran_on():
    """forward under `with torch.cuda.stream(s)`, backward() called outside it — autograd runs the nodes on s,
    the finishes queue on s, the end-of-backward flush runs them THERE and the caller's stream waits."""
    torch.manual_seed(0)
    model = _tiny().to(DEV)
    x = torch.rand(2, 4, 32, 32, 32, device=DEV)
    t = (torch.rand(2, 3, 32, 32, 32, device=DEV) > 0.5).float()
    opt = FlatAdamW(model, lr=1e-3, deferred_finishes=True)
    try:
        s = torch.cuda.Stream()
        opt.zero_grad()
        s.wait_stream(torch.cuda.current_stream())
        f0 = PW._Defer.flushed
        with torch.cuda.stream(s):
            loss = ft.dice_ce_loss(model(x), t)
        loss.backward()
        g_def = [p.grad.detach().clone() for p in model.parameters()]     # on the current stream: ordered behind s by the flush
        torch.cuda.synchronize()
        assert PW._Defer.flushed - f0 > 0 and N.lib().fz_finish_pending() == 0
        PW.defer_finishes(False)
        opt.zero_grad()
        ft.dice_ce_loss(model(x), t).backward()
        torch.cuda.synchronize()
        for a, p in zip(g_def, model.parameters()):
            assert torch.equal(a, p.grad)
    finally:
        PW.defer_finishes(False)
