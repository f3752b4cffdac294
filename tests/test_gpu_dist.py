"""-m gpu: the data-parallel gradient path on the device with RCCL actually executing (process group "nccl" of ONE rank —
the GPU boxes of this pool have one GPU; world 2 runs on gloo in tests/test_distributed_cpu.py).

What is checked here and nowhere else: with `overlap=True` the weight-gradient kernels write INTO the bucket
memory (gradbuf.py), autograd adopts those views as p.grad while post-accumulate hooks are registered, a bucket's
all-reduce is launched from the hook of its last parameter on RCCL's own stream — i.e. RCCL kernels run beside the rest of
the backward — and `finish()` hands the reduced views to the optimizer.  The gradients after `finish()` must equal the
unattached run's bit for bit (SUM over one rank = identity), the matrix weights must alias the flat buffer (no packing
copy), and replays must be bitwise identical although another queue is busy during the backward (the co-residency
condition of profiles/r03_two_stream_interaction.md).  Replaces DistributedDataParallel of
model_zoo/factorizer_brats23/configs/train_multigpu.yaml:3-6."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
from torch import nn

import factorizer_amd as ft
from factorizer_amd.parallel import FlatGradSync

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture()
def rccl_world1():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert not dist.is_initialized()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        yield
    finally:
        dist.destroy_process_group()


def _model(S, widths, strides, patch):
    return ft.Factorizer(in_channels=4, out_channels=3, spatial_size=S, encoder_depth=(1,) * len(widths), encoder_width=widths,
                         strides=strides, decoder_depth=(1,) * (len(widths) - 1), norm=ft.LayerNorm,
                         reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": patch}), act=nn.ReLU, factorize=ft.NMF, rank=1,
                         num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)


@pytest.mark.parametrize("S,widths,strides,patch,B,amp", [
    ((128, 128, 128), (32, 64, 128, 256, 512), (1, 2, 2, 2, 2), 8, 1, False),   # the README model: every fused backward form
    ((128, 128, 128), (32, 64, 128, 256, 512), (1, 2, 2, 2, 2), 8, 1, True),    # ... under bf16 autocast
    ((32, 32, 32), (32, 64, 128), (1, 2, 2), 4, 2, False)])                      # small: the unfused / generic-patch forms
def test_overlapped_allreduce_gradients_equal_unattached_run(rccl_world1, S, widths, strides, patch, B, amp):
    torch.manual_seed(0)
    model = _model(S, widths, strides, patch).to(DEV)
    x = torch.rand(B, 4, *S, device=DEV)
    t = (torch.rand(B, 3, *S, device=DEV) > 0.5).float()

    def loss_of():
        if amp:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return ft.dice_ce_loss(model(x), t)
        return ft.dice_ce_loss(model(x), t)

    # unattached: plain autograd, no flat buffer, no hooks, no collective
    model.zero_grad(set_to_none=True)
    loss_of().backward()
    ref = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)

    sync = FlatGradSync(model, num_buckets=4, overlap=True, force_collectives=True)
    assert sync.active and sync.overlap and len(sync.buckets) == 4
    opt = ft.FlatAdamW(model, lr=1e-4, weight_decay=1e-5, flat_grad=sync.flat, grad_views=sync.views)
    for rep in range(3):
        sync.zero_grad()
        loss_of().backward()
        assert all(sync._launched), "every bucket must have been launched from a hook, during the backward"
        # produced in place: autograd adopted the kernel's output — a view of the bucket memory — as p.grad (no packing copy)
        for n, p in model.named_parameters():
            if n.endswith("linear.weight"):
                assert p.grad.data_ptr() == sync.views[p].data_ptr(), ("not produced in the bucket memory", n)
        scale = sync.finish(average=False)
        assert scale == 1.0
        torch.cuda.synchronize()
        in_place = 0
        for n, p in model.named_parameters():
            assert p.grad.data_ptr() == sync.views[p].data_ptr(), n
            assert torch.equal(p.grad, ref[n]), (rep, n, float((p.grad - ref[n]).abs().max()))
            in_place += p.dim() >= 2
        assert in_place > 0
    # the optimizer consumes the same buffer
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    opt.step(grad_scale=scale)
    moved = sum(1 for n, p in model.named_parameters() if not torch.equal(p.detach(), before[n]))
    assert moved >= len(before) - 2


def test_bucket_allreduce_is_ordered_after_the_queued_weight_gradient_kernels(rccl_world1, monkeypatch):
    """A bucket's all-reduce is launched from the post-accumulate hook of its last parameter — on the HOST, while the
    weight-gradient kernels that write INTO that bucket may still be queued on the compute stream (the host runs milliseconds
    ahead of the device).  Correct only if RCCL's stream waits for the compute stream's queue as of the call.  One rank cannot
    show a violation in the DATA (an in-place all-reduce over one rank leaves whatever the later kernel writes), so the test
    makes the ordering observable in TIME: right before every all-reduce of the backward it queues a long spin kernel on the
    compute stream and then asks the returned work handle, from the host, whether the collective has completed while that spin
    is provably still running (its end event has not fired).  A collective that did not wait for the compute stream completes
    within microseconds and would be seen completed.  Also poisons the flat buffer before the backward: no NaN may survive.
    Replaces what DistributedDataParallel guarantees for model_zoo/factorizer_brats23/configs/train_multigpu.yaml:3-6."""
    import time
    torch.manual_seed(0)
    S, widths, strides = (32, 32, 32), (32, 64, 128), (1, 2, 2)
    model = _model(S, widths, strides, 4).to(DEV)
    x = torch.rand(2, 4, *S, device=DEV)
    t = (torch.rand(2, 3, *S, device=DEV) > 0.5).float()
    sync = FlatGradSync(model, num_buckets=4, overlap=True, force_collectives=True)
    real = dist.all_reduce
    seen = []

    def spying_all_reduce(tensor, *a, **kw):
        torch.cuda._sleep(400_000_000)                     # ~0.2 s of spinning on the compute stream, queued BEFORE the collective
        done = torch.cuda.Event()
        done.record()
        work = real(tensor, *a, **kw)
        time.sleep(0.02)                                    # two orders of magnitude more than an unordered collective needs
        still_spinning = not done.query()
        seen.append((still_spinning, bool(work.is_completed())))
        return work

    monkeypatch.setattr(dist, "all_reduce", spying_all_reduce)
    sync.zero_grad()
    sync.flat.fill_(float("nan"))                           # every element must be overwritten by a kernel or the packing copy
    ft.dice_ce_loss(model(x), t).backward()
    assert all(sync._launched) and len(seen) == 4
    sync.finish(average=False)
    torch.cuda.synchronize()
    for spinning, completed in seen:
        assert spinning, "the spin kernel had already ended: the probe measured nothing (raise the cycle count)"
        assert not completed, "an all-reduce completed while earlier compute-stream work was still running: not ordered after it"
    for n, p in model.named_parameters():                   # (the alignment padding between slices belongs to no gradient)
        assert torch.isfinite(sync.views[p]).all(), ("poison survived: reduced before (or without) being produced", n)
