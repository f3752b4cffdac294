"""world_size-2 gloo test of the data-parallel gradient path (factorizer_amd/parallel.py):
after backward + finish(), every rank holds the average of the per-rank gradients, and one
optimizer step keeps the replicas bit-identical."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, overlap, out, flat_opt=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import factorizer_amd as ft
        from factorizer_amd.parallel import FlatGradSync
        torch.manual_seed(100 + rank)  # different init per rank on purpose: broadcast must fix it
        model = ft.Factorizer(in_channels=2, out_channels=2, spatial_size=(8, 8, 8), encoder_depth=(1, 1),
                              encoder_width=(8, 16), strides=(1, 2), decoder_depth=(1,),
                              reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}), act=nn.ReLU,
                              factorize=ft.NMF, rank=1, num_iters=3, solver="hals", mlp_ratio=2, dropout=0.0)
        sync = FlatGradSync(model, num_buckets=3, overlap=overlap, late_wgrad_join=overlap)
        sync.broadcast_state(0)
        opt = (ft.FlatAdamW(model, lr=0.01, weight_decay=1e-5, flat_grad=sync.flat, grad_views=sync.views)
               if flat_opt else torch.optim.SGD(model.parameters(), lr=0.1))
        torch.manual_seed(7 + rank)
        x = torch.rand(2, 2, 8, 8, 8)
        sync.zero_grad()
        model(x).square().mean().backward()
        local = torch.cat([p.grad.reshape(-1).clone() for p in model.parameters()]) if False else None
        sync.finish()
        g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
        opt.step()
        w = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        b = torch.cat([t.reshape(-1) for t in model.buffers()])
        gathered = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(gathered, g)
        wg = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(wg, w)
        bg = [torch.zeros_like(b) for _ in range(world)]
        dist.all_gather(bg, b)
        if rank == 0:
            torch.save({"g": gathered, "w": wg, "b": bg}, out)
    finally:
        dist.destroy_process_group()


def _single_process_reference(world):
    import factorizer_amd as ft
    torch.manual_seed(100)
    model = ft.Factorizer(in_channels=2, out_channels=2, spatial_size=(8, 8, 8), encoder_depth=(1, 1),
                          encoder_width=(8, 16), strides=(1, 2), decoder_depth=(1,),
                          reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}), act=nn.ReLU,
                          factorize=ft.NMF, rank=1, num_iters=3, solver="hals", mlp_ratio=2, dropout=0.0)
    grads = []
    for rank in range(world):
        torch.manual_seed(7 + rank)
        x = torch.rand(2, 2, 8, 8, 8)
        model.zero_grad()
        model(x).square().mean().backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]))
    return torch.stack(grads).mean(0)


@pytest.mark.parametrize("overlap", [False, True])
def test_flat_grad_sync_world2(tmp_path, overlap):
    world = 2
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(world, _free_port(), overlap, out), nprocs=world, join=True)
    res = torch.load(out)
    ref = _single_process_reference(world)
    for r in range(world):
        assert torch.allclose(res["g"][r], ref, rtol=1e-5, atol=1e-7)
    assert torch.equal(res["g"][0], res["g"][1])
    assert torch.equal(res["w"][0], res["w"][1])   # replicas stay identical after the step
    assert torch.equal(res["b"][0], res["b"][1])   # u0/v0 buffers were broadcast


def test_flat_adamw_on_the_reduced_buffer_world2(tmp_path):
    """bench.py's optimizer path: FlatAdamW reads the gradient buffer RCCL/gloo reduced in place; the
    replicas stay bit-identical after the step."""
    world = 2
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(world, _free_port(), True, out, True), nprocs=world, join=True)
    res = torch.load(out)
    assert torch.equal(res["w"][0], res["w"][1]) and torch.equal(res["g"][0], res["g"][1])
    assert torch.isfinite(res["w"][0]).all()


def _worker_unused(rank, world, port, out):
    """A requires_grad parameter that takes no part in the loss never fires its hook: finish() must still
    reduce its bucket (zero-filled slice) instead of handing back stale memory (parallel.py)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from factorizer_amd.parallel import FlatGradSync
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(4, 4), nn.Linear(4, 4), nn.Linear(4, 4))
        sync = FlatGradSync(model, num_buckets=3, overlap=True)
        sync.broadcast_state(0)
        res = []
        for step in range(2):  # the second step would see the first step's data if the slice were stale
            torch.manual_seed(10 * step + rank)
            x = torch.rand(3, 4)
            sync.zero_grad()
            (model[2](x).sum() + model[0](x).sum()).backward()   # model[1] unused
            scale = sync.finish(average=(step == 0))
            res.append(torch.cat([p.grad.reshape(-1) * scale for p in model.parameters()]))
        gathered = [torch.zeros_like(res[1]) for _ in range(world)]
        dist.all_gather(gathered, res[1])
        if rank == 0:
            torch.save({"g0": res[0], "g1": gathered}, out)
    finally:
        dist.destroy_process_group()


def test_unused_parameter_bucket_is_reduced_world2(tmp_path):
    world = 2
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker_unused, args=(world, _free_port(), out), nprocs=world, join=True)
    res = torch.load(out)
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(4, 4), nn.Linear(4, 4), nn.Linear(4, 4))
    for step, got in ((0, res["g0"]), (1, res["g1"][0])):
        gs = []
        for rank in range(world):
            torch.manual_seed(10 * step + rank)
            x = torch.rand(3, 4)
            model.zero_grad()
            (model[2](x).sum() + model[0](x).sum()).backward()
            gs.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                 for p in model.parameters()]))
        assert torch.allclose(got, torch.stack(gs).mean(0), rtol=1e-6, atol=1e-7)
    assert torch.equal(res["g1"][0], res["g1"][1])
    n0, n1 = 20, 40  # model[1]'s weights + bias occupy [20, 40) of the concatenation
    assert res["g1"][0][n0:n1].abs().max().item() == 0.0


def test_force_collectives_single_rank_runs_the_hook_path(tmp_path):
    """bench.py --force-dist: a one-rank group still launches the bucket all-reduces from the hooks."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from factorizer_amd.parallel import FlatGradSync
        torch.manual_seed(0)
        model = nn.Sequential(nn.Linear(4, 4), nn.Linear(4, 2))
        ref = nn.Sequential(nn.Linear(4, 4), nn.Linear(4, 2))
        ref.load_state_dict(model.state_dict())
        sync = FlatGradSync(model, num_buckets=2, overlap=True, force_collectives=True)
        assert sync.active and sync.overlap
        x = torch.rand(5, 4)
        sync.zero_grad()
        model(x).sum().backward()
        assert any(sync._launched), "no bucket was launched from the hooks"
        assert sync.finish(average=False) == 1.0
        ref(x).sum().backward()
        for p, q in zip(model.parameters(), ref.parameters()):
            assert p.grad.data_ptr() == sync.views[p].data_ptr() and torch.equal(p.grad, q.grad)
    finally:
        dist.destroy_process_group()
