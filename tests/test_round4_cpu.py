"""CPU-side tests of the round-4 host logic (no GPU): the per-call `products` / `tune` plumbing, the flat-gradient destination table holding buffers weakly, and the
read-once diagnostic knobs of the Python layer.  Reference behaviour these serve: SWMatricize's default two windows
(operations.py:395-398), DistributedDataParallel's bucket memory (train_multigpu.yaml:3-6)."""
import copy
import ctypes
import gc
import importlib

import pytest
import torch

from factorizer_amd import _native as N
from factorizer_amd import gradbuf


def test_products_setting_nests_and_reaches_the_descriptors():
    assert N.products() == N.PRODUCTS_DEFAULT
    with N.use_products(N.PRODUCTS_FP32_MFMA):
        assert N.products() == N.PRODUCTS_FP32_MFMA
        with N.use_products(N.PRODUCTS_SPLIT_BF16):
            assert N.products() == N.PRODUCTS_SPLIT_BF16
        assert N.products() == N.PRODUCTS_FP32_MFMA
    assert N.products() == N.PRODUCTS_DEFAULT
    # the field is the LAST-but-one int of fz_gemm_desc (then `tune`), the last of fz_wgrad_desc, and in fz_mlp_desc the last field
    # before round 5's pre_* / post_* block: a descriptor that leaves them zero follows the process default, and the process default
    # itself is untouched by the context manager
    assert [f[0] for f in N.GemmDesc._fields_][-2:] == ["products", "tune"]
    assert N.WgradDesc._fields_[-1][0] == "products"
    mlp = [f[0] for f in N.MlpDesc._fields_]
    assert mlp[mlp.index("products") + 1:] == ["pre_in", "pre_w", "pre_b", "pre_res", "pre_out", "post_w", "post_b", "post_out", "post_m"]
    assert N.lib().fz_gemm_bx_enable(-1) == 1
    d = N.GemmDesc()
    assert d.products == 0 and d.tune == 0


def test_gradient_destinations_do_not_keep_dead_buffers_alive():
    gradbuf._DST.clear()
    p = torch.nn.Parameter(torch.zeros(4, 4))
    flat = torch.zeros(16)
    view = flat[:16].view(4, 4)
    gradbuf.register(flat, {p: view})
    out = gradbuf.out_like(p)
    assert out.data_ptr() == flat.data_ptr()          # the slice is handed out ...
    assert gradbuf.out_like(p).data_ptr() != flat.data_ptr()   # ... once per release()
    gradbuf.release(flat)
    ptr = flat.data_ptr()
    del out, view, flat
    gc.collect()
    got = gradbuf.out_like(p)                         # the owner is gone: a fresh tensor, and the entry is dropped
    assert got.shape == (4, 4) and p.data_ptr() not in gradbuf._DST
    del ptr


def test_deep_copied_optimizer_registers_its_own_buffer():
    import factorizer_amd as ft
    gradbuf._DST.clear()
    lin = torch.nn.Linear(4, 4)
    opt = ft.FlatAdamW(lin, lr=1e-3)
    pair = copy.deepcopy({"m": lin, "o": opt})       # scripts/utils.py:21 deep-copies {network, optimizer} before loading
    m2, o2 = pair["m"], pair["o"]
    w2 = m2.weight
    assert w2.data_ptr() == o2.flat_param[o2.offsets[w2]:].data_ptr()
    dst = gradbuf.out_like(w2)
    assert dst.untyped_storage().data_ptr() == o2.flat_grad.untyped_storage().data_ptr()
    assert dst.untyped_storage().data_ptr() != opt.flat_grad.untyped_storage().data_ptr()


def test_decoder_knobs_are_read_once_and_validated(monkeypatch):
    from factorizer_amd import blocks
    monkeypatch.setenv("FZ_UP_FUSED_MIN", "not-a-number")
    with pytest.raises(ValueError, match="FZ_UP_FUSED_MIN"):
        blocks._env_int("FZ_UP_FUSED_MIN", 1)
    monkeypatch.setenv("FZ_UP_FUSED_MIN", "0x100")
    assert blocks._env_int("FZ_UP_FUSED_MIN", 1) == 256
    # the module-level values were fixed at import: changing the environment later does not reach forward_up_pair
    assert isinstance(blocks._UP_FUSED_MIN, int) and isinstance(blocks._UP_FUSED, bool)


def test_products_setting_is_per_thread_and_travels_with_the_autograd_node():
    """the setting a forward ran under must be the one its backward uses — also when the backward runs after the
    `with` block has exited, or on another thread (autograd's workers) — and two threads must not see each other's."""
    import threading
    seen = {}

    class Fn(torch.autograd.Function):
        @staticmethod
        @N.capture_products
        def forward(ctx, x):
            seen["fwd"] = N.products()
            return x * 2

        @staticmethod
        @N.with_products
        def backward(ctx, g):
            seen["bwd"] = N.products()
            return g * 2

    x = torch.ones(3, requires_grad=True)
    with N.use_products(N.PRODUCTS_FP32_MFMA):
        y = Fn.apply(x).sum()
        other = []
        th = threading.Thread(target=lambda: other.append(N.products()))
        th.start(); th.join()
        assert other == [N.PRODUCTS_DEFAULT]          # another thread keeps its own setting
    assert N.products() == N.PRODUCTS_DEFAULT
    y.backward()                                       # after the block: the node re-establishes what its forward saw
    assert seen == {"fwd": N.PRODUCTS_FP32_MFMA, "bwd": N.PRODUCTS_FP32_MFMA}
    assert N.products() == N.PRODUCTS_DEFAULT
