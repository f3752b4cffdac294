"""CPU-side tests of the round-4 host logic (no GPU): the geometry rules of the two-window launch (pure host functions of the
library), the per-call `products` / `tune` plumbing, the flat-gradient destination table holding buffers weakly, and the
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


def _sup(B, C, S, sh, R=1, dt=0, nshift=2):
    arr = (N._i * 6)(*sh[0], *sh[1])
    return N.lib().fz_nmf_cf2_supported(B, C, *S, arr, nshift, R, 5, 5, dt)


def test_two_window_launch_geometry_rules():
    ok = [(0, 0, 0), (4, 4, 4)]
    assert _sup(2, 32, (128, 128, 128), ok) == 1                      # README stage 0
    assert _sup(2, 64, (64, 64, 64), ok) == 1                         # stage 1
    assert _sup(2, 32, (128, 128, 128), ok, dt=1) == 1                # bf16 storage
    assert _sup(2, 128, (32, 32, 32), ok) == 0                        # W / 8 = 4: no eight-patch tile
    assert _sup(2, 32, (128, 128, 128), ok, R=2) == 0                 # rank 2 stays on the one-window launches
    assert _sup(2, 32, (128, 128, 128), ok, nshift=4) == 0            # only the two-window default
    assert _sup(2, 32, (128, 128, 128), [(4, 4, 4), (0, 0, 0)]) == 0  # window 0 must be the unshifted one
    assert _sup(2, 32, (128, 128, 128), [(0, 0, 0), (0, 4, 4)]) == 0  # D shift 0: not the plane - 1 / plane dependency
    assert _sup(2, 32, (128, 128, 128), [(0, 0, 0), (8, 4, 4)]) == 0
    assert _sup(2, 32, (128, 128, 128), [(0, 0, 0), (4, 4, 6)]) == 0  # W shift 2 (mod 4)
    assert _sup(2, 32, (128, 128, 128), [(0, 0, 0), (-4, 4, 4)]) == 0 # -4 = 124 (mod 128): needs planes plane + 15 / + 16, not built
    assert _sup(16, 32, (128, 128, 128), ok) == 0                     # 4.3 GB tensor: beyond 32-bit buffer offsets


def test_two_window_workspace_size():
    lib = N.lib()
    n = lib.fz_nmf_cf2_workspace_bytes(2, 32, 128)
    assert n % 16 == 0 and n >= (16 + 2 * 4 * 16) * 4
    assert lib.fz_nmf_cf2_workspace_bytes(-1, 32, 128) == 0
    # null pointers are refused before anything is launched
    arr = (N._i * 6)(0, 0, 0, 4, 4, 4)
    rc = lib.fz_nmf_cf_fwd2(None, None, None, None, 2, 32, 128, 128, 128, arr, 1, 5, 1, 1e-16, 0, None, None, None)
    assert rc != 0 and b"null" in lib.fz_last_error_string()


def test_products_setting_nests_and_reaches_the_descriptors():
    assert N.products() == N.PRODUCTS_DEFAULT
    with N.use_products(N.PRODUCTS_FP32_MFMA):
        assert N.products() == N.PRODUCTS_FP32_MFMA
        with N.use_products(N.PRODUCTS_SPLIT_BF16):
            assert N.products() == N.PRODUCTS_SPLIT_BF16
        assert N.products() == N.PRODUCTS_FP32_MFMA
    assert N.products() == N.PRODUCTS_DEFAULT
    # the field is the LAST-but-one int of fz_gemm_desc (then `tune`), the last of fz_mlp_desc / fz_wgrad_desc: a descriptor that
    # leaves them zero follows the process default, and the process default itself is untouched by the context manager
    assert [f[0] for f in N.GemmDesc._fields_][-2:] == ["products", "tune"]
    assert N.MlpDesc._fields_[-1][0] == "products" and N.WgradDesc._fields_[-1][0] == "products"
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
