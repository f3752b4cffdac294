"""Host-side mirrors of the reference modules (factorizer_amd/*) on CPU tensors: constructor
surface, RNG / state_dict parity with the reference, and values vs the reference goldens.
(The composed CPU path is plumbing — BASELINE config 0 — the device path is tested in -m gpu.)"""
import numpy as np
import pytest
import torch
from torch import nn

import factorizer_amd as ft
from test_oracle_golden import G1_CASES, NMF_CASES


@pytest.mark.parametrize("name", sorted(G1_CASES))
def test_swmatricize_cpu_bit_exact(golden, name):
    g = golden("g1_swmatricize").case(name)
    shape, kw = G1_CASES[name]
    x = torch.arange(int(np.prod(shape)), dtype=torch.float32).reshape(shape)
    m = ft.SWMatricize((None, *shape[1:]), **kw)
    y = m(x)
    assert torch.equal(y.to(torch.int32), g["y"])
    assert [(-1 if s is None else s) for s in m.output_size] == g["output_size"].tolist()
    z = m.inverse_forward(y)
    assert torch.equal(z, g["z"])
    if "yr" in g:
        assert torch.equal(m.inverse_forward(g["yr"]), g["zr"])


def test_matricize_single_window_roundtrip():
    m = ft.Matricize((None, 16, 8, 8, 8), num_heads=1, grid_size=1)
    assert m.output_size == (None, 1, 16, 512)
    x = torch.rand(2, 16, 8, 8, 8)
    y = m(x)
    assert y.shape == (2, 1, 16, 512)
    assert torch.equal(y.reshape(2, 16, 512), x.reshape(2, 16, 512))
    assert torch.equal(m.inverse_forward(y), x)
    m2 = ft.Matricize((None, 8, 8, 8), head_dim=4, patch_size=4, shifts=1)  # 2-D
    x2 = torch.rand(3, 8, 8, 8)
    assert torch.equal(m2.inverse_forward(m2(x2)), x2)


def test_matricize_validation():
    with pytest.raises(ValueError):
        ft.SWMatricize((None, 32, 10, 12, 10), head_dim=8, patch_size=8)  # BASELINE cfg 5 shape, p=8
    with pytest.raises(ValueError):
        ft.SWMatricize((None, 30, 8, 8, 8), head_dim=8, patch_size=4)
    with pytest.raises(AssertionError):
        ft.SWMatricize((None, 32, 8, 8, 8), patch_size=4)
    ft.SWMatricize((None, 32, 10, 12, 10), head_dim=8, patch_size=(5, 6, 5))  # the valid cfg-5 choice


@pytest.mark.parametrize("name", sorted(NMF_CASES))
def test_nmf_cpu_vs_reference(golden, name):
    g = golden("g2_nmf").case(name)
    kw = dict(NMF_CASES[name])
    M, N = g["x"].shape[-2:]
    R = g["u0"].shape[1]
    torch.manual_seed(0)  # same seed + same construction order => same buffers (a-3)
    rank = None if name == "rank_auto" else R
    nmf = ft.NMF(size=(M, N), rank=rank, init="uniform", **kw)
    assert nmf.rank == R
    assert torch.equal(nmf.init.u0, g["u0"]) and torch.equal(nmf.init.v0, g["v0"])
    x = g["x"].clone().requires_grad_(True)
    u, v = nmf.decompose(x)
    y = nmf(x)
    assert torch.allclose(u, g["u"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(v, g["v"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(y, g["y"], rtol=1e-5, atol=1e-6)
    (gx,) = torch.autograd.grad(y, x, g["gy"])
    s = g["gx"].abs().max().item()
    assert (gx - g["gx"]).abs().max().item() <= 2e-5 * s + 1e-6
    assert torch.allclose(nmf.loss(x, u, v), g["loss"], rtol=1e-5, atol=1e-7)


class TestNMFContract:
    """Restates the reference's tests/test_nmf.py:9-39 (shape / sign contract)."""

    size = (2, 4, 8, 16)
    rank = 3

    def setup_method(self):
        self.nmf = ft.NMF(size=self.size[-2:], rank=self.rank, init="uniform", solver="hals")

    def test_decompose(self):
        x = torch.rand(self.size, requires_grad=True)
        u, v = self.nmf.decompose(x)
        assert u.shape == (*self.size[:-2], self.size[-2], self.rank)
        assert v.shape == (*self.size[:-2], self.size[-1], self.rank)
        assert (u >= 0).all() and (v >= 0).all()

    def test_forward(self):
        x = torch.rand(self.size, requires_grad=True)
        assert self.nmf(x).shape == x.shape

    def test_reconstruct_and_loss(self):
        x = torch.rand(self.size)
        u = torch.rand(*self.size[:-2], self.size[-2], self.rank)
        v = torch.rand(*self.size[:-2], self.size[-1], self.rank)
        assert self.nmf.reconstruct(u, v).shape == self.size
        loss = self.nmf.loss(x, u, v)
        assert loss.shape == self.size[:1] and (loss >= 0).all()


def test_solver_string_table_and_partial_specs():
    assert isinstance(ft.NMF((8, 16), rank=2, solver="mu").solver, ft.MultiplicativeUpdate)
    assert isinstance(ft.NMF((8, 16), rank=2, solver="hals").solver.project, nn.ReLU)
    assert isinstance(ft.MatrixFactorization((8, 16), rank=2).solver.project, nn.Identity)  # "cd"
    comp = ft.NMF((8, 16), rank=2, solver=["mu", "hals"])
    assert isinstance(comp.solver, ft.Compose) and len(comp.solver.solvers) == 2
    ft.NMF((8, 16), rank=2, solver=[ft.MultiplicativeUpdate, {"eps": 1e-8}])  # YAML-style list spec
    assert isinstance(ft.NMF((8, 16), rank=2, solver="fmu").solver, ft.FastMultiplicativeUpdate)
    assert isinstance(ft.NMF((8, 16), rank=2, solver="nnls-1").solver.project, nn.ReLU)
    assert isinstance(ft.NMF((8, 16), rank=2, init="nndsvd").init, ft.NNDSVDInit)
    with pytest.raises(ValueError):
        ft.NMF((8, 16), rank=2, solver="nope")
    f = ft.partialize((nn.Linear, (3,), {"out_features": 4}))
    assert f().weight.shape == (4, 3)


BLOCK_KW = {
    "hals_r1": dict(rank=1, num_iters=5, solver="hals"),
    "mu_r2": dict(rank=2, num_iters=3, solver="mu"),
}


@pytest.mark.parametrize("name", sorted(BLOCK_KW))
def test_block_seed_state_dict_and_values(golden, name):
    g = golden("g5_block").case(name)
    torch.manual_seed(0)
    blk = ft.FactorizerBlock(channels=16, spatial_size=(8, 8, 8), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}), act=nn.ReLU,
                             factorize=ft.NMF, init="uniform", mlp_ratio=2, dropout=0.0, **BLOCK_KW[name])
    sd = blk.state_dict()
    ref = {k[3:]: v for k, v in g.items() if k.startswith("sd:")}
    assert sorted(sd) == sorted(ref)
    for k in sd:  # same seed + same construction order => identical parameters AND buffers
        assert torch.equal(sd[k], ref[k]), k
    x = g["x"].clone().requires_grad_(True)
    y = blk(x)
    assert torch.allclose(y, g["y"], rtol=1e-4, atol=1e-5)
    names = [k for k, _ in blk.named_parameters()]
    grads = torch.autograd.grad(y, [x] + list(blk.parameters()), g["gy"])
    assert torch.allclose(grads[0], g["gx"], rtol=1e-3, atol=2e-4)
    for k, gr in zip(names, grads[1:]):
        r = g["grad:" + k]
        assert (gr - r).abs().max().item() <= 1e-3 * (r.abs().max().item() + 1e-6), k


def _tiny_model():
    return ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(16, 16, 16), encoder_depth=(1, 1, 1),
                         encoder_width=(8, 16, 32), strides=(1, 2, 2), decoder_depth=(1, 1),
                         norm=ft.LayerNorm, reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}),
                         act=nn.ReLU, factorize=ft.NMF, rank=1, num_iters=5, init="uniform",
                         solver="hals", mlp_ratio=2, dropout=0.1)


def test_model_seed_state_dict_and_values(golden):
    g = golden("g6_model")
    torch.manual_seed(0)
    model = _tiny_model().eval()
    ref = g.case("sd")
    sd = model.state_dict()
    assert sorted(sd) == sorted(ref)
    for k in sd:
        assert torch.equal(sd[k], ref[k]), k
    assert sum(p.numel() for p in model.parameters()) == int(g["num_params"][0])
    x = g["x"].clone().requires_grad_(True)
    y = model(x)
    assert torch.allclose(y, g["y"], rtol=1e-4, atol=1e-5)
    names = [k for k, _ in model.named_parameters()]
    grads = torch.autograd.grad(y, [x] + list(model.parameters()), g["gy"])
    assert torch.allclose(grads[0], g["gx"], rtol=1e-3, atol=1e-4)
    for k, gr in zip(names, grads[1:]):
        r = g["grad:" + k]
        assert (gr - r).abs().max().item() <= 2e-3 * (r.abs().max().item() + 1e-6), k
    # dropout only reaches the bottleneck pos_drop (SURVEY headline 6)
    drops = {n: m.p for n, m in model.named_modules() if isinstance(m, nn.Dropout)}
    assert [n for n, p in drops.items() if p > 0] == ["encoder.blocks.2.block.pos_drop"]


def test_readme_model_state_dict_inventory(golden):
    """Keys and shapes of the README Swin Factorizer (128^3, widths 32..512) match the
    reference's, so its checkpoints load."""
    g = golden("g6_readme_model_keys")
    with torch.device("meta"):
        model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(128, 128, 128),
                              norm=ft.LayerNorm, reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}),
                              act=nn.ReLU, factorize=ft.NMF, rank=1, num_iters=5, init="uniform",
                              solver="hals", mlp_ratio=2, dropout=0.1)
    sd = model.state_dict()
    ref = {k: tuple(g[k].tolist()) for k in g.keys() if k != "__num_params__"}
    assert sorted(sd) == sorted(ref)
    for k, v in sd.items():
        assert tuple(v.shape) == ref[k], k
    assert sum(p.numel() for p in model.parameters()) == int(g["__num_params__"][0]) == 5855619


def test_variable_batch_and_reference_factorizer_test_config():
    """Restates tests/test_factorizer.py:112-165 at reduced spatial size (num_heads=8 form:
    M = C/8 varies per stage, N = 64)."""
    model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(16, 16, 16),
                          encoder_depth=(1, 1, 1), encoder_width=(32, 64, 128), strides=(1, 2, 2),
                          decoder_depth=(1, 1), reshape=(ft.SWMatricize, {"num_heads": 8, "patch_size": 4}),
                          act=nn.ReLU, factorize=ft.NMF, rank=1, num_iters=5, num_grad_steps=None,
                          init="uniform", solver="hals", mlp_ratio=2, dropout=0.1)
    for b in (1, 2, 3):
        y = model(torch.rand(b, 4, 16, 16, 16))
        assert y.shape == (b, 3, 16, 16, 16) and torch.isfinite(y).all()


def test_factmixer_global_matricize_mu():
    """tests/test_factorizer.py:14-48 at reduced size: global Matricize, MU, rank 1."""
    fm = ft.FactMixer(in_channels=16, out_channels=16, spatial_size=(8, 8, 8),
                      reshape=(ft.Matricize, {"num_heads": 1, "grid_size": 1}), act=nn.ReLU,
                      factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="mu", dropout=0.1)
    x = torch.rand(1, 16, 8, 8, 8, requires_grad=True)
    y = fm(x)
    assert y.shape == x.shape and torch.isfinite(y).all()
    y.sum().backward()
    assert torch.isfinite(x.grad).all()


# ---- sliding-window inference -------------------------------------------------------
def _toy_net(w):
    conv = torch.nn.Conv3d(2, 3, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(w)
        conv.bias.fill_(0.1)
    return lambda x: torch.tanh(conv(x))


@pytest.mark.parametrize("size,roi,ov,mode", [((20, 24, 27), (16, 16, 16), 0.5, "gaussian"),
                                              ((16, 16, 16), (16, 16, 16), 0.5, "gaussian"),
                                              ((24, 16, 20), (16, 16, 8), 0.25, "constant")])
def test_sliding_window_inference_matches_restated_algorithm(size, roi, ov, mode):
    """ft.sliding_window_inference (composed CPU path) against the dense-map restatement of MONAI's
    published algorithm in oracle/cpu_ref.py; window grid of the bundle's settings checked explicitly."""
    from oracle import cpu_ref as O
    from factorizer_amd import inference as I
    torch.manual_seed(0)
    net = _toy_net(torch.randn(3, 2, 3, 3, 3) * 0.2)
    x = torch.randn(2, 2, *size)
    with torch.no_grad():
        y = ft.sliding_window_inference(x, roi, 2, net, overlap=ov, mode=mode)
        yo = O.sliding_window_oracle(x, roi, 2, net, overlap=ov, mode=mode)
    assert y.shape == (2, 3, *size)
    assert torch.allclose(y, yo, rtol=1e-5, atol=1e-6)
    # BraTS geometry of the bundle (inference.yaml:96-102): 240x240x155, roi 128^3, overlap 0.5
    st = I.window_starts((240, 240, 155), (128,) * 3, I.scan_interval((240, 240, 155), (128,) * 3, (0.5,) * 3))
    assert len(st) == 3 * 3 * 2 and st[0] == (0, 0, 0) and st[-1] == (112, 112, 27)


def test_sliding_window_pads_small_volumes():
    torch.manual_seed(1)
    net = _toy_net(torch.randn(3, 2, 3, 3, 3) * 0.2)
    x = torch.randn(1, 2, 12, 16, 10)
    inf = ft.SlidingWindowInfererAdapt(roi_size=(16, 16, 16), sw_batch_size=2, overlap=0.5, mode="gaussian",
                                       cache_roi_weight_map=True)
    with torch.no_grad():
        y = inf(x, net)
        xp = torch.nn.functional.pad(x, (3, 3, 0, 0, 2, 2))
        ref = net(xp)[:, :, 2:14, :, 3:13]
    assert y.shape == (1, 3, 12, 16, 10)
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-6)  # a single window: the weights cancel


# ---- training-step pieces -------------------------------------------------------------
def test_flat_adamw_matches_torch_adamw():
    """FlatAdamW (one flat buffer; train.yaml:72-76 recipe) against torch.optim.AdamW, 6 steps, and the
    parameters stay views of the flat buffer through state_dict round trips."""
    import copy
    torch.manual_seed(0)
    m1 = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.GELU(), torch.nn.Linear(5, 3))
    m2 = copy.deepcopy(m1)
    o1 = ft.FlatAdamW(m1, lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    o2 = torch.optim.AdamW(m2.parameters(), lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    for _ in range(6):
        x = torch.randn(4, 7)
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            m(x).pow(2).sum().backward()
            o.step()
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    m1.load_state_dict(m2.state_dict())
    assert all(p.data_ptr() >= o1.flat_param.data_ptr() for p in m1.parameters())
    assert torch.equal(o1.flat_param[:35].view(5, 7), m2[0].weight)


def test_warmup_cosine_schedule_values():
    """MONAI WarmupCosineSchedule restated; the bundle's numbers (train.yaml:26-32,79-83): 500 epochs,
    warmup_steps 5, t_total 501, warmup_multiplier 0.1."""
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    s = ft.WarmupCosineSchedule(opt, warmup_steps=5, t_total=501, warmup_multiplier=0.1)
    assert abs(opt.param_groups[0]["lr"] - 1e-5) < 1e-12        # step 0: multiplier 0.1
    lrs = []
    for _ in range(501):
        s.step()
        lrs.append(opt.param_groups[0]["lr"])
    assert abs(lrs[0] - 1e-4 * (0.1 + 0.9 * 1 / 5)) < 1e-12      # step 1
    assert abs(lrs[4] - 1e-4) < 1e-12                            # step 5: end of warm-up
    assert abs(lrs[252] - 1e-4 * 0.5) < 1e-9                     # step 253: half-way through the cosine
    assert lrs[-1] < 1e-12 and all(a >= b - 1e-15 for a, b in zip(lrs[4:], lrs[5:]))


def test_load_checkpoint_interchange(tmp_path):
    """scripts/utils.py:10-26 semantics: objects[key].load_state_dict(checkpoint[key]); reference-keyed
    state_dicts load into the module unchanged (keys pinned by tests/golden/g6_readme_model_keys.npz)."""
    torch.manual_seed(0)
    kw = dict(in_channels=2, out_channels=2, spatial_size=(16, 16, 16), encoder_depth=(1, 1), encoder_width=(8, 16),
              strides=(1, 2), decoder_depth=(1,), norm=ft.LayerNorm,
              reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}), act=nn.ReLU, factorize=ft.NMF, rank=1,
              num_iters=2, init="uniform", solver="hals", mlp_ratio=2, dropout=0.0)
    src, dst = ft.Factorizer(**kw), ft.Factorizer(**kw)
    path = tmp_path / "ckpt.pt"
    torch.save({"network": src.state_dict(), "epoch": 3}, path)
    before = {k: v.clone() for k, v in dst.state_dict().items()}
    loaded = ft.load_checkpoint({"network": dst}, str(path))["network"]   # a loaded COPY (scripts/utils.py:21)
    assert loaded is not dst
    for (k1, v1), (k2, v2) in zip(src.state_dict().items(), loaded.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    for k, v in dst.state_dict().items():
        assert torch.equal(v, before[k])                        # the object passed in is untouched
    ft.load_checkpoint({"network": dst}, str(path), inplace=True)
    for (k1, v1), (k2, v2) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    ft.load_checkpoint({"network": dst}, src.state_dict())     # a bare state_dict
    with pytest.raises(KeyError):
        ft.load_checkpoint({"optimizer": dst}, {"network": src.state_dict(), "x": 1})
    # load_checkpoints (scripts/utils.py:29-32; the fold ensemble of inference.yaml:141-142): independent copies
    src2 = ft.Factorizer(**kw)
    p2 = tmp_path / "ckpt2.pt"
    torch.save({"network": src2.state_dict()}, p2)
    a, b = ft.load_checkpoints({"network": dst}, [str(path), str(p2)])
    assert a["network"] is not b["network"]
    assert torch.equal(a["network"].state_dict()["stem.weight"], src.state_dict()["stem.weight"])
    assert torch.equal(b["network"].state_dict()["stem.weight"], src2.state_dict()["stem.weight"])


def test_optimizer_and_scheduler_checkpoint_interchange_with_torch():
    """The recipe checkpoints {trainer, model, optimizer, lr_scheduler} (train.yaml:354-358): FlatAdamW emits and
    accepts torch.optim.AdamW's state_dict layout, the schedule carries last_epoch / base_lrs."""
    import copy
    torch.manual_seed(0)
    m_t = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.GELU(), torch.nn.Linear(5, 3))
    m_f = copy.deepcopy(m_t)
    o_t = torch.optim.AdamW(m_t.parameters(), lr=1e-2, weight_decay=1e-2)
    s_t = ft.WarmupCosineSchedule(o_t, warmup_steps=2, t_total=11, warmup_multiplier=0.1)
    xs = [torch.randn(4, 7) for _ in range(6)]
    for x in xs[:3]:
        o_t.zero_grad()
        m_t(x).pow(2).sum().backward()
        o_t.step()
        s_t.step()
    # torch -> flat: resume the same run with the flat optimizer
    o_f = ft.FlatAdamW(m_f, lr=1e-2, weight_decay=1e-2)
    s_f = ft.WarmupCosineSchedule(o_f, warmup_steps=2, t_total=11, warmup_multiplier=0.1)
    objs = ft.load_checkpoint({"model": m_f, "optimizer": o_f, "lr_scheduler": s_f},
                              {"model": m_t.state_dict(), "optimizer": o_t.state_dict(),
                               "lr_scheduler": s_t.state_dict()})
    m_f, o_f, s_f = objs["model"], objs["optimizer"], objs["lr_scheduler"]
    assert s_f.optimizer is o_f and all(p.data_ptr() >= o_f.flat_param.data_ptr() for p in m_f.parameters())
    assert abs(o_f.lr - o_t.param_groups[0]["lr"]) < 1e-15 and o_f.base_lr == 1e-2 and o_f.t == 3
    for x in xs[3:]:
        for m, o, sch in ((m_t, o_t, s_t), (m_f, o_f, s_f)):
            o.zero_grad()
            m(x).pow(2).sum().backward()
            o.step()
            sch.step()
    for a, b in zip(m_t.parameters(), m_f.parameters()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    assert abs(o_f.lr - o_t.param_groups[0]["lr"]) < 1e-15
    # flat -> torch
    m_2 = copy.deepcopy(m_t)
    o_2 = torch.optim.AdamW(m_2.parameters(), lr=1.0)
    o_2.load_state_dict(o_f.state_dict())
    sd_t = o_t.state_dict()
    for i, st in o_2.state_dict()["state"].items():
        assert float(st["step"]) == float(sd_t["state"][i]["step"]) == 6.0
        assert torch.allclose(st["exp_avg"], sd_t["state"][i]["exp_avg"], rtol=1e-5, atol=1e-8)
        assert torch.allclose(st["exp_avg_sq"], sd_t["state"][i]["exp_avg_sq"], rtol=1e-5, atol=1e-10)
    assert abs(o_2.param_groups[0]["lr"] - o_t.param_groups[0]["lr"]) < 1e-15


def test_flat_adamw_param_groups_are_persistent_and_single():
    """Utilities written against torch optimizers WRITE param_groups (g["lr"] = ...): the group is one persistent dict
    whose hyperparameters are the optimizer's own; a checkpoint with per-group hyperparameters is refused instead of
    being loaded with the first group's."""
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = ft.FlatAdamW(ps, lr=1e-2, weight_decay=1e-2)
    o_ref = torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2)
    assert opt.param_groups[0] is opt.param_groups[0]
    for o in (opt, o_ref):
        for g in o.param_groups:
            g["lr"] = 3e-3
            g["weight_decay"] = 0.5
    assert opt.lr == 3e-3 and opt.weight_decay == 0.5
    opt.lr = 2e-3
    assert opt.param_groups[0]["lr"] == 2e-3
    o_ref.param_groups[0]["lr"] = 2e-3
    for p, r in zip(ps, ref):
        g = torch.randn_like(p)
        p.grad, r.grad = g.clone(), g.clone()
    opt.step()
    o_ref.step()
    for p, r in zip(ps, ref):
        assert torch.allclose(p, r, rtol=1e-6, atol=1e-7)
    two = torch.optim.AdamW([{"params": [ref[0]], "lr": 1e-3}, {"params": [ref[1]], "lr": 5e-4}])
    with pytest.raises(ValueError, match="param_groups differ"):
        opt.load_state_dict(two.state_dict())
    same = torch.optim.AdamW([{"params": [ref[0]]}, {"params": [ref[1]]}], lr=1e-3)
    opt.load_state_dict(same.state_dict())
    assert opt.lr == 1e-3


def test_flat_adamw_skips_parameters_without_gradient():
    """torch.optim.AdamW leaves a parameter whose .grad is None untouched (no decay, no step count)."""
    import copy
    torch.manual_seed(0)
    m1 = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 4))
    m2 = copy.deepcopy(m1)
    o1 = ft.FlatAdamW(m1, lr=1e-2, weight_decay=0.1)
    o2 = torch.optim.AdamW(m2.parameters(), lr=1e-2, weight_decay=0.1)
    for it in range(5):
        x = torch.randn(3, 4)
        for m, o in ((m1, o1), (m2, o2)):
            o.zero_grad()
            (m[0](x).sum() + (m[2](x).pow(2).sum() if it != 2 else 0.0)).backward()   # m[1] never, m[2] skips once
            o.step()
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-8)
    assert o1.steps[m1[1].weight] == 0 and o1.steps[m1[2].weight] == 4 and o1.steps[m1[0].weight] == 5


def test_dice_ce_loss_matches_oracle_and_torch():
    """The recipe's DiceCELoss(sigmoid=True, squared_pred=True) in MONAI >= 1.3 semantics (softmax cross entropy
    with float multi-label targets for a multi-channel head; BCE for one channel)."""
    from oracle import cpu_ref as O
    torch.manual_seed(3)
    for C in (1, 3):
        z = torch.randn(2, C, 4, 6, 8, requires_grad=True)
        t = (torch.rand(2, C, 4, 6, 8) > 0.5).float()
        a, b = ft.dice_ce_loss(z, t), O.dice_ce_loss(z, t)
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    z = torch.randn(2, 3, 4, 6, 8)
    ce = -(t * torch.log_softmax(z, dim=1)).sum(1).mean()      # the formula in the docstring
    p = torch.sigmoid(z)
    dice = (1 - (2 * (p * t).sum((2, 3, 4)) + 1e-5) / ((p * p).sum((2, 3, 4)) + (t * t).sum((2, 3, 4)) + 1e-5)).mean()
    assert torch.allclose(ft.DiceCELoss()(z, t), dice + ce, rtol=1e-6, atol=1e-7)


# ---- remaining solver / initialiser keys ----------------------------------------------
F3_CASES = {"fmu": dict(solver="fmu", init="uniform"), "smu": dict(solver="smu", init="uniform"),
            "ls": dict(solver="ls", init="uniform"), "nnls": dict(solver="nnls", init="uniform"),
            "cd": dict(solver="cd", init="normal"), "nncd": dict(solver="nncd", init="uniform"),
            "mu_0": dict(solver="mu-0", init="uniform"), "hals_1": dict(solver="hals-1", init="uniform"),
            "compose_mu_hals": dict(solver=["mu", "hals"], init="uniform"),
            "compose_ls1_nnls0": dict(solver=["ls-1", "nnls-0"], init="uniform-normal"),
            "mu_svd": dict(solver="mu", init="svd"), "hals_nndsvd": dict(solver="hals", init="nndsvd"),
            "ls_tall": dict(solver="ls", init="uniform")}


@pytest.mark.parametrize("name", sorted(F3_CASES))
def test_solver_and_init_keys_vs_reference(golden, name):
    """Every key of the reference's INIT / SOLVER dispatch tables (matrix_factorization.py:581-618)
    against outputs of the reference itself (tests/golden/g8_solvers.npz, tools/make_goldens.py:g8)."""
    g = golden("g8_solvers").case(name)
    kw = dict(F3_CASES[name])
    torch.manual_seed(0)
    mf = ft.MatrixFactorization(size=tuple(g["x"].shape[-2:]), rank=2, num_iters=3, **kw)
    x = g["x"].clone().requires_grad_(True)
    u0, v0 = mf.init(x)
    tol = dict(rtol=2e-4, atol=2e-5) if kw["init"] in ("svd", "nndsvd") else dict(rtol=1e-5, atol=1e-6)
    assert torch.allclose(u0, g["u_init"], **tol) and torch.allclose(v0, g["v_init"], **tol)
    u, v = mf.decompose(x)
    y = mf(x)
    assert torch.allclose(u, g["u"], **tol) and torch.allclose(v, g["v"], **tol)
    assert torch.allclose(y, g["y"], **tol)
    if "gx" in g:
        (gx,) = torch.autograd.grad(y, x, g["gy"])
        s = g["gx"].abs().max().item()
        assert (gx - g["gx"]).abs().max().item() <= 1e-4 * s + 1e-6


def test_weighted_mu_and_svd_layer_vs_reference(golden):
    g = golden("g8_solvers").case("wmu")
    torch.manual_seed(0)
    mf = ft.NMF(size=(8, 24), rank=2, num_iters=3, init="uniform", solver="wmu")
    u, v = mf.decompose(g["x"], g["w"])
    assert torch.allclose(u, g["u"], rtol=1e-5, atol=1e-6) and torch.allclose(v, g["v"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(mf.loss(g["x"], u, v, g["w"]), g["loss"], rtol=1e-5, atol=1e-7)
    s = golden("g8_solvers").case("svd")
    from factorizer_amd.nmf import SVD
    torch.manual_seed(123)
    y = SVD(size=(8, 24), rank=3)(s["x"])
    assert torch.allclose(y, s["y"], rtol=2e-4, atol=2e-5)
    assert torch.initial_seed() == 42   # the reference's global re-seed (matrix_factorization.py:434)


def test_compose_is_a_sequence_of_solvers():
    from factorizer_amd.nmf import Compose, MultiplicativeUpdate, CoordinateDescent
    c = Compose(["mu", ("hals-1", {})] if False else [MultiplicativeUpdate, (CoordinateDescent, {"factor": 1})],
                size=(8, 16), rank=2)
    assert len(c) == 2 and isinstance(c[0], MultiplicativeUpdate) and c.factor == [(0, 1), (1,)]
    assert c.size == (8, 16) and c.rank == 2


def test_empty_batch_keeps_shapes():
    """Batch 0 (the reference's einops/ATen chain accepts it): shapes survive forward and backward."""
    m = ft.SWMatricize((None, 16, 8, 8, 8), head_dim=8, patch_size=4)
    x = torch.rand(0, 16, 8, 8, 8)
    y = m(x)
    assert y.shape == (0, 8, 8, 64) and m.inverse_forward(y).shape == x.shape
    nmf = ft.NMF(size=(8, 64), rank=2, num_iters=3, init="uniform", solver="hals")
    assert nmf(y).shape == y.shape
    blk = ft.FactorizerBlock(channels=32, spatial_size=(8, 8, 8), norm=ft.LayerNorm,
                             reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                             factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2,
                             dropout=0.0)
    xb = torch.rand(0, 32, 8, 8, 8, requires_grad=True)
    yb = blk(xb)
    yb.sum().backward()
    assert yb.shape == xb.shape and xb.grad.shape == xb.shape


# ---- the N-D generic path on 2-D and 1-D tensors (operations.py:318-325; goldens g10 from the imported reference) ----
G10_BLOCKS = {
    "blk2d_hals_r1": (dict(rank=1, num_iters=5, solver="hals"), 16, (16, 16), 4),
    "blk2d_mu_r2": (dict(rank=2, num_iters=3, solver="mu"), 16, (16, 24), (4, 8)),
    "blk2d_c32_p8": (dict(rank=1, num_iters=5, solver="hals"), 32, (32, 32), 8),
    "blk1d_hals_r1": (dict(rank=1, num_iters=5, solver="hals"), 16, (64,), 16),
}


def lower_d_block(name):
    kw, C, S, patch = G10_BLOCKS[name]
    return ft.FactorizerBlock(channels=C, spatial_size=S, norm=ft.LayerNorm, reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": patch}),
                              act=nn.ReLU, factorize=ft.NMF, init="uniform", mlp_ratio=2, dropout=0.0, **kw)


def lower_d_model():
    return ft.Factorizer(in_channels=3, out_channels=2, spatial_size=(32, 32), encoder_depth=(1, 1), encoder_width=(16, 32), strides=(1, 2),
                         decoder_depth=(1,), norm=ft.LayerNorm, reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}), act=nn.ReLU,
                         factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2, dropout=0.1)


@pytest.mark.parametrize("name", sorted(G10_BLOCKS))
def test_lower_d_block_seed_state_dict_and_values(golden, name):
    g = golden("g10_lower_d").case(name)
    torch.manual_seed(0)
    blk = lower_d_block(name)
    sd = blk.state_dict()
    ref = {k[3:]: v for k, v in g.items() if k.startswith("sd:")}
    assert sorted(sd) == sorted(ref)
    for k in sd:
        assert torch.equal(sd[k], ref[k]), k
    x = g["x"].clone().requires_grad_(True)
    y = blk(x)
    assert torch.allclose(y, g["y"], rtol=1e-4, atol=1e-5)
    names = [k for k, _ in blk.named_parameters()]
    grads = torch.autograd.grad(y, [x] + list(blk.parameters()), g["gy"])
    assert torch.allclose(grads[0], g["gx"], rtol=1e-3, atol=2e-4)
    for k, gr in zip(names, grads[1:]):
        r = g["grad:" + k]
        assert (gr - r).abs().max().item() <= 1e-3 * (r.abs().max().item() + 1e-6), k


def test_lower_d_model_seed_state_dict_and_values(golden):
    g = golden("g10_lower_d").case("model2d")
    torch.manual_seed(0)
    model = lower_d_model().eval()
    ref = {k[3:]: v for k, v in g.items() if k.startswith("sd:")}
    sd = model.state_dict()
    assert sorted(sd) == sorted(ref)
    for k in sd:
        assert torch.equal(sd[k], ref[k]), k
    x = g["x"].clone().requires_grad_(True)
    y = model(x)
    assert torch.allclose(y, g["y"], rtol=1e-4, atol=1e-5)
    names = [k for k, _ in model.named_parameters()]
    grads = torch.autograd.grad(y, [x] + list(model.parameters()), g["gy"])
    assert torch.allclose(grads[0], g["gx"], rtol=1e-3, atol=2e-4)
    for k, gr in zip(names, grads[1:]):
        r = g["grad:" + k]
        assert (gr - r).abs().max().item() <= 1e-3 * (r.abs().max().item() + 1e-6), k
