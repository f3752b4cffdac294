"""Generate tests/golden/*.npz by running the *reference* (pashtari/factorizer, mounted
read-only at /root/reference) on CPU in the build container.

Only inputs / outputs / state_dicts (data) are written; no reference source travels.
Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py
The reference imports `opt_einsum` (matrix_factorization.py:8) but never uses it; it is not
installed here, so an empty stub module stands in for the import.
"""
import hashlib
import os
import sys
import types

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
sys.modules.setdefault("opt_einsum", types.ModuleType("opt_einsum"))
sys.path.insert(0, "/root/reference")
import factorizer as ft  # noqa: E402  (the reference)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def sd_arrays(module, prefix="sd:"):
    return {prefix + k: v for k, v in module.state_dict().items()}


# ---- G1: SWMatricize forward + inverse ---------------------------------------------------
def g1():
    cases = {
        "a": dict(shape=(1, 16, 16, 16, 16), kw=dict(head_dim=8, patch_size=8)),
        "b": dict(shape=(1, 8, 8, 8, 8), kw=dict(head_dim=4, patch_size=4, shifts=[None, 1, 2, 3])),
        "c": dict(shape=(1, 8, 8, 16, 4), kw=dict(head_dim=8, patch_size=(4, 8, 2))),
        "d": dict(shape=(2, 16, 8, 8, 8), kw=dict(num_heads=8, patch_size=4)),
        "e": dict(shape=(1, 8, 8, 8, 8), kw=dict(head_dim=8, patch_size=4, shifts=[None, (1, 2, 3), 2])),
    }
    arrs = {}
    for name, c in cases.items():
        shape = c["shape"]
        x = torch.arange(int(np.prod(shape)), dtype=torch.float32).reshape(shape)
        m = ft.SWMatricize((None, *shape[1:]), **c["kw"])
        y = m(x)
        z = m.inverse_forward(y)
        torch.manual_seed(7)
        yr = torch.rand_like(y)
        zr = m.inverse_forward(yr)
        arrs[f"{name}:y"] = y.to(torch.int32)  # exact integers (arange < 2^24)
        arrs[f"{name}:z"] = z
        if name != "a":
            arrs[f"{name}:yr"] = yr
            arrs[f"{name}:zr"] = zr
        arrs[f"{name}:output_size"] = np.array([-1 if s is None else s for s in m.output_size])
    # SHA-256 of the cfg-2 sized output for a seeded input (hash only)
    torch.manual_seed(0)
    x = torch.rand(1, 32, 128, 128, 128)
    m = ft.SWMatricize((None, 32, 128, 128, 128), head_dim=8, patch_size=8)
    y = m(x)
    arrs["cfg2:sha256_y"] = np.frombuffer(hashlib.sha256(y.numpy().tobytes()).digest(), dtype=np.uint8)
    arrs["cfg2:shape_y"] = np.array(y.shape)
    z = m.inverse_forward(y)
    arrs["cfg2:inverse_equal_x"] = np.array([int(torch.equal(z, x))])
    npz("g1_swmatricize", **arrs)


# ---- G2..G4: NMF ----------------------------------------------------------------------------
def run_nmf(name, arrs, shape, x=None, zero_first=False, **kw):
    torch.manual_seed(0)
    nmf = ft.NMF(size=shape[-2:], **kw)
    if x is None:
        x = torch.rand(*shape)
    if zero_first:
        x.view(-1, *shape[-2:])[0].zero_()
    x = x.clone().requires_grad_(True)
    u, v = nmf.decompose(x)
    y = nmf(x)
    torch.manual_seed(1)
    gy = torch.rand_like(y)
    (gx,) = torch.autograd.grad(y, x, gy)
    arrs.update({f"{name}:x": x, f"{name}:u0": nmf.init.u0, f"{name}:v0": nmf.init.v0,
                 f"{name}:u": u, f"{name}:v": v, f"{name}:y": y, f"{name}:gy": gy,
                 f"{name}:gx": gx,
                 f"{name}:loss": nmf.loss(x, u, v)})


def g2_g4():
    arrs = {}
    run_nmf("cfg1_mu_r2_t5", arrs, (1, 8, 512), rank=2, num_iters=5, init="uniform", solver="mu")
    run_nmf("cfg2_hals_r1_t5", arrs, (3, 8, 512), rank=1, num_iters=5, init="uniform", solver="hals")
    run_nmf("hals_r2_t10_8x512", arrs, (2, 8, 512), rank=2, num_iters=10, init="uniform", solver="hals")
    for R in (1, 2, 3):
        for T in (5, 10):
            run_nmf(f"hals_r{R}_t{T}", arrs, (2, 3, 8, 64), zero_first=True, rank=R,
                    num_iters=T, init="uniform", solver="hals")
            run_nmf(f"mu_r{R}_t{T}", arrs, (2, 3, 8, 64), zero_first=True, rank=R,
                    num_iters=T, init="uniform", solver="mu")
    run_nmf("hals_r2_t5_g1", arrs, (2, 3, 8, 64), rank=2, num_iters=5, num_grad_steps=1,
            init="uniform", solver="hals")
    run_nmf("mu_r2_t5_g2", arrs, (2, 3, 8, 64), rank=2, num_iters=5, num_grad_steps=2,
            init="uniform", solver="mu")
    run_nmf("hals_r1_t5_g1", arrs, (2, 3, 8, 64), rank=1, num_iters=5, num_grad_steps=1,
            init="uniform", solver="hals")
    run_nmf("test_nmf_shape", arrs, (2, 4, 8, 16), rank=3, init="uniform", solver="hals")
    run_nmf("rank_auto", arrs, (2, 8, 512), init="uniform", solver="hals")  # rank=None -> 1
    run_nmf("heads8_m4_n64", arrs, (4, 2, 4, 64), rank=1, init="uniform", solver="hals")
    npz("g2_nmf", **arrs)


# ---- G5: FactorizerBlock ------------------------------------------------------------------
def g5():
    arrs = {}
    for name, kw, C, S in [
        ("hals_r1", dict(rank=1, num_iters=5, solver="hals"), 16, (8, 8, 8)),
        ("mu_r2", dict(rank=2, num_iters=3, solver="mu"), 16, (8, 8, 8)),
    ]:
        torch.manual_seed(0)
        blk = ft.FactorizerBlock(
            channels=C, spatial_size=S, norm=ft.LayerNorm,
            reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}), act=nn.ReLU,
            factorize=ft.NMF, init="uniform", mlp_ratio=2, dropout=0.0, **kw)
        x = torch.randn(2, C, *S).requires_grad_(True)
        y = blk(x)
        torch.manual_seed(1)
        gy = torch.rand_like(y)
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), gy)
        arrs.update({f"{name}:x": x, f"{name}:y": y, f"{name}:gy": gy, f"{name}:gx": grads[0]})
        arrs.update(sd_arrays(blk, f"{name}:sd:"))
        for (k, _), g in zip(blk.named_parameters(), grads[1:]):
            arrs[f"{name}:grad:{k}"] = g
    npz("g5_block", **arrs)


# ---- G6: tiny Factorizer model ------------------------------------------------------------
def g6():
    arrs = {}
    torch.manual_seed(0)
    model = ft.Factorizer(
        in_channels=4, out_channels=3, spatial_size=(16, 16, 16),
        encoder_depth=(1, 1, 1), encoder_width=(8, 16, 32), strides=(1, 2, 2),
        decoder_depth=(1, 1), norm=ft.LayerNorm,
        reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}), act=nn.ReLU,
        factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals",
        mlp_ratio=2, dropout=0.1).eval()
    x = torch.rand(2, 4, 16, 16, 16).requires_grad_(True)
    y = model(x)
    torch.manual_seed(1)
    gy = torch.rand_like(y)
    params = dict(model.named_parameters())
    grads = torch.autograd.grad(y, [x] + list(params.values()), gy)
    arrs.update({"x": x, "y": y, "gy": gy, "gx": grads[0]})
    arrs.update(sd_arrays(model))
    for k, g in zip(params.keys(), grads[1:]):
        arrs[f"grad:{k}"] = g
    arrs["num_params"] = np.array([sum(p.numel() for p in model.parameters())])
    npz("g6_model", **arrs)

    # state_dict key/shape inventory of the README model (keys + shapes only, no values)
    torch.manual_seed(0)
    big = ft.Factorizer(
        in_channels=4, out_channels=3, spatial_size=(128, 128, 128),
        encoder_depth=(1, 1, 1, 1, 1), encoder_width=(32, 64, 128, 256, 512),
        strides=(1, 2, 2, 2, 2), decoder_depth=(1, 1, 1, 1), norm=ft.LayerNorm,
        reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
        factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals",
        mlp_ratio=2, dropout=0.1)
    inv = {k: np.array(v.shape) for k, v in big.state_dict().items()}
    inv["__num_params__"] = np.array([sum(p.numel() for p in big.parameters())])
    npz("g6_readme_model_keys", **inv)


# ---- G7: conv / layer micro-cases -----------------------------------------------------------
def g7():
    arrs = {}

    def run(name, mod, x):
        x = x.clone().requires_grad_(True)
        y = mod(x)
        torch.manual_seed(1)
        gy = torch.rand_like(y)
        params = dict(mod.named_parameters())
        grads = torch.autograd.grad(y, [x] + list(params.values()), gy)
        arrs.update({f"{name}:x": x, f"{name}:y": y, f"{name}:gy": gy, f"{name}:gx": grads[0]})
        arrs.update(sd_arrays(mod, f"{name}:sd:"))
        for k, g in zip(params.keys(), grads[1:]):
            arrs[f"{name}:grad:{k}"] = g

    torch.manual_seed(0)
    run("conv_k2s2", nn.Conv3d(8, 16, kernel_size=2, stride=2), torch.randn(2, 8, 8, 8, 8))
    run("tconv_k2s2", nn.ConvTranspose3d(16, 8, kernel_size=2, stride=2), torch.randn(2, 16, 4, 4, 4))
    run("conv_k3", nn.Conv3d(4, 8, kernel_size=3, padding=1, bias=False), torch.randn(2, 4, 8, 8, 8))
    run("conv_k1", nn.Conv3d(8, 3, kernel_size=1), torch.randn(2, 8, 4, 4, 4))
    run("linear", ft.Linear(16, 24), torch.randn(2, 16, 4, 4, 4))
    run("linear_nobias", ft.Linear(16, 16, bias=False), torch.randn(2, 16, 4, 4, 4))
    ln = ft.LayerNorm(16)
    with torch.no_grad():
        ln.norm.weight.uniform_(0.5, 1.5)
        ln.norm.bias.uniform_(-0.5, 0.5)
    run("layernorm", ln, torch.randn(2, 16, 4, 4, 4) * 2 + 0.3)
    run("mlp", ft.MLP(16, ratio=2), torch.randn(2, 16, 4, 4, 4))
    run("posembed", ft.PositionalEmbedding(8, (4, 4, 4)), torch.randn(2, 8, 4, 4, 4))
    npz("g7_layers", **arrs)


def g8():
    """the remaining solver / initialiser keys of the dispatch tables
    (matrix_factorization.py:581-618) on (3, 8, 24), rank 2, 3 iterations."""
    arrs = {}
    cases = {"fmu": dict(solver="fmu", init="uniform"), "smu": dict(solver="smu", init="uniform"),
             "ls": dict(solver="ls", init="uniform"), "nnls": dict(solver="nnls", init="uniform"),
             "cd": dict(solver="cd", init="normal"), "nncd": dict(solver="nncd", init="uniform"),
             "mu_0": dict(solver="mu-0", init="uniform"), "hals_1": dict(solver="hals-1", init="uniform"),
             "compose_mu_hals": dict(solver=["mu", "hals"], init="uniform"),
             "compose_ls1_nnls0": dict(solver=["ls-1", "nnls-0"], init="uniform-normal"),
             "mu_svd": dict(solver="mu", init="svd"), "hals_nndsvd": dict(solver="hals", init="nndsvd"),
             "ls_tall": dict(solver="ls", init="uniform", shape=(2, 24, 8))}
    for name, kw in cases.items():
        kw = dict(kw)
        shape = kw.pop("shape", (3, 8, 24))
        torch.manual_seed(0)
        mf = ft.MatrixFactorization(size=shape[-2:], rank=2, num_iters=3, **kw)
        torch.manual_seed(5)
        x = (torch.rand(*shape) if kw["solver"] != "cd" else torch.randn(*shape)).requires_grad_(True)
        u0, v0 = mf.init(x)
        u, v = mf.decompose(x)
        y = mf(x)
        arrs.update({f"{name}:x": x, f"{name}:u_init": u0, f"{name}:v_init": v0, f"{name}:u": u, f"{name}:v": v,
                     f"{name}:y": y})
        if kw["init"] not in ("svd", "nndsvd"):
            torch.manual_seed(1)
            gy = torch.rand_like(y)
            (gx,) = torch.autograd.grad(y, x, gy)
            arrs.update({f"{name}:gy": gy, f"{name}:gx": gx})
    # weighted multiplicative update: the weights travel through decompose(x, w)
    torch.manual_seed(0)
    mf = ft.NMF(size=(8, 24), rank=2, num_iters=3, init="uniform", solver="wmu")
    torch.manual_seed(5)
    x, w = torch.rand(3, 8, 24), torch.rand(3, 8, 24)
    u, v = mf.decompose(x, w)
    arrs.update({"wmu:x": x, "wmu:w": w, "wmu:u": u, "wmu:v": v, "wmu:loss": mf.loss(x, u, v, w)})
    # SVD layer
    svd = ft.SVD(size=(8, 24), rank=3) if hasattr(ft, "SVD") else None
    if svd is None:
        from factorizer.factorization.matrix_factorization import SVD
        svd = SVD(size=(8, 24), rank=3)
    arrs.update({"svd:x": x, "svd:y": svd(x)})
    npz("g8_solvers", **arrs)


# ---- G9: Deconver family (factorization/deconvolution.py:60-260, deconver.py:9-230) --------
def g9():
    arrs = {}

    def grads(mod, x, y, tag):
        torch.manual_seed(5)
        gy = torch.rand_like(y)
        names = [k for k, _ in mod.named_parameters()]
        gs = torch.autograd.grad(y, [x] + list(mod.parameters()), gy, allow_unused=True)
        arrs[f"{tag}:gy"], arrs[f"{tag}:gx"] = gy, gs[0]
        for k, g in zip(names, gs[1:]):
            if g is not None:
                arrs[f"{tag}:grad:{k}"] = g

    # Deconv: the reference's own test shape (tests/test_deconver.py:16-40) and variants
    for tag, S, kw in [
        ("deconv2d_src", (1, 20, 12, 12), dict(channels=20, ratio=2, groups=5, kernel_size=(3, 3), num_iters=3)),
        ("deconv2d_both", (2, 20, 12, 12), dict(channels=20, ratio=2, groups=5, kernel_size=(3, 3), update_source=True,
                                               update_filter=True, num_iters=3)),
        ("deconv3d_dw", (2, 8, 6, 6, 8), dict(channels=8, ratio=2, groups=-1, kernel_size=(3, 3, 3), num_iters=2)),
        ("deconv3d_g1_k5", (1, 8, 6, 6, 8), dict(channels=8, ratio=1, groups=1, kernel_size=(5, 3, 3), num_iters=2,
                                                  num_grad_iters=1)),
    ]:
        torch.manual_seed(0)
        m = ft.Deconv(**kw)
        torch.manual_seed(1)
        x = torch.rand(*S, requires_grad=True)
        y = m(x)
        for k, v in sd_arrays(m, f"{tag}:sd:").items():
            arrs[k] = v
        arrs[f"{tag}:x"], arrs[f"{tag}:y"] = x, y
        grads(m, x, y, tag)
        with torch.no_grad():
            s, h = m.fit(x)
            arrs[f"{tag}:fit_s"], arrs[f"{tag}:fit_h"] = s, h
            arrs[f"{tag}:recon"] = m.reconstruct(s, h)
            arrs[f"{tag}:loss"] = m.loss(*[t for t in (m.split_channels(x), m.split_channels(s), m.split_channels(h))]) \
                if m.groups != 1 else m.loss(x, s, h)
    # DeconverBlock / DeconverStage (tests/test_deconver.py:76-120)
    torch.manual_seed(0)
    blk = ft.DeconverBlock(channels=16, kernel_size=(3, 3), num_iters=3, num_grad_iters=1, mlp_ratio=3)
    torch.manual_seed(1)
    x = torch.rand(2, 16, 12, 12, requires_grad=True)
    y = blk(x)
    for k, v in sd_arrays(blk, "block2d:sd:").items():
        arrs[k] = v
    arrs["block2d:x"], arrs["block2d:y"] = x, y
    grads(blk, x, y, "block2d")
    torch.manual_seed(0)
    st = ft.DeconverStage(in_channels=8, out_channels=16, depth=2, kernel_size=(3, 3, 3), num_iters=2, mlp_ratio=2)
    torch.manual_seed(1)
    x = torch.rand(1, 8, 6, 6, 8, requires_grad=True)
    y = st(x)
    for k, v in sd_arrays(st, "stage3d:sd:").items():
        arrs[k] = v
    arrs["stage3d:x"], arrs["stage3d:y"] = x, y
    grads(st, x, y, "stage3d")
    # tiny Deconver models, 2-D (tests/test_deconver.py:122-160 form with groups=-1; its ratio 0.5 gives round(0.5) = 0 source channels and cannot run: ratio 1 here) and 3-D
    for tag, S, kw in [
        ("model2d", (1, 4, 16, 16), dict(spatial_dims=2, encoder_width=(8, 16, 32), encoder_depth=(1, 1, 1), strides=(1, 2, 2),
                                         decoder_depth=(1, 1), act=nn.ReLU, groups=-1, ratio=1, kernel_size=(3, 3),
                                         num_iters=3, num_grad_iters=1, mlp_ratio=2, dropout=0.0)),
        ("model3d", (1, 4, 8, 8, 8), dict(spatial_dims=3, encoder_width=(8, 16), encoder_depth=(1, 1), strides=(1, 2),
                                          decoder_depth=(1,), act=nn.ReLU, groups=4, ratio=2, kernel_size=(3, 3, 3),
                                          num_iters=2, mlp_ratio=2, dropout=0.0)),
    ]:
        torch.manual_seed(0)
        model = ft.Deconver(in_channels=4, out_channels=3, **kw).eval()
        torch.manual_seed(1)
        x = torch.rand(*S, requires_grad=True)
        y = model(x)
        for k, v in sd_arrays(model, f"{tag}:sd:").items():
            arrs[k] = v
        arrs[f"{tag}:x"], arrs[f"{tag}:y"] = x, y
        grads(model, x, y, tag)
    npz("g9_deconver", **arrs)


# ---- G10: the N-D generic path on 2-D and 1-D tensors (operations.py:318-325; the reference's own tests build 2-D models) ----
G10_BLOCKS = {
    "blk2d_hals_r1": (dict(rank=1, num_iters=5, solver="hals"), 16, (16, 16), 4),
    "blk2d_mu_r2": (dict(rank=2, num_iters=3, solver="mu"), 16, (16, 24), (4, 8)),
    "blk2d_c32_p8": (dict(rank=1, num_iters=5, solver="hals"), 32, (32, 32), 8),       # 64-voxel patches, both fused-GEMM widths
    "blk1d_hals_r1": (dict(rank=1, num_iters=5, solver="hals"), 16, (64,), 16),
}


def g10():
    arrs = {}
    for name, (kw, C, S, patch) in G10_BLOCKS.items():
        torch.manual_seed(0)
        blk = ft.FactorizerBlock(
            channels=C, spatial_size=S, norm=ft.LayerNorm,
            reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": patch}), act=nn.ReLU,
            factorize=ft.NMF, init="uniform", mlp_ratio=2, dropout=0.0, **kw)
        x = torch.randn(2, C, *S).requires_grad_(True)
        y = blk(x)
        torch.manual_seed(1)
        gy = torch.rand_like(y)
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), gy)
        arrs.update({f"{name}:x": x, f"{name}:y": y, f"{name}:gy": gy, f"{name}:gx": grads[0]})
        arrs.update(sd_arrays(blk, f"{name}:sd:"))
        for (k, _), g in zip(blk.named_parameters(), grads[1:]):
            arrs[f"{name}:grad:{k}"] = g
    # a 2-D Factorizer (Conv2d stem / down / up / head from the reference's N-D generic UNet)
    torch.manual_seed(0)
    model = ft.Factorizer(
        in_channels=3, out_channels=2, spatial_size=(32, 32),
        encoder_depth=(1, 1), encoder_width=(16, 32), strides=(1, 2),
        decoder_depth=(1,), norm=ft.LayerNorm,
        reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 4}), act=nn.ReLU,
        factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals",
        mlp_ratio=2, dropout=0.1).eval()
    x = torch.rand(2, 3, 32, 32).requires_grad_(True)
    y = model(x)
    torch.manual_seed(1)
    gy = torch.rand_like(y)
    params = dict(model.named_parameters())
    grads = torch.autograd.grad(y, [x] + list(params.values()), gy)
    arrs.update({"model2d:x": x, "model2d:y": y, "model2d:gy": gy, "model2d:gx": grads[0]})
    arrs.update(sd_arrays(model, "model2d:sd:"))
    for k, g in zip(params.keys(), grads[1:]):
        arrs[f"model2d:grad:{k}"] = g
    npz("g10_lower_d", **arrs)


if __name__ == "__main__":
    g1()
    g2_g4()
    g5()
    g6()
    g7()
    g8()
    g9()
    g10()
