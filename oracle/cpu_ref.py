"""TEST INFRASTRUCTURE ONLY — CPU oracle for the Factorizer hot path.

A from-scratch restatement, in plain PyTorch-CPU ops, of the arithmetic the reference
(pashtari/factorizer @ 2025-02-27) performs on the path named by BASELINE.json:

    FactorizerBlock = LayerNorm -> in_proj -> SWMatricize -> ReLU -> NMF -> inverse SWMatricize
                      -> out_proj -> residual ; LayerNorm -> MLP -> residual
    plus the strided 3-D convolutions of the U-shaped encoder/decoder.

Every function cites the reference file:line (relative to the reference repo root) it
restates.  The restatement is written from the index formulas / update rules,
not from the reference's einops/roll/bmm call sequence.

PARITY PINNING: the reference's own tests hold no numeric vectors for this path
(tests/test_nmf.py:14-39 and tests/test_factorizer.py:41-47 assert shape / finiteness / >=0
only).  This oracle is therefore pinned against outputs of the reference itself, generated in
the build container by ``tools/make_goldens.py`` (imports /root/reference) and committed as
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function here against
those vectors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (factorizer_amd/) never does.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

EPS = 1e-16  # matrix_factorization.py:200,236


# ----------------------------------------------------------------------------------------
# SWMatricize  (factorization/operations.py:266-272, 299-355, 381-434)
# ----------------------------------------------------------------------------------------
def _ntuple(v, n):
    if v is None:
        return (None,) * n
    if isinstance(v, (tuple, list)):
        assert len(v) == n
        return tuple(v)
    return (v,) * n


def matricize_geometry(channels, spatial, num_heads=None, head_dim=None, grid_size=None,
                       patch_size=None):
    """Resolve (h, d, grid, patch) the way Matricize/Reshape.infer_dims does
    (operations.py:199-236, 328-339): unknown member of each group = size // known."""
    nd = len(spatial)
    assert (num_heads, head_dim) != (None, None)
    if head_dim is not None:
        d = max(head_dim, 1)
        h = max(num_heads, 1) if num_heads is not None else channels // d
    else:
        h = max(num_heads, 1)
        d = channels // h
    grid, patch = [], []
    gs, ps = _ntuple(grid_size, nd), _ntuple(patch_size, nd)
    for s, g, p in zip(spatial, gs, ps):
        assert (g, p) != (None, None)
        if p is not None:
            p = max(p, 1)
            g = max(g, 1) if g is not None else s // p
        else:
            g = max(g, 1)
            p = s // g
        grid.append(g)
        patch.append(p)
    return h, d, tuple(grid), tuple(patch)


def default_shifts(patch):
    """operations.py:397-398: [None, tuple(p // 2 for p in patch_size)]."""
    return [None, tuple(p // 2 for p in patch)]


def _norm_shift(s, nd):
    if s is None:
        return (0,) * nd
    if isinstance(s, (tuple, list)):
        return tuple(int(v) for v in s)
    return (int(s),) * nd


def swm_gather_index(B, C, spatial, h, d, grid, patch, shift):
    """Flat index into x.reshape(-1) for one window, shape (B*h, G, d, P).

    y[b*h+hh, (g0,g1,g2), dd, (p0,p1,p2)] = x[b, hh*d+dd, (g_i*p_i + p_i - s_i) mod S_i]
    — torch.roll(x, +s) then 'b (h d) (g0 p0).. -> (b h) (g0..) d (p0..)'
    (operations.py:268-271, 321-325)."""
    nd = len(spatial)
    shift = _norm_shift(shift, nd)
    coords = []
    for i in range(nd):
        g = torch.arange(grid[i]).view(-1, 1)
        p = torch.arange(patch[i]).view(1, -1)
        coords.append((g * patch[i] + p - shift[i]) % spatial[i])  # (g_i, p_i)
    # spatial flat offset with shape (g0,g1,..,p0,p1,..)
    strides = [1] * nd
    for i in range(nd - 2, -1, -1):
        strides[i] = strides[i + 1] * spatial[i + 1]
    off = torch.zeros([1] * (2 * nd), dtype=torch.long)
    for i in range(nd):
        shp = [1] * (2 * nd)
        shp[i] = grid[i]
        shp[nd + i] = patch[i]
        off = off + (coords[i] * strides[i]).view(shp)
    G = math.prod(grid)
    P = math.prod(patch)
    off = off.reshape(G, 1, P)
    V = math.prod(spatial)
    b = torch.arange(B).view(B, 1, 1, 1, 1)
    hh = torch.arange(h).view(1, h, 1, 1, 1)
    dd = torch.arange(d).view(1, 1, 1, d, 1)
    idx = (b * C + hh * d + dd) * V + off.view(1, 1, G, 1, P)
    return idx.reshape(B * h, G, d, P)


def swm_forward(x, num_heads=None, head_dim=None, grid_size=None, patch_size=None,
                shifts=None):
    """SWMatricize.forward (operations.py:417-421): windows concatenated on dim 0."""
    B, C = x.shape[:2]
    spatial = tuple(x.shape[2:])
    h, d, grid, patch = matricize_geometry(C, spatial, num_heads, head_dim, grid_size,
                                           patch_size)
    if shifts is None:
        shifts = default_shifts(patch)
    flat = x.reshape(-1)
    outs = [flat[swm_gather_index(B, C, spatial, h, d, grid, patch, s)] for s in shifts]
    return torch.cat(outs, dim=0)


def swm_inverse(y, channels, spatial, num_heads=None, head_dim=None, grid_size=None,
                patch_size=None, shifts=None):
    """SWMatricize.inverse_forward (operations.py:423-434):
    out = (((0.0 + z_0) + z_1) + ...) / num_shifts, z_j = inverse window j of chunk j."""
    h, d, grid, patch = matricize_geometry(channels, spatial, num_heads, head_dim,
                                           grid_size, patch_size)
    if shifts is None:
        shifts = default_shifts(patch)
    nw = len(shifts)
    chunk = y.shape[0] // nw
    B = chunk // h
    V = math.prod(spatial)
    out = 0.0
    for j, s in enumerate(shifts):
        idx = swm_gather_index(B, channels, spatial, h, d, grid, patch, s).reshape(-1)
        z = torch.empty(B * channels * V, dtype=y.dtype)
        z[idx] = y[j * chunk:(j + 1) * chunk].reshape(-1)  # idx is a permutation
        out = out + z
    out = out / nw
    return out.reshape(B, channels, *spatial)


# ----------------------------------------------------------------------------------------
# NMF  (factorization/matrix_factorization.py)
# ----------------------------------------------------------------------------------------
def nmf_rank(M, N, rank=None, compression=10):
    """matrix_factorization.py:488-491."""
    if rank is None:
        return max(math.ceil(M * N / (compression * (M + N))), 1)
    return rank


def mu_update(z, w, s, eps=EPS):
    """MultiplicativeUpdate.update_u (matrix_factorization.py:241-247) for z ≈ w sᵀ."""
    a = z @ s
    b = s.mT @ s
    return (w * a + eps) / (w @ b + eps)


def hals_update(z, w, s, eps=EPS):
    """CoordinateDescent.update_u with project=ReLU (matrix_factorization.py:210-229)."""
    R = w.shape[-1]
    a = z @ s
    b = s.mT @ s
    if R == 1:
        return torch.relu((a + eps) / (b + eps))
    cols = [w[..., r] for r in range(R)]
    for r in range(R):
        acc = None
        for j in range(R):
            if j == r:
                continue
            t = cols[j] * b[..., j, r].unsqueeze(-1)
            acc = t if acc is None else acc + t
        num = a[..., r] - acc + eps
        den = b[..., r, r].unsqueeze(-1) + eps
        cols[r] = torch.relu(num / den)
    return torch.stack(cols, dim=-1)


_UPDATES = {"mu": mu_update, "hals": hals_update}


def nmf_decompose(x, u0, v0, num_iters=5, solver="hals", num_grad_steps=None, eps=EPS):
    """MatrixFactorization.decompose (matrix_factorization.py:514-530) with RandomInit
    broadcast (``:52-58``) and BCDSolver alternation U then V, V sees the new U (``:122-136``).
    Iterations it < T - num_grad_steps + 1 run under no_grad (``:506-512``)."""
    upd = _UPDATES[solver]
    T = num_iters
    G = T if num_grad_steps is None else num_grad_steps
    lead = x.shape[:-2]
    with torch.no_grad():
        u = u0.expand(*lead, *u0.shape)
        v = v0.expand(*lead, *v0.shape)
    for it in range(1, T + 1):
        ctx = torch.no_grad() if it < T - G + 1 else _Null()
        with ctx:
            u = upd(x, u, v, eps)
            v = upd(x.mT, v, u, eps)
    return u, v


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def nmf_forward(x, u0, v0, num_iters=5, solver="hals", num_grad_steps=None, eps=EPS):
    """MatrixFactorization.forward = reconstruct(decompose(x)) = u @ v.mT (``:532-546``)."""
    u, v = nmf_decompose(x, u0, v0, num_iters, solver, num_grad_steps, eps)
    return u @ v.mT


def relative_error(x, y, eps=EPS):
    """operations.py:99-122 with norm2 (``:34-51``): per leading-batch relative L2 error."""
    num = torch.sqrt((x - y).flatten(1).square().sum(1)) + eps
    den = torch.sqrt(x.flatten(1).square().sum(1)) + eps
    return num / den


def hals_gate_margin(x, u0, v0, num_iters=5, eps=EPS):
    """Test helper: per matrix, the smallest |pre-activation| of any HALS ReLU gate relative
    to its leading term a/b, over all iterations (float64).  The gradient of HALS is
    discontinuous where a pre-activation crosses 0 (ReLU gate [w > 0] of SURVEY.md
    Appendix A): a matrix whose margin is within fp32 rounding (≲1e-6) has no well-defined
    fp32 gradient, and parity tests exclude it."""
    x = x.double()
    lead = x.shape[:-2]
    u = u0.double().expand(*lead, *u0.shape)
    v = v0.double().expand(*lead, *v0.shape)
    R = u0.shape[1]
    margin = torch.full(lead, float("inf"), dtype=torch.float64)

    def half(z, w, s):
        nonlocal margin
        a = z @ s
        b = s.mT @ s
        cols = [w[..., r] for r in range(R)]
        for r in range(R):
            acc = torch.zeros_like(cols[0])
            for j in range(R):
                if j != r:
                    acc = acc + cols[j] * b[..., j, r].unsqueeze(-1)
            den = b[..., r, r].unsqueeze(-1) + eps
            q = (a[..., r] - acc + eps) / den
            ref = (a[..., r].abs() + acc.abs()) / den + 1e-300
            margin = torch.minimum(margin, (q.abs() / ref).amin(-1))
            cols[r] = torch.relu(q)
        return torch.stack(cols, dim=-1)

    for _ in range(num_iters):
        u = half(x, u, v)
        v = half(x.mT, v, u)
    return margin


# ---- hand-derived backward, used to validate the HIP backward ----
def _half_bwd_mu(z, w, s, w_new, gw_new, eps):
    a = z @ s
    b = s.mT @ s
    dn = w @ b + eps
    gn = gw_new / dn
    gdn = -gw_new * w_new / dn
    gw = gn * a + gdn @ b.mT
    ga = gn * w
    gb = w.mT @ gdn
    gz = ga @ s.mT
    gs = z.mT @ ga + s @ (gb + gb.mT)
    return gw, gs, gz


def _half_bwd_hals(z, w, s, w_new, gw_new, eps):
    R = w.shape[-1]
    a = z @ s
    b = s.mT @ s
    gwn = [gw_new[..., r].clone() for r in range(R)]   # grads of new columns (modified)
    gwo = [torch.zeros_like(gwn[0]) for _ in range(R)]  # grads of old columns
    ga = torch.zeros_like(a)
    gb = torch.zeros_like(b)
    for r in range(R - 1, -1, -1):
        den = b[..., r, r].unsqueeze(-1) + eps
        gq = gwn[r] * (w_new[..., r] > 0).to(z.dtype)
        gnum = gq / den
        gb[..., r, r] += -(gq * w_new[..., r]).sum(-1) / den.squeeze(-1)
        ga[..., r] += gnum
        for j in range(R):
            if j == r:
                continue
            what_j = w_new[..., j] if j < r else w[..., j]
            gb[..., j, r] += -(gnum * what_j).sum(-1)
            if j < r:
                gwn[j] = gwn[j] - gnum * b[..., j, r].unsqueeze(-1)
            else:
                gwo[j] = gwo[j] - gnum * b[..., j, r].unsqueeze(-1)
    gw = torch.stack(gwo, dim=-1)
    gz = ga @ s.mT
    gs = z.mT @ ga + s @ (gb + gb.mT)
    return gw, gs, gz


def nmf_backward(x, u0, v0, gy, num_iters=5, solver="hals", num_grad_steps=None, eps=EPS):
    """dL/dx of ``nmf_forward`` given dL/dy, by the reverse sweep of SURVEY.md Appendix A
    (what autograd does through matrix_factorization.py:522-533, written out)."""
    upd = _UPDATES[solver]
    half = {"mu": _half_bwd_mu, "hals": _half_bwd_hals}[solver]
    T = num_iters
    G = T if num_grad_steps is None else num_grad_steps
    lead = x.shape[:-2]
    with torch.no_grad():
        us = [u0.expand(*lead, *u0.shape)]
        vs = [v0.expand(*lead, *v0.shape)]
        for _ in range(T):
            un = upd(x, us[-1], vs[-1], eps)
            vn = upd(x.mT, vs[-1], un, eps)
            us.append(un)
            vs.append(vn)
        gu = gy @ vs[T]
        gv = gy.mT @ us[T]
        gx = torch.zeros_like(x)
        for t in range(T, max(T - G, 0), -1):
            # undo V-update: v_t = upd(xᵀ, v_{t-1}, u_t)
            gv_old, gs, gz = half(x.mT, vs[t - 1], us[t], vs[t], gv, eps)
            gu = gu + gs
            gx = gx + gz.mT
            # undo U-update: u_t = upd(x, u_{t-1}, v_{t-1})
            gu_old, gs, gz = half(x, us[t - 1], vs[t - 1], us[t], gu, eps)
            gv = gv_old + gs
            gx = gx + gz
            gu = gu_old
    return gx


# ----------------------------------------------------------------------------------------
# channels-first layers (layers/norm.py:29-34, linear.py:53-58, mlp.py:54-63, pos_embed.py:89)
# ----------------------------------------------------------------------------------------
def layernorm_cf(x, weight, bias, eps=1e-5):
    """LayerNorm over the channel dim of a channels-first tensor (norm.py:29-34)."""
    xt = x.movedim(1, -1)
    y = F.layer_norm(xt, (x.shape[1],), weight, bias, eps)
    return y.movedim(-1, 1)


def linear_cf(x, weight, bias=None):
    """Conv1d(k=1) on flatten(2) (linear.py:44-58); weight (out,in,1)."""
    B, C = x.shape[:2]
    w = weight.reshape(weight.shape[0], weight.shape[1])
    y = torch.matmul(w, x.reshape(B, C, -1))
    if bias is not None:
        y = y + bias.view(1, -1, 1)
    return y.reshape(B, w.shape[0], *x.shape[2:])


def mlp_cf(x, w1, b1, w2, b2):
    """MLP.block: Linear -> GELU(erf) -> Linear (mlp.py:54-60), dropout p=0."""
    return linear_cf(F.gelu(linear_cf(x, w1, b1)), w2, b2)


def fact_mixer(x, sd, prefix, cfg):
    """FactMixer.forward (factorizer.py:34-57) with SWMatricize / ReLU / NMF, dropout 0."""
    B, C = x.shape[:2]
    spatial = tuple(x.shape[2:])
    t = linear_cf(x, sd[prefix + "in_proj.linear.weight"])
    m = swm_forward(t, **cfg["reshape"])
    m = torch.relu(m)
    m = nmf_forward(m, sd[prefix + "factorize.init.u0"], sd[prefix + "factorize.init.v0"],
                    cfg.get("num_iters", 5), cfg.get("solver", "hals"),
                    cfg.get("num_grad_steps"))
    t = swm_inverse(m, C, spatial, **cfg["reshape"])
    return linear_cf(t, sd[prefix + "out_proj.linear.weight"],
                     sd[prefix + "out_proj.linear.bias"])


def factorizer_block(x, sd, prefix, cfg):
    """FactorizerBlock.forward (factorizer.py:74-77)."""
    y = layernorm_cf(x, sd[prefix + "norm1.norm.weight"], sd[prefix + "norm1.norm.bias"])
    x = x + fact_mixer(y, sd, prefix + "fact.", cfg)
    y = layernorm_cf(x, sd[prefix + "norm2.norm.weight"], sd[prefix + "norm2.norm.bias"])
    x = x + mlp_cf(y, sd[prefix + "mlp.block.0.linear.weight"],
                   sd[prefix + "mlp.block.0.linear.bias"],
                   sd[prefix + "mlp.block.3.linear.weight"],
                   sd[prefix + "mlp.block.3.linear.bias"])
    return x


class _StoreBF16(torch.autograd.Function):
    """A tensor kept in HBM as bf16: rounded (nearest-even) on the way forward, its gradient rounded on the
    way back — the storage points of the mixed-precision device path (FZ_STORE_BF16)."""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def factorizer_block_bf16_storage(x, sd, prefix, cfg):
    """`factorizer_block` in fp32 arithmetic with every tensor the device path stores between kernels rounded
    to bf16 (t, the matricized tensor and its gradient, the NMF output, the window average a, x1, z1, x2):
    the reference for the mixed-precision mode of BASELINE configs[4].  It separates what bf16 STORAGE does
    to the result (this function vs `factorizer_block`: conditioning of the block, large for T = 10 HALS
    iterations) from what the kernels do (device vs this function)."""
    q = _StoreBF16.apply
    f = prefix + "fact."
    B, C = x.shape[:2]
    spatial = tuple(x.shape[2:])
    y = layernorm_cf(x, sd[prefix + "norm1.norm.weight"], sd[prefix + "norm1.norm.bias"])
    t = q(torch.relu(linear_cf(y, sd[f + "in_proj.linear.weight"])))
    m = q(swm_forward(t, **cfg["reshape"]))
    m = q(nmf_forward(m, sd[f + "factorize.init.u0"], sd[f + "factorize.init.v0"], cfg.get("num_iters", 5),
                      cfg.get("solver", "hals"), cfg.get("num_grad_steps")))
    a = q(swm_inverse(m, C, spatial, **cfg["reshape"]))
    x1 = q(x + linear_cf(a, sd[f + "out_proj.linear.weight"], sd[f + "out_proj.linear.bias"]))
    y = layernorm_cf(x1, sd[prefix + "norm2.norm.weight"], sd[prefix + "norm2.norm.bias"])
    z1 = q(linear_cf(y, sd[prefix + "mlp.block.0.linear.weight"], sd[prefix + "mlp.block.0.linear.bias"]))
    return q(x1 + linear_cf(F.gelu(z1), sd[prefix + "mlp.block.3.linear.weight"],
                            sd[prefix + "mlp.block.3.linear.bias"]))


def factorizer_stage(x, sd, prefix, cfg, depth=1):
    """FactorizerStage.forward (factorizer.py:114-122); pos_drop is identity (eval / p=0)."""
    if prefix + "adapter.linear.weight" in sd:
        x = linear_cf(x, sd[prefix + "adapter.linear.weight"])
    if prefix + "pos_embed.pos" in sd:
        x = x + sd[prefix + "pos_embed.pos"]
    for k in range(depth):
        x = factorizer_block(x, sd, f"{prefix}blocks.{k}.", cfg)
    return x


# ----------------------------------------------------------------------------------------
# U-shape convolutions (unet.py:53,123,231,253; factorizer.py:145-149)
# ----------------------------------------------------------------------------------------
def conv_k2s2(x, weight, bias):
    """Conv3d(k=2, stride=2) (unet.py:53) restated as space-to-depth + GEMM."""
    B, C, D, H, W = x.shape
    O = weight.shape[0]
    xs = x.reshape(B, C, D // 2, 2, H // 2, 2, W // 2, 2)
    xs = xs.permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, -1, C * 8)
    y = xs @ weight.reshape(O, C * 8).mT + bias
    return y.reshape(B, D // 2, H // 2, W // 2, O).permute(0, 4, 1, 2, 3).contiguous()


def tconv_k2s2(x, weight, bias):
    """ConvTranspose3d(k=2, stride=2) (unet.py:123) restated as GEMM + depth-to-space;
    weight (Cin, Cout, 2,2,2)."""
    B, C, D, H, W = x.shape
    O = weight.shape[1]
    xs = x.permute(0, 2, 3, 4, 1).reshape(B, -1, C)
    y = xs @ weight.reshape(C, O * 8)
    y = y.reshape(B, D, H, W, O, 2, 2, 2).permute(0, 4, 1, 5, 2, 6, 3, 7)
    y = y.reshape(B, O, 2 * D, 2 * H, 2 * W) + bias.view(1, -1, 1, 1, 1)
    return y.contiguous()


def conv_k3(x, weight):
    """stem Conv3d(k=3, padding=1, bias=False) (factorizer.py:145-149)."""
    return F.conv3d(x, weight, None, stride=1, padding=1)


def conv_k1(x, weight, bias):
    """head Conv3d(k=1) (unet.py:226,253)."""
    return linear_cf(x, weight.reshape(weight.shape[0], weight.shape[1], 1), bias)


def factorizer_forward(x, sd, cfg):
    """Factorizer/UNet.forward (unet.py:260-276; factorizer.py:128-171), single head,
    eval mode.  cfg: {"widths", "strides", "encoder_depth", "decoder_depth",
    "reshape": {...}, "num_iters", "solver", "num_grad_steps"}."""
    widths = cfg["widths"]
    strides = cfg["strides"]
    edepth = cfg.get("encoder_depth", (1,) * len(widths))
    ddepth = cfg.get("decoder_depth", (1,) * (len(widths) - 1))
    x = conv_k3(x, sd["stem.weight"])
    feats = []
    for i in range(len(widths)):
        p = f"encoder.blocks.{i}."
        if strides[i] != 1:
            x = conv_k2s2(x, sd[p + "downsample.weight"], sd[p + "downsample.bias"])
        x = factorizer_stage(x, sd, p + "block.", cfg, edepth[i])
        feats.append(x)
    for j in range(len(ddepth)):
        p = f"decoder.blocks.{j}."
        up = tconv_k2s2(feats[-1 - j], sd[p + "upsample.weight"], sd[p + "upsample.bias"])
        cat = torch.cat([feats[-2 - j], up], dim=1)
        feats[-2 - j] = factorizer_stage(cat, sd, p + "block.", cfg, ddepth[j])
    return conv_k1(feats[0], sd["head.weight"], sd["head.bias"])


def dice_bce_loss(logits, target, smooth=1e-5):
    """Training loss used for cfg 4 timing: BCEWithLogits + soft Dice (sigmoid, squared
    denominators), the form of MONAI DiceCELoss(sigmoid=True, squared_pred=True) named at
    model_zoo/factorizer_brats23/configs/train.yaml:67-70.  MONAI is not importable here,
    so this loss is *unpinned* (timing only)."""
    p = torch.sigmoid(logits)
    dims = tuple(range(2, logits.ndim))
    inter = (p * target).sum(dims)
    den = (p * p).sum(dims) + (target * target).sum(dims)
    dice = 1.0 - (2.0 * inter + smooth) / (den + smooth)
    bce = F.binary_cross_entropy_with_logits(logits, target)
    return dice.mean() + bce


def dice_ce_loss(logits, target, smooth=1e-5):
    """The recipe's loss, `DiceCELoss(sigmoid=True, squared_pred=True)`
    (model_zoo/factorizer_brats23/configs/train.yaml:67-70), as MONAI 1.4.0 evaluates it
    (pinned at model_zoo/factorizer_brats23/docs/requirements.txt:11; third-party, absent from the
    tree — published algorithm restated: monai/losses/dice.py, `DiceLoss.forward` with
    sigmoid / squared_pred / smooth_nr = smooth_dr = 1e-5 / reduction mean over (b, c), and
    `DiceCELoss.forward`: `self.ce` = `nn.CrossEntropyLoss()(input, float target)` when the prediction
    has more than one channel, `self.bce` = `nn.BCEWithLogitsLoss()` for one channel).  The CE part IS
    torch's own function; the Dice part is unpinned against MONAI (not importable here)."""
    p = torch.sigmoid(logits)
    dims = tuple(range(2, logits.ndim))
    inter = (p * target).sum(dims)
    den = (p * p).sum(dims) + (target * target).sum(dims)
    dice = (1.0 - (2.0 * inter + smooth) / (den + smooth)).mean()
    if logits.shape[1] == 1:
        return dice + F.binary_cross_entropy_with_logits(logits, target)
    return dice + torch.nn.CrossEntropyLoss()(logits, target)


# ---- sliding-window inference ---------------------------------------------------------
def sliding_window_oracle(inputs, roi, sw_batch, predictor, overlap=0.5, mode="gaussian", sigma_scale=0.125):
    """Plain restatement of MONAI's published `sliding_window_inference` (monai/inferers/utils.py,
    the bundle pins it through SlidingWindowInfererAdapt, inference.yaml:96-102) with explicit loops and a
    DENSE importance map, for volumes at least as large as the roi: dense windows at interval
    int(roi*(1-overlap)) with the last one pulled back to the border; map = product of 1-D Gaussians
    (sigma = sigma_scale*roi) clamped at max(min, 1e-3); output = sum(map*net(window)) / sum(map).
    parity: unpinned against MONAI itself (absent from the image); pinned to its documented algorithm."""
    import itertools
    import math
    B = inputs.shape[0]
    size = tuple(inputs.shape[2:])
    roi = tuple(roi)
    assert all(s >= r for s, r in zip(size, roi))
    starts = []
    for im, r in zip(size, roi):
        it = r if r == im else max(int(r * (1 - overlap)), 1)
        n = int(math.ceil((im - r) / it)) + 1
        starts.append([min(i * it, im - r) for i in range(n)])
    if mode == "gaussian":
        w = torch.ones(roi, dtype=torch.float64)
        for ax, r in enumerate(roi):
            x = torch.arange(r, dtype=torch.float64) - (r - 1) / 2.0
            g = torch.exp(-(x ** 2) / (2 * (sigma_scale * r) ** 2)).to(torch.float32).to(torch.float64)
            shape = [1, 1, 1]
            shape[ax] = r
            w = w * g.view(shape)
        w = w.to(torch.float32)
        w = w.clamp_min(max(float(w[w != 0].min()), 1e-3))
    else:
        w = torch.ones(roi, dtype=torch.float32)
    out = cnt = None
    jobs = [(b, s) for b in range(B) for s in itertools.product(*starts)]
    for j0 in range(0, len(jobs), sw_batch):
        chunk = jobs[j0:j0 + sw_batch]
        win = torch.stack([inputs[b, :, s[0]:s[0] + roi[0], s[1]:s[1] + roi[1], s[2]:s[2] + roi[2]] for b, s in chunk])
        prob = predictor(win)
        if out is None:
            out = torch.zeros((B, prob.shape[1], *size), dtype=prob.dtype)
            cnt = torch.zeros((B, 1, *size), dtype=prob.dtype)
        for i, (b, s) in enumerate(chunk):
            sl = (slice(s[0], s[0] + roi[0]), slice(s[1], s[1] + roi[1]), slice(s[2], s[2] + roi[2]))
            out[(b, slice(None)) + sl] += w * prob[i]
            cnt[(b, 0) + sl] += w
    return out / cnt
