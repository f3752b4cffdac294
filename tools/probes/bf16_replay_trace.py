"""Record the output of every native launch wrapper during two identical bf16 forward+backward runs of the cfg-5 model
(80x96x80, four stages) and report the first call whose output differs."""
import sys

import torch
from torch import nn

sys.path.insert(0, ".")
import factorizer_amd as ft  # noqa: E402
from factorizer_amd import functional as Fn  # noqa: E402
from factorizer_amd import pointwise as PW  # noqa: E402

dev = "cuda:0"
LOG = []
SYNC = "--nosync" not in sys.argv


def tensors(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (tuple, list)):
        return [t for x in o for t in tensors(x)]
    return []


def wrap(mod, name):
    orig = getattr(mod, name)

    def f(*a, **k):
        out = orig(*a, **k)
        if SYNC:
            torch.cuda.synchronize()
        if name == "_wgrad" and not SYNC:
            return out   # (written on the side stream: a clone on this stream would race with it)
        LOG.append((name + ":" + str(k.get("name", "")), [t.detach().float().clone() for t in tensors(out)]))
        return out
    setattr(mod, name, f)


_only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--main-wgrad=")]
if _only:
    _w = PW._wgrad

    def _wgrad_sel(*a, **k):
        if any(o in k.get("name", "") for o in _only) and torch.cuda.current_stream() != torch.cuda.default_stream():
            # launched from inside `with torch.cuda.stream(side)`: run it on the default stream instead, ordered after the side stream
            side = torch.cuda.current_stream()
            main = torch.cuda.default_stream()
            main.wait_stream(side)
            with torch.cuda.stream(main):
                out = _w(*a, **k)
            side.wait_stream(main)
            return out
        return _w(*a, **k)
    PW._wgrad = _wgrad_sel
for n in ("_gemm", "_wgrad", "_ln_backward", "_dgrad_lnbwd", "_gemm_dw", "_mlp_fwd_chain", "_mlp_bwd_chain", "_mlp_bwd_chain_wgrad"):
    wrap(PW, n)
_ob = Fn.FactCoreFn.backward
_of = Fn.FactCoreFn.forward


def cb(ctx, ga):
    if "--join-before-core" in sys.argv:
        for st in PW._SIDE.values():
            torch.cuda.current_stream().wait_stream(st)
    out = _ob(ctx, ga)
    if "--join-after-core" in sys.argv:
        for st in PW._SIDE.values():
            st.wait_stream(torch.cuda.current_stream())
    if SYNC:
        torch.cuda.synchronize()
    LOG.append(("core_bwd", [out[0].detach().float().clone()]))
    return out


def cf(ctx, *a):
    out = _of(ctx, *a)
    if SYNC:
        torch.cuda.synchronize()
    LOG.append(("core_fwd", [out.detach().float().clone()]))
    return out


Fn.FactCoreFn.backward = staticmethod(cb)
Fn.FactCoreFn.forward = staticmethod(cf)

S, widths, strides = (80, 96, 80), (32, 64, 128, 256), (1, 2, 2, 2)
if "--full" in sys.argv:
    S, widths, strides = (160, 192, 160), (32, 64, 128, 256, 512), (1, 2, 2, 2, 2)
torch.manual_seed(0)
model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=S, encoder_depth=(1,) * len(widths), encoder_width=widths, strides=strides,
                      decoder_depth=(1,) * (len(widths) - 1), norm=ft.LayerNorm, reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": (5, 6, 5)}),
                      act=nn.ReLU, factorize=ft.NMF, rank=2, num_iters=10, init="uniform", solver="hals", mlp_ratio=2,
                      dropout=0.0).to(dev)
x = torch.rand(1, 4, *S, device=dev)
t = (torch.rand(1, 3, *S, device=dev) > 0.5).float()
logs = []
for rep in range(2):
    LOG.clear()
    model.zero_grad(set_to_none=True)
    if "--fp32" in sys.argv:
        loss = ft.dice_ce_loss(model(x), t)
    else:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = ft.dice_ce_loss(model(x), t)
    loss.backward()
    torch.cuda.synchronize()
    logs.append(list(LOG))
a, b = logs
print("calls:", len(a), len(b), "sync after every call:", SYNC)
nd = 0
for i, ((n1, t1), (n2, t2)) in enumerate(zip(a, b)):
    assert n1 == n2
    for j, (u, v) in enumerate(zip(t1, t2)):
        if u.shape == v.shape and not torch.equal(u, v):
            d = (u - v).abs().max().item() / (u.abs().max().item() + 1e-30)
            print(f"  call {i:3d} {n1:32s} output {j} shape {tuple(u.shape)} differs: rel {d:.2e}, {int((u != v).sum())} elements")
            if nd == 0:
                idx = (u != v).nonzero()
                print("     indices:", idx[:12].tolist())
                print("     run0:", [round(u[tuple(k)].item(), 5) for k in idx[:12]])
                print("     run1:", [round(v[tuple(k)].item(), 5) for k in idx[:12]])
            nd += 1
            break
    if nd >= 6:
        break

print("differing outputs found:", nd)
