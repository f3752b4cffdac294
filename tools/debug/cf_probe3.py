"""nmf_cf forward: direct gather vs line-coalesced (LDS exchange) kernels, per stage shape."""
import os, sys, torch
sys.path.insert(0, '.')
import factorizer_amd as ft
from factorizer_amd import functional as Fn
DEV = 'cuda:0'
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (C, S) in ((32, 128), (64, 64), (128, 32)):
    m = ft.SWMatricize((None, C, S, S, S), head_dim=8, patch_size=8)
    geo = m.geometry
    t = torch.rand(2, C, S, S, S, device=DEV); u0 = torch.rand(8, 1, device=DEV); v0 = torch.rand(512, 1, device=DEV)
    ga = torch.rand_like(t)
    U = t.numel() * 4
    ref = None
    for T in (5,):
        for tile in ("0", "16", "8"):
            os.environ["FZ_CF_TILE"] = tile
            def fwd():
                ctx = type('X', (), {'save_for_backward': lambda self, *a: None})()
                return Fn.FactCoreFn.forward(ctx, t, u0, v0, geo, T, T, 'hals', 1e-16, True)
            y = fwd()
            if tile == "0": ref = y.clone()
            ok = torch.equal(ref, y)
            ms = timeit(fwd)
            print(f"C={C} S={S} T={T} tile={tile:>2}: {ms:.3f} ms {5*U/ms/1e6:.0f} GB/s bitexact={ok}")
    ctx = type('X', (), {})(); ctx.saved_tensors = (t, u0, v0); ctx.cfg = (geo, 5, 5, 'hals', 1e-16, True)
    for tile in ("0", "4", "8"):
        os.environ["FZ_CF_TILE_BWD"] = tile
        g = Fn.FactCoreFn.backward(ctx, ga)[0]
        if tile == "0": gref = g.clone()
        ms = timeit(lambda: Fn.FactCoreFn.backward(ctx, ga))
        print(f"C={C} S={S} bwd tile={tile}: {ms:.3f} ms {7*U/ms/1e6:.0f} GB/s maxdiff={(gref-g).abs().max().item():.2e}")
