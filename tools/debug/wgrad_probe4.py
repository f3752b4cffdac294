"""Register-operand weight-gradient kernel: fp32 MFMA vs the three-term bf16 split (FZ_WGRAD_BF3): time and error
against a float64 reference."""
import os, sys, torch
sys.path.insert(0, '.')
from factorizer_amd import pointwise as PW
DEV = 'cuda:0'
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B = 2
for (M, K, S) in ((32, 32, 128), (32, 64, 128), (64, 32, 128), (3, 32, 128), (64, 64, 64), (64, 128, 64), (128, 64, 64), (128, 128, 32), (128, 256, 32), (256, 256, 16), (512, 512, 8), (512, 1024, 8)):
    V = S ** 3
    torch.manual_seed(0)
    p = torch.randn(B, M, V, device=DEV); q = torch.randn(B, K, V, device=DEV)
    st = torch.rand(B, 2, V, device=DEV) + 0.5
    g, bt = torch.rand(K, device=DEV), torch.rand(K, device=DEV)
    ref = torch.einsum('bmv,bkv->mk', p.double(), q.double())
    out = []
    for bf in ("0", "1"):
        os.environ["FZ_WGRAD_BF3"] = bf
        gw = torch.empty(M, K, device=DEV); gb = torch.empty(M, device=DEV)
        t0 = timeit(lambda: PW._wgrad(p, [q], gw, B=B, M=M, Cin=K, K=K, Vq=V, Ncols=V, gbias=gb))
        err = (gw.double() - ref).abs().max().item() / ref.abs().max().item()
        t1 = timeit(lambda: PW._wgrad(p, [q], gw, B=B, M=M, Cin=K, K=K, Vq=V, Ncols=V, gbias=gb, stats=st, ln=(g, bt)))
        out.append(f"bf3={bf}: {t0*1e3:.1f} us err {err:.1e} (ln {t1*1e3:.1f} us)")
    print(f"{M:4d}x{K:4d} {S}^3: " + " | ".join(out))
