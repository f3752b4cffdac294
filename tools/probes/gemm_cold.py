"""Deep-stage fz_gemm launches hot (back to back: weights and activations still in L2 / Infinity Cache) against cold (a 1 GiB
streaming kernel between launches, as inside a training step, where ~10 ms of stage-0 traffic separates two uses of a weight),
and cold with the WEIGHTS touched just before the launch (a prefetch into the Infinity Cache).  HIP events per launch."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from factorizer_amd import pointwise as PW  # noqa: E402
DEV = "cuda:0"
B = 2
flush_src = torch.empty(256 * 1024 * 1024, device=DEV)      # 1 GiB
flush_dst = torch.empty_like(flush_src)


def flush():
    flush_dst.copy_(flush_src)


def timed(fn, pre=None, iters=12):
    ts = []
    for _ in range(iters):
        if pre is not None:
            pre()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for Cin, M, E, w_t in [(128, 128, 32, False), (128, 128, 32, True), (256, 256, 16, False), (256, 256, 16, True), (512, 512, 8, False),
                       (512, 512, 8, True), (512, 1024, 8, False), (1024, 512, 8, False), (256, 512, 16, False)]:
    V = E ** 3
    x = torch.randn(B, Cin, E, E, E, device=DEV)
    w2 = (torch.randn(M, Cin, device=DEV) / Cin ** 0.5) if not w_t else (torch.randn(Cin, M, device=DEV) / Cin ** 0.5)
    y = torch.empty(B, M, E, E, E, device=DEV)
    if w_t:
        fn = lambda: PW._gemm([x], w2, y, B=B, Cin=Cin, Vin=V, M=M, K=Cin, Ncol=V, w_t=True, ldw=M)  # noqa: E731
    else:
        fn = lambda: PW._gemm([x], w2, y, B=B, Cin=Cin, Vin=V, M=M, K=Cin, Ncol=V)  # noqa: E731
    for _ in range(3):
        fn()
    hot = timed(fn)
    cold = timed(fn, pre=flush)

    def flush_then_touch_w():
        flush()
        w2.sum()            # reads the weights: they are in the Infinity Cache when the launch starts

    def flush_then_touch_both():
        flush()
        w2.sum(); x.sum()
    warm_w = timed(fn, pre=flush_then_touch_w)
    warm_wx = timed(fn, pre=flush_then_touch_both)
    print(json.dumps({"shape": f"{Cin}->{M}@{E}^3" + (" (W^T)" if w_t else ""), "hot_us": round(hot, 1), "cold_us": round(cold, 1),
                      "cold_weights_touched_us": round(warm_w, 1), "cold_weights_and_x_touched_us": round(warm_wx, 1)}), flush=True)
