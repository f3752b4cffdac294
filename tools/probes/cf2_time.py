"""Times of the fused FactMixer core: one launch per window (rounds 1-3) against both windows in ONE slab-major launch
(fz_nmf_cf_fwd2 / fz_nmf_cf_bwd2) over the tune space {slices per group} x {lag} x {workgroups}.  JSON lines on stdout."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from factorizer_amd import _native as N

DEV = "cuda:0"


def timeit(fn, it=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


def run(B, C, S, dt):
    lib = N.lib()
    t = torch.rand(B, C, *S, device=DEV).to(dt); ga = torch.randn(B, C, *S, device=DEV).to(dt)
    u0, v0 = torch.rand(8, 1, device=DEV), torch.rand(512, 1, device=DEV)
    out, gt = torch.empty_like(t), torch.empty_like(t)
    sh = [(0, 0, 0), (4, 4, 4)]
    one = [(N._i * 3)(*s) for s in sh]; both = (N._i * 6)(*sh[0], *sh[1])
    ad, st = N.act_dtype(t), N.stream_ptr(t)
    ws = torch.empty(int(lib.fz_nmf_cf2_workspace_bytes(B, C, S[0])) // 4, dtype=torch.int32, device=DEV)
    U = t.numel() * t.element_size()

    def f2():
        for w in range(2):
            lib.fz_nmf_cf_fwd(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), out.data_ptr(), B, C, *S, one[w], int(w > 0), 2 if w else 1, 1, 5, 1, 1e-16, ad, st)
    def b2():
        for w in range(2):
            lib.fz_nmf_cf_bwd(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), ga.data_ptr(), gt.data_ptr(), B, C, *S, one[w], int(w > 0), 2, 1, 1, 5, 5, 1, 1e-16, ad, st)
    tf, tb = timeit(f2), timeit(b2)
    print(json.dumps({"B": B, "C": C, "S": S, "dtype": str(dt), "form": "one launch per window", "fwd_ms": round(tf, 4), "bwd_ms": round(tb, 4),
                      "fwd_TBps_alg": round(5 * U / tf / 1e9, 2), "bwd_TBps_alg": round(7 * U / tb / 1e9, 2)}), flush=True)
    nslice = B * C // 8
    groups = [g for g in (1, 2, 4, 8, 16) if g <= nslice and nslice % g == 0]
    for grp in [0] + groups:
        for lag in ([0] if grp == 0 else [0, 1, 2]):
            for wgs in ([0] if grp == 0 else [0, 256]):
                tn = (N._i * 3)(grp, lag if grp else -1, wgs)
                def f1():
                    rc = lib.fz_nmf_cf_fwd2(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), out.data_ptr(), B, C, *S, both, 1, 5, 1, 1e-16, ad, ws.data_ptr(), tn, st)
                    assert rc == 0, lib.fz_last_error_string()
                def b1():
                    rc = lib.fz_nmf_cf_bwd2(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), ga.data_ptr(), gt.data_ptr(), B, C, *S, both, 1, 1, 5, 5, 1, 1e-16, ad, ws.data_ptr(), tn, st)
                    assert rc == 0, lib.fz_last_error_string()
                tf1, tb1 = timeit(f1), timeit(b1)
                print(json.dumps({"B": B, "C": C, "S": S, "dtype": str(dt), "form": "both windows, one launch", "group": grp or "auto", "lag": lag if grp else "auto", "wgs": wgs or "resident",
                                  "fwd_ms": round(tf1, 4), "bwd_ms": round(tb1, 4), "fwd_TBps_alg": round(5 * U / tf1 / 1e9, 2),
                                  "bwd_TBps_alg": round(7 * U / tb1 / 1e9, 2), "timeout_word": int(ws[1])}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "short":
        run(2, 32, (128, 128, 128), torch.float32)
        sys.exit(0)
    run(2, 32, (128, 128, 128), torch.float32)
    run(2, 64, (64, 64, 64), torch.float32)
    run(2, 32, (128, 128, 128), torch.bfloat16)
