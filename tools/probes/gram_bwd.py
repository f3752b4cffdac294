"""Row-space (Gram) backward of the fused FactMixer core (csrc/nmf_cf_gram.hip, nmf_gram.h) against the general wave
program: values (relu_gate = 0 runs the general kernel; gated by hand) and launch times per window at the README sizes.
FZ_CF_GRAM=0 in the environment times the general kernel behind the same call.  JSON lines on stdout."""
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


def bwd(lib, t, u0, v0, ga, gt, B, C, S, sh, w, nshift, gate, T, G):
    arr = (N._i * 3)(*sh)
    rc = lib.fz_nmf_cf_bwd(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), ga.data_ptr(), gt.data_ptr(), B, C, *S, arr, int(w > 0), nshift,
                           gate, 1, T, G, 1, 1e-16, N.act_dtype(t), N.stream_ptr(t))
    assert rc == 0, lib.fz_last_error_string()


def check(B, C, S, dt, T, G, shifts):
    lib = N.lib()
    torch.manual_seed(1)
    t = torch.relu(torch.randn(B, C, *S, device=DEV)).to(dt)
    t[0, :8, :8, :8, :8] = 0
    ga = torch.randn(B, C, *S, device=DEV).to(dt)
    u0, v0 = torch.rand(8, 1, device=DEV), torch.rand(512, 1, device=DEV)
    g1, g0 = torch.empty_like(t), torch.empty_like(t)
    for w, sh in enumerate(shifts):
        bwd(lib, t, u0, v0, ga, g1, B, C, S, sh, w, len(shifts), 1, T, G)
        bwd(lib, t, u0, v0, ga, g0, B, C, S, sh, w, len(shifts), 0, T, G)
    ref = g0.float() * (t > 0)
    err = (g1.float() - ref).abs().max().item() / ref.abs().max().item()
    print(json.dumps({"check": [B, C, list(S)], "dtype": str(dt), "T": T, "G": G, "shifts": shifts, "rel_err_vs_general": err,
                      "finite": bool(torch.isfinite(g1.float()).all())}), flush=True)


def run(B, C, S, dt):
    lib = N.lib()
    pad = int(os.environ.get("PROBE_PLANE_PAD", "0"))   # with a library built -DFZ_PROBE_PLANE_PAD=<pad>: planes that far apart
    V = S[0] * S[1] * S[2]
    t = torch.relu(torch.randn(B, C, V + pad, device=DEV)).to(dt); ga = torch.randn(B, C, V + pad, device=DEV).to(dt)
    u0, v0 = torch.rand(8, 1, device=DEV), torch.rand(512, 1, device=DEV)
    gt = torch.empty_like(t)
    U = t.numel() * t.element_size()
    out = torch.empty_like(t)

    def fwd(sh, w):
        arr = (N._i * 3)(*sh)
        rc = lib.fz_nmf_cf_fwd(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), out.data_ptr(), B, C, *S, arr, int(w > 0), 2 if w else 1, 1, 5, 1,
                               1e-16, N.act_dtype(t), N.stream_ptr(t))
        assert rc == 0
    f0, f1 = timeit(lambda: fwd((0, 0, 0), 0)), timeit(lambda: fwd((4, 4, 4), 1))
    print(json.dumps({"B": B, "C": C, "S": list(S), "dtype": str(dt), "kernel": "forward", "w0_us": round(f0 * 1e3, 1), "w1_us": round(f1 * 1e3, 1),
                      "w0_TBps_alg": round(2 * U / f0 / 1e9, 3), "w1_TBps_alg": round(3 * U / f1 / 1e9, 3)}), flush=True)
    for gate in (1, 0):
        t0 = timeit(lambda: bwd(lib, t, u0, v0, ga, gt, B, C, S, (0, 0, 0), 0, 2, gate, 5, 5))
        t1 = timeit(lambda: bwd(lib, t, u0, v0, ga, gt, B, C, S, (4, 4, 4), 1, 2, gate, 5, 5))
        print(json.dumps({"B": B, "C": C, "S": list(S), "dtype": str(dt), "relu_gate": gate,
                          "kernel": "row space (gram)" if gate and os.environ.get("FZ_CF_GRAM") != "0" else "general wave program",
                          "w0_us": round(t0 * 1e3, 1), "w1_us": round(t1 * 1e3, 1), "w0_TBps_alg": round(3 * U / t0 / 1e9, 3),
                          "w1_TBps_alg": round(4 * U / t1 / 1e9, 3), "avg_frac_of_8TBps": round(3.5 * U / ((t0 + t1) / 2) / 1e9 / 8, 4)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        run(2, 32, (128, 128, 128), torch.float32)
        run(2, 64, (64, 64, 64), torch.float32)
        sys.exit(0)
    for dt in (torch.float32, torch.bfloat16):
        check(2, 16, (16, 16, 32), dt, 5, 5, [(0, 0, 0), (4, 4, 4)])
        check(1, 8, (8, 8, 32), dt, 4, 3, [(0, 0, 0), (2, 6, 2)])
        check(1, 8, (8, 16, 8), dt, 4, 1, [(0, 0, 0), (4, 0, 2)])
        check(1, 8, (8, 8, 32), dt, 1, 1, [(0, 0, 0)])
    run(2, 32, (128, 128, 128), torch.float32)
    run(2, 64, (64, 64, 64), torch.float32)
    run(2, 32, (128, 128, 128), torch.bfloat16)
