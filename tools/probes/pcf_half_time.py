"""nmf_pcf_fwd at the cfg-5 stage-0 launch (B x 32 x 160 x 192 x 160, patch (5,6,5): 8 x 150 matrices, HALS rank 2, 10 iterations),
fp32 and bf16 storage: ms per launch.  Run once with FZ_PCF_HALF=0 (one matrix per wave) and once without (two per wave)."""
import os, sys, json
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_)
import torch
from factorizer_amd import _native as N

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = "cuda:0"
torch.manual_seed(0)
C, S, P = 32, (160, 192, 160), (5, 6, 5)
R, T = 2, 10
u0 = torch.rand(8, R, device=dev); v0 = torch.rand(150, R, device=dev)
res = {}
for dt, ad in ((torch.float32, N.STORE_F32), (torch.bfloat16, N.STORE_BF16)):
    t = torch.rand(B, C, *S, device=dev).to(dt)
    out = torch.empty_like(t)
    arr = (N._i * 3)(0, 0, 0)
    def run():
        N.check(N.lib().fz_nmf_pcf_fwd(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), out.data_ptr(), B, C, *S, *P, arr, 0, 1, R, T,
                                       N.SOLVER_ID["hals"], 1e-8, ad, N.stream_ptr(t)), "pcf")
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / 10
    ga = torch.rand(B, C, *S, device=dev).to(dt)
    gt = torch.empty_like(t)
    def runb():
        N.check(N.lib().fz_nmf_pcf_bwd(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), ga.data_ptr(), gt.data_ptr(), B, C, *S, *P, arr, 0, 1, 1,
                                       R, T, T, N.SOLVER_ID["hals"], 1e-8, ad, N.stream_ptr(t)), "pcf bwd")
    for _ in range(2): runb()
    e0.record()
    for _ in range(5): runb()
    e1.record(); torch.cuda.synchronize()
    res[str(dt)] = {"fwd_ms": round(fwd_ms, 4), "bwd_ms": round(e0.elapsed_time(e1) / 5, 4), "checksum": float(out.float().double().sum()),
                    "checksum_bwd": float(gt.float().double().sum())}
    del ga, gt
print(json.dumps({"FZ_PCF_HALF": os.environ.get("FZ_PCF_HALF", "default(1)"), "B": B, **res}))
