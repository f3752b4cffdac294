"""nmf_pcf_bwd at the cfg-5 stage-0 launch with the arguments and the kind of data the model gives it (ReLU output: half zeros;
two windows: accumulate, gradient scale 1/2; ReLU gate) — against tools/probes/pcf_half_time.py's dense input."""
import os, sys, json
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_)
import torch
from factorizer_amd import _native as N
B = 4
dev = "cuda:0"
torch.manual_seed(0)
C, S, P = 32, (160, 192, 160), (5, 6, 5)
R, T = 2, 10
u0 = torch.rand(8, R, device=dev); v0 = torch.rand(150, R, device=dev)
res = {}
for name, zeros, shift, acc, nshift, gate in (("dense", False, (0, 0, 0), 0, 1, 0), ("relu data", True, (0, 0, 0), 0, 1, 0),
                                              ("relu data, model args window 0", True, (0, 0, 0), 0, 2, 1),
                                              ("relu data, model args window 1", True, (2, 3, 2), 1, 2, 1),
                                              ("shift only (2,3,2)", True, (2, 3, 2), 0, 1, 0), ("accumulate only", True, (0, 0, 0), 1, 1, 0),
                                              ("shift (0,0,2)", True, (0, 0, 2), 0, 1, 0), ("shift (2,3,0)", True, (2, 3, 0), 0, 1, 0),
                                              ("shift (0,0,5)", True, (0, 0, 5), 0, 1, 0)):
    t = torch.randn(B, C, *S, device=dev) if zeros else torch.rand(B, C, *S, device=dev)
    t = torch.relu(t).bfloat16()
    ga = torch.randn(B, C, *S, device=dev).bfloat16()
    gt = torch.zeros_like(t)
    arr = (N._i * 3)(*shift)
    def runb():
        N.check(N.lib().fz_nmf_pcf_bwd(t.data_ptr(), u0.data_ptr(), v0.data_ptr(), ga.data_ptr(), gt.data_ptr(), B, C, *S, *P, arr, acc, nshift, gate,
                                       R, T, T, N.SOLVER_ID["hals"], 1e-16, N.STORE_BF16, N.stream_ptr(t)), "pcf bwd")
    for _ in range(2): runb()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): runb()
    e1.record(); torch.cuda.synchronize()
    res[name] = round(e0.elapsed_time(e1) / 5, 3)
    del t, ga, gt
print(json.dumps({"FZ_PCF_HALF": os.environ.get("FZ_PCF_HALF", "1"), "bwd_ms": res}))
