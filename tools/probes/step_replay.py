"""README-size training step replayed N times from the same state: output, loss and every parameter gradient must be
bitwise equal from run to run (fp32 and bf16 autocast).  usage: python tools/probes/step_replay.py [N]"""
import sys, json, torch
from torch import nn
sys.path.insert(0, ".")
import factorizer_amd as ft
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
DEV = "cuda:0"
torch.manual_seed(0)
model = ft.Factorizer(in_channels=4, out_channels=3, spatial_size=(128, 128, 128), norm=ft.LayerNorm,
                      reshape=(ft.SWMatricize, {"head_dim": 8, "patch_size": 8}), act=nn.ReLU,
                      factorize=ft.NMF, rank=1, num_iters=5, init="uniform", solver="hals", mlp_ratio=2,
                      dropout=0.1).to(DEV).eval()
x = torch.rand(2, 4, 128, 128, 128, device=DEV)
t = (torch.rand(2, 3, 128, 128, 128, device=DEV) > 0.5).float()
out = {}
for mode in ("f32", "bf16"):
    ref = None
    bad = 0
    for i in range(N):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode == "bf16")):
            y = model(x)
            loss = ft.dice_ce_loss(y, t)
        loss.backward()
        cur = [y.detach().clone(), loss.detach().clone()] + [p.grad.detach().clone() for p in model.parameters()]
        if ref is None:
            ref = cur
        else:
            bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(ref, cur))
    torch.cuda.synchronize()
    out[mode] = {"replays": N, "tensors_compared_per_replay": len(ref), "tensors_that_differed": bad}
print(json.dumps(out))
